/* ref_shim/imgproc/SL_Image.h -- stand-in (see math/SL_Matrix.h): image containers the reference's headers mention as
 * members / parameters; nothing on the tracker or BA call path reads pixels through them except SingleSLAM::m_img.data
 * (the 8-bit image handed to GPUKLT::next, src/app/SL_SingleSLAM.cpp:329-331). */
#ifndef REF_SHIM_SL_IMAGE_H
#define REF_SHIM_SL_IMAGE_H
#include "math/SL_Matrix.h"
template <int CH>
class ImgU8 {
public:
    int w, h, m, n, rows, cols;
    unsigned char* data;
    ImgU8() : w(0), h(0), m(0), n(0), rows(0), cols(0), data(0) {}
    ~ImgU8() { delete[] data; }
    void clear() {
        delete[] data;
        data = 0;
        w = h = m = n = rows = cols = 0;
    }
    void resize(int W, int H) {
        delete[] data;
        data = new unsigned char[(size_t)W * H * CH]();
        w = n = cols = W;
        h = m = rows = H;
    }
    bool empty() const { return data == 0; }
    unsigned char* operator()(int x, int y) { return data + (size_t)CH * ((size_t)y * w + x); }
    operator unsigned char*() { return data; }
    operator const unsigned char*() const { return data; }
private:
    ImgU8(const ImgU8&);
    ImgU8& operator=(const ImgU8&);
};
typedef ImgU8<1> ImgG;
typedef ImgU8<3> ImgRGB;
#endif
