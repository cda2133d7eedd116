/* ref_shim/imgproc/SL_ImageOp.h -- stand-in (see math/SL_Matrix.h): image operations are not on the call path. */
#ifndef REF_SHIM_SL_IMAGEOP_H
#define REF_SHIM_SL_IMAGEOP_H
#include "imgproc/SL_Image.h"
template <class IMG>
void cloneImg(const IMG& src, IMG& dst) {
    dst.resize(src.w, src.h);
    for (int i = 0; i < src.w * src.h; ++i) dst.data[i] = src.data[i];
}
#endif
