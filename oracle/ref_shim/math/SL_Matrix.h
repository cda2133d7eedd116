/* ref_shim/math/SL_Matrix.h -- stand-in for the un-vendored LibVisualSLAM header of the same name (see SL_LinAlg.h).
 * TEST INFRASTRUCTURE: lets oracle/Makefile compile the reference's own callers (src/tracking/GPUKLT.cpp,
 * src/app/SL_CoSLAMRobustBA.cpp, src/app/SL_InterCamPoseEstimator.cpp and the data-model sources they need) in place.
 * Only what those files use: a row-major dense matrix owning its storage, with the members the reference touches
 * (data, rows, cols, m, n; resize, cloneFrom, fill, operator T*, operator[]). */
#ifndef REF_SHIM_SL_MATRIX_H
#define REF_SHIM_SL_MATRIX_H
#include <cassert>
#include <cstddef>
#include <cstring>
#include <map>
#include <vector>
using namespace std; /* the real header evidently leaks it: src/slam/SL_MapPoint.h:131 uses an unqualified `vector` */

#include "SL_error.h"

typedef unsigned char uchar;

template <class T>
class MyMat {
public:
    int rows, cols;
    int& m;  // LibVisualSLAM exposes both spellings
    int& n;
    T* data;
    MyMat() : rows(0), cols(0), m(rows), n(cols), data(0) {}
    MyMat(int r, int c) : rows(0), cols(0), m(rows), n(cols), data(0) { resize(r, c); }
    MyMat(int r, int c, const T* src) : rows(0), cols(0), m(rows), n(cols), data(0) { cloneFrom(src, r, c); }
    MyMat(const MyMat& o) : rows(0), cols(0), m(rows), n(cols), data(0) { cloneFrom(o.data, o.rows, o.cols); }
    MyMat& operator=(const MyMat& o) {
        if (this != &o) cloneFrom(o.data, o.rows, o.cols);
        return *this;
    }
    ~MyMat() { delete[] data; }
    void clear() {
        delete[] data;
        data = 0;
        rows = cols = 0;
    }
    void resize(int r, int c) {
        if ((size_t)r * c != (size_t)rows * cols || !data) {
            delete[] data;
            data = ((size_t)r * c > 0) ? new T[(size_t)r * c]() : 0;
        }
        rows = r;
        cols = c;
    }
    void cloneFrom(const T* src, int r, int c) {
        resize(r, c);
        if (src && data) memcpy(data, src, sizeof(T) * (size_t)r * c);
    }
    void cloneFrom(const MyMat& o) { cloneFrom(o.data, o.rows, o.cols); }
    void fill(T v) {
        for (size_t i = 0; i < (size_t)rows * cols; ++i) data[i] = v;
    }
    bool empty() const { return data == 0 || rows * cols == 0; }
    operator T*() { return data; }
    operator const T*() const { return data; }
    T& operator()(int r, int c) { return data[(size_t)r * cols + c]; }
    const T& operator()(int r, int c) const { return data[(size_t)r * cols + c]; }
};
typedef MyMat<double> Mat_d;
typedef MyMat<float> Mat_f;
typedef MyMat<int> Mat_i;
typedef MyMat<char> Mat_c;
typedef MyMat<unsigned char> Mat_uc;
typedef MyMat<unsigned int> Mat_ui;
#endif
