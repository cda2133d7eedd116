/* ref_shim stand-in (see math/SL_Matrix.h): declarations only (CoSLAM::saveCurrentImages, off the call path) */
#ifndef REF_SHIM_SL_IMAGEIO_H
#define REF_SHIM_SL_IMAGEIO_H
#include "imgproc/SL_Image.h"
void savePGM(const ImgG& img, const char* path);
#endif
