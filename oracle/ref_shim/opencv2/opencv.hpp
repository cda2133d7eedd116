/* ref_shim/opencv2/opencv.hpp -- stand-in: declarations only (NO definitions) of the few OpenCV names
 * src/slam/SL_NCCBlock.cpp mentions in getNCCBlock / getNCCBlocks / getScaledNCCBlocks (cv::getRectSubPix on a cv::resize'd
 * image).  Those functions are not on the path under test (NCCBlock::compute, matchNCCBlock are); the objects are built with
 * -ffunction-sections and linked with --gc-sections, so nothing here is ever resolved.  TEST INFRASTRUCTURE. */
#ifndef REF_SHIM_OPENCV_HPP
#define REF_SHIM_OPENCV_HPP
#include "ref_not_on_path.h"
#endif
