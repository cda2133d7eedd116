/* ref_shim/tools/SL_AVIReader.h -- stand-in (see math/SL_Matrix.h): CoSLAM::init casts its readers to this
 * (src/app/SL_CoSLAM.cpp:76-77); nothing compiled for the tests decodes video. */
#ifndef REF_SHIM_SL_AVIREADER_H
#define REF_SHIM_SL_AVIREADER_H
#include <string>
#include "tools/SL_VideoReader.h"
class AVIReader : public VideoReader {
public:
    std::string filePath;
};
#endif
