/* ref_shim/SL_error.h -- stand-in (see math/SL_Matrix.h): repErr throws like LibVisualSLAM's (callers catch SL_Exception,
 * src/gui/CoSLAMThread.cpp:133-144), warn / logInfo print. */
#ifndef REF_SHIM_SL_ERROR_H
#define REF_SHIM_SL_ERROR_H
#include <cassert>
#include <cstdarg>
#include <cstdio>
#include <iostream>
#include <stdexcept>
#include <string>
class SL_Exception : public std::runtime_error {
public:
    explicit SL_Exception(const std::string& s) : std::runtime_error(s) {}
};
void repErr(const char* fmt, ...);
void warn(const char* fmt, ...);
void logInfo(const char* fmt, ...);
/* printf-style path formatting of the write / read helpers (src/slam/SL_GlobalPoseEstimation.cpp:1365) */
#define GET_FMT_STR(fmtstr, buf) \
    {                            \
        va_list args;            \
        va_start(args, fmtstr);  \
        vsprintf(buf, fmtstr, args); \
        va_end(args);            \
    }
#endif
