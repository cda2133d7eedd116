/* ref_shim/tools/SL_WriteRead.h -- stand-in (see math/SL_Matrix.h): matrix text I/O is not on the call path. */
#ifndef REF_SHIM_SL_WRITEREAD_H
#define REF_SHIM_SL_WRITEREAD_H
#endif
