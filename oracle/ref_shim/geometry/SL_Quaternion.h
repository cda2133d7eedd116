/* ref_shim/geometry/SL_Quaternion.h -- stand-in (see math/SL_Matrix.h). */
#ifndef REF_SHIM_SL_QUATERNION_H
#define REF_SHIM_SL_QUATERNION_H
#endif
