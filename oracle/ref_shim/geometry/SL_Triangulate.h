/* ref_shim/geometry/SL_Triangulate.h -- SL_IntraCamPose.cpp includes it but calls nothing from it. */
#ifndef REF_SHIM_SL_TRIANGULATE_H
#define REF_SHIM_SL_TRIANGULATE_H
#endif
