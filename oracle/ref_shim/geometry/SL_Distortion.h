/* ref_shim/geometry/SL_Distortion.h -- stand-in (see math/SL_Matrix.h).  undistorPoint is external to the reference
 * (only its call, src/tracking/GPUKLT.cpp:45, and the 7-vector k_ud, src/tracking/GPUKLT.h:44-47, are in the tree):
 * the definition here is OURS, the same one coslam_amd/csrc/handback.hip and oracle/handback_oracle.c use -- normalise
 * with K, scale by 1 + sum_i k_ud[i] r^(2(i+1)), map back with K; k_ud = 0 is the identity. */
#ifndef REF_SHIM_SL_DISTORTION_H
#define REF_SHIM_SL_DISTORTION_H
void undistorPoint(const double* K, const double* kud, const double* in, double* out);
#endif
