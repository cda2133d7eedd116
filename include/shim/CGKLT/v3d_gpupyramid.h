// -*- C++ -*-
// include/shim/CGKLT/v3d_gpupyramid.h -- the reference's GPUKLT.h includes this header; the pyramid now lives
// inside libcoslam_hip.so (coslam_amd/csrc/klt_pyramid.hip) and needs no public type.
#ifndef V3D_GPU_PYRAMID_H
#define V3D_GPU_PYRAMID_H
#endif
