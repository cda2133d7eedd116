/* ref_shim/geometry/SL_BundleHelper.h -- stand-in (see math/SL_Matrix.h): the sba-style helpers are not on the call path. */
#ifndef REF_SHIM_SL_BUNDLEHELPER_H
#define REF_SHIM_SL_BUNDLEHELPER_H
struct sbaGlobs { /* members of the abandoned sba estimator classes only */
    double* intrcalib;
    int nccalib, ncdist, cnp, pnp, mnp;
    double *rot0params, *camparams, *ptparams;
};
#endif
