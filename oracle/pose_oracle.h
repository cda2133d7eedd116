/* oracle/pose_oracle.h -- CPU restatement of intraCamEstimate (TEST INFRASTRUCTURE, see pose_oracle.c). */
#ifndef COSLAM_POSE_ORACLE_H
#define COSLAM_POSE_ORACLE_H
#ifdef __cplusplus
extern "C" {
#endif

/* == class IntraCamPoseOption, src/slam/SL_IntraCamPose.h:19-57 */
typedef struct okp_option {
    int maxIterLM, maxIterRW;
    double epsErrorChangeLM, epsParamChangeLM, epsErrorChangeRW;
    int verboseLM, verboseRW;
    double lambda0, lambda;
    double err0, err, errRW;
    int retTypeLM, npts, nIterLM, nIterRW;
} okp_option;

void okp_option_default(okp_option* o);
void okp_so3_exp(const double w[3], double R[9]);
void okp_project(const double* K, const double* R, const double* t, const double* M, double* m);
void okp_mat_inv(int n, const double* A, double* invA);
int okp_weighted_lm(const double* K, const double* R0, const double* t0, int npts, const double* Ws, const double* Ms,
                    const double* ms, double* R_opt, double* t_opt, okp_option* opt);
int okp_intracam_estimate(const double* K, const double* R0, const double* t0, int npts, const double* prevErrs,
                          const double* Ms, const double* ms, double tau, double* R_opt, double* t_opt,
                          okp_option* opt);
#ifdef __cplusplus
}
#endif
#endif
