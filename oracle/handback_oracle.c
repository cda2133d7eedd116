/*
 * oracle/handback_oracle.c -- CPU restatement of the host glue between the tracker and the pose solve.
 *
 * TEST INFRASTRUCTURE ONLY (see klt_oracle.h).  Follows, statement by statement:
 *   GPUKLT::addToFeaturePoints        /root/reference/src/tracking/GPUKLT.cpp:36-60
 *   SingleSLAM::chooseStaticFeatPts   /root/reference/src/app/SL_SingleSLAM.cpp:345-397
 *   SingleSLAM::poseUpdate3D          /root/reference/src/app/SL_SingleSLAM.cpp:620-640 (the Ms / ms packing)
 * with the pointer lists (FeaturePoints, Track2D) reduced to what those statements read: per slot the track length, the
 * tail point's undistorted pixel and its map point.
 *
 * PARITY UNPINNED for undistorPoint: it lives in un-vendored LibVisualSLAM (only the call, GPUKLT.cpp:45, and the
 * 7-vector k_ud, GPUKLT.h:44-47, are in the reference).  Definition used here and in coslam_amd/csrc/handback.hip:
 * normalise with K, scale by 1 + sum_{i=0..6} k_ud[i] r^(2(i+1)), map back with K; k_ud = 0 is the identity.
 */
#include <stdlib.h>
#include <string.h>

#include "klt_oracle.h"

void ohb_undistort_point(const double K[9], const double kud[7], const double in[2], double out[2]) {
    const double yn = (in[1] - K[5]) / K[4];
    const double xn = ((in[0] - K[2]) - K[1] * yn) / K[0];
    const double r2 = xn * xn + yn * yn;
    double f = kud[6];
    for (int i = 5; i >= 0; --i) f = f * r2 + kud[i];
    f = 1.0 + f * r2;
    const double xu = xn * f, yu = yn * f;
    out[0] = (K[0] * xu + K[1] * yu) + K[2];
    out[1] = K[4] * yu + K[5];
}

/* One frame (number `frame`) of one camera.  In/out per slot: slot2map (FeaturePoint::mpt of the track's tail, -1 = none),
 * trackSpan (Track2D::f1 / f2: first[N] then last[N] frame, -1 = empty; length() = f2 - f1 + 1, SL_Track2D.h:63-65),
 * xy (tail point, x[N] then y[N]).  Out: state[N] (0 tracked, 1 new, -1 dead, -2 dropped),
 * selBlk[nColBlk * nRowBlk] (featPts of chooseStaticFeatPts in block order, -1 = none), and the packed 3D-2D
 * correspondences (at most ptsStride).  Returns the number of correspondences. */
int ohb_handback(int N, int W, int H, int frame, const okl_tracked_feature* features, const double K[9], const double kud[7],
                 const double* mapPts, const unsigned char* isStatic, int* slot2map, int* trackSpan, double* xy, int* state,
                 int nColBlk, int nRowBlk, int* selBlk, int ptsStride, double* Ms, double* ms, int* sel) {
    int* tf1 = trackSpan;
    int* tf2 = trackSpan + N;
    /* ---- GPUKLT::addToFeaturePoints, GPUKLT.cpp:36-60 */
    for (int i = 0; i < N; i++) {
        if (features[i].status >= 0) {
            double in[2], out[2];
            in[0] = features[i].pos[0] * W; /* float * int -> float, then widened (:43-44) */
            in[1] = features[i].pos[1] * H;
            ohb_undistort_point(K, kud, in, out); /* :45 */
            if (out[0] >= W || out[1] >= H) {     /* :46-47 continue: neither added nor cleared */
                state[i] = -2;
                continue;
            }
            xy[i] = out[0]; /* ips.add(m_frame, m_camId, out[0], out[1]) */
            xy[N + i] = out[1];
            if (features[i].status != 0) { /* m_tks[i].clear(): a new, unmapped feature point starts the track */
                tf1[i] = tf2[i] = -1;
                slot2map[i] = -1;
            }
            /* m_tks[i].add(p), SL_Track2D.h:78-104: the frame span grows to this frame */
            if (frame < tf1[i] || tf1[i] < 0) tf1[i] = frame;
            if (frame > tf2[i] || tf2[i] < 0) tf2[i] = frame;
            state[i] = features[i].status;
        } else {
            tf1[i] = tf2[i] = -1; /* m_tks[i].clear() */
            slot2map[i] = -1;
            state[i] = -1;
        }
    }
    /* ---- SingleSLAM::chooseStaticFeatPts, SL_SingleSLAM.cpp:345-397 */
    const int blkW = W / nColBlk, blkH = H / nRowBlk; /* :270-271 */
    const int len = nRowBlk * nColBlk;
    int* tracks = (int*)malloc(sizeof(int) * (size_t)len);
    for (int b = 0; b < len; b++) tracks[b] = -1;
    for (int i = 0; i < N; i++) {
        if (tf1[i] < 0) continue; /* tk->empty() */
        const int mapped = slot2map[i] >= 0;
        /* fp->type == STATIC || (fp->mpt && fp->mpt->isCertainStatic()); a track born in this frame has no predecessor to take a
         * type from (propagateFeatureStates, :40-42) and keeps the constructor's type(0) = STATIC (SL_FeaturePoint.cpp:23),
         * whatever the slot's previous track left in isStatic[] */
        if (mapped || (isStatic && (isStatic[i] || tf1[i] == frame))) {
            int bx = (int)(xy[i] / blkW);
            int by = (int)(xy[N + i] / blkH);
            if (bx >= nColBlk || by >= nRowBlk) continue;
            if (bx < 0 || by < 0) continue; /* (the reference indexes out of bounds here; never happens for tracked points) */
            int bi = by * nColBlk + bx;
            int old = tracks[bi];
            if (old < 0) {
                tracks[bi] = i;
            } else if (!(slot2map[old] >= 0)) { /* !fpOld->mpt */
                if (mapped) {
                    tracks[bi] = i;
                } else if (tf2[old] - tf1[old] + 1 < tf2[i] - tf1[i] + 1) { /* tkOld->length() < tk->length() */
                    tracks[bi] = i;
                }
            }
        }
    }
    /* ---- poseUpdate3D, :620-640: the mapped ones, in featPts order */
    int n = 0;
    for (int b = 0; b < len; b++) {
        if (selBlk) selBlk[b] = tracks[b];
        int i = tracks[b];
        if (i < 0 || slot2map[i] < 0) continue;
        if (n < ptsStride) {
            sel[n] = i;
            ms[2 * n] = xy[i];
            ms[2 * n + 1] = xy[N + i];
            memcpy(Ms + 3 * n, mapPts + 3 * (size_t)slot2map[i], sizeof(double) * 3);
        }
        n++;
    }
    free(tracks);
    return n < ptsStride ? n : ptsStride;
}

/* MapPoint::pFeatures[iCam] restricted to the current frame, as a table column: for map points 0..P-1 the slot of this
 * camera's feature of THIS frame (state 0 or 1) whose FeaturePoint::mpt is the point, else -1 -- what the registration
 * loops test with `p->pFeatures[iCam] && p->pFeatures[iCam]->f == curFrame` (SL_CoSLAM.cpp:737-738).  When two slots
 * carry the same point (the reference's pointer can only hold one) the higher slot is reported. */
void ohb_point_features(int N, const int* state, const int* slot2map, int P, int stride, int* pointFeat) {
    for (int p = 0; p < P; p++) pointFeat[(size_t)p * stride] = -1;
    for (int i = 0; i < N; i++) {
        const int mp = slot2map[i];
        if (mp >= 0 && mp < P && (state[i] == 0 || state[i] == 1)) pointFeat[(size_t)mp * stride] = i;
    }
}
