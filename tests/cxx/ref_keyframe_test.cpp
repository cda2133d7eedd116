// ref_keyframe_test.cpp -- the reference's OWN key-frame decision: CoSLAM::IsReadyForKeyFrame (src/app/SL_CoSLAM.cpp:1269-1279) with
// getCurMapCenterViewFrom (:1224-1247), IsMappedPtsDecreaseBelow (:1249-1268), SingleSLAM::getNumMappedStaticPts
// (src/app/SL_SingleSLAM.cpp:121-136), getViewAngleChangeSelf / getCameraTranslationSelf (:825-834) and getCamDist / getViewAngleChange
// (src/slam/SL_SLAMHelper.cpp:201-217), all compiled in place (oracle/Makefile), on cameras whose state is built with the reference's
// classes: this frame's feature points with their map points, the last key pose (frame, nMappedPts), the last self-motion key pose, the
// current pose.  Writes the scenes and the reference's answers for tests/golden/make_golden.py (CPU only).
//   ref_keyframe_test golden <out.bin>
// Layout: int32 nScenes; per scene: int32 nCams, N, nMap, curFrame; double ratio, minViewAngle, minTranslation; per map point M[3],
// int32 localType, uncertain, firstFrame; per camera: R[9], t[3] (current), selfR[9], selfT[3], int32 keyFrame, keyMapped, then N x
// int32 map index of the slot's feature (-2: the slot has no feature in this frame, -1: a feature without a map point); then per camera
// the reference's int32 ready, nMappedStatic, double center[3].
// The features are added to SingleSLAM::m_featPts in slot order.  Both loops that walk them stop BEFORE the frame's last feature
// (`fp && fp != pTail`, :1233, :1258) while getNumMappedStaticPts includes it -- the vectors pin that.
// TEST INFRASTRUCTURE; built into oracle/_ref/ where the reference tree exists.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "app/SL_CoSLAM.h"
#include "app/SL_GlobParam.h"

static unsigned long long g_rng = 0x9E3779B97F4A7C15ull;
static double urand() {
    g_rng ^= g_rng << 13;
    g_rng ^= g_rng >> 7;
    g_rng ^= g_rng << 17;
    return (double)(g_rng >> 11) / 9007199254740992.0;
}
static void rodrigues(const double w[3], double R[9]) {
    const double th = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
    const double k[3] = {th > 0 ? w[0] / th : 0, th > 0 ? w[1] / th : 0, th > 0 ? w[2] / th : 0};
    const double c = cos(th), s = sin(th), v = 1 - c;
    const double M[9] = {c + k[0] * k[0] * v,        k[0] * k[1] * v - k[2] * s, k[0] * k[2] * v + k[1] * s,
                         k[1] * k[0] * v + k[2] * s, c + k[1] * k[1] * v,        k[1] * k[2] * v - k[0] * s,
                         k[2] * k[0] * v - k[1] * s, k[2] * k[1] * v + k[0] * s, c + k[2] * k[2] * v};
    memcpy(R, M, sizeof(M));
}
template <class T> static void put(FILE* f, const T* p, size_t n) { fwrite(p, sizeof(T), n, f); }
static void puti(FILE* f, int v) { fwrite(&v, 4, 1, f); }

int main(int argc, char** argv) {
    if (argc < 3 || strcmp(argv[1], "golden")) {
        fprintf(stderr, "usage: %s golden <out.bin>\n", argv[0]);
        return 2;
    }
    FILE* f = fopen(argv[2], "wb");
    if (!f) return 1;
    const int nScenes = 6;
    puti(f, nScenes);
    int hist[4] = {0, 0, 0, 0}, nSmall = 0;
    for (int sc = 0; sc < nScenes; ++sc) {
        const int nCams = 3 + sc % 4, N = 400, nMap = 300, curFrame = 400 + 7 * sc;
        CoSLAM* co = new CoSLAM();
        co->numCams = nCams;
        co->curFrame = curFrame;
        co->m_mappedPtsReduceRatio = 0.93;        // (the constructor's values, :37-38; the translation threshold is set at start-up, :208 / :291)
        co->m_minViewAngleChange = 5.0;
        co->m_minCamTranslation = 0.05 + 0.1 * urand();
        puti(f, nCams), puti(f, N), puti(f, nMap), puti(f, curFrame);
        put(f, &co->m_mappedPtsReduceRatio, 1), put(f, &co->m_minViewAngleChange, 1), put(f, &co->m_minCamTranslation, 1);
        std::vector<MapPoint*> pts(nMap);
        for (int p = 0; p < nMap; ++p) {
            const int first = curFrame - (int)(urand() * 60);
            MapPoint* mp = new MapPoint(-2 + 4 * urand(), -1.5 + 3 * urand(), 6 + 5 * urand(), first);
            const double u = urand();
            if (u < 0.15) mp->setLocalDynamic();
            else if (u < 0.25) mp->setFalse();
            else mp->setLocalStatic();
            if (urand() < 0.12) mp->setUncertain();
            put(f, mp->M, 3);
            puti(f, mp->iLocalType), puti(f, mp->bUncertain ? 1 : 0), puti(f, mp->firstFrame);
            pts[p] = mp;
        }
        for (int c = 0; c < nCams; ++c) {
            SingleSLAM* s = &co->slam[c];
            s->camId = c;
            s->m_tracker.m_frame = curFrame;
            // the current pose, the pose of the last self-motion key frame (a step back along the camera's path: from none to a few
            // degrees of view angle / a few centimetres), the last key pose
            const int kind = (c + sc) % 4;   // barely moved / a few centimetres (translation) / a wide step (view angle) / anything
            const double stepBack = kind == 0 ? 0.002 : (kind == 1 ? 0.2 + 0.4 * urand() : (kind == 2 ? 0.9 + 0.6 * urand() : 1.5 * urand()));
            double wS[3] = {0.01 * c, 0.1 * c - 0.05 * stepBack, 0.0}, wC[3] = {0.01 * c, 0.1 * c, 0.0}, RS[9], RC[9], tS[3], tC[3];
            rodrigues(wS, RS), rodrigues(wC, RC);
            const double posC[3] = {1.2 * c, 0.05 * c, 0}, posS[3] = {1.2 * c - stepBack, 0.05 * c + 0.1 * stepBack, -0.2 * stepBack};
            for (int r = 0; r < 3; ++r) {
                tC[r] = -(RC[3 * r] * posC[0] + RC[3 * r + 1] * posC[1] + RC[3 * r + 2] * posC[2]);
                tS[r] = -(RS[3 * r] * posS[0] + RS[3 * r + 1] * posS[1] + RS[3 * r + 2] * posS[2]);
            }
            const int keyFrame = curFrame - 3 - (int)(urand() * 25);
            CamPoseItem* camS = s->m_camPos.add(keyFrame - 4, c, RS, tS);
            CamPoseItem* camK = s->m_camPos.add(keyFrame, c, RS, tS);
            s->m_camPos.add(curFrame, c, RC, tC);
            KeyPose* kpS = s->m_keyPose.add(keyFrame - 4, camS);
            kpS->bSelfMotion = true;
            s->m_selfKeyPose.push_back(kpS);
            KeyPose* kp = s->m_keyPose.add(keyFrame, camK);
            // the features of this frame, slot order: some slots empty, some features unmapped; the share of mapped ones varies per camera
            const double pMapped = 0.2 + 0.7 * urand(), pEmpty = (c == nCams - 1 && sc % 2) ? 0.85 : 0.1;   // (one camera with few features: num < 30)
            std::vector<int> mapOf(N, -2);
            int num = 0;
            for (int k = 0; k < N; ++k) {
                if (urand() < pEmpty) continue;
                int m = -1;
                if (urand() < pMapped) m = (int)(urand() * nMap) % nMap;
                mapOf[k] = m;
                FeaturePoint* fp = s->m_featPts.add(curFrame, c, 10 + 6 * k, 20 + k);
                fp->mpt = m >= 0 ? pts[m] : nullptr;
                if (m >= 0 && pts[m]->firstFrame <= keyFrame) ++num;
            }
            // nMappedPts of the last key pose around the count of this frame, so that the 0.93 test falls on both sides
            const int keyMapped = (int)(num * (0.7 + 0.45 * urand()));
            kp->setNumMappedPoints(keyMapped);
            put(f, RC, 9), put(f, tC, 3), put(f, RS, 9), put(f, tS, 3);
            puti(f, keyFrame), puti(f, keyMapped);
            put(f, mapOf.data(), N);
            if (num < 30) ++nSmall;
        }
        for (int c = 0; c < nCams; ++c) {
            const int nStatic = co->slam[c].getNumMappedStaticPts();
            const int ready = co->IsReadyForKeyFrame(c);
            double center[3];
            co->getCurMapCenterViewFrom(c, center);
            puti(f, ready), puti(f, nStatic);
            put(f, center, 3);
            ++hist[ready & 3];
        }
    }
    fclose(f);
    printf("ref_keyframe_test: %d scenes; IsReadyForKeyFrame said 0 / decrease / view angle / translation for %d / %d / %d / %d cameras (%d with fewer than 30 mapped features)\n",
           nScenes, hist[0], hist[1], hist[2], hist[3], nSmall);
    return (hist[0] > 0 && hist[1] > 1 && hist[2] > 1 && hist[3] > 1) ? 0 : 1;
}
