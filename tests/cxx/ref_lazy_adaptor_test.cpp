// ref_lazy_adaptor_test.cpp -- SURVEY 8f-1, second half: the lazy FeaturePoints / Track2D adaptor (include/shim/tracking/GPUKLTGroup.h)
// against the reference's own synchronous facade.
//
// oracle/Makefile compiles /root/reference/src/tracking/GPUKLT.cpp, SL_Track2D.cpp, src/slam/SL_FeaturePoint(s).cpp, SL_MapPoint.cpp and
// src/app/SL_SingleSLAM.cpp IN PLACE over include/shim/.  Two rigs of three SingleSLAM objects see the same 14 frames:
//   rig A, as the reference runs: per camera m_tracker.first / next(img, m_featPts) -- redetect + read-back + addToFeaturePoints per call;
//   rig B, the group path: GPUKLTGroup::first / next enqueue ONE set of launches per frame for all cameras and touch no list; sync() is
//          called only at frames 4, 9 and 13 and replays what it has not seen.
// At each of those frames both rigs' host state must be indistinguishable to the reference's own consumers:
//   every Track2D (empty / f1 / f2 / length, every node's frame and pixel, the FeaturePoint preFrame / nextFrame chain),
//   every frame's FeaturePoints list (count and order), and -- after the same labelling of the tails' types and map points --
//   SingleSLAM::chooseStaticFeatPts (src/app/SL_SingleSLAM.cpp:345-397) and getNumMappedStaticPts (:121-136) called on both.
// TEST INFRASTRUCTURE; run by tests/test_cxx_dropin_gpu.py on the GPU box.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "app/SL_SingleSLAM.h"
#include "tracking/GPUKLTGroup.h"

#define CHECK(c)                                                           \
    do {                                                                   \
        if (!(c)) {                                                        \
            fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); \
            return 1;                                                      \
        }                                                                  \
    } while (0)

static unsigned long long g_rng = 0x9E3779B97F4A7C15ull;
static double urand() {
    g_rng ^= g_rng << 13, g_rng ^= g_rng >> 7, g_rng ^= g_rng << 17;
    return (double)(g_rng >> 11) / 9007199254740992.0;
}
static void render(int W, int H, const std::vector<double>& bx, const std::vector<double>& by, const std::vector<double>& amp, double dx,
                   double dy, unsigned char* img) {
    std::vector<double> acc((size_t)W * H, 110.0);
    for (size_t k = 0; k < bx.size(); ++k) {
        const double u = bx[k] + dx, v = by[k] + dy;
        const int ci = (int)floor(u), cj = (int)floor(v);
        for (int j = cj - 5; j <= cj + 5; ++j)
            for (int i = ci - 5; i <= ci + 5; ++i) {
                if (i < 0 || j < 0 || i >= W || j >= H) continue;
                const double d2 = (i + 0.5 - u) * (i + 0.5 - u) + (j + 0.5 - v) * (j + 0.5 - v);
                acc[(size_t)j * W + i] += amp[k] * exp(-d2 / (2 * 1.3 * 1.3));
            }
    }
    for (size_t p = 0; p < acc.size(); ++p) {
        double v = floor(acc[p] + 0.5);
        img[p] = (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v));
    }
}

// the labelling a frame's pose / mapping code would leave on the tails (the same function of (camera, slot, frame) for both rigs)
static void label(SingleSLAM& s, std::vector<MapPoint>& mpts, int frame) {
    for (int i = 0; i < s.m_tracker.m_nMaxCorners; ++i) {
        Track2D& tk = s.m_tracker.m_tks[i];
        if (tk.empty() || tk.tail->f != frame) continue;
        FeaturePoint* fp = tk.tail->pt;
        fp->type = ((i + frame + s.camId) % 3) ? TYPE_FEATPOINT_STATIC : TYPE_FEATPOINT_DYNAMIC;
        if (i % 4 == 0 && tk.length() >= 2) {
            MapPoint* m = &mpts[i];
            if (i % 8 == 0) m->setLocalStatic();
            else m->setLocalDynamic();
            fp->mpt = m;
            m->pFeatures[s.camId] = fp;
        }
    }
}

static bool same_tracks(GPUKLT& a, GPUKLT& b) {
    if (a.m_nMaxCorners != b.m_nMaxCorners || a.m_frame != b.m_frame) return false;
    for (int i = 0; i < a.m_nMaxCorners; ++i) {
        const Track2D &ta = a.m_tks[i], &tb = b.m_tks[i];
        if (ta.empty() != tb.empty()) return false;
        if (ta.empty()) continue;
        if (ta.f1 != tb.f1 || ta.f2 != tb.f2 || ta.length() != tb.length()) return false;
        const Track2DNode *pa = ta.head.next, *pb = tb.head.next;
        const FeaturePoint* prevA = 0;
        for (; pa && pb; pa = pa->next, pb = pb->next) {
            if (pa->f != pb->f || pa->x != pb->x || pa->y != pb->y) return false;
            if (!pa->pt || !pb->pt || pa->pt->x != pb->pt->x || pa->pt->y != pb->pt->y || pa->pt->f != pb->pt->f) return false;
            if ((pa->pt->preFrame == 0) != (pb->pt->preFrame == 0)) return false;           // the chain propagateFeatureStates walks
            if (pa->pt->preFrame && pa->pt->preFrame != prevA) return false;
            if ((pa->pt->nextFrame == 0) != (pb->pt->nextFrame == 0)) return false;
            prevA = pa->pt;
        }
        if (pa || pb) return false;
        if ((ta.tail == 0) != (tb.tail == 0) || ta.tail->f != tb.tail->f) return false;
    }
    return true;
}
static bool same_lists(const FeaturePoints& a, const FeaturePoints& b, int f0, int f1) {
    for (int f = f0; f <= f1; ++f) {
        if (a.totalFrameNum(f) != b.totalFrameNum(f)) return false;
        const FeaturePoint *pa = a.getFrameHead(f), *pb = b.getFrameHead(f);
        if ((pa == 0) != (pb == 0)) return false;
        if (!pa) continue;
        const FeaturePoint *ea = a.getFrameTail(f)->next, *eb = b.getFrameTail(f)->next;
        for (; pa != ea && pb != eb; pa = pa->next, pb = pb->next)
            if (pa->x != pb->x || pa->y != pb->y || pa->f != pb->f || pa->camId != pb->camId) return false;
        if ((pa == ea) != (pb == eb)) return false;
    }
    return true;
}

#include <chrono>
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// `bench`: what the two forms cost per 8-camera frame (640 x 480, CoSLAM's 32 x 32 slots and KLT configuration), host lists included
static int bench() {
    const int W = 640, H = 480, NC = 8, NF = 120, WARM = 20;
    std::vector<double> bx, by, amp;
    for (int k = 0; k < 1500; ++k) bx.push_back(urand() * W), by.push_back(urand() * H), amp.push_back((60 + 100 * urand()) * (urand() < 0.5 ? -1 : 1));
    const int NI = 24;   // distinct images, cycled forwards and backwards (a slow pan)
    std::vector<std::vector<unsigned char> > frames(NI, std::vector<unsigned char>((size_t)W * H));
    for (int f = 0; f < NI; ++f) render(W, H, bx, by, amp, 0.9 * f, -0.5 * f, frames[f].data());
    auto img = [&](int f, int c) { const int k = (f + 3 * c) % (2 * NI - 2); return frames[k < NI ? k : 2 * NI - 2 - k].data(); };
    V3D_GPU::KLT_SequenceTrackerConfig cfg;
    cfg.minDistance = 8, cfg.minCornerness = 1500.0f, cfg.nLevels = 6, cfg.windowWidth = 6, cfg.convergenceThreshold = 1.0f;
    cfg.SSD_Threshold = 20000.0f, cfg.trackWithGain = true;
    const double K[9] = {0.82 * W, 0, W / 2.0, 0, 0.82 * W, H / 2.0, 0, 0, 1};
    const double iK[9] = {1 / K[0], 0, -K[2] / K[0], 0, 1 / K[4], -K[5] / K[4], 0, 0, 1};
    const double kud[7] = {0, 0, 0, 0, 0, 0, 0};
    double usSync = 0, usEnq = 0, usLazy1 = 0, usLazy10 = 0, usLazyEnd = 0;
    size_t ptsSync = 0, ptsLazy = 0;
    {   // the reference's loop: camera after camera, lists built every frame
        std::vector<GPUKLT> k(NC);
        std::vector<FeaturePoints> ips(NC);
        for (int c = 0; c < NC; ++c) k[c].init(c, W, H, &cfg), k[c].setIntrinsicParam(K, iK, kud);
        double t0 = 0;
        for (int f = 0; f < WARM + NF; ++f) {
            if (f == WARM) t0 = now_us();
            for (int c = 0; c < NC; ++c) f == 0 ? k[c].first(0, img(f, c), ips[c]) : k[c].next(img(f, c), ips[c]);
        }
        usSync = (now_us() - t0) / NF;
        for (int c = 0; c < NC; ++c) ptsSync += ips[c].num;
    }
    for (int mode = 0; mode < 3; ++mode) {   // group path: sync() every frame / every 10th / only behind the last
        std::vector<GPUKLT> k(NC);
        std::vector<FeaturePoints> ips(NC);
        GPUKLT* kp[16];
        FeaturePoints* ip[16];
        for (int c = 0; c < NC; ++c) k[c].init(c, W, H, &cfg), k[c].setIntrinsicParam(K, iK, kud), kp[c] = &k[c], ip[c] = &ips[c];
        GPUKLTGroup rig(kp, ip, NC, 64);
        double t0 = 0, enq = 0;
        for (int f = 0; f < WARM + NF; ++f) {
            if (f == WARM) cs_klt_hostview_synchronize(rig.view()), rig.sync(), t0 = now_us(), enq = 0;
            const double e0 = now_us();
            for (int c = 0; c < NC; ++c) memcpy(rig.imageBuffer(c), img(f, c), (size_t)W * H);   // (a capture thread would decode straight into it)
            f == 0 ? rig.first(0, 0) : rig.next(0);
            enq += now_us() - e0;
            if (mode == 0 || (mode == 1 && f % 10 == 9) || (mode == 2 && rig.pendingFrames() >= 60)) rig.sync();
        }
        rig.sync();
        const double us = (now_us() - t0) / NF;
        (mode == 0 ? usLazy1 : mode == 1 ? usLazy10 : usLazyEnd) = us;
        if (mode == 2) usEnq = enq / NF;
        ptsLazy = 0;
        for (int c = 0; c < NC; ++c) ptsLazy += ips[c].num;
    }
    printf("{\"cameras\": %d, \"frames_timed\": %d, \"sync_per_camera_us_per_frame\": %.1f, \"group_sync_every_frame_us\": %.1f, "
           "\"group_sync_every_10th_us\": %.1f, \"group_sync_every_60th_us\": %.1f, \"group_enqueue_only_host_us\": %.1f, "
           "\"feature_points_sync\": %zu, \"feature_points_lazy\": %zu}\n",
           NC, NF, usSync, usLazy1, usLazy10, usLazyEnd, usEnq, ptsSync, ptsLazy);
    return ptsSync == ptsLazy ? 0 : 1;
}

int main(int argc, char** argv) {
    if (argc > 1 && !strcmp(argv[1], "bench")) return bench();
    const int W = 640, H = 480, NF = 14, NC = 3;
    std::vector<double> bx, by, amp;
    for (int k = 0; k < 1500; ++k) bx.push_back(urand() * W), by.push_back(urand() * H), amp.push_back((60 + 100 * urand()) * (urand() < 0.5 ? -1 : 1));
    std::vector<std::vector<unsigned char> > frames((size_t)NF * NC, std::vector<unsigned char>((size_t)W * H));
    for (int f = 0; f < NF; ++f)
        for (int c = 0; c < NC; ++c) render(W, H, bx, by, amp, (1.3 - 0.4 * c) * f + 7 * c, (-0.7 + 0.5 * c) * f, frames[(size_t)f * NC + c].data());

    V3D_GPU::KLT_SequenceTrackerConfig cfg;   // CoSLAM's own configuration (src/app/SL_SingleSLAM.cpp:291-298, src/app/SL_GlobParam.cpp:28-34)
    cfg.minDistance = 8, cfg.minCornerness = 1500.0f, cfg.nLevels = 6, cfg.windowWidth = 6, cfg.convergenceThreshold = 1.0f;
    cfg.SSD_Threshold = 20000.0f, cfg.trackWithGain = true;
    const double K[9] = {0.82 * W, 0, W / 2.0, 0, 0.82 * W, H / 2.0, 0, 0, 1};
    const double iK[9] = {1 / K[0], 0, -K[2] / K[0], 0, 1 / K[4], -K[5] / K[4], 0, 0, 1};
    const double kud[7] = {0.2, 0.02, 0, 0, 0, 0, 0};   // pushes border features outwards: the out >= W | H rule fires

    SingleSLAM* A = new SingleSLAM[NC];
    SingleSLAM* B = new SingleSLAM[NC];
    std::vector<std::vector<MapPoint> > mA(NC), mB(NC);
    GPUKLT* klts[NC];
    FeaturePoints* ips[NC];
    for (int c = 0; c < NC; ++c) {
        SingleSLAM* two[2] = {&A[c], &B[c]};
        for (int r = 0; r < 2; ++r) {
            SingleSLAM* s = two[r];
            s->camId = c, s->W = W, s->H = H;
            s->blkW = W / s->nColBlk, s->blkH = H / s->nRowBlk;   // SL_SingleSLAM.cpp:270-271
            s->m_tracker.init(c, W, H, &cfg);                      // :299-301
            s->m_tracker.setIntrinsicParam(K, iK, kud);
        }
        mA[c].resize(A[c].m_tracker.m_nMaxCorners), mB[c].resize(B[c].m_tracker.m_nMaxCorners);
        klts[c] = &B[c].m_tracker, ips[c] = &B[c].m_featPts;
    }
    GPUKLTGroup rig(klts, ips, NC, /*depth*/ 8);
    const int N = A[0].m_tracker.m_nMaxCorners;

    int lastChecked = -1, longest = 0, chosenTotal = 0, mappedTotal = 0, dropped = 0;
    for (int f = 0; f < NF; ++f) {
        const unsigned char* imgs[NC];
        for (int c = 0; c < NC; ++c) imgs[c] = frames[(size_t)f * NC + c].data();
        // rig A: the reference's loop, camera after camera (src/app/SL_CoSLAM.cpp:299-305)
        for (int c = 0; c < NC; ++c) {
            if (f == 0) A[c].m_tracker.first(0, imgs[c], A[c].m_featPts);
            else A[c].m_tracker.next(imgs[c], A[c].m_featPts);
            label(A[c], mA[c], f);
        }
        // rig B: one enqueue for all cameras; NO list is touched here
        if (f == 0) rig.first(0, imgs);
        else if (f % 2) rig.next(imgs);
        else {   // every other frame decoded straight into the view's pinned buffers (no host copy inside next())
            for (int c = 0; c < NC; ++c) memcpy(rig.imageBuffer(c), imgs[c], (size_t)W * H);
            rig.next(0);
        }
        CHECK(B[0].m_tracker.currentFrame() == f);
        if (f != 4 && f != 9 && f != NF - 1) continue;
        CHECK(rig.pendingFrames() == f - lastChecked);
        const int added = rig.sync();
        CHECK(added > 0 && rig.pendingFrames() == 0);
        // the labelling the frames in between would have received, on the lazily built lists: label(.., g) touches the tails that date from
        // frame g -- the newest frame's features, and the tails of tracks whose feature the out >= W | H rule has dropped since (the track
        // keeps its old tail, GPUKLT.cpp:47-48); labels of features that are no tail any more are not read by the consumers below
        for (int c = 0; c < NC; ++c) {
            for (int g = lastChecked + 1; g <= f; ++g) label(B[c], mB[c], g);
            CHECK(same_tracks(A[c].m_tracker, B[c].m_tracker));
            CHECK(same_lists(A[c].m_featPts, B[c].m_featPts, 0, f));
            std::vector<FeaturePoint*> va, vb;
            const int ka = A[c].chooseStaticFeatPts(va), kb = B[c].chooseStaticFeatPts(vb);   // the reference's own code on both
            CHECK(ka == kb && ka > 20);
            for (int q = 0; q < ka; ++q) CHECK(va[q]->x == vb[q]->x && va[q]->y == vb[q]->y && va[q]->f == vb[q]->f && (va[q]->mpt == 0) == (vb[q]->mpt == 0));
            const int na = A[c].getNumMappedStaticPts(), nb = B[c].getNumMappedStaticPts();
            CHECK(na == nb);
            chosenTotal += ka, mappedTotal += na;
            for (int i = 0; i < N; ++i)
                if (!B[c].m_tracker.m_tks[i].empty() && B[c].m_tracker.m_tks[i].length() > longest) longest = B[c].m_tracker.m_tks[i].length();
        }
        lastChecked = f;
        printf("frame %d: lists rebuilt from %d feature points; tracks, frame lists, chooseStaticFeatPts, getNumMappedStaticPts equal on %d cameras\n", f,
               added, NC);
    }
    {   // the drop rule was exercised: some frame list is shorter than the number of live slots
        const int* st;
        const double* xy;
        CHECK(cs_klt_hostview_fetch(rig.view(), NF - 1, &st, &xy) == CS_OK);
        for (int i = 0; i < NC * N; ++i) dropped += st[i] == -2;
        CHECK(cs_klt_hostview_fetch(rig.view(), 2, &st, &xy) != CS_OK);   // (depth 8: frame 2 has been overwritten -- and says so)
    }
    CHECK(longest == NF && mappedTotal > 0 && dropped > 0);
    printf("lazy adaptor ok: %d cameras x %d frames, longest track %d, %d chosen / %d mapped static features compared, %d drops by the >= W|H rule\n", NC,
           NF, longest, chosenTotal, mappedTotal, dropped);
    return 0;
}
