// frame_loop.cpp -- the headline frame loop of bench.py driven from C++ through the C-ABI only (include/coslam_hip.h): no
// Python, no torch.  north_star: "Host stays C++".  Same workload, same streams, same key-frame cadence, same drain as
// bench.py's timed loop; the workload (synthetic frames, map, BA problems, pose graphs) is read from the file bench.py writes
// with --export-workload (tools: bench.py export_workload()).
//   hipcc -O2 -std=c++17 -Iinclude tools/cxx/frame_loop.cpp -Lcoslam_amd/lib -lcoslam_hip -Wl,-rpath,$PWD/coslam_amd/lib \
//         -o tools/cxx/frame_loop.bin
//   tools/cxx/frame_loop.bin <workload file> <steps> <warmup> [cams per tracker launch]
// Per frame (reference call sites in bench.py's docstring): camera-group redetect (+ prefetch of the next frame's front) on the
// tracker stream; hand-back + intraCamEstimate of all cameras + both registration passes on the pose stream, event-ordered
// behind the tracker; at key frames the inter-camera solve and the joint local BA (parsed on the device from the window ring) on their
// workspaces' worker threads; `lag` key-frame intervals behind its key frame every joint BA's packed result is written back into
// the LIVE map, the pose history and the window (cs_ba_output_apply_dev = RobustBundleRTS::output(): key poses, points, outlier
// points false, relaxation of the non-key frames, updateNewPosesPoints) -- the pose stream waits for the record on the device.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <algorithm>
#include <string>
#include <vector>

#include "coslam_hip.h"

#define HIPCHK(x)                                                                              \
    do {                                                                                       \
        hipError_t e_ = (x);                                                                   \
        if (e_ != hipSuccess) {                                                                \
            fprintf(stderr, "%s failed: %s (%s:%d)\n", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(2);                                                                           \
        }                                                                                      \
    } while (0)
#define CSCHK(x)                                                                     \
    do {                                                                             \
        int rc_ = (x);                                                               \
        if (rc_ != CS_OK) {                                                          \
            fprintf(stderr, "%s failed (%d): %s (%s:%d)\n", #x, rc_, cs_last_error(), __FILE__, __LINE__); \
            exit(3);                                                                 \
        }                                                                            \
    } while (0)

struct Reader {
    FILE* f;
    template <class T>
    std::vector<T> vec(size_t n) {
        std::vector<T> v(n);
        if (n && fread(v.data(), sizeof(T), n, f) != n) {
            fprintf(stderr, "workload file truncated\n");
            exit(4);
        }
        return v;
    }
    int i32() { return vec<int>(1)[0]; }
    double f64() { return vec<double>(1)[0]; }
};

template <class T>
static T* to_dev(const std::vector<T>& h) {
    T* d = nullptr;
    HIPCHK(hipMalloc((void**)&d, sizeof(T) * (h.size() ? h.size() : 1)));
    if (!h.empty()) HIPCHK(hipMemcpy(d, h.data(), sizeof(T) * h.size(), hipMemcpyHostToDevice));
    return d;
}
template <class T>
static T* dev_zeros(size_t n) {
    T* d = nullptr;
    HIPCHK(hipMalloc((void**)&d, sizeof(T) * (n ? n : 1)));
    HIPCHK(hipMemset(d, 0, sizeof(T) * (n ? n : 1)));
    return d;
}

struct BaProblem {
    int C, P, nObs, nCamsCon, nPtsCon, maxIter, inner;
    double maxErr;
    std::vector<double> Ks, Rs, Ts, pts, xy;
    std::vector<int> ptr, cam;
    cs_ba* ws = nullptr;
    double *dR = nullptr, *dT = nullptr, *dM = nullptr;
    void read(Reader& r) {
        C = r.i32(), P = r.i32(), nObs = r.i32(), nCamsCon = r.i32(), nPtsCon = r.i32(), maxIter = r.i32(), inner = r.i32();
        maxErr = r.f64();
        Ks = r.vec<double>(9 * (size_t)C), Rs = r.vec<double>(9 * (size_t)C), Ts = r.vec<double>(3 * (size_t)C);
        pts = r.vec<double>(3 * (size_t)P);
        ptr = r.vec<int>((size_t)P + 1), cam = r.vec<int>((size_t)nObs), xy = r.vec<double>(2 * (size_t)nObs);
    }
    void upload(int dev) {
        ws = cs_ba_create(dev);
        if (!ws) {
            fprintf(stderr, "cs_ba_create: %s\n", cs_last_error());
            exit(3);
        }
        CSCHK(cs_ba_upload(ws, C, P, nObs, Ks.data(), Rs.data(), Ts.data(), pts.data(), ptr.data(), cam.data(), xy.data()));
        dR = to_dev(Rs), dT = to_dev(Ts), dM = to_dev(pts);
    }
    void solve_async(hipStream_t after) {
        CSCHK(cs_ba_solve_async(ws, (void*)after, C, P, nObs, dR, dT, dM, nCamsCon, nPtsCon, maxErr, maxIter, inner));
    }
};

int main(int argc, char** argv) {
    if (argc < 4) {
        fprintf(stderr, "usage: %s <workload file> <steps> <warmup> [cams per tracker launch] [BA apply lag in key-frame intervals: 2] [first timed frame]\n", argv[0]);
        return 1;
    }
    const int steps = atoi(argv[2]), warmup = atoi(argv[3]);
    const int camsPerLaunchArg = argc > 4 ? atoi(argv[4]) : -1;
    const int baLag = argc > 5 && atoi(argv[5]) > 0 ? atoi(argv[5]) : 2;
    // the frame at which the timed region starts (bench.py hands over its own: both loops then time the SAME stretch of the sequence -- the
    // first hundreds of frames, while the map settles, are heavier than the steady state); 0: right behind the set-up
    const int timedFrom = argc > 6 ? atoi(argv[6]) : 0;
    const bool useWindow = true;
    Reader rd{fopen(argv[1], "rb")};
    if (!rd.f) {
        perror(argv[1]);
        return 1;
    }
    char magic[8];
    if (fread(magic, 1, 8, rd.f) != 8 || memcmp(magic, "CSWL1\0\0\0", 8) != 0) {
        fprintf(stderr, "%s is not a workload file\n", argv[1]);
        return 1;
    }
    const std::vector<int> hd = rd.vec<int>(16);
    const int nCams = hd[0], W = hd[1], H = hd[2], L = hd[3], FW = hd[4], FH = hd[5], nFrames = hd[6], orderLen = hd[7],
              nPts = hd[8], P_REG = hd[9], PTS = hd[10], nColBlk = hd[11], nRowBlk = hd[12], keyEvery = hd[13];
    const int camsPerLaunch = camsPerLaunchArg >= 0 ? camsPerLaunchArg : hd[14];
    // One process per GPU (RANK / WORLD_SIZE / LOCAL_RANK as torch.distributed.run sets them; none: one rank).  Rank r owns cameras
    // r * nc .. r * nc + nc - 1: their images, trackers, hand-backs and pose solves; everything behind the per-frame all-gather of
    // {dest[], R, t} is replayed on every rank's own replica of the map.  COSLAM_FORCE_DEVICE: ranks sharing one GPU (tests).
    auto envi = [](const char* k, int dflt) { const char* e = getenv(k); return e && e[0] ? atoi(e) : dflt; };
    const int world = envi("WORLD_SIZE", 1), rank = envi("RANK", 0);
    const int N = FW * FH, dev = envi("COSLAM_FORCE_DEVICE", envi("LOCAL_RANK", 0));
    if (world < 1 || rank < 0 || rank >= world || nCams % world) {
        fprintf(stderr, "%d cameras do not shard over %d ranks (rank %d)\n", nCams, world, rank);
        return 1;
    }
    const int nc = nCams / world, c0 = rank * nc;
    const std::vector<int> order = rd.vec<int>(orderLen);
    const std::vector<double> K = rd.vec<double>(9);
    cs_klt_config cfg;
    {
        const std::vector<int> ci = rd.vec<int>(6);      // nIterations, nLevels, levelSkip, windowWidth, trackWithGain, minDistance
        const std::vector<float> cf = rd.vec<float>(5);  // trackBorderMargin, convergenceThreshold, SSD_Threshold, minCornerness, detectBorderMargin
        cfg.nIterations = ci[0], cfg.nLevels = ci[1], cfg.levelSkip = ci[2], cfg.windowWidth = ci[3], cfg.trackWithGain = ci[4],
        cfg.minDistance = ci[5];
        cfg.trackBorderMargin = cf[0], cfg.convergenceThreshold = cf[1], cfg.SSD_Threshold = cf[2], cfg.minCornerness = cf[3],
        cfg.detectBorderMargin = cf[4];
    }
    HIPCHK(hipSetDevice(dev));
    // frames: resident in HBM before the clock starts, like bench.py's headline
    std::vector<uint8_t*> dFrames(nCams);
    const size_t imgBytes = (size_t)W * H;
    for (int c = 0; c < nCams; ++c) dFrames[c] = to_dev(rd.vec<uint8_t>(imgBytes * nFrames));
    // the map keeps spare capacity behind the scene's points: NewMapPtsNCC's new points are appended (cs_newpts_from_pairs_dev)
    const int MAP_SPARE = 8192, nMap = nPts + MAP_SPARE;
    const std::vector<double> mapPts = rd.vec<double>(3 * (size_t)nPts);
    // projections of the visible map points in the first frame (for the slot -> map point association that stands in for the
    // map initialisation, as in bench.py associate())
    std::vector<std::vector<int>> visIdx(nCams);
    std::vector<std::vector<double>> visUV(nCams);
    for (int c = 0; c < nCams; ++c) {
        const int nv = rd.i32();
        visIdx[c] = rd.vec<int>(nv);
        visUV[c] = rd.vec<double>(2 * (size_t)nv);
    }
    const std::vector<double> R0 = rd.vec<double>(9 * (size_t)nCams), t0 = rd.vec<double>(3 * (size_t)nCams);
    const std::vector<double> cov = rd.vec<double>(9 * (size_t)std::max(nPts, 2 * P_REG));  // MapPoint::cov of every map point
    BaProblem joint, ic;
    joint.read(rd);
    ic.read(rd);
    const int pgGraphs = rd.i32(), pgNodesPer = rd.i32();
    const int pgNodes = pgGraphs * pgNodesPer, pgEdges = pgGraphs * (pgNodesPer - 1);
    const std::vector<uint8_t> pgFixed = rd.vec<uint8_t>(pgNodes);
    const std::vector<double> pgR = rd.vec<double>(9 * (size_t)pgNodes), pgT = rd.vec<double>(3 * (size_t)pgNodes);
    const std::vector<int> pgCam = rd.vec<int>(joint.C);
    (void)rd.vec<double>((size_t)nFrames * (nCams - 1) * 9);  // (the file's ground-truth fundamental matrices [frame][pair][9]: not used --
                                                              // the matching leg forms F from the poses it has solved, cs_ncc_fmats_dev)
    fclose(rd.f);

    // ---- trackers, group, streams ----
    hipStream_t kltS, poseS;
    HIPCHK(hipStreamCreateWithFlags(&kltS, hipStreamNonBlocking));
    HIPCHK(hipStreamCreateWithFlags(&poseS, hipStreamNonBlocking));
    std::vector<cs_klt*> trk(nc);   // the rank's own cameras: c0 + k
    for (int c = 0; c < nc; ++c) {
        trk[c] = cs_klt_create(&cfg, dev, 0);
        if (!trk[c]) {
            fprintf(stderr, "cs_klt_create: %s\n", cs_last_error());
            return 3;
        }
        CSCHK(cs_klt_allocate(trk[c], W, H, L, FW, FH, 0, 0));
    }
    cs_klt_group* grp = cs_klt_group_create(trk.data(), nc);
    if (!grp) {
        fprintf(stderr, "cs_klt_group_create: %s\n", cs_last_error());
        return 3;
    }
    CSCHK(cs_klt_group_set_stream(grp, (void*)kltS));
    {   // a camera's tracker workgroups on that camera's own XCD (COSLAM_KLT_XCD=0: cameras as grid rows, for the A/B)
        const char* e = getenv("COSLAM_KLT_XCD");
        for (cs_klt* k : trk) CSCHK(cs_klt_set_xcd_placement(k, !(e && e[0] == '0')));
    }
    if (camsPerLaunch > 0)  // the co-residency budget of `camsPerLaunch` cameras (250 waves each, 8 resident waves per CU)
        for (cs_klt* k : trk) CSCHK(cs_klt_set_cu_count(k, std::min(256, (250 * camsPerLaunch + 60) / 8 + 5)));

    double* dK = to_dev(K);
    std::vector<double> Kall;
    for (int c = 0; c < nCams; ++c) Kall.insert(Kall.end(), K.begin(), K.end());
    double* dKall = to_dev(Kall);
    double* dKud = dev_zeros<double>(7);
    double* dMap = dev_zeros<double>(3 * (size_t)nMap);
    double* dCov = dev_zeros<double>(9 * (size_t)std::max(nMap, 2 * P_REG));
    HIPCHK(hipMemcpy(dMap, mapPts.data(), sizeof(double) * mapPts.size(), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dCov, cov.data(), sizeof(double) * cov.size(), hipMemcpyHostToDevice));
    int* dMapCount = dev_zeros<int>(1);
    HIPCHK(hipMemcpy(dMapCount, &nPts, sizeof(int), hipMemcpyHostToDevice));
    int* dS2M = dev_zeros<int>((size_t)nCams * N);
    int* dSpan = dev_zeros<int>((size_t)nCams * 2 * N);
    HIPCHK(hipMemset(dS2M, 0xff, sizeof(int) * (size_t)nCams * N));
    HIPCHK(hipMemset(dSpan, 0xff, sizeof(int) * (size_t)nCams * 2 * N));
    double* dXY = dev_zeros<double>((size_t)nCams * 2 * N);
    int* dState = dev_zeros<int>((size_t)nCams * N);
    double* dMs = dev_zeros<double>((size_t)nCams * PTS * 3);
    double* dms = dev_zeros<double>((size_t)nCams * PTS * 2);
    int* dSel = dev_zeros<int>((size_t)nCams * PTS);
    int* dNpts = dev_zeros<int>(nCams);
    cs_pose_option* dOpt = (cs_pose_option*)dev_zeros<unsigned char>((size_t)nCams * sizeof(cs_pose_option));
    int* dOk = dev_zeros<int>(nCams);
    int* dPf = dev_zeros<int>((size_t)nMap * nCams);  // MapPoint::pFeatures of this frame, nMap x nCams (the hand-back writes it)
    HIPCHK(hipMemset(dPf, 0xff, sizeof(int) * (size_t)nMap * nCams));
    // poseUpdate3D's second half + detectDynamicFeaturePoints behind the pose solve (cs_pose_update_frame_dev)
    const std::vector<double> iKh = {1 / K[0], -K[1] / (K[0] * K[4]), (K[1] * K[5] - K[2] * K[4]) / (K[0] * K[4]), 0, 1 / K[4], -K[5] / K[4], 0, 0, 1};
    double* diK = to_dev(iKh);
    double* dFm = dev_zeros<double>((size_t)16 * 9);   // the camera pairs' fundamental matrices of a matching run
    unsigned char* dIsStatic = dev_zeros<unsigned char>((size_t)nCams * N);
    HIPCHK(hipMemset(dIsStatic, 1, (size_t)nCams * N));
    double* dReproj = dev_zeros<double>((size_t)nCams * N);
    unsigned char* dMapFlags = dev_zeros<unsigned char>(nMap);
    unsigned char* dMergeable = dev_zeros<unsigned char>((size_t)nMap * nCams);   // the registration's tables are indexed by the MAP index
    // CoSLAM::mapPointsClassify behind the pose update: MapPoint::bNewPt / staticFrameNum / firstFrame of every map point
    unsigned char* dNewPt = dev_zeros<unsigned char>(nMap);
    int* dSfn = dev_zeros<int>(nMap);
    int* dFirstFrm = dev_zeros<int>(nMap);
    // walks 64 frames deep; 4096 frames of pixels + poses kept behind them for the running whole-track mergability verdict
    cs_track_history* hist = cs_track_history_create_ex(dev, nCams, N, 64, 4096);
    if (!hist) {
        fprintf(stderr, "cs_track_history_create_ex: %s\n", cs_last_error());
        return 3;
    }
    // ---- N > 1: the communicator (RCCL through the library's own cs_comm_*; COSLAM_COMM=host:<segment>: the test transport for ranks
    // sharing one GPU), the per-frame exchange of {dest[], R, t}, the candidates' and the NCC records' all-gathers, the BA result's broadcast
    cs_comm* comm = nullptr;
    cs_exchange* xchg = nullptr;
    unsigned char* xRecv = nullptr;
    size_t xRecBytes = 0;
    int* dCandSend = nullptr;
    int* dCandRecv = nullptr;
    unsigned char* dRecvRec[2] = {nullptr, nullptr};
    if (world > 1) {
        const char* how = getenv("COSLAM_COMM");
        if (how && !strncmp(how, "host:", 5)) {
            comm = cs_comm_create_host(how + 5, world, rank, dev);
        } else {
            // rank 0 creates the unique id and leaves it in COSLAM_COMM_ID_FILE (written to a temporary name, then renamed); the others poll
            const char* path = getenv("COSLAM_COMM_ID_FILE");
            unsigned char id[128];
            if (!path || !cs_comm_available()) {
                fprintf(stderr, "N > 1 needs COSLAM_COMM_ID_FILE (or COSLAM_COMM=host:<name>) and RCCL: %s\n", cs_last_error());
                return 3;
            }
            if (rank == 0) {
                CSCHK(cs_comm_unique_id(id));
                const std::string tmp = std::string(path) + ".tmp";
                FILE* f = fopen(tmp.c_str(), "wb");
                if (!f || fwrite(id, 1, 128, f) != 128 || fclose(f) != 0 || rename(tmp.c_str(), path) != 0) {
                    perror(path);
                    return 3;
                }
            } else {
                bool got = false;
                for (int tries = 0; tries < 12000 && !got; ++tries) {   // up to 60 s
                    FILE* f = fopen(path, "rb");
                    if (f) {
                        got = fread(id, 1, 128, f) == 128;
                        fclose(f);
                    }
                    if (!got) std::this_thread::sleep_for(std::chrono::milliseconds(5));
                }
                if (!got) {
                    fprintf(stderr, "rank %d: no unique id in %s after 60 s\n", rank, path);
                    return 3;
                }
            }
            comm = cs_comm_create(id, world, rank, dev);
        }
        if (!comm || !(xchg = cs_exchange_create(comm, nc, N))) {
            fprintf(stderr, "rank %d: communicator: %s\n", rank, cs_last_error());
            return 3;
        }
        void* rv = nullptr;
        CSCHK(cs_exchange_buffers(xchg, &rv, &xRecBytes));
        xRecv = (unsigned char*)rv;
        dCandSend = dev_zeros<int>((size_t)3 * nc * P_REG);
        dCandRecv = dev_zeros<int>((size_t)3 * nc * P_REG * world);
    }
    void* dMergeCache = dev_zeros<unsigned char>(cs_register_mergability_cache_bytes(nMap, nCams));
    // MapPoint::pFeatures as feature references (stale features are views, re-linked chains: SL_CoSLAM.cpp:775-779); COSLAM_FEATURE_CHAINS=0:
    // this frame's features on their own tracks
    if (getenv("COSLAM_MERGE_PRINT")) cs_debug_set("merge_print", 1);   // (k_decide_merge prints its own account per call)
    const bool chains = !(getenv("COSLAM_FEATURE_CHAINS") && getenv("COSLAM_FEATURE_CHAINS")[0] == '0');
    cs_feat_ref* dFref = nullptr;
    unsigned char* dRstat = nullptr;
    int* dFrefCnt = dev_zeros<int>(5);
    if (chains) {
        HIPCHK(hipMalloc((void**)&dFref, sizeof(cs_feat_ref) * (size_t)nMap * nCams));
        HIPCHK(hipMemset(dFref, 0xff, sizeof(cs_feat_ref) * (size_t)nMap * nCams));   // (-1 everywhere: no feature)
        dRstat = dev_zeros<unsigned char>((size_t)nMap * nCams);
    }
    int* dCurList = dev_zeros<int>(nMap);
    int* dCurCount = dev_zeros<int>(1);
    int* dCurOverflow = dev_zeros<int>(1);
    // the second visits' rounds (cs_register_revisit_*)
    const int RV_CAP = 1024, RV_ROUNDS = getenv("COSLAM_REVISIT_ROUNDS") ? atoi(getenv("COSLAM_REVISIT_ROUNDS")) : 2;
    int* dRvList = dev_zeros<int>(RV_CAP);
    int *dRvVisit = dev_zeros<int>(nMap), *dRvNext = dev_zeros<int>(nMap), *dRvCnt = dev_zeros<int>(4), *dRvListCnt = dev_zeros<int>(4);
    const bool fusedRounds = chains && !(getenv("COSLAM_FUSED_ROUNDS") && getenv("COSLAM_FUSED_ROUNDS")[0] == '0');   // the lists by the walks, advance + refine as one launch
    int *dRvLists = dev_zeros<int>((size_t)(RV_ROUNDS > 0 ? RV_ROUNDS : 1) * RV_CAP), *dRvCounts = dev_zeros<int>(RV_ROUNDS + 1);
    unsigned char* dRvReg[2] = {dev_zeros<unsigned char>(nMap), dev_zeros<unsigned char>(nMap)};   // current points beyond the list's cap P_REG (left out of that frame's registration)
    int* dMergeRun = dev_zeros<int>(4);
    double* dR[2] = {to_dev(R0), to_dev(R0)};
    double* dT[2] = {to_dev(t0), to_dev(t0)};
    cs_klt_feature* dDest[2][16];
    int* dCnt[16];
    for (int c = 0; c < nCams; ++c) {
        dDest[0][c] = dev_zeros<cs_klt_feature>(N);
        dDest[1][c] = dev_zeros<cs_klt_feature>(N);
        dCnt[c] = dev_zeros<int>(4);
    }
    struct RegOut {
        int *slot, *flags;
        double *m, *var, *dist;
    } reg[1];
    for (RegOut& o : reg) {
        o.slot = dev_zeros<int>((size_t)nMap * nCams), o.flags = dev_zeros<int>((size_t)nMap * nCams);
        o.m = dev_zeros<double>((size_t)nMap * nCams * 2), o.var = dev_zeros<double>((size_t)nMap * nCams * 4);
        o.dist = dev_zeros<double>((size_t)nMap * nCams);
        HIPCHK(hipMemset(o.slot, 0xff, sizeof(int) * (size_t)nMap * nCams));
    }
    auto hb_cams = [&](int b) {
        std::vector<cs_handback_cam> v(nCams);
        for (int c = 0; c < nCams; ++c) {
            cs_handback_cam& h = v[c];
            memset(&h, 0, sizeof(h));
            h.dest = dDest[b][c], h.K = dK, h.kud = dKud, h.mapPts = dMap, h.slot2map = dS2M + (size_t)c * N;
            h.trackSpan = dSpan + (size_t)c * 2 * N, h.xy = dXY + (size_t)c * 2 * N, h.state = dState + (size_t)c * N;
            h.Ms = dMs + (size_t)c * PTS * 3, h.ms = dms + (size_t)c * PTS * 2, h.sel = dSel + (size_t)c * PTS;
            h.npts = dNpts + c, h.opt = dOpt + c, h.pointFeat = dPf + c, h.pointFeatStride = nCams, h.nPointFeat = nMap;
            h.isStatic = dIsStatic + (size_t)c * N;
        }
        return v;
    };
    const std::vector<cs_handback_cam> hb[2] = {hb_cams(0), hb_cams(1)};   // all cameras (the window's push reads xy / state / slot2map)
    std::vector<cs_handback_cam> hbOwn[2], hbOther;   // the rank's own cameras from their trackers' dest[]; the others from the gathered records
    for (int b = 0; b < 2; ++b) hbOwn[b].assign(hb[b].begin() + c0, hb[b].begin() + c0 + nc);
    for (int c = 0; c < nCams && world > 1; ++c)
        if (c < c0 || c >= c0 + nc) {
            cs_handback_cam h = hb[0][c];
            h.dest = (const cs_klt_feature*)(xRecv + (size_t)c * xRecBytes);
            hbOther.push_back(h);
        }
    auto reg_cams = [&](int dst) {
        std::vector<cs_register_cam> v(nCams);
        for (int c = 0; c < nCams; ++c) {
            memset(&v[c], 0, sizeof(v[c]));
            v[c].K = dK, v[c].R = dR[dst] + 9 * c, v[c].t = dT[dst] + 3 * c, v[c].xy = dXY + (size_t)c * 2 * N;
            v[c].state = dState + (size_t)c * N, v[c].slot2map = dS2M + (size_t)c * N;
            v[c].isStatic = dIsStatic + (size_t)c * N;   // (FeaturePoint::type as the pose update keeps it: a static point's walk passes DYNAMIC features by)
        }
        return v;
    };
    const std::vector<cs_register_cam> rc[2] = {reg_cams(0), reg_cams(1)};
    std::vector<cs_poseupdate_cam> pu(nCams);
    for (int c = 0; c < nCams; ++c) {
        memset(&pu[c], 0, sizeof(pu[c]));
        pu[c].K = dK, pu[c].iK = diK, pu[c].xy = dXY + (size_t)c * 2 * N, pu[c].state = dState + (size_t)c * N;
        pu[c].slot2map = dS2M + (size_t)c * N, pu[c].trackSpan = dSpan + (size_t)c * 2 * N;
        pu[c].reprojErr = dReproj + (size_t)c * N, pu[c].isStatic = dIsStatic + (size_t)c * N;
    }

    // ---- key-frame solves: workspaces, pose graphs as the joint BA's follow-up ----
    // joint local BA: parsed on the device from the ring of the last 5 key frames (cs_ba_window_*: the hand-back's records and
    // the poses of every camera at the key frame), like bench.py's N = 1 default; `0` as the 5th argument keeps the pre-baked one
    const int WIN_KF = 5;
    cs_ba_window* win = nullptr;
    if (useWindow) {
        joint.ws = cs_ba_create(dev);
        win = cs_ba_window_create(dev, nCams, WIN_KF, N, nMap);
        if (!joint.ws || !win) {
            fprintf(stderr, "cs_ba_window_create: %s\n", cs_last_error());
            return 3;
        }
        CSCHK(cs_ba_reserve_for_window(joint.ws, win));  // (the result buffers' addresses are final from here on)
    } else {
        joint.upload(dev);
    }
    // the inter-camera problem is built on the device from every key frame's records (InterCamPoseEstimator::addMapPoints); the
    // file's pre-baked one is only read past
    ic.ws = cs_ba_create(dev);
    cs_ba_intercam* icam = cs_ba_intercam_create(dev, nCams, N, PTS, nMap, 60);
    if (!ic.ws || !icam) {
        fprintf(stderr, "cs_ba_intercam_create: %s\n", cs_last_error());
        return 3;
    }
    std::vector<cs_intercam_cam> icCams(nCams);
    for (int c = 0; c < nCams; ++c) {
        icCams[c].K = dK, icCams[c].xy = dXY + (size_t)c * 2 * N, icCams[c].state = dState + (size_t)c * N;
        icCams[c].slot2map = dS2M + (size_t)c * N, icCams[c].trackSpan = dSpan + (size_t)c * 2 * N, icCams[c].isStatic = dIsStatic + (size_t)c * N;
    }
    // RobustBundleRTS::output(): every window solve's result packed by the worker, applied `baLag` key-frame intervals later
    // COSLAM_KEYFRAME_DRIVES=1: the key frames where CoSLAM::genNewMapPoints' decision puts them (src/app/SL_CoSLAM.cpp:1294-1346: one camera's
    // mapped points decreased -> addKeyFrame for all cameras -> requestForBA) instead of the fixed cadence -- and no host wait per frame: the
    // decision word of frame i goes into pinned memory behind an event, the host acts on the decision of frame i - COSLAM_KEYFRAME_LAG (>= 1),
    // the key frame's own records and poses come out of a ring of LAG + 1 snapshots (coslam_amd/frameloop.py: LoopConfig.keyframe_lag)
    const bool kfDrives = getenv("COSLAM_KEYFRAME_DRIVES") && getenv("COSLAM_KEYFRAME_DRIVES")[0] == '1';
    const int kfLag = std::max(1, envi("COSLAM_KEYFRAME_LAG", 1));
    const double kfRatio = getenv("COSLAM_KEYFRAME_RATIO") ? atof(getenv("COSLAM_KEYFRAME_RATIO")) : 0.93;   // m_mappedPtsReduceRatio
    cs_ba_output* bout = cs_ba_output_create(dev, nCams, WIN_KF, nMap, kfDrives ? std::min(baLag * keyEvery + 6, 64) : 8);
    if (!bout) {
        fprintf(stderr, "cs_ba_output_create: %s\n", cs_last_error());
        return 3;
    }
    CSCHK(cs_ba_output_attach(bout, joint.ws));
    if (chains) CSCHK(cs_ba_output_set_feat_refs(bout, dFref, dRstat));
    if (chains && !getenv("COSLAM_CLASSIFY_PLAIN")) CSCHK(cs_track_history_set_classify_refs(hist, (cs_feat_ref*)dFref, dRstat));   // mapPointsClassify over the references
    if (chains && !getenv("COSLAM_MERGE_PLAIN")) CSCHK(cs_track_history_set_merge_refs(hist, (cs_feat_ref*)dFref, dRstat));         // ... and the bMerge walks
    (void)pgFixed, (void)pgR, (void)pgT, (void)pgCam, (void)pgEdges;   // (the file's pre-baked camera graphs: the graphs are built live now)
    struct Due {
        int frame, firstKey;
        long long seq;   // the record's sequence number ON ITS OWNER (windows go round the ranks: the owner's (k / world)-th solve)
        int k, owner;    // window number, the rank that solves it (k % world)
        std::vector<int> frames;   // the window's key frames where the decision put them (empty: firstKey + j * keyEvery)
    };
    std::vector<Due> due;   // applies still to come, in frame order
    int nPushed = 0, nApplied = 0, nKey = 0;
    // ---- the key-frame decision's state (cs_keyframe_ready_dev) as CoSLAM::initMap leaves it: a key pose with self motion in every camera at
    // frame 0, nMappedPts 0, m_minCamTranslation = the mean distance between the cameras / 4.5 (src/app/SL_CoSLAM.cpp:246-256, :278-291)
    int *dKfFrame = dev_zeros<int>(nCams), *dKfMapped = dev_zeros<int>(nCams), *dKfReady = dev_zeros<int>(nCams + 2), *dKfCnt = dev_zeros<int>(2 * nCams);
    int* dKfStats = dev_zeros<int>(5);
    double *dKfSelfR = dev_zeros<double>(9 * (size_t)nCams), *dKfSelfT = dev_zeros<double>(3 * (size_t)nCams), *dKfCen = dev_zeros<double>(3 * (size_t)nCams);
    double kfMinTranslation = 0.1;
    std::vector<cs_keyframe_cam> kfCams[2];
    struct KfSnap {
        double *xy, *R, *t;
        int *st, *s2m, *word;   // word: pinned host memory
        hipEvent_t ev;
        int frame;
        std::vector<cs_handback_cam> hb;
    };
    std::vector<KfSnap> kfRing;
    std::vector<int> kfPlaced, kfPushedFrames;
    int kfNotApplied = 0;
    if (kfDrives) {
        HIPCHK(hipMemcpy(dKfSelfR, dR[0], sizeof(double) * 9 * nCams, hipMemcpyDeviceToDevice));
        HIPCHK(hipMemcpy(dKfSelfT, dT[0], sizeof(double) * 3 * nCams, hipMemcpyDeviceToDevice));
        std::vector<double> hR(9 * (size_t)nCams), hT(3 * (size_t)nCams), cen(3 * (size_t)nCams);
        HIPCHK(hipMemcpy(hR.data(), dR[0], sizeof(double) * hR.size(), hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(hT.data(), dT[0], sizeof(double) * hT.size(), hipMemcpyDeviceToHost));
        for (int c = 0; c < nCams; ++c)
            for (int k = 0; k < 3; ++k) cen[3 * c + k] = -(hR[9 * c + k] * hT[3 * c] + hR[9 * c + 3 + k] * hT[3 * c + 1] + hR[9 * c + 6 + k] * hT[3 * c + 2]);
        double sum = 0;
        int n = 0;
        for (int a = 0; a < nCams; ++a)
            for (int c = a + 1; c < nCams; ++c, ++n) {
                const double d0 = cen[3 * a] - cen[3 * c], d1 = cen[3 * a + 1] - cen[3 * c + 1], d2 = cen[3 * a + 2] - cen[3 * c + 2];
                sum += sqrt(d0 * d0 + d1 * d1 + d2 * d2);
            }
        if (n > 0) kfMinTranslation = sum / n / 4.5;
        for (int q = 0; q < 2; ++q) {
            kfCams[q].resize(nCams);
            for (int c = 0; c < nCams; ++c) {
                cs_keyframe_cam& k = kfCams[q][c];
                k.state = dState + (size_t)c * N, k.slot2map = dS2M + (size_t)c * N, k.R = dR[q] + 9 * c, k.t = dT[q] + 3 * c;
                k.keyFrame = dKfFrame + c, k.keyMapped = dKfMapped + c, k.selfR = dKfSelfR + 9 * c, k.selfT = dKfSelfT + 3 * c;
            }
        }
        kfRing.resize(kfLag + 1);
        for (auto& sn : kfRing) {
            sn.xy = dev_zeros<double>((size_t)nCams * 2 * N), sn.R = dev_zeros<double>(9 * (size_t)nCams), sn.t = dev_zeros<double>(3 * (size_t)nCams);
            sn.st = dev_zeros<int>((size_t)nCams * N), sn.s2m = dev_zeros<int>((size_t)nCams * N);
            HIPCHK(hipHostMalloc((void**)&sn.word, sizeof(int), hipHostMallocDefault));
            *sn.word = 0;
            HIPCHK(hipEventCreateWithFlags(&sn.ev, hipEventDisableTiming));
            sn.frame = -1;
            sn.hb = hb[0];
            for (int c = 0; c < nCams; ++c)
                sn.hb[c].xy = sn.xy + (size_t)c * 2 * N, sn.hb[c].state = sn.st + (size_t)c * N, sn.hb[c].slot2map = sn.s2m + (size_t)c * N;
        }
    }
    long long nRequested = 0, nMySolves = 0;
    const size_t recordBytes = cs_ba_output_record_bytes(bout);
    if (world > 1)
        for (int q = 0; q < 2; ++q) dRecvRec[q] = dev_zeros<unsigned char>(recordBytes);   // records solved by other ranks arrive here
    int* dApplyCnt = dev_zeros<int>(3);

    // inter-camera NCC matching every 4th frame: getNCCBlocks per camera on the full frame, the matrices per consecutive pair
    const int NCC_EVERY = 4;
    int wsS = 0, hsS = 0;
    CSCHK(cs_ncc_scaled_dims(W, H, 0.3, &wsS, &hsS));
    unsigned char* dSmall = dev_zeros<unsigned char>((size_t)nCams * wsS * hsS);
    unsigned char* dBlk = dev_zeros<unsigned char>((size_t)nCams * N * 128);
    double* dAbc = dev_zeros<double>((size_t)nCams * N * 4);
    int* dValid = dev_zeros<int>((size_t)nCams * N);
    const int NCC_PAIR_CAP = 1 << 16;   // passing pairs kept per camera pair and run (cs_ncc_epi_pairs_dev)
    cs_ncc_pair* dPairs = (cs_ncc_pair*)dev_zeros<unsigned char>((size_t)(nCams > 1 ? nCams - 1 : 1) * NCC_PAIR_CAP * sizeof(cs_ncc_pair));
    int* dPairCount = dev_zeros<int>(nCams);
    void* dNpScratch = dev_zeros<unsigned char>(cs_newpts_scratch_bytes(nCams, N));
    int* dNpCounts = dev_zeros<int>(4 + nCams);
    int nccRuns = 0;

    hipEvent_t kltDone[2], destFree[2];
    for (int b = 0; b < 2; ++b) {
        HIPCHK(hipEventCreateWithFlags(&kltDone[b], hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&destFree[b], hipEventDisableTiming));
    }
    auto img_ptrs = [&](int f, const void** out) {   // the rank's own cameras
        for (int k = 0; k < nc; ++k) out[k] = dFrames[c0 + k] + imgBytes * f;
    };
    std::vector<int*> s2mPtrs(nCams);
    for (int c = 0; c < nCams; ++c) s2mPtrs[c] = dS2M + (size_t)c * N;
    unsigned char* dAttached = dev_zeros<unsigned char>((size_t)nMap * nCams);
    unsigned char* dRegged = dev_zeros<unsigned char>(nMap);
    void* dDecScratch = dev_zeros<unsigned char>(cs_register_decide_scratch_bytes(nCams, N, nMap));
    void* dMergeScratch = dev_zeros<unsigned char>(cs_register_decide_merge_scratch_bytes(nMap, P_REG, nCams));
    int* dDecCnt = dev_zeros<int>(4);
    int* dMergeCnt = dev_zeros<int>(4);
    int nMergeFrames = 0;
    // Const::PIXEL_ERR_VAR = 10 (reference src/app/SL_GlobParam.cpp:37) is a VARIANCE (the retired define beside it: `SLAM_PIXEL_ERR_VAR 4
    // //2 pixels error`, src/slam/SL_Define.h:16); this library's getProjectionCovMat / seqTriangulate / getTriangulateCovMat take a standard
    // deviation: sqrt(10) px.  COSLAM_PIXEL_ERR_STD=1: the constant handed over as it is (a 10 px gate: rounds 1-4).  DESIGN.md 5.1
    const double PIXVAR = 10.0;
    const bool pixIsStd = getenv("COSLAM_PIXEL_ERR_STD") && getenv("COSLAM_PIXEL_ERR_STD")[0] == '1';
    const double PIX = pixIsStd ? PIXVAR : sqrt(PIXVAR), PIX_CLASSIFY = pixIsStd ? 12.0 : sqrt(12.0);
    auto step = [&](int i, bool key) {
        const int f = order[i % orderLen], fn = order[(i + 1) % orderLen], b = i & 1;
        const void *cur[16], *nxt[16];
        void *dst[16], *cnt[16];
        img_ptrs(f, cur);
        img_ptrs(fn, nxt);
        for (int k = 0; k < nc; ++k) dst[k] = dDest[b][c0 + k], cnt[k] = dCnt[c0 + k];
        if (i >= 2) HIPCHK(hipStreamWaitEvent(kltS, destFree[b], 0));
        CSCHK(cs_klt_group_prefetch_dev(grp, nxt));
        CSCHK(cs_klt_group_redetect_dev(grp, cur, dst, cnt));
        CSCHK(cs_klt_group_advance(grp));
        HIPCHK(hipEventRecord(kltDone[b], kltS));
        HIPCHK(hipStreamWaitEvent(poseS, kltDone[b], 0));
        const int src = (i + 1) & 1, dsti = i & 1;
        // output() of the window whose lag ends at this frame, before anything of frame i touches the map: the pose stream waits ON
        // THE DEVICE for the worker to publish the record; the host goes on enqueueing
        if (!due.empty() && due.front().frame == i && !due.front().frames.empty() && (i - 1) - due.front().frames.front() + 1 > 4096) {
            due.erase(due.begin());   // (the camera graphs would start behind the pose history's oldest frame: the record is consumed, nothing written back)
            ++kfNotApplied;
        }
        if (!due.empty() && due.front().frame == i) {
            void* rec = nullptr;
            if (due.front().owner == rank)
                CSCHK(cs_ba_output_wait_dev(bout, due.front().seq, (void*)poseS, 0, &rec));
            else
                rec = dRecvRec[due.front().k & 1];
            if (world > 1) CSCHK(cs_comm_broadcast_dev(comm, (void*)poseS, rec, recordBytes, due.front().owner));   // the owner's record to every replica
            if (!due.front().frames.empty())
                CSCHK(cs_ba_output_apply_frames_dev(bout, rec, due.front().seq, (void*)poseS, hist, win, pu.data(), dPf, nMap, dMap, dCov, dMapFlags, PIX,
                                                    due.front().frames.data(), (int)due.front().frames.size(), dR[src], dT[src], dApplyCnt));
            else
                CSCHK(cs_ba_output_apply_seq_dev(bout, rec, due.front().seq, (void*)poseS, hist, win, pu.data(), dPf, nMap, dMap, dCov, dMapFlags, PIX,
                                                 due.front().firstKey, keyEvery, dR[src], dT[src], dApplyCnt));
            due.erase(due.begin());
            ++nApplied;
        }
        CSCHK(cs_klt_handback_dev(dev, (void*)poseS, nc, hbOwn[b].data(), N, W, H, nColBlk, nRowBlk, PTS, i));
        CSCHK(cs_pose_intracam_batch_dev(dev, (void*)poseS, nc, PTS, dKall, dR[src] + 9 * c0, dT[src] + 3 * c0, dNpts + c0, nullptr,
                                         dMs + (size_t)c0 * PTS * 3, dms + (size_t)c0 * PTS * 2, 10.0, dR[dsti] + 9 * c0, dT[dsti] + 3 * c0,
                                         dOpt + c0, dOk + c0));
        if (world > 1) {
            // the merge step: every camera's {dest[], R, t} to every rank (ONE all-gather), the other ranks' poses into the pose arrays, their
            // cameras through the same hand-back
            CSCHK(cs_exchange_allgather_dev(xchg, (void*)poseS, (const void* const*)dst, dR[dsti] + 9 * c0, dT[dsti] + 3 * c0));
            CSCHK(cs_exchange_unpack_poses_dev(xchg, (void*)poseS, dR[dsti], dT[dsti], 1));
            CSCHK(cs_klt_handback_dev(dev, (void*)poseS, nCams - nc, hbOther.data(), N, W, H, nColBlk, nRowBlk, PTS, i));
        }
        // parallelPoseUpdate(false): the gate + seqTriangulate loop of poseUpdate3D, detectDynamicFeaturePoints(20, 5, 3, MAX_EPI_ERR)
        // and mapPointsClassify(12.0) (SL_CoSLAM.cpp:385): the uncertain / dynamic points of this frame decided again -- CoSLAM::poseUpdate as two
        // launches (the gate's lane of a point also lists it for the classification)
        CSCHK(cs_pose_update_classify_frame_dev(hist, (void*)poseS, pu.data(), dPf, nMap, dR[dsti], dT[dsti], dMap, dCov, dMapFlags, 0, PIX, i, 20, 5,
                                                3, 6.0, nullptr, nullptr, nullptr, nullptr, nullptr, dNewPt, dSfn, dFirstFrm, PIX_CLASSIFY, nullptr));
        if (kfDrives)   // genNewMapPoints' first half (:1294-1346): is a camera ready for a key frame; addKeyFrame's bookkeeping when `decrease` holds
            CSCHK(cs_keyframe_ready_dev(dev, (void*)poseS, nCams, N, kfCams[dsti].data(), nMap, dMap, dMapFlags, dFirstFrm, i, kfRatio, 5.0, kfMinTranslation, 1,
                                        dKfReady, dKfCnt, dKfCen, dKfStats));
        // genNewMapPoints every 4th frame -- BEFORE currentMapPointsRegister, as in the reference's frame (src/gui/CoSLAMThread.cpp:104-118):
        // the new map points take their features before the current points' registration looks at them
        if (nCams >= 2 && i % NCC_EVERY == 0) {
            // NewMapPtsNCC::addSlam's features: this frame's, on tracks of more than three frames, unmapped or on a false point
            CSCHK(cs_ncc_candidate_mask_dev(dev, (void*)poseS, nCams, N, dState, dS2M, dSpan, dMapFlags, nMap, 3, dValid, 0));
            // the whole run in a handful of launches: resize + cutter of all cameras, the passing pairs of all camera pairs, then
            // seeds + disparity guide + greedy matches, featTracksFromMatches, reconstructTracks, output: new points behind *dMapCount
            std::vector<cs_ncc_cam> ncams(nCams);
            std::vector<cs_ncc_pair_job> jb(nCams - 1);
            std::vector<const cs_ncc_pair*> pairPtr(nCams - 1);
            std::vector<const int*> cntPtr(nCams - 1);
            for (int c = 0; c < nCams; ++c) {
                ncams[c].img = dFrames[c] + imgBytes * f, ncams[c].x = dXY + (size_t)c * 2 * N, ncams[c].y = dXY + (size_t)c * 2 * N + N;
                ncams[c].scaled = dSmall + (size_t)c * wsS * hsS, ncams[c].blocks = dBlk + (size_t)c * N * 128, ncams[c].abc = dAbc + (size_t)c * N * 4;
                ncams[c].valid = dValid + (size_t)c * N;
            }
            for (int c = 0; c + 1 < nCams; ++c) {
                memset(&jb[c], 0, sizeof(jb[c]));
                jb[c].dF = dFm + 9 * (size_t)c;   // E and F from the poses this frame has solved (matchBetween, SL_NewMapPointsInterCam.cpp:284-292)
                jb[c].camA = c, jb[c].camB = c + 1, jb[c].pairs = dPairs + (size_t)c * NCC_PAIR_CAP, jb[c].count = dPairCount + c;
                pairPtr[c] = jb[c].pairs, cntPtr[c] = jb[c].count;
            }
            // the blocks of the rank's own cameras (it holds their images); N > 1: blocks and line coefficients of every camera to every
            // rank (two all-gathers in place, 256 + 64 KB per camera, every 4th frame); the candidate masks come from replicated state
            CSCHK(cs_ncc_get_blocks_group_dev(dev, (void*)poseS, nc, ncams.data() + c0, W, H, N, 0.3));
            if (world > 1) {
                CSCHK(cs_comm_allgather_dev(comm, (void*)poseS, dBlk + (size_t)c0 * N * 128, dBlk, (size_t)nc * N * 128));
                CSCHK(cs_comm_allgather_dev(comm, (void*)poseS, dAbc + (size_t)c0 * N * 4, dAbc, sizeof(double) * (size_t)nc * N * 4));
            }
            {
                std::vector<int> ca(nCams - 1), cb(nCams - 1);
                std::vector<const double*> ik(nCams, diK);
                for (int c = 0; c + 1 < nCams; ++c) ca[c] = c, cb[c] = c + 1;
                CSCHK(cs_ncc_fmats_dev(dev, (void*)poseS, nCams, nCams - 1, ca.data(), cb.data(), ik.data(), dR[dsti], dT[dsti], dFm));
            }
            CSCHK(cs_ncc_epi_pairs_group_dev(dev, (void*)poseS, nCams, ncams.data(), N, nCams - 1, jb.data(), 50.0, 0.80, NCC_PAIR_CAP));
            CSCHK(cs_newpts_from_pairs_dev(dev, (void*)poseS, nCams, N, pu.data(), pairPtr.data(), cntPtr.data(), NCC_PAIR_CAP, dR[dsti], dT[dsti],
                                           dMap, dCov, dMapFlags, dNewPt, dFirstFrm, dPf, nMap, dMapCount, i, 80.0, 3.0, PIX, 2, W, H, dNpScratch,
                                           dNpCounts));
            ++nccRuns;
        }
        // currentMapPointsRegister, search step: curMapPts of this frame as a list (the points with a feature of this frame, wherever they
        // sit in the map -- the ones genNewMapPoints just appended included), ONE pass over it; the tables are indexed by the map index.
        // (activeMapPointsRegister's search is not run: the reference's attach loop behind it cannot be reached, tests/cxx/ref_active_test.cpp)
        CSCHK(cs_register_list_current_cap_dev(dev, (void*)poseS, nCams, nMap, dMapCount, dPf, dMapFlags, dCurList, dCurCount, reg[0].slot, P_REG, dCurOverflow));
        {
            cs_register_pass ps[1];
            memset(ps, 0, sizeof(ps));
            ps[0].P = P_REG, ps[0].sigmaSearch = PIX, ps[0].maxDist = 3 * PIXVAR, ps[0].sigmaMerge = PIX;   // (maxDist: a common scale of a search's distances)
            ps[0].M = dMap, ps[0].cov = dCov, ps[0].pointFeat = dPf, ps[0].list = dCurList;
            ps[0].mapFlags = dMapFlags, ps[0].maxDistDynamic = 4 * PIXVAR;   // (the certainly dynamic points' scale: SL_CoSLAM.cpp:973)
            ps[0].slot = reg[0].slot, ps[0].m = reg[0].m, ps[0].var = reg[0].var, ps[0].dist = reg[0].dist, ps[0].flags = reg[0].flags;
            CSCHK(cs_register_search_passes_range_dev(dev, (void*)poseS, nCams, c0, nc, rc[dsti].data(), N, W, H, 1, ps));   // the own cameras' columns
        }
        // staticCheckMergability of the candidates over their WHOLE tracks (SL_CoSLAM.cpp:714-729, :768) as a running verdict
        CSCHK(cs_register_mergability_running_list_dev(hist, (void*)poseS, c0, nc, pu.data(), nMap, dCurList, P_REG, dMap, dCov, reg[0].slot, reg[0].flags, PIX,
                                                       0.5, dMergeCache, dMergeable, dMergeRun));
        if (world > 1) {
            // the own cameras' columns of the candidate tables (the listed rows only) to every rank: ONE all-gather, then every rank takes the
            // same decisions on its replica
            CSCHK(cs_register_candidates_pack_list_dev(dev, (void*)poseS, P_REG, nCams, c0, nc, dCurList, reg[0].slot, reg[0].flags, dMergeable, dCandSend));
            CSCHK(cs_comm_allgather_dev(comm, (void*)poseS, dCandSend, dCandRecv, sizeof(int) * (size_t)3 * nc * P_REG));
            CSCHK(cs_register_candidates_unpack_list_dev(dev, (void*)poseS, P_REG, nCams, nc, rank, dCurList, dCandRecv, reg[0].slot, reg[0].flags, dMergeable));
        }
        // the decision (curStaticPointsRegInGroup, bMerge false: who attaches which feature), then refineMapPoint of the points that gained one
        // currentMapPointsRegister's decisions: the certainly static points, behind them the certainly dynamic ones (kinds 3), one call;
        // every 50th frame with bMerge (CoSLAMThread.cpp:117-118): the static points' walks one after the other, checkUnify at a conflict
        // refineMapPoint of the points that gained a feature: with the references brought up to this frame first (tracked on / first feature /
        // re-linked behind an older one / stale / detached: cs_feat_ref_advance_dev, idempotent within a frame)
        auto refine = [&]() {
            if (chains) {
                CSCHK(cs_feat_ref_advance_dev(hist, (void*)poseS, pu.data(), nMap, dPf, i, dFref, dRstat, dFrefCnt));
                CSCHK(cs_refine_map_points_ref_dev(hist, (void*)poseS, pu.data(), dFref, nMap, dRegged, dMap, dCov, PIX, nullptr));
            } else
                CSCHK(cs_refine_map_points_dev(hist, (void*)poseS, pu.data(), dPf, nMap, dRegged, dMap, dCov, PIX, nullptr));
        };
        int kinds = 3;
        if (i % 50 == 0) {
            CSCHK(cs_register_decide_merge_list_dev(hist, (void*)poseS, pu.data(), nMap, 0, dCurList, P_REG, reg[0].slot, reg[0].flags, dMergeable, dMapFlags, dPf, dMap, dCov,
                                               PIX, dAttached, dRegged, dMergeScratch, dMergeCnt, /*onlyCam*/ -1));
            refine();
            ++nMergeFrames;
            kinds = 2;
        }
        if (fusedRounds && kinds == 3) {
            // the same with the lists built by the walks themselves and advance + refine as one launch: 4 launches per round instead of 6, 2 instead
            // of 3 behind the single pass (cs_register_decide_kinds_rounds_dev, cs_feat_ref_advance_refine_dev)
            CSCHK(cs_register_decide_kinds_rounds_dev(dev, (void*)poseS, nCams, N, nMap, 0, reg[0].slot, reg[0].flags, dMergeable, dMapFlags, dPf, s2mPtrs.data(),
                                                      dAttached, dRegged, dDecScratch, 0, dDecCnt, -1, 3, RV_ROUNDS > 0 ? dRvLists : nullptr, RV_CAP, RV_ROUNDS,
                                                      dRvCounts, dRvVisit, dRvNext));
            CSCHK(cs_feat_ref_advance_refine_dev(hist, (void*)poseS, pu.data(), nMap, dPf, i, dFref, dRstat, dFrefCnt, dCurList, P_REG, 1, dRegged, 0, dMap, dCov, PIX));
            for (int r = 0; r < RV_ROUNDS; ++r) {
                int* list = dRvLists + (size_t)r * RV_CAP;
                cs_register_pass ps[1];
                memset(ps, 0, sizeof(ps));
                ps[0].P = RV_CAP, ps[0].sigmaSearch = PIX, ps[0].maxDist = 3 * PIXVAR, ps[0].sigmaMerge = PIX;
                ps[0].M = dMap, ps[0].cov = dCov, ps[0].pointFeat = dPf, ps[0].list = list;
                ps[0].mapFlags = dMapFlags, ps[0].maxDistDynamic = 4 * PIXVAR;
                ps[0].slot = reg[0].slot, ps[0].m = reg[0].m, ps[0].var = reg[0].var, ps[0].dist = reg[0].dist, ps[0].flags = reg[0].flags;
                CSCHK(cs_register_search_passes_range_dev(dev, (void*)poseS, nCams, 0, nCams, rc[dsti].data(), N, W, H, 1, ps));
                CSCHK(cs_register_mergability_running_list_dev(hist, (void*)poseS, 0, nCams, pu.data(), nMap, list, RV_CAP, dMap, dCov, reg[0].slot, reg[0].flags,
                                                               PIX, 0.0, dMergeCache, dMergeable, nullptr));
                const bool more = r + 1 < RV_ROUNDS;
                CSCHK(cs_register_revisit_decide_next_dev(dev, (void*)poseS, nCams, N, nMap, RV_CAP, 0, 3, list, dRvNext, dRvVisit, reg[0].slot, reg[0].flags, dMergeable,
                                                          dMapFlags, dPf, s2mPtrs.data(), dAttached, dRvReg[0], dDecScratch, dCurList, dCurCount, P_REG, dRvCnt,
                                                          dRvCounts + r, more ? list + RV_CAP : nullptr, more ? dRvCounts + r + 1 : nullptr, dRvCounts + RV_ROUNDS));
                CSCHK(cs_feat_ref_advance_refine_dev(hist, (void*)poseS, pu.data(), nMap, dPf, i, dFref, dRstat, dFrefCnt, list, RV_CAP, 0, dRvReg[0], 1, dMap, dCov, PIX));
            }
            kinds = 0;   // (done)
        } else {
            CSCHK(cs_register_decide_kinds_dev(dev, (void*)poseS, nCams, N, nMap, 0, reg[0].slot, reg[0].flags, dMergeable, dMapFlags, dPf, s2mPtrs.data(),
                                               dAttached, dRegged, dDecScratch, /*nSweeps: until settled*/ 0, dDecCnt, /*onlyCam*/ -1, kinds));
            refine();
            // the reference's SECOND VISITS (SL_CoSLAM.cpp:864-869, :889-893): the points that registered are refined and visited again in their next
            // camera's loop -- rounds of list + search + whole-track mergability + walks + refine over just those points, every rank for ALL cameras
            // on its replica (cs_register_revisit_*; tools/r06_exact_vs_single.py: with two rounds the map is the reference order's, frame after frame)
        }
        if (kinds == 3) {
            unsigned char* regIn = dRegged;
            for (int r = 0; r < RV_ROUNDS; ++r) {
                unsigned char* regOut = dRvReg[r & 1];
                CSCHK(cs_register_revisit_list_dev(dev, (void*)poseS, nCams, nMap, RV_CAP, r == 0, dPf, dAttached, regIn, r == 0, regOut, dRvVisit, dRvNext, dRvList,
                                                   dRvListCnt));
                cs_register_pass ps[1];
                memset(ps, 0, sizeof(ps));
                ps[0].P = RV_CAP, ps[0].sigmaSearch = PIX, ps[0].maxDist = 3 * PIXVAR, ps[0].sigmaMerge = PIX;
                ps[0].M = dMap, ps[0].cov = dCov, ps[0].pointFeat = dPf, ps[0].list = dRvList;
                ps[0].mapFlags = dMapFlags, ps[0].maxDistDynamic = 4 * PIXVAR;
                ps[0].slot = reg[0].slot, ps[0].m = reg[0].m, ps[0].var = reg[0].var, ps[0].dist = reg[0].dist, ps[0].flags = reg[0].flags;
                CSCHK(cs_register_search_passes_range_dev(dev, (void*)poseS, nCams, 0, nCams, rc[dsti].data(), N, W, H, 1, ps));
                CSCHK(cs_register_mergability_running_list_dev(hist, (void*)poseS, 0, nCams, pu.data(), nMap, dRvList, RV_CAP, dMap, dCov, reg[0].slot, reg[0].flags,
                                                               PIX, 0.0, dMergeCache, dMergeable, nullptr));
                CSCHK(cs_register_revisit_decide_dev(dev, (void*)poseS, nCams, N, nMap, RV_CAP, 0, 3, dRvList, dRvNext, dRvVisit, reg[0].slot, reg[0].flags, dMergeable,
                                                     dMapFlags, dPf, s2mPtrs.data(), dAttached, regOut, dDecScratch, dCurList, dCurCount, P_REG, dRvCnt, dRvListCnt));
                if (chains) {
                    CSCHK(cs_feat_ref_advance_list_dev(hist, (void*)poseS, pu.data(), nMap, dPf, i, dFref, dRstat, dFrefCnt, dRvList, RV_CAP));
                    CSCHK(cs_refine_map_points_ref_dev(hist, (void*)poseS, pu.data(), dFref, nMap, regOut, dMap, dCov, PIX, nullptr));
                } else
                    CSCHK(cs_refine_map_points_dev(hist, (void*)poseS, pu.data(), dPf, nMap, regOut, dMap, dCov, PIX, nullptr));
                regIn = regOut;
            }
        }
        // the tracker of frame i + 2 is released at the END of the frame's pose work (released right behind the hand-back it runs two frames
        // ahead and under more of the pose stream's kernels: -10 %, profiles/r04_ab_runs.txt)
        HIPCHK(hipEventRecord(destFree[b], poseS));
        // a key frame's actions: the inter-camera solve, the frame's records and poses into the window's ring, the window's request
        auto key_frame_actions = [&](int f, const cs_handback_cam* cams, const double* Rk, const double* tk, bool placed) {
            // InterCamPoseEstimator::addMapPoints + apply: every camera's current pose, the block-voted static features' map points
            // fixed, the dynamic points free; sigma 6, 3 x 40
            // (key frame k's inter-camera solve on rank (k + world / 2) % world, its window on rank k % world: the two chains on different GPUs)
            if ((nKey + world / 2) % world == rank)
                CSCHK(cs_ba_solve_intercam_async(ic.ws, icam, (void*)poseS, icCams.data(), W, H, nColBlk, nRowBlk, dR[dsti], dT[dsti], dMap, dMapFlags,
                                                 dNewPt, dPf, 6.0, 3, 40));
            ++nKey;
            // requestForBA(5, 2, 2, 30): the numCams * 2 oldest key cameras and 2 points held, maxIter 2, inner 10; static points only
            CSCHK(cs_ba_window_push_dev(win, (void*)poseS, cams, dK, 1, Rk, tk, f));
            kfPushedFrames.push_back(f);
            if ((int)kfPushedFrames.size() > WIN_KF) kfPushedFrames.erase(kfPushedFrames.begin());
            if (++nPushed >= WIN_KF) {
                // window k is solved by rank k % world (the ring is identical on every rank); its packed result is broadcast and applied by
                // every rank baLag key-frame intervals behind its key frame
                const int k = (int)nRequested++, owner = k % world;
                long long seq = k / world;
                if (owner == rank) {
                    CSCHK(cs_ba_solve_window_flags_async(joint.ws, win, (void*)poseS, dMap, dMapFlags, 2 * nCams, 2, 6.0, 2, 10));
                    seq = nMySolves++;
                }
                Due d{f + baLag * keyEvery, f - (WIN_KF - 1) * keyEvery, seq, k, owner, {}};
                if (placed) d.frames = kfPushedFrames, d.firstKey = kfPushedFrames.front();
                due.push_back(d);
            }
        };
        if (kfDrives) {
            // this frame's decision word, records and poses into slot i % (LAG + 1) of the ring (no host wait), then the decision of frame i - LAG
            KfSnap& sn = kfRing[i % (kfLag + 1)];
            CSCHK(cs_keyframe_snapshot_dev(dev, (void*)poseS, nCams, N, dXY, dState, dS2M, dR[dsti], dT[dsti], dKfReady + nCams + 1, sn.xy, sn.st, sn.s2m, sn.R,
                                           sn.t, sn.word));
            HIPCHK(hipEventRecord(sn.ev, poseS));
            sn.frame = i;
            const int f = i - kfLag;
            KfSnap& old = kfRing[((f % (kfLag + 1)) + (kfLag + 1)) % (kfLag + 1)];
            if (f >= 1 && old.frame == f) {
                HIPCHK(hipEventSynchronize(old.ev));   // (a frame LAG behind: fired long ago unless the host has caught up with the device)
                if (*old.word) {
                    kfPlaced.push_back(f);
                    key_frame_actions(f, old.hb.data(), old.R, old.t, true);
                }
            }
        } else if (key) {
            key_frame_actions(i, hb[b].data(), dR[dsti], dT[dsti], false);
        }
    };
    int* dBar = dev_zeros<int>(64);
    // a window / a rig that holds no usable point (every map point of its key frames false, say: the closed orbit starves after some thousands
    // of frames) is a solve with nothing to do -- it packed an empty record (ok = 0, applies nothing) -- not a failure of the
    // loop: counted (coslam_amd/frameloop.py: drain())
    int nEmptySolves = 0;
    auto wait_ws = [&](cs_ba* ws) {
        for (;;) {
            const int rc = cs_ba_wait(ws);
            if (rc == CS_OK) return;
            const char* e = cs_last_error();
            if (e && (strstr(e, "no map point has two feature points") || strstr(e, "no static feature point carries a map point"))) {
                ++nEmptySolves;
                continue;   // (the worker goes on with the next request: wait again)
            }
            fprintf(stderr, "cs_ba_wait failed (%d): %s\n", rc, e ? e : "?");
            exit(3);
        }
    };
    auto barrier = [&]() {
        wait_ws(ic.ws);
        wait_ws(joint.ws);
        HIPCHK(hipDeviceSynchronize());
        if (world > 1) {   // every rank has drained: a small all-gather as the barrier between the ranks
            CSCHK(cs_comm_allgather_dev(comm, (void*)poseS, dBar + rank, dBar, sizeof(int)));
            HIPCHK(hipDeviceSynchronize());
        }
    };

    // ---- first frame: detect, map association, first hand-back (GPUKLT::first + map initialisation stand-in) ----
    {
        const void* cur[16];
        void *dst[16], *cnt[16];
        img_ptrs(order[0], cur);
        for (int k = 0; k < nc; ++k) dst[k] = dDest[0][c0 + k], cnt[k] = dCnt[c0 + k];
        CSCHK(cs_klt_group_detect_dev(grp, cur, dst, cnt));
        CSCHK(cs_klt_group_advance(grp));
        CSCHK(cs_klt_group_synchronize(grp));
        if (world > 1) {   // every camera's first dest[] to every rank
            CSCHK(cs_exchange_allgather_dev(xchg, (void*)poseS, (const void* const*)dst, dR[0] + 9 * c0, dT[0] + 3 * c0));
            HIPCHK(hipDeviceSynchronize());
        }
    }
    auto associate = [&]() {
        std::vector<cs_klt_feature> d(N);
        std::vector<int> s2m(N);
        for (int c = 0; c < nCams; ++c) {
            const void* from = (c >= c0 && c < c0 + nc) ? (const void*)dDest[0][c] : (const void*)(xRecv + (size_t)c * xRecBytes);
            HIPCHK(hipMemcpy(d.data(), from, sizeof(cs_klt_feature) * N, hipMemcpyDeviceToHost));
            const std::vector<double>& uv = visUV[c];
            const int nv = (int)visIdx[c].size();
            for (int s = 0; s < N; ++s) {
                s2m[s] = -1;
                if (d[s].status < 0) continue;
                const double px = (double)d[s].pos[0] * W, py = (double)d[s].pos[1] * H;
                double best = 1.0;  // nearest projected point within 1 px
                for (int q = 0; q < nv; ++q) {
                    const double dx = uv[2 * q] - px, dy = uv[2 * q + 1] - py, dd = std::sqrt(dx * dx + dy * dy);
                    if (dd < best) best = dd, s2m[s] = visIdx[c][q];
                }
            }
            HIPCHK(hipMemcpy(dS2M + (size_t)c * N, s2m.data(), sizeof(int) * N, hipMemcpyHostToDevice));
        }
    };
    associate();
    CSCHK(cs_klt_handback_dev(dev, (void*)poseS, nc, hbOwn[0].data(), N, W, H, nColBlk, nRowBlk, PTS, 0));
    if (world > 1) CSCHK(cs_klt_handback_dev(dev, (void*)poseS, nCams - nc, hbOther.data(), N, W, H, nColBlk, nRowBlk, PTS, 0));
    HIPCHK(hipDeviceSynchronize());
    associate();  // (the first hand-back starts every track as new, i.e. unmapped: put the map back)
    HIPCHK(hipDeviceSynchronize());
    // frame 0 into the history as well (its pixels and poses: the first term of every track born in it, which a whole-track mergability
    // walk ends with); the dynamic test has nothing to say about one-frame tracks
    CSCHK(cs_detect_dynamic_dev(hist, (void*)poseS, 0, nCams, pu.data(), dR[0], dT[0], nMap, dMapFlags, 0, 20, 5, 3, 6.0, nullptr));
    HIPCHK(hipDeviceSynchronize());

    if (kfDrives) {   // nMappedPts of frame 0's key pose: the certainly static mapped features of the frame (enable_keyframe_decision, coslam_amd/frameloop.py)
        std::vector<int> st((size_t)nCams * N), sm((size_t)nCams * N), km(nCams, 0);
        std::vector<unsigned char> fl(nMap);
        HIPCHK(hipMemcpy(st.data(), dState, sizeof(int) * st.size(), hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(sm.data(), dS2M, sizeof(int) * sm.size(), hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(fl.data(), dMapFlags, fl.size(), hipMemcpyDeviceToHost));
        for (int c = 0; c < nCams; ++c)
            for (int q = 0; q < N; ++q) {
                const int sv = st[(size_t)c * N + q], m = sm[(size_t)c * N + q];
                if ((sv == 0 || sv == 1) && m >= 0 && m < nMap && (fl[m] & 7) == 0) ++km[c];
            }
        HIPCHK(hipMemcpy(dKfMapped, km.data(), sizeof(int) * nCams, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(dKfSelfR, dR[0], sizeof(double) * 9 * nCams, hipMemcpyDeviceToDevice));
        HIPCHK(hipMemcpy(dKfSelfT, dT[0], sizeof(double) * 3 * nCams, hipMemcpyDeviceToDevice));
    }
    // set-up (one key-frame interval: graph capture in the BA workers, lazy code-object loading), warm-up, timed loop
    // (with the window: 5 key-frame intervals, so that every timed solve has its 5 key frames = 5 x nCams cameras); the frame
    // sequence runs on through set-up, warm-up and the timed region
    int nDone = 0;
    auto run = [&](int n) {
        for (int q = 0; q < n; ++q, ++nDone) step(nDone + 1, keyEvery > 0 && nDone % keyEvery == 0);
    };
    run(WIN_KF * std::max(keyEvery, 1) + 1);
    barrier();
    run((std::max(keyEvery, 1) - nDone % std::max(keyEvery, 1)) % std::max(keyEvery, 1));
    run(4 * std::max(keyEvery, 1));   // (one set-up round: bench.py --setup-rounds 1)
    barrier();
    if (timedFrom - warmup - 1 > nDone) {   // untimed, like bench.py's set-up loop: up to where its warm-up started
        run(timedFrom - warmup - 1 - nDone);
        barrier();
    }
    run(warmup);
    barrier();
    const int applied0 = nApplied;
    int rvCnt0[4] = {0, 0, 0, 0};   // (the second visits' counters at the start of the timed region)
    HIPCHK(hipMemcpy(rvCnt0, dRvCnt, sizeof(rvCnt0), hipMemcpyDeviceToHost));
    const auto t0c = std::chrono::steady_clock::now();
    run(steps);
    const auto t1c = std::chrono::steady_clock::now();
    barrier();
    const auto t2c = std::chrono::steady_clock::now();
    const double dt = std::chrono::duration<double>(t2c - t0c).count(), dtHost = std::chrono::duration<double>(t1c - t0c).count();

    // sanity of what was computed: live features, pose flags, the solves' statistics
    int okAll = 1, minLive = N;
    {
        std::vector<int> ok(nCams);
        HIPCHK(hipMemcpy(ok.data(), dOk, sizeof(int) * nCams, hipMemcpyDeviceToHost));
        for (int c = c0; c < c0 + nc; ++c) okAll &= (ok[c] != 0);   // (the rank's own cameras: it solves their poses)
        std::vector<cs_klt_feature> d(N);
        const int last = nDone & 1;
        for (int c = c0; c < c0 + nc; ++c) {
            HIPCHK(hipMemcpy(d.data(), dDest[last][c], sizeof(cs_klt_feature) * N, hipMemcpyDeviceToHost));
            int live = 0;
            for (const cs_klt_feature& q : d) live += q.status >= 0;
            minLive = std::min(minLive, live);
        }
    }
    cs_ba_stats sj, si;
    memset(&sj, 0, sizeof(sj)), memset(&si, 0, sizeof(si));
    int jC = joint.C, jP = joint.P, jO = joint.nObs;
    if (win) CSCHK(cs_ba_window_last_problem(win, &jC, &jP, &jO, nullptr, nullptr));
    if (nMySolves > 0) CSCHK(cs_ba_download(joint.ws, jC, jP, jO, nullptr, nullptr, nullptr, nullptr, &sj));   // (a rank solves every world-th window)
    int iC = 0, iP = 0, iO = 0, iS = 0;
    CSCHK(cs_ba_intercam_last_problem(icam, &iC, &iP, &iO, &iS, nullptr));
    if (iC > 0) CSCHK(cs_ba_download(ic.ws, iC, iP, iO, nullptr, nullptr, nullptr, nullptr, &si));
    // the state every rank must agree on after the last frame (and a one-rank run must reproduce): FNV-1a over the map points in use, their
    // flags, every camera's slot -> point table and track spans, the current poses
    unsigned long long digest = 1469598103934665603ull;
    {
        int cnt = 0;
        HIPCHK(hipMemcpy(&cnt, dMapCount, sizeof(int), hipMemcpyDeviceToHost));
        auto eat = [&](const void* dptr, size_t bytes) {
            std::vector<unsigned char> h(bytes);
            HIPCHK(hipMemcpy(h.data(), dptr, bytes, hipMemcpyDeviceToHost));
            for (unsigned char v : h) digest = (digest ^ v) * 1099511628211ull;
        };
        eat(dMap, sizeof(double) * 3 * (size_t)cnt), eat(dCov, sizeof(double) * 9 * (size_t)cnt), eat(dMapFlags, (size_t)cnt);
        eat(dS2M, sizeof(int) * (size_t)nCams * N), eat(dSpan, sizeof(int) * (size_t)nCams * 2 * N);
        eat(dR[nDone & 1], sizeof(double) * 9 * nCams), eat(dT[nDone & 1], sizeof(double) * 3 * nCams);
    }
    int mapCountNow = 0, npCounts[4] = {0, 0, 0, 0};
    HIPCHK(hipMemcpy(&mapCountNow, dMapCount, sizeof(int), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(npCounts, dNpCounts, sizeof(npCounts), hipMemcpyDeviceToHost));
    int curOverflow = 0;
    HIPCHK(hipMemcpy(&curOverflow, dCurOverflow, sizeof(int), hipMemcpyDeviceToHost));
    int rvCnt[4] = {0, 0, 0, 0}, rvListCnt[4] = {0, 0, 0, 0};
    HIPCHK(hipMemcpy(rvCnt, dRvCnt, sizeof(rvCnt), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(rvListCnt, dRvListCnt, sizeof(rvListCnt), hipMemcpyDeviceToHost));
    if (fusedRounds) HIPCHK(hipMemcpy(&rvListCnt[1], dRvCounts + RV_ROUNDS, sizeof(int), hipMemcpyDeviceToHost));
    std::string placedJson = "[";
    for (size_t q = 0; q < kfPlaced.size(); ++q) placedJson += (q ? ", " : "") + std::to_string(kfPlaced[q]);
    placedJson += "]";
    int decUnsettled = 0;   // (the decision scratch's last int: sticky "some call's sweeps did not settle")
    HIPCHK(hipMemcpy(&decUnsettled, (char*)dDecScratch + cs_register_decide_scratch_bytes(nCams, N, nMap) - sizeof(int), sizeof(int),
                     hipMemcpyDeviceToHost));
    printf("{\"frames_per_s\": %.3f, \"ms_per_step\": %.5f, \"steps\": %d, \"warmup\": %d, \"host_enqueue_ms_per_step\": %.5f, "
           "\"cams_per_tracker_launch\": %d, \"pose_ok\": %s, \"min_live_features\": %d, \"joint_lm_steps\": %d, \"joint_cost\": %.6f, "
           "\"intercam_lm_steps\": %d, \"intercam_cost\": %.6f, \"ncc_runs\": %d, \"joint_ba_from_window\": %s, \"joint_cameras\": %d, "
           "\"joint_points\": %d, \"joint_measurements\": %d, \"ba_lag\": %d, \"windows_applied_in_timed_region\": %d, \"apply_wait_errors\": %d, "
           "\"intercam_static_points\": %d, \"intercam_dynamic_points\": %d, \"map_points_at_start\": %d, \"map_points_in_use\": %d, "
           "\"map_capacity\": %d, \"new_map_points_last_run\": %d, \"register_decisions_unsettled\": %s, \"bmerge_frames\": %d, \"current_points_beyond_the_cap\": %d, \"second_visit_rounds\": %d, \"second_visit_features_attached\": %d, "
           "\"second_visit_conflicts\": %d, \"second_visit_conflicts_in_timed_region\": %d, \"second_visit_points_beyond_the_list\": %d, "
           "\"key_frames_placed_by_the_decision\": %s, \"keyframe_lag\": %d, \"frames_run\": %d, \"windows_requested\": %lld, \"windows_applied\": %d, \"windows_not_applied_history_too_short\": %d, "
           "\"rank\": %d, \"world\": %d, \"cameras_per_rank\": %d, \"transport\": \"%s\", \"digest\": \"%016llx\"}\n",
           steps / dt, dt / steps * 1e3, steps, warmup, dtHost / steps * 1e3, camsPerLaunch, okAll ? "true" : "false", minLive,
           sj.nIterTotal, sj.cost, si.nIterTotal, si.cost, nccRuns, win ? "true" : "false", jC, jP, jO, baLag, nApplied - applied0,
           cs_ba_output_wait_errors(bout), iS, iP - iS, nPts, mapCountNow, nMap, npCounts[0], decUnsettled ? "true" : "false", nMergeFrames, curOverflow, RV_ROUNDS, rvCnt[0], rvCnt[2], rvCnt[2] - rvCnt0[2], rvListCnt[1],
           kfDrives ? placedJson.c_str() : "null", kfDrives ? kfLag : 0, nDone, (long long)nRequested, nApplied, kfNotApplied,
           rank, world, nc, world == 1 ? "none" : (getenv("COSLAM_COMM") && !strncmp(getenv("COSLAM_COMM"), "host:", 5) ? "host segment (test)" : "rccl"),
           digest);
    fflush(stdout);
    if (xchg) cs_exchange_destroy(xchg);
    if (comm) cs_comm_destroy(comm);
    return 0;
}
