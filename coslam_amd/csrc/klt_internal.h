// klt_internal.h -- structs and launchers shared by the KLT translation units.
#pragma once
#include "cs_common.h"

struct CsCand {
    unsigned key;  // Morton code of (x,y), x minor == HistoPyramid traversal order
    float x, y, c;
};

struct CsTrackLevels {
    int L;
    int w[CS_MAX_LEVELS], h[CS_MAX_LEVELS];
    long long off[CS_MAX_LEVELS];
};

struct CsGainPassArgs {
    const cs_texel* lvl0;
    const cs_texel* lvl1;
    int Wl, Hl;
    float whx, why;
    int fw, fh, N, hw;
    const float* feat0;   // X0 list (features0_tex)
    const float* featIn;  // features_tex
    float* featOut;
    float sqrConvThr, ssdThr;
    float vr[4];
    float lambda, delta;
    int n1x[4], n1y[4];  // the four "betaN1" neighbour offsets in slot texels
};

// ---- rows tracker (klt_track_rows.hip): several features per wave, several cameras per launch ------------------------
constexpr int CS_MAX_CAMS = 16;  // SLAM_MAX_NUM is 13 (src/slam/SL_Define.h:11)

struct CsRowsCam {
    const cs_texel* pyr0;
    const cs_texel* pyr1;
    const float* feat0;      // X0 list (features0_tex)
    const float* featStart;  // iterate at entry (persistent) / features_tex (per pass)
    float* outLast;          // result of the last pass (persistent) / featOut (per pass)
    float* outPrev;          // result of the pass before it (persistent only)
    unsigned long long* gran;  // [passes + 1][N] {tag, beta} hand-off granules (persistent only)
    const unsigned* tagWord;   // frame-unique tag base
    int* err;
    cs_klt_feature* dest;  // fused k_post_track; null: not fused
    float* corner;
    unsigned long long* probe;  // diagnostic cycle counters (8 per wave) or null
};

struct CsRowsArgs {
    CsTrackLevels lv;
    int W, H, fw, fh, N, nIter, levelSkip, doSuppress;
    float sqrConvThr, ssdThr;
    float vr[4];
    float lambda, delta;
    int n1x[4], n1y[4];
    int level;  // per-pass kernel only: the level this launch works on
    int nCams;
    int xcdsPerCam;  // persistent kernel: 0 = grid (workgroups of a camera, cameras); q > 0 = 1-D grid, camera c on XCDs c q .. c q + q - 1
    CsRowsCam cam[CS_MAX_CAMS];
};

// window widths the rows tracker covers (2 * hw + 1 <= 15); wider windows use the wave-per-feature kernels
bool cs_rows_supported(int hw);
size_t cs_rows_lds_bytes(int hw);
int cs_rows_waves(int hw, int N);  // waves (= 64-thread workgroups) one camera needs
int cs_rows_max_resident_blocks(int hw, int device, int* blocksPerCu);
int cs_launch_track_rows_fused(const CsRowsArgs& a, int hw, hipStream_t stream);
int cs_launch_track_rows_pass(const CsRowsArgs& a, int hw, hipStream_t stream);

// mode 0: detect (all slots free, v3d_gpuklt.cpp:716-734)
// mode 1: detect with present points appended after the detected ones (:667-690)
// mode 2: redetect (free slots in ascending index, :775-797)
struct CsFillArgs {
    int mode, N, withGain, nPresentGiven;
    const float* present3;  // mode 1
    const CsCand* sel;
    int* ctr;
    cs_klt_feature* dest;
    float* list_a;  // provide target 1 (buffer1)
    float* list_b;  // provide target 2 (buffer2, with gain only; may be null)
    int* counts;    // user-visible counts[4]
    // the frame's last kernel leaves the candidate counter zeroed and bumps the tracker's frame tag for the next frame
    unsigned* tagWord;
};

// ---- camera batches: every frame-schedule kernel takes the cameras of a group as one more grid dimension ------------
// (the reference runs its cameras one after the other through one GL context, src/app/SL_CoSLAM.cpp:299-305)
struct CsFrontCam {
    const uint8_t* img;
    cs_texel* pyr;   // pyramid base (level 0 at offset 0)
    float* corner;   // raw cornerness map or null
    int* ctr;        // frame counters to zero or null
};
struct CsNonmaxCam {
    const float* in;
    float* out;
    CsCand* cand;
    int* ctr;
};
struct CsSelectCam {  // selection + slot fill of one camera
    const CsCand* cand;
    int* rankM;
    CsCand* sel;
    int maxKeepFixed;  // >= 0: use it; else N - tracked
    CsFillArgs fill;
};

int cs_launch_frame_front(const CsFrontCam* cams, int n, const CsPyrLayout& lay, int tap_mode, bool withCorner,
                          float minCornerness, float margin, hipStream_t stream);
int cs_launch_pyr_down_tail(cs_texel* const* pyrs, int n, const CsPyrLayout& lay, int tap_mode, int first, hipStream_t stream);
size_t cs_nonmax_lds_bytes(int d);
int cs_launch_tail_with_next_front(const CsNonmaxCam* nm, const CsSelectCam* sel, const CsFrontCam* next, int n, int W, int H,
                                   int d, int maxCand, int cap, const CsPyrLayout& lay, int tap_mode, float minCornerness,
                                   float margin, hipStream_t stream);
int cs_launch_track_nogain(const cs_texel* pyr0, const cs_texel* pyr1, const CsPyrLayout& lay, int levelSkip, int hw,
                           int nIterShader, float margin, float convThr, float ssdThr, int N, const float* featIn,
                           float* featOut, hipStream_t stream);
int cs_launch_track_gain_pass(const CsGainPassArgs& a, hipStream_t stream);
int cs_launch_reset_beta(float* feat, int N, hipStream_t stream);
int cs_launch_suppress_list(float* corner, int W, int H, int n, const float* d_list3, hipStream_t stream);
int cs_launch_post_track(const float* feat, int N, cs_klt_feature* dest, int* ctr, float* corner, int W, int H,
                         int doSuppress, hipStream_t stream);
int cs_launch_clear_dest(cs_klt_feature* dest, int N, hipStream_t stream);
int cs_nonmax_prepare(int d);
int cs_launch_nonmax_compact(const CsNonmaxCam* cams, int n, int W, int H, int d, int maxCand, hipStream_t stream);
int cs_launch_select_fill(const CsSelectCam* cams, int n, int maxCand, int cap, hipStream_t stream);
int cs_launch_counts_track(const cs_klt_feature* dest, int N, int* counts, int* ctr, unsigned* tagWord, hipStream_t stream);
