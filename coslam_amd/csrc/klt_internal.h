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

struct CsGainFusedArgs {
    const cs_texel* pyr0;
    const cs_texel* pyr1;
    CsTrackLevels lv;
    int W, H, fw, fh, N, hw, nIter, levelSkip;
    const float* feat0;      // X0 list (features0_tex)
    const float* featStart;  // iterate at entry (x, y; gain restarts at 1)
    float* outLast;          // result of the last pass
    float* outPrev;          // result of the pass before it (what the ping-pong schedule leaves behind)
    unsigned long long* gran;  // [passes + 1][N] {tag, beta} granules: one row per pass, never overwritten within a frame
    const unsigned* tagWord;   // frame-unique tag base (device word, bumped by the frame's last kernel)
    float sqrConvThr, ssdThr;
    float vr[4];
    float lambda, delta;
    int n1x[4], n1y[4];
    int* err;
    int pollGap;                      // s_sleep units between re-polls of the hand-off sweep
    int patchR;                       // side of the wave-private LDS patch in texels (set by the launcher)
    // fused k_post_track (v3d_gpuklt.cpp:872-888 status loop + :744-752 present scatter); dest == null: not fused
    cs_klt_feature* dest;
    int* ctr;
    float* corner;
    int doSuppress;
    unsigned long long* probe;        // diagnostic per-wave cycle counters (8 per slot) or null
};

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

int cs_launch_frame_front(const uint8_t* d_img, const CsPyrLayout& lay, cs_texel* d_pyr, int tap_mode, float* corner_out,
                          float minCornerness, float margin, int* ctr, unsigned long long* gran, int nGran,
                          hipStream_t stream);
size_t cs_nonmax_lds_bytes(int d);
int cs_launch_pyr_down_from(const CsPyrLayout& lay, cs_texel* d_pyr, int tap_mode, int first, hipStream_t stream);
int cs_launch_tail_with_next_front(const float* in, int W, int H, int d, float* out, const CsCand* cand, int maxCand, int cap,
                                   int maxKeepFixed, int* rankM, CsCand* sel, const CsFillArgs& a, const uint8_t* d_img_next,
                                   const CsPyrLayout& lay, cs_texel* d_pyr_next, int tap_mode, float* corner_next,
                                   float minCornerness, float margin, hipStream_t stream);
int cs_launch_track_nogain(const cs_texel* pyr0, const cs_texel* pyr1, const CsPyrLayout& lay, int levelSkip, int hw,
                           int nIterShader, float margin, float convThr, float ssdThr, int N, const float* featIn,
                           float* featOut, hipStream_t stream);
int cs_launch_track_gain_pass(const CsGainPassArgs& a, hipStream_t stream);
int cs_launch_reset_beta(float* feat, int N, hipStream_t stream);
int cs_launch_track_gain_fused(const CsGainFusedArgs& a, hipStream_t stream);
int cs_launch_suppress_list(float* corner, int W, int H, int n, const float* d_list3, hipStream_t stream);
int cs_launch_post_track(const float* feat, int N, cs_klt_feature* dest, int* ctr, float* corner, int W, int H,
                         int doSuppress, hipStream_t stream);
int cs_launch_clear_dest(cs_klt_feature* dest, int N, hipStream_t stream);
int cs_nonmax_prepare(int d);
int cs_launch_nonmax_compact(const float* in, int W, int H, int d, float* out, CsCand* cand, int maxCand, int* ctr,
                             hipStream_t stream);
int cs_launch_select_fill(const CsCand* cand, int maxCand, int cap, int maxKeepFixed, int* rankM, CsCand* sel,
                          const CsFillArgs& a, hipStream_t stream);
int cs_launch_counts_track(const cs_klt_feature* dest, int N, int* counts, int* ctr, unsigned* tagWord, hipStream_t stream);
