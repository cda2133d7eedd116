/*
 * coslam_hip.h -- C-ABI of libcoslam_hip.so: the MI355X (gfx950) implementation of CoSLAM's
 * per-frame hot path (pyramidal KLT tracker, intra-camera pose, robust multi-camera BA).
 *
 * Every entry point names the reference interface it replaces (paths relative to the
 * danping/CoSLAM root).  Plain C types only: pointers, sizes, ints, floats.  Functions return
 * CS_OK (0) or a negative CS_ERR_* code; cs_last_error() gives the message for the calling thread.
 *
 * Threading: one cs_klt handle is bound to one device and one HIP stream and must be driven by a
 * single caller thread at a time -- the same rule the reference has for its GL context
 * (src/gui/CoSLAMThread.cpp:48-54).  Different handles are independent.
 *
 * Host-pointer entry points mirror the reference exactly (caller-owned host image and dest[]).
 * The *_dev entry points take device pointers (image already in HBM, results left in HBM) and only
 * enqueue work on the handle's stream; they are what a multi-camera / multi-GPU driver uses.
 */
#ifndef COSLAM_HIP_H
#define COSLAM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CS_OK 0
#define CS_ERR_INVALID (-1)   /* bad argument / wrong state */
#define CS_ERR_NO_DEVICE (-2) /* no usable HIP device: there is NO CPU fallback */
#define CS_ERR_HIP (-3)       /* a HIP runtime call failed */
#define CS_ERR_ALLOC (-4)
#define CS_ERR_NUMERIC (-5) /* solver failure (LM diverged, Cholesky broke down) */

int cs_version(void);
const char* cs_last_error(void);
/* Test / diagnostic switches of the whole process (nothing in the library reads the environment): key one of
 *   "ba_syrk"     0: never the matrix-core Schur product, 2: always (parity tests force a size-dependent path onto small problems)
 *   "ba_packed"   0: the wave-per-point / workgroup-per-pair LM kernels instead of the packed ones (A/B runs, parity tests)
 *   "ba_graphs"   0: eager launches instead of the captured graph (kernel names under a profiler)
 *   "merge_print" 1: cs_register_decide_merge_dev's kernel prints its own time account
 * value -1 restores the default.  Returns CS_ERR_INVALID for an unknown key. */
int cs_debug_set(const char* key, int value);
int cs_device_count(void);

/* ------------------------------------------------------------------------------------------
 * KLT sequence tracker
 * ------------------------------------------------------------------------------------------ */

/* == V3D_GPU::KLT_TrackedFeature, src/tracking/CGKLT/v3d_gpuklt.h:166-176 (same layout, 20 bytes) */
typedef struct cs_klt_feature {
    int status; /* 0 tracked from previous frame, 1 newly created, -1 invalidated */
    float pos[2]; /* normalized [0,1] image coordinates, pixel centres at (i+0.5)/W */
    float gain;
    int fed; /* >=0: index of the externally fed point */
} cs_klt_feature;

/* == V3D_GPU::KLT_SequenceTrackerConfig, v3d_gpuklt.h:180-199 (same field order; bool -> int) */
typedef struct cs_klt_config {
    int nIterations, nLevels, levelSkip, windowWidth;
    float trackBorderMargin, convergenceThreshold, SSD_Threshold;
    int trackWithGain;
    int minDistance;
    float minCornerness, detectBorderMargin;
} cs_klt_config;

/* fills the defaults of v3d_gpuklt.h:181-191 */
void cs_klt_config_default(cs_klt_config* cfg);

typedef struct cs_klt cs_klt;

/* KLT_SequenceTracker::KLT_SequenceTracker(config), v3d_gpuklt.h:203-208.
 * device: HIP device ordinal.  tap_mode: 0 = GL-spec floor for the on-boundary decimation taps
 * (v3d_gpupyramid.cpp:407-418), 1 = geometrically centred taps (DESIGN.md "pyramid taps"). */
cs_klt* cs_klt_create(const cs_klt_config* cfg, int device, int tap_mode);
void cs_klt_destroy(cs_klt* k);

/* KLT_SequenceTracker::allocate(w,h,nLevels,fw,fh,plw,plh), v3d_gpuklt.h:213-216, v3d_gpuklt.cpp:592-624.
 * Pass plw = plh = 0 for the 5-argument overload (2*fw, 2*fh). */
int cs_klt_allocate(cs_klt* k, int width, int height, int nLevels, int featuresWidth, int featuresHeight,
                    int pointListWidth, int pointListHeight);
/* KLT_SequenceTracker::deallocate(), v3d_gpuklt.cpp:626-647 */
int cs_klt_deallocate(cs_klt* k);

/* v3d_gpuklt.h:219-240 */
int cs_klt_set_border_margin(cs_klt* k, float margin);
int cs_klt_set_convergence_threshold(cs_klt* k, float thr);
int cs_klt_set_ssd_threshold(cs_klt* k, float thr);

/* KLT_SequenceTracker::detect(image,nDetected,dest), v3d_gpuklt.cpp:692-735.  image: W*H bytes host. dest: fw*fh. */
int cs_klt_detect(cs_klt* k, const uint8_t* image, int* nDetectedFeatures, cs_klt_feature* dest);
/* KLT_SequenceTracker::detect(image,nDetected,dest,nPresent,present), v3d_gpuklt.cpp:650-691. present: nPresent*3 floats */
int cs_klt_detect_present(cs_klt* k, const uint8_t* image, int* nDetectedFeatures, cs_klt_feature* dest, int nPresent,
                          const float* present);
/* KLT_SequenceTracker::redetect(image,nNew,dest), v3d_gpuklt.cpp:737-805 */
int cs_klt_redetect(cs_klt* k, const uint8_t* image, int* nNewFeatures, cs_klt_feature* dest);
/* KLT_SequenceTracker::track(image,nPresent,dest), v3d_gpuklt.cpp:857-889 */
int cs_klt_track(cs_klt* k, const uint8_t* image, int* nPresentFeatures, cs_klt_feature* dest);
/* GPUKLT::next in two halves (asynchronous host form): cs_klt_redetect_async_h enqueues the upload, the redetect and the copy
 * of dest[] back and returns; cs_klt_fetch blocks until that frame's results are on the host.  With one handle per camera all
 * cameras' frames are in flight together -- which hides the copies but makes eight single-camera tracker launches share the chip:
 * measured 518 frames/s for 8 cameras against ~740 called one after the other (profiles/r04_dropin_cxx_latency.txt); for a rig of
 * cameras use the camera group (cs_klt_group_stage_h + cs_klt_group_redetect_dev: one launch for all).  One frame per handle may be
 * outstanding. */
int cs_klt_redetect_async_h(cs_klt* k, const unsigned char* image);
int cs_klt_fetch(cs_klt* k, int* nNew, cs_klt_feature* dest);
/* KLT_SequenceTracker::feedExternFeaturePoints(npts,featPts,trackIds,nFed), v3d_gpuklt.cpp:808-855.
 * featPts: npts*3 floats (stride 3 as GPUKLT.cpp:165-171 passes it); trackIds: npts ints. */
int cs_klt_feed(cs_klt* k, int npts, const float* featPts, int* trackIds, int* nFed);
/* ... with the point list, the slots and the count in device memory, asynchronous on the tracker's stream (one launch: the
 * distance test per slot, then the free slots ranked in slot order).  cs_klt_feed is this plus the copies. */
int cs_klt_feed_dev(cs_klt* k, int npts, const float* d_featPts, int* d_trackIds, int* d_nFed);
/* KLT_SequenceTracker::advanceFrame(), v3d_gpuklt.h:252-259 */
int cs_klt_advance(cs_klt* k);

/* ---- device-resident variants: enqueue on the handle's stream, no host synchronisation ----
 * d_image: W*H bytes in HBM.  d_dest: fw*fh cs_klt_feature in HBM (caller-owned).
 * d_counts: 4 ints in HBM: [0] = the count the host variant returns, [1] = tracked (status 0),
 * [2] = detector survivors before selection, [3] = reserved. */
int cs_klt_set_stream(cs_klt* k, void* hip_stream); /* NULL = the handle's own stream */
int cs_klt_detect_dev(cs_klt* k, const void* d_image, void* d_dest, void* d_counts);
int cs_klt_redetect_dev(cs_klt* k, const void* d_image, void* d_dest, void* d_counts);
int cs_klt_track_dev(cs_klt* k, const void* d_image, void* d_dest, void* d_counts);
/* Frame-front prefetch: call BEFORE the cs_klt_redetect_dev / cs_klt_detect_dev of the current frame with the NEXT
 * frame's image (device pointer, must stay valid and unchanged until consumed).  The current frame's detector tail
 * then builds the next pyramid + cornerness map in the same two launches (horizontal fusion: the tail leaves the chip
 * mostly idle), and the following cs_klt_redetect_dev / cs_klt_track_dev called with that pointer starts at the
 * tracker.  Results are identical with and without; a request is dropped by a track-only frame, by hipGraph replay
 * and by any call with a different image.  (The reference builds the pyramid inside track()/redetect(),
 * v3d_gpuklt.cpp:737-745,856-860.) */
int cs_klt_prefetch_dev(cs_klt* k, const void* d_image_next);
int cs_klt_synchronize(cs_klt* k);
/* Replay the *_dev frame schedules from cached hipGraphs (one host launch per frame instead of ~60).  The image is
 * first copied into the handle's staging buffer (device-to-device) so that one graph serves every frame. */
int cs_klt_enable_graphs(cs_klt* k, int on);
/* Gain tracker schedule: 1 (default) = one persistent launch for all levels x iterations (and all cameras of a group),
 * neighbours exchange gains through 8-byte {tag, beta} granules; 0 = one launch per Gauss-Newton pass as the reference
 * schedules its shader (v3d_gpuklt.cpp:254-287).  Both give bit-identical results.  The persistent schedule needs every
 * wave co-resident (checked against the occupancy the runtime reports); otherwise the per-pass schedule runs.
 * Env COSLAM_KLT_FUSED=0 sets the default to 0. */
int cs_klt_set_fused(cs_klt* k, int on);
/* compute units available to the handle's stream when it carries a CU mask (co-residency budget of the persistent tracker) */
int cs_klt_set_cu_count(cs_klt* k, int n_cus);
/* Persistent tracker placement: 1 (default) = the workgroups of a camera are numbered so that they land on that camera's own XCD(s)
 * (workgroup b runs on XCD b % 8 on MI355X): its pyramids are fetched into one L2 instead of eight -- FETCH_SIZE 25.8 MB per
 * 8-camera launch instead of 52.7 (profiles/r05_tracker_pmc.json).  Applies when the launch carries 1, 2, 4 or 8 cameras and a
 * camera's share of workgroups fits its XCDs (two 7 x 7 workgroups per CU: 64 per XCD); results are bit-identical either way.  The first camera of a group decides. */
int cs_klt_set_xcd_placement(cs_klt* k, int on);
/* how many handles of this device may have their persistent tracker in flight at the same time (0 = every live handle) */
int cs_klt_set_concurrent_handles(cs_klt* k, int n);
/* a stream confined to the CU-mask bits [first_cu, first_cu + n_cus) (hipExtStreamCreateWithCUMask): keeps BA kernels
 * off the SIMDs of the lock-stepped persistent tracker; returns a hipStream_t (null on error).  On MI355X mask bit i is
 * a CU of XCC (i % 8) -- shader engine (i / 8) % 4, CU (i / 8) / 4 -- so a range is "the same few CUs of every XCD";
 * use sizes that are multiples of 32 (one CU per shader engine per XCD each). */
void* cs_stream_create_cu_range(int device, int first_cu, int n_cus);
/* mask bits with (index % period) < take, or (complement != 0) all the others.  NOTE: an XCC whose share of the mask
 * is empty gets all its CUs, so period = 8 (or 4, 2) masks do not partition the chip at all; kept for experiments. */
void* cs_stream_create_cu_interleaved(int device, int period, int take, int complement);
int cs_stream_destroy(void* stream);
/* diagnostic only: per-wave cycle counters (8 x uint64 per wave of 8 slots; host_out8 holds 8 * N) of the persistent gain
 * tracker; see klt_seq.hip */
int cs_klt_debug_probe(cs_klt* k, int on, unsigned long long* host_out8);
/* HIP-event timing of the tracker stage on the handle's stream (used by bench.py's roofline leg).  While on, the
 * *_dev calls launch eagerly and bracket the tracker kernel(s) of every frame with an event pair. */
int cs_klt_set_profiling(cs_klt* k, int on);
int cs_klt_get_profile(cs_klt* k, double* tracker_us_total, int* n_frames, int* launches_per_frame);

/* ---- camera groups: the per-frame call of SEVERAL cameras in one set of launches ----
 * CoSLAM::featureTracking() (src/app/SL_CoSLAM.cpp:299-305) calls GPUKLT::next -> KLT_SequenceTracker::redetect
 * camera by camera.  A group issues the same per-camera work with the camera as one more grid dimension of every
 * kernel: 3-5 launches per frame for all cameras together, the gain tracker of all cameras in ONE persistent launch.
 * Results are bit-identical to driving the handles one by one.  The handles must live on one device and share image
 * size, slot grid and configuration (cs_klt_group_create returns NULL otherwise); they stay owned by the caller, and
 * every array argument has one entry per handle, in the order given at creation. */
typedef struct cs_klt_group cs_klt_group;
cs_klt_group* cs_klt_group_create(cs_klt* const* handles, int n); /* n <= 16 */
void cs_klt_group_destroy(cs_klt_group* g);                        /* the handles are NOT destroyed */
int cs_klt_group_size(const cs_klt_group* g);
int cs_klt_group_set_stream(cs_klt_group* g, void* hip_stream); /* also rebinds every member handle */
int cs_klt_group_detect_dev(cs_klt_group* g, const void* const* d_images, void* const* d_dests, void* const* d_counts);
int cs_klt_group_redetect_dev(cs_klt_group* g, const void* const* d_images, void* const* d_dests, void* const* d_counts);
int cs_klt_group_track_dev(cs_klt_group* g, const void* const* d_images, void* const* d_dests, void* const* d_counts);
int cs_klt_group_prefetch_dev(cs_klt_group* g, const void* const* d_images_next);
/* Host images into the device OFF the frame's critical path (GPUKLT::next uploads, then tracks: reference
 * src/tracking/GPUKLT.cpp:144-161).  cs_klt_group_stage_h copies the n host images of a FUTURE frame (W*H bytes each; pinned
 * memory -- cs_pinned_alloc -- is PULLED by a copy kernel: a kernel launch for the caller; hipMemcpyAsync blocks the calling
 * thread for 160-280 us per 2.4 MB here; pageable memory falls back to it) into the next slot of a ring of 3 and returns
 * the slot; the pull is enqueued on the group's stream, ahead of every later use of the slot.  cs_klt_group_staged returns the
 * slot's device images, which then go to cs_klt_group_prefetch_dev / _redetect_dev like any device image.
 * Stage frame f+2 while f is tracked and f+1 is prefetched.  Images written back to back (one ring entry per frame) go in
 * one copy.  LIFETIME: the call only ENQUEUES the pull -- the host images must stay valid and unchanged until the group's
 * stream has passed it (e.g. until the frame that uses the slot has been fetched, or cs_klt_group_synchronize); a slot is
 * reused three stage calls later, by which time the frames that read it must have been enqueued (they are, in stream order). */
int cs_klt_group_stage_h(cs_klt_group* g, const unsigned char* const* h_images, int* slot);
int cs_klt_group_staged(cs_klt_group* g, int slot, const void** d_images);
void* cs_pinned_alloc(size_t bytes);
void cs_pinned_free(void* p);
int cs_klt_group_advance(cs_klt_group* g);     /* advanceFrame() on every handle */
int cs_klt_group_synchronize(cs_klt_group* g); /* + the persistent tracker's error words */

/* ---- introspection used by the parity tests (host copies; synchronise the stream) ----
 * which: 0 = _pyrCreator0 (previous frame), 1 = _pyrCreator1 (frame most recently built). */
size_t cs_klt_pyramid_texels(const cs_klt* k); /* texels of 4 halfs, all levels */
int cs_klt_pyramid_level_offset(const cs_klt* k, int level, int64_t* off_texels, int* w, int* h);
int cs_klt_read_pyramid(cs_klt* k, int which, uint16_t* host_out);
int cs_klt_read_cornerness(cs_klt* k, float* host_out); /* W*H floats after non-max suppression */
int cs_klt_read_features(cs_klt* k, float* host_out3);  /* what readFeatures() returns, fw*fh*3 */
/* standalone kernels, for unit parity */
int cs_klt_build_pyramid(cs_klt* k, const uint8_t* image); /* builds into _pyrCreator1 */

/* ------------------------------------------------------------------------------------------
 * Intra-camera pose: robust 6-DoF pose of one camera from 3D-2D correspondences
 * ------------------------------------------------------------------------------------------ */

/* == class IntraCamPoseOption, src/slam/SL_IntraCamPose.h:19-57 (inputs and outputs, same names) */
typedef struct cs_pose_option {
    int maxIterLM, maxIterRW;                                    /* 100, 5 */
    double epsErrorChangeLM, epsParamChangeLM, epsErrorChangeRW; /* 1e-7, 1e-6, 1e-6 */
    int verboseLM, verboseRW;                                    /* ignored on input; on return verboseRW = LM steps taken over all
                                                                    re-weighting rounds (a diagnostic the reference does not have) */
    double lambda0, lambda;                                      /* 1e-3 in; both updated as the reference does */
    double err0, err, errRW;
    int retTypeLM, npts, nIterLM, nIterRW;
} cs_pose_option;

void cs_pose_option_default(cs_pose_option* opt); /* IntraCamPoseOption(), SL_IntraCamPose.h:42-46 */

/* bool intraCamEstimate(K,R0,t0,npts,prevErrs,Ms,ms,tau,R_opt,t_opt,opt), src/slam/SL_IntraCamPose.h:92-95,
 * SL_IntraCamPose.cpp:626-709.  Host pointers, row-major doubles, prevErrs may be NULL.
 * Returns 1 (true), 0 (false: LM failed, as the reference) or a negative CS_ERR_* code. */
int cs_pose_intracam(const double K[9], const double R0[9], const double t0[3], int npts, const double* prevErrs,
                     const double* Ms, const double* ms, double tau, double R_opt[9], double t_opt[3],
                     cs_pose_option* opt, int device);

/* Batched, device-resident form: nProb independent cameras in one launch (one workgroup each), enqueued on
 * hip_stream.  All pointers are device pointers.  Problem p reads K[9p..], R0[9p..], t0[3p..], npts[p],
 * Ms[3*ptsStride*p ..], ms[2*ptsStride*p ..], prevErrs (NULL or ptsStride*p ..) and writes R_opt[9p..],
 * t_opt[3p..], opt[p] (opt[p] must be initialised, e.g. with cs_pose_option_default), ok[p] in {0,1}. */
int cs_pose_intracam_batch_dev(int device, void* hip_stream, int nProb, int ptsStride, const double* K,
                               const double* R0, const double* t0, const int* npts, const double* prevErrs,
                               const double* Ms, const double* ms, double tau, double* R_opt, double* t_opt,
                               cs_pose_option* opt, int* ok);

/* ------------------------------------------------------------------------------------------
 * Hand-back of the tracker's output to the pose stage, on the device (every camera of a group in one launch)
 * ------------------------------------------------------------------------------------------
 * Replaces GPUKLT::addToFeaturePoints (src/tracking/GPUKLT.cpp:36-60: x W,H -> undistorPoint -> drop when out >= W | H
 * -> FeaturePoints::add + Track2D add / clear), SingleSLAM::chooseStaticFeatPts (src/app/SL_SingleSLAM.cpp:345-397: one
 * track per (W / nColBlk) x (H / nRowBlk) block, a mapped one first, else the longest) and the Ms / ms packing of
 * SingleSLAM::poseUpdate3D (:620-640).  The reference allocates one FeaturePoint per feature per frame on the host
 * (src/slam/SL_FeaturePoints.cpp:81-87); here dest[] stays in HBM and the outputs are structure-of-arrays records
 * plus the packed correspondences cs_pose_intracam_batch_dev consumes.  All pointers are DEVICE pointers.
 * undistorPoint itself is external (LibVisualSLAM); our definition: normalise with K, scale by
 * 1 + sum_i k_ud[i] r^(2(i+1)), i = 0..6, map back with K (k_ud = 0: identity). */
typedef struct cs_handback_cam {
    const cs_klt_feature* dest; /* N: what cs_klt_*_dev wrote for this frame */
    const double* K;            /* 9, row-major */
    const double* kud;          /* 7 (GPUKLT::m_kud, src/tracking/GPUKLT.h:44-47) */
    const double* mapPts;       /* P x 3: MapPoint::M of the point a slot is associated with */
    const unsigned char* isStatic; /* N or NULL: slots classified static without a map point (FeaturePoint::type) */
    int* slot2map;              /* N in/out: map point index of the slot's track or -1; new / dead slots are reset to -1 */
    int* trackSpan;             /* 2N in/out: first[N] then last[N] frame of the slot's Track2D (-1 = empty); its length()
                                   is last - first + 1 (src/tracking/SL_Track2D.h:63-65) */
    double* xy;                 /* 2N in/out: undistorted pixel, x[N] then y[N]; kept for a slot dropped by the >= W|H rule */
    int* state;                 /* N out: 0 tracked, 1 new, -1 dead, -2 dropped by the out >= W | H rule */
    int* selBlk;                /* nColBlk * nRowBlk out or NULL: chosen slot of every block (-1: none), featPts order */
    double* Ms;                 /* ptsStride x 3 out */
    double* ms;                 /* ptsStride x 2 out */
    int* sel;                   /* ptsStride out: slot of every packed correspondence */
    int* npts;                  /* 1 out */
    cs_pose_option* opt;        /* 1 out or NULL: reset to IntraCamPoseOption() for the pose solve that follows */
    int* pointFeat;             /* out or NULL: for map points 0 .. nPointFeat-1 the slot of this camera's feature of THIS
                                   frame attached to the point (MapPoint::pFeatures[iCam] with ->f == curFrame), else -1;
                                   entry p is pointFeat[p * pointFeatStride] -- a column of the P x nCams table
                                   cs_register_search_dev reads */
    int pointFeatStride, nPointFeat;
} cs_handback_cam;
/* cams: HOST array of nCams (<= 16) records of device pointers.  nColBlk x nRowBlk = 16 x 12 in CoSLAM
 * (src/app/SL_SingleSLAM.h:36-37); ptsStride >= the most correspondences wanted per camera (192). */
int cs_klt_handback_dev(int device, void* hip_stream, int nCams, const cs_handback_cam* cams, int N, int W, int H,
                        int nColBlk, int nRowBlk, int ptsStride, int frame /* GPUKLT::m_frame of this call, >= 0 */);

/* ---- SURVEY 8f-1, second half: the group's frames as HOST records, for a lazy FeaturePoints / Track2D adaptor ----
 * GPUKLT::next (src/tracking/GPUKLT.cpp:144-161) = redetect + addToFeaturePoints + advanceFrame per camera; addToFeaturePoints
 * (:36-60) heap-allocates one FeaturePoint and one Track2DNode per feature per frame (src/slam/SL_FeaturePoints.cpp:81-87,
 * src/tracking/SL_Track2D.h:79-82).  A host view runs the frame for ALL cameras on the device -- cs_klt_group_* + cs_klt_handback_dev
 * -- and streams what addToFeaturePoints would have appended, per slot {state (0 tracked, 1 new, -1 dead, -2 dropped by the
 * out >= W | H rule), undistorted x, y}, into a ring of `depth` frames in pinned host memory (stores from a kernel; nothing on the
 * host waits).  include/shim/tracking/GPUKLTGroup.h replays the frames a consumer has not seen into the reference's lists when it
 * asks.  handles: the cameras' trackers (KLT_SequenceTracker::handle()), allocated, same size / grid / configuration; K [nCams][9],
 * kud [nCams][7] (GPUKLT::m_K, m_kud) are HOST arrays.  The view owns its stream. */
typedef struct cs_klt_hostview cs_klt_hostview;
cs_klt_hostview* cs_klt_hostview_create(int device, cs_klt* const* handles, int nCams, int W, int H, int N, const double* K,
                                        const double* kud, int depth /* frames the ring holds, >= 2 */);
void cs_klt_hostview_destroy(cs_klt_hostview* v); /* the handles are NOT destroyed */
/* W*H bytes of pinned memory for the NEXT frame's image of camera `cam` (decode straight into it: no host copy).  Blocks only
 * while the device still has to pull the frame that used this buffer 4 frames ago. */
unsigned char* cs_klt_hostview_image(cs_klt_hostview* v, int cam);
/* One frame for all cameras, enqueued: images (h_images NULL: already in cs_klt_hostview_image(); else nCams host pointers that are
 * COPIED there, so the caller may reuse its buffers at once), detect (first != 0: GPUKLT::first, :113-121) or redetect (GPUKLT::next),
 * advanceFrame, hand-back with GPUKLT::m_frame = frame, the records into ring slot frame % depth.  Frames must increase. */
int cs_klt_hostview_frame(cs_klt_hostview* v, const unsigned char* const* h_images, int frame, int first);
/* Blocks until `frame`'s records are on the host: state [nCams][N], xy [nCams][2N] (x[N] then y[N]) inside the ring -- valid until
 * `depth` more frames have been enqueued.  CS_ERR_INVALID when the frame has been overwritten. */
int cs_klt_hostview_fetch(cs_klt_hostview* v, int frame, const int** state, const double** xy);
int cs_klt_hostview_oldest(const cs_klt_hostview* v); /* oldest frame still in the ring, -1: none */
int cs_klt_hostview_newest(const cs_klt_hostview* v);
int cs_klt_hostview_synchronize(cs_klt_hostview* v);

/* ------------------------------------------------------------------------------------------
 * Map-point registration: the search step for all map points x all cameras in one launch
 * ------------------------------------------------------------------------------------------
 * Replaces, inside CoSLAM::curStaticPointRegInGroup (src/app/SL_CoSLAM.cpp:731-757), curDynamicPointRegInGroup (:955-980)
 * and activeMapPointRegisterInGroup (:1118-1145), the statements they share per (map point, camera): isAtCameraBack,
 * project, the image test, getProjectionCovMat, searchMahaNearestFeatPt (src/app/SL_SingleSLAM.cpp:1141-1164), plus the
 * candidate's own term of staticCheckMergability (:714-729).  A caller runs it once per frame before its loop over the
 * points and reads the candidates from the tables; what happens to a candidate afterwards stays in the caller.
 * The three loops differ only in the scalars:   sigmaSearch  maxDist          sigmaMerge
 *     current static points                     pixelErrVar  3 pixelErrVar    pixelErrVar
 *     current dynamic points                    pixelErrVar  4 pixelErrVar    pixelErrVar
 *     active points                             2.5 pixelErrVar  3 pixelErrVar  pixelErrVar
 * Features of a camera = the hand-back records of this frame (cs_klt_handback_dev): slots with state 0 or 1, in slot
 * order (the order GPUKLT::addToFeaturePoints appends them to FeaturePoints). */
typedef struct cs_register_cam {
    const double* K;                /* 9: slam[iCam].K */
    const double* R;                /* 9: m_camPos.current()->R */
    const double* t;                /* 3 */
    const double* xy;               /* 2N: undistorted pixel, x[N] then y[N] (FeaturePoint::m) */
    const int* state;               /* N: 0 tracked, 1 new; anything else = not in this frame's list */
    const int* slot2map;            /* N: FeaturePoint::mpt as an index, < 0 = none */
    const unsigned char* isDynamic; /* N or NULL: FeaturePoint::type == TYPE_FEATPOINT_DYNAMIC */
    const unsigned char* isStatic;  /* N or NULL, read when isDynamic is NULL: the complement, as the pose update keeps it
                                       (cs_poseupdate_cam::isStatic: 0 = TYPE_FEATPOINT_DYNAMIC) */
} cs_register_cam;
/* Outputs are P x nCams tables (point-major).  slot: >= 0 the nearest feature's slot; -1 pointFeat says the point already
 * has a feature of this frame in the camera; -2 behind the camera; -3 projects outside [0,W) x [0,H); -4 the camera has
 * no feature this frame.  m (x2): the projection; var (x4): its covariance; dist: the winner's Mahalanobis distance
 * scaled by 1 / maxDist (the reference applies NO threshold to it); flags: bit 0 the candidate has no map point, bit 1
 * it is dynamic (a property of the FEATURE, cs_register_cam::isDynamic: the same for every point that finds it -- the registration
 * decision relies on that), bit 2 it passes the candidate's own mergability term (distance under var(sigmaMerge) <= 1).
 * pointFeat (P x nCams): slot of p->pFeatures[iCam] when that feature belongs to the current frame, else -1.
 * cams: HOST array of nCams (<= 16) records; in the _dev form their members and every d_* argument are DEVICE pointers,
 * in the host form everything is host memory (one upload, one launch, one read-back). */
/* One pass = one of the registration loops' searches (its map points, its three scalars, its output tables); the passes of a
 * frame -- active points, current static points: CoSLAMThread.cpp:108-118 -- can share ONE launch. */
typedef struct cs_register_pass {
    int P;
    double sigmaSearch, maxDist, sigmaMerge;
    const double* M;      /* P x 3 */
    const double* cov;    /* P x 9 */
    const int* pointFeat; /* P x nCams */
    int* slot;            /* P x nCams out, like cs_register_search_dev's tables */
    double* m;
    double* var;
    double* dist;
    int* flags;
    /* optional (NULL / 0: every point uses maxDist): the pass's points' CS_MAP_* bytes [P] and the scale of the certainly DYNAMIC ones --
     * curDynamicPointRegInGroup searches with pixelErrVar * 4 where the static loop uses * 3 (src/app/SL_CoSLAM.cpp:973, :754), so ONE pass
     * over the current points serves both registrations */
    const unsigned char* mapFlags;
    double maxDistDynamic;
    /* optional (NULL: the pass's points are map rows 0 .. P - 1): the pass's points as a COMPACT list of up to P map indices (the first
     * entry < 0 ends it) -- every table above stays indexed by the map index (M, cov, pointFeat, slot, ... are then whole-map tables),
     * only the rows of listed points are read and written.  cs_register_list_current_dev builds the list of a frame's current points. */
    const int* list;
} cs_register_pass;
/* The frame's CURRENT map points -- what CoSLAM::currentMapPointsRegister walks: curMapPts = the points with a feature of this frame in
 * at least one camera (mapStateUpdate, src/app/SL_CoSLAM.cpp:1176-1194; :734 / :958 ask numVisCam > 0 once more), wherever they sit in
 * the map, the points genNewMapPoints has just appended included -- as a compact list in map order: d_list [nMap] (entries behind the
 * list: -1), d_listCount [1] or NULL.  A point is listed when its index is below *d_mapCount (NULL: nMap), it is not CS_MAP_FALSE
 * (d_mapFlags, may be NULL) and d_pointFeat [nMap][nCams] holds a feature of it.  d_list / d_listCount are IN / OUT: they still hold
 * the previous call's list (start: count 0), and the rows of d_slotTable (nMap x nCams ints, or NULL; start: -1 everywhere) of the
 * points that have LEFT the list since are set to -1 -- a search driven by the list writes listed rows only, so no stale candidate
 * survives a frame and no frame clears the whole table. */
int cs_register_list_current_dev(int device, void* hip_stream, int nCams, int nMap, const int* d_mapCount, const int* d_pointFeat,
                                 const unsigned char* d_mapFlags, int* d_list, int* d_listCount, int* d_slotTable);
/* the same with a CAP on the list: the passes a frame loop runs behind the list (search, running mergability, the merge walk, the
 * candidate records between ranks) are sized for listCap rows.  The first listCap current points in map order are listed; every
 * current point beyond them is left out for this frame -- its row of d_slotTable is cleared, so the decision (which visits every point
 * that has a feature) finds no candidate for it rather than an older frame's -- and counted into *d_overflow (accumulating; or NULL). */
int cs_register_list_current_cap_dev(int device, void* hip_stream, int nCams, int nMap, const int* d_mapCount, const int* d_pointFeat,
                                     const unsigned char* d_mapFlags, int* d_list, int* d_listCount, int* d_slotTable, int listCap,
                                     int* d_overflow);
int cs_register_search_passes_dev(int device, void* hip_stream, int nCams, const cs_register_cam* cams, int N, int W, int H,
                                  int nPass /* 1 or 2 */, const cs_register_pass* passes /* host array */);
/* the same for cameras cam0 .. cam0 + nCamsRun - 1 only: their columns of the nCams-wide tables (with the cameras sharded over
 * several GPUs every rank holds all cameras' records and searches for the cameras it owns) */
int cs_register_search_passes_range_dev(int device, void* hip_stream, int nCams, int cam0, int nCamsRun, const cs_register_cam* cams,
                                        int N, int W, int H, int nPass, const cs_register_pass* passes);
int cs_register_search_dev(int device, void* hip_stream, int nCams, const cs_register_cam* cams, int N, int W, int H, int P,
                           const double* d_M, const double* d_cov, const int* d_pointFeat, double sigmaSearch, double maxDist,
                           double sigmaMerge, int* d_slot, double* d_m, double* d_var, double* d_dist, int* d_flags);
int cs_register_search(int device, int nCams, const cs_register_cam* cams, int N, int W, int H, int P, const double* M,
                       const double* cov, const int* pointFeat, double sigmaSearch, double maxDist, double sigmaMerge, int* slot,
                       double* m, double* var, double* dist, int* flags);

/* The DECISION behind the current-static search -- CoSLAM::curStaticPointsRegInGroup / curStaticPointRegInGroup with bMerge == false
 * (src/app/SL_CoSLAM.cpp:854-898, 731-830) -- for all points and cameras in one launch: for every camera o in order, the certainly
 * static points with a feature of this frame in o, in map order, each walking the cameras in order: a camera where the point already
 * has a feature of this frame, where the search found nothing or found a DYNAMIC feature is passed by; the nearest feature is
 * attached when it is unmapped and mergeable over its whole track (d_mergeable == 1: cs_register_mergability_dev; compareFeaturePt
 * returns true whatever it computes, :546-558); a feature that already carries a map point -- before the pass, or taken by an earlier
 * walk -- ends the point's walk (:789-790).  The sequential "first claimant wins" is a recursion along the order of the walk steps;
 * nSweeps Jacobi sweeps (one small launch each) solve it -- exactly once two consecutive sweeps agree (d_counts[3] = 1), which takes as
 * many sweeps as the longest chain of walks cutting each other short: 1-2 on tracked frames (csrc/register.hip).  nSweeps = 0: ONE
 * launch whose workgroups sweep behind a grid barrier until a sweep changes nothing (at most 64 sweeps; d_counts[2] = the sweeps it took)
 * -- what a frame loop should pass: the answer is then always the sequential one.
 * d_slot / d_flags: the search's P x nCams tables; d_mapFlags [P]: CS_MAP_* bytes of the pass's points (map points mapBase ..
 * mapBase + P - 1); IN / OUT: d_pointFeat [P][nCams] (MapPoint::addFeature) and every camera's slot2map [N] (the attached feature's
 * whole track takes the point, :771-775); OUT: d_attached [P][nCams], d_regged [P] (refineMapPoint is due: hand it to
 * cs_refine_map_points_dev as d_select), d_counts [4] or NULL (features attached, points regged, sweeps, converged).
 * d_scratch: cs_register_decide_scratch_bytes; its LAST int COUNTS the calls whose sweeps did not settle (the decision then is not
 * the sequential one), the int before it how many of those ended on a grid-barrier time-out of the self-settling launch (its workgroups were
 * not co-resident within 20 ms: nothing was attached that frame); never cleared here -- zero the scratch once, read the words at the end of a run.  Not done here: the projections are those of the search as it ran (the reference
 * refines a point before the next camera's round of walks, :889-893), and the bMerge == true branch (every 50th frame: checkUnify on
 * a conflict) -- that one is cs_register_decide_merge_dev further down. */
size_t cs_register_decide_scratch_bytes(int nCams, int N, int P);
int cs_register_decide_static_dev(int device, void* hip_stream, int nCams, int N, int P, int mapBase, const int* d_slot, const int* d_flags,
                                  const unsigned char* d_mergeable, const unsigned char* d_mapFlags, int* d_pointFeat,
                                  int* const* d_slot2map /* host array of nCams device pointers */, unsigned char* d_attached,
                                  unsigned char* d_regged, void* d_scratch, int nSweeps, int* d_counts);
/* ONE camera's loop of curStaticPointsRegInGroup (:864-893): only the points that hold a feature of this frame in camera onlyCam walk
 * (map order); onlyCam < 0: the call above.  With it the reference's run is reproduced step for step: for onlyCam = 0 .. nCams - 1 --
 * search (cs_register_search_dev on the points as they stand), cs_register_mergability_dev, this call, cs_refine_map_points_dev of
 * d_regged -- so that a point refined in one camera's loop is projected from its new position in the next (the single call above takes
 * every decision against ONE search).  nCams times the launches: a parity mode, tests/test_register_decide_gpu.py pins it
 * to the reference's own run (tests/golden/decide_golden.npz) bit for bit. */
int cs_register_decide_static_cam_dev(int device, void* hip_stream, int nCams, int N, int P, int mapBase, const int* d_slot, const int* d_flags,
                                      const unsigned char* d_mergeable, const unsigned char* d_mapFlags, int* d_pointFeat,
                                      int* const* d_slot2map, unsigned char* d_attached, unsigned char* d_regged, void* d_scratch, int nSweeps,
                                      int* d_counts, int onlyCam);
/* The same with the point kinds spelled out -- kinds bit 0: the certainly static points (curStaticPointsRegInGroup: what the two calls above
 * do), bit 1: the certainly DYNAMIC points (curDynamicPointsRegInGroup, :904-1020: the same walk over DYNAMIC features only -- a static
 * candidate is passed by, a dynamic one that carries a point ends the walk -- which CoSLAM::currentMapPointsRegister runs behind the static
 * loops, :834-853).  The two kinds never meet at a feature, so kinds = 3 registers both in ONE call with the result of one after the other
 * (the search pass must then have searched the dynamic points with their own scale: cs_register_pass::mapFlags / maxDistDynamic). */
int cs_register_decide_kinds_dev(int device, void* hip_stream, int nCams, int N, int P, int mapBase, const int* d_slot, const int* d_flags,
                                 const unsigned char* d_mergeable, const unsigned char* d_mapFlags, int* d_pointFeat, int* const* d_slot2map,
                                 unsigned char* d_attached, unsigned char* d_regged, void* d_scratch, int nSweeps, int* d_counts, int onlyCam,
                                 int kinds);

/* The SECOND VISITS of the reference's order (src/app/SL_CoSLAM.cpp:864-869, :889-893): a point that registers a feature in camera a's
 * loop is refined at the end of that loop and visited AGAIN in the loop of the next camera in which it now holds a feature of this frame.
 * (A later visit of a point that did not register changes nothing, so cs_register_decide_kinds_dev's single pass IS the reference's run
 * up to these visits.)  Played in rounds behind the single pass + its refine:
 *   cs_register_revisit_list_dev    the points that registered in the previous round (d_regIn [P]: the single pass's d_regged with
 *                                   firstRound = 1 / keepIn = 1, later the previous round's d_regOut) and have a later loop: d_list [cap],
 *                                   padded with -1; d_visitLoop / d_nextLoop [P] ints keep the loops; d_counts [4]: listed, beyond cap
 *   -- then the caller's search (cs_register_pass.list = d_list, P = cap) and mergability pass over those rows at their refined positions --
 *   cs_register_revisit_decide_dev  the listed points' walks in their next loop, ordered among themselves like all walks; attaches, sets
 *                                   d_regOut; d_decideScratch = the single pass's scratch (its owner arrays are reused); d_curList /
 *                                   d_curCount: the frame's current points.  d_counts [4], accumulating: features attached, points
 *                                   registered, CONFLICTS (this visit met a feature that a later-ordered visit of the frame had already
 *                                   taken, or took one that such a visit had walked past before attaching elsewhere: the reference's order
 *                                   would have ended differently -- counted, not repaired), rounds whose sweeps did not settle
 *   -- then the caller's refine of d_regOut's points, and the next round.
 * cap <= 1024 rows; one workgroup each. */
int cs_register_revisit_list_dev(int device, void* hip_stream, int nCams, int P, int cap, int firstRound, const int* d_pointFeat,
                                 const unsigned char* d_attached, unsigned char* d_regIn, int keepIn, unsigned char* d_regOutClear /* [P] or NULL: the
                                 array this round's cs_register_revisit_decide_dev will mark, zeroed here; not d_regIn */,
                                 int* d_visitLoop, int* d_nextLoop, int* d_list, int* d_counts);
int cs_register_revisit_decide_dev(int device, void* hip_stream, int nCams, int N, int P, int cap, int mapBase, int kinds, const int* d_list,
                                   const int* d_nextLoop, int* d_visitLoop, const int* d_slot, const int* d_flags, const unsigned char* d_mergeable,
                                   const unsigned char* d_mapFlags, int* d_pointFeat, int* const* d_slot2map, unsigned char* d_attached,
                                   unsigned char* d_regOut, void* d_decideScratch, const int* d_curList, const int* d_curCount, int curCap,
                                   int* d_counts, const int* d_listCount /* cs_register_revisit_list_dev's d_counts (its [0]: rows listed) or
                                   NULL: an empty list ends the launch at once */);

/* The same rounds with TWO launches fewer each: the lists are built by the walks themselves.
 *   cs_register_decide_kinds_rounds_dev   cs_register_decide_kinds_dev (nSweeps 0, all cameras' loops) whose prepare launch clears
 *       d_rvLists [nRounds][rvCap] to -1 and d_rvCounts [nRounds] to 0 and whose walks append the points that registered and hold a
 *       feature in a later loop to list 0 (d_visitLoop / d_nextLoop [P] written as cs_register_revisit_list_dev(firstRound) writes them;
 *       d_rvCounts[0] = points appended, beyond rvCap they are not visited again and counted in d_rvCounts[nRounds], which is NOT cleared:
 *       the array has nRounds + 1 entries);
 *   cs_register_revisit_decide_next_dev   cs_register_revisit_decide_dev whose walks append the points that registered again to the NEXT
 *       round's list (d_nextList = d_rvLists + (r + 1) * rvCap, d_nextCount = d_rvCounts + r + 1; NULL in the last round;
 *       d_overflow = d_rvCounts + nRounds or NULL).
 * The marks of a round (d_regOut) are consumed by cs_feat_ref_advance_refine_dev(clearSelect), so nothing has to clear them.
 * A list's order is the order of the appends (the decisions do not depend on it: every visit carries its own order key). */
int cs_register_decide_kinds_rounds_dev(int device, void* hip_stream, int nCams, int N, int P, int mapBase, const int* d_slot, const int* d_flags,
                                        const unsigned char* d_mergeable, const unsigned char* d_mapFlags, int* d_pointFeat, int* const* d_slot2map,
                                        unsigned char* d_attached, unsigned char* d_regged, void* d_scratch, int nSweeps, int* d_counts, int onlyCam,
                                        int kinds, int* d_rvLists, int rvCap, int nRounds, int* d_rvCounts, int* d_visitLoop, int* d_nextLoop);
int cs_register_revisit_decide_next_dev(int device, void* hip_stream, int nCams, int N, int P, int cap, int mapBase, int kinds, const int* d_list,
                                        int* d_nextLoop, int* d_visitLoop, const int* d_slot, const int* d_flags, const unsigned char* d_mergeable,
                                        const unsigned char* d_mapFlags, int* d_pointFeat, int* const* d_slot2map, unsigned char* d_attached,
                                        unsigned char* d_regOut, void* d_decideScratch, const int* d_curList, const int* d_curCount, int curCap,
                                        int* d_counts, const int* d_listCount, int* d_nextList, int* d_nextCount, int* d_overflow);

/* Cameras sharded over GPUs: a rank searches for its own cameras (cs_register_search_passes_range_dev, cs_register_mergability_range_dev);
 * the decision needs every camera's candidates.  pack: columns cam0 .. cam0 + nOwn - 1 of the P x nCams tables into a send record of
 * 3 * nOwn * P ints; unpack: the records of all ranks (cs_comm_allgather_dev: rank r owns cameras r * nOwn ..) into the tables
 * (skipRank: that rank's columns are left alone, -1 = none). */
int cs_register_candidates_pack_dev(int device, void* hip_stream, int P, int nCams, int cam0, int nOwn, const int* d_slot, const int* d_flags,
                                    const unsigned char* d_mergeable, int* d_send);
int cs_register_candidates_unpack_dev(int device, void* hip_stream, int P, int nCams, int nOwn, int skipRank, const int* d_recv, int* d_slot,
                                      int* d_flags, unsigned char* d_mergeable);
/* the same for tables indexed by the MAP index and a list of the rows that matter (cs_register_list_current_dev; identical on every
 * rank): row j of a record = map point d_list[j], j < cap -- records of 3 * nOwn * cap ints whatever the map's capacity is */
int cs_register_candidates_pack_list_dev(int device, void* hip_stream, int cap, int nCams, int cam0, int nOwn, const int* d_list, const int* d_slot,
                                         const int* d_flags, const unsigned char* d_mergeable, int* d_send);
int cs_register_candidates_unpack_list_dev(int device, void* hip_stream, int cap, int nCams, int nOwn, int skipRank, const int* d_list,
                                           const int* d_recv, int* d_slot, int* d_flags, unsigned char* d_mergeable);

/* ------------------------------------------------------------------------------------------
 * What a frame does with the cameras' new poses: the gate + seqTriangulate loop of poseUpdate3D and the dynamic-point test
 * ------------------------------------------------------------------------------------------
 * Replaces the second half of SingleSLAM::poseUpdate3D (src/app/SL_SingleSLAM.cpp:672-708, nodes of getStaticMappedTrackNodes
 * :60-75) and SingleSLAM::detectDynamicFeaturePoints (:784-824, nodes of getUnMappedAndDynamicTrackNodes :91-105) for every
 * camera of a group, behind cs_pose_intracam_batch_dev.  All pointers are DEVICE pointers unless noted.
 *  gate: per map point that isCertainStatic() and per camera IN CAMERA ORDER (CoSLAM::parallelPoseUpdate runs the cameras one
 *        after the other, src/app/SL_CoSLAM.cpp:398-410) the feature of this frame attached to it (d_pointFeat: the hand-back's
 *        nMap x nCams table): Mahalanobis distance of the projection under the new pose against 2.0 (6.0 with largeErr);
 *        inlier: reprojErr[slot] = the distance, the point and its covariance updated in place by seqTriangulate; outlier:
 *        reprojErr[slot] = the pixel distance, the point's CS_MAP_UNCERTAIN bit set (MapPoint::setUncertain) -- it is no node
 *        of the cameras that follow.
 *  dynamic test: per slot with a track of >= minLen frames whose feature is unmapped or on a certain-dynamic point: the
 *        epipolar error of the current position against the track's past positions (ring of the last histLen frames' pixels
 *        and poses, cs_track_history), counted against maxEpiErr; more than minOutNum: isStatic[slot] = 0
 *        (TYPE_FEATPOINT_DYNAMIC), else an unmapped slot's isStatic = 1.  maxLen has no effect, as in the reference (:799: its
 *        loop never advances the counter it compares with maxLen); the walk is bounded by histLen.
 * reprojErr[] and isStatic[] persist per slot between frames: that is propagateFeatureStates' hand-down along a track
 * (:40-42, :54); isStatic[] is what the next frame's cs_klt_handback_dev takes as cs_handback_cam.isStatic.
 * project / getProjectionCovMat / mat22Inv / mahaDist2 / dist2 / seqTriangulate / formEMat / getFMat / epipolarError are
 * un-vendored LibVisualSLAM: definitions in DESIGN.md. */
#define CS_MAP_DYNAMIC 1   /* MapPoint::iLocalType == TYPE_MAP_DYNAMIC (src/slam/SL_MapPoint.h:22-24) */
#define CS_MAP_FALSE 2     /*                      == TYPE_MAP_FALSE; neither bit: TYPE_MAP_STATIC */
#define CS_MAP_UNCERTAIN 4 /* MapPoint::bUncertain */
typedef struct cs_poseupdate_cam {
    const double* K;         /* 9 */
    const double* iK;        /* 9: SingleSLAM::iK (dynamic test) */
    const double* xy;        /* 2N: the hand-back's undistorted pixels of this frame */
    const int* state;        /* N: the hand-back's state (0 tracked, 1 new: the slot has a feature in this frame) */
    const int* slot2map;     /* N */
    const int* trackSpan;    /* 2N (dynamic test) */
    double* reprojErr;       /* N in/out: FeaturePoint::reprojErr (gate) */
    unsigned char* isStatic; /* N in/out: FeaturePoint::type == TYPE_FEATPOINT_STATIC (dynamic test) */
} cs_poseupdate_cam;
/* A history handle also owns scratch that the map-point calls below fill and read (the camera centres of the ring by walk depth,
 * the worklist of cs_map_points_classify_dev): calls on ONE handle belong on ONE stream at a time, also those that take it
 * as `const` (two streams running e.g. cs_refine_map_points_dev on the same handle at once would share that scratch). */
typedef struct cs_track_history cs_track_history;
cs_track_history* cs_track_history_create(int device, int nCams, int N, int histLen /* <= 512 */);
/* The same with a STORE behind the walks: storeLen (histLen <= storeLen <= 65536) frames of pixels and poses are kept -- 16 N + 96 bytes
 * per camera and frame: 4096 frames of 8 cameras x 2000 slots are 1 GB of the 288 -- while every walk stays histLen deep; only
 * cs_register_mergability_running_dev reads beyond (the tail of a whole-track walk whose cached verdict it has to rebuild). */
cs_track_history* cs_track_history_create_ex(int device, int nCams, int N, int histLen, int storeLen);
void cs_track_history_destroy(cs_track_history* h);
int cs_track_history_frames(const cs_track_history* h); /* consecutive frames held (<= histLen) */
/* cams: HOST array of nCams records; cameras cam0 .. cam0 + nCamsRun - 1 are processed (all: 0, nCams).  d_R nCams x 9, d_t
 * nCams x 3: the new poses.  d_numNodes / d_numOut / d_numDyn: nCams counters or NULL (poseUpdate3D's `num`, `numOut`,
 * detectDynamicFeaturePoints' return value). */
int cs_pose_update3d_dev(int device, void* hip_stream, int nCams, int cam0, int nCamsRun, const cs_poseupdate_cam* cams, int N,
                         const int* d_pointFeat, int nMap, const double* d_R, const double* d_t, double* d_mapPts, double* d_mapCov,
                         unsigned char* d_mapFlags, int largeErr, double pixelErrVar, int* d_numNodes, int* d_numOut);
/* `frame`: this frame's number; the history is kept while the numbers are consecutive (Track2D::length() counts frames) */
int cs_detect_dynamic_dev(cs_track_history* h, void* hip_stream, int cam0, int nCamsRun, const cs_poseupdate_cam* cams,
                          const double* d_R, const double* d_t, int nMap, const unsigned char* d_mapFlags, int frame, int maxLen,
                          int minLen, int minOutNum, double maxEpiErr, int* d_numDyn);
/* both, all cameras, ONE launch (what a frame loop calls behind cs_pose_intracam_batch_dev) */
int cs_pose_update_frame_dev(cs_track_history* h, void* hip_stream, const cs_poseupdate_cam* cams, const int* d_pointFeat, int nMap,
                             const double* d_R, const double* d_t, double* d_mapPts, double* d_mapCov, unsigned char* d_mapFlags,
                             int largeErr, double pixelErrVar, int frame, int maxLen, int minLen, int minOutNum, double maxEpiErr,
                             int* d_numNodes, int* d_numOut, int* d_numDyn);

/* CoSLAM::staticCheckMergability (src/app/SL_CoSLAM.cpp:714-729) for every candidate of a registration search, in one launch:
 * the candidate feature and every earlier feature of its track must lie within Mahalanobis distance 1 of the map point's
 * projection under the pose of its own frame (covariance J cov J^T + pixelErrVar^2 I).  d_slot: the P x nCams candidate table of
 * cs_register_search_dev (entries < 0: no candidate); the tracks' past pixels and the frames' poses come from the history h, whose
 * newest entry must be THIS frame (what cs_pose_update_frame_dev / cs_detect_dynamic_dev pushed); cams: K and trackSpan of every
 * camera.  d_mergeable [P x nCams]: 1 mergeable, 0 not, 255 no candidate, 2 = the track is LONGER than the history (the reference
 * walks the whole chain: the older frames cannot be judged, so the held ones are not looked at either -- a caller that must not be more
 * permissive than the reference treats 2 as 0, as cs_register_decide_static_dev does; or sizes the history for its tracks).  (The search's own flag bit 2 is the first term of this
 * walk -- this frame only.)  What the registration loops do with a mergeable candidate -- pointer updates, refineMapPoint,
 * checkUnify -- stays with the caller; compareFeaturePt, which they also call, returns true whatever its NCC score is (:546-558). */
int cs_register_mergability_dev(const cs_track_history* h, void* hip_stream, const cs_poseupdate_cam* cams, int P, const double* d_M,
                                const double* d_cov, const int* d_slot, double pixelErrVar, unsigned char* d_mergeable);

int cs_register_mergability_range_dev(const cs_track_history* h, void* hip_stream, int cam0, int nCamsRun, const cs_poseupdate_cam* cams,
                                      int P, const double* d_M, const double* d_cov, const int* d_slot, double pixelErrVar,
                                      unsigned char* d_mergeable); /* columns cam0 .. cam0 + nCamsRun - 1 only */
/* The walk over WHOLE tracks (the reference's: fp, fp->preFrame, ... to the track's first frame) as a RUNNING verdict: the newest
 * histLen frames of a candidate's track are walked every frame with the point, its covariance and the poses as they stand; the
 * verdict over every OLDER frame is cached per (map point, camera) in d_cache and extended by one term when a frame crosses from
 * the window into the tail.  The cache entry is dropped, and the tail walked again from the frames the store holds
 * (cs_track_history_create_ex), when the candidate is another slot, the slot's track restarted, or the point moved by more than
 * tolPix pixels in this camera's image since the tail was judged.  With point and poses held still the verdicts are the whole-track
 * walk's, term for term (tests/golden/mergability_long_golden.npz: the reference's own function on tracks of 200-420 frames).
 * d_cache: cs_register_mergability_cache_bytes(P, nCams) bytes, zero-filled before the first call, point p of one call = point p of
 * the next; verdict 2 only for a track whose first frame has left the store; d_counts [4] or NULL, added to: cache hits, full tail
 * walks, verdicts 2, tail terms evaluated. */
size_t cs_register_mergability_cache_bytes(int P, int nCams);
int cs_register_mergability_running_dev(const cs_track_history* h, void* hip_stream, int cam0, int nCamsRun, const cs_poseupdate_cam* cams,
                                        int P, const double* d_M, const double* d_cov, const int* d_slot, double pixelErrVar, double tolPix,
                                        void* d_cache, unsigned char* d_mergeable, int* d_counts);
/* ... for the rows d_list[0 .. nList) of whole-map tables only (a compact list: the first entry < 0 ends it; cs_register_pass::list's
 * companion; NULL: all P rows): d_M, d_cov, d_slot, d_cache, d_mergeable are indexed by the map index, P = the tables' rows.  d_flags (the search's flags
 * table, or NULL): a candidate that already carries a map point (bit 0 clear) is not judged -- verdict 0; the registration walks end at
 * such a feature or ask checkUnify about it, they never put it to this test. */
int cs_register_mergability_running_list_dev(const cs_track_history* h, void* hip_stream, int cam0, int nCamsRun, const cs_poseupdate_cam* cams,
                                             int P, const int* d_list, int nList, const double* d_M, const double* d_cov, const int* d_slot,
                                             const int* d_flags, double pixelErrVar, double tolPix, void* d_cache, unsigned char* d_mergeable,
                                             int* d_counts);

/* RobustBundleRTS::updateNewPosesPoints (src/app/SL_CoSLAMRobustBA.cpp:248-271) in one launch: behind a bundle adjustment and the
 * relaxation of the non-key frames, every map point with lastFrame > firstKeyFrame is triangulated again from the moved poses --
 * a locally static point (flags without CS_MAP_DYNAMIC / CS_MAP_FALSE) by updateStaticPointPosition (src/slam/SL_CoSLAMHelper.cpp:
 * 338-394: per camera holding a feature of it, that feature and the one further back on the same track with the largest parallax
 * angle at the point), a locally dynamic point by updateDynamicPointPosition (:455-484: this frame's features, at least one of
 * them dynamic, at least two); triangulateMultiView + getTriangulateCovMat (definitions: DESIGN.md 3.9) write d_mapPts [nMap][3]
 * and d_mapCov [nMap][9] IN PLACE; a point with fewer than two views is left alone.
 * The tracks' pixels AND the frames' poses come from the history h, newest entry = the frame whose hand-back produced d_pointFeat
 * ([nMap][nCams] slot of the point's feature, < 0 none).  The reference's features share their frame's CamPoseItem, so the
 * adjusted poses reach them by themselves; here the ring holds copies: cs_track_history_set_poses_dev scatters n poses (d_R [n][9],
 * d_t [n][3]) into the entries of (d_cam[i], d_frame[i]) first -- the BA's key poses (output(), :283-285) and the relaxed non-key
 * poses (updateNonKeyCameraPoses, :239-244; cs_posegraph_relax_dev's newR / newT); pairs the ring does not hold are skipped.
 * d_lastFrame [nMap] (MapPoint::lastFrame) or NULL = every point passes the frame test; d_isCurrent [nMap] or NULL = all points are
 * on curMapPts: a point with isCurrent == 0 is on actMapPts, whose dynamic points the reference never updates (:266-269 tests
 * isLocalStatic() twice) -- such a point must still have its features of THIS frame in d_pointFeat to be touched at all.
 * cams: K, iK, trackSpan, isStatic (feature types) of every camera.  d_counts [2] or NULL: static / dynamic points re-triangulated.
 * The launch keeps the camera centres of the ring's entries in a scratch array of h: calls that share a history must be ordered
 * (one stream, or events), like every other call that advances or reads it. */
int cs_track_history_set_poses_dev(cs_track_history* h, void* hip_stream, int n, const int* d_cam, const int* d_frame, const double* d_R,
                                   const double* d_t);
/* A run of consecutive frames of every camera out of / into the history, camera-major (d_R [nCams][nFrames][9], d_t [nCams][nFrames][3]):
 * the nodes of the camera graphs RobustBundleRTS::constructCameraGraphs builds (src/app/SL_CoSLAMRobustBA.cpp:182-227: every pose
 * from the window's first key frame to the newest frame) and where updateNonKeyCameraPoses writes the relaxed poses (:230-247).
 * All of firstFrame .. firstFrame + nFrames - 1 must be in the ring. */
int cs_track_history_get_span_dev(const cs_track_history* h, void* hip_stream, int firstFrame, int nFrames, double* d_R, double* d_t);
int cs_track_history_set_span_dev(cs_track_history* h, void* hip_stream, int firstFrame, int nFrames, const double* d_R, const double* d_t);
int cs_track_history_newest_frame(const cs_track_history* h); /* the frame of the newest entry */
int cs_track_history_cams(const cs_track_history* h);
int cs_update_new_poses_points_dev(const cs_track_history* h, void* hip_stream, const cs_poseupdate_cam* cams, const int* d_pointFeat,
                                   int nMap, const int* d_lastFrame, const unsigned char* d_isCurrent, int firstKeyFrame,
                                   double* d_mapPts, double* d_mapCov, const unsigned char* d_mapFlags, double pixelErrVar,
                                   int* d_counts);
/* CoSLAM::refineMapPoint (src/app/SL_CoSLAM.cpp:666-713) for every map point d_select names (uint8 [nMap]; NULL = all) in one
 * launch: what the registration loops call on a point that has just gained a feature (:896, :948, :1166).  The views are those of
 * updateStaticPointPosition (per camera holding a feature of the point in d_pointFeat: that feature and the widest-parallax one
 * further back on its track), then triangulateMultiView + getTriangulateCovMat IN PLACE -- whatever the point's type, no frame
 * test.  The reference does not look at the number of views; here a point with fewer than two is left alone.  The same kernel,
 * history and ordering rules as cs_update_new_poses_points_dev; cams: K, iK, trackSpan.  d_count [1] or NULL: points refined.
 * This form knows the features of THIS frame only and walks the slot's current track, [trackSpan first, this frame]: a point that
 * still holds a STALE feature in another camera, or was registered to a new track where it held an older one (the reference re-links
 * pFeat->preFrame to it, :775-779), has other views in the reference.  cs_refine_map_points_ref_dev (below) is the form that follows
 * the reference there. */
int cs_refine_map_points_dev(const cs_track_history* h, void* hip_stream, const cs_poseupdate_cam* cams, const int* d_pointFeat, int nMap,
                             const unsigned char* d_select, double* d_mapPts, double* d_mapCov, double pixelErrVar, int* d_count);

/* ---- SingleSLAM::newMapPoints: the intra-camera source of new map points (src/app/SL_SingleSLAM.cpp:922-1004) ------------------------------
 * What CoSLAM::genNewMapPoints calls for a camera that IsReadyForKeyFrame (SL_CoSLAM.cpp:1310-1330): every unmapped feature of this frame
 * on a track of at least minTrackLen frames (Param::nMinFeatTrkLen = 20; getUnMappedAndTrackedFeatPts, :152-172) is triangulated from its
 * own track -- the oldest feature of the track the history still holds against the current one --, thrown out when the point is behind the
 * camera, nearer than sqrt(trace cov) (:960-962) or re-projects further off than maxEpiErr (2.0) in either view; refineTriangulation
 * (:1005-1049) pairs the current view with the widest-parallax one behind it (at most maxWalk nodes back; the reference walks the whole
 * track) and the tests run once more.  Cameras: all (d_ready NULL) or those with d_ready[c] >= readyMin (cs_keyframe_ready_dev's codes: the
 * reference asks > 1 with several cameras, > 0 with one).  A dynamic feature stops the reference's walk (:938-944): the slot's isStatic
 * stands for its whole track here.  New points are appended behind *d_mapCount in (camera, slot) order -- the order the reference's loops
 * create them in -- with MapPoint(M, firstFrame = the oldest view's frame), the covariance, TYPE_MAP_STATIC, bNewPt, the feature attached
 * (d_pointFeat row, cams[c].slot2map -- written through the const pointer); points beyond mapCap are dropped and counted.
 * d_scratch: cs_newpts_intracam_scratch_bytes(nCams, N).  d_counts [3] or NULL: candidates tried, points added, points dropped.
 * The current frame is the history's newest.  Pinned against the reference's own function compiled in place
 * (tests/cxx/ref_intracam_newpts_test.cpp -> tests/golden/intracam_newpts_golden.npz); binTriangulate / getBinTriangulateCovMat are the
 * multi-view definitions of DESIGN.md 5.1 over two views. */
size_t cs_newpts_intracam_scratch_bytes(int nCams, int N);
int cs_newpts_intracam_dev(const cs_track_history* h, void* hip_stream, const cs_poseupdate_cam* cams, const int* d_ready, int readyMin,
                           int minTrackLen, int maxWalk, double maxEpiErr, double pixelErrVar, double* d_mapPts, double* d_mapCov,
                           unsigned char* d_mapFlags, unsigned char* d_newPt, int* d_firstFrame, int* d_pointFeat, int mapCap, int* d_mapCount,
                           void* d_scratch, int* d_counts);

/* ---- The key-frame decision (CoSLAM::genNewMapPoints' first half; VERDICT r04 missing 7) ------------------------------------------------
 * CoSLAM::IsReadyForKeyFrame (src/app/SL_CoSLAM.cpp:1269-1279) for every camera in one launch: READY_FOR_KEY_FRAME_DECREASE (1) when the
 * frame's features whose map point is older than the camera's last key pose number fewer than `ratio` (m_mappedPtsReduceRatio, 0.93)
 * times that key pose's nMappedPts, or fewer than 30 (:1249-1268); else _VIEWANGLE (2) when the angle at the mean of the frame's mapped,
 * not false points (:1224-1247) between the last self-motion key pose's centre and the current one exceeds minViewAngleDeg
 * (m_minViewAngleChange, 5.0; the reference's PI = 3.14 kept); else _TRANSLATION (3) when the two centres are further apart than
 * minTranslation (m_minCamTranslation); else 0.  Both feature loops stop before the frame's LAST feature (`fp != pTail`, :1233, :1258);
 * SingleSLAM::getNumMappedStaticPts (the count that becomes the next key pose's nMappedPts) does not.  Features = the hand-back's records
 * (state 0 / 1, slot2map) in slot order.  d_ready [nCams + 2]: the codes, then genNewMapPoints' nReady and `decrease` (:1298-1309);
 * d_mapped [2 nCams]: m_nMappedStaticPts, then the decrease test's count; d_center [nCams][3]; d_stats [5] or NULL, accumulated: frames
 * with nReady > 0, frames with decrease, cameras that said 1 / 2 / 3.  addKeyFrame != 0: when `decrease` holds, every camera's key-pose
 * state is moved on as CoSLAM::addKeyFrame -> SingleSLAM::addKeyPose does (:1280-1293, SL_SingleSLAM.cpp:835-862): keyFrame = curFrame,
 * keyMapped = m_nMappedStaticPts, and the current pose becomes the self-motion key pose of the cameras whose code is > 0.
 * Pinned against the reference's own functions compiled in place (tests/cxx/ref_keyframe_test.cpp -> tests/golden/keyframe_golden.npz). */
typedef struct {
    const int* state;      /* N: the hand-back's state (0 tracked, 1 new: a feature of this frame) */
    const int* slot2map;   /* N: map point index of the slot's feature or -1 */
    const double* R;       /* 9, 3: m_camPos.current() */
    const double* t;
    int* keyFrame;         /* [1] in / out: m_keyPose.tail->frame */
    int* keyMapped;        /* [1] in / out: m_keyPose.tail->nMappedPts */
    double* selfR;         /* 9, 3 in / out: m_selfKeyPose.back()->cam */
    double* selfT;
} cs_keyframe_cam;
int cs_keyframe_ready_dev(int device, void* hip_stream, int nCams, int N, const cs_keyframe_cam* cams, int nMap, const double* d_mapPts,
                          const unsigned char* d_mapFlags, const int* d_firstFrame, int curFrame, double ratio, double minViewAngleDeg,
                          double minTranslation, int addKeyFrame, int* d_ready, int* d_mapped, double* d_center, int* d_stats);

/* A frame loop whose key frames the decision places WITHOUT a host wait per frame: behind a frame's registration, what a key
 * frame's push would read of it -- every camera's xy [nCams][2N], state and slot2map [nCams][N], the poses [nCams][9] / [3] -- into one slot of
 * the caller's ring of snapshots, and the decision word (d_word = cs_keyframe_ready_dev's d_ready + nCams + 1) into PINNED host memory
 * (h_word: hipHostMalloc'ed, read by the host behind an event it records next on the stream).  One launch. */
int cs_keyframe_snapshot_dev(int device, void* hip_stream, int nCams, int N, const double* d_xy, const int* d_state, const int* d_slot2map,
                             const double* d_R, const double* d_t, const int* d_word, double* d_xyOut, int* d_stateOut, int* d_slot2mapOut,
                             double* d_ROut, double* d_tOut, int* h_word);

/* ---- MapPoint::pFeatures as the reference holds them: feature references (round 5; VERDICT r04 missing 3) ---------------------------------
 * p->pFeatures[c] is the feature of this frame while camera c tracks the point -- and STAYS what it last was when the camera loses it
 * (nothing clears the pointer): updateStaticPointPosition / updateDynamicPointPosition (src/slam/SL_CoSLAMHelper.cpp:338-394, 455-484),
 * CoSLAM::refineMapPoint (src/app/SL_CoSLAM.cpp:666-713) and checkUnify (:561-665) take such a stale feature as a view, with the pose of its
 * own frame.  Behind a feature hangs its FeaturePoint::preFrame chain: the earlier frames of its track, and -- once the point is registered
 * to a NEW track in a camera where it still holds an older feature -- `pFeat->preFrame = p->pFeatures[iCam]` (:775-779, :997-1000): the OLD
 * chain behind the new feature, the new track's own earlier frames cut off.  cs_feat_ref is that pointer, cs_feat_seg a linked segment:
 *   cs_feat_ref {slot (< 0: none), frame (of the feature), first (oldest frame of the run of consecutive frames behind it on `slot`),
 *                seg (index of the first linked segment in the camera's pool, -1 none)}        table [nMap][nCams], the caller's
 *   cs_feat_seg {slot, last, first, next}                                                      pool [nCams][cap], the history's
 * Pixels and poses of a node are the history's entry of its frame (cs_track_history_create_ex: storeLen frames are kept); a walk ends at a
 * node older than the store and after histLen NODES (the bound every walk here has; the reference has none).
 * cs_feat_ref_advance_dev, EVERY frame behind the registration's decisions, does to the table what the reference does to the pointers:
 * tracked on -> the frame moves; gained a feature of another track while holding an older one -> the old reference becomes a pool
 * segment, the new one {slot, frame, first = frame, that segment}; first feature in that camera -> {slot, frame, the track's first frame
 * (trackSpan), -1}; not seen -> unchanged (stale) -- unless the reference was alive in the frame before and its slot's track lives on
 * without the point: detached by the classification or a unification (:470-472, :810), cleared.  d_refStatic [nMap][nCams] or NULL: the
 * feature's type (isStatic[slot]) as of its own frame.  d_counts [5] or NULL (accumulated, not cleared here): tracked on, first, re-linked,
 * links dropped because the pool (32768 segments per camera, never recycled) is full, detached.  cams: trackSpan, isStatic. */
typedef struct { int slot, frame, first, seg; } cs_feat_ref;
typedef struct { int slot, last, first, next; } cs_feat_seg;
int cs_feat_ref_advance_dev(cs_track_history* h, void* hip_stream, const cs_poseupdate_cam* cams, int nMap, const int* d_pointFeat,
                            int curFrame, cs_feat_ref* d_featRef, unsigned char* d_refStatic, int* d_counts);
/* a FURTHER call within the same frame over the rows d_list[0 .. nList) only (entries < 0 skipped): behind a registration round that changed
 * just those points' features (cs_register_revisit_decide_dev) -- every other row stands as the frame's first call left it */
int cs_feat_ref_advance_list_dev(cs_track_history* h, void* hip_stream, const cs_poseupdate_cam* cams, int nMap, const int* d_pointFeat,
                                 int curFrame, cs_feat_ref* d_featRef, unsigned char* d_refStatic, int* d_counts, const int* d_list, int nList);
/* cs_feat_ref_advance_(list_)dev and cs_refine_map_points_ref_dev of the same rows as ONE launch (the pose stream's launches are its
 * time: DESIGN.md 3.13.1): the rows of d_list[0 .. nList) with d_select[m] != 0 are advanced AND refined (CoSLAM::refineMapPoint,
 * src/app/SL_CoSLAM.cpp:666-713, over the references just written), every other row -- of the whole map when advanceAll, else of the
 * list -- is advanced only.  A row that is to be refined has to be on the list.  clearSelect: the marks are consumed (set to 0).  Same
 * tables and map as the two calls. */
int cs_feat_ref_advance_refine_dev(cs_track_history* h, void* hip_stream, const cs_poseupdate_cam* cams, int nMap, const int* d_pointFeat,
                                   int curFrame, cs_feat_ref* d_featRef, unsigned char* d_refStatic, int* d_counts, const int* d_list, int nList,
                                   int advanceAll, unsigned char* d_select, int clearSelect, double* d_mapPts, double* d_mapCov,
                                   double pixelErrVar);
/* the pools: device pointer ([nCams][cap]), capacity per camera, device counters [nCams] */
int cs_track_history_segments(const cs_track_history* h, cs_feat_seg** d_pool, int* cap, int** d_count);
/* the counters copied to the host (counts [nCams]); synchronous */
int cs_track_history_segment_counts(const cs_track_history* h, int* counts);
/* n segments per camera from the host ([nCams][n]) into the pools, counters = n (tests, restoring a saved state); synchronous */
int cs_track_history_load_segments(cs_track_history* h, const cs_feat_seg* segs, int n);
/* the first n segments of every camera's pool back ([nCams][n]); synchronous */
int cs_track_history_download_segments(const cs_track_history* h, cs_feat_seg* segs, int n);
/* cs_update_new_poses_points_dev / cs_refine_map_points_dev / cs_check_unify_dev with the points' features as references: stale features
 * are views, the widest-parallax walk follows the links.  The current frame is the history's newest.  Pinned against the reference's own
 * functions on chains built with its classes (tests/cxx/ref_update_points_test.cpp golden_relink -> tests/golden/update_points_relink_golden.npz). */
int cs_update_new_poses_points_ref_dev(const cs_track_history* h, void* hip_stream, const cs_poseupdate_cam* cams, const cs_feat_ref* d_featRef,
                                       const unsigned char* d_refStatic, int nMap, const int* d_lastFrame, const unsigned char* d_isCurrent,
                                       int firstKeyFrame, double* d_mapPts, double* d_mapCov, const unsigned char* d_mapFlags, double pixelErrVar,
                                       int* d_counts);
int cs_refine_map_points_ref_dev(const cs_track_history* h, void* hip_stream, const cs_poseupdate_cam* cams, const cs_feat_ref* d_featRef,
                                 int nMap, const unsigned char* d_select, double* d_mapPts, double* d_mapCov, double pixelErrVar, int* d_count);

/* CoSLAM::mapPointsClassify (src/app/SL_CoSLAM.cpp:418-520) over the references: from the next call on cs_map_points_classify_dev and
 * cs_pose_update_classify_frame_dev read d_featRef ([nMap][nCams], the table cs_feat_ref_advance_dev keeps: as the END of the previous
 * frame left it, or already at this frame) next to d_pointFeat (this frame's features): a camera that lost the point still gives
 * isStaticPoint (inside its 60 frames), isLittleMove and isStaticRemovable its stale feature with the pose of that frame
 * (src/slam/SL_CoSLAMHelper.cpp:67-115, :117-250, :314-330), isStaticPoint's backward walks follow the linked segments, the view
 * isStaticRemovable drops may be a stale one (its reference is cleared: `p->pFeatures[outlierViewId] = 0`, :470-472), a point that
 * returns to static sets d_refStatic ([nMap][nCams] or NULL) of its stale features (:494-498).  d_featFrame / d_featFirst are then not
 * read.  d_featRef NULL: back to d_pointFeat alone.  Pinned against the reference's own functions over chains built with its classes
 * (tests/cxx/ref_classify_test.cpp golden_relink -> tests/golden/classify_relink_golden.npz). */
int cs_track_history_set_classify_refs(cs_track_history* h, cs_feat_ref* d_featRef, unsigned char* d_refStatic);

/* The bMerge walks (cs_register_decide_merge_dev / _list_dev below) over the references: from the next call on checkUnify reads both points'
 * rows as the reference holds MapPoint::pFeatures at that moment (this frame's features with their chains, stale features of cameras that
 * lost the point), a feature attached where the point held a stale one gets the old chain linked behind it at once (SL_CoSLAM.cpp:775-779;
 * the table and the camera's pool are written), and the hand-over of a unification follows `pFt && !p->pFeatures[v]` (:806-816): a stale
 * feature of the walking point blocks it in its camera, the other point's stale features move with their chains.  NULL: this frame's
 * features alone.  Pinned against the reference's own curStaticPointsRegInGroup over stale and re-linked chains
 * (tests/cxx/ref_decide_test.cpp golden_relink -> tests/golden/decide_relink_golden.npz). */
int cs_track_history_set_merge_refs(cs_track_history* h, cs_feat_ref* d_featRef, unsigned char* d_refStatic);

/* CoSLAM::checkUnify (src/app/SL_CoSLAM.cpp:561-665) for nPairs pairs of map points in one launch: what the registration loops ask
 * on a conflict -- the point's nearest feature already carries another static point (:791-796, bMerge: every 50th frame).  Per pair
 * the slots of both points' features of this frame per camera (d_pf1 / d_pf2 [nPairs][nCams], < 0 none) and the points' positions
 * (d_M1 / d_M2 [nPairs][3]: the widest-parallax second view of a feature's track is chosen around its own point); out: d_ok [nPairs]
 * (1: the views of both points agree with ONE point), d_M [nPairs][3], d_cov [nPairs][9] -- what p->updatePosition(M, cov) takes when
 * ok.  The gate's `Rs + 3 * i` of :657 is reproduced as written.  History, cams (K, iK, trackSpan) and ordering rules as
 * cs_refine_map_points_dev.  What happens on ok -- the second point set false, its features moved over (:797-822) -- is the caller's. */
int cs_check_unify_dev(const cs_track_history* h, void* hip_stream, const cs_poseupdate_cam* cams, int nPairs, const int* d_pf1,
                       const int* d_pf2, const double* d_M1, const double* d_M2, double pixelErrVar, unsigned char* d_ok, double* d_M,
                       double* d_cov);
/* ... with the two points' features as references (rows of cs_feat_ref, [nPairs][nCams] each) */
int cs_check_unify_ref_dev(const cs_track_history* h, void* hip_stream, const cs_poseupdate_cam* cams, int nPairs, const cs_feat_ref* d_ref1,
                           const cs_feat_ref* d_ref2, const double* d_M1, const double* d_M2, double pixelErrVar, unsigned char* d_ok, double* d_M,
                           double* d_cov);
/* curStaticPointsRegInGroup with bMerge == true (src/app/SL_CoSLAM.cpp:854-898, 731-830: every 50th frame, CoSLAMThread.cpp:117-118) -- the
 * walks in the reference's order on ONE wave: unmapped mergeable candidates are attached as above; a candidate that carries ANOTHER static
 * point asks checkUnify with both points as they stand and, on a yes, the walking point takes the unified position, the other becomes false
 * and hands over its features in the cameras up to the one of the conflict (:797-826 as written), which ends the walk; on a no the walk
 * goes on.  Tables as cs_register_decide_static_dev's (one search + cs_register_mergability_dev before it); d_mapFlags / d_pointFeat /
 * d_mapPts / d_mapCov [P] and the cameras' slot2map are updated IN PLACE; d_scratch: P bytes; d_counts [4] or NULL: features attached,
 * points registered, points unified away, checkUnify calls; onlyCam >= 0: that camera's loop only (then refine d_regged and search again
 * before the next: the reference's run step for step).  Sequential: ~5 us per conflict -- the parity mode's entry, pinned to the reference's
 * own run with bMerge on tests/golden/decide_golden.npz scenes 5, 6; the dynamic points' loops ignore bMerge (cs_register_decide_kinds_dev,
 * kinds 2, behind it). */
int cs_register_decide_merge_dev(const cs_track_history* h, void* hip_stream, const cs_poseupdate_cam* cams, int P, int mapBase, const int* d_slot,
                                 const int* d_flags, const unsigned char* d_mergeable, unsigned char* d_mapFlags, int* d_pointFeat, double* d_mapPts,
                                 double* d_mapCov, double pixelErrVar, unsigned char* d_attached, unsigned char* d_regged, void* d_scratch,
                                 int* d_counts, int onlyCam);
/* the same walking a LIST of the points (cs_register_list_current_dev: the frame's current points in map order, entries < 0 behind them;
 * mapBase must be 0) over whole-map tables of P rows: the single wave's loops are as long as the list, not as the map's capacity */
size_t cs_register_decide_merge_scratch_bytes(int P, int nList, int nCams); /* d_scratch of the list form (the plain form: P bytes) */
int cs_register_decide_merge_list_dev(const cs_track_history* h, void* hip_stream, const cs_poseupdate_cam* cams, int P, int mapBase,
                                      const int* d_list, int nList, const int* d_slot, const int* d_flags, const unsigned char* d_mergeable,
                                      unsigned char* d_mapFlags, int* d_pointFeat, double* d_mapPts, double* d_mapCov, double pixelErrVar,
                                      unsigned char* d_attached, unsigned char* d_regged, void* d_scratch, int* d_counts, int onlyCam);


/* CoSLAM::mapPointsClassify (src/app/SL_CoSLAM.cpp:418-520) in one launch: what CoSLAM::poseUpdate runs every frame behind the pose
 * update (:381-385, pixelVar = 12.0) -- every map point with a feature in this frame that is uncertain (CS_MAP_UNCERTAIN: what the
 * gate of cs_pose_update_frame_dev made of it) or locally dynamic is decided again: static (isStaticPoint over the last 60 frames),
 * dynamic (isDynamicPoint), static without its worst view (isStaticRemovable: that feature is detached), or false; dynamic points
 * that stand still for more than 50 frames (isLittleMove) may return to static (src/slam/SL_CoSLAMHelper.cpp:67-330).  In / out per
 * point: d_mapPts, d_mapCov, d_mapFlags (setFalse keeps CS_MAP_UNCERTAIN and clears CS_MAP_DYNAMIC, setLocalDynamic / setLocalStatic
 * clear the others), d_newPt (MapPoint::bNewPt, uint8), d_staticFrameNum; in: d_firstFrame (MapPoint::firstFrame).  A point's
 * features are MapPoint::pFeatures[iCam]: d_pointFeat [nMap][nCams] (slot, < 0 none; a detached feature becomes -1 and its slot's
 * entry in cams[c].slot2map -- written through the const pointer when given -- too), with, optionally, d_featFrame (the feature's
 * frame: a camera that lost the point keeps its last feature in the reference and three of the helpers still use it; NULL = all of
 * this frame) and d_featFirst (first frame of that feature's track; NULL = the slot's trackSpan).  cams: K, iK, trackSpan, isStatic
 * (feature types; a point that returns to static sets its current features' types), slot2map.  Pixels and poses come from the
 * history h, whose newest entry must be curFrame; features older than the history are treated as absent (60 frames are looked at).
 * d_counts [2] or NULL: points examined / points that became false.  One lane per map point (the examined points are few). */
int cs_map_points_classify_dev(const cs_track_history* h, void* hip_stream, const cs_poseupdate_cam* cams, int* d_pointFeat, int nMap,
                               const int* d_featFrame, const int* d_featFirst, int curFrame, double* d_mapPts, double* d_mapCov,
                               unsigned char* d_mapFlags, unsigned char* d_newPt, int* d_staticFrameNum, const int* d_firstFrame,
                               double pixelVar, int* d_counts);
/* cs_pose_update_frame_dev and cs_map_points_classify_dev of the same frame -- CoSLAM::poseUpdate as a whole (src/app/SL_CoSLAM.cpp:398-417:
 * every camera's poseUpdate3D + detectDynamicFeaturePoints, then mapPointsClassify(12.0)) -- as TWO launches instead of four: the gate's
 * lane of a map point also decides whether the classification examines it, the camera centres of the walks are blocks of the same
 * launch, then the classification's worker.  Same arguments as the two calls (d_clsCounts = the classification's d_counts), same
 * results; nMap >= 1. */
int cs_pose_update_classify_frame_dev(cs_track_history* h, void* hip_stream, const cs_poseupdate_cam* cams, int* d_pointFeat, int nMap,
                                      const double* d_R, const double* d_t, double* d_mapPts, double* d_mapCov, unsigned char* d_mapFlags,
                                      int largeErr, double pixelErrVar, int frame, int maxLen, int minLen, int minOutNum, double maxEpiErr,
                                      int* d_numNodes, int* d_numOut, int* d_numDyn, const int* d_featFrame, const int* d_featFirst,
                                      unsigned char* d_newPt, int* d_staticFrameNum, const int* d_firstFrame, double pixelVarClassify,
                                      int* d_clsCounts);

/* ------------------------------------------------------------------------------------------
 * Pose-graph relaxation of the non-key frames after a bundle adjustment, all camera graphs in one launch
 * ------------------------------------------------------------------------------------------
 * Replaces GlobalPoseGraph::computeNewCameraRotations + computeNewCameraTranslations
 * (src/slam/SL_GlobalPoseEstimation.cpp:52-219, 220-359) as RobustBundleRTS::updateNonKeyCameraPoses runs them per camera
 * after every BA (src/app/SL_CoSLAMRobustBA.cpp:230-247): each edge (id1 -> id2, R, t) with a free end asks for
 * R_2 = R R_1 and t_2 = R t_1 + t; the free nodes' poses are the least-squares solution with the fixed nodes (the key
 * frames the BA just moved) held, rotations projected to the nearest orthogonal matrix.  include/shim/slam/
 * coslam_posegraph.h carries the two member functions over this interface.
 * Topology is given once (create): nGraphs graphs, graph g owns nodes [nodePtr[g], nodePtr[g+1]) and edges
 * [edgePtr[g], edgePtr[g+1]) of the flat arrays; id1 / id2 are node indices LOCAL to the edge's graph (CamPoseEdge::id1,
 * id2); fixed[i] != 0 is CamPoseNode::fixed.  Any topology is accepted (band width follows the node order; CoSLAM's
 * chains give the minimum).  Edges with CamPoseEdge::uncertainScale are not supported (the shim refuses them).
 * Values per call, row-major: nodeR [N][9], nodeT [N][3] (CamPoseNode::R, t), edgeR [E][9], edgeT [E][3] (CamPoseEdge::R, t)
 * -> newR [N][9], newT [N][3] (CamPoseNode::newR, newt; fixed nodes are copied).  new* must not alias node*.
 * A free node that no edge constrains makes its graph fail: cs_posegraph_status / the host form return CS_ERR_NUMERIC. */
typedef struct cs_posegraph cs_posegraph;
int cs_posegraph_create(int device, int nGraphs, const int* nodePtr, const int* edgePtr, const unsigned char* fixed,
                        const int* id1, const int* id2, cs_posegraph** out);
void cs_posegraph_destroy(cs_posegraph* g);
int cs_posegraph_counts(const cs_posegraph* g, int* nNodes, int* nEdges, int* nComponents, int* maxHalfBandwidth);
/* device pointers, asynchronous on hip_stream */
int cs_posegraph_relax_dev(cs_posegraph* g, void* hip_stream, const double* d_nodeR, const double* d_nodeT,
                           const double* d_edgeR, const double* d_edgeT, double* d_newR, double* d_newT);
/* synchronises hip_stream and reads the last launch's verdict: CS_OK, or CS_ERR_NUMERIC with the failed graph */
int cs_posegraph_status(cs_posegraph* g, void* hip_stream, int* nFailed, int* firstFailedGraph);
/* host pointers: upload, one launch, read-back, status */
int cs_posegraph_relax(cs_posegraph* g, const double* nodeR, const double* nodeT, const double* edgeR, const double* edgeT,
                       double* newR, double* newT);
/* the data movement around the solve, on the device: every edge's relative transform from the poses of its ends
 * (constructCameraGraphs, src/app/SL_CoSLAMRobustBA.cpp:216-227: getRigidTransFromTo = R2 R1^T, t2 - R t1) -- run BEFORE
 * the adjusted key poses are written -- and the scatter of n poses (d_R [n][9], d_t [n][3]) into nodes d_nodeIdx[n]
 * (output(), :283-294: the BA's key poses into the fixed nodes; index < 0 = skip). */
int cs_posegraph_edges_dev(cs_posegraph* g, void* hip_stream, const double* d_nodeR, const double* d_nodeT, double* d_edgeR,
                           double* d_edgeT);
int cs_posegraph_set_poses_dev(int device, void* hip_stream, int n, const int* d_nodeIdx, const double* d_R, const double* d_t,
                               double* d_nodeR, double* d_nodeT);
/* The non-key-frame update of RobustBundleRTS::output() as a follow-up of the BA (cs_ba_set_followup(b,
 * cs_posegraph_after_ba, &rec)): scatter the adjusted key poses d_Rs / d_Ts (cs_ba_result_buffers) into the nodes d_camNode
 * names (one entry per BA camera, < 0 = not a node), then relax all graphs -- two launches behind the solve's last kernel.
 * The record must stay alive and unchanged while it is installed; d_edgeR / d_edgeT were computed (cs_posegraph_edges_dev)
 * from the poses before the adjustment. */
typedef struct cs_posegraph_after_ba_rec {
    cs_posegraph* g;
    int device;
    int nCams;              /* entries of d_camNode = cameras of the BA problem */
    const int* d_camNode;
    const double* d_Rs;
    const double* d_Ts;
    double* d_nodeR;
    double* d_nodeT;
    const double* d_edgeR;
    const double* d_edgeT;
    double* d_newR;
    double* d_newT;
} cs_posegraph_after_ba_rec;
int cs_posegraph_after_ba(void* hip_stream, void* rec /* cs_posegraph_after_ba_rec* */);

/* ------------------------------------------------------------------------------------------
 * Result text files of a run (host code, no device work)
 * ------------------------------------------------------------------------------------------
 * Replaces the body of CoSLAM::exportResultsVer1 (src/app/SL_CoSLAM.cpp:1914-2028) for a caller that holds its results as
 * arrays: writes input_videos.txt, mappts.txt, <c>_campose.txt and <c>_featpts.txt into dirPath (created if missing), byte
 * for byte what the reference's `ofstream <<` statements write.  The reference puts them in $HOME/slam_results/<time>/;
 * the directory is the caller's choice here. */
typedef struct cs_export_cam {
    const char* videoFilePath;    /* SingleSLAM::videoFilePath */
    const double* K;              /* 9 */
    const double* kc;             /* 5: SingleSLAM::k_c */
    int W, H;
    int startFrameInVideo;        /* frames are written as startFrameInVideo + f + 1 (CoSLAM::getFrameInVideo) */
    int nPoses;                   /* m_camPos.size(), >= 1 */
    const int* poseFrame;         /* [nPoses] CamPoseItem::f, list order; poseFrame[0] = m_camPos.first()->f */
    const double* poseR;          /* [nPoses][9] */
    const double* poseT;          /* [nPoses][3] */
    const int* featPtr;           /* [curFrame - poseFrame[0] + 2]: row r = frame poseFrame[0] + r owns entries
                                     [featPtr[r], featPtr[r+1]) of the two arrays below */
    const long long* featPointId; /* id of the (certain static) map point each feature point carries */
    const double* featXY;         /* [n][2] FeaturePoint::x, y */
} cs_export_cam;
/* nPts static map points in the order they are to be listed (the reference: address order of the objects, the address
 * being the id, :1862): ptId [nPts], ptM [nPts][3], ptCov [nPts][9].  covAsReference != 0: the 9 numbers of every point are
 * ptCov[k][k], k = 0..8 -- what the reference's shadowed loop variable makes it write (:1965-1966); 0: each point's own 9,
 * the "<3x3 covariance>" the reference's README.md:152-158 documents. */
int cs_export_results_v1(const char* dirPath, int nCams, const cs_export_cam* cams, int curFrame, int nPts, const long long* ptId,
                         const double* ptM, const double* ptCov, int covAsReference);

/* ------------------------------------------------------------------------------------------
 * Inter-camera NCC matching: blocks and the epipolar / NCC matrices of one camera pair
 * ------------------------------------------------------------------------------------------
 * Replaces NCCBlock::compute / computeScaled (src/slam/SL_NCCBlock.cpp:15-54), matchNCCBlock (:258-264) and getEpiNccMat
 * (src/slam/SL_FeatureMatching.cpp:3-46) -- the matrices NewMapPtsNCC::matchBetween
 * (src/app/SL_NewMapPointsInterCam.cpp:273-317) hands to its greedy matcher.
 * Block record: 128 bytes per feature, the 121 bytes of NCCBlock::I row by row, then 7 bytes of 0x80; abc: 4 doubles per
 * feature (NCCBlock::A, B, C, avgI); valid: compute()'s return value (0: the block would leave the image; its record is
 * 0x80 / zeros).  Feature positions are x[] and y[] in pixels of the FULL image; the block is cut at (int)(x * scale),
 * (int)(y * scale) of the small image (SingleSLAM::m_smallImg, m_smallScale = 0.3).
 * Matrices: M x N doubles, row = feature of camera 1.  Entry (i, j): e = epipolarError(F, p1_i, p2_j) (distance of p1_i
 * from the line F (p2_j, 1)); if e <= epiMax and matchNCCBlock >= nccMin: epiMat = e, nccMat = the score; else both
 * wNone (-1 in the reference).  A pair with a missing block is wNone. */
int cs_ncc_blocks_dev(int device, void* hip_stream, const unsigned char* d_img, int W, int H, int n, const double* d_x,
                      const double* d_y, double scale, unsigned char* d_blocks, double* d_abc, int* d_valid);
int cs_ncc_epi_mat_dev(int device, void* hip_stream, const double F[9] /* host */, int M, const double* d_x1, const double* d_y1,
                       const unsigned char* d_blocks1, const double* d_abc1, const int* d_valid1, int N, const double* d_x2,
                       const double* d_y2, const unsigned char* d_blocks2, const double* d_abc2, const int* d_valid2,
                       double epiMax, double nccMin, double wNone, double* d_epiMat, double* d_nccMat);
/* The same test of every pair, with only the pairs that PASS it written out: {i, j, epipolar error, NCC score} records through
 * one atomic counter (d_pairCount is zeroed by the call and counts every passing pair, also those beyond pairCap, which are
 * dropped; the order of the list is not defined).  The dense matrices above are this list scattered into wNone-filled arrays:
 * 64 MB of output per 2000 x 2000 camera pair, of which a matching run keeps a few dozen entries -- a device-resident caller
 * (or one that hands the greedy matcher a candidate list) wants this form.  CONSUMERS: the list's order changes from run to run;
 * anything that walks it in order (a greedy matcher) must first sort it by (score, i, j) -- cs_newpts_from_pairs_dev does -- or
 * scatter it; *d_pairCount > pairCap means pairs were dropped in no fixed order (cs_newpts_from_pairs_dev reports that as bit 0
 * of its flags; rerun with a larger pairCap or use the dense form). */
typedef struct cs_ncc_pair {
    int i, j;        /* feature of camera 1, of camera 2 */
    double epi, ncc; /* epiMat(i, j), nccMat(i, j) */
} cs_ncc_pair;
int cs_ncc_epi_pairs_dev(int device, void* hip_stream, const double F[9] /* host */, int M, const double* d_x1, const double* d_y1,
                         const unsigned char* d_blocks1, const double* d_abc1, const int* d_valid1, int N, const double* d_x2,
                         const double* d_y2, const unsigned char* d_blocks2, const double* d_abc2, const int* d_valid2,
                         double epiMax, double nccMin, cs_ncc_pair* d_pairs, int pairCap, int* d_pairCount);
/* A whole matching run (NewMapPtsNCC::run: matchBetween for the consecutive cameras of the group,
 * src/app/SL_NewMapPointsInterCam.cpp:150-158) in THREE launches instead of two per camera and one per pair:
 * cs_ncc_get_blocks_group_dev = getNCCBlocks of every camera (one resize launch, one cutter launch; `valid` is NOT written: it is
 * the caller's mask, e.g. cs_ncc_unmapped_mask_dev's), cs_ncc_epi_pairs_group_dev = cs_ncc_epi_pairs_dev of every camera pair. */
typedef struct cs_ncc_cam {
    const unsigned char* img; /* the full frame, W x H */
    const double* x;          /* n positions in full-image pixels */
    const double* y;
    unsigned char* scaled;    /* scratch for the resized image (cs_ncc_scaled_dims bytes) */
    unsigned char* blocks;    /* n x 128 out */
    double* abc;              /* n x 4 out */
    int* valid;               /* n: which features take part in the matrices (the caller's mask: read by cs_ncc_epi_pairs_group_dev) */
} cs_ncc_cam;
typedef struct cs_ncc_pair_job {
    double F[9];        /* fundamental matrix of (camA, camB) */
    int camA, camB;
    cs_ncc_pair* pairs; /* pairCap records out */
    int* count;         /* 1 out: zeroed by the call */
    const double* dF;   /* NULL, or the matrix in DEVICE memory (9 doubles), used instead of F: what cs_ncc_fmats_dev formed from the poses the
                         * frame has just solved (NewMapPtsNCC::matchBetween, src/app/SL_NewMapPointsInterCam.cpp:284-292) */
} cs_ncc_pair_job;
/* d_F [nPairs][9] <- F of the pairs (camA[k], camB[k]) from the cameras' current poses: x_A = R x_B + t, E = [t]x R, F = iK_A^T E iK_B
 * (formEMat / getFMat: our definitions).  camA / camB / d_iK (nCams pointers to 9 doubles): host arrays; <= 8 pairs. */
int cs_ncc_fmats_dev(int device, void* hip_stream, int nCams, int nPairs, const int* camA, const int* camB, const double* const* d_iK,
                     const double* d_R, const double* d_t, double* d_F);
int cs_ncc_get_blocks_group_dev(int device, void* hip_stream, int nCams, const cs_ncc_cam* cams /* host */, int W, int H, int n,
                                double scale);
int cs_ncc_epi_pairs_group_dev(int device, void* hip_stream, int nCams, const cs_ncc_cam* cams /* host */, int n, int nJobs,
                               const cs_ncc_pair_job* jobs /* host, <= 8 */, double epiMax, double nccMin, int pairCap);
/* How NewMapPtsNCC::matchBetween itself cuts its blocks: getNCCBlocks (src/slam/SL_NCCBlock.cpp:79-155, called at
 * src/app/SL_NewMapPointsInterCam.cpp:280-282 with the FULL image and blockScale 0.3) = cv::resize(img, Size(), scale, scale)
 * [INTER_LINEAR, 8-bit] once per image, then per point cv::getRectSubPix(small, 11 x 11, (x scale, y scale)) [8u -> 8u,
 * replicated border] and A, B, C.  OpenCV (un-vendored, version unpinned) is restated from its published generic C++ paths
 * (oracle/ncc_oracle.c says which); every point gets a block.  d_scaled: scratch for the resized image, cs_ncc_scaled_dims
 * bytes (unused for scale == 1.0, where the patches are cut from the image itself, :93-121); d_valid (may be NULL) is set to 1. */
/* valid[i] = slot i is an unmapped feature of this frame (hand-back: state 0 / 1, slot2map < 0) -- the features
 * NewMapPtsNCC::addSlam hands to matchBetween; n may span several cameras' records laid out back to back */
int cs_ncc_unmapped_mask_dev(int device, void* hip_stream, int n, const int* d_state, const int* d_slot2map, int* d_valid);
int cs_ncc_scaled_dims(int W, int H, double scale, int* Ws, int* Hs);
int cs_ncc_get_blocks_dev(int device, void* hip_stream, const unsigned char* d_img, int W, int H, int n, const double* d_x,
                          const double* d_y, double scale, unsigned char* d_scaled, unsigned char* d_blocks, double* d_abc, int* d_valid);
/* Host memory in and out, the whole stage for one camera pair (blocks of both cameras, then the matrices); the block
 * outputs (blocks / abc / valid) may be NULL.  cs_ncc_match_between cuts the blocks with NCCBlock::computeScaled from the SMALL
 * images it is given; cs_ncc_match_between_full is matchBetween's own path: FULL images, getNCCBlocks (scale <= 1). */
int cs_ncc_match_between_full(int device, const unsigned char* img1, int W1, int H1, int M, const double* x1, const double* y1,
                              const unsigned char* img2, int W2, int H2, int N, const double* x2, const double* y2, double scale,
                              const double F[9], double epiMax, double nccMin, double wNone, double* epiMat, double* nccMat,
                              unsigned char* blocks1, double* abc1, unsigned char* blocks2, double* abc2);
int cs_ncc_match_between(int device, const unsigned char* img1, int W1, int H1, int M, const double* x1, const double* y1,
                         const unsigned char* img2, int W2, int H2, int N, const double* x2, const double* y2, double scale,
                         const double F[9], double epiMax, double nccMin, double wNone, double* epiMat, double* nccMat,
                         unsigned char* blocks1, double* abc1, int* valid1, unsigned char* blocks2, double* abc2, int* valid2);

/* ------------------------------------------------------------------------------------------
 * New map points from the inter-camera NCC candidates (NewMapPtsNCC::run / output behind getEpiNccMat)
 * ------------------------------------------------------------------------------------------
 * Replaces, in src/app/SL_NewMapPointsInterCam.cpp: matchBetween's tail (:295-316: getSeedsBetween :97-127, getDisparityMat,
 * greedyGuidedNCCMatch / greedyNCCMatch), featTracksFromMatches (:631-690), reconstructTracks (:194-270) and output (:163-192,
 * decidePointType :22-93) -- two launches for all camera pairs and tracks.  In: per consecutive camera pair (a, a + 1) the candidate
 * list cs_ncc_epi_pairs_group_dev wrote (d_pairs[a], *d_pairCount[a] <= pairCap); cams (K, iK, xy, state, slot2map, isStatic;
 * reprojErr may be NULL): the frame's records; d_R / d_t: the cameras' current poses; the map as structure-of-arrays with capacity
 * mapCap of which *d_mapCount are in use.  Out: every track of >= minLen (2) views whose triangulation lies within maxRpErr (3.0)
 * pixels of each view and in front of each camera is APPENDED to the map in track order (position, covariance, flags: more than one
 * DYNAMIC feature CS_MAP_DYNAMIC, else CS_MAP_UNCERTAIN -- and decidePointType (:25-91, frame size W x H): an uncertain new point none of
 * whose features lies within 20 pixels of a feature of a CERTAIN dynamic point of this frame (this run's included) becomes certain
 * static, flags 0; newPt 1; firstFrame curFrame; its row of d_pointFeat), *d_mapCount grows, the
 * features' slot2map entries take the point (written through cams[c].slot2map), their reprojErr the pixel error.  Seeds = the map
 * points (not false, not uncertain) with a feature of this frame in both cameras, at most 512 per pair (map order); maxDisp 80.
 * d_counts [4 + nCams] or NULL: new points, tracks, tracks of >= minLen views, flags (bit 0: a pair had more than 2048 guided
 * candidates -- the rest were dropped; bit 1: the map is full; bit 2: more than 4096 dynamic features), then the matches of every pair.  d_scratch:
 * cs_newpts_scratch_bytes.  greedyNCCMatch, greedyGuidedNCCMatch, getDisparityMat are un-vendored LibVisualSLAM: OUR definitions
 * (csrc/newpts.hip).  NewMapPtsNCC's candidates are the features of this frame on tracks of more than three frames
 * that are unmapped or mapped to a FALSE point (addSlam, SL_NewMapPointsInterCam.h:103-131): cs_ncc_candidate_mask_dev writes that
 * mask (d_state / d_slot2map [nCams][N], d_trackSpan [nCams][2N]) for cs_ncc_epi_pairs_group_dev's `valid`. */
int cs_ncc_candidate_mask_dev(int device, void* hip_stream, int nCams, int N, const int* d_state, const int* d_slot2map,
                              const int* d_trackSpan, const unsigned char* d_mapFlags, int mapCap, int minTrack /* 3 */, int* d_valid,
                              size_t validStride /* ints from one camera's mask to the next; 0: N, back to back */);
size_t cs_newpts_scratch_bytes(int nCams, int N);
int cs_newpts_from_pairs_dev(int device, void* hip_stream, int nCams, int N, const cs_poseupdate_cam* cams,
                             const cs_ncc_pair* const* d_pairs /* host array [nCams - 1] */, const int* const* d_pairCount /* host array */,
                             int pairCap, const double* d_R, const double* d_t, double* d_mapPts, double* d_mapCov, unsigned char* d_mapFlags,
                             unsigned char* d_newPt, int* d_firstFrame, int* d_pointFeat, int mapCap, int* d_mapCount, int curFrame,
                             double maxDisp, double maxRpErr, double pixelErrVar, int minLen, int W, int H, void* d_scratch, int* d_counts);

/* ------------------------------------------------------------------------------------------
 * Robust multi-camera bundle adjustment
 * ------------------------------------------------------------------------------------------ */

#define CS_BA_FLAG_CHOL_FAILED 1    /* at least one LM step's reduced system could not be factorised (the step was rejected) */
#define CS_BA_FLAG_NO_PROGRESS 2    /* ... and NO step of the whole solve was accepted: the call returns CS_ERR_NUMERIC */
#define CS_BA_FLAG_SOLVER_TIMEOUT 4 /* the dataflow Cholesky gave up waiting for a block column: CS_ERR_NUMERIC */
typedef struct cs_ba_stats {
    double cost0, cost; /* sum of squared inlier residuals before / after */
    int nIterTotal, nOuter, nOutliers, flags; /* flags: CS_BA_FLAG_* */
} cs_ba_stats;

/* bundleAdjustRobust(int nCamsCon, vector<Mat_d>& Ks, vector<Mat_d>& Rs, vector<Mat_d>& Ts, int nPtsCon,
 *                    vector<Point3d>& pts, vector<vector<Meas2D>>& meas, double maxErr, int maxIter, int innerMaxIter)
 * -- external LibVisualSLAM geometry/SL_BundleAdjust.h; call sites src/app/SL_CoSLAMRobustBA.cpp:174,
 * src/app/SL_InterCamPoseEstimator.cpp:95, src/app/SL_MergeCameraGroup.cpp:646-647.
 * Flat form: Ks/Rs (C x 9, row-major), Ts (C x 3), pts (P x 3); meas[i] = measurements obs_ptr[i]..obs_ptr[i+1]
 * with obs_cam = Meas2D::viewId and obs_xy = (Meas2D::x, Meas2D::y).  Rs, Ts, pts are updated in place (host
 * memory); out_outlier[nObs] receives Meas2D::outlier (may be NULL); the first nCamsCon cameras and the first
 * nPtsCon points are held fixed.  Returns CS_OK or a negative CS_ERR_* code; CS_ERR_NUMERIC when the solver broke down
 * (stats->flags has CS_BA_FLAG_NO_PROGRESS or CS_BA_FLAG_SOLVER_TIMEOUT; the arrays are still written back -- unchanged in the
 * first case).  The C++ shim turns any error into an exception, which the reference's callers catch
 * (src/app/SL_CoSLAMRobustBA.cpp:173-179). */
int cs_ba_robust(int C, int P, int nObs, const double* Ks, double* Rs, double* Ts, double* pts, const int* obs_ptr,
                 const int* obs_cam, const double* obs_xy, int nCamsCon, int nPtsCon, double maxErr, int maxIter,
                 int innerMaxIter, int* out_outlier, cs_ba_stats* stats, int device);

/* Workspace form: buffers stay allocated between calls; *_dev leaves inputs and results in HBM. */
typedef struct cs_ba cs_ba;
cs_ba* cs_ba_create(int device);
void cs_ba_destroy(cs_ba* b);
/* the stream the host-pointer entry points, cs_ba_solve_dev(NULL stream) and the asynchronous worker enqueue on; and a way to
 * replace it by the caller's -- e.g. one confined to a CU range (cs_stream_create_cu_range) so that the solve's short
 * dependent kernels never wait behind the per-frame streams' workgroups.  The caller keeps ownership; NULL restores the
 * workspace's own.  Waits for queued asynchronous solves. */
void* cs_ba_stream(cs_ba* b);
int cs_ba_set_stream(cs_ba* b, void* hip_stream);
int cs_ba_robust_h(cs_ba* b, int C, int P, int nObs, const double* Ks, double* Rs, double* Ts, double* pts,
                   const int* obs_ptr, const int* obs_cam, const double* obs_xy, int nCamsCon, int nPtsCon,
                   double maxErr, int maxIter, int innerMaxIter, int* out_outlier, cs_ba_stats* stats);
/* upload the problem (host pointers) into the workspace without solving */
int cs_ba_upload(cs_ba* b, int C, int P, int nObs, const double* Ks, const double* Rs, const double* Ts,
                 const double* pts, const int* obs_ptr, const int* obs_cam, const double* obs_xy);
/* enqueue one full robust BA on hip_stream (NULL = the workspace's own stream) starting from the device-resident
 * initial estimate d_Rs0/d_Ts0/d_pts0; no host synchronisation */
int cs_ba_solve_dev(cs_ba* b, void* hip_stream, int C, int P, int nObs, const double* d_Rs0, const double* d_Ts0,
                    const double* d_pts0, int nCamsCon, int nPtsCon, double maxErr, int maxIter, int innerMaxIter);
/* The same solve the way the reference runs it -- on a worker thread next to tracking (src/app/SL_CoSLAM.cpp:1702-1784):
 * records an event on after_stream (the solve starts once the work enqueued there so far has finished), queues the
 * request for the workspace's own thread and returns.  That thread enqueues the schedule in chunks of LM steps and
 * stops at convergence, where cs_ba_solve_dev issues all maxIter x innerMaxIter steps up front.  Requests of one
 * workspace run in order; cs_ba_wait blocks until they are all done (cs_ba_download waits too). */
int cs_ba_solve_async(cs_ba* b, void* after_stream, int C, int P, int nObs, const double* d_Rs0, const double* d_Ts0,
                      const double* d_pts0, int nCamsCon, int nPtsCon, double maxErr, int maxIter, int innerMaxIter);
int cs_ba_wait(cs_ba* b);
/* Where the asynchronous solves of this workspace stand; neither call blocks.  cs_ba_pending: queued or running (0 = the last
 * result and its follow-up are complete on the device).  cs_ba_completed: solves finished since the workspace was created -- a
 * frame loop that runs ahead of the device always has the next solve queued, so it watches THIS count to learn that a result is
 * there and enqueues what the reference's BA thread does in output() under the lock it shares with the tracking thread
 * (src/app/SL_CoSLAM.cpp:1713-1720) on the stream that owns the map: cs_track_history_set_poses_dev,
 * cs_update_new_poses_points_dev. */
int cs_ba_pending(cs_ba* b);
long long cs_ba_completed(cs_ba* b);
/* The bundle adjuster's inputs built ON THE DEVICE from the tracker's own records: RobustBundleRTS::addKeyFrames / addPoints /
 * parseInputs (src/app/SL_CoSLAMRobustBA.cpp:37-78,109-165) fed by CoSLAM::requestForBA's walk over the last key frames
 * (src/app/SL_CoSLAM.cpp:1731-1784).  A window is a ring of nKeyFrames key frames x nCams cameras: per key frame and camera the
 * hand-back's records of that frame (undistorted pixels, which slot carries which map point) and K, R, t.  Cameras of the
 * problem: key frame (oldest first) x nCams + camera; points: every (static) map point with more than one feature point in the
 * window, in map-index order; measurements of a point in camera order -- the flattening parseInputs produces (pinned against
 * the reference's own parseInputs in tests/cxx/ref_ba_dropin_test.cpp).  The solve runs on the workspace's worker thread like
 * cs_ba_solve_async; the estimate it starts from is the key poses as pushed and the map as it stands. */
typedef struct cs_ba_window cs_ba_window;
cs_ba_window* cs_ba_window_create(int device, int nCams, int nKeyFrames, int N, int nMapPts);
void cs_ba_window_destroy(cs_ba_window* w); /* after cs_ba_wait() of every workspace that still has a request of this window queued */
/* cams: HOST array of nCams records whose xy / state / slot2map (device) are the hand-back's output of this frame;
 * d_K: nCams x 9, or one 9 shared by all cameras (kShared != 0); d_R nCams x 9, d_t nCams x 3.  Asynchronous on hip_stream.
 * A solve request (cs_ba_solve_window_async) fixes the window -- its ring slots -- and a snapshot of the map when it is MADE; the
 * ring holds two key frames more than a window, so the caller may push two key frames beyond a request whose parse has not run
 * yet; a third push waits (on the host) for the oldest outstanding parse. */
int cs_ba_window_push_dev(cs_ba_window* w, void* hip_stream, const cs_handback_cam* cams, const double* d_K, int kShared,
                          const double* d_R, const double* d_t, int frame);
int cs_ba_solve_window_async(cs_ba* b, cs_ba_window* w, void* after_stream, const double* d_mapPts, const unsigned char* d_mapStatic,
                             int nCamsCon, int nPtsCon, double maxErr, int maxIter, int innerMaxIter);
/* The same request with the map's CS_MAP_* flag bytes (cs_pose_update_frame_dev's d_mapFlags) instead of a 0 / 1 table: a point takes
 * part when it isLocalStatic() -- neither CS_MAP_DYNAMIC nor CS_MAP_FALSE (addPoints, src/app/SL_CoSLAMRobustBA.cpp:56-66). */
int cs_ba_solve_window_flags_async(cs_ba* b, cs_ba_window* w, void* after_stream, const double* d_mapPts, const unsigned char* d_mapFlags,
                                   int nCamsCon, int nPtsCon, double maxErr, int maxIter, int innerMaxIter);
/* RobustBundleRTS::output() (src/app/SL_CoSLAMRobustBA.cpp:273-316) for the window solves, in two halves.  The reference's BA thread
 * writes a finished adjustment back under the lock it shares with tracking (src/app/SL_CoSLAM.cpp:1713-1720).  Here
 *   (1) the solve's worker thread PACKS the result -- key poses, points, their map indices, which points have an outlier measurement
 *       -- into the next record of a small ring, behind the solve's last kernel on the solve's stream (cs_ba_output_attach), and
 *   (2) the stream that owns the map APPLIES a record between two frames (cs_ba_output_apply_dev): key poses into the pose
 *       history, the window's ring and the camera graphs' fixed nodes; points into the map, outlier points set false;
 *       constructCameraGraphs + updateNonKeyCameraPoses over the history's frames from the window's first key frame to the newest
 *       (cs_posegraph_*); the newest relaxed pose = the camera's current pose; updateNewPosesPoints over the live map.
 * Records are numbered in request order (0, 1, ...) over everything attached to the ring; a frame loop applies record k at a
 * frame of its own choosing (a fixed lag behind the key frame keeps a run reproducible) after cs_ba_output_wait(k).  A record is
 * self-contained plain memory of cs_ba_output_record_bytes bytes: with the cameras sharded over several GPUs the rank that solved
 * window k broadcasts it (cs_comm_broadcast_dev) and every rank applies the same bytes to its replica of the map.  A solve that
 * failed packs an empty record (header ok = 0), which applies nothing. */
typedef struct cs_ba_output cs_ba_output;
cs_ba_output* cs_ba_output_create(int device, int nCams, int nKeyFrames, int nMapPts, int nSlots);
void cs_ba_output_destroy(cs_ba_output* o);
int cs_ba_output_attach(cs_ba_output* o, cs_ba* b); /* o == NULL detaches; waits for b's queued solves */
size_t cs_ba_output_record_bytes(const cs_ba_output* o);
long long cs_ba_output_packed(cs_ba_output* o);      /* records complete on the device; never blocks */
int cs_ba_output_wait(cs_ba_output* o, long long seq, void** d_record); /* blocks until record seq is complete */
/* the same wait on the DEVICE: a polling lane enqueued on hip_stream; returns the record's address at once, what is enqueued on
 * hip_stream afterwards runs when the record is complete (the solve must be on another stream).  timeoutMs of GPU time (0 = 2000),
 * after which the stream goes on regardless and cs_ba_output_wait_errors (synchronises) counts it. */
int cs_ba_output_wait_dev(cs_ba_output* o, long long seq, void* hip_stream, int timeoutMs, void** d_record);
int cs_ba_output_wait_errors(cs_ba_output* o);
int cs_ba_output_slot(cs_ba_output* o, long long seq, void** d_record); /* the slot record seq uses (a broadcast's receive buffer) */
/* hdr8: C, P, nObs, nKf, nCams, seq, ok, 0; keyFrames[16]: frame of key frame j or -1.  Synchronises hip_stream. */
int cs_ba_output_header(cs_ba_output* o, const void* d_record, void* hip_stream, int hdr8[8], int keyFrames[16]);
int cs_ba_output_arrays(cs_ba_output* o, const void* d_record, const double** d_Rs, const double** d_Ts, const double** d_pts,
                        const int** d_pointMap, const unsigned char** d_ptOutlier);
/* h: the pose history (newest entry = the last frame whose pose update ran); w: the window whose ring copies of the key poses are
 * rewritten, or NULL; cams / d_pointFeat / the map: as cs_update_new_poses_points_dev takes them; the record's key frames are
 * firstKeyFrame + j * keyEvery, j < nKeyFrames; d_Rcur [nCams][9], d_tcur [nCams][3]: the cameras' current poses (rewritten with
 * the newest relaxed ones); d_counts [3] or NULL: static / dynamic points re-triangulated, points set false. */
int cs_ba_output_apply_dev(cs_ba_output* o, const void* d_record, void* hip_stream, cs_track_history* h, cs_ba_window* w,
                           const cs_poseupdate_cam* cams, const int* d_pointFeat, int nMap, double* d_mapPts, double* d_mapCov,
                           unsigned char* d_mapFlags, double pixelErrVar, int firstKeyFrame, int keyEvery, double* d_Rcur, double* d_tcur,
                           int* d_counts);
/* The same with the record's sequence number stated (the number cs_ba_output_wait / _wait_dev was asked for on the rank that solved
 * the window; -1: unchecked = cs_ba_output_apply_dev): the kernels compare it -- and the key frames' numbers -- with the record's header and
 * move NOTHING when the slot holds another window's record (a device-side wait that gave up leaves the record of nSlots solves
 * ago in place); such a refusal is counted in cs_ba_output_wait_errors. */
int cs_ba_output_apply_seq_dev(cs_ba_output* o, const void* d_record, long long seq, void* hip_stream, cs_track_history* h, cs_ba_window* w,
                               const cs_poseupdate_cam* cams, const int* d_pointFeat, int nMap, double* d_mapPts, double* d_mapCov,
                               unsigned char* d_mapFlags, double pixelErrVar, int firstKeyFrame, int keyEvery, double* d_Rcur,
                               double* d_tcur, int* d_counts);
/* The same for key frames that are NOT equally spaced -- the reference's fall where CoSLAM::genNewMapPoints' decision puts them
 * (src/app/SL_CoSLAM.cpp:1294-1346; cs_keyframe_ready_dev), and constructCameraGraphs fixes whichever frames the window's key poses
 * belong to (src/app/SL_CoSLAMRobustBA.cpp:182-229).  keyFrames: HOST array of nKeyFrames (= the output's key-frame count) frame
 * numbers, strictly ascending; node keyFrames[j] - keyFrames[0] of every camera's chain takes the record's pose j.  With seq >= 0 the
 * kernels compare EVERY key frame's number with the record's header (hdr[8 + j]).  The camera graphs are rebuilt on the host whenever
 * the spacing or the span differs from the previous apply's (cs_ba_output_apply_dev / _seq_dev are this call on
 * firstKeyFrame + j * keyEvery). */
int cs_ba_output_apply_frames_dev(cs_ba_output* o, const void* d_record, long long seq, void* hip_stream, cs_track_history* h, cs_ba_window* w,
                                  const cs_poseupdate_cam* cams, const int* d_pointFeat, int nMap, double* d_mapPts, double* d_mapCov,
                                  unsigned char* d_mapFlags, double pixelErrVar, const int* keyFrames, int nKeyFrames, double* d_Rcur,
                                  double* d_tcur, int* d_counts);
/* RobustBundleRTS::updateNewPosesPoints of every later apply over feature references (d_featRef [nMap][nCams] cs_feat_ref kept by
 * cs_feat_ref_advance_dev, d_refStatic [nMap][nCams] or NULL): stale features are views, the walks follow re-linked chains and
 * `mpt->lastFrame <= firstKeyFrame->f` (src/app/SL_CoSLAMRobustBA.cpp:250) is judged from the references' frames.  NULL: back to d_pointFeat. */
int cs_ba_output_set_feat_refs(cs_ba_output* o, const void* d_featRef, const unsigned char* d_refStatic);
/* diagnostic: which parts of output() an apply performs (default CS_BA_APPLY_ALL).  POSES: key poses into history / window ring +
 * relaxation of the non-key frames + the current poses; POINTS: adjusted points into the map; FALSE: points with an outlier
 * measurement set false; UPDATE: updateNewPosesPoints.  tools/r05_drift.py separates their effects on the closed loop with it. */
#define CS_BA_APPLY_POSES 1
#define CS_BA_APPLY_POINTS 2
#define CS_BA_APPLY_FALSE 4
#define CS_BA_APPLY_UPDATE 8
#define CS_BA_APPLY_ALL 15
int cs_ba_output_set_apply_mask(cs_ba_output* o, int mask);
/* InterCamPoseEstimator::addMapPoints + apply's solve (src/app/SL_InterCamPoseEstimator.cpp:18-95) with the problem built ON THE DEVICE
 * from the frame's own records: cameras = every camera's current pose, all free; static points = per camera chooseStaticFeatPts
 * (src/app/SL_SingleSLAM.cpp:345-397) run on the records as they stand -- per 40 x 40 block the first track with a map point, else
 * the first of the longest, among tracks whose newest feature is static (isStatic) or belongs to a certainly static map point --
 * and of its winners those with a map point (the point as the map holds it now, ONE measurement, held fixed); dynamic points = chooseDynamicFeatPts per camera (src/app/SL_SingleSLAM.cpp:398-447), the union of
 * their map points in map order, the first maxDyn + 1 of them, one measurement per camera with a feature of this frame
 * (d_pointFeat: the hand-back's nMap x nCams table; numVisCam = its non-negative entries per point).  d_mapFlags: CS_MAP_* bytes;
 * d_newPt: MapPoint::bNewPt.  The solve -- bundleAdjustRobust(0, ..., m_numStatic, ..., maxErr, maxIter, innerMaxIter) -- runs on
 * workspace b's worker thread like cs_ba_solve_async; the result stays in the workspace (cs_ba_result_buffers, cs_ba_download with
 * cs_ba_intercam_last_problem's sizes).  The reference never consumes it (CoSLAM::interCamPoseUpdate has no caller). */
typedef struct cs_intercam_cam {
    const double* K;     /* 9 */
    const double* xy;    /* 2N: the hand-back's undistorted pixels */
    const int* state;    /* N */
    const int* slot2map; /* N */
    const int* trackSpan;          /* 2N: first[N], last[N] frame of the slot's track (Track2D::length()) */
    const unsigned char* isStatic; /* N: FeaturePoint::type == TYPE_FEATPOINT_STATIC */
} cs_intercam_cam;
typedef struct cs_ba_intercam cs_ba_intercam;
cs_ba_intercam* cs_ba_intercam_create(int device, int nCams, int N, int ptsStride, int nMapPts, int maxDyn /* 60, :66 */);
void cs_ba_intercam_destroy(cs_ba_intercam* ic);
int cs_ba_solve_intercam_async(cs_ba* b, cs_ba_intercam* ic, void* after_stream, const cs_intercam_cam* cams /* host, nCams */, int W, int H,
                               int nColBlk, int nRowBlk, const double* d_R, const double* d_t, const double* d_mapPts,
                               const unsigned char* d_mapFlags, const unsigned char* d_newPt, const int* d_pointFeat, double maxErr,
                               int maxIter, int innerMaxIter);
int cs_ba_intercam_last_problem(cs_ba_intercam* ic, int* C, int* P, int* nObs, int* nStatic, const int** d_pointMap);
/* InterCamPoseEstimator::apply's write-back (src/app/SL_InterCamPoseEstimator.cpp:100-136; VERDICT r04 missing 5) behind a FINISHED solve
 * of ic on workspace b (cs_ba_wait): the solved poses become the cameras' current poses -- d_Rcur / d_tcur (nCams x 9 / 3), and the newest
 * frame of h when h != NULL --, then per camera the gate over its static mapped track nodes under the new pose: error < 2 -> reprojErr and
 * seqTriangulate, else reprojErr = the pixel distance and the point uncertain.  The loop is SingleSLAM::poseUpdate3D's own
 * (src/app/SL_SingleSLAM.cpp:677-706): cs_pose_update3d_dev's kernel, its arguments (cams: K, xy, state, slot2map, reprojErr; d_numNodes /
 * d_numOut [nCams] or NULL).  The reference's shipped loop never calls interCamPoseUpdate (src/gui/CoSLAMThread.cpp:95-130); neither do
 * the frame loops here -- an entry point for a caller that does. */
int cs_ba_intercam_apply_dev(cs_ba* b, cs_ba_intercam* ic, void* hip_stream, cs_track_history* h, const cs_poseupdate_cam* cams, int N,
                             const int* d_pointFeat, int nMap, double* d_Rcur, double* d_tcur, double* d_mapPts, double* d_mapCov,
                             unsigned char* d_mapFlags, double pixelErrVar, int* d_numNodes, int* d_numOut);
/* size and bind workspace b for the largest problem w can produce (cs_ba_solve_window_async does it on first use); afterwards
 * cs_ba_result_buffers' addresses stay put across the window's solves -- a follow-up record can be built before the first */
int cs_ba_reserve_for_window(cs_ba* b, cs_ba_window* w);
/* diagnostics: solves completed by the workspace's worker thread since the last call and the time they held its stream (GPU
 * clock, milliseconds, from the moment the work the solve waits for was done); the call resets the sums */
int cs_ba_worker_stats(cs_ba* b, int* jobs, double* gpu_ms_total, double* gpu_ms_last, double* gpu_ms_max,
                       double* gpu_ms_parse /* of gpu_ms_total: the window parses (cs_ba_solve_window_async), up to their host round trip */);
int cs_ba_window_last_problem(cs_ba_window* w, int* C, int* P, int* nObs, const int** d_pointMap, int* keyFrames);
int cs_ba_problem_buffers(cs_ba* b, const double** d_Ks, const int** d_obs_ptr, const int** d_obs_cam, const double** d_obs_xy);
/* Work that belongs right behind every solve of this workspace, on the solve's own stream and without a host round trip:
 * `fn(stream, user)` is called (from cs_ba_solve_dev's caller thread, or from the workspace's worker thread for
 * cs_ba_solve_async) once the solve's last kernel is enqueued; it enqueues more work on `stream` and returns CS_OK.  This is
 * where RobustBundleRTS::output()'s update of the non-key frames goes (src/app/SL_CoSLAMRobustBA.cpp:311-315): see
 * cs_posegraph_after_ba.  NULL removes it.  Waits for queued asynchronous solves first. */
typedef int (*cs_ba_followup_fn)(void* hip_stream, void* user);
int cs_ba_set_followup(cs_ba* b, cs_ba_followup_fn fn, void* user);
/* device addresses of the workspace's current estimate: Rs [C][9], Ts [C][3], pts [P][3] (valid until the next upload of a
 * larger problem) */
int cs_ba_result_buffers(cs_ba* b, double** d_Rs, double** d_Ts, double** d_pts);
/* synchronise and copy the workspace's current estimate back (any pointer may be NULL) */
/* Distributed solve (one process per GPU, points sliced by rank; SURVEY.md 8e collective 2): the reduced camera system
 * S || rhs is all-reduced once per LM step by the caller between the phases below; every launch is asynchronous on
 * `hip_stream`, the LM / outlier control flow stays on the device.  Schedule: see coslam_amd/multicam.py. */
enum {
    CS_BA_PH_COST0 = 0,        /* partial cost of the current estimate          -> all-reduce scal   */
    CS_BA_PH_CONTROL0 = 1,     /* start of an LM run                                                   */
    CS_BA_PH_LIN_SCHUR = 2,    /* linearise own points, partial S || rhs        -> all-reduce S||rhs  */
    CS_BA_PH_SOLVE_UPDATE = 3, /* solve, tentative step, partial tentative cost -> all-reduce scal   */
    CS_BA_PH_CONTROL1 = 4,     /* accept / reject, commit                                              */
    CS_BA_PH_FLAG = 5,         /* outlier flags of own measurements             -> all-reduce scal   */
    CS_BA_PH_OUTER_END = 6,
    CS_BA_PH_FINAL_PREP = 7,   /* zero foreign points / flags                   -> all-reduce pts, outlier */
    CS_BA_PH_FINISH = 8        /* final cost and statistics                                            */
};
int cs_ba_dist_begin(cs_ba* b, void* hip_stream, int C, int P, int nObs, const double* d_Rs0, const double* d_Ts0,
                     const double* d_pts0, int nCamsCon, int nPtsCon, double maxErr, int innerMaxIter, int pLo, int pHi,
                     int addLambda);
int cs_ba_dist_phase(cs_ba* b, void* hip_stream, int phase);
int cs_ba_dist_buffers(cs_ba* b, void** d_S_rhs, int* n_red, void** d_scal, void** d_pts, void** d_outlier);
int cs_ba_download(cs_ba* b, int C, int P, int nObs, double* Rs, double* Ts, double* pts, int* out_outlier,
                   cs_ba_stats* stats);

/* ------------------------------------------------------------------------------------------
 * Multi-GPU merge step: RCCL collectives over xGMI issued by the library (one process per GPU)
 * ------------------------------------------------------------------------------------------
 * The reference is one process; it reads every camera's features and pose directly (src/app/SL_CoSLAM.cpp:299-305,
 * src/app/SL_InterCamPoseEstimator.cpp:24-37).  With the cameras sharded over the GPUs these entry points carry the
 * same information: one all-gather per frame, and the joint bundleAdjustRobust sliced by points with one all-reduce of
 * S || rhs per LM step.  RCCL is loaded at run time (dlopen); without it the cs_comm_* calls fail and nothing else is
 * affected. */
typedef struct cs_comm cs_comm;
int cs_comm_available(void); /* 1: RCCL loaded with every entry point used here -- agree on it across the ranks BEFORE cs_comm_create: a rank
                                that cannot load the library would leave the others waiting inside ncclCommInitRank */
int cs_comm_unique_id(unsigned char id[128]); /* rank 0 creates it; the caller ships it to the other ranks */
cs_comm* cs_comm_create(const unsigned char id[128], int world, int rank, int device); /* ncclCommInitRank */
/* TEST TRANSPORT for ranks that share ONE GPU (RCCL refuses two ranks on a device): the same all-gather / broadcast entry points staged
 * through a POSIX shared-memory segment `name` (rank 0 creates it) -- stream synchronised, device -> segment, barrier, segment -> device.
 * It blocks the host; it exists so that a frame loop's N > 1 paths can run where one GPU is visible (what gloo is for the Python loop's
 * tests).  Never the transport of a measured number; cs_ba_dist_solve refuses it. */
cs_comm* cs_comm_create_host(const char* name, int world, int rank, int device);
void cs_comm_destroy(cs_comm* c);
int cs_comm_world(const cs_comm* c);
int cs_comm_rank(const cs_comm* c);

/* collective 1: every camera's {N x cs_klt_feature, R[9], t[3]} to every rank.  nCamsLocal cameras per rank (<= 16),
 * the same on every rank.  One kernel packs the rank's records, one ncclAllGather ships them; both on hip_stream. */
typedef struct cs_exchange cs_exchange;
cs_exchange* cs_exchange_create(cs_comm* c, int nCamsLocal, int nFeatures);
void cs_exchange_destroy(cs_exchange* x);
/* d_dests: HOST array of nCamsLocal device pointers to dest[]; d_R / d_t: device arrays 9 / 3 doubles per local camera */
int cs_exchange_allgather_dev(cs_exchange* x, void* hip_stream, const void* const* d_dests, const double* d_R,
                              const double* d_t);
/* gathered records: global camera g = rank * nCamsLocal + local index at d_recv + g * record_bytes */
int cs_exchange_buffers(cs_exchange* x, void** d_recv, size_t* record_bytes);

/* every gathered camera's pose into d_R [world * nCamsLocal][9] / d_t [..][3] (skipOwn != 0: all but this rank's own cameras) */
int cs_exchange_unpack_poses_dev(cs_exchange* x, void* hip_stream, double* d_R, double* d_t, int skipOwn);
/* collective 3: one buffer from rank `root` to every rank, in place (ncclBroadcast on hip_stream) -- a packed bundle-adjustment
 * result (cs_ba_output_*) from the rank that solved the window to every replica of the map.  World size 1: no-op. */
int cs_comm_broadcast_dev(cs_comm* c, void* hip_stream, void* d_buf, size_t bytes, int root);

/* every rank's `bytes` at d_send into every rank's d_recv (rank r at r * bytes): ncclAllGather on hip_stream */
int cs_comm_allgather_dev(cs_comm* c, void* hip_stream, const void* d_send, void* d_recv, size_t bytes);

/* collective 2: bundleAdjustRobust over all ranks of c.  Every rank uploads the same problem (cs_ba_upload) and calls
 * this with the same arguments; rank r linearises its contiguous slice of the points, S || rhs is all-reduced once per
 * LM step, every rank takes the same LM / outlier decisions on the device, points and flags are summed at the end.
 * Everything is enqueued on hip_stream (no host synchronisation); cs_ba_download reads the (replicated) result. */
int cs_ba_dist_solve(cs_ba* b, cs_comm* c, void* hip_stream, int C, int P, int nObs, const double* d_Rs0, const double* d_Ts0,
                     const double* d_pts0, int nCamsCon, int nPtsCon, double maxErr, int maxIter, int innerMaxIter);

#ifdef __cplusplus
}
#endif
#endif /* COSLAM_HIP_H */
