// -*- C++ -*-
// include/shim/CGKLT/v3d_gpuklt.h -- header-compatible replacement of the reference's
// src/tracking/CGKLT/v3d_gpuklt.h: the same V3D_GPU names, members and call signatures
// (KLT_TrackedFeature :166-176, KLT_SequenceTrackerConfig :180-199, KLT_SequenceTracker :202-294), implemented
// over the C-ABI of libcoslam_hip.so instead of OpenGL + Cg.  The reference's src/tracking/GPUKLT.cpp compiles
// against this header unchanged (it only uses the sequence tracker).  Put include/shim first on the include path.
#ifndef V3D_GPU_KLT_H
#define V3D_GPU_KLT_H

#include <stdexcept>
#include <string>

#include "coslam_hip.h"

namespace V3D_GPU {

typedef unsigned char uchar;

struct KLT_TrackedFeature {
    KLT_TrackedFeature() : status(-1), gain(1.0f), fed(-1) {}
    //! 0 means tracked from previous frame, 1 is newly created and -1 means invalidated track.
    int status;
    float pos[2];
    float gain;
    int fed;  // >=0 is the id of a feature point fed to tracking
};

struct KLT_SequenceTrackerConfig {
    KLT_SequenceTrackerConfig()
        : nIterations(12), nLevels(3), levelSkip(2), windowWidth(5), trackBorderMargin(4.0f),
          convergenceThreshold(0.1f), SSD_Threshold(5000.0f), trackWithGain(false), minDistance(8),
          minCornerness(1000.0f), detectBorderMargin(4.0f) {}
    int nIterations, nLevels, levelSkip, windowWidth;
    float trackBorderMargin, convergenceThreshold, SSD_Threshold;
    bool trackWithGain;
    int minDistance;
    float minCornerness, detectBorderMargin;
};

struct KLT_SequenceTracker {
    // COSLAM_HIP_DEVICE / COSLAM_HIP_TAP_MODE are optional compile-time knobs (GPU ordinal, decimation taps)
#ifndef COSLAM_HIP_DEVICE
#define COSLAM_HIP_DEVICE 0
#endif
#ifndef COSLAM_HIP_TAP_MODE
#define COSLAM_HIP_TAP_MODE 0
#endif
    KLT_SequenceTracker(KLT_SequenceTrackerConfig const& config) : _config(config), _h(0), _device(COSLAM_HIP_DEVICE) {}
    ~KLT_SequenceTracker() {
        if (_h) cs_klt_destroy(_h);
    }

    void setDevice(int device) { _device = device; }  // extension: which MI355X this camera lives on

    void allocate(int width, int height, int nLevels, int featuresWidth, int featuresHeight) {
        this->allocate(width, height, nLevels, featuresWidth, featuresHeight, 2 * featuresWidth, 2 * featuresHeight);
    }
    void allocate(int width, int height, int nLevels, int featuresWidth, int featuresHeight, int pointListWidth,
                  int pointListHeight) {
        if (!_h) {
            cs_klt_config c;
            c.nIterations = _config.nIterations;
            c.nLevels = _config.nLevels;
            c.levelSkip = _config.levelSkip;
            c.windowWidth = _config.windowWidth;
            c.trackBorderMargin = _config.trackBorderMargin;
            c.convergenceThreshold = _config.convergenceThreshold;
            c.SSD_Threshold = _config.SSD_Threshold;
            c.trackWithGain = _config.trackWithGain ? 1 : 0;
            c.minDistance = _config.minDistance;
            c.minCornerness = _config.minCornerness;
            c.detectBorderMargin = _config.detectBorderMargin;
            _h = cs_klt_create(&c, _device, COSLAM_HIP_TAP_MODE);
            if (!_h) throw std::runtime_error(std::string("KLT_SequenceTracker: ") + cs_last_error());
        }
        check(cs_klt_allocate(_h, width, height, nLevels, featuresWidth, featuresHeight, pointListWidth, pointListHeight));
    }
    void deallocate() {
        if (_h) check(cs_klt_deallocate(_h));
    }

    void setBorderMargin(float margin) { check(cs_klt_set_border_margin(need(), margin)); }
    void setConvergenceThreshold(float thr) { check(cs_klt_set_convergence_threshold(need(), thr)); }
    void setSSD_Threshold(float thr) { check(cs_klt_set_ssd_threshold(need(), thr)); }

    void detect(unsigned char const* image, int& nDetectedFeatures, KLT_TrackedFeature* dest) {
        check(cs_klt_detect(need(), image, &nDetectedFeatures, reinterpret_cast<cs_klt_feature*>(dest)));
    }
    // Add by Danping Zou for feeding custom feature points to track
    void detect(unsigned char const* image, int& nDetectedFeatures, KLT_TrackedFeature* dest, int nPresent,
                float* present) {
        check(cs_klt_detect_present(need(), image, &nDetectedFeatures, reinterpret_cast<cs_klt_feature*>(dest), nPresent,
                                    present));
    }
    void redetect(unsigned char const* image, int& nNewFeatures, KLT_TrackedFeature* dest) {
        check(cs_klt_redetect(need(), image, &nNewFeatures, reinterpret_cast<cs_klt_feature*>(dest)));
    }
    void feedExternFeaturePoints(int npts, float* featPts, int* trackIds, int& nFed) {
        check(cs_klt_feed(need(), npts, featPts, trackIds, &nFed));
    }
    void track(unsigned char const* image, int& nPresentFeatures, KLT_TrackedFeature* dest) {
        check(cs_klt_track(need(), image, &nPresentFeatures, reinterpret_cast<cs_klt_feature*>(dest)));
    }
    void advanceFrame() { check(cs_klt_advance(need())); }

    unsigned int getCurrentFrameTextureID() const { return 0; }  // there is no GL texture any more

    cs_klt* handle() { return _h; }  // extension: access to the *_dev entry points

protected:
    cs_klt* need() {
        if (!_h) throw std::runtime_error("KLT_SequenceTracker used before allocate()");
        return _h;
    }
    static void check(int rc) {
        if (rc != CS_OK) throw std::runtime_error(std::string("libcoslam_hip: ") + cs_last_error());
    }
    KLT_SequenceTrackerConfig const _config;
    cs_klt* _h;
    int _device;

private:
    KLT_SequenceTracker(KLT_SequenceTracker const&);
    KLT_SequenceTracker& operator=(KLT_SequenceTracker const&);
};

}  // namespace V3D_GPU

static_assert(sizeof(V3D_GPU::KLT_TrackedFeature) == sizeof(cs_klt_feature), "KLT_TrackedFeature layout");

#endif
