// -*- C++ -*-
// include/shim/tracking/GPUKLTGroup.h -- SURVEY 8f-1: the reference's GPUKLT objects of a camera rig driven by ONE set of
// launches per frame, their FeaturePoints / Track2D lists filled LAZILY.
//
// The reference tracks its cameras one after the other, and every GPUKLT::next (src/tracking/GPUKLT.cpp:144-161) ends in
// addToFeaturePoints (:36-60): a FeaturePoint and a Track2DNode newed per feature per frame (src/slam/SL_FeaturePoints.cpp:81-87,
// src/tracking/SL_Track2D.h:79-82), read-back and host loop on the frame's critical path.  Everything outside the hot path reads
// those lists directly -- SingleSLAM::chooseStaticFeatPts / getNumMappedStaticPts (src/app/SL_SingleSLAM.cpp:345-397,
// :125-140), propagateFeatureStates (:35-60), the GUI -- so the lists must exist WHEN somebody looks.  This adaptor keeps them
// exactly as GPUKLT.cpp would have left them, but builds them only then:
//
//     GPUKLT* klts[nCams] = {&slam[0].m_tracker, ...};       // the reference's own objects, init() / setIntrinsicParam() as before
//     FeaturePoints* ips[nCams] = {&slam[0].m_featPts, ...};
//     GPUKLTGroup rig(klts, ips, nCams);
//     rig.first(f0, images);                     // == every camera's GPUKLT::first: all cameras in one set of launches, enqueued
//     rig.next(images);  ...                     // == every camera's GPUKLT::next:  enqueued, returns at once, no list touched
//     rig.sync();                                // lists brought up to date: replays the frames not yet seen, in order
//     slam[c].chooseStaticFeatPts(v);            // the reference's own code reads m_tracker.m_tks[i] / m_featPts as it always did
//
// Underneath: cs_klt_hostview (coslam_amd/csrc/hostview.hip) = cs_klt_group_* + cs_klt_handback_dev on the device and a ring of
// per-frame {state, undistorted x, y} records in pinned host memory.  sync() must be called at least every `depth` frames.
// Compiled against the reference's own headers (tracking/GPUKLT.h over include/shim/CGKLT/v3d_gpuklt.h).
#ifndef COSLAM_GPUKLT_GROUP_H
#define COSLAM_GPUKLT_GROUP_H

#include <stdexcept>
#include <string>
#include <vector>

#include "tracking/GPUKLT.h"

#include "coslam_hip.h"

// GPUKLT keeps its sequence tracker protected (src/tracking/GPUKLT.h:29-31); a using-declaration in a derived class names the member,
// and a pointer to it applies to any GPUKLT -- the reference's class and the objects that embed it (SingleSLAM::m_tracker) stay as they are
struct GPUKLTPeek : GPUKLT {
    using GPUKLT::_tracker;
    static cs_klt* handle(GPUKLT& k) {
        V3D_GPU::KLT_SequenceTracker* t = k.*(&GPUKLTPeek::_tracker);
        return t ? t->handle() : 0;
    }
};

class GPUKLTGroup {
public:
    // klts: initialised (init + setIntrinsicParam) trackers of one rig -- same image size and configuration; ips[c]: the FeaturePoints
    // list camera c's features go to (SingleSLAM::m_featPts).  depth: frames the host ring holds between two sync() calls.
    GPUKLTGroup(GPUKLT* const* klts, FeaturePoints* const* ips, int nCams, int depth = 64, int device = 0)
        : _klts(klts, klts + nCams), _ips(ips, ips + nCams), _view(0), _seen(-1) {
        if (nCams < 1) throw std::runtime_error("GPUKLTGroup: no cameras");
        std::vector<cs_klt*> hs;
        std::vector<double> K, kud;
        for (int c = 0; c < nCams; ++c) {
            GPUKLT* k = _klts[c];
            if (!k || !GPUKLTPeek::handle(*k) || k->m_W != _klts[0]->m_W || k->m_H != _klts[0]->m_H || k->m_nMaxCorners != _klts[0]->m_nMaxCorners)
                throw std::runtime_error("GPUKLTGroup: every camera needs init() first, and the same image size and slot grid");
            hs.push_back(GPUKLTPeek::handle(*k));
            K.insert(K.end(), k->m_K.data, k->m_K.data + 9);
            kud.insert(kud.end(), k->m_kud.data, k->m_kud.data + 7);
        }
        _view = cs_klt_hostview_create(device, hs.data(), nCams, _klts[0]->m_W, _klts[0]->m_H, _klts[0]->m_nMaxCorners, K.data(), kud.data(),
                                       depth);
        if (!_view) throw std::runtime_error(std::string("GPUKLTGroup: ") + cs_last_error());
    }
    ~GPUKLTGroup() {
        if (_view) cs_klt_hostview_destroy(_view);
    }

    // where camera c's NEXT image may be decoded to (pinned; saves the copy first()/next() make of caller-owned buffers)
    unsigned char* imageBuffer(int c) { return cs_klt_hostview_image(_view, c); }

    // == GPUKLT::first(f, img, ips) for every camera (src/tracking/GPUKLT.cpp:113-121); images NULL: already in imageBuffer(c)
    void first(int f, const unsigned char* const* images) {
        check(cs_klt_hostview_frame(_view, images, f, 1));
        for (size_t c = 0; c < _klts.size(); ++c) _klts[c]->m_frame = f;
        _seen = f - 1;
    }
    // == GPUKLT::next(img, ips) for every camera (:144-161): redetect, addToFeaturePoints, advanceFrame -- enqueued, nothing read back
    void next(const unsigned char* const* images) {
        const int f = _klts[0]->m_frame + 1;
        check(cs_klt_hostview_frame(_view, images, f, 0));
        for (size_t c = 0; c < _klts.size(); ++c) _klts[c]->m_frame = f;
    }

    int pendingFrames() const { return cs_klt_hostview_newest(_view) - _seen; }

    // The lists as GPUKLT::addToFeaturePoints (:36-60) would have left them after the newest frame: every frame not yet replayed,
    // oldest first, slot by slot -- status >= 0: FeaturePoints::add(frame, camId, x, y), then Track2D::add (tracked) or clear + add
    // (new); status -1: clear; a feature the out >= W | H rule dropped (:47-48 `continue`) leaves its track as it was.
    // Returns the number of FeaturePoints appended.
    int sync() {
        const int newest = cs_klt_hostview_newest(_view);
        int added = 0;
        for (int f = _seen + 1; f <= newest; ++f) {
            const int* state;
            const double* xy;
            check(cs_klt_hostview_fetch(_view, f, &state, &xy));
            for (size_t c = 0; c < _klts.size(); ++c) {
                GPUKLT& k = *_klts[c];
                const int N = k.m_nMaxCorners;
                const int* st = state + c * N;
                const double *x = xy + c * 2 * N, *y = x + N;
                for (int i = 0; i < N; ++i) {
                    if (st[i] >= 0) {
                        FeaturePoint* p = _ips[c]->add(f, k.m_camId, x[i], y[i]);
                        if (st[i] != 0) k.m_tks[i].clear();
                        k.m_tks[i].add(p);
                        ++added;
                    } else if (st[i] == -1)
                        k.m_tks[i].clear();
                }
            }
            _seen = f;
        }
        return added;
    }

    cs_klt_hostview* view() { return _view; }

private:
    static void check(int rc) {
        if (rc != CS_OK) throw std::runtime_error(std::string("libcoslam_hip: ") + cs_last_error());
    }
    std::vector<GPUKLT*> _klts;
    std::vector<FeaturePoints*> _ips;
    cs_klt_hostview* _view;
    int _seen;
    GPUKLTGroup(const GPUKLTGroup&);
    GPUKLTGroup& operator=(const GPUKLTGroup&);
};

#endif
