// ref_export_test.cpp -- the reference's OWN CoSLAM::exportResults against cs_export_results_v1 (result text files).
//
// oracle/Makefile compiles /root/reference/src/app/SL_CoSLAM.cpp IN PLACE (through a pipe that rewrites its three
// `pointer > 0` comparisons, which current compilers reject, to `!= 0` -- nothing else) against oracle/ref_shim/.  This driver
// fills a CoSLAM object through the reference's own containers (CamPoseList::add, FeaturePoints::add, MapPoint), lets the
// reference write $HOME/slam_results/<name>/ with HOME pointed at the work directory, writes the same run with
// cs_export_results_v1 from arrays gathered out of the same object, and compares the six files byte for byte.  No GPU involved.
//   ref_export_test <workdir>            compare; prints "ref_export_test: OK"
//   ref_export_test golden <workdir>     additionally dumps the arrays (<workdir>/inputs.bin) for tests/golden/make_golden.py
// TEST INFRASTRUCTURE; built into oracle/_ref/ where the reference tree exists.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include "app/SL_CoSLAM.h"

#include "coslam_hip.h"

#define CHECK(c)                                                         \
    do {                                                                 \
        if (!(c)) {                                                      \
            fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); \
            return 1;                                                    \
        }                                                                \
    } while (0)

static unsigned long long g_rng = 0xA0761D6478BD642Full;
static double urand() {
    g_rng ^= g_rng << 13;
    g_rng ^= g_rng >> 7;
    g_rng ^= g_rng << 17;
    return (double)(g_rng >> 11) / 9007199254740992.0;
}

static std::string slurp(const std::string& path) {
    std::ifstream f(path.c_str(), std::ios::binary);
    std::stringstream ss;
    ss << f.rdbuf();
    return ss.str();
}

struct CamArrays {
    std::vector<int> poseFrame, featPtr;
    std::vector<double> poseR, poseT, featXY;
    std::vector<long long> featId;
};

int main(int argc, char** argv) {
    const bool golden = argc >= 3 && !strcmp(argv[1], "golden");
    if (argc < (golden ? 3 : 2)) {
        fprintf(stderr, "usage: %s [golden] <workdir>\n", argv[0]);
        return 2;
    }
    const std::string work = argv[golden ? 2 : 1];
    setenv("HOME", work.c_str(), 1);

    CoSLAM* co = new CoSLAM();
    const int nCams = 2, curFrame = 12, firstFrame[2] = {0, 3};
    co->numCams = nCams;
    co->curFrame = curFrame;
    for (int c = 0; c < nCams; ++c) {
        SingleSLAM& s = co->slam[c];
        s.videoFilePath = c == 0 ? "/data/seq/cam 0.avi" : "/data/seq/cam1.avi";
        s.startFrameInVideo = 100 * c + 3;
        s.W = 640 + 16 * c, s.H = 480;
        s.K.resize(3, 3);
        s.K.fill(0);
        s.K.data[0] = 525.123456789 + c, s.K.data[4] = 524.5, s.K.data[2] = 319.5, s.K.data[5] = 239.25, s.K.data[8] = 1;
        s.k_c.resize(5, 1);
        const double kc[5] = {-0.2871, 0.09, 1.25e-5, -3e-4, 0};
        for (int i = 0; i < 5; ++i) s.k_c.data[i] = kc[i] * (1 + c);
        for (int f = firstFrame[c]; f <= curFrame; ++f) {
            double R[9], t[3];
            for (int i = 0; i < 9; ++i) R[i] = 2 * urand() - 1;  // the writer formats numbers, it does not care what they are
            R[0] = 1, R[4] = 1e-7 * urand(), R[8] = -123456.789 * urand();
            for (int i = 0; i < 3; ++i) t[i] = 100 * (2 * urand() - 1);
            s.m_camPos.add(f, c, R, t);
        }
    }
    // map points: static, dynamic, uncertain, false
    std::vector<MapPoint*> pts;
    for (int i = 0; i < 16; ++i) {
        MapPoint* p = new MapPoint(10 * urand() - 5, 10 * urand() - 5, 1e3 * urand());
        for (int k = 0; k < 9; ++k) p->cov[k] = 1e-3 * urand() * (1 + k);
        if (i % 7 == 3)
            p->setLocalDynamic();
        else
            p->setLocalStatic();
        if (i == 5) p->setUncertain();
        pts.push_back(p);
    }
    for (int c = 0; c < nCams; ++c)
        for (int f = firstFrame[c]; f <= curFrame; ++f) {
            const int n = f == 7 ? 0 : 3 + (int)(6 * urand());  // frame 7: no feature at all
            for (int i = 0; i < n; ++i) {
                FeaturePoint* fp = co->slam[c].m_featPts.add(f, c, 640 * urand(), 480 * urand());
                const double r = urand();
                fp->mpt = r < 0.25 ? 0 : pts[(int)(urand() * pts.size()) % pts.size()];
            }
        }
    co->exportResults("ref");
    const std::string refDir = work + "/slam_results/ref", ourDir = work + "/ours";

    // the same run as arrays
    std::vector<MapPoint*> listed;
    co->getAllStaticMapPoints(listed);  // the order and the ids (= addresses) the reference just used
    CHECK(listed.size() >= 9 && listed.size() < pts.size());
    std::vector<long long> ptId;
    std::vector<double> ptM, ptCov;
    for (size_t i = 0; i < listed.size(); ++i) {
        ptId.push_back(listed[i]->id);
        ptM.push_back(listed[i]->x), ptM.push_back(listed[i]->y), ptM.push_back(listed[i]->z);
        for (int k = 0; k < 9; ++k) ptCov.push_back(listed[i]->cov[k]);
    }
    std::vector<CamArrays> arr(nCams);
    std::vector<cs_export_cam> cams(nCams);
    for (int c = 0; c < nCams; ++c) {
        const SingleSLAM& s = co->slam[c];
        CamArrays& a = arr[c];
        for (CamPoseItem* cam = s.m_camPos.first(); cam; cam = cam->next) {
            a.poseFrame.push_back(cam->f);
            a.poseR.insert(a.poseR.end(), cam->R, cam->R + 9);
            a.poseT.insert(a.poseT.end(), cam->t, cam->t + 3);
        }
        a.featPtr.push_back(0);
        for (int f = a.poseFrame[0]; f <= curFrame; ++f) {
            std::vector<FeaturePoint*> v;
            s.m_featPts.getFrame(f, v);
            for (size_t i = 0; i < v.size(); ++i)
                if (v[i]->mpt && v[i]->mpt->isCertainStatic()) {
                    a.featId.push_back(v[i]->mpt->id);
                    a.featXY.push_back(v[i]->x), a.featXY.push_back(v[i]->y);
                }
            a.featPtr.push_back((int)a.featId.size());
        }
        cs_export_cam& q = cams[c];
        q.videoFilePath = s.videoFilePath.c_str();
        q.K = s.K.data, q.kc = s.k_c.data, q.W = s.W, q.H = s.H, q.startFrameInVideo = s.startFrameInVideo;
        q.nPoses = (int)a.poseFrame.size(), q.poseFrame = a.poseFrame.data(), q.poseR = a.poseR.data(), q.poseT = a.poseT.data();
        q.featPtr = a.featPtr.data(), q.featPointId = a.featId.data(), q.featXY = a.featXY.data();
    }
    int rc = cs_export_results_v1(ourDir.c_str(), nCams, cams.data(), curFrame, (int)listed.size(), ptId.data(), ptM.data(),
                                  ptCov.data(), 1);
    if (rc != CS_OK) {
        fprintf(stderr, "cs_export_results_v1: %s\n", cs_last_error());
        return 1;
    }
    const char* names[6] = {"input_videos.txt", "mappts.txt", "0_campose.txt", "1_campose.txt", "0_featpts.txt", "1_featpts.txt"};
    size_t total = 0;
    for (int i = 0; i < 6; ++i) {
        const std::string a = slurp(refDir + "/" + names[i]), b = slurp(ourDir + "/" + names[i]);
        if (a.empty() || a != b) {
            fprintf(stderr, "FAILED: %s differs (%zu vs %zu bytes)\n", names[i], a.size(), b.size());
            return 1;
        }
        total += a.size();
    }
    if (golden) {
        FILE* f = fopen((work + "/inputs.bin").c_str(), "wb");
        CHECK(f);
        const int hdr[4] = {nCams, curFrame, (int)listed.size(), 0};
        fwrite(hdr, sizeof(int), 4, f);
        fwrite(ptId.data(), sizeof(long long), ptId.size(), f);
        fwrite(ptM.data(), sizeof(double), ptM.size(), f);
        fwrite(ptCov.data(), sizeof(double), ptCov.size(), f);
        for (int c = 0; c < nCams; ++c) {
            const cs_export_cam& q = cams[c];
            const int n[4] = {q.W, q.H, q.startFrameInVideo, q.nPoses};
            const int pl = (int)strlen(q.videoFilePath), nf = (int)arr[c].featId.size(), np = (int)arr[c].featPtr.size();
            fwrite(&pl, sizeof(int), 1, f);
            fwrite(q.videoFilePath, 1, pl, f);
            fwrite(n, sizeof(int), 4, f);
            fwrite(q.K, sizeof(double), 9, f);
            fwrite(q.kc, sizeof(double), 5, f);
            fwrite(q.poseFrame, sizeof(int), q.nPoses, f);
            fwrite(q.poseR, sizeof(double), 9 * q.nPoses, f);
            fwrite(q.poseT, sizeof(double), 3 * q.nPoses, f);
            fwrite(&np, sizeof(int), 1, f);
            fwrite(q.featPtr, sizeof(int), np, f);
            fwrite(&nf, sizeof(int), 1, f);
            fwrite(q.featPointId, sizeof(long long), nf, f);
            fwrite(q.featXY, sizeof(double), 2 * nf, f);
        }
        fclose(f);
    }
    printf("ref_export_test: OK (6 files, %zu bytes, identical to the reference's exportResults; %zu static map points)\n", total,
           listed.size());
    return 0;
}
