// Compiles the reference-shaped C++ surface (include/shim) against libcoslam_hip.so and, when an MI355X is
// visible, runs one detect + intraCamEstimate + bundleAdjustRobust through it.  Without a GPU it checks that the
// calls fail loudly (exceptions carrying the C-ABI error text) instead of silently computing on the CPU.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

#include "CGKLT/v3d_gpuklt.h"
#include "geometry/SL_BundleAdjust.h"
#include "slam/SL_IntraCamPose.h"

// stand-ins for the LibVisualSLAM types, with the members CoSLAM uses (SL_CoSLAMRobustBA.cpp:90-92,128,153-154)
struct Mat_d {
    int rows, cols;
    std::vector<double> store;
    double* data;
    Mat_d(int r, int c, const double* d) : rows(r), cols(c), store(d, d + r * c), data(0) { data = store.data(); }
    Mat_d(const Mat_d& o) : rows(o.rows), cols(o.cols), store(o.store), data(0) { data = store.data(); }
};
struct Point3d {
    double M[3];
    Point3d(double x, double y, double z) {
        M[0] = x;
        M[1] = y;
        M[2] = z;
    }
};
struct Meas2D {
    int viewId;
    double x, y;
    int outlier;
    Meas2D(int v, double x_, double y_) : viewId(v), x(x_), y(y_), outlier(0) {}
};

int main() {
    V3D_GPU::KLT_SequenceTrackerConfig cfg;
    cfg.nLevels = 3;
    cfg.trackWithGain = true;
    cfg.minCornerness = 500.0f;
    const bool haveGpu = cs_device_count() > 0;
    printf("cs_version %d, devices %d\n", cs_version(), cs_device_count());

    const int W = 160, H = 120, fw = 8, fh = 8;
    std::vector<unsigned char> img(W * H, 100);
    for (int y = 40; y < 80; ++y)
        for (int x = 50; x < 110; ++x) img[y * W + x] = 200;  // a bright rectangle: four corners
    std::vector<V3D_GPU::KLT_TrackedFeature> feats(fw * fh);
    try {
        V3D_GPU::KLT_SequenceTracker trk(cfg);
        trk.allocate(W, H, cfg.nLevels, fw, fh);
        int n = 0;
        trk.detect(img.data(), n, feats.data());
        trk.advanceFrame();
        int n2 = 0;
        trk.redetect(img.data(), n2, feats.data());
        trk.advanceFrame();
        trk.deallocate();
        if (!haveGpu) {
            printf("FAIL: tracker ran without a GPU\n");
            return 1;
        }
        if (n < 4 || n2 < 4) {
            printf("FAIL: expected the rectangle corners, got %d / %d\n", n, n2);
            return 1;
        }
        printf("tracker: %d corners, %d after redetect\n", n, n2);
    } catch (const std::exception& e) {
        if (haveGpu) {
            printf("FAIL: %s\n", e.what());
            return 1;
        }
        printf("no GPU, tracker refused: %s\n", e.what());
    }

    // pose + BA on a tiny exact problem
    double K[9] = {500, 0, 320, 0, 500, 240, 0, 0, 1}, I3[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, t0[3] = {0.01, -0.02, 0.0};
    std::vector<double> Ms, ms;
    for (int i = 0; i < 20; ++i) {
        double X = -2 + 0.2 * i, Y = -1 + 0.13 * i, Z = 8 + 0.3 * (i % 5);
        Ms.push_back(X), Ms.push_back(Y), Ms.push_back(Z);
        ms.push_back(500 * X / Z + 320), ms.push_back(500 * Y / Z + 240);
    }
    double Ropt[9], topt[3];
    IntraCamPoseOption opt;
    try {
        bool ok = intraCamEstimate(K, I3, t0, 20, 0, Ms.data(), ms.data(), 10.0, Ropt, topt, &opt);
        if (!haveGpu) {
            printf("FAIL: pose ran without a GPU\n");
            return 1;
        }
        if (!ok || std::fabs(topt[0]) > 1e-5 || std::fabs(topt[1]) > 1e-5) {
            printf("FAIL: pose not recovered (%g %g %g)\n", topt[0], topt[1], topt[2]);
            return 1;
        }
        std::vector<Mat_d> Ks, Rs, Ts;
        for (int c = 0; c < 3; ++c) {
            double tc[3] = {-0.5 * c, 0, 0};
            Ks.push_back(Mat_d(3, 3, K));
            Rs.push_back(Mat_d(3, 3, I3));
            Ts.push_back(Mat_d(3, 1, tc));
        }
        std::vector<Point3d> pts;
        std::vector<std::vector<Meas2D> > meas;
        for (int i = 0; i < 20; ++i) {
            pts.push_back(Point3d(Ms[3 * i] + 0.01, Ms[3 * i + 1] - 0.01, Ms[3 * i + 2] + 0.05));
            std::vector<Meas2D> m;
            for (int c = 0; c < 3; ++c)
                m.push_back(Meas2D(c, 500 * (Ms[3 * i] - 0.5 * c) / Ms[3 * i + 2] + 320, 500 * Ms[3 * i + 1] / Ms[3 * i + 2] + 240));
            meas.push_back(m);
        }
        bundleAdjustRobust(3, Ks, Rs, Ts, 0, pts, meas, 6.0, 2, 20);
        double err = 0;
        for (int i = 0; i < 20; ++i)
            for (int q = 0; q < 3; ++q) err = std::fmax(err, std::fabs(pts[i].M[q] - Ms[3 * i + q]));
        if (err > 1e-6) {
            printf("FAIL: BA did not recover the points (%g)\n", err);
            return 1;
        }
        printf("pose + BA ok (max point error %.2e)\n", err);
    } catch (const std::exception& e) {
        if (haveGpu) {
            printf("FAIL: %s\n", e.what());
            return 1;
        }
        printf("no GPU, pose/BA refused: %s\n", e.what());
    }
    printf("shim ok\n");
    return 0;
}
