// results.cpp -- the result text files of a CoSLAM run (second half): what CoSLAM::exportResultsVer1 writes
// (src/app/SL_CoSLAM.cpp:1914-2028) from the structure-of-arrays the HIP path keeps.  Host code only -- no kernel, no device
// call; it lives in libcoslam_hip.so so that a caller holding SoA results needs nothing else to produce the files its tools read.
//
// Files, byte for byte as the reference's `ofstream <<` with default formatting (6 significant digits) writes them:
//   input_videos.txt   per camera: video path / the 9 entries of K / the 5 distortion coefficients / "W H"     (:1929-1943)
//   mappts.txt         count, then per static map point: id / "x y z" / 9 numbers                              (:1947-1969)
//   <c>_campose.txt    count, then per pose: frame in the video / the 9 entries of R and t on one line         (:1973-1990)
//   <c>_featpts.txt    frame count, then per frame from the camera's first pose to curFrame: "frame n" and the n static
//                      feature points as "id x y " triples on one line                                         (:1993-2027)
// The 9 numbers of a map point are NOT its covariance in the reference: the inner loop at :1965-1966 re-declares `i`, so every
// point's line is mapPoints[k]->cov[k], k = 0..8 -- entry k of the k-th point of the list (and a read past the list with
// fewer than 9 points).  covAsReference != 0 writes exactly that (0 where the reference would read past the list);
// covAsReference == 0 writes each point's own 9 entries.  Map point ids are the caller's: the reference uses the objects'
// addresses (:1862) and lists the points in address order.
#include <sys/stat.h>
#include <sys/types.h>

#include <cerrno>
#include <cstring>
#include <fstream>
#include <string>

#include "cs_common.h"

namespace {

bool open_out(std::ofstream& f, const std::string& path) {
    f.open(path.c_str());
    if (!f) {
        cs_set_error("cs_export_results_v1: cannot open '%s' to write: %s", path.c_str(), strerror(errno));
        return false;
    }
    return true;
}

}  // namespace

extern "C" int cs_export_results_v1(const char* dirPath, int nCams, const cs_export_cam* cams, int curFrame, int nPts,
                                    const long long* ptId, const double* ptM, const double* ptCov, int covAsReference) {
    if (!dirPath || nCams < 0 || (nCams > 0 && !cams) || nPts < 0 || (nPts > 0 && (!ptId || !ptM || !ptCov))) {
        cs_set_error("cs_export_results_v1: bad argument");
        return CS_ERR_INVALID;
    }
    for (int c = 0; c < nCams; ++c) {
        const cs_export_cam& q = cams[c];
        if (!q.videoFilePath || !q.K || !q.kc || q.nPoses < 1 || !q.poseFrame || !q.poseR || !q.poseT || !q.featPtr) {
            cs_set_error("cs_export_results_v1: camera %d: null pointer or no pose (the reference reads m_camPos.first())", c);
            return CS_ERR_INVALID;
        }
    }
    if (mkdir(dirPath, S_IRWXU | S_IRWXG | S_IROTH | S_IXOTH) != 0 && errno != EEXIST) {
        cs_set_error("cs_export_results_v1: cannot create '%s': %s", dirPath, strerror(errno));
        return CS_ERR_INVALID;
    }
    const std::string dir(dirPath);
    std::ofstream file;
    if (!open_out(file, dir + "/input_videos.txt")) return CS_ERR_INVALID;
    for (int c = 0; c < nCams; ++c) {
        file << cams[c].videoFilePath << std::endl;
        for (int i = 0; i < 9; ++i) file << cams[c].K[i] << " ";
        file << std::endl;
        for (int i = 0; i < 5; ++i) file << cams[c].kc[i] << " ";
        file << std::endl;
        file << cams[c].W << " " << cams[c].H << std::endl;
    }
    file.close();

    if (!open_out(file, dir + "/mappts.txt")) return CS_ERR_INVALID;
    file << (size_t)nPts << std::endl;
    for (int p = 0; p < nPts; ++p) {
        file << ptId[p] << std::endl;
        file << ptM[3 * p] << " " << ptM[3 * p + 1] << " " << ptM[3 * p + 2] << std::endl;
        for (int k = 0; k < 9; ++k) {
            if (covAsReference)
                file << (k < nPts ? ptCov[9 * (size_t)k + k] : 0.0) << " ";
            else
                file << ptCov[9 * (size_t)p + k] << " ";
        }
        file << std::endl;
    }
    file.close();

    for (int c = 0; c < nCams; ++c) {
        const cs_export_cam& q = cams[c];
        if (!open_out(file, dir + "/" + std::to_string(c) + "_campose.txt")) return CS_ERR_INVALID;
        file << (size_t)q.nPoses << std::endl;
        for (int i = 0; i < q.nPoses; ++i) {
            file << q.startFrameInVideo + q.poseFrame[i] + 1 << std::endl;  // getFrameInVideo (SL_CoSLAM.h:220-222)
            for (int k = 0; k < 9; ++k) file << q.poseR[9 * (size_t)i + k] << " ";
            file << q.poseT[3 * (size_t)i] << " " << q.poseT[3 * (size_t)i + 1] << " " << q.poseT[3 * (size_t)i + 2] << std::endl;
        }
        file.close();
    }
    for (int c = 0; c < nCams; ++c) {
        const cs_export_cam& q = cams[c];
        if (!open_out(file, dir + "/" + std::to_string(c) + "_featpts.txt")) return CS_ERR_INVALID;
        const int first = q.poseFrame[0];
        int nf = 0;
        for (int f = 0; f <= curFrame; ++f)
            if (f >= first) ++nf;
        file << nf << std::endl;
        for (int f = 0, row = 0; f <= curFrame; ++f) {
            if (f < first) continue;
            const int a = q.featPtr[row], b = q.featPtr[row + 1];
            ++row;
            if (b < a || (b > a && (!q.featPointId || !q.featXY))) {
                cs_set_error("cs_export_results_v1: camera %d: bad feature table at frame %d", c, f);
                return CS_ERR_INVALID;
            }
            file << q.startFrameInVideo + f + 1 << " " << (size_t)(b - a) << std::endl;
            for (int i = a; i < b; ++i) file << q.featPointId[i] << " " << q.featXY[2 * (size_t)i] << " " << q.featXY[2 * (size_t)i + 1] << " ";
            file << std::endl;
        }
        file.close();
    }
    return CS_OK;
}
