/* epnp.h -- stand-in for the un-vendored EPnP solver src/slam/SL_SLAMHelper.cpp includes (solvePnP, :111-130): off the path of every
 * driver here (the file is compiled in place for getCamCenter / getCamDist / getViewAngleChange, :197-217); --gc-sections drops solvePnP.
 * TEST INFRASTRUCTURE. */
#pragma once
class epnp {
public:
    void set_internal_parameters(double, double, double, double) {}
    void set_maximum_number_of_correspondences(int) {}
    void reset_correspondences() {}
    void add_correspondence(double, double, double, double, double) {}
    double compute_pose(double R[3][3], double t[3]) {
        for (int i = 0; i < 3; ++i) {
            t[i] = 0;
            for (int j = 0; j < 3; ++j) R[i][j] = i == j;
        }
        return 0;
    }
};
