/* ref_shim/matching/SL_Matching.h -- stand-in (see math/SL_Matrix.h) for LibVisualSLAM's list of feature matches, as
 * src/app/SL_NewMapPointsInterCam.cpp uses it: add(idx1, idx2, dist), clear(), reserve(), operator[] -> {idx1, idx2, dist}, num. */
#ifndef REF_SHIM_SL_MATCHING_H
#define REF_SHIM_SL_MATCHING_H
#include <vector>
struct MatchingItem {
    int idx1, idx2;
    double dist;
};
class Matching {
public:
    int num;
    std::vector<MatchingItem> data;
    Matching() : num(0) {}
    void clear() { data.clear(), num = 0; }
    void reserve(size_t n) { data.reserve(n); }
    void add(int i1, int i2, double d) {
        MatchingItem m = {i1, i2, d};
        data.push_back(m);
        num = (int)data.size();
    }
    const MatchingItem& operator[](int i) const { return data[i]; }
    MatchingItem& operator[](int i) { return data[i]; }
};
#endif
