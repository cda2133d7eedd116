/* ref_shim/geometry/SL_Point.h -- stand-in (see math/SL_Matrix.h): the point types the reference derives from and
 * indexes both by name and as arrays (FeaturePoint : Point2d uses x, y, m; MapPoint : Point3dId uses x, y, z, M, id). */
#ifndef REF_SHIM_SL_POINT_H
#define REF_SHIM_SL_POINT_H
typedef long long longInt_pt;
class Point2d {
public:
    union {
        struct {
            double x, y;
        };
        double m[2];
    };
    Point2d() : x(0), y(0) {}
    Point2d(double a, double b) : x(a), y(b) {}
    void set(double a, double b) { x = a, y = b; }
};
class Point3d {
public:
    union {
        struct {
            double x, y, z;
        };
        double M[3];
    };
    Point3d() : x(0), y(0), z(0) {}
    Point3d(double a, double b, double c) : x(a), y(b), z(c) {}
    void set(double a, double b, double c) { x = a, y = b, z = c; }
};
class Point3dId : public Point3d {
public:
    long long id;
    Point3dId() : Point3d(), id(0) {}
    Point3dId(double a, double b, double c) : Point3d(a, b, c), id(0) {}
    Point3dId(double a, double b, double c, long long i) : Point3d(a, b, c), id(i) {}
};
#endif
