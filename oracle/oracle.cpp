// oracle.cpp -- CPU restatement (float64, C++17) of raytrace's per-pixel
// path-tracing loop.  TEST INFRASTRUCTURE ONLY: this file is the checker the HIP
// library is compared against and the CPU baseline timed beside it.  Nothing in
// raytrace_amd/ links, loads or calls it.
//
// Every function cites the reference file:line it follows (paths relative to the
// reference tree).  The arithmetic order of the reference is kept, and the file
// is built with -ffp-contract=off because the Go compiler does not fuse
// multiply-adds on amd64.
//
// Parity status
//   pinned   : Triangle / Quad / Cylinder / AABB / MicrofacetBRDF / r3 against
//              the reference's own table tests and // Output: examples
//              (tests/golden/*.json, tests/test_oracle_golden.py).
//   unpinned : everything that consumes random numbers.  The reference draws
//              from Go's math/rand (an additive lagged Fibonacci generator
//              whose 607-word seeding table lives in the Go standard library,
//              not in the reference tree, and no reference test fixes its
//              stream), reseeded per tile row (render.go:234).  Here each ray
//              owns a Philox4x32-10 stream keyed by (seed, pixel, sample,
//              position in the path tree); distributions are the reference's,
//              individual sample values are not.
//   deviation: NewBVH's equal-count fallback uses sort.Slice (unstable) in the
//              reference (shape_bvh.go:213); std::stable_sort is used here, so
//              shapes with exactly equal centroids may be ordered differently.

#include "../include/rt_hip.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <future>
#include <limits>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace {

constexpr double kEps = 1e-6;  // phys/math_extra.go:6
constexpr double kNaN = std::numeric_limits<double>::quiet_NaN();
constexpr double kInf = std::numeric_limits<double>::infinity();
constexpr double kMaxFloat = std::numeric_limits<double>::max();  // math.MaxFloat64
constexpr double kPi = 3.14159265358979323846264338327950288;     // math.Pi

// Go's math.Max / math.Min (NaN propagates, +Inf/-Inf win first, signed zeros ordered).
inline double go_max(double x, double y) {
    if (x == kInf || y == kInf) return kInf;
    if (std::isnan(x) || std::isnan(y)) return kNaN;
    if (x == 0 && x == y) return std::signbit(x) ? y : x;
    return x > y ? x : y;
}
inline double go_min(double x, double y) {
    if (x == -kInf || y == -kInf) return -kInf;
    if (std::isnan(x) || std::isnan(y)) return kNaN;
    if (x == 0 && x == y) return std::signbit(x) ? x : y;
    return x < y ? x : y;
}

// ---- r3 / r2 (r3/vec.go, r3/point.go, r3/mat.go, r2/point.go) --------------------

struct V3 {
    double x = 0, y = 0, z = 0;
    double get(int i) const { return i == 0 ? x : (i == 1 ? y : z); }  // vec.go:123-133
};
inline V3 add(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }        // vec.go:16
inline V3 sub(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }        // vec.go:21
inline V3 mul(V3 a, V3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }        // vec.go:26
inline V3 muls(V3 a, double s) { return {a.x * s, a.y * s, a.z * s}; }         // vec.go:36
inline V3 divs(V3 a, double s) { return {a.x / s, a.y / s, a.z / s}; }         // vec.go:41
inline double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }    // vec.go:48
inline V3 cross(V3 a, V3 b) {                                                   // vec.go:54
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
inline double length(V3 a) { return std::sqrt(dot(a, a)); }                     // vec.go:85
inline V3 unit(V3 a) {                                                          // vec.go:91-97
    double l = length(a);
    if (l == 0) return {0, 0, 0};
    return divs(a, l);
}
inline bool is_nan(V3 a) { return std::isnan(a.x) || std::isnan(a.y) || std::isnan(a.z); }
inline bool is_inf(V3 a) { return std::isinf(a.x) || std::isinf(a.y) || std::isinf(a.z); }
inline V3 lerp(V3 a, V3 b, double t) {  // vec.go:62-70 (t clamped)
    t = go_max(0, go_min(1, t));
    return {a.x + t * (b.x - a.x), a.y + t * (b.y - a.y), a.z + t * (b.z - a.z)};
}
inline V3 clip(V3 a, double lo, double hi) {  // vec.go:101-103
    return {go_min(go_max(a.x, lo), hi), go_min(go_max(a.y, lo), hi), go_min(go_max(a.z, lo), hi)};
}

struct V2 {
    double x = 0, y = 0;
};
inline V2 add2(V2 a, V2 b) { return {a.x + b.x, a.y + b.y}; }      // r2/point.go:17
inline V2 muls2(V2 a, double s) { return {a.x * s, a.y * s}; }     // r2/point.go:26
inline V2 lerp2(V2 a, V2 b, double t) {                             // r2/point.go:38-45
    t = go_max(0, go_min(1, t));
    return {(1 - t) * a.x + t * b.x, (1 - t) * a.y + t * b.y};
}
inline V2 clip2(V2 a, double lo, double hi) {                       // r2/point.go:64-66
    return {go_min(go_max(a.x, lo), hi), go_min(go_max(a.y, lo), hi)};
}

struct M3 {
    double m[3][3];
};
inline V3 mulvec(const M3& m, V3 v) {  // r3/mat.go:22-28
    return {m.m[0][0] * v.x + m.m[0][1] * v.y + m.m[0][2] * v.z,
            m.m[1][0] * v.x + m.m[1][1] * v.y + m.m[1][2] * v.z,
            m.m[2][0] * v.x + m.m[2][1] * v.y + m.m[2][2] * v.z};
}
inline M3 transpose(const M3& m) {  // r3/mat.go:46-54
    M3 r;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) r.m[i][j] = m.m[j][i];
    return r;
}

// ---- counter based random streams (replaces phys/rand.go:13-21) ------------------

inline void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1) {
    for (int round = 0; round < 10; round++) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
}

struct Rng {
    uint32_t k0 = 0, k1 = 0;         // seed
    uint32_t pixel = 0, sample = 0;  // which pixel sample
    uint32_t path = 1;               // position in the path tree: root 1, children 2p and 2p+1
    uint32_t n = 0;                  // draws made so far on this ray
    // Float64 in [0,1) with 53 random bits (math/rand Float64 has the same range).
    double float64() {
        uint32_t c[4] = {pixel, sample, path, n >> 1};
        philox4x32_10(c, k0, k1);
        uint64_t bits = (n & 1) ? ((uint64_t)c[3] << 32 | c[2]) : ((uint64_t)c[1] << 32 | c[0]);
        n++;
        return (double)(bits >> 11) * 0x1.0p-53;
    }
    Rng child(int which) const {
        Rng r = *this;
        r.path = path * 2u + (uint32_t)which;
        r.n = 0;
        return r;
    }
};

V3 in_unit_sphere(Rng& r) {  // rand.go:26-39
    for (;;) {
        V3 p{r.float64(), r.float64(), r.float64()};
        p = sub(muls(p, 2), V3{1, 1, 1});
        if (length(p) < 1.0) return p;
    }
}
V3 in_unit_disk(Rng& r) {  // rand.go:62-75
    for (;;) {
        V3 p{r.float64(), r.float64(), 0};
        p = sub(muls(p, 2), V3{1, 1, 0});
        if (dot(p, p) < 1.0) return p;
    }
}
V3 cosine_weighted_hemisphere(Rng& r, V3 normal) {  // rand.go:79-105
    double u1 = r.float64();
    double u2 = r.float64();
    double r1 = std::sqrt(u1);
    double theta = 2 * kPi * u2;
    double x = r1 * std::cos(theta);
    double y = r1 * std::sin(theta);
    double z = std::sqrt(1 - u1);
    V3 tangent;
    if (std::fabs(normal.x) > std::fabs(normal.y))
        tangent = unit(V3{-normal.z, 0, normal.x});
    else
        tangent = unit(V3{0, normal.z, -normal.y});
    V3 bitangent = cross(normal, tangent);
    return unit(add(add(muls(tangent, x), muls(bitangent, y)), muls(normal, z)));
}

// ---- ray, collision (render.go:22-35, shape.go:11-16) ----------------------------

struct Ray {
    V3 radiance{1, 1, 1};
    V3 origin, direction;
    int depth = 0;
    Rng rng;
    V3 at(double t) const { return add(origin, muls(direction, t)); }  // render.go:32-35
};

struct Collision {
    double t = 0;
    V3 at;
    V2 uv;
    V3 normal;
    int64_t prim = 0;  // face index inside a mesh (bookkeeping only; not in the reference)
};

struct Counters {
    uint64_t boxes = 0;      // AABB.hit calls
    uint64_t prims = 0;      // Face.Collide calls made from mesh leaves
    uint64_t mesh_rays = 0;  // Mesh.Collide calls
};
thread_local Counters* tl_counters = nullptr;

struct AABB {  // shape.go:27-96
    V3 mn, mx;
    double surface_area() const {  // shape.go:32-37
        double dx = mx.x - mn.x, dy = mx.y - mn.y, dz = mx.z - mn.z;
        return 2 * (dx * dy + dy * dz + dz * dx);
    }
    int longest_axis() const {  // shape.go:39-50
        double dx = mx.x - mn.x, dy = mx.y - mn.y, dz = mx.z - mn.z;
        if (dx > dy && dx > dz) return 0;
        if (dy > dz) return 1;
        return 2;
    }
    AABB unite(const AABB& o) const {  // shape.go:52-65
        return {{go_min(mn.x, o.mn.x), go_min(mn.y, o.mn.y), go_min(mn.z, o.mn.z)},
                {go_max(mx.x, o.mx.x), go_max(mx.y, o.mx.y), go_max(mx.z, o.mx.z)}};
    }
    V3 center() const { return {(mn.x + mx.x) * 0.5, (mn.y + mx.y) * 0.5, (mn.z + mx.z) * 0.5}; }  // shape.go:67-73
    bool intersects(const AABB& o) const {  // shape.go:75-79
        return mn.x <= o.mx.x && mx.x >= o.mn.x && mn.y <= o.mx.y && mx.y >= o.mn.y && mn.z <= o.mx.z &&
               mx.z >= o.mn.z;
    }
    bool hit(const Ray& r, double tmin, double tmax) const {  // shape.go:81-96
        if (tl_counters) tl_counters->boxes++;
        for (int axis = 0; axis < 3; axis++) {
            double invD = 1.0 / r.direction.get(axis);
            double t0 = (mn.get(axis) - r.origin.get(axis)) * invD;
            double t1 = (mx.get(axis) - r.origin.get(axis)) * invD;
            if (invD < 0.0) std::swap(t0, t1);
            tmin = go_max(t0, tmin);
            tmax = go_min(t1, tmax);
            if (tmax <= tmin) return false;
        }
        return true;
    }
};
inline AABB empty_box() { return {{kInf, kInf, kInf}, {-kInf, -kInf, -kInf}}; }

struct Shape {
    virtual ~Shape() = default;
    virtual bool collide(const Ray& r, double tmin, double tmax, Collision& c) const = 0;
    virtual AABB bounds() const = 0;
};

// Moller-Trumbore shared by Triangle / TriangleUV / Face; the three differ in the
// epsilon, the uv rule and the normal (shape_triangle.go:105-139,
// shape_triangle_uv.go:117-150, shape_mesh.go:95-142).
struct Tri : Shape {
    V3 p0, p1, p2;
    V2 uv0, uv1, uv2;
    V3 normal;
    int kind = RT_TRI_FACE;
    int64_t index = 0;
    bool collide(const Ray& r, double tmin, double tmax, Collision& c) const override {
        const double e = (kind == RT_TRI_FACE) ? 1e-8 : kEps;
        if (kind == RT_TRI_FACE && tl_counters) tl_counters->prims++;
        V3 edge1 = sub(p1, p0);
        V3 edge2 = sub(p2, p0);
        V3 h = cross(r.direction, edge2);
        double a = dot(edge1, h);
        if (a > -e && a < e) return false;
        double f = 1 / a;
        V3 s = sub(r.origin, p0);
        double u = f * dot(s, h);
        if (u < -e || u > 1.0 + e) return false;
        V3 q = cross(s, edge1);
        double v = f * dot(r.direction, q);
        if (v < -e || u + v > 1.0 + e) return false;
        double t = f * dot(edge2, q);
        if (t < tmin || t > tmax) return false;
        c.t = t;
        c.at = r.at(t);
        c.prim = index;
        if (kind == RT_TRI_TRIANGLE) {
            c.normal = unit(cross(edge1, edge2));
            c.uv = {u, v};
        } else if (kind == RT_TRI_TRIANGLE_UV) {
            c.uv = add2(lerp2(uv0, uv1, u), muls2(uv2, v));
            c.normal = unit(normal);
        } else {
            c.normal = unit(cross(edge1, edge2));
            double w = 1 - u - v;
            c.uv = add2(add2(muls2(uv0, w), muls2(uv1, u)), muls2(uv2, v));
        }
        return true;
    }
    AABB bounds() const override {  // shape_triangle.go:147-159, shape_mesh.go:145-161
        return {{go_min(go_min(p0.x, p1.x), p2.x), go_min(go_min(p0.y, p1.y), p2.y), go_min(go_min(p0.z, p1.z), p2.z)},
                {go_max(go_max(p0.x, p1.x), p2.x), go_max(go_max(p0.y, p1.y), p2.y), go_max(go_max(p0.z, p1.z), p2.z)}};
    }
};

V2 equirect_uv(V3 n, bool north_up) {  // shape_sphere.go:353-370
    double phi = std::atan2(n.z, n.x);
    if (phi < 0) phi += 2 * kPi;
    double u = phi / (2 * kPi);
    double yy = 1 < n.y ? 1 : n.y;  // min(1, n.Y) with the package's own generic min (render.go:116-121)
    yy = -1 > yy ? -1 : yy;         // max(-1, .)  (render.go:124-129)
    double theta = std::acos(yy);
    double v = north_up ? 1 - theta / kPi : theta / kPi;
    return clip2({u, v}, 0, 1);
}
V2 box_map_uv(V3 n) {  // shape_sphere.go:380-426
    double ax = std::fabs(n.x), ay = std::fabs(n.y), az = std::fabs(n.z);
    double max_axis, uc, vc;
    if (ax >= ay && ax >= az) {
        max_axis = ax;
        if (n.x >= 0) { uc = -n.z; vc = n.y; } else { uc = n.z; vc = n.y; }
    } else if (ay >= ax && ay >= az) {
        max_axis = ay;
        if (n.y >= 0) { uc = n.x; vc = -n.z; } else { uc = n.x; vc = n.z; }
    } else {
        max_axis = az;
        if (n.z >= 0) { uc = n.x; vc = n.y; } else { uc = -n.x; vc = n.y; }
    }
    double u = (uc / std::fabs(max_axis) + 1) / 2;
    double v = (vc / std::fabs(max_axis) + 1) / 2;
    return clip2({u, v}, 0, 1);
}

struct Sphere : Shape {  // shape_sphere.go:245-339
    V3 center;
    double radius = 0;
    int uvmap = 0;
    bool collide(const Ray& r, double tmin, double tmax, Collision& c) const override {
        V3 oc = sub(r.origin, center);
        double a = dot(r.direction, r.direction);
        double b = dot(oc, r.direction);
        double cc = dot(oc, oc) - radius * radius;
        double disc = b * b - a * cc;
        if (disc < 0) return false;
        double sq = std::sqrt(disc);
        double t = (-b - sq) / a;
        if (t < tmin || t > tmax) {
            t = (-b + sq) / a;
            if (t < tmin || t > tmax) return false;
        }
        c.t = t;
        c.at = r.at(t);
        c.normal = unit(sub(c.at, center));
        c.prim = 0;
        switch (uvmap) {
            case RT_UVMAP_EQUIRECT_SOUTH_UP: c.uv = equirect_uv(c.normal, false); break;
            case RT_UVMAP_BOX: c.uv = box_map_uv(c.normal); break;
            default: c.uv = equirect_uv(c.normal, true); break;
        }
        return true;
    }
    AABB bounds() const override {  // shape_sphere.go:274-280
        return {{center.x - radius, center.y - radius, center.z - radius},
                {center.x + radius, center.y + radius, center.z + radius}};
    }
};

struct Quad : Shape {  // shape_quad.go:15-152
    V3 center, normal;
    double width = 0, height = 0;
    void corners(V3& nrm, V3& p0, V3& p1, V3& p2, V3& p3) const {  // shape_quad.go:40-64
        nrm = unit(normal);
        V3 arbitrary = std::fabs(nrm.x) < 0.9 ? V3{1, 0, 0} : V3{0, 1, 0};
        V3 u = unit(cross(nrm, arbitrary));
        V3 v = unit(cross(nrm, u));
        u = muls(u, width / 2);
        v = muls(v, height / 2);
        p0 = sub(sub(center, u), v);
        p1 = sub(add(center, u), v);
        p2 = add(add(center, u), v);
        p3 = add(sub(center, u), v);
    }
    bool collide(const Ray& r, double tmin, double tmax, Collision& c) const override {
        V3 nrm, p0, p1, p2, p3;
        corners(nrm, p0, p1, p2, p3);
        Tri t1, t2;
        t1.kind = t2.kind = RT_TRI_TRIANGLE;
        t1.p0 = p0; t1.p1 = p1; t1.p2 = p2;
        t2.p0 = p0; t2.p1 = p2; t2.p2 = p3;
        Collision c1, c2;
        bool hit1 = t1.collide(r, tmin, tmax, c1);
        bool hit2 = t2.collide(r, tmin, tmax, c2);
        bool hit = false;
        if (hit1 && (!hit2 || c1.t < c2.t)) { hit = true; c = c1; }
        else if (hit2) { hit = true; c = c2; }
        if (hit) {  // shape_quad.go:85-109
            V3 local_u = sub(p2, p1);
            V3 local_v = sub(p0, p1);
            V3 hp = sub(c.at, p1);
            double uc = dot(hp, local_u) / dot(local_u, local_u);
            double vc = dot(hp, local_v) / dot(local_v, local_v);
            uc = go_max(0, go_min(1, uc));
            vc = go_max(0, go_min(1, vc));
            c.uv = {1 - uc, 1 - vc};
            c.normal = nrm;
            c.prim = 0;
        }
        return hit;
    }
    AABB bounds() const override {  // shape_quad.go:118-152
        V3 nrm, p0, p1, p2, p3;
        corners(nrm, p0, p1, p2, p3);
        return {{go_min(go_min(p0.x, p1.x), go_min(p2.x, p3.x)), go_min(go_min(p0.y, p1.y), go_min(p2.y, p3.y)),
                 go_min(go_min(p0.z, p1.z), go_min(p2.z, p3.z))},
                {go_max(go_max(p0.x, p1.x), go_max(p2.x, p3.x)), go_max(go_max(p0.y, p1.y), go_max(p2.y, p3.y)),
                 go_max(go_max(p0.z, p1.z), go_max(p2.z, p3.z))}};
    }
};

struct Cylinder : Shape {  // shape_cylinder.go:13-171
    V3 origin, direction;
    double radius = 0, height = 0;
    bool collide(const Ray& r, double tmin, double tmax, Collision& out) const override {
        V3 d = unit(direction);
        V3 oc = sub(r.origin, origin);
        double d_rd = dot(d, r.direction);
        double d_oc = dot(d, oc);
        V3 rd_perp = sub(r.direction, muls(d, d_rd));
        V3 oc_perp = sub(oc, muls(d, d_oc));
        double a = dot(rd_perp, rd_perp);
        double b = 2.0 * dot(rd_perp, oc_perp);
        double cv = dot(oc_perp, oc_perp) - radius * radius;
        double closest = kMaxFloat;
        Collision best;
        bool hit = false;
        if (a > kEps) {
            double disc = b * b - 4 * a * cv;
            if (disc >= 0) {
                double sq = std::sqrt(disc);
                double ts[2] = {(-b - sq) / (2.0 * a), (-b + sq) / (2.0 * a)};
                for (double t : ts) {
                    if (t < tmin || t > tmax) continue;
                    double y = d_oc + t * d_rd;
                    if (y >= 0 && y <= height) {
                        if (t < closest) {
                            V3 at = r.at(t);
                            V3 n = unit(sub(at, add(origin, muls(d, y))));
                            closest = t;
                            best = Collision{};
                            best.t = t; best.at = at; best.normal = n;  // uv stays (0,0)
                            hit = true;
                        }
                    }
                }
            }
        }
        struct Cap { V3 center, normal; };
        Cap caps[2] = {{origin, muls(d, -1)}, {add(origin, muls(d, height)), d}};
        for (const Cap& cap : caps) {
            double denom = dot(cap.normal, r.direction);
            if (std::fabs(denom) < kEps) continue;
            double t = dot(cap.normal, sub(cap.center, r.origin)) / denom;
            if (t < tmin || t > tmax) continue;
            V3 p = r.at(t);
            V3 pc = sub(p, cap.center);
            if (dot(pc, pc) <= radius * radius) {
                if (t < closest) {
                    closest = t;
                    best = Collision{};
                    best.t = t; best.at = p; best.normal = cap.normal; best.uv = {0.5, 0.5};
                    hit = true;
                }
            }
        }
        if (hit) out = best;
        return hit;
    }
    AABB bounds() const override {  // shape_cylinder.go:136-171
        V3 d = unit(direction);
        V3 orth = std::fabs(d.x) > std::fabs(d.y) ? unit(V3{-d.z, 0, d.x}) : unit(V3{0, d.z, -d.y});
        V3 u = orth, v = cross(d, u);
        bool first = true;
        V3 mn, mx;
        for (int i = 0; i <= 1; i++) {
            V3 base = add(origin, muls(d, (double)i * height));
            for (double theta = 0.0; theta < 2 * kPi; theta += kPi / 4) {
                V3 p = add(add(base, muls(u, radius * std::cos(theta))), muls(v, radius * std::sin(theta)));
                if (first) { mn = mx = p; first = false; continue; }
                mn = {go_min(mn.x, p.x), go_min(mn.y, p.y), go_min(mn.z, p.z)};
                mx = {go_max(mx.x, p.x), go_max(mx.y, p.y), go_max(mx.z, p.z)};
            }
        }
        return {mn, mx};
    }
};

struct Transform {  // transform.go:53-102
    V3 translation;
    M3 rotation;
    V3 scale;
    V3 apply_point(V3 p) const { return add(mulvec(rotation, mul(p, scale)), translation); }
    V3 apply_vector(V3 v) const { return mulvec(rotation, mul(v, scale)); }
    Transform inverse() const {
        V3 inv_scale{1 / scale.x, 1 / scale.y, 1 / scale.z};
        M3 inv_rot = transpose(rotation);
        V3 inv_tr = mul(mulvec(inv_rot, muls(translation, -1)), inv_scale);
        return {inv_tr, inv_rot, inv_scale};
    }
};

struct Transformed : Shape {  // shape_transformed_shape.go:12-95
    const Shape* child = nullptr;
    Transform tf;
    bool collide(const Ray& r, double tmin, double tmax, Collision& c) const override {
        Transform inv = tf.inverse();
        Ray local = r;
        local.origin = inv.apply_point(r.origin);
        local.direction = inv.apply_vector(r.direction);
        Collision lc;
        if (!child->collide(local, tmin, tmax, lc)) return false;
        c = lc;
        c.at = tf.apply_point(lc.at);
        c.normal = unit(tf.apply_vector(lc.normal));
        return true;
    }
    AABB bounds() const override {
        AABB b = child->bounds();
        V3 mn, mx;
        bool first = true;
        for (int k = 0; k < 8; k++) {
            V3 corner{(k & 1) ? b.mx.x : b.mn.x, (k & 2) ? b.mx.y : b.mn.y, (k & 4) ? b.mx.z : b.mn.z};
            V3 p = tf.apply_point(corner);
            if (first) { mn = mx = p; first = false; continue; }
            mn = {go_min(mn.x, p.x), go_min(mn.y, p.y), go_min(mn.z, p.z)};
            mx = {go_max(mx.x, p.x), go_max(mx.y, p.y), go_max(mx.z, p.z)};
        }
        return {mn, mx};
    }
};

// ---- BVH (shape_bvh.go) ---------------------------------------------------------

struct Group : Shape {  // shape_bvh.go:375-421
    std::vector<const Shape*> shapes;
    bool collide(const Ray& r, double tmin, double tmax, Collision& c) const override {
        bool any = false;
        double closest = tmax;
        for (const Shape* s : shapes) {
            Collision cc;
            if (s->collide(r, tmin, closest, cc)) {
                any = true;
                closest = cc.t;
                c = cc;
            }
        }
        return any;
    }
    AABB bounds() const override {
        AABB b = shapes[0]->bounds();
        for (size_t i = 1; i < shapes.size(); i++) b = b.unite(shapes[i]->bounds());
        return b;
    }
};

struct BVH : Shape {  // shape_bvh.go:15-78
    const Shape* left = nullptr;
    const Shape* right = nullptr;
    AABB box;
    std::unique_ptr<Shape> own_left, own_right;  // ownership only
    bool collide(const Ray& r, double tmin, double tmax, Collision& c) const override {
        if (!box.hit(r, tmin, tmax)) return false;
        bool hit_l = false, hit_r = false;
        Collision cl, cr;
        if (left) {
            hit_l = left->collide(r, tmin, tmax, cl);
            if (hit_l) tmax = go_min(tmax, cl.t);
        }
        if (right && right != left) hit_r = right->collide(r, tmin, tmax, cr);
        if (!hit_l && !hit_r) return false;
        if (hit_l && hit_r) {
            c = (cl.t < cr.t) ? cl : cr;
            return true;
        }
        c = hit_l ? cl : cr;
        return true;
    }
    AABB bounds() const override { return box; }
};

struct BvhStats {
    std::atomic<uint64_t> nodes{0}, leaves{0};
    std::atomic<uint32_t> max_depth{0}, max_leaf{0};
    void depth(uint32_t d) { uint32_t c = max_depth.load(); while (d > c && !max_depth.compare_exchange_weak(c, d)) {} }
    void leaf(uint32_t n) { uint32_t c = max_leaf.load(); while (n > c && !max_leaf.compare_exchange_weak(c, n)) {} }
};

// NewBVH, shape_bvh.go:81-272.
std::unique_ptr<BVH> new_bvh(std::vector<const Shape*> shapes, int depth, BvhStats& st) {
    constexpr int kMaxDepth = 32, kMinShapesPerLeaf = 4, kBins = 16;
    if (shapes.empty()) return nullptr;
    st.nodes++;
    st.depth((uint32_t)depth);
    if (shapes.size() == 1) {
        auto n = std::make_unique<BVH>();
        n->left = n->right = shapes[0];
        n->box = shapes[0]->bounds();
        st.leaves++;
        st.leaf(1);
        return n;
    }
    if (depth >= kMaxDepth || (int)shapes.size() <= kMinShapesPerLeaf) {
        auto g = std::make_unique<Group>();
        g->shapes = shapes;
        auto n = std::make_unique<BVH>();
        n->box = g->bounds();
        n->left = n->right = g.get();
        n->own_left = std::move(g);
        st.leaves++;
        st.leaf((uint32_t)shapes.size());
        return n;
    }
    AABB bbox = shapes[0]->bounds();
    for (size_t i = 1; i < shapes.size(); i++) bbox = bbox.unite(shapes[i]->bounds());
    int axis = bbox.longest_axis();

    struct Info { const Shape* shape; AABB b; double centroid; };
    std::vector<Info> infos(shapes.size());
    for (size_t i = 0; i < shapes.size(); i++) {
        AABB b = shapes[i]->bounds();
        infos[i] = {shapes[i], b, b.center().get(axis)};
    }
    const double lo = bbox.mn.get(axis), hi = bbox.mx.get(axis);
    auto bin_of = [&](double centroid) {
        double f = kBins * ((centroid - lo) / (hi - lo));
        // Go's int(NaN) is not a usable index (the reference would panic when the
        // longest extent is zero); such input goes to bin 0 here.
        int idx = std::isnan(f) ? 0 : (int)f;
        if (idx == kBins) idx = kBins - 1;
        return idx;
    };
    struct Bin { AABB b = empty_box(); int count = 0; } bins[kBins];
    for (const Info& si : infos) {
        int bi = bin_of(si.centroid);
        bins[bi].count++;
        bins[bi].b = bins[bi].b.unite(si.b);
    }
    int left_counts[kBins], right_counts[kBins];
    AABB left_bounds[kBins], right_bounds[kBins];
    {
        int count = 0;
        AABB b = empty_box();
        for (int i = 0; i < kBins; i++) { count += bins[i].count; b = b.unite(bins[i].b); left_counts[i] = count; left_bounds[i] = b; }
        count = 0;
        b = empty_box();
        for (int i = kBins - 1; i >= 0; i--) { count += bins[i].count; b = b.unite(bins[i].b); right_counts[i] = count; right_bounds[i] = b; }
    }
    double total_sa = bbox.surface_area();
    double best_cost = kMaxFloat;
    int best_split = -1;
    for (int i = 0; i < kBins - 1; i++) {
        double p_left = left_bounds[i].surface_area() / total_sa;
        double p_right = right_bounds[i + 1].surface_area() / total_sa;
        double cost = 1 + ((double)left_counts[i] * p_left + (double)right_counts[i + 1] * p_right);
        if (cost < best_cost) { best_cost = cost; best_split = i; }
    }
    std::vector<const Shape*> ls, rs;
    if (best_split == -1) {
        size_t mid = infos.size() / 2;
        std::stable_sort(infos.begin(), infos.end(), [](const Info& a, const Info& b) { return a.centroid < b.centroid; });
        for (size_t i = 0; i < mid; i++) ls.push_back(infos[i].shape);
        for (size_t i = mid; i < infos.size(); i++) rs.push_back(infos[i].shape);
    } else {
        for (const Info& si : infos) {
            if (bin_of(si.centroid) <= best_split) ls.push_back(si.shape);
            else rs.push_back(si.shape);
        }
    }
    auto n = std::make_unique<BVH>();
    n->box = bbox;
    std::unique_ptr<BVH> l, r;
    if (shapes.size() > 65536 && depth < 6) {  // the reference forks two goroutines per split (shape_bvh.go:249-261)
        auto fut = std::async(std::launch::async, [&] { return new_bvh(std::move(ls), depth + 1, st); });
        r = new_bvh(std::move(rs), depth + 1, st);
        l = fut.get();
    } else {
        l = new_bvh(std::move(ls), depth + 1, st);
        r = new_bvh(std::move(rs), depth + 1, st);
    }
    n->left = l.get();
    n->right = r.get();
    n->own_left = std::move(l);
    n->own_right = std::move(r);
    return n;
}

struct Mesh : Shape {  // shape_mesh.go:164-211
    std::vector<Tri> faces;
    std::unique_ptr<BVH> bvh;
    rt_bvh_info info{};
    bool collide(const Ray& r, double tmin, double tmax, Collision& c) const override {
        if (tl_counters) tl_counters->mesh_rays++;
        return bvh->collide(r, tmin, tmax, c);
    }
    AABB bounds() const override { return bvh->bounds(); }
};

// ---- textures (texture_*.go) ---------------------------------------------------------

struct Image {
    int w = 0, h = 0, format = 0;
    std::vector<uint8_t> px;
    // image.At(x,y).RGBA(): alpha-premultiplied 16-bit channels (Go image/color).
    void rgba16(int x, int y, uint32_t out[3]) const {
        if (format == RT_IMG_RGBA16) {
            const uint16_t* p = reinterpret_cast<const uint16_t*>(px.data()) + ((size_t)y * w + x) * 4;
            out[0] = p[0]; out[1] = p[1]; out[2] = p[2];
        } else {  // color.NRGBA.RGBA()
            const uint8_t* p = px.data() + ((size_t)y * w + x) * 4;
            uint32_t a = p[3];
            for (int k = 0; k < 3; k++) {
                uint32_t v = p[k];
                v |= v << 8;
                v *= a;
                v /= 0xff;
                out[k] = v;
            }
        }
    }
};

struct Texture {
    int kind = RT_TEX_UNIFORM;
    V3 color;
    const Texture* odd = nullptr;
    const Texture* even = nullptr;
    double frequency = 0;
    const Image* image = nullptr;
    int interp = 0, wrap = 0;
    V3 at(double u, double v) const {
        if (kind == RT_TEX_UNIFORM) return color;  // texture_uniform.go:18-20
        if (kind == RT_TEX_CHECKERBOARD) {         // texture_checkerboard.go:63-69
            double sines = std::sin(frequency * u) * std::sin(frequency * v);
            return sines < 0 ? odd->at(u, v) : even->at(u, v);
        }
        // texture_image.go:87-136
        if (wrap == RT_WRAP_CLAMP) {
            u = go_min(go_max(u, 0.0), 1.0);
            v = go_min(go_max(v, 0.0), 1.0);
        } else {
            u = u - std::floor(u);
            v = v - std::floor(v);
        }
        v = 1.0 - v;
        double x = u * (double)(image->w - 1);
        double y = v * (double)(image->h - 1);
        uint32_t c[3];
        if (interp == RT_INTERP_BILINEAR) {  // texture_image.go:139-188
            int x0 = (int)std::floor(x), x1 = x0 + 1;
            int y0 = (int)std::floor(y), y1 = y0 + 1;
            double fx = x - (double)x0, fy = y - (double)y0;
            auto cl = [](int a, int lo, int hi) { return a < lo ? lo : (a > hi ? hi : a); };
            uint32_t c00[3], c10[3], c01[3], c11[3];
            image->rgba16(cl(x0, 0, image->w - 1), cl(y0, 0, image->h - 1), c00);
            image->rgba16(cl(x1, 0, image->w - 1), cl(y0, 0, image->h - 1), c10);
            image->rgba16(cl(x0, 0, image->w - 1), cl(y1, 0, image->h - 1), c01);
            image->rgba16(cl(x1, 0, image->w - 1), cl(y1, 0, image->h - 1), c11);
            for (int k = 0; k < 3; k++) {
                double c0 = (double)c00[k] * (1 - fx) + (double)c10[k] * fx;
                double c1 = (double)c01[k] * (1 - fx) + (double)c11[k] * fx;
                double val = c0 * (1 - fy) + c1 * fy;
                c[k] = (uint16_t)val;  // NRGBA64 with A = 65535 returns the channel unchanged
            }
        } else {
            int ix = (int)std::round(x);  // math.Round: half away from zero
            int iy = (int)std::round(y);
            image->rgba16(ix, iy, c);
        }
        return {(double)c[0] / 65535.0, (double)c[1] / 65535.0, (double)c[2] / 65535.0};
    }
};

// ---- microfacet (microfacet.go:18-103) ----------------------------------------------

struct Microfacet {
    double roughness;
    V3 f0;
    double D(V3 h, V3 n) const {
        double rough = go_max(roughness, kEps);
        double ct = dot(n, h);
        if (ct <= 0) return 0;
        double m2 = rough * rough;
        double ct2 = ct * ct;
        double exponent = (ct2 - 1) / (m2 * ct2);
        return std::exp(exponent) / (kPi * m2 * ct2 * ct2);
    }
    double G1(V3 v, V3 n, V3 h) const {
        double cv = go_max(0, dot(n, v));
        double ch = go_max(0, dot(h, v));
        if (cv <= 0 || ch <= 0) return 0;
        double tan_v = std::sqrt(1 - cv * cv) / cv;
        double a = 1 / (roughness * tan_v);
        if (a >= 1.6) return 1;
        return (3.535 * a + 2.181 * a * a) / (1 + 2.276 * a + 2.577 * a * a);
    }
    double G(V3 wo, V3 wi, V3 n, V3 h) const { return G1(wo, n, h) * G1(wi, n, h); }
    V3 F(V3 wo, V3 h) const {
        double ct = go_max(0, dot(h, wo));
        double p5 = std::pow(1 - ct, 5);
        return add(f0, muls(sub(V3{1, 1, 1}, f0), p5));
    }
    V3 evaluate(V3 wo, V3 wi, V3 n) const {
        wo = unit(wo); wi = unit(wi); n = unit(n);
        V3 h = unit(add(wo, wi));
        double d = D(h, n);
        double g = G(wo, wi, n, h);
        V3 f = F(wo, h);
        double denom = 4 * go_max(0, dot(n, wo)) * go_max(0, dot(n, wi)) + kEps;
        return muls(f, d * g / denom);
    }
};

// ---- scene ------------------------------------------------------------------------------

struct Light { V3 position, intensity; };  // light.go:16-35

struct Material {
    int kind = 0;
    const Texture* texture = nullptr;
    V3 albedo;
    double fuzz = 0, ior_in = 1, ior_out = 1, roughness = 0;
};

struct Node {
    const Shape* shape = nullptr;
    const Material* material = nullptr;
};

struct Scene {
    std::vector<std::unique_ptr<Shape>> owned;
    std::vector<std::unique_ptr<Texture>> textures;
    std::vector<std::unique_ptr<Image>> images;
    std::vector<Material> materials;
    std::vector<Light> lights;
    std::vector<Node> nodes;
    std::vector<Mesh*> meshes;
};

struct Interaction {  // material.go:14-19
    Ray incoming;
    V3 outgoing;
    Collision collision;
    int node = -1;
};

double reflectance(double cos_theta, double ni, double nt) {  // material_dielectric.go:175-179
    double r0 = (ni - nt) / (ni + nt);
    r0 = r0 * r0;
    return r0 + (1 - r0) * std::pow(1 - cos_theta, 5);
}
V3 reflect_ray(V3 v, V3 n) { return unit(sub(v, muls(n, 2 * dot(v, n)))); }  // material_dielectric.go:182-184
bool refract(V3 v, V3 n, double ni_over_nt, V3& out) {  // material_dielectric.go:187-196
    V3 uv = unit(v);
    double dt = dot(uv, n);
    double disc = 1 - ni_over_nt * ni_over_nt * (1 - dt * dt);
    if (disc > 0) {
        out = sub(muls(sub(uv, muls(n, dt)), ni_over_nt), muls(n, std::sqrt(disc)));
        return true;
    }
    return false;
}

struct Resolution {
    Ray scattered[2];
    int n_scattered = 0;
    V3 emission;
};

Resolution resolve(const Material& m, Interaction& s) {
    Resolution res;
    const Collision& c = s.collision;
    switch (m.kind) {
        case RT_MAT_LAMBERTIAN: {  // material_lambertian.go:59-81
            V3 n = unit(c.normal);
            V3 dir = cosine_weighted_hemisphere(s.incoming.rng, n);
            V3 albedo = m.texture->at(c.uv.x, c.uv.y);
            Ray nr;
            nr.origin = c.at;
            nr.direction = dir;
            nr.depth = s.incoming.depth + 1;
            nr.radiance = mul(s.incoming.radiance, albedo);
            nr.rng = s.incoming.rng.child(0);
            res.scattered[res.n_scattered++] = nr;
            break;
        }
        case RT_MAT_METAL: {  // material_metal.go:28-48
            V3 reflected = reflect_ray(unit(s.incoming.direction), c.normal);
            V3 dir = add(reflected, muls(in_unit_sphere(s.incoming.rng), m.fuzz));
            if (dot(dir, c.normal) > 0) {
                Ray nr;
                nr.origin = c.at;
                nr.direction = unit(dir);
                nr.depth = s.incoming.depth + 1;
                nr.radiance = mul(s.incoming.radiance, m.albedo);
                nr.rng = s.incoming.rng.child(0);
                res.scattered[res.n_scattered++] = nr;
            }
            break;
        }
        case RT_MAT_DIELECTRIC: {  // material_dielectric.go:29-97
            V3 outward;
            double ni_over_nt, cosine, n1, n2;
            const V3 d = s.incoming.direction;
            if (dot(d, c.normal) > 0) {
                outward = muls(c.normal, -1);
                ni_over_nt = m.ior_in / m.ior_out;
                n1 = m.ior_in; n2 = m.ior_out;
                cosine = dot(d, c.normal) / length(d);
                cosine = std::sqrt(1 - ni_over_nt * ni_over_nt * (1 - cosine * cosine));
            } else {
                outward = c.normal;
                ni_over_nt = m.ior_out / m.ior_in;
                n1 = m.ior_out; n2 = m.ior_in;
                cosine = -dot(d, c.normal) / length(d);
            }
            V3 refracted;
            bool ok = refract(d, outward, ni_over_nt, refracted);
            double reflect_prob = 1.0;
            if (ok) {
                reflect_prob = reflectance(cosine, n1, n2);
                if (m.roughness > 0) refracted = unit(add(refracted, muls(in_unit_sphere(s.incoming.rng), m.roughness)));
                Ray tr;
                tr.origin = c.at;
                tr.direction = refracted;
                tr.depth = s.incoming.depth + 1;
                tr.radiance = muls(s.incoming.radiance, 1 - reflect_prob);
                tr.rng = s.incoming.rng.child(res.n_scattered);
                res.scattered[res.n_scattered++] = tr;
            }
            V3 reflected = reflect_ray(d, c.normal);
            if (m.roughness > 0) reflected = unit(add(reflected, muls(in_unit_sphere(s.incoming.rng), m.roughness)));
            Ray rr;
            rr.origin = c.at;
            rr.direction = reflected;
            rr.depth = s.incoming.depth + 1;
            rr.radiance = muls(s.incoming.radiance, reflect_prob);
            rr.rng = s.incoming.rng.child(res.n_scattered);
            res.scattered[res.n_scattered++] = rr;
            break;
        }
        case RT_MAT_EMITTER: {  // material_emitter.go:22-25
            V3 e = m.texture->at(c.uv.x, c.uv.y);
            res.emission = mul(e, s.incoming.radiance);
            break;
        }
        case RT_MAT_DEBUG_UV: {  // material_debug_uv.go:18-28
            if (c.uv.x < 0.0 || c.uv.x > 1.0) { res.emission = {1.0, 0.0, 0.0}; break; }
            double u = go_min(go_max(c.uv.x, 0.0), 1.0);
            double v = go_min(go_max(c.uv.y, 0.0), 1.0);
            res.emission = {u, v, 0.5};
            break;
        }
        case RT_MAT_DEBUG_NORMAL: {  // material_debug_normal.go:45-81
            V3 n = c.normal;
            double len = std::sqrt(n.x * n.x + n.y * n.y + n.z * n.z);
            if (!(len > 0)) len = 1;
            double r = 0.5 * (n.x / len + 1.0), g = 0.5 * (n.y / len + 1.0), b = 0.5 * (n.z / len + 1.0);
            auto cl = [](double q) { return q < 0 ? 0.0 : (q > 1 ? 1.0 : q); };
            res.emission = {cl(r), cl(g), cl(b)};
            break;
        }
    }
    return res;
}

struct Stats {
    uint64_t total = 0, exceeded = 0, left = 0, shadow = 0;
};

// Occlusion test common to the three lit materials (material_lambertian.go:34-45,
// material_metal.go:82-93, material_dielectric.go:146-157).
bool occluded(const Scene& sc, const Ray& shadow, double dist, int self_node, Stats& st) {
    st.shadow++;
    for (size_t i = 0; i < sc.nodes.size(); i++) {
        if (sc.nodes[i].shape == sc.nodes[self_node].shape) continue;
        Collision c;
        if (sc.nodes[i].shape->collide(shadow, kEps, dist, c)) return true;
    }
    return false;
}

V3 direct_lighting(const Scene& sc, const Material& m, const Interaction& s, Stats& st) {
    V3 direct{0, 0, 0};
    const Collision& c = s.collision;
    if (m.kind == RT_MAT_LAMBERTIAN) {  // material_lambertian.go:18-54
        V3 p = c.at;
        V3 n = unit(c.normal);
        V3 albedo = m.texture->at(c.uv.x, c.uv.y);
        for (const Light& l : sc.lights) {
            V3 dir = sub(l.position, p);  // light.go:29-35
            double dist = length(dir);
            dir = divs(dir, dist);
            Ray sh;
            sh.origin = add(p, muls(n, kEps));
            sh.direction = dir;
            sh.depth = s.incoming.depth + 1;
            if (!occluded(sc, sh, dist, s.node, st)) {
                double ndl = go_max(0, dot(n, dir));
                direct = add(direct, muls(mul(albedo, l.intensity), ndl));
            }
        }
    } else if (m.kind == RT_MAT_METAL) {  // material_metal.go:56-108
        V3 p = c.at;
        V3 n = unit(c.normal);
        V3 wo = unit(s.outgoing);
        Microfacet brdf{m.fuzz, m.albedo};
        for (const Light& l : sc.lights) {
            V3 dir = sub(l.position, p);
            double dist = length(dir);
            dir = divs(dir, dist);
            V3 wi = unit(dir);
            Ray sh;
            sh.origin = add(p, muls(n, kEps));
            sh.direction = wi;
            sh.depth = s.incoming.depth + 1;
            if (!occluded(sc, sh, dist, s.node, st)) {
                V3 val = brdf.evaluate(wo, wi, n);
                double ct = go_max(0, dot(n, wi));
                direct = add(direct, muls(mul(l.intensity, val), ct));
            }
        }
    } else if (m.kind == RT_MAT_DIELECTRIC) {  // material_dielectric.go:105-172
        V3 p = c.at;
        V3 n = unit(c.normal);
        V3 wo = unit(s.outgoing);
        bool outside = dot(wo, n) > 0;
        double eta_i = m.ior_out, eta_t = m.ior_in;
        V3 normal = n;
        if (!outside) { std::swap(eta_i, eta_t); normal = muls(n, -1); }
        Microfacet brdf{m.roughness, V3{1, 1, 1}};
        for (const Light& l : sc.lights) {
            V3 dir = sub(l.position, p);
            double dist = length(dir);
            dir = divs(dir, dist);
            V3 wi = unit(dir);
            double cos_i = go_max(0, dot(wi, normal));
            double fresnel = reflectance(cos_i, eta_i, eta_t);
            Ray sh;
            sh.origin = add(p, muls(normal, kEps));
            sh.direction = wi;
            sh.depth = s.incoming.depth + 1;
            if (!occluded(sc, sh, dist, s.node, st)) {
                V3 val = muls(brdf.evaluate(wo, wi, normal), fresnel);
                double ct = go_max(0, dot(normal, wi));
                direct = add(direct, muls(mul(l.intensity, val), ct));
            }
        }
    }
    return direct;  // Emitter / DebugUV / DebugNormal: zero
}

bool closest_hit(const Scene& sc, const Ray& r, double tmin, double tmax, Collision& best, int& node) {  // render.go:155-169
    double min_dist = tmax;
    bool hit = false;
    for (size_t i = 0; i < sc.nodes.size(); i++) {
        Collision c;
        if (sc.nodes[i].shape->collide(r, tmin, min_dist, c) && c.t < min_dist) {
            min_dist = c.t;
            best = c;
            node = (int)i;
            hit = true;
        }
    }
    return hit;
}

V3 trace_path(const Scene& sc, int max_depth, const Ray& r, Stats& st) {  // render.go:142-183
    st.total++;
    if (is_nan(r.origin) || is_inf(r.origin) || is_nan(r.direction) || is_inf(r.direction)) return {0, 0, 0};
    if (r.depth > max_depth) { st.exceeded++; return {0, 0, 0}; }
    Interaction si;
    if (!closest_hit(sc, r, kEps, kMaxFloat, si.collision, si.node)) { st.left++; return {0, 0, 0}; }
    si.incoming = r;
    si.outgoing = muls(r.direction, -1);
    const Material& m = *sc.nodes[si.node].material;
    Resolution res = resolve(m, si);
    V3 rgb = res.emission;
    rgb = add(rgb, direct_lighting(sc, m, si, st));
    for (int i = 0; i < res.n_scattered; i++) rgb = add(rgb, trace_path(sc, max_depth, res.scattered[i], st));
    return rgb;
}

// ---- cameras --------------------------------------------------------------------------------

void undistort_normalized(const rt_camera_desc& ci, double xd, double yd, double& x, double& y) {  // camera_calibrated.go:172-199
    x = xd; y = yd;
    for (int i = 0; i < 8; i++) {
        double r2 = x * x + y * y, r4 = r2 * r2, r6 = r4 * r2;
        double num = 1.0 + ci.k1 * r2 + ci.k2 * r4 + ci.k3 * r6;
        double den = 1.0 + ci.k4 * r2 + ci.k5 * r4 + ci.k6 * r6;
        if (den == 0) den = 1;
        double radial = num / den;
        double dx = 2.0 * ci.p1 * x * y + ci.p2 * (r2 + 2.0 * x * x);
        double dy = ci.p1 * (r2 + 2.0 * y * y) + 2.0 * ci.p2 * x * y;
        x = (xd - dx) / radial;
        y = (yd - dy) / radial;
    }
}

inline V3 v3(const double* p) { return {p[0], p[1], p[2]}; }

Ray cast(const rt_camera_desc& cam, double s, double t, Rng rng) {
    Ray r;
    r.rng = rng;
    if (cam.kind == RT_CAM_PINHOLE) {  // camera_pinhole.go:20-34
        V3 h = muls(v3(cam.horizontal), s);
        V3 v = muls(v3(cam.vertical), t);
        V3 ip = add(add(v3(cam.lower_left_corner), h), v);
        r.origin = v3(cam.origin);
        r.direction = unit(sub(ip, v3(cam.origin)));
        return r;
    }
    V3 from = v3(cam.look_from), at = v3(cam.look_at);
    V3 w = unit(sub(from, at));
    V3 u = unit(cross(v3(cam.vup), w));
    V3 v = cross(w, u);
    if (cam.kind == RT_CAM_ORTHOGRAPHIC) {  // camera_orthographic.go:24-46
        r.origin = add(add(from, muls(u, cam.fov_width * (s - 0.5))), muls(v, cam.fov_height * (t - 0.5)));
        r.direction = unit(sub(at, from));
    } else if (cam.kind == RT_CAM_FOCUSABLE) {  // camera_focusable.go:26-63
        V3 horizontal = muls(u, cam.fov_width * cam.working_distance);
        V3 vertical = muls(v, cam.fov_height * cam.working_distance);
        V3 llc = sub(sub(sub(from, divs(horizontal, 2)), divs(vertical, 2)), muls(w, cam.working_distance));
        double lens_radius = cam.aperture / 2;
        V3 rd = muls(in_unit_disk(r.rng), lens_radius);
        V3 offset = add(muls(u, rd.x), muls(v, rd.y));
        r.origin = add(from, offset);
        V3 ip = add(add(llc, muls(horizontal, s)), muls(vertical, t));
        r.direction = unit(sub(ip, r.origin));
    } else {  // camera_calibrated.go:42-79
        double upix = s * (double)cam.width, vpix = t * (double)cam.height;
        double xd = (upix - cam.cx) / cam.fx, yd = (vpix - cam.cy) / cam.fy;
        double x, y;
        undistort_normalized(cam, xd, yd, x, y);
        V3 dc = unit(V3{x, -y, -1.0});
        r.origin = from;
        r.direction = unit(add(add(muls(u, dc.x), muls(v, dc.y)), muls(w, dc.z)));
    }
    return r;
}

// ---- scene construction from the flat description -------------------------------------

std::string g_error;

const Shape* resolve_shape(Scene& sc, const rt_scene_desc& d, int kind, int index,
                           std::vector<std::vector<const Shape*>>& cache, int depth);

const Shape* make_shape(Scene& sc, const rt_scene_desc& d, int kind, int index,
                        std::vector<std::vector<const Shape*>>& cache, int depth) {
    switch (kind) {
        case RT_SHAPE_SPHERE: {
            auto s = std::make_unique<Sphere>();
            s->center = v3(d.spheres[index].center);
            s->radius = d.spheres[index].radius;
            s->uvmap = d.spheres[index].uvmap;
            sc.owned.push_back(std::move(s));
            break;
        }
        case RT_SHAPE_QUAD: {
            auto s = std::make_unique<Quad>();
            s->center = v3(d.quads[index].center);
            s->normal = v3(d.quads[index].normal);
            s->width = d.quads[index].width;
            s->height = d.quads[index].height;
            sc.owned.push_back(std::move(s));
            break;
        }
        case RT_SHAPE_CYLINDER: {
            auto s = std::make_unique<Cylinder>();
            s->origin = v3(d.cylinders[index].origin);
            s->direction = v3(d.cylinders[index].direction);
            s->radius = d.cylinders[index].radius;
            s->height = d.cylinders[index].height;
            sc.owned.push_back(std::move(s));
            break;
        }
        case RT_SHAPE_TRIANGLE: {
            const rt_triangle& t = d.triangles[index];
            auto s = std::make_unique<Tri>();
            s->p0 = v3(t.p[0]); s->p1 = v3(t.p[1]); s->p2 = v3(t.p[2]);
            s->uv0 = {t.uv[0][0], t.uv[0][1]}; s->uv1 = {t.uv[1][0], t.uv[1][1]}; s->uv2 = {t.uv[2][0], t.uv[2][1]};
            s->normal = v3(t.normal);
            s->kind = t.kind;
            sc.owned.push_back(std::move(s));
            break;
        }
        case RT_SHAPE_MESH: {
            const rt_mesh& m = d.meshes[index];
            auto s = std::make_unique<Mesh>();
            s->faces.resize(m.n_faces);
            std::vector<const Shape*> ptrs(m.n_faces);
            for (uint64_t i = 0; i < m.n_faces; i++) {
                const rt_face& f = d.faces[m.first_face + i];
                Tri& t = s->faces[i];
                t.kind = RT_TRI_FACE;
                t.index = (int64_t)i;
                t.p0 = {f.v[0][0], f.v[0][1], f.v[0][2]}; t.uv0 = {f.v[0][3], f.v[0][4]};
                t.p1 = {f.v[1][0], f.v[1][1], f.v[1][2]}; t.uv1 = {f.v[1][3], f.v[1][4]};
                t.p2 = {f.v[2][0], f.v[2][1], f.v[2][2]}; t.uv2 = {f.v[2][3], f.v[2][4]};
                ptrs[i] = &t;
            }
            BvhStats st;
            auto t0 = std::chrono::steady_clock::now();
            s->bvh = new_bvh(std::move(ptrs), 0, st);  // NewMesh, shape_mesh.go:170-181
            s->info.build_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            s->info.n_faces = m.n_faces;
            s->info.n_nodes = st.nodes;
            s->info.n_leaves = st.leaves;
            s->info.max_depth = st.max_depth;
            s->info.max_leaf_size = st.max_leaf;
            sc.meshes[index] = s.get();
            sc.owned.push_back(std::move(s));
            break;
        }
        case RT_SHAPE_TRANSFORMED: {
            const rt_transformed& t = d.transformed[index];
            if (depth > 16) return nullptr;
            const Shape* child = resolve_shape(sc, d, t.child_kind, t.child_index, cache, depth + 1);
            if (!child) return nullptr;
            auto s = std::make_unique<Transformed>();
            s->child = child;
            s->tf.translation = v3(t.translation);
            std::memcpy(s->tf.rotation.m, t.rotation, sizeof(double) * 9);
            s->tf.scale = v3(t.scale);
            sc.owned.push_back(std::move(s));
            break;
        }
        default: return nullptr;
    }
    return sc.owned.back().get();
}

const Shape* resolve_shape(Scene& sc, const rt_scene_desc& d, int kind, int index,
                           std::vector<std::vector<const Shape*>>& cache, int depth) {
    const uint64_t counts[6] = {d.n_spheres, d.n_quads, d.n_cylinders, d.n_triangles, d.n_meshes, d.n_transformed};
    if (kind < 0 || kind > 5 || index < 0 || (uint64_t)index >= counts[kind]) return nullptr;
    if (!cache[kind][index]) cache[kind][index] = make_shape(sc, d, kind, index, cache, depth);
    return cache[kind][index];
}

}  // namespace

// ---- C ABI (loaded with ctypes by tests/, bench.py's cpu_baseline leg, smoke()) ------------

extern "C" {

typedef struct oracle_scene oracle_scene;

const char* oracle_last_error(void) { return g_error.c_str(); }

int oracle_scene_create(const rt_scene_desc* d, oracle_scene** out) {
    auto sc = std::make_unique<Scene>();
    for (uint64_t i = 0; i < d->n_images; i++) {
        auto im = std::make_unique<Image>();
        im->w = d->images[i].width; im->h = d->images[i].height; im->format = d->images[i].format;
        size_t bytes = (size_t)im->w * im->h * (im->format == RT_IMG_RGBA16 ? 8 : 4);
        im->px.assign((const uint8_t*)d->images[i].pixels, (const uint8_t*)d->images[i].pixels + bytes);
        sc->images.push_back(std::move(im));
    }
    for (uint64_t i = 0; i < d->n_textures; i++) sc->textures.push_back(std::make_unique<Texture>());
    for (uint64_t i = 0; i < d->n_textures; i++) {
        const rt_texture& t = d->textures[i];
        Texture& x = *sc->textures[i];
        x.kind = t.kind;
        x.color = v3(t.color);
        x.frequency = t.frequency;
        x.interp = t.interp; x.wrap = t.wrap;
        if (t.kind == RT_TEX_CHECKERBOARD) {
            if (t.odd < 0 || t.even < 0 || (uint64_t)t.odd >= d->n_textures || (uint64_t)t.even >= d->n_textures) { g_error = "bad checkerboard child"; return RT_ERR_INVALID; }
            x.odd = sc->textures[t.odd].get(); x.even = sc->textures[t.even].get();
        }
        if (t.kind == RT_TEX_IMAGE) {
            if (t.image < 0 || (uint64_t)t.image >= d->n_images) { g_error = "bad image index"; return RT_ERR_INVALID; }
            x.image = sc->images[t.image].get();
        }
    }
    sc->materials.resize(d->n_materials);
    for (uint64_t i = 0; i < d->n_materials; i++) {
        const rt_material& m = d->materials[i];
        Material& x = sc->materials[i];
        x.kind = m.kind;
        if (m.kind == RT_MAT_LAMBERTIAN || m.kind == RT_MAT_EMITTER) {
            if (m.texture < 0 || (uint64_t)m.texture >= d->n_textures) { g_error = "bad texture index"; return RT_ERR_INVALID; }
            x.texture = sc->textures[m.texture].get();
        }
        x.albedo = v3(m.albedo); x.fuzz = m.fuzz; x.ior_in = m.ior_interior; x.ior_out = m.ior_exterior; x.roughness = m.roughness;
    }
    for (uint64_t i = 0; i < d->n_lights; i++) sc->lights.push_back({v3(d->lights[i].position), v3(d->lights[i].intensity)});
    sc->meshes.assign(d->n_meshes, nullptr);
    std::vector<std::vector<const Shape*>> cache(6);
    cache[0].assign(d->n_spheres, nullptr); cache[1].assign(d->n_quads, nullptr); cache[2].assign(d->n_cylinders, nullptr);
    cache[3].assign(d->n_triangles, nullptr); cache[4].assign(d->n_meshes, nullptr); cache[5].assign(d->n_transformed, nullptr);
    for (uint64_t i = 0; i < d->n_nodes; i++) {
        const rt_node& n = d->nodes[i];
        const Shape* s = resolve_shape(*sc, *d, n.shape_kind, n.shape_index, cache, 0);
        if (!s) { g_error = "bad shape reference in node " + std::to_string(i); return RT_ERR_INVALID; }
        if (n.material < 0 || (uint64_t)n.material >= d->n_materials) { g_error = "bad material index"; return RT_ERR_INVALID; }
        sc->nodes.push_back({s, &sc->materials[n.material]});
    }
    *out = reinterpret_cast<oracle_scene*>(sc.release());
    return RT_OK;
}

void oracle_scene_destroy(oracle_scene* s) { delete reinterpret_cast<Scene*>(s); }

int oracle_scene_bvh_info(const oracle_scene* s, uint64_t mesh, rt_bvh_info* out) {
    const Scene* sc = reinterpret_cast<const Scene*>(s);
    if (mesh >= sc->meshes.size() || !sc->meshes[mesh]) return RT_ERR_INVALID;
    *out = sc->meshes[mesh]->info;
    return RT_OK;
}

// tracePath's closest-hit loop (render.go:155-169) on a ray batch; fills the per ray
// counts of AABB.hit and Face.Collide calls the reference order makes.
int oracle_intersect_batch(const oracle_scene* s, const double* o, const double* d, uint64_t n, double tmin, double tmax,
                           rt_hit* hits, int n_threads) {
    const Scene& sc = *reinterpret_cast<const Scene*>(s);
    if (n_threads <= 0) n_threads = (int)std::thread::hardware_concurrency();
    if ((uint64_t)n_threads > n) n_threads = n ? (int)n : 1;
    std::vector<std::thread> pool;
    for (int w = 0; w < n_threads; w++) {
        pool.emplace_back([&, w] {
            Counters cnt;
            tl_counters = &cnt;
            for (uint64_t i = (uint64_t)w; i < n; i += (uint64_t)n_threads) {
                Ray r;
                r.origin = v3(o + 3 * i);
                r.direction = v3(d + 3 * i);
                cnt = Counters{};
                Collision c;
                int node = -1;
                rt_hit& h = hits[i];
                std::memset(&h, 0, sizeof(h));
                if (closest_hit(sc, r, tmin, tmax, c, node)) {
                    h.node = node; h.prim = (int32_t)c.prim; h.t = c.t;
                    h.at[0] = c.at.x; h.at[1] = c.at.y; h.at[2] = c.at.z;
                    h.normal[0] = c.normal.x; h.normal[1] = c.normal.y; h.normal[2] = c.normal.z;
                    h.uv[0] = c.uv.x; h.uv[1] = c.uv.y;
                } else {
                    h.node = -1;
                }
                h.boxes_tested = (uint32_t)cnt.boxes;
                h.prims_tested = (uint32_t)cnt.prims;
            }
            tl_counters = nullptr;
        });
    }
    for (auto& t : pool) t.join();
    return RT_OK;
}

int oracle_cast_batch(const rt_camera_desc* cam, const double* st, const uint32_t* pixel_sample, uint64_t n, int64_t seed,
                      double* o_out, double* d_out) {
    for (uint64_t i = 0; i < n; i++) {
        Rng rng;
        rng.k0 = (uint32_t)((uint64_t)seed & 0xffffffffu); rng.k1 = (uint32_t)((uint64_t)seed >> 32);
        rng.pixel = pixel_sample[2 * i]; rng.sample = pixel_sample[2 * i + 1];
        Ray r = cast(*cam, st[2 * i], st[2 * i + 1], rng);
        o_out[3 * i] = r.origin.x; o_out[3 * i + 1] = r.origin.y; o_out[3 * i + 2] = r.origin.z;
        d_out[3 * i] = r.direction.x; d_out[3 * i + 1] = r.direction.y; d_out[3 * i + 2] = r.direction.z;
    }
    return RT_OK;
}

typedef struct oracle_render_extra {
    int32_t n_threads;   // <= 0: one worker per hardware thread (runtime.NumCPU(), render.go:297)
    int32_t tile_step;   // render only tiles with tx % step == 0 && ty % step == 0 (1 = all); others stay black
    double* film;        // optional dx*dy*3 sums before the division by rays_per_pixel (row 0 = top)
    uint64_t boxes_tested, prims_tested, mesh_rays;  // out: AABB.hit / Face.Collide / Mesh.Collide calls
    int32_t threads_used;  // out
    int32_t reserved;
} oracle_render_extra;

// Render / renderScene / renderTile / renderPixel, render.go:186-349.
int oracle_render(const oracle_scene* s, const rt_camera_desc* cam, const rt_render_opts* opts, uint8_t* rgba, size_t stride,
                  rt_stats* stats, oracle_render_extra* extra) {
    const Scene& sc = *reinterpret_cast<const Scene*>(s);
    const int dx = opts->dx, dy = opts->dy, spp = opts->rays_per_pixel;
    if (opts->seed < 0 || spp <= 0 || opts->max_ray_depth <= 0 || dx <= 0 || dy <= 0) {  // render.go:81-98
        g_error = "bad RenderOptions";
        return RT_ERR_INVALID;
    }
    int n_threads = extra && extra->n_threads > 0 ? extra->n_threads : (int)std::thread::hardware_concurrency();
    const int step = extra && extra->tile_step > 1 ? extra->tile_step : 1;
    const int band_world = opts->band_world > 0 ? opts->band_world : 1;
    const int tile = 16;
    const int ntx = (dx + tile - 1) / tile, nty = (dy + tile - 1) / tile;
    std::atomic<int> next{0};
    std::mutex mu;
    Stats total;
    Counters ctotal;
    auto t0 = std::chrono::steady_clock::now();
    std::vector<std::thread> pool;
    for (int w = 0; w < n_threads; w++) {
        pool.emplace_back([&] {
            Stats st;
            Counters cnt;
            tl_counters = &cnt;
            for (;;) {
                int ti = next.fetch_add(1);
                if (ti >= ntx * nty) break;
                int tx = ti % ntx, ty = ti / ntx;
                if (tx % step || ty % step) continue;
                int x0 = tx * tile, x1 = std::min((tx + 1) * tile, dx);
                int y0 = ty * tile, y1 = std::min((ty + 1) * tile, dy);
                for (int y = y0; y < y1; y++) {
                    int imgy = dy - 1 - y;  // render.go:195
                    if ((imgy / tile) % band_world != opts->band_rank) continue;
                    for (int x = x0; x < x1; x++) {
                        V3 rgb{0, 0, 0};
                        const int s_first = opts->sample_count > 0 ? opts->sample_begin : 0;
                        const int s_end = opts->sample_count > 0 ? opts->sample_begin + opts->sample_count : spp;
                        for (int sample = s_first; sample < s_end; sample++) {
                            Rng rng;
                            rng.k0 = (uint32_t)((uint64_t)opts->seed & 0xffffffffu);
                            rng.k1 = (uint32_t)((uint64_t)opts->seed >> 32);
                            rng.pixel = (uint32_t)(y * dx + x);
                            rng.sample = (uint32_t)sample;
                            double sx, sy;
                            if (spp == 1) {  // render.go:202-210
                                sx = ((double)x + 0.5) / (double)dx;
                                sy = ((double)y + 0.5) / (double)dy;
                            } else {
                                sx = ((double)x + rng.float64()) / (double)dx;
                                sy = ((double)y + rng.float64()) / (double)dy;
                            }
                            Ray r = cast(*cam, sx, sy, rng);
                            rgb = add(rgb, trace_path(sc, opts->max_ray_depth, r, st));
                        }
                        if (extra && extra->film) {
                            double* f = extra->film + ((size_t)imgy * dx + x) * 3;
                            f[0] = rgb.x; f[1] = rgb.y; f[2] = rgb.z;
                        }
                        rgb = divs(rgb, (double)spp);
                        uint8_t* px = rgba + (size_t)imgy * stride + (size_t)x * 4;  // render.go:219-224
                        px[0] = (uint8_t)go_min(255, 255.99 * rgb.x);
                        px[1] = (uint8_t)go_min(255, 255.99 * rgb.y);
                        px[2] = (uint8_t)go_min(255, 255.99 * rgb.z);
                        px[3] = 255;
                    }
                }
            }
            tl_counters = nullptr;
            std::lock_guard<std::mutex> lk(mu);
            total.total += st.total; total.exceeded += st.exceeded; total.left += st.left; total.shadow += st.shadow;
            ctotal.boxes += cnt.boxes; ctotal.prims += cnt.prims; ctotal.mesh_rays += cnt.mesh_rays;
        });
    }
    for (auto& t : pool) t.join();
    double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (stats) {
        std::memset(stats, 0, sizeof(*stats));
        stats->total_rays = total.total; stats->rays_exceeded_depth = total.exceeded; stats->rays_left_scene = total.left;
        stats->shadow_rays = total.shadow; stats->render_seconds = secs;
    }
    if (extra) {
        extra->boxes_tested = ctotal.boxes; extra->prims_tested = ctotal.prims; extra->mesh_rays = ctotal.mesh_rays;
        extra->threads_used = n_threads;
    }
    return RT_OK;
}

// ---- small probes used to pin the restatement against the reference's unit tests ---------------

// MicrofacetBRDF: which = 0 D(h=a, n=b), 1 G1(v=a, n=b, h=c), 2 G(wo=a, wi=b, n=c, h=d), 3 F(wo=a, h=b) -> out[3],
// 4 Evaluate(wo=a, wi=b, n=c) -> out[3]
int oracle_microfacet(int which, double roughness, const double* f0, const double* a, const double* b, const double* c,
                      const double* d, double* out) {
    Microfacet m{roughness, v3(f0)};
    switch (which) {
        case 0: out[0] = m.D(v3(a), v3(b)); return RT_OK;
        case 1: out[0] = m.G1(v3(a), v3(b), v3(c)); return RT_OK;
        case 2: out[0] = m.G(v3(a), v3(b), v3(c), v3(d)); return RT_OK;
        case 3: { V3 f = m.F(v3(a), v3(b)); out[0] = f.x; out[1] = f.y; out[2] = f.z; return RT_OK; }
        case 4: { V3 f = m.evaluate(v3(a), v3(b), v3(c)); out[0] = f.x; out[1] = f.y; out[2] = f.z; return RT_OK; }
    }
    return RT_ERR_INVALID;
}

// Material.Resolve + ComputeDirectLighting for explicit hits (the checker of rt_shade_batch): the reference's own two
// calls, tracePath's lines render.go:174-176, on a surfaceInteraction built from the record.
int oracle_shade_batch(const oracle_scene* s, const rt_shade_in* in, uint64_t n, int64_t seed, rt_shade_out* out) {
    const Scene& sc = *reinterpret_cast<const Scene*>(s);
    for (uint64_t i = 0; i < n; i++) {
        const rt_shade_in& h = in[i];
        if (h.material < 0 || (size_t)h.material >= sc.materials.size() || h.node < 0 || (size_t)h.node >= sc.nodes.size()) return RT_ERR_INVALID;
        Interaction si;
        si.incoming.direction = v3(h.d);
        si.incoming.radiance = v3(h.radiance);
        si.incoming.origin = sub(v3(h.at), muls(v3(h.d), h.t));
        si.incoming.rng.k0 = (uint32_t)((uint64_t)seed & 0xffffffffu);
        si.incoming.rng.k1 = (uint32_t)((uint64_t)seed >> 32);
        si.incoming.rng.pixel = h.pixel;
        si.incoming.rng.sample = h.sample;
        si.incoming.rng.path = h.path;
        si.incoming.rng.n = h.draws;
        si.outgoing = muls(si.incoming.direction, -1);
        si.collision.t = h.t;
        si.collision.at = v3(h.at);
        si.collision.normal = v3(h.normal);
        si.collision.uv = V2{h.uv[0], h.uv[1]};
        si.node = h.node;
        const Material& m = sc.materials[(size_t)h.material];
        Resolution res = resolve(m, si);
        Stats st;
        const V3 direct = direct_lighting(sc, m, si, st);
        rt_shade_out& o = out[i];
        std::memset(&o, 0, sizeof(o));
        o.emission[0] = res.emission.x; o.emission[1] = res.emission.y; o.emission[2] = res.emission.z;
        o.n_children = res.n_scattered;
        o.draws_after = si.incoming.rng.n;
        for (int k = 0; k < res.n_scattered; k++) {
            const Ray& c = res.scattered[k];
            o.dir[k][0] = c.direction.x; o.dir[k][1] = c.direction.y; o.dir[k][2] = c.direction.z;
            o.rad[k][0] = c.radiance.x; o.rad[k][1] = c.radiance.y; o.rad[k][2] = c.radiance.z;
        }
        o.direct[0] = direct.x; o.direct[1] = direct.y; o.direct[2] = direct.z;
        o.shadow_rays = (uint32_t)st.shadow;
    }
    return RT_OK;
}

// r3 operations: op = 0 add, 1 sub, 2 mul, 3 muls(s), 4 divs(s), 5 dot -> out[0], 6 cross, 7 length -> out[0], 8 unit,
// 9 lerp(s), 10 clip(s, t)
int oracle_r3(int op, const double* a, const double* b, double s, double t, double* out) {
    V3 r{};
    switch (op) {
        case 0: r = add(v3(a), v3(b)); break;
        case 1: r = sub(v3(a), v3(b)); break;
        case 2: r = mul(v3(a), v3(b)); break;
        case 3: r = muls(v3(a), s); break;
        case 4: r = divs(v3(a), s); break;
        case 5: out[0] = dot(v3(a), v3(b)); return RT_OK;
        case 6: r = cross(v3(a), v3(b)); break;
        case 7: out[0] = length(v3(a)); return RT_OK;
        case 8: r = unit(v3(a)); break;
        case 9: r = lerp(v3(a), v3(b), s); break;
        case 10: r = clip(v3(a), s, t); break;
        default: return RT_ERR_INVALID;
    }
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
    return RT_OK;
}

int oracle_aabb_intersects(const double* a_min, const double* a_max, const double* b_min, const double* b_max) {
    AABB a{v3(a_min), v3(a_max)}, b{v3(b_min), v3(b_max)};
    return a.intersects(b) ? 1 : 0;
}

int oracle_aabb_hit(const double* bmin, const double* bmax, const double* o, const double* d, double tmin, double tmax) {
    AABB a{v3(bmin), v3(bmax)};
    Ray r;
    r.origin = v3(o); r.direction = v3(d);
    return a.hit(r, tmin, tmax) ? 1 : 0;
}

// Bounds() of node `node` of a scene -> out[6] = min xyz, max xyz.
int oracle_node_bounds(const oracle_scene* s, uint64_t node, double* out) {
    const Scene& sc = *reinterpret_cast<const Scene*>(s);
    if (node >= sc.nodes.size()) return RT_ERR_INVALID;
    AABB b = sc.nodes[node].shape->bounds();
    out[0] = b.mn.x; out[1] = b.mn.y; out[2] = b.mn.z; out[3] = b.mx.x; out[4] = b.mx.y; out[5] = b.mx.z;
    return RT_OK;
}

// Texture.At(u, v) of texture `tex` for n points.
int oracle_texture_at(const oracle_scene* s, uint64_t tex, const double* uv, uint64_t n, double* rgb) {
    const Scene& sc = *reinterpret_cast<const Scene*>(s);
    if (tex >= sc.textures.size()) return RT_ERR_INVALID;
    for (uint64_t i = 0; i < n; i++) {
        V3 c = sc.textures[tex]->at(uv[2 * i], uv[2 * i + 1]);
        rgb[3 * i] = c.x; rgb[3 * i + 1] = c.y; rgb[3 * i + 2] = c.z;
    }
    return RT_OK;
}

// ApplySeparableFilterRGBA, reconstruct.go:95-167, with the kernels of reconstruct.go:31-87
// (0 = box, 1 = tent, 2 = Mitchell-Netravali B = C = 1/3).  Packed dx*dy RGBA8 in and out.
static double recon_eval(int filter, double x) {
    if (filter == 0) return std::fabs(x) < 0.5 ? 1.0 : 0.0;
    if (filter == 1) {
        double ax = std::fabs(x);
        return ax >= 1 ? 0.0 : 1 - ax;
    }
    const double B = 1.0 / 3.0, C = 1.0 / 3.0;
    x = std::fabs(x);
    if (x >= 2) return 0;
    double x2 = x * x, x3 = x2 * x;
    if (x < 1) return ((12 - 9 * B - 6 * C) * x3 + (-18 + 12 * B + 6 * C) * x2 + (6 - 2 * B)) / 6.0;
    return ((-B - 6 * C) * x3 + (6 * B + 30 * C) * x2 + (-12 * B - 48 * C) * x + (8 * B + 24 * C)) / 6.0;
}

int oracle_apply_filter(const uint8_t* src, int32_t w, int32_t h, int filter, uint8_t* dst) {
    if (!src || !dst || w <= 0 || h <= 0 || filter < 0 || filter > 2) return RT_ERR_INVALID;
    const double radius = filter == 0 ? 0.5 : (filter == 1 ? 1.0 : 2.0);
    std::vector<double> tmp((size_t)w * h * 3);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            int x0 = (int)std::floor((double)x - radius), x1 = (int)std::ceil((double)x + radius);
            if (x0 < 0) x0 = 0;
            if (x1 > w - 1) x1 = w - 1;
            double wr = 0, wg = 0, wb = 0, wsum = 0;
            for (int xi = x0; xi <= x1; xi++) {
                double wk = recon_eval(filter, (double)x - (double)xi);
                const uint8_t* p = src + ((size_t)y * w + xi) * 4;
                double r = (double)((uint32_t)p[0] * 0x101u) / 65535.0;
                double g = (double)((uint32_t)p[1] * 0x101u) / 65535.0;
                double b = (double)((uint32_t)p[2] * 0x101u) / 65535.0;
                wr += wk * r; wg += wk * g; wb += wk * b;
                wsum += wk;
            }
            if (wsum > 0) { wr /= wsum; wg /= wsum; wb /= wsum; }
            double* t = &tmp[((size_t)y * w + x) * 3];
            t[0] = wr; t[1] = wg; t[2] = wb;
        }
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            int y0 = (int)std::floor((double)y - radius), y1 = (int)std::ceil((double)y + radius);
            if (y0 < 0) y0 = 0;
            if (y1 > h - 1) y1 = h - 1;
            double wr = 0, wg = 0, wb = 0, wsum = 0;
            for (int yi = y0; yi <= y1; yi++) {
                double wk = recon_eval(filter, (double)y - (double)yi);
                const double* t = &tmp[((size_t)yi * w + x) * 3];
                wr += wk * t[0]; wg += wk * t[1]; wb += wk * t[2];
                wsum += wk;
            }
            if (wsum > 0) { wr /= wsum; wg /= wsum; wb /= wsum; }
            uint8_t* o = dst + ((size_t)y * w + x) * 4;
            o[0] = (uint8_t)go_max(0, go_min(255, 255.0 * wr));
            o[1] = (uint8_t)go_max(0, go_min(255, 255.0 * wg));
            o[2] = (uint8_t)go_max(0, go_min(255, 255.0 * wb));
            o[3] = 255;
        }
    return RT_OK;
}

// n consecutive Float64 draws of the stream (seed, pixel, sample, path).
int oracle_random_stream(int64_t seed, uint32_t pixel, uint32_t sample, uint32_t path, uint64_t n, double* out) {
    Rng r;
    r.k0 = (uint32_t)((uint64_t)seed & 0xffffffffu); r.k1 = (uint32_t)((uint64_t)seed >> 32);
    r.pixel = pixel; r.sample = sample; r.path = path;
    for (uint64_t i = 0; i < n; i++) out[i] = r.float64();
    return RT_OK;
}

// Raw Philox4x32-10 block, for checking against the published known-answer vectors.
void oracle_philox4x32_10(const uint32_t* ctr, const uint32_t* key, uint32_t* out) {
    uint32_t c[4] = {ctr[0], ctr[1], ctr[2], ctr[3]};
    philox4x32_10(c, key[0], key[1]);
    std::memcpy(out, c, sizeof(c));
}

// Sampling helpers: which = 0 InUnitSphere, 1 InUnitDisk, 2 CosineWeightedHemisphere(normal)
int oracle_sample(int which, int64_t seed, uint32_t pixel, uint32_t sample, const double* normal, double* out) {
    Rng r;
    r.k0 = (uint32_t)((uint64_t)seed & 0xffffffffu); r.k1 = (uint32_t)((uint64_t)seed >> 32);
    r.pixel = pixel; r.sample = sample;
    V3 v = which == 0 ? in_unit_sphere(r) : (which == 1 ? in_unit_disk(r) : cosine_weighted_hemisphere(r, v3(normal)));
    out[0] = v.x; out[1] = v.y; out[2] = v.z;
    return RT_OK;
}

}  // extern "C"
