// Ray / primitive intersection in float64, following the reference's Collide methods
// operation for operation (phys/shape_*.go).  All functions return whether the ray hits
// within [tmin, tmax] and fill a Col.
#pragma once

#include "dev_math.h"
#include "scene_types.h"

namespace rt {

struct Col {  // shape.go:11-16
    double t;
    V3 at;
    V3 normal;
    double u, v;
};

struct Ray {
    V3 o, d;
};

RT_DEV V3 ray_at(const Ray& r, double t) { return r.o + r.d * t; }  // render.go:32-35

// Moller-Trumbore core shared by Triangle / TriangleUV / Face (shape_triangle.go:105-131,
// shape_triangle_uv.go:117-141, shape_mesh.go:95-124).  e = 1e-6 or 1e-8.
RT_DEV bool mt_core(const Ray& r, V3 p0, V3 p1, V3 p2, double e, double tmin, double tmax, double& t, double& u, double& v) {
    V3 edge1 = p1 - p0;
    V3 edge2 = p2 - p0;
    V3 h = cross(r.d, edge2);
    double a = dot(edge1, h);
    if (a > -e && a < e) return false;
    double f = 1 / a;
    V3 s = r.o - p0;
    u = f * dot(s, h);
    if (u < -e || u > 1.0 + e) return false;
    V3 q = cross(s, edge1);
    v = f * dot(r.d, q);
    if (v < -e || u + v > 1.0 + e) return false;
    t = f * dot(edge2, q);
    if (t < tmin || t > tmax) return false;
    return true;
}

// Face attributes for the winning hit (shape_mesh.go:125-141).
RT_DEV void face_finish(const Ray& r, V3 p0, V3 p1, V3 p2, const double* uv6, double t, double u, double v, Col& c) {
    c.t = t;
    c.at = ray_at(r, t);
    c.normal = unit(cross(p1 - p0, p2 - p0));
    double w = 1 - u - v;
    c.u = (uv6[0] * w + uv6[2] * u) + uv6[4] * v;
    c.v = (uv6[1] * w + uv6[3] * u) + uv6[5] * v;
}

RT_DEV bool tri_collide(const DTri& tri, const Ray& r, double tmin, double tmax, Col& c) {
    V3 p0 = ld3(tri.p0), p1 = ld3(tri.p1), p2 = ld3(tri.p2);
    double t, u, v;
    const double e = tri.kind == RT_TRI_FACE ? 1e-8 : kEps;
    if (!mt_core(r, p0, p1, p2, e, tmin, tmax, t, u, v)) return false;
    if (tri.kind == RT_TRI_FACE) {
        face_finish(r, p0, p1, p2, tri.uv, t, u, v, c);
        return true;
    }
    c.t = t;
    c.at = ray_at(r, t);
    if (tri.kind == RT_TRI_TRIANGLE) {  // shape_triangle.go:132-138
        c.normal = unit(cross(p1 - p0, p2 - p0));
        c.u = u;
        c.v = v;
    } else {  // shape_triangle_uv.go:143-149; Lerp clamps its parameter (r2/point.go:38-45)
        double tt = go_max(0, go_min(1, u));
        double lx = (1 - tt) * tri.uv[0] + tt * tri.uv[2];
        double ly = (1 - tt) * tri.uv[1] + tt * tri.uv[3];
        c.u = lx + tri.uv[4] * v;
        c.v = ly + tri.uv[5] * v;
        c.normal = ld3(tri.n);
    }
    return true;
}

RT_DEV void clip01(double& a, double& b) {  // r2.Point.Clip(0, 1), r2/point.go:64-66
    a = go_min(go_max(a, 0), 1);
    b = go_min(go_max(b, 0), 1);
}

RT_DEV void equirect_uv(V3 n, bool north_up, double& u, double& v) {  // shape_sphere.go:353-370
    double phi = atan2(n.z, n.x);
    if (phi < 0) phi += 2 * kPi;
    u = phi / (2 * kPi);
    double yy = 1 < n.y ? 1 : n.y;  // the package's generic min / max (render.go:116-129)
    yy = -1 > yy ? -1 : yy;
    double theta = acos(yy);
    v = north_up ? 1 - theta / kPi : theta / kPi;
    clip01(u, v);
}

RT_DEV void box_map_uv(V3 n, double& u, double& v) {  // shape_sphere.go:380-426
    double ax = fabs(n.x), ay = fabs(n.y), az = fabs(n.z);
    double m, uc, vc;
    if (ax >= ay && ax >= az) {
        m = ax;
        if (n.x >= 0) { uc = -n.z; vc = n.y; } else { uc = n.z; vc = n.y; }
    } else if (ay >= ax && ay >= az) {
        m = ay;
        if (n.y >= 0) { uc = n.x; vc = -n.z; } else { uc = n.x; vc = n.z; }
    } else {
        m = az;
        if (n.z >= 0) { uc = n.x; vc = n.y; } else { uc = -n.x; vc = n.y; }
    }
    u = (uc / fabs(m) + 1) / 2;
    v = (vc / fabs(m) + 1) / 2;
    clip01(u, v);
}

// Sphere.Collide, shape_sphere.go:291-339.  The uv (atan2 / acos) is only needed for the
// final hit, so it is computed by sphere_uv() once the closest node is known.
RT_DEV bool sphere_collide(const DSphere& s, const Ray& r, double tmin, double tmax, Col& c) {
    V3 ctr = ld3(s.c);
    V3 oc = r.o - ctr;
    double a = dot(r.d, r.d);
    double b = dot(oc, r.d);
    double cc = dot(oc, oc) - s.r * s.r;
    double disc = b * b - a * cc;
    if (disc < 0) return false;
    double sq = sqrt(disc);
    double t = (-b - sq) / a;
    if (t < tmin || t > tmax) {
        t = (-b + sq) / a;
        if (t < tmin || t > tmax) return false;
    }
    c.t = t;
    c.at = ray_at(r, t);
    c.normal = unit(c.at - ctr);
    c.u = 0;
    c.v = 0;
    return true;
}
RT_DEV void sphere_uv(const DSphere& s, Col& c) {
    if (s.uvmap == RT_UVMAP_EQUIRECT_SOUTH_UP) equirect_uv(c.normal, false, c.u, c.v);
    else if (s.uvmap == RT_UVMAP_BOX) box_map_uv(c.normal, c.u, c.v);
    else equirect_uv(c.normal, true, c.u, c.v);
}

// Quad.Collide, shape_quad.go:39-115 (corners precomputed in DQuad).
RT_DEV bool quad_collide(const DQuad& q, const Ray& r, double tmin, double tmax, Col& c) {
    V3 p0 = ld3(q.p0), p1 = ld3(q.p1), p2 = ld3(q.p2), p3 = ld3(q.p3);
    double t1, u1, v1, t2, u2, v2;
    bool hit1 = mt_core(r, p0, p1, p2, kEps, tmin, tmax, t1, u1, v1);
    bool hit2 = mt_core(r, p0, p2, p3, kEps, tmin, tmax, t2, u2, v2);
    double t;
    if (hit1 && (!hit2 || t1 < t2)) t = t1;
    else if (hit2) t = t2;
    else return false;
    c.t = t;
    c.at = ray_at(r, t);
    V3 hp = c.at - p1;
    double uc = dot(hp, ld3(q.lu)) / q.lulu;
    double vc = dot(hp, ld3(q.lv)) / q.lvlv;
    uc = go_max(0, go_min(1, uc));
    vc = go_max(0, go_min(1, vc));
    c.u = 1 - uc;
    c.v = 1 - vc;
    c.normal = ld3(q.nrm);
    return true;
}

// Cylinder.Collide, shape_cylinder.go:40-134.
RT_DEV bool cylinder_collide(const DCylinder& cy, const Ray& r, double tmin, double tmax, Col& out) {
    V3 d = ld3(cy.d), org = ld3(cy.o);
    V3 oc = r.o - org;
    double d_rd = dot(d, r.d);
    double d_oc = dot(d, oc);
    V3 rd_perp = r.d - d * d_rd;
    V3 oc_perp = oc - d * d_oc;
    double a = dot(rd_perp, rd_perp);
    double b = 2.0 * dot(rd_perp, oc_perp);
    double cv = dot(oc_perp, oc_perp) - cy.r * cy.r;
    double closest = kMaxFloat;
    bool hit = false;
    if (a > kEps) {
        double disc = b * b - 4 * a * cv;
        if (disc >= 0) {
            double sq = sqrt(disc);
            double ts[2] = {(-b - sq) / (2.0 * a), (-b + sq) / (2.0 * a)};
#pragma unroll
            for (int k = 0; k < 2; k++) {
                double t = ts[k];
                if (t < tmin || t > tmax) continue;
                double y = d_oc + t * d_rd;
                if (y >= 0 && y <= cy.h && t < closest) {
                    V3 at = ray_at(r, t);
                    out.t = t;
                    out.at = at;
                    out.normal = unit(at - (org + d * y));
                    out.u = 0;
                    out.v = 0;
                    closest = t;
                    hit = true;
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 2; k++) {
        V3 cc = k == 0 ? org : ld3(cy.top);
        V3 cn = k == 0 ? d * -1.0 : d;
        double denom = dot(cn, r.d);
        if (fabs(denom) < kEps) continue;
        double t = dot(cn, cc - r.o) / denom;
        if (t < tmin || t > tmax) continue;
        V3 p = ray_at(r, t);
        V3 pc = p - cc;
        if (dot(pc, pc) <= cy.r * cy.r && t < closest) {
            closest = t;
            out.t = t;
            out.at = p;
            out.normal = cn;
            out.u = 0.5;
            out.v = 0.5;
            hit = true;
        }
    }
    return hit;
}

RT_DEV V3 xf_point(const DXform& x, V3 p) {  // transform.go:69-75
    V3 s = p * ld3(x.sc);
    V3 r = mk(x.rot[0] * s.x + x.rot[1] * s.y + x.rot[2] * s.z, x.rot[3] * s.x + x.rot[4] * s.y + x.rot[5] * s.z,
              x.rot[6] * s.x + x.rot[7] * s.y + x.rot[8] * s.z);
    return r + ld3(x.tr);
}
RT_DEV V3 xf_vector(const DXform& x, V3 v) {  // transform.go:78-83
    V3 s = v * ld3(x.sc);
    return mk(x.rot[0] * s.x + x.rot[1] * s.y + x.rot[2] * s.z, x.rot[3] * s.x + x.rot[4] * s.y + x.rot[5] * s.z,
              x.rot[6] * s.x + x.rot[7] * s.y + x.rot[8] * s.z);
}

}  // namespace rt
