// Mesh BVH traversal: the reference-order float64 walk and the fast float32-culled walk.
#pragma once

#include "dev_shapes.h"

namespace rt {

struct VisitCounts {
    uint32_t boxes, prims;
};

// AABB.hit, shape.go:81-96: per axis 1/d, swap on negative, Go's NaN-propagating Max/Min,
// reject when tmax <= tmin.
RT_DEV bool aabb_hit_ref(const double* box, const Ray& r, double tmin, double tmax) {
#pragma unroll
    for (int axis = 0; axis < 3; axis++) {
        double inv_d = 1.0 / comp(r.d, axis);
        double o = comp(r.o, axis);
        double t0 = (box[axis] - o) * inv_d;
        double t1 = (box[3 + axis] - o) * inv_d;
        if (inv_d < 0.0) { double s = t0; t0 = t1; t1 = s; }
        tmin = go_max(t0, tmin);
        tmax = go_min(t1, tmax);
        if (tmax <= tmin) return false;
    }
    return true;
}

// BVH.Collide + Group.Collide (shape_bvh.go:44-78, 408-421) without recursion.
//
// The recursion visits Left, shrinks tmax to the left hit, then visits Right and returns
// Right's hit when both hit (Right can only hit at t <= the shrunk tmax).  Unrolled, that is
// a depth-first, left-first walk with ONE running tmax: every box and every face is tested
// against the t of the last accepted face, and the result is the last accepted face.  The
// right child is pushed and the left child (stored next in memory) is visited first.
RT_DEV bool mesh_collide_ref(const DScene& sc, const DMesh& m, const Ray& r, double tmin, double tmax, int& prim, double& t_out,
                             double& u_out, double& v_out, VisitCounts& cnt) {
    const RefNode* nodes = sc.ref_nodes + m.ref_off;
    const double* pos = sc.prim_pos + m.prim_off * 9;
    int32_t stack[40];
    int sp = 0;
    int32_t cur = 0;
    bool hit = false;
    for (;;) {
        const RefNode* n = nodes + cur;
        cnt.boxes++;
        bool descend = false;
        if (aabb_hit_ref(n->box, r, tmin, tmax)) {
            const int32_t count = n->count;
            if (count > 0) {
                const int32_t first = n->first;
                for (int32_t k = 0; k < count; k++) {
                    const double* p = pos + (int64_t)(first + k) * 9;
                    double t, u, v;
                    cnt.prims++;
                    if (mt_core(r, ld3(p), ld3(p + 3), ld3(p + 6), 1e-8, tmin, tmax, t, u, v)) {
                        hit = true;
                        tmax = t;  // closestT = coll.t (shape_bvh.go:416), then min(tmax, t) in the parents (:56)
                        prim = first + k;
                        t_out = t; u_out = u; v_out = v;
                    }
                }
            } else {
                if (sp < 40) stack[sp++] = n->right;
                cur = cur + 1;
                descend = true;
            }
        }
        if (!descend) {
            if (sp == 0) break;
            cur = stack[--sp];
        }
    }
    return hit;
}

// ---- fast walk -----------------------------------------------------------------------------
//
// Float32 boxes are conservative (rounded outward and padded on the host), so culling never
// loses a face the float64 test would accept; faces are tested in float64 with Face.Collide's
// own arithmetic, so the accepted t / barycentrics are the reference's.  Children are visited
// nearest first.  `stack` is this lane's column of an LDS array: entry i lives at
// stack[i * stride].

struct FastRay {
    float ox, oy, oz;
    float ix, iy, iz;  // 1 / direction
    double t_shift;    // the float32 ray starts at o + d * t_shift
};

// Root-box clip in float64 and set-up of the float32 ray in the mesh frame.
RT_DEV bool fast_setup(const DMesh& m, const Ray& r, double tmin, double tmax, FastRay& f) {
    double t_in = tmin, t_out = tmax;
    const double pad = (double)m.pad;
#pragma unroll
    for (int a = 0; a < 3; a++) {
        double inv = 1.0 / comp(r.d, a);
        double o = comp(r.o, a);
        double t0 = ((m.root_box[a] - pad) - o) * inv;
        double t1 = ((m.root_box[3 + a] + pad) - o) * inv;
        double lo = fmin(t0, t1), hi = fmax(t0, t1);  // fmin / fmax drop NaN: an in-plane axis does not constrain
        t_in = fmax(t_in, lo);
        t_out = fmin(t_out, hi);
    }
    if (!(t_in <= t_out)) return false;
    f.t_shift = t_in > 0 ? t_in : 0;
    V3 ol = (r.o + r.d * f.t_shift) - ld3(m.center);
    f.ox = (float)ol.x; f.oy = (float)ol.y; f.oz = (float)ol.z;
    f.ix = 1.0f / (float)r.d.x; f.iy = 1.0f / (float)r.d.y; f.iz = 1.0f / (float)r.d.z;
    return true;
}

RT_DEV float up32(double x) {  // float32 not below x
    float f = (float)x;
    if ((double)f < x) f = __int_as_float(__float_as_int(f) + (f >= 0.0f ? 1 : -1));  // next float toward +inf
    return f;
}

// Traversal stack of (node, entry distance) pairs.  The first kLdsPairs pairs live in LDS
// (this lane's column: dword i at lds[i * stride]); deeper pairs spill to private memory.
#ifndef RT_LDS_PAIRS
#define RT_LDS_PAIRS 12
#endif
constexpr int kLdsPairs = RT_LDS_PAIRS;
constexpr int kSpillPairs = 40 - RT_LDS_PAIRS;  // together >= the builder's depth limit of 32 plus the root

struct PairStack {
    int32_t* lds;
    int stride;
    int sp;
    int32_t spill[2 * kSpillPairs];
    RT_DEV void push(int32_t node, float t) {
        if (sp < kLdsPairs) {
            lds[(2 * sp) * stride] = node;
            lds[(2 * sp + 1) * stride] = __float_as_int(t);
        } else if (sp - kLdsPairs < kSpillPairs) {
            spill[2 * (sp - kLdsPairs)] = node;
            spill[2 * (sp - kLdsPairs) + 1] = __float_as_int(t);
        }
        sp++;
    }
    RT_DEV void pop(int32_t& node, float& t) {
        sp--;
        if (sp < kLdsPairs) {
            node = lds[(2 * sp) * stride];
            t = __int_as_float(lds[(2 * sp + 1) * stride]);
        } else {
            node = spill[2 * (sp - kLdsPairs)];
            t = __int_as_float(spill[2 * (sp - kLdsPairs) + 1]);
        }
    }
};

template <bool ANY_HIT>
RT_DEV bool mesh_collide_fast(const DScene& sc, const DMesh& m, const Ray& r, double tmin, double tmax, int& prim, double& t_out,
                              double& u_out, double& v_out, int32_t* lds, int stride, VisitCounts& cnt) {
    FastRay f;
    if (!fast_setup(m, r, tmin, tmax, f)) return false;
    const FastNode* nodes = sc.fast_nodes + m.fast_off;
    const double* pos = sc.prim_pos + m.prim_off * 9;
    const float tmin32 = fmaxf(0.0f, (float)(tmin - f.t_shift));
    float tmax32 = up32(tmax - f.t_shift);
    PairStack st;
    st.lds = lds;
    st.stride = stride;
    st.sp = 0;
    int32_t cur = 0;
    bool hit = false;
    for (;;) {
        const FastNode n = nodes[cur];
        cnt.boxes += 2;  // diagnostics only; dead code unless the caller reads the counts
        float tn[2];
        bool go[2];
#pragma unroll
        for (int k = 0; k < 2; k++) {
            float x0 = (n.lo[k][0] - f.ox) * f.ix, x1 = (n.hi[k][0] - f.ox) * f.ix;
            float y0 = (n.lo[k][1] - f.oy) * f.iy, y1 = (n.hi[k][1] - f.oy) * f.iy;
            float z0 = (n.lo[k][2] - f.oz) * f.iz, z1 = (n.hi[k][2] - f.oz) * f.iz;
            // fminf / fmaxf return the other operand for a NaN (0 * inf on an in-plane axis): not constraining
            float tnear = fmaxf(fmaxf(fminf(x0, x1), fminf(y0, y1)), fmaxf(fminf(z0, z1), tmin32));
            float tfar = fminf(fminf(fmaxf(x0, x1), fmaxf(y0, y1)), fminf(fmaxf(z0, z1), tmax32));
            tn[k] = tnear;
            go[k] = (tnear <= tfar) && n.count[k] >= 0;
        }
        int32_t next = -1, other = -1;
        float other_t = 0.0f;
        const int first_k = (go[0] && go[1] && tn[1] < tn[0]) ? 1 : 0;  // nearer child first
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const int k = j == 0 ? first_k : 1 - first_k;
            if (!go[k]) continue;
            if (n.count[k] > 0) {
                if (tn[k] > tmax32) continue;  // the other leaf of this node already shrank the interval
                const int32_t first = n.child[k];
                for (int32_t q = 0; q < n.count[k]; q++) {
                    const double* p = pos + (int64_t)(first + q) * 9;
                    double t, u, v;
                    cnt.prims++;
                    if (mt_core(r, ld3(p), ld3(p + 3), ld3(p + 6), 1e-8, tmin, tmax, t, u, v)) {
                        hit = true;
                        if (ANY_HIT) return true;
                        tmax = t;
                        tmax32 = up32(t - f.t_shift);
                        prim = first + q;
                        t_out = t; u_out = u; v_out = v;
                    }
                }
            } else if (next < 0) {
                next = n.child[k];
            } else {
                other = n.child[k];
                other_t = tn[k];
            }
        }
        if (next >= 0) {
            if (other >= 0) st.push(other, other_t);
            cur = next;
            continue;
        }
        bool found = false;
        while (st.sp > 0) {  // pop, skipping subtrees that begin beyond the current hit
            float tnear;
            st.pop(cur, tnear);
            if (tnear <= tmax32) { found = true; break; }
        }
        if (!found) break;
    }
    return hit;
}

}  // namespace rt
