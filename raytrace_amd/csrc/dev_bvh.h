// Mesh BVH traversal: the reference-order float64 walk and the fast float32-culled walk.
#pragma once

#include "dev_shapes.h"

namespace rt {

struct VisitCounts {
    uint32_t boxes, prims;
    // Set when a traversal stack ran full (the host sizes them so that it cannot happen).  Kept in a register and turned
    // into one atomic at the very end of a kernel: an atomic inside the walk would make every load after it "possibly
    // clobbered", and the compiler then gives up the scalar loads of wave-uniform scene data.
    bool overflow = false;
#ifdef RT_VISIT_STATS
    // per-lane walk, diagnostics: [0] walks of the wave, [1] lanes in them, [2] inner-node steps of the wave, [3] lanes in
    // them, [4] face steps, [5] lanes in them; [6..11] the same for any-hit walks (all wave-uniform sums); k_walk's rounds use
    // [0..5] the same way (face steps = float32 pre-tests) and [12] float64 face steps, [13] lanes in them
    uint32_t w[16] = {};
#endif
};

// Diagnostics (builds with -DRT_VISIT_STATS only): events per wave, summed over the launch.
#ifdef RT_WAVE_TIMES  // diagnostics build: when the persistent waves of k_beam begin and end
__device__ unsigned long long g_wave_times[4 * 8192];  // per wave: start, end of its last item (constant-rate clock), items traced and the last two in detail (k_beam)
#endif
#ifdef RT_VISIT_STATS
__device__ unsigned long long g_packet_stats[16];  // packet walk: walks, inner nodes, leaves, float32 face tests, float64 face tests, pops, pushes, -, cycles in walks, cycles in rounds, rounds, cycles in leaves
__device__ unsigned long long g_trace_stats[64][64];   // k_trace / k_direct, striped by workgroup; see render.cpp's print-out
#define RT_STAT(i, n) do { if ((threadIdx.x & 63) == (unsigned)(__ffsll((long long)__builtin_amdgcn_ballot_w64(true)) - 1)) atomicAdd(&g_packet_stats[i], (unsigned long long)(n)); } while (0)
#define RT_TSTAT(i, n) do { const unsigned long long v_ = (unsigned long long)(n); if ((threadIdx.x & 63) == 0) atomicAdd(&g_trace_stats[(blockIdx.x + blockIdx.y * 7u) & 63u][i], v_); } while (0)
#define RT_WSTAT(i, n) do { cnt.w[(ANY_HIT ? 6 : 0) + (i)] += (uint32_t)(n); } while (0)
#define RT_FIRST_ACTIVE (((threadIdx.x & 63) == (unsigned)(__ffsll((long long)__builtin_amdgcn_ballot_w64(true)) - 1)) ? 1 : 0)
// every lane adds its own tallies (first-active-lane tallies add up to the wave's step counts)
#define RT_WFLUSH(c) do { for (int k_ = 0; k_ < 16; k_++) if ((c).w[k_]) atomicAdd(&g_trace_stats[(blockIdx.x + threadIdx.x) & 63u][16 + k_], (unsigned long long)(c).w[k_]); } while (0)
#else
#define RT_STAT(i, n) do { } while (0)
#define RT_TSTAT(i, n) do { } while (0)
#define RT_WSTAT(i, n) do { } while (0)
#define RT_WFLUSH(c) do { } while (0)
#endif

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
                if (sp < 40) stack[sp++] = n->right;  // NewBVH stops at depth 32 (shape_bvh.go:104): 33 entries at most
                else cnt.overflow = true;
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

typedef float f32x2 __attribute__((ext_vector_type(2)));

struct FastRay {
    float ox, oy, oz;
    float ix, iy, iz;     // 1 / direction, magnitude capped at 2^90 (see slab_inverse)
    float nx, ny, nz;     // -o * (1 / direction): a slab distance is one fused multiply-add, plane * i + n
    double t_shift;       // the float32 ray starts at o + d * t_shift
};

// 1 / d with |result| <= 2^90, so that every slab product stays finite.  For a direction component
// smaller than 2^-90 the slab then spans |t| >= 2^90 * (distance from the origin to the nearer plane),
// and that distance is at least most of the box padding for any ray the unpadded box contains: far
// beyond every t a traversal compares with, so the cap never culls such a ray.
RT_DEV float slab_inverse(float d) {
    const float cap = 1.2379400392853803e27f;  // 2^90
    const float inv = __builtin_amdgcn_rcpf(d);  // within 1 ulp of 1 / d: one more 2^-24 of slack the box padding has to spare
    return fabsf(inv) <= cap ? inv : copysignf(cap, d);
}

// 1 / d to about 2^-50 (hardware estimate and one Newton step) instead of the dozen instructions of an exact division:
// what is computed from it, where the ray enters the padded root box, only has to be right to a fraction of the padding.
RT_DEV double near_reciprocal(double d) {
    const double r0 = __builtin_amdgcn_rcp(d);
    const double r1 = __builtin_fma(__builtin_fma(-d, r0, 1.0), r0, r0);
    return __builtin_isinf(r0) ? r0 : r1;  // d = 0: keep the infinity (the Newton step would make it NaN)
}

// Root-box clip in float64 and set-up of the float32 ray in the mesh frame.
RT_DEV bool fast_setup(const DMesh& m, const Ray& r, double tmin, double tmax, FastRay& f) {
    double t_in = tmin, t_out = tmax;
    const double pad = (double)m.pad;
#pragma unroll
    for (int a = 0; a < 3; a++) {
        double inv = near_reciprocal(comp(r.d, a));
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
    f.ix = slab_inverse((float)r.d.x); f.iy = slab_inverse((float)r.d.y); f.iz = slab_inverse((float)r.d.z);
    f.nx = -f.ox * f.ix; f.ny = -f.oy * f.iy; f.nz = -f.oz * f.iz;
    return true;
}

RT_DEV float up32(double x) {  // float32 not below x
    float f = (float)x;
    if ((double)f < x) f = __int_as_float(__float_as_int(f) + (f >= 0.0f ? 1 : -1));  // next float toward +inf
    return f;
}

// A node read through the constant address space: with a wave-uniform index the compiler emits scalar loads
// (s_load_dwordx4 into SGPRs, served by the scalar data cache) instead of per-lane vector loads.
typedef int32_t i32x4 __attribute__((ext_vector_type(4)));
#define RT_CONST_AS __attribute__((address_space(4)))
RT_DEV FastNode load_node_scalar(const FastNode* nodes, int32_t index) {
    const RT_CONST_AS i32x4* p = (const RT_CONST_AS i32x4*)(nodes + index);
    union {
        i32x4 v[4];
        FastNode n;
    } u;
    u.v[0] = p[0]; u.v[1] = p[1]; u.v[2] = p[2]; u.v[3] = p[3];
    return u.n;
}

RT_DEV Tri32 load_tri32_scalar(const Tri32* tris, int32_t index) {
    const RT_CONST_AS i32x4* p = (const RT_CONST_AS i32x4*)(tris + index);
    union {
        i32x4 v[3];
        Tri32 t;
    } u;
    u.v[0] = p[0]; u.v[1] = p[1]; u.v[2] = p[2];
    return u.t;
}
// Conservative float32 pre-test of a face: true = the float64 test cannot accept this face.
//
// Moller-Trumbore without the division: with a = e1.(d x e2), U = s.(d x e2), V = d.(s x e1) the
// float64 test accepts only if -e*a <= U, -e*a <= V and U + V <= (1 + e)*a for a > 0 (mirrored
// for a < 0).  The float32 values differ from the exact ones by at most
//   |dU| <= c * eps32 * |d| * |e2| * R,  |dV| <= c * eps32 * |d| * |e1| * R,  |da| <= c * eps32 * |d| * |e1| * |e2|
// (R = mesh radius: inputs are rounded to float32 relative to the mesh centre, the shifted ray
// origin lies within the padded root box, and every product is bounded by these magnitudes;
// worst-case c is about 20, kErr carries 64).  A face is rejected only when a bound is violated
// by more than these margins and the sign of a is certain.
constexpr float kErr = 64.0f * 5.9604645e-8f;  // 64 * 2^-24

RT_DEV bool tri32_rejects(const Tri32& t, const FastRay& f, float dx, float dy, float dz, float err_r /* kErr * |d| * R */,
                          float err_d /* kErr * |d| */) {
#pragma clang fp contract(fast)  // fusing only removes roundings: the error bounds above still hold
    const float hx = dy * t.e2[2] - dz * t.e2[1], hy = dz * t.e2[0] - dx * t.e2[2], hz = dx * t.e2[1] - dy * t.e2[0];
    const float a = t.e1[0] * hx + t.e1[1] * hy + t.e1[2] * hz;
    const float sx = f.ox - t.p0[0], sy = f.oy - t.p0[1], sz = f.oz - t.p0[2];
    const float U = sx * hx + sy * hy + sz * hz;
    const float qx = sy * t.e1[2] - sz * t.e1[1], qy = sz * t.e1[0] - sx * t.e1[2], qz = sx * t.e1[1] - sy * t.e1[0];
    const float V = dx * qx + dy * qy + dz * qz;
    const float eu = err_r * t.len2, ev = err_r * t.len1, ea = err_d * t.len1 * t.len2;
    const float aa = fabsf(a);
    if (!(aa > ea)) return false;  // sign of a uncertain (grazing or degenerate): let float64 decide
    const float us = a < 0 ? -U : U, vs = a < 0 ? -V : V;
    return us < -eu || vs < -ev || us + vs > aa + eu + ev + ea;
}

// Traversal stack of (child reference, entry distance) pairs, entirely in LDS: this lane's
// column of a [2 * pairs][lanes] array (dword i at lds[i * stride]).  Near-first descent pushes at
// most one pair per tree level, so the host sizes the array from the depth of the deepest fast tree
// of the scene and the stack cannot overflow; nothing goes to private (scratch) memory.  A push that
// finds it full all the same sets `overflow` (the host turns that into a failed render).
struct PairStack {
    int32_t* lds;
    int stride;
    int cap;  // pairs available
    int sp;
    bool overflow;
    RT_DEV void push(int32_t ref, float t) {
        if (sp < cap) {
            lds[(2 * sp) * stride] = ref;
            lds[(2 * sp + 1) * stride] = __float_as_int(t);
            sp++;
        } else {
            overflow = true;
        }
    }
    RT_DEV void pop(int32_t& ref, float& t) {
        sp--;
        ref = lds[(2 * sp) * stride];
        t = __int_as_float(lds[(2 * sp + 1) * stride]);
    }
};

// Closest (or any) hit of one mesh.  while-while form: every lane first walks inner nodes until it
// holds a leaf (or runs out of work), then the lanes of the wave test their leaves together, so the
// float64 intersections run with most lanes active instead of stalling the node loop.
template <bool ANY_HIT>
RT_DEV bool mesh_collide_fast(const DScene& sc, const DMesh& m, const Ray& r, double tmin, double tmax, int& prim, double& t_out,
                              double& u_out, double& v_out, int32_t* lds, int stride, int cap, VisitCounts& cnt) {
    FastRay f;
    if (!fast_setup(m, r, tmin, tmax, f)) return false;
    const FastNode* nodes = sc.fast_nodes + m.fast_off;
    const double* pos = sc.prim_pos + m.prim_off * 9;
    const Tri32* tri32 = sc.prim_tri32 + m.prim_off;
    const float tmin32 = fmaxf(0.0f, (float)(tmin - f.t_shift));
    float tmax32 = up32(tmax - f.t_shift);
    const float dx = (float)r.d.x, dy = (float)r.d.y, dz = (float)r.d.z;
    const float dlen = sqrtf(dx * dx + dy * dy + dz * dz);
    const float err_d = kErr * dlen, err_r = err_d * m.radius;
    PairStack st;
    st.lds = lds;
    st.stride = stride;
    st.cap = cap;
    st.sp = 0;
    st.overflow = false;
    constexpr int32_t kDone = kFastEmpty;
    int32_t cur = 0;
    bool hit = false;
    RT_WSTAT(0, RT_FIRST_ACTIVE);
    RT_WSTAT(1, 1);
    for (;;) {
        while (cur >= 0) {  // inner node: test both children, go to the nearer, push the farther
            RT_WSTAT(2, RT_FIRST_ACTIVE);
            RT_WSTAT(3, 1);
            // When every active lane stands on the same node (the usual case when the lanes of a wave trace samples of
            // one pixel) the node is fetched once through the scalar cache into SGPRs instead of 64 times through the
            // vector memory pipeline, whose return path (64 B per clock per CU) is what bounds this loop otherwise.
            const int32_t ucur = __builtin_amdgcn_readfirstlane(cur);
            cnt.boxes += 2;  // diagnostics only; dead code unless the caller reads the counts
            float tn[2];
            bool go[2];
            int32_t child[2];
            // The slab tests, written once and instantiated for a node held in SGPRs and for one held in VGPRs (the
            // planes are then direct scalar operands of the packed multiply-adds instead of being copied to VGPRs).
            auto test_children = [&](const FastNode& n) {
#pragma unroll
                for (int k = 0; k < 2; k++) {
                    // both planes of a slab in one packed fused multiply-add; the rounding of n and of the sum moves a
                    // plane by a few 2^-24 of the mesh radius, well inside the padding of the boxes
                    const f32x2 x = __builtin_elementwise_fma(f32x2{n.b[k][0][0], n.b[k][0][1]}, f32x2{f.ix, f.ix}, f32x2{f.nx, f.nx});
                    const f32x2 y = __builtin_elementwise_fma(f32x2{n.b[k][1][0], n.b[k][1][1]}, f32x2{f.iy, f.iy}, f32x2{f.ny, f.ny});
                    const f32x2 z = __builtin_elementwise_fma(f32x2{n.b[k][2][0], n.b[k][2][1]}, f32x2{f.iz, f.iz}, f32x2{f.nz, f.nz});
                    const float tnear = fmaxf(fmaxf(fminf(x[0], x[1]), fminf(y[0], y[1])), fmaxf(fminf(z[0], z[1]), tmin32));
                    const float tfar = fminf(fminf(fmaxf(x[0], x[1]), fmaxf(y[0], y[1])), fminf(fmaxf(z[0], z[1]), tmax32));
                    tn[k] = tnear;
                    go[k] = (tnear <= tfar) && n.child[k] != kFastEmpty;
                    child[k] = n.child[k];
                }
            };
            if (__builtin_amdgcn_ballot_w64(cur != ucur) == 0) test_children(load_node_scalar(nodes, ucur));
            else test_children(nodes[cur]);
            if (go[0] && go[1]) {
                const bool swap = tn[1] < tn[0];
                st.push(swap ? child[0] : child[1], swap ? tn[0] : tn[1]);
                cur = swap ? child[1] : child[0];
            } else if (go[0] || go[1]) {
                cur = go[0] ? child[0] : child[1];
            } else {
                cur = kDone;
                while (st.sp > 0) {  // pop, skipping subtrees that begin beyond the current hit
                    int32_t ref;
                    float tnear;
                    st.pop(ref, tnear);
                    if (tnear <= tmax32) { cur = ref; break; }
                }
            }
        }
        if (cur == kDone) break;
        {  // leaf; like the nodes, its float32 faces come through the scalar cache when all active lanes hold the same leaf
           // (the float64 vertices of the few survivors stay on the vector path: 18 more SGPRs cost more than they save)
            const int32_t uleaf = __builtin_amdgcn_readfirstlane(cur);
            const bool uniform = __builtin_amdgcn_ballot_w64(cur != uleaf) == 0;
            const uint32_t code = (uint32_t)~(uniform ? uleaf : cur);
            const int32_t first = (int32_t)(code >> 3);
            const int32_t count = (int32_t)(code & 7u) + 1;
            for (int32_t q = 0; q < count; q++) {
                cnt.prims++;
                RT_WSTAT(4, RT_FIRST_ACTIVE);
                RT_WSTAT(5, 1);
                const Tri32 t32 = uniform ? load_tri32_scalar(tri32, first + q) : tri32[first + q];
                if (tri32_rejects(t32, f, dx, dy, dz, err_r, err_d)) continue;
                const double* p = pos + (int64_t)(first + q) * 9;
                const V3 p0 = ld3(p), p1 = ld3(p + 3), p2 = ld3(p + 6);
                double t, u, v;
                if (mt_core(r, p0, p1, p2, 1e-8, tmin, tmax, t, u, v)) {
                    if (ANY_HIT) { cnt.overflow |= st.overflow; return true; }
                    if (hit && t == t_out && first + q < prim) continue;  // equal distances: see mesh_collide_packet
                    hit = true;
                    tmax = t;
                    tmax32 = up32(t - f.t_shift);
                    prim = first + q;
                    t_out = t; u_out = u; v_out = v;
                }
            }
        }
        cur = kDone;
        while (st.sp > 0) {
            int32_t ref;
            float tnear;
            st.pop(ref, tnear);
            if (tnear <= tmax32) { cur = ref; break; }
        }
    }
    cnt.overflow |= st.overflow;
    return hit;
}

// ---- resumable per-lane walk -------------------------------------------------------------------
//
// The same search as mesh_collide_fast, cut into rounds so that a wave can hand finished lanes a new ray while the
// others keep walking (k_walk): everything a lane needs between rounds lives in WalkLane and in its LDS stack.  The
// rays of a wave are unrelated here, so nodes and faces are always fetched per lane.

// The closest hit found so far over all nodes of the scene.  Among hits at exactly the same distance the lower node
// wins (tracePath keeps the first node: `c.t < minDist`, render.go:161) and, inside a mesh, the face latest in leaf
// order (see mesh_collide_packet), whatever order the nodes and faces were visited in.
struct BestHit {
    double t;
    int32_t node;  // -1: nothing hit yet
    int32_t leaf;  // mesh: position of the face in leaf order; other shapes: -1
    RT_DEV bool takes(double t_new, int32_t node_new, int32_t leaf_new) const {
        if (node < 0 || t_new < t) return true;
        if (t_new > t) return false;
        return node_new < node || (node_new == node && leaf_new > leaf);
    }
};

struct WalkLane {
    Ray lr;                          // the ray in the frame of the mesh being walked
    float ox, oy, oz;                // its float32 origin: o + d * t_shift, relative to the mesh centre
    double t_shift;
    // slab distance of grid plane p of axis a: p * ga + gb (ga = grid step / direction, gb = (grid origin - ray origin) / direction)
    float gax, gay, gaz, gbx, gby, gbz;
    float tmin32, tmax32;
    float dx, dy, dz, err_d, err_r;  // float32 direction and the error scales of the face pre-test
    uint32_t node_base, prim_base;   // first FastNode / first prim (leaf order) of that mesh
    int32_t cur;                     // >= 0: inner node; kFastEmpty: not walking; any other negative value: a leaf
    int32_t sp;                      // entries on the stack
    int32_t base;                    // entries [base, sp) are in the LDS ring, [0, base) in the spill area
#ifdef RT_VISIT_STATS
    uint32_t steps = 0;              // inner nodes visited for the current ray (diagnostics)
#endif
};

// The stack of a resumable walk: its top kWalkRing entries in an LDS ring (entry i in slot i % kWalkRing of this lane's
// column), older ones in a per-lane spill area in device memory.  A 22-level tree would otherwise need 23 entries per
// lane in LDS, which caps the kernel at 3 waves per SIMD; most pushes and pops never leave the ring.
#ifndef RT_WALK_RING
#define RT_WALK_RING 8
#endif
constexpr int kWalkRing = RT_WALK_RING;
static_assert((kWalkRing & (kWalkRing - 1)) == 0, "the ring index is a mask");
struct WalkStack {
    int32_t* lds;   // this lane's column: dword j of ring slot s at lds[(2 * s + j) * 64]
    int2* spill;    // this lane's column of the wave's spill area: entry i at spill[i * 64]
    int cap;        // entries the spill area holds (depth of the deepest tree + 1)
};

RT_DEV bool walk_begin(const DMesh& m, const Ray& lr, double tmin, double tmax, WalkLane& w) {
    FastRay f;
    if (!fast_setup(m, lr, tmin, tmax, f)) return false;
    w.lr = lr;
    w.ox = f.ox; w.oy = f.oy; w.oz = f.oz;
    w.t_shift = f.t_shift;
    w.gax = m.grid_step[0] * f.ix; w.gay = m.grid_step[1] * f.iy; w.gaz = m.grid_step[2] * f.iz;
    w.gbx = (m.grid_lo[0] - f.ox) * f.ix; w.gby = (m.grid_lo[1] - f.oy) * f.iy; w.gbz = (m.grid_lo[2] - f.oz) * f.iz;
    w.tmin32 = fmaxf(0.0f, (float)(tmin - f.t_shift));
    w.tmax32 = up32(tmax - f.t_shift);
    w.dx = (float)lr.d.x; w.dy = (float)lr.d.y; w.dz = (float)lr.d.z;
    const float dlen = sqrtf(w.dx * w.dx + w.dy * w.dy + w.dz * w.dz);
    w.err_d = kErr * dlen;
    w.err_r = w.err_d * m.radius;
#ifdef RT_WALK_BINARY
    w.node_base = (uint32_t)m.fast_off;
#else
    w.node_base = (uint32_t)m.q4_off;
#endif
    w.prim_base = (uint32_t)m.prim_off;
    w.cur = 0;
    w.sp = 0;
    w.base = 0;
    return true;
}

// One round for a walking lane: inner nodes down to the next leaf, that leaf's faces, then the next subtree from the
// stack.  Leaves w.cur == kFastEmpty when the walk is over.  `node` is the scene node being walked (for the tie rule),
// tmin the ray's lower limit; best.t is the search limit.  ANY_HIT: the first face hit ends the walk (best.node >= 0 then).
// RING: entries the LDS part of the stack holds, a power of two (older entries go to st.spill); 0: the LDS column has
// room for all st.cap entries and nothing is ever spilled.
template <bool ANY_HIT, int RING>
RT_DEV void walk_round(const DScene& sc, WalkLane& w, int32_t node, double tmin, BestHit& best, const WalkStack& st, bool& overflow) {
    const QNode* nodes = sc.q_nodes + w.node_base;
    const double* pos = sc.prim_pos + (size_t)w.prim_base * 9;
    const Tri32* tri32 = sc.prim_tri32 + w.prim_base;
    FastRay f;  // what the face pre-test reads of it
    f.ox = w.ox; f.oy = w.oy; f.oz = w.oz;
    auto pop_next = [&]() {
        w.cur = kFastEmpty;
        while (w.sp > 0) {  // skipping subtrees that begin beyond the current hit
            w.sp--;
            int32_t ref, tbits;
            if (RING == 0 || w.sp >= w.base) {
                const int slot = RING ? (w.sp & (RING - 1)) : w.sp;
                ref = st.lds[(2 * slot) * 64];
                tbits = st.lds[(2 * slot + 1) * 64];
            } else {  // the ring is empty: the next entry comes back from the spill area
                const int2 e = st.spill[(size_t)w.sp * 64];
                ref = e.x;
                tbits = e.y;
                w.base = w.sp;
            }
            if (__int_as_float(tbits) <= w.tmax32) { w.cur = ref; break; }
        }
    };
    while (w.cur >= 0) {
#ifdef RT_VISIT_STATS
        w.steps++;
#endif
        const QNode n = nodes[w.cur];
        float tn[2];
        bool go[2];
#pragma unroll
        for (int k = 0; k < 2; k++) {
            // plane positions are whole numbers below 2^16: exact in float32; the fused multiply-add rounds once
            const f32x2 x = __builtin_elementwise_fma(f32x2{(float)n.q[k][0][0], (float)n.q[k][0][1]}, f32x2{w.gax, w.gax}, f32x2{w.gbx, w.gbx});
            const f32x2 y = __builtin_elementwise_fma(f32x2{(float)n.q[k][1][0], (float)n.q[k][1][1]}, f32x2{w.gay, w.gay}, f32x2{w.gby, w.gby});
            const f32x2 z = __builtin_elementwise_fma(f32x2{(float)n.q[k][2][0], (float)n.q[k][2][1]}, f32x2{w.gaz, w.gaz}, f32x2{w.gbz, w.gbz});
            const float tnear = fmaxf(fmaxf(fminf(x[0], x[1]), fminf(y[0], y[1])), fmaxf(fminf(z[0], z[1]), w.tmin32));
            const float tfar = fminf(fminf(fmaxf(x[0], x[1]), fmaxf(y[0], y[1])), fminf(fmaxf(z[0], z[1]), w.tmax32));
            tn[k] = tnear;
            go[k] = (tnear <= tfar) && n.child[k] != kFastEmpty;
        }
        if (go[0] && go[1]) {
            const bool swap = tn[1] < tn[0];
            if (w.sp < st.cap) {
                if (RING != 0 && w.sp - w.base == RING) {  // the ring is full: its oldest entry moves to the spill area
                    const int old = w.base & (RING - 1);
                    st.spill[(size_t)w.base * 64] = int2{st.lds[(2 * old) * 64], st.lds[(2 * old + 1) * 64]};
                    w.base++;
                }
                const int slot = RING ? (w.sp & (RING - 1)) : w.sp;
                st.lds[(2 * slot) * 64] = swap ? n.child[0] : n.child[1];
                st.lds[(2 * slot + 1) * 64] = __float_as_int(swap ? tn[0] : tn[1]);
                w.sp++;
            } else {
                overflow = true;  // the host sizes the stack from the deepest tree: cannot happen, but must not pass silently
            }
            w.cur = swap ? n.child[1] : n.child[0];
        } else if (go[0] || go[1]) {
            w.cur = go[0] ? n.child[0] : n.child[1];
        } else {
            pop_next();
        }
    }
    if (w.cur == kFastEmpty) return;
    const uint32_t code = (uint32_t)~w.cur;
    const int32_t first = (int32_t)(code >> 3);
    const int32_t count = (int32_t)(code & 7u) + 1;
    // The float32 pre-tests of the whole leaf first, then the float64 tests of the survivors: the lanes of a wave stand on
    // different leaves, so with both in one loop the long float64 block would run in nearly every iteration for the few
    // lanes whose face passed; apart, it runs as often as the lane with the most survivors needs (once or twice).
    uint32_t pending = 0;
    for (int32_t q = 0; q < count; q++)
        if (!tri32_rejects(tri32[first + q], f, w.dx, w.dy, w.dz, w.err_r, w.err_d)) pending |= 1u << q;
    while (pending != 0) {
        const int32_t face = first + (__builtin_ffs((int)pending) - 1);
        pending &= pending - 1;
        const double* p = pos + (int64_t)face * 9;
        double t, u, v;
        if (mt_core(w.lr, ld3(p), ld3(p + 3), ld3(p + 6), 1e-8, tmin, best.t, t, u, v)) {
            if (ANY_HIT) {
                best.node = node;
                w.cur = kFastEmpty;
                return;
            }
            if (!best.takes(t, node, face)) continue;
            best.t = t;
            best.node = node;
            best.leaf = face;
            w.tmax32 = up32(t - w.t_shift);
        }
    }
    pop_next();
}

// walk_round over the four-wide nodes (Q4Node): the same search with half as many dependent node fetches.  The (up to
// four) children a ray reaches are sorted by entry distance; the nearest is visited next, the others are pushed farthest
// first.  WalkLane::node_base must then be the mesh's first Q4Node.
template <bool ANY_HIT, int RING>
RT_DEV void walk_round4(const DScene& sc, WalkLane& w, int32_t node, double tmin, BestHit& best, const WalkStack& st, bool& overflow
#ifdef RT_VISIT_STATS
                        , uint32_t* ws
#endif
) {
#ifdef RT_VISIT_STATS
    ws[0] += RT_FIRST_ACTIVE; ws[1] += 1;
#endif
    const Q4Node* nodes = sc.q4_nodes + w.node_base;
    const double* pos = sc.prim_pos + (size_t)w.prim_base * 9;
    const Tri32* tri32 = sc.prim_tri32 + w.prim_base;
    const float inf32 = __builtin_inff();
    FastRay f;  // what the face pre-test reads of it
    f.ox = w.ox; f.oy = w.oy; f.oz = w.oz;
    auto pop_next = [&]() {
        w.cur = kFastEmpty;
        while (w.sp > 0) {  // skipping subtrees that begin beyond the current hit
            w.sp--;
            int32_t ref, tbits;
            if (RING == 0 || w.sp >= w.base) {
                const int slot = RING ? (w.sp & (RING - 1)) : w.sp;
                ref = st.lds[(2 * slot) * 64];
                tbits = st.lds[(2 * slot + 1) * 64];
            } else {  // the ring is empty: the next entry comes back from the spill area
                const int2 e = st.spill[(size_t)w.sp * 64];
                ref = e.x;
                tbits = e.y;
                w.base = w.sp;
            }
            if (__int_as_float(tbits) <= w.tmax32) { w.cur = ref; break; }
        }
    };
    auto push = [&](int32_t ref, float t) {
        if (w.sp < st.cap) {
            if (RING != 0 && w.sp - w.base == RING) {  // the ring is full: its oldest entry moves to the spill area
                const int old = w.base & (RING - 1);
                st.spill[(size_t)w.base * 64] = int2{st.lds[(2 * old) * 64], st.lds[(2 * old + 1) * 64]};
                w.base++;
            }
            const int slot = RING ? (w.sp & (RING - 1)) : w.sp;
            st.lds[(2 * slot) * 64] = ref;
            st.lds[(2 * slot + 1) * 64] = __float_as_int(t);
            w.sp++;
        } else {
            overflow = true;  // the host sizes the stack from the deepest tree: cannot happen, but must not pass silently
        }
    };
    // DScene::walk_break (per cent, 0 = off): the round's descent ends once fewer than that share of the lanes that began it
    // are still among inner nodes
    const int stay_n = sc.walk_break ? __popcll(__builtin_amdgcn_ballot_w64(true)) * sc.walk_break : 0;
    while (w.cur >= 0) {
#ifdef RT_VISIT_STATS
        w.steps++;
        ws[2] += RT_FIRST_ACTIVE; ws[3] += 1;
#endif
        const Q4Node n = nodes[w.cur];
        float tn[4];
        int32_t ch[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const f32x2 x = __builtin_elementwise_fma(f32x2{(float)n.q[k][0][0], (float)n.q[k][0][1]}, f32x2{w.gax, w.gax}, f32x2{w.gbx, w.gbx});
            const f32x2 y = __builtin_elementwise_fma(f32x2{(float)n.q[k][1][0], (float)n.q[k][1][1]}, f32x2{w.gay, w.gay}, f32x2{w.gby, w.gby});
            const f32x2 z = __builtin_elementwise_fma(f32x2{(float)n.q[k][2][0], (float)n.q[k][2][1]}, f32x2{w.gaz, w.gaz}, f32x2{w.gbz, w.gbz});
            const float tnear = fmaxf(fmaxf(fminf(x[0], x[1]), fminf(y[0], y[1])), fmaxf(fminf(z[0], z[1]), w.tmin32));
            const float tfar = fminf(fminf(fmaxf(x[0], x[1]), fmaxf(y[0], y[1])), fminf(fmaxf(z[0], z[1]), w.tmax32));
            ch[k] = n.child[k];
            tn[k] = ((tnear <= tfar) && ch[k] != kFastEmpty) ? tnear : inf32;
        }
        auto order = [&](int a, int b) {  // tn[a] <= tn[b] afterwards
            const bool sw = tn[b] < tn[a];
            const float ta = sw ? tn[b] : tn[a], tb = sw ? tn[a] : tn[b];
            const int32_t ca = sw ? ch[b] : ch[a], cb = sw ? ch[a] : ch[b];
            tn[a] = ta; tn[b] = tb; ch[a] = ca; ch[b] = cb;
        };
        order(0, 1); order(2, 3); order(0, 2); order(1, 3); order(1, 2);
        if (tn[0] < inf32) {
            if (tn[1] < inf32) {
                if (tn[2] < inf32) {
                    if (tn[3] < inf32) push(ch[3], tn[3]);
                    push(ch[2], tn[2]);
                }
                push(ch[1], tn[1]);
            }
            w.cur = ch[0];
        } else {
            pop_next();
        }
        // the few lanes still among inner nodes stop here and go on in the next round: the others' leaves are not kept waiting
        if (__popcll(__builtin_amdgcn_ballot_w64(w.cur >= 0)) * 100 < stay_n) break;
    }
    if (w.cur == kFastEmpty || w.cur >= 0) return;
    const uint32_t code = (uint32_t)~w.cur;
    const int32_t first = (int32_t)(code >> 3);
    const int32_t count = (int32_t)(code & 7u) + 1;
    uint32_t pending = 0;  // pre-tests of the whole leaf first, then the float64 tests of the survivors (see walk_round)
    for (int32_t q = 0; q < count; q++) {
#ifdef RT_VISIT_STATS
        ws[4] += RT_FIRST_ACTIVE; ws[5] += 1;
#endif
        if (!tri32_rejects(tri32[first + q], f, w.dx, w.dy, w.dz, w.err_r, w.err_d)) pending |= 1u << q;
    }
    while (pending != 0) {
#ifdef RT_VISIT_STATS
        ws[12] += RT_FIRST_ACTIVE; ws[13] += 1;
#endif
        const int32_t face = first + (__builtin_ffs((int)pending) - 1);
        pending &= pending - 1;
        const double* p = pos + (int64_t)face * 9;
        double t, u, v;
        if (mt_core(w.lr, ld3(p), ld3(p + 3), ld3(p + 6), 1e-8, tmin, best.t, t, u, v)) {
            if (ANY_HIT) {
                best.node = node;
                w.cur = kFastEmpty;
                return;
            }
            if (!best.takes(t, node, face)) continue;
            best.t = t;
            best.node = node;
            best.leaf = face;
            w.tmax32 = up32(t - w.t_shift);
        }
    }
    pop_next();
}

// Two-input maximum / minimum as single instructions, without the canonicalisation of the operands that the compiler
// puts in front of fmaxf / fminf when it cannot prove them canonical (every operand here is a finite number or an
// infinity by construction).
RT_DEV float max_raw(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
RT_DEV float min_raw(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }

// The same search for a whole wave at once: one traversal, shared by all active lanes, visits every node that the
// ray of at least one lane can reach.  The current node, the stack depth and every branch of the walk are wave-uniform,
// so nodes and float32 faces always come through the scalar cache into SGPRs, the walk's control flow is scalar
// branches instead of per-lane masks, and lanes never drift apart.  It pays when the rays of the wave are close together
// (the samples of one pixel, or of neighbouring pixels): the union of their walks is then hardly longer than each one.
// Results per lane are those of mesh_collide_fast up to the choice among faces hit at exactly the same distance.
//
// Per lane: cur_tn is the entry distance of its ray into the current node (+inf when it misses the node's box; the
// lane then only follows along) and tmax32 its current search limit.  The stack (one entry per tree level: near-first
// descent) holds each lane's own entry distance, lds[level * stride], and after those cap rows (of the wave's 64 lanes)
// the child references, one per level since they are wave-uniform.
// Must be entered by the lanes of a wave together (any subset); each lane gets its own ray's answer.
template <bool ANY_HIT>
RT_DEV bool mesh_collide_packet(const DScene& sc, const DMesh& m, const Ray& r, double tmin, double tmax, int& prim, double& t_out,
                                double& u_out, double& v_out, int32_t* lds, int stride, int cap, VisitCounts& cnt) {
    FastRay f;
    const bool alive = fast_setup(m, r, tmin, tmax, f);
    if (__builtin_amdgcn_ballot_w64(alive) == 0) return false;
    RT_STAT(0, 1);
#ifdef RT_VISIT_STATS
    const long long walk_t0 = clock64();
    long long leaf_cycles = 0;
#endif
    const FastNode* nodes = sc.fast_nodes + m.fast_off;
    const double* pos = sc.prim_pos + m.prim_off * 9;
    const Tri32* tri32 = sc.prim_tri32 + m.prim_off;
    const float inf32 = __builtin_inff();
    const float tmin32 = alive ? fmaxf(0.0f, (float)(tmin - f.t_shift)) : inf32;
    float tmax32 = alive ? up32(tmax - f.t_shift) : -1.0f;  // a lane whose ray misses the root box fails every test below
    const float dx = (float)r.d.x, dy = (float)r.d.y, dz = (float)r.d.z;
    const float dlen = sqrtf(dx * dx + dy * dy + dz * dz);
    const float err_d = kErr * dlen, err_r = err_d * m.radius;
    constexpr int32_t kDone = kFastEmpty;
    int32_t* const refs = lds - (int)(threadIdx.x & 63) + cap * stride;  // the wave's own region: cap rows of `stride` = 64 lanes, then the references
    int32_t cur = 0;  // wave-uniform
    int sp = 0;       // wave-uniform
    float cur_tn = tmin32;
    bool hit = false;
    // direction signs over the lanes whose rays reach the mesh (a zero component counts by the sign slab_inverse gave it)
    const unsigned long long act = __builtin_amdgcn_ballot_w64(alive);
    const unsigned long long ngx = __builtin_amdgcn_ballot_w64(alive && f.ix < 0.0f), ngy = __builtin_amdgcn_ballot_w64(alive && f.iy < 0.0f),
                             ngz = __builtin_amdgcn_ballot_w64(alive && f.iz < 0.0f);
    const bool same = (ngx == 0 || ngx == act) && (ngy == 0 || ngy == act) && (ngz == 0 || ngz == act);
    const int oct = same ? ((ngx != 0 ? 1 : 0) | (ngy != 0 ? 2 : 0) | (ngz != 0 ? 4 : 0)) : -1;
    // Inner nodes, from `cur` down to the next leaf (left in `cur`; kDone when the walk is over): test both children, go
    // to the one most lanes reach first, push the other; with nothing reached, pop.
    // O >= 0: every ray of the wave runs into octant O (bit a set = down axis a), so which plane of a slab is entered and
    // which is left is known when the code is compiled: the packed multiply-add gets the two planes in that order (an
    // operand-select bit, no instruction) and the per-lane minimum / maximum of each pair of distances falls away,
    // 12 of a node's 36 vector instructions.  O < 0: rays of mixed octants, planes sorted per lane.
    // Only this loop exists once per octant; the leaves' float64 code below is shared.
    auto descend = [&](auto oct_tag) {
        constexpr int O = decltype(oct_tag)::value;
        while (cur >= 0) {
            const FastNode n = load_node_scalar(nodes, cur);
            RT_STAT(1, 1);
            cnt.boxes += 2;  // diagnostics only; dead code unless the caller reads the counts
            float tn[2];
            unsigned long long reach[2];  // lanes whose rays enter the child's box
#pragma unroll
            for (int k = 0; k < 2; k++) {
                bool in_k;
                float tnear;
                if (O >= 0) {
                    constexpr int sx = O >= 0 ? (O & 1) : 0, sy = O >= 0 ? ((O >> 1) & 1) : 0, sz = O >= 0 ? ((O >> 2) & 1) : 0;
                    const f32x2 x = __builtin_elementwise_fma(f32x2{n.b[k][0][sx], n.b[k][0][1 - sx]}, f32x2{f.ix, f.ix}, f32x2{f.nx, f.nx});
                    const f32x2 y = __builtin_elementwise_fma(f32x2{n.b[k][1][sy], n.b[k][1][1 - sy]}, f32x2{f.iy, f.iy}, f32x2{f.ny, f.ny});
                    const f32x2 z = __builtin_elementwise_fma(f32x2{n.b[k][2][sz], n.b[k][2][1 - sz]}, f32x2{f.iz, f.iz}, f32x2{f.nz, f.nz});
                    tnear = fmaxf(fmaxf(x[0], y[0]), max_raw(z[0], tmin32));
                    const float tfar = fminf(fminf(x[1], y[1]), min_raw(z[1], tmax32));
                    in_k = tnear <= tfar;  // an empty slot holds an inverted box, which this ordered test can never enter
                } else {
                    const f32x2 x = __builtin_elementwise_fma(f32x2{n.b[k][0][0], n.b[k][0][1]}, f32x2{f.ix, f.ix}, f32x2{f.nx, f.nx});
                    const f32x2 y = __builtin_elementwise_fma(f32x2{n.b[k][1][0], n.b[k][1][1]}, f32x2{f.iy, f.iy}, f32x2{f.ny, f.ny});
                    const f32x2 z = __builtin_elementwise_fma(f32x2{n.b[k][2][0], n.b[k][2][1]}, f32x2{f.iz, f.iz}, f32x2{f.nz, f.nz});
                    tnear = fmaxf(fmaxf(fminf(x[0], x[1]), fminf(y[0], y[1])), max_raw(fminf(z[0], z[1]), tmin32));
                    const float tfar = fminf(fminf(fmaxf(x[0], x[1]), fmaxf(y[0], y[1])), min_raw(fmaxf(z[0], z[1]), tmax32));
                    in_k = (tnear <= tfar) & (n.child[k] != kFastEmpty);  // the sorting test would enter an inverted box
                }
                reach[k] = __builtin_amdgcn_ballot_w64(in_k);
                tn[k] = in_k ? tnear : inf32;
            }
            const unsigned long long m0 = reach[0], m1 = reach[1];
            if (m0 != 0 && m1 != 0) {
                const unsigned long long first1 = __builtin_amdgcn_ballot_w64(tn[1] < tn[0]);  // lanes that reach child 1 first
                const bool swap = 2 * __popcll(first1) > __popcll(m0 | m1);
                if (sp < cap) {
                    RT_STAT(6, 1);
                    refs[sp] = swap ? n.child[0] : n.child[1];  // every lane stores the same value
                    lds[sp * stride] = __float_as_int(swap ? tn[0] : tn[1]);
                    sp++;
                } else {
                    cnt.overflow = true;  // the host sizes the stack from the deepest tree: cannot happen, but must not pass silently
                }
                cur = swap ? n.child[1] : n.child[0];
                cur_tn = swap ? tn[1] : tn[0];
            } else if ((m0 | m1) != 0) {
                cur = m1 != 0 ? n.child[1] : n.child[0];
                cur_tn = m1 != 0 ? tn[1] : tn[0];
            } else {
                cur = kDone;
                while (sp > 0) {  // pop, skipping subtrees that begin beyond every lane's current hit
                    sp--;
                    RT_STAT(5, 1);
                    const float tnear = __int_as_float(lds[sp * stride]);
                    if (__builtin_amdgcn_ballot_w64(tnear <= tmax32) != 0) {
                        cur = __builtin_amdgcn_readfirstlane(refs[sp]);
                        cur_tn = tnear;
                        break;
                    }
                }
                if (cur == kDone) return;
            }
        }
    };
    for (;;) {
        switch (oct) {
            case 0: descend(std::integral_constant<int, 0>{}); break;
            case 1: descend(std::integral_constant<int, 1>{}); break;
            case 2: descend(std::integral_constant<int, 2>{}); break;
            case 3: descend(std::integral_constant<int, 3>{}); break;
            case 4: descend(std::integral_constant<int, 4>{}); break;
            case 5: descend(std::integral_constant<int, 5>{}); break;
            case 6: descend(std::integral_constant<int, 6>{}); break;
            case 7: descend(std::integral_constant<int, 7>{}); break;
            default: descend(std::integral_constant<int, -1>{}); break;
        }
        if (cur == kDone) break;
        {  // leaf
            const uint32_t code = (uint32_t)~cur;
            const int32_t first = (int32_t)(code >> 3);
            const int32_t count = (int32_t)(code & 7u) + 1;
            RT_STAT(2, 1);
            RT_STAT(3, count);
#ifdef RT_VISIT_STATS
            const long long leaf_t0 = clock64();
#endif
            for (int32_t q = 0; q < count; q++) {
                const Tri32 t32 = load_tri32_scalar(tri32, first + q);
                const bool maybe = cur_tn <= tmax32 && !tri32_rejects(t32, f, dx, dy, dz, err_r, err_d);
                if (__builtin_amdgcn_ballot_w64(maybe) == 0) continue;
                RT_STAT(4, 1);
                if (maybe) {
                    cnt.prims++;
                    // the face is the same for every lane: its vertices come through the scalar cache into SGPRs as well
                    const RT_CONST_AS double* p = (const RT_CONST_AS double*)(pos + (int64_t)(first + q) * 9);
                    const V3 p0 = mk(p[0], p[1], p[2]), p1 = mk(p[3], p[4], p[5]), p2 = mk(p[6], p[7], p[8]);
                    double t, u, v;
                    // Among faces hit at exactly the same distance the one latest in leaf order wins, whatever the order of
                    // the walk: that is the face the reference's left-first walk keeps (shape_bvh.go:53-72: the right subtree wins
                    // unless the left hit is strictly nearer; shape_mesh.go:122 accepts t == tmax), and it makes a ray's answer independent of its wave.
                    if (mt_core(r, p0, p1, p2, 1e-8, tmin, tmax, t, u, v) && !(hit && t == t_out && first + q < prim)) {
                        hit = true;
                        tmax = t;
                        tmax32 = ANY_HIT ? -1.0f : up32(t - f.t_shift);  // any hit: this lane is finished
                        prim = first + q;
                        t_out = t; u_out = u; v_out = v;
                    }
                }
            }
#ifdef RT_VISIT_STATS
            leaf_cycles += clock64() - leaf_t0;
#endif
            if (ANY_HIT && __builtin_amdgcn_ballot_w64(tmax32 >= 0.0f) == 0) break;
        }
        cur = kDone;
        while (sp > 0) {  // pop, skipping subtrees that begin beyond every lane's current hit
            sp--;
            RT_STAT(5, 1);
            const float tnear = __int_as_float(lds[sp * stride]);
            if (__builtin_amdgcn_ballot_w64(tnear <= tmax32) != 0) {
                cur = __builtin_amdgcn_readfirstlane(refs[sp]);
                cur_tn = tnear;
                break;
            }
        }
        if (cur == kDone) break;
    }
#ifdef RT_VISIT_STATS
    RT_STAT(8, clock64() - walk_t0);
    RT_STAT(11, leaf_cycles);
#endif
    return hit;
}

// ---- pixel beams -------------------------------------------------------------------------------
//
// The 64 lanes of a primary wave trace samples of ONE pixel.  Under an orthographic camera those rays are parallel and
// start inside the pixel's footprint on the image plane, a parallelogram a fraction of a face wide: instead of walking
// the tree once per round of 64 samples (mesh_collide_packet), the tree is walked ONCE per pixel for the whole bundle --
// a "fat ray": the centre ray against boxes grown by the half-extent of the footprint -- by a single lane, while
// neighbouring lanes do the same for the neighbouring pixels of the work item.  The walk prunes nothing by distance (it
// does not know where the samples will hit); it lists the leaves the bundle can reach, nearest first, with the bundle's
// entry distance.  Every sample then tests only the faces of those leaves (beam_test), with the same float32 pre-test
// and float64 intersection as the other walks, so the hit it finds is the one they find.
//
// Why the list cannot miss a face: a sample's ray enters a (padded) box no earlier and leaves it no later, axis by
// axis, than the fat ray enters / leaves the box grown by the footprint, so whenever the per-ray slab test of
// mesh_collide_fast passes, the fat ray's passes; float32 rounding of the bundle's own numbers is covered by a margin of
// 2^-21 of the mesh radius on top of the boxes' padding.

#ifndef RT_BEAM_PIXELS
#define RT_BEAM_PIXELS 4
#endif
constexpr int kBeamPixels = RT_BEAM_PIXELS;      // pixels of a work item at most (one lane walks each)
// Leaves listed per pixel at a time (`cands`, at most kBeamCandsMax: PassDesc::beam_cands); a longer list is produced in instalments.
// The lists of a wave's pixels live in LDS, which caps the occupancy: the host sizes them so that the workgroups a CU's registers
// allow also fit its LDS (launch_beam).
#ifndef RT_BEAM_CANDS_MAX
#define RT_BEAM_CANDS_MAX 96
#endif
constexpr int kBeamCandsMax = RT_BEAM_CANDS_MAX;  // (a list's length travels in eight bits)
constexpr int kBeamStateDwords = 48;
constexpr int kBeamLeafFaces = 1 << RT_BEAM_LEAF_BITS;             // faces a leaf of the beams' tree holds at most
constexpr uint32_t kBeamLeafMask = (1u << kBeamLeafFaces) - 1u;  // one bit per face of such a leaf

// What a lane's walk keeps between instalments, in LDS (dwords): 0-2 a, 3-5 b (slab distance of a min plane p is
// p * i + a, of a max plane p * i + b), 6-8 i = 1 / direction, 9 t_lo (where the rays start, in the bundle's frame),
// 10-11 T (float64: the frame starts at o + d * T), 12 cur, 13 sp, 14 entry distance of cur, 15 the caller's (length
// of the list), 16-18 origin of the centre ray in that frame, 19-21 half-extent of the footprint (margins included),
// 22-24 direction (float32); 26-29 the footprint along the camera's image axes: centre (2), half sizes (2).  25: survivors of the pixel's list (beam_cull_item32), then those of them that the compact array
// holds; 46: the leaf at which the list goes on after them (beam_compact_prefix).
// A pyramid of rays from one point (beam_setup_pyr) uses instead: 16-18 the point in the mesh frame, 19-21 the centre of the
// box of its directions, 22-24 that box's half-widths, 26-43 per axis (1 / d_lo, 1 / d_hi and the four offsets of
// beam_child_pyr), 44 bit k set = the directions' k-th components straddle zero, 45 factor from slab parameter to distance.
//
// Sets the walk of the pixel (lx, y) up.  False: the bundle misses the mesh.
RT_DEV bool beam_setup(const DCamera& cam, int32_t dx_img, int32_t dy_img, const DMesh& m, int lx, int y, int32_t* state) {
    // camera_orthographic.go:24-46: o = from + u * (fov_w * (s - 0.5)) + v * (fov_h * (t - 0.5)), s in [lx, lx + 1] / dx, t in [y, y + 1] / dy
    const double a0 = cam.fov_w * ((double)lx / (double)dx_img - 0.5), a1 = cam.fov_w * ((double)(lx + 1) / (double)dx_img - 0.5);
    const double b0 = cam.fov_h * ((double)y / (double)dy_img - 0.5), b1 = cam.fov_h * ((double)(y + 1) / (double)dy_img - 0.5);
    double c[3], e[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const double ua = cam.u[k] * a0, ub = cam.u[k] * a1, va = cam.v[k] * b0, vb = cam.v[k] * b1;
        const double lo = cam.from[k] + fmin(ua, ub) + fmin(va, vb), hi = cam.from[k] + fmax(ua, ub) + fmax(va, vb);
        c[k] = 0.5 * (lo + hi);
        // the samples' own arithmetic rounds differently from this one: a relative 1e-12 dwarfs every such difference
        e[k] = 0.5 * (hi - lo) + 1e-12 * (fabs(cam.from[k]) + fabs(ua) + fabs(ub) + fabs(va) + fabs(vb));
    }
    // the centre ray against the root box grown by the footprint (as fast_setup clips a single ray)
    double t_in = 0, t_out = kMaxFloat;
    const double pad = (double)m.pad;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const double inv = near_reciprocal(cam.dir[k]);
        const double t0 = ((m.root_box[k] - pad - e[k]) - c[k]) * inv, t1 = ((m.root_box[3 + k] + pad + e[k]) - c[k]) * inv;
        t_in = fmax(t_in, fmin(t0, t1));
        t_out = fmin(t_out, fmax(t0, t1));
    }
    if (!(t_in <= t_out)) return false;
    const double T = t_in;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const double bc = (c[k] + cam.dir[k] * T) - m.center[k];
        const float ocf = (float)bc;
        const float ef = up32(e[k] + fabs(bc) * 1.1920929e-7 + (double)m.radius * 4.76837158203125e-7);  // + 2^-23 |o| + 2^-21 R
        const float i = slab_inverse((float)cam.dir[k]);
        state[k] = __float_as_int(-(ocf + ef) * i);
        state[3 + k] = __float_as_int(-(ocf - ef) * i);
        state[6 + k] = __float_as_int(i);
        state[16 + k] = __float_as_int(ocf);
        state[19 + k] = __float_as_int(ef);
        state[22 + k] = __float_as_int((float)cam.dir[k]);
    }
    {  // the footprint along the camera's image axes, about the mesh's centre: the origins are from + u A + v B with A in [a0, a1], B in [b0, b1]
        const V3 rel = ld3(cam.from) - ld3(m.center);
        const double ru = dot(rel, ld3(cam.u)), rv = dot(rel, ld3(cam.v));
        const double cu = ru + 0.5 * (a0 + a1), cv = rv + 0.5 * (b0 + b1);
        // + 2^-20 R: the float32 rounding of this centre and of the faces' extents (each below 2^-23 of a coordinate), u and v
        // not exactly orthonormal, the reference's own tolerance of 1e-8 in the barycentric coordinates
        const double slop = (double)m.radius * 9.5367431640625e-7 + 1e-12 * (fabs(ru) + fabs(rv) + fabs(a0) + fabs(a1) + fabs(b0) + fabs(b1));
        state[26] = __float_as_int((float)cu);
        state[27] = __float_as_int((float)cv);
        state[28] = __float_as_int(up32(0.5 * fabs(a1 - a0) + slop));
        state[29] = __float_as_int(up32(0.5 * fabs(b1 - b0) + slop));
    }
    const float t_lo = -up32(T);
    state[9] = __float_as_int(t_lo);
    state[10] = (int32_t)(__double_as_longlong(T) & 0xffffffffll);
    state[11] = (int32_t)(__double_as_longlong(T) >> 32);
    state[12] = 0;
    state[13] = 0;
    state[14] = __float_as_int(t_lo);
    return true;
}

// What beam_fill keeps in registers of a bundle, and its test of one child box: entry distance and whether the bundle can
// reach the box at all.  Parallel rays (orthographic camera): the centre ray against the box grown by the footprint.
struct BeamSlabsOrtho {
    float ax, ay, az, bx, by, bz, ix, iy, iz, t_lo;
    RT_DEV void load(const int32_t* state) {
        ax = __int_as_float(state[0]); ay = __int_as_float(state[1]); az = __int_as_float(state[2]);
        bx = __int_as_float(state[3]); by = __int_as_float(state[4]); bz = __int_as_float(state[5]);
        ix = __int_as_float(state[6]); iy = __int_as_float(state[7]); iz = __int_as_float(state[8]);
        t_lo = __int_as_float(state[9]);
    }
    RT_DEV bool child(const FastNode& n, int k, float& tnear) const {
        const float x0 = __builtin_fmaf(n.b[k][0][0], ix, ax), x1 = __builtin_fmaf(n.b[k][0][1], ix, bx);
        const float y0 = __builtin_fmaf(n.b[k][1][0], iy, ay), y1 = __builtin_fmaf(n.b[k][1][1], iy, by);
        const float z0 = __builtin_fmaf(n.b[k][2][0], iz, az), z1 = __builtin_fmaf(n.b[k][2][1], iz, bz);
        tnear = fmaxf(fmaxf(fminf(x0, x1), fminf(y0, y1)), fmaxf(fminf(z0, z1), t_lo));
        const float tfar = fminf(fminf(fmaxf(x0, x1), fmaxf(y0, y1)), fmaxf(z0, z1));
        return tnear <= tfar;
    }
};

// Rays from one point o whose (unnormalised) directions fill a box [lo, hi] per axis: a pyramid.  Some ray of it is inside
// the slab [bmin, bmax] of an axis for the parameters between the smallest and the largest of the four quotients
// (bmin - o) / lo, (bmin - o) / hi, (bmax - o) / lo, (bmax - o) / hi -- whatever the signs, as long as lo and hi have the
// same one.  When they straddle zero a ray parallel to the slab belongs to the bundle: nothing bounds the exit, and the
// bundle is inside from the larger of (bmin - o) / hi and (bmax - o) / lo on (both negative when o is inside the slab).
// Offsets a = -(o +- e) * i make each quotient one multiply-add; e grows the box by the rounding of o.
struct BeamSlabsPyr {
    float i0[3], i1[3], a00[3], a01[3], a10[3], a11[3];
    uint32_t straddle;
    RT_DEV void load(const int32_t* state) {
#pragma unroll
        for (int k = 0; k < 3; k++) {
            i0[k] = __int_as_float(state[26 + 6 * k]); i1[k] = __int_as_float(state[27 + 6 * k]);
            a00[k] = __int_as_float(state[28 + 6 * k]); a01[k] = __int_as_float(state[29 + 6 * k]);
            a10[k] = __int_as_float(state[30 + 6 * k]); a11[k] = __int_as_float(state[31 + 6 * k]);
        }
        straddle = (uint32_t)state[44];
    }
    RT_DEV bool child(const FastNode& n, int c, float& tnear) const {
        float tn = 0.0f, tf = __builtin_inff();  // the rays start at the point
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const float A = __builtin_fmaf(n.b[c][k][0], i0[k], a00[k]), B = __builtin_fmaf(n.b[c][k][0], i1[k], a01[k]);
            const float C = __builtin_fmaf(n.b[c][k][1], i0[k], a10[k]), D = __builtin_fmaf(n.b[c][k][1], i1[k], a11[k]);
            if ((straddle >> k) & 1u) {
                tn = fmaxf(tn, fmaxf(B, C));
            } else {
                tn = fmaxf(tn, fminf(fminf(A, B), fminf(C, D)));
                tf = fminf(tf, fmaxf(fmaxf(A, B), fmaxf(C, D)));
            }
        }
        tnear = tn;
        return tn <= tf;
    }
};

// Sets up the walk of the pyramid of rays from `org` through pixel (lx, y) of the image plane llc + s horizontal + t vertical
// (camera_pinhole.go:20-34; a FocusableCamera with no aperture, camera_focusable.go:26-63, is the same from LookFrom).
RT_DEV void beam_setup_pyr(const DCamera& cam, V3 org, int32_t dx_img, int32_t dy_img, const DMesh& m, int lx, int y, int32_t* state) {
    const double s0 = (double)lx / (double)dx_img, s1 = (double)(lx + 1) / (double)dx_img;
    const double t0 = (double)y / (double)dy_img, t1 = (double)(y + 1) / (double)dy_img;
    const V3 llc = ld3(cam.llc), hz = ld3(cam.horizontal), vt = ld3(cam.vertical);
    // directions are llc + s horizontal + t vertical - org: linear in (s, t), so the corners of the pixel bound every component
    const V3 wc = ((llc + hz * (0.5 * (s0 + s1))) + vt * (0.5 * (t0 + t1))) - org;
    const double lc = sqrt(dot(wc, wc));
    double lo[3], hi[3], lmin = kMaxFloat;
#pragma unroll
    for (int k = 0; k < 3; k++) { lo[k] = kMaxFloat; hi[k] = -kMaxFloat; }
#pragma unroll
    for (int c = 0; c < 4; c++) {
        const V3 w = ((llc + hz * ((c & 1) ? s1 : s0)) + vt * ((c & 2) ? t1 : t0)) - org;
        lo[0] = fmin(lo[0], w.x); hi[0] = fmax(hi[0], w.x);
        lo[1] = fmin(lo[1], w.y); hi[1] = fmax(hi[1], w.y);
        lo[2] = fmin(lo[2], w.z); hi[2] = fmax(hi[2], w.z);
        lmin = fmin(lmin, sqrt(dot(w, w)));
    }
    const double o3[3] = {org.x - m.center[0], org.y - m.center[1], org.z - m.center[2]};
    uint32_t straddle = 0;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        // the box of directions in units of the centre ray's length, widened by 2^-21 of its size (float32 rounding of the
        // bounds and of their reciprocals, and the samples' own arithmetic, all far below that)
        const double a = lo[k] / lc, b = hi[k] / lc, wide = (fabs(a) + fabs(b)) * 4.76837158203125e-7 + 1e-30;
        const float flo = -up32(-(a - wide)), fhi = up32(b + wide);
        const float of = (float)o3[k];
        const float ef = up32(fabs(o3[k]) * 4.76837158203125e-7 + (double)m.radius * 4.76837158203125e-7);  // 2^-21 (|o| + R)
        const float i0 = slab_inverse(flo), i1 = slab_inverse(fhi);
        if (flo <= 0.0f && fhi >= 0.0f) straddle |= 1u << k;
        state[26 + 6 * k] = __float_as_int(i0);
        state[27 + 6 * k] = __float_as_int(i1);
        state[28 + 6 * k] = __float_as_int(-(of + ef) * i0);  // with a box's min plane
        state[29 + 6 * k] = __float_as_int(-(of + ef) * i1);
        state[30 + 6 * k] = __float_as_int(-(of - ef) * i0);  // with its max plane
        state[31 + 6 * k] = __float_as_int(-(of - ef) * i1);
        state[16 + k] = __float_as_int(of);
        state[19 + k] = __float_as_int((float)(0.5 * (a + b)));
        state[22 + k] = __float_as_int(up32(0.5 * (b - a) + wide + fabs(0.5 * (a + b)) * 1.1920929e-7));
    }
    state[44] = (int32_t)straddle;
    // a ray's distance along its unit direction is the slab parameter times the length of its direction in these units: at
    // least lmin / lc of it
    state[45] = __float_as_int((float)(lmin / lc * (1.0 - 4.76837158203125e-7)));
    state[10] = 0;  // the frame's origin is the point itself: no shift along the rays
    state[11] = 0;
    state[12] = 0;
    state[13] = 0;
    state[14] = 0;
}

// Rewinds a walk that beam_setup / beam_setup_pyr has set up (and beam_fill may have advanced) to its start: the walk's own
// part of the state is the node it is at, the stack's height and that node's entry distance.
template <bool PYR>
RT_DEV void beam_restart(int32_t* state) {
    state[12] = 0;
    state[13] = 0;
    state[14] = PYR ? 0 : state[9];
}

// One instalment of a pixel's walk: up to `cands` leaves, (reference, entry distance, face mask) triples in cand.  The
// walk is over when state[12] == kFastEmpty afterwards.  stack: this lane's own 2 * cap dwords.  `scale` turns the entry
// parameter into a distance no ray of the bundle undercuts (1 for parallel rays).
template <typename Slabs>
RT_DEV int beam_fill(const FastNode* nodes, int32_t* state, int32_t* stack, int cap, int32_t* cand, int cands, bool& overflow, float scale
#ifdef RT_VISIT_STATS
                     , uint32_t& steps
#endif
) {
    Slabs slabs;
    slabs.load(state);
    int32_t cur = state[12];
    int sp = state[13];
    float cur_tn = __int_as_float(state[14]);
    int count = 0;
    auto pop = [&]() {
        if (sp > 0) {
            sp--;
            cur = stack[2 * sp];
            cur_tn = __int_as_float(stack[2 * sp + 1]);
        } else {
            cur = kFastEmpty;
        }
    };
    while (cur != kFastEmpty) {
        if (cur >= 0) {
#ifdef RT_VISIT_STATS
            steps++;
#endif
            const FastNode n = nodes[cur];
            float tn[2];
            bool go[2];
#pragma unroll
            for (int k = 0; k < 2; k++) go[k] = slabs.child(n, k, tn[k]) && n.child[k] != kFastEmpty;
            if (go[0] && go[1]) {
                const bool swap = tn[1] < tn[0];
                if (sp < cap) {
                    stack[2 * sp] = swap ? n.child[0] : n.child[1];
                    stack[2 * sp + 1] = __float_as_int(swap ? tn[0] : tn[1]);
                    sp++;
                } else {
                    overflow = true;  // the host sizes the stack from the deepest tree: cannot happen, but must not pass silently
                }
                cur = swap ? n.child[1] : n.child[0];
                cur_tn = swap ? tn[1] : tn[0];
            } else if (go[0] || go[1]) {
                cur = go[0] ? n.child[0] : n.child[1];
                cur_tn = go[0] ? tn[0] : tn[1];
            } else {
                pop();
            }
        } else {
            if (count == cands) break;  // the list is full: the next instalment resumes at this leaf
            cand[3 * count] = cur;
            cand[3 * count + 1] = __float_as_int(cur_tn * scale);
            cand[3 * count + 2] = (int32_t)(kBeamLeafMask >> (kBeamLeafFaces - 1 - (int)((uint32_t)~cur & (uint32_t)(kBeamLeafFaces - 1))));  // every face of the leaf, until beam_cull_item32 has looked
            count++;
            pop();
        }
    }
    state[12] = cur;
    state[13] = sp;
    state[14] = __float_as_int(cur_tn);
    return count;
}

// Turns the entry distances of a complete list into suffix minima: entry c then holds the smallest entry distance of
// the leaves c, c + 1, ... that still have a face to offer.  The walk lists leaves nearest first only roughly; with the
// minima a sample can stop at the first entry beyond its hit (nothing after it begins nearer) instead of looking at
// every leaf of the far side of the mesh.
RT_DEV void beam_suffix_min(int32_t* cand, int n) {
    float m = __builtin_inff();
    for (int c = n - 1; c >= 0; c--) {
        if (cand[3 * c + 2] != 0) m = fminf(m, __int_as_float(cand[3 * c + 1]));
        cand[3 * c + 1] = __float_as_int(m);
    }
}

// How much of a complete, thinned list the compact array holds: the survivors are appended in list order, so those of
// the leaves [0, L) fill its first slots exactly, L the first leaf whose survivors no longer fit (n when all do).  Returns
// L; n_kept = survivors of the leaves before it.  A pixel's rounds test those from the array and go on in the list at leaf L.
RT_DEV int beam_compact_prefix(const int32_t* cand, int n, int capacity, int& n_kept) {
    int kept = 0, c = 0;
    for (; c < n; c++) {
        const int more = __popc((uint32_t)cand[3 * c + 2]);
        if (kept + more > capacity) break;
        kept += more;
    }
    n_kept = kept;
    return c;
}

// A sample's search state over the instalments of its pixel's list.
struct BeamHit {
    float tmax32;  // the search limit (t) in the bundle's frame, rounded up
    int prim;      // leaf-order position of the face hit; -1: none yet
    double t;      // the closest hit so far; before the first one, the search limit
    double u, v;
};

}  // namespace rt
