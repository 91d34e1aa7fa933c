// HIP kernels of the wavefront path tracer (gfx950).
//
//   k_primary   renderPixel for the samples of a pass; the lanes of a wave share a pixel (or a few) and
//               take different samples of it: camera ray -> closest hit -> misses counted, hits on
//               emitting-only materials (Emitter, DebugUV, DebugNormal) summed over the pixel's lanes
//               and added to the film once
//               and, for the other materials, Material.Resolve right away: scattered children appended
//               (wave64 ballot compaction) to the next generation's ray queue, lit surface points to
//               the light loop's queue
//   k_trace     the same for a queued generation of scattered rays, one ray per thread
//   k_direct    one thread per (surface point, light): occlusion ray + BRDF, added to the film
//   k_resolve   film sums -> 8-bit RGBA exactly as renderPixel quantises (render.go:218-224)
#include "kernels.h"

#include <atomic>
#include <cstring>
#include <map>
#include <mutex>
#include <tuple>
#include <type_traits>

#include "dev_bvh.h"
#include "dev_shade.h"
#include "dev_beam.h"

namespace rt {

constexpr int kBlock = 256;
// Traversal kernels run one wave per workgroup: waves of a tile that misses the mesh retire early and
// a one-wave workgroup frees its slot (registers, LDS stack) at once instead of waiting for its slowest sibling.
#ifndef RT_TRACE_BLOCK
#define RT_TRACE_BLOCK 64
#endif
constexpr int kTraceBlock = RT_TRACE_BLOCK;
static_assert(kTraceBlock == 64, "k_primary maps one 8x8 pixel tile to one single-wave workgroup");

// This lane's column of the dynamic-LDS traversal stack (see PairStack) and its depth in pairs.
struct LdsStack {
    int32_t* lds;
    int stride;
    int cap;
};
extern __shared__ int32_t dyn_lds[];
// Waves per SIMD the traversal kernels are compiled for in fast mode (caps VGPRs at 512 / waves).
#ifndef RT_PACKET_WAVES
#define RT_PACKET_WAVES 4  // the same for the primary kernel of mesh-only scenes (wave-packet traversal)
#endif
#ifndef RT_PACKET_SHADE_WAVES
#define RT_PACKET_SHADE_WAVES 3  // ... and when that kernel also runs Material.Resolve (168 VGPRs; at 4 waves it spills 392 B per lane: scan-lit 4.55 -> 4.75 Grays/s)
#endif
#ifndef RT_TRACE_WAVES
#define RT_TRACE_WAVES 3  // k_trace, mesh-only scenes
#endif
#ifndef RT_TRACE_SHAPES_WAVES
#define RT_TRACE_SHAPES_WAVES 3  // k_trace for scenes without any mesh: 168 VGPRs and 136 B of scratch per lane; 2 waves (201 VGPRs, nothing spilled) measured 4 % slower on cornell and 9 % on material, 4 waves 15 % and 6 % slower than 3
#endif
#ifndef RT_SHADE_WAVES
#define RT_SHADE_WAVES 3  // k_shade
#endif
#ifndef RT_DIRECT_SHAPES_WAVES
#define RT_DIRECT_SHAPES_WAVES 4  // k_direct of scenes without any mesh: 99 VGPRs without the walk (167 with it)
#endif
#ifndef RT_DIRECT_SHAPES_RUNS_WAVES
#define RT_DIRECT_SHAPES_RUNS_WAVES 3  // ... with runs of lights per thread: 88 B of spills instead of 320; the 257-light charuco frame 54.9 -> 46.1 ms (2 waves: 57.4, 4 waves: 45.9)
#endif
#ifndef RT_DIRECT_WAVES
#define RT_DIRECT_WAVES 3  // k_direct
#endif
#ifndef RT_FAST_SHADE_WAVES
#define RT_FAST_SHADE_WAVES 2  // the primary kernel of scenes with analytic shapes when it also runs Material.Resolve (at 3 waves: 600 B of spills per lane; cornell and teapot +3 %)
#endif
#ifndef RT_FAST_WAVES
#define RT_FAST_WAVES 3
#endif
#ifndef RT_SHAPES_SHADE_WAVES
#define RT_SHAPES_SHADE_WAVES 2  // the primary kernel of scenes without any mesh, with Material.Resolve
#endif
#ifndef RT_SHAPES_WAVES
#define RT_SHAPES_WAVES 4  // ... and without (every material only emits): charuco orbit 696 frames/s at 3 waves, 735 at 4, 578 at 2
#endif

// ---- scene level intersection ---------------------------------------------------------------

struct SceneHit {
    int node, prim;
    Col c;
    // Mesh-only scenes leave the attributes of the winning face to finish_hit(): most hits (emitters) need the
    // texture coordinates only, not the normal (a square root and three divisions) or the hit point.
    int leaf;      // >= 0: position of the face in leaf order, attributes pending; c.u / c.v hold the barycentrics
    int rec_leaf;  // what a hit record carries: a mesh's face by its position in leaf order (also under transforms), -1 for the other shapes
};

template <int MODE, bool PACKET>
RT_DEV bool mesh_node_collide(const DScene& sc, int index, const Ray& r, double tmin, double tmax, Col& c, int& prim, int& leaf, VisitCounts& cnt,
                              LdsStack ls);

// Shape.Collide for one top-level shape.  Transformed shapes (shape_transformed_shape.go:26-57)
// are unwound iteratively: the ray goes through each inverse transform (direction not
// renormalised, so t is preserved), the innermost shape is intersected, and the hit point and
// normal come back through the forward transforms.
// PACKET (fast mode): the lanes of the wave trace neighbouring rays and walk a mesh together (mesh_collide_packet).
template <int MODE, bool PACKET = false, bool NO_MESH = false>
// uv_now = false: the texture coordinates of an untransformed sphere (atan2 / acos, some hundred instructions) are left
// at 0; the caller computes them with sphere_uv() from the normal if the hit wins and its material reads them.
RT_DEV bool shape_collide(const DScene& sc, int kind, int index, const Ray& r_in, double tmin, double tmax, Col& c, int& prim, int& leaf,
                          VisitCounts& cnt, LdsStack ls, bool uv_now = true) {
    Ray r = r_in;
    int chain[4];
    int nchain = 0;
    while (kind == RT_SHAPE_TRANSFORMED && nchain < 4) {
        const DTransformed& x = sc.xforms[index];
        r.o = xf_point(x.inv, r.o);
        r.d = xf_vector(x.inv, r.d);
        chain[nchain++] = index;
        kind = x.child_kind;
        index = x.child_index;
    }
    bool hit = false;
    prim = 0;
    leaf = -1;
    switch (kind) {
        case RT_SHAPE_SPHERE:
            hit = sphere_collide(sc.spheres[index], r, tmin, tmax, c);
            if (hit && (uv_now || nchain > 0)) sphere_uv(sc.spheres[index], c);  // a transformed sphere: from the normal in its own frame
            break;
        case RT_SHAPE_QUAD: hit = quad_collide(sc.quads[index], r, tmin, tmax, c); break;
        case RT_SHAPE_CYLINDER: hit = cylinder_collide(sc.cylinders[index], r, tmin, tmax, c); break;
        case RT_SHAPE_TRIANGLE: hit = tri_collide(sc.tris[index], r, tmin, tmax, c); break;
        case RT_SHAPE_MESH: if (!NO_MESH) hit = mesh_node_collide<MODE, PACKET>(sc, index, r, tmin, tmax, c, prim, leaf, cnt, ls); break;  // NO_MESH: the host has checked that there is none
        default: break;
    }
    if (hit) {
        for (int j = nchain - 1; j >= 0; j--) {
            const DTransformed& x = sc.xforms[chain[j]];
            c.at = xf_point(x.fwd, c.at);
            c.normal = unit(xf_vector(x.fwd, c.normal));
        }
    }
    return hit;
}

// Mesh.Collide for a top-level (untransformed) mesh node, attributes of the winning face included.
template <int MODE, bool PACKET>
RT_DEV bool mesh_node_collide(const DScene& sc, int index, const Ray& r, double tmin, double tmax, Col& c, int& prim, int& leaf, VisitCounts& cnt,
                              LdsStack ls) {
    const DMesh& m = sc.meshes[index];
    double t, u, v;
    int p = 0;
    bool hit;
    if (MODE == RT_MODE_REFERENCE) hit = mesh_collide_ref(sc, m, r, tmin, tmax, p, t, u, v, cnt);
    else if (PACKET) hit = mesh_collide_packet<false>(sc, m, r, tmin, tmax, p, t, u, v, ls.lds, ls.stride, ls.cap, cnt);
    else hit = mesh_collide_fast<false>(sc, m, r, tmin, tmax, p, t, u, v, ls.lds, ls.stride, ls.cap, cnt);
    if (hit) {
        const double* pos = sc.prim_pos + (m.prim_off + p) * 9;
        prim = sc.prim_face[m.prim_off + p];
        leaf = p;
        face_finish(r, ld3(pos), ld3(pos + 3), ld3(pos + 6), sc.face_uv + (m.face_off + prim) * 6, t, u, v, c);
    }
    return hit;
}

// Attributes of a pending mesh hit (shape_mesh.go:125-141): everything, or just the texture coordinates.
RT_DEV void finish_hit(const DScene& sc, const Ray& r, SceneHit& h, bool full) {
    if (h.leaf < 0) return;
    const DMesh& m = sc.meshes[sc.nodes[h.node].index];
    const double u = h.c.u, v = h.c.v;
    const double* uv6 = sc.face_uv + (m.face_off + h.prim) * 6;
    if (full) {
        const double* pos = sc.prim_pos + (m.prim_off + h.leaf) * 9;
        face_finish(r, ld3(pos), ld3(pos + 3), ld3(pos + 6), uv6, h.c.t, u, v, h.c);
    } else {
        const double w = 1 - u - v;
        h.c.u = (uv6[0] * w + uv6[2] * u) + uv6[4] * v;
        h.c.v = (uv6[1] * w + uv6[3] * u) + uv6[5] * v;
    }
    h.leaf = -1;
}

// Fast mode only: can the ray reach node i's padded bounds within [tmin, tmax]?  fmin / fmax drop the NaN of
// 0 * inf (origin on a slab plane of an axis the ray does not move along): that axis then does not constrain.
RT_DEV bool node_box_reachable(const double* b, const Ray& r, V3 inv, double tmin, double tmax) {
    const double x0 = (b[0] - r.o.x) * inv.x, x1 = (b[3] - r.o.x) * inv.x;
    const double y0 = (b[1] - r.o.y) * inv.y, y1 = (b[4] - r.o.y) * inv.y;
    const double z0 = (b[2] - r.o.z) * inv.z, z1 = (b[5] - r.o.z) * inv.z;
    const double tn = fmax(fmax(fmin(x0, x1), fmin(y0, y1)), fmax(fmin(z0, z1), tmin));
    const double tf = fmin(fmin(fmax(x0, x1), fmax(y0, y1)), fmin(fmax(z0, z1), tmax));
    return tn <= tf;
}

// The float32 form of a ray for the packed bounds tests of DScene::node_pairs.  ok = false: the scene has no such frame or
// the ray starts too far from it for float32 (a camera a long way off): the caller then uses node_box_reachable.
struct Ray32 {
    float ix, iy, iz, nx, ny, nz;
    bool ok;
};
RT_DEV Ray32 scene_ray32(const DScene& sc, const Ray& r) {
    Ray32 q;
    const V3 ol = r.o - ld3(sc.pair_center);
    const double far = fmax(fmax(fabs(ol.x), fabs(ol.y)), fabs(ol.z));
    q.ok = sc.node_pairs != nullptr && far <= 8.0 * sc.pair_radius;  // rounding the origin then moves a plane by at most 2^-21 of the radius: the padding is 2^-16
    q.ix = slab_inverse((float)r.d.x); q.iy = slab_inverse((float)r.d.y); q.iz = slab_inverse((float)r.d.z);
    q.nx = -(float)ol.x * q.ix; q.ny = -(float)ol.y * q.iy; q.nz = -(float)ol.z * q.iz;
    return q;
}
// Can the ray reach the bounds of nodes 2 * pair and 2 * pair + 1 within [0, tmax32]?  The pair comes through the scalar cache
// (the index is the same for every lane), both nodes go through each packed instruction.
RT_DEV void pair_reachable(const DScene& sc, int pair, const Ray32& q, float tmax32, bool& even, bool& odd) {
    const RT_CONST_AS f32x2* b = (const RT_CONST_AS f32x2*)(sc.node_pairs + pair);
    const f32x2 x0 = __builtin_elementwise_fma(b[0], f32x2{q.ix, q.ix}, f32x2{q.nx, q.nx}), x1 = __builtin_elementwise_fma(b[3], f32x2{q.ix, q.ix}, f32x2{q.nx, q.nx});
    const f32x2 y0 = __builtin_elementwise_fma(b[1], f32x2{q.iy, q.iy}, f32x2{q.ny, q.ny}), y1 = __builtin_elementwise_fma(b[4], f32x2{q.iy, q.iy}, f32x2{q.ny, q.ny});
    const f32x2 z0 = __builtin_elementwise_fma(b[2], f32x2{q.iz, q.iz}, f32x2{q.nz, q.nz}), z1 = __builtin_elementwise_fma(b[5], f32x2{q.iz, q.iz}, f32x2{q.nz, q.nz});
    const f32x2 zero = {0.0f, 0.0f}, lim = {tmax32, tmax32};
    const f32x2 tnear = __builtin_elementwise_max(__builtin_elementwise_max(__builtin_elementwise_min(x0, x1), __builtin_elementwise_min(y0, y1)),
                                                  __builtin_elementwise_max(__builtin_elementwise_min(z0, z1), zero));
    const f32x2 tfar = __builtin_elementwise_min(__builtin_elementwise_min(__builtin_elementwise_max(x0, x1), __builtin_elementwise_max(y0, y1)),
                                                 __builtin_elementwise_min(__builtin_elementwise_max(z0, z1), lim));
    even = tnear[0] <= tfar[0];
    odd = tnear[1] <= tfar[1];
}

RT_DEV bool material_reads_uv(const DScene& sc, int material) {
    const rt_material& m = sc.materials[material];
    if (m.kind == RT_MAT_DEBUG_UV) return true;
    if (m.kind == RT_MAT_LAMBERTIAN || m.kind == RT_MAT_EMITTER) return sc.textures[m.texture].kind != RT_TEX_UNIFORM;
    return false;  // Metal, Dielectric, DebugNormal never call Texture.At
}

// The node loop of tracePath (render.go:155-169).  MESH_ONLY: the host has checked that every
// node is a plain mesh, so the other shapes' code (and its registers) is left out.
// PACKET (fast mode): the lanes of the wave trace neighbouring rays and walk each mesh together.
// all_uv: texture coordinates also for hits whose material never reads them (validation entry point).
// NO_MESH: the host has checked that no node holds a mesh.
template <int MODE, bool MESH_ONLY = false, bool PACKET = false, bool NO_MESH = false>
RT_DEV bool scene_closest(const DScene& sc, const Ray& r, double tmin, double tmax, SceneHit& h, VisitCounts& cnt, LdsStack ls, bool all_uv = false) {
    double min_dist = tmax;
    bool hit = false;
    const bool cull = MODE == RT_MODE_FAST && !MESH_ONLY;  // a mesh clips the ray against its own root box
    V3 inv = mk(0, 0, 0);
    Ray32 q32{};
    q32.ok = false;
    if (cull) {
        q32 = scene_ray32(sc, r);
        if (!q32.ok) inv = mk(1.0 / r.d.x, 1.0 / r.d.y, 1.0 / r.d.z);
    }
    bool reach_even = true, reach_odd = true;
    h.leaf = -1;
    // Collide of node i and the reference's update rule: keep when c.t < minDist (render.go:161)
    auto visit = [&](int i, const DNode nd) {
        Col c;
        int prim, leaf;
        const bool got = shape_collide<MODE, PACKET, NO_MESH>(sc, nd.kind, nd.index, r, tmin, min_dist, c, prim, leaf, cnt, ls, false);
        if (got && c.t < min_dist) {
            min_dist = c.t;
            h.c = c;
            h.node = i;
            h.prim = prim;
            h.rec_leaf = leaf;
            hit = true;
        }
    };
    if (cull && !PACKET && sc.node_pairs != nullptr && sc.n_nodes <= 64) {
        // Unrelated rays (scattered, or a batch): each lane first collects the nodes whose bounds ITS ray enters, then
        // visits only those, in node order.  With the loop below a wave runs the intersection code of every node that any
        // of its 64 rays reaches -- nearly all of them, each with a few lanes enabled.
        if (q32.ok) {
            unsigned long long reached = 0;
            const float lim = up32(min_dist);
            for (int pair = 0; 2 * pair < sc.n_nodes; pair++) {
                bool even, odd;
                pair_reachable(sc, pair, q32, lim, even, odd);
                reached |= ((even ? 1ull : 0ull) | (odd ? 2ull : 0ull)) << (2 * pair);
            }
            if (sc.n_nodes < 64) reached &= (1ull << sc.n_nodes) - 1ull;
            // A node that holds a mesh is visited by all its lanes at the same time (the walk fetches the tree through the
            // scalar cache while the lanes stand on the same node of the same mesh); each lane takes the nodes of its list that
            // come before that one first, so the order per ray stays the node order.
            unsigned long long meshes = NO_MESH ? 0ull : sc.walk_mask;
            for (;;) {
                const int k = meshes != 0 ? __ffsll((long long)meshes) - 1 : 64;
                unsigned long long mine = k < 64 ? reached & ((1ull << k) - 1ull) : reached;
                reached &= ~mine;
                while (mine != 0) {
                    const int i = __ffsll((long long)mine) - 1;
                    mine &= mine - 1;
                    visit(i, sc.nodes[i]);
                }
                if (NO_MESH || k == 64) break;
                if ((reached >> k) & 1ull) visit(k, sc.nodes[k]);
                reached &= ~(1ull << k);
                meshes &= meshes - 1;
            }
        } else {
            for (int i = 0; i < sc.n_nodes; i++) {
                const DNode nd = sc.nodes[i];
                if (nd.kind != RT_SHAPE_MESH && !node_box_reachable(sc.node_box + (size_t)i * 6, r, inv, tmin, min_dist)) continue;
                visit(i, nd);
            }
        }
    } else
    for (int i = 0; i < sc.n_nodes; i++) {
        if (cull && q32.ok) {  // two nodes' bounds per packed test
            if ((i & 1) == 0) pair_reachable(sc, i >> 1, q32, up32(min_dist), reach_even, reach_odd);
            if (!((i & 1) ? reach_odd : reach_even)) continue;
        }
        const DNode nd = sc.nodes[i];
        if (MESH_ONLY) {  // Mesh.Collide without the attributes: finish_hit() computes what the caller needs
            const DMesh& m = sc.meshes[nd.index];
            double t, u, v;
            int p = 0;
            bool got;
            if (MODE == RT_MODE_REFERENCE) got = mesh_collide_ref(sc, m, r, tmin, min_dist, p, t, u, v, cnt);
            else if (PACKET) got = mesh_collide_packet<false>(sc, m, r, tmin, min_dist, p, t, u, v, ls.lds, ls.stride, ls.cap, cnt);
            else got = mesh_collide_fast<false>(sc, m, r, tmin, min_dist, p, t, u, v, ls.lds, ls.stride, ls.cap, cnt);
            if (got && t < min_dist) {
                min_dist = t;
                h.c.t = t; h.c.u = u; h.c.v = v;
                h.node = i;
                h.leaf = p;
                h.rec_leaf = p;
                h.prim = sc.prim_face[m.prim_off + p];
                hit = true;
            }
            continue;
        }
        if (cull && !q32.ok && nd.kind != RT_SHAPE_MESH && !node_box_reachable(sc.node_box + (size_t)i * 6, r, inv, tmin, min_dist)) continue;
        visit(i, nd);
    }
    if (!MESH_ONLY && hit) {  // the texture coordinates of the sphere that won, when something will read them
        const DNode nd = sc.nodes[h.node];
        if (nd.kind == RT_SHAPE_SPHERE && (all_uv || material_reads_uv(sc, nd.material))) sphere_uv(sc.spheres[nd.index], h.c);
    }
    return hit;
}

// Occlusion loop of ComputeDirectLighting (material_lambertian.go:34-45): any node other than
// the one whose shape was hit.  Reference mode runs the full closest-hit Collide per node as
// the reference does; fast mode stops at the first face hit.
// NO_MESH: the host has checked that no node holds a mesh; the tree walk's code and registers are left out.
template <int MODE, bool NO_MESH = false>
RT_DEV bool scene_occluded(const DScene& sc, const Ray& r, double dist, int self_node, LdsStack ls, VisitCounts& cnt) {
    const DNode self = sc.nodes[self_node];
    V3 inv = mk(0, 0, 0);
    Ray32 q32{};
    q32.ok = false;
    if (MODE == RT_MODE_FAST) {
        q32 = scene_ray32(sc, r);
        if (!q32.ok) inv = mk(1.0 / r.d.x, 1.0 / r.d.y, 1.0 / r.d.z);
    }
    const float dist32 = up32(dist);
    bool reach_even = true, reach_odd = true;
    if (MODE == RT_MODE_FAST && sc.node_pairs != nullptr && sc.n_nodes <= 64 && sc.n_meshes == 0 && q32.ok) {
        // as in scene_closest: each lane visits the nodes its own ray reaches (scenes without a mesh: the mesh walk below
        // wants the lanes of a wave on the same node at the same time)
        unsigned long long reached = 0;
        for (int pair = 0; 2 * pair < sc.n_nodes; pair++) {
            bool even, odd;
            pair_reachable(sc, pair, q32, dist32, even, odd);
            reached |= ((even ? 1ull : 0ull) | (odd ? 2ull : 0ull)) << (2 * pair);
        }
        if (sc.n_nodes < 64) reached &= (1ull << sc.n_nodes) - 1ull;
        while (reached != 0) {
            const int i = __ffsll((long long)reached) - 1;
            reached &= reached - 1;
            const DNode nd = sc.nodes[i];
            if (nd.kind == self.kind && nd.index == self.index) continue;
            Col c;
            int prim, leaf;
            if (shape_collide<MODE, false, NO_MESH>(sc, nd.kind, nd.index, r, kEps, dist, c, prim, leaf, cnt, ls, false)) return true;
        }
        return false;
    }
    for (int i = 0; i < sc.n_nodes; i++) {
        if (MODE == RT_MODE_FAST && q32.ok) {  // two nodes' bounds per packed test
            if ((i & 1) == 0) pair_reachable(sc, i >> 1, q32, dist32, reach_even, reach_odd);
            if (!((i & 1) ? reach_odd : reach_even)) continue;
        }
        const DNode nd = sc.nodes[i];
        if (nd.kind == self.kind && nd.index == self.index) continue;
        if (MODE == RT_MODE_FAST && !q32.ok && nd.kind != RT_SHAPE_MESH && !node_box_reachable(sc.node_box + (size_t)i * 6, r, inv, kEps, dist)) continue;
        if (MODE == RT_MODE_FAST && !NO_MESH && nd.kind == RT_SHAPE_MESH) {
            double t, u, v;
            int p;
            // (Two other walks were tried here and lost.  The four-wide one of k_walk: a third slower, also on the points of
            // scattered rays alone.  The wave-packet one of the camera rays, for the points a primary pass queued, whose shadow
            // rays to one light run side by side: also a third slower; this walk already fetches a node the whole wave stands
            // on once, through the scalar cache.)
            if (mesh_collide_fast<true>(sc, sc.meshes[nd.index], r, kEps, dist, p, t, u, v, ls.lds, ls.stride, ls.cap, cnt)) return true;
            continue;
        }
        Col c;
        int prim, leaf;
        if (shape_collide<MODE, false, NO_MESH>(sc, nd.kind, nd.index, r, kEps, dist, c, prim, leaf, cnt, ls, false)) return true;
    }
    return false;
}

// ---- cameras --------------------------------------------------------------------------------

RT_DEV void undistort_normalized(const rt_camera_desc& ci, double xd, double yd, double& x, double& y) {  // camera_calibrated.go:172-199
    x = xd;
    y = yd;
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

// Camera.Cast for the four concrete cameras; the orthonormal basis is computed once on the host.
RT_DEV Ray camera_cast(const DCamera& cam, double s, double t, Rng& rng) {
    Ray r;
    if (cam.kind == RT_CAM_PINHOLE) {  // camera_pinhole.go:20-34
        V3 ip = (ld3(cam.llc) + ld3(cam.horizontal) * s) + ld3(cam.vertical) * t;
        r.o = ld3(cam.origin);
        r.d = unit(ip - r.o);
    } else if (cam.kind == RT_CAM_ORTHOGRAPHIC) {  // camera_orthographic.go:24-46
        r.o = (ld3(cam.from) + ld3(cam.u) * (cam.fov_w * (s - 0.5))) + ld3(cam.v) * (cam.fov_h * (t - 0.5));
        r.d = ld3(cam.dir);
    } else if (cam.kind == RT_CAM_FOCUSABLE) {  // camera_focusable.go:26-63
        V3 rd = in_unit_disk(rng) * cam.lens_radius;
        V3 offset = ld3(cam.u) * rd.x + ld3(cam.v) * rd.y;
        r.o = ld3(cam.from) + offset;
        V3 ip = (ld3(cam.llc) + ld3(cam.horizontal) * s) + ld3(cam.vertical) * t;
        r.d = unit(ip - r.o);
    } else {  // camera_calibrated.go:42-79
        const rt_camera_desc& ci = cam.intr;
        double upix = s * (double)ci.width, vpix = t * (double)ci.height;
        double xd = (upix - ci.cx) / ci.fx, yd = (vpix - ci.cy) / ci.fy;
        double x, y;
        undistort_normalized(ci, xd, yd, x, y);
        V3 dc = unit(mk(x, -y, -1.0));
        r.o = ld3(cam.from);
        r.d = unit((ld3(cam.u) * dc.x + ld3(cam.v) * dc.y) + ld3(cam.w) * dc.z);
    }
    return r;
}

// ---- wave64 helpers --------------------------------------------------------------------------

// Slot in an append-only queue for every lane with `want`; one atomic per wave.  Must be
// reached by all lanes of the wave.
RT_DEV uint32_t wave_append(unsigned int* counter, bool want) {
    const unsigned long long mask = __ballot(want);
    if (mask == 0) return 0;
    const int lane = threadIdx.x & 63;
    const int leader = __ffsll((long long)mask) - 1;
    unsigned int base = 0;
    if (lane == leader) base = atomicAdd(counter, (unsigned int)__popcll(mask));
    base = __shfl(base, leader, 64);
    return base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
}

RT_DEV uint32_t wave_sum(uint32_t v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// Adds per-lane tallies to the RenderStats counters, one atomic per wave and counter, on the stripe of this workgroup.
RT_DEV void wave_tally(Counters* c, uint32_t total, uint32_t left, uint32_t exceeded) {
    total = wave_sum(total);
    left = wave_sum(left);
    exceeded = wave_sum(exceeded);
    if ((threadIdx.x & 63) == 0) {
        Counters::Tally& t = c->tally[(blockIdx.x + blockIdx.y * 5u) & (kTallyStripes - 1)];
        if (total) atomicAdd(&t.total, (unsigned long long)total);
        if (left) atomicAdd(&t.left, (unsigned long long)left);
        if (exceeded) atomicAdd(&t.exceeded, (unsigned long long)exceeded);
    }
}

// The last thing a kernel does with its VisitCounts: one atomic per wave if any lane's traversal stack ran full.
RT_DEV void report_overflow(Counters* c, const VisitCounts& vc) {
    if (__ballot(vc.overflow) != 0 && (threadIdx.x & 63) == (unsigned)(__ffsll((long long)__ballot(true)) - 1)) atomicAdd(&c->errors.v, 1u);
}

RT_DEV void film_add(double* film, uint32_t pixel, V3 v) {
    double* f = film + (size_t)pixel * 3;
    if (v.x != 0) atomicAdd(f + 0, v.x);
    if (v.y != 0) atomicAdd(f + 1, v.y);
    if (v.z != 0) atomicAdd(f + 2, v.z);
}

// film_add for a whole wave: runs of neighbouring lanes that name the same pixel are summed with shuffles first
// and only the first lane of a run issues the atomics.  Queue entries of one pixel's samples lie next to each other
// (a wave appends its lanes in order), and an atomic on the film is performed at the memory side, beyond the
// per-XCD L2s, where thousands of waves queue: a run of 16 samples costs 3 of them instead of 48.  Must be reached
// by all lanes of the wave; a lane with nothing to add passes v = 0 (any pixel).
RT_DEV void film_add_wave(double* film, uint32_t pixel, V3 v) {
    if (__ballot(v.x != 0 || v.y != 0 || v.z != 0) == 0) return;  // nothing to add in the whole wave (a NaN is not zero)
    const int lane = threadIdx.x & 63;
    const uint32_t prev = __shfl_up(pixel, 1, 64);
    const bool head = lane == 0 || prev != pixel;
    const unsigned long long heads = __ballot(head);
    const unsigned long long above = lane == 63 ? 0ull : heads >> (lane + 1);
    const int after = above ? __ffsll((long long)above) - 1 : 63 - lane;  // lanes of this run that follow this one
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const double ox = __shfl_down(v.x, off, 64), oy = __shfl_down(v.y, off, 64), oz = __shfl_down(v.z, off, 64);
        if (off <= after) { v.x += ox; v.y += oy; v.z += oz; }
    }
    if (head) film_add(film, pixel, v);
}

RT_DEV bool terminal_kind(int kind) { return kind == RT_MAT_EMITTER || kind == RT_MAT_DEBUG_UV || kind == RT_MAT_DEBUG_NORMAL; }

// Resolve of the three materials that only emit (material_emitter.go:22-25,
// material_debug_uv.go:18-28, material_debug_normal.go:45-81).
RT_DEV V3 terminal_emission(const DScene& sc, const rt_material& m, const Col& c, V3 rad) {
    if (m.kind == RT_MAT_EMITTER) return texture_at(sc, m.texture, c.u, c.v) * rad;
    if (m.kind == RT_MAT_DEBUG_UV) {
        if (c.u < 0.0 || c.u > 1.0) return mk(1.0, 0.0, 0.0);
        return mk(go_min(go_max(c.u, 0.0), 1.0), go_min(go_max(c.v, 0.0), 1.0), 0.5);
    }
    V3 n = c.normal;
    double len = sqrt(n.x * n.x + n.y * n.y + n.z * n.z);
    if (!(len > 0)) len = 1;
    double r = 0.5 * (n.x / len + 1.0), g = 0.5 * (n.y / len + 1.0), b = 0.5 * (n.z / len + 1.0);
    r = r < 0 ? 0 : (r > 1 ? 1 : r);
    g = g < 0 ? 0 : (g > 1 ? 1 : g);
    b = b < 0 ? 0 : (b > 1 ? 1 : b);
    return mk(r, g, b);
}

struct PathState {
    V3 rad;
    uint32_t pixel, sample, path, depth_n;
};

// The second half of tracePath for the lanes with `live` (render.go:174-183): Material.Resolve on the hit, children
// appended to the next generation (ballot compaction, one atomic per wave and queue), the surface point appended
// for ComputeDirectLighting.  Returns the hit's emission, which the caller adds to the film.  Must be reached by
// all lanes of the wave.
// EARLY (fast mode): a child whose ray enters no node's bounds has left the scene the moment it is made: it is counted as
// traced and as having left (render.go:170-173: a miss adds nothing) and never queued -- in an open scene that is most of
// the children.  The bounds are the packed ones of scene_closest; the node the ray starts on (whose bounds hold the
// origin) is tested exactly instead when it is a plain shape, so the outcome is the one the node loop would reach.
// (Not compiled into the kernels of mesh-only scenes, which have no such bounds: there it only costs registers.)
RT_DEV bool simple_collide(const DScene& sc, int kind, int index, const Ray& r, double tmin, double tmax, Col& c);
template <bool EARLY>
RT_DEV V3 shade_lane(const DScene& sc, const PassDesc& pass, const Queues& q, int out, uint32_t stripe, bool live, const Ray& r,
                       const PathState& ps, const SceneHit& h) {
    const RayQ& next = q.rays[out];
    V3 emission = mk(0, 0, 0);
    uint32_t n_cut = 0, n_counted = 0;  // children beyond MaxRayDepth or invalid: counted, never traced
    bool want_surf = false, child_ok[2] = {false, false};
    Scatter sct;
    V3 at = mk(0, 0, 0), s_n = mk(0, 0, 0), s_wo = mk(0, 0, 0), s_alb = mk(0, 0, 0);
    uint32_t depth = 0;
    int mat_index = 0, inside = 0;
    if (live) {
        depth = ps.depth_n >> 24;
        mat_index = sc.nodes[h.node].material;
        const rt_material m = sc.materials[mat_index];
        const Col& c = h.c;
        at = c.at;
        const uint32_t x = ps.pixel % (uint32_t)pass.dx, row = ps.pixel / (uint32_t)pass.dx;
        Rng rng{pass.seed_lo, pass.seed_hi, (uint32_t)(pass.dy - 1 - (int)row) * (uint32_t)pass.dx + x, ps.sample, ps.path, ps.depth_n & 0xFFFFFFu};
        resolve(sc, m, r.d, ps.rad, c, rng, sct);
        emission = sct.emission;
#pragma unroll
        for (int k = 0; k < 2; k++) {
            if (k >= sct.n) continue;
            if (bad3(c.at) || bad3(sct.dir[k])) { n_counted++; continue; }            // render.go:147-150
            if ((int)depth + 1 > pass.max_depth) { n_counted++; n_cut++; continue; }  // render.go:151-154
            child_ok[k] = true;
        }
        if (sc.n_lights > 0 && (m.kind == RT_MAT_LAMBERTIAN || m.kind == RT_MAT_METAL || m.kind == RT_MAT_DIELECTRIC)) {
            want_surf = true;
            const LitSurface lit = lit_surface(m, c.normal, r.d, sct.albedo);
            s_n = lit.n;
            s_wo = lit.wo;
            s_alb = lit.albedo;
            inside = lit.inside;
        }
    }
    // class of a child: 1 when it can reach a mesh (scenes that mix meshes with other shapes, see DScene::walk_box)
    int cls[2] = {0, 0};
    uint32_t n_gone = 0;
#pragma unroll
    for (int k = 0; k < 2; k++) {
        if (!child_ok[k]) continue;
        const Ray cr{at, sct.dir[k]};
        bool may_walk = true;
        if (EARLY && sc.early_miss && sc.node_pairs != nullptr && sc.n_nodes <= 64) {
            const Ray32 q32 = scene_ray32(sc, cr);
            if (q32.ok) {
                unsigned long long reached = 0;
                for (int pair = 0; 2 * pair < sc.n_nodes; pair++) {
                    bool even, odd;
                    pair_reachable(sc, pair, q32, __builtin_inff(), even, odd);
                    reached |= ((even ? 1ull : 0ull) | (odd ? 2ull : 0ull)) << (2 * pair);
                }
                if (sc.n_nodes < 64) reached &= (1ull << sc.n_nodes) - 1ull;
                const DNode self = sc.nodes[h.node];
                if (((reached >> h.node) & 1ull) && self.kind != RT_SHAPE_MESH && self.kind != RT_SHAPE_TRANSFORMED) {
                    Col c;
                    if (!simple_collide(sc, self.kind, self.index, cr, kEps, kMaxFloat, c)) reached &= ~(1ull << h.node);
                }
                if (reached == 0) {
                    child_ok[k] = false;
                    n_gone++;
                    continue;
                }
                may_walk = (reached & sc.walk_mask) != 0;
            }
        }
        if (sc.walk_classes && may_walk) {
            const V3 inv = mk(1.0 / cr.d.x, 1.0 / cr.d.y, 1.0 / cr.d.z);
            cls[k] = node_box_reachable(sc.walk_box, cr, inv, kEps, kMaxFloat) ? 1 : 0;
        }
    }
    wave_tally(q.counters, n_counted + n_gone, n_gone, n_cut);
    for (int c = 0; c <= sc.walk_classes; c++) {
        // both children of every lane in one reservation per class: first children first, in lane order, then the second ones
        const unsigned long long m0 = __ballot(child_ok[0] && cls[0] == c), m1 = __ballot(child_ok[1] && cls[1] == c);
        const int lane = threadIdx.x & 63;
        const uint32_t n0 = (uint32_t)__popcll(m0), n1 = (uint32_t)__popcll(m1);
        const uint32_t out_stripe = (uint32_t)c * kQueueStripes + stripe;
        uint32_t base = 0;
        if (n0 + n1 != 0) {
            if (lane == 0) base = atomicAdd(&q.counters->ray_cursor[out][out_stripe].v, n0 + n1);
            base = __shfl(base, 0, 64);
        }
        const uint32_t first = out_stripe * q.ray_stripe_cap;
        const unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const uint32_t at_stripe = k == 0 ? base + (uint32_t)__popcll(m0 & below) : base + n0 + (uint32_t)__popcll(m1 & below);
            const uint32_t slot = first + at_stripe;
            if (child_ok[k] && cls[k] == c && at_stripe < q.ray_stripe_cap) {  // the host sizes the launches so that a stripe never fills up, and checks the cursors
                next.o[0][slot] = at.x; next.o[1][slot] = at.y; next.o[2][slot] = at.z;
                next.d[0][slot] = sct.dir[k].x; next.d[1][slot] = sct.dir[k].y; next.d[2][slot] = sct.dir[k].z;
                next.rad[0][slot] = sct.rad[k].x; next.rad[1][slot] = sct.rad[k].y; next.rad[2][slot] = sct.rad[k].z;
                next.pixel[slot] = ps.pixel;
                next.sample[slot] = ps.sample;
                next.path[slot] = ps.path * 2u + (uint32_t)k;
                next.depth_n[slot] = (depth + 1u) << 24;
            }
        }
    }
    const uint32_t ss_stripe = wave_append(&q.counters->surf_cursor[stripe].v, want_surf);
    const uint32_t ss = stripe * q.surf_stripe_cap + ss_stripe;
    if (want_surf && ss_stripe < q.surf_stripe_cap) {
        q.surfs.p[0][ss] = at.x; q.surfs.p[1][ss] = at.y; q.surfs.p[2][ss] = at.z;
        q.surfs.n[0][ss] = s_n.x; q.surfs.n[1][ss] = s_n.y; q.surfs.n[2][ss] = s_n.z;
        q.surfs.wo[0][ss] = s_wo.x; q.surfs.wo[1][ss] = s_wo.y; q.surfs.wo[2][ss] = s_wo.z;
        q.surfs.albedo[0][ss] = s_alb.x; q.surfs.albedo[1][ss] = s_alb.y; q.surfs.albedo[2][ss] = s_alb.z;
        q.surfs.pixel[ss] = ps.pixel;
        q.surfs.node[ss] = h.node;
        q.surfs.material[ss] = mat_index;
        q.surfs.inside[ss] = inside;
    }
    return emission;
}

// What tracePath does once the closest hit is known (render.go:170-183), up to Material.Resolve: 0 = the ray left the
// scene, 1 = hit an emitting-only material (`emission` set), 2 = the hit needs Material.Resolve (`h` complete).
RT_DEV int after_closest(const DScene& sc, const Ray& r, V3 rad, bool any, SceneHit& h, V3& emission) {
    if (!any) return 0;
    const rt_material& m = sc.materials[sc.nodes[h.node].material];
    if (terminal_kind(m.kind)) {
        finish_hit(sc, r, h, m.kind == RT_MAT_DEBUG_NORMAL);
        emission = terminal_emission(sc, m, h.c, rad);
        return 1;
    }
    finish_hit(sc, r, h, true);
    return 2;
}

// The first half of tracePath for one ray (render.go:143-173): NaN guard, closest hit, miss.
// Returns -1 = invalid ray, 0 = the ray left the scene, 1 = hit an emitting-only material
// (`emission` set), 2 = hit needs Material.Resolve (`h` set).
template <int MODE, bool MESH_ONLY, bool PACKET = false, bool NO_MESH = false>
RT_DEV int trace_one(const DScene& sc, const Ray& r, V3 rad, SceneHit& h, V3& emission, LdsStack ls, VisitCounts& cnt) {
    if (bad3(r.o) || bad3(r.d)) return -1;  // render.go:147-150
#ifdef RT_VISIT_STATS
    const long long closest_t0 = clock64();
#endif
    const bool any = scene_closest<MODE, MESH_ONLY, PACKET, NO_MESH>(sc, r, kEps, kMaxFloat, h, cnt, ls);
#ifdef RT_VISIT_STATS
    RT_STAT(12, clock64() - closest_t0);
#endif
    return after_closest(sc, r, rad, any, h, emission);
}

// ---- kernels -----------------------------------------------------------------------------------

// One work item of a primary pass: pass.block_px consecutive pixels (sub-block `sub`) of tile v, all samples of the
// pass, traced by one wave (lane p of it).  See k_primary for how lanes are dealt to samples and pixels.
template <int MODE, bool MESH_ONLY, bool SHADE, bool NO_MESH>
RT_DEV void primary_item(const DScene& sc, const DCamera& cam, const PassDesc& pass, const Queues& q, double* __restrict__ film, long long v,
                         uint32_t sub, int p, LdsStack ls, double* acc_lds, uint32_t& n_total, uint32_t& n_left, uint32_t& n_culled, uint32_t& rot,
                         VisitCounts& vc) {
    const int lanes_s = 1 << pass.log2_s;       // lanes (= samples in flight) per pixel
    const int px_per_iter = 64 >> pass.log2_s;  // pixels a wave traces at a time
    const int my_px = p >> pass.log2_s, my_s = p & (lanes_s - 1);
    const int rounds = (pass.sample_count + lanes_s - 1) >> pass.log2_s;
    const int64_t tile = pass.tile_begin + v;
    const int64_t group = tile >> 6;
    const int w = (int)(tile & 63);
    const int tile_x = ((int)(group % pass.groups_x) * 8 + (w & 7)) * 8;
    const int tile_y = ((int)(group / pass.groups_x) * 8 + (w >> 3)) * 8;
    const double inv_dx = 1.0 / (double)pass.dx, inv_dy = 1.0 / (double)pass.dy;  // for div_by: the divisions of render.go:202-210
    for (int pb = 0; pb < pass.block_px; pb += px_per_iter) {
        const int pix = (int)sub * pass.block_px + pb + my_px;  // index of this lane's pixel inside the tile, row-major
        const int lx = tile_x + (pix & 7);
        const int ly = tile_y + (pix >> 3);
        const int row = ((ly >> 4) * pass.band_world + pass.band_rank) * 16 + (ly & 15);  // image row, 0 = top
        const bool live = lx < pass.dx && ly < pass.local_rows && row < pass.dy;
        if (__ballot(live) == 0) continue;  // outside the frame
        const int y = pass.dy - 1 - row;  // the reference's y grows upward (render.go:195)
        const uint32_t pixel = (uint32_t)(row * pass.dx + lx);
        // The pixel's running sum lives in LDS (acc_lds[c * 64], this lane's column), not in registers: it is touched once
        // per sample and would otherwise be carried, or spilled, across every walk.
        acc_lds[0] = 0; acc_lds[64] = 0; acc_lds[128] = 0;
        for (int round = 0; round < rounds; round++) {
#ifdef RT_VISIT_STATS
            const long long round_t0 = clock64();
#endif
            const int k = (round << pass.log2_s) + my_s;
            const int s = pass.sample_begin + k;
            bool want = false, missed = false;
            Ray r;
            SceneHit h;
            PathState ps;
            const bool traced = live && k < pass.sample_count;
            n_total += (uint32_t)__popcll(__ballot(traced));  // wave-uniform tallies: one value per wave, not per lane
            if (traced) {
                // Opaque copies: without them the compiler hoists everything that depends only on the pixel (the first
                // Philox round, the int -> double conversions) out of this loop and then has to spill it to scratch
                // across the traversal; recomputing a dozen instructions per sample is far cheaper.
                int lxv = lx, yv = y;
                asm volatile("" : "+v"(lxv), "+v"(yv));
                Rng rng{pass.seed_lo, pass.seed_hi, (uint32_t)(yv * pass.dx + lxv), (uint32_t)s, 1u, 0u};
                double fs, ft;
                if (pass.spp == 1) {  // render.go:202-210
                    fs = div_by((double)lxv + 0.5, (double)pass.dx, inv_dx);
                    ft = div_by((double)yv + 0.5, (double)pass.dy, inv_dy);
                } else {
                    fs = div_by((double)lxv + rng.float64(), (double)pass.dx, inv_dx);
                    ft = div_by((double)yv + rng.float64(), (double)pass.dy, inv_dy);
                }
                r = camera_cast(cam, fs, ft, rng);
#ifdef RT_VISIT_STATS
                RT_STAT(13, clock64() - round_t0);
#endif
                ps.rad = mk(1, 1, 1);
                ps.pixel = pixel;
                ps.sample = (uint32_t)s;
                ps.path = 1;
                ps.depth_n = rng.n;
                V3 em;
                const int what = trace_one<MODE, MESH_ONLY, MODE == RT_MODE_FAST, NO_MESH>(sc, r, mk(1, 1, 1), h, em, ls, vc);
                if (what == 1) { acc_lds[0] += em.x; acc_lds[64] += em.y; acc_lds[128] += em.z; }
                want = what == 2;
                missed = what == 0;
            }
            n_left += (uint32_t)__popcll(__ballot(missed));
            if (SHADE) {
                // successive rounds of a wave append to successive stripes: over a launch every stripe receives its share
                const V3 em2 = shade_lane<MODE == RT_MODE_FAST && !MESH_ONLY>(sc, pass, q, 1, rot++ & (kQueueStripes - 1), want, r, ps, h);
                if (want) { acc_lds[0] += em2.x; acc_lds[64] += em2.y; acc_lds[128] += em2.z; }
            }
#ifdef RT_VISIT_STATS
            RT_STAT(9, clock64() - round_t0);
            RT_STAT(10, 1);
#endif
        }
        V3 acc = mk(acc_lds[0], acc_lds[64], acc_lds[128]);
        // sum over the lanes of a pixel (they are adjacent: xor-shuffles below lanes_s stay inside the pixel)
        for (int off = lanes_s >> 1; off > 0; off >>= 1) {
            acc.x += __shfl_xor(acc.x, off, 64);
            acc.y += __shfl_xor(acc.y, off, 64);
            acc.z += __shfl_xor(acc.z, off, 64);
        }
        if (live && my_s == 0) {  // the only writer of this pixel during the launch
            double* f = film + (size_t)pixel * 3;
            if (acc.x != 0) f[0] += acc.x;
            if (acc.y != 0) f[1] += acc.y;
            if (acc.z != 0) f[2] += acc.z;
        }
    }
}

// renderPixel (render.go:186-217) for one pass.  Persistent single-wave workgroups pull work items from a
// queue until the pass is drained.
//
// Lanes are dealt to SAMPLES first: a wave traces 2^pass.log2_s samples of each of 64 >> log2_s neighbouring
// pixels at a time (with 64 or more samples per pass: all 64 lanes on one pixel).  Samples of one pixel differ by
// a sub-pixel jitter only, so their rays walk the same BVH nodes and test the same faces: the lanes of the wave
// stay converged through the traversal and its float64 face tests, and read the same cache lines.  The per-pixel
// sum is a shuffle reduction over the pixel's lanes; one lane adds it to the film (no atomics: a pixel belongs to
// exactly one item of a launch).
//
// An item is pass.block_px consecutive pixels of an 8x8 tile (all the pass's samples of them): small enough that
// the slowest items (silhouette tiles cost >10x the mean) cannot leave most of the chip idle at the end of the
// launch.  Tiles are numbered group by group (8x8 tiles = 64x64 pixels); group g belongs to the queue of XCD g % 8,
// so the rays an XCD has in flight share BVH nodes in its L2, and a wave whose own queue is empty takes items from
// the other XCDs' queues.  The queue cursors only grow, so every wave reaches the exit after finding all eight
// queues exhausted.
// SHADE = false: the host has checked that every material only emits, so no hit ever needs Material.Resolve and
// its code (and registers) is left out.
template <int MODE, bool MESH_ONLY, bool SHADE, bool NO_MESH = false>
__global__ void __launch_bounds__(kTraceBlock, MODE == RT_MODE_FAST ? (MESH_ONLY ? (SHADE ? RT_PACKET_SHADE_WAVES : RT_PACKET_WAVES) : (NO_MESH ? (SHADE ? RT_SHAPES_SHADE_WAVES : RT_SHAPES_WAVES) : (SHADE ? RT_FAST_SHADE_WAVES : RT_FAST_WAVES))) : 2) k_primary(DScene sc, const DCamera* __restrict__ camp, PassDesc pass, const Queues* __restrict__ qp, double* __restrict__ film) {
    // The camera and the queue descriptors are read through pointers where they are used: as by-value arguments
    // their ~100 dwords would sit in SGPRs for the whole kernel and push other scalars out into VGPR lanes.
    const DCamera& cam = *camp;
    const Queues& q = *qp;
    Counters* const counters = q.counters;
    const int p = (int)threadIdx.x;
    const int64_t n_tiles = pass.n_pixels >> 6;
    const uint32_t n_sub = 64u / (uint32_t)pass.block_px;  // items per tile
    const uint32_t xcd = blockIdx.x & 7;  // workgroup b is dispatched to XCD b % 8
    uint32_t n_total = 0, n_left = 0, n_culled = 0;
    uint32_t rot = blockIdx.x;  // stripe of the append queues this wave's next round writes to
    VisitCounts vc{0, 0};
    // Draws the next item of one queue: tile number (or -1 when that queue is exhausted) and sub-block.
    auto draw = [&](uint32_t owner, long long& tile_out, uint32_t& sub_out) {
        const uint32_t item = atomicAdd(&counters->tile_cursor[owner].v, 1u);
        const uint32_t idx = item / n_sub;
        long long cand = -1;
        if (pass.order) {
            if (idx < pass.queue_len[owner]) cand = pass.order[pass.queue_off[owner] + idx];
        } else {
            cand = ((long long)(idx >> 6) * 8 + owner) * 64 + (idx & 63);
        }
        tile_out = (cand >= 0 && cand < n_tiles) ? cand : -1;
        sub_out = item - idx * n_sub;
    };
    // The item after the current one is drawn from the wave's own queue while the current one is traced, so the
    // round trip of the atomic (and of the order look-up) is off the critical path.
    long long ahead_v = -1;
    uint32_t ahead_sub = 0;
    if (p == 0) draw(xcd, ahead_v, ahead_sub);
    for (;;) {
        long long v = -1;
        uint32_t sub = 0;
        if (p == 0) {
            if (ahead_v >= 0) {
                v = ahead_v;
                sub = ahead_sub;
                draw(xcd, ahead_v, ahead_sub);
            } else {  // own queue exhausted (cursors only grow): take from the other XCDs' queues
                for (uint32_t k = 1; k < 8 && v < 0; k++) draw((xcd + k) & 7, v, sub);
            }
        }
        v = __shfl(v, 0, 64);
        if (v < 0) break;
        const long long t_start = pass.cost ? clock64() : 0;
        sub = __shfl(sub, 0, 64);
        primary_item<MODE, MESH_ONLY, SHADE, NO_MESH>(sc, cam, pass, q, film, v, sub, p, LdsStack{dyn_lds + threadIdx.x, kTraceBlock, sc.stack_pairs},
                                             reinterpret_cast<double*>(dyn_lds + pass.acc_offset) + threadIdx.x, n_total, n_left, n_culled, rot, vc);
        if (pass.cost && p == 0) atomicAdd(&pass.cost[v], (unsigned int)(clock64() - t_start));
    }
    if (p == 0) {  // the tallies are per wave already
        Counters::Tally& t = counters->tally[blockIdx.x & (kTallyStripes - 1)];
        if (n_total) atomicAdd(&t.total, (unsigned long long)n_total);
        if (n_left) atomicAdd(&t.left, (unsigned long long)n_left);
        if (n_culled) atomicAdd(&t.culled, (unsigned long long)n_culled);
    }
    report_overflow(counters, vc);
}

// ---- pixel beams: the primary kernel of orthographic frames of one emitting mesh ------------------------------------
//
// renderPixel (render.go:186-217) + tracePath (render.go:142-183) for a scene of one untransformed mesh whose material
// only emits, under an orthographic camera, 64 lanes on the samples of one pixel: the tree is walked once per pixel for
// the bundle of its rays ("pixel beams", dev_bvh.h), the rounds test the listed faces only, against tables prepared
// for the camera's direction (dev_beam.h).  The persistent single-wave workgroups and their per-XCD queues are
// k_primary's.
#ifndef RT_BEAM_WAVES
#define RT_BEAM_WAVES 5  // measured on the 4K scan (items of 4 pixels, 12 compact faces): 4 waves 85.5 ms, 5 waves 78.5 ms, 6 waves (8 compact faces) 90.3 ms; at a later state of the kernel (8 compact faces, 128 B of scratch at 5 waves) 60.3 / 55.9 / 63.1
#endif

// The tables of dev_beam.h for direction (dx, dy, dz), or (PYR) for rays from the point (dx, dy, dz): one thread per face, leaf order.
template <bool PYR>
__global__ void __launch_bounds__(kBlock) k_beam_prepare(DScene sc, double dx, double dy, double dz, V3 axis_u, V3 axis_v, BeamFace32* __restrict__ f32,
                                                         BeamFace64* __restrict__ f64, double* __restrict__ uv) {
    const DMesh& m = sc.meshes[sc.nodes[0].index];
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= m.n_faces) return;
    const double* pos = sc.prim_pos + (m.prim_off + i) * 9;
    const V3 p0 = ld3(pos), p1 = ld3(pos + 3), p2 = ld3(pos + 6), d = mk(dx, dy, dz);
    const double* uv_src = sc.face_uv + (m.face_off + sc.prim_face[m.prim_off + i]) * 6;
#pragma unroll
    for (int k = 0; k < 6; k++) uv[i * 6 + k] = uv_src[k];
    if (PYR) {
        // shape_mesh.go:99-120 for a ray from the point d: what does not depend on the ray's direction
        const V3 edge1 = p1 - p0;
        const V3 edge2 = p2 - p0;
        const V3 s = d - p0;
        const V3 q = cross(s, edge1);
        BeamFace64 o;
        st3(o.p0, s); st3(o.e1, edge1); st3(o.e2, edge2); st3(o.h, q);
        o.f = dot(edge2, q);
        o.pad = 0;
        f64[i] = o;
        const Tri32 t = sc.prim_tri32[m.prim_off + i];
        BeamFace32 r;
        {
#pragma clang fp contract(fast)
            const float ox = (float)(dx - m.center[0]), oy = (float)(dy - m.center[1]), oz = (float)(dz - m.center[2]);
            const float sx = ox - t.p0[0], sy = oy - t.p0[1], sz = oz - t.p0[2];
            // U = s . (d x e2) = d . (e2 x s), V = d . (s x e1), a = e1 . (d x e2) = d . (e2 x e1)
            r.p0[0] = t.e2[1] * sz - t.e2[2] * sy; r.p0[1] = t.e2[2] * sx - t.e2[0] * sz; r.p0[2] = t.e2[0] * sy - t.e2[1] * sx;
            r.h[0] = sy * t.e1[2] - sz * t.e1[1]; r.h[1] = sz * t.e1[0] - sx * t.e1[2]; r.h[2] = sx * t.e1[1] - sy * t.e1[0];
            r.g[0] = t.e2[1] * t.e1[2] - t.e2[2] * t.e1[1]; r.g[1] = t.e2[2] * t.e1[0] - t.e2[0] * t.e1[2]; r.g[2] = t.e2[0] * t.e1[1] - t.e2[1] * t.e1[0];
            // margins of tri32_rejects for directions of length up to 1.01, with the largest |component| of s in the place of the
            // mesh radius (the rays' origin need not lie inside the mesh's bounds)
            const float reach = fmaxf(fmaxf(fabsf(ox), fabsf(oy)), fabsf(oz)) + m.radius;
            const float err_d = kErr * 1.01f, err_r = err_d * reach;
            r.neg_eu = err_r * t.len2; r.neg_ev = err_r * t.len1; r.lim = err_d * t.len1 * t.len2;
            r.umin = r.vmin = -__builtin_inff();  // (not read for rays from a point)
            r.umax = r.vmax = __builtin_inff();
        }
        f32[i] = r;
        return;
    }
    // shape_mesh.go:99-110, as mt_core computes it for every ray
    const V3 edge1 = p1 - p0;
    const V3 edge2 = p2 - p0;
    const V3 h = cross(d, edge2);
    const double a = dot(edge1, h);
    const bool never = a > -1e-8 && a < 1e-8;
    const double f = 1 / a;
    BeamFace64 o;
    st3(o.p0, p0); st3(o.e1, edge1); st3(o.e2, edge2); st3(o.h, h);
    o.f = never ? __longlong_as_double(0x7ff8000000000000ll) : f;
    o.pad = 0;
    f64[i] = o;
    const Tri32 t = sc.prim_tri32[m.prim_off + i];
    BeamFace32 q;
    {
#pragma clang fp contract(fast)
        const float fx = (float)dx, fy = (float)dy, fz = (float)dz;
        const float hx = fy * t.e2[2] - fz * t.e2[1], hy = fz * t.e2[0] - fx * t.e2[2], hz = fx * t.e2[1] - fy * t.e2[0];
        const float gx = t.e1[1] * fz - t.e1[2] * fy, gy = t.e1[2] * fx - t.e1[0] * fz, gz = t.e1[0] * fy - t.e1[1] * fx;  // e1 x d
        const float a32 = t.e1[0] * hx + t.e1[1] * hy + t.e1[2] * hz;
        const float err_d = kErr * sqrtf(fx * fx + fy * fy + fz * fz), err_r = err_d * m.radius;
        const float eu = err_r * t.len2, ev = err_r * t.len1, ea = err_d * t.len1 * t.len2;
        const float aa = fabsf(a32), sg = a32 < 0 ? -1.0f : 1.0f;
        const float inf = __builtin_inff();
        q.p0[0] = t.p0[0]; q.p0[1] = t.p0[1]; q.p0[2] = t.p0[2];
        q.h[0] = sg * hx; q.h[1] = sg * hy; q.h[2] = sg * hz;
        q.g[0] = sg * gx; q.g[1] = sg * gy; q.g[2] = sg * gz;
        if (never) {  // rejected before any ray is looked at
            q.h[0] = q.h[1] = q.h[2] = q.g[0] = q.g[1] = q.g[2] = 0.0f;
            q.neg_eu = inf; q.neg_ev = inf; q.lim = -inf;
        } else if (!(aa > ea)) {  // sign of a uncertain in float32 (grazing): float64 decides
            q.neg_eu = -inf; q.neg_ev = -inf; q.lim = inf;
        } else {
            q.neg_eu = -eu; q.neg_ev = -ev; q.lim = aa + eu + ev + ea;
        }
    }
    {  // extent along the camera's image axes, float64, rounded outward
        const V3 c = ld3(m.center);
        const double u0 = dot(p0 - c, axis_u), u1 = dot(p1 - c, axis_u), u2 = dot(p2 - c, axis_u);
        const double v0 = dot(p0 - c, axis_v), v1 = dot(p1 - c, axis_v), v2 = dot(p2 - c, axis_v);
        q.umin = -up32(-fmin(u0, fmin(u1, u2))); q.umax = up32(fmax(u0, fmax(u1, u2)));
        q.vmin = -up32(-fmin(v0, fmin(v1, v2))); q.vmax = up32(fmax(v0, fmax(v1, v2)));
    }
    f32[i] = q;
}

// The tree walks of `walk_px` consecutive pixels of tile v (from pixel `first` of the tile on), a lane each: more lanes walk at a
// time than an item has pixels, and the items of the batch follow one after the other (k_beam).  Every lane leaves its walk's
// state (what is read of it later: kBeamKeptDwords) and the first instalment of its list in `out`, this wave's stretch of device
// memory: kBeamWalkPixels states, then as many lists of `cands` triples.  Dynamic LDS, all of which is free at this point: the
// lanes' stacks (2 * cap dwords each); a walking lane's state is private to it.  `defer`: a pixel whose list goes on beyond its
// first instalment is put into the launch's queue of such pixels (long_px, counters->long_px) instead; true: at least one was.
constexpr int kBeamWalkPixels = 32;
// What of a walk's state the item's thinning and rounds read (dev_bvh.h): T, the list's length and flags, the bundle's test
// parameters -- kBeamKeptDwords per pixel, in this order.
constexpr int kBeamKeptDwords = 16;
__device__ constexpr int kBeamKeptOrtho[13] = {10, 11, 15, 16, 17, 18, 19, 20, 21, 26, 27, 28, 29};
__device__ constexpr int kBeamKeptPyr[10] = {10, 11, 15, 19, 20, 21, 22, 23, 24, 45};
template <bool PYR>
RT_DEV bool beam_walk_batch(const DScene& sc, const DCamera& cam, const PassDesc& pass, long long v, int first, int walk_px, int32_t* __restrict__ out, int p,
                            int cap, bool defer, Counters* __restrict__ counters, unsigned long long* __restrict__ long_px, VisitCounts& vc) {
    const int64_t tile = pass.tile_begin + v;
    const int64_t group = tile >> 6;
    const int w = (int)(tile & 63);
    const int tile_x = ((int)(group % pass.groups_x) * 8 + (w & 7)) * 8;
    const int tile_y = ((int)(group / pass.groups_x) * 8 + (w >> 3)) * 8;
    const int cands = pass.beam_cands;
    const DMesh& m = sc.meshes[sc.nodes[0].index];
    const FastNode* const tree = sc.fast_nodes + m.beam_off;
    typedef typename std::conditional<PYR, BeamSlabsPyr, BeamSlabsOrtho>::type Slabs;
    bool handed_on = false;
    if (p < walk_px) {
        int32_t st[kBeamStateDwords];
        int32_t* const list = out + kBeamWalkPixels * kBeamKeptDwords + p * cands * 3;
        const int pix = first + p;
        const int lx = tile_x + (pix & 7), ly = tile_y + (pix >> 3);
        const int row = ((ly >> 4) * pass.band_world + pass.band_rank) * 16 + (ly & 15);
        int32_t found = 0;
        if (lx < pass.dx && ly < pass.local_rows && row < pass.dy) {
            bool reaches = true;
            if (PYR) beam_setup_pyr(cam, cam.kind == RT_CAM_PINHOLE ? ld3(cam.origin) : ld3(cam.from), pass.dx, pass.dy, m, lx, pass.dy - 1 - row, st);
            else reaches = beam_setup(cam, pass.dx, pass.dy, m, lx, pass.dy - 1 - row, st);
            if (reaches) {
#ifdef RT_VISIT_STATS
                uint32_t steps = 0;
                found = beam_fill<Slabs>(tree, st, dyn_lds + p * 2 * cap, cap, list, cands, vc.overflow, PYR ? __int_as_float(st[45]) : 1.0f, steps);
#else
                found = beam_fill<Slabs>(tree, st, dyn_lds + p * 2 * cap, cap, list, cands, vc.overflow, PYR ? __int_as_float(st[45]) : 1.0f);
#endif
                if (st[12] != kFastEmpty) {  // the list goes on: this pixel's rounds walk it in instalments
                    found |= 0x100;
                    if (defer) {  // such a pixel takes milliseconds: not one of several of an item, but an item of its own for the wave that draws next (k_beam)
                        const uint32_t slot = (uint32_t)(atomicAdd(&counters->long_px.head_tail, 1ull << 32) >> 32);
                        if (slot < pass.long_cap) {
                            atomicExch(&long_px[slot], ((unsigned long long)pass.beam_serial << 32) | (unsigned long long)(uint32_t)(v * 64 + pix));
                            found = 0x200;  // nothing listed, and beam_item leaves the pixel alone
                            handed_on = true;
                        }
                    }
                }
            }
        }
        st[15] = found;
        if (PYR) {
#pragma unroll
            for (int k = 0; k < 10; k++) out[p * kBeamKeptDwords + k] = st[kBeamKeptPyr[k]];
        } else {
#pragma unroll
            for (int k = 0; k < 13; k++) out[p * kBeamKeptDwords + k] = st[kBeamKeptOrtho[k]];
        }
    }
    // what the lanes stored is read by other lanes (beam_item), and the LDS is laid out anew
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
    return __ballot(handed_on) != 0;
}

// One work item: block_px (at most kBeamPixels) consecutive pixels of tile v, all samples of the pass.  Dynamic LDS:
// the walking lanes' stacks (lane j: 2 * cap dwords at j * 2 * cap), then at pass.beam_offset the lists (kBeamPixels x
// pass.beam_cands triples), the walks' states, kBeamCullDwords for beam_cull_item32 and the pixels' compact face arrays.
// PYR: the rays come from one point (pinhole camera; focusable camera without aperture) instead of sharing a direction.
// The walks of the item's pixels have been done by beam_walk_batch: `walked` is this wave's stretch of device memory with their
// states and lists, pixel `walked_first` of the batch being the item's first.
template <bool IMG, bool PYR>
RT_DEV void beam_item(const DScene& sc, const DCamera& cam, const PassDesc& pass, const BeamTables& bt, double* __restrict__ film, long long v,
                      uint32_t sub, int block_px, const int32_t* __restrict__ walked, int walked_first, int p, int cap, uint32_t& n_total, uint32_t& n_left,
                      uint32_t& n_culled, VisitCounts& vc) {
    const int rounds = (pass.sample_count + 63) >> 6;
    const int64_t tile = pass.tile_begin + v;
    const int64_t group = tile >> 6;
    const int w = (int)(tile & 63);
    const int tile_x = ((int)(group % pass.groups_x) * 8 + (w & 7)) * 8;
    const int tile_y = ((int)(group / pass.groups_x) * 8 + (w >> 3)) * 8;
    const double inv_dx = 1.0 / (double)pass.dx, inv_dy = 1.0 / (double)pass.dy;  // for div_by: the divisions of render.go:202-210
    int32_t* const beam_cand = dyn_lds + pass.beam_offset;
    const int cands = pass.beam_cands;
    int32_t* const beam_state = beam_cand + kBeamPixels * cands * 3;
    int32_t* const beam_compact = beam_state + kBeamPixels * kBeamStateDwords + kBeamCullDwords;
    const DMesh& m = sc.meshes[sc.nodes[0].index];
    const FastNode* const tree = sc.fast_nodes + m.beam_off;
    typedef typename std::conditional<PYR, BeamSlabsPyr, BeamSlabsOrtho>::type Slabs;
    const V3 org = cam.kind == RT_CAM_PINHOLE ? ld3(cam.origin) : ld3(cam.from);  // PYR: where every ray starts
    // sets the walk of pixel (lx, y) up in st; false: its bundle misses the mesh
    auto setup = [&](int lx, int y, int32_t* st) -> bool {
        if (PYR) {
            beam_setup_pyr(cam, org, pass.dx, pass.dy, m, lx, y, st);
            return true;
        }
        return beam_setup(cam, pass.dx, pass.dy, m, lx, y, st);
    };
    // the pixels' walk states and the first instalments of their lists, from the batch's stretch of device memory
    {
        const int32_t* const w_state = walked + walked_first * kBeamKeptDwords;
        if (p < block_px * kBeamKeptDwords) {  // (kBeamPixels * kBeamKeptDwords = 64: a lane per dword)
            const int j = p / kBeamKeptDwords, k = p - j * kBeamKeptDwords;
            if (k < (PYR ? 10 : 13)) beam_state[j * kBeamStateDwords + (PYR ? kBeamKeptPyr[k < 10 ? k : 0] : kBeamKeptOrtho[k < 13 ? k : 0])] = w_state[p];
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const int32_t* const w_list = walked + kBeamWalkPixels * kBeamKeptDwords + walked_first * cands * 3;
        for (int j = 0; j < block_px; j++) {
            const int n3 = 3 * (__builtin_amdgcn_readfirstlane(beam_state[j * kBeamStateDwords + 15]) & 0xff);
            for (int k = p; k < n3; k += 64) beam_cand[j * cands * 3 + k] = w_list[j * cands * 3 + k];
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    beam_cull_item32<PYR>(bt.f32, beam_state, beam_cand, cands, beam_state + kBeamPixels * kBeamStateDwords, beam_compact, block_px, p);
    if (p < block_px) {
        const int32_t found = beam_state[p * kBeamStateDwords + 15];
        if (!(found & 0x100)) {
            int n_kept = 0;
            beam_state[p * kBeamStateDwords + 46] = beam_compact_prefix(beam_cand + p * cands * 3, found & 0xff, kBeamFacesMax, n_kept);
            beam_state[p * kBeamStateDwords + 25] = n_kept;
            beam_suffix_min(beam_cand + p * cands * 3, found & 0xff);
#ifdef RT_VISIT_STATS
            if (found & 0xff) {  // leaves listed, faces they hold, faces that survive the bundle's test (per-lane atomics: diagnostics only)
                int faces_listed = 0, survivors = 0;
                for (int c = 0; c < (found & 0xff); c++) {
                    faces_listed += (int)((uint32_t)~beam_cand[(p * cands + c) * 3] & (uint32_t)(kBeamLeafFaces - 1)) + 1;
                    survivors += __popc((uint32_t)beam_cand[(p * cands + c) * 3 + 2]);
                }
                atomicAdd(&g_packet_stats[1], (unsigned long long)(found & 0xff));
                atomicAdd(&g_packet_stats[11], (unsigned long long)faces_listed);
                atomicAdd(&g_packet_stats[14], (unsigned long long)survivors);
                atomicAdd(&g_packet_stats[15], 1ull);
            }
#endif
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    beam_patch_distances(beam_state, beam_cand, cands, beam_compact, block_px, p);

    // what every hit of the item samples (wave-uniform: scalar loads)
    const rt_material& mat = sc.materials[sc.nodes[0].material];
    const uint8_t* img_px = nullptr;
    int img_w = 0, img_h = 0, img_format = 0, img_interp = 0, img_wrap = 0;
    if (IMG) {
        const rt_texture& tx = sc.textures[mat.texture];
        const DImage& im = sc.images[tx.image];
        img_px = im.px; img_w = im.w; img_h = im.h; img_format = im.format;
        img_interp = tx.interp; img_wrap = tx.wrap;
    }
    const V3 d_all = ld3(cam.dir);  // every ray of an orthographic camera has the camera's direction (camera_orthographic.go:44)
    const bool d_sound = PYR || !bad3(d_all);
    for (int pb = 0; pb < block_px; pb++) {
        // all 64 lanes are on this pixel: everything about it is wave-uniform
        const int pix = (int)sub * block_px + pb;
        const int lx = tile_x + (pix & 7);
        const int ly = tile_y + (pix >> 3);
        const int row = ((ly >> 4) * pass.band_world + pass.band_rank) * 16 + (ly & 15);  // image row, 0 = top
        if (!(lx < pass.dx && ly < pass.local_rows && row < pass.dy)) continue;          // outside the frame
        const int y = pass.dy - 1 - row;  // the reference's y grows upward (render.go:195)
        const uint32_t pixel = (uint32_t)(row * pass.dx + lx);
        int32_t* const px_state = beam_state + pb * kBeamStateDwords;
        int32_t* const px_cand = beam_cand + pb * cands * 3;
        const int32_t found = __builtin_amdgcn_readfirstlane(px_state[15]);
        if (found & 0x200) continue;  // handed on by beam_walk_batch: an item of its own
        const int beam_first = found & 0xff;
#if defined(RT_KO_LONG_CODE)  // knock-out builds (wrong pictures): the pixels with long lists skipped, and the code of their rounds left out as well
        if (found & 0x100) continue;
        const bool beam_long = false;
#elif defined(RT_KO_LONG)
        if (found & 0x100) continue;
        const bool beam_long = (found & 0x100) != 0;
#else
        const bool beam_long = (found & 0x100) != 0;
#endif
        // a complete list: its first survivors from the compact array, then (seldom reached: the list is ordered by distance) the
        // leaves from `beam_resume` on, face data from memory
        const int n_compact = __builtin_amdgcn_readfirstlane(px_state[25]);
        const int beam_resume = __builtin_amdgcn_readfirstlane(px_state[46]);
        const bool use_compact = !beam_long;
        const int32_t* const px_compact = beam_compact + pb * kBeamFacesMax * kBeamFaceDwords;
        const double beam_T = __longlong_as_double((long long)(((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane(px_state[11]) << 32) |
                                                               (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane(px_state[10])));
        // nothing listed: no ray through this pixel can hit, whatever its jitter -- and every one of them leaves the scene, provided none
        // can be unsound (render.go:147-150 drops such a ray uncounted): else they are generated and looked at like any other
        if (pass.cull_pixels && found == 0 && bt.safe_camera) {
            n_total += (uint32_t)pass.sample_count;
            n_left += (uint32_t)pass.sample_count;
            n_culled += (uint32_t)pass.sample_count;
            continue;
        }
        // origin of the bundle's float32 frame: a ray's origin in it is o + (d T - centre)
        const V3 shift = d_all * beam_T - ld3(m.center);
        const float scale = PYR ? __int_as_float(__builtin_amdgcn_readfirstlane(px_state[45])) : 1.0f;
        const double xd = (double)lx, yd = (double)y;
        const uint32_t ctr0 = (uint32_t)(y * pass.dx + lx);
        if (beam_long) {
            // The whole state of the pixel's walk, once (the batch's walk handed on only what the thinning and the rounds of other
            // pixels read); a round then only rewinds it.  In the rounds themselves the compiler moved the set-up's divisions
            // ahead of the rounds of EVERY pixel and spilled the results.
            if (p == pb) setup(lx, y, px_state);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        if (found == 0) {
            // Nothing listed (and the pixel is not settled by count): its rays are drawn, cast and looked at like any others, in
            // rounds of their own -- no list, no hit, nothing for the film -- which the rounds below, with all they carry, do not
            // slow: 5.6 % of the 4K scan frame with every tile traced, two thirds of whose pixels see nothing.
            for (int round = 0; round < rounds; round++) {
                const int k = (round << 6) + p;
                const bool traced = k < pass.sample_count;
                const int left_here = pass.sample_count - (round << 6);
                n_total += (uint32_t)(left_here < 64 ? left_here : 64);
                uint32_t k0 = pass.seed_lo, k1 = pass.seed_hi;
                asm volatile("" : "+s"(k0), "+s"(k1));
                uint32_t blk[4] = {ctr0, (uint32_t)(pass.sample_begin + k), 1u, 0u};
                philox4x32_10(blk, k0, k1);
                const double fs = div_by(xd + draw53(blk[0], blk[1]), (double)pass.dx, inv_dx);  // render.go:202-210
                const double ft = div_by(yd + draw53(blk[2], blk[3]), (double)pass.dy, inv_dy);
                V3 ray;  // what the guard looks at: the origin (parallel rays) or the direction (rays from a point)
                if (PYR) ray = unit(((ld3(cam.llc) + ld3(cam.horizontal) * fs) + ld3(cam.vertical) * ft) - org);
                else ray = (ld3(cam.from) + ld3(cam.u) * (cam.fov_w * (fs - 0.5))) + ld3(cam.v) * (cam.fov_h * (ft - 0.5));
                asm volatile("" ::"v"(ray.x), "v"(ray.y), "v"(ray.z));  // (the ray is cast whether or not anything reads it)
                unsigned long long sound_lanes = d_sound ? __ballot(traced) : 0ull;
                if (!bt.safe_camera) {  // render.go:147-150
                    asm volatile("");
                    sound_lanes = __ballot(traced && d_sound && !bad3(ray));
                }
                n_left += (uint32_t)__popcll(sound_lanes);  // render.go:170-173
            }
            continue;
        }
        V3 acc = mk(0, 0, 0);
#ifdef RT_VISIT_STATS
        const long long px_t0 = clock64();
        RT_STAT(0, 1);                                     // pixels that list something
        if (beam_long) RT_STAT(5, 1);                      // ... whose list goes on beyond one instalment
        if (!beam_long && beam_resume < beam_first) RT_STAT(6, 1);     // ... with more survivors than the compact array holds
#endif
        for (int round = 0; round < rounds; round++) {
            const int k = (round << 6) + p;
            const bool traced = k < pass.sample_count;
            const int left_here = pass.sample_count - (round << 6);
            n_total += (uint32_t)(left_here < 64 ? left_here : 64);
            // the two jitter draws: draws 0 and 1 of the stream (pixel, sample, path 1), one Philox block.  The key schedule is
            // rebuilt from the seed in every round (scalar additions): kept across the rounds it would occupy twenty scalar
            // registers that the list walk needs.
            uint32_t k0 = pass.seed_lo, k1 = pass.seed_hi;
            asm volatile("" : "+s"(k0), "+s"(k1));
            uint32_t blk[4] = {ctr0, (uint32_t)(pass.sample_begin + k), 1u, 0u};
            philox4x32_10(blk, k0, k1);
            const double fs = div_by(xd + draw53(blk[0], blk[1]), (double)pass.dx, inv_dx);  // render.go:202-210
            const double ft = div_by(yd + draw53(blk[2], blk[3]), (double)pass.dy, inv_dy);
            V3 o, d;
            float fx, fy, fz;
            if (PYR) {  // camera_pinhole.go:20-34 (camera_focusable.go:26-63 with a lens of radius 0: the same from LookFrom)
                const V3 ip = (ld3(cam.llc) + ld3(cam.horizontal) * fs) + ld3(cam.vertical) * ft;
                o = org;
                d = unit(ip - o);
                fx = (float)d.x; fy = (float)d.y; fz = (float)d.z;
            } else {  // camera_orthographic.go:24-46
                o = (ld3(cam.from) + ld3(cam.u) * (cam.fov_w * (fs - 0.5))) + ld3(cam.v) * (cam.fov_h * (ft - 0.5));
                d = d_all;
                fx = (float)(o.x + shift.x); fy = (float)(o.y + shift.y); fz = (float)(o.z + shift.z);
            }
            bool sound = traced && d_sound;
            unsigned long long sound_lanes = d_sound ? __ballot(traced) : 0ull;  // (of the comparison itself: a scalar register pair; see beam_test_compact)
            if (!bt.safe_camera) {  // render.go:147-150
                // (a branch, made so by the empty statement: as a select the three classifications run in every round of every
                // pixel although the host has nearly always ruled the case out -- 4 % of the 4K scan frame)
                asm volatile("");
                sound = sound && !(PYR ? bad3(d) : bad3(o));
                sound_lanes = __ballot(sound);
            }
            BeamHit bh{__builtin_inff(), -1, kMaxFloat, 0, 0};
            bool walking = beam_long;  // wave-uniform
            if (walking && p == pb) beam_restart<PYR>(px_state);  // this round's walk starts over
#ifdef RT_VISIT_STATS
            const long long faces_t0 = clock64();
#endif
            if (use_compact) {
                beam_test_compact<PYR>(bt, o, d, fx, fy, fz, beam_T, px_compact, n_compact, kEps, sound, sound_lanes, bh);
                if (beam_resume < beam_first)
                    beam_test_prepared<PYR>(bt, o, d, fx, fy, fz, beam_T, px_cand + 3 * beam_resume, beam_first - beam_resume, true, kEps, sound, bh);
            } else do {
                int n = beam_first;
                if (walking) {
#ifdef RT_VISIT_STATS
                    uint32_t again = 0;
                    if (p == pb) px_state[15] = beam_fill<Slabs>(tree, px_state, dyn_lds + pb * 2 * cap, cap, px_cand, cands, vc.overflow, scale, again);
#else
                    if (p == pb) px_state[15] = beam_fill<Slabs>(tree, px_state, dyn_lds + pb * 2 * cap, cap, px_cand, cands, vc.overflow, scale);
#endif
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    n = __builtin_amdgcn_readfirstlane(px_state[15]);
                }
                if (n > 0) beam_test_prepared<PYR>(bt, o, d, fx, fy, fz, beam_T, px_cand, n, !beam_long, kEps, sound, bh);
                if (walking) {
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    walking = __builtin_amdgcn_readfirstlane(px_state[12]) != kFastEmpty;
                }
            } while (walking);
#ifdef RT_VISIT_STATS
            RT_STAT(12, clock64() - faces_t0);
            RT_STAT(10, 1);
#endif
            n_left += (uint32_t)__popcll(sound_lanes & __ballot(bh.prim < 0));  // render.go:170-173
            if (sound && bh.prim >= 0) {
                if (IMG) {  // Face.Collide's texture coordinates (shape_mesh.go:131-140), Emitter.Resolve (material_emitter.go:22-25) with radiance (1, 1, 1)
                    const double* uv6 = bt.uv + (size_t)bh.prim * 6;
                    const double wgt = 1 - bh.u - bh.v;
                    const double cu = (uv6[0] * wgt + uv6[2] * bh.u) + uv6[4] * bh.v;
                    const double cv = (uv6[1] * wgt + uv6[3] * bh.u) + uv6[5] * bh.v;
                    const V3 em = image_at_uniform(img_px, img_w, img_h, img_format, img_interp, img_wrap, cu, cv);
                    acc.x += em.x; acc.y += em.y; acc.z += em.z;
                } else {  // any other emitting material, through the code every other kernel runs
                    SceneHit h;
                    h.c.t = bh.t; h.c.u = bh.u; h.c.v = bh.v;
                    h.node = 0;
                    h.leaf = bh.prim;
                    h.rec_leaf = bh.prim;
                    h.prim = sc.prim_face[m.prim_off + bh.prim];
                    V3 em;
                    const Ray r{o, d};
                    if (after_closest(sc, r, mk(1, 1, 1), true, h, em) == 1) { acc.x += em.x; acc.y += em.y; acc.z += em.z; }
                }
            }
        }
#ifdef RT_VISIT_STATS
        RT_STAT(9, clock64() - px_t0);                                       // cycles in the rounds of all pixels
        if (beam_long) RT_STAT(8, clock64() - px_t0);                        // ... of the pixels with long lists
        if (!beam_long && beam_resume < beam_first) RT_STAT(7, clock64() - px_t0);       // ... of those with too many survivors
#endif
        // sum over the 64 lanes of the pixel, in the order k_primary adds them
        for (int off = 32; off > 0; off >>= 1) {
            acc.x += __shfl_xor(acc.x, off, 64);
            acc.y += __shfl_xor(acc.y, off, 64);
            acc.z += __shfl_xor(acc.z, off, 64);
        }
        if (p == 0) {  // the only writer of this pixel during the launch
            double* f = film + (size_t)pixel * 3;
            if (acc.x != 0) f[0] += acc.x;
            if (acc.y != 0) f[1] += acc.y;
            if (acc.z != 0) f[2] += acc.z;
        }
    }
}

template <bool IMG, bool PYR, int WAVES>
__global__ void __launch_bounds__(kTraceBlock, WAVES) k_beam(DScene sc, const DCamera* __restrict__ camp, PassDesc pass, Counters* __restrict__ counters,
                                                                     BeamTables bt, double* __restrict__ film) {
    const DCamera& cam = *camp;
    const int p = (int)threadIdx.x;
    const int64_t n_tiles = pass.n_pixels >> 6;
    const uint32_t xcd = blockIdx.x & 7;  // workgroup b is dispatched to XCD b % 8
    uint32_t n_total = 0, n_left = 0, n_culled = 0;
    VisitCounts vc{0, 0};
#ifdef RT_WAVE_TIMES
    const unsigned long long wave_t0 = wall_clock64();
    unsigned long long wave_items = 0, wave_last_end = wave_t0, wave_last = 0, wave_prev = 0, wave_last_tile = 0;
#endif
    // Draws the next batch of one queue: tile number (or -1 when that queue is exhausted) and which batch of the tile.
    // look_first: when the queues run out, every wave draws from all eight, and an atomic on one address is performed only some
    // tens of millions of times a second, whereas a load is not held up by the others (it is a round trip more, though).
    const uint32_t walk_px = (uint32_t)pass.walk_px, n_batches = 64u / walk_px;  // batches per tile
    const uint32_t n_groups = (uint32_t)((n_tiles + 63) >> 6);
    // A scheduled queue ends with its cheap tiles (queue_cheap of them), which have a cursor of their own (`cheap`): they are
    // taken up when the others of ALL queues are gone, so that the batches still under way when the work runs out are short ones.
    uint32_t drained = 0;  // (lane 0's) queues known to have run out, a bit each; from bit 8: their cheap ends
    auto draw = [&](uint32_t owner, bool cheap, long long& tile_out, uint32_t& sub_out, bool look_first) {
        tile_out = -1;
        const uint32_t bit = 1u << (owner + (cheap ? 8u : 0u));
        if (drained & bit) return;
        const uint32_t n_cheap = pass.order ? pass.queue_cheap[owner] : 0u;
        const uint32_t n_dear = (pass.order ? pass.queue_len[owner] : ((n_groups + 7u - owner) >> 3) * 64u) - n_cheap;
        const uint32_t all = (cheap ? n_cheap : n_dear) * n_batches;
        unsigned int* const cursor = cheap ? &counters->cheap_cursor[owner].v : &counters->tile_cursor[owner].v;
        if (all == 0 || (look_first && __hip_atomic_load(cursor, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= all)) {
            drained |= bit;
            return;
        }
        const uint32_t item = atomicAdd(cursor, 1u);
        if (item >= all) {
            drained |= bit;
            return;
        }
        const uint32_t idx = (cheap ? n_dear : 0u) + item / n_batches;
        sub_out = item % n_batches;
        long long cand = -1;
        if (pass.order) {
            if (idx < pass.queue_len[owner]) cand = pass.order[pass.queue_off[owner] + idx];
        } else {
            cand = ((long long)(idx >> 6) * 8 + owner) * 64 + (idx & 63);
        }
        tile_out = (cand >= 0 && cand < n_tiles) ? cand : -1;
        if (tile_out < 0) drained |= bit;
    };
    // The pixels that take milliseconds each (their list of leaves goes on beyond one instalment, and every round walks the tree
    // again) are not traced by the wave that meets them in a batch: it puts them into a queue of the launch, and a wave that is
    // about to draw takes one of those first, as an item of one pixel -- so no wave gets several of them, wherever they lie in the
    // tile queues.  An entry carries the launch's number: the slot may have been reserved and not yet written.  Takes the
    // oldest entry (false: none); `ht`: the cursors as the caller has read them (glance: a load, which may be a little old;
    // look: a read-modify-write, which is performed at the memory side whatever the caches hold).
    auto glance = [&]() { return __hip_atomic_load(&counters->long_px.head_tail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    auto look_up = [&]() { return atomicAdd(&counters->long_px.head_tail, 0ull); };
    auto draw_long = [&](long long& tile_out, uint32_t& sub_out, unsigned long long ht) -> bool {
        uint32_t slot;
        for (;;) {
            slot = (uint32_t)ht;
            const uint32_t reserved = (uint32_t)(ht >> 32);
            if (slot >= (reserved < pass.long_cap ? reserved : pass.long_cap)) return false;
            const unsigned long long seen = atomicCAS(&counters->long_px.head_tail, ht, ht + 1);  // (fails whenever another wave took or added one: so many retries at most)
            if (seen == ht) break;
            ht = seen;
        }
        unsigned long long e = 0;
        for (int spin = 0; spin < (1 << 22); spin++) {  // the wave that reserved the slot writes it within a few instructions
            e = atomicOr(&bt.long_px[slot], 0ull);
            if ((uint32_t)(e >> 32) == pass.beam_serial) break;
            __builtin_amdgcn_s_sleep(8);
        }
        if ((uint32_t)(e >> 32) != pass.beam_serial) {  // (never seen; reported like a stack that ran full rather than left to hang)
            vc.overflow = 1;
            return false;
        }
        tile_out = (long long)((uint32_t)e >> 6);
        sub_out = (uint32_t)e & 63u;
        return true;
    };
    const bool may_defer = walk_px > 1 && pass.beam_serial != 0 && n_tiles <= (1ll << 25);  // (an entry holds tile * 64 + pixel in 32 bits)
    int32_t* const walked = bt.walk_scratch + (size_t)blockIdx.x * (size_t)bt.walk_stride;
    unsigned long long seen_ht = 0;  // the long pixels' cursors as read after the last draw (while the batch was traced, the load was under way)
    bool pusher = may_defer;  // (lane 0's) this wave may still add to the queue of long pixels: it has not found the tile queues empty
    bool pushed = false;      // ... and has added to it
    for (;;) {
        long long v = -1;
        uint32_t sub = 0, px = 1, wpx = 1;
        if (p == 0) {
            bool tiles_left = true;
            if (may_defer && draw_long(v, sub, seen_ht)) {
                // (px = wpx = 1)
            } else {  // a batch of this XCD's queue; when that has run out (cursors only grow), of the others'
                for (uint32_t k = 0; k < 16 && v < 0; k++) draw((xcd + k) & 7, k >= 8, v, sub, (k & 7) > 0);
                tiles_left = v >= 0;
                px = (uint32_t)pass.block_px;
                wpx = walk_px;
                if (may_defer) seen_ht = glance();
            }
            if (!tiles_left && may_defer) {
                // No batches left.  What this wave has put into the queue of long pixels is there (the fence), and whatever any wave
                // still puts there that wave will find when it comes here, so a wave that finds the queue empty may go.  But the
                // last batches' long pixels are better shared: some of the waves (512: more would only crowd the memory channel
                // of the cursors) stay until no wave can add any more, looking every few microseconds, twenty thousand times at most (a
                // quarter of a second: the wait must end by itself should this grid ever share the chip and not be resident as a whole).
                if (pusher) {
                    __threadfence();
                    atomicAdd(&counters->long_done[xcd].v, 1u);
                    pusher = false;
                }
                const bool stays = blockIdx.x < 512;
                auto all_done = [&](bool surely) {
                    uint32_t n = 0;
                    for (int k = 0; k < 8; k++)
                        n += surely ? atomicAdd(&counters->long_done[k].v, 0u) : __hip_atomic_load(&counters->long_done[k].v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    return n >= gridDim.x;
                };
                for (int look = 0; look < 20000; look++) {
                    // (a wave that has added nothing may go on a glance; one that has must see its own additions)
                    if (draw_long(v, sub, (look == 0 && pushed) ? look_up() : glance()) || !stays) break;
                    if (all_done(false) && all_done(true)) {
                        __threadfence();
                        (void)draw_long(v, sub, look_up());
                        break;
                    }
                    for (int nap = 0; nap < 4; nap++) __builtin_amdgcn_s_sleep(127);
                }
                px = wpx = 1;
            }
        }
        // lane 0's values, as scalars: everything derived from them (the item's pixels) stays in scalar registers
        const int v_lo = __builtin_amdgcn_readfirstlane((int)(v & 0xffffffffll)), v_hi = __builtin_amdgcn_readfirstlane((int)(v >> 32));
        v = (long long)(((unsigned long long)(uint32_t)v_hi << 32) | (uint32_t)v_lo);
        if (v < 0) break;
        sub = (uint32_t)__builtin_amdgcn_readfirstlane((int)sub);
        px = (uint32_t)__builtin_amdgcn_readfirstlane((int)px);
        wpx = (uint32_t)__builtin_amdgcn_readfirstlane((int)wpx);
        const long long t_start = pass.cost ? clock64() : 0;
#ifdef RT_WAVE_TIMES
        const unsigned long long item_t0 = wall_clock64();
#endif
        if (beam_walk_batch<PYR>(sc, cam, pass, v, (int)(sub * wpx), (int)wpx, walked, p, sc.stack_pairs, may_defer && wpx > 1, counters, bt.long_px, vc)) pushed = true;
        for (uint32_t part = 0; part * px < wpx; part++)
            beam_item<IMG, PYR>(sc, cam, pass, bt, film, v, sub * (wpx / px) + part, (int)px, walked, (int)(part * px), p, sc.stack_pairs, n_total, n_left, n_culled, vc);
        if (pass.cost && p == 0) atomicAdd(&pass.cost[v], (unsigned int)(clock64() - t_start));
#ifdef RT_WAVE_TIMES
        wave_items++;
        wave_last_end = wall_clock64();
        wave_prev = wave_last;
        wave_last_tile = (unsigned long long)v * 64 + sub * wpx;
        wave_last = (wave_last_end - item_t0) | ((unsigned long long)(wpx == 1 ? 1 : 0) << 32);
#endif
    }
#ifdef RT_WAVE_TIMES
    if (p == 0 && blockIdx.x < 8192) {
        g_wave_times[4 * blockIdx.x] = wave_t0;
        g_wave_times[4 * blockIdx.x + 1] = wave_last_end;   // (the end of its last item: some waves stay on to look for long pixels)
        g_wave_times[4 * blockIdx.x + 2] = wave_items | (wave_last << 16);  // items; from bit 16: the last item's duration (32 bits), then 1 if it was a long pixel
        g_wave_times[4 * blockIdx.x + 3] = (wave_prev & 0xffffffull) | ((unsigned long long)wave_last_tile << 24);  // the duration of the item before it; from bit 24: the last item's first pixel (tile * 64 + pixel)
    }
#endif
    if (p == 0) {  // the tallies are per wave already
        Counters::Tally& t = counters->tally[blockIdx.x & (kTallyStripes - 1)];
        if (n_total) atomicAdd(&t.total, (unsigned long long)n_total);
        if (n_left) atomicAdd(&t.left, (unsigned long long)n_left);
        if (n_culled) atomicAdd(&t.culled, (unsigned long long)n_culled);
    }
    report_overflow(counters, vc);
}

// tracePath for one generation of scattered rays in a scene WITHOUT any mesh, one ray per lane: closest hit over the
// nodes, Material.Resolve, children and surface point appended.  Nothing walks a tree here, so the lanes of a wave stay
// together and the hit never has to leave the registers (scenes with a mesh go through k_walk and k_shade below).
template <int MODE, int WAVES>
__global__ void __launch_bounds__(kTraceBlock, WAVES) k_trace(DScene sc, PassDesc pass, Queues q, int in, RayStripeCounts cnt, double* __restrict__ film) {
    // blockIdx.y is the stripe of rays[in] this wave reads; its children go to stripe (x + y) % 32 of their class, so
    // of the waves that share an x at most gridDim.y / 32 write to the same stripe
    const uint32_t at_stripe = blockIdx.x * kTraceBlock + threadIdx.x;
    const bool live = at_stripe < cnt.n[blockIdx.y];
    const uint32_t i = blockIdx.y * q.ray_stripe_cap + at_stripe;
    const RayQ& rays = q.rays[in];
    bool want = false;
    uint32_t n_left = 0;
    Ray r;
    SceneHit h;
    PathState ps;
    ps.pixel = 0xffffffffu;
    V3 add = mk(0, 0, 0);
    VisitCounts vc{0, 0};
    if (live) {
        r.o = mk(rays.o[0][i], rays.o[1][i], rays.o[2][i]);
        r.d = mk(rays.d[0][i], rays.d[1][i], rays.d[2][i]);
        ps.rad = mk(rays.rad[0][i], rays.rad[1][i], rays.rad[2][i]);
        ps.pixel = rays.pixel[i];
        ps.sample = rays.sample[i];
        ps.path = rays.path[i];
        ps.depth_n = rays.depth_n[i];
        V3 em;
        const int what = trace_one<MODE, false, false, true>(sc, r, ps.rad, h, em, LdsStack{dyn_lds + threadIdx.x, kTraceBlock, sc.stack_pairs}, vc);
        if (what == 0) n_left = 1;
        if (what == 1) add = em;
        want = what == 2;
    }
    const V3 em2 = shade_lane<MODE == RT_MODE_FAST>(sc, pass, q, 1 - in, (blockIdx.x + blockIdx.y) & (kQueueStripes - 1), want, r, ps, h);
    if (want) add = em2;
    film_add_wave(film, ps.pixel, add);
    wave_tally(q.counters, 0, n_left, 0);  // the rays of this launch themselves are counted by the host
    report_overflow(q.counters, vc);
}

// ---- scattered rays: k_walk (closest hits, lanes refilled as they finish) and k_shade (the rest of tracePath) ----

// The TransformedShape chain of a node's shape (shape_transformed_shape.go:26-57), innermost shape last.
struct Chain {
    int xform[4];
    int n;
    int kind, index;  // the innermost shape
};

// Takes the ray through each inverse transform (direction not renormalised, so t is preserved).
RT_DEV void chain_down(const DScene& sc, int kind, int index, Ray& r, Chain& ch) {
    ch.n = 0;
    while (kind == RT_SHAPE_TRANSFORMED && ch.n < 4) {
        const DTransformed& x = sc.xforms[index];
        r.o = xf_point(x.inv, r.o);
        r.d = xf_vector(x.inv, r.d);
        ch.xform[ch.n++] = index;
        kind = x.child_kind;
        index = x.child_index;
    }
    ch.kind = kind;
    ch.index = index;
}

// Hit point and normal back through the forward transforms.
RT_DEV void chain_up(const DScene& sc, const Chain& ch, Col& c) {
    for (int j = ch.n - 1; j >= 0; j--) {
        const DTransformed& x = sc.xforms[ch.xform[j]];
        c.at = xf_point(x.fwd, c.at);
        c.normal = unit(xf_vector(x.fwd, c.normal));
    }
}

// Collide of an innermost shape that is not a mesh, in its own frame.
RT_DEV bool simple_collide(const DScene& sc, int kind, int index, const Ray& r, double tmin, double tmax, Col& c) {
    switch (kind) {
        case RT_SHAPE_SPHERE: return sphere_collide(sc.spheres[index], r, tmin, tmax, c);
        case RT_SHAPE_QUAD: return quad_collide(sc.quads[index], r, tmin, tmax, c);
        case RT_SHAPE_CYLINDER: return cylinder_collide(sc.cylinders[index], r, tmin, tmax, c);
        case RT_SHAPE_TRIANGLE: return tri_collide(sc.tris[index], r, tmin, tmax, c);
        default: return false;
    }
}

// Lanes a wave of k_walk wants idle before it fetches new rays for them (fetching and setting up a ray -- every node
// that holds no mesh is tested there, in float64 -- runs with only those lanes enabled, walking runs with the others).
// Measured, fastest of five frames, 20 / 32 / 40 / 48 / 56 lanes: teapot 5.35 / 5.50 / 5.59 / 5.67 / 5.64 Grays/s,
// scan-lit 5.17 / 5.34 / 5.34 / 5.27 / 5.16, the 21 M-face mesh 2.01 / 2.08 / 2.09 / 2.10 / 2.03.
#ifndef RT_REFILL_MIN
#define RT_REFILL_MIN 40
#endif

// tracePath's closest hit (render.go:155-169) for one generation of scattered rays.
//
// Scattered rays of a wave have nothing in common: their walks through a mesh's tree differ in length by an order of
// magnitude, and a wave that traces 64 rays to the end runs most of the time with a quarter of its lanes.  So the
// waves are persistent and a lane that has finished its ray gets the next one: whenever RT_REFILL_MIN lanes of a wave
// are idle it reserves that many rays of a queue stripe (one atomic), hands them out, and walks on.  What a lane
// carries between rounds is its walk state (WalkLane, the LDS stack) and the best hit so far; the ray itself stays in
// the queue.  Per ray: every node that holds no mesh is tested when the ray is fetched (a short loop), then the trees
// of the nodes that do (sc.walk_nodes) are walked one after the other in rounds of walk_round().  Finished rays that
// hit something leave a record (ray, node, face, t, barycentrics) in q.hits, in blocks of kClaim entries the wave
// reserves per stripe, for k_shade; misses are only counted.  Reference mode keeps the reference's order (nodes in
// order, left-first walk) and does all of it when the ray is fetched.
// The loop ends for every wave: the claim cursors only grow, and a wave that finds every stripe drained stops fetching
// and finishes the rays it holds.
template <int MODE, bool MESH_ONLY, int WAVES>
__global__ void __launch_bounds__(kTraceBlock, WAVES) k_walk(DScene sc, Queues q, int in, RayStripeCounts cnt, int n_stripes) {
    const RayQ& rays = q.rays[in];
    const HitQ& hq = q.hits;
    Counters* const C = q.counters;
    const int lane = (int)threadIdx.x;
    const unsigned long long below = (1ull << lane) - 1ull;
    const WalkStack stack{dyn_lds + threadIdx.x, q.walk_spill + (size_t)blockIdx.x * (size_t)sc.walk_cap * 64 + threadIdx.x, sc.walk_cap};
    // wave-uniform: the blocks of rays and of hit records this wave has reserved
    int in_stripe = (int)(blockIdx.x % (unsigned)n_stripes), drained = 0;
    uint32_t in_pos = 0, in_end = 0;
    int out_stripe = (int)(blockIdx.x % (unsigned)kQueueStripes);
    uint32_t out_pos = 0, out_end = 0;
    bool exhausted = false;
    // per lane
    uint32_t slot = kNoRay;
    WalkLane w;
    w.cur = kFastEmpty;
    w.sp = 0;
    w.base = 0;
    BestHit best{kMaxFloat, -1, -1};
    int32_t walk_k = 0, walk_node = 0;  // position in sc.walk_nodes of the tree being walked, and that node
    uint32_t n_left = 0;
    VisitCounts vc{0, 0};

    // Reserves up to `want` rays of the stripe the wave stands on, moving on over drained stripes; false when every
    // stripe is drained.  A wave reserves only what its idle lanes take at once: rays held back in a private block would
    // be out of reach of the waves that run dry at the end of the launch.
    auto claim_in = [&](uint32_t want) -> bool {
        while (drained < n_stripes) {
            const uint32_t n = cnt.n[in_stripe];
            uint32_t base = 0;
            if (lane == 0 && n != 0) base = atomicAdd(&C->in_cursor[in_stripe].v, want);
            base = (uint32_t)__builtin_amdgcn_readfirstlane((int)__shfl(base, 0, 64));
            if (n != 0 && base < n) {
                in_pos = base;
                in_end = base + want < n ? base + want : n;
                drained = 0;
                return true;
            }
            in_stripe = in_stripe + 1 == n_stripes ? 0 : in_stripe + 1;
            drained++;
        }
        return false;
    };
    auto claim_out = [&]() -> bool {  // a block of hit records to fill; a full stripe sends the wave on to the next one
        for (int tries = 0; tries < kQueueStripes; tries++) {
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(&C->hit_cursor[out_stripe].v, kClaim);
            base = (uint32_t)__builtin_amdgcn_readfirstlane((int)__shfl(base, 0, 64));
            if (base + kClaim <= q.hit_stripe_cap) {
                out_pos = base;
                out_end = base + kClaim;
                return true;
            }
            out_stripe = out_stripe + 1 == kQueueStripes ? 0 : out_stripe + 1;
        }
        return false;
    };
    auto load_ray = [&]() { return Ray{mk(rays.o[0][slot], rays.o[1][slot], rays.o[2][slot]), mk(rays.d[0][slot], rays.d[1][slot], rays.d[2][slot])}; };
    // the next tree of this lane's ray that its ray can reach (fast mode); w.cur stays kFastEmpty when there is none
    auto next_walk = [&](const Ray& r) {
        while (walk_k < sc.n_walk_nodes) {
            const int node = MESH_ONLY ? walk_k : sc.walk_nodes[walk_k];
            const DNode nd = sc.nodes[node];
            Ray lr = r;
            int mesh = nd.index;
            if (!MESH_ONLY) {
                Chain ch;
                chain_down(sc, nd.kind, nd.index, lr, ch);
                mesh = ch.index;
            }
            if (walk_begin(sc.meshes[mesh], lr, kEps, best.t, w)) {
                walk_node = node;
                return;
            }
            walk_k++;
        }
    };

#ifdef RT_VISIT_STATS
    long long ws_refill = 0, ws_walk = 0, ws_out = 0, ws_rounds = 0, ws_lanes = 0, ws_iters = 0, ws_fetched = 0;
    const long long ws_begin = clock64();
#endif
    for (;;) {
#ifdef RT_VISIT_STATS
        const long long ws_t0 = clock64();
        ws_iters++;
#endif
        // ---- hand new rays to the idle lanes
        unsigned long long idle = __ballot(slot == kNoRay);
        bool fresh = false;
        if (!exhausted && (__popcll(idle) >= RT_REFILL_MIN)) {
            while (idle != 0) {
                if (in_pos == in_end && !claim_in((uint32_t)__popcll(idle))) {
                    exhausted = true;
                    break;
                }
                const uint32_t want = (uint32_t)__popcll(idle), have = in_end - in_pos;
                const uint32_t take = want < have ? want : have;
                const uint32_t rank = (uint32_t)__popcll(idle & below);
                if (slot == kNoRay && rank < take) {
                    slot = (uint32_t)in_stripe * q.ray_stripe_cap + in_pos + rank;
                    fresh = true;
                }
                in_pos += take;
                idle = __ballot(slot == kNoRay);
            }
        }
        if (fresh) {
            const Ray r = load_ray();
            best = BestHit{kMaxFloat, -1, -1};
            walk_k = 0;
            w.cur = kFastEmpty;
            w.sp = 0;
            w.base = 0;
            if (bad3(r.o) || bad3(r.d)) {
                best.node = -2;  // render.go:147-150: an invalid ray is black and counts neither as a hit nor as having left
            } else if (MODE == RT_MODE_REFERENCE) {
                for (int i = 0; i < sc.n_nodes; i++) {  // nodes in order, Shape.Collide(r, eps, minDist), keep when c.t < minDist
                    const DNode nd = sc.nodes[i];
                    Ray lr = r;
                    Chain ch;
                    chain_down(sc, nd.kind, nd.index, lr, ch);
                    if (ch.kind == RT_SHAPE_MESH) {
                        double t, u, v;
                        int p = 0;
                        if (mesh_collide_ref(sc, sc.meshes[ch.index], lr, kEps, best.t, p, t, u, v, vc) && t < best.t) best = BestHit{t, i, p};
                    } else {
                        Col c;
                        if (simple_collide(sc, ch.kind, ch.index, lr, kEps, best.t, c) && c.t < best.t) best = BestHit{c.t, i, -1};
                    }
                }
            } else {
                if (!MESH_ONLY) {  // every node without a mesh, now; a node whose padded bounds the ray misses is skipped
                    const V3 inv = mk(1.0 / r.d.x, 1.0 / r.d.y, 1.0 / r.d.z);
                    for (int i = 0; i < sc.n_nodes; i++) {
                        const DNode nd = sc.nodes[i];
                        if (nd.pad) continue;  // holds a mesh: walked below
                        if (!node_box_reachable(sc.node_box + (size_t)i * 6, r, inv, kEps, best.t)) continue;
                        Ray lr = r;
                        Chain ch;
                        chain_down(sc, nd.kind, nd.index, lr, ch);
                        Col c;
                        if (simple_collide(sc, ch.kind, ch.index, lr, kEps, best.t, c) && c.t < best.t) best = BestHit{c.t, i, -1};
                    }
                }
                next_walk(r);
            }
        }
#ifdef RT_VISIT_STATS
        const long long ws_t1 = clock64();
        ws_refill += ws_t1 - ws_t0;
        ws_fetched += __popcll(__ballot(fresh));
#endif
        // ---- walk
        unsigned long long walking = __ballot(w.cur != kFastEmpty);
        while (walking != 0) {
#ifdef RT_VISIT_STATS
            ws_rounds++;
            ws_lanes += __popcll(walking);
#endif
            if (w.cur != kFastEmpty) {
#ifdef RT_WALK_BINARY
                walk_round<false, kWalkRing>(sc, w, walk_node, kEps, best, stack, vc.overflow);
#else
#ifdef RT_VISIT_STATS
                walk_round4<false, kWalkRing>(sc, w, walk_node, kEps, best, stack, vc.overflow, vc.w);
#else
                walk_round4<false, kWalkRing>(sc, w, walk_node, kEps, best, stack, vc.overflow);
#endif
#endif
                if (w.cur == kFastEmpty && sc.n_walk_nodes > 1) {  // this tree is done: on to the next one the ray can reach
                    walk_k++;
                    next_walk(load_ray());
                }
            }
            walking = __ballot(w.cur != kFastEmpty);
            if (!exhausted && __popcll(walking) <= 64 - RT_REFILL_MIN) break;
        }
#ifdef RT_VISIT_STATS
        const long long ws_t2 = clock64();
        ws_walk += ws_t2 - ws_t1;
#endif
        // ---- finished rays: misses are counted, hits leave a record
        const bool fin = slot != kNoRay && w.cur == kFastEmpty;
        if (fin && best.node == -1) n_left++;
#ifdef RT_VISIT_STATS
        if (fin) {  // histogram of inner nodes per ray, buckets of powers of two, and the longest walk
            int b = 0;
            while ((2u << b) <= w.steps && b < 13) b++;
            atomicAdd(&g_trace_stats[(blockIdx.x + threadIdx.x) & 63u][32 + b], 1ull);
            atomicMax(&g_trace_stats[0][47], (unsigned long long)w.steps);
            w.steps = 0;
        }
#endif
        bool pending = fin && best.node >= 0;
        unsigned long long out = __ballot(pending);
        while (out != 0) {
            if (out_pos == out_end && !claim_out()) {
                if (lane == 0) atomicAdd(&C->errors.v, 1u);  // every stripe of the hit queue is full: the host sizes it so that this cannot happen
                break;
            }
            const uint32_t want = (uint32_t)__popcll(out), have = out_end - out_pos;
            const uint32_t take = want < have ? want : have;
            const uint32_t rank = (uint32_t)__popcll(out & below);
            if (pending && rank < take) {
                const size_t at = (size_t)out_stripe * q.hit_stripe_cap + out_pos + rank;
                hq.ray[at] = slot;
                hq.node[at] = best.node;
                hq.leaf[at] = best.leaf;
                hq.t[at] = best.t;
                pending = false;
            }
            out_pos += take;
            out = __ballot(pending);
        }
        if (fin) slot = kNoRay;
#ifdef RT_VISIT_STATS
        ws_out += clock64() - ws_t2;
#endif
        if (exhausted && __ballot(slot != kNoRay) == 0) break;
    }
#ifdef RT_VISIT_STATS
    RT_TSTAT(48, 1);
    RT_TSTAT(49, clock64() - ws_begin);
    RT_TSTAT(50, ws_refill);
    RT_TSTAT(51, ws_walk);
    RT_TSTAT(52, ws_out);
    RT_TSTAT(53, ws_rounds);
    RT_TSTAT(54, ws_lanes);
    RT_TSTAT(55, ws_iters);
    RT_TSTAT(56, ws_fetched);
#endif
    // the rest of the last block of hit records stays unused
    for (uint32_t i = out_pos + (uint32_t)lane; i < out_end; i += 64) hq.ray[(size_t)out_stripe * q.hit_stripe_cap + i] = kNoRay;
    wave_tally(C, 0, n_left, 0);  // the rays of this launch themselves are counted by the host
    RT_WFLUSH(vc);
    report_overflow(C, vc);
}

// Threads of the workgroups that put their items in the order of their materials before Material.Resolve (four waves).
constexpr int kSortBlock = 256;

// Sort key of a hit: the code Material.Resolve runs for it.  0 Lambertian, 1 Metal, 2 Dielectric, 3 the materials that only emit.
RT_DEV int material_key(const DScene& sc, int32_t node) {
    const int kind = sc.materials[sc.nodes[node].material].kind;
    return kind == RT_MAT_LAMBERTIAN ? 0 : (kind == RT_MAT_METAL ? 1 : (kind == RT_MAT_DIELECTRIC ? 2 : 3));
}

// Puts the items of a kSortBlock-thread workgroup in the order of their keys (0..3; 4 = this thread has no item), keeping the
// order inside a key: thread j gets back the thread whose item is j-th (-1 when there are fewer items) and the number of
// items.  Materials differ in what Resolve costs by an order of magnitude and share no code; hits arrive in whatever order
// their rays ended, so a wave that takes them as they come runs every material's code with a fraction of its lanes.  After
// this the waves of the workgroup hold one material each (at most three of the four see a boundary).  Must be reached by
// every thread of the workgroup; two barriers.
RT_DEV int block_order(int key, int& total) {
    __shared__ int s_count[4 * 4];
    __shared__ int s_src[kSortBlock];
    const int lane = (int)(threadIdx.x & 63u), wave = (int)(threadIdx.x >> 6);
    const unsigned long long m0 = __ballot(key == 0), m1 = __ballot(key == 1), m2 = __ballot(key == 2), m3 = __ballot(key == 3);
    if (lane < 4) s_count[wave * 4 + lane] = __popcll(lane == 0 ? m0 : (lane == 1 ? m1 : (lane == 2 ? m2 : m3)));
    __syncthreads();
    int base = 0, n = 0;
#pragma unroll
    for (int k = 0; k < 4; k++)
#pragma unroll
        for (int w = 0; w < 4; w++) {
            const int c = s_count[w * 4 + k];
            if (k < key || (k == key && w < wave)) base += c;
            n += c;
        }
    if (key < 4) {
        const unsigned long long mine = key == 0 ? m0 : (key == 1 ? m1 : (key == 2 ? m2 : m3));
        s_src[base + __popcll(mine & ((1ull << lane) - 1ull))] = (int)threadIdx.x;
    }
    __syncthreads();
    total = n;
    return (int)threadIdx.x < n ? s_src[threadIdx.x] : -1;
}

// The second half of tracePath (render.go:174-183) for one recorded hit per lane: the ray comes back from its queue slot,
// the attributes of the hit are computed again from (node, face, distance), then Material.Resolve, the children appended to
// the next generation, the surface point queued for the light loop, emission added to the film.  Must be reached by all
// lanes of the wave (`live`: this lane has a record).
template <int MODE, bool MESH_ONLY>
RT_DEV void shade_record(const DScene& sc, const PassDesc& pass, const Queues& q, int in, uint32_t out_stripe, bool live, uint32_t slot, int32_t node,
                         int32_t leaf, double t, double* __restrict__ film) {
    const RayQ& rays = q.rays[in];
    bool want = false;
    Ray r;
    SceneHit h;
    PathState ps;
    ps.pixel = 0xffffffffu;
    V3 add = mk(0, 0, 0);
    if (live) {
        r.o = mk(rays.o[0][slot], rays.o[1][slot], rays.o[2][slot]);
        r.d = mk(rays.d[0][slot], rays.d[1][slot], rays.d[2][slot]);
        ps.rad = mk(rays.rad[0][slot], rays.rad[1][slot], rays.rad[2][slot]);
        ps.pixel = rays.pixel[slot];
        ps.sample = rays.sample[slot];
        ps.path = rays.path[slot];
        ps.depth_n = rays.depth_n[slot];
        h.node = node;
        h.leaf = -1;
        h.prim = 0;
        const DNode nd = sc.nodes[h.node];
        const rt_material& m = sc.materials[nd.material];
        const bool terminal = terminal_kind(m.kind);
        Ray lr = r;
        Chain ch;
        if (MESH_ONLY) { ch.n = 0; ch.kind = RT_SHAPE_MESH; ch.index = nd.index; }
        else chain_down(sc, nd.kind, nd.index, lr, ch);
        bool ok = true;
        if (ch.kind == RT_SHAPE_MESH) {
            // Face.Collide on the recorded face, limited to the recorded distance: the same t again, and the barycentrics
            const DMesh& mesh = sc.meshes[ch.index];
            const double* pos = sc.prim_pos + (mesh.prim_off + leaf) * 9;
            const V3 p0 = ld3(pos), p1 = ld3(pos + 3), p2 = ld3(pos + 6);
            double t2, u, v;
            ok = mt_core(lr, p0, p1, p2, 1e-8, kEps, t, t2, u, v);
            h.prim = sc.prim_face[mesh.prim_off + leaf];
            face_finish(lr, p0, p1, p2, sc.face_uv + (mesh.face_off + h.prim) * 6, t, u, v, h.c);
        } else {
            // Any other shape: only the distance was kept.  The shape's Collide limited to that distance returns
            // the same hit again (each returns its nearest root inside [tmin, tmax]).
            ok = simple_collide(sc, ch.kind, ch.index, lr, kEps, t, h.c);
            if (ok && ch.kind == RT_SHAPE_SPHERE && (ch.n > 0 || material_reads_uv(sc, nd.material))) sphere_uv(sc.spheres[ch.index], h.c);
        }
        if (!ok) {
            atomicAdd(&q.counters->errors.v, 1u);
        } else {
            if (ch.n > 0) chain_up(sc, ch, h.c);
            if (terminal) add = terminal_emission(sc, m, h.c, ps.rad);
            else want = true;
        }
    }
    const V3 em2 = shade_lane<MODE == RT_MODE_FAST && !MESH_ONLY>(sc, pass, q, 1 - in, out_stripe, want, r, ps, h);
    if (want) add = em2;
    film_add_wave(film, ps.pixel, add);
}

// shade_record for the hits k_walk recorded.  Persistent workgroups of four waves deal the chunks of kSortBlock records among
// themselves round-robin (the work per record is even) and take the records of a chunk in the order of their materials.
template <int MODE, bool MESH_ONLY, int WAVES>
__global__ void __launch_bounds__(kSortBlock, WAVES) k_shade(DScene sc, PassDesc pass, Queues q, int in, double* __restrict__ film) {
    const HitQ& hq = q.hits;
    const Counters::Cursor* const counts = q.counters->hit_cursor;
    const uint32_t tid = threadIdx.x, wave = tid >> 6;
    uint32_t most = 0;
    for (int s = 0; s < kQueueStripes; s++) {
        const uint32_t n = counts[s].v;
        most = most > n ? most : n;
    }
    most = most < q.hit_stripe_cap ? most : q.hit_stripe_cap;
    // chunk c: records [kSortBlock (c / stripes), + kSortBlock) of stripe c % stripes; its wave w appends to stripe
    // (stripe + 4 (c / stripes) + w) % stripes, so a stripe takes the children of at most most / 64 waves
    const uint32_t chunks = (most + (uint32_t)kSortBlock - 1u) / (uint32_t)kSortBlock * (uint32_t)kQueueStripes;
    for (uint32_t c = blockIdx.x; c < chunks; c += gridDim.x) {
        const uint32_t stripe = c & (kQueueStripes - 1), k = c / kQueueStripes;
        uint32_t n_s = counts[stripe].v;
        n_s = n_s < q.hit_stripe_cap ? n_s : q.hit_stripe_cap;
        const size_t first = (size_t)stripe * q.hit_stripe_cap + (size_t)k * kSortBlock;
        int key = 4;
        if (k * (uint32_t)kSortBlock + tid < n_s && hq.ray[first + tid] != kNoRay) key = sc.sort_kinds ? material_key(sc, hq.node[first + tid]) : 0;
        int total;
        const int src = block_order(key, total);
        if (total == 0) continue;
        const bool live = src >= 0;
        const size_t at = first + (size_t)(live ? src : 0);
        shade_record<MODE, MESH_ONLY>(sc, pass, q, in, (stripe + 4u * k + wave) & (kQueueStripes - 1), live, live ? hq.ray[at] : kNoRay, live ? hq.node[at] : 0,
                                      live ? hq.leaf[at] : -1, live ? hq.t[at] : 0.0, film);
    }
}

// ComputeDirectLighting's loop body for one (surface point, light) pair
// (material_lambertian.go:23-52, material_metal.go:68-106, material_dielectric.go:128-170).
// A thread takes one surface point and a run of `per_thread` lights (all lanes of a wave are on the same light at the same
// time, so the light comes through the scalar cache): the point is fetched once for the run and what the lights add to its
// pixel is summed before the film sees it, which with hundreds of lights is what the kernel's time went into (one set of
// film atomics per (point, light) pair otherwise).
// RUNS = false is the same with one light per thread and no loop around the occlusion test (the loop costs the
// kernel 300 B of spills per lane): what scenes with a handful of lights use.
#ifndef RT_DIRECT_RUNS_WAVES
#define RT_DIRECT_RUNS_WAVES 3  // 2 waves (no spills) measured 107 ms on the 257-light frame, 3 waves (320 B of spills) 95 ms
#endif
template <int MODE, bool RUNS, bool NO_MESH = false>
__global__ void __launch_bounds__(kTraceBlock, MODE == RT_MODE_FAST ? (NO_MESH ? (RUNS ? RT_DIRECT_SHAPES_RUNS_WAVES : RT_DIRECT_SHAPES_WAVES) : (RUNS ? RT_DIRECT_RUNS_WAVES : RT_DIRECT_WAVES)) : 2) k_direct(DScene sc, Queues q, StripeCounts cnt, int per_thread_arg, uint32_t n_runs, uint32_t group, double* __restrict__ film) {
    const int per_thread = RUNS ? per_thread_arg : 1;
    // blockIdx.y = stripe of the surface-point queue.  blockIdx.x = `group` chunks of 64 points for the first run of lights, the
    // same chunks for the second run, ..., then the next group of chunks.  group = 8 (many lights): workgroups are dealt
    // round-robin to the 8 XCDs, so the runs of a chunk execute on one XCD one right after the other and all but the first
    // find the points in its L2 instead of in device memory.  group = every chunk of the stripe (a handful of lights): a run
    // is finished before the next begins; side by side the runs' film atomics on the same pixels get in each other's way,
    // which costs more than the second read of the points.
    const uint32_t stripe = blockIdx.y, run = (blockIdx.x / group) % n_runs, chunk = (blockIdx.x / (group * n_runs)) * group + blockIdx.x % group;
    const uint32_t light0 = run * (uint32_t)per_thread;
    const uint32_t light1 = !RUNS ? light0 + 1 : (light0 + (uint32_t)per_thread < (uint32_t)sc.n_lights ? light0 + (uint32_t)per_thread : (uint32_t)sc.n_lights);
    const uint32_t at_stripe = chunk * kTraceBlock + threadIdx.x;
    const bool live = at_stripe < cnt.n[stripe];
    const uint32_t i = stripe * q.surf_stripe_cap + at_stripe;
    uint32_t pixel = 0xffffffffu;
    V3 contrib = mk(0, 0, 0);
    VisitCounts vc{0, 0};
    if (live) {
        pixel = q.surfs.pixel[i];
        const rt_material& m = sc.materials[q.surfs.material[i]];  // read where it is used: a copy would sit in 18 registers across the loop
        const V3 p = mk(q.surfs.p[0][i], q.surfs.p[1][i], q.surfs.p[2][i]);
        const V3 n = mk(q.surfs.n[0][i], q.surfs.n[1][i], q.surfs.n[2][i]);
        const int node = q.surfs.node[i];
        uint32_t light = light0;
        do {
            const rt_light L = sc.lights[light];
            V3 dir;
            double dist;
            const Ray sh = shadow_ray(m, L, p, n, dir, dist);
            // What the light adds if nothing is in the way comes first: when that is exactly zero (a light behind the surface:
            // half of all pairs on a closed mesh) the sum is the same with or without it, and the occlusion test, which for such
            // a ray is a walk that starts inside the mesh, is not run at all.  (A NaN is not zero: it is still tested and added.)
            V3 wo = mk(0, 0, 0), alb = mk(0, 0, 0);  // only the one the material reads is fetched
            if (m.kind == RT_MAT_LAMBERTIAN) alb = mk(q.surfs.albedo[0][i], q.surfs.albedo[1][i], q.surfs.albedo[2][i]);
            else wo = mk(q.surfs.wo[0][i], q.surfs.wo[1][i], q.surfs.wo[2][i]);
            const V3 adds = light_contribution(m, L, n, wo, alb, q.surfs.inside[i], dir, sh.d);
            if (MODE == RT_MODE_FAST && adds.x == 0 && adds.y == 0 && adds.z == 0) continue;
            if (!scene_occluded<MODE, NO_MESH>(sc, sh, dist, node, LdsStack{dyn_lds + threadIdx.x, kTraceBlock, sc.stack_pairs}, vc)) contrib = contrib + adds;
        } while (RUNS && ++light < light1);
        RT_WFLUSH(vc);
    }
    film_add_wave(film, pixel, contrib);
    report_overflow(q.counters, vc);
}

// renderPixel's quantisation (render.go:218-224).
__global__ void __launch_bounds__(kBlock) k_resolve(const double* __restrict__ film, int32_t n_pixels, int32_t spp, uchar4* __restrict__ rgba) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n_pixels) return;
    const double r = film[(size_t)i * 3 + 0] / (double)spp;
    const double g = film[(size_t)i * 3 + 1] / (double)spp;
    const double b = film[(size_t)i * 3 + 2] / (double)spp;
    uchar4 o;
    // uint8(math.Min(255, 255.99*c)); a NaN or negative product converts to 0 on this path
    o.x = (unsigned char)go_min(255, 255.99 * r);
    o.y = (unsigned char)go_min(255, 255.99 * g);
    o.z = (unsigned char)go_min(255, 255.99 * b);
    o.w = 255;
    rgba[i] = o;
}

template <int MODE>
__global__ void __launch_bounds__(kTraceBlock) k_intersect_batch(DScene sc, const double* __restrict__ o, const double* __restrict__ d, uint64_t n,
                                                           double tmin, double tmax, rt_hit* __restrict__ hits) {
    const uint64_t i = (uint64_t)blockIdx.x * kTraceBlock + threadIdx.x;
    if (i >= n) return;
    Ray r{ld3(o + 3 * i), ld3(d + 3 * i)};
    SceneHit sh;
    VisitCounts cnt{0, 0};
    rt_hit h;
    h.node = -1;
    h.prim = 0;
    h.t = 0;
    for (int k = 0; k < 3; k++) { h.at[k] = 0; h.normal[k] = 0; }
    h.uv[0] = h.uv[1] = 0;
    if (scene_closest<MODE>(sc, r, tmin, tmax, sh, cnt, LdsStack{dyn_lds + threadIdx.x, kTraceBlock, sc.stack_pairs}, true)) {
        h.node = sh.node;
        h.prim = sh.prim;
        h.t = sh.c.t;
        st3(h.at, sh.c.at);
        st3(h.normal, sh.c.normal);
        h.uv[0] = sh.c.u;
        h.uv[1] = sh.c.v;
    }
    h.boxes_tested = cnt.boxes;
    h.prims_tested = cnt.prims;
    hits[i] = h;
}

__global__ void __launch_bounds__(kBlock) k_cast_batch(DCamera cam, const double* __restrict__ st, const uint32_t* __restrict__ ps, uint64_t n,
                                                      uint32_t seed_lo, uint32_t seed_hi, double* __restrict__ o, double* __restrict__ d) {
    const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    Rng rng{seed_lo, seed_hi, ps[2 * i], ps[2 * i + 1], 1u, 0u};
    Ray r = camera_cast(cam, st[2 * i], st[2 * i + 1], rng);
    st3(o + 3 * i, r.o);
    st3(d + 3 * i, r.d);
}

// ---- validation entry points for materials -------------------------------------------------------

// MicrofacetBRDF.D / G1 / G / F / Evaluate (microfacet.go:28-103) for n argument sets.
__global__ void __launch_bounds__(kBlock) k_microfacet_batch(int which, uint64_t n, const double* __restrict__ roughness, const double* __restrict__ f0,
                                                            const double* __restrict__ args, double* __restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const V3 a = ld3(args + 12 * i), b = ld3(args + 12 * i + 3), c = ld3(args + 12 * i + 6), d = ld3(args + 12 * i + 9);
    const double rough = roughness[i];
    V3 r = mk(0, 0, 0);
    switch (which) {
        case 0: r.x = mf_D(rough, a, b); break;
        case 1: r.x = mf_G1(rough, a, b, c); break;
        case 2: r.x = mf_G(rough, a, b, c, d); break;
        case 3: r = mf_F(ld3(f0 + 3 * i), a, b); break;
        default: r = mf_evaluate(rough, ld3(f0 + 3 * i), a, b, c); break;
    }
    st3(out + 3 * i, r);
}

// Material.Resolve and ComputeDirectLighting for explicit hits (material_*.go), one thread per hit: what shade_lane and
// k_direct compute for a queued hit, returned instead of queued.
template <int MODE>
__global__ void __launch_bounds__(kTraceBlock, 2) k_shade_batch(DScene sc, const rt_shade_in* __restrict__ in, uint64_t n, uint32_t seed_lo, uint32_t seed_hi,
                                                               rt_shade_out* __restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * kTraceBlock + threadIdx.x;
    if (i >= n) return;
    const rt_shade_in h = in[i];
    const rt_material m = sc.materials[h.material];
    Col c;
    c.t = h.t;
    c.at = ld3(h.at);
    c.normal = ld3(h.normal);
    c.u = h.uv[0];
    c.v = h.uv[1];
    Rng rng{seed_lo, seed_hi, h.pixel, h.sample, h.path, h.draws};
    Scatter sct;
    resolve(sc, m, ld3(h.d), ld3(h.radiance), c, rng, sct);
    rt_shade_out o;
    st3(o.emission, sct.emission);
    o.n_children = sct.n;
    o.draws_after = rng.n;
    for (int k = 0; k < 2; k++) {
        st3(o.dir[k], k < sct.n ? sct.dir[k] : mk(0, 0, 0));
        st3(o.rad[k], k < sct.n ? sct.rad[k] : mk(0, 0, 0));
    }
    V3 direct = mk(0, 0, 0);
    uint32_t shadow = 0;
    VisitCounts vc{0, 0};
    if (m.kind == RT_MAT_LAMBERTIAN || m.kind == RT_MAT_METAL || m.kind == RT_MAT_DIELECTRIC) {
        const LitSurface lit = lit_surface(m, c.normal, ld3(h.d), sct.albedo);
        for (int l = 0; l < sc.n_lights; l++) {
            const rt_light L = sc.lights[l];
            V3 dir;
            double dist;
            const Ray sh = shadow_ray(m, L, c.at, lit.n, dir, dist);
            shadow++;
            if (scene_occluded<MODE>(sc, sh, dist, h.node, LdsStack{dyn_lds + threadIdx.x, kTraceBlock, sc.stack_pairs}, vc)) continue;
            direct = direct + light_contribution(m, L, lit.n, lit.wo, lit.albedo, lit.inside, dir, sh.d);
        }
    }
    st3(o.direct, direct);
    o.shadow_rays = shadow;
    out[i] = o;
}

// ---- launchers ---------------------------------------------------------------------------------

static inline unsigned blocks_for(uint64_t n) { return (unsigned)((n + kBlock - 1) / kBlock); }
static inline unsigned trace_blocks(uint64_t n) { return (unsigned)((n + kTraceBlock - 1) / kTraceBlock); }
// dynamic LDS of a traversal launch: the per-lane stack of (reference, distance) pairs, fast mode only
static inline size_t stack_bytes(const DScene& sc, int mode) {
    return mode == RT_MODE_FAST ? (size_t)2 * sc.stack_pairs * kTraceBlock * sizeof(int32_t) : 0;
}

// Resident single-wave workgroups of a k_primary variant on the current device, rounded down to whole
// rounds over the 8 XCDs.
template <typename K>
static unsigned persistent_grid(K kernel, size_t lds, int block = kTraceBlock) {
    // the answer depends only on (device, kernel, LDS bytes)
    static std::mutex mutex;
    static std::map<std::tuple<const void*, int, size_t, int>, unsigned> cache;
    int device = 0;
    (void)hipGetDevice(&device);
    const auto key = std::make_tuple(reinterpret_cast<const void*>(kernel), device, lds, block);
    {
        std::lock_guard<std::mutex> lock(mutex);
        const auto it = cache.find(key);
        if (it != cache.end()) return it->second;
    }
    int cus = 0, per_cu = 0;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device);
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, block, lds) != hipSuccess || per_cu < 1) per_cu = block == kTraceBlock ? 4 : 1;
    if (getenv("RT_DEBUG")) fprintf(stderr, "[rt] persistent kernel %p: %d CUs x %d workgroups, %zu B LDS each\n", reinterpret_cast<const void*>(kernel), cus, per_cu, lds);
    const unsigned n = ((unsigned)std::max(cus, 8) * (unsigned)per_cu) & ~7u;
    std::lock_guard<std::mutex> lock(mutex);
    cache[key] = n;
    return n;
}

int beam_walk_waves() {  // the most persistent waves a launch of k_beam can have on the current device: its launch bounds times the SIMDs
    int device = 0, cus = 0;
    (void)hipGetDevice(&device);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device);
    return std::max(cus, 8) * RT_BEAM_WAVES * 4;
}
int beam_pixels_max() { return kBeamPixels; }

void launch_beam_prepare(const DScene& sc, int64_t n_faces, bool pyr, const double key[9], BeamFace32* f32, BeamFace64* f64, double* uv, hipStream_t s) {
    if (n_faces <= 0) return;
    const dim3 grid(blocks_for((size_t)n_faces)), block(kBlock);
    const V3 axis_u = mk(key[3], key[4], key[5]), axis_v = mk(key[6], key[7], key[8]);
    if (pyr) hipLaunchKernelGGL(k_beam_prepare<true>, grid, block, 0, s, sc, key[0], key[1], key[2], axis_u, axis_v, f32, f64, uv);
    else hipLaunchKernelGGL(k_beam_prepare<false>, grid, block, 0, s, sc, key[0], key[1], key[2], axis_u, axis_v, f32, f64, uv);
}

int beam_walk_stride() { return kBeamWalkPixels * (kBeamKeptDwords + kBeamCandsMax * 3); }
static_assert(kBeamPixels * kBeamKeptDwords <= 64, "beam_item fetches the kept states with a lane per dword");

size_t beam_table_bytes(int64_t n_faces, size_t* off_f64, size_t* off_uv) {
    const size_t n = (size_t)n_faces, a = (n * sizeof(BeamFace32) + 255) & ~(size_t)255, b = (n * sizeof(BeamFace64) + 255) & ~(size_t)255;
    if (off_f64) *off_f64 = a;
    if (off_uv) *off_uv = a + b;
    return a + b + n * 6 * sizeof(double);
}

// The pixel-beam launch: when the pass allows it (64 lanes per pixel, items of at most kBeamPixels pixels).  False: the caller
// launches k_primary instead.
bool launch_beam(const DScene& sc, const DCamera* cam, const PassDesc& pass, Counters* counters, const BeamTables& bt, double* film, hipStream_t s) {
    if (pass.n_pixels <= 0) return true;
    if (pass.log2_s != 6 || pass.block_px > kBeamPixels || pass.spp <= 1) return false;
    // the stacks of the kBeamPixels walking lanes, a (reference, distance) pair per level each; behind them the lists, the
    // walks' states and beam_cull_item32's prefix sums
    size_t lds = (size_t)kBeamPixels * 2 * sc.stack_pairs * sizeof(int32_t);
    lds = (lds + 15) & ~(size_t)15;
    PassDesc pd = pass;
    pd.acc_offset = 0;
    pd.beam_offset = (int32_t)(lds / sizeof(int32_t));
    // Leaves per list: what is left of a workgroup's share of the CU's LDS.  The hardware hands LDS out in pieces of 1280 bytes
    // (measured: with 7696 bytes per workgroup 18 of them are resident on a CU, with 7600 bytes the 20 that the registers allow;
    // 18 instead of 20 cost 5 % on the 4K scan), so the share is the largest multiple of that which lets all workgroups in.
    const size_t per_cu = 160 * 1024, piece = 1280, groups = (size_t)RT_BEAM_WAVES * 4;
    const size_t share = per_cu / groups / piece * piece;
    const size_t fixed = ((size_t)kBeamPixels * (kBeamStateDwords + kBeamFacesMax * kBeamFaceDwords) + kBeamCullDwords) * sizeof(int32_t);
    const size_t per_leaf = (size_t)kBeamPixels * 3 * sizeof(int32_t);
    int cands = 16;  // (a tree so deep that even this does not fit costs occupancy, nothing else)
    if (share > lds + fixed + 16 * per_leaf) cands = (int)std::min<size_t>(kBeamCandsMax, (share - lds - fixed) / per_leaf);
    pd.beam_cands = cands;
    lds += fixed + (size_t)cands * per_leaf;
    // Pixels walked together: as many lanes as the LDS has room for their stacks (before the lists are needed),
    // for a frame or a large share of one; an item's own for a small share, whose launch wants many small pieces of work.
    // Measured with 4 / 8 / 16 pixels per walk, ms per launch: the 4K scan frame 67.0 / 62.9 / 60.9, a quarter of it - / 16.2 / 15.6, an
    // eighth - / 8.40 / 8.14; the same scene at 1920x1080 20.3 / 19.1 / 18.6, a quarter of that 5.96 / 6.18 / 6.49, an eighth 4.74 / 5.05 / 5.24.
    // Later, 16 / 32 pixels: the frame 55.9 / 55.0, a half 28.3 / 27.9, a quarter 14.4 / 14.3, an eighth 7.5 / 8.1.
    int walk_px = pass.block_px;
    int want_walk = pass.n_pixels >= 1500000 ? 32 : (pass.n_pixels >= 700000 ? 16 : pass.block_px);
    if (const char* e = getenv("RT_BEAM_WALK_PX")) want_walk = atoi(e);  // tuning knob
    const size_t walk_lane = 2 * (size_t)sc.stack_pairs * sizeof(int32_t);  // a walking lane's stack
    while (walk_px * 2 <= want_walk && walk_px * 2 <= kBeamWalkPixels && (size_t)(walk_px * 2) * walk_lane <= lds) walk_px *= 2;
    pd.walk_px = walk_px;
    // The end of a scheduled queue in small batches: an eighth of it, at most what keeps the XCD's waves busy for a few large batches' time.
    if (!bt.walk_scratch || bt.walk_stride < beam_walk_stride() || bt.walk_waves < 8 || !bt.long_px) return false;
    static std::atomic<uint32_t> serial{0};
    do pd.beam_serial = ++serial; while (pd.beam_serial == 0);
    pd.long_cap = kBeamLongCap;
    if (const char* e = getenv("RT_BEAM_LONG_CAP")) pd.long_cap = (uint32_t)std::min<long long>(kBeamLongCap, std::max(0ll, atoll(e)));  // test knob: a queue that runs full
    if (getenv("RT_NO_LONG_QUEUE")) pd.beam_serial = 0;  // test knob: every wave traces the pixels with long lists that it meets
    const unsigned tiles = trace_blocks((uint64_t)pass.n_pixels);
    auto go = [&](auto kernel) {
        const unsigned resident = persistent_grid(kernel, lds);
        const unsigned waves = std::min(std::min(resident, (tiles + 7u) & ~7u), (unsigned)bt.walk_waves & ~7u);  // (a stretch of walk_scratch each)
        hipLaunchKernelGGL(kernel, dim3(waves), dim3(kTraceBlock), lds, s, sc, cam, pd, counters, bt, film);
    };
    if (bt.pyramid) { if (bt.image) go(k_beam<true, true, RT_BEAM_WAVES>); else go(k_beam<false, true, RT_BEAM_WAVES>); }
    else { if (bt.image) go(k_beam<true, false, RT_BEAM_WAVES>); else go(k_beam<false, false, RT_BEAM_WAVES>); }
    return true;
}

void launch_primary(const DScene& sc, const DCamera* cam, const PassDesc& pass, const Queues* q, int mode, bool mesh_only, bool shade,
                    double* film, hipStream_t s) {
    if (pass.n_pixels <= 0) return;
    size_t lds = stack_bytes(sc, mode);
    // wave-packet traversal: one distance per lane and level, one reference per level
    if (mode == RT_MODE_FAST) lds = (size_t)sc.stack_pairs * (kTraceBlock + kTraceBlock / 64) * sizeof(int32_t);
    // behind the stack: the running sums of the pixels being traced, three float64 per lane
    lds = (lds + 7) & ~(size_t)7;
    PassDesc pd = pass;
    pd.acc_offset = (int32_t)(lds / sizeof(int32_t));
    lds += (size_t)3 * kTraceBlock * sizeof(double);
    pd.beam_offset = 0;
    const unsigned tiles = trace_blocks((uint64_t)pass.n_pixels);
    auto go = [&](auto kernel) {
        const unsigned resident = persistent_grid(kernel, lds);
        unsigned waves = std::min(resident, (tiles + 7u) & ~7u);
        if (pd.max_waves > 0) waves = std::min(waves, (unsigned)pd.max_waves);
        const dim3 grid(waves), block(kTraceBlock);
        hipLaunchKernelGGL(kernel, grid, block, lds, s, sc, cam, pd, q, film);
    };
    auto pick = [&](auto mode_tag) {
        constexpr int M = decltype(mode_tag)::value;
        if (mesh_only) { if (shade) go(k_primary<M, true, true>); else go(k_primary<M, true, false>); }
        else if (M == RT_MODE_FAST && sc.n_meshes == 0) {  // no tree to walk: the walk's code and registers are left out
            if (shade) go(k_primary<RT_MODE_FAST, false, true, true>); else go(k_primary<RT_MODE_FAST, false, false, true>);
        }
        else           { if (shade) go(k_primary<M, false, true>); else go(k_primary<M, false, false>); }
    };
    if (mode == RT_MODE_REFERENCE) pick(std::integral_constant<int, RT_MODE_REFERENCE>{});
    else pick(std::integral_constant<int, RT_MODE_FAST>{});
}

static uint32_t most(const StripeCounts& c) {
    uint32_t m = 0;
    for (uint32_t v : c.n) m = std::max(m, v);
    return m;
}

void launch_trace(const DScene& sc, const PassDesc& pass, const Queues& q, int in, const RayStripeCounts& n, int mode, bool mesh_only, double* film,
                  hipStream_t s) {
    uint64_t total = 0;
    uint32_t most_n = 0;
    for (uint32_t v : n.n) { total += v; most_n = std::max(most_n, v); }
    if (total == 0) return;
    const int n_stripes = sc.walk_classes ? kRayStripes : kQueueStripes;
    if (sc.n_meshes == 0) {  // no tree to walk: one fused launch, one ray per lane, blockIdx.y = stripe
        const dim3 grid(trace_blocks(most_n), n_stripes), block(kTraceBlock);
        if (mode == RT_MODE_REFERENCE) hipLaunchKernelGGL((k_trace<RT_MODE_REFERENCE, 2>), grid, block, 0, s, sc, pass, q, in, n, film);
        else hipLaunchKernelGGL((k_trace<RT_MODE_FAST, RT_TRACE_SHAPES_WAVES>), grid, block, 0, s, sc, pass, q, in, n, film);
        return;
    }
    const size_t lds = mode == RT_MODE_FAST ? (size_t)2 * kWalkRing * kTraceBlock * sizeof(int32_t) : 0;  // the top of every lane's walk stack
    const unsigned want_waves = (unsigned)std::min<uint64_t>((total + 63) / 64, (uint64_t)kWalkWavesMax);
    auto walk = [&](auto kernel) {
        const unsigned waves = std::max(1u, std::min(persistent_grid(kernel, lds), want_waves));
        hipLaunchKernelGGL(kernel, dim3(waves), dim3(kTraceBlock), lds, s, sc, q, in, n, n_stripes);
    };
    auto shade = [&](auto kernel) {  // workgroups of four waves
        const unsigned groups = std::max(1u, std::min(persistent_grid(kernel, 0, kSortBlock), (want_waves + 3u) / 4u + (unsigned)kQueueStripes));
        hipLaunchKernelGGL(kernel, dim3(groups), dim3(kSortBlock), 0, s, sc, pass, q, in, film);
    };
    if (mode == RT_MODE_REFERENCE) {
        if (mesh_only) { walk(k_walk<RT_MODE_REFERENCE, true, 2>); shade(k_shade<RT_MODE_REFERENCE, true, 2>); }
        else { walk(k_walk<RT_MODE_REFERENCE, false, 2>); shade(k_shade<RT_MODE_REFERENCE, false, 2>); }
    } else {
        if (mesh_only) { walk(k_walk<RT_MODE_FAST, true, RT_TRACE_WAVES>); shade(k_shade<RT_MODE_FAST, true, RT_SHADE_WAVES>); }
        else { walk(k_walk<RT_MODE_FAST, false, RT_TRACE_WAVES>); shade(k_shade<RT_MODE_FAST, false, RT_SHADE_WAVES>); }
    }
}

void launch_direct(const DScene& sc, const Queues& q, const StripeCounts& n_surf, int mode, double* film, hipStream_t s) {
    if (most(n_surf) == 0 || sc.n_lights == 0) return;
    // lights per thread: one for the usual handful; with many lights, runs of sixteen (257 lights: 17 runs)
    int per_thread = sc.n_lights > 8 ? 16 : 1;
    if (const char* e = getenv("RT_LIGHTS_PER_THREAD")) per_thread = std::max(1, atoi(e));
    const unsigned runs = ((unsigned)sc.n_lights + (unsigned)per_thread - 1) / (unsigned)per_thread;
    const unsigned chunks = (trace_blocks(most(n_surf)) + 7u) / 8u * 8u, group = runs > 2 ? 8u : chunks;
    dim3 grid(chunks * runs, kQueueStripes);
    if (mode == RT_MODE_REFERENCE) {
        if (per_thread > 1) hipLaunchKernelGGL((k_direct<RT_MODE_REFERENCE, true>), grid, dim3(kTraceBlock), 0, s, sc, q, n_surf, per_thread, runs, group, film);
        else hipLaunchKernelGGL((k_direct<RT_MODE_REFERENCE, false>), grid, dim3(kTraceBlock), 0, s, sc, q, n_surf, per_thread, runs, group, film);
    } else {
        if (sc.n_meshes == 0) {  // no tree to walk
            if (per_thread > 1) hipLaunchKernelGGL((k_direct<RT_MODE_FAST, true, true>), grid, dim3(kTraceBlock), 0, s, sc, q, n_surf, per_thread, runs, group, film);
            else hipLaunchKernelGGL((k_direct<RT_MODE_FAST, false, true>), grid, dim3(kTraceBlock), 0, s, sc, q, n_surf, per_thread, runs, group, film);
        }
        else if (per_thread > 1) hipLaunchKernelGGL((k_direct<RT_MODE_FAST, true>), grid, dim3(kTraceBlock), stack_bytes(sc, mode), s, sc, q, n_surf, per_thread, runs, group, film);
        else hipLaunchKernelGGL((k_direct<RT_MODE_FAST, false>), grid, dim3(kTraceBlock), stack_bytes(sc, mode), s, sc, q, n_surf, per_thread, runs, group, film);
    }
}

#ifdef RT_WAVE_TIMES
void debug_wave_times(unsigned long long* out, bool reset) {  // out: 3 * 8192 values
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wave_times), 4 * 8192 * sizeof(unsigned long long));
    if (reset) {
        static const unsigned long long zero[4 * 8192] = {};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_wave_times), zero, sizeof(zero));
    }
}
#endif
#ifdef RT_VISIT_STATS
void debug_packet_stats(unsigned long long* out8, bool reset) {  // out8: 16 values
    (void)hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_packet_stats), 16 * sizeof(unsigned long long));
    if (reset) {
        unsigned long long zero[16] = {};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_packet_stats), zero, sizeof(zero));
    }
}
void debug_trace_stats(unsigned long long* out32, bool reset) {
    static unsigned long long all[64][64];
    (void)hipMemcpyFromSymbol(all, HIP_SYMBOL(g_trace_stats), sizeof(all));
    for (int i = 0; i < 64; i++) {
        out32[i] = 0;
        for (int s = 0; s < 64; s++) out32[i] += all[s][i];
    }
    if (reset) {
        std::memset(all, 0, sizeof(all));
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_trace_stats), all, sizeof(all));
    }
}
#endif

// ---- reconstruction filter (reconstruct.go:28-167) ---------------------------------------------

RT_DEV double recon_radius(int filter) { return filter == RT_RECON_BOX ? 0.5 : (filter == RT_RECON_TENT ? 1.0 : 2.0); }

RT_DEV double recon_eval(int filter, double x) {
    if (filter == RT_RECON_BOX) return fabs(x) < 0.5 ? 1.0 : 0.0;  // reconstruct.go:35-40
    if (filter == RT_RECON_TENT) {                                  // reconstruct.go:51-57
        const double ax = fabs(x);
        return ax >= 1 ? 0.0 : 1 - ax;
    }
    const double B = 1.0 / 3.0, C = 1.0 / 3.0;                      // reconstruct.go:67-85
    x = fabs(x);
    if (x >= 2) return 0;
    const double x2 = x * x, x3 = x2 * x;
    if (x < 1) return ((12 - 9 * B - 6 * C) * x3 + (-18 + 12 * B + 6 * C) * x2 + (6 - 2 * B)) / 6.0;
    return ((-B - 6 * C) * x3 + (6 * B + 30 * C) * x2 + (-12 * B - 48 * C) * x + (8 * B + 24 * C)) / 6.0;
}

// Horizontal pass (reconstruct.go:103-135): one thread per pixel, taps summed left to right as the reference does.
__global__ void __launch_bounds__(kBlock) k_filter_h(const uchar4* __restrict__ src, int32_t w, int32_t h, int filter, double* __restrict__ tmp) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= w * h) return;
    const int x = i % w, y = i / w;
    const double radius = recon_radius(filter);
    int x0 = (int)floor((double)x - radius), x1 = (int)ceil((double)x + radius);
    if (x0 < 0) x0 = 0;
    if (x1 > w - 1) x1 = w - 1;
    double wr = 0, wg = 0, wb = 0, wsum = 0;
    for (int xi = x0; xi <= x1; xi++) {
        const double wk = recon_eval(filter, (double)x - (double)xi);
        const uchar4 p = src[(size_t)y * w + xi];
        const double r = (double)((uint32_t)p.x * 0x101u) / 65535.0;  // color.RGBA.RGBA(): 8 -> 16 bit
        const double g = (double)((uint32_t)p.y * 0x101u) / 65535.0;
        const double b = (double)((uint32_t)p.z * 0x101u) / 65535.0;
        wr += wk * r; wg += wk * g; wb += wk * b;
        wsum += wk;
    }
    if (wsum > 0) { wr /= wsum; wg /= wsum; wb /= wsum; }
    tmp[(size_t)i * 3 + 0] = wr; tmp[(size_t)i * 3 + 1] = wg; tmp[(size_t)i * 3 + 2] = wb;
}

// Vertical pass and quantisation (reconstruct.go:138-165).
__global__ void __launch_bounds__(kBlock) k_filter_v(const double* __restrict__ tmp, int32_t w, int32_t h, int filter, uchar4* __restrict__ dst) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= w * h) return;
    const int x = i % w, y = i / w;
    const double radius = recon_radius(filter);
    int y0 = (int)floor((double)y - radius), y1 = (int)ceil((double)y + radius);
    if (y0 < 0) y0 = 0;
    if (y1 > h - 1) y1 = h - 1;
    double wr = 0, wg = 0, wb = 0, wsum = 0;
    for (int yi = y0; yi <= y1; yi++) {
        const double wk = recon_eval(filter, (double)y - (double)yi);
        const double* p = tmp + ((size_t)yi * w + x) * 3;
        wr += wk * p[0]; wg += wk * p[1]; wb += wk * p[2];
        wsum += wk;
    }
    if (wsum > 0) { wr /= wsum; wg /= wsum; wb /= wsum; }
    uchar4 o;
    o.x = (unsigned char)go_max(0, go_min(255, 255.0 * wr));
    o.y = (unsigned char)go_max(0, go_min(255, 255.0 * wg));
    o.z = (unsigned char)go_max(0, go_min(255, 255.0 * wb));
    o.w = 255;
    dst[i] = o;
}

void launch_filter(const uint8_t* src, int32_t w, int32_t h, int filter, double* tmp, uint8_t* dst, hipStream_t s) {
    const unsigned blocks = blocks_for((uint64_t)w * (uint64_t)h);
    hipLaunchKernelGGL(k_filter_h, dim3(blocks), dim3(kBlock), 0, s, reinterpret_cast<const uchar4*>(src), w, h, filter, tmp);
    hipLaunchKernelGGL(k_filter_v, dim3(blocks), dim3(kBlock), 0, s, tmp, w, h, filter, reinterpret_cast<uchar4*>(dst));
}

// dst += src over n float64 (the films of two devices that rendered different samples of the same pixels)
__global__ void __launch_bounds__(kBlock) k_film_add(double* __restrict__ dst, const double* __restrict__ src, size_t n) {
    const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i < n) dst[i] += src[i];
}

void launch_film_add(double* dst, const double* src, size_t n, hipStream_t s) {
    if (n == 0) return;
    hipLaunchKernelGGL(k_film_add, dim3(blocks_for(n)), dim3(kBlock), 0, s, dst, src, n);
}

// dst += src on the rows of the 16-row bands first, first + step, first + 2 step, ... of a film of dy rows of row_doubles
// numbers: one device's share of a frame divided by rows, added into the frame on the first device.
__global__ void __launch_bounds__(kBlock) k_film_add_bands(double* __restrict__ dst, const double* __restrict__ src, size_t row_doubles, int dy, int first,
                                                           int step, size_t n) {
    const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const size_t local_row = i / row_doubles, col = i - local_row * row_doubles;
    const size_t row = ((local_row >> 4) * (size_t)step + (size_t)first) * 16 + (local_row & 15);
    if (row >= (size_t)dy) return;  // the last band of the frame may be a partial one
    dst[row * row_doubles + col] += src[row * row_doubles + col];
}

void launch_film_add_bands(double* dst, const double* src, int dx, int dy, int first, int step, hipStream_t s) {
    const int bands = (dy + 15) / 16;
    if (first >= bands || step <= 0) return;
    const size_t mine = (size_t)((bands - first + step - 1) / step) * 16, row_doubles = (size_t)dx * 3, n = mine * row_doubles;
    hipLaunchKernelGGL(k_film_add_bands, dim3(blocks_for(n)), dim3(kBlock), 0, s, dst, src, row_doubles, dy, first, step, n);
}

// ---- static scratch ------------------------------------------------------------------------------
// Device memory the library owns without ever allocating it: every GPU that loads this code object holds its own
// copy of the array, at its own address, until the process ends.  rt_render_film_noalloc keeps its counters, camera,
// queue descriptors and tile schedule here.  In host code the array's name is only a handle for
// hipGetSymbolAddress; what kernels, memsets and copies receive is the address that call returns on the device
// in question.
static __device__ __align__(256) unsigned char scratch[kStaticScratchBytes];

namespace {
std::mutex g_scratch_mutex;
void* g_scratch_ptr[kStaticScratchDevices] = {};  // per device ordinal; null = not looked up yet
}  // namespace

hipError_t static_scratch(int device, void** ptr) {
    *ptr = nullptr;
    if (device < 0 || device >= kStaticScratchDevices) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[device]) {
        int before = 0;
        hipError_t e = hipGetDevice(&before);
        if (e != hipSuccess) return e;
        e = hipSetDevice(device);  // the symbol resolves to the copy of the CURRENT device
        if (e != hipSuccess) return e;
        void* p = nullptr;
        e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        (void)hipSetDevice(before);
        if (e != hipSuccess) return e;
        g_scratch_ptr[device] = p;
    }
    *ptr = g_scratch_ptr[device];
    return hipSuccess;
}

void launch_resolve(const double* film, int32_t dx, int32_t dy, int32_t spp, uint8_t* rgba, hipStream_t s) {
    const int32_t n = dx * dy;
    hipLaunchKernelGGL(k_resolve, dim3(blocks_for((uint64_t)n)), dim3(kBlock), 0, s, film, n, spp, reinterpret_cast<uchar4*>(rgba));
}

void launch_intersect_batch(const DScene& sc, const double* o, const double* d, uint64_t n, double tmin, double tmax, int mode,
                            rt_hit* hits, hipStream_t s) {
    if (n == 0) return;
    if (mode == RT_MODE_REFERENCE)
        hipLaunchKernelGGL(k_intersect_batch<RT_MODE_REFERENCE>, dim3(trace_blocks(n)), dim3(kTraceBlock), 0, s, sc, o, d, n, tmin, tmax, hits);
    else
        hipLaunchKernelGGL(k_intersect_batch<RT_MODE_FAST>, dim3(trace_blocks(n)), dim3(kTraceBlock), stack_bytes(sc, mode), s, sc, o, d, n, tmin,
                           tmax, hits);
}

void launch_microfacet_batch(int which, uint64_t n, const double* roughness, const double* f0, const double* args, double* out, hipStream_t s) {
    if (n == 0) return;
    hipLaunchKernelGGL(k_microfacet_batch, dim3(blocks_for(n)), dim3(kBlock), 0, s, which, n, roughness, f0, args, out);
}

void launch_shade_batch(const DScene& sc, const rt_shade_in* in, uint64_t n, uint32_t seed_lo, uint32_t seed_hi, int mode, rt_shade_out* out, hipStream_t s) {
    if (n == 0) return;
    if (mode == RT_MODE_REFERENCE)
        hipLaunchKernelGGL(k_shade_batch<RT_MODE_REFERENCE>, dim3(trace_blocks(n)), dim3(kTraceBlock), 0, s, sc, in, n, seed_lo, seed_hi, out);
    else
        hipLaunchKernelGGL(k_shade_batch<RT_MODE_FAST>, dim3(trace_blocks(n)), dim3(kTraceBlock), stack_bytes(sc, mode), s, sc, in, n, seed_lo, seed_hi, out);
}

void launch_cast_batch(const DCamera& cam, const double* st, const uint32_t* pixel_sample, uint64_t n, uint32_t seed_lo,
                       uint32_t seed_hi, double* o, double* d, hipStream_t s) {
    if (n == 0) return;
    hipLaunchKernelGGL(k_cast_batch, dim3(blocks_for(n)), dim3(kBlock), 0, s, cam, st, pixel_sample, n, seed_lo, seed_hi, o, d);
}

}  // namespace rt
