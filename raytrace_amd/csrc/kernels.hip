// HIP kernels of the wavefront path tracer (gfx950).
//
// One launch per stage and generation:
//   k_generate  camera rays for a pass (8x8 pixel tiles x a run of sample indices)
//   k_extend    closest hit of every queued ray: analytic shapes in float64, meshes by BVH walk
//   k_shade     Material.Resolve: emission to the film, children appended to the next queue
//               (wave64 ballot compaction), lit surface points appended for the light loop
//   k_direct    one thread per (surface point, light): occlusion ray + BRDF, added to the film
//   k_resolve   film sums -> 8-bit RGBA exactly as renderPixel quantises (render.go:218-224)
#include "kernels.h"

#include "dev_bvh.h"
#include "dev_shade.h"

namespace rt {

constexpr int kBlock = 256;
constexpr uint32_t kInvalidPixel = 0xFFFFFFFFu;

// ---- scene level intersection ---------------------------------------------------------------

struct SceneHit {
    int node, prim;
    Col c;
};

// Shape.Collide for one top-level shape.  Transformed shapes (shape_transformed_shape.go:26-57)
// are unwound iteratively: the ray goes through each inverse transform (direction not
// renormalised, so t is preserved), the innermost shape is intersected, and the hit point and
// normal come back through the forward transforms.
template <int MODE>
RT_DEV bool shape_collide(const DScene& sc, int kind, int index, const Ray& r_in, double tmin, double tmax, Col& c, int& prim,
                          VisitCounts& cnt, int32_t* lds, int stride) {
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
    switch (kind) {
        case RT_SHAPE_SPHERE:
            hit = sphere_collide(sc.spheres[index], r, tmin, tmax, c);
            if (hit) sphere_uv(sc.spheres[index], c);
            break;
        case RT_SHAPE_QUAD: hit = quad_collide(sc.quads[index], r, tmin, tmax, c); break;
        case RT_SHAPE_CYLINDER: hit = cylinder_collide(sc.cylinders[index], r, tmin, tmax, c); break;
        case RT_SHAPE_TRIANGLE: hit = tri_collide(sc.tris[index], r, tmin, tmax, c); break;
        case RT_SHAPE_MESH: {
            const DMesh& m = sc.meshes[index];
            double t, u, v;
            int p = 0;
            if (MODE == RT_MODE_REFERENCE) hit = mesh_collide_ref(sc, m, r, tmin, tmax, p, t, u, v, cnt);
            else hit = mesh_collide_fast<false>(sc, m, r, tmin, tmax, p, t, u, v, lds, stride);
            if (hit) {
                const double* pos = sc.prim_pos + (m.prim_off + p) * 9;
                prim = sc.prim_face[m.prim_off + p];
                face_finish(r, ld3(pos), ld3(pos + 3), ld3(pos + 6), sc.face_uv + (m.face_off + prim) * 6, t, u, v, c);
            }
            break;
        }
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

// The node loop of tracePath (render.go:155-169).
template <int MODE>
RT_DEV bool scene_closest(const DScene& sc, const Ray& r, double tmin, double tmax, SceneHit& h, VisitCounts& cnt, int32_t* lds,
                          int stride) {
    double min_dist = tmax;
    bool hit = false;
    for (int i = 0; i < sc.n_nodes; i++) {
        const DNode nd = sc.nodes[i];
        Col c;
        int prim;
        if (shape_collide<MODE>(sc, nd.kind, nd.index, r, tmin, min_dist, c, prim, cnt, lds, stride) && c.t < min_dist) {
            min_dist = c.t;
            h.c = c;
            h.node = i;
            h.prim = prim;
            hit = true;
        }
    }
    return hit;
}

// Occlusion loop of ComputeDirectLighting (material_lambertian.go:34-45): any node other than
// the one whose shape was hit.  Reference mode runs the full closest-hit Collide per node as
// the reference does; fast mode stops at the first face hit.
template <int MODE>
RT_DEV bool scene_occluded(const DScene& sc, const Ray& r, double dist, int self_node, int32_t* lds, int stride) {
    const DNode self = sc.nodes[self_node];
    VisitCounts cnt{0, 0};
    for (int i = 0; i < sc.n_nodes; i++) {
        const DNode nd = sc.nodes[i];
        if (nd.kind == self.kind && nd.index == self.index) continue;
        if (MODE == RT_MODE_FAST && nd.kind == RT_SHAPE_MESH) {
            double t, u, v;
            int p;
            if (mesh_collide_fast<true>(sc, sc.meshes[nd.index], r, kEps, dist, p, t, u, v, lds, stride)) return true;
            continue;
        }
        Col c;
        int prim;
        if (shape_collide<MODE>(sc, nd.kind, nd.index, r, kEps, dist, c, prim, cnt, lds, stride)) return true;
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

RT_DEV void wave_count(unsigned long long* counter, bool flag) {
    const unsigned long long mask = __ballot(flag);
    if (mask != 0 && (int)(threadIdx.x & 63) == __ffsll((long long)mask) - 1) atomicAdd(counter, (unsigned long long)__popcll(mask));
}

RT_DEV void film_add(double* film, uint32_t pixel, V3 v) {
    double* f = film + (size_t)pixel * 3;
    if (v.x != 0) atomicAdd(f + 0, v.x);
    if (v.y != 0) atomicAdd(f + 1, v.y);
    if (v.z != 0) atomicAdd(f + 2, v.z);
}

// ---- kernels -----------------------------------------------------------------------------------

// renderPixel's sample loop body up to camera.Cast (render.go:197-214) for one pass.
__global__ void __launch_bounds__(kBlock) k_generate(DCamera cam, PassDesc pass, RayRec* __restrict__ rays) {
    const int64_t g = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (g >= pass.n_paths) return;
    const int p = (int)(g & 63);
    const int64_t wave = g >> 6;
    const int s = pass.sample_begin + (int)(wave % pass.sample_count);
    const int64_t tile = pass.tile_begin + wave / pass.sample_count;
    const int lx = (int)(tile % pass.tiles_x) * 8 + (p & 7);
    const int ly = (int)(tile / pass.tiles_x) * 8 + (p >> 3);
    RayRec rec;
    rec.pixel = kInvalidPixel;
    rec.sample = (uint32_t)s;
    rec.path = 1;
    rec.depth_n = 0;
    const int row = ((ly >> 4) * pass.band_world + pass.band_rank) * 16 + (ly & 15);  // image row, 0 = top
    if (lx < pass.dx && ly < pass.local_rows && row < pass.dy) {
        const int y = pass.dy - 1 - row;  // the reference's y grows upward (render.go:195)
        Rng rng{pass.seed_lo, pass.seed_hi, (uint32_t)(y * pass.dx + lx), (uint32_t)s, 1u, 0u};
        double fs, ft;
        if (pass.spp == 1) {  // render.go:202-210
            fs = ((double)lx + 0.5) / (double)pass.dx;
            ft = ((double)y + 0.5) / (double)pass.dy;
        } else {
            fs = ((double)lx + rng.float64()) / (double)pass.dx;
            ft = ((double)y + rng.float64()) / (double)pass.dy;
        }
        Ray r = camera_cast(cam, fs, ft, rng);
        st3(rec.o, r.o);
        st3(rec.d, r.d);
        rec.rad[0] = rec.rad[1] = rec.rad[2] = 1.0;
        rec.pixel = (uint32_t)(row * pass.dx + lx);
        rec.depth_n = rng.n;
    } else {
        rec.o[0] = rec.o[1] = rec.o[2] = 0;
        rec.d[0] = rec.d[1] = rec.d[2] = 0;
        rec.rad[0] = rec.rad[1] = rec.rad[2] = 0;
    }
    rays[g] = rec;
}

template <int MODE>
__global__ void __launch_bounds__(kBlock) k_extend(DScene sc, const RayRec* __restrict__ rays, HitRec* __restrict__ hits, uint32_t n) {
    __shared__ int32_t lds[MODE == RT_MODE_FAST ? 2 * kLdsPairs * kBlock : 1];
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const RayRec rec = rays[i];
    HitRec h;
    h.node = -1;
    h.prim = 0;
    h.t = 0;
    if (rec.pixel != kInvalidPixel) {
        Ray r{ld3(rec.o), ld3(rec.d)};
        SceneHit sh;
        VisitCounts cnt{0, 0};
        if (!bad3(r.o) && !bad3(r.d) && scene_closest<MODE>(sc, r, kEps, kMaxFloat, sh, cnt, lds + threadIdx.x, kBlock)) {
            h.node = sh.node;
            h.prim = sh.prim;
            h.t = sh.c.t;
            st3(h.at, sh.c.at);
            st3(h.normal, sh.c.normal);
            h.uv[0] = sh.c.u;
            h.uv[1] = sh.c.v;
        }
    }
    hits[i] = h;
}

// tracePath after the intersection (render.go:143-182): counters, Material.Resolve, emission,
// children.  ComputeDirectLighting is deferred to k_direct through a SurfRec.
__global__ void __launch_bounds__(kBlock) k_shade(DScene sc, PassDesc pass, Queues q, uint32_t n, double* __restrict__ film) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    const bool in_range = i < n;
    RayRec rec;
    rec.pixel = kInvalidPixel;
    if (in_range) rec = q.rays[i];
    const bool live = in_range && rec.pixel != kInvalidPixel;
    bool left = false;
    int n_child = 0;
    int n_cut = 0;       // children beyond MaxRayDepth: counted, not traced (render.go:151-154)
    int n_counted = 0;   // children that only bump TotalRays (cut or invalid)
    bool want_surf = false;
    RayRec child[2];
    SurfRec surf;
    Ray r{ld3(rec.o), ld3(rec.d)};
    const bool valid = live && !bad3(r.o) && !bad3(r.d);  // render.go:147-150
    if (valid) {
        const HitRec h = q.hits[i];
        if (h.node < 0) {
            left = true;
        } else {
            const rt_material m = sc.materials[sc.nodes[h.node].material];
            Col c;
            c.t = h.t;
            c.at = ld3(h.at);
            c.normal = ld3(h.normal);
            c.u = h.uv[0];
            c.v = h.uv[1];
            const uint32_t x = rec.pixel % (uint32_t)pass.dx, row = rec.pixel / (uint32_t)pass.dx;
            const uint32_t depth = rec.depth_n >> 24;
            Rng rng{pass.seed_lo, pass.seed_hi, (uint32_t)(pass.dy - 1 - (int)row) * (uint32_t)pass.dx + x, rec.sample, rec.path,
                    rec.depth_n & 0xFFFFFFu};
            Scatter sct;
            resolve(sc, m, r.d, ld3(rec.rad), c, rng, sct);
            film_add(film, rec.pixel, sct.emission);
            for (int k = 0; k < sct.n; k++) {
                const bool bad = bad3(c.at) || bad3(sct.dir[k]);
                if (bad) { n_counted++; continue; }
                if ((int)depth + 1 > pass.max_depth) { n_counted++; n_cut++; continue; }
                RayRec& ch = child[n_child++];
                st3(ch.o, c.at);
                st3(ch.d, sct.dir[k]);
                st3(ch.rad, sct.rad[k]);
                ch.pixel = rec.pixel;
                ch.sample = rec.sample;
                ch.path = rec.path * 2u + (uint32_t)k;
                ch.depth_n = (depth + 1u) << 24;
            }
            if (sc.n_lights > 0 && (m.kind == RT_MAT_LAMBERTIAN || m.kind == RT_MAT_METAL || m.kind == RT_MAT_DIELECTRIC)) {
                want_surf = true;
                V3 nn = unit(c.normal);
                V3 wo = unit(r.d * -1.0);  // s.outgoing.Unit() (material_metal.go:59)
                surf.inside = 0;
                V3 alb = mk(0, 0, 0);
                if (m.kind == RT_MAT_LAMBERTIAN) alb = texture_at(sc, m.texture, c.u, c.v);
                else if (m.kind == RT_MAT_METAL) alb = ld3(m.albedo);
                else if (!(dot(wo, nn) > 0)) { surf.inside = 1; nn = nn * -1.0; }  // material_dielectric.go:112-120
                st3(surf.p, c.at);
                st3(surf.n, nn);
                st3(surf.wo, wo);
                st3(surf.albedo, alb);
                surf.pixel = rec.pixel;
                surf.node = h.node;
                surf.material = sc.nodes[h.node].material;
            }
        }
    }
    // counters (render.go:143, 152, 171), one atomic per wave
    {
        const unsigned long long m_live = __ballot(live), m_left = __ballot(left);
        const unsigned long long c1 = __ballot(n_counted >= 1), c2 = __ballot(n_counted >= 2);
        const unsigned long long x1 = __ballot(n_cut >= 1), x2 = __ballot(n_cut >= 2);
        if ((threadIdx.x & 63) == 0) {
            const unsigned long long tot = __popcll(m_live) + __popcll(c1) + __popcll(c2);
            if (tot) atomicAdd(&q.counters->total, tot);
            if (m_left) atomicAdd(&q.counters->left, (unsigned long long)__popcll(m_left));
            const unsigned long long ex = __popcll(x1) + __popcll(x2);
            if (ex) atomicAdd(&q.counters->exceeded, ex);
        }
    }
    // wave64 ballot compaction of the surviving paths into the next generation
    const uint32_t s0 = wave_append(&q.counters->next_count, n_child >= 1);
    const uint32_t s1 = wave_append(&q.counters->next_count, n_child >= 2);
    if (n_child >= 1 && s0 < q.next_capacity) q.next[s0] = child[0];
    if (n_child >= 2 && s1 < q.next_capacity) q.next[s1] = child[1];
    const uint32_t ss = wave_append(&q.counters->surf_count, want_surf);
    if (want_surf && ss < q.surf_capacity) q.surfs[ss] = surf;
}

// ComputeDirectLighting's loop body for one (surface point, light) pair
// (material_lambertian.go:23-52, material_metal.go:68-106, material_dielectric.go:128-170).
template <int MODE>
__global__ void __launch_bounds__(kBlock) k_direct(DScene sc, Queues q, uint32_t n_surf, double* __restrict__ film) {
    __shared__ int32_t lds[MODE == RT_MODE_FAST ? 2 * kLdsPairs * kBlock : 1];
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    const bool live = i < n_surf;
    wave_count(&q.counters->shadow, live);
    if (!live) return;
    const SurfRec s = q.surfs[i];
    const rt_light L = sc.lights[blockIdx.y];
    const rt_material m = sc.materials[s.material];
    const V3 p = ld3(s.p), n = ld3(s.n);
    V3 dir = ld3(L.position) - p;  // PointLight.Sample, light.go:29-35
    const double dist = length(dir);
    dir = dir / dist;
    const V3 wi = m.kind == RT_MAT_LAMBERTIAN ? dir : unit(dir);
    Ray sh{p + n * kEps, wi};
    if (scene_occluded<MODE>(sc, sh, dist, s.node, lds + threadIdx.x, kBlock)) return;
    V3 contrib;
    if (m.kind == RT_MAT_LAMBERTIAN) {
        const double ndl = go_max(0, dot(n, dir));
        contrib = (ld3(s.albedo) * ld3(L.intensity)) * ndl;
    } else if (m.kind == RT_MAT_METAL) {
        const V3 val = mf_evaluate(m.fuzz, ld3(m.albedo), ld3(s.wo), wi, n);
        const double ct = go_max(0, dot(n, wi));
        contrib = (ld3(L.intensity) * val) * ct;
    } else {
        double eta_i = m.ior_exterior, eta_t = m.ior_interior;
        if (s.inside) { const double tmp = eta_i; eta_i = eta_t; eta_t = tmp; }
        const double cos_i = go_max(0, dot(wi, n));
        const double fresnel = reflectance(cos_i, eta_i, eta_t);
        const V3 val = mf_evaluate(m.roughness, mk(1, 1, 1), ld3(s.wo), wi, n) * fresnel;
        const double ct = go_max(0, dot(n, wi));
        contrib = (ld3(L.intensity) * val) * ct;
    }
    film_add(film, s.pixel, contrib);
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
__global__ void __launch_bounds__(kBlock) k_intersect_batch(DScene sc, const double* __restrict__ o, const double* __restrict__ d, uint64_t n,
                                                           double tmin, double tmax, rt_hit* __restrict__ hits) {
    __shared__ int32_t lds[MODE == RT_MODE_FAST ? 2 * kLdsPairs * kBlock : 1];
    const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
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
    if (scene_closest<MODE>(sc, r, tmin, tmax, sh, cnt, lds + threadIdx.x, kBlock)) {
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

// ---- launchers ---------------------------------------------------------------------------------

static inline unsigned blocks_for(uint64_t n) { return (unsigned)((n + kBlock - 1) / kBlock); }

void launch_generate(const DCamera& cam, const PassDesc& pass, RayRec* rays, hipStream_t s) {
    if (pass.n_paths == 0) return;
    hipLaunchKernelGGL(k_generate, dim3(blocks_for((uint64_t)pass.n_paths)), dim3(kBlock), 0, s, cam, pass, rays);
}

void launch_extend(const DScene& sc, const RayRec* rays, HitRec* hits, uint32_t n, int mode, hipStream_t s) {
    if (n == 0) return;
    if (mode == RT_MODE_REFERENCE)
        hipLaunchKernelGGL(k_extend<RT_MODE_REFERENCE>, dim3(blocks_for(n)), dim3(kBlock), 0, s, sc, rays, hits, n);
    else
        hipLaunchKernelGGL(k_extend<RT_MODE_FAST>, dim3(blocks_for(n)), dim3(kBlock), 0, s, sc, rays, hits, n);
}

void launch_shade(const DScene& sc, const PassDesc& pass, const Queues& q, uint32_t n, double* film, hipStream_t s) {
    if (n == 0) return;
    hipLaunchKernelGGL(k_shade, dim3(blocks_for(n)), dim3(kBlock), 0, s, sc, pass, q, n, film);
}

void launch_direct(const DScene& sc, const Queues& q, uint32_t n_surf, int mode, double* film, hipStream_t s) {
    if (n_surf == 0 || sc.n_lights == 0) return;
    dim3 grid(blocks_for(n_surf), (unsigned)sc.n_lights);
    if (mode == RT_MODE_REFERENCE)
        hipLaunchKernelGGL(k_direct<RT_MODE_REFERENCE>, grid, dim3(kBlock), 0, s, sc, q, n_surf, film);
    else
        hipLaunchKernelGGL(k_direct<RT_MODE_FAST>, grid, dim3(kBlock), 0, s, sc, q, n_surf, film);
}

void launch_resolve(const double* film, int32_t dx, int32_t dy, int32_t spp, uint8_t* rgba, hipStream_t s) {
    const int32_t n = dx * dy;
    hipLaunchKernelGGL(k_resolve, dim3(blocks_for((uint64_t)n)), dim3(kBlock), 0, s, film, n, spp, reinterpret_cast<uchar4*>(rgba));
}

void launch_intersect_batch(const DScene& sc, const double* o, const double* d, uint64_t n, double tmin, double tmax, int mode,
                            rt_hit* hits, hipStream_t s) {
    if (n == 0) return;
    if (mode == RT_MODE_REFERENCE)
        hipLaunchKernelGGL(k_intersect_batch<RT_MODE_REFERENCE>, dim3(blocks_for(n)), dim3(kBlock), 0, s, sc, o, d, n, tmin, tmax, hits);
    else
        hipLaunchKernelGGL(k_intersect_batch<RT_MODE_FAST>, dim3(blocks_for(n)), dim3(kBlock), 0, s, sc, o, d, n, tmin, tmax, hits);
}

void launch_cast_batch(const DCamera& cam, const double* st, const uint32_t* pixel_sample, uint64_t n, uint32_t seed_lo,
                       uint32_t seed_hi, double* o, double* d, hipStream_t s) {
    if (n == 0) return;
    hipLaunchKernelGGL(k_cast_batch, dim3(blocks_for(n)), dim3(kBlock), 0, s, cam, st, pixel_sample, n, seed_lo, seed_hi, o, d);
}

}  // namespace rt
