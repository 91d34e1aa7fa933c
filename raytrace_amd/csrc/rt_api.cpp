// C ABI of librt_hip.so (include/rt_hip.h): scene upload, the wavefront render loop and the
// batch entry points.  Host code only; the kernels live in kernels.hip.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "bvh_build.h"
#include "dev_math.h"
#include "kernels.h"
#include "scene_types.h"

namespace {

using namespace rt;

thread_local std::string tl_error;
std::atomic<bool> g_initialized{false};
int g_device = 0;

int fail(int code, const std::string& msg) {
    tl_error = msg;
    return code;
}

#define HIP_TRY(expr)                                                                                         \
    do {                                                                                                      \
        hipError_t e_ = (expr);                                                                               \
        if (e_ != hipSuccess) return fail(RT_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

struct DeviceBuffers {
    std::vector<void*> ptrs;
    ~DeviceBuffers() {
        for (void* p : ptrs) (void)hipFree(p);
    }
    template <typename T>
    hipError_t upload(const std::vector<T>& host, const T** out) {
        T* d = nullptr;
        size_t bytes = std::max<size_t>(host.size(), 1) * sizeof(T);
        hipError_t e = hipMalloc(reinterpret_cast<void**>(&d), bytes);
        if (e != hipSuccess) return e;
        ptrs.push_back(d);
        if (!host.empty()) e = hipMemcpy(d, host.data(), host.size() * sizeof(T), hipMemcpyHostToDevice);
        *out = d;
        return e;
    }
    template <typename T>
    hipError_t alloc(size_t count, T** out) {
        T* d = nullptr;
        hipError_t e = hipMalloc(reinterpret_cast<void**>(&d), std::max<size_t>(count, 1) * sizeof(T));
        if (e == hipSuccess) ptrs.push_back(d);
        *out = d;
        return e;
    }
};

struct WorkBuffers {  // queues of the wavefront loop, allocated at the first render
    DeviceBuffers mem;
    uint32_t q = 0;  // rays processed per launch at most
    RayRec* rays[2] = {nullptr, nullptr};
    HitRec* hits = nullptr;
    SurfRec* surfs = nullptr;
    Counters* counters = nullptr;
    RayRec* pool = nullptr;  // overflow stack for generations that outgrow q
    uint64_t pool_capacity = 0;
    uint8_t* rgba = nullptr;
    size_t rgba_bytes = 0;
    double* film = nullptr;
    size_t film_doubles = 0;
};

}  // namespace

struct rt_scene {
    DeviceBuffers mem;
    DScene dev{};
    std::vector<rt_bvh_info> bvh_info;
    std::atomic<bool> cancel{false};
    std::unique_ptr<WorkBuffers> work;
    int device = 0;
};

namespace {

// ---- host-side precomputation, same float64 operations as the reference's per-call set-up ----

DQuad make_quad(const rt_quad& q) {  // shape_quad.go:40-64, 90-97
    DQuad d;
    V3 nrm = unit(ld3(q.normal));
    V3 arbitrary = std::fabs(nrm.x) < 0.9 ? mk(1, 0, 0) : mk(0, 1, 0);
    V3 u = unit(cross(nrm, arbitrary));
    V3 v = unit(cross(nrm, u));
    u = u * (q.width / 2);
    v = v * (q.height / 2);
    V3 c = ld3(q.center);
    V3 p0 = (c - u) - v, p1 = (c + u) - v, p2 = (c + u) + v, p3 = (c - u) + v;
    V3 lu = p2 - p1, lv = p0 - p1;
    st3(d.nrm, nrm);
    st3(d.p0, p0); st3(d.p1, p1); st3(d.p2, p2); st3(d.p3, p3);
    st3(d.lu, lu); st3(d.lv, lv);
    d.lulu = dot(lu, lu);
    d.lvlv = dot(lv, lv);
    return d;
}

DCylinder make_cylinder(const rt_cylinder& c) {  // shape_cylinder.go:41, 100
    DCylinder d;
    V3 dir = unit(ld3(c.direction));
    st3(d.o, ld3(c.origin));
    st3(d.d, dir);
    st3(d.top, ld3(c.origin) + dir * c.height);
    d.r = c.radius;
    d.h = c.height;
    return d;
}

DXform make_xform(const double* tr, const double* rot, const double* sc) {
    DXform x;
    std::memcpy(x.tr, tr, sizeof(x.tr));
    std::memcpy(x.rot, rot, sizeof(x.rot));
    std::memcpy(x.sc, sc, sizeof(x.sc));
    return x;
}

DXform invert(const DXform& t) {  // Transform.Inverse, transform.go:86-102
    DXform r;
    V3 inv_scale = mk(1 / t.sc[0], 1 / t.sc[1], 1 / t.sc[2]);
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) r.rot[i * 3 + j] = t.rot[j * 3 + i];
    V3 nt = ld3(t.tr) * -1.0;
    V3 rt_ = mk(r.rot[0] * nt.x + r.rot[1] * nt.y + r.rot[2] * nt.z, r.rot[3] * nt.x + r.rot[4] * nt.y + r.rot[5] * nt.z,
                r.rot[6] * nt.x + r.rot[7] * nt.y + r.rot[8] * nt.z);
    st3(r.tr, rt_ * inv_scale);
    st3(r.sc, inv_scale);
    return r;
}

int make_camera(const rt_camera_desc& c, DCamera& d) {
    std::memset(&d, 0, sizeof(d));
    d.kind = c.kind;
    d.intr = c;
    if (c.kind == RT_CAM_PINHOLE) {
        std::memcpy(d.llc, c.lower_left_corner, sizeof(d.llc));
        std::memcpy(d.origin, c.origin, sizeof(d.origin));
        std::memcpy(d.horizontal, c.horizontal, sizeof(d.horizontal));
        std::memcpy(d.vertical, c.vertical, sizeof(d.vertical));
        return RT_OK;
    }
    if (c.kind != RT_CAM_FOCUSABLE && c.kind != RT_CAM_ORTHOGRAPHIC && c.kind != RT_CAM_CALIBRATED)
        return fail(RT_ERR_INVALID, "unknown camera kind");
    V3 from = ld3(c.look_from), at = ld3(c.look_at);
    V3 w = unit(from - at);             // camera_orthographic.go:26-28, camera_focusable.go:28-30,
    V3 u = unit(cross(ld3(c.vup), w));  // camera_calibrated.go:47-49
    V3 v = cross(w, u);
    st3(d.from, from);
    st3(d.u, u); st3(d.v, v); st3(d.w, w);
    st3(d.dir, unit(at - from));
    d.fov_w = c.fov_width;
    d.fov_h = c.fov_height;
    if (c.kind == RT_CAM_FOCUSABLE) {  // camera_focusable.go:33-43
        V3 horizontal = u * (c.fov_width * c.working_distance);
        V3 vertical = v * (c.fov_height * c.working_distance);
        V3 llc = ((from - horizontal / 2.0) - vertical / 2.0) - w * c.working_distance;
        st3(d.horizontal, horizontal);
        st3(d.vertical, vertical);
        st3(d.llc, llc);
        d.lens_radius = c.aperture / 2;
    }
    if (c.kind == RT_CAM_CALIBRATED && (c.width <= 0 || c.height <= 0 || !(c.fx > 0 && c.fy > 0)))
        return fail(RT_ERR_INVALID, "CameraIntrinsics bad image size or focal lengths");
    return RT_OK;
}

int check_shape_ref(const rt_scene_desc& d, int kind, int index, int depth) {
    const uint64_t counts[6] = {d.n_spheres, d.n_quads, d.n_cylinders, d.n_triangles, d.n_meshes, d.n_transformed};
    if (kind < 0 || kind > 5 || index < 0 || (uint64_t)index >= counts[kind]) return fail(RT_ERR_INVALID, "shape reference out of range");
    if (kind == RT_SHAPE_TRANSFORMED) {
        if (depth >= 4) return fail(RT_ERR_UNSUPPORTED, "TransformedShape nested deeper than 4");
        return check_shape_ref(d, d.transformed[index].child_kind, d.transformed[index].child_index, depth + 1);
    }
    return RT_OK;
}

size_t env_size(const char* name, size_t fallback) {
    const char* v = std::getenv(name);
    if (!v || !*v) return fallback;
    return (size_t)std::strtoull(v, nullptr, 10);
}

int ensure_work(rt_scene* sc, int max_depth, int dx, int dy) {
    if (!sc->work) {
        auto w = std::make_unique<WorkBuffers>();
        w->q = (uint32_t)env_size("RT_QUEUE_RAYS", (size_t)1 << 23);
        if (w->q < 64 * 64) w->q = 64 * 64;
        HIP_TRY(w->mem.alloc((size_t)w->q * 2, &w->rays[0]));
        HIP_TRY(w->mem.alloc((size_t)w->q * 2, &w->rays[1]));
        HIP_TRY(w->mem.alloc((size_t)w->q, &w->hits));
        HIP_TRY(w->mem.alloc((size_t)w->q, &w->surfs));
        HIP_TRY(w->mem.alloc(1, &w->counters));
        sc->work = std::move(w);
    }
    WorkBuffers& w = *sc->work;
    (void)max_depth;
    const size_t px = (size_t)dx * dy;
    if (w.rgba_bytes < px * 4) {
        if (w.rgba) (void)hipFree(w.rgba);
        w.rgba = nullptr;
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&w.rgba), px * 4));
        w.rgba_bytes = px * 4;
    }
    return RT_OK;
}

struct EventPair {
    hipEvent_t a = nullptr, b = nullptr;
};

// The wavefront loop for one process's share of the image; adds radiance sums into film.
int render_into_film(rt_scene* sc, const rt_camera_desc* camera, const rt_render_opts* o, double* film, rt_stats* stats) {
    if (!g_initialized) return fail(RT_ERR_NOT_INITIALIZED, "rt_init has not been called");
    if (!sc || !camera || !o || !film) return fail(RT_ERR_INVALID, "null argument");
    // RenderOptions.Validate, render.go:81-98
    if (o->seed < 0) return fail(RT_ERR_INVALID, "bad Seed must be non-negative but got " + std::to_string(o->seed));
    if (o->rays_per_pixel <= 0) return fail(RT_ERR_INVALID, "bad RaysPerPixel must be positive but got " + std::to_string(o->rays_per_pixel));
    if (o->max_ray_depth <= 0) return fail(RT_ERR_INVALID, "bad MaxRayDepth must be positive but got " + std::to_string(o->max_ray_depth));
    if (o->dx <= 0) return fail(RT_ERR_INVALID, "bad Dx must be positive but got " + std::to_string(o->dx));
    if (o->dy <= 0) return fail(RT_ERR_INVALID, "bad Dy must be positive but got " + std::to_string(o->dy));
    if (o->max_ray_depth > 30) return fail(RT_ERR_UNSUPPORTED, "MaxRayDepth above 30 is not supported (path-tree index is 32 bit)");
    if ((uint64_t)o->dx * (uint64_t)o->dy >= 0xFFFFFFFFull) return fail(RT_ERR_UNSUPPORTED, "image too large");
    if (o->mode != RT_MODE_REFERENCE && o->mode != RT_MODE_FAST) return fail(RT_ERR_INVALID, "unknown mode");
    const int world = o->band_world > 0 ? o->band_world : 1;
    const int rank = o->band_rank;
    if (rank < 0 || rank >= world) return fail(RT_ERR_INVALID, "band_rank out of range");
    if (o->sample_count < 0 || o->sample_begin < 0 || (o->sample_count > 0 && o->sample_begin + o->sample_count > o->rays_per_pixel))
        return fail(RT_ERR_INVALID, "sample range outside [0, RaysPerPixel)");
    HIP_TRY(hipSetDevice(sc->device));
    DCamera cam;
    if (int rc = make_camera(*camera, cam)) return rc;
    if (int rc = ensure_work(sc, o->max_ray_depth, o->dx, o->dy)) return rc;
    WorkBuffers& w = *sc->work;
    hipStream_t stream = reinterpret_cast<hipStream_t>(o->stream);
    sc->cancel.store(false);

    const auto t_start = std::chrono::steady_clock::now();
    int local_rows = 0;
    for (int r = 0; r < o->dy; r++)
        if ((r / 16) % world == rank) local_rows++;
    PassDesc pass{};
    pass.dx = o->dx;
    pass.dy = o->dy;
    pass.band_rank = rank;
    pass.band_world = world;
    pass.local_rows = local_rows;
    pass.tiles_x = (o->dx + 7) / 8;
    pass.spp = o->rays_per_pixel;
    pass.seed_lo = (uint32_t)((uint64_t)o->seed & 0xffffffffu);
    pass.seed_hi = (uint32_t)((uint64_t)o->seed >> 32);
    pass.max_depth = o->max_ray_depth;
    const int64_t tiles_total = (int64_t)pass.tiles_x * ((local_rows + 7) / 8);
    const int64_t tiles_per_launch = std::max<int64_t>(1, w.q / 64);

    HIP_TRY(hipMemsetAsync(w.counters, 0, sizeof(Counters), stream));
    Queues q{};
    q.hits = w.hits;
    q.surfs = w.surfs;
    q.counters = w.counters;
    q.next_capacity = w.q * 2;
    q.surf_capacity = w.q;

    std::vector<EventPair> events;
    size_t events_used = 0;
    double traverse_ms = 0;
    uint64_t traverse_launches = 0, traverse_rays = 0;
    auto drain_events = [&]() -> hipError_t {
        for (size_t i = 0; i < events_used; i++) {
            float ms = 0;
            hipError_t e = hipEventElapsedTime(&ms, events[i].a, events[i].b);
            if (e != hipSuccess) return e;
            traverse_ms += ms;
        }
        events_used = 0;
        return hipSuccess;
    };
    auto cleanup_events = [&]() {
        for (auto& ev : events) { (void)hipEventDestroy(ev.a); (void)hipEventDestroy(ev.b); }
    };
    Counters host_counters{};
    uint64_t pool_top = 0;
    int rc = RT_OK;

    // one generation: extend, shade (+ light loop); returns the number of children in rays[1 - cur]
    auto run_generation = [&](int cur, uint32_t n, uint32_t& n_next) -> int {
        // zero the two append counters, keep the statistics
        HIP_TRY(hipMemsetAsync(&w.counters->next_count, 0, 2 * sizeof(unsigned int), stream));
        if (events_used == events.size()) {
            EventPair ev;
            HIP_TRY(hipEventCreate(&ev.a));
            HIP_TRY(hipEventCreate(&ev.b));
            events.push_back(ev);
        }
        EventPair& ev = events[events_used++];
        HIP_TRY(hipEventRecord(ev.a, stream));
        launch_extend(sc->dev, w.rays[cur], w.hits, n, o->mode, stream);
        HIP_TRY(hipEventRecord(ev.b, stream));
        traverse_launches++;
        traverse_rays += n;
        q.rays = w.rays[cur];
        q.next = w.rays[1 - cur];
        launch_shade(sc->dev, pass, q, n, film, stream);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(&host_counters, w.counters, sizeof(Counters), hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        HIP_TRY(drain_events());
        if (host_counters.surf_count > q.surf_capacity || host_counters.next_count > q.next_capacity)
            return fail(RT_ERR_DEVICE, "internal queue overflow");
        if (host_counters.surf_count > 0) {
            launch_direct(sc->dev, q, host_counters.surf_count, o->mode, film, stream);
            HIP_TRY(hipGetLastError());
        }
        n_next = host_counters.next_count;
        return RT_OK;
    };

    const int s_first = o->sample_count > 0 ? o->sample_begin : 0;
    const int s_end = o->sample_count > 0 ? o->sample_begin + o->sample_count : o->rays_per_pixel;
    for (int s0 = s_first; s0 < s_end && rc == RT_OK;) {
        // a pass covers either all tiles for a run of samples, or a run of tiles for one sample
        int s_count = 1;
        if (tiles_total <= tiles_per_launch) s_count = (int)std::min<int64_t>(s_end - s0, tiles_per_launch / std::max<int64_t>(tiles_total, 1));
        for (int64_t t0 = 0; t0 < tiles_total && rc == RT_OK; t0 += tiles_per_launch) {
            const int64_t nt = std::min(tiles_per_launch, tiles_total - t0);
            pass.sample_begin = s0;
            pass.sample_count = s_count;
            pass.tile_begin = t0;
            pass.n_paths = nt * 64 * s_count;
            int cur = 0;
            launch_generate(cam, pass, w.rays[cur], stream);
            uint32_t n = (uint32_t)pass.n_paths;
            while (rc == RT_OK) {
                if (sc->cancel.load()) { rc = fail(RT_ERR_CANCELLED, "render cancelled"); break; }
                uint32_t n_next = 0;
                if (n > 0) {
                    rc = run_generation(cur, n, n_next);
                    if (rc != RT_OK) break;
                    cur = 1 - cur;
                }
                if (n_next > w.q) {  // park the excess on the overflow stack
                    const uint64_t excess = n_next - w.q;
                    if (!w.pool) {  // a generation can outgrow q only through Dielectric's two children; bounded by depth
                        const uint64_t want = (uint64_t)w.q * (uint64_t)(o->max_ray_depth + 1);
                        hipError_t pe = hipMalloc(reinterpret_cast<void**>(&w.pool), want * sizeof(RayRec));
                        if (pe != hipSuccess) { w.pool = nullptr; rc = fail(RT_ERR_NOMEM, hipGetErrorString(pe)); break; }
                        w.pool_capacity = want;
                    }
                    if (pool_top + excess > w.pool_capacity) { rc = fail(RT_ERR_NOMEM, "ray pool exhausted"); break; }
                    hipError_t e = hipMemcpyAsync(w.pool + pool_top, w.rays[cur] + w.q, excess * sizeof(RayRec), hipMemcpyDeviceToDevice, stream);
                    if (e != hipSuccess) { rc = fail(RT_ERR_DEVICE, hipGetErrorString(e)); break; }
                    pool_top += excess;
                    n_next = w.q;
                }
                n = n_next;
                if (n == 0) {
                    if (pool_top == 0) break;
                    const uint64_t take = std::min<uint64_t>(pool_top, w.q);
                    hipError_t e = hipMemcpyAsync(w.rays[cur], w.pool + (pool_top - take), take * sizeof(RayRec), hipMemcpyDeviceToDevice, stream);
                    if (e != hipSuccess) { rc = fail(RT_ERR_DEVICE, hipGetErrorString(e)); break; }
                    pool_top -= take;
                    n = (uint32_t)take;
                }
            }
        }
        s0 += s_count;
    }
    if (rc == RT_OK) {
        hipError_t e = hipMemcpyAsync(&host_counters, w.counters, sizeof(Counters), hipMemcpyDeviceToHost, stream);
        if (e == hipSuccess) e = hipStreamSynchronize(stream);
        if (e == hipSuccess) e = drain_events();
        if (e != hipSuccess) rc = fail(RT_ERR_DEVICE, hipGetErrorString(e));
    } else {
        (void)hipStreamSynchronize(stream);
    }
    cleanup_events();
    if (rc != RT_OK) return rc;
    if (stats) {
        std::memset(stats, 0, sizeof(*stats));
        stats->total_rays = host_counters.total;
        stats->rays_exceeded_depth = host_counters.exceeded;
        stats->rays_left_scene = host_counters.left;
        stats->shadow_rays = host_counters.shadow;
        stats->render_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
        stats->traverse_seconds = traverse_ms * 1e-3;
        stats->traverse_launches = traverse_launches;
        stats->traverse_rays = traverse_rays;
    }
    return RT_OK;
}

}  // namespace

extern "C" {

const char* rt_last_error(void) { return tl_error.c_str(); }

int rt_init(const int* gpu_ids, int n) {
    int count = 0;
    HIP_TRY(hipGetDeviceCount(&count));
    if (count <= 0) return fail(RT_ERR_DEVICE, "no HIP device visible");
    int dev = (gpu_ids && n > 0) ? gpu_ids[0] : 0;
    if (dev < 0 || dev >= count) return fail(RT_ERR_INVALID, "device ordinal out of range");
    HIP_TRY(hipSetDevice(dev));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, dev));
    if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0)
        return fail(RT_ERR_UNSUPPORTED, std::string("this library is built for gfx950 only, device is ") + prop.gcnArchName);
    g_device = dev;
    g_initialized = true;
    return RT_OK;
}

void rt_shutdown(void) { g_initialized = false; }

int rt_scene_create(const rt_scene_desc* d, rt_scene** out) {
    if (!g_initialized) return fail(RT_ERR_NOT_INITIALIZED, "rt_init has not been called");
    if (!d || !out) return fail(RT_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(g_device));
    // the array-level checks of Scene.Validate (scene.go:29-79) and of each Validate() it calls
    for (uint64_t i = 0; i < d->n_nodes; i++) {
        const rt_node& n = d->nodes[i];
        if (int rc = check_shape_ref(*d, n.shape_kind, n.shape_index, 0)) return rc;
        if (n.material < 0 || (uint64_t)n.material >= d->n_materials) return fail(RT_ERR_INVALID, "node " + std::to_string(i) + " has no material");
    }
    for (uint64_t i = 0; i < d->n_transformed; i++)
        if (int rc = check_shape_ref(*d, RT_SHAPE_TRANSFORMED, (int)i, 0)) return rc;
    for (uint64_t i = 0; i < d->n_spheres; i++)
        if (!(d->spheres[i].radius > 0)) return fail(RT_ERR_INVALID, "invalid Sphere radius");
    for (uint64_t i = 0; i < d->n_quads; i++)
        if (!(d->quads[i].width > 0) || !(d->quads[i].height > 0)) return fail(RT_ERR_INVALID, "invalid Quad size");
    for (uint64_t i = 0; i < d->n_cylinders; i++)
        if (!(d->cylinders[i].radius > 0) || !(d->cylinders[i].height > 0)) return fail(RT_ERR_INVALID, "invalid Cylinder size");
    for (uint64_t i = 0; i < d->n_meshes; i++) {
        const rt_mesh& m = d->meshes[i];
        if (m.n_faces == 0) return fail(RT_ERR_INVALID, "Mesh must contain at least one face");
        if (m.first_face + m.n_faces > d->n_faces || m.n_faces > 0x7fffffffull) return fail(RT_ERR_INVALID, "mesh face range out of bounds");
    }
    for (uint64_t i = 0; i < d->n_materials; i++) {
        const rt_material& m = d->materials[i];
        if (m.kind < RT_MAT_LAMBERTIAN || m.kind > RT_MAT_DEBUG_NORMAL) return fail(RT_ERR_INVALID, "unknown material kind");
        if ((m.kind == RT_MAT_LAMBERTIAN || m.kind == RT_MAT_EMITTER) && (m.texture < 0 || (uint64_t)m.texture >= d->n_textures))
            return fail(RT_ERR_INVALID, "material texture index out of range");
        if (m.kind == RT_MAT_METAL && (m.fuzz < 0 || m.fuzz > 1)) return fail(RT_ERR_INVALID, "invalid Metal fuzz must be in the range [0, 1]");
        if (m.kind == RT_MAT_DIELECTRIC && (m.ior_interior < 1 || m.ior_exterior < 1 || m.roughness < 0 || m.roughness > 1))
            return fail(RT_ERR_INVALID, "invalid Dielectric parameters");
    }
    for (uint64_t i = 0; i < d->n_textures; i++) {
        const rt_texture& t = d->textures[i];
        if (t.kind == RT_TEX_CHECKERBOARD) {
            if (t.odd < 0 || t.even < 0 || (uint64_t)t.odd >= d->n_textures || (uint64_t)t.even >= d->n_textures)
                return fail(RT_ERR_INVALID, "TextureCheckerboard child out of range");
            if (!(t.frequency > 0)) return fail(RT_ERR_INVALID, "error TextureChecker.Frequency is negative");
        } else if (t.kind == RT_TEX_IMAGE) {
            if (t.image < 0 || (uint64_t)t.image >= d->n_images) return fail(RT_ERR_INVALID, "image texture is nil");
        } else if (t.kind != RT_TEX_UNIFORM) {
            return fail(RT_ERR_INVALID, "unknown texture kind");
        }
    }
    for (uint64_t i = 0; i < d->n_images; i++) {
        const rt_image& im = d->images[i];
        if (im.width <= 0 || im.height <= 0 || !im.pixels || (im.format != RT_IMG_NRGBA8 && im.format != RT_IMG_RGBA16))
            return fail(RT_ERR_INVALID, "bad image");
    }
    for (uint64_t i = 0; i < d->n_lights; i++)
        if (d->lights[i].intensity[0] < 0 || d->lights[i].intensity[1] < 0 || d->lights[i].intensity[2] < 0)
            return fail(RT_ERR_INVALID, "invalid PointLight RadiantIntensity (should be non-negative)");
    if (d->n_nodes > 0x7fffffffull || d->n_lights > 65535) return fail(RT_ERR_UNSUPPORTED, "too many nodes or lights");

    auto sc = std::make_unique<rt_scene>();
    sc->device = g_device;
    std::vector<DNode> nodes(d->n_nodes);
    for (uint64_t i = 0; i < d->n_nodes; i++) nodes[i] = {d->nodes[i].shape_kind, d->nodes[i].shape_index, d->nodes[i].material, 0};
    std::vector<DSphere> spheres(d->n_spheres);
    for (uint64_t i = 0; i < d->n_spheres; i++) {
        std::memcpy(spheres[i].c, d->spheres[i].center, sizeof(double) * 3);
        spheres[i].r = d->spheres[i].radius;
        spheres[i].uvmap = d->spheres[i].uvmap;
        spheres[i].pad = 0;
    }
    std::vector<DQuad> quads(d->n_quads);
    for (uint64_t i = 0; i < d->n_quads; i++) quads[i] = make_quad(d->quads[i]);
    std::vector<DCylinder> cyls(d->n_cylinders);
    for (uint64_t i = 0; i < d->n_cylinders; i++) cyls[i] = make_cylinder(d->cylinders[i]);
    std::vector<DTri> tris(d->n_triangles);
    for (uint64_t i = 0; i < d->n_triangles; i++) {
        const rt_triangle& t = d->triangles[i];
        if (t.kind < RT_TRI_TRIANGLE || t.kind > RT_TRI_FACE) return fail(RT_ERR_INVALID, "unknown triangle kind");
        std::memcpy(tris[i].p0, t.p[0], 24); std::memcpy(tris[i].p1, t.p[1], 24); std::memcpy(tris[i].p2, t.p[2], 24);
        std::memcpy(tris[i].uv, t.uv, 48);
        st3(tris[i].n, unit(ld3(t.normal)));
        tris[i].kind = t.kind;
        tris[i].pad = 0;
    }
    std::vector<DTransformed> xforms(d->n_transformed);
    for (uint64_t i = 0; i < d->n_transformed; i++) {
        const rt_transformed& t = d->transformed[i];
        xforms[i].child_kind = t.child_kind;
        xforms[i].child_index = t.child_index;
        xforms[i].fwd = make_xform(t.translation, t.rotation, t.scale);
        xforms[i].inv = invert(xforms[i].fwd);
    }
    // meshes: build every BVH, concatenate the per-mesh arrays
    std::vector<DMesh> meshes(d->n_meshes);
    std::vector<RefNode> ref_nodes;
    std::vector<FastNode> fast_nodes;
    std::vector<double> prim_pos;
    std::vector<int32_t> prim_face;
    std::vector<double> face_uv(d->n_faces * 6);
    for (uint64_t i = 0; i < d->n_faces; i++)
        for (int v = 0; v < 3; v++) { face_uv[i * 6 + v * 2] = d->faces[i].v[v][3]; face_uv[i * 6 + v * 2 + 1] = d->faces[i].v[v][4]; }
    sc->bvh_info.resize(d->n_meshes);
    for (uint64_t i = 0; i < d->n_meshes; i++) {
        MeshBvh b;
        build_mesh_bvh(d->faces + d->meshes[i].first_face, d->meshes[i].n_faces, b);
        DMesh& m = meshes[i];
        m.ref_off = (int64_t)ref_nodes.size();
        m.fast_off = (int64_t)fast_nodes.size();
        m.prim_off = (int64_t)prim_face.size();
        m.face_off = (int64_t)d->meshes[i].first_face;
        std::memcpy(m.center, b.center, sizeof(m.center));
        std::memcpy(m.root_box, b.root_box, sizeof(m.root_box));
        m.pad = b.pad;
        m.n_faces = (int32_t)d->meshes[i].n_faces;
        ref_nodes.insert(ref_nodes.end(), b.ref_nodes.begin(), b.ref_nodes.end());
        fast_nodes.insert(fast_nodes.end(), b.fast_nodes.begin(), b.fast_nodes.end());
        prim_pos.insert(prim_pos.end(), b.prim_pos.begin(), b.prim_pos.end());
        prim_face.insert(prim_face.end(), b.prim_face.begin(), b.prim_face.end());
        sc->bvh_info[i] = b.info;
    }
    std::vector<rt_material> materials(d->materials, d->materials + d->n_materials);
    std::vector<rt_texture> textures(d->textures, d->textures + d->n_textures);
    std::vector<rt_light> lights(d->lights, d->lights + d->n_lights);
    std::vector<DImage> images(d->n_images);
    for (uint64_t i = 0; i < d->n_images; i++) {
        const rt_image& im = d->images[i];
        const size_t bytes = (size_t)im.width * im.height * (im.format == RT_IMG_RGBA16 ? 8 : 4);
        uint8_t* px = nullptr;
        HIP_TRY(sc->mem.alloc(bytes, &px));
        HIP_TRY(hipMemcpy(px, im.pixels, bytes, hipMemcpyHostToDevice));
        images[i] = {px, im.width, im.height, im.format, 0};
    }
    DScene& dev = sc->dev;
    HIP_TRY(sc->mem.upload(nodes, &dev.nodes));
    HIP_TRY(sc->mem.upload(spheres, &dev.spheres));
    HIP_TRY(sc->mem.upload(quads, &dev.quads));
    HIP_TRY(sc->mem.upload(cyls, &dev.cylinders));
    HIP_TRY(sc->mem.upload(tris, &dev.tris));
    HIP_TRY(sc->mem.upload(xforms, &dev.xforms));
    HIP_TRY(sc->mem.upload(meshes, &dev.meshes));
    HIP_TRY(sc->mem.upload(ref_nodes, &dev.ref_nodes));
    HIP_TRY(sc->mem.upload(fast_nodes, &dev.fast_nodes));
    HIP_TRY(sc->mem.upload(prim_pos, &dev.prim_pos));
    HIP_TRY(sc->mem.upload(prim_face, &dev.prim_face));
    HIP_TRY(sc->mem.upload(face_uv, &dev.face_uv));
    HIP_TRY(sc->mem.upload(materials, &dev.materials));
    HIP_TRY(sc->mem.upload(textures, &dev.textures));
    HIP_TRY(sc->mem.upload(images, &dev.images));
    HIP_TRY(sc->mem.upload(lights, &dev.lights));
    dev.n_nodes = (int32_t)d->n_nodes;
    dev.n_lights = (int32_t)d->n_lights;
    *out = sc.release();
    return RT_OK;
}

void rt_scene_destroy(rt_scene* scene) {
    if (!scene) return;
    (void)hipSetDevice(scene->device);
    if (scene->work) {
        if (scene->work->pool) (void)hipFree(scene->work->pool);
        if (scene->work->rgba) (void)hipFree(scene->work->rgba);
        if (scene->work->film) (void)hipFree(scene->work->film);
    }
    delete scene;
}

int rt_scene_bvh_info(const rt_scene* scene, uint64_t mesh_index, rt_bvh_info* out) {
    if (!scene || !out || mesh_index >= scene->bvh_info.size()) return fail(RT_ERR_INVALID, "bad mesh index");
    *out = scene->bvh_info[mesh_index];
    return RT_OK;
}

void rt_cancel(rt_scene* scene) {
    if (scene) scene->cancel.store(true);
}

int rt_render_film(rt_scene* scene, const rt_camera_desc* camera, const rt_render_opts* opts, double* film_dev, rt_stats* stats) {
    return render_into_film(scene, camera, opts, film_dev, stats);
}

int rt_film_resolve(const double* film_dev, int32_t dx, int32_t dy, int32_t rays_per_pixel, uint8_t* rgba_out, size_t stride, void* stream_) {
    if (!g_initialized) return fail(RT_ERR_NOT_INITIALIZED, "rt_init has not been called");
    if (!film_dev || !rgba_out || dx <= 0 || dy <= 0 || rays_per_pixel <= 0 || stride < (size_t)dx * 4) return fail(RT_ERR_INVALID, "bad argument");
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    uint8_t* rgba = nullptr;
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&rgba), (size_t)dx * dy * 4));
    launch_resolve(film_dev, dx, dy, rays_per_pixel, rgba, stream);
    hipError_t e = hipMemcpy2DAsync(rgba_out, stride, rgba, (size_t)dx * 4, (size_t)dx * 4, (size_t)dy, hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    (void)hipFree(rgba);
    if (e != hipSuccess) return fail(RT_ERR_DEVICE, hipGetErrorString(e));
    return RT_OK;
}

int rt_render(rt_scene* scene, const rt_camera_desc* camera, const rt_render_opts* opts, uint8_t* rgba_out, size_t stride, rt_stats* stats) {
    if (!scene || !opts || !rgba_out) return fail(RT_ERR_INVALID, "null argument");
    if (opts->dx <= 0 || opts->dy <= 0) return fail(RT_ERR_INVALID, "bad Dx / Dy must be positive");
    if (stride < (size_t)opts->dx * 4) return fail(RT_ERR_INVALID, "stride smaller than a row");
    if (int rc = ensure_work(scene, std::max(opts->max_ray_depth, 1), opts->dx, opts->dy)) return rc;
    WorkBuffers& w = *scene->work;
    const size_t n = (size_t)opts->dx * opts->dy * 3;
    if (w.film_doubles < n) {
        if (w.film) (void)hipFree(w.film);
        w.film = nullptr;
        w.film_doubles = 0;
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&w.film), n * sizeof(double)));
        w.film_doubles = n;
    }
    hipStream_t stream = reinterpret_cast<hipStream_t>(opts->stream);
    const auto t0 = std::chrono::steady_clock::now();
    HIP_TRY(hipMemsetAsync(w.film, 0, n * sizeof(double), stream));
    if (int rc = render_into_film(scene, camera, opts, w.film, stats)) return rc;
    launch_resolve(w.film, opts->dx, opts->dy, opts->rays_per_pixel, w.rgba, stream);
    HIP_TRY(hipMemcpy2DAsync(rgba_out, stride, w.rgba, (size_t)opts->dx * 4, (size_t)opts->dx * 4, (size_t)opts->dy, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    if (stats) stats->render_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return RT_OK;
}

int rt_intersect_batch(rt_scene* scene, const double* o, const double* d, uint64_t n, double tmin, double tmax, int mode, rt_hit* hits) {
    if (!g_initialized) return fail(RT_ERR_NOT_INITIALIZED, "rt_init has not been called");
    if (!scene || (n && (!o || !d || !hits))) return fail(RT_ERR_INVALID, "null argument");
    if (mode != RT_MODE_REFERENCE && mode != RT_MODE_FAST) return fail(RT_ERR_INVALID, "unknown mode");
    if (n == 0) return RT_OK;
    HIP_TRY(hipSetDevice(scene->device));
    DeviceBuffers tmp;
    double *od = nullptr, *dd = nullptr;
    rt_hit* hd = nullptr;
    HIP_TRY(tmp.alloc(n * 3, &od));
    HIP_TRY(tmp.alloc(n * 3, &dd));
    HIP_TRY(tmp.alloc(n, &hd));
    HIP_TRY(hipMemcpy(od, o, n * 24, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dd, d, n * 24, hipMemcpyHostToDevice));
    launch_intersect_batch(scene->dev, od, dd, n, tmin, tmax, mode, hd, nullptr);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(hits, hd, n * sizeof(rt_hit), hipMemcpyDeviceToHost));
    return RT_OK;
}

int rt_cast_batch(const rt_camera_desc* camera, const double* st, const uint32_t* pixel_sample, uint64_t n, int64_t seed, double* o_out,
                  double* d_out) {
    if (!g_initialized) return fail(RT_ERR_NOT_INITIALIZED, "rt_init has not been called");
    if (!camera || (n && (!st || !pixel_sample || !o_out || !d_out))) return fail(RT_ERR_INVALID, "null argument");
    if (n == 0) return RT_OK;
    HIP_TRY(hipSetDevice(g_device));
    DCamera cam;
    if (int rc = make_camera(*camera, cam)) return rc;
    DeviceBuffers tmp;
    double *std_ = nullptr, *od = nullptr, *dd = nullptr;
    uint32_t* psd = nullptr;
    HIP_TRY(tmp.alloc(n * 2, &std_));
    HIP_TRY(tmp.alloc(n * 2, &psd));
    HIP_TRY(tmp.alloc(n * 3, &od));
    HIP_TRY(tmp.alloc(n * 3, &dd));
    HIP_TRY(hipMemcpy(std_, st, n * 16, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(psd, pixel_sample, n * 8, hipMemcpyHostToDevice));
    launch_cast_batch(cam, std_, psd, n, (uint32_t)((uint64_t)seed & 0xffffffffu), (uint32_t)((uint64_t)seed >> 32), od, dd, nullptr);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(o_out, od, n * 24, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(d_out, dd, n * 24, hipMemcpyDeviceToHost));
    return RT_OK;
}

}  // extern "C"
