// The wavefront render loop: replacement for the body of renderScene (phys/render.go:285-349).
//
// The reference hands 16x16 tiles to NumCPU goroutines, each of which recurses through
// tracePath per sample.  Here the image (this process's row bands of it) is cut into passes of
// at most `q` paths; a pass is one k_primary launch (a persistent grid draining a queue of small pixel blocks) and
// then, only while rays were scattered, generations of k_direct (light loop of the hits just made) / k_trace (the
// scattered rays, Material.Resolve included).  A scene whose materials all just emit (the scan configuration:
// Emitter + image texture) never leaves k_primary and is rendered without a single host synchronisation per pass.
// For orthographic and pinhole cameras the tiles that cannot see any node are left out of the primary launch's
// schedule (Occupancy below) and their rays counted here.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstddef>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <tuple>
#include <vector>

#include "host_common.h"
#include "host_threads.h"

namespace rt {

struct WorkBuffers {
    DeviceBuffers mem;
    uint32_t q = 0;  // paths per launch at most
    uint32_t primary_waves = 0;  // persistent waves of a primary launch at most (the stripes are sized for that many)
    uint32_t stripe_rays = 0;    // rays of one stripe a trace launch takes at most
    int ray_stripes = kQueueStripes;  // stripes of a ray queue: twice as many when the scene queues two classes of rays
    Queues queues{};
    Queues* d_queues = nullptr;   // device copies read by k_primary
    DCamera* d_camera = nullptr;
    RayQ pool{};     // overflow stack for generations that outgrow q (Dielectric's two children)
    uint64_t pool_capacity = 0;
    void* pool_base = nullptr;
    // Film sums, the 8-bit image and its pinned staging copy (the caller's buffer is pageable) for rt_render; a frame
    // loop (rt_render_frames) alternates between two such sets so that one frame copies out while the next is traced.
    struct Frame {
        uint8_t* rgba = nullptr;
        uint8_t* rgba_host = nullptr;
        size_t rgba_bytes = 0;
        double* film = nullptr;
        size_t film_doubles = 0;
    } frame[2];
    // Cost-ordered schedule of the primary launch, measured by a short pilot pass and kept while the
    // camera and the frame stay the same (animation frames and progressive passes re-use it).
    uint32_t* tile_cost = nullptr;
    int32_t* tile_order = nullptr;
    size_t order_tiles = 0;
    std::string order_key;
    bool order_valid = false;
    uint32_t queue_off[8] = {}, queue_len[8] = {}, queue_cheap[8] = {};
    uint64_t culled_pixels = 0;  // pixels of the tiles the schedule leaves out (outside the screen bounds of the scene)
    bool is_static = false;      // the buffers live in the library's static scratch: nothing to allocate, grow or free
};

void destroy_work(WorkBuffers* w) {
    if (!w || w->is_static) return;
    if (w->pool_base) (void)hipFree(w->pool_base);
    for (WorkBuffers::Frame& f : w->frame) {
        if (f.rgba) (void)hipFree(f.rgba);
        if (f.rgba_host) (void)hipHostFree(f.rgba_host);
        if (f.film) (void)hipFree(f.film);
    }
    if (w->tile_cost) (void)hipFree(w->tile_cost);
    if (w->tile_order) (void)hipFree(w->tile_order);
    delete w;
}

namespace {

size_t env_size(const char* name, size_t fallback) {
    const char* v = std::getenv(name);
    if (!v || !*v) return fallback;
    return (size_t)std::strtoull(v, nullptr, 10);
}

// Carve structure-of-arrays views out of one allocation of cap * bytes-per-element.
void carve(RayQ& r, void* base, size_t cap) {
    char* p = static_cast<char*>(base);
    for (int k = 0; k < 3; k++) { r.o[k] = reinterpret_cast<double*>(p); p += cap * 8; }
    for (int k = 0; k < 3; k++) { r.d[k] = reinterpret_cast<double*>(p); p += cap * 8; }
    for (int k = 0; k < 3; k++) { r.rad[k] = reinterpret_cast<double*>(p); p += cap * 8; }
    r.pixel = reinterpret_cast<uint32_t*>(p); p += cap * 4;
    r.sample = reinterpret_cast<uint32_t*>(p); p += cap * 4;
    r.path = reinterpret_cast<uint32_t*>(p); p += cap * 4;
    r.depth_n = reinterpret_cast<uint32_t*>(p);
}
void carve(HitQ& h, void* base, size_t cap) {
    char* p = static_cast<char*>(base);
    h.t = reinterpret_cast<double*>(p); p += cap * 8;
    h.ray = reinterpret_cast<uint32_t*>(p); p += cap * 4;
    h.node = reinterpret_cast<int32_t*>(p); p += cap * 4;
    h.leaf = reinterpret_cast<int32_t*>(p);
}
void carve(SurfQ& s, void* base, size_t cap) {
    char* p = static_cast<char*>(base);
    for (int k = 0; k < 3; k++) { s.p[k] = reinterpret_cast<double*>(p); p += cap * 8; }
    for (int k = 0; k < 3; k++) { s.n[k] = reinterpret_cast<double*>(p); p += cap * 8; }
    for (int k = 0; k < 3; k++) { s.wo[k] = reinterpret_cast<double*>(p); p += cap * 8; }
    for (int k = 0; k < 3; k++) { s.albedo[k] = reinterpret_cast<double*>(p); p += cap * 8; }
    s.pixel = reinterpret_cast<uint32_t*>(p); p += cap * 4;
    s.node = reinterpret_cast<int32_t*>(p); p += cap * 4;
    s.material = reinterpret_cast<int32_t*>(p); p += cap * 4;
    s.inside = reinterpret_cast<int32_t*>(p);
}

// Field-wise copy of n rays between two structure-of-arrays queues.
hipError_t copy_rays(const RayQ& dst, uint64_t dst_at, const RayQ& src, uint64_t src_at, uint64_t n, hipStream_t s) {
    hipError_t e = hipSuccess;
    for (int k = 0; k < 3 && e == hipSuccess; k++) e = hipMemcpyAsync(dst.o[k] + dst_at, src.o[k] + src_at, n * 8, hipMemcpyDeviceToDevice, s);
    for (int k = 0; k < 3 && e == hipSuccess; k++) e = hipMemcpyAsync(dst.d[k] + dst_at, src.d[k] + src_at, n * 8, hipMemcpyDeviceToDevice, s);
    for (int k = 0; k < 3 && e == hipSuccess; k++) e = hipMemcpyAsync(dst.rad[k] + dst_at, src.rad[k] + src_at, n * 8, hipMemcpyDeviceToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(dst.pixel + dst_at, src.pixel + src_at, n * 4, hipMemcpyDeviceToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(dst.sample + dst_at, src.sample + src_at, n * 4, hipMemcpyDeviceToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(dst.path + dst_at, src.path + src_at, n * 4, hipMemcpyDeviceToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(dst.depth_n + dst_at, src.depth_n + src_at, n * 4, hipMemcpyDeviceToDevice, s);
    return e;
}

// want_paths: the camera paths of the render about to run (pixels of this device's share x samples of the call).  The
// queues of a scattering scene cost 650 bytes per path of a launch (21 GB at 2^25 paths; 1000 bytes with two classes of
// rays), so they are sized for the frame at hand, between 2^18 and 2^25 paths, and grow when a later frame needs more: a
// host that keeps many small scenes alive does not pay for a 4K frame in each.  A frame of more paths than that is
// traced in passes, each with its tail of small launches for the late generations; one more doubling (a 1080p frame at
// 32 samples in one pass instead of two: 8 % faster) is taken only when it leaves more than half of the device's memory
// free.  RT_QUEUE_RAYS fixes the size instead.
int ensure_work(const rt_scene* sc, Replica& rep, uint64_t want_paths) {
    size_t q_want = (size_t)1 << 18, q_top = (size_t)1 << 25;
    if (rep.work && rep.work->q > q_top) {
        q_top = rep.work->q;  // taken before
    } else if (want_paths > q_top && !sc->all_terminal) {
        size_t free_b = 0, total_b = 0;
        const size_t cost = 2 * q_top * (rep.dev.walk_classes ? 1000 : 650);
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b > cost && free_b - cost > total_b / 2) q_top *= 2;
    }
    while (q_want < q_top && q_want < want_paths) q_want <<= 1;
    if (std::getenv("RT_QUEUE_RAYS")) q_want = env_size("RT_QUEUE_RAYS", (size_t)1 << 25);
    if (q_want < 64 * 64) q_want = 64 * 64;
    if (rep.work && (sc->all_terminal || rep.work->q >= q_want)) return RT_OK;  // a scene that never shades never queues
    // built aside and published only when complete: a failed allocation must not leave half-made queues behind
    struct Holder {
        WorkBuffers* w = new WorkBuffers();
        ~Holder() { destroy_work(w); }
    } hold;
    WorkBuffers* w = hold.w;
    w->q = (uint32_t)q_want;
    const size_t q = w->q;
    char *rays0 = nullptr, *rays1 = nullptr, *surfs = nullptr, *hits = nullptr;
    // a scene that never shades needs none of the queues (kernels still get valid pointers)
    const size_t qq = sc->all_terminal ? 64 : q;
    // Stripe capacities.  A primary launch traces at most q paths in rounds of a wave that are at least 90 % full, and
    // wave w sends its r-th round to stripe (w + r) % stripes: a stripe receives at most rounds / stripes + waves rounds
    // of 64 surface points and 128 children.  A trace launch takes at most stripe_rays rays of each stripe and writes
    // at most ceil(stripe_rays / 64) waves' children and surface points to each stripe (see k_trace).
    w->primary_waves = (uint32_t)std::min<size_t>(4096, std::max<size_t>(64, qq / 2048));
    const size_t rounds = (qq + 63) / 64 * 10 / 9 / kQueueStripes + w->primary_waves + 2;
    const size_t ray_cap = rounds * 128, surf_cap = rounds * 64;
    // with two classes of rays (DScene::walk_classes) a class has stripes of its own, and of the trace waves that share a
    // grid x two write to the same stripe: a launch then takes half as many rays of a stripe
    w->ray_stripes = rep.dev.walk_classes ? kRayStripes : kQueueStripes;
    w->stripe_rays = (uint32_t)((rep.dev.walk_classes ? rounds / 2 : rounds) * 64 - 64);
    HIP_TRY(w->mem.alloc((size_t)w->ray_stripes * ray_cap * kRayBytes, &rays0));
    HIP_TRY(w->mem.alloc((size_t)w->ray_stripes * ray_cap * kRayBytes, &rays1));
    HIP_TRY(w->mem.alloc(kQueueStripes * surf_cap * kSurfBytes, &surfs));
    // Hit records: a launch traces at most q rays, and every wave of k_walk may leave one reserved block partly unused;
    // a wave that finds a stripe full moves on to the next, so only the sum over the stripes has to fit.
    const size_t hit_cap = (sc->all_terminal || rep.dev.n_meshes == 0) ? kClaim : ((qq + (size_t)kWalkWavesMax * kClaim) / kQueueStripes / kClaim + 2) * kClaim;
    HIP_TRY(w->mem.alloc(kQueueStripes * hit_cap * kHitBytes, &hits));
    carve(w->queues.hits, hits, kQueueStripes * hit_cap);
    w->queues.hit_stripe_cap = (uint32_t)hit_cap;
    if (rep.dev.n_meshes > 0 && !sc->all_terminal) HIP_TRY(w->mem.alloc((size_t)kWalkWavesMax * 64 * (size_t)rep.dev.walk_cap, &w->queues.walk_spill));
    HIP_TRY(w->mem.alloc(1, &w->queues.counters));
    carve(w->queues.rays[0], rays0, (size_t)w->ray_stripes * ray_cap);
    carve(w->queues.rays[1], rays1, (size_t)w->ray_stripes * ray_cap);
    carve(w->queues.surfs, surfs, kQueueStripes * surf_cap);
    w->queues.ray_stripe_cap = (uint32_t)ray_cap;
    w->queues.surf_stripe_cap = (uint32_t)surf_cap;
    HIP_TRY(w->mem.alloc(1, &w->d_queues));
    HIP_TRY(w->mem.alloc(1, &w->d_camera));
    HIP_TRY(hipMemcpy(w->d_queues, &w->queues, sizeof(Queues), hipMemcpyHostToDevice));
    if (rep.work) {  // larger queues replace smaller ones: what does not depend on their size moves over
        WorkBuffers* old = rep.work;
        for (int k = 0; k < 2; k++) std::swap(w->frame[k], old->frame[k]);
        std::swap(w->tile_cost, old->tile_cost);
        std::swap(w->tile_order, old->tile_order);
        std::swap(w->order_tiles, old->order_tiles);
        std::swap(w->order_key, old->order_key);
        std::swap(w->order_valid, old->order_valid);
        std::swap(w->culled_pixels, old->culled_pixels);
        for (int k = 0; k < 8; k++) { std::swap(w->queue_off[k], old->queue_off[k]); std::swap(w->queue_len[k], old->queue_len[k]); std::swap(w->queue_cheap[k], old->queue_cheap[k]); }
        destroy_work(old);
    }
    rep.work = w;
    hold.w = nullptr;
    return RT_OK;
}

// Image of world-space boxes in the camera's image coordinates (s, t), the arguments of Camera.Cast: a ray that can
// touch the box has s in [r[0], r[1]] and t in [r[2], r[3]].  Orthographic (camera_orthographic.go:24-46): the ray
// through (s, t) starts at from + u * fov_w * (s - 0.5) + v * fov_h * (t - 0.5) and runs along -w, so a world point lies
// on exactly the ray whose (s, t) are its u / v coordinates.  Pinhole (camera_pinhole.go:20-34): the ray from `origin`
// through llc + s * horizontal + t * vertical; a point in front of the camera is solved for (s, t).  The other cameras
// (lens, distortion) have no such image.
struct Projector {
    const DCamera* cam = nullptr;
    double inv[3][3] = {};
    bool init(const DCamera& c) {
        cam = &c;
        if (c.kind == RT_CAM_ORTHOGRAPHIC) return c.fov_w > 0 && c.fov_h > 0;
        if (c.kind != RT_CAM_PINHOLE) return false;
        double m[3][3];  // columns: llc - origin, horizontal, vertical; P - origin = M * (l, l * s, l * t)
        for (int r = 0; r < 3; r++) { m[r][0] = c.llc[r] - c.origin[r]; m[r][1] = c.horizontal[r]; m[r][2] = c.vertical[r]; }
        const double det = m[0][0] * (m[1][1] * m[2][2] - m[1][2] * m[2][1]) - m[0][1] * (m[1][0] * m[2][2] - m[1][2] * m[2][0]) +
                           m[0][2] * (m[1][0] * m[2][1] - m[1][1] * m[2][0]);
        if (!(std::fabs(det) > 0) || !std::isfinite(det)) return false;
        inv[0][0] = (m[1][1] * m[2][2] - m[1][2] * m[2][1]) / det; inv[0][1] = (m[0][2] * m[2][1] - m[0][1] * m[2][2]) / det; inv[0][2] = (m[0][1] * m[1][2] - m[0][2] * m[1][1]) / det;
        inv[1][0] = (m[1][2] * m[2][0] - m[1][0] * m[2][2]) / det; inv[1][1] = (m[0][0] * m[2][2] - m[0][2] * m[2][0]) / det; inv[1][2] = (m[0][2] * m[1][0] - m[0][0] * m[1][2]) / det;
        inv[2][0] = (m[1][0] * m[2][1] - m[1][1] * m[2][0]) / det; inv[2][1] = (m[0][1] * m[2][0] - m[0][0] * m[2][1]) / det; inv[2][2] = (m[0][0] * m[1][1] - m[0][1] * m[1][0]) / det;
        return true;
    }
    // false: the box has no bounded image (not finite, or a corner beside or behind a pinhole camera)
    bool box(const double lo[3], const double hi[3], double r[4]) const {
        double lo_s = 1e300, hi_s = -1e300, lo_t = 1e300, hi_t = -1e300;
        if (cam->kind == RT_CAM_ORTHOGRAPHIC) {  // a linear map: the image of the box is the image of its centre +- the sum over the axes
            double cs = 0, ct = 0, hs = 0, ht = 0;
            for (int k = 0; k < 3; k++) {
                const double c = 0.5 * (lo[k] + hi[k]) - cam->from[k], h = 0.5 * (hi[k] - lo[k]);
                cs += c * cam->u[k]; ct += c * cam->v[k];
                hs += std::fabs(cam->u[k]) * h; ht += std::fabs(cam->v[k]) * h;
            }
            if (!std::isfinite(cs) || !std::isfinite(ct) || !std::isfinite(hs) || !std::isfinite(ht)) return false;
            lo_s = 0.5 + (cs - hs) / cam->fov_w; hi_s = 0.5 + (cs + hs) / cam->fov_w;
            lo_t = 0.5 + (ct - ht) / cam->fov_h; hi_t = 0.5 + (ct + ht) / cam->fov_h;
        } else
        for (int c = 0; c < 8; c++) {
            const double p[3] = {(c & 1) ? hi[0] : lo[0], (c & 2) ? hi[1] : lo[1], (c & 4) ? hi[2] : lo[2]};
            double s, t;
            if (cam->kind == RT_CAM_ORTHOGRAPHIC) {
                double du = 0, dv = 0;
                for (int k = 0; k < 3; k++) { du += (p[k] - cam->from[k]) * cam->u[k]; dv += (p[k] - cam->from[k]) * cam->v[k]; }
                s = 0.5 + du / cam->fov_w;
                t = 0.5 + dv / cam->fov_h;
            } else {
                double x[3] = {0, 0, 0};
                for (int r2 = 0; r2 < 3; r2++)
                    for (int k = 0; k < 3; k++) x[r2] += inv[r2][k] * (p[k] - cam->origin[k]);
                if (!(x[0] > 1e-9)) return false;
                s = x[1] / x[0];
                t = x[2] / x[0];
            }
            if (!std::isfinite(s) || !std::isfinite(t)) return false;
            lo_s = std::min(lo_s, s); hi_s = std::max(hi_s, s);
            lo_t = std::min(lo_t, t); hi_t = std::max(hi_t, t);
        }
        // generous against the rounding above and in Camera.Cast: a millionth of the frame plus a millionth of the extent
        const double ms = 1e-6 * (1.0 + std::fabs(lo_s) + std::fabs(hi_s)), mt = 1e-6 * (1.0 + std::fabs(lo_t) + std::fabs(hi_t));
        r[0] = lo_s - ms; r[1] = hi_s + ms; r[2] = lo_t - mt; r[3] = hi_t + mt;
        return true;
    }
};

// Which 8x8 pixel blocks of the image (block (bx, by) = columns 8 bx .., image rows 8 by .., row 0 = top) can see
// anything of the scene.  Every node's padded bounds are projected; an untransformed mesh is followed down its fast
// tree until a box covers no more than a few blocks, so the map follows the mesh's outline, not just its bounds.
// Conservative throughout (padded boxes, one extra pixel, outward rounding): a block left unmarked cannot be reached
// by any camera ray.  Returns false when there is no such map (see Projector); then nothing is culled.
struct Occupancy {
    int bw = 0, bh = 0, dx = 0, dy = 0;
    std::vector<uint8_t> seen;
    void mark(const double r[4]) {
        auto px = [](double v) { return (int64_t)std::max(-1.0e9, std::min(1.0e9, v)); };
        const int64_t x0 = px(std::floor(r[0] * dx) - 1), x1 = px(std::ceil(r[1] * dx) + 1);
        const int64_t y0 = px(std::floor(r[2] * dy) - 1), y1 = px(std::ceil(r[3] * dy) + 1);  // counted upward, as the reference's y
        const int64_t row0 = (int64_t)dy - 1 - y1, row1 = (int64_t)dy - 1 - y0;
        const int64_t bx0 = std::max<int64_t>(0, x0 >> 3), bx1 = std::min<int64_t>(bw - 1, x1 >> 3);
        const int64_t by0 = std::max<int64_t>(0, row0 >> 3), by1 = std::min<int64_t>(bh - 1, row1 >> 3);
        // several threads mark at once (build_occupancy): every store writes the same value, relaxed atomics keep that defined
        for (int64_t by = by0; by <= by1; by++)
            for (int64_t bx = bx0; bx <= bx1; bx++) __atomic_store_n(&seen[(size_t)(by * bw + bx)], (uint8_t)1, __ATOMIC_RELAXED);
    }
    bool small(const double r[4]) const { return (r[1] - r[0]) * dx <= 16.0 && (r[3] - r[2]) * dy <= 16.0; }
};

bool build_occupancy(const rt_scene* sc, const DCamera& cam, int dx, int dy, Occupancy& occ) {
    Projector proj;
    if (!proj.init(cam)) return false;
    occ.dx = dx; occ.dy = dy;
    occ.bw = (dx + 7) / 8; occ.bh = (dy + 7) / 8;
    occ.seen.assign((size_t)occ.bw * occ.bh, 0);
    std::vector<int32_t> stack;
    for (size_t i = 0; i < sc->nodes.size(); i++) {
        const double* nb = &sc->node_box[i * 6];
        double r[4];
        if (!proj.box(nb, nb + 3, r)) return false;
        if (sc->nodes[i].kind != RT_SHAPE_MESH) { occ.mark(r); continue; }
        const DMesh& m = sc->meshes[(size_t)sc->nodes[i].index];
        const FastNode* nodes = sc->fast_nodes.data() + m.fast_off;
        // one inner node: its children are marked (leaves, and boxes at most 16 pixels wide) or handed back to be opened
        auto open_node = [&](int32_t at, std::vector<int32_t>& todo) -> bool {
            const FastNode& n = nodes[at];
            for (int k = 0; k < 2; k++) {
                if (n.child[k] == kFastEmpty) continue;
                double lo[3], hi[3], rr[4];
                for (int a = 0; a < 3; a++) { lo[a] = (double)n.b[k][a][0] + m.center[a]; hi[a] = (double)n.b[k][a][1] + m.center[a]; }
                if (!proj.box(lo, hi, rr)) return false;
                if (n.child[k] < 0 || occ.small(rr)) occ.mark(rr);
                else todo.push_back(n.child[k]);
            }
            return true;
        };
        // breadth first until there are a few hundred subtrees, then those side by side on the host's threads (the map
        // of a 1.3 M-face scan took 20 ms on one thread, a tenth of a cold 4K frame, while the device waited)
        stack.assign(1, 0);
        size_t head = 0;
        while (head < stack.size() && stack.size() - head < 256)
            if (!open_node(stack[head++], stack)) return false;
        std::atomic<bool> ok{true};
        parallel_chunks(head, stack.size(), stack.size() - head >= 64 ? host_threads() : 1, [&](int, uint64_t a, uint64_t e) {
            std::vector<int32_t> mine(stack.begin() + (ptrdiff_t)a, stack.begin() + (ptrdiff_t)e);
            while (!mine.empty() && ok.load(std::memory_order_relaxed)) {
                const int32_t at = mine.back();
                mine.pop_back();
                if (!open_node(at, mine)) ok.store(false, std::memory_order_relaxed);
            }
        });
        if (!ok.load()) return false;
    }
    return true;
}

struct EventPair {
    hipEvent_t a = nullptr, b = nullptr;
};

struct EventPool {  // HIP-event timing of the traversal launches, on the stream they run on
    std::vector<EventPair> events;
    size_t used = 0;
    double ms = 0;
    ~EventPool() {
        for (auto& ev : events) { (void)hipEventDestroy(ev.a); (void)hipEventDestroy(ev.b); }
    }
    hipError_t next(EventPair** out) {
        if (used == events.size()) {
            EventPair ev;
            hipError_t e = hipEventCreate(&ev.a);
            if (e == hipSuccess) e = hipEventCreate(&ev.b);
            if (e != hipSuccess) return e;
            events.push_back(ev);
        }
        *out = &events[used++];
        return hipSuccess;
    }
    hipError_t drain() {  // call only after the stream has been synchronised
        for (size_t i = 0; i < used; i++) {
            float t = 0;
            hipError_t e = hipEventElapsedTime(&t, events[i].a, events[i].b);
            if (e != hipSuccess) return e;
            ms += t;
        }
        used = 0;
        return hipSuccess;
    }
};

// Turns the pilot pass's per-tile costs into the work order of the 8 queues.
// Where the tiles of a pass lie in the image, and the map of the blocks that can see something of the scene (null: no
// such map, every tile is traced).
struct TileGeometry {
    int32_t dx, dy, groups_x, local_rows, band_rank, band_world;
    const Occupancy* occ;
    // Live pixels of tile t (k_primary's numbering) and whether none of them can see anything.
    bool outside(int64_t t, uint32_t& live_pixels) const {
        const int64_t group = t >> 6;
        const int w = (int)(t & 63);
        const int tile_x = ((int)(group % groups_x) * 8 + (w & 7)) * 8;
        const int tile_y = ((int)(group / groups_x) * 8 + (w >> 3)) * 8;
        const int row0 = ((tile_y >> 4) * band_world + band_rank) * 16 + (tile_y & 15);  // a tile lies inside one band of 16 rows
        const int cols = std::max(0, std::min(8, dx - tile_x));
        const int rows = std::max(0, std::min(8, std::min(local_rows - tile_y, dy - row0)));
        live_pixels = (uint32_t)(cols * rows);
        if (live_pixels == 0 || !occ) return false;  // outside the frame: the kernel skips it anyway
        return occ->seen[(size_t)(row0 >> 3) * occ->bw + (size_t)(tile_x >> 3)] == 0;  // tile_x and row0 are multiples of 8
    }
};

int build_tile_order(WorkBuffers& w, int64_t tiles_total, const TileGeometry& geo, hipStream_t stream) {
    std::vector<uint32_t> cost((size_t)tiles_total);
    HIP_TRY(hipMemcpyAsync(cost.data(), w.tile_cost, cost.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    w.culled_pixels = 0;
    std::vector<int32_t> order;
    order.reserve((size_t)tiles_total);
    double sum = 0;
    size_t busy = 0;
    for (uint32_t c : cost)
        if (c) { sum += c; busy++; }
    const double mean = busy ? sum / (double)busy : 1.0;
    // four classes only: the few tiles that cost several times the mean go first (they would otherwise end the
    // launch with most of the chip idle) and the cheap ones last (k_beam takes those up when the others of all queues
    // are gone: the waves then end within a cheap batch's time of each other); everything else keeps its spatial order
    // and with it its L2 locality
    // (Cheap: under half the mean, and so are the eight tiles around it.  A tile at the mesh's outline costs the pilot's
    // packets little, few of its pixels seeing anything, but those that do see faces edge-on: the dearest pixels of k_beam.)
    const int tw = geo.groups_x * 8;
    const int th = (int)(((tiles_total + 63) / 64 + geo.groups_x - 1) / geo.groups_x) * 8;
    std::vector<uint8_t> low((size_t)tw * (size_t)th, 1);
    auto place = [&](int64_t t) {
        const int64_t group = t >> 6;
        const int w8 = (int)(t & 63);
        return (size_t)((int)(group / geo.groups_x) * 8 + (w8 >> 3)) * (size_t)tw + (size_t)((int)(group % geo.groups_x) * 8 + (w8 & 7));
    };
    for (int64_t t = 0; t < tiles_total; t++) low[place(t)] = (double)cost[(size_t)t] < 0.5 * mean;
    auto klass = [&](int32_t t) {
        const double c = (double)cost[(size_t)t];
        if (c >= 2.0 * mean) return c >= 4.0 * mean ? 2 : 1;
        const size_t at = place(t);
        const int x = (int)(at % (size_t)tw), y = (int)(at / (size_t)tw);
        for (int yy = std::max(0, y - 1); yy <= std::min(th - 1, y + 1); yy++)
            for (int xx = std::max(0, x - 1); xx <= std::min(tw - 1, x + 1); xx++)
                if (!low[(size_t)yy * (size_t)tw + (size_t)xx]) return 0;
        return -1;
    };
    for (int k = 0; k < 8; k++) {
        const size_t begin = order.size();
        for (int64_t g = k; g * 64 < tiles_total; g += 8)
            for (int t = 0; t < 64 && g * 64 + t < tiles_total; t++) {
                // A tile no ray of which can hit anything is left out; the host adds its rays to the statistics.
                uint32_t live = 0;
                if (geo.outside(g * 64 + t, live)) { w.culled_pixels += live; continue; }
                order.push_back((int32_t)(g * 64 + t));
            }
        std::stable_sort(order.begin() + begin, order.end(), [&](int32_t a, int32_t b) { return klass(a) > klass(b); });
        w.queue_off[k] = (uint32_t)begin;
        w.queue_len[k] = (uint32_t)(order.size() - begin);
        w.queue_cheap[k] = (uint32_t)std::count_if(order.begin() + begin, order.end(), [&](int32_t t) { return klass(t) < 0; });
    }
    if (std::getenv("RT_DEBUG")) {
        std::vector<uint32_t> sorted(cost);
        std::sort(sorted.begin(), sorted.end());
        size_t n_class[4] = {0, 0, 0, 0};
        for (int32_t t : order) n_class[klass(t) + 1]++;
        auto at = [&](double f) { return sorted[(size_t)(f * (double)(sorted.size() - 1))]; };
        fprintf(stderr, "[rt] tile schedule: %zu tiles, pilot cost mean %.0f (of the %zu that cost anything), percentiles 10/25/50/75/90/99: %u %u %u %u %u %u; classes cheap/plain/dear/dearest: %zu %zu %zu %zu\n",
                order.size(), mean, busy, at(0.1), at(0.25), at(0.5), at(0.75), at(0.9), at(0.99), n_class[0], n_class[1], n_class[2], n_class[3]);
    }
    HIP_TRY(hipMemcpyAsync(w.tile_order, order.data(), order.size() * sizeof(int32_t), hipMemcpyHostToDevice, stream));
    HIP_TRY(hipStreamSynchronize(stream));  // `order` goes out of scope
    return RT_OK;
}

}  // namespace

// The static scratch of every device dressed up as WorkBuffers.  Only scenes whose materials all just emit render from
// it (they queue nothing), so the queue views stay empty; the schedule arrays are the scratch's own.
WorkBuffers* static_work_for(int device, int64_t tiles_total) {
    static WorkBuffers table[kStaticScratchDevices];
    if (device < 0 || device >= kStaticScratchDevices) {
        fail(RT_ERR_UNSUPPORTED, "device ordinal beyond the static scratch table");
        return nullptr;
    }
    if (tiles_total > kStaticTiles) {
        fail(RT_ERR_UNSUPPORTED, "frame too large for the static scratch: " + std::to_string(tiles_total) + " tiles of 8x8 pixels, at most " +
                                     std::to_string(kStaticTiles) + " (4096 x 4096 pixels)");
        return nullptr;
    }
    static std::mutex table_mutex;  // two threads may make their first call on a device at the same time
    std::lock_guard<std::mutex> lock(table_mutex);
    WorkBuffers& w = table[device];
    if (!w.is_static) {
        void* base = nullptr;
        const hipError_t e = static_scratch(device, &base);
        if (e != hipSuccess) {
            fail(RT_ERR_DEVICE, std::string("static scratch: ") + hipGetErrorString(e));
            return nullptr;
        }
        char* p = static_cast<char*>(base);
        w.q = 64 * 64;
        w.primary_waves = 64;
        w.queues.counters = reinterpret_cast<Counters*>(p);
        w.d_camera = reinterpret_cast<DCamera*>(p + kStaticOffCamera);
        w.d_queues = reinterpret_cast<Queues*>(p + kStaticOffQueues);
        w.tile_cost = reinterpret_cast<uint32_t*>(p + kStaticOffCost);
        w.tile_order = reinterpret_cast<int32_t*>(p + kStaticOffOrder);
        w.order_tiles = (size_t)kStaticTiles;
        w.is_static = true;
    }
    return &w;
}

int scene_film(const rt_scene* sc, Replica& rep, int dx, int dy, double** film, uint8_t** rgba, uint8_t** rgba_host, int slot) {
    if (int rc = ensure_work(sc, rep, 0)) return rc;  // the render itself asks for the queues its frame needs
    WorkBuffers::Frame& w = rep.work->frame[slot & 1];
    const size_t px = (size_t)dx * dy;
    if (w.rgba_bytes < px * 4) {
        if (w.rgba) (void)hipFree(w.rgba);
        if (w.rgba_host) (void)hipHostFree(w.rgba_host);
        w.rgba = nullptr;
        w.rgba_host = nullptr;
        w.rgba_bytes = 0;
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&w.rgba), px * 4));
        HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&w.rgba_host), px * 4, hipHostMallocDefault));
        w.rgba_bytes = px * 4;
    }
    if (w.film_doubles < px * 3) {
        if (w.film) (void)hipFree(w.film);
        w.film = nullptr;
        w.film_doubles = 0;
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&w.film), px * 3 * sizeof(double)));
        w.film_doubles = px * 3;
    }
    *film = w.film;
    *rgba = w.rgba;
    *rgba_host = w.rgba_host;
    return RT_OK;
}

int render_into_film(rt_scene* sc, Replica& rep, const rt_camera_desc* camera, const rt_render_opts* o, double* film, rt_stats* stats,
                     WorkBuffers* static_work) {
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
    HIP_TRY(hipSetDevice(rep.device));
    DCamera cam;
    if (int rc = make_camera(*camera, cam)) return rc;
    int rows_here = 0;
    for (int r = 0; r < o->dy; r++)
        if ((r / 16) % world == rank) rows_here++;
    if (static_work) {
        if (!sc->all_terminal)
            return fail(RT_ERR_UNSUPPORTED, "this scene scatters rays: its ray queues do not fit the static scratch (only scenes whose materials all just emit do)");
    } else if (int rc = ensure_work(sc, rep, (uint64_t)o->dx * (uint64_t)rows_here * (uint64_t)(o->sample_count > 0 ? o->sample_count : o->rays_per_pixel))) {
        return rc;
    }
    WorkBuffers& w = static_work ? *static_work : *rep.work;
    Queues& q = w.queues;
    hipStream_t stream = reinterpret_cast<hipStream_t>(o->stream);
    if (static_work) HIP_TRY(hipMemcpyAsync(w.d_queues, &w.queues, sizeof(Queues), hipMemcpyHostToDevice, stream));
    // rt_cancel's flag belongs to the API call (cleared where rt_render / rt_render_film / rt_render_frames begin): it is only
    // read here -- before every launch and once more when the last launch has ended -- so a frame that is a single launch,
    // and the other devices' shares of a frame, see it too.
    HIP_TRY(hipMemcpyAsync(w.d_camera, &cam, sizeof(DCamera), hipMemcpyHostToDevice, stream));  // pageable source: staged before the call returns

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
    pass.groups_x = (o->dx + 63) / 64;
    pass.spp = o->rays_per_pixel;
    pass.seed_lo = (uint32_t)((uint64_t)o->seed & 0xffffffffu);
    pass.seed_hi = (uint32_t)((uint64_t)o->seed >> 32);
    pass.max_depth = o->max_ray_depth;
    const bool want_cull = std::getenv("RT_NO_PIXEL_CULL") == nullptr;
    const int64_t tiles_total = (int64_t)pass.groups_x * ((local_rows + 63) / 64) * 64;  // whole groups; tiles outside the image idle
    const int s_first = o->sample_count > 0 ? o->sample_begin : 0;
    const int s_end = o->sample_count > 0 ? o->sample_begin + o->sample_count : o->rays_per_pixel;
    // paths per primary launch: bounded by the queues (two children and one surface point per path) unless nothing is ever queued
    const int64_t paths_per_launch = sc->all_terminal ? ((int64_t)1 << 40) : (int64_t)w.q;

    HIP_TRY(hipMemsetAsync(q.counters, 0, sizeof(Counters), stream));
    EventPool events;
    const bool timed = static_work == nullptr;  // creating events allocates: the static path goes without launch timing
    uint64_t traverse_launches = 0, culled_rays = 0;
    Counters host{};
    uint64_t pool_top = 0;
    uint64_t launched_rays = 0, shadow_rays = 0;  // what the host can count itself is not counted on the device

    auto read_counters = [&]() -> int {
        HIP_TRY(hipMemcpyAsync(&host, q.counters, sizeof(Counters), hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        HIP_TRY(events.drain());
        if (host.errors.v != 0)
            return fail(RT_ERR_DEVICE, "internal error counted by a kernel (" + std::to_string(host.errors.v) + "): a traversal stack or a hit queue ran full");
        return RT_OK;
    };
    auto reset_cursors = [&]() -> int {  // zero every append, claim and tile cursor, keep the statistics
        HIP_TRY(hipMemsetAsync(&q.counters->ray_cursor[0][0], 0, sizeof(Counters) - offsetof(Counters, ray_cursor), stream));
        return RT_OK;
    };
    auto reset_for_trace = [&](int out) -> int {  // before a launch that traces one ray queue and appends to the other (`out`)
        HIP_TRY(hipMemsetAsync(&q.counters->ray_cursor[out][0], 0, sizeof(q.counters->ray_cursor[out]), stream));
        HIP_TRY(hipMemsetAsync(&q.counters->surf_cursor[0], 0, sizeof(Counters) - offsetof(Counters, surf_cursor), stream));
        return RT_OK;
    };

    // Cost-ordered schedule: worth a pilot pass when the frame has many tiles and more than a few samples.
    const std::string order_key = std::string(reinterpret_cast<const char*>(camera), sizeof(*camera)) + "|" + std::to_string(o->dx) + "x" +
                                  std::to_string(o->dy) + "|" + std::to_string(rank) + "/" + std::to_string(world) + "|" + std::to_string(o->mode) +
                                  (want_cull ? "|c" : "|a") + "|" + std::to_string(sc->serial);
    const bool want_order = tiles_total >= 2048 && std::getenv("RT_NO_TILE_ORDER") == nullptr;
    if (want_order && !w.is_static && (w.order_tiles != (size_t)tiles_total || !w.tile_cost)) {
        if (w.tile_cost) (void)hipFree(w.tile_cost);
        if (w.tile_order) (void)hipFree(w.tile_order);
        w.tile_cost = nullptr;
        w.tile_order = nullptr;
        w.order_valid = false;
        w.order_tiles = 0;
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&w.tile_cost), (size_t)tiles_total * sizeof(uint32_t)));
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&w.tile_order), (size_t)tiles_total * sizeof(int32_t)));
        w.order_tiles = (size_t)tiles_total;
    }
    if (w.order_valid && w.order_key != order_key) w.order_valid = false;
    constexpr int kPilotSamples = 2;
    // Samples per pass: as many as a launch can take.  What a launch costs apart from its samples -- with pixel beams one tree walk
    // per pixel, and the ramp and tail of the persistent grid -- is 5 ms on a 4K frame and 3 ms on an eighth of it, so a 1024-sample
    // frame in one launch instead of four runs 8 % faster on one device and a third faster on its share among eight.  (Scenes
    // that queue rays are bounded by their queues below; cancellation is looked at between launches, here every 0.2 s at most.)
    int pass_samples = 1024;
    if (const char* e = getenv("RT_PASS_SAMPLES")) pass_samples = std::max(1, atoi(e));
    int lanes_override = -1;   // RT_LOG2_LANES / RT_ITEM_RAYS: tuning knobs of the primary launch
    int64_t item_rays = 256;   // measured 64 / 128 / 256 / 512, Grays/s: scan-lit 5.54 / 5.53 / 5.44 / 5.20, material 7.31 / 7.30 / 7.28 / 7.18, teapot 5.43 / 5.50 / 5.54 / 5.56, charuco orbit 610 / 670 / 681 / 662 frames/s
    if (const char* e = getenv("RT_LOG2_LANES")) lanes_override = std::max(0, atoi(e));
    // Pixel beams (one tree walk per pixel instead of one per 64 samples, see dev_bvh.h): parallel rays through one mesh.
    // Orthographic: the rays share a direction.  Pinhole, or focusable with a lens of radius 0 (camera_focusable.go:26-63: the
    // lens offset is then zero, whatever is drawn): they share their origin, and the image plane must not pass through it.
    bool beam_pyr = false;
    double beam_key[9] = {cam.dir[0], cam.dir[1], cam.dir[2], cam.u[0], cam.u[1], cam.u[2], cam.v[0], cam.v[1], cam.v[2]};
    bool beam_cam = cam.kind == RT_CAM_ORTHOGRAPHIC;
    if (cam.kind == RT_CAM_PINHOLE || (cam.kind == RT_CAM_FOCUSABLE && cam.lens_radius == 0)) {
        const double* org = cam.kind == RT_CAM_PINHOLE ? cam.origin : cam.from;
        const double n[3] = {cam.horizontal[1] * cam.vertical[2] - cam.horizontal[2] * cam.vertical[1], cam.horizontal[2] * cam.vertical[0] - cam.horizontal[0] * cam.vertical[2],
                             cam.horizontal[0] * cam.vertical[1] - cam.horizontal[1] * cam.vertical[0]};
        const double nn = std::sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
        const double away = ((cam.llc[0] - org[0]) * n[0] + (cam.llc[1] - org[1]) * n[1] + (cam.llc[2] - org[2]) * n[2]) / nn;
        const double extent = std::sqrt(cam.horizontal[0] * cam.horizontal[0] + cam.horizontal[1] * cam.horizontal[1] + cam.horizontal[2] * cam.horizontal[2]) +
                              std::sqrt(cam.vertical[0] * cam.vertical[0] + cam.vertical[1] * cam.vertical[1] + cam.vertical[2] * cam.vertical[2]);
        beam_cam = beam_pyr = std::isfinite(away) && std::isfinite(extent) && std::fabs(away) > 1e-9 * extent;
        for (int k = 0; k < 3; k++) beam_key[k] = org[k];
        for (int k = 3; k < 9; k++) beam_key[k] = 0;
    }
    const bool beam = o->mode == RT_MODE_FAST && sc->mesh_only && sc->all_terminal && rep.dev.n_nodes == 1 && beam_cam && rep.beam_faces > 0 &&
                      o->rays_per_pixel > 1 && !getenv("RT_NO_BEAM");
    if (beam) {
        // the tables belong to a direction (every ray of an orthographic frame has the camera's, camera_orthographic.go:44) or to a point
        if (!rep.beam_key_valid || (rep.beam.pyramid != 0) != beam_pyr || std::memcmp(rep.beam_key, beam_key, sizeof(rep.beam_key)) != 0) {
            launch_beam_prepare(rep.dev, rep.beam_faces, beam_pyr, beam_key, const_cast<BeamFace32*>(rep.beam.f32), const_cast<BeamFace64*>(rep.beam.f64),
                                const_cast<double*>(rep.beam.uv), stream);
            HIP_TRY(hipGetLastError());
            std::memcpy(rep.beam_key, beam_key, sizeof(rep.beam_key));
            rep.beam_key_valid = true;
            rep.beam.pyramid = beam_pyr ? 1 : 0;
        }
        // Finite constants far from the float64 range cannot produce a NaN or an infinity in a ray (orthographic: origins
        // from + u a + v b with |a| <= |FOVWidth| / 2, |b| <= |FOVHeight| / 2; from a point: unit(llc + s h + t v - o) with a
        // non-zero argument), so the guard of render.go:147-150 never fires and is left out
        bool safe = true;
        if (beam_pyr) {
            for (int k = 0; k < 3; k++)
                for (double c : {cam.llc[k], cam.horizontal[k], cam.vertical[k], beam_key[k]}) safe = safe && std::isfinite(c) && std::fabs(c) < 1e150;
        } else {
            for (int k = 0; k < 3; k++)
                for (double c : {cam.from[k], cam.u[k], cam.v[k], cam.dir[k]}) safe = safe && std::isfinite(c) && std::fabs(c) < 1e150;
            safe = safe && std::isfinite(cam.fov_w) && std::isfinite(cam.fov_h) && std::fabs(cam.fov_w) < 1e150 && std::fabs(cam.fov_h) < 1e150;
        }
        rep.beam.safe_camera = safe ? 1 : 0;
    }
    const bool item_rays_fixed = getenv("RT_ITEM_RAYS") != nullptr;
    if (const char* e = getenv("RT_ITEM_RAYS")) item_rays = std::max(1, atoi(e));
    for (int s0 = s_first; s0 < s_end;) {
        // a pass covers all tiles for a run of samples when that fits, else a run of tiles for one sample
        int s_count = 1;
        if (tiles_total * 64 <= paths_per_launch)
            s_count = (int)std::min<int64_t>(std::min(s_end - s0, pass_samples), paths_per_launch / std::max<int64_t>(tiles_total * 64, 1));
        int log2_s = 0;
        int64_t tiles_per_launch = 1;
        auto plan = [&](int samples) {
            // lanes per pixel: the widest power of two that keeps >= 90 % of the lanes busy over the rounds
            log2_s = 0;
            for (int l = 6; l > 0; l--) {
                const int lanes = 1 << l, rounds = (samples + lanes - 1) / lanes;
                if (samples * 10 >= rounds * lanes * 9) { log2_s = l; break; }
            }
            if (lanes_override >= 0) log2_s = std::min(lanes_override, 6);
            // a tile takes 2^log2_s * ceil(samples / 2^log2_s) rounds of a wave; the queue stripes are sized for launches
            // of at most q / 64 * 10 / 9 rounds (ensure_work), whatever the lanes are
            const int64_t rounds_per_tile = ((int64_t)1 << log2_s) * ((samples + (1 << log2_s) - 1) >> log2_s);
            tiles_per_launch = std::max<int64_t>(1, paths_per_launch / (64 * (int64_t)samples));
            if (!sc->all_terminal) tiles_per_launch = std::max<int64_t>(1, std::min(tiles_per_launch, ((int64_t)w.q + 63) / 64 * 10 / 9 / rounds_per_tile));
        };
        // A pilot: a short pass over the WHOLE frame in one launch that records what every tile costs.  Fewer than 32
        // samples to go do not repay it, and a frame that takes several launches per pass never runs from the schedule.
        bool pilot = false;
        if (want_order && !w.order_valid && tiles_total * 64 <= paths_per_launch && s_end - s0 >= 32) {
            plan(std::min(s_count, kPilotSamples));
            pilot = tiles_per_launch >= tiles_total;
        }
        if (pilot) s_count = std::min(s_count, kPilotSamples);
        else plan(s_count);
        for (int64_t t0 = 0; t0 < tiles_total; t0 += tiles_per_launch) {
            if (sc->cancel.load()) {
                (void)hipStreamSynchronize(stream);
                return fail(RT_ERR_CANCELLED, "render cancelled");
            }
            const int64_t nt = std::min(tiles_per_launch, tiles_total - t0);
            pass.sample_begin = s0;
            pass.sample_count = s_count;
            {
                pass.log2_s = log2_s;
                // pixels per item: about item_rays rays (smaller items shorten the tail of a launch), at least what the wave traces at a time, at most the tile
                int block = 64 >> log2_s;
                int64_t want_rays = item_rays;
                if (beam && !item_rays_fixed) {
                    // Pixel beams: a lane walks the tree for each pixel of an item, so more pixels keep more lanes busy in that
                    // phase (kBeamPixels at most); against that, the fewer items a resident wave gets the longer the chip waits for
                    // the last ones.  (The pixels that take milliseconds each become items of their own whatever this says: k_beam's
                    // queue of long pixels.)  Measured with k_beam at 1024 samples, ms per launch with items of
                    // 1 / 2 / 4 pixels: the 4K scan frame 97.2 / 81.2 / 73.3, a quarter of it 24.7 / 20.6 / 18.8, an eighth 12.8 / 10.7 / 9.7;
                    // the same scene at 1920x1080 29.3 / 25.5 / 23.8, a quarter of that 7.7 / 6.8 / 6.8, an eighth 5.6 / 5.1 / 5.8.
                    const int64_t share_px = tiles_total * 64;
                    int64_t px = share_px >= 400000 ? 4 : (share_px >= 50000 ? 2 : 1);
                    px = std::min<int64_t>(px, std::max<int64_t>(1, 4096 / s_count));
                    px = std::max<int64_t>(px, std::min<int64_t>(8, 1024 / std::max(1, s_count)));
                    while (px & (px - 1)) px &= px - 1;  // down to a power of two
                    want_rays = px * s_count;
                }
                while (block < 64 && (int64_t)block * s_count < want_rays) block <<= 1;
                if (beam && log2_s == 6) block = std::min(block, beam_pixels_max());
                pass.block_px = block;
            }
            pass.tile_begin = t0;
            pass.n_pixels = nt * 64;
            pass.max_waves = sc->all_terminal ? 0 : (int32_t)w.primary_waves;  // nothing is queued when every material only emits
            pass.cull_pixels = want_cull ? 1 : 0;
            const bool whole = t0 == 0 && nt == tiles_total;
            pass.order = (want_order && w.order_valid && whole) ? w.tile_order : nullptr;
            pass.cost = (pilot && whole) ? w.tile_cost : nullptr;
            for (int k = 0; k < 8; k++) { pass.queue_off[k] = w.queue_off[k]; pass.queue_len[k] = w.queue_len[k]; pass.queue_cheap[k] = w.queue_cheap[k]; }
            if (pass.cost) HIP_TRY(hipMemsetAsync(w.tile_cost, 0, (size_t)tiles_total * sizeof(uint32_t), stream));
            if (int rc = reset_cursors()) return rc;
            EventPair* ev = nullptr;
            if (timed) {
                HIP_TRY(events.next(&ev));
                HIP_TRY(hipEventRecord(ev->a, stream));
            }
            if (!(beam && launch_beam(rep.dev, w.d_camera, pass, q.counters, rep.beam, film, stream)))
                launch_primary(rep.dev, w.d_camera, pass, w.d_queues, o->mode, sc->mesh_only, !sc->all_terminal, film, stream);
            if (pass.order) culled_rays += w.culled_pixels * (uint64_t)s_count;  // tiles the schedule leaves out: nothing to trace
            if (timed) HIP_TRY(hipEventRecord(ev->b, stream));
            HIP_TRY(hipGetLastError());
            traverse_launches++;
            if (sc->all_terminal) {
                if (events.used >= 256) {  // bound the number of live events without stalling every pass
                    HIP_TRY(hipStreamSynchronize(stream));
                    HIP_TRY(events.drain());
                }
                continue;
            }
            // generations of scattered rays: the launch just queued left its children in rays[cur]
            int cur = 1;
            for (;;) {
                if (sc->cancel.load()) {
                    (void)hipStreamSynchronize(stream);
                    return fail(RT_ERR_CANCELLED, "render cancelled");
                }
                if (int rc = read_counters()) return rc;
                RayStripeCounts rays_n{};
                StripeCounts surf_n{};
                uint64_t n_surf = 0;
                for (int k = 0; k < w.ray_stripes; k++) {
                    rays_n.n[k] = host.ray_cursor[cur][k].v;
                    if (rays_n.n[k] > q.ray_stripe_cap) return fail(RT_ERR_DEVICE, "internal queue overflow");
                }

                for (int k = 0; k < kQueueStripes; k++) {
                    surf_n.n[k] = host.surf_cursor[k].v;
                    if (surf_n.n[k] > q.surf_stripe_cap) return fail(RT_ERR_DEVICE, "internal queue overflow");
                    n_surf += surf_n.n[k];
                }
                if (n_surf > 0) {
                    launch_direct(rep.dev, q, surf_n, o->mode, film, stream);
                    shadow_rays += n_surf * (uint64_t)rep.dev.n_lights;  // one occlusion ray per surface point and light
                    HIP_TRY(hipGetLastError());
                }
                // This launch takes at most q rays in all and stripe_rays of a stripe; the rest is parked on the overflow
                // stack (it grows on demand: a generation parks at most 2 q rays, and at most MaxRayDepth + 1 generations
                // are parked at a time) and comes back when a generation has died out.
                uint64_t n = 0;
                for (int k = 0; k < w.ray_stripes; k++) {
                    const uint32_t keep = (uint32_t)std::min<uint64_t>(std::min(rays_n.n[k], w.stripe_rays), (uint64_t)w.q - n);
                    const uint64_t excess = rays_n.n[k] - keep;
                    if (excess) {
                        if (pool_top + excess > w.pool_capacity) {  // grow: a new stack, the parked rays moved over
                            const uint64_t want = std::max<uint64_t>(std::max<uint64_t>(2 * w.pool_capacity, pool_top + excess), 2 * (uint64_t)w.q);
                            void* bigger = nullptr;
                            hipError_t pe = hipMalloc(&bigger, want * kRayBytes);
                            if (pe != hipSuccess) return fail(RT_ERR_NOMEM, std::string("overflow stack of ") + std::to_string(want) + " rays: " + hipGetErrorString(pe));
                            RayQ grown{};
                            carve(grown, bigger, want);
                            if (pool_top) pe = copy_rays(grown, 0, w.pool, 0, pool_top, stream);
                            if (pe == hipSuccess) pe = hipStreamSynchronize(stream);
                            if (pe != hipSuccess) { (void)hipFree(bigger); return fail(RT_ERR_DEVICE, hipGetErrorString(pe)); }
                            if (w.pool_base) (void)hipFree(w.pool_base);
                            w.pool_base = bigger;
                            w.pool = grown;
                            w.pool_capacity = want;
                        }
                        HIP_TRY(copy_rays(w.pool, pool_top, q.rays[cur], (uint64_t)k * q.ray_stripe_cap + keep, excess, stream));
                        pool_top += excess;
                    }
                    rays_n.n[k] = keep;
                    n += keep;
                }
                if (n == 0) {
                    if (pool_top == 0) break;
                    for (int k = 0; k < w.ray_stripes && pool_top > 0 && n < w.q; k++) {  // deal the parked rays to the stripes (their class is not kept: it only packs the waves)
                        const uint64_t take = std::min<uint64_t>(std::min<uint64_t>(pool_top, w.stripe_rays), (uint64_t)w.q - n);
                        HIP_TRY(copy_rays(q.rays[cur], (uint64_t)k * q.ray_stripe_cap, w.pool, pool_top - take, take, stream));
                        pool_top -= take;
                        rays_n.n[k] = (uint32_t)take;
                        n += take;
                    }
                }
                if (int rc = reset_for_trace(1 - cur)) return rc;
                HIP_TRY(events.next(&ev));
                HIP_TRY(hipEventRecord(ev->a, stream));
                launch_trace(rep.dev, pass, q, cur, rays_n, o->mode, sc->mesh_only, film, stream);
                launched_rays += n;
#ifdef RT_VISIT_STATS
                if (std::getenv("RT_DEBUG")) {
                    HIP_TRY(hipStreamSynchronize(stream));
                    unsigned long long ts[64];
                    debug_trace_stats(ts, true);
                    const double wv = ts[48] ? (double)ts[48] : 1.0;
                    fprintf(stderr, "[rt] k_walk: %llu rays, %llu waves; per wave: %.0f cycles (refill %.0f, walk %.0f, output %.0f), %.1f loop iterations, %.1f rounds at %.1f lanes, %.1f rays fetched\n",
                            (unsigned long long)n, ts[48], ts[49] / wv, ts[50] / wv, ts[51] / wv, ts[52] / wv, ts[55] / wv, ts[53] / wv, ts[53] ? (double)ts[54] / ts[53] : 0.0, ts[56] / wv);
                }
#endif
                HIP_TRY(hipEventRecord(ev->b, stream));
                HIP_TRY(hipGetLastError());
                traverse_launches++;
                cur = 1 - cur;
            }
        }
        if (pilot) {
            Occupancy occ;
            TileGeometry geo{o->dx, o->dy, pass.groups_x, pass.local_rows, pass.band_rank, pass.band_world, nullptr};
            const auto t_occ = std::chrono::steady_clock::now();
            if (want_cull && build_occupancy(sc, cam, o->dx, o->dy, occ)) geo.occ = &occ;
            if (std::getenv("RT_DEBUG"))
                fprintf(stderr, "[rt] occupancy map: %s, %.2f ms\n", geo.occ ? "built" : "none",
                        1e3 * std::chrono::duration<double>(std::chrono::steady_clock::now() - t_occ).count());
            if (int rc = build_tile_order(w, tiles_total, geo, stream)) return rc;
            w.order_key = order_key;
            w.order_valid = true;
        }
        s0 += s_count;
    }
    if (int rc = read_counters()) return rc;
    if (sc->cancel.load()) return fail(RT_ERR_CANCELLED, "render cancelled");  // the film holds a frame nobody waits for any more
#ifdef RT_WAVE_TIMES
    {
        {  // when the persistent waves of the last k_beam launch began and ended
            std::vector<unsigned long long> wt(4 * 8192);
            debug_wave_times(wt.data(), true);
            unsigned long long t0 = ~0ull, t1 = 0;
            size_t n = 0;
            for (size_t i = 0; i < 8192; i++)
                if (wt[4 * i + 1]) { t0 = std::min(t0, wt[4 * i]); t1 = std::max(t1, wt[4 * i + 1]); n++; }
            if (n) {
                std::vector<double> ends, starts;
                unsigned long long items_min = ~0ull, items_max = 0;
                for (size_t i = 0; i < 8192; i++)
                    if (wt[4 * i + 1]) {
                        ends.push_back((double)(wt[4 * i + 1] - t0) * 1e-5);   // ms at 100 MHz
                        starts.push_back((double)(wt[4 * i] - t0) * 1e-5);
                        items_min = std::min(items_min, wt[4 * i + 2] & 0xffff); items_max = std::max(items_max, wt[4 * i + 2] & 0xffff);
                    }
                std::sort(ends.begin(), ends.end());
                std::sort(starts.begin(), starts.end());
                auto q = [&](const std::vector<double>& v, double f) { return v[(size_t)(f * (double)(v.size() - 1))]; };
                size_t late = 0;
                for (double t : starts) late += t > 0.5 * ends.back();
                fprintf(stderr, "[rt] k_beam waves %zu (%zu of them started in the second half of the launch: not resident before): start (ms) median %.3f last %.3f; end (ms) first %.3f 10%% %.3f median %.3f 90%% %.3f 99%% %.3f last %.3f; items per wave %llu..%llu\n",
                        n, late, q(starts, 0.5), starts.back(), ends.front(), q(ends, 0.1), q(ends, 0.5), q(ends, 0.9), q(ends, 0.99), ends.back(), items_min, items_max);
                std::vector<std::tuple<double, unsigned long long, unsigned long long>> lastw;
                for (size_t i = 0; i < 8192; i++)
                    if (wt[4 * i + 1]) lastw.push_back({(double)(wt[4 * i + 1] - t0) * 1e-5, wt[4 * i + 2], wt[4 * i + 3]});
                std::sort(lastw.begin(), lastw.end());
                fprintf(stderr, "[rt] k_beam waves, the last to end (end ms / their last item's ms / the one before's; L = a long pixel):");
                for (size_t i = lastw.size() > 24 ? lastw.size() - 24 : 0; i < lastw.size(); i++)
                    {
                    const unsigned long long px = std::get<2>(lastw[i]) >> 24, tl = px >> 6, grp = tl >> 6;
                    fprintf(stderr, " %.3f/%.3f%s/%.3f@%d,%d+%d(cost %u)", std::get<0>(lastw[i]), (double)((std::get<1>(lastw[i]) >> 16) & 0xffffffffull) * 1e-5, ((std::get<1>(lastw[i]) >> 48) & 1) ? "L" : "",
                            (double)(std::get<2>(lastw[i]) & 0xffffffull) * 1e-5, (int)(grp % (unsigned long long)pass.groups_x) * 8 + (int)(tl & 7), (int)(grp / (unsigned long long)pass.groups_x) * 8 + (int)((tl >> 3) & 7), (int)(px & 63), 0u);
                }
                fprintf(stderr, "\n");
            }
        }
    }
#endif
#ifdef RT_VISIT_STATS
    {
        unsigned long long ps[16];
        debug_packet_stats(ps, true);
        const double w = ps[0] ? (double)ps[0] : 1.0, rounds = ps[10] ? (double)ps[10] : 1.0;
        fprintf(stderr, "[rt] packet walks %llu: per walk %.2f inner nodes, %.2f leaves, %.2f float32 face tests, %.2f float64 face tests, %.2f pops, %.2f pushes\n",
                ps[0], ps[1] / w, ps[2] / w, ps[3] / w, ps[4] / w, ps[5] / w, ps[6] / w);
        if (getenv("RT_DEBUG_BEAM"))
            fprintf(stderr, "[rt] pixel beams: %llu pixels list faces; %llu of them (%.3f %%) have a list longer than one instalment and take %.2f %% of the cycles of all rounds, "
                            "%llu (%.3f %%) have more survivors than the compact array holds and take %.2f %%; per round %.2f leaves looked at, %.2f float32 face tests, %.2f float64\n",
                    ps[0], ps[5], 100.0 * ps[5] / w, ps[9] ? 100.0 * ps[8] / ps[9] : 0.0, ps[6], 100.0 * ps[6] / w, ps[9] ? 100.0 * ps[7] / ps[9] : 0.0,
                    ps[2] / rounds, ps[3] / rounds, ps[4] / rounds);
        if (getenv("RT_DEBUG_BEAM") && ps[15])
            fprintf(stderr, "[rt] complete lists %llu: %.2f leaves with %.2f faces each, of which %.2f survive the bundle's test\n", ps[15], (double)ps[1] / ps[15],
                    (double)ps[11] / ps[15], (double)ps[14] / ps[15]);
        fprintf(stderr, "[rt] wave rounds %llu: %.0f cycles each on average; walks %.0f cycles each (%.0f of them in leaves), %.1f %% of all round cycles\n",
                ps[10], ps[9] / rounds, ps[8] / w, ps[11] / w, ps[9] ? 100.0 * ps[8] / ps[9] : 0.0);
        unsigned long long ts[64];
        debug_trace_stats(ts, true);
        auto d = [](unsigned long long a, unsigned long long b) { return b ? (double)a / (double)b : 0.0; };
        fprintf(stderr, "[rt] k_walk inner nodes per ray, rays in [0,2) [2,4) ... [8192,inf): ");
        for (int b = 0; b < 14; b++) fprintf(stderr, "%llu ", ts[32 + b]);
        fprintf(stderr, "; longest %llu\n", ts[47]);
        fprintf(stderr, "[rt] k_trace waves %llu, live lanes %.1f, lanes needing Resolve %.1f; cycles per wave: closest hit %.0f, Resolve + appends %.0f, film %.0f\n",
                ts[8], d(ts[9], ts[8]), d(ts[11], ts[8]), d(ts[10], ts[8]), d(ts[12], ts[8]), d(ts[13], ts[8]));
        fprintf(stderr, "[rt] per-lane closest walks: %llu waves at %.1f lanes; per wave %.1f inner steps at %.1f lanes, %.1f face steps at %.1f lanes\n",
                ts[16], d(ts[17], ts[16]), d(ts[18], ts[16]), d(ts[19], ts[18]), d(ts[20], ts[16]), d(ts[21], ts[20]));
        fprintf(stderr, "[rt] k_walk rounds: %llu at %.1f lanes; per round %.2f descent steps at %.1f lanes, %.2f float32 face steps at %.1f lanes, %.2f float64 face steps at %.1f lanes\n",
                ts[16], d(ts[17], ts[16]), d(ts[18], ts[16]), d(ts[19], ts[18]), d(ts[20], ts[16]), d(ts[21], ts[20]), d(ts[28], ts[16]), d(ts[29], ts[28]));
        fprintf(stderr, "[rt] per-lane any-hit walks: %llu waves at %.1f lanes; per wave %.1f inner steps at %.1f lanes, %.1f face steps at %.1f lanes\n",
                ts[22], d(ts[23], ts[22]), d(ts[24], ts[22]), d(ts[25], ts[24]), d(ts[26], ts[22]), d(ts[27], ts[26]));
        fprintf(stderr, "[rt] share of round cycles: ray generation %.1f %%, closest hit (set-up + walk) %.1f %%, rest (attributes, shading, film) %.1f %%\n",
                100.0 * ps[13] / (double)ps[9], 100.0 * ps[12] / (double)ps[9], 100.0 * ((double)ps[9] - ps[13] - ps[12]) / (double)ps[9]);
    }
#endif
    if (stats) {
        std::memset(stats, 0, sizeof(*stats));
        uint64_t total = launched_rays, exceeded = 0, left = 0, culled_in_kernel = 0;
        for (const Counters::Tally& t : host.tally) { total += t.total; exceeded += t.exceeded; left += t.left; culled_in_kernel += t.culled; }
        // pixels the primary kernel itself settled without generating their rays move from its tallies to the culled ones
        total -= culled_in_kernel;
        left -= culled_in_kernel;
        culled_rays += culled_in_kernel;
        stats->total_rays = total + culled_rays;  // rays of culled tiles and pixels count as traced and as having left the scene
        stats->rays_exceeded_depth = exceeded;
        stats->rays_left_scene = left + culled_rays;
        stats->shadow_rays = shadow_rays;
        stats->render_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
        stats->traverse_seconds = events.ms * 1e-3;
        stats->traverse_launches = traverse_launches;
        stats->traverse_rays = total;  // every ray a kernel generated and traced was settled by exactly one of the timed launches
        stats->culled_rays = culled_rays;
    }
    return RT_OK;
}

// ---- one frame over several devices ----------------------------------------------------------------------------
//
// The reference renders a frame with NumCPU goroutines that pull 16x16 tiles from one channel and write disjoint
// pixels of one image (render.go:297-331).  Here every device of rt_init holds a replica of the scene and renders a
// share of the frame into a film in its own memory, driven by a host thread of its own:
//   rows     band b of 16 rows (the reference's tile height) goes to device b % n; afterwards every device copies its
//            bands to the first device over its own link, where they are added to the caller's film (disjoint rows:
//            the frame is exact, and a film that already holds earlier passes keeps them);
//   samples  device i renders the i-th run of sample indices of EVERY pixel (the random streams are keyed by pixel
//            and sample index, so the samples are the ones a single device would draw); the films are copied to the
//            first device and added there.  For frames with few pixels and many samples per pixel.
// The first device renders straight into the caller's film on the caller's stream.

namespace {

struct Share {
    int rc = RT_OK;
    std::string error;
    rt_stats stats{};
};

int ensure_peer_buffers(Replica& first, Replica& rep, size_t film_doubles, bool want_inbox) {
    if (rep.film_doubles < film_doubles) {
        HIP_TRY(hipSetDevice(rep.device));
        if (rep.film) (void)hipFree(rep.film);
        rep.film = nullptr;
        rep.film_doubles = 0;
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&rep.film), film_doubles * sizeof(double)));
        rep.film_doubles = film_doubles;
    }
    if (want_inbox && rep.inbox_doubles < film_doubles) {
        HIP_TRY(hipSetDevice(first.device));
        if (rep.inbox) (void)hipFree(rep.inbox);
        rep.inbox = nullptr;
        rep.inbox_doubles = 0;
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&rep.inbox), film_doubles * sizeof(double)));
        rep.inbox_doubles = film_doubles;
    }
    return RT_OK;
}

}  // namespace

int render_all_devices(rt_scene* sc, const rt_camera_desc* camera, const rt_render_opts* o, double* film, rt_stats* stats) {
    if (!sc || !camera || !o || !film) return fail(RT_ERR_INVALID, "null argument");
    const int n = (int)sc->replicas.size();
    if (n == 1) return render_into_film(sc, *sc->replicas[0], camera, o, film, stats);
    if (o->band_world > 1) return fail(RT_ERR_INVALID, "band_rank / band_world cannot be combined with a scene that lives on several devices");
    if (o->dx <= 0 || o->dy <= 0 || o->rays_per_pixel <= 0) return fail(RT_ERR_INVALID, "bad Dx / Dy / RaysPerPixel must be positive");
    if (o->sample_count < 0 || o->sample_begin < 0 || (o->sample_count > 0 && o->sample_begin + o->sample_count > o->rays_per_pixel))
        return fail(RT_ERR_INVALID, "sample range outside [0, RaysPerPixel)");
    const int s_first = o->sample_count > 0 ? o->sample_begin : 0;
    const int s_total = o->sample_count > 0 ? o->sample_count : o->rays_per_pixel;
    const size_t film_doubles = (size_t)o->dx * o->dy * 3;
    int split = o->device_split;
    if (split != RT_SPLIT_AUTO && split != RT_SPLIT_ROWS && split != RT_SPLIT_SAMPLES) return fail(RT_ERR_INVALID, "unknown device_split");
    if (split == RT_SPLIT_AUTO)  // few pixels, many samples: every device takes a run of samples (a film of at most 100 MB travels)
        split = (s_total / n >= 256 && (int64_t)o->dx * o->dy <= (int64_t)2048 * 2048) ? RT_SPLIT_SAMPLES : RT_SPLIT_ROWS;
    if (split == RT_SPLIT_SAMPLES && s_total < n) split = RT_SPLIT_ROWS;
    Replica& first = *sc->replicas[0];
    // every other device's share lands in an inbox of its own on the first device and is ADDED to the caller's film there:
    // rt_render_film adds to the film it is given (progressive passes), whichever way the frame is divided
    for (int i = 1; i < n; i++)
        if (int rc = ensure_peer_buffers(first, *sc->replicas[(size_t)i], film_doubles, true)) return rc;
    HIP_TRY(hipSetDevice(first.device));
    hipStream_t stream0 = reinterpret_cast<hipStream_t>(o->stream);

    const auto t_start = std::chrono::steady_clock::now();
    std::vector<Share> shares((size_t)n);
    auto run = [&](int i) {
        Share& sh = shares[(size_t)i];
        Replica& rep = *sc->replicas[(size_t)i];
        auto fail_here = [&](hipError_t e, const char* what) {
            sh.rc = RT_ERR_DEVICE;
            sh.error = std::string(what) + ": " + hipGetErrorString(e);
        };
        hipError_t e = hipSetDevice(rep.device);
        if (e != hipSuccess) return fail_here(e, "hipSetDevice");
        rt_render_opts mine = *o;
        mine.band_rank = 0;
        mine.band_world = 1;
        if (split == RT_SPLIT_ROWS) {
            mine.band_rank = i;
            mine.band_world = n;
        } else {  // the i-th of n nearly equal runs of the sample indices
            const int lo = (int)((int64_t)s_total * i / n), hi = (int)((int64_t)s_total * (i + 1) / n);
            mine.sample_begin = s_first + lo;
            mine.sample_count = hi - lo;
        }
        double* target = film;
        if (i > 0) {
            mine.stream = rep.stream;
            target = rep.film;
            e = hipMemsetAsync(target, 0, film_doubles * sizeof(double), rep.stream);
            if (e != hipSuccess) return fail_here(e, "hipMemsetAsync");
        }
        sh.rc = render_into_film(sc, rep, camera, &mine, target, &sh.stats);
        if (sh.rc != RT_OK) { sh.error = last_error_text(); return; }
        if (i == 0) return;
        // render_into_film has synchronised rep.stream: the share is complete.  Send it to its inbox on the first device:
        // the bands of this device (rows split) or the whole film (samples split).
        const bool direct = peer_copy_allowed(rep.device, first.device);
        const size_t row_doubles = (size_t)o->dx * 3;
        auto send = [&](size_t off, size_t count) -> hipError_t {
            if (direct) return hipMemcpyPeerAsync(rep.inbox + off, first.device, rep.film + off, rep.device, count * sizeof(double), rep.stream);
            // no direct path between the two devices: through pinned host memory (device -> host here, host -> inbox below)
            return hipMemcpyAsync(rep.staging + off, rep.film + off, count * sizeof(double), hipMemcpyDeviceToHost, rep.stream);
        };
        if (!direct && rep.staging_doubles < film_doubles) {
            if (rep.staging) (void)hipHostFree(rep.staging);
            rep.staging = nullptr;
            rep.staging_doubles = 0;
            e = hipHostMalloc(reinterpret_cast<void**>(&rep.staging), film_doubles * sizeof(double), hipHostMallocPortable);
            if (e != hipSuccess) return fail_here(e, "hipHostMalloc (staging copy between devices without peer access)");
            rep.staging_doubles = film_doubles;
        }
        if (split == RT_SPLIT_ROWS) {
            for (int band = i; band * 16 < o->dy && e == hipSuccess; band += n) {
                const size_t rows = (size_t)std::min(16, o->dy - band * 16), off = (size_t)band * 16 * row_doubles;
                e = send(off, rows * row_doubles);
            }
        } else {
            e = send(0, film_doubles);
        }
        if (e == hipSuccess) e = hipStreamSynchronize(rep.stream);
        if (e == hipSuccess && !direct) {
            e = hipSetDevice(first.device);
            if (split == RT_SPLIT_ROWS) {
                for (int band = i; band * 16 < o->dy && e == hipSuccess; band += n) {
                    const size_t rows = (size_t)std::min(16, o->dy - band * 16), off = (size_t)band * 16 * row_doubles;
                    e = hipMemcpy(rep.inbox + off, rep.staging + off, rows * row_doubles * sizeof(double), hipMemcpyHostToDevice);
                }
            } else if (e == hipSuccess) {
                e = hipMemcpy(rep.inbox, rep.staging, film_doubles * sizeof(double), hipMemcpyHostToDevice);
            }
        }
        if (e != hipSuccess) return fail_here(e, "copy to the first device");
    };
    std::vector<std::thread> threads;
    for (int i = 1; i < n; i++) threads.emplace_back(run, i);
    run(0);
    for (std::thread& t : threads) t.join();
    HIP_TRY(hipSetDevice(first.device));
    for (const Share& sh : shares)
        if (sh.rc != RT_OK) return fail(sh.rc, sh.error);
    for (int i = 1; i < n; i++) {
        if (split == RT_SPLIT_SAMPLES) launch_film_add(film, sc->replicas[(size_t)i]->inbox, film_doubles, stream0);
        else launch_film_add_bands(film, sc->replicas[(size_t)i]->inbox, o->dx, o->dy, i, n, stream0);
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(stream0));
    if (stats) {
        std::memset(stats, 0, sizeof(*stats));
        for (const Share& sh : shares) {
            stats->total_rays += sh.stats.total_rays;
            stats->rays_exceeded_depth += sh.stats.rays_exceeded_depth;
            stats->rays_left_scene += sh.stats.rays_left_scene;
            stats->shadow_rays += sh.stats.shadow_rays;
            stats->traverse_launches += sh.stats.traverse_launches;
            stats->traverse_rays += sh.stats.traverse_rays;
            stats->culled_rays += sh.stats.culled_rays;
            stats->traverse_seconds = std::max(stats->traverse_seconds, sh.stats.traverse_seconds);  // the slowest device's launches
        }
        stats->render_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
    }
    return RT_OK;
}

}  // namespace rt
