// Declarations shared by the host-side translation units of librt_hip.so.
#pragma once

#include <hip/hip_runtime.h>

#include <atomic>
#include <memory>
#include <string>
#include <vector>

#include "kernels.h"
#include "scene_types.h"

namespace rt {

int fail(int code, const std::string& msg);  // records the thread-local message, returns code
const std::string& last_error_text();        // this thread's message (threads of a multi-device render hand theirs to the caller)
extern std::atomic<bool> g_initialized;
// The devices rt_init selected, in the caller's order.  The same ordinal may appear more than once: every entry gets a
// replica and a stream of its own (a one-GPU machine can rehearse the multi-device path that way).
extern std::vector<int> g_devices;
// Whether films may travel from device `from` to device `to` with a direct device-to-device copy: the same device, or a pair
// for which rt_init found and enabled peer access.  Otherwise they go through pinned host memory.  RT_NO_PEER=1 (read by
// rt_init) forbids every direct copy, which lets a machine with one GPU run the staged route.
bool peer_copy_allowed(int from, int to);

#define HIP_TRY(expr)                                                                                             \
    do {                                                                                                          \
        hipError_t e_ = (expr);                                                                                   \
        if (e_ != hipSuccess) return rt::fail(RT_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

struct DeviceBuffers {
    std::vector<void*> ptrs;
    DeviceBuffers() = default;
    DeviceBuffers(const DeviceBuffers&) = delete;
    DeviceBuffers& operator=(const DeviceBuffers&) = delete;
    ~DeviceBuffers() {
        for (void* p : ptrs) (void)hipFree(p);
    }
    template <typename T>
    hipError_t upload(const std::vector<T>& host, const T** out) {
        T* d = nullptr;
        size_t bytes = (host.empty() ? 1 : host.size()) * sizeof(T);
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
        hipError_t e = hipMalloc(reinterpret_cast<void**>(&d), (count ? count : 1) * sizeof(T));
        if (e == hipSuccess) ptrs.push_back(d);
        *out = d;
        return e;
    }
};

struct WorkBuffers;  // queues of the wavefront loop (render.cpp)
void destroy_work(WorkBuffers* w);

int make_camera(const rt_camera_desc& c, DCamera& d);

// One device's copy of a scene: the flattened arrays, the queues of the wavefront loop and, on every device but the
// first, a stream and a film of its own for its share of a multi-device render.
struct Replica {
    int device = 0;
    DeviceBuffers mem;
    DScene dev{};
    WorkBuffers* work = nullptr;
    hipStream_t stream = nullptr;  // created for replicas 1..n-1; replica 0 renders on the caller's stream
    double* film = nullptr;        // replicas 1..n-1: this device's radiance sums of the frame being rendered
    size_t film_doubles = 0;
    double* inbox = nullptr;       // on the FIRST device: where this replica's share of the film lands before it is added
    size_t inbox_doubles = 0;
    double* staging = nullptr;     // pinned host memory: the way to the inbox when the two devices have no direct path
    size_t staging_doubles = 0;
    hipEvent_t done = nullptr;     // recorded on `stream` when this replica's share of a frame is finished
    // Pixel-beam tables (scenes the beam kernel can render: allocated with the scene, filled by k_beam_prepare whenever a
    // frame's camera direction differs from beam_dir)
    BeamTables beam{};
    int64_t beam_faces = 0;
    double beam_key[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};  // what the tables were last filled for: a direction and the camera's image axes u, v, or (beam.pyramid) a point
    bool beam_key_valid = false;
};

}  // namespace rt

struct rt_scene {
    std::vector<std::unique_ptr<rt::Replica>> replicas;  // one per device of rt_init, in that order
    uint64_t serial = 0;  // distinguishes scene handles over the life of the process (addresses are reused)
    std::vector<rt_bvh_info> bvh_info;
    // host copies of what the screen-space occupancy map of a camera is made from
    std::vector<double> node_box;        // DScene::node_box, 6 per node
    std::vector<rt::DNode> nodes;
    size_t n_materials = 0;
    std::vector<rt::DMesh> meshes;
    std::vector<rt::FastNode> fast_nodes;
    std::atomic<bool> cancel{false};
    bool all_terminal = false;  // every material only emits: no shade / light-loop launches are ever needed
    bool mesh_only = false;     // every node is an untransformed mesh: kernels without the analytic shapes are used
    ~rt_scene();
};

namespace rt {
// The wavefront loop for one replica's share of the image; adds radiance sums into film (memory of that replica's device).
// static_work: render with these buffers (the library's static scratch) instead of the replica's own; nothing is allocated then.
int render_into_film(rt_scene* sc, Replica& rep, const rt_camera_desc* camera, const rt_render_opts* o, double* film, rt_stats* stats,
                     WorkBuffers* static_work = nullptr);
// The whole frame over every replica of the scene: bands of rows or runs of samples per device, combined on the first device.
int render_all_devices(rt_scene* sc, const rt_camera_desc* camera, const rt_render_opts* o, double* film, rt_stats* stats);
// Scratch buffers owned by a replica for rt_render (film sums, the 8-bit image and its pinned host copy), grown on demand.
// There are two such sets (slot 0 / 1): a frame loop alternates between them.
int scene_film(const rt_scene* sc, Replica& rep, int dx, int dy, double** film, uint8_t** rgba, uint8_t** rgba_host, int slot = 0);
// The WorkBuffers that live in the static scratch of `device` (see kernels.hip); null with a message when the frame does not fit.
WorkBuffers* static_work_for(int device, int64_t tiles_total);
}  // namespace rt
