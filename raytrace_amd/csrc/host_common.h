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
extern std::atomic<bool> g_initialized;
extern int g_device;

#define HIP_TRY(expr)                                                                                             \
    do {                                                                                                          \
        hipError_t e_ = (expr);                                                                                   \
        if (e_ != hipSuccess) return rt::fail(RT_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

struct DeviceBuffers {
    std::vector<void*> ptrs;
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

}  // namespace rt

struct rt_scene {
    rt::DeviceBuffers mem;
    rt::DScene dev{};
    std::vector<rt_bvh_info> bvh_info;
    // host copies of what the screen-space occupancy map of a camera is made from
    std::vector<double> node_box;        // DScene::node_box, 6 per node
    std::vector<rt::DNode> nodes;
    std::vector<rt::DMesh> meshes;
    std::vector<rt::FastNode> fast_nodes;
    std::atomic<bool> cancel{false};
    rt::WorkBuffers* work = nullptr;
    bool all_terminal = false;  // every material only emits: no shade / light-loop launches are ever needed
    bool mesh_only = false;     // every node is an untransformed mesh: kernels without the analytic shapes are used
    int device = 0;
};

namespace rt {
// The wavefront loop for one process's share of the image; adds radiance sums into film (device).
int render_into_film(rt_scene* sc, const rt_camera_desc* camera, const rt_render_opts* o, double* film, rt_stats* stats);
// Scratch buffers owned by the scene for rt_render (film sums, the 8-bit image and its pinned host copy), grown on demand.
int scene_film(rt_scene* sc, int dx, int dy, double** film, uint8_t** rgba, uint8_t** rgba_host);
}  // namespace rt
