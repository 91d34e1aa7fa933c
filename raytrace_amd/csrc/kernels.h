// Host-callable launchers of the HIP kernels (defined in kernels.hip).
#pragma once

#include <hip/hip_runtime.h>

#include "scene_types.h"

namespace rt {

// Which pixels a render pass covers and how path indices map to (pixel, sample).
struct PassDesc {
    int32_t dx, dy;
    int32_t band_rank, band_world;  // rows are cut into bands of 16; this process owns band b iff b % world == rank
    int32_t local_rows;             // image rows owned by this process
    int32_t tiles_x;                // 8x8 pixel tiles per row of tiles
    int32_t spp;                    // RaysPerPixel of the whole render
    int32_t sample_begin, sample_count;  // samples handled by this pass
    int64_t tile_begin;             // first 8x8 tile (in local tile order) of this pass
    int64_t n_paths;                // tiles * 64 * sample_count
    uint32_t seed_lo, seed_hi;
    int32_t max_depth;
    int32_t pad;
};

struct Queues {
    RayRec* rays;       // current generation
    HitRec* hits;
    RayRec* next;       // next generation (append)
    SurfRec* surfs;     // lit surface points of this generation (append)
    Counters* counters;
    uint32_t next_capacity, surf_capacity;
};

void launch_generate(const DCamera& cam, const PassDesc& pass, RayRec* rays, hipStream_t s);
void launch_extend(const DScene& sc, const RayRec* rays, HitRec* hits, uint32_t n, int mode, hipStream_t s);
void launch_shade(const DScene& sc, const PassDesc& pass, const Queues& q, uint32_t n, double* film, hipStream_t s);
void launch_direct(const DScene& sc, const Queues& q, uint32_t n_surf, int mode, double* film, hipStream_t s);
void launch_resolve(const double* film, int32_t dx, int32_t dy, int32_t spp, uint8_t* rgba, hipStream_t s);
void launch_intersect_batch(const DScene& sc, const double* o, const double* d, uint64_t n, double tmin, double tmax, int mode,
                            rt_hit* hits, hipStream_t s);
void launch_cast_batch(const DCamera& cam, const double* st, const uint32_t* pixel_sample, uint64_t n, uint32_t seed_lo,
                       uint32_t seed_hi, double* o, double* d, hipStream_t s);

}  // namespace rt
