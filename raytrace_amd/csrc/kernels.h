// Host-callable launchers of the HIP kernels (defined in kernels.hip).
#pragma once

#include <hip/hip_runtime.h>

#include "scene_types.h"

namespace rt {

// Which pixels a render pass covers and how its threads map to (pixel, sample).
struct PassDesc {
    int32_t dx, dy;
    int32_t band_rank, band_world;  // rows are cut into bands of 16; this process owns band b iff b % world == rank
    int32_t local_rows;             // image rows owned by this process
    int32_t groups_x;               // 64x64-pixel groups (8x8 tiles of 8x8 pixels) per row of groups
    int32_t spp;                    // RaysPerPixel of the whole render
    int32_t sample_begin, sample_count;  // sample indices handled by this pass (each thread loops over them)
    int32_t max_depth;
    int32_t log2_s;                 // primary launch: 2^log2_s lanes of a wave trace samples of the same pixel
    int32_t block_px;               // primary launch: pixels of an 8x8 tile per work item (power of two >= 64 >> log2_s)
    int32_t acc_offset;             // primary launch: where the pixels' running sums start in dynamic LDS, in dwords (set by launch_primary)
    int32_t beam_offset;            // ... and where the pixel beams' lists and walk states start (set by launch_beam)
    int32_t beam_cands;             // pixel beams: leaves a pixel's list holds at a time (set by launch_beam)
    uint32_t long_cap;              // pixel beams: entries of BeamTables::long_px a launch may use (set by launch_beam); a wave that finds them used up traces its long pixels itself
    uint32_t beam_serial;           // pixel beams: number of the launch (set by launch_beam): marks the entries of BeamTables::long_px that belong to it; 0: that queue is not used
    int32_t walk_px;                // pixel beams: pixels whose trees are walked together, a lane each: block_px times a power of two, at most 32 (set by launch_beam)
    int64_t tile_begin;             // first 8x8 tile of this pass; tiles are numbered group by group
    int64_t n_pixels;               // tiles * 64 = pixels (threads) of this pass
    uint32_t seed_lo, seed_hi;
    // Work order of the primary launch.  Tiles belong to 8 queues (group of 8x8 tiles g -> queue g % 8).  With
    // `order` set, queue k hands out order[queue_off[k] + i], i < queue_len[k] (tile numbers, costliest
    // first); without it, its tiles in ascending order.  With `cost` set, every work item adds the clock ticks
    // it took to cost[tile].
    int32_t max_waves;              // primary launch: at most this many persistent waves (the queue stripes are sized for them)
    int32_t cull_pixels;            // pixel beams: a pixel whose bundle of rays misses the mesh is settled without generating its rays (they are counted)
    const int32_t* order;
    uint32_t* cost;
    uint32_t queue_off[8], queue_len[8];
    uint32_t queue_cheap[8];        // that many tiles at the end of each queue cost little (k_beam hands them out once the others of ALL queues are gone)
};

struct Queues {
    RayQ rays[2];   // two generations: a launch traces rays[in] and appends the scattered children to rays[out]
    SurfQ surfs;    // lit surface points of the generation being shaded
    HitQ hits;      // closest hits of the generation being traced
    Counters* counters;
    uint32_t ray_stripe_cap, surf_stripe_cap;  // entries per stripe; stripe s of a queue is [s * cap, (s + 1) * cap)
    uint32_t hit_stripe_cap;                   // a multiple of kClaim
    int2* walk_spill;  // k_walk: what the LDS ring of a lane's walk stack cannot hold; wave w, entry i, lane l at [(w * walk_cap + i) * 64 + l]
};

constexpr uint32_t kWalkWavesMax = 8192;  // persistent waves of k_walk at most (the hit queue has room for a partly filled block of each)

// Entries waiting in each stripe of a queue (what the cursors said, less what the host set aside).
struct StripeCounts {
    uint32_t n[kQueueStripes];
};
struct RayStripeCounts {
    uint32_t n[kRayStripes];  // the upper half stays 0 unless the scene queues two classes of rays
};

// cam and q point to DEVICE copies of the structs
void launch_primary(const DScene& sc, const DCamera* cam, const PassDesc& pass, const Queues* q, int mode, bool mesh_only, bool shade,
                    double* film, hipStream_t s);
int beam_pixels_max();
// The pixel-beam kernel (k_beam) and the pre-pass that fills its tables for a camera direction (dev_beam.h).  launch_beam
// returns false when the pass cannot run on it (fewer than 64 lanes per pixel, larger items, one sample per pixel).
void launch_beam_prepare(const DScene& sc, int64_t n_faces, bool pyr, const double key[9], BeamFace32* f32, BeamFace64* f64, double* uv, hipStream_t s);  // key: the direction and the camera's two image axes, or (pyr) the point and zeros
bool launch_beam(const DScene& sc, const DCamera* cam, const PassDesc& pass, Counters* counters, const BeamTables& bt, double* film, hipStream_t s);
int beam_walk_stride();  // dwords of a persistent wave's stretch of BeamTables::walk_scratch
int beam_walk_waves();   // stretches a launch on the current device can need
size_t beam_table_bytes(int64_t n_faces, size_t* off_f64, size_t* off_uv);  // one allocation: BeamFace32[n], then BeamFace64[n], then 6 doubles per face
// tracePath for the first n.n[s] rays of every stripe s of rays[in]: closest hit, Material.Resolve, children appended to
// rays[1 - in], lit surface points to q.surfs, emission to the film.  Scenes with a mesh: k_walk (closest hits into
// q.hits, lanes refilled as their rays finish) then k_shade; scenes without: one fused launch (k_trace).
void launch_trace(const DScene& sc, const PassDesc& pass, const Queues& q, int in, const RayStripeCounts& n, int mode, bool mesh_only, double* film,
                  hipStream_t s);
void launch_direct(const DScene& sc, const Queues& q, const StripeCounts& n_surf, int mode, double* film, hipStream_t s);
void launch_resolve(const double* film, int32_t dx, int32_t dy, int32_t spp, uint8_t* rgba, hipStream_t s);
void launch_film_add(double* dst, const double* src, size_t n, hipStream_t s);  // dst[i] += src[i], both on the current device
void launch_film_add_bands(double* dst, const double* src, int dx, int dy, int first, int step, hipStream_t s);  // the same on the rows of every step-th 16-row band from `first`

// Static scratch of rt_render_film_noalloc: a file-scope __device__ array in kernels.hip.  Layout (offsets in bytes, each
// a multiple of 256): Counters, DCamera, Queues, then kStaticTiles tile costs and kStaticTiles tile numbers.
constexpr int64_t kStaticTiles = 1 << 18;  // 8x8-pixel tiles: frames up to 4096 x 4096 pixels
constexpr int kStaticScratchDevices = 64;
constexpr size_t align256(size_t n) { return (n + 255) & ~(size_t)255; }
constexpr size_t kStaticOffCamera = align256(sizeof(Counters));
constexpr size_t kStaticOffQueues = kStaticOffCamera + align256(sizeof(DCamera));
constexpr size_t kStaticOffCost = kStaticOffQueues + align256(sizeof(Queues));
constexpr size_t kStaticOffOrder = kStaticOffCost + (size_t)kStaticTiles * sizeof(uint32_t);
constexpr size_t kStaticScratchBytes = kStaticOffOrder + (size_t)kStaticTiles * sizeof(int32_t);
// Address of that array on `device` (looked up once per device with that device current, then kept).
hipError_t static_scratch(int device, void** ptr);
// src and dst: packed dx*dy RGBA8 device images; tmp: dx*dy*3 float64
#ifdef RT_WAVE_TIMES
void debug_wave_times(unsigned long long* out, bool reset);  // diagnostics build: k_beam's persistent waves, start / end (100 MHz clock) / items each
#endif
#ifdef RT_VISIT_STATS
void debug_packet_stats(unsigned long long* out8, bool reset);  // diagnostics build only
void debug_trace_stats(unsigned long long* out32, bool reset);
#endif
void launch_filter(const uint8_t* src, int32_t w, int32_t h, int filter, double* tmp, uint8_t* dst, hipStream_t s);
void launch_intersect_batch(const DScene& sc, const double* o, const double* d, uint64_t n, double tmin, double tmax, int mode,
                            rt_hit* hits, hipStream_t s);
void launch_microfacet_batch(int which, uint64_t n, const double* roughness, const double* f0, const double* args, double* out, hipStream_t s);
void launch_shade_batch(const DScene& sc, const rt_shade_in* in, uint64_t n, uint32_t seed_lo, uint32_t seed_hi, int mode, rt_shade_out* out, hipStream_t s);
void launch_cast_batch(const DCamera& cam, const double* st, const uint32_t* pixel_sample, uint64_t n, uint32_t seed_lo,
                       uint32_t seed_hi, double* o, double* d, hipStream_t s);

}  // namespace rt
