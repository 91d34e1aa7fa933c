// Plain structs shared by the host code that uploads a scene and the kernels that read it.
#pragma once

#include <cstdint>

#include "../../include/rt_hip.h"
#include "bvh_build.h"

namespace rt {

struct DNode {
    int32_t kind, index, material, pad;
};

struct DSphere {
    double c[3];
    double r;
    int32_t uvmap, pad;
};

// Quad with the per-call set-up of Quad.Collide (shape_quad.go:40-64, 90-97) done once
// on the host with the same float64 operations in the same order.
struct DQuad {
    double nrm[3];
    double p0[3], p1[3], p2[3], p3[3];
    double lu[3], lv[3];  // localU = p2 - p1, localV = p0 - p1
    double lulu, lvlv;    // localU.localU, localV.localV
};

struct DCylinder {
    double o[3];
    double d[3];    // Direction.Unit() (shape_cylinder.go:41)
    double top[3];  // Origin + d * Height (shape_cylinder.go:100)
    double r, h;
};

struct DTri {
    double p0[3], p1[3], p2[3];
    double uv[6];
    double n[3];  // Normal.Unit() for TriangleUV (shape_triangle_uv.go:147)
    int32_t kind, pad;
};

struct DXform {  // transform.go:53-102
    double tr[3];
    double rot[9];
    double sc[3];
};

struct DTransformed {
    int32_t child_kind, child_index;
    DXform fwd, inv;  // inv = Transform.Inverse() (transform.go:86-102)
};

struct DMesh {
    int64_t ref_off;   // first RefNode of this mesh
    int64_t fast_off;  // first FastNode
    int64_t prim_off;  // first prim (leaf order) in prim_pos / prim_face
    int64_t face_off;  // first face (caller order) in face_uv
    double center[3];
    double root_box[6];
    float pad;
    float radius;  // largest |coordinate| in the mesh frame (error bound of the float32 face pre-test)
    int32_t n_faces;
    int32_t pad2;
    float grid_lo[3], grid_step[3];  // the grid of this mesh's QNodes (mesh frame)
    int64_t q4_off;    // first Q4Node
    int64_t beam_off;  // first FastNode of the tree the pixel-beam walk uses (leaves folded to eight faces); = fast_off when there is none
};

struct NodePair32 {
    float lo[3][2];  // [axis][node of the pair]
    float hi[3][2];
};

struct DImage {
    const uint8_t* px;
    int32_t w, h, format, pad;
};

struct DScene {
    const DNode* nodes;
    const DSphere* spheres;
    const DQuad* quads;
    const DCylinder* cylinders;
    const DTri* tris;
    const DTransformed* xforms;
    const DMesh* meshes;
    const RefNode* ref_nodes;
    const FastNode* fast_nodes;
    const QNode* q_nodes;      // the same trees, quantised: index for index fast_nodes
    const Q4Node* q4_nodes;    // the same trees, quantised, two levels per node (DMesh::q4_off)
    const double* prim_pos;    // 9 per prim, leaf order
    const Tri32* prim_tri32;   // leaf order
    const int32_t* prim_face;  // leaf order -> face index within its mesh
    const double* face_uv;     // 6 per face, caller order
    const rt_material* materials;
    const rt_texture* textures;
    const DImage* images;
    const rt_light* lights;
    const double* node_box;  // per node: padded world-space bounds (min xyz, max xyz); fast mode skips a node the ray misses
    // The same bounds for two nodes at a time as float32 in the scene frame (origin pair_center, rounded outward and padded by
    // 2^-16 of pair_radius): a ray that starts within 8 radii of the centre tests a pair with packed instructions, nine per
    // node instead of the 26 of the float64 test.  Nodes that hold a mesh (which clips its own root box) or whose bounds dwarf
    // the rest of the scene (a sky sphere) carry infinite bounds: always entered.  Null when the scene has no such frame.
    const NodePair32* node_pairs;
    double pair_center[3];
    double pair_radius;
    int32_t n_nodes, n_lights;
    int32_t stack_pairs;  // depth of the deepest fast tree + 1: capacity of the LDS traversal stack
    int32_t walk_cap;     // entries a k_walk lane's stack can need: a four-wide node pushes up to three per level
    int32_t n_meshes;     // meshes in the scene (0: no tree walk anywhere, launches favour registers over occupancy)
    // Scenes that mix meshes with other shapes queue scattered rays in two classes: those that can reach a mesh (they
    // pass through walk_box, the bounds of all mesh nodes) and the others, so that the waves of the next launch hold
    // either rays that walk a tree or rays that do not, instead of a few walkers per wave.
    double walk_box[6];
    int32_t walk_classes;  // 1: the ray queues have kRayStripes stripes, the upper half for the rays that reach walk_box
    // The nodes whose shape is a mesh, directly or under transforms, in node order: the scattered-ray kernel tests every
    // other node when it fetches a ray and then walks these trees one after the other, resumably.
    int32_t n_walk_nodes;
    const int32_t* walk_nodes;
    uint64_t walk_mask;  // the same as bits, for scenes of at most 64 nodes (bit i: node i holds a mesh)
    int32_t early_miss;  // 1: a scattered ray that enters no node's bounds is settled where it is made (0 with RT_NO_EARLY_MISS set: it is queued like the others)
    // k_walk, per cent: a round's descent to the next leaves ends once fewer than this share of its lanes are still among inner
    // nodes (they go on in the next round); 0 = every lane reaches a leaf first.  Deep trees, where descents differ most in length.
    int32_t walk_break;
    int32_t pad_walk_break;
    int32_t sort_kinds;  // 1: the nodes' materials run more than one kind of Resolve code: hits are taken in the order of their materials (k_shade)
};

// ---- tables of the pixel-beam kernel (dev_beam.h): per face of the mesh and per camera direction ----

// float32 pre-test of one face for one direction, mesh-centred frame.  h and g carry the sign of a, so that "inside" always
// means us >= neg_eu, vs >= neg_ev, us + vs <= lim.  A face whose float64 test can never pass (|a| < 1e-8) has
// neg_eu = +inf (always rejected); one whose sign of a float32 cannot decide has neg_eu = neg_ev = -inf, lim = +inf
// (never rejected: float64 decides).
struct BeamFace32 {
    float p0[3];
    float h[3];  // sign(a) d x e2
    float g[3];  // sign(a) e1 x d
    float neg_eu, neg_ev, lim;
    // parallel rays only: the face's extent along the camera's two image axes, in the mesh-centred frame (rounded outward).  A
    // pixel's bundle whose footprint lies beside it cannot hit the face: the axes the three edge tests above leave out.
    float umin, umax, vmin, vmax;
};
static_assert(sizeof(BeamFace32) == 64, "the first three 16-byte pieces are what a ray's pre-test reads");

// The ray-independent half of Face.Collide for one direction.  f is NaN for a face the reference rejects whatever the ray.
struct BeamFace64 {
    double p0[3], e1[3], e2[3], h[3];
    double f;
    double pad;
};
static_assert(sizeof(BeamFace64) == 112, "seven 16-byte scalar loads");

// What the beam kernel reads besides the scene: the tables above and the faces' texture coordinates (6 per face), all in
// the leaf order of the mesh's trees.
struct BeamTables {
    const BeamFace32* f32;
    const BeamFace64* f64;
    const double* uv;
    int32_t safe_camera;  // 1: every origin the camera can produce is finite (the host has looked at its constants): the NaN / Inf guard of render.go:147-150 cannot fire
    int32_t image;        // 1: the material is an Emitter whose texture is a TextureImage (the kernel samples it in line)
    int32_t pyramid;      // 0: the tables are for a direction (orthographic camera); 1: for a point every ray starts from (pinhole camera, focusable camera without aperture)
    int32_t walk_stride;     // dwords of a wave's stretch of walk_scratch
    int32_t walk_waves;      // stretches in walk_scratch
    int32_t* walk_scratch;   // one stretch per persistent wave of k_beam: the states and lists of the pixels whose trees it has just walked (kernels.hip, beam_walk_batch)
    unsigned long long* long_px;  // kBeamLongCap entries: the pixels of a launch whose rounds walk the tree again (milliseconds each), handed from the wave that met them to whichever wave draws next (kernels.hip, k_beam)
};
constexpr uint32_t kBeamLongCap = 1u << 20;  // (a wave that finds the queue full traces the pixel itself)

struct DCamera {
    int32_t kind, pad;
    double from[3];
    double u[3], v[3], w[3];  // basis (camera_orthographic.go:26-28 and friends)
    double dir[3];            // orthographic: unit(LookAt - LookFrom)
    double fov_w, fov_h;
    double horizontal[3], vertical[3], llc[3];  // focusable / pinhole image plane
    double lens_radius;
    double origin[3];  // pinhole
    rt_camera_desc intr;  // calibrated intrinsics are read from here
};

// Queues are structure-of-arrays: lane i of a wave reads element i of every field, so each
// load or store instruction of a wave covers one contiguous run of memory.

// Path segments waiting to be traced (render.go:22-30).
struct RayQ {
    double* o[3];
    double* d[3];
    double* rad[3];     // ray.radiance
    uint32_t* pixel;    // film index: image row * dx + x (row 0 = top)
    uint32_t* sample;
    uint32_t* path;     // position in the path tree (root 1, children 2p, 2p+1)
    uint32_t* depth_n;  // depth << 24 | random draws made so far on this ray's stream
};
constexpr size_t kRayBytes = 9 * 8 + 4 * 4;

// Lit surface points waiting for their light loop (material_*.go ComputeDirectLighting).
struct SurfQ {
    double* p[3];
    double* n[3];       // normal used for the offset and the cosine (flipped inside a dielectric)
    double* wo[3];      // unit direction back along the incoming ray
    double* albedo[3];  // Lambertian texture value / Metal albedo
    uint32_t* pixel;
    int32_t* node;
    int32_t* material;
    int32_t* inside;    // dielectric: the hit was from inside
};
constexpr size_t kSurfBytes = 12 * 8 + 4 * 4;

// Closest hits of the generation of rays being traced, written by k_primary / k_walk and shaded by k_shade.  A record
// names the ray (its slot in the ray queue), the node, the distance and, for a mesh, the face (position in leaf order).
// Everything else about the hit (barycentrics, point, normal, texture coordinates) is computed by k_shade, which
// intersects that one face or shape again, limited to the recorded distance: the same arithmetic on the same inputs.
struct HitQ {
    uint32_t* ray;   // slot in rays[in]; kNoRay = unused entry of a claimed block
    int32_t* node;
    int32_t* leaf;   // mesh: position of the face in leaf order; other shapes: -1
    double* t;
};
constexpr size_t kHitBytes = 8 + 3 * 4;
constexpr uint32_t kNoRay = 0xffffffffu;
constexpr uint32_t kClaim = 256;  // queue entries a wave of k_walk reserves at a time (rays to trace, hit records to write)

// An atomic on one address is performed at the memory side (the L2s of the XCDs are not coherent with each other)
// and only some tens of millions of them finish per second: every wave of a launch adding to the same counter is
// what bounds the launch.  So the statistics are kept in kTallyStripes copies, each in a 128-byte line of its own
// (a wave picks one by its workgroup index; the host adds them up), what the host can count itself (rays launched,
// occlusion rays) is not counted on the device at all, and each cursor has a line of its own.
constexpr int kTallyStripes = 64;
// The append queues are cut into kQueueStripes regions of equal capacity with a cursor each, for the same reason: a
// wave reserves its slots in one stripe (chosen so that the stripes fill evenly), the host reads all cursors and the
// next launch walks every stripe.
constexpr int kQueueStripes = 32;
constexpr int kRayStripes = 2 * kQueueStripes;  // ray queues of scenes with two classes of rays (DScene::walk_classes)
struct Counters {
    struct alignas(128) Tally {
        unsigned long long total;     // RenderStats (render.go:38-45): rays counted without being launched (children cut or invalid)
        unsigned long long exceeded;
        unsigned long long left;
        unsigned long long culled;    // camera rays counted (in total and left) without being generated: their pixel's bundle of rays cannot reach the scene
    } tally[kTallyStripes];
    struct alignas(128) Cursor {
        unsigned int v;
    };
    Cursor errors;                             // what must never happen, counted: a traversal stack or a hit stripe running full (kept with the statistics: the host checks it whenever it reads them)
    Cursor ray_cursor[2][kRayStripes];         // append cursors of the two ray queues, one per stripe (first of the cursors the host resets)
    Cursor surf_cursor[kQueueStripes];         // append cursors of the lit surface points
    Cursor tile_cursor[8];                     // next tile of each XCD's share of a primary launch
    Cursor in_cursor[kRayStripes];             // k_walk: next unclaimed ray of each stripe of the generation being traced
    Cursor hit_cursor[kQueueStripes];          // k_walk: append cursors of the hit records, one per stripe
    struct alignas(128) LongPixels {           // k_beam: the queue BeamTables::long_px
        unsigned long long head_tail;          // entries taken (low half) and entries reserved (high half): one load reads both
    } long_px;
    Cursor cheap_cursor[8];                    // k_beam: next of the cheap tiles at the end of each XCD's share
    Cursor long_done[8];                       // k_beam: waves of each XCD that have found the tile queues empty, and so add nothing more to that queue
};

}  // namespace rt
