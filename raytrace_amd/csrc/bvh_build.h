// Host-side BVH construction for one mesh and its two flat layouts.
#pragma once

#include <cstdint>
#include <vector>

#include "../../include/rt_hip.h"

// Bits of a leaf reference that hold (faces - 1) in the pixel beams' own tree (DMesh::beam_off); 3 in every other tree.
// Leaves of up to 16 faces (4 bits) were measured and changed nothing on the 4K scan (135.5 against 135.2 ms per frame).
#ifndef RT_BEAM_LEAF_BITS
#define RT_BEAM_LEAF_BITS 3
#endif

namespace rt {

// Node of the reference-order tree, depth-first pre-order: the left child of an
// inner node i is node i + 1, the right child is `right`.  count > 0 marks a leaf
// holding prims [first, first + count) of the leaf-ordered face arrays.  64 bytes.
struct RefNode {
    double box[6];  // min xyz, max xyz (float64, as the reference's AABB)
    int32_t right;
    int32_t first;
    int32_t count;
    int32_t pad;
};
static_assert(sizeof(RefNode) == 64, "RefNode layout");

// Node of the traversal-optimised tree: both children's boxes live in the parent so
// one 64-byte fetch decides both descents.  Boxes are float32, relative to the
// mesh's centre, rounded outward and padded (see kBoxPad in bvh_build.cpp).
// child[k] >= 0: inner node index.  child[k] == kFastEmpty: nothing there.  Any other negative
// value is a leaf: ~child[k] = first_prim << 3 | (count - 1), count <= 8 (larger leaves of the
// reference tree are cut into runs of <= 8 prims under extra inner nodes).
constexpr int32_t kFastEmpty = INT32_MIN;
struct FastNode {
    float b[2][3][2];  // [child][axis][0 = min, 1 = max]: the two planes of a slab are adjacent for packed math
    int32_t child[2];
    int32_t pad[2];
};
static_assert(sizeof(FastNode) == 64, "FastNode layout");

// The same node for walks that fetch it per lane (scattered and shadow rays: every lane of a wave stands on a different
// node, and what bounds those walks is the number of 16-byte requests the lanes send to the L1, not arithmetic): the
// twelve planes as 16-bit positions on a grid laid over the mesh's padded bounds, 32 bytes instead of 64.  Plane p of
// axis a lies at grid_lo[a] + p * grid_step[a] in the mesh frame; mins are rounded down and maxes up and both moved one
// more step outward, so a quantised box contains the float32 box of the FastNode with the same index and child
// references.  An empty slot keeps child == kFastEmpty.
struct QNode {
    uint16_t q[2][3][2];  // [child][axis][0 = min, 1 = max]
    int32_t child[2];
};
static_assert(sizeof(QNode) == 32, "QNode layout");

// Two levels of that tree in one 64-byte node: up to four (grand)children with their quantised boxes.  A walk that fetches
// its nodes per lane is bound by the chain of dependent fetches, one per level; this halves the chain.  A child that is
// a leaf (or whose own children are leaves) simply fills fewer slots; empty slots keep child == kFastEmpty and an
// inverted box.  Inner references index this array.
struct Q4Node {
    uint16_t q[4][3][2];  // [slot][axis][0 = min, 1 = max]
    int32_t child[4];
};
static_assert(sizeof(Q4Node) == 64, "Q4Node layout");

// Float32 copy of a face for the conservative pre-test that runs before the float64
// intersection: p0 relative to the mesh centre, the two edges, and their lengths (the
// error bound of the pre-test scales with them).  48 bytes.
struct Tri32 {
    float p0[3];
    float e1[3];
    float e2[3];
    float len1, len2;
    float pad;
};
static_assert(sizeof(Tri32) == 48, "Tri32 layout");

struct MeshBvh {
    std::vector<RefNode> ref_nodes;
    std::vector<FastNode> fast_nodes;
    // The same tree with every subtree of at most 8 faces folded into one leaf, for the pixel-beam walk, which lists leaves and
    // thins them with a lane per face: fewer, fuller leaves there (4K scan +8 %); the other walks lose 3-5 % on such a tree and
    // keep fast_nodes.  Empty unless asked for.
    std::vector<FastNode> beam_nodes;
    std::vector<QNode> q_nodes;       // index for index the same tree as fast_nodes
    std::vector<Q4Node> q4_nodes;     // the same tree, two levels per node
    int32_t q4_depth = 1;             // levels of inner nodes of q4_nodes
    float grid_lo[3], grid_step[3];   // the grid of q_nodes, mesh frame
    std::vector<int32_t> prim_face;   // leaf order -> face index inside the mesh
    std::vector<double> prim_pos;     // leaf order, 9 float64 per face (P0, P1, P2)
    std::vector<Tri32> prim_tri32;    // leaf order
    double center[3];                 // origin of the float32 frame
    double root_box[6];
    float pad;                        // absolute padding applied to every float32 box
    float radius;                     // largest |coordinate| of the mesh in its own frame
    int32_t fast_depth;               // levels of inner nodes in the fast tree
    rt_bvh_info info;
};

// Builds the tree NewBVH (phys/shape_bvh.go:81-272) builds for these faces.
void build_mesh_bvh(const rt_face* faces, uint64_t n_faces, MeshBvh& out, bool want_beam_tree = false);

}  // namespace rt
