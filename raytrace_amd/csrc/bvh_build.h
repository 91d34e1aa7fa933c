// Host-side BVH construction for one mesh and its two flat layouts.
#pragma once

#include <cstdint>
#include <vector>

#include "../../include/rt_hip.h"

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
// child[k] >= 0 with count[k] == 0: inner node index; count[k] > 0: leaf with prims
// [child[k], child[k] + count[k]); count[k] < 0: nothing there (skip).
struct FastNode {
    float lo[2][3];
    float hi[2][3];
    int32_t child[2];
    int32_t count[2];
};
static_assert(sizeof(FastNode) == 64, "FastNode layout");

struct MeshBvh {
    std::vector<RefNode> ref_nodes;
    std::vector<FastNode> fast_nodes;
    std::vector<int32_t> prim_face;   // leaf order -> face index inside the mesh
    std::vector<double> prim_pos;     // leaf order, 9 float64 per face (P0, P1, P2)
    double center[3];                 // origin of the float32 frame
    double root_box[6];
    float pad;                        // absolute padding applied to every float32 box
    rt_bvh_info info;
};

// Builds the tree NewBVH (phys/shape_bvh.go:81-272) builds for these faces.
void build_mesh_bvh(const rt_face* faces, uint64_t n_faces, MeshBvh& out);

}  // namespace rt
