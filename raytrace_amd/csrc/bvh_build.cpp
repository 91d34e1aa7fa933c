// Builds, for one mesh, the binary BVH that the reference's NewBVH builds
// (phys/shape_bvh.go:81-272: binned SAH over the union-bounds extent of the longest
// axis, 16 bins, leaves of <= 4 shapes or at depth 32, stable left/right partition,
// equal-count fallback when no split plane has shapes on both sides), then lays it
// out twice: reference order (RefNode) and traversal-optimised (FastNode).
//
// Compiled with -ffp-contract=off: bounds, centroids and SAH costs must round the
// way the reference's float64 arithmetic does or the tree shape would differ.

#include "bvh_build.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <future>
#include <limits>
#include <memory>

namespace rt {
namespace {

constexpr double kInf = std::numeric_limits<double>::infinity();
constexpr int kMaxDepth = 32;         // shape_bvh.go:82
constexpr int kMinShapesPerLeaf = 4;  // shape_bvh.go:83
constexpr int kBins = 16;             // shape_bvh.go:84

// Padding of the float32 boxes as a fraction of the mesh radius.  The float32 slab
// test works on a ray re-based to the mesh frame; rounding the origin, the direction
// and the slab products moves a box face by at most ~10 * 2^-24 of the radius, so
// 2^-17 keeps every box conservative with a wide margin.
constexpr double kBoxPad = 1.0 / 131072.0;

// Go's math.Min / math.Max on finite-or-infinite inputs (shape.go:52-65 uses them).
inline double gmin(double x, double y) {
    if (x == -kInf || y == -kInf) return -kInf;
    if (std::isnan(x) || std::isnan(y)) return std::numeric_limits<double>::quiet_NaN();
    if (x == 0 && x == y) return std::signbit(x) ? x : y;
    return x < y ? x : y;
}
inline double gmax(double x, double y) {
    if (x == kInf || y == kInf) return kInf;
    if (std::isnan(x) || std::isnan(y)) return std::numeric_limits<double>::quiet_NaN();
    if (x == 0 && x == y) return std::signbit(x) ? y : x;
    return x > y ? x : y;
}

struct Box {
    double mn[3], mx[3];
};
inline Box empty_box() { return {{kInf, kInf, kInf}, {-kInf, -kInf, -kInf}}; }
inline Box unite(const Box& a, const Box& b) {  // AABB.Union, shape.go:52-65
    Box r;
    for (int k = 0; k < 3; k++) {
        r.mn[k] = gmin(a.mn[k], b.mn[k]);
        r.mx[k] = gmax(a.mx[k], b.mx[k]);
    }
    return r;
}
inline double surface_area(const Box& b) {  // shape.go:32-37
    double dx = b.mx[0] - b.mn[0], dy = b.mx[1] - b.mn[1], dz = b.mx[2] - b.mn[2];
    return 2 * (dx * dy + dy * dz + dz * dx);
}
inline int longest_axis(const Box& b) {  // shape.go:39-50
    double dx = b.mx[0] - b.mn[0], dy = b.mx[1] - b.mn[1], dz = b.mx[2] - b.mn[2];
    if (dx > dy && dx > dz) return 0;
    if (dy > dz) return 1;
    return 2;
}

struct TreeNode {
    Box box;
    std::unique_ptr<TreeNode> left, right;  // both null: leaf
    uint32_t first = 0, count = 0;          // leaf: range of the shared index array
    uint32_t depth = 0;
};

struct Builder {
    const Box* face_box;         // Face.Bounds(), shape_mesh.go:145-161
    std::vector<uint32_t> order; // shapes of the current recursion, partitioned in place
    std::vector<uint32_t> scratch;

    std::unique_ptr<TreeNode> build(uint32_t lo, uint32_t hi, int depth) {
        const uint32_t n = hi - lo;
        auto node = std::make_unique<TreeNode>();
        node->depth = (uint32_t)depth;
        if (n == 1 || depth >= kMaxDepth || (int)n <= kMinShapesPerLeaf) {  // shape_bvh.go:91-108
            Box b = face_box[order[lo]];
            for (uint32_t i = lo + 1; i < hi; i++) b = unite(b, face_box[order[i]]);
            node->box = b;
            node->first = lo;
            node->count = n;
            return node;
        }
        Box bbox = face_box[order[lo]];
        for (uint32_t i = lo + 1; i < hi; i++) bbox = unite(bbox, face_box[order[i]]);
        node->box = bbox;
        const int axis = longest_axis(bbox);
        const double bmin = bbox.mn[axis], bmax = bbox.mx[axis];
        auto centroid = [&](uint32_t f) { return (face_box[f].mn[axis] + face_box[f].mx[axis]) * 0.5; };  // shape.go:67-73
        auto bin_of = [&](double c) {  // shape_bvh.go:153-156
            double f = kBins * ((c - bmin) / (bmax - bmin));
            int idx = std::isnan(f) ? 0 : (int)f;  // zero extent: the reference would panic on int(NaN)
            if (idx == kBins) idx = kBins - 1;
            return idx;
        };
        struct Bin { Box b; int count; } bins[kBins];
        for (auto& b : bins) { b.b = empty_box(); b.count = 0; }
        for (uint32_t i = lo; i < hi; i++) {
            int bi = bin_of(centroid(order[i]));
            bins[bi].count++;
            bins[bi].b = unite(bins[bi].b, face_box[order[i]]);
        }
        int lc[kBins], rc[kBins];
        Box lb[kBins], rb[kBins];
        {
            int c = 0;
            Box b = empty_box();
            for (int i = 0; i < kBins; i++) { c += bins[i].count; b = unite(b, bins[i].b); lc[i] = c; lb[i] = b; }
            c = 0;
            b = empty_box();
            for (int i = kBins - 1; i >= 0; i--) { c += bins[i].count; b = unite(b, bins[i].b); rc[i] = c; rb[i] = b; }
        }
        const double total_sa = surface_area(bbox);
        double best_cost = std::numeric_limits<double>::max();
        int best_split = -1;
        for (int i = 0; i < kBins - 1; i++) {  // shape_bvh.go:198-207; an empty side gives 0*inf = NaN and is skipped
            double pl = surface_area(lb[i]) / total_sa;
            double pr = surface_area(rb[i + 1]) / total_sa;
            double cost = 1 + ((double)lc[i] * pl + (double)rc[i + 1] * pr);
            if (cost < best_cost) { best_cost = cost; best_split = i; }
        }
        uint32_t mid;
        if (best_split == -1) {  // shape_bvh.go:210-231 (sort.Slice there; a stable sort here)
            std::stable_sort(order.begin() + lo, order.begin() + hi,
                             [&](uint32_t a, uint32_t b) { return centroid(a) < centroid(b); });
            mid = lo + n / 2;
        } else {  // shape_bvh.go:233-246: order-preserving split
            auto it = std::stable_partition(order.begin() + lo, order.begin() + hi,
                                            [&](uint32_t f) { return bin_of(centroid(f)) <= best_split; });
            mid = (uint32_t)(it - order.begin());
        }
        if (n > 32768 && depth < 5) {  // the reference forks two goroutines per split (shape_bvh.go:249-261)
            auto fut = std::async(std::launch::async, [&, lo, mid, depth] { return build(lo, mid, depth + 1); });
            node->right = build(mid, hi, depth + 1);
            node->left = fut.get();
        } else {
            node->left = build(lo, mid, depth + 1);
            node->right = build(mid, hi, depth + 1);
        }
        return node;
    }
};

inline float round_down(double x) {
    float f = (float)x;
    if ((double)f > x) f = std::nextafterf(f, -std::numeric_limits<float>::infinity());
    return f;
}
inline float round_up(double x) {
    float f = (float)x;
    if ((double)f < x) f = std::nextafterf(f, std::numeric_limits<float>::infinity());
    return f;
}

// The reference's slab test rejects with `tmax <= tmin` (shape.go:91), so a box with
// zero extent on an axis is never entered (t0 == t1 on that axis for every ray that is
// not lying in the plane).  Such subtrees are unreachable and are dropped from the
// padded float32 tree, where they would otherwise become hittable.
inline bool never_hit(const Box& b) {
    return b.mn[0] == b.mx[0] || b.mn[1] == b.mx[1] || b.mn[2] == b.mx[2];
}

struct Flattener {
    MeshBvh& out;
    uint32_t max_depth = 0, max_leaf = 0;
    uint64_t leaves = 0;

    void ref(const TreeNode* n) {
        const int32_t me = (int32_t)out.ref_nodes.size();
        out.ref_nodes.emplace_back();
        {
            RefNode& r = out.ref_nodes[me];
            for (int k = 0; k < 3; k++) { r.box[k] = n->box.mn[k]; r.box[3 + k] = n->box.mx[k]; }
            r.right = -1; r.first = (int32_t)n->first; r.count = (int32_t)n->count; r.pad = 0;
        }
        max_depth = std::max(max_depth, n->depth);
        if (!n->left) {
            leaves++;
            max_leaf = std::max(max_leaf, n->count);
            return;
        }
        ref(n->left.get());
        out.ref_nodes[me].right = (int32_t)out.ref_nodes.size();
        ref(n->right.get());
    }

    void set_child(FastNode& f, int k, const TreeNode* c) {
        for (int a = 0; a < 3; a++) {
            f.lo[k][a] = round_down(c->box.mn[a] - out.center[a]) - out.pad;
            f.hi[k][a] = round_up(c->box.mx[a] - out.center[a]) + out.pad;
        }
    }

    int32_t fast(const TreeNode* n) {  // n is an inner node
        const int32_t me = (int32_t)out.fast_nodes.size();
        out.fast_nodes.emplace_back();
        const TreeNode* kids[2] = {n->left.get(), n->right.get()};
        for (int k = 0; k < 2; k++) {
            const TreeNode* c = kids[k];
            FastNode f = out.fast_nodes[me];
            set_child(f, k, c);
            if (never_hit(c->box)) { f.child[k] = 0; f.count[k] = -1; }
            else if (!c->left) { f.child[k] = (int32_t)c->first; f.count[k] = (int32_t)c->count; }
            else { f.count[k] = 0; out.fast_nodes[me] = f; int32_t ci = fast(c); f = out.fast_nodes[me]; f.child[k] = ci; }
            out.fast_nodes[me] = f;
        }
        return me;
    }
};

}  // namespace

void build_mesh_bvh(const rt_face* faces, uint64_t n_faces, MeshBvh& out) {
    auto t0 = std::chrono::steady_clock::now();
    std::vector<Box> boxes(n_faces);
    for (uint64_t i = 0; i < n_faces; i++) {
        const rt_face& f = faces[i];
        for (int k = 0; k < 3; k++) {
            boxes[i].mn[k] = gmin(gmin(f.v[0][k], f.v[1][k]), f.v[2][k]);
            boxes[i].mx[k] = gmax(gmax(f.v[0][k], f.v[1][k]), f.v[2][k]);
        }
    }
    Builder b;
    b.face_box = boxes.data();
    b.order.resize(n_faces);
    for (uint64_t i = 0; i < n_faces; i++) b.order[i] = (uint32_t)i;
    std::unique_ptr<TreeNode> root = b.build(0, (uint32_t)n_faces, 0);

    out.ref_nodes.clear();
    out.fast_nodes.clear();
    out.ref_nodes.reserve(n_faces);
    Flattener fl{out};
    fl.ref(root.get());

    double radius = 0;
    for (int k = 0; k < 3; k++) {
        out.root_box[k] = root->box.mn[k];
        out.root_box[3 + k] = root->box.mx[k];
        out.center[k] = 0.5 * (root->box.mn[k] + root->box.mx[k]);
        radius = std::max(radius, std::max(std::fabs(root->box.mn[k] - out.center[k]), std::fabs(root->box.mx[k] - out.center[k])));
    }
    out.pad = round_up(radius * kBoxPad);
    if (root->left) {
        out.fast_nodes.reserve(n_faces / 2 + 1);
        fl.fast(root.get());
    } else {  // the whole mesh is one leaf: give it a parent so traversal has a node to start from
        FastNode f{};
        fl.set_child(f, 0, root.get());
        if (never_hit(root->box)) { f.child[0] = 0; f.count[0] = -1; }
        else { f.child[0] = (int32_t)root->first; f.count[0] = (int32_t)root->count; }
        f.child[1] = 0; f.count[1] = -1;
        for (int a = 0; a < 3; a++) { f.lo[1][a] = 1.0f; f.hi[1][a] = -1.0f; }
        out.fast_nodes.push_back(f);
    }

    out.prim_face.resize(n_faces);
    out.prim_pos.resize(n_faces * 9);
    for (uint64_t i = 0; i < n_faces; i++) {
        const uint32_t fi = b.order[i];
        out.prim_face[i] = (int32_t)fi;
        for (int v = 0; v < 3; v++)
            for (int k = 0; k < 3; k++) out.prim_pos[i * 9 + v * 3 + k] = faces[fi].v[v][k];
    }
    out.info.n_faces = n_faces;
    out.info.n_nodes = out.ref_nodes.size();
    out.info.n_leaves = fl.leaves;
    out.info.max_depth = fl.max_depth;
    out.info.max_leaf_size = fl.max_leaf;
    out.info.build_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

}  // namespace rt
