// Builds, for one mesh, the binary BVH that the reference's NewBVH builds
// (phys/shape_bvh.go:81-272: binned SAH over the union-bounds extent of the longest
// axis, 16 bins, leaves of <= 4 shapes or at depth 32, stable left/right partition,
// equal-count fallback when no split plane has shapes on both sides), then lays it
// out twice: reference order (RefNode) and traversal-optimised (FastNode).
//
// Compiled with -ffp-contract=off: bounds, centroids and SAH costs must round the
// way the reference's float64 arithmetic does or the tree shape would differ.

#include "bvh_build.h"
#include "host_threads.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include <future>
#include <limits>
#include <memory>
#include <thread>
#include <vector>

namespace rt {
namespace {

constexpr uint64_t kWideFaces = 1u << 17;  // passes over at least this many faces are cut into chunks

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
    std::vector<uint32_t> scratch;  // the other buffer of the partition (recursions work on disjoint ranges of all three arrays)
    std::vector<uint8_t> bin_at;    // bin of the shape at each position of `order`, from the binning pass of the node at hand

    // known: the union of the shapes' bounds when the parent already has it (the bins of its split hold it), else null
    std::unique_ptr<TreeNode> build(uint32_t lo, uint32_t hi, int depth, const Box* known = nullptr) {
        const uint32_t n = hi - lo;
        auto node = std::make_unique<TreeNode>();
        node->depth = (uint32_t)depth;
        if (n == 1 || depth >= kMaxDepth || (int)n <= kMinShapesPerLeaf) {  // shape_bvh.go:91-108
            Box b = face_box[order[lo]];
            if (known) b = *known;
            else for (uint32_t i = lo + 1; i < hi; i++) b = unite(b, face_box[order[i]]);
            node->box = b;
            node->first = lo;
            node->count = n;
            return node;
        }
        // A node of very many shapes runs its three passes (bounds, binning, partition) in chunks on several threads.  Union
        // and counting are associative and the partition keeps the order, so the tree is the same tree.
        const bool wide = n >= kWideFaces && host_threads() > 1;
        const int parts = wide ? host_threads() : 1;
        Box bbox = face_box[order[lo]];
        if (known) {
            bbox = *known;
        } else if (wide) {
            std::vector<Box> part_box((size_t)parts, empty_box());
            parallel_chunks(lo, hi, parts, [&](int p, uint64_t a, uint64_t e) {
                Box b = face_box[order[a]];
                for (uint64_t i = a + 1; i < e; i++) b = unite(b, face_box[order[i]]);
                part_box[(size_t)p] = b;
            });
            bbox = part_box[0];
            for (int p = 1; p < parts; p++) bbox = unite(bbox, part_box[(size_t)p]);
        } else {
            for (uint32_t i = lo + 1; i < hi; i++) bbox = unite(bbox, face_box[order[i]]);
        }
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
        if (wide) {
            std::vector<Bin> part_bins((size_t)parts * kBins);
            parallel_chunks(lo, hi, parts, [&](int p, uint64_t a, uint64_t e) {
                Bin* mine = part_bins.data() + (size_t)p * kBins;
                for (int k = 0; k < kBins; k++) { mine[k].b = empty_box(); mine[k].count = 0; }
                for (uint64_t i = a; i < e; i++) {
                    const int bi = bin_of(centroid(order[i]));
                    bin_at[i] = (uint8_t)bi;
                    mine[bi].count++;
                    mine[bi].b = unite(mine[bi].b, face_box[order[i]]);
                }
            });
            for (int p = 0; p < parts; p++)
                for (int k = 0; k < kBins; k++) {
                    bins[k].count += part_bins[(size_t)p * kBins + k].count;
                    bins[k].b = unite(bins[k].b, part_bins[(size_t)p * kBins + k].b);
                }
        } else
        for (uint32_t i = lo; i < hi; i++) {
            int bi = bin_of(centroid(order[i]));
            bin_at[i] = (uint8_t)bi;
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
        } else if (wide) {  // the same order-preserving split: every chunk knows where its left and right shapes go
            std::vector<uint32_t> lefts((size_t)parts + 1, 0), rights((size_t)parts + 1, 0);
            parallel_chunks(lo, hi, parts, [&](int p, uint64_t a, uint64_t e) {
                uint32_t l = 0;
                for (uint64_t i = a; i < e; i++) l += bin_at[i] <= best_split ? 1u : 0u;
                lefts[(size_t)p + 1] = l;
                rights[(size_t)p + 1] = (uint32_t)(e - a) - l;
            });
            for (int p = 0; p < parts; p++) { lefts[(size_t)p + 1] += lefts[(size_t)p]; rights[(size_t)p + 1] += rights[(size_t)p]; }
            mid = lo + lefts[(size_t)parts];
            parallel_chunks(lo, hi, parts, [&](int p, uint64_t a, uint64_t e) {
                uint32_t l = lo + lefts[(size_t)p], r = mid + rights[(size_t)p];
                for (uint64_t i = a; i < e; i++) {
                    if (bin_at[i] <= best_split) scratch[l++] = order[i];
                    else scratch[r++] = order[i];
                }
            });
            parallel_chunks(lo, hi, parts, [&](int, uint64_t a, uint64_t e) { std::memcpy(order.data() + a, scratch.data() + a, (e - a) * sizeof(uint32_t)); });
        } else {  // shape_bvh.go:233-246: order-preserving split (the bins are those of the binning pass)
            uint32_t l = lo;
            for (uint32_t i = lo; i < hi; i++) l += bin_at[i] <= best_split ? 1u : 0u;
            mid = l;
            uint32_t r = mid;
            l = lo;
            for (uint32_t i = lo; i < hi; i++) {
                if (bin_at[i] <= best_split) scratch[l++] = order[i];
                else scratch[r++] = order[i];
            }
            std::memcpy(order.data() + lo, scratch.data() + lo, (size_t)n * sizeof(uint32_t));
        }
        // A bin's box is the union of its shapes' bounds and unions are exact, so the two sides of a bin split already have
        // theirs (what the reference recomputes at the top of the recursion, shape_bvh.go:110-114).
        const Box* left_box = best_split >= 0 ? &lb[best_split] : nullptr;
        const Box* right_box = best_split >= 0 ? &rb[best_split + 1] : nullptr;
        if (n > 32768 && depth < 5) {  // the reference forks two goroutines per split (shape_bvh.go:249-261)
            auto fut = std::async(std::launch::async, [&, lo, mid, depth, left_box] { return build(lo, mid, depth + 1, left_box); });
            node->right = build(mid, hi, depth + 1, right_box);
            node->left = fut.get();
        } else {
            node->left = build(lo, mid, depth + 1, left_box);
            node->right = build(mid, hi, depth + 1, right_box);
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

    const Box* face_box = nullptr;   // Face.Bounds() per face
    const uint32_t* order = nullptr; // leaf order -> face

    void set_box(FastNode& f, int k, const Box& b) {
        for (int a = 0; a < 3; a++) {
            f.b[k][a][0] = round_down(b.mn[a] - out.center[a]) - out.pad;
            f.b[k][a][1] = round_up(b.mx[a] - out.center[a]) + out.pad;
        }
    }
    // An empty slot gets an inverted box (min above max on every axis): a slab test that takes the planes in the order
    // the ray meets them finds its entry beyond its exit for every ray, so such a test needs no look at the reference.
    static void set_empty(FastNode& f, int k) {
        for (int a = 0; a < 3; a++) { f.b[k][a][0] = 1.0f; f.b[k][a][1] = -1.0f; }
        f.child[k] = kFastEmpty;
    }
    Box range_box(uint32_t first, uint32_t count) const {
        Box b = face_box[order[first]];
        for (uint32_t i = 1; i < count; i++) b = unite(b, face_box[order[first + i]]);
        return b;
    }
    // A leaf: the complement of (first face << leaf_bits | faces - 1).  3 bits (up to 8 faces) in the trees every walk reads;
    // the pixel beams' tree uses 4 (up to 16 faces; first face below 2^27).
    int leaf_bits = 3;
    int32_t leaf_ref(uint32_t first, uint32_t count) const { return ~(int32_t)((first << leaf_bits) | (count - 1)); }

    std::vector<FastNode>* dst = nullptr;  // the tree fast() writes: out.fast_nodes or out.beam_nodes
    uint32_t merge_faces = 0;              // > 0: a subtree of at most that many faces becomes one leaf

    // A run of prims that is too long for one leaf reference: halve it under extra nodes.
    int32_t fast_depth = 0, cur_depth = 0;
    struct DepthScope {
        Flattener& f;
        explicit DepthScope(Flattener& fl) : f(fl) { f.cur_depth++; f.fast_depth = std::max(f.fast_depth, f.cur_depth); }
        ~DepthScope() { f.cur_depth--; }
    };

    int32_t chunk(uint32_t first, uint32_t count) {
        if (count <= (1u << leaf_bits)) return leaf_ref(first, count);
        DepthScope scope(*this);
        const int32_t me = (int32_t)(*dst).size();
        (*dst).emplace_back();
        const uint32_t half = count / 2;
        const uint32_t f0[2] = {first, first + half}, c0[2] = {half, count - half};
        for (int k = 0; k < 2; k++) {
            const Box b = range_box(f0[k], c0[k]);
            const int32_t ref = never_hit(b) ? kFastEmpty : chunk(f0[k], c0[k]);
            FastNode f = (*dst)[me];
            set_box(f, k, b);
            f.child[k] = ref;
            if (ref == kFastEmpty) set_empty(f, k);
            (*dst)[me] = f;
        }
        return me;
    }

    // Faces of a subtree (a contiguous run in leaf order); false when there are more than `limit` or a box in it can never be entered.
    bool span(const TreeNode* c, uint32_t limit, uint32_t& first, uint32_t& count) const {
        if (never_hit(c->box)) return false;
        if (!c->left) {
            if (count == 0) first = c->first;
            count += c->count;
            return count <= limit;
        }
        return span(c->left.get(), limit, first, count) && span(c->right.get(), limit, first, count);
    }

    int32_t child_ref(const TreeNode* c) {
        if (never_hit(c->box)) return kFastEmpty;
        if (!c->left) return chunk(c->first, c->count);
        if (merge_faces > 0) {
            uint32_t first = 0, count = 0;
            if (span(c, merge_faces, first, count)) return leaf_ref(first, count);
        }
        return fast(c);
    }

    int32_t fast(const TreeNode* n) {  // n is an inner node
        DepthScope scope(*this);
        const int32_t me = (int32_t)(*dst).size();
        (*dst).emplace_back();
        const TreeNode* kids[2] = {n->left.get(), n->right.get()};
        for (int k = 0; k < 2; k++) {
            const int32_t ref = child_ref(kids[k]);  // may grow (reallocate) fast_nodes
            FastNode f = (*dst)[me];
            set_box(f, k, kids[k]->box);
            f.child[k] = ref;
            if (ref == kFastEmpty) set_empty(f, k);
            (*dst)[me] = f;
        }
        return me;
    }

    // The whole traversal tree under `root` into *dst.
    void tree(const TreeNode* root) {
        if (root->left) {
            fast(root);
        } else {  // the whole mesh is one leaf: give it a parent so traversal has a node to start from
            dst->emplace_back();
            const int32_t ref = child_ref(root);
            FastNode f = (*dst)[0];
            set_box(f, 0, root->box);
            f.child[0] = ref;
            if (ref == kFastEmpty) set_empty(f, 0);
            set_empty(f, 1);
            (*dst)[0] = f;
        }
    }
};

}  // namespace

void build_mesh_bvh(const rt_face* faces, uint64_t n_faces, MeshBvh& out, bool want_beam_tree) {
    auto t0 = std::chrono::steady_clock::now();
    auto t_last = t0;
    const bool debug = std::getenv("RT_DEBUG") != nullptr && n_faces > 100000;
    auto stage = [&](const char* what) {
        if (!debug) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[rt] build_mesh_bvh %-22s %7.1f ms\n", what, 1e3 * std::chrono::duration<double>(now - t_last).count());
        t_last = now;
    };
    const int threads = n_faces >= kWideFaces ? host_threads() : 1;
    std::unique_ptr<Box[]> boxes(new Box[n_faces]);  // not a vector: its pages are first touched by the threads that fill them
    parallel_chunks(0, n_faces, threads, [&](int, uint64_t a, uint64_t e) {
        for (uint64_t i = a; i < e; i++) {
            const rt_face& f = faces[i];
            for (int k = 0; k < 3; k++) {
                boxes[i].mn[k] = gmin(gmin(f.v[0][k], f.v[1][k]), f.v[2][k]);
                boxes[i].mx[k] = gmax(gmax(f.v[0][k], f.v[1][k]), f.v[2][k]);
            }
        }
    });
    stage("face bounds");
    Builder b;
    b.face_box = boxes.get();
    b.order.resize(n_faces);
    for (uint64_t i = 0; i < n_faces; i++) b.order[i] = (uint32_t)i;
    b.scratch.resize(n_faces);
    b.bin_at.resize(n_faces);
    std::unique_ptr<TreeNode> root = b.build(0, (uint32_t)n_faces, 0);
    stage("NewBVH");

    out.ref_nodes.clear();
    out.fast_nodes.clear();
    out.ref_nodes.reserve(n_faces);
    Flattener fl{out};
    // The reference-order nodes, the traversal trees and the faces in leaf order are written from the finished tree
    // independently of each other (different arrays, different members of the flattener): three threads.
    std::thread ref_thread([&] { fl.ref(root.get()); });

    double radius = 0;
    for (int k = 0; k < 3; k++) {
        out.root_box[k] = root->box.mn[k];
        out.root_box[3 + k] = root->box.mx[k];
        out.center[k] = 0.5 * (root->box.mn[k] + root->box.mx[k]);
        radius = std::max(radius, std::max(std::fabs(root->box.mn[k] - out.center[k]), std::fabs(root->box.mx[k] - out.center[k])));
    }
    out.pad = round_up(radius * kBoxPad);
    out.radius = (float)radius;
    std::thread prim_thread([&] {
    out.prim_face.resize(n_faces);
    out.prim_pos.resize(n_faces * 9);
    out.prim_tri32.resize(n_faces);
    parallel_chunks(0, n_faces, threads, [&](int, uint64_t pa, uint64_t pe) {
    for (uint64_t i = pa; i < pe; i++) {
        const uint32_t fi = b.order[i];
        out.prim_face[i] = (int32_t)fi;
        for (int v = 0; v < 3; v++)
            for (int k = 0; k < 3; k++) out.prim_pos[i * 9 + v * 3 + k] = faces[fi].v[v][k];
        Tri32& t = out.prim_tri32[i];
        double l1 = 0, l2 = 0;
        for (int k = 0; k < 3; k++) {
            const double e1 = faces[fi].v[1][k] - faces[fi].v[0][k], e2 = faces[fi].v[2][k] - faces[fi].v[0][k];
            t.p0[k] = (float)(faces[fi].v[0][k] - out.center[k]);
            t.e1[k] = (float)e1;
            t.e2[k] = (float)e2;
            l1 += e1 * e1;
            l2 += e2 * e2;
        }
        t.len1 = round_up(std::sqrt(l1));
        t.len2 = round_up(std::sqrt(l2));
        t.pad = 0;
    }
    });
    });
    fl.face_box = boxes.get();
    fl.order = b.order.data();
    fl.dst = &out.fast_nodes;
    out.fast_nodes.reserve(n_faces + 1);
    // the tree with folded leaves, when asked for: a flattener and a thread of its own
    out.beam_nodes.clear();
    Flattener fl_beam{out};
    fl_beam.face_box = boxes.get();
    fl_beam.order = b.order.data();
    fl_beam.dst = &out.beam_nodes;
    fl_beam.leaf_bits = RT_BEAM_LEAF_BITS;
    fl_beam.merge_faces = 1u << RT_BEAM_LEAF_BITS;  // what a leaf reference of that tree can hold
    std::thread beam_thread;
    if (want_beam_tree) {
        out.beam_nodes.reserve(n_faces / 4 + 1);
        beam_thread = std::thread([&] { fl_beam.tree(root.get()); });
    }
    fl.tree(root.get());

    stage("fast tree");
    // The quantised copy of the tree.  The grid spans the union of all child boxes (they carry the padding), widened a little.
    // (Not for a scene the pixel-beam tree was asked for: its material only emits, so no ray is ever scattered and the walk
    // that reads these nodes never runs.)
    out.q_nodes.clear();
    out.q4_nodes.clear();
    out.q4_depth = 1;
    for (int a = 0; a < 3; a++) { out.grid_lo[a] = 0; out.grid_step[a] = 1; }
    if (!want_beam_tree) {
        float lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
        bool any = false;
        for (const FastNode& f : out.fast_nodes)
            for (int k = 0; k < 2; k++) {
                if (f.child[k] == kFastEmpty) continue;
                for (int a = 0; a < 3; a++) {
                    lo[a] = any ? std::min(lo[a], f.b[k][a][0]) : f.b[k][a][0];
                    hi[a] = any ? std::max(hi[a], f.b[k][a][1]) : f.b[k][a][1];
                }
                any = true;
            }
        for (int a = 0; a < 3; a++) {
            const float ext = std::max(hi[a] - lo[a], 1e-30f);
            out.grid_lo[a] = lo[a] - 4.0f * ext / 65535.0f;
            out.grid_step[a] = round_up((double)(hi[a] + 4.0f * ext / 65535.0f - out.grid_lo[a]) / 65520.0);  // every plane lands inside [8, 65528)
        }
        out.q_nodes.resize(out.fast_nodes.size());
        parallel_chunks(0, out.fast_nodes.size(), threads, [&](int, uint64_t qa, uint64_t qe) {
        for (size_t i = qa; i < qe; i++) {
            const FastNode& f = out.fast_nodes[i];
            QNode& qn = out.q_nodes[i];
            for (int k = 0; k < 2; k++) {
                qn.child[k] = f.child[k];
                for (int a = 0; a < 3; a++) {
                    if (f.child[k] == kFastEmpty) { qn.q[k][a][0] = 65535; qn.q[k][a][1] = 0; continue; }
                    // float64 here: the division is exact enough that one extra step on each side covers its rounding and
                    // that of the device's float32 plane * step + offset
                    const double pmin = ((double)f.b[k][a][0] - (double)out.grid_lo[a]) / (double)out.grid_step[a];
                    const double pmax = ((double)f.b[k][a][1] - (double)out.grid_lo[a]) / (double)out.grid_step[a];
                    qn.q[k][a][0] = (uint16_t)std::max(0.0, std::floor(pmin) - 1.0);
                    qn.q[k][a][1] = (uint16_t)std::min(65535.0, std::ceil(pmax) + 1.0);
                }
            }
        }
        });
    }

    stage("quantised nodes");
    // Two levels per node.  Built without recursion (a work list of (binary node, slot to patch)): trees of scans are deep.
    if (!want_beam_tree) {
        out.q4_nodes.clear();
        out.q4_nodes.reserve(out.q_nodes.size() / 2 + 1);
        struct Todo { int32_t bin; int32_t parent; int slot; int32_t depth; };
        std::vector<Todo> todo;
        todo.push_back({0, -1, 0, 1});
        out.q4_depth = 1;
        while (!todo.empty()) {
            const Todo t = todo.back();
            todo.pop_back();
            const int32_t me = (int32_t)out.q4_nodes.size();
            out.q4_nodes.emplace_back();
            if (t.parent >= 0) out.q4_nodes[(size_t)t.parent].child[t.slot] = me;
            out.q4_depth = std::max(out.q4_depth, t.depth);
            Q4Node node;
            int n = 0;
            auto add = [&](const uint16_t (&q)[3][2], int32_t ref) {
                std::memcpy(node.q[n], q, sizeof(node.q[n]));
                node.child[n] = ref;
                n++;
            };
            auto area = [&](int i) {
                const double dx = (double)node.q[i][0][1] - node.q[i][0][0], dy = (double)node.q[i][1][1] - node.q[i][1][0],
                             dz = (double)node.q[i][2][1] - node.q[i][2][0];
                return dx * dy + dy * dz + dz * dx;
            };
            {
                const QNode& b = out.q_nodes[(size_t)t.bin];
                for (int k = 0; k < 2; k++)
                    if (b.child[k] != kFastEmpty) add(b.q[k], b.child[k]);
            }
            // open the largest inner slot until the node is full (or holds leaves only): the big boxes are the ones most
            // rays enter anyway, so their children belong one level up
            for (;;) {
                int pick = -1;
                for (int i = 0; i < n; i++)
                    if (node.child[i] >= 0 && (pick < 0 || area(i) > area(pick))) pick = i;
                if (pick < 0) break;
                const QNode& g = out.q_nodes[(size_t)node.child[pick]];
                const int kids = (g.child[0] != kFastEmpty) + (g.child[1] != kFastEmpty);
                if (n - 1 + kids > 4) break;
                // the picked slot is replaced by the first child, the second (if any) goes to the end
                bool placed = false;
                for (int j = 0; j < 2; j++) {
                    if (g.child[j] == kFastEmpty) continue;
                    if (!placed) {
                        std::memcpy(node.q[pick], g.q[j], sizeof(node.q[pick]));
                        node.child[pick] = g.child[j];
                        placed = true;
                    } else {
                        add(g.q[j], g.child[j]);
                    }
                }
                if (!placed) {  // an inner node without children cannot be reached by any ray: drop the slot
                    n--;
                    std::memcpy(node.q[pick], node.q[n], sizeof(node.q[pick]));
                    node.child[pick] = node.child[n];
                }
            }
            for (int i = n; i < 4; i++) {
                for (int a = 0; a < 3; a++) { node.q[i][a][0] = 65535; node.q[i][a][1] = 0; }
                node.child[i] = kFastEmpty;
            }
            for (int i = 0; i < n; i++)
                if (node.child[i] >= 0) {
                    todo.push_back({node.child[i], me, i, t.depth + 1});
                    node.child[i] = kFastEmpty;  // patched when the child is made
                }
            out.q4_nodes[(size_t)me] = node;
        }
    }

    stage("four-wide nodes");
    out.fast_depth = std::max(fl.fast_depth, 1);
    prim_thread.join();
    ref_thread.join();
    if (beam_thread.joinable()) beam_thread.join();
    stage("joined the other two");
    out.info.n_faces = n_faces;
    out.info.n_nodes = out.ref_nodes.size();
    out.info.n_leaves = fl.leaves;
    out.info.max_depth = fl.max_depth;
    out.info.max_leaf_size = fl.max_leaf;
    out.info.build_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

}  // namespace rt
