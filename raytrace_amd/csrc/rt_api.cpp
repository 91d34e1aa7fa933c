// C ABI of librt_hip.so (include/rt_hip.h): scene upload, the wavefront render loop and the
// batch entry points.  Host code only; the kernels live in kernels.hip.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "bvh_build.h"
#include "dev_math.h"
#include "host_common.h"

namespace rt {

thread_local std::string tl_error;
std::atomic<bool> g_initialized{false};
std::vector<int> g_devices;
static bool g_peer_ok[64][64];  // [from][to], by device ordinal; written by rt_init only
bool peer_copy_allowed(int from, int to) { return from >= 0 && to >= 0 && from < 64 && to < 64 && g_peer_ok[from][to]; }
std::atomic<uint64_t> g_scene_serial{0};

int fail(int code, const std::string& msg) {
    tl_error = msg;
    return code;
}
const std::string& last_error_text() { return tl_error; }

}  // namespace rt

namespace {

using namespace rt;

// ---- host-side precomputation, same float64 operations as the reference's per-call set-up ----

DQuad make_quad(const rt_quad& q) {  // shape_quad.go:40-64, 90-97
    DQuad d;
    V3 nrm = unit(ld3(q.normal));
    V3 arbitrary = std::fabs(nrm.x) < 0.9 ? mk(1, 0, 0) : mk(0, 1, 0);
    V3 u = unit(cross(nrm, arbitrary));
    V3 v = unit(cross(nrm, u));
    u = u * (q.width / 2);
    v = v * (q.height / 2);
    V3 c = ld3(q.center);
    V3 p0 = (c - u) - v, p1 = (c + u) - v, p2 = (c + u) + v, p3 = (c - u) + v;
    V3 lu = p2 - p1, lv = p0 - p1;
    st3(d.nrm, nrm);
    st3(d.p0, p0); st3(d.p1, p1); st3(d.p2, p2); st3(d.p3, p3);
    st3(d.lu, lu); st3(d.lv, lv);
    d.lulu = dot(lu, lu);
    d.lvlv = dot(lv, lv);
    return d;
}

DCylinder make_cylinder(const rt_cylinder& c) {  // shape_cylinder.go:41, 100
    DCylinder d;
    V3 dir = unit(ld3(c.direction));
    st3(d.o, ld3(c.origin));
    st3(d.d, dir);
    st3(d.top, ld3(c.origin) + dir * c.height);
    d.r = c.radius;
    d.h = c.height;
    return d;
}

DXform make_xform(const double* tr, const double* rot, const double* sc) {
    DXform x;
    std::memcpy(x.tr, tr, sizeof(x.tr));
    std::memcpy(x.rot, rot, sizeof(x.rot));
    std::memcpy(x.sc, sc, sizeof(x.sc));
    return x;
}

DXform invert(const DXform& t) {  // Transform.Inverse, transform.go:86-102
    DXform r;
    V3 inv_scale = mk(1 / t.sc[0], 1 / t.sc[1], 1 / t.sc[2]);
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) r.rot[i * 3 + j] = t.rot[j * 3 + i];
    V3 nt = ld3(t.tr) * -1.0;
    V3 rt_ = mk(r.rot[0] * nt.x + r.rot[1] * nt.y + r.rot[2] * nt.z, r.rot[3] * nt.x + r.rot[4] * nt.y + r.rot[5] * nt.z,
                r.rot[6] * nt.x + r.rot[7] * nt.y + r.rot[8] * nt.z);
    st3(r.tr, rt_ * inv_scale);
    st3(r.sc, inv_scale);
    return r;
}

}  // namespace

int rt::make_camera(const rt_camera_desc& c, DCamera& d) {
    std::memset(&d, 0, sizeof(d));
    d.kind = c.kind;
    d.intr = c;
    if (c.kind == RT_CAM_PINHOLE) {
        std::memcpy(d.llc, c.lower_left_corner, sizeof(d.llc));
        std::memcpy(d.origin, c.origin, sizeof(d.origin));
        std::memcpy(d.horizontal, c.horizontal, sizeof(d.horizontal));
        std::memcpy(d.vertical, c.vertical, sizeof(d.vertical));
        return RT_OK;
    }
    if (c.kind != RT_CAM_FOCUSABLE && c.kind != RT_CAM_ORTHOGRAPHIC && c.kind != RT_CAM_CALIBRATED)
        return fail(RT_ERR_INVALID, "unknown camera kind");
    V3 from = ld3(c.look_from), at = ld3(c.look_at);
    V3 w = unit(from - at);             // camera_orthographic.go:26-28, camera_focusable.go:28-30,
    V3 u = unit(cross(ld3(c.vup), w));  // camera_calibrated.go:47-49
    V3 v = cross(w, u);
    st3(d.from, from);
    st3(d.u, u); st3(d.v, v); st3(d.w, w);
    st3(d.dir, unit(at - from));
    d.fov_w = c.fov_width;
    d.fov_h = c.fov_height;
    if (c.kind == RT_CAM_FOCUSABLE) {  // camera_focusable.go:33-43
        V3 horizontal = u * (c.fov_width * c.working_distance);
        V3 vertical = v * (c.fov_height * c.working_distance);
        V3 llc = ((from - horizontal / 2.0) - vertical / 2.0) - w * c.working_distance;
        st3(d.horizontal, horizontal);
        st3(d.vertical, vertical);
        st3(d.llc, llc);
        d.lens_radius = c.aperture / 2;
    }
    if (c.kind == RT_CAM_CALIBRATED && (c.width <= 0 || c.height <= 0 || !(c.fx > 0 && c.fy > 0)))
        return fail(RT_ERR_INVALID, "CameraIntrinsics bad image size or focal lengths");
    return RT_OK;
}

namespace {

int check_shape_ref(const rt_scene_desc& d, int kind, int index, int depth) {
    const uint64_t counts[6] = {d.n_spheres, d.n_quads, d.n_cylinders, d.n_triangles, d.n_meshes, d.n_transformed};
    if (kind < 0 || kind > 5 || index < 0 || (uint64_t)index >= counts[kind]) return fail(RT_ERR_INVALID, "shape reference out of range");
    if (kind == RT_SHAPE_TRANSFORMED) {
        if (depth >= 4) return fail(RT_ERR_UNSUPPORTED, "TransformedShape nested deeper than 4");
        return check_shape_ref(d, d.transformed[index].child_kind, d.transformed[index].child_index, depth + 1);
    }
    return RT_OK;
}

}  // namespace

rt_scene::~rt_scene() {
    if (replicas.empty()) return;
    (void)hipSetDevice(replicas[0]->device);
    for (auto& rep : replicas)
        if (rep->inbox) (void)hipFree(rep->inbox);  // the inboxes of all replicas live on the first device
    for (auto& rep : replicas) {
        (void)hipSetDevice(rep->device);
        rt::destroy_work(rep->work);
        if (rep->film) (void)hipFree(rep->film);
        if (rep->staging) (void)hipHostFree(rep->staging);
        if (rep->stream) (void)hipStreamDestroy(rep->stream);
        if (rep->done) (void)hipEventDestroy(rep->done);
        rep.reset();  // frees the scene arrays with this replica's device current
    }
    if (!rt::g_devices.empty()) (void)hipSetDevice(rt::g_devices[0]);
}

extern "C" {

const char* rt_last_error(void) { return rt::tl_error.c_str(); }

int rt_init(const int* gpu_ids, int n) {
    int count = 0;
    HIP_TRY(hipGetDeviceCount(&count));
    if (count <= 0) return fail(RT_ERR_DEVICE, "no HIP device visible");
    std::vector<int> devices;
    if (gpu_ids && n > 0) devices.assign(gpu_ids, gpu_ids + n);
    else devices.push_back(0);
    if (devices.size() > 64) return fail(RT_ERR_INVALID, "more than 64 devices");
    const bool no_peer = std::getenv("RT_NO_PEER") != nullptr, verbose = std::getenv("RT_DEBUG") != nullptr;
    for (int dev : devices) {
        if (dev < 0 || dev >= count || dev >= 64) return fail(RT_ERR_INVALID, "device ordinal out of range");
        hipDeviceProp_t prop;
        HIP_TRY(hipGetDeviceProperties(&prop, dev));
        if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0)
            return fail(RT_ERR_UNSUPPORTED, std::string("this library is built for gfx950 only, device is ") + prop.gcnArchName);
    }
    for (int dev : devices) {
        HIP_TRY(hipSetDevice(dev));
        // the static scratch of rt_render_film_noalloc: its address on this device is looked up now, so that the
        // entry point itself only reads it (the look-up may load the code object, which allocates)
        void* scratch_ptr = nullptr;
        HIP_TRY(static_scratch(dev, &scratch_ptr));
        for (int other : devices) {  // films travel between the devices: direct copies where the links allow them, through the host where not
            bool ok = !no_peer;
            if (ok && other != dev) {
                int can = 0;
                if (hipDeviceCanAccessPeer(&can, dev, other) != hipSuccess || !can) {
                    ok = false;
                } else {
                    const hipError_t e = hipDeviceEnablePeerAccess(other, 0);
                    if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) ok = false;
                }
                (void)hipGetLastError();
            }
            g_peer_ok[dev][other] = ok;
            if (verbose && other != dev) fprintf(stderr, "[rt] rt_init: device %d -> %d: %s\n", dev, other, ok ? "direct copies" : "no peer access, films are staged through pinned host memory");
        }
    }
    HIP_TRY(hipSetDevice(devices[0]));
    g_devices = devices;
    g_initialized = true;
    return RT_OK;
}

void rt_shutdown(void) { g_initialized = false; }

namespace {

// Conservative world-space bounds of a shape for the fast mode's per-node culling.  They only have to
// contain every point the shape's Collide can return, tolerances included (the triangle tests accept
// barycentrics up to 1e-6 outside the triangle), so they are padded generously afterwards.
struct Bounds {
    double lo[3], hi[3];
    void grow(V3 p) {
        const double c[3] = {p.x, p.y, p.z};
        for (int a = 0; a < 3; a++) { lo[a] = std::min(lo[a], c[a]); hi[a] = std::max(hi[a], c[a]); }
    }
};

Bounds empty_bounds() {
    const double inf = std::numeric_limits<double>::infinity();
    return Bounds{{inf, inf, inf}, {-inf, -inf, -inf}};
}

Bounds shape_bounds(int kind, int index, const std::vector<DSphere>& spheres, const std::vector<DQuad>& quads,
                    const std::vector<DCylinder>& cyls, const std::vector<DTri>& tris, const std::vector<DTransformed>& xforms,
                    const std::vector<DMesh>& meshes, int depth = 0) {
    Bounds b = empty_bounds();
    const double inf = std::numeric_limits<double>::infinity();
    const Bounds everything{{-inf, -inf, -inf}, {inf, inf, inf}};
    switch (kind) {
        case RT_SHAPE_SPHERE: {
            const DSphere& s = spheres[index];
            b.grow(mk(s.c[0] - s.r, s.c[1] - s.r, s.c[2] - s.r));
            b.grow(mk(s.c[0] + s.r, s.c[1] + s.r, s.c[2] + s.r));
            break;
        }
        case RT_SHAPE_QUAD: {
            const DQuad& q = quads[index];
            b.grow(ld3(q.p0)); b.grow(ld3(q.p1)); b.grow(ld3(q.p2)); b.grow(ld3(q.p3));
            break;
        }
        case RT_SHAPE_CYLINDER: {
            const DCylinder& c = cyls[index];
            for (const double* e : {c.o, c.top}) {
                b.grow(mk(e[0] - c.r, e[1] - c.r, e[2] - c.r));
                b.grow(mk(e[0] + c.r, e[1] + c.r, e[2] + c.r));
            }
            break;
        }
        case RT_SHAPE_TRIANGLE: {
            const DTri& t = tris[index];
            b.grow(ld3(t.p0)); b.grow(ld3(t.p1)); b.grow(ld3(t.p2));
            break;
        }
        case RT_SHAPE_MESH: {
            const DMesh& m = meshes[index];
            b.grow(ld3(m.root_box)); b.grow(ld3(m.root_box + 3));
            break;
        }
        case RT_SHAPE_TRANSFORMED: {
            if (depth >= 4) return everything;
            const DTransformed& x = xforms[index];
            const Bounds c = shape_bounds(x.child_kind, x.child_index, spheres, quads, cyls, tris, xforms, meshes, depth + 1);
            // A world ray is taken into the child's frame by x.inv (shape_transformed_shape.go:29-35), so the region of
            // world space where the child can be hit is the pre-image of its bounds under x.inv -- which is x.fwd's
            // image only for uniform scales, because Transform.Inverse (transform.go:86-102) is not the exact inverse
            // otherwise.  Invert x.inv's matrix R * diag(S) numerically.
            double m[9], im[9];
            for (int r = 0; r < 3; r++)
                for (int k = 0; k < 3; k++) m[r * 3 + k] = x.inv.rot[r * 3 + k] * x.inv.sc[k];
            const double det = m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) + m[2] * (m[3] * m[7] - m[4] * m[6]);
            if (!(std::fabs(det) > 1e-300) || !std::isfinite(det)) return everything;
            im[0] = (m[4] * m[8] - m[5] * m[7]) / det; im[1] = (m[2] * m[7] - m[1] * m[8]) / det; im[2] = (m[1] * m[5] - m[2] * m[4]) / det;
            im[3] = (m[5] * m[6] - m[3] * m[8]) / det; im[4] = (m[0] * m[8] - m[2] * m[6]) / det; im[5] = (m[2] * m[3] - m[0] * m[5]) / det;
            im[6] = (m[3] * m[7] - m[4] * m[6]) / det; im[7] = (m[1] * m[6] - m[0] * m[7]) / det; im[8] = (m[0] * m[4] - m[1] * m[3]) / det;
            for (int k = 0; k < 8; k++) {
                const V3 p = mk((k & 1) ? c.hi[0] : c.lo[0], (k & 2) ? c.hi[1] : c.lo[1], (k & 4) ? c.hi[2] : c.lo[2]) - ld3(x.inv.tr);
                if (!std::isfinite(p.x) || !std::isfinite(p.y) || !std::isfinite(p.z)) return everything;
                b.grow(mk(im[0] * p.x + im[1] * p.y + im[2] * p.z, im[3] * p.x + im[4] * p.y + im[5] * p.z, im[6] * p.x + im[7] * p.y + im[8] * p.z));
            }
            break;
        }
        default: return everything;
    }
    double ext = 0, mag = 0;
    for (int a = 0; a < 3; a++) {
        if (!(b.lo[a] <= b.hi[a]) || !std::isfinite(b.lo[a]) || !std::isfinite(b.hi[a])) return everything;
        ext = std::max(ext, b.hi[a] - b.lo[a]);
        mag = std::max(mag, std::max(std::fabs(b.lo[a]), std::fabs(b.hi[a])));
    }
    const double pad = 1e-4 * ext + 1e-9 * mag + 1e-300;
    for (int a = 0; a < 3; a++) { b.lo[a] -= pad; b.hi[a] += pad; }
    return b;
}

}  // namespace

int rt_scene_create(const rt_scene_desc* d, rt_scene** out) {
    if (!g_initialized) return fail(RT_ERR_NOT_INITIALIZED, "rt_init has not been called");
    if (!d || !out) return fail(RT_ERR_INVALID, "null argument");
    // the array-level checks of Scene.Validate (scene.go:29-79) and of each Validate() it calls
    for (uint64_t i = 0; i < d->n_nodes; i++) {
        const rt_node& n = d->nodes[i];
        if (int rc = check_shape_ref(*d, n.shape_kind, n.shape_index, 0)) return rc;
        if (n.material < 0 || (uint64_t)n.material >= d->n_materials) return fail(RT_ERR_INVALID, "node " + std::to_string(i) + " has no material");
    }
    for (uint64_t i = 0; i < d->n_transformed; i++)
        if (int rc = check_shape_ref(*d, RT_SHAPE_TRANSFORMED, (int)i, 0)) return rc;
    for (uint64_t i = 0; i < d->n_spheres; i++)
        if (!(d->spheres[i].radius > 0)) return fail(RT_ERR_INVALID, "invalid Sphere radius");
    for (uint64_t i = 0; i < d->n_quads; i++)
        if (!(d->quads[i].width > 0) || !(d->quads[i].height > 0)) return fail(RT_ERR_INVALID, "invalid Quad size");
    for (uint64_t i = 0; i < d->n_cylinders; i++)
        if (!(d->cylinders[i].radius > 0) || !(d->cylinders[i].height > 0)) return fail(RT_ERR_INVALID, "invalid Cylinder size");
    for (uint64_t i = 0; i < d->n_meshes; i++) {
        const rt_mesh& m = d->meshes[i];
        if (m.n_faces == 0) return fail(RT_ERR_INVALID, "Mesh must contain at least one face");
        if (m.first_face + m.n_faces > d->n_faces || m.n_faces > 0x7fffffffull) return fail(RT_ERR_INVALID, "mesh face range out of bounds");
        if (m.n_faces >= ((uint64_t)1 << 28) - 16) return fail(RT_ERR_UNSUPPORTED, "meshes above 2^28 faces are not supported");
    }
    for (uint64_t i = 0; i < d->n_materials; i++) {
        const rt_material& m = d->materials[i];
        if (m.kind < RT_MAT_LAMBERTIAN || m.kind > RT_MAT_DEBUG_NORMAL) return fail(RT_ERR_INVALID, "unknown material kind");
        if ((m.kind == RT_MAT_LAMBERTIAN || m.kind == RT_MAT_EMITTER) && (m.texture < 0 || (uint64_t)m.texture >= d->n_textures))
            return fail(RT_ERR_INVALID, "material texture index out of range");
        if (m.kind == RT_MAT_METAL && (m.fuzz < 0 || m.fuzz > 1)) return fail(RT_ERR_INVALID, "invalid Metal fuzz must be in the range [0, 1]");
        if (m.kind == RT_MAT_DIELECTRIC && (m.ior_interior < 1 || m.ior_exterior < 1 || m.roughness < 0 || m.roughness > 1))
            return fail(RT_ERR_INVALID, "invalid Dielectric parameters");
    }
    for (uint64_t i = 0; i < d->n_textures; i++) {
        const rt_texture& t = d->textures[i];
        if (t.kind == RT_TEX_CHECKERBOARD) {
            if (t.odd < 0 || t.even < 0 || (uint64_t)t.odd >= d->n_textures || (uint64_t)t.even >= d->n_textures)
                return fail(RT_ERR_INVALID, "TextureCheckerboard child out of range");
            if (!(t.frequency > 0)) return fail(RT_ERR_INVALID, "error TextureChecker.Frequency is negative");
        } else if (t.kind == RT_TEX_IMAGE) {
            if (t.image < 0 || (uint64_t)t.image >= d->n_images) return fail(RT_ERR_INVALID, "image texture is nil");
        } else if (t.kind != RT_TEX_UNIFORM) {
            return fail(RT_ERR_INVALID, "unknown texture kind");
        }
    }
    for (uint64_t i = 0; i < d->n_images; i++) {
        const rt_image& im = d->images[i];
        if (im.width <= 0 || im.height <= 0 || !im.pixels || (im.format != RT_IMG_NRGBA8 && im.format != RT_IMG_RGBA16))
            return fail(RT_ERR_INVALID, "bad image");
    }
    for (uint64_t i = 0; i < d->n_lights; i++)
        if (d->lights[i].intensity[0] < 0 || d->lights[i].intensity[1] < 0 || d->lights[i].intensity[2] < 0)
            return fail(RT_ERR_INVALID, "invalid PointLight RadiantIntensity (should be non-negative)");
    // the light loop's launch has one grid row per (light, queue stripe), and a grid has at most 65535 rows
    if (d->n_nodes > 0x7fffffffull || d->n_lights > 65535 / kQueueStripes)
        return fail(RT_ERR_UNSUPPORTED, "too many nodes or lights (at most " + std::to_string(65535 / kQueueStripes) + " point lights)");

    std::unique_ptr<rt_scene> sc(new rt_scene());
    sc->serial = ++g_scene_serial;
    std::vector<DNode> nodes(d->n_nodes);
    for (uint64_t i = 0; i < d->n_nodes; i++) nodes[i] = {d->nodes[i].shape_kind, d->nodes[i].shape_index, d->nodes[i].material, 0};
    std::vector<DSphere> spheres(d->n_spheres);
    for (uint64_t i = 0; i < d->n_spheres; i++) {
        std::memcpy(spheres[i].c, d->spheres[i].center, sizeof(double) * 3);
        spheres[i].r = d->spheres[i].radius;
        spheres[i].uvmap = d->spheres[i].uvmap;
        spheres[i].pad = 0;
    }
    std::vector<DQuad> quads(d->n_quads);
    for (uint64_t i = 0; i < d->n_quads; i++) quads[i] = make_quad(d->quads[i]);
    std::vector<DCylinder> cyls(d->n_cylinders);
    for (uint64_t i = 0; i < d->n_cylinders; i++) cyls[i] = make_cylinder(d->cylinders[i]);
    std::vector<DTri> tris(d->n_triangles);
    for (uint64_t i = 0; i < d->n_triangles; i++) {
        const rt_triangle& t = d->triangles[i];
        if (t.kind < RT_TRI_TRIANGLE || t.kind > RT_TRI_FACE) return fail(RT_ERR_INVALID, "unknown triangle kind");
        std::memcpy(tris[i].p0, t.p[0], 24); std::memcpy(tris[i].p1, t.p[1], 24); std::memcpy(tris[i].p2, t.p[2], 24);
        std::memcpy(tris[i].uv, t.uv, 48);
        st3(tris[i].n, unit(ld3(t.normal)));
        tris[i].kind = t.kind;
        tris[i].pad = 0;
    }
    std::vector<DTransformed> xforms(d->n_transformed);
    for (uint64_t i = 0; i < d->n_transformed; i++) {
        const rt_transformed& t = d->transformed[i];
        xforms[i].child_kind = t.child_kind;
        xforms[i].child_index = t.child_index;
        xforms[i].fwd = make_xform(t.translation, t.rotation, t.scale);
        xforms[i].inv = invert(xforms[i].fwd);
    }
    // meshes: build every BVH, concatenate the per-mesh arrays
    std::vector<DMesh> meshes(d->n_meshes);
    std::vector<RefNode> ref_nodes;
    std::vector<FastNode> fast_nodes;
    std::vector<QNode> q_nodes;
    std::vector<Q4Node> q4_nodes;
    int32_t walk_cap = 2, deepest_fast = 0;
    std::vector<double> prim_pos;
    std::vector<Tri32> prim_tri32;
    std::vector<int32_t> prim_face;
    std::vector<double> face_uv(d->n_faces * 6);
    for (uint64_t i = 0; i < d->n_faces; i++)
        for (int v = 0; v < 3; v++) { face_uv[i * 6 + v * 2] = d->faces[i].v[v][3]; face_uv[i * 6 + v * 2 + 1] = d->faces[i].v[v][4]; }
    sc->bvh_info.resize(d->n_meshes);
    int32_t stack_pairs = 2;
    // What the pixel-beam kernel renders: one untransformed mesh whose material only emits (the camera decides per frame)
    bool beam_scene = d->n_nodes == 1 && d->n_meshes == 1 && nodes[0].kind == RT_SHAPE_MESH && std::getenv("RT_NO_BEAM") == nullptr;
    if (beam_scene) {
        const int k = d->materials[nodes[0].material].kind;
        beam_scene = k == RT_MAT_EMITTER || k == RT_MAT_DEBUG_UV || k == RT_MAT_DEBUG_NORMAL;
    }
    for (uint64_t i = 0; i < d->n_meshes; i++) {
        MeshBvh b;
        build_mesh_bvh(d->faces + d->meshes[i].first_face, d->meshes[i].n_faces, b, beam_scene);
        DMesh& m = meshes[i];
        m.ref_off = (int64_t)ref_nodes.size();
        m.fast_off = (int64_t)fast_nodes.size();
        m.q4_off = (int64_t)q4_nodes.size();
        m.prim_off = (int64_t)prim_face.size();
        m.face_off = (int64_t)d->meshes[i].first_face;
        std::memcpy(m.center, b.center, sizeof(m.center));
        std::memcpy(m.root_box, b.root_box, sizeof(m.root_box));
        m.pad = b.pad;
        m.radius = b.radius;
        m.pad2 = 0;
        std::memcpy(m.grid_lo, b.grid_lo, sizeof(m.grid_lo));
        std::memcpy(m.grid_step, b.grid_step, sizeof(m.grid_step));
        m.n_faces = (int32_t)d->meshes[i].n_faces;
        // the first mesh hands its arrays over (a scan is one mesh of hundreds of megabytes), the others are appended
        auto take = [](auto& all, auto& part) {
            if (all.empty()) all = std::move(part);
            else all.insert(all.end(), part.begin(), part.end());
        };
        take(ref_nodes, b.ref_nodes);
        take(fast_nodes, b.fast_nodes);
        take(q_nodes, b.q_nodes);
        take(q4_nodes, b.q4_nodes);
        m.beam_off = m.fast_off;
        if (!b.beam_nodes.empty()) {  // behind the mesh's own nodes; its child references are relative to beam_off
            m.beam_off = (int64_t)fast_nodes.size();
            fast_nodes.insert(fast_nodes.end(), b.beam_nodes.begin(), b.beam_nodes.end());
        }
        walk_cap = std::max(walk_cap, std::max(b.fast_depth + 1, 3 * b.q4_depth + 1));
        deepest_fast = std::max(deepest_fast, (int32_t)b.fast_depth);
        take(prim_pos, b.prim_pos);
        take(prim_tri32, b.prim_tri32);
        take(prim_face, b.prim_face);
        sc->bvh_info[i] = b.info;
        stack_pairs = std::max(stack_pairs, b.fast_depth + 1);
    }
    std::vector<double> node_box(d->n_nodes * 6);
    for (uint64_t i = 0; i < d->n_nodes; i++) {
        const Bounds b = shape_bounds(nodes[i].kind, nodes[i].index, spheres, quads, cyls, tris, xforms, meshes);
        std::memcpy(&node_box[i * 6], b.lo, 24);
        std::memcpy(&node_box[i * 6 + 3], b.hi, 24);
    }
    // Packed float32 bounds, two nodes per entry (see DScene::node_pairs).  The frame is the union of the "ordinary" nodes:
    // those whose bounds are finite and at most 64 times as large as the median node's.
    std::vector<NodePair32> node_pairs;
    double pair_center[3] = {0, 0, 0}, pair_radius = 0;
    {
        const float inf = std::numeric_limits<float>::infinity();
        std::vector<double> sizes;
        std::vector<char> ordinary(d->n_nodes, 0);
        auto size_of = [&](uint64_t i) {
            double m = 0;
            for (int k = 0; k < 3; k++) m = std::max(m, node_box[i * 6 + 3 + k] - node_box[i * 6 + k]);
            return m;
        };
        for (uint64_t i = 0; i < d->n_nodes; i++) {
            const double sz = size_of(i);
            if (nodes[i].kind != RT_SHAPE_MESH && std::isfinite(sz) && sz > 0) sizes.push_back(sz);
        }
        if (!sizes.empty()) {
            std::nth_element(sizes.begin(), sizes.begin() + sizes.size() / 2, sizes.end());
            const double median = sizes[sizes.size() / 2];
            double lo[3], hi[3];
            bool any = false;
            for (uint64_t i = 0; i < d->n_nodes; i++) {
                const double sz = size_of(i);
                if (nodes[i].kind == RT_SHAPE_MESH || !std::isfinite(sz) || !(sz > 0) || sz > 64.0 * median) continue;
                bool finite = true;
                for (int k = 0; k < 6; k++) finite = finite && std::isfinite(node_box[i * 6 + k]);
                if (!finite) continue;
                ordinary[i] = 1;
                for (int k = 0; k < 3; k++) {
                    lo[k] = any ? std::min(lo[k], node_box[i * 6 + k]) : node_box[i * 6 + k];
                    hi[k] = any ? std::max(hi[k], node_box[i * 6 + 3 + k]) : node_box[i * 6 + 3 + k];
                }
                any = true;
            }
            if (any) {
                for (int k = 0; k < 3; k++) {
                    pair_center[k] = 0.5 * (lo[k] + hi[k]);
                    pair_radius = std::max(pair_radius, 0.5 * (hi[k] - lo[k]));
                }
            }
            if (any && pair_radius > 0 && std::isfinite(pair_radius)) {
                const double pad = pair_radius / 65536.0;
                // A mesh that lies within 16 radii of the frame gets its real bounds too (float32 rounding of the slab
                // products then moves a plane by at most 2^-20 of the radius): a ray that misses them is spared the mesh's
                // own float64 clip, which is what most shadow rays and most camera rays of a small mesh on a large floor pay.
                for (uint64_t i = 0; i < d->n_nodes; i++) {
                    if (nodes[i].kind != RT_SHAPE_MESH) continue;
                    bool near = true;
                    for (int k = 0; k < 3; k++)
                        near = near && std::isfinite(node_box[i * 6 + k]) && std::isfinite(node_box[i * 6 + 3 + k]) &&
                               std::fabs(node_box[i * 6 + k] - pair_center[k]) <= 16.0 * pair_radius &&
                               std::fabs(node_box[i * 6 + 3 + k] - pair_center[k]) <= 16.0 * pair_radius;
                    if (near) ordinary[i] = 1;
                }
                node_pairs.resize((d->n_nodes + 1) / 2);
                for (uint64_t i = 0; i < 2 * node_pairs.size(); i++) {
                    NodePair32& np = node_pairs[i / 2];
                    const int w = (int)(i & 1);
                    for (int k = 0; k < 3; k++) {
                        if (i >= d->n_nodes) { np.lo[k][w] = inf; np.hi[k][w] = -inf; continue; }  // the odd one out: never entered
                        if (!ordinary[i]) { np.lo[k][w] = -inf; np.hi[k][w] = inf; continue; }
                        const double a = node_box[i * 6 + k] - pair_center[k] - pad, b = node_box[i * 6 + 3 + k] - pair_center[k] + pad;
                        float fa = (float)a, fb = (float)b;
                        if ((double)fa > a) fa = std::nextafterf(fa, -inf);
                        if ((double)fb < b) fb = std::nextafterf(fb, inf);
                        np.lo[k][w] = fa;
                        np.hi[k][w] = fb;
                    }
                }
            }
        }
    }
    std::vector<rt_material> materials(d->materials, d->materials + d->n_materials);
    std::vector<rt_texture> textures(d->textures, d->textures + d->n_textures);
    std::vector<rt_light> lights(d->lights, d->lights + d->n_lights);
    sc->node_box = node_box;
    sc->n_materials = materials.size();
    sc->meshes = meshes;
    sc->fast_nodes = fast_nodes;
    sc->all_terminal = true;
    for (const rt_material& m : materials)
        if (m.kind != RT_MAT_EMITTER && m.kind != RT_MAT_DEBUG_UV && m.kind != RT_MAT_DEBUG_NORMAL) sc->all_terminal = false;
    sc->mesh_only = d->n_nodes > 0;
    for (const DNode& n : nodes)
        if (n.kind != RT_SHAPE_MESH) sc->mesh_only = false;
    // nodes that hold a mesh, directly or under transforms (flagged in DNode::pad): the trees the scattered-ray kernel walks
    std::vector<int32_t> walk_nodes;
    for (uint64_t i = 0; i < d->n_nodes; i++) {
        int kind = nodes[i].kind, index = nodes[i].index;
        for (int depth = 0; kind == RT_SHAPE_TRANSFORMED && depth < 4; depth++) {
            const DTransformed& x = xforms[(size_t)index];
            kind = x.child_kind;
            index = x.child_index;
        }
        nodes[i].pad = kind == RT_SHAPE_MESH ? 1 : 0;
        if (nodes[i].pad) walk_nodes.push_back((int32_t)i);
    }
    sc->nodes = nodes;
    // RT_STACK_PAIRS: a test knob that gives the traversal stacks fewer entries than the deepest tree needs, to
    // exercise the overflow report (a render then fails with RT_ERR_DEVICE instead of returning a wrong image)
    if (const char* e = std::getenv("RT_STACK_PAIRS")) stack_pairs = walk_cap = std::max(1, std::atoi(e));
    DScene proto{};
    proto.n_walk_nodes = (int32_t)walk_nodes.size();
    proto.early_miss = std::getenv("RT_NO_EARLY_MISS") == nullptr ? 1 : 0;
    {   // kinds of Material.Resolve code among the nodes' materials (the three that scatter, and "only emits")
        unsigned kinds = 0;
        for (const DNode& n : nodes) {
            const int k = materials[(size_t)n.material].kind;
            kinds |= 1u << (k == RT_MAT_LAMBERTIAN ? 0 : (k == RT_MAT_METAL ? 1 : (k == RT_MAT_DIELECTRIC ? 2 : 3)));
        }
        proto.sort_kinds = ((kinds & (kinds - 1)) != 0 && std::getenv("RT_NO_KIND_SORT") == nullptr) ? 1 : 0;
    }
    if (d->n_nodes <= 64)
        for (int32_t i : walk_nodes) proto.walk_mask |= 1ull << i;
    proto.n_nodes = (int32_t)d->n_nodes;
    proto.n_lights = (int32_t)d->n_lights;
    proto.stack_pairs = stack_pairs;
    proto.walk_cap = walk_cap;
    // measured: +3 % (328 k faces, 20 levels) and +8 % (21 M faces) on the scattered rays of a scan, -2 % on a 6 k-face teapot (13 levels)
    proto.walk_break = deepest_fast >= 17 ? 15 : 0;
    if (const char* e = std::getenv("RT_WALK_BREAK")) proto.walk_break = std::max(0, std::min(100, std::atoi(e)));
    proto.n_meshes = (int32_t)d->n_meshes;
    {   // bounds of every node that holds a mesh (directly or under transforms): rays that miss them never walk a tree
        bool any = false;
        double lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
        for (uint64_t i = 0; i < d->n_nodes; i++) {
            if (nodes[i].kind != RT_SHAPE_MESH && nodes[i].kind != RT_SHAPE_TRANSFORMED) continue;
            for (int k = 0; k < 3; k++) {
                lo[k] = any ? std::min(lo[k], node_box[i * 6 + k]) : node_box[i * 6 + k];
                hi[k] = any ? std::max(hi[k], node_box[i * 6 + 3 + k]) : node_box[i * 6 + 3 + k];
            }
            any = true;
        }
        for (int k = 0; k < 3; k++) { proto.walk_box[k] = lo[k]; proto.walk_box[3 + k] = hi[k]; }
        proto.walk_classes = (any && d->n_meshes > 0 && !sc->mesh_only && std::getenv("RT_NO_RAY_CLASSES") == nullptr) ? 1 : 0;
    }
    // One host build, one upload per device: every device of rt_init gets the same arrays.
    for (size_t r = 0; r < g_devices.size(); r++) {
        std::unique_ptr<Replica> rep(new Replica());
        rep->device = g_devices[r];
        HIP_TRY(hipSetDevice(rep->device));
        std::vector<DImage> images(d->n_images);
        for (uint64_t i = 0; i < d->n_images; i++) {
            const rt_image& im = d->images[i];
            const size_t bytes = (size_t)im.width * im.height * (im.format == RT_IMG_RGBA16 ? 8 : 4);
            uint8_t* px = nullptr;
            HIP_TRY(rep->mem.alloc(bytes, &px));
            HIP_TRY(hipMemcpy(px, im.pixels, bytes, hipMemcpyHostToDevice));
            images[i] = {px, im.width, im.height, im.format, 0};
        }
        DScene& dev = rep->dev;
        dev = proto;
        HIP_TRY(rep->mem.upload(nodes, &dev.nodes));
        HIP_TRY(rep->mem.upload(spheres, &dev.spheres));
        HIP_TRY(rep->mem.upload(quads, &dev.quads));
        HIP_TRY(rep->mem.upload(cyls, &dev.cylinders));
        HIP_TRY(rep->mem.upload(tris, &dev.tris));
        HIP_TRY(rep->mem.upload(xforms, &dev.xforms));
        HIP_TRY(rep->mem.upload(meshes, &dev.meshes));
        HIP_TRY(rep->mem.upload(ref_nodes, &dev.ref_nodes));
        HIP_TRY(rep->mem.upload(fast_nodes, &dev.fast_nodes));
        HIP_TRY(rep->mem.upload(q_nodes, &dev.q_nodes));
        HIP_TRY(rep->mem.upload(q4_nodes, &dev.q4_nodes));
        HIP_TRY(rep->mem.upload(prim_pos, &dev.prim_pos));
        HIP_TRY(rep->mem.upload(prim_tri32, &dev.prim_tri32));
        HIP_TRY(rep->mem.upload(prim_face, &dev.prim_face));
        HIP_TRY(rep->mem.upload(face_uv, &dev.face_uv));
        HIP_TRY(rep->mem.upload(materials, &dev.materials));
        HIP_TRY(rep->mem.upload(textures, &dev.textures));
        HIP_TRY(rep->mem.upload(images, &dev.images));
        HIP_TRY(rep->mem.upload(lights, &dev.lights));
        HIP_TRY(rep->mem.upload(node_box, &dev.node_box));
        if (!node_pairs.empty() && std::getenv("RT_NO_NODE_PAIRS") == nullptr) {
            HIP_TRY(rep->mem.upload(node_pairs, &dev.node_pairs));
            std::memcpy(dev.pair_center, pair_center, sizeof(pair_center));
            dev.pair_radius = pair_radius;
        }
        HIP_TRY(rep->mem.upload(walk_nodes, &dev.walk_nodes));
        if (beam_scene) {  // the pixel-beam kernel's per-direction tables: filled when a frame's camera is known
            size_t off_f64 = 0, off_uv = 0;
            const int64_t n = (int64_t)d->meshes[0].n_faces;
            uint8_t* base = nullptr;
            HIP_TRY(rep->mem.alloc(beam_table_bytes(n, &off_f64, &off_uv), &base));
            rep->beam.f32 = reinterpret_cast<const BeamFace32*>(base);
            rep->beam.f64 = reinterpret_cast<const BeamFace64*>(base + off_f64);
            rep->beam.uv = reinterpret_cast<const double*>(base + off_uv);
            rep->beam.image = d->materials[nodes[0].material].kind == RT_MAT_EMITTER &&
                              d->textures[d->materials[nodes[0].material].texture].kind == RT_TEX_IMAGE;
            rep->beam_faces = n;
            // a stretch per persistent wave for the walks of 32 pixels: their states and lists of up to kBeamCandsMax leaves (199 MB on 256 CUs)
            rep->beam.walk_stride = beam_walk_stride();
            rep->beam.walk_waves = beam_walk_waves();
            HIP_TRY(rep->mem.alloc((size_t)rep->beam.walk_waves * (size_t)rep->beam.walk_stride, &rep->beam.walk_scratch));
            HIP_TRY(rep->mem.alloc((size_t)kBeamLongCap, &rep->beam.long_px));
            HIP_TRY(hipMemset(rep->beam.long_px, 0, (size_t)kBeamLongCap * sizeof(unsigned long long)));  // (no launch has the number 0)
        }
        HIP_TRY(hipEventCreateWithFlags(&rep->done, hipEventDisableTiming));
        if (r > 0) HIP_TRY(hipStreamCreateWithFlags(&rep->stream, hipStreamNonBlocking));
        sc->replicas.push_back(std::move(rep));
    }
    HIP_TRY(hipSetDevice(g_devices[0]));
    *out = sc.release();
    return RT_OK;
}

void rt_scene_destroy(rt_scene* scene) {
    delete scene;
}

int rt_scene_bvh_info(const rt_scene* scene, uint64_t mesh_index, rt_bvh_info* out) {
    if (!scene || !out || mesh_index >= scene->bvh_info.size()) return fail(RT_ERR_INVALID, "bad mesh index");
    *out = scene->bvh_info[mesh_index];
    return RT_OK;
}

int rt_bvh_build(const rt_face* faces, uint64_t n_faces, rt_bvh_node* nodes, uint64_t node_cap, int32_t* leaf_faces, uint64_t* n_nodes) {
    static_assert(sizeof(rt_bvh_node) == sizeof(RefNode), "rt_bvh_node mirrors RefNode");
    if (!faces || n_faces == 0 || n_faces > (uint64_t)INT32_MAX || !n_nodes) return fail(RT_ERR_INVALID, "bad argument");
    MeshBvh b;
    build_mesh_bvh(faces, n_faces, b);
    *n_nodes = b.ref_nodes.size();
    if (!nodes || !leaf_faces || node_cap < b.ref_nodes.size()) return fail(RT_ERR_INVALID, "node array too small");
    for (size_t i = 0; i < b.ref_nodes.size(); i++) {
        const RefNode& r = b.ref_nodes[i];
        rt_bvh_node& o = nodes[i];
        for (int k = 0; k < 3; k++) { o.bounds_min[k] = r.box[k]; o.bounds_max[k] = r.box[3 + k]; }
        const bool leaf = r.right < 0;
        o.right = r.right;
        o.first = leaf ? r.first : 0;
        o.count = leaf ? r.count : 0;
        o.pad = 0;
    }
    std::memcpy(leaf_faces, b.prim_face.data(), n_faces * sizeof(int32_t));
    return RT_OK;
}

void rt_cancel(rt_scene* scene) {
    if (scene) scene->cancel.store(true);
}

int rt_render_film(rt_scene* scene, const rt_camera_desc* camera, const rt_render_opts* opts, double* film_dev, rt_stats* stats) {
    if (!g_initialized) return fail(RT_ERR_NOT_INITIALIZED, "rt_init has not been called");
    if (scene) scene->cancel.store(false);
    return render_all_devices(scene, camera, opts, film_dev, stats);
}

int rt_render_film_noalloc(rt_scene* scene, const rt_camera_desc* camera, const rt_render_opts* opts, double* film_dev, rt_stats* stats) {
    if (!g_initialized) return fail(RT_ERR_NOT_INITIALIZED, "rt_init has not been called");
    if (!scene || !camera || !opts || !film_dev) return fail(RT_ERR_INVALID, "null argument");
    if (scene->replicas.size() != 1) return fail(RT_ERR_UNSUPPORTED, "rt_render_film_noalloc renders scenes that live on one device");
    if (opts->dx <= 0 || opts->dy <= 0) return fail(RT_ERR_INVALID, "bad Dx / Dy must be positive");
    // this call's rows (its bands of 16) in whole groups of 64 x 64 pixels, as the render loop counts its tiles
    const int world = opts->band_world > 0 ? opts->band_world : 1;
    int64_t local_rows = 0;
    for (int r = 0; r < opts->dy; r++)
        if ((r / 16) % world == opts->band_rank) local_rows++;
    const int64_t tiles = (int64_t)((opts->dx + 63) / 64) * ((local_rows + 63) / 64) * 64;
    Replica& rep = *scene->replicas[0];
    WorkBuffers* w = static_work_for(rep.device, tiles);
    if (!w) return RT_ERR_UNSUPPORTED;  // message set by static_work_for
    scene->cancel.store(false);
    return render_into_film(scene, rep, camera, opts, film_dev, stats, w);
}

int rt_film_resolve(const double* film_dev, int32_t dx, int32_t dy, int32_t rays_per_pixel, uint8_t* rgba_out, size_t stride, void* stream_) {
    if (!g_initialized) return fail(RT_ERR_NOT_INITIALIZED, "rt_init has not been called");
    if (!film_dev || !rgba_out || dx <= 0 || dy <= 0 || rays_per_pixel <= 0 || stride < (size_t)dx * 4) return fail(RT_ERR_INVALID, "bad argument");
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    HIP_TRY(hipSetDevice(g_devices[0]));
    uint8_t* rgba = nullptr;
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&rgba), (size_t)dx * dy * 4));
    launch_resolve(film_dev, dx, dy, rays_per_pixel, rgba, stream);
    hipError_t e = stride == (size_t)dx * 4
                       ? hipMemcpyAsync(rgba_out, rgba, (size_t)dx * 4 * (size_t)dy, hipMemcpyDeviceToHost, stream)
                       : hipMemcpy2DAsync(rgba_out, stride, rgba, (size_t)dx * 4, (size_t)dx * 4, (size_t)dy, hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    (void)hipFree(rgba);
    if (e != hipSuccess) return fail(RT_ERR_DEVICE, hipGetErrorString(e));
    return RT_OK;
}

int rt_apply_filter(const uint8_t* rgba_in, size_t stride_in, int32_t dx, int32_t dy, int filter, uint8_t* rgba_out, size_t stride_out) {
    if (!g_initialized) return fail(RT_ERR_NOT_INITIALIZED, "rt_init has not been called");
    if (!rgba_in || !rgba_out || dx <= 0 || dy <= 0 || stride_in < (size_t)dx * 4 || stride_out < (size_t)dx * 4)
        return fail(RT_ERR_INVALID, "bad argument");
    if (filter < RT_RECON_BOX || filter > RT_RECON_MITCHELL) return fail(RT_ERR_INVALID, "unknown reconstruction filter");
    HIP_TRY(hipSetDevice(g_devices[0]));
    const size_t row = (size_t)dx * 4, px = (size_t)dx * dy;
    DeviceBuffers tmp;
    uint8_t *src = nullptr, *dst = nullptr;
    double* mid = nullptr;
    HIP_TRY(tmp.alloc(px * 4, &src));
    HIP_TRY(tmp.alloc(px * 4, &dst));
    HIP_TRY(tmp.alloc(px * 3, &mid));
    HIP_TRY(hipMemcpy2D(src, row, rgba_in, stride_in, row, (size_t)dy, hipMemcpyHostToDevice));
    launch_filter(src, dx, dy, filter, mid, dst, nullptr);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy2D(rgba_out, stride_out, dst, row, row, (size_t)dy, hipMemcpyDeviceToHost));
    return RT_OK;
}

int rt_render(rt_scene* scene, const rt_camera_desc* camera, const rt_render_opts* opts, uint8_t* rgba_out, size_t stride, rt_stats* stats) {
    if (!scene || !opts || !rgba_out) return fail(RT_ERR_INVALID, "null argument");
    if (opts->dx <= 0 || opts->dy <= 0) return fail(RT_ERR_INVALID, "bad Dx / Dy must be positive");
    if (stride < (size_t)opts->dx * 4) return fail(RT_ERR_INVALID, "stride smaller than a row");
    if (!g_initialized) return fail(RT_ERR_NOT_INITIALIZED, "rt_init has not been called");
    Replica& first = *scene->replicas[0];
    HIP_TRY(hipSetDevice(first.device));
    scene->cancel.store(false);  // a cancel that came after an earlier call had ended is not meant for this one
    double* film = nullptr;
    uint8_t *rgba = nullptr, *rgba_host = nullptr;
    if (int rc = scene_film(scene, first, opts->dx, opts->dy, &film, &rgba, &rgba_host)) return rc;
    const size_t n = (size_t)opts->dx * opts->dy * 3;
    hipStream_t stream = reinterpret_cast<hipStream_t>(opts->stream);
    const auto t0 = std::chrono::steady_clock::now();
    HIP_TRY(hipMemsetAsync(film, 0, n * sizeof(double), stream));
    if (int rc = render_all_devices(scene, camera, opts, film, stats)) return rc;
    const auto t1 = std::chrono::steady_clock::now();
    launch_resolve(film, opts->dx, opts->dy, opts->rays_per_pixel, rgba, stream);
    // device -> pinned staging buffer at full PCIe rate, then rows into the caller's (pageable, possibly strided) image
    const size_t row = (size_t)opts->dx * 4;
    HIP_TRY(hipMemcpyAsync(rgba_host, rgba, row * (size_t)opts->dy, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    if (stride == row) std::memcpy(rgba_out, rgba_host, row * (size_t)opts->dy);
    else
        for (int y = 0; y < opts->dy; y++) std::memcpy(rgba_out + (size_t)y * stride, rgba_host + (size_t)y * row, row);
    if (std::getenv("RT_DEBUG"))
        fprintf(stderr, "[rt] rt_render: film %.3f ms, resolve + copy out %.3f ms\n", std::chrono::duration<double>(t1 - t0).count() * 1e3,
                std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count() * 1e3);
    if (stats) stats->render_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return RT_OK;
}

int rt_render_frames(rt_scene* scene, const rt_camera_desc* cameras, int32_t n_frames, const rt_render_opts* opts, uint8_t* rgba_out,
                     size_t stride, size_t frame_stride, rt_stats* stats, int32_t overlap) {
    if (!scene || !opts || !rgba_out || (n_frames > 0 && !cameras)) return fail(RT_ERR_INVALID, "null argument");
    if (n_frames < 0) return fail(RT_ERR_INVALID, "n_frames must not be negative");
    if (opts->dx <= 0 || opts->dy <= 0) return fail(RT_ERR_INVALID, "bad Dx / Dy must be positive");
    const size_t row = (size_t)opts->dx * 4;
    if (stride < row) return fail(RT_ERR_INVALID, "stride smaller than a row");
    if (n_frames > 1 && frame_stride < stride * (size_t)(opts->dy - 1) + row) return fail(RT_ERR_INVALID, "frame_stride smaller than a frame");
    if (!g_initialized) return fail(RT_ERR_NOT_INITIALIZED, "rt_init has not been called");
    if (n_frames == 0) return RT_OK;
    Replica& first = *scene->replicas[0];
    HIP_TRY(hipSetDevice(first.device));
    const int depth = overlap && n_frames > 1 ? 2 : 1;
    double* film[2] = {};
    uint8_t *rgba[2] = {}, *host[2] = {};
    for (int s = 0; s < depth; s++)
        if (int rc = scene_film(scene, first, opts->dx, opts->dy, &film[s], &rgba[s], &host[s], s)) return rc;
    hipStream_t stream = reinterpret_cast<hipStream_t>(opts->stream);

    // Copies go out on a stream of their own (not synchronised with the null stream) and are moved from the pinned
    // staging buffer into the caller's frames by a second host thread, so that neither holds up the next frame's trace.
    struct Pipe {
        hipStream_t copy = nullptr;
        hipEvent_t resolved[2] = {}, copied[2] = {};
        ~Pipe() {
            for (hipEvent_t e : resolved) if (e) (void)hipEventDestroy(e);
            for (hipEvent_t e : copied) if (e) (void)hipEventDestroy(e);
            if (copy) (void)hipStreamDestroy(copy);
        }
    } pipe;
    HIP_TRY(hipStreamCreateWithFlags(&pipe.copy, hipStreamNonBlocking));
    for (int s = 0; s < depth; s++) {
        HIP_TRY(hipEventCreateWithFlags(&pipe.resolved[s], hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&pipe.copied[s], hipEventDisableTiming));
    }
    std::mutex mu;
    std::condition_variable cv;
    int submitted = 0, drained = 0;
    bool closed = false;
    hipError_t drain_error = hipSuccess;
    std::thread drain([&]() {
        (void)hipSetDevice(first.device);
        for (int i = 0;; i++) {
            {
                std::unique_lock<std::mutex> lock(mu);
                cv.wait(lock, [&] { return submitted > i || closed; });
                if (submitted <= i) return;
            }
            const int s = i % depth;
            hipError_t e = hipEventSynchronize(pipe.copied[s]);
            if (e == hipSuccess) {
                uint8_t* dst = rgba_out + (size_t)i * frame_stride;
                if (stride == row) std::memcpy(dst, host[s], row * (size_t)opts->dy);
                else
                    for (int y = 0; y < opts->dy; y++) std::memcpy(dst + (size_t)y * stride, host[s] + (size_t)y * row, row);
            }
            std::lock_guard<std::mutex> lock(mu);
            if (e != hipSuccess && drain_error == hipSuccess) drain_error = e;
            drained = i + 1;
            cv.notify_all();
        }
    });
    auto finish = [&]() {
        {
            std::lock_guard<std::mutex> lock(mu);
            closed = true;
        }
        cv.notify_all();
        drain.join();
    };

    int rc = RT_OK;
    std::string message;
    hipError_t e = hipSuccess;
    const size_t n = (size_t)opts->dx * opts->dy * 3;
    scene->cancel.store(false);  // the flag belongs to this call: render_into_film reads it, nobody resets it until the call ends
    for (int i = 0; i < n_frames && rc == RT_OK && e == hipSuccess; i++) {
        if (scene->cancel.load()) {  // between frames: a loop of one-launch frames has no other place to stop
            rc = RT_ERR_CANCELLED;
            message = "render cancelled";
            break;
        }
        const int s = i % depth;
        {  // the set's staging buffer is free once the frame that used it last has been handed over
            std::unique_lock<std::mutex> lock(mu);
            cv.wait(lock, [&] { return drained > i - depth; });
        }
        const auto t0 = std::chrono::steady_clock::now();
        rt_stats st{};
        e = hipMemsetAsync(film[s], 0, n * sizeof(double), stream);
        if (e != hipSuccess) break;
        rc = render_all_devices(scene, &cameras[i], opts, film[s], &st);
        if (rc != RT_OK) {
            message = last_error_text();
            break;
        }
        launch_resolve(film[s], opts->dx, opts->dy, opts->rays_per_pixel, rgba[s], stream);
        e = hipEventRecord(pipe.resolved[s], stream);
        if (e == hipSuccess) e = hipStreamWaitEvent(pipe.copy, pipe.resolved[s], 0);
        if (e == hipSuccess) e = hipMemcpyAsync(host[s], rgba[s], row * (size_t)opts->dy, hipMemcpyDeviceToHost, pipe.copy);
        if (e == hipSuccess) e = hipEventRecord(pipe.copied[s], pipe.copy);
        if (e != hipSuccess) break;
        st.render_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (stats) stats[i] = st;
        {
            std::lock_guard<std::mutex> lock(mu);
            submitted = i + 1;
        }
        cv.notify_all();
    }
    finish();
    if (rc != RT_OK) return fail(rc, message);
    if (e == hipSuccess) e = drain_error;
    if (e != hipSuccess) return fail(RT_ERR_DEVICE, std::string("rt_render_frames: ") + hipGetErrorString(e));
    return RT_OK;
}

int rt_intersect_batch(rt_scene* scene, const double* o, const double* d, uint64_t n, double tmin, double tmax, int mode, rt_hit* hits) {
    if (!g_initialized) return fail(RT_ERR_NOT_INITIALIZED, "rt_init has not been called");
    if (!scene || (n && (!o || !d || !hits))) return fail(RT_ERR_INVALID, "null argument");
    if (mode != RT_MODE_REFERENCE && mode != RT_MODE_FAST) return fail(RT_ERR_INVALID, "unknown mode");
    if (n == 0) return RT_OK;
    HIP_TRY(hipSetDevice(scene->replicas[0]->device));
    DeviceBuffers tmp;
    double *od = nullptr, *dd = nullptr;
    rt_hit* hd = nullptr;
    HIP_TRY(tmp.alloc(n * 3, &od));
    HIP_TRY(tmp.alloc(n * 3, &dd));
    HIP_TRY(tmp.alloc(n, &hd));
    HIP_TRY(hipMemcpy(od, o, n * 24, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dd, d, n * 24, hipMemcpyHostToDevice));
    launch_intersect_batch(scene->replicas[0]->dev, od, dd, n, tmin, tmax, mode, hd, nullptr);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(hits, hd, n * sizeof(rt_hit), hipMemcpyDeviceToHost));
    return RT_OK;
}

int rt_microfacet_batch(int which, uint64_t n, const double* roughness, const double* f0, const double* args, double* out) {
    if (!g_initialized) return fail(RT_ERR_NOT_INITIALIZED, "rt_init has not been called");
    if (which < RT_MF_D || which > RT_MF_EVALUATE) return fail(RT_ERR_INVALID, "unknown microfacet function");
    if (n && (!roughness || !f0 || !args || !out)) return fail(RT_ERR_INVALID, "null argument");
    if (n == 0) return RT_OK;
    HIP_TRY(hipSetDevice(g_devices[0]));
    DeviceBuffers tmp;
    double *rd = nullptr, *fd = nullptr, *ad = nullptr, *od = nullptr;
    HIP_TRY(tmp.alloc(n, &rd));
    HIP_TRY(tmp.alloc(n * 3, &fd));
    HIP_TRY(tmp.alloc(n * 12, &ad));
    HIP_TRY(tmp.alloc(n * 3, &od));
    HIP_TRY(hipMemcpy(rd, roughness, n * 8, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(fd, f0, n * 24, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(ad, args, n * 96, hipMemcpyHostToDevice));
    launch_microfacet_batch(which, n, rd, fd, ad, od, nullptr);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(out, od, n * 24, hipMemcpyDeviceToHost));
    return RT_OK;
}

int rt_shade_batch(rt_scene* scene, const rt_shade_in* hits, uint64_t n, int64_t seed, int mode, rt_shade_out* out) {
    if (!g_initialized) return fail(RT_ERR_NOT_INITIALIZED, "rt_init has not been called");
    if (!scene || (n && (!hits || !out))) return fail(RT_ERR_INVALID, "null argument");
    if (mode != RT_MODE_REFERENCE && mode != RT_MODE_FAST) return fail(RT_ERR_INVALID, "unknown mode");
    if (seed < 0) return fail(RT_ERR_INVALID, "bad Seed must be non-negative");
    const Replica& rep = *scene->replicas[0];
    for (uint64_t i = 0; i < n; i++) {
        if (hits[i].material < 0 || (size_t)hits[i].material >= scene->n_materials) return fail(RT_ERR_INVALID, "material index out of range");
        if (hits[i].node < 0 || (size_t)hits[i].node >= scene->nodes.size()) return fail(RT_ERR_INVALID, "node index out of range");
    }
    if (n == 0) return RT_OK;
    HIP_TRY(hipSetDevice(rep.device));
    DeviceBuffers tmp;
    rt_shade_in* hd = nullptr;
    rt_shade_out* od = nullptr;
    HIP_TRY(tmp.alloc(n, &hd));
    HIP_TRY(tmp.alloc(n, &od));
    HIP_TRY(hipMemcpy(hd, hits, n * sizeof(rt_shade_in), hipMemcpyHostToDevice));
    launch_shade_batch(rep.dev, hd, n, (uint32_t)((uint64_t)seed & 0xffffffffu), (uint32_t)((uint64_t)seed >> 32), mode, od, nullptr);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(out, od, n * sizeof(rt_shade_out), hipMemcpyDeviceToHost));
    return RT_OK;
}

int rt_cast_batch(const rt_camera_desc* camera, const double* st, const uint32_t* pixel_sample, uint64_t n, int64_t seed, double* o_out,
                  double* d_out) {
    if (!g_initialized) return fail(RT_ERR_NOT_INITIALIZED, "rt_init has not been called");
    if (!camera || (n && (!st || !pixel_sample || !o_out || !d_out))) return fail(RT_ERR_INVALID, "null argument");
    if (n == 0) return RT_OK;
    HIP_TRY(hipSetDevice(g_devices[0]));
    DCamera cam;
    if (int rc = make_camera(*camera, cam)) return rc;
    DeviceBuffers tmp;
    double *std_ = nullptr, *od = nullptr, *dd = nullptr;
    uint32_t* psd = nullptr;
    HIP_TRY(tmp.alloc(n * 2, &std_));
    HIP_TRY(tmp.alloc(n * 2, &psd));
    HIP_TRY(tmp.alloc(n * 3, &od));
    HIP_TRY(tmp.alloc(n * 3, &dd));
    HIP_TRY(hipMemcpy(std_, st, n * 16, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(psd, pixel_sample, n * 8, hipMemcpyHostToDevice));
    launch_cast_batch(cam, std_, psd, n, (uint32_t)((uint64_t)seed & 0xffffffffu), (uint32_t)((uint64_t)seed >> 32), od, dd, nullptr);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(o_out, od, n * 24, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(d_out, dd, n * 24, hipMemcpyDeviceToHost));
    return RT_OK;
}

}  // extern "C"
