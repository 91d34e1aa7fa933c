/*
 * rt_hip.h -- C ABI of librt_hip.so, the MI355X (gfx950) implementation of
 * raytrace's per-pixel path-tracing loop.
 *
 * The reference has no FFI of its own; the seam this library plugs into is the
 * Go function renderScene(ctx, *Scene, Camera) (phys/render.go:285-349), called
 * only from phys.Render (phys/render.go:351-364).  Every entry point below cites
 * the reference code whose behaviour it replaces.
 *
 * Conventions
 *   - plain C types only; all geometry is float64 in the caller's world units
 *     (the reference uses nanometres, phys/distance.go:7-19);
 *   - every function returns RT_OK (0) or a negative RT_ERR_* code and leaves a
 *     human readable message retrievable with rt_last_error() (thread local);
 *   - the caller owns every input array; the library deep-copies what it needs
 *     during the call, so no pointer has to outlive the call that received it
 *     (this is what cgo's pointer-passing rules require).
 */
#ifndef RT_HIP_H
#define RT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RT_OK 0
#define RT_ERR_INVALID (-1)         /* bad argument / inconsistent scene description */
#define RT_ERR_UNSUPPORTED (-2)     /* valid in the reference, not handled by this library */
#define RT_ERR_DEVICE (-3)          /* a HIP call failed (message carries hipGetErrorString) */
#define RT_ERR_CANCELLED (-4)       /* rt_cancel() was called; mirrors ctx.Err() (render.go:340-343) */
#define RT_ERR_NOMEM (-5)
#define RT_ERR_NOT_INITIALIZED (-6) /* rt_init() has not been called */

/* ---- shapes (phys/shape_*.go) ------------------------------------------------ */

enum rt_shape_kind {
    RT_SHAPE_SPHERE = 0,      /* phys.Sphere            shape_sphere.go:245-256  */
    RT_SHAPE_QUAD = 1,        /* phys.Quad              shape_quad.go:15-20      */
    RT_SHAPE_CYLINDER = 2,    /* phys.Cylinder          shape_cylinder.go:13-18  */
    RT_SHAPE_TRIANGLE = 3,    /* phys.Triangle / TriangleUV / a lone Face         */
    RT_SHAPE_MESH = 4,        /* phys.Mesh (BVH over Faces) shape_mesh.go:164-167 */
    RT_SHAPE_TRANSFORMED = 5  /* phys.TransformedShape  shape_transformed_shape.go:12-15 */
};

enum rt_uvmap_kind { /* phys.UVMapKind, shape_sphere.go:212-226 */
    RT_UVMAP_EQUIRECT = 0,
    RT_UVMAP_EQUIRECT_SOUTH_UP = 1,
    RT_UVMAP_BOX = 2
};

enum rt_triangle_kind {
    RT_TRI_TRIANGLE = 0,    /* phys.Triangle:   eps 1e-6, uv = (u,v), normal = unit(e1 x e2)  shape_triangle.go:105-139 */
    RT_TRI_TRIANGLE_UV = 1, /* phys.TriangleUV: eps 1e-6, uv = UV0.Lerp(UV1,u)+UV2*v, stored normal  shape_triangle_uv.go:117-150 */
    RT_TRI_FACE = 2         /* phys.Face:       eps 1e-8, barycentric uv, normal = unit(e1 x e2)  shape_mesh.go:95-142 */
};

typedef struct rt_sphere {
    double center[3];
    double radius;
    int32_t uvmap; /* rt_uvmap_kind */
    int32_t reserved;
} rt_sphere;

typedef struct rt_quad {
    double center[3];
    double normal[3];
    double width;
    double height;
} rt_quad;

typedef struct rt_cylinder {
    double origin[3];
    double direction[3];
    double radius;
    double height;
} rt_cylinder;

typedef struct rt_triangle {
    double p[3][3];   /* P0, P1, P2 */
    double uv[3][2];  /* UV0..UV2 (TriangleUV, Face); ignored for Triangle */
    double normal[3]; /* stored Normal (TriangleUV only) */
    int32_t kind;     /* rt_triangle_kind */
    int32_t reserved;
} rt_triangle;

/* One mesh face.  Identical to the in-memory layout of the Go value
 * phys.Face{Vertex [3]Vertex{Position r3.Point; UV r2.Point}} (shape_mesh.go:12-21):
 * 3 x (x, y, z, u, v) float64 = 120 bytes, so a Go []Face can be passed as is. */
typedef struct rt_face {
    double v[3][5];
} rt_face;

/* phys.Mesh: a run of faces in rt_scene_desc.faces.  The library builds the BVH
 * itself with the reference's binned-SAH rules (NewBVH, shape_bvh.go:81-272), so
 * a host may skip NewMesh's own build. */
typedef struct rt_mesh {
    uint64_t first_face;
    uint64_t n_faces;
} rt_mesh;

/* phys.TransformedShape{Shape, Transform{Translation, Rotation, Scale}}
 * (shape_transformed_shape.go:12-15, transform.go:53-57).  The child may be any
 * shape kind, another RT_SHAPE_TRANSFORMED included. */
typedef struct rt_transformed {
    int32_t child_kind;  /* rt_shape_kind */
    int32_t child_index; /* index into the array of that kind */
    double translation[3];
    double rotation[9]; /* row major r3.Mat3x3.M */
    double scale[3];
} rt_transformed;

/* ---- textures, materials, lights ---------------------------------------------- */

enum rt_texture_kind {
    RT_TEX_UNIFORM = 0,      /* phys.TextureUniform       texture_uniform.go:13-20 */
    RT_TEX_CHECKERBOARD = 1, /* phys.TextureCheckerboard  texture_checkerboard.go:21-69 */
    RT_TEX_IMAGE = 2         /* phys.TextureImage         texture_image.go:18-136 */
};
enum rt_interp_kind { RT_INTERP_NEAREST = 0, RT_INTERP_BILINEAR = 1 }; /* TextureImage.Interp: "bilinear" or anything else */
enum rt_wrap_kind { RT_WRAP_REPEAT = 0, RT_WRAP_CLAMP = 1 };           /* TextureImage.WrapMode: "clamp" or anything else */

enum rt_image_format {
    RT_IMG_NRGBA8 = 0, /* 4 x uint8 non-premultiplied (Go image.NRGBA, what png.Decode yields for RGBA PNGs) */
    RT_IMG_RGBA16 = 1  /* 4 x uint16 alpha-premultiplied, i.e. the values color.Color.RGBA() returns */
};

typedef struct rt_image {
    int32_t width;
    int32_t height;
    int32_t format;     /* rt_image_format */
    int32_t reserved;
    const void* pixels; /* row major, row 0 = top, tightly packed */
} rt_image;

typedef struct rt_texture {
    int32_t kind;     /* rt_texture_kind */
    int32_t odd;      /* checkerboard: texture index of Odd  */
    int32_t even;     /* checkerboard: texture index of Even */
    int32_t image;    /* image: index into images */
    int32_t interp;   /* rt_interp_kind */
    int32_t wrap;     /* rt_wrap_kind */
    double color[3];  /* uniform */
    double frequency; /* checkerboard */
} rt_texture;

enum rt_material_kind {
    RT_MAT_LAMBERTIAN = 0,  /* material_lambertian.go  */
    RT_MAT_METAL = 1,       /* material_metal.go       */
    RT_MAT_DIELECTRIC = 2,  /* material_dielectric.go  */
    RT_MAT_EMITTER = 3,     /* material_emitter.go     */
    RT_MAT_DEBUG_UV = 4,    /* material_debug_uv.go    */
    RT_MAT_DEBUG_NORMAL = 5 /* material_debug_normal.go */
};

typedef struct rt_material {
    int32_t kind;    /* rt_material_kind */
    int32_t texture; /* Lambertian / Emitter: texture index */
    double albedo[3];     /* Metal.Albedo */
    double fuzz;          /* Metal.Fuzz */
    double ior_interior;  /* Dielectric.RefractiveIndexInterior */
    double ior_exterior;  /* Dielectric.RefractiveIndexExterior */
    double roughness;     /* Dielectric.Roughness */
} rt_material;

typedef struct rt_light { /* phys.PointLight, light.go:16-19 */
    double position[3];
    double intensity[3];
} rt_light;

typedef struct rt_node { /* phys.Node{Shape, Material}, node.go:21-26 */
    int32_t shape_kind;  /* rt_shape_kind */
    int32_t shape_index; /* index into the array of that kind */
    int32_t material;    /* index into materials */
    int32_t reserved;
} rt_node;

/* A whole phys.Scene minus cameras and RenderOptions (scene.go:9-14), flattened. */
typedef struct rt_scene_desc {
    const rt_node* nodes;             uint64_t n_nodes;
    const rt_sphere* spheres;         uint64_t n_spheres;
    const rt_quad* quads;             uint64_t n_quads;
    const rt_cylinder* cylinders;     uint64_t n_cylinders;
    const rt_triangle* triangles;     uint64_t n_triangles;
    const rt_mesh* meshes;            uint64_t n_meshes;
    const rt_face* faces;             uint64_t n_faces;
    const rt_transformed* transformed; uint64_t n_transformed;
    const rt_material* materials;     uint64_t n_materials;
    const rt_texture* textures;       uint64_t n_textures;
    const rt_image* images;           uint64_t n_images;
    const rt_light* lights;           uint64_t n_lights;
} rt_scene_desc;

/* ---- cameras (phys/camera_*.go) ------------------------------------------------ */

enum rt_camera_kind {
    RT_CAM_PINHOLE = 0,      /* camera_pinhole.go:12-34      */
    RT_CAM_FOCUSABLE = 1,    /* camera_focusable.go:14-63    */
    RT_CAM_ORTHOGRAPHIC = 2, /* camera_orthographic.go:14-46 */
    RT_CAM_CALIBRATED = 3    /* camera_calibrated.go:32-79, 110-136, 172-199 */
};

/* phys.AnimatedCamera (camera_animated.go:70-74) is a Go closure; the host
 * resolves it to one of the four concrete kinds before calling. */
typedef struct rt_camera_desc {
    int32_t kind; /* rt_camera_kind */
    int32_t reserved;
    /* Focusable / Orthographic / Calibrated extrinsics */
    double look_from[3];
    double look_at[3];
    double vup[3];
    double fov_height;
    double fov_width;
    double aperture;         /* Focusable */
    double working_distance; /* Focusable */
    /* Pinhole */
    double lower_left_corner[3];
    double origin[3];
    double horizontal[3];
    double vertical[3];
    /* Calibrated intrinsics (CameraIntrinsics) */
    int32_t width;
    int32_t height;
    double fx, fy, cx, cy;
    double k1, k2, p1, p2, k3, k4, k5, k6;
} rt_camera_desc;

/* ---- render ------------------------------------------------------------------------ */

enum rt_mode {
    /* float64 everywhere, BVH visited in the reference's order (left child first,
     * shrinking tmax, shape_bvh.go:44-78) with Go's NaN-propagating Max/Min:
     * hits are bit-identical to the reference algorithm's. */
    RT_MODE_REFERENCE = 0,
    /* float32 conservative box culling, nearest-child-first order, any-hit shadow
     * rays; primitives are still intersected in float64 so t/uv/normal of the hit
     * primitive are the reference's values. */
    RT_MODE_FAST = 1
};

/* How a scene that lives on several devices (rt_init with n > 1) divides a frame among them. */
enum rt_device_split {
    RT_SPLIT_AUTO = 0,    /* samples when every device gets >= 256 of them and the frame has <= 2048 x 2048 pixels, else rows */
    RT_SPLIT_ROWS = 1,    /* band b of 16 rows -> device b % n; each device copies its rows into the first device's film */
    RT_SPLIT_SAMPLES = 2  /* device i renders the i-th run of sample indices of every pixel; the films are added on the first device */
};

typedef struct rt_render_opts { /* phys.RenderOptions, render.go:73-79 */
    int64_t seed;
    int32_t rays_per_pixel;
    int32_t max_ray_depth;
    int32_t dx;
    int32_t dy;
    int32_t mode; /* rt_mode */
    /* Row-band sharding: the image is cut into bands of 16 rows (the reference's
     * tile height, render.go:298); this call renders band b iff
     * b % band_world == band_rank.  1 process: band_rank 0, band_world 1. */
    int32_t band_rank;
    int32_t band_world;
    /* Sample sub-range: render only samples [sample_begin, sample_begin + sample_count) of the
     * rays_per_pixel samples of each pixel (progressive accumulation into one film);
     * sample_count == 0 means all of them. */
    int32_t sample_begin;
    int32_t sample_count;
    int32_t device_split; /* rt_device_split; ignored by a scene on one device */
    void* stream; /* hipStream_t (of the first device) to launch on; NULL = the null stream */
} rt_render_opts;

typedef struct rt_stats { /* phys.RenderStats, render.go:38-45 */
    uint64_t total_rays;          /* every tracePath call (render.go:143) */
    uint64_t rays_exceeded_depth; /* render.go:151-153 */
    uint64_t rays_left_scene;     /* render.go:170-172 */
    uint64_t shadow_rays;         /* occlusion rays; the reference does not count these */
    double render_seconds;        /* wall time of the call, scene already resident */
    double traverse_seconds;      /* sum of HIP-event time of the closest-hit traversal launches */
    uint64_t traverse_launches;
    uint64_t traverse_rays;       /* rays those launches generated and traced (total_rays less culled_rays) */
    uint64_t culled_rays;         /* camera rays (counted in total_rays and rays_left_scene) that were not generated because their
                                     pixel lies outside the screen bounds of every node: they cannot hit anything */
} rt_stats;

typedef struct rt_hit {
    int32_t node;    /* index of the scene node hit, -1 = miss */
    int32_t prim;    /* face index inside the mesh (order of rt_scene_desc.faces), else 0 */
    double t;
    double at[3];
    double normal[3];
    double uv[2];
    uint32_t boxes_tested; /* AABB.hit calls made for this ray (reference order only) */
    uint32_t prims_tested; /* Face.Collide calls made for this ray (reference order only) */
} rt_hit;

typedef struct rt_scene rt_scene;

/* Select the GPUs this process uses (device ordinals) and create per-device
 * state.  gpu_ids == NULL or n == 0 selects device 0.  With n > 1 every scene
 * created afterwards is replicated on all of them and rt_render / rt_render_film
 * divide each frame among them (rt_device_split); the first device is where
 * film_dev lives and where the frame is assembled.  An ordinal may be listed
 * more than once: every entry gets a replica and a stream of its own, which lets
 * a one-GPU machine run the multi-device path.  Replaces the reference's worker
 * pool of runtime.NumCPU() goroutines (render.go:297-331); called once from the
 * Go package's init. */
int rt_init(const int* gpu_ids, int n);
void rt_shutdown(void);

/* Thread-local message of the last failing call on this thread.  Go wraps it as
 * fmt.Errorf("failed to render scene: %v", ...) like render.go:361. */
const char* rt_last_error(void);

/* Deep-copies the scene to the device: flattens shapes, builds every mesh BVH
 * (NewBVH, shape_bvh.go:81-272), uploads textures.  Performs the checks of
 * Scene.Validate that concern the arrays (scene.go:29-79). */
int rt_scene_create(const rt_scene_desc* desc, rt_scene** out);
void rt_scene_destroy(rt_scene* scene);

/* rt_scene_create from the text Scene.MarshalJSON writes (scene.go:82-114; the
 * {"Type","Data"} wrapping of every interface value is json.go:52-100), for a
 * host that holds a scene as JSON -- an exported file, the body of the
 * playground's POST /raytrace/render (playground/main.go:66-110) -- and would
 * otherwise decode it only to flatten it again.  Reads everything the
 * reference writes: the registered types, PinholeCamera / FocusableCamera /
 * Triangle (written but never registered there), untagged interface fields
 * (Lambertian.Texture, TransformedShape.Shape) and image pixels carried
 * inline (image.NRGBA / image.RGBA64 as encoding/json writes them).  A mesh's
 * BVH member is skipped: the library builds that tree from the faces.  A
 * TextureImage without pixels is RT_ERR_UNSUPPORTED (the library decodes no
 * image files).  info (may be NULL) receives the RenderOptions as render
 * options (mode RT_MODE_FAST, one band) and the first cameras of the text. */
typedef struct rt_json_scene_info {
    rt_render_opts opts;
    uint64_t n_cameras; /* len(Scene.Camera); cameras[] holds the first 4 */
    uint64_t n_nodes;
    uint64_t n_lights;
    rt_camera_desc cameras[4];
} rt_json_scene_info;
int rt_scene_create_from_json(const char* text, size_t len, rt_scene** out, rt_json_scene_info* info);

/* Replacement for the body of renderScene (render.go:285-349): renders
 * opts->dx x opts->dy pixels with opts->rays_per_pixel samples each and writes
 * 8-bit RGBA (A = 255, row 0 = top) into rgba_out, `stride` bytes per row, the
 * layout of image.RGBA.Pix.  Blocking.  One render in flight per scene handle. */
int rt_render(rt_scene* scene, const rt_camera_desc* camera, const rt_render_opts* opts,
              uint8_t* rgba_out, size_t stride, rt_stats* stats);

/* A frame loop over one resident scene: frame i is rt_render with cameras[i],
 * written to rgba_out + i * frame_stride (rows `stride` bytes apart).  This is
 * the loop of example/charuco (main.go:259-339: one scene, a camera per frame,
 * every frame a full phys.Render) and of the lab's orbit viewer
 * (lab/worker/worker.go:214-267) moved behind the boundary, so that with
 * overlap != 0 the quantised image of frame i crosses PCIe and is copied into
 * the caller's memory (second host thread, second set of film / staging
 * buffers) while frame i + 1 is being traced.  overlap == 0 runs the frames
 * strictly one after the other, as n calls of rt_render would.  The frames are
 * the same bytes either way.  stats (may be NULL) has n_frames entries;
 * render_seconds of an entry is the time to trace and enqueue that frame.
 * On an error (RT_ERR_CANCELLED included) the frames already handed over stay
 * valid and the rest are unspecified. */
int rt_render_frames(rt_scene* scene, const rt_camera_desc* cameras, int32_t n_frames, const rt_render_opts* opts,
                     uint8_t* rgba_out, size_t stride, size_t frame_stride, rt_stats* stats, int32_t overlap);

/* The same render, stopping before quantisation: adds this call's radiance sums
 * into film_dev, a DEVICE buffer of dx*dy*3 float64 (row 0 = top, RGB), which
 * the caller zeroes first.  With band sharding each process fills only its
 * bands, so summing the films of all processes (one RCCL reduce) gives the whole
 * image.  No counterpart in the reference: this is the multi-GPU seam. */
int rt_render_film(rt_scene* scene, const rt_camera_desc* camera, const rt_render_opts* opts,
                   double* film_dev, rt_stats* stats);

/* rt_render_film without a single device allocation, for callers that cannot
 * allocate: tight frame loops, harnesses that forbid hipMalloc, hosts that
 * prepare everything up front.  (The call still synchronises its stream and
 * copies counters to the host, so it cannot itself run inside a stream
 * capture.)  No counterpart in the reference.  Same arguments, same
 * film, same statistics as rt_render_film; the counters, the camera, the queue
 * descriptors and the tile schedule live in a static array in device memory
 * (2.1 MB, held by every GPU the library runs on until the process ends)
 * instead of buffers owned by the scene.  Limits, each rejected with
 * RT_ERR_UNSUPPORTED and a message: the scene must be one whose materials all
 * just emit (Emitter, DebugUV, DebugNormal -- what the OBJ route produces,
 * node.go:269-309; scattering materials need gigabytes of ray queues), it must
 * live on one device, and the frame (this call's row bands of it) must have at
 * most 262144 tiles of 8x8 pixels (4096 x 4096 pixels).  stats->traverse_seconds
 * stays 0 (timing events would have to be created).  All calls on a GPU share
 * that GPU's array: two of them overlapping on one GPU race, whatever their
 * scenes or streams. */
int rt_render_film_noalloc(rt_scene* scene, const rt_camera_desc* camera, const rt_render_opts* opts,
                           double* film_dev, rt_stats* stats);

/* Quantise a summed film exactly as renderPixel does (render.go:218-224):
 * c = sum / rays_per_pixel; byte = uint8(min(255, 255.99 * c)).  rgba_out is a
 * host buffer. */
int rt_film_resolve(const double* film_dev, int32_t dx, int32_t dy, int32_t rays_per_pixel,
                    uint8_t* rgba_out, size_t stride, void* stream);

/* Request cancellation of the render running on `scene` (from another thread).
 * The running rt_render / rt_render_film / rt_render_film_noalloc /
 * rt_render_frames call then returns RT_ERR_CANCELLED; mirrors ctx.Done()
 * (render.go:229-233, 340-343).  The request is looked at before every kernel
 * launch of a frame, once more when a frame's last launch has ended (a frame
 * that is a single launch still reports it) and between the frames of
 * rt_render_frames.  It belongs to the call that is running: every render
 * entry point clears it when it begins, so a request that arrives when no call
 * runs is dropped, as a cancelled context no longer matters once Render has
 * returned. */
void rt_cancel(rt_scene* scene);

/* Closest hit of n rays against the scene, the loop of tracePath over
 * scene.Node (render.go:155-169): nodes in order, Shape.Collide(r, tmin, minDist),
 * keep when c.t < minDist; minDist starts at tmax.  o and d are n x 3 float64
 * host arrays; hits is a host array of n records. */
int rt_intersect_batch(rt_scene* scene, const double* o, const double* d, uint64_t n,
                       double tmin, double tmax, int mode, rt_hit* hits);

/* Primary rays for an explicit list of samples: (s, t) in [0,1]^2 per ray as
 * Camera.Cast receives them (camera.go:12-15), plus the pixel / sample index
 * that key the random stream (FocusableCamera draws a lens point).  Writes
 * n x 3 origins and directions to host arrays.  Test entry point. */
int rt_cast_batch(const rt_camera_desc* camera, const double* st, const uint32_t* pixel_sample,
                  uint64_t n, int64_t seed, double* o_out, double* d_out);

/* ---- validation entry points for materials --------------------------------------------------
 * MicrofacetBRDF.D / G1 / G / F / Evaluate (phys/microfacet.go:28-103) evaluated on the device for n
 * argument sets, so that the reference's own tables (microfacet_test.go:12-443) can be run against
 * the functions the render kernels call.  args holds 4 vectors per set (12 float64), in the order
 * of the Go signatures; a function ignores the ones it does not take.  roughness: n values; f0:
 * n x 3; out: n x 3 (a scalar result in out[3 i]).  All host arrays. */
enum rt_microfacet_fn {
    RT_MF_D = 0,        /* D(h, n)              args: h, n          microfacet.go:28-39 */
    RT_MF_G1 = 1,       /* G1(v, n, h)          args: v, n, h       microfacet.go:44-62 */
    RT_MF_G = 2,        /* G(wo, wi, n, h)      args: wo, wi, n, h  microfacet.go:66-68 */
    RT_MF_F = 3,        /* F(wo, h)             args: wo, h         microfacet.go:73-78 */
    RT_MF_EVALUATE = 4  /* Evaluate(wo, wi, n)  args: wo, wi, n     microfacet.go:83-103 */
};
int rt_microfacet_batch(int which, uint64_t n, const double* roughness, const double* f0, const double* args, double* out);

/* Material.Resolve and Material.ComputeDirectLighting (material.go:28-32; material_lambertian.go:18-81,
 * material_metal.go:28-108, material_dielectric.go:29-172, material_emitter.go:22-30,
 * material_debug_uv.go:18-32, material_debug_normal.go:45-91) for n explicit hits: exactly what the
 * render kernels compute for a queued hit, returned instead of queued.  An rt_shade_in is a
 * surfaceInteraction (material.go:14-19): the incoming ray's direction and radiance, the collision,
 * the node that was hit (its shape is the one the shadow rays skip) and the material to apply, plus
 * the position on the random stream (pixel, sample, path-tree index, draws made so far). */
typedef struct rt_shade_in {
    int32_t material; /* index into rt_scene_desc.materials */
    int32_t node;     /* the node hit: skipped by the occlusion test (material_lambertian.go:37) */
    double d[3];        /* incoming.direction */
    double radiance[3]; /* incoming.radiance  */
    double t, at[3], normal[3], uv[2]; /* collision (shape.go:11-16) */
    uint32_t pixel, sample, path, draws;
} rt_shade_in;
typedef struct rt_shade_out {
    double emission[3];  /* resolution.emission */
    double dir[2][3];    /* resolution.scattered[k].direction (their origin is the hit point) */
    double rad[2][3];    /* resolution.scattered[k].radiance */
    double direct[3];    /* ComputeDirectLighting */
    int32_t n_children;  /* len(resolution.scattered) */
    uint32_t draws_after; /* draws made on the stream once Resolve returned */
    uint32_t shadow_rays; /* occlusion rays cast */
    uint32_t reserved;
} rt_shade_out;
int rt_shade_batch(rt_scene* scene, const rt_shade_in* hits, uint64_t n, int64_t seed, int mode, rt_shade_out* out);

/* Reconstruction filters of phys/reconstruct.go:28-87 (1-D, even, compact support). */
enum rt_recon_filter {
    RT_RECON_BOX = 0,     /* BoxFilter:  radius 0.5, w = 1 for |x| < 0.5              reconstruct.go:31-43 */
    RT_RECON_TENT = 1,    /* TentFilter: radius 1,   w = 1 - |x|                      reconstruct.go:47-60 */
    RT_RECON_MITCHELL = 2 /* MitchellNetravaliFilter, B = C = 1/3, radius 2          reconstruct.go:66-87 */
};

/* ApplySeparableFilterRGBA (reconstruct.go:95-167) on the device: a horizontal pass from the 8-bit
 * image (read as Go's 16-bit RGBA(): v * 0x101, / 65535) into float64, a vertical pass back to
 * 8 bits (uint8(max(0, min(255, 255 * c))), A = 255), edges clamped, weights renormalised per pixel.
 * rgba_in and rgba_out are HOST images of dx x dy pixels with the given strides in bytes (they may
 * be the same buffer). */
int rt_apply_filter(const uint8_t* rgba_in, size_t stride_in, int32_t dx, int32_t dy, int filter,
                    uint8_t* rgba_out, size_t stride_out);

/* Per-mesh BVH statistics of a created scene (nodes, leaves, depth). */
typedef struct rt_bvh_info {
    uint64_t n_faces;
    uint64_t n_nodes;  /* BVH structs in the reference's tree */
    uint64_t n_leaves;
    uint32_t max_depth;
    uint32_t max_leaf_size;
    double build_seconds;
} rt_bvh_info;
int rt_scene_bvh_info(const rt_scene* scene, uint64_t mesh_index, rt_bvh_info* out);

/* The tree NewBVH(shapes, 0) builds over the faces of a mesh (shape_bvh.go:81-272), as a flat array
 * (host code; needs neither a GPU nor rt_init).  It is what a Mesh carries in its BVH field and what
 * (*BVH).MarshalJSON writes (shape_bvh.go:275-297), so a host can export a mesh without building the
 * tree a second time.  Nodes are in depth-first pre-order: the Left child of inner node i is node
 * i + 1, its Right child node `right`.  A node with count > 0 is a leaf BVH{Left == Right}: it holds
 * the faces leaf_faces[first .. first + count), in the order of the Group's Shapes (count == 1: the
 * face itself, no Group, shape_bvh.go:91-97).  nodes has room for node_cap records (2 * n_faces is
 * always enough), leaf_faces for n_faces indices; *n_nodes receives the number of nodes, also when
 * node_cap is too small (RT_ERR_INVALID then, nothing written). */
typedef struct rt_bvh_node {
    double bounds_min[3], bounds_max[3]; /* BVH.bounds (AABB, shape.go:27-30) */
    int32_t right;                       /* inner node: index of the Right child; leaf: -1 */
    int32_t first, count;                /* leaf: range of leaf_faces; inner node: count == 0 */
    int32_t pad;
} rt_bvh_node;
int rt_bvh_build(const rt_face* faces, uint64_t n_faces, rt_bvh_node* nodes, uint64_t node_cap,
                 int32_t* leaf_faces, uint64_t* n_nodes);

/* ---- OBJ geometry scanner (host code; needs neither a GPU nor rt_init) ------------------------------------------
 * The statements obj.ParseFS reads (obj/obj.go:236-468): v, vt, vn, f, usemtl, mtllib, in one pass over the text of a
 * .obj file, for scans of millions of lines.  Indices are resolved as parseIndex / resolveIndex do (obj.go:359-439:
 * 1-based, negative = from the end, 0 = absent).  Faces keep their corners (no triangulation: node.go:121-150 fans
 * them later).  The scanner reads only the well-formed subset; at the first line it is not sure to read like the
 * reference's line parser it stops and sets fallback_line (1-based): the caller must then run the line parser over the
 * whole text, so errors and corner cases are the line parser's.  usemtl / mtllib arguments are returned as byte
 * ranges of the text, in order of appearance. */
enum rt_obj_event_kind { RT_OBJ_USEMTL = 0, RT_OBJ_MTLLIB = 1 };
typedef struct rt_obj_event {
    int32_t kind;        /* rt_obj_event_kind */
    uint32_t line;       /* 1-based line number */
    uint64_t line_off, line_len; /* the statement without surrounding white space */
    uint64_t arg_off, arg_len;   /* what follows the first blank */
} rt_obj_event;
typedef struct rt_obj_scan {
    uint64_t n_vertices, n_texcoords, n_normals, n_faces, n_corners, n_events;
    const double* vertices;      /* 3 per vertex  (obj.go:270-290) */
    const double* texcoords;     /* 2 per entry   (obj.go:293-309) */
    const double* normals;       /* 3 per normal  (obj.go:312-332) */
    const int32_t* corners;      /* 3 per corner: vertex, texcoord, normal index */
    const uint64_t* face_start;  /* n_faces + 1 offsets into corners (in corners, not ints) */
    const int32_t* face_event;   /* per face: index of the usemtl event in force, -1 = none yet */
    const rt_obj_event* events;
    uint64_t fallback_line;      /* 0 = the whole text was read */
    void* owner;                 /* storage of the arrays; release with rt_obj_scan_free */
} rt_obj_scan;
int rt_obj_scan_text(const char* text, uint64_t len, rt_obj_scan* out);
void rt_obj_scan_free(rt_obj_scan* scan);

#ifdef __cplusplus
}
#endif
#endif /* RT_HIP_H */
