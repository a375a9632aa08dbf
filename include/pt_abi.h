/*
 * pt_abi.h -- C-ABI of the MI355X-native path-tracing renderer (libptamd.so).
 *
 * This is the drop-in boundary for the reference's renderer entry point
 *     void cudaRaytraceCore(uchar4* pos, camera* renderCam, int frame, int iterations,
 *                           material* materials, int numberOfMaterials, geom* geoms, int numberOfGeoms);
 * (ref: src/raytraceKernel.h:17, body src/raytraceKernel.cu:106-165).  That symbol is C++-mangled and its
 * `camera` argument holds a std::string (ref: src/sceneStructs.h:50-61), so it cannot itself be a C-ABI.
 * The reference-side binding is a ~40-line C++ shim with that exact signature which flattens the
 * arguments exactly as src/raytraceKernel.cu:123-146 does and calls the functions below
 * (shown in INTEGRATION.md, shipped as project3-pathtracer_amd/csrc/pt_shim.cpp).
 *
 * Everything here is POD: plain pointers, ints, floats.  No torch, HIP or C++ types in any signature
 * (the stream / device pointers are passed as void*).  All functions return PT_OK (0) or a negative
 * pt_status; pt_last_error() gives the message of the calling thread's last failure.  Nothing below the
 * shim ever calls exit() (the reference's checkCUDAError does, ref: src/raytraceKernel.cu:19-25; the shim
 * keeps that behaviour for drop-in parity).
 *
 * There is NO CPU fallback: without a usable gfx950 device pt_create() fails with PT_ERR_NO_DEVICE.
 */
#ifndef PT_ABI_H
#define PT_ABI_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PT_ABI_VERSION 2     /* 2: pt_options grew (sequences, motion_per_ray, resident); pt_abi_version / pt_options_size added */
#define PT_MAX_DEPTH 64
#define PT_MAX_SEQUENCES 4   /* launch sequences in flight per context (pt_options.sequences) */

typedef enum pt_status {
    PT_OK = 0,
    PT_ERR_INVALID = -1,      /* bad argument / call order */
    PT_ERR_NO_DEVICE = -2,    /* no gfx950 device or HIP runtime failure at create */
    PT_ERR_HIP = -3,          /* a HIP call failed; message in pt_last_error() */
    PT_ERR_OOM = -4
} pt_status;

/* ---- POD layouts: byte-identical to the reference's device structs (x86-64), so a caller can pass its
 *      own arrays unchanged.  Sizes/offsets are pinned by tests/test_abi.py against
 *      tests/golden/reference_vectors.json "layouts". ---- */
typedef struct pt_vec2 { float x, y; } pt_vec2;
typedef struct pt_vec3 { float x, y, z; } pt_vec3;
typedef struct pt_vec4 { float x, y, z, w; } pt_vec4;
typedef struct pt_mat4 { pt_vec4 x, y, z, w; } pt_mat4;             /* cudaMat4: four ROWS, ref: src/cudaMat4.h:18-23 */
typedef struct pt_uchar4 { unsigned char x, y, z, w; } pt_uchar4;   /* PBO texel, ref: src/raytraceKernel.cu:83-87 */

enum { PT_SPHERE = 0, PT_CUBE = 1, PT_MESH = 2 };                   /* GEOMTYPE, ref: src/sceneStructs.h:14 */

typedef struct pt_static_geom {                                     /* staticGeom, ref: src/sceneStructs.h:32-40, 172 B */
    int type;
    int materialid;
    pt_vec3 translation, rotation, scale;                           /* carried, never read by the renderer */
    pt_mat4 transform, inverseTransform;
} pt_static_geom;

typedef struct pt_material {                                        /* material, ref: src/sceneStructs.h:63-74, 64 B */
    pt_vec3 color;
    float specularExponent;
    pt_vec3 specularColor;
    float hasReflective;
    float hasRefractive;
    float indexOfRefraction;
    float hasScatter;
    pt_vec3 absorptionCoefficient;
    float reducedScatterCoefficient;
    float emittance;
} pt_material;

typedef struct pt_camera_data {                                     /* cameraData, ref: src/sceneStructs.h:42-48, 52 B */
    pt_vec2 resolution;
    pt_vec3 position, view, up;
    pt_vec2 fov;                                                    /* degrees, half-angles (ref: src/scene.cpp:204-207) */
} pt_camera_data;

/* Knobs the reference keeps as source constants (trace depth, ref: src/raytraceKernel.cu:110) or does not
 * have at all.  pt_default_options() fills the defaults. */
typedef struct pt_options {
    int depth;            /* bounces per path, 1..PT_MAX_DEPTH (default 8) */
    int rr_start;         /* first bounce with Russian roulette, <0 = off (default -1) */
    unsigned seed;        /* RNG stream selector (default 0) */
    int compaction;       /* live-ray compaction after every bounce: 1 = per-wave reservation in 32 pool segments, no
                             barrier (default); 2 = workgroup LDS scan + one counter; 0 = off, rays keep their slot */
    int workgroup;        /* threads per workgroup: 64, 128, 256, 512 or 1024 (default 0 = library choice) */
    int geom_path;        /* how primitives reach the lanes: 0 = library choice (default: 5 up to 40 primitives, else 7),
                             1 = scalar (SGPR) loads, 2 = staged in LDS, 3 = scalar candidate test + wave-private LDS hit
                             queue, 4 = per-lane walk of an LDS-resident bounding-box hierarchy (large scenes), 5 = per-lane
                             box pre-test + wave-private queue of (ray, primitive) pairs, exact test on full batches,
                             6 = hierarchy walk with box tests only that queues the leaves it reaches as pairs (large scenes),
                             7 = 4-wide hierarchy walked in full 64-entry batches of (ray, node) entries from a wave-private
                             LDS stack, leaves queued as pairs (large scenes), 8 = the same with the nodes read through L1/L2
                             instead of an LDS copy (hierarchies too large for the LDS).  Scenes with triangles
                             (pt_set_meshes) run on 1, 7 or 8 */
    int row_begin;        /* tile rendered by this context: rows [row_begin, row_end) of the frame; */
    int row_end;          /*   0,0 = the whole frame.  RNG streams are keyed on the global pixel index. */
    int use_graph;        /* 1 = replay one captured hipGraph per launch sequence (default), 0 = eager launches */
    int batch;            /* iterations rendered concurrently by one launch sequence, 1..16 (default 0 = library choice 16);
                             their samples are folded into the running mean in iteration order, so the image does not
                             depend on it */
    int direct_light;     /* 1 = sample the lights explicitly at every diffuse vertex (one shadow ray; the reference's
                             getRandomPointOnCube / getRandomPointOnSphere samplers, ref: src/intersections.h:133-182; an
                             emissive MESH geom is one light: area-weighted triangle pick + uniform point) and do not count
                             a light of the table hit by chance after such a vertex; 0 = pure path tracing (default).
                             The table holds the first 16 emissive geoms; further emitters are reached by chance only.
                             Needs compaction 1. */
    int absorption;       /* 1 = Beer-Lambert absorption (material ABSCOEFF) over path segments that end on the inner side of
                             a refractive surface: calculateTransmission, ref stub src/interactions.h:31-33; 0 = off (default) */
    int strip_rows;       /* > 0: interleaved row strips instead of one band -- the frame is cut into strips of strip_rows */
    int strip_world;      /*   rows, strip k belongs to context k % strip_world, and this context (strip_rank) renders its */
    int strip_rank;       /*   strips packed in order (balances ranks when path lengths vary down the frame); row_begin =
                               row_end = 0 then.  pt_strip_local_rows / pt_strip_global_row give the mapping.  0 = off */
    int scatter;          /* 1 = subsurface random walk inside SCATTER materials that are not mirrors (calculateScatterAndAbsorption,
                             ref stub src/interactions.h:36-39): free flight -ln(1-u)/RSCTCOEFF, isotropic re-direction
                             (getRandomDirectionInSphere), Beer-Lambert with ABSCOEFF along every segment inside; the surface is
                             a dielectric when REFR is set, index-matched otherwise; 0 = off (default) */
    float lens_radius;    /* > 0: thin-lens camera (depth of field): rays start on a disc of this radius around the eye and */
    float focal_distance; /*   aim at the pinhole ray's point on the plane focal_distance along the view axis; 0 = pinhole */
    int sequences;        /* launch sequences in flight, 1..PT_MAX_SEQUENCES (default 0 = library choice): batch n + 1 of a
                             pt_render call renders on a second stream, with ray pools of its own, while batch n does -- its
                             launches fill the compute units the tail of every bounce launch leaves idle; only the accumulates
                             are ordered (iteration order), so the image does not depend on it */
    int motion_per_ray;   /* motion blur (pt_set_motion): 0 = one scene state per run of 16 iterations (default, every scene);
                             1 = a shutter time PER RAY: every path draws its time as the third number of its camera stream
                             and sees, at all bounces, the transforms (and camera vectors) interpolated entry-wise between the two
                             of `slices` + 1 knot states around it, with their inverses computed from them -- exact for
                             translations, a chord approximation of
                             rotations that tightens with `slices`.  Runs on the pair queue (up to 40 primitives, or geom_path 5:
                             pre-test against boxes swept over the shutter interval) and on the scalar loop (geom_path 1), with
                             direct lighting (lights sampled where they are at the ray's time) and scattering; excludes meshes */
    int resident;         /* the later bounces of a batch as ONE launch with the paths resident in registers (a wave keeps the
                             paths that go on, refills the lanes of those that ended from the camera launch's ray pool): 1 = on
                             where a kernel exists for the launch shape (pair queue and batched walks, i.e. geom_path 0 / 5 / 7 / 8,
                             without motion_per_ray, depth >= 3), -1 = off (one launch per bounce), 0 = library choice (default: on
                             from depth 5 on for the plain estimator, where it pays; with direct_light / scatter the instances exist
                             but measure - 3 ... + 1 %, so those stay on one launch per bounce unless asked).  The image does not
                             depend on it */
} pt_options;

/* Device memory of a context, per launch sequence in flight: two ray pools of 40 B per ray slot (batch x tile pixels slots + up
 * to 16 % of segment slack) and the radiance planes, 16 B per (iteration in flight, pixel) -- 6.5 GB per sequence for a 1920x1080
 * tile at batch 16, about 26 GB per sequence at the cap of 2^28 rays per launch.  The library default is two sequences; when that
 * allocation fails the context falls back to one before pt_render reports PT_ERR_OOM.  pt_get_launch_info tells what it took.
 *
 * Environment variables the library reads (none changes a result; all are read when a context configures unless noted):
 *   PT_RESIDENT=1|0, PT_SEQUENCES=n     the library's choice where pt_options.resident / .sequences leave it open (== 0) -- an
 *                                       explicit option always wins
 *   PT_REFILL_MIN=n                     resident paths: free lanes that trigger a wave's refill (1..64, default 8)
 *   PT_NO_SELF_SKIP, PT_NO_SLAB         ablations of the pair reductions of DESIGN.md 5.1 (a resident path skipping the primitive it just
 *                                       left; tilted cubes clipped against the slab of their thinnest axis)
 *   PT_NO_NOISE_PAD                     ablation: culling bounds without the reach-based pads (rounds 1-3's bounds -- NOT conservative for
 *                                       spheres far smaller than the rays that reach them are long, or scenes ~ 1e6 units wide: DESIGN.md 2)
 *   PT_MAX_WG_PER_CU, PT_EXTRA_LDS, PT_NO_CULL, PT_NO_EYE_CULL     launch-shape / culling ablations behind DESIGN.md's sweeps
 *   PT_DEBUG_CLOCK, PT_DEBUG_PHASE, PT_DEBUG_PHASE2, PT_DEBUG_PAIR, PT_DEBUG_W4, PT_DEBUG_SPAN, PT_DEBUG_BOUNDS
 *                                       print diagnostics at pt_get_stats (the counters exist in diagnostic builds only)
 *   PT_SERIAL_BUDGET=n                  TEST HOOK (read once per process): batches a context renders before the radiance planes'
 *                                       serial numbers start over (default 2^29) -- lets tests reach the restart */
typedef struct pt_stats {
    unsigned long long iterations;             /* iterations rendered since create / pt_reset_stats */
    unsigned long long ray_bounces;            /* sum over iterations and bounces of live rays entering the bounce */
    unsigned long long live_in[PT_MAX_DEPTH];  /* the same, per bounce */
    double gpu_ms;                             /* HIP-event time of all pt_render calls on the render stream */
    unsigned long long bounce_launches;        /* per-bounce kernel launches inside those calls */
    unsigned long long shadow_rays;            /* shadow rays cast by direct lighting (not part of ray_bounces) */
} pt_stats;

typedef struct pt_ctx pt_ctx;

/* interleaved-strip tiles (pt_options.strip_*): rows a rank owns, and the frame row of its local row */
int  pt_strip_local_rows(int height, int strip_rows, int world, int rank);
int  pt_strip_global_row(int strip_rows, int world, int rank, int local_row);

/* ---- lifetime ---- */
int  pt_device_count(void);
int  pt_create(int device, pt_ctx **out);             /* persistent device context (replaces the per-call
                                                         cudaMalloc/cudaFree of ref: src/raytraceKernel.cu:118-158) */
void pt_destroy(pt_ctx *ctx);
const char *pt_last_error(void);
const char *pt_version(void);
/* What the loaded library was built against: a client compares them with its own PT_ABI_VERSION and sizeof(pt_options)
 * before the first pt_set_options (the struct travels by pointer, without a size: a client built against an older header
 * would hand over a shorter one).  The Python binding and the shim check both when they load the library. */
int  pt_abi_version(void);
size_t pt_options_size(void);

/* ---- inputs (replaces ref: src/raytraceKernel.cu:123-146) ---- */
void pt_default_options(pt_options *opt);
int  pt_set_options(pt_ctx *ctx, const pt_options *opt);
int  pt_get_options(pt_ctx *ctx, pt_options *opt);
int  pt_set_scene(pt_ctx *ctx, const pt_static_geom *geoms, int numberOfGeoms,
                  const pt_material *materials, int numberOfMaterials);
int  pt_set_camera(pt_ctx *ctx, const pt_camera_data *cam);
/* Triangles of the MESH geoms (the reference's loader names a .obj file per MESH object and loads nothing,
 * ref: src/scene.cpp:57-66; its intersection header leaves the triangle test as an option, src/intersections.h:79).
 * vertices: 9 floats per triangle (v0, v1, v2) in the object space the geom's transform places.  Copied; call after
 * pt_set_scene (which drops the meshes of the previous scene).  Triangles are tested after the geoms, in the order
 * given: primitive index = numberOfGeoms + running triangle number. */
typedef struct pt_mesh { int geom; int n_triangles; const float *vertices; } pt_mesh;
int  pt_set_meshes(pt_ctx *ctx, const pt_mesh *meshes, int numberOfMeshes);
int  pt_set_stream(pt_ctx *ctx, void *hip_stream);    /* render on a caller-owned hipStream_t; NULL = own stream */

/* Motion blur (the reference keeps per-frame TRANS / ROTAT / SCALE and camera arrays, ref: src/sceneStructs.h:21-30,50-61,
 * and lists motion blur among the features to build, README.md:72-81): the shutter stays open from the frame given to
 * pt_set_scene / pt_set_camera to the NEXT frame given here.  `slices` scene states are built at shutter times
 * (k + .5)/slices by interpolating translation / rotation / scale (and the camera's position / view / up) component-wise,
 * matrices rebuilt as the loader does (rotat_units: PT_ROTAT_*); iterations are dealt to the slices in runs of
 * PT_SLICE_ITERATIONS: iteration i renders slice ((i - 1) / PT_SLICE_ITERATIONS) % slices, so every launch sequence still
 * sees one static scene and the running mean converges to the time average.  cam_next = NULL: the camera is at rest.
 * With pt_options.motion_per_ray the same call gives the two frames and `slices` (>= 1) is the number of linear segments
 * between slices + 1 knot states at shutter times k / slices.
 * geoms_next = NULL, slices < 1 (or slices == 1 without motion_per_ray) turns it off.  Call after pt_set_scene /
 * pt_set_camera (pt_set_scene drops it). */
#define PT_SLICE_ITERATIONS 16
int  pt_set_motion(pt_ctx *ctx, const pt_static_geom *geoms_next_frame, const pt_camera_data *cam_next_frame_or_null,
                   int slices, int rotat_units);

/* ---- framebuffer: fp32 RGB, 12 B/pixel, index = x + y*W (ref: src/raytraceKernel.cu:98), tile rows only.
 *      It always holds the running mean over the iterations rendered so far. ---- */
size_t pt_image_bytes(pt_ctx *ctx);
int  pt_bind_image(pt_ctx *ctx, void *device_rgb);    /* render into caller-owned device memory (>= pt_image_bytes);
                                                         NULL = library-owned buffer */
int  pt_image_device_pointer(pt_ctx *ctx, void **device_rgb_out);   /* where the tile's framebuffer lives on the device */
int  pt_clear_image(pt_ctx *ctx);                     /* clearImage, ref: src/raytraceKernel.cu:48-55 */
int  pt_upload_image(pt_ctx *ctx, const float *host_rgb);    /* H2D, ref: src/raytraceKernel.cu:120 */
int  pt_download_image(pt_ctx *ctx, float *host_rgb);        /* D2H (synchronous), ref: src/raytraceKernel.cu:154 */

/* ---- the hot path ---- */
/* Enqueue iterations [iter_first, iter_first + iter_count) (1-based, as ref: src/main.cpp:95 passes them):
 * camera rays (raycastFromCameraKernel), per-bounce intersect/shade (raytraceRay) and live-ray compaction.
 * Asynchronous on the render stream. */
int  pt_render(pt_ctx *ctx, int iter_first, int iter_count);
/* sendImageToPBO (ref: src/raytraceKernel.cu:58-89): fp32 RGB x255, clamp, RGBA8 into a DEVICE buffer of
 * tile pixels.  Asynchronous on the render stream. */
int  pt_send_image_to_pbo(pt_ctx *ctx, pt_uchar4 *device_pbo);
int  pt_synchronize(pt_ctx *ctx);                     /* cudaThreadSynchronize, ref: src/raytraceKernel.cu:162 */
/* Record a caller-owned hipEvent_t on the render stream, behind everything enqueued so far: lets a host order streams of its
 * own behind a render without waiting on the CPU (the multi-device gather does). */
int  pt_record_event(pt_ctx *ctx, void *hip_event);

/* One reference-style iteration: optional H2D of host_image_inout when iteration > 1 and the context has no
 * accumulated state, render `iteration`, optional PBO, D2H into host_image_inout (if non-NULL), synchronize. */
int  pt_render_iteration(pt_ctx *ctx, pt_uchar4 *device_pbo_or_null, float *host_image_inout_or_null, int iteration);

/* ---- measurement ---- */
int  pt_get_stats(pt_ctx *ctx, pt_stats *out);        /* synchronizes the render stream */
int  pt_reset_stats(pt_ctx *ctx);
/* The launch shape the library chose for the current scene / camera / options (configures the context if needed):
 * what bench.py reports instead of guessing from the options and the environment. */
typedef struct pt_launch_info {
    int geom_path;        /* pt_options.geom_path numbering (1..8) of the path actually used */
    int workgroup;        /* threads per workgroup */
    int grid;             /* workgroups per bounce launch (persistent grid) */
    int batch;            /* iterations in flight per launch sequence */
    int sequences;        /* launch sequences in flight */
    int resident;         /* 1 = the later bounces run as ONE launch with the paths resident in registers */
    int refill_min;       /*   free lanes that trigger a wave's refill there */
    int launches_per_batch;   /* bounce-kernel launches per batch: depth, or 2 with resident paths */
    int lds_bytes;        /* dynamic LDS per workgroup of the later-bounce kernel */
    int slab_pretest;     /* 1 = the pair queue's pre-test also clips tilted cubes against the slab of their thinnest axis (kernel instances of
                             their own, taken where a cube of the scene has such a slab: DESIGN.md 5.1); was reserved[0], the size is unchanged */
    int reserved[6];
} pt_launch_info;
int  pt_get_launch_info(pt_ctx *ctx, pt_launch_info *out);

/* Like pt_render (eager launches), but every per-bounce kernel launch is bracketed by its own pair of HIP
 * events on the render stream; bounce_ms_out[b] (depth entries) receives the summed duration of bounce b's
 * launches over the rendered iterations (resident paths: [0] the camera launch, [1] the one launch of all later bounces, the
 * rest 0).  Synchronous.  For roofline accounting, not for throughput. */
int  pt_render_profiled(pt_ctx *ctx, int iter_first, int iter_count, double *bounce_ms_out);
/* Device self-test: the kernels' short correctly-rounded sqrt / reciprocal / reciprocal-sqrt sequences against
 * the compiler's general ones for ALL 2^32 fp32 inputs.  mismatches_out[0..2] must come back 0. */
int  pt_selftest_math(pt_ctx *ctx, unsigned long long mismatches_out[3]);

/* Known-answer tests of single DEVICE functions (one GPU thread evaluates the kernels' own implementation of
 * a reference function), so that GPU results can be pinned directly to the reference's golden vectors.
 * `in` / `out` are host arrays of fp32; integers travel as bit patterns.  Transforms are 16 floats, row-major
 * rows x,y,z,w (cudaMat4).  Synchronous. */
enum {
    PT_KAT_HASH = 1,            /* in: a (bits)                                  out: hash(a) (bits)      ref src/intersections.h:26-34 */
    PT_KAT_U01_SEQUENCE = 2,    /* in: engine seed (bits)                        out: n_out u01 draws      thrust minstd_rand + uniform_real */
    PT_KAT_NOISE = 3,           /* in: resx, resy, time, x, y                    out: rgb                  ref src/raytraceKernel.cu:29-36 */
    PT_KAT_INTERSECT = 4,       /* in: type (bits), transform[16], inverse[16], o[3], d[3]   out: t, p[3], n[3]   ref src/intersections.h:72-117 */
    PT_KAT_HEMISPHERE = 5,      /* in: n[3], xi1, xi2                            out: dir[3]               ref src/interactions.h:62-87 */
    PT_KAT_RADIUSES = 6,        /* in: transform[16]                             out: radii[3]             ref src/intersections.h:120-129 */
    PT_KAT_POINT_ON_CUBE = 7,   /* in: transform[16], seed                       out: p[3]                 ref src/intersections.h:133-175 */
    PT_KAT_POINT_ON_SPHERE = 8, /* in: transform[16], seed                       out: p[3]                 ref src/intersections.h:177-182 */
    PT_KAT_MULTIPLY_MV = 9,     /* in: m[16], v[4]                               out: r[3]                 ref src/intersections.h:53-59 */
    PT_KAT_POINT_ON_RAY = 10,   /* in: o[3], d[3], t                             out: p[3]                 ref src/intersections.h:46-48 */
    PT_KAT_REFLECT = 11,        /* in: normal[3], incident[3]                    out: dir[3]               ref src/interactions.h:47-50 */
    PT_KAT_REFRACT = 12,        /* in: normal[3], incident[3], n1, n2            out: dir[3] (0 on TIR)    ref src/interactions.h:42-44 */
    PT_KAT_FRESNEL = 13,        /* in: normal[3], incident[3], n1, n2, trans[3]  out: reflectance          ref src/interactions.h:53-59 */
    PT_KAT_TRANSMISSION = 14,   /* in: absorption[3], distance                   out: rgb transmittance    ref src/interactions.h:31-33 */
    PT_KAT_SAMPLE_LIGHT = 15,   /* in: type (bits), transform[16], seed          out: p[3], n[3]           direct lighting sampler */
    PT_KAT_LOG = 16,            /* in: x                                         out: ln(x)                deterministic log of the free-flight sampler */
    PT_KAT_SCATTER = 17,        /* in: o[3], d[3], depth, absorption[3], rsct, T[3], u1, u2, u3
                                   out: scattered (0/1), o[3], d[3], depth, T[3]                           ref src/interactions.h:36-39 */
    PT_KAT_SAMPLE_TRIANGLE = 18 /* in: v0[3], e1[3], e2[3], u_a, u_b                 out: p[3], area         mesh-light sampler (direct lighting) */
};
int  pt_device_kat(pt_ctx *ctx, int op, const float *in, int n_in, float *out, int n_out);

/* ---- scene files (ref: src/scene.cpp, src/utilities.cpp:74-90; format README.md:160-217) ---- */
enum { PT_ROTAT_RADIANS = 0,   /* what the reference binary does (GLM_FORCE_RADIANS, ref: src/utilities.cpp:7) */
       PT_ROTAT_DEGREES = 1 }; /* what the scene author evidently meant */
typedef struct pt_scene pt_scene;
int  pt_scene_load(const char *path, int rotat_units, pt_scene **out);
void pt_scene_free(pt_scene *s);
int  pt_scene_counts(const pt_scene *s, int *n_objects, int *n_materials, int *n_camera_frames);
int  pt_scene_camera_info(const pt_scene *s, unsigned *iterations, char *image_name, size_t image_name_cap);
/* flatten frame `frame` into the POD arrays above (what ref: src/raytraceKernel.cu:123-146 does) */
int  pt_scene_get_frame(const pt_scene *s, int frame, pt_static_geom *geoms_out, pt_material *materials_out,
                        pt_camera_data *camera_out);
/* RES override: recomputes fov.x from fov.y as the loader does (ref: src/scene.cpp:204-207) */
int  pt_camera_set_resolution(pt_camera_data *cam, int width, int height);
/* triangles of MESH object `object` as read from its .obj file (object space; NULL / 0 when there are none) */
int  pt_scene_mesh(const pt_scene *s, int object, const float **vertices_out, int *n_triangles_out);

/* ---- several devices of one node behind one handle (single process): device k renders the k-th band of rows,
 *      no communication while rendering, bands gathered to the host or to one device over xGMI peer copies.
 *      `devices` may name a device more than once (each entry gets its own context). ---- */
typedef struct pt_multi pt_multi;
int  pt_multi_create(const int *devices, int n, pt_multi **out);
void pt_multi_destroy(pt_multi *m);
int  pt_multi_count(const pt_multi *m);
int  pt_multi_set_options(pt_multi *m, const pt_options *opt);      /* row_begin/row_end are set per band */
int  pt_multi_set_scene(pt_multi *m, const pt_static_geom *geoms, int numberOfGeoms,
                        const pt_material *materials, int numberOfMaterials);
int  pt_multi_set_camera(pt_multi *m, const pt_camera_data *cam);
int  pt_multi_set_meshes(pt_multi *m, const pt_mesh *meshes, int numberOfMeshes);
int  pt_multi_set_motion(pt_multi *m, const pt_static_geom *geoms_next_frame, const pt_camera_data *cam_next_frame_or_null,
                         int slices, int rotat_units);
int  pt_multi_band(const pt_multi *m, int k, int *row_begin, int *row_end);
int  pt_multi_clear_image(pt_multi *m);
int  pt_multi_upload_image(pt_multi *m, const float *host_rgb_full_frame);
int  pt_multi_render(pt_multi *m, int iter_first, int iter_count);  /* asynchronous on every device */
int  pt_multi_synchronize(pt_multi *m);
int  pt_multi_download_image(pt_multi *m, float *host_rgb_full_frame);
int  pt_multi_gather_to_device(pt_multi *m, void *device_rgb_full_frame, int dst_device);
/* the same, returning as soon as every device's copies are enqueued on its copy stream (behind its render stream, no host
 * wait); pt_multi_synchronize -- or the next gather -- joins.  One gather in flight per handle. */
int  pt_multi_gather_to_device_async(pt_multi *m, void *device_rgb_full_frame, int dst_device);
int  pt_multi_gather_times(pt_multi *m, double *enqueue_ms, double *total_ms);   /* host clock of the last device gather */
/* peer access between two devices as the handle found it (1 = enabled: copies cross xGMI directly, 0 = none: the runtime stages) */
int  pt_multi_peer_access(pt_multi *m, int src_device, int dst_device, int *direct);
int  pt_multi_set_strips(pt_multi *m, int strip_rows);   /* > 0: interleaved strips (device k: strips k, k+n, ...); 0: bands */
int  pt_multi_send_image_to_pbo(pt_multi *m, pt_uchar4 *device_pbo);  /* single-device handles only */
int  pt_multi_get_stats(pt_multi *m, pt_stats *out);                /* sums over devices (gpu_ms: max) */

/* ---- end-of-render image write-out (ref: src/main.cpp:116-141, src/image.cpp:41-88): horizontal flip
 *      buffer (x,y) -> picture (W-1-x, y), gamma 1.0, clamp(v*255, 0, 255) truncation ---- */
int  pt_image_to_rgb8(const float *host_rgb, int width, int height, int flip_x, unsigned char *rgb8_out);
int  pt_save_image_bmp(const char *path, const float *host_rgb, int width, int height, int flip_x);
int  pt_save_image_png(const char *path, const float *host_rgb, int width, int height, int flip_x);   /* 8-bit RGB, stored deflate */
int  pt_save_image(const char *path, const float *host_rgb, int width, int height, int flip_x);       /* "...bmp" -> BMP, else PNG
                                                                                                         (ref: src/image.cpp:68-87) */

#ifdef __cplusplus
}
#endif
#endif /* PT_ABI_H */
