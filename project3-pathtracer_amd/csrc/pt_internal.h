// pt_internal.h -- shared between the C-ABI host code (pt_context.hip) and the kernels (pt_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/pt_abi.h"
#include "pt_device.h"

namespace pt {

// Ray pool, SoA of 16-byte lanes: one ray = 40 B = a(16) + b(16) + c(8).  A wave's 64 consecutive rays are
// read/written with two global_load/store_dwordx4 and one dwordx2 per lane, fully coalesced.
//   a = (origin.x, origin.y, origin.z, direction.x)
//   b = (direction.y, direction.z, throughput.r, throughput.g)
//   c = (throughput.b, bit pattern of the tile-local pixel index)
struct RayPool {
    float4 *a;
    float4 *b;
    float2 *c;
};

// Device-resident per-render state.  The iteration counter lives here (not in a kernel argument) so that
// one captured hipGraph can be replayed for every iteration.
// Live-ray counters are sharded: survivors of a bounce are appended to one of NSHARD dense pool segments, each
// with its own reservation counter on its own 128-byte line, so that per-wave reservations do not all queue on
// one address (one global counter saturates near 88 M atomics/s on this chip).
static constexpr int NSHARD = 32;
static constexpr int PT_MAX_BATCH = 16;         // iterations rendered concurrently by one launch sequence
static constexpr int CNT_STRIDE = 32;            // uint32 per counter slot = 128 B
__host__ __device__ constexpr int cnt_index(int bounce, int shard) { return (bounce * NSHARD + shard) * CNT_STRIDE; }

struct IterState {
    uint32_t iter;                               // first iteration of the running batch (1-based, ref: src/main.cpp:95)
    // Batch schedule of the running pt_render call, kept on the device so that ONE captured hipGraph serves every
    // iteration count: the call's iterations are cut into `sched_n` batches of sched_q (+1 for the first sched_r)
    // iterations; k_iter_begin advances `iter` by the previous batch and sets `nslot` for the one that starts.
    uint32_t nslot;                              // iterations in flight in the running batch (1..PT_MAX_BATCH)
    uint32_t sched_q, sched_r, sched_j;          // batch size, batches that carry one more, index of the next batch
    // Launch sequences in flight (pt_options.sequences): every sequence has an IterState of its own and takes every
    // sched_stride-th batch of the call, starting with its own number; a batch's first iteration follows from its index
    uint32_t sched_first, sched_stride;          // first iteration of the pt_render call; batches between two of this sequence's
    // Serial number of the running batch of this sequence (k_iter_begin counts it up; never reset while the radiance planes
    // live): a radiance-plane entry carries the serial of the batch that wrote it, so only NON-ZERO samples are ever written
    // (nineteen of twenty paths end with a zero sample) and nothing has to be cleared between batches
    uint32_t serial;
    uint32_t pad[24];
    uint32_t counts[(PT_MAX_DEPTH + 1) * NSHARD * CNT_STRIDE];   // live rays entering bounce b, per segment
    // resident paths: the waves of the one later-bounce launch DRAW their 64-ray chunks of the pool instead of being dealt equal
    // shares (the SIMDs favour their oldest waves: with equal shares the first workgroup of a CU leaves at 55 % of the launch and
    // the CU runs out its last third with ever fewer waves) -- NSHARD counters, each over its own range of chunks, own 128-B lines
    uint32_t draw[NSHARD * CNT_STRIDE];
    unsigned long long live_in[PT_MAX_DEPTH];    // summed over segments and iterations (stats)
    unsigned long long iterations;
    unsigned long long clk[4];                   // diagnostics: shader-clock / real-time ticks spent by workgroup 0 of the last bounce-1 launch
    unsigned long long shadow_rays;              // shadow rays cast (direct lighting), stats
    unsigned long long dbg[8];                   // hierarchy-walk diagnostics (DEBUG_BVH builds only)
    unsigned long long lane_budget[16];          // -DPT_DEBUG_PHASE=2 builds: clocks / lane-clocks per phase of a later-bounce trip (pair path)
};

// direct lighting: one entry of the light table -- an emissive sphere / cube (tri_count 0, prim = its index) or an emissive
// MESH geom as a whole: its triangles of positive area are light_tris[tri_first .. tri_first + tri_count) (triangle numbers,
// primitive index = number of geoms + that), light_cdf[] beside them holds the running fp32 sum of their areas
struct LightRec { int prim; int tri_first; int tri_count; float area; };

struct KParams {
    // camera basis (host-side part of raycastFromCameraKernel)
    float eye[3], M[3], H[3], V[3];
    float resx, resy;
    float A[3], B[3], vn[3];     // unit screen-right, screen-up and view axes (thin lens)
    float lens_radius, focal_distance;   // lens_radius > 0: depth of field; camera rays then have their own origins
    int W;                 // frame width
    int row_begin;         // first frame row of this context's tile
    int npix;              // pixels in the tile
    uint32_t pix_offset;   // global pixel index of tile-local pixel 0 (= row_begin * W)
    int nG, nM;
    int depth;
    int rr_start;
    uint32_t seed;
    const ptd::Prim *prims;
    const ptd::BvhNode *bvh;   // culling hierarchy over the primitives (GEOM 3), depth-first with skip links
    int nnodes;
    const float *bvh4;     // the same hierarchy collapsed to 4-wide nodes (GEOM_WALK4), ptd::W4_FLOATS floats per node
    int nnodes4;
    int wdepth;            // its depth in nodes (bounds the traversal stack)
    int ntri;              // triangle records (type 3) among the nG primitives: the flattened MESH geoms, behind the geoms
    int nbig;              // primitives too large to cull (walls...): tested by every ray before the walk
    int big[16];           // their indices
    const float *face_n;   // per primitive: 8 float4, entry `face code` = world normal of that box face (boxNormal's result)
    const float *ro_eye;   // per primitive: inverseTransform*(eye,1) as float4 (camera rays share their origin)
    int eye_cull;          // 1 = camera-ray waves skip primitives outside their boxes (box_eye), ablation switch
    const float *box_world; // per primitive: padded world box (lo.xyz,0)(hi.xyz,0): per-lane pre-test of the pair queue
    const float *box_eye;  // per primitive: padded world box minus the eye, (lo.xyz,0)(hi.xyz,0): wave cull of camera rays
    const float *slab_n;   // pair path: per primitive (n.xyz, d_lo) of the slab the pre-test clips rays of tilted cubes against (d_hi: box[7])
    uint32_t slab_mask;    // bit g: primitive g (< 32) has such a slab
    const uint32_t *span_off;   // batched walks, camera rays: per span (offset, count) into span_list; count 0xFFFFFFFF = walk (nullptr: no lists)
    const uint32_t *span_list;  // primitive indices
    const uint32_t *span_mask;  // pair path, camera rays: per 64 tile-local pixels, bit g = primitive g can be seen from them (nullptr: no table)
    const float *mats;     // M_PLANES planes of nM floats
    float *image;          // tile framebuffer, fp32 RGB packed (12 B/pixel)
    int cull;              // 1 = skip primitives whose bounding sphere no lane of the wave can hit (large scenes)
    int nslot;             // most iterations in flight per launch sequence (1..PT_MAX_BATCH); a batch's own count is IterState::nslot
    float *lbuf;           // nslot planes of npix entries (r, g, b, serial of the batch that wrote them: 16 B), folded into `image` by k_accumulate
    int nshard;            // pool segments in use: NSHARD (compaction 1), 1 otherwise
    uint32_t segcap;       // slots per pool segment
    IterState *st;
    RayPool pool[2];
    // interleaved-strip tiles: local pixel pl -> global pixel pl + strip_span*(j*(strip_world-1) + strip_rank), j = pl/strip_span
    uint32_t strip_span;   // W * strip_rows, 0 = contiguous band (pix_offset)
    uint32_t strip_rows, strip_world, strip_rank;
    uint32_t strip_magic;  // j = (pl * strip_magic) >> strip_shift, exact for pl < 2^28
    uint32_t strip_shift;
    uint32_t w_magic;      // y = (global pixel * w_magic) >> w_shift, exact for every pixel index of the frame (< 2^29)
    uint32_t w_shift;
    int absorption;        // 1 = Beer-Lambert absorption inside refractive objects (material planes M_AR..M_AB)
    int scatter;           // 1 = subsurface random walk inside SCATTER materials (material planes M_SCAT, M_RSCT)
    int nlights;           // direct lighting: entries of the light table (0 = feature off)
    const LightRec *lights;
    const int *light_tris;     // mesh lights: triangle numbers and the running sum of their areas, per entry
    const float *light_cdf;
    int ngeoms;            // geoms among the nG primitives (= nG - ntri): primitive index of triangle t = ngeoms + t
    // motion blur with a shutter time per ray (FEAT_MOTION kernels): nknots scene states at shutter times k / (nknots - 1)
    const float *knots;    // [nknots][nG][12]: transform rows 0..2 of every geom at every knot (the inverse is computed per ray)
    const float *knot_cam; // [nknots][12]: camera position, view, up (xyz each, padded to 4)
    int nknots;            // 0 = off
    float tan_x, tan_y;    // tan of the half field-of-view angles (the per-ray camera basis is built on the device)
    int refill_min;        // resident paths (FEAT_RESIDENT kernels): a wave refills its free lanes from the pool once this many are free
};

struct LaunchCfg {
    int workgroup;   // 64..1024
    int grid;        // workgroups per bounce launch
    int geom;        // 0 scalar direct, 1 LDS direct, 2 hit queue, 3 per-lane hierarchy walk, 4 pair queue, 5 walk + pairs,
                     // 6 batched 4-wide walk + pairs, 7 the same with the nodes in global memory (pt_bounce.h)
    int compact;     // 0 off, 1 per-wave sharded reservation, 2 workgroup scan + single counter
    int nee;         // 1 = explicit light sampling at diffuse vertices (compact must be 1)
    int media;       // 1 = subsurface random walk inside SCATTER materials (compact must be 1, workgroup 256 or 512)
    int motion;      // 1 = per-ray shutter time (geom 0, workgroup 256, compact 1, neither nee nor media)
    int resident;    // 1 = bounces 1 .. depth - 1 in ONE launch, paths resident in registers (geom 4, 6, 7; workgroup 256 / 512; compact 1;
                     //     none of nee / media / motion)
    int slab;        // 1 = the pair path's pre-test also clips tilted cubes against the slab of their thinnest axis (KParams::slab_mask != 0;
                     //     geom 4, workgroup 256 / 512, compact 1, none of nee / media / motion): kernel instances of their own, so that scenes
                     //     without such cubes keep the shorter loop
};

// kernels (pt_kernels.hip)
hipError_t launch_iter_set(hipStream_t s, IterState *st, uint32_t iter_first, uint32_t q, uint32_t r, uint32_t j0, uint32_t stride);
hipError_t launch_iter_begin(hipStream_t s, IterState *st, int npix, int depth, int compact);
hipError_t launch_accumulate(hipStream_t s, float *image, const float *lbuf, const IterState *st, int npix);
hipError_t launch_iter_fold(hipStream_t s, IterState *st, int depth);
hipError_t launch_bounce(hipStream_t s, const KParams &p, const LaunchCfg &cfg, int bounce);
hipError_t launch_send_image_to_pbo(hipStream_t s, pt_uchar4 *pbo, const float *image, int npix);
hipError_t launch_selftest_math(hipStream_t s, unsigned long long *out3);
hipError_t launch_device_kat(hipStream_t s, int op, const float *in, float *out, int n_out);
// error reporting shared by the C-ABI translation units: records the message behind pt_last_error(), returns `code`
int fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));

size_t bounce_lds_bytes(const KParams &p, const LaunchCfg &cfg);      // (cfg.resident: of the resident-path kernel)
int bounce_max_blocks_per_cu(const KParams &p, const LaunchCfg &cfg);
bool bounce_resident_available(const LaunchCfg &cfg);                  // is there a resident-path kernel for this launch shape?

}  // namespace pt
