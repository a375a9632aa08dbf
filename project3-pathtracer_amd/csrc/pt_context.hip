// pt_context.hip -- host side of the C-ABI (include/pt_abi.h): persistent device context, scene/camera
// flattening, launch sequencing, hipGraph capture, event timing.
//
// Replaces the host body of cudaRaytraceCore (ref: src/raytraceKernel.cu:106-165): instead of
// cudaMalloc + H2D + 2 launches + D2H + cudaFree on every iteration, the context keeps the framebuffer,
// ray pools and scene resident and enqueues 1 + depth launches per iteration with no host round trip.
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <vector>

#include "pt_internal.h"
#include "pt_scene.h"

// library choices of the resident-path launch (pt_options.resident == 0; measurements in DESIGN.md section 5.1)
#ifndef PT_RESIDENT_DEFAULT
#define PT_RESIDENT_DEFAULT 1
#endif
#ifndef PT_REFILL_MIN_DEFAULT
#define PT_REFILL_MIN_DEFAULT 8
#endif

namespace {
thread_local std::string g_last_error;
}

int pt::fail(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}

namespace {

using pt::fail;

#define HIP_TRY(expr)                                                                                  \
    do {                                                                                               \
        hipError_t e__ = (expr);                                                                       \
        if (e__ != hipSuccess)                                                                         \
            return fail(PT_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, __LINE__); \
    } while (0)

struct v3 { float x, y, z; };
v3 cross3(v3 x, v3 y) { return {x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y}; }
v3 normalize3(v3 v)
{
    float inv = 1.0f / sqrtf(v.x * v.x + v.y * v.y + v.z * v.z);
    return {v.x * inv, v.y * inv, v.z * inv};
}
float length3(v3 v) { return sqrtf(v.x * v.x + v.y * v.y + v.z * v.z); }

}  // namespace

struct pt_ctx {
    int device = 0;
    int cu_count = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;

    pt_options opt;
    bool have_scene = false, have_camera = false;
    std::vector<pt_static_geom> geoms;
    std::vector<pt_material> mats;
    pt_camera_data cam;
    // triangles of the MESH geoms (pt_set_meshes), object space: 9 floats each + the geom they belong to
    std::vector<float> tri_obj;
    std::vector<int> tri_geom;
    // motion blur (pt_set_motion): one child context per shutter slice, all rendering into THIS context's framebuffer
    // on this context's stream; pt_render hands every run of 16 iterations to the slice it belongs to
    std::vector<pt_static_geom> geoms_next;
    pt_camera_data cam_next;
    bool have_cam_next = false;
    int motion_slices = 0, motion_rotat = 0;
    std::vector<pt_ctx *> slice_ctx;
    bool motion_dirty = false;

    // device state
    bool dirty = true;          // scene / camera / options changed since the last configure
    pt::KParams kp;             // sequence 0's parameters (what everything outside the launch sequencing reads)
    pt::KParams kps[PT_MAX_SEQUENCES];   // per launch sequence: own ray pools, radiance planes and IterState
    int nseq = 1;               // launch sequences in flight (pt_options.sequences)
    hipStream_t seq_stream[PT_MAX_SEQUENCES] = {};   // [0] = `stream`; [k > 0] own streams
    hipEvent_t seq_acc[PT_MAX_SEQUENCES] = {};       // recorded behind a sequence's accumulate
    hipEvent_t seq_fork = nullptr;
    pt::LaunchCfg cfg;
    ptd::Prim *d_prims = nullptr;
    float *d_mats = nullptr;
    float *d_ro_eye = nullptr;
    float *d_face_n = nullptr;
    float *d_box_eye = nullptr;
    std::vector<float> h_box_eye;          // host copy (the span table of the camera rays is built from it)
    std::vector<ptd::BvhNode> h_bvh;       // host copies for the per-span primitive lists of the batched walks
    std::vector<float> h_boxes;            //   padded world boxes (lo.xyz, hi.xyz) by expanded primitive index
    uint32_t *d_span_off = nullptr, *d_span_list = nullptr;
    size_t span_off_cap = 0, span_list_cap = 0;
    uint32_t *d_span_mask = nullptr;
    size_t span_mask_cap = 0;
    float *d_box_world = nullptr;
    float *d_slab_n = nullptr;          // per primitive (n.xyz, d_lo) of the pre-test's slab (KParams::slab_mask)
    pt::LightRec *d_lights = nullptr;    // direct lighting: the light table
    float *d_knots = nullptr, *d_knot_cam = nullptr;   // per-ray motion blur: knot states (matrices, camera vectors)
    int *d_light_tris = nullptr;         //   mesh lights: triangle numbers ...
    float *d_light_cdf = nullptr;        //   ... and the running sums of their areas
    ptd::BvhNode *d_bvh = nullptr;
    float *d_bvh4 = nullptr;    // 4-wide hierarchy (geom_path 7)
    float *d_image_own = nullptr;
    float *d_image_bound = nullptr;
    size_t image_bytes = 0, image_cap = 0;
    void *d_pool = nullptr;
    size_t pool_cap = 0;
    float *d_lbuf = nullptr;    // per-iteration radiance planes of the running batch
    size_t lbuf_cap = 0;
    unsigned long long batches_stamped = 0;   // batches rendered since the radiance planes were last zeroed (serial-number budget)
    int batch = 1;              // iterations in flight per launch sequence
    pt::IterState *d_state = nullptr;          // PT_MAX_SEQUENCES of them
    bool image_valid = false;   // framebuffer holds iterations 1..k of the current frame

    // one batch of one sequence (bookkeeping, depth bounce launches; with a single sequence also the accumulate); the
    // batch's iteration count lives on the device, so this graph serves every pt_render
    // slots [0, PT_MAX_SEQUENCES): a sequence's batch without the accumulate; slot PT_MAX_SEQUENCES: sequence 0's with it
    hipGraph_t graph[PT_MAX_SEQUENCES + 1] = {};
    hipGraphExec_t graph_exec[PT_MAX_SEQUENCES + 1] = {};

    std::vector<std::pair<hipEvent_t, hipEvent_t>> timers;
    double gpu_ms = 0.0;
    unsigned long long bounce_launches = 0;
};

namespace {

void drop_graph(pt_ctx *c)
{
    for (int q = 0; q <= PT_MAX_SEQUENCES; ++q) {
        if (c->graph_exec[q]) { (void)hipGraphExecDestroy(c->graph_exec[q]); c->graph_exec[q] = nullptr; }
        if (c->graph[q]) { (void)hipGraphDestroy(c->graph[q]); c->graph[q] = nullptr; }
    }
}

// Every stream the context may have work on: the render stream and the side streams of the launch sequences.  Whoever needs the
// context idle (configure, statistics, image copies, the error paths of pt_render after its fork) waits for all of them, so a
// call that failed between fork and join cannot leave a side stream running against buffers the next call reuses.
hipError_t sync_all_streams(pt_ctx *c)
{
    hipError_t first = hipStreamSynchronize(c->stream);
    for (int sq = 1; sq < PT_MAX_SEQUENCES; ++sq)
        if (c->seq_stream[sq]) {
            const hipError_t e = hipStreamSynchronize(c->seq_stream[sq]);
            if (first == hipSuccess) first = e;
        }
    return first;
}

int fold_timers(pt_ctx *c)
{
    for (auto &t : c->timers) {
        float ms = 0.0f;
        HIP_TRY(hipEventSynchronize(t.second));
        HIP_TRY(hipEventElapsedTime(&ms, t.first, t.second));
        c->gpu_ms += (double)ms;
        (void)hipEventDestroy(t.first);
        (void)hipEventDestroy(t.second);
    }
    c->timers.clear();
    return PT_OK;
}

float *image_ptr(pt_ctx *c) { return c->d_image_bound ? c->d_image_bound : c->d_image_own; }

// ---- culling hierarchy over the primitives (host build; the kernels only read it) -------------------------
struct Aabb { float lo[3], hi[3]; };

// What the REFERENCE'S sphere test hits is not the sphere.  Its radicand, b*b - (ro.ro - 0.25) with b = ro.rd, is formed from fp32
// products of object-space coordinates (ref: src/intersections.h:81-117; the outer subtraction alone is in double): for a ray that
// starts R object units away both terms are ~ R^2 and carry ~ 16 ulp-halves of it, so the test reports a hit whenever the ray passes
// within sqrt(0.25 + E) of the centre, E <= 20 eps R^2 + 64 eps (s_max / s_min) R (the second term: rounding of the transformed
// direction, seen from R away), eps = 2^-24 -- and its hit point lies within that radius too.  A sphere of radius 0.11 met from 15 world
// units away is 1 % larger than it is; one of radius 0.01 across a room of 100 is hit by rays that pass several radii away.  The oracle
// renders exactly that, so every culling bound of a sphere is taken around the INFLATED sphere: this returns sqrt(1 + 4 E) for rays
// that start at most D world units from the centre (configure() bounds D by the scene).  Cubes: the slab distances err by a few ulps
// of themselves, i.e. by ~ 1e-7 D in world units whatever the scale: an absolute pad.
double sphere_noise_factor(const pt_static_geom &g, double D)
{
    if (!(D > 0.0)) return 1.0;
    const float *r0 = &g.transform.x.x, *r1 = &g.transform.y.x, *r2 = &g.transform.z.x;
    double len[3], col[3][3], F2 = 0.0;
    for (int j = 0; j < 3; ++j) {
        col[j][0] = r0[j]; col[j][1] = r1[j]; col[j][2] = r2[j];
        len[j] = sqrt(col[j][0] * col[j][0] + col[j][1] * col[j][1] + col[j][2] * col[j][2]);
        F2 += len[j] * len[j];
    }
    bool orthogonal = true;
    for (int a = 0; a < 3; ++a)
        for (int b = a + 1; b < 3; ++b)
            if (fabs(col[a][0] * col[b][0] + col[a][1] * col[b][1] + col[a][2] * col[b][2]) > 1e-4 * len[a] * len[b]) orthogonal = false;
    double smin, smax;
    if (orthogonal) {
        smin = std::min(len[0], std::min(len[1], len[2])) * (1.0 - 2e-4);
        smax = std::max(len[0], std::max(len[1], len[2])) * (1.0 + 2e-4);
    } else {                                  // singular values: s_max <= ||M||_F, s_min = |det| / (s_1 s_2) >= 2 |det| / ||M||_F^2
        const double det = col[0][0] * (col[1][1] * col[2][2] - col[1][2] * col[2][1]) - col[1][0] * (col[0][1] * col[2][2] - col[0][2] * col[2][1]) +
                           col[2][0] * (col[0][1] * col[1][2] - col[0][2] * col[1][1]);
        smax = sqrt(F2);
        smin = 2.0 * fabs(det) / F2;
    }
    const double eps = 5.9604644775390625e-8, FMAX = 1e6;
    if (!(smin > 0.0) || !(smax > 0.0)) return FMAX;
    const double R = D / smin + 0.5;
    const double E = 20.0 * eps * R * R + 64.0 * eps * (smax / smin) * R;
    const double f = sqrt(1.0 + 4.0 * E);
    return f < FMAX ? f : FMAX;                 // (a NaN compares false: FMAX)
}

// Padded world-space box of one primitive: |x_a - c_a| <= 0.5 * sum_i |M[a][i]| for the unit cube and
// <= 0.5 * sqrt(sum_i M[a][i]^2) for the r = .5 sphere (M = rows 0..2 of the transform); +1 % and +1e-3 absorb the
// fp32 rounding of the traversal's slab test and the 1e-4 back-off of getPointOnRay.  D: how far from the primitive's centre a ray
// tested against it can start (0: not known yet) -- sphere_noise_factor above.
Aabb prim_bounds(const pt_static_geom &g, double rel = 1.01, double abs_pad = 1e-3, double D = 0.0)
{
    const float *rows[3] = {&g.transform.x.x, &g.transform.y.x, &g.transform.z.x};
    Aabb b;
    const double noise = (g.type == PT_SPHERE) ? sphere_noise_factor(g, D) : 1.0;
    abs_pad += 2e-6 * D;
    for (int a = 0; a < 3; ++a) {
        const double m0 = rows[a][0], m1 = rows[a][1], m2 = rows[a][2], c = rows[a][3];
        double h = (g.type == PT_SPHERE) ? 0.5 * sqrt(m0 * m0 + m1 * m1 + m2 * m2) * noise : 0.5 * (fabs(m0) + fabs(m1) + fabs(m2));
        h = h * rel + abs_pad + 1e-6 * fabs(c);
        b.lo[a] = (float)(c - h);
        b.hi[a] = (float)(c + h);
    }
    return b;
}

void build_bvh(const std::vector<Aabb> &boxes, std::vector<int> &idx, size_t lo, size_t hi, std::vector<ptd::BvhNode> &nodes)
{
    const size_t me = nodes.size();
    nodes.push_back(ptd::BvhNode());
    Aabb u = boxes[(size_t)idx[lo]];
    float cmin[3], cmax[3];
    for (int a = 0; a < 3; ++a) cmin[a] = cmax[a] = 0.5f * (u.lo[a] + u.hi[a]);
    for (size_t k = lo + 1; k < hi; ++k) {
        const Aabb &b = boxes[(size_t)idx[k]];
        for (int a = 0; a < 3; ++a) {
            if (b.lo[a] < u.lo[a]) u.lo[a] = b.lo[a];
            if (b.hi[a] > u.hi[a]) u.hi[a] = b.hi[a];
            const float cc = 0.5f * (b.lo[a] + b.hi[a]);
            if (cc < cmin[a]) cmin[a] = cc;
            if (cc > cmax[a]) cmax[a] = cc;
        }
    }
    if (hi - lo == 1) {
        nodes[me].prim = idx[lo];          // the caller ORs the primitive type into bit 30
    } else {
        // surface-area heuristic, full sweep: for every axis sort by centroid and take the split that minimises
        // area(left)*count(left) + area(right)*count(right)  (the hierarchy only culls, so any split is correct)
        const size_t n = hi - lo;
        auto half_area = [](const Aabb &b) {
            const double dx = (double)b.hi[0] - b.lo[0], dy = (double)b.hi[1] - b.lo[1], dz = (double)b.hi[2] - b.lo[2];
            return dx * dy + dy * dz + dz * dx;
        };
        auto grow = [](Aabb &u2, const Aabb &b) {
            for (int a = 0; a < 3; ++a) {
                if (b.lo[a] < u2.lo[a]) u2.lo[a] = b.lo[a];
                if (b.hi[a] > u2.hi[a]) u2.hi[a] = b.hi[a];
            }
        };
        double best_cost = 1e300;
        int best_axis = -1;
        size_t best_left = n / 2;
        auto by_centroid = [&](int axis) {
            return [&boxes, axis](int x, int y) {
                return boxes[(size_t)x].lo[axis] + boxes[(size_t)x].hi[axis] < boxes[(size_t)y].lo[axis] + boxes[(size_t)y].hi[axis];
            };
        };
        if (n <= 4096) {
            // full sweep; only (axis, split position) of the best split are kept during it -- the winning order is rebuilt
            // by one more sort afterwards (stable_sort of the same input: the same order)
            std::vector<int> order(n);
            std::vector<double> right_area(n);
            for (int axis = 0; axis < 3; ++axis) {
                if (!(cmax[axis] > cmin[axis])) continue;
                std::copy(idx.begin() + (long)lo, idx.begin() + (long)hi, order.begin());
                std::stable_sort(order.begin(), order.end(), by_centroid(axis));
                Aabb acc = boxes[(size_t)order[n - 1]];
                for (size_t k = n - 1; k >= 1; --k) {                 // right_area[k] = area of order[k..n)
                    grow(acc, boxes[(size_t)order[k]]);
                    right_area[k] = half_area(acc);
                }
                acc = boxes[(size_t)order[0]];
                for (size_t k = 1; k < n; ++k) {                      // split: left = order[0..k), right = order[k..n)
                    grow(acc, boxes[(size_t)order[k - 1]]);
                    const double cost = half_area(acc) * (double)k + right_area[k] * (double)(n - k);
                    if (cost < best_cost) { best_cost = cost; best_axis = axis; best_left = k; }
                }
            }
            if (best_axis >= 0) std::stable_sort(idx.begin() + (long)lo, idx.begin() + (long)hi, by_centroid(best_axis));
        } else {
            // large nodes (triangle meshes): binned surface-area heuristic, 32 centroid bins per axis, O(n) per node
            constexpr int NB = 32;
            int best_bin = -1;
            for (int axis = 0; axis < 3; ++axis) {
                if (!(cmax[axis] > cmin[axis])) continue;
                Aabb bb[NB];
                size_t cnt[NB] = {0};
                const double scale = (double)NB / ((double)cmax[axis] - (double)cmin[axis]);
                auto bin_of = [&](int x) {
                    const double cc = 0.5 * ((double)boxes[(size_t)x].lo[axis] + (double)boxes[(size_t)x].hi[axis]);
                    int b = (int)((cc - (double)cmin[axis]) * scale);
                    return b < 0 ? 0 : (b >= NB ? NB - 1 : b);
                };
                for (size_t k = lo; k < hi; ++k) {
                    const int b = bin_of(idx[k]);
                    if (cnt[b]++ == 0) bb[b] = boxes[(size_t)idx[k]]; else grow(bb[b], boxes[(size_t)idx[k]]);
                }
                double ra[NB];
                size_t rc[NB];
                Aabb acc{};
                size_t c2 = 0;
                for (int b = NB - 1; b >= 1; --b) {                  // right side = bins [b, NB)
                    if (cnt[b]) { if (c2 == 0) acc = bb[b]; else grow(acc, bb[b]); c2 += cnt[b]; }
                    ra[b] = c2 ? half_area(acc) : 0.0;
                    rc[b] = c2;
                }
                c2 = 0;
                for (int b = 1; b < NB; ++b) {                       // left side = bins [0, b)
                    if (cnt[b - 1]) { if (c2 == 0) acc = bb[b - 1]; else grow(acc, bb[b - 1]); c2 += cnt[b - 1]; }
                    if (c2 == 0 || rc[b] == 0) continue;
                    const double cost = half_area(acc) * (double)c2 + ra[b] * (double)rc[b];
                    if (cost < best_cost) { best_cost = cost; best_axis = axis; best_bin = b; best_left = c2; }
                }
            }
            if (best_axis >= 0) {
                const int axis = best_axis;
                const double scale = (double)NB / ((double)cmax[axis] - (double)cmin[axis]);
                std::stable_partition(idx.begin() + (long)lo, idx.begin() + (long)hi, [&](int x) {
                    const double cc = 0.5 * ((double)boxes[(size_t)x].lo[axis] + (double)boxes[(size_t)x].hi[axis]);
                    int b = (int)((cc - (double)cmin[axis]) * scale);
                    b = b < 0 ? 0 : (b >= NB ? NB - 1 : b);
                    return b < best_bin;
                });
            }
        }
        const size_t mid = lo + best_left;
        nodes[me].prim = -1;
        build_bvh(boxes, idx, lo, mid, nodes);
        build_bvh(boxes, idx, mid, hi, nodes);
    }
    for (int a = 0; a < 3; ++a) { nodes[me].lo[a] = u.lo[a]; nodes[me].hi[a] = u.hi[a]; }
    nodes[me].skip = (uint32_t)nodes.size();          // first node after this subtree
}


// 4-wide hierarchy for geom_path 7, collapsed from the binary one (depth-first, skip links: the left child of internal
// node i is i + 1, the right child is nodes[i + 1].skip): a wide node takes the two children of a binary node and keeps
// replacing the child with the largest box by that child's own two children until it has four (or only leaves are left).
// Record = ptd::W4_FLOATS floats: per axis the four children's lo planes then their hi planes (x: [0..8), y: [8..16),
// z: [16..24)), then four child words (bit 31 leaf, bit 30 cube, bit 29 triangle, bits 9..28 node / primitive index; the low
// nine bits stay free for the owner lane and the three direction signs of a traversal entry).  An empty child has lo = +3e38, hi = -3e38: no ray passes.
int build_wide4(const std::vector<ptd::BvhNode> &bin, const std::vector<int> &ptype, int b, int depth,
                std::vector<float> &out, int &maxdepth)
{
    auto is_leaf = [&](int i) { return bin[(size_t)i].prim >= 0; };
    auto area = [&](int i) {
        const ptd::BvhNode &n = bin[(size_t)i];
        const double dx = (double)n.hi[0] - n.lo[0], dy = (double)n.hi[1] - n.lo[1], dz = (double)n.hi[2] - n.lo[2];
        return dx * dy + dy * dz + dz * dx;
    };
    std::vector<int> kids;
    if (is_leaf(b)) kids.push_back(b);                    // a one-primitive hierarchy: the root is that leaf
    else { kids.push_back(b + 1); kids.push_back((int)bin[(size_t)b + 1].skip); }
    while (kids.size() < 4) {
        int best = -1;
        for (size_t k = 0; k < kids.size(); ++k)
            if (!is_leaf(kids[k]) && (best < 0 || area(kids[k]) > area(kids[(size_t)best]))) best = (int)k;
        if (best < 0) break;
        const int n = kids[(size_t)best];
        kids[(size_t)best] = n + 1;
        kids.push_back((int)bin[(size_t)n + 1].skip);
    }
    const int me = (int)(out.size() / ptd::W4_FLOATS);
    out.resize(out.size() + ptd::W4_FLOATS, 0.0f);
    if (depth > maxdepth) maxdepth = depth;
    for (int c = 0; c < 4; ++c) {
        uint32_t word = 0u;
        float lo[3] = {3e38f, 3e38f, 3e38f}, hi[3] = {-3e38f, -3e38f, -3e38f};
        if (c < (int)kids.size()) {
            const ptd::BvhNode &n = bin[(size_t)kids[(size_t)c]];
            for (int a = 0; a < 3; ++a) { lo[a] = n.lo[a]; hi[a] = n.hi[a]; }
            if (n.prim >= 0) {
                const uint32_t prim = (uint32_t)n.prim & 0x3FFFFFFFu;
                word = 0x80000000u | (ptype[prim] == PT_CUBE ? 0x40000000u : 0u) | (ptype[prim] == 3 ? 0x20000000u : 0u) | (prim << 9);
            } else {
                word = (uint32_t)build_wide4(bin, ptype, kids[(size_t)c], depth + 1, out, maxdepth) << 9;
            }
        }
        float *rec = out.data() + (size_t)me * ptd::W4_FLOATS;       // (re-fetched: the recursion grows the vector)
        for (int a = 0; a < 3; ++a) { rec[8 * a + c] = lo[a]; rec[8 * a + 4 + c] = hi[a]; }
        memcpy(&rec[24 + c], &word, 4);
    }
    return me;
}

// motion blur modes: one scene state per run of 16 iterations (child contexts), or a shutter time per ray
bool motion_by_slices(const pt_ctx *c) { return c->motion_slices > 1 && !c->opt.motion_per_ray && !c->geoms_next.empty(); }
bool motion_per_ray(const pt_ctx *c) { return c->motion_slices >= 1 && c->opt.motion_per_ray && !c->geoms_next.empty(); }
float lerp1(float a, float b, float t);
pt_vec3 lerp3(pt_vec3 a, pt_vec3 b, float t);

// (re)build everything that depends on scene, camera or options
int configure(pt_ctx *c)
{
    if (!c->have_scene) return fail(PT_ERR_INVALID, "pt_set_scene has not been called");
    if (!c->have_camera) return fail(PT_ERR_INVALID, "pt_set_camera has not been called");
    HIP_TRY(hipSetDevice(c->device));
    if (!c->dirty) return PT_OK;
    HIP_TRY(sync_all_streams(c));
    drop_graph(c);

    const pt_options &o = c->opt;
    const int W = (int)c->cam.resolution.x, H = (int)c->cam.resolution.y;
    if (W < 1 || H < 1) return fail(PT_ERR_INVALID, "camera resolution %dx%d", W, H);
    int r0 = o.row_begin, r1 = o.row_end;
    if (r0 == 0 && r1 == 0) r1 = H;
    if (r0 < 0 || r1 > H || r0 >= r1) return fail(PT_ERR_INVALID, "tile rows [%d,%d) outside frame height %d", r0, r1, H);
    int tile_rows = r1 - r0;
    if (o.strip_rows > 0) {
        tile_rows = pt_strip_local_rows(H, o.strip_rows, o.strip_world, o.strip_rank);
        if (tile_rows < 1) return fail(PT_ERR_INVALID, "rank %d of %d owns no strip of %d rows in a frame of height %d",
                                       o.strip_rank, o.strip_world, o.strip_rows, H);
    }
    const long long npix_ll = (long long)W * (long long)tile_rows;
    if ((long long)W * (long long)H > 0x7FFFFFFFLL / 4) return fail(PT_ERR_INVALID, "frame %dx%d too large", W, H);
    const int npix = (int)npix_ll;

    pt::KParams &k = c->kp;
    memset(&k, 0, sizeof k);

    // camera basis: pixel-independent part of raycastFromCameraKernel (DESIGN.md "Canonical semantics", camera)
    {
        const v3 eye = {c->cam.position.x, c->cam.position.y, c->cam.position.z};
        const v3 view = {c->cam.view.x, c->cam.view.y, c->cam.view.z};
        const v3 up = {c->cam.up.x, c->cam.up.y, c->cam.up.z};
        const v3 A = normalize3(cross3(view, up));
        const v3 B = normalize3(cross3(A, view));
        const float lenV = length3(view);
        const double PI = 3.1415926535897932384626422832795028841971;
        const float tx = (float)tan((double)c->cam.fov.x * (PI / 180.0));
        const float ty = (float)tan((double)c->cam.fov.y * (PI / 180.0));
        const float hx = lenV * tx, vy = lenV * ty;
        k.eye[0] = eye.x; k.eye[1] = eye.y; k.eye[2] = eye.z;
        k.M[0] = eye.x + view.x; k.M[1] = eye.y + view.y; k.M[2] = eye.z + view.z;
        k.H[0] = hx * A.x; k.H[1] = hx * A.y; k.H[2] = hx * A.z;
        k.V[0] = vy * B.x; k.V[1] = vy * B.y; k.V[2] = vy * B.z;
        k.resx = c->cam.resolution.x; k.resy = c->cam.resolution.y;
        const v3 vn = normalize3(view);
        k.A[0] = A.x; k.A[1] = A.y; k.A[2] = A.z;
        k.B[0] = B.x; k.B[1] = B.y; k.B[2] = B.z;
        k.vn[0] = vn.x; k.vn[1] = vn.y; k.vn[2] = vn.z;
        k.lens_radius = o.lens_radius > 0.0f ? o.lens_radius : 0.0f;
        k.focal_distance = o.focal_distance;
    }
    k.W = W;
    k.row_begin = r0;
    k.npix = npix;
    k.pix_offset = (uint32_t)r0 * (uint32_t)W;
    {
        // division by the frame width as multiply + shift: s = 29 + ceil(log2 W), m = ceil(2^s / W) < 2^30, and
        // m*W - 2^s < W <= 2^(s-29), so the quotient is exact for every index < 2^29 (the frame has < 2^29 pixels)
        uint32_t lg = 0;
        while ((1ull << lg) < (unsigned long long)W) ++lg;
        k.w_shift = 29u + lg;
        k.w_magic = (uint32_t)(((1ull << k.w_shift) + (unsigned long long)W - 1ull) / (unsigned long long)W);
    }
    if (o.strip_rows > 0) {
        k.strip_rows = (uint32_t)o.strip_rows; k.strip_world = (uint32_t)o.strip_world; k.strip_rank = (uint32_t)o.strip_rank;
        k.strip_span = (uint32_t)W * (uint32_t)o.strip_rows;
        // division by strip_span as multiply + shift, exact for every tile-local pixel index (< 2^28):
        // s = 28 + ceil(log2 span), m = ceil(2^s / span) < 2^29, and m*span - 2^s < span <= 2^(s-28)
        uint32_t lg = 0;
        while ((1ull << lg) < (unsigned long long)k.strip_span) ++lg;
        k.strip_shift = 28u + lg;
        k.strip_magic = (uint32_t)(((1ull << k.strip_shift) + k.strip_span - 1ull) / k.strip_span);
    }
    // triangles of the MESH geoms in world space (12 floats each: v0, e1, e2, unit normal), evaluated with multiplyMV's
    // operation order; they are primitives of their own behind the geoms: index = number of geoms + triangle number
    const size_t nGeoms = c->geoms.size(), nT = c->tri_geom.size(), nP = nGeoms + nT;
    std::vector<float> triw(nT * 12);
    for (size_t t = 0; t < nT; ++t) {
        const pt_mat4 &m = c->geoms[(size_t)c->tri_geom[t]].transform;
        float w[3][3];
        for (int j = 0; j < 3; ++j) {
            const float x = c->tri_obj[9 * t + 3 * (size_t)j], y = c->tri_obj[9 * t + 3 * (size_t)j + 1], z = c->tri_obj[9 * t + 3 * (size_t)j + 2];
            w[j][0] = (m.x.x * x) + (m.x.y * y) + (m.x.z * z) + (m.x.w * 1.0f);
            w[j][1] = (m.y.x * x) + (m.y.y * y) + (m.y.z * z) + (m.y.w * 1.0f);
            w[j][2] = (m.z.x * x) + (m.z.y * y) + (m.z.z * z) + (m.z.w * 1.0f);
        }
        float *o12 = &triw[12 * t];
        const v3 e1 = {w[1][0] - w[0][0], w[1][1] - w[0][1], w[1][2] - w[0][2]}, e2 = {w[2][0] - w[0][0], w[2][1] - w[0][1], w[2][2] - w[0][2]};
        const v3 cr = cross3(e1, e2);
        const float l2 = cr.x * cr.x + cr.y * cr.y + cr.z * cr.z;
        const v3 n = (l2 > 0.0f) ? normalize3(cr) : v3{0, 0, 0};
        o12[0] = w[0][0]; o12[1] = w[0][1]; o12[2] = w[0][2];
        o12[3] = e1.x; o12[4] = e1.y; o12[5] = e1.z;
        o12[6] = e2.x; o12[7] = e2.y; o12[8] = e2.z;
        o12[9] = n.x; o12[10] = n.y; o12[11] = n.z;
    }
    // centre and radius of a sphere around every point a ray can start from (set by the reach computation below) and, from them, how
    // far from triangle t a ray tested against it can start
    double scene_c0[3] = {0, 0, 0}, scene_rs = 0.0;
    bool scene_known = false;
    auto tri_reach = [&](size_t t) -> double {
        if (!scene_known) return 0.0;
        const float *w = &triw[12 * t];
        const double cx = w[0] + ((double)w[3] + w[6]) / 3.0, cy = w[1] + ((double)w[4] + w[7]) / 3.0, cz = w[2] + ((double)w[5] + w[8]) / 3.0;
        const double dx = cx - scene_c0[0], dy = cy - scene_c0[1], dz = cz - scene_c0[2];
        return (sqrt(dx * dx + dy * dy + dz * dz) + scene_rs) * 1.002;
    };
    // padded world box of triangle t (culling only), same padding rule as prim_bounds: + 2e-6 x reach for what the reference's test and
    // the kernels' slab tests lose on a ray that starts that far away (a unit triangle seen from 1e6 units is fuzzy by ~ 0.1)
    auto tri_bounds = [&](size_t t, double rel, double abs_pad) {
        const float *w = &triw[12 * t];
        Aabb b;
        abs_pad += 2e-6 * tri_reach(t);
        for (int a = 0; a < 3; ++a) {
            const double p0 = w[a], p1 = (double)w[a] + w[3 + a], p2 = (double)w[a] + w[6 + a];
            const double lo = std::min(p0, std::min(p1, p2)), hi = std::max(p0, std::max(p1, p2));
            const double cc = 0.5 * (lo + hi), h = 0.5 * (hi - lo) * rel + abs_pad + 1e-6 * fabs(cc);
            b.lo[a] = (float)(cc - h);
            b.hi[a] = (float)(cc + h);
        }
        return b;
    };
    // reach[i]: how far from the centre of geom i a ray that is tested against it can start -- the eye (the lens around it) or a point
    // of (the padded bounds of) any primitive, at either end of a motion.  The bounds of spheres depend on it (sphere_noise_factor), so
    // this is a fixed point: the inflation is ~ 1e-3 of the reach for ordinary shapes and the iteration settles at once; a scene where it
    // does not (spheres squeezed by factors of hundreds) gets the largest reach the factor knows, and culls nothing of them.
    std::vector<double> reach(nGeoms, 0.0);
    {
        bool settled = false;
        for (int pass = 0; pass < 8 && !settled; ++pass) {
            double lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
            bool first = true;
            auto grow = [&](const double *l, const double *h) {
                for (int a = 0; a < 3; ++a) {
                    if (first || l[a] < lo[a]) lo[a] = l[a];
                    if (first || h[a] > hi[a]) hi[a] = h[a];
                }
                first = false;
            };
            auto grow_box = [&](const Aabb &b) {
                const double l[3] = {b.lo[0], b.lo[1], b.lo[2]}, h[3] = {b.hi[0], b.hi[1], b.hi[2]};
                grow(l, h);
            };
            auto grow_eye = [&](const pt_camera_data &cam) {
                const double r = o.lens_radius > 0.0f ? (double)o.lens_radius * 1.001 : 0.0;
                const double l[3] = {cam.position.x - r, cam.position.y - r, cam.position.z - r};
                const double h[3] = {cam.position.x + r, cam.position.y + r, cam.position.z + r};
                grow(l, h);
            };
            grow_eye(c->cam);
            if (c->have_cam_next) grow_eye(c->cam_next);
            for (size_t i = 0; i < nGeoms; ++i) {
                if (c->geoms[i].type == PT_MESH) continue;
                grow_box(prim_bounds(c->geoms[i], 1.01, 1e-3, reach[i]));
                if (!c->geoms_next.empty()) grow_box(prim_bounds(c->geoms_next[i], 1.01, 1e-3, reach[i]));
            }
            for (size_t t = 0; t < nT; ++t) grow_box(tri_bounds(t, 1.01, 1e-3));
            const double c0[3] = {0.5 * (lo[0] + hi[0]), 0.5 * (lo[1] + hi[1]), 0.5 * (lo[2] + hi[2])};
            const double rs = 0.5 * sqrt((hi[0] - lo[0]) * (hi[0] - lo[0]) + (hi[1] - lo[1]) * (hi[1] - lo[1]) + (hi[2] - lo[2]) * (hi[2] - lo[2]));
            scene_c0[0] = c0[0]; scene_c0[1] = c0[1]; scene_c0[2] = c0[2];
            scene_rs = rs * 1.002;             // (the triangles' own pads, 2e-6 of it, are inside the 0.2 %)
            settled = true;
            for (size_t i = 0; i < nGeoms; ++i) {
                auto dist = [&](const pt_static_geom &g) {
                    const double dx = g.transform.x.w - c0[0], dy = g.transform.y.w - c0[1], dz = g.transform.z.w - c0[2];
                    return sqrt(dx * dx + dy * dy + dz * dz);
                };
                double d = dist(c->geoms[i]);
                if (!c->geoms_next.empty()) d = std::max(d, dist(c->geoms_next[i]));
                d += rs;
                if (!(d <= reach[i])) { settled = false; reach[i] = d * 1.002; }          // (a NaN never settles)
            }
        }
        if (!settled) { for (size_t i = 0; i < nGeoms; ++i) reach[i] = 1e9; scene_rs = 1e9; }
        scene_known = true;
        if (getenv("PT_NO_NOISE_PAD")) { for (size_t i = 0; i < nGeoms; ++i) reach[i] = 0.0; scene_known = false; }      // (ablation: rounds 1-3's bounds)
    }
    k.nG = (int)nP;
    k.ntri = (int)nT;
    k.nM = (int)c->mats.size();
    k.depth = o.depth;
    k.rr_start = o.rr_start;
    k.seed = o.seed;

    // scene -> device records
    {
        std::vector<ptd::Prim> prims(nP ? nP : 1);
        memset(prims.data(), 0, prims.size() * sizeof(ptd::Prim));
        for (size_t i = 0; i < c->geoms.size(); ++i) {
            const pt_static_geom &g = c->geoms[i];
            ptd::Prim &p = prims[i];
            p.type = (uint32_t)g.type;
            p.material = (uint32_t)g.materialid;
            memcpy(p.inv, &g.inverseTransform, 12 * sizeof(float));
            memcpy(p.fwd, &g.transform, 12 * sizeof(float));
            // multiplyMV(transform, (0,0,0,1)) with the reference's operation order (ref: src/intersections.h:53-59,111)
            const pt_mat4 &m = g.transform;
            p.cx = (m.x.x * 0.0f) + (m.x.y * 0.0f) + (m.x.z * 0.0f) + (m.x.w * 1.0f);
            p.cy = (m.y.x * 0.0f) + (m.y.y * 0.0f) + (m.y.z * 0.0f) + (m.y.w * 1.0f);
            p.cz = (m.z.x * 0.0f) + (m.z.y * 0.0f) + (m.z.z * 0.0f) + (m.z.w * 1.0f);
            // padded bounding sphere: every point of the unit cube / r=.5 sphere lies within half the sum of the
            // transform's column lengths of the centre; +2 % and +1e-3 absorb fp32 rounding of the cull test
            double len[3], colv[3][3], sum = 0.0, sumsq = 0.0, maxlen = 0.0;
            for (int col = 0; col < 3; ++col) {
                const float *r0 = &m.x.x, *r1 = &m.y.x, *r2 = &m.z.x;
                colv[col][0] = r0[col]; colv[col][1] = r1[col]; colv[col][2] = r2[col];
                len[col] = sqrt(colv[col][0] * colv[col][0] + colv[col][1] * colv[col][1] + colv[col][2] * colv[col][2]);
                sum += len[col]; sumsq += len[col] * len[col];
                if (len[col] > maxlen) maxlen = len[col];
            }
            bool orthogonal = true;                       // T*R*S transforms have orthogonal columns
            for (int a = 0; a < 3; ++a)
                for (int b = a + 1; b < 3; ++b) {
                    const double dp = colv[a][0] * colv[b][0] + colv[a][1] * colv[b][1] + colv[a][2] * colv[b][2];
                    if (fabs(dp) > 1e-4 * len[a] * len[b]) orthogonal = false;
                }
            double rad;
            if (orthogonal) rad = (g.type == PT_SPHERE) ? 0.5 * maxlen : 0.5 * sqrt(sumsq);   // r*s_max / half diagonal
            else rad = (g.type == PT_SPHERE) ? 0.5 * sqrt(sumsq) : 0.5 * sum;                  // Frobenius / triangle bound
            if (g.type == PT_SPHERE) rad *= sphere_noise_factor(g, reach[i]);                  // (what the reference's test hits)
            // (+ what the reference's inverseTransform * origin and the cull's own c - o lose far from the world origin)
            rad = rad * 1.02 + 1e-3 + 2e-6 * reach[i] + 4e-6 * (fabs((double)p.cx) + fabs((double)p.cy) + fabs((double)p.cz));
            p.bound_r2 = (float)(rad * rad);
            // self_r2: may a ray that leaves this primitive on its OUTSIDE skip it at its next bounce (resident paths)?  Mathematically a
            // ray that starts outside a convex primitive and moves away from it cannot meet it again; the kernels may only rely on
            // that where the REFERENCE'S ARITHMETIC agrees for every such ray (the oracle tests the primitive all the same):
            //  (i)  the shading normal must be the surface normal -- a uniformly scaled sphere (the reference's sphere normal is the
            //       direction from the centre, which an ellipsoid's is not: a ray sampled about it can dive back into an ellipsoid) or a
            //       cube under a transform with orthogonal columns;
            //  (ii) the new origin, hit point + 0.0002 * normal, must really lie outside.  A cube's slab distances are accurate to a few
            //       ulps of the ray's length (error along the face axis ~ 3e-7 * distance in WORLD units, whatever the scale): with the
            //       scene within 128 units that is <= 1e-4 of margin against the 2e-4 bias.  A SPHERE's quadratic is not: b*b - c cancels
            //       for a ray that comes from far away relative to the sphere's size (fp32 error ~ (distance / scale)^2 * 6e-8 object
            //       units), and the reported point can lie inside the sphere by more than the bias -- the path then bounces on inside
            //       (cloud256: a sphere of radius 0.11 met from 12.5 units away; the oracle renders exactly that).  So for spheres the
            //       kernels check the hit point itself: skipping needs |p - c| > r - 0.0002 + margin, i.e. an origin at least `margin`
            //       outside, margin = 2e-5 + 1e-5 * scale + rounding of the check (coordinates up to `coord`);
            //  (iii) moderate sizes: the object-space offset of the bias, 0.0002 / scale, must stay far above fp32 noise (scale <= 64).
            // The bounds-checking build runs the exact test on every skipped pair and reports a hit.
            {
                bool tight = true;                        // orthogonality to 1e-6 (the 1e-4 above only sizes a culling sphere)
                for (int a = 0; a < 3; ++a)
                    for (int b = a + 1; b < 3; ++b) {
                        const double dp = colv[a][0] * colv[b][0] + colv[a][1] * colv[b][1] + colv[a][2] * colv[b][2];
                        if (fabs(dp) > 1e-6 * len[a] * len[b]) tight = false;
                    }
                double minlen = len[0];
                for (int col = 1; col < 3; ++col) if (len[col] < minlen) minlen = len[col];
                const bool uniform = maxlen - minlen <= 1e-6 * maxlen;
                const double coord = sqrt((double)p.cx * p.cx + (double)p.cy * p.cy + (double)p.cz * p.cz) + rad;       // (largest coordinate of its surface)
                const bool sane = maxlen <= 64.0 && minlen > 0.0 && coord <= 128.0;      // (a NaN fails every compare)
                p.self_r2 = 0.0f;
                if (tight && sane && g.type == PT_CUBE) p.self_r2 = 1.0f;
                if (tight && sane && g.type == PT_SPHERE && uniform) {
                    const double r = 0.5 * maxlen, margin = 2e-5 + 1e-5 * maxlen + 8.0 * coord * 6e-8;
                    const double rc = r - 0.0002 + margin;
                    if (margin < 1.5e-4 && rc > 0.0) p.self_r2 = (float)(rc * rc * (1.0 + 1e-6));
                }
                if (getenv("PT_NO_SELF_SKIP")) p.self_r2 = 0.0f;      // (ablation switch)
            }
        }
        for (size_t t = 0; t < nT; ++t) {
            ptd::Prim &p = prims[nGeoms + t];
            const float *w = &triw[12 * t];
            p.type = 3u;
            p.material = (uint32_t)c->geoms[(size_t)c->tri_geom[t]].materialid;
            for (int q = 0; q < 9; ++q) p.inv[q] = w[q];
            p.fwd[0] = w[9]; p.fwd[1] = w[10]; p.fwd[2] = w[11];
            const double cx = w[0] + (w[3] + w[6]) / 3.0, cy = w[1] + (w[4] + w[7]) / 3.0, cz = w[2] + (w[5] + w[8]) / 3.0;
            double r2 = 0.0;
            for (int j = 0; j < 3; ++j) {
                const double px = w[0] + (j == 1 ? w[3] : (j == 2 ? w[6] : 0.0)) - cx, py = w[1] + (j == 1 ? w[4] : (j == 2 ? w[7] : 0.0)) - cy,
                             pz = w[2] + (j == 1 ? w[5] : (j == 2 ? w[8] : 0.0)) - cz;
                r2 = std::max(r2, px * px + py * py + pz * pz);
            }
            const double rad = sqrt(r2) * 1.02 + 1e-3 + 2e-6 * tri_reach(t) + 4e-6 * (fabs(cx) + fabs(cy) + fabs(cz));
            p.cx = (float)cx; p.cy = (float)cy; p.cz = (float)cz;
            p.bound_r2 = (float)(rad * rad);
        }
        // direct lighting: the light table -- emissive geoms in list order (at most 16 entries).  A sphere / cube is an
        // entry with its surface area from getRadiuses' half-extents (ref: src/intersections.h:120-129) in fp32 with the
        // sampler's own operation order; an emissive MESH geom is ONE entry made of its triangles of positive area
        // (area-weighted pick by the running fp32 sum of 0.5 |e1 x e2|, then a uniform point).  Prim::area > 0 marks the
        // primitives the table covers: only those are left to the explicit sampling when a path hits them by chance.
        std::vector<pt::LightRec> lights;
        std::vector<int> ltris;
        std::vector<float> lcdf;
        for (size_t i = 0; i < c->geoms.size() && lights.size() < 16; ++i) {
            const pt_static_geom &g = c->geoms[i];
            if (!(c->mats[(size_t)g.materialid].emittance > 0.0f)) continue;
            if (g.type == PT_MESH) {
                float acc = 0.0f;
                const size_t first = ltris.size();
                for (size_t t = 0; t < nT; ++t) {
                    if (c->tri_geom[t] != (int)i) continue;
                    const float *w = &triw[12 * t];
                    const v3 cr = cross3(v3{w[3], w[4], w[5]}, v3{w[6], w[7], w[8]});
                    const float a = 0.5f * sqrtf(cr.x * cr.x + cr.y * cr.y + cr.z * cr.z);
                    if (!(a > 0.0f)) continue;
                    acc = acc + a;
                    ltris.push_back((int)t);
                    lcdf.push_back(acc);
                    prims[nGeoms + t].area = a;
                }
                if (ltris.size() == first) continue;          // no surface: not a light
                lights.push_back(pt::LightRec{(int)i, (int)first, (int)(ltris.size() - first), acc});
                continue;
            }
            const pt_mat4 &m = g.transform;
            auto mv = [&](float x, float y, float z, float out[3]) {
                out[0] = (m.x.x * x) + (m.x.y * y) + (m.x.z * z) + (m.x.w * 1.0f);
                out[1] = (m.y.x * x) + (m.y.y * y) + (m.y.z * z) + (m.y.w * 1.0f);
                out[2] = (m.z.x * x) + (m.z.y * y) + (m.z.z * z) + (m.z.w * 1.0f);
            };
            float org[3], ax[3][3], r[3];
            mv(0, 0, 0, org); mv(.5f, 0, 0, ax[0]); mv(0, .5f, 0, ax[1]); mv(0, 0, .5f, ax[2]);
            for (int a = 0; a < 3; ++a) {
                const float dx = ax[a][0] - org[0], dy = ax[a][1] - org[1], dz = ax[a][2] - org[2];
                r[a] = sqrtf(dx * dx + dy * dy + dz * dz);
            }
            float area;
            if (g.type == PT_CUBE) {
                const float side1 = r[0] * r[1] * 4.0f, side2 = r[2] * r[1] * 4.0f, side3 = r[0] * r[2] * 4.0f;
                area = 2.0f * (side1 + side2 + side3);
            } else {
                area = 4.18879020478639098f * ((r[0] * r[1] + r[0] * r[2]) + r[1] * r[2]);
            }
            prims[i].area = area;
            lights.push_back(pt::LightRec{(int)i, 0, 0, area});
        }
        k.nlights = o.direct_light ? (int)lights.size() : 0;
        k.ngeoms = (int)nGeoms;
        k.absorption = o.absorption ? 1 : 0;
        k.scatter = o.scatter ? 1 : 0;
        if (c->d_lights) { (void)hipFree(c->d_lights); c->d_lights = nullptr; }
        if (c->d_light_tris) { (void)hipFree(c->d_light_tris); c->d_light_tris = nullptr; }
        if (c->d_light_cdf) { (void)hipFree(c->d_light_cdf); c->d_light_cdf = nullptr; }
        HIP_TRY(hipMalloc((void **)&c->d_lights, (lights.size() ? lights.size() : 1) * sizeof(pt::LightRec)));
        if (!lights.empty()) HIP_TRY(hipMemcpy(c->d_lights, lights.data(), lights.size() * sizeof(pt::LightRec), hipMemcpyHostToDevice));
        HIP_TRY(hipMalloc((void **)&c->d_light_tris, (ltris.size() ? ltris.size() : 1) * sizeof(int)));
        HIP_TRY(hipMalloc((void **)&c->d_light_cdf, (lcdf.size() ? lcdf.size() : 1) * sizeof(float)));
        if (!ltris.empty()) {
            HIP_TRY(hipMemcpy(c->d_light_tris, ltris.data(), ltris.size() * sizeof(int), hipMemcpyHostToDevice));
            HIP_TRY(hipMemcpy(c->d_light_cdf, lcdf.data(), lcdf.size() * sizeof(float), hipMemcpyHostToDevice));
        }
        k.lights = c->d_lights;
        k.light_tris = c->d_light_tris;
        k.light_cdf = c->d_light_cdf;

        if (c->d_prims) { (void)hipFree(c->d_prims); c->d_prims = nullptr; }
        HIP_TRY(hipMalloc((void **)&c->d_prims, prims.size() * sizeof(ptd::Prim)));
        HIP_TRY(hipMemcpy(c->d_prims, prims.data(), prims.size() * sizeof(ptd::Prim), hipMemcpyHostToDevice));

        // camera rays all start at the eye: multiplyMV(inverseTransform, (eye,1)) once per primitive, with the
        // reference's operation order (ref: src/intersections.h:53-59,85)
        std::vector<float> ro(prims.size() * 4, 0.0f);
        for (size_t i = 0; i < c->geoms.size(); ++i) {
            const pt_mat4 &m = c->geoms[i].inverseTransform;
            const float ex = c->cam.position.x, ey = c->cam.position.y, ez = c->cam.position.z;
            ro[4 * i + 0] = (m.x.x * ex) + (m.x.y * ey) + (m.x.z * ez) + (m.x.w * 1.0f);
            ro[4 * i + 1] = (m.y.x * ex) + (m.y.y * ey) + (m.y.z * ez) + (m.y.w * 1.0f);
            ro[4 * i + 2] = (m.z.x * ex) + (m.z.y * ey) + (m.z.z * ez) + (m.z.w * 1.0f);
        }
        for (size_t t = 0; t < nT; ++t) {
            ro[4 * (nGeoms + t) + 0] = c->cam.position.x; ro[4 * (nGeoms + t) + 1] = c->cam.position.y; ro[4 * (nGeoms + t) + 2] = c->cam.position.z;
        }
        if (c->d_ro_eye) { (void)hipFree(c->d_ro_eye); c->d_ro_eye = nullptr; }
        HIP_TRY(hipMalloc((void **)&c->d_ro_eye, ro.size() * sizeof(float)));
        HIP_TRY(hipMemcpy(c->d_ro_eye, ro.data(), ro.size() * sizeof(float), hipMemcpyHostToDevice));

        // The world normal of a box face depends on (primitive, face) only: normalize(multiplyMV(transform, (+-axis, 0)))
        // evaluated here once with the kernels' operation order (boxNormal in pt_device.h), looked up per hit.
        std::vector<float> fnorm(prims.size() * 32, 0.0f);
        for (size_t i = 0; i < c->geoms.size(); ++i) {
            const pt_mat4 &m = c->geoms[i].transform;
            for (int face = 0; face < 8; ++face) {
                const int axis = face & 3;
                if (axis == 3) continue;
                const float sgn = (face & 4) ? -1.0f : 1.0f;
                const float nx = axis == 0 ? sgn : 0.0f, ny = axis == 1 ? sgn : 0.0f, nz = axis == 2 ? sgn : 0.0f;
                const v3 w = {(m.x.x * nx) + (m.x.y * ny) + (m.x.z * nz) + (m.x.w * 0.0f),
                              (m.y.x * nx) + (m.y.y * ny) + (m.y.z * nz) + (m.y.w * 0.0f),
                              (m.z.x * nx) + (m.z.y * ny) + (m.z.z * nz) + (m.z.w * 0.0f)};
                const v3 n = normalize3(w);
                fnorm[i * 32 + (size_t)face * 4 + 0] = n.x; fnorm[i * 32 + (size_t)face * 4 + 1] = n.y; fnorm[i * 32 + (size_t)face * 4 + 2] = n.z;
            }
        }
        // cube lights: the sampler's face-choice thresholds (ref: src/intersections.h:140-170, same fp32 operations),
        // parked in the table's unused entries 3 (t1..t4) and 7 (t5)
        for (size_t i = 0; i < c->geoms.size(); ++i) {
            const pt_static_geom &g = c->geoms[i];
            if (g.type != PT_CUBE) continue;
            const pt_mat4 &m = g.transform;
            auto mv = [&](float x, float y, float z, float out[3]) {
                out[0] = (m.x.x * x) + (m.x.y * y) + (m.x.z * z) + (m.x.w * 1.0f);
                out[1] = (m.y.x * x) + (m.y.y * y) + (m.y.z * z) + (m.y.w * 1.0f);
                out[2] = (m.z.x * x) + (m.z.y * y) + (m.z.z * z) + (m.z.w * 1.0f);
            };
            float org[3], ax[3][3], r[3];
            mv(0, 0, 0, org); mv(.5f, 0, 0, ax[0]); mv(0, .5f, 0, ax[1]); mv(0, 0, .5f, ax[2]);
            for (int a2 = 0; a2 < 3; ++a2) {
                const float dx = ax[a2][0] - org[0], dy = ax[a2][1] - org[1], dz = ax[a2][2] - org[2];
                r[a2] = sqrtf(dx * dx + dy * dy + dz * dz);
            }
            const float side1 = r[0] * r[1] * 4.0f, side2 = r[2] * r[1] * 4.0f, side3 = r[0] * r[2] * 4.0f;
            const float totalarea = 2.0f * (side1 + side2 + side3);
            fnorm[i * 32 + 12 + 0] = side1 / totalarea;
            fnorm[i * 32 + 12 + 1] = (side1 * 2) / totalarea;
            fnorm[i * 32 + 12 + 2] = ((side1 * 2) + (side2)) / totalarea;
            fnorm[i * 32 + 12 + 3] = ((side1 * 2) + (side2 * 2)) / totalarea;
            fnorm[i * 32 + 28 + 0] = ((side1 * 2) + (side2 * 2) + (side3)) / totalarea;
        }
        if (c->d_face_n) { (void)hipFree(c->d_face_n); c->d_face_n = nullptr; }
        HIP_TRY(hipMalloc((void **)&c->d_face_n, fnorm.size() * sizeof(float)));
        HIP_TRY(hipMemcpy(c->d_face_n, fnorm.data(), fnorm.size() * sizeof(float), hipMemcpyHostToDevice));

        // ... and a wave of camera rays covers a small solid angle: the padded world box of every primitive,
        // relative to the eye, lets a wave skip the primitives none of its rays can reach (culling only)
        std::vector<float> be(prims.size() * 8, 0.0f), bw(prims.size() * 8, 0.0f), slab_n(prims.size() * 4, 0.0f);
        uint32_t slab_mask = 0u;
        for (size_t i = 0; i < nP; ++i) {
            // hit-or-miss culling only (no distance pruning): the padding just has to cover fp32 rounding (1e-6 of the
            // coordinates), and a tight one keeps a ray that leaves a wall (0.0002 above it) outside that wall's box
            const Aabb b = i < nGeoms ? prim_bounds(c->geoms[i], 1.005, 1e-4, reach[i]) : tri_bounds(i - nGeoms, 1.005, 1e-4);
            for (int a = 0; a < 3; ++a) { bw[8 * i + (size_t)a] = b.lo[a]; bw[8 * i + 4 + (size_t)a] = b.hi[a]; }
            // a TILTED cube fills little of its world box (the bundled scene's walls, with ROTAT read as radians: 10 x 0.01 x 10 slabs at
            // arbitrary angles, whose boxes fill the room): the pre-test also clips the ray against the slab between the two faces of the
            // cube's THINNEST axis, planes n.x = d_lo / d_hi (n in the record's LDS pad, d_lo beside it, d_hi in the far entry's w).
            // Padded like the box (0.5 % + 1e-4 + what the reference's slab arithmetic errs by, 2e-6 x reach, + the rounding of n.o).
            if (i < nGeoms && i < 32 && c->geoms[i].type == PT_CUBE && !getenv("PT_NO_SLAB")) {
                const pt_mat4 &m = c->geoms[i].transform;
                const float *r0 = &m.x.x, *r1 = &m.y.x, *r2 = &m.z.x;
                double col[3][3], len[3];
                for (int j = 0; j < 3; ++j) {
                    col[j][0] = r0[j]; col[j][1] = r1[j]; col[j][2] = r2[j];
                    len[j] = sqrt(col[j][0] * col[j][0] + col[j][1] * col[j][1] + col[j][2] * col[j][2]);
                }
                bool orth = len[0] > 0.0 && len[1] > 0.0 && len[2] > 0.0;
                for (int a = 0; a < 3 && orth; ++a)
                    for (int b2 = a + 1; b2 < 3; ++b2)
                        if (!(fabs(col[a][0] * col[b2][0] + col[a][1] * col[b2][1] + col[a][2] * col[b2][2]) <= 1e-5 * len[a] * len[b2])) orth = false;
                const double vbox = ((double)b.hi[0] - b.lo[0]) * ((double)b.hi[1] - b.lo[1]) * ((double)b.hi[2] - b.lo[2]);
                const int t = len[0] <= len[1] ? (len[0] <= len[2] ? 0 : 2) : (len[1] <= len[2] ? 1 : 2);
                // (worth its dozen and a half instructions per ray and trip only where the slab is a small part of the box: 0.25 below)
                const double hx = 0.5 * ((double)b.hi[0] - b.lo[0]), hy = 0.5 * ((double)b.hi[1] - b.lo[1]), hz = 0.5 * ((double)b.hi[2] - b.lo[2]);
                if (orth && std::isfinite(vbox) && vbox > 0.0) {
                    const double n[3] = {col[t][0] / len[t], col[t][1] / len[t], col[t][2] / len[t]};
                    const double cc[3] = {m.x.w, m.y.w, m.z.w};
                    const double nc = n[0] * cc[0] + n[1] * cc[1] + n[2] * cc[2];
                    const double coord = fabs(cc[0]) + fabs(cc[1]) + fabs(cc[2]) + hx + hy + hz;      // |n.o| of any origin that matters is below this + reach
                    const double half = 0.5 * len[t] * 1.005 + 1e-4 + 2e-6 * reach[i] + 4e-7 * (coord + reach[i]);
                    // the box's extent along n: the slab only helps where it is much thinner than that
                    const double ext = fabs(n[0]) * hx + fabs(n[1]) * hy + fabs(n[2]) * hz;
                    if (half < 0.25 * ext) {
                        slab_n[4 * i + 0] = (float)n[0]; slab_n[4 * i + 1] = (float)n[1]; slab_n[4 * i + 2] = (float)n[2];
                        // (n rounded to fp32 tilts the planes by <= 1e-7 rad: <= 1e-7 x the cube's size across its faces, inside the pad)
                        slab_n[4 * i + 3] = (float)(nc - half);
                        bw[8 * i + 7] = (float)(nc + half);
                        slab_mask |= 1u << i;
                    }
                }
            }
            const double e[3] = {c->cam.position.x, c->cam.position.y, c->cam.position.z};
            for (int a = 0; a < 3; ++a) {
                const double lo = (double)b.lo[a] - e[a], hi = (double)b.hi[a] - e[a];
                const double pad = 1e-6 * (fabs(lo) + fabs(hi) + fabs(e[a])) + 1e-6;      // fp32 rounding of the subtraction
                be[8 * i + (size_t)a] = (float)(lo - pad);
                be[8 * i + 4 + (size_t)a] = (float)(hi + pad);
            }
            be[8 * i + 7] = bw[8 * i + 7];                  // (the slab's planes stay in world space: the kernel forms n.o for camera rays too)
        }
        k.slab_mask = slab_mask;
        if (c->d_slab_n) { (void)hipFree(c->d_slab_n); c->d_slab_n = nullptr; }
        HIP_TRY(hipMalloc((void **)&c->d_slab_n, slab_n.size() * sizeof(float)));
        HIP_TRY(hipMemcpy(c->d_slab_n, slab_n.data(), slab_n.size() * sizeof(float), hipMemcpyHostToDevice));
        k.slab_n = c->d_slab_n;
        c->h_box_eye = be;
        if (c->d_box_eye) { (void)hipFree(c->d_box_eye); c->d_box_eye = nullptr; }
        HIP_TRY(hipMalloc((void **)&c->d_box_eye, be.size() * sizeof(float)));
        HIP_TRY(hipMemcpy(c->d_box_eye, be.data(), be.size() * sizeof(float), hipMemcpyHostToDevice));
        if (c->d_box_world) { (void)hipFree(c->d_box_world); c->d_box_world = nullptr; }
        HIP_TRY(hipMalloc((void **)&c->d_box_world, bw.size() * sizeof(float)));
        HIP_TRY(hipMemcpy(c->d_box_world, bw.data(), bw.size() * sizeof(float), hipMemcpyHostToDevice));

        const size_t nM = c->mats.size();
        std::vector<float> planes((nM ? nM : 1) * ptd::M_PLANES, 0.0f);
        for (size_t i = 0; i < nM; ++i) {
            const pt_material &m = c->mats[i];
            planes[ptd::M_CR * nM + i] = m.color.x; planes[ptd::M_CG * nM + i] = m.color.y; planes[ptd::M_CB * nM + i] = m.color.z;
            planes[ptd::M_SR * nM + i] = m.specularColor.x; planes[ptd::M_SG * nM + i] = m.specularColor.y;
            planes[ptd::M_SB * nM + i] = m.specularColor.z;
            planes[ptd::M_REFL * nM + i] = m.hasReflective; planes[ptd::M_REFR * nM + i] = m.hasRefractive;
            planes[ptd::M_IOR * nM + i] = m.indexOfRefraction; planes[ptd::M_EMIT * nM + i] = m.emittance;
            planes[ptd::M_AR * nM + i] = m.absorptionCoefficient.x; planes[ptd::M_AG * nM + i] = m.absorptionCoefficient.y;
            planes[ptd::M_AB * nM + i] = m.absorptionCoefficient.z;
            planes[ptd::M_SCAT * nM + i] = m.hasScatter; planes[ptd::M_RSCT * nM + i] = m.reducedScatterCoefficient;
        }
        if (c->d_mats) { (void)hipFree(c->d_mats); c->d_mats = nullptr; }
        HIP_TRY(hipMalloc((void **)&c->d_mats, planes.size() * sizeof(float)));
        HIP_TRY(hipMemcpy(c->d_mats, planes.data(), planes.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    // motion blur with a shutter time per ray: knot states at shutter times j / slices, built as the slices are (TRS
    // interpolated component-wise, matrices rebuilt by the loader's buildTransformationMatrix); the kernels interpolate the
    // transform's rows and the camera vectors entry-wise between the two knots around a ray's time and invert the result
    k.nknots = 0;
    if (motion_per_ray(c)) {
        if (nT > 0) return fail(PT_ERR_INVALID, "motion_per_ray: triangle meshes need the slice scheme (motion_per_ray = 0)");
        for (const pt_static_geom &g : c->geoms)
            if (g.type == PT_MESH) return fail(PT_ERR_INVALID, "motion_per_ray: MESH objects need the slice scheme (motion_per_ray = 0)");
        if (o.compaction != 1) return fail(PT_ERR_INVALID, "motion_per_ray needs compaction 1 (got %d)", o.compaction);
        if (!(o.geom_path == 0 || o.geom_path == 1 || o.geom_path == 5))
            return fail(PT_ERR_INVALID, "motion_per_ray runs on the scalar and the pair-queue geometry paths (geom_path 0, 1 or 5, got %d)", o.geom_path);
        if (!(o.workgroup == 0 || o.workgroup == 256)) return fail(PT_ERR_INVALID, "motion_per_ray needs workgroup 0 or 256 (got %d)", o.workgroup);
        const int nk = c->motion_slices + 1;
        std::vector<float> kn((size_t)nk * nGeoms * 12, 0.0f), kc((size_t)nk * 12, 0.0f);
        // the primitives' padded world boxes SWEPT over the shutter interval (pair path pre-test): a point of the primitive at a
        // time between two knots is a convex combination of its images at the knots (the rows are interpolated entry-wise),
        // so the box around the knots' boxes holds it
        std::vector<float> swept(nGeoms * 8, 0.0f);
        for (int j = 0; j < nk; ++j) {
            const float t = (float)j / (float)(nk - 1);
            for (size_t i = 0; i < nGeoms; ++i) {
                const pt_static_geom &a = c->geoms[i], &b = c->geoms_next[i];
                pt_mat4 inv;
                const pt_mat4 fwd = ptamd::buildTransformationMatrix(lerp3(a.translation, b.translation, t), lerp3(a.rotation, b.rotation, t),
                                                                     lerp3(a.scale, b.scale, t), c->motion_rotat, &inv);
                memcpy(&kn[((size_t)j * nGeoms + i) * 12], &fwd, 12 * sizeof(float));
                pt_static_geom gk = a;
                gk.transform = fwd;
                const Aabb bk = prim_bounds(gk, 1.005, 1e-4, reach[i]);
                for (int ax = 0; ax < 3; ++ax) {
                    if (j == 0 || bk.lo[ax] < swept[8 * i + (size_t)ax]) swept[8 * i + (size_t)ax] = bk.lo[ax];
                    if (j == 0 || bk.hi[ax] > swept[8 * i + 4 + (size_t)ax]) swept[8 * i + 4 + (size_t)ax] = bk.hi[ax];
                }
            }
            const pt_camera_data &ca = c->cam;
            const pt_camera_data &cb = c->have_cam_next ? c->cam_next : c->cam;
            const pt_vec3 pos = c->have_cam_next ? lerp3(ca.position, cb.position, t) : ca.position;
            const pt_vec3 view = c->have_cam_next ? lerp3(ca.view, cb.view, t) : ca.view;
            const pt_vec3 up = c->have_cam_next ? lerp3(ca.up, cb.up, t) : ca.up;
            float *cd = &kc[(size_t)j * 12];
            cd[0] = pos.x; cd[1] = pos.y; cd[2] = pos.z;
            cd[4] = view.x; cd[5] = view.y; cd[6] = view.z;
            cd[8] = up.x; cd[9] = up.y; cd[10] = up.z;
        }
        if (c->d_knots) { (void)hipFree(c->d_knots); c->d_knots = nullptr; }
        if (c->d_knot_cam) { (void)hipFree(c->d_knot_cam); c->d_knot_cam = nullptr; }
        HIP_TRY(hipMalloc((void **)&c->d_knots, (kn.size() ? kn.size() : 1) * sizeof(float)));
        HIP_TRY(hipMalloc((void **)&c->d_knot_cam, kc.size() * sizeof(float)));
        if (!kn.empty()) HIP_TRY(hipMemcpy(c->d_knots, kn.data(), kn.size() * sizeof(float), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(c->d_knot_cam, kc.data(), kc.size() * sizeof(float), hipMemcpyHostToDevice));
        if (!swept.empty()) HIP_TRY(hipMemcpy(c->d_box_world, swept.data(), swept.size() * sizeof(float), hipMemcpyHostToDevice));
        k.slab_mask = 0u;                                    // (a moving cube's slab moves with it)
        k.knots = c->d_knots;
        k.knot_cam = c->d_knot_cam;
        k.nknots = nk;
        const double PI = 3.1415926535897932384626422832795028841971;
        k.tan_x = (float)tan((double)c->cam.fov.x * (PI / 180.0));
        k.tan_y = (float)tan((double)c->cam.fov.y * (PI / 180.0));
    }
    k.prims = c->d_prims;
    k.ro_eye = c->d_ro_eye;
    k.face_n = c->d_face_n;

    k.box_eye = c->d_box_eye;
    k.eye_cull = getenv("PT_NO_EYE_CULL") ? 0 : 1;
    k.box_world = c->d_box_world;

    // culling hierarchy (used by geom_path 4 / large scenes); MESH primitives have no geometry and stay out
    {
        std::vector<Aabb> boxes(nP);
        std::vector<int> ptype(nP, 3);                           // primitive type by expanded index (3 = triangle)
        for (size_t i = 0; i < nGeoms; ++i) ptype[i] = c->geoms[i].type;
        std::vector<int> idx;
        float slo[3] = {0, 0, 0}, shi[3] = {0, 0, 0};
        bool first_box = true;
        for (size_t i = 0; i < nP; ++i) {
            boxes[i] = i < nGeoms ? prim_bounds(c->geoms[i], 1.005, 1e-4, reach[i]) : tri_bounds(i - nGeoms, 1.005, 1e-4);    // hit-or-miss culling: the distance pruning has its own slack
            if (ptype[i] == PT_MESH) continue;
            for (int a = 0; a < 3; ++a) {
                if (first_box || boxes[i].lo[a] < slo[a]) slo[a] = boxes[i].lo[a];
                if (first_box || boxes[i].hi[a] > shi[a]) shi[a] = boxes[i].hi[a];
            }
            first_box = false;
        }
        float sext = 0.0f;
        for (int a = 0; a < 3; ++a) if (shi[a] - slo[a] > sext) sext = shi[a] - slo[a];
        k.nbig = 0;
        for (size_t i = 0; i < nP; ++i) {
            if (ptype[i] == PT_MESH) continue;
            float ext = 0.0f;
            for (int a = 0; a < 3; ++a) if (boxes[i].hi[a] - boxes[i].lo[a] > ext) ext = boxes[i].hi[a] - boxes[i].lo[a];
            // a primitive spanning > 40 % of the scene goes to the always-tested list (at most 16 of them)
            if (ext > 0.4f * sext && k.nbig < 16 && nP > 16) k.big[k.nbig++] = (int)i;
            else idx.push_back((int)i);
        }
        std::vector<ptd::BvhNode> nodes;
        if (!idx.empty()) build_bvh(boxes, idx, 0, idx.size(), nodes);
        for (auto &nd : nodes)
            if (nd.prim >= 0) nd.prim |= (ptype[(size_t)nd.prim] == PT_CUBE ? 1 : 0) << 30;
        k.nnodes = (int)nodes.size();
        c->h_bvh = nodes;
        c->h_boxes.assign(nP * 6, 0.0f);
        for (size_t i = 0; i < nP; ++i) for (int a = 0; a < 3; ++a) { c->h_boxes[6 * i + (size_t)a] = boxes[i].lo[a]; c->h_boxes[6 * i + 3 + (size_t)a] = boxes[i].hi[a]; }
        {
            std::vector<float> wide;
            int wdepth = 0;
            if (!nodes.empty()) build_wide4(nodes, ptype, 0, 1, wide, wdepth);
            k.nnodes4 = (int)(wide.size() / ptd::W4_FLOATS);
            k.wdepth = wdepth;
            if (c->d_bvh4) { (void)hipFree(c->d_bvh4); c->d_bvh4 = nullptr; }
            if (wide.empty()) wide.resize(ptd::W4_FLOATS, 0.0f);
            HIP_TRY(hipMalloc((void **)&c->d_bvh4, wide.size() * sizeof(float)));
            HIP_TRY(hipMemcpy(c->d_bvh4, wide.data(), wide.size() * sizeof(float), hipMemcpyHostToDevice));
            k.bvh4 = c->d_bvh4;
        }
        if (c->d_bvh) { (void)hipFree(c->d_bvh); c->d_bvh = nullptr; }
        if (nodes.empty()) nodes.push_back(ptd::BvhNode());
        HIP_TRY(hipMalloc((void **)&c->d_bvh, nodes.size() * sizeof(ptd::BvhNode)));
        HIP_TRY(hipMemcpy(c->d_bvh, nodes.data(), nodes.size() * sizeof(ptd::BvhNode), hipMemcpyHostToDevice));
        k.bvh = c->d_bvh;
    }
    k.mats = c->d_mats;

    // framebuffer
    const size_t img_bytes = (size_t)npix * 3 * sizeof(float);
    if (img_bytes != c->image_bytes) c->image_valid = false;
    c->image_bytes = img_bytes;
    if (!c->d_image_bound && c->image_cap < img_bytes) {
        if (c->d_image_own) (void)hipFree(c->d_image_own);
        c->d_image_own = nullptr; c->image_cap = 0;
        HIP_TRY(hipMalloc((void **)&c->d_image_own, img_bytes));
        c->image_cap = img_bytes;
        HIP_TRY(hipMemset(c->d_image_own, 0, img_bytes));
        c->image_valid = false;
    }
    k.image = image_ptr(c);

    k.st = c->d_state;

    // launch shape: persistent workgroups, as many as are resident at once
    pt::LaunchCfg &cfg = c->cfg;
    cfg.workgroup = o.workgroup ? o.workgroup : 256;
    // library choice: the pair queue's pre-test is linear in the primitive count, the hierarchy walks logarithmic:
    // measured crossover near 40 primitives (profiles/r01/crossover_pair_vs_walk.txt); above it the batched 4-wide walk
    cfg.geom = o.geom_path == 0 ? (k.nG <= 40 ? 4 : 6) : o.geom_path - 1;
    if (k.ntri > 0) {
        // triangle records are understood by the scalar loop and the batched walks only
        if (o.geom_path == 0) cfg.geom = 6;
        else if (!(cfg.geom == 0 || cfg.geom == 6 || cfg.geom == 7))
            return fail(PT_ERR_INVALID, "scenes with triangle meshes need geom_path 0, 1, 7 or 8 (got %d)", o.geom_path);
    }
    // the batched walks and the pair queues keep 20-bit primitive / node indices beside their flag bits
    if ((cfg.geom == 4 || cfg.geom == 5 || cfg.geom == 6 || cfg.geom == 7) && (k.nG >= (1 << 20) || k.nnodes4 >= (1 << 20))) {
        if (o.geom_path != 0)
            return fail(PT_ERR_INVALID, "%d primitives (triangles included), %d wide nodes: at most %d on this geometry path", k.nG, k.nnodes4, (1 << 20) - 1);
        cfg.geom = 0;
    }
    // the batched walk's entry ring must hold a full batch above its depth-first reserve (pt_bounce.h W4_STACK = 512): a
    // degenerate hierarchy (e.g. sizes in a long geometric progression) can be deeper than that allows
    if ((cfg.geom == 6 || cfg.geom == 7) && 64 + 3 * k.wdepth + 4 > 512) {
        if (o.geom_path != 0)
            return fail(PT_ERR_INVALID, "hierarchy depth %d: the batched walk's entry ring holds at most depth %d", k.wdepth, (512 - 64 - 4) / 3);
        cfg.geom = 0;
    }
    cfg.compact = o.compaction;
    cfg.nee = k.nlights > 0 ? 1 : 0;             // no lights: nothing to sample, the plain kernels are exact
    // the scattering kernels only when some material that can hold a medium asks for it: else the plain kernels are exact
    cfg.motion = k.nknots > 0 ? 1 : 0;
    // (per-ray shutter time: the pair queue up to 40 primitives -- pre-test against the swept boxes, exact tests in full batches with
    // per-pair interpolated rows --, the scalar loop above that or on request)
    if (cfg.motion) { cfg.geom = (o.geom_path == 5 || (o.geom_path == 0 && k.nG <= 40)) ? 4 : 0; cfg.workgroup = 256; }
    cfg.media = 0;
    if (k.scatter)
        for (const pt_material &m : c->mats)
            if (m.hasScatter > 0.0f && !(m.hasReflective > 0.0f)) cfg.media = 1;
    if (cfg.media && !(cfg.workgroup == 256 || cfg.workgroup == 512))
        return fail(PT_ERR_INVALID, "scatter needs workgroup 0, 256 or 512 (got %d)", cfg.workgroup);
    k.cull = (k.nG > 32 && !getenv("PT_NO_CULL")) ? 1 : 0;
    k.nshard = (cfg.compact == 1) ? pt::NSHARD : 1;
    // the pre-test with slabs (pair path): kernel instances of their own, taken only where a cube of the scene has a slab (decided again at
    // the end, should a fallback below change the launch shape)
    auto slab_wanted = [&]() {
        return (k.slab_mask != 0u && cfg.geom == 4 && cfg.compact == 1 && (cfg.workgroup == 256 || cfg.workgroup == 512) && !cfg.nee && !cfg.media &&
                !cfg.motion) ? 1 : 0;
    };
    cfg.slab = slab_wanted();
    // resident paths (pt_options.resident): the camera launch as ever, then ONE launch that traces bounces 1 .. depth - 1 with the
    // paths kept in registers -- where a kernel instance exists (pair queue / batched walks, plain kernels) and there is more
    // than one later bounce to fuse; everything else keeps the launch per bounce
    {
        int want = o.resident;
        if (want == 0 && getenv("PT_RESIDENT")) want = atoi(getenv("PT_RESIDENT")) > 0 ? 1 : -1;      // (only when the option leaves the choice open)
        // Library choice: from 5 bounces on.  The one launch saves depth - 2 launches and their tails and pays with a drain (the last
        // paths of every wave run up to depth - 1 trips alone): 1920x1080: - 5 % / +- 0 / + 4 % / + 4 % at depth 3 / 4 / 5 / 6;
        // 400x400: - 13 % / - 4 % / + 1 % / + 3 % (profiles/r04/ab_resident_paths.txt).  An explicit 1 takes it from 3 bounces on.
        // With direct lighting or scattering the instances exist and are bit-exact, but do not pay (config 2 with direct lighting - 3.4 %,
        // configs 3 / 5 +- 0 / + 1 %: the shadow-ray pass doubles a trip's work and halves the drain's weight): on request only.
        const int min_depth = (want == 0) ? 5 : 3;
        if (want == 0) want = (cfg.nee || cfg.media) ? -1 : PT_RESIDENT_DEFAULT;
        cfg.resident = 0;
        if (want > 0 && k.depth >= min_depth) {
            pt::LaunchCfg t = cfg;
            if ((t.geom == 6 || t.geom == 7) && o.workgroup == 0) t.workgroup = 512;     // (the walks pick 256 or 512 below: both exist)
            cfg.resident = pt::bounce_resident_available(t) ? 1 : 0;
        }
        k.refill_min = getenv("PT_REFILL_MIN") ? atoi(getenv("PT_REFILL_MIN")) : PT_REFILL_MIN_DEFAULT;
        if (k.refill_min < 1) k.refill_min = 1;
        if (k.refill_min > 64) k.refill_min = 64;
    }
    size_t lds = pt::bounce_lds_bytes(k, cfg);
    if ((cfg.geom == 6 || cfg.geom == 7) && o.workgroup == 0) {
        // the batched walk keeps 4.5 KiB of LDS per wave beside the node copy: take the workgroup size that puts most
        // waves on a CU (a larger workgroup shares one node copy among more waves)
        int best_wg = cfg.workgroup;
        long long best_waves = 0;
        for (int wg : {256, 512}) {
            pt::LaunchCfg t = cfg;
            t.workgroup = wg;
            const long long fit = (160LL * 1024) / ((long long)pt::bounce_lds_bytes(k, t) + 1300);
            const long long waves = std::min<long long>(fit * (wg / 64), 24);
            if (waves > best_waves) { best_waves = waves; best_wg = wg; }
        }
        cfg.workgroup = best_wg;
        lds = pt::bounce_lds_bytes(k, cfg);
    }
    if (cfg.geom == 6 && lds > 96 * 1024 && o.geom_path == 0) {
        cfg.geom = 7;                              // node copy too large for the LDS: the same walk, nodes through L1/L2
        cfg.workgroup = o.workgroup ? o.workgroup : 256;
        lds = pt::bounce_lds_bytes(k, cfg);
    }
    if (lds > 160 * 1024 && o.geom_path == 0) {
        // the hierarchy (32 B per node, 2 nodes per primitive) no longer fits the CU's LDS: fall back to the scalar
        // loop with the per-wave bounding-sphere cull, which needs none
        cfg.geom = 0;
        lds = pt::bounce_lds_bytes(k, cfg);
    }
    if (lds > 160 * 1024) return fail(PT_ERR_INVALID, "scene needs %zu B of LDS per workgroup (> 160 KiB)", lds);
    int per_cu = pt::bounce_max_blocks_per_cu(k, cfg);
    if (per_cu < 1) return fail(PT_ERR_HIP, "occupancy query failed for workgroup=%d (%s)", cfg.workgroup,
                                hipGetErrorString(hipGetLastError()));
    // The occupancy query over-counts near the LDS limit: 5 workgroups of 31.2 KB are resident together, 5 of 31.8 KB are
    // not (measured), i.e. about 1.3 KB per workgroup is spoken for.  A persistent grid sized for workgroups that are not
    // resident runs them as a second round (-7...-23 %), so the count is corrected here.
    {
        const long long fit = (160LL * 1024) / ((long long)lds + 1300);
        if (fit >= 1 && per_cu > (int)fit) per_cu = (int)fit;
    }
    // Persistent grid: 6 workgroups of 256 threads per CU is the measured optimum (config 2: 4 -> 32.0, 5 -> 34.0,
    // 6 -> 34.5, 7 -> 30.5 G ray-bounces/s; the same shape on the glass scene and at 4K): a seventh costs 11 %.
    int cap = 6 * 256 / cfg.workgroup;
    if (cap < 1) cap = 1;
    if (getenv("PT_MAX_WG_PER_CU") && atoi(getenv("PT_MAX_WG_PER_CU")) > 0) cap = atoi(getenv("PT_MAX_WG_PER_CU"));
    if (per_cu > cap) per_cu = cap;
    // iterations in flight per launch sequence: every bounce launch then carries batch x npix paths, which
    // amortises the per-launch fixed cost (launch, LDS staging, ramp, tail) over `batch` samples per pixel
    int batch = o.batch == 0 ? pt::PT_MAX_BATCH : o.batch;
    while (batch > 1 && (long long)npix * batch > (1LL << 28)) --batch;
    if (npix >= (1 << 27)) return fail(PT_ERR_INVALID, "tile of %d pixels too large (pixel index must fit 27 bits)", npix);
    c->batch = batch;
    k.nslot = batch;
    const long long nrays = (long long)npix * batch;
    const long long want = (nrays + cfg.workgroup - 1) / cfg.workgroup;
    long long grid = (long long)c->cu_count * per_cu;
    if (grid > want) grid = want;
    if (grid < 1) grid = 1;
    cfg.grid = (int)grid;
    if (cfg.resident && !pt::bounce_resident_available(cfg)) cfg.resident = 0;      // (a fallback above changed the launch shape)
    cfg.slab = slab_wanted();
    if (!cfg.slab) k.slab_mask = 0u;
    if (getenv("PT_DEBUG_CLOCK"))
        fprintf(stderr, "[ptamd] launch: geom %d, workgroup %d, %zu B LDS, %d workgroups/CU, grid %d, batch %d, resident paths %d (refill at %d free lanes)\n", cfg.geom, cfg.workgroup,
                lds, per_cu, cfg.grid, batch, cfg.resident, k.refill_min);

    // Launch sequences in flight: sequence q renders batches q, q + nseq, ... of a pt_render call on a stream of its own,
    // with its own ray pools, radiance planes and IterState; only the accumulates are ordered across the sequences
    // (iteration order).  The workgroups of a bounce launch do not finish together -- their lifetimes spread over
    // 65..100 % of the launch -- and a second sequence's launches fill the compute units the first one's tail leaves idle.
    // Library choice: two (config 2 on one box: 37.2 / 43.0 / 42.9 / 41.4 G ray-bounces/s with 1 / 2 / 3 / 4 sequences;
    // config 5: 17.7 / 20.5 / 19.7 / 18.8 G -- profiles/r03/sweep_sequences.txt).
    int nseq = o.sequences == 0 ? 2 : o.sequences;
    // (the environment only where the option leaves the choice to the library; pt_get_launch_info reports what was taken)
    if (o.sequences == 0 && getenv("PT_SEQUENCES") && atoi(getenv("PT_SEQUENCES")) >= 1 && atoi(getenv("PT_SEQUENCES")) <= PT_MAX_SEQUENCES)
        nseq = atoi(getenv("PT_SEQUENCES"));
    c->nseq = nseq;
    // per-iteration radiance planes (one write per path, folded into the image by k_accumulate), per sequence
    // (16-byte entries: r, g, b and the serial number of the batch that wrote them; zeroed once -- no batch has serial 0)
    const size_t lbuf_bytes = ((size_t)nrays * 4 * sizeof(float) + 255) & ~(size_t)255;

    // Camera rays: which primitives can the 64 pixels of a bounce-0 chunk see at all?  A primitive is dropped for a span of
    // 64 tile-local pixels when its padded box lies wholly outside one of the four side planes of the span's pixel frustum
    // (grown by a pixel on every side; a span that runs over a row end takes the whole rows) or wholly behind the eye.
    // Conservative: that only ever drops (ray, primitive) pairs whose box pre-test would fail in every lane -- the
    // bounds-checking build runs those tests anyway and reports any that passes.  Needs rays that start at the eye (no
    // lens); the lists also need chunks that are spans (npix % 64 == 0), the table copes with chunks across spans.
    //   pair path (<= 32 primitives): one word per span, bit g = primitive g stays (KParams::span_mask);
    //   batched walks: per span the list of the primitives that stay -- found by walking the hierarchy with the
    //   frustum -- which, when it is short, replaces the walk for the camera rays (KParams::span_off / span_list).
    k.span_mask = nullptr;
    k.span_off = nullptr;
    k.span_list = nullptr;
    const bool spans_ok = k.eye_cull && !(k.lens_radius > 0.0f) && npix / 64 <= (1 << 21);    // (host work: <= 2 M spans, a 134 Mpx tile)
    const bool want_mask = spans_ok && !cfg.motion && cfg.geom == 4 && k.nG <= 32 && k.ntri == 0 && c->h_box_eye.size() >= (size_t)k.nG * 8;
    const bool want_lists = spans_ok && npix % 64 == 0 && (cfg.geom == 6 || cfg.geom == 7) && c->h_boxes.size() >= (size_t)k.nG * 6 && k.nG <= 65536;
    if (want_mask || want_lists) {
        const int nspan = (npix + 63) / 64;                  // (the last one may be short; the lists need npix % 64 == 0)
        const double ex = k.eye[0], ey = k.eye[1], ez = k.eye[2];
        const double eye3[3] = {ex, ey, ez};
        const double vw[3] = {(double)k.M[0] - ex, (double)k.M[1] - ey, (double)k.M[2] - ez};
        auto gpix = [&](uint32_t pl) -> uint32_t {
            if (k.strip_span == 0u) return pl + k.pix_offset;
            const uint32_t j = pl / k.strip_span;
            return pl + k.strip_span * (j * (k.strip_world - 1u) + k.strip_rank);
        };
        auto dirOf = [&](double sx, double sy, double *o3) {
            for (int a = 0; a < 3; ++a) o3[a] = vw[a] + (1.0 - 2.0 * sx) * (double)k.H[a] + (1.0 - 2.0 * sy) * (double)k.V[a];
        };
        // The reference forms the point on the image plane in fp32 WORLD coordinates, (M + a H) + b V with M = eye + view, and only then
        // subtracts the eye: far from the world origin the ray directions are quantised to the ulp of the eye's coordinates (an eye at
        // 3e6 sees a 0.25-unit grid through a 1-unit view vector) and leave the ideal pixel frustum by far more than a pixel.  Each
        // component of P - eye errs by <= ~ 3 ulp of the largest coordinate involved; in screen units that is a margin of
        // err / (2 |H|) and err / (2 |V|) on top of the pixel the frustum is grown by.  Past a full screen of margin: no culling.
        double sx_margin, sy_margin;
        {
            double big = 0.0, lh = 0.0, lv = 0.0;
            for (int a = 0; a < 3; ++a) {
                big = std::max(big, fabs(eye3[a]) + fabs(vw[a]) + fabs((double)k.H[a]) + fabs((double)k.V[a]));
                lh += (double)k.H[a] * (double)k.H[a]; lv += (double)k.V[a] * (double)k.V[a];
            }
            const double err = 3.0 * 1.1920929e-7 * big * 1.7320508;           // (vector norm of three such components)
            sx_margin = lh > 0.0 ? err / (2.0 * sqrt(lh)) : 1e30;
            sy_margin = lv > 0.0 ? err / (2.0 * sqrt(lv)) : 1e30;
        }
        const bool ray_grid_ok = sx_margin < 1.0 && sy_margin < 1.0;          // (a NaN fails)
        struct Frustum { double nrm[4][3], cc[3], lcc; bool planes_ok, narrow; };
        auto frustumOf = [&](int sp) -> Frustum {
            Frustum f;
            const uint32_t pl1 = (uint32_t)sp * 64u + 63u < (uint32_t)npix ? (uint32_t)sp * 64u + 63u : (uint32_t)npix - 1u;
            const uint32_t g0 = gpix((uint32_t)sp * 64u), g1 = gpix(pl1);
            const int y0 = (int)(g0 / (uint32_t)W), y1 = (int)(g1 / (uint32_t)W);
            int xa = (int)(g0 % (uint32_t)W), xb = (int)(g1 % (uint32_t)W);
            if (y0 != y1) { xa = 0; xb = W - 1; }
            const int ya = y0 < y1 ? y0 : y1, yb = y0 < y1 ? y1 : y0;
            const double sx0 = ((double)xa - 1.0) / (double)k.resx - sx_margin, sx1 = ((double)xb + 2.0) / (double)k.resx + sx_margin;
            const double sy0 = ((double)ya - 1.0) / (double)k.resy - sy_margin, sy1 = ((double)yb + 2.0) / (double)k.resy + sy_margin;
            double cn[4][3];
            dirOf(sx0, sy0, cn[0]); dirOf(sx1, sy0, cn[1]); dirOf(sx1, sy1, cn[2]); dirOf(sx0, sy1, cn[3]);
            dirOf(0.5 * (sx0 + sx1), 0.5 * (sy0 + sy1), f.cc);
            f.planes_ok = ray_grid_ok;
            for (int e = 0; e < 4 && f.planes_ok; ++e) {
                const double *a3 = cn[e], *b3 = cn[(e + 1) & 3];
                const double n3[3] = {a3[1] * b3[2] - b3[1] * a3[2], a3[2] * b3[0] - b3[2] * a3[0], a3[0] * b3[1] - b3[0] * a3[1]};
                const double len = sqrt(n3[0] * n3[0] + n3[1] * n3[1] + n3[2] * n3[2]);
                if (!(len > 1e-12)) { f.planes_ok = false; break; }
                const double sg = (n3[0] * f.cc[0] + n3[1] * f.cc[1] + n3[2] * f.cc[2]) > 0.0 ? -1.0 / len : 1.0 / len;    // outward: the centre is inside
                for (int a = 0; a < 3; ++a) f.nrm[e][a] = n3[a] * sg;
            }
            // "behind the eye" only when every corner direction is within 60 degrees of the centre direction
            f.lcc = sqrt(f.cc[0] * f.cc[0] + f.cc[1] * f.cc[1] + f.cc[2] * f.cc[2]);
            f.narrow = f.planes_ok && f.lcc > 1e-12;
            for (int e = 0; e < 4 && f.narrow; ++e) {
                const double l = sqrt(cn[e][0] * cn[e][0] + cn[e][1] * cn[e][1] + cn[e][2] * cn[e][2]);
                if (!((cn[e][0] * f.cc[0] + cn[e][1] * f.cc[1] + cn[e][2] * f.cc[2]) > 0.5 * l * f.lcc)) f.narrow = false;
            }
            return f;
        };
        // box given relative to the eye
        auto outside = [&](const Frustum &f, const double *lo, const double *hi) -> bool {
            if (!f.planes_ok) return false;
            double scale = 1.0;
            for (int a = 0; a < 3; ++a) scale += fabs(lo[a]) + fabs(hi[a]);
            for (int e = 0; e < 4; ++e) {
                double vmin = 0.0;                  // the box corner deepest inside this plane
                for (int a = 0; a < 3; ++a) { const double x = f.nrm[e][a] * lo[a], y = f.nrm[e][a] * hi[a]; vmin += x < y ? x : y; }
                if (vmin > 1e-5 * scale) return true;
            }
            if (f.narrow) {
                double vmax = 0.0;
                for (int a = 0; a < 3; ++a) { const double x = f.cc[a] * lo[a], y = f.cc[a] * hi[a]; vmax += x > y ? x : y; }
                if (vmax < -1e-5 * scale * f.lcc) return true;
            }
            return false;
        };
        if (want_mask) {
            std::vector<uint32_t> tab((size_t)nspan, 0u);
            for (int sp = 0; sp < nspan; ++sp) {
                const Frustum f = frustumOf(sp);
                uint32_t m = 0u;
                for (int g = 0; g < k.nG; ++g) {
                    const float *bx = &c->h_box_eye[(size_t)g * 8];
                    const double lo[3] = {bx[0], bx[1], bx[2]}, hi[3] = {bx[4], bx[5], bx[6]};
                    if (!outside(f, lo, hi)) m |= 1u << g;
                }
                tab[(size_t)sp] = m;
            }
            if (c->span_mask_cap < tab.size()) {
                if (c->d_span_mask) (void)hipFree(c->d_span_mask);
                c->d_span_mask = nullptr; c->span_mask_cap = 0;
                HIP_TRY(hipMalloc((void **)&c->d_span_mask, tab.size() * sizeof(uint32_t)));
                c->span_mask_cap = tab.size();
            }
            HIP_TRY(hipStreamSynchronize(c->stream));          // (launches of the previous configuration may still read the table)
            HIP_TRY(hipMemcpy(c->d_span_mask, tab.data(), tab.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
            k.span_mask = c->d_span_mask;
        }
        if (want_lists) {
            const uint32_t LIST_MAX = 40;                      // longer than this: the walk is the better deal
            std::vector<uint32_t> off((size_t)nspan * 2, 0u), lst;
            lst.reserve((size_t)nspan * 12);
            const std::vector<ptd::BvhNode> &bn = c->h_bvh;
            std::vector<uint32_t> cand;
            auto relBox = [&](const float *lo, const float *hi, double *rl, double *rh) {
                for (int a = 0; a < 3; ++a) {
                    const double l = (double)lo[a] - eye3[a], h = (double)hi[a] - eye3[a];
                    const double pad = 1e-6 * (fabs(l) + fabs(h) + fabs(eye3[a])) + 1e-6;
                    rl[a] = l - pad; rh[a] = h + pad;
                }
            };
            for (int sp = 0; sp < nspan; ++sp) {
                const Frustum f = frustumOf(sp);
                cand.clear();
                double rl[3], rh[3];
                for (int b = 0; b < k.nbig; ++b) {
                    const size_t g = (size_t)k.big[b];
                    relBox(&c->h_boxes[6 * g], &c->h_boxes[6 * g + 3], rl, rh);
                    if (!outside(f, rl, rh)) cand.push_back((uint32_t)g);
                }
                bool too_many = false;
                for (uint32_t i = 0; i < (uint32_t)k.nnodes && !too_many;) {
                    const ptd::BvhNode &nd = bn[i];
                    relBox(nd.lo, nd.hi, rl, rh);
                    if (outside(f, rl, rh)) { i = nd.skip; continue; }
                    if (nd.prim >= 0) {
                        cand.push_back((uint32_t)nd.prim & 0x3FFFFFFFu);
                        if (cand.size() > LIST_MAX) too_many = true;
                    }
                    i = i + 1u;
                }
                off[2 * (size_t)sp] = (uint32_t)lst.size();
                if (too_many) off[2 * (size_t)sp + 1] = 0xFFFFFFFFu;
                else { off[2 * (size_t)sp + 1] = (uint32_t)cand.size(); lst.insert(lst.end(), cand.begin(), cand.end()); }
            }
            if (lst.empty()) lst.push_back(0u);
            if (c->span_off_cap < off.size()) {
                if (c->d_span_off) (void)hipFree(c->d_span_off);
                c->d_span_off = nullptr; c->span_off_cap = 0;
                HIP_TRY(hipMalloc((void **)&c->d_span_off, off.size() * sizeof(uint32_t)));
                c->span_off_cap = off.size();
            }
            if (c->span_list_cap < lst.size()) {
                if (c->d_span_list) (void)hipFree(c->d_span_list);
                c->d_span_list = nullptr; c->span_list_cap = 0;
                HIP_TRY(hipMalloc((void **)&c->d_span_list, lst.size() * sizeof(uint32_t)));
                c->span_list_cap = lst.size();
            }
            HIP_TRY(hipStreamSynchronize(c->stream));
            HIP_TRY(hipMemcpy(c->d_span_off, off.data(), off.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
            HIP_TRY(hipMemcpy(c->d_span_list, lst.data(), lst.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
            k.span_off = c->d_span_off;
            k.span_list = c->d_span_list;
            if (getenv("PT_DEBUG_CLOCK")) {
                size_t walked = 0;
                for (int sp = 0; sp < nspan; ++sp) walked += off[2 * (size_t)sp + 1] == 0xFFFFFFFFu;
                fprintf(stderr, "[ptamd] camera-ray lists: %d spans, %.1f primitives per listed span, %zu spans walk\n", nspan,
                        (double)lst.size() / (double)(nspan - (int)walked > 0 ? nspan - (int)walked : 1), walked);
            }
        }
    }

    // ray pools: 2 x nshard segments of `segcap` rays x 40 B, carved from one allocation.  A segment must hold
    // every survivor its writers can produce in one launch: each wave appends at most 64 rays per round and runs
    // at most `rounds` rounds (chunks are 64 rays; per-segment padding adds at most NSHARD chunks).
    {
        const long long nw = cfg.workgroup / 64;
        const long long total_waves = (long long)cfg.grid * nw;
        const long long chunks = (nrays + 63) / 64 + pt::NSHARD;
        const long long rounds = (chunks + total_waves - 1) / total_waves;
        long long writers = total_waves;                                   // nshard == 1: everyone writes segment 0
        if (k.nshard > 1) writers = (total_waves + k.nshard - 1) / k.nshard;
        long long segcap = writers * rounds * 64;
        const long long npad = (nrays + 63) & ~63LL;
        if (k.nshard == 1 || segcap > npad) segcap = npad;                 // never more than all rays
        k.segcap = (uint32_t)segcap;
    }
    const size_t slots = (size_t)k.segcap * (size_t)k.nshard;
    const size_t one = slots * (16 + 16 + 8);
    // Device memory per launch sequence: two ray pools of 40 B per slot (<= 1.16 x batch x pixels slots) and the radiance planes,
    // 16 B per (iteration in flight, pixel) -- 6.5 GB per sequence for a 1920x1080 tile at batch 16, ~ 52 GB at the 2^28-ray cap.
    // When the allocation for the sequences asked for fails, the context falls back to ONE sequence before it gives up.
    for (;;) {
        hipError_t e = hipSuccess;
        if (c->lbuf_cap < lbuf_bytes * (size_t)nseq) {
            if (c->d_lbuf) (void)hipFree(c->d_lbuf);
            c->d_lbuf = nullptr; c->lbuf_cap = 0;
            e = hipMalloc((void **)&c->d_lbuf, lbuf_bytes * (size_t)nseq);
            if (e == hipSuccess) {
                c->lbuf_cap = lbuf_bytes * (size_t)nseq;
                HIP_TRY(hipMemsetAsync(c->d_lbuf, 0, c->lbuf_cap, c->stream));
                HIP_TRY(hipStreamSynchronize(c->stream));
                // (fresh planes: the serial numbers may start over as well)
                for (uint32_t sq = 0; sq < (uint32_t)PT_MAX_SEQUENCES; ++sq)
                    HIP_TRY(hipMemcpy(&c->d_state[sq].serial, &sq, sizeof sq, hipMemcpyHostToDevice));
                c->batches_stamped = 0;
            } else c->d_lbuf = nullptr;
        }
        if (e == hipSuccess && c->pool_cap < 2 * one * (size_t)nseq) {
            if (c->d_pool) (void)hipFree(c->d_pool);
            c->d_pool = nullptr; c->pool_cap = 0;
            e = hipMalloc(&c->d_pool, 2 * one * (size_t)nseq);
            if (e == hipSuccess) c->pool_cap = 2 * one * (size_t)nseq;
            else c->d_pool = nullptr;
        }
        if (e == hipSuccess) break;
        (void)hipGetLastError();
        if (nseq > 1) { nseq = 1; c->nseq = 1; continue; }      // (half the pools and planes)
        c->dirty = true;
        return fail(PT_ERR_OOM, "device memory: %zu B of ray pools + %zu B of radiance planes per launch sequence (%s)", 2 * one, lbuf_bytes,
                    hipGetErrorString(e));
    }
    k.lbuf = c->d_lbuf;
    for (int sq = 0; sq < PT_MAX_SEQUENCES; ++sq) {
        pt::KParams &ks = c->kps[sq];
        ks = k;
        if (sq >= nseq) continue;
        for (int q = 0; q < 2; ++q) {
            unsigned char *base = (unsigned char *)c->d_pool + (size_t)(2 * sq + q) * one;
            ks.pool[q].a = (float4 *)base;
            ks.pool[q].b = (float4 *)(base + slots * 16);
            ks.pool[q].c = (float2 *)(base + slots * 32);
        }
        ks.lbuf = (float *)((unsigned char *)c->d_lbuf + (size_t)sq * lbuf_bytes);
        ks.st = c->d_state + sq;
    }
    k = c->kps[0];
    if (getenv("PT_DEBUG_CLOCK")) fprintf(stderr, "[ptamd] launch sequences in flight: %d\n", nseq);

    c->dirty = false;
    return PT_OK;
}

// one launch sequence = one batch of consecutive iterations (their number is device state, IterState::nslot):
// bookkeeping, depth bounce launches, accumulate.
// ev (optional): 2*depth events recorded around the bounce launches (pt_render_profiled).
// sq: the launch sequence (its pools, planes and IterState); with_accumulate = false leaves the accumulate to the caller,
// who orders it behind the previous batch's.
int bounce_launches_per_batch(const pt_ctx *c) { return c->cfg.resident ? 2 : c->kp.depth; }

// Serial numbers stamp the radiance-plane entries (32 bits, + PT_MAX_SEQUENCES per batch): long before they could come round
// again -- 2^29 batches of one context, weeks of rendering -- the planes are zeroed and the count starts over.  Every path that
// renders batches (pt_render, the ordered batches of the motion-blur slices, pt_render_profiled) books them here first.
// PT_SERIAL_BUDGET (environment, read once): a smaller budget, so that tests reach the restart (include/pt_abi.h, test hooks).
int book_batch_serials(pt_ctx *c, unsigned long long nb)
{
    static const unsigned long long serial_budget =
        (getenv("PT_SERIAL_BUDGET") && atoll(getenv("PT_SERIAL_BUDGET")) > 0) ? (unsigned long long)atoll(getenv("PT_SERIAL_BUDGET")) : (1ull << 29);
    if (c->batches_stamped + nb > serial_budget) {
        HIP_TRY(sync_all_streams(c));
        HIP_TRY(hipMemset(c->d_lbuf, 0, c->lbuf_cap));
        for (uint32_t sq = 0; sq < (uint32_t)PT_MAX_SEQUENCES; ++sq)
            HIP_TRY(hipMemcpy(&c->d_state[sq].serial, &sq, sizeof sq, hipMemcpyHostToDevice));
        c->batches_stamped = 0;
    }
    c->batches_stamped += nb;
    return PT_OK;
}

int enqueue_batch(pt_ctx *c, hipStream_t s, hipEvent_t *ev, int sq = 0, bool with_accumulate = true)
{
    const pt::KParams &kp = c->kps[sq];
    HIP_TRY(pt::launch_iter_begin(s, kp.st, kp.npix, kp.depth, c->cfg.compact));
    const int launches = bounce_launches_per_batch(c);      // (resident paths: the camera launch + one for all later bounces)
    for (int b = 0; b < launches; ++b) {
        if (ev) HIP_TRY(hipEventRecord(ev[2 * b], s));
        HIP_TRY(pt::launch_bounce(s, kp, c->cfg, b));
        if (ev) HIP_TRY(hipEventRecord(ev[2 * b + 1], s));
    }
    if (with_accumulate) HIP_TRY(pt::launch_accumulate(s, kp.image, kp.lbuf, kp.st, kp.npix));
    return PT_OK;
}

// iter_count iterations as near-equal batches of at most `batch`: n batches of q, the first r of them one more
// (20 iterations at batch 16 -> 10 + 10, not 16 + 4: a launch that carries 4 iterations pays the same fixed cost)
void batch_schedule(int iter_count, int batch, int *n, int *q, int *r)
{
    *n = (iter_count + batch - 1) / batch;
    *q = iter_count / *n;
    *r = iter_count % *n;
}

// motion blur: scene state of slice k of n = both frames interpolated at shutter time (k + .5)/n, component-wise in fp32
// as a + (b - a) * t on translation / rotation / scale (and the camera vectors), matrices rebuilt by the loader's
// buildTransformationMatrix (ref: src/utilities.cpp:74-90)
float lerp1(float a, float b, float t) { return a + (b - a) * t; }
pt_vec3 lerp3(pt_vec3 a, pt_vec3 b, float t) { return {lerp1(a.x, b.x, t), lerp1(a.y, b.y, t), lerp1(a.z, b.z, t)}; }

void drop_slices(pt_ctx *c)
{
    for (pt_ctx *k : c->slice_ctx) pt_destroy(k);
    c->slice_ctx.clear();
}

int build_slices(pt_ctx *c)
{
    if (!c->motion_dirty) return PT_OK;
    drop_slices(c);
    const int n = c->motion_slices;
    for (int k = 0; k < n; ++k) {
        pt_ctx *ch = nullptr;
        int rc = pt_create(c->device, &ch);
        if (rc != PT_OK) return rc;
        c->slice_ctx.push_back(ch);
        const float t = ((float)k + 0.5f) / (float)n;
        std::vector<pt_static_geom> g = c->geoms;
        for (size_t i = 0; i < g.size(); ++i) {
            const pt_static_geom &a = c->geoms[i], &b = c->geoms_next[i];
            g[i].translation = lerp3(a.translation, b.translation, t);
            g[i].rotation = lerp3(a.rotation, b.rotation, t);
            g[i].scale = lerp3(a.scale, b.scale, t);
            g[i].transform = ptamd::buildTransformationMatrix(g[i].translation, g[i].rotation, g[i].scale, c->motion_rotat, &g[i].inverseTransform);
        }
        pt_camera_data cam = c->cam;
        if (c->have_cam_next) {
            cam.position = lerp3(c->cam.position, c->cam_next.position, t);
            cam.view = lerp3(c->cam.view, c->cam_next.view, t);
            cam.up = lerp3(c->cam.up, c->cam_next.up, t);
        }
        rc = pt_set_options(ch, &c->opt);
        if (rc == PT_OK) rc = pt_set_scene(ch, g.data(), (int)g.size(), c->mats.data(), (int)c->mats.size());
        if (rc == PT_OK && !c->tri_geom.empty()) {
            // the same object-space triangles under the slice's transforms: one pt_mesh per run of triangles of one geom
            std::vector<pt_mesh> ms;
            for (size_t a0 = 0; a0 < c->tri_geom.size();) {
                size_t a1 = a0;
                while (a1 < c->tri_geom.size() && c->tri_geom[a1] == c->tri_geom[a0]) ++a1;
                ms.push_back(pt_mesh{c->tri_geom[a0], (int)(a1 - a0), c->tri_obj.data() + 9 * a0});
                a0 = a1;
            }
            rc = pt_set_meshes(ch, ms.data(), (int)ms.size());
        }
        if (rc == PT_OK) rc = pt_set_camera(ch, &cam);
        if (rc != PT_OK) return rc;
    }
    c->motion_dirty = false;
    return PT_OK;
}

// One batch (count <= the context's batch size) of context c on stream ss, for a caller that keeps several contexts busy on two
// streams (motion blur by slices): bookkeeping and bounce launches as soon as the stream gets to them, the accumulate behind
// `wait_ev` (the previous batch's accumulate, possibly on the other stream; nullptr: none), `done_ev` recorded behind it.
int render_batch_ordered(pt_ctx *c, hipStream_t ss, int iter_first, int count, hipEvent_t wait_ev, hipEvent_t done_ev)
{
    int rc = configure(c);
    if (rc != PT_OK) return rc;
    if (count < 1 || count > c->batch) return fail(PT_ERR_INVALID, "render_batch_ordered: %d iterations (batch %d)", count, c->batch);
    rc = book_batch_serials(c, 1ull);
    if (rc != PT_OK) return rc;
    if (c->opt.use_graph && !c->graph_exec[0]) {        // slot 0: sequence 0's batch without its accumulate (as in pt_render)
        HIP_TRY(hipStreamBeginCapture(ss, hipStreamCaptureModeThreadLocal));
        rc = enqueue_batch(c, ss, nullptr, 0, false);
        hipError_t ce = hipStreamEndCapture(ss, &c->graph[0]);
        if (rc != PT_OK) { drop_graph(c); return rc; }
        if (ce != hipSuccess) { drop_graph(c); return fail(PT_ERR_HIP, "hipStreamEndCapture: %s", hipGetErrorString(ce)); }
        HIP_TRY(hipGraphInstantiate(&c->graph_exec[0], c->graph[0], nullptr, nullptr, 0));
    }
    HIP_TRY(pt::launch_iter_set(ss, c->d_state, (uint32_t)iter_first, (uint32_t)count, 0u, 0u, 1u));
    if (c->opt.use_graph) HIP_TRY(hipGraphLaunch(c->graph_exec[0], ss));
    else { rc = enqueue_batch(c, ss, nullptr, 0, false); if (rc != PT_OK) return rc; }
    if (wait_ev) HIP_TRY(hipStreamWaitEvent(ss, wait_ev, 0));
    const pt::KParams &ks = c->kps[0];
    HIP_TRY(pt::launch_accumulate(ss, ks.image, ks.lbuf, ks.st, ks.npix));
    HIP_TRY(pt::launch_iter_fold(ss, c->d_state, c->kp.depth));
    if (done_ev) HIP_TRY(hipEventRecord(done_ev, ss));
    c->bounce_launches += (unsigned long long)bounce_launches_per_batch(c);
    c->image_valid = true;
    return PT_OK;
}

}  // namespace

extern "C" {

const char *pt_last_error(void) { return g_last_error.c_str(); }
const char *pt_version(void) { return "ptamd 0.1 (gfx950, abi 2)"; }
int pt_abi_version(void) { return PT_ABI_VERSION; }
size_t pt_options_size(void) { return sizeof(pt_options); }

int pt_strip_local_rows(int height, int strip_rows, int world, int rank)
{
    if (height < 1 || strip_rows < 1 || world < 1 || rank < 0 || rank >= world) return 0;
    const long long nstrips = ((long long)height + strip_rows - 1) / strip_rows;      // the last one may be short
    long long rows = 0;
    for (long long k = rank; k < nstrips; k += world) {
        const long long left = (long long)height - k * strip_rows;
        rows += left < strip_rows ? left : strip_rows;
    }
    return (int)rows;
}

int pt_strip_global_row(int strip_rows, int world, int rank, int local_row)
{
    if (strip_rows < 1 || world < 1 || rank < 0 || rank >= world || local_row < 0) return -1;
    const int j = local_row / strip_rows, w = local_row % strip_rows;
    return (j * world + rank) * strip_rows + w;
}

int pt_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

void pt_default_options(pt_options *o)
{
    if (!o) return;
    memset(o, 0, sizeof *o);
    o->depth = 8;
    o->rr_start = -1;
    o->seed = 0;
    o->compaction = 1;
    o->workgroup = 0;
    o->geom_path = 0;
    o->use_graph = 1;
    o->batch = 0;
    o->lens_radius = 0.0f;
    o->focal_distance = 1.0f;
}

int pt_create(int device, pt_ctx **out)
{
    if (!out) return fail(PT_ERR_INVALID, "pt_create: out is NULL");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n < 1)
        return fail(PT_ERR_NO_DEVICE, "no HIP device (%s); this renderer has no CPU path", hipGetErrorString(e));
    if (device < 0 || device >= n) return fail(PT_ERR_INVALID, "device %d out of range (0..%d)", device, n - 1);
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(PT_ERR_NO_DEVICE, "device %d is %s; kernels are built for gfx950 (MI355X) only", device, prop.gcnArchName);
    HIP_TRY(hipSetDevice(device));
    pt_ctx *c = new pt_ctx();
    c->device = device;
    c->cu_count = prop.multiProcessorCount;
    pt_default_options(&c->opt);
    if (hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess) {
        delete c;
        return fail(PT_ERR_HIP, "hipStreamCreate failed");
    }
    c->stream = c->own_stream;
    if (hipMalloc((void **)&c->d_state, PT_MAX_SEQUENCES * sizeof(pt::IterState)) != hipSuccess ||
        hipMemset(c->d_state, 0, PT_MAX_SEQUENCES * sizeof(pt::IterState)) != hipSuccess) {
        pt_destroy(c);
        return fail(PT_ERR_OOM, "cannot allocate iteration state");
    }
    // batch serial numbers (the stamps of the radiance-plane entries): sequence q counts q + 4, q + 8, ... -- never zero, never
    // another sequence's, never one of its own again, so no entry an earlier batch left anywhere in the planes is ever mistaken
    for (uint32_t q = 1; q < (uint32_t)PT_MAX_SEQUENCES; ++q)
        if (hipMemcpy(&c->d_state[q].serial, &q, sizeof q, hipMemcpyHostToDevice) != hipSuccess) {
            pt_destroy(c);
            return fail(PT_ERR_HIP, "cannot initialise the iteration state");
        }
    *out = c;
    return PT_OK;
}

void pt_destroy(pt_ctx *c)
{
    if (!c) return;
    drop_slices(c);
    (void)hipSetDevice(c->device);
    if (c->stream) (void)sync_all_streams(c);
    drop_graph(c);
    for (auto &t : c->timers) { (void)hipEventDestroy(t.first); (void)hipEventDestroy(t.second); }
    if (c->d_prims) (void)hipFree(c->d_prims);
    if (c->d_mats) (void)hipFree(c->d_mats);
    if (c->d_ro_eye) (void)hipFree(c->d_ro_eye);
    if (c->d_face_n) (void)hipFree(c->d_face_n);
    if (c->d_box_eye) (void)hipFree(c->d_box_eye);
    if (c->d_span_mask) (void)hipFree(c->d_span_mask);
    if (c->d_box_world) (void)hipFree(c->d_box_world);
    if (c->d_slab_n) (void)hipFree(c->d_slab_n);
    if (c->d_lights) (void)hipFree(c->d_lights);
    if (c->d_light_tris) (void)hipFree(c->d_light_tris);
    if (c->d_light_cdf) (void)hipFree(c->d_light_cdf);
    if (c->d_knots) (void)hipFree(c->d_knots);
    if (c->d_knot_cam) (void)hipFree(c->d_knot_cam);
    if (c->d_bvh) (void)hipFree(c->d_bvh);
    if (c->d_bvh4) (void)hipFree(c->d_bvh4);
    if (c->d_image_own) (void)hipFree(c->d_image_own);
    if (c->d_pool) (void)hipFree(c->d_pool);
    if (c->d_lbuf) (void)hipFree(c->d_lbuf);
    if (c->d_state) (void)hipFree(c->d_state);
    if (c->d_span_off) (void)hipFree(c->d_span_off);
    if (c->d_span_list) (void)hipFree(c->d_span_list);
    for (int q = 1; q < PT_MAX_SEQUENCES; ++q)
        if (c->seq_stream[q]) { (void)hipStreamSynchronize(c->seq_stream[q]); (void)hipStreamDestroy(c->seq_stream[q]); }
    for (int q = 0; q < PT_MAX_SEQUENCES; ++q)
        if (c->seq_acc[q]) (void)hipEventDestroy(c->seq_acc[q]);
    if (c->seq_fork) (void)hipEventDestroy(c->seq_fork);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
}

int pt_set_options(pt_ctx *c, const pt_options *o)
{
    if (!c || !o) return fail(PT_ERR_INVALID, "pt_set_options: NULL argument");
    if (o->depth < 1 || o->depth > PT_MAX_DEPTH) return fail(PT_ERR_INVALID, "depth %d not in 1..%d", o->depth, PT_MAX_DEPTH);
    const int wg = o->workgroup;
    if (!(wg == 0 || wg == 64 || wg == 128 || wg == 256 || wg == 512 || wg == 1024))
        return fail(PT_ERR_INVALID, "workgroup %d not one of 0,64,128,256,512,1024", wg);
    if (o->geom_path < 0 || o->geom_path > 8) return fail(PT_ERR_INVALID, "geom_path %d not in 0..8", o->geom_path);
    if (o->row_begin < 0 || o->row_end < o->row_begin) return fail(PT_ERR_INVALID, "tile rows [%d,%d)", o->row_begin, o->row_end);
    if (o->strip_rows < 0 || (o->strip_rows > 0 && (o->strip_world < 1 || o->strip_rank < 0 || o->strip_rank >= o->strip_world)))
        return fail(PT_ERR_INVALID, "strips: rows %d, rank %d of %d", o->strip_rows, o->strip_rank, o->strip_world);
    if (o->strip_rows > 0 && (o->row_begin != 0 || o->row_end != 0))
        return fail(PT_ERR_INVALID, "strip tiles and a row band [%d,%d) exclude each other", o->row_begin, o->row_end);
    if (o->batch < 0 || o->batch > pt::PT_MAX_BATCH) return fail(PT_ERR_INVALID, "batch %d not in 0..%d", o->batch, pt::PT_MAX_BATCH);
    if (o->compaction < 0 || o->compaction > 2) return fail(PT_ERR_INVALID, "compaction %d not 0, 1 or 2", o->compaction);
    if (o->direct_light < 0 || o->direct_light > 1) return fail(PT_ERR_INVALID, "direct_light %d not 0 or 1", o->direct_light);
    if (o->absorption < 0 || o->absorption > 1) return fail(PT_ERR_INVALID, "absorption %d not 0 or 1", o->absorption);
    if (o->scatter < 0 || o->scatter > 1) return fail(PT_ERR_INVALID, "scatter %d not 0 or 1", o->scatter);
    if (!(o->lens_radius >= 0.0f) || (o->lens_radius > 0.0f && !(o->focal_distance > 0.0f)))
        return fail(PT_ERR_INVALID, "lens radius %g / focal distance %g", (double)o->lens_radius, (double)o->focal_distance);
    if (o->motion_per_ray < 0 || o->motion_per_ray > 1) return fail(PT_ERR_INVALID, "motion_per_ray %d not 0 or 1", o->motion_per_ray);
    if (o->sequences < 0 || o->sequences > PT_MAX_SEQUENCES) return fail(PT_ERR_INVALID, "sequences %d not in 0..%d", o->sequences, PT_MAX_SEQUENCES);
    if (o->resident < -1 || o->resident > 1) return fail(PT_ERR_INVALID, "resident %d not -1, 0 or 1", o->resident);
    if (o->direct_light && o->compaction != 1) return fail(PT_ERR_INVALID, "direct_light needs compaction 1 (got %d)", o->compaction);
    if (o->scatter && o->compaction != 1) return fail(PT_ERR_INVALID, "scatter needs compaction 1 (got %d)", o->compaction);
    c->opt = *o;
    c->dirty = true;
    c->motion_dirty = c->motion_slices >= 1;
    return PT_OK;
}

int pt_get_options(pt_ctx *c, pt_options *o)
{
    if (!c || !o) return fail(PT_ERR_INVALID, "pt_get_options: NULL argument");
    *o = c->opt;
    return PT_OK;
}

int pt_set_scene(pt_ctx *c, const pt_static_geom *geoms, int nG, const pt_material *mats, int nM)
{
    if (!c || nG < 0 || nM < 0 || (nG > 0 && !geoms) || (nM > 0 && !mats))
        return fail(PT_ERR_INVALID, "pt_set_scene: bad arguments (nG=%d nM=%d)", nG, nM);
    for (int i = 0; i < nG; ++i) {
        if (geoms[i].type < PT_SPHERE || geoms[i].type > PT_MESH)
            return fail(PT_ERR_INVALID, "geom %d: type %d is not SPHERE/CUBE/MESH", i, geoms[i].type);
        if (geoms[i].type != PT_MESH && (geoms[i].materialid < 0 || geoms[i].materialid >= nM))
            return fail(PT_ERR_INVALID, "geom %d: materialid %d outside 0..%d", i, geoms[i].materialid, nM - 1);
    }
    c->geoms.assign(geoms, geoms + nG);
    c->mats.assign(mats, mats + nM);
    c->tri_obj.clear();
    c->tri_geom.clear();
    c->geoms_next.clear();                 // motion blur belongs to the scene it was set for
    c->motion_slices = 0;
    drop_slices(c);
    c->have_scene = true;
    c->dirty = true;
    return PT_OK;
}

int pt_set_meshes(pt_ctx *c, const pt_mesh *meshes, int n)
{
    if (!c || n < 0 || (n > 0 && !meshes)) return fail(PT_ERR_INVALID, "pt_set_meshes: bad arguments (n=%d)", n);
    if (!c->have_scene) return fail(PT_ERR_INVALID, "pt_set_meshes: call pt_set_scene first");
    std::vector<float> tri;
    std::vector<int> owner;
    for (int k = 0; k < n; ++k) {
        const pt_mesh &m = meshes[k];
        if (m.geom < 0 || m.geom >= (int)c->geoms.size() || c->geoms[(size_t)m.geom].type != PT_MESH)
            return fail(PT_ERR_INVALID, "mesh %d: geom %d is not a MESH object", k, m.geom);
        if (m.n_triangles < 0 || (m.n_triangles > 0 && !m.vertices)) return fail(PT_ERR_INVALID, "mesh %d: %d triangles, no vertices", k, m.n_triangles);
        const int mat = c->geoms[(size_t)m.geom].materialid;
        if (m.n_triangles > 0 && (mat < 0 || mat >= (int)c->mats.size()))
            return fail(PT_ERR_INVALID, "mesh %d: materialid %d of geom %d outside 0..%d", k, mat, m.geom, (int)c->mats.size() - 1);
        tri.insert(tri.end(), m.vertices, m.vertices + 9 * (size_t)m.n_triangles);
        owner.insert(owner.end(), (size_t)m.n_triangles, m.geom);
    }
    c->tri_obj.swap(tri);
    c->tri_geom.swap(owner);
    c->dirty = true;
    c->motion_dirty = c->motion_slices >= 1;
    return PT_OK;
}

int pt_set_motion(pt_ctx *c, const pt_static_geom *geoms_next, const pt_camera_data *cam_next, int slices, int rotat_units)
{
    if (!c) return fail(PT_ERR_INVALID, "pt_set_motion: NULL context");
    if (slices < 1 || !geoms_next) {                     // motion off
        c->motion_slices = 0;
        c->geoms_next.clear();
        c->have_cam_next = false;
        drop_slices(c);
        c->dirty = true;
        return PT_OK;
    }
    if (!c->have_scene || !c->have_camera) return fail(PT_ERR_INVALID, "pt_set_motion: call pt_set_scene and pt_set_camera first");
    if (slices > 64) return fail(PT_ERR_INVALID, "pt_set_motion: %d slices (at most 64)", slices);
    if (rotat_units != PT_ROTAT_RADIANS && rotat_units != PT_ROTAT_DEGREES) return fail(PT_ERR_INVALID, "pt_set_motion: rotat_units %d", rotat_units);
    for (size_t i = 0; i < c->geoms.size(); ++i)
        if (geoms_next[i].type != c->geoms[i].type || geoms_next[i].materialid != c->geoms[i].materialid)
            return fail(PT_ERR_INVALID, "pt_set_motion: object %d changes type or material between the frames", (int)i);
    c->geoms_next.assign(geoms_next, geoms_next + c->geoms.size());
    c->have_cam_next = cam_next != nullptr;
    if (cam_next) c->cam_next = *cam_next;
    c->motion_slices = slices;
    c->motion_rotat = rotat_units;
    c->motion_dirty = true;
    c->dirty = true;                                      // (per-ray mode: the knots are part of the configuration)
    return PT_OK;
}

int pt_set_camera(pt_ctx *c, const pt_camera_data *cam)
{
    if (!c || !cam) return fail(PT_ERR_INVALID, "pt_set_camera: NULL argument");
    if (!(cam->resolution.x >= 1.0f) || !(cam->resolution.y >= 1.0f))
        return fail(PT_ERR_INVALID, "camera resolution %gx%g", cam->resolution.x, cam->resolution.y);
    c->cam = *cam;
    c->have_camera = true;
    c->dirty = true;
    c->motion_dirty = c->motion_slices >= 1;
    return PT_OK;
}

int pt_set_stream(pt_ctx *c, void *hip_stream)
{
    if (!c) return fail(PT_ERR_INVALID, "pt_set_stream: NULL context");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(sync_all_streams(c));
    if (fold_timers(c) != PT_OK) return PT_ERR_HIP;
    drop_graph(c);
    c->stream = hip_stream ? (hipStream_t)hip_stream : c->own_stream;
    return PT_OK;
}

size_t pt_image_bytes(pt_ctx *c)
{
    if (!c || !c->have_camera) return 0;
    const int W = (int)c->cam.resolution.x, H = (int)c->cam.resolution.y;
    int r0 = c->opt.row_begin, r1 = c->opt.row_end;
    if (r0 == 0 && r1 == 0) r1 = H;
    if (r1 > H || r0 >= r1) return 0;
    int rows = r1 - r0;
    if (c->opt.strip_rows > 0) rows = pt_strip_local_rows(H, c->opt.strip_rows, c->opt.strip_world, c->opt.strip_rank);
    if (rows < 1) return 0;
    return (size_t)W * (size_t)rows * 3 * sizeof(float);
}

int pt_bind_image(pt_ctx *c, void *device_rgb)
{
    if (!c) return fail(PT_ERR_INVALID, "pt_bind_image: NULL context");
    c->d_image_bound = (float *)device_rgb;
    c->image_valid = false;
    c->dirty = true;
    return PT_OK;
}

int pt_image_device_pointer(pt_ctx *c, void **out)
{
    if (!c || !out) return fail(PT_ERR_INVALID, "pt_image_device_pointer: NULL argument");
    int rc = configure(c);
    if (rc != PT_OK) return rc;
    *out = image_ptr(c);
    return PT_OK;
}

int pt_clear_image(pt_ctx *c)
{
    if (!c) return fail(PT_ERR_INVALID, "pt_clear_image: NULL context");
    int rc = configure(c);
    if (rc != PT_OK) return rc;
    HIP_TRY(hipMemsetAsync(image_ptr(c), 0, c->image_bytes, c->stream));
    c->image_valid = false;
    return PT_OK;
}

int pt_upload_image(pt_ctx *c, const float *host_rgb)
{
    if (!c || !host_rgb) return fail(PT_ERR_INVALID, "pt_upload_image: NULL argument");
    int rc = configure(c);
    if (rc != PT_OK) return rc;
    HIP_TRY(hipMemcpyAsync(image_ptr(c), host_rgb, c->image_bytes, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->image_valid = true;
    return PT_OK;
}

int pt_download_image(pt_ctx *c, float *host_rgb)
{
    if (!c || !host_rgb) return fail(PT_ERR_INVALID, "pt_download_image: NULL argument");
    int rc = configure(c);
    if (rc != PT_OK) return rc;
    HIP_TRY(hipMemcpyAsync(host_rgb, image_ptr(c), c->image_bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return PT_OK;
}

int pt_render(pt_ctx *c, int iter_first, int iter_count)
{
    if (!c) return fail(PT_ERR_INVALID, "pt_render: NULL context");
    if (iter_first < 1 || iter_count < 0) return fail(PT_ERR_INVALID, "pt_render: iterations [%d,+%d)", iter_first, iter_count);
    int rc = configure(c);
    if (rc != PT_OK) return rc;
    if (iter_count == 0) return PT_OK;
    hipStream_t s = c->stream;
    if (!motion_by_slices(c) && !c->slice_ctx.empty()) drop_slices(c);      // (left over from the slice scheme)
    if (motion_by_slices(c)) {
        // motion blur: every run of PT_SLICE_ITERATIONS iterations belongs to one shutter slice, rendered by that
        // slice's context into this context's framebuffer (the running mean is stateless given image and iteration)
        rc = build_slices(c);
        if (rc != PT_OK) return rc;
        for (pt_ctx *ch : c->slice_ctx) {
            if (ch->d_image_bound != image_ptr(c)) { rc = pt_bind_image(ch, image_ptr(c)); if (rc != PT_OK) return rc; }
            if (ch->stream != s) { rc = pt_set_stream(ch, (void *)s); if (rc != PT_OK) return rc; }
        }
        // Consecutive runs belong to different slices -- different contexts with pools of their own --, so they alternate
        // between the render stream and a second one like the launch sequences of a static scene do: a run's launches fill
        // the tails of its neighbour's, only the accumulates into the shared framebuffer are chained (iteration order).
        // (A child whose batch is smaller than a run -- huge tiles -- takes the plain, serial path.)
        bool two_streams = c->nseq > 1;
        for (pt_ctx *ch : c->slice_ctx) {
            rc = configure(ch);
            if (rc != PT_OK) return rc;
            if (ch->batch < PT_SLICE_ITERATIONS) two_streams = false;
        }
        if (two_streams) {
            if (!c->seq_stream[1]) HIP_TRY(hipStreamCreateWithFlags(&c->seq_stream[1], hipStreamNonBlocking));
            for (int sq = 0; sq < 2; ++sq)
                if (!c->seq_acc[sq]) HIP_TRY(hipEventCreateWithFlags(&c->seq_acc[sq], hipEventDisableTiming | hipEventDisableSystemFence));
            if (!c->seq_fork) HIP_TRY(hipEventCreateWithFlags(&c->seq_fork, hipEventDisableTiming | hipEventDisableSystemFence));
            if (c->timers.size() >= 1024) { rc = fold_timers(c); if (rc != PT_OK) return rc; }
            hipEvent_t e0, e1;
            HIP_TRY(hipEventCreateWithFlags(&e0, hipEventDisableSystemFence));
            HIP_TRY(hipEventCreateWithFlags(&e1, hipEventDisableSystemFence));
            c->timers.emplace_back(e0, e1);
            HIP_TRY(hipEventRecord(e0, s));
            HIP_TRY(hipEventRecord(c->seq_fork, s));
            HIP_TRY(hipStreamWaitEvent(c->seq_stream[1], c->seq_fork, 0));
            int j = 0;
            for (int it = iter_first; it < iter_first + iter_count; ++j) {
                const int run = (it - 1) / PT_SLICE_ITERATIONS;
                int end = (run + 1) * PT_SLICE_ITERATIONS + 1;
                if (end > iter_first + iter_count) end = iter_first + iter_count;
                hipStream_t ss = (j & 1) ? c->seq_stream[1] : s;
                rc = render_batch_ordered(c->slice_ctx[(size_t)(run % c->motion_slices)], ss, it, end - it,
                                          j > 0 ? c->seq_acc[(j - 1) & 1] : nullptr, c->seq_acc[j & 1]);
                if (rc != PT_OK) { (void)sync_all_streams(c); c->image_valid = false; return rc; }      // (nothing keeps running on the second stream)
                it = end;
            }
            if (j > 1) HIP_TRY(hipStreamWaitEvent(s, c->seq_acc[1], 0));        // join (the last event of the second stream)
            HIP_TRY(hipEventRecord(e1, s));
            c->image_valid = true;
            return PT_OK;
        }
        for (int it = iter_first; it < iter_first + iter_count;) {
            const int run = (it - 1) / PT_SLICE_ITERATIONS;
            int end = (run + 1) * PT_SLICE_ITERATIONS + 1;
            if (end > iter_first + iter_count) end = iter_first + iter_count;
            rc = pt_render(c->slice_ctx[(size_t)(run % c->motion_slices)], it, end - it);
            if (rc != PT_OK) return rc;
            it = end;
        }
        c->image_valid = true;
        return PT_OK;
    }

    int nb, q, r;
    batch_schedule(iter_count, c->batch, &nb, &q, &r);
    // sequences actually used by this call (a call of one batch runs on sequence 0 alone)
    const int nseq = nb < c->nseq ? nb : c->nseq;
    const bool own_acc = nseq > 1;       // the accumulates are ordered across the sequences by events, outside the graphs
    c->seq_stream[0] = s;
    if (c->nseq > 1) {
        for (int sq = 1; sq < c->nseq; ++sq)
            if (!c->seq_stream[sq]) HIP_TRY(hipStreamCreateWithFlags(&c->seq_stream[sq], hipStreamNonBlocking));
        for (int sq = 0; sq < c->nseq; ++sq)
            if (!c->seq_acc[sq]) HIP_TRY(hipEventCreateWithFlags(&c->seq_acc[sq], hipEventDisableTiming | hipEventDisableSystemFence));
        if (!c->seq_fork) HIP_TRY(hipEventCreateWithFlags(&c->seq_fork, hipEventDisableTiming | hipEventDisableSystemFence));
    }
    if (c->opt.use_graph) {
        // every graph this configuration can replay is captured by its first pt_render, whatever that call's size: slots
        // [0, nseq) = a sequence's batch without the accumulate, slot PT_MAX_SEQUENCES = sequence 0's with it (one-batch
        // calls) -- a later, longer call then meets no capture / instantiate (milliseconds of host time) on its way
        for (int gs = 0; gs <= PT_MAX_SEQUENCES; ++gs) {
            const bool whole = gs == PT_MAX_SEQUENCES;
            if ((!whole && (c->nseq < 2 || gs >= c->nseq)) || c->graph_exec[gs]) continue;
            const int sq = whole ? 0 : gs;
            hipStream_t ss = c->seq_stream[sq];
            HIP_TRY(hipStreamBeginCapture(ss, hipStreamCaptureModeThreadLocal));
            rc = enqueue_batch(c, ss, nullptr, sq, whole);
            hipError_t ce = hipStreamEndCapture(ss, &c->graph[gs]);
            if (rc != PT_OK) { drop_graph(c); return rc; }
            if (ce != hipSuccess) { drop_graph(c); return fail(PT_ERR_HIP, "hipStreamEndCapture: %s", hipGetErrorString(ce)); }
            HIP_TRY(hipGraphInstantiate(&c->graph_exec[gs], c->graph[gs], nullptr, nullptr, 0));
        }
    }
    rc = book_batch_serials(c, (unsigned long long)nb);
    if (rc != PT_OK) return rc;
    if (c->timers.size() >= 1024) { rc = fold_timers(c); if (rc != PT_OK) return rc; }
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreateWithFlags(&e0, hipEventDisableSystemFence));
    HIP_TRY(hipEventCreateWithFlags(&e1, hipEventDisableSystemFence));
    c->timers.emplace_back(e0, e1);
    HIP_TRY(hipEventRecord(e0, s));

    // fork ... join: a failure in between must not leave a side stream running (it would still touch the pools, the planes and
    // the shared framebuffer while the caller reconfigures or frees them): the lambda's error returns land in the wait below
    auto forked = [&]() -> int {
        if (nseq > 1) {
            // fork: the other sequences start behind everything already on the render stream
            HIP_TRY(hipEventRecord(c->seq_fork, s));
            for (int sq = 1; sq < nseq; ++sq) HIP_TRY(hipStreamWaitEvent(c->seq_stream[sq], c->seq_fork, 0));
        }
        for (int sq = 0; sq < nseq; ++sq)
            HIP_TRY(pt::launch_iter_set(c->seq_stream[sq], c->d_state + sq, (uint32_t)iter_first, (uint32_t)q, (uint32_t)r, (uint32_t)sq, (uint32_t)nseq));
        for (int i = 0; i < nb; ++i) {
            const int sq = i % nseq;
            hipStream_t ss = c->seq_stream[sq];
            if (c->opt.use_graph) {
                const int gs = own_acc ? sq : PT_MAX_SEQUENCES;       // (a one-batch call runs on sequence 0 with its accumulate)
                HIP_TRY(hipGraphLaunch(c->graph_exec[gs], ss));
            } else {
                const int rc2 = enqueue_batch(c, ss, nullptr, sq, !own_acc);
                if (rc2 != PT_OK) return rc2;
            }
            if (own_acc) {
                // the running mean takes the batches in iteration order: batch i's accumulate behind batch i - 1's
                if (i > 0) HIP_TRY(hipStreamWaitEvent(ss, c->seq_acc[(i - 1) % nseq], 0));
                const pt::KParams &ks = c->kps[sq];
                HIP_TRY(pt::launch_accumulate(ss, ks.image, ks.lbuf, ks.st, ks.npix));
                HIP_TRY(hipEventRecord(c->seq_acc[sq], ss));
            }
        }
        for (int sq = 0; sq < nseq; ++sq) HIP_TRY(pt::launch_iter_fold(c->seq_stream[sq], c->d_state + sq, c->kp.depth));
        if (nseq > 1) {
            // join: the render stream continues behind every sequence
            for (int sq = 1; sq < nseq; ++sq) {
                HIP_TRY(hipEventRecord(c->seq_acc[sq], c->seq_stream[sq]));
                HIP_TRY(hipStreamWaitEvent(s, c->seq_acc[sq], 0));
            }
        }
        HIP_TRY(hipEventRecord(e1, s));
        return PT_OK;
    };
    rc = forked();
    if (rc != PT_OK) {
        (void)sync_all_streams(c);
        c->image_valid = false;
        return rc;
    }
    c->bounce_launches += (unsigned long long)nb * (unsigned long long)bounce_launches_per_batch(c);
    c->image_valid = true;
    return PT_OK;
}

int pt_get_launch_info(pt_ctx *c, pt_launch_info *out)
{
    if (!c || !out) return fail(PT_ERR_INVALID, "pt_get_launch_info: NULL argument");
    int rc = configure(c);
    if (rc != PT_OK) return rc;
    memset(out, 0, sizeof *out);
    out->geom_path = c->cfg.geom + 1;
    out->workgroup = c->cfg.workgroup;
    out->grid = c->cfg.grid;
    out->batch = c->batch;
    out->sequences = c->nseq;
    out->resident = c->cfg.resident;
    out->refill_min = c->kp.refill_min;
    out->launches_per_batch = bounce_launches_per_batch(c);
    out->lds_bytes = (int)pt::bounce_lds_bytes(c->kp, c->cfg);
    out->slab_pretest = c->cfg.slab;
    return PT_OK;
}

int pt_render_profiled(pt_ctx *c, int iter_first, int iter_count, double *bounce_ms_out)
{
    if (!c || !bounce_ms_out) return fail(PT_ERR_INVALID, "pt_render_profiled: NULL argument");
    if (iter_first < 1 || iter_count < 0) return fail(PT_ERR_INVALID, "pt_render_profiled: iterations [%d,+%d)", iter_first, iter_count);
    if (motion_by_slices(c)) return fail(PT_ERR_INVALID, "pt_render_profiled: not with motion blur by slices (profile a slice's scene instead)");
    int rc = configure(c);
    if (rc != PT_OK) return rc;
    const int depth = c->kp.depth;
    for (int b = 0; b < depth; ++b) bounce_ms_out[b] = 0.0;
    if (iter_count == 0) return PT_OK;
    hipStream_t s = c->stream;
    std::vector<hipEvent_t> ev((size_t)2 * (size_t)depth, nullptr);
    // (every exit below destroys the events that exist)
    struct EventGuard {
        std::vector<hipEvent_t> &ev;
        ~EventGuard() { for (auto &e : ev) if (e) { (void)hipEventDestroy(e); e = nullptr; } }
    } guard{ev};
    // no system-scope fence at the events: a default event makes every kernel end with an L2 write-back and start with
    // a cold cache, which showed up as +12 % on the launches bracketed this way
    for (auto &e : ev) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableSystemFence));
    int nb, q, r;
    batch_schedule(iter_count, c->batch, &nb, &q, &r);
    rc = book_batch_serials(c, (unsigned long long)nb);
    if (rc != PT_OK) return rc;
    HIP_TRY(pt::launch_iter_set(s, c->d_state, (uint32_t)iter_first, (uint32_t)q, (uint32_t)r, 0u, 1u));
    int launches = 0;
    for (int i = 0; i < nb; ++i) {
        rc = enqueue_batch(c, s, ev.data());
        if (rc != PT_OK) { (void)hipStreamSynchronize(s); return rc; }
        launches++;
        HIP_TRY(hipStreamSynchronize(s));
        for (int b = 0; b < bounce_launches_per_batch(c); ++b) {      // (resident paths: [1] = the one launch of all later bounces)
            float ms = 0.0f;
            HIP_TRY(hipEventElapsedTime(&ms, ev[2 * b], ev[2 * b + 1]));
            bounce_ms_out[b] += (double)ms;
        }
    }
    HIP_TRY(pt::launch_iter_fold(s, c->d_state, depth));
    HIP_TRY(hipStreamSynchronize(s));
    c->bounce_launches += (unsigned long long)launches * (unsigned long long)bounce_launches_per_batch(c);
    c->image_valid = true;
    return PT_OK;
}

int pt_selftest_math(pt_ctx *c, unsigned long long mismatches_out[3])
{
    if (!c || !mismatches_out) return fail(PT_ERR_INVALID, "pt_selftest_math: NULL argument");
    HIP_TRY(hipSetDevice(c->device));
    unsigned long long *d = nullptr;
    HIP_TRY(hipMalloc((void **)&d, 3 * sizeof(unsigned long long)));
    hipError_t e = hipMemsetAsync(d, 0, 3 * sizeof(unsigned long long), c->stream);
    if (e == hipSuccess) e = pt::launch_selftest_math(c->stream, d);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess) e = hipMemcpy(mismatches_out, d, 3 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(PT_ERR_HIP, "pt_selftest_math: %s", hipGetErrorString(e));
    return PT_OK;
}

int pt_device_kat(pt_ctx *c, int op, const float *in, int n_in, float *out, int n_out)
{
    if (!c || !in || !out || n_in < 1 || n_out < 1 || n_in > 4096 || n_out > 4096)
        return fail(PT_ERR_INVALID, "pt_device_kat: bad arguments");
    static const int need_in[] = {0, 1, 1, 5, 39, 5, 16, 17, 17, 20, 7, 6, 8, 11, 4, 18, 1, 17, 11};
    if (op < 1 || op > 18 || n_in < need_in[op]) return fail(PT_ERR_INVALID, "pt_device_kat: op %d needs %d inputs", op, op >= 1 && op <= 18 ? need_in[op] : 0);
    HIP_TRY(hipSetDevice(c->device));
    float *d = nullptr;
    HIP_TRY(hipMalloc((void **)&d, (size_t)(n_in + n_out) * sizeof(float)));
    hipError_t e = hipMemcpy(d, in, (size_t)n_in * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemset(d + n_in, 0, (size_t)n_out * sizeof(float));
    if (e == hipSuccess) e = pt::launch_device_kat(c->stream, op, d, d + n_in, n_out);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess) e = hipMemcpy(out, d + n_in, (size_t)n_out * sizeof(float), hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(PT_ERR_HIP, "pt_device_kat: %s", hipGetErrorString(e));
    return PT_OK;
}

int pt_send_image_to_pbo(pt_ctx *c, pt_uchar4 *device_pbo)
{
    if (!c || !device_pbo) return fail(PT_ERR_INVALID, "pt_send_image_to_pbo: NULL argument");
    int rc = configure(c);
    if (rc != PT_OK) return rc;
    HIP_TRY(pt::launch_send_image_to_pbo(c->stream, device_pbo, image_ptr(c), c->kp.npix));
    return PT_OK;
}

int pt_record_event(pt_ctx *c, void *hip_event)
{
    if (!c || !hip_event) return fail(PT_ERR_INVALID, "pt_record_event: NULL argument");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipEventRecord((hipEvent_t)hip_event, c->stream));
    return PT_OK;
}

int pt_synchronize(pt_ctx *c)
{
    if (!c) return fail(PT_ERR_INVALID, "pt_synchronize: NULL context");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(sync_all_streams(c));
    HIP_TRY(hipGetLastError());
    return PT_OK;
}

int pt_render_iteration(pt_ctx *c, pt_uchar4 *pbo, float *host_image, int iteration)
{
    if (!c) return fail(PT_ERR_INVALID, "pt_render_iteration: NULL context");
    int rc = configure(c);
    if (rc != PT_OK) return rc;
    // The running mean is stateless given (image, iteration) (ref: src/raytraceKernel.cu:120,154): a context
    // that has not rendered iterations 1..iteration-1 itself takes them from the caller's buffer.
    if (iteration > 1 && !c->image_valid && host_image) {
        rc = pt_upload_image(c, host_image);
        if (rc != PT_OK) return rc;
    }
    rc = pt_render(c, iteration, 1);
    if (rc != PT_OK) return rc;
    if (pbo) { rc = pt_send_image_to_pbo(c, pbo); if (rc != PT_OK) return rc; }
    if (host_image) return pt_download_image(c, host_image);
    return pt_synchronize(c);
}

int pt_get_stats(pt_ctx *c, pt_stats *out)
{
    if (!c || !out) return fail(PT_ERR_INVALID, "pt_get_stats: NULL argument");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(sync_all_streams(c));
    int rc = fold_timers(c);
    if (rc != PT_OK) return rc;
    pt::IterState h;
    HIP_TRY(hipMemcpy(&h, c->d_state, sizeof h, hipMemcpyDeviceToHost));
    memset(out, 0, sizeof *out);
    out->iterations = h.iterations;
    for (int b = 0; b < PT_MAX_DEPTH; ++b) { out->live_in[b] = h.live_in[b]; out->ray_bounces += h.live_in[b]; }
    for (int sq = 1; sq < PT_MAX_SEQUENCES; ++sq) {         // the other launch sequences' counts (diagnostics: sequence 0's)
        pt::IterState h2;
        HIP_TRY(hipMemcpy(&h2, c->d_state + sq, sizeof h2, hipMemcpyDeviceToHost));
        out->iterations += h2.iterations;
        for (int b = 0; b < PT_MAX_DEPTH; ++b) { out->live_in[b] += h2.live_in[b]; out->ray_bounces += h2.live_in[b]; }
        h.shadow_rays += h2.shadow_rays;
    }
    out->gpu_ms = c->gpu_ms;
    if (getenv("PT_DEBUG_CLOCK")) fprintf(stderr, "[ptamd] cull: tested %llu skipped %llu (wave x primitive)\n", h.clk[2], h.clk[3]);
    if (getenv("PT_DEBUG_CLOCK") && h.clk[1])
        fprintf(stderr, "[ptamd] bounce-1 workgroup 0: %llu shader clocks in %llu x 10 ns -> %.0f MHz\n", h.clk[0], h.clk[1],
                (double)h.clk[0] / (double)h.clk[1] * 100.0);
    if (getenv("PT_DEBUG_SPAN") && atoi(getenv("PT_DEBUG_SPAN")) == 2) {     // -DPT_DEBUG_SPAN=2 builds
        fprintf(stderr, "[ptamd] bounce-1 workgroup lifetime sums by blockIdx %% 8 (10 ns ticks):");
        for (int x = 0; x < 8; ++x) fprintf(stderr, " %llu", h.dbg[x]);
        fprintf(stderr, "\n");
    } else if (getenv("PT_DEBUG_SPAN") && h.dbg[1])                          // -DPT_DEBUG_SPAN=1 builds
        fprintf(stderr, "[ptamd] bounce-1 workgroups: %llu lifetimes, mean %.1f us, shortest %.1f us, longest %.1f us; first start to last end over all launches %.1f us\n",
                h.dbg[1], (double)h.dbg[0] / (double)h.dbg[1] * 0.01, (double)(~h.dbg[3]) * 0.01, (double)h.dbg[2] * 0.01,
                (double)(h.dbg[5] - ~h.dbg[4]) * 0.01);
    out->bounce_launches = c->bounce_launches;
    out->shadow_rays = h.shadow_rays;
    for (pt_ctx *ch : c->slice_ctx) {                    // motion blur: the slices did the work
        pt_stats cs;
        rc = pt_get_stats(ch, &cs);
        if (rc != PT_OK) return rc;
        out->iterations += cs.iterations;
        out->ray_bounces += cs.ray_bounces;
        for (int b = 0; b < PT_MAX_DEPTH; ++b) out->live_in[b] += cs.live_in[b];
        out->gpu_ms += cs.gpu_ms;
        out->bounce_launches += cs.bounce_launches;
        out->shadow_rays += cs.shadow_rays;
    }
    if (getenv("PT_DEBUG_PHASE") && h.dbg[3] && h.dbg[7])
        fprintf(stderr, "[ptamd] shader clocks per chunk (wave latency): bounce 0: load/gen %.0f, nearest hit %.0f, shade+write %.0f; later: %.0f, %.0f, %.0f\n",
                (double)h.dbg[0] / h.dbg[3], (double)h.dbg[1] / h.dbg[3], (double)h.dbg[2] / h.dbg[3],
                (double)h.dbg[4] / h.dbg[7], (double)h.dbg[5] / h.dbg[7], (double)h.dbg[6] / h.dbg[7]);
    if (getenv("PT_DEBUG_BOUNDS") && h.dbg[0])
        fprintf(stderr, "[ptamd] BOUNDS violation: code %llu value %llu limit %llu block %llu thread %llu\n", h.dbg[0], h.dbg[1], h.dbg[2], h.dbg[4], h.dbg[5]);
    if (getenv("PT_DEBUG_PHASE2") && h.dbg[7])
        fprintf(stderr, "[ptamd] later bounces, shader clocks per chunk: load %.0f | nearest hit %.0f (pre-test + full batches %.0f, last batches %.0f, "
                        "result %.0f) | shade + radiance write %.0f | compaction + pool write %.0f\n",
                (double)h.dbg[0] / h.dbg[7], (double)h.dbg[3] / h.dbg[7], (double)h.dbg[1] / h.dbg[7], (double)h.dbg[2] / h.dbg[7],
                ((double)h.dbg[3] - (double)h.dbg[1] - (double)h.dbg[2]) / h.dbg[7], (double)h.dbg[4] / h.dbg[7], (double)h.dbg[5] / h.dbg[7]);
    if (getenv("PT_DEBUG_PHASE2") && h.lane_budget[13]) {
        // the lane budget of the later bounces (pair path): where the wave's clocks go and how many of the 64 lanes had work there
        const unsigned long long *B = h.lane_budget;
        const double rounds = (double)B[13];
        struct { const char *name; double clk, lane_clk; } ph[7] = {
            {"load / refill", (double)B[0], (double)B[1]}, {"pre-test loop", (double)B[2], (double)B[3]}, {"full batches", (double)B[4], (double)B[4] * 64.0},
            {"last batches", (double)B[5], (double)B[6]}, {"result", (double)B[7], (double)B[8]}, {"shade + radiance write", (double)B[9], (double)B[10]},
            {"compaction / hand-over", (double)B[11], (double)B[12]}};
        double tot = 0.0, tot_lane = 0.0;
        for (auto &x : ph) { tot += x.clk; tot_lane += x.lane_clk; }
        fprintf(stderr, "[ptamd] lane budget, later bounces: %.0f trips, %.1f rays per trip, %.0f clocks per trip; lanes with work, clock-weighted: %.1f of 64\n",
                rounds, (double)B[14] / rounds, tot / rounds, tot_lane / tot);
        fprintf(stderr, "[ptamd]   %-26s %10s %8s %12s %14s\n", "phase", "clocks/trip", "share", "busy lanes", "idle share");
        for (auto &x : ph)
            fprintf(stderr, "[ptamd]   %-26s %10.0f %7.1f%% %12.1f %13.1f%%\n", x.name, x.clk / rounds, 100.0 * x.clk / tot, x.clk > 0 ? x.lane_clk / x.clk : 0.0,
                    100.0 * (x.clk * 64.0 - x.lane_clk) / (tot * 64.0));
    }
    if (getenv("PT_DEBUG_PAIR") && h.dbg[4])
        fprintf(stderr, "[ptamd] pair queue: per ray %.2f sphere + %.2f box pairs; candidates that hit: %.2f + %.2f per ray; batches per wave round %.2f\n",
                (double)h.dbg[0] / h.dbg[4], (double)h.dbg[1] / h.dbg[4], (double)h.dbg[2] / h.dbg[4], (double)h.dbg[3] / h.dbg[4], (double)h.dbg[5] / h.dbg[6]);
    if (getenv("PT_DEBUG_CLOCK") && h.dbg[2])
        fprintf(stderr, "[ptamd] hierarchy walk: %.1f nodes, %.2f leaves per ray; per wave (longest lane): %.1f nodes, %.2f leaves, %.2f rounds\n",
                (double)h.dbg[0] / (double)h.dbg[2], (double)h.dbg[1] / (double)h.dbg[2], (double)h.dbg[3] / (double)h.dbg[6],
                (double)h.dbg[4] / (double)h.dbg[6], (double)h.dbg[5] / (double)h.dbg[6]);
    if (getenv("PT_DEBUG_W4") && h.dbg[2] && h.dbg[3] && h.dbg[6])
        fprintf(stderr, "[ptamd] batched walk: %.2f entries per ray, %.1f steps per wave round, %.1f entries per step, %.1f pairs per round\n",
                (double)h.dbg[0] / (double)h.dbg[2], (double)h.dbg[3] / (double)h.dbg[6], (double)h.dbg[0] / (double)h.dbg[3],
                (double)h.dbg[7] / (double)h.dbg[6]);
    if (getenv("PT_DEBUG_CLOCK") && h.dbg[7]) fprintf(stderr, "[ptamd] pairs per wave round: %.1f\n", (double)h.dbg[7] / (double)h.dbg[6]);
    return PT_OK;
}

int pt_reset_stats(pt_ctx *c)
{
    if (!c) return fail(PT_ERR_INVALID, "pt_reset_stats: NULL context");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(sync_all_streams(c));
    int rc = fold_timers(c);
    if (rc != PT_OK) return rc;
    // (everything behind the header: the header holds the batch serial number, which outlives the statistics)
    for (int sq = 0; sq < PT_MAX_SEQUENCES; ++sq)
        HIP_TRY(hipMemset((unsigned char *)(c->d_state + sq) + offsetof(pt::IterState, counts), 0, sizeof(pt::IterState) - offsetof(pt::IterState, counts)));
    c->gpu_ms = 0.0;
    c->bounce_launches = 0;
    for (pt_ctx *ch : c->slice_ctx) { rc = pt_reset_stats(ch); if (rc != PT_OK) return rc; }
    return PT_OK;
}

}  // extern "C"
