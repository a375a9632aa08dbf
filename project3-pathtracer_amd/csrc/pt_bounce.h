// pt_bounce.h -- the per-bounce kernel k_bounce<WG, FIRST, GEOM, COMPACT, FEAT> and the nearest-hit machinery behind it
// (raycastFromCameraKernel fused into bounce 0, one bounce of raytraceRay, stream compaction: see pt_kernels.hip's header).
// Templates only: each geometry path is instantiated in its own translation unit (pt_bounce_g<N>.hip) so that the eight
// of them compile in parallel; pt_kernels.hip picks the instance at launch time through bounce_kernel_g<N>().
#pragma once
#include <stdlib.h>

#include "pt_internal.h"

namespace pt {
using namespace ptd;

static constexpr uint32_t DEAD = 0xFFFFFFFFu;
// diagnostic builds: make OUT=../lib_dbg EXTRA_HIPFLAGS=-DPT_DEBUG_BVH=1 (the default library carries none of this)
#ifndef PT_DEBUG_PAIR
#define PT_DEBUG_PAIR 0
#endif
#ifndef PT_DEBUG_PHASE
#define PT_DEBUG_PHASE 0
#endif
#ifndef PT_DEBUG_BVH
#define PT_DEBUG_BVH 0
#endif
#ifndef PT_DEBUG_SPAN
#define PT_DEBUG_SPAN 0
#endif
static constexpr bool DEBUG_SPAN = PT_DEBUG_SPAN != 0;       // workgroup lifetimes of the bounce-1 launches into IterState::dbg
#ifndef PT_DEBUG_BOUNDS
#define PT_DEBUG_BOUNDS 0
#endif
static constexpr bool DEBUG_BOUNDS = PT_DEBUG_BOUNDS != 0;   // index checks before global accesses; first violation -> IterState::dbg
static constexpr bool DEBUG_CULL = false;
static constexpr bool DEBUG_PAIR = PT_DEBUG_PAIR != 0;
static constexpr bool DEBUG_PHASE = PT_DEBUG_PHASE == 1;     // per-wave shader-clock stamps between the phases of a chunk -> IterState::dbg
static constexpr bool DEBUG_PHASE2 = PT_DEBUG_PHASE == 2;    // finer split of the later bounces (pair-queue path): load, pre-test loop
                                                             // + full batches, last batches, result, RNG, lobe + radiance write, compaction
// the lane budget (clocks and busy lanes per phase of a trip, pair path): -DPT_DEBUG_PHASE=2 books the later bounces, =3 the camera launch
template <bool FIRST> __device__ __forceinline__ constexpr bool LANE_BUDGET() { return FIRST ? PT_DEBUG_PHASE == 3 : PT_DEBUG_PHASE == 2; }
static constexpr bool DEBUG_BVH = PT_DEBUG_BVH != 0;         // count node / leaf visits of the hierarchy walk into IterState::dbg
static constexpr int MAXSLOT = 16;                // iterations in flight per launch sequence (pt_internal.h PT_MAX_BATCH)
static constexpr uint32_t SLOT_SHIFT = 27;        // pixel word = tile-local pixel | slot << 27 | NEE mark << 31
static constexpr uint32_t PIX_MASK = (1u << SLOT_SHIFT) - 1u;

typedef const __attribute__((address_space(4))) uint32_t *const_u32_ptr;
typedef float v4f __attribute__((ext_vector_type(4)));      // operand types of the non-temporal load / store builtins
typedef float v2f __attribute__((ext_vector_type(2)));
// streaming (`nt`) loads / stores of data that is written once and read once (-DPT_NO_NT: plain ones, for A/B runs)
template <class V> __device__ __forceinline__ void nt_store(V v, V *p)
{
#if defined(PT_NO_NT)
    *p = v;
#else
    __builtin_nontemporal_store(v, p);
#endif
}
template <class V> __device__ __forceinline__ V nt_load(const V *p)
{
#if defined(PT_NO_NT)
    return *p;
#else
    return __builtin_nontemporal_load(p);
#endif
}

// diagnostic builds (-DPT_DEBUG_BOUNDS=1, PT_DEBUG_BOUNDS=1 at run time prints it): report the first out-of-range index
// (code, value, limit) in IterState::dbg and let the caller make the access harmless
__device__ __forceinline__ bool dbgInRange(const KParams &p, int code, unsigned long long v, unsigned long long lim)
{
    if (!DEBUG_BOUNDS || v < lim) return true;
    if (atomicCAS(&p.st->dbg[0], 0ull, (unsigned long long)code) == 0ull) { p.st->dbg[1] = v; p.st->dbg[2] = lim; p.st->dbg[4] = blockIdx.x; p.st->dbg[5] = threadIdx.x; }
    return false;
}

// Wave-uniform primitive fetch through the constant address space: hipcc turns this into s_load_dwordx16 x2
// and keeps the record in SGPRs.
__device__ __forceinline__ Prim load_prim_scalar(const Prim *prims, int g)
{
    Prim r;
    const_u32_ptr q = (const_u32_ptr)(uintptr_t)(prims + g);
    uint32_t *dst = reinterpret_cast<uint32_t *>(&r);
#pragma unroll
    for (int k = 0; k < 32; ++k) dst[k] = q[k];
    return r;
}

// LDS copy of a primitive record: 144-byte stride.  A lane of a pair batch gathers the rows of ITS primitive with
// ds_read_b128 (banks = dword address mod 64, 16 lanes per group): at the records' own 128-byte stride every primitive
// of the same parity sits on the same banks (58 % of the LDS cycles of round 1's kernel were bank conflicts); 36 dwords
// walk the banks in steps of 4 with period 16, so up to 16 different primitives are conflict-free.  The 8-entry
// face-normal table behind the records is padded the same way (9 float4 per primitive).
struct PrimPad : Prim { uint32_t pad_lds[4]; };
static_assert(sizeof(PrimPad) == 144, "PrimPad must be 144 B");
template <class PR> __device__ __forceinline__ constexpr uint32_t faceStride() { return (uint32_t)(sizeof(PR) / 16); }

struct Hit {
    f3 p, n;
    uint32_t material;
    uint32_t prim;      // index of the primitive hit (used by shadow rays)
    float t;            // world-space distance to the hit
    bool any;
    uint32_t self_ok;   // resident paths: may the next bounce skip the primitive hit?  (Prim::self_r2, and for a sphere the hit point checked against it)
    unsigned long long dbg0, dbg1;   // DEBUG_PHASE2 builds: shader clocks of the pre-test loop / the last batches (else unused)
    unsigned long long dbg2, dbg3;   //   ... of the full batches inside the loop; lane-clocks of the last batches (clocks x busy lanes)
};

// GEOM selects how the primitive list reaches the lanes:
//   0  direct, scalar: records through s_load into SGPRs; hit work inside the wave-uniform primitive loop
//   1  direct, LDS:    records staged in LDS, broadcast reads; same loop
//   2  hit queue:      the uniform loop only finds candidates (object-space t); every (ray, primitive, t)
//                      candidate is appended to a wave-private LDS queue and the expensive hit work (world
//                      point, distance) runs on full 64-candidate batches with every lane busy, whichever
//                      ray or primitive a candidate belongs to.  Results flow back to the owning lane through
//                      a 64-bit LDS atomic min on (distance bits, primitive index).
//   3  hierarchy walk: for large primitive lists.  Every lane walks a bounding-box hierarchy of the primitives
//                      (LDS-resident, depth-first with skip links, no stack) and runs the exact test only on the
//                      leaves its ray can reach; per-lane primitive records are gathered from the LDS copy.  The
//                      boxes are padded and only ever cull, so the nearest hit (ties -> lowest index) is unchanged.
//   4  pair queue:     the uniform loop only runs a cheap per-lane test of the ray against the primitive's padded world
//                      box and queues the (ray, primitive) pairs that pass -- spheres and boxes apart; the candidate test
//                      itself AND the hit work then run on full 64-pair batches, each lane on its own pair (ray from
//                      the owner's LDS slot, primitive record gathered from the LDS copy), results through the same
//                      64-bit LDS atomic min.  The box test only ever drops pairs the exact test would miss.
//   5  walk + pairs:   large primitive lists.  The scene-spanning primitives go through the pair queue first, which gives
//                      every ray an upper bound on its hit distance; each lane then walks the hierarchy with box tests
//                      only, pruned by that bound, and queues the leaves it reaches as (ray, primitive) pairs; the exact
//                      tests run on full batches as in 4.
//   6  batched 4-wide walk + pairs: large primitive lists, no per-lane loop at all.  The hierarchy is collapsed to 4-wide
//                      nodes; (ray, node) entries live on a wave-private LDS stack and the wave pops up to 64 of them at a
//                      time, each lane testing ITS entry's ray against the four child boxes of ITS entry's node (near / far
//                      planes picked by address from the ray's direction signs: 6 fma + max3 + min3 per box, no min/max
//                      per slab), and pushes the children that pass -- inner nodes back on the stack, leaves into the pair
//                      queues of 4.  Lanes stay busy whatever the length of a single ray's walk: the per-lane walk of 5
//                      ran 78 loop trips per wave for 28.6 nodes per ray.
//   7  the same with the 4-wide nodes read through L1/L2 instead of an LDS copy (hierarchies too large for the LDS,
//                      e.g. triangle meshes of a few thousand faces)
// Triangle records (type 3, the flattened MESH geoms) are understood by 0, 6 and 7.
enum { GEOM_SCALAR = 0, GEOM_LDS = 1, GEOM_QUEUE = 2, GEOM_BVH = 3, GEOM_PAIR = 4, GEOM_WALK_PAIR = 5, GEOM_WALK4 = 6,
       GEOM_WALK4G = 7 };

// Conservative cull for large primitive lists: true when NO lane of the wave can hit the primitive, judged by a
// padded world-space bounding sphere (centre = transform*(0,0,0,1), radius^2 in the record).  It only ever skips
// primitives whose exact test would miss for every lane, so results are unchanged; the wave-uniform ballot makes
// the skip a scalar branch.
__device__ __forceinline__ bool waveMissesBound(const Prim &P, f3 o, f3 d)
{
    const f3 oc = o - mk(P.cx, P.cy, P.cz);
    const float b = dot(oc, d);
    const float oc2 = dot(oc, oc);
    const float perp2 = oc2 - b * b;                       // squared distance of the centre from the ray's line
    // origin outside and leaving, or the line passes outside the sphere; oc2*1e-5 covers the fp32 cancellation in
    // perp2 and |d| = 1 +- 1e-6 (the radius itself is already padded by 2 % + 1e-3 on the host)
    const bool reject = (oc2 > P.bound_r2 && b > 0.0f) || (perp2 > P.bound_r2 + oc2 * 1e-5f);
    return __ballot(!reject) == 0ull;
}

// wave-uniform fetch of inverseTransform*(eye,1) of primitive g (camera rays only)
__device__ __forceinline__ f3 load_ro_eye(const float *ro_eye, int g)
{
    const_u32_ptr q = (const_u32_ptr)(uintptr_t)(ro_eye + 4 * g);
    return mk(__uint_as_float(q[0]), __uint_as_float(q[1]), __uint_as_float(q[2]));
}

// Camera rays only (they share the eye as origin): true when no ray of the wave can reach primitive g, judged by
// its padded world box relative to the eye (host-side, KParams::box_eye).  It only ever skips primitives the exact
// test would miss for every lane.  inv = approximate, finite reciprocal of the ray direction.
__device__ __forceinline__ bool waveMissesBoxFromEye(const float *box_eye, int g, f3 inv, bool valid)
{
    const_u32_ptr q = (const_u32_ptr)(uintptr_t)(box_eye + 8 * g);
    const float x0 = __uint_as_float(q[0]) * inv.x, x1 = __uint_as_float(q[4]) * inv.x;
    const float y0 = __uint_as_float(q[1]) * inv.y, y1 = __uint_as_float(q[5]) * inv.y;
    const float z0 = __uint_as_float(q[2]) * inv.z, z1 = __uint_as_float(q[6]) * inv.z;
    const float tn = fmaxf(fmaxf(fminf(x0, x1), fminf(y0, y1)), fmaxf(fminf(z0, z1), 0.0f));
    const float tf = fminf(fminf(fmaxf(x0, x1), fmaxf(y0, y1)), fmaxf(z0, z1));
    return (__builtin_amdgcn_fcmpf(tn, tf, FCMP_OLE) & __ballot(valid)) == 0ull;
}
// finite reciprocal of a direction (culling only): |component| kept above 1e-20 with its sign -- v_max + v_bfi instead of
// two compare / select pairs per component (a zero component takes the sign of its zero: either sign brackets the slab)
__device__ __forceinline__ f3 approxInverse(f3 d)
{
    return mk(__builtin_amdgcn_rcpf(__builtin_copysignf(fmaxf(fabsf(d.x), 1e-20f), d.x)),
              __builtin_amdgcn_rcpf(__builtin_copysignf(fmaxf(fabsf(d.y), 1e-20f), d.y)),
              __builtin_amdgcn_rcpf(__builtin_copysignf(fmaxf(fabsf(d.z), 1e-20f), d.z)));
}

template <int GEOM, bool FIRST, class PR>
__device__ __forceinline__ Hit nearestHitDirect(const KParams &p, const PR *s_prims, f3 o, f3 d)
{
    Hit h;
    h.any = false;
    h.material = 0;
    h.prim = 0;
    h.p = mk(0, 0, 0);
    h.n = mk(0, 0, 0);
    float best_t = 0.0f;
    for (int g = 0; g < p.nG; ++g) {
        f3 ip, in;
        float t;
        uint32_t mat;
        if (GEOM == GEOM_LDS) {
            const Prim &P = s_prims[g];
            if (p.cull && waveMissesBound(P, o, d)) continue;
            t = intersectPrim<FIRST>(P, o, d, FIRST ? load_ro_eye(p.ro_eye, g) : o, ip, in);
            mat = P.material;
        } else {
            const Prim P = load_prim_scalar(p.prims, g);
            if (p.cull) {
                const bool skip = waveMissesBound(P, o, d);
                if (DEBUG_CULL && __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) == 0) atomicAdd(&p.st->clk[skip ? 3 : 2], 1ull);
                if (skip) continue;
            }
            t = intersectPrim<FIRST>(P, o, d, FIRST ? load_ro_eye(p.ro_eye, g) : o, ip, in);
            mat = P.material;
        }
        if (t > 0 && (!h.any || t < best_t)) {      // smallest t > 0, ties keep the lowest index
            h.any = true;
            best_t = t;
            h.p = ip;
            h.n = in;
            h.material = mat;
            h.prim = (uint32_t)g;
        }
    }
    h.t = best_t;
    return h;
}

template <bool FIRST>
__device__ __forceinline__ Hit nearestHitBvh(const KParams &p, const Prim *s_prims, const float4 *s_nodes, f3 o, f3 d)
{
    Hit h;
    h.any = false;
    h.material = 0;
    h.p = mk(0, 0, 0);
    h.n = mk(0, 0, 0);
    float best_t = 0.0f;
    uint32_t best_g = 0xFFFFFFFFu;
    // reciprocal direction for the box slabs: culling only, so an approximate, finite value is enough
    const float ix = __builtin_amdgcn_rcpf(fabsf(d.x) > 1e-20f ? d.x : (d.x < 0 ? -1e-20f : 1e-20f));
    const float iy = __builtin_amdgcn_rcpf(fabsf(d.y) > 1e-20f ? d.y : (d.y < 0 ? -1e-20f : 1e-20f));
    const float iz = __builtin_amdgcn_rcpf(fabsf(d.z) > 1e-20f ? d.z : (d.z < 0 ? -1e-20f : 1e-20f));
    // primitives that span most of the scene (walls, big lights) would bloat every box above them: they are kept
    // out of the hierarchy and tested by every ray, wave-uniformly through the scalar unit
    // the normal is only needed for the final winner: candidates keep (distance, primitive, point, box face)
    uint32_t best_face = 0u;
    for (int k = 0; k < p.nbig; ++k) {
        const int g = p.big[k];
        const Prim P = load_prim_scalar(p.prims, g);
        f3 ro = FIRST ? load_ro_eye(p.ro_eye, g) : o, rd;
        float tc;
        uint32_t face;
        if (!candidateT<FIRST>(P.type, P.inv, o, d, ro, rd, tc, face)) continue;
        f3 ip;
        const float t = hitPoint(P.fwd, o, ro, rd, tc, ip);
        if (t > 0 && (!h.any || t < best_t || (t == best_t && (uint32_t)g < best_g))) {
            h.any = true;
            best_t = t;
            best_g = (uint32_t)g;
            best_face = face;
            h.p = ip;
        }
    }
    uint32_t i = 0;
    const uint32_t nn = (uint32_t)p.nnodes;
    uint32_t dbg_nodes = 0, dbg_leaves = 0, dbg_outer = 0, dbg_inner_wave = 0;
    for (;;) {
        // walk to the next leaf this ray can reach (box tests only), so that the lanes of the wave then run the
        // long exact test together instead of one lane at a time
        int prim = -1;
        if (DEBUG_BVH) dbg_outer++;
        while (i < nn) {
            if (DEBUG_BVH) dbg_nodes++;
            const float4 a = s_nodes[2 * i], b = s_nodes[2 * i + 1];
            const float x0 = (a.x - o.x) * ix, x1 = (b.x - o.x) * ix;
            const float y0 = (a.y - o.y) * iy, y1 = (b.y - o.y) * iy;
            const float z0 = (a.z - o.z) * iz, z1 = (b.z - o.z) * iz;
            const float tn = fmaxf(fmaxf(fminf(x0, x1), fminf(y0, y1)), fmaxf(fminf(z0, z1), 0.0f));
            const float tf = fminf(fminf(fmaxf(x0, x1), fmaxf(y0, y1)), fmaxf(z0, z1));
            // miss, or the box starts beyond the current best hit (with slack for |d| != 1 and rounding)
            if (!(tn <= tf) || (h.any && tn * 0.999f - 1e-3f > best_t)) {
                i = __float_as_uint(a.w);
                continue;
            }
            i = i + 1u;
            prim = (int)__float_as_uint(b.w);
            if (prim >= 0) break;
        }
        if (prim < 0) break;
        prim &= 0x3FFFFFFF;                                  // bit 30 = primitive type (used by the walk + pairs path)
        if (DEBUG_BVH) dbg_leaves++;
        const Prim &P = s_prims[prim];                     // per-lane gather of the record (L1/L2)
        f3 ro = o, rd;
        float tc;
        uint32_t face;
        if (!candidateT<false>(P.type, P.inv, o, d, ro, rd, tc, face)) continue;
        f3 ip;
        const float t = hitPoint(P.fwd, o, ro, rd, tc, ip);
        if (t > 0 && (!h.any || t < best_t || (t == best_t && (uint32_t)prim < best_g))) {
            h.any = true;
            best_t = t;
            best_g = (uint32_t)prim;
            best_face = face;
            h.p = ip;
        }
    }
    if (h.any) {
        const Prim &P = s_prims[best_g];
        h.material = P.material;
        if (P.type == 0u) h.n = sphereNormal(h.p, mk(P.cx, P.cy, P.cz));
        else {
            const float4 fn = reinterpret_cast<const float4 *>(p.face_n)[best_g * 8u + best_face];
            h.n = mk(fn.x, fn.y, fn.z);
        }
    }
    if (DEBUG_BVH) {
        // per ray: nodes, leaves; per wave: the longest lane (what the wave pays)
        uint32_t mn = dbg_nodes, ml = dbg_leaves, mo = dbg_outer;
        for (int s = 32; s > 0; s >>= 1) {
            mn = max(mn, (uint32_t)__shfl_xor((int)mn, s));
            ml = max(ml, (uint32_t)__shfl_xor((int)ml, s));
            mo = max(mo, (uint32_t)__shfl_xor((int)mo, s));
        }
        atomicAdd(&p.st->dbg[0], (unsigned long long)dbg_nodes);
        atomicAdd(&p.st->dbg[1], (unsigned long long)dbg_leaves);
        atomicAdd(&p.st->dbg[2], 1ull);
        if (__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) == 0) {
            atomicAdd(&p.st->dbg[3], (unsigned long long)mn);
            atomicAdd(&p.st->dbg[4], (unsigned long long)ml);
            atomicAdd(&p.st->dbg[5], (unsigned long long)mo);
            atomicAdd(&p.st->dbg[6], 1ull);
        }
        (void)dbg_inner_wave;
    }
    h.prim = best_g;
    h.t = best_t;
    return h;
}

// wave-private LDS scratch of the hit queue
struct WaveQueue {
    float4 *rec;                  // [QCAP][2]: (ro.xyz, rd.x) (rd.y, rd.z, t, meta)
    unsigned long long *key;      // [64] per owner lane: min over candidates of (distance bits << 32 | prim << 8)
    float4 *best;                 // [64] per owner lane: (hit point xyz, meta) of the current minimum
    float4 *org;                  // [64] per owner lane: ray origin
};
static constexpr uint32_t QCAP = 128;                                  // >= 63 pending + 64 appended
static constexpr uint32_t WAVE_QUEUE_BYTES = QCAP * 32 + 64 * 8 + 64 * 16 + 64 * 16;
static constexpr unsigned long long KEY_NONE = ~0ull;

__device__ __forceinline__ void wave_lds_fence() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); }

// Must be entered by all 64 lanes of the wave (lanes without a ray pass valid = false): lanes are consumers
// of queued candidates independently of their own ray.
template <bool FIRST, class PR>
__device__ __forceinline__ Hit nearestHitQueued(const KParams &p, const PR *s_prims, const WaveQueue q, f3 o, f3 d,
                                                bool valid, uint32_t lane)
{
    q.key[lane] = KEY_NONE;
    q.org[lane] = make_float4(o.x, o.y, o.z, 0.0f);
    wave_lds_fence();
    uint32_t qhead = 0, qtail = 0;                       // wave-uniform
    const bool eye_cull = FIRST && p.eye_cull != 0;
    const f3 dinv = eye_cull ? approxInverse(d) : mk(0, 0, 0);
    for (int g = 0; g <= p.nG; ++g) {
        if (g < p.nG && !(eye_cull && waveMissesBoxFromEye(p.box_eye, g, dinv, valid))) {
            // candidate test: wave-uniform primitive (type + inverse transform through the scalar unit)
            const_u32_ptr hp = (const_u32_ptr)(uintptr_t)(p.prims + g);
            const uint32_t type = hp[0];
            float inv[12];
#pragma unroll
            for (int k = 0; k < 12; ++k) inv[k] = __uint_as_float(hp[4 + k]);
            f3 ro = FIRST ? load_ro_eye(p.ro_eye, g) : mk(0, 0, 0), rd = mk(0, 0, 0);
            float t = 0.0f;
            uint32_t face = 0u;
            const bool cand = valid && candidateT<FIRST>(type, inv, o, d, ro, rd, t, face);
            const uint64_t mask = __ballot(cand);
            if (mask != 0ull) {
                if (cand) {
                    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32),
                                                                    __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
                    const uint32_t pos = (qtail + rank) & (QCAP - 1u);
                    const uint32_t meta = lane | ((uint32_t)g << 8) | (face << 28);
                    q.rec[2 * pos] = make_float4(ro.x, ro.y, ro.z, rd.x);
                    q.rec[2 * pos + 1] = make_float4(rd.y, rd.z, t, __uint_as_float(meta));
                }
                qtail += (uint32_t)__popcll(mask);
            }
        }
        const uint32_t pending = qtail - qhead;
        if (pending >= 64u || (g == p.nG && pending > 0u)) {
            // hit work on a batch of candidates: lane l takes candidate qhead + l, whoever owns it
            const uint32_t nb = pending < 64u ? pending : 64u;
            wave_lds_fence();
            unsigned long long mykey = KEY_NONE;
            uint32_t owner = 0u;
            float4 mine = make_float4(0, 0, 0, 0);
            if (lane < nb) {
                const uint32_t pos = (qhead + lane) & (QCAP - 1u);
                const float4 r0 = q.rec[2 * pos], r1 = q.rec[2 * pos + 1];
                const uint32_t meta = __float_as_uint(r1.w);
                owner = meta & 63u;
                const uint32_t prim = (meta >> 8) & 0xFFFFFu;
                const float4 *fw = reinterpret_cast<const float4 *>(s_prims[prim].fwd);
                const float4 f0 = fw[0], f1 = fw[1], f2 = fw[2];
                const float fwd[12] = {f0.x, f0.y, f0.z, f0.w, f1.x, f1.y, f1.z, f1.w, f2.x, f2.y, f2.z, f2.w};
                const float4 oo = q.org[owner];
                f3 real;
                const float dist = hitPoint(fwd, mk(oo.x, oo.y, oo.z), mk(r0.x, r0.y, r0.z), mk(r0.w, r1.x, r1.y), r1.z, real);
                if (dist > 0) {                              // same admission test as the direct path: t > 0
                    mykey = ((unsigned long long)__float_as_uint(dist) << 32) | (unsigned long long)(prim << 8);
                    mine = make_float4(real.x, real.y, real.z, r1.w);
                    __hip_atomic_fetch_min(&q.key[owner], mykey, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                }
            }
            wave_lds_fence();
            if (mykey != KEY_NONE && q.key[owner] == mykey) q.best[owner] = mine;   // unique writer: keys are unique
            qhead += nb;
            wave_lds_fence();
        }
    }
    Hit h;
    h.any = false;
    h.material = 0;
    h.prim = 0;
    h.t = 0.0f;
    h.p = mk(0, 0, 0);
    h.n = mk(0, 0, 0);
    const unsigned long long k = q.key[lane];
    if (valid && k != KEY_NONE) {
        const float4 b = q.best[lane];
        const uint32_t meta = __float_as_uint(b.w);
        const uint32_t prim = (meta >> 8) & 0xFFFFFu;
        const uint32_t face = meta >> 28;
        const Prim &P = s_prims[prim];                       // per-lane gather from the LDS copy
        h.any = true;
        h.prim = prim;
        h.t = __uint_as_float((uint32_t)(k >> 32));
        h.p = mk(b.x, b.y, b.z);
        h.material = P.material;
        if (P.type == 0u) {
            const float4 c = *reinterpret_cast<const float4 *>(&P.cx);
            h.n = sphereNormal(h.p, mk(c.x, c.y, c.z));
        } else {
            const float4 fn = reinterpret_cast<const float4 *>(s_prims + p.nG)[prim * faceStride<PR>() + face];   // face-normal table behind the records
            h.n = mk(fn.x, fn.y, fn.z);
        }
    }
    return h;
}

// ---------------------------------------------------------------------------------------------------------------
// FEAT_MOTION: a shutter time per ray
// ---------------------------------------------------------------------------------------------------------------
// Every path draws its time as the third number of its camera stream and sees the scene interpolated between the two
// knot states around it: segment k of the nknots - 1 between the knots and the fraction f inside it.  Nothing about a
// primitive is wave-uniform any more: each lane gathers the two knots' transform rows of the primitive, interpolates them
// entry-wise, a + (b - a) * f, and inverts the result (the oracle's scene_at_time).
struct MotionTime { uint32_t k = 0u; float f = 0.0f; };
__device__ __forceinline__ MotionTime motionTime(const KParams &p, float u_t)
{
    const int K = p.nknots - 1;
    const float tau = u_t * (float)K;
    int k = (int)tau;
    if (k > K - 1) k = K - 1;
    MotionTime mt;
    mt.k = (uint32_t)k;
    mt.f = tau - (float)k;
    return mt;
}
// rows 0..2 of the transform of primitive g at the lane's time -- the two knots' rows interpolated entry-wise -- and of its inverse,
// computed from them: adjugate over determinant for the 3x3 part, then -inverse * translation, in the oracle's operation order
// (o_affineInverse).  The object a ray meets is then exactly the interpolated transform's image of the unit shape.
__device__ __forceinline__ void motionRows(const KParams &p, uint32_t g, MotionTime mt, float *inv, float *fwd)
{
    const float4 *A = reinterpret_cast<const float4 *>(p.knots) + ((size_t)mt.k * (size_t)p.nG + g) * 3u;
    const float4 *B = A + (size_t)p.nG * 3u;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const float4 a = A[r], b = B[r];
        fwd[4 * r + 0] = a.x + (b.x - a.x) * mt.f;
        fwd[4 * r + 1] = a.y + (b.y - a.y) * mt.f;
        fwd[4 * r + 2] = a.z + (b.z - a.z) * mt.f;
        fwd[4 * r + 3] = a.w + (b.w - a.w) * mt.f;
    }
    const float a0 = fwd[0], a1 = fwd[1], a2 = fwd[2], a3 = fwd[3], b0 = fwd[4], b1 = fwd[5], b2 = fwd[6], b3 = fwd[7];
    const float c0 = fwd[8], c1 = fwd[9], c2 = fwd[10], c3 = fwd[11];
    const float k00 = b1 * c2 - b2 * c1, k01 = b2 * c0 - b0 * c2, k02 = b0 * c1 - b1 * c0;
    const float det = (a0 * k00 + a1 * k01) + a2 * k02;
    const float id = rcp_rn(det);
    inv[0] = k00 * id; inv[1] = (a2 * c1 - a1 * c2) * id; inv[2] = (a1 * b2 - a2 * b1) * id;
    inv[4] = k01 * id; inv[5] = (a0 * c2 - a2 * c0) * id; inv[6] = (a2 * b0 - a0 * b2) * id;
    inv[8] = k02 * id; inv[9] = (a1 * c0 - a0 * c1) * id; inv[10] = (a0 * b1 - a1 * b0) * id;
    inv[3] = -((inv[0] * a3 + inv[1] * b3) + inv[2] * c3);
    inv[7] = -((inv[4] * a3 + inv[5] * b3) + inv[6] * c3);
    inv[11] = -((inv[8] * a3 + inv[9] * b3) + inv[10] * c3);
}
__device__ __forceinline__ Hit nearestHitMotion(const KParams &p, f3 o, f3 d, MotionTime mt)
{
    Hit h;
    h.any = false;
    h.material = 0;
    h.prim = 0;
    h.p = mk(0, 0, 0);
    h.n = mk(0, 0, 0);
    float best_t = 0.0f;
    for (int g = 0; g < p.nG; ++g) {
        const_u32_ptr hp = (const_u32_ptr)(uintptr_t)(p.prims + g);
        const uint32_t type = hp[0], mat = hp[1];
        if (type > 1u) continue;
        Prim P;
        P.type = type;
        motionRows(p, (uint32_t)g, mt, P.inv, P.fwd);
        P.cx = P.fwd[3]; P.cy = P.fwd[7]; P.cz = P.fwd[11];      // transform * (0,0,0,1): the translation column, exactly
        f3 ip, in;
        const float t = intersectPrim<false>(P, o, d, o, ip, in);
        if (t > 0 && (!h.any || t < best_t)) {                    // smallest t > 0, ties keep the lowest index
            h.any = true;
            best_t = t;
            h.p = ip;
            h.n = in;
            h.material = mat;
            h.prim = (uint32_t)g;
        }
    }
    h.t = best_t;
    return h;
}

// ---------------------------------------------------------------------------------------------------------------
// GEOM_PAIR
// ---------------------------------------------------------------------------------------------------------------
struct PairQueue {
    uint32_t *q[2];               // [QCAP] pending pairs, lane | prim << 8: [0] spheres, [1] boxes
    uint32_t *tq;                 // [QCAP] pending (ray, triangle) pairs (batched walk only)
    unsigned long long *key;      // [64] per owner lane: min over hits of (distance bits << 32 | prim << 8)
    float4 *best;                 // [64] per owner lane: (hit point xyz, meta) of the current minimum
    float4 *org;                  // [64] per owner lane: the ray, (origin.xyz, direction.x) ...
    float2 *dir;                  // [64] ... (direction.y, direction.z)
    f3 po, pd;                    // org == nullptr (batched walk): the calling lane's OWN ray; a pair's lane fetches its owner's
                                  // through ds_bpermute instead of from an LDS copy (1.5 KB per wave less)
    uint2 *mt;                    // [64] per owner lane (FEAT_MOTION only): the ray's shutter segment and fraction (MotionTime)
    unsigned long long *dbg;
};
static constexpr uint32_t PAIR_QUEUE_BYTES = 2 * QCAP * 4 + 64 * 8 + 2 * 64 * 16 + 64 * 8;
// LDS bytes per primitive of the pre-test's box table: eight (near, far) entries, one per direction octant
static constexpr int PAIR_BOX_BYTES = 256;
static constexpr uint32_t PAIR_QUEUE_MOTION_BYTES = PAIR_QUEUE_BYTES + 64 * 8;      // + the owners' shutter times
static_assert(PAIR_QUEUE_BYTES <= WAVE_QUEUE_BYTES, "the pair queue lives in the hit queue's LDS region");

// one batch: lane l takes pair head + l of queue TYPE (0 sphere, 1 box), whoever owns it.  TYPE 2 = the last, mixed
// batch of a chunk: lanes [0, nb) take the sphere queue's leftovers, lanes [nb, nb + nb2) the box queue's.
// MOTION (FEAT_MOTION, a shutter time per ray): the pair's lane interpolates ITS primitive's rows at ITS owner's time from the
// knot states instead of reading the static record.
template <uint32_t TYPE, bool FIRST, class PR, bool MOTION = false>
__device__ __forceinline__ void pairBatch(const KParams &p, const PR *s_prims, const PairQueue &q, uint32_t head, uint32_t nb,
                                          uint32_t lane, uint32_t head2 = 0u, uint32_t nb2 = 0u)
{
    wave_lds_fence();
    unsigned long long mykey = KEY_NONE;
    uint32_t owner = 0u;
    float4 mine = make_float4(0, 0, 0, 0);
    const bool busy = lane < nb + nb2;
    // TYPE 3: a batch of (ray, triangle) pairs from the triangle queue
    const uint32_t e = !busy ? 0u
                       : (TYPE == 3u) ? q.tq[(head + lane) & (QCAP - 1u)]
                       : ((TYPE == 2u && lane >= nb) ? q.q[1][(head2 + lane - nb) & (QCAP - 1u)]
                                                     : q.q[(TYPE == 2u || TYPE == 3u) ? 0u : TYPE][(head + lane) & (QCAP - 1u)]);
    owner = e & 63u;
    f3 o = mk(0, 0, 0), d = mk(0, 0, 0);
    if (q.org == nullptr) {                                  // (known at compile time; every lane of the wave is here)
        const int oaddr = (int)(owner << 2);
        auto fetch = [&](float v) { return __int_as_float(__builtin_amdgcn_ds_bpermute(oaddr, __float_as_int(v))); };
        o = mk(fetch(q.po.x), fetch(q.po.y), fetch(q.po.z));
        d = mk(fetch(q.pd.x), fetch(q.pd.y), fetch(q.pd.z));
    }
    if (busy) {
        const uint32_t type = (TYPE == 2u) ? (lane < nb ? 0u : 1u) : TYPE;
        uint32_t prim = e >> 8;
        if (!dbgInRange(p, 10 + (int)TYPE, prim, (unsigned long long)p.nG)) prim = 0u;
        if (q.org != nullptr) {
            const float4 oo = q.org[owner];
            const float2 dd = q.dir[owner];
            o = mk(oo.x, oo.y, oo.z);
            d = mk(oo.w, dd.x, dd.y);
        }
        const float4 *iv = reinterpret_cast<const float4 *>(s_prims[prim].inv);
        float inv[12], mfwd[12];
        if (MOTION) {
            const uint2 tm = q.mt[owner];
            MotionTime mt;
            mt.k = tm.x;
            mt.f = __uint_as_float(tm.y);
            motionRows(p, prim, mt, inv, mfwd);
        } else {
            const float4 i0 = iv[0], i1 = iv[1], i2 = iv[2];
            inv[0] = i0.x; inv[1] = i0.y; inv[2] = i0.z; inv[3] = i0.w; inv[4] = i1.x; inv[5] = i1.y; inv[6] = i1.z; inv[7] = i1.w;
            inv[8] = i2.x; inv[9] = i2.y; inv[10] = i2.z; inv[11] = i2.w;
        }
        f3 ro = o, rd;
        if (FIRST) {                                         // camera rays: inverseTransform*(eye,1) comes from the host
            const float4 re = reinterpret_cast<const float4 *>(p.ro_eye)[prim];
            ro = mk(re.x, re.y, re.z);
        }
        float t;
        uint32_t face;
        const bool ch = candidateT<FIRST, TYPE == 3u>(type, inv, o, d, ro, rd, t, face);
        if (DEBUG_PAIR) { const uint64_t mm = __ballot(ch); if (__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) == 0) atomicAdd((unsigned long long *)q.dbg + 2 + (TYPE & 1u), (unsigned long long)__popcll(mm)); }
        if (ch) {
            f3 real;
            float dist;
            if (TYPE == 3u) dist = hitPointTriangle(o, ro, rd, t, real);
            else if (MOTION) dist = hitPoint(mfwd, o, ro, rd, t, real);
            else {
                const float4 f0 = iv[3], f1 = iv[4], f2 = iv[5];          // fwd rows follow the inverse rows
                const float fwd[12] = {f0.x, f0.y, f0.z, f0.w, f1.x, f1.y, f1.z, f1.w, f2.x, f2.y, f2.z, f2.w};
                dist = hitPoint(fwd, o, ro, rd, t, real);
            }
            if (dist > 0) {                                  // same admission test as the direct path: t > 0
                mykey = ((unsigned long long)__float_as_uint(dist) << 32) | (unsigned long long)(prim << 8);
                mine = make_float4(real.x, real.y, real.z, __uint_as_float((prim << 8) | (face << 28)));
                __hip_atomic_fetch_min(&q.key[owner], mykey, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            }
        }
    }
    wave_lds_fence();
    if (mykey != KEY_NONE && q.key[owner] == mykey) q.best[owner] = mine;   // unique writer: keys are unique
    wave_lds_fence();
}

// Must be entered by all 64 lanes of the wave (lanes without a ray pass valid = false).
// MOTION: s_boxes holds the primitives' boxes swept over the shutter interval (host), mymt = the calling lane's shutter time.
// SKIP (resident paths): `skip` = the primitive this lane's ray just left on its OUTSIDE (0xFFFFFFFF: none).  A ray that starts
// 0.0002 outside a convex primitive and moves away from it cannot meet it again -- the exact test of that pair always misses (for the
// primitives the host marks: Prim::self_r2, pt_context.hip) -- but its origin lies inside the primitive's padded box, so the pre-test
// would queue the pair every time: on the bundled scene, whose walls are tilted (ROTAT in radians) and fill the room with their
// boxes, HALF of all pairs were such self pairs.  The bounds-checking build runs the exact test on every skipped pair and reports a hit.
template <bool FIRST, class PR, bool MOTION = false, bool SKIP = false, bool SLAB = false>
__device__ __forceinline__ Hit nearestHitPairs(const KParams &p, const PR *s_prims, const float4 *s_boxes, const PairQueue q,
                                               f3 o, f3 d, bool valid, uint32_t lane, uint32_t primmask, MotionTime mymt = MotionTime(),
                                               uint32_t skip = 0xFFFFFFFFu)
{
    const unsigned long long ph_in = LANE_BUDGET<FIRST>() ? __builtin_amdgcn_s_memtime() : 0ull;
    unsigned long long fb_clk = 0ull, lb_lane_clk = 0ull;      // DEBUG_PHASE2: clocks inside full batches; lane-clocks of the last batches
    q.key[lane] = KEY_NONE;
    q.org[lane] = make_float4(o.x, o.y, o.z, d.x);
    q.dir[lane] = make_float2(d.y, d.z);
    if (MOTION) q.mt[lane] = make_uint2(mymt.k, __float_as_uint(mymt.f));
    uint32_t head[2] = {0u, 0u}, tail[2] = {0u, 0u};        // wave-uniform
    const uint64_t vmask = __ballot(valid);
    const f3 dinv = approxInverse(d);
    // slab distances as fma(plane, 1/d, -o/d): one instruction per plane; against (plane - o)/d this moves a plane by
    // less than 1.2e-6 |o|, far inside the boxes' padding
    const f3 oinv = FIRST ? mk(0, 0, 0) : mk(-(o.x * dinv.x), -(o.y * dinv.y), -(o.z * dinv.z));
    // the ray's direction octant as an offset (in float4) into a primitive's eight (near, far) entries
    const uint32_t octoff = ((__float_as_uint(dinv.x) >> 31) | ((__float_as_uint(dinv.y) >> 31) << 1) | ((__float_as_uint(dinv.z) >> 31) << 2)) * 2u;
    for (int g = 0; g < p.nG; ++g) {
        // camera rays: the host's table says which primitives the 64 pixels of this chunk can see at all (primmask, bit g;
        // all ones without a table).  The bounds-checking build runs the test anyway and reports a pair that passes.
        // (First thing in the loop: a culled primitive then costs three scalar instructions, not a wait for its type.)
        const bool culled = FIRST && ((primmask >> ((uint32_t)g & 31u)) & 1u) == 0u;
        if (culled && !DEBUG_BOUNDS) continue;
        const_u32_ptr hp = (const_u32_ptr)(uintptr_t)(p.prims + g);
        const uint32_t type = hp[0];
        if (type > 1u) continue;                             // MESH: never has geometry
        // padded world box of the primitive against this lane's ray (camera rays: box relative to the shared eye), through
        // an LDS broadcast read: VGPR operands keep the six fma at the full VALU rate (SGPR operands halve it)
        // near and far planes picked by ADDRESS: the LDS table holds every box once per direction octant as (near.xyz, 0)(far.xyz, -), so
        // the slab test needs no min / max per axis (6 VALU per primitive and trip less: + 5.7 % on config 2, profiles/r04/
        // ab_pretest_octant_planes.txt).  Same tn / tf bit for bit as the min / max form: fma is monotonic in the plane, so for 1/d > 0 the
        // lo plane IS the smaller product.  The eight 32-byte entries of a primitive cover all 64 banks once.  (The near entry's w
        // holds the 0 of "not behind the origin", which makes the read a full 16-byte one: + 0.6 %.)
        const float4 n4 = s_boxes[16 * g + octoff], f4 = s_boxes[16 * g + octoff + 1];
        const float tn = fmaxf(fmaxf(__builtin_fmaf(n4.x, dinv.x, oinv.x), __builtin_fmaf(n4.y, dinv.y, oinv.y)),
                               fmaxf(__builtin_fmaf(n4.z, dinv.z, oinv.z), n4.w));
        float tf = fminf(fminf(__builtin_fmaf(f4.x, dinv.x, oinv.x), __builtin_fmaf(f4.y, dinv.y, oinv.y)),
                         __builtin_fmaf(f4.z, dinv.z, oinv.z));
        float tnn = tn;
        const float tf_box = tf;
        static_assert(sizeof(PR) == 144, "the slab sits in the pad of the records' LDS copies");
        // a tilted cube fills little of its world box: the ray is also clipped against the slab between the two faces of the cube's thinnest
        // axis (host: KParams::slab_mask; planes n.x = d_lo / d_hi, n and d_lo in the pad of the record's LDS copy, d_hi in the far entry's w)
        if (SLAB && (uint32_t)g < 32u && ((p.slab_mask >> (uint32_t)g) & 1u) != 0u) {                 // wave-uniform
            const float4 s4 = reinterpret_cast<const float4 *>(&s_prims[g])[8];
            const float nd = __builtin_fmaf(s4.z, d.z, __builtin_fmaf(s4.y, d.y, s4.x * d.x));
            const float no = __builtin_fmaf(s4.z, o.z, __builtin_fmaf(s4.y, o.y, s4.x * o.x));
            const float ri = __builtin_amdgcn_rcpf(nd);        // (nd = 0: +-inf -- a ray inside the slab keeps (-inf, inf), one outside gets an empty range)
            const float ta = (s4.w - no) * ri, tb = (f4.w - no) * ri;
            tnn = fmaxf(tnn, fminf(ta, tb));
            tf = fminf(tf, fmaxf(ta, tb));
        }
        // the compare's lane mask straight from v_cmp, and back into a predicate without VALU work
        uint64_t mask = __builtin_amdgcn_fcmpf(tnn, tf, FCMP_OLE) & vmask;
        if (DEBUG_BOUNDS && SLAB) {
            // (bounds-checking build: the exact test of every pair the slab turned away; a hit is reported)
            const uint64_t turned = (__builtin_amdgcn_fcmpf(tn, tf_box, FCMP_OLE) & vmask) & ~mask;
            if (turned != 0ull) {
                f3 ip, in;
                const float ts = __builtin_amdgcn_inverse_ballot_w64(turned) ? intersectPrim<false>(s_prims[g], o, d, o, ip, in) : -1.0f;
                if (ts > 0.0f) dbgInRange(p, 43, (unsigned long long)g + 1000ull, 0ull);
            }
        }
        if (SKIP) {
            const uint64_t own = __builtin_amdgcn_uicmp(skip, (uint32_t)g, 32 /* ICMP_EQ */);      // lanes whose ray just left primitive g
            if (DEBUG_BOUNDS && (mask & own) != 0ull) {
                // (bounds-checking build: the skipped pair's exact test, in place; a hit is reported)
                f3 ip, in;
                const float ts = __builtin_amdgcn_inverse_ballot_w64(mask & own) ? intersectPrim<false>(s_prims[g], o, d, o, ip, in) : -1.0f;
                if (ts > 0.0f) dbgInRange(p, 40, (unsigned long long)g + 1000ull, 0ull);
            }
            mask &= ~own;
        }
        if (mask == 0ull) continue;
        if (DEBUG_BOUNDS && culled) { dbgInRange(p, 30, (unsigned long long)g + 1000ull, 0ull); continue; }
        const bool pass = __builtin_amdgcn_inverse_ballot_w64(mask);
        const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
        if (type == 0u) {                                    // wave-uniform
            if (pass) q.q[0][(tail[0] + rank) & (QCAP - 1u)] = lane | ((uint32_t)g << 8);
            tail[0] += (uint32_t)__popcll(mask);
            if (tail[0] - head[0] >= 64u) {
                const unsigned long long tb = LANE_BUDGET<FIRST>() ? __builtin_amdgcn_s_memtime() : 0ull;
                pairBatch<0u, FIRST, PR, MOTION>(p, s_prims, q, head[0], 64u, lane); head[0] += 64u;
                if (LANE_BUDGET<FIRST>()) fb_clk += __builtin_amdgcn_s_memtime() - tb;
            }
        } else {
            if (pass) q.q[1][(tail[1] + rank) & (QCAP - 1u)] = lane | ((uint32_t)g << 8);
            tail[1] += (uint32_t)__popcll(mask);
            if (tail[1] - head[1] >= 64u) {
                const unsigned long long tb = LANE_BUDGET<FIRST>() ? __builtin_amdgcn_s_memtime() : 0ull;
                pairBatch<1u, FIRST, PR, MOTION>(p, s_prims, q, head[1], 64u, lane); head[1] += 64u;
                if (LANE_BUDGET<FIRST>()) fb_clk += __builtin_amdgcn_s_memtime() - tb;
            }
        }
    }
    const uint64_t dbg_valid = DEBUG_PAIR ? __ballot(valid) : 0ull;
    if (DEBUG_PAIR && lane == 0) {
        atomicAdd(&p.st->dbg[0], (unsigned long long)tail[0]);
        atomicAdd(&p.st->dbg[1], (unsigned long long)tail[1]);
        atomicAdd(&p.st->dbg[4], (unsigned long long)__popcll(dbg_valid));
        atomicAdd(&p.st->dbg[5], (unsigned long long)((tail[0] + 63) / 64 + (tail[1] + 63) / 64));
        atomicAdd(&p.st->dbg[6], 1ull);
    }
    const unsigned long long ph_a = LANE_BUDGET<FIRST>() ? __builtin_amdgcn_s_memtime() : 0ull;
    const uint32_t left0 = tail[0] - head[0], left1 = tail[1] - head[1];      // both < 64
    if (left0 != 0u && left1 != 0u && left0 + left1 <= 64u) {
        pairBatch<2u, FIRST, PR, MOTION>(p, s_prims, q, head[0], left0, lane, head[1], left1);   // one mixed batch instead of two partial ones
        if (LANE_BUDGET<FIRST>()) lb_lane_clk += (__builtin_amdgcn_s_memtime() - ph_a) * (unsigned long long)(left0 + left1);
    } else {
        if (left0 != 0u) pairBatch<0u, FIRST, PR, MOTION>(p, s_prims, q, head[0], left0, lane);
        const unsigned long long ph_b = LANE_BUDGET<FIRST>() ? __builtin_amdgcn_s_memtime() : 0ull;
        if (LANE_BUDGET<FIRST>()) lb_lane_clk += (ph_b - ph_a) * (unsigned long long)left0;
        if (left1 != 0u) pairBatch<1u, FIRST, PR, MOTION>(p, s_prims, q, head[1], left1, lane);
        if (LANE_BUDGET<FIRST>()) lb_lane_clk += (__builtin_amdgcn_s_memtime() - ph_b) * (unsigned long long)left1;
    }
    wave_lds_fence();
    Hit h;
    if (LANE_BUDGET<FIRST>()) {
        h.dbg0 = ph_a - ph_in - fb_clk;                        // the pre-test loop without the full batches it ran
        h.dbg1 = __builtin_amdgcn_s_memtime() - ph_a;
        h.dbg2 = fb_clk;
        h.dbg3 = lb_lane_clk;
    }
    h.any = false;
    h.material = 0;
    h.prim = 0;
    h.self_ok = 0u;
    h.t = 0.0f;
    h.p = mk(0, 0, 0);
    h.n = mk(0, 0, 0);
    const unsigned long long k = q.key[lane];
    if (valid && k != KEY_NONE) {
        const float4 b = q.best[lane];
        const uint32_t meta = __float_as_uint(b.w);
        const uint32_t prim = (meta >> 8) & 0xFFFFFu;
        const uint32_t face = meta >> 28;
        const Prim &P = s_prims[prim];                       // per-lane gather from the LDS copy
        h.any = true;
        h.prim = prim;
        h.t = __uint_as_float((uint32_t)(k >> 32));
        h.p = mk(b.x, b.y, b.z);
        h.material = P.material;
        h.self_ok = P.self_r2 > 0.0f ? 1u : 0u;
        if (MOTION) {
            // the winner's transform at this lane's time: sphere centre = its translation column, box normal from its rows
            float minv[12], mfwd[12];
            motionRows(p, prim, mymt, minv, mfwd);
            h.n = (P.type == 0u) ? sphereNormal(h.p, mk(mfwd[3], mfwd[7], mfwd[11])) : boxNormal(mfwd, face);
        } else if (P.type == 0u) {
            const float4 c = *reinterpret_cast<const float4 *>(&P.cx);
            h.n = sphereNormal(h.p, mk(c.x, c.y, c.z));
            // (the normal's own squared length: is the reported point where a hit from outside belongs -- see Prim::self_r2)
            const f3 v = h.p - mk(c.x, c.y, c.z);
            h.self_ok = (P.self_r2 > 0.0f && dot(v, v) > P.self_r2) ? 1u : 0u;
        } else {
            const float4 fn = reinterpret_cast<const float4 *>(s_prims + p.nG)[prim * faceStride<PR>() + face];   // face-normal table behind the records
            h.n = mk(fn.x, fn.y, fn.z);
        }
    }
    return h;
}

// ---------------------------------------------------------------------------------------------------------------
// GEOM_WALK_PAIR
// ---------------------------------------------------------------------------------------------------------------
// append the lanes' pairs (sphere / box flags are per lane, g = primitive index) and run a batch when one is full;
// every lane of the wave must make the call
// (owner = the lane whose ray the pair belongs to: the calling lane itself except in the batched walk)
template <bool FIRST>
__device__ __forceinline__ void pushPairs(const KParams &p, const Prim *prims, const PairQueue &q, uint32_t (&head)[2],
                                          uint32_t (&tail)[2], bool is_sphere, bool is_box, uint32_t g, uint32_t lane, uint32_t owner)
{
    const uint64_t m0 = __ballot(is_sphere), m1 = __ballot(is_box);
    if (m0 != 0ull) {
        const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m0 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m0, 0u));
        if (is_sphere) q.q[0][(tail[0] + rank) & (QCAP - 1u)] = owner | (g << 8);
        tail[0] += (uint32_t)__popcll(m0);
        if (tail[0] - head[0] >= 64u) { pairBatch<0u, FIRST>(p, prims, q, head[0], 64u, lane); head[0] += 64u; }
    }
    if (m1 != 0ull) {
        const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m1 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m1, 0u));
        if (is_box) q.q[1][(tail[1] + rank) & (QCAP - 1u)] = owner | (g << 8);
        tail[1] += (uint32_t)__popcll(m1);
        if (tail[1] - head[1] >= 64u) { pairBatch<1u, FIRST>(p, prims, q, head[1], 64u, lane); head[1] += 64u; }
    }
}
template <bool FIRST>
__device__ __forceinline__ void flushPairs(const KParams &p, const Prim *prims, const PairQueue &q, uint32_t (&head)[2],
                                           uint32_t (&tail)[2], uint32_t lane)
{
    const uint32_t left0 = tail[0] - head[0], left1 = tail[1] - head[1];      // both < 64
    if (left0 != 0u && left1 != 0u && left0 + left1 <= 64u) {
        pairBatch<2u, FIRST>(p, prims, q, head[0], left0, lane, head[1], left1);
    } else {
        if (left0 != 0u) pairBatch<0u, FIRST>(p, prims, q, head[0], left0, lane);
        if (left1 != 0u) pairBatch<1u, FIRST>(p, prims, q, head[1], left1, lane);
    }
    head[0] = tail[0];
    head[1] = tail[1];
}

// Must be entered by all 64 lanes of the wave.  prims: global records (gathered per lane through L1/L2).
template <bool FIRST>
__device__ __forceinline__ Hit nearestHitWalkPairs(const KParams &p, const Prim *prims, const float4 *s_nodes, const PairQueue q,
                                                   f3 o, f3 d, bool valid, uint32_t lane)
{
    q.key[lane] = KEY_NONE;
    q.org[lane] = make_float4(o.x, o.y, o.z, d.x);
    q.dir[lane] = make_float2(d.y, d.z);
    uint32_t head[2] = {0u, 0u}, tail[2] = {0u, 0u};        // wave-uniform
    const f3 dinv = approxInverse(d);
    const f3 oinv = mk(-(o.x * dinv.x), -(o.y * dinv.y), -(o.z * dinv.z));
    // 1. the scene-spanning primitives (walls, big lights), wave-uniformly: box pre-test, pairs
    for (int k = 0; k < p.nbig; ++k) {
        const int g = p.big[k];
        const_u32_ptr hp = (const_u32_ptr)(uintptr_t)(p.prims + g);
        const uint32_t type = hp[0];
        if (type == 2u || type > 3u) continue;
        const_u32_ptr bq = (const_u32_ptr)(uintptr_t)(p.box_world + 8 * g);
        const float x0 = __builtin_fmaf(__uint_as_float(bq[0]), dinv.x, oinv.x), x1 = __builtin_fmaf(__uint_as_float(bq[4]), dinv.x, oinv.x);
        const float y0 = __builtin_fmaf(__uint_as_float(bq[1]), dinv.y, oinv.y), y1 = __builtin_fmaf(__uint_as_float(bq[5]), dinv.y, oinv.y);
        const float z0 = __builtin_fmaf(__uint_as_float(bq[2]), dinv.z, oinv.z), z1 = __builtin_fmaf(__uint_as_float(bq[6]), dinv.z, oinv.z);
        const float tn = fmaxf(fmaxf(fminf(x0, x1), fminf(y0, y1)), fmaxf(fminf(z0, z1), 0.0f));
        const float tf = fminf(fminf(fmaxf(x0, x1), fmaxf(y0, y1)), fmaxf(z0, z1));
        const bool pass = valid && tn <= tf;
        pushPairs<FIRST>(p, prims, q, head, tail, pass && type == 0u, pass && type == 1u, (uint32_t)g, lane, lane);
    }
    flushPairs<FIRST>(p, prims, q, head, tail, lane);
    wave_lds_fence();
    // 2. the hierarchy: box tests only, pruned by the distance of the best hit so far (reported distances fall short of
    // the true ones by getPointOnRay's epsilon: the slack covers it), leaves become pairs
    float tmax = __builtin_inff();
    {
        const unsigned long long k0 = q.key[lane];
        if (k0 != KEY_NONE) tmax = __uint_as_float((uint32_t)(k0 >> 32));
    }
    const uint32_t nn = (uint32_t)p.nnodes;
    uint32_t i = valid ? 0u : nn;
    uint32_t dbg_nodes = 0, dbg_leaves = 0, dbg_outer = 0, dbg_prev = 0, dbg_trips = 0;
    for (;;) {
        // A lane keeps walking past the leaves it reaches (they need no work here) and only stops when it has collected
        // four or left the tree: breaking at every leaf would make the wave wait for its slowest lane once per leaf
        // (measured: 141 loop trips per wave against 62 for the longest single walk).
        int l0 = -1, l1 = -1, l2 = -1, l3 = -1;
        if (DEBUG_BVH) dbg_outer++;
        while (i < nn) {
            if (DEBUG_BVH) dbg_nodes++;
            const float4 a = s_nodes[2 * i], b = s_nodes[2 * i + 1];
            const float x0 = __builtin_fmaf(a.x, dinv.x, oinv.x), x1 = __builtin_fmaf(b.x, dinv.x, oinv.x);
            const float y0 = __builtin_fmaf(a.y, dinv.y, oinv.y), y1 = __builtin_fmaf(b.y, dinv.y, oinv.y);
            const float z0 = __builtin_fmaf(a.z, dinv.z, oinv.z), z1 = __builtin_fmaf(b.z, dinv.z, oinv.z);
            const float tn = fmaxf(fmaxf(fminf(x0, x1), fminf(y0, y1)), fmaxf(fminf(z0, z1), 0.0f));
            const float tf = fminf(fminf(fmaxf(x0, x1), fmaxf(y0, y1)), fmaxf(z0, z1));
            if (!(tn <= tf) || tn * 0.999f - 1e-3f > tmax) {
                i = __float_as_uint(a.w);
                continue;
            }
            i = i + 1u;
            const int pr = (int)__float_as_uint(b.w);
            if (pr >= 0) {
                l3 = l2; l2 = l1; l1 = l0; l0 = pr;
                if (l3 >= 0) break;
            }
        }
        const bool leaf = l0 >= 0;
        if (DEBUG_BVH) dbg_leaves += (l0 >= 0) + (l1 >= 0) + (l2 >= 0) + (l3 >= 0);
        if (DEBUG_BVH) {
            uint32_t r = dbg_nodes - dbg_prev;
            dbg_prev = dbg_nodes;
            for (int s = 32; s > 0; s >>= 1) r = max(r, (uint32_t)__shfl_xor((int)r, s));
            dbg_trips += r;
        }
        if (__ballot(leaf) == 0ull) break;                   // every lane has left the tree
        const uint32_t done_before = head[0] + head[1];
        {
            const uint32_t type = (uint32_t)l0 >> 30, g = (uint32_t)l0 & 0x3FFFFFFFu;
            pushPairs<FIRST>(p, prims, q, head, tail, l0 >= 0 && type == 0u, l0 >= 0 && type == 1u, g, lane, lane);
        }
        if (__ballot(l1 >= 0) != 0ull) {
            const uint32_t type = (uint32_t)l1 >> 30, g = (uint32_t)l1 & 0x3FFFFFFFu;
            pushPairs<FIRST>(p, prims, q, head, tail, l1 >= 0 && type == 0u, l1 >= 0 && type == 1u, g, lane, lane);
        }
        if (__ballot(l2 >= 0) != 0ull) {
            const uint32_t type = (uint32_t)l2 >> 30, g = (uint32_t)l2 & 0x3FFFFFFFu;
            pushPairs<FIRST>(p, prims, q, head, tail, l2 >= 0 && type == 0u, l2 >= 0 && type == 1u, g, lane, lane);
        }
        if (__ballot(l3 >= 0) != 0ull) {
            const uint32_t type = (uint32_t)l3 >> 30, g = (uint32_t)l3 & 0x3FFFFFFFu;
            pushPairs<FIRST>(p, prims, q, head, tail, l3 >= 0 && type == 0u, l3 >= 0 && type == 1u, g, lane, lane);
        }
        if (head[0] + head[1] != done_before) {              // a batch ran: the bound may have come down
            const unsigned long long k1 = q.key[lane];
            if (k1 != KEY_NONE) tmax = __uint_as_float((uint32_t)(k1 >> 32));
        }
    }
    flushPairs<FIRST>(p, prims, q, head, tail, lane);
    wave_lds_fence();
    if (DEBUG_BVH) {
        uint32_t mn = dbg_nodes, ml = dbg_leaves, mo = dbg_outer;
        for (int s = 32; s > 0; s >>= 1) {
            mn = max(mn, (uint32_t)__shfl_xor((int)mn, s));
            ml = max(ml, (uint32_t)__shfl_xor((int)ml, s));
            mo = max(mo, (uint32_t)__shfl_xor((int)mo, s));
        }
        if (valid) { atomicAdd(&p.st->dbg[0], (unsigned long long)dbg_nodes); atomicAdd(&p.st->dbg[1], (unsigned long long)dbg_leaves); atomicAdd(&p.st->dbg[2], 1ull); }
        if (lane == 0) {
            atomicAdd(&p.st->dbg[3], (unsigned long long)dbg_trips);
            atomicAdd(&p.st->dbg[4], (unsigned long long)ml);
            atomicAdd(&p.st->dbg[5], (unsigned long long)mo);
            atomicAdd(&p.st->dbg[6], 1ull);
            atomicAdd(&p.st->dbg[7], (unsigned long long)(tail[0] + tail[1]));
        }
    }
    Hit h;
    h.any = false;
    h.material = 0;
    h.prim = 0;
    h.t = 0.0f;
    h.p = mk(0, 0, 0);
    h.n = mk(0, 0, 0);
    const unsigned long long k = q.key[lane];
    if (valid && k != KEY_NONE) {
        const float4 b = q.best[lane];
        const uint32_t meta = __float_as_uint(b.w);
        const uint32_t prim = (meta >> 8) & 0xFFFFFu;
        const uint32_t face = meta >> 28;
        const Prim &P = prims[prim];
        h.any = true;
        h.prim = prim;
        h.t = __uint_as_float((uint32_t)(k >> 32));
        h.p = mk(b.x, b.y, b.z);
        h.material = P.material;
        if (P.type == 0u) h.n = sphereNormal(h.p, mk(P.cx, P.cy, P.cz));
        else {
            const float4 fn = reinterpret_cast<const float4 *>(p.face_n)[prim * 8u + face];
            h.n = mk(fn.x, fn.y, fn.z);
        }
    }
    return h;
}


// ---------------------------------------------------------------------------------------------------------------
// GEOM_WALK4
// ---------------------------------------------------------------------------------------------------------------
static constexpr uint32_t W4_STACK = 512;                              // traversal entries per wave (a power of two: the ring wraps with an AND)
// per wave: the two pair queues, keys and best hits (2.5 KB), the entry ring and -- only when the scene has triangles --
// their queue: 4.5 KB.  The rays and their reciprocals stay in their owners' registers and reach the lane that tests an
// entry or a pair through ds_bpermute, which costs no LDS storage (it was 7.5 KB with LDS copies of both, and 9 VGPRs
// more for their address arithmetic): three 512-thread workgroups per CU beside a 13 KB node copy instead of two.
static constexpr uint32_t W4_PAIR_BYTES = 2 * QCAP * 4 + 64 * 8 + 64 * 16;
__host__ __device__ constexpr uint32_t walk4_wave_bytes(int ntri) { return W4_PAIR_BYTES + W4_STACK * 4 + (ntri > 0 ? QCAP * 4 : 0); }
struct Walk4 {
    const unsigned char *nodes;   // LDS copy of the 4-wide hierarchy (ptd::W4_FLOATS floats per node)
    uint32_t *stack;              // [W4_STACK] owner lane | direction signs << 6 | child word (bit 31 leaf, bit 30 cube, bit 29 triangle, bits 9..28 index)
};

// append the lanes' (ray, triangle) pairs and run a batch when one is full; every lane of the wave must make the call
template <bool FIRST>
__device__ __forceinline__ void pushTriangles(const KParams &p, const Prim *prims, const PairQueue &q, uint32_t &thead, uint32_t &ttail,
                                              bool is_tri, uint32_t g, uint32_t lane, uint32_t owner)
{
    const uint64_t m = __ballot(is_tri);
    if (m == 0ull) return;
    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
    if (is_tri) q.tq[(ttail + rank) & (QCAP - 1u)] = owner | (g << 8);
    ttail += (uint32_t)__popcll(m);
    if (ttail - thead >= 64u) { pairBatch<3u, FIRST>(p, prims, q, thead, 64u, lane); thead += 64u; }
}

// Must be entered by all 64 lanes of the wave.  prims: global records (gathered per lane through L1/L2).
// SKIP (resident paths): `skip` = the primitive this lane's ray just left on its outside (see nearestHitPairs); a popped leaf entry
// asks its OWNER's skip through ds_bpermute.
template <bool FIRST, bool SKIP = false>
__device__ __forceinline__ Hit nearestHitWalk4(const KParams &p, const Prim *prims, const Walk4 w, const PairQueue q,
                                               f3 o, f3 d, bool valid, uint32_t lane, uint32_t span, uint32_t skip = 0xFFFFFFFFu)
{
    q.key[lane] = KEY_NONE;                                  // (q.po / q.pd: the ray stays in this lane's registers)
    uint32_t head[2] = {0u, 0u}, tail[2] = {0u, 0u};        // wave-uniform
    uint32_t thead = 0u, ttail = 0u;                         // the triangle queue
    const f3 dinv = approxInverse(d);
    const f3 oinv = mk(-(o.x * dinv.x), -(o.y * dinv.y), -(o.z * dinv.z));
    // the ray's three direction signs travel IN its entries (bits 6..8), so a popped entry addresses its node's near and
    // far planes by itself and the node reads go out together with the reads of the owner's ray (one LDS round trip
    // less per step than with the signs parked beside the reciprocals)
    const uint32_t mysigns = ((__float_as_uint(dinv.x) >> 31) << 6) | ((__float_as_uint(dinv.y) >> 31) << 7) | ((__float_as_uint(dinv.z) >> 31) << 8);
    // 1. the scene-spanning primitives (walls, big lights), wave-uniformly: box pre-test, pairs.
    // Camera rays (no lens, tiles whose chunks are 64-pixel spans): the host lists, per span, every primitive whose padded
    // box reaches into the span's pixel frustum (pt_context.hip); when that list is short it replaces both the
    // scene-spanning primitives and the walk -- a span of the 256-sphere cloud sees a dozen primitives, not a hierarchy.
    // (The bounds-checking build walks anyway and reports a leaf that passes its box test without being on the list.)
    uint32_t list_off = 0u, list_n = 0xFFFFFFFFu;
    if (FIRST && span != 0xFFFFFFFFu) {
        const_u32_ptr so = (const_u32_ptr)(uintptr_t)(p.span_off + 2u * span);
        list_off = so[0];
        list_n = so[1];                                       // 0xFFFFFFFF: too many to list, walk
    }
    const bool listed = list_n != 0xFFFFFFFFu;
    const int ncand = listed ? (int)list_n : p.nbig;
    const_u32_ptr lst = (const_u32_ptr)(uintptr_t)(p.span_list + list_off);
    for (int k = 0; k < ncand; ++k) {
        const int g = listed ? (int)lst[k] : p.big[k];
        const_u32_ptr hp = (const_u32_ptr)(uintptr_t)(p.prims + g);
        const uint32_t type = hp[0];
        if (type == 2u || type > 3u) continue;                // (MESH geoms have no geometry of their own)
        const_u32_ptr bq = (const_u32_ptr)(uintptr_t)(p.box_world + 8 * g);
        const float x0 = __builtin_fmaf(__uint_as_float(bq[0]), dinv.x, oinv.x), x1 = __builtin_fmaf(__uint_as_float(bq[4]), dinv.x, oinv.x);
        const float y0 = __builtin_fmaf(__uint_as_float(bq[1]), dinv.y, oinv.y), y1 = __builtin_fmaf(__uint_as_float(bq[5]), dinv.y, oinv.y);
        const float z0 = __builtin_fmaf(__uint_as_float(bq[2]), dinv.z, oinv.z), z1 = __builtin_fmaf(__uint_as_float(bq[6]), dinv.z, oinv.z);
        const float tn = fmaxf(fmaxf(fminf(x0, x1), fminf(y0, y1)), fmaxf(fminf(z0, z1), 0.0f));
        const float tf = fminf(fminf(fmaxf(x0, x1), fmaxf(y0, y1)), fmaxf(z0, z1));
        bool pass = valid && tn <= tf;
        if (SKIP) {
            if (DEBUG_BOUNDS && pass && skip == (uint32_t)g) {      // (bounds-checking build: the skipped pair's exact test; a hit is reported)
                f3 ip, in;
                if (intersectPrim<false>(prims[g], o, d, o, ip, in) > 0.0f) dbgInRange(p, 41, (unsigned long long)g + 1000ull, 0ull);
            }
            pass = pass && skip != (uint32_t)g;
        }
        if (type == 3u) pushTriangles<FIRST>(p, prims, q, thead, ttail, pass, (uint32_t)g, lane, lane);
        else pushPairs<FIRST>(p, prims, q, head, tail, pass && type == 0u, pass && type == 1u, (uint32_t)g, lane, lane);
    }
    // (no flush here: the pairs wait for the leaves' pairs to fill their batches.  A bound from the walls would prune
    // nothing in a closed room -- the whole hierarchy lies in front of them -- and a full batch runs as soon as 64 pairs
    // are queued anyway.)
    // 2. the hierarchy.  Every ray starts with one entry for the root; a step pops the top nb entries (one per lane).
    uint32_t top = 0u;                                        // wave-uniform
    if (p.nnodes4 > 0 && (!listed || DEBUG_BOUNDS)) {
        const uint64_t vm = __ballot(valid);
        const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(vm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)vm, 0u));
        if (valid) w.stack[rank] = lane | mysigns;            // owner | signs | node 0
        top = (uint32_t)__popcll(vm);
    }
    // The entries live in a ring (a deque): pushes go to its tail; a step pops its nb oldest entries from the HEAD while
    // the ring has room for a full batch's children -- generation by generation, so the steps stay full (the rays of a
    // chunk descend the hierarchy level by level: 10.8 -> 9.x steps per round on the 256-primitive cloud) -- and its nb
    // newest from the TAIL when it has not: depth-first order, whose growth is bounded.  A popped entry leaves at most 4
    // behind (net +3): tail batches only run while 3*depth + 4 entries stay in reserve, below that entries are popped
    // one at a time, and plain depth-first descent needs no more than that reserve above the point it started from.
    const int reserve = 3 * p.wdepth + 4;
    uint32_t qh = 0u;                                         // ring index of the oldest entry (wave-uniform)
    uint32_t dbg_steps = 0, dbg_entries = 0;
    auto wrap = [](uint32_t x) -> uint32_t { return x & (W4_STACK - 1u); };
    while (top != 0u) {                                       // top = entries in the ring
        const bool fifo = (int)top + 3 * 64 <= (int)W4_STACK - reserve;
        uint32_t nb = top < 64u ? top : 64u;
        if (!fifo) {
            const int room = ((int)W4_STACK - reserve - (int)top) / 3;
            if (room < (int)nb) nb = room < 1 ? 1u : (uint32_t)room;
        }
        wave_lds_fence();
        const bool act = lane < nb;
        const uint32_t e = act ? w.stack[wrap(fifo ? qh + lane : qh + top - 1u - lane)] : 0u;
        if (fifo) qh = wrap(qh + nb);
        top -= nb;
        if (DEBUG_BVH) { dbg_steps++; dbg_entries += nb; }
        // lane masks straight from the compares (an inactive lane holds e = 0: neither leaf nor, below, inner)
        const uint64_t leafm = __builtin_amdgcn_sicmp((int)e, 0, 40 /* ICMP_SLT: bit 31 set */);
        const bool leaf = __builtin_amdgcn_inverse_ballot_w64(leafm);
        const uint32_t owner = e & 63u, index = (e >> 9) & 0xFFFFFu;
        if (leafm != 0ull) {
            // child words: 100x.. sphere, 110x.. cube, 101x.. triangle leaf
            const uint32_t cls = e >> 29;
            bool sph = __builtin_amdgcn_inverse_ballot_w64(__builtin_amdgcn_uicmp(cls, 4u, 32 /* ICMP_EQ */));
            bool cube = __builtin_amdgcn_inverse_ballot_w64(__builtin_amdgcn_uicmp(cls, 6u, 32));
            const bool tri = __builtin_amdgcn_inverse_ballot_w64(__builtin_amdgcn_uicmp(cls, 5u, 32));
            if (SKIP) {
                // the leaf of the primitive the owner's ray just left: no pair (every lane of the wave is here: uniform control flow)
                const uint32_t oskip = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(owner << 2), (int)skip);
                const bool own = leaf && index == oskip;
                if (DEBUG_BOUNDS && __ballot(own) != 0ull) {
                    // (bounds-checking build: the skipped pair's exact test on the owner's ray; a hit is reported)
                    auto fo = [&](float v) { return __int_as_float(__builtin_amdgcn_ds_bpermute((int)(owner << 2), __float_as_int(v))); };
                    const f3 oo = mk(fo(o.x), fo(o.y), fo(o.z)), od = mk(fo(d.x), fo(d.y), fo(d.z));
                    f3 ip, in;
                    if (own && intersectPrim<false>(prims[index], oo, od, oo, ip, in) > 0.0f) dbgInRange(p, 42, (unsigned long long)index + 1000ull, 0ull);
                }
                sph = sph && !own;
                cube = cube && !own;
            }
            if (DEBUG_BOUNDS && listed) {                     // the list already queued the pairs: only check it
                bool on_list = !leaf;
                for (uint32_t j = 0; j < list_n; ++j) on_list = on_list || lst[j] == index;
                if (!on_list) dbgInRange(p, 31, (unsigned long long)index + 1000ull, 0ull);
            } else {
            pushPairs<FIRST>(p, prims, q, head, tail, sph, cube, index, lane, owner);
            if (p.ntri > 0) pushTriangles<FIRST>(p, prims, q, thead, ttail, tri, index, lane, owner);
            }
        }
        const uint64_t im = (nb >= 64u ? ~0ull : ((1ull << nb) - 1ull)) & ~leafm;      // active and not a leaf
        const bool inner = __builtin_amdgcn_inverse_ballot_w64(im);
        if (im == 0ull) continue;
        // the owner's (1/d, -o/d) out of ITS registers (every lane of the wave is here: uniform control flow)
        const int oaddr = (int)(owner << 2);
        auto fetch = [&](float v) { return __int_as_float(__builtin_amdgcn_ds_bpermute(oaddr, __float_as_int(v))); };
        const float4 A = make_float4(fetch(dinv.x), fetch(dinv.y), fetch(dinv.z), fetch(oinv.x));
        const float2 B = make_float2(fetch(oinv.y), fetch(oinv.z));
        // the owner's best hit so far bounds the boxes worth entering (reported distances fall short of the true ones by
        // getPointOnRay's epsilon and |d| is 1 +- 1e-6: the slack covers both); no hit yet: all-ones key -> +inf
        uint32_t kh = reinterpret_cast<const uint32_t *>(q.key)[2 * owner + 1];
        kh = kh < 0x7F800000u ? kh : 0x7F800000u;
        const float tmax = (__uint_as_float(kh) + 1e-3f) * 1.0011f;
        // (lanes whose entry is a leaf or that have none read node 0: a leaf's index is a PRIMITIVE index, and with the
        // nodes in global memory -- GEOM_WALK4G -- reading "node" 5 999 of a 2 400-node hierarchy is a page fault: it
        // was one, whenever the pages behind the node array happened to be unmapped)
        const uint32_t nidx = (inner && dbgInRange(p, 20, index, (unsigned long long)p.nnodes4)) ? index : 0u;
        const unsigned char *nd = w.nodes + nidx * (uint32_t)(W4_FLOATS * 4);
        // byte offset of the near planes inside an axis' 32 bytes: the hi planes (16) when the ray runs towards -axis
        const uint32_t sx = (e >> 2) & 16u, sy = (e >> 3) & 16u, sz = (e >> 4) & 16u;
        const float4 nx = *reinterpret_cast<const float4 *>(nd + sx), fx = *reinterpret_cast<const float4 *>(nd + (sx ^ 16u));
        const float4 ny = *reinterpret_cast<const float4 *>(nd + 32 + sy), fy = *reinterpret_cast<const float4 *>(nd + 32 + (sy ^ 16u));
        const float4 nz = *reinterpret_cast<const float4 *>(nd + 64 + sz), fz = *reinterpret_cast<const float4 *>(nd + 64 + (sz ^ 16u));
        const uint4 ch = *reinterpret_cast<const uint4 *>(nd + 96);
        const float nxa[4] = {nx.x, nx.y, nx.z, nx.w}, fxa[4] = {fx.x, fx.y, fx.z, fx.w};
        const float nya[4] = {ny.x, ny.y, ny.z, ny.w}, fya[4] = {fy.x, fy.y, fy.z, fy.w};
        const float nza[4] = {nz.x, nz.y, nz.z, nz.w}, fza[4] = {fz.x, fz.y, fz.z, fz.w};
        const uint32_t cha[4] = {ch.x, ch.y, ch.z, ch.w};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float tn = fmaxf(fmaxf(__builtin_fmaf(nxa[c], A.x, A.w), __builtin_fmaf(nya[c], A.y, B.x)),
                                   fmaxf(__builtin_fmaf(nza[c], A.z, B.y), 0.0f));
            const float tf = fminf(fminf(__builtin_fmaf(fxa[c], A.x, A.w), __builtin_fmaf(fya[c], A.y, B.x)),
                                   fminf(__builtin_fmaf(fza[c], A.z, B.y), tmax));
            const uint64_t mask = __builtin_amdgcn_fcmpf(tn, tf, FCMP_OLE) & im;
            if (mask == 0ull) continue;
            const bool pass = __builtin_amdgcn_inverse_ballot_w64(mask);
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
            if (pass) w.stack[wrap(wrap(qh + top) + rank)] = (e & 0x1FFu) | cha[c];
            top += (uint32_t)__popcll(mask);
        }
    }
    flushPairs<FIRST>(p, prims, q, head, tail, lane);
    if (ttail != thead) pairBatch<3u, FIRST>(p, prims, q, thead, ttail - thead, lane);
    wave_lds_fence();
    const uint64_t dbg_vm = DEBUG_BVH ? __ballot(valid) : 0ull;
    if (DEBUG_BVH && lane == 0) {
        atomicAdd(&p.st->dbg[0], (unsigned long long)dbg_entries);
        atomicAdd(&p.st->dbg[3], (unsigned long long)dbg_steps);
        atomicAdd(&p.st->dbg[6], 1ull);
        atomicAdd(&p.st->dbg[7], (unsigned long long)(tail[0] + tail[1]));
        atomicAdd(&p.st->dbg[2], (unsigned long long)__popcll(dbg_vm));
    }
    Hit h;
    h.any = false;
    h.material = 0;
    h.prim = 0;
    h.t = 0.0f;
    h.p = mk(0, 0, 0);
    h.n = mk(0, 0, 0);
    const unsigned long long k = q.key[lane];
    if (valid && k != KEY_NONE) {
        const float4 b = q.best[lane];
        const uint32_t meta = __float_as_uint(b.w);
        const uint32_t prim = (meta >> 8) & 0xFFFFFu;
        const uint32_t face = meta >> 28;
        const Prim &P = prims[prim];
        h.any = true;
        h.prim = prim;
        h.t = __uint_as_float((uint32_t)(k >> 32));
        h.p = mk(b.x, b.y, b.z);
        h.material = P.material;
        h.self_ok = P.self_r2 > 0.0f ? 1u : 0u;
        if (P.type == 0u) {
            h.n = sphereNormal(h.p, mk(P.cx, P.cy, P.cz));
            const f3 v = h.p - mk(P.cx, P.cy, P.cz);             // (see Prim::self_r2)
            h.self_ok = (P.self_r2 > 0.0f && dot(v, v) > P.self_r2) ? 1u : 0u;
        }
        else if (P.type == 3u) h.n = mk(P.fwd[0], P.fwd[1], P.fwd[2]);
        else {
            const float4 fn = reinterpret_cast<const float4 *>(p.face_n)[prim * 8u + face];
            h.n = mk(fn.x, fn.y, fn.z);
        }
    }
    return h;
}

// nearest hit of (o, d) for the lanes with want == true, by the GEOM path.  Every lane of the wave must make the call
// (the hit queue uses all 64 lanes as workers whatever their own ray).
template <int GEOM, bool FIRST, bool MOTION = false, bool SKIP = false, bool SLAB = false>
__device__ __forceinline__ Hit nearestHit(const KParams &p, const PrimPad *s_prims, const float4 *s_nodes, const WaveQueue &wq,
                                          f3 o, f3 d, bool want, uint32_t lane, uint32_t primmask = 0xFFFFFFFFu, MotionTime mt = MotionTime(),
                                          uint32_t skip = 0xFFFFFFFFu)
{
    if (GEOM == GEOM_QUEUE) return nearestHitQueued<FIRST>(p, s_prims, wq, o, d, want, lane);
    if (GEOM == GEOM_PAIR || GEOM == GEOM_WALK_PAIR || GEOM == GEOM_WALK4 || GEOM == GEOM_WALK4G) {
        PairQueue pq;
        unsigned char *b = reinterpret_cast<unsigned char *>(wq.rec);         // same LDS region as the hit queue
        pq.q[0] = reinterpret_cast<uint32_t *>(b);
        pq.q[1] = reinterpret_cast<uint32_t *>(b + QCAP * 4);
        pq.key = reinterpret_cast<unsigned long long *>(b + 2 * QCAP * 4);
        pq.best = reinterpret_cast<float4 *>(b + 2 * QCAP * 4 + 64 * 8);
        pq.org = reinterpret_cast<float4 *>(b + 2 * QCAP * 4 + 64 * 8 + 64 * 16);
        pq.dir = reinterpret_cast<float2 *>(b + 2 * QCAP * 4 + 64 * 8 + 2 * 64 * 16);
        pq.mt = reinterpret_cast<uint2 *>(b + PAIR_QUEUE_BYTES);           // (carved for FEAT_MOTION instances only)
        pq.dbg = p.st->dbg;
        pq.tq = nullptr;
        if (GEOM == GEOM_WALK4 || GEOM == GEOM_WALK4G) {
            Walk4 w4;
            w4.nodes = (GEOM == GEOM_WALK4G) ? reinterpret_cast<const unsigned char *>(p.bvh4) : reinterpret_cast<const unsigned char *>(s_nodes);
            w4.stack = reinterpret_cast<uint32_t *>(b + W4_PAIR_BYTES);
            pq.tq = reinterpret_cast<uint32_t *>(b + W4_PAIR_BYTES + W4_STACK * 4);
            pq.org = nullptr;                                    // rays by ds_bpermute from their owners' registers
            pq.dir = nullptr;
            pq.po = o;
            pq.pd = d;
            return nearestHitWalk4<FIRST, SKIP>(p, p.prims, w4, pq, o, d, want, lane, (FIRST && p.span_off != nullptr) ? primmask : 0xFFFFFFFFu, skip);
        }
        if (GEOM == GEOM_WALK_PAIR) return nearestHitWalkPairs<FIRST>(p, p.prims, s_nodes, pq, o, d, want, lane);
        return nearestHitPairs<FIRST, PrimPad, MOTION, SKIP, SLAB && GEOM == GEOM_PAIR>(p, s_prims, s_nodes, pq, o, d, want, lane, primmask, mt, skip);
    }
    Hit h;
    h.any = false;
    h.material = 0;
    h.prim = 0;
    h.self_ok = 0u;
    h.t = 0.0f;
    h.p = mk(0, 0, 0);
    h.n = mk(0, 0, 0);
    if (GEOM == GEOM_BVH) {
        if (want) h = nearestHitBvh<FIRST>(p, p.prims, s_nodes, o, d);
    } else {
        if (want) h = nearestHitDirect<(GEOM == GEOM_LDS ? GEOM_LDS : GEOM_SCALAR), FIRST, PrimPad>(p, s_prims, o, d);
    }
    return h;
}

// global pixel index (x + y*W of the frame: what the RNG streams are keyed on) of tile-local pixel pl
__device__ __forceinline__ uint32_t globalPixel(const KParams &p, uint32_t pl)
{
    if (p.strip_span == 0u) return pl + p.pix_offset;
    const uint32_t j = (uint32_t)(((unsigned long long)pl * p.strip_magic) >> p.strip_shift);     // pl / strip_span
    return pl + p.strip_span * (j * (p.strip_world - 1u) + p.strip_rank);
}

// COMPACT: 0 = rays keep their slot (validation / ablation), 1 = per-wave reservation on sharded counters
// (no workgroup barrier), 2 = LDS scan over the workgroup's waves + one atomic per workgroup.
// NEE: explicit light sampling at diffuse vertices (pt_options.direct_light).  A ray then carries in bit 31 of its
// pixel word whether its previous vertex sampled the lights (a light hit by chance adds nothing), the radiance planes
// accumulate along the path (every path initialises its entry at bounce 0), and each chunk makes a second pass through
// the nearest-hit machinery for the shadow rays.
// 5 waves per SIMD (<= 96 VGPRs) is where the plain kernels sit and what hides their LDS / pool latency: ask for it, so
// that a few registers more do not silently drop a wave (4 waves: -5 %).
// FEAT: optional features built as kernel instances of their own, so that the default path keeps its registers and
// instruction count (measured: the scattering code alone costs the plain kernel 1.1 % when compiled in):
// bit 0 = NEE (pt_options.direct_light), bit 1 = MEDIA (pt_options.scatter: subsurface random walk).
// bit 2 = MOTION (pt_options.motion_per_ray: a shutter time per ray; scalar geometry path only).
// bit 3 = RESIDENT (pt_options.resident, round 4): ONE launch for all the later bounces.  `bounce` is the first bounce the
//         launch traces (1: the camera kernel's survivors); a wave takes its rays from that bounce's pool, and a path that goes on
//         stays in its lane's registers -- origin, direction, throughput, pixel word and its own bounce number -- for the next
//         trip of the wave's loop instead of travelling through the pools: no reservation atomic, no pool write, no pool read
//         for bounces 2 .. depth - 1, and depth - 2 launches (ramp, LDS staging, tail) fewer per batch.  Lanes whose path has
//         ended are refilled from the pool (KParams::refill_min free lanes trigger it).  Nothing depends on which lane or trip
//         traces a path: every RNG stream is keyed on (global pixel, iteration, bounce) -- the stream keys of all bounces sit in an
//         LDS table --, a path still writes its one radiance sample when it ends, and the per-bounce live counts come from an
//         LDS histogram of the bounce each path ended at.
enum { FEAT_NEE = 1, FEAT_MEDIA = 2, FEAT_MOTION = 4, FEAT_RESIDENT = 8, FEAT_SLAB = 16 };
template <int WG, bool FIRST, int GEOM, int COMPACT, int FEAT = 0>
// (the resident-path instances of the batched walks are held to 80 VGPRs -- 6 waves per SIMD, three 512-thread workgroups per CU,
// what the launch-per-bounce kernels reach unasked)
__global__ __launch_bounds__(WG, ((FEAT & ~FEAT_SLAB) == FEAT_RESIDENT && (GEOM == GEOM_WALK4 || GEOM == GEOM_WALK4G)) ? 6
                                 : ((WG <= 256 && (FEAT & ~(FEAT_RESIDENT | FEAT_SLAB)) == 0) ? 5 : 1)) void k_bounce(const KParams p, const int bounce)
{
    constexpr bool NEE = (FEAT & FEAT_NEE) != 0, MEDIA = (FEAT & FEAT_MEDIA) != 0, MOTION = (FEAT & FEAT_MOTION) != 0;
    constexpr bool RESIDENT = (FEAT & FEAT_RESIDENT) != 0;
    constexpr bool SLAB = (FEAT & FEAT_SLAB) != 0;       // pair path: the pre-test also clips tilted cubes against the slab of their thinnest axis (plain kernels only)
    static_assert(!RESIDENT || (!FIRST && COMPACT == 1 && !MOTION), "resident paths: later bounces, compaction 1, no per-ray shutter time");
    constexpr int NW = WG / 64;
    constexpr bool PRIMS_IN_LDS = (GEOM == GEOM_LDS || GEOM == GEOM_QUEUE || GEOM == GEOM_PAIR);   // GEOM_BVH gathers records from global memory (L1/L2)
    extern __shared__ __attribute__((aligned(128))) unsigned char smem[];
    // LDS carve: [prims nG*128 B (GEOM 1,2)] [per-wave hit queues (GEOM 2)] [material planes] [scan scratch]
    PrimPad *s_prims = reinterpret_cast<PrimPad *>(smem);
    const int prim_bytes = PRIMS_IN_LDS ? p.nG * (int)(2 * sizeof(PrimPad)) : 0;     // records, then 8 face normals each (padded alike)
    const float4 *s_nodes = reinterpret_cast<const float4 *>(smem + prim_bytes);
    // pair queue: the same region holds the primitives' padded boxes (2 float4 each; relative to the eye for camera rays)
    const int node_bytes = (GEOM == GEOM_BVH || GEOM == GEOM_WALK_PAIR) ? p.nnodes * (int)sizeof(BvhNode)
                           : (GEOM == GEOM_WALK4 ? ((p.nnodes4 * W4_FLOATS * 4 + 127) & ~127)
                                                 : (GEOM == GEOM_PAIR ? p.nG * PAIR_BOX_BYTES * (NEE ? 2 : 1) : 0));
    unsigned char *s_queue = smem + prim_bytes + node_bytes;
    const int WAVE_LDS = (GEOM == GEOM_QUEUE) ? (int)WAVE_QUEUE_BYTES
                         : ((GEOM == GEOM_PAIR || GEOM == GEOM_WALK_PAIR) ? (int)((MOTION && GEOM == GEOM_PAIR) ? PAIR_QUEUE_MOTION_BYTES : PAIR_QUEUE_BYTES)
                            : ((GEOM == GEOM_WALK4 || GEOM == GEOM_WALK4G) ? (int)walk4_wave_bytes(p.ntri) : 0));
    const int queue_bytes = NW * WAVE_LDS;
    float *s_mats = reinterpret_cast<float *>(s_queue + queue_bytes);
    const int mat_words = (p.nM * M_PLANES + 3) & ~3;
    uint32_t *s_scan = reinterpret_cast<uint32_t *>(s_mats + mat_words);   // [2][NW] wave totals, [2] bases
    // RESIDENT: stream keys of every (bounce, iteration slot) and the histogram of the bounce each path ended at
    uint32_t *s_keys = s_scan + ((2 * NW + 2 + 3) & ~3);                   // [depth][MAXSLOT]
    uint32_t *s_term = s_keys + (RESIDENT ? p.depth * MAXSLOT : 0);        // [depth]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);       // wave-uniform: chunk bookkeeping runs on the scalar unit

    WaveQueue wq;
    {
        unsigned char *b = s_queue + wave * WAVE_LDS;
        wq.rec = reinterpret_cast<float4 *>(b);
        wq.key = reinterpret_cast<unsigned long long *>(b + QCAP * 32);
        wq.best = reinterpret_cast<float4 *>(b + QCAP * 32 + 64 * 8);
        wq.org = reinterpret_cast<float4 *>(b + QCAP * 32 + 64 * 8 + 64 * 16);
    }

    if (PRIMS_IN_LDS) {
        // global records are 8 uint4 each; the LDS copies take 9 (PrimPad), the pad stays unwritten
        const uint4 *src = reinterpret_cast<const uint4 *>(p.prims);
        uint4 *dst = reinterpret_cast<uint4 *>(s_prims);
        for (int k = tid; k < p.nG * 8; k += WG) dst[(k >> 3) * 9 + (k & 7)] = src[k];
        if (SLAB)                                             // ... and in the pad the pre-test's slab (n.xyz, d_lo)
            for (int k = tid; k < p.nG; k += WG) dst[k * 9 + 8] = reinterpret_cast<const uint4 *>(p.slab_n)[k];
        const uint4 *fsrc = reinterpret_cast<const uint4 *>(p.face_n);
        for (int k = tid; k < p.nG * 8; k += WG) dst[p.nG * 9 + (k >> 3) * 9 + (k & 7)] = fsrc[k];
    }
    if (GEOM == GEOM_PAIR) {
        // (FEAT_MOTION: box_world holds the boxes swept over the shutter interval, and camera rays have no common eye)
        const uint4 *src = reinterpret_cast<const uint4 *>((FIRST && !MOTION && !(p.lens_radius > 0.0f)) ? p.box_eye : p.box_world);
        uint4 *dst = reinterpret_cast<uint4 *>(smem + prim_bytes);
        // per primitive eight entries (near.xyz, 0)(far.xyz, -), one per direction octant (bit 0 / 1 / 2 = the ray runs towards -x / -y / -z)
        auto stage_oct = [&](const uint4 *from, uint4 *to) {
            for (int k = tid; k < p.nG * 8; k += WG) {
                const uint4 lo = from[2 * (k >> 3)], hi = from[2 * (k >> 3) + 1];
                const int oc = k & 7;
                to[2 * k] = make_uint4((oc & 1) ? hi.x : lo.x, (oc & 2) ? hi.y : lo.y, (oc & 4) ? hi.z : lo.z, 0u);
                to[2 * k + 1] = make_uint4((oc & 1) ? lo.x : hi.x, (oc & 2) ? lo.y : hi.y, (oc & 4) ? lo.z : hi.z, hi.w);     // (w: the slab's d_hi)
            }
        };
        stage_oct(src, dst);
        if (NEE) stage_oct(reinterpret_cast<const uint4 *>(p.box_world), dst + p.nG * 16);      // shadow rays start anywhere: world boxes, second half
    }
    if (GEOM == GEOM_BVH || GEOM == GEOM_WALK_PAIR) {
        const uint4 *src = reinterpret_cast<const uint4 *>(p.bvh);
        uint4 *dst = reinterpret_cast<uint4 *>(smem + prim_bytes);
        for (int k = tid; k < p.nnodes * 2; k += WG) dst[k] = src[k];
    }
    if (GEOM == GEOM_WALK4) {
        const uint4 *src = reinterpret_cast<const uint4 *>(p.bvh4);
        uint4 *dst = reinterpret_cast<uint4 *>(smem + prim_bytes);
        for (int k = tid; k < p.nnodes4 * (W4_FLOATS / 4); k += WG) dst[k] = src[k];
    }
    for (int k = tid; k < p.nM * M_PLANES; k += WG) s_mats[k] = p.mats[k];
    __syncthreads();

    IterState *st = p.st;
    const uint32_t iter = st->iter;
    const uint32_t serial = st->serial;                // stamp of this batch's radiance-plane entries
    unsigned long long clk0 = 0, rt0 = 0;
    if (bounce == 1 && blockIdx.x == 0 && tid == 0) { clk0 = __builtin_amdgcn_s_memtime(); rt0 = __builtin_amdgcn_s_memrealtime(); }
    const unsigned long long span0 = (DEBUG_SPAN && bounce == 1 && tid == 0) ? __builtin_amdgcn_s_memrealtime() : 0ull;
    const RayPool in = p.pool[bounce & 1];
    const RayPool out = p.pool[(bounce + 1) & 1];
    const bool last = (bounce == p.depth - 1);
    // Up to MAXSLOT consecutive iterations are in flight in one launch sequence; a ray carries its iteration
    // slot in the top bits of its pixel word.  The per-(iteration, bounce) stream keys are wave-uniform.
    // The per-(iteration, bounce) stream keys sit in an LDS table indexed by the slot: one ds_read per ray instead of a
    // compare/select chain over the slots (v_cmp / v_cndmask issue at half rate).
    __shared__ uint32_t s_key[MAXSLOT], s_key_cam[MAXSLOT];
    if (tid < MAXSLOT) {
        s_key[tid] = stream_key(iter + (uint32_t)tid, (uint32_t)bounce + 1u, p.seed);
        if (FIRST || MOTION) s_key_cam[tid] = stream_key(iter + (uint32_t)tid, 0u, p.seed);
    }
    if (RESIDENT) {
        for (int k = tid; k < p.depth * MAXSLOT; k += WG)
            s_keys[k] = stream_key(iter + (uint32_t)(k & (MAXSLOT - 1)), (uint32_t)(k / MAXSLOT) + 1u, p.seed);
        if (tid < p.depth) s_term[tid] = 0u;
    }
    const uint32_t npix = (uint32_t)p.npix;

    // Input: the live rays of this bounce sit in up to NSHARD dense segments of the pool (one per reservation
    // counter).  A wave works on 64-ray chunks; chunk -> (segment, offset) is wave-uniform scalar arithmetic.
    // The per-segment ray counts sit in a small LDS table; a wave visits its chunks in increasing order, so it keeps a
    // cursor (segment, first chunk of it, rays in it) in scalar registers and only touches the table when it crosses
    // into the next segment.
    __shared__ uint32_t s_segn[NSHARD];
    if (tid < NSHARD) {
        uint32_t ns = 0;
        if (FIRST || COMPACT == 0) ns = (tid == 0) ? (uint32_t)p.npix * st->nslot : 0u;
        else if (tid < p.nshard) ns = st->counts[cnt_index(bounce, tid)];
        s_segn[tid] = ns;
    }
    __syncthreads();
    uint32_t total_chunks = 0;
#pragma unroll
    for (int k = 0; k < NSHARD; ++k) total_chunks += ((uint32_t)__builtin_amdgcn_readfirstlane((int)s_segn[k]) + 63u) >> 6;
    struct Cursor { uint32_t sh, c0, nseg; };
    // the segment this wave (COMPACT 1) / workgroup (COMPACT 2) appends its survivors to
    const uint32_t gwave = blockIdx.x * NW + wave;
    const uint32_t myshard = (COMPACT == 1 ? gwave : blockIdx.x) & (uint32_t)(p.nshard - 1);
    uint32_t *const out_counter = &st->counts[cnt_index(bounce + 1, myshard)];
    const uint32_t out_base = myshard * p.segcap;

    uint32_t live_count = 0;      // COMPACT 0: rays this wave found alive on entry
    uint32_t shadow_count = 0;    // NEE: shadow rays this wave traced
    int round = 0;
    unsigned long long ph[5] = {0, 0, 0, 0, 0}, ph5 = 0, ph6 = 0;
    // DEBUG_PHASE2: the lane budget of the later bounces -- per phase of a trip the shader clocks it took and the same weighted by
    // the lanes that had work in it (IterState::lane_budget, printed by pt_get_stats under PT_DEBUG_PHASE2=1)
    unsigned long long lbud[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    uint32_t lb_nvalid = 0, lb_nhit = 0, lb_nbsdf = 0;
    // pool slot of this lane in the chunk workgroup-round R2 gives this wave, and whether there is a ray in it
    auto locate = [&](Cursor &cu, uint32_t chunk, uint32_t &slot_i) -> bool {
        while (cu.sh + 1u < (uint32_t)NSHARD && chunk >= cu.c0 + ((cu.nseg + 63u) >> 6)) {
            cu.c0 += (cu.nseg + 63u) >> 6;
            cu.sh += 1u;
            cu.nseg = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_segn[cu.sh]);
        }
        const uint32_t idx = (chunk - cu.c0) * 64u + lane;       // index inside the segment
        slot_i = cu.sh * p.segcap + idx;                          // pool slot (FIRST / COMPACT 0: sh == 0, i == idx)
        return chunk < total_chunks && idx < cu.nseg;
    };
    Cursor cur = {0u, 0u, (uint32_t)__builtin_amdgcn_readfirstlane((int)s_segn[0])};
    // RESIDENT: chunks are DRAWN, not dealt.  With equal shares a launch's workgroups do not finish together -- the SIMDs favour
    // their oldest waves: lifetimes of 1.10 .. 2.10 ms in a 2.10 ms launch, profiles/r04/workgroup_lifetimes.txt -- and every CU
    // runs out the last third of the launch with ever fewer waves.  NSHARD draw counters (IterState::draw, zeroed by
    // k_iter_begin), counter k over chunks [k*Q, (k+1)*Q): a wave draws from the counter it starts with until that range is
    // used up, then from the next one that is not (one 32-lane read of all counters), and ends when none is left -- so the
    // waves of a launch run out of pool together.  The next draw is in flight while the current chunk is consumed (lane 0's
    // returning atomic, read a trip or more later).  What a wave keeps between trips is three scalars: the next pool slot to
    // hand out, the end of the chunk in hand, and a word of flags (its counter, "a draw is in flight", "the pool has more");
    // chunk -> (segment, offset) goes through two 32-entry LDS tables (the batched walk has no scalar registers to spare).
    // (The camera launch keeps its dealt chunks: with drawn ones its workgroups leave together too, and the OTHER launch sequence's
    // launch, whose workgroups take the slots the leaving ones free, can no longer start inside it -- 51.2 -> 45.3 G ray-bounces/s
    // on config 2 with two sequences, + 1.5 % with one: profiles/r04/ab_camera_drawn_chunks.txt.)
    uint32_t *const draw_ctr = st->draw;
    __shared__ uint32_t s_cfirst[NSHARD];            // first chunk of each segment (chunks are numbered segment by segment)
    __shared__ uint32_t s_draw[2];                   // chunks in the pool, chunks per draw counter
    if (RESIDENT) {
        if (tid < NSHARD) {
            uint32_t cf = 0;
            for (int k = 0; k < tid; ++k) cf += (s_segn[k] + 63u) >> 6;
            s_cfirst[tid] = cf;
        }
        if (tid == 0) { s_draw[0] = total_chunks; s_draw[1] = (total_chunks + (uint32_t)NSHARD - 1u) / (uint32_t)NSHARD; }
        __syncthreads();
    }
    uint32_t cpos = 0u, cend = 0u;                   // RESIDENT: next pool slot to hand out / end of the chunk in hand (wave-uniform)
    enum : uint32_t { WF_SHARD = 31u, WF_DRAWN = 32u, WF_MORE = 64u };
    uint32_t wflags = (gwave & WF_SHARD) | WF_MORE;  // RESIDENT, wave-uniform
    uint32_t drawn = 0u;                             // RESIDENT, lane 0: the draw in flight
    // (the batched walk sits at its register limit -- 80 VGPRs: three 512-thread workgroups per CU -- and has no register for a
    // draw in flight across the nearest-hit search: it draws when it needs a chunk, every third trip or so, and the other waves
    // cover that round trip)
    constexpr bool DRAW_AHEAD = true;
    // the next chunk of the pool for this wave (RESIDENT: sets cpos / cend); false = the pool is used up
    auto draw_chunk = [&](uint32_t &chunk_out) -> bool {
        const uint32_t total = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_draw[0]);
        const uint32_t Q = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_draw[1]);
        // chunks of counter `lane`'s range (lanes 0 .. NSHARD - 1)
        const uint32_t lo_l = (uint32_t)lane * Q, hi_l = (lo_l + Q < total) ? lo_l + Q : total;
        const uint32_t size_l = (lane < NSHARD && hi_l > lo_l) ? hi_l - lo_l : 0u;
        for (;;) {
            const uint32_t dshard = wflags & WF_SHARD;
            if ((wflags & WF_DRAWN) == 0u && lane == 0) drawn = atomicAdd(&draw_ctr[dshard * (uint32_t)CNT_STRIDE], 1u);
            const uint32_t idx = (uint32_t)__builtin_amdgcn_readfirstlane((int)drawn);
            wflags &= ~WF_DRAWN;
            const uint32_t size = (uint32_t)__builtin_amdgcn_readlane((int)size_l, (int)dshard);
            if (idx < size) {
                const uint32_t chunk = dshard * Q + idx;
                if (DRAW_AHEAD) {
                    if (lane == 0) drawn = atomicAdd(&draw_ctr[dshard * (uint32_t)CNT_STRIDE], 1u);  // the next one, used a trip or more later
                    wflags |= WF_DRAWN;
                }
                chunk_out = chunk;
                if (!RESIDENT) return true;
                // chunk -> segment: the last segment that starts at or before it (an empty segment starts where the next one does)
                const uint32_t cf = lane < NSHARD ? s_cfirst[lane] : 0xFFFFFFFFu;
                const uint32_t sh = (uint32_t)__popcll(__ballot(cf <= chunk)) - 1u;
                const uint32_t first = (chunk - (uint32_t)__builtin_amdgcn_readlane((int)cf, (int)sh)) * 64u;
                const uint32_t nseg = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_segn[sh]);
                cpos = sh * p.segcap + first;
                cend = cpos + (nseg - first < 64u ? nseg - first : 64u);
                return true;
            }
            // this range is used up: which ones are not?  (The counters only grow: a stale value can show a used-up range as
            // open -- the draw then says so --, never an open one as used up.)
            const uint32_t seen = lane < NSHARD ? __hip_atomic_load(&draw_ctr[(uint32_t)lane * (uint32_t)CNT_STRIDE], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0xFFFFFFFFu;
            const uint32_t open = (uint32_t)__ballot(seen < size_l);
            if (open == 0u) return false;
            const uint32_t rot = (dshard + 1u) & WF_SHARD;
            const uint32_t m = (open >> rot) | (open << ((32u - rot) & 31u));    // the open ranges from `rot` on, cyclically
            wflags = (wflags & ~WF_SHARD) | ((rot + (uint32_t)__builtin_ctz(m)) & WF_SHARD);
        }
    };
    // RESIDENT: the path a lane carries from one trip of the loop to the next (else: this trip's ray)
    bool valid = false;
    f3 o = mk(0, 0, 0), d = mk(0, 0, 0), T = mk(1, 1, 1);
    uint32_t pix = 0;
    uint32_t lb = (uint32_t)bounce;                  // RESIDENT, per lane: the bounce the lane's path is at (bits 0..7) | (1 + the primitive its
                                                     // ray just left on the outside and cannot meet again) << 8  (0: none)
    for (uint32_t R = blockIdx.x; RESIDENT || R * NW < total_chunks; R += RESIDENT ? 0u : gridDim.x, ++round) {     // (dealt chunks: workgroup-uniform trip count)
        const uint32_t my_chunk = R * NW + wave;                  // dealt: round-robin over the workgroups' waves
        const unsigned long long tc0 = (DEBUG_PHASE || PT_DEBUG_PHASE == 2 || PT_DEBUG_PHASE == 3) ? __builtin_amdgcn_s_memtime() : 0ull;
        uint32_t i = 0;
        bool in_pool = false;
        if (RESIDENT) {
            // refill: lanes without a path take the next rays of the wave's chunks, in pool order (consecutive addresses,
            // whichever lanes are free), once `refill_min` lanes are free -- or none is busy
            uint64_t vm = __ballot(valid);
            uint32_t nfree = 64u - (uint32_t)__popcll(vm);
            if ((wflags & WF_MORE) != 0u && (nfree >= (uint32_t)p.refill_min || vm == 0ull)) {
                for (;;) {
                    uint32_t dchunk;
                    if (cpos == cend && !draw_chunk(dchunk)) { wflags &= ~WF_MORE; break; }      // (also the first time: 0 == 0)
                    const uint64_t fm = ~vm;
                    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(fm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)fm, 0u));
                    const uint32_t avail = cend - cpos;
                    if (!valid && rank < avail) {
                        uint32_t k = cpos + rank;
                        if (!dbgInRange(p, 1, k, (unsigned long long)p.segcap * (unsigned long long)p.nshard)) k = 0u;
                        const v2f c_ = nt_load(reinterpret_cast<const v2f *>(&in.c[k]));
                        const v4f a_ = nt_load(reinterpret_cast<const v4f *>(&in.a[k]));
                        const v4f b_ = nt_load(reinterpret_cast<const v4f *>(&in.b[k]));
                        pix = __float_as_uint(c_.y);
                        o = mk(a_.x, a_.y, a_.z);
                        d = mk(a_.w, b_.x, b_.y);
                        T = mk(b_.z, b_.w, c_.x);
                        lb = (uint32_t)bounce;
                        valid = true;
                    }
                    const uint32_t took = nfree < avail ? nfree : avail;
                    cpos += took;
                    nfree -= took;
                    if (nfree == 0u) break;
                    vm = __ballot(valid);
                }
            }
            if (__ballot(valid) == 0ull) break;                   // pool drained and every path of the wave has ended
        } else {
            valid = locate(cur, my_chunk, i);
            in_pool = valid;                                      // the slot exists (COMPACT 0: it may hold a dead ray)
            o = mk(0, 0, 0); d = mk(0, 0, 0); T = mk(1, 1, 1);
            pix = 0;
        }
        const uint32_t chunk_first_ray = my_chunk * 64u;          // (bounce 0: rays are numbered slot by slot, pixel by pixel)
        const uint32_t cb = RESIDENT ? (lb & 0xFFu) : (uint32_t)bounce;            // this trip's bounce (RESIDENT: per lane)
        const bool lastb = RESIDENT ? (cb == (uint32_t)(p.depth - 1)) : last;
        MotionTime mt;
        mt.k = 0u;
        mt.f = 0.0f;
        if (RESIDENT) {
        } else if (FIRST) {
            if (valid) {
                // raycastFromCameraKernel: jittered pinhole ray through tile-local pixel pl of iteration slot
                // With at least 64 pixels per slot, 64 consecutive ray indices meet at most one slot boundary: the slot
                // of the chunk's first ray on the scalar unit, one compare per lane for the step.  Tiny tiles (a chunk
                // spans several slots) take the general count.
                uint32_t slot = 0;
                if (npix >= 64u) {
                    const uint32_t base = (uint32_t)__builtin_amdgcn_readfirstlane((int)(i - (uint32_t)lane));
                    uint32_t slot_lo = 0;
#pragma unroll
                    for (uint32_t k = 1; k < (uint32_t)MAXSLOT; ++k) slot_lo += (base >= k * npix) ? 1u : 0u;
                    slot = slot_lo + ((i >= (slot_lo + 1u) * npix) ? 1u : 0u);
                } else {
#pragma unroll
                    for (uint32_t k = 1; k < (uint32_t)MAXSLOT; ++k) slot += (i >= k * npix) ? 1u : 0u;
                }
                const uint32_t pl = i - slot * npix;
                pix = pl | (slot << SLOT_SHIFT);
                const uint32_t gp = globalPixel(p, pl);
                const uint32_t y = (uint32_t)(((unsigned long long)gp * p.w_magic) >> p.w_shift);   // gp / W
                const uint32_t x = gp - y * (uint32_t)p.W;
                const uint32_t kc = s_key_cam[slot];
                uint32_t s = minstd_seed(wang_hash(gp ^ kc));
                s = minstd_next(s);
                const float jx = u01_of(s);
                s = minstd_next(s);
                const float jy = u01_of(s);
                const float sx = ((float)x + jx) / p.resx;
                const float sy = ((float)y + jy) / p.resy;
                f3 eye = mk(p.eye[0], p.eye[1], p.eye[2]);
                f3 cM = mk(p.M[0], p.M[1], p.M[2]), cH = mk(p.H[0], p.H[1], p.H[2]), cV = mk(p.V[0], p.V[1], p.V[2]);
                f3 cA = mk(p.A[0], p.A[1], p.A[2]), cB = mk(p.B[0], p.B[1], p.B[2]), cvn = mk(p.vn[0], p.vn[1], p.vn[2]);
                if (MOTION) {
                    // the path's shutter time (third draw of the camera stream) and the camera basis at that time, with
                    // the host's operation order (configure(): camera basis)
                    s = minstd_next(s);
                    mt = motionTime(p, u01_of(s));
                    const float4 *ca = reinterpret_cast<const float4 *>(p.knot_cam) + mt.k * 3u, *cb = ca + 3;
                    auto lerp4 = [&](float4 a, float4 b) { return mk(a.x + (b.x - a.x) * mt.f, a.y + (b.y - a.y) * mt.f, a.z + (b.z - a.z) * mt.f); };
                    eye = lerp4(ca[0], cb[0]);
                    const f3 view = lerp4(ca[1], cb[1]), up = lerp4(ca[2], cb[2]);
                    cA = normalize(cross(view, up));
                    cB = normalize(cross(cA, view));
                    const float lenV = length(view);
                    cM = eye + view;
                    cH = (lenV * p.tan_x) * cA;
                    cV = (lenV * p.tan_y) * cB;
                    cvn = normalize(view);
                }
                const f3 P = (cM + (1.0f - 2.0f * sx) * cH) + (1.0f - 2.0f * sy) * cV;
                o = eye;
                d = normalize(P - eye);
                if (p.lens_radius > 0.0f) {
                    // thin lens (depth of field): the pinhole ray fixes the point in focus; start on the lens disc
                    s = minstd_next(s);
                    const float u1 = u01_of(s);
                    s = minstd_next(s);
                    const float u2 = u01_of(s);
                    const float tf = p.focal_distance / dot(d, cvn);
                    const f3 Pf = eye + tf * d;
                    const float rr = p.lens_radius * sqrt_rn(u1);
                    const float around = (float)((double)u2 * 6.2831853071795864769252867665590057683943);
                    float sn, cs;
                    sincos_poly(around, sn, cs);
                    o = eye + ((rr * cs) * cA + (rr * sn) * cB);
                    d = normalize(Pf - o);
                }
            }
        } else {
            if (valid && !dbgInRange(p, 1, i, (unsigned long long)p.segcap * (unsigned long long)p.nshard)) valid = false;
            if (valid) {
                const v2f c_ = nt_load(reinterpret_cast<const v2f *>(&in.c[i]));
                const float2 c = make_float2(c_.x, c_.y);
                pix = __float_as_uint(c.y);
                if (COMPACT == 0 && pix == DEAD) valid = false;
                if (valid) {
                    const v4f a_ = nt_load(reinterpret_cast<const v4f *>(&in.a[i]));
                    const v4f b_ = nt_load(reinterpret_cast<const v4f *>(&in.b[i]));
                    const float4 a = make_float4(a_.x, a_.y, a_.z, a_.w), b = make_float4(b_.x, b_.y, b_.z, b_.w);
                    o = mk(a.x, a.y, a.z);
                    d = mk(a.w, b.x, b.y);
                    T = mk(b.z, b.w, c.x);
                    if (MOTION) {
                        // the path's shutter time again: third draw of its camera stream (nothing is stored in the ray)
                        const uint32_t sc = minstd_seed(wang_hash(globalPixel(p, pix & PIX_MASK) ^ s_key_cam[(pix >> SLOT_SHIFT) & (uint32_t)(MAXSLOT - 1)]));
                        mt = motionTime(p, u01_of(minstd_jump(sc, MINSTD_A3)));
                    }
                }
            }
        }
        if (COMPACT == 0) live_count += (uint32_t)__popcll(__ballot(valid));

        bool alive = false;
        bool leaves_outside = false;      // RESIDENT: the new ray starts on the OUTSIDE of the primitive it was shaded on, moving away from it
        const unsigned long long c1 = (DEBUG_PHASE || PT_DEBUG_PHASE == 2 || PT_DEBUG_PHASE == 3) ? __builtin_amdgcn_s_memtime() : 0ull;
        // camera rays share the eye (host-side eye transforms and eye-relative boxes) unless a lens spreads their origins
        // camera rays of the pair path: which primitives the chunk's 64 pixels can see (host-built table, one word per span
        // of 64 tile-local pixels, no lens).  A chunk is one span when npix % 64 == 0; else it lies across two of them
        // and, where it runs over the end of an iteration slot, across the tile's first span as well.
        uint32_t primmask = 0xFFFFFFFFu;
        if (FIRST && GEOM == GEOM_PAIR && p.span_mask != nullptr) {
            const_u32_ptr sm = (const_u32_ptr)(uintptr_t)p.span_mask;
            if (npix <= 64u) primmask = sm[0];                            // one span is the whole tile
            else {
                uint32_t b0 = chunk_first_ray;                            // wave-uniform
                while (b0 >= npix) b0 -= npix;                           // its tile-local pixel (at most MAXSLOT - 1 trips)
                const uint32_t e = b0 + 63u;
                primmask = sm[b0 >> 6] | sm[(e < npix ? e : npix - 1u) >> 6];
                if (e >= npix) primmask |= sm[0];
            }
        }
        if (FIRST && (GEOM == GEOM_WALK4 || GEOM == GEOM_WALK4G) && p.span_off != nullptr) {
            uint32_t b0 = chunk_first_ray;
            while (b0 >= npix) b0 -= npix;
            primmask = b0 >> 6;                                           // (the batched walks take the span's number)
        }
        Hit h;
        if (MOTION && GEOM == GEOM_PAIR) {
            // pair path: per-lane pre-test against the swept boxes, exact tests in full batches with per-pair interpolated rows
            h = nearestHit<GEOM, false, true>(p, s_prims, s_nodes, wq, o, d, valid, (uint32_t)lane, 0xFFFFFFFFu, mt);
        } else if (MOTION) {
            h.any = false; h.material = 0; h.prim = 0; h.t = 0.0f; h.p = mk(0, 0, 0); h.n = mk(0, 0, 0);
            if (valid) h = nearestHitMotion(p, o, d, mt);
        } else {
            if (RESIDENT) h = nearestHit<GEOM, false, false, true, SLAB>(p, s_prims, s_nodes, wq, o, d, valid, (uint32_t)lane, 0xFFFFFFFFu, MotionTime(), (lb >> 8) - 1u);
            else h = (FIRST && p.lens_radius > 0.0f) ? nearestHit<GEOM, false, false, false, SLAB>(p, s_prims, s_nodes, wq, o, d, valid, (uint32_t)lane)
                                                     : nearestHit<GEOM, FIRST, false, false, SLAB>(p, s_prims, s_nodes, wq, o, d, valid, (uint32_t)lane, primmask);
        }
        const unsigned long long c2 = (DEBUG_PHASE || PT_DEBUG_PHASE == 2 || PT_DEBUG_PHASE == 3) ? __builtin_amdgcn_s_memtime() : 0ull;
        if (LANE_BUDGET<FIRST>()) { lb_nvalid = (uint32_t)__popcll(__ballot(valid)); lb_nhit = (uint32_t)__popcll(__ballot(valid && h.any)); }
        bool did_bsdf = false;            // (DEBUG_PHASE2)
        f3 L = mk(0, 0, 0);               // radiance this vertex adds to the path's sample
        // direct lighting: the shadow ray this lane wants traced and what it is worth if the light is visible
        bool want_shadow = false;
        f3 so = mk(0, 0, 0), sd = mk(0, 0, 0), Ld = mk(0, 0, 0);
        uint32_t lprim = 0;
        float ldist = 0.0f;
        if (valid) {
            if (h.any) {
                dbgInRange(p, 8, h.prim, (unsigned long long)p.nG);
                dbgInRange(p, 9, h.material, (unsigned long long)p.nM);
                const uint32_t m = h.material;
                const float emit = s_mats[M_EMIT * p.nM + m];
                if (emit > 0.0f) {
                    // (direct lighting: a light the previous vertex could have sampled adds nothing when hit by chance;
                    // an emitter outside the light table -- Prim::area is 0 for those -- still counts)
                    bool counts = true;
                    if (NEE && (pix >> 31) != 0u) counts = !((PRIMS_IN_LDS ? s_prims[h.prim].area : p.prims[h.prim].area) > 0.0f);
                    if (counts) {
                        const f3 col = mk(s_mats[M_CR * p.nM + m], s_mats[M_CG * p.nM + m], s_mats[M_CB * p.nM + m]);
                        L = emit * (T * col);
                    }
                } else if (!lastb || NEE) {
                    // calculateBSDF: pick the lobe, build the next ray
                    did_bsdf = true;
                    const uint32_t slot = NEE ? ((pix >> SLOT_SHIFT) & (uint32_t)(MAXSLOT - 1)) : (pix >> SLOT_SHIFT);
                    const uint32_t kb = RESIDENT ? s_keys[cb * (uint32_t)MAXSLOT + slot] : s_key[slot];
                    // the bounce's draws in stream order u_select, xi1, xi2, u_rr, then (light sampling) u_light, u_seed or
                    // (inside a medium) u_sd, u_s2, u_s3: each by its own jump from the seed, computed where it is used
                    const uint32_t s0 = minstd_seed(wang_hash(globalPixel(p, pix & PIX_MASK) ^ kb));

                    const float ndotd = dot(h.n, d);
                    const bool backside = ndotd > 0.0f;
                    const f3 nf = backside ? -h.n : h.n;
                    const float refr = s_mats[M_REFR * p.nM + m];
                    const float refl = s_mats[M_REFL * p.nM + m];
                    // subsurface scattering (pt_options.scatter): a SCATTER material that is not a mirror encloses a medium
                    const bool medium = MEDIA && s_mats[M_SCAT * p.nM + m] > 0.0f && !(refl > 0.0f);
                    const bool diffuse = !(refr > 0.0f) && !(refl > 0.0f) && !medium;
                    if (NEE && diffuse) {
                        // one light, one point on it (the reference's float-seeded samplers), one shadow ray;
                        // estimator T*c/pi * Le * cos_x cos_y / d^2 * (area * number of lights)
                        const float u_light = u01_of(minstd_jump(s0, MINSTD_A5));
                        const float u_seed = u01_of(minstd_jump(s0, MINSTD_A6));
                        int j = (int)(u_light * (float)p.nlights);
                        if (j > p.nlights - 1) j = p.nlights - 1;
                        const int4 lr = *reinterpret_cast<const int4 *>(p.lights + j);      // prim, tri_first, tri_count, area
                        const float larea = __int_as_float(lr.w);
                        float larea_m = larea;                         // (FEAT_MOTION: the light's area at the path's time)
                        so = h.p + 0.0002f * nf;
                        f3 yl, nl;
                        uint32_t lmat;
                        if (lr.z > 0) {
                            // mesh light: the float-seeded engine draws the triangle (first one whose running area exceeds
                            // u_t * total, else the last) and a uniform point on it; the normal faces the shading point
                            uint32_t rng = minstd_seed(wang_hash((uint32_t)(u_seed * 16777216.0f)));
                            const float u_t = uniform_real(rng, 0, 1), u_a = uniform_real(rng, 0, 1), u_b = uniform_real(rng, 0, 1);
                            const float *cdf = p.light_cdf + lr.y;
                            const float target = u_t * larea;
                            int lo = 0, hi = lr.z - 1;
                            while (lo < hi) { const int mid = (lo + hi) >> 1; if (cdf[mid] > target) hi = mid; else lo = mid + 1; }
                            lprim = (uint32_t)(p.ngeoms + p.light_tris[lr.y + lo]);
                            const Prim *LP = &p.prims[lprim];                           // (triangles never sit in the LDS copies)
                            const float4 *tw = reinterpret_cast<const float4 *>(LP->inv);
                            const float4 t0 = tw[0], t1 = tw[1], t2 = tw[2];           // v0.xyz e1.x | e1.yz e2.xy | e2.z ...
                            const f3 v0 = mk(t0.x, t0.y, t0.z), e1 = mk(t0.w, t1.x, t1.y), e2 = mk(t1.z, t1.w, t2.x);
                            yl = sampleTriangle(v0, e1, e2, u_a, u_b);
                            nl = mk(LP->fwd[0], LP->fwd[1], LP->fwd[2]);
                            if (dot(nl, yl - so) > 0.0f) nl = -nl;
                            lmat = LP->material;
                        } else if (MOTION) {
                            // the light where it is at this path's time: rows interpolated between the knots, area from them
                            // (getRadiuses' half-extents, the oracle's o_lightArea)
                            lprim = (uint32_t)lr.x;
                            const uint32_t ltype = p.prims[lprim].type;
                            float linv[12], lfwd[12];
                            motionRows(p, lprim, mt, linv, lfwd);
                            sampleLight(ltype, lfwd, mk(lfwd[3], lfwd[7], lfwd[11]), u_seed * 16777216.0f, yl, nl);
                            const f3 rad = getRadiuses(lfwd);
                            if (ltype == 1u) {
                                const float side1 = rad.x * rad.y * 4.0f, side2 = rad.z * rad.y * 4.0f, side3 = rad.x * rad.z * 4.0f;
                                larea_m = 2.0f * (side1 + side2 + side3);
                            } else {
                                larea_m = 4.18879020478639098f * ((rad.x * rad.y + rad.x * rad.z) + rad.y * rad.z);
                            }
                            lmat = p.prims[lprim].material;
                        } else {
                        lprim = (uint32_t)lr.x;
                        const Prim *LP = PRIMS_IN_LDS ? &s_prims[lprim] : &p.prims[lprim];
                        const uint4 hd = *reinterpret_cast<const uint4 *>(LP);           // type, material, area, pad
                        const float4 *fw = reinterpret_cast<const float4 *>(LP->fwd);
                        const float4 f0 = fw[0], f1 = fw[1], f2 = fw[2], cc = fw[3];   // fwd rows, (cx, cy, cz, bound)
                        const float fwd[12] = {f0.x, f0.y, f0.z, f0.w, f1.x, f1.y, f1.z, f1.w, f2.x, f2.y, f2.z, f2.w};
                        lmat = hd.y;
                        if (hd.x == 1u) {                            // cube light: thresholds and face normals from the table
                            const float4 *tab = PRIMS_IN_LDS ? reinterpret_cast<const float4 *>(s_prims + p.nG) + lprim * 9u
                                                             : reinterpret_cast<const float4 *>(p.face_n) + lprim * 8u;
                            sampleCubeLightTab(fwd, tab, u_seed * 16777216.0f, yl, nl);
                        } else {
                            sampleLight(hd.x, fwd, mk(cc.x, cc.y, cc.z), u_seed * 16777216.0f, yl, nl);
                        }
                        }
                        const f3 wi = yl - so;
                        const float d2 = dot(wi, wi);
                        ldist = sqrt_rn(d2);
                        sd = normalize(wi);
                        const float cx = dot(nf, sd), cy = -dot(nl, sd);
                        if (cx > 0.0f && cy > 0.0f) {
                            want_shadow = true;
                            const float G = (cx * cy) / d2;
                            const float wgt = (G * (larea_m * (float)p.nlights)) * 0.318309886f;
                            const uint32_t lm = lmat;
                            const f3 col = mk(s_mats[M_CR * p.nM + m], s_mats[M_CG * p.nM + m], s_mats[M_CB * p.nM + m]);
                            const f3 lcol = mk(s_mats[M_CR * p.nM + lm], s_mats[M_CG * p.nM + lm], s_mats[M_CB * p.nM + lm]);
                            Ld = wgt * ((T * col) * (s_mats[M_EMIT * p.nM + lm] * lcol));
                        }
                    }
                    if (!lastb) {
                    f3 nd;
                    f3 bias_n = nf;
                    float bias = 0.0002f;                 // RAY_BIAS_AMOUNT, ref: src/utilities.h:26
                    bool scattered = false, pass_through = false, transmitted = false;
                    if (medium) {
                        if (backside) {
                            // the segment ran through the medium: three more draws of the bounce's stream decide whether
                            // the path scatters before the boundary or reaches it (calculateScatterAndAbsorption)
                            const float u_sd = u01_of(minstd_jump(s0, MINSTD_A5));
                            const float u_s2 = u01_of(minstd_jump(s0, MINSTD_A6));
                            const float u_s3 = u01_of(minstd_jump(s0, MINSTD_A7));
                            const f3 sa = mk(s_mats[M_AR * p.nM + m], s_mats[M_AG * p.nM + m], s_mats[M_AB * p.nM + m]);
                            f3 mo = o, md = d;
                            float seg = h.t;
                            scattered = calculateScatterAndAbsorption(mo, md, seg, sa, s_mats[M_RSCT * p.nM + m], T, u_sd, u_s2, u_s3);
                            if (scattered) { o = mo; nd = md; }
                        }
                        if (!scattered && !(refr > 0.0f)) {
                            // index-matched boundary: the ray goes straight on; entering picks up the surface colour once
                            if (!backside) T = T * mk(s_mats[M_CR * p.nM + m], s_mats[M_CG * p.nM + m], s_mats[M_CB * p.nM + m]);
                            pass_through = true;
                        }
                    }
                    if (scattered) {
                        // the new ray starts inside the medium, where the walk left it
                    } else if (pass_through) {
                        nd = d;
                        bias_n = -nf;
                        f3 v;
                        if (MOTION) {
                            float minv[12], mfwd[12];
                            motionRows(p, h.prim, mt, minv, mfwd);
                            v = mulMV(minv, d, 0.0f);
                        } else {
                        const Prim *HP = PRIMS_IN_LDS ? &s_prims[h.prim] : &p.prims[h.prim];
                        const float4 *iv = reinterpret_cast<const float4 *>(HP->inv);
                        const float4 i0 = iv[0], i1 = iv[1], i2 = iv[2];
                        const float inv[12] = {i0.x, i0.y, i0.z, i0.w, i1.x, i1.y, i1.z, i1.w, i2.x, i2.y, i2.z, i2.w};
                        v = (HP->type == 3u) ? d : mulMV(inv, d, 0.0f);      // a triangle is tested in world space
                        }
                        bias = 0.0002f + 1e-4f * rsqrt_rn(dot(v, v));
                    } else if (refr > 0.0f) {
                        const float ior = s_mats[M_IOR * p.nM + m];
                        const float n1 = backside ? ior : 1.0f;
                        const float n2 = backside ? 1.0f : ior;
                        const f3 rdir = reflectionDirection(nf, d);
                        bool tir;
                        const f3 tdir = transmissionDirection(nf, d, n1, n2, tir);
                        const float Rf = fresnelReflectance(nf, d, n1, n2, tdir);
                        if (p.absorption && backside && !medium) {
                            // the segment that ends on the inner side of the surface ran through the medium
                            const f3 sa = mk(s_mats[M_AR * p.nM + m], s_mats[M_AG * p.nM + m], s_mats[M_AB * p.nM + m]);
                            if (sa.x != 0.0f || sa.y != 0.0f || sa.z != 0.0f) T = T * calculateTransmission(sa, h.t);
                        }
                        T = T * mk(s_mats[M_SR * p.nM + m], s_mats[M_SG * p.nM + m], s_mats[M_SB * p.nM + m]);
                        if (u01_of(minstd_jump(s0, MINSTD_A1)) < Rf) nd = rdir;      // u_select
                        else {
                            // transmitted: start beyond the surface.  h.p was pulled back by getPointOnRay's 1e-4
                            // object-space epsilon (ref: src/intersections.h:46-48) = 1e-4/|inverseTransform*d| in
                            // world units, more than the bias for objects scaled by > 2
                            nd = tdir;
                            bias_n = -nf;
                            transmitted = true;
                            f3 v;
                            if (MOTION) {
                                float minv[12], mfwd[12];
                                motionRows(p, h.prim, mt, minv, mfwd);
                                v = mulMV(minv, d, 0.0f);
                            } else {
                            const Prim *HP = PRIMS_IN_LDS ? &s_prims[h.prim] : &p.prims[h.prim];
                            const float4 *iv = reinterpret_cast<const float4 *>(HP->inv);
                            const float4 i0 = iv[0], i1 = iv[1], i2 = iv[2];
                            const float inv[12] = {i0.x, i0.y, i0.z, i0.w, i1.x, i1.y, i1.z, i1.w, i2.x, i2.y, i2.z, i2.w};
                            v = (HP->type == 3u) ? d : mulMV(inv, d, 0.0f);  // a triangle is tested in world space
                            }
                            bias = 0.0002f + 1e-4f * rsqrt_rn(dot(v, v));
                        }
                    } else if (refl > 0.0f) {
                        T = T * mk(s_mats[M_SR * p.nM + m], s_mats[M_SG * p.nM + m], s_mats[M_SB * p.nM + m]);
                        nd = reflectionDirection(nf, d);
                    } else {
                        T = T * mk(s_mats[M_CR * p.nM + m], s_mats[M_CG * p.nM + m], s_mats[M_CB * p.nM + m]);
                        nd = randomDirectionInHemisphere(nf, u01_of(minstd_jump(s0, MINSTD_A2)), u01_of(minstd_jump(s0, MINSTD_A3)));   // xi1, xi2
                    }
                    if (!scattered) o = h.p + bias * bias_n;
                    d = nd;
                    alive = true;
                    // the new ray starts 0.0002 outside the primitive, in the hemisphere of its outward normal (diffuse, mirror or Fresnel
                    // reflection off the OUTER side): the next bounce's search need not look at that primitive (where the host vouches
                    // for it: Prim::self_r2)
                    leaves_outside = RESIDENT && !scattered && !pass_through && !transmitted && !backside && h.self_ok != 0u;
                    if (p.rr_start >= 0 && (int)cb >= p.rr_start) {      // Russian roulette
                        float q = T.x;
                        if (T.y > q) q = T.y;
                        if (T.z > q) q = T.z;
                        q = (q < 0.05f) ? 0.05f : ((q > 1.0f) ? 1.0f : q);
                        if (u01_of(minstd_jump(s0, MINSTD_A4)) >= q) alive = false;      // u_rr
                        else T = mk(T.x / q, T.y / q, T.z / q);
                    }
                    if (NEE) pix = (pix & 0x7FFFFFFFu) | (diffuse ? 0x80000000u : 0u);
                    }
                }
            }
        }
        if (NEE) {
            const uint64_t wmask = __ballot(want_shadow);
            if (wmask != 0ull) {                              // wave-uniform
                shadow_count += (uint32_t)__popcll(wmask);
                Hit hs;
                if (MOTION && GEOM == GEOM_PAIR) {
                    hs = nearestHit<GEOM, false, true>(p, s_prims, s_nodes + (PAIR_BOX_BYTES / 16) * p.nG, wq, so, sd, want_shadow, (uint32_t)lane, 0xFFFFFFFFu, mt);
                } else if (MOTION) {
                    hs.any = false; hs.material = 0; hs.prim = 0; hs.t = 0.0f; hs.p = mk(0, 0, 0); hs.n = mk(0, 0, 0);
                    if (want_shadow) hs = nearestHitMotion(p, so, sd, mt);
                } else {
                    hs = nearestHit<GEOM, false>(p, s_prims, (GEOM == GEOM_PAIR) ? s_nodes + (PAIR_BOX_BYTES / 16) * p.nG : s_nodes, wq, so, sd, want_shadow, (uint32_t)lane);
                }
                if (want_shadow && hs.any && hs.prim == lprim) {
                    const float tol = 1e-3f * ((ldist > 1.0f) ? ldist : 1.0f);
                    if (fabsf(hs.t - ldist) <= tol) L = L + Ld;          // the sampled point itself is what the ray reached
                }
            }
            if (valid && (L.x != 0.0f || L.y != 0.0f || L.z != 0.0f)) {
                // the plane entry accumulates along the path (exclusive owner); an entry of another batch counts as zero
                float4 *lp = reinterpret_cast<float4 *>(p.lbuf) + ((size_t)((pix >> SLOT_SHIFT) & (uint32_t)(MAXSLOT - 1)) * npix + (size_t)(pix & PIX_MASK));
                if (!dbgInRange(p, 6, (unsigned long long)((pix >> SLOT_SHIFT) & (uint32_t)(MAXSLOT - 1)) * npix + (pix & PIX_MASK), (unsigned long long)npix * st->nslot)) lp = reinterpret_cast<float4 *>(p.lbuf);
                float4 e = *lp;
                if (__float_as_uint(e.w) != serial) e = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                *lp = make_float4(e.x + L.x, e.y + L.y, e.z + L.z, __uint_as_float(serial));
            }
        } else if (valid && !alive && (L.x != 0.0f || L.y != 0.0f || L.z != 0.0f)) {
            // The path ends here with a non-zero radiance sample: it goes to its iteration slot's plane, stamped with the batch's
            // serial number; k_accumulate folds the planes into the running mean in iteration order once the launch sequence is
            // done and takes every entry that does not carry this batch's serial for zero.  (Nineteen of twenty paths of the
            // benchmark scenes end with a zero sample -- they leave the scene or run out of bounces -- and round 3's sensitivity
            // runs priced the unconditional 12-byte store at 7-10 % of the kernel: profiles/r03/knockout.txt.)
            float4 *lp = reinterpret_cast<float4 *>(p.lbuf) + ((size_t)(pix >> SLOT_SHIFT) * npix + (size_t)(pix & PIX_MASK));
            if (!dbgInRange(p, 7, (unsigned long long)(pix >> SLOT_SHIFT) * npix + (pix & PIX_MASK), (unsigned long long)npix * st->nslot)) lp = reinterpret_cast<float4 *>(p.lbuf);
            nt_store((v4f){L.x, L.y, L.z, __uint_as_float(serial)}, reinterpret_cast<v4f *>(lp));
        }

        const unsigned long long c3 = (DEBUG_PHASE || PT_DEBUG_PHASE == 2 || PT_DEBUG_PHASE == 3) ? __builtin_amdgcn_s_memtime() : 0ull;
        if (DEBUG_PHASE) { ph[0] += c1 - tc0; ph[1] += c2 - c1; ph[2] += c3 - c2; ph[4] += 1; }
        if (LANE_BUDGET<FIRST>()) { ph[0] += c1 - tc0; ph[1] += c2 - c1; ph[2] += c3 - c2; ph[4] += 1; ph5 += h.dbg0 + h.dbg2; ph6 += h.dbg1; }
        if (LANE_BUDGET<FIRST>() && GEOM == GEOM_PAIR) {
            lb_nbsdf = (uint32_t)__popcll(__ballot(did_bsdf));
            const unsigned long long res = (c2 - c1) - h.dbg0 - h.dbg1 - h.dbg2;
            lbud[0] += c1 - tc0;  lbud[1] += (c1 - tc0) * lb_nvalid;          // load / refill
            lbud[2] += h.dbg0;    lbud[3] += h.dbg0 * lb_nvalid;              // pre-test loop
            lbud[4] += h.dbg2;                                                 // full batches: 64 lanes
            lbud[5] += h.dbg1;    lbud[6] += h.dbg3;                           // last batches
            lbud[7] += res;       lbud[8] += res * lb_nhit;                    // result (winner's normal, material)
            lbud[9] += c3 - c2;   lbud[10] += (c3 - c2) * lb_nbsdf;            // shading + radiance write
            lbud[13] += 1;        lbud[14] += lb_nvalid;
        }
        if (RESIDENT) {
            // a path that ended is counted at the bounce it ended at (the per-bounce live counts follow from the histogram);
            // one that goes on stays where it is, a bounce further
            if (valid && !alive) atomicAdd(&s_term[cb], 1u);
            valid = alive;
            lb = (cb + 1u) | (leaves_outside ? ((h.prim + 1u) << 8) : 0u);
            if (DEBUG_PHASE2 && GEOM == GEOM_PAIR) { const unsigned long long c4 = __builtin_amdgcn_s_memtime() - c3; lbud[11] += c4; lbud[12] += c4 * (unsigned long long)__popcll(__ballot(alive)); }
            continue;
        }
        if (last) continue;      // wave-uniform: nothing survives the last bounce

        if (COMPACT != 0) {
            // stream compaction: wave ballot/mbcnt prefix, then a reservation in this wave's / workgroup's segment
            const uint64_t mask = __ballot(alive);
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32),
                                                            __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
            const uint32_t wtot = (uint32_t)__popcll(mask);
            uint32_t dst;
            if (COMPACT == 1 || NW == 1) {
                uint32_t b = 0;
                if (lane == 0 && wtot) b = atomicAdd(out_counter, wtot);
                dst = (uint32_t)__builtin_amdgcn_readfirstlane((int)b) + rank;
            } else {
                const int par = round & 1;
                uint32_t *tot = s_scan + par * NW;
                uint32_t *gbase = s_scan + 2 * NW + par;
                if (lane == 0) tot[wave] = wtot;
                __syncthreads();
                if (tid == 0) {
                    uint32_t sum = 0;
#pragma unroll
                    for (int w = 0; w < NW; ++w) sum += tot[w];
                    *gbase = sum ? atomicAdd(out_counter, sum) : 0u;
                }
                __syncthreads();
                uint32_t off = *gbase;
                for (int w = 0; w < wave; ++w) off += tot[w];
                dst = off + rank;
            }
            dst += out_base;
            if (alive && !dbgInRange(p, 4, dst, (unsigned long long)p.segcap * (unsigned long long)p.nshard)) alive = false;
            if (alive && !dbgInRange(p, 5, dst - out_base, p.segcap)) alive = false;
            if (alive) {
                // (the pools are written once and read once, a launch later and gigabytes apart: streaming stores / loads)
                nt_store((v4f){o.x, o.y, o.z, d.x}, reinterpret_cast<v4f *>(&out.a[dst]));
                nt_store((v4f){d.y, d.z, T.x, T.y}, reinterpret_cast<v4f *>(&out.b[dst]));
                nt_store((v2f){T.z, __uint_as_float(pix)}, reinterpret_cast<v2f *>(&out.c[dst]));
            }
        } else {
            // no compaction (validation / ablation mode): the ray keeps slot i, dead slots are tagged
            if (in_pool) {
                if (alive) {
                    out.a[i] = make_float4(o.x, o.y, o.z, d.x);
                    out.b[i] = make_float4(d.y, d.z, T.x, T.y);
                    out.c[i] = make_float2(T.z, __uint_as_float(pix));
                } else {
                    out.c[i] = make_float2(0.0f, __uint_as_float(DEAD));
                }
            }
        }
        if (LANE_BUDGET<FIRST>()) {
            const unsigned long long c4 = __builtin_amdgcn_s_memtime() - c3;
            ph[3] += c4;
            if (GEOM == GEOM_PAIR) { lbud[11] += c4; lbud[12] += c4 * (unsigned long long)__popcll(__ballot(alive)); }      // compaction + pool write
        }
    }
    if (COMPACT == 0) {
        if (lane == 0 && live_count) atomicAdd(&st->counts[cnt_index(bounce, 0)], live_count);
    }
    if (RESIDENT) {
        // live rays entering bounce b = the paths that ended at b or later (every path that enters the launch ends in it)
        __syncthreads();
        if (tid > bounce && tid < p.depth) {
            uint32_t sum = 0;
            for (int e = tid; e < p.depth; ++e) sum += s_term[e];
            if (sum) atomicAdd(&st->counts[cnt_index(tid, (int)(blockIdx.x & (uint32_t)(NSHARD - 1)))], sum);
        }
    }
    if (NEE) {
        if (lane == 0 && shadow_count) atomicAdd(&st->shadow_rays, (unsigned long long)shadow_count);
    }
    if (LANE_BUDGET<FIRST>() && GEOM == GEOM_PAIR && lane == 0) {
#pragma unroll
        for (int k = 0; k < 16; ++k) atomicAdd(&st->lane_budget[k], lbud[k]);
    }
    if (LANE_BUDGET<FIRST>() && lane == 0) {
        atomicAdd(&st->dbg[1], ph5);
        atomicAdd(&st->dbg[2], ph6);
        atomicAdd(&st->dbg[0], ph[0]);       // load
        atomicAdd(&st->dbg[3], ph[1]);       // whole nearest hit (dbg[1], dbg[2] = its first two parts)
        atomicAdd(&st->dbg[4], ph[2]);       // shading + radiance write
        atomicAdd(&st->dbg[5], ph[3]);       // compaction + pool write
        atomicAdd(&st->dbg[7], ph[4]);
    }
    if (DEBUG_PHASE && lane == 0) {
        const int base = FIRST ? 0 : 4;
        atomicAdd(&st->dbg[base + 0], ph[0]);
        atomicAdd(&st->dbg[base + 1], ph[1]);
        atomicAdd(&st->dbg[base + 2], ph[2]);
        atomicAdd(&st->dbg[base + 3], ph[4]);
    }
    if (DEBUG_SPAN && bounce == 1 && tid == 0) {
        // lifetime of this workgroup in the bounce-1 launch (10 ns ticks)
        const unsigned long long t1 = __builtin_amdgcn_s_memrealtime(), life = t1 - span0;
#if PT_DEBUG_SPAN == 2
        atomicAdd(&st->dbg[blockIdx.x & 7], life);          // per XCD (workgroups go round the 8 XCDs in dispatch order)
#else
        atomicAdd(&st->dbg[0], life);                       // sum, count, longest, shortest (as max of the complement)
        atomicAdd(&st->dbg[1], 1ull);
        atomicMax(&st->dbg[2], life);
        atomicMax(&st->dbg[3], ~life);
        atomicMax(&st->dbg[4], ~span0);                     // earliest start / latest end over all launches so far
        atomicMax(&st->dbg[5], t1);
#endif
    }
    if (bounce == 1 && blockIdx.x == 0 && tid == 0) {      // clock diagnostics (one thread per launch)
        st->clk[0] = __builtin_amdgcn_s_memtime() - clk0;
        st->clk[1] = __builtin_amdgcn_s_memrealtime() - rt0;
    }
}

// the k_bounce instance for one geometry path (nullptr: combination not built); defined in pt_bounce_g<GEOM>.hip.
// feat = FEAT_* bits; the feature instances exist for compaction 1 only, the MEDIA ones for workgroups of 256 and 512
// (the library's own choices) only.
template <int WG, int GEOM, int FEAT>
static const void *bounce_fn_feat(bool first)
{
    return first ? (const void *)k_bounce<WG, true, GEOM, 1, FEAT> : (const void *)k_bounce<WG, false, GEOM, 1, FEAT>;
}
template <int WG, int GEOM>
static const void *bounce_fn_geom(bool first, int compact, int feat)
{
    if ((feat & FEAT_RESIDENT) != 0) {        // resident paths: the pair queue and the batched walks, workgroups of 256 / 512; plain,
                                              // with direct lighting and / or scattering (not with a shutter time per ray)
        if constexpr ((GEOM == GEOM_PAIR || GEOM == GEOM_WALK4 || GEOM == GEOM_WALK4G) && (WG == 256 || WG == 512)) {
            if (compact != 1 || first) return nullptr;
            if (feat == FEAT_RESIDENT) return (const void *)k_bounce<WG, false, GEOM, 1, FEAT_RESIDENT>;
            if constexpr (GEOM == GEOM_PAIR)
                if (feat == (FEAT_RESIDENT | FEAT_SLAB)) return (const void *)k_bounce<WG, false, GEOM, 1, FEAT_RESIDENT | FEAT_SLAB>;
            if (feat == (FEAT_RESIDENT | FEAT_NEE)) return (const void *)k_bounce<WG, false, GEOM, 1, FEAT_RESIDENT | FEAT_NEE>;
            if (feat == (FEAT_RESIDENT | FEAT_MEDIA)) return (const void *)k_bounce<WG, false, GEOM, 1, FEAT_RESIDENT | FEAT_MEDIA>;
            if (feat == (FEAT_RESIDENT | FEAT_NEE | FEAT_MEDIA)) return (const void *)k_bounce<WG, false, GEOM, 1, FEAT_RESIDENT | FEAT_NEE | FEAT_MEDIA>;
        }
        return nullptr;
    }
    if (feat != 0) {
        if (compact != 1) return nullptr;
        if ((feat & FEAT_SLAB) != 0) {        // the pre-test with slabs: plain pair-queue kernels, workgroups of 256 / 512
            if constexpr (GEOM == GEOM_PAIR && (WG == 256 || WG == 512)) {
                if (feat == FEAT_SLAB) return bounce_fn_feat<WG, GEOM, FEAT_SLAB>(first);
            }
            return nullptr;
        }
        if ((feat & FEAT_MOTION) != 0) {      // per-ray shutter time: the scalar and the pair path, 256-thread workgroups only
            if constexpr ((GEOM == GEOM_SCALAR || GEOM == GEOM_PAIR) && WG == 256) {
                if (feat == FEAT_MOTION) return bounce_fn_feat<256, GEOM, FEAT_MOTION>(first);
                if (feat == (FEAT_MOTION | FEAT_NEE)) return bounce_fn_feat<256, GEOM, FEAT_MOTION | FEAT_NEE>(first);
                if (feat == (FEAT_MOTION | FEAT_MEDIA)) return bounce_fn_feat<256, GEOM, FEAT_MOTION | FEAT_MEDIA>(first);
                return bounce_fn_feat<256, GEOM, FEAT_MOTION | FEAT_NEE | FEAT_MEDIA>(first);
            }
            return nullptr;
        }
        if (feat == FEAT_NEE) return bounce_fn_feat<WG, GEOM, FEAT_NEE>(first);
        if (WG == 256 || WG == 512) {
            if (feat == FEAT_MEDIA) return bounce_fn_feat<(WG == 256 || WG == 512) ? WG : 256, GEOM, FEAT_MEDIA>(first);
            return bounce_fn_feat<(WG == 256 || WG == 512) ? WG : 256, GEOM, FEAT_NEE | FEAT_MEDIA>(first);
        }
        return nullptr;
    }
    if (compact == 1) return first ? (const void *)k_bounce<WG, true, GEOM, 1> : (const void *)k_bounce<WG, false, GEOM, 1>;
    if (compact == 2) return first ? (const void *)k_bounce<WG, true, GEOM, 2> : (const void *)k_bounce<WG, false, GEOM, 2>;
    return first ? (const void *)k_bounce<WG, true, GEOM, 0> : (const void *)k_bounce<WG, false, GEOM, 0>;
}

template <int GEOM>
static const void *bounce_kernel_for(int workgroup, bool first, int compact, int feat)
{
    switch (workgroup) {
    case 64: return bounce_fn_geom<64, GEOM>(first, compact, feat);
    case 128: return bounce_fn_geom<128, GEOM>(first, compact, feat);
    case 256: return bounce_fn_geom<256, GEOM>(first, compact, feat);
    case 512: return bounce_fn_geom<512, GEOM>(first, compact, feat);
    case 1024: return bounce_fn_geom<1024, GEOM>(first, compact, feat);
    default: return nullptr;
    }
}

}  // namespace pt
