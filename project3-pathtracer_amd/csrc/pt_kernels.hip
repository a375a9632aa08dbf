// pt_kernels.hip -- hand-written HIP kernels of the path-tracing hot path for gfx950 (MI355X, CDNA4).
//
//   k_bounce<FIRST=true>   raycastFromCameraKernel fused into bounce 0      (ref stub: src/raytraceKernel.cu:38-45)
//   k_bounce               one bounce of raytraceRay: nearest hit over the primitive list, shading,
//                          Russian roulette, framebuffer accumulate            (ref stub: src/raytraceKernel.cu:91-104)
//                          + stream compaction of the surviving rays           (ref: README.md:70, absent in the code)
//   k_send_image_to_pbo    sendImageToPBO                                      (ref: src/raytraceKernel.cu:58-89)
//   k_iter_*               iteration bookkeeping (live-ray counters, stats) kept on the device so that one
//                          hipGraph can be replayed per iteration
//
// Execution model (wave64): one lane = one live ray, up to 16 iterations in flight per launch.  The nearest-hit
// search comes in seven interchangeable forms (GEOM_* in pt_bounce.h, where k_bounce lives); the default ones first run a cheap per-lane box
// pre-test and then do the exact intersection work on full 64-wide batches of (ray, primitive) pairs drawn
// from a wave-private LDS queue.  Rays live in SoA pools (pt_internal.h RayPool); survivors of a bounce are
// written densely into the other pool: wave ballot + v_mbcnt prefix and ONE atomic per wave on one of 32
// sharded counters (no workgroup barrier).  Results do not depend on the order rays land in the pool because
// every RNG stream is keyed on (global pixel, iteration, bounce).  DESIGN.md section 5 has the measurements.
//
// Compile with -ffp-contract=off (see pt_device.h).
#include <stdlib.h>

#include <algorithm>

#include "pt_bounce.h"

namespace pt {

// ---------------------------------------------------------------------------------------------
// self-test of the short exact sqrt / reciprocal sequences (pt_device.h): every one of the 2^32 fp32 bit
// patterns against hipcc's correctly rounded sqrtf(x), 1.0f/x and 1.0f/sqrtf(x).  out[0..2] = mismatch counts.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_selftest_math(unsigned long long *out)
{
    const uint64_t stride = (uint64_t)gridDim.x * 256u;
    unsigned long long bad0 = 0, bad1 = 0, bad2 = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x; i < (1ull << 32); i += stride) {
        const float x = __uint_as_float((uint32_t)i);
        const float a = sqrtf(x), b = sqrt_rn(x);
        if (__float_as_uint(a) != __float_as_uint(b) && !(a != a && b != b)) bad0++;
        const float c = 1.0f / x, d = rcp_rn(x);
        if (__float_as_uint(c) != __float_as_uint(d) && !(c != c && d != d)) bad1++;
        const float e = 1.0f / sqrtf(x), f = rsqrt_rn(x), g = rsqrt_near_one(x);
        if (__float_as_uint(e) != __float_as_uint(f) && !(e != e && f != f)) bad2++;
        if (__float_as_uint(e) != __float_as_uint(g) && !(e != e && g != g)) bad2++;
        // minstd_seed's branch-free form against the definition: s mod (2^31 - 1), 0 -> 1 (counted with the 1/sqrt checks)
        const uint32_t sd = (uint32_t)i, md = sd % 2147483647u;
        if (minstd_seed(sd) != (md == 0u ? 1u : md)) bad2++;
    }
    if (bad0) atomicAdd(&out[0], bad0);
    if (bad1) atomicAdd(&out[1], bad1);
    if (bad2) atomicAdd(&out[2], bad2);
}

// ---------------------------------------------------------------------------------------------
// known-answer tests of the device functions (one thread): see pt_device_kat() in include/pt_abi.h
// ---------------------------------------------------------------------------------------------
__global__ void k_device_kat(int op, const float *in, float *out, int n_out)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    auto u = [&](int k) { return __float_as_uint(in[k]); };
    auto put3 = [&](int k, f3 v) { out[k] = v.x; out[k + 1] = v.y; out[k + 2] = v.z; };
    switch (op) {
    case PT_KAT_HASH: out[0] = __uint_as_float(wang_hash(u(0))); break;
    case PT_KAT_U01_SEQUENCE: {
        uint32_t s = minstd_seed(u(0));
        for (int k = 0; k < n_out; ++k) { s = minstd_next(s); out[k] = u01_of(s); }
        break;
    }
    case PT_KAT_NOISE: put3(0, generateRandomNumberFromThread(in[0], in[2], (int)in[3], (int)in[4])); break;
    case PT_KAT_INTERSECT: {
        Prim P;
        P.type = u(0);
        for (int k = 0; k < 12; ++k) { P.fwd[k] = in[1 + k]; P.inv[k] = in[17 + k]; }
        const f3 c = mulMV(P.fwd, mk(0, 0, 0), 1.0f);
        P.cx = c.x; P.cy = c.y; P.cz = c.z;
        f3 p = mk(0, 0, 0), n = mk(0, 0, 0);
        const f3 o = mk(in[33], in[34], in[35]), d = mk(in[36], in[37], in[38]);
        out[0] = intersectPrim<false>(P, o, d, o, p, n);
        put3(1, p);
        put3(4, n);
        break;
    }
    case PT_KAT_HEMISPHERE: put3(0, randomDirectionInHemisphere(mk(in[0], in[1], in[2]), in[3], in[4])); break;
    case PT_KAT_RADIUSES: put3(0, getRadiuses(in)); break;
    case PT_KAT_POINT_ON_CUBE: put3(0, getRandomPointOnCube(in, in[16])); break;
    case PT_KAT_POINT_ON_SPHERE: put3(0, getRandomPointOnSphere(in, in[16])); break;
    case PT_KAT_MULTIPLY_MV: put3(0, mulMV(in, mk(in[16], in[17], in[18]), in[19])); break;
    case PT_KAT_POINT_ON_RAY: put3(0, pointOnRay(mk(in[0], in[1], in[2]), mk(in[3], in[4], in[5]), in[6])); break;
    case PT_KAT_REFLECT: put3(0, reflectionDirection(mk(in[0], in[1], in[2]), mk(in[3], in[4], in[5]))); break;
    case PT_KAT_REFRACT: {
        bool tir;
        put3(0, transmissionDirection(mk(in[0], in[1], in[2]), mk(in[3], in[4], in[5]), in[6], in[7], tir));
        break;
    }
    case PT_KAT_FRESNEL:
        out[0] = fresnelReflectance(mk(in[0], in[1], in[2]), mk(in[3], in[4], in[5]), in[6], in[7], mk(in[8], in[9], in[10]));
        break;
    case PT_KAT_TRANSMISSION: {
        const f3 t = calculateTransmission(mk(in[0], in[1], in[2]), in[3]);
        out[0] = t.x; out[1] = t.y; out[2] = t.z;
        break;
    }
    case PT_KAT_SAMPLE_LIGHT: {
        float fwd[12];
        for (int k = 0; k < 12; ++k) fwd[k] = in[1 + k];
        f3 y, n;
        sampleLight(__float_as_uint(in[0]), fwd, mulMV(fwd, mk(0, 0, 0), 1.0f), in[17], y, n);
        out[0] = y.x; out[1] = y.y; out[2] = y.z; out[3] = n.x; out[4] = n.y; out[5] = n.z;
        break;
    }
    case PT_KAT_LOG: out[0] = log_poly(in[0]); break;
    case PT_KAT_SCATTER: {
        f3 o = mk(in[0], in[1], in[2]), d = mk(in[3], in[4], in[5]), T = mk(in[11], in[12], in[13]);
        float depth = in[6];
        const bool sc = calculateScatterAndAbsorption(o, d, depth, mk(in[7], in[8], in[9]), in[10], T, in[14], in[15], in[16]);
        out[0] = sc ? 1.0f : 0.0f;
        put3(1, o);
        put3(4, d);
        out[7] = depth;
        put3(8, T);
        break;
    }
    case PT_KAT_SAMPLE_TRIANGLE: {
        const f3 v0 = mk(in[0], in[1], in[2]), e1 = mk(in[3], in[4], in[5]), e2 = mk(in[6], in[7], in[8]);
        put3(0, sampleTriangle(v0, e1, e2, in[9], in[10]));
        out[3] = triangleArea(e1, e2);
        break;
    }
    default: break;
    }
}

hipError_t launch_device_kat(hipStream_t s, int op, const float *in, float *out, int n_out)
{
    hipLaunchKernelGGL(k_device_kat, dim3(1), dim3(64), 0, s, op, in, out, n_out);
    return hipGetLastError();
}

hipError_t launch_selftest_math(hipStream_t s, unsigned long long *out)
{
    hipLaunchKernelGGL(k_selftest_math, dim3(256 * 8), dim3(256), 0, s, out);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// iteration bookkeeping
// ---------------------------------------------------------------------------------------------
// starts a pt_render call: first iteration and the batch schedule (q iterations per batch, the first r batches one more);
// this launch sequence renders batches j0, j0 + stride, ... of the call
__global__ void k_iter_set(IterState *st, uint32_t iter_first, uint32_t q, uint32_t r, uint32_t j0, uint32_t stride)
{
    st->iter = iter_first;
    st->nslot = 0u;
    st->sched_q = q;
    st->sched_r = r;
    st->sched_j = j0;
    st->sched_first = iter_first;
    st->sched_stride = stride;
}

// starts a batch: fold the sequence's previous batch's per-bounce live counts into the stats, reset them, and publish this
// batch's first iteration and iteration count (batch j of the call starts j*q + min(j, r) iterations after the call's first)
__global__ void k_iter_begin(IterState *st, uint32_t npix, int depth, int compact)
{
    const int b = threadIdx.x;
    const uint32_t j = st->sched_j;
    const uint32_t nslot = st->sched_q + (j < st->sched_r ? 1u : 0u);
    const uint32_t iter = st->sched_first + j * st->sched_q + (j < st->sched_r ? j : st->sched_r);
    __syncthreads();                             // everyone has read the schedule before thread 0 moves it on
    if (b <= depth) {
        unsigned long long sum = 0;
        for (int sh = 0; sh < NSHARD; ++sh) {
            sum += st->counts[cnt_index(b, sh)];
            st->counts[cnt_index(b, sh)] = (b == 0 && sh == 0 && compact) ? npix * nslot : 0u;
        }
        if (b < depth) st->live_in[b] += sum;
    }
    if (b < NSHARD) st->draw[b * CNT_STRIDE] = 0u;    // (resident paths: chunk draw counters of the later-bounce launch)
    if (b == 0) {
        st->serial += (uint32_t)PT_MAX_SEQUENCES;     // (stays congruent to the sequence's number: pt_create)
        st->iter = iter;
        st->nslot = nslot;
        st->sched_j = j + st->sched_stride;
        st->iterations += nslot;
    }
}

// Accumulation (DESIGN.md "Canonical semantics" 6): image = (image*(i-1) + L_i)/i for the batch's iterations in
// order, one thread per pixel, streaming.  The per-path samples were written to lbuf by k_bounce -- the non-zero ones
// only, stamped with the batch's serial number: an entry with another stamp is a sample of zero.  All of a thread's plane
// loads (16 bytes each, coalesced) are in flight before the dependent chain of divisions starts.
__global__ __launch_bounds__(256) void k_accumulate(float *image, const float4 *lbuf, const IterState *st, int npix)
{
    const uint32_t iter0 = st->iter, serial = st->serial;
    const int nslot = (int)st->nslot;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < npix; i += gridDim.x * 256) {
        float r = image[3 * i], g = image[3 * i + 1], b = image[3 * i + 2];
        float4 L[PT_MAX_BATCH];
#pragma unroll
        for (int k = 0; k < PT_MAX_BATCH; ++k)
            if (k < nslot) {
                const v4f e = __builtin_nontemporal_load(reinterpret_cast<const v4f *>(lbuf + (size_t)k * (size_t)npix + (size_t)i));
                L[k] = make_float4(e.x, e.y, e.z, e.w);
            }
#pragma unroll
        for (int k = 0; k < PT_MAX_BATCH; ++k) {
            if (k < nslot) {
                const bool have = __float_as_uint(L[k].w) == serial;
                const float lr = have ? L[k].x : 0.0f, lg = have ? L[k].y : 0.0f, lb = have ? L[k].z : 0.0f;
                const uint32_t it = iter0 + (uint32_t)k;
                const float fi = (float)it, fim1 = (float)(it - 1u);
                // iteration 1 restarts the running mean: (old*0 + L)/1 == L for every finite old value, and taking L
                // directly keeps a NaN / Inf left in a caller-owned buffer from surviving the restart
                r = (it == 1u) ? lr : (r * fim1 + lr) / fi;
                g = (it == 1u) ? lg : (g * fim1 + lg) / fi;
                b = (it == 1u) ? lb : (b * fim1 + lb) / fi;
            }
        }
        image[3 * i] = r;
        image[3 * i + 1] = g;
        image[3 * i + 2] = b;
    }
}

__global__ void k_iter_fold(IterState *st, int depth)
{
    const int b = threadIdx.x;
    if (b <= depth) {
        unsigned long long sum = 0;
        for (int sh = 0; sh < NSHARD; ++sh) {
            sum += st->counts[cnt_index(b, sh)];
            st->counts[cnt_index(b, sh)] = 0u;
        }
        if (b < depth) st->live_in[b] += sum;
    }
}

// sendImageToPBO (ref: src/raytraceKernel.cu:58-89): x255 (the reference multiplies by the double 255.0 and
// stores to float: one rounding, identical to the fp32 product), clamp above, truncate, w = 0.
__global__ __launch_bounds__(256) void k_send_image_to_pbo(pt_uchar4 *pbo, const float *image, int npix)
{
    for (int i = blockIdx.x * 256 + threadIdx.x; i < npix; i += gridDim.x * 256) {
        float r = image[3 * i] * 255.0f, g = image[3 * i + 1] * 255.0f, b = image[3 * i + 2] * 255.0f;
        if (r > 255) r = 255;
        if (g > 255) g = 255;
        if (b > 255) b = 255;
        pt_uchar4 o;
        o.x = (unsigned char)r;
        o.y = (unsigned char)g;
        o.z = (unsigned char)b;
        o.w = 0;
        pbo[i] = o;
    }
}

// ---------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------
size_t bounce_lds_bytes(const KParams &p, const LaunchCfg &cfg)
{
    size_t prim = (cfg.geom == GEOM_LDS || cfg.geom == GEOM_QUEUE || cfg.geom == GEOM_PAIR) ? (size_t)p.nG * 2 * sizeof(PrimPad) : 0;
    size_t queue = (size_t)(cfg.workgroup / 64) * (cfg.geom == GEOM_QUEUE ? WAVE_QUEUE_BYTES
                                                   : ((cfg.geom == GEOM_PAIR || cfg.geom == GEOM_WALK_PAIR) ? ((cfg.motion && cfg.geom == GEOM_PAIR) ? PAIR_QUEUE_MOTION_BYTES : PAIR_QUEUE_BYTES)
                                                      : ((cfg.geom == GEOM_WALK4 || cfg.geom == GEOM_WALK4G) ? walk4_wave_bytes(p.ntri) : 0)));
    if (cfg.geom == GEOM_WALK4) prim += ((size_t)p.nnodes4 * W4_FLOATS * 4 + 127) & ~(size_t)127;
    if (cfg.geom == GEOM_BVH || cfg.geom == GEOM_WALK_PAIR) prim += (size_t)p.nnodes * sizeof(BvhNode);
    if (cfg.geom == GEOM_PAIR) prim += (size_t)p.nG * PAIR_BOX_BYTES * (cfg.nee ? 2 : 1);
    size_t mats = (size_t)((p.nM * M_PLANES + 3) & ~3) * sizeof(float);
    size_t scan = (size_t)((2 * (cfg.workgroup / 64) + 2 + 3) & ~3) * sizeof(uint32_t);
    static const size_t extra = getenv("PT_EXTRA_LDS") ? (size_t)atol(getenv("PT_EXTRA_LDS")) : 0;   // occupancy experiments
    // resident paths: stream keys of every (bounce, slot) and the histogram of the bounces the paths ended at
    const size_t resident = cfg.resident ? (size_t)(p.depth * MAXSLOT + ((p.depth + 3) & ~3)) * sizeof(uint32_t) : 0;
    return prim + queue + mats + scan + resident + extra;
}

// one translation unit per geometry path (pt_bounce_g<N>.hip)
const void *bounce_kernel_g0(int workgroup, bool first, int compact, int feat);
const void *bounce_kernel_g1(int workgroup, bool first, int compact, int feat);
const void *bounce_kernel_g2(int workgroup, bool first, int compact, int feat);
const void *bounce_kernel_g3(int workgroup, bool first, int compact, int feat);
const void *bounce_kernel_g4(int workgroup, bool first, int compact, int feat);
const void *bounce_kernel_g5(int workgroup, bool first, int compact, int feat);
const void *bounce_kernel_g6(int workgroup, bool first, int compact, int feat);
const void *bounce_kernel_g7(int workgroup, bool first, int compact, int feat);

static const void *select_bounce(const LaunchCfg &cfg, bool first)
{
    const int feat = cfg.nee | (cfg.media << 1) | (cfg.motion << 2) | ((cfg.resident && !first) ? FEAT_RESIDENT : 0) | (cfg.slab ? FEAT_SLAB : 0);
    switch (cfg.geom) {
    case GEOM_SCALAR: return bounce_kernel_g0(cfg.workgroup, first, cfg.compact, feat);
    case GEOM_LDS: return bounce_kernel_g1(cfg.workgroup, first, cfg.compact, feat);
    case GEOM_QUEUE: return bounce_kernel_g2(cfg.workgroup, first, cfg.compact, feat);
    case GEOM_BVH: return bounce_kernel_g3(cfg.workgroup, first, cfg.compact, feat);
    case GEOM_PAIR: return bounce_kernel_g4(cfg.workgroup, first, cfg.compact, feat);
    case GEOM_WALK_PAIR: return bounce_kernel_g5(cfg.workgroup, first, cfg.compact, feat);
    case GEOM_WALK4: return bounce_kernel_g6(cfg.workgroup, first, cfg.compact, feat);
    case GEOM_WALK4G: return bounce_kernel_g7(cfg.workgroup, first, cfg.compact, feat);
    default: return nullptr;
    }
}

bool bounce_resident_available(const LaunchCfg &cfg)
{
    LaunchCfg t = cfg;
    t.resident = 1;
    return !cfg.motion && cfg.compact == 1 && select_bounce(t, false) != nullptr;
}

int bounce_max_blocks_per_cu(const KParams &p, const LaunchCfg &cfg)
{
    const void *fn = select_bounce(cfg, false);
    int nb = 0;
    if (!fn) return 0;
    // more than 64 KiB of dynamic LDS per workgroup has to be asked for (large hierarchies)
    const size_t lds = bounce_lds_bytes(p, cfg);
    if (lds > 64 * 1024) {
        const void *f0 = select_bounce(cfg, true);
        if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess ||
            hipFuncSetAttribute(f0, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return 0;
    }
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, cfg.workgroup, bounce_lds_bytes(p, cfg)) != hipSuccess)
        return 0;
    return nb;
}

// cfg.resident: bounce 0 is the camera kernel as ever, bounce 1 the ONE launch that traces bounces 1 .. depth - 1
hipError_t launch_bounce(hipStream_t s, const KParams &p, const LaunchCfg &cfg, int bounce)
{
    LaunchCfg lc = cfg;
    if (bounce == 0) lc.resident = 0;
    const void *fn = select_bounce(lc, bounce == 0);
    if (!fn) return hipErrorInvalidValue;
    KParams pc = p;
    int b = bounce;
    void *args[] = {(void *)&pc, (void *)&b};
    return hipLaunchKernel(fn, dim3((unsigned)cfg.grid), dim3((unsigned)cfg.workgroup), args,
                           bounce_lds_bytes(p, lc), s);
}

hipError_t launch_iter_set(hipStream_t s, IterState *st, uint32_t iter_first, uint32_t q, uint32_t r, uint32_t j0, uint32_t stride)
{
    hipLaunchKernelGGL(k_iter_set, dim3(1), dim3(1), 0, s, st, iter_first, q, r, j0, stride);
    return hipGetLastError();
}

hipError_t launch_iter_begin(hipStream_t s, IterState *st, int npix, int depth, int compact)
{
    hipLaunchKernelGGL(k_iter_begin, dim3(1), dim3(128), 0, s, st, (uint32_t)npix, depth, compact != 0 ? 1 : 0);
    return hipGetLastError();
}

hipError_t launch_accumulate(hipStream_t s, float *image, const float *lbuf, const IterState *st, int npix)
{
    int grid = (npix + 255) / 256;
    if (grid > 16384) grid = 16384;
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(k_accumulate, dim3(grid), dim3(256), 0, s, image, reinterpret_cast<const float4 *>(lbuf), st, npix);
    return hipGetLastError();
}

hipError_t launch_iter_fold(hipStream_t s, IterState *st, int depth)
{
    hipLaunchKernelGGL(k_iter_fold, dim3(1), dim3(128), 0, s, st, depth);
    return hipGetLastError();
}

hipError_t launch_send_image_to_pbo(hipStream_t s, pt_uchar4 *pbo, const float *image, int npix)
{
    int grid = (npix + 255) / 256;
    if (grid > 4096) grid = 4096;
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(k_send_image_to_pbo, dim3(grid), dim3(256), 0, s, pbo, image, npix);
    return hipGetLastError();
}

}  // namespace pt
