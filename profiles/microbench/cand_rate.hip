// cand_rate.hip -- cost of the wave-uniform candidate test (pt_device.h candidateT) per primitive per wave,
// with the primitive's matrix as SGPR operands (s_load), as VGPRs copied from SGPRs, or read from LDS.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <string.h>
#include "../../project3-pathtracer_amd/csrc/pt_device.h"
using namespace ptd;
typedef const __attribute__((address_space(4))) uint32_t *const_u32_ptr;

// branch-free candidate tests (no early outs, unguarded exact cores) so that two of them can be interleaved
__device__ __forceinline__ f3 normalize_core(f3 v) { float inv = rcp_core(sqrt_core(v.x * v.x + v.y * v.y + v.z * v.z)); return mk(v.x * inv, v.y * inv, v.z * inv); }
__device__ __forceinline__ bool cand_box_nb(const float *inv, f3 o, f3 d, float &t, uint32_t &face)
{
    f3 ro = mulMV(inv, o, 1.0f);
    f3 rd = normalize_core(mulMV(inv, d, 0.0f));
    float ix = rcp_core(rd.x), iy = rcp_core(rd.y), iz = rcp_core(rd.z);
    float t0 = (-0.5f - ro.x) * ix, t1 = (0.5f - ro.x) * ix;
    float tmin = (t0 < t1) ? t0 : t1, tmax = (t0 < t1) ? t1 : t0;
    int amin = 0, amax = 0;
    t0 = (-0.5f - ro.y) * iy; t1 = (0.5f - ro.y) * iy;
    float tn = (t0 < t1) ? t0 : t1, tf = (t0 < t1) ? t1 : t0;
    if (tn > tmin) { tmin = tn; amin = 1; }
    if (tf < tmax) { tmax = tf; amax = 1; }
    t0 = (-0.5f - ro.z) * iz; t1 = (0.5f - ro.z) * iz;
    tn = (t0 < t1) ? t0 : t1; tf = (t0 < t1) ? t1 : t0;
    if (tn > tmin) { tmin = tn; amin = 2; }
    if (tf < tmax) { tmax = tf; amax = 2; }
    const bool hit = !(tmax < tmin || tmax < 0);
    const bool entry = tmin > 0;
    t = entry ? tmin : tmax;
    const int axis = entry ? amin : amax;
    const float da = (axis == 0) ? rd.x : (axis == 1) ? rd.y : rd.z;
    face = (uint32_t)axis | ((entry ? (da > 0) : !(da > 0)) ? 4u : 0u);
    return hit;
}
__device__ __forceinline__ bool cand_sphere_nb(const float *inv, f3 o, f3 d, float &t)
{
    f3 ro = mulMV(inv, o, 1.0f);
    f3 rd = normalize_core(mulMV(inv, d, 0.0f));
    float vDot = dot(ro, rd);
    float radicand = (float)((double)(vDot * vDot) - ((double)dot(ro, ro) - 0.25));
    float sq = sqrt_core(radicand < 0 ? 0.0f : radicand);
    float t1 = -vDot + sq, t2 = -vDot - sq;
    t = (t1 > 0 && t2 > 0) ? ((t2 < t1) ? t2 : t1) : ((t1 < t2) ? t2 : t1);
    return !(radicand < 0) && !(t1 < 0 && t2 < 0);
}

template <int MODE>
__global__ __launch_bounds__(256) void k(const Prim *prims, int nG, int iters, float *out)
{
    extern __shared__ __attribute__((aligned(128))) unsigned char smem[];
    Prim *s_prims = reinterpret_cast<Prim *>(smem);
    if (MODE == 2) {
        const uint4 *src = reinterpret_cast<const uint4 *>(prims);
        uint4 *dst = reinterpret_cast<uint4 *>(s_prims);
        for (int k = threadIdx.x; k < nG * 8; k += 256) dst[k] = src[k];
        __syncthreads();
    }
    const float u = (threadIdx.x + blockIdx.x * 256) * 1e-5f;
    f3 o = mk(0.1f + u, 4.5f - u, 3.0f), d = normalize(mk(0.3f - u, -0.2f + u, -1.0f));
    float acc = 0.0f;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 3 || MODE == 4) {
            // spheres first (3), then boxes (6): type-sorted, MODE 3 one at a time, MODE 4 two per trip
            const int step = (MODE == 4) ? 2 : 1;
            for (int g = 0; g < nG; g += step) {
                float inv0[12], inv1[12];
                const_u32_ptr h0 = (const_u32_ptr)(uintptr_t)(prims + g);
                const_u32_ptr h1 = (const_u32_ptr)(uintptr_t)(prims + (g + 1 < nG ? g + 1 : g));
#pragma unroll
                for (int k = 0; k < 12; ++k) { inv0[k] = __uint_as_float(h0[4 + k]); inv1[k] = __uint_as_float(h1[4 + k]); }
                const uint32_t ty0 = h0[0], ty1 = h1[0];
                float t0 = 0, t1 = 0; uint32_t f0 = 0, f1 = 0; bool c0, c1 = false;
                if (MODE == 4 && ty0 == ty1 && g + 1 < nG) {
                    if (ty0 == 0u) { c0 = cand_sphere_nb(inv0, o, d, t0); c1 = cand_sphere_nb(inv1, o, d, t1); }
                    else { c0 = cand_box_nb(inv0, o, d, t0, f0); c1 = cand_box_nb(inv1, o, d, t1, f1); }
                } else {
                    c0 = (ty0 == 0u) ? cand_sphere_nb(inv0, o, d, t0) : cand_box_nb(inv0, o, d, t0, f0);
                    if (MODE == 4 && g + 1 < nG) c1 = (ty1 == 0u) ? cand_sphere_nb(inv1, o, d, t1) : cand_box_nb(inv1, o, d, t1, f1);
                }
                acc += (c0 ? t0 : 0.0f) + (c1 ? t1 : 0.0f) + (float)(f0 + f1);
            }
        } else
        for (int g = 0; g < nG; ++g) {
            f3 ro, rd; float t; uint32_t face; bool c;
            if (MODE == 0) {
                const_u32_ptr hp = (const_u32_ptr)(uintptr_t)(prims + g);
                const uint32_t type = hp[0];
                float inv[12];
#pragma unroll
                for (int k = 0; k < 12; ++k) inv[k] = __uint_as_float(hp[4 + k]);
                c = candidateT(type, inv, o, d, ro, rd, t, face);
            } else if (MODE == 1) {
                const_u32_ptr hp = (const_u32_ptr)(uintptr_t)(prims + g);
                const uint32_t type = hp[0];
                float inv[12];
#pragma unroll
                for (int k = 0; k < 12; ++k) { float v = __uint_as_float(hp[4 + k]); asm volatile("v_mov_b32 %0, %1" : "=v"(inv[k]) : "s"(v)); }
                c = candidateT(type, inv, o, d, ro, rd, t, face);
            } else {
                const Prim &P = s_prims[g];
                c = candidateT(P.type, P.inv, o, d, ro, rd, t, face);
            }
            acc += c ? t : 0.0f;
        }
        o.x += 1e-3f * acc;   // make iterations dependent
        d = normalize(mk(d.x + 1e-4f, d.y, d.z));
    }
    out[threadIdx.x + blockIdx.x * 256] = acc;
}

int main()
{
    const int nG = 9, iters = 20000;
    std::vector<Prim> h(nG);
    for (int g = 0; g < nG; g++) {
        memset(&h[g], 0, sizeof(Prim));
        h[g].type = g < 3 ? 0 : 1;
        for (int r = 0; r < 3; r++) for (int c = 0; c < 4; c++) h[g].inv[r * 4 + c] = (r == c ? 0.3f : 0.01f * (g + 1)) + (c == 3 ? 0.2f * g : 0);
    }
    Prim *dp; float *dout;
    (void)hipMalloc(&dp, nG * sizeof(Prim)); (void)hipMemcpy(dp, h.data(), nG * sizeof(Prim), hipMemcpyHostToDevice);
    (void)hipMalloc(&dout, 256 * 8 * 256 * 4);
    const char *names[5] = {"SGPR operands (s_load)", "VGPR copies of SGPRs", "LDS broadcast reads", "branch-free, 1 per trip", "branch-free, 2 per trip"};
    for (int wps : {2, 4, 5, 7}) {
        for (int mode : {0, 3, 4}) {
            size_t lds = 1152 + (wps == 5 ? 27000 : 0);
            hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
            auto launch = [&](int it) {
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256 * wps), dim3(256), lds, 0, dp, nG, it, dout);
                if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(256 * wps), dim3(256), lds, 0, dp, nG, it, dout);
                if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(256 * wps), dim3(256), lds, 0, dp, nG, it, dout);
                if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(256 * wps), dim3(256), lds, 0, dp, nG, it, dout);
                if (mode == 4) hipLaunchKernelGGL(k<4>, dim3(256 * wps), dim3(256), lds, 0, dp, nG, it, dout);
            };
            launch(10);
            (void)hipEventRecord(e0); launch(iters); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            double tests_per_simd = (double)wps * iters * nG;     // wave-level primitive tests per SIMD
            printf("waves/SIMD=%d %-26s %8.2f ms  %.0f cycles per (wave x primitive) @2.3GHz  -> %.1f G ray-prim tests/s chip\n", wps, names[mode], ms,
                   ms * 1e-3 * 2.3e9 / tests_per_simd, 256.0 * 4 * tests_per_simd * 64 / (ms * 1e-3) / 1e9);
        }
    }
    return 0;
}
