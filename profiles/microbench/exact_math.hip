// exact_math.hip -- exhaustive check (all 2^32 bit patterns) of short correctly-rounded sqrt / reciprocal
// sequences against hipcc's own correctly rounded sqrtf(x) and 1.0f/x, to find where they may replace them.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>

__device__ __forceinline__ float sqrt_fast(float x)
{
    float y = __builtin_amdgcn_sqrtf(x);
    float ym = __uint_as_float(__float_as_uint(y) - 1u), yp = __uint_as_float(__float_as_uint(y) + 1u);
    float rm = __builtin_fmaf(-ym, y, x), rp = __builtin_fmaf(-yp, y, x);
    y = (rm <= 0.0f) ? ym : y;
    y = (rp > 0.0f) ? yp : y;
    return y;
}
__device__ __forceinline__ float rcp_fast(float d)
{
    float r0 = __builtin_amdgcn_rcpf(d);
    float e0 = __builtin_fmaf(-d, r0, 1.0f);
    float r1 = __builtin_fmaf(e0, r0, r0);
    float e1 = __builtin_fmaf(-d, r1, 1.0f);
    float q1 = __builtin_fmaf(e1, r1, r1);
    float e2 = __builtin_fmaf(-d, q1, 1.0f);
    return __builtin_fmaf(e2, r1, q1);
}
__device__ __forceinline__ float rcp_fast2(float d)   // one refinement less
{
    float r0 = __builtin_amdgcn_rcpf(d);
    float e0 = __builtin_fmaf(-d, r0, 1.0f);
    float r1 = __builtin_fmaf(e0, r0, r0);
    float e1 = __builtin_fmaf(-d, r1, 1.0f);
    return __builtin_fmaf(e1, r1, r1);
}

// per-exponent mismatch histograms: [0..255] positive, [256..511] negative
__global__ void check(unsigned long long *bad_sqrt, unsigned long long *bad_rcp, unsigned long long *bad_rcp2)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (1ull << 32); i += stride) {
        const uint32_t u = (uint32_t)i;
        const float x = __uint_as_float(u);
        const uint32_t bin = ((u >> 23) & 0xffu) + ((u >> 31) ? 256u : 0u);
        const float a = sqrtf(x), b = sqrt_fast(x);
        if (__float_as_uint(a) != __float_as_uint(b) && !(a != a && b != b)) atomicAdd(&bad_sqrt[bin], 1ull);
        const float c = 1.0f / x, d = rcp_fast(x), e = rcp_fast2(x);
        if (__float_as_uint(c) != __float_as_uint(d) && !(c != c && d != d)) atomicAdd(&bad_rcp[bin], 1ull);
        if (__float_as_uint(c) != __float_as_uint(e) && !(c != c && e != e)) atomicAdd(&bad_rcp2[bin], 1ull);
    }
}

static void report(const char *name, const unsigned long long *h)
{
    unsigned long long total = 0;
    for (int i = 0; i < 512; i++) total += h[i];
    printf("%s: %llu mismatching inputs; exponent bins with mismatches:", name, total);
    int shown = 0;
    for (int i = 0; i < 512; i++)
        if (h[i]) { if (shown++ < 40) printf(" %s%d(%llu)", i >= 256 ? "-" : "+", (i & 255) - 127, h[i]); }
    printf("%s\n", shown > 40 ? " ..." : "");
}

int main()
{
    unsigned long long *d, h[3 * 512];
    (void)hipMalloc(&d, sizeof h);
    (void)hipMemset(d, 0, sizeof h);
    hipLaunchKernelGGL(check, dim3(256 * 8), dim3(256), 0, 0, d, d + 512, d + 1024);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    report("sqrt_fast vs sqrtf", h);
    report("rcp_fast (2 refinements + final) vs 1.0f/x", h + 512);
    report("rcp_fast2 (1 refinement + final) vs 1.0f/x", h + 1024);
    return 0;
}
