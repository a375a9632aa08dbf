// valu_rate.hip -- how many wave64 VALU instructions per cycle does one gfx950 SIMD retire?
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, int iters, float a, float b)
{
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (MODE == 0) {   // 8 independent v_fma_f32
                x0 = __builtin_fmaf(x0, a, b); x1 = __builtin_fmaf(x1, a, b); x2 = __builtin_fmaf(x2, a, b); x3 = __builtin_fmaf(x3, a, b);
                x4 = __builtin_fmaf(x4, a, b); x5 = __builtin_fmaf(x5, a, b); x6 = __builtin_fmaf(x6, a, b); x7 = __builtin_fmaf(x7, a, b);
            } else if (MODE == 1) {   // mul + add (no fma), 16 instructions
                x0 = x0 * a; x1 = x1 * a; x2 = x2 * a; x3 = x3 * a; x4 = x4 * a; x5 = x5 * a; x6 = x6 * a; x7 = x7 * a;
                x0 = x0 + b; x1 = x1 + b; x2 = x2 + b; x3 = x3 + b; x4 = x4 + b; x5 = x5 + b; x6 = x6 + b; x7 = x7 + b;
            } else {   // one dependent chain
                x0 = __builtin_fmaf(x0, a, b); x0 = __builtin_fmaf(x0, a, b); x0 = __builtin_fmaf(x0, a, b); x0 = __builtin_fmaf(x0, a, b);
                x0 = __builtin_fmaf(x0, a, b); x0 = __builtin_fmaf(x0, a, b); x0 = __builtin_fmaf(x0, a, b); x0 = __builtin_fmaf(x0, a, b);
            }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}

template <int MODE>
void run(const char *name, int blocks_per_cu, int instr_per_inner)
{
    int cus = 256, iters = 4096;
    float *d;
    hipMalloc(&d, (size_t)cus * blocks_per_cu * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(cus * blocks_per_cu), dim3(256), 0, 0, d, 16, 1.0001f, 0.5f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(cus * blocks_per_cu), dim3(256), 0, 0, d, iters, 1.0001f, 0.5f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double wave_instr = (double)cus * blocks_per_cu * 4 * iters * 8.0 * instr_per_inner;
    double per_simd_per_s = wave_instr / (ms * 1e-3) / (cus * 4);
    printf("%-28s waves/SIMD=%d  %.3f ms  %.3f G wave-instr/s/SIMD  (= %.2f cycles/instr at 2.4 GHz)  %.1f T lane-ops/s chip\n", name, blocks_per_cu,
           ms, per_simd_per_s / 1e9, 2.4e9 / per_simd_per_s, wave_instr * 64 / (ms * 1e-3) / 1e12);
    hipFree(d);
}

int main()
{
    for (int w : {1, 2, 4, 8}) run<0>("8 independent v_fma_f32", w, 8);
    for (int w : {1, 2, 4, 8}) run<1>("mul+add (16 instr)", w, 16);
    for (int w : {1, 2, 4, 8}) run<2>("dependent v_fma chain", w, 8);
    return 0;
}
