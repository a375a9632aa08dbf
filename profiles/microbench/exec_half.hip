// exec_half.hip -- does a gfx950 SIMD retire a wave64 VALU instruction faster when half (or three quarters) of EXEC is zero?
// (RDNA skips an all-zero half in wave64 mode; if CDNA4 did, a pair batch with <= 32 pairs would cost half.)
// Build: hipcc --offload-arch=gfx950 -O3 -o exec_half exec_half.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int OP>
__global__ __launch_bounds__(256) void k(float *out, int iters, float a, float b, int lanes)
{
    const int lane = threadIdx.x & 63;
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    if (lane < lanes) {                       // the whole loop runs under this EXEC mask
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (OP == 0) {                // v_fma_f32 (full rate)
                    x0 = __builtin_fmaf(x0, a, b); x1 = __builtin_fmaf(x1, a, b); x2 = __builtin_fmaf(x2, a, b); x3 = __builtin_fmaf(x3, a, b);
                    x4 = __builtin_fmaf(x4, a, b); x5 = __builtin_fmaf(x5, a, b); x6 = __builtin_fmaf(x6, a, b); x7 = __builtin_fmaf(x7, a, b);
                } else {                      // v_max_f32 / v_min_f32 pairs
                    x0 = fmaxf(x0, a); x1 = fminf(x1, b); x2 = fmaxf(x2, a); x3 = fminf(x3, b);
                    x4 = fmaxf(x4, b); x5 = fminf(x5, a); x6 = fmaxf(x6, b); x7 = fminf(x7, a);
                    x0 += 1.0f; x1 += 1.0f; x2 += 1.0f; x3 += 1.0f; x4 += 1.0f; x5 += 1.0f; x6 += 1.0f; x7 += 1.0f;
                }
            }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}

template <int OP>
void run(const char *name, int blocks_per_cu, int lanes, int instr_per_inner)
{
    int cus = 256, iters = 4096;
    float *d;
    hipMalloc(&d, (size_t)cus * blocks_per_cu * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(cus * blocks_per_cu), dim3(256), 0, 0, d, 16, 1.0001f, 0.5f, lanes);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(cus * blocks_per_cu), dim3(256), 0, 0, d, iters, 1.0001f, 0.5f, lanes);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double wave_instr = (double)cus * blocks_per_cu * 4 * iters * 8.0 * instr_per_inner;
    double per_simd_per_s = wave_instr / (ms * 1e-3) / (cus * 4);
    printf("%-22s active lanes %2d  waves/SIMD=%d  %.3f ms  %.2f cycles/instr at 2.4 GHz\n", name, lanes, blocks_per_cu, ms, 2.4e9 / per_simd_per_s);
    hipFree(d);
}

int main()
{
    for (int w : {1, 4}) for (int lanes : {64, 48, 32, 16, 1}) run<0>("8 x v_fma_f32", w, lanes, 8);
    for (int w : {1, 4}) for (int lanes : {64, 32, 16}) run<1>("min/max + add", w, lanes, 16);
    return 0;
}
