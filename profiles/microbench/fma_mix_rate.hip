// fma_mix_rate.hip -- issue cost of v_fma_mix_f32 (fp16 operand x fp32 + fp32) against v_fma_f32 and v_cvt_f32_f16 on gfx950,
// independent operands, 4 and 6 waves per SIMD.  Inline asm so that the compiler cannot rewrite the mix.
//   hipcc --offload-arch=gfx950 -O3 -o fma_mix_rate fma_mix_rate.hip && ./fma_mix_rate
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, int iters)
{
    float a = threadIdx.x * 0.001f + 1.0f, b = 0.5f;
    float x0 = a, x1 = a + 1, x2 = a + 2, x3 = a + 3, x4 = a + 4, x5 = a + 5, x6 = a + 6, x7 = a + 7;
    unsigned h = 0x3C003C00u + threadIdx.x;
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {
            asm volatile("v_fma_f32 %0, %8, %9, %0\n v_fma_f32 %1, %8, %9, %1\n v_fma_f32 %2, %8, %9, %2\n v_fma_f32 %3, %8, %9, %3\n"
                         "v_fma_f32 %4, %8, %9, %4\n v_fma_f32 %5, %8, %9, %5\n v_fma_f32 %6, %8, %9, %6\n v_fma_f32 %7, %8, %9, %7\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
        } else if (MODE == 1) {
            asm volatile("v_fma_mix_f32 %0, %8, %9, %0 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %1, %8, %9, %1 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n"
                         "v_fma_mix_f32 %2, %8, %9, %2 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %3, %8, %9, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n"
                         "v_fma_mix_f32 %4, %8, %9, %4 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %5, %8, %9, %5 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n"
                         "v_fma_mix_f32 %6, %8, %9, %6 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %7, %8, %9, %7 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(h), "v"(b));
        } else if (MODE == 2) {
            asm volatile("v_cvt_f32_f16 %0, %8\n v_cvt_f32_f16 %1, %8\n v_cvt_f32_f16 %2, %8\n v_cvt_f32_f16 %3, %8\n"
                         "v_cvt_f32_f16 %4, %8\n v_cvt_f32_f16 %5, %8\n v_cvt_f32_f16 %6, %8\n v_cvt_f32_f16 %7, %8\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(h));
        } else if (MODE == 3) {
            asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n"
                         "v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a) : "vcc");
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}

template <int MODE>
void run(const char *name, int wps)
{
    int cus = 256, iters = 200000;
    float *d;
    (void)hipMalloc(&d, (size_t)cus * wps * 256 * 4);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(cus * wps), dim3(256), 0, 0, d, 16);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(cus * wps), dim3(256), 0, 0, d, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    double wave_instr = (double)cus * wps * 4 * iters * 8.0;
    double per_simd_per_s = wave_instr / (ms * 1e-3) / (cus * 4);
    printf("%-28s waves/SIMD=%d %8.3f ms  %.3f G instr/s/SIMD  %.2f cyc/instr @2.4GHz\n", name, wps, ms, per_simd_per_s / 1e9, 2.4e9 / per_simd_per_s);
    (void)hipFree(d);
}

int main()
{
    for (int wps : {4, 6}) {
        run<0>("v_fma_f32", wps);
        run<1>("v_fma_mix_f32 (f16 src0)", wps);
        run<2>("v_cvt_f32_f16", wps);
        run<3>("v_cndmask_b32 (vcc)", wps);
    }
    return 0;
}
