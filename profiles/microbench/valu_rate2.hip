// valu_rate2.hip -- issue cost of packed fp32, transcendental and f64 VALU ops on gfx950, with independent
// and dependent operands, at 4 waves/SIMD.  Inline asm so the compiler cannot rewrite the instruction mix.
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef float float2v __attribute__((ext_vector_type(2)));

#define REP8(S) S S S S S S S S

template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, int iters)
{
    float a = threadIdx.x * 0.001f + 1.0f, b = 0.5f;
    float x0 = a, x1 = a + 1, x2 = a + 2, x3 = a + 3, x4 = a + 4, x5 = a + 5, x6 = a + 6, x7 = a + 7;
    float2v p0 = {a, a}, p1 = {a + 1, a}, p2 = {a + 2, a}, p3 = {a + 3, a}, pa = {1.0001f, 1.0001f}, pb = {0.5f, 0.5f};
    double d0 = a, d1 = a + 1, d2 = a + 2, d3 = a + 3, da = 1.0001, db = 0.5;
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {        // 8 x independent v_pk_fma_f32 (4 regs x2)
            asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                         "v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pa), "v"(pb));
        } else if (MODE == 1) { // 8 x dependent v_pk_fma_f32
            asm volatile(REP8("v_pk_fma_f32 %0, %0, %1, %2\n") : "+v"(p0) : "v"(pa), "v"(pb));
        } else if (MODE == 2) { // 8 x independent v_pk_mul_f32
            asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n"
                         "v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pa));
        } else if (MODE == 3) { // 8 x independent v_rcp_f32
            asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
        } else if (MODE == 4) { // 8 x dependent v_rcp_f32
            asm volatile(REP8("v_rcp_f32 %0, %0\n s_nop 0\n") : "+v"(x0));
        } else if (MODE == 5) { // 8 x independent v_fma_f64 (4 regs x2)
            asm volatile("v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5\n"
                         "v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5\n"
                         : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(da), "v"(db));
        } else if (MODE == 6) { // dependent pairs: 2 interleaved chains of v_fma_f32
            asm volatile("v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3\n v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3\n"
                         "v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3\n v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3\n"
                         : "+v"(x0), "+v"(x1) : "v"(a), "v"(b));
        } else if (MODE == 7) { // 3 interleaved chains (9 instr; counted as 8 below -> scale)
            asm volatile("v_fma_f32 %0, %0, %3, %4\n v_fma_f32 %1, %1, %3, %4\n v_fma_f32 %2, %2, %3, %4\n v_fma_f32 %0, %0, %3, %4\n"
                         "v_fma_f32 %1, %1, %3, %4\n v_fma_f32 %2, %2, %3, %4\n v_fma_f32 %0, %0, %3, %4\n v_fma_f32 %1, %1, %3, %4\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2) : "v"(a), "v"(b));
        } else if (MODE == 8) { // 8 x dependent v_fma_f32 (asm)
            asm volatile(REP8("v_fma_f32 %0, %0, %1, %2\n") : "+v"(x0) : "v"(a), "v"(b));
        } else if (MODE == 9) { // dependent v_mul with SGPR operand
            asm volatile(REP8("v_mul_f32 %0, %1, %0\n") : "+v"(x0) : "s"(1.0001f));
        } else if (MODE == 10) { // 8 x independent v_sqrt_f32
            asm volatile("v_sqrt_f32 %0, %0\n v_sqrt_f32 %1, %1\n v_sqrt_f32 %2, %2\n v_sqrt_f32 %3, %3\n v_sqrt_f32 %4, %4\n v_sqrt_f32 %5, %5\n v_sqrt_f32 %6, %6\n v_sqrt_f32 %7, %7\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
        } else if (MODE == 11) { // v_cndmask chain dependent via vcc
            asm volatile(REP8("v_cmp_gt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc\n") : "+v"(x0) : "v"(a) : "vcc");
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + p0.x + p1.x + p2.x + p3.x + p0.y + (float)(d0 + d1 + d2 + d3);
}

template <int MODE>
void run(const char *name, int wps, double instr_per_iter)
{
    int cus = 256, iters = 400000;
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
    double wave_instr = (double)cus * wps * 4 * iters * instr_per_iter;
    double per_simd_per_s = wave_instr / (ms * 1e-3) / (cus * 4);
    printf("%-44s waves/SIMD=%d %8.3f ms  %.3f G instr/s/SIMD  %.2f cyc/instr @2.4GHz\n", name, wps, ms, per_simd_per_s / 1e9, 2.4e9 / per_simd_per_s);
    (void)hipFree(d);
}

int main()
{
    for (int w : {1, 2, 4, 8}) {
        run<0>("v_pk_fma_f32 independent", w, 8);
        run<1>("v_pk_fma_f32 dependent", w, 8);
        run<2>("v_pk_mul_f32 independent", w, 8);
        run<3>("v_rcp_f32 independent", w, 8);
        run<4>("v_rcp_f32 dependent (+s_nop)", w, 8);
        run<10>("v_sqrt_f32 independent", w, 8);
        run<5>("v_fma_f64 independent", w, 8);
        run<8>("v_fma_f32 dependent", w, 8);
        run<6>("v_fma_f32 2 interleaved chains", w, 8);
        run<7>("v_fma_f32 3 interleaved chains", w, 8);
        run<9>("v_mul_f32 dependent, SGPR operand", w, 8);
        run<11>("v_cmp+v_cndmask dependent (16 instr)", w, 16);
    }
    return 0;
}
