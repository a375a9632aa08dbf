// does gfx950 execute scalar memory atomics (s_atomic_add ... glc), and how fast is a contended counter through them?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#include <chrono>
__global__ void k_s(unsigned *c, unsigned *out, int reps) {
    unsigned last = 0;
    for (int r = 0; r < reps; ++r) {
        unsigned v = 1;
        asm volatile("s_atomic_add %0, %1, 0x0 glc\n s_waitcnt lgkmcnt(0)" : "+s"(v) : "s"(c) : "memory");
        last = v;
        if (reps == 1 && (threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = v;
    }
    if (reps > 1 && threadIdx.x == 0 && last == 0xFFFFFFFFu) out[0] = last;
}
__global__ void k_v(unsigned *c, unsigned *out, int reps) {
    unsigned last = 0;
    for (int r = 0; r < reps; ++r) {
        unsigned v = 0;
        if ((threadIdx.x & 63) == 0) v = atomicAdd(c, 1u);
        last = __builtin_amdgcn_readfirstlane(v);
    }
    if (threadIdx.x == 0 && last == 0xFFFFFFFFu) out[0] = last;
}
int main() {
    unsigned *c, *out;
    const int blocks = 1536, threads = 256, waves = blocks * threads / 64;
    hipMalloc(&c, 4096); hipMalloc(&out, waves * 4);
    hipMemset(c, 0, 4096);
    hipLaunchKernelGGL(k_s, dim3(blocks), dim3(threads), 0, 0, c, out, 1);
    if (hipDeviceSynchronize() != hipSuccess) { printf("scalar atomic kernel failed\n"); return 1; }
    std::vector<unsigned> h(waves); unsigned total;
    hipMemcpy(h.data(), out, waves * 4, hipMemcpyDeviceToHost); hipMemcpy(&total, c, 4, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    bool ok = total == (unsigned)waves;
    for (int i = 0; i < waves; ++i) ok = ok && h[i] == (unsigned)i;
    printf("s_atomic_add: %d waves, counter %u, returned values a permutation of 0..n-1: %s\n", waves, total, ok ? "yes" : "NO");
    for (int pass = 0; pass < 2; ++pass) {
        const int reps = 200;
        hipMemset(c, 0, 4096); hipDeviceSynchronize();
        auto t0 = std::chrono::steady_clock::now();
        if (pass == 0) hipLaunchKernelGGL(k_s, dim3(blocks), dim3(threads), 0, 0, c, out, reps);
        else hipLaunchKernelGGL(k_v, dim3(blocks), dim3(threads), 0, 0, c, out, reps);
        hipDeviceSynchronize();
        double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        hipMemcpy(&total, c, 4, hipMemcpyDeviceToHost);
        printf("%s: %u atomics on one counter in %.3f ms -> %.1f M/s\n", pass == 0 ? "scalar" : "vector", total, dt * 1e3, total / dt / 1e6);
    }
    return ok ? 0 : 2;
}
