// ref_thrust_probe.cpp -- golden-vector generator, TEST INFRASTRUCTURE ONLY.
//
// The reference draws its random numbers from thrust::default_random_engine
// and thrust::uniform_real_distribution<float> (call sites ref:
// src/raytraceKernel.cu:32-35, src/intersections.h:135-137).  thrust is a
// third-party dependency that is not vendored in /root/reference (it ships
// with the CUDA toolkit); this image carries rocThrust 2.8.5
// (/opt/rocm/include/thrust), which implements the same published engines.
// This probe runs the real library (host side only, no GPU needed) and
// prints the draws -> tests/golden/thrust_rng_vectors.json via
// oracle/make_golden.py.
#include <thrust/random.h>

#include <cstdint>
#include <cstdio>
#include <cstring>

static uint32_t bits(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }

int main()
{
    const unsigned seeds[] = {0u, 1u, 7u, 12345u, 2147483646u, 2147483647u, 2147483648u, 4294967294u, 4294967295u,
                              1800329511u, 3028713910u, 3058842707u, 48271u, 0x9E3779B9u};
    const int nseeds = (int)(sizeof seeds / sizeof seeds[0]);
    std::printf("{\n\"generator\": \"oracle/ref_thrust_probe.cpp against rocThrust (/opt/rocm/include/thrust), hipcc host-only\",\n");
    std::printf("\"engine\": [\n");
    for (int i = 0; i < nseeds; i++) {
        thrust::default_random_engine rng(seeds[i]);
        std::printf("{\"seed\": %u, \"raw\": [", seeds[i]);
        for (int k = 0; k < 8; k++) std::printf("%u%s", (unsigned)rng(), k == 7 ? "" : ", ");
        std::printf("], ");
        thrust::default_random_engine r2(seeds[i]);
        thrust::uniform_real_distribution<float> u01(0, 1);
        std::printf("\"u01\": [");
        for (int k = 0; k < 8; k++) std::printf("%u%s", bits((float)u01(r2)), k == 7 ? "" : ", ");
        std::printf("], ");
        thrust::default_random_engine r3(seeds[i]);
        thrust::uniform_real_distribution<float> u02(-0.5, 0.5);
        std::printf("\"u02\": [");
        for (int k = 0; k < 8; k++) std::printf("%u%s", bits((float)u02(r3)), k == 7 ? "" : ", ");
        std::printf("]}%s\n", i == nseeds - 1 ? "" : ",");
    }
    std::printf("],\n\"min\": %u, \"max\": %u\n}\n", (unsigned)thrust::default_random_engine::min,
                (unsigned)thrust::default_random_engine::max);
    return 0;
}
