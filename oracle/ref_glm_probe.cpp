// ref_glm_probe.cpp -- golden-vector generator, TEST INFRASTRUCTURE ONLY.
//
// Compiles against the reference's own vendored GLM 0.9.5.4, in place
// (-I/root/reference/external/include; nothing is copied), and prints bit
// patterns of the GLM calls the reference's transform builder and
// intersection code make (ref: src/utilities.cpp:74-90, src/scene.cpp:125-127,
// src/intersections.h:46-48,85-86,113-116,120-129).  The reference's own .cpp/.h files cannot be compiled
// here (they include <cuda_runtime.h>, absent from this image), so the probe
// issues the same GLM calls directly.  Output -> tests/golden/glm_vectors.json
// via oracle/make_golden.py.
#define GLM_FORCE_RADIANS
#include <glm/glm.hpp>
#include <glm/gtc/matrix_transform.hpp>
#include <glm/gtc/matrix_inverse.hpp>

#include <cstdint>
#include <cstdio>
#include <cstring>

static uint32_t bits(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }

// small deterministic generator (xorshift32) -> floats in [lo, hi)
static uint32_t g_state = 0x2545F491u;
static float rnd(float lo, float hi)
{
    g_state ^= g_state << 13; g_state ^= g_state >> 17; g_state ^= g_state << 5;
    return lo + (hi - lo) * (float)(g_state >> 8) * (1.0f / 16777216.0f);
}

static void print_rows(const char *key, glm::mat4 a, const char *tail)
{
    a = glm::transpose(a);   // glmMat4ToCudaMat4: rows of the cudaMat4
    std::printf("\"%s\": [", key);
    for (int r = 0; r < 4; r++)
        for (int c = 0; c < 4; c++) std::printf("%u%s", bits(a[r][c]), (r == 3 && c == 3) ? "" : ", ");
    std::printf("]%s", tail);
}

static void print_v3(const char *key, glm::vec3 v, const char *tail)
{
    std::printf("\"%s\": [%u, %u, %u]%s", key, bits(v.x), bits(v.y), bits(v.z), tail);
}

int main()
{
    std::printf("{\n\"generator\": \"oracle/ref_glm_probe.cpp against /root/reference/external/include/glm (0.9.5.4), g++ -O2 -ffp-contract=off\",\n");
    // ---- TRS builds: the 9 objects of sampleScene.txt plus random ones
    const float fixed[][9] = {
        {0, 0, 0, 0, 0, 90, .01f, 10, 10}, {0, 5, -5, 0, 90, 0, .01f, 10, 10}, {0, 10, 0, 0, 0, 90, .01f, 10, 10},
        {-5, 5, 0, 0, 0, 0, .01f, 10, 10}, {5, 5, 0, 0, 0, 0, .01f, 10, 10}, {0, 2, 0, 0, 180, 0, 3, 3, 3},
        {2, 5, 2, 0, 180, 0, 2.5f, 2.5f, 2.5f}, {-2, 5, -2, 0, 180, 0, 3, 3, 3}, {0, 10, 0, 0, 0, 90, .3f, 3, 3}};
    std::printf("\"trs\": [\n");
    const int nfixed = 9, nrand = 40;
    for (int i = 0; i < nfixed + nrand; i++) {
        float p[9];
        if (i < nfixed) std::memcpy(p, fixed[i], sizeof p);
        else {
            for (int k = 0; k < 3; k++) p[k] = rnd(-10, 10);
            for (int k = 3; k < 6; k++) p[k] = rnd(-7, 7);
            for (int k = 6; k < 9; k++) p[k] = rnd(0.05f, 6);
        }
        glm::vec3 translation(p[0], p[1], p[2]), rotation(p[3], p[4], p[5]), scale(p[6], p[7], p[8]);
        // the call sequence of utilityCore::buildTransformationMatrix
        glm::mat4 translationMat = glm::translate(glm::mat4(), translation);
        glm::mat4 rotationMat = glm::rotate(glm::mat4(), rotation.x, glm::vec3(1, 0, 0));
        rotationMat = rotationMat * glm::rotate(glm::mat4(), rotation.y, glm::vec3(0, 1, 0));
        rotationMat = rotationMat * glm::rotate(glm::mat4(), rotation.z, glm::vec3(0, 0, 1));
        glm::mat4 scaleMat = glm::scale(glm::mat4(), scale);
        glm::mat4 transform = translationMat * rotationMat * scaleMat;
        std::printf("{");
        print_v3("translation", translation, ", ");
        print_v3("rotation", rotation, ", ");
        print_v3("scale", scale, ", ");
        print_rows("transform", transform, ", ");
        print_rows("inverse", glm::inverse(transform), "");
        std::printf("}%s\n", (i == nfixed + nrand - 1) ? "" : ",");
    }
    std::printf("],\n");
    // ---- vector ops
    std::printf("\"vec\": [\n");
    const int nvec = 64;
    for (int i = 0; i < nvec; i++) {
        glm::vec3 a(rnd(-9, 9), rnd(-9, 9), rnd(-9, 9)), b(rnd(-9, 9), rnd(-9, 9), rnd(-9, 9));
        float s = rnd(-4, 4);
        std::printf("{");
        print_v3("a", a, ", ");
        print_v3("b", b, ", ");
        std::printf("\"s\": %u, ", bits(s));
        print_v3("normalize_a", glm::normalize(a), ", ");
        print_v3("cross_ab", glm::cross(a, b), ", ");
        print_v3("a_plus_s_times_norm_b", a + float(s - .0001f) * glm::normalize(b), ", ");
        std::printf("\"dot_ab\": %u, \"length_a\": %u, \"distance_ab\": %u", bits(glm::dot(a, b)), bits(glm::length(a)),
                    bits(glm::distance(a, b)));
        std::printf("}%s\n", (i == nvec - 1) ? "" : ",");
    }
    std::printf("]\n}\n");
    return 0;
}
