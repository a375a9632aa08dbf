// pt_device.h -- device-side math of the path tracer (gfx950 only).
//
// Every function here is the MI355X implementation of one function on the reference's hot path
// (ref = /root/reference): intersections.h, interactions.h and the device functions of
// raytraceKernel.cu.  Arithmetic is fp32, one rounding per operation (this file must be compiled with
// -ffp-contract=off and without fast-math): results are required to equal the CPU oracle's bit for bit,
// because a path tracer amplifies a 1-ulp difference into a different hit/miss decision.
// hipcc's fp32 '/' and sqrtf are correctly rounded by default (-fhip-fp32-correctly-rounded-divide-sqrt).
//
// Primitive records are wave-uniform: all 64 lanes test the same primitive at the same time, so the
// matrices are read once per wave (scalar loads into SGPRs, or one LDS broadcast read) and only the ray is
// per-lane data.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ptd {

struct f3 { float x, y, z; };

__device__ __forceinline__ f3 mk(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ f3 operator+(f3 a, f3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ f3 operator-(f3 a, f3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ f3 operator*(f3 a, f3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
__device__ __forceinline__ f3 operator*(float s, f3 v) { return mk(s * v.x, s * v.y, s * v.z); }
__device__ __forceinline__ f3 operator-(f3 a) { return mk(-a.x, -a.y, -a.z); }
// GLM 0.9.5.4 semantics (ref: external/include/glm/detail/func_geometric.inl:66-73,108-115,215-227,256-265)
__device__ __forceinline__ float dot(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ f3 cross(f3 x, f3 y)
{
    return mk(x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y);
}
// ---------------------------------------------------------------------------------------------
// Correctly rounded sqrt and reciprocal, shorter than hipcc's general expansions.
//
// hipcc expands sqrtf(x) to 16 and 1.0f/x to 11 VALU instructions because it must also cover denormal
// inputs/results, zero, inf and NaN.  Inside the usual range a v_sqrt_f32 / v_rcp_f32 seed plus the same
// FMA corrections is already exactly rounded: checked against hipcc's results for ALL 2^32 inputs
// (profiles/microbench/exact_math.hip, and pt_selftest_math() in every GPU test run):
//   sqrt_core(x) == sqrtf(x)   for every 2^-96 <= x < +inf
//   rcp_core(x)  == 1.0f / x   for every normal |x| < 2^126
// Outside those ranges the wave falls back to the compiler's expansion.  The range test is made
// wave-uniform with a ballot, so it costs one scalar branch and both sides stay straight-line code.
// ---------------------------------------------------------------------------------------------
// (round 3: v_rsq seed + one step on the product form, fma only -- round 2's v_sqrt seed was corrected by two compare /
// select pairs on exact residuals, and v_cmp / v_cndmask issue at 4 cycles against 2.5 for an fma: 19 instead of 35 cycles.
// Exhaustive check of the candidates: profiles/microbench/exact_sqrt2.hip, profiles/r03/exact_sqrt2.txt.)
__device__ __forceinline__ float sqrt_core(float x)
{
    const float s = __builtin_amdgcn_rsqf(x);                          // <= 1 ulp
    const float g = x * s, h = 0.5f * s;
    const float r = __builtin_fmaf(-g, g, x);                         // exact residual of the estimate g
    return __builtin_fmaf(r, h, g);
}
__device__ __forceinline__ float rcp_core(float d)
{
    const float r0 = __builtin_amdgcn_rcpf(d);                         // <= 1 ulp
    const float e0 = __builtin_fmaf(-d, r0, 1.0f);
    const float r1 = __builtin_fmaf(e0, r0, r0);
    const float e1 = __builtin_fmaf(-d, r1, 1.0f);
    return __builtin_fmaf(e1, r1, r1);
}
// The wave-uniform range tests are written with the compare builtins, whose result IS the lane mask (v_cmp into an
// SGPR pair): a ballot of a bool built from several compares costs an extra v_cndmask + v_cmp per test.
enum { FCMP_OGE = 3, FCMP_OLE = 5, FCMP_UGE = 11, FCMP_ULT = 12, ICMP_UGE = 35 };     // llvm::CmpInst predicates
__device__ __forceinline__ float sqrt_rn(float x)                      // == sqrtf(x) for every x
{
    // 2^-96 <= x < +inf as one unsigned compare on the bit pattern: tiny, zero, negative numbers, +inf and NaN fail it
    if (__builtin_amdgcn_uicmp(__float_as_uint(x) - 0x0F800000u, 0x7F800000u - 0x0F800000u, ICMP_UGE) == 0ull) return sqrt_core(x);
    return sqrtf(x);
}
__device__ __forceinline__ float rcp_rn(float x)                       // == 1.0f / x for every x
{
    const float ax = fabsf(x);
    if ((__builtin_amdgcn_fcmpf(ax, 0x1p-126f, FCMP_ULT) | __builtin_amdgcn_fcmpf(ax, 0x1p126f, FCMP_UGE)) == 0ull) return rcp_core(x);
    return 1.0f / x;
}
__device__ __forceinline__ float rsqrt_rn(float s)                     // == 1.0f / sqrtf(s) for every s
{
    // 2^-96 <= s < +inf (then sqrt(s) is in rcp_core's range), as one unsigned compare on the bit pattern:
    // negative numbers and NaNs wrap around to the top of the range and fail it
    if (__builtin_amdgcn_uicmp(__float_as_uint(s) - 0x0F800000u, 0x7F800000u - 0x0F800000u, ICMP_UGE) == 0ull)
        return rcp_core(sqrt_core(s));
    return 1.0f / sqrtf(s);
}

// 1.0f / sqrtf(s) for call sites whose argument is the squared length of an (almost) unit vector: getPointOnRay
// re-normalises a direction that is already normalised, and the second tangent of the hemisphere frame is the cross
// product of two orthogonal unit vectors.  Within 1024 ulp of 1.0 the doubly rounded result follows from the bit
// pattern alone (k = distance from 1.0 in ulps; sqrt halves it rounding towards 1 -- the series' second-order term
// breaks the ties --, the reciprocal mirrors it across 1.0 where the spacing changes by 2); checked for every input
// by pt_selftest_math.  No v_sqrt / v_rcp: 8 integer operations instead of ~22 with two quarter-rate ones.
__device__ __forceinline__ float rsqrt_near_one(float s)
{
    const uint32_t b = __float_as_uint(s);
    if (__builtin_amdgcn_uicmp(b - (0x3F800000u - 1024u), 2048u, 34 /* ICMP_UGT */) == 0ull) {
        const int k = (int)(b - 0x3F800000u);
        const uint32_t above = 0x3F800000u - ((uint32_t)k & ~1u);                                 // s >= 1
        const uint32_t below = 0x3F800000u + (((((uint32_t)(-k) + 1u) >> 1) + 1u) >> 1);          // s < 1
        return __uint_as_float(k >= 0 ? above : below);
    }
    return rsqrt_rn(s);
}
__device__ __forceinline__ f3 normalize_unit(f3 v)          // normalize() of a vector that is (almost) unit length already
{
    float inv = rsqrt_near_one(v.x * v.x + v.y * v.y + v.z * v.z);
    return mk(v.x * inv, v.y * inv, v.z * inv);
}

__device__ __forceinline__ float length(f3 v) { return sqrt_rn(v.x * v.x + v.y * v.y + v.z * v.z); }
__device__ __forceinline__ f3 normalize(f3 v)
{
    float inv = rsqrt_rn(v.x * v.x + v.y * v.y + v.z * v.z);
    return mk(v.x * inv, v.y * inv, v.z * inv);
}

// ---------------------------------------------------------------------------------------------
// Primitive record as the kernels see it: 32 dwords = 128 B, so one primitive is two s_load_dwordx16
// (or two LDS broadcast b128 x4 groups).  translation/rotation/scale of staticGeom are dropped (no
// intersection function reads them, ref: src/intersections.h:81-117).
// ---------------------------------------------------------------------------------------------
struct Prim {
    uint32_t type;       // 0 sphere, 1 cube, 2 mesh (never hit: its triangles are records of their own), 3 triangle
    uint32_t material;
    float area;          // surface area when the primitive is a light (direct lighting), else 0
    float self_r2;       // resident paths: may a ray that leaves this primitive on its outside skip it at its next bounce?  0: never.
                         // Cube (orthogonal transform, moderate size and place): any value > 0.  Sphere (uniformly scaled): the hit point must
                         // lie beyond this squared distance from the centre -- the reference's quadratic loses digits for rays that come
                         // from far away relative to the sphere's size and can report a hit point INSIDE it by more than the 0.0002 bias;
                         // the bounce then starts inside and the primitive IS met again (pt_context.hip)
    float inv[12];       // inverseTransform rows x,y,z (x y z w each)
    float fwd[12];       // transform rows x,y,z
    float cx, cy, cz;    // transform * (0,0,0,1), evaluated on the host with multiplyMV's operation order
    float bound_r2;      // squared radius of a padded world-space bounding sphere about (cx,cy,cz) (culling only)
};
static_assert(sizeof(Prim) == 128, "Prim must be 128 B");

// Node of the culling hierarchy used for large primitive lists: padded world-space box, depth-first order,
// `skip` = index of the next node when this subtree is missed (so traversal needs no stack), prim >= 0 on leaves.
struct BvhNode {
    float lo[3];
    uint32_t skip;
    float hi[3];
    int32_t prim;
};
static_assert(sizeof(BvhNode) == 32, "BvhNode must be 32 B");

// 4-wide hierarchy node (geom_path 7): 28 dwords = 112 B.  floats [8a, 8a+4) = lo planes of the four children on axis a,
// [8a+4, 8a+8) = their hi planes; dwords 24..27 = child words.  A lane reads the NEAR planes of all four children with one
// ds_read_b128 at (8a + (direction negative ? 4 : 0)) and the FAR planes at the other half: no min/max per slab.  The
// 28-dword stride walks the 64 LDS banks in steps of 4 with period 16, so the 16 lanes of a ds_read_b128 group reading
// the same field of different nodes rarely meet on a bank.
static constexpr int W4_FLOATS = 28;

// Material planes (SoA): plane k of material id at mats[k * nM + id]
enum { M_CR = 0, M_CG, M_CB, M_SR, M_SG, M_SB, M_REFL, M_REFR, M_IOR, M_EMIT, M_AR, M_AG, M_AB, M_SCAT, M_RSCT, M_PLANES };

// ---------------------------------------------------------------------------------------------
// RNG: Wang hash (ref: src/intersections.h:26-34) + thrust::minstd_rand + uniform_real_distribution<float>
// (call sites ref: src/raytraceKernel.cu:32-35).  Integer-exact, so streams are identical to the oracle's.
// ---------------------------------------------------------------------------------------------
__host__ __device__ __forceinline__ uint32_t wang_hash(uint32_t a)
{
    a = (a + 0x7ed55d16u) + (a << 12);
    a = (a ^ 0xc761c23cu) ^ (a >> 19);
    a = (a + 0x165667b1u) + (a << 5);
    a = (a + 0xd3a2646cu) ^ (a << 9);
    a = (a + 0xfd7046c5u) + (a << 3);
    a = (a ^ 0xb55a4f09u) ^ (a >> 16);
    return a;
}
// wave-uniform part of the stream key: hash(iteration ^ hash(key + seed*golden))
__host__ __device__ __forceinline__ uint32_t stream_key(uint32_t iteration, uint32_t key, uint32_t seed)
{
    return wang_hash(iteration ^ wang_hash(key + seed * 2654435769u));
}
__device__ __forceinline__ uint32_t minstd_seed(uint32_t s)
{
    // s % m for s < 2^32 = 2m + 2 needs at most two subtractions; each is "the smaller of s and s - m as unsigned" (s < m:
    // s - m wraps around to the top of the range; s >= m: s - m < s) -- v_sub + v_min instead of compare / select / subtract;
    // 0 -> 1 is a v_max.  Same value for every s (integer identity, checked for all 2^32 seeds by pt_selftest_math).
    const uint32_t m = 2147483647u;
    uint32_t w = s - m;
    s = w < s ? w : s;
    w = s - m;
    s = w < s ? w : s;
    return s > 1u ? s : 1u;
}
__device__ __forceinline__ uint32_t minstd_next(uint32_t x)
{
    const uint32_t m = 2147483647u;            // 48271*x mod (2^31-1) by Mersenne folding == Schrage's result
    uint64_t p = (uint64_t)x * 48271u;
    uint32_t r = (uint32_t)(p & m) + (uint32_t)(p >> 31);
    const uint32_t w = r - m;                  // r < 2m: the reduced value is the smaller of r and r - m as unsigned
    return w < r ? w : r;                      // (v_min_u32 instead of a compare + select)
}
// k steps of the engine at once: x * 48271^k mod (2^31 - 1), with 48271^k reduced on the host side of this header
// (MINSTD_A[k-1]).  The product of two values below 2^31 fits 62 bits, so the same fold applies.  The draws of a
// bounce are then independent of each other: a lobe that needs only xi1, xi2 does not run the engine through u_select,
// and the chain of dependent 64-bit multiplies is gone.  Values are those of k successive minstd_next calls.
__device__ __forceinline__ uint32_t minstd_jump(uint32_t x, uint32_t a_pow_k)
{
    const uint32_t m = 2147483647u;
    const uint64_t p = (uint64_t)x * a_pow_k;
    const uint32_t r = (uint32_t)(p & m) + (uint32_t)(p >> 31);      // < 2m
    const uint32_t w = r - m;
    return w < r ? w : r;
}
static constexpr uint32_t MINSTD_A1 = 48271u, MINSTD_A2 = 182605794u, MINSTD_A3 = 1291394886u, MINSTD_A4 = 1914720637u,
                          MINSTD_A5 = 2078669041u, MINSTD_A6 = 407355683u, MINSTD_A7 = 1105902161u;   // 48271^k mod (2^31 - 1)
__device__ __forceinline__ float u01_of(uint32_t x) { return (float)(x - 1u) / 2147483648.0f; }

// ---------------------------------------------------------------------------------------------
// deterministic sincos on [0, 2pi] (same polynomial, same operation order as the oracle's o_sincos_poly)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void sincos_poly(float a, float &sn, float &cs)
{
    int k = (int)(a * 0.636619772f + 0.5f);
    float fk = (float)k;
    float r = ((a - fk * 1.5703125f) - fk * 4.837512969970703125e-4f) - fk * 7.54978995489188216e-8f;
    float z = r * r;
    float s = ((-1.9515295891e-4f * z + 8.3321608736e-3f) * z - 1.6666654611e-1f) * z * r + r;
    float c = ((2.443315711809948e-5f * z - 1.388731625493765e-3f) * z + 4.166664568298827e-2f) * z * z
              - 0.5f * z + 1.0f;
    // quadrant q = k & 3:  sin = s, c, -s, -c;  cos = c, -s, -c, s.  One select per output on the quadrant's low bit and the
    // sign flipped through the bit pattern (a negation IS a flip of bit 31): 2 v_cndmask instead of 6 compare / select pairs
    const uint32_t q = (uint32_t)k & 3u;
    const bool odd = (q & 1u) != 0u;
    sn = __uint_as_float(__float_as_uint(odd ? c : s) ^ ((q & 2u) << 30));
    cs = __uint_as_float(__float_as_uint(odd ? s : c) ^ (((q + 1u) & 2u) << 30));
}

// multiplyMV (ref: src/intersections.h:53-59): rows 0..2 of a row-major 3x4 block dotted with (v, w)
__device__ __forceinline__ f3 mulMV(const float *m, f3 v, float w)
{
    f3 r;
    r.x = (m[0] * v.x) + (m[1] * v.y) + (m[2] * v.z) + (m[3] * w);
    r.y = (m[4] * v.x) + (m[5] * v.y) + (m[6] * v.z) + (m[7] * w);
    r.z = (m[8] * v.x) + (m[9] * v.y) + (m[10] * v.z) + (m[11] * w);
    return r;
}

// getPointOnRay (ref: src/intersections.h:46-48)
__device__ __forceinline__ f3 pointOnRay(f3 o, f3 d, float t) { return o + (t - .0001f) * normalize_unit(d); }

// ---------------------------------------------------------------------------------------------
// One primitive against one ray, split into the three stages the kernels schedule separately:
//   candidateT   object-space ray + parametric distance t (wave-uniform primitive, all lanes busy)
//   hitPoint     world-space hit point and distance for one (ray, primitive, t) candidate
//   hitNormal    world-space normal of the winning candidate
// sphere: ref src/intersections.h:81-117 (bit-faithful, including the double-precision radicand);
// cube:   the reference's stub (src/intersections.h:72-77) defined as a slab test in object space
//         (DESIGN.md "Canonical semantics", box).
// Box face code: bits 1..0 = axis, bit 2 = 1 when the object-space normal points along -axis.
// ---------------------------------------------------------------------------------------------
// RO_GIVEN: `ro` already holds inverseTransform*(origin,1) -- for camera rays it is the same for every lane and
// is evaluated once per primitive on the host with the same operation order.
// Triangle records (type 3, the flattened MESH geoms): inv[0..8] = v0, e1, e2 in WORLD space, fwd[0..2] = unit normal.
// Moeller-Trumbore, two-sided, on the ray with its direction normalised first (as the sphere test does); ro = the ray
// origin, t = distance along rd.  Same operations in the same order as the oracle's o_triangleIntersectionTest.
__device__ __forceinline__ bool candidateTriangle(const float *w, f3 o, f3 d, f3 &ro, f3 &rd, float &t)
{
    ro = o;
    rd = normalize(d);
    const f3 v0 = mk(w[0], w[1], w[2]), e1 = mk(w[3], w[4], w[5]), e2 = mk(w[6], w[7], w[8]);
    const f3 pv = cross(rd, e2);
    const float det = dot(e1, pv);
    if (fabsf(det) < 1e-12f) return false;
    const float inv = rcp_rn(det);
    const f3 tv = o - v0;
    const float u = dot(tv, pv) * inv;
    if (u < 0.0f || u > 1.0f) return false;
    const f3 qv = cross(tv, e1);
    const float v = dot(rd, qv) * inv;
    if (v < 0.0f || u + v > 1.0f) return false;
    t = dot(e2, qv) * inv;
    return t > 0.0f;
}

// TRI: the call site may meet triangle records (type 3); without it types 2 and 3 never hit (a MESH geom has no geometry of
// its own: its triangles are primitives of their own behind the geoms)
template <bool RO_GIVEN = false, bool TRI = false>
__device__ __forceinline__ bool candidateT(uint32_t type, const float *inv, f3 o, f3 d, f3 &ro, f3 &rd, float &t,
                                           uint32_t &face)
{
    face = 0u;
    t = 0.0f;
    if (TRI && type == 3u) return candidateTriangle(inv, o, d, ro, rd, t);
    if (type > 1u) return false;                         // MESH: parsed by the loader, never has geometry
    if (!RO_GIVEN) ro = mulMV(inv, o, 1.0f);
    rd = normalize(mulMV(inv, d, 0.0f));
    if (type == 0u) {
        float vDot = dot(ro, rd);
        float radicand = (float)((double)(vDot * vDot) - ((double)dot(ro, ro) - 0.25));
        if (radicand < 0) return false;
        float squareRoot = sqrt_rn(radicand);
        float firstTerm = -vDot;
        float t1 = firstTerm + squareRoot;
        float t2 = firstTerm - squareRoot;
        // ref lines 98-106: both negative -> miss; both positive -> min; otherwise max.  squareRoot >= 0 and rounding
        // is monotonic, so t2 <= t1 always and the three-way test collapses to this (same result for every input,
        // NaN included: then every compare is false on both forms and t = t1)
        if (t1 < 0) return false;
        t = (t2 > 0) ? t2 : t1;
        return true;
    }
    // Slab test.  Usual case (every lane's direction components normal and < 2^126, tested once per wave): exact
    // short reciprocals, min/max instead of compare+select pairs, entry/exit face found by equality afterwards.
    // Identical to the reference form below whenever no slab distance is NaN, which needs a zero direction
    // component -- excluded here -- or a non-finite ray.
    {
        const float ax = fabsf(rd.x), ay = fabsf(rd.y), az = fabsf(rd.z);
        if ((__builtin_amdgcn_fcmpf(fminf(fminf(ax, ay), az), 0x1p-126f, FCMP_ULT) |
             __builtin_amdgcn_fcmpf(fmaxf(fmaxf(ax, ay), az), 0x1p126f, FCMP_UGE)) == 0ull) {
            const float ix = rcp_core(rd.x), iy = rcp_core(rd.y), iz = rcp_core(rd.z);
            const float x0 = (-0.5f - ro.x) * ix, x1 = (0.5f - ro.x) * ix;
            const float y0 = (-0.5f - ro.y) * iy, y1 = (0.5f - ro.y) * iy;
            const float z0 = (-0.5f - ro.z) * iz, z1 = (0.5f - ro.z) * iz;
            const float nx = fminf(x0, x1), fx = fmaxf(x0, x1);
            const float ny = fminf(y0, y1), fy = fmaxf(y0, y1);
            const float nz = fminf(z0, z1), fz = fmaxf(z0, z1);
            const float tmin = fmaxf(fmaxf(nx, ny), nz), tmax = fminf(fminf(fx, fy), fz);
            if (tmax < tmin || tmax < 0) return false;
            const bool entry = tmin > 0;
            t = entry ? tmin : tmax;
            // lowest axis attaining the extremum == the axis the sequential strict compares below end on
            const bool isx = t == (entry ? nx : fx), isy = t == (entry ? ny : fy);
            const float da = isx ? rd.x : (isy ? rd.y : rd.z);
            face = (isx ? 0u : (isy ? 1u : 2u)) | (((da > 0) == entry) ? 4u : 0u);
            return true;
        }
    }
    float inv1, t0, t1, tn, tf;
    inv1 = rcp_rn(rd.x); t0 = (-0.5f - ro.x) * inv1; t1 = (0.5f - ro.x) * inv1;
    float tmin = (t0 < t1) ? t0 : t1, tmax = (t0 < t1) ? t1 : t0;
    int amin = 0, amax = 0;
    inv1 = rcp_rn(rd.y); t0 = (-0.5f - ro.y) * inv1; t1 = (0.5f - ro.y) * inv1;
    tn = (t0 < t1) ? t0 : t1; tf = (t0 < t1) ? t1 : t0;
    if (tn > tmin) { tmin = tn; amin = 1; }
    if (tf < tmax) { tmax = tf; amax = 1; }
    inv1 = rcp_rn(rd.z); t0 = (-0.5f - ro.z) * inv1; t1 = (0.5f - ro.z) * inv1;
    tn = (t0 < t1) ? t0 : t1; tf = (t0 < t1) ? t1 : t0;
    if (tn > tmin) { tmin = tn; amin = 2; }
    if (tf < tmax) { tmax = tf; amax = 2; }
    if (tmax < tmin || tmax < 0) return false;
    const bool entry = tmin > 0;
    t = entry ? tmin : tmax;
    const int axis = entry ? amin : amax;
    const float da = (axis == 0) ? rd.x : (axis == 1) ? rd.y : rd.z;
    const bool negative = entry ? (da > 0) : !(da > 0);   // entry face looks against the ray, exit face along it
    face = (uint32_t)axis | (negative ? 4u : 0u);
    return true;
}

// world-space hit point = transform * getPointOnRay(object ray, t); returns the distance from the ray origin
__device__ __forceinline__ float hitPoint(const float *fwd, f3 o, f3 ro, f3 rd, float t, f3 &real)
{
    real = mulMV(fwd, pointOnRay(ro, rd, t), 1.0f);
    return length(o - real);
}

__device__ __forceinline__ f3 sphereNormal(f3 real, f3 center) { return normalize(real - center); }
__device__ __forceinline__ f3 boxNormal(const float *fwd, uint32_t face)
{
    const uint32_t axis = face & 3u;
    const float sgn = (face & 4u) ? -1.0f : 1.0f;
    const f3 nobj = mk(axis == 0u ? sgn : 0.0f, axis == 1u ? sgn : 0.0f, axis == 2u ? sgn : 0.0f);
    return normalize(mulMV(fwd, nobj, 0.0f));
}

// the three stages back to back (direct path: hit work is done inside the wave-uniform primitive loop)
// a triangle's hit point: getPointOnRay on the normalised world ray, no transform
__device__ __forceinline__ float hitPointTriangle(f3 o, f3 ro, f3 rd, float t, f3 &real)
{
    real = pointOnRay(ro, rd, t);
    return length(o - real);
}

template <bool RO_GIVEN = false>
__device__ __forceinline__ float intersectPrim(const Prim &g, f3 o, f3 d, f3 ro_given, f3 &point, f3 &normal)
{
    f3 ro = ro_given, rd;
    float t;
    uint32_t face;
    if (!candidateT<RO_GIVEN, true>(g.type, g.inv, o, d, ro, rd, t, face)) return -1.0f;
    f3 real;
    if (g.type == 3u) {
        const float dist = hitPointTriangle(o, ro, rd, t, real);
        point = real;
        normal = mk(g.fwd[0], g.fwd[1], g.fwd[2]);
        return dist;
    }
    const float dist = hitPoint(g.fwd, o, ro, rd, t, real);
    point = real;
    if (g.type == 0u) normal = sphereNormal(real, mk(g.cx, g.cy, g.cz));
    else normal = boxNormal(g.fwd, face);
    return dist;
}

// calculateRandomDirectionInHemisphere (ref: src/interactions.h:62-87), deterministic trig
__device__ __forceinline__ f3 randomDirectionInHemisphere(f3 normal, float xi1, float xi2)
{
    float up = sqrt_rn(xi1);
    float over = sqrt_rn(1 - up * up);
    float around = (float)((double)xi2 * 6.2831853071795864769252867665590057683943);
    f3 notNormal;
    // ref: abs(normal.x) < SQRT_OF_ONE_THIRD with the double constant (src/interactions.h:68,73).  For a float f,
    // (double)f < 0.57735026918962576... holds exactly when f < 0.5773503184318542f, the smallest float above the
    // constant (float(constant) = 0.5773502588272095 lies below it): same decision, no double conversion / compare.
    if (fabsf(normal.x) < 0.5773503184318542f) notNormal = mk(1, 0, 0);
    else if (fabsf(normal.y) < 0.5773503184318542f) notNormal = mk(0, 1, 0);
    else notNormal = mk(0, 0, 1);
    f3 p1 = normalize(cross(normal, notNormal));
    f3 p2 = normalize_unit(cross(normal, p1));
    float sn, cs;
    sincos_poly(around, sn, cs);
    return ((up * normal) + ((cs * over) * p1)) + ((sn * over) * p2);
}

// calculateReflectionDirection (stub ref: src/interactions.h:47-50)
__device__ __forceinline__ f3 reflectionDirection(f3 normal, f3 incident)
{
    float k = 2.0f * dot(incident, normal);
    return incident - k * normal;
}

// calculateTransmissionDirection (stub ref: src/interactions.h:42-44); tir = total internal reflection
__device__ __forceinline__ f3 transmissionDirection(f3 normal, f3 incident, float n1, float n2, bool &tir)
{
    float eta = n1 / n2;
    float cosi = -dot(normal, incident);
    float sin2t = (eta * eta) * (1.0f - cosi * cosi);
    tir = sin2t > 1.0f;
    if (tir) return mk(0, 0, 0);
    float cost = sqrt_rn(1.0f - sin2t);
    float k = eta * cosi - cost;
    return (eta * incident) + (k * normal);
}

// calculateFresnel (stub ref: src/interactions.h:53-59): reflection coefficient of an unpolarised dielectric
__device__ __forceinline__ float fresnelReflectance(f3 normal, f3 incident, float n1, float n2, f3 trans)
{
    if (trans.x == 0.0f && trans.y == 0.0f && trans.z == 0.0f) return 1.0f;
    float cosi = -dot(normal, incident);
    float cost = -dot(normal, trans);
    float rs = (n1 * cosi - n2 * cost) / (n1 * cosi + n2 * cost);
    float rp = (n2 * cosi - n1 * cost) / (n2 * cosi + n1 * cost);
    return 0.5f * (rs * rs + rp * rp);
}

// ---------------------------------------------------------------------------------------------
// Sampling helpers of intersections.h / interactions.h / raytraceKernel.cu that are not on the pure
// path-tracing path but belong to the reference's device interface; exercised on the
// GPU by pt_device_kat() against the reference's golden vectors.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float uniform_real(uint32_t &state, float a, float b)    // thrust uniform_real_distribution<float>
{
    state = minstd_next(state);
    float r = (float)(state - 1u);
    r /= (1.0f + (float)(2147483646u - 1u));
    return (r * (b - a)) + a;
}

// generateRandomNumberFromThread (ref: src/raytraceKernel.cu:29-36): index and the seed product are float arithmetic
__device__ __forceinline__ f3 generateRandomNumberFromThread(float resx, float time, int x, int y)
{
    const int index = (int)((float)x + ((float)y * resx));
    uint32_t s = minstd_seed(wang_hash((uint32_t)((float)index * time)));
    const float a = uniform_real(s, 0, 1), b = uniform_real(s, 0, 1), c = uniform_real(s, 0, 1);
    return mk(a, b, c);
}

// getRadiuses (ref: src/intersections.h:120-129); glm::distance(p0,p1) = length(p1 - p0)
__device__ __forceinline__ f3 getRadiuses(const float *fwd)
{
    const f3 origin = mulMV(fwd, mk(0, 0, 0), 1.0f);
    const f3 xmax = mulMV(fwd, mk(.5f, 0, 0), 1.0f), ymax = mulMV(fwd, mk(0, .5f, 0), 1.0f), zmax = mulMV(fwd, mk(0, 0, .5f), 1.0f);
    return mk(length(xmax - origin), length(ymax - origin), length(zmax - origin));
}

// getRandomPointOnCube (ref: src/intersections.h:133-175): area-weighted face choice, then a point on the face;
// nobj = object-space normal of that face
__device__ __forceinline__ f3 cubePoint(const float *fwd, float randomSeed, f3 &nobj)
{
    uint32_t rng = minstd_seed(wang_hash((uint32_t)randomSeed));
    const f3 radii = getRadiuses(fwd);
    const float side1 = radii.x * radii.y * 4.0f, side2 = radii.z * radii.y * 4.0f, side3 = radii.x * radii.z * 4.0f;
    const float totalarea = 2.0f * (side1 + side2 + side3);
    const float russianRoulette = uniform_real(rng, 0, 1);
    const float a = uniform_real(rng, -0.5f, 0.5f), b = uniform_real(rng, -0.5f, 0.5f);
    f3 point;
    if (russianRoulette < (side1 / totalarea)) { point = mk(a, b, .5f); nobj = mk(0, 0, 1); }
    else if (russianRoulette < ((side1 * 2) / totalarea)) { point = mk(a, b, -.5f); nobj = mk(0, 0, -1); }
    else if (russianRoulette < (((side1 * 2) + (side2)) / totalarea)) { point = mk(.5f, a, b); nobj = mk(1, 0, 0); }
    else if (russianRoulette < (((side1 * 2) + (side2 * 2)) / totalarea)) { point = mk(-.5f, a, b); nobj = mk(-1, 0, 0); }
    else if (russianRoulette < (((side1 * 2) + (side2 * 2) + (side3)) / totalarea)) { point = mk(a, .5f, b); nobj = mk(0, 1, 0); }
    else { point = mk(a, -.5f, b); nobj = mk(0, -1, 0); }
    return mulMV(fwd, point, 1.0f);
}
__device__ __forceinline__ f3 getRandomPointOnCube(const float *fwd, float randomSeed)
{
    f3 nobj;
    return cubePoint(fwd, randomSeed, nobj);
}

// getRandomDirectionInSphere (stub ref: src/interactions.h:89-95): uniform direction, deterministic trig
__device__ __forceinline__ f3 getRandomDirectionInSphere(float xi1, float xi2)
{
    const float z = 1.0f - 2.0f * xi1;
    const float rr = 1.0f - z * z;
    const float rad = sqrt_rn(rr < 0.0f ? 0.0f : rr);
    const float around = (float)((double)xi2 * 6.2831853071795864769252867665590057683943);
    float sn, cs;
    sincos_poly(around, sn, cs);
    return mk(rad * cs, rad * sn, z);
}

// getRandomPointOnSphere (stub ref: src/intersections.h:177-182): uniform point on the object-space sphere r = .5
__device__ __forceinline__ f3 getRandomPointOnSphere(const float *fwd, float randomSeed)
{
    uint32_t rng = minstd_seed(wang_hash((uint32_t)randomSeed));
    const float xi1 = uniform_real(rng, 0, 1), xi2 = uniform_real(rng, 0, 1);
    const f3 d = getRandomDirectionInSphere(xi1, xi2);
    return mulMV(fwd, 0.5f * d, 1.0f);
}

// deterministic exp from fp32 + - * and floor only (Cody-Waite by ln 2 + cephes expf polynomial); the oracle runs the
// same sequence, so results agree bit for bit
__device__ __forceinline__ float exp_poly(float x)
{
    if (x < -87.0f) return 0.0f;
    if (x > 88.0f) x = 88.0f;
    const float kf = floorf(x * 1.44269504f + 0.5f);
    const int k = (int)kf;
    const float r = (x - kf * 0.693359375f) - kf * -2.12194440e-4f;
    const float z = r * r;
    const float p = ((((1.9875691500e-4f * r + 1.3981999507e-3f) * r + 8.3334519073e-3f) * r + 4.1665795894e-2f) * r +
                     1.6666665459e-1f) * r + 5.0000001201e-1f;
    const float y = (p * z + r) + 1.0f;
    return y * __uint_as_float((uint32_t)(k + 127) << 23);
}

// calculateTransmission (stub ref: src/interactions.h:31-33): Beer-Lambert transmittance per channel
__device__ __forceinline__ f3 calculateTransmission(f3 absorptionCoefficient, float distance)
{
    return mk(exp_poly(-absorptionCoefficient.x * distance), exp_poly(-absorptionCoefficient.y * distance),
              exp_poly(-absorptionCoefficient.z * distance));
}

// deterministic natural log of a positive normal fp32 (same sequence as the oracle's o_log_poly: cephes logf with the
// exponent taken from the bit pattern); x <= 0 -> -inf
__device__ __forceinline__ float log_poly(float x)
{
    uint32_t b = __float_as_uint(x);
    if (!(x > 0.0f)) return __uint_as_float(0xFF800000u);
    int e = (int)((b >> 23) & 0xFFu) - 126;
    float m = __uint_as_float((b & 0x807FFFFFu) | 0x3F000000u);
    if (m < 0.707106781186547524f) { e -= 1; m = (m + m) - 1.0f; }
    else m = m - 1.0f;
    float z = m * m;
    float y = ((((((((7.0376836292e-2f * m - 1.1514610310e-1f) * m + 1.1676998740e-1f) * m - 1.2420140846e-1f) * m
                   + 1.4249322787e-1f) * m - 1.6668057665e-1f) * m + 2.0000714765e-1f) * m - 2.4999993993e-1f) * m
               + 3.3333331174e-1f) * m * z;
    const float fe = (float)e;
    y = y + -2.12194440e-4f * fe;
    y = y + -0.5f * z;
    z = m + y;
    return z + 0.693359375f * fe;
}

// calculateScatterAndAbsorption (stub ref: src/interactions.h:36-39): one step of the random walk inside a scattering
// medium.  (o, d) starts inside and would reach the boundary after `depth`; free flight s = -ln(1 - u)/sigma_s'.
// true: scattered at o + s*d into a uniformly random direction, throughput absorbed over s, depth = s;
// false: boundary reached, throughput absorbed over the whole segment.
__device__ __forceinline__ bool calculateScatterAndAbsorption(f3 &o, f3 &d, float &depth, f3 absorptionCoefficient,
                                                              float reducedScatteringCoefficient, f3 &unabsorbedColor,
                                                              float randomFloatForScatteringDistance, float randomFloat2,
                                                              float randomFloat3)
{
    bool scattered = false;
    float s = 0.0f;
    if (reducedScatteringCoefficient > 0.0f) {
        const float a = 1.0f - randomFloatForScatteringDistance;
        if (a > 0.0f) {
            s = -log_poly(a) / reducedScatteringCoefficient;
            scattered = s < depth;
        }
    }
    if (scattered) {
        unabsorbedColor = unabsorbedColor * calculateTransmission(absorptionCoefficient, s);
        o = o + s * d;
        d = getRandomDirectionInSphere(randomFloat2, randomFloat3);
        depth = s;
        return true;
    }
    unabsorbedColor = unabsorbedColor * calculateTransmission(absorptionCoefficient, depth);
    return false;
}

// getRandomPointOnCube with the five face-choice thresholds (side areas over the total) and the face normals taken from
// host-built tables: they depend on the light only (pt_context.hip evaluates them with this file's operation order).
// tab = the primitive's 8-entry face table: entries 0-2 / 4-6 hold the +axis / -axis world normals, entry 3 holds the
// thresholds t1..t4, entry 7.x holds t5.
__device__ __forceinline__ void sampleCubeLightTab(const float *fwd, const float4 *tab, float randomSeed, f3 &point, f3 &normal)
{
    uint32_t rng = minstd_seed(wang_hash((uint32_t)randomSeed));
    const float4 t = tab[3];
    const float t5 = tab[7].x;
    const float russianRoulette = uniform_real(rng, 0, 1);
    const float a = uniform_real(rng, -0.5f, 0.5f), b = uniform_real(rng, -0.5f, 0.5f);
    f3 pobj;
    uint32_t face;
    if (russianRoulette < t.x) { pobj = mk(a, b, .5f); face = 2u; }
    else if (russianRoulette < t.y) { pobj = mk(a, b, -.5f); face = 6u; }
    else if (russianRoulette < t.z) { pobj = mk(.5f, a, b); face = 0u; }
    else if (russianRoulette < t.w) { pobj = mk(-.5f, a, b); face = 4u; }
    else if (russianRoulette < t5) { pobj = mk(a, .5f, b); face = 1u; }
    else { pobj = mk(a, -.5f, b); face = 5u; }
    point = mulMV(fwd, pobj, 1.0f);
    const float4 n = tab[face];
    normal = mk(n.x, n.y, n.z);
}

// mesh lights (direct lighting): a uniform point on the triangle v0, v0 + e1, v0 + e2 from two uniform numbers
// (square-root parametrisation) and the triangle's area; same operations as the oracle's o_sampleTriangle / o_triangleArea
__device__ __forceinline__ f3 sampleTriangle(f3 v0, f3 e1, f3 e2, float u_a, float u_b)
{
    const float sq = sqrt_rn(u_a);
    const float ba = sq * (1.0f - u_b), bb = sq * u_b;
    return (v0 + ba * e1) + bb * e2;
}
__device__ __forceinline__ float triangleArea(f3 e1, f3 e2) { return 0.5f * length(cross(e1, e2)); }

// Direct lighting: a point on a light and the geometric normal there, from one float seed (the reference's sampler
// interface).  Normals as the intersection tests define them (boxNormal / sphereNormal).
__device__ __forceinline__ void sampleLight(uint32_t type, const float *fwd, f3 center, float randomSeed, f3 &point, f3 &normal)
{
    if (type == 1u) {
        f3 nobj;
        point = cubePoint(fwd, randomSeed, nobj);
        normal = normalize(mulMV(fwd, nobj, 0.0f));
    } else {
        point = getRandomPointOnSphere(fwd, randomSeed);
        normal = normalize(point - center);
    }
}

}  // namespace ptd
