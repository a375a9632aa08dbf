/*
 * pt_oracle.c -- CPU oracle (TEST INFRASTRUCTURE ONLY, see pt_oracle.h).
 *
 * Citations "ref:" are relative to /root/reference/.  "spec:" marks parts the
 * reference ships as TODO stubs; their semantics are defined in DESIGN.md
 * ("Canonical semantics") and this file is the normative statement.
 *
 * Build: gcc -O2 -std=c99 -ffp-contract=off -fno-fast-math (see Makefile).
 */
#include "pt_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#define O_PI 3.1415926535897932384626422832795028841971            /* ref: src/utilities.h:20 (double) */
#define O_TWO_PI 6.2831853071795864769252867665590057683943        /* ref: src/utilities.h:21 (double) */
#define O_SQRT_OF_ONE_THIRD 0.5773502691896257645091487805019574556476 /* ref: src/utilities.h:22 (double) */
#define O_EPSILON .000000001                                        /* ref: src/utilities.h:24 (double) */
#define O_RAY_BIAS_AMOUNT 0.0002f                                   /* ref: src/utilities.h:26, used as fp32 */

/* ------------------------------------------------------------------ */
/* GLM 0.9.5.4 vector arithmetic, restated (fp32, operation order kept) */
/* ------------------------------------------------------------------ */
static o_vec3 v3(float x, float y, float z) { o_vec3 r; r.x = x; r.y = y; r.z = z; return r; }
static o_vec4 v4(o_vec3 v, float w) { o_vec4 r; r.x = v.x; r.y = v.y; r.z = v.z; r.w = w; return r; }
static o_vec3 add3(o_vec3 a, o_vec3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
static o_vec3 sub3(o_vec3 a, o_vec3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
static o_vec3 mul3(o_vec3 a, o_vec3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }
static o_vec3 scale3(float s, o_vec3 v) { return v3(s * v.x, s * v.y, s * v.z); } /* float*vec3 == vec3*float */
static o_vec3 neg3(o_vec3 a) { return v3(-a.x, -a.y, -a.z); }
/* ref: external/include/glm/detail/func_geometric.inl:66-73  tmp = x*y; tmp.x + tmp.y + tmp.z */
static float dot3(o_vec3 a, o_vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
/* ref: func_geometric.inl:215-227 */
static o_vec3 cross3(o_vec3 x, o_vec3 y)
{
    return v3(x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y);
}
/* ref: func_geometric.inl:108-115  sqrt(x*x + y*y + z*z) */
static float length3(o_vec3 v) { float sqr = v.x * v.x + v.y * v.y + v.z * v.z; return sqrtf(sqr); }
/* ref: func_geometric.inl:256-265 + func_exponential.inl:226-229  x * (1.0f / sqrt(sqr)) */
static o_vec3 normalize3(o_vec3 v)
{
    float sqr = v.x * v.x + v.y * v.y + v.z * v.z;
    float inv = 1.0f / sqrtf(sqr);
    return v3(v.x * inv, v.y * inv, v.z * inv);
}

/* ------------------------------------------------------------------ */
/* RNG                                                                 */
/* ------------------------------------------------------------------ */
/* ref: src/intersections.h:26-34 */
unsigned o_hash(unsigned a)
{
    a = (a + 0x7ed55d16u) + (a << 12);
    a = (a ^ 0xc761c23cu) ^ (a >> 19);
    a = (a + 0x165667b1u) + (a << 5);
    a = (a + 0xd3a2646cu) ^ (a << 9);
    a = (a + 0xfd7046c5u) + (a << 3);
    a = (a ^ 0xb55a4f09u) ^ (a >> 16);
    return a;
}

/* thrust::default_random_engine == minstd_rand == LCG(a=48271, c=0, m=2^31-1), third-party (CUDA toolkit
 * thrust, not vendored; call sites ref: src/raytraceKernel.cu:32-35, src/intersections.h:135-137).
 * Restated from rocThrust 2.8.5 thrust/random/detail/linear_congruential_engine.inl (seed) and
 * thrust/random/detail/mod.h (Schrage's method). */
#define O_MINSTD_A 48271u
#define O_MINSTD_M 2147483647u
void o_minstd_seed(unsigned *state, unsigned s)
{
    unsigned x = s % O_MINSTD_M;
    if (x == 0u) x = 1u % O_MINSTD_M;
    *state = x;
}
unsigned o_minstd_next(unsigned *state)
{
    const unsigned q = O_MINSTD_M / O_MINSTD_A, r = O_MINSTD_M % O_MINSTD_A;
    unsigned x = *state;
    unsigned t1 = O_MINSTD_A * (x % q);
    unsigned t2 = r * (x / q);
    if (t1 >= t2) x = t1 - t2; else x = O_MINSTD_M - t2 + t1;
    *state = x;
    return x;
}
/* thrust::uniform_real_distribution<float>(a,b), thrust/random/detail/uniform_real_distribution.inl:
 * result = float(urng() - min); result /= (1.0f + float(max - min)); return result*(b-a) + a */
float o_uniform_real(unsigned *state, float a, float b)
{
    float result = (float)(o_minstd_next(state) - 1u);
    result /= (1.0f + (float)(2147483646u - 1u));
    return (result * (b - a)) + a;
}
float o_u01(unsigned *state) { return o_uniform_real(state, 0.0f, 1.0f); }

/* spec: one minstd stream per (global pixel, iteration, key); key 0 = camera jitter,
 * key b+1 = bounce b.  Integer-only, so identical on every device and independent of ray slot. */
unsigned o_stream_seed(unsigned pixel, unsigned iteration, unsigned key, unsigned seed)
{
    return o_hash(pixel ^ o_hash(iteration ^ o_hash(key + seed * 2654435769u)));
}

/* ------------------------------------------------------------------ */
/* intersections.h                                                     */
/* ------------------------------------------------------------------ */
/* ref: src/intersections.h:37-43 (EPSILON is a double literal: the compare is done in double) */
int o_epsilonCheck(float a, float b)
{
    return ((double)fabsf(fabsf(a) - fabsf(b)) < O_EPSILON) ? 1 : 0;
}

/* ref: src/intersections.h:46-48 */
o_vec3 o_getPointOnRay(o_ray r, float t)
{
    return add3(r.origin, scale3((float)(t - .0001f), normalize3(r.direction)));
}

/* ref: src/intersections.h:53-59 (row w of the matrix is never read) */
o_vec3 o_multiplyMV(o_mat4 m, o_vec4 v)
{
    o_vec3 r;
    r.x = (m.x.x * v.x) + (m.x.y * v.y) + (m.x.z * v.z) + (m.x.w * v.w);
    r.y = (m.y.x * v.x) + (m.y.y * v.y) + (m.y.z * v.z) + (m.y.w * v.w);
    r.z = (m.z.x * v.x) + (m.z.y * v.y) + (m.z.z * v.z) + (m.z.w * v.w);
    return r;
}

/* ref: src/intersections.h:62-64 (1.0 is a double literal: divide in double, store as fp32) */
o_vec3 o_getInverseDirectionOfRay(o_ray r)
{
    return v3((float)(1.0 / (double)r.direction.x), (float)(1.0 / (double)r.direction.y),
              (float)(1.0 / (double)r.direction.z));
}

/* ref: src/intersections.h:67-70 */
o_vec3 o_getSignOfRay(o_ray r)
{
    o_vec3 inv = o_getInverseDirectionOfRay(r);
    return v3((float)(int)(inv.x < 0), (float)(int)(inv.y < 0), (float)(int)(inv.z < 0));
}

/* ref: src/intersections.h:81-117 */
float o_sphereIntersectionTest(const o_staticGeom *sphere, o_ray r, o_vec3 *intersectionPoint, o_vec3 *normal)
{
    float radius = .5f;

    o_vec3 ro = o_multiplyMV(sphere->inverseTransform, v4(r.origin, 1.0f));
    o_vec3 rd = normalize3(o_multiplyMV(sphere->inverseTransform, v4(r.direction, 0.0f)));

    o_ray rt; rt.origin = ro; rt.direction = rd;

    float vDotDirection = dot3(rt.origin, rt.direction);
    /* pow(radius, 2) is std::pow(float,int) -> double (C++11), so the bracket and the outer subtraction
     * are evaluated in double and rounded once on assignment (ref line 90). */
    float radicand = (float)((double)(vDotDirection * vDotDirection) -
                             ((double)dot3(rt.origin, rt.origin) - pow((double)radius, 2.0)));
    if (radicand < 0) return -1;

    float squareRoot = sqrtf(radicand);
    float firstTerm = -vDotDirection;
    float t1 = firstTerm + squareRoot;
    float t2 = firstTerm - squareRoot;

    float t = 0;
    if (t1 < 0 && t2 < 0) {
        return -1;
    } else if (t1 > 0 && t2 > 0) {
        t = (t2 < t1) ? t2 : t1;          /* std::min(t1, t2) */
    } else {
        t = (t1 < t2) ? t2 : t1;          /* std::max(t1, t2) */
    }

    o_vec3 realIntersectionPoint = o_multiplyMV(sphere->transform, v4(o_getPointOnRay(rt, t), 1.0f));
    o_vec3 realOrigin = o_multiplyMV(sphere->transform, v4(v3(0, 0, 0), 1.0f));

    *intersectionPoint = realIntersectionPoint;
    *normal = normalize3(sub3(realIntersectionPoint, realOrigin));

    return length3(sub3(r.origin, realIntersectionPoint));
}

/* spec (ref stub: src/intersections.h:72-77; "should work in the same way as sphereIntersectionTest",
 * README.md:121-123): unit cube [-.5,.5]^3 in object space, slab test.  min/max are written as explicit
 * compares so that inf/NaN slabs behave identically on every target. */
float o_boxIntersectionTest(const o_staticGeom *box, o_ray r, o_vec3 *intersectionPoint, o_vec3 *normal)
{
    o_vec3 ro = o_multiplyMV(box->inverseTransform, v4(r.origin, 1.0f));
    o_vec3 rd = normalize3(o_multiplyMV(box->inverseTransform, v4(r.direction, 0.0f)));
    o_ray rt; rt.origin = ro; rt.direction = rd;

    const float o[3] = { ro.x, ro.y, ro.z };
    const float d[3] = { rd.x, rd.y, rd.z };
    float tmin = 0, tmax = 0;
    int amin = 0, amax = 0;
    for (int a = 0; a < 3; a++) {
        float inv = 1.0f / d[a];
        float t0 = (-0.5f - o[a]) * inv;
        float t1 = (0.5f - o[a]) * inv;
        float tn = (t0 < t1) ? t0 : t1;
        float tf = (t0 < t1) ? t1 : t0;
        if (a == 0 || tn > tmin) { tmin = tn; amin = a; }
        if (a == 0 || tf < tmax) { tmax = tf; amax = a; }
    }
    if (tmax < tmin || tmax < 0) return -1;

    float t; int axis; float sgn;
    if (tmin > 0) { t = tmin; axis = amin; sgn = (d[axis] > 0) ? -1.0f : 1.0f; }   /* entry face */
    else          { t = tmax; axis = amax; sgn = (d[axis] > 0) ? 1.0f : -1.0f; }   /* exit face (origin inside) */

    o_vec3 n_obj = v3(axis == 0 ? sgn : 0.0f, axis == 1 ? sgn : 0.0f, axis == 2 ? sgn : 0.0f);

    o_vec3 realIntersectionPoint = o_multiplyMV(box->transform, v4(o_getPointOnRay(rt, t), 1.0f));
    *intersectionPoint = realIntersectionPoint;
    *normal = normalize3(o_multiplyMV(box->transform, v4(n_obj, 0.0f)));
    return length3(sub3(r.origin, realIntersectionPoint));
}

/* ref: src/intersections.h:120-129; glm::distance(p0,p1) = length(p1 - p0) */
o_vec3 o_getRadiuses(const o_staticGeom *geom)
{
    o_vec3 origin = o_multiplyMV(geom->transform, v4(v3(0, 0, 0), 1.0f));
    o_vec3 xmax = o_multiplyMV(geom->transform, v4(v3(.5f, 0, 0), 1.0f));
    o_vec3 ymax = o_multiplyMV(geom->transform, v4(v3(0, .5f, 0), 1.0f));
    o_vec3 zmax = o_multiplyMV(geom->transform, v4(v3(0, 0, .5f), 1.0f));
    return v3(length3(sub3(xmax, origin)), length3(sub3(ymax, origin)), length3(sub3(zmax, origin)));
}

/* ref: src/intersections.h:133-175; also reports the object-space normal of the chosen face (direct lighting) */
static o_vec3 cube_point(const o_staticGeom *cube, float randomSeed, o_vec3 *face_normal_obj)
{
    unsigned rng; o_minstd_seed(&rng, o_hash((unsigned)randomSeed));

    o_vec3 radii = o_getRadiuses(cube);
    float side1 = radii.x * radii.y * 4.0f;
    float side2 = radii.z * radii.y * 4.0f;
    float side3 = radii.x * radii.z * 4.0f;
    float totalarea = 2.0f * (side1 + side2 + side3);

    float russianRoulette = o_uniform_real(&rng, 0, 1);

    o_vec3 point, nobj;
    if (russianRoulette < (side1 / totalarea)) {
        float a = o_uniform_real(&rng, -0.5f, 0.5f), b = o_uniform_real(&rng, -0.5f, 0.5f);
        point = v3(a, b, .5f); nobj = v3(0, 0, 1);
    } else if (russianRoulette < ((side1 * 2) / totalarea)) {
        float a = o_uniform_real(&rng, -0.5f, 0.5f), b = o_uniform_real(&rng, -0.5f, 0.5f);
        point = v3(a, b, -.5f); nobj = v3(0, 0, -1);
    } else if (russianRoulette < (((side1 * 2) + (side2)) / totalarea)) {
        float a = o_uniform_real(&rng, -0.5f, 0.5f), b = o_uniform_real(&rng, -0.5f, 0.5f);
        point = v3(.5f, a, b); nobj = v3(1, 0, 0);
    } else if (russianRoulette < (((side1 * 2) + (side2 * 2)) / totalarea)) {
        float a = o_uniform_real(&rng, -0.5f, 0.5f), b = o_uniform_real(&rng, -0.5f, 0.5f);
        point = v3(-.5f, a, b); nobj = v3(-1, 0, 0);
    } else if (russianRoulette < (((side1 * 2) + (side2 * 2) + (side3)) / totalarea)) {
        float a = o_uniform_real(&rng, -0.5f, 0.5f), b = o_uniform_real(&rng, -0.5f, 0.5f);
        point = v3(a, .5f, b); nobj = v3(0, 1, 0);
    } else {
        float a = o_uniform_real(&rng, -0.5f, 0.5f), b = o_uniform_real(&rng, -0.5f, 0.5f);
        point = v3(a, -.5f, b); nobj = v3(0, -1, 0);
    }
    if (face_normal_obj) *face_normal_obj = nobj;
    return o_multiplyMV(cube->transform, v4(point, 1.0f));
}
o_vec3 o_getRandomPointOnCube(const o_staticGeom *cube, float randomSeed) { return cube_point(cube, randomSeed, NULL); }

/* spec (DESIGN.md "Direct lighting"): surface area of a light, from getRadiuses' half-extents.  Cube: the
 * total area getRandomPointOnCube weights its faces with.  Sphere (r = .5 in object space): 4pi(ab+ac+bc)/3,
 * exact for uniform scale. */
float o_lightArea(const o_staticGeom *g)
{
    o_vec3 r = o_getRadiuses(g);
    if (g->type == O_CUBE) {
        float side1 = r.x * r.y * 4.0f, side2 = r.z * r.y * 4.0f, side3 = r.x * r.z * 4.0f;
        return 2.0f * (side1 + side2 + side3);
    }
    return 4.18879020478639098f * ((r.x * r.y + r.x * r.z) + r.y * r.z);
}

/* spec: a point on light `g` and the geometric normal there, from one float seed (the reference's sampler
 * interface, src/intersections.h:133,179) */
void o_sampleLight(const o_staticGeom *g, float randomSeed, o_vec3 *point, o_vec3 *normal)
{
    if (g->type == O_CUBE) {
        o_vec3 nobj;
        *point = cube_point(g, randomSeed, &nobj);
        *normal = normalize3(o_multiplyMV(g->transform, v4(nobj, 0.0f)));         /* as the box test's normal */
    } else {
        *point = o_getRandomPointOnSphere(g, randomSeed);
        *normal = normalize3(sub3(*point, o_multiplyMV(g->transform, v4(v3(0, 0, 0), 1.0f))));   /* as the sphere test's */
    }
}

/* spec (ref stub: src/intersections.h:177-182): uniform point on the object-space sphere r=.5 */
o_vec3 o_getRandomPointOnSphere(const o_staticGeom *sphere, float randomSeed)
{
    unsigned rng; o_minstd_seed(&rng, o_hash((unsigned)randomSeed));
    float xi1 = o_uniform_real(&rng, 0, 1), xi2 = o_uniform_real(&rng, 0, 1);
    o_vec3 d = o_getRandomDirectionInSphere(xi1, xi2, O_TRIG_POLY);
    return o_multiplyMV(sphere->transform, v4(scale3(0.5f, d), 1.0f));
}

/* ------------------------------------------------------------------ */
/* interactions.h                                                      */
/* ------------------------------------------------------------------ */
/* spec: deterministic sincos for a in [0, 2*pi] built from fp32 + - * only (3-term Cody-Waite reduction
 * by pi/2, cephes minimax polynomials), so the HIP kernels and this oracle agree bit for bit.  glibc and
 * ROCm OCML sinf/cosf differ in the last place, which a path tracer amplifies into different hit/miss
 * decisions; max abs error of this routine vs libm on [0,2pi] is < 2.4e-7 (tests/test_oracle_kat.py). */
void o_sincos_poly(float a, float *s_out, float *c_out)
{
    int k = (int)(a * 0.636619772f + 0.5f);
    float fk = (float)k;
    float r = ((a - fk * 1.5703125f) - fk * 4.837512969970703125e-4f) - fk * 7.54978995489188216e-8f;
    float z = r * r;
    float s = ((-1.9515295891e-4f * z + 8.3321608736e-3f) * z - 1.6666654611e-1f) * z * r + r;
    float c = ((2.443315711809948e-5f * z - 1.388731625493765e-3f) * z + 4.166664568298827e-2f) * z * z
              - 0.5f * z + 1.0f;
    switch (k & 3) {
    case 0: *s_out = s;  *c_out = c;  break;
    case 1: *s_out = c;  *c_out = -s; break;
    case 2: *s_out = -s; *c_out = -c; break;
    default: *s_out = -c; *c_out = s; break;
    }
}

/* ref: src/interactions.h:62-87 */
o_vec3 o_calculateRandomDirectionInHemisphere(o_vec3 normal, float xi1, float xi2, int trig_mode)
{
    float up = sqrtf(xi1);
    float over = sqrtf(1 - up * up);
    float around = (float)((double)xi2 * O_TWO_PI);          /* TWO_PI is a double literal */

    o_vec3 directionNotNormal;
    if ((double)fabsf(normal.x) < O_SQRT_OF_ONE_THIRD) {      /* compared in double */
        directionNotNormal = v3(1, 0, 0);
    } else if ((double)fabsf(normal.y) < O_SQRT_OF_ONE_THIRD) {
        directionNotNormal = v3(0, 1, 0);
    } else {
        directionNotNormal = v3(0, 0, 1);
    }

    o_vec3 perpendicularDirection1 = normalize3(cross3(normal, directionNotNormal));
    o_vec3 perpendicularDirection2 = normalize3(cross3(normal, perpendicularDirection1));

    float sn, cs;
    if (trig_mode == O_TRIG_LIBM) { cs = cosf(around); sn = sinf(around); }
    else o_sincos_poly(around, &sn, &cs);

    return add3(add3(scale3(up, normal), scale3(cs * over, perpendicularDirection1)),
                scale3(sn * over, perpendicularDirection2));
}

/* spec (ref stub: src/interactions.h:89-95): uniform direction on the unit sphere */
o_vec3 o_getRandomDirectionInSphere(float xi1, float xi2, int trig_mode)
{
    float z = 1.0f - 2.0f * xi1;
    float rr = 1.0f - z * z;
    float rad = sqrtf(rr < 0.0f ? 0.0f : rr);
    float around = (float)((double)xi2 * O_TWO_PI);
    float sn, cs;
    if (trig_mode == O_TRIG_LIBM) { cs = cosf(around); sn = sinf(around); }
    else o_sincos_poly(around, &sn, &cs);
    return v3(rad * cs, rad * sn, z);
}

/* spec: deterministic exp built from fp32 + - * and floor only (Cody-Waite reduction by ln 2, cephes expf
 * polynomial), so the HIP kernels and this oracle agree bit for bit; relative error < 2e-7 on [-87, 88]. */
float o_exp_poly(float x)
{
    if (x < -87.0f) return 0.0f;
    if (x > 88.0f) x = 88.0f;
    float kf = floorf(x * 1.44269504f + 0.5f);
    int k = (int)kf;
    float r = (x - kf * 0.693359375f) - kf * -2.12194440e-4f;
    float z = r * r;
    float p = ((((1.9875691500e-4f * r + 1.3981999507e-3f) * r + 8.3334519073e-3f) * r + 4.1665795894e-2f) * r
               + 1.6666665459e-1f) * r + 5.0000001201e-1f;
    float y = (p * z + r) + 1.0f;
    union { unsigned u; float f; } s;
    s.u = (unsigned)(k + 127) << 23;                     /* 2^k, k in [-126, 127] on the clamped range */
    return y * s.f;
}

/* spec (ref stub: src/interactions.h:31-33): Beer-Lambert transmittance exp(-sigma_a * distance) per channel */
o_vec3 o_calculateTransmission(o_vec3 absorptionCoefficient, float distance)
{
    return v3(o_exp_poly(-absorptionCoefficient.x * distance), o_exp_poly(-absorptionCoefficient.y * distance),
              o_exp_poly(-absorptionCoefficient.z * distance));
}

/* spec: deterministic natural logarithm of a positive normal fp32 from integer exponent extraction and fp32 + - * only
 * (cephes logf: mantissa folded to [sqrt(1/2), sqrt(2)), degree-9 polynomial, Cody-Waite ln 2), so the HIP kernels and
 * this oracle agree bit for bit; relative error < 2e-7.  x <= 0 (only 1 - u01 == 0 reaches it) -> -inf. */
float o_log_poly(float x)
{
    union { float f; unsigned u; } b;
    b.f = x;
    if (!(x > 0.0f)) { b.u = 0xFF800000u; return b.f; }
    int e = (int)((b.u >> 23) & 0xFFu) - 126;
    b.u = (b.u & 0x807FFFFFu) | 0x3F000000u;              /* mantissa in [0.5, 1) */
    float m = b.f;
    if (m < 0.707106781186547524f) { e -= 1; m = (m + m) - 1.0f; }
    else m = m - 1.0f;
    float z = m * m;
    float y = ((((((((7.0376836292e-2f * m - 1.1514610310e-1f) * m + 1.1676998740e-1f) * m - 1.2420140846e-1f) * m
                   + 1.4249322787e-1f) * m - 1.6668057665e-1f) * m + 2.0000714765e-1f) * m - 2.4999993993e-1f) * m
               + 3.3333331174e-1f) * m * z;
    float fe = (float)e;
    y = y + -2.12194440e-4f * fe;
    y = y + -0.5f * z;
    z = m + y;
    return z + 0.693359375f * fe;
}

/* spec (ref stub: src/interactions.h:36-39, fields src/sceneStructs.h:63-74, README.md:179-184): one step of the random
 * walk inside a scattering medium.  `r` starts inside the medium and would reach its boundary after *depth (world
 * distance of the nearest hit).  Free flight s = -ln(1 - u)/sigma_s' with the REDUCED scattering coefficient; s < *depth:
 * the path scatters at r.origin + s*r.direction into a uniformly random direction (getRandomDirectionInSphere of the two
 * other draws), Beer-Lambert absorption over s -- returns 1 with r, *depth = s updated; otherwise Beer-Lambert over
 * the whole segment, returns 0 and the caller handles the boundary.  `m` is the medium's material (unused beyond the
 * properties, as in the reference's signature). */
int o_calculateScatterAndAbsorption(o_ray *r, float *depth, const o_AbsorptionAndScatteringProperties *cur,
                                    o_vec3 *unabsorbedColor, const o_material *m, float randomFloatForScatteringDistance,
                                    float randomFloat2, float randomFloat3)
{
    (void)m;
    const float sigma = cur->reducedScatteringCoefficient;
    int scattered = 0;
    float s = 0.0f;
    if (sigma > 0.0f) {
        const float a = 1.0f - randomFloatForScatteringDistance;
        if (a > 0.0f) {
            s = -o_log_poly(a) / sigma;
            scattered = s < *depth;
        }
    }
    if (scattered) {
        *unabsorbedColor = mul3(*unabsorbedColor, o_calculateTransmission(cur->absorptionCoefficient, s));
        r->origin = add3(r->origin, scale3(s, r->direction));
        r->direction = o_getRandomDirectionInSphere(randomFloat2, randomFloat3, O_TRIG_POLY);
        *depth = s;
        return 1;
    }
    *unabsorbedColor = mul3(*unabsorbedColor, o_calculateTransmission(cur->absorptionCoefficient, *depth));
    return 0;
}

/* spec (ref stub: src/interactions.h:47-50): mirror law d - 2(d.n)n */
o_vec3 o_calculateReflectionDirection(o_vec3 normal, o_vec3 incident)
{
    float k = 2.0f * dot3(incident, normal);
    return sub3(incident, scale3(k, normal));
}

/* spec (ref stub: src/interactions.h:42-44): Snell refraction; `normal` faces the incident side
 * (dot(normal, incident) <= 0), `incident` is unit length.  Total internal reflection -> (0,0,0). */
o_vec3 o_calculateTransmissionDirection(o_vec3 normal, o_vec3 incident, float incidentIOR, float transmittedIOR)
{
    float eta = incidentIOR / transmittedIOR;
    float cosi = -dot3(normal, incident);
    float sin2t = (eta * eta) * (1.0f - cosi * cosi);
    if (sin2t > 1.0f) return v3(0, 0, 0);
    float cost = sqrtf(1.0f - sin2t);
    float k = eta * cosi - cost;
    return add3(scale3(eta, incident), scale3(k, normal));
}

/* spec (ref stub: src/interactions.h:53-59): exact unpolarised dielectric Fresnel, 0.5*(rs^2 + rp^2) */
o_Fresnel o_calculateFresnel(o_vec3 normal, o_vec3 incident, float incidentIOR, float transmittedIOR,
                             o_vec3 reflectionDirection, o_vec3 transmissionDirection)
{
    o_Fresnel f;
    (void)reflectionDirection;
    if (transmissionDirection.x == 0.0f && transmissionDirection.y == 0.0f && transmissionDirection.z == 0.0f) {
        f.reflectionCoefficient = 1.0f; f.transmissionCoefficient = 0.0f;     /* total internal reflection */
        return f;
    }
    float cosi = -dot3(normal, incident);
    float cost = -dot3(normal, transmissionDirection);
    float rs = (incidentIOR * cosi - transmittedIOR * cost) / (incidentIOR * cosi + transmittedIOR * cost);
    float rp = (transmittedIOR * cosi - incidentIOR * cost) / (transmittedIOR * cosi + incidentIOR * cost);
    f.reflectionCoefficient = 0.5f * (rs * rs + rp * rp);
    f.transmissionCoefficient = 1.0f - f.reflectionCoefficient;
    return f;
}

/* spec (ref stub: src/interactions.h:97-104; material fields README.md:167-186, src/scene.cpp:236-258).
 * Emission and absorption are handled by the caller; this picks the scattering lobe. */
int o_calculateBSDF(o_ray *r, o_vec3 intersect, o_vec3 normal, o_vec3 *color, const o_material *m,
                    float u_select, float xi1, float xi2, int trig_mode)
{
    o_vec3 d = r->direction;
    float ndotd = dot3(normal, d);
    int backside = (ndotd > 0.0f);
    o_vec3 nf = backside ? neg3(normal) : normal;          /* shading normal faces the incident ray */

    if (m->hasRefractive > 0.0f) {
        float n1 = backside ? m->indexOfRefraction : 1.0f;
        float n2 = backside ? 1.0f : m->indexOfRefraction;
        o_vec3 refl = o_calculateReflectionDirection(nf, d);
        o_vec3 trans = o_calculateTransmissionDirection(nf, d, n1, n2);
        o_Fresnel f = o_calculateFresnel(nf, d, n1, n2, refl, trans);
        *color = mul3(*color, m->specularColor);
        if (u_select < f.reflectionCoefficient) {
            r->origin = add3(intersect, scale3(O_RAY_BIAS_AMOUNT, nf));
            r->direction = refl;
            return 1;
        }
        r->origin = add3(intersect, scale3(O_RAY_BIAS_AMOUNT, neg3(nf)));
        r->direction = trans;
        return 2;
    }
    if (m->hasReflective > 0.0f) {
        *color = mul3(*color, m->specularColor);
        r->origin = add3(intersect, scale3(O_RAY_BIAS_AMOUNT, nf));
        r->direction = o_calculateReflectionDirection(nf, d);
        return 1;
    }
    *color = mul3(*color, m->color);
    r->origin = add3(intersect, scale3(O_RAY_BIAS_AMOUNT, nf));
    r->direction = o_calculateRandomDirectionInHemisphere(nf, xi1, xi2, trig_mode);
    return 0;
}

/* ------------------------------------------------------------------ */
/* raytraceKernel.cu                                                   */
/* ------------------------------------------------------------------ */
/* ref: src/raytraceKernel.cu:29-36 (index and the seed product are float arithmetic) */
o_vec3 o_generateRandomNumberFromThread(o_vec2 resolution, float time, int x, int y)
{
    int index = (int)((float)x + ((float)y * resolution.x));
    unsigned rng; o_minstd_seed(&rng, o_hash((unsigned)((float)index * time)));
    float a = o_uniform_real(&rng, 0, 1), b = o_uniform_real(&rng, 0, 1), c = o_uniform_real(&rng, 0, 1);
    return v3(a, b, c);
}

/* camera basis: the part of the camera ray that does not depend on the pixel (host-side in the product) */
typedef struct { o_vec3 eye, M, H, V; float resx, resy; o_vec3 A, B, vn; } cam_basis;
static cam_basis camera_basis(o_vec2 resolution, o_vec3 eye, o_vec3 view, o_vec3 up, o_vec2 fov)
{
    cam_basis b;
    o_vec3 A = normalize3(cross3(view, up));              /* screen-right */
    o_vec3 B = normalize3(cross3(A, view));               /* screen-up */
    float lenV = length3(view);
    float tx = (float)tan((double)fov.x * (O_PI / 180.0));
    float ty = (float)tan((double)fov.y * (O_PI / 180.0));
    b.eye = eye;
    b.M = add3(eye, view);
    b.H = scale3(lenV * tx, A);
    b.V = scale3(lenV * ty, B);
    b.resx = resolution.x; b.resy = resolution.y;
    b.A = A; b.B = B; b.vn = normalize3(view);            /* unit axes of the lens plane and its normal */
    return b;
}

/* spec (depth of field): thin lens.  The pinhole ray fixes the point in focus on the plane
 * focal_distance along the view axis; the ray starts on a disc of lens_radius around the eye in the (A, B) plane
 * -- r = R sqrt(u1), angle 2 pi u2 with the deterministic sincos -- and aims at that point. */
static o_ray lens_ray(const cam_basis *b, o_ray pinhole, float lens_radius, float focal_distance, float u1, float u2)
{
    float tf = focal_distance / dot3(pinhole.direction, b->vn);
    o_vec3 Pf = add3(b->eye, scale3(tf, pinhole.direction));
    float rr = lens_radius * sqrtf(u1);
    float around = (float)((double)u2 * O_TWO_PI);
    float sn, cs;
    o_sincos_poly(around, &sn, &cs);
    o_ray r;
    r.origin = add3(b->eye, add3(scale3(rr * cs, b->A), scale3(rr * sn, b->B)));
    r.direction = normalize3(sub3(Pf, r.origin));
    return r;
}
static o_ray camera_ray(const cam_basis *b, int x, int y, float jx, float jy)
{
    float sx = ((float)x + jx) / b->resx;
    float sy = ((float)y + jy) / b->resy;
    /* buffer x=0 is screen-right and y=0 is the top row, so that the reference harness's flips
     * (ref: src/main.cpp:120-125, 245-251) produce an unmirrored picture */
    o_vec3 P = add3(add3(b->M, scale3(1.0f - 2.0f * sx, b->H)), scale3(1.0f - 2.0f * sy, b->V));
    o_ray r;
    r.origin = b->eye;
    r.direction = normalize3(sub3(P, b->eye));
    return r;
}

/* spec (ref stub: src/raytraceKernel.cu:38-45; fov convention src/scene.cpp:204-207): jittered pinhole
 * camera.  `time` is the 1-based iteration the reference passes as (float)iterations (:149). */
o_ray o_raycastFromCameraKernel(o_vec2 resolution, float time, int x, int y, o_vec3 eye, o_vec3 view,
                                o_vec3 up, o_vec2 fov, unsigned seed)
{
    cam_basis b = camera_basis(resolution, eye, view, up, fov);
    unsigned pixel = (unsigned)x + (unsigned)y * (unsigned)(int)resolution.x;
    unsigned rng; o_minstd_seed(&rng, o_stream_seed(pixel, (unsigned)time, 0u, seed));
    float jx = o_u01(&rng), jy = o_u01(&rng);
    return camera_ray(&b, x, y, jx, jy);
}

/* ref: src/raytraceKernel.cu:58-89 (x255.0 in double, clamp above only, float->uchar truncation, w=0) */
void o_sendImageToPBO(unsigned char *pbo, int npixels, const float *image)
{
    for (int i = 0; i < npixels; i++) {
        float c[3];
        for (int k = 0; k < 3; k++) {
            c[k] = (float)((double)image[3 * i + k] * 255.0);
            if (c[k] > 255) c[k] = 255;
        }
        pbo[4 * i + 3] = 0;
        pbo[4 * i + 0] = (unsigned char)c[0];
        pbo[4 * i + 1] = (unsigned char)c[1];
        pbo[4 * i + 2] = (unsigned char)c[2];
    }
}

/* ref: src/raytraceKernel.cu:48-55 */
void o_clearImage(float *image, int npixels) { memset(image, 0, (size_t)npixels * 3 * sizeof(float)); }

/* spec: nearest hit over the whole primitive list; smallest t > 0, ties -> lowest index */
/* spec (the reference names the triangle test as optional, src/intersections.h:79, and parses *.obj objects as MESH
 * without ever loading them, src/scene.cpp:57-66): Moeller-Trumbore in WORLD space on the pre-transformed triangle
 * (v0, e1 = v1 - v0, e2 = v2 - v0), two-sided, with the ray direction normalised first as the sphere test does;
 * hit point = getPointOnRay on the normalised ray (the 1e-4 pull-back is in world units here), normal = the triangle's
 * geometric unit normal n (not flipped), result = distance from the ray origin to that point. */
float o_triangleIntersectionTest(o_vec3 v0, o_vec3 e1, o_vec3 e2, o_vec3 n, o_ray r, o_vec3 *intersectionPoint, o_vec3 *normal)
{
    o_ray rt;
    rt.origin = r.origin;
    rt.direction = normalize3(r.direction);
    o_vec3 pv = cross3(rt.direction, e2);
    float det = dot3(e1, pv);
    if (fabsf(det) < 1e-12f) return -1;
    float inv = 1.0f / det;
    o_vec3 tv = sub3(rt.origin, v0);
    float u = dot3(tv, pv) * inv;
    if (u < 0.0f || u > 1.0f) return -1;
    o_vec3 qv = cross3(tv, e1);
    float v = dot3(rt.direction, qv) * inv;
    if (v < 0.0f || u + v > 1.0f) return -1;
    float t = dot3(e2, qv) * inv;
    if (!(t > 0.0f)) return -1;
    o_vec3 real = o_getPointOnRay(rt, t);
    *intersectionPoint = real;
    *normal = n;
    return length3(sub3(r.origin, real));
}

void o_triangleToWorld(const o_staticGeom *g, const float v[9], float out12[12])
{
    o_vec3 w0 = o_multiplyMV(g->transform, v4(v3(v[0], v[1], v[2]), 1.0f));
    o_vec3 w1 = o_multiplyMV(g->transform, v4(v3(v[3], v[4], v[5]), 1.0f));
    o_vec3 w2 = o_multiplyMV(g->transform, v4(v3(v[6], v[7], v[8]), 1.0f));
    o_vec3 e1 = sub3(w1, w0), e2 = sub3(w2, w0);
    o_vec3 c = cross3(e1, e2);
    float l2 = dot3(c, c);
    o_vec3 n = (l2 > 0.0f) ? normalize3(c) : v3(0, 0, 0);
    out12[0] = w0.x; out12[1] = w0.y; out12[2] = w0.z;
    out12[3] = e1.x; out12[4] = e1.y; out12[5] = e1.z;
    out12[6] = e2.x; out12[7] = e2.y; out12[8] = e2.z;
    out12[9] = n.x; out12[10] = n.y; out12[11] = n.z;
}

/* all triangles of a scene state in world space; primitive index of triangle k = nG + k (after every geom) */
typedef struct { int n; float *w; int *geom; } tri_table;

static int build_tri_table(const o_staticGeom *geoms, int nG, const o_extras *ex, tri_table *tt)
{
    tt->n = 0; tt->w = NULL; tt->geom = NULL;
    if (!ex || ex->n_meshes <= 0) return 0;
    long long total = 0;
    for (int k = 0; k < ex->n_meshes; k++) {
        const o_mesh *m = &ex->meshes[k];
        if (m->geom < 0 || m->geom >= nG || geoms[m->geom].type != O_MESH || m->n_triangles < 0 || (m->n_triangles > 0 && !m->vertices)) return -6;
        total += m->n_triangles;
    }
    if (total == 0) return 0;
    tt->w = (float *)malloc((size_t)total * 12 * sizeof(float));
    tt->geom = (int *)malloc((size_t)total * sizeof(int));
    if (!tt->w || !tt->geom) return -5;
    for (int k = 0; k < ex->n_meshes; k++) {
        const o_mesh *m = &ex->meshes[k];
        for (int t = 0; t < m->n_triangles; t++) {
            o_triangleToWorld(&geoms[m->geom], m->vertices + 9 * (size_t)t, tt->w + 12 * (size_t)tt->n);
            tt->geom[tt->n] = m->geom;
            tt->n++;
        }
    }
    return 0;
}
static void free_tri_table(tri_table *tt) { free(tt->w); free(tt->geom); tt->w = NULL; tt->geom = NULL; tt->n = 0; }

/* nearest hit over the geoms, then the triangles; returns the primitive index (>= nG: triangle index - nG) */
static int nearest_hit(const o_staticGeom *geoms, int nG, const tri_table *tt, o_ray r, o_vec3 *p, o_vec3 *n, float *t_out)
{
    int best = -1; float best_t = 0;
    for (int i = 0; i < nG; i++) {
        o_vec3 ip, in; float t;
        if (geoms[i].type == O_SPHERE) t = o_sphereIntersectionTest(&geoms[i], r, &ip, &in);
        else if (geoms[i].type == O_CUBE) t = o_boxIntersectionTest(&geoms[i], r, &ip, &in);
        else t = -1;                                      /* MESH: its triangles follow the geoms */
        if (t > 0 && (best < 0 || t < best_t)) { best = i; best_t = t; *p = ip; *n = in; }
    }
    if (tt)
        for (int k = 0; k < tt->n; k++) {
            const float *w = tt->w + 12 * (size_t)k;
            o_vec3 ip, in;
            float t = o_triangleIntersectionTest(v3(w[0], w[1], w[2]), v3(w[3], w[4], w[5]), v3(w[6], w[7], w[8]),
                                                 v3(w[9], w[10], w[11]), r, &ip, &in);
            if (t > 0 && (best < 0 || t < best_t)) { best = nG + k; best_t = t; *p = ip; *n = in; }
        }
    if (t_out) *t_out = best_t;
    return best;
}

/* spec: one path.  Returns the radiance sample L_i of this iteration. */
/* Light table.  Entry j is an emissive sphere / cube (tri_count 0, prim = its index) or an emissive MESH geom as a whole
 * (prim = the geom; its triangles of positive area, in primitive order, are tris[tri_first .. tri_first + tri_count) with
 * the running fp32 sum of their areas in cdf[]; area = that sum's last value).  sampled[i] = primitive i (geoms, then
 * triangles) belongs to an entry: only those are left to the explicit sampling when hit by chance. */
typedef struct {
    int n; int prim[O_MAX_LIGHTS]; float area[O_MAX_LIGHTS]; int tri_first[O_MAX_LIGHTS], tri_count[O_MAX_LIGHTS];
    int *tris; float *cdf; unsigned char *sampled;
} light_table;

/* spec: area of a world-space triangle (e1, e2 = its edges from v0), fp32 */
float o_triangleArea(o_vec3 e1, o_vec3 e2) { return 0.5f * length3(cross3(e1, e2)); }

/* spec: uniform point on the triangle v0, v0 + e1, v0 + e2 from two uniform numbers (square-root parametrisation) */
o_vec3 o_sampleTriangle(o_vec3 v0, o_vec3 e1, o_vec3 e2, float u_a, float u_b)
{
    float s = sqrtf(u_a);
    float a = s * (1.0f - u_b), b = s * u_b;
    return add3(add3(v0, scale3(a, e1)), scale3(b, e2));
}

/* spec (DESIGN.md "Direct lighting"): emissive geoms in list order, at most O_MAX_LIGHTS entries; a MESH geom is one
 * entry made of its triangles (area-weighted pick, then a uniform point: the reference names the triangle as an optional
 * primitive, src/intersections.h:79, and its samplers take (geom, float seed), :133,179) */
static int collect_lights(const o_staticGeom *geoms, int nG, const o_material *mats, const tri_table *tt, light_table *lt)
{
    lt->n = 0; lt->tris = NULL; lt->cdf = NULL; lt->sampled = NULL;
    const int ntri = tt ? tt->n : 0;
    lt->sampled = (unsigned char *)calloc((size_t)(nG + ntri) + 1, 1);
    if (ntri > 0) {
        lt->tris = (int *)malloc((size_t)ntri * sizeof(int));
        lt->cdf = (float *)malloc((size_t)ntri * sizeof(float));
    }
    if (!lt->sampled || (ntri > 0 && (!lt->tris || !lt->cdf))) return -5;
    int used = 0;
    for (int i = 0; i < nG && lt->n < O_MAX_LIGHTS; i++) {
        if (!(mats[geoms[i].materialid].emittance > 0.0f)) continue;
        if (geoms[i].type != O_MESH) {
            lt->prim[lt->n] = i;
            lt->area[lt->n] = o_lightArea(&geoms[i]);
            lt->tri_first[lt->n] = 0; lt->tri_count[lt->n] = 0;
            lt->sampled[i] = 1;
            lt->n++;
            continue;
        }
        float acc = 0.0f;
        const int first = used;
        for (int k = 0; k < ntri; k++) {
            if (tt->geom[k] != i) continue;
            const float *w = tt->w + 12 * (size_t)k;
            const float a = o_triangleArea(v3(w[3], w[4], w[5]), v3(w[6], w[7], w[8]));
            if (!(a > 0.0f)) continue;
            acc = acc + a;
            lt->tris[used] = k; lt->cdf[used] = acc; used++;
            lt->sampled[nG + k] = 1;
        }
        if (used == first) continue;                              /* no surface: not a light */
        lt->prim[lt->n] = i;
        lt->area[lt->n] = acc;
        lt->tri_first[lt->n] = first; lt->tri_count[lt->n] = used - first;
        lt->n++;
    }
    return 0;
}
static void free_lights(light_table *lt) { free(lt->tris); free(lt->cdf); free(lt->sampled); lt->tris = NULL; lt->cdf = NULL; lt->sampled = NULL; }

/* per-ray motion blur: the knots of the shutter interval and scratch room for one path's interpolated scene */
typedef struct {
    int n_knots; const o_staticGeom *knot_geoms; const o_cameraData *knot_cams; o_vec2 resolution, fov;
    o_staticGeom *scratch;        /* nG entries, private to the calling thread */
} motion_knots;

static float lerp1f(float a, float b, float f) { return a + (b - a) * f; }
static o_vec3 lerp3f(o_vec3 a, o_vec3 b, float f) { return v3(lerp1f(a.x, b.x, f), lerp1f(a.y, b.y, f), lerp1f(a.z, b.z, f)); }
static o_vec4 lerp4f(o_vec4 a, o_vec4 b, float f) { o_vec4 r; r.x = lerp1f(a.x, b.x, f); r.y = lerp1f(a.y, b.y, f); r.z = lerp1f(a.z, b.z, f); r.w = lerp1f(a.w, b.w, f); return r; }

/* spec (per-ray motion blur): inverse of an affine transform given by rows 0..2 (a | a3), (b | b3), (c | c3): adjugate over
 * determinant for the 3x3 part, then -inverse * translation; fp32, this operation order (the HIP kernels run the same) */
void o_affineInverse(const o_mat4 *m, o_mat4 *out)
{
    const float a0 = m->x.x, a1 = m->x.y, a2 = m->x.z, a3 = m->x.w;
    const float b0 = m->y.x, b1 = m->y.y, b2 = m->y.z, b3 = m->y.w;
    const float c0 = m->z.x, c1 = m->z.y, c2 = m->z.z, c3 = m->z.w;
    const float k00 = b1 * c2 - b2 * c1, k01 = b2 * c0 - b0 * c2, k02 = b0 * c1 - b1 * c0;
    const float det = (a0 * k00 + a1 * k01) + a2 * k02;
    const float id = 1.0f / det;
    out->x.x = k00 * id; out->x.y = (a2 * c1 - a1 * c2) * id; out->x.z = (a1 * b2 - a2 * b1) * id;
    out->y.x = k01 * id; out->y.y = (a0 * c2 - a2 * c0) * id; out->y.z = (a2 * b0 - a0 * b2) * id;
    out->z.x = k02 * id; out->z.y = (a1 * c0 - a0 * c1) * id; out->z.z = (a0 * b1 - a1 * b0) * id;
    out->x.w = -((out->x.x * a3 + out->x.y * b3) + out->x.z * c3);
    out->y.w = -((out->y.x * a3 + out->y.y * b3) + out->y.z * c3);
    out->z.w = -((out->z.x * a3 + out->z.y * b3) + out->z.z * c3);
    out->w.x = 0.0f; out->w.y = 0.0f; out->w.z = 0.0f; out->w.w = 1.0f;
}

/* spec (per-ray motion blur): the scene a path with shutter draw u_t sees -- segment k = min(floor(u_t * K), K - 1) of the
 * K = n_knots - 1 between the knots, f = u_t * K - k, rows 0..2 of the TRANSFORM (and the camera vectors) a + (b - a) * f,
 * the inverse transform computed from that (o_affineInverse): the object a ray meets is exactly the interpolated transform's
 * image of the unit shape -- a point of it is the convex combination of its places at the two knots */
static void scene_at_time(const motion_knots *mk, int nG, float u_t, o_staticGeom *out, cam_basis *cb, const o_cameraData *cam0)
{
    const int K = mk->n_knots - 1;
    const float tau = u_t * (float)K;
    int k = (int)tau;
    if (k > K - 1) k = K - 1;
    const float f = tau - (float)k;
    const o_staticGeom *A = mk->knot_geoms + (size_t)k * (size_t)nG, *B = A + nG;
    for (int g = 0; g < nG; g++) {
        out[g] = A[g];
        out[g].transform.x = lerp4f(A[g].transform.x, B[g].transform.x, f);
        out[g].transform.y = lerp4f(A[g].transform.y, B[g].transform.y, f);
        out[g].transform.z = lerp4f(A[g].transform.z, B[g].transform.z, f);
        o_affineInverse(&out[g].transform, &out[g].inverseTransform);
    }
    if (mk->knot_cams) {
        const o_cameraData *ca = &mk->knot_cams[k], *cn = ca + 1;
        *cb = camera_basis(mk->resolution, lerp3f(ca->position, cn->position, f), lerp3f(ca->view, cn->view, f),
                           lerp3f(ca->up, cn->up, f), mk->fov);
    } else {
        *cb = camera_basis(mk->resolution, cam0->position, cam0->view, cam0->up, mk->fov);
    }
}

static o_vec3 trace_path(const o_staticGeom *geoms, int nG, const tri_table *tt, const o_material *mats, const cam_basis *cb,
                         const o_options *opt, const light_table *lt, int W, int x, int y, unsigned iteration,
                         int *bounces, unsigned long long *live_in, unsigned long long *shadow_rays,
                         const motion_knots *mk, const o_cameraData *cam0)
{
    unsigned pixel = (unsigned)x + (unsigned)y * (unsigned)W;
    unsigned rng; o_minstd_seed(&rng, o_stream_seed(pixel, iteration, 0u, opt->seed));
    float jx = o_u01(&rng), jy = o_u01(&rng);
    cam_basis cb_t;
    if (mk) {
        /* the path's shutter time: third draw of the camera stream; its whole scene is interpolated once */
        float u_t = o_u01(&rng);
        scene_at_time(mk, nG, u_t, mk->scratch, &cb_t, cam0);
        geoms = mk->scratch;
        cb = &cb_t;
    }
    o_ray r = camera_ray(cb, x, y, jx, jy);
    if (opt->lens_radius > 0.0f) {
        float u1 = o_u01(&rng), u2 = o_u01(&rng);
        r = lens_ray(cb, r, opt->lens_radius, opt->focal_distance, u1, u2);
    }
    o_vec3 T = v3(1, 1, 1), L = v3(0, 0, 0);
    const int nee = opt->direct_light && lt && lt->n > 0;          /* explicit light sampling at diffuse vertices */
    int suppress = 0;     /* the previous vertex sampled the lights explicitly: hitting one by chance adds nothing */

    for (int b = 0; b < opt->depth; b++) {
        if (bounces) (*bounces)++;
        if (live_in) live_in[b]++;
        o_vec3 p, n;
        float t_hit;
        int hit = nearest_hit(geoms, nG, tt, r, &p, &n, &t_hit);
        if (hit < 0) break;                                        /* background is black */
        const o_staticGeom *hg = (hit < nG) ? &geoms[hit] : &geoms[tt->geom[hit - nG]];   /* triangle: its MESH geom */
        const o_material *m = &mats[hg->materialid];
        if (m->emittance > 0.0f) {                                 /* light: emit and stop */
            /* (an emitter the explicit sampling does not cover -- beyond the table's O_MAX_LIGHTS entries -- still counts) */
            if (!(suppress && lt->sampled[hit])) L = add3(L, scale3(m->emittance, mul3(T, m->color)));
            break;
        }
        if (b == opt->depth - 1 && !nee) break;                    /* depth exhausted: no contribution */

        o_minstd_seed(&rng, o_stream_seed(pixel, iteration, (unsigned)b + 1u, opt->seed));
        float u_select = o_u01(&rng), xi1 = o_u01(&rng), xi2 = o_u01(&rng), u_rr = o_u01(&rng);

        /* spec (optional, opt->scatter): a SCATTER material that is not a mirror encloses a participating
         * medium; its surface is a dielectric when REFR is set and index-matched (rays pass straight through) otherwise */
        const int medium = opt->scatter && m->hasScatter > 0.0f && !(m->hasReflective > 0.0f);
        const int diffuse = !(m->hasRefractive > 0.0f) && !(m->hasReflective > 0.0f) && !medium;
        if (nee && diffuse) {
            /* spec (ref samplers src/intersections.h:133-182): one light by u_light, one point on
             * it by the reference's float-seeded sampler, one shadow ray; estimator
             * T*c/pi * Le * cos_x cos_y / d^2 * (area * number of lights) */
            float u_light = o_u01(&rng), u_seed = o_u01(&rng);
            int j = (int)(u_light * (float)lt->n);
            if (j > lt->n - 1) j = lt->n - 1;
            const o_staticGeom *lg = &geoms[lt->prim[j]];
            o_vec3 yl, nl;
            int lprim = lt->prim[j];                               /* the primitive the shadow ray must reach */
            o_vec3 nf = (dot3(n, r.direction) > 0.0f) ? neg3(n) : n;
            o_ray sr;
            sr.origin = add3(p, scale3(O_RAY_BIAS_AMOUNT, nf));
            if (lt->tri_count[j] > 0) {
                /* mesh light: the float-seeded engine of the reference's samplers draws the triangle (by area: first one
                 * whose running area exceeds u_t * total, else the last) and a uniform point on it; triangles emit from
                 * both sides, so the normal is taken on the side that faces the shading point */
                unsigned lr; o_minstd_seed(&lr, o_hash((unsigned)(u_seed * 16777216.0f)));
                float u_t = o_uniform_real(&lr, 0, 1), u_a = o_uniform_real(&lr, 0, 1), u_b = o_uniform_real(&lr, 0, 1);
                const float *cdf = lt->cdf + lt->tri_first[j];
                float target = u_t * lt->area[j];
                int lo = 0, hi = lt->tri_count[j] - 1;
                while (lo < hi) { int mid = (lo + hi) >> 1; if (cdf[mid] > target) hi = mid; else lo = mid + 1; }
                const int k = lt->tris[lt->tri_first[j] + lo];
                const float *w = tt->w + 12 * (size_t)k;
                yl = o_sampleTriangle(v3(w[0], w[1], w[2]), v3(w[3], w[4], w[5]), v3(w[6], w[7], w[8]), u_a, u_b);
                nl = v3(w[9], w[10], w[11]);
                if (dot3(nl, sub3(yl, sr.origin)) > 0.0f) nl = neg3(nl);
                lprim = nG + k;
            } else {
                o_sampleLight(lg, u_seed * 16777216.0f, &yl, &nl);
            }
            o_vec3 wi = sub3(yl, sr.origin);
            float d2 = dot3(wi, wi);
            float dist = sqrtf(d2);
            sr.direction = normalize3(wi);
            float cx = dot3(nf, sr.direction), cy = -dot3(nl, sr.direction);
            if (cx > 0.0f && cy > 0.0f) {
                if (shadow_rays) (*shadow_rays)++;
                o_vec3 hp, hn; float ht;
                int hs = nearest_hit(geoms, nG, tt, sr, &hp, &hn, &ht);
                float tol = 1e-3f * ((dist > 1.0f) ? dist : 1.0f);
                if (hs == lprim && fabsf(ht - dist) <= tol) {
                    const o_material *lm = &mats[lg->materialid];
                    float G = (cx * cy) / d2;
                    /* (per-ray motion blur: the light's area at the path's time) */
                    const float larea = (mk && lt->tri_count[j] == 0) ? o_lightArea(lg) : lt->area[j];
                    float wgt = (G * (larea * (float)lt->n)) * 0.318309886f;
                    o_vec3 c = mul3(mul3(T, m->color), scale3(lm->emittance, lm->color));
                    L = add3(L, scale3(wgt, c));
                }
            }
        }
        if (b == opt->depth - 1) break;                            /* depth exhausted */

        /* spec (optional): a segment that ends on the inner side of a refractive surface ran through
         * the medium: Beer-Lambert with the material's ABSCOEFF over the segment's world length */
        if (opt->absorption && !medium && m->hasRefractive > 0.0f && dot3(n, r.direction) > 0.0f &&
            (m->absorptionCoefficient.x != 0.0f || m->absorptionCoefficient.y != 0.0f || m->absorptionCoefficient.z != 0.0f))
            T = mul3(T, o_calculateTransmission(m->absorptionCoefficient, t_hit));

        const o_vec3 d_in = r.direction;
        int lobe;
        int scattered = 0, pass_through = 0;
        if (medium) {
            const int inside = dot3(n, d_in) > 0.0f;
            if (inside) {
                /* the segment ran through the medium: three more draws of the bounce's stream decide whether the path
                 * scatters before the boundary (isotropic, absorbed over the free flight) or reaches it (absorbed over
                 * the whole segment) */
                float u_sd = o_u01(&rng), u_s2 = o_u01(&rng), u_s3 = o_u01(&rng);
                o_AbsorptionAndScatteringProperties props;
                props.absorptionCoefficient = m->absorptionCoefficient;
                props.reducedScatteringCoefficient = m->reducedScatterCoefficient;
                float seg = t_hit;
                scattered = o_calculateScatterAndAbsorption(&r, &seg, &props, &T, m, u_sd, u_s2, u_s3);
            }
            if (!scattered && !(m->hasRefractive > 0.0f)) {
                /* index-matched boundary: the ray goes straight on; entering picks up the surface colour once */
                if (!inside) T = mul3(T, m->color);
                pass_through = 1;
            }
        }
        if (scattered) lobe = 3;
        else if (pass_through) lobe = 2;
        else lobe = o_calculateBSDF(&r, p, n, &T, m, u_select, xi1, xi2, opt->trig_mode);
        if (lobe == 2) {
            /* spec: a transmitted ray must start on the far side of the surface, but `p` was pulled back towards
             * the ray origin by getPointOnRay's 1e-4 object-space epsilon (ref: src/intersections.h:46-48), which
             * for objects scaled by more than 2 exceeds RAY_BIAS_AMOUNT: the ray would meet the same surface
             * again from the same side.  The offset therefore adds that pull-back's world length,
             * 1e-4 / |inverseTransform * d|. */
            /* (a triangle is tested in world space: its pull-back is 1e-4 / |d|) */
            o_vec3 v = (hit < nG) ? o_multiplyMV(geoms[hit].inverseTransform, v4(d_in, 0.0f)) : d_in;
            float pb = 1e-4f * (1.0f / sqrtf(dot3(v, v)));
            o_vec3 nf = (dot3(n, d_in) > 0.0f) ? neg3(n) : n;
            r.origin = add3(p, scale3(O_RAY_BIAS_AMOUNT + pb, neg3(nf)));
        }
        suppress = nee && diffuse;

        if (opt->rr_start >= 0 && b >= opt->rr_start) {            /* Russian roulette */
            float q = T.x;
            if (T.y > q) q = T.y;
            if (T.z > q) q = T.z;
            q = (q < 0.05f) ? 0.05f : ((q > 1.0f) ? 1.0f : q);
            if (u_rr >= q) break;
            T = v3(T.x / q, T.y / q, T.z / q);
        }
    }
    return L;
}

static int validate(const o_staticGeom *geoms, int nG, int nM, const o_cameraData *cam, const o_options *opt)
{
    if (!cam || !opt || nG < 0 || nM < 0 || (nG > 0 && !geoms)) return -1;
    if (opt->depth < 1 || (int)cam->resolution.x < 1 || (int)cam->resolution.y < 1) return -2;
    for (int i = 0; i < nG; i++)
        if (geoms[i].type != O_MESH && (geoms[i].materialid < 0 || geoms[i].materialid >= nM)) return -3;
    return 0;
}

/* one scene state (a motion-blur slice, or the static scene) */
typedef struct { const o_staticGeom *geoms; tri_table tt; light_table lt; cam_basis cb; } scene_state;

static int n_states(const o_extras *ex) { return (ex && ex->n_slices > 0 && ex->slice_geoms) ? ex->n_slices : 1; }

static int build_states(const o_staticGeom *geoms, int nG, const o_material *mats, const o_cameraData *cam,
                        const o_extras *ex, scene_state **out)
{
    const int n = n_states(ex);
    scene_state *st = (scene_state *)calloc((size_t)n, sizeof(*st));
    if (!st) return -5;
    for (int k = 0; k < n; k++) {
        const int sliced = (ex && ex->n_slices > 0 && ex->slice_geoms);
        st[k].geoms = sliced ? ex->slice_geoms + (size_t)k * (size_t)nG : geoms;
        const o_cameraData *c = (sliced && ex->slice_cams) ? &ex->slice_cams[k] : cam;
        st[k].cb = camera_basis(cam->resolution, c->position, c->view, c->up, cam->fov);
        st[k].lt.tris = NULL; st[k].lt.cdf = NULL; st[k].lt.sampled = NULL;
        int rc = build_tri_table(st[k].geoms, nG, ex, &st[k].tt);
        if (rc == 0) rc = collect_lights(st[k].geoms, nG, mats, &st[k].tt, &st[k].lt);
        if (rc != 0) { for (int j = 0; j <= k; j++) { free_tri_table(&st[j].tt); free_lights(&st[j].lt); } free(st); return rc; }
    }
    *out = st;
    return 0;
}
static void free_states(scene_state *st, int n) { for (int k = 0; k < n; k++) { free_tri_table(&st[k].tt); free_lights(&st[k].lt); } free(st); }
static const scene_state *state_of(const scene_state *st, int n, unsigned iteration)
{
    return &st[n > 1 ? (int)(((iteration - 1u) / (unsigned)O_SLICE_ITERATIONS) % (unsigned)n) : 0];
}

/* per-ray motion blur asked for?  0 = no, 1 = yes, < 0 = invalid combination */
static int motion_mode(const o_extras *ex, const o_options *opt, const o_staticGeom *geoms, int nG)
{
    if (!ex || ex->n_knots == 0) return 0;
    if (ex->n_knots < 2 || !ex->knot_geoms) return -7;
    if (ex->n_slices > 0 || ex->n_meshes > 0) return -7;
    for (int i = 0; i < nG; i++) if (geoms[i].type == O_MESH) return -7;
    return 1;
}

o_vec3 o_trace_path_ex(const o_staticGeom *geoms, int nG, const o_material *mats, int nM, const o_cameraData *cam,
                       const o_options *opt, const o_extras *ex, int x, int y, unsigned iteration, int *bounces_out)
{
    if (validate(geoms, nG, nM, cam, opt) != 0) return v3(-1, -1, -1);
    const int mm = motion_mode(ex, opt, geoms, nG);
    if (mm < 0) return v3(-1, -1, -1);
    scene_state *st;
    if (build_states(geoms, nG, mats, cam, ex, &st) != 0) return v3(-1, -1, -1);
    const int n = n_states(ex);
    const scene_state *s = state_of(st, n, iteration);
    if (bounces_out) *bounces_out = 0;
    motion_knots mk;
    if (mm) {
        mk.n_knots = ex->n_knots; mk.knot_geoms = ex->knot_geoms; mk.knot_cams = ex->knot_cams;
        mk.resolution = cam->resolution; mk.fov = cam->fov;
        mk.scratch = (o_staticGeom *)malloc((size_t)(nG > 0 ? nG : 1) * sizeof(o_staticGeom));
        if (!mk.scratch) { free_states(st, n); return v3(-1, -1, -1); }
    }
    o_vec3 L = trace_path(s->geoms, nG, &s->tt, mats, &s->cb, opt, &s->lt, (int)cam->resolution.x, x, y, iteration, bounces_out, NULL, NULL,
                          mm ? &mk : NULL, cam);
    if (mm) free(mk.scratch);
    free_states(st, n);
    return L;
}

o_vec3 o_trace_path(const o_staticGeom *geoms, int nG, const o_material *mats, int nM, const o_cameraData *cam,
                    const o_options *opt, int x, int y, unsigned iteration, int *bounces_out)
{
    return o_trace_path_ex(geoms, nG, mats, nM, cam, opt, NULL, x, y, iteration, bounces_out);
}

typedef struct {
    const scene_state *states; int n_states; int nG; const o_material *mats; const o_options *opt;
    float *image; int W, H; int iter_first, iter_count; int row0, row1;
    unsigned long long *live_in;   /* private per thread, depth entries */
    unsigned long long shadow_rays;
    const motion_knots *mk; const o_cameraData *cam0;   /* per-ray motion blur (mk->scratch private per thread), else NULL */
} job;

static void *render_rows(void *arg)
{
    job *j = (job *)arg;
    for (int it = j->iter_first; it < j->iter_first + j->iter_count; it++) {
        const scene_state *s = state_of(j->states, j->n_states, (unsigned)it);
        for (int y = j->row0; y < j->row1; y++) {
            for (int x = 0; x < j->W; x++) {
                o_vec3 L = trace_path(s->geoms, j->nG, &s->tt, j->mats, &s->cb, j->opt, &s->lt, j->W, x, y, (unsigned)it,
                                      NULL, j->live_in, &j->shadow_rays, j->mk, j->cam0);
                /* spec: running mean, stateless given (image, iteration) */
                float *px = &j->image[3 * ((size_t)x + (size_t)y * (size_t)j->W)];
                float fi = (float)it, fim1 = (float)(it - 1);
                px[0] = (px[0] * fim1 + L.x) / fi;
                px[1] = (px[1] * fim1 + L.y) / fi;
                px[2] = (px[2] * fim1 + L.z) / fi;
            }
        }
    }
    return NULL;
}

int o_render(const o_staticGeom *geoms, int nG, const o_material *mats, int nM, const o_cameraData *cam,
             const o_options *opt, float *image, int iter_first, int iter_count,
             unsigned long long *live_in, int nthreads)
{
    return o_render_counted(geoms, nG, mats, nM, cam, opt, image, iter_first, iter_count, live_in, NULL, nthreads);
}

int o_render_counted(const o_staticGeom *geoms, int nG, const o_material *mats, int nM, const o_cameraData *cam,
                     const o_options *opt, float *image, int iter_first, int iter_count,
                     unsigned long long *live_in, unsigned long long *shadow_rays, int nthreads)
{
    return o_render_ex(geoms, nG, mats, nM, cam, opt, NULL, image, iter_first, iter_count, live_in, shadow_rays, nthreads);
}

int o_render_ex(const o_staticGeom *geoms, int nG, const o_material *mats, int nM, const o_cameraData *cam,
                const o_options *opt, const o_extras *ex, float *image, int iter_first, int iter_count,
                unsigned long long *live_in, unsigned long long *shadow_rays, int nthreads)
{
    int rc = validate(geoms, nG, nM, cam, opt);
    if (rc != 0) return rc;
    if (!image || iter_first < 1 || iter_count < 0) return -4;
    const int mm = motion_mode(ex, opt, geoms, nG);
    if (mm < 0) return mm;
    int W = (int)cam->resolution.x, H = (int)cam->resolution.y;
    scene_state *states;
    rc = build_states(geoms, nG, mats, cam, ex, &states);
    if (rc != 0) return rc;
    const int ns = n_states(ex);
    if (nthreads < 1) nthreads = 1;
    if (nthreads > H) nthreads = H;
    if (nthreads > 256) nthreads = 256;

    job jobs[256]; pthread_t th[256];
    motion_knots *mkt = NULL;
    o_staticGeom *scratch = NULL;
    if (mm) {
        mkt = (motion_knots *)malloc((size_t)nthreads * sizeof(motion_knots));
        scratch = (o_staticGeom *)malloc((size_t)nthreads * (size_t)(nG > 0 ? nG : 1) * sizeof(o_staticGeom));
        if (!mkt || !scratch) { free(mkt); free(scratch); free_states(states, ns); return -5; }
    }
    unsigned long long *counts = (unsigned long long *)calloc((size_t)nthreads * (size_t)opt->depth, sizeof(*counts));
    if (!counts) { free(mkt); free(scratch); free_states(states, ns); return -5; }
    /* rows are dealt in contiguous blocks; each pixel is owned by exactly one thread */
    for (int t = 0; t < nthreads; t++) {
        job *j = &jobs[t];
        j->states = states; j->n_states = ns; j->nG = nG; j->mats = mats; j->opt = opt; j->image = image;
        j->W = W; j->H = H; j->iter_first = iter_first; j->iter_count = iter_count;
        j->row0 = (int)((long long)H * t / nthreads); j->row1 = (int)((long long)H * (t + 1) / nthreads);
        j->live_in = counts + (size_t)t * (size_t)opt->depth;
        j->shadow_rays = 0;
        j->mk = NULL; j->cam0 = cam;
        if (mm) {
            mkt[t].n_knots = ex->n_knots; mkt[t].knot_geoms = ex->knot_geoms; mkt[t].knot_cams = ex->knot_cams;
            mkt[t].resolution = cam->resolution; mkt[t].fov = cam->fov;
            mkt[t].scratch = scratch + (size_t)t * (size_t)(nG > 0 ? nG : 1);
            j->mk = &mkt[t];
        }
    }
    if (nthreads == 1) render_rows(&jobs[0]);
    else {
        for (int t = 0; t < nthreads; t++) pthread_create(&th[t], NULL, render_rows, &jobs[t]);
        for (int t = 0; t < nthreads; t++) pthread_join(th[t], NULL);
    }
    if (live_in)
        for (int b = 0; b < opt->depth; b++) {
            unsigned long long s = 0;
            for (int t = 0; t < nthreads; t++) s += counts[(size_t)t * (size_t)opt->depth + (size_t)b];
            live_in[b] += s;
        }
    if (shadow_rays)
        for (int t = 0; t < nthreads; t++) *shadow_rays += jobs[t].shadow_rays;
    free(counts);
    free(mkt);
    free(scratch);
    free_states(states, ns);
    return 0;
}
