/*
 * pt_oracle_scene.c -- CPU oracle, scene-file loader and transform builder.
 * TEST INFRASTRUCTURE ONLY (see pt_oracle.h).
 *
 * Restates ref: src/scene.cpp (grammar), src/utilities.cpp:74-90 (TRS build,
 * row layout) and the GLM 0.9.5.4 routines they call:
 *   external/include/glm/gtc/matrix_transform.inl:35-47 (translate), :49-88
 *   (rotate), :128-141 (scale); detail/type_mat4x4.inl:753-775 (mat4*mat4),
 *   :476-535 (inverse); detail/func_matrix.inl:344-369 (transpose).
 * Pinned by tests/golden/glm_vectors.json (real GLM, oracle/ref_glm_probe.cpp)
 * and the loader dump in tests/golden/reference_vectors.json.
 */
#include "pt_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define O_PI 3.1415926535897932384626422832795028841971

/* column-major like glm::mat4: m.c[col][row] */
typedef struct { float c[4][4]; } gmat4;

static gmat4 g_identity(void)
{
    gmat4 m; memset(&m, 0, sizeof m);
    m.c[0][0] = m.c[1][1] = m.c[2][2] = m.c[3][3] = 1.0f;
    return m;
}

/* Result[3] = m[0]*v[0] + m[1]*v[1] + m[2]*v[2] + m[3] */
static gmat4 g_translate(gmat4 m, o_vec3 v)
{
    gmat4 r = m;
    for (int k = 0; k < 4; k++)
        r.c[3][k] = m.c[0][k] * v.x + m.c[1][k] * v.y + m.c[2][k] * v.z + m.c[3][k];
    return r;
}

static gmat4 g_rotate(gmat4 m, float angle, o_vec3 v)
{
    float a = angle;                       /* GLM_FORCE_RADIANS, ref: src/utilities.cpp:7 */
    float c = cosf(a), s = sinf(a);
    float sqr = v.x * v.x + v.y * v.y + v.z * v.z;
    float inv = 1.0f / sqrtf(sqr);
    float axis[3] = { v.x * inv, v.y * inv, v.z * inv };
    float temp[3] = { (1.0f - c) * axis[0], (1.0f - c) * axis[1], (1.0f - c) * axis[2] };
    float R[3][3];
    R[0][0] = c + temp[0] * axis[0];
    R[0][1] = 0 + temp[0] * axis[1] + s * axis[2];
    R[0][2] = 0 + temp[0] * axis[2] - s * axis[1];
    R[1][0] = 0 + temp[1] * axis[0] - s * axis[2];
    R[1][1] = c + temp[1] * axis[1];
    R[1][2] = 0 + temp[1] * axis[2] + s * axis[0];
    R[2][0] = 0 + temp[2] * axis[0] + s * axis[1];
    R[2][1] = 0 + temp[2] * axis[1] - s * axis[0];
    R[2][2] = c + temp[2] * axis[2];
    gmat4 r;
    for (int j = 0; j < 3; j++)
        for (int k = 0; k < 4; k++)
            r.c[j][k] = m.c[0][k] * R[j][0] + m.c[1][k] * R[j][1] + m.c[2][k] * R[j][2];
    for (int k = 0; k < 4; k++) r.c[3][k] = m.c[3][k];
    return r;
}

static gmat4 g_scale(gmat4 m, o_vec3 v)
{
    gmat4 r;
    for (int k = 0; k < 4; k++) {
        r.c[0][k] = m.c[0][k] * v.x;
        r.c[1][k] = m.c[1][k] * v.y;
        r.c[2][k] = m.c[2][k] * v.z;
        r.c[3][k] = m.c[3][k];
    }
    return r;
}

/* Result[j] = A[0]*B[j][0] + A[1]*B[j][1] + A[2]*B[j][2] + A[3]*B[j][3] */
static gmat4 g_mul(gmat4 A, gmat4 B)
{
    gmat4 r;
    for (int j = 0; j < 4; j++)
        for (int k = 0; k < 4; k++)
            r.c[j][k] = A.c[0][k] * B.c[j][0] + A.c[1][k] * B.c[j][1] + A.c[2][k] * B.c[j][2] + A.c[3][k] * B.c[j][3];
    return r;
}

static gmat4 g_inverse(gmat4 M)
{
#define m(i, j) M.c[i][j]
    float Coef00 = m(2,2) * m(3,3) - m(3,2) * m(2,3);
    float Coef02 = m(1,2) * m(3,3) - m(3,2) * m(1,3);
    float Coef03 = m(1,2) * m(2,3) - m(2,2) * m(1,3);
    float Coef04 = m(2,1) * m(3,3) - m(3,1) * m(2,3);
    float Coef06 = m(1,1) * m(3,3) - m(3,1) * m(1,3);
    float Coef07 = m(1,1) * m(2,3) - m(2,1) * m(1,3);
    float Coef08 = m(2,1) * m(3,2) - m(3,1) * m(2,2);
    float Coef10 = m(1,1) * m(3,2) - m(3,1) * m(1,2);
    float Coef11 = m(1,1) * m(2,2) - m(2,1) * m(1,2);
    float Coef12 = m(2,0) * m(3,3) - m(3,0) * m(2,3);
    float Coef14 = m(1,0) * m(3,3) - m(3,0) * m(1,3);
    float Coef15 = m(1,0) * m(2,3) - m(2,0) * m(1,3);
    float Coef16 = m(2,0) * m(3,2) - m(3,0) * m(2,2);
    float Coef18 = m(1,0) * m(3,2) - m(3,0) * m(1,2);
    float Coef19 = m(1,0) * m(2,2) - m(2,0) * m(1,2);
    float Coef20 = m(2,0) * m(3,1) - m(3,0) * m(2,1);
    float Coef22 = m(1,0) * m(3,1) - m(3,0) * m(1,1);
    float Coef23 = m(1,0) * m(2,1) - m(2,0) * m(1,1);
    float Fac0[4] = { Coef00, Coef00, Coef02, Coef03 };
    float Fac1[4] = { Coef04, Coef04, Coef06, Coef07 };
    float Fac2[4] = { Coef08, Coef08, Coef10, Coef11 };
    float Fac3[4] = { Coef12, Coef12, Coef14, Coef15 };
    float Fac4[4] = { Coef16, Coef16, Coef18, Coef19 };
    float Fac5[4] = { Coef20, Coef20, Coef22, Coef23 };
    float Vec0[4] = { m(1,0), m(0,0), m(0,0), m(0,0) };
    float Vec1[4] = { m(1,1), m(0,1), m(0,1), m(0,1) };
    float Vec2[4] = { m(1,2), m(0,2), m(0,2), m(0,2) };
    float Vec3[4] = { m(1,3), m(0,3), m(0,3), m(0,3) };
    const float SignA[4] = { +1, -1, +1, -1 }, SignB[4] = { -1, +1, -1, +1 };
    gmat4 Inv;
    for (int k = 0; k < 4; k++) {
        float Inv0 = Vec1[k] * Fac0[k] - Vec2[k] * Fac1[k] + Vec3[k] * Fac2[k];
        float Inv1 = Vec0[k] * Fac0[k] - Vec2[k] * Fac3[k] + Vec3[k] * Fac4[k];
        float Inv2 = Vec0[k] * Fac1[k] - Vec1[k] * Fac3[k] + Vec3[k] * Fac5[k];
        float Inv3 = Vec0[k] * Fac2[k] - Vec1[k] * Fac4[k] + Vec2[k] * Fac5[k];
        Inv.c[0][k] = Inv0 * SignA[k];
        Inv.c[1][k] = Inv1 * SignB[k];
        Inv.c[2][k] = Inv2 * SignA[k];
        Inv.c[3][k] = Inv3 * SignB[k];
    }
    float Dot0[4] = { m(0,0) * Inv.c[0][0], m(0,1) * Inv.c[1][0], m(0,2) * Inv.c[2][0], m(0,3) * Inv.c[3][0] };
    float Dot1 = (Dot0[0] + Dot0[1]) + (Dot0[2] + Dot0[3]);
    float OneOverDeterminant = 1.0f / Dot1;
    for (int j = 0; j < 4; j++)
        for (int k = 0; k < 4; k++) Inv.c[j][k] = Inv.c[j][k] * OneOverDeterminant;
    return Inv;
#undef m
}

/* ref: src/utilities.cpp:83-90 glmMat4ToCudaMat4: transpose, then m.x = a[0] ... => cudaMat4 rows */
static o_mat4 g_to_rows(gmat4 a)
{
    o_mat4 r;
    r.x.x = a.c[0][0]; r.x.y = a.c[1][0]; r.x.z = a.c[2][0]; r.x.w = a.c[3][0];
    r.y.x = a.c[0][1]; r.y.y = a.c[1][1]; r.y.z = a.c[2][1]; r.y.w = a.c[3][1];
    r.z.x = a.c[0][2]; r.z.y = a.c[1][2]; r.z.z = a.c[2][2]; r.z.w = a.c[3][2];
    r.w.x = a.c[0][3]; r.w.y = a.c[1][3]; r.w.z = a.c[2][3]; r.w.w = a.c[3][3];
    return r;
}

/* ref: src/utilities.cpp:74-81 + src/scene.cpp:125-127.  rotat_units=O_ROTAT_DEGREES converts the angles
 * with glm::radians' formula (degrees * pi/180 in fp32) before the same rotate calls. */
o_mat4 o_buildTransformationMatrix(o_vec3 translation, o_vec3 rotation, o_vec3 scale, int rotat_units,
                                   o_mat4 *inverse_out)
{
    if (rotat_units == O_ROTAT_DEGREES) {
        const float k = (float)(O_PI / 180.0);
        rotation.x = rotation.x * k; rotation.y = rotation.y * k; rotation.z = rotation.z * k;
    }
    o_vec3 ax = { 1, 0, 0 }, ay = { 0, 1, 0 }, az = { 0, 0, 1 };
    gmat4 translationMat = g_translate(g_identity(), translation);
    gmat4 rotationMat = g_rotate(g_identity(), rotation.x, ax);
    rotationMat = g_mul(rotationMat, g_rotate(g_identity(), rotation.y, ay));
    rotationMat = g_mul(rotationMat, g_rotate(g_identity(), rotation.z, az));
    gmat4 scaleMat = g_scale(g_identity(), scale);
    gmat4 t = g_mul(g_mul(translationMat, rotationMat), scaleMat);
    if (inverse_out) *inverse_out = g_to_rows(g_inverse(t));
    return g_to_rows(t);
}

/* ------------------------------------------------------------------ */
/* scene grammar, ref: src/scene.cpp                                   */
/* ------------------------------------------------------------------ */
/* ref: src/utilities.cpp:109-140 safeGetline: LF, CRLF and lone CR all end a line */
static int safe_getline(FILE *f, char *buf, size_t cap, int *eof)
{
    size_t n = 0;
    for (;;) {
        int c = fgetc(f);
        if (c == '\n') break;
        if (c == '\r') { int d = fgetc(f); if (d != '\n' && d != EOF) ungetc(d, f); break; }
        if (c == EOF) { if (n == 0) *eof = 1; break; }
        if (n + 1 < cap) buf[n++] = (char)c;
    }
    buf[n] = 0;
    return (int)n;
}

static int g_frame = 0;      /* which animation frame o_scene_load_frame keeps (ref: src/sceneStructs.h:21-30) */

#define MAXTOK 8
static int tokenize(char *line, char *tok[MAXTOK])
{
    int n = 0; char *p = line;
    while (*p && n < MAXTOK) {
        while (*p == ' ' || *p == '\t' || *p == '\v' || *p == '\f') p++;
        if (!*p) break;
        tok[n++] = p;
        while (*p && !(*p == ' ' || *p == '\t' || *p == '\v' || *p == '\f')) p++;
        if (*p) *p++ = 0;
    }
    return n;
}
static float tokf(char *tok[MAXTOK], int n, int i) { return (i < n) ? (float)atof(tok[i]) : 0.0f; }

typedef struct { o_staticGeom *v; int n, cap; } geom_vec;
typedef struct { o_material *v; int n, cap; } mat_vec;

static int load_material(FILE *f, const char *idtok, mat_vec *mv, int *eof)
{
    if (atoi(idtok) != mv->n) return -1;                  /* ref: src/scene.cpp:223-227 */
    o_material m; memset(&m, 0, sizeof m);
    char line[1024]; char *t[MAXTOK];
    for (int i = 0; i < 10; i++) {                         /* exactly 10 property lines, ref :232 */
        safe_getline(f, line, sizeof line, eof);
        int n = tokenize(line, t);
        if (n == 0) continue;
        if (!strcmp(t[0], "RGB")) { m.color.x = tokf(t, n, 1); m.color.y = tokf(t, n, 2); m.color.z = tokf(t, n, 3); }
        else if (!strcmp(t[0], "SPECEX")) m.specularExponent = tokf(t, n, 1);
        else if (!strcmp(t[0], "SPECRGB")) { m.specularColor.x = tokf(t, n, 1); m.specularColor.y = tokf(t, n, 2); m.specularColor.z = tokf(t, n, 3); }
        else if (!strcmp(t[0], "REFL")) m.hasReflective = tokf(t, n, 1);
        else if (!strcmp(t[0], "REFR")) m.hasRefractive = tokf(t, n, 1);
        else if (!strcmp(t[0], "REFRIOR")) m.indexOfRefraction = tokf(t, n, 1);
        else if (!strcmp(t[0], "SCATTER")) m.hasScatter = tokf(t, n, 1);
        else if (!strcmp(t[0], "ABSCOEFF")) { m.absorptionCoefficient.x = tokf(t, n, 1); m.absorptionCoefficient.y = tokf(t, n, 2); m.absorptionCoefficient.z = tokf(t, n, 3); }
        else if (!strcmp(t[0], "RSCTCOEFF")) m.reducedScatterCoefficient = tokf(t, n, 1);
        else if (!strcmp(t[0], "EMITTANCE")) m.emittance = tokf(t, n, 1);
    }
    if (mv->n == mv->cap) { mv->cap = mv->cap ? 2 * mv->cap : 16; mv->v = (o_material *)realloc(mv->v, (size_t)mv->cap * sizeof m); }
    mv->v[mv->n++] = m;
    return 1;
}

/* ref: src/scene.cpp:204-207 (tan in double, atan on a float -> atanf, /PI in double) */
o_vec2 o_camera_fov(float fovy, o_vec2 resolution)
{
    float yscaled = (float)tan((double)fovy * (O_PI / 180));
    float xscaled = (yscaled * resolution.x) / resolution.y;
    float fovx = (float)((double)(atanf(xscaled) * 180) / O_PI);
    o_vec2 fov; fov.x = fovx; fov.y = fovy;
    return fov;
}

static int load_camera(FILE *f, o_scene *s, int *eof)
{
    char line[1024]; char *t[MAXTOK];
    float fovy = 0; int resx = 0, resy = 0;
    for (int i = 0; i < 4; i++) {                          /* ref: src/scene.cpp:143-156 */
        safe_getline(f, line, sizeof line, eof);
        int n = tokenize(line, t);
        if (n == 0) continue;
        if (!strcmp(t[0], "RES")) { resx = (n > 1) ? atoi(t[1]) : 0; resy = (n > 2) ? atoi(t[2]) : 0; }
        else if (!strcmp(t[0], "FOVY")) fovy = tokf(t, n, 1);
        else if (!strcmp(t[0], "ITERATIONS")) s->iterations = (unsigned)((n > 1) ? atoi(t[1]) : 0);
        else if (!strcmp(t[0], "FILE")) { if (n > 1) { strncpy(s->image_name, t[1], sizeof s->image_name - 1); } }
    }
    s->camera.resolution.x = (float)resx; s->camera.resolution.y = (float)resy;
    int frames = 0;
    safe_getline(f, line, sizeof line, eof);
    while (line[0] && !*eof) {                             /* ref: src/scene.cpp:165-190 */
        int n = tokenize(line, t);
        if (n < 2 || strcmp(t[0], "frame") || atoi(t[1]) != frames) return -1;
        for (int i = 0; i < 3; i++) {
            safe_getline(f, line, sizeof line, eof);
            n = tokenize(line, t);
            if (n == 0) continue;
            o_vec3 v = { tokf(t, n, 1), tokf(t, n, 2), tokf(t, n, 3) };
            if (frames == g_frame) {                       /* the oracle keeps one frame */
                if (!strcmp(t[0], "EYE")) s->camera.position = v;
                else if (!strcmp(t[0], "VIEW")) s->camera.view = v;
                else if (!strcmp(t[0], "UP")) s->camera.up = v;
            }
        }
        frames++;
        safe_getline(f, line, sizeof line, eof);
    }
    s->n_frames_camera = frames;
    s->camera.fov = o_camera_fov(fovy, s->camera.resolution);
    return 1;
}

static int load_object(FILE *f, const char *idtok, geom_vec *gv, int rotat_units, int *eof)
{
    if (atoi(idtok) != gv->n) return -1;                   /* ref: src/scene.cpp:38-41 */
    o_staticGeom g; memset(&g, 0, sizeof g);
    char line[1024]; char *t[MAXTOK];
    safe_getline(f, line, sizeof line, eof);
    if (line[0] && !*eof) {                                /* whole-line compare, ref :49-70 */
        if (!strcmp(line, "sphere")) g.type = O_SPHERE;
        else if (!strcmp(line, "cube")) g.type = O_CUBE;
        else {
            char *dot = strchr(line, '.');
            char ext[8] = { 0 };
            if (dot) { strncpy(ext, dot + 1, 7); char *d2 = strchr(ext, '.'); if (d2) *d2 = 0; }
            if (!strcmp(ext, "obj")) g.type = O_MESH; else return -1;
        }
    }
    safe_getline(f, line, sizeof line, eof);
    if (line[0] && !*eof) { int n = tokenize(line, t); g.materialid = (n > 1) ? atoi(t[1]) : 0; }
    int frames = 0;
    safe_getline(f, line, sizeof line, eof);
    while (line[0] && !*eof) {                             /* ref: src/scene.cpp:88-113 */
        int n = tokenize(line, t);
        if (n < 2 || strcmp(t[0], "frame") || atoi(t[1]) != frames) return -1;
        for (int i = 0; i < 3; i++) {
            safe_getline(f, line, sizeof line, eof);
            n = tokenize(line, t);
            if (n == 0) continue;
            o_vec3 v = { tokf(t, n, 1), tokf(t, n, 2), tokf(t, n, 3) };
            if (frames == g_frame) {
                if (!strcmp(t[0], "TRANS")) g.translation = v;
                else if (!strcmp(t[0], "ROTAT")) g.rotation = v;
                else if (!strcmp(t[0], "SCALE")) g.scale = v;
            }
        }
        frames++;
        safe_getline(f, line, sizeof line, eof);
    }
    g.transform = o_buildTransformationMatrix(g.translation, g.rotation, g.scale, rotat_units, &g.inverseTransform);
    if (gv->n == gv->cap) { gv->cap = gv->cap ? 2 * gv->cap : 16; gv->v = (o_staticGeom *)realloc(gv->v, (size_t)gv->cap * sizeof g); }
    gv->v[gv->n++] = g;
    return 1;
}

int o_scene_load_frame(const char *path, int rotat_units, int frame, o_scene *out)
{
    g_frame = frame;
    int rc = o_scene_load(path, rotat_units, out);
    g_frame = 0;
    return rc;
}

int o_scene_load(const char *path, int rotat_units, o_scene *out)
{
    memset(out, 0, sizeof *out);
    FILE *f = fopen(path, "rb");
    if (!f) return -1;
    geom_vec gv = { 0, 0, 0 }; mat_vec mv = { 0, 0, 0 };
    char line[1024]; char *t[MAXTOK]; int eof = 0;
    while (!eof) {                                         /* ref: src/scene.cpp:17-33 */
        safe_getline(f, line, sizeof line, &eof);
        if (!line[0]) continue;
        int n = tokenize(line, t);
        if (n == 0) continue;
        if (!strcmp(t[0], "MATERIAL") && n > 1) load_material(f, t[1], &mv, &eof);
        else if (!strcmp(t[0], "OBJECT") && n > 1) load_object(f, t[1], &gv, rotat_units, &eof);
        else if (!strcmp(t[0], "CAMERA")) load_camera(f, out, &eof);
    }
    fclose(f);
    out->objects = gv.v; out->n_objects = gv.n;
    out->materials = mv.v; out->n_materials = mv.n;
    return 0;
}

void o_scene_free(o_scene *s)
{
    free(s->objects); free(s->materials);
    memset(s, 0, sizeof *s);
}


/* ------------------------------------------------------------------ */
/* Wavefront OBJ (MESH objects, ref: src/scene.cpp:57-66 names the file and loads nothing): `v x y z` and            */
/* `f a b c ...` records, indices 1-based or negative (relative), a/b/c forms, polygons cut into fans around their    */
/* first vertex; everything else is skipped.  *vertices_out = malloc'ed 9 floats per triangle (o_free_obj).          */
/* ------------------------------------------------------------------ */
int o_load_obj(const char *path, float **vertices_out, int *n_triangles_out)
{
    FILE *f = fopen(path, "rb");
    if (!f) return -1;
    float *verts = NULL, *tris = NULL;
    long nv = 0, capv = 0, nt = 0, capt = 0;
    char line[4096];
    while (fgets(line, sizeof line, f)) {
        char *s = line;
        while (*s == ' ' || *s == '\t') s++;
        if (s[0] == 'v' && (s[1] == ' ' || s[1] == '\t')) {
            double x, y, z;
            if (sscanf(s + 1, "%lf %lf %lf", &x, &y, &z) != 3) continue;
            if (nv == capv) { capv = capv ? 2 * capv : 256; verts = (float *)realloc(verts, (size_t)capv * 3 * sizeof(float)); }
            verts[3 * nv] = (float)x; verts[3 * nv + 1] = (float)y; verts[3 * nv + 2] = (float)z;
            nv++;
        } else if (s[0] == 'f' && (s[1] == ' ' || s[1] == '\t')) {
            long idx[64]; int n = 0, bad = 0;
            char *tok = strtok(s + 1, " \t\r\n");
            while (tok && n < 64) {
                long i = strtol(tok, NULL, 10);
                if (i < 0) i = nv + i + 1;
                if (i < 1 || i > nv) { bad = 1; break; }
                idx[n++] = i - 1;
                tok = strtok(NULL, " \t\r\n");
            }
            if (bad) continue;
            for (int k = 2; k < n; k++) {
                if (nt == capt) { capt = capt ? 2 * capt : 256; tris = (float *)realloc(tris, (size_t)capt * 9 * sizeof(float)); }
                const long a[3] = {idx[0], idx[k - 1], idx[k]};
                for (int j = 0; j < 3; j++) {
                    tris[9 * nt + 3 * j] = verts[3 * a[j]];
                    tris[9 * nt + 3 * j + 1] = verts[3 * a[j] + 1];
                    tris[9 * nt + 3 * j + 2] = verts[3 * a[j] + 2];
                }
                nt++;
            }
        }
    }
    fclose(f);
    free(verts);
    *vertices_out = tris;
    *n_triangles_out = (int)nt;
    return 0;
}

void o_free_obj(float *vertices) { free(vertices); }


/* ------------------------------------------------------------------ */
/* Motion blur: the scene state at shutter time t in [0,1] between two animation frames of one      */
/* object (ref: per-frame TRANS / ROTAT / SCALE arrays, src/sceneStructs.h:21-30): translation, rotation and scale  */
/* interpolated component-wise in fp32 as a + (b - a) * t, matrices rebuilt by buildTransformationMatrix.           */
/* ------------------------------------------------------------------ */
static float lerp1(float a, float b, float t) { return a + (b - a) * t; }
static o_vec3 lerp3(o_vec3 a, o_vec3 b, float t) { o_vec3 r; r.x = lerp1(a.x, b.x, t); r.y = lerp1(a.y, b.y, t); r.z = lerp1(a.z, b.z, t); return r; }

o_staticGeom o_interpolateGeom(const o_staticGeom *a, const o_staticGeom *b, float t, int rotat_units)
{
    o_staticGeom g = *a;
    g.translation = lerp3(a->translation, b->translation, t);
    g.rotation = lerp3(a->rotation, b->rotation, t);
    g.scale = lerp3(a->scale, b->scale, t);
    g.transform = o_buildTransformationMatrix(g.translation, g.rotation, g.scale, rotat_units, &g.inverseTransform);
    return g;
}

o_cameraData o_interpolateCamera(const o_cameraData *a, const o_cameraData *b, float t)
{
    o_cameraData c = *a;
    c.position = lerp3(a->position, b->position, t);
    c.view = lerp3(a->view, b->view, t);
    c.up = lerp3(a->up, b->up, t);
    return c;
}

/* shutter time of slice k of n: the middle of its interval, in fp32 */
float o_sliceTime(int k, int n) { return ((float)k + 0.5f) / (float)n; }
/* shutter time of knot k of n_knots (per-ray motion blur): both ends of the interval are knots */
float o_knotTime(int k, int n_knots) { return (float)k / (float)(n_knots - 1); }
