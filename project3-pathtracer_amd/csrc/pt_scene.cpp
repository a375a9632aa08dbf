// pt_scene.cpp -- scene-file loader of the renderer: keeps the reference's text format
// (ref: src/scene.cpp; README.md:160-217) and produces the per-frame row-major matrices the hot path
// consumes (ref: src/utilities.cpp:74-90, GLM 0.9.5.4 translate/rotate/scale/inverse semantics, fp32).
//
// One-time host work; not accelerated.  Compiled with -ffp-contract=off so that matrices equal the
// reference's bit for bit (checked against real GLM vectors in tests/test_oracle_kat.py (loader dump) and tests/test_abi.py).
#include "pt_scene.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <fstream>
#include <iterator>
#include <sstream>

namespace ptamd {
namespace {

// 4x4 fp32 matrix in glm's column-major convention: col[c][r]
struct M4 {
    float col[4][4];
    static M4 identity()
    {
        M4 m;
        for (int c = 0; c < 4; ++c)
            for (int r = 0; r < 4; ++r) m.col[c][r] = (c == r) ? 1.0f : 0.0f;
        return m;
    }
};

// glm::operator*(mat4, mat4): Result[j] = A[0]*B[j][0] + A[1]*B[j][1] + A[2]*B[j][2] + A[3]*B[j][3]
M4 mul(const M4 &A, const M4 &B)
{
    M4 R;
    for (int j = 0; j < 4; ++j)
        for (int r = 0; r < 4; ++r)
            R.col[j][r] = A.col[0][r] * B.col[j][0] + A.col[1][r] * B.col[j][1] + A.col[2][r] * B.col[j][2] +
                          A.col[3][r] * B.col[j][3];
    return R;
}

// glm::translate(mat4(), v)
M4 translation(pt_vec3 v)
{
    const M4 I = M4::identity();
    M4 R = I;
    for (int r = 0; r < 4; ++r) R.col[3][r] = I.col[0][r] * v.x + I.col[1][r] * v.y + I.col[2][r] * v.z + I.col[3][r];
    return R;
}

// glm::scale(mat4(), v)
M4 scaling(pt_vec3 v)
{
    const M4 I = M4::identity();
    M4 R;
    for (int r = 0; r < 4; ++r) {
        R.col[0][r] = I.col[0][r] * v.x;
        R.col[1][r] = I.col[1][r] * v.y;
        R.col[2][r] = I.col[2][r] * v.z;
        R.col[3][r] = I.col[3][r];
    }
    return R;
}

// glm::rotate(mat4(), angle, axis) with GLM_FORCE_RADIANS (ref: src/utilities.cpp:7)
M4 rotation(float angle, float ax, float ay, float az)
{
    const float c = cosf(angle), s = sinf(angle);
    const float inv = 1.0f / sqrtf(ax * ax + ay * ay + az * az);
    const float a[3] = {ax * inv, ay * inv, az * inv};
    const float t[3] = {(1.0f - c) * a[0], (1.0f - c) * a[1], (1.0f - c) * a[2]};
    float Rm[3][3];
    Rm[0][0] = c + t[0] * a[0];
    Rm[0][1] = 0 + t[0] * a[1] + s * a[2];
    Rm[0][2] = 0 + t[0] * a[2] - s * a[1];
    Rm[1][0] = 0 + t[1] * a[0] - s * a[2];
    Rm[1][1] = c + t[1] * a[1];
    Rm[1][2] = 0 + t[1] * a[2] + s * a[0];
    Rm[2][0] = 0 + t[2] * a[0] + s * a[1];
    Rm[2][1] = 0 + t[2] * a[1] - s * a[0];
    Rm[2][2] = c + t[2] * a[2];
    const M4 I = M4::identity();
    M4 R;
    for (int j = 0; j < 3; ++j)
        for (int r = 0; r < 4; ++r) R.col[j][r] = I.col[0][r] * Rm[j][0] + I.col[1][r] * Rm[j][1] + I.col[2][r] * Rm[j][2];
    for (int r = 0; r < 4; ++r) R.col[3][r] = I.col[3][r];
    return R;
}

// glm::inverse(mat4): cofactor expansion, glm 0.9.5.4 detail/type_mat4x4.inl compute_inverse
M4 inverse(const M4 &m)
{
    auto e = [&](int c, int r) { return m.col[c][r]; };
    const float c00 = e(2, 2) * e(3, 3) - e(3, 2) * e(2, 3), c02 = e(1, 2) * e(3, 3) - e(3, 2) * e(1, 3),
                c03 = e(1, 2) * e(2, 3) - e(2, 2) * e(1, 3);
    const float c04 = e(2, 1) * e(3, 3) - e(3, 1) * e(2, 3), c06 = e(1, 1) * e(3, 3) - e(3, 1) * e(1, 3),
                c07 = e(1, 1) * e(2, 3) - e(2, 1) * e(1, 3);
    const float c08 = e(2, 1) * e(3, 2) - e(3, 1) * e(2, 2), c10 = e(1, 1) * e(3, 2) - e(3, 1) * e(1, 2),
                c11 = e(1, 1) * e(2, 2) - e(2, 1) * e(1, 2);
    const float c12 = e(2, 0) * e(3, 3) - e(3, 0) * e(2, 3), c14 = e(1, 0) * e(3, 3) - e(3, 0) * e(1, 3),
                c15 = e(1, 0) * e(2, 3) - e(2, 0) * e(1, 3);
    const float c16 = e(2, 0) * e(3, 2) - e(3, 0) * e(2, 2), c18 = e(1, 0) * e(3, 2) - e(3, 0) * e(1, 2),
                c19 = e(1, 0) * e(2, 2) - e(2, 0) * e(1, 2);
    const float c20 = e(2, 0) * e(3, 1) - e(3, 0) * e(2, 1), c22 = e(1, 0) * e(3, 1) - e(3, 0) * e(1, 1),
                c23 = e(1, 0) * e(2, 1) - e(2, 0) * e(1, 1);
    const float F0[4] = {c00, c00, c02, c03}, F1[4] = {c04, c04, c06, c07}, F2[4] = {c08, c08, c10, c11};
    const float F3[4] = {c12, c12, c14, c15}, F4[4] = {c16, c16, c18, c19}, F5[4] = {c20, c20, c22, c23};
    const float V0[4] = {e(1, 0), e(0, 0), e(0, 0), e(0, 0)}, V1[4] = {e(1, 1), e(0, 1), e(0, 1), e(0, 1)};
    const float V2[4] = {e(1, 2), e(0, 2), e(0, 2), e(0, 2)}, V3[4] = {e(1, 3), e(0, 3), e(0, 3), e(0, 3)};
    const float SA[4] = {+1, -1, +1, -1}, SB[4] = {-1, +1, -1, +1};
    M4 inv;
    for (int k = 0; k < 4; ++k) {
        inv.col[0][k] = (V1[k] * F0[k] - V2[k] * F1[k] + V3[k] * F2[k]) * SA[k];
        inv.col[1][k] = (V0[k] * F0[k] - V2[k] * F3[k] + V3[k] * F4[k]) * SB[k];
        inv.col[2][k] = (V0[k] * F1[k] - V1[k] * F3[k] + V3[k] * F5[k]) * SA[k];
        inv.col[3][k] = (V0[k] * F2[k] - V1[k] * F4[k] + V2[k] * F5[k]) * SB[k];
    }
    const float d0 = e(0, 0) * inv.col[0][0], d1 = e(0, 1) * inv.col[1][0], d2 = e(0, 2) * inv.col[2][0],
                d3 = e(0, 3) * inv.col[3][0];
    const float oneOverDet = 1.0f / ((d0 + d1) + (d2 + d3));
    for (int c = 0; c < 4; ++c)
        for (int r = 0; r < 4; ++r) inv.col[c][r] = inv.col[c][r] * oneOverDet;
    return inv;
}

// glmMat4ToCudaMat4 (ref: src/utilities.cpp:83-90): transpose -> four rows
pt_mat4 toRows(const M4 &m)
{
    pt_mat4 r;
    r.x = {m.col[0][0], m.col[1][0], m.col[2][0], m.col[3][0]};
    r.y = {m.col[0][1], m.col[1][1], m.col[2][1], m.col[3][1]};
    r.z = {m.col[0][2], m.col[1][2], m.col[2][2], m.col[3][2]};
    r.w = {m.col[0][3], m.col[1][3], m.col[2][3], m.col[3][3]};
    return r;
}

// Line source of the loader: the whole file in memory, cut at LF, CRLF or CR (the line ends the reference's reader
// accepts, ref: src/utilities.cpp:109-140).  A last line without a line end is still a line; `more()` turns false only
// once a read found nothing at all, which is when the reference's stream stops being good().
}  // namespace

bool LineReader::open(const std::string &path)
{
    std::ifstream f(path.c_str(), std::ios::binary);
    if (!f.is_open()) return false;
    text_.assign(std::istreambuf_iterator<char>(f), std::istreambuf_iterator<char>());
    pos_ = 0;
    exhausted_ = false;
    return true;
}

void LineReader::next(std::string &line)
{
    const size_t n = text_.size();
    if (pos_ >= n) {
        line.clear();
        exhausted_ = true;
        return;
    }
    const size_t end = text_.find_first_of("\r\n", pos_);
    if (end == std::string::npos) {
        line.assign(text_, pos_, n - pos_);
        pos_ = n;
        return;
    }
    line.assign(text_, pos_, end - pos_);
    pos_ = end + ((text_[end] == '\r' && end + 1 < n && text_[end + 1] == '\n') ? 2 : 1);
}

namespace {
std::vector<std::string> tokenizeString(const std::string &str)
{
    std::istringstream ss(str);
    std::vector<std::string> out;
    std::string w;
    while (ss >> w) out.push_back(w);
    return out;
}

float fieldf(const std::vector<std::string> &t, size_t i) { return i < t.size() ? (float)atof(t[i].c_str()) : 0.0f; }
pt_vec3 field3(const std::vector<std::string> &t) { return {fieldf(t, 1), fieldf(t, 2), fieldf(t, 3)}; }

}  // namespace

pt_mat4 buildTransformationMatrix(pt_vec3 t, pt_vec3 r, pt_vec3 s, int rotat_units, pt_mat4 *inverse_out)
{
    if (rotat_units == PT_ROTAT_DEGREES) {
        const float k = (float)(3.1415926535897932384626422832795028841971 / 180.0);
        r.x = r.x * k; r.y = r.y * k; r.z = r.z * k;
    }
    const M4 translationMat = translation(t);
    M4 rotationMat = rotation(r.x, 1, 0, 0);
    rotationMat = mul(rotationMat, rotation(r.y, 0, 1, 0));
    rotationMat = mul(rotationMat, rotation(r.z, 0, 0, 1));
    const M4 scaleMat = scaling(s);
    const M4 transform = mul(mul(translationMat, rotationMat), scaleMat);
    if (inverse_out) *inverse_out = toRows(inverse(transform));
    return toRows(transform);
}

pt_vec2 cameraFov(float fovy, pt_vec2 resolution)
{
    const double PI = 3.1415926535897932384626422832795028841971;
    const float yscaled = (float)tan((double)fovy * (PI / 180));
    const float xscaled = (yscaled * resolution.x) / resolution.y;
    const float fovx = (float)((double)(atanf(xscaled) * 180) / PI);
    return {fovx, fovy};
}

bool loadObjTriangles(const std::string &path, std::vector<float> &out)
{
    LineReader f;
    if (!f.open(path)) return false;
    std::vector<pt_vec3> verts;
    std::string line;
    for (f.next(line); f.more(); f.next(line)) {
        const std::vector<std::string> t = tokenizeString(line);
        if (t.empty()) continue;
        if (t[0] == "v" && t.size() >= 4) {
            verts.push_back({(float)atof(t[1].c_str()), (float)atof(t[2].c_str()), (float)atof(t[3].c_str())});
        } else if (t[0] == "f" && t.size() >= 4) {
            std::vector<long> idx;
            for (size_t k = 1; k < t.size(); ++k) {
                long i = atol(t[k].c_str());                 // "a", "a/b", "a//c", "a/b/c": the vertex index comes first
                if (i < 0) i = (long)verts.size() + i + 1;   // negative: relative to the vertices read so far
                if (i < 1 || i > (long)verts.size()) { idx.clear(); break; }
                idx.push_back(i - 1);
            }
            for (size_t k = 2; k < idx.size(); ++k) {        // fan around the first vertex
                const pt_vec3 tri[3] = {verts[(size_t)idx[0]], verts[(size_t)idx[k - 1]], verts[(size_t)idx[k]]};
                for (const pt_vec3 &v : tri) { out.push_back(v.x); out.push_back(v.y); out.push_back(v.z); }
            }
        }
    }
    return true;
}

scene::scene(const std::string &filename, int rotat_units) : rotat_units_(rotat_units)
{
    const size_t slash = filename.find_last_of("/\\");
    dir_ = slash == std::string::npos ? std::string() : filename.substr(0, slash + 1);
    if (!fp_in.open(filename)) {
        errors.push_back("cannot open " + filename);
        return;
    }
    ok = true;
    while (fp_in.more()) {
        std::string line;
        fp_in.next(line);
        if (line.empty()) continue;
        const std::vector<std::string> tokens = tokenizeString(line);
        if (tokens.empty()) continue;
        if (tokens[0] == "MATERIAL" && tokens.size() > 1) loadMaterial(tokens[1]);
        else if (tokens[0] == "OBJECT" && tokens.size() > 1) loadObject(tokens[1]);
        else if (tokens[0] == "CAMERA") loadCamera();
    }
}

int scene::loadMaterial(const std::string &materialid)
{
    if (atoi(materialid.c_str()) != (int)materials.size()) {
        errors.push_back("MATERIAL ID does not match expected number of materials");
        return -1;
    }
    pt_material m;
    memset(&m, 0, sizeof m);
    for (int i = 0; i < 10; ++i) {                       // exactly 10 property lines
        std::string line;
        fp_in.next(line);
        const std::vector<std::string> t = tokenizeString(line);
        if (t.empty()) continue;
        const std::string &k = t[0];
        if (k == "RGB") m.color = field3(t);
        else if (k == "SPECEX") m.specularExponent = fieldf(t, 1);
        else if (k == "SPECRGB") m.specularColor = field3(t);
        else if (k == "REFL") m.hasReflective = fieldf(t, 1);
        else if (k == "REFR") m.hasRefractive = fieldf(t, 1);
        else if (k == "REFRIOR") m.indexOfRefraction = fieldf(t, 1);
        else if (k == "SCATTER") m.hasScatter = fieldf(t, 1);
        else if (k == "ABSCOEFF") m.absorptionCoefficient = field3(t);
        else if (k == "RSCTCOEFF") m.reducedScatterCoefficient = fieldf(t, 1);
        else if (k == "EMITTANCE") m.emittance = fieldf(t, 1);
    }
    materials.push_back(m);
    return 1;
}

int scene::loadCamera()
{
    camera cam;
    float fovy = 0;
    for (int i = 0; i < 4; ++i) {
        std::string line;
        fp_in.next(line);
        const std::vector<std::string> t = tokenizeString(line);
        if (t.empty()) continue;
        if (t[0] == "RES") cam.resolution = {(float)(t.size() > 1 ? atoi(t[1].c_str()) : 0), (float)(t.size() > 2 ? atoi(t[2].c_str()) : 0)};
        else if (t[0] == "FOVY") fovy = fieldf(t, 1);
        else if (t[0] == "ITERATIONS") cam.iterations = (unsigned)(t.size() > 1 ? atoi(t[1].c_str()) : 0);
        else if (t[0] == "FILE") cam.imageName = t.size() > 1 ? t[1] : "";
    }
    int frameCount = 0;
    std::string line;
    fp_in.next(line);
    while (!line.empty() && fp_in.more()) {
        std::vector<std::string> t = tokenizeString(line);
        if (t.size() < 2 || t[0] != "frame" || atoi(t[1].c_str()) != frameCount) {
            errors.push_back("Incorrect frame count!");
            return -1;
        }
        for (int i = 0; i < 3; ++i) {
            fp_in.next(line);
            t = tokenizeString(line);
            if (t.empty()) continue;
            if (t[0] == "EYE") cam.positions.push_back(field3(t));
            else if (t[0] == "VIEW") cam.views.push_back(field3(t));
            else if (t[0] == "UP") cam.ups.push_back(field3(t));
        }
        frameCount++;
        fp_in.next(line);
    }
    cam.frames = frameCount;
    cam.fov = cameraFov(fovy, cam.resolution);
    const size_t npix = (size_t)(int)cam.resolution.x * (size_t)(int)cam.resolution.y;
    cam.image.assign(npix, pt_vec3{0, 0, 0});
    renderCam = cam;
    return 1;
}

int scene::loadObject(const std::string &objectid)
{
    if (atoi(objectid.c_str()) != (int)objects.size()) {
        errors.push_back("OBJECT ID does not match expected number of objects");
        return -1;
    }
    geom g;
    std::string line;
    fp_in.next(line);
    if (!line.empty() && fp_in.more()) {                 // the whole line is compared (ref: src/scene.cpp:49-70)
        if (line == "sphere") g.type = PT_SPHERE;
        else if (line == "cube") g.type = PT_CUBE;
        else {
            std::istringstream liness(line);
            std::string name, extension;
            getline(liness, name, '.');
            getline(liness, extension, '.');
            if (extension == "obj") {
                g.type = PT_MESH;
                g.meshFile = line;
                if (!loadObjTriangles(dir_ + line, g.meshVertices) && !loadObjTriangles(line, g.meshVertices))
                    errors.push_back("cannot read mesh " + line + " (the object stays empty)");
            } else {
                errors.push_back(line + " is not a valid object type!");
                return -1;
            }
        }
    }
    fp_in.next(line);
    if (!line.empty() && fp_in.more()) {
        const std::vector<std::string> t = tokenizeString(line);
        g.materialid = t.size() > 1 ? atoi(t[1].c_str()) : 0;
    }
    int frameCount = 0;
    fp_in.next(line);
    while (!line.empty() && fp_in.more()) {
        std::vector<std::string> t = tokenizeString(line);
        if (t.size() < 2 || t[0] != "frame" || atoi(t[1].c_str()) != frameCount) {
            errors.push_back("Incorrect frame count!");
            return -1;
        }
        for (int i = 0; i < 3; ++i) {
            fp_in.next(line);
            t = tokenizeString(line);
            if (t.empty()) continue;
            if (t[0] == "TRANS") g.translations.push_back(field3(t));
            else if (t[0] == "ROTAT") g.rotations.push_back(field3(t));
            else if (t[0] == "SCALE") g.scales.push_back(field3(t));
        }
        frameCount++;
        fp_in.next(line);
    }
    g.frames = frameCount;
    if ((int)g.translations.size() != frameCount || (int)g.rotations.size() != frameCount || (int)g.scales.size() != frameCount) {
        errors.push_back("object " + objectid + ": every frame needs TRANS, ROTAT and SCALE");
        return -1;
    }
    for (int i = 0; i < frameCount; ++i) {
        pt_mat4 inv;
        g.transforms.push_back(buildTransformationMatrix(g.translations[i], g.rotations[i], g.scales[i], rotat_units_, &inv));
        g.inverseTransforms.push_back(inv);
    }
    objects.push_back(g);
    return 1;
}

}  // namespace ptamd

// ---------------------------------------------------------------------------------------------
// C-ABI
// ---------------------------------------------------------------------------------------------
extern "C" {

int pt_scene_load(const char *path, int rotat_units, pt_scene **out)
{
    if (!path || !out) return PT_ERR_INVALID;
    *out = nullptr;
    ptamd::scene *s = new ptamd::scene(path, rotat_units);
    if (!s->ok) {
        delete s;
        return PT_ERR_INVALID;
    }
    pt_scene *h = new pt_scene;
    h->s = s;
    *out = h;
    return PT_OK;
}

void pt_scene_free(pt_scene *s)
{
    if (!s) return;
    delete s->s;
    delete s;
}

int pt_scene_counts(const pt_scene *s, int *n_objects, int *n_materials, int *n_camera_frames)
{
    if (!s) return PT_ERR_INVALID;
    if (n_objects) *n_objects = (int)s->s->objects.size();
    if (n_materials) *n_materials = (int)s->s->materials.size();
    if (n_camera_frames) *n_camera_frames = s->s->renderCam.frames;
    return PT_OK;
}

int pt_scene_camera_info(const pt_scene *s, unsigned *iterations, char *image_name, size_t cap)
{
    if (!s) return PT_ERR_INVALID;
    if (iterations) *iterations = s->s->renderCam.iterations;
    if (image_name && cap) {
        strncpy(image_name, s->s->renderCam.imageName.c_str(), cap - 1);
        image_name[cap - 1] = 0;
    }
    return PT_OK;
}

int pt_scene_mesh(const pt_scene *s, int object, const float **vertices_out, int *n_triangles_out)
{
    if (!s || !s->s || object < 0 || object >= (int)s->s->objects.size() || !vertices_out || !n_triangles_out) return PT_ERR_INVALID;
    const ptamd::geom &g = s->s->objects[(size_t)object];
    *vertices_out = g.meshVertices.empty() ? nullptr : g.meshVertices.data();
    *n_triangles_out = (int)(g.meshVertices.size() / 9);
    return PT_OK;
}

int pt_scene_get_frame(const pt_scene *s, int frame, pt_static_geom *geoms_out, pt_material *materials_out,
                       pt_camera_data *camera_out)
{
    if (!s || frame < 0) return PT_ERR_INVALID;
    const ptamd::scene &sc = *s->s;
    if (geoms_out) {
        for (size_t i = 0; i < sc.objects.size(); ++i) {
            const ptamd::geom &g = sc.objects[i];
            if (frame >= g.frames) return PT_ERR_INVALID;
            pt_static_geom &o = geoms_out[i];        // what ref: src/raytraceKernel.cu:123-134 packs
            o.type = g.type;
            o.materialid = g.materialid;
            o.translation = g.translations[frame];
            o.rotation = g.rotations[frame];
            o.scale = g.scales[frame];
            o.transform = g.transforms[frame];
            o.inverseTransform = g.inverseTransforms[frame];
        }
    }
    if (materials_out)
        for (size_t i = 0; i < sc.materials.size(); ++i) materials_out[i] = sc.materials[i];
    if (camera_out) {
        const ptamd::camera &c = sc.renderCam;
        if (frame >= c.frames || (int)c.positions.size() <= frame || (int)c.views.size() <= frame || (int)c.ups.size() <= frame)
            return PT_ERR_INVALID;
        camera_out->resolution = c.resolution;       // ref: src/raytraceKernel.cu:141-146
        camera_out->position = c.positions[frame];
        camera_out->view = c.views[frame];
        camera_out->up = c.ups[frame];
        camera_out->fov = c.fov;
    }
    return PT_OK;
}

int pt_camera_set_resolution(pt_camera_data *cam, int width, int height)
{
    if (!cam || width < 1 || height < 1) return PT_ERR_INVALID;
    cam->resolution = {(float)width, (float)height};
    cam->fov = ptamd::cameraFov(cam->fov.y, cam->resolution);
    return PT_OK;
}

}  // extern "C"
