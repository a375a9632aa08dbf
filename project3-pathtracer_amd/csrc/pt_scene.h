// pt_scene.h -- host-side mirror of the reference's scene interface (ref: src/scene.h:19-32,
// src/sceneStructs.h:21-61): same class name, same public members (objects, materials, renderCam) and the
// same per-frame array layout, built on the POD types of include/pt_abi.h instead of glm.
#pragma once
#include <fstream>
#include <string>
#include <vector>

#include "../../include/pt_abi.h"

namespace ptamd {

// geom, ref: src/sceneStructs.h:21-30 (per-frame arrays owned by the scene)
struct geom {
    int type = PT_SPHERE;
    int materialid = 0;
    int frames = 0;                 // the reference never initialises this field; here it is the frame count
    std::vector<pt_vec3> translations, rotations, scales;
    std::vector<pt_mat4> transforms, inverseTransforms;
    // MESH only: the triangles of the .obj file the object names (9 floats each: v0, v1, v2 in object space).  The
    // reference parses the type and loads nothing (ref: src/scene.cpp:57-66); empty when the file is not found.
    std::string meshFile;
    std::vector<float> meshVertices;
};

// camera, ref: src/sceneStructs.h:50-61
struct camera {
    pt_vec2 resolution = {0, 0};
    std::vector<pt_vec3> positions, views, ups;
    int frames = 0;
    pt_vec2 fov = {0, 0};
    unsigned int iterations = 0;
    std::vector<pt_vec3> image;     // W*H running-mean framebuffer, zero-initialised (ref: src/scene.cpp:212-216)
    std::string imageName;
};

// the scene file as a sequence of lines (LF, CRLF or CR line ends)
class LineReader {
public:
    bool open(const std::string &path);
    void next(std::string &line);            // the next line without its line end; empty once the text is used up
    bool more() const { return !exhausted_; }   // false after a read that found nothing left
private:
    std::string text_;
    size_t pos_ = 0;
    bool exhausted_ = true;
};

class scene {
public:
    explicit scene(const std::string &filename, int rotat_units = PT_ROTAT_RADIANS);
    std::vector<geom> objects;
    std::vector<pt_material> materials;
    camera renderCam;
    bool ok = false;                // file opened
    std::vector<std::string> errors;

private:
    LineReader fp_in;
    int rotat_units_;
    std::string dir_;               // directory of the scene file: .obj names are relative to it
    int loadMaterial(const std::string &materialid);
    int loadObject(const std::string &objectid);
    int loadCamera();
};

// utilityCore::buildTransformationMatrix + glmMat4ToCudaMat4 (ref: src/utilities.cpp:74-90)
pt_mat4 buildTransformationMatrix(pt_vec3 translation, pt_vec3 rotation, pt_vec3 scale, int rotat_units,
                                  pt_mat4 *inverse_out);
pt_vec2 cameraFov(float fovy, pt_vec2 resolution);   // ref: src/scene.cpp:204-207
// Wavefront OBJ: `v x y z` and `f a b c ...` records (1-based or negative indices, a/b/c forms, polygons cut into
// fans); everything else is skipped.  Appends 9 floats per triangle; false when the file cannot be read.
bool loadObjTriangles(const std::string &path, std::vector<float> &out);

}  // namespace ptamd

struct pt_scene {
    ptamd::scene *s;
};
