// pt_refstructs.h -- the argument types of the reference's renderer entry point, for building the shim
// and the headless driver OUTSIDE the reference tree.
//
// cudaRaytraceCore(uchar4*, camera*, int, int, material*, int, geom*, int) (ref: src/raytraceKernel.h:17)
// takes the reference's own structs (ref: src/sceneStructs.h, src/cudaMat4.h), which are built on glm vector
// types.  When pt_shim.cpp is compiled inside the reference tree it includes the reference's
// "sceneStructs.h" instead of this file; here the same layouts are declared on minimal
// stand-alone vector types so the drop-in can be built and tested without the reference.  Layouts are
// pinned below against the sizes/offsets measured on the reference (tests/golden/reference_vectors.json).
#pragma once
#include <cstddef>
#include <string>

namespace glm {
struct vec2 { float x, y; };
struct vec3 { float x, y, z; };
struct vec4 { float x, y, z, w; };
}  // namespace glm

struct uchar4 { unsigned char x, y, z, w; };
struct cudaMat4 { glm::vec4 x, y, z, w; };   // four rows

enum GEOMTYPE { SPHERE, CUBE, MESH };

struct material {
    glm::vec3 color;
    float specularExponent;
    glm::vec3 specularColor;
    float hasReflective, hasRefractive, indexOfRefraction, hasScatter;
    glm::vec3 absorptionCoefficient;
    float reducedScatterCoefficient, emittance;
};
static_assert(sizeof(material) == 64 && offsetof(material, emittance) == 60, "material layout");

struct ray { glm::vec3 origin, direction; };
static_assert(sizeof(ray) == 24, "ray layout");

// per-frame arrays, one entry per animation frame
struct geom {
    enum GEOMTYPE type;
    int materialid;
    int frames;
    glm::vec3 *translations, *rotations, *scales;
    cudaMat4 *transforms, *inverseTransforms;
};
static_assert(sizeof(geom) == 56 && offsetof(geom, transforms) == 40 && offsetof(geom, inverseTransforms) == 48, "geom layout");

struct camera {
    glm::vec2 resolution;
    glm::vec3 *positions, *views, *ups;
    int frames;
    glm::vec2 fov;
    unsigned int iterations;
    glm::vec3 *image;
    ray *rayList;
    std::string imageName;
};
static_assert(offsetof(camera, fov) == 36 && offsetof(camera, iterations) == 44 && offsetof(camera, image) == 48 &&
                  offsetof(camera, imageName) == 64, "camera layout");

void cudaRaytraceCore(uchar4 *pos, camera *renderCam, int frame, int iterations, material *materials,
                      int numberOfMaterials, geom *geoms, int numberOfGeoms);
