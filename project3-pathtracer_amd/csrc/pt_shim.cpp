// pt_shim.cpp -- reference-side binding: the renderer entry point with the reference's exact signature
// (ref: src/raytraceKernel.h:17), implemented on the C-ABI of libptamd.so.
//
// Drop-in use: compile this file in place of src/raytraceKernel.cu, with "sceneStructs.h" from the
// reference instead of pt_refstructs.h, and link libptamd.so.  It does what the host
// body of the reference's function does (ref: src/raytraceKernel.cu:106-165) -- flatten geoms[i] at `frame`
// into staticGeom records, pack cameraData, run one iteration, fill the PBO and renderCam->image -- but on
// a persistent context: no per-call cudaMalloc/cudaFree, and the D2H copy of the image only when the caller
// will look at it.
//
// Knobs the reference keeps as source constants come from the environment (read once):
//   PT_DEPTH (8)          bounces per path (the reference's traceDepth, src/raytraceKernel.cu:110)
//   PT_RR_START (-1)      first bounce with Russian roulette, -1 = off
//   PT_DIRECT_LIGHT (0)   1 = sample the lights explicitly at diffuse vertices (getRandomPointOnCube / ...OnSphere)
//   PT_ABSORPTION (0)     1 = Beer-Lambert absorption (ABSCOEFF) inside refractive objects (calculateTransmission)
//   PT_SCATTER (0)        1 = subsurface random walk inside SCATTER materials (calculateScatterAndAbsorption)
//   PT_LENS_RADIUS (0), PT_FOCAL_DISTANCE (1)   thin-lens camera (depth of field); radius 0 = pinhole
//   PT_SEED (0)           RNG stream selector
//   PT_DEVICES (0)        comma-separated HIP devices; with several, each renders a band of rows of the frame
//                         (pt_multi_*), the bands are gathered into renderCam->image; PBO output then needs 1 device
//   PT_STRIP_ROWS (8)     with several devices: rows per interleaved strip (device k renders strips k, k+n, ...); 0 = one
//                         contiguous band per device
//   PT_MOTION_SLICES (0)  > 1: motion blur over the interval to the next animation frame, that many shutter slices
//   PT_MOTION_PER_RAY (0) 1: a shutter time per ray instead (pt_options.motion_per_ray); PT_MOTION_SLICES >= 1 then counts
//                         the linear segments between the knot states
//                         (PT_ROTAT_UNITS 0 radians / 1 degrees: the unit of the scene's ROTAT values)
//   PT_SHIM_BATCH (1)     iterations that may be pending inside the shim before they are rendered together
//                         (only while nobody can observe them: no PBO, no read-back due); 1 = render every call
//   PT_READBACK_EVERY (1) copy renderCam->image back every N iterations; 1 = on every call, as the reference does
//                         (src/raytraceKernel.cu:154); 0 = only on the last one (iterations == renderCam->iterations,
//                         which is when src/main.cpp:114-125 reads it)
// The defaults reproduce the reference call for call: every iteration is rendered and renderCam->image updated
// before cudaRaytraceCore returns.  A caller that only reads the image at the end (src/main.cpp does) may opt in to
// deferral with pt_shim_configure(64, 0) -- the headless driver pt_main.cpp does -- and must then call
// pt_shim_flush() if it stops before iterations == renderCam->iterations: that renders what is still pending and
// copies the image back, so no accepted iteration is lost.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "../../include/pt_abi.h"
#include "pt_refstructs.h"
#include "pt_shim.h"

namespace {

struct ShimState {
    pt_multi *ctx = nullptr;
    int ndev = 1;
    std::vector<pt_static_geom> geoms;
    std::vector<pt_material> mats;
    pt_camera_data cam;
    bool have_cam = false;
    int readback_every = 1;     // PT_READBACK_EVERY / pt_shim_configure
    int last_iteration = 0;     // the last iteration handed to this shim (rendered or pending)
    int pend_first = 0, pend_count = 0;   // iterations accepted but not enqueued yet (rendered in batches)
    int defer = 1;              // PT_SHIM_BATCH / pt_shim_configure: how many iterations may be pending; 1 = render on every call
    bool configured = false;    // pt_shim_configure was called: the environment no longer decides defer / readback_every
    float *last_image = nullptr;   // renderCam->image of the last call (pt_shim_flush copies the image back there)
    std::vector<std::vector<float>> mesh_vertices;   // pt_shim_set_meshes: copies of the triangles ...
    std::vector<pt_mesh> meshes;                     // ... and the descriptors pointing into them
    bool meshes_dirty = false;
    int motion_slices = -1, motion_rotat = 0;      // -1: take PT_MOTION_SLICES / PT_ROTAT_UNITS at the first call
    bool motion_dirty = true;
    int motion_frame = -1;
    int motion_per_ray = 0;                        // PT_MOTION_PER_RAY: PT_MOTION_SLICES then counts linear segments (>= 1)
};
ShimState g;

int env_int(const char *name, int dflt)
{
    const char *v = getenv(name);
    return (v && *v) ? atoi(v) : dflt;
}

// checkCUDAError's convention (ref: src/raytraceKernel.cu:19-25): print and exit
void check(int rc, const char *msg)
{
    if (rc == PT_OK) return;
    fprintf(stderr, "Cuda error: %s: %s.\n", msg, pt_last_error());
    exit(EXIT_FAILURE);
}

// Iterations whose outputs nobody can observe yet (no PBO, no image read-back) are collected and rendered as one
// pt_render call, so that the renderer can keep several of them in flight (pt_options.batch).
void flush_pending()
{
    if (g.pend_count > 0) check(pt_multi_render(g.ctx, g.pend_first, g.pend_count), "pt_render");
    g.pend_count = 0;
}

}  // namespace

void pt_shim_configure(int batch, int readback_every)
{
    if (g.ctx) flush_pending();
    g.defer = batch < 1 ? 1 : batch;
    g.readback_every = readback_every < 0 ? 0 : readback_every;
    g.configured = true;
}

void pt_shim_set_meshes(const pt_mesh *meshes, int n)
{
    if (g.ctx) flush_pending();
    g.mesh_vertices.clear();
    g.meshes.clear();
    for (int k = 0; k < n; ++k) {
        g.mesh_vertices.emplace_back(meshes[k].vertices, meshes[k].vertices + 9 * (size_t)meshes[k].n_triangles);
        g.meshes.push_back(meshes[k]);
    }
    for (size_t k = 0; k < g.meshes.size(); ++k) g.meshes[k].vertices = g.mesh_vertices[k].data();
    g.meshes_dirty = true;
}

void pt_shim_set_motion(int slices, int rotat_units)
{
    if (g.ctx) flush_pending();
    g.motion_slices = slices < 0 ? 0 : slices;
    g.motion_rotat = rotat_units ? 1 : 0;
    g.motion_dirty = true;
}

void pt_shim_flush(void)
{
    if (!g.ctx) return;
    flush_pending();
    if (g.last_image) check(pt_multi_download_image(g.ctx, g.last_image), "pt_download_image");
    check(pt_multi_synchronize(g.ctx), "Kernel failed!");
}

void cudaRaytraceCore(uchar4 *PBOpos, camera *renderCam, int frame, int iterations, material *materials,
                      int numberOfMaterials, geom *geoms, int numberOfGeoms)
{
    if (!g.ctx) {
        std::vector<int> devs;
        const char *dl = getenv("PT_DEVICES");
        if (dl && *dl) {
            for (const char *q = dl; *q;) {
                devs.push_back(atoi(q));
                while (*q && *q != ',') ++q;
                if (*q == ',') ++q;
            }
        }
        if (devs.empty()) devs.push_back(env_int("PT_DEVICE", 0));
        g.ndev = (int)devs.size();
        // pt_options travels by pointer without a size: a library built against another header would read or miss fields
        if (pt_abi_version() != PT_ABI_VERSION || pt_options_size() != sizeof(pt_options)) {
            fprintf(stderr, "Cuda error: libptamd.so has ABI %d / pt_options of %zu B, this binding was built for ABI %d / %zu B.\n",
                    pt_abi_version(), pt_options_size(), PT_ABI_VERSION, sizeof(pt_options));
            exit(EXIT_FAILURE);
        }
        check(pt_multi_create(devs.data(), g.ndev, &g.ctx), "pt_create");
        pt_options o;
        pt_default_options(&o);
        o.depth = env_int("PT_DEPTH", o.depth);
        o.rr_start = env_int("PT_RR_START", o.rr_start);
        o.direct_light = env_int("PT_DIRECT_LIGHT", o.direct_light);
        o.absorption = env_int("PT_ABSORPTION", o.absorption);
        o.scatter = env_int("PT_SCATTER", o.scatter);
        o.motion_per_ray = env_int("PT_MOTION_PER_RAY", o.motion_per_ray);
        g.motion_per_ray = o.motion_per_ray;
        if (getenv("PT_LENS_RADIUS")) o.lens_radius = (float)atof(getenv("PT_LENS_RADIUS"));
        if (getenv("PT_FOCAL_DISTANCE")) o.focal_distance = (float)atof(getenv("PT_FOCAL_DISTANCE"));
        o.seed = (unsigned)env_int("PT_SEED", 0);
        check(pt_multi_set_options(g.ctx, &o), "pt_set_options");
        if (g.ndev > 1) check(pt_multi_set_strips(g.ctx, env_int("PT_STRIP_ROWS", 8)), "pt_multi_set_strips");
        if (!g.configured) {
            g.readback_every = env_int("PT_READBACK_EVERY", 1);
            g.defer = env_int("PT_SHIM_BATCH", 1);
        }
        if (g.defer < 1) g.defer = 1;
        if (g.readback_every < 0) g.readback_every = 0;
        if (g.motion_slices < 0) {
            g.motion_slices = env_int("PT_MOTION_SLICES", 0);
            g.motion_rotat = env_int("PT_ROTAT_UNITS", 0) ? 1 : 0;
        }
    }

    // package geometry (ref: src/raytraceKernel.cu:123-134).  geom::frames is never initialised by the
    // reference's loader, so it is not read here either.
    std::vector<pt_static_geom> list((size_t)numberOfGeoms);
    for (int i = 0; i < numberOfGeoms; i++) {
        pt_static_geom &s = list[(size_t)i];
        memset(&s, 0, sizeof s);
        s.type = (int)geoms[i].type;
        s.materialid = geoms[i].materialid;
        memcpy(&s.translation, &geoms[i].translations[frame], sizeof s.translation);
        memcpy(&s.rotation, &geoms[i].rotations[frame], sizeof s.rotation);
        memcpy(&s.scale, &geoms[i].scales[frame], sizeof s.scale);
        memcpy(&s.transform, &geoms[i].transforms[frame], sizeof s.transform);
        memcpy(&s.inverseTransform, &geoms[i].inverseTransforms[frame], sizeof s.inverseTransform);
    }
    std::vector<pt_material> mats((size_t)numberOfMaterials);
    static_assert(sizeof(pt_material) == sizeof(material), "material layout");
    if (numberOfMaterials > 0) memcpy(mats.data(), materials, (size_t)numberOfMaterials * sizeof(pt_material));

    const bool scene_changed =
        list.size() != g.geoms.size() || mats.size() != g.mats.size() ||
        (!list.empty() && memcmp(list.data(), g.geoms.data(), list.size() * sizeof(pt_static_geom)) != 0) ||
        (!mats.empty() && memcmp(mats.data(), g.mats.data(), mats.size() * sizeof(pt_material)) != 0);
    if (scene_changed) {
        flush_pending();                       // pending iterations belong to the previous scene
        check(pt_multi_set_scene(g.ctx, list.data(), numberOfGeoms, mats.data(), numberOfMaterials), "pt_set_scene");
        g.geoms = list;
        g.mats = mats;
        g.meshes_dirty = !g.meshes.empty();    // pt_set_scene drops the triangles of the previous scene
    }
    if (g.meshes_dirty) {
        flush_pending();
        check(pt_multi_set_meshes(g.ctx, g.meshes.data(), (int)g.meshes.size()), "pt_set_meshes");
        g.meshes_dirty = false;
    }

    // package camera (ref: src/raytraceKernel.cu:141-146)
    pt_camera_data cam;
    memset(&cam, 0, sizeof cam);
    cam.resolution = {renderCam->resolution.x, renderCam->resolution.y};
    memcpy(&cam.position, &renderCam->positions[frame], sizeof cam.position);
    memcpy(&cam.view, &renderCam->views[frame], sizeof cam.view);
    memcpy(&cam.up, &renderCam->ups[frame], sizeof cam.up);
    cam.fov = {renderCam->fov.x, renderCam->fov.y};
    if (!g.have_cam || memcmp(&cam, &g.cam, sizeof cam) != 0) {
        flush_pending();
        check(pt_multi_set_camera(g.ctx, &cam), "pt_set_camera");
        g.cam = cam;
        g.have_cam = true;
    }

    // motion blur: the shutter stays open until the next frame of the caller's arrays, when there is one
    if (scene_changed || g.motion_dirty || g.motion_frame != frame) {
        flush_pending();
        if ((g.motion_slices > 1 || (g.motion_per_ray && g.motion_slices >= 1)) && frame + 1 < renderCam->frames) {
            std::vector<pt_static_geom> next = list;
            for (int i = 0; i < numberOfGeoms; i++) {
                memcpy(&next[(size_t)i].translation, &geoms[i].translations[frame + 1], sizeof next[0].translation);
                memcpy(&next[(size_t)i].rotation, &geoms[i].rotations[frame + 1], sizeof next[0].rotation);
                memcpy(&next[(size_t)i].scale, &geoms[i].scales[frame + 1], sizeof next[0].scale);
                memcpy(&next[(size_t)i].transform, &geoms[i].transforms[frame + 1], sizeof next[0].transform);
                memcpy(&next[(size_t)i].inverseTransform, &geoms[i].inverseTransforms[frame + 1], sizeof next[0].inverseTransform);
            }
            pt_camera_data cn = cam;
            memcpy(&cn.position, &renderCam->positions[frame + 1], sizeof cn.position);
            memcpy(&cn.view, &renderCam->views[frame + 1], sizeof cn.view);
            memcpy(&cn.up, &renderCam->ups[frame + 1], sizeof cn.up);
            check(pt_multi_set_motion(g.ctx, next.data(), &cn, g.motion_slices, g.motion_rotat), "pt_set_motion");
        } else {
            check(pt_multi_set_motion(g.ctx, nullptr, nullptr, 0, 0), "pt_set_motion");
        }
        g.motion_dirty = false;
        g.motion_frame = frame;
    }

    // one iteration.  iterations == 1 restarts the running mean (the old image is not read); a later
    // iteration on a context without history first takes the caller's image (ref: src/raytraceKernel.cu:120).
    float *host_image = reinterpret_cast<float *>(renderCam->image);
    g.last_image = host_image;
    const bool last = (unsigned)iterations >= renderCam->iterations;
    const bool readback = host_image && (last || (g.readback_every > 0 && iterations % g.readback_every == 0));
    if (iterations != g.last_iteration + 1) {
        flush_pending();                       // not the continuation of what is pending
        if (iterations > 1 && host_image) check(pt_multi_upload_image(g.ctx, host_image), "pt_upload_image");
    }
    if (g.pend_count == 0) g.pend_first = iterations;
    g.pend_count++;
    g.last_iteration = iterations;
    if (PBOpos || readback || g.pend_count >= g.defer) {
        flush_pending();
        if (PBOpos) {
            if (g.ndev != 1) { fprintf(stderr, "Cuda error: PBO output needs a single device (PT_DEVICES): %d given.\n", g.ndev); exit(EXIT_FAILURE); }
            check(pt_multi_send_image_to_pbo(g.ctx, reinterpret_cast<pt_uchar4 *>(PBOpos)), "pt_send_image_to_pbo");
        }
        if (readback) check(pt_multi_download_image(g.ctx, host_image), "pt_download_image");
        // make certain the kernels have completed (ref: src/raytraceKernel.cu:162-164)
        check(pt_multi_synchronize(g.ctx), "Kernel failed!");
    }
}
