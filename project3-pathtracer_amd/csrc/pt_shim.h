// pt_shim.h -- optional controls of the reference-side binding (pt_shim.cpp).  The renderer entry point itself is
// declared by the reference's own header (ref: src/raytraceKernel.h:17); a caller that never includes this file gets
// the reference's behaviour: every cudaRaytraceCore call renders its iteration and updates renderCam->image.
#pragma once

// batch: iterations the shim may hold back and render together while nobody can observe them (no PBO, no image
//        read-back due); 1 = render on every call.
// readback_every: copy renderCam->image back every N iterations; 0 = only when iterations == renderCam->iterations.
// Overrides PT_SHIM_BATCH / PT_READBACK_EVERY.
void pt_shim_configure(int batch, int readback_every);
// Render whatever is still pending, copy the image into the renderCam->image of the last call, synchronize.
// Needed only after opting in to deferral, when the caller stops before the scene's last iteration.
void pt_shim_flush(void);
// The reference's geom struct has no room for a mesh (ref: src/sceneStructs.h:21-30; its loader reads the .obj name and
// nothing else), so triangles reach the renderer beside the call: n meshes as pt_mesh of include/pt_abi.h (copied).
// They stay attached across cudaRaytraceCore calls until replaced.
struct pt_mesh;
void pt_shim_set_meshes(const pt_mesh *meshes, int n);
// Motion blur (PT_MOTION_SLICES > 1 in the environment, or slices here): when the scene has a frame after the one being
// rendered, the shutter stays open from `frame` to `frame + 1` of the caller's per-frame arrays (pt_set_motion).
// rotat_units: the unit the scene's ROTAT values are in (0 radians = what the reference's loader assumes, 1 degrees),
// needed to rebuild the in-between matrices.
void pt_shim_set_motion(int slices, int rotat_units);
