"""ctypes binding of oracle/liboracle.so -- the CPU oracle (test infrastructure only).

Imported only by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "liboracle.so")


class Vec2(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float)]


class Vec3(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float)]

    def tup(self):
        return (self.x, self.y, self.z)


class Vec4(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float), ("w", C.c_float)]


class Mat4(C.Structure):
    _fields_ = [("x", Vec4), ("y", Vec4), ("z", Vec4), ("w", Vec4)]

    def rows(self):
        return [[r.x, r.y, r.z, r.w] for r in (self.x, self.y, self.z, self.w)]


class Ray(C.Structure):
    _fields_ = [("origin", Vec3), ("direction", Vec3)]


class StaticGeom(C.Structure):
    _fields_ = [("type", C.c_int), ("materialid", C.c_int), ("translation", Vec3), ("rotation", Vec3),
                ("scale", Vec3), ("transform", Mat4), ("inverseTransform", Mat4)]


class Material(C.Structure):
    _fields_ = [("color", Vec3), ("specularExponent", C.c_float), ("specularColor", Vec3),
                ("hasReflective", C.c_float), ("hasRefractive", C.c_float), ("indexOfRefraction", C.c_float),
                ("hasScatter", C.c_float), ("absorptionCoefficient", Vec3),
                ("reducedScatterCoefficient", C.c_float), ("emittance", C.c_float)]


class CameraData(C.Structure):
    _fields_ = [("resolution", Vec2), ("position", Vec3), ("view", Vec3), ("up", Vec3), ("fov", Vec2)]


class Fresnel(C.Structure):
    _fields_ = [("reflectionCoefficient", C.c_float), ("transmissionCoefficient", C.c_float)]


class Options(C.Structure):
    _fields_ = [("depth", C.c_int), ("rr_start", C.c_int), ("seed", C.c_uint), ("trig_mode", C.c_int),
                ("direct_light", C.c_int), ("absorption", C.c_int), ("lens_radius", C.c_float), ("focal_distance", C.c_float),
                ("scatter", C.c_int)]


class ScatterProps(C.Structure):       # AbsorptionAndScatteringProperties, ref: src/interactions.h:16-19
    _fields_ = [("absorptionCoefficient", Vec3), ("reducedScatteringCoefficient", C.c_float)]


class OMesh(C.Structure):           # o_mesh
    _fields_ = [("geom", C.c_int), ("n_triangles", C.c_int), ("vertices", C.POINTER(C.c_float))]


class Extras(C.Structure):          # o_extras
    _fields_ = [("n_meshes", C.c_int), ("meshes", C.POINTER(OMesh)), ("n_slices", C.c_int),
                ("slice_geoms", C.POINTER(StaticGeom)), ("slice_cams", C.POINTER(CameraData)),
                ("n_knots", C.c_int), ("knot_geoms", C.POINTER(StaticGeom)), ("knot_cams", C.POINTER(CameraData))]


class Scene(C.Structure):
    _fields_ = [("n_objects", C.c_int), ("n_materials", C.c_int), ("n_frames_camera", C.c_int),
                ("objects", C.POINTER(StaticGeom)), ("materials", C.POINTER(Material)),
                ("camera", CameraData), ("iterations", C.c_uint), ("image_name", C.c_char * 256)]


SPHERE, CUBE, MESH = 0, 1, 2
TRIG_POLY, TRIG_LIBM = 0, 1
ROTAT_RADIANS, ROTAT_DEGREES = 0, 1

_lib = None


def build():
    subprocess.run(["make", "-C", ORACLE_DIR, "liboracle.so"], check=True, capture_output=True)


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        build()
    L = C.CDLL(LIB_PATH)
    u, f, i = C.c_uint, C.c_float, C.c_int
    P = C.POINTER
    sig = {
        "o_hash": (u, [u]),
        "o_minstd_seed": (None, [P(u), u]),
        "o_minstd_next": (u, [P(u)]),
        "o_uniform_real": (f, [P(u), f, f]),
        "o_u01": (f, [P(u)]),
        "o_stream_seed": (u, [u, u, u, u]),
        "o_epsilonCheck": (i, [f, f]),
        "o_getPointOnRay": (Vec3, [Ray, f]),
        "o_multiplyMV": (Vec3, [Mat4, Vec4]),
        "o_getInverseDirectionOfRay": (Vec3, [Ray]),
        "o_getSignOfRay": (Vec3, [Ray]),
        "o_boxIntersectionTest": (f, [P(StaticGeom), Ray, P(Vec3), P(Vec3)]),
        "o_sphereIntersectionTest": (f, [P(StaticGeom), Ray, P(Vec3), P(Vec3)]),
        "o_getRadiuses": (Vec3, [P(StaticGeom)]),
        "o_getRandomPointOnCube": (Vec3, [P(StaticGeom), f]),
        "o_getRandomPointOnSphere": (Vec3, [P(StaticGeom), f]),
        "o_sincos_poly": (None, [f, P(f), P(f)]),
        "o_calculateRandomDirectionInHemisphere": (Vec3, [Vec3, f, f, i]),
        "o_getRandomDirectionInSphere": (Vec3, [f, f, i]),
        "o_calculateReflectionDirection": (Vec3, [Vec3, Vec3]),
        "o_calculateTransmissionDirection": (Vec3, [Vec3, Vec3, f, f]),
        "o_calculateFresnel": (Fresnel, [Vec3, Vec3, f, f, Vec3, Vec3]),
        "o_calculateBSDF": (i, [P(Ray), Vec3, Vec3, P(Vec3), P(Material), f, f, f, i]),
        "o_generateRandomNumberFromThread": (Vec3, [Vec2, f, i, i]),
        "o_raycastFromCameraKernel": (Ray, [Vec2, f, i, i, Vec3, Vec3, Vec3, Vec2, u]),
        "o_sendImageToPBO": (None, [C.c_void_p, i, C.c_void_p]),
        "o_clearImage": (None, [C.c_void_p, i]),
        "o_render": (i, [P(StaticGeom), i, P(Material), i, P(CameraData), P(Options), C.c_void_p, i, i,
                         C.c_void_p, i]),
        "o_render_counted": (i, [P(StaticGeom), i, P(Material), i, P(CameraData), P(Options), C.c_void_p, i, i,
                                 C.c_void_p, P(C.c_ulonglong), i]),
        "o_lightArea": (f, [P(StaticGeom)]),
        "o_exp_poly": (f, [f]),
        "o_calculateTransmission": (Vec3, [Vec3, f]),
        "o_sampleLight": (None, [P(StaticGeom), f, P(Vec3), P(Vec3)]),
        "o_log_poly": (f, [f]),
        "o_triangleIntersectionTest": (f, [Vec3, Vec3, Vec3, Vec3, Ray, P(Vec3), P(Vec3)]),
        "o_triangleToWorld": (None, [P(StaticGeom), P(f), P(f)]),
        "o_render_ex": (i, [P(StaticGeom), i, P(Material), i, P(CameraData), P(Options), P(Extras), C.c_void_p, i, i,
                            C.c_void_p, P(C.c_ulonglong), i]),
        "o_trace_path_ex": (Vec3, [P(StaticGeom), i, P(Material), i, P(CameraData), P(Options), P(Extras), i, i, u, P(i)]),
        "o_load_obj": (i, [C.c_char_p, P(P(f)), P(i)]),
        "o_triangleArea": (f, [Vec3, Vec3]),
        "o_sampleTriangle": (Vec3, [Vec3, Vec3, Vec3, f, f]),
        "o_interpolateGeom": (StaticGeom, [P(StaticGeom), P(StaticGeom), f, i]),
        "o_interpolateCamera": (CameraData, [P(CameraData), P(CameraData), f]),
        "o_sliceTime": (f, [i, i]),
        "o_knotTime": (f, [i, i]),
        "o_affineInverse": (None, [P(Mat4), P(Mat4)]),
        "o_free_obj": (None, [P(f)]),
        "o_calculateScatterAndAbsorption": (i, [P(Ray), P(f), P(ScatterProps), P(Vec3), P(Material), f, f, f]),
        "o_trace_path": (Vec3, [P(StaticGeom), i, P(Material), i, P(CameraData), P(Options), i, i, u, P(i)]),
        "o_buildTransformationMatrix": (Mat4, [Vec3, Vec3, Vec3, i, P(Mat4)]),
        "o_camera_fov": (Vec2, [f, Vec2]),
        "o_scene_load": (i, [C.c_char_p, i, P(Scene)]),
        "o_scene_load_frame": (i, [C.c_char_p, i, i, P(Scene)]),
        "o_scene_free": (None, [P(Scene)]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


# ---------------------------------------------------------------- helpers
def v3(x, y=None, z=None):
    if y is None:
        x, y, z = x
    return Vec3(float(x), float(y), float(z))


def f32(x):
    return float(np.float32(x))


def bits(x):
    return int(np.float32(x).view(np.uint32))


def from_bits(u):
    return float(np.uint32(u).view(np.float32))


def make_geom(gtype, materialid, t, r, s, rotat_units=ROTAT_RADIANS):
    g = StaticGeom()
    g.type, g.materialid = gtype, materialid
    g.translation, g.rotation, g.scale = v3(t), v3(r), v3(s)
    inv = Mat4()
    g.transform = lib().o_buildTransformationMatrix(g.translation, g.rotation, g.scale, rotat_units, C.byref(inv))
    g.inverseTransform = inv
    return g


def make_material(color=(1, 1, 1), spec=(1, 1, 1), refl=0.0, refr=0.0, ior=0.0, emittance=0.0, scatter=0.0,
                  absorption=(0, 0, 0), rsct=0.0):
    m = Material()
    m.color, m.specularColor = v3(color), v3(spec)
    m.hasReflective, m.hasRefractive, m.indexOfRefraction, m.emittance = refl, refr, ior, emittance
    m.hasScatter, m.absorptionCoefficient, m.reducedScatterCoefficient = scatter, v3(absorption), rsct
    return m


def make_camera(w, h, eye, view, up, fovy):
    """fov per ref: src/scene.cpp:204-207 (same arithmetic as the loaders)."""
    cam = CameraData()
    cam.resolution = Vec2(float(w), float(h))
    cam.position, cam.view, cam.up = v3(eye), v3(view), v3(up)
    cam.fov = lib().o_camera_fov(float(fovy), cam.resolution)
    return cam


class LoadedScene:
    """Scene loaded by the oracle's loader, as ctypes arrays."""

    def __init__(self, path, rotat_units=ROTAT_RADIANS, frame=0):
        s = Scene()
        rc = lib().o_scene_load_frame(path.encode(), rotat_units, frame, C.byref(s))
        if rc != 0:
            raise IOError(f"oracle loader failed on {path}: {rc}")
        self.n_objects, self.n_materials = s.n_objects, s.n_materials
        self.geoms = (StaticGeom * max(1, s.n_objects))()
        self.mats = (Material * max(1, s.n_materials))()
        for k in range(s.n_objects):
            self.geoms[k] = s.objects[k]
        for k in range(s.n_materials):
            self.mats[k] = s.materials[k]
        self.camera = CameraData.from_buffer_copy(s.camera)
        self.iterations = s.iterations
        self.image_name = s.image_name.decode()
        self.n_frames_camera = s.n_frames_camera
        lib().o_scene_free(C.byref(s))
        # MESH objects: the oracle's own OBJ reader on the file the object names (relative to the scene file);
        # {object index: float32 [n, 9]} in object space
        self.meshes = {}
        blocks = open(path).read().replace("\r\n", "\n").replace("\r", "\n").split("\n")
        for k, ln in enumerate(blocks):
            tok = ln.split()
            if len(tok) == 2 and tok[0] == "OBJECT" and k + 1 < len(blocks) and blocks[k + 1].strip().endswith(".obj"):
                vp, nt = C.POINTER(C.c_float)(), C.c_int()
                name = blocks[k + 1].strip()
                for cand in (os.path.join(os.path.dirname(path), name), name):
                    if lib().o_load_obj(cand.encode(), C.byref(vp), C.byref(nt)) == 0:
                        if nt.value > 0:
                            self.meshes[int(tok[1])] = np.ctypeslib.as_array(vp, shape=(nt.value, 9)).astype(np.float32).copy()
                        lib().o_free_obj(vp)
                        break

    def set_resolution(self, w, h):
        """RES override; fov.x recomputed exactly as the loader does (ref: src/scene.cpp:204-207)."""
        c = make_camera(w, h, self.camera.position.tup(), self.camera.view.tup(), self.camera.up.tup(),
                        self.camera.fov.y)
        self.camera = c


def make_extras(meshes=None, slice_geoms=None, slice_cams=None, nG=0, knot_geoms=None, knot_cams=None):
    """o_extras from {geom index: [n, 9] float32 array} and / or motion slices (a list of StaticGeom arrays of nG entries
    each, optionally a list of CameraData).  Returns (Extras, keep-alive list)."""
    ex, keep = Extras(), []
    if meshes:
        arrs = [(int(g), np.ascontiguousarray(v, dtype=np.float32).reshape(-1, 9)) for g, v in sorted(meshes.items())]
        desc = (OMesh * len(arrs))()
        for k, (g, v) in enumerate(arrs):
            desc[k] = OMesh(g, v.shape[0], v.ctypes.data_as(C.POINTER(C.c_float)))
        ex.n_meshes, ex.meshes = len(arrs), desc
        keep += [arrs, desc]
    if slice_geoms:
        flat = (StaticGeom * (len(slice_geoms) * nG))()
        for k, sg in enumerate(slice_geoms):
            for j in range(nG):
                flat[k * nG + j] = sg[j]
        ex.n_slices, ex.slice_geoms = len(slice_geoms), flat
        keep.append(flat)
        if slice_cams:
            cams = (CameraData * len(slice_cams))(*slice_cams)
            ex.slice_cams = cams
            keep.append(cams)
    if knot_geoms:                                  # per-ray motion blur: scene states at shutter times k / (n - 1)
        flat = (StaticGeom * (len(knot_geoms) * nG))()
        for k, sg in enumerate(knot_geoms):
            for j in range(nG):
                flat[k * nG + j] = sg[j]
        ex.n_knots, ex.knot_geoms = len(knot_geoms), flat
        keep.append(flat)
        if knot_cams:
            cams = (CameraData * len(knot_cams))(*knot_cams)
            ex.knot_cams = cams
            keep.append(cams)
    return ex, keep


def motion_slices(geoms_a, geoms_b, nG, cam_a, cam_b, slices, rotat_units=ROTAT_RADIANS):
    """The `slices` scene states of a shutter interval from frame a to frame b: (list of StaticGeom arrays, list of
    cameras), interpolated by the oracle's o_interpolateGeom / o_interpolateCamera at o_sliceTime(k, slices)."""
    L = lib()
    sg, sc = [], []
    for k in range(slices):
        t = L.o_sliceTime(k, slices)
        arr = (StaticGeom * max(1, nG))()
        for j in range(nG):
            arr[j] = L.o_interpolateGeom(C.byref(geoms_a[j]), C.byref(geoms_b[j]), t, rotat_units)
        sg.append(arr)
        sc.append(L.o_interpolateCamera(C.byref(cam_a), C.byref(cam_b), t) if cam_b is not None else CameraData.from_buffer_copy(cam_a))
    return sg, sc


def motion_knots(geoms_a, geoms_b, nG, cam_a, cam_b, segments, rotat_units=ROTAT_RADIANS):
    """The segments + 1 knot states of per-ray motion blur from frame a to frame b (shutter times k / segments):
    (list of StaticGeom arrays, list of cameras or None), by the oracle's o_interpolateGeom / o_interpolateCamera."""
    L = lib()
    kg, kc = [], []
    for k in range(segments + 1):
        t = L.o_knotTime(k, segments + 1)
        arr = (StaticGeom * max(1, nG))()
        for j in range(nG):
            arr[j] = L.o_interpolateGeom(C.byref(geoms_a[j]), C.byref(geoms_b[j]), t, rotat_units)
        kg.append(arr)
        if cam_b is not None:
            kc.append(L.o_interpolateCamera(C.byref(cam_a), C.byref(cam_b), t))
    return kg, (kc if cam_b is not None else None)


def render(geoms, nG, mats, nM, cam, depth, iters=1, iter_first=1, rr_start=-1, seed=0, trig=TRIG_POLY,
           image=None, nthreads=None, direct_light=0, shadow_out=None, absorption=0, lens_radius=0.0, focal_distance=1.0,
           scatter=0, meshes=None, slice_geoms=None, slice_cams=None, knot_geoms=None, knot_cams=None):
    """Returns (image[H,W,3] float32, live_in[depth] uint64); shadow_out (a list) receives the shadow-ray count."""
    W, H = int(cam.resolution.x), int(cam.resolution.y)
    if image is None:
        image = np.zeros((H, W, 3), dtype=np.float32)
    else:
        image = np.ascontiguousarray(image, dtype=np.float32).copy()
    live = np.zeros(depth, dtype=np.uint64)
    opt = Options(depth, rr_start, seed, trig, direct_light, absorption, lens_radius, focal_distance, scatter)
    if nthreads is None:
        nthreads = os.cpu_count() or 1
    shadow = C.c_ulonglong(0)
    ex, keep = make_extras(meshes, slice_geoms, slice_cams, nG, knot_geoms, knot_cams)
    rc = lib().o_render_ex(geoms, nG, mats, nM, C.byref(cam), C.byref(opt), C.byref(ex), image.ctypes.data, iter_first, iters,
                           live.ctypes.data, C.byref(shadow), nthreads)
    del keep
    if rc != 0:
        raise RuntimeError(f"o_render failed: {rc}")
    if shadow_out is not None:
        shadow_out.append(int(shadow.value))
    return image, live
