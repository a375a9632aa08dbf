"""project3-pathtracer_amd -- MI355X-native renderer behind the reference's cudaRaytraceCore() boundary.

The product is the C-ABI shared library ``lib/libptamd.so`` (include/pt_abi.h): hand-written HIP kernels
for gfx950 plus a C++ host.  This module is only the ctypes plumbing used by tests/ and bench.py:
it loads the library (and FAILS LOUDLY if it is missing -- there is no CPU or PyTorch fallback),
declares every entry point, and wraps a context in a small ``Renderer`` class whose methods map 1:1 onto
the C functions.  PyTorch, when present, is used for device buffers, streams and torch.distributed only.

The directory name contains a hyphen (it is fixed by the project layout), so import it with
``__graft_entry__.load_package()`` / ``tests/ptamd.py`` rather than a plain ``import``.
"""
import ctypes as C
import os
import subprocess

import numpy as np

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC_DIR = os.path.join(PKG_DIR, "csrc")
LIB_DIR = os.path.join(PKG_DIR, "lib")
LIB_PATH = os.environ.get("PT_LIBPTAMD") or os.path.join(LIB_DIR, "libptamd.so")   # PT_LIBPTAMD: a diagnostic build (csrc/Makefile OUT=...)
HEADLESS_PATH = os.path.join(LIB_DIR, "pt_headless")
ROOT = os.path.dirname(PKG_DIR)
HEADER_PATH = os.path.join(ROOT, "include", "pt_abi.h")

PT_MAX_DEPTH = 64
PT_ABI_VERSION = 2          # include/pt_abi.h; checked against the loaded library in lib()
PT_OK = 0
SPHERE, CUBE, MESH = 0, 1, 2
ROTAT_RADIANS, ROTAT_DEGREES = 0, 1


class Vec2(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float)]


class Vec3(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float)]


class Vec4(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float), ("w", C.c_float)]


class Mat4(C.Structure):
    _fields_ = [("x", Vec4), ("y", Vec4), ("z", Vec4), ("w", Vec4)]


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


class Options(C.Structure):
    _fields_ = [("depth", C.c_int), ("rr_start", C.c_int), ("seed", C.c_uint), ("compaction", C.c_int),
                ("workgroup", C.c_int), ("geom_path", C.c_int), ("row_begin", C.c_int), ("row_end", C.c_int),
                ("use_graph", C.c_int), ("batch", C.c_int), ("direct_light", C.c_int), ("absorption", C.c_int), ("strip_rows", C.c_int), ("strip_world", C.c_int),
                ("strip_rank", C.c_int), ("scatter", C.c_int), ("lens_radius", C.c_float), ("focal_distance", C.c_float),
                ("sequences", C.c_int), ("motion_per_ray", C.c_int), ("resident", C.c_int)]


class Stats(C.Structure):
    _fields_ = [("iterations", C.c_ulonglong), ("ray_bounces", C.c_ulonglong),
                ("live_in", C.c_ulonglong * PT_MAX_DEPTH), ("gpu_ms", C.c_double),
                ("bounce_launches", C.c_ulonglong), ("shadow_rays", C.c_ulonglong)]


class LaunchInfo(C.Structure):      # pt_launch_info
    _fields_ = [("geom_path", C.c_int), ("workgroup", C.c_int), ("grid", C.c_int), ("batch", C.c_int), ("sequences", C.c_int),
                ("resident", C.c_int), ("refill_min", C.c_int), ("launches_per_batch", C.c_int), ("lds_bytes", C.c_int),
                ("slab_pretest", C.c_int), ("reserved", C.c_int * 6)]


class Mesh(C.Structure):            # pt_mesh: triangles of one MESH geom, object space, 9 floats each
    _fields_ = [("geom", C.c_int), ("n_triangles", C.c_int), ("vertices", C.POINTER(C.c_float))]


class PtError(RuntimeError):
    pass


_lib = None


def build(verbose=False):
    """Compile libptamd.so and pt_headless for gfx950 with hipcc (cross-compiles without a GPU)."""
    res = subprocess.run(["make", "-C", CSRC_DIR, "-j8"], capture_output=True, text=True)
    if verbose or res.returncode != 0:
        print(res.stdout)
        print(res.stderr)
    if res.returncode != 0:
        raise PtError("building libptamd.so failed")


def lib():
    """Load libptamd.so.  Raises if it is not built: the hot path has no fallback implementation."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise PtError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                      "(hipcc, gfx950). There is no CPU/PyTorch fallback for the render path.")
    try:  # PyTorch bundles its own libamdhip64: load it first so both share one HIP runtime
        import torch  # noqa: F401
    except Exception:
        pass
    L = C.CDLL(LIB_PATH)
    i, u, vp, cp, sz = C.c_int, C.c_uint, C.c_void_p, C.c_char_p, C.c_size_t
    P = C.POINTER
    sig = {
        "pt_device_count": (i, []),
        "pt_strip_local_rows": (i, [i, i, i, i]),
        "pt_strip_global_row": (i, [i, i, i, i]),
        "pt_create": (i, [i, P(vp)]),
        "pt_destroy": (None, [vp]),
        "pt_last_error": (cp, []),
        "pt_version": (cp, []),
        "pt_get_launch_info": (i, [vp, P(LaunchInfo)]),
        "pt_abi_version": (i, []),
        "pt_options_size": (sz, []),
        "pt_default_options": (None, [P(Options)]),
        "pt_set_options": (i, [vp, P(Options)]),
        "pt_get_options": (i, [vp, P(Options)]),
        "pt_set_scene": (i, [vp, P(StaticGeom), i, P(Material), i]),
        "pt_set_camera": (i, [vp, P(CameraData)]),
        "pt_set_meshes": (i, [vp, P(Mesh), i]),
        "pt_set_motion": (i, [vp, P(StaticGeom), P(CameraData), i, i]),
        "pt_multi_set_motion": (i, [vp, P(StaticGeom), P(CameraData), i, i]),
        "pt_multi_set_meshes": (i, [vp, P(Mesh), i]),
        "pt_scene_mesh": (i, [vp, i, P(P(C.c_float)), P(i)]),
        "pt_set_stream": (i, [vp, vp]),
        "pt_image_bytes": (sz, [vp]),
        "pt_bind_image": (i, [vp, vp]),
        "pt_clear_image": (i, [vp]),
        "pt_image_device_pointer": (i, [vp, P(vp)]),
        "pt_record_event": (i, [vp, vp]),
        "pt_multi_gather_to_device_async": (i, [vp, vp, i]),
        "pt_multi_gather_times": (i, [vp, P(C.c_double), P(C.c_double)]),
        "pt_multi_peer_access": (i, [vp, i, i, P(i)]),
        "pt_multi_create": (i, [P(i), i, P(vp)]),
        "pt_multi_destroy": (None, [vp]),
        "pt_multi_count": (i, [vp]),
        "pt_multi_set_options": (i, [vp, P(Options)]),
        "pt_multi_set_scene": (i, [vp, P(StaticGeom), i, P(Material), i]),
        "pt_multi_set_camera": (i, [vp, P(CameraData)]),
        "pt_multi_band": (i, [vp, i, P(i), P(i)]),
        "pt_multi_clear_image": (i, [vp]),
        "pt_multi_upload_image": (i, [vp, vp]),
        "pt_multi_render": (i, [vp, i, i]),
        "pt_multi_synchronize": (i, [vp]),
        "pt_multi_download_image": (i, [vp, vp]),
        "pt_multi_gather_to_device": (i, [vp, vp, i]),
        "pt_multi_set_strips": (i, [vp, i]),
        "pt_multi_send_image_to_pbo": (i, [vp, vp]),
        "pt_multi_get_stats": (i, [vp, P(Stats)]),
        "pt_upload_image": (i, [vp, vp]),
        "pt_download_image": (i, [vp, vp]),
        "pt_render": (i, [vp, i, i]),
        "pt_send_image_to_pbo": (i, [vp, vp]),
        "pt_synchronize": (i, [vp]),
        "pt_render_iteration": (i, [vp, vp, vp, i]),
        "pt_get_stats": (i, [vp, P(Stats)]),
        "pt_reset_stats": (i, [vp]),
        "pt_render_profiled": (i, [vp, i, i, P(C.c_double)]),
        "pt_selftest_math": (i, [vp, P(C.c_ulonglong)]),
        "pt_device_kat": (i, [vp, i, vp, i, vp, i]),
        "pt_scene_load": (i, [cp, i, P(vp)]),
        "pt_scene_free": (None, [vp]),
        "pt_scene_counts": (i, [vp, P(i), P(i), P(i)]),
        "pt_scene_camera_info": (i, [vp, P(u), cp, sz]),
        "pt_scene_get_frame": (i, [vp, i, P(StaticGeom), P(Material), P(CameraData)]),
        "pt_camera_set_resolution": (i, [P(CameraData), i, i]),
        "pt_image_to_rgb8": (i, [vp, i, i, i, vp]),
        "pt_save_image_png": (i, [cp, vp, i, i, i]),
        "pt_save_image": (i, [cp, vp, i, i, i]),
        "pt_save_image_bmp": (i, [cp, vp, i, i, i]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    L._declared = sorted(sig)
    # pt_options travels by pointer without a size: refuse a library built against another layout
    if L.pt_abi_version() != PT_ABI_VERSION or L.pt_options_size() != C.sizeof(Options):
        raise PtError(f"{LIB_PATH}: ABI {L.pt_abi_version()} / pt_options of {L.pt_options_size()} B, this binding expects "
                      f"ABI {PT_ABI_VERSION} / {C.sizeof(Options)} B -- rebuild the library (make -C csrc)")
    _lib = L
    return L


def _check(rc, what):
    if rc != PT_OK:
        raise PtError(f"{what} failed ({rc}): {lib().pt_last_error().decode()}")


class SceneFile:
    """Scene file parsed by the library's loader (ref: src/scene.cpp), flattened at one frame."""

    def __init__(self, path, rotat_units=ROTAT_RADIANS, frame=0):
        L = lib()
        h = C.c_void_p()
        rc = L.pt_scene_load(path.encode(), rotat_units, C.byref(h))
        if rc != PT_OK:
            raise PtError(f"pt_scene_load({path}) failed ({rc})")
        try:
            no, nm, nf = C.c_int(), C.c_int(), C.c_int()
            L.pt_scene_counts(h, C.byref(no), C.byref(nm), C.byref(nf))
            self.n_objects, self.n_materials, self.n_camera_frames = no.value, nm.value, nf.value
            self.geoms = (StaticGeom * max(1, no.value))()
            self.mats = (Material * max(1, nm.value))()
            self.camera = CameraData()
            _check(L.pt_scene_get_frame(h, frame, self.geoms, self.mats, C.byref(self.camera)), "pt_scene_get_frame")
            it = C.c_uint()
            name = C.create_string_buffer(256)
            L.pt_scene_camera_info(h, C.byref(it), name, 256)
            self.iterations, self.image_name = it.value, name.value.decode()
            # triangles of the MESH objects (object space), {object index: float32 array [n, 9]}
            self.meshes = {}
            for k in range(no.value):
                vp_, nt = C.POINTER(C.c_float)(), C.c_int()
                if L.pt_scene_mesh(h, k, C.byref(vp_), C.byref(nt)) == PT_OK and nt.value > 0:
                    self.meshes[k] = np.ctypeslib.as_array(vp_, shape=(nt.value, 9)).astype(np.float32).copy()
        finally:
            L.pt_scene_free(h)

    def set_resolution(self, w, h):
        _check(lib().pt_camera_set_resolution(C.byref(self.camera), w, h), "pt_camera_set_resolution")


class Renderer:
    """One persistent device context (pt_ctx).  Methods mirror include/pt_abi.h one to one."""

    def __init__(self, device=0):
        self.L = lib()
        self.h = C.c_void_p()
        _check(self.L.pt_create(device, C.byref(self.h)), "pt_create")
        self.opt = Options()
        self.L.pt_default_options(C.byref(self.opt))
        self._keep = []

    def close(self):
        if self.h:
            self.L.pt_destroy(self.h)
            self.h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def set_options(self, **kw):
        for k, v in kw.items():
            if not hasattr(self.opt, k):
                raise AttributeError(k)
            setattr(self.opt, k, v)
        _check(self.L.pt_set_options(self.h, C.byref(self.opt)), "pt_set_options")

    def set_scene(self, geoms, n_geoms, mats, n_mats):
        _check(self.L.pt_set_scene(self.h, C.cast(geoms, C.POINTER(StaticGeom)), n_geoms,
                                   C.cast(mats, C.POINTER(Material)), n_mats), "pt_set_scene")

    def set_meshes(self, meshes):
        """meshes: {geom index: float32 array [n_triangles, 9]} (object space), e.g. SceneFile.meshes"""
        arrs = [(int(g), np.ascontiguousarray(v, dtype=np.float32).reshape(-1, 9)) for g, v in sorted(meshes.items())]
        desc = (Mesh * max(1, len(arrs)))()
        for k, (g, v) in enumerate(arrs):
            desc[k] = Mesh(g, v.shape[0], v.ctypes.data_as(C.POINTER(C.c_float)))
        _check(self.L.pt_set_meshes(self.h, desc, len(arrs)), "pt_set_meshes")

    def set_motion(self, geoms_next, cam_next, slices, rotat_units=ROTAT_RADIANS):
        """Motion blur from the frame of set_scene / set_camera to the next frame (pt_set_motion); cam_next may be None."""
        cn = C.byref(CameraData.from_buffer_copy(cam_next)) if cam_next is not None else None
        gn = C.cast(geoms_next, C.POINTER(StaticGeom)) if geoms_next is not None else None
        _check(self.L.pt_set_motion(self.h, gn, cn, slices, rotat_units), "pt_set_motion")

    def set_camera(self, cam):
        self.cam = CameraData.from_buffer_copy(cam)
        _check(self.L.pt_set_camera(self.h, C.cast(C.byref(self.cam), C.POINTER(CameraData))), "pt_set_camera")

    def set_stream(self, stream_ptr):
        _check(self.L.pt_set_stream(self.h, stream_ptr), "pt_set_stream")

    def image_bytes(self):
        return self.L.pt_image_bytes(self.h)

    def tile_shape(self):
        W = int(self.cam.resolution.x)
        return (self.image_bytes() // (12 * W), W, 3)

    def bind_image(self, device_ptr):
        _check(self.L.pt_bind_image(self.h, device_ptr), "pt_bind_image")

    def clear_image(self):
        _check(self.L.pt_clear_image(self.h), "pt_clear_image")

    def upload_image(self, arr):
        a = np.ascontiguousarray(arr, dtype=np.float32)
        assert a.nbytes == self.image_bytes()
        _check(self.L.pt_upload_image(self.h, a.ctypes.data), "pt_upload_image")

    def download_image(self):
        out = np.empty(self.tile_shape(), dtype=np.float32)
        _check(self.L.pt_download_image(self.h, out.ctypes.data), "pt_download_image")
        return out

    def render(self, iter_first, iter_count):
        _check(self.L.pt_render(self.h, iter_first, iter_count), "pt_render")

    def send_image_to_pbo(self, device_ptr):
        _check(self.L.pt_send_image_to_pbo(self.h, device_ptr), "pt_send_image_to_pbo")

    def synchronize(self):
        _check(self.L.pt_synchronize(self.h), "pt_synchronize")

    def render_iteration(self, iteration, host_image=None, pbo_ptr=None):
        hp = host_image.ctypes.data if host_image is not None else None
        _check(self.L.pt_render_iteration(self.h, pbo_ptr, hp, iteration), "pt_render_iteration")

    def stats(self):
        s = Stats()
        _check(self.L.pt_get_stats(self.h, C.byref(s)), "pt_get_stats")
        return s

    def launch_info(self):
        """The launch shape the library chose (geometry path, workgroup, grid, batch, sequences, resident paths)."""
        li = LaunchInfo()
        _check(self.L.pt_get_launch_info(self.h, C.byref(li)), "pt_get_launch_info")
        return li

    def render_profiled(self, iter_first, iter_count):
        """Per-bounce kernel time (ms, summed over the iterations), one HIP event pair per launch."""
        ms = (C.c_double * PT_MAX_DEPTH)()
        _check(self.L.pt_render_profiled(self.h, iter_first, iter_count, ms), "pt_render_profiled")
        return [float(ms[b]) for b in range(self.opt.depth)]

    def device_kat(self, op, inputs, n_out):
        """Evaluate one device function on the GPU (include/pt_abi.h PT_KAT_*); inputs/outputs are float32 arrays."""
        a = np.ascontiguousarray(inputs, dtype=np.float32)
        out = np.zeros(n_out, dtype=np.float32)
        _check(self.L.pt_device_kat(self.h, op, a.ctypes.data, a.size, out.ctypes.data, n_out), "pt_device_kat")
        return out

    def selftest_math(self):
        """Mismatch counts of the kernels' exact sqrt / rcp / rsqrt sequences over all 2^32 inputs (must be 0)."""
        out = (C.c_ulonglong * 3)()
        _check(self.L.pt_selftest_math(self.h, out), "pt_selftest_math")
        return [int(x) for x in out]

    def reset_stats(self):
        _check(self.L.pt_reset_stats(self.h), "pt_reset_stats")


def algorithmic_bytes(npix, live_in, iterations):
    """per iteration  P*40 (generate) + sum_b (live_in(b) + live_out(b))*40 + P*24 (pixel RMW),
    live_out(b) = live_in(b+1), live_out(last) = 0.  `live_in` is summed over `iterations` iterations."""
    live = [int(x) for x in live_in]
    total = iterations * npix * (40 + 24)
    for b, n in enumerate(live):
        nxt = live[b + 1] if b + 1 < len(live) else 0
        total += (n + nxt) * 40
    return total
