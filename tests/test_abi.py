"""C-ABI boundary checks that need no GPU: the library loads, exports every symbol include/pt_abi.h declares,
its POD structs have the reference's layouts, and the host-only entry points (scene loader, image write-out,
option validation, error reporting) behave.  No compute call is made."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

import oracle_lib as O
from __graft_entry__ import load_package

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCENES = os.path.join(ROOT, "scenes")


@pytest.fixture(scope="module")
def pkg():
    p = load_package()
    if not os.path.exists(p.LIB_PATH):
        p.build()
    p.lib()
    return p


def declared_functions():
    text = open(os.path.join(ROOT, "include", "pt_abi.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = re.findall(r"\b(pt_[a-z0-9_]+)\s*\(", text)
    return sorted(set(names))


def test_library_exports_every_declared_symbol(pkg):
    L = pkg.lib()
    names = declared_functions()
    assert len(names) >= 30
    for n in names:
        assert hasattr(L, n), f"libptamd.so does not export {n}"
    # and the Python binding declares all of them
    assert set(names) <= set(L._declared) | {"pt_status"}, set(names) - set(L._declared)


def test_no_torch_or_cxx_types_in_abi():
    text = open(os.path.join(ROOT, "include", "pt_abi.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)      # declarations only, comments stripped
    for bad in ("torch", "at::", "std::", "hipStream_t", "template", "class ", "&"):
        assert bad not in text, bad


def test_library_does_not_link_the_oracle(pkg):
    """The product must not route through oracle/: libptamd.so neither needs liboracle nor defines o_* symbols."""
    out = subprocess.run(["readelf", "-d", pkg.LIB_PATH], capture_output=True, text=True).stdout
    assert "liboracle" not in out
    syms = subprocess.run(["nm", "-D", "--defined-only", pkg.LIB_PATH], capture_output=True, text=True).stdout
    assert not re.search(r"\bo_(render|hash|trace_path)\b", syms)
    for root, _, files in os.walk(os.path.join(ROOT, "project3-pathtracer_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")) or f == "Makefile":
                src = open(os.path.join(root, f), errors="ignore").read()
                assert "oracle/" not in src and "pt_oracle" not in src and "oracle_lib" not in src, f


def test_shim_exports_the_symbol_main_cpp_links(pkg):
    """The reference's main.cpp calls cudaRaytraceCore through its C++-mangled name (ref: src/raytraceKernel.h:17,
    src/main.cpp:110): the binding must define exactly that symbol -- Itanium mangling of
    void cudaRaytraceCore(uchar4*, camera*, int, int, material*, int, geom*, int) -- plus the two optional controls."""
    assert os.path.exists(pkg.HEADLESS_PATH)
    syms = subprocess.run(["nm", "--defined-only", pkg.HEADLESS_PATH], capture_output=True, text=True).stdout
    defined = {ln.split()[-1] for ln in syms.splitlines() if ln.split()}
    assert "_Z16cudaRaytraceCoreP6uchar4P6cameraiiP8materialiP4geomi" in defined
    assert "_Z17pt_shim_configureii" in defined and "_Z13pt_shim_flushv" in defined
    # and what it leaves undefined is the C-ABI only (no kernels, no HIP runtime in the binding itself)
    und = subprocess.run(["nm", "--undefined-only", pkg.HEADLESS_PATH], capture_output=True, text=True).stdout
    wanted = {ln.split()[-1] for ln in und.splitlines() if ln.split() and ln.split()[-1].startswith("pt_")}
    assert wanted and wanted <= set(declared_functions()), wanted - set(declared_functions())
    assert not re.search(r"\bhip[A-Z]", und)


def test_pod_layouts_match_reference(pkg, golden):
    lay = golden["reference_vectors"]["layouts"]
    for cls, key in ((pkg.StaticGeom, "staticGeom"), (pkg.Material, "material"), (pkg.CameraData, "cameraData")):
        assert C.sizeof(cls) == lay[key]["size"]
        for name, _ in cls._fields_:
            assert getattr(cls, name).offset == lay[key][name], (key, name)
    assert C.sizeof(pkg.Mat4) == lay["cudaMat4"]
    # the binding's Options/Stats mirror the header
    assert C.sizeof(pkg.Options) == 21 * 4 == pkg.lib().pt_options_size()
    assert pkg.lib().pt_abi_version() == pkg.PT_ABI_VERSION == 2
    assert C.sizeof(pkg.LaunchInfo) == 16 * 4
    assert C.sizeof(pkg.Stats) == 8 * 2 + 8 * pkg.PT_MAX_DEPTH + 8 + 8 + 8


def test_shim_struct_layouts_compile_time(pkg):
    """pt_refstructs.h static_asserts the reference's geom/camera/material layouts; building the headless driver
    compiled them."""
    assert os.path.exists(pkg.HEADLESS_PATH)


def test_default_options(pkg):
    o = pkg.Options()
    pkg.lib().pt_default_options(C.byref(o))
    assert (o.depth, o.rr_start, o.seed, o.compaction, o.use_graph) == (8, -1, 0, 1, 1)


def test_create_fails_loudly_without_gpu(pkg):
    """No CPU fallback: on a machine without a gfx950 device pt_create reports an error instead of rendering."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    h = C.c_void_p()
    rc = pkg.lib().pt_create(0, C.byref(h))
    assert rc != 0 and not h.value
    assert b"no HIP device" in pkg.lib().pt_last_error() or b"gfx950" in pkg.lib().pt_last_error()
    with pytest.raises(pkg.PtError):
        pkg.Renderer(0)


# ---------------------------------------------------------------- scene loader of the product vs oracle/reference
@pytest.mark.parametrize("scene", ["sampleScene.txt", "sampleScene_spec.txt", "cornell_glass.txt", "cloud256.txt"])
@pytest.mark.parametrize("rotat", [0, 1])
def test_product_loader_equals_oracle_loader(pkg, scene, rotat):
    a = pkg.SceneFile(os.path.join(SCENES, scene), rotat)
    b = O.LoadedScene(os.path.join(SCENES, scene), rotat)
    assert (a.n_objects, a.n_materials) == (b.n_objects, b.n_materials)
    assert bytes(a.geoms)[: a.n_objects * 172] == bytes(b.geoms)[: b.n_objects * 172]
    assert bytes(a.mats)[: a.n_materials * 64] == bytes(b.mats)[: b.n_materials * 64]
    assert bytes(a.camera) == bytes(b.camera)
    assert (a.iterations, a.image_name) == (b.iterations, b.image_name)


def test_product_loader_matches_real_glm(pkg, golden):
    """pt_scene.cpp's transform builder against real GLM 0.9.5.4 vectors, bit for bit."""
    L = pkg.lib()
    import tempfile
    for e in golden["glm_vectors"]["trs"][:20]:
        t, r, s = ([O.from_bits(u) for u in e[k]] for k in ("translation", "rotation", "scale"))
        text = ("MATERIAL 0\n" + "\n".join(f"{k} 0" if k not in ("RGB", "SPECRGB", "ABSCOEFF") else f"{k} 0 0 0" for k in
                                          ["RGB", "SPECEX", "SPECRGB", "REFL", "REFR", "REFRIOR", "SCATTER", "ABSCOEFF",
                                           "RSCTCOEFF", "EMITTANCE"]) +
                "\n\nCAMERA\nRES 8 8\nFOVY 25\nITERATIONS 1\nFILE x.bmp\nframe 0\nEYE 0 0 0\nVIEW 0 0 -1\nUP 0 1 0\n\n"
                "OBJECT 0\ncube\nmaterial 0\nframe 0\n"
                f"TRANS {t[0]!r} {t[1]!r} {t[2]!r}\nROTAT {r[0]!r} {r[1]!r} {r[2]!r}\nSCALE {s[0]!r} {s[1]!r} {s[2]!r}\n")
        with tempfile.NamedTemporaryFile("w", suffix=".txt", delete=False) as f:
            f.write(text)
        try:
            sc = pkg.SceneFile(f.name)
        finally:
            os.unlink(f.name)
        g = sc.geoms[0]
        got = np.frombuffer(bytes(g.transform), dtype=np.uint32).tolist()
        goti = np.frombuffer(bytes(g.inverseTransform), dtype=np.uint32).tolist()
        assert got == e["transform"] and goti == e["inverse"]


def test_resolution_override_recomputes_fov(pkg):
    sc = pkg.SceneFile(os.path.join(SCENES, "sampleScene.txt"))
    assert (sc.camera.fov.x, sc.camera.fov.y) == (25.0, 25.0)
    sc.set_resolution(1920, 1080)
    assert abs(sc.camera.fov.x - 39.66) < 0.01 and sc.camera.fov.y == 25.0
    o = O.LoadedScene(os.path.join(SCENES, "sampleScene.txt"))
    o.set_resolution(1920, 1080)
    assert bytes(sc.camera) == bytes(o.camera)


def test_loader_missing_file(pkg):
    with pytest.raises(pkg.PtError):
        pkg.SceneFile("/nonexistent/scene.txt")


# ---------------------------------------------------------------- image write-out (ref src/main.cpp:116-141)
def test_image_write_out_flip_and_quantisation(pkg, tmp_path):
    W, H = 5, 3
    img = np.zeros((H, W, 3), dtype=np.float32)
    img[0, 0] = (1.0, 0.5, 0.25)          # buffer x=0 -> picture x=W-1
    img[2, 4] = (2.0, -1.0, 0.999)        # clamped
    out = np.zeros((H, W, 3), dtype=np.uint8)
    assert pkg.lib().pt_image_to_rgb8(img.ctypes.data, W, H, 1, out.ctypes.data) == 0
    assert tuple(out[0, W - 1]) == (255, 127, 63)
    assert tuple(out[2, 0]) == (255, 0, 254)
    path = str(tmp_path / "t.0.bmp")
    assert pkg.lib().pt_save_image_bmp(path.encode(), img.ctypes.data, W, H, 1) == 0
    raw = open(path, "rb").read()
    assert raw[:2] == b"BM" and len(raw) == 54 + ((W * 3 + 3) // 4 * 4) * H
    # bottom-up BGR rows: last stored row is picture row 0; its last pixel is buffer (0,0)
    stride = (W * 3 + 3) // 4 * 4
    row0 = raw[54 + (H - 1) * stride: 54 + (H - 1) * stride + W * 3]
    assert tuple(row0[-3:]) == (63, 127, 255)


# ---------------------------------------------------------------- animation frames (ref src/sceneStructs.h:21-30,50-61)
@pytest.mark.parametrize("frame", [0, 1, 2])
def test_animation_frames_product_loader_equals_oracle(pkg, frame):
    path = os.path.join(SCENES, "sampleScene_anim.txt")
    a = pkg.SceneFile(path, 1, frame=frame)
    b = O.LoadedScene(path, 1, frame=frame)
    assert a.n_camera_frames == 3 and b.n_frames_camera == 3
    assert bytes(a.geoms)[: a.n_objects * 172] == bytes(b.geoms)[: b.n_objects * 172]
    assert bytes(a.camera) == bytes(b.camera)
    if frame:
        f0 = pkg.SceneFile(path, 1, frame=0)
        assert bytes(a.geoms) != bytes(f0.geoms) and bytes(a.camera) != bytes(f0.camera)


def test_animation_frame_out_of_range(pkg):
    with pytest.raises(pkg.PtError):
        pkg.SceneFile(os.path.join(SCENES, "sampleScene_anim.txt"), 0, frame=3)


def read_png_rgb8(path):
    """Minimal PNG reader (8-bit RGB, filter 0 rows): enough to check the writer against zlib and the CRCs."""
    import struct
    import zlib
    raw = open(path, "rb").read()
    assert raw[:8] == b"\x89PNG\r\n\x1a\n"
    pos, idat, W, H = 8, b"", 0, 0
    while pos < len(raw):
        n, typ = struct.unpack(">I4s", raw[pos:pos + 8])
        data = raw[pos + 8:pos + 8 + n]
        crc, = struct.unpack(">I", raw[pos + 8 + n:pos + 12 + n])
        assert zlib.crc32(typ + data) & 0xFFFFFFFF == crc, typ
        if typ == b"IHDR":
            W, H, depth, ctype, comp, filt, inter = struct.unpack(">IIBBBBB", data)
            assert (depth, ctype, comp, filt, inter) == (8, 2, 0, 0, 0)
        elif typ == b"IDAT":
            idat += data
        pos += 12 + n
    rows = np.frombuffer(zlib.decompress(idat), dtype=np.uint8).reshape(H, 1 + 3 * W)      # checks Adler-32 too
    assert (rows[:, 0] == 0).all()
    return rows[:, 1:].reshape(H, W, 3)


@pytest.mark.parametrize("w,h", [(5, 3), (200, 150)])            # the second needs several stored deflate blocks
def test_png_write_out(pkg, tmp_path, w, h):
    """PNG container (ref: src/image.cpp:86): same pixels as the BMP path -- x flip, clamp(v*255) truncation -- rows
    top-down; pt_save_image picks the container from the name like the reference (…bmp -> BMP, else PNG)."""
    rng = np.random.default_rng(3)
    img = rng.uniform(-0.2, 1.3, (h, w, 3)).astype(np.float32)
    want = np.zeros((h, w, 3), dtype=np.uint8)
    assert pkg.lib().pt_image_to_rgb8(img.ctypes.data, w, h, 1, want.ctypes.data) == 0
    path = str(tmp_path / "t.0.png")
    assert pkg.lib().pt_save_image_png(path.encode(), img.ctypes.data, w, h, 1) == 0
    assert np.array_equal(read_png_rgb8(path), want)
    p2, p3 = str(tmp_path / "a.png"), str(tmp_path / "a.bmp")
    assert pkg.lib().pt_save_image(p2.encode(), img.ctypes.data, w, h, 1) == 0
    assert pkg.lib().pt_save_image(p3.encode(), img.ctypes.data, w, h, 1) == 0
    assert open(p2, "rb").read(4) == b"\x89PNG" and open(p3, "rb").read(2) == b"BM"


def test_loaders_agree_on_random_scene_files(pkg, tmp_path):
    """The product's C++ loader and the oracle's C loader on randomly generated scene files in the reference format
    (several materials, several frames, spheres / cubes / meshes, fractional and negative numbers): identical records
    for every frame and both ROTAT readings."""
    rng = np.random.default_rng(99)
    keys = ["RGB", "SPECEX", "SPECRGB", "REFL", "REFR", "REFRIOR", "SCATTER", "ABSCOEFF", "RSCTCOEFF", "EMITTANCE"]
    for case in range(25):
        nm, no, nf = int(rng.integers(1, 6)), int(rng.integers(1, 12)), int(rng.integers(1, 4))
        lines = []
        for m in range(nm):
            lines.append(f"MATERIAL {m}")
            for k in keys:
                n = 3 if k in ("RGB", "SPECRGB", "ABSCOEFF") else 1
                lines.append(k + " " + " ".join(f"{rng.uniform(0, 3):.4f}" if rng.random() < 0.7 else str(int(rng.integers(0, 3)))
                                                for _ in range(n)))
            lines.append("")
        lines += ["CAMERA", f"RES {int(rng.integers(1, 300))} {int(rng.integers(1, 300))}", f"FOVY {rng.uniform(5, 60):.3f}",
                  f"ITERATIONS {int(rng.integers(1, 100))}", "FILE out.bmp"]
        for f in range(nf):
            lines += [f"frame {f}", "EYE " + " ".join(f"{v:.3f}" for v in rng.uniform(-9, 9, 3)),
                      "VIEW " + " ".join(f"{v:.3f}" for v in rng.uniform(-1, 1, 3)),
                      "UP " + " ".join(f"{v:.3f}" for v in rng.uniform(-1, 1, 3))]
        lines.append("")
        for o in range(no):
            kind = str(rng.choice(["sphere", "cube", f"model{o}.obj"]))       # a mesh is named by its .obj file (ref: src/scene.cpp:57-66)
            lines += [f"OBJECT {o}", kind, f"material {int(rng.integers(0, nm))}"]
            for f in range(nf):
                lines += [f"frame {f}", "TRANS " + " ".join(f"{v:.4f}" for v in rng.uniform(-6, 6, 3)),
                          "ROTAT " + " ".join(f"{v:.2f}" for v in rng.uniform(-180, 180, 3)),
                          "SCALE " + " ".join(f"{v:.4f}" for v in rng.uniform(0.01, 9, 3))]
            lines.append("")
        path = str(tmp_path / f"random{case}.txt")
        open(path, "w").write("\n".join(lines))
        for rotat in (0, 1):
            for frame in range(nf):
                a = pkg.SceneFile(path, rotat, frame=frame)
                b = O.LoadedScene(path, rotat, frame=frame)
                assert (a.n_objects, a.n_materials, a.n_camera_frames) == (b.n_objects, b.n_materials, b.n_frames_camera) == (no, nm, nf)
                assert bytes(a.geoms)[: no * 172] == bytes(b.geoms)[: no * 172], (case, rotat, frame)
                assert bytes(a.mats)[: nm * 64] == bytes(b.mats)[: nm * 64]
                assert bytes(a.camera) == bytes(b.camera)
