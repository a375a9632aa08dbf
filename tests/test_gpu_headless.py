"""End-to-end drop-in test on the GPU: the headless driver (reference main.cpp protocol) -> cudaRaytraceCore shim
-> C-ABI -> HIP kernels -> BMP, compared with the oracle's image pushed through the reference's write-out rules
(x flip, clamp(v*255, 0, 255) truncation; ref src/main.cpp:116-141, src/image.cpp:41-88)."""
import os
import struct
import subprocess

import numpy as np
import pytest

import oracle_lib as O
from __graft_entry__ import load_package

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def read_bmp(path):
    raw = open(path, "rb").read()
    assert raw[:2] == b"BM"
    w, h = struct.unpack("<ii", raw[18:26])
    stride = (w * 3 + 3) // 4 * 4
    img = np.zeros((h, w, 3), np.uint8)
    for y in range(h):
        row = np.frombuffer(raw[54 + (h - 1 - y) * stride: 54 + (h - 1 - y) * stride + w * 3], np.uint8).reshape(w, 3)
        img[y] = row[:, ::-1]
    return img


@pytest.mark.parametrize("shim_batch,iters", [(4, 7), (1, 3), (4, 4)])
def test_headless_driver_matches_oracle(tmp_path, shim_batch, iters):
    pkg = load_package()
    assert os.path.exists(pkg.HEADLESS_PATH)
    W, H, depth = 96, 64, 4
    scene = os.path.join(ROOT, "scenes", "sampleScene_spec.txt")
    env = dict(os.environ, PT_DEPTH=str(depth), PT_SHIM_BATCH=str(shim_batch))
    res = subprocess.run([pkg.HEADLESS_PATH, f"scene={scene}", "frame=0", f"res={W}x{H}", f"iterations={iters}",
                          f"out={tmp_path}"], env=env, capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "Saved frame 0" in res.stdout
    got = read_bmp(os.path.join(tmp_path, "spec.0.bmp"))

    sc = O.LoadedScene(scene)
    sc.set_resolution(W, H)
    ref, _ = O.render(sc.geoms, sc.n_objects, sc.mats, sc.n_materials, sc.camera, depth, iters=iters)
    q = np.clip(ref * np.float32(255.0), 0, 255).astype(np.uint8)[:, ::-1, :]      # buffer x -> picture W-1-x
    assert got.shape == q.shape
    assert np.array_equal(got, q)


def test_headless_error_convention(tmp_path):
    """The shim keeps checkCUDAError's print-and-exit convention (ref src/raytraceKernel.cu:19-25)."""
    pkg = load_package()
    scene = os.path.join(ROOT, "scenes", "sampleScene.txt")
    env = dict(os.environ, PT_DEPTH="0")       # invalid depth -> pt_set_options fails inside the shim
    res = subprocess.run([pkg.HEADLESS_PATH, f"scene={scene}", "res=16x16", "iterations=1", f"out={tmp_path}"],
                         env=env, capture_output=True, text=True, timeout=120)
    assert res.returncode != 0
    assert "Cuda error:" in res.stderr


def test_headless_renders_every_animation_frame(tmp_path):
    """No `frame=` argument: the driver walks all frames like the reference's runCuda (ref src/main.cpp:142-157),
    one image file per frame, each equal to the oracle's render of that frame."""
    pkg = load_package()
    W, H, depth, iters = 80, 60, 4, 3
    scene = os.path.join(ROOT, "scenes", "sampleScene_anim.txt")
    env = dict(os.environ, PT_DEPTH=str(depth))
    res = subprocess.run([pkg.HEADLESS_PATH, f"scene={scene}", f"res={W}x{H}", f"iterations={iters}", "rotat=degrees",
                          f"out={tmp_path}"], env=env, capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stdout + res.stderr
    for frame in range(3):
        got = read_bmp(os.path.join(tmp_path, f"anim.{frame}.bmp"))
        sc = O.LoadedScene(scene, O.ROTAT_DEGREES, frame=frame)
        sc.set_resolution(W, H)
        ref, _ = O.render(sc.geoms, sc.n_objects, sc.mats, sc.n_materials, sc.camera, depth, iters=iters)
        q = np.clip(ref * np.float32(255.0), 0, 255).astype(np.uint8)[:, ::-1, :]
        assert np.array_equal(got, q), f"frame {frame}"
    frames = [read_bmp(os.path.join(tmp_path, f"anim.{f}.bmp")) for f in range(3)]
    assert not np.array_equal(frames[0], frames[1]) and not np.array_equal(frames[1], frames[2])


def test_headless_on_several_contexts(tmp_path):
    """PT_DEVICES=0,0,0: the shim shards the frame into row bands (one context each); same picture."""
    pkg = load_package()
    W, H, depth, iters = 90, 50, 4, 5
    scene = os.path.join(ROOT, "scenes", "sampleScene_spec.txt")
    env = dict(os.environ, PT_DEPTH=str(depth), PT_DEVICES="0,0,0")
    res = subprocess.run([pkg.HEADLESS_PATH, f"scene={scene}", "frame=0", f"res={W}x{H}", f"iterations={iters}",
                          f"out={tmp_path}"], env=env, capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stdout + res.stderr
    got = read_bmp(os.path.join(tmp_path, "spec.0.bmp"))
    sc = O.LoadedScene(scene)
    sc.set_resolution(W, H)
    ref, _ = O.render(sc.geoms, sc.n_objects, sc.mats, sc.n_materials, sc.camera, depth, iters=iters)
    q = np.clip(ref * np.float32(255.0), 0, 255).astype(np.uint8)[:, ::-1, :]
    assert np.array_equal(got, q)


def test_headless_direct_lighting(tmp_path):
    """PT_DIRECT_LIGHT=1 through the reference-signature entry point: the picture the oracle renders with explicit
    light sampling."""
    pkg = load_package()
    W, H, depth, iters = 72, 48, 4, 6
    scene = os.path.join(ROOT, "scenes", "sampleScene.txt")
    env = dict(os.environ, PT_DEPTH=str(depth), PT_DIRECT_LIGHT="1")
    res = subprocess.run([pkg.HEADLESS_PATH, f"scene={scene}", "frame=0", "rotat=degrees", f"res={W}x{H}", f"iterations={iters}",
                          f"out={tmp_path}"], env=env, capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stdout + res.stderr
    got = read_bmp(os.path.join(tmp_path, "test.0.bmp"))
    sc = O.LoadedScene(scene, 1)
    sc.set_resolution(W, H)
    ref, _ = O.render(sc.geoms, sc.n_objects, sc.mats, sc.n_materials, sc.camera, depth, iters=iters, direct_light=1)
    q = np.clip(ref * np.float32(255.0), 0, 255).astype(np.uint8)[:, ::-1, :]
    assert np.array_equal(got, q)


def test_headless_writes_png_when_the_scene_asks_for_one(tmp_path):
    """FILE name.png in the scene file -> "<name>.<frame>.png" (ref: src/main.cpp:138, src/image.cpp:86)."""
    from test_abi import read_png_rgb8
    pkg = load_package()
    W, H, depth, iters = 64, 40, 3, 4
    text = open(os.path.join(ROOT, "scenes", "sampleScene_spec.txt")).read().replace("FILE spec.bmp", "FILE shot.png")
    scene = os.path.join(tmp_path, "scene_png.txt")
    open(scene, "w").write(text)
    env = dict(os.environ, PT_DEPTH=str(depth))
    res = subprocess.run([pkg.HEADLESS_PATH, f"scene={scene}", "frame=0", f"res={W}x{H}", f"iterations={iters}", f"out={tmp_path}"],
                         env=env, capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stdout + res.stderr
    got = read_png_rgb8(os.path.join(tmp_path, "shot.0.png"))
    sc = O.LoadedScene(scene)
    sc.set_resolution(W, H)
    ref, _ = O.render(sc.geoms, sc.n_objects, sc.mats, sc.n_materials, sc.camera, depth, iters=iters)
    q = np.clip(ref * np.float32(255.0), 0, 255).astype(np.uint8)[:, ::-1, :]
    assert np.array_equal(got, q)


def test_headless_driver_renders_mesh_objects(tmp_path):
    """scene=mesh_cornell.txt through the reference protocol: the loader reads the .obj files, the driver hands the
    triangles to the binding (pt_shim_set_meshes), cudaRaytraceCore renders them; BMP == oracle."""
    pkg = load_package()
    W, H, depth, iters = 72, 64, 5, 3
    scene = os.path.join(ROOT, "scenes", "mesh_cornell.txt")
    env = dict(os.environ, PT_DEPTH=str(depth))
    res = subprocess.run([pkg.HEADLESS_PATH, f"scene={scene}", "frame=0", "rotat=degrees", f"res={W}x{H}", f"iterations={iters}",
                          f"out={tmp_path}"], env=env, capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stdout + res.stderr
    got = read_bmp(os.path.join(tmp_path, "mesh.0.bmp"))
    sc = O.LoadedScene(scene, O.ROTAT_DEGREES)
    sc.set_resolution(W, H)
    img, _ = O.render(sc.geoms, sc.n_objects, sc.mats, sc.n_materials, sc.camera, depth, iters=iters, meshes=sc.meshes)
    want = np.clip(img * np.float32(255.0), 0, 255).astype(np.uint8)[:, ::-1, :]      # buffer x -> picture W-1-x
    assert np.array_equal(got, want)


def test_headless_motion_blur(tmp_path):
    """motion=3: every frame of sampleScene_anim.txt that has a successor is rendered with the shutter open until that
    successor (three slices); the last frame is static.  Each image == the oracle's."""
    pkg = load_package()
    W, H, depth, iters, K = 64, 48, 4, 36, 3
    scene = os.path.join(ROOT, "scenes", "sampleScene_anim.txt")
    env = dict(os.environ, PT_DEPTH=str(depth))
    res = subprocess.run([pkg.HEADLESS_PATH, f"scene={scene}", f"res={W}x{H}", f"iterations={iters}", "rotat=degrees", f"motion={K}",
                          f"out={tmp_path}"], env=env, capture_output=True, text=True, timeout=180)
    assert res.returncode == 0, res.stdout + res.stderr
    for frame in range(3):
        got = read_bmp(os.path.join(tmp_path, f"anim.{frame}.bmp"))
        a = O.LoadedScene(scene, O.ROTAT_DEGREES, frame=frame)
        a.set_resolution(W, H)
        kw = {}
        if frame < 2:
            b = O.LoadedScene(scene, O.ROTAT_DEGREES, frame=frame + 1)
            sg, sc = O.motion_slices(a.geoms, b.geoms, a.n_objects, a.camera, b.camera, K, O.ROTAT_DEGREES)
            kw = dict(slice_geoms=sg, slice_cams=sc)
        img, _ = O.render(a.geoms, a.n_objects, a.mats, a.n_materials, a.camera, depth, iters=iters, **kw)
        want = np.clip(img * np.float32(255.0), 0, 255).astype(np.uint8)[:, ::-1, :]
        assert np.array_equal(got, want), frame


def test_headless_motion_blur_per_ray(tmp_path):
    """PT_MOTION_PER_RAY=1 motion=2: the same walk through the frames with a shutter time per ray (two linear segments
    between three knot states) through the reference-signature shim; each image == the oracle's."""
    pkg = load_package()
    W, H, depth, iters, K = 64, 48, 4, 20, 2
    scene = os.path.join(ROOT, "scenes", "sampleScene_anim.txt")
    env = dict(os.environ, PT_DEPTH=str(depth), PT_MOTION_PER_RAY="1")
    res = subprocess.run([pkg.HEADLESS_PATH, f"scene={scene}", f"res={W}x{H}", f"iterations={iters}", "rotat=degrees", f"motion={K}",
                          f"out={tmp_path}"], env=env, capture_output=True, text=True, timeout=180)
    assert res.returncode == 0, res.stdout + res.stderr
    for frame in range(3):
        got = read_bmp(os.path.join(tmp_path, f"anim.{frame}.bmp"))
        a = O.LoadedScene(scene, O.ROTAT_DEGREES, frame=frame)
        a.set_resolution(W, H)
        kw = {}
        if frame < 2:
            b = O.LoadedScene(scene, O.ROTAT_DEGREES, frame=frame + 1)
            kg, kc = O.motion_knots(a.geoms, b.geoms, a.n_objects, a.camera, b.camera, K, O.ROTAT_DEGREES)
            kw = dict(knot_geoms=kg, knot_cams=kc)
        img, _ = O.render(a.geoms, a.n_objects, a.mats, a.n_materials, a.camera, depth, iters=iters, **kw)
        want = np.clip(img * np.float32(255.0), 0, 255).astype(np.uint8)[:, ::-1, :]
        assert np.array_equal(got, want), frame
