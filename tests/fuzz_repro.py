#!/usr/bin/env python3
"""Re-run single fuzz cases with option toggles: python tests/fuzz_repro.py <case> ..."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O  # noqa: E402
from __graft_entry__ import load_package  # noqa: E402
from test_gpu_parity import _extreme_scene, _random_scene, _skip_stress_scene  # noqa: E402

pkg = load_package()


def run(case, override):
    rng = np.random.default_rng(90000 + case)
    n_prims = int(rng.choice([2, 3, 5, 9, 14, 33, 60, 97, 130, 300]))
    geoms, mats, eye, view, up, fovy = _random_scene(5000 + case, n_prims)
    if rng.random() < 0.3:
        mats[3].absorptionCoefficient = O.v3(*rng.uniform(0, 3, 3))
    W, H = int(rng.integers(1, 90)), int(rng.integers(1, 60))
    if case % 3 == 0:
        W = int(rng.choice([8, 16, 24, 32, 40, 64, 72, 88, 128]))
        step = 64 // int(np.gcd(W, 64))
        H = step * int(rng.integers(1, max(2, 56 // step)))
    depth = int(rng.integers(1, 10))
    iters = int(rng.integers(1, 5))
    opts = dict(rr_start=int(rng.integers(-1, depth)), seed=int(rng.integers(0, 1000)))
    gopts = dict(geom_path=int(rng.choice([0, 1, 2, 3, 4, 5, 6, 7, 8] if case % 3 == 0 else [0, 1, 2, 3, 4, 5, 6, 7])), batch=int(rng.choice([0, 1, 2, 3, 7, 16])))
    rng4 = np.random.default_rng(770000 + case)               # (round 4's dimensions, as tests/fuzz_gpu.py draws them)
    gopts["resident"] = int(rng4.choice([-1, 1, 1]))
    os.environ.setdefault("PT_REFILL_MIN", str(int(rng4.choice([1, 4, 16, 33, 64]))))
    if rng4.random() < 0.2:
        geoms, mats, eye, view, up, fovy = _skip_stress_scene(5000 + case)
        n_prims = len(geoms)
    elif rng4.random() < 0.1:
        geoms, mats, eye, view, up, fovy = _extreme_scene(("far", "huge", "tiny", "needle", "zero", "neg")[case % 6], 5000 + case)
        n_prims = len(geoms)
    if rng.random() < 0.4:
        opts["direct_light"] = 1
    if rng.random() < 0.4:
        opts["absorption"] = 1
    if rng.random() < 0.35:
        opts["scatter"] = 1
        mats[3].hasScatter, mats[3].reducedScatterCoefficient = 1.0, float(np.float32(rng.uniform(0.2, 6)))
        if rng.random() < 0.5:
            mats[1].hasScatter, mats[1].reducedScatterCoefficient = 1.0, float(np.float32(rng.uniform(0.2, 6)))
            mats[1].absorptionCoefficient = O.v3(*rng.uniform(0, 2, 3))
    if rng.random() < 0.3:
        opts["lens_radius"] = float(np.float32(rng.uniform(0.05, 0.6)))
        opts["focal_distance"] = float(np.float32(rng.uniform(2, 12)))
    meshes = None
    if rng.random() < 0.2:
        nt = int(rng.integers(1, 40))
        geoms.append(O.make_geom(O.MESH, int(rng.integers(0, 4)), rng.uniform(-4, 4, 3), rng.uniform(-3.2, 3.2, 3), rng.uniform(0.5, 4.0, 3)))
        meshes = {len(geoms) - 1: rng.uniform(-0.5, 0.5, (nt, 9)).astype(np.float32)}
        gopts["geom_path"] = int(rng.choice([0, 1, 7, 8]))
    for k, v in override.items():
        if k in ("geom_path", "batch"):
            gopts[k] = v
        elif k == "iters":
            iters = v
        elif k == "depth":
            depth = v
        elif v is None:
            opts.pop(k, None)
        else:
            opts[k] = v
    ga = (O.StaticGeom * len(geoms))(*geoms)
    ma = (O.Material * len(mats))(*mats)
    cam = O.make_camera(W, H, eye, view, up, fovy)
    sh = []
    ref, live = O.render(ga, len(geoms), ma, len(mats), cam, depth, iters=iters, shadow_out=sh, meshes=meshes, **opts)
    with pkg.Renderer(0) as r:
        r.set_options(depth=depth, **opts, **gopts)
        r.set_scene(C.cast(ga, C.POINTER(pkg.StaticGeom)), len(geoms), C.cast(ma, C.POINTER(pkg.Material)), len(mats))
        if meshes:
            r.set_meshes(meshes)
        r.set_camera(pkg.CameraData.from_buffer_copy(cam))
        r.clear_image()
        r.render(1, iters)
        img = r.download_image()
        st = r.stats()
    d = np.abs(img - ref)
    print(f"case {case} {override}: {W}x{H} depth={depth} iters={iters} {opts} {gopts}: max|d|={d.max():g} "
          f"live gpu={[int(x) for x in st.live_in[:depth]]} cpu={[int(x) for x in live]} shadows {int(st.shadow_rays)}/{sh[0]}")
    if d.max() > 0:
        ys, xs = np.nonzero(d.max(axis=2))
        for y, x in list(zip(ys, xs))[:4]:
            print("   px", x, y, "gpu", img[y, x], "cpu", ref[y, x])


if __name__ == "__main__":
    for c in sys.argv[1:]:
        c = int(c)
        run(c, {})
        for ov in ({"iters": 1}, {"batch": 1}, {"geom_path": 1}, {"geom_path": 3}, {"geom_path": 5}, {"absorption": None},
                   {"direct_light": None}, {"lens_radius": None}, {"rr_start": -1}, {"depth": 2}):
            run(c, ov)
