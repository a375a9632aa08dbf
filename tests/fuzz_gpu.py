#!/usr/bin/env python3
"""Randomised differential run of the HIP path against the oracle (not collected by pytest: run by hand on a GPU box).

    python tests/fuzz_gpu.py [cases=200] [seed0=0]

Every case draws a random scene (tests/test_gpu_parity.py::_random_scene), camera, and a random mix of options
(depth, Russian roulette, seed, geometry path, batch, direct lighting, absorption, thin lens, strip tile) and demands
bit-identical images, live-ray counts and shadow-ray counts."""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O  # noqa: E402
from __graft_entry__ import load_package  # noqa: E402
from test_gpu_parity import _extreme_scene, _random_scene, _skip_stress_scene  # noqa: E402


class Case:
    """One fuzz case as data: scene (ctypes arrays), camera, the oracle's and the library's options, and how it is to be rendered."""


def build_case(case):
    """Every random draw of case number `case`, in the order the fuzz has always made them (a case number must keep meaning the same
    scene and options for good: the cases that found something are regression tests).  Renders nothing."""
    rng = np.random.default_rng(90000 + case)
    n_prims = int(rng.choice([2, 3, 5, 9, 14, 33, 60, 97, 130, 300]))
    geoms, mats, eye, view, up, fovy = _random_scene(5000 + case, n_prims)
    if rng.random() < 0.3:
        mats[3].absorptionCoefficient = O.v3(*rng.uniform(0, 3, 3))
    W, H = int(rng.integers(1, 90)), int(rng.integers(1, 60))
    if case % 3 == 0:
        # tiles whose pixel count is a multiple of 64: the camera rays then go through the host's span table / lists
        W = int(rng.choice([8, 16, 24, 32, 40, 64, 72, 88, 128]))
        step = 64 // int(np.gcd(W, 64))
        H = step * int(rng.integers(1, max(2, 56 // step)))
    depth = int(rng.integers(1, 10))
    iters = int(rng.integers(1, 5))
    opts = dict(rr_start=int(rng.integers(-1, depth)), seed=int(rng.integers(0, 1000)))
    gopts = dict(geom_path=int(rng.choice([0, 1, 2, 3, 4, 5, 6, 7, 8] if case % 3 == 0 else [0, 1, 2, 3, 4, 5, 6, 7])), batch=int(rng.choice([0, 1, 2, 3, 7, 16])))
    # round 4: resident paths (one launch for all later bounces) on or off, and the number of free lanes that triggers a
    # wave's refill -- from a generator of its own, so that the cases of the earlier rounds stay what they were
    rng4 = np.random.default_rng(770000 + case)
    gopts["resident"] = int(rng4.choice([-1, 1, 1]))
    os.environ["PT_REFILL_MIN"] = str(int(rng4.choice([1, 4, 16, 33, 64])))
    # (PT_FUZZ_SCENES=stress | extreme: every case from that generator -- for a targeted run)
    p_stress = {"stress": 1.0, "extreme": 0.0}.get(os.environ.get("PT_FUZZ_SCENES", ""), 0.2)
    p_extreme = {"stress": 0.0, "extreme": 1.0}.get(os.environ.get("PT_FUZZ_SCENES", ""), 0.1)
    if rng4.random() < p_stress:
        # ... and scenes where the reference's sphere arithmetic loses its digits (spheres far smaller than the rays that reach them
        # are long: the culling bounds have to hold what the test HITS, and a resident path may only skip the primitive it leaves
        # where the test would miss it -- Prim::self_r2)
        geoms, mats, eye, view, up, fovy = _skip_stress_scene(5000 + case)
        n_prims = len(geoms)
    elif rng4.random() < p_extreme:
        # ... and scenes at the edges of fp32 (far from the origin, huge, tiny, needles, zero and negative scales)
        geoms, mats, eye, view, up, fovy = _extreme_scene(("far", "huge", "tiny", "needle", "zero", "neg")[case % 6], 5000 + case)
        n_prims = len(geoms)
    if rng.random() < 0.4:
        opts["direct_light"] = 1
    if rng.random() < 0.4:
        opts["absorption"] = 1
    if rng.random() < 0.35:
        # a scattering medium behind the glass material and (sometimes) an index-matched one behind a diffuse one
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
        # a MESH object with a handful of random triangles (object space), placed like any other object
        nt = int(rng.integers(1, 40))
        # (material 4 is the light's: the mesh is then an entry of the light table -- sampled by triangle when direct
        # lighting is on -- and some scenes get more emissive geoms than the table's 16 entries)
        geoms.append(O.make_geom(O.MESH, int(rng.integers(0, 5)), rng.uniform(-4, 4, 3), rng.uniform(-3.2, 3.2, 3), rng.uniform(0.5, 4.0, 3)))
        meshes = {len(geoms) - 1: rng.uniform(-0.5, 0.5, (nt, 9)).astype(np.float32)}
        gopts["geom_path"] = int(rng.choice([0, 1, 7, 8]))
    if rng.random() < 0.1:
        for g in geoms[2:min(len(geoms), 24)]:          # many small emitters: the light table overflows past 16
            if g.type != O.MESH and rng.random() < 0.8:
                g.materialid = 4
    if os.environ.get("PT_FUZZ_FORCE_GEOM"):
        gopts["geom_path"] = int(os.environ["PT_FUZZ_FORCE_GEOM"])
    strip = None
    if rng.random() < 0.3 and H >= 4:
        world = int(rng.integers(2, 4))
        srows = int(rng.integers(1, max(2, H // world)))
        if srows * world <= H + srows - 1:
            strip = (srows, world, int(rng.integers(0, world)))
    ga = (O.StaticGeom * len(geoms))(*geoms)
    ma = (O.Material * len(mats))(*mats)
    cam = O.make_camera(W, H, eye, view, up, fovy)
    motion = None
    if meshes is None and rng.random() < 0.15:
        # motion blur with a shutter time per ray: a second frame (every object moved, turned and rescaled a little,
        # sometimes the camera too), 1..5 linear segments; scalar geometry path
        gb = [O.make_geom(g.type, g.materialid, np.array(g.translation.tup()) + rng.normal(0, 0.4, 3),
                          np.array(g.rotation.tup()) + rng.normal(0, 0.3, 3), np.array(g.scale.tup()) * rng.uniform(0.8, 1.25, 3)) for g in geoms]
        gba = (O.StaticGeom * len(gb))(*gb)
        cam_b = O.make_camera(W, H, eye + rng.normal(0, 0.3, 3), view + rng.normal(0, 0.05, 3), up, fovy) if rng.random() < 0.5 else None
        motion = (gba, cam_b, int(rng.integers(1, 6)))
        gopts["geom_path"] = int(rng.choice([0, 1, 5])) if n_prims <= 100 else int(rng.choice([0, 1]))
        strip = None if rng.random() < 0.5 else strip
    c = Case()
    c.case, c.rng = case, rng
    c.geoms, c.mats, c.ga, c.ma, c.cam = geoms, mats, ga, ma, cam
    c.n_prims, c.W, c.H, c.depth, c.iters = n_prims, W, H, depth, iters
    c.opts, c.gopts, c.strip, c.meshes, c.motion = opts, gopts, strip, meshes, motion
    # the single-process multi-device handle (several contexts on device 0): bands or strips, host gather
    c.multi = None
    if motion is None and strip is None and rng.random() < 0.25 and H >= 4:
        ndev = int(rng.integers(1, 5))
        srows = int(rng.choice([0, 1, 2, 8]))
        if ndev <= H and (srows == 0 or srows * ndev <= H + srows - 1):
            c.multi = (ndev, srows)
    return c


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    pkg = load_package()
    from project3_pathtracer_amd import sharding
    bad = 0
    t0 = time.time()
    for case in range(seed0, seed0 + cases):
        cs = build_case(case)
        rng, geoms, mats, ga, ma, cam = cs.rng, cs.geoms, cs.mats, cs.ga, cs.ma, cs.cam
        n_prims, W, H, depth, iters = cs.n_prims, cs.W, cs.H, cs.depth, cs.iters
        opts, gopts, strip, meshes, motion = cs.opts, cs.gopts, cs.strip, cs.meshes, cs.motion
        if os.environ.get("PT_FUZZ_DUMP"):
            # the case as data (for a look at it with the oracle alone): geoms, materials, camera as bytes, the options
            import pickle
            with open(os.environ["PT_FUZZ_DUMP"], "wb") as f:
                pickle.dump(dict(case=case, geoms=bytes(ga), n_geoms=len(geoms), mats=bytes(ma), n_mats=len(mats), cam=bytes(cam), W=W, H=H,
                                 depth=depth, iters=iters, opts=opts, gopts=gopts, strip=strip, meshes=meshes, motion=bool(motion)), f)
            return 0
        sh = []
        if motion:
            kg, kc = O.motion_knots(ga, motion[0], len(geoms), cam, motion[1], motion[2])
            ref, live = O.render(ga, len(geoms), ma, len(mats), cam, depth, iters=iters, shadow_out=sh, knot_geoms=kg, knot_cams=kc, **opts)
        else:
            ref, live = O.render(ga, len(geoms), ma, len(mats), cam, depth, iters=iters, shadow_out=sh, meshes=meshes, **opts)
        if cs.multi:
            # the single-process multi-device handle (several contexts on device 0): bands or strips, host gather
            ndev, srows = cs.multi
            if True:
                L = pkg.lib()
                devs = (C.c_int * ndev)(*([0] * ndev))
                m = C.c_void_p()
                assert L.pt_multi_create(devs, ndev, C.byref(m)) == 0
                try:
                    o = pkg.Options()
                    L.pt_default_options(C.byref(o))
                    o.depth = depth
                    for k, v in dict(opts, **gopts).items():
                        setattr(o, k, v)
                    if gopts["geom_path"] in (2, 3, 5) and n_prims > 200 and not meshes:
                        o.geom_path = 0
                    assert L.pt_multi_set_options(m, C.byref(o)) == 0, L.pt_last_error()
                    assert L.pt_multi_set_strips(m, srows) == 0
                    assert L.pt_multi_set_scene(m, C.cast(ga, C.POINTER(pkg.StaticGeom)), len(geoms), C.cast(ma, C.POINTER(pkg.Material)), len(mats)) == 0
                    if meshes:
                        (gi, tv), = meshes.items()
                        md = (pkg.Mesh * 1)(pkg.Mesh(gi, tv.shape[0], tv.ctypes.data_as(C.POINTER(C.c_float))))
                        assert L.pt_multi_set_meshes(m, md, 1) == 0, L.pt_last_error()
                    assert L.pt_multi_set_camera(m, C.cast(C.byref(cam), C.POINTER(pkg.CameraData))) == 0
                    assert L.pt_multi_clear_image(m) == 0
                    assert L.pt_multi_render(m, 1, iters) == 0, L.pt_last_error()
                    host = np.zeros((H, W, 3), dtype=np.float32)
                    assert L.pt_multi_download_image(m, host.ctypes.data) == 0
                    st = pkg.Stats()
                    assert L.pt_multi_get_stats(m, C.byref(st)) == 0
                finally:
                    L.pt_multi_destroy(m)
                ok = (np.array_equal(host.view(np.uint32), ref.view(np.uint32)) and
                      [int(x) for x in st.live_in[:depth]] == [int(x) for x in live] and int(st.shadow_rays) == sh[0])
                if not ok:
                    bad += 1
                    print(f"MISMATCH (multi) case {case}: ndev={ndev} strips={srows} prims={n_prims} {W}x{H} depth={depth} iters={iters} "
                          f"{opts} {gopts} max|d|={np.abs(host - ref).max():g}", flush=True)
                continue
        try:
            with pkg.Renderer(0) as r:
                so = dict(strip_rows=strip[0], strip_world=strip[1], strip_rank=strip[2]) if strip else {}
                r.set_options(depth=depth, **opts, **gopts, **so)
                r.set_scene(C.cast(ga, C.POINTER(pkg.StaticGeom)), len(geoms), C.cast(ma, C.POINTER(pkg.Material)), len(mats))
                if meshes:
                    r.set_meshes(meshes)
                r.set_camera(pkg.CameraData.from_buffer_copy(cam))
                if motion:
                    r.set_options(motion_per_ray=1)
                    r.set_motion(C.cast(motion[0], C.POINTER(pkg.StaticGeom)), motion[1], motion[2], pkg.ROTAT_RADIANS)
                r.clear_image()
                r.render(1, iters)
                img = r.download_image()
                st = r.stats()
        except pkg.PtError as e:
            if "LDS" in str(e) and gopts["geom_path"] in (2, 3, 5):      # explicit LDS-resident path, scene too large for it
                continue
            raise
        if strip:
            want = ref[sharding.strip_global_rows(H, strip[1], strip[2], strip[0])]
            ok = np.array_equal(img.view(np.uint32), want.view(np.uint32))
        else:
            ok = (np.array_equal(img.view(np.uint32), ref.view(np.uint32)) and
                  [int(x) for x in st.live_in[:depth]] == [int(x) for x in live] and int(st.shadow_rays) == sh[0])
        if not ok and os.environ.get("PT_FUZZ_VERBOSE"):
            d = np.abs(img - (want if strip else ref)).max(axis=2)
            ys, xs = np.nonzero(d)
            print(f"   {len(ys)} pixels differ; live gpu {[int(x) for x in st.live_in[:depth]]} cpu {[int(x) for x in live]}")
            for y, x in list(zip(ys, xs))[:6]:
                print("   px", x, y, "gpu", img[y, x], "cpu", (want if strip else ref)[y, x])
        if not ok:
            bad += 1
            print(f"MISMATCH case {case}: prims={n_prims} {W}x{H} depth={depth} iters={iters} {opts} {gopts} strip={strip} motion={motion[2] if motion else None} "
                  f"max|d|={np.abs(img - (want if strip else ref)).max():g}", flush=True)
        if (case - seed0) % 25 == 24:
            print(f"... {case - seed0 + 1} cases, {bad} mismatches, {time.time() - t0:.0f} s", flush=True)
    print(f"fuzz: {cases} cases, {bad} mismatches")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
