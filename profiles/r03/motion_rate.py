#!/usr/bin/env python3
"""Throughput of motion blur on sampleScene_anim.txt at 1920x1080, 8 bounces: static frame, slice scheme (4 slices), a shutter time per ray
(4 segments).  python3 profiles/r03/motion_rate.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package

pkg = load_package()
path = os.path.join(ROOT, "scenes", "sampleScene_anim.txt")
a, b = pkg.SceneFile(path, 1, frame=0), pkg.SceneFile(path, 1, frame=1)
a.set_resolution(1920, 1080)
for label, per_ray, slices in (("static", 0, 0), ("slices x4", 0, 4), ("per ray, 4 segments", 1, 4)):
    with pkg.Renderer(0) as r:
        r.set_options(depth=8, motion_per_ray=per_ray)
        r.set_scene(a.geoms, a.n_objects, a.mats, a.n_materials)
        r.set_camera(a.camera)
        if slices:
            r.set_motion(b.geoms, b.camera, slices, pkg.ROTAT_DEGREES)
        r.clear_image()
        r.render(1, 64)
        r.synchronize()
        r.reset_stats()
        t0 = time.perf_counter()
        r.render(65, 128)
        r.synchronize()
        dt = time.perf_counter() - t0
        st = r.stats()
        print(f"{label:22s} {st.ray_bounces / dt / 1e6:9.0f} Mray-bounces/s  ({dt * 1e3 / 128:.3f} ms per 1-spp frame)", flush=True)
