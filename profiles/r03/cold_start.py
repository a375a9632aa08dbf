#!/usr/bin/env python3
"""Where does the cold start go?  One context, config 2; K-step calls back to back from a cold GPU, each timed on the
host (synchronised), with the in-kernel shader clock of the call's last bounce-1 launch (PT_DEBUG_CLOCK=1 prints it at
pt_get_stats).  Then an idle gap and the same again.  PT_PRETOUCH=1 writes the pools once at configure.
    PT_DEBUG_CLOCK=1 python3 profiles/r03/cold_start.py [steps per call] 2> clocks.txt"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package

pkg = load_package()
K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
sc = pkg.SceneFile(os.path.join(ROOT, "scenes", "sampleScene_spec.txt"))
sc.set_resolution(1920, 1080)
t_start = time.perf_counter()
r = pkg.Renderer(0)
r.set_options(depth=8)
r.set_scene(sc.geoms, sc.n_objects, sc.mats, sc.n_materials)
r.set_camera(sc.camera)
t0 = time.perf_counter()
r.clear_image()
r.synchronize()
print(f"configure + clear: {(time.perf_counter() - t0) * 1e3:.1f} ms", flush=True)
it = 1


def burst(n, label):
    global it
    for k in range(n):
        r.reset_stats()
        t0 = time.perf_counter()
        r.render(it, K)
        r.synchronize()
        dt = time.perf_counter() - t0
        st = r.stats()
        it += K
        print(f"{label} call {k:2d}: t = {(time.perf_counter() - t_start) * 1e3:7.1f} ms, {dt * 1e3:6.2f} ms host, {st.gpu_ms:6.2f} ms events, "
              f"{st.ray_bounces / dt / 1e6:8.0f} Mray-bounces/s (host), {st.ray_bounces / (st.gpu_ms * 1e-3) / 1e6:8.0f} (events)", flush=True)


burst(24, "cold")
time.sleep(0.5)
burst(8, "after 500 ms idle")
time.sleep(0.05)
burst(4, "after 50 ms idle")
r.close()
