#!/usr/bin/env python3
"""Pins the canonical render semantics across rounds: SHA-256 of the oracle's fp32 images and the
live-ray counts for a handful of small configurations, written to tests/golden/render_checksums.json.

    python oracle/make_render_golden.py          # rewrites the fixture (only when the semantics change on purpose)

tests/test_oracle_kat.py::test_render_checksums recomputes them; the GPU parity tests compare against the same oracle,
so a drift of either side shows up."""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O  # noqa: E402

CASES = [
    # name, scene, rotat, W, H, depth, iters, options
    ("config1_400x400_depth4", "sampleScene.txt", 0, 400, 400, 4, 1, {}),
    ("spec_radians", "sampleScene_spec.txt", 0, 96, 54, 8, 3, {}),
    ("spec_degrees_rr", "sampleScene_spec.txt", 1, 96, 54, 12, 2, {"rr_start": 2}),
    ("glass", "cornell_glass.txt", 1, 80, 60, 10, 2, {}),
    ("glass_absorption", "cornell_glass.txt", 1, 80, 60, 10, 2, {"absorption": 1}),
    ("direct_light", "sampleScene.txt", 1, 80, 60, 4, 2, {"direct_light": 1}),
    ("cloud256", "cloud256.txt", 1, 64, 36, 6, 1, {"seed": 5}),
    ("thin_lens", "sampleScene_spec.txt", 1, 72, 54, 5, 2, {"lens_radius": 0.3, "focal_distance": 11.0}),
    ("subsurface", "sss_blobs.txt", 1, 72, 72, 10, 2, {"scatter": 1}),
    # round 3 (options with a leading underscore are handled by run() / the GPU test, not passed to the renderer as they are):
    # an emissive MESH geom as a light of the table; motion blur with a shutter time per ray over 2 segments (frames 0 -> 1)
    ("mesh_light", "mesh_light.txt", 1, 80, 64, 5, 2, {"direct_light": 1, "rr_start": 2, "_meshes": 1}),
    ("motion_per_ray", "sampleScene_anim.txt", 1, 80, 60, 5, 3, {"_knots": 2}),
]


def run(case):
    name, scene, rotat, W, H, depth, iters, opts = case
    sc = O.LoadedScene(os.path.join(ROOT, "scenes", scene), rotat)
    sc.set_resolution(W, H)
    kw = {k: v for k, v in opts.items() if not k.startswith("_")}
    if opts.get("_meshes"):
        kw["meshes"] = sc.meshes
    if opts.get("_knots"):
        nxt = O.LoadedScene(os.path.join(ROOT, "scenes", scene), rotat, frame=1)
        kw["knot_geoms"], kw["knot_cams"] = O.motion_knots(sc.geoms, nxt.geoms, sc.n_objects, sc.camera, nxt.camera, opts["_knots"], rotat)
    img, live = O.render(sc.geoms, sc.n_objects, sc.mats, sc.n_materials, sc.camera, depth, iters=iters, **kw)
    return {"scene": scene, "rotat": rotat, "width": W, "height": H, "depth": depth, "iterations": iters, "options": opts,
            "sha256": hashlib.sha256(img.tobytes()).hexdigest(), "live_in": [int(x) for x in live],
            "mean_rgb": [float(x) for x in img.mean(axis=(0, 1))]}


if __name__ == "__main__":
    out = {c[0]: run(c) for c in CASES}
    path = os.path.join(ROOT, "tests", "golden", "render_checksums.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", path)
