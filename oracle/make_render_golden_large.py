#!/usr/bin/env python3
"""SHA-256 of the ORACLE's fp32 images at the shapes bench.py actually launches (BASELINE.json configs 2-5), written to
tests/golden/render_checksums_large.json.  Takes a few minutes of CPU time (all host cores).

    python oracle/make_render_golden_large.py [case ...]

The GPU tests (tests/test_gpu_bench_shapes.py) hash the HIP path's image of the same configuration and demand the same
digest and live-ray counts, so the benchmarked launch shapes (1080p with 16 iterations in flight, a 4K frame as one tile
and as 8 strip tiles, the 256-primitive cloud at depth 32) are pinned to the oracle, not to themselves.  The oracle side is
re-checked by tests/test_oracle_kat.py::test_render_checksums_large when PT_TEST_LARGE_GOLDEN=1 (too slow for the
default CPU suite)."""
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O  # noqa: E402

CASES = [
    # name, scene, rotat, W, H, depth, iters, options
    ("config2_1080p_32spp", "sampleScene_spec.txt", 0, 1920, 1080, 8, 32, {}),
    ("config3_glass_1080p_16spp_depth16", "cornell_glass.txt", 1, 1920, 1080, 16, 16, {}),
    ("config4_4k_16spp", "sampleScene_spec.txt", 0, 3840, 2160, 8, 16, {}),
    ("config5_cloud_480x270_depth32", "cloud256.txt", 1, 480, 270, 32, 4, {}),
    ("config5_cloud_480x270_depth32_rr", "cloud256.txt", 1, 480, 270, 32, 4, {"rr_start": 3}),
]
PATH = os.path.join(ROOT, "tests", "golden", "render_checksums_large.json")


def run(case):
    name, scene, rotat, W, H, depth, iters, opts = case
    sc = O.LoadedScene(os.path.join(ROOT, "scenes", scene), rotat)
    sc.set_resolution(W, H)
    t0 = time.time()
    img, live = O.render(sc.geoms, sc.n_objects, sc.mats, sc.n_materials, sc.camera, depth, iters=iters, **opts)
    # a few rows kept verbatim so that a mismatch can be localised without re-running the oracle
    rows = {str(y): hashlib.sha256(img[y].tobytes()).hexdigest() for y in (0, H // 3, H // 2, H - 1)}
    print(f"{name}: {time.time() - t0:.1f} s, {int(live.sum())} ray-bounces", flush=True)
    return {"scene": scene, "rotat": rotat, "width": W, "height": H, "depth": depth, "iterations": iters, "options": opts,
            "sha256": hashlib.sha256(img.tobytes()).hexdigest(), "row_sha256": rows, "live_in": [int(x) for x in live],
            "mean_rgb": [float(x) for x in img.mean(axis=(0, 1))]}


if __name__ == "__main__":
    O.build()
    out = {}
    if os.path.exists(PATH):
        out = json.load(open(PATH))
    want = sys.argv[1:]
    for c in CASES:
        if want and c[0] not in want:
            continue
        out[c[0]] = run(c)
        with open(PATH, "w") as f:
            json.dump(out, f, indent=1)
    print("wrote", PATH)
