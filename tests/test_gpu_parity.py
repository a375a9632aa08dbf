"""GPU parity tests: the HIP path, called through the C-ABI, against the CPU oracle.

Bar (BASELINE.json north_star): float RGB within 1e-4 of the CPU re-execution at a fixed seed.  Because both
sides evaluate the same fp32 operations with one rounding each, the images are expected to be IDENTICAL; the
tests assert max-abs error <= TOL and report whether the match was bit-exact.  Ray-bounce counts (integer
work) must be equal.
"""
import ctypes as C
import os
import sys

import numpy as np
import pytest

import oracle_lib as O
from __graft_entry__ import load_package

pytestmark = pytest.mark.gpu

TOL = 1e-4   # north_star tolerance for fp32 RGB
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCENES = os.path.join(ROOT, "scenes")


@pytest.fixture(scope="module")
def pkg():
    p = load_package()
    p.lib()   # raises if libptamd.so is missing: there is no fallback path
    return p


def gpu_render(pkg, scene, w, h, depth, iters=1, iter_first=1, image=None, rotat=0, **opts):
    sc = pkg.SceneFile(os.path.join(SCENES, scene), rotat)
    sc.set_resolution(w, h)
    with pkg.Renderer(0) as r:
        r.set_options(depth=depth, **opts)
        r.set_scene(sc.geoms, sc.n_objects, sc.mats, sc.n_materials)
        r.set_camera(sc.camera)
        if image is None:
            r.clear_image()
        else:
            r.upload_image(image)
        r.render(iter_first, iters)
        img = r.download_image()
        st = r.stats()
    return img, [int(x) for x in st.live_in[:depth]], st


def cpu_render(scene, w, h, depth, iters=1, iter_first=1, image=None, rotat=0, rr_start=-1, seed=0, **more):
    sc = O.LoadedScene(os.path.join(SCENES, scene), rotat)
    sc.set_resolution(w, h)
    img, live = O.render(sc.geoms, sc.n_objects, sc.mats, sc.n_materials, sc.camera, depth, iters=iters,
                         iter_first=iter_first, rr_start=rr_start, seed=seed, image=image, **more)
    return img, [int(x) for x in live]


def check(gpu, cpu, live_g, live_c, what):
    err = float(np.abs(gpu - cpu).max())
    exact = bool((gpu.view(np.uint32) == cpu.view(np.uint32)).all())
    print(f"{what}: max|GPU-CPU| = {err:g}, bit-exact = {exact}, ray-bounces = {sum(live_g)}")
    assert err <= TOL, what
    assert live_g == live_c, f"{what}: live-ray counts differ {live_g} vs {live_c}"
    assert np.isfinite(gpu).all()


def test_exact_math_sequences_exhaustive(pkg):
    """The kernels' short sqrt / 1/x / 1/sqrt sequences equal the compiler's correctly rounded ones for all
    2^32 fp32 inputs (so replacing them cannot change a single bit of any image)."""
    with pkg.Renderer(0) as r:
        assert r.selftest_math() == [0, 0, 0]


# ---------------------------------------------------------------- BASELINE config 1 and friends
def test_config1_sample_scene_400x400_depth4(pkg):
    """BASELINE.json configs[0]: sampleScene.txt, 400x400, 1 spp, 4 bounces, fixed seed."""
    g, lg, _ = gpu_render(pkg, "sampleScene.txt", 400, 400, 4)
    c, lc = cpu_render("sampleScene.txt", 400, 400, 4)
    check(g, c, lg, lc, "config1")
    assert lg[0] == 160000 and all(a >= b for a, b in zip(lg, lg[1:]))
    assert g.max() > 0     # some paths reach the light


def test_config1_degrees(pkg):
    g, lg, _ = gpu_render(pkg, "sampleScene.txt", 400, 400, 4, rotat=1)
    c, lc = cpu_render("sampleScene.txt", 400, 400, 4, rotat=1)
    check(g, c, lg, lc, "config1 rotat=degrees")


def test_russian_roulette(pkg):
    g, lg, _ = gpu_render(pkg, "sampleScene.txt", 200, 200, 8, rr_start=2)
    c, lc = cpu_render("sampleScene.txt", 200, 200, 8, rr_start=2)
    check(g, c, lg, lc, "russian roulette")
    g0, lg0, _ = gpu_render(pkg, "sampleScene.txt", 200, 200, 8)
    assert sum(lg) < sum(lg0)      # roulette removes rays


def test_specular_scene(pkg):
    g, lg, _ = gpu_render(pkg, "sampleScene_spec.txt", 320, 180, 8, rotat=1)
    c, lc = cpu_render("sampleScene_spec.txt", 320, 180, 8, rotat=1)
    check(g, c, lg, lc, "diffuse+specular")


def test_glass_scene(pkg):
    g, lg, _ = gpu_render(pkg, "cornell_glass.txt", 320, 180, 16, rotat=1)
    c, lc = cpu_render("cornell_glass.txt", 320, 180, 16, rotat=1)
    check(g, c, lg, lc, "glass (refraction + fresnel)")


def test_cloud_256_primitives(pkg):
    g, lg, _ = gpu_render(pkg, "cloud256.txt", 160, 90, 6, rotat=1, rr_start=3)
    c, lc = cpu_render("cloud256.txt", 160, 90, 6, rotat=1, rr_start=3)
    check(g, c, lg, lc, "256-primitive cloud")


@pytest.mark.parametrize("rotat", [0, 1])
@pytest.mark.parametrize("geom_path", [1, 4, 5, 6, 7])
def test_cloud_cull_is_conservative(pkg, rotat, geom_path):
    """Large primitive lists are culled (geom_path 1: per wave with padded bounding spheres; 4: per lane with a
    padded bounding-box hierarchy).  Culling may only skip primitives that cannot be hit, so a bigger sample of
    the 256-primitive scene must still match the oracle bit for bit (a wrongly culled hit is an O(1) pixel error)."""
    g, lg, _ = gpu_render(pkg, "cloud256.txt", 384, 216, 8, iters=2, rotat=rotat, geom_path=geom_path)
    c, lc = cpu_render("cloud256.txt", 384, 216, 8, iters=2, rotat=rotat)
    check(g, c, lg, lc, f"cloud256 cull rotat={rotat} geom_path={geom_path}")


def test_seed_changes_image(pkg):
    a, _, _ = gpu_render(pkg, "sampleScene.txt", 128, 128, 4, seed=0)
    b, lb, _ = gpu_render(pkg, "sampleScene.txt", 128, 128, 4, seed=7)
    c, lc = cpu_render("sampleScene.txt", 128, 128, 4, seed=7)
    assert not np.array_equal(a, b)
    check(b, c, lb, lc, "seed=7")


# ---------------------------------------------------------------- accumulation / statelessness
def test_multi_iteration_running_mean(pkg):
    g, lg, st = gpu_render(pkg, "sampleScene.txt", 160, 120, 5, iters=6)
    c, lc = cpu_render("sampleScene.txt", 160, 120, 5, iters=6)
    check(g, c, lg, lc, "6 iterations")
    assert st.iterations == 6


@pytest.mark.parametrize("batch", [1, 2, 3, 4, 5, 8, 11, 16])
def test_iteration_batching_is_invisible(pkg, batch):
    """1..8 iterations in flight per launch sequence; samples are folded in iteration order, so the running
    mean is the same bits, also when the iteration count is not a multiple of the batch."""
    iters = 37 if batch > 8 else 11
    g, lg, st = gpu_render(pkg, "sampleScene_spec.txt", 121, 67, 5, iters=iters, batch=batch, rr_start=2)
    c, lc = cpu_render("sampleScene_spec.txt", 121, 67, 5, iters=iters, rr_start=2)
    check(g, c, lg, lc, f"batch={batch}")
    assert st.iterations == iters


@pytest.mark.parametrize("batch,iters", [(16, 37), (4, 11), (1, 5), (16, 16), (3, 12)])
def test_launch_sequences_in_flight_are_invisible(pkg, batch, iters):
    """pt_options.sequences = 2: batch n + 1 renders on a second stream (own ray pools, planes, counters) beside batch n;
    only the accumulates are ordered, in iteration order, so image, counts and stats are those of one sequence -- also
    for odd batch counts (the second sequence gets one batch fewer) and calls of a single batch."""
    a, la, sa = gpu_render(pkg, "sampleScene_spec.txt", 121, 67, 5, iters=iters, batch=batch, rr_start=2, sequences=1)
    b, lb, sb = gpu_render(pkg, "sampleScene_spec.txt", 121, 67, 5, iters=iters, batch=batch, rr_start=2, sequences=2)
    c, lc = cpu_render("sampleScene_spec.txt", 121, 67, 5, iters=iters, rr_start=2)
    check(b, c, lb, lc, f"sequences=2 batch={batch}")
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32)) and la == lb
    assert sa.iterations == sb.iterations == iters


def test_launch_sequences_eager_launches_and_strip_tiles(pkg):
    """Two sequences without hipGraphs (use_graph = 0: eager launches on both streams) and on an interleaved-strip tile
    (global pixel numbering), several batches: the tile's rows of the oracle's frame, bit for bit."""
    from project3_pathtracer_amd import sharding
    W, H, depth, iters = 128, 72, 5, 23
    c, _ = cpu_render("sampleScene_spec.txt", W, H, depth, iters=iters, rr_start=1)
    for use_graph in (0, 1):
        sc = pkg.SceneFile(os.path.join(SCENES, "sampleScene_spec.txt"))
        sc.set_resolution(W, H)
        with pkg.Renderer(0) as r:
            r.set_options(depth=depth, rr_start=1, batch=4, sequences=2, use_graph=use_graph, strip_rows=8, strip_world=3, strip_rank=1)
            r.set_scene(sc.geoms, sc.n_objects, sc.mats, sc.n_materials)
            r.set_camera(sc.camera)
            r.clear_image()
            r.render(1, iters)
            g = r.download_image()
        want = c[sharding.strip_global_rows(H, 3, 1, 8)]
        assert np.array_equal(g.view(np.uint32), want.view(np.uint32)), f"use_graph={use_graph}"


def test_launch_sequences_many_small_batches(pkg):
    """96 one-iteration batches alternate between the two sequences: 95 cross-stream hand-overs of the framebuffer (each
    accumulate waits for the other stream's previous one).  Same bits as one sequence and as the oracle."""
    a, la, _ = gpu_render(pkg, "sampleScene_spec.txt", 320, 200, 4, iters=96, batch=1, sequences=1)
    b, lb, _ = gpu_render(pkg, "sampleScene_spec.txt", 320, 200, 4, iters=96, batch=1, sequences=2)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32)) and la == lb
    c, lc = cpu_render("sampleScene_spec.txt", 320, 200, 4, iters=96)
    check(b, c, lb, lc, "96 one-iteration batches on two sequences")


def test_radiance_plane_serials_start_over(pkg, tmp_path):
    """The radiance-plane entries carry 32-bit batch serial numbers; long before they could come round again the library
    zeroes the planes and restarts the count.  PT_SERIAL_BUDGET=3 (read when the library first renders in this process)
    forces that between pt_render calls: a child process renders 14 iterations in 7 calls and must match the oracle."""
    import subprocess
    code = """
import os, sys
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, 'tests'))
import numpy as np
from __graft_entry__ import load_package
pkg = load_package()
sc = pkg.SceneFile(os.path.join({root!r}, 'scenes', 'sampleScene_spec.txt'))
sc.set_resolution(96, 64)
with pkg.Renderer(0) as r:
    r.set_options(depth=4, batch=1)
    r.set_scene(sc.geoms, sc.n_objects, sc.mats, sc.n_materials)
    r.set_camera(sc.camera)
    r.clear_image()
    for k in range(7):
        r.render(1 + 2 * k, 2)
    np.save(sys.argv[1], r.download_image())
""".format(root=ROOT)
    path = str(tmp_path / "serial_budget_test.npy")
    res = subprocess.run([sys.executable, "-c", code, path], env=dict(os.environ, PT_SERIAL_BUDGET="3"), capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout + res.stderr
    g = np.load(path)
    c, _ = cpu_render("sampleScene_spec.txt", 96, 64, 4, iters=14)
    assert np.array_equal(g.view(np.uint32), c.view(np.uint32))


def test_radiance_plane_serials_start_over_on_every_path(pkg, tmp_path):
    """The same restart on the two other paths that render batches: the ordered batches of the motion-blur slice contexts (two
    streams) and pt_render_profiled.  PT_SERIAL_BUDGET=2 in a child process: 3 slices x several runs of 16 iterations, then a
    profiled render on a static scene -- both must match the oracle bit for bit."""
    import subprocess
    code = """
import os, sys
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, 'tests'))
import numpy as np
from __graft_entry__ import load_package
pkg = load_package()
path = os.path.join({root!r}, 'scenes', 'sampleScene_anim.txt')
a, b = pkg.SceneFile(path, 1, frame=0), pkg.SceneFile(path, 1, frame=1)
a.set_resolution(64, 48)
with pkg.Renderer(0) as r:
    r.set_options(depth=4)
    r.set_scene(a.geoms, a.n_objects, a.mats, a.n_materials)
    r.set_camera(a.camera)
    r.set_motion(b.geoms, b.camera, 3, pkg.ROTAT_DEGREES)
    r.clear_image()
    for k in range(6):                      # 6 x 32 iterations: every slice context renders four runs of 16
        r.render(1 + 32 * k, 32)
    np.save(sys.argv[1], r.download_image())
sc = pkg.SceneFile(os.path.join({root!r}, 'scenes', 'sampleScene_spec.txt'))
sc.set_resolution(64, 48)
with pkg.Renderer(0) as r:
    r.set_options(depth=4, batch=2)
    r.set_scene(sc.geoms, sc.n_objects, sc.mats, sc.n_materials)
    r.set_camera(sc.camera)
    r.clear_image()
    for k in range(5):
        r.render_profiled(1 + 2 * k, 2)
    np.save(sys.argv[2], r.download_image())
""".format(root=ROOT)
    p1, p2 = str(tmp_path / "slices.npy"), str(tmp_path / "profiled.npy")
    res = subprocess.run([sys.executable, "-c", code, p1, p2], env=dict(os.environ, PT_SERIAL_BUDGET="2"), capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout + res.stderr
    path = os.path.join(SCENES, "sampleScene_anim.txt")
    oa, ob = O.LoadedScene(path, 1, frame=0), O.LoadedScene(path, 1, frame=1)
    oa.set_resolution(64, 48)
    sg, scm = O.motion_slices(oa.geoms, ob.geoms, oa.n_objects, oa.camera, ob.camera, 3, O.ROTAT_DEGREES)
    c1, _ = O.render(oa.geoms, oa.n_objects, oa.mats, oa.n_materials, oa.camera, 4, iters=192, slice_geoms=sg, slice_cams=scm)
    assert np.array_equal(np.load(p1).view(np.uint32), c1.view(np.uint32))
    c2, _ = cpu_render("sampleScene_spec.txt", 64, 48, 4, iters=10)
    assert np.array_equal(np.load(p2).view(np.uint32), c2.view(np.uint32))


def test_launch_sequences_across_calls_and_features(pkg):
    """Two sequences in flight over several pt_render calls (resume), with direct lighting (planes accumulate along the
    path) and on the batched walk: same bits as the oracle."""
    sc = pkg.SceneFile(os.path.join(SCENES, "sampleScene_spec.txt"))
    sc.set_resolution(96, 64)
    with pkg.Renderer(0) as r:
        r.set_options(depth=4, batch=2, sequences=2, direct_light=1)
        r.set_scene(sc.geoms, sc.n_objects, sc.mats, sc.n_materials)
        r.set_camera(sc.camera)
        r.clear_image()
        r.render(1, 5)
        r.render(6, 1)
        r.render(7, 6)
        img = r.download_image()
        st = r.stats()
    c, lc = cpu_render("sampleScene_spec.txt", 96, 64, 4, iters=12, direct_light=1)
    check(img, c, [int(x) for x in st.live_in[:4]], lc, "sequences=2 over three calls, direct lighting")
    g, lg, _ = gpu_render(pkg, "cloud256.txt", 160, 90, 6, iters=9, batch=2, sequences=2, rotat=1)
    c2, lc2 = cpu_render("cloud256.txt", 160, 90, 6, iters=9, rotat=1)
    check(g, c2, lg, lc2, "sequences=2 on the batched walk")


def test_resume_from_host_image(pkg):
    """(image, iteration) is a complete state: 1..3 then 4..6 from the downloaded image == 1..6."""
    full, lf, _ = gpu_render(pkg, "sampleScene.txt", 96, 64, 4, iters=6)
    half, lh, _ = gpu_render(pkg, "sampleScene.txt", 96, 64, 4, iters=3)
    rest, lr, _ = gpu_render(pkg, "sampleScene.txt", 96, 64, 4, iters=3, iter_first=4, image=half)
    assert np.array_equal(full, rest)
    assert [a + b for a, b in zip(lh, lr)] == lf


def test_render_iteration_reference_protocol(pkg):
    """pt_render_iteration == one cudaRaytraceCore call: host image in/out each call (ref src/main.cpp:93-113)."""
    sc = pkg.SceneFile(os.path.join(SCENES, "sampleScene.txt"))
    sc.set_resolution(64, 48)
    host = np.zeros((48, 64, 3), dtype=np.float32)
    with pkg.Renderer(0) as r:
        r.set_options(depth=3)
        r.set_scene(sc.geoms, sc.n_objects, sc.mats, sc.n_materials)
        r.set_camera(sc.camera)
        for it in range(1, 5):
            r.render_iteration(it, host_image=host)
    c, _ = cpu_render("sampleScene.txt", 64, 48, 3, iters=4)
    assert float(np.abs(host - c).max()) <= TOL


# ---------------------------------------------------------------- invariants of compaction / launch shape
def test_compaction_modes_identical(pkg):
    """per-wave sharded reservation (1), workgroup scan + one counter (2) and no compaction (0): same bits."""
    a, la, _ = gpu_render(pkg, "sampleScene_spec.txt", 200, 150, 6, compaction=1)
    b, lb, _ = gpu_render(pkg, "sampleScene_spec.txt", 200, 150, 6, compaction=0)
    c, lc, _ = gpu_render(pkg, "sampleScene_spec.txt", 200, 150, 6, compaction=2)
    assert np.array_equal(a, b) and np.array_equal(a, c)
    assert la == lb == lc


@pytest.mark.parametrize("compaction", [1, 2])
@pytest.mark.parametrize("wg", [64, 256, 1024])
def test_compaction_modes_against_oracle(pkg, compaction, wg):
    g, lg, _ = gpu_render(pkg, "cornell_glass.txt", 231, 97, 9, iters=2, rotat=1, compaction=compaction, workgroup=wg)
    c, lc = cpu_render("cornell_glass.txt", 231, 97, 9, iters=2, rotat=1)
    check(g, c, lg, lc, f"compaction={compaction} wg={wg}")


@pytest.mark.parametrize("wg", [64, 128, 256, 512, 1024])
def test_workgroup_size_invariance(pkg, wg):
    ref, lr, _ = gpu_render(pkg, "sampleScene.txt", 150, 100, 5)
    a, la, _ = gpu_render(pkg, "sampleScene.txt", 150, 100, 5, workgroup=wg)
    assert np.array_equal(a, ref) and la == lr


def test_geometry_paths_identical(pkg):
    """scalar-direct, LDS-direct and the LDS hit-queue nearest-hit give the same bits."""
    a, la, _ = gpu_render(pkg, "cloud256.txt", 128, 72, 4, geom_path=1, rotat=1)
    b, lb, _ = gpu_render(pkg, "cloud256.txt", 128, 72, 4, geom_path=2, rotat=1)
    c, lc, _ = gpu_render(pkg, "cloud256.txt", 128, 72, 4, geom_path=3, rotat=1)
    d, ld, _ = gpu_render(pkg, "cloud256.txt", 128, 72, 4, geom_path=4, rotat=1)
    assert np.array_equal(a, b) and la == lb
    assert np.array_equal(a, c) and la == lc
    assert np.array_equal(a, d) and la == ld


@pytest.mark.parametrize("geom_path", [0, 1, 2, 3, 4, 5, 6, 7])
@pytest.mark.parametrize("scene,depth,rotat", [("sampleScene.txt", 6, 0), ("cornell_glass.txt", 10, 1),
                                               ("cloud256.txt", 5, 1)])
def test_every_geometry_path_against_oracle(pkg, geom_path, scene, depth, rotat):
    g, lg, _ = gpu_render(pkg, scene, 192, 108, depth, iters=2, rotat=rotat, geom_path=geom_path, rr_start=3)
    c, lc = cpu_render(scene, 192, 108, depth, iters=2, rotat=rotat, rr_start=3)
    check(g, c, lg, lc, f"{scene} geom_path={geom_path}")


@pytest.mark.parametrize("wg", [64, 256, 1024])
def test_hit_queue_workgroup_sizes(pkg, wg):
    ref, lr, _ = gpu_render(pkg, "sampleScene_spec.txt", 150, 100, 6, geom_path=1)
    a, la, _ = gpu_render(pkg, "sampleScene_spec.txt", 150, 100, 6, geom_path=3, workgroup=wg)
    b, lb, _ = gpu_render(pkg, "sampleScene_spec.txt", 150, 100, 6, geom_path=3, workgroup=wg, compaction=0)
    assert np.array_equal(a, ref) and la == lr
    assert np.array_equal(b, ref) and lb == lr


def test_graph_vs_eager(pkg):
    a, la, _ = gpu_render(pkg, "sampleScene.txt", 128, 96, 4, iters=5, use_graph=1)
    b, lb, _ = gpu_render(pkg, "sampleScene.txt", 128, 96, 4, iters=5, use_graph=0)
    assert np.array_equal(a, b) and la == lb


def test_tiles_reassemble_to_full_frame(pkg):
    """Pixel sharding: rows split over contexts, RNG keyed on the global pixel -> bit-identical union."""
    W, H, depth = 120, 90, 5
    full, lf, _ = gpu_render(pkg, "sampleScene.txt", W, H, depth, iters=2)
    parts, live = [], [0] * depth
    for r0, r1 in ((0, 17), (17, 64), (64, 90)):
        t, lt, _ = gpu_render(pkg, "sampleScene.txt", W, H, depth, iters=2, row_begin=r0, row_end=r1)
        assert t.shape == (r1 - r0, W, 3)
        parts.append(t)
        live = [a + b for a, b in zip(live, lt)]
    assert np.array_equal(np.concatenate(parts, axis=0), full)
    assert live == lf


# ---------------------------------------------------------------- edge cases
@pytest.mark.parametrize("w,h", [(1, 1), (7, 3), (65, 1), (64, 64), (37, 23), (257, 5)])
def test_ragged_resolutions(pkg, w, h):
    g, lg, _ = gpu_render(pkg, "sampleScene.txt", w, h, 3)
    c, lc = cpu_render("sampleScene.txt", w, h, 3)
    check(g, c, lg, lc, f"{w}x{h}")


def test_depth_one_and_max(pkg):
    g, lg, _ = gpu_render(pkg, "sampleScene.txt", 64, 64, 1)
    c, lc = cpu_render("sampleScene.txt", 64, 64, 1)
    check(g, c, lg, lc, "depth 1")
    g, lg, _ = gpu_render(pkg, "sampleScene.txt", 48, 32, 64, rr_start=3)
    c, lc = cpu_render("sampleScene.txt", 48, 32, 64, rr_start=3)
    check(g, c, lg, lc, "depth 64")


def test_empty_scene_is_black(pkg):
    sc = pkg.SceneFile(os.path.join(SCENES, "sampleScene.txt"))
    sc.set_resolution(32, 32)
    with pkg.Renderer(0) as r:
        r.set_options(depth=4)
        r.set_scene(sc.geoms, 0, sc.mats, sc.n_materials)
        r.set_camera(sc.camera)
        r.upload_image(np.ones((32, 32, 3), dtype=np.float32))
        r.render(1, 1)
        img = r.download_image()
        st = r.stats()
    assert (img == 0).all()
    assert [int(x) for x in st.live_in[:4]] == [1024, 0, 0, 0]


def test_axis_parallel_rays_hit_boxes(pkg):
    """Camera looking straight down an axis at an axis-aligned box: slab divisions by exactly 0."""
    geoms = (O.StaticGeom * 2)(O.make_geom(O.CUBE, 0, (0, 0, -5), (0, 0, 0), (2, 2, 2)),
                               O.make_geom(O.CUBE, 1, (0, 0, 3), (0, 0, 0), (20, 20, .1)))
    mats = (O.Material * 2)(O.make_material(color=(.5, .6, .7)), O.make_material(emittance=2.0))
    cam = O.make_camera(33, 33, (0, 0, 0), (0, 0, -1), (0, 1, 0), 20)
    ref, live = O.render(geoms, 2, mats, 2, cam, 3, iters=2)
    with pkg.Renderer(0) as r:
        r.set_options(depth=3)
        r.set_scene(C.cast(geoms, C.POINTER(pkg.StaticGeom)), 2, C.cast(mats, C.POINTER(pkg.Material)), 2)
        r.set_camera(pkg.CameraData.from_buffer_copy(cam))
        r.clear_image()
        r.render(1, 2)
        img = r.download_image()
        st = r.stats()
    check(img, ref, [int(x) for x in st.live_in[:3]], [int(x) for x in live], "axis-parallel")
    assert img.max() > 0


def test_pbo_matches_oracle(pkg):
    import torch
    W, H = 96, 64
    sc = pkg.SceneFile(os.path.join(SCENES, "sampleScene.txt"))
    sc.set_resolution(W, H)
    pbo = torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda:0")
    with pkg.Renderer(0) as r:
        r.set_options(depth=4)
        r.set_scene(sc.geoms, sc.n_objects, sc.mats, sc.n_materials)
        r.set_camera(sc.camera)
        r.clear_image()
        r.render(1, 3)
        r.send_image_to_pbo(pbo.data_ptr())
        r.synchronize()
        img = r.download_image()
    want = np.zeros((H * W, 4), dtype=np.uint8)
    O.lib().o_sendImageToPBO(want.ctypes.data, H * W, np.ascontiguousarray(img).ctypes.data)
    assert np.array_equal(pbo.cpu().numpy().reshape(-1, 4), want)


def test_bound_external_framebuffer_and_stream(pkg):
    """The context renders into caller-owned device memory on a caller-owned stream (torch = plumbing)."""
    import torch
    W, H = 80, 60
    sc = pkg.SceneFile(os.path.join(SCENES, "sampleScene.txt"))
    sc.set_resolution(W, H)
    fb = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda:0")
    stream = torch.cuda.Stream()
    with pkg.Renderer(0) as r:
        r.set_options(depth=4)
        r.set_scene(sc.geoms, sc.n_objects, sc.mats, sc.n_materials)
        r.set_camera(sc.camera)
        r.bind_image(fb.data_ptr())
        r.set_stream(stream.cuda_stream)
        r.render(1, 2)
        r.synchronize()
    c, _ = cpu_render("sampleScene.txt", W, H, 4, iters=2)
    assert float(np.abs(fb.cpu().numpy() - c).max()) <= TOL


def test_errors_are_reported_not_fatal(pkg):
    with pkg.Renderer(0) as r:
        with pytest.raises(pkg.PtError):
            r.render(1, 1)                       # no scene yet
        with pytest.raises(pkg.PtError):
            r.set_options(depth=0)
        sc = pkg.SceneFile(os.path.join(SCENES, "sampleScene.txt"))
        sc.geoms[0].materialid = 99
        with pytest.raises(pkg.PtError):
            r.set_scene(sc.geoms, sc.n_objects, sc.mats, sc.n_materials)
    assert pkg.lib().pt_create(999, C.byref(C.c_void_p())) != 0


# ---------------------------------------------------------------- full-size properties (oracle too slow here)
def test_full_size_properties_1080p(pkg):
    """BASELINE configs[1] shape (1920x1080, 8 bounces): determinism, conservation, tiling, sample rows vs oracle."""
    W, H, depth = 1920, 1080, 8
    a, la, st = gpu_render(pkg, "sampleScene_spec.txt", W, H, depth, iters=2)
    b, lb, _ = gpu_render(pkg, "sampleScene_spec.txt", W, H, depth, iters=2, workgroup=512, compaction=0)
    assert np.array_equal(a, b) and la == lb                      # order / launch-shape independent
    assert la[0] == 2 * W * H and all(x >= y for x, y in zip(la, la[1:]))
    assert np.isfinite(a).all() and a.min() >= 0
    top, lt, _ = gpu_render(pkg, "sampleScene_spec.txt", W, H, depth, iters=2, row_begin=0, row_end=540)
    bot, lb2, _ = gpu_render(pkg, "sampleScene_spec.txt", W, H, depth, iters=2, row_begin=540, row_end=1080)
    assert np.array_equal(np.concatenate([top, bot]), a)
    assert [x + y for x, y in zip(lt, lb2)] == la
    # oracle on a few full-width rows of the same frame (single paths, both iterations)
    sc = O.LoadedScene(os.path.join(SCENES, "sampleScene_spec.txt"))
    sc.set_resolution(W, H)
    opt = O.Options(depth, -1, 0, O.TRIG_POLY)
    L = O.lib()
    for y in (0, 333, 540, 1079):
        for x in range(0, W, 16):
            l1 = L.o_trace_path(sc.geoms, sc.n_objects, sc.mats, sc.n_materials, C.byref(sc.camera), C.byref(opt), x, y, 1, None)
            l2 = L.o_trace_path(sc.geoms, sc.n_objects, sc.mats, sc.n_materials, C.byref(sc.camera), C.byref(opt), x, y, 2, None)
            want = [np.float32((np.float32(np.float32(u * np.float32(1.0)) * np.float32(1.0) + np.float32(v))) / np.float32(2.0))
                    for u, v in zip(l1.tup(), l2.tup())]
            # running mean: it=1 -> (0*0 + L1)/1 = L1 ; it=2 -> (L1*1 + L2)/2
            assert np.allclose(a[y, x], want, atol=TOL, rtol=0), (x, y)


# ---------------------------------------------------------------- randomized scenes
def _random_scene(seed, n_prims):
    rng = np.random.default_rng(seed)
    mats = [O.make_material(color=rng.uniform(0.1, 1.0, 3)),
            O.make_material(color=rng.uniform(0.1, 1.0, 3)),
            O.make_material(color=(0.9, 0.9, 0.9), spec=rng.uniform(0.5, 1.0, 3), refl=1.0),
            O.make_material(color=(0, 0, 0), spec=(1, 1, 1), refr=1.0, ior=float(rng.uniform(1.2, 2.4))),
            O.make_material(color=(1, 1, 1), emittance=float(rng.uniform(2, 20)))]
    geoms = []
    # an enclosing room (one big cube seen from inside), a light, then random spheres / cubes, some overlapping
    geoms.append(O.make_geom(O.CUBE, 0, (0, 0, 0), rng.uniform(-3, 3, 3), (14, 12, 16)))
    geoms.append(O.make_geom(O.CUBE if rng.random() < 0.5 else O.SPHERE, 4, (0, 4.5, 0), rng.uniform(-3, 3, 3), (4, 0.6, 4)))
    for _ in range(n_prims - 2):
        kind = O.SPHERE if rng.random() < 0.5 else O.CUBE
        s = rng.uniform(0.3, 3.0, 3) if rng.random() < 0.5 else np.full(3, rng.uniform(0.3, 3.0))
        geoms.append(O.make_geom(kind, int(rng.integers(0, 4)), rng.uniform(-5, 5, 3), rng.uniform(-3.2, 3.2, 3), s))
    eye = rng.uniform(-2, 2, 3)
    view = rng.normal(size=3)
    view /= np.linalg.norm(view)
    up = np.cross(view, rng.normal(size=3))
    up /= np.linalg.norm(up)
    return geoms, mats, eye, view, up, float(rng.uniform(15, 40))


@pytest.mark.parametrize("seed", range(12))
def test_random_scenes_bit_exact(pkg, seed):
    """Random TRS primitives (overlapping, camera possibly inside objects), all material lobes, random cameras:
    every geometry path of the kernel must reproduce the oracle."""
    n_prims = [3, 6, 9, 17, 40, 70][seed % 6]
    geoms, mats, eye, view, up, fovy = _random_scene(1000 + seed, n_prims)
    W, H, depth, iters = 73, 41, 7, 2
    ga = (O.StaticGeom * len(geoms))(*geoms)
    ma = (O.Material * len(mats))(*mats)
    cam = O.make_camera(W, H, eye, view, up, fovy)
    ref, live = O.render(ga, len(geoms), ma, len(mats), cam, depth, iters=iters, rr_start=3, seed=seed)
    for geom_path in (1, 3, 4, 5, 6, 7) if n_prims <= 32 else (1, 2, 4, 5, 6, 7):
        with pkg.Renderer(0) as r:
            r.set_options(depth=depth, rr_start=3, seed=seed, geom_path=geom_path)
            r.set_scene(C.cast(ga, C.POINTER(pkg.StaticGeom)), len(geoms), C.cast(ma, C.POINTER(pkg.Material)), len(mats))
            r.set_camera(pkg.CameraData.from_buffer_copy(cam))
            r.clear_image()
            r.render(1, iters)
            img = r.download_image()
            st = r.stats()
        check(img, ref, [int(x) for x in st.live_in[:depth]], [int(x) for x in live], f"random scene {seed} geom_path={geom_path}")


@pytest.mark.parametrize("seed", range(16))
def test_camera_span_table_is_conservative(pkg, seed):
    """Pair path, camera rays: the host's table of the primitives each 64-pixel span can see (built when a tile's pixel
    count is a multiple of 64 and there is no lens) must only drop pairs that would miss -- random cameras (inside
    objects, narrow and very wide fields of view, rolled), random TRS primitives, whole frames, bands and strip tiles."""
    rng = np.random.default_rng(4200 + seed)
    n_prims = [3, 9, 20, 32][seed % 4] if seed % 2 else [60, 150, 24, 300][(seed // 2) % 4]
    geoms, mats, eye, view, up, _ = _random_scene(7000 + seed, n_prims)
    W, H = [(64, 48), (96, 64), (128, 40), (192, 8), (32, 30)][seed % 5]
    if (W * H) % 64:
        H = (H + 63) // 64 * 64
    fovy = float([8.0, 25.0, 42.0, 60.0][seed % 4] * rng.uniform(0.8, 1.1))
    depth, iters = 4, 2
    ga = (O.StaticGeom * len(geoms))(*geoms)
    ma = (O.Material * len(mats))(*mats)
    cam = O.make_camera(W, H, eye * (3.0 if seed % 3 == 0 else 1.0), view, up, fovy)      # every third camera outside the cluster
    ref, live = O.render(ga, len(geoms), ma, len(mats), cam, depth, iters=iters, seed=seed)
    tiles = [dict(), dict(geom_path=5)]
    if seed % 2 == 0:
        tiles = [dict(), dict(geom_path=7), dict(geom_path=8)]      # the batched walks take per-span primitive lists
    if H % 16 == 0:
        gp = 5 if seed % 2 else 7
        tiles.append(dict(geom_path=gp, strip_rows=8, strip_world=2, strip_rank=seed % 2))
        tiles.append(dict(geom_path=gp, row_begin=H // 2, row_end=H))
    for opts in tiles:
        with pkg.Renderer(0) as r:
            r.set_options(depth=depth, seed=seed, **opts)
            r.set_scene(C.cast(ga, C.POINTER(pkg.StaticGeom)), len(geoms), C.cast(ma, C.POINTER(pkg.Material)), len(mats))
            r.set_camera(pkg.CameraData.from_buffer_copy(cam))
            r.clear_image()
            r.render(1, iters)
            img = r.download_image()
            st = r.stats()
        if "strip_rows" in opts:
            from project3_pathtracer_amd import sharding
            rows = sharding.strip_global_rows(H, 2, opts["strip_rank"], 8)
            assert np.array_equal(img.view(np.uint32), ref[rows].view(np.uint32)), f"span table, strips, seed {seed}"
        elif "row_begin" in opts:
            assert np.array_equal(img.view(np.uint32), ref[H // 2:].view(np.uint32)), f"span table, band, seed {seed}"
        else:
            check(img, ref, [int(x) for x in st.live_in[:depth]], [int(x) for x in live], f"span table seed {seed} {opts}")


# ---------------------------------------------------------------- several contexts behind one handle (pt_multi_*)
@pytest.mark.parametrize("ndev", [1, 2, 3])
def test_multi_device_handle_matches_single_context(pkg, ndev):
    """pt_multi_*: row bands on `ndev` contexts (all on device 0 here; one per GPU on a real node), gathered to the
    host and to one device buffer: identical to the single-context frame, counters add up."""
    import torch
    L = pkg.lib()
    W, H, depth, iters = 144, 81, 5, 3
    sc = pkg.SceneFile(os.path.join(SCENES, "sampleScene_spec.txt"))
    sc.set_resolution(W, H)
    ref, lr, _ = gpu_render(pkg, "sampleScene_spec.txt", W, H, depth, iters=iters)
    devs = (C.c_int * ndev)(*([0] * ndev))
    m = C.c_void_p()
    assert L.pt_multi_create(devs, ndev, C.byref(m)) == 0
    try:
        o = pkg.Options()
        L.pt_default_options(C.byref(o))
        o.depth = depth
        assert L.pt_multi_set_options(m, C.byref(o)) == 0
        assert L.pt_multi_set_scene(m, sc.geoms, sc.n_objects, sc.mats, sc.n_materials) == 0
        assert L.pt_multi_set_camera(m, C.byref(sc.camera)) == 0
        assert L.pt_multi_count(m) == ndev
        rows = []
        for k in range(ndev):
            r0, r1 = C.c_int(), C.c_int()
            assert L.pt_multi_band(m, k, C.byref(r0), C.byref(r1)) == 0
            rows.append((r0.value, r1.value))
        assert rows[0][0] == 0 and rows[-1][1] == H and all(a[1] == b[0] for a, b in zip(rows, rows[1:]))
        assert L.pt_multi_clear_image(m) == 0
        assert L.pt_multi_render(m, 1, iters) == 0
        assert L.pt_multi_synchronize(m) == 0
        host = np.zeros((H, W, 3), dtype=np.float32)
        assert L.pt_multi_download_image(m, host.ctypes.data) == 0
        assert np.array_equal(host, ref)
        dev = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda:0")
        assert L.pt_multi_gather_to_device(m, dev.data_ptr(), 0) == 0
        torch.cuda.synchronize()
        assert np.array_equal(dev.cpu().numpy(), ref)
        st = pkg.Stats()
        assert L.pt_multi_get_stats(m, C.byref(st)) == 0
        assert [int(x) for x in st.live_in[:depth]] == lr and st.iterations == iters
        # resume from a host frame: upload, render the next iterations, same as rendering them in one go
        assert L.pt_multi_upload_image(m, host.ctypes.data) == 0
        assert L.pt_multi_render(m, iters + 1, 2) == 0
        assert L.pt_multi_download_image(m, host.ctypes.data) == 0
        ref5, _, _ = gpu_render(pkg, "sampleScene_spec.txt", W, H, depth, iters=iters + 2)
        assert np.array_equal(host, ref5)
    finally:
        L.pt_multi_destroy(m)


# ---------------------------------------------------------------- direct lighting
def cpu_render_dl(scene, w, h, depth, iters=1, rotat=0, rr_start=-1):
    sc = O.LoadedScene(os.path.join(SCENES, scene), rotat)
    sc.set_resolution(w, h)
    sh = []
    img, live = O.render(sc.geoms, sc.n_objects, sc.mats, sc.n_materials, sc.camera, depth, iters=iters,
                         rr_start=rr_start, direct_light=1, shadow_out=sh)
    return img, [int(x) for x in live], sh[0]


@pytest.mark.parametrize("scene,rotat,depth,iters", [("sampleScene_spec.txt", 0, 5, 3), ("sampleScene.txt", 1, 4, 2),
                                                     ("cornell_glass.txt", 1, 8, 2), ("sampleScene.txt", 1, 1, 2)])
def test_direct_lighting_matches_oracle(pkg, scene, rotat, depth, iters):
    """pt_options.direct_light: explicit light sampling (the reference's float-seeded samplers) + one shadow ray per
    diffuse vertex.  Image, live-ray counts and the number of shadow rays equal the oracle's."""
    W, H = 96, 64
    cpu, lc, shadows = cpu_render_dl(scene, W, H, depth, iters=iters, rotat=rotat)
    gpu, lg, st = gpu_render(pkg, scene, W, H, depth, iters=iters, rotat=rotat, direct_light=1)
    check(gpu, cpu, lg, lc, f"direct lighting {scene} depth {depth}")
    assert int(st.shadow_rays) == shadows and shadows > 0
    plain, _, st0 = gpu_render(pkg, scene, W, H, depth, iters=iters, rotat=rotat)
    assert int(st0.shadow_rays) == 0 and not np.array_equal(plain, gpu)


@pytest.mark.parametrize("geom_path", [1, 2, 3, 4, 5, 6, 7])
def test_direct_lighting_on_every_geometry_path(pkg, geom_path):
    W, H, depth = 80, 60, 4
    cpu, lc, shadows = cpu_render_dl("sampleScene_spec.txt", W, H, depth, iters=2, rotat=1)
    gpu, lg, st = gpu_render(pkg, "sampleScene_spec.txt", W, H, depth, iters=2, rotat=1, direct_light=1, geom_path=geom_path)
    check(gpu, cpu, lg, lc, f"direct lighting geom_path {geom_path}")
    assert int(st.shadow_rays) == shadows


def test_direct_lighting_batching_rr_and_large_scene(pkg):
    W, H = 64, 48
    cpu, lc, shadows = cpu_render_dl("cloud256.txt", W, H, 6, iters=5, rotat=1, rr_start=2)
    for batch in (1, 3, 8, 16):
        gpu, lg, st = gpu_render(pkg, "cloud256.txt", W, H, 6, iters=5, rotat=1, direct_light=1, rr_start=2, batch=batch)
        check(gpu, cpu, lg, lc, f"direct lighting cloud256 batch {batch}")
        assert int(st.shadow_rays) == shadows


@pytest.mark.parametrize("geom_path", [0, 1, 7, 8])
def test_mesh_lights_match_oracle(pkg, geom_path):
    """scenes/mesh_light.txt: mesh_cornell with the icosphere made of the light's material -- an emissive MESH geom is one
    entry of the light table (area-weighted triangle pick + uniform point, the shadow ray must reach THAT triangle) beside
    the cube light.  Image, live-ray and shadow-ray counts equal the oracle's on the scalar loop and the batched walks."""
    W, H, depth, iters = 96, 80, 5, 3
    sc = pkg.SceneFile(os.path.join(SCENES, "mesh_light.txt"), 1)
    sc.set_resolution(W, H)
    with pkg.Renderer(0) as r:
        r.set_options(depth=depth, direct_light=1, geom_path=geom_path, rr_start=2)
        r.set_scene(sc.geoms, sc.n_objects, sc.mats, sc.n_materials)
        r.set_meshes(sc.meshes)
        r.set_camera(sc.camera)
        r.clear_image()
        r.render(1, iters)
        g = r.download_image()
        st = r.stats()
    osc = O.LoadedScene(os.path.join(SCENES, "mesh_light.txt"), 1)
    osc.set_resolution(W, H)
    sh = []
    c, lc = O.render(osc.geoms, osc.n_objects, osc.mats, osc.n_materials, osc.camera, depth, iters=iters, meshes=osc.meshes,
                     direct_light=1, rr_start=2, shadow_out=sh)
    check(g, c, [int(x) for x in st.live_in[:depth]], [int(x) for x in lc], f"mesh lights geom_path={geom_path}")
    assert int(st.shadow_rays) == sh[0] > 0
    # the mesh light matters: with the icosphere diffuse (mesh_cornell.txt) the picture is another one
    base, _ = O.render(*(lambda s: (s.geoms, s.n_objects, s.mats, s.n_materials, s.camera))(_loaded("mesh_cornell.txt", W, H)), depth,
                       iters=iters, meshes=osc.meshes, direct_light=1, rr_start=2)
    assert not np.array_equal(base, c)


def _loaded(scene, w, h, rotat=1):
    s = O.LoadedScene(os.path.join(SCENES, scene), rotat)
    s.set_resolution(w, h)
    return s


@pytest.mark.parametrize("geom_path", [0, 1, 2, 3, 4, 5, 6, 7, 8])
def test_emitters_beyond_the_light_table(pkg, geom_path):
    """17 emissive spheres: the table takes 16, the 17th is reached by chance only and such a hit still counts after a
    diffuse vertex (Prim::area marks what the table covers).  Every geometry path == oracle."""
    mats = [O.make_material(color=(0.8, 0.8, 0.8)), O.make_material(color=(1, 1, 1), emittance=3.0), O.make_material(color=(0.9, 0.9, 0.9), refl=1.0)]
    geoms = [O.make_geom(O.CUBE, 0, (0, -0.05, 0), (0, 0, 0), (14, 0.1, 14)), O.make_geom(O.CUBE, 2, (0, 2, -4), (0, 0, 0), (14, 6, 0.1))]
    for k in range(17):
        geoms.append(O.make_geom(O.SPHERE, 1, (-4.0 + 0.5 * k, 1.5 + 0.1 * (k % 3), -1.0 + 0.3 * (k % 4)), (0, 0, 0), (0.4, 0.4, 0.4)))
    geoms[-1] = O.make_geom(O.SPHERE, 1, (0.0, 4.0, 1.0), (0, 0, 0), (2.5, 2.5, 2.5))
    ga = (O.StaticGeom * len(geoms))(*geoms)
    ma = (O.Material * len(mats))(*mats)
    W, H, depth = 64, 40, 4
    cam = O.make_camera(W, H, (0, 2.5, 10), (0, -0.15, -1), (0, 1, 0), 25)
    sh = []
    ref, live = O.render(ga, len(geoms), ma, len(mats), cam, depth, iters=2, direct_light=1, shadow_out=sh)
    with pkg.Renderer(0) as r:
        r.set_options(depth=depth, direct_light=1, geom_path=geom_path)
        r.set_scene(C.cast(ga, C.POINTER(pkg.StaticGeom)), len(geoms), C.cast(ma, C.POINTER(pkg.Material)), len(mats))
        r.set_camera(pkg.CameraData.from_buffer_copy(cam))
        r.clear_image()
        r.render(1, 2)
        img = r.download_image()
        st = r.stats()
    check(img, ref, [int(x) for x in st.live_in[:depth]], [int(x) for x in live], f"17 emitters geom_path={geom_path}")
    assert int(st.shadow_rays) == sh[0] > 0


def test_direct_lighting_needs_compaction_1(pkg):
    with pkg.Renderer(0) as r:
        with pytest.raises(pkg.PtError):
            r.set_options(direct_light=1, compaction=2)
        with pytest.raises(pkg.PtError):
            r.set_options(direct_light=2)


# ---------------------------------------------------------------- absorption inside refractive objects
def test_absorption_in_glass_matches_oracle(pkg):
    """pt_options.absorption: Beer-Lambert with the material's ABSCOEFF (.02 5.1 5.7 on the bundled glass) over segments
    that end on the inner side of a refractive surface -- calculateTransmission with a deterministic exp."""
    W, H, depth, iters = 96, 64, 8, 3
    sc = O.LoadedScene(os.path.join(SCENES, "cornell_glass.txt"), 1)
    sc.set_resolution(W, H)
    cpu, lc = O.render(sc.geoms, sc.n_objects, sc.mats, sc.n_materials, sc.camera, depth, iters=iters, absorption=1)
    gpu, lg, _ = gpu_render(pkg, "cornell_glass.txt", W, H, depth, iters=iters, rotat=1, absorption=1)
    check(gpu, cpu, lg, [int(x) for x in lc], "absorption in glass")
    plain, lp, _ = gpu_render(pkg, "cornell_glass.txt", W, H, depth, iters=iters, rotat=1)
    assert lp == lg and not np.array_equal(plain, gpu)          # same paths, tinted throughput
    assert gpu.sum() < plain.sum()
    # together with direct lighting and Russian roulette (throughput now steers path lengths)
    sh = []
    cpu2, lc2 = O.render(sc.geoms, sc.n_objects, sc.mats, sc.n_materials, sc.camera, depth, iters=iters, absorption=1,
                         direct_light=1, rr_start=1, shadow_out=sh)
    gpu2, lg2, st2 = gpu_render(pkg, "cornell_glass.txt", W, H, depth, iters=iters, rotat=1, absorption=1, direct_light=1, rr_start=1)
    check(gpu2, cpu2, lg2, [int(x) for x in lc2], "absorption + direct lighting + RR")
    assert int(st2.shadow_rays) == sh[0]


# ---------------------------------------------------------------- interleaved-strip tiles
@pytest.mark.parametrize("world,strip,h", [(2, 8, 90), (3, 8, 90), (4, 4, 37), (8, 8, 128), (3, 5, 16)])
def test_strip_tiles_reassemble_to_full_frame(pkg, world, strip, h):
    """pt_options.strip_*: strip k of `strip` rows belongs to rank k % world; every rank renders its strips packed.
    RNG is keyed on the global pixel, so putting the rows back gives the single-context frame bit for bit -- including a
    short last strip and direct lighting (which reads and writes per-pixel state by local index)."""
    from project3_pathtracer_amd import sharding
    W, depth = 72, 4
    full, lf, _ = gpu_render(pkg, "sampleScene_spec.txt", W, h, depth, iters=3, direct_light=1)
    frame = np.full_like(full, -1.0)
    live = [0] * depth
    for rank in range(world):
        t, lt, _ = gpu_render(pkg, "sampleScene_spec.txt", W, h, depth, iters=3, direct_light=1, strip_rows=strip,
                              strip_world=world, strip_rank=rank)
        rows = sharding.strip_global_rows(h, world, rank, strip)
        assert t.shape == (len(rows), W, 3) == (pkg.lib().pt_strip_local_rows(h, strip, world, rank), W, 3)
        frame[rows] = t
        live = [a + b for a, b in zip(live, lt)]
    assert np.array_equal(frame, full) and live == lf


def test_strip_options_are_validated(pkg):
    with pkg.Renderer(0) as r:
        for bad in (dict(strip_rows=8, strip_world=0), dict(strip_rows=8, strip_world=2, strip_rank=2),
                    dict(strip_rows=8, strip_world=2, strip_rank=0, row_begin=0, row_end=4), dict(strip_rows=-1)):
            with pytest.raises(pkg.PtError):
                r.set_options(**bad)
            r.set_options(strip_rows=0, strip_world=0, strip_rank=0, row_begin=0, row_end=0)


@pytest.mark.parametrize("ndev,strip", [(2, 8), (3, 4)])
def test_multi_device_handle_with_strips(pkg, ndev, strip):
    """pt_multi_set_strips: the handle shards by interleaved strips; host download, host upload (resume) and the
    device gather put every row at its place."""
    import torch
    L = pkg.lib()
    W, H, depth, iters = 100, 53, 4, 2
    sc = pkg.SceneFile(os.path.join(SCENES, "sampleScene_spec.txt"))
    sc.set_resolution(W, H)
    ref, lr, _ = gpu_render(pkg, "sampleScene_spec.txt", W, H, depth, iters=iters)
    ref4, _, _ = gpu_render(pkg, "sampleScene_spec.txt", W, H, depth, iters=iters + 2)
    devs = (C.c_int * ndev)(*([0] * ndev))
    m = C.c_void_p()
    assert L.pt_multi_create(devs, ndev, C.byref(m)) == 0
    try:
        o = pkg.Options()
        L.pt_default_options(C.byref(o))
        o.depth = depth
        assert L.pt_multi_set_options(m, C.byref(o)) == 0
        assert L.pt_multi_set_strips(m, strip) == 0
        assert L.pt_multi_set_scene(m, sc.geoms, sc.n_objects, sc.mats, sc.n_materials) == 0
        assert L.pt_multi_set_camera(m, C.byref(sc.camera)) == 0
        assert L.pt_multi_clear_image(m) == 0
        assert L.pt_multi_render(m, 1, iters) == 0
        host = np.zeros((H, W, 3), dtype=np.float32)
        assert L.pt_multi_download_image(m, host.ctypes.data) == 0
        assert np.array_equal(host, ref)
        dev = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda:0")
        assert L.pt_multi_gather_to_device(m, dev.data_ptr(), 0) == 0
        torch.cuda.synchronize()
        assert np.array_equal(dev.cpu().numpy(), ref)
        st = pkg.Stats()
        assert L.pt_multi_get_stats(m, C.byref(st)) == 0
        assert [int(x) for x in st.live_in[:depth]] == lr
        assert L.pt_multi_upload_image(m, host.ctypes.data) == 0
        assert L.pt_multi_render(m, iters + 1, 2) == 0
        assert L.pt_multi_download_image(m, host.ctypes.data) == 0
        assert np.array_equal(host, ref4)
    finally:
        L.pt_multi_destroy(m)


@pytest.mark.parametrize("strip", [0, 8])
def test_multi_device_gather_is_enqueued_not_awaited(pkg, strip):
    """pt_multi_gather_to_device_async on a 3840x2160 frame cut into 8 tiles (bands, and 8-row strips as the shim uses them):
    the call orders each device's copy stream behind its render stream with an event and returns when the copies are ENQUEUED
    -- while the render it depends on is still running -- and pt_multi_synchronize is the one join.  The frame that arrives is the
    single-context frame; the handle reports peer access for every pair (same device: direct) and the gather's host times."""
    import time
    import torch
    L = pkg.lib()
    W, H, depth, iters, ndev = 3840, 2160, 3, 16, 8
    sc = pkg.SceneFile(os.path.join(SCENES, "sampleScene_spec.txt"))
    sc.set_resolution(W, H)
    ref, _, _ = gpu_render(pkg, "sampleScene_spec.txt", W, H, depth, iters=iters)
    devs = (C.c_int * ndev)(*([0] * ndev))
    m = C.c_void_p()
    assert L.pt_multi_create(devs, ndev, C.byref(m)) == 0
    try:
        o = pkg.Options()
        L.pt_default_options(C.byref(o))
        o.depth = depth
        assert L.pt_multi_set_options(m, C.byref(o)) == 0
        assert L.pt_multi_set_strips(m, strip) == 0
        assert L.pt_multi_set_scene(m, sc.geoms, sc.n_objects, sc.mats, sc.n_materials) == 0
        assert L.pt_multi_set_camera(m, C.byref(sc.camera)) == 0
        assert L.pt_multi_clear_image(m) == 0
        assert L.pt_multi_render(m, 1, 1) == 0            # (graphs captured, pools allocated)
        assert L.pt_multi_synchronize(m) == 0
        assert L.pt_multi_clear_image(m) == 0
        dev = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda:0")
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        assert L.pt_multi_render(m, 1, iters) == 0         # ~ 10 ms of GPU work, enqueued in well under a millisecond
        assert L.pt_multi_gather_to_device_async(m, dev.data_ptr(), 0) == 0
        t_enq = time.perf_counter() - t0
        assert L.pt_multi_synchronize(m) == 0
        t_all = time.perf_counter() - t0
        enq, tot = C.c_double(), C.c_double()
        assert L.pt_multi_gather_times(m, C.byref(enq), C.byref(tot)) == 0
        print(f"strips={strip}: render + gather enqueued in {t_enq * 1e3:.2f} ms, done after {t_all * 1e3:.2f} ms; gather enqueue {enq.value:.3f} ms, joined after {tot.value:.2f} ms")
        assert np.array_equal(dev.cpu().numpy().view(np.uint32), ref.view(np.uint32))
        # the gather call came back long before the render it waits for had finished: it waited for nothing on the host
        assert t_enq < 0.5 * t_all and enq.value < 0.5 * tot.value, (t_enq, t_all, enq.value, tot.value)
        d = C.c_int(-1)
        assert L.pt_multi_peer_access(m, 0, 0, C.byref(d)) == 0 and d.value == 1
        # a second gather while nothing renders, the plain (joining) form
        dev.zero_()
        torch.cuda.synchronize()
        assert L.pt_multi_gather_to_device(m, dev.data_ptr(), 0) == 0
        assert np.array_equal(dev.cpu().numpy().view(np.uint32), ref.view(np.uint32))
    finally:
        L.pt_multi_destroy(m)


@pytest.mark.parametrize("n_prims,paths", [(1500, (0, 4, 8)), (6000, (0, 8))])
def test_very_large_primitive_lists(pkg, n_prims, paths):
    """1 500 primitives: the binary hierarchy needs more than 64 KiB of LDS per workgroup (asked for explicitly) and the
    4-wide one no longer fits beside the per-wave queues: the library reads its nodes through L1/L2 (geom_path 8); 6 000
    likewise.  Same bits.  (Round 2 found a real bug here: lanes holding a leaf entry read "node" <primitive index> --
    harmless garbage from LDS, a page fault from global memory whenever the pages behind the node array were unmapped.)"""
    geoms, mats, eye, view, up, fovy = _random_scene(4242, n_prims)
    W, H, depth = 48, 32, 3
    ga = (O.StaticGeom * len(geoms))(*geoms)
    ma = (O.Material * len(mats))(*mats)
    cam = O.make_camera(W, H, eye, view, up, fovy)
    ref, live = O.render(ga, n_prims, ma, len(mats), cam, depth, iters=1)
    for gp in paths:
        with pkg.Renderer(0) as r:
            r.set_options(depth=depth, geom_path=gp)
            r.set_scene(C.cast(ga, C.POINTER(pkg.StaticGeom)), n_prims, C.cast(ma, C.POINTER(pkg.Material)), len(mats))
            r.set_camera(pkg.CameraData.from_buffer_copy(cam))
            r.clear_image()
            r.render(1, 1)
            img = r.download_image()
            st = r.stats()
        check(img, ref, [int(x) for x in st.live_in[:depth]], [int(x) for x in live], f"{n_prims} primitives geom_path={gp}")


# ---------------------------------------------------------------- subsurface scattering
@pytest.mark.parametrize("geom_path", [0, 1, 2, 3, 4, 5, 6, 7])
def test_subsurface_scattering_matches_oracle(pkg, geom_path):
    """pt_options.scatter: random walk inside the SCATTER media of sss_blobs.txt (an index-matched sphere, a scattering
    glass cube): free flights, isotropic re-direction, absorption and the three extra draws per inside-segment follow
    the oracle bit for bit on every geometry path."""
    g, lg, _ = gpu_render(pkg, "sss_blobs.txt", 96, 96, 12, iters=3, rotat=1, scatter=1, geom_path=geom_path)
    c, lc = cpu_render("sss_blobs.txt", 96, 96, 12, iters=3, rotat=1, scatter=1)
    check(g, c, lg, lc, f"subsurface scattering geom_path={geom_path}")


def test_subsurface_scattering_with_everything_else(pkg):
    """... together with Russian roulette, absorption (refractive non-scatter materials only), direct lighting (media
    are not diffuse vertices) and iteration batching."""
    kw = dict(rr_start=2, absorption=1, direct_light=1)
    g, lg, st = gpu_render(pkg, "sss_blobs.txt", 80, 60, 10, iters=5, rotat=1, scatter=1, batch=4, **kw)
    sh = []
    c, lc = cpu_render("sss_blobs.txt", 80, 60, 10, iters=5, rotat=1, scatter=1, shadow_out=sh, **kw)
    check(g, c, lg, lc, "scatter + rr + absorption + direct light")
    assert int(st.shadow_rays) == sh[0]
    off, lo, _ = gpu_render(pkg, "sss_blobs.txt", 80, 60, 10, iters=5, rotat=1, batch=4, **kw)
    assert not np.array_equal(off, g)
    with pkg.Renderer(0) as r:
        with pytest.raises(pkg.PtError):
            r.set_options(scatter=2)


# ---------------------------------------------------------------- triangle meshes (MESH objects)
def mesh_render(pkg, w, h, depth, iters, **opts):
    sc = pkg.SceneFile(os.path.join(SCENES, "mesh_cornell.txt"), 1)
    sc.set_resolution(w, h)
    with pkg.Renderer(0) as r:
        r.set_options(depth=depth, **opts)
        r.set_scene(sc.geoms, sc.n_objects, sc.mats, sc.n_materials)
        r.set_meshes(sc.meshes)
        r.set_camera(sc.camera)
        r.clear_image()
        r.render(1, iters)
        img = r.download_image()
        st = r.stats()
    return img, [int(x) for x in st.live_in[:depth]], st


@pytest.mark.parametrize("geom_path", [0, 1, 7, 8])
def test_triangle_meshes_match_oracle(pkg, geom_path):
    """mesh_cornell.txt: 350 triangles from two .obj files (a diffuse icosphere, a glass gem made of polygons) beside a
    sphere and a cube; product loader -> pt_set_meshes -> triangle records in the batched walks / the scalar loop ==
    oracle loader -> o_render_ex, bit for bit."""
    g, lg, _ = mesh_render(pkg, 112, 96, 9, 3, geom_path=geom_path)
    osc = O.LoadedScene(os.path.join(SCENES, "mesh_cornell.txt"), 1)
    osc.set_resolution(112, 96)
    c, lc = O.render(osc.geoms, osc.n_objects, osc.mats, osc.n_materials, osc.camera, 9, iters=3, meshes=osc.meshes)
    check(g, c, lg, [int(x) for x in lc], f"triangle meshes geom_path={geom_path}")
    assert sum(lg) > 0


@pytest.mark.parametrize("geom_path", [0, 1, 7, 8])
def test_scene_spanning_triangles(pkg, geom_path, tmp_path):
    """A mesh whose two triangles span the room: they join the scene-spanning primitives that the batched walks test
    before the hierarchy (that loop once skipped every record that was not a sphere or a cube), and they are on most
    spans' camera-ray lists."""
    import shutil
    shutil.copytree(os.path.join(SCENES, "meshes"), tmp_path / "meshes")
    (tmp_path / "meshes" / "quad.obj").write_text("v -0.5 0 -0.5\nv 0.5 0 -0.5\nv 0.5 0 0.5\nv -0.5 0 0.5\nf 1 2 3 4\n")
    text = open(os.path.join(SCENES, "mesh_cornell.txt")).read().rstrip("\n")
    text += "\n\nOBJECT 10\nmeshes/quad.obj\nmaterial 1\nframe 0\nTRANS 0 3.5 -1\nROTAT 25 0 12\nSCALE 9 9 9\n"
    path = tmp_path / "big_quad.txt"
    path.write_text(text)
    W, H, depth, iters = 128, 96, 6, 2
    sc = pkg.SceneFile(str(path), 1)
    sc.set_resolution(W, H)
    with pkg.Renderer(0) as r:
        r.set_options(depth=depth, geom_path=geom_path)
        r.set_scene(sc.geoms, sc.n_objects, sc.mats, sc.n_materials)
        r.set_meshes(sc.meshes)
        r.set_camera(sc.camera)
        r.clear_image()
        r.render(1, iters)
        g = r.download_image()
        st = r.stats()
    osc = O.LoadedScene(str(path), 1)
    osc.set_resolution(W, H)
    c, lc = O.render(osc.geoms, osc.n_objects, osc.mats, osc.n_materials, osc.camera, depth, iters=iters, meshes=osc.meshes)
    check(g, c, [int(x) for x in st.live_in[:depth]], [int(x) for x in lc], f"scene-spanning triangles geom_path={geom_path}")
    base, _ = O.render(osc.geoms, osc.n_objects - 1, osc.mats, osc.n_materials, osc.camera, depth, iters=iters,
                       meshes={i: m for i, m in osc.meshes.items() if i < osc.n_objects - 1})
    assert not np.array_equal(base, c)                    # the quad is in the picture


def test_large_mesh_configures_in_seconds(pkg, tmp_path):
    """A 131 072-triangle height field: the hierarchy build is O(n log n) (binned surface-area splits above 4 096
    primitives per node; the full sweep used to copy the index vector at every cost improvement: minutes at this size),
    the 4-wide node copy no longer fits the LDS (geom_path 8 by the library's own choice), and sample pixels agree with
    the oracle's brute-force loop."""
    import time
    n = 256
    xs = np.linspace(-0.5, 0.5, n + 1, dtype=np.float32)
    gx, gz = np.meshgrid(xs, xs, indexing="ij")
    gy = (0.05 * np.sin(9.0 * gx) * np.cos(7.0 * gz)).astype(np.float32)
    P = np.stack([gx, gy, gz], axis=-1)
    a, b, c, d = P[:-1, :-1], P[1:, :-1], P[1:, 1:], P[:-1, 1:]
    tris = np.concatenate([np.stack([a, c, b], axis=2), np.stack([a, d, c], axis=2)], axis=0).reshape(-1, 9).astype(np.float32)
    assert tris.shape[0] == 2 * n * n
    text = open(os.path.join(SCENES, "mesh_cornell.txt")).read().rstrip("\n")
    (tmp_path / "meshes").mkdir()
    for f in os.listdir(os.path.join(SCENES, "meshes")):
        (tmp_path / "meshes" / f).write_bytes(open(os.path.join(SCENES, "meshes", f), "rb").read())
    (tmp_path / "meshes" / "empty.obj").write_text("v 0 0 0\n")
    text += "\n\nOBJECT 10\nmeshes/empty.obj\nmaterial 1\nframe 0\nTRANS 0 2.5 0\nROTAT 0 0 0\nSCALE 8 8 8\n"
    path = tmp_path / "field.txt"
    path.write_text(text)
    W, H, depth = 64, 48, 3
    sc = pkg.SceneFile(str(path), 1)
    sc.set_resolution(W, H)
    meshes = dict(sc.meshes)
    meshes[sc.n_objects - 1] = tris
    with pkg.Renderer(0) as r:
        r.set_options(depth=depth)
        r.set_scene(sc.geoms, sc.n_objects, sc.mats, sc.n_materials)
        r.set_meshes(meshes)
        r.set_camera(sc.camera)
        t0 = time.perf_counter()
        r.clear_image()                                  # configures: hierarchy build, camera-ray lists
        dt = time.perf_counter() - t0
        r.render(1, 1)
        g = r.download_image()
        st = r.stats()
    print(f"configure with {tris.shape[0]} triangles: {dt:.2f} s")
    assert dt < 20.0
    osc = O.LoadedScene(str(path), 1)
    osc.set_resolution(W, H)
    om = dict(osc.meshes)
    om[osc.n_objects - 1] = tris
    opt = O.Options(depth, -1, 0, O.TRIG_POLY)
    ex, keep = O.make_extras(om, None, None, osc.n_objects)
    L = O.lib()
    rng = np.random.default_rng(5)
    for _ in range(96):                                  # the oracle tests every triangle: sample pixels
        x, y = int(rng.integers(W)), int(rng.integers(H))
        l = np.array(L.o_trace_path_ex(osc.geoms, osc.n_objects, osc.mats, osc.n_materials, C.byref(osc.camera), C.byref(opt),
                                       C.byref(ex), x, y, 1, None).tup(), np.float32)
        assert np.array_equal(g[y, x].view(np.uint32), l.view(np.uint32)), f"pixel ({x},{y}): {g[y, x]} vs {l}"
    del keep
    assert int(st.live_in[0]) == W * H


def test_contexts_do_not_leak_device_memory(pkg):
    """create / configure / render / destroy in a loop (the batched walk with camera-ray lists, motion blur's child
    contexts, two launch sequences): the device's free memory comes back (pt_destroy once forgot the span tables)."""
    import torch
    sc = pkg.SceneFile(os.path.join(SCENES, "cloud256.txt"), 1)
    sc.set_resolution(128, 64)
    nxt = pkg.SceneFile(os.path.join(SCENES, "cloud256.txt"), 1)

    def once(motion):
        with pkg.Renderer(0) as r:
            r.set_options(depth=3, batch=2)
            r.set_scene(sc.geoms, sc.n_objects, sc.mats, sc.n_materials)
            r.set_camera(sc.camera)
            if motion:
                r.set_motion(nxt.geoms, None, 3, 1)
            r.clear_image()
            r.render(1, 5)
            r.synchronize()
    once(True)
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info(0)[0]
    for k in range(12):
        once(k % 3 == 0)
    torch.cuda.synchronize()
    free1 = torch.cuda.mem_get_info(0)[0]
    assert free0 - free1 < (8 << 20), f"{(free0 - free1) / 2**20:.1f} MiB of device memory lost over 12 contexts"


def test_triangle_meshes_with_options_and_errors(pkg):
    kw = dict(rr_start=2, direct_light=1, absorption=1, batch=4)
    g, lg, st = mesh_render(pkg, 80, 60, 8, 5, **kw)
    osc = O.LoadedScene(os.path.join(SCENES, "mesh_cornell.txt"), 1)
    osc.set_resolution(80, 60)
    sh = []
    c, lc = O.render(osc.geoms, osc.n_objects, osc.mats, osc.n_materials, osc.camera, 8, iters=5, meshes=osc.meshes,
                     rr_start=2, direct_light=1, absorption=1, shadow_out=sh)
    check(g, c, lg, [int(x) for x in lc], "meshes + rr + direct light + absorption")
    assert int(st.shadow_rays) == sh[0]
    sc = pkg.SceneFile(os.path.join(SCENES, "mesh_cornell.txt"), 1)
    with pkg.Renderer(0) as r:
        with pytest.raises(pkg.PtError):
            r.set_meshes(sc.meshes)                       # no scene yet
        r.set_scene(sc.geoms, sc.n_objects, sc.mats, sc.n_materials)
        with pytest.raises(pkg.PtError):
            r.set_meshes({0: sc.meshes[5]})               # geom 0 is a cube
        r.set_meshes(sc.meshes)
        r.set_camera(sc.camera)
        r.set_options(depth=4, geom_path=5)
        with pytest.raises(pkg.PtError):
            r.render(1, 1)                                # the pair queue does not know triangles
        r.set_options(depth=4, geom_path=0)
        r.set_scene(sc.geoms, sc.n_objects, sc.mats, sc.n_materials)      # drops the meshes again
        r.clear_image()
        r.render(1, 1)
        nomesh = r.download_image()
    sc.set_resolution(int(sc.camera.resolution.x), int(sc.camera.resolution.y))
    ref, _ = O.render(osc.geoms, osc.n_objects, osc.mats, osc.n_materials, O.LoadedScene(os.path.join(SCENES, "mesh_cornell.txt"), 1).camera, 4, iters=1)
    assert np.array_equal(nomesh, ref)


# ---------------------------------------------------------------- motion blur
@pytest.mark.parametrize("slices,iters,extra", [(4, 40, {}), (2, 33, {"batch": 5, "rr_start": 2}), (3, 50, {"direct_light": 1, "geom_path": 7})])
def test_motion_blur_matches_oracle(pkg, slices, iters, extra):
    """pt_set_motion: the shutter open from frame 0 to frame 1 of sampleScene_anim.txt (spheres move, the camera dollies);
    runs of 16 iterations go to the slice contexts in turn and accumulate in one framebuffer == the oracle's per-iteration
    scene states."""
    path = os.path.join(SCENES, "sampleScene_anim.txt")
    W, H, depth = 96, 72, 5
    a, b = pkg.SceneFile(path, 1, frame=0), pkg.SceneFile(path, 1, frame=1)
    a.set_resolution(W, H)
    with pkg.Renderer(0) as r:
        r.set_options(depth=depth, **extra)
        r.set_scene(a.geoms, a.n_objects, a.mats, a.n_materials)
        r.set_camera(a.camera)
        r.set_motion(b.geoms, b.camera, slices, pkg.ROTAT_DEGREES)
        r.clear_image()
        r.render(1, 7)                                    # in pieces, across slice boundaries
        r.render(8, iters - 7)
        g = r.download_image()
        st = r.stats()
        r.set_motion(None, None, 0)                       # off again: the static frame
        r.clear_image()
        r.render(1, 3)
        static = r.download_image()
    oa, ob = O.LoadedScene(path, 1, frame=0), O.LoadedScene(path, 1, frame=1)
    oa.set_resolution(W, H)
    sg, sc = O.motion_slices(oa.geoms, ob.geoms, oa.n_objects, oa.camera, ob.camera, slices, O.ROTAT_DEGREES)
    sh = []
    okw = {k: v for k, v in extra.items() if k in ("rr_start", "direct_light")}
    c, lc = O.render(oa.geoms, oa.n_objects, oa.mats, oa.n_materials, oa.camera, depth, iters=iters, slice_geoms=sg, slice_cams=sc,
                     shadow_out=sh, **okw)
    check(g, c, [int(x) for x in st.live_in[:depth]], [int(x) for x in lc], f"motion blur {slices} slices")
    assert int(st.iterations) == iters and int(st.shadow_rays) == sh[0]
    c0, _ = O.render(oa.geoms, oa.n_objects, oa.mats, oa.n_materials, oa.camera, depth, iters=3, **okw)
    assert np.array_equal(static, c0) and not np.array_equal(g, c0)


@pytest.mark.parametrize("segments,iters,extra", [(1, 20, {}), (4, 37, {"batch": 5, "rr_start": 2}), (7, 18, {"lens_radius": 0.3, "focal_distance": 9.0, "absorption": 1}),
                                                  (2, 35, {"sequences": 1, "geom_path": 1}), (3, 21, {"geom_path": 5, "rr_start": 1}),
                                                  (5, 9, {"geom_path": 1, "lens_radius": 0.2, "focal_distance": 10.0}),
                                                  (2, 19, {"direct_light": 1}), (3, 17, {"direct_light": 1, "geom_path": 1, "rr_start": 2})])
def test_motion_blur_per_ray_matches_oracle(pkg, segments, iters, extra):
    """pt_options.motion_per_ray: every path draws its shutter time (third number of its camera stream) and sees matrices
    and camera vectors interpolated entry-wise between the two knots around it, at all of its bounces (FEAT_MOTION
    kernels: the pair queue by default -- pre-test against boxes swept over the shutter interval, per-pair interpolated rows --
    and the scalar loop; the time is re-drawn from the stream at every bounce, nothing is stored in the ray)."""
    path = os.path.join(SCENES, "sampleScene_anim.txt")
    W, H, depth = 96, 72, 6
    a, b = pkg.SceneFile(path, 1, frame=0), pkg.SceneFile(path, 1, frame=1)
    a.set_resolution(W, H)
    with pkg.Renderer(0) as r:
        r.set_options(depth=depth, motion_per_ray=1, **extra)
        r.set_scene(a.geoms, a.n_objects, a.mats, a.n_materials)
        r.set_camera(a.camera)
        r.set_motion(b.geoms, b.camera, segments, pkg.ROTAT_DEGREES)
        r.clear_image()
        r.render(1, 7)
        r.render(8, iters - 7)
        g = r.download_image()
        st = r.stats()
        r.set_motion(None, None, 0)                       # off again: the static frame
        r.clear_image()
        r.render(1, 3)
        static = r.download_image()
    oa, ob = O.LoadedScene(path, 1, frame=0), O.LoadedScene(path, 1, frame=1)
    oa.set_resolution(W, H)
    kg, kc = O.motion_knots(oa.geoms, ob.geoms, oa.n_objects, oa.camera, ob.camera, segments, O.ROTAT_DEGREES)
    okw = {k: v for k, v in extra.items() if k in ("rr_start", "lens_radius", "focal_distance", "absorption", "direct_light")}
    sh = []
    c, lc = O.render(oa.geoms, oa.n_objects, oa.mats, oa.n_materials, oa.camera, depth, iters=iters, knot_geoms=kg, knot_cams=kc, shadow_out=sh, **okw)
    check(g, c, [int(x) for x in st.live_in[:depth]], [int(x) for x in lc], f"per-ray motion blur, {segments} segment(s)")
    assert int(st.iterations) == iters and int(st.shadow_rays) == sh[0]
    c0, _ = O.render(oa.geoms, oa.n_objects, oa.mats, oa.n_materials, oa.camera, depth, iters=3, **okw)
    assert np.array_equal(static, c0) and not np.array_equal(g, c0)


@pytest.mark.parametrize("geom_path", [0, 1])
def test_motion_blur_per_ray_with_scattering(pkg, geom_path):
    """A shutter time per ray through the scattering media of sss_blobs.txt (every object shifted, turned and rescaled a
    little between the two frames): random walk, index-matched pass-through and its offset at the ray's own time."""
    path = os.path.join(SCENES, "sss_blobs.txt")
    W, H, depth, iters = 72, 72, 10, 3
    sc = O.LoadedScene(path, 1)
    sc.set_resolution(W, H)
    rng = np.random.default_rng(12)
    gb = [O.make_geom(g.type, g.materialid, np.array(g.translation.tup()) + rng.normal(0, 0.15, 3), np.array(g.rotation.tup()) + rng.normal(0, 5.0, 3),
                      np.array(g.scale.tup()) * rng.uniform(0.9, 1.1, 3), O.ROTAT_DEGREES) for g in list(sc.geoms)[:sc.n_objects]]
    gba = (O.StaticGeom * len(gb))(*gb)
    kg, _ = O.motion_knots(sc.geoms, gba, sc.n_objects, sc.camera, None, 2, O.ROTAT_DEGREES)
    c, lc = O.render(sc.geoms, sc.n_objects, sc.mats, sc.n_materials, sc.camera, depth, iters=iters, knot_geoms=kg, scatter=1, direct_light=1)
    with pkg.Renderer(0) as r:
        r.set_options(depth=depth, motion_per_ray=1, scatter=1, direct_light=1, geom_path=geom_path)
        r.set_scene(C.cast(sc.geoms, C.POINTER(pkg.StaticGeom)), sc.n_objects, C.cast(sc.mats, C.POINTER(pkg.Material)), sc.n_materials)
        r.set_camera(pkg.CameraData.from_buffer_copy(sc.camera))
        r.set_motion(C.cast(gba, C.POINTER(pkg.StaticGeom)), None, 2, pkg.ROTAT_DEGREES)
        r.clear_image()
        r.render(1, iters)
        g = r.download_image()
        st = r.stats()
    check(g, c, [int(x) for x in st.live_in[:depth]], [int(x) for x in lc], f"per-ray motion blur with scattering + direct lighting, geom_path={geom_path}")


def test_motion_blur_per_ray_camera_at_rest_and_errors(pkg):
    path = os.path.join(SCENES, "sampleScene_anim.txt")
    W, H, depth = 64, 48, 4
    a, b = pkg.SceneFile(path, 1, frame=0), pkg.SceneFile(path, 1, frame=1)
    a.set_resolution(W, H)
    with pkg.Renderer(0) as r:
        r.set_options(depth=depth, motion_per_ray=1)
        r.set_scene(a.geoms, a.n_objects, a.mats, a.n_materials)
        r.set_camera(a.camera)
        r.set_motion(b.geoms, None, 3, pkg.ROTAT_DEGREES)      # the camera does not move
        r.clear_image()
        r.render(1, 5)
        g = r.download_image()
        st = r.stats()
        for bad in (dict(geom_path=3), dict(workgroup=512)):
            r.set_options(**bad)
            with pytest.raises(pkg.PtError):
                r.clear_image()
            r.set_options(**{k: 0 for k in bad})
    oa, ob = O.LoadedScene(path, 1, frame=0), O.LoadedScene(path, 1, frame=1)
    oa.set_resolution(W, H)
    kg, _ = O.motion_knots(oa.geoms, ob.geoms, oa.n_objects, oa.camera, None, 3, O.ROTAT_DEGREES)
    c, lc = O.render(oa.geoms, oa.n_objects, oa.mats, oa.n_materials, oa.camera, depth, iters=5, knot_geoms=kg)
    check(g, c, [int(x) for x in st.live_in[:depth]], [int(x) for x in lc], "per-ray motion blur, camera at rest")


# ---------------------------------------------------------------- committed golden fixtures
def _golden_cases():
    import json
    return json.load(open(os.path.join(ROOT, "tests", "golden", "render_checksums.json")))


@pytest.mark.parametrize("name", sorted(_golden_cases()))
def test_gpu_image_hashes_to_the_committed_golden(pkg, name):
    """The HIP path against the committed fixtures themselves (tests/golden/render_checksums.json: SHA-256 of the fp32
    image + live-ray counts, generated by oracle/make_render_golden.py), not only against the oracle run at test time."""
    import hashlib
    c = _golden_cases()[name]
    opts = {k: v for k, v in c["options"].items() if not k.startswith("_")}
    if any(k.startswith("_") for k in c["options"]):
        # round 3's cases: MESH triangles (the library's own .obj loader) / knot states of per-ray motion blur (frames 0 -> 1)
        sc = pkg.SceneFile(os.path.join(SCENES, c["scene"]), c["rotat"])
        sc.set_resolution(c["width"], c["height"])
        with pkg.Renderer(0) as r:
            r.set_options(depth=c["depth"], motion_per_ray=1 if c["options"].get("_knots") else 0, **opts)
            r.set_scene(sc.geoms, sc.n_objects, sc.mats, sc.n_materials)
            if c["options"].get("_meshes"):
                r.set_meshes(sc.meshes)
            r.set_camera(sc.camera)
            if c["options"].get("_knots"):
                nxt = pkg.SceneFile(os.path.join(SCENES, c["scene"]), c["rotat"], frame=1)
                r.set_motion(nxt.geoms, nxt.camera, c["options"]["_knots"], c["rotat"])
            r.clear_image()
            r.render(1, c["iterations"])
            img = r.download_image()
            live = [int(x) for x in r.stats().live_in[:c["depth"]]]
    else:
        img, live, _ = gpu_render(pkg, c["scene"], c["width"], c["height"], c["depth"], iters=c["iterations"], rotat=c["rotat"], **opts)
    assert live == c["live_in"]
    assert hashlib.sha256(np.ascontiguousarray(img).tobytes()).hexdigest() == c["sha256"]


# ---------------------------------------------------------------- depth of field
@pytest.mark.parametrize("geom_path,extra", [(0, {}), (3, {}), (4, {}), (6, {}), (7, {}), (7, {"direct_light": 1}), (5, {"direct_light": 1}),
                                             (0, {"strip_rows": 4, "strip_world": 2, "strip_rank": 1})])
def test_thin_lens_camera_matches_oracle(pkg, geom_path, extra):
    """pt_options.lens_radius / focal_distance: camera rays start on the lens disc, so the camera kernel gives up its
    shared-eye shortcuts (host eye transforms, eye-relative boxes) -- on every geometry path, with direct lighting and
    on a strip tile."""
    W, H, depth, iters = 88, 66, 4, 3
    sc = O.LoadedScene(os.path.join(SCENES, "sampleScene_spec.txt"), 1)
    sc.set_resolution(W, H)
    sh = []
    cpu, lc = O.render(sc.geoms, sc.n_objects, sc.mats, sc.n_materials, sc.camera, depth, iters=iters, lens_radius=0.35,
                       focal_distance=11.5, direct_light=extra.get("direct_light", 0), shadow_out=sh)
    gpu, lg, st = gpu_render(pkg, "sampleScene_spec.txt", W, H, depth, iters=iters, rotat=1, geom_path=geom_path,
                             lens_radius=0.35, focal_distance=11.5, **extra)
    if "strip_rows" in extra:
        from project3_pathtracer_amd import sharding
        rows = sharding.strip_global_rows(H, 2, 1, 4)
        assert np.array_equal(gpu, cpu[rows])
    else:
        check(gpu, cpu, lg, [int(x) for x in lc], f"thin lens geom_path={geom_path} {extra}")
        assert int(st.shadow_rays) == sh[0]
    pin, _, _ = gpu_render(pkg, "sampleScene_spec.txt", W, H, depth, iters=iters, rotat=1)
    assert "strip_rows" in extra or not np.array_equal(pin, gpu)


def test_lens_options_are_validated(pkg):
    with pkg.Renderer(0) as r:
        for bad in (dict(lens_radius=-1.0), dict(lens_radius=0.5, focal_distance=0.0), dict(lens_radius=float("nan"))):
            with pytest.raises(pkg.PtError):
                r.set_options(**bad)
            r.set_options(lens_radius=0.0, focal_distance=1.0)


@pytest.mark.parametrize("w,h,iters,batch", [(1, 8, 3, 0), (18, 1, 4, 3), (5, 5, 16, 16), (63, 1, 7, 16), (65, 1, 5, 4)])
def test_tiny_tiles_with_several_iterations_in_flight(pkg, w, h, iters, batch):
    """Fewer than 64 pixels per iteration slot: one 64-ray chunk of the camera kernel spans several slots (found by
    tests/fuzz_gpu.py: the one-boundary shortcut of the slot computation does not apply there)."""
    g, lg, _ = gpu_render(pkg, "sampleScene_spec.txt", w, h, 3, iters=iters, batch=batch, direct_light=1)
    sc = O.LoadedScene(os.path.join(SCENES, "sampleScene_spec.txt"))
    sc.set_resolution(w, h)
    c, lc = O.render(sc.geoms, sc.n_objects, sc.mats, sc.n_materials, sc.camera, 3, iters=iters, direct_light=1)
    check(g, c, lg, [int(x) for x in lc], f"{w}x{h} x{iters} batch {batch}")


def test_fuzz_sample(pkg):
    """A slice of tests/fuzz_gpu.py (random scenes x random option mixes, bit-exact vs the oracle) on every run; the full
    run (`python tests/fuzz_gpu.py 8000`) takes about half a minute on the GPU box."""
    import fuzz_gpu
    old = sys.argv
    try:
        sys.argv = ["fuzz_gpu.py", "120", "777000"]
        assert fuzz_gpu.main() == 0
    finally:
        sys.argv = old


@pytest.mark.parametrize("case", [3514219, 3505609, 3403963, 3103489])
def test_fuzz_cases_that_found_something(pkg, case, monkeypatch):
    """Cases of tests/fuzz_gpu.py that exposed a defect, kept as they are.  3514219 / 3505609 / 3403963 (round 4): a mesh light two units
    across in a scene 1.2e6 units wide -- the culling bounds of TRIANGLES had no pad for what a test loses on a ray that starts a million
    units away, a shadow ray lost its occluder; 3103489: the camera span table against the fp32 ray grid of a huge scene (the image was
    right, the bounds-checking build flagged the table)."""
    import fuzz_gpu
    monkeypatch.setattr(sys, "argv", ["fuzz_gpu.py", "1", str(case)])
    assert fuzz_gpu.main() == 0


# ---------------------------------------------------------------------------------------------------------------
# resident paths (pt_options.resident, round 4): the later bounces of a batch in ONE launch, paths kept in registers
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("scene,w,h,depth,iters,opts", [
    ("sampleScene_spec.txt", 121, 67, 8, 19, {"batch": 16}),                     # pair queue, ragged tile, two batches
    ("sampleScene_spec.txt", 64, 48, 3, 5, {"batch": 2}),                        # the shortest path that has anything to fuse
    ("sampleScene_spec.txt", 200, 120, 8, 6, {"rr_start": 2, "batch": 3}),       # Russian roulette keyed on each lane's own bounce
    ("cornell_glass.txt", 160, 90, 16, 4, {"batch": 4}),                         # refraction, 16 bounces
    ("cloud256.txt", 160, 90, 12, 3, {"rotat": 1}),                              # batched 4-wide walk
    ("cloud256.txt", 96, 54, 32, 2, {"rotat": 1, "rr_start": 3, "geom_path": 8}),  # ... nodes through L1/L2, depth 32
    ("sampleScene_spec.txt", 7, 5, 6, 40, {"batch": 16}),                        # a tile smaller than a wave
    ("sampleScene_spec.txt", 128, 72, 5, 9, {"workgroup": 512, "sequences": 1}),
])
def test_resident_paths_are_invisible(pkg, scene, w, h, depth, iters, opts):
    """pt_options.resident = 1: the camera launch, then ONE launch in which a wave keeps the paths that go on in its registers
    (per-lane bounce numbers, stream keys of every bounce from an LDS table) and refills the lanes of those that ended.  Image,
    per-bounce live counts and launch counts: the oracle's, and the launch-per-bounce path's, bit for bit."""
    o = dict(opts)
    rotat = o.pop("rotat", 0)
    a, la, sa = gpu_render(pkg, scene, w, h, depth, iters=iters, rotat=rotat, resident=-1, **o)
    b, lb, sb = gpu_render(pkg, scene, w, h, depth, iters=iters, rotat=rotat, resident=1, **o)
    c, lc = cpu_render(scene, w, h, depth, iters=iters, rotat=rotat, rr_start=o.get("rr_start", -1))
    check(b, c, lb, lc, f"resident paths, {scene} {w}x{h} depth {depth}")
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32)) and la == lb
    nb = -(-iters // (o.get("batch", 0) or 16))
    assert sa.bounce_launches == nb * depth and sb.bounce_launches == nb * 2, (sa.bounce_launches, sb.bounce_launches)


@pytest.mark.parametrize("refill", [1, 7, 32, 64])
def test_resident_paths_refill_thresholds(pkg, refill, monkeypatch):
    """The number of free lanes that triggers a refill (PT_REFILL_MIN, read when a context configures) changes which lane and
    which trip of a wave's loop traces a path -- and nothing else."""
    monkeypatch.setenv("PT_REFILL_MIN", str(refill))
    b, lb, _ = gpu_render(pkg, "sampleScene_spec.txt", 150, 83, 8, iters=5, batch=5, rr_start=3, resident=1)
    c, lc = cpu_render("sampleScene_spec.txt", 150, 83, 8, iters=5, rr_start=3)
    check(b, c, lb, lc, f"resident paths, refill at {refill} free lanes")


def test_resident_paths_fall_back_where_no_kernel_exists(pkg):
    """The per-bounce geometry paths and depth < 3 keep one launch per bounce whatever the option asks for: same image,
    launch count = depth."""
    for opts, depth in (({"geom_path": 1}, 5), ({"geom_path": 3}, 4), ({}, 2), ({}, 1)):
        b, lb, sb = gpu_render(pkg, "sampleScene_spec.txt", 96, 54, depth, iters=3, batch=3, resident=1, **opts)
        c, lc = cpu_render("sampleScene_spec.txt", 96, 54, depth, iters=3)
        check(b, c, lb, lc, f"resident asked for with {opts}, depth {depth}")
        assert sb.bounce_launches == depth


def _skip_stress_scene(seed):
    """Scenes built around Prim::self_r2: spheres far smaller than the rays that reach them are long (the reference's sphere quadratic
    then reports hit points INSIDE the sphere by more than its 0.0002 bias, and the path bounces on inside -- the oracle renders
    exactly that), mirror and diffuse, beside ellipsoids, huge and far-away primitives, thin slabs, overlapping pairs."""
    rng = np.random.default_rng(seed)
    mats = [O.make_material(color=rng.uniform(0.3, 1.0, 3)),                       # (the layout of _random_scene: tests/fuzz_gpu.py edits 1 and 3)
            O.make_material(color=rng.uniform(0.3, 1.0, 3)),
            O.make_material(color=(0.9, 0.9, 0.9), spec=(1, 1, 1), refl=1.0),
            O.make_material(color=(0, 0, 0), spec=(1, 1, 1), refr=1.0, ior=1.5),
            O.make_material(color=(1, 1, 1), emittance=8.0)]
    room = float(rng.choice([12.0, 40.0, 110.0, 300.0]))
    geoms = [O.make_geom(O.CUBE, 0, (0, 0, 0), (0, 0, 0) if seed % 2 else rng.uniform(-3, 3, 3), (room, room, room)),
             O.make_geom(O.CUBE, 4, (0, room * 0.45, 0), (0, 0, 0), (room * 0.4, 0.05 * room, room * 0.4))]
    ats = []
    for i in range(int(rng.integers(6, 28))):
        kind = O.SPHERE if rng.random() < 0.7 else O.CUBE
        r = rng.random()
        if r < 0.5:   s = np.full(3, rng.uniform(0.01, 0.3))                # tiny
        elif r < 0.6: s = np.full(3, rng.uniform(0.3, 3.0)) * (1.0 + rng.uniform(-2e-6, 2e-6, 3))     # almost uniform
        elif r < 0.7: s = rng.uniform(0.05, 2.0, 3)                         # ellipsoid / brick
        elif r < 0.8: s = np.array([rng.uniform(1, 4), 0.004, rng.uniform(1, 4)])   # thin slab
        elif r < 0.9: s = np.full(3, rng.uniform(60.0, 70.0) if room > 100 else rng.uniform(2.0, 5.0))
        else:         s = np.full(3, rng.uniform(0.3, 1.5))
        at = rng.uniform(-0.42, 0.42, 3) * room
        ats.append(at)
        geoms.append(O.make_geom(kind, int(rng.integers(0, 4)), at, rng.uniform(-3.2, 3.2, 3), s))
        if rng.random() < 0.2:                                                # a second one through the first
            geoms.append(O.make_geom(kind, int(rng.integers(0, 4)), at + rng.uniform(-0.5, 0.5, 3) * s, rng.uniform(-3.2, 3.2, 3), s))
    eye = rng.uniform(-0.3, 0.3, 3) * room
    tgt = ats[int(rng.integers(0, len(ats)))]
    view = tgt - eye
    view = view / np.linalg.norm(view) if np.linalg.norm(view) > 1e-3 else np.array([0, 0, -1.0])
    up = np.cross(view, rng.normal(size=3))
    up /= np.linalg.norm(up)
    return geoms, mats, eye, view, up, float(rng.uniform(4, 30))


def _extreme_scene(kind, seed):
    """Scenes at the edges of fp32: far from the world origin, huge, tiny, needle-shaped cubes, zero and negative scales (singular and
    mirrored transforms: the reference's parser accepts them, its arithmetic then does what it does -- and so must every culling
    bound and shortcut of the kernels)."""
    rng = np.random.default_rng(seed)
    mats = [O.make_material(color=rng.uniform(0.3, 1.0, 3)), O.make_material(color=rng.uniform(0.3, 1.0, 3)),
            O.make_material(color=(0.9, 0.9, 0.9), spec=(1, 1, 1), refl=1.0),
            O.make_material(color=(0, 0, 0), spec=(1, 1, 1), refr=1.0, ior=1.5), O.make_material(color=(1, 1, 1), emittance=8.0)]
    off, room, unit = np.zeros(3), 12.0, 1.0
    if kind == "far":
        off = rng.uniform(-1, 1, 3) * float(rng.choice([300.0, 2000.0, 20000.0]))
    if kind == "huge":
        unit = float(rng.choice([1e3, 1e5]))
    if kind == "tiny":
        unit = float(rng.choice([1e-2, 1e-4]))
    room *= unit
    geoms = [O.make_geom(O.CUBE, 0, off, rng.uniform(-3, 3, 3), (room, room, room)),
             O.make_geom(O.CUBE, 4, off + np.array([0, room * 0.45, 0]), (0, 0, 0), (room * 0.4, 0.05 * room, room * 0.4))]
    for _ in range(int(rng.integers(5, 20))):
        k = O.SPHERE if rng.random() < 0.5 else O.CUBE
        s = rng.uniform(0.3, 3.0, 3) * unit
        if kind == "needle":
            s = np.array([rng.uniform(2, 8), 1e-4 * rng.uniform(1, 50), rng.uniform(0.01, 3)])[rng.permutation(3)]
        if kind == "zero" and rng.random() < 0.4:
            s[int(rng.integers(0, 3))] = 0.0
        if kind == "neg" and rng.random() < 0.5:
            s = s * rng.choice([-1.0, 1.0], 3)
        geoms.append(O.make_geom(k, int(rng.integers(0, 4)), off + rng.uniform(-0.42, 0.42, 3) * room, rng.uniform(-3.2, 3.2, 3), s))
    eye = off + rng.uniform(-0.3, 0.3, 3) * room
    view = rng.normal(size=3)
    view /= np.linalg.norm(view)
    up = np.cross(view, rng.normal(size=3))
    up /= np.linalg.norm(up)
    return geoms, mats, eye, view, up, float(rng.uniform(15, 40))


@pytest.mark.parametrize("seed", range(2))
@pytest.mark.parametrize("kind", ["far", "huge", "tiny", "needle", "zero", "neg"])
def test_extreme_scenes_bit_exact(pkg, kind, seed):
    """The edges of fp32 (see _extreme_scene) on the per-primitive loop, the pair queue and the batched walk, resident paths on."""
    geoms, mats, eye, view, up, fovy = _extreme_scene(kind, 100 + seed)
    W, H, depth, iters = 64, 40, 8, 2
    ga = (O.StaticGeom * len(geoms))(*geoms)
    ma = (O.Material * len(mats))(*mats)
    cam = O.make_camera(W, H, eye, view, up, fovy)
    ref, live = O.render(ga, len(geoms), ma, len(mats), cam, depth, iters=iters, rr_start=-1, seed=seed)
    for geom_path in (1, 3, 5, 7):
        with pkg.Renderer(0) as r:
            r.set_options(depth=depth, seed=seed, geom_path=geom_path, batch=2, resident=1)
            r.set_scene(C.cast(ga, C.POINTER(pkg.StaticGeom)), len(geoms), C.cast(ma, C.POINTER(pkg.Material)), len(mats))
            r.set_camera(pkg.CameraData.from_buffer_copy(cam))
            r.clear_image()
            r.render(1, iters)
            img = r.download_image()
            st = r.stats()
        assert np.array_equal(img.view(np.uint32), ref.view(np.uint32)), (kind, seed, geom_path)
        assert [int(x) for x in st.live_in[:depth]] == [int(x) for x in live], (kind, seed, geom_path)


def test_slab_pretest_is_chosen_per_scene(pkg):
    """The pre-test with slabs (FEAT_SLAB kernel instances) is taken where a cube of the scene is tilted enough to have one -- the bundled
    scene with ROTAT read as radians -- and only by the plain pair-queue kernels; the image is the oracle's either way."""
    def info(rotat, **opts):
        sc = pkg.SceneFile(os.path.join(SCENES, "sampleScene_spec.txt"), rotat)
        sc.set_resolution(64, 48)
        with pkg.Renderer(0) as r:
            r.set_options(depth=6, **opts)
            r.set_scene(sc.geoms, sc.n_objects, sc.mats, sc.n_materials)
            r.set_camera(sc.camera)
            r.clear_image()
            r.render(1, 1)
            return r.launch_info()
    assert info(0).slab_pretest == 1 and info(0).resident == 1
    assert info(0, resident=-1).slab_pretest == 1                 # the launch-per-bounce kernels take it too
    assert info(1).slab_pretest == 0                              # degrees: axis-aligned walls
    assert info(0, direct_light=1).slab_pretest == 0              # shadow rays share the queues: plain kernels only
    assert info(0, geom_path=7).slab_pretest == 0                 # the batched walk has no such loop
    for rotat in (0, 1):
        g, lg, _ = gpu_render(pkg, "sampleScene_spec.txt", 97, 61, 7, iters=3, rotat=rotat)
        c, lc = cpu_render("sampleScene_spec.txt", 97, 61, 7, iters=3, rotat=rotat)
        check(g, c, lg, lc, f"slab pre-test, rotat {rotat}")


@pytest.mark.parametrize("seed", range(10))
def test_resident_paths_self_skip_stress(pkg, seed):
    """A resident path that leaves a convex primitive on its outside skips that primitive at its next bounce -- only where the
    reference's own arithmetic would miss it too (Prim::self_r2).  Scenes aimed at the places where it would not."""
    geoms, mats, eye, view, up, fovy = _skip_stress_scene(7000 + seed)
    W, H, depth, iters = 96, 54, 14, 3
    ga = (O.StaticGeom * len(geoms))(*geoms)
    ma = (O.Material * len(mats))(*mats)
    cam = O.make_camera(W, H, eye, view, up, fovy)
    ref, live = O.render(ga, len(geoms), ma, len(mats), cam, depth, iters=iters, rr_start=-1, seed=seed)
    for geom_path in (5, 7, 8, 1):                              # pair queue, batched walks (resident), per-primitive loop (launch per bounce)
        with pkg.Renderer(0) as r:
            r.set_options(depth=depth, seed=seed, geom_path=geom_path, resident=1, batch=3)
            r.set_scene(C.cast(ga, C.POINTER(pkg.StaticGeom)), len(geoms), C.cast(ma, C.POINTER(pkg.Material)), len(mats))
            r.set_camera(pkg.CameraData.from_buffer_copy(cam))
            r.clear_image()
            r.render(1, iters)
            img = r.download_image()
            st = r.stats()
            info = r.launch_info()
        check(img, ref, [int(x) for x in st.live_in[:depth]], [int(x) for x in live], f"self-skip stress {seed} geom_path={geom_path}")
        assert info.resident == (0 if geom_path == 1 else 1), (seed, geom_path)


@pytest.mark.parametrize("scene,depth,opts", [
    ("sampleScene_spec.txt", 6, {"direct_light": 1}),                                   # shadow-ray pass inside the resident loop
    ("sampleScene_spec.txt", 8, {"direct_light": 1, "rr_start": 2, "batch": 3}),
    ("sss_blobs.txt", 12, {"scatter": 1, "rotat": 1}),                                  # random walks inside media, paths resident
    ("sss_blobs.txt", 10, {"scatter": 1, "direct_light": 1, "absorption": 1, "rr_start": 3, "rotat": 1}),
    ("cloud256.txt", 9, {"direct_light": 1, "rotat": 1}),                               # batched walk + light sampling
])
def test_resident_paths_with_light_sampling_and_media(pkg, scene, depth, opts):
    """The resident-path launch also exists with direct lighting (the chunk's second pass through the nearest-hit machinery for the
    shadow rays runs inside the wave's loop; radiance entries accumulate along the path) and with scattering media: image,
    live-ray and shadow-ray counts are the oracle's and the launch-per-bounce path's; two bounce launches per batch."""
    o = dict(opts)
    rotat = o.pop("rotat", 0)
    w, h, iters = 112, 63, 5
    a, la, sa = gpu_render(pkg, scene, w, h, depth, iters=iters, rotat=rotat, resident=-1, **o)
    b, lb, sb = gpu_render(pkg, scene, w, h, depth, iters=iters, rotat=rotat, resident=1, **o)
    sh = []
    ok = {k: v for k, v in o.items() if k in ("direct_light", "scatter", "absorption", "rr_start")}
    c, lc = cpu_render(scene, w, h, depth, iters=iters, rotat=rotat, shadow_out=sh, **ok)
    check(b, c, lb, lc, f"resident paths with {opts} on {scene}")
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32)) and la == lb
    assert int(sb.shadow_rays) == int(sa.shadow_rays) == sh[0]
    nb = -(-iters // (o.get("batch", 0) or 16))
    assert sb.bounce_launches == 2 * nb and sa.bounce_launches == depth * nb


def test_resident_paths_on_strip_tiles_and_resume(pkg):
    """Resident paths on an interleaved-strip tile (global pixel numbering), two sequences, several pt_render calls."""
    from project3_pathtracer_amd import sharding
    W, H, depth = 128, 72, 6
    c, _ = cpu_render("sampleScene_spec.txt", W, H, depth, iters=21, rr_start=1)
    sc = pkg.SceneFile(os.path.join(SCENES, "sampleScene_spec.txt"))
    sc.set_resolution(W, H)
    with pkg.Renderer(0) as r:
        r.set_options(depth=depth, rr_start=1, batch=4, sequences=2, resident=1, strip_rows=8, strip_world=3, strip_rank=2)
        r.set_scene(sc.geoms, sc.n_objects, sc.mats, sc.n_materials)
        r.set_camera(sc.camera)
        r.clear_image()
        r.render(1, 9)
        r.render(10, 1)
        r.render(11, 11)
        g = r.download_image()
    want = c[sharding.strip_global_rows(H, 3, 2, 8)]
    assert np.array_equal(g.view(np.uint32), want.view(np.uint32))
