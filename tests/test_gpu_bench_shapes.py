"""GPU parity at the shapes bench.py actually launches (BASELINE.json configs 2-5).

The small-frame parity tests (test_gpu_parity.py) never reach the launch shapes of the benchmark: 1920x1080 with 16
iterations in flight (33 M rays per launch, pool segments sized by writers x rounds), a 3840x2160 tile (27-bit pixel
packing, w_magic / strip_magic at W = 3840), the 256-primitive cloud at depth 32.  Here every one of them is rendered
through the C-ABI and pinned to the ORACLE:
  * the SHA-256 of the whole fp32 frame and the per-bounce live-ray counts against tests/golden/render_checksums_large.json
    (made by oracle/make_render_golden_large.py: the oracle's own image of the same configuration), and
  * single paths of sample pixels re-traced by the oracle in this process.
Bit-exact is expected; the bar of BASELINE.json (1e-4) applies to the sampled pixels.
"""
import ctypes as C
import hashlib
import json
import os

import numpy as np
import pytest

import oracle_lib as O
from __graft_entry__ import load_package

pytestmark = pytest.mark.gpu

TOL = 1e-4
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCENES = os.path.join(ROOT, "scenes")
GOLDEN = json.load(open(os.path.join(ROOT, "tests", "golden", "render_checksums_large.json")))


@pytest.fixture(scope="module")
def pkg():
    p = load_package()
    p.lib()
    return p


def sha(img):
    return hashlib.sha256(np.ascontiguousarray(img, dtype=np.float32).tobytes()).hexdigest()


def gpu_render(pkg, g, **opts):
    sc = pkg.SceneFile(os.path.join(SCENES, g["scene"]), g["rotat"])
    sc.set_resolution(g["width"], g["height"])
    kw = dict(g["options"])
    kw.update(opts)
    with pkg.Renderer(0) as r:
        r.set_options(depth=g["depth"], **kw)
        r.set_scene(sc.geoms, sc.n_objects, sc.mats, sc.n_materials)
        r.set_camera(sc.camera)
        r.clear_image()
        r.render(1, g["iterations"])
        img = r.download_image()
        st = r.stats()
    return img, [int(x) for x in st.live_in[:g["depth"]]]


def oracle_pixels(g, pixels):
    """Running mean over the iterations of single paths traced by the oracle (fp32, the accumulate kernel's formula)."""
    sc = O.LoadedScene(os.path.join(SCENES, g["scene"]), g["rotat"])
    sc.set_resolution(g["width"], g["height"])
    o = g["options"]
    opt = O.Options(g["depth"], o.get("rr_start", -1), o.get("seed", 0), O.TRIG_POLY)
    L = O.lib()
    out = {}
    for (x, y) in pixels:
        acc = np.zeros(3, np.float32)
        for it in range(1, g["iterations"] + 1):
            l = np.array(L.o_trace_path(sc.geoms, sc.n_objects, sc.mats, sc.n_materials, C.byref(sc.camera), C.byref(opt),
                                        x, y, it, None).tup(), np.float32)
            acc = (acc * np.float32(it - 1) + l) / np.float32(it)
        out[(x, y)] = acc
    return out


def check_against_golden(img, live, g, what):
    H = g["height"]
    for y, want in g["row_sha256"].items():
        assert sha(img[int(y)]) == want, f"{what}: row {y} differs from the oracle's"
    assert live == g["live_in"], f"{what}: live-ray counts differ from the oracle's"
    assert sha(img) == g["sha256"], f"{what}: frame differs from the oracle's"
    assert img.shape[0] == H and np.isfinite(img).all()


def check_sample_pixels(img, g, pixels, what):
    want = oracle_pixels(g, pixels)
    for (x, y), w in want.items():
        assert np.allclose(img[y, x], w, atol=TOL, rtol=0), (what, x, y, img[y, x], w)


# ---------------------------------------------------------------- config 2: 1920x1080, 16 iterations in flight
def test_config2_1080p_batch16_all_compaction_modes(pkg):
    """bench.py's default launch shape: 1920x1080, depth 8, 32 iterations as two batches of 16 (33 M rays per launch,
    32 pool segments): identical bits with compaction 1, 2 and 0, and the oracle's frame."""
    g = GOLDEN["config2_1080p_32spp"]
    a, la = gpu_render(pkg, g, batch=16)
    check_against_golden(a, la, g, "config 2, compaction 1")
    W, H = g["width"], g["height"]
    check_sample_pixels(a, g, [(x, y) for y in (0, 511, H - 1) for x in range(5, W, 479)], "config 2")
    b, lb = gpu_render(pkg, g, batch=16, compaction=2)
    assert np.array_equal(a, b) and la == lb
    c, lc = gpu_render(pkg, g, batch=16, compaction=0)
    assert np.array_equal(a, c) and la == lc


@pytest.mark.parametrize("iters", [20, 5, 17])
def test_uneven_iteration_counts_split_into_equal_batches(pkg, iters):
    """The driver's `--steps 20 --warmup 5`: 20 iterations run as 10 + 10 (not 16 + 4) off the one captured graph, after a
    5-iteration call on the same context; the image is that of 25 iterations one at a time."""
    g = dict(GOLDEN["config2_1080p_32spp"], width=480, height=270)
    sc = pkg.SceneFile(os.path.join(SCENES, g["scene"]), g["rotat"])
    sc.set_resolution(g["width"], g["height"])
    imgs = []
    for batch, plan in ((16, (5, iters)), (1, (5 + iters,))):
        with pkg.Renderer(0) as r:
            r.set_options(depth=g["depth"], batch=batch)
            r.set_scene(sc.geoms, sc.n_objects, sc.mats, sc.n_materials)
            r.set_camera(sc.camera)
            r.clear_image()
            first = 1
            for n in plan:
                r.render(first, n)
                first += n
            imgs.append(r.download_image())
            st = r.stats()
            assert int(st.iterations) == 5 + iters
            if batch == 16:      # 5 -> one batch; 20 -> 10 + 10; 17 -> 9 + 8
                per_batch = r.launch_info().launches_per_batch      # depth, or 2 with resident paths (the library default here)
                assert per_batch in (2, g["depth"])
                assert int(st.bounce_launches) == per_batch * (1 + (iters + 15) // 16)
    assert np.array_equal(imgs[0], imgs[1])
    ref, _ = O.render(*_oracle_scene(g), g["depth"], iters=5 + iters)
    assert np.array_equal(imgs[0], ref)


def _oracle_scene(g):
    sc = O.LoadedScene(os.path.join(SCENES, g["scene"]), g["rotat"])
    sc.set_resolution(g["width"], g["height"])
    return sc.geoms, sc.n_objects, sc.mats, sc.n_materials, sc.camera


def test_restart_ignores_stale_framebuffer_contents(pkg):
    """Iteration 1 restarts the running mean without reading the old image: NaN / Inf left in a caller-owned buffer
    (torch.empty, a previous frame) must not survive."""
    import torch
    W, H, depth = 96, 64, 4
    sc = pkg.SceneFile(os.path.join(SCENES, "sampleScene.txt"))
    sc.set_resolution(W, H)
    fb = torch.full((H, W, 3), float("nan"), dtype=torch.float32, device="cuda:0")
    fb[::2] = float("inf")
    torch.cuda.synchronize()
    with pkg.Renderer(0) as r:
        r.set_options(depth=depth)
        r.set_scene(sc.geoms, sc.n_objects, sc.mats, sc.n_materials)
        r.set_camera(sc.camera)
        r.bind_image(fb.data_ptr())
        r.render(1, 3)
        r.synchronize()
    g = {"scene": "sampleScene.txt", "rotat": 0, "width": W, "height": H}
    ref, _ = O.render(*_oracle_scene(g), depth, iters=3)
    assert np.array_equal(fb.cpu().numpy(), ref)


# ---------------------------------------------------------------- config 3: glass, depth 16
def test_config3_glass_1080p_depth16(pkg):
    g = GOLDEN["config3_glass_1080p_16spp_depth16"]
    a, la = gpu_render(pkg, g)
    check_against_golden(a, la, g, "config 3")


# ---------------------------------------------------------------- config 4: 3840x2160 as one tile and as 8 strip tiles
def test_config4_4k_one_tile_and_eight_strip_tiles(pkg):
    """BASELINE configs[3]: the 4K frame on one context (8.3 M pixels x 16 iterations in flight = 133 M rays per launch,
    pixel indices up to 2^23 beside the 4-bit slot) and as the 8 interleaved-strip tiles the 8-GPU run renders, put back
    together: both equal the oracle's frame."""
    g = GOLDEN["config4_4k_16spp"]
    W, H = g["width"], g["height"]
    a, la = gpu_render(pkg, g)
    check_against_golden(a, la, g, "config 4, one tile")
    check_sample_pixels(a, g, [(x, y) for y in (1, H // 2 + 3, H - 2) for x in (0, 1917, W - 1)], "config 4")
    from project3_pathtracer_amd import sharding
    world = 8
    frame = np.empty_like(a)
    live = [0] * g["depth"]
    for rank in range(world):
        t, lt = gpu_render(pkg, g, strip_rows=sharding.STRIP_ROWS, strip_world=world, strip_rank=rank)
        rows = sharding.strip_global_rows(H, world, rank)
        assert t.shape[0] == len(rows)
        frame[rows] = t
        live = [x + y for x, y in zip(live, lt)]
    assert np.array_equal(frame, a) and live == la


# ---------------------------------------------------------------- config 5: 256-primitive cloud at depth 32
@pytest.mark.parametrize("name", ["config5_cloud_480x270_depth32", "config5_cloud_480x270_depth32_rr"])
def test_config5_cloud_depth32(pkg, name):
    """BASELINE configs[4]'s scene and depth (32 bounces, with and without Russian roulette) on the library's default
    large-scene path and on the per-lane hierarchy walk."""
    g = GOLDEN[name]
    a, la = gpu_render(pkg, g)
    check_against_golden(a, la, g, name)
    for gp in (4, 6):
        b, lb = gpu_render(pkg, g, geom_path=gp)
        assert np.array_equal(a, b) and la == lb, gp


def test_config5_cloud_1080p_properties(pkg):
    """The cloud at the full 1920x1080 x depth 32 (the oracle needs minutes for it): every geometry path and launch shape
    gives the same bits, counts fall monotonically, sample pixels equal single oracle paths."""
    g = {"scene": "cloud256.txt", "rotat": 1, "width": 1920, "height": 1080, "depth": 32, "iterations": 2,
         "options": {"rr_start": 3}}
    a, la = gpu_render(pkg, g)
    b, lb = gpu_render(pkg, g, geom_path=4, workgroup=512)
    assert np.array_equal(a, b) and la == lb
    assert la[0] == 2 * 1920 * 1080 and all(x >= y for x, y in zip(la, la[1:]))
    check_sample_pixels(a, g, [(x, y) for y in (7, 540, 1071) for x in range(11, 1920, 313)], "config 5 at 1080p")


# ---------------------------------------------------------------- the RCCL branch of the gather, on one GPU
def test_gather_over_nccl_world_size_1(pkg):
    """bench.py's N > 1 exchange step is ONE torch.distributed.gather on the nccl (= RCCL) backend of device tensors.
    With a single rank the collective still goes through RCCL on cuda:0: init, gather into a list on the root, strip rows
    put in place -- the frame must equal the tile."""
    import torch
    import torch.distributed as dist
    from project3_pathtracer_amd import sharding
    if dist.is_initialized():
        pytest.skip("a process group already exists in this process")
    W, H, depth = 256, 100, 4
    sc = pkg.SceneFile(os.path.join(SCENES, "sampleScene_spec.txt"))
    sc.set_resolution(W, H)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29561")
    dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=dev)
    try:
        fb = torch.zeros((H, W, 3), dtype=torch.float32, device=dev)
        with pkg.Renderer(0) as r:
            r.set_options(depth=depth, strip_rows=sharding.STRIP_ROWS, strip_world=1, strip_rank=0)
            r.set_scene(sc.geoms, sc.n_objects, sc.mats, sc.n_materials)
            r.set_camera(sc.camera)
            r.bind_image(fb.data_ptr())
            r.render(1, 2)
            r.synchronize()
        # the world_size == 1 shortcut of gather_strips is bypassed on purpose: run the collective itself
        bufs = [torch.empty_like(fb)]
        dist.gather(fb, bufs, dst=0)
        rows = torch.as_tensor(sharding.strip_global_rows(H, 1, 0), dtype=torch.long, device=dev)
        frame = torch.empty_like(fb)
        frame.index_copy_(0, rows, bufs[0][: rows.numel()])
        t = torch.tensor([1.0, 2.0], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.barrier(device_ids=[0])
        torch.cuda.synchronize(dev)
        assert torch.equal(frame, fb) and t.tolist() == [1.0, 2.0]
        g = {"scene": "sampleScene_spec.txt", "rotat": 0, "width": W, "height": H}
        ref, _ = O.render(*_oracle_scene(g), depth, iters=2)
        assert np.array_equal(frame.cpu().numpy(), ref)
    finally:
        dist.destroy_process_group()
