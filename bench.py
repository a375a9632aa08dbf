#!/usr/bin/env python3
"""bench.py -- ray-bounces/s of the path-tracing hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--config 1..5] [--scaling weak|strong]

`--config` picks one of BASELINE.json's configurations (scene, resolution, depth; default 2 = the headline one);
`--steps` defaults to the config's spp.  `--scaling strong` keeps the config's frame fixed and cuts it into N tiles
(configs 4 and 5 are quoted that way: one 3840x2160 / 1920x1080 frame over the 8 GPUs of a node); the default for
N > 1 is weak scaling (every rank renders ~1920x1080 pixels of a frame that grows with N).

One *step* = one iteration (1 sample per pixel) of the hot path over the whole frame: camera rays, up to
`depth` bounce launches (intersect + shade + compaction), framebuffer accumulate.  At N=1 the workload is
BASELINE.json configs[1]: the bundled sampleScene at 1920x1080, 8 bounces, diffuse+specular, K=256 steps =
its 256 spp.  For N>1 (launched by torch.distributed.run, one rank per GPU) the frame grows with N at fixed
aspect and camera (N=4 is configs[3]'s 3840x2160) and every rank renders one band of rows of it: per-GPU work
is fixed ("weak"), no collective on the data path during rendering, and ONE RCCL gather of framebuffer tiles to
rank 0 inside the timed region (the path's only real exchange step): the tiles of the frame the K steps just rendered,
behind them (`value`).  A second region times the steady-state variant -- the previous frame's tiles gathered on RCCL's
stream WHILE the next K steps render -- and is reported beside it as `value_pipelined_gather`.

Prints ONE JSON line (rank 0).  `value` = ray-bounces of all ranks / wall time, in Mray-bounces/s, with
inputs resident in HBM before the timed region.  The `roofline` block prices the dominant kernel (k_bounce)
against HBM: algorithmic bytes of the timed region's launches / the HIP-event time of the timed
region on the render stream (two launch sequences are in flight, so a launch's own duration overlaps its
neighbour's: the block also carries the kernel-alone figure from per-launch events).  `cpu_baseline` times the CPU
oracle on a bounded sample of the same workload on this box's host cores (rank 0, N=1 only).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
VALU_PEAK_TFLOPS = 157.3  # fp32 vector peak (FMA = 2 flops), same guide / SURVEY 8(d)


# BASELINE.json `configs`, in order.  spp = the ITERATIONS the config is quoted with (= default --steps).
CONFIGS = {
    1: dict(scene="sampleScene.txt", width=400, height=400, depth=4, spp=1, lobes="diffuse",
            note="configs[0]: the reference's own CPU-runnable case (parity config)"),
    2: dict(scene="sampleScene_spec.txt", width=1920, height=1080, depth=8, spp=256, lobes="diffuse+specular",
            note="configs[1]: the configuration `metric` is quoted on"),
    3: dict(scene="cornell_glass.txt", width=1920, height=1080, depth=16, spp=1024, lobes="diffuse+refraction", rotat="degrees",
            note="configs[2]: Cornell box with glass sphere"),
    4: dict(scene="sampleScene_spec.txt", width=3840, height=2160, depth=8, spp=512, lobes="diffuse+specular",
            note="configs[3]: one 3840x2160 frame, pixel-tiled over the GPUs (use --scaling strong for N > 1)"),
    5: dict(scene="cloud256.txt", width=1920, height=1080, depth=32, spp=4096, lobes="diffuse+specular+refraction", rotat="degrees",
            note="configs[4]: 256-primitive cloud, compaction / divergence stress (use --scaling strong for N > 1)"),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--config", type=int, default=2, choices=sorted(CONFIGS), help="BASELINE.json configuration (default 2)")
    ap.add_argument("--steps", type=int, default=None, help="iterations timed (default: the config's spp, at most 4096)")
    ap.add_argument("--warmup", type=int, default=256)      # 50 ms: the GPU clock needs that long to settle (16: -3.5 %)
    ap.add_argument("--settle-ms", type=float, default=120.0,
                    help="after the W warm-up steps keep rendering untimed iterations until the GPU has been busy this long "
                         "(DVFS: a 1 ms warm-up leaves the clock ~16 %% low for the whole of a 4 ms timed region); 0 = off")
    ap.add_argument("--settle-max-ms", type=float, default=1500.0,
                    help="upper bound of the settle phase (it ends earlier once the untimed chunks stop getting faster)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="N > 1: weak = ~1920x1080 pixels per rank of a frame that grows with N; strong = the config's frame cut into N tiles")
    ap.add_argument("--scene", default=None)
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--depth", type=int, default=None)
    ap.add_argument("--rr-start", type=int, default=-1)
    ap.add_argument("--rotat", choices=["radians", "degrees"], default=None)
    ap.add_argument("--workgroup", type=int, default=0)
    ap.add_argument("--geom-path", type=int, default=0)
    ap.add_argument("--no-compaction", action="store_true")
    ap.add_argument("--batch", type=int, default=0, help="iterations in flight per launch sequence (0 = library default)")
    ap.add_argument("--sequences", type=int, default=0, help="launch sequences in flight (0 = library default 2)")
    ap.add_argument("--resident", type=int, default=0, choices=[-1, 0, 1],
                    help="later bounces as ONE launch with the paths resident in registers: 1 on, -1 off, 0 = library default")
    ap.add_argument("--dist-timeout", type=float, default=600.0,
                    help="N > 1: seconds a rank waits for the others before giving up (a fresh box pages torch in for a minute or two)")
    ap.add_argument("--compaction", type=int, default=1, help="1 per-wave sharded (default), 2 workgroup scan, 0 off")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--direct-light", action="store_true", help="explicit light sampling (not the headline workload)")
    ap.add_argument("--bands", action="store_true", help="N>1: one contiguous row band per rank instead of interleaved 8-row strips")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU time of the baseline sample")
    ap.add_argument("--traffic-json", default=os.path.join(ROOT, "profiles", "traffic_latest.json"),
                    help="PMC-derived HBM bytes per launch per config (written from profiles/collect_pmc.py summaries), if present")
    args = ap.parse_args()
    cfg = CONFIGS[args.config]
    args.custom = any(v is not None for v in (args.scene, args.width, args.height, args.depth, args.rotat))
    args.scene = args.scene or cfg["scene"]
    args.width = args.width or cfg["width"]
    args.height = args.height or cfg["height"]
    args.depth = args.depth or cfg["depth"]
    args.rotat = args.rotat or cfg.get("rotat", "radians")
    if args.steps is None:
        args.steps = cfg["spp"]
    return args


def cpu_baseline(args, scene_path, rotat):
    """Oracle (kind 'port') on the host cores, bounded sample of the same workload: the config's frame at as many spp as
    fit ~cpu-seconds; a frame the oracle cannot finish once in that time is sampled at 1/2, 1/4 ... of the resolution
    (same scene, camera, depth: the ray-bounce rate does not depend on the pixel count)."""
    import numpy as np  # noqa: F401
    import oracle_lib as O
    O.build()
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    if os.environ.get("PT_BENCH_CPU_THREADS"):
        cores = max(1, min(cores, int(os.environ["PT_BENCH_CPU_THREADS"])))
    dl = 1 if args.direct_light else 0

    def run(w, h, spp, threads):
        sc = O.LoadedScene(scene_path, rotat)
        sc.set_resolution(w, h)
        t0 = time.perf_counter()
        _, live = O.render(sc.geoms, sc.n_objects, sc.mats, sc.n_materials, sc.camera, args.depth, iters=spp,
                           rr_start=args.rr_start, nthreads=threads, direct_light=dl)
        return int(live.sum()), time.perf_counter() - t0

    # calibration on 1/64 of the pixels
    cw, ch = max(16, args.width // 8), max(16, args.height // 8)
    rb_c, dt_c = run(cw, ch, 1, cores)
    per_px = dt_c / (cw * ch)
    scale = 1
    while per_px * (args.width // scale) * (args.height // scale) > args.cpu_seconds and scale < 8:
        scale *= 2
    w, h = max(16, args.width // scale), max(16, args.height // scale)
    spp = max(1, min(64, args.steps, int(args.cpu_seconds / max(per_px * w * h, 1e-4))))
    rb, dt = run(w, h, spp, cores)
    # per-core figure: the calibration frame on one thread
    rb1, dt1 = run(cw, ch, 1, 1)
    return {"value": rb / dt / 1e6, "unit": "Mray-bounces/s", "cores": cores, "kind": "port",
            "single_thread": rb1 / dt1 / 1e6,
            "sample": f"{w}x{h} x {spp} spp x {args.depth} bounces of the same scene and camera"
                      f"{'' if scale == 1 else f' (1/{scale} of the resolution)'} "
                      f"({rb} ray-bounces in {dt:.2f} s, oracle/pt_oracle.c, {cores} threads)",
            "ms_per_frame": dt / spp * 1e3 * scale * scale}


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world

    import numpy as np
    import torch
    import torch.distributed as dist
    from __graft_entry__ import load_package
    pkg = load_package()
    pkg.lib()     # fails loudly when libptamd.so is missing: no fallback path

    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU (the render path has no CPU fallback)")
    # rehearsal mode for a 1-GPU box: all ranks share device 0 and the collective runs on gloo (RCCL refuses two
    # ranks on one device); the driver's real runs use one GPU per rank and nccl (= RCCL over xGMI)
    rehearsal = os.environ.get("PT_BENCH_REHEARSAL", "0") == "1"
    dev_index = 0 if rehearsal else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        import datetime
        # a missing rank must not hang the run until the driver's limit: the rendezvous gives up after --dist-timeout
        # seconds and this rank exits non-zero (a fresh process is the retry; nothing here re-executes itself)
        try:
            if rehearsal:
                dist.init_process_group(backend="gloo", timeout=datetime.timedelta(seconds=args.dist_timeout))
            else:
                dist.init_process_group(backend="nccl", device_id=dev, timeout=datetime.timedelta(seconds=args.dist_timeout))
        except Exception as e:      # noqa: BLE001
            print(f"bench.py: rank {rank}/{world}: init_process_group gave up after {args.dist_timeout:.0f} s: {e}", file=sys.stderr, flush=True)
            sys.exit(3)

    rotat = pkg.ROTAT_DEGREES if args.rotat == "degrees" else pkg.ROTAT_RADIANS
    scene_path = os.path.join(ROOT, "scenes", args.scene)
    sc = pkg.SceneFile(scene_path, rotat)
    # weak scaling: the frame grows with the GPU count at fixed aspect and camera (N=4 is BASELINE configs[3]'s
    # 3840x2160), so every rank owns a band of ~1920*1080 pixels of the same picture
    from project3_pathtracer_amd import sharding
    W, Hfull = sharding.scaled_frame(args.width, args.height, world, args.scaling)
    sc.set_resolution(W, Hfull)
    # N>1: interleaved 8-row strips (strip k -> rank k % N) so that every rank sees the same mix of ceiling, walls
    # and floor; --bands switches to one contiguous band per rank (up to ~9 % slower at N=8: the bands differ in
    # path length)
    strips = world > 1 and not args.bands
    r0, r1 = sharding.band_rows(Hfull, world, rank)
    if strips:
        Hband = sharding.strip_local_rows(Hfull, world, rank)
        Hmax = sharding.max_strip_rows(Hfull, world)
    else:
        Hband = r1 - r0
        Hmax = sharding.max_band_rows(Hfull, world)

    fb = torch.zeros((Hmax, W, 3), dtype=torch.float32, device=dev)
    torch.cuda.synchronize(dev)      # the zero-fill ran on torch's stream, the renderer has its own
    r = pkg.Renderer(dev_index)
    r.set_options(depth=args.depth, rr_start=args.rr_start, workgroup=args.workgroup, geom_path=args.geom_path,
                  compaction=0 if args.no_compaction else args.compaction, batch=args.batch, use_graph=0 if args.no_graph else 1,
                  row_begin=r0 if (world > 1 and not strips) else 0, row_end=r1 if (world > 1 and not strips) else 0,
                  strip_rows=sharding.STRIP_ROWS if strips else 0, strip_world=world if strips else 0,
                  strip_rank=rank if strips else 0, direct_light=1 if args.direct_light else 0, sequences=args.sequences,
                  resident=args.resident)
    r.set_scene(sc.geoms, sc.n_objects, sc.mats, sc.n_materials)
    r.set_camera(sc.camera)
    r.bind_image(fb.data_ptr())

    def barrier():
        if rehearsal:
            dist.barrier()
        else:
            dist.barrier(device_ids=[dev_index])

    # receive buffers and the frame of the gather: allocated once, here, so that nothing allocates inside the timed region
    gbufs, gframe = sharding.gather_buffers(fb.cpu() if rehearsal else fb, Hfull, world, rank, dst=0)

    def gather(tile):
        fn = sharding.gather_strips if strips else sharding.gather_bands
        if rehearsal:      # gloo moves host tensors
            return fn(tile.cpu(), Hfull, world, rank, dist=dist, dst=0, bufs=gbufs, frame=gframe)
        return fn(tile, Hfull, world, rank, dist=dist, dst=0, bufs=gbufs, frame=gframe)       # RCCL over xGMI

    def sync():
        r.synchronize()
        torch.cuda.synchronize(dev)
        if world > 1:
            barrier()
            torch.cuda.synchronize(dev)

    def tile_checksum(t):
        # order-independent: the bit patterns summed as integers (a float sum depends on the reduction's shape)
        return int(t.contiguous().view(torch.int32).to(torch.int64).sum().item())

    # warmup (untimed): also captures the per-sequence hipGraphs
    t_w = time.perf_counter()
    if args.warmup > 0:
        r.render(1, args.warmup)
    gather_ms = None
    if world > 1:
        sync()
        t_g = time.perf_counter()
        frame = gather(fb)            # (also brings RCCL's communicator up before anything is timed)
        sync()
        gather_ms = (time.perf_counter() - t_g) * 1e3
    sync()
    next_iter = args.warmup + 1
    # cold figure (N = 1, reported as `value_cold`, never as `value`): the same K steps right behind the W warm-up steps,
    # before the GPU clock has settled -- what the first ~100 ms of a real render run at
    value_cold = None
    if world == 1 and args.settle_ms > 0:
        r.reset_stats()
        sync()
        t0 = time.perf_counter()
        r.render(next_iter, args.steps)
        r.synchronize()
        dt_c = time.perf_counter() - t0
        value_cold = int(r.stats().ray_bounces) / dt_c / 1e6
        next_iter += args.steps
    # clock settle (untimed, reported): the W warm-up steps last W x ~0.2 ms here, and the GPU needs ~50-100 ms of load
    # to reach the clock it then holds.  The same frame keeps being rendered until the device has been busy for
    # settle_ms; the timed region is untouched: exactly K steps, bracketed as before.
    # The phase lasts AT LEAST settle_ms and then goes on while the chunks are still getting faster -- a box that sat idle through a
    # CPU-only phase (the oracle leg of a previous bench.py, a cold start) can take several hundred ms to reach its clock --, at most
    # settle_max_ms: it ends when two chunks in a row are no more than 2 % faster than the one before them.
    settle_iters = 0
    chunk = 64
    chunk_s = []
    while args.settle_ms > 0 and settle_iters < 65536:
        elapsed_ms = (time.perf_counter() - t_w) * 1e3
        if elapsed_ms >= args.settle_max_ms:
            break
        if elapsed_ms >= args.settle_ms and len(chunk_s) >= 3 and chunk_s[-1] > 0.98 * chunk_s[-2] and chunk_s[-2] > 0.98 * chunk_s[-3]:
            break
        t_c = time.perf_counter()
        r.render(next_iter + settle_iters, chunk)
        r.synchronize()
        chunk_s.append(time.perf_counter() - t_c)
        settle_iters += chunk
    if world > 1:
        # every rank leaves the settle phase at its own iteration count (it only has to be warm); agree on the largest so
        # that the timed iterations carry the same numbers everywhere (RNG streams are keyed on them)
        tmp = torch.tensor([float(settle_iters)], dtype=torch.float64, device="cpu" if rehearsal else dev)
        dist.all_reduce(tmp, op=dist.ReduceOp.MAX)
        more = int(tmp[0]) - settle_iters
        if more > 0:
            r.render(next_iter + settle_iters, more)
            settle_iters += more
    sync()
    r.reset_stats()

    # timed region: exactly K steps, then (N > 1) the ONE framebuffer gather of the frame those steps rendered -- the
    # path's only exchange step, dependent on the render, inside the barriers.
    first = next_iter + settle_iters
    sync()
    t0 = time.perf_counter()
    r.render(first, args.steps)
    r.synchronize()
    if world > 1:
        frame = gather(fb)
    sync()
    dt = time.perf_counter() - t0

    st = r.stats()
    rb_local = int(st.ray_bounces)
    live_in = [int(x) for x in st.live_in[:args.depth]]
    reduce_dev = "cpu" if rehearsal else dev
    tens = torch.tensor([dt, float(rb_local), float(st.gpu_ms)], dtype=torch.float64, device=reduce_dev)
    gather_check = None
    if world > 1:
        tmax = tens.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = tens.clone()
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        dt_max, rb_total = float(tmax[0]), float(tsum[1])
        # the gathered frame against the ranks' own tiles: integer checksum of every tile's bit patterns
        own = torch.tensor([tile_checksum(fb[:Hband])], dtype=torch.int64, device=reduce_dev)
        sums = [torch.zeros_like(own) for _ in range(world)]
        dist.all_gather(sums, own)
        if rank == 0:
            ok = True
            for k in range(world):
                if strips:
                    rows = torch.as_tensor(sharding.strip_global_rows(Hfull, world, k), dtype=torch.long, device=frame.device)
                    part = frame.index_select(0, rows)
                else:
                    k0, k1 = sharding.band_rows(Hfull, world, k)
                    part = frame[k0:k1]
                ok = ok and tile_checksum(part) == int(sums[k].item())
            gather_check = "frame == rank tiles (integer checksum per rank)" if ok else "MISMATCH"
            if not ok:
                print("bench.py: the gathered frame does not match the ranks' tiles", file=sys.stderr, flush=True)
        # second region (reported as value_pipelined_gather, never as `value`): a renderer in steady state hands frame f's
        # tiles to RCCL while frame f + 1 renders -- the gather of a snapshot of the finished frame runs on RCCL's stream
        # beside the next K steps
        r.reset_stats()
        prev_frame_tile = fb.clone()
        sync()
        t0p = time.perf_counter()
        frame2 = gather(prev_frame_tile)     # enqueued first (host-blocking only under the gloo rehearsal)
        r.render(first + args.steps, args.steps)
        r.synchronize()
        sync()
        dtp = time.perf_counter() - t0p
        st2 = r.stats()
        tp = torch.tensor([dtp], dtype=torch.float64, device=reduce_dev)
        dist.all_reduce(tp, op=dist.ReduceOp.MAX)
        rp = torch.tensor([float(st2.ray_bounces)], dtype=torch.float64, device=reduce_dev)
        dist.all_reduce(rp, op=dist.ReduceOp.SUM)
        value_pipelined = float(rp[0]) / float(tp[0]) / 1e6
        del frame2
        next_free = first + 2 * args.steps
    else:
        dt_max, rb_total = dt, float(rb_local)
        value_pipelined = None
        next_free = first + args.steps

    if rank == 0:
        npix = W * Hband
        alg_bytes = pkg.algorithmic_bytes(npix, live_in, args.steps)      # this rank, the timed K steps
        # Dominant kernel k_bounce.  Two launch sequences are in flight (batch n + 1 on a second stream beside batch n), so
        # the kernel is priced over the TIMED REGION: its launches' algorithmic bytes / the HIP-event time of the region on
        # the render stream (the library's own event pair around the pt_render call; it spans both sequences and includes
        # the batches' accumulate and bookkeeping kernels, ~4 %: conservative).  A launch's own duration overlaps its
        # neighbour's; `kernel_alone` below is the per-launch figure with one sequence (eager launches, one event pair each).
        li = r.launch_info()                   # what the library actually chose (not the options, not the environment)
        lib_batch = li.batch
        nb_timed = (args.steps + lib_batch - 1) // lib_batch
        timed_batches = [args.steps // nb_timed + (1 if j < args.steps % nb_timed else 0) for j in range(nb_timed)]
        launches_timed = int(st.bounce_launches)      # one launch carries one bounce of one batch (the camera kernel: bounces 0 and 1)
        region_ms = float(st.gpu_ms)
        achieved = alg_bytes / (region_ms * 1e-3) / 1e9
        prof_steps = timed_batches[0]          # one batch of the timed region's size: the same launch shape
        r.reset_stats()
        bounce_ms = r.render_profiled(next_free, prof_steps)
        pst = r.stats()
        p_live = [int(x) for x in pst.live_in[:args.depth]]
        p_bytes = pkg.algorithmic_bytes(npix, p_live, prof_steps)
        p_ms = sum(bounce_ms)
        p_launches = int(pst.bounce_launches)
        # the profiled batch kernel by kernel: algorithmic bytes of the bounces a launch traces (generate + the records its bounces
        # read and write + the pixel read-modify-write of the paths that END in it) over its own event time
        per_kernel = []
        first_b = list(range(args.depth)) if not li.resident else [0, 1]
        for k, b0 in enumerate(first_b):
            b1 = (b0 + 1) if (not li.resident or b0 == 0) else args.depth          # bounces [b0, b1) run in this launch
            nxt = lambda b: p_live[b] if b < args.depth else 0
            kb = sum((p_live[b] + nxt(b + 1)) * 40 for b in range(b0, b1)) + (p_live[b0] - nxt(b1)) * 24
            if b0 == 0:
                kb += prof_steps * npix * 40
            ms = bounce_ms[k]
            per_kernel.append({"kernel": "k_bounce<FIRST> (camera + bounce 0)" if b0 == 0 else
                               (f"k_bounce<RESIDENT> (bounces 1..{args.depth - 1})" if li.resident else f"k_bounce (bounce {b0})"),
                               "ms": ms, "bytes": kb, "achieved": kb / (ms * 1e-3) / 1e9 if ms > 0 else None,
                               "frac": kb / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS if ms > 0 else None})
        # HBM bytes from the PMC counters cannot be collected inside this run (rocprofv3 --pmc, one pass per counter
        # group): the figure is the STORED result of the last collection for this config (profiles/collect_pmc.py), kept
        # per iteration-bounce and scaled to the iterations one launch of THIS run carries
        traffic, traffic_source, flops_exec = None, None, None
        iters_per_launch = args.steps / nb_timed
        if os.path.exists(args.traffic_json) and not args.custom and world == 1:
            try:
                with open(args.traffic_json) as f:
                    ent = json.load(f).get(f"config{args.config}")
                if ent:
                    if ent.get("hbm_bytes_per_iteration"):      # (round 4 on: per iteration, whatever the launch structure)
                        traffic = ent["hbm_bytes_per_iteration"] * args.steps / launches_timed
                    else:
                        traffic = ent["hbm_bytes_per_launch"] / ent.get("iterations_per_launch", 16) * iters_per_launch
                    traffic_source = "stored: " + str(ent.get("source"))
                    flops_exec = ent.get("executed_fp32_flops_per_ray_bounce")
            except Exception:
                traffic = None
        tile_note = ", interleaved 8-row strips" if strips else ""
        value = rb_total / dt_max / 1e6
        out = {
            "metric": "ray-bounces/sec",
            "value": value,
            "unit": "Mray-bounces/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "warmup_settle": {"extra_untimed_iterations": settle_iters + (args.steps if value_cold is not None else 0),
                              "settle_ms": args.settle_ms, "settle_max_ms": args.settle_max_ms,
                              "last_chunks_ms": [round(x * 1e3, 3) for x in chunk_s[-4:]],
                              "why": "GPU clock ramp (DVFS); the timed region is exactly `steps` iterations; value_cold = the same "
                                     "K steps right behind the W warm-up steps"},
            "value_cold": value_cold,
            "ms_per_step": dt_max / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": args.scaling if world > 1 else "weak",
            "vs_baseline": None,
            "vs_baseline_note": "BASELINE.json `published` is empty (the reference renders noise and quotes no number); "
                                "vs_cpu_baseline = value / cpu_baseline.value",
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"BASELINE config {args.config}{' (modified)' if args.custom else ''}: {args.scene} {W}x{Hfull} "
                            f"({world} tile(s) of ~{W}x{Hmax} rows{tile_note}), {args.steps} spp, "
                            f"{args.depth} bounces, {CONFIGS[args.config]['lobes']}, rotat={args.rotat}, rr_start={args.rr_start}",
                "baseline_config": args.config, "scene": args.scene, "width": W, "height": Hfull, "depth": args.depth, "spp": args.steps,
                "rotat_units": args.rotat, "primitives": sc.n_objects, "materials": sc.n_materials,
                "compaction": 0 if args.no_compaction else args.compaction, "iteration_batch": lib_batch, "timed_batches": timed_batches,
                "launch_sequences_in_flight": li.sequences, "geom_path": li.geom_path, "workgroup": li.workgroup, "grid": li.grid,
                "resident_paths": bool(li.resident), "slab_pretest": bool(li.slab_pretest), "bounce_launches_per_batch": li.launches_per_batch, "lds_bytes_per_workgroup": li.lds_bytes,
                "hip_graph": not args.no_graph, "direct_light": bool(args.direct_light),
                "parallelism": (f"pixel-strips x{world}" if strips else f"pixel-bands x{world}") +
                               (", 1 RCCL gather of the rendered frame behind the timed steps" if world > 1 else ""),
                "gather_ms_standalone": gather_ms,
                "gather_check": gather_check,
            },
            "ray_bounces": int(rb_total),
            "shadow_rays": int(st.shadow_rays),
            "live_in_per_bounce": live_in,
            "ms_per_frame_1spp": dt_max / args.steps * 1e3,
            "total_ms": dt_max * 1e3,
            "gpu_event_ms": region_ms,
            "algorithmic_bytes_timed_region": alg_bytes,
            "algorithmic_GBs_whole_job": alg_bytes / dt_max / 1e9,
            "roofline": {
                "bound": "hbm",
                "kernel": "pt::k_bounce (resident-path instance: bounces 1..depth-1 in one launch; camera instance: bounce 0)" if li.resident else "pt::k_bounce",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "traffic_source": traffic_source,
                "bytes_per_launch": alg_bytes / launches_timed,
                "avg_launch_ms": region_ms / launches_timed,
                "launches_measured": launches_timed,
                "iterations_per_launch": iters_per_launch,
                "bytes_per_ray_bounce": alg_bytes / max(1, sum(live_in)),
                "kernel_alone": {"avg_launch_ms": p_ms / p_launches, "bytes_per_launch": p_bytes / p_launches,
                                 "achieved": p_bytes / (p_ms * 1e-3) / 1e9, "frac": p_bytes / (p_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                 "launches_measured": p_launches,
                                 "per_kernel": per_kernel if (li.resident or args.depth <= 8) else per_kernel[:2] + per_kernel[-1:],
                                 "how": "one sequence, eager launches, one HIP event pair per launch (pt_render_profiled)"},
                "note": "achieved = algorithmic bytes of the timed region's k_bounce launches (SURVEY 8(d): P*40 + "
                        "sum_b(live_in+live_out)*40 + P*24 per iteration) / HIP-event time of the timed region on the render "
                        "stream; avg_launch_ms = that time / launches: two launch sequences overlap, so a launch's own "
                        "duration in a rocprofv3 kernel trace is longer than this -- profiles/trace_union.py turns a trace "
                        "into the same figure (union of the k_bounce intervals / launches).  No single resource binds the "
                        "kernel (sensitivity runs, profiles/r03/knockout.txt): see DESIGN.md 5.3",
            },
        }
        if value_pipelined is not None:
            out["value_serial_gather"] = value          # (= `value`: the gather of the rendered frame behind the K steps)
            out["value_pipelined_gather"] = value_pipelined
            out["value_pipelined_gather_note"] = ("second region: the previous frame's tiles gathered on RCCL's stream while the "
                                                  "next K steps render (steady state of an animation); `value` has the gather of "
                                                  "the rendered frame behind the K steps")
        # the physically binding roof is fp32 VALU issue.  nominal = 95*nG + 150 flops per ray-bounce (the
        # survey's count of a test-everything renderer); executed = fp32 add/mul/fma instructions x active lanes (FMA = 2)
        # from the stored PMC pass of this config, per ray-bounce.  Peak 157.3 TFLOP/s counts FMA as 2 flops; the path's
        # arithmetic is mul/add without contraction (bit parity), so at most half of it is reachable by construction.
        flops_rb = 95 * sc.n_objects + 150
        rate = sum(live_in) / (region_ms * 1e-3)
        out["valu_roofline"] = {"bound": "valu_fp32", "achieved": rate * flops_rb / 1e12, "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                                "frac": rate * flops_rb / 1e12 / VALU_PEAK_TFLOPS, "flops_per_ray_bounce": flops_rb,
                                "kind": "nominal (95*nG + 150 per ray-bounce)",
                                "executed": None if flops_exec is None else
                                {"achieved": rate * flops_exec / 1e12, "frac": rate * flops_exec / 1e12 / VALU_PEAK_TFLOPS,
                                 "flops_per_ray_bounce": flops_exec,
                                 "how": "stored PMC pass: (SQ_INSTS_VALU_ADD_F32 + MUL_F32 + 2*FMA_F32) x average active lanes"}}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args, scene_path, rotat)
            out["vs_cpu_baseline"] = value / out["cpu_baseline"]["value"]
        print(json.dumps(out), flush=True)

    r.close()
    if world > 1:
        barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
