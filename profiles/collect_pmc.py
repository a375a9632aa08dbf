#!/usr/bin/env python3
"""Collect rocprofv3 evidence for bench.py on the GPU box and summarise it.

Run on an MI355X box from the repo root (through gpurun):

    python3 profiles/collect_pmc.py --tag r01 [--steps 16] [bench.py flags after --]

Passes (each its own rocprofv3 run, as MI355X_MICROARCH.md prescribes: counters never share a run with
--kernel-trace/--stats, FETCH_SIZE and WRITE_SIZE never share a pass):
    trace   --kernel-trace --stats                     -> per-kernel durations
    sq1/sq2 --pmc SQ_* (8 SQ slots per pass)           -> instruction mix, VALU lane utilisation, wait shares
    fetch   --pmc FETCH_SIZE                           -> HBM read  KB per dispatch
    write   --pmc WRITE_SIZE                           -> HBM write KB per dispatch
    tcc     --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_ATOMIC_sum GRBM_GUI_ACTIVE

gfx950 corrections applied to `traffic` (MI355X_MICROARCH.md, HBM section): FETCH_SIZE is reported in KB and
counts 64 B per 128-B request for wide (16 B/lane) coalesced reads -> x2; WRITE_SIZE (KB) is exact for
16-B/lane streaming stores.  The ray pools are read/written as 16 B + 16 B + 8 B per lane; the 8-B part and
the scattered 12-B pixel read-modify-writes are uncalibrated, so both the raw and the corrected figures are
kept in the summary.

Output: gpurun_out/pmc_<tag>/summary.json (+ the raw CSVs); copy summary.json and the *_kernel_stats.csv
into profiles/ to have them judged.  Also writes profiles/traffic_latest.json-compatible content under
summary["traffic_for_bench"].
"""
import argparse
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

PASSES = {
    "sq1": ["SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_VMEM", "SQ_INSTS_LDS", "SQ_WAVE_CYCLES",
            "SQ_BUSY_CYCLES", "SQ_ACTIVE_INST_VALU"],
    "sq2": ["SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_THREAD_CYCLES_VALU", "SQ_ACTIVE_INST_SCA",
            "SQ_INSTS_SMEM", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_LDS"],
    "sq3": ["SQ_INSTS_VALU_TRANS_F32", "SQ_INSTS_VALU_FMA_F32", "SQ_INSTS_VALU_ADD_F32", "SQ_INSTS_VALU_MUL_F32",
            "SQ_INSTS_VALU_INT32", "SQ_INSTS_VALU_CVT", "SQ_INSTS_VALU_ADD_F64", "SQ_LDS_BANK_CONFLICT"],
    "fetch": ["FETCH_SIZE"],
    "write": ["WRITE_SIZE"],
    "tcc": ["TCC_HIT_sum", "TCC_MISS_sum", "TCC_EA0_ATOMIC_sum", "GRBM_GUI_ACTIVE"],
}


def run(cmd, cwd="/tmp"):
    env = dict(os.environ, TMPDIR="/tmp")
    print("+", " ".join(cmd), flush=True)
    res = subprocess.run(cmd, cwd=cwd, env=env, capture_output=True, text=True)
    if res.returncode != 0:
        print(res.stdout[-2000:])
        print(res.stderr[-2000:])
    return res


def short(name):
    if "k_bounce" in name:
        first = "true" in name.split(",")[1] if "," in name else False
        return "k_bounce<first>" if first else "k_bounce"
    for k in ("k_iter_begin", "k_iter_fold", "k_iter_set", "k_send_image_to_pbo"):
        if k in name:
            return k
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tag", default="r01")
    ap.add_argument("--steps", type=int, default=32)
    ap.add_argument("--warmup", type=int, default=16)     # one full batch, so every counted launch is a full-batch launch
    ap.add_argument("--passes", default="trace,sq1,sq2,sq3,fetch,write,tcc")
    ap.add_argument("bench_args", nargs="*")
    a = ap.parse_args()
    out = os.path.join(ROOT, "gpurun_out", f"pmc_{a.tag}")
    os.makedirs(out, exist_ok=True)
    bench = ["python3", os.path.join(ROOT, "bench.py"), "--steps", str(a.steps), "--warmup", str(a.warmup),
             "--no-cpu-baseline", "--settle-ms", "0", "--traffic-json", "/nonexistent"] + a.bench_args
    # the kernel-trace pass times bench.py's DEFAULT run (the command the driver runs, minus the CPU leg), so its
    # average k_bounce duration is comparable with roofline.avg_launch_ms of the committed bench line
    bench_trace = ["python3", os.path.join(ROOT, "bench.py"), "--no-cpu-baseline"] + a.bench_args
    summary = {"tag": a.tag, "bench_cmd": " ".join(bench[1:]), "trace_cmd": " ".join(bench_trace[1:]), "kernels": {}}

    for p in a.passes.split(","):
        d = os.path.join(out, p)
        os.makedirs(d, exist_ok=True)
        if p == "trace":
            cmd = ["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", d, "-o", p, "--"] + bench_trace
        else:
            cmd = ["rocprofv3", "--pmc"] + PASSES[p] + ["--output-format", "csv", "-d", d, "-o", p, "--"] + bench
        res = run(cmd)
        with open(os.path.join(d, "stdout.txt"), "w") as f:
            f.write(res.stdout)
        with open(os.path.join(d, "stderr.txt"), "w") as f:
            f.write(res.stderr[-20000:])
        if p == "trace":
            for line in res.stdout.splitlines():
                if line.startswith("{"):
                    try:
                        j = json.loads(line)
                        summary["bench_under_trace"] = {k: j[k] for k in ("value", "steps", "warmup", "ms_per_step", "roofline")}
                    except Exception:
                        pass
            # two launch sequences overlap: union of the k_bounce intervals (profiles/trace_union.py)
            for fn in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
                u = subprocess.run([sys.executable, os.path.join(ROOT, "profiles", "trace_union.py"), fn], capture_output=True, text=True)
                try:
                    summary["k_bounce_trace_union"] = json.loads(u.stdout.strip().splitlines()[-1])
                except Exception:
                    pass
            for fn in glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True):
                with open(fn) as f:
                    for row in csv.DictReader(f):
                        s = short(row["Name"])
                        if s:
                            k = summary["kernels"].setdefault(s, {})
                            k["calls"] = int(row["Calls"])
                            k["avg_us"] = float(row["AverageNs"]) / 1e3
                            k["total_ms"] = float(row["TotalDurationNs"]) / 1e6
                            k["pct"] = float(row["Percentage"])
        else:
            for line in res.stdout.splitlines():             # the bench line of the pass (ray-bounces per iteration)
                if line.startswith("{") and "bench_under_pmc" not in summary:
                    try:
                        j = json.loads(line)
                        summary["bench_under_pmc"] = {"ray_bounces": j["ray_bounces"], "steps": j["steps"],
                                                      "iteration_batch": j["config"]["iteration_batch"]}
                    except Exception:
                        pass
            for fn in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                with open(fn) as f:
                    for row in csv.DictReader(f):
                        s = short(row["Kernel_Name"])
                        if not s:
                            continue
                        k = summary["kernels"].setdefault(s, {}).setdefault("pmc", {})
                        c = row["Counter_Name"]
                        e = k.setdefault(c, {"sum": 0.0, "dispatches": 0})
                        e["sum"] += float(row["Counter_Value"])
                        e["dispatches"] += 1

    # derived figures for the bounce kernels (both template instances together)
    def total(counter):
        t, n = 0.0, 0
        for kn in ("k_bounce", "k_bounce<first>"):
            e = summary["kernels"].get(kn, {}).get("pmc", {}).get(counter)
            if e:
                t += e["sum"]
                n += e["dispatches"]
        return t, n

    der = {}
    fetch_kb, nf = total("FETCH_SIZE")
    write_kb, nw = total("WRITE_SIZE")
    if nf and nw:
        # per ITERATION (independent of how the bounces are cut into launches: one per bounce, or camera + one resident-path launch):
        # every batch of a PMC pass is a full one, and the camera kernel is dispatched once per batch
        ncam = summary["kernels"].get("k_bounce<first>", {}).get("pmc", {}).get("FETCH_SIZE", {}).get("dispatches", 0)
        batch = (summary.get("bench_under_pmc") or {}).get("iteration_batch", 16)
        if ncam:
            der["hbm_bytes_per_iteration"] = ((2.0 * fetch_kb + write_kb) * 1024) / (ncam * batch)
        der["launches"] = nf
        der["fetch_bytes_per_launch_raw"] = fetch_kb * 1024 / nf
        der["write_bytes_per_launch_raw"] = write_kb * 1024 / nw
        der["hbm_bytes_per_launch"] = (2.0 * fetch_kb * 1024) / nf + write_kb * 1024 / nw
        der["correction"] = "FETCH_SIZE KB x1024 x2 (gfx950: 64 B tallied per 128-B request on wide reads) + WRITE_SIZE KB x1024"
    valu, _ = total("SQ_INSTS_VALU")
    thr, _ = total("SQ_THREAD_CYCLES_VALU")
    act, _ = total("SQ_ACTIVE_INST_VALU")
    wc, _ = total("SQ_WAVE_CYCLES")
    if valu:
        der["valu_insts"] = valu
    for c in ("SQ_INSTS_LDS", "SQ_ACTIVE_INST_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_INSTS_SALU", "SQ_INSTS_VALU_INT32"):
        v, _ = total(c)
        if v:
            der[c] = v
    if thr and act:
        der["valu_active_lanes_avg"] = thr / act      # of 64
    for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU"):
        v, _ = total(c)
        if v and wc:
            der[c + "_over_WAVE_CYCLES"] = v / wc
    kb, kf = summary["kernels"].get("k_bounce", {}), summary["kernels"].get("k_bounce<first>", {})
    if kb.get("calls") and kf.get("calls"):
        der["trace_avg_launch_ms"] = (kb["total_ms"] + kf["total_ms"]) / (kb["calls"] + kf["calls"])
        der["trace_launches"] = kb["calls"] + kf["calls"]
    # executed fp32 work per ray-bounce: (add + mul + 2 fma instructions) x that kernel instance's average active lanes over
    # every k_bounce dispatch of the pass, over the ray-bounces those dispatches carried (every batch of a PMC pass is a
    # full one: steps and warm-up are multiples of the batch)
    bu = summary.get("bench_under_pmc")
    if bu:
        flops, nfirst = 0.0, 0
        for kn in ("k_bounce", "k_bounce<first>"):
            pm = summary["kernels"].get(kn, {}).get("pmc", {})
            if not all(c in pm for c in ("SQ_INSTS_VALU_ADD_F32", "SQ_INSTS_VALU_MUL_F32", "SQ_INSTS_VALU_FMA_F32",
                                         "SQ_THREAD_CYCLES_VALU", "SQ_ACTIVE_INST_VALU")):
                flops = None
                break
            lanes = pm["SQ_THREAD_CYCLES_VALU"]["sum"] / pm["SQ_ACTIVE_INST_VALU"]["sum"]
            flops += (pm["SQ_INSTS_VALU_ADD_F32"]["sum"] + pm["SQ_INSTS_VALU_MUL_F32"]["sum"] + 2.0 * pm["SQ_INSTS_VALU_FMA_F32"]["sum"]) * lanes
            if kn.endswith(">"):
                nfirst = pm["SQ_INSTS_VALU_FMA_F32"]["dispatches"]
        if flops and nfirst:
            der["executed_fp32_flops_per_ray_bounce"] = flops / (nfirst * bu["iteration_batch"] * bu["ray_bounces"] / bu["steps"])
    summary["k_bounce_derived"] = der
    if "hbm_bytes_per_launch" in der:
        summary["traffic_for_bench"] = {"hbm_bytes_per_launch": der["hbm_bytes_per_launch"], "iterations_per_launch": (bu or {}).get("iteration_batch", 16),
                                        "hbm_bytes_per_iteration": der.get("hbm_bytes_per_iteration"),
                                        "executed_fp32_flops_per_ray_bounce": der.get("executed_fp32_flops_per_ray_bounce"),
                                        "source": f"pmc_{a.tag}"}
    with open(os.path.join(out, "summary.json"), "w") as f:
        json.dump(summary, f, indent=1)
    print(json.dumps(summary.get("k_bounce_derived"), indent=1))
    return 0


if __name__ == "__main__":
    sys.exit(main())
