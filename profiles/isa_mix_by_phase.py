#!/usr/bin/env python3
"""Static VALU mix of one k_bounce instance by source region (which part of the kernel the INT32 share of the
VALU instructions comes from).  Compiles a bounce unit with line tables, walks the instance's assembly and books every VALU
instruction on the innermost source line its `.loc` names:

    python3 profiles/isa_mix_by_phase.py [unit=4] [instance regex]  > profiles/r04/int32_by_phase.txt

Regions are line ranges of pt_bounce.h / pt_device.h (looked up by function name at run time, so the table follows the sources).
Static counts: the pre-test loop body runs once per primitive and a pair-batch body once per batch, so the per-trip weights given
in the output (9 primitives, 1.1 full + 1 mixed batch per trip on config 2: profiles/r04/lane_budget.txt) turn them into a
dynamic estimate."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "project3-pathtracer_amd", "csrc")
FLAGS = ("--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-gpu-flush-denormals-to-zero "
         "-fno-slp-vectorize --cuda-device-only -gline-tables-only -S").split()

INT32 = re.compile(r"^v_(add|sub|subrev|mul_lo|mul_hi|mad|lshl|lshr|ashr|and|or|xor|not|bfe|bfi|mbcnt|min_u|max_u|min_i|max_i|alignbit|"
                   r"add3|lshl_add|lshl_or|and_or|or3|xad|add_lshl|perm|cvt_u32|cvt_f32_u32|cvt_f32_i32|cvt_i32|sad|bcnt|ffb)"
                   r"[a-z0-9_]*(u32|i32|b32|u16|i16|u24|i24|u64|b64|_f32_u32|_f32_i32|u32_f32|i32_f32)?")
FP32 = re.compile(r"^v_(add|sub|subrev|mul|fma|mac|fmac|mad)_(f32|legacy_f32)")
FPMISC = re.compile(r"^v_(min|max|min3|max3|med3)_f32|^v_(rcp|rsq|sqrt|exp|log|sin|cos|floor|fract|trunc|rndne|ceil|ldexp|frexp)[a-z_]*f32")


def cls(op):
    if op.startswith("v_cmp") or op.startswith("v_cmpx"):
        return "compare"
    if op.startswith("v_cndmask"):
        return "select"
    if op.startswith("v_mov") or op.startswith("v_readlane") or op.startswith("v_readfirstlane") or op.startswith("v_writelane") or op.startswith("v_accvgpr"):
        return "move"
    if FP32.match(op):
        return "fp32 add/mul/fma"
    if FPMISC.match(op):
        return "fp32 min/max/trans"
    if op.startswith("v_cvt_f32_u32") or op.startswith("v_cvt_f32_i32") or op.startswith("v_cvt_u32_f32") or op.startswith("v_cvt_i32_f32"):
        return "int32"
    if re.match(r"^v_.*(_u32|_i32|_b32|_u16|_i16|_u24|_i24|_b64|_u64)(_e32|_e64|_dpp|_sdwa)?$", op) or op.startswith("v_mbcnt"):
        return "int32"
    return "other"


def fn_ranges(path):
    """[(first line, last line, name)] of the __device__ / __global__ functions of a header (brace matching on column 0)."""
    lines = open(path).read().split("\n")
    out, i = [], 0
    while i < len(lines):
        m = re.match(r"^(?:template.*\n)?(?:__host__ )?(?:__device__|__global__)[^;]*?(\w+)\(", lines[i]) or \
            re.match(r"^(?:__host__ )?(?:__device__|__global__).*?\b(\w+)\(", lines[i])
        if m and not lines[i].rstrip().endswith(";"):
            name = m.group(1)
            j = i
            if "{" in lines[i] and lines[i].count("{") == lines[i].count("}"):
                out.append((i + 1, i + 1, name))
                i += 1
                continue
            while j < len(lines) and lines[j] != "}":
                j += 1
            out.append((i + 1, j + 1, name))
            i = j + 1
        else:
            i += 1
    return out


def main():
    unit = sys.argv[1] if len(sys.argv) > 1 else "4"
    pat = sys.argv[2] if len(sys.argv) > 2 else r"_ZN2pt8k_bounceILi256ELb0ELi%sELi1ELi8EEEvNS_7KParamsEi" % unit
    out = os.path.join(tempfile.gettempdir(), f"ptmix_g{unit}.s")
    subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + ["-o", out, os.path.join(CSRC, f"pt_bounce_g{unit}.hip")], check=True,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    dev = fn_ranges(os.path.join(CSRC, "pt_device.h"))
    bnc = fn_ranges(os.path.join(CSRC, "pt_bounce.h"))
    src_b = open(os.path.join(CSRC, "pt_bounce.h")).read().split("\n")

    def find(text, start=0):
        for k in range(start, len(src_b)):
            if text in src_b[k]:
                return k + 1
        raise KeyError(text)
    kb0 = find("void k_bounce(const KParams p, const int bounce)")
    refill0 = find("// refill: lanes without a path take the next rays")
    trace0 = find("bool alive = false;", refill0)
    shade0 = find("f3 L = mk(0, 0, 0);", trace0)
    tail0 = find("const unsigned long long c3 = ", shade0)
    pre0 = find("for (int g = 0; g < p.nG; ++g) {", find("Hit nearestHitPairs("))
    pre1 = find("const uint64_t dbg_valid = DEBUG_PAIR", pre0)
    nhp1 = [r for r in bnc if r[2] == "nearestHitPairs"][0][1]
    files, cur = {}, (0, 0)
    rng = {"wang_hash", "stream_key", "minstd_seed", "minstd_next", "minstd_jump", "u01_of", "uniform_real"}
    isect = {"candidateT", "hitPoint", "mulMV", "pointOnRay", "normalize_unit", "sqrt_rn", "rcp_rn", "rsqrt_rn", "rsqrt_near_one", "sqrt_core", "rcp_core",
             "length", "normalize", "dot", "cross", "mk", "sphereNormal", "boxNormal"}

    def region(fid, line):
        f = files.get(fid, "")
        if f.endswith("pt_device.h"):
            for a, b, n in dev:
                if a <= line <= b:
                    if n in rng:
                        return "RNG (hash, minstd, u01)"
                    if n in ("randomDirectionInHemisphere", "sincos_poly", "reflectionDirection", "transmissionDirection", "fresnelReflectance"):
                        return "BSDF sampling (pt_device.h)"
                    return "vector / exact math helpers (pt_device.h)"
            return "pt_device.h (other)"
        if f.endswith("pt_bounce.h"):
            for a, b, n in bnc:
                if a <= line <= b and n == "pairBatch":
                    return "pair batch: gathers, keys, atomic min"
                if a <= line <= b and n == "approxInverse":
                    return "pre-test: per-ray setup"
            if pre0 <= line < pre1:
                return "pre-test loop body (x primitives)"
            if pre1 <= line <= nhp1 or ([r for r in bnc if r[2] == "nearestHitPairs"][0][0] <= line < pre0):
                return "nearest hit: setup, last batches' dispatch, result"
            if refill0 <= line < trace0:
                return "refill / chunk draw"
            if shade0 <= line < tail0:
                return "shading body (materials, lobes, radiance write)"
            if line >= tail0:
                return "trip end / histogram / kernel end"
            if kb0 <= line < refill0:
                return "kernel prologue (LDS staging, tables)"
            return "pt_bounce.h (other)"
        return "runtime headers (ballot, mbcnt, atomics)"

    table, inside, parent = {}, False, None
    for ln in open(out):
        m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"\s+"([^"]*)"', ln)
        if m:
            files[int(m.group(1))] = m.group(3)
            continue
        if re.match(pat + ":", ln):
            inside = True
            continue
        if inside and ln.startswith("\ts_endpgm"):
            break
        if not inside:
            continue
        m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", ln)
        if m:
            cur = (int(m.group(1)), int(m.group(2)))
            continue
        m = re.match(r"\s+(v_[a-z0-9_]+)", ln)
        if m:
            r = region(*cur)
            # an inlined helper called from a pair batch / the shading body is booked on the helper, with its caller unknown to a line
            # table: the exact-test arithmetic (candidateT, hitPoint) only ever runs inside pair batches
            d = table.setdefault(r, {})
            c = cls(m.group(1).replace("_e32", "").replace("_e64", "").replace("_dpp", "").replace("_sdwa", ""))
            d[c] = d.get(c, 0) + 1
    cols = ["fp32 add/mul/fma", "fp32 min/max/trans", "int32", "compare", "select", "move", "other"]
    print(f"# static VALU instructions of {pat} by source region ({os.path.basename(out)}: hipcc {' '.join(FLAGS[1:9])} ...)")
    print(f"{'region':62s}" + "".join(f"{c:>20s}" for c in cols) + f"{'all':>8s}")
    tot = {c: 0 for c in cols}
    for r, d in sorted(table.items(), key=lambda kv: -sum(kv[1].values())):
        print(f"{r:62s}" + "".join(f"{d.get(c, 0):20d}" for c in cols) + f"{sum(d.values()):8d}")
        for c in cols:
            tot[c] += d.get(c, 0)
    print(f"{'all':62s}" + "".join(f"{tot[c]:20d}" for c in cols) + f"{sum(tot.values()):8d}")


if __name__ == "__main__":
    main()
