#!/usr/bin/env python3
"""Per-kernel resource table of the product build: VGPRs, SGPRs, scratch, code bytes and the compiler's occupancy figure of every
k_bounce instance the library can select (compaction 1, workgroups of 256 and 512), read from the gfx950 assembly hipcc emits with
the Makefile's flags.  python3 profiles/kernel_resources.py > profiles/r03/kernel_resources.txt   (8 compiles, ~1 min each, in parallel)"""
import concurrent.futures as cf
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "project3-pathtracer_amd", "csrc")
FLAGS = "--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-gpu-flush-denormals-to-zero -fno-slp-vectorize --cuda-device-only -S".split()
GEOM = ["scalar", "LDS", "hit queue", "per-lane walk", "pair queue", "walk + pairs", "batched walk", "batched walk, nodes in L1/L2"]
FEAT = {0: "plain", 1: "direct lighting", 2: "scattering", 3: "direct lighting + scattering", 4: "per-ray shutter time",
        5: "shutter time + direct lighting", 6: "shutter time + scattering", 7: "shutter time + both",
        8: "resident paths", 9: "resident + direct lighting", 10: "resident + scattering", 11: "resident + both",
        16: "plain, slab pre-test", 24: "resident paths, slab pre-test"}


def one(g):
    out = os.path.join(tempfile.gettempdir(), f"ptres_g{g}.s")
    subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + ["-o", out, os.path.join(CSRC, f"pt_bounce_g{g}.hip")], check=True,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    rows, name, cur = [], None, {}
    for line in open(out):
        m = re.match(r"^(_ZN2pt8k_bounceILi(\d+)ELb(\d)ELi(\d)ELi(\d)ELi(\d+)EEEvNS_7KParamsEi):", line)
        if m:
            name, cur = m.groups()[1:], {}
            continue
        if name:
            for key, pat in (("code", r"; codeLenInByte = (\d+)"), ("sgpr", r"; TotalNumSgprs: (\d+)"), ("vgpr", r"; NumVgprs: (\d+)"),
                             ("scratch", r"; ScratchSize: (\d+)"), ("occ", r"; Occupancy: (\d+)"), ("lds", r"; LDSByteSize: (\d+)")):
                mm = re.match(pat, line)
                if mm:
                    cur[key] = int(mm.group(1))
            if "occ" in cur and "lds" in cur:
                wg, first, geom, compact, feat = (int(x) for x in name)
                if compact == 1 and wg in (256, 512):
                    rows.append((geom, feat, wg, first, cur))
                name = None
    return rows


def main():
    with cf.ThreadPoolExecutor(4) as ex:
        allrows = [r for rows in ex.map(one, range(8)) for r in rows]
    print("# k_bounce<WG, FIRST, GEOM, 1, FEAT> instances of the product build (compaction 1): resources from the gfx950 assembly (hipcc -S with the")
    print("# Makefile's flags).  occupancy = waves per SIMD the register counts allow (512 VGPRs per SIMD lane); static LDS is the key tables only,")
    print("# the dynamic part is sized by the host (pt_kernels.hip bounce_lds_bytes).  Library defaults: pair queue WG 256 (<= 40 primitives), batched walk WG 512.")
    print(f"{'geometry path':30s} {'feature':30s} {'WG':>4s} {'kernel':>7s} {'VGPR':>5s} {'SGPR':>5s} {'scratch':>8s} {'code B':>7s} {'occ':>4s}")
    for geom, feat, wg, first, c in sorted(allrows):
        print(f"{str(geom + 1) + ' ' + GEOM[geom]:30s} {FEAT.get(feat, str(feat)):30s} {wg:4d} {'camera' if first else 'later':>7s} {c['vgpr']:5d} {c['sgpr']:5d} "
              f"{c['scratch']:8d} {c['code']:7d} {c['occ']:4d}")


if __name__ == "__main__":
    sys.exit(main())
