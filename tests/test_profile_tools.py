"""Host-side evidence tools (no GPU): profiles/trace_union.py turns a rocprofv3 kernel trace into the figure bench.py's
roofline block reports when two launch sequences overlap."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_trace_union_merges_overlapping_dispatches(tmp_path):
    rows = ['"Kind","Agent_Id","Queue_Id","Kernel_Name","Start_Timestamp","End_Timestamp"']
    # two streams: launches of 100 ns each, the second stream's shifted by 50 ns -> union 450 ns over 8 launches
    for k in range(4):
        rows.append(f'"KERNEL_DISPATCH",1,1,"void pt::k_bounce<256, false, 4, 1, 0>(pt::KParams, int)",{1000 + 100 * k},{1100 + 100 * k}')
        rows.append(f'"KERNEL_DISPATCH",1,2,"void pt::k_bounce<256, true, 4, 1, 0>(pt::KParams, int)",{1050 + 100 * k},{1150 + 100 * k}')
    rows.append('"KERNEL_DISPATCH",1,1,"pt::k_accumulate(float*)",5000,9000')
    p = tmp_path / "t_kernel_trace.csv"
    p.write_text("\n".join(rows) + "\n")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "profiles", "trace_union.py"), str(p)], capture_output=True, text=True, check=True)
    j = json.loads(out.stdout.strip().splitlines()[-1])
    assert j["dispatches"] == 8
    assert abs(j["sum_ms"] - 800e-6) < 1e-12 and abs(j["union_ms"] - 450e-6) < 1e-12
    assert abs(j["eff_ms"] - 450e-6 / 8) < 1e-12 and abs(j["overlap"] - 800 / 450) < 1e-9
    out = subprocess.run([sys.executable, os.path.join(ROOT, "profiles", "trace_union.py"), str(p), "k_accumulate"], capture_output=True, text=True, check=True)
    assert json.loads(out.stdout.strip().splitlines()[-1])["dispatches"] == 1
