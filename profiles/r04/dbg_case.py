# scratch: re-run a range of fuzz cases one by one (python -u profiles/r04/dbg_case.py first last)
import sys
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np
import fuzz_repro
a, b = int(sys.argv[1]), int(sys.argv[2])
for case in range(a, b + 1):
    rng4 = np.random.default_rng(770000 + case)
    rng4.choice([-1, 1, 1]); rng4.choice([1, 4, 16, 33, 64])
    kind = "stress" if rng4.random() < 0.2 else (("far", "huge", "tiny", "needle", "zero", "neg")[case % 6] if rng4.random() < 0.1 else "random")
    print("==", case, kind, flush=True)
    try:
        fuzz_repro.run(case, {})
    except Exception as e:
        print("  (repro cannot build this case:", str(e)[:80], ")", flush=True)
