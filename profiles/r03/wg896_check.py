# experiment helper: the batched walk with 896-thread workgroups (variant build -DPT_WG896) against 512-thread ones, same library, bit for bit
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from __graft_entry__ import load_package
pkg = load_package()
from test_gpu_parity import gpu_render
ok = True
for scene, w, h, depth, iters in (("cloud256.txt", 320, 180, 12, 3), ("cloud256.txt", 1920, 1080, 32, 16)):
    a, la, _ = gpu_render(pkg, scene, w, h, depth, iters=iters, rotat=1, geom_path=7, workgroup=512)
    b, lb, _ = gpu_render(pkg, scene, w, h, depth, iters=iters, rotat=1, geom_path=7, workgroup=896)
    same = bool(np.array_equal(a, b)) and la == lb
    print(scene, w, h, depth, iters, "identical" if same else "DIFFERENT", flush=True)
    ok = ok and same
sys.exit(0 if ok else 1)
