# scratch: which configuration of the stress scenes differs from the oracle
import os, sys, ctypes as C, numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import importlib
pkg = importlib.import_module("project3-pathtracer_amd")
import oracle_lib as O
import test_gpu_parity as T
for seed in (0, 1, 2):
    geoms, mats, eye, view, up, fovy = T._skip_stress_scene(7000 + seed)
    W, H, depth, iters = 96, 54, 14, 3
    ga = (O.StaticGeom * len(geoms))(*geoms); ma = (O.Material * len(mats))(*mats)
    cam = O.make_camera(W, H, eye, view, up, fovy)
    ref, live = O.render(ga, len(geoms), ma, len(mats), cam, depth, iters=iters, rr_start=-1, seed=seed)
    for name, env, opts in (("perbounce", {}, dict(resident=-1)), ("res_noskip", {"PT_NO_SELF_SKIP": "1"}, dict(resident=1)), ("res_skip", {}, dict(resident=1)),
                            ("perbounce_d2", {}, dict(resident=-1, depth=2))):
        for k in ("PT_NO_SELF_SKIP",): os.environ.pop(k, None)
        os.environ.update(env)
        o = dict(depth=depth, seed=seed, geom_path=5, batch=3); o.update(opts)
        with pkg.Renderer(0) as r:
            r.set_options(**o)
            r.set_scene(C.cast(ga, C.POINTER(pkg.StaticGeom)), len(geoms), C.cast(ma, C.POINTER(pkg.Material)), len(mats))
            r.set_camera(pkg.CameraData.from_buffer_copy(cam))
            r.clear_image(); r.render(1, iters)
            img = r.download_image(); st = r.stats()
        if o["depth"] != depth:
            ref2, live2 = O.render(ga, len(geoms), ma, len(mats), cam, o["depth"], iters=iters, rr_start=-1, seed=seed)
        else: ref2, live2 = ref, live
        d = np.abs(img - ref2)
        print(seed, name, "n", len(geoms), "err", float(d.max()), "pixels", int((d.max(axis=-1) > 0).sum()), "live", [int(x) for x in st.live_in[:4]], [int(x) for x in live2[:4]], flush=True)
