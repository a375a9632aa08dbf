"""Known-answer tests: the CPU oracle against the reference's golden vectors.

Pins the oracle (oracle/pt_oracle.c) to
  * tests/golden/reference_vectors.json  -- SURVEY.md Appendix B, values captured from the reference's
    implemented functions (printed with %.9g, which round-trips fp32 => compared bit for bit unless noted);
  * tests/golden/glm_vectors.json        -- real GLM 0.9.5.4 (the reference's vendored copy);
  * tests/golden/thrust_rng_vectors.json -- real thrust engines (rocThrust).
No GPU needed.
"""
import ctypes as C
import json
import math
import os

import numpy as np
import pytest

import oracle_lib as O


def f32(x):
    return float(np.float32(x))


def assert_bits(got, want, what=""):
    g, w = np.float32(got), np.float32(want)
    assert g.view(np.uint32) == w.view(np.uint32), f"{what}: got {float(g)!r} want {float(w)!r}"


def trs_geom(trs, gtype=O.SPHERE):
    return O.make_geom(gtype, 0, trs["t"], trs["r"], trs["s"], O.ROTAT_RADIANS)


def normalize32(v):
    # glm::normalize in fp32: x * (1/sqrt(x.x+y.y+z.z))
    v = [np.float32(c) for c in v]
    sqr = np.float32(np.float32(np.float32(v[0] * v[0]) + np.float32(v[1] * v[1])) + np.float32(v[2] * v[2]))
    inv = np.float32(np.float32(1.0) / np.sqrt(sqr))
    return [float(np.float32(c * inv)) for c in v]


# ------------------------------------------------------------------ RNG
def test_hash(oracle, golden):
    L = oracle.lib()
    for a, want in golden["reference_vectors"]["hash"]:
        assert L.o_hash(a) == want


def test_minstd_u01_reference(oracle, golden):
    L = oracle.lib()
    g = golden["reference_vectors"]["minstd_u01"]
    st = C.c_uint()
    L.o_minstd_seed(C.byref(st), L.o_hash(g["hash_arg"]))
    for want in g["u01"]:
        assert_bits(L.o_u01(C.byref(st)), want, "u01")


def test_minstd_against_thrust(oracle, golden):
    """seed(), operator() and uniform_real_distribution<float> bit for bit vs the real library."""
    L = oracle.lib()
    t = golden["thrust_rng_vectors"]
    assert (t["min"], t["max"]) == (1, 2147483646)
    for e in t["engine"]:
        st = C.c_uint()
        L.o_minstd_seed(C.byref(st), e["seed"])
        assert [L.o_minstd_next(C.byref(st)) for _ in e["raw"]] == e["raw"]
        L.o_minstd_seed(C.byref(st), e["seed"])
        assert [O.bits(L.o_u01(C.byref(st))) for _ in e["u01"]] == e["u01"]
        L.o_minstd_seed(C.byref(st), e["seed"])
        assert [O.bits(L.o_uniform_real(C.byref(st), -0.5, 0.5)) for _ in e["u02"]] == e["u02"]


def test_u01_range(oracle):
    L = oracle.lib()
    st = C.c_uint()
    L.o_minstd_seed(C.byref(st), 12345)
    v = np.array([L.o_u01(C.byref(st)) for _ in range(20000)])
    assert v.min() >= 0.0 and v.max() <= 1.0
    assert abs(v.mean() - 0.5) < 0.01


def test_noise(oracle, golden):
    L = oracle.lib()
    for n in golden["reference_vectors"]["noise"]:
        got = L.o_generateRandomNumberFromThread(O.Vec2(*map(float, n["res"])), float(n["time"]), n["x"], n["y"])
        for g, w in zip(got.tup(), n["rgb"]):
            assert_bits(g, w, "noise")


# ------------------------------------------------------------------ GLM restatement
def test_transform_builder_vs_glm(oracle, golden):
    L = oracle.lib()
    for e in golden["glm_vectors"]["trs"]:
        t, r, s = ([O.from_bits(u) for u in e[k]] for k in ("translation", "rotation", "scale"))
        inv = O.Mat4()
        m = L.o_buildTransformationMatrix(O.v3(t), O.v3(r), O.v3(s), O.ROTAT_RADIANS, C.byref(inv))
        got = [O.bits(x) for row in m.rows() for x in row]
        assert got == e["transform"], (t, r, s)
        goti = [O.bits(x) for row in inv.rows() for x in row]
        assert goti == e["inverse"], (t, r, s)


def test_vector_ops_vs_glm(oracle, golden):
    """normalize / getPointOnRay arithmetic / length via the oracle's public functions."""
    L = oracle.lib()
    for e in golden["glm_vectors"]["vec"]:
        a = [O.from_bits(u) for u in e["a"]]
        b = [O.from_bits(u) for u in e["b"]]
        s = O.from_bits(e["s"])
        # getPointOnRay(r={a,b}, t=s) = a + float(s - .0001f) * normalize(b)
        got = L.o_getPointOnRay(O.Ray(O.v3(a), O.v3(b)), s)
        assert [O.bits(x) for x in got.tup()] == e["a_plus_s_times_norm_b"]
        # reflection uses dot: k = 2*dot(incident, normal)  -> check dot through it: incident - k*normal
        refl = L.o_calculateReflectionDirection(O.v3(b), O.v3(a))
        dot = np.float32(O.from_bits(e["dot_ab"]))
        k = np.float32(2.0) * dot
        want = [np.float32(np.float32(ai) - np.float32(k * np.float32(bi))) for ai, bi in zip(a, b)]
        assert [O.bits(x) for x in refl.tup()] == [O.bits(x) for x in want]
    # identity matrix multiplyMV and length via getRadiuses on a pure scale
    g = O.make_geom(O.CUBE, 0, (0, 0, 0), (0, 0, 0), (2, 4, 6))
    assert L.o_getRadiuses(C.byref(g)).tup() == (1.0, 2.0, 3.0)


# ------------------------------------------------------------------ intersections.h
def test_loader_matches_reference_dump(oracle, golden, scenes_dir):
    ref = golden["reference_vectors"]["loader_sampleScene_radians"]
    s = O.LoadedScene(scenes_dir + "/sampleScene.txt", O.ROTAT_RADIANS)
    assert (s.n_objects, s.n_materials) == (ref["n_objects"], ref["n_materials"])
    cam = ref["camera"]
    assert (s.camera.resolution.x, s.camera.resolution.y) == tuple(cam["res"])
    assert (s.camera.fov.x, s.camera.fov.y) == tuple(cam["fov"])
    assert s.iterations == cam["iterations"] and s.image_name == cam["file"] and s.n_frames_camera == cam["frames"]
    assert s.camera.position.tup() == tuple(cam["eye"]) and s.camera.view.tup() == tuple(cam["view"])
    assert s.camera.up.tup() == tuple(cam["up"])
    names = {"sphere": O.SPHERE, "cube": O.CUBE}
    for k, o in enumerate(ref["objects"]):
        g = s.geoms[k]
        assert g.type == names[o["type"]] and g.materialid == o["material"]
        rows = g.transform.rows()
        np.testing.assert_allclose(rows[:3], o["rows"], atol=1.5e-6)     # dump printed with 6 decimals
        assert rows[3] == [0, 0, 0, 1]
    for k, m in enumerate(ref["materials"]):
        got = s.mats[k]
        np.testing.assert_allclose(got.color.tup(), m["color"], rtol=1e-7)
        assert f32(got.indexOfRefraction) == f32(m["ior"]) and got.hasReflective == m["refl"]
        assert got.hasRefractive == m["refr"] and got.emittance == m["emittance"]
    assert s.mats[5].absorptionCoefficient.tup() == tuple(f32(x) for x in ref["materials"][5]["abs"])
    assert s.mats[5].reducedScatterCoefficient == 13


def test_loader_grammar_tolerances(oracle, tmp_path, scenes_dir):
    """CRLF line ends, trailing //comments and tabs parse to the same scene (ref src/utilities.cpp:109-140)."""
    text = open(scenes_dir + "/sampleScene.txt").read()
    crlf = tmp_path / "crlf.txt"
    crlf.write_bytes(text.replace("MATERIAL 0", "MATERIAL 0\t\t//white diffuse").replace("\n", "\r\n").encode())
    a = O.LoadedScene(scenes_dir + "/sampleScene.txt")
    b = O.LoadedScene(str(crlf))
    assert bytes(a.geoms) == bytes(b.geoms) and bytes(a.mats) == bytes(b.mats) and bytes(a.camera) == bytes(b.camera)


def test_rotat_units(oracle, scenes_dir):
    rad = O.LoadedScene(scenes_dir + "/sampleScene.txt", O.ROTAT_RADIANS)
    deg = O.LoadedScene(scenes_dir + "/sampleScene.txt", O.ROTAT_DEGREES)
    # radians: "90" = 90 rad -> tilted floor; degrees: axis-aligned floor, normal = +-y
    r = rad.geoms[0].transform.rows()
    assert abs(r[0][0] - (-0.004480736)) < 1e-8
    d = np.array(deg.geoms[0].transform.rows())
    assert abs(d[0][0]) < 1e-8 and abs(abs(d[1][0]) - 0.01) < 1e-8      # x axis of the slab maps onto y


def test_sphere_golden(oracle, golden):
    L = oracle.lib()
    for e in golden["reference_vectors"]["sphere"]:
        g = trs_geom(e["trs"])
        d = e["d"] if "d" in e else normalize32(e["d_unnormalized"])
        p, n = O.Vec3(), O.Vec3()
        t = L.o_sphereIntersectionTest(C.byref(g), O.Ray(O.v3(e["o"]), O.v3(d)), C.byref(p), C.byref(n))
        if e["t"] == -1:
            assert t == -1.0
            continue
        assert_bits(t, e["t"], "sphere t")
        for got, want in zip(p.tup() + n.tup(), e["p"] + e["n"]):
            # components that are pure rounding noise (|x| < 1e-6) depend on the libm cosf/sinf of the
            # transform build; compare those absolutely
            if abs(want) < 1e-6:
                assert abs(got - want) < 1e-6
            else:
                assert_bits(got, want, "sphere p/n")


def test_hemisphere_golden(oracle, golden):
    L = oracle.lib()
    for e in golden["reference_vectors"]["hemisphere"]:
        n = e["n"] if "n" in e else normalize32(e["n_unnormalized"])
        libm = L.o_calculateRandomDirectionInHemisphere(O.v3(n), e["xi"][0], e["xi"][1], O.TRIG_LIBM)
        poly = L.o_calculateRandomDirectionInHemisphere(O.v3(n), e["xi"][0], e["xi"][1], O.TRIG_POLY)
        for got, gp, want in zip(libm.tup(), poly.tup(), e["out"]):
            if abs(want) < 1e-6:
                assert abs(got - want) < 1e-6       # cos(pi) * 0 noise terms
            else:
                assert_bits(got, want, "hemisphere (libm trig)")
            assert abs(gp - want) < 3e-7            # deterministic trig: within 2 ulp of the reference


def test_sincos_poly_accuracy(oracle):
    L = oracle.lib()
    s, c = C.c_float(), C.c_float()
    worst = 0.0
    for a in np.linspace(0, 2 * math.pi, 20001, dtype=np.float32):
        L.o_sincos_poly(float(a), C.byref(s), C.byref(c))
        worst = max(worst, abs(s.value - math.sin(float(a))), abs(c.value - math.cos(float(a))))
    assert worst < 2.4e-7
    L.o_sincos_poly(0.0, C.byref(s), C.byref(c))
    assert (s.value, c.value) == (0.0, 1.0)


def test_cube_sampling_golden(oracle, golden):
    L = oracle.lib()
    e = golden["reference_vectors"]["cube_sampling"]
    g = trs_geom(e["trs"], O.CUBE)
    for got, want in zip(L.o_getRadiuses(C.byref(g)).tup(), e["radiuses"]):
        assert_bits(got, want, "radiuses")
    for pt in e["points"]:
        got = L.o_getRandomPointOnCube(C.byref(g), float(pt["seed"]))
        for a, b in zip(got.tup(), pt["p"]):
            assert_bits(a, b, "cube point")
    mv = L.o_multiplyMV(g.transform, O.Vec4(*e["multiplyMV_half"]["v"]))
    for a, b in zip(mv.tup(), e["multiplyMV_half"]["out"]):
        assert_bits(a, b, "multiplyMV")


def test_misc_golden(oracle, golden):
    L = oracle.lib()
    m = golden["reference_vectors"]["misc"]
    r = O.Ray(O.v3(m["getPointOnRay"]["o"]), O.v3(m["getPointOnRay"]["d"]))
    for a, b in zip(L.o_getPointOnRay(r, float(m["getPointOnRay"]["t"])).tup(), m["getPointOnRay"]["out"]):
        assert_bits(a, b, "getPointOnRay")
    assert L.o_getSignOfRay(r).tup() == tuple(m["getSignOfRay"]["out"])
    assert L.o_epsilonCheck(1.0, f32(np.float32(1) + np.float32(1e-10))) == 1
    assert L.o_epsilonCheck(1.0, 1.1) == 0


def test_struct_layouts(oracle, golden):
    lay = golden["reference_vectors"]["layouts"]
    assert C.sizeof(O.Ray) == lay["ray"] and C.sizeof(O.Mat4) == lay["cudaMat4"]
    for cls, key in ((O.StaticGeom, "staticGeom"), (O.Material, "material"), (O.CameraData, "cameraData")):
        assert C.sizeof(cls) == lay[key]["size"]
        for name, _ in cls._fields_:
            assert getattr(cls, name).offset == lay[key][name], (key, name)


def test_pbo_conversion(oracle):
    L = oracle.lib()
    img = np.array([[0.0, 0.5, 1.0], [2.0, 0.999, 1e-3], [0.25, 1.0 / 255, 0.9999999]], dtype=np.float32)
    out = np.zeros((3, 4), dtype=np.uint8)
    L.o_sendImageToPBO(out.ctypes.data, 3, img.ctypes.data)
    want = np.array([[0, 127, 255, 0], [255, 254, 0, 0], [63, 1, 254, 0]], dtype=np.uint8)
    assert (out == want).all()


def test_render_checksums():
    """The canonical render semantics are pinned across rounds: the oracle's images for a handful of small
    configurations hash to the committed values (tests/golden/render_checksums.json, oracle/make_render_golden.py)."""
    import hashlib
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("make_render_golden", os.path.join(root, "oracle", "make_render_golden.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    want = json.load(open(os.path.join(root, "tests", "golden", "render_checksums.json")))
    assert sorted(want) == sorted(c[0] for c in gen.CASES)
    for case in gen.CASES:
        got = gen.run(case)
        assert got["sha256"] == want[case[0]]["sha256"] and got["live_in"] == want[case[0]]["live_in"], case[0]


def test_render_checksums_large():
    """The benchmark-shape fixture (tests/golden/render_checksums_large.json) names every case of its generator and the
    cheapest case re-hashes to the committed value; PT_TEST_LARGE_GOLDEN=1 re-runs all of them (minutes of CPU time)."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("make_render_golden_large", os.path.join(root, "oracle", "make_render_golden_large.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    want = json.load(open(os.path.join(root, "tests", "golden", "render_checksums_large.json")))
    assert sorted(want) == sorted(c[0] for c in gen.CASES)
    everything = os.environ.get("PT_TEST_LARGE_GOLDEN", "0") == "1"
    for case in gen.CASES:
        if not everything and case[0] != "config5_cloud_480x270_depth32_rr":
            continue
        got = gen.run(case)
        assert got["sha256"] == want[case[0]]["sha256"] and got["live_in"] == want[case[0]]["live_in"], case[0]
        assert got["row_sha256"] == want[case[0]]["row_sha256"]
