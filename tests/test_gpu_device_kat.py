"""GPU known-answer tests: the kernels' OWN device functions (evaluated on the MI355X through pt_device_kat)
against the reference's golden vectors (tests/golden/reference_vectors.json), real thrust
vectors, and the oracle on random inputs.  This pins the HIP implementation directly to the reference's
implemented functions, not only through the oracle."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as O
from __graft_entry__ import load_package

pytestmark = pytest.mark.gpu

HASH, U01SEQ, NOISE, INTERSECT, HEMI, RADIUSES, ON_CUBE, ON_SPHERE, MULMV, ON_RAY, REFLECT, REFRACT, FRESNEL = range(1, 14)


@pytest.fixture(scope="module")
def gpu():
    pkg = load_package()
    with pkg.Renderer(0) as r:
        yield r


def bits(x):
    return np.array(x, dtype=np.uint32).view(np.float32)


def rows16(m):
    return [v for row in m.rows() for v in row]


def assert_bits(got, want, what):
    g, w = np.asarray(got, np.float32), np.asarray(want, np.float32)
    assert (g.view(np.uint32) == w.view(np.uint32)).all(), f"{what}: got {g.tolist()} want {w.tolist()}"


def normalize32(v):
    v = [np.float32(c) for c in v]
    sqr = np.float32(np.float32(np.float32(v[0] * v[0]) + np.float32(v[1] * v[1])) + np.float32(v[2] * v[2]))
    inv = np.float32(np.float32(1.0) / np.sqrt(sqr))
    return [float(np.float32(c * inv)) for c in v]


def test_hash_on_device(gpu, golden):
    for a, want in golden["reference_vectors"]["hash"]:
        out = gpu.device_kat(HASH, bits([a]), 1)
        assert int(out.view(np.uint32)[0]) == want


def test_minstd_u01_on_device_vs_thrust(gpu, golden):
    for e in golden["thrust_rng_vectors"]["engine"]:
        out = gpu.device_kat(U01SEQ, bits([e["seed"]]), len(e["u01"]))
        assert out.view(np.uint32).tolist() == e["u01"], e["seed"]
    g = golden["reference_vectors"]["minstd_u01"]
    out = gpu.device_kat(U01SEQ, bits([O.lib().o_hash(g["hash_arg"])]), 3)
    assert_bits(out, g["u01"], "u01 after hash(7)")


def test_noise_on_device(gpu, golden):
    for n in golden["reference_vectors"]["noise"]:
        out = gpu.device_kat(NOISE, [n["res"][0], n["res"][1], n["time"], n["x"], n["y"]], 3)
        assert_bits(out, n["rgb"], "generateRandomNumberFromThread")


def test_sphere_intersection_on_device_golden(gpu, golden):
    for e in golden["reference_vectors"]["sphere"]:
        g = O.make_geom(O.SPHERE, 0, e["trs"]["t"], e["trs"]["r"], e["trs"]["s"])
        d = e["d"] if "d" in e else normalize32(e["d_unnormalized"])
        inp = list(bits([O.SPHERE])) + rows16(g.transform) + rows16(g.inverseTransform) + list(e["o"]) + list(d)
        out = gpu.device_kat(INTERSECT, inp, 7)
        if e["t"] == -1:
            assert out[0] == -1.0
            continue
        assert_bits(out[0], e["t"], "sphere t")
        for got, want in zip(out[1:7], e["p"] + e["n"]):
            if abs(want) < 1e-6:
                assert abs(got - want) < 1e-6          # rounding-noise components of the libm-built transform
            else:
                assert_bits(got, want, "sphere p/n")


def test_hemisphere_on_device_golden(gpu, golden):
    for e in golden["reference_vectors"]["hemisphere"]:
        n = e["n"] if "n" in e else normalize32(e["n_unnormalized"])
        out = gpu.device_kat(HEMI, list(n) + list(e["xi"]), 3)
        np.testing.assert_allclose(out, e["out"], atol=3e-7)       # deterministic trig: within 2 ulp of the reference
        want = O.lib().o_calculateRandomDirectionInHemisphere(O.v3(n), e["xi"][0], e["xi"][1], O.TRIG_POLY)
        assert_bits(out, want.tup(), "hemisphere vs oracle (same trig)")


def test_cube_sampling_on_device_golden(gpu, golden):
    e = golden["reference_vectors"]["cube_sampling"]
    g = O.make_geom(O.CUBE, 0, e["trs"]["t"], e["trs"]["r"], e["trs"]["s"])
    assert_bits(gpu.device_kat(RADIUSES, rows16(g.transform), 3), e["radiuses"], "getRadiuses")
    for pt in e["points"]:
        out = gpu.device_kat(ON_CUBE, rows16(g.transform) + [float(pt["seed"])], 3)
        assert_bits(out, pt["p"], "getRandomPointOnCube")
    mv = e["multiplyMV_half"]
    assert_bits(gpu.device_kat(MULMV, rows16(g.transform) + list(mv["v"]), 3), mv["out"], "multiplyMV")
    m = golden["reference_vectors"]["misc"]["getPointOnRay"]
    assert_bits(gpu.device_kat(ON_RAY, list(m["o"]) + list(m["d"]) + [m["t"]], 3), m["out"], "getPointOnRay")


def test_random_inputs_against_oracle(gpu):
    """Sphere / box tests, sphere-surface sampling, reflection, refraction and Fresnel on random inputs: the
    device functions equal the oracle bit for bit."""
    L = O.lib()
    rng = np.random.default_rng(7)
    for k in range(60):
        kind = O.SPHERE if k % 2 else O.CUBE
        s = rng.uniform(0.3, 4, 3) if k % 3 else np.full(3, rng.uniform(0.3, 4))
        g = O.make_geom(kind, 0, rng.uniform(-3, 3, 3), rng.uniform(-3, 3, 3), s)
        o = rng.uniform(-6, 6, 3).astype(np.float32)
        tgt = np.array(g.translation.tup()) + rng.uniform(-1, 1, 3) * s * 0.6
        d = np.array(normalize32(tgt - o), np.float32)
        p, n = O.Vec3(), O.Vec3()
        fn = L.o_sphereIntersectionTest if kind == O.SPHERE else L.o_boxIntersectionTest
        t = fn(C.byref(g), O.Ray(O.v3(o), O.v3(d)), C.byref(p), C.byref(n))
        out = gpu.device_kat(INTERSECT, list(bits([kind])) + rows16(g.transform) + rows16(g.inverseTransform) + list(o) + list(d), 7)
        assert_bits(out[0], t, "t")
        if t > 0:
            assert_bits(out[1:7], p.tup() + n.tup(), "p, n")
        seed = float(k)
        assert_bits(gpu.device_kat(ON_SPHERE, rows16(g.transform) + [seed], 3), L.o_getRandomPointOnSphere(C.byref(g), seed).tup(), "getRandomPointOnSphere")
        assert_bits(gpu.device_kat(ON_CUBE, rows16(g.transform) + [seed], 3), L.o_getRandomPointOnCube(C.byref(g), seed).tup(), "getRandomPointOnCube")
        nn = normalize32(rng.normal(size=3))
        ii = normalize32(rng.normal(size=3))
        if np.dot(nn, ii) > 0:
            nn = [-c for c in nn]
        assert_bits(gpu.device_kat(REFLECT, nn + ii, 3), L.o_calculateReflectionDirection(O.v3(nn), O.v3(ii)).tup(), "reflect")
        n1, n2 = (1.0, float(np.float32(rng.uniform(1.1, 2.5)))) if k % 2 else (float(np.float32(rng.uniform(1.1, 2.5))), 1.0)
        tr = L.o_calculateTransmissionDirection(O.v3(nn), O.v3(ii), n1, n2)
        assert_bits(gpu.device_kat(REFRACT, nn + ii + [n1, n2], 3), tr.tup(), "refract")
        fr = L.o_calculateFresnel(O.v3(nn), O.v3(ii), n1, n2, O.v3(0, 0, 0), tr)
        assert_bits(gpu.device_kat(FRESNEL, nn + ii + [n1, n2] + list(tr.tup()), 1)[0], fr.reflectionCoefficient, "fresnel")


def test_bad_kat_arguments(gpu):
    pkg = load_package()
    with pytest.raises(pkg.PtError):
        gpu.device_kat(99, [0.0], 1)
    with pytest.raises(pkg.PtError):
        gpu.device_kat(INTERSECT, [0.0] * 5, 7)


TRANSMISSION, SAMPLE_LIGHT = 14, 15


def test_transmission_and_light_sampler_against_oracle(gpu):
    """calculateTransmission (deterministic exp) and the direct-lighting sampler on the device == the oracle."""
    L = O.lib()
    rng = np.random.default_rng(11)
    for k in range(200):
        sig = [float(np.float32(v)) for v in rng.uniform(0, 6, 3)]
        dist = float(np.float32(10 ** rng.uniform(-4, 1.5)))
        want = L.o_calculateTransmission(O.v3(sig), dist)
        assert_bits(gpu.device_kat(TRANSMISSION, sig + [dist], 3), want.tup(), "calculateTransmission")
    assert_bits(gpu.device_kat(TRANSMISSION, [0.0, 0.0, 1000.0, 1.0], 3), [1.0, 1.0, 0.0], "exp(0), underflow")
    for k in range(60):
        kind = O.CUBE if k % 2 else O.SPHERE
        s = rng.uniform(0.3, 4, 3) if kind == O.CUBE else np.full(3, rng.uniform(0.3, 4))
        g = O.make_geom(kind, 0, rng.uniform(-3, 3, 3), rng.uniform(-3, 3, 3), s)
        p, n = O.Vec3(), O.Vec3()
        L.o_sampleLight(C.byref(g), float(k * 37), C.byref(p), C.byref(n))
        out = gpu.device_kat(SAMPLE_LIGHT, list(bits([kind])) + rows16(g.transform) + [float(k * 37)], 6)
        assert_bits(out, p.tup() + n.tup(), "sampleLight")


LOG, SCATTER = 16, 17


def test_log_and_scatter_step_against_oracle(gpu):
    """The free-flight sampler's deterministic log and calculateScatterAndAbsorption on the device == the oracle."""
    L = O.lib()
    rng = np.random.default_rng(17)
    xs = np.concatenate([rng.random(400), 2.0 ** -rng.uniform(0, 24, 100), [1.0, 0.5, 0.70710677, 0.70710683, 5.9604645e-08]]).astype(np.float32)
    for x in xs:
        assert_bits(gpu.device_kat(LOG, [float(x)], 1)[0], L.o_log_poly(float(x)), "log_poly")
    assert gpu.device_kat(LOG, [0.0], 1)[0] == -np.inf
    for k in range(300):
        o = [float(np.float32(v)) for v in rng.uniform(-3, 3, 3)]
        d = normalize32(rng.normal(size=3))
        depth = float(np.float32(10 ** rng.uniform(-3, 1)))
        sa = [float(np.float32(v)) for v in rng.uniform(0, 4, 3)]
        rsct = float(np.float32(rng.choice([0.0, 0.5, 3.0, 13.0])))
        T = [float(np.float32(v)) for v in rng.uniform(0, 1, 3)]
        u = [float(np.float32(v)) for v in rng.random(3)]
        if k % 50 == 0:
            u[0] = 1.0
        r = O.Ray(O.v3(o), O.v3(d))
        dep = C.c_float(depth)
        props = O.ScatterProps(O.v3(sa), rsct)
        col = O.v3(T)
        m = O.make_material()
        sc = L.o_calculateScatterAndAbsorption(C.byref(r), C.byref(dep), C.byref(props), C.byref(col), C.byref(m), u[0], u[1], u[2])
        out = gpu.device_kat(SCATTER, o + d + [depth] + sa + [rsct] + T + u, 11)
        assert int(out[0]) == sc
        assert_bits(out[1:11], r.origin.tup() + r.direction.tup() + (dep.value,) + col.tup(), "calculateScatterAndAbsorption")


SAMPLE_TRIANGLE = 18


def test_triangle_sampler_against_oracle(gpu):
    """Mesh lights: the device's uniform point on a triangle and its area == o_sampleTriangle / o_triangleArea, bit for bit."""
    L = O.lib()
    rng = np.random.default_rng(18)
    for k in range(300):
        v0, e1, e2 = ([float(np.float32(v)) for v in rng.uniform(-5, 5, 3)] for _ in range(3))
        ua, ub = float(np.float32(rng.random())), float(np.float32(rng.random()))
        if k % 60 == 0:
            ua, ub = (0.0, 1.0) if k % 120 == 0 else (1.0, 0.0)
        out = gpu.device_kat(SAMPLE_TRIANGLE, v0 + e1 + e2 + [ua, ub], 4)
        p = L.o_sampleTriangle(O.v3(v0), O.v3(e1), O.v3(e2), ua, ub)
        assert_bits(out[:3], p.tup(), "sampleTriangle")
        assert_bits(out[3], L.o_triangleArea(O.v3(e1), O.v3(e2)), "triangleArea")
