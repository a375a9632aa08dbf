"""Closed-form checks of the parts of the oracle that the reference ships as TODO stubs (no reference
behaviour to pin to): box slab test, mirror law, Snell/TIR, Fresnel, camera rays, cosine-weighted sampling,
Russian-roulette unbiasedness, energy conservation (furnace).  No GPU needed."""
import ctypes as C
import math

import numpy as np
import pytest

import os

import oracle_lib as O

SCENES = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scenes")


def ray(o, d):
    return O.Ray(O.v3(o), O.v3(d))


def unit(v):
    v = np.asarray(v, dtype=np.float64)
    return (v / np.linalg.norm(v)).tolist()


# ---------------------------------------------------------------- boxIntersectionTest (stub src/intersections.h:72-77)
def box_hit(g, o, d):
    p, n = O.Vec3(), O.Vec3()
    t = O.lib().o_boxIntersectionTest(C.byref(g), ray(o, d), C.byref(p), C.byref(n))
    return t, p.tup(), n.tup()


def test_box_axis_aligned_front_face(oracle):
    g = O.make_geom(O.CUBE, 0, (0, 0, 0), (0, 0, 0), (2, 2, 2))           # [-1,1]^3
    t, p, n = box_hit(g, (0, 0, 5), (0, 0, -1))
    assert abs(t - 4.0) < 3e-4 and abs(p[2] - 1.0) < 3e-4                  # 1e-4 object-space back-off x scale 2
    assert n == (0.0, 0.0, 1.0)
    assert abs(p[0]) < 1e-6 and abs(p[1]) < 1e-6


def test_box_miss_and_behind(oracle):
    g = O.make_geom(O.CUBE, 0, (0, 0, 0), (0, 0, 0), (2, 2, 2))
    assert box_hit(g, (3, 0, 5), (0, 0, -1))[0] == -1.0                    # passes beside
    assert box_hit(g, (0, 0, 5), (0, 0, 1))[0] == -1.0                     # box is behind the ray
    assert box_hit(g, (0, 0, 5), unit((0, 1, -1)))[0] == -1.0              # over the top


def test_box_from_inside_exit_face(oracle):
    g = O.make_geom(O.CUBE, 0, (0, 0, 0), (0, 0, 0), (2, 4, 6))
    t, p, n = box_hit(g, (0, 0, 0), (1, 0, 0))
    assert abs(t - 1.0) < 3e-4 and n == (1.0, 0.0, 0.0)                    # exit face, normal along the ray
    t, p, n = box_hit(g, (0, 0, 0), (0, -1, 0))
    assert abs(t - 2.0) < 5e-4 and n == (0.0, -1.0, 0.0)
    t, p, n = box_hit(g, (0.5, 0.5, 0), (0, 0, 1))
    assert abs(t - 3.0) < 7e-4 and n == (0.0, 0.0, 1.0)


def test_box_rotated_nonuniform_scale(oracle):
    # rotate 90 deg about z (radians), scale (2,6,2): the long axis ends up along world x
    g = O.make_geom(O.CUBE, 0, (1, 2, 3), (0, 0, math.pi / 2), (2, 6, 2))
    t, p, n = box_hit(g, (10, 2, 3), (-1, 0, 0))
    assert abs(p[0] - 4.0) < 1e-3 and abs(t - 6.0) < 1e-3
    assert abs(n[0] - 1.0) < 1e-6 and abs(n[1]) < 1e-6 and abs(n[2]) < 1e-6
    t, p, n = box_hit(g, (1, 10, 3), (0, -1, 0))
    assert abs(p[1] - 3.0) < 1e-3 and abs(n[1] - 1.0) < 1e-6


def test_box_grazing_and_axis_parallel(oracle):
    g = O.make_geom(O.CUBE, 0, (0, 0, 0), (0, 0, 0), (2, 2, 2))
    # ray parallel to two axes (object-space direction components exactly 0 -> slabs divide by zero)
    t, p, n = box_hit(g, (0.25, -0.75, 9), (0, 0, -1))
    assert abs(t - 8.0) < 3e-4 and n == (0.0, 0.0, 1.0)
    # parallel ray outside the slab on x misses
    assert box_hit(g, (1.5, 0, 9), (0, 0, -1))[0] == -1.0
    # random rays: every reported hit point lies on the box surface, normal is a unit axis
    rng = np.random.default_rng(1)
    hits = 0
    for _ in range(300):
        o = rng.uniform(-4, 4, 3)
        d = unit(rng.uniform(-1, 1, 3) - o)             # aim at the box so that most rays hit
        t, p, n = box_hit(g, o.tolist(), d)
        if t < 0:
            continue
        hits += 1
        assert abs(max(abs(c) for c in p) - 1.0) < 2e-3
        assert sorted(abs(c) for c in n)[:2] == [0.0, 0.0] and abs(max(abs(c) for c in n) - 1.0) < 1e-6
        assert abs(np.linalg.norm(np.asarray(p) - o) - t) < 1e-5
    assert hits > 30


def test_sphere_box_consistency_on_same_ray(oracle):
    """A ray through the common centre hits the inscribed sphere (r=.5*s) further in than the cube face."""
    s = O.make_geom(O.SPHERE, 0, (0, 0, 0), (0, 0, 0), (2, 2, 2))
    b = O.make_geom(O.CUBE, 0, (0, 0, 0), (0, 0, 0), (2, 2, 2))
    p, n = O.Vec3(), O.Vec3()
    ts = O.lib().o_sphereIntersectionTest(C.byref(s), ray((0, 0, 5), (0, 0, -1)), C.byref(p), C.byref(n))
    tb = box_hit(b, (0, 0, 5), (0, 0, -1))[0]
    assert abs(ts - 4.0) < 3e-4 and abs(tb - 4.0) < 3e-4
    ts = O.lib().o_sphereIntersectionTest(C.byref(s), ray((5, 5, 5), unit((-1, -1, -1))), C.byref(p), C.byref(n))
    tb = box_hit(b, (5, 5, 5), unit((-1, -1, -1)))[0]
    assert ts > tb > 0


# ---------------------------------------------------------------- reflection / refraction / Fresnel stubs
def test_reflection_law(oracle):
    L = oracle.lib()
    r = L.o_calculateReflectionDirection(O.v3(0, 1, 0), O.v3(unit((1, -1, 0)))).tup()
    np.testing.assert_allclose(r, unit((1, 1, 0)), atol=1e-7)
    rng = np.random.default_rng(2)
    for _ in range(50):
        n, d = unit(rng.normal(size=3)), unit(rng.normal(size=3))
        r = np.asarray(L.o_calculateReflectionDirection(O.v3(n), O.v3(d)).tup())
        assert abs(np.linalg.norm(r) - 1) < 1e-6
        assert abs(np.dot(r, n) + np.dot(d, n)) < 1e-6                 # angle in = angle out
        np.testing.assert_allclose(np.cross(np.cross(d, n), np.cross(r, n)), 0, atol=1e-6)   # same plane


def test_snell_and_total_internal_reflection(oracle):
    L = oracle.lib()
    n = (0, 1, 0)
    for deg in (0, 10, 30, 60, 85):
        th = math.radians(deg)
        d = (math.sin(th), -math.cos(th), 0)
        t = L.o_calculateTransmissionDirection(O.v3(n), O.v3(d), 1.0, 1.5).tup()
        sin_t = math.hypot(t[0], t[2])
        assert abs(sin_t - math.sin(th) / 1.5) < 1e-6                  # n1 sin(i) = n2 sin(t)
        assert t[1] < 0 and abs(np.linalg.norm(t) - 1) < 1e-6
    # glass -> air beyond the critical angle asin(1/1.5) = 41.8 deg
    th = math.radians(45)
    t = L.o_calculateTransmissionDirection(O.v3(n), O.v3((math.sin(th), -math.cos(th), 0)), 1.5, 1.0).tup()
    assert t == (0.0, 0.0, 0.0)
    th = math.radians(40)
    t = L.o_calculateTransmissionDirection(O.v3(n), O.v3((math.sin(th), -math.cos(th), 0)), 1.5, 1.0).tup()
    assert abs(math.hypot(t[0], t[2]) - 1.5 * math.sin(th)) < 1e-6


def test_fresnel(oracle):
    L = oracle.lib()
    n, d = (0, 1, 0), (0, -1, 0)
    for n1, n2 in ((1.0, 1.5), (1.0, 2.2), (1.5, 1.0), (1.33, 1.0)):
        t = L.o_calculateTransmissionDirection(O.v3(n), O.v3(d), n1, n2)
        f = L.o_calculateFresnel(O.v3(n), O.v3(d), n1, n2, O.v3(0, 1, 0), t)
        want = ((n1 - n2) / (n1 + n2)) ** 2                             # normal incidence
        assert abs(f.reflectionCoefficient - want) < 1e-6
        assert abs(f.reflectionCoefficient + f.transmissionCoefficient - 1) < 1e-7
    # grazing incidence -> reflectance -> 1 ; TIR -> exactly 1
    th = math.radians(89.9)
    d = (math.sin(th), -math.cos(th), 0)
    t = L.o_calculateTransmissionDirection(O.v3(n), O.v3(d), 1.0, 1.5)
    f = L.o_calculateFresnel(O.v3(n), O.v3(d), 1.0, 1.5, O.v3(0, 1, 0), t)
    assert f.reflectionCoefficient > 0.98
    f = L.o_calculateFresnel(O.v3(n), O.v3(d), 1.5, 1.0, O.v3(0, 1, 0), O.v3(0, 0, 0))
    assert (f.reflectionCoefficient, f.transmissionCoefficient) == (1.0, 0.0)
    # Brewster's angle: p-polarised reflectance vanishes -> R = rs^2 / 2
    thb = math.atan(1.5)
    d = (math.sin(thb), -math.cos(thb), 0)
    t = L.o_calculateTransmissionDirection(O.v3(n), O.v3(d), 1.0, 1.5)
    f = L.o_calculateFresnel(O.v3(n), O.v3(d), 1.0, 1.5, O.v3(0, 1, 0), t)
    ci, ct = math.cos(thb), math.sqrt(1 - (math.sin(thb) / 1.5) ** 2)
    rs = (ci - 1.5 * ct) / (ci + 1.5 * ct)
    assert abs(f.reflectionCoefficient - 0.5 * rs * rs) < 1e-6


def test_bsdf_lobe_selection(oracle):
    L = oracle.lib()
    n, p = O.v3(0, 1, 0), O.v3(0, 0, 0)
    d = unit((1, -1, 0))
    diffuse = O.make_material(color=(.2, .4, .6))
    mirror = O.make_material(color=(.2, .4, .6), spec=(.9, .8, .7), refl=1.0)
    glass = O.make_material(color=(0, 0, 0), spec=(1, 1, 1), refr=1.0, ior=1.5)
    r = ray((0, 1, 0), d)
    col = O.v3(1, 1, 1)
    assert L.o_calculateBSDF(C.byref(r), p, n, C.byref(col), C.byref(diffuse), 0.5, 0.3, 0.7, O.TRIG_POLY) == 0
    np.testing.assert_allclose(col.tup(), (.2, .4, .6), rtol=1e-7)
    assert r.direction.y > 0 and abs(r.origin.y - 0.0002) < 1e-9            # leaves on the normal's side, biased
    r, col = ray((0, 1, 0), d), O.v3(1, 1, 1)
    assert L.o_calculateBSDF(C.byref(r), p, n, C.byref(col), C.byref(mirror), 0.5, 0.3, 0.7, O.TRIG_POLY) == 1
    np.testing.assert_allclose(r.direction.tup(), unit((1, 1, 0)), atol=1e-7)
    np.testing.assert_allclose(col.tup(), (.9, .8, .7), rtol=1e-7)
    r, col = ray((0, 1, 0), d), O.v3(1, 1, 1)
    assert L.o_calculateBSDF(C.byref(r), p, n, C.byref(col), C.byref(glass), 0.9999, 0.3, 0.7, O.TRIG_POLY) == 2
    assert r.direction.y < 0 and r.origin.y < 0                              # transmitted, biased to the far side
    r, col = ray((0, 1, 0), d), O.v3(1, 1, 1)
    assert L.o_calculateBSDF(C.byref(r), p, n, C.byref(col), C.byref(glass), 0.0, 0.3, 0.7, O.TRIG_POLY) == 1
    # hit from the back side: the shading normal flips, a diffuse bounce leaves on the ray's side
    r, col = ray((0, -1, 0), (0, 1, 0)), O.v3(1, 1, 1)
    L.o_calculateBSDF(C.byref(r), p, n, C.byref(col), C.byref(diffuse), 0.5, 0.3, 0.7, O.TRIG_POLY)
    assert r.direction.y < 0 and r.origin.y < 0


# ---------------------------------------------------------------- sampling
def test_cosine_weighted_moments(oracle):
    L = oracle.lib()
    rng = np.random.default_rng(3)
    n = unit((0.3, 0.8, -0.5))
    cos = []
    for _ in range(20000):
        v = L.o_calculateRandomDirectionInHemisphere(O.v3(n), float(rng.random()), float(rng.random()), O.TRIG_POLY).tup()
        assert abs(np.linalg.norm(v) - 1) < 1e-5
        cos.append(float(np.dot(v, n)))
    cos = np.asarray(cos)
    assert cos.min() >= -1e-6
    assert abs(cos.mean() - 2.0 / 3.0) < 0.01          # E[cos] = 2/3 for p ~ cos
    assert abs((cos ** 2).mean() - 0.5) < 0.01


def test_uniform_sphere_direction(oracle):
    L = oracle.lib()
    rng = np.random.default_rng(4)
    v = np.array([L.o_getRandomDirectionInSphere(float(rng.random()), float(rng.random()), O.TRIG_POLY).tup()
                  for _ in range(20000)])
    np.testing.assert_allclose(np.linalg.norm(v, axis=1), 1, atol=1e-5)
    assert np.abs(v.mean(axis=0)).max() < 0.02
    assert abs((v[:, 2] ** 2).mean() - 1.0 / 3.0) < 0.01


# ---------------------------------------------------------------- camera (stub src/raytraceKernel.cu:38-45)
def test_camera_rays_span_the_fov(oracle):
    L = oracle.lib()
    W, H = 401, 301
    cam = O.make_camera(W, H, (0, 4.5, 12), (0, 0, -1), (0, 1, 0), 25)

    def mean_dir(x, y):
        acc = np.zeros(3)
        for it in range(1, 201):                       # average out the sub-pixel jitter
            r = L.o_raycastFromCameraKernel(cam.resolution, float(it), x, y, cam.position, cam.view, cam.up, cam.fov, 0)
            assert r.origin.tup() == (0.0, 4.5, 12.0)
            acc += r.direction.tup()
        return acc / np.linalg.norm(acc)

    c = mean_dir(W // 2, H // 2)
    np.testing.assert_allclose(c, (0, 0, -1), atol=2e-3)
    top = mean_dir(W // 2, 0)                          # buffer y=0 is the top row
    assert top[1] > 0 and abs(math.degrees(math.atan2(top[1], -top[2])) - 25) < 0.2
    right_edge = mean_dir(0, H // 2)                   # buffer x=0 is screen-right (ref src/main.cpp:123 flips it back)
    fovx = cam.fov.x
    assert right_edge[0] > 0 and abs(math.degrees(math.atan2(right_edge[0], -right_edge[2])) - fovx) < 0.2


def test_camera_jitter_stays_inside_the_pixel(oracle):
    L = oracle.lib()
    W, H = 64, 64
    cam = O.make_camera(W, H, (0, 0, 0), (0, 0, -1), (0, 1, 0), 30)
    ty = math.tan(math.radians(30))
    for it in range(1, 100):
        r = L.o_raycastFromCameraKernel(cam.resolution, float(it), 10, 20, cam.position, cam.view, cam.up, cam.fov, 0)
        d = r.direction.tup()
        sy = (1 - (d[1] / -d[2]) / ty) / 2 * H         # invert P = M + (1-2sx)H + (1-2sy)V
        sx = (1 - (d[0] / -d[2]) / math.tan(math.radians(cam.fov.x))) / 2 * W
        assert 20 - 1e-3 <= sy <= 21 + 1e-3 and 10 - 1e-3 <= sx <= 11 + 1e-3


# ---------------------------------------------------------------- whole-path statistics
def furnace_scene(emit, albedo):
    """Camera inside a closed cube whose walls all emit `emit` and reflect `albedo` diffusely."""
    geoms = (O.StaticGeom * 1)(O.make_geom(O.CUBE, 0, (0, 0, 0), (0, 0, 0), (10, 10, 10)))
    mats = (O.Material * 1)(O.make_material(color=(albedo,) * 3, emittance=emit))
    cam = O.make_camera(24, 24, (0, 0, 0), (0, 0, -1), (0, 1, 0), 30)
    return geoms, mats, cam


def test_emissive_enclosure_returns_emission(oracle):
    geoms, mats, cam = furnace_scene(2.0, 0.5)
    img, live = O.render(geoms, 1, mats, 1, cam, 4, iters=3)
    np.testing.assert_allclose(img, 2.0 * 0.5, rtol=1e-6)      # every path hits a light first: L = T*color*emittance
    assert [int(x) for x in live] == [3 * 24 * 24, 0, 0, 0]


def test_russian_roulette_is_unbiased(oracle):
    """Diffuse box lit by a ceiling light: means with and without roulette agree within noise."""
    sc = O.LoadedScene(__import__("os").path.join(__import__("os").path.dirname(__file__), "..", "scenes", "sampleScene.txt"), 1)
    sc.set_resolution(32, 32)
    a, la = O.render(sc.geoms, sc.n_objects, sc.mats, sc.n_materials, sc.camera, 12, iters=150, rr_start=-1)
    b, lb = O.render(sc.geoms, sc.n_objects, sc.mats, sc.n_materials, sc.camera, 12, iters=150, rr_start=2, seed=11)
    assert lb.sum() < la.sum()
    ma, mb = float(a.mean()), float(b.mean())
    assert ma > 0.05
    assert abs(ma - mb) / ma < 0.08


def test_depth_limits_path_length(oracle):
    sc = O.LoadedScene(__import__("os").path.join(__import__("os").path.dirname(__file__), "..", "scenes", "sampleScene.txt"))
    sc.set_resolution(40, 40)
    opt = O.Options(5, -1, 0, O.TRIG_POLY)
    nb = C.c_int()
    total = 0
    for y in range(0, 40, 5):
        for x in range(0, 40, 5):
            O.lib().o_trace_path(sc.geoms, sc.n_objects, sc.mats, sc.n_materials, C.byref(sc.camera), C.byref(opt), x, y, 1, C.byref(nb))
            assert 1 <= nb.value <= 5
            total += nb.value
    assert total > 64


def test_invalid_inputs_rejected(oracle):
    geoms, mats, cam = furnace_scene(1.0, 0.5)
    geoms[0].materialid = 3
    with pytest.raises(RuntimeError):
        O.render(geoms, 1, mats, 1, cam, 2)
    geoms[0].materialid = 0
    with pytest.raises(RuntimeError):
        O.render(geoms, 1, mats, 1, cam, 0)


# ---------------------------------------------------------------- direct lighting
def test_light_area_and_sampler():
    """o_lightArea is the area getRandomPointOnCube weights its faces with; o_sampleLight's point is the reference
    sampler's point and its normal is the outward normal of the face the point lies on."""
    L = O.lib()
    g = O.make_geom(O.CUBE, 0, (0, 10, 0), (0, 0, 0), (3, 0.3, 2))
    assert abs(L.o_lightArea(C.byref(g)) - 2 * (3 * 0.3 + 0.3 * 2 + 3 * 2)) < 1e-4
    s = O.make_geom(O.SPHERE, 0, (1, 2, 3), (0, 0, 0), (2, 2, 2))
    assert abs(L.o_lightArea(C.byref(s)) - 4 * np.pi) < 1e-4
    for seed in range(40):
        p, n = O.Vec3(), O.Vec3()
        L.o_sampleLight(C.byref(g), float(seed), C.byref(p), C.byref(n))
        assert p.tup() == L.o_getRandomPointOnCube(C.byref(g), float(seed)).tup()
        obj = (np.array(p.tup()) - np.array([0, 10, 0])) / np.array([3, 0.3, 2])
        k = int(np.argmax(np.abs(np.array(n.tup()))))
        assert abs(abs(obj[k]) - 0.5) < 1e-5 and np.sign(obj[k]) == np.sign(n.tup()[k]) and abs(np.linalg.norm(n.tup()) - 1) < 1e-6
        L.o_sampleLight(C.byref(s), float(seed), C.byref(p), C.byref(n))
        assert np.allclose(np.array(p.tup()) - np.array([1, 2, 3]), np.array(n.tup()), atol=1e-5)


def test_direct_lighting_estimates_the_same_image():
    """Explicit light sampling changes the estimator, not the integral: with d vertices + a light connection it
    converges to what pure path tracing gives with d + 1 bounces (statistically, 2 % on the frame mean)."""
    sc = O.LoadedScene(os.path.join(SCENES, "sampleScene.txt"), 1)
    sc.set_resolution(40, 40)
    a, _ = O.render(sc.geoms, sc.n_objects, sc.mats, sc.n_materials, sc.camera, 4, iters=1500)
    sh = []
    b, _ = O.render(sc.geoms, sc.n_objects, sc.mats, sc.n_materials, sc.camera, 3, iters=1500, direct_light=1, shadow_out=sh)
    assert sh[0] > 0
    ma, mb = a.mean(axis=(0, 1)), b.mean(axis=(0, 1))
    assert np.all(np.abs(ma - mb) <= 0.02 * ma), (ma, mb)
    # and it is the lower-variance one (the point of sampling lights): pixel noise against a long reference
    ref, _ = O.render(sc.geoms, sc.n_objects, sc.mats, sc.n_materials, sc.camera, 3, iters=6000, direct_light=1, seed=7)
    a16, _ = O.render(sc.geoms, sc.n_objects, sc.mats, sc.n_materials, sc.camera, 4, iters=16)
    b16, _ = O.render(sc.geoms, sc.n_objects, sc.mats, sc.n_materials, sc.camera, 3, iters=16, direct_light=1)
    assert np.mean((b16 - ref) ** 2) < 0.5 * np.mean((a16 - ref) ** 2)


def test_triangle_sampler_is_uniform():
    """o_sampleTriangle: every point lies in the triangle, the sample mean is the centroid and each of the four
    medial sub-triangles receives a quarter of the points; o_triangleArea is half the cross product's length."""
    L = O.lib()
    v0, e1, e2 = (1.0, 2.0, 3.0), (2.0, 0.0, 1.0), (0.0, 3.0, -1.0)
    assert abs(L.o_triangleArea(O.v3(e1), O.v3(e2)) - 0.5 * np.linalg.norm(np.cross(e1, e2))) < 1e-6
    rng = np.random.default_rng(3)
    n = 20000
    uv = rng.random((n, 2)).astype(np.float32)
    E = np.stack([e1, e2], axis=1)
    bary = np.empty((n, 2))
    for k in range(n):
        p = np.array(L.o_sampleTriangle(O.v3(v0), O.v3(e1), O.v3(e2), float(uv[k, 0]), float(uv[k, 1])).tup()) - np.array(v0)
        bary[k] = np.linalg.lstsq(E, p, rcond=None)[0]
    a, b = bary[:, 0], bary[:, 1]
    assert (a >= -1e-6).all() and (b >= -1e-6).all() and (a + b <= 1 + 1e-6).all()
    assert abs(a.mean() - 1 / 3) < 0.01 and abs(b.mean() - 1 / 3) < 0.01
    quarter = [(a > 0.5), (b > 0.5), (a + b < 0.5), (a <= 0.5) & (b <= 0.5) & (a + b >= 0.5)]
    for q in quarter:
        assert abs(q.mean() - 0.25) < 0.015


def _room_with_mesh_light(with_cube_light):
    """Diffuse floor and back wall, an emissive two-triangle quad above the floor, optionally a cube light beside it."""
    mats = [O.make_material(color=(0.8, 0.8, 0.8)), O.make_material(color=(1, 1, 1), emittance=6.0),
            O.make_material(color=(1, 0.9, 0.8), emittance=4.0)]
    geoms = [O.make_geom(O.CUBE, 0, (0, -0.05, 0), (0, 0, 0), (12, 0.1, 12)),
             O.make_geom(O.CUBE, 0, (0, 3, -4), (0, 0, 0), (12, 6, 0.1)),
             O.make_geom(O.MESH, 1, (-1.0, 3.0, 0.0), (0.3, 0.2, 0.1), (2.5, 1, 2.5))]
    if with_cube_light:
        geoms.append(O.make_geom(O.CUBE, 2, (2.5, 2.0, 0.5), (0, 0.4, 0), (0.8, 0.8, 0.8)))
    quad = np.array([[-0.5, 0, -0.5, 0.5, 0, -0.5, 0.5, 0, 0.5], [-0.5, 0, -0.5, 0.5, 0, 0.5, -0.5, 0, 0.5]], dtype=np.float32)
    ga = (O.StaticGeom * len(geoms))(*geoms)
    ma = (O.Material * len(mats))(*mats)
    cam = O.make_camera(32, 24, (0, 2.0, 9), (0, -0.1, -1), (0, 1, 0), 25)
    return ga, len(geoms), ma, len(mats), cam, {2: quad}


@pytest.mark.parametrize("with_cube_light", [False, True])
def test_mesh_lights_are_sampled_explicitly(with_cube_light):
    """An emissive MESH geom is a light of the table: with explicit sampling (d vertices + a connection) the frame mean
    equals pure path tracing with d + 1 bounces -- also beside a cube light (round 2 dropped the mesh's light at
    diffuse vertices in that case: it was suppressed as "sampled" without being in the table)."""
    ga, nG, ma, nM, cam, meshes = _room_with_mesh_light(with_cube_light)
    a, _ = O.render(ga, nG, ma, nM, cam, 3, iters=1200, meshes=meshes)
    sh = []
    b, _ = O.render(ga, nG, ma, nM, cam, 2, iters=1200, meshes=meshes, direct_light=1, shadow_out=sh)
    assert sh[0] > 0
    m_a, m_b = a.mean(axis=(0, 1)), b.mean(axis=(0, 1))
    assert m_a.min() > 0.05
    assert np.all(np.abs(m_a - m_b) <= 0.03 * m_a), (m_a, m_b)
    # lower variance with the explicit connection
    ref, _ = O.render(ga, nG, ma, nM, cam, 2, iters=4000, meshes=meshes, direct_light=1, seed=11)
    a8, _ = O.render(ga, nG, ma, nM, cam, 3, iters=8, meshes=meshes)
    b8, _ = O.render(ga, nG, ma, nM, cam, 2, iters=8, meshes=meshes, direct_light=1)
    assert np.mean((b8 - ref) ** 2) < 0.6 * np.mean((a8 - ref) ** 2)


def test_emitters_beyond_the_light_table_still_count():
    """The table holds 16 entries; a 17th emitter is reached by chance only, and such a hit is NOT suppressed after a
    diffuse vertex (it would otherwise contribute nothing at all): means with and without explicit sampling agree."""
    mats = [O.make_material(color=(0.8, 0.8, 0.8)), O.make_material(color=(1, 1, 1), emittance=3.0)]
    geoms = [O.make_geom(O.CUBE, 0, (0, -0.05, 0), (0, 0, 0), (14, 0.1, 14))]
    for k in range(17):
        geoms.append(O.make_geom(O.SPHERE, 1, (-4.0 + 0.5 * k, 1.5 + 0.1 * (k % 3), -1.0 + 0.3 * (k % 4)), (0, 0, 0), (0.4, 0.4, 0.4)))
    geoms[-1] = O.make_geom(O.SPHERE, 1, (0.0, 4.0, 1.0), (0, 0, 0), (2.5, 2.5, 2.5))      # the 17th: large, off the table
    ga = (O.StaticGeom * len(geoms))(*geoms)
    ma = (O.Material * len(mats))(*mats)
    cam = O.make_camera(24, 16, (0, 2.5, 10), (0, -0.15, -1), (0, 1, 0), 25)
    a, _ = O.render(ga, len(geoms), ma, 2, cam, 3, iters=1500)
    b, _ = O.render(ga, len(geoms), ma, 2, cam, 2, iters=1500, direct_light=1)
    m_a, m_b = a.mean(axis=(0, 1)), b.mean(axis=(0, 1))
    # pure path tracing with d + 1 bounces sees what d vertices + connection see, except the big off-table sphere's light
    # at the LAST vertex (no bounce left to reach it by chance): b is the smaller one, but far above "17th light ignored"
    g17 = (O.StaticGeom * 17)(*geoms[:17])
    c, _ = O.render(g17, 17, ma, 2, cam, 2, iters=1500, direct_light=1)
    m_c = c.mean(axis=(0, 1))
    assert np.all(m_b > 1.5 * m_c), (m_b, m_c)
    assert np.all(m_b <= 1.03 * m_a)


def test_direct_lighting_off_without_lights_or_flag():
    sc = O.LoadedScene(os.path.join(SCENES, "sampleScene.txt"), 1)
    sc.set_resolution(24, 24)
    a, la = O.render(sc.geoms, sc.n_objects, sc.mats, sc.n_materials, sc.camera, 4, iters=2)
    for k in range(sc.n_materials):
        sc.mats[k].emittance = 0.0
    sh = []
    b, lb = O.render(sc.geoms, sc.n_objects, sc.mats, sc.n_materials, sc.camera, 4, iters=2, direct_light=1, shadow_out=sh)
    assert sh[0] == 0 and not b.any()


def test_exp_poly_and_transmission():
    """The deterministic exp is within 1.5e-7 (relative) of libm on its whole range; transmission is Beer-Lambert."""
    L = O.lib()
    xs = np.concatenate([np.linspace(-87, 88, 20001), -np.logspace(-8, 1.9, 500), [0.0, -0.0]]).astype(np.float32)
    for x in xs:
        assert abs(L.o_exp_poly(float(x)) - math.exp(float(x))) <= 1.5e-7 * math.exp(float(x))
    assert L.o_exp_poly(-0.0) == 1.0 and L.o_exp_poly(-1000.0) == 0.0
    t = L.o_calculateTransmission(O.v3(0.02, 5.1, 5.7), 0.5)
    assert np.allclose(t.tup(), np.exp(-np.array([0.02, 5.1, 5.7]) * 0.5), rtol=1e-6)


def test_absorption_only_acts_inside_refractive_objects():
    sc = O.LoadedScene(os.path.join(SCENES, "sampleScene.txt"), 1)       # no refractive object in this scene
    sc.set_resolution(24, 24)
    a, _ = O.render(sc.geoms, sc.n_objects, sc.mats, sc.n_materials, sc.camera, 4, iters=2)
    b, _ = O.render(sc.geoms, sc.n_objects, sc.mats, sc.n_materials, sc.camera, 4, iters=2, absorption=1)
    assert np.array_equal(a, b)
    g = O.LoadedScene(os.path.join(SCENES, "cornell_glass.txt"), 1)
    g.set_resolution(32, 32)
    a, la = O.render(g.geoms, g.n_objects, g.mats, g.n_materials, g.camera, 6, iters=4)
    b, lb = O.render(g.geoms, g.n_objects, g.mats, g.n_materials, g.camera, 6, iters=4, absorption=1)
    assert np.array_equal(la, lb) and (b <= a + 1e-6).all() and b.sum() < a.sum()


@pytest.mark.parametrize("scale", [1.0, 4.0, 9.0])
def test_transmitted_rays_cross_scaled_objects(scale):
    """A ray through the centre of an index-matched (ior 1, so R = 0 and no bending) sphere or cube of any size reaches
    the light behind it in exactly three vertices -- enter, leave, light -- with the light's full radiance.  (With a
    fixed 0.0002 offset the transmitted ray of an object scaled by more than 2 would meet the entry surface again.)"""
    L = O.lib()
    mats = (O.Material * 2)(O.make_material(color=(0, 0, 0), refr=1.0, ior=1.0),
                            O.make_material(color=(1, 0.5, 0.25), emittance=3.0))
    cam = O.make_camera(1, 1, (0, 0, 20), (0, 0, -1), (0, 1, 0), 1.0)
    for kind in (O.SPHERE, O.CUBE):
        geoms = (O.StaticGeom * 2)(O.make_geom(kind, 0, (0, 0, 0), (0, 0, 0), (scale, scale, scale)),
                                   O.make_geom(O.CUBE, 1, (0, 0, -12), (0, 0, 0), (30, 30, 0.5)))
        for depth, want in ((3, (3.0, 1.5, 0.75)), (2, (0.0, 0.0, 0.0))):
            opt = O.Options(depth, -1, 0, O.TRIG_POLY)
            nb = C.c_int(0)
            v = L.o_trace_path(geoms, 2, mats, 2, C.byref(cam), C.byref(opt), 0, 0, 1, C.byref(nb))
            assert v.tup() == want and nb.value == depth, (kind, scale, depth, v.tup(), nb.value)


def test_thin_lens_focus():
    """Depth of field: a small light on the plane in focus images as sharply as through the pinhole; off that plane it
    spreads over more pixels at the same total energy (within sampling noise)."""
    mats = (O.Material * 1)(O.make_material(color=(1, 1, 1), emittance=1.0))
    cam = O.make_camera(64, 64, (0, 0, 10), (0, 0, -1), (0, 1, 0), 6.0)
    geoms = (O.StaticGeom * 1)(O.make_geom(O.SPHERE, 0, (0, 0, 0), (0, 0, 0), (0.5, 0.5, 0.5)))

    def lit(**kw):
        img, _ = O.render(geoms, 1, mats, 1, cam, 1, iters=64, **kw)
        return int((img[..., 0] > 0.02).sum()), float(img[..., 0].sum())

    n_pin, e_pin = lit()
    n_focus, e_focus = lit(lens_radius=0.4, focal_distance=10.0)
    n_blur, e_blur = lit(lens_radius=0.4, focal_distance=5.0)
    assert abs(n_focus - n_pin) <= 0.35 * n_pin and n_blur > 2.5 * n_pin
    assert abs(e_focus - e_pin) <= 0.1 * e_pin and abs(e_blur - e_pin) <= 0.15 * e_pin


# ------------------------------------------------------------------ subsurface scattering
def test_log_poly_accuracy():
    L = O.lib()
    xs = np.concatenate([np.linspace(1e-7, 1.0, 5001), 2.0 ** -np.arange(1, 25), [0.70710677, 0.70710683]]).astype(np.float32)
    for x in xs:
        got = L.o_log_poly(float(x))
        want = math.log(float(x))
        assert abs(got - want) <= 2e-7 * max(1.0, abs(want)), (float(x), got, want)
    assert L.o_log_poly(1.0) == 0.0 and L.o_log_poly(0.0) == -math.inf


def _scatter(o, d, depth, sa, rsct, T, u1, u2, u3):
    L = O.lib()
    r = O.Ray(O.v3(o), O.v3(d))
    dep = C.c_float(depth)
    props = O.ScatterProps(O.v3(sa), rsct)
    col = O.v3(T)
    m = O.make_material()
    sc = L.o_calculateScatterAndAbsorption(C.byref(r), C.byref(dep), C.byref(props), C.byref(col), C.byref(m), u1, u2, u3)
    return sc, r, dep.value, col.tup()


def test_scatter_free_flight_distribution():
    """Free flight -ln(1-u)/sigma: mean 1/sigma over the events, P(scatter before D) = 1 - exp(-sigma D); a scattered
    ray starts where the walk left it, in a unit direction, absorbed over the flight only."""
    rng = np.random.default_rng(3)
    sigma, D = 2.5, 0.9
    flights, n = [], 20000
    for u in rng.random(n).astype(np.float32):
        sc, r, dep, col = _scatter((1, 2, 3), (0, 0, 1), 1e9, (0.3, 0.0, 1.0), sigma, (1, 1, 1), float(u), 0.3, 0.6)
        assert sc == 1
        flights.append(dep)
        assert abs(r.origin.z - (3 + dep)) < 1e-5 * max(1, dep) and r.origin.x == 1 and r.origin.y == 2
        dn = math.sqrt(sum(c * c for c in r.direction.tup()))
        assert abs(dn - 1) < 1e-5
        assert abs(col[0] - math.exp(-0.3 * dep)) < 1e-5 and col[1] == 1.0 and abs(col[2] - math.exp(-dep)) < 1e-5
    assert abs(np.mean(flights) - 1 / sigma) < 0.01
    hits = sum(_scatter((0, 0, 0), (1, 0, 0), D, (0, 0, 0), sigma, (1, 1, 1), float(u), 0.1, 0.2)[0]
               for u in rng.random(n).astype(np.float32))
    assert abs(hits / n - (1 - math.exp(-sigma * D))) < 0.01


def test_scatter_boundary_and_degenerate_cases():
    # no scattering coefficient: never scatters, absorbs over the whole segment
    sc, r, dep, col = _scatter((0, 0, 0), (0, 1, 0), 2.0, (0.5, 0.0, 0.25), 0.0, (1, 0.5, 1), 0.999, 0.5, 0.5)
    assert sc == 0 and dep == 2.0 and r.origin.tup() == (0, 0, 0) and r.direction.tup() == (0, 1, 0)
    assert abs(col[0] - math.exp(-1.0)) < 1e-6 and col[1] == 0.5 and abs(col[2] - math.exp(-0.5)) < 1e-6
    # u == 1: ln(0): no scattering event
    assert _scatter((0, 0, 0), (0, 1, 0), 2.0, (0, 0, 0), 5.0, (1, 1, 1), 1.0, 0.5, 0.5)[0] == 0
    # flight exactly at / beyond the boundary is not a scattering event
    assert _scatter((0, 0, 0), (0, 1, 0), 0.0, (0, 0, 0), 5.0, (1, 1, 1), 0.5, 0.5, 0.5)[0] == 0


def test_scatter_render_conserves_and_differs(scenes_dir):
    """The option changes the picture of a scene with SCATTER materials, leaves one without them alone, and a
    non-absorbing medium in a furnace does not create energy."""
    sc = O.LoadedScene(scenes_dir + "/sss_blobs.txt", O.ROTAT_DEGREES)
    sc.set_resolution(64, 64)
    a, la = O.render(sc.geoms, sc.n_objects, sc.mats, sc.n_materials, sc.camera, 8, iters=2)
    b, lb = O.render(sc.geoms, sc.n_objects, sc.mats, sc.n_materials, sc.camera, 8, iters=2, scatter=1)
    assert not np.array_equal(a, b) and np.isfinite(b).all() and b.min() >= 0
    assert int(lb.sum()) > int(la.sum())            # the walk inside the media costs bounces
    s2 = O.LoadedScene(scenes_dir + "/sampleScene_spec.txt")
    s2.set_resolution(48, 48)
    c, _ = O.render(s2.geoms, s2.n_objects, s2.mats, s2.n_materials, s2.camera, 6, iters=1)
    d, _ = O.render(s2.geoms, s2.n_objects, s2.mats, s2.n_materials, s2.camera, 6, iters=1, scatter=1)
    assert np.array_equal(c, d)


# ------------------------------------------------------------------ triangles / MESH objects
def _tri(v0, v1, v2, o, d):
    L = O.lib()
    g = O.make_geom(O.MESH, 0, (0, 0, 0), (0, 0, 0), (1, 1, 1))
    w = (C.c_float * 12)()
    L.o_triangleToWorld(C.byref(g), (C.c_float * 9)(*(list(v0) + list(v1) + list(v2))), w)
    p, n = O.Vec3(), O.Vec3()
    t = L.o_triangleIntersectionTest(O.v3(w[0:3]), O.v3(w[3:6]), O.v3(w[6:9]), O.v3(w[9:12]), O.Ray(O.v3(o), O.v3(d)), C.byref(p), C.byref(n))
    return t, p.tup(), n.tup()


def test_triangle_intersection_closed_forms():
    """World-space Moeller-Trumbore: distance to the pulled-back point, geometric normal (not flipped), two-sided, edges
    inside, outside and parallel rays miss, the direction need not be unit length."""
    A, B, Cc = (0, 0, 0), (2, 0, 0), (0, 2, 0)                        # normal +z
    t, p, n = _tri(A, B, Cc, (0.5, 0.5, 3), (0, 0, -1))
    assert abs(t - (3 - 1e-4)) < 1e-6 and abs(p[2] - 1e-4) < 1e-6 and n == (0, 0, 1)
    t2, p2, n2 = _tri(A, B, Cc, (0.5, 0.5, -3), (0, 0, 5))           # from behind, unnormalised direction
    assert abs(t2 - (3 - 1e-4)) < 1e-6 and n2 == (0, 0, 1) and abs(p2[2] + 1e-4) < 1e-6
    assert _tri(A, B, Cc, (1.5, 1.5, 3), (0, 0, -1))[0] == -1        # beyond the hypotenuse
    assert _tri(A, B, Cc, (-0.1, 0.5, 3), (0, 0, -1))[0] == -1
    assert _tri(A, B, Cc, (0.5, 0.5, 3), (0, 0, 1))[0] == -1         # pointing away
    assert _tri(A, B, Cc, (0.5, 0.5, 3), (1, 0, 0))[0] == -1         # parallel to the plane
    assert _tri(A, B, Cc, (1.0, 0.0, 2), (0, 0, -1))[0] > 0          # on an edge: inside (u, v >= 0 inclusive)
    assert _tri(A, A, Cc, (0.5, 0.5, 3), (0, 0, -1))[0] == -1        # degenerate triangle never hit
    # oblique: hit point lies in the plane (up to the pull-back), distance = |o - p|
    t3, p3, _ = _tri((1, 1, 1), (4, 1, 2), (1, 5, 3), (2, 2, 9), (0.1, 0.05, -1))
    assert t3 > 0 and abs(math.dist((2, 2, 9), p3) - t3) < 1e-5


def test_obj_readers_agree_and_fan_polygons(scenes_dir, tmp_path):
    """The oracle's OBJ reader and the product's read the same triangles from the bundled meshes; polygons are cut into
    fans around their first vertex; negative indices count back from the vertices read so far."""
    from __graft_entry__ import load_package
    pkg = load_package()
    a = O.LoadedScene(scenes_dir + "/mesh_cornell.txt", O.ROTAT_DEGREES)
    b = pkg.SceneFile(scenes_dir + "/mesh_cornell.txt", pkg.ROTAT_DEGREES)
    assert sorted(a.meshes) == sorted(b.meshes) == [5, 6]
    assert a.meshes[5].shape == (320, 9) and a.meshes[6].shape == (1 * 6 + 8 * 2 + 8, 9)
    for k in a.meshes:
        assert np.array_equal(a.meshes[k], b.meshes[k])
    assert a.geoms[5].type == O.MESH and b.geoms[5].type == pkg.MESH
    obj = tmp_path / "quad.obj"
    obj.write_text("v 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nvn 0 0 1\nf 1/1/1 2/2/1 3/3/1 4/4/1\nf -4 -2 -1\nf 1 2 9\n")
    vp, nt = C.POINTER(C.c_float)(), C.c_int()
    assert O.lib().o_load_obj(str(obj).encode(), C.byref(vp), C.byref(nt)) == 0 and nt.value == 3
    tri = np.ctypeslib.as_array(vp, shape=(3, 9)).copy()
    O.lib().o_free_obj(vp)
    assert tri[0].tolist() == [0, 0, 0, 1, 0, 0, 1, 1, 0] and tri[1].tolist() == [0, 0, 0, 1, 1, 0, 0, 1, 0]
    assert tri[2].tolist() == [0, 0, 0, 1, 1, 0, 0, 1, 0]            # -4 -2 -1 -> vertices 1 3 4; the face with index 9 is dropped
    scene = tmp_path / "s.txt"
    scene.write_text(open(scenes_dir + "/mesh_cornell.txt").read().replace("meshes/icosphere.obj", "quad.obj").replace("meshes/gem.obj", "quad.obj"))
    s2 = pkg.SceneFile(str(scene), pkg.ROTAT_DEGREES)
    assert np.array_equal(s2.meshes[5], tri) and np.array_equal(s2.meshes[6], tri)


def test_mesh_render_is_seen(scenes_dir):
    sc = O.LoadedScene(scenes_dir + "/mesh_cornell.txt", O.ROTAT_DEGREES)
    sc.set_resolution(48, 48)
    a, la = O.render(sc.geoms, sc.n_objects, sc.mats, sc.n_materials, sc.camera, 5, iters=1)
    b, lb = O.render(sc.geoms, sc.n_objects, sc.mats, sc.n_materials, sc.camera, 5, iters=1, meshes=sc.meshes)
    assert not np.array_equal(a, b) and np.isfinite(b).all() and int(lb.sum()) > int(la.sum())


# ------------------------------------------------------------------ motion blur
def test_motion_slices_schedule_and_interpolation(scenes_dir):
    """Slice k of n sits at shutter time (k + .5)/n between the two frames; iterations are dealt to the slices in runs of
    16; one slice with identical frames is the static render; a moving sphere smears."""
    L = O.lib()
    a = O.LoadedScene(scenes_dir + "/sampleScene_anim.txt", O.ROTAT_DEGREES, frame=0)
    b = O.LoadedScene(scenes_dir + "/sampleScene_anim.txt", O.ROTAT_DEGREES, frame=1)
    for s in (a, b):
        s.set_resolution(48, 36)
    assert [L.o_sliceTime(k, 4) for k in range(4)] == [0.125, 0.375, 0.625, 0.875]
    g = L.o_interpolateGeom(C.byref(a.geoms[5]), C.byref(b.geoms[5]), 0.5, O.ROTAT_DEGREES)
    for c in "xyz":
        want = np.float32(getattr(a.geoms[5].translation, c)) + (np.float32(getattr(b.geoms[5].translation, c)) - np.float32(getattr(a.geoms[5].translation, c))) * np.float32(0.5)
        assert getattr(g.translation, c) == float(want)
    same = O.motion_slices(a.geoms, a.geoms, a.n_objects, a.camera, a.camera, 3, O.ROTAT_DEGREES)
    st, _ = O.render(a.geoms, a.n_objects, a.mats, a.n_materials, a.camera, 4, iters=20)
    mv, _ = O.render(a.geoms, a.n_objects, a.mats, a.n_materials, a.camera, 4, iters=20, slice_geoms=same[0], slice_cams=same[1])
    assert np.array_equal(st, mv)                      # nothing moves: the slices are the static scene
    sg, sc = O.motion_slices(a.geoms, b.geoms, a.n_objects, a.camera, b.camera, 2, O.ROTAT_DEGREES)
    blur, _ = O.render(a.geoms, a.n_objects, a.mats, a.n_materials, a.camera, 4, iters=32, slice_geoms=sg, slice_cams=sc)
    # iterations 1..16 see slice 0, 17..32 slice 1: the mean of the two half-renders
    h0, _ = O.render(sg[0], a.n_objects, a.mats, a.n_materials, sc[0], 4, iters=16)
    assert not np.array_equal(blur, st)
    first16, _ = O.render(a.geoms, a.n_objects, a.mats, a.n_materials, a.camera, 4, iters=16, slice_geoms=sg, slice_cams=sc)
    sc0 = O.CameraData.from_buffer_copy(sc[0]); sc0.resolution = a.camera.resolution; sc0.fov = a.camera.fov
    h0, _ = O.render(sg[0], a.n_objects, a.mats, a.n_materials, sc0, 4, iters=16)
    assert np.array_equal(first16, h0)


# ---------------------------------------------------------------- motion blur with a shutter time per ray
def _moving_light_scene():
    mats = [O.make_material(color=(1, 1, 1), emittance=1.0)]
    a = [O.make_geom(O.SPHERE, 0, (-1.5, 0, 0), (0, 0, 0), (1, 1, 1))]
    b = [O.make_geom(O.SPHERE, 0, (1.5, 0, 0), (0, 0, 0), (1, 1, 1))]
    ga, gb = (O.StaticGeom * 1)(*a), (O.StaticGeom * 1)(*b)
    ma = (O.Material * 1)(*mats)
    cam = O.make_camera(48, 16, (0, 0, 8), (0, 0, -1), (0, 1, 0), 8)
    return ga, gb, ma, cam


def test_per_ray_shutter_time_is_the_time_average():
    """An emissive sphere translating across the frame, seen directly (depth 1): with a time per ray the frame mean is
    the time average -- the same as many shutter slices give -- and a row through the sweep is flat where the sphere
    passes completely (no ghost copies, which K slices show as K separate discs)."""
    ga, gb, ma, cam = _moving_light_scene()
    kg, _ = O.motion_knots(ga, gb, 1, cam, None, 1)
    a, _ = O.render(ga, 1, ma, 1, cam, 1, iters=1600, knot_geoms=kg)
    sg, sc = O.motion_slices(ga, gb, 1, cam, None, 50)
    b, _ = O.render(ga, 1, ma, 1, cam, 1, iters=1600, slice_geoms=sg)
    assert abs(a.mean() - b.mean()) <= 0.02 * b.mean()
    row = a[8, :, 0]
    mid = row[16:32]
    assert mid.std() <= 0.12 * mid.mean() and mid.mean() > 0.15          # a smooth streak
    # 3 slices over 1600 iterations: three separate discs -- pixels between them stay dark
    sg3, _ = O.motion_slices(ga, gb, 1, cam, None, 3)
    c, _ = O.render(ga, 1, ma, 1, cam, 1, iters=96, slice_geoms=sg3)
    assert (c[8, :, 0] == 0).sum() > (row == 0).sum() + 6


def test_per_ray_motion_with_equal_knots_is_the_static_scene():
    """Both knots the same scene: the interpolated transform is the transform, its computed inverse the loader's inverse up to
    rounding (adjugate over determinant here, GLM's general 4x4 cofactor inverse there) -- the static picture except for the odd
    silhouette pixel; with a lens the extra draw shifts the lens draws, so the images differ."""
    sc = O.LoadedScene(os.path.join(SCENES, "sampleScene_spec.txt"), 1)
    sc.set_resolution(40, 30)
    kg = [sc.geoms, sc.geoms, sc.geoms]
    a, la = O.render(sc.geoms, sc.n_objects, sc.mats, sc.n_materials, sc.camera, 5, iters=3, knot_geoms=kg)
    b, lb = O.render(sc.geoms, sc.n_objects, sc.mats, sc.n_materials, sc.camera, 5, iters=3)
    assert (np.abs(a - b).max(axis=2) > 1e-4).mean() < 0.03 and abs(int(la.sum()) - int(lb.sum())) <= 0.01 * int(lb.sum())
    a2, _ = O.render(sc.geoms, sc.n_objects, sc.mats, sc.n_materials, sc.camera, 5, iters=3, knot_geoms=kg, lens_radius=0.2, focal_distance=8.0)
    b2, _ = O.render(sc.geoms, sc.n_objects, sc.mats, sc.n_materials, sc.camera, 5, iters=3, lens_radius=0.2, focal_distance=8.0)
    assert not np.array_equal(a2, b2)


def test_affine_inverse():
    """o_affineInverse against numpy on random TRS transforms, and against the loader's own inverse."""
    L = O.lib()
    rng = np.random.default_rng(2)
    for _ in range(50):
        g = O.make_geom(O.CUBE, 0, rng.uniform(-5, 5, 3), rng.uniform(-3, 3, 3), rng.uniform(0.2, 6, 3))
        out = O.Mat4()
        L.o_affineInverse(C.byref(g.transform), C.byref(out))
        m = np.array([[getattr(getattr(g.transform, r), c) for c in "xyzw"] for r in "xyzw"], dtype=np.float64)
        want = np.linalg.inv(m)
        got = np.array([[getattr(getattr(out, r), c) for c in "xyzw"] for r in "xyzw"], dtype=np.float64)
        assert np.allclose(got, want, rtol=2e-5, atol=2e-6)
        glm = np.array([[getattr(getattr(g.inverseTransform, r), c) for c in "xyzw"] for r in "xyzw"], dtype=np.float64)
        assert np.allclose(got[:3], glm[:3], rtol=2e-5, atol=2e-6)


def test_per_ray_motion_rejects_unsupported_combinations():
    sc = O.LoadedScene(os.path.join(SCENES, "sampleScene_spec.txt"), 1)
    sc.set_resolution(16, 12)
    kg = [sc.geoms, sc.geoms]
    for kw in (dict(slice_geoms=[sc.geoms, sc.geoms]), dict(meshes={0: np.zeros((1, 9), np.float32)})):
        with pytest.raises(RuntimeError):
            O.render(sc.geoms, sc.n_objects, sc.mats, sc.n_materials, sc.camera, 3, knot_geoms=kg, **kw)
