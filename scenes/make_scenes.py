#!/usr/bin/env python3
"""Emit the benchmark / parity scenes in the reference's scene-file format.

Format: ref src/scene.cpp (grammar), README.md:160-217.  The numbers of
`sampleScene.txt` are those of the reference's bundled
data/scenes/sampleScene.txt (the scene BASELINE.json's configs are quoted on);
the other files are derived scenes the BASELINE configs call for:

  sampleScene.txt       config 1/2/4 input (all REFL 0: diffuse + one light)
  sampleScene_spec.txt  same, REFL 1 on materials 3/4/6 ("diffuse+specular")
  cornell_glass.txt     Cornell box with a glass sphere (REFR 1, REFRIOR 2.2), config 3
  cloud256.txt          256 random spheres/cubes + walls + light, config 5 (compaction stress)
  sss_blobs.txt         Cornell box with two scattering media (SCATTER 1): an index-matched wax-like sphere and a
                        scattering glass cube, for pt_options.scatter (calculateScatterAndAbsorption)

Run: python scenes/make_scenes.py   (deterministic; outputs are committed)
"""
import os

HERE = os.path.dirname(os.path.abspath(__file__))

MAT_KEYS = ["RGB", "SPECEX", "SPECRGB", "REFL", "REFR", "REFRIOR", "SCATTER", "ABSCOEFF", "RSCTCOEFF", "EMITTANCE"]


def mat(rgb, specrgb=(1, 1, 1), refl=0, refr=0, ior=0, absc=(0, 0, 0), rsct=0, emit=0, scatter=0):
    return dict(RGB=rgb, SPECEX=0, SPECRGB=specrgb, REFL=refl, REFR=refr, REFRIOR=ior, SCATTER=scatter, ABSCOEFF=absc,
                RSCTCOEFF=rsct, EMITTANCE=emit)


def fmt(v):
    if isinstance(v, (tuple, list)):
        return " ".join(fmt(x) for x in v)
    if isinstance(v, float):
        return repr(round(v, 6)).rstrip("0").rstrip(".") if v != int(v) else str(int(v))
    return str(v)


def emit_animated(path, materials, camera_frames, camera, objects_frames):
    """objects_frames: list of (kind, material, [(t, r, s) per frame]); camera_frames: [(eye, view, up) per frame]."""
    out = []
    for i, m in enumerate(materials):
        out.append(f"MATERIAL {i}")
        for k in MAT_KEYS:
            out.append(f"{k} {fmt(m[k])}")
        out.append("")
    out += ["CAMERA", f"RES {camera['res'][0]} {camera['res'][1]}", f"FOVY {fmt(camera['fovy'])}",
            f"ITERATIONS {camera['iterations']}", f"FILE {camera['file']}"]
    for f, (eye, view, up) in enumerate(camera_frames):
        out += [f"frame {f}", f"EYE {fmt(eye)}", f"VIEW {fmt(view)}", f"UP {fmt(up)}"]
    out.append("")
    for i, (kind, material, frames) in enumerate(objects_frames):
        out += [f"OBJECT {i}", kind, f"material {material}"]
        for f, (t, r, sc) in enumerate(frames):
            out += [f"frame {f}", f"TRANS {fmt(t)}", f"ROTAT {fmt(r)}", f"SCALE {fmt(sc)}"]
        out.append("")
    with open(os.path.join(HERE, path), "w") as fh:
        fh.write("\n".join(out))
    print("wrote", path, len(objects_frames), "objects,", len(camera_frames), "frames")


def emit(path, materials, camera, objects):
    out = []
    for i, m in enumerate(materials):
        out.append(f"MATERIAL {i}")
        for k in MAT_KEYS:
            out.append(f"{k} {fmt(m[k])}")
        out.append("")
    out.append("CAMERA")
    out.append(f"RES {camera['res'][0]} {camera['res'][1]}")
    out.append(f"FOVY {fmt(camera['fovy'])}")
    out.append(f"ITERATIONS {camera['iterations']}")
    out.append(f"FILE {camera['file']}")
    out.append("frame 0")
    out.append(f"EYE {fmt(camera['eye'])}")
    out.append(f"VIEW {fmt(camera['view'])}")
    out.append(f"UP {fmt(camera['up'])}")
    out.append("")
    for i, (kind, material, t, r, s) in enumerate(objects):
        out.append(f"OBJECT {i}")
        out.append(kind)
        out.append(f"material {material}")
        out.append("frame 0")
        out.append(f"TRANS {fmt(t)}")
        out.append(f"ROTAT {fmt(r)}")
        out.append(f"SCALE {fmt(s)}")
        out.append("")
    with open(os.path.join(HERE, path), "w") as f:
        f.write("\n".join(out))
    print("wrote", path, len(objects), "objects")


# ---- the bundled scene's values (ref data/scenes/sampleScene.txt)
SAMPLE_MATERIALS = [
    mat((1, 1, 1)),                                     # 0 white diffuse
    mat((.63, .06, .04)),                               # 1 red diffuse
    mat((.15, .48, .09)),                               # 2 green diffuse
    mat((.63, .06, .04), ior=2),                        # 3 red glossy
    mat((1, 1, 1), ior=2),                              # 4 white glossy
    mat((0, 0, 0), refr=1, ior=2.2, absc=(.02, 5.1, 5.7), rsct=13),  # 5 glass
    mat((.15, .48, .09), ior=2.6),                      # 6 green glossy
    mat((1, 1, 1), specrgb=(0, 0, 0), emit=1),          # 7 light
    mat((1, 1, 1), specrgb=(0, 0, 0), emit=15),         # 8 light
]
SAMPLE_CAMERA = dict(res=(800, 800), fovy=25, iterations=5000, file="test.bmp", eye=(0, 4.5, 12), view=(0, 0, -1),
                     up=(0, 1, 0))
WALLS = [
    ("cube", 0, (0, 0, 0), (0, 0, 90), (.01, 10, 10)),      # floor
    ("cube", 0, (0, 5, -5), (0, 90, 0), (.01, 10, 10)),     # back wall
    ("cube", 0, (0, 10, 0), (0, 0, 90), (.01, 10, 10)),     # ceiling
    ("cube", 1, (-5, 5, 0), (0, 0, 0), (.01, 10, 10)),      # left (red)
    ("cube", 2, (5, 5, 0), (0, 0, 0), (.01, 10, 10)),       # right (green)
]
LIGHT = ("cube", 8, (0, 10, 0), (0, 0, 90), (.3, 3, 3))
SAMPLE_OBJECTS = WALLS + [
    ("sphere", 4, (0, 2, 0), (0, 180, 0), (3, 3, 3)),
    ("sphere", 3, (2, 5, 2), (0, 180, 0), (2.5, 2.5, 2.5)),
    ("sphere", 6, (-2, 5, -2), (0, 180, 0), (3, 3, 3)),
    LIGHT,
]


# ---- integer RNG of the renderer (ref src/intersections.h:26-34 + minstd), for the cloud scene
def _hash(a):
    a &= 0xFFFFFFFF
    a = ((a + 0x7ed55d16) + (a << 12)) & 0xFFFFFFFF
    a = ((a ^ 0xc761c23c) ^ (a >> 19)) & 0xFFFFFFFF
    a = ((a + 0x165667b1) + (a << 5)) & 0xFFFFFFFF
    a = ((a + 0xd3a2646c) ^ (a << 9)) & 0xFFFFFFFF
    a = ((a + 0xfd7046c5) + (a << 3)) & 0xFFFFFFFF
    a = ((a ^ 0xb55a4f09) ^ (a >> 16)) & 0xFFFFFFFF
    return a


class Minstd:
    def __init__(self, seed):
        self.x = seed % 2147483647 or 1

    def u01(self):
        self.x = (48271 * self.x) % 2147483647
        return (self.x - 1) / 2147483648.0

    def uni(self, a, b):
        return a + (b - a) * self.u01()


def cloud(n=256, seed=565):
    rng = Minstd(_hash(seed))
    materials = list(SAMPLE_MATERIALS)
    materials.append(mat((.9, .9, .9), refl=1))                    # 9 mirror
    materials.append(mat((.2, .35, .8)))                           # 10 blue diffuse
    materials.append(mat((.85, .7, .2)))                           # 11 yellow diffuse
    objects = list(WALLS)
    for _ in range(n - len(WALLS) - 1):
        kind = "sphere" if rng.u01() < 0.5 else "cube"
        t = (round(rng.uni(-4.5, 4.5), 4), round(rng.uni(0.5, 9.0), 4), round(rng.uni(-4.5, 4.5), 4))
        s = round(rng.uni(0.2, 0.8), 4)
        r = (round(rng.uni(0, 360), 3), round(rng.uni(0, 360), 3), round(rng.uni(0, 360), 3))
        u = rng.u01()
        if u < 0.7:
            m = [0, 1, 2, 10, 11][int(rng.u01() * 5) % 5]
        elif u < 0.9:
            m = 9
        else:
            m = 5
        objects.append((kind, m, t, r, (s, s, s)))
    objects.append(LIGHT)
    return materials, objects


if __name__ == "__main__":
    emit("sampleScene.txt", SAMPLE_MATERIALS, SAMPLE_CAMERA, SAMPLE_OBJECTS)

    spec = [dict(m) for m in SAMPLE_MATERIALS]
    for i in (3, 4, 6):
        spec[i]["REFL"] = 1
    emit("sampleScene_spec.txt", spec, dict(SAMPLE_CAMERA, file="spec.bmp"), SAMPLE_OBJECTS)

    glass_objects = WALLS + [
        ("sphere", 5, (0, 2.5, 0.5), (0, 0, 0), (4, 4, 4)),     # glass sphere
        ("sphere", 3, (-2.6, 1.2, -2), (0, 0, 0), (2.4, 2.4, 2.4)),
        ("cube", 6, (2.7, 1.5, -2), (0, 30, 0), (2.2, 3, 2.2)),
        LIGHT,
    ]
    emit("cornell_glass.txt", SAMPLE_MATERIALS, dict(SAMPLE_CAMERA, res=(1920, 1080), iterations=1024, file="glass.bmp"),
         glass_objects)

    # three animation frames: the spheres move, the camera dollies (ref: src/sceneStructs.h:21-30,50-61)
    anim_objects = []
    for (kind, material, t, r, sc) in SAMPLE_OBJECTS:
        frames = []
        for f in range(3):
            tt = (t[0] + 0.6 * f, t[1] + 0.3 * f, t[2] - 0.4 * f) if kind == "sphere" else t
            frames.append((tt, (r[0], r[1] + (25 * f if kind == "sphere" else 0), r[2]), sc))
        anim_objects.append((kind, material, frames))
    anim_cam = [((0, 4.5, 12 - 1.5 * f), (0.05 * f, 0, -1), (0, 1, 0)) for f in range(3)]
    emit_animated("sampleScene_anim.txt", spec, anim_cam, dict(SAMPLE_CAMERA, res=(320, 240), iterations=8, file="anim.bmp"),
                  anim_objects)

    sss_materials = list(SAMPLE_MATERIALS) + [
        mat((.95, .85, .7), scatter=1, absc=(.05, .25, .6), rsct=3),                 # 9 wax: index-matched medium
        mat((0, 0, 0), refr=1, ior=1.5, scatter=1, absc=(.4, .1, .05), rsct=1.5),    # 10 cloudy glass
    ]
    sss_objects = WALLS + [
        ("sphere", 9, (-1.8, 2.2, 0.5), (0, 0, 0), (3.6, 3.6, 3.6)),
        ("cube", 10, (2.2, 1.8, -0.5), (0, 25, 0), (2.6, 3.6, 2.6)),
        ("sphere", 3, (0, 6.5, -2.5), (0, 0, 0), (2, 2, 2)),
        LIGHT,
    ]
    emit("sss_blobs.txt", sss_materials, dict(SAMPLE_CAMERA, res=(800, 800), iterations=1000, file="sss.bmp"), sss_objects)

    cm, co = cloud()
    emit("cloud256.txt", cm, dict(SAMPLE_CAMERA, res=(1920, 1080), iterations=4096, file="cloud.bmp"), co)
