#!/usr/bin/env python3
"""Regenerate the probe-derived golden fixtures under tests/golden/.

TEST INFRASTRUCTURE ONLY.  Runs in the authoring container (needs
/root/reference for the vendored GLM and /opt/rocm for rocThrust):

    make -C oracle ref && python oracle/make_golden.py

Writes
    tests/golden/glm_vectors.json         <- oracle/_ref/glm_probe
    tests/golden/thrust_rng_vectors.json  <- oracle/_ref/thrust_probe

tests/golden/reference_vectors.json is NOT generated here: it is the data of
SURVEY.md Appendix B (values captured from the reference's implemented
functions), transcribed by hand because the reference's own sources cannot be
compiled in this image without stand-in CUDA headers.
"""
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "..", "tests", "golden")


def run(probe, out_name):
    exe = os.path.join(HERE, "_ref", probe)
    if not os.path.exists(exe):
        sys.exit(f"{exe} missing: run `make -C oracle ref` first")
    text = subprocess.run([exe], check=True, capture_output=True, text=True).stdout
    data = json.loads(text)  # validates
    path = os.path.join(GOLDEN, out_name)
    with open(path, "w") as f:
        json.dump(data, f, separators=(",", ":"))
        f.write("\n")
    print(f"wrote {os.path.relpath(path)} ({os.path.getsize(path)} bytes)")


if __name__ == "__main__":
    os.makedirs(GOLDEN, exist_ok=True)
    run("glm_probe", "glm_vectors.json")
    run("thrust_probe", "thrust_rng_vectors.json")
