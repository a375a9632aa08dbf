import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (HERE, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import oracle_lib
    oracle_lib.build()
    return oracle_lib


@pytest.fixture(scope="session")
def golden():
    import json
    out = {}
    for name in ("reference_vectors", "glm_vectors", "thrust_rng_vectors"):
        with open(os.path.join(HERE, "golden", name + ".json")) as f:
            out[name] = json.load(f)
    return out


@pytest.fixture(scope="session")
def scenes_dir():
    return os.path.join(ROOT, "scenes")
