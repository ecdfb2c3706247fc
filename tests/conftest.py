import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "gsplat.js_amd", "py"))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.lib()
    return O


@pytest.fixture(scope="session")
def scenes(oracle):
    """name -> (rows, data, positions), packed by the oracle's Scene.setData restatement."""
    from gsplat_hip import synth
    cache = {}

    def get(name_or_n, seed=None, **kw):
        key = (name_or_n, seed, tuple(sorted(kw.items())))
        if key not in cache:
            rows = synth.config_rows(name_or_n) if isinstance(name_or_n, str) else synth.synth_rows(name_or_n, seed, **kw)
            data, pos = oracle.scene_pack(rows)
            cache[key] = (rows, data, pos)
        return cache[key]

    return get
