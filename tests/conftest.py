import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "needs_reference: needs the reference's Python tree: /root/reference (build container) or the "
                            "staged archive oracle/_ref/pytree.zip (GPU box)")


def pytest_collection_modifyitems(config, items):
    have_ref = os.path.isdir("/root/reference/imaginaire") or os.path.exists(os.path.join(ROOT, "oracle", "_ref", "pytree.zip"))
    skip_ref = pytest.mark.skip(reason="neither /root/reference nor oracle/_ref/pytree.zip on this machine")
    for item in items:
        if "needs_reference" in item.keywords and not have_ref:
            item.add_marker(skip_ref)


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def lut():
    return json.load(open(os.path.join(ROOT, "scenedreamer_amd", "data", "mc2reduced.json")))["lut"]


@pytest.fixture(scope="session")
def scene256():
    from scenedreamer_amd import synth
    return synth.make_scene(256, 3407)


@pytest.fixture(scope="session")
def weights_full():
    """Synthetic weights including the full 2^19-row hash table (268 MB, ~3 s to regenerate)."""
    from scenedreamer_amd import synth
    return synth.make_weights(0)


def golden(name):
    return np.load(os.path.join(GOLDEN, name))


def bits(a):
    """Reinterpret float32 as int32 for exact comparison.  NaNs are canonicalised (x86 and gfx950
    generate different NaN sign/payload bits for 0/0), so NaN POSITIONS must match exactly."""
    a = np.ascontiguousarray(a, dtype=np.float32)
    b = a.view(np.int32).copy()
    b[np.isnan(a)] = 0x7FC00000
    return b
