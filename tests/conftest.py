import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def orc_mod():
    from oracle import oracle_py
    oracle_py.build()
    oracle_py.lib()
    return oracle_py


def _stream(width, height, n, density=1.0):
    from rebvio_amd import synth
    return synth.render_stream(width, height, n, density=density)


@pytest.fixture(scope="session")
def small_stream():
    """12 frames of a 192x144 synthetic stream (fast on the oracle, still hundreds of keylines)."""
    frames, cam = _stream(192, 144, 12, density=1.0)
    return frames, cam


@pytest.fixture(scope="session")
def c2_stream():
    """8 frames of the 640x480 bench stream."""
    frames, cam = _stream(640, 480, 8)
    return frames, cam


def params_for(mod, cam, **over):
    kw = dict(fm=cam.fm, cx=cam.cx, cy=cam.cy)
    kw.update(over)
    return mod.default_params(cam.height, cam.width, **kw)
