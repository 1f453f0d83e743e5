import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden(name):
    return np.load(os.path.join(GOLDEN, name))


def tracker_frames(g):
    """Yield (det (M,2), info (M,3)) per frame from a tracker_*.npz fixture."""
    off = g["det_off"]
    for f in range(len(off) - 1):
        yield g["det"][off[f]:off[f + 1]], g["det_info"][off[f]:off[f + 1]]


def rects_of(det, info):
    return [((float(d[0]), float(d[1])), (float(i[0]), float(i[1]), float(i[2]))) for d, i in zip(det, info)]


@pytest.fixture(scope="session")
def oracle():
    from oracle import ysmr_oracle
    ysmr_oracle.build()
    return ysmr_oracle
