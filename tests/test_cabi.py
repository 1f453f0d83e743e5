"""CPU-side checks of the C-ABI library: it loads without a GPU and exports every symbol that
include/ysmr_hip.h declares; host-only entry points behave (no compute calls here)."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import ROOT


@pytest.fixture(scope="module")
def lib():
    from ysmr_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    return _lib.lib()


def test_header_symbols_are_exported(lib):
    from ysmr_amd import _lib
    header = open(os.path.join(ROOT, "include", "ysmr_hip.h")).read()
    declared = set(re.findall(r"\b(ysmr_[a-z0-9_]+)\s*\(", header))
    assert declared == set(_lib.EXPORTS)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.ysmr_abi_version() == int(re.search(r"#define YSMR_ABI_VERSION\s+(\d+)", header).group(1)) == 15


def test_row_struct_layout():
    from ysmr_amd import _lib
    assert _lib.ROW_DTYPE.itemsize == 40
    assert [_lib.ROW_DTYPE.fields[k][1] for k in ("frame", "track_id", "x", "y", "w", "h", "angle", "disappeared")] == \
        [0, 4, 8, 16, 24, 28, 32, 36]


def test_workspace_size_and_bad_arguments(lib):
    assert lib.ysmr_detect_workspace_bytes(0, 10, 10, 10) == 0
    assert lib.ysmr_detect_workspace_bytes(4, 922, 1228, 2048) > 0
    rc = lib.ysmr_threshold_batch(None, None, 1, 10, 10, 2, 0, 5, 7, 1, None, 0)
    assert rc == 1 and b"channels" in lib.ysmr_last_error()


def test_gsff_gains_closed_form_matches_reference_formula(lib):
    from oracle.ysmr_oracle import horizon_sizes, lsf_gain
    for fps, n_min, n_max, n_f in [(30.0, 0, 30.0, 3), (29.97, 0, -1.0, 3), (25.0, 4, 24.0, 4)]:
        n_i = (ctypes.c_int32 * n_f)()
        assert lib.ysmr_gsff_gains(fps, n_min, n_max, n_f, n_i, None) == 0
        expect = horizon_sizes(n_min, fps if n_max < 0 else n_max, n_f)
        assert list(n_i) == expect
        total = sum(4 * n for n in expect)
        gains = np.zeros(total)
        assert lib.ysmr_gsff_gains(fps, n_min, n_max, n_f, n_i, gains.ctypes.data) == 0
        off = 0
        for n in expect:
            np.testing.assert_allclose(gains[off:off + 4 * n].reshape(2, 2 * n), lsf_gain(n, 1 / fps)[:2], atol=1e-12)
            off += 4 * n
    n_i = (ctypes.c_int32 * 3)()
    assert lib.ysmr_gsff_gains(30.0, 0, 2.0, 3, n_i, None) == 1      # horizons [0,1,2]: rejected


def test_threshold_params_match_oracle(oracle):
    from ysmr_amd.detect import threshold_params
    for args in [(True, 5, 2.0), (True, 5, 2.5), (True, 3, 0.0), (False, 5, 2.0), (False, 7, 0.5), (True, 0, 1.0)]:
        p = threshold_params(*args)
        assert (p.inv, p.t_low, p.t_high, p.use_high) == oracle.threshold_params(*args)


def _model_unused_order(m, used):
    """Python model of the device code's CPython-set emulation (csrc/track.hip)."""
    unused = [c for c in range(m) if c not in used]
    if (m >> 2) > len(used):
        return unused
    def insert(table, mask, key):
        perturb, i = key, key & mask
        while True:
            probes = 9 if i + 9 <= mask else 0
            for j in range(probes + 1):
                if table[i + j] < 0:
                    table[i + j] = key
                    return
            perturb >>= 5
            i = (i * 5 + 1 + perturb) & mask
    table, mask, fill = [-1] * 8, 7, 0
    for c in unused:
        insert(table, mask, c)
        fill += 1
        if fill * 5 >= mask * 3:
            minused = fill * 2 if fill > 50000 else fill * 4
            size = 8
            while size <= minused:
                size <<= 1
            new = [-1] * size
            for k in table:
                if k >= 0:
                    insert(new, size - 1, k)
            table, mask = new, size - 1
    return [k for k in table if k >= 0]


def test_cpython_set_model_matches_this_interpreter():
    rng = np.random.default_rng(0)
    for _ in range(3000):
        m = int(rng.integers(1, 700))
        k = int(rng.integers(0, m + 1))
        used = set(int(v) for v in rng.choice(m, size=k, replace=False))
        assert _model_unused_order(m, used) == list(set(range(0, m)).difference(used)), (m, sorted(used))
