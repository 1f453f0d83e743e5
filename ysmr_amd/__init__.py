"""ysmr_amd -- MI355X-native implementation of YSMR's per-frame detect-and-link hot path.

The public surface mirrors the reference package (``ysmr/__init__.py`` re-exports ``main``,
``plot_functions``, ``track_eval``): ``ysmr()``, ``analyse()``, ``track_bacteria()``,
``select_tracks()``, ``evaluate_tracks()``, ``CentroidTracker``, ``GaussianSumFIR`` and the tracking.ini helpers.  Importing this package does
not touch the GPU; the HIP library (``csrc/libysmr_hip.so``) is loaded on first use and there is no
CPU fallback.
"""
import os as _os

# Kernel arguments in device memory (the HIP runtime reads this when it is loaded, i.e. at `import torch`): the link is one
# short kernel per frame whose first instructions read their arguments -- from host memory that is a PCIe round trip per
# launch (measured: 90 k -> 62 k frames/s end to end with HIP_FORCE_DEV_KERNARG=0).  A value the user set is kept; the
# default only takes effect when this package is imported before torch (bench.py sets it itself) -- ROCm 7.2's own default
# is device memory as well.
_os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")

__version__ = "0.1.0"

__all__ = ["ysmr", "analyse", "track_bacteria", "select_tracks", "evaluate_tracks", "CentroidTracker", "GaussianSumFIR", "get_configs",
           "create_configs", "default_settings"]


def __getattr__(name):   # lazy: keep `import ysmr_amd.synth` usable without torch
    if name in ("ysmr", "analyse"):
        from . import main
        return getattr(main, name)
    if name == "track_bacteria":
        from .track_eval import track_bacteria
        return track_bacteria
    if name == "select_tracks":
        from .select import select_tracks
        return select_tracks
    if name == "evaluate_tracks":
        from .evaluate import evaluate_tracks
        return evaluate_tracks
    if name == "CentroidTracker":
        from .tracker import CentroidTracker
        return CentroidTracker
    if name == "GaussianSumFIR":
        from .gsff import GaussianSumFIR
        return GaussianSumFIR
    if name in ("get_configs", "create_configs", "default_settings"):
        from . import helper_file
        return getattr(helper_file, name)
    raise AttributeError(name)
