"""ctypes binding of libysmr_hip.so (include/ysmr_hip.h).

The HIP library is the product: there is no CPU fallback.  If the shared object has not been
built (``python -c "import __graft_entry__ as g; g.build()"`` or ``make -C ysmr_amd/csrc``) every
entry point raises :class:`YsmrLibraryError`.
"""
from __future__ import annotations

import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("YSMR_HIP_LIB") or os.path.join(_HERE, "csrc", "libysmr_hip.so")  # env: tuning builds

YSMR_OK = 0
YSMR_ERR_ARG = 1
DET_OVERFLOW = 1
DET_ARENA = 2
DET_STALLED = 4
#: ``cv_flavour`` bits (include/ysmr_hip.h): which OpenCV release the a1 / a6 arithmetic follows
CV_DEFAULT, CV_ANGLE_PRE451, CV_GRAY_3X = 0, 1, 2
#: scheduling hint in the same argument: the one-launch link runs on another stream beside the call (YSMR_BESIDE_LINK)
BESIDE_LINK = 4
#: ... and for the one-launch-per-batch link, which holds one compute unit (YSMR_BESIDE_BATCH_LINK)
BESIDE_BATCH_LINK = 8
#: ... and for the two-launch link of large tables (YSMR_BESIDE_SPLIT_LINK): the matrix-pipe threshold kernel on half the units
BESIDE_SPLIT_LINK = 16
#: test hook in a detection workspace's header (include/ysmr_hip.h)
WS_FAULT_OFFSET, WS_FAULT_RESIDUE_STALL = 32, 0xFA17057A


def cv_flavour_of(version):
    """``cv_flavour`` for an OpenCV version string ('3.4.18', '4.5.0', '4.10.0', ...) or None (the default:
    OpenCV >= 4.5.1).  Also accepts the flag value itself."""
    if version is None or version is False:
        return CV_DEFAULT
    if isinstance(version, int):
        return version
    parts = [int("".join(ch for ch in p if ch.isdigit()) or 0) for p in str(version).split(".")[:3]]
    parts += [0] * (3 - len(parts))
    flags = CV_GRAY_3X if parts[0] < 4 else 0
    if tuple(parts) < (4, 5, 1):
        flags |= CV_ANGLE_PRE451
    return flags

#: numpy view of ``struct ysmr_row`` (40 bytes)
ROW_DTYPE = np.dtype([("frame", "<i4"), ("track_id", "<i4"), ("x", "<f8"), ("y", "<f8"),
                      ("w", "<f4"), ("h", "<f4"), ("angle", "<f4"), ("disappeared", "<i4")])

ABI_VERSION = 15   # YSMR_ABI_VERSION of include/ysmr_hip.h these argtypes were written against

EXPORTS = ("ysmr_abi_version", "ysmr_last_error", "ysmr_detect_workspace_bytes", "ysmr_detect_workspace_init",
           "ysmr_threshold_batch", "ysmr_threshold_batch_variant", "ysmr_threshold_timing", "ysmr_threshold_workgroups", "ysmr_mean_threshold_state_bytes", "ysmr_mean_threshold_batch",
           "ysmr_components_batch", "ysmr_detect_batch", "ysmr_gsff_gains", "ysmr_tracker_create", "ysmr_tracker_destroy",
           "ysmr_tracker_reset", "ysmr_tracker_update", "ysmr_tracker_run", "ysmr_tracker_fused", "ysmr_tracker_batched", "ysmr_tracker_link_mode", "ysmr_tracker_prepare", "ysmr_tracker_peek",
           "ysmr_tracker_info", "ysmr_rows_sort_workspace_bytes", "ysmr_rows_sort", "ysmr_rows_csv_bound",
           "ysmr_rows_format_csv", "ysmr_rows_write_csv", "ysmr_rows_write_csv_columns", "ysmr_rows_format_device_workspace_bytes", "ysmr_rows_format_device", "ysmr_rows_format_csv_devicelike", "ysmr_rows_stream_create", "ysmr_rows_stream_push", "ysmr_rows_stream_count", "ysmr_rows_stream_finish", "ysmr_rows_stream_destroy", "ysmr_rows_columns", "ysmr_select_workspace_bytes", "ysmr_select_tracks",
           "ysmr_evaluate_workspace_bytes", "ysmr_evaluate_tracks", "ysmr_unpack_dib_batch", "ysmr_file_read")

SELECT_OK, SELECT_TOO_SHORT, SELECT_TOO_SHORT_CLEANED, SELECT_NONE = 0, 1, 2, 3


class SelectParams(ctypes.Structure):
    """``struct ysmr_select_params``"""
    _fields_ = [(k, ctypes.c_double) for k in ("area_lo", "area_hi", "area_factor", "q_area", "motility_stop_fraction",
                                               "max_empty_ratio", "ratio_min", "ratio_max", "edge_fraction")] + \
               [(k, ctypes.c_int32) for k in ("min_length_frames", "limit_frames", "limit_exact", "omit_motility",
                                              "max_holes", "max_recursion", "frame_height", "frame_width")]


class SelectSummary(ctypes.Structure):
    """``struct ysmr_select_summary``"""
    _fields_ = [("status", ctypes.c_int32), ("outliers_used", ctypes.c_int32),
                ("rows_before", ctypes.c_longlong), ("tracks_before", ctypes.c_longlong),
                ("rows_after", ctypes.c_longlong), ("tracks_after", ctypes.c_longlong),
                ("area_lo", ctypes.c_double), ("area_hi", ctypes.c_double), ("q1_dist", ctypes.c_double),
                ("q3_dist", ctypes.c_double), ("dist_fence", ctypes.c_double), ("dist_outliers", ctypes.c_longlong),
                ("kick_reasons", ctypes.c_longlong * 9), ("good_tracks", ctypes.c_longlong),
                ("rows_selected", ctypes.c_longlong)]


class EvaluateParams(ctypes.Structure):
    """``struct ysmr_evaluate_params``"""
    _fields_ = [("pixel_per_micrometre", ctypes.c_double), ("fps", ctypes.c_double), ("min_turn_angle", ctypes.c_double),
                ("angle_lag", ctypes.c_int32), ("reach_lag", ctypes.c_int32), ("median_kernel", ctypes.c_int32),
                ("reserved", ctypes.c_int32)]


class YsmrLibraryError(RuntimeError):
    pass


class YsmrCapacityError(YsmrLibraryError):
    """A frame held more components than ``max_det`` or the tracker more tracks than ``capacity``."""


_lib = None


def lib():
    """Load (once) and return the ctypes handle with argtypes set."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise YsmrLibraryError(
            f"{LIB_PATH} is missing: build the HIP library first (make -C ysmr_amd/csrc, or "
            "__graft_entry__.build()). ysmr_amd has no CPU fallback.")
    try:
        L = ctypes.CDLL(LIB_PATH)
    except OSError as exc:  # pragma: no cover
        raise YsmrLibraryError(f"cannot load {LIB_PATH}: {exc}") from exc
    vp, ci, cd = ctypes.c_void_p, ctypes.c_int, ctypes.c_double
    L.ysmr_abi_version.restype = ci
    if os.environ.get("HIP_FORCE_DEV_KERNARG") == "0":
        # (a launch's first instructions read its arguments: from host memory that is a PCIe round trip per launch --
        # 90 k -> 62 k frames/s with one launch per frame; ROCm's default and this package's is 1)
        import logging
        logging.getLogger("ysmr").warning("HIP_FORCE_DEV_KERNARG=0 is in force: kernel arguments in host memory cost the "
                                          "per-frame link about a third of its rate; unset it or set it to 1")
    if L.ysmr_abi_version() != ABI_VERSION:
        raise YsmrLibraryError(f"{LIB_PATH} has ABI version {L.ysmr_abi_version()}, this host code binds version "
                               f"{ABI_VERSION} (include/ysmr_hip.h): rebuild the library (make -C ysmr_amd/csrc)")
    L.ysmr_last_error.restype = ctypes.c_char_p
    L.ysmr_detect_workspace_bytes.argtypes = [ci, ci, ci, ci]
    L.ysmr_detect_workspace_bytes.restype = ctypes.c_size_t
    L.ysmr_detect_workspace_init.argtypes = [vp, vp, ctypes.c_size_t]
    L.ysmr_threshold_batch.argtypes = [vp, vp, ci, ci, ci, ci, ci, ci, ci, ci, vp, ci]
    L.ysmr_threshold_batch_variant.argtypes = [vp, vp, ci, ci, ci, ci, ci, ci, ci, ci, vp, ci, ci]
    L.ysmr_threshold_timing.argtypes = [vp, vp]
    L.ysmr_threshold_workgroups.argtypes = [ctypes.c_int]
    L.ysmr_mean_threshold_state_bytes.argtypes = [ci]
    L.ysmr_mean_threshold_state_bytes.restype = ctypes.c_size_t
    L.ysmr_mean_threshold_batch.argtypes = [vp, vp, ci, ci, ci, ci, ci, cd, ci, vp, vp, vp, vp, ci]
    L.ysmr_components_batch.argtypes = [vp, ci, ci, ci, vp, ctypes.c_size_t, vp, vp, vp, vp, vp, vp, ci, vp, ci]
    L.ysmr_detect_batch.argtypes = [vp, vp, ci, ci, ci, ci, ci, ci, ci, ci, vp, ctypes.c_size_t, vp, vp, vp,
                                    vp, vp, vp, ci, vp, ci]
    L.ysmr_gsff_gains.argtypes = [cd, ci, cd, ci, vp, vp]
    L.ysmr_tracker_create.argtypes = [cd, cd, ci, cd, ci, ci, ci, ci, vp, ctypes.POINTER(vp)]
    L.ysmr_tracker_destroy.argtypes = [vp]
    L.ysmr_tracker_reset.argtypes = [vp, vp]
    L.ysmr_tracker_update.argtypes = [vp, vp, vp, ci, ci, vp, ctypes.c_int32, vp, vp, vp, vp, vp, vp]
    L.ysmr_tracker_peek.argtypes = [vp, vp, vp, vp, vp, vp]
    L.ysmr_tracker_run.argtypes = [vp, vp, vp, vp, ci, ctypes.c_int32, vp, ctypes.c_int64, vp]
    L.ysmr_tracker_fused.argtypes = [vp]
    L.ysmr_tracker_batched.argtypes = [vp]
    L.ysmr_tracker_link_mode.argtypes = [vp, ci]
    L.ysmr_tracker_prepare.argtypes = [vp, vp, vp, vp, ci, ci]
    L.ysmr_tracker_info.argtypes = [vp, vp, ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int32),
                                    ctypes.POINTER(ctypes.c_int32)]
    L.ysmr_rows_sort_workspace_bytes.argtypes = [ctypes.c_longlong]
    L.ysmr_rows_sort_workspace_bytes.restype = ctypes.c_size_t
    L.ysmr_rows_sort.argtypes = [vp, vp, ctypes.c_longlong, vp, ctypes.c_size_t, vp]
    L.ysmr_rows_csv_bound.argtypes = [ctypes.c_longlong, ci]
    L.ysmr_rows_csv_bound.restype = ctypes.c_size_t
    L.ysmr_rows_format_csv.argtypes = [vp, ctypes.c_longlong, ci, ci, ci, vp, ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t)]
    L.ysmr_file_read.argtypes = [ci, vp, ctypes.c_size_t, ctypes.c_longlong, ci]
    L.ysmr_rows_write_csv.argtypes = [vp, ctypes.c_longlong, ci, ci, ci, ctypes.c_char_p, ctypes.POINTER(ctypes.c_size_t)]
    L.ysmr_rows_write_csv_columns.argtypes = [vp, ctypes.c_longlong, ci, ci, ci, ctypes.c_char_p, ctypes.POINTER(ctypes.c_size_t)] + [vp] * 7
    L.ysmr_rows_format_device_workspace_bytes.argtypes = [ctypes.c_longlong]
    L.ysmr_rows_format_device_workspace_bytes.restype = ctypes.c_size_t
    L.ysmr_rows_format_device.argtypes = [vp, vp, ctypes.c_longlong, ci, ci, vp, ctypes.c_size_t, vp, ctypes.c_size_t, vp] + [vp] * 7 + [vp]
    L.ysmr_rows_format_csv_devicelike.argtypes = [vp, ctypes.c_longlong, ci, ci, vp, ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t), vp,
                                                  ctypes.POINTER(ctypes.c_longlong)]
    L.ysmr_rows_stream_create.argtypes = [ci, ci, ctypes.POINTER(vp)]
    L.ysmr_rows_stream_push.argtypes = [vp, vp, ctypes.c_longlong]
    L.ysmr_rows_stream_count.argtypes = [vp]
    L.ysmr_rows_stream_count.restype = ctypes.c_longlong
    L.ysmr_rows_stream_finish.argtypes = [vp, ci, ctypes.c_char_p, ctypes.POINTER(ctypes.c_size_t)] + [vp] * 7
    L.ysmr_rows_stream_destroy.argtypes = [vp]
    L.ysmr_rows_columns.argtypes = [vp, ctypes.c_longlong, ci, vp, vp, vp, vp, vp, vp, vp]
    L.ysmr_select_workspace_bytes.argtypes = [ctypes.c_longlong, ci]
    L.ysmr_select_workspace_bytes.restype = ctypes.c_size_t
    L.ysmr_select_tracks.argtypes = [vp, ctypes.c_longlong, vp, vp, vp, vp, vp, vp, ctypes.POINTER(SelectParams), vp,
                                     ctypes.c_size_t, vp, vp, ctypes.POINTER(SelectSummary)]
    L.ysmr_unpack_dib_batch.argtypes = [vp, vp, ci, ctypes.c_size_t, ci, ci, ci, ci, ci, vp, vp]
    L.ysmr_evaluate_workspace_bytes.argtypes = [ctypes.c_longlong]
    L.ysmr_evaluate_workspace_bytes.restype = ctypes.c_size_t
    L.ysmr_evaluate_tracks.argtypes = [vp, ctypes.c_longlong, vp, vp, vp, vp, vp, vp, ctypes.POINTER(EvaluateParams), vp,
                                       ctypes.c_size_t, vp, vp, vp, vp, vp, vp, vp, vp, vp, ctypes.POINTER(ctypes.c_longlong)]
    for name in EXPORTS:
        if name not in ("ysmr_evaluate_workspace_bytes", "ysmr_select_workspace_bytes", "ysmr_last_error", "ysmr_detect_workspace_bytes", "ysmr_abi_version",
                        "ysmr_rows_sort_workspace_bytes", "ysmr_rows_csv_bound", "ysmr_rows_stream_count",
                        "ysmr_mean_threshold_state_bytes"):
            getattr(L, name).restype = ci
    _lib = L
    return L


def check(rc, what):
    """Map a non-zero status to an exception carrying ysmr_last_error()."""
    if rc != YSMR_OK:
        msg = lib().ysmr_last_error().decode("utf-8", "replace")
        raise YsmrLibraryError(f"{what} failed (code {rc}): {msg}")


def stream_ptr(device=None):
    """The current torch HIP stream OF ``device`` as a hipStream_t value (``None``: of the current
    device).  Every entry point that launches work passes the device its buffers live on: the current
    device of a fresh worker is cuda:0 whatever GPU its job was dealt to."""
    import torch
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def on(device):
    """Context: make ``device`` the current HIP device (kernels are launched on the current device; a
    stream of another device is an invalid handle there)."""
    import torch
    if torch.cuda.device_count() == 0:      # a host without GPUs: let the first device call report it
        import contextlib
        return contextlib.nullcontext()
    return torch.cuda.device(torch.device(device))
