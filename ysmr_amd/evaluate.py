"""``evaluate_tracks`` -- per-track statistics of the selected tracks on the device.

Host mirror of the reference's function (ysmr/track_eval.py:846-1318): same signature, same argument checks and
log lines, same result files (``<name>_statistics.csv``, ``<name>_analysed.csv``, written with the reference's own
``DataFrame.to_csv`` call) and the same return value ``(df, df_stats)``.  Everything numerical happens in
``ysmr_evaluate_tracks`` (``csrc/evaluate.hip``).  The plots (matplotlib / seaborn figures: angle histogram, rose
graph, overview, violin plots) are presentation, not part of the HIP path; settings that ask for them are noted
in the log and skipped.
"""
from __future__ import annotations

import ctypes
import logging
import os

import numpy as np

from . import _lib
from .helper_file import create_results_folder, get_configs, get_data, save_df_to_csv

__all__ = ["evaluate_tracks", "evaluate_params", "evaluate_columns"]

STATS_COLUMNS = ["Turn Points (TP/s)", "Distance (µm)", "Speed (µm/s)", "Time (s)", "Displacement (µm)",
                 "Perc. Motile", "Arc-Chord Ratio", "Bacteria Length", "Displacement divided by length",
                 "Motility Phenotype", "TRACK_ID", "Median Speed"]
ROW_COLUMNS = ["TRACK_ID", "POSITION_T", "POSITION_X", "POSITION_Y", "WIDTH", "HEIGHT", "DEGREES_ANGLE", "angle_diff",
               "moving", "turn_points", "tp_of_tracks", "travelled_dist", "motility_phenotype"]
_PLOT_KEYS = ("save large plots", "save rose plot", "save time violin plot", "save acr violin plot",
              "save length violin plot", "save turning point violin plot", "save speed violin plot",
              "save angle distribution plot / bins", "save displacement violin plot", "save percent motile plot")


def evaluate_params(settings, fps) -> _lib.EvaluateParams:
    """The scalars ``ysmr_evaluate_tracks`` needs (track_eval.py:931-934, 941, 957, 990-1000)."""
    p = _lib.EvaluateParams()
    p.pixel_per_micrometre = float(settings["pixel per micrometre"])
    p.fps = float(fps)
    p.min_turn_angle = float(settings["minimal angle in degrees for turning point"])
    p.angle_lag = int(settings["compare angle between n frames"])
    spans = [10] + [v / 2 for v in (settings["minimal length in seconds"], settings["limit track length to x seconds"])
                    if 0 < v / 2 < 10]
    p.reach_lag = int(round(fps * min(spans), 0))
    second = int(round(fps, 0))
    p.median_kernel = second + 1 if second % 2 == 0 else second
    return p


def evaluate_columns(df, params: _lib.EvaluateParams, device="cuda:0"):
    """Run ``ysmr_evaluate_tracks`` on the table's six input columns.  Returns (dict of per-row arrays,
    statistics array [tracks, 12])."""
    import torch
    n = len(df)
    L = _lib.lib()
    dev = torch.device(device)
    with _lib.on(dev):
        def up(name, dtype):
            return torch.from_numpy(np.ascontiguousarray(df[name].to_numpy(), dtype=dtype).view(
                np.int32 if dtype == np.uint32 else dtype)).to(dev)
        cols = [up("TRACK_ID", np.uint32), up("POSITION_T", np.uint32), up("POSITION_X", np.float64),
                up("POSITION_Y", np.float64), up("WIDTH", np.float64), up("HEIGHT", np.float64)]
        ws = torch.empty(max(L.ysmr_evaluate_workspace_bytes(n), 256), dtype=torch.uint8, device=dev)
        f64 = lambda: torch.empty(max(n, 1), dtype=torch.float64, device=dev)   # noqa: E731
        i8 = lambda: torch.empty(max(n, 1), dtype=torch.int8, device=dev)       # noqa: E731
        out = {"WIDTH": f64(), "HEIGHT": f64(), "angle_diff": torch.empty(max(n, 1), dtype=torch.int32, device=dev),
               "moving": i8(), "turn_points": i8(), "tp_of_tracks": f64(), "travelled_dist": f64(), "motility_phenotype": i8()}
        stats = torch.empty(max(n, 1), 12, dtype=torch.float64, device=dev)
        n_tracks = ctypes.c_longlong(0)
        rc = L.ysmr_evaluate_tracks(_lib.stream_ptr(dev), n, *[c.data_ptr() for c in cols], ctypes.byref(params), ws.data_ptr(),
                                    ws.numel(), out["WIDTH"].data_ptr(), out["HEIGHT"].data_ptr(), out["angle_diff"].data_ptr(),
                                    out["moving"].data_ptr(), out["turn_points"].data_ptr(), out["tp_of_tracks"].data_ptr(),
                                    out["travelled_dist"].data_ptr(), out["motility_phenotype"].data_ptr(), stats.data_ptr(),
                                    ctypes.byref(n_tracks))
        _lib.check(rc, "ysmr_evaluate_tracks")
        return {k: v[:n].cpu().numpy() for k, v in out.items()}, stats[: n_tracks.value].cpu().numpy()


def evaluate_tracks(path_to_file, results_directory=None, df=None, settings=None, fps=None, device="cuda:0", **_):
    """Statistics of the selected tracks (track_eval.py:846-1318): returns ``(df, df_stats)`` -- the table with the
    per-row columns of ``<name>_analysed.csv`` and the per-track table of ``<name>_statistics.csv`` (plus its
    'Categories (...)' column, 'All', as upstream) -- or None after logging the reason."""
    import pandas as pd
    logger = logging.getLogger("ysmr").getChild(__name__)
    settings = get_configs(settings)
    if settings is None:
        logger.critical("No settings provided.")
        return None
    if fps is None or fps <= 0 or settings["force tracking.ini fps settings"]:
        fps = settings["frames per second"]
        if not fps > 0:
            logger.critical("fps value is negative or zero; cannot continue.")
            return None
    if results_directory is None:
        results_directory = create_results_folder(path_to_file)
    file_name = os.path.splitext(os.path.basename(path_to_file))[0]
    if not isinstance(df, pd.DataFrame):
        df = get_data(path_to_file)
    if df is None:
        logger.critical("Error reading data frame from file {}".format(path_to_file))
        return None
    if len(df) == 0:
        logger.critical("Error reading data frame from file {}".format(path_to_file))
        return None
    table = df.reset_index(drop=True)
    first = table.groupby("TRACK_ID")["POSITION_T"].transform("first")
    if (table["POSITION_T"].astype(np.int64) < first.astype(np.int64)).any():
        logger.critical("POSITION_T contains negative values")
        return None
    try:
        rows, stats = evaluate_columns(table, evaluate_params(settings, fps), device=device)
    except (_lib.YsmrLibraryError, RuntimeError) as exc:
        logger.critical("Device path failed for file {}: {}".format(path_to_file, exc))
        return None
    out = table.loc[:, ["TRACK_ID", "POSITION_T", "POSITION_X", "POSITION_Y", "WIDTH", "HEIGHT", "DEGREES_ANGLE"]].copy()
    for name in ("WIDTH", "HEIGHT", "angle_diff", "moving", "turn_points", "tp_of_tracks", "travelled_dist",
                 "motility_phenotype"):
        out[name] = rows[name]
    out = out.loc[:, ROW_COLUMNS]
    track_ids = stats[:, 10].astype(table["TRACK_ID"].dtype)
    index = pd.Index(track_ids, name="TRACK_ID")
    df_stats = pd.DataFrame({name: stats[:, k] for k, name in enumerate(STATS_COLUMNS)}, index=index)
    df_stats["Bacteria Length"] = df_stats["Bacteria Length"].astype(np.float32)     # (pandas' float16 group mean)
    df_stats["Motility Phenotype"] = df_stats["Motility Phenotype"].astype(np.int8)
    df_stats["TRACK_ID"] = track_ids
    save_path = os.path.join(results_directory, file_name) + "_{}{}"
    if settings["store generated statistical .csv file"]:
        save_df_to_csv(df=df_stats, save_path=save_path.format("statistics", ".csv"))
    share = [float((df_stats["Motility Phenotype"] == k).sum()) / len(df_stats) for k in (0, 1, 2)]
    logger.info("Nonmotile: {:.2%}, twitching: {:.2%}, motile: {:.2%}".format(*share))
    q1, q2, q3 = np.quantile(df_stats["Time (s)"], (0.25, 0.5, 0.75))
    logger.debug("Time duration of selected tracks min: {:.3f}, max: {:.3f}, Quantiles (25/50/75%): {:.3f}, {:.3f}, {:.3f}"
                 "".format(df_stats["Time (s)"].min(), df_stats["Time (s)"].max(), q1, q2, q3))
    # the category column the reference adds for its violin plots (track_eval.py:1150-1183)
    wanted = settings["split results by (Turn Points / Distance / Speed / Time / Displacement / perc. motile)"]
    parameter = next((name for name in STATS_COLUMNS if str(wanted).lower() in name.lower()), None)
    if not parameter:
        logger.warning("Setting 'split results by parameter (Turn Points / Distance / Speed / Time / Displacement / % motile)' "
                       "could not be assigned, reverted to 'perc. motile'.")
        parameter = STATS_COLUMNS[5]
    df_stats["Categories ({})".format(parameter)] = "All"
    asked = [k for k in _PLOT_KEYS if settings.get(k)]
    if asked:
        logger.info("Plots are not part of the HIP path; skipped: {}".format(", ".join(asked)))
    if settings["store final analysed .csv file"]:
        save_df_to_csv(df=out, save_path=save_path.format("analysed", ".csv"))
    logging.info("Done evaluating file {}".format(file_name))
    return out, df_stats
