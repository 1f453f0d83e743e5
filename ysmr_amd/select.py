"""``select_tracks`` on the device: the stage that consumes the detect-and-link table.

Mirror of ``select_tracks`` / ``find_good_tracks`` (ysmr/track_eval.py:408-843): same arguments,
same log messages, same return value (the selected rows as a DataFrame with the reference's
leading ``index`` column, ``<name>_selected_data.csv`` if 'store processed .csv file' is set), same
"log and return None" failures.  All the arithmetic -- per-track medians, quantiles, distance
outliers, the recursive splitting of tracks, the means of the surviving segments -- runs in
``ysmr_select_tracks`` (csrc/select.hip); this module only moves the table to the device and slices
the DataFrame with the row numbers that come back.
"""
from __future__ import annotations

import ctypes
import logging
import os

import numpy as np

from . import _lib
from .helper_file import create_results_folder, get_configs, get_data, save_df_to_csv

__all__ = ["select_tracks", "select_params", "select_rows"]

_COLUMNS = ["TRACK_ID", "POSITION_T", "POSITION_X", "POSITION_Y", "WIDTH", "HEIGHT", "DEGREES_ANGLE"]


def select_params(settings, fps, frame_height, frame_width) -> _lib.SelectParams:
    """``struct ysmr_select_params`` from a settings dict (track_eval.py:574-583 for the two lengths)."""
    p = _lib.SelectParams()
    p.area_lo = settings["extreme area outliers lower end in px*px"]
    p.area_hi = settings["extreme area outliers upper end in px*px"]
    p.area_factor = settings["exclude measurement when above x times average area"] or 0.0
    p.q_area = settings["percent quantiles excluded area"]
    p.motility_stop_fraction = settings["stop excluding motility outliers if total count above percent"]
    p.max_empty_ratio = settings["maximal empty frames in %"]
    p.ratio_min = settings["average width/height ratio min."]
    p.ratio_max = settings["average width/height ratio max."]
    p.edge_fraction = settings["percent of screen edges to exclude"]
    p.min_length_frames = int(round(fps, 0) * settings["minimal length in seconds"])
    p.limit_frames = int(round(fps, 0) * settings["limit track length to x seconds"])
    p.limit_exact = int(bool(settings["limit track length exactly"]))
    p.omit_motility = int(bool(settings["try to omit motility outliers"]))
    p.max_holes = int(settings["maximal consecutive holes"])
    p.max_recursion = int(settings["maximal recursion depth"])
    p.frame_height, p.frame_width = int(frame_height), int(frame_width)
    return p


def select_rows(df, params: _lib.SelectParams, device="cuda:0"):
    """Run ``ysmr_select_tracks`` on the six columns the selection reads.
    Returns (rows of ``df``, their index in the cleaned table, summary)."""
    import torch
    n = len(df)
    summary = _lib.SelectSummary()
    L = _lib.lib()
    dev = torch.device(device)
    with torch.cuda.device(dev):
        def up(name, dtype):
            return torch.from_numpy(np.ascontiguousarray(df[name].to_numpy(), dtype=dtype).view(
                np.int32 if dtype == np.uint32 else dtype)).to(dev)
        cols = [up("TRACK_ID", np.uint32), up("POSITION_T", np.uint32), up("POSITION_X", np.float64),
                up("POSITION_Y", np.float64), up("WIDTH", np.float64), up("HEIGHT", np.float64)]
        ws_bytes = L.ysmr_select_workspace_bytes(n, params.max_recursion)
        ws = torch.empty(max(ws_bytes, 256), dtype=torch.uint8, device=dev)
        sel_row = torch.empty(max(n, 1), dtype=torch.int64, device=dev)
        sel_index = torch.empty(max(n, 1), dtype=torch.int64, device=dev)
        rc = L.ysmr_select_tracks(_lib.stream_ptr(dev), n, *[c.data_ptr() for c in cols], ctypes.byref(params),
                                  ws.data_ptr(), ws.numel(), sel_row.data_ptr(), sel_index.data_ptr(),
                                  ctypes.byref(summary))
        _lib.check(rc, "ysmr_select_tracks")
        k = int(summary.rows_selected)
        return sel_row[:k].cpu().numpy(), sel_index[:k].cpu().numpy(), summary


def select_tracks(path_to_file=None, df=None, results_directory=None, fps=None, frame_height=None, frame_width=None,
                  settings=None, device="cuda:0", **_):
    """Selection of good tracks from a ``*_list.csv`` or its DataFrame (track_eval.py:536-843)."""
    import pandas as pd
    logger = logging.getLogger("ysmr").getChild(__name__)
    settings = get_configs(settings)
    if settings is None:
        logger.critical("No settings provided / could not get settings for start_it_up().")
        return None
    if path_to_file is None:
        # (the reference reads settings['path to test .csv'] here, a key its own get_configs never sets)
        logger.critical("select_tracks needs path_to_file (it names the result files), got None")
        return None
    if results_directory is None:
        results_directory = create_results_folder(path_to_file)
    file_name = os.path.splitext(os.path.basename(path_to_file))[0]
    if fps is None or fps <= 0 or settings["force tracking.ini fps settings"]:
        fps = settings["frames per second"]
    if frame_width is None or frame_height is None:
        logger.debug("Retrieving frame width/height from tracking.ini.")
        frame_width, frame_height = settings["frame width"], settings["frame height"]
    lo_key, hi_key = "extreme area outliers lower end in px*px", "extreme area outliers upper end in px*px"
    # what must hold before any device work; the messages are the reference's (track_eval.py:573-611)
    preconditions = (
        (fps > 0, "fps value is negative or zero; cannot continue."),
        (settings[lo_key] < settings[hi_key],
         "Minimal area exclusion in px^2 larger or equal to maximum; will not be able to find tracks. "
         "Please update tracking.ini. {}: {}, {}: {}".format(lo_key, settings[lo_key], hi_key, settings[hi_key])),
        (frame_height > 0 and frame_width > 0,
         "Frame width or frame height 0 or negative; cannot continue. Width: {}, height: {}".format(frame_width, frame_height)),
        (settings["pixel per micrometre"] > 0,
         "'pixel per micrometre' setting in tracking.ini 0 or negative. Cannot continue. Value: {}".format(
             settings["pixel per micrometre"])),
    )
    for holds, complaint in preconditions:
        if not holds:
            logger.critical(complaint)
            return None
    if not isinstance(df, pd.DataFrame):
        df = get_data(path_to_file)
    if df is None:
        logger.critical("Error reading data frame from file {}".format(path_to_file))
        return None
    params = select_params(settings, fps, frame_height, frame_width)
    try:
        rows, index, s = select_rows(df, params, device=device)
    except (_lib.YsmrLibraryError, RuntimeError) as exc:
        logger.critical("Device path failed for file {}: {}".format(path_to_file, exc))
        return None
    if s.status == _lib.SELECT_TOO_SHORT:
        logger.critical("File is empty/of insufficient length before initial clean-up. "
                        "Minimal size (frames): {}, length: {}, path: {}".format(params.min_length_frames, len(df), path_to_file))
        return None
    if s.status == _lib.SELECT_TOO_SHORT_CLEANED:
        logger.warning("File is empty/of insufficient length after initial clean-up. "
                       "Minimal size: {}, length: {}, path: {}".format(params.min_length_frames, s.rows_after, path_to_file))
        return None
    logger.info("Tracks before initial cleanup: {}, after: {}, loss: {:.4%}, "
                "data frame entries before: {}, after: {}, loss: {:.4%}".format(
                    s.tracks_before, s.tracks_after, (s.tracks_before - s.tracks_after) / s.tracks_before,
                    s.rows_before, s.rows_after, (s.rows_before - s.rows_after) / s.rows_before))
    if settings["percent quantiles excluded area"] > 0:
        logger.info("Area quartiles: 10%: {:.2f}, 90%: {:.2f}".format(s.area_lo, s.area_hi))
    if settings["try to omit motility outliers"]:
        share = s.dist_outliers / s.rows_after
        logger.info("25/75 % Distance quartiles: {:.3f}, {:.3f} upper outliers: {:.3f} counts: {}, of all entries: {:.4%}".format(
            s.q1_dist, s.q3_dist, s.dist_fence, s.dist_outliers, share))
        if not s.outliers_used:
            logger.warning("Motility outliers more than {:.2%} of all data points ({:.2%}); recommend to "
                           "re-analyse file with outlier removal changed if upper quartile is especially low"
                           "(Quartile: {:.3f})".format(
                               settings["stop excluding motility outliers if total count above percent"], share, s.q3_dist))
            logger.info("Distance outlier exclusion switched off due to too many outliers")
    kicks = list(s.kick_reasons)
    logger.info("All tracks before fine selection: {}, left over: {}, difference: {}".format(
        s.tracks_after, s.good_tracks, s.tracks_after - s.good_tracks))
    text = ("Total: {9}; size < 600: {8}; holes > 6: {7}; distance outlier: {6}; duration 5% over size: {5}; "
            "area out of bounds: {4}; ratio wrong: {3}; average x/y not within bounds: {2}; "
            "min/max xy not within screen: {1}; passed: {0}").format(*kicks, sum(kicks))
    if kicks[0] < 1000 and kicks[0] / sum(kicks) < 0.3:
        logger.warning("Low amount of accepted tracks")
        logger.warning(text)
    else:
        logger.info(text)
    if s.status == _lib.SELECT_NONE:
        logger.warning("File {} has no acceptable tracks.".format(path_to_file))
        return None
    out = df.iloc[rows][_COLUMNS].copy()
    out.insert(0, "index", index)            # the reference's reset_index(inplace=True) keeps the old index as a column
    out.reset_index(drop=True, inplace=True)
    if settings["store processed .csv file"]:
        save_df_to_csv(df=out, save_path=os.path.join(results_directory, file_name) + "_selected_data.csv")
    return out
