"""``ysmr()`` / ``analyse()`` entry points with the reference's signatures (ysmr/main.py:32, 175).

Only the detect-and-link stage is implemented here (``track_bacteria``); the reference's offline
stages (``select_tracks``, ``evaluate_tracks``, ``annotate_video``, plots, xlsx collation) are out
of scope and are skipped with a log message.  ``analyse`` therefore returns the DataFrame of
``<name>_list.csv`` (``return_df=True``) or ``True``; ``None`` signals an error, as upstream.

Independent videos are embarrassingly parallel (the reference runs one process per path,
main.py:281-288): ``ysmr(..., multiprocess=True)`` shards the paths one process per GPU, no
collective involved.
"""
from __future__ import annotations

import logging
import os
from datetime import datetime

from .helper_file import create_results_folder, get_configs, get_loggers, metadata_file
from .track_eval import track_bacteria

__all__ = ["analyse", "ysmr"]

_OFFLINE_KEYS = ("store processed .csv file", "store generated statistical .csv file",
                 "store final analysed .csv file", "save large plots", "save rose plot", "save time violin plot",
                 "save acr violin plot", "save length violin plot", "save turning point violin plot",
                 "save speed violin plot", "save angle distribution plot / bins", "collate results csv to xlsx",
                 "save video")


def analyse(path, settings=None, result_folder=None, return_df=False, device="cuda:0", **kwargs):
    """Run the detect-and-link stage on one video (main.py:32-172).  ``kwargs`` go to the
    ``_meta.json`` side file together with fps and frame size (main.py:99-108)."""
    t0 = datetime.now()
    settings = get_configs(settings)
    if settings is None:
        return None
    get_loggers(log_level=settings["log_level"], logfile_name=settings["log file path"],
                short_stream_output=settings["shorten displayed logging output"],
                short_file_output=settings["shorten logfile logging output"], log_to_file=settings["log to file"],
                settings=settings)
    logger = logging.getLogger("ysmr").getChild(__name__)
    if result_folder is None:
        result_folder = create_results_folder(path)
    os.makedirs(result_folder, exist_ok=True)
    logger.debug("Starting process. PID: {} Result folder: {}".format(os.getpid(), result_folder))
    if any(tag in path for tag in ("_analysed.csv", "_statistics.csv", "_annotated_output.")):
        logger.warning("File already evaluated. File: {}".format(path))
        return None
    if ".csv" in path:
        logger.warning("{}: .csv inputs belong to the offline stages, which this package does not implement".format(path))
        return None
    result = track_bacteria(video_path=path, settings=settings, result_folder=result_folder, device=device)
    if result is None:
        logger.warning("Error during video analysis of file {}.".format(path))
        logger.info("Error during process. PID: {}, elapsed time: {}".format(os.getpid(), datetime.now() - t0))
        return None
    df, fps, f_height, f_width, csv_file = result
    metadata_file(path=os.path.join(result_folder, os.path.basename(path)), verbose=settings["verbose"], fps=fps,
                  frame_height=f_height, frame_width=f_width, **kwargs)
    if any(settings.get(k) for k in _OFFLINE_KEYS):
        logger.info("select_tracks / evaluate_tracks / plots are not part of the HIP path; stopping after "
                    "{}".format(csv_file))
    if settings["delete .csv file after analysis"] and csv_file:
        try:
            os.remove(csv_file)
        except OSError:
            pass
    logger.info("Finished with process. PID: {}, elapsed time: {}".format(os.getpid(), datetime.now() - t0))
    return df if return_df else True


def _worker(args):
    path, settings, result_folder, device = args
    return path, analyse(path, settings=settings, result_folder=result_folder, device=device)


def ysmr(paths=None, settings=None, result_folder=None, multiprocess=False):
    """Analyse one or several videos (main.py:175-331); returns ``[(path, result), ...]``.

    Interactive settings (``user input``, ``select files``) need a desktop session and are
    rejected; pass paths explicitly.  With ``multiprocess=True`` the paths are dealt round-robin to
    the visible GPUs, one worker process per GPU.
    """
    settings = get_configs(settings)
    if settings is None:
        print("Fatal error in retrieving tracking.ini")
        return None
    logger = get_loggers(log_level=settings["log_level"], logfile_name=settings["log file path"],
                         log_to_file=settings["log to file"], settings=settings).getChild(__name__)
    if isinstance(paths, (str, os.PathLike)):
        paths = [paths]
    if settings["debugging"] and not paths:
        paths = [settings["path to test video"]]
    if not paths:
        if settings["select files"]:
            logger.critical("No files selected ('select files' needs a file dialog; pass paths instead).")
            return None
        paths = [settings["path to test video"]]
    if settings["user input"]:
        logger.warning("'user input' = True ignored: no interactive confirmation in the HIP path")
    paths = [os.path.expanduser(p) for p in paths]
    logger.info("Total number of files: {}".format(len(paths)))
    if result_folder is None:
        result_folder = create_results_folder(paths[0])
    os.makedirs(result_folder, exist_ok=True)

    import torch
    n_gpu = max(torch.cuda.device_count(), 1)
    jobs = [(p, dict(settings), result_folder, "cuda:{}".format(i % n_gpu)) for i, p in enumerate(paths)]
    finished, failed = [], []
    if multiprocess and len(jobs) > 1 and n_gpu > 1:
        import torch.multiprocessing as mp
        ctx = mp.get_context("spawn")
        with ctx.Pool(processes=min(n_gpu, len(jobs)), maxtasksperchild=1) as pool:
            results = pool.map(_worker, jobs)
    else:
        results = [_worker(j) for j in jobs]
    for path, res in results:
        finished.append((path, res))
        if res is None:
            failed.append(path)
    if failed:
        logger.critical("Failed to analyse {} of {} file(s):".format(len(failed), len(paths)))
        for p in failed:
            logger.critical("{}".format(p))
    else:
        logger.info("Finished with all files.")
    return finished
