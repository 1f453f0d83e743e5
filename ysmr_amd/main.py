"""``ysmr()`` / ``analyse()`` entry points with the reference's signatures (ysmr/main.py:32, 175).

``analyse`` runs detect-and-link (``track_bacteria``), ``select_tracks`` and the statistics of
``evaluate_tracks`` on the device; plots, ``annotate_video`` and the xlsx collation are presentation, not part
of the HIP path, and are skipped with a log message.  It returns the last stage's result (``return_df=True``:
a DataFrame, or evaluate_tracks' ``(df, df_stats)``) or ``True``; ``None`` signals an error, as upstream.

Independent videos are embarrassingly parallel (the reference runs one process per path,
main.py:281-288): ``ysmr(..., multiprocess=True)`` shards the paths one process per GPU, no
collective involved.
"""
from __future__ import annotations

import logging
import os
from datetime import datetime

from .helper_file import create_results_folder, get_configs, get_loggers, metadata_file
from .evaluate import evaluate_tracks
from .select import select_tracks
from .track_eval import track_bacteria

__all__ = ["analyse", "ysmr"]

_EVALUATE_KEYS = ("store generated statistical .csv file",
                 "store final analysed .csv file", "save large plots", "save rose plot", "save time violin plot",
                 "save acr violin plot", "save length violin plot", "save turning point violin plot",
                 "save speed violin plot", "save angle distribution plot / bins", "collate results csv to xlsx",
                 "save video")
_OFFLINE_KEYS = ("store processed .csv file",) + _EVALUATE_KEYS


def analyse(path, settings=None, result_folder=None, return_df=False, device="cuda:0", batch=None, max_det=None,
            capacity=None, **kwargs):
    """Detect-and-link on one video, then ``select_tracks`` on its table (main.py:32-172); a
    ``*_list.csv`` of an earlier run goes straight to the selection.  ``kwargs`` go to the ``_meta.json``
    side file together with fps and frame size (main.py:99-108).  ``device``, ``batch``, ``max_det`` and
    ``capacity`` are passed on to ``track_bacteria``.  Returns the last stage's DataFrame (``return_df``) or
    True; None after any failure, including "no acceptable tracks" and -- as upstream -- a .csv input for
    which the settings ask for no further stage."""
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
    plots_eval = any(settings.get(k) for k in _EVALUATE_KEYS)
    produced = {"csv": None}

    def stages():
        """The body of the reference's ``while True: ... break`` (main.py:82-156): the value of the last
        stage that ran, None as soon as one fails."""
        value = None
        df, fps, f_height, f_width = None, None, None, None
        if any(tag in path for tag in ("_analysed.csv", "_statistics.csv", "_annotated_output.")):
            logger.warning("File already evaluated. File: {}".format(path))
            return None
        if ".csv" not in path:   # "as long as it's not a .csv, it should be a video" (main.py:88-97)
            result = track_bacteria(video_path=path, settings=settings, result_folder=result_folder, device=device,
                                    batch=batch, max_det=max_det, capacity=capacity)
            if result is None:
                logger.warning("Error during video analysis of file {}.".format(path))
                return None
            df, fps, f_height, f_width, produced["csv"] = result
            value = df
        # a *_list.csv of an earlier run: fps and frame size come from its _meta.json
        meta = metadata_file(path=os.path.join(result_folder, os.path.basename(path)), additional_search_paths=path,
                             verbose=settings["verbose"], fps=fps, frame_height=f_height, frame_width=f_width, **kwargs)
        if "selected_data.csv" not in path and (plots_eval or settings["store processed .csv file"]):
            df = select_tracks(path_to_file=path, df=df, results_directory=result_folder, settings=settings, device=device,
                               **{k: meta.get(k) for k in ("fps", "frame_height", "frame_width")})
            if df is None:
                logger.warning("Error during video analysis of file {}.".format(path))
                return None
            value = df
        if plots_eval:     # statistics of the selected tracks (main.py:131-139); None after an error, as upstream
            value = evaluate_tracks(path_to_file=path, results_directory=result_folder, df=df, settings=settings,
                                    device=device, fps=meta.get("fps"))
            if settings["save video"]:
                logger.warning("'save video' is enabled: annotating videos is not part of the HIP path")
        elif "selected_data.csv" in path:
            logger.warning("No evaluation set to True in settings. Did not evaluate {}".format(path))
        return value

    value = stages()
    if settings["delete .csv file after analysis"] and produced["csv"]:   # (also after a failed stage, main.py:156-163)
        try:
            os.remove(produced["csv"])
        except OSError:
            pass
    logger.info("{} process. PID: {}, elapsed time: {}".format("Finished with" if value is not None else "Error during",
                                                                os.getpid(), datetime.now() - t0))
    if value is None:
        return None
    return value if return_df else True


def _visible(device):
    """A worker may see fewer devices than the process that dealt the jobs (a scheduler's cgroup, a HIP_VISIBLE_DEVICES
    this library did not set): fold the ordinal instead of failing the video -- and say so, because two GPUs' videos then
    share one device.  (The worker processes ``ysmr`` starts itself see exactly one GPU and get their jobs renamed to
    cuda:0 before they arrive here.)"""
    import torch
    try:        # the HIP runtime's own count (device_count() may answer from the management library, which lists
        seen = int(torch._C._cuda_getDeviceCount())      # every GPU of the host whatever this process may use)
    except (AttributeError, RuntimeError):
        seen = torch.cuda.device_count()
    index = torch.device(device).index or 0
    if seen and index >= seen:
        folded = "cuda:{}".format(index % seen)
        logging.getLogger("ysmr").getChild(__name__).warning(
            "a video dealt to {} runs on {}: this process sees {} device(s)".format(device, folded, seen))
        return folded
    return device


def _worker(args):
    """One video on the GPU it was dealt to.  The device is made current here: a fresh worker's current
    device is cuda:0 whatever its job says (the entry points below it select their device themselves as
    well; this covers allocations made in between)."""
    path, settings, result_folder, device = args[:4]
    from . import _lib
    device = _visible(device)
    with _lib.on(device):
        return path, analyse(path, settings=settings, result_folder=result_folder, device=device)


def _gpu_worker(args):
    """All videos dealt to one GPU: ``streams`` of them at a time, each on a thread with its own HIP
    stream (two streams inside one process overlap far better than two processes on one device)."""
    jobs, streams = args[:2]
    own_process = len(args) > 2 and args[2] is not None
    if own_process:
        # A worker process of its own GPU: it must SEE only that GPU -- set before anything here initialises HIP -- so that
        # no stray context, allocation or default-device call lands on another worker's device, and it runs on the CPUs
        # of that GPU's NUMA node.
        from . import dist
        usable = dist.usable_gpus()
        if args[2].isdigit() and usable and int(args[2]) >= usable:
            pass    # the parent counted GPUs this process cannot open (a device cgroup): _visible() folds and says so
        else:
            os.environ["HIP_VISIBLE_DEVICES"] = args[2]
            os.environ.pop("CUDA_VISIBLE_DEVICES", None)     # (the parent's restriction is folded into args[2]: dist.physical_device)
            jobs = [(j[0], j[1], j[2], "cuda:0") + tuple(j[4:]) for j in jobs]
        pinned = dist.pin_to_gpu(0)
        logging.getLogger("ysmr").getChild(__name__).debug(
            "worker {} for GPU {}: cpus {}".format(os.getpid(), args[2], sorted(pinned) if pinned else "unpinned"))
    if streams <= 1 or len(jobs) <= 1:
        return [_worker(j) for j in jobs]
    from concurrent.futures import ThreadPoolExecutor

    import torch

    from . import _lib
    _lib.lib()                                     # load the library and configure logging once, before the threads start
    st = jobs[0][1]
    get_loggers(log_level=st["log_level"], logfile_name=st["log file path"],
                short_stream_output=st["shorten displayed logging output"],
                short_file_output=st["shorten logfile logging output"], log_to_file=st["log to file"], settings=st)

    def run(job):
        device = torch.device(_visible(job[3]))
        with torch.cuda.device(device), torch.cuda.stream(torch.cuda.Stream(device=device)):
            return _worker(job)
    with ThreadPoolExecutor(max_workers=streams, thread_name_prefix="ysmr-stream") as pool:
        return list(pool.map(run, jobs))


def ysmr(paths=None, settings=None, result_folder=None, multiprocess=False, streams_per_gpu=2):
    """Analyse one or several videos (main.py:175-331); returns ``[(path, result), ...]``.

    Interactive settings (``user input``, ``select files``) need a desktop session and are
    rejected; pass paths explicitly.  With ``multiprocess=True`` the paths are dealt round-robin to
    the visible GPUs, one worker process per GPU (main.py:283 starts one per file), and every worker
    runs ``streams_per_gpu`` videos at a time on threads with their own HIP streams: one video cannot
    fill a GPU -- its frames are linked one after the other -- so two or three streams on the same device
    deliver about 1.5 x the frames/s of one.
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
    if multiprocess and len(jobs) > 1:
        streams = max(1, int(streams_per_gpu))
        from . import dist
        per_gpu = [([j for j in jobs if j[3] == "cuda:{}".format(g)], streams, g) for g in range(n_gpu)]
        per_gpu = [a for a in per_gpu if a[0]]
        if len(per_gpu) > 1 and torch.cuda.is_initialized():
            # This process already holds a GPU context (a notebook, a test runner, an earlier call): starting
            # worker processes from it would fork/exec out of a GPU-initialised parent.  Keep everything
            # in-process instead: one thread per GPU, each running its videos on that GPU's streams.
            logger.warning("multiprocess=True, but this process has already initialised the GPU: "
                           "running the {} per-GPU workers as threads of this process".format(len(per_gpu)))
            from concurrent.futures import ThreadPoolExecutor
            with ThreadPoolExecutor(max_workers=len(per_gpu), thread_name_prefix="ysmr-gpu") as pool:
                parts = list(pool.map(_gpu_worker, [(a[0], a[1], None) for a in per_gpu]))
        elif len(per_gpu) > 1:                     # one worker process per GPU, as many as there are GPUs with work
            import multiprocessing as mp           # (started before anything here has touched the GPU)
            ctx = mp.get_context("spawn")
            with ctx.Pool(processes=len(per_gpu), maxtasksperchild=1) as pool:
                parts = pool.map(_gpu_worker, [(a[0], a[1], dist.physical_device(a[2])) for a in per_gpu], chunksize=1)
        else:
            parts = [_gpu_worker((per_gpu[0][0], per_gpu[0][1], None))]
        # (a path given twice is analysed twice, as upstream's pool would: results go back by position, not by path)
        order = [i for a in per_gpu for i, j in enumerate(jobs) if j in a[0]]
        by_index = dict(zip(order, (r for part in parts for _, r in part)))
        results = [(j[0], by_index.get(i)) for i, j in enumerate(jobs)]
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
