"""``track_bacteria`` -- the per-frame detect-and-link loop of YSMR on the GPU.

Same contract as ``ysmr.track_eval.track_bacteria`` (ysmr/track_eval.py:38-405): given a video
path and the tracking.ini settings it writes ``<name>_list.csv`` and returns
``(DataFrame, fps, frame_height, frame_width, csv_path)``, or ``None`` after logging the reason.
The loop body (track_eval.py:156-366) is replaced by batched device work:

    frames (host) --H2D--> ysmr_detect_batch  (threshold, hysteresis, components, minAreaRect)
                           ysmr_tracker_run   (claims, lifecycle, GSFF, rows; state stays in HBM)
                     <--D2H-- rows, every `list save length interval` rows

Detection of batch b+1 is issued on a second HIP stream while batch b is being linked.
"""
from __future__ import annotations

import logging
import os
import threading
import gc
import time

import numpy as np
import torch

from . import _lib
from .detect import Detector, MeanGrayState, mean_gray_params, threshold_params
from .frames import DeviceFrameFeed, open_video
from .helper_file import (COLOR_BGR2GRAY, RowStream, create_results_folder, get_configs, get_loggers, rows_to_csv_bytes, rows_to_csv_file, rows_to_csv_file_and_dataframe, rows_device_to_csv_file_and_dataframe,
                          rows_to_dataframe, save_list, wait_for_removals)
from .tracker import DeviceTracker, rows_to_numpy, sort_rows

__all__ = ["track_bacteria", "TrackingPipeline", "select_tracks", "evaluate_tracks"]

#: where the wall time of the last _device_pass went, in seconds since it began (diagnostics: scripts/e2e_profile.py)
LAST_PASS_MARKS = {}
#: which device buffers and which link path the last _device_pass ran with (the re-run of a video too dense for its first attempt)
LAST_PIPELINE_FACTS = {}

#: most rows kept on the device for one video (40 B each); longer tables are moved to the host in between
ROW_BUDGET_MAX = 32 << 20


def __getattr__(name):   # select_tracks / evaluate_tracks live in track_eval upstream (track_eval.py:536, 846)
    if name == "select_tracks":
        from .select import select_tracks
        return select_tracks
    if name == "evaluate_tracks":
        from .evaluate import evaluate_tracks
        return evaluate_tracks
    raise AttributeError(name)


def _on_own_device(method):
    """Run a TrackingPipeline method with the pipeline's GPU as the current device (the caller's current
    device is cuda:0 in a fresh worker, whatever GPU its video was dealt to)."""
    import functools

    @functools.wraps(method)
    def wrapped(self, *args, **kwargs):
        with _lib.on(self.device):
            return method(self, *args, **kwargs)
    return wrapped


class DeviceRows:
    """A video's rows, ordered, still on the device (``TrackingPipeline.take_rows_device``)."""

    def __init__(self, rows_u8, count):
        self.rows_u8, self.count = rows_u8, int(count)

    def __len__(self):
        return self.count

    def to_numpy(self):
        return rows_to_numpy(self.rows_u8, self.count)


class TrackingPipeline:
    """Device-resident detect+link over consecutive batches of one video stream: ``detect_async`` issues a batch's detection
    on a side stream (two detectors ping-pong), ``link`` its 64 link launches on the caller's stream, ``take_rows`` hands
    the rows over (ordered on the device with ``sort=True``).  Rows accumulate in ``self.rows`` (``rows_per_flush`` of
    them); ``track_bacteria`` empties the buffer when it fills and -- with the optional settings key 'hip persist rows' --
    appends every full buffer to ``<name>_list.csv`` as the reference does, so that an interrupted run keeps what was
    tracked.  ``link=False``: detection only (no tracker neighbour: the matrix-pipe threshold kernel is used)."""

    def __init__(self, height, width, fps, settings, batch=64, max_det=2048, capacity=768, device="cuda:0",
                 rows_per_flush=None, link=True):
        self.device = torch.device(device)
        self.B = int(batch)
        offset = settings["threshold offset for detection"]
        mean_state = None
        if settings["adaptive double threshold"] < 0:
            # mean-gray branch (track_eval.py:219-253): the moving average runs through the video, so
            # both detectors share its state (their launches are ordered on the side stream)
            params = mean_gray_params(settings["white bacteria on dark background"], offset, fps)
            mean_state = MeanGrayState(params.window, self.device)
        else:
            params = threshold_params(settings["white bacteria on dark background"], offset,
                                      settings["adaptive double threshold"])
        # two detectors: batch b+1 is detected (stream 1) while batch b is linked (stream 0)
        # optional settings key 'opencv version' ('4.5.0', '4.10.0', '3.4.18', ...): which release the BGR2GRAY
        # coefficients and the minAreaRect angle convention follow (include/ysmr_hip.h: cv_flavour)
        # Which threshold kernel: detection of batch b+1 runs BESIDE the link of batch b, and there the float32-chain strip
        # kernel is the better neighbour -- its resident grid leaves a wave slot, registers and LDS on every compute unit
        # for the link's chain of 10 us kernels, while the matrix-pipe kernel takes whole units (160 KB of LDS, every
        # register) and the link waits for them: 80.7 k against 85.7 k frames/s end to end, and giving it the chip to itself
        # between two batches' link chains costs more (77.8 k) than its 15 % buy (profiles/r03_threshold_kernels_in_the_pipeline.log;
        # at 4K 14.8 k against 15.2 k).  Detection without a link has no neighbour and takes the matrix-pipe kernel.
        self.trk = DeviceTracker(max_disappeared=fps, fps=fps, n_min=settings["minimum horizon size"],
                                 n_max=settings["maximum horizon size"], n_f=settings["number of LSFFs"],
                                 use_gsff=not settings["disable gsff"], capacity=capacity, max_det=max_det,
                                 device=self.device)
        # (a handle that links a whole batch with ONE launch -- one workgroup on one compute unit -- is nobody's neighbour:
        # detection then takes the matrix-pipe kernel and its full resident grids; beside the per-frame kernels, one-launch
        # or split (4K), it keeps round 3's choice: 21.0 k against 22.3 k frames/s at 4K with the matrix-pipe kernel)
        # (... except beside the SPLIT link of large tables, where the matrix-pipe kernel on half the compute units beats both:
        # 24.3 k, profiles/r04_4k_thr_grid.log)
        beside_split_link = bool(link) and not self.trk.batched and not self.trk.fused
        beside_fused_link = bool(link) and not self.trk.batched and not beside_split_link
        #: the threshold kernel is issued on the LINK stream, between two batches' link chains, where it has the chip to
        #: itself (the labelling chain still runs beside the link, on the side stream)
        self.exclusive_threshold = False
        mode = os.environ.get("YSMR_THRESHOLD_MODE")   # (measurement aid: scripts/thr_kernels_in_pipeline.sh)
        if mode in ("beside-strip", "beside-mfma", "exclusive-mfma", "exclusive-strip"):
            beside_fused_link = mode.endswith("strip")
            self.exclusive_threshold = mode.startswith("exclusive")
        self.det = [Detector(self.B, height, width, max_det=max_det, params=params, device=self.device,
                             mean_state=mean_state, cv_flavour=settings.get("opencv version"),
                             threshold_variant=1 if beside_fused_link else 0,
                             beside_batch_link=bool(link) and self.trk.batched, beside_split_link=beside_split_link)
                    for _ in range(2)]
        self.capacity = int(capacity)
        self._link = bool(link)
        n_rows = self.B * self.capacity if rows_per_flush is None else int(rows_per_flush)
        self.rows = torch.empty(n_rows * _lib.ROW_DTYPE.itemsize, dtype=torch.uint8, device=self.device)
        self.row_count = torch.zeros(1, dtype=torch.int64, device=self.device)
        self.side = torch.cuda.Stream(device=self.device)
        self._done = [None, None]   # event: detector i's outputs consumed by the tracker
        # (events are made once and recorded again every batch: creating one costs the host tens of microseconds, and the
        # host has little slack next to a link that wants a launch every ~11 us)
        self._ev = [{k: torch.cuda.Event() for k in ("ready", "done", "thresholded")} for _ in range(2)]
        self._k = 0

    @_on_own_device
    def detect_async(self, frames_dev, threshold_events=None, chain_events=None, events=None, frames_ready=None):
        """Issue detection of one batch on the side stream; returns (slot, result, ready_event).
        ``frames_ready``: what the side stream has to wait for before it reads ``frames_dev`` -- None: everything issued
        on the caller's stream so far (the frames were produced or uploaded there); an event: that event only; False:
        nothing, the frames have been resident since before the caller's stream was given its pending work (a clip in
        HBM).  The blanket wait also holds the batch back behind every link launch already issued, which drains the
        pipeline wherever one clip ends and the next begins.
        ``threshold_events``: list that receives a (start, stop) HIP event pair that the threshold kernel's own dispatch
        sets to its start and end on the device (bench.py's roofline measurement; ``Detector.threshold(timing=...)``);
        ``chain_events``: the same for the labelling / geometry chain behind it; ``events``: five ready-made timing
        events to use for these records (the first three here, the last two in ``link``) instead of new ones."""
        slot = self._k & 1
        self._k += 1
        cur = torch.cuda.current_stream(self.device)
        det = self.det[slot]
        thresholded = None
        if self.exclusive_threshold:
            e0, e1 = (events[0], events[1]) if events else (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            self._timed_threshold(det, frames_dev, cur, e0, e1, threshold_events is not None, bool(events))
            if threshold_events is not None:
                threshold_events.append((e0, e1, frames_dev.shape[0]))
            thresholded = self._ev[slot]["thresholded"]
            thresholded.record(cur)
        with torch.cuda.stream(self.side):
            if frames_ready is None:
                self.side.wait_stream(cur)      # the frames were produced/uploaded on the caller's stream
            elif frames_ready is not False:
                self.side.wait_event(frames_ready)
            if self._done[slot] is not None:
                self.side.wait_event(self._done[slot])
            if thresholded is not None:
                res = det.components(frames_dev.shape[0])
                if chain_events is not None:
                    e2 = events[2] if events else torch.cuda.Event(enable_timing=True)
                    e2.record(self.side)
                    chain_events.append((e1, e2, frames_dev.shape[0]))
            elif threshold_events is None:
                res = det.detect(frames_dev)
            else:
                e0, e1 = (events[0], events[1]) if events else (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                self._timed_threshold(det, frames_dev, self.side, e0, e1, True, bool(events))   # (the detector's own choice of kernel)
                threshold_events.append((e0, e1, frames_dev.shape[0]))
                res = det.components(frames_dev.shape[0])
                if chain_events is not None:
                    e2 = events[2] if events else torch.cuda.Event(enable_timing=True)
                    e2.record(self.side)
                    chain_events.append((e1, e2, frames_dev.shape[0]))
            if self._link and self.trk.batched:      # the link's binning of these detections, off the link stream
                self.trk.prepare(res.det, res.det_count, slot)
            ready = self._ev[slot]["ready"]
            ready.record(self.side)
        return slot, res, ready

    @staticmethod
    def _timed_threshold(det, frames_dev, stream, e0, e1, timed, events_exist):
        """The batch's threshold launch with its duration between e0 and e1: set by the kernel's own dispatch (one
        kernel), or recorded on the stream around the call (the mean-gray branch: three kernels)."""
        if not timed:
            det.threshold(frames_dev)
        elif det.mean_state is not None:
            e0.record(stream)
            det.threshold(frames_dev)
            e1.record(stream)
        else:
            if not events_exist:
                e0.record(stream); e1.record(stream)     # (creates them; the dispatch sets them again)
            det.threshold(frames_dev, timing=(e0, e1))

    @_on_own_device
    def reset(self):
        self.trk.reset()
        self.row_count.zero_()

    @_on_own_device
    def link(self, slot, res, ready, first_frame, link_events=None, events=None):
        """Link one detected batch on the current stream; rows accumulate in self.rows.
        ``link_events``: list that receives a (start, stop, frames, host_seconds) record around the batch's
        launches -- HIP events on the link stream, and how long the host took to issue them."""
        cur = torch.cuda.current_stream(self.device)
        cur.wait_event(ready)
        n = int(res.det_count.shape[0])

        def run():
            self.trk.run(res.det, res.det_count, first_frame, self.rows, self.row_count)

        if link_events is None:
            run()
        else:
            e0, e1 = (events[3], events[4]) if events else (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            e0.record(cur)
            t0 = time.perf_counter()
            run()
            host = time.perf_counter() - t0
            e1.record(cur)
            link_events.append((e0, e1, n, host))
        done = self._ev[slot]["done"]
        done.record(cur)
        self._done[slot] = done
        return res

    @_on_own_device
    def take_rows(self, sort=False):
        """Synchronise, download the accumulated rows and reset the row buffer.  ``sort``: order them
        by (TRACK_ID, POSITION_T) on the device first (``sort_list``, helper_file.py:1538-1574)."""
        n = int(self.row_count.item())
        cap = self.rows.numel() // _lib.ROW_DTYPE.itemsize
        if n > cap:
            raise _lib.YsmrLibraryError(f"row buffer overflow: {n} rows > capacity {cap}")
        rows = rows_to_numpy(sort_rows(self.rows, n) if sort else self.rows, n)   # (a fresh host array)
        self.row_count.zero_()
        return rows

    @_on_own_device
    def take_rows_device(self):
        """Synchronise, order the accumulated rows by (TRACK_ID, POSITION_T) on the device (``sort_list``,
        helper_file.py:1538-1574), LEAVE them there and reset the row buffer: ``DeviceRows`` for ``_finish``, which prints them
        on the device too."""
        n = int(self.row_count.item())
        cap = self.rows.numel() // _lib.ROW_DTYPE.itemsize
        if n > cap:
            raise _lib.YsmrLibraryError(f"row buffer overflow: {n} rows > capacity {cap}")
        out = DeviceRows(sort_rows(self.rows, n), n)
        self.row_count.zero_()
        return out

    @_on_own_device
    def check(self, res):
        status = int(res.status.max().item())
        _, _, err = self.trk.info()
        if status or err:
            kind = _lib.YsmrCapacityError if (status & _lib.DET_OVERFLOW or err) else _lib.YsmrLibraryError
            raise kind(f"device path reported errors: detection status {status}, tracker {err} "
                       f"(max_det {self.det[0].max_det}, capacity {self.capacity})")


#: device-side limits when neither the call nor the settings dict name any; a video that exceeds them is
#: re-run with doubled limits (up to LIMIT_MAX), so they only decide how much HBM the first attempt takes
DEFAULT_BATCH, DEFAULT_MAX_DET, DEFAULT_CAPACITY, LIMIT_MAX = 256, 2048, 768, 32768
DEFAULT_BATCH_BYTES = 300e6


def auto_batch(height, width, channels=1):
    """Frames per batch for a video of this geometry when nobody names a number: about DEFAULT_BATCH_BYTES of frames, a
    multiple of 8 between 16 and DEFAULT_BATCH -- and, where that reaches the number of workgroups the threshold kernel
    runs on beside the one-launch batch link (``ysmr_threshold_workgroups``: 248), exactly that many: every workgroup then
    takes one whole frame and starts one item instead of two (the kernel is 5 % faster per frame)."""
    frame_bytes = max(1, int(height) * int(width) * (3 if channels == 3 else 1))
    batch = max(16, min(DEFAULT_BATCH, int(DEFAULT_BATCH_BYTES // frame_bytes) // 8 * 8))
    workgroups = int(_lib.lib().ysmr_threshold_workgroups(_lib.BESIDE_BATCH_LINK))
    return workgroups if batch >= workgroups >= 16 else batch


def track_bacteria(video_path, settings=None, result_folder=None, batch=None, max_det=None, capacity=None,
                   device="cuda:0"):
    """Detect and track bright (or dark) spots in a video; write ``<name>_list.csv``.

    Returns ``(DataFrame, fps, frame_height, frame_width, csv_path)`` or ``None`` (errors are
    logged on logger 'ysmr', never raised -- track_eval.py:50-77, 389-392, 402-404).

    ``batch`` (frames per detection launch), ``max_det`` (components per frame) and ``capacity`` (live
    tracks) size the device buffers; the reference has no such limits, so they may also be given as the
    optional settings keys 'hip frames per batch', 'hip max detections per frame', 'hip max tracks', and a
    video that overflows them is run again with both doubled (checked after the first batch and at the end).
    Optional settings key 'hip persist rows' (default False): keep at most 'list save length interval' rows on the
    device and append every full buffer to ``<name>_list.csv`` as the reference does (helper_file.py:1403-1478), so
    that an interrupted run leaves the rows tracked so far; the file is rewritten in order at the end either way.
    Optional settings key 'hip print rows on device' (default True): the ordered rows' csv text and DataFrame columns are worked
    out on the device (``ysmr_rows_format_device``); False: on the host's threads (``ysmr_rows_write_csv_columns``) -- the same
    bytes and bits either way.  'hip stream rows' (default False): the host's threads print the rows while the video runs.
    """
    logger = logging.getLogger("ysmr").getChild(__name__)
    settings = get_configs(settings)
    if settings is None:
        logger.critical("No settings provided / could not get settings for track_bacteria().")
        return None
    get_loggers(log_level=settings["log_level"], logfile_name=settings["log file path"],
                short_stream_output=settings["shorten displayed logging output"],
                short_file_output=settings["shorten logfile logging output"], log_to_file=settings["log to file"])
    if not os.path.isfile(video_path):
        logger.critical("File {} does not exist".format(video_path))
        return None
    for key, why in (("include luminosity in tracking calculation", "luminosity as a third tracking dimension"),
                     ("display video analysis", "interactive display")):
        if settings[key]:
            logger.critical("'{}' = True ({}) is not supported by the HIP path".format(key, why))
            return None
    if settings["color filter"] != COLOR_BGR2GRAY:
        logger.critical("Only 'color filter = COLOR_BGR2GRAY' is supported by the HIP path")
        return None
    try:
        video = open_video(video_path, default_fps=settings["frames per second"])
    except (OSError, ValueError) as exc:
        logger.exception("Cannot open file {} due to error: {}".format(video_path, exc))
        return None
    frame_count = video.frame_count
    if frame_count < settings["minimal frame count"]:
        logger.warning("File {} too short; file was skipped. Limit for 'minimal frame count': {}".format(
            video_path, settings["minimal frame count"]))
        return None
    fps_of_file = settings["frames per second"] if settings["force tracking.ini fps settings"] else video.fps
    if fps_of_file <= 0:
        logger.critical("User defined fps unacceptable: {}".format(fps_of_file))
        return None
    if not result_folder:
        result_folder = create_results_folder(video_path)
    logger.info("Starting with file {}".format(video_path))
    old_list, list_name = save_list(path=video_path, result_folder=result_folder, first_call=True,
                                    rename_old_list=settings["rename previous result .csv"])
    try:
        # The reference flips the sign of the offset IN the caller's dict for dark-on-bright videos
        # (track_eval.py:132), so a dict reused across files alternates; kept for drop-in parity.
        # threshold_params() applies the same sign change internally, from the value seen on entry.
        offset_on_entry = settings["threshold offset for detection"]
        if not settings["white bacteria on dark background"]:
            settings["threshold offset for detection"] = offset_on_entry * -1
        local = dict(settings)
        local["threshold offset for detection"] = offset_on_entry

        frame_height, frame_width = video.height, video.width
        # frames per batch when nobody names a number: ~256 MB of frames, between 16 and 256 frames (248 at 1228 x 922, 32 at 4K).
        # A batch's fixed costs -- one link launch, the detection kernels' starts, the reader's calls -- are paid per batch
        # (a 1920-frame 1228 x 922 file: 114-119 ms at 64 frames per batch, 99 ms at 256, scripts/e2e_batches.py); three pinned
        # staging buffers and two detectors' outputs of that many frames are what it costs in memory.
        batch = int(batch or settings.get("hip frames per batch") or auto_batch(frame_height, frame_width, video.channels))
        max_det = int(max_det or settings.get("hip max detections per frame") or DEFAULT_MAX_DET)
        capacity = int(capacity or settings.get("hip max tracks") or DEFAULT_CAPACITY)
        while True:
            outcome = _device_pass(video, video_path, frame_count, fps_of_file, local, batch, max_det, capacity, device,
                                   settings, logger, list_name if settings.get("hip persist rows") else None)
            if outcome[0] == "overflow" and 2 * max(max_det, capacity) <= LIMIT_MAX:
                max_det, capacity = 2 * max_det, max(2 * capacity, max_det)
                logger.warning("More objects than the device buffers hold in file {}: running it again with "
                               "max_det = {}, capacity = {}".format(video_path, max_det, capacity))
                continue
            if outcome[0] == "overflow":
                logger.critical("File {} holds more objects per frame than the largest device buffers ({}); "
                                "not analysed".format(video_path, LIMIT_MAX))
            break
        video.close()
        _, sorted_rows, frames_done, error_during_read, t_start, t_frames = outcome
        return _finish(video_path, settings, logger, sorted_rows, frames_done, frame_count, error_during_read, old_list,
                       list_name, fps_of_file, frame_height, frame_width, t_start, t_frames)
    finally:
        wait_for_removals()   # (a large previous list is unlinked beside the run: helper_file._remove_previous_list)


def usable_cpus():
    """CPUs this process may really use: its affinity mask, cut by the cgroup's quota (a box of this pool shows 256 CPUs and
    grants 16)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:
            quota, period = fh.read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as fq, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as fp:
                quota, period = int(fq.read()), int(fp.read())
            if quota > 0:
                n = min(n, max(1, quota // period))
        except (OSError, ValueError):
            pass
    return n


class _RowDrain:
    """The tail of a video, taken off its end (39 of a 1920-frame file's 97 ms came after the last frame, all
    serial): a thread that, batch by batch, waits for the batch's link launch, brings the rows it added to the device buffer
    to the host through pinned memory on a stream of its own, and hands them to a ``RowStream`` (``ysmr_rows_stream_*``),
    whose threads print them while later batches run.  A row's text does not depend on any other row; its place in the file
    -- by (TRACK_ID, POSITION_T) -- is worked out once, at the end."""

    def __init__(self, pipe, threads=None):
        import queue
        self.pipe = pipe
        self.rows_cap = pipe.rows.numel() // _lib.ROW_DTYPE.itemsize
        self.stream = RowStream(via_pandas=True, threads=threads or int(os.environ.get("YSMR_STREAM_THREADS", 0)) or max(2, min(12, usable_cpus() // 2)))
        self.copy_stream = torch.cuda.Stream(device=pipe.device)
        self.count_host = torch.zeros(1, dtype=torch.int64).pin_memory()
        self.staging = torch.empty(min(self.rows_cap, pipe.B * pipe.capacity) * _lib.ROW_DTYPE.itemsize, dtype=torch.uint8).pin_memory()
        self.taken = 0
        self.error = None
        self.q = queue.Queue()
        self.thread = threading.Thread(target=self._run, name="ysmr-row-drain", daemon=True)
        self.thread.start()

    def submit(self, linked_event):
        """``linked_event``: recorded on the link stream behind the batch's launch."""
        self.q.put(linked_event)

    def _run(self):
        size = _lib.ROW_DTYPE.itemsize
        with torch.cuda.device(self.pipe.device):
            while True:
                ev = self.q.get()
                try:
                    if ev is None:
                        return
                    if self.error is not None:
                        continue
                    ev.synchronize()
                    with torch.cuda.stream(self.copy_stream):
                        self.count_host.copy_(self.pipe.row_count, non_blocking=True)
                        self.copy_stream.synchronize()
                        n = int(self.count_host.item())
                        if n > self.rows_cap:
                            raise _lib.YsmrLibraryError(f"row buffer overflow: {n} rows > capacity {self.rows_cap}")
                        per = self.staging.numel() // size
                        while self.taken < n:
                            k = min(n - self.taken, per)
                            self.staging[:k * size].copy_(self.pipe.rows[self.taken * size:(self.taken + k) * size], non_blocking=True)
                            self.copy_stream.synchronize()
                            self.stream.push(self.staging.data_ptr(), k)
                            self.taken += k
                except Exception as exc:           # (kept for the caller's thread: finish() / wait_idle() raise it)
                    self.error = exc
                finally:
                    self.q.task_done()

    def wait_idle(self):
        """Every batch submitted so far is in the stream (the device buffer may be reused from its start)."""
        self.q.join()
        if self.error is not None:
            raise self.error

    def restart_buffer(self):
        self.taken = 0

    def finish(self):
        """-> the RowStream holding every row of the video (its threads may still be printing the last batch)."""
        self.q.put(None)
        self.thread.join()
        if self.error is not None:
            self.stream.close()
            raise self.error
        return self.stream

    def abort(self):
        self.q.put(None)
        self.thread.join()
        self.stream.close()


_GC_LOCK = threading.Lock()
_GC_HOLDERS = 0          # frame loops of this process that currently want the heap frozen
_GC_OURS = False         # the freeze in force is this module's (not the caller's own)


def _gc_hold():
    """The first frame loop of the process freezes the collector's view of the heap, unless the caller already has
    (``gc.get_freeze_count() != 0``: that freeze is theirs and stays theirs); later loops only count themselves in."""
    global _GC_HOLDERS, _GC_OURS
    with _GC_LOCK:
        if _GC_HOLDERS == 0:
            _GC_OURS = gc.get_freeze_count() == 0
            if _GC_OURS:
                gc.freeze()
        _GC_HOLDERS += 1


def _gc_release():
    """... and only the last one to leave thaws it (two stream threads per GPU worker overlap their passes)."""
    global _GC_HOLDERS, _GC_OURS
    with _GC_LOCK:
        _GC_HOLDERS -= 1
        if _GC_HOLDERS == 0 and _GC_OURS:
            gc.unfreeze()
            _GC_OURS = False


def _linked_event(device):
    """An event of its own behind a batch's link launch (the pipeline's per-slot events are recorded again two batches on)."""
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream(device))
    return ev


def _persist_chunk(list_name, rows, first):
    """'hip persist rows': the rows of one full device buffer, in the order they were tracked, appended to the list
    file -- what save_list (helper_file.py:1403-1478) does every 'list save length interval' rows."""
    with open(list_name, "wb" if first else "ab") as fh:
        fh.write(rows_to_csv_bytes(rows, header=first, via_pandas=False))


def _device_pass(video, video_path, frame_count, fps_of_file, local, batch, max_det, capacity, device, settings,
                 logger, persist_to=None):
    """The frame loop of one attempt.  Returns (verdict, sorted rows or None, frames done, error flag,
    start time, time when the last frame was linked); verdict 'overflow' asks for larger buffers."""
    frame_height, frame_width = video.height, video.width
    error_during_read = False
    frames_done = 0
    sorted_rows = None
    feed = None
    verdict = "done"
    froze = False
    drain = None
    t_start = t_frames = time.perf_counter()
    try:
        # Rows stay on the device for the whole video when they fit (capacity rows per frame is the
        # worst case; 32 M rows = 1.3 GB) and are ordered there at the end; otherwise full buffers are
        # moved to the host in between.  ('list save length interval' bounded the reference's Python
        # list, helper_file.py:171; here it only sets the smallest buffer.)
        row_budget = min(max(frame_count, 1) * capacity, ROW_BUDGET_MAX)
        if persist_to:      # as upstream: the rows leave for the file every 'list save length interval'
            row_budget = 0
        row_budget = max(row_budget, 2 * batch * capacity, int(settings["list save length interval"]))
        pipe = TrackingPipeline(frame_height, frame_width, fps_of_file, local, batch=batch, max_det=max_det,
                                capacity=capacity, device=device, rows_per_flush=row_budget)
        LAST_PIPELINE_FACTS.clear()
        LAST_PIPELINE_FACTS.update(capacity=int(capacity), max_det=int(max_det), batched=bool(pipe.trk.batched), fused=bool(pipe.trk.fused))
        chunks = []
        pending = None
        res = None
        checked_early = False
        row_capacity = pipe.rows.numel() // _lib.ROW_DTYPE.itemsize
        rows_upper = 0     # host-side bound on the rows in the device buffer (no sync per batch)
        LAST_PASS_MARKS.clear()
        LAST_PASS_MARKS["pipeline built"] = time.perf_counter() - t_start
        feed = DeviceFrameFeed(video, pipe.B, pipe.device)
        LAST_PASS_MARKS["feed built"] = time.perf_counter() - t_start
        # (rows leave for the host and are printed batch by batch, while later batches run -- unless they are to be appended
        # to the list file in the order they were tracked, 'hip persist rows', which is the older, serial path)
        # ('hip stream rows', default off: measured on a 16-CPU box the overlap LOSES -- 107-154 ms against the serial tail's 98 for a
        # 1920-frame file, profiles/r05_e2e_stream_threads.log: the loop's one host copy per byte of video and the printing of
        # 971 k rows are 0.43 core-seconds each, and side by side they only slow each other down)
        drain = _RowDrain(pipe) if (persist_to is None and settings.get("hip stream rows", False)) else None
        # A full pass of CPython's garbage collector walks every tracked object of the process (~40 ms with torch, numpy
        # and pandas loaded) and would stall the loop that keeps the GPU fed for as long as 50 batches take; what is alive
        # now is moved out of its sight for the duration of the loop (collections of the loop's own garbage stay on).
        # (a count under a lock: with two stream threads per GPU worker the first pass to finish must not thaw the heap
        # under the other's loop)
        _gc_hold()
        froze = True
        for dev, f0, n_read, feed_slot in feed:
            LAST_PASS_MARKS.setdefault("first batch on the device", time.perf_counter() - t_start)
            nxt = (pipe.detect_async(dev), f0, n_read)
            feed.release(feed_slot, nxt[0][2])   # the batch's frames are free once its detection has run
            if pending is not None:
                (slot, r, ready), p0, cnt = pending
                if rows_upper + cnt * pipe.capacity > row_capacity:
                    if res is not None:
                        pipe.check(res)
                    if drain is not None:       # everything linked so far is with the stream: the buffer starts over
                        drain.wait_idle()
                        pipe.row_count.zero_()
                        drain.restart_buffer()
                    else:
                        chunks.append(pipe.take_rows())
                        if persist_to:
                            _persist_chunk(persist_to, chunks[-1], len(chunks) == 1)
                    rows_upper = 0
                res = pipe.link(slot, r, ready, p0)
                if drain is not None:
                    drain.submit(_linked_event(pipe.device))
                rows_upper += cnt * pipe.capacity
                frames_done = p0 + cnt
                if not checked_early:           # a video too dense for the buffers is found out after its
                    pipe.check(res)             # first batch, not after its last (one sync per video)
                    checked_early = True
            pending = nxt
        if pending is not None:
            (slot, r, ready), p0, cnt = pending
            if rows_upper + cnt * pipe.capacity > row_capacity:
                if res is not None:
                    pipe.check(res)
                if drain is not None:
                    drain.wait_idle()
                    pipe.row_count.zero_()
                    drain.restart_buffer()
                else:
                    chunks.append(pipe.take_rows())
                    if persist_to:
                        _persist_chunk(persist_to, chunks[-1], len(chunks) == 1)
            res = pipe.link(slot, r, ready, p0)
            if drain is not None:
                drain.submit(_linked_event(pipe.device))
            frames_done = p0 + cnt
        LAST_PASS_MARKS["last batch issued"] = time.perf_counter() - t_start
        if froze:
            _gc_release()
            froze = False
        if res is not None:
            torch.cuda.synchronize(pipe.device)
            t_frames = time.perf_counter()
            LAST_PASS_MARKS["last batch linked"] = t_frames - t_start
            pipe.check(res)
            if drain is not None:
                sorted_rows = drain.finish()       # (a RowStream: _finish orders and writes it)
                drain = None
                LAST_PASS_MARKS["last rows with the stream"] = time.perf_counter() - t_start
            elif chunks:      # did not fit: gather on the host, order on the device in one go
                chunks.append(pipe.take_rows())
                everything = np.concatenate(chunks)
                on_dev = torch.from_numpy(everything.view(np.uint8)).to(pipe.device)
                sorted_rows = rows_to_numpy(sort_rows(on_dev, len(everything)), len(everything))
            elif settings.get("hip print rows on device", True):
                sorted_rows = pipe.take_rows_device()      # (_finish prints them there: ysmr_rows_format_device)
            else:
                sorted_rows = pipe.take_rows(sort=True)
        # the reference reads until cap.read() fails and accepts that only where the container said it would end, or
        # one frame earlier ("some file formats skip one frame", track_eval.py:170-174); anything else -- frames missing
        # OR frames beyond the reported count -- is its read error (:175-178)
        if frames_done not in (frame_count, frame_count - 1):
            logger.critical("Error during read with file {}".format(video_path))
            error_during_read = settings["stop evaluation on error"]
    except _lib.YsmrCapacityError as exc:
        logger.warning("{} (file {})".format(exc, video_path))
        verdict, sorted_rows, error_during_read = "overflow", None, True
    except (_lib.YsmrLibraryError, RuntimeError, ValueError) as exc:
        logger.critical("Device path failed for file {}: {}".format(video_path, exc))
        error_during_read = True
    finally:
        if froze:
            _gc_release()
        if feed is not None:
            feed.close()
        if drain is not None:       # (an attempt that ended early: its rows go nowhere)
            drain.abort()
    return verdict, sorted_rows, frames_done, error_during_read, t_start, t_frames


def _finish(video_path, settings, logger, sorted_rows, frames_done, frame_count, error_during_read, old_list, list_name,
            fps_of_file, frame_height, frame_width, t_start, t_frames):
    """Everything after the frame loop (track_eval.py:368-405): restore the old list after an error, write
    the ordered csv, build the DataFrame."""
    if old_list and error_during_read:
        try:
            os.remove(list_name)
            os.rename(old_list, list_name)
            logger.info("Restoring old list: {}".format(list_name))
        except OSError as exc:
            logger.error("Could not restore {}: {!r}".format(list_name, exc.args))
    stream = sorted_rows if isinstance(sorted_rows, RowStream) else None
    on_device = sorted_rows if isinstance(sorted_rows, DeviceRows) else None
    if sorted_rows is None or len(sorted_rows) == 0:
        if stream is not None:
            stream.close()
        logger.warning("Did not track any objects. File: {}".format(video_path))
        return None
    # track_eval.py:387-392 asks the LAST frame's tracker output for its last object id: a video whose final
    # frame has no live track (nothing seen for more than a second) "did not track any objects" upstream,
    # whatever the earlier frames held -- and the unsorted list stays on disk.  Same verdict here (the list
    # is written, ordered, before returning).
    # track_eval.py:393: sort_list(file_path=list_name, save_file=not settings['delete .csv ...']) re-reads
    # the csv with pandas, sorts it and rewrites it; the same DataFrame and the same bytes come
    # straight from the rows here (helper_file.rows_to_dataframe / rows_to_csv_bytes).
    t_rows = time.perf_counter()
    df_for_eval = None
    keep_file = not settings["delete .csv file after analysis"]   # (else analyse() removes the file anyway, main.py:156)
    if stream is not None:
        # the rows were printed while the video ran (_RowDrain): what is left is their order, the file, the columns
        n_rows_total = len(stream)
        try:
            try:
                _, df_for_eval = stream.finish(list_name if keep_file else None)
            except (OSError, _lib.YsmrLibraryError) as exc:
                if not keep_file:
                    raise
                logger.error("Could not write {}: {}".format(list_name, exc))
                _, df_for_eval = stream.finish(None)
        finally:
            stream.close()
        alive_at_end = df_for_eval["TRACK_ID"].to_numpy()[df_for_eval["POSITION_T"].to_numpy() == frames_done - 1]
    if on_device is not None:
        # Round 5: text and columns come from the device, where the ordered rows are (shortest digits, CPython's layout, pandas'
        # reading: csrc/fmt.h) -- the host copies them over and writes the file.  A table with a value the device form does not
        # print (none that a track produces) is brought over and takes the host path below.
        n_rows_total = len(on_device)
        made = None
        try:
            made = rows_device_to_csv_file_and_dataframe(on_device.rows_u8, n_rows_total, list_name if keep_file else None)
        except OSError as exc:
            if not keep_file:
                raise
            logger.error("Could not write {}: {}".format(list_name, exc))
            made = rows_device_to_csv_file_and_dataframe(on_device.rows_u8, n_rows_total, None)
        if made is None:
            sorted_rows = on_device.to_numpy()
            on_device = None
        else:
            df_for_eval = made[1]
            alive_at_end = df_for_eval["TRACK_ID"].to_numpy()[df_for_eval["POSITION_T"].to_numpy() == frames_done - 1]
    if stream is None and on_device is None:
        alive_at_end = sorted_rows["track_id"][sorted_rows["frame"] == frames_done - 1]
        n_rows_total = len(sorted_rows)
        # the csv and the DataFrame's columns come out of ONE native pass over the rows (the values the text is printed from are
        # the values pandas would read back from it: worked out once, by the formatting threads themselves)
        if keep_file:
            try:
                _, df_for_eval = rows_to_csv_file_and_dataframe(sorted_rows, list_name)
            except (OSError, _lib.YsmrLibraryError) as exc:
                logger.error("Could not write {}: {}".format(list_name, exc))
        if df_for_eval is None:
            df_for_eval = rows_to_dataframe(sorted_rows)
    last_id = int(alive_at_end.max()) if len(alive_at_end) else -1
    t_df = time.perf_counter()
    LAST_PASS_MARKS["csv and DataFrame done"] = t_df - t_start
    logger.debug("phases: frames {:.1f} ms, rows to host (sorted) {:.1f} ms, DataFrame {:.1f} ms, csv {:.1f} ms".format(
        (t_frames - t_start) * 1e3, (t_rows - t_frames) * 1e3, (t_df - t_rows) * 1e3, (time.perf_counter() - t_df) * 1e3))
    logger.info("objects: {}, frames: {} of {}, rows: {}, csv: {}".format(last_id + 1, frames_done, frame_count,
                                                                          n_rows_total, list_name))
    if len(alive_at_end) == 0:
        logger.warning("Did not track any objects. File: {}".format(video_path))
        return None
    if error_during_read:
        logger.critical("Error during read, stopping before evaluation. File: {}".format(video_path))
        return None
    return df_for_eval, fps_of_file, frame_height, frame_width, list_name
