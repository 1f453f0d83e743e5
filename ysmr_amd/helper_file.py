"""Configuration surface and result-file helpers of the detect-and-link path.

Mirrors the parts of ``ysmr/helper_file.py`` that the hot path touches: the ``tracking.ini``
layout and the flat settings dict (``create_configs`` :143-316, ``get_configs`` :586-843), the
``*_list.csv`` wire format (``save_list`` :1403-1478, ``sort_list`` :1538-1574, ``get_data``
:846-919), ``reshape_result`` (:1336-1347) and the ``_meta.json`` side file.  Everything else in the
reference's helper module (GUI dialogs, xlsx collation, log roll-over, shutdown) belongs to the
offline/desktop half and is out of scope.
"""
from __future__ import annotations

import configparser
import json
import logging
import os
from datetime import datetime

import numpy as np

__all__ = ["create_configs", "get_configs", "default_settings", "reshape_result", "save_list", "sort_list",
           "get_data", "create_results_folder", "metadata_file", "get_loggers", "COLOR_BGR2GRAY"]

COLOR_BGR2GRAY = 6  # value of cv2.COLOR_BGR2GRAY; the only colour filter the HIP path implements

#: section -> [(key, default)], same names and defaults as helper_file.py:160-284
TRACKING_INI = {
    "BASIC RECORDING SETTINGS": [
        ("pixel per micrometre", 1.41888781), ("frames per second", 30.0), ("frame height", 922),
        ("frame width", 1228), ("white bacteria on dark background", True), ("rod shaped bacteria", True),
        ("threshold offset for detection", 5)],
    "BASIC TRACK DATA ANALYSIS SETTINGS": [
        ("minimal length in seconds", 20.0), ("limit track length to x seconds", 20.0),
        ("minimal angle in degrees for turning point", 30.0), ("extreme area outliers lower end in px*px", 2),
        ("extreme area outliers upper end in px*px", 50)],
    "DISPLAY SETTINGS": [
        ("user input", True), ("select files", True), ("display video analysis", True), ("save video", False)],
    "RESULTS SETTINGS": [
        ("rename previous result .csv", False), ("delete .csv file after analysis", False),
        ("store processed .csv file", True), ("store generated statistical .csv file", True),
        ("store final analysed .csv file", True),
        ("split results by (Turn Points / Distance / Speed / Time / Displacement / perc. motile)", "perc. motile"),
        ("split violin plots on", "0.0, 20.0, 40.0, 60.0, 80.0, 100.01"), ("save large plots", True),
        ("save rose plot", True), ("save time violin plot", True), ("save acr violin plot", True),
        ("save length violin plot", True), ("save turning point violin plot", True),
        ("save speed violin plot", True), ("save angle distribution plot / bins", 36),
        ("save displacement violin plot", True), ("save percent motile plot", True),
        ("collate results csv to xlsx", True)],
    "PLOT Y-AXIS LIMITS": [
        ("turning point violin plot min", 0.0), ("turning point violin plot max", False),
        ("length violin plot min", 0.0), ("length violin plot max", False),
        ("speed violin plot min", 0.0), ("speed violin plot max", False),
        ("time violin plot min", 0.0), ("time violin plot max", False),
        ("displacement violin plot min", 0.0), ("displacement violin plot max", False),
        ("percent motile plot min", 0.0), ("percent motile plot max", 100.0),
        ("acr violin plot min", 0.0), ("acr violin plot max", 1.0)],
    "LOGGING SETTINGS": [
        ("log to file", True), ("log file path", "./logfile.log"), ("shorten displayed logging output", False),
        ("shorten logfile logging output", False), ("set logging level (debug/info/warning/critical)", "debug"),
        ("verbose", False)],
    "ADVANCED VIDEO SETTINGS": [
        ("include luminosity in tracking calculation", False), ("color filter", "COLOR_BGR2GRAY"),
        ("minimal frame count", 600), ("stop evaluation on error", True), ("list save length interval", 10000),
        ("save video file extension", ".mp4"), ("save video fourcc codec", "mp4v"),
        ("adaptive double threshold", 2.0)],
    "ADVANCED TRACK DATA ANALYSIS SETTINGS": [
        ("maximal consecutive holes", 5), ("maximal empty frames in %", 5.0),
        ("percent quantiles excluded area", 10.0), ("try to omit motility outliers", True),
        ("stop excluding motility outliers if total count above percent", 5.0),
        ("exclude measurement when above x times average area", 1.5),
        ("rod average width/height ratio min.", 0.125), ("rod average width/height ratio max.", 0.67),
        ("coccoid average width/height ratio min.", 0.8), ("coccoid average width/height ratio max.", 1.0),
        ("percent of screen edges to exclude", 5.0), ("maximal recursion depth", 960),
        ("limit track length exactly", False), ("compare angle between n frames", 10),
        ("force tracking.ini fps settings", False)],
    "GAUSSIAN-SUM FIR FILTER SETTINGS": [
        ("disable gsff", False), ("number of LSFFs", 3), ("minimum horizon size", 0),
        ("maximum horizon size", 30)],
    "HOUSEKEEPING": [("previous directory", "./"), ("shut down after analysis", False)],
    "TEST SETTINGS": [("debugging", False), ("path to test video", "Q:/test_video.avi")],
}

_LOG_LEVELS = {"debug": logging.DEBUG, "info": logging.INFO, "warning": logging.WARNING,
               "critical": logging.CRITICAL}


def _logger():
    return logging.getLogger("ysmr").getChild(__name__)


def create_configs(config_filepath=None):
    """(Re)write ``tracking.ini`` with the default values; an existing file is renamed with a
    timestamp first (helper_file.py:152-158).  Unlike the reference this does not try to open the
    file in a desktop editor."""
    if config_filepath is None:
        config_filepath = os.path.join(os.path.abspath("./"), "tracking.ini")
    if os.path.isfile(config_filepath):
        root, ext = os.path.splitext(config_filepath)
        os.rename(config_filepath, "{}_{}{}".format(root, datetime.now().strftime("%y%m%d%H%M%S"), ext))
    parser = configparser.ConfigParser(allow_no_value=True)
    for section, items in TRACKING_INI.items():
        parser[section] = {k: str(v) for k, v in items}
    try:
        with open(config_filepath, "w+") as fh:
            parser.write(fh)
        _logger().critical("tracking.ini was reset to default values. Path: %s", config_filepath)
    except OSError as exc:
        _logger().exception("Could not create config file: %s", exc)


def _float_or_false(text):
    try:
        return float(text)
    except (TypeError, ValueError):
        return False


def _build_settings(parser, ini_path):
    rec, trk = parser["BASIC RECORDING SETTINGS"], parser["BASIC TRACK DATA ANALYSIS SETTINGS"]
    dsp, res, yax = parser["DISPLAY SETTINGS"], parser["RESULTS SETTINGS"], parser["PLOT Y-AXIS LIMITS"]
    log, vid = parser["LOGGING SETTINGS"], parser["ADVANCED VIDEO SETTINGS"]
    adv, gsf = parser["ADVANCED TRACK DATA ANALYSIS SETTINGS"], parser["GAUSSIAN-SUM FIR FILTER SETTINGS"]
    hk, tst = parser["HOUSEKEEPING"], parser["TEST SETTINGS"]

    verbose = log.getboolean("verbose")
    level_name = log.get("set logging level (debug/info/warning/critical)")
    level = logging.DEBUG if verbose else _LOG_LEVELS.get(level_name.lower(), logging.DEBUG)
    shape = "rod" if rec.getboolean("rod shaped bacteria") else "coccoid"
    colour = vid.get("color filter")
    if colour == "COLOR_BGR2GRAY":
        colour = COLOR_BGR2GRAY
    elif colour.isdigit():
        colour = int(colour)
    splits = [float(v.strip()) for v in res.get("split violin plots on").split(",")]
    split_by = res.get("split results by (Turn Points / Distance / Speed / Time / Displacement / perc. motile)")
    warn = False
    if "perc. motile" in split_by.lower() and max(splits) == 100:
        warn = ["Violin plots are set to 'perc. motile', but 'split violin plots on' highest value is 100."]
    try:
        n_max = int(gsf.get("maximum horizon size"))
        n_max = n_max if n_max > 0 else None
    except (TypeError, ValueError):
        n_max = None

    s = {
        "pixel per micrometre": rec.getfloat("pixel per micrometre"),
        "frames per second": rec.getfloat("frames per second"),
        "frame height": rec.getint("frame height"),
        "frame width": rec.getint("frame width"),
        "white bacteria on dark background": rec.getboolean("white bacteria on dark background"),
        "rod shaped bacteria": rec.getboolean("rod shaped bacteria"),
        "threshold offset for detection": rec.getint("threshold offset for detection"),
        "minimal length in seconds": trk.getfloat("minimal length in seconds"),
        "limit track length to x seconds": trk.getfloat("limit track length to x seconds"),
        "minimal angle in degrees for turning point": trk.getfloat("minimal angle in degrees for turning point"),
        "extreme area outliers lower end in px*px": trk.getint("extreme area outliers lower end in px*px"),
        "extreme area outliers upper end in px*px": trk.getint("extreme area outliers upper end in px*px"),
        "user input": dsp.getboolean("user input"),
        "select files": dsp.getboolean("select files"),
        "display video analysis": dsp.getboolean("display video analysis"),
        "save video": dsp.getboolean("save video"),
        "split results by (Turn Points / Distance / Speed / Time / Displacement / perc. motile)": split_by,
        "split violin plots on": splits,
        "save angle distribution plot / bins": res.getint("save angle distribution plot / bins"),
    }
    for key in ("rename previous result .csv", "delete .csv file after analysis", "store processed .csv file",
                "store generated statistical .csv file", "store final analysed .csv file", "save large plots",
                "save rose plot", "save time violin plot", "save acr violin plot", "save length violin plot",
                "save turning point violin plot", "save speed violin plot", "save displacement violin plot",
                "save percent motile plot", "collate results csv to xlsx"):
        s[key] = res.getboolean(key)
    for key, _ in TRACKING_INI["PLOT Y-AXIS LIMITS"]:
        s[key] = _float_or_false(yax.get(key))
    s.update({
        "log to file": log.getboolean("log to file"),
        "log file path": log.get("log file path"),
        "shorten displayed logging output": log.getboolean("shorten displayed logging output"),
        "shorten logfile logging output": log.getboolean("shorten logfile logging output"),
        "set logging level (debug/info/warning/critical)": level_name,
        "log_level": level,
        "verbose": verbose,
        "include luminosity in tracking calculation": vid.getboolean("include luminosity in tracking calculation"),
        "color filter": colour,
        "minimal frame count": vid.getint("minimal frame count"),
        "stop evaluation on error": vid.getboolean("stop evaluation on error"),
        "list save length interval": vid.getint("list save length interval"),
        "save video file extension": vid.get("save video file extension"),
        "save video fourcc codec": vid.get("save video fourcc codec"),
        "adaptive double threshold": vid.getfloat("adaptive double threshold"),
        "maximal consecutive holes": adv.getint("maximal consecutive holes"),
        "maximal empty frames in %": adv.getfloat("maximal empty frames in %") / 100 + 1,
        "percent quantiles excluded area": adv.getfloat("percent quantiles excluded area") / 100,
        "try to omit motility outliers": adv.getboolean("try to omit motility outliers"),
        "stop excluding motility outliers if total count above percent":
            adv.getfloat("stop excluding motility outliers if total count above percent") / 100,
        "exclude measurement when above x times average area":
            adv.getfloat("exclude measurement when above x times average area"),
        "average width/height ratio min.": adv.getfloat(f"{shape} average width/height ratio min."),
        "average width/height ratio max.": adv.getfloat(f"{shape} average width/height ratio max."),
        "percent of screen edges to exclude": adv.getfloat("percent of screen edges to exclude") / 100,
        "maximal recursion depth": adv.getint("maximal recursion depth"),
        "limit track length exactly": adv.getboolean("limit track length exactly"),
        "compare angle between n frames": adv.getint("compare angle between n frames"),
        "force tracking.ini fps settings": adv.getboolean("force tracking.ini fps settings"),
        "disable gsff": gsf.getboolean("disable gsff"),
        "number of LSFFs": gsf.getint("number of LSFFs"),
        "minimum horizon size": gsf.getint("minimum horizon size"),
        "maximum horizon size": n_max,
        "previous directory": hk.get("previous directory", fallback="./"),
        "shut down after analysis": hk.getboolean("shut down after analysis"),
        "debugging": tst.getboolean("debugging"),
        "path to test video": tst.get("path to test video"),
        "tracking_ini_filepath": ini_path,
        "perc_motile_warning": warn,
    })
    assert s["minimum horizon size"] >= 0, "'minimum horizon size' less than 0"
    assert s["number of LSFFs"] > 1, "'number of LSFFs' less than 2"
    assert s["frames per second"] > 0, "'frames per second' zero or negative"
    assert s["pixel per micrometre"] > 0 and s["frame height"] > 0 and s["frame width"] > 0
    return s


def default_settings(**overrides):
    """The settings dict ``get_configs`` would build from a pristine tracking.ini, without touching
    the file system.  The three interactive defaults (``user input``, ``select files``,
    ``display video analysis``) stay as upstream; pass overrides for headless use."""
    parser = configparser.ConfigParser(allow_no_value=True)
    for section, items in TRACKING_INI.items():
        parser[section] = {k: str(v) for k, v in items}
    s = _build_settings(parser, None)
    s.update(overrides)
    return s


def get_configs(tracking_ini_filepath=None):
    """Read ``tracking.ini`` into the flat settings dict, or pass an existing dict through
    (helper_file.py:595-596).  A missing or broken file is regenerated with defaults and ``None`` is
    returned, as upstream (helper_file.py:840-843)."""
    if isinstance(tracking_ini_filepath, dict):
        return tracking_ini_filepath
    if tracking_ini_filepath is None:
        tracking_ini_filepath = os.path.join(os.path.abspath("./"), "tracking.ini")
    path = os.path.abspath(tracking_ini_filepath)
    parser = configparser.ConfigParser(allow_no_value=True)
    settings = None
    try:
        parser.read(path)
        settings = _build_settings(parser, path)
        missing = [k for k, v in settings.items() if v is None and k not in ("maximum horizon size",)]
        if missing:
            _logger().critical("tracking.ini is missing a value in %s", missing[0])
            settings = None
    except (TypeError, ValueError, KeyError, AssertionError, AttributeError) as exc:
        _logger().exception("An exception of type %s occurred while attempting to read tracking.ini: %r",
                            type(exc).__name__, exc.args)
    if not settings:
        create_configs(config_filepath=path)
        return None
    return settings


def get_loggers(log_level=logging.DEBUG, logfile_name="./logfile.log", short_stream_output=False,
                short_file_output=False, log_to_file=False, settings=None):
    """Attach a stream handler (and optionally a file handler) to the 'ysmr' logger once."""
    logger = logging.getLogger("ysmr")
    logger.setLevel(log_level)
    if not logger.handlers:
        fmt = "%(message)s" if short_stream_output else "{asctime} {name:25} {levelname:8} {message}"
        handler = logging.StreamHandler()
        handler.setFormatter(logging.Formatter(fmt, style="%" if short_stream_output else "{"))
        logger.addHandler(handler)
        if log_to_file and logfile_name:
            try:
                fh = logging.FileHandler(logfile_name)
                fh.setFormatter(logging.Formatter("{asctime} {name:25} {levelname:8} {message}", style="{"))
                logger.addHandler(fh)
            except OSError:
                pass
    return logger


def create_results_folder(path):
    """``<folder of path>/<yymmdd>_Results`` (created on demand)."""
    folder = os.path.join(os.path.dirname(os.path.abspath(path)), "{}_Results".format(datetime.now().strftime("%y%m%d")))
    os.makedirs(folder, exist_ok=True)
    return folder


def reshape_result(tuple_of_tuples, *args):
    """((x, y), (w, h), deg) -> ((x, y, *args), (w, h, deg))   (helper_file.py:1336-1347)."""
    (x, y), (w, h), degrees = tuple_of_tuples
    return tuple([x, y, *args]), (w, h, degrees)


CSV_HEADER = "TRACK_ID,POSITION_T,POSITION_X,POSITION_Y,WIDTH,HEIGHT,DEGREES_ANGLE\n"

# Removing a previous list of tens of megabytes is milliseconds of kernel work (its pages in the page cache are freed one by
# one: 6 ms for the 80 MB list of a 1920-frame video, as long as 150 frames take).  save_list frees the NAME at once -- the old
# file is renamed aside -- and a thread unlinks it while the video runs; track_bacteria waits for that thread before it returns.
_REMOVE_ASIDE_FROM = 8 << 20
_REMOVALS = []


def _remove_previous_list(csv_path):
    if os.path.getsize(csv_path) < _REMOVE_ASIDE_FROM:
        os.remove(csv_path)
        return
    import threading
    aside = "{}.{}.removing".format(csv_path, os.getpid())
    os.rename(csv_path, aside)

    def unlink():
        try:
            os.remove(aside)
        except OSError as exc:
            logging.getLogger("ysmr").getChild(__name__).error("Could not remove {}: {!r}".format(aside, exc.args))
    th = threading.Thread(target=unlink, name="ysmr-remove-previous-list")
    th.start()
    _REMOVALS.append(th)


def wait_for_removals():
    """Join the threads that unlink previous lists (see above)."""
    while _REMOVALS:
        _REMOVALS.pop().join()


def save_list(path, result_folder=None, coords=None, first_call=False, rename_old_list=True, illumination=False):
    """Create ``<name>_list.csv`` with its header (first_call) or append rows.

    ``coords`` items are ``(frame, id, (x, y), (w, h, deg))`` like upstream; formatting follows
    helper_file.py:1455-1475 byte for byte (``str.format`` of ints/floats)."""
    if illumination:
        raise NotImplementedError("luminosity tracking is out of scope")
    if first_call:
        folder = result_folder if result_folder is not None else os.path.dirname(path)
        name = os.path.splitext(os.path.basename(path))[0]
        csv_path = os.path.join(folder, "{}_list.csv".format(name))
        old = False
        if os.path.isfile(csv_path):
            if rename_old_list:
                root, ext = os.path.splitext(csv_path)
                old = "{}_{}{}".format(root, datetime.now().strftime("%y%m%d%H%M%S"), ext)
                os.rename(csv_path, old)
            else:
                _remove_previous_list(csv_path)
        with open(csv_path, "w+", newline="") as fh:
            fh.write(CSV_HEADER)
        return old, csv_path
    if coords:
        lines = []
        for frame, obj_id, xy, (w, h, deg) in coords:
            lines.append("{0},{1},{2},{3},{4},{5},{6}\n".format(int(obj_id), int(frame), xy[0], xy[1], w, h, deg))
        with open(path, "a", newline="") as fh:
            fh.write("".join(lines))
    return None, None


def rows_to_csv_text(rows):
    """Device rows (structured ``ysmr_row`` array) -> the text ``save_list`` would have appended.
    Disappeared tracks carry the integer zeros the reference writes (tracker.py:101, 205)."""
    out = []
    for r in rows:
        if r["disappeared"] > 0:
            w = h = deg = 0
        else:
            w, h, deg = float(r["w"]), float(r["h"]), float(r["angle"])
        out.append("{0},{1},{2},{3},{4},{5},{6}\n".format(int(r["track_id"]), int(r["frame"]), np.float64(r["x"]),
                                                          np.float64(r["y"]), w, h, deg))
    return "".join(out)


def rows_to_csv_bytes(rows, header=True, via_pandas=True, threads=0):
    """The csv the reference ends up with for these rows, as bytes: header of save_list
    (helper_file.py:1451) and one line per row formatted the way ``DataFrame.to_csv`` prints the
    uint32 / float64 columns (helper_file.py:1366-1400) -- produced by the library's native
    formatter (``ysmr_rows_format_csv``, a host function) instead of Python string formatting and
    a pandas round trip.  ``rows``: structured ``ysmr_row`` array, already in file order.
    ``via_pandas``: reproduce the 1-ulp noise of the reference's text -> read_csv detour
    (include/ysmr_hip.h); False prints the exact values."""
    return _rows_csv_buffer(rows, header, via_pandas, threads).tobytes()


def _rows_csv_buffer(rows, header, via_pandas, threads):
    import ctypes
    from . import _lib
    rows = np.ascontiguousarray(rows, dtype=_lib.ROW_DTYPE)
    L = _lib.lib()
    cap = L.ysmr_rows_csv_bound(len(rows), int(bool(header)))
    out = np.empty(cap, dtype=np.uint8)
    n = ctypes.c_size_t(0)
    _lib.check(L.ysmr_rows_format_csv(rows.ctypes.data, len(rows), int(bool(header)), int(bool(via_pandas)), int(threads),
                                      out.ctypes.data, cap, ctypes.byref(n)), "ysmr_rows_format_csv")
    return out[:n.value]


def rows_to_csv_file(rows, path, header=True, via_pandas=True, threads=0):
    """:func:`rows_to_csv_bytes` written straight to ``path`` by the formatting threads themselves
    (``ysmr_rows_write_csv``: each writes its own piece at its place in the file)."""
    import ctypes
    from . import _lib
    rows = np.ascontiguousarray(rows, dtype=_lib.ROW_DTYPE)
    n = ctypes.c_size_t(0)
    _lib.check(_lib.lib().ysmr_rows_write_csv(rows.ctypes.data, len(rows), int(bool(header)), int(bool(via_pandas)), int(threads),
                                              os.fsencode(path), ctypes.byref(n)), "ysmr_rows_write_csv")
    return n.value


def rows_to_csv_file_and_dataframe(rows, path, header=True, via_pandas=True, threads=0):
    """``rows_to_csv_file`` and ``rows_to_dataframe`` in ONE native pass (``ysmr_rows_write_csv_columns``): the values the csv is
    printed from are the values pandas would read back from it -- worked out once for both.  Returns ``(bytes, DataFrame)``."""
    import ctypes
    import pandas as pd
    from . import _lib
    rows = np.ascontiguousarray(rows, dtype=_lib.ROW_DTYPE)
    n = len(rows)
    ids, t = np.empty(n, np.uint32), np.empty(n, np.uint32)
    cols = [np.empty(n, np.float64) for _ in range(5)]
    length = ctypes.c_size_t(0)
    _lib.check(_lib.lib().ysmr_rows_write_csv_columns(rows.ctypes.data, n, int(bool(header)), int(bool(via_pandas)), int(threads),
                                                      os.fsencode(path), ctypes.byref(length), ids.ctypes.data, t.ctypes.data,
                                                      *[c.ctypes.data for c in cols]), "ysmr_rows_write_csv_columns")
    df = pd.DataFrame({"TRACK_ID": ids, "POSITION_T": t, "POSITION_X": cols[0], "POSITION_Y": cols[1],
                       "WIDTH": cols[2], "HEIGHT": cols[3], "DEGREES_ANGLE": cols[4]})
    return length.value, df


#: pinned staging of the device formatter's results, kept between videos (pinning 100 MB costs more than the copies) -- per THREAD:
#: a GPU worker runs two stream threads whose tails may overlap
import threading as _threading
_PINNED = _threading.local()
#: (diagnostics) seconds since the call began at which the last rows_device_to_csv_file_and_dataframe reached its steps
LAST_DEVICE_ROWS_MARKS = {}


def _pinned(name, nbytes):
    import torch
    held = _PINNED.__dict__.setdefault("buffers", {})
    buf = held.get(name)
    if buf is None or buf.numel() < nbytes:
        buf = held[name] = torch.empty(max(int(nbytes * 1.25), 1 << 20), dtype=torch.uint8, pin_memory=True)
    return buf[:nbytes]


def rows_device_to_csv_file_and_dataframe(rows_u8, n, path, header=True, via_pandas=True):
    """``rows_to_csv_file_and_dataframe`` for rows that are on the DEVICE, in order (``tracker.sort_rows``): text and columns are
    worked out there (``ysmr_rows_format_device``), copied over and the text written to ``path`` (None: no file).  Returns
    ``(bytes, DataFrame)``, or ``None`` when the table holds a value the device form does not print (NaN, infinities,
    magnitudes outside 2^-20 .. 2^24): the caller then takes the host path (helper_file.py:1403-1478, 860-905, 1366-1400)."""
    import time
    import pandas as pd
    import torch
    from . import _lib
    t0 = time.perf_counter()
    marks = LAST_DEVICE_ROWS_MARKS
    marks.clear()
    L = _lib.lib()
    n = int(n)
    dev = rows_u8.device
    cap = int(L.ysmr_rows_csv_bound(n, int(bool(header))))
    ws_bytes = int(L.ysmr_rows_format_device_workspace_bytes(n)) if n else 0
    with torch.cuda.device(dev):
        ws = torch.empty(max(ws_bytes, 1), dtype=torch.uint8, device=dev)
        csv = torch.empty(cap, dtype=torch.uint8, device=dev)
        meta = torch.zeros(2, dtype=torch.int64, device=dev)          # csv length; unserved rows (low 32 bits)
        # the seven columns in one buffer: TRACK_ID, POSITION_T (uint32), then the five float64 columns
        colbuf = torch.empty(max(n, 1) * 48, dtype=torch.uint8, device=dev)
        base = colbuf.data_ptr()
        _lib.check(L.ysmr_rows_format_device(_lib.stream_ptr(dev), rows_u8.data_ptr(), n, int(bool(header)), int(bool(via_pandas)),
                                             ws.data_ptr(), ws_bytes, csv.data_ptr(), cap, meta.data_ptr(), base, base + 4 * n,
                                             *[base + 8 * n + 8 * n * k for k in range(5)], meta[1:].data_ptr()), "ysmr_rows_format_device")
        host_cols = _pinned("columns", 48 * n)
        host_cols.copy_(colbuf[:48 * n], non_blocking=True)
        length, unserved = (int(v) for v in meta.cpu())
        marks["formatted"] = time.perf_counter() - t0
        if unserved & 0xFFFFFFFF:
            torch.cuda.current_stream(dev).synchronize()
            return None
        text = _pinned("text", length)
        text.copy_(csv[:length], non_blocking=True)
        torch.cuda.current_stream(dev).synchronize()
    marks["copied"] = time.perf_counter() - t0
    # the file is written by a thread of its own (the kernel's copy into the page cache: no GIL) while this one builds the table
    writer, failure = None, []
    if path is not None:
        import threading

        def write():
            try:
                _write_file(path, text.numpy())
            except OSError as exc:
                failure.append(exc)
        writer = threading.Thread(target=write)
        writer.start()
    hc = host_cols.numpy()
    ids_h, t_h = hc[:4 * n].view(np.uint32), hc[4 * n:8 * n].view(np.uint32)
    cols_h = hc[8 * n:48 * n].view(np.float64).reshape(5, n)
    # (the DataFrame copies its columns into its own blocks: the pinned staging is free for the next video)
    df = pd.DataFrame({"TRACK_ID": ids_h, "POSITION_T": t_h, "POSITION_X": cols_h[0], "POSITION_Y": cols_h[1],
                       "WIDTH": cols_h[2], "HEIGHT": cols_h[3], "DEGREES_ANGLE": cols_h[4]})
    marks["frame"] = time.perf_counter() - t0
    if writer is not None:
        writer.join()
        if failure:
            raise failure[0]
    marks["written"] = time.perf_counter() - t0
    return length, df


def _write_file(path, data):
    """``data`` (a uint8 array) to ``path`` (created or truncated).  One writer: buffered writes to ONE file take the inode's lock
    one at a time, so eight threads writing a piece each were no faster than one (60 MB: 13.5 against 14.0 ms,
    scripts/file_write_probe.py); reserving the blocks first is (11 ms)."""
    n = len(data)
    fd = os.open(path, os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o666)
    try:
        if n >= (1 << 20):
            try:
                os.posix_fallocate(fd, 0, n)
            except OSError:
                pass                    # (a file system without it: the writes extend the file)
        view, at = memoryview(data), 0
        while at < n:
            at += os.pwrite(fd, view[at:at + (64 << 20)], at)
    except OSError:
        os.close(fd)
        fd = -1
        try:
            os.unlink(path)             # (as the native writer: no partial file is left behind)
        except OSError:
            pass
        raise
    finally:
        if fd >= 0:
            os.close(fd)


class RowStream:
    """``ysmr_rows_stream_*``: the csv and the DataFrame worked out while the video runs.  ``push`` rows (a structured
    ``_lib.ROW_DTYPE`` array, or a raw pointer and a count) as the link emits them; ``finish`` orders them by
    (TRACK_ID, POSITION_T), writes ``path`` (None: no file) and returns ``(bytes, DataFrame)`` -- what
    ``rows_to_csv_file_and_dataframe`` returns for the same rows in sorted order."""

    def __init__(self, via_pandas=True, threads=0):
        import ctypes
        from . import _lib
        self._h = ctypes.c_void_p()
        _lib.check(_lib.lib().ysmr_rows_stream_create(int(threads), int(bool(via_pandas)), ctypes.byref(self._h)),
                   "ysmr_rows_stream_create")

    def push(self, rows, n=None):
        from . import _lib
        if n is None:
            rows = np.ascontiguousarray(rows, dtype=_lib.ROW_DTYPE)
            ptr, n = rows.ctypes.data, len(rows)
        else:
            ptr = int(rows)
        _lib.check(_lib.lib().ysmr_rows_stream_push(self._h, ptr, int(n)), "ysmr_rows_stream_push")

    def __len__(self):
        from . import _lib
        return int(_lib.lib().ysmr_rows_stream_count(self._h))

    def finish(self, path=None, header=True):
        import ctypes
        import pandas as pd
        from . import _lib
        n = len(self)
        ids, t = np.empty(n, np.uint32), np.empty(n, np.uint32)
        cols = [np.empty(n, np.float64) for _ in range(5)]
        length = ctypes.c_size_t(0)
        _lib.check(_lib.lib().ysmr_rows_stream_finish(self._h, int(bool(header)), None if path is None else os.fsencode(path),
                                                      ctypes.byref(length), ids.ctypes.data, t.ctypes.data,
                                                      *[c.ctypes.data for c in cols]), "ysmr_rows_stream_finish")
        df = pd.DataFrame({"TRACK_ID": ids, "POSITION_T": t, "POSITION_X": cols[0], "POSITION_Y": cols[1],
                           "WIDTH": cols[2], "HEIGHT": cols[3], "DEGREES_ANGLE": cols[4]})
        return length.value, df

    def close(self):
        from . import _lib
        if self._h:
            _lib.lib().ysmr_rows_stream_destroy(self._h)
            self._h = None

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass


def rows_to_dataframe(rows, via_pandas=True):
    """Rows (already sorted) -> the DataFrame ``get_data`` would have read back from the csv
    (dtypes of helper_file.py:881-889; ``via_pandas`` as in :func:`rows_to_csv_bytes`)."""
    import pandas as pd
    from . import _lib
    rows = np.ascontiguousarray(rows, dtype=_lib.ROW_DTYPE)
    n = len(rows)
    ids, t = np.empty(n, np.uint32), np.empty(n, np.uint32)
    cols = [np.empty(n, np.float64) for _ in range(5)]
    _lib.check(_lib.lib().ysmr_rows_columns(rows.ctypes.data, n, int(bool(via_pandas)), ids.ctypes.data, t.ctypes.data,
                                            *[c.ctypes.data for c in cols]), "ysmr_rows_columns")
    return pd.DataFrame({"TRACK_ID": ids, "POSITION_T": t, "POSITION_X": cols[0], "POSITION_Y": cols[1],
                         "WIDTH": cols[2], "HEIGHT": cols[3], "DEGREES_ANGLE": cols[4]})


_DTYPES = {"TRACK_ID": np.uint32, "POSITION_T": np.uint32, "POSITION_X": np.float64, "POSITION_Y": np.float64,
           "WIDTH": np.float64, "HEIGHT": np.float64, "DEGREES_ANGLE": np.float64}


def get_data(csv_file_path, dtype=None, check_sorted=True):
    """Load a ``*_list.csv`` into a DataFrame with the reference's dtypes (helper_file.py:881-889)."""
    import pandas as pd
    dtype = _DTYPES if dtype is None else dtype
    try:
        with open(csv_file_path, "r", newline="\n") as fh:
            df = pd.read_csv(fh, sep=",", header=0, usecols=list(dtype.keys()), dtype=dtype)
    except (ValueError, OSError) as exc:
        _logger().exception(exc)
        return None
    if check_sorted and {"TRACK_ID", "POSITION_T"} <= set(dtype) and df.loc[:5, "TRACK_ID"].is_unique:
        df = sort_list(df=df, save_file=False)
    return df


def sort_list(file_path=None, sort=None, df=None, save_file=False):
    """Sort by ``TRACK_ID, POSITION_T`` (helper_file.py:1538-1574); optionally rewrite the csv."""
    sort = ["TRACK_ID", "POSITION_T"] if sort is None else ([sort] if isinstance(sort, (str, bytes)) else sort)
    if file_path is not None and df is None:
        df = get_data(file_path, check_sorted=False)
    if df is None:
        _logger().warning("No Dataframe read")
        return None
    df.sort_values(by=sort, inplace=True, na_position="first")
    df.reset_index(drop=True, inplace=True)
    if save_file and file_path is not None:
        df.to_csv(file_path, index=False)
    return df


def save_df_to_csv(df, save_path, rename_old_file=True):
    """Write a DataFrame as csv without its index (helper_file.py:1366-1400); an existing file of the
    same name is first moved aside to ``<yymmddHHMMSS>.<old file name>``."""
    if rename_old_file:
        folder, name = os.path.split(save_path)
        try:
            moved = os.path.join(folder, "{}.{}".format(datetime.now().strftime("%y%m%d%H%M%S"), name))
            os.rename(save_path, moved)
            _logger().critical("Old {} renamed to {}".format(name, moved))
        except (FileNotFoundError, FileExistsError):
            pass
        except OSError as exc:
            _logger().exception("Could not move previous file {} aside: {}".format(save_path, exc))
    try:
        with open(save_path, "w+", newline="\n") as fh:
            df.to_csv(fh, index=False, encoding="utf-8")
        _logger().debug("Selected results saved to: {}".format(save_path))
    except OSError as exc:
        _logger().exception("Could not save {}: {}".format(save_path, exc))


def _meta_path_of(path):
    """``<name>_meta.json`` for a video, one of the pipeline's own csv files, or a meta file itself."""
    path = os.fspath(path)
    if path.endswith("_meta.json"):
        return path
    for ext in ("_analysed.csv", "_list.csv", "_selected_data.csv", "_statistics.csv"):
        if path.endswith(ext):
            return path[: -len(ext)] + "_meta.json"
    return os.path.splitext(path)[0] + "_meta.json"


def metadata_file(path=None, verbose=False, additional_search_paths=None, **kwargs):
    """Read/update the ``<name>_meta.json`` side file (fps, frame_height, frame_width, caller's notes).
    Looked for next to ``path``, in the folder above it, then next to every additional search path; the
    first one found is read and, when new non-None values are given, rewritten in place (none found: it is
    created next to ``path``).  New values win over stored ones (helper_file.py:1267-1333)."""
    path = os.fspath(path)
    folder, name = os.path.split(path)
    candidates = [path, os.path.join(os.path.dirname(folder), name)]
    if additional_search_paths:
        if isinstance(additional_search_paths, (str, os.PathLike)):
            candidates.append(additional_search_paths)
        else:
            candidates.extend(additional_search_paths)
    candidates = [_meta_path_of(c) for c in candidates]
    meta, meta_path = {}, candidates[0]
    for cand in candidates:
        if verbose:
            _logger().debug("Searching for meta file in path: {}".format(cand))
        try:
            with open(cand) as fh:
                stored = json.load(fh)
        except (OSError, ValueError):
            continue
        meta.update({k: v for k, v in stored.items() if v is not None})
        meta_path = cand
        break
    new = {k: v for k, v in kwargs.items() if v is not None}
    if new:
        meta.update(new)
        try:
            with open(meta_path, "w+") as fh:
                json.dump(meta, fh)
        except OSError as exc:
            _logger().exception(exc)
    return meta
