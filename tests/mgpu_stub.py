"""Stand-in for ysmr_amd.main._gpu_worker in the multi-GPU dispatch tests (importable by name, so that
the spawned worker processes of ysmr(multiprocess=True) can unpickle it)."""
import os
import threading


def fake_gpu_worker(args):
    jobs, streams, physical = args
    return [(path, {"device": device, "pid": os.getpid(), "thread": threading.current_thread().name,
                    "streams": streams, "physical": physical}) for path, _settings, _folder, device in jobs]
