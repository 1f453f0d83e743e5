"""What the HOST of this box costs: pure Python, a kernel launch, an event record, a cross-stream wait -- and whether the
cgroup throttled the process meanwhile (cpu.stat).  Beside bench.py's host_enqueue_ms_per_step on boxes that differ 5x."""
import os, sys, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

def cpu_stat():
    out = {}
    for p in ("/sys/fs/cgroup/cpu.stat",):
        try:
            for l in open(p):
                k, v = l.split(); out[k] = int(v)
        except OSError:
            pass
    return out

def where():
    try:
        return ctypes.CDLL(None).sched_getcpu()
    except Exception:
        return -1

s0 = cpu_stat()
t = time.perf_counter(); x = 0
for i in range(2_000_000): x += i
py = time.perf_counter() - t
a = torch.zeros(64, device="cuda"); torch.cuda.synchronize()
side = torch.cuda.Stream()
def timed(fn, n=3000):
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    dt = time.perf_counter() - t; torch.cuda.synchronize()
    return dt / n * 1e6
ev = torch.cuda.Event()
launch = timed(lambda: a.add_(1.0))
rec = timed(lambda: ev.record())
def cross():
    ev.record(); side.wait_event(ev)
wait = timed(cross)
s1 = cpu_stat()
print(f"cpu {where()}, python loop {py * 1e3:.0f} ms per 2 M iterations, launch {launch:.1f} us, event record {rec:.1f} us, record + cross-stream wait {wait:.1f} us")
print("cgroup cpu.stat delta:", {k: s1[k] - s0.get(k, 0) for k in s1 if k in ("nr_periods", "nr_throttled", "throttled_usec", "usage_usec")})
print("cpu.max:", open("/sys/fs/cgroup/cpu.max").read().strip() if os.path.exists("/sys/fs/cgroup/cpu.max") else "?", " threads:", len(os.listdir("/proc/self/task")), " affinity:", len(os.sched_getaffinity(0)))
