"""Timeline of the two-launch link (k_link, k_track) on the 4K configuration from device realtime stamps: where the
time between a k_link's entry and the next one goes -- with the link alone on the GPU, and with the detection of the
next batch running beside it as in the pipeline.  Needs a stamps build of the library (scripts/build_stamps.sh,
YSMR_HIP_LIB=scripts/var_stamps.so)."""
import sys, os, ctypes, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ysmr_amd.helper_file import default_settings
from ysmr_amd.synth import SyntheticVideo
from ysmr_amd.track_eval import TrackingPipeline
from ysmr_amd import _lib
B, H, W = 16, 2160, 3840
NB = 6
F = NB * B
frames = torch.from_numpy(SyntheticVideo(H, W, 5000, seed=0).frames(F)).cuda()
pipe = TrackingPipeline(H, W, 30.0, default_settings(), batch=B, max_det=8192, capacity=8192, rows_per_flush=4 * F * 8192)
L = _lib.lib()
PH = {9: "link entry", 10: "counters", 11: "tables cleared", 17: "rows loaded", 18: "column minima", 19: "column winners", 12: "claims", 13: "ageing", 14: "compaction", 15: "registration", 16: "link end",
      4: "track entry", 5: "track slot known", 6: "track filters done", 7: "track block 0 done", 8: "track last block done"}
ORDER = [9, 10, 11, 17, 18, 19, 12, 13, 14, 15, 16, 4, 5, 6, 7, 8]


def read_ring():
    buf = (ctypes.c_ulonglong * 8192)(); n = ctypes.c_uint(0)
    L.ysmr_debug_read_ring(buf, ctypes.byref(n))
    a = np.array(buf[:], dtype=np.uint64).reshape(4096, 2)
    k = min(int(n.value), 4096)
    return sorted((int(t), int(tag) >> 40, int(tag) & 0xFFFFFFFF) for tag, t in a[:k])


def wave_ends():
    """frame & 63 -> when the last wave of that frame's k_track finished (the newest such frame)"""
    buf = (ctypes.c_ulonglong * (64 * 8192))()
    L.ysmr_debug_read_wave_end(buf)
    return np.array(buf[:], dtype=np.uint64).reshape(64, 8192).max(axis=1)


def report(title, ev, keep):
    by = {}
    # (k_track_lanes, the per-track half since round 5, carries no stamps: its time is then the interval between k_link's end and
    # the next k_link's entry)
    order = ORDER if any(ph == 4 for _, ph, _ in ev) else [p for p in ORDER if p >= 9]
    for t, ph, fr in ev:
        if ph in PH and keep(fr): by.setdefault(fr, {})[ph] = t
    if order is ORDER:
        ends = wave_ends()
        newest = max(fr for _, ph, fr in ev if ph == 4)
        for fr in by:                      # (the per-wave slots hold the last 64 frames only)
            if fr > newest - 60: by[fr][8] = int(ends[fr & 63])
    rows = []
    for fr in sorted(by):
        d, nx = by[fr], by.get(fr + 1)
        if all(p in d for p in order) and nx and 9 in nx:
            ts = [d[p] for p in order] + [nx[9]]
            rows.append(np.diff(ts))
    r = np.array(rows) / 100.0          # s_memrealtime ticks at 100 MHz
    print(f"{title}: {len(r)} frames; median / mean / p90 us")
    names = [f"{PH[a]} -> {PH[b]}" for a, b in zip(order, order[1:])] + [("track last block done" if order is ORDER else "link end (k_track_lanes, unstamped)") + " -> next link entry"]
    for i, nm in enumerate(names):
        print(f"  {nm:44s} {np.median(r[:, i]):7.2f} {r[:, i].mean():7.2f} {np.percentile(r[:, i], 90):7.2f}")
    tot = r.sum(axis=1)
    print(f"  {'frame to frame':44s} {np.median(tot):7.2f} {tot.mean():7.2f} {np.percentile(tot, 90):7.2f}")


res = [pipe.det[i].detect(frames[i * B:(i + 1) * B]) for i in range(2)]
torch.cuda.synchronize()
for _ in range(2):
    pipe.reset()
    for k in range(4): pipe.trk.run(res[k & 1].det, res[k & 1].det_count, k * B, pipe.rows, pipe.row_count)
torch.cuda.synchronize()
ev = read_ring()
report("link alone (no detection running)", ev[len(ev) // 2:], lambda fr: fr % B not in (0, B - 1))

for rep in range(2):
    pipe.reset()
    pending = None
    for f0 in list(range(0, F, B)) + [None]:
        nxt = None
        if f0 is not None:
            nxt = (pipe.detect_async(frames[f0:f0 + B], frames_ready=False), f0)
        if pending is not None:
            (slot, r, ready), p0 = pending
            pipe.link(slot, r, ready, p0)
        pending = nxt
torch.cuda.synchronize()
ev = read_ring()
report("link beside the next batch's detection", [e for e in ev if e[2] >= 2 * B], lambda fr: fr % B not in (0, B - 1) and fr < F - B)
