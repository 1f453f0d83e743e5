#!/usr/bin/env python3
"""Benchmark of the YSMR detect+link hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

Workload (BASELINE.json configs[2], the configuration `metric` is quoted on): one synthetic
1228x922 video stream with ~500 blobs per GPU.  A *step* is one pass of the whole hot path over a
clip of --frames distinct frames that are already resident in HBM: fused gray->blur->adaptive
double threshold, hysteresis + component labelling + minAreaRect (batched over --batch frames per
launch), then the strictly sequential link (nearest-detection claims, track lifecycle, GSFF, row
emission) frame by frame from a freshly reset tracker.  Every output the path owes -- class map,
final mask, label map, detections, rows -- is written inside the timed region.

Independent streams shard one per GPU (weak scaling, no data-path collective; torch.distributed is
used only for the barrier and the max-over-ranks time).

One JSON line on rank 0: frames/s (whole job), the roofline of the threshold kernel measured live
with HIP events on the launch stream, and the CPU oracle timed on this host's cores on a bounded
sample of the same clip (a reported baseline, not the target).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")   # (before torch loads the HIP runtime; see ysmr_amd/__init__.py)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--frames", type=int, default=None, help="distinct frames per clip (default: two batches, 496 x 1.13 MB > 256 MiB L3)")
    ap.add_argument("--batch", type=int, default=None, help="frames per detection launch")
    ap.add_argument("--height", type=int, default=922)
    ap.add_argument("--width", type=int, default=1228)
    ap.add_argument("--blobs", type=int, default=500)
    ap.add_argument("--max-det", type=int, default=2048)
    ap.add_argument("--capacity", type=int, default=768,
                    help="live tracks the link's tables hold (the clip peaks at ~535); <= 768 links a whole batch with one launch")
    ap.add_argument("--streams-per-gpu", type=int, default=1,
                    help="independent video streams processed concurrently on each GPU (the metric's configuration is 1)")
    ap.add_argument("--channels", type=int, default=1, choices=(1, 3),
                    help="1: gray frames (the metric's configuration); 3: the same frames as BGR (B=G=R), which adds a1")
    ap.add_argument("--adt", type=float, default=2.0,
                    help="'adaptive double threshold' of tracking.ini (default 2.0); < 0 selects the reference's "
                         "mean-gray threshold branch (track_eval.py:219-253)")
    ap.add_argument("--detect-only", action="store_true",
                    help="BASELINE configs[1]: threshold + labelling + minAreaRect only, no link (a parity-test "
                         "configuration; the metric is quoted on the default, configs[2])")
    ap.add_argument("--config", type=int, choices=(0, 1, 2, 4), default=None,
                    help="preset of a BASELINE.json configuration on this GPU: 0 = ~50 blobs, 1 = detection only, "
                         "2 = the default, 4 = 3840x2160 / ~5000 blobs (batch 8, two-launch link)")
    ap.add_argument("--dist-backend", default="nccl", choices=("nccl", "gloo"),
                    help="process-group backend for the barrier and the max-over-ranks time (nccl = RCCL; gloo only "
                         "to rehearse the multi-rank path on a box with fewer GPUs than ranks)")
    ap.add_argument("--device-index", type=int, default=None,
                    help="GPU of this rank (default LOCAL_RANK); rehearsals put several ranks on one GPU")
    ap.add_argument("--cpu-sample", type=int, default=200, help="frames of the clip timed on the CPU oracle (0 = skip)")
    args = ap.parse_args()
    if args.config == 0:
        args.blobs = 50
    elif args.config == 1:
        args.detect_only = True
    elif args.config == 4:
        args.height, args.width, args.blobs, args.frames = 2160, 3840, 5000, 64
        args.batch = args.batch or 16
        args.max_det = args.capacity = 8192
        args.cpu_sample = min(args.cpu_sample, 8)
    # 256 frames per detection batch = per link launch (the whole 512-frame clip is resident anyway): the fixed costs of a
    # batch -- the threshold kernel's item starts, the link launch's state in and out and the gap to the next launch, the host's
    # calls -- are paid half / a quarter as often as at 128 / 64 (same box: 161.5 / 163.5 / 165 k frames/s, threshold kernel
    # 0.214 / 0.248 / 0.265 of the roofline, host 1.0 / 0.5 / 0.24 ms per step; profiles/r04_batch_sizes.log)
    # Round 5: as many frames as the threshold kernel has workgroups beside the batch link (248: ysmr_threshold_workgroups) -- every
    # workgroup takes one whole frame and starts one item; 256 frames on 248 workgroups start two each (0.344 instead of 0.327 of
    # the roofline, same frames/s: profiles/r05_batch_248.log).  Detection only: 256 workgroups, 256 frames.
    # (resolve_batch, called once the library may be loaded: parse() must not touch it -- main() builds it in a child process first)
    return args


def resolve_batch(args):
    """Frames per detection batch and per clip when the command line names none (see parse())."""
    if not args.batch:
        from ysmr_amd import _lib
        args.batch = 256 if args.detect_only else int(_lib.lib().ysmr_threshold_workgroups(_lib.BESIDE_BATCH_LINK))
    args.frames = args.frames or 2 * args.batch
    return args


def cpu_baseline(frames_np, sample, fps, adt=2.0, detect_only=False):
    """Time the CPU oracle (single thread, like the reference's one process per video) on the
    first `sample` frames of the clip."""
    if sample <= 0:
        return None
    from oracle import ysmr_oracle as yo
    yo.build()
    sample = min(sample, len(frames_np))
    t0 = time.perf_counter()
    if detect_only:
        rows = []
        for frame in frames_np[:sample]:
            rows.extend(yo.detect_frame(frame).det)
    else:
        rows, _ = yo.track_frames(frames_np[:sample], fps=fps, adt=adt)
    dt = time.perf_counter() - t0
    return {"value": sample / dt, "unit": "frames/s", "cores": 1, "kind": "port",
            "sample": f"first {sample} frames of the clip ({len(rows)} rows), oracle/ysmr_oracle.{{c,py}} "
                      f"single process; host has {os.cpu_count()} cpus"}


def gpu_clocks(index):
    """Current sclk / mclk / fclk of the device from sysfs (the level marked '*'); best effort, no subprocess."""
    import glob
    out = {}
    cards = sorted(glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk"))
    if not cards:
        return None
    base = os.path.dirname(cards[min(index, len(cards) - 1)])
    for name in ("sclk", "mclk", "fclk"):
        try:
            with open(os.path.join(base, "pp_dpm_" + name)) as fh:
                cur = [ln.split(":")[1].replace("*", "").strip() for ln in fh if "*" in ln]
            out[name] = cur[0] if cur else None
        except OSError:
            out[name] = None
    return out


def stats_us(vals):
    vals = sorted(vals)
    if not vals:
        return None
    pick = lambda q: vals[min(len(vals) - 1, int(q * len(vals)))]
    return {"avg": sum(vals) / len(vals), "min": vals[0], "p50": pick(0.5), "p99": pick(0.99), "max": vals[-1], "n": len(vals)}


def main():
    args = parse()
    # (re)build the library before anything touches the GPU: no child process is started from a process that
    # holds a GPU context, and a failed build stops the run instead of benchmarking a stale binary
    if int(os.environ.get("RANK", "0")) == 0 and not os.environ.get("YSMR_HIP_LIB"):
        import subprocess
        rc = subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "ysmr_amd", "csrc")]).returncode
        if rc:
            raise SystemExit(f"bench.py: building libysmr_hip.so failed (make exit code {rc})")
    import torch

    from ysmr_amd import dist
    info = dist.rank_info()
    rank, local_rank, world = info.rank, info.local_rank, info.world
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if args.device_index is not None:
        local_rank = args.device_index
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # every rank runs on the CPUs next to its GPU (ysmr_amd/dist.py: sysfs local_cpulist of the device's PCI address): the
    # link wants a launch every ~11 us from this process, and N ranks' threads left to the scheduler share cores
    pinned = dist.pin_to_gpu(local_rank) if world > 1 or os.environ.get("YSMR_BENCH_PIN") == "1" else None
    dist.init(info, backend=args.dist_backend, device=dev)   # barrier + max-over-ranks time only; no data-path collective

    dist.barrier(info)   # (rank 0 built the library before any rank loads it)
    from ysmr_amd.helper_file import default_settings
    from ysmr_amd.synth import SyntheticVideo
    from ysmr_amd.track_eval import TrackingPipeline

    resolve_batch(args)
    F, B, H, W, S = args.frames, args.batch, args.height, args.width, max(1, args.streams_per_gpu)
    fps_video = 30.0
    settings = default_settings()                    # tracking.ini defaults: offset 5, adt 2.0, GSFF 10/20/30
    settings["adaptive double threshold"] = args.adt
    mean_gray = args.adt < 0
    clips_np, clips, pipes, link_streams = [], [], [], []
    for k in range(S):                               # one independent stream per (rank, k)
        video = SyntheticVideo(H, W, args.blobs, seed=rank * S + k, fps=fps_video)
        frames_np = video.frames(F)
        frames = torch.from_numpy(frames_np).to(dev)  # resident in HBM before the timed region
        if args.channels == 3:
            frames = frames.unsqueeze(-1).expand(-1, -1, -1, 3).contiguous()
        clips_np.append(frames_np)
        clips.append(frames)
        pipes.append(TrackingPipeline(H, W, fps_video, settings, batch=B, max_det=args.max_det, capacity=args.capacity,
                                      device=dev, rows_per_flush=F * args.capacity, link=not args.detect_only))
        link_streams.append(torch.cuda.current_stream(dev) if S == 1 else torch.cuda.Stream(device=dev))
    frames_np, pipe = clips_np[0], pipes[0]
    thr_events, chain_events, link_events, enqueue_s = [], [], [], []
    DIAG = int(os.environ.get('YSMR_BENCH_DIAG', '3'))
    # timing events are created (first record) BEFORE the timed region and reused inside it: creating one costs the host
    # tens of microseconds, and the host has only ~1.4x the time it needs to keep this pipeline fed
    n_batches = (F + B - 1) // B
    pool = [[torch.cuda.Event(enable_timing=True) for _ in range(5)] for _ in range(args.steps * n_batches)]
    for evs in pool:
        for e in evs:
            e.record(torch.cuda.current_stream(dev))
    torch.cuda.synchronize()
    pool_i = [0]
    host_calls = {"detect_async": [], "link": [], "reset": []}   # seconds per call of the timed steps (where a slow host loses its time)

    def step(timed):
        # one clip per stream, fresh trackers; detection of batch b+1 (side stream) overlaps the link of
        # batch b; with several streams per GPU their (serial) link chains interleave as well
        pending = [None] * S
        t_host = time.perf_counter()
        for k in range(S):
            with torch.cuda.stream(link_streams[k]):
                tc = time.perf_counter()
                pipes[k].reset()
                if timed: host_calls["reset"].append(time.perf_counter() - tc)
        for f0 in list(range(0, F, B)) + [None]:
            for k in range(S):
                with torch.cuda.stream(link_streams[k]):
                    nxt = None
                    if f0 is not None:
                        probe = timed and k == 0
                        evs = None
                        if probe:
                            evs = pool[pool_i[0] % len(pool)]; pool_i[0] += 1
                        # the threshold kernel's two events ride on its own dispatch (free); the labelling chain's and the
                        # link's are extra packets on their streams (1.5 % end to end when taken around every batch):
                        # every fourth batch is sampled
                        sampled = probe and (pool_i[0] % 4 == 1)
                        tc = time.perf_counter()
                        nxt = (pipes[k].detect_async(clips[k][f0:f0 + B], thr_events if probe else None,
                                                     chain_events if (sampled and DIAG & 1) else None, events=evs,
                                                     frames_ready=False), f0, evs, sampled)   # (the clip is resident in HBM)
                        if timed: host_calls["detect_async"].append(time.perf_counter() - tc)
                    if pending[k] is not None and not args.detect_only:
                        (slot, res, ready), p0, pevs, psampled = pending[k]
                        tc = time.perf_counter()
                        pipes[k].link(slot, res, ready, p0, link_events if (psampled and DIAG & 2) else None, events=pevs)
                        if timed: host_calls["link"].append(time.perf_counter() - tc)
                    pending[k] = nxt
        if timed:
            enqueue_s.append(time.perf_counter() - t_host)

    # The warm-up steps ARE timed steps (same launches, same probes) whose records are thrown away: the first launch that
    # carries timing events switches its hardware queue to profiling, once per queue, and on some hosts of this pool that
    # one call blocks for 40 ms (scripts/host_probe.py, diagnostics.host_call_us: 86 us median, 42 ms maximum, inside
    # the timed region, when warm-up ran without the probes: 159 k frames/s became 76-106 k on such a box)
    # A full collection of CPython's garbage collector walks every tracked object of the process (torch, numpy, pandas
    # modules: ~40 ms on this pool's hosts) and its trigger is an allocation count, so it fell INSIDE the timed region of
    # some command lines and not of others: 159 k frames/s became 76-106 k (diagnostics.host_call_us showed one
    # detect_async call of 42 ms among 80 of 86 us).  Everything alive now is moved out of the collector's sight
    # (gc.freeze), as ysmr_amd.track_eval does around its frame loop; collections of what the steps allocate stay on.
    # YSMR_BENCH_GC=1: leave the collector alone (the measurement of the above).
    import gc
    if os.environ.get("YSMR_BENCH_GC") != "1":
        gc.collect()
        gc.freeze()
    for _ in range(args.warmup):
        step(True)
    torch.cuda.synchronize()
    for rec in (thr_events, chain_events, link_events, enqueue_s, *host_calls.values()):
        rec.clear()
    pool_i[0] = 0
    dist.barrier(info)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    torch.cuda.synchronize()
    own_elapsed = time.perf_counter() - t0          # this rank's K steps, before it waits for the others
    dist.barrier(info)
    torch.cuda.synchronize()
    red = dev if args.dist_backend == "nccl" else "cpu"
    elapsed = dist.max_over_ranks(time.perf_counter() - t0, info, device=red)
    # every rank's own rate (frames it linked / its own time): a straggler GPU or host shows as one low entry
    per_rank_fps = dist.gather_over_ranks(S * F * args.steps / own_elapsed, info, device=red)
    # (how many ranks reached this line with their steps done, and what they did: a SCALE record then shows that N ranks
    # took part, not one rank's figure times N)
    ranks_seen = int(round(dist.sum_over_ranks(1.0, info, device=red)))
    frames_done = int(round(dist.sum_over_ranks(float(S * F * args.steps), info, device=red)))

    # the path must have produced sane output: no overflow/arena flags, no tracker errors, rows
    n_rows = 0
    for p in pipes:
        n_tracks, next_id, err = p.trk.info()
        rows_k = int(p.row_count.item())
        status = max(int(d.status.max().item()) for d in p.det)
        if err or status or (rows_k <= 0 and not args.detect_only):
            raise SystemExit(f"hot path reported errors: tracker={err} detect_status={status} rows={rows_k}")
        n_rows += rows_k

    if rank == 0:
        ms = [e0.elapsed_time(e1) for e0, e1, _ in thr_events]
        px = [b * H * W for _, _, b in thr_events]
        # algorithmic bytes of the fused threshold kernel: 1 B/px read + 1 B/px class map written
        # (3 B/px read for BGR input), per launch of `batch` frames
        # (mean-gray branch: the frame is read twice -- statistics, then blur + compare -- and there are
        # three launches between the events: k_gray_sums, k_mean_levels, k_level_threshold)
        alg_bytes = ((2.0 if mean_gray else 1.0) * args.channels + 1.0) * sum(px) / len(px)
        avg_ms = sum(ms) / len(ms)
        achieved = alg_bytes / (avg_ms * 1e-3) / 1e9
        traffic, traffic_source = None, None
        # (the matrix-pipe kernel serves gray frames whose width is a multiple of 4: thr_mfma.hip, ysmr_thr::supported)
        thr_kernel = "k_threshold_strip" if (pipe.det[0].threshold_variant == 1 or args.channels != 1 or W % 4) else "k_threshold_mfma"
        pmc_name = "threshold_pmc.json" if thr_kernel == "k_threshold_strip" else "threshold_mfma_pmc.json"
        pmc = os.path.join(ROOT, "profiles", pmc_name)
        if os.path.exists(pmc) and not mean_gray:
            try:
                rec = json.load(open(pmc))
                if rec.get("batch") == B and rec.get("height") == H and rec.get("width") == W and rec.get("kernel") == thr_kernel:
                    traffic = rec.get("hbm_bytes_per_launch")
                    traffic_source = f"profiles/{pmc_name} (separate rocprofv3 --pmc run of this kernel and geometry, not this run)"
            except Exception:
                traffic = None
        which = {(1228, 922, 500): "BASELINE configs[2], the configuration the metric is quoted on",
                 (1228, 922, 50): "BASELINE configs[0] geometry and blob count", (3840, 2160, 5000): "BASELINE configs[4]"}.get(
                     (W, H, args.blobs), "custom geometry")
        out = {
            "metric": "frames/sec detect+link, 1228x922 ~500 blobs, 1/2/4/8 GPU; HBM GB/s %peak",
            "value": frames_done / elapsed,
            "unit": "frames/s",
            "n_gpus": world,
            "ranks_seen": ranks_seen,
            "per_rank_frames_per_s": [round(x, 1) for x in per_rank_fps],
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8 image / f32 threshold+geometry / f64 link",
            "data": "synthetic",
            "config": {"workload": f"{W}x{H} stream, ~{args.blobs} blobs, detect+link end to end ({which})"
                                   + (", mean-gray threshold branch (adaptive double threshold < 0)" if mean_gray else "")
                                   + (" -- DETECTION ONLY (BASELINE configs[1]), not the metric's configuration" if args.detect_only else ""),
                       "frames_per_step": F, "detect_batch": B, "channels": args.channels, "streams": world * S, "parallelism": f"{S} stream{'s' if S > 1 else ''}/GPU x{world}",
                       "rows_per_step": n_rows, "tracks_alive": n_tracks, "ids_issued": next_id},
            "roofline": {"kernel": "k_gray_sums+k_mean_levels+k_level_threshold" if mean_gray else thr_kernel, "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                         "avg_launch_ms": avg_ms, "algorithmic_bytes_per_launch": alg_bytes,
                         "launches_timed": len(ms),
                         "timed_with": ("HIP events recorded on the kernels' stream around the three launches" if mean_gray else
                                        "HIP events set by the kernel's own dispatch (hipExtLaunchKernel start/stop): the duration "
                                        "a kernel trace reports, inside the timed region")},
        }
        # what the headline is made of, so that one line says which part was slow on this box: the link of a batch is one
        # launch on one compute unit (k_batch), frames strictly in order, so `value` ~ 1e6 / link.us_per_frame.avg as long
        # as detection (threshold + chain, on the other stream) hides behind it and the host keeps ahead
        diag = {"frames_per_batch": B,
                # (per 64 frames: what earlier rounds' lines, at 64 frames per batch, called "per batch")
                "threshold_us_per_64_frames": sum(ms) / len(ms) * 1e3 * 64.0 / B,
                "components_us_per_64_frames": (sum(e0.elapsed_time(e1) for e0, e1, _ in chain_events) / len(chain_events) * 1e3 * 64.0 / B) if chain_events else None,
                "threshold_us_per_batch": stats_us([m * 1e3 for m in ms]),
                # in launch order (two per step): a launch at twice the others' time had a workgroup wait for a compute unit
                "threshold_us_by_launch": [round(m * 1e3) for m in ms],
                "components_us_per_batch": stats_us([e0.elapsed_time(e1) * 1e3 for e0, e1, _ in chain_events]),
                "link_us_per_frame": stats_us([e0.elapsed_time(e1) * 1e3 / n for e0, e1, n, _ in link_events]),
                "link_host_issue_us_per_frame": stats_us([h * 1e6 / n for _, _, n, h in link_events]),
                "host_enqueue_ms_per_step": sum(enqueue_s) / len(enqueue_s) * 1e3 if enqueue_s else None,
                # per call of the step loop: a host that blocks somewhere shows as a max far above the p50
                "host_call_us": {k: stats_us([x * 1e6 for x in v]) for k, v in host_calls.items() if v},
                "clocks": gpu_clocks(local_rank), "kernargs_in_device_memory": os.environ.get("HIP_FORCE_DEV_KERNARG") == "1",
                "host_cpus": os.cpu_count(),
                "cpus_usable": len(os.sched_getaffinity(0)), "pinned_to_gpu_numa_node": bool(pinned),
                "host_load_1m": os.getloadavg()[0]}
        out["diagnostics"] = diag
        if world == 1:
            out["cpu_baseline"] = cpu_baseline(frames_np, args.cpu_sample, fps_video, args.adt, args.detect_only)
        print(json.dumps(out), flush=True)
    dist.finish(info)


if __name__ == "__main__":
    main()
