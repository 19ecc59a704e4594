#!/usr/bin/env python3
"""Headline benchmark: frames/sec of rebvio's edge-detect + edge-track hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One process per GPU, one independent 640x480 camera stream per GPU (the path shards by stream: no data-path
collective, "weak" scaling). A step = one frame pushed through the pipeline: EdgeDetector::detect of frame k on the
detect stream overlapped with the full tracking step of pair (k-2, k-1) on the track stream, u8 frames resident
in HBM. Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md, chip-level parameters)


def pmc_traffic_bytes(kernel: str):
    """HBM bytes per launch of `kernel` from the committed PMC passes of this same command (profiles/rNN_pmc_hbm.json:
    `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` in separate runs). Counters are in KiB; per
    MI355X_MICROARCH.md §HBM FETCH_SIZE reads half of a wide coalesced stream on gfx950 -> doubled; other access
    widths are uncalibrated, so this is an upper estimate for gather-style kernels. None when no pass is committed."""
    path = pmc_profile_path()
    if path is None:
        return None
    try:
        d = json.load(open(path)).get(kernel)
        if not d:
            return None
        f = d.get("FETCH_SIZE", {}).get("mean_KiB_per_launch")
        w = d.get("WRITE_SIZE", {}).get("mean_KiB_per_launch")
        if f is None or w is None:
            return None
        return (2.0 * f + w) * 1024.0
    except Exception:
        return None


def pmc_profile_path():
    return next((q for q in (os.path.join(ROOT, "profiles", f"r{r:02d}_pmc_hbm.json") for r in range(9, 0, -1)) if os.path.exists(q)), None)


def rocprof_stats_path():
    return next((q for q in (os.path.join(ROOT, "profiles", f"r{r:02d}_kernel_stats.csv") for r in range(9, 0, -1)) if os.path.exists(q)), None)


def rocprof_avg_us(kernel: str):
    """Mean duration of `kernel` in the committed `rocprofv3 --kernel-trace --stats` summary of this same command
    (profiles/rNN_kernel_stats.csv, newest round): (us, calls), or None when the file or the kernel is missing."""
    import csv
    import re
    path = rocprof_stats_path()
    if path is None:
        return None
    want = kernel.replace(" ", "")
    try:
        for r in csv.DictReader(open(path)):
            name = re.sub(r"\(.*", "", r["Name"]).replace("rh::", "").replace("void ", "").replace(" ", "")
            if name == want:
                return float(r["AverageNs"]) / 1e3, int(r["Calls"])
    except Exception:
        return None
    return None

CONFIGS = {
    # BASELINE.json configs[1]: the configuration the metric is quoted on
    "c2": dict(width=640, height=480, keylines_ref=15000, keylines_max=16000,
               name="640x480 synthetic stream, ~15k keylines, detect+track+IRLS pose, fp32"),
    # configs[2] (parity-test case; selectable for inspection, never the default bench line)
    "c3": dict(width=1280, height=960, keylines_ref=60000, keylines_max=64000, density=2.2, warmup=150,
               name="1280x960 synthetic stream, ~60k keylines, detect+track+IRLS pose, fp32"),
}


def algorithmic_bytes(kernel: str, P: int, N: int) -> float:
    """Algorithmic (compulsory) HBM bytes of ONE launch: every array element the kernel must read or write
    counted once ('Kernels'). P = pixels, N = keylines."""
    table = {
        "k_rowscan<0>": 5 * P,            # u8 in, fp32 row prefix out
        "k_rowscan<1>": 8 * P,
        "k_rowscan<2>": 16 * P,           # 2 filters x (integral in + row prefix out)
        "k_colscan": 16 * P,              # 2 filters x (in + out); the first pass (1 filter) is 8P
        "k_dog_mag": 16 * P,              # 2 integrals in, DoG + squared gradient out
        "k_keyline_flag": 8 * P + 16 * N,
        "k_keyline_emit": 4 * P + 116 * N,  # dense mask + SoA keylines (the tiled distance field needs no clearing pass)
        "k_front_end_u8": 13 * P,           # u8 gather + 8-byte map in, fp32 frame out
        "k_regularize_ekf": 108 * N,
        "k_join_edges": 40 * N + 8 * N + 130 * N,  # chaining + unit gradient + ~4.07 tile-list entries of 32 B per keyline
        "k_df_build": 340 * N,            # scatter variant: 80 cells x 4 B atomics + 20 B keyline
        "k_df_tiles": 8 * P + 20 * N,     # mask-driven variant (REBVIO_HIP_DF=tiles): mask read once + field written once + geometry
        "k_df_lists": 4 * P + 130 * N,    # keyline-driven default: field written once + ~4.07 list entries of 32 B per keyline
        "k_rotate": 52 * N,
        "k_try_vel": 68 * N,
        # persistent minimizeVel + forwardMatch + extRotVel: the traffic of the launches it replaces
        # ((iterations + 1) x k_try_vel + k_ext_rot_vel); keeping the keyline in registers only lowers
        # what is actually moved, not the algorithmic figure the roofline fraction is quoted on
        "k_lm_chain": 6 * 68 * N + 130 * N,
        "k_ext_rot_vel": 130 * N,
        "k_directed_match_c": 192 * N,    # whole searchMatch in one launch (round 4): head probes and long searches
        "k_regularize": 60 * N,
        "k_depth_ekf": 48 * N,
    }
    base = kernel.split("<")[0] if kernel.startswith(("k_lm_chain", "k_df_tiles", "k_df_lists", "k_directed_match_c")) else kernel
    # same work under other names: the speculative LM kernel, the four-column column pass, the batched (_b) forms
    base = {"k_lm_chain_spec": "k_lm_chain", "k_lm_chain_spec_b": "k_lm_chain", "k_lm_chain_b": "k_lm_chain",
            "k_colscan4": "k_colscan", "k_directed_match_c_b": "k_directed_match_c"}.get(base, base)
    return float(table.get(base, 0))


def launches_per_frame(kernel: str, iterations: int = 5) -> int:
    return {"k_rowscan<2>": 2, "k_colscan": 3, "k_colscan4": 3, "k_try_vel": iterations + 1, "k_rotate": 2}.get(kernel, 1)


# Stages at the reference's REBVIO_TIMER tick sites -> the kernels that do that work here.
STAGES = (
    ("detect", "edge_detector.cpp:31,41", ("k_front_end_u8", "k_rowscan", "k_colscan", "k_dog_mag", "k_keyline_flag",
                                           "k_keyline_emit", "k_join_edges")),
    ("buildDistanceField", "core.cpp:34-36", ("k_df_",)),
    # one persistent launch does minimizeVel (core.cpp:152,187), forwardMatch and extRotVel's sums (core.cpp:193,258)
    ("minimizeVel+forwardMatch+extRotVel", "core.cpp:152,187,193,258", ("k_lm_chain", "k_try_vel", "k_ext_rot_vel", "k_rotate")),
    ("directedMatch", "edge_map.cpp:189,216", ("k_directed_match",)),
    ("regularize1Iter+updateInverseDepth", "edge_map.cpp:220-259, core.cpp:417-456 (untimed in the reference)",
     ("k_regularize", "k_depth_ekf")),
)


def event_pair_overhead_us(n: int = 64) -> float:
    """Median time between two timing events recorded back to back on an idle stream (what an event pair adds to a sample)."""
    import torch
    torch.cuda.synchronize()
    pairs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for e0, e1 in pairs:
        e0.record()
        e1.record()
    torch.cuda.synchronize()
    return float(np.median([e0.elapsed_time(e1) * 1e3 for e0, e1 in pairs]))


def stage_us(per_frame: dict) -> dict:
    out = {}
    for name, site, prefixes in STAGES:
        us = sum(v for k, v in per_frame.items() if k.startswith(prefixes))
        out[name] = {"us_per_frame": round(us, 3), "reference_tick_site": site}
    return out


def percentiles_ms(stamps_s) -> dict:
    """p50 / p99 / mean of the intervals between consecutive frame completions."""
    d = np.diff(np.asarray(stamps_s, np.float64)) * 1e3
    if d.size == 0:
        return {"p50": None, "p99": None, "mean": None}
    return {"p50": float(np.percentile(d, 50)), "p99": float(np.percentile(d, 99)), "mean": float(d.mean())}


def self_launch(n: int, argv) -> int:
    """`python bench.py --gpus N` without a launcher: start N ranks of this script (rank r -> GPU r) BEFORE this process
    makes any GPU call, pass rank 0's JSON line through, fail if any rank fails. (The driver's
    `python -m torch.distributed.run ... bench.py --gpus N` sets WORLD_SIZE itself and never comes here.)"""
    import socket
    import subprocess
    if "--no-cpu-baseline" not in argv:
        build_cpu_oracle()  # once, before the ranks exist: they find it up to date (and would serialise on its lock otherwise)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out0, _ = procs[0].communicate()
    codes = [procs[0].returncode] + [q.wait() for q in procs[1:]]
    if any(codes):
        print(f"bench.py --gpus {n}: rank exit codes {codes}", file=sys.stderr)
        return 1
    sys.stdout.write(out0.decode())
    sys.stdout.flush()
    return 0


def build_cpu_oracle():
    """The -march=native build of the CPU restatement for THIS host (oracle_py.build: file lock around make, real prerequisites,
    the library renamed into place), or None when it cannot be built - the baseline leg then uses the portable build."""
    from oracle import oracle_py as O
    try:
        return O.build(native=True)
    except Exception as e:
        print(f"native CPU oracle not built ({e}); using the portable build", file=sys.stderr)
        return None


def build_cpu_oracle_ranked(rank, world, barrier):
    """N ranks under a launcher: rank 0 builds, the others wait at a barrier and then find the library up to date."""
    path = build_cpu_oracle() if rank == 0 else None
    if world > 1:
        barrier()
        if rank != 0:
            path = build_cpu_oracle()
    return path


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4000)
    ap.add_argument("--warmup", type=int, default=None,
                    help="untimed warm-up frames directly in front of the timed region (default 1000). NOTE: the stream needs ~1000 "
                         "frames until depths / match queues reach their steady state, so when W < 1000 another 1000 - W untimed "
                         "'settle' frames run BEFORE the W warm-up frames (reported as config.settle_frames): the timed window "
                         "always measures the sustained rate, never the faster first frames of a stream")
    ap.add_argument("--config", default="c2", choices=sorted(CONFIGS))
    ap.add_argument("--base-frames", type=int, default=24, help="distinct rendered frames (ping-pong replay)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--lanes", default="4,8",
                    help="extra runs (N=1 only, never the headline): this many camera streams per GPU advanced in lock-step by "
                         "batched launches (rebvio_hip_batch_*); '' or 0 skips them")
    ap.add_argument("--no-host-class", action="store_true")
    ap.add_argument("--no-pcie", action="store_true", help="skip the extra window with every frame handed over from host memory")
    ap.add_argument("--batched-child", type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument("--child-keylines", type=int, default=15000, help=argparse.SUPPRESS)
    ap.add_argument("--control-plane-only", action="store_true",
                    help="rehearsal without a GPU: rank launch, rendezvous, barrier and max-over-ranks only (value = null)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(self_launch(args.gpus, sys.argv[1:]))
    if args.batched_child:
        import torch  # noqa: F401  (same load order as the parent)
        from rebvio_amd import synth
        cfg = CONFIGS[args.config]
        frames, cam = synth.render_stream(cfg["width"], cfg["height"], args.base_frames, density=cfg.get("density", 1.0))
        kw = dict(fm=cam.fm, cx=cam.cx, cy=cam.cy, keylines_ref=cfg["keylines_ref"], keylines_max=cfg["keylines_max"])
        res = batched_run(args.batched_child, frames, cam, cfg, kw, args.base_frames, cfg["width"] * cfg["height"], args.child_keylines)
        print(json.dumps(res), flush=True)
        return

    # The contract is ONE JSON line on stdout. Native libraries write there too (gloo reports its mesh connections on
    # stdout, "[Gloo] Rank 0 is connected to ..."): everything this process prints before the result goes to stderr.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    from rebvio_amd import shard
    rank, local_rank, world = shard.env_ranks()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    import torch
    import torch.distributed as dist

    if args.control_plane_only:
        shard.init_group("gloo", rank, world)
        if world > 1:
            dist.barrier()
        tmax = shard.max_over_ranks(float(rank + 1), world, "cpu")
        # the CPU-baseline leg's build + load path, as every rank of a real run takes it (no GPU needed)
        oracle_ok = 0.0
        if not args.no_cpu_baseline:
            from oracle import oracle_py as O
            path = build_cpu_oracle_ranked(rank, world, dist.barrier)
            try:
                O.Oracle(O.default_params(48, 64), path)
                oracle_ok = 1.0
            except Exception as e:
                print(f"[rank {rank}] CPU oracle not loadable: {e}", file=sys.stderr)
        loaded = shard.sum_over_ranks(oracle_ok, world, "cpu") if world > 1 else oracle_ok
        if rank == 0:
            os.dup2(real_stdout, 1)
            print(json.dumps({"metric": "control plane rehearsal", "value": None, "n_gpus": world, "max_over_ranks": tmax,
                              "cpu_oracle_loaded_ranks": int(loaded)}),
                  flush=True)
            os.dup2(2, 1)
        if world > 1:
            dist.destroy_process_group()
        return

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP backend has no CPU fallback")
    # rehearsal knobs (never set by the driver): run several ranks on ONE card over gloo to exercise the N>1 path
    dev_index = int(os.environ.get("REBVIO_BENCH_DEVICE", local_rank))
    # Control plane of the bench (the data path has no collective): barrier + max over ranks on CPU tensors over gloo, so
    # that no communicator streams share the GPU's hardware queues with the pipeline while it is timed (the
    # runtime's stream -> queue mapping is sensitive to every extra stream in the process). REBVIO_BENCH_BACKEND=nccl puts
    # both on RCCL instead.
    backend_name = os.environ.get("REBVIO_BENCH_BACKEND", "gloo")
    torch.cuda.set_device(dev_index)
    numa_node = -1
    if os.environ.get("REBVIO_BENCH_NUMA", "1") != "0":  # before any thread or pinned buffer of the pipeline exists
        pr = torch.cuda.get_device_properties(dev_index)
        if all(hasattr(pr, a) for a in ("pci_domain_id", "pci_bus_id", "pci_device_id")):
            numa_node = shard.bind_to_gpu_numa_node(f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0")
    shard.init_group(backend_name, rank, world, torch.device("cuda", dev_index))  # barrier + max-time only
    local_rank = dev_index

    from rebvio_amd import backend as B
    from rebvio_amd import synth

    # diagnostic knob (never set by the driver): extra live streams in this process, to see how the runtime's stream -> queue
    # mapping affects the pipeline
    _extra = [torch.cuda.Stream() for _ in range(int(os.environ.get("REBVIO_BENCH_EXTRA_STREAMS", "0")))]
    for _s in _extra:
        with torch.cuda.stream(_s):
            torch.zeros(16, device="cuda").add_(1)
    torch.cuda.synchronize()

    cfg = CONFIGS[args.config]
    W, H = cfg["width"], cfg["height"]
    steps = args.steps
    warmup = max(1000, cfg.get("warmup", 3)) if args.warmup is None else max(0, args.warmup)

    # ---- synthetic stream (one per rank), frames resident in HBM -------------------------------------------
    # scene density: enough texture for the servo to settle at keylines_ref (1280x960 needs a denser scene for ~60k)
    frames, cam = synth.render_stream(W, H, args.base_frames, stream_id=shard.stream_id_for_rank(rank),
                                      density=cfg.get("density", 1.0))
    kw = dict(fm=cam.fm, cx=cam.cx, cy=cam.cy, keylines_ref=cfg["keylines_ref"], keylines_max=cfg["keylines_max"],
              device_id=local_rank)
    if os.environ.get("REBVIO_BENCH_POOL"):  # diagnostic: edge maps allocated up front (the pool grows by itself otherwise)
        kw["map_pool"] = int(os.environ["REBVIO_BENCH_POOL"])
    ctx = B.Context(B.default_params(H, W, **kw))
    dev = ctx.upload_frames(frames)
    npx = W * H
    # The stream needs ~1000 frames until depth filters and match queues reach their steady state (the first frames run faster:
    # few long searches). A run with a short --warmup is first brought there by `settle` untimed frames, so that the timed
    # window measures the SUSTAINED rate whatever W is; the W warm-up frames as given follow, then the timed region.
    settle = max(0, 1000 - warmup)
    # A window of few frames also carries the pipeline's fill and drain (GPU idle at both brackets, ~130 us = 1.6 frames,
    # 8 % of a 20-frame window): runs shorter than 1000 steps are followed by an extra, separately bracketed window of 2000
    # frames whose rate is reported next to the contract's figure (config.long_window), never instead of it.
    long_steps = 2000 if steps < 1000 else 0
    order = synth.pingpong_indices(args.base_frames, settle + warmup + steps + long_steps + 800 + 96)  # GPU leg only; the CPU leg builds its own list

    def push(i):
        return ctx.push_frame_u8_device(dev + int(order[i]) * npx, i * 50000)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    # ---- warm-up (threshold servo settles at keylines_ref; pools, streams, code objects warm) --------------
    k = 0
    last_warm = min(warmup, 32)  # the warm-up frames that run directly in front of the timed region (below)
    for _ in range(settle + warmup - last_warm):
        push(k)
        k += 1
    torch.cuda.synchronize()

    # ---- find the dominant kernel (all-kernel event pass, untimed; BEFORE the warm-up so that the W warm-up frames run
    # right in front of the timed region - reading ~300 event pairs back keeps the host busy and the GPU idle for
    # milliseconds, after which the first frames run a few per cent slower) ------------------------------------
    ctx.profile_reset()
    ctx.profile(True)
    nprof = 24
    kl_counts = []
    for _ in range(nprof):
        out, n = push(k)
        kl_counts.append(n)
        k += 1
    torch.cuda.synchronize()
    prof = ctx.profile_read()
    ctx.profile(False)
    per_frame = {name: avg * calls / nprof for name, (avg, calls) in prof.items()}
    # The event pair around a launch adds its own few microseconds to every sample, which favours kernels launched several
    # times per frame when two candidates are close (k_colscan x 3 against the LM kernel x 1). The choice is made on the
    # per-frame sums with an empty event pair's time taken off every launch; the reported figures stay as measured.
    ev_over = event_pair_overhead_us()
    debiased = {kname: max(0.0, us - ev_over) * cnt / nprof for kname, (us, cnt) in prof.items() if cnt > 0}
    dominant = max(debiased, key=debiased.get)
    n_keylines = int(np.median([c for c in kl_counts if c >= 0])) if kl_counts else 0

    # ---- the last warm-up frames, directly in front of the timed region --------------------------------------------
    for _ in range(last_warm):
        push(k)
        k += 1
    torch.cuda.synchronize()

    # ---- timed region: EXACTLY `steps` frames ------------------------------------------------------------------
    ctx.profile_reset()
    # (no event pair inside the timed region: an event pair is two more packets on the track stream, ~4 us on the chain of the
    # pair it brackets - measured on a 20-frame window: every 2nd launch 11.4 k frames/s, every 8th 12.1 k, none 12.2 k. The
    # dominant kernel's launch time is sampled in a window of its own behind the timed region, below.)
    statuses = []
    matches = []
    push_done = np.zeros(steps, np.float64)
    barrier()
    pairs_before = ctx.pairs_started()
    t0 = time.perf_counter()
    for j in range(steps):
        out, n = push(k)
        push_done[j] = time.perf_counter()  # (a push queues the frame's detect kernels and whatever pairs can start; it blocks only on result slots)
        statuses.append(out.status)
        matches.append(out.klm_num)
        k += 1
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    pairs_in_window = ctx.pairs_started() - pairs_before
    if world > 1:
        dist.barrier()
    elapsed = t1 - t0
    long_elapsed = 0.0
    if long_steps:
        barrier()
        tl0 = time.perf_counter()
        for _ in range(long_steps):
            push(k)
            k += 1
        torch.cuda.synchronize()
        long_elapsed = time.perf_counter() - tl0
        if world > 1:
            dist.barrier()
    # ---- launch time of the dominant kernel: HIP events around every 8th launch (on the stream it runs on) over 800 further
    # frames of the same stream, ~100 samples whatever --steps is; never inside a timed window
    sample_steps = 800
    ctx.profile_reset()
    ctx.profile(True, only=dominant, stride=int(os.environ.get("REBVIO_BENCH_STRIDE", 8)))
    for _ in range(sample_steps):
        push(k)
        k += 1
    torch.cuda.synchronize()
    dom = ctx.profile_read().get(dominant, (0.0, 0))
    ctx.profile(False)
    if world > 1:
        dist.barrier()
    # ---- the same loop with every frame handed over from HOST memory (the reference's contract starts at a host image,
    # rebvio.cpp:38-48): pinned ring + asynchronous copy ahead of the frame's scans. Reported as config.pcie_inclusive_fps,
    # never as `value`.
    pcie_steps = 0 if args.no_pcie else 1500
    pcie_elapsed = 0.0
    if pcie_steps:
        barrier()
        tp0 = time.perf_counter()
        for _ in range(pcie_steps):
            ctx.push_frame_u8(frames[int(order[k % len(order)])], k * 50000)
            k += 1
        torch.cuda.synchronize()
        pcie_elapsed = time.perf_counter() - tp0
        if world > 1:
            dist.barrier()
    ctx.flush()

    tmax = shard.max_over_ranks(elapsed, world, "cuda" if backend_name == "nccl" else "cpu")
    long_tmax = shard.max_over_ranks(long_elapsed, world, "cuda" if backend_name == "nccl" else "cpu") if long_steps else 0.0
    pcie_tmax = shard.max_over_ranks(pcie_elapsed, world, "cuda" if backend_name == "nccl" else "cpu") if pcie_steps else 0.0
    # CPU baseline of an N-stream run: N instances of the restatement x 2 threads, one per rank, at the same
    # time; rank 0 reports the sum. (N = 1: rank 0's own leg below, with the serial figure and the stage split.)
    cpu_multi = None
    if world > 1 and not args.no_cpu_baseline:
        build_cpu_oracle_ranked(rank, world, dist.barrier)  # rank 0 builds, the others find it up to date
        mine = cpu_baseline(frames, cam, cfg, args.base_frames, min(args.cpu_seconds, 10.0), serial_leg=False)
        tot = shard.sum_over_ranks(mine["value"], world, "cuda" if backend_name == "nccl" else "cpu")
        cpu_multi = {"value": tot, "unit": "frames/s", "cores": 2 * world, "kind": "port",
                     "sample": f"{world} instances of the CPU restatement at the same time, one per rank on its own stream, detect || track "
                               f"on 2 threads each; rank 0's instance: {mine['sample']}",
                     "rank0": {k: mine[k] for k in ("value", "frames_run", "seconds", "keylines")}}
    bad = sum(1 for s in statuses if s not in (0, -1))  # -1: no finished pair to report in that call
    if bad:
        print(f"[rank {rank}] WARNING: {bad} of {steps} frame pairs ended with a non-zero tracking status", file=sys.stderr)

    if rank == 0:
        fps = shard.whole_job_fps(world, steps, tmax)
        dom_us = dom[0]
        ab = algorithmic_bytes(dominant, npx, n_keylines)
        if dominant in ("k_colscan", "k_colscan4"):
            ab = (8 * npx + 16 * npx + 16 * npx) / 3.0  # mean over its three launches per frame (1, 2, 2 filters)
        achieved = ab / (dom_us * 1e-6) / 1e9 if dom_us > 0 else 0.0
        rp = rocprof_avg_us(dominant)
        result = {
            "metric": "frames/sec at 640x480 (~15k keylines); 1-GPU and 8-stream/8-GPU batch",
            "value": fps,
            "unit": "frames/s",
            "n_gpus": world,
            "steps": steps,
            "warmup": warmup,
            "ms_per_step": tmax / steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": cfg["name"], "streams": world, "keylines": n_keylines,
                       "mean_matches": float(np.mean([m for m in matches if m > 0])) if any(m > 0 for m in matches) else 0.0,
                       "frames_in_hbm": args.base_frames, "settle_frames": settle,
                       # a step = one frame pushed: its detection + the tracking of an earlier pair. Pairs are queued in groups, so
                       # a window may start a few pairs more or fewer than it pushes frames (the difference is frames waiting in the
                       # queue at its ends); over the long window the two are equal
                       "pairs_tracked_in_window": pairs_in_window,
                       "long_window": ({"steps": long_steps, "value": shard.whole_job_fps(world, long_steps, long_tmax), "unit": "frames/s",
                                        "note": "same loop, bracketed the same way, right after the timed region: the rate without "
                                                "the weight a short window gives to pipeline fill and drain"} if long_steps else None),
                       "parallelism": f"{world} independent streams, 1 per GPU",
                       "rank0_numa_node": numa_node,
                       "pcie_inclusive_fps": (shard.whole_job_fps(world, pcie_steps, pcie_tmax) if pcie_steps else None),
                       "pcie_inclusive_note": ("every frame handed over from host memory (rebvio_hip_push_frame_u8: pinned ring + async copy), "
                                               f"{pcie_steps} frames right after the timed region; `value` is with frames resident in HBM"
                                               if pcie_steps else None),
                       "pose_tolerance": "keyline set/order and every per-keyline output bit-exact vs the CPU restatement (parity "
                                         "unpinned: the reference holds no vector for this path); pair velocity within 1e-2 (rel.; "
                                         "observed <= 5.5e-3) of the restatement run with double-accumulated sums while the LM decisions "
                                         "agree - the sequential-fp32 restatement itself is 0.25-4.8 % from that "
                                         "(tests/test_parity_gpu.py::test_stream_divergence_report)"},
            "frame_ms": percentiles_ms(push_done),
            "stage_us": stage_us(per_frame),
            "roofline": {"bound": "hbm", "kernel": dominant, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": pmc_traffic_bytes(dominant),
                         "traffic_source": (f"{os.path.relpath(pmc_profile_path(), ROOT)}: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE passes of "
                                            "this command (separate runs, committed), 2 x FETCH_SIZE + WRITE_SIZE per launch - NOT measured in "
                                            "this run" if pmc_profile_path() else None),
                         "avg_launch_us": dom_us, "launches": dom[1], "algorithmic_bytes_per_launch": ab,
                         "avg_launch_us_basis": "HIP events around every 8th launch on the kernel's stream, in a window of 800 frames behind "
                                                "the timed region (an event pair brackets the dispatch as well: a few us above rocprofv3's figure)",
                         "rocprof_avg_launch_us": (rp[0] if rp else None), "rocprof_launches": (rp[1] if rp else None),
                         "frac_rocprof": ((ab / (rp[0] * 1e-6) / 1e9 / HBM_PEAK_GBS) if rp else None),
                         "rocprof_source": (f"{os.path.relpath(rocprof_stats_path(), ROOT)}: rocprofv3 --kernel-trace --stats of this command, "
                                            "committed - NOT measured in this run" if rocprof_stats_path() else None),
                         "frame_algorithmic_bytes": 112 * npx + 1740 * n_keylines,
                         "frame_achieved_GBs": (112 * npx + 1740 * n_keylines) * (fps / world) / 1e9},
            "kernel_us_per_frame": {kname: round(v, 3) for kname, v in sorted(per_frame.items(), key=lambda kv: -kv[1])},
            # per kernel: algorithmic bytes of one launch; its mean duration in the all-kernel event pass ("hip_event_us": 24 frames,
            # every launch bracketed by a HIP event pair on its stream, ~2-4 us above rocprofv3's figure for the short kernels) and in
            # the committed rocprofv3 summary of this command ("rocprof_us"); fraction of the HBM peak on either basis
            "kernel_roofline": kernel_roofline_table(prof, npx, n_keylines),
        }
        if cpu_multi is not None:
            result["cpu_baseline"] = cpu_multi
        if world == 1 and args.lanes not in ("", "0"):
            # Each batched run gets a process of its own: the HIP runtime maps streams onto a pool of hardware queues that
            # outlives the streams, and the batch's three streams must not end up sharing queues with this process's earlier
            # ones (measured: the same batch ran at half its rate after the single-stream context had lived here).
            ctx.close()
            torch.cuda.synchronize()
            import subprocess
            runs = []
            for b in [int(x) for x in args.lanes.split(",") if int(x) > 0]:
                r = subprocess.run([sys.executable, os.path.abspath(__file__), "--batched-child", str(b), "--config", args.config,
                                    "--base-frames", str(args.base_frames), "--child-keylines", str(n_keylines)],
                                   capture_output=True, text=True)
                line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
                if r.returncode == 0 and line:
                    runs.append(json.loads(line[-1]))
                else:
                    print(f"batched run with {b} lanes failed: {r.stderr[-800:]}", file=sys.stderr)
            result["streams_per_gpu"] = runs
        if world == 1 and args.config == "c2" and not args.no_host_class:
            try:
                hc = host_class_rate(frames, cam, cfg)
                result["config"]["host_class_fps"] = hc["fps"]
                result["config"]["host_class"] = hc
                result["config"]["host_class_note"] = ("rebvio::Rebvio (the reference's C++ class: camera + 200 Hz IMU, gyro prior, scale / "
                                                       "attitude / bias filter per pair, three host threads) replaying the same stream "
                                                       "through rebvio_replay; wall clock, first to last odometry record")
            except Exception as e:  # the extra figure must never cost the line
                print(f"host class rate not measured: {e}", file=sys.stderr)
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(frames, cam, cfg, args.base_frames, args.cpu_seconds)
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(json.dumps(result), flush=True)
        os.dup2(2, 1)  # process-group teardown may log again
    if world > 1:
        dist.destroy_process_group()


def kernel_roofline_table(prof, npx, n_keylines):
    out = {}
    for kname, (us, calls) in sorted(prof.items(), key=lambda kv: -kv[1][0] * kv[1][1]):
        ab = algorithmic_bytes(kname, npx, n_keylines)
        if kname in ("k_colscan", "k_colscan4"):
            ab = (8 * npx + 16 * npx + 16 * npx) / 3.0
        if not ab or us <= 0:
            continue
        gbs = ab / (us * 1e-6) / 1e9
        rp = rocprof_avg_us(kname)
        out[kname] = {"algorithmic_bytes": ab, "hip_event_us": round(us, 3), "launches": calls, "achieved_GBs": round(gbs, 1),
                      "frac": round(gbs / HBM_PEAK_GBS, 5),
                      "rocprof_us": (round(rp[0], 3) if rp else None),
                      "frac_rocprof": (round(ab / (rp[0] * 1e-6) / 1e9 / HBM_PEAK_GBS, 5) if rp else None)}
    return out


def host_class_rate(frames, cam, cfg, n=4000):
    """Frames/s of the drop-in C++ class rebvio::Rebvio (full camera + IMU fusion) on the bench's stream: rebvio_replay over
    a raw file; wall-clock rate between the first and the last odometry record (process start excluded), with the per-pair
    timers of the fusion thread and the per-frame time of the acquisition thread next to it (REBVIO_HOST_TIMERS)."""
    import re
    import subprocess
    import tempfile
    from rebvio_amd import synth
    exe = os.path.join(ROOT, "rebvio_amd", "_build", "rebvio_replay")
    with tempfile.TemporaryDirectory() as d:
        order = synth.pingpong_indices(len(frames), n)
        frames[order].tofile(os.path.join(d, "f.u8"))
        ts, gyro, acc = synth.imu_samples(synth.make_scene(0), n, noise_seed=1)
        rec = np.zeros(len(ts), dtype=[("ts", "<i8"), ("gyro", "<f4", 3), ("acc", "<f4", 3)])
        rec["ts"], rec["gyro"], rec["acc"] = ts, gyro * 0, acc  # ping-pong replay: a still gyro
        rec.tofile(os.path.join(d, "imu.bin"))
        # three runs, the median one is reported (a run is 0.35 s of three host threads beside a GPU-bound chain: one preemption of
        # the fusion thread shows as hundreds of frames/s; all three rates are listed)
        runs = []
        for _ in range(3):
            r = subprocess.run([exe, "--raw", os.path.join(d, "f.u8"), "--size", str(cam.width), str(cam.height), "--imu", os.path.join(d, "imu.bin"),
                                "--camera", str(cam.fm), str(cam.cx), str(cam.cy), "--keylines", str(cfg["keylines_ref"]), str(cfg["keylines_max"]),
                                "--out", os.path.join(d, "o.txt")], capture_output=True, text=True, env=dict(os.environ, REBVIO_HOST_TIMERS="1"),
                               timeout=300)
            w = re.search(r"\[replay\] (\d+) odometry records in ([0-9.]+) s = ([0-9.]+) frames/s", r.stderr)
            if r.returncode != 0 or not w:
                raise RuntimeError(r.stderr[-400:])
            runs.append((float(w.group(3)), r))
        runs.sort(key=lambda t: t[0])
        all_fps = [t[0] for t in runs]
        r = runs[1][1]
    m = re.search(r"per pair \(us\): first half on device ([0-9.]+)\s+acceleration \+ bias/scale filter ([0-9.]+)\s+second half on device "
                  r"([0-9.]+)\s+pose \+ callbacks ([0-9.]+)\s+\(between pairs: input queue \+ IMU read ([0-9.]+)\)", r.stderr)
    w = re.search(r"\[replay\] (\d+) odometry records in ([0-9.]+) s = ([0-9.]+) frames/s", r.stderr)
    a = re.search(r"EdgeDetector::detect \(staging \+ enqueue\) ([0-9.]+) us per frame", r.stderr)
    if r.returncode != 0 or not m or not w:
        raise RuntimeError(r.stderr[-400:])
    st = [float(v) for v in m.groups()]
    return {"fps": float(w.group(3)), "runs_fps": all_fps,
            "basis": "wall clock between the first and the last odometry record, %d records; median of three runs" % int(w.group(1)),
            "fusion_thread_us_per_pair": {"first_half_on_device": st[0], "acceleration_bias_scale_filter": st[1], "second_half_on_device": st[2],
                                          "pose_callbacks": st[3], "between_pairs": st[4]},
            "acquisition_thread_us_per_frame": float(a.group(1)) if a else None}


def batched_run(lanes, frames0, cam, cfg, kw, base_frames, npx, n_keylines, steps=1200, warmup=800):
    """`lanes` independent camera streams on ONE GPU advanced in lock-step (rebvio_hip_batch_*: every kernel of the frame path
    launched once per step for all lanes). An EXTRA figure next to the single-stream headline: one 640x480 stream keeps under
    a tenth of the chip busy (its kernels are latency chains), batching is how the rest of it is used."""
    import torch
    from rebvio_amd import backend as B
    from rebvio_amd import synth
    kwb = {k: v for k, v in kw.items() if k != "map_pool"}
    bat = B.Batch(B.default_params(cam.height, cam.width, **kwb), lanes)
    devs = []
    for lane in range(lanes):  # lane l sees camera stream l (its own scene)
        fr = frames0 if lane == 0 else synth.render_stream(cam.width, cam.height, base_frames, stream_id=lane, density=cfg.get("density", 1.0))[0]
        devs.append(bat.lanes[lane].upload_frames(fr))
    order = synth.pingpong_indices(base_frames, warmup + steps + 64)
    k = 0
    for _ in range(warmup):
        bat.push_u8_device([d + int(order[k]) * npx for d in devs], k * 50000)
        k += 1
    torch.cuda.synchronize()
    c0 = bat.lanes[0]
    c0.profile_reset()
    c0.profile(True, only="k_lm_chain*", stride=8)  # k_lm_chain_spec_b<512>, or k_lm_chain_b<512> under REBVIO_HIP_LM=seq
    bad = 0
    t0 = time.perf_counter()
    for _ in range(steps):
        outs, _n = bat.push_u8_device([d + int(order[k]) * npx for d in devs], k * 50000)
        bad += sum(1 for lane in range(lanes) if outs[lane].status not in (0, -1))
        k += 1
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    prof = c0.profile_read()
    dom_name = next((kname for kname in prof if kname.startswith("k_lm_chain")), "k_lm_chain_spec_b<512>")
    dom = prof.get(dom_name, (0.0, 0))
    c0.profile(False)
    bat.flush()
    bat.close()
    ab = lanes * algorithmic_bytes("k_lm_chain", npx, n_keylines)
    ach = ab / (dom[0] * 1e-6) / 1e9 if dom[0] > 0 else 0.0
    fps = lanes * steps / (t1 - t0)
    return {"lanes": lanes, "value": fps, "unit": "frames/s (all lanes)", "us_per_step": (t1 - t0) / steps * 1e6, "steps": steps,
            "non_zero_statuses": bad,
            "roofline": {"bound": "hbm", "kernel": dom_name, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": ach / HBM_PEAK_GBS, "avg_launch_us": dom[0], "launches": dom[1], "algorithmic_bytes_per_launch": ab},
            "frame_achieved_GBs": (112 * npx + 1740 * n_keylines) * fps / 1e9}


def cpu_baseline(frames, cam, cfg, base_frames, seconds, serial_leg=True):
    """The CPU oracle (faithful port of the reference path: the reference itself cannot be built here) timed on this
    host: 2 threads per stream like the reference's detect/track workers (rebvio.cpp:28-29) = `value`, plus the serial
    1-thread figure, p50/p99 frame time and ms per stage at the reference's REBVIO_TIMER tick sites.
    Bounded sample of the same stream: `seconds` of CPU work split 2:1 between the two legs; the index list is the
    leg's own, and every rate is frames actually run / seconds measured."""
    from oracle import oracle_py as O
    from rebvio_amd import synth
    path = build_cpu_oracle()  # -O3 -march=native for the machine the bench runs on (a no-op when a rank built it already)
    kw = dict(fm=cam.fm, cx=cam.cx, cy=cam.cy, keylines_ref=cfg["keylines_ref"], keylines_max=cfg["keylines_max"])
    p = O.default_params(cam.height, cam.width, **kw)
    probe_idx = synth.pingpong_indices(base_frames, 24)
    try:
        probe = O.Oracle(p, path).run_stream(frames, probe_idx, threads=2)
    except OSError as e:  # the native library could not be loaded (built for another host, or damaged): the portable build
        print(f"native CPU oracle not loadable ({e}); using the portable build", file=sys.stderr)
        path = None
        probe = O.Oracle(p, path).run_stream(frames, probe_idx, threads=2)
    per = probe["seconds"] / probe["frames"]
    n2 = int(min(max(seconds * (2.0 / 3.0) / per, 30), 1000))
    idx2 = synth.pingpong_indices(base_frames, n2)
    res = O.Oracle(p, path).run_stream(frames, idx2, threads=2)
    skip = min(20, res["frames"] // 2)  # servo warm-up of the sample (keyline count not settled yet)
    if not serial_leg:
        return {"value": res["frames"] / res["seconds"], "unit": "frames/s", "cores": 2, "kind": "port",
                "sample": f"first {res['frames']} frames of the rank's {cam.width}x{cam.height} stream (g++ -O3 -march=native -ffp-contract=off), "
                          f"{os.cpu_count()} host cpus visible",
                "frames_run": int(res["frames"]), "seconds": res["seconds"], "keylines": int(np.median(res["keyline_counts"][skip:]))}
    n1 = max(20, n2 // 4)  # the serial leg costs ~2x per frame: a quarter of the frames = the remaining third of the budget
    idx1 = synth.pingpong_indices(base_frames, n1)
    res1 = O.Oracle(p, path).run_stream(frames, idx1, threads=1)
    stage_ms = {k: v / res1["frames"] * 1e3 for k, v in res1["stage_seconds"].items()}
    return {"value": res["frames"] / res["seconds"], "unit": "frames/s", "cores": 2, "kind": "port",
            "sample": f"first {res['frames']} frames of the same {cam.width}x{cam.height} stream, detect || track on 2 threads "
                      f"(g++ -O3 -march=native -ffp-contract=off), {os.cpu_count()} host cpus visible; serial leg: first "
                      f"{res1['frames']} frames on 1 thread",
            "frames_run": int(res["frames"]), "seconds": res["seconds"],
            "keylines": int(np.median(res["keyline_counts"][skip:])),
            "frame_ms": percentiles_ms(res["frame_done_s"][skip:]),
            "one_thread": {"value": res1["frames"] / res1["seconds"], "unit": "frames/s", "cores": 1,
                           "frames_run": int(res1["frames"]), "seconds": res1["seconds"],
                           "frame_ms": percentiles_ms(res1["frame_done_s"][min(20, res1["frames"] // 2):]),
                           "stage_ms_per_frame": {k: round(v, 4) for k, v in stage_ms.items()}}}


if __name__ == "__main__":
    main()
