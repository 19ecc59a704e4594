"""CPU tests of bench.py's own logic (no GPU): the `--gpus N` self-launcher over gloo, the WORLD_SIZE check, the CPU
baseline leg (rate = frames actually run / seconds, independent of --steps / --warmup), stage and percentile helpers."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env_without_ranks():
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return env


def test_gpus_n_without_launcher_starts_n_ranks():
    """`python bench.py --gpus 2` with WORLD_SIZE unset must start 2 ranks itself (here: control plane only, gloo)."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--control-plane-only"], env=_env_without_ranks(),
                       capture_output=True, timeout=300)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1, lines  # ONE JSON line, nothing else on stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2
    assert d["max_over_ranks"] == 2.0  # rank 1 reports 2.0: the maximum crossed the process boundary
    # every rank went through the CPU-baseline leg's build + load path (built once before the ranks exist, found up to date by
    # the ranks; the file lock and the rename into place keep a concurrent loader from mapping a half-written library)
    assert d["cpu_oracle_loaded_ranks"] == 2


def test_ranks_under_a_launcher_build_the_cpu_oracle_once():
    """The driver's form (`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`): no self-launcher runs, so the
    ranks meet the build themselves - rank 0 builds, a barrier, the others find the library up to date. Started here from a
    clean native directory, twice in a row (the second run finds everything built)."""
    import glob
    import shutil
    import socket
    for d in glob.glob(os.path.join(ROOT, "oracle", "_build", "native-*")):
        shutil.rmtree(d)
    for _ in range(2):
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        procs = []
        for r in range(2):
            env = dict(_env_without_ranks(), RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
            procs.append(subprocess.Popen([sys.executable, BENCH, "--gpus", "2", "--control-plane-only"], env=env, stdout=subprocess.PIPE,
                                          stderr=subprocess.PIPE))
        outs = [q.communicate(timeout=600) for q in procs]
        assert all(q.returncode == 0 for q in procs), [o[1].decode()[-800:] for o in outs]
        d = json.loads([ln for ln in outs[0][0].decode().splitlines() if ln.strip()][-1])
        assert d["cpu_oracle_loaded_ranks"] == 2
    built = glob.glob(os.path.join(ROOT, "oracle", "_build", "native-*", "librebvio_oracle.so"))
    assert len(built) == 1 and not glob.glob(os.path.join(ROOT, "oracle", "_build", "native-*", "*.tmp.*"))


def test_world_size_mismatch_is_an_error():
    env = _env_without_ranks()
    env.update(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "8", "--control-plane-only"], env=env, capture_output=True, timeout=120)
    assert r.returncode != 0
    assert b"n_gpus" not in r.stdout  # never a line that claims 1 GPU for --gpus 8


def test_cpu_baseline_counts_the_frames_it_ran(orc_mod):
    import bench
    from rebvio_amd import synth
    frames, cam = synth.render_stream(192, 144, 8)
    cfg = dict(keylines_ref=1500, keylines_max=2000)
    res = bench.cpu_baseline(frames, cam, cfg, 8, 1.0)
    assert res["kind"] == "port" and res["cores"] == 2
    assert res["frames_run"] >= 30
    assert abs(res["value"] - res["frames_run"] / res["seconds"]) <= 1e-9 * res["value"]
    assert f"first {res['frames_run']} frames" in res["sample"]
    one = res["one_thread"]
    assert one["cores"] == 1 and abs(one["value"] - one["frames_run"] / one["seconds"]) <= 1e-9 * one["value"]
    st = one["stage_ms_per_frame"]
    assert set(st) == {"detect", "buildDistanceField", "minimizeVel", "extRotVel", "directedMatch", "other_track"}
    assert all(v >= 0 for v in st.values()) and st["detect"] > 0 and st["minimizeVel"] > 0
    assert res["frame_ms"]["p50"] > 0 and res["frame_ms"]["p99"] >= res["frame_ms"]["p50"]

    def rates_fit(r, r2):
        o = r["one_thread"]
        return (0.7 * o["value"] <= r["value"] <= 2.6 * o["value"]  # two workers: not slower than ~the serial run, not > 2x faster
                and 0.8 <= sum(o["stage_ms_per_frame"].values()) / (1e3 / o["value"]) <= 1.05  # the stage timers cover the serial frame
                # a second call with the same budget reproduces the rate (the 11x error of round 1 came from a foreign index list)
                and abs(r2["value"] - r["value"]) <= 0.35 * r["value"])

    # one-second samples on shared host cores: a neighbour's burst can bend one of them - measured again once before it counts
    res2 = bench.cpu_baseline(frames, cam, cfg, 8, 1.0)
    if not rates_fit(res, res2):
        res, res2 = bench.cpu_baseline(frames, cam, cfg, 8, 1.0), bench.cpu_baseline(frames, cam, cfg, 8, 1.0)
    assert rates_fit(res, res2), (res["value"], res["one_thread"]["value"], res2["value"], res["one_thread"]["stage_ms_per_frame"])


def test_stage_map_covers_every_kernel_name():
    import bench
    names = ["k_front_end_u8", "k_rowscan<0>", "k_rowscan<1>", "k_rowscan<2>", "k_colscan", "k_dog_mag", "k_keyline_flag",
             "k_keyline_emit", "k_join_edges", "k_df_tiles<32>", "k_df_bin", "k_lm_chain<512>", "k_directed_match_c<512,8>",
             "k_directed_match_c_b<64,1>", "k_regularize_ekf", "k_rotate"]
    per = {n: 1.0 for n in names}
    st = bench.stage_us(per)
    assert sum(v["us_per_frame"] for v in st.values()) == pytest.approx(len(names))
    assert st["detect"]["us_per_frame"] == 9.0
    assert st["buildDistanceField"]["us_per_frame"] == 2.0


def test_percentiles():
    import bench
    p = bench.percentiles_ms(np.arange(101) * 1e-3)
    assert p["p50"] == pytest.approx(1.0) and p["p99"] == pytest.approx(1.0)
    assert bench.percentiles_ms([0.0])["p50"] is None
