"""world_size-2 CPU (gloo) test of the multi-GPU path: ranks own distinct camera streams, no data-path collective,
the bench's barrier / max-over-ranks / whole-job aggregation."""
import os
import socket

import numpy as np
import pytest
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    from oracle import oracle_py as O
    from rebvio_amd import shard, synth
    shard.init_group("gloo", rank, world)
    frames, cam = synth.render_stream(128, 96, 2, stream_id=shard.stream_id_for_rank(rank))
    # each rank runs ITS stream end to end with the checker (the HIP backend needs a GPU; the sharding logic does not)
    orc = O.Oracle(O.default_params(cam.height, cam.width, fm=cam.fm, cx=cam.cx, cy=cam.cy))
    n = orc.detect_u8(frames[0]).size()
    dist.barrier()
    elapsed = 1.0 + rank  # rank 1 is the slow one
    tmax = shard.max_over_ranks(elapsed, world)
    fps = shard.whole_job_fps(world, 100, tmax)
    cpu_sum = shard.sum_over_ranks(70.0 + rank, world)   # N instances of the CPU baseline, one per rank: the sum is reported
    q.put((rank, int(frames.sum()), n, tmax, fps, cpu_sum))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_independent_streams():
    from oracle import oracle_py as O
    O.build()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (r0, sum0, n0, t0, f0, c0), (r1, sum1, n1, t1, f1, c1) = res
    assert c0 == c1 == 141.0
    assert (r0, r1) == (0, 1)
    assert sum0 != sum1                      # different scenes per rank
    assert n0 > 0 and n1 > 0
    assert t0 == t1 == 2.0                   # max over ranks
    assert f0 == f1 == 2 * 100 / 2.0         # whole-job frames / slowest rank


def test_numa_binding_helper(tmp_path):
    """bind_to_gpu_numa_node against a fake sysfs tree: binds to the node's CPUs that this process may use, and leaves
    the affinity alone when the node is unknown or lists none of them."""
    import os
    from rebvio_amd import shard
    allowed = sorted(os.sched_getaffinity(0))
    if len(allowed) < 3:
        pytest.skip("needs at least 3 usable CPUs")
    dev = tmp_path / "bus/pci/devices/0000:c1:00.0"
    dev.mkdir(parents=True)
    node = tmp_path / "devices/system/node/node1"
    node.mkdir(parents=True)
    try:
        (dev / "numa_node").write_text("-1\n")
        assert shard.bind_to_gpu_numa_node("0000:C1:00.0", str(tmp_path)) == -1
        (dev / "numa_node").write_text("1\n")
        (node / "cpulist").write_text("100000-100003\n")  # nothing this process may run on
        assert shard.bind_to_gpu_numa_node("0000:c1:00.0", str(tmp_path)) == -1
        assert sorted(os.sched_getaffinity(0)) == allowed
        (node / "cpulist").write_text(f"{allowed[0]}-{allowed[0]},{allowed[1]},100000\n")
        assert shard.bind_to_gpu_numa_node("0000:c1:00.0", str(tmp_path)) == 1
        assert sorted(os.sched_getaffinity(0)) == allowed[:2]
        assert shard.bind_to_gpu_numa_node("0000:ff:00.0", str(tmp_path)) == -1  # unknown device
    finally:
        os.sched_setaffinity(0, allowed)
