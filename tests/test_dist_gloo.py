"""world_size-2 test (gloo, CPU) of the multi-GPU plumbing in bench.py: batch sharding with no
data-path collective, and the max-over-ranks step time."""
import os
import socket
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench

    lo, hi = bench.shard_batch(world, world, rank)
    lo5, hi5 = bench.shard_batch(5, world, rank)
    t = bench.aggregate_time(0.010 * (rank + 1), world)  # rank 1 is slower -> 0.020
    out.put((rank, lo, hi, lo5, hi5, t))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_and_aggregate_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (r0, lo0, hi0, a0, b0, t0), (r1, lo1, hi1, a1, b1, t1) = res
    assert (lo0, hi0, lo1, hi1) == (0, 1, 1, 2)          # one batch element per GPU
    assert (a0, b0, a1, b1) == (0, 3, 3, 5)              # ragged split covers every element once
    assert abs(t0 - 0.020) < 1e-9 and abs(t1 - 0.020) < 1e-9  # job time = slowest rank


def test_algorithmic_bytes_match_survey():
    import bench

    assert bench.algorithmic_bytes("snapkv", 131072, 0.5)["total"] == 805306368
    assert bench.algorithmic_bytes("knorm", 32768, 0.5)["total"] == 201326592       # config 2
    assert bench.algorithmic_bytes("ea", 131072, 0.7)["total"] == 1932730368        # config 4
    assert bench.algorithmic_bytes("ea", 131072, 0.7)["n_kept"] == 39321
