"""N>1 path on CPU: world_size-2 gloo run of the fleet sharding + counter all-reduce that bench.py
uses over RCCL (static shard, no data-path collective)."""
import os
import socket

import pytest


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_total, q):
    import torch.distributed as dist
    from navigation_amd.sharding import reduce_counters, shard_range
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    first, count = shard_range(n_total, rank, world)
    # each "robot" i scores 1000 + i trajectories; rank r takes 0.01 * (r + 1) s
    scored = sum(1000 + i for i in range(first, first + count))
    elapsed, (tot_scored, tot_robots) = reduce_counters(dist, 0.01 * (rank + 1), [scored, count])
    dist.barrier()
    q.put((rank, first, count, elapsed, tot_scored, tot_robots))
    dist.destroy_process_group()


def test_shard_ranges_partition_the_fleet():
    from navigation_amd.sharding import owner_of, shard_range
    for n_total in (1, 7, 8, 255, 256, 2048, 2049):
        for world in (1, 2, 3, 8):
            seen = []
            for r in range(world):
                f, c = shard_range(n_total, r, world)
                seen.extend(range(f, f + c))
                for i in range(f, f + c):
                    assert owner_of(i, n_total, world) == r
            assert seen == list(range(n_total))


@pytest.mark.timeout(120)
def test_gloo_world2_counter_allreduce():
    import torch.multiprocessing as mp
    world, n_total = 2, 257
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_total, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=100) for _ in range(world))
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    assert [o[2] for o in out] == [129, 128] and out[0][1] == 0 and out[1][1] == 129
    for o in out:
        assert abs(o[3] - 0.02) < 1e-12                     # MAX over ranks
        assert o[4] == sum(1000 + i for i in range(n_total))  # SUM over ranks
        assert o[5] == n_total


@pytest.mark.timeout(240)
def test_gloo_world8_configs3_shape():
    """configs[3]'s shape on CPU: 2048 robots over 8 ranks (gloo), every rank 256, counters summed, MAX elapsed."""
    import torch.multiprocessing as mp
    world, n_total = 8, 2048
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_total, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=200) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert [o[2] for o in out] == [256] * 8 and [o[1] for o in out] == [256 * r for r in range(8)]
    for o in out:
        assert abs(o[3] - 0.08) < 1e-12 and o[4] == sum(1000 + i for i in range(n_total)) and o[5] == n_total


def test_bench_refuses_a_world_size_that_disagrees_with_gpus():
    """`bench.py --gpus N` under a launcher with a different WORLD_SIZE must fail loudly, before it touches torch or HIP."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True, timeout=60)
    assert r.returncode == 2 and "WORLD_SIZE=3" in r.stderr and r.stdout.strip() == ""
