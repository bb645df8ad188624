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


@pytest.mark.timeout(300)
def test_bench_two_ranks_stub_fleet_json_line():
    """bench.py's N > 1 code path end to end on the CPU box: two ranks under torch.distributed.run (gloo), a host-only stand-in
    for the fleet (--stub-fleet: computes nothing), the counter all-reduce, ranks_seen, per-rank step times, and the ONE JSON
    line rank 0 prints."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--stub-fleet", "--steps", "6", "--warmup", "1", "--instances", "16",
                        "--groups", "2", "--no-cpu-baseline", "--no-single"], env=env, capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 6 and d["scaling"] == "weak" and d["config"]["groups_per_gpu"] == 2
    assert [x["rank"] for x in d["ranks_seen"]] == [0, 1] and "rehearsal" in d
    assert d["trajectories_per_step"] == 2 * 16 * 17000            # both ranks' robots, every cycle's own count
    assert abs(d["value"] - d["trajectories_per_step"] / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
    pr = d["per_rank_ms_per_step"]
    assert len(pr["all"]) == 2 and pr["min"] <= pr["max"] and abs(pr["max"] - d["ms_per_step"]) < 1e-6  # MAX over ranks is the job's time
    assert d["roofline"]["kernel"] and "in_schedule" in d["roofline"]


def test_bench_refuses_ranks_that_share_a_device(monkeypatch):
    """Two ranks on one device are a rehearsal: without --rehearse-on-one-gpu (or the stand-in) bench.py must stop, not report."""
    import importlib
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    bench = importlib.import_module("bench")

    class P:
        pci_bus_id, pci_domain_id, pci_device_id = 0x43, 0, 0

    class T:
        class cuda:
            @staticmethod
            def get_device_properties(i):
                return P()
    assert bench.device_identity(T, 0, False) == bench.device_identity(T, 1, False) == "0000:43:00"
    assert bench.device_identity(T, 1, True) == "stub:1"
