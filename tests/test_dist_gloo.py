"""world_size-2 gloo test of the N>1 path (no GPU): contiguous sharding + the single MIN all-reduce
that implements the cross-GPU AND of the Gt==identity flags."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, ws, port, bad_check, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    from zkvm_pairings_amd import dist as zd
    n = 1001
    seen = []

    def check_fn(lo, hi):
        seen.append((lo, hi))
        ok = 0 if (bad_check is not None and lo <= bad_check < hi) else 1
        return torch.tensor([ok], dtype=torch.int32)

    res = zd.sharded_pairing_check(check_fn, n, torch.device("cpu"))
    t = zd.max_over_ranks(1.0 + rank, torch.device("cpu"))
    q.put((rank, seen[0], res, t))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("bad_check", [None, 3, 1000])
def test_sharded_and_reduce_gloo(bad_check):
    ws = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + (0 if bad_check is None else bad_check % 7 + 1)
    procs = [ctx.Process(target=_worker, args=(r, ws, port, bad_check, q)) for r in range(ws)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=120) for _ in range(ws))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, s0, res0, t0), (r1, s1, res1, t1) = out
    assert s0 == (0, 501) and s1 == (501, 1001)            # contiguous, sizes differ by at most one
    assert res0 == res1 == (bad_check is None)              # AND over ranks == MIN of {0,1}
    assert t0 == t1 == 2.0                                  # max over ranks


def _worker_product(rank, ws, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    from zkvm_pairings_amd import dist as zd

    def miller_product_fn(lo, hi):       # stand-in for engine.miller_product: a recognisable 72-word record
        return torch.arange(72, dtype=torch.int64) * 1000 + lo * 7 + hi

    def finish_fn(parts):
        q.put((rank, parts.tolist()))
        return parts.shape == (ws, 72)

    res = zd.sharded_product_check(miller_product_fn, finish_fn, 1001)
    q.put((rank, res))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_product_check_gathers_every_rank_part_gloo():
    ws = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker_product, args=(r, ws, port, q)) for r in range(ws)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(2 * ws)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = [[j * 1000 + lo * 7 + hi for j in range(72)] for lo, hi in ((0, 501), (501, 1001))]
    parts = [v for _, v in got if isinstance(v, list)]
    flags = [v for _, v in got if isinstance(v, bool)]
    assert parts == [want, want] and flags == [True, True]      # both ranks see both parts, in rank order


def _worker_failing(rank, ws, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    from zkvm_pairings_amd import dist as zd

    def check_fn(lo, hi):
        if rank == 1:
            raise RuntimeError("rank 1's own block failed")
        return torch.ones(1, dtype=torch.int32)

    def miller_product_fn(lo, hi):
        if rank == 1:
            raise RuntimeError("rank 1's own block failed")
        return torch.arange(1, 73, dtype=torch.int64)

    out = []
    for call in (lambda: zd.sharded_pairing_check(check_fn, 1001, torch.device("cpu")),
                 lambda: zd.sharded_product_check(miller_product_fn, lambda parts: bool((parts != 0).any(dim=1).all()), 1001)):
        try:
            out.append(call())
        except RuntimeError as e:
            out.append(str(e))
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def test_a_failing_rank_still_joins_the_collective_gloo():
    """a rank whose own block raises takes part with flag 0 / the zero record: the peer gets False (no hang), the failing rank its error"""
    ws = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker_failing, args=(r, ws, port, q)) for r in range(ws)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(ws))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got[0] == [False, False]
    assert got[1] == ["rank 1's own block failed"] * 2


def test_shard_range_partitions():
    from zkvm_pairings_amd.dist import shard_range
    for n in (0, 1, 7, 8, 1 << 20, (1 << 20) + 5):
        for ws in (1, 2, 3, 8):
            blocks = [shard_range(n, r, ws) for r in range(ws)]
            assert blocks[0][0] == 0 and blocks[-1][1] == n
            assert all(blocks[i][1] == blocks[i + 1][0] for i in range(ws - 1))
            sizes = [b - a for a, b in blocks]
            assert max(sizes) - min(sizes) <= 1


def test_bench_launches_its_own_ranks_when_started_plainly():
    """`python bench.py --gpus N` without torchrun around it (how the driver starts the N = 1 bench): the process becomes a launcher
    before torch or HIP is touched, runs the N ranks under torch.distributed.run as a child on 127.0.0.1, relays rank 0's line alone
    on stdout and returns the ranks' code.  --rank-echo makes the ranks report and leave before any GPU use, so this runs on CPU."""
    import json
    import subprocess
    import sys
    bench = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, bench, "--gpus", "2", "--rank-echo"], capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stderr[-1500:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1 and json.loads(lines[0]) == {"metric": "rank-echo", "world": 2, "gpus": 2, "master": "127.0.0.1"}
    assert "torch.distributed.run" in out.stderr
    # a WORLD_SIZE that contradicts --gpus is still refused (rc 2), and a failing rank's code comes back through the launcher
    out = subprocess.run([sys.executable, bench, "--gpus", "2", "--rank-echo"], capture_output=True, text=True, timeout=60, env=dict(env, WORLD_SIZE="3"))
    assert out.returncode == 2
