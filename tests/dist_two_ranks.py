#!/usr/bin/env python3
"""Run under `python -m torch.distributed.run --nproc-per-node 2` by tests/test_gpu_configs.py: two ranks (sharing
cuda:0 on a one-GPU box; gloo carries the path's single collective) compute the sharded pairing check and the sharded
ONE-product check with the real engine and compare them with the single-rank results."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import zkvm_pairings_amd as z  # noqa: E402
from zkvm_pairings_amd import configs, dist as zd, synthetic  # noqa: E402


def main():
    dist.init_process_group(os.environ.get("ZKP_BENCH_BACKEND", "gloo"))
    rank, world = dist.get_rank(), dist.get_world_size()
    assert world == 2
    local = 0 if os.environ.get("ZKP_BENCH_SHARE_GPU") == "1" else int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    cpu = torch.device("cpu")
    red_dev = dev if dist.get_backend() == "nccl" else cpu      # gloo reduces host tensors
    eng = z.PairingEngine(local)
    n = 3001                                                    # odd: the two blocks differ in size
    g1, g2, _, _ = synthetic.random_pairs(eng, n, seed=1234, device_tensors=True)
    # k = 1 checks: all fail (random pairings); with infinities on rank 1's block only: flags differ per rank
    for case in range(3):
        inf = torch.zeros(n, dtype=torch.uint8, device=dev)
        if case == 1:
            inf[:] = 1                                          # every pairing is the identity: AND true
        if case == 2:
            inf[n // 2 + 5:] = 1                                # rank 1's block passes, rank 0's does not: AND false
        ok_ref, all_ref = eng.pairing_check(g1, g2, 1, inf, None)

        def check_fn(lo, hi):
            _, f = eng.pairing_check(g1[lo:hi].contiguous(), g2[lo:hi].contiguous(), 1, inf[lo:hi].contiguous(), None)
            return f.to(red_dev)

        got = zd.sharded_pairing_check(check_fn, n, red_dev)
        assert got == bool(all_ref.item()) == (case == 1), (case, got)
    # the whole batch as ONE product check: cancelling pairs interleaved so that both blocks are non-trivial
    m = 512
    h1 = g1[:m].cpu().numpy().view(np.uint64)
    G1 = torch.cat([g1[:m], torch.from_numpy(configs.negate_g1(eng, h1).view(np.int64)).to(dev)]).contiguous()
    G2 = torch.cat([g2[:m], g2[:m]]).contiguous()

    def finish(parts):
        prod = eng.fp12_product(parts.to(dev).contiguous())
        gt = eng.final_exponentiation(prod.reshape(1, 72))
        return np.array_equal(gt.cpu().numpy().view(np.uint64)[0], eng.gt_identity())

    mp = lambda lo, hi: eng.miller_product(G1[lo:hi].contiguous(), G2[lo:hi].contiguous()).to(red_dev)
    assert zd.sharded_product_check(mp, finish, 2 * m) is True
    mp_bad = lambda lo, hi: eng.miller_product(G1[lo:hi].contiguous(), G2[lo:hi].contiguous()).to(red_dev) if hi < 2 * m else \
        eng.miller_product(G1[lo:hi - 1].contiguous(), G2[lo:hi - 1].contiguous()).to(red_dev)
    assert zd.sharded_product_check(mp_bad, finish, 2 * m) is False
    single_gt, single_one = eng.pairing_product_check(G1, G2)
    assert int(single_one.item()) == 1
    dist.barrier()
    eng.close()
    if rank == 0:
        print("TWO RANKS OK")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
