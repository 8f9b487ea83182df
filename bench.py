#!/usr/bin/env python3
"""bench.py -- BLS12-381 pairings/s on synthetic random (G1,G2) pairs (BASELINE.json metric).

One process per GPU (torchrun sets RANK/LOCAL_RANK/WORLD_SIZE).  The workload is BASELINE.json's: ONE batch of
2^20 random (G1,G2) pairs, sharded in contiguous blocks over the ranks (strong scaling: 2^20 pairs on 1 GPU,
2^17 per GPU on 8).  A step = one pass of the hot path over this rank's resident shard: fused Miller loop + final
exponentiation -> Gt (bit-exact vs the CPU oracle on a sample) + the Gt==identity flags, then - when there is more
than one rank - ONE all-reduce(MIN) of the per-rank AND flag over RCCL (the only collective the path has).  Inputs
are generated on the GPU before the timed region and stay resident in HBM.

Launch: `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...` (one rank per GPU), or plainly
`python bench.py --gpus N ...`: without WORLD_SIZE in the environment the process becomes a launcher - BEFORE it imports torch or touches
HIP - that starts exactly that torchrun command as a CHILD process on 127.0.0.1, relays rank 0's JSON line and returns the child's code.

--collective torch (default): the step's all-reduce goes through torch.distributed (backend nccl = RCCL).
--collective abi: the step is ONE call of zkp_pairing_gt_check_batch_allreduce_dev per rank - the library's own RCCL communicator
(zkp_comm_init_rank, the id handed round through a gloo group), i.e. the path a Rust / C host takes is the one that is timed."""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

GLOBAL_PAIRS = 1 << 20
# algorithmic work per pairing with the reference-shaped tower = 21,869 Fp-mul-equivalents (10,140 in
# the Miller loop, 11,116 + 613 for the one inversion in the final exponentiation) x 300 32x32->64 multiply-adds of a
# 12-limb CIOS multiply
FPMUL_MILLER, FPMUL_FEXP = 10140, 11116 + 613
# of the Miller loop's 10,140: the G2 doubling / addition steps with their line coefficients (63 x (3 mul + 8 sqr in Fp2 + 2
# scalings) + 5 x (7 mul + 8 sqr + 2 scalings), Fp2 mul = 4 and sqr = 2 Fp mul in the reference-shaped count) - the part
# k_prep_lines computes; the rest is the Fp12 accumulator (line products and squarings): the Miller step program
FPMUL_LINES = 63 * (3 * 4 + 8 * 2 + 4) + 5 * (7 * 4 + 8 * 2 + 4)
MACS_PER_FPMUL = 300
MACS_PER_PAIRING = (FPMUL_MILLER + FPMUL_FEXP) * MACS_PER_FPMUL
# BASELINE config 4: a 3-pair check = Miller(3 pairs) 21,492 + one final exponentiation 11,729 Fp-mul-equivalents
FPMUL_CHECK3 = 21492 + 11729
# BASELINE config 5: is_valid = on-curve + torsion test.  The reference's affine chains cost ~520 inversions per point
#  - no yardstick; the algorithmic count here is the inversion-free Jacobian form of the same tests with
# the reference-shaped tower (Fp2 mul = 4, sqr = 2 Fp mul): doubling 2M + 5S, mixed addition 7M + 4S.
#   G1: -[x^2]P == (beta x, y): two chains of 63 doublings + 5 additions (7 and 11 Fp mul) + curve equation, endomorphism, comparison
#   G2: psi(P) == [x]P: one chain of 63 doublings + 5 additions (18 and 36 Fp mul) + curve equation, psi, comparison
FPMUL_G1_VALID = 2 * (63 * 7 + 5 * 11) + 3 + 1 + 6
FPMUL_G2_VALID = (63 * 18 + 5 * 36) + 8 + 8 + 16
# measured on MI355X (tools/ubench_valu.hip): v_mad_u64_u32 issues one wave-instruction per ~4 cycles per SIMD
# = 16 lanes/clk/SIMD;  256 CU x 4 SIMD x 16 x clock.  The datasheet clock is 2.4 GHz; under this load the chip
# sustains less (measured live by the clock probe), so both fractions are printed.
LANES_PER_CLK = 256 * 4 * 16
NOMINAL_GHZ = 2.4
PEAK_MACS = LANES_PER_CLK * NOMINAL_GHZ * 1e9


def usable_cores():
    """threads the CPU leg may use: the scheduler affinity, capped by the cgroup CPU quota (a GPU box hands one GPU's job
    a share of the host - 16 cores - while os.cpu_count() still reports every core of the machine)"""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                parts = f.read().split()
            if path.endswith("cpu.max"):
                if parts[0] != "max":
                    n = min(n, max(1, int(parts[0]) // int(parts[1])))
            elif int(parts[0]) > 0:
                with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                    n = min(n, max(1, int(parts[0]) // int(f.read())))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


def launch_ranks(n_ranks, argv):
    """the launcher half of `python bench.py --gpus N` (N > 1, no WORLD_SIZE): run the N ranks under torch.distributed.run as a child
    process (never an exec: nothing here has touched the GPU, and nothing will), relay rank 0's JSON line to stdout - everything else
    the ranks print goes to stderr - and return the child's exit code (3 if no line came back from a child that claimed success)"""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    # --standalone: torchrun's own rendezvous on 127.0.0.1 binds a free port ITSELF (round 6: no port is probed here and closed again before
    # torchrun binds it - a race on a busy box).  A launch that dies before any rank reports (a bind / rendezvous failure) is started once
    # more, as a fresh child process ("dies before any rank reports" = non-zero code within 30 s and no result line).
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1", "--nproc-per-node", str(n_ranks),
           os.path.abspath(__file__)] + list(argv)
    rc, line = 1, None
    for attempt in (1, 2):
        sys.stderr.write("bench.py: launching %d ranks (attempt %d): %s\n" % (n_ranks, attempt, " ".join(cmd)))
        t_start = time.time()
        proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True)
        line = None
        for out in proc.stdout:
            txt = out.strip()
            is_line = False
            if txt.startswith("{") and '"metric"' in txt:
                try:
                    json.loads(txt)
                    is_line = True
                except ValueError:
                    pass
            if is_line:
                line = txt
            else:
                sys.stderr.write(out)
        rc = proc.wait()
        if rc == 0 or line is not None or time.time() - t_start > 30:
            break
    if line is not None:
        print(line, flush=True)
    elif rc == 0:
        sys.stderr.write("bench.py: the ranks ended with code 0 but printed no result line\n")
        rc = 3
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--pairs", type=int, default=GLOBAL_PAIRS, help="global batch, sharded over the ranks")
    ap.add_argument("--pairs-per-gpu", type=int, default=0, help="override: fixed shard per rank (weak scaling)")
    ap.add_argument("--kernel", default=os.environ.get("ZKP_KERNEL", "auto"))
    ap.add_argument("--cpu-sample", type=int, default=16384, help="pairs of the oracle leg (parity sample + all-cores baseline)")
    ap.add_argument("--cpu-threads", type=int, default=0, help="threads of the all-cores oracle leg (default: usable cores, at most 16 per GPU of the job)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sweep", action="store_true", help="skip the batch-size sweep (n = 1 .. 2^18, k = 1 and 3; outside the timed region)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the config-4 / config-5 / host-API legs (they run outside the timed region)")
    ap.add_argument("--bare", action="store_true", help="profiling runs: only warm-up + timed passes reach the GPU (no phase timing, clock "
                                                        "probe or oracle leg), so that a rocprofv3 counter run holds exactly those passes")
    ap.add_argument("--collective", choices=("torch", "abi"), default=os.environ.get("ZKP_BENCH_COLLECTIVE", "torch"),
                    help="torch: all-reduce(MIN) through torch.distributed (nccl = RCCL); abi: the step is zkp_pairing_gt_check_batch_allreduce_dev "
                         "on the library's own RCCL communicator (what a Rust / C host runs)")
    ap.add_argument("--rank-echo", action="store_true", help=argparse.SUPPRESS)   # launcher self-test: ranks report and exit before any GPU use
    args = ap.parse_args()
    if args.bare:
        args.no_cpu_baseline = True

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.rank_echo:
        if rank == 0:
            print(json.dumps({"metric": "rank-echo", "world": world, "gpus": args.gpus, "master": os.environ.get("MASTER_ADDR")}), flush=True)
        sys.exit(0 if world == args.gpus else 2)
    if world != args.gpus:
        sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE is %d - launch one rank per GPU with `python -m torch.distributed.run "
                         "--nnodes=1 --nproc-per-node %d ... bench.py --gpus %d`\n" % (args.gpus, world, args.gpus, args.gpus))
        sys.exit(2)

    import numpy as np
    import torch
    import torch.distributed as dist
    import zkvm_pairings_amd as z
    from zkvm_pairings_amd import dist as zdist
    from zkvm_pairings_amd import synthetic

    # test-only knobs to rehearse N > 1 on a one-GPU box: every rank on cuda:0, gloo instead of RCCL
    shared_gpu = os.environ.get("ZKP_BENCH_SHARE_GPU") == "1"
    if shared_gpu:
        local_rank = 0
    backend = os.environ.get("ZKP_BENCH_BACKEND", "nccl")
    abi_step = args.collective == "abi"
    if abi_step:
        if shared_gpu and world > 1:
            sys.stderr.write("bench.py: --collective abi needs one GPU per rank (RCCL refuses two ranks on one device)\n")
            sys.exit(2)
        backend = "gloo"        # control plane only (id hand-round, barriers, the max over ranks): the data path's collective is the library's
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    ctl = None                  # a gloo group beside torch's RCCL one: agreement about a stuck RCCL call must not itself go through RCCL
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)  # nccl == RCCL on ROCm
            if os.environ.get("ZKP_BENCH_FORCE_ABI_PROBE") == "1":     # only the opt-in probe needs the second group: the default run
                ctl = dist.new_group(backend="gloo")                   # depends on nothing but torch's RCCL communicator
        else:
            dist.init_process_group(backend)

    eng = z.PairingEngine(local_rank)
    if args.kernel != "auto":
        eng.set_kernel(args.kernel)
    # rank-disjoint contiguous slices of one global seeded stream
    if args.pairs_per_gpu:
        lo, hi = rank * args.pairs_per_gpu, (rank + 1) * args.pairs_per_gpu
        global_pairs, scaling = world * args.pairs_per_gpu, "weak"
    else:
        lo, hi = zdist.shard_range(args.pairs, rank, world)
        global_pairs, scaling = args.pairs, "strong"
    n = hi - lo
    g1, g2, sc_a, sc_b = synthetic.random_pairs(eng, n, seed=synthetic.SEED, offset=lo, device_tensors=True)
    out_gt = torch.empty((n, 72), dtype=torch.int64, device=dev)
    ok = torch.empty(n, dtype=torch.uint8, device=dev)
    flag = torch.empty(1, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()

    if abi_step:
        # the library's own communicator: 128 id bytes from rank 0, handed round by the host's means (here: the gloo group)
        uid0 = [z.PairingEngine.comm_unique_id() if rank == 0 else None]
        if world > 1:
            dist.broadcast_object_list(uid0, src=0)
        init_ok = 1
        try:
            eng.comm_init_rank(world, rank, uid0[0])
        except z.ZkpError as ex:
            init_ok = 0
            sys.stderr.write("bench.py: rank %d: zkp_comm_init_rank failed: %r\n" % (rank, ex))
        if world > 1:
            # a rank whose init failed locally (after the bootstrap) is not a member any more and its peers cannot know: agree over the
            # control plane BEFORE anyone enters a collective (include/zkp_pairings.h, zkp_comm_init_rank)
            agreed = torch.tensor([init_ok], dtype=torch.int32)
            dist.all_reduce(agreed, op=dist.ReduceOp.MIN)
            init_ok = int(agreed.item())
        if not init_ok:
            sys.stderr.write("bench.py: rank %d: the library communicator is not complete on every rank - leaving\n" % rank)
            sys.exit(4)

        def step():
            # ONE C-ABI call per rank: Gt + ok bytes of the rank's shard, then ncclAllReduce(count 1, int32, MIN) of the AND flag
            eng.pairing_gt_check_allreduce(g1, g2, 1, out_gt, ok, flag)
    else:
        def step():
            eng.pairing_gt_check(g1, g2, 1, out_gt, ok, flag)
            zdist.and_reduce(flag)  # AND of {0,1} flags == MIN; the only collective on the path (nothing to do on one rank)

    ranks = 1
    if world > 1 and abi_step:
        probe = torch.ones(1, dtype=torch.int32, device=dev)
        eng.and_allreduce(probe)
        torch.cuda.synchronize()
        ranks = eng.comm_info()[0]
    elif world > 1:
        # establish the communicator outside the timed region even when --warmup 0 is requested
        probe = torch.ones(1, dtype=torch.int32, device=dev)
        zdist.and_reduce(probe)
        torch.cuda.synchronize()
        ranks = dist.get_world_size()
    for _ in range(args.warmup):
        step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt_rank = time.perf_counter() - t0
    dt = zdist.max_over_ranks(dt_rank, dev)
    all_ok = int(flag.item())
    # fingerprint of rank 0's Gt block (wrap-around sums on the GPU): two runs of the same build on the same inputs print the same two
    # numbers whatever carried the collective
    w64 = torch.arange(1, 73, dtype=torch.int64, device=dev) * 0x1E3779B97F4A7C15 | 1
    gt_fp = [int(out_gt.sum().item()), int((out_gt * w64).sum().item())]
    del w64

    # ---- diagnosis of an N > 1 line (outside the timed region, every rank): what each rank's kernels take on their own, what
    # the collective takes on its own, each rank's wall time of the timed steps - gathered to rank 0.  A bad scaling curve then
    # says which of the three it was.
    rank_diag = None
    if world > 1:
        own_kernel_ms = eng.time_pairing(g1, g2, out_gt, 1)                 # HIP events on the launching stream, no collective
        torch.cuda.synchronize()
        dist.barrier()
        tcoll = time.perf_counter()
        for _ in range(8):
            if abi_step:
                eng.and_allreduce(flag)
            else:
                zdist.and_reduce(flag)
        torch.cuda.synchronize()
        coll_ms = (time.perf_counter() - tcoll) * 1e3 / 8
        mine = torch.tensor([own_kernel_ms, coll_ms, 1e3 * dt_rank / args.steps, float(n)], dtype=torch.float64)
        rank_diag = zdist.gather_rows(mine, dev)

    # ---- the same collective through the C ABI (zkp_comm_init_rank / zkp_and_allreduce_dev / zkp_pairing_check_batch_allreduce_dev:
    # what a Rust / C host runs, librccl linked into libzkp_pairings.so), outside the timed region and behind a watchdog: a hang or
    # an error here costs the line one field, never the measurement above
    abi_coll, abi_hung = None, False
    # (ZKP_BENCH_FORCE_ABI_PROBE=1 switches the probe on: a one-rank communicator on a single rank - the GPU suite's rehearsal of this code)
    if abi_step:
        abi_coll = {"ok": True, "ranks_ok": ranks, "what": "--collective abi: the TIMED step is zkp_pairing_gt_check_batch_allreduce_dev on the library's "
                                                           "communicator (%d ranks); no separate probe" % ranks}
    elif os.environ.get("ZKP_BENCH_FORCE_ABI_PROBE") == "1" and (world == 1 or (backend == "nccl" and not shared_gpu)):
        # opt-in since round 5: a second RCCL communicator beside torch's live one has never run on more than one rank, and a crash in
        # it would cost the line that is printed below - the ABI path is timed properly by `--collective abi` instead
        import threading
        uid = [z.PairingEngine.comm_unique_id() if rank == 0 else None]
        if world > 1:
            dist.broadcast_object_list(uid, src=0, group=ctl)
        res = {}

        def abi_probe():
            try:
                torch.cuda.set_device(dev)
                eng.comm_init_rank(world, rank, uid[0])
                f = torch.ones(1, dtype=torch.int32, device=dev)
                eng.and_allreduce(f)
                torch.cuda.synchronize()
                tc = time.perf_counter()
                for _ in range(8):
                    eng.and_allreduce(f)
                torch.cuda.synchronize()
                res["allreduce_ms"] = (time.perf_counter() - tc) * 1e3 / 8
                m = min(n, 4096)
                okb, fl = eng.pairing_check_allreduce(g1[:m].contiguous(), g2[:m].contiguous(), 1)
                torch.cuda.synchronize()
                res["check_flag"] = int(fl.item())          # random pairs: never the identity -> 0 on every rank
                res["and_of_ones"] = int(f.item())
                eng.comm_destroy()
                res["ok"] = res["check_flag"] == 0 and res["and_of_ones"] == 1
            except Exception as ex:      # reported in the line
                res["ok"], res["error"] = False, repr(ex)

        th = threading.Thread(target=abi_probe, daemon=True)
        th.start()
        th.join(90)
        abi_hung = th.is_alive()
        abi_coll = {"ok": False, "error": "no answer within 90 s"} if abi_hung else dict(res)
        # every rank leaves the same way: if any rank's probe is stuck, all skip the orderly shutdown below
        abi_coll["ranks_ok"] = (1 if abi_coll.get("ok") else 0) if world == 1 else None
        if world > 1:
            # over the gloo group: a rank whose probe is stuck inside RCCL must still be able to say so
            both = torch.tensor([1 if abi_hung else 0, 1 if abi_coll.get("ok") else 0], dtype=torch.int32)
            mx = both.clone()
            dist.all_reduce(mx, op=dist.ReduceOp.MAX, group=ctl)
            dist.all_reduce(both, op=dist.ReduceOp.SUM, group=ctl)
            abi_hung = bool(mx[0].item())
            abi_coll["ranks_ok"] = int(both[1].item())
        abi_coll["what"] = ("the path's one collective through the C ABI: zkp_comm_init_rank on %d ranks, 8 x zkp_and_allreduce_dev (mean ms), one "
                            "zkp_pairing_check_batch_allreduce_dev of 4096 pairs per rank; rank 0's view" % world)

    line = None
    if rank == 0:
        # ---- everything below is measurement bookkeeping outside the timed region
        # dominant-kernel duration: HIP events recorded on the stream the kernels are launched on
        kern_ms = (1e3 * dt / args.steps) if args.bare else eng.time_pairing(g1, g2, out_gt, 2)

        def timed_ms(fn, reps=2):
            fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                r = fn()
            e1.record()
            e1.synchronize()
            return e0.elapsed_time(e1) / reps, r

        # the two phases on their own (the *_dev calls fork and join on torch's current stream, so its events see them)
        ml_ms = fe_ms = sustained_ghz = None
        if not args.bare:
            ml_ms, ml = timed_ms(lambda: eng.multi_miller_loop(g1, g2, 1))
            fe_ms, _ = timed_ms(lambda: eng.final_exponentiation(ml))
            del ml
            # clock the chip sustains under this load: a one-wavefront probe on a second stream beside a pass.  The probe is queued FIRST
            # (it starts at once and spins for a fixed wall time, half a pass) and two passes behind it: queued behind the pass it could
            # be held back until the pass's streams had drained - and then read the idle clock (round 5: with phase C on the pipelines'
            # streams it did, 2.40 GHz on every box)
            side = torch.cuda.Stream(device=dev)
            torch.cuda.synchronize()
            ticks, wall_khz = eng.clock_probe(side, spin_us=max(20000, int(kern_ms * 500)))
            eng.pairing_gt_check(g1, g2, 1, out_gt, ok, flag)
            eng.pairing_gt_check(g1, g2, 1, out_gt, ok, flag)
            torch.cuda.synchronize()
            tk = ticks.cpu().numpy()
            sustained_ghz = float(tk[0]) / float(tk[1]) * wall_khz * 1e3 / 1e9 if tk[1] > 0 else None

        # what the kernels EXECUTE (multiply-add instructions of the generated asm blocks and step programs x lanes issued, idle
        # lanes included), beside the algorithmic count the roofline fraction is priced on
        executed = None
        try:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import executed_macs
            em = executed_macs.per_pairing()
            executed = {"total": em["total"], "per_kernel": {k: v for k, v in em.items() if k != "total"},
                        "source": "tools/executed_macs.py: v_mad_[iu]64_[iu]32 instructions of the generated asm blocks, T x 147 + 196 per MULACC step of "
                                  "the step programs, x lanes that execute (the interpreter's wavefronts: 60 - lanes 60..63 are switched off since round 5; rounds "
                                  "1-4 counted 64); the compiled decompression / inversion kernels are estimates from their source"}
        except Exception as ex:      # the counter needs tools/ and tests/golden/ of the repository
            executed = {"total": None, "error": repr(ex)}
        # per kernel class: one pass with every launch timed on its own (zkp_profile_pairing_dev: a single pipeline, no overlap) against
        # the multiply-adds that class executes - the table a reader would otherwise rebuild from rocprofv3's kernel stats
        per_kernel = None
        if not args.bare and args.kernel in ("auto", "coop") and executed and executed.get("total"):
            cls_macs = {"k_prep_lines": "k_prep_lines<true>", "k_coop<30,4> miller": "k_coop miller1", "k_coop<24,34> fexp_a": "k_coop fexp_a",
                        "k_batch_inv": "k_batch_inv (estimate from the source)", "k_ksq": "k_ksq", "k_kdec_a": "k_kdec_a (estimate from the source)",
                        "k_kdec_b": "k_kdec_b (estimate from the source)", "k_coop<36,24> hard-part step programs": "k_coop<36,24> hard-part step programs",
                        "k_coop<24,34> phase-C step programs": "k_coop<24,34> phase-C step programs"}
            try:
                prof = eng.profile_pairing(g1, g2, out_gt)
                per_kernel = {"pairs": n, "classes": {}, "sum_ms": sum(v[0] for v in prof.values()),
                              "what": "THE per-kernel table (round 6: the only one - the cool-burst timings of single launches that flattered the "
                                      "Miller program by 10 % are gone): one pass with every launch bracketed by HIP events on ONE pipeline (the timed "
                                      "pass overlaps two; its wall time is kernel_ms): ms and launches per kernel class, executed multiply-adds per "
                                      "pairing of that class (tools/executed_macs.py) with the share of the peak multiply-add issue (256 CU x 4 SIMD x "
                                      "16 lanes x 2.4 GHz) it reaches, and the ALGORITHMIC fraction (SURVEY 8(d)'s reference-shaped count of the work "
                                      "that class does, x 300 multiply-adds) where the count splits by kernel: the line steps, the Fp12 accumulator of "
                                      "the Miller loop, and the final exponentiation's classes together"}
                # SURVEY 8(d) splits by PHASE, not by kernel: the G2 steps + lines (k_prep_lines), the Fp12 accumulator (Miller program), and
                # the final exponentiation as a whole (every other class together)
                alg = {"k_prep_lines": FPMUL_LINES, "k_coop<30,4> miller": FPMUL_MILLER - FPMUL_LINES}
                fexp_ms = sum(v[0] for k_, v in prof.items() if k_ not in alg)
                for name, (ms_c, cnt) in prof.items():
                    mc = executed["per_kernel"].get(cls_macs[name])
                    per_kernel["classes"][name] = {"ms": ms_c, "launches": cnt, "executed_macs_per_pairing": mc,
                                                   "executed_frac_of_peak": (n * mc / (ms_c * 1e-3) / PEAK_MACS) if (mc and ms_c > 0) else None,
                                                   "algorithmic_fp_mul_equivalents": alg.get(name),
                                                   "algorithmic_frac": (n * alg[name] * MACS_PER_FPMUL / (ms_c * 1e-3) / PEAK_MACS) if (name in alg and ms_c > 0) else None}
                per_kernel["final_exponentiation_classes_together"] = {
                    "ms": fexp_ms, "algorithmic_fp_mul_equivalents": FPMUL_FEXP,
                    "algorithmic_frac": (n * FPMUL_FEXP * MACS_PER_FPMUL / (fexp_ms * 1e-3) / PEAK_MACS) if fexp_ms > 0 else None}
            except z.ZkpError as ex:
                per_kernel = {"error": repr(ex)}
        value = global_pairs * args.steps / dt
        achieved = (n * MACS_PER_PAIRING) / (kern_ms * 1e-3)
        phase = lambda fpm, ms: ((n * fpm * MACS_PER_FPMUL) / (ms * 1e-3) / PEAK_MACS) if ms else None
        try:
            sec_all = executed_macs.secondary()
        except Exception:
            sec_all = {}
        phase_exec = lambda key, ms: ((n * sec_all[key]) / (ms * 1e-3) / PEAK_MACS) if (ms and sec_all.get(key)) else None
        traffic = traffic_src = None
        for rnd in ("r06", "r05", "r04", "r03", "r02", "r01"):
            tpath = os.path.join(ROOT, "profiles", rnd, "pmc", "traffic.json")
            if os.path.exists(tpath) and args.kernel in ("auto", "coop"):
                with open(tpath) as tf:   # rocprofv3 PMC passes over one 2^20-pair pass, gfx950-corrected (tools/pmc_traffic.py); linear in n
                    tj = json.load(tf)
                    traffic = tj["hbm_bytes_per_step"] * n / tj["pairs_per_step"]
                    traffic_src = "profiles/%s/pmc/traffic.json" % rnd
                break
        # ---- the other BASELINE configs and the host-pointer entry points, outside the timed region (one GPU, full run only)
        secondary = host_api = None
        if not args.bare and world == 1 and args.kernel in ("auto", "coop") and not args.no_secondary:
            def wall_ms(fn, reps=2):
                fn()
                torch.cuda.synchronize()
                tw = time.perf_counter()
                for _ in range(reps):
                    fn()
                torch.cuda.synchronize()
                return (time.perf_counter() - tw) * 1e3 / reps

            secondary = {}
            try:
                sec_macs = executed_macs.secondary()      # executed multiply-adds per unit, counted from what the generators emit
            except Exception:
                sec_macs = {}
            exec_frac = lambda key, units, ms: (units * sec_macs[key] / (ms * 1e-3) / PEAK_MACS) if sec_macs.get(key) else None
            nc4 = min(1 << 18, max(1, n // 3))                 # config 4: 2^18 three-pair checks, one shared final exponentiation each
            c4_ms = wall_ms(lambda: eng.pairing_gt_check(g1[:3 * nc4], g2[:3 * nc4], 3, None, ok[:nc4], flag))
            secondary["config4_three_pair_checks"] = {
                "checks": nc4, "ms": c4_ms, "checks_per_s": nc4 / c4_ms * 1e3, "fp_mul_equivalents_per_check": FPMUL_CHECK3,
                "frac": nc4 * FPMUL_CHECK3 * MACS_PER_FPMUL / (c4_ms * 1e-3) / PEAK_MACS,
                "executed_macs_per_check": sec_macs.get("config4_three_pair_check"), "executed_frac_of_peak": exec_frac("config4_three_pair_check", nc4, c4_ms),
                "macs_source": "SURVEY.md 8(d): Miller(3 pairs) 21,492 + final exponentiation 11,729 Fp-mul-equivalents x 300; "
                               "recomputable from profiles/r04/workloads_a.json (roofline.config4_three_pair_checks)",
                "what": "multi-Miller loop over 3 pairs (miller3 program: shared squarings) + ONE final exponentiation + identity flag per check; "
                        "the timed batch's points regrouped in threes (the rate does not depend on the flags' values)"}
            v1_ms = wall_ms(lambda: eng.g1_is_valid(g1))       # config 5: on-curve + subgroup check of 2^20 G1 and 2^20 G2 points
            v2_ms = wall_ms(lambda: eng.g2_is_valid(g2))
            secondary["config5_validity"] = {
                "points": n,
                "g1": {"ms": v1_ms, "points_per_s": n / v1_ms * 1e3, "fp_mul_equivalents_per_point": FPMUL_G1_VALID,
                       "frac": n * FPMUL_G1_VALID * MACS_PER_FPMUL / (v1_ms * 1e-3) / PEAK_MACS,
                       "executed_macs_per_point": sec_macs.get("g1_is_valid_point"), "executed_frac_of_peak": exec_frac("g1_is_valid_point", n, v1_ms),
                       "kernel": "k_g1_valid_fast (asm doubling / addition steps, 3 waves per SIMD) + k_g1_valid28 on the points it marks"},
                "g2": {"ms": v2_ms, "points_per_s": n / v2_ms * 1e3, "fp_mul_equivalents_per_point": FPMUL_G2_VALID,
                       "frac": n * FPMUL_G2_VALID * MACS_PER_FPMUL / (v2_ms * 1e-3) / PEAK_MACS,
                       "executed_macs_per_point": sec_macs.get("g2_is_valid_point"), "executed_frac_of_peak": exec_frac("g2_is_valid_point", n, v2_ms),
                       "kernel": "k_g2_valid_fast3 (asm steps, two lanes per point, 3 waves per SIMD: X / Z of the running point in LDS) + k_g2_valid28 on the points it marks"},
                "macs_source": "bench.py FPMUL_G1_VALID / FPMUL_G2_VALID: doubling 2M + 5S, mixed addition 7M + 4S of the inversion-free Jacobian form "
                               "(Fp2: M = 4, S = 2 Fp mul), G1 two chains of 63 + 5 steps, G2 one, + curve equation / endomorphism / comparison, x 300; "
                               "recomputable from profiles/r04/workloads_a.json (roofline.config5_*)",
                "what": "G1Affine::is_valid / G2Affine::is_valid (on-curve + torsion) of the timed batch's points, device-resident; algorithmic count = "
                        "the inversion-free Jacobian form of the reference's tests (its own affine chains cost ~520 inversions per point)"}
            # config 5 END TO END through the one entry point: raw uncompressed bytes -> decode -> is_valid -> pairing check
            # (zkp_points_check_batch[_dev]; one pair per check), resident and from page-locked host bytes
            b1, b2 = eng.encode_points_dev(g1, 1), eng.encode_points_dev(g2, 2)
            st1 = torch.empty(n, dtype=torch.uint8, device=dev)
            st2 = torch.empty(n, dtype=torch.uint8, device=dev)
            pc_ms = wall_ms(lambda: eng.points_check(b1, b2, 1, st1, st2, ok, flag))
            pc_all_valid = bool((st1 == 0).all().item() and (st2 == 0).all().item()) and not bool(ok.any().item())
            hb1, hb2 = eng.host_array((n, 96), np.uint8), eng.host_array((n, 192), np.uint8)
            hb1[:] = b1.cpu().numpy()
            hb2[:] = b2.cpu().numpy()
            torch.cuda.synchronize()
            tw = time.perf_counter()
            hs1, hs2, hok, hall = eng.points_check(hb1, hb2, 1)
            pc_host_ms = (time.perf_counter() - tw) * 1e3
            fpm5 = FPMUL_G1_VALID + FPMUL_G2_VALID + FPMUL_MILLER + FPMUL_FEXP
            secondary["config5_points_check"] = {
                "pairs": n, "resident": {"ms": pc_ms, "pairings_per_s": n / pc_ms * 1e3, "points_per_s": 2 * n / pc_ms * 1e3},
                "from_page_locked_host_bytes": {"ms": pc_host_ms, "pairings_per_s": n / pc_host_ms * 1e3, "points_per_s": 2 * n / pc_host_ms * 1e3,
                                                "equal_resident": bool(np.array_equal(hs1, st1.cpu().numpy()) and np.array_equal(hok, ok.cpu().numpy()))},
                "fp_mul_equivalents_per_pair": fpm5, "frac": n * fpm5 * MACS_PER_FPMUL / (pc_ms * 1e-3) / PEAK_MACS,
                "all_points_valid_no_pairing_is_one": pc_all_valid,
                "what": "BASELINE config 5 as ONE call (zkp_points_check_batch_dev): 2^20 uncompressed (G1, G2) byte strings -> Fp::from_bytes range "
                        "check, G1 / G2 is_valid, fused pairing + Gt::identity() check, status bytes + flags out; decoded points and intermediate "
                        "status bytes stay in HBM.  PCIe-inclusive figure from page-locked host bytes beside it (never `value`)"}
            # round 6: the two ratios the GPU gate used to assert on wall time are measured HERE
            nh = min(n, 1 << 18)
            bh = b1[:nh].clone()
            bh[::2, 95] ^= 1                                     # y of every second G1 point: off the curve
            pv_ms = wall_ms(lambda: eng.points_check(b1[:nh], b2[:nh], 1, st1[:nh], st2[:nh], ok[:nh], flag))
            ph_ms = wall_ms(lambda: eng.points_check(bh, b2[:nh], 1, st1[:nh], st2[:nh], ok[:nh], flag))
            n_bad = int((st1[:nh] != 0).sum().item())
            secondary["config5_points_check"]["half_invalid_ratio"] = {
                "pairs": nh, "all_valid_ms": pv_ms, "every_second_check_invalid_ms": ph_ms, "ratio": ph_ms / pv_ms, "invalid_checks": n_bad,
                "what": "a check with an invalid point costs its validity tests and nothing else: the valid checks are listed on the device and "
                        "the pairing kernels size themselves from the device-resident count (no host read-back since round 6)"}
            del bh
            k9 = 9
            n9 = min(n, 1 << 18) // k9 * k9
            t_new = wall_ms(lambda: eng.multi_miller_loop(g1[:n9], g2[:n9], k9))
            os.environ["ZKP_COOP_NO_STREAM"] = "1"              # read at zkp_init: a second context runs the rounds-1-4 flow (groups of eight + f12mul)
            try:
                eng_old = z.PairingEngine(local_rank)
                t_old = wall_ms(lambda: eng_old.multi_miller_loop(g1[:n9], g2[:n9], k9))
                eng_old.close()
            finally:
                del os.environ["ZKP_COOP_NO_STREAM"]
            secondary["k9_shared_squarings_ratio"] = {
                "pairs": n9, "pairs_per_check": k9, "one_accumulator_ms": t_new, "groups_of_eight_ms": t_old, "ratio": t_new / t_old,
                "what": "multi_miller_loop() of nine-pair checks: ONE accumulator through the run-time-k Miller program (round 5) against two groups "
                        "joined by f12mul (rounds 1-4, ZKP_COOP_NO_STREAM=1, a second context in this process)"}
            del b1, b2, st1, st2, hb1, hb2
            # host-pointer C ABI (what a Rust / C host binds): H2D + kernels + D2H inside the call, PCIe-inclusive, never `value`
            hp1, hp2 = eng.host_array((n, 12)), eng.host_array((n, 24))
            hgt = eng.host_array((n, 72))
            hp1[:] = g1.cpu().numpy().view(np.uint64)
            hp2[:] = g2.cpu().numpy().view(np.uint64)

            def host_ms(fn, reps=2):
                fn()
                tw = time.perf_counter()
                for _ in range(reps):
                    fn()
                return (time.perf_counter() - tw) * 1e3 / reps

            pin_gt = host_ms(lambda: eng.pairing(hp1, hp2, out=hgt))
            pin_ok = bool(np.array_equal(hgt[:: max(1, n // 4096)], out_gt[:: max(1, n // 4096)].cpu().numpy().view(np.uint64)))
            pin_fl = host_ms(lambda: eng.pairing_check(hp1, hp2, 1))
            pg1, pg2 = np.array(hp1), np.array(hp2)              # pageable copies (plain numpy = what a Rust Vec is)
            pgt = np.zeros((n, 72), dtype=np.uint64)             # the caller's own output array, touched before (a reused Vec)
            pag_gt = host_ms(lambda: eng.pairing(pg1, pg2, out=pgt), reps=2)
            pag_ok = bool(np.array_equal(pgt[:: max(1, n // 4096)], out_gt[:: max(1, n // 4096)].cpu().numpy().view(np.uint64)))
            pag_fresh = host_ms(lambda: eng.pairing(pg1, pg2), reps=1)      # a NEW output array per call: its 600 MB of pages are faulted in inside the call
            pag_fl = host_ms(lambda: eng.pairing_check(pg1, pg2, 1), reps=2)
            host_api = {"pairs": n,
                        "pinned_gt_out": {"ms": pin_gt, "pairings_per_s": n / pin_gt * 1e3},
                        "pinned_flags_only": {"ms": pin_fl, "pairings_per_s": n / pin_fl * 1e3},
                        "pageable_gt_out": {"ms": pag_gt, "pairings_per_s": n / pag_gt * 1e3, "over_pinned": pag_gt / pin_gt},
                        "pageable_gt_out_fresh_pages": {"ms": pag_fresh, "pairings_per_s": n / pag_fresh * 1e3, "over_pinned": pag_fresh / pin_gt},
                        "pageable_flags_only": {"ms": pag_fl, "pairings_per_s": n / pag_fl * 1e3, "over_pinned": pag_fl / pin_fl},
                        "gt_equal_resident_path": pin_ok and pag_ok,
                        "what": "zkp_pairing_batch / zkp_pairing_check_batch on HOST arrays: upload, kernels and download inside the call "
                                "(slices of 2^19 pairs, copies on their own streams); pinned = zkp_host_alloc memory, pageable = plain numpy.  "
                                "pageable_gt_out writes into an output array the caller has used before: ROCm's own pageable copies are then at the "
                                "page-locked time (a staging layer built in round 6 lost 2.7 % against them and was removed: "
                                "profiles/r06/host_api_staging_ab.txt); ..._fresh_pages writes into a new np.empty per call - ~25 ms of page faults "
                                "inside the call (the driver faults 600 MB in when it pins them) + ~25 ms of mmap / munmap in the caller's allocator: "
                                "what round 5's 312 ms was.  Reuse the output buffer"}
            del hp1, hp2, hgt, pg1, pg2, pgt
        # bit-exact parity of a seeded sample vs the CPU oracle + timing of the oracle on the host cores
        cpu = None
        parity = None
        if not args.no_cpu_baseline:
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import oracle_lib as o  # cpu_baseline leg / checker only
            host_cores = os.cpu_count() or 1
            threads = args.cpu_threads or min(usable_cores(), 16 * world)
            ns = min(args.cpu_sample, n)
            idx = torch.arange(0, n, n // ns, device=dev)[:ns]
            h1 = g1[idx].cpu().numpy().view(np.uint64)
            h2 = g2[idx].cpu().numpy().view(np.uint64)
            tc = time.perf_counter()
            want = o.pairing_batch(h1, h2, nthreads=threads)
            t_all = time.perf_counter() - tc
            got = out_gt[idx].cpu().numpy().view(np.uint64)
            parity = bool(np.array_equal(got, want))
            # the inputs themselves come from the GPU scalar multiplication: pin a sample of them to the oracle's [a] G1gen, [b] G2gen
            hi_ = idx[:64].cpu().numpy()
            inputs_ok = bool(np.array_equal(h1[:64], o.g1_mul_batch(np.tile(synthetic.G1_GENERATOR, (len(hi_), 1)), sc_a[hi_])) and
                             np.array_equal(h2[:64], o.g2_mul_batch(np.tile(synthetic.G2_GENERATOR, (len(hi_), 1)), sc_b[hi_])))
            parity = parity and inputs_ok
            n1 = min(512, ns)
            tc = time.perf_counter()
            one = o.pairing_batch(h1[:n1], h2[:n1], nthreads=1)
            t_one = time.perf_counter() - tc
            n_slow = min(64, ns)
            tc = time.perf_counter()
            slow = o.pairing_batch_slow(h1[:n_slow], h2[:n_slow], nthreads=1)
            t_slow = time.perf_counter() - tc
            n_slow_all = min(128 * threads, ns)
            tc = time.perf_counter()
            slow_all = o.pairing_batch_slow(h1[:n_slow_all], h2[:n_slow_all], nthreads=threads)
            t_slow_all = time.perf_counter() - tc
            parity = parity and bool(np.array_equal(one, want[:n1])) and bool(np.array_equal(slow, want[:n_slow])) and \
                bool(np.array_equal(slow_all, want[:n_slow_all]))
            cpu = {"value": ns / t_all, "unit": "pairings/s", "cores": threads, "kind": "port", "host_cores": host_cores,
                   "single_thread": {"value": n1 / t_one, "unit": "pairings/s", "cores": 1, "sample": "first %d pairs of the sample" % n1},
                   "reference_faithful_slow_mode": {
                       "value": n_slow / t_slow, "unit": "pairings/s", "cores": 1,
                       "what": "the same restatement built with -DORC_SLOW: canonical integers, 768-bit schoolbook product + long "
                               "division per Fp::mul as in the reference's src/fp.rs:416-434, Fermat inversions, no Montgomery form; "
                               "a restatement, NOT the Rust crate (which cannot be built here and has no pairing)",
                       "sample": "first %d pairs of the sample" % n_slow,
                       "all_threads": {"value": n_slow_all / t_slow_all, "unit": "pairings/s", "cores": threads,
                                       "sample": "first %d pairs of the sample" % n_slow_all}},
                   "sample": "%d of rank 0's %d pairs (every %d-th), CPU restatement oracle/, %d threads (the job's CPU share; the host "
                             "reports %d cores)" % (ns, n, n // ns, threads, host_cores)}
        if abi_step:
            collective = "step = zkp_pairing_gt_check_batch_allreduce_dev: 1 ncclAllReduce(MIN) of the AND flag on the library's RCCL communicator, %d rank(s)" % ranks
        else:
            collective = ("1 all-reduce(MIN) of the AND flag over %d ranks (%s)" % (ranks, "RCCL through torch.distributed" if backend == "nccl" else backend + ", shared-GPU rehearsal")) \
                if ranks > 1 else "one rank: no collective"
        # what ONE GPU says about the strong-scaling curve: a rank's shard at N = 2 / 4 / 8 timed on this build, this box (HIP events, no
        # collective) - the kernels' tails weigh more on a smaller grid, so N ranks cannot beat N x (shard rate / full rate)
        scaling_bound = None
        if world == 1 and not args.bare and not args.pairs_per_gpu and n >= (1 << 13):
            scaling_bound = {"full": {"pairs": n, "ms": kern_ms}, "shards": []}
            for parts in (2, 4, 8):
                m = n // parts
                ms_m = eng.time_pairing(g1[:m], g2[:m], out_gt[:m], 2)
                scaling_bound["shards"].append({"n_gpus": parts, "pairs": m, "ms": ms_m, "rate_vs_full": (m / ms_m) / (n / kern_ms),
                                                "max_speedup": parts * (m / ms_m) / (n / kern_ms)})
            scaling_bound["what"] = ("measured in this run: one pass over the first 2^20/N pairs of the batch on this GPU against the full pass; "
                                     "max_speedup is the ceiling of the N-GPU strong-scaling curve before collective and host cost anything")
        # ---- round 6: the small / medium-batch regime - latency and rate of ONE call at n = 1 .. 2^18 checks, k = 1 and 3,
        # resident inputs, outside the timed region; BASELINE config 2 (2^16 pairs on one GPU) is the n = 65536, k = 1 row
        sweep = None
        if world == 1 and not args.bare and not args.no_sweep and args.kernel in ("auto", "coop") and n >= 64:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import batch_sweep
            sizes = [x for x in batch_sweep.DEFAULT_SIZES if x <= n // 4 or x == 1]
            rows = batch_sweep.sweep(eng, g1, g2, sizes, (1, 3), graph=True)
            full = {1: n / kern_ms}                              # checks per ms of the full pass (k = 1); k = 3: config 4's rate
            if secondary and secondary.get("config4_three_pair_checks"):
                full[3] = secondary["config4_three_pair_checks"]["checks"] / secondary["config4_three_pair_checks"]["ms"]
            for r in rows:
                r["frac_of_full_rate"] = (r["n"] / r["ms_queued"]) / full[r["k"]] if full.get(r["k"]) else None
                r.pop("fp", None)
                if r["n"] == (1 << 16) and r["k"] == 1:
                    r["baseline_config"] = "config 2: batch of 2^16 random (G1,G2) pairings on one GPU"
            sweep = {"rows": rows, "full_rate_checks_per_s": {str(k_): v * 1e3 for k_, v in full.items()},
                     "what": "one zkp_pairing_gt_check_batch_dev call per row (Gt + ok bytes + flag out): ms_call_sync = call + stream synchronisation as a "
                             "host sees it (mean), ms_queued = calls queued back to back (HIP events), checks_per_s from the latter; "
                             "frac_of_full_rate against the 2^20-pair pass (k = 1) / config 4 (k = 3) of this run; hipgraph (n * k <= 4096) = the same call "
                             "captured into a hipGraph and replayed, bytes compared with the plain call.  A single pairing walks ~30 "
                             "dependent launches of one wavefront each: launch gaps are 0.7 % of it (profiles/r06/v58_trace_n1.txt), the rest "
                             "is one wavefront's instruction chain - flat up to ~4096 checks, where the GPU starts to fill"}
            # a STREAM of medium batches (what a verifier with many config-2-sized jobs has): the same calls alternating over two / three
            # contexts on as many streams - one call's latency-bound islands run beside another call's Miller loop
            try:
                import two_contexts
                streams = []
                for ns_ in (1 << 14, 1 << 16, 1 << 17):
                    if ns_ * 3 > n:
                        continue
                    tc_ = two_contexts.run(ns_, 12, 3)
                    streams.append({"pairs_per_call": ns_, "calls": tc_["calls"], "gt_equal": tc_["gt_equal"],
                                    "pairings_per_s": {str(m): tc_["contexts_%d" % m]["pairings_per_s"] for m in (1, 2, 3)},
                                    "frac_of_full_rate": {str(m): tc_["contexts_%d" % m]["pairings_per_s"] / (full[1] * 1e3) for m in (1, 2, 3)}})
                sweep["streams_of_medium_batches"] = {
                    "rows": streams,
                    "what": "12 calls of n pairs queued back to back: on ONE context (key 1), alternating over TWO / THREE contexts, each with its own "
                            "stream, workspace and pipelines (keys 2, 3; tools/two_contexts.py).  A single medium call cannot hide its six "
                            "inversion launches; several calls in flight hide each other's - the way to run a stream of config-2-sized batches"}
            except Exception as ex:      # a diagnostic outside the timed region: never costs the line
                sweep["streams_of_medium_batches"] = {"error": repr(ex)}
            if cpu and cpu.get("single_thread"):
                cpu1_ms = 1e3 / cpu["single_thread"]["value"]
                cpuN_ms = 1e3 / cpu["value"]
                lat = sorted((r["n"], r["ms_call_sync"]) for r in rows if r["k"] == 1)

                def crossover(per_pairing_ms):
                    for m in range(1, lat[-1][0] + 1):
                        gpu_ms = next(ms for nn, ms in lat if nn >= m)        # the latency of the next sweep size up: an upper bound
                        if gpu_ms < m * per_pairing_ms:
                            return m
                    return None
                sweep["cpu_crossover"] = {
                    "cpu_single_thread_ms_per_pairing": cpu1_ms, "gpu_ms_one_pairing": lat[0][1],
                    "n_from_which_one_gpu_call_beats_one_cpu_core": crossover(cpu1_ms),
                    "n_from_which_one_gpu_call_beats_all_cpu_threads": crossover(cpuN_ms), "cpu_threads": cpu["cores"],
                    "what": "CPU restatement (oracle/, -O3) on this box's host cores against ONE call on resident inputs; below the crossover a caller "
                            "with a single pairing is better served by a CPU core (1 pairing: %.2f ms on the GPU, %.2f ms on one core)" % (lat[0][1], cpu1_ms)}
        roof = {"bound": "valu-int (neither hbm nor mfma: 384-bit modular arithmetic; algorithmic intensity 6.56 M MAC per 864 B of I/O = 7,600 MAC/B, "
                         "executed intensity ~30 MAC/B with the line stream and the per-check state that pass through HBM - still 6x above the "
                         "4.9 MAC/B ridge of 39.3 T MAC/s over 8 TB/s)",
                "achieved": achieved / 1e12, "peak": PEAK_MACS / 1e12, "unit": "T u32-MAC/s",
                "frac": achieved / PEAK_MACS,
                "frac_basis": "ALGORITHMIC-equivalent: SURVEY.md 8(d)'s reference-shaped schoolbook count (6.56 M MAC per pairing) over time - it credits "
                              "work the kernels avoid (lazy reduction, Karatsuba, compressed squarings); executed_frac_of_peak beside it is what the "
                              "multiply-add pipe actually issues",
                "traffic": traffic, "traffic_source": traffic_src,
                "traffic_over_algorithmic_io": (traffic / (864.0 * n)) if traffic else None,      # 864 B = 288 in + 576 out per pairing

                "peak_clock_ghz": NOMINAL_GHZ, "sustained_clock_ghz": sustained_ghz,
                "frac_at_sustained_clock": (achieved / (LANES_PER_CLK * sustained_ghz * 1e9)) if sustained_ghz else None,
                "kernel_ms": kern_ms, "algorithmic_macs_per_pairing": MACS_PER_PAIRING,
                "algorithmic_macs_source": "SURVEY.md 8(d): 21,869 Fp-mul-equivalents of the reference-shaped schoolbook tower x 300 multiply-adds of a 12-limb CIOS multiply",
                "executed_macs_per_pairing": executed,
                "executed_frac_of_peak": ((n * executed["total"]) / (kern_ms * 1e-3) / PEAK_MACS) if executed and executed.get("total") else None,
                "phases": {
                    "miller_loop": {"ms": ml_ms, "frac": phase(FPMUL_MILLER, ml_ms), "fp_mul_equivalents": FPMUL_MILLER,
                                    "executed_frac_of_peak": phase_exec("multi_miller_loop_pair", ml_ms),
                                    "kernels": "k_prep_lines<false> (upstream-shaped lines) + k_coop<30,4> (miller1), Gt-less: Miller value to wire"},
                    "final_exponentiation": {"ms": fe_ms, "frac": phase(FPMUL_FEXP, fe_ms), "fp_mul_equivalents": FPMUL_FEXP,
                                             "executed_frac_of_peak": phase_exec("final_exponentiation", fe_ms),
                                             "kernels": "k_coop<24,34> (fexp_a, fexp_c0..5), k_batch_inv, k_ksq, k_kdec_a, k_kdec_b"}},
                "kernels_executed": per_kernel,
                "launch": "one pass over the resident batch = phase A per 2^16-check chunk on two overlapped HIP streams (k_prep_lines, k_coop "
                          "miller, k_coop fexp_a), ONE k_batch_inv, then the phase C plan over the whole shard: six step programs "
                          "alternating with five compressed squaring runs (k_ksq: 57 squarings, 3 snapshots each) and their decompression (k_kdec_a, k_batch_inv, "
                          "k_kdec_b); kernel_ms is that pass timed with HIP events on the launching stream; the per-kernel split is in "
                          "profiles/r04/"}
        line = {
            "metric": "BLS12-381 pairings/s on 2^20 random (G1,G2) pairs; bit-exact Gt vs ref (CPU oracle: the reference's pairings.rs is empty)",
            "value": value, "unit": "pairings/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
            "dtype": "u32", "data": "synthetic",
            "config": {"workload": "batch of %d random (G1,G2) pairs (BASELINE config 3) sharded over %d GPU(s): fused Miller loop + final "
                                   "exponentiation, Gt + identity flags out; %s" % (global_pairs, world, collective),
                       "pairs_per_gpu": n, "global_pairs": global_pairs, "ranks": ranks, "kernel_family": args.kernel,
                       "shared_gpu_rehearsal": shared_gpu, "collective": args.collective,
                       "collective_backend": ("rccl (library communicator, zkp_comm_init_rank)" if abi_step else (backend if ranks > 1 else None)),
                       "all_ok_flag": all_ok, "gt_sample_bit_exact": parity, "gt_fingerprint_rank0": gt_fp},
            "roofline": roof,
            "cpu_baseline": cpu,
            "secondary_workloads": secondary,
            "host_api": host_api,
            "abi_collective": abi_coll,
            "ranks_detail": None if rank_diag is None else {
                "per_rank": [{"rank": i, "pairs": int(r[3]), "kernel_ms_alone": r[0], "collective_ms_alone": r[1], "step_wall_ms": r[2]}
                             for i, r in enumerate(rank_diag)],
                "what": "kernel_ms_alone: one pass of the rank's shard timed with HIP events, no collective; collective_ms_alone: one "
                        "all-reduce(MIN) of the flag (mean of 8, after a barrier); step_wall_ms: the rank's own wall time per timed step",
                "single_gpu_bound": "see strong_scaling_bound of the N = 1 line of the same build (a shard of 2^20 / N pairs timed on one GPU)"},
            "strong_scaling_bound": scaling_bound,
            "batch_sweep": sweep,
        }
        print(json.dumps(line), flush=True)
    if abi_hung:
        os._exit(0 if (line is None or line["config"]["gt_sample_bit_exact"] is not False) else 3)   # a stuck RCCL thread must not hold the exit
    if abi_step:
        torch.cuda.synchronize()
        eng.comm_destroy()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    eng.close()
    if line is not None and line["config"]["gt_sample_bit_exact"] is False:
        sys.exit(3)


if __name__ == "__main__":
    main()
