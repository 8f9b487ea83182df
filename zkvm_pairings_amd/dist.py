"""Multi-GPU layer: one process per GPU (torch.distributed; backend "nccl" is RCCL on ROCm, "gloo" in
the CPU tests).  The pairing path shards embarrassingly - every pairing / k-pair check is
independent - so the batch is split into contiguous blocks per rank and the ONLY
collective is one all-reduce of the per-rank AND flag.  RCCL has no bitwise-AND reduction, so the
AND of {0,1} flags is MIN.  Bulk data never crosses xGMI.

The ONE-product variant (prod over the WHOLE batch of e(P_i,Q_i) == 1, the BLS batch
verification shape) has a real exchange step: every rank reduces its shard to one Fp12 Miller value,
the ranks all-gather those 576-byte values, multiply them and run ONE final exponentiation."""
import torch
import torch.distributed as dist


def world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_range(n_total, rank, world_size):
    """contiguous block [lo, hi) of rank: element i of n_total goes to rank floor(i * world / n_total)-ish,
    block sizes differ by at most one"""
    base, rem = divmod(n_total, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def _active():
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def _host_backend():
    """gloo (CPU tests, one-GPU rehearsals) reduces host tensors; nccl (= RCCL) reduces device tensors in place"""
    return dist.get_backend() != "nccl"


def and_reduce(flag):
    """flag: int32 tensor of shape (1,) holding 0/1 on this rank's device -> global AND, in place.
    One all-reduce(MIN) per call; on a single rank there is nothing to do."""
    if _active():
        if flag.is_cuda and _host_backend():
            h = flag.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.MIN)
            flag.copy_(h)
        else:
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    return flag


def max_over_ranks(seconds, device):
    if not _active():
        return float(seconds)
    t = torch.tensor([seconds], dtype=torch.float64, device="cpu" if _host_backend() else device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_rows(row, device):
    """row: 1-D float64 host tensor of this rank -> (world, len) list of lists on every rank (diagnostics of bench.py: one
    all_gather of a few numbers, outside any timed region)"""
    if not _active():
        return [row.tolist()]
    src = row.contiguous() if _host_backend() else row.to(device)
    parts = [torch.empty_like(src) for _ in range(dist.get_world_size())]
    dist.all_gather(parts, src)
    return [p.cpu().tolist() for p in parts]


def sharded_pairing_check(check_fn, n_checks, device):
    """check_fn(lo, hi) -> int32 tensor (1,) with the AND over this rank's checks [lo, hi).
    Returns the global AND as a python bool on every rank."""
    rank, ws = world()
    lo, hi = shard_range(n_checks, rank, ws)
    try:
        flag = check_fn(lo, hi) if hi > lo else torch.ones(1, dtype=torch.int32, device=device)
    except Exception:
        # a rank whose own block fails still joins the collective - with flag 0 - so that no peer waits for it and every rank
        # reads False; the failure is raised here afterwards (the C ABI's zkp_pairing_check_batch_allreduce does the same)
        and_reduce(torch.zeros(1, dtype=torch.int32, device=device))
        raise
    and_reduce(flag)
    return bool(flag.item())


def gather_parts(part):
    """part: (72,) int64/uint64 tensor (this rank's Fp12 Miller product) -> (world, 72) on every rank.
    One all_gather of 576 B per rank (latency bound; xGMI bandwidth is irrelevant)."""
    if _active():
        src = part.contiguous()
        if src.is_cuda and _host_backend():
            src = src.cpu()
        parts = [torch.empty_like(src) for _ in range(dist.get_world_size())]
        dist.all_gather(parts, src)
        return torch.stack(parts).to(part.device)
    return part.reshape(1, 72)


def sharded_product_check(miller_product_fn, finish_fn, n_pairs, device=None, dtype=torch.int64):
    """prod_{i < n_pairs} e(P_i, Q_i) == Gt::identity() over ranks.
    miller_product_fn(lo, hi) -> (72,) tensor: Miller product of pairs [lo, hi) (engine.miller_product; the
    empty range gives Fp12::one()); finish_fn(parts (world,72)) -> bool: final_exponentiation(prod parts) ==
    identity (engine.fp12_product + engine.final_exponentiation).  Every rank returns the same bool.
    device / dtype: where and as what miller_product_fn returns its value ON EVERY RANK (default: the current CUDA device under
    RCCL, the host under gloo; int64) - the record a failing rank contributes must match its peers' in both, or the all-gather
    meant to keep them from hanging is itself mismatched."""
    rank, ws = world()
    lo, hi = shard_range(n_pairs, rank, ws)
    try:
        part = miller_product_fn(lo, hi)
    except Exception:
        # as above: the failing rank contributes the ZERO record (product 0, never the identity: every rank reads False) and raises
        if device is None:
            device = "cpu" if not _active() or _host_backend() else torch.device("cuda", torch.cuda.current_device())
        gather_parts(torch.zeros(72, dtype=dtype, device=device))
        raise
    return bool(finish_fn(gather_parts(part)))
