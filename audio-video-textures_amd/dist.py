"""Multi-GPU layout of the hot path: one process per MI355X, torch.distributed over RCCL/xGMI.

Replaces the reference's single-process torch.nn.DataParallel (contrastive_video_textures/main.py:420;
validate.py:320, 349-363, 442-445), which replicates the query, re-broadcasts every weight on every call
and gathers outputs on GPU0.  Here weights are resident per rank; clip windows shard by index (they are
independent through packing and encoding, models.py:364-402); the only exchange is ONE all-gather of the
target-side embedding shard before the N x N build; each rank then owns a block of transition rows and
only [N/G, k] survivors travel to rank 0.  The serial stitch walk (validate.py:324, 572) runs on rank 0.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """torchrun / torch.distributed.run environment -> (rank, world, local_rank).  No-op at world 1, unless
    AVT_FORCE_PG=1 asks for a one-rank process group (lets a 1-GPU box drive the RCCL calls of the N>1 path)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if (world > 1 or os.environ.get("AVT_FORCE_PG") == "1") and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"  # "nccl" is RCCL on ROCm
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_range(n, rank, world):
    """Contiguous window-index block of `rank`: [lo, hi).  Remainder spread over the first ranks."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def all_gather_rows(local, n_total, group=None):
    """Rows [n_r, D] of every rank -> [n_total, D] on every rank, in rank order (ragged shards padded to
    the largest).  One collective; on a fully connected xGMI node RCCL drives all 7 links at once."""
    if not dist.is_initialized():
        return local
    world = dist.get_world_size(group)
    sizes = [shard_range(n_total, r, world) for r in range(world)]
    width = max(hi - lo for lo, hi in sizes)
    pad = local
    if local.shape[0] < width:
        pad = torch.zeros((width,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        pad[: local.shape[0]] = local
    out = torch.empty((world * width,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    if local.is_cuda and dist.get_backend(group) == "gloo":
        # device rows under a gloo group (several ranks SHARING one GPU: the sharded path with the real kernels on a one-GPU
        # box — tests/test_gpu_e2e.py; RCCL refuses two ranks on one device): gloo gathers host tensors, staged here
        host = torch.empty(out.shape, dtype=out.dtype)
        dist.all_gather_into_tensor(host, pad.contiguous().cpu(), group=group)
        out.copy_(host)
    else:
        dist.all_gather_into_tensor(out, pad.contiguous(), group=group)
    if all(hi - lo == width for lo, hi in sizes):
        return out
    return torch.cat([out[r * width : r * width + (hi - lo)] for r, (lo, hi) in enumerate(sizes)], 0)


def gather_to_root(local, n_total, group=None, root=0):
    """Row blocks [n_r, ...] -> [n_total, ...] on `root` (None elsewhere): a real gather — every other rank SENDS its block and
    receives nothing (as an all-gather, the driving-audio path's `want_sim` moved the whole N x N matrix to all
    ranks, 1 GB x 8 at N = 16 384).  Ragged shards are padded to the widest block on the wire and cut on the root."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return local
    rank = dist.get_rank(group)
    sizes = [shard_range(n_total, r, world) for r in range(world)]
    width = max(hi - lo for lo, hi in sizes)
    pad = local
    if local.shape[0] < width:
        pad = torch.zeros((width,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        pad[: local.shape[0]] = local
    pad = pad.contiguous()
    host = local.is_cuda and dist.get_backend(group) == "gloo"  # (ranks sharing one GPU under gloo: staged through the host)
    send = pad.cpu() if host else pad
    dst = dist.get_global_rank(group, root) if group is not None else root
    if rank != root:
        dist.gather(send, None, dst=dst, group=group)
        return None
    bufs = [torch.empty_like(send) for _ in range(world)]
    dist.gather(send, bufs, dst=dst, group=group)
    full = torch.cat([bufs[r][: hi - lo] for r, (lo, hi) in enumerate(sizes)], 0)
    return full.to(local.device) if host else full


def _all_reduce_max(t):
    """In-place MAX over the ranks (device tensors under a gloo group travel through the host, as in all_gather_rows)."""
    if t.is_cuda and dist.get_backend() == "gloo":
        h = t.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.MAX)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return t


def barrier_max_time(seconds, device):
    """Max over ranks of a local wall time (bench contract)."""
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    if dist.is_initialized():
        _all_reduce_max(t)
    return float(t.item())


def all_ranks_scalar(value, device):
    """A local scalar of every rank -> list in rank order on every rank (bench.py: per-rank step times, so that a straggler is
    visible next to the max-over-ranks figure the contract reports)."""
    if not dist.is_initialized():
        return [float(value)]
    world = dist.get_world_size()
    host = dist.get_backend() == "gloo"
    t = torch.tensor([float(value)], dtype=torch.float64, device="cpu" if host else device)
    out = torch.empty(world, dtype=torch.float64, device=t.device)
    dist.all_gather_into_tensor(out, t)
    return [float(v) for v in out.cpu()]


class HipCompute:
    """The device arithmetic of the sharded build: the C-ABI kernels (include/avt.h).  The orchestration below only
    needs these three calls; tests/test_dist_gloo.py injects the CPU oracle in their place to run the collectives and
    the index math under gloo (the product never does: ops.* reject host tensors)."""

    def __init__(self, temp, precision="f32"):
        self.temp, self.precision = temp, precision

    def l2norm(self, v, a=None):
        """-> the planes that travel: (fp32,) in f32 mode, (bf16 hi[, bf16 lo]) in the bf16 MFMA modes."""
        from . import ops

        n, hi, lo = ops.l2norm_rows(v, a, want_f32=self.precision == "f32", want_split=self.precision != "f32")
        return (n,) if self.precision == "f32" else ((hi, lo) if self.precision == "bf16x3" else (hi,))

    def sim(self, q, t):
        from . import ops

        if self.precision == "f32":
            return ops.sim_gemm_nt(q[0], t[0], self.temp, "f32")
        return ops.sim_gemm_nt(q[0], t[0], self.temp, self.precision, q_lo=q[1] if len(q) > 1 else None,
                               t_lo=t[1] if len(t) > 1 else None)

    def select(self, sim, q_ids, threshold, cap):
        from . import ops

        return ops.row_transition(sim, q_ids=q_ids, threshold=threshold, cap=cap)


def sharded_survivors(encode_block, n_total, threshold, compute, rank=0, world=1, want_sim=False):
    """The N x N build of validate() sharded over `world` ranks (replaces the DataParallel scatter/gather
    of validate.py:320, 349-363, 442-445, 481-493): rank r encodes windows [lo, hi) with both encoders
    (encode_block(lo, hi) -> (q_rows, t_rows, audio_rows | None)), normalises them, ONE all-gather of the normalised
    target planes, its row block of sim = Q_r T^T / temp, the row select, and only the survivors travel to rank 0:
    the widest survivor list is agreed by an all-reduce(MAX) so nothing is truncated.
    -> on rank 0 a dict of host arrays (idx, seg, p [N, kmax], cnt [N], stats [N, 4][, sim [N, N] when want_sim]);
    None on the other ranks."""
    lo, hi = shard_range(n_total, rank, world)
    qv, tv, av = encode_block(lo, hi)
    qn = compute.l2norm(qv, av)
    tn = compute.l2norm(tv, av)
    t_all = tuple(all_gather_rows(p, n_total) for p in tn)
    sim = compute.sim(qn, t_all)
    q_ids = torch.arange(lo, hi, device=sim.device, dtype=torch.int64)
    sel = compute.select(sim, q_ids, threshold, n_total)
    kmax = sel["cnt"].max().to(torch.int64).reshape(1) if hi > lo else torch.zeros(1, dtype=torch.int64, device=sim.device)
    if dist.is_initialized():
        _all_reduce_max(kmax)
    k = max(int(kmax.item()), 1)
    out = {}
    for key in ("idx", "seg", "p"):
        out[key] = gather_to_root(sel[key][:, :k].contiguous(), n_total)
    for key in ("cnt", "stats"):
        out[key] = gather_to_root(sel[key], n_total)
    if want_sim:
        out["sim"] = gather_to_root(sim, n_total)
    if dist.is_initialized() and dist.get_rank() != 0:
        return None
    return {key: v.cpu().numpy() for key, v in out.items()}


def sharded_transition_build(engine, n_total, threshold, cap, precision="f32", rank=0, world=1, starts=None):
    """Config 4 pipeline on one rank: encode own window block with both encoders -> all-gather T (and the
    audio table) -> own row block of sim = Q_r T^T / temp -> select.  Returns the rank's survivor dict and
    its [lo, hi) row range."""
    import numpy as np

    from . import ops

    lo, hi = shard_range(n_total, rank, world)
    own = (np.arange(lo, hi, dtype=np.int64) * engine.S) if starts is None else starts[lo:hi]
    qv, tv = engine.embed_windows([engine.q_enc, engine.t_enc], starts=own)
    qn, qh, ql = ops.l2norm_rows(qv, want_split=precision != "f32")
    tn, th, tl = ops.l2norm_rows(tv, want_split=precision != "f32")
    if precision == "f32":
        t_all = all_gather_rows(tn, n_total)
        sim = ops.sim_gemm_nt(qn, t_all, engine.temp, "f32")
    else:
        th_all = all_gather_rows(th, n_total)
        tl_all = all_gather_rows(tl, n_total) if precision == "bf16x3" else None
        sim = ops.sim_gemm_nt(qh, th_all, engine.temp, precision, q_lo=ql, t_lo=tl_all)
    q_ids = torch.arange(lo, hi, device=sim.device, dtype=torch.int64)
    sel = ops.row_transition(sim, q_ids=q_ids, threshold=threshold, cap=cap)
    return sel, (lo, hi), sim
