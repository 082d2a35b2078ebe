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
    dist.all_gather_into_tensor(out, pad.contiguous(), group=group)
    if all(hi - lo == width for lo, hi in sizes):
        return out
    return torch.cat([out[r * width : r * width + (hi - lo)] for r, (lo, hi) in enumerate(sizes)], 0)


def gather_to_root(local, n_total, group=None, root=0):
    """Row blocks [n_r, ...] -> [n_total, ...] on `root` (None elsewhere)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return local
    full = all_gather_rows(local, n_total, group)
    return full if dist.get_rank(group) == root else None


def barrier_max_time(seconds, device):
    """Max over ranks of a local wall time (bench contract)."""
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    if dist.is_initialized():
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


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
