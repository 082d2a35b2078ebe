"""N > 1 path on CPU: world_size-2 gloo processes exercise the shard layout and the collectives of dist.py.
The per-rank compute is the oracle here (test code), the collectives and index math are the product's."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, n, d, ret):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    import avtex
    from avtex import dist as adist
    from oracle import cref

    r, w, _ = adist.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    g = torch.Generator().manual_seed(0)
    Q = torch.randn((n, d), generator=g)
    T = torch.randn((n, d), generator=torch.Generator().manual_seed(1))
    lo, hi = adist.shard_range(n, rank, world)
    # each rank "encodes" only its block, normalises it, all-gathers the target side (ragged shards)
    qn, _, _ = cref.l2norm_rows(Q[lo:hi].numpy(), want_split=False)
    tn, _, _ = cref.l2norm_rows(T[lo:hi].numpy(), want_split=False)
    t_all = adist.all_gather_rows(torch.from_numpy(tn), n)
    assert t_all.shape == (n, d)
    sim = cref.sim_f32(qn, t_all.numpy(), 0.1)  # rows lo:hi of the N x N matrix
    sel = cref.row_transition(sim, q_ids=np.arange(lo, hi), n_seg=n, threshold=0.3, cap=16)
    seg = adist.gather_to_root(torch.from_numpy(sel["seg"]), n)
    cnt = adist.gather_to_root(torch.from_numpy(sel["cnt"]), n)
    tmax = adist.barrier_max_time(float(rank + 1), torch.device("cpu"))
    assert tmax == float(world)
    if rank == 0:
        ret["seg"], ret["cnt"], ret["t_all"] = seg.numpy(), cnt.numpy(), t_all.numpy()
    else:
        assert seg is None
    dist.destroy_process_group()


@pytest.mark.parametrize("n", [37, 64])
def test_sharded_build_equals_single_process(n):
    from oracle import cref

    d, world = 48, 2
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 29600 + n
    mp.spawn(_worker, args=(world, port, n, d, ret), nprocs=world, join=True)
    Q = torch.randn((n, d), generator=torch.Generator().manual_seed(0)).numpy()
    T = torch.randn((n, d), generator=torch.Generator().manual_seed(1)).numpy()
    qn, _, _ = cref.l2norm_rows(Q, want_split=False)
    tn, _, _ = cref.l2norm_rows(T, want_split=False)
    assert np.array_equal(ret["t_all"], tn)
    one = cref.row_transition(cref.sim_f32(qn, tn, 0.1), q_ids=np.arange(n), threshold=0.3, cap=16)
    assert np.array_equal(ret["cnt"], one["cnt"]) and np.array_equal(ret["seg"], one["seg"])  # identical stitch indices


def test_shard_range_partitions():
    from avtex import dist as adist

    for n in (1, 7, 4096, 16384, 16385):
        for w in (1, 2, 3, 8):
            r = [adist.shard_range(n, k, w) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n and all(r[i][1] == r[i + 1][0] for i in range(w - 1))
            assert max(h - l for l, h in r) - min(h - l for l, h in r) <= 1


class _OracleCompute:
    """CPU stand-in for dist.HipCompute (TEST code: the product's arithmetic is the HIP kernels and rejects host tensors).
    Lets the product's orchestration — shard ranges, all-gather, agreed survivor width, gather to root — run under gloo."""

    def __init__(self, temp):
        self.temp = temp

    def l2norm(self, v, a=None):
        from oracle import cref

        y, _, _ = cref.l2norm_rows(v.numpy(), None if a is None else a.numpy(), want_split=False)
        return (torch.from_numpy(y),)

    def sim(self, q, t):
        from oracle import cref

        return torch.from_numpy(cref.sim_f32(q[0].numpy(), t[0].numpy(), self.temp))

    def select(self, sim, q_ids, threshold, cap):
        from oracle import cref

        o = cref.row_transition(sim.numpy(), q_ids=q_ids.numpy(), n_seg=sim.shape[1], threshold=threshold, cap=cap)
        return {k: torch.from_numpy(v) for k, v in o.items()}


def _tables(n, dv, da):
    """Embedding tables with structure (neighbours similar) so survivor lists have different lengths per row."""
    g = torch.Generator().manual_seed(3)
    base = torch.randn((n // 6 + 2, dv), generator=g)
    t_ = torch.arange(n, dtype=torch.float32) / 6
    i0 = t_.floor().long()
    fr = (t_ - i0.float()).view(-1, 1)
    q = (1 - fr) * base[i0] + fr * base[i0 + 1] + 0.05 * torch.randn((n, dv), generator=g)
    t = q.roll(-1, 0) + 0.05 * torch.randn((n, dv), generator=g)
    a = torch.randn((n, da), generator=g).abs()
    return q.contiguous(), t.contiguous(), a.contiguous()


def _walk_worker(rank, world, port, n, with_audio, ret):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    import avtex
    from avtex import agreement, dist as adist

    if world > 1:
        adist.init_from_env(backend="gloo")
    q, t, a = _tables(n, 40, 24)
    seen = []

    def encode_block(lo, hi):  # a rank only ever "encodes" its own block
        seen.append((lo, hi))
        return q[lo:hi].contiguous(), t[lo:hi].contiguous(), (a[lo:hi].contiguous() if with_audio else None)

    surv = adist.sharded_survivors(encode_block, n, 0.3, _OracleCompute(0.1), rank, world, want_sim=True)
    assert seen == [adist.shard_range(n, rank, world)]
    if rank == 0:
        W, S = 6, 2
        frames, chosen = agreement.walk_from_survivors(surv["idx"], surv["seg"], surv["cnt"], n * S + W, W, S, 200, q_id=10,
                                                       rng=np.random.RandomState(5))
        ret["frames"], ret["cnt"], ret["k"], ret["sim"] = frames, surv["cnt"], surv["idx"].shape[1], surv["sim"]
    else:
        assert surv is None
    if world > 1:
        dist.destroy_process_group()


@pytest.mark.parametrize("with_audio", [False, True])
def test_sharded_walk_world2_equals_world1(with_audio):
    """validate()'s aligned multi-rank path (dist.sharded_survivors + the rank-0 walk): the frames list at world 2 equals
    world 1's, with ragged shards (n odd), ragged survivor lists and the m=2 audio columns joined into the normalise."""
    n = 51
    mgr = mp.Manager()
    one, two = mgr.dict(), mgr.dict()
    _walk_worker(0, 1, 0, n, with_audio, one)
    mp.spawn(_walk_worker, args=(2, 29711 + int(with_audio), n, with_audio, two), nprocs=2, join=True)
    assert list(two["frames"]) == list(one["frames"]) and len(one["frames"]) >= 200
    assert np.array_equal(two["cnt"], one["cnt"]) and two["k"] == one["k"] == int(one["cnt"].max())
    assert np.array_equal(two["sim"], one["sim"])  # bit-identical rows whatever the shard boundary
    assert one["cnt"].min() < one["cnt"].max()  # ragged survivor lists: the agreed width really is a maximum
