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


class _OraclePlaneCompute(_OracleCompute):
    """CPU stand-in for dist.HipCompute(temp, "bf16x3"): the planes that travel are the bf16 hi / lo bit patterns of the
    normalised rows (typed bfloat16, as the product's planes are), the similarity is the oracle's three-product bf16 form."""

    def l2norm(self, v, a=None):
        from oracle import cref

        _, hi, lo = cref.l2norm_rows(v.numpy(), None if a is None else a.numpy(), want_split=True)
        return (torch.from_numpy(hi.view(np.int16)).view(torch.bfloat16), torch.from_numpy(lo.view(np.int16)).view(torch.bfloat16))

    def sim(self, q, t):
        from oracle import cref

        u = lambda x: np.ascontiguousarray(x.contiguous().view(torch.int16).numpy()).view(np.uint16)
        return torch.from_numpy(cref.sim_bf16(u(q[0]), u(q[1]), u(t[0]), u(t[1]), self.temp, True))


def _walk_worker_planes(rank, world, port, n, ret):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    import avtex  # noqa: F401
    from avtex import agreement, dist as adist

    if world > 1:
        adist.init_from_env(backend="gloo")
    q, t, _ = _tables(n, 40, 24)
    surv = adist.sharded_survivors(lambda lo, hi: (q[lo:hi].contiguous(), t[lo:hi].contiguous(), None), n, 0.3,
                                   _OraclePlaneCompute(0.1), rank, world, want_sim=True)
    if rank == 0:
        frames, _ = agreement.walk_from_survivors(surv["idx"], surv["seg"], surv["cnt"], n * 2 + 6, 6, 2, 120, q_id=10,
                                                  rng=np.random.RandomState(5))
        ret["frames"], ret["sim"], ret["cnt"] = frames, surv["sim"], surv["cnt"]
    if world > 1:
        dist.destroy_process_group()


def test_sharded_walk_world4_equals_world1():
    """Four ranks, ragged shards (51 = 13 + 13 + 13 + 12), audio columns: same frames list, same matrix bits as one rank."""
    n = 51
    mgr = mp.Manager()
    one, four = mgr.dict(), mgr.dict()
    _walk_worker(0, 1, 0, n, True, one)
    mp.spawn(_walk_worker, args=(4, 29741, n, True, four), nprocs=4, join=True)
    assert list(four["frames"]) == list(one["frames"]) and len(one["frames"]) >= 200
    assert np.array_equal(four["cnt"], one["cnt"]) and np.array_equal(four["sim"], one["sim"])


def test_sharded_walk_world8_equals_world1():
    """Eight ranks — the node north_star names — with ragged shards (51 = 3 x 7 + 5 x 6) and the audio columns: same frames list,
    same matrix bits and survivor counts as one rank (dist.gather_to_root: ranks 1..7 send their blocks and get nothing back)."""
    n = 51
    mgr = mp.Manager()
    one, eight = mgr.dict(), mgr.dict()
    _walk_worker(0, 1, 0, n, True, one)
    mp.spawn(_walk_worker, args=(8, 29761, n, True, eight), nprocs=8, join=True)
    assert list(eight["frames"]) == list(one["frames"]) and len(one["frames"]) >= 200
    assert np.array_equal(eight["cnt"], one["cnt"]) and np.array_equal(eight["sim"], one["sim"]) and eight["k"] == one["k"]


def _gather_worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import avtex  # noqa: F401
    from avtex import dist as adist

    adist.init_from_env(backend="gloo")
    n = 11  # 11 rows over 3 ranks: 4 + 4 + 3
    lo, hi = adist.shard_range(n, rank, world)
    full = torch.arange(n * 5, dtype=torch.float32).view(n, 5)
    got = adist.gather_to_root(full[lo:hi].contiguous(), n, root=0)
    ret[rank] = None if got is None else got.numpy()
    dist.destroy_process_group()


def test_gather_to_root_sends_nothing_back():
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_gather_worker, args=(3, 29771, ret), nprocs=3, join=True)
    assert ret[1] is None and ret[2] is None
    assert np.array_equal(ret[0], np.arange(55, dtype=np.float32).reshape(11, 5))


def test_sharded_plane_exchange_world2_equals_world1():
    """--sim_precision bf16x3 on the sharded route (validate.py): the all-gather carries the two bf16 planes of T_hat
    (N*D*2 B each, north_star's exchange) instead of fp32 rows; rows computed from gathered planes are bit-identical to
    one rank's, so the frames list is too."""
    n = 37
    mgr = mp.Manager()
    one, two = mgr.dict(), mgr.dict()
    _walk_worker_planes(0, 1, 0, n, one)
    mp.spawn(_walk_worker_planes, args=(2, 29751, n, two), nprocs=2, join=True)
    assert np.array_equal(two["sim"], one["sim"]) and np.array_equal(two["cnt"], one["cnt"])
    assert list(two["frames"]) == list(one["frames"])


def _accum_worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from avtex import dist as adist, train_ops

    adist.init_from_env(backend="gloo")
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Tanh(), torch.nn.Linear(5, 3))
    ddp = torch.nn.parallel.DistributedDataParallel(net)
    xs = torch.randn(world, 3, 4, 6, generator=torch.Generator().manual_seed(1))  # [rank][micro-batch][rows][features]
    acc = train_ops.MicroBatchGradients(net.parameters())
    for step in range(2):  # the second step meets used accumulators
        acc.begin(3)
        for k in range(3):
            last = k == 2
            ctx = ddp.no_sync() if not last else __import__("contextlib").nullcontext()
            with ctx:
                loss = ddp(xs[rank, k] * (1.0 + step)).square().sum()
                if last:
                    acc.before_last_backward()
                loss.backward()
                if not last:
                    acc.after_backward()
        acc.finish()
    if rank == 0:
        ret["grads"] = [p.grad.clone() for p in net.parameters()]
    dist.destroy_process_group()


def test_micro_batch_gradients_under_ddp_world2():
    """train_ops.MicroBatchGradients with DistributedDataParallel (gloo, 2 ranks, 3 passes per rank and step): the all-reduce of
    the last pass sees the sum of all passes — the gradients are the mean over ranks of each rank's summed passes."""
    ret = mp.Manager().dict()
    mp.spawn(_accum_worker, args=(2, 29761, ret), nprocs=2, join=True)
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Tanh(), torch.nn.Linear(5, 3))
    xs = torch.randn(2, 3, 4, 6, generator=torch.Generator().manual_seed(1))
    for r in range(2):
        for k in range(3):
            (net(xs[r, k] * 2.0).square().sum() / 2).backward()  # step 1's scale; mean over the 2 ranks
    for got, p in zip(ret["grads"], net.parameters()):
        assert torch.allclose(got, p.grad, rtol=1e-5, atol=1e-6), (got, p.grad)
