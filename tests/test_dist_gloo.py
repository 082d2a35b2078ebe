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
