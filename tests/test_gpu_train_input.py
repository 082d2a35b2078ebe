"""Device-side training input path (config 5): negative sampling from NumPy's own MT19937 stream on the MI355X
(csrc/negsample.hip) and gather packing (clip_pack_gather) vs the host dataset (dataset/dataset.py:121-253) and the
reference's fixture G9."""
import os
import sys
from types import SimpleNamespace

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _video(n, hw, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randint(0, 256, (n, hw, hw, 3), generator=g, dtype=torch.uint8)


def test_device_negative_sampling_follows_numpy_stream(avt, dev):
    """Same draws as np.random.choice(others, n_negs, replace=False) + hard negatives, item after item, and the same
    generator state afterwards — including across a state regeneration (624 draws) and for every edge idx."""
    from avtex.dataset import DeviceSegmentBatcher

    args = SimpleNamespace(vdata="/tmp", adata=None, n_negs=14, img_size=16, enc_arch="slowfast", window=0, stride=0)
    torch.manual_seed(5)
    ds = avt.AudioVideoSegments(args, "x", split="train", video=(_video(3000, 16, 1), 20.0))  # ~746 segments
    n = len(ds)
    bat = DeviceSegmentBatcher(ds, dev)
    for seed, idxs in ((100, [0, 1, 2, 3, 4, 5]), (7, [n - 1, n - 2, n - 5, n // 2, 17, n - 6, 9, 300])):
        np.random.seed(seed)
        bat.seed_from_numpy()
        want = [ds.sample_ids(i) for i in idxs]  # advances np.random
        host_state = np.random.get_state()
        pos, neg = bat.sample(torch.tensor(idxs))
        assert pos.cpu().tolist() == [w[0] for w in want]
        assert neg.cpu().tolist() == [[int(v) for v in w[1]] for w in want]
        np.random.seed(0)  # clobber, then take the device's state back
        bat.sync_to_numpy()
        got_state = np.random.get_state()
        assert np.array_equal(got_state[1], host_state[1]) and got_state[2] == host_state[2]


def test_device_negative_sampling_matches_reference_fixture_g9(avt, dev):
    """The reference's own dataset under seeded NumPy (fixture G9, tools/gen_golden.py) — through the device sampler."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from tools.gen_golden import make_video

    from avtex.dataset import DeviceSegmentBatcher

    g = np.load(os.path.join(GOLD, "g9_dataset.npz"))
    args = SimpleNamespace(vdata="/tmp", adata=None, n_negs=10, img_size=16, enc_arch="slowfast", window=0, stride=0)
    torch.manual_seed(5)
    ds = avt.AudioVideoSegments(args, "g9", split="train", video=(make_video(3, 150, 16, 16), 20.0))
    assert len(ds) == int(g["len"])
    bat = DeviceSegmentBatcher(ds, dev)
    n_checked = 0
    for key in g.files:
        if not key.startswith("idx"):
            continue
        idx = int(key[3:])
        np.random.seed(100 + idx)
        bat.seed_from_numpy()
        pos, neg = bat.sample(torch.tensor([idx]))
        assert [int(pos[0])] + neg[0].cpu().tolist() == [int(x) for x in g[key]]
        n_checked += 1
    assert n_checked >= 3


def test_device_batch_equals_host_dataset_items(avt, dev):
    """batch(idx): the packed query / target clips equal AudioVideoSegments.__getitem__'s (the reference's per-item CPU
    preprocessing restated in torch) for the same NumPy stream; shapes as the DataLoader's default collate delivers."""
    from avtex.dataset import DeviceSegmentBatcher

    args = SimpleNamespace(vdata="/tmp", adata=None, n_negs=9, img_size=32, enc_arch="slowfast", window=0, stride=0)
    torch.manual_seed(5)
    ds = avt.AudioVideoSegments(args, "x", split="train", video=(_video(400, 24, 2), 20.0))
    idxs = [3, 40, len(ds) - 1]
    np.random.seed(11)
    items = [ds[i] for i in idxs]
    np.random.seed(11)
    bat = DeviceSegmentBatcher(ds, dev).seed_from_numpy()
    qf, tf, qa, ta = bat.batch(torch.tensor(idxs))
    assert qf[0].shape == (3, 3, 8, 32, 32) and qf[1].shape == (3, 3, 32, 32, 32)
    assert tf[0].shape == (3, 10, 3, 8, 32, 32) and tf[1].shape == (3, 10, 3, 32, 32, 32) and qa is None
    for b, it in enumerate(items):
        for k in range(2):
            assert (qf[k][b].cpu() - it[0][k]).abs().max() < 2e-5
            assert (tf[k][b].cpu() - it[3][k]).abs().max() < 2e-5


def test_training_step_at_size_matches_fp32_autograd(avt, dev):
    """Config 5 at its per-GPU size (batch 8 over 8 GPUs = 1 query + 15 targets per replica, train.py:114-141) with the
    REAL SlowFast-8x8-R50 encoders at 224^2: logits, loss and encoder gradients of the product step (device-sampled and
    device-packed batch, fused HIP normalise->bmm->/temp + HIP CE, analytic backward into MIOpen autograd) against plain
    PyTorch fp32 on a copy of the same model (F.normalize + bmm + /temp + nn.CrossEntropyLoss, models.py:385-417)."""
    import copy

    import torch.nn.functional as F

    from avtex.dataset import DeviceSegmentBatcher
    from avtex.slowfast import SlowFast

    args = SimpleNamespace(vdata="/tmp", adata=None, n_negs=14, img_size=224, enc_arch="slowfast", window=0, stride=0)
    torch.manual_seed(5)
    from avtex import synth

    ds = avt.AudioVideoSegments(args, "x", split="train", video=(synth.structured_video(3, 600, 64, 64), 30.0))
    torch.manual_seed(0)
    model = avt.ContrastivePredictionTemporal(SlowFast(), SlowFast(), None, 1, 128, temp=0.1, window=ds.window,
                                              stride=ds.stride, enc_arch="slowfast", img_size=224)
    synth.randomise_bn(model, 4, 0.0)  # non-zero residual branches (c_bn is zero-initialised)
    model = model.to(dev).train()
    ref = copy.deepcopy(model)
    np.random.seed(3)
    bat = DeviceSegmentBatcher(ds, dev).seed_from_numpy()
    q, t, _, _ = bat.batch(torch.tensor([20]))
    assert t[0].shape == (1, 15, 3, 8, 224, 224)
    # product step
    out = model(q, t)
    loss = avt.InfoNCECriterion()(out, torch.zeros(1, dtype=torch.long, device=dev))
    loss.backward()
    # plain PyTorch fp32 on the copy
    qv = ref.q_encoder(q).view(1, -1)
    tv = ref.t_encoder([t[0].view(15, 3, 8, 224, 224), t[1].view(15, 3, 32, 224, 224)]).view(1, 15, -1)
    logits = torch.bmm(F.normalize(qv, dim=1).unsqueeze(1), F.normalize(tv, dim=2).permute(0, 2, 1)).view(1, 15) / 0.1
    rloss = torch.nn.CrossEntropyLoss()(logits, torch.zeros(1, dtype=torch.long, device=dev))
    rloss.backward()
    assert (out - logits).abs().max().item() < 1e-3 and abs(float(loss) - float(rloss)) < 1e-4
    checked = 0
    for (name, p), (_, pr) in zip(model.named_parameters(), ref.named_parameters()):
        if p.grad is None:
            assert pr.grad is None
            continue
        if name.endswith("conv.weight") or ".pathway0_res2.branch2.c.weight" in name or name.endswith("s1_fuse.conv_f2s.weight"):
            rel = ((p.grad - pr.grad).norm() / pr.grad.norm().clamp_min(1e-20)).item()
            assert rel < 2e-3, (name, rel)  # two MIOpen backward passes (atomics in wgrad) + fp32 head arithmetic
            checked += 1
    assert checked >= 6
