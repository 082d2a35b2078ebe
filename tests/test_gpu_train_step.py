"""Config 5's DEFAULT training path at its per-GPU size (train.py:114-141; batch 8 over 8 GPUs = 1 query + 15 targets per
replica at 224^2): the model in the training layout (main.py --train_layout ndhwc = channels_last_3d, --train_conv x3), so
the step under test is the hand-written one — conv_x3 IO32 forward / input gradients, wgrad_x3, bn_train — not MIOpen
autograd.  Judged against the SAME step in fp64, next to the stock fp32 step (MIOpen, torch's layout): logits within 1e-3,
gradients no further from fp64 than the stock fp32 step's are (fp32 SlowFast in train mode is ill-conditioned at this size:
two fp32 runs cannot be held to each other)."""
import copy
import os
import sys
from types import SimpleNamespace

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _grad_dist(g, g64):
    per = {k: float((g[k] - g64[k]).norm()) / (float(g64[k].norm()) + 1e-30) for k in g64}
    num = sum(float((g[k] - g64[k]).norm()) ** 2 for k in g64)
    den = sum(float(g64[k].norm()) ** 2 for k in g64)
    return (num / den) ** 0.5, max(per.values())


def test_config5_default_step_at_size_runs_the_hand_written_kernels(avt, dev):
    from avtex import synth, train_ops
    from avtex.dataset import DeviceSegmentBatcher
    from avtex.slowfast import SlowFast

    args = SimpleNamespace(vdata="/tmp", adata=None, n_negs=14, img_size=224, enc_arch="slowfast", window=0, stride=0)
    torch.manual_seed(5)
    ds = avt.AudioVideoSegments(args, "x", split="train", video=(synth.structured_video(3, 600, 64, 64), 30.0))
    torch.manual_seed(0)
    base = avt.ContrastivePredictionTemporal(SlowFast(), SlowFast(), None, 1, 128, temp=0.1, window=ds.window,
                                             stride=ds.stride, enc_arch="slowfast", img_size=224)
    synth.randomise_bn(base, 4, 0.0)  # non-zero residual branches (c_bn is zero-initialised)
    np.random.seed(3)
    bat = DeviceSegmentBatcher(ds, dev).seed_from_numpy()
    q, t, _, _ = bat.batch(torch.tensor([20]))
    assert t[0].shape == (1, 15, 3, 8, 224, 224)
    label = torch.zeros(1, dtype=torch.long, device=dev)

    def run(dtype, product):
        m = copy.deepcopy(base).to(dev, dtype).train()
        if product:
            m = m.to(memory_format=torch.channels_last_3d)
        qq, tt = [v.to(dtype) for v in q], [v.to(dtype) for v in t]
        if product:
            out = m(qq, tt)  # models._InfoNCELogits (HIP normalise -> bmm -> /temp) + HIP CE
            loss = avt.InfoNCECriterion()(out, label)
        else:  # plain PyTorch in `dtype` (models.py:385-417 restated): F.normalize + bmm + /temp + CrossEntropyLoss
            F = torch.nn.functional
            qv = m.q_encoder(qq).view(1, -1)
            tv = m.t_encoder([tt[0].view(15, 3, 8, 224, 224), tt[1].view(15, 3, 32, 224, 224)]).view(1, 15, -1)
            out = torch.bmm(F.normalize(qv, dim=1).unsqueeze(1), F.normalize(tv, dim=2).permute(0, 2, 1)).view(1, 15) / 0.1
            loss = torch.nn.CrossEntropyLoss()(out, label)
        loss.backward()
        torch.cuda.synchronize()
        g = {k: p.grad.detach().double().cpu() for k, p in m.named_parameters() if p.grad is not None}
        return out.detach().double().cpu(), float(loss), g

    assert train_ops.conv_mode() == "x3"  # the default of main.py --train_conv
    before = dict(train_ops.CALLS)
    out_p, loss_p, g_p = run(torch.float32, True)
    ran = {k: train_ops.CALLS[k] - before[k] for k in before}
    print("hand-written training launches of one item:", ran)
    # every convolution's forward, the stride-1 input gradients, the weight gradients and every BatchNorm ran on the HIP kernels
    n_conv = sum(1 for mod in list(base.q_encoder.modules()) + list(base.t_encoder.modules()) if isinstance(mod, torch.nn.Conv3d))
    assert ran["conv_fwd_x3"] == n_conv, (ran, n_conv)  # 110 per encoder
    assert ran["conv_fwd_x3"] >= 2 * 100 and ran["bn_fwd"] >= 2 * 100 and ran["bn_bwd"] == ran["bn_fwd"], ran
    assert ran["wgrad_x3"] + ran["wgrad_stem_x3"] + ran["wgrad_stem_patch"] + ran["miopen_wgrad"] == ran["conv_fwd_x3"], (ran, n_conv)
    assert ran["wgrad_x3"] >= 2 * 100 and ran["dgrad_x3"] >= 2 * 80, ran
    # the four stems on the patch-resident kernels (forward and weight gradient); 16 strided input gradients per encoder
    assert ran["miopen_wgrad"] == 0 and ran["stem_fwd_patch"] == 4 and ran["wgrad_stem_patch"] == 4 and ran["wgrad_stem_x3"] == 0, ran
    assert ran["miopen_dgrad"] == 0 and ran["dgrad_strided_x3"] == 2 * 16, ran  # no MIOpen convolution left in the step

    out_s, loss_s, g_s = run(torch.float32, False)
    out_64, loss_64, g_64 = run(torch.float64, False)
    d_p, d_s = float((out_p - out_64).abs().max()), float((out_s - out_64).abs().max())
    print("logits vs fp64: product %.3e, stock fp32 %.3e; loss %.6f / %.6f / %.6f" % (d_p, d_s, loss_p, loss_s, loss_64))
    assert d_p < 1e-3, d_p  # the stated tolerance (north_star: scores within 1e-3)
    assert abs(loss_p - loss_64) < 1e-4
    assert set(g_p) == set(g_64)
    all_p, worst_p = _grad_dist(g_p, g_64)
    all_s, worst_s = _grad_dist(g_s, g_64)
    print("gradients vs fp64: product %.3e (worst tensor %.3e), stock fp32 %.3e (worst %.3e)" % (all_p, worst_p, all_s, worst_s))
    assert all_p <= 1.5 * all_s + 1e-4, (all_p, all_s)
    assert worst_p <= 2.0 * worst_s + 1e-3, (worst_p, worst_s)


def test_train_conv_switch_selects_miopen_fp32(avt, dev):
    """main.py --train_conv fp32 (train_ops.set_conv_mode): the convolutions of the training step are MIOpen's fp32 kernels,
    the reference's arithmetic — no split-plane launch happens; x3 is restored afterwards."""
    from avtex import train_ops
    from avtex.slowfast import SlowFast

    torch.manual_seed(1)
    net = SlowFast().to(dev).to(memory_format=torch.channels_last_3d).train()
    slow = torch.randn(1, 3, 8, 64, 64, device=dev)
    fast = torch.randn(1, 3, 32, 64, 64, device=dev)
    try:
        assert train_ops.set_conv_mode("fp32") == "fp32" and train_ops.conv_mode() == "fp32"
        before = dict(train_ops.CALLS)
        net([slow, fast]).square().mean().backward()
        assert all(train_ops.CALLS[k] == before[k] for k in ("conv_fwd_x3", "dgrad_x3", "wgrad_x3"))
        assert train_ops.CALLS["bn_fwd"] > before["bn_fwd"]  # the fused BatchNorm passes stay (they compute what stock BN computes)
        g32 = [p.grad.clone() for p in net.parameters()]
    finally:
        train_ops.set_conv_mode("x3")
    net.zero_grad()
    before = dict(train_ops.CALLS)
    net([slow, fast]).square().mean().backward()
    assert train_ops.CALLS["conv_fwd_x3"] > before["conv_fwd_x3"]
    num = sum(float((a - p.grad).norm()) ** 2 for a, p in zip(g32, net.parameters()))
    den = sum(float(a.norm()) ** 2 for a in g32)
    assert (num / den) ** 0.5 < 5e-2  # same step, two arithmetics (this network's fp32 conditioning)
    with pytest.raises(ValueError):
        train_ops.set_conv_mode("bf16")


def test_items_as_one_batch_equal_the_loop_over_items(avt, dev):
    """Config 5's items as ONE batch with per-item BatchNorm groups (train_ops.bn_replicas; what bench.py --mode train runs)
    against one forward/backward per item (round 2's loop = one DataParallel replica each): same logits, same summed gradients
    up to fp32 summation order; running statistics = the FIRST item's update only, as DataParallel keeps replica 0's buffers
    (reference main.py:420)."""
    from avtex import synth, train_ops
    from avtex.dataset import DeviceSegmentBatcher
    from avtex.slowfast import SlowFast

    args = SimpleNamespace(vdata="/tmp", adata=None, n_negs=3, img_size=64, enc_arch="slowfast", window=0, stride=0)
    torch.manual_seed(5)
    ds = avt.AudioVideoSegments(args, "x", split="train", video=(synth.structured_video(3, 400, 48, 48), 30.0))
    torch.manual_seed(0)
    base = avt.ContrastivePredictionTemporal(SlowFast(), SlowFast(), None, 1, 128, temp=0.1, window=ds.window,
                                             stride=ds.stride, enc_arch="slowfast", img_size=64)
    synth.randomise_bn(base, 4, 0.0)
    base = base.to(dev).train().to(memory_format=torch.channels_last_3d)
    np.random.seed(3)
    bat = DeviceSegmentBatcher(ds, dev).seed_from_numpy()
    q, t, _, _ = bat.batch(torch.tensor([20, 31, 7]))
    crit = avt.InfoNCECriterion()

    def grads_of(m):
        return {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None}

    m1 = copy.deepcopy(base)
    with train_ops.bn_replicas(3):
        out1 = m1(q, t)
    crit(out1, torch.zeros(3, dtype=torch.long, device=dev)).backward()
    m2 = copy.deepcopy(base)
    outs = []
    for i in range(3):
        o = m2([v[i : i + 1] for v in q], [v[i : i + 1] for v in t])
        (crit(o, torch.zeros(1, dtype=torch.long, device=dev)) / 3).backward()
        outs.append(o.detach())
        if i == 0:  # what DataParallel keeps: the buffers of the replica on device 0
            b2 = {k: v.detach().clone() for k, v in m2.named_buffers()}
    torch.cuda.synchronize()
    out2 = torch.cat(outs, 0)
    assert float((out1.detach() - out2).abs().max()) < 2e-4 * float(out2.abs().max())
    g1, g2 = grads_of(m1), grads_of(m2)
    assert set(g1) == set(g2)
    num = sum(float((g1[k] - g2[k]).norm()) ** 2 for k in g2) ** 0.5
    den = sum(float(g2[k].norm()) ** 2 for k in g2) ** 0.5
    print("batched vs per-item: logits %.2e, gradients %.2e of the norm" % (float((out1.detach() - out2).abs().max()), num / den))
    assert num / den < 1e-3
    b1 = dict(m1.named_buffers())
    worst = max(float((b1[k].float() - b2[k].float()).abs().max()) / (float(b2[k].float().abs().max()) + 1e-12) for k in b2)
    assert worst < 1e-5, worst


def test_lateral_fusions_concatenate_without_copies_and_give_the_same_step(avt, dev):
    """The four lateral fusions of a SlowFast encoder in train mode: the slow pathway's producers and the lateral BatchNorms write
    slices of one buffer, the gradient slices are read in place (train_ops.join_channels) — same embeddings and the same
    gradients (up to the weight-gradient kernels' atomic summation order) as torch.cat + contiguous gradient copies; and no torch.cat runs inside the encoder."""
    from avtex import synth, train_ops
    from avtex.slowfast import SlowFast

    torch.manual_seed(1)
    base = SlowFast()
    synth.randomise_bn(base, 2, 0.0)
    base = base.to(dev).train().to(memory_format=torch.channels_last_3d)
    g = torch.Generator(device="cpu").manual_seed(7)
    slow = torch.randn(2, 3, 8, 64, 64, generator=g).to(dev)
    fast = torch.randn(2, 3, 32, 64, 64, generator=g).to(dev)

    def step(join):
        train_ops._JOIN = join
        cats = []
        real_cat = torch.cat

        def spy(ts, dim=0, **k):
            cats.append(tuple(ts[0].shape))
            return real_cat(ts, dim, **k)

        torch.cat = spy
        try:
            m = copy.deepcopy(base)
            with train_ops.bn_replicas(2):
                z = m([slow, fast])
            (z * torch.linspace(-1, 1, z.shape[1], device=dev)).sum().backward()
            torch.cuda.synchronize()
        finally:
            torch.cat = real_cat
            train_ops._JOIN = 1
        return z.detach().clone(), {k: p.grad.clone() for k, p in m.named_parameters()}, [c for c in cats if len(c) == 5 and c[0] == 2 and c[3] > 1]  # (activations of the 2-clip batch before the head's pooled
        # concatenation; weight-shaped concatenations are the plane builders')

    z1, g1, cats1 = step(1)
    z0, g0, cats0 = step(0)
    assert len(cats0) == 4 and cats1 == [], (cats0, cats1)  # the four fusions: copies without the buffer protocol, none with it
    assert torch.equal(z1, z0)
    assert set(g1) == set(g0)
    for k in g0:  # (the weight-gradient kernels add their K-slices with fp32 atomics: equal up to that order)
        assert float((g1[k] - g0[k]).abs().max()) <= 2e-5 * float(g0[k].abs().max()) + 1e-12, k


def test_side_streams_are_one_process_wide_list(avt, dev):
    """The synthesis engine's two-stream mode and the training step's query-encoder stream draw from ONE list (ops.side_streams): the
    k-th stream a process creates lands on another hardware queue, and a training side stream created after a two-stream engine run
    shared a queue with the step's own stream (3.5 % slower config-5 leg inside the default bench run: profiles/r05/trainleg_order.log)."""
    from avtex import models, ops

    a = ops.side_streams(dev, 2)
    b = ops.side_streams(dev, 1)
    c = ops.side_streams(dev, 3)
    assert len(a) == 2 and b[0] is a[0] and c[0] is a[0] and c[1] is a[1] and len({s.cuda_stream for s in c}) == 3
    cur = torch.cuda.current_stream(dev)
    s = models._side_stream(dev, cur)
    assert s is models._side_stream(dev, cur) and s.cuda_stream != cur.cuda_stream
    assert any(s is x for x in ops.side_streams(dev, 4))
    with torch.cuda.stream(a[1]):  # a step that itself runs on a side stream gets a different one for its query encoder
        s2 = models._side_stream(dev, torch.cuda.current_stream(dev))
    assert s2.cuda_stream not in (a[1].cuda_stream, s.cuda_stream)


def test_fast_pathway_on_a_side_stream_equals_one_stream(avt, dev):
    """slowfast.PATHWAY_STREAMS: a training forward on the default stream runs its fast pathway on a side stream (and autograd its
    backward there).  Same kernels on the same data in the same per-tensor order: the embedding is bit-identical, the gradients equal
    up to the weight gradient's fp32 atomics."""
    import avtex.slowfast as sf
    from avtex.slowfast import SlowFast

    torch.manual_seed(0)
    net = SlowFast().to(dev).to(memory_format=torch.channels_last_3d).train()
    with torch.no_grad():
        for m in net.modules():
            if isinstance(m, torch.nn.BatchNorm3d):
                m.weight.uniform_(0.5, 1.0)
    g = torch.Generator().manual_seed(1)
    slow = torch.randn((4, 3, 8, 64, 64), generator=g).to(dev).contiguous(memory_format=torch.channels_last_3d)
    fast = torch.randn((4, 3, 32, 64, 64), generator=g).to(dev).contiguous(memory_format=torch.channels_last_3d)
    wout = torch.randn((4, 2304), generator=g).to(dev)

    def run(flag):
        keep, sf.PATHWAY_STREAMS = sf.PATHWAY_STREAMS, flag
        try:
            net.zero_grad(set_to_none=True)
            with avt.train_ops.bn_replicas(2):
                y = net([slow, fast])
            (y * wout).sum().backward()
            torch.cuda.synchronize()
        finally:
            sf.PATHWAY_STREAMS = keep
        return y.detach().clone(), [p.grad.detach().clone() for p in net.parameters()]

    y1, g1 = run(1)
    y0, g0 = run(0)
    assert torch.equal(y1, y0)
    num = sum(float((a - b).double().pow(2).sum()) for a, b in zip(g1, g0)) ** 0.5
    den = sum(float(b.double().pow(2).sum()) for b in g0) ** 0.5
    assert num / den < 1e-5, num / den


def test_graphed_step_equals_the_eager_step():
    """train_ops.GraphedStep (round 6): the device side of a training step — convolutions forward / input gradient / weight gradient,
    fused BatchNorm passes with epilogue statistics, the loss and SGD with momentum — captured ONCE as a HIP graph and replayed,
    against the same steps run eagerly from the same state: same parameters and running statistics to the weight gradient's atomic
    fp32 summation order.  Inputs travel through a static tensor; the host-side routing of the step (weight-plane caches keyed on
    version counters, statistics handed over by data pointer) is frozen at capture and must stay valid on replay."""
    import copy

    from avtex import train_ops
    from avtex.slowfast import ResBlock

    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    net0 = torch.nn.Sequential(ResBlock(16, 64, 16, 3, 1), ResBlock(64, 64, 16, 3, 1)).to(dev).to(memory_format=torch.channels_last_3d).train()
    xs = [torch.randn(4, 16, 4, 12, 12, device=dev).contiguous(memory_format=torch.channels_last_3d) for _ in range(6)]

    def run(graphed):
        net = copy.deepcopy(net0)
        train_ops.invalidate_weight_cache()
        opt = torch.optim.SGD(net.parameters(), lr=0.05, momentum=0.9)
        x_buf = xs[0].clone()

        def step():
            opt.zero_grad(set_to_none=True)
            with train_ops.bn_replicas(2):
                y = net(x_buf)
            loss = y.square().mean()
            loss.backward()
            opt.step()
            return loss.detach()

        losses = []
        if graphed:
            # the three warm-up steps of GraphedStep run on xs[0..2] through the static buffer: feed them by hand
            it = iter(xs)
            orig = step

            def warm_step():
                x_buf.copy_(next(it))
                return orig()

            # warm-up (3 eager steps on xs[0], xs[1], xs[2], on a side stream as GraphedStep does it), capture, then replay on xs[3..5]
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                for _ in range(3):
                    losses.append(float(warm_step()))
            torch.cuda.current_stream(dev).wait_stream(side)
            torch.cuda.synchronize()
            gs = train_ops.GraphedStep(step, dev, warmup=0)
            for x in xs[3:]:
                x_buf.copy_(x)
                losses.append(float(gs()))
        else:
            for x in xs:
                x_buf.copy_(x)
                losses.append(float(step()))
        torch.cuda.synchronize()
        return losses, [p.detach().clone() for p in net.parameters()], [b.detach().clone() for b in net.buffers()]

    le, pe, be = run(False)
    lg, pg, bg = run(True)
    assert all(abs(a - b) <= 1e-5 * max(1.0, abs(a)) for a, b in zip(le, lg)), (le, lg)
    for a, b in zip(pe, pg):
        assert float((a - b).abs().max()) <= 1e-5 * float(a.abs().max()) + 1e-7
    for a, b in zip(be, bg):
        assert float((a.float() - b.float()).abs().max()) <= 1e-5 * float(a.float().abs().max()) + 1e-6
