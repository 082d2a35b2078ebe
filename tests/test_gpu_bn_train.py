"""Fused train-mode BatchNorm3d (+ shortcut add + ReLU) forward / backward (csrc/bn_train.hip via avtex.train_ops) against
torch.nn.BatchNorm3d + add + ReLU through autograd on the same inputs — the ops the reference's training step runs
(contrastive_video_textures/train.py:114-141 with the SlowFast blocks of models/models.py:385-417 in train mode)."""
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _cl(t):
    return t.contiguous(memory_format=torch.channels_last_3d)


@pytest.mark.parametrize("c,shape,res,relu", [
    (64, (2, 4, 14, 14), False, True),     # stem-like
    (8, (3, 8, 9, 7), False, True),        # the fast pathway's narrow layers, ragged rows
    (256, (2, 2, 7, 7), True, True),       # block exit: BN + shortcut + ReLU
    (256, (2, 2, 7, 7), False, False),     # projection shortcut: BN only
    (2048, (2, 1, 4, 4), True, True),      # C/4 > one workgroup
    (4096, (1, 1, 3, 3), False, True),
    (32, (1, 1, 1, 5), True, False),       # fewer rows than threads
    (64, (2, 8, 56, 56), True, True),      # more chunks than one sweep of the grid: every thread walks several rows
    (8, (4, 32, 56, 56), False, True),
    (2048, (4, 8, 7, 7), True, True),      # wide rows, several sweeps
])
def test_bn_act_matches_torch(c, shape, res, relu):
    from avtex import train_ops
    torch.manual_seed(c + shape[1])
    dev = "cuda:0"
    b, t, h, w = shape
    x0 = _cl(torch.randn(b, c, t, h, w, device=dev) * 2.0 + 0.7)
    r0 = _cl(torch.randn(b, c, t, h, w, device=dev)) if res else None
    gy = _cl(torch.randn(b, c, t, h, w, device=dev))

    def run(fused):
        bn = nn.BatchNorm3d(c).to(dev)
        with torch.no_grad():
            bn.weight.copy_(torch.linspace(0.5, 1.5, c))
            bn.bias.copy_(torch.linspace(-0.3, 0.3, c))
            bn.running_mean.fill_(0.25)
            bn.running_var.fill_(2.0)
        bn.train()
        x = x0.clone().requires_grad_(True)
        r = r0.clone().requires_grad_(True) if res else None
        if fused:
            assert train_ops.fusable(x, bn, r)
            y = train_ops.bn_act(x, bn, res=r, relu=relu)
        else:
            y = bn(x)
            if res:
                y = y + r
            if relu:
                y = F.relu(y)
        y.backward(gy)
        return (y.detach(), x.grad, None if r is None else r.grad, bn.weight.grad, bn.bias.grad, bn.running_mean.clone(),
                bn.running_var.clone(), int(bn.num_batches_tracked))

    a, e = run(True), run(False)
    names = ["y", "dx", "dres", "dgamma", "dbeta", "running_mean", "running_var"]
    for n, u, v in zip(names, a[:7], e[:7]):
        if v is None:
            assert u is None
            continue
        scale = float(v.abs().max()) + 1e-12
        err = float((u - v).abs().max()) / scale
        # fp32 ops against fp32 ops with fp64 statistics on our side: a few ulps of the largest value
        assert err < 2e-5, (n, err)
    assert a[7] == e[7] == 1


def test_bn_act_falls_back_outside_its_domain():
    from avtex import train_ops
    dev = "cuda:0"
    bn = nn.BatchNorm3d(24).to(dev).train()            # not a power of two
    x = _cl(torch.randn(2, 24, 2, 5, 5, device=dev))
    assert not train_ops.fusable(x, bn)
    assert torch.equal(train_ops.bn_act(x, bn, relu=True), F.relu(nn.BatchNorm3d(24).to(dev).train()(x)))
    bn = nn.BatchNorm3d(16).to(dev).train()
    x = torch.randn(2, 16, 2, 5, 5, device=dev)       # NCDHW layout
    assert not train_ops.fusable(x, bn)
    bn.eval()
    assert not train_ops.fusable(_cl(x), bn)           # eval mode: running statistics, the stock op


def test_slowfast_train_step_is_as_close_to_fp64_as_the_stock_ops():
    """One train-mode forward / backward of the SlowFast encoder with the fused passes, judged against the same step in
    fp64.  The first form of this test asked for fused == stock fp32 within 2e-3 on every parameter's gradient and failed at
    5.4e-2; tools/probe_bn_train.py (log: profiles/r02/probe_bn_train.log) showed that is this network's fp32 conditioning at
    this size, not the kernels: against fp64 the stock NCDHW, stock channels-last and fused runs are 1.9e-2, 2.2e-2 and
    1.9e-2 away over all gradients with the SAME parameters worst (3-5 %), while every fused call's forward output is within
    2e-7 of fp64 on its own input.  So the claim tested is the one that can hold: the fused step is no further from the
    fp64 gradients than the stock fp32 step is.  The running statistics went the same way: the second form compared them
    with fp64 at rtol 1e-4 / atol 1e-6 and failed; the probe then showed the stock fp32 runs miss that too (worst error
    relative to the tensor's largest entry 2.6e-5 NCDHW, 2.8e-5 channels-last, 2.4e-5 fused, same tensors worst), so they
    are held to the stock step's distance as well.  Per layer, on the same input, the fused pass IS held to torch's numbers
    directly (test_bn_act_matches_torch: 2e-5 on y, dx, dres, dgamma, dbeta and both running statistics)."""
    import copy
    from avtex import slowfast, train_ops
    torch.manual_seed(3)
    dev = "cuda:0"
    net = slowfast.SlowFast()
    for m in net.modules():  # (the zero-initialised last BatchNorm of every block would hide its branch from the gradients)
        if isinstance(m, nn.BatchNorm3d):
            nn.init.uniform_(m.weight, 0.5, 1.5)
            nn.init.uniform_(m.bias, -0.2, 0.2)
    state = copy.deepcopy(net.state_dict())
    clip = torch.randn(2, 3, 32, 64, 64)

    def step(dtype, fused):
        n = slowfast.SlowFast()
        n.load_state_dict(state)
        n = n.to(dev, dtype).to(memory_format=torch.channels_last_3d).train()
        fast = _cl(clip.to(dev, dtype))
        slow = _cl(clip.to(dev, dtype)[:, :, ::4])
        old = (train_ops._FUSED, train_ops._CONV_X3)
        train_ops._FUSED = train_ops._CONV_X3 = 1 if fused else 0
        try:
            out = n([slow, fast])
            loss = (out.double() ** 2).mean()
            loss.backward()
        finally:
            train_ops._FUSED, train_ops._CONV_X3 = old
        return (float(loss.detach()), out.detach().double(), {k: p.grad.double() for k, p in n.named_parameters()},
                {k: v.double() for k, v in n.state_dict().items() if "running" in k})

    l64, o64, g64, r64 = step(torch.float64, False)
    ls, os_, gs, rs = step(torch.float32, False)
    lf, of, gf, rf = step(torch.float32, True)

    def dist(g):
        per = {k: float((g[k] - g64[k]).norm()) / (float(g64[k].norm()) + 1e-30) for k in g64}
        num = sum(float((g[k] - g64[k]).norm()) ** 2 for k in g64)
        den = sum(float(g64[k].norm()) ** 2 for k in g64)
        return (num / den) ** 0.5, max(per.values())

    all_s, worst_s = dist(gs)
    all_f, worst_f = dist(gf)
    assert abs(lf - l64) <= 1e-5 * abs(l64)
    assert float((of - o64).abs().max()) <= 2.0 * float((os_ - o64).abs().max()) + 1e-6
    assert all_f <= 1.5 * all_s + 1e-4, (all_f, all_s)          # measured 1.87e-2 against 2.23e-2
    assert worst_f <= 2.0 * worst_s + 1e-3, (worst_f, worst_s)   # measured 5.1e-2 against 5.5e-2

    def stat_dist(r):  # worst error of a running statistic, relative to that tensor's largest entry
        return max(float((r[k] - r64[k]).abs().max()) / (float(r64[k].abs().max()) + 1e-30) for k in r64)

    assert stat_dist(rf) <= 2.0 * stat_dist(rs) + 1e-6, (stat_dist(rf), stat_dist(rs))  # measured 2.4e-5 against 2.8e-5


@pytest.mark.parametrize("c,dims,groups,res,relu", [
    (64, (8, 2, 6, 6), 4, True, True),
    (32, (6, 3, 5, 7), 3, False, True),
    (256, (8, 1, 4, 4), 8, False, False),
    (2048, (4, 1, 2, 2), 2, True, True),   # a row wider than one workgroup
])
def test_replica_groups_equal_a_loop_over_the_groups(c, dims, groups, res, relu):
    """train_ops.bn_replicas(n): ONE launch over the batch with n groups of statistics == n launches over the groups
    (the reference's per-replica BatchNorm under DataParallel, main.py:420): outputs and input gradients bit for bit, dgamma /
    dbeta to fp32 rounding (summed over the groups in fp64 here, in fp32 by autograd there).  Running statistics and
    num_batches_tracked: GROUP 0's update only — DataParallel keeps the buffers of the replica on device 0 and drops the others'."""
    from avtex import train_ops
    torch.manual_seed(c + groups)
    b, t, h, w = dims
    x = (torch.randn(b, c, t, h, w, device="cuda:0") * 2 + 0.5).contiguous(memory_format=torch.channels_last_3d)
    r = torch.randn_like(x).contiguous(memory_format=torch.channels_last_3d) if res else None
    gy = torch.randn_like(x).contiguous(memory_format=torch.channels_last_3d)

    def make():
        bn = torch.nn.BatchNorm3d(c).to("cuda:0").train()
        with torch.no_grad():
            bn.weight.uniform_(0.5, 1.5); bn.bias.uniform_(-0.3, 0.3)
        return bn

    bn1, bn2 = make(), make()
    bn2.load_state_dict(bn1.state_dict())
    x1, x2 = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    r1 = r.clone().requires_grad_(True) if res else None
    r2 = r.clone().requires_grad_(True) if res else None
    with train_ops.bn_replicas(groups):
        y1 = train_ops.bn_act(x1, bn1, res=r1, relu=relu)
    y1.backward(gy)
    ys = []
    for g in range(groups):
        sl = slice(g * b // groups, (g + 1) * b // groups)
        ys.append(train_ops.bn_act(x2[sl], bn2, res=None if r2 is None else r2[sl], relu=relu))
        if g == 0:  # what a DataParallel run keeps: replica 0's buffers
            keep = (bn2.running_mean.clone(), bn2.running_var.clone(), int(bn2.num_batches_tracked))
    y2 = torch.cat(ys, 0)
    y2.backward(gy)
    assert torch.equal(y1, y2) and torch.equal(x1.grad, x2.grad)
    if res:
        assert torch.equal(r1.grad, r2.grad)
    assert torch.equal(bn1.running_mean, keep[0]) and torch.equal(bn1.running_var, keep[1])
    assert int(bn1.num_batches_tracked) == keep[2] == 1
    for p1, p2 in ((bn1.weight.grad, bn2.weight.grad), (bn1.bias.grad, bn2.bias.grad)):
        assert float((p1 - p2).abs().max()) <= 2e-6 * float(p2.abs().max()) + 1e-7


@pytest.mark.parametrize("c_a,c_b,shape,groups,res", [
    (64, 16, (2, 4, 14, 14), 1, False),    # pool-like producer + lateral
    (256, 64, (4, 2, 7, 7), 2, True),      # block exit (+ shortcut) + lateral, replica groups
    (1024, 256, (2, 2, 5, 3), 1, True),    # rows wider than one workgroup (C / 4 > 256) in the first slice
    (8, 8, (3, 8, 9, 7), 3, False),        # narrow, ragged
])
def test_concatenation_without_the_copy_equals_torch_cat(c_a, c_b, shape, groups, res):
    """The lateral fusion's torch.cat([slow, lateral], 1) with both producers writing slices of one buffer (bn_act cat_extra /
    cat_into, join_channels) and their backward passes reading slices of its gradient through a leading dimension: the same
    numbers, bit for bit, as the tensors of their own + torch.cat + contiguous gradients (SlowFast's FuseFastToSlow under
    train.py:114-141)."""
    from avtex import train_ops
    dev = "cuda:0"
    b, t, h, w = shape
    torch.manual_seed(c_a + c_b)
    xa0 = _cl(torch.randn(b, c_a, t, h, w, device=dev) * 1.5 + 0.3)
    xb0 = _cl(torch.randn(b, c_b, t, h, w, device=dev) - 0.2)
    r0 = _cl(torch.randn(b, c_a, t, h, w, device=dev)) if res else None
    gy = _cl(torch.randn(b, c_a + c_b, t, h, w, device=dev))
    wmix = torch.randn(c_a + c_b, device=dev).view(1, -1, 1, 1, 1)

    def run(join):
        train_ops._JOIN = join
        try:
            bna, bnb = nn.BatchNorm3d(c_a).to(dev).train(), nn.BatchNorm3d(c_b).to(dev).train()
            xa, xb = xa0.clone().requires_grad_(True), xb0.clone().requires_grad_(True)
            r = r0.clone().requires_grad_(True) if res else None
            with train_ops.bn_replicas(groups):
                ya = train_ops.bn_act(xa, bna, res=r, relu=True, cat_extra=c_b)
                tag = getattr(ya, "_avt_cat", None)
                assert (tag is not None) == bool(join)
                yb = train_ops.bn_act(xb, bnb, relu=True, cat_into=None if tag is None else (tag[0], c_a))
            z = train_ops.join_channels(ya, yb)
            assert z.is_contiguous(memory_format=torch.channels_last_3d)
            if join:
                assert z.data_ptr() == ya.data_ptr() and yb.data_ptr() == ya.data_ptr() + 4 * c_a  # no copy was made
            (z * wmix).backward(gy)  # (a consumer whose gradient is a fresh contiguous tensor, as a convolution's is)
            return (z.detach().clone(), xa.grad, xb.grad, None if r is None else r.grad, bna.weight.grad, bna.bias.grad, bnb.weight.grad,
                    bnb.bias.grad, bna.running_mean.clone(), bnb.running_var.clone())
        finally:
            train_ops._JOIN = 1

    a, e = run(1), run(0)
    for u, v in zip(a, e):
        assert (u is None) == (v is None)
        if u is not None:
            assert torch.equal(u, v)
