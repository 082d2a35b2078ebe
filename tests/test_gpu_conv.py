"""MFMA implicit-GEMM convolution (csrc/conv_igemm.hip) and the fused SlowFast runner vs plain PyTorch fp32."""
import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _run(avt, dev, cin, cout, k, s, p, dims, relu, with_res, ld_extra=0):
    from avtex.fused_slowfast import Act, FusedConv

    torch.manual_seed(cin * 131 + cout)
    conv = nn.Conv3d(cin, cout, k, stride=s, padding=p, bias=False)
    bn = nn.BatchNorm3d(cout)
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5); bn.bias.uniform_(-0.5, 0.5)
        bn.running_mean.uniform_(-0.2, 0.2); bn.running_var.uniform_(0.5, 1.5)
    bn.eval()
    b, t, h, w = dims
    x = torch.randn(b, cin, t, h, w)
    xb = x.to(torch.bfloat16)
    fc = FusedConv(conv, bn, relu, dev)
    m_in = b * t * h * w
    buf = torch.zeros((m_in, cin + ld_extra), dtype=torch.bfloat16, device=dev)
    buf[:, :cin] = xb.permute(0, 2, 3, 4, 1).reshape(m_in, cin).to(dev)
    od = fc.out_dims(dims)
    m_out = od[0] * od[1] * od[2] * od[3]
    res = None
    ref = bn(conv(xb.float()))
    if with_res:
        r = torch.randn(m_out, cout).to(torch.bfloat16)
        res = Act(r.to(dev), od)
        ref = ref + r.float().view(od[0], od[1], od[2], od[3], cout).permute(0, 4, 1, 2, 3)
    if relu:
        ref = F.relu(ref)
    out = fc(Act(buf, dims, 0, cin), res=res)
    torch.cuda.synchronize()
    got = out.buf.float().cpu().view(od[0], od[1], od[2], od[3], cout).permute(0, 4, 1, 2, 3)
    err = (got - ref).abs().max().item()
    scale = ref.abs().max().item()
    return err, scale


@pytest.mark.parametrize("cin,cout,k,s,p,dims", [
    (64, 64, (1, 3, 3), (1, 1, 1), (0, 1, 1), (2, 4, 14, 14)),      # bottleneck b
    (128, 128, (1, 3, 3), (1, 2, 2), (0, 1, 1), (1, 3, 14, 14)),    # strided b
    (80, 64, (1, 1, 1), (1, 1, 1), (0, 0, 0), (2, 4, 7, 9)),        # a, Cin = 80 (after fusion), ragged M
    (256, 512, (1, 1, 1), (1, 2, 2), (0, 0, 0), (1, 4, 8, 8)),      # strided shortcut
    (320, 128, (3, 1, 1), (1, 1, 1), (1, 0, 0), (1, 8, 5, 5)),      # temporal a
    (8, 16, (7, 1, 1), (4, 1, 1), (3, 0, 0), (2, 32, 6, 6)),        # lateral fusion conv, Cout = 16
    (8, 8, (1, 3, 3), (1, 1, 1), (0, 1, 1), (1, 8, 12, 12)),        # fast pathway, 8 channels
    (32, 32, (3, 1, 1), (1, 1, 1), (1, 0, 0), (1, 16, 6, 6)),
    (512, 2048, (1, 1, 1), (1, 1, 1), (0, 0, 0), (1, 2, 7, 7)),
    (72, 128, (1, 1, 1), (1, 1, 1), (0, 0, 0), (1, 3, 9, 11)),      # wide tile (LDS-DMA path) with a K tail, ragged M
    (24, 136, (3, 3, 3), (1, 1, 1), (1, 1, 1), (2, 5, 6, 7)),       # 27 taps, Cout not a tile multiple
    (256, 256, (1, 3, 3), (1, 1, 1), (0, 1, 1), (2, 3, 14, 14)),    # GEMM-like: 256x256 LDS-DMA tile when enabled
    (128, 328, (3, 1, 1), (1, 1, 1), (1, 0, 0), (1, 8, 9, 7)),      # ... with an N tail (328 = 256 + 72) and ragged M
    (264, 512, (1, 1, 1), (1, 2, 2), (0, 0, 0), (1, 4, 10, 10)),    # ... K tail (264 = 4 K-steps + 8), strided
    (1024, 256, (3, 1, 1), (1, 1, 1), (1, 0, 0), (1, 4, 5, 5)),     # XB tile (fragment-order weights in registers): temporal
    (256, 328, (1, 3, 3), (1, 1, 1), (0, 1, 1), (1, 2, 9, 7)),      # ... N tail, ragged M
    (160, 256, (1, 3, 3), (1, 2, 2), (0, 1, 1), (2, 2, 12, 10)),    # ... 45 K units (three all-zero trailing units), strided
])
@pytest.mark.parametrize("relu,with_res", [(True, False), (True, True), (False, False)])
def test_conv_igemm_matches_torch(avt, dev, cin, cout, k, s, p, dims, relu, with_res):
    err, scale = _run(avt, dev, cin, cout, k, s, p, dims, relu, with_res)
    # bf16 inputs/weights/outputs, fp32 accumulate: error budget = bf16 rounding of weights (2^-9 rel per term,
    # random signs) + output rounding (2^-9 of the value)
    assert err < 0.02 * max(scale, 1.0), (err, scale)


def test_conv_writes_channel_slice(avt, dev):
    from avtex.fused_slowfast import Act, FusedConv

    torch.manual_seed(0)
    conv = nn.Conv3d(16, 32, (1, 1, 1), bias=False)
    fc = FusedConv(conv, None, False, dev)
    x = torch.randn(50, 16).to(torch.bfloat16).to(dev)
    wide = torch.full((50, 96), 7.0, dtype=torch.bfloat16, device=dev)
    fc(Act(x, (1, 2, 5, 5)), out=Act(wide, (1, 2, 5, 5), 40, 32))
    torch.cuda.synchronize()
    ref = x.float().cpu() @ conv.weight.detach().view(32, 16).to(torch.bfloat16).float().t()
    assert (wide[:, 40:72].float().cpu() - ref).abs().max() < 0.05
    assert (wide[:, :40] == 7).all() and (wide[:, 72:] == 7).all()  # neighbours untouched


def test_fused_slowfast_matches_module(avt, dev):
    """Whole encoder: MFMA runner vs the PyTorch module in fp32 on the same (randomised) weights."""
    from avtex.fused_slowfast import SlowFastMFMA
    from avtex.slowfast import SlowFast

    torch.manual_seed(3)
    m = SlowFast().eval()
    with torch.no_grad():
        for mod in m.modules():
            if isinstance(mod, nn.BatchNorm3d):
                mod.weight.uniform_(0.6, 1.2); mod.bias.uniform_(-0.1, 0.1)
                mod.running_mean.uniform_(-0.1, 0.1); mod.running_var.uniform_(0.8, 1.2)
    slow, fast = torch.randn(2, 3, 8, 224, 224), torch.randn(2, 3, 32, 224, 224)
    fused = SlowFastMFMA(m, dev)
    y = fused([slow.to(dev), fast.to(dev)]).cpu()
    with torch.no_grad():
        ref = m.to(dev).float()([slow.to(dev), fast.to(dev)]).cpu()
        ref16 = m.to(torch.bfloat16)([slow.to(dev, torch.bfloat16), fast.to(dev, torch.bfloat16)]).float().cpu()
    cos = F.cosine_similarity(y, ref, dim=1)
    cos16 = F.cosine_similarity(ref16, ref, dim=1)
    rel = ((y - ref).norm(dim=1) / ref.norm(dim=1)).max().item()
    rel16 = ((ref16 - ref).norm(dim=1) / ref.norm(dim=1)).max().item()
    print("fused vs fp32: cos", cos.tolist(), "rel", rel, "| torch bf16 vs fp32: cos", cos16.tolist(), "rel", rel16)
    assert y.shape == (2, 2304) and torch.isfinite(y).all()
    assert cos.min() > 0.999 and rel < max(2.5 * rel16, 0.02)


def test_clip_pack_ndhwc4_equals_ncthw(avt, dev):
    """The channels-last clip the MFMA stem reads holds exactly the values of the NCTHW clip (+ a zero channel)."""
    g = torch.Generator().manual_seed(9)
    W, S, n = 20, 4, 5
    frames = torch.randint(0, 256, ((n - 1) * S + W + 1, 96, 128, 3), generator=g, dtype=torch.uint8).to(dev)
    starts = np.arange(n) * S
    s0, f0 = avt.ops.clip_pack(frames, starts, W, out_hw=224, dtype=torch.bfloat16)
    s1, f1 = avt.ops.clip_pack(frames, starts, W, out_hw=224, dtype=torch.bfloat16, layout="ndhwc4")
    assert s1.shape == (n, 8, 224, 224, 4) and f1.shape == (n, 32, 224, 224, 4)
    assert torch.equal(s1[..., :3].permute(0, 4, 1, 2, 3), s0) and torch.equal(f1[..., :3].permute(0, 4, 1, 2, 3), f0)
    assert (s1[..., 3] == 0).all() and (f1[..., 3] == 0).all()


def test_maxpool_matches_torch(avt, dev):
    torch.manual_seed(2)
    x = torch.randn(3, 16, 4, 30, 22).to(torch.bfloat16)  # [B,C,T,H,W]
    ref = F.max_pool3d(x.float(), (1, 3, 3), (1, 2, 2), (0, 1, 1))
    rows = x.permute(0, 2, 3, 4, 1).reshape(-1, 16).contiguous().to(dev)
    ho, wo = ref.shape[3], ref.shape[4]
    out = torch.full((3 * 4 * ho * wo, 24), 5.0, dtype=torch.bfloat16, device=dev)
    avt.ops.maxpool_hw3s2(rows.data_ptr(), out.data_ptr() + 2 * 8, 3 * 4, 30, 22, 16, 16, 24, tgroup=1)
    torch.cuda.synchronize()
    got = out[:, 8:24].float().cpu().view(3, 4, ho, wo, 16).permute(0, 4, 1, 2, 3)
    assert torch.equal(got, ref) and (out[:, :8] == 5).all()


def test_stem_on_mfma_matches_torch(avt, dev):
    """Pixel-pair stem (conv [kt,7,7] s2 + BN + ReLU) + max-pool vs the PyTorch stem module, both pathways."""
    from avtex.fused_slowfast import SlowFastMFMA
    from avtex.slowfast import SlowFast

    torch.manual_seed(4)
    m = SlowFast().eval()
    with torch.no_grad():
        for mod in m.s1.modules():
            if isinstance(mod, nn.BatchNorm3d):
                mod.weight.uniform_(0.6, 1.2); mod.bias.uniform_(-0.2, 0.2)
                mod.running_mean.uniform_(-0.2, 0.2); mod.running_var.uniform_(0.8, 1.2)
    fused = SlowFastMFMA(m, dev)
    for conv, stem, t in ((fused.stem_s, m.s1.pathway0_stem, 3), (fused.stem_f, m.s1.pathway1_stem, 8)):
        x = torch.randn(2, 3, t, 64, 48).to(torch.bfloat16)
        with torch.no_grad():
            ref = stem(x.float())
        clip = torch.zeros((2, t, 64, 48, 4), dtype=torch.bfloat16)
        clip[..., :3] = x.permute(0, 2, 3, 4, 1)
        act, pd = fused._stem(conv, clip.to(dev))
        torch.cuda.synchronize()
        got = act.buf.float().cpu().view(*pd, conv.frame_channels).permute(0, 4, 1, 2, 3)
        assert got.shape == ref.shape
        assert (got - ref).abs().max() < 0.03 * max(ref.abs().max().item(), 1.0)


def test_maxpool2x2_matches_torch(avt, dev):
    """VGGish pool (MaxPool2d(2,2), floor mode: an odd last row is dropped) — exact."""
    torch.manual_seed(5)
    for h, w in ((100, 64), (25, 16), (7, 5)):
        x = torch.randn(3, 24, h, w).to(torch.bfloat16)  # [B,C,H,W]
        ref = F.max_pool2d(x.float(), 2, 2)
        rows = x.permute(0, 2, 3, 1).reshape(-1, 24).contiguous().to(dev)
        out = torch.empty((3 * (h // 2) * (w // 2), 24), dtype=torch.bfloat16, device=dev)
        avt.ops.maxpool_hw2s2(rows.data_ptr(), out.data_ptr(), 3, h, w, 24, 24, 24)
        torch.cuda.synchronize()
        assert torch.equal(out.float().cpu().view(3, h // 2, w // 2, 24).permute(0, 3, 1, 2), ref)


def test_fused_vggish_matches_module(avt, dev):
    """VGGish feature stack on the MFMA convolution vs the fp32 plugin module (audio_models/vggish.py:42-46 contract:
    NHWC flatten, fc skipped).  Tolerance: bf16 activations — compared with torch's own bf16 run of the module."""
    from avtex.fused_vggish import VGGishMFMA
    from avtex.vggish import VGGish

    torch.manual_seed(6)
    m = VGGish().eval()
    with torch.no_grad():
        for mod in m.features:
            if isinstance(mod, nn.Conv2d):  # He-style init so activations neither vanish nor explode over 6 layers
                nn.init.kaiming_normal_(mod.weight, nonlinearity="relu")
                mod.bias.uniform_(-0.1, 0.1)
    x = torch.randn(5, 1, 100, 64) * 2 - 1  # log-mel range
    fused = VGGishMFMA(m, dev)
    y = fused(x).cpu()
    with torch.no_grad():
        ref = m.to(dev)(x.to(dev)).cpu()
        ref16 = m.to(torch.bfloat16)(x.to(dev, torch.bfloat16)).float().cpu()
    assert y.shape == (5, 12288) and torch.isfinite(y).all()
    cos = F.cosine_similarity(y, ref, dim=1)
    rel = ((y - ref).norm(dim=1) / ref.norm(dim=1)).max().item()
    rel16 = ((ref16 - ref).norm(dim=1) / ref.norm(dim=1)).max().item()
    print("vggish fused vs fp32: cos", cos.tolist(), "rel", rel, "| torch bf16 vs fp32 rel", rel16)
    assert cos.min() > 0.9995 and rel < max(2.5 * rel16, 0.02)
    # first layer alone (1 input channel, 8 mel bins per 16-byte chunk): bf16 inputs/weights, fp32 accumulate
    conv, _, cout = fused.layers[0]
    xb = x.to(torch.bfloat16)
    w0 = m.features[0].weight.float().to(torch.bfloat16).float()  # the module is bf16 by now; same rounding as the packer
    ref0 = F.relu(F.conv2d(xb.float().to(dev), w0.to(dev), m.features[0].bias.float().to(dev), padding=1)).cpu()
    from avtex.fused_slowfast import Act
    a = conv(Act(xb.to(dev).contiguous().view(5 * 100 * 8, 8), (5, 1, 100, 8)))
    got = a.buf.view(5, 100, 64, cout).permute(0, 3, 1, 2).float().cpu()
    assert (got - ref0).abs().max() < 0.01 * max(ref0.abs().max().item(), 1.0)


@pytest.mark.parametrize("hw", [224, 64])
def test_stem_lds_kernel_matches_generic_and_torch(avt, dev, hw):
    """csrc/stem_conv.hip (input patch resident in LDS, 16x16x32 MFMA) against the implicit-GEMM kernel on the same
    packed weights (same products, different fp32 summation order -> <= 1 bf16 ulp) and against the PyTorch stem
    module, both pathways, including the clip's first/last frames (frame taps outside the clip are skipped)."""
    from avtex.fused_slowfast import Act, SlowFastMFMA
    from avtex.slowfast import SlowFast

    torch.manual_seed(8)
    m = SlowFast().eval()
    with torch.no_grad():
        for mod in m.s1.modules():
            if isinstance(mod, nn.BatchNorm3d):
                mod.weight.uniform_(0.6, 1.2); mod.bias.uniform_(-0.2, 0.2)
                mod.running_mean.uniform_(-0.2, 0.2); mod.running_var.uniform_(0.8, 1.2)
    fused = SlowFastMFMA(m, dev)
    for conv, stem, t in ((fused.stem_s, m.s1.pathway0_stem, 3), (fused.stem_f, m.s1.pathway1_stem, 8)):
        assert avt.ops.stem_conv_supported(hw, hw // 2, conv.cout)
        x = torch.randn(2, 3, t, hw, hw).to(torch.bfloat16)
        clip = torch.zeros((2, t, hw, hw, 4), dtype=torch.bfloat16)
        clip[..., :3] = x.permute(0, 2, 3, 4, 1)
        clip = clip.to(dev)
        xa = Act(clip.view(-1, 8), (2, t, hw, hw // 2))
        gen = conv(xa)  # implicit GEMM
        od = gen.dims
        y = torch.full((od[0] * od[1] * od[2] * od[3], conv.cout), 7.0, dtype=torch.bfloat16, device=dev)
        avt.ops.stem_conv(xa.ptr, conv.wt_lds, conv.bias, y.data_ptr(), 2, t, hw, hw // 2, conv.cout, conv.kernel[0],
                          conv.stride[0], conv.pad[0])
        torch.cuda.synchronize()
        a, g = y.float(), gen.buf.float()
        scale = max(g.abs().max().item(), 1.0)
        assert (a - g).abs().max().item() <= 0.01 * scale
        assert (a != g).float().mean().item() < 0.05  # a few last-bit differences only
        # fused max-pool variant == the unfused kernel followed by the pool kernel, bit for bit (into a channel slice)
        tg, cf = conv.tgroup, conv.frame_channels
        hp, wp = od[2] // 2, od[3] // 2
        ref_pool = torch.empty((2 * t * hp * wp, cf), dtype=torch.bfloat16, device=dev)
        avt.ops.maxpool_hw3s2(y.data_ptr(), ref_pool.data_ptr(), 2 * od[1], od[2], od[3], conv.cout, conv.cout, cf, tgroup=tg)
        wide = torch.full((2 * t * hp * wp, cf + 16), 3.0, dtype=torch.bfloat16, device=dev)
        avt.ops.stem_conv_pool(xa.ptr, conv.wt_lds, conv.bias, wide.data_ptr() + 2 * 8, 2, t, hw, hw // 2, conv.cout,
                               conv.kernel[0], conv.stride[0], conv.pad[0], tg, cf + 16)
        torch.cuda.synchronize()
        assert torch.equal(wide[:, 8 : 8 + cf], ref_pool)
        assert (wide[:, :8] == 3).all() and (wide[:, 8 + cf :] == 3).all()
        # through the pool, against torch (as test_stem_on_mfma_matches_torch does for the generic path)
        act, pd = fused._stem(conv, clip)
        with torch.no_grad():
            ref = stem(x.float())
        got = act.buf.float().cpu().view(*pd, conv.frame_channels).permute(0, 4, 1, 2, 3)
        assert got.shape == ref.shape
        assert (got - ref).abs().max() < 0.03 * max(ref.abs().max().item(), 1.0)


@pytest.mark.parametrize("c,cm,dims,tchunk", [
    (32, 8, (2, 7, 11, 12), 3),    # ragged strips (5+5+1 rows), frame chunks 3+3+1, partial tiles / DMA instructions
    (64, 16, (1, 5, 9, 10), 8),    # 128-byte records, one chunk of frames
    (32, 8, (1, 4, 56, 56), 2),    # the res2 fast-pathway shape
    (64, 16, (1, 3, 28, 28), 8),   # the res3 fast-pathway shape
    (128, 32, (2, 5, 7, 6), 2),    # wide form (weights in LDS, single-tap k-steps), ragged strips 3+3+1
    (128, 32, (1, 4, 14, 14), 8),  # the res4 fast-pathway shape
])
def test_bottleneck_fused_matches_module_and_unfused(avt, dev, c, cm, dims, tchunk):
    """csrc/bottleneck_fused.hip: a whole identity bottleneck ([3,1,1] -> [1,3,3] -> [1,1,1] + x, BN folded, ReLUs) in one
    kernel vs the PyTorch block in fp32 (bf16 intermediates: tolerance) and vs the three-launch MFMA path."""
    import avtex.fused_slowfast as fsf
    from avtex.slowfast import ResBlock

    torch.manual_seed(c + dims[1])
    blk = ResBlock(c, c, cm, 3, 1).eval()
    with torch.no_grad():
        for mod in blk.modules():
            if isinstance(mod, nn.BatchNorm3d):
                mod.weight.uniform_(0.6, 1.2); mod.bias.uniform_(-0.2, 0.2)
                mod.running_mean.uniform_(-0.2, 0.2); mod.running_var.uniform_(0.8, 1.2)
    assert not hasattr(blk, "branch1")
    b, t, h, w = dims
    x = torch.randn(b, c, t, h, w).to(torch.bfloat16)
    with torch.no_grad():
        ref = blk(x.float())
    rows = x.permute(0, 2, 3, 4, 1).reshape(-1, c).contiguous().to(dev)
    fb = fsf._Block(blk, dev)
    assert fb.fused is not None and avt.ops.bottleneck_fused_supported(c, w)
    y = torch.full((rows.shape[0], c), 9.0, dtype=torch.bfloat16, device=dev)
    avt.ops.bottleneck_fused(rows.data_ptr(), y.data_ptr(), fb.fused, b, t, h, w, c, tchunk=tchunk)
    torch.cuda.synchronize()
    got = y.float().cpu().view(b, t, h, w, c).permute(0, 4, 1, 2, 3)
    scale = max(ref.abs().max().item(), 1.0)
    assert (got - ref).abs().max().item() < 0.03 * scale
    # the unfused path: same folded weights, intermediates rounded to bf16 in HBM instead of LDS -> near-identical
    fused_flag, fb.fused = fb.fused, None
    un = fb(fsf.Act(rows, dims)).buf.float().cpu().view(b, t, h, w, c).permute(0, 4, 1, 2, 3)
    fb.fused = fused_flag
    assert (got - un).abs().max().item() < 0.02 * scale
    assert (got - un).abs().mean().item() < 1e-3 * scale
    # through _Block.__call__ (dispatch) and independent of the frame chunking
    y2 = fb(fsf.Act(rows, dims))
    y3 = torch.empty_like(y)
    avt.ops.bottleneck_fused(rows.data_ptr(), y3.data_ptr(), fb.fused, b, t, h, w, c, tchunk=t)
    torch.cuda.synchronize()
    assert torch.equal(y2.buf, y3) and torch.equal(y, y3)


def test_mean_positions_head_pool(avt, dev):
    """The head's global average pool kernel: bf16 rows -> fp32 means into a column slice; deterministic."""
    torch.manual_seed(12)
    for b, p, c, ld in ((3, 392, 2048, 2048), (2, 1568, 256, 256), (2, 7, 24, 40)):
        x = torch.randn(b * p, ld).to(torch.bfloat16).to(dev)
        out = torch.full((b, c + 16), -1.0, dtype=torch.float32, device=dev)
        avt.ops.mean_positions(x.data_ptr(), b, p, c, ld, out, 8)
        out2 = torch.full_like(out, -1.0)
        avt.ops.mean_positions(x.data_ptr(), b, p, c, ld, out2, 8)
        ref = x[:, :c].float().view(b, p, c).double().mean(1).float()
        assert torch.allclose(out[:, 8 : 8 + c], ref, rtol=1e-5, atol=1e-6)
        assert (out[:, :8] == -1).all() and (out[:, 8 + c :] == -1).all() and torch.equal(out, out2)


@pytest.mark.parametrize("cin,c,cm,stride,dims,tchunk", [
    (8, 32, 8, 1, (2, 7, 11, 12), 3), (8, 32, 8, 1, (1, 4, 56, 56), 8),           # res2's first fast block
    (32, 64, 16, 2, (2, 5, 10, 12), 2), (32, 64, 16, 2, (1, 3, 56, 56), 8),        # res3's: b and shortcut stride 2
    (64, 128, 32, 2, (1, 4, 6, 8), 3), (64, 128, 32, 2, (1, 3, 28, 28), 8),        # res4's (wide form)
])
def test_bottleneck_first_block_matches_module_and_unfused(avt, dev, cin, c, cm, stride, dims, tchunk):
    """The first fast-pathway blocks (1x1x1 shortcut conv; res3 / res4: spatial stride 2) through the fused kernel."""
    import avtex.fused_slowfast as fsf
    from avtex.slowfast import ResBlock

    torch.manual_seed(dims[2] + cin)
    blk = ResBlock(cin, c, cm, 3, stride).eval()
    with torch.no_grad():
        for mod in blk.modules():
            if isinstance(mod, nn.BatchNorm3d):
                mod.weight.uniform_(0.6, 1.2); mod.bias.uniform_(-0.2, 0.2)
                mod.running_mean.uniform_(-0.2, 0.2); mod.running_var.uniform_(0.8, 1.2)
    assert hasattr(blk, "branch1")
    b, t, h, w = dims
    x = torch.randn(b, cin, t, h, w).to(torch.bfloat16)
    with torch.no_grad():
        ref = blk(x.float())
    rows = x.permute(0, 2, 3, 4, 1).reshape(-1, cin).contiguous().to(dev)
    fb = fsf._Block(blk, dev)
    assert fb.fused_first is not None and fb.fused is None
    y = fb(fsf.Act(rows, dims))
    torch.cuda.synchronize()
    ho, wo = h // stride, w // stride
    assert y.dims == (b, t, ho, wo)
    got = y.buf.float().cpu().view(b, t, ho, wo, c).permute(0, 4, 1, 2, 3)
    scale = max(ref.abs().max().item(), 1.0)
    assert (got - ref).abs().max().item() < 0.03 * scale
    keep, fb.fused_first = fb.fused_first, None
    un = fb(fsf.Act(rows, dims)).buf.float().cpu().view(b, t, ho, wo, c).permute(0, 4, 1, 2, 3)
    fb.fused_first = keep
    assert (got - un).abs().max().item() < 0.02 * scale and (got - un).abs().mean().item() < 1e-3 * scale
    y2 = torch.empty_like(y.buf)
    avt.ops.bottleneck_first(rows.data_ptr(), y2.data_ptr(), fb.fused_first, b, t, h, w, cin, c, tchunk=tchunk)
    torch.cuda.synchronize()
    assert torch.equal(y.buf, y2)  # independent of the frame chunking


@pytest.mark.parametrize("dims,relu", [((2, 3, 10, 16), True), ((1, 2, 56, 56), True), ((1, 1, 7, 16), False)])
def test_conv33_c64_matches_igemm_and_torch(avt, dev, dims, relu):
    """csrc/conv33_c64.hip ([1,3,3] 64 -> 64 with the input strip resident in LDS) vs the implicit-GEMM kernel on the same
    folded weights and vs torch; ragged last strip, writes into a channel slice."""
    from avtex.fused_slowfast import Act, FusedConv, pack_c33

    torch.manual_seed(dims[2] * 7)
    conv = nn.Conv3d(64, 64, (1, 3, 3), padding=(0, 1, 1), bias=False)
    bn = nn.BatchNorm3d(64).eval()
    with torch.no_grad():
        bn.weight.uniform_(0.6, 1.2); bn.bias.uniform_(-0.2, 0.2)
        bn.running_mean.uniform_(-0.2, 0.2); bn.running_var.uniform_(0.8, 1.2)
    fc = FusedConv(conv, bn, relu, dev)
    b, t, h, w = dims
    x = torch.randn(b, 64, t, h, w).to(torch.bfloat16)
    rows = x.permute(0, 2, 3, 4, 1).reshape(-1, 64).contiguous().to(dev)
    gen = fc(Act(rows, dims)).buf
    wide = torch.full((rows.shape[0], 64 + 16), 4.0, dtype=torch.bfloat16, device=dev)
    avt.ops.conv33_c64(rows.data_ptr(), pack_c33(fc._folded[0], dev), fc.bias, wide.data_ptr() + 2 * 8, b, t, h, w, 80, relu=relu)
    torch.cuda.synchronize()
    got = wide[:, 8:72]
    assert (wide[:, :8] == 4).all() and (wide[:, 72:] == 4).all()
    scale = max(gen.float().abs().max().item(), 1.0)
    assert (got.float() - gen.float()).abs().max().item() <= 0.01 * scale
    assert (got != gen).float().mean().item() < 0.05  # same products, different fp32 summation order
    with torch.no_grad():
        ref = bn(conv(x.float()))
        ref = F.relu(ref) if relu else ref
    g5 = got.float().cpu().view(b, t, h, w, 64).permute(0, 4, 1, 2, 3)
    assert (g5 - ref).abs().max().item() < 0.02 * max(ref.abs().max().item(), 1.0)


@pytest.mark.parametrize("k1,n1,n2,has_res,m,k2x", [(64, 256, 64, True, 16 * 50 + 5, 0), (144, 256, 64, False, 16 * 37 + 11, 0),
                                                     (128, 512, 128, True, 16 * 41 + 1, 0), (64, 256, 64, True, 7, 0),
                                                     (64, 256, 128, True, 16 * 23 + 9, 64)])
def test_pw_chain_matches_two_launches_and_torch(avt, dev, k1, n1, n2, has_res, m, k2x):
    """csrc/pw_chain.hip (block i's c + residual + ReLU and block i+1's a + ReLU in one pass, y in registers between
    the GEMMs) vs the two implicit-GEMM launches on the same folded weights and vs fp32 torch; ragged last tile,
    row strides wider than the channels, nothing written outside the slices."""
    from avtex.fused_slowfast import Act, FusedConv, pack_pw

    torch.manual_seed(k1 + m)
    def layer(cin, cout):
        conv = nn.Conv3d(cin, cout, 1, bias=False)
        bn = nn.BatchNorm3d(cout).eval()
        with torch.no_grad():
            bn.weight.uniform_(0.6, 1.2); bn.bias.uniform_(-0.2, 0.2)
            bn.running_mean.uniform_(-0.2, 0.2); bn.running_var.uniform_(0.8, 1.2)
        return conv, bn
    (c1, bn1), (c2, bn2) = layer(k1, n1), layer(n1 + k2x, n2)
    f1, f2 = FusedConv(c1, bn1, True, dev), FusedConv(c2, bn2, True, dev)
    dims = (1, 1, 1, m)
    ldx, ldr, ldy, ldz = k1 + 16, n1 + 8, n1 + 24 + k2x, n2 + 8
    xw = torch.randn(m, ldx).to(torch.bfloat16).to(dev)
    rw = torch.randn(m, ldr).to(torch.bfloat16).to(dev)
    x, r = Act(xw, dims, 8, k1), Act(rw, dims, 8, n1)
    yw = torch.full((m, ldy), 3.0, dtype=torch.bfloat16, device=dev)
    if k2x:  # the second layer reads [y | x2] from one row buffer (the stage-boundary concat)
        yw[:, 16 + n1:16 + n1 + k2x] = torch.randn(m, k2x).to(torch.bfloat16).to(dev)
    x2 = Act(yw, dims, 16 + n1, k2x) if k2x else None
    x2_keep = yw[:, 16 + n1:16 + n1 + k2x].clone()
    y_ref = f1(x, res=r if has_res else None, relu=True)
    if k2x:
        cat = torch.cat([y_ref.buf, x2_keep], 1).contiguous()
        z_ref = f2(Act(cat, dims))
    else:
        z_ref = f2(y_ref)
    zw = torch.full((m, ldz), 3.0, dtype=torch.bfloat16, device=dev)
    assert avt.ops.pw_chain_supported(k1, n1, n2, has_res, k2x)
    avt.ops.pw_chain(x.ptr, ldx, k1, pack_pw(f1._folded[0], dev), f1.bias, r.ptr if has_res else 0, ldr if has_res else 0,
                     yw.data_ptr() + 2 * 16, ldy, n1, pack_pw(f2._folded[0], dev), f2.bias, zw.data_ptr() + 2 * 8, ldz, n2, m,
                     x2_ptr=x2.ptr if k2x else 0, ldx2=ldy if k2x else 0, k2x=k2x)
    torch.cuda.synchronize()
    assert (yw[:, :16] == 3).all() and (yw[:, 16 + n1 + k2x:] == 3).all() and (zw[:, :8] == 3).all()
    assert torch.equal(yw[:, 16 + n1:16 + n1 + k2x], x2_keep)
    y, z = yw[:, 16:16 + n1], zw[:, 8:8 + n2]
    for got, ref in ((y, y_ref.buf), (z, z_ref.buf)):
        scale = max(ref.float().abs().max().item(), 1.0)
        assert (got.float() - ref.float()).abs().max().item() <= 0.02 * scale
    # the implicit GEMM rounds conv + bias to bf16 BEFORE the residual add (two roundings); the chained pass adds the
    # residual in fp32 (one rounding).  Pin it against an fp64 evaluation on the same bf16 weights instead:
    ye = xw[:, 8:8 + k1].double() @ f1.wt.double().t() + f1.bias.double()
    if has_res:
        ye = ye + rw[:, 8:8 + n1].double()
    ye = ye.clamp_min(0).float().to(torch.bfloat16)
    assert (y != ye).float().mean().item() < 0.01  # only fp32 summation-order effects at rounding boundaries
    ze = (torch.cat([y, x2_keep], 1).double() @ f2.wt.double().t() + f2.bias.double()).clamp_min(0).float().to(torch.bfloat16)
    assert (z != ze).float().mean().item() < 0.01
    if not has_res:
        assert (y != y_ref.buf).float().mean().item() < 0.05  # same products, different fp32 summation order
    with torch.no_grad():
        xin = xw[:, 8:8 + k1].float().cpu().t().reshape(1, k1, 1, 1, m)
        t = bn1(c1(xin))
        if has_res:
            t = t + rw[:, 8:8 + n1].float().cpu().t().reshape(1, n1, 1, 1, m)
        t = F.relu(t)
        if k2x:
            t = torch.cat([t, x2_keep.float().cpu().t().reshape(1, k2x, 1, 1, m)], 1)
        t2 = F.relu(bn2(c2(t)))
        t = t[:, :n1]
    assert (y.float().cpu().t().reshape(1, n1, 1, 1, m) - t).abs().max().item() < 0.03 * max(t.abs().max().item(), 1.0)
    assert (z.float().cpu().t().reshape(1, n2, 1, 1, m) - t2).abs().max().item() < 0.03 * max(t2.abs().max().item(), 1.0)
    assert not avt.ops.pw_chain_supported(64, 256, 64, False) and not avt.ops.pw_chain_supported(64, 256, 64, True, 64)
    from avtex._lib import AvtError
    with pytest.raises(AvtError):
        avt.ops.pw_chain(x.ptr, ldx, 72, pack_pw(f1._folded[0], dev), f1.bias, 0, 0, yw.data_ptr(), ldy, n1,
                         pack_pw(f2._folded[0], dev), f2.bias, zw.data_ptr(), ldz, n2, m)


def test_chained_pointwise_passes_are_used_and_agree_with_unchained(avt, dev, monkeypatch):
    """The slow pathway's chained passes (csrc/pw_chain.hip) against the same runner with fused_slowfast._CHAIN off: six passes
    per forward (res2: first block, identity block, the res2 -> res3 boundary; res3: three), embeddings agree to bf16
    rounding (the chained pass adds the residual before its one rounding, the implicit GEMM after its first)."""
    import avtex.fused_slowfast as fsf
    from avtex.slowfast import SlowFast

    torch.manual_seed(11)
    m = SlowFast().eval()
    with torch.no_grad():
        for mod in m.modules():
            if isinstance(mod, nn.BatchNorm3d):
                mod.weight.uniform_(0.6, 1.2); mod.bias.uniform_(-0.1, 0.1)
                mod.running_mean.uniform_(-0.1, 0.1); mod.running_var.uniform_(0.8, 1.2)
    slow, fast = torch.randn(2, 3, 8, 96, 64).to(dev), torch.randn(2, 3, 32, 96, 64).to(dev)
    fused = fsf.SlowFastMFMA(m, dev)
    calls = []
    orig = fsf.ops.pw_chain
    monkeypatch.setattr(fsf.ops, "pw_chain", lambda *a, **k: (calls.append((a[2], a[9], a[14], k.get("k2x", 0))), orig(*a, **k))[1])
    rows = []
    orig_conv = fsf.ops.conv3d_igemm
    monkeypatch.setattr(fsf.ops, "conv3d_igemm", lambda *a, **k: (rows.append(k.get("out_rows")), orig_conv(*a, **k))[1])
    y1 = fused([slow, fast]).cpu()
    # res3's first block folds its strided shortcut into c (K-concatenation), so only blocks 1->2 and 2->3 chain there
    assert sorted(calls) == sorted([(144, 256, 64, 0), (64, 256, 64, 0), (64, 256, 128, 64)] + [(128, 512, 128, 0)] * 2)
    assert sorted(r for r in rows if r is not None) == [(2, 6, 4), (2, 12, 8), (2, 24, 16)]  # the strided b convs of res5, res4, res3
    monkeypatch.setattr(fsf, "_CHAIN", 0)
    monkeypatch.setattr(fsf, "_FUSE_SCAT", 0)
    plain = fsf.SlowFastMFMA(m, dev)
    n, rows[:] = len(calls), []
    y0 = plain([slow, fast]).cpu()
    assert len(calls) == n and all(r is None for r in rows)
    cos = F.cosine_similarity(y1, y0, dim=1)
    assert cos.min() > 0.9999 and ((y1 - y0).norm(dim=1) / y0.norm(dim=1)).max().item() < 0.01


@pytest.mark.parametrize("c,dims", [(128, (1, 2, 12, 8)), (256, (2, 1, 10, 6))])
def test_conv_rows_remap_writes_behind_the_input_channels(avt, dev, c, dims):
    """avt_conv3d_igemm_rows_bf16: a stride-2 [1,3,3] conv writing output position (f, ho, wo) to row (f*H + 2ho)*W + 2wo
    of a wider buffer (128 channels: the 128x128 tile; 256: the LDS-DMA tile) — equal to the plain launch scattered by
    hand, every other byte untouched; a residual or a grid too small for the remap is refused."""
    from avtex.fused_slowfast import Act, FusedConv
    from avtex._lib import AvtError

    torch.manual_seed(c)
    conv = nn.Conv3d(c, c, (1, 3, 3), stride=(1, 2, 2), padding=(0, 1, 1), bias=False)
    bn = nn.BatchNorm3d(c).eval()
    fc = FusedConv(conv, bn, True, dev)
    b, t, h, w = dims
    x = torch.randn(b * t * h * w, c).to(torch.bfloat16).to(dev)
    plain = fc(Act(x, dims))
    ho, wo = plain.dims[2], plain.dims[3]
    wide = torch.full((b * t * h * w, c + 24), 7.0, dtype=torch.bfloat16, device=dev)
    fc(Act(x, dims), out=Act(wide, dims, 8, c), out_rows=(2, h, w))
    torch.cuda.synchronize()
    want = torch.full_like(wide, 7.0)
    rows = ((torch.arange(b * t).view(-1, 1, 1) * h + 2 * torch.arange(ho).view(1, -1, 1)) * w + 2 * torch.arange(wo).view(1, 1, -1)).reshape(-1)
    want[rows.to(dev), 8:8 + c] = plain.buf
    assert torch.equal(wide, want)
    with pytest.raises(AvtError):
        fc(Act(x, dims), out=Act(wide, dims, 8, c), out_rows=(2, ho, wo))  # grid too small for 2 x (ho, wo)
    with pytest.raises(AvtError):
        fc(Act(x, dims), out=Act(wide, dims, 8, c), res=plain, out_rows=(2, h, w))


@pytest.mark.parametrize("cin,cout,k,p,dims", [(256, 256, (1, 3, 3), (0, 1, 1), (2, 3, 14, 14)),
                                                (1024, 264, (3, 1, 1), (1, 0, 0), (1, 4, 6, 5))])
def test_xb_and_xl_tiles_agree(avt, dev, cin, cout, k, p, dims):
    """The two long-K tiles on the same layer: XB (fragment-order weights straight into registers, the default) and XL (both
    operands through the LDS-DMA ring, what a caller without packed weights gets) — same products, different fp32
    summation order only in the K walk they share, so the bf16 outputs agree almost everywhere; + residual path."""
    from avtex.fused_slowfast import Act, FusedConv

    torch.manual_seed(cin + cout)
    conv = nn.Conv3d(cin, cout, k, padding=p, bias=False)
    bn = nn.BatchNorm3d(cout).eval()
    fc = FusedConv(conv, bn, True, dev)
    assert fc.wfrag is not None and avt.ops.conv3d_wfrag_supported(cin, cout, k)
    b, t, h, w = dims
    x = Act(torch.randn(b * t * h * w, cin).to(torch.bfloat16).to(dev), dims)
    res = Act(torch.randn(b * t * h * w, cout).to(torch.bfloat16).to(dev), dims)
    xb, xb_r = fc(x).buf.clone(), fc(x, res=res).buf.clone()
    keep, fc.wfrag = fc.wfrag, None
    xl, xl_r = fc(x).buf.clone(), fc(x, res=res).buf.clone()
    fc.wfrag = keep
    torch.cuda.synchronize()
    for got, ref in ((xb, xl), (xb_r, xl_r)):
        scale = max(ref.float().abs().max().item(), 1.0)
        assert (got.float() - ref.float()).abs().max().item() <= 0.01 * scale
        assert (got != ref).float().mean().item() < 0.02
