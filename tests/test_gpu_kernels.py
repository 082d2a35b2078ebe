"""Parity of every HIP kernel (through the C ABI) against the CPU oracle.

Bar: bit-exact for indices, counts and for the canonical fp32 paths (l2norm,
AVT_SIM_F32, survivor probabilities); stated tolerances for the bf16 MFMA modes
and for transcendental reporting values (ce, entropy)."""
import os

import numpy as np
import pytest
import torch

from oracle import cref, ref_py

pytestmark = pytest.mark.gpu


def _bits(t):
    return t.view(torch.int16).cpu().numpy().view(np.uint16)


def _rand(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).numpy().astype(np.float32)


# ---------------------------------------------------------------- l2norm
@pytest.mark.parametrize("n,d0,d1", [(64, 2304, 0), (33, 2304, 12288), (7, 64, 0), (5, 37, 11), (3, 4100, 0), (1, 1, 0)])
def test_l2norm_bit_exact(avt, dev, n, d0, d1):
    x0 = _rand((n, d0), 1)
    x1 = _rand((n, d1), 2, 3.0) if d1 else None
    y, hi, lo = cref.l2norm_rows(x0, x1)
    gy, ghi, glo = avt.ops.l2norm_rows(torch.from_numpy(x0).to(dev), torch.from_numpy(x1).to(dev) if d1 else None,
                                       want_split=True)
    assert np.array_equal(gy.cpu().numpy().view(np.uint32), y.view(np.uint32))
    assert np.array_equal(_bits(ghi), hi)
    assert np.array_equal(_bits(glo), lo)


def test_l2norm_zero_row_uses_eps(avt, dev):
    x = np.zeros((2, 64), np.float32)
    x[1, 3] = 2.0
    gy, _, _ = avt.ops.l2norm_rows(torch.from_numpy(x).to(dev))
    y, _, _ = cref.l2norm_rows(x, want_split=False)
    assert np.array_equal(gy.cpu().numpy(), y)
    assert not np.isnan(y).any()


# ---------------------------------------------------------------- similarity
@pytest.mark.parametrize("nq,nt,d", [(128, 128, 64), (200, 333, 2304), (1, 517, 2304), (64, 64, 14592), (37, 53, 100),
                                     (5, 9, 7)])
def test_sim_f32_bit_exact(avt, dev, nq, nt, d):
    q, _, _ = cref.l2norm_rows(_rand((nq, d), 3), want_split=False)
    t, _, _ = cref.l2norm_rows(_rand((nt, d), 4), want_split=False)
    ref = cref.sim_f32(q, t, 0.1)
    out = avt.ops.sim_gemm_nt(torch.from_numpy(q).to(dev), torch.from_numpy(t).to(dev), 0.1, "f32").cpu().numpy()
    assert np.array_equal(out.view(np.uint32), ref.view(np.uint32))


def test_sim_f32_identity_asymmetric(avt, dev):
    """A = I against an asymmetric B catches a transposed C write (guide §3)."""
    n = 128
    q = np.eye(n, dtype=np.float32)
    t = (np.arange(n * n, dtype=np.float32).reshape(n, n) % 97) / 7.0
    out = avt.ops.sim_gemm_nt(torch.from_numpy(q).to(dev), torch.from_numpy(t).to(dev), 1.0, "f32").cpu().numpy()
    assert np.array_equal(out, t.T)


@pytest.mark.parametrize("mode,tol", [("bf16x3", 2e-5), ("bf16", 4e-3)])
@pytest.mark.parametrize("nq,nt,d", [(256, 384, 2304), (70, 45, 200), (9, 5, 13),
                                     (300, 512, 2304), (1024, 768, 256)])  # (bf16x3: the 256 x 256 LDS-DMA tile — ragged rows, the shortest K)
def test_sim_bf16_modes(avt, dev, mode, tol, nq, nt, d):
    """bf16 MFMA modes: vs their own fp64 emulation (tight) and vs the canonical fp32 scores.
    Score tolerance (cos/0.1): bf16x3 2e-5 << the 1e-3 contract; plain bf16 4e-3 (not used for index decisions)."""
    q, qh, ql = cref.l2norm_rows(_rand((nq, d), 5))
    t, th, tl = cref.l2norm_rows(_rand((nt, d), 6))
    tq = lambda a: torch.from_numpy(a.view(np.int16)).view(torch.bfloat16).to(dev)
    keep, avt.ops.SIM_XL = avt.ops.SIM_XL, "always"  # (the shapes the 256 x 256 tile can take run on it, whatever their tile count)
    try:
        out = avt.ops.sim_gemm_nt(tq(qh), tq(th), 0.1, mode, q_lo=tq(ql), t_lo=tq(tl)).cpu().numpy()
    finally:
        avt.ops.SIM_XL = keep
    emu = cref.sim_bf16(qh, ql, th, tl, 0.1, mode == "bf16x3")
    err_emu = np.abs(out - emu).max()
    err_can = np.abs(out - cref.sim_f32(q, t, 0.1)).max()
    print("mode", mode, "shape", (nq, nt, d), "vs emulation", err_emu, "vs canonical", err_can)
    assert err_emu < 5e-6 * 10 + 1e-6  # fp32-accumulate noise only
    if d >= 2304:  # the contract is stated at the embedding widths of the path (D=2304 / 14592)
        assert err_can < tol


def test_sim_bf16x3_on_the_encoder_tile_equals_the_plain_tile(avt, dev):
    """Round 5: the bf16x3 similarity runs on the encoder's 256 x 256 LDS-DMA tile when the shape allows (ops.SIM_XL).  Same three
    products per term, fp32 accumulation in another order: within 6e-6 of the plain tile's scores (cos / 0.1), written through a
    row pitch wider than the row, rows past a ragged M untouched."""
    g = torch.Generator().manual_seed(2)
    nq, nt, d = 700, 1024, 2304
    q = torch.nn.functional.normalize(torch.randn((nq, d), generator=g), dim=1).to(dev)
    t = torch.nn.functional.normalize(torch.randn((nt, d), generator=g), dim=1).to(dev)
    _, qh, ql = avt.ops.l2norm_rows(q, want_split=True)
    _, th, tl = avt.ops.l2norm_rows(t, want_split=True)
    buf = torch.full((nq + 3, nt + 64), 7.0, device=dev)
    keep = avt.ops.SIM_XL
    try:
        avt.ops.SIM_XL = "always"  # (the default takes the tile from 192 tiles up; 3 x 4 here)
        a = avt.ops.sim_gemm_nt(qh, th, 0.1, "bf16x3", q_lo=ql, t_lo=tl, out=buf[:nq, :nt])
        avt.ops.SIM_XL = False
        b = avt.ops.sim_gemm_nt(qh, th, 0.1, "bf16x3", q_lo=ql, t_lo=tl)
    finally:
        avt.ops.SIM_XL = keep
    assert float((a - b).abs().max()) < 6e-6  # (measured 3-4e-6 on scores of +-10)
    assert float((buf[nq:] - 7.0).abs().max()) == 0.0 and float((buf[:, nt:] - 7.0).abs().max()) == 0.0


def test_sim_bf16x3_row_blocks_are_bitwise_the_unsharded_build(avt, dev):
    """the bf16x3 tile is chosen from the WHOLE build's shape (rows of T), not from the rows of the call — so the row blocks
    a sharded build computes (512 of 4096 rows per rank at world 8) are bit for bit the rows of the one-rank matrix, with the default
    tile choice (4096^2 takes the 256 x 256 tile; a [512, 4096] block alone would have 32 tiles and used to take the plain one)."""
    g = torch.Generator().manual_seed(5)
    n, d = 4096, 2304
    q = torch.nn.functional.normalize(torch.randn((n, d), generator=g), dim=1).to(dev)
    t = torch.nn.functional.normalize(torch.randn((n, d), generator=g), dim=1).to(dev)
    _, qh, ql = avt.ops.l2norm_rows(q, want_split=True)
    _, th, tl = avt.ops.l2norm_rows(t, want_split=True)
    whole = avt.ops.sim_gemm_nt(qh, th, 0.1, "bf16x3", q_lo=ql, t_lo=tl)
    for lo in (0, 512, 3584):
        part = avt.ops.sim_gemm_nt(qh[lo : lo + 512].contiguous(), th, 0.1, "bf16x3", q_lo=ql[lo : lo + 512].contiguous(), t_lo=tl)
        assert torch.equal(part, whole[lo : lo + 512]), lo


# ---------------------------------------------------------------- transition select
def _check_transition(g, o, cap):
    cnt = o["cnt"]
    assert np.array_equal(g["cnt"].cpu().numpy(), cnt)
    gi, gs, gp = g["idx"].cpu().numpy(), g["seg"].cpu().numpy(), g["p"].cpu().numpy()
    for r in range(len(cnt)):
        k = min(cnt[r], cap)
        assert np.array_equal(gi[r, :k], o["idx"][r, :k])
        assert np.array_equal(gs[r, :k], o["seg"][r, :k])
        assert np.array_equal(gp[r, :k].view(np.uint32), o["p"][r, :k].view(np.uint32))
    gst, ost = g["stats"].cpu().numpy(), o["stats"]
    assert np.array_equal(gst[:, :2].view(np.uint32), ost[:, :2].view(np.uint32))  # row_sum, row_max exact
    np.testing.assert_allclose(gst[:, 2:], ost[:, 2:], rtol=5e-6, atol=2e-6)  # ce, entropy: reporting values (fast exp/log)


@pytest.mark.parametrize("th", [0.0, 0.3, 0.9])
@pytest.mark.parametrize("n", [45, 300, 1000])
def test_transition_target_order(avt, dev, th, n):
    rng = np.random.default_rng(n)
    sim = (rng.random((n, n), dtype=np.float32) * 8 + 1).astype(np.float32)
    q_ids = np.arange(n, dtype=np.int64)
    o = cref.row_transition(sim, q_ids=q_ids, threshold=th, cap=n)
    g = avt.ops.row_transition(torch.from_numpy(sim).to(dev), q_ids=torch.from_numpy(q_ids).to(dev), threshold=th,
                               cap=n)
    _check_transition(g, o, n)


def test_transition_audio_blend_and_negative_scores(avt, dev):
    n = 257
    sim = _rand((n, n), 11) * 3  # mixed signs: the row sum may be negative (validate.py:524 divides anyway)
    sim_a = np.abs(_rand((n, n), 12)) + 0.1
    q_ids = np.arange(n, dtype=np.int64)
    for th in (0.0, 0.3):
        o = cref.row_transition(sim, q_ids=q_ids, sim_a=sim_a, alpha=0.5, threshold=th, cap=n)
        g = avt.ops.row_transition(torch.from_numpy(sim).to(dev), q_ids=torch.from_numpy(q_ids).to(dev),
                                   sim_a=torch.from_numpy(sim_a).to(dev), alpha=0.5, threshold=th, cap=n)
        _check_transition(g, o, n)


def test_transition_identity_rows_ties_and_cap(avt, dev):
    sim = np.ones((6, 40), np.float32)
    sim[1, 5] = sim[1, 17] = 3.0  # exact tie for the max -> two survivors at th=0
    sim[2, :] = np.arange(40)
    o = cref.row_transition(sim, threshold=0.0, cap=4)
    g = avt.ops.row_transition(torch.from_numpy(sim).to(dev), threshold=0.0, cap=4)
    _check_transition(g, o, 4)
    assert o["cnt"][0] == 40 and o["cnt"][1] == 2  # cnt is the true count even past cap


def test_transition_long_row_uncached(avt, dev):
    rng = np.random.default_rng(5)
    sim = (rng.random((3, 20000), dtype=np.float32) + 0.5).astype(np.float32)
    o = cref.row_transition(sim, threshold=0.05, cap=256)
    g = avt.ops.row_transition(torch.from_numpy(sim).to(dev), threshold=0.05, cap=256)
    _check_transition(g, o, 256)


@pytest.mark.parametrize("nt", [4097, 8192, 12000, 16384])
@pytest.mark.parametrize("th", [0.0, 0.05])
def test_transition_config4_width_rows(avt, dev, nt, th):
    """Config 4's rows (N = 16384 windows; 2048 per rank x 16384): the 1024-thread register-resident select, bit-identical to
    the oracle (survivors, their order, probabilities, row sum / max) at every width that takes it, with the reference's
    [pos] + others target order (q_ids) and ragged last rounds."""
    rng = np.random.default_rng(nt)
    nq = 6
    sim = (rng.random((nq, nt), dtype=np.float32) * 4 + 0.5).astype(np.float32)
    sim[2, 100] = sim[2, nt - 1] = 9.0  # an exact tie for the maximum, first and last rounds
    q_ids = np.array([0, 1, nt // 2, nt - 3, nt - 2, nt - 1], dtype=np.int64)
    o = cref.row_transition(sim, q_ids=q_ids, n_seg=nt, threshold=th, cap=512)
    g = avt.ops.row_transition(torch.from_numpy(sim).to(dev), q_ids=torch.from_numpy(q_ids).to(dev), threshold=th, cap=512)
    _check_transition(g, o, 512)


@pytest.mark.parametrize("nt", [3000, 4096, 5000, 16384])
def test_topk_wide_rows(avt, dev, nt):
    sim = _rand((12, nt), nt)
    sim[3, 10] = sim[3, nt - 7] = 99.0  # tie -> lower column first
    sim[4, :] = -np.inf                # nothing but -inf: still k distinct picks, lowest columns first
    self_col = np.arange(12, dtype=np.int64) * (nt // 12)
    oi, ov = cref.row_topk(sim, 8, self_col)
    gi, gv = avt.ops.row_topk(torch.from_numpy(sim).to(dev), 8, torch.from_numpy(self_col).to(dev))
    assert np.array_equal(gi.cpu().numpy(), oi)
    assert np.array_equal(gv.cpu().numpy(), ov)


def test_topk(avt, dev):
    sim = _rand((50, 3000), 21)
    sim[3, 10] = sim[3, 2000] = 99.0  # tie -> lower column first
    self_col = np.arange(50, dtype=np.int64)
    oi, ov = cref.row_topk(sim, 8, self_col)
    gi, gv = avt.ops.row_topk(torch.from_numpy(sim).to(dev), 8, torch.from_numpy(self_col).to(dev))
    assert np.array_equal(gi.cpu().numpy(), oi)
    assert np.array_equal(gv.cpu().numpy(), ov)


# ---------------------------------------------------------------- InfoNCE CE
def test_softmax_ce(avt, dev):
    logits = _rand((8, 15), 31) * 10  # train.py: B=8, 1+negs=15, temp 0.1
    loss, prob = cref.softmax_ce_fwd(logits)
    gl, gp = avt.ops.softmax_ce_fwd(torch.from_numpy(logits).to(dev))
    np.testing.assert_allclose(gl.cpu().numpy(), loss, rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(gp.cpu().numpy(), prob, rtol=1e-6, atol=1e-7)
    ref = torch.nn.functional.cross_entropy(torch.from_numpy(logits), torch.zeros(8, dtype=torch.long),
                                            reduction="none")
    np.testing.assert_allclose(gl.cpu().numpy(), ref.numpy(), rtol=1e-5, atol=1e-5)
    d = cref.softmax_ce_bwd(gp.cpu().numpy(), scale=1.0 / 8)
    gd = avt.ops.softmax_ce_bwd(gp, scale=1.0 / 8)
    assert np.array_equal(gd.cpu().numpy(), d)


# ---------------------------------------------------------------- clip_pack
@pytest.mark.parametrize("W,S,H,Wd,hw", [(20, 4, 128, 128, 224), (15, 6, 90, 120, 224), (20, 4, 64, 48, 100)])
def test_clip_pack(avt, dev, W, S, H, Wd, hw):
    g = torch.Generator().manual_seed(123)
    n_win = 6
    F_ = (n_win - 1) * S + W + 3
    frames = torch.randint(0, 256, (F_, H, Wd, 3), generator=g, dtype=torch.uint8)
    starts = np.arange(n_win) * S
    slow, fast = avt.ops.clip_pack(frames.to(dev), starts, W, out_hw=hw, dtype=torch.float32)
    sb, fb = avt.ops.clip_pack(frames.to(dev), starts, W, out_hw=hw, dtype=torch.bfloat16)
    for i, st in enumerate(starts):
        rs, rf = ref_py.pack_clip(frames, int(st), W, out_hw=hw)
        # fp32: same op order as torch up to FMA contraction -> 1e-5 abs on values in [-2.1, 2.5]
        assert (slow[i].cpu() - rs).abs().max() < 1e-5
        assert (fast[i].cpu() - rf).abs().max() < 1e-5
        # bf16 output: one bf16 ulp (2^-7 relative) where the fp32 value sits on a rounding boundary
        assert (sb[i].float().cpu() - rs).abs().max() < 2.5 * 2 ** -7
        assert (fb[i].float().cpu() - rf.bfloat16().float()).abs().max() < 2.5 * 2 ** -6
        assert ((fb[i].float().cpu() - rf.bfloat16().float()).abs() > 0).float().mean() < 1e-3


def test_errors_are_loud(avt, dev):
    with pytest.raises(avt._lib.AvtError):
        avt.ops.l2norm_rows(torch.zeros(4, 8))  # CPU tensor: no fallback
    with pytest.raises(avt._lib.AvtError):
        avt.ops.sim_gemm_nt(torch.zeros(4, 8, device=dev), torch.zeros(4, 8, device=dev), 0.0, "f32")  # temp == 0


def test_empty_inputs_are_no_ops(avt, dev):
    """Edge cases: zero rows / zero windows return empty results instead of faulting."""
    z = torch.zeros((0, 64), device=dev)
    y, _, _ = avt.ops.l2norm_rows(z)
    assert y.shape == (0, 64)
    t = torch.randn((5, 64), device=dev)
    assert avt.ops.sim_gemm_nt(z, t, 0.1, "f32").shape == (0, 5)
    assert avt.ops.sim_gemm_nt(t, z, 0.1, "f32").shape == (5, 0)
    sel = avt.ops.row_transition(torch.zeros((0, 7), device=dev), threshold=0.3, cap=4)
    assert sel["cnt"].shape == (0,)
    frames = torch.zeros((30, 16, 16, 3), dtype=torch.uint8, device=dev)
    s, f = avt.ops.clip_pack(frames, np.zeros(0, np.int32), 20, out_hw=32)
    assert s.shape == (0, 3, 8, 32, 32) and f.shape == (0, 3, 32, 32, 32)


def test_single_window_and_two_segment_rows(avt, dev):
    """Smallest legal shapes: n_seg = 2 (row length 1 or 2) and a one-row similarity."""
    sim = np.array([[0.5, 2.0], [3.0, 1.0]], np.float32)
    q_ids = np.array([0, 1], np.int64)
    o = cref.row_transition(sim, q_ids=q_ids, threshold=0.0, cap=2)
    g = avt.ops.row_transition(torch.from_numpy(sim).to(dev), q_ids=torch.from_numpy(q_ids).to(dev), threshold=0.0, cap=2)
    _check_transition(g, o, 2)
    assert o["cnt"].tolist() == [1, 1] and o["seg"][0, 0] == 1  # q=0: only target is segment 1; q=1 (last): [self, 0]


@pytest.mark.parametrize("b,n,d", [(8, 15, 2304), (3, 5, 48), (2, 21, 14592), (1, 1, 7)])
def test_infonce_fused_fwd_bwd_matches_autograd(avt, dev, b, n, d):
    """Fused training branch (normalise -> bmm -> /temp) vs torch autograd in fp64 on the same inputs."""
    torch.manual_seed(b * 100 + n)
    q = torch.randn(b, d, device=dev, requires_grad=True)
    t = torch.randn(b, n, d, device=dev, requires_grad=True)
    g = torch.randn(b, n, device=dev)
    out, _, _ = avt.models._InfoNCELogits.apply_ops(q, t, 0.1)
    out.backward(g)
    q64, t64 = q.detach().double().requires_grad_(), t.detach().double().requires_grad_()
    ref = torch.bmm(torch.nn.functional.normalize(q64, dim=1).unsqueeze(1),
                    torch.nn.functional.normalize(t64, dim=2).permute(0, 2, 1)).squeeze(1) / 0.1
    ref.backward(g.double())
    assert (out.detach().double() - ref.detach()).abs().max() < 2e-5
    assert (q.grad.double() - q64.grad).abs().max() < 1e-5 * max(1.0, q64.grad.abs().max().item())
    assert (t.grad.double() - t64.grad).abs().max() < 1e-5 * max(1.0, t64.grad.abs().max().item())


def test_classic_pairwise_l2_matches_reference_fixture(avt, dev):
    """BASELINE config 1 on the GPU: D1 / P1 of the classic baseline vs the reference's own output (G8) and, at the
    config's full size (200 frames of 128x128x3), vs the oracle."""
    from oracle import ref_py

    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g8_classic.npz"))
    d1, p1, s1 = avt.classic.compute_D1_device(g["frames"], 0.1, dev)
    np.testing.assert_allclose(d1.cpu().numpy(), g["d1"], rtol=2e-6, atol=1e-3)
    np.testing.assert_allclose(p1.cpu().numpy(), g["p1"], rtol=1e-4, atol=1e-6)
    frames = torch.randint(0, 256, (200, 128, 128, 3), generator=torch.Generator().manual_seed(7), dtype=torch.uint8)
    d_gpu = avt.ops.pairwise_l2(frames.float().reshape(200, -1).to(dev)).cpu().numpy()
    d_ref, _, _ = ref_py.classic_d1_p1(frames.numpy(), 0.1)
    # uint8 inputs: every squared difference and the whole sum are exact in fp64 -> identical up to the final rounding
    assert np.array_equal(d_gpu, d_ref)
    assert (np.diag(d_gpu) == 0).all() and np.array_equal(d_gpu, d_gpu.T)


def test_classic_d2_and_q_learning_on_device(avt, dev):
    """Config 1's remaining matrix steps on the GPU (csrc/classic.hip) vs the reference's outputs (fixture G8: computeD2 and
    q_learning run from the imported reference) and vs the CPU restatement at the BASELINE size (200 frames, filter 16)."""
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g8_classic.npz"))
    d1 = torch.from_numpy(g["d1"]).to(dev)
    d2, p2, s2, _ = avt.classic.compute_D2_device(d1, 0.1, filter_size=4)
    np.testing.assert_allclose(d2.cpu().numpy(), g["d2"], rtol=1e-6, atol=1e-4)
    np.testing.assert_allclose(p2.cpu().numpy(), g["p2"], rtol=1e-4, atol=1e-7)
    d3, p3, p3n, s3 = avt.classic.q_learning_device(torch.from_numpy(g["d2"]).to(dev), 0.1)
    np.testing.assert_allclose(d3.cpu().numpy(), g["d3"], rtol=2e-6, atol=1e-4)  # (device pow differs from the CPU's in the last ulp)
    np.testing.assert_allclose(p3.cpu().numpy(), g["p3"], rtol=1e-3, atol=1e-7)
    assert np.array_equal(p3n.cpu().numpy() > 0, g["p3_thresholded"] > 0)
    # BASELINE config 1 size: 200 frames 128x128x3 -> D1 200^2 -> D2 185^2 -> D3 185^2 (LDS-resident: 137 KB)
    gen = torch.Generator().manual_seed(7)
    frames = torch.randint(0, 256, (200, 32, 32, 3), generator=gen).float()
    d1c, _, _ = avt.classic.compute_D1(frames, 0.1)
    d2c, _, _, _ = avt.classic.compute_D2(d1c, 0.1, filter_size=16)
    d3c, p3c, _, _ = avt.classic.q_learning(d2c, 0.1)
    d2d, _, _, _ = avt.classic.compute_D2_device(d1c.to(dev), 0.1, filter_size=16)
    np.testing.assert_allclose(d2d.cpu().numpy(), d2c.numpy(), rtol=1e-5, atol=1e-3)
    d3d, p3d, _, _ = avt.classic.q_learning_device(d2c.to(dev), 0.1)
    np.testing.assert_allclose(d3d.cpu().numpy(), d3c.numpy(), rtol=1e-5, atol=1e-3)
    np.testing.assert_allclose(p3d.cpu().numpy(), p3c.numpy(), rtol=2e-3, atol=1e-7)
