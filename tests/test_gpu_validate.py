"""End-to-end parity on the MI355X: this build's validate()/TextureEngine against what the REFERENCE's
validate() produced on the same video, audio, weights and seeds (tests/golden/g5_*.npz), and the aligned
N x N pipeline against the CPU oracle."""
import math
import os
import sys
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from oracle import cref, ref_py

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from tiny_encoders import TinyR3D, TinySlowFast, checksum, seeded  # noqa: E402

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _gold(case):
    return np.load(os.path.join(GOLD, "g5_validate_%s.npz" % case), allow_pickle=True)


def _model(avt, g, dev):
    s = [int(x) for x in g["seeds"]]
    n_frames, W, S, mbs, G, hw = [int(x) for x in g["cfg"][:6]]
    th, alpha, temp = [float(x) for x in g["th_alpha_temp"]]
    vgg = seeded(avt.VGGish, s[2])
    assert abs(checksum(vgg) - g["enc_ck"][2]) < 1e-6
    arch = str(g["arch"])
    cls = TinySlowFast if arch == "slowfast" else TinyR3D
    m = avt.ContrastivePredictionTemporal(seeded(cls, s[0]), seeded(cls, s[1]), vgg, 2, 128, temp, W,
                                          S, th, mini_batchsize=mbs, enc_arch=arch, img_size=hw)
    return m.to(dev).eval()


def _args(g):
    n_frames, W, S, mbs, G, hw, L, nvl, fps, with_da = [int(x) for x in g["cfg"]]
    th, alpha, temp = [float(x) for x in g["th_alpha_temp"]]
    return SimpleNamespace(vdata=None, adata=None, dadata=None, subsample_rate=1, fps=fps, stride=S, window=W,
                           enc_arch=str(g["arch"]), img_size=hw, model_type=2, mini_batchsize=mbs, threshold=th,
                           alpha=alpha, temp=temp, driving_audio=None, da_feats="VGG", interpolation=False,
                           new_video_length=nvl, results_folder=None, logname="exp", batch_size=24,
                           stitch_mode="compat", ref_num_gpus=G, enc_batch=16)


@pytest.mark.parametrize("case", ["sf_th03", "sf_th00", "sf_g2", "sf_da", "r3d_th03"])
def test_validate_reproduces_reference_frames_list(avt, dev, case, capsys):
    g = _gold(case)
    model, args = _model(avt, g, dev), _args(g)
    with_da = int(g["cfg"][9])
    np.random.seed(1234)
    torch.manual_seed(4321)
    frames = avt.validate(model, args, video_name="gold", model_type=2, video=(g["video"], float(g["cfg"][8])),
                          audio=(g["wave"], 16000), driving_audio=(g["wave_da"], 16000) if with_da else None)
    out = capsys.readouterr().out
    assert frames == [int(x) for x in g["frames_list"]]  # bit-exact stitch indices
    chosen = [int(x.split(":")[1]) for x in out.splitlines() if x.startswith("Chosen next frame:")]
    assert chosen == [int(x) for x in g["chosen"]]
    assert "Frames list: " in out


@pytest.mark.parametrize("case", ["sf_th03", "sf_g2", "sf_da"])
def test_engine_rows_match_reference_logits(avt, dev, case):
    """Encode-once rows vs the logits the reference got by re-encoding every window at every step: <= 1e-3
    (BASELINE contract; observed ~1e-5, the GPU conv/VGGish summation order)."""
    from avtex.audio_frontend import waveform_to_examples
    from avtex.texture import TextureEngine

    g = _gold(case)
    n_frames, W, S, mbs, G, hw, L, nvl, fps, with_da = [int(x) for x in g["cfg"]]
    th, alpha, temp = [float(x) for x in g["th_alpha_temp"]]
    m = _model(avt, g, dev)
    eng = TextureEngine(m.q_encoder, m.t_encoder, m.t_a_encoder, window=W, stride=S, temp=temp, img_size=hw,
                        model_type=2, device=dev, enc_batch=16)
    assert eng.set_video(torch.from_numpy(g["video"])) == L
    wave = g["wave"][: n_frames * math.floor(16000 / fps)]
    aeg = torch.from_numpy(waveform_to_examples(wave, 16000)).unsqueeze(1).float()[:L]
    deg = torch.from_numpy(waveform_to_examples(g["wave_da"], 16000)).unsqueeze(1).float() if with_da else None
    eng.set_audio(aeg, deg)
    worst = 0.0
    for step, q in enumerate(g["queries"]):
        out, out_a, seg = eng.compat_row(int(q), step + 1, mbs, G)
        ref = np.asarray(g["raw_logits"][step], np.float32)[: len(seg)]
        worst = max(worst, float(np.abs(out.cpu().numpy()[0] - ref).max()))
        if with_da:
            ra = np.asarray(g["raw_logits_a"][step], np.float32)[: len(seg)]
            worst = max(worst, float(np.abs(out_a.cpu().numpy()[0] - ra).max()))
        choices, _ = eng.select(out, out_a, th, alpha)
        assert np.array_equal(choices, np.asarray(g["choices"][step], np.int64))
    print("worst |logit - reference| =", worst)
    assert worst < 1e-3
    assert eng.encoded < 3 * n_frames  # encode-once (the reference pushes ~steps * L windows through)


def test_aligned_pipeline_matches_oracle(avt, dev):
    """Aligned mode (what the N x N metric measures): tables -> l2norm -> MFMA sim -> select, against the
    oracle run on the SAME embedding tables: matrix bit-identical, survivors identical."""
    from avtex.texture import TextureEngine

    g = _gold("sf_th03")
    n_frames, W, S, mbs, G, hw, L = [int(x) for x in g["cfg"][:7]]
    m = _model(avt, g, dev)
    m.model_type = 1
    eng = TextureEngine(m.q_encoder, m.t_encoder, None, window=W, stride=S, temp=0.1, img_size=hw, model_type=1,
                        device=dev, enc_batch=16)
    eng.set_video(torch.from_numpy(g["video"]))
    qv, tv = eng.build_tables()
    eng.normalise(split=True)
    sim = eng.similarity("f32").cpu().numpy()
    qn, _, _ = cref.l2norm_rows(qv.cpu().numpy(), want_split=False)
    tn, _, _ = cref.l2norm_rows(tv.cpu().numpy(), want_split=False)
    ref = cref.sim_f32(qn, tn, 0.1)
    assert np.array_equal(sim.view(np.uint32), ref.view(np.uint32))
    for th in (0.0, 0.3):
        sel = eng.transitions(th, cap=L)
        o = cref.row_transition(ref, q_ids=np.arange(L), threshold=th, cap=L)
        assert np.array_equal(sel["cnt"].cpu().numpy(), o["cnt"])
        assert np.array_equal(sel["seg"].cpu().numpy(), o["seg"])
    # and the embeddings themselves against CPU torch on the oracle's clip packing (fp32 encoders)
    starts = np.arange(L) * S
    packs = [ref_py.pack_clip(g["video"], int(s), W, out_hw=hw) for s in starts[:8]]
    with torch.no_grad():
        cpu_t = seeded(TinySlowFast, int(g["seeds"][1]))([torch.stack([p[0] for p in packs]),
                                                          torch.stack([p[1] for p in packs])]).numpy()
    np.testing.assert_allclose(tv[:8].cpu().numpy(), cpu_t, rtol=0, atol=2e-5)
    # bf16x3 MFMA mode stays inside the 1e-3 contract on the same tables
    s3 = eng.similarity("bf16x3").cpu().numpy()
    assert np.abs(s3 - ref).max() < 1e-3


def test_operator_forward_matches_reference(avt, dev):
    """ContrastivePredictionTemporal.forward (inference + driving-audio VGG branch) vs G3 vectors."""
    g = np.load(os.path.join(GOLD, "g3_g4_operator.npz"))
    W, S, mbs, hw = [int(x) for x in g["m1_cfg"]]
    vgg = seeded(avt.VGGish, 21)
    cpt = avt.ContrastivePredictionTemporal(seeded(TinySlowFast, 15), seeded(TinySlowFast, 16), vgg, 2, 128, temp=0.1,
                                            window=W, stride=S, mini_batchsize=mbs, enc_arch="slowfast",
                                            img_size=hw).to(dev).eval()
    qwin = torch.from_numpy(g["m1_qwin"]).to(dev)
    qf = avt.models.process_cv2_inputs(qwin)
    t = lambda k: torch.from_numpy(g[k]).to(dev)
    with torch.no_grad():
        o, oa, q, tt = cpt(qf, t("m1_chunk"), q_audio_eg=t("m2_qa"), t_audio_eg=t("m2_ta"), is_inference=True,
                           driving_audio=t("m2_da"), da_model=cpt.t_a_encoder, da_feats="VGG", cam_viz=True)
    assert oa.shape == g["m2_out_a"].shape and q.shape == g["m2_q"].shape and tt.shape == g["m2_t"].shape
    np.testing.assert_allclose(o.cpu().numpy(), g["m2_out"], rtol=0, atol=1e-3)
    np.testing.assert_allclose(oa.cpu().numpy(), g["m2_out_a"], rtol=0, atol=1e-3)
    np.testing.assert_allclose(q.cpu().numpy(), g["m2_q"], rtol=0, atol=1e-5)
    # m=1
    cpt1 = avt.ContrastivePredictionTemporal(seeded(TinySlowFast, 11), seeded(TinySlowFast, 12), None, 1, 128,
                                             temp=0.1, window=W, stride=S, mini_batchsize=mbs, enc_arch="slowfast",
                                             img_size=hw).to(dev).eval()
    with torch.no_grad():
        o1, q1, t1 = cpt1(qf, t("m1_chunk"), is_inference=True, cam_viz=True)
    np.testing.assert_allclose(o1.cpu().numpy(), g["m1_out"], rtol=0, atol=1e-4)
    np.testing.assert_allclose(t1.cpu().numpy(), g["m1_t"], rtol=0, atol=1e-5)


def test_training_step_matches_reference(avt, dev):
    """Training branch + HIP InfoNCE criterion: logits, loss and encoder gradients vs G4."""
    g = np.load(os.path.join(GOLD, "g3_g4_operator.npz"))
    W, S, mbs, hw = [int(x) for x in g["m1_cfg"]]
    cpt = avt.ContrastivePredictionTemporal(seeded(TinySlowFast, 17), seeded(TinySlowFast, 18), None, 1, 128, temp=0.1,
                                            window=W, stride=S, enc_arch="slowfast", img_size=hw).to(dev).train()
    t = lambda k: torch.from_numpy(g[k]).to(dev)
    logits = cpt([t("tr_qs"), t("tr_qf")], [t("tr_ts"), t("tr_tf")])
    loss = avt.InfoNCECriterion()(logits, torch.zeros(logits.shape[0], dtype=torch.long, device=dev))
    loss.backward()
    np.testing.assert_allclose(logits.detach().cpu().numpy(), g["tr_logits"], rtol=0, atol=1e-4)
    assert abs(float(loss) - float(g["tr_loss"])) < 1e-5
    np.testing.assert_allclose(cpt.q_encoder.fc.weight.grad.cpu().numpy(), g["tr_grad_fc_q"], rtol=1e-3, atol=1e-6)
    np.testing.assert_allclose(cpt.t_encoder.fc.weight.grad.cpu().numpy(), g["tr_grad_fc_t"], rtol=1e-3, atol=1e-6)


def test_validate_with_real_slowfast_on_mfma_encoder(avt, dev, capsys):
    """validate() end to end with the real SlowFast-8x8-R50 (random init) on the MFMA convolution path, aligned
    mode, against the oracle run on the SAME embedding tables: identical frames list."""
    from avtex.slowfast import SlowFast
    from avtex.texture import TextureEngine
    from avtex.fused_slowfast import SlowFastMFMA

    torch.manual_seed(0)
    W, S, L = 20, 4, 14
    g = torch.Generator().manual_seed(3)
    video = torch.randint(0, 256, (L * S + W + 1, 64, 64, 3), generator=g, dtype=torch.uint8)
    q_mod, t_mod = SlowFast().eval(), SlowFast().eval()
    with torch.no_grad():  # non-degenerate residual branches (c_bn is zero-initialised)
        for m in list(q_mod.modules()) + list(t_mod.modules()):
            if isinstance(m, torch.nn.BatchNorm3d):
                m.weight.uniform_(0.5, 1.0)
    model = avt.ContrastivePredictionTemporal(q_mod, t_mod, None, 1, 128, 0.1, W, S, 0.3, mini_batchsize=8,
                                              enc_arch="slowfast", img_size=224).to(dev).eval()
    args = SimpleNamespace(vdata=None, adata=None, dadata=None, subsample_rate=1, fps=4, stride=S, window=W,
                           enc_arch="slowfast", img_size=224, model_type=1, mini_batchsize=8, threshold=0.3, alpha=0.5,
                           temp=0.1, driving_audio=None, da_feats="VGG", interpolation=False, new_video_length=12,
                           results_folder=None, logname="exp", batch_size=24, stitch_mode="aligned", enc_batch=8,
                           enc_impl="mfma", enc_dtype="bf16")
    np.random.seed(7)
    frames = avt.validate(model, args, video_name="x", model_type=1, video=(video, 4.0))
    assert "Frames list: " in capsys.readouterr().out and len(frames) >= 48
    # oracle walk on the same tables
    eng = TextureEngine(SlowFastMFMA(q_mod, dev), SlowFastMFMA(t_mod, dev), None, window=W, stride=S, temp=0.1,
                        img_size=224, model_type=1, device=dev, enc_batch=8)
    assert eng.set_video(video) == L
    qv, tv = eng.build_tables()
    qn, _, _ = cref.l2norm_rows(qv.cpu().numpy(), want_split=False)
    tn, _, _ = cref.l2norm_rows(tv.cpu().numpy(), want_split=False)
    sim = cref.sim_f32(qn, tn, 0.1)

    def row_fn(q):
        o = cref.row_transition(sim[q : q + 1], q_ids=np.array([q]), n_seg=L, threshold=0.3, cap=L)
        return o["idx"][0, : o["cnt"][0]], ref_py.target_segment_ids(q, L)

    ref_frames, _, _ = ref_py.stitch_walk(row_fn, len(video), W, S, 48, q_id=10, rng=np.random.RandomState(7))
    assert frames == ref_frames


def test_sharded_build_single_rank_equals_engine(avt, dev):
    """dist.sharded_transition_build (config 4 pipeline) at world 1 == the unsharded engine path, and a 2-way row
    split of the same build gives the same survivors (no dependence on the shard boundary)."""
    from avtex import dist as adist
    from avtex.texture import TextureEngine

    g = _gold("sf_th03")
    n_frames, W, S, mbs, G, hw, L = [int(x) for x in g["cfg"][:7]]
    m = _model(avt, g, dev)
    eng = TextureEngine(m.q_encoder, m.t_encoder, None, window=W, stride=S, temp=0.1, img_size=hw, model_type=1,
                        device=dev, enc_batch=16)
    eng.set_video(torch.from_numpy(g["video"]))
    sel, (lo, hi), sim = adist.sharded_transition_build(eng, L, 0.3, cap=L, precision="f32", rank=0, world=1)
    assert (lo, hi) == (0, L)
    eng.build_tables()
    eng.normalise()
    ref = eng.similarity("f32")
    assert torch.equal(sim, ref)
    full = eng.transitions(0.3, cap=L)
    assert torch.equal(sel["seg"], full["seg"]) and torch.equal(sel["cnt"], full["cnt"])
    for r in range(2):  # emulate two ranks in one process: rows [lo,hi) against the full (gathered) target table
        lo, hi = adist.shard_range(L, r, 2)
        q_ids = torch.arange(lo, hi, device=dev, dtype=torch.int64)
        part = avt.ops.row_transition(avt.ops.sim_gemm_nt(eng.Qn[lo:hi].contiguous(), eng.Tn, 0.1, "f32"), q_ids=q_ids,
                                      threshold=0.3, cap=L)
        assert torch.equal(part["seg"], full["seg"][lo:hi]) and torch.equal(part["cnt"], full["cnt"][lo:hi])
