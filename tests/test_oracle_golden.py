"""Pins the CPU oracle (oracle/) to the reference: every check here compares the
oracle's restatement with vectors the reference's OWN code produced when
tools/gen_golden.py imported and ran it (tests/golden/*.npz).  CPU only."""
import math
import os
import sys

import numpy as np
import pytest
import torch

from oracle import cref, ref_py

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from tiny_encoders import TinyR3D, TinySlowFast, checksum, seeded  # noqa: E402

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def gold(name):
    return np.load(os.path.join(GOLD, name), allow_pickle=True)


def _f(x):
    return np.asarray(x, np.float32)


def _i(x):
    return np.asarray(x, np.int64)


# ------------------------------------------------------------------ G1 chunk helpers (utils.py:208-260)
def test_g1_split_helpers():
    g = gold("g1_split.npz")
    for i in range(5):
        n, mbs, W, S = g["ov%d_args" % i]
        out, nv = ref_py.split_into_overlapping_segments(np.arange(1, n + 1, dtype=np.float32), mbs, W, S)
        assert np.array_equal(out, g["ov%d_out" % i]) and nv == g["ov%d_nvalid" % i]
    for i in range(4):
        n, mbs = g["sb%d_args" % i]
        out, nv = ref_py.split_into_batches(np.arange(1, n + 1, dtype=np.float32)[None], mbs)
        assert np.array_equal(out, g["sb%d_out" % i]) and nv == g["sb%d_nvalid" % i]


# ------------------------------------------------------------------ G3 operator (models.py:307-467)
def test_g3_similarity_and_temperature():
    g = gold("g3_g4_operator.npz")
    for pre in ("m1", "m2"):
        q, t, out = g[pre + "_q"], g[pre + "_t"], g[pre + "_out"]  # q [1,1,D] t [1,D,mbs] out [1,mbs]
        sim = cref.sim_f32(q[0], np.ascontiguousarray(t[0].T), 0.1)
        # the reference's own fp32 bmm carries ~sqrt(D) ulp of summation noise (D = 48 / 12336)
        np.testing.assert_allclose(sim, out, rtol=0, atol=2e-6 if pre == "m1" else 3e-5)
    # unnormalised embeddings -> oracle l2norm == the reference's F.normalize output
    W, S, mbs, hw = g["m1_cfg"]
    q_enc, t_enc = seeded(TinySlowFast, 11), seeded(TinySlowFast, 12)
    assert abs(checksum(q_enc) - g["m1_ck"][0]) < 1e-9 and abs(checksum(t_enc) - g["m1_ck"][1]) < 1e-9
    chunk = torch.from_numpy(g["m1_chunk"])[0]
    packs = [ref_py.pack_clip_ids(chunk, np.arange(i * S, i * S + W), hw) for i in range(mbs)]
    with torch.no_grad():
        emb = t_enc([torch.stack([p[0] for p in packs]), torch.stack([p[1] for p in packs])]).numpy()
    tn, _, _ = cref.l2norm_rows(emb, want_split=False)
    np.testing.assert_allclose(tn.T, g["m1_t"][0], rtol=0, atol=2e-7)
    qp = ref_py.pack_clip_ids(torch.from_numpy(g["m1_qwin"]), np.arange(W), hw)
    with torch.no_grad():
        qe = q_enc([qp[0][None], qp[1][None]]).numpy()
    qn, _, _ = cref.l2norm_rows(qe, want_split=False)
    np.testing.assert_allclose(qn, g["m1_q"][0], rtol=0, atol=2e-7)
    np.testing.assert_allclose(cref.sim_f32(qn, tn, 0.1), g["m1_out"], rtol=0, atol=5e-6)


def test_g3_audio_concat_is_jointly_normalised():
    """m=2: q = normalize(cat(video, vggish(audio))) (models.py:347-351); driving branch :424-439."""
    g = gold("g3_g4_operator.npz")
    q = g["m2_q"][0, 0]
    assert q.shape[0] == 48 + 12288 and abs(np.linalg.norm(q) - 1) < 1e-5
    oa = g["m2_out_a"]  # [1,1,mbs] — note the extra dim the reference keeps (models.py:439)
    assert oa.shape == (1, 1, 5)


def test_g7_vggish_layout():
    g = gold("g3_g4_operator.npz")
    raw, feats = g["g7_raw_nchw"], g["g7_feats"]  # [n,512,6,4] -> NHWC flatten [n,12288]
    assert feats.shape == (5, 12288)
    assert np.array_equal(raw.transpose(0, 2, 3, 1).reshape(5, -1), feats)


# ------------------------------------------------------------------ G4 training branch (train.py:129-135)
def test_g4_infonce_loss_and_gradient():
    g = gold("g3_g4_operator.npz")
    logits = g["tr_logits"]
    loss, prob = cref.softmax_ce_fwd(logits)
    assert abs(loss.mean() - float(g["tr_loss"])) < 1e-6
    d = cref.softmax_ce_bwd(prob, scale=1.0 / logits.shape[0])
    np.testing.assert_allclose(d, g["tr_dlogits"], rtol=0, atol=1e-7)


# ------------------------------------------------------------------ G5 validate() end to end
CASES = ["sf_th03", "sf_th00", "sf_g2", "sf_da"]
ALL_CASES = CASES + ["r3d_th03"]


@pytest.mark.parametrize("case", CASES)
def test_g5_window_map_q3_q4(case):
    """[quirks Q3/Q4] which frames each output slot really scores, and the labels attached."""
    g = gold("g5_validate_%s.npz" % case)
    n_frames, W, S, mbs, G, hw, L = [int(x) for x in g["cfg"][:7]]
    assert L == ref_py.num_segments(n_frames, W, S)
    for step, q in enumerate(g["queries"]):
        wins, seg = ref_py.compat_window_frames(int(q), n_frames, W, S, mbs, G)
        rec = g["window_frames"][step]
        assert np.array_equal(rec[0], np.arange(q * S, q * S + W))  # the query window (validate.py:333)
        assert np.array_equal(rec[1 : 1 + len(wins)], wins)
        assert len(seg) == len(g["rows_pre"][step])


@pytest.mark.parametrize("case", ALL_CASES)
def test_g5_row_postprocess_and_walk(case):
    """validate.py:524-572 + :580-615 from the reference's raw logits: same rows, survivors, RNG draws, frames."""
    g = gold("g5_validate_%s.npz" % case)
    n_frames, W, S, mbs, G, hw, L, nvl, fps, with_da = [int(x) for x in g["cfg"]]
    th, alpha, temp = [float(x) for x in g["th_alpha_temp"]]
    rng = np.random.RandomState(1234)
    q_id, p_q, frames = 10, -1, []
    if with_da:
        q_id = int(g["queries"][0])  # audio-argmax start (validate.py:223-240), checked in test_audio_start
    for step in range(len(g["chosen"])):
        assert q_id == g["queries"][step]
        raw = np.asarray(g["raw_logits"][step], np.float32)
        raw_a = np.asarray(g["raw_logits_a"][step], np.float32) if with_da else None
        n_out = len(g["rows_pre"][step])
        r = ref_py.row_postprocess(raw[:n_out], th, raw_a[:n_out] if with_da else None, alpha)
        np.testing.assert_allclose(r["p_pre"], _f(g["rows_pre"][step]), rtol=0, atol=1e-7)
        np.testing.assert_allclose(r["p_post"], _f(g["rows_post"][step]), rtol=0, atol=1e-7)
        assert np.array_equal(r["choices"], _i(g["choices"][step]))
        # the canonical C oracle (fp64-accumulated sums) selects the same survivors as torch's fp32 sums
        c = cref.row_transition(raw[None, :n_out], sim_a=raw_a[None, :n_out] if with_da else None, alpha=alpha,
                                threshold=th, cap=n_out)
        k = c["cnt"][0]
        assert np.array_equal(c["idx"][0, :k], _i(g["choices"][step]))
        np.testing.assert_allclose(c["p"][0, :k], _f(g["rows_post"][step])[_i(g["choices"][step])], rtol=2e-6, atol=0)
        seg = ref_py.target_segment_ids(q_id, L)
        rdm = rng.choice(r["choices"])
        assert rdm == g["rdm"][step]
        nq = int(seg[rdm])
        ids, _ = ref_py.frame_bookkeeping(nq, p_q, W, S)
        frames.extend(int(i) for i in ids)
        p_q = q_id = nq
        assert q_id == g["chosen"][step]
    assert frames == list(g["frames_list"])
    assert len(frames) >= math.ceil(fps) * nvl


@pytest.mark.parametrize("case", ["sf_th03", "sf_g2", "sf_da"])
def test_g5_encode_once_reproduces_reference_rows(case):
    """The restructure the build is about: every distinct window encoded ONCE, rows by table lookup,
    reproduces the logits the reference got by re-encoding every window every step."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import avtex
    from avtex.vggish import VGGish
    from avtex.audio_frontend import waveform_to_examples

    g = gold("g5_validate_%s.npz" % case)
    n_frames, W, S, mbs, G, hw, L, nvl, fps, with_da = [int(x) for x in g["cfg"]]
    th, alpha, temp = [float(x) for x in g["th_alpha_temp"]]
    s = [int(x) for x in g["seeds"]]
    q_enc, t_enc, vgg = seeded(TinySlowFast, s[0]), seeded(TinySlowFast, s[1]), seeded(VGGish, s[2])
    assert abs(checksum(vgg) - g["enc_ck"][2]) < 1e-6
    wave = g["wave"][: n_frames * math.floor(16000 / fps)]  # validate.py:155-158
    audio_eg = waveform_to_examples(wave, 16000)[:, None].astype(np.float32)[:L]
    drv = waveform_to_examples(g["wave_da"], 16000)[:, None].astype(np.float32) if with_da else None
    orc = ref_py.CompatOracle(g["video"], W, S, mbs, G, hw, q_enc, t_enc, temp, audio_eg, vgg, drv)
    for step in range(0, len(g["chosen"]), 3):
        q = int(g["queries"][step])
        out, out_a, seg = orc.row(q, step + 1)
        n_out = len(seg)
        np.testing.assert_allclose(out, np.asarray(g["raw_logits"][step], np.float32)[:n_out], rtol=0, atol=3e-5)
        if with_da:
            np.testing.assert_allclose(out_a, np.asarray(g["raw_logits_a"][step], np.float32)[:n_out], rtol=0,
                                       atol=3e-5)
        r = ref_py.row_postprocess(out, th, out_a, alpha)
        assert np.array_equal(r["choices"], _i(g["choices"][step]))
    assert len(orc.cache_t) < 2 * n_frames  # encode-once: far fewer encodes than steps * N


def test_g5_audio_start_segment():
    """validate.py:223-240: start at the segment whose log-mel is most cosine-similar to driving example 0."""
    import avtex
    from avtex.audio_frontend import waveform_to_examples
    from avtex.validate import audio_start_segment

    g = gold("g5_validate_sf_da.npz")
    n_frames, W, S = [int(x) for x in g["cfg"][:3]]
    L, fps = int(g["cfg"][6]), int(g["cfg"][8])
    wave = g["wave"][: n_frames * math.floor(16000 / fps)]
    a = waveform_to_examples(wave, 16000)[:L].astype(np.float32)
    d = waveform_to_examples(g["wave_da"], 16000).astype(np.float32)
    assert audio_start_segment(a, d[0]) == int(g["queries"][0])


# ------------------------------------------------------------------ G6 audio front-end
def test_g6_waveform_to_examples():
    import avtex
    from avtex.audio_frontend import waveform_to_examples

    g = gold("g6_logmel.npz")
    ex = waveform_to_examples(g["wave"], 16000)
    assert ex.dtype == np.float64 and ex.shape == g["examples"].shape
    np.testing.assert_allclose(ex, g["examples"], rtol=1e-12, atol=1e-12)


def test_g6_oracle_log_mel():
    """The oracle's own restatement (explicit DFT) against the reference's output: what the GPU front-end is checked
    against in tests/test_gpu_audio.py."""
    g = gold("g6_logmel.npz")
    ex = ref_py.logmel_examples(ref_py.log_mel(g["wave"]))
    assert ex.shape == g["examples"].shape
    np.testing.assert_allclose(ex, g["examples"], rtol=0, atol=1e-10)
    assert ref_py.log_mel(np.zeros(399)).shape == (0, 64) and ref_py.log_mel(np.zeros(400)).shape == (1, 64)


# ------------------------------------------------------------------ G8 classic baseline (config 1)
def test_g8_classic_d1_p1_d2():
    g = gold("g8_classic.npz")
    d1, p1, s1 = ref_py.classic_d1_p1(g["frames"], 0.1)
    np.testing.assert_allclose(d1, g["d1"], rtol=2e-6, atol=1e-3)
    np.testing.assert_allclose(d1, g["d1_fast"], rtol=2e-6, atol=1e-3)
    np.testing.assert_allclose(p1, g["p1"], rtol=2e-4, atol=1e-7)
    assert abs(s1 - float(g["sigma1"])) / s1 < 1e-5
    d2, p2, s2 = ref_py.classic_d2(g["d1"], 0.1, filter_size=4)
    np.testing.assert_allclose(d2, g["d2"], rtol=1e-5, atol=1e-3)
    np.testing.assert_allclose(p2, g["p2"], rtol=2e-4, atol=1e-7)


# ------------------------------------------------------------------ oracle self-consistency
def test_blocked_sim_equals_definition():
    rng = np.random.default_rng(0)
    q = rng.standard_normal((37, 100)).astype(np.float32)
    t = rng.standard_normal((53, 100)).astype(np.float32)
    assert np.array_equal(cref.sim_f32(q, t, 0.1), cref.sim_f32(q, t, 0.1, naive=True))
    np.testing.assert_allclose(cref.sim_f32(q, t, 0.1), q.astype(np.float64) @ t.T.astype(np.float64) / 0.1, rtol=1e-5, atol=1e-4)


def test_target_order_matches_reference_construction():
    for L in (2, 3, 27, 30):
        for q in range(L):
            assert np.array_equal(cref.target_order(q, L), ref_py.target_segment_ids(q, L))
