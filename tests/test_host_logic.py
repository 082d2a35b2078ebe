"""Host-side logic of the product package against the oracle restatement and the reference-made fixtures."""
import os
import sys
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from oracle import ref_py

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_compat_window_map_equals_oracle_and_reference(avt):
    from avtex.texture import compat_window_frames

    rng = np.random.default_rng(1)
    for _ in range(60):
        W, S, mbs, G = int(rng.integers(4, 24)), int(rng.integers(1, 8)), int(rng.integers(2, 40)), int(rng.integers(1, 4))
        F_ = int(rng.integers(W + 3 * S + 2, 260))
        L = ref_py.num_segments(F_, W, S)
        if L < 3:
            continue
        q = int(rng.integers(0, L))
        a, sa = ref_py.compat_window_frames(q, F_, W, S, mbs, G)
        b, sb = compat_window_frames(q, F_, W, S, mbs, G)
        assert np.array_equal(a, b) and np.array_equal(sa, sb)
    for case in ("sf_th03", "sf_g2"):  # and against the windows the reference's encoder really received
        g = np.load(os.path.join(GOLD, "g5_validate_%s.npz" % case), allow_pickle=True)
        n_frames, W, S, mbs, G = [int(x) for x in g["cfg"][:5]]
        for step, q in enumerate(g["queries"]):
            wins, _ = compat_window_frames(int(q), n_frames, W, S, mbs, G)
            assert np.array_equal(g["window_frames"][step][1 : 1 + len(wins)], wins)


def test_split_helpers_match_reference_vectors(avt):
    g = np.load(os.path.join(GOLD, "g1_split.npz"))
    for i in range(5):
        n, mbs, W, S = [int(x) for x in g["ov%d_args" % i]]
        out, nv = avt.utils.split_into_overlapping_segments(torch.arange(1, n + 1).float().view(n, 1), mbs, W, S)
        assert np.array_equal(out.numpy()[..., 0], g["ov%d_out" % i]) and nv == g["ov%d_nvalid" % i]
    for i in range(4):
        n, mbs = [int(x) for x in g["sb%d_args" % i]]
        out, nv = avt.utils.split_into_batches(torch.arange(1, n + 1).float().view(1, n, 1), mbs)
        assert np.array_equal(out.numpy()[..., 0], g["sb%d_out" % i]) and nv == g["sb%d_nvalid" % i]
        assert avt.utils.combine_batches(out, nv).shape == (1, n, 1)


def test_dataset_negative_sampling_matches_reference(avt):
    """[A14] dataset.py:183-190 incl. its duplicate-prone hard negatives, under the same NumPy seeds."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from tools.gen_golden import make_video

    g = np.load(os.path.join(GOLD, "g9_dataset.npz"))
    args = SimpleNamespace(vdata="/tmp", adata=None, n_negs=10, img_size=16, enc_arch="slowfast", window=0, stride=0)
    torch.manual_seed(5)
    ds = avt.AudioVideoSegments(args, "g9", split="train", video=(make_video(3, 150, 16, 16), 20.0))
    assert len(ds) == int(g["len"]) and [args.window, args.stride] == list(g["window_stride"])
    for key in g.files:
        if not key.startswith("idx"):
            continue
        idx = int(key[3:])
        np.random.seed(100 + idx)
        pos, neg = ds.sample_ids(idx)
        assert [pos] + [int(x) for x in neg] == [int(x) for x in g[key]]
    np.random.seed(101)
    item = ds[1]
    assert len(item) == 6 and item[0][0].shape == (3, 8, 16, 16) and item[3][1].shape == (11, 3, 32, 16, 16)


def test_classic_baseline_matches_reference(avt):
    """Config 1 (CPU plumbing): D1/P1/D2 of baselines/classic_video_textures vs G8."""
    g = np.load(os.path.join(GOLD, "g8_classic.npz"))
    d1, p1, s1 = avt.classic.compute_D1(g["frames"], 0.1, batch_size=7)
    np.testing.assert_allclose(d1.numpy(), g["d1"], rtol=1e-5, atol=1e-3)
    np.testing.assert_allclose(p1.numpy(), g["p1"], rtol=1e-4, atol=1e-7)
    d2, p2, s2, _ = avt.classic.compute_D2(torch.from_numpy(g["d1"]), 0.1, filter_size=4)
    np.testing.assert_allclose(d2.numpy(), g["d2"], rtol=1e-5, atol=1e-3)
    np.testing.assert_allclose(p2.numpy(), g["p2"], rtol=1e-4, atol=1e-7)
    d3, p3, p3n, s3 = avt.classic.q_learning(torch.from_numpy(g["d2"]), 0.1)  # pinned: the reference's own q_learning run
    assert np.array_equal(d3.numpy(), g["d3"]) and np.array_equal(p3.numpy(), g["p3"])
    assert np.array_equal(p3n.numpy(), g["p3_thresholded"]) and float(s3) == float(g["sigma3"])
    assert torch.isfinite(p3).all() and (p3n.sum(1) > 0).all()
    seq = avt.classic.random_walk(p3n.numpy(), 30, rng=np.random.RandomState(0))
    assert len(seq) == 30 and max(seq) < p3n.shape[0]


def test_classic_config1_full_size_runs_on_cpu(avt):
    """BASELINE config 1: 200-frame 128x128 clip, raw-pixel L2 transition matrix, CPU only."""
    g = torch.Generator().manual_seed(7)
    frames = torch.randint(0, 256, (200, 128, 128, 3), generator=g, dtype=torch.uint8)
    d1, p1, sigma = avt.classic.compute_D1(frames, 0.1, batch_size=48)
    assert d1.shape == (200, 200) and torch.allclose(p1.sum(1), torch.ones(200), atol=1e-5)
    assert torch.allclose(d1, d1.t(), atol=1e-2) and (torch.diag(d1) == 0).all()


def test_cli_flags_match_reference(avt):
    p = avt.main.build_parser() if hasattr(avt, "main") else __import__("avtex.main", fromlist=["x"]).build_parser()
    a = p.parse_args(["-vdata", "v", "-ea", "slowfast", "-w", "20", "-stride", "4", "-temp", "0.1", "-th", "0.3", "-bs",
                      "24", "-e", "-mbs", "100", "-m", "2", "-alpha", "0.5", "-negs", "20", "-nintp"])
    assert (a.enc_arch, a.window, a.stride, a.threshold, a.mini_batchsize, a.evaluate) == ("slowfast", 20, 4, 0.3, 100, True)
    assert a.interpolation is False and a.model_type == 2 and a.stitch_mode == "compat" and a.vcam is False


def test_every_reference_flag_is_frozen_in_the_parser(avt):
    """every flag of the reference's main.py:41-296 with the same option strings, dest, default, type, action,
    choices and nargs.  tests/golden/g11_cli.npz holds the 47 add_argument calls as data (tools/gen_golden.py gen_g11 reads the
    reference's syntax tree); a default that drifts here fails by name."""
    import json
    import os

    import numpy as np

    rows = [json.loads(x) for x in np.load(os.path.join(os.path.dirname(__file__), "golden", "g11_cli.npz"))["flags"]]
    assert len(rows) == 47
    p = __import__("avtex.main", fromlist=["x"]).build_parser()
    by_dest = {}
    for act in p._actions:
        by_dest.setdefault(act.dest, act)
    kinds = {"_StoreAction": "store", "_StoreTrueAction": "store_true", "_StoreFalseAction": "store_false"}
    for r in rows:
        act = by_dest.get(r["dest"])
        assert act is not None, "flag %s (main.py:%d) is missing" % (r["opts"], r["line"])
        assert sorted(act.option_strings) == sorted(r["opts"]), (r["dest"], act.option_strings, r["opts"])
        assert kinds[type(act).__name__] == r["action"], (r["dest"], type(act).__name__, r["action"])
        assert act.default == r["default"] and type(act.default) is type(r["default"]), (r["dest"], act.default, r["default"])
        assert (act.type.__name__ if act.type is not None else None) == r["type"], (r["dest"], act.type, r["type"])
        assert (list(act.choices) if act.choices is not None else None) == r["choices"], (r["dest"], act.choices, r["choices"])
        nargs = act.nargs if r["action"] == "store" else None  # (argparse gives the flag actions an implicit nargs = 0)
        assert nargs == r["nargs"] and bool(act.required) == r["required"], (r["dest"], act.nargs, act.required)


def test_max_enc_batch_follows_the_image_size(avt):
    """the encoder batch is cut to what the kernels' 32-bit element offsets take at THIS image size."""
    from avtex.texture import max_enc_batch

    assert max_enc_batch(224) >= 249 and max_enc_batch(224) * 8 * 56 * 56 * 320 < (1 << 31)
    assert max_enc_batch(256) < 249 and max_enc_batch(256) * 8 * 64 * 64 * 320 < (1 << 31)
    assert max_enc_batch(224, planes=False) == 166 and max_enc_batch(64, planes=False) > 249


def test_checkpoint_keys_are_the_reference_prefixes(avt):
    from avtex.slowfast import SlowFast

    m = avt.ContrastivePredictionTemporal(SlowFast(), SlowFast(), avt.VGGish(), 2, 128, enc_arch="slowfast")
    keys = list(m.state_dict().keys())
    for pre in ("q_encoder.", "t_encoder.", "q_a_encoder.", "t_a_encoder.", "q_a_mlp.", "t_a_mlp."):
        assert any(k.startswith(pre) for k in keys), pre
    assert "q_encoder.s1.pathway0_stem.conv.weight" in keys and "t_encoder.s5.pathway1_res2.branch2.c_bn.weight" in keys
    assert "q_encoder.s1_fuse.conv_f2s.weight" in keys
    net, fc_dim = avt.ModelBuilder3D.build_network("resnet18", img_size=32, window=16, pretrained=False)
    assert fc_dim == 128 and net.eval()(torch.zeros(1, 3, 16, 32, 32)).shape[:2] == (1, 512)


def test_operator_and_train_refuse_cpu(avt):
    """No CPU fallback anywhere on the path: the operator (both branches), train() and the engine raise off-device."""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from tiny_encoders import TinySlowFast, seeded
    from avtex._lib import AvtError

    m = avt.ContrastivePredictionTemporal(seeded(TinySlowFast, 1), seeded(TinySlowFast, 2), None, 1, 128,
                                          enc_arch="slowfast", img_size=32, window=5, stride=2, mini_batchsize=2)
    q = [torch.zeros(2, 3, 8, 32, 32), torch.zeros(2, 3, 32, 32, 32)]
    t = [torch.zeros(2, 3, 3, 8, 32, 32), torch.zeros(2, 3, 3, 32, 32, 32)]
    m.train()
    with pytest.raises(AvtError, match="no CPU fallback"):
        m(q, t)
    with pytest.raises(AvtError, match="no CPU fallback"):
        avt.train([], m, None, None, 0)
    m.eval()
    with pytest.raises(AvtError, match="no CPU fallback"):
        m(q, torch.zeros(1, 12, 32, 32, 3), is_inference=True)
    with pytest.raises(AvtError, match="no CPU fallback"):
        avt.ops.l2norm_rows(torch.zeros(4, 8))


def test_pack_wfrag_is_the_documented_fragment_order(avt):
    """fused_slowfast.pack_wfrag against include/avt.h's description of avt_conv3d_igemm_wfrag_bf16's weight array, element
    by element on small shapes (host code: runs without a GPU)."""
    import torch
    from avtex.fused_slowfast import pack_wfrag

    for cout, cin, taps in ((40, 64, 3), (264, 32, 1), (32, 96, 9)):
        g = torch.Generator().manual_seed(cout + taps)
        w = torch.randn(cout, taps * cin, generator=g)
        f = pack_wfrag(w, cin, taps)
        nu = -(-taps * cin // 32)
        assert f.shape == (-(-cout // 256) * 8, -(-nu // 4) * 4 + 4, 2, 64, 8) and f.dtype == torch.bfloat16
        wb = w.to(torch.bfloat16)
        for tile, unit, ks, lane, e in ((0, 0, 0, 0, 0), (1, 2, 1, 37, 3), (0, nu - 1, 1, 63, 7), (1, nu, 0, 5, 1), (7, 1, 0, 31, 2)):
            row = tile * 32 + (lane & 31)
            cc, tap = (unit // taps, unit % taps) if taps > 1 else (unit, 0)
            ch = cc * 32 + ks * 16 + (lane >> 5) * 8 + e
            want = wb[row, tap * cin + ch] if (row < cout and unit < nu and ch < cin) else torch.tensor(0.0, dtype=torch.bfloat16)
            assert f[tile, unit, ks, lane, e] == want, (cout, cin, taps, tile, unit, ks, lane, e)


def test_split_planes_error_bounds(avt):
    """The contract-grade number format (include/avt.h): x = hi + lo to 2^-16 |x| (bf16 planes) / 2^-22 |x| for |x| >= 2^-3 and
    2^-24 absolute below (fp16 planes)."""
    from avtex import ops
    from avtex.fused_slowfast import split_planes

    g = torch.Generator().manual_seed(0)
    x = torch.cat([torch.randn(4096, generator=g) * 3, torch.randn(4096, generator=g) * 1e-3, torch.tensor([0.0, 1.0, -2.5, 60000.0])])
    for pd, dt in ((ops.X3_BF16, torch.bfloat16), (ops.X3_F16, torch.float16)):
        hi, lo = split_planes(x, pd)
        back = hi.view(dt).float() + lo.view(dt).float()
        err = (back.double() - x.double()).abs()
        if pd == ops.X3_BF16:
            assert (err <= 2.0 ** -16 * x.abs().double() + 1e-45).all()
        else:
            big = x.abs() >= 2.0 ** -3
            assert (err[big] <= 2.0 ** -22 * x[big].abs().double()).all() and (err[~big] <= 2.0 ** -24).all()


def test_pack_pw_planes_is_the_documented_fragment_order(avt):
    """include/avt.h: fragment [nt][ks][lane][e] = W[32*(nt/2) + 8*(r/4) + 4*(nt%2) + r%4][32*ks + 8*(lane>>4) + e], r = lane & 15,
    zero beyond K — for a raw 16-bit plane."""
    from avtex.fused_slowfast import pack_pw_planes

    n_out, k = 64, 80
    w = torch.arange(n_out * k, dtype=torch.int16).view(n_out, k).view(torch.bfloat16)
    f = pack_pw_planes(w).view(torch.int16)
    assert f.shape == (n_out // 16, 3, 64, 8)
    raw = w.view(torch.int16)
    for nt, ks, lane, e in ((0, 0, 0, 0), (1, 2, 17, 3), (3, 1, 63, 7), (2, 2, 40, 5)):
        r, q = lane & 15, lane >> 4
        row, col = 32 * (nt // 2) + 8 * (r // 4) + 4 * (nt % 2) + r % 4, 32 * ks + 8 * q + e
        assert int(f[nt, ks, lane, e]) == (int(raw[row, col]) if col < k else 0)


def test_walk_from_survivors_consumes_numpy_like_validate(avt):
    """agreement.walk_from_survivors (the rank-0 walk of the sharded validate): one rng.choice per step over the survivor
    positions, first step appends W frames, later steps S (validate.py:570-615)."""
    from avtex import agreement

    n, W, S = 9, 6, 2
    idx = np.tile(np.arange(3), (n, 1)).astype(np.int32)
    seg = np.stack([np.array([(q + 1) % n, (q + 4) % n, (q + 6) % n]) for q in range(n)]).astype(np.int32)
    cnt = np.array([3, 1, 2, 3, 3, 1, 2, 3, 3], np.int32)
    frames, chosen = agreement.walk_from_survivors(idx, seg, cnt, n * S + W, W, S, 20, q_id=3, rng=np.random.RandomState(4))
    rng, q, want = np.random.RandomState(4), 3, []
    while len(want) < 20:
        q = int(seg[q, rng.choice(idx[q, : cnt[q]])])
        want.extend(range(q * S, q * S + W) if not want else range(q * S + W - S, q * S + W))
    assert frames == want and len(chosen) == 1 + (len(frames) - W) // S


def test_synth_inputs_are_deterministic_and_structured(avt):
    from avtex import synth

    a, b = synth.structured_video(3, 100, 16, 16), synth.structured_video(3, 100, 16, 16)
    assert a.dtype == torch.uint8 and a.shape == (100, 16, 16, 3) and torch.equal(a, b)
    d_near = (a[10].float() - a[11].float()).abs().mean()
    d_far = (a[10].float() - a[80].float()).abs().mean()
    assert d_near * 3 < d_far  # neighbouring frames similar, distant scenes not
    m = synth.randomise_bn(torch.nn.Sequential(torch.nn.Conv3d(3, 4, 1), torch.nn.BatchNorm3d(4)), 1, 0.5)
    assert float(m[1].weight.min()) >= 0.5 and float(m[1].running_var.min()) >= 0.8


def test_save_video_raw_keeps_frames_without_ffmpeg(avt, tmp_path, monkeypatch):
    """Output side (validate.py:789-872): the stitched frames go out by index gather, no PNG folder; without an ffmpeg binary
    they are kept losslessly in the .npz container read_video() reads back."""
    import shutil as _sh

    from avtex.utils import save_video_raw
    from avtex.validate import read_video

    monkeypatch.setattr(_sh, "which", lambda name: None)
    video = torch.randint(0, 256, (12, 8, 10, 3), dtype=torch.uint8)
    frames = video[torch.tensor([3, 4, 5, 9, 10, 0])]
    out = str(tmp_path / "clip.mp4")
    assert save_video_raw(frames, out, 10) is False
    back, fps = read_video(out)
    assert fps == 10.0 and torch.equal(back, frames)
    with pytest.raises(ValueError):
        save_video_raw(frames.float(), out, 10)


def test_interpolated_timeline_follows_the_reference_bookkeeping():
    """validate.py:588-650: every source frame 1 + int((SF-1)/2) times; at a jump the last frame's copies are taken back,
    SF - 1 interpolated frames go in and the first frame after the jump is shown once — so the reference's own check
    (validate.py:812) len == int((SF+1)/2) * len(new_frames) holds for every odd SF."""
    import torch
    from avtex import slowmo
    from avtex._lib import AvtError
    video = torch.arange(40, dtype=torch.uint8).view(40, 1, 1, 1).expand(40, 2, 2, 3).contiguous()
    for sf in (3, 5, 7):
        tl = slowmo.IntpTimeline(sf)
        shown = []
        for idx in (4, 5, 6):
            tl.append(idx)
            shown.append(idx)
        marks = [torch.full((2, 2, 3), 200 + k, dtype=torch.uint8) for k in range(sf - 1)]
        tl.jump(marks)
        for n, idx in enumerate((20, 21)):
            tl.append(idx, first_after_jump=n == 0)
            shown.append(idx)
        assert len(tl) == int((sf + 1) / 2) * len(shown)
        seq = tl.frames(video)[:, 0, 0, 0].tolist()
        d = int((sf - 1) / 2)
        expect = [4] * (1 + d) + [5] * (1 + d) + [6] + [200 + k for k in range(sf - 1)] + [20] + [21] * (1 + d)
        assert seq == expect
    for bad in (1, 2, 4):
        with pytest.raises(AvtError):
            slowmo.IntpTimeline(bad)


def test_training_passes_fall_back_to_torch_off_the_device():
    """train_ops.bn_act / conv3d on CPU tensors (or in eval mode) are the stock torch ops — the fused HIP passes only take
    channels-last fp32 device tensors in train mode — so the SlowFast module keeps working as a plain nn.Module."""
    import torch.nn as nn
    import torch.nn.functional as F
    from avtex import train_ops
    torch.manual_seed(0)
    bn = nn.BatchNorm3d(16).train()
    x, r = torch.randn(2, 16, 2, 5, 5), torch.randn(2, 16, 2, 5, 5)
    ref = nn.BatchNorm3d(16).train()
    assert not train_ops.fusable(x, bn, r)
    assert torch.equal(train_ops.bn_act(x, bn, res=r, relu=True), F.relu(ref(x) + r))
    assert torch.equal(bn.running_mean, ref.running_mean) and int(bn.num_batches_tracked) == 1
    conv = nn.Conv3d(16, 8, (1, 3, 3), padding=(0, 1, 1), bias=False).train()
    assert not train_ops.conv_fusable(x, conv)
    assert torch.equal(train_ops.conv3d(x, conv), conv(x))


def test_frames_bar_matches_the_reference_slicing():
    """validate.py:634-638 in NumPy, as the reference writes it, against validate.frames_bar."""
    import importlib
    V = importlib.import_module("avtex.validate")  # (the package re-exports the function under the module's name)
    rng = np.random.RandomState(0)
    video = rng.randint(0, 256, (50, 40, 64, 3)).astype(np.uint8)
    ids = [0, 1, 2, 3, 25, 49, None]
    got = torch.from_numpy(np.stack([video[i if i is not None else 0] for i in ids]).copy())
    V.frames_bar(got, ids, 50)
    for k, idx in enumerate(ids):
        frame_arr = np.array(video[idx if idx is not None else 0])
        bar = np.zeros((15, video.shape[-2], 3))
        if idx is not None:
            frame_n = int(idx * video.shape[-2] / 50)
            bar[:, frame_n - 3 : frame_n + 3, :] = [255, 0, 0]
        frame_arr[-25:-10, :, :] = bar
        assert np.array_equal(got[k].numpy(), frame_arr), (k, idx)
    assert got[0, -25:-10].sum() == 0 and got[4, -25:-10, :, 0].sum() == 15 * 6 * 255   # idx 0: start -3 -> nothing drawn


def _load_bench():
    import importlib.util

    spec = importlib.util.spec_from_file_location("avt_bench", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_bench_line_is_compact_and_round_trips():
    """The driver's capture holds about 8 KB of stdout: the one JSON line must stay under bench.LINE_LIMIT (4096 bytes) with
    every field of the contract in it, and must survive json.loads (round 2's 20 KB line came back with parsed = null)."""
    import json

    bench = _load_bench()
    long_name = "conv_x3_kernel<128,128,64,f16>" + "x" * 40
    out = {
        "metric": bench.baseline_metric(), "value": 1234.56789123, "unit": "clip-windows/s", "n_gpus": 8, "steps": 20, "warmup": 5,
        "ms_per_step": 3318.123456789, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f16x3",
        "data": "synthetic",
        "config": {"workload": "w" * 260, "windows_per_gpu": 4096, "windows_total": 32768, "embedding_dim": 2304,
                   "encoder_precision": "f16x3 (contract grade, split-plane MFMA)", "sim_precision": "f32", "encoder_streams": 2,
                   "parallelism": "windows sharded x8, all-gather(T_hat)"},
        "roofline": {"kernel": long_name, "bound": "mfma", "achieved": 321.1343313029423, "peak": 833.3333333333334,
                     "unit": "TFLOP/s", "frac": 0.38536119756353077, "traffic": 2128169230.0,
                     "encoder_family_achieved": 202.33, "encoder_family_frac": 0.2428, "step_frac": 0.24},
        "nxn_build_ms": 0.9123456, "survivors_per_row": 3332.123, "fast_mode_value": 2874.123, "fast_mode_ms_per_step": 1425.0,
        "precision_max_abs_dscore": 2.288818359375e-05, "precision_windows": 128,
        "frames_lists_identical": {"0.0": "3/3", "0.3": "3/3"}, "nxn_build_ms_seeded_th0": 0.91,
        "cpu_baseline": {"value": 1.4544949820793884, "unit": "clip-windows/s", "cores": 16, "kind": "port", "sample": "s" * 200},
        "train_clips_per_s": 174.123, "train_ms_per_step": 735.12,
        "hbm_GBps": {"step_pmc": 2987.123456, "pmc_coverage": 0.9712345, "hbm_bound_family_algorithmic": 4123.123456, "peak": 8000.0},
        "value_inputs_r03": 1301.123456, "value_inputs_r03_steps": 4, "ms_per_step_rank_min": 3301.123456, "ms_per_step_rank_max": 3318.123456,
        "rccl_ranks": 8, "allgather_ms": 0.123456, "allgather_bytes_per_rank": 37748736,
        "trained_weights": {"value": 1388.123456, "survivor_fraction": 0.0712345, "precision_max_abs_dscore": 3.1e-05, "frames_lists_identical": "3/3"},
    }
    line = bench.compact_line(out)
    assert len(line.encode()) < bench.LINE_LIMIT <= 4096 and "\n" not in line
    back = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in back
    assert back["roofline"]["frac"] == pytest.approx(0.385361, rel=1e-5) and back["config"]["workload"] == "w" * 260
    assert set(back["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"}
    assert set(back["cpu_baseline"]) >= {"value", "unit", "cores", "kind", "sample"}
    with pytest.raises(RuntimeError):
        bench.compact_line(dict(out, junk="j" * 4096))


def test_cli_and_bench_usage_text_renders(avt):
    """argparse expands help strings with `help % params`: a bare '%' in one of them makes --help (and format_help) raise
    TypeError (both parsers had one).  Every help string of the product CLI and of bench.py must render."""
    p = avt.main.build_parser() if hasattr(avt, "main") else __import__("avtex.main", fromlist=["x"]).build_parser()
    assert "--enc_batch" in p.format_help()
    text = _load_bench().build_parser().format_help()
    assert "--enc-batch" in text and "--config" in text and "--dist-backend" in text
    a = _load_bench().build_parser().parse_args([])
    assert a.dist_backend == "nccl" and a.gpus == 1  # RCCL unless asked otherwise


def test_bench_config4_is_one_flag(monkeypatch):
    """`bench.py --config 4` = BASELINE.json's config 4 on one rank: 2048 windows per GPU and the top-k (k = 8) leg."""
    bench = _load_bench()
    a = bench.build_parser().parse_args(["--config", "4", "--gpus", "8", "--sim-precision", "bf16x3"])
    assert a.config == 4 and a.gpus == 8 and a.sim_precision == "bf16x3"
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "args.windows = 2048" in src and "args.topk = args.topk or 8" in src


def test_bench_self_launch_command(monkeypatch):
    """`python bench.py --gpus N` with no launcher around it starts N ranks as child processes (never an exec of a process that
    has touched the GPU) with the rendezvous on 127.0.0.1 and the same arguments."""
    import subprocess

    bench = _load_bench()
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 0

    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3", "--warmup", "1"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 0
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-6:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" or "HSA_ENABLE_IPC_MODE_LEGACY" in os.environ


def _caffe2_names_of(key):
    """TEST-side inverse, written from the caffe2 naming convention (not from checkpoint.py): a `SlowFast` state-dict key ->
    the blob name the Kinetics model-zoo pickle uses for it."""
    bn = {"weight": "s", "bias": "b", "running_mean": "rm", "running_var": "riv"}
    parts = key.split(".")
    if parts[0].endswith("_fuse"):  # s1_fuse.conv_f2s.weight / s3_fuse.bn.running_var
        stage = int(parts[0][1])
        last_block = {2: 2, 3: 3, 4: 5}  # the fusion hangs off the LAST fast block of stages 2-4 (depths 3, 4, 6)
        pre = "t_pool1_subsample" if stage == 1 else "t_res%d_%d_branch2c_bn_subsample" % (stage, last_block[stage])
        return pre + ("_w" if parts[1] == "conv_f2s" else "_bn_" + bn[parts[2]])
    if parts[0] == "s1":  # s1.pathway1_stem.conv.weight
        t = "t_" if parts[1].startswith("pathway1") else ""
        return t + ("conv1_w" if parts[2] == "conv" else "res_conv1_bn_" + bn[parts[3]])
    stage = int(parts[0][1])
    t = "t_" if parts[1].startswith("pathway1") else ""
    idx = int(parts[1].split("_res")[1])
    pre = "%sres%d_%d_" % (t, stage, idx)
    if parts[2] == "branch1":
        return pre + "branch1_w"
    if parts[2] == "branch1_bn":
        return pre + "branch1_bn_" + bn[parts[3]]
    name = parts[3]  # branch2.a.weight / branch2.b_bn.bias
    return pre + ("branch2%s_w" % name if "_bn" not in name else "branch2%s_bn_%s" % (name[0], bn[parts[4]]))


def test_kinetics_caffe2_checkpoint_loads_into_slowfast(tmp_path, monkeypatch):
    """models.py:565-580: the reference's encoders start from the caffe2 pickle SLOWFAST_8x8_R50.pkl.  A synthetic blob dict
    in the model zoo's naming covers EVERY parameter and running statistic of slowfast.SlowFast (+ the classifier, solver
    momentum blobs and scalars a real file carries, which must be dropped); ModelBuilder3D loads it through
    AVT_PRETRAINED_SLOWFAST and every tensor arrives in its place."""
    import pickle

    import avtex
    from avtex import checkpoint
    from avtex.slowfast import SlowFast

    torch.manual_seed(0)
    src = SlowFast()
    want = {k: v for k, v in src.state_dict().items() if not k.endswith("num_batches_tracked")}
    rng = np.random.RandomState(0)
    blobs, expect = {}, {}
    for k, v in want.items():
        name = _caffe2_names_of(k)
        assert name not in blobs, (k, name)
        blobs[name] = rng.standard_normal(tuple(v.shape)).astype(np.float32)
        expect[k] = blobs[name]
        if name.endswith("_w") or name.endswith("_s") or name.endswith("_b"):
            blobs[name + "_momentum"] = np.zeros(tuple(v.shape), np.float32)
    assert len(blobs) > 2 * len(want) * 0.5 and "t_res4_5_branch2c_bn_subsample_bn_riv" in blobs and "res5_2_branch2c_bn_s" in blobs
    blobs.update({"pred_w": np.zeros((400, 2304), np.float32), "pred_b": np.zeros(400, np.float32),
                  "lr": np.float32(0.1), "model_iter": np.float32(1.0)})
    path = tmp_path / "SLOWFAST_8x8_R50.pkl"
    with open(path, "wb") as f:
        pickle.dump({"blobs": blobs}, f, protocol=2)
    # every model key is hit exactly once, nothing else maps anywhere
    sd, dropped = checkpoint.convert_caffe2_slowfast(blobs)
    assert set(sd) == set(want) and {"pred_w", "pred_b", "lr", "model_iter"} <= set(dropped)
    assert all(d.endswith("_momentum") or d in ("pred_w", "pred_b", "lr", "model_iter") for d in dropped)
    monkeypatch.setenv("AVT_PRETRAINED_SLOWFAST", str(path))
    model, fc_dim = avtex.ModelBuilder3D.build_network("slowfast", 224, 20, pretrained=True)
    assert fc_dim == 128
    got = model.state_dict()
    for k, v in expect.items():
        assert np.array_equal(got[k].numpy(), v), k
    # a file with a missing blob is rejected, not half-loaded
    del blobs["t_res3_1_branch2b_w"]
    with open(path, "wb") as f:
        pickle.dump({"blobs": blobs}, f, protocol=2)
    with pytest.raises(ValueError, match="missing"):
        checkpoint.load_kinetics_slowfast(SlowFast(), str(path))


def test_committed_pmc_summary_resolves_the_headline_kernels_traffic():
    """bench.py's roofline.traffic comes from the committed rocprofv3 PMC summary, matched by device kernel symbol: a renamed
    template parameter list must not silently turn it into null (it did once: conv_x3_xl_kernel<true> -> <true, 1>)."""
    import types

    import bench
    if not os.path.exists(bench.PMC_SUMMARY):
        pytest.skip("no committed PMC summary")
    args = types.SimpleNamespace(windows=4096, enc_batch=bench.PMC_BATCH, sim_precision="f32")
    kern = [{"kernel": k} for k in ("conv_x3_xl_kernel<f16>", "pw_x3_kernel<f16>", "stem_kernel<x3>", "bneck_x3_kernel",
                                    "conv33_x3_kernel", "sim_gemm_nt", "row_transition", "clip_pack", "l2norm_rows")]
    bench.attach_pmc_traffic(kern, args, "f16x3")
    missing = [k["kernel"] for k in kern if not k.get("traffic")]
    assert not missing, missing


def test_committed_roofline_follows_from_the_committed_rocprof_stats():
    """the roofline of the committed bench line must be reproducible from the committed profile of the SAME
    command — tools/roofline_from_rocprof.py divides the per-symbol algorithmic flops per step (bench_detail) by the profiler's
    total duration of that symbol; the dominant kernel's fraction in the line (HIP events, sampled in the run) agrees within 5 %."""
    import importlib.util

    prof_dir = os.path.join(ROOT, "profiles", "r06")
    stats, detail = os.path.join(prof_dir, "rocprof_kernel_stats.csv"), os.path.join(prof_dir, "bench_detail_profiled.json")
    if not (os.path.exists(stats) and os.path.exists(detail)):
        pytest.skip("no committed round-6 profile pair")
    spec = importlib.util.spec_from_file_location("roofline_from_rocprof", os.path.join(ROOT, "tools", "roofline_from_rocprof.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    out = mod.recompute(stats, detail)
    dom = out["dominant"]
    assert dom is not None and dom["rocprof_achieved"], out
    assert abs(dom["rocprof_frac"] / out["line_frac"] - 1.0) < 0.05, (dom, out["line_frac"])
    # every encoder symbol of the line resolves to profiler rows, and the launches the line counts are the launches the profiler saw
    for r in out["rows"]:
        assert r["rocprof_achieved"], r
        assert abs(r["rocprof_calls_per_step"] / r["bench_launches_per_step"] - 1.0) < 0.02, r
    # the breakdown sums to about the wall time (the sampler extrapolates by clips from full batches)
    import json

    head = json.load(open(detail))["headline"]
    b = head["breakdown_ms_per_step"]
    assert 0.9 < sum(v for k, v in b.items() if k != "wall") / b["wall"] < 1.1, b


def test_bench_symbol_matching_names():
    import bench

    xl = "void (anonymous namespace)::conv_x3_xl_kernel<true, false, false>((anonymous namespace)::ConvArgs)"
    assert bench.symbol_matches("conv_x3_xl_kernel<f16>", xl, "f16x3") and not bench.symbol_matches("conv_x3_xl_kernel<f16>", xl, "bf16x3")
    pw = "void (anonymous namespace)::pw_x3_kernel<8, 8, true>((anonymous namespace)::PxArgs)"
    assert bench.symbol_matches("pw_x3_kernel<f16>", pw, "f16x3") and not bench.symbol_matches("pw_x3_kernel<bf16>", pw, "f16x3")
    ct = "void (anonymous namespace)::conv_x3_kernel<128, 128, 64, true, false, false>((anonymous namespace)::ConvArgs)"
    assert bench.symbol_matches("conv_x3_kernel<128,128,64,f16>", ct, "f16x3")
    assert not bench.symbol_matches("conv_x3_kernel<128,64,64,f16>", ct, "f16x3")
    st = "void (anonymous namespace)::stem_kernel<7, false, 2>((anonymous namespace)::StemArgs)"
    assert bench.symbol_matches("stem_kernel<x3>", st, "f16x3") and not bench.symbol_matches("stem_kernel", st, "f16x3")
    pc = "void (anonymous namespace)::pw_chain_x3_kernel<2, 16, 4, true>((anonymous namespace)::PcArgs)"
    assert bench.symbol_matches("pw_chain_x3_kernel", pc, "f16x3") and not bench.symbol_matches("pw_x3_kernel<f16>", pc, "f16x3")


def test_pack_res2_x3_is_the_documented_stream_order(avt):
    """fused_slowfast.pack_res2_x3 -> include/avt.h avt_res2_x3's weight stream: [17 chunks][8 pairs][2 planes][64 lanes][8]; chunks 0-3 = a
    (pair 4 kk + nt: k-step 2 chunk + kk, n-tile nt), 4-12 = b (one tap each, pair 4 kk + nt), 13-16 = c (pair 2 nn + kk: n-tile
    4 (chunk - 13) + nn); lane l of a pair holds W[channel(nt, l & 15)][32 k + 8 (l >> 4) + e], channel(nt, r) = 32 (nt / 2) + 8 (r / 4)
    + 4 (nt % 2) + r % 4; fp16 planes scaled per output channel into [2^9, 2^10), coef = [1/sa | ba | 1/sb | bb | 1/sc | bc]."""
    import avtex.fused_slowfast as fsf
    from avtex import ops

    torch.manual_seed(4)
    wa, wb, wc = torch.randn(64, 256, 1, 1, 1), torch.randn(64, 64, 1, 3, 3), torch.randn(256, 64, 1, 1, 1)
    ba, bb, bc = torch.randn(64), torch.randn(64), torch.randn(256)
    wf, cf = fsf.pack_res2_x3(wa, ba, wb, bb, wc, bc, ops.X3_F16, "cpu")
    assert tuple(wf.shape) == (136, 2, 64, 8) and wf.dtype == torch.bfloat16 and tuple(cf.shape) == (768,)
    val = wf.view(torch.float16).float()
    full = val[:, 0] + val[:, 1]                      # hi + lo: the scaled weight to 2^-22
    sa, sb, sc = 1.0 / cf[0:64], 1.0 / cf[128:192], 1.0 / cf[256:512]
    assert torch.equal(cf[64:128], ba) and torch.equal(cf[192:256], bb) and torch.equal(cf[512:768], bc)
    for scale, w in ((sa, wa), (sb, wb), (sc, wc)):   # powers of two that bring every channel's largest weight into [2^9, 2^10)
        assert torch.equal(torch.exp2(torch.round(torch.log2(scale))), scale)
        mx = (w.reshape(w.shape[0], -1).abs().amax(1) * scale)
        assert bool(((mx >= 512) & (mx < 1024)).all())
    chan = lambda nt, r: 32 * (nt // 2) + 8 * (r // 4) + 4 * (nt % 2) + r % 4
    rng = np.random.RandomState(0)
    for _ in range(200):
        chunk, pr, lane, e = int(rng.randint(17)), int(rng.randint(8)), int(rng.randint(64)), int(rng.randint(8))
        r, q = lane & 15, lane >> 4
        if chunk < 4:
            kk, nt = pr // 4, pr % 4
            o, k = chan(nt, r), 32 * (2 * chunk + kk) + 8 * q + e
            want = wa[o, k, 0, 0, 0] * sa[o]
        elif chunk < 13:
            kk, nt, tap = pr // 4, pr % 4, chunk - 4
            o, k = chan(nt, r), 32 * kk + 8 * q + e
            want = wb[o, k, 0, tap // 3, tap % 3] * sb[o]
        else:
            nn, kk = pr // 2, pr % 2
            o, k = chan(4 * (chunk - 13) + nn, r), 32 * kk + 8 * q + e
            want = wc[o, k, 0, 0, 0] * sc[o]
        got = full[chunk * 8 + pr, lane, e]
        assert abs(float(got) - float(want)) <= 2.0 ** -21 * abs(float(want)) + 1e-30, (chunk, pr, lane, e)


def test_concatenation_protocol_host_side(avt):
    """train_ops.join_channels / _row_ld / _alias (the in-place lateral fusion of the training step): on the CPU — where no
    producer tags its output — the join is torch.cat; the row-pitch detector accepts channel slices of channels-last tensors
    and contiguous ones, and refuses anything else; an alias shares storage without being an autograd view."""
    from avtex import train_ops

    a, b = torch.randn(2, 8, 3, 4, 5), torch.randn(2, 4, 3, 4, 5)
    assert torch.equal(train_ops.join_channels(a, b), torch.cat([a, b], 1))
    buf = torch.zeros((2, 12, 3, 4, 5)).contiguous(memory_format=torch.channels_last_3d)
    assert train_ops._row_ld(buf) == 12 and train_ops._row_ld(buf[:, :8]) == 12 and train_ops._row_ld(buf[:, 8:]) == 12
    assert train_ops._row_ld(torch.zeros(2, 12, 3, 4, 5)) is None            # NCDHW memory
    assert train_ops._row_ld(buf[:, :, :, ::2]) is None                       # rows without a constant pitch
    assert train_ops._row_ld(buf[:, 2:10]) is None                            # a slice that is not 16-byte aligned
    assert train_ops._row_ld(buf.double()) is None
    one = torch.zeros((1, 16, 1, 1, 7)).contiguous(memory_format=torch.channels_last_3d)
    assert train_ops._row_ld(one[:, :8]) == 16                                # size-1 dimensions: their strides do not matter
    y = train_ops._alias(buf, 8, 4)
    assert y._base is None and y.shape == (2, 4, 3, 4, 5) and y.data_ptr() == buf.data_ptr() + 8 * 4
    y.fill_(1.0)
    assert float(buf[:, 8:].sum()) == y.numel() and float(buf[:, :8].abs().sum()) == 0.0
    # tags that do not describe the two halves of one buffer fall back to the copy
    a2, b2 = train_ops._alias(buf, 0, 8), torch.randn(2, 4, 3, 4, 5)
    a2._avt_cat, b2._avt_cat = (buf, 0), (torch.zeros_like(buf), 8)
    assert torch.equal(train_ops.join_channels(a2, b2), torch.cat([a2, b2], 1))


def test_hardware_queue_default_is_set_before_hip_initialises():
    """The package asks for 8 HIP hardware queues unless the environment already says otherwise (profiles/r05/hw_queues.log: the
    three-stream training step shares queues with any other stream of the process on HIP's default of 4)."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = "import os, sys; sys.path.insert(0, %r); import avtex; print(os.environ.get('GPU_MAX_HW_QUEUES'))" % root
    env = {k: v for k, v in os.environ.items() if k != "GPU_MAX_HW_QUEUES"}
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.strip().splitlines()[-1] == "8"
    env["GPU_MAX_HW_QUEUES"] = "4"
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.strip().splitlines()[-1] == "4"


def test_isa_loop_tools_read_a_loop(tmp_path, capsys, monkeypatch):
    """tools/diag/isa_loop_summary.py / isa_loop_mix.py on a hand-made listing: the loop is found by its backward branch, the
    wait in front of the ds_write shows up between the runs of loads and MFMAs, the mix counts by instruction class."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    listing = "\n".join([
        "_ZN12_GLOBAL__N_18toy_kernelEv:",
        "\ts_load_dwordx2 s[0:1], s[4:5], 0x0",
        ".LBB0_1:",
        "\tbuffer_load_dwordx4 v[0:3], v8, s[0:3], 0 offen",
        "\tbuffer_load_dwordx4 v[4:7], v9, s[0:3], 0 offen",
        "\tv_mfma_f32_32x32x16_bf16 v[16:31], v[10:13], v[12:15], v[16:31]",
        "\tv_add_u32_e32 v8, 64, v8",
        "\ts_waitcnt vmcnt(0)",
        "\tds_write_b128 v40, v[0:3]",
        "\ts_cbranch_scc1 .LBB0_1",
        "\ts_endpgm",
    ])
    path = tmp_path / "toy.s"
    path.write_text(listing)
    for name in ("isa_loop_summary", "isa_loop_mix"):
        spec = importlib.util.spec_from_file_location(name, os.path.join(root, "tools", "diag", name + ".py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        monkeypatch.setattr(sys, "argv", [name, str(path), "toy_kernel"])
        mod.main()
    out = capsys.readouterr().out
    assert "vmload x2" in out and "s_waitcnt vmcnt(0)" in out and "ds_write x1" in out  # the summary
    assert "'mfma': 1" in out and "'valu': 1" in out and "'vmem': 2" in out and "v_add_u32 x1" in out  # the mix
