#!/usr/bin/env python3
"""Convergence evidence for the training arithmetic, and the train -> checkpoint -> `-e` round trip.

    python tools/train_convergence.py --steps 300 --lr 1e-2 --modes x3 fp32 --out gpurun_out/train_convergence.json

For every mode ("x3" = the hand-written split-plane convolutions, train_ops' default; "fp32" = MIOpen's fp32 convolutions,
the reference's arithmetic, `main.py --train_conv fp32`) the SAME model (same seed), the SAME structured synthetic video and the
SAME batches (the device batcher re-seeded from the same NumPy state, the same item indices) run `--steps` optimizer steps of
BASELINE config 5 — batch of 8 items x (1 query + 1 positive + 14 negatives) at 224^2 through the real SlowFast-8x8-R50 pair in
train mode, per-item BatchNorm groups, HIP InfoNCE + CE, SGD momentum 0.9 wd 1e-4 (README.md:38 / main.py:440-446; the learning
rate is an argument: the README's 1e-4 is for Kinetics-pretrained encoders that this box does not have).  Recorded per mode: the
per-step loss, its EMA (0.9), the step time.  The reference's loop is train.py:114-141, its stop rule main.py:475-477.

`--roundtrip`: the pair trained by the FIRST mode is saved with the reference's checkpoint keys (main.py:464-470), loaded back
through `main.py -e --resume` on the training video (an .npz beside the checkpoint), and held to the contract on its own weights:
f16x3 MFMA encoders against the fp32 nn.Modules on the same frames (agreement.compare_tables), survivor fraction at th 0.3, and
the largest |activation| of every BatchNorm / block output against the fp16 planes' clamp at 65504.
"""
import argparse
import contextlib
import json
import os
import sys
import time
from types import SimpleNamespace

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def build_model(dev, init):
    """The operator as main.py builds it (VGGish and the never-called MLPs included, so that the checkpoint loads by key)."""
    import avtex
    from avtex import ops, synth
    from avtex.slowfast import SlowFast

    torch.manual_seed(0)
    if init == "bench":  # bench.py's r04 pair: sparse, input-dependent features; t = a slightly diverged copy of q (the reference's
        # two encoders start from one Kinetics checkpoint, main.py:329-334)
        q = synth.randomise_bn(SlowFast(), 10, 2.0, 0.1)
        t = synth.perturbed_copy(q, 11, 0.05)
    else:  # PySlowFast's own initialisation (kaiming, zero last BatchNorm scale)
        q, t = SlowFast(), SlowFast()
    model = avtex.ContrastivePredictionTemporal(q, t, avtex.VGGish(), 1, 2304, temp=0.1, window=15, stride=6,
                                                enc_arch="slowfast", img_size=224)
    return model.to(dev)


def train_run(mode, args, dev, video, keep=False):
    import avtex
    from avtex import train_ops
    from avtex.dataset import DeviceSegmentBatcher

    B, negs = 8, 14
    train_ops.set_conv_mode(mode)
    train_ops.invalidate_weight_cache()
    dargs = SimpleNamespace(vdata="/tmp", adata=None, n_negs=negs, img_size=224, enc_arch="slowfast", window=0, stride=0)
    torch.manual_seed(5)
    with contextlib.redirect_stdout(sys.stderr):
        ds = avtex.AudioVideoSegments(dargs, "synthetic", split="train", video=(video, 30.0))
    model = build_model(dev, args.init).train().to(memory_format=torch.channels_last_3d)
    params = [p for n, p in model.named_parameters() if n.startswith(("q_encoder.", "t_encoder."))]
    opt = torch.optim.SGD(params, lr=args.lr, momentum=0.9, weight_decay=1e-4)
    crit = avtex.InfoNCECriterion()
    bat = DeviceSegmentBatcher(ds, dev)
    np.random.seed(1)
    bat.seed_from_numpy()
    rng = np.random.RandomState(99)
    labels = torch.zeros(B, dtype=torch.long, device=dev)
    losses, t_steps, top1 = [], [], []
    torch.cuda.synchronize()
    # --epochs E: the reference's own loop shape — every epoch walks a shuffled permutation of the segments in
    # batches of 8 (main.py:195-198, DataLoader shuffle), the epoch loss is the mean of its steps' losses (train.py:210) and
    # training STOPS when it falls below --stop-loss (main.py:475-477: `if loss < 0.07: break`)
    # (callers that pass a bare namespace — bench.py --weights trained, the suite's 40-step form — get the step mode)
    epochs, stop_loss, decay_at = getattr(args, "epochs", 0), getattr(args, "stop_loss", 0.07), list(getattr(args, "lr_decay_epochs", []))
    per_epoch = (len(ds) // B) if epochs else 0
    n_steps = epochs * per_epoch if epochs else args.steps
    epoch_loss, stopped_at, perm = [], None, None
    for it in range(n_steps):
        t0 = time.perf_counter()
        if epochs:
            if it % per_epoch == 0:
                perm = rng.permutation(len(ds))
                # the reference's schedule shape (main.py:448-449: StepLR, x 0.1 at fixed epochs): --lr-decay-epochs
                if (it // per_epoch) in decay_at:
                    for g in opt.param_groups:
                        g["lr"] *= 0.1
                    print("[%s] epoch %d: lr -> %g" % (mode, it // per_epoch, opt.param_groups[0]["lr"]), file=sys.stderr, flush=True)
            ids = [int(i) for i in perm[(it % per_epoch) * B : (it % per_epoch + 1) * B]]
        else:
            ids = [int(i) for i in rng.randint(0, len(ds), size=B)]
        q, t, _, _ = bat.batch(torch.tensor(ids))
        q = [v.contiguous(memory_format=torch.channels_last_3d) for v in q]
        opt.zero_grad(set_to_none=True)
        with train_ops.bn_replicas(B):
            out = model(q, t)
        loss = crit(out.float(), labels)
        loss.backward()
        opt.step()
        train_ops.invalidate_weight_cache()
        losses.append(float(loss))
        top1.append(float((out.detach().argmax(1) == 0).float().mean()))  # items whose positive has the best logit
        t_steps.append(time.perf_counter() - t0)
        if it % 20 == 0 or it == args.steps - 1:
            print("[%s] step %d loss %.4f top1 %.2f (%.0f ms)" % (mode, it, losses[-1], float(np.mean(top1[-20:])), t_steps[-1] * 1e3),
                  file=sys.stderr, flush=True)
        if not np.isfinite(losses[-1]):
            break
        if epochs and (it + 1) % per_epoch == 0:
            epoch_loss.append(float(np.mean(losses[-per_epoch:])))
            print("[%s] epoch %d loss %.4f" % (mode, len(epoch_loss) - 1, epoch_loss[-1]), file=sys.stderr, flush=True)
            if epoch_loss[-1] < stop_loss:
                stopped_at = len(epoch_loss) - 1
                break
    ema, e = [], None
    for v in losses:
        e = v if e is None else 0.9 * e + 0.1 * v
        ema.append(e)
    rec = {"mode": mode, "lr": args.lr, "steps": len(losses), "loss": losses, "loss_ema": ema, "top1": top1,
           "top1_last50": float(np.mean(top1[-50:])),
           "ms_per_step_median": float(np.median(t_steps[5:]) * 1e3) if len(t_steps) > 5 else None,
           "max_memory_gb": torch.cuda.max_memory_allocated(dev) / 2 ** 30,
           "calls": {k: v for k, v in train_ops.CALLS.items() if v}}
    if epochs:
        rec.update({"segments": len(ds), "steps_per_epoch": per_epoch, "epoch_loss": epoch_loss, "stop_loss": stop_loss,
                    "lr_decay_epochs": decay_at,
                    "stopped_at_epoch": stopped_at, "epochs_run": len(epoch_loss)})
    for k in train_ops.CALLS:
        train_ops.CALLS[k] = 0
    if keep:
        return rec, model
    del model, opt, bat
    torch.cuda.empty_cache()
    return rec, None


def activation_maxima(mod, slow, fast):
    """max |value| of every BatchNorm output / residual-block output / stem output of an fp32 eval forward: the values the
    contract-grade encoder carries as fp16 plane pairs (csrc/split_planes.h clamps finite values at 65504)."""
    from avtex.slowfast import ResBlock, Stem

    peaks, hooks = {}, []
    for name, m in mod.named_modules():
        if isinstance(m, (torch.nn.BatchNorm3d, ResBlock, Stem)):
            def hook(_m, _i, o, name=name):
                o = o[0] if isinstance(o, (list, tuple)) else o
                peaks[name] = max(peaks.get(name, 0.0), float(o.detach().abs().max()))
            hooks.append(m.register_forward_hook(hook))
    with torch.no_grad():
        mod([slow, fast])
    for h in hooks:
        h.remove()
    return peaks


def roundtrip(model, video, args, dev, workdir):
    """checkpoint (reference keys) -> main.py -e --resume -> frames list; the contract on the trained weights."""
    import io

    import avtex
    from avtex import agreement, ops
    from avtex.fused_slowfast import SlowFastMFMA
    from avtex.main import cli, save_checkpoint
    from avtex.texture import TextureEngine

    out = {}
    model = model.to(memory_format=torch.contiguous_format).eval()
    os.makedirs(workdir, exist_ok=True)
    vdir = os.path.join(workdir, "videos")
    os.makedirs(vdir, exist_ok=True)
    np.savez(os.path.join(vdir, "clip.npz"), video=video.cpu().numpy(), fps=30.0)
    stem = os.path.join(workdir, "trained")
    save_checkpoint({"epoch": 1, "arch": "slowfast", "state_dict": model.state_dict(), "best_loss": 0.0}, True, stem)
    ckpt = stem + "_best.pth.tar"
    out["checkpoint_mb"] = os.path.getsize(ckpt) / 2 ** 20
    # (1) the CLI route: `main.py -e --resume CKPT` (reference README.md:44), aligned N x N mode, th 0.3
    cwd = os.getcwd()
    os.chdir(workdir)
    buf = io.StringIO()
    try:
        np.random.seed(11)
        with contextlib.redirect_stdout(buf):
            cli(["-vdata", vdir, "-vl", "clip", "-ea", "slowfast", "-m", "1", "-e", "-nintp", "-th", "0.3", "-temp", "0.1", "-mbs", "20",
                 "-nvl", "10", "--resume", ckpt, "--stitch_mode", "aligned", "--enc_batch", "64", "--logdir", os.path.join(workdir, "logs")])
    finally:
        os.chdir(cwd)
    text = buf.getvalue()
    assert "=> loaded checkpoint" in text and "Frames list: " in text, text[-2000:]
    frames = [int(x) for x in text.split("Frames list: ")[1].split("]")[0].strip(" [").split(",")]
    out["cli_frames"] = len(frames)
    out["cli_frames_head"] = frames[:24]
    # (2) the contract on these weights: f16x3 MFMA encoders vs the fp32 modules on the same frames
    W, S = 15, 6
    q_mod, t_mod = model.q_encoder.float().eval(), model.t_encoder.float().eval()

    def tables(qe, te, batch):
        eng = TextureEngine(qe, te, None, window=W, stride=S, temp=0.1, img_size=224, model_type=1, device=dev, enc_batch=batch)
        eng.set_video(video)
        qv, tv = eng.build_tables()
        torch.cuda.synchronize()
        return qv.clone(), tv.clone()

    q32, t32 = tables(q_mod, t_mod, 16)
    qx, tx = tables(SlowFastMFMA(q_mod, dev, precision="f16x3"), SlowFastMFMA(t_mod, dev, precision="f16x3"), 64)
    rep = agreement.compare_tables(qx, tx, q32, t32, 0.1, W, S)
    out["contract_f16x3_vs_fp32_modules"] = rep
    n = q32.shape[0]
    sim = agreement._build(qx, tx, 0.1)
    sel = ops.row_transition(sim, q_ids=torch.arange(n, device=dev, dtype=torch.int64), threshold=0.3, cap=n)
    out["windows"] = n
    out["survivor_fraction_th0.3"] = float(sel["cnt"].sum()) / (n * (n - 1.0))
    sel0 = ops.row_transition(sim, q_ids=torch.arange(n, device=dev, dtype=torch.int64), threshold=0.0, cap=n)
    out["survivors_per_row_th0.0"] = float(sel0["cnt"].sum()) / n
    # is the positive (the NEXT segment, what the InfoNCE labels train for: dataset.py:159-179) the row's best candidate?
    arg = sim.clone()
    arg[torch.arange(n), torch.arange(n)] = -1e30
    out["argmax_is_next_segment"] = float((arg[: n - 1].argmax(1) == torch.arange(1, n, device=dev)).float().mean())
    # (3) per-layer largest |activation| against the fp16 planes' clamp
    starts = np.linspace(0, n - 1, 16).astype(np.int64) * S
    slow, fast = ops.clip_pack(video.to(dev), starts, W, out_hw=224, dtype=torch.float32)
    peaks = {"q": activation_maxima(q_mod, slow, fast), "t": activation_maxima(t_mod, slow, fast)}
    worst = max((v, e + "." + k) for e in peaks for k, v in peaks[e].items())
    out["activation_peak"] = {"max_abs": worst[0], "layer": worst[1], "fp16_plane_clamp": 65504.0, "margin_x": 65504.0 / max(worst[0], 1e-30),
                              "top5": sorted(((round(v, 3), e + "." + k) for e in peaks for k, v in peaks[e].items()), reverse=True)[:5]}
    return out


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--lr", type=float, default=1e-2)
    ap.add_argument("--modes", nargs="+", default=["x3", "fp32"], choices=["x3", "fp32"])
    ap.add_argument("--init", default="bench", choices=["bench", "default"])
    ap.add_argument("--frames", type=int, default=1500)
    ap.add_argument("--frame-hw", type=int, default=128)
    ap.add_argument("--scene-len", type=int, default=24, help="frames per scene of the synthetic video (24 = bench.py's)")
    ap.add_argument("--epochs", type=int, default=0, help="epoch mode: at most this many epochs over a shuffled permutation of the segments "
                    "(batches of 8), stopping at the reference's rule (epoch loss < --stop-loss, main.py:475-477); 0 = --steps random batches")
    ap.add_argument("--stop-loss", type=float, default=0.07)
    ap.add_argument("--lr-decay-epochs", type=lambda v: [int(x) for x in v.split(",") if x], default=[],
                    help="epoch mode: epochs at which the learning rate is multiplied by 0.1 (the reference's StepLR, main.py:448-449)")
    ap.add_argument("--no-miopen-find", action="store_true", help="torch.backends.cudnn.benchmark off: the fp32 (MIOpen) mode without its "
                    "exhaustive solver search (minutes before the first step)")
    ap.add_argument("--roundtrip", action="store_true")
    ap.add_argument("--workdir", default="/tmp/avt_train_convergence")
    ap.add_argument("--against", default=None, help="a recorded result of this tool (same config): the largest EMA distance of each "
                    "run here to each recorded curve goes into the output as `against`")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "train_convergence.json"))
    args = ap.parse_args()

    from avtex import ops, synth

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    ops.device_check()
    torch.backends.cudnn.benchmark = not args.no_miopen_find
    video = synth.structured_video(123, args.frames, args.frame_hw, args.frame_hw, scene_len=args.scene_len, variety=1)
    res = {"config": {"steps": args.steps, "lr": args.lr, "init": args.init, "batch": 8, "negs": 14, "temp": 0.1, "img_size": 224,
                      "video": "synth.structured_video(123, %d, %d, %d, scene_len=%d, variety=1), fps 30 -> W 15, S 6" % (
                          args.frames, args.frame_hw, args.frame_hw, args.scene_len),
                      "log_1_plus_negs": float(np.log(15.0))}, "runs": {}}
    kept = None
    for k, mode in enumerate(args.modes):
        rec, model = train_run(mode, args, dev, video, keep=(args.roundtrip and k == 0))
        res["runs"][mode] = rec
        if model is not None:
            kept = model
    if len(res["runs"]) == 2:
        a, b = (res["runs"][m]["loss_ema"] for m in args.modes)
        n = min(len(a), len(b))
        res["ema_gap_max_after_20"] = float(max(abs(a[i] - b[i]) for i in range(min(20, n - 1), n)))
        res["ema_final"] = {m: res["runs"][m]["loss_ema"][-1] for m in args.modes}
    if args.against:
        with open(args.against) as f:
            old = json.load(f)
        same = all(old["config"].get(k) == res["config"].get(k) for k in ("lr", "init", "batch", "negs", "temp", "img_size", "video"))
        res["against"] = {"file": os.path.relpath(args.against, ROOT), "same_config": bool(same)}
        for m, r in res["runs"].items():
            for om, orun in old["runs"].items():
                a, b = r["loss_ema"], orun["loss_ema"]
                n = min(len(a), len(b))
                res["against"]["%s_vs_recorded_%s" % (m, om)] = {
                    "steps": n, "ema_gap_max_after_20": float(max(abs(a[i] - b[i]) for i in range(min(20, n - 1), n))),
                    "ema_last": [a[n - 1], b[n - 1]]}
    if kept is not None:
        res["roundtrip"] = roundtrip(kept, video, args, dev, args.workdir)
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    with open(args.out, "w") as f:
        json.dump(res, f, indent=1)
    brief = {m: {"first": r["loss"][0], "ema_last": r["loss_ema"][-1], "top1_last50": r["top1_last50"], "ms": r["ms_per_step_median"]}
             for m, r in res["runs"].items()}
    print(json.dumps({"brief": brief, "ema_gap_max_after_20": res.get("ema_gap_max_after_20"), "against": res.get("against"),
                      "roundtrip": {k: v for k, v in res.get("roundtrip", {}).items() if k not in ("cli_frames_head",)}}, default=str))


if __name__ == "__main__":
    main()
