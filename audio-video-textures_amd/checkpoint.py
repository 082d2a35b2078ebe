"""Kinetics-400 SlowFast checkpoint ingestion: the caffe2 pickle the reference starts config 5 from.

The reference builds its encoders through PySlowFast's `ActionPredictor` with `cfg.TEST.CHECKPOINT_TYPE = "caffe2"` and
`CHECKPOINT_FILE_PATH = .../pretrained/SLOWFAST_8x8_R50.pkl` (contrastive_video_textures/models/models.py:565-580): the
model-zoo file is a pickle `{"blobs": {caffe2 blob name: ndarray}}` (latin1), and PySlowFast renames the blobs to its module
names while loading.  `slowfast.SlowFast` here uses those module names, so this file only has to restate the renaming
(third-party behaviour, not vendored by the reference: PARITY UNPINNED like the architecture itself — no real .pkl exists in
this environment; tests/test_host_logic.py covers every parameter and buffer of `SlowFast` with a synthetic blob dict).

Blob grammar of SLOWFAST_8x8_R50 (no non-local blocks):
    [t_]conv1_w                                   stem convolution          (t_ = fast pathway)
    [t_]res_conv1_bn_{s,b,rm,riv}                 stem BatchNorm: scale, bias, running mean, running (inverse-)variance slot
    [t_]res<S>_<I>_branch1_w / _branch1_bn_*      stage S (2..5) block I shortcut
    [t_]res<S>_<I>_branch2{a,b,c}_w / _bn_*       bottleneck convolutions
    t_pool1_subsample_w / _bn_*                   lateral fusion after the stem          -> s1_fuse
    t_res<S>_<I>_branch2c_bn_subsample_w / _bn_*  lateral fusion after stage S (2..4)    -> s<S>_fuse
    pred_w, pred_b                                classifier (the reference replaces the head's projection by Identity: dropped)
    *_momentum, lr, model_iter, ...               solver state: dropped
"""
import pickle
import re

import numpy as np
import torch

_BN_FIELD = {"s": "weight", "b": "bias", "rm": "running_mean", "riv": "running_var"}
_BLOCK = re.compile(r"^(t_)?res([2-5])_(\d+)_branch(1|2[abc])_(w|bn_(?:s|b|rm|riv))$")
_STEM = re.compile(r"^(t_)?(?:conv1_(w)|res_conv1_bn_(s|b|rm|riv))$")
_FUSE = re.compile(r"^t_(?:pool1|res([2-4])_\d+_branch2c_bn)_subsample_(w|bn_(?:s|b|rm|riv))$")


def caffe2_blob_to_module_name(blob):
    """One caffe2 blob name -> this package's `SlowFast` state-dict key, or None for blobs that have no place in it
    (classifier, solver state)."""
    m = _FUSE.match(blob)  # (before the block pattern: the fusion blobs of stages 2-4 carry a block prefix)
    if m:
        stage, what = m.group(1) or "1", m.group(2)
        return "s%s_fuse.%s" % (stage, "conv_f2s.weight" if what == "w" else "bn." + _BN_FIELD[what[3:]])
    m = _STEM.match(blob)
    if m:
        stem = "s1.pathway%d_stem." % (1 if m.group(1) else 0)
        return stem + ("conv.weight" if m.group(2) else "bn." + _BN_FIELD[m.group(3)])
    m = _BLOCK.match(blob)
    if m:
        fast, stage, idx, branch, what = m.groups()
        base = "s%s.pathway%d_res%s." % (stage, 1 if fast else 0, idx)
        if branch == "1":
            return base + ("branch1.weight" if what == "w" else "branch1_bn." + _BN_FIELD[what[3:]])
        conv = branch[1]
        return base + "branch2." + (conv + ".weight" if what == "w" else conv + "_bn." + _BN_FIELD[what[3:]])
    return None


def convert_caffe2_slowfast(blobs):
    """{caffe2 blob: ndarray} -> (state dict for slowfast.SlowFast, sorted list of the blobs that were dropped)."""
    sd, dropped = {}, []
    for blob, value in blobs.items():
        key = caffe2_blob_to_module_name(blob)
        if key is None:
            dropped.append(blob)
            continue
        if key in sd:
            raise ValueError("caffe2 checkpoint: blobs collide on %s" % key)
        sd[key] = torch.from_numpy(np.ascontiguousarray(np.asarray(value, dtype=np.float32)))
    return sd, sorted(dropped)


def load_caffe2_pkl(path):
    with open(path, "rb") as f:
        data = pickle.load(f, encoding="latin1")
    return data["blobs"] if isinstance(data, dict) and "blobs" in data else data


def load_kinetics_slowfast(model, path, strict=True):
    """Loads SLOWFAST_8x8_R50.pkl (caffe2) — or a torch state dict (.pth / .pyth, optionally under "state_dict" /
    "model_state") — into a `slowfast.SlowFast`.  strict: every parameter and running statistic of the model must be present
    with the right shape (num_batches_tracked, which caffe2 does not have, excepted).  -> list of dropped blob / key names."""
    if str(path).endswith(".pkl"):
        sd, dropped = convert_caffe2_slowfast(load_caffe2_pkl(path))
    else:
        sd = torch.load(path, map_location="cpu")
        for k in ("state_dict", "model_state"):
            if isinstance(sd, dict) and k in sd:
                sd = sd[k]
        dropped = [k for k in sd if k.startswith("head.projection")]
        sd = {k: v for k, v in sd.items() if not k.startswith("head.projection")}
    own = model.state_dict()
    want = [k for k in own if not k.endswith("num_batches_tracked")]
    missing = [k for k in want if k not in sd]
    extra = [k for k in sd if k not in own]
    bad = [k for k in sd if k in own and tuple(sd[k].shape) != tuple(own[k].shape)]
    if bad:
        raise ValueError("checkpoint %s: shape mismatch on %s" % (path, ", ".join(
            "%s %s vs %s" % (k, tuple(sd[k].shape), tuple(own[k].shape)) for k in bad[:5])))
    if strict and (missing or extra):
        raise ValueError("checkpoint %s: %d missing (%s ...), %d unexpected (%s ...)" % (
            path, len(missing), ", ".join(missing[:3]), len(extra), ", ".join(extra[:3])))
    model.load_state_dict({k: v for k, v in sd.items() if k in own}, strict=False)
    return dropped
