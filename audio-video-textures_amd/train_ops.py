"""Training-step kernels on the MI355X, as torch.autograd.Functions the SlowFast modules call in train mode
when the model is in the training layout (channels_last_3d, main.py --train_layout ndhwc):

* `bn_act`: train-mode BatchNorm3d + shortcut add + ReLU as one HIP statistics pass + one apply pass in each direction
  (csrc/bn_train.hip) instead of MIOpenBatchNormFwdTrainSpatial / BwdSpatial + separate add / ReLU passes;
* `conv3d` / `conv3d_fork`: the convolutions' forward and stride-1 input gradient on the split-plane MFMA kernel
  (csrc/conv_x3.hip, fp32 in / fp32 out) and their weight gradient on csrc/wgrad_x3.hip; the two stems ([kt,7,7] on the
  3-channel clip) on the patch-resident kernels (csrc/stem_conv.hip forward with fp32 output, csrc/stem_train.hip weight
  gradient, both in the pixel-pair form); the strided layers' input gradients as one stride-1 convolution of dY per residue
  class of the input position (`_dgrad_strided`).  No MIOpen convolution is left in the step.

ARITHMETIC (main.py --train_conv, set_conv_mode): "x3" (the default on the MI355X) computes every product from two 16-bit
planes with fp32 accumulation — forward in fp16 planes (2^-22 per product), input and weight gradients in bf16 planes (2^-16
per product, fp32's exponent range; fp32 atomics in the weight gradient, so its summation order is not fixed) — NOT bit-for-bit
the reference's fp32 (train.py:114-141); "fp32" leaves every convolution to MIOpen's fp32 kernels (what the reference runs)
and keeps only the fused BatchNorm passes.  tests/test_gpu_train_step.py bounds the loss / gradient deviation of "x3" from an
fp64 step by the deviation of the stock fp32 step.
Weight planes are cached per tensor and rebuilt when `weight._version` changes (optimizer.step(), load_state_dict, any
in-place op on the parameter).  In-place updates through `.data` (EMA / momentum encoders, `p.data.clamp_()`) do NOT bump
that counter: call `invalidate_weight_cache()` after them (train.train() does after every optimizer step)."""
import contextlib
import ctypes as C
import weakref

import numpy as np
import torch
import torch.nn.functional as F

from . import _lib

# Which passes of the training step run on the hand-written kernels.  Module constants (no environment switches since round 4);
# `set_conv_mode` (main.py --train_conv) is the user-facing switch, tests set some of these to 0 for the stock-op reference path.
_FUSED = 1          # train-mode BatchNorm (+ shortcut + ReLU) on csrc/bn_train.hip
_CONV_X3 = 1        # convolution forward / stride-1 input gradient on conv_x3 (fp32 I/O)
_WGRAD_X3 = 1       # weight gradient on csrc/wgrad_x3.hip
_DGRAD_S_X3 = 1     # strided input gradients as residue-class convolutions on conv_x3 (0: MIOpen's bwd_data)
_STEM_WGRAD_X3 = 1  # the stems' weight gradient on the hand-written kernels
_PLANES_HIP = 1     # weight planes by csrc/stem_train.hip's two kernels (0: torch ops)
_STEM_PATCH = 1     # the stems on the patch-resident kernels (0: conv_x3 + wgrad slices)
_FORK = 1           # the shortcut's gradient summed in the a-convolution's input-gradient epilogue
_EPI_STATS = 1      # BatchNorm forward statistics on the producing convolution's epilogue (round 5: no statistics pass over its output)
_EPI_BWD = 1        # BatchNorm backward statistics on the epilogue of the consuming convolution's input-gradient launch (round 5)


def set_conv_mode(mode):
    """"x3": convolutions of the training step on the split-plane MFMA kernels (see the module docstring for the precision);
    "fp32": MIOpen's fp32 convolutions (the reference's arithmetic).  -> the mode now in force."""
    global _CONV_X3, _WGRAD_X3
    if mode not in ("x3", "fp32"):
        raise ValueError("train_conv must be 'x3' or 'fp32', got %r" % (mode,))
    _CONV_X3 = _WGRAD_X3 = 1 if mode == "x3" else 0
    return mode


def conv_mode():
    return "x3" if _CONV_X3 else "fp32"


def invalidate_weight_cache(drop=False):
    """Mark every cached set of weight planes stale (needed after in-place updates through `.data`, which `_version` does not see): the
    next lookup re-makes them — all of them in one launch (_refresh_planes).  drop=True forgets the entries themselves (tests that
    compare two ways of MAKING the planes)."""
    _PLANES_GEN[0] += 1
    if drop:
        _PLANES.clear()
        _PLANE_TABLES.clear()
    _STEM_IMAGES.clear()


# Optimizers that update their parameters without moving Tensor._version — torch's FUSED kernels (SGD / Adam(W) with fused=True:
# measured, _version 0 -> 0 over optimizer.step(), tools/experimental/debug_sgd_arena.py), updates through `.data` — would leave the
# version-keyed plane cache serving the OLD weights to every convolution after them.  Every optimizer step of the process therefore
# marks the cache stale (one global hook; the next lookup re-makes all planes in one launch, which a version bump would have asked for too).
try:
    from torch.optim.optimizer import register_optimizer_step_post_hook as _register_step_hook

    _STEP_HOOK = _register_step_hook(lambda *_a, **_k: invalidate_weight_cache())
except ImportError:  # (torch < 2.0: callers invalidate by hand, as train() always has)
    _STEP_HOOK = None


# per-process launch counters of the hand-written training kernels (tests assert that the default path really runs them)
CALLS = {"conv_fwd_x3": 0, "dgrad_x3": 0, "dgrad_strided_x3": 0, "wgrad_x3": 0, "wgrad_stem_x3": 0, "bn_fwd": 0, "bn_bwd": 0,
         "miopen_dgrad": 0, "miopen_wgrad": 0, "stem_fwd_patch": 0, "wgrad_stem_patch": 0, "maxpool_hip": 0, "pw_f32": 0,
         "bn_fwd_pre": 0, "bn_bwd_pre": 0, "dgrad_bwdstats": 0, "planes_multi": 0}


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


_RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", None)  # the current stream's handle without building a Stream object


def _stream():
    # (one call per launch, ~2900 a step at one item per rank — a host-bound step: the raw getter is ~1.5 us cheaper than
    #  torch.cuda.current_stream().cuda_stream)
    if _RAW_STREAM is not None:
        return C.c_void_p(_RAW_STREAM(torch.cuda.current_device()))
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _rows(x):
    """(m, c) of a [B,C,T,H,W] tensor whose memory is channels-last rows [B*T*H*W, C]."""
    return x.numel() // x.shape[1], x.shape[1]


_WS_BYTES = {}


def _workspace(m, c, groups, device):
    """Scratch for the per-workgroup partial sums and the per-channel coefficients (sized by the library)."""
    n = _WS_BYTES.get((m, c, groups))
    if n is None:
        n = _lib.lib().avt_bn_train_ws_bytes(m, c, groups)
        if n < 0:
            raise ValueError("bn_train: %d rows x %d channels in %d groups is outside the kernel's domain" % (m, c, groups))
        _WS_BYTES[(m, c, groups)] = n
    return torch.empty(n, dtype=torch.uint8, device=device)


# Per-replica BatchNorm in one launch: inside `with bn_replicas(n):` every fused BatchNorm treats its batch as n equal groups of
# consecutive samples, each normalised with its own statistics — the n items of a batch as the reference's DataParallel replicas
# see them (main.py:420: one item per GPU at batch 8 on 8 GPUs) — so that a rank's items go through the encoders as ONE batch
# (convolutions over 8 x 16 clips instead of 8 launches over 16: the deep layers have 8 x the tiles) with unchanged arithmetic.
# Running statistics and num_batches_tracked take GROUP 0's update only: DataParallel keeps the buffer updates of the replica on
# device 0 and drops the other replicas' (their buffers are per-forward broadcast copies), so a checkpoint trained here carries the
# eval-mode statistics the reference's would (one momentum step per forward, from item 0).
_BN_GROUPS = 1


class bn_replicas:
    def __init__(self, n):
        self.n = int(n)

    def __enter__(self):
        global _BN_GROUPS
        self.keep, _BN_GROUPS = _BN_GROUPS, self.n
        return self

    def __exit__(self, *exc):
        global _BN_GROUPS
        _BN_GROUPS = self.keep
        return False


def fusable(x, bn, res=None):
    c = x.shape[1]
    return (_FUSED and bn.training and x.is_cuda and x.dtype == torch.float32 and x.dim() == 5 and c >= 8 and (c & (c - 1)) == 0 and
            x.is_contiguous(memory_format=torch.channels_last_3d) and bn.weight is not None and bn.weight.dtype == torch.float32 and
            (res is None or (res.shape == x.shape and res.dtype == torch.float32 and
                             res.is_contiguous(memory_format=torch.channels_last_3d))))


# Concatenation without the copy (round 4).  SlowFast's lateral fusion is torch.cat([slow, lateral], 1): on channels-last tensors a
# strided copy of both operands forward and, backward, a .contiguous() of each gradient slice — 13 GB per config-5 step that no
# convolution needs.  Instead the two producers (the stage's last BatchNorm / the stem's max-pool, and the lateral's BatchNorm)
# write channel slices of ONE buffer: bn_act / max_pool_hw(cat_extra = e) allocate rows of c + e channels and return their first
# c as a view tagged `_avt_cat = (buffer, c)`; bn_act(cat_into = (buffer, offset)) fills the rest; join_channels hands out the
# buffer as the concatenation, and its backward hands each producer its slice of the gradient, which the kernels read through a
# leading dimension (avt_bn_train_bwd ld_dy).  Only the fused HIP paths set the tag: anything else keeps torch.cat.
_JOIN = 1  # (tests set 0: torch.cat and contiguous gradients, the same numbers)


def _cat_buffer(shape, c_total, device):
    b, _, t, h, w = shape
    return torch.empty((b, c_total, t, h, w), dtype=torch.float32, device=device, memory_format=torch.channels_last_3d)


def _alias(buf, c_off, c):
    """Channels c_off .. c_off + c of the buffer's rows as a tensor that SHARES the storage without being an autograd view of it
    (set_: a version counter of its own) — the producers fill the slices through raw pointers, and the view bookkeeping ("a view's
    base was modified in place") has nothing to say about tensors whose only writers are those kernels."""
    t = torch.empty(0, dtype=buf.dtype, device=buf.device)
    size = (buf.shape[0], c) + tuple(buf.shape[2:])
    return t.set_(buf.untyped_storage(), buf.storage_offset() + c_off, size, buf.stride())


def _row_ld(t):
    """Leading dimension of a [B,C,T,H,W] tensor that is channels-last ROWS with a constant row pitch (a channel slice of a
    channels-last tensor, or a contiguous one: pitch C), else None."""
    if t.dim() != 5 or t.dtype != torch.float32 or t.data_ptr() % 16:
        return None
    b, c, tt, h, w = t.shape
    st = t.stride()
    if c > 1 and st[1] != 1:
        return None
    ld = st[4] if w > 1 else (st[3] if h > 1 else (st[2] if tt > 1 else (st[0] if b > 1 else c)))
    want = (tt * h * w * ld, 1, h * w * ld, w * ld, ld)
    if any(n > 1 and s != e for n, s, e in zip(t.shape, st, want)) or ld < c or ld % 4:
        return None
    return ld


class _JoinChannels(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b, buf):
        ctx.c0 = a.shape[1]
        # INVARIANT: the slices and the joined tensor share storage OUTSIDE autograd's view tracking — their only
        # writers are the producing kernels, and nothing may modify them in place afterwards (an nn.ReLU(inplace=True) or `+=` on
        # one of them would corrupt the others' saved activations without autograd noticing).  The version counters are stamped
        # here and compared in backward, so that such an edit raises instead of training on garbage.
        out = _alias(buf, 0, buf.shape[1])  # the buffer both producers wrote, as a tensor of this node's own
        # (the joined tensor by weak reference: a strong one is the cycle out -> grad_fn -> ctx -> out, which only backward broke —
        #  a grad-enabled forward that is never backpropagated kept the whole buffer until the cyclic collector ran)
        ctx.stamp = ((a, a._version), (b, b._version))
        ctx.out_ref, ctx.out_version = weakref.ref(out), out._version
        return out

    @staticmethod
    def backward(ctx, d):
        out = ctx.out_ref()
        for t, v in ctx.stamp + (((out, ctx.out_version),) if out is not None else ()):
            if t._version != v:
                raise RuntimeError("train_ops.join_channels: a channel slice of the concatenation buffer (or the joined tensor) was "
                                   "modified in place after the join; the slices share storage outside autograd's view tracking")
        ctx.stamp = ctx.out_ref = None
        return d[:, : ctx.c0], d[:, ctx.c0 :], None


def join_channels(a, b):
    """torch.cat([a, b], 1) — without a copy when a and b are the two channel slices of one buffer (see above)."""
    ta, tb = getattr(a, "_avt_cat", None), getattr(b, "_avt_cat", None)
    if (ta is not None and tb is not None and ta[0] is tb[0] and ta[1] == 0 and tb[1] == a.shape[1] and
            ta[0].shape[1] == a.shape[1] + b.shape[1]):
        return _JoinChannels.apply(a, b, ta[0])
    return torch.cat([a, b], 1)


class _BnHandle:
    """What the input-gradient launch of the convolution that consumes a fused BatchNorm's output needs to know about it (the
    tensors are the ones the BatchNorm's own backward keeps: no extra memory).  Travels as `y._avt_bn` on the BatchNorm's output."""
    __slots__ = ("x", "mask", "weight", "bias", "mean", "invstd", "relu", "groups", "has_res", "c")

    def __init__(self, x, mask, weight, bias, mean, invstd, relu, groups, has_res):
        self.x, self.mask, self.weight, self.bias, self.mean, self.invstd = x, mask, weight, bias, mean, invstd
        self.relu, self.groups, self.has_res, self.c = bool(relu), int(groups), bool(has_res), x.shape[1]


_LAST_BN = None   # the handle of the last _BNAct.forward, for bn_act to tag its output with
_BWD_STATS = {}   # data_ptr of a masked output gradient -> (workspace, rows of partials, handle): written by the input-gradient
#                   launch (_conv_backward), consumed by that BatchNorm's backward a moment later; emptied at the next forward


class _BNAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, res, relu, momentum, eps, tracked=None, groups=1, cat=None, pre=None):
        m, c = _rows(x)
        if cat is None:
            y, ldy = torch.empty_like(x), 0  # keeps the channels-last strides
        else:  # (buffer, channel offset): the output is that slice of the concatenation buffer's rows
            y, ldy = _alias(cat[0], cat[1], c), cat[0].shape[1]
        save_mean = torch.empty(groups * c, dtype=torch.float32, device=x.device)
        save_invstd = torch.empty(groups * c, dtype=torch.float32, device=x.device)
        CALLS["bn_fwd"] += 1
        # ReLU with a shortcut: the mask leaves the apply pass as 4 bits per float4 chunk and is what the backward reads;
        # without a shortcut the backward recomputes it from x.  Either way the output is not read again, nor kept alive here
        mask = torch.empty(m * c // 4, dtype=torch.uint8, device=x.device) if (relu and res is not None) else None
        if pre is not None:  # (workspace, rows of partials per group): the producing convolution's epilogue has summed the rows
            ws, pre_rows = pre
            CALLS["bn_fwd_pre"] += 1
            _lib.check(_lib.lib().avt_bn_train_fwd_pre(_p(x), _p(res), _p(y), m, c, _p(weight), _p(bias), float(eps), float(momentum),
                                                       1 if relu else 0, groups, _p(ws), ws.numel(), _p(save_mean), _p(save_invstd),
                                                       _p(running_mean), _p(running_var), _p(tracked), _p(mask), ldy, int(pre_rows),
                                                       _stream()), "avt_bn_train_fwd_pre")
        else:
            ws = _workspace(m, c, groups, x.device)
            _lib.check(_lib.lib().avt_bn_train_fwd(_p(x), _p(res), _p(y), m, c, _p(weight), _p(bias), float(eps), float(momentum),
                                                   1 if relu else 0, groups, _p(ws), ws.numel(), _p(save_mean), _p(save_invstd),
                                                   _p(running_mean), _p(running_var), _p(tracked), _p(mask), ldy, _stream()),
                       "avt_bn_train_fwd")
        ctx.save_for_backward(x, mask, weight, bias, save_mean, save_invstd)
        ctx.has_res = res is not None
        ctx.relu = bool(relu)
        ctx.groups = groups
        global _LAST_BN
        if _BWD_STATS:
            _BWD_STATS.clear()  # (left-overs of a backward pass whose BatchNorm did not pick them up)
        ctx.handle = _LAST_BN = _BnHandle(x, mask, weight, bias, save_mean, save_invstd, relu, groups, res is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, mask, weight, bias, save_mean, save_invstd = ctx.saved_tensors
        m, c = _rows(x)
        pre = _BWD_STATS.pop(dy.data_ptr(), None) if _BWD_STATS else None
        # (autograd frees saved_tensors after backward but never ctx.__dict__ — the handle holds x and the mask, so it
        #  is dropped here, before anything else can keep the node alive through the next step's forward)
        handle, ctx.handle = ctx.handle, None
        # (valid only for the very tensor the launch wrote: autograd sums gradients that reach a tensor by several paths — into a
        #  new tensor, or IN PLACE into one of them, which the version counter shows; a summed gradient takes the plain passes
        #  below, where masking an already masked part again changes nothing)
        if (pre is not None and pre[2] is handle and dy._version == pre[3] and dy.shape == x.shape and dy.dtype == torch.float32 and
                dy.is_contiguous(memory_format=torch.channels_last_3d)):
            # dy is ALREADY g = mask * dz, and its producer (the input-gradient launch of the convolution that consumed this
            # BatchNorm's output) left the sums of g and g * xhat behind: finalize + apply; the shortcut's gradient is g itself
            dx = torch.empty_like(x)
            dgamma = torch.empty(c, dtype=torch.float32, device=x.device)
            dbeta = torch.empty(c, dtype=torch.float32, device=x.device)
            CALLS["bn_bwd"] += 1
            CALLS["bn_bwd_pre"] += 1
            _lib.check(_lib.lib().avt_bn_train_bwd_pre(_p(dy), _p(x), m, c, _p(weight), _p(save_mean), _p(save_invstd), ctx.groups,
                                                       _p(pre[0]), pre[0].numel(), int(pre[1]), _p(dx), _p(dgamma), _p(dbeta), _stream()),
                       "avt_bn_train_bwd_pre")
            return dx, dgamma, dbeta, None, None, (dy if ctx.has_res else None), None, None, None, None, None, None, None
        ld_dy = _row_ld(dy)  # a slice of a concatenation's gradient is read in place (rows with a pitch)
        if ld_dy is None:
            dy, ld_dy = dy.contiguous(memory_format=torch.channels_last_3d), c
        dx = torch.empty_like(x)
        dres = torch.empty_like(x) if ctx.has_res else None
        ws = _workspace(m, c, ctx.groups, x.device)
        dgamma = torch.empty(c, dtype=torch.float32, device=x.device)
        dbeta = torch.empty(c, dtype=torch.float32, device=x.device)
        CALLS["bn_bwd"] += 1
        _lib.check(_lib.lib().avt_bn_train_bwd(_p(dy), None, _p(x), m, c, _p(weight), _p(bias), _p(save_mean), _p(save_invstd),
                                               1 if ctx.relu else 0, ctx.groups, _p(mask), _p(ws), ws.numel(), _p(dx), _p(dres),
                                               _p(dgamma), _p(dbeta), ld_dy, _stream()), "avt_bn_train_bwd")
        return dx, dgamma, dbeta, None, None, dres, None, None, None, None, None, None, None


def bn_act(x, bn, res=None, relu=True, cat_extra=0, cat_into=None):
    """act(bn(x) [+ res]) for a BatchNorm3d module `bn`: the fused HIP pass in train mode on channels-last fp32 device tensors,
    the stock torch ops otherwise (eval mode, other layouts / dtypes, CPU) — same result, same running-statistics update.
    Inside `bn_replicas(n)` the batch is n groups with statistics of their own (the stock path: a loop over the groups).
    cat_extra = e / cat_into = (buffer, offset): the output is the first / a later channel slice of a concatenation buffer
    (join_channels; fused path only — the stock path returns a tensor of its own and the caller's join falls back to torch.cat)."""
    groups = _BN_GROUPS if (bn.training and _BN_GROUPS > 1) else 1
    if groups > 1 and x.shape[0] % groups:
        raise ValueError("bn_replicas(%d): a batch of %d samples does not split into the replicas" % (groups, x.shape[0]))
    if not fusable(x, bn, res):
        if groups > 1:  # per-replica statistics on the stock ops: one BatchNorm call per group; running statistics from group 0
            # only (DataParallel keeps the buffer updates of the replica on device 0 alone, reference main.py:420)
            xs = x.chunk(groups, 0)
            y = torch.cat([bn(xs[0])] + [F.batch_norm(xg, None, None, bn.weight, bn.bias, True, 0.0, bn.eps) for xg in xs[1:]], 0)
        else:
            y = bn(x)
        if res is not None:
            y = y + res
        return F.relu(y) if relu else y
    momentum = bn.momentum
    tracked = bn.num_batches_tracked if bn.track_running_stats else None
    if tracked is not None and momentum is None:  # cumulative moving average: the factor needs the count on the host
        if groups > 1:
            raise ValueError("bn_replicas: BatchNorm with momentum=None (cumulative average) is not supported in groups")
        tracked.add_(1)
        momentum = 1.0 / float(tracked)
        tracked = None
    # (otherwise num_batches_tracked is incremented by the statistics kernel itself: a launch per BatchNorm and pass less)
    rm, rv = (bn.running_mean, bn.running_var) if bn.track_running_stats else (None, None)
    cat = None
    if not _JOIN:
        pass
    elif (cat_into is not None and cat_into[0].shape[1] >= cat_into[1] + x.shape[1] and cat_into[0].shape[0] == x.shape[0] and
            cat_into[0].shape[2:] == x.shape[2:]):
        cat = cat_into
    elif cat_extra:
        cat = (_cat_buffer(x.shape, x.shape[1] + int(cat_extra), x.device), 0)
    # statistics the producing convolution left behind (conv3d(..., stats=bn)): valid for exactly this BatchNorm geometry
    pre = getattr(x, "_avt_stats", None)
    if pre is not None and (pre[2] != groups or pre[3] != x.shape[1]):
        pre = None
    y = _BNAct.apply(x, bn.weight, bn.bias, rm, rv, res, relu, 0.0 if momentum is None else momentum, bn.eps, tracked, groups, cat,
                     None if pre is None else pre[:2])
    if cat is not None:
        y._avt_cat = cat
    global _LAST_BN
    if _LAST_BN is not None and _EPI_BWD and cat is None:
        y._avt_bn = _LAST_BN  # (the convolution that consumes y hands it to its input-gradient launch: conv3d / conv3d_fork)
    _LAST_BN = None
    return y


# ------------------------------------------------------------------------------------------------------------------------
# One training step as a replayed HIP graph (round 6).  Config 5 on 8 GPUs gives every rank ONE item (16 clips) per step: ~3000
# launches whose device time is below the host's time to issue them — the host never waits for the device in that step
# (profiles/r06/train_host_bound_one_item.log: 65 ms per step, 0.05 s of device waits in 4 s).  Every launch of the step goes to
# torch's current stream (the C ABI takes the stream), autograd replays the backward on the forward's streams, the side streams
# fork from and join the capturing stream by events, and no tensor of the step is read on the host — so the whole device side of a
# step (sample + pack, forward, loss, backward, optimizer, weight planes) is capturable as it stands.
class GraphedStep:
    """step_fn() -> tensor (e.g. the loss): run `warmup` times eagerly on a side stream, then captured ONCE; every call replays the
    graph and returns the same (graph-owned) result tensor.  Inputs must be STATIC tensors that step_fn closes over (fill them with
    copy_ before each call); anything the host decides inside step_fn (branches, cache lookups, shapes) is frozen at capture."""

    def __init__(self, step_fn, device, warmup=3):
        self.graph, self.out = None, None
        # warm-up and capture run on ONE stream of this object: the step's side streams are chosen per step stream
        # (models._side_stream), and a capture stream that differs from the warm-up's was handed the fast pathway's side stream for
        # its query encoder — two of the graph's three branches on one stream: 65 ms per one-item step instead of 52
        # (profiles/r06/graph_stream_order.log)
        # The first warm-up steps run eagerly on the CALLER's stream, the last one on the capture stream: the step's side streams of
        # the caller's stream exist before the capture stream and its own side streams are created.  Measured, not understood
        # (profiles/r06/graph_stream_order.log, ..._fixed.log): with the capture stream as the process's FIRST stream the replayed
        # one-item step takes 76 ms, after an eager step on the default stream 53 ms — the graph's parallel branches land on other
        # hardware queues
        for _ in range(max(warmup - 1, 0)):
            step_fn()
        torch.cuda.synchronize(device)
        self.stream = torch.cuda.Stream(device=device)
        self.stream.wait_stream(torch.cuda.current_stream(device))
        with torch.cuda.stream(self.stream):
            if warmup > 0:
                step_fn()
        torch.cuda.current_stream(device).wait_stream(self.stream)
        torch.cuda.synchronize(device)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=self.stream):
            self.out = step_fn()
        self.graph = g

    def __call__(self):
        self.graph.replay()
        return self.out


# ------------------------------------------------------------------------------------------------------------------------
# Gradient accumulation over micro-batches (one optimizer step = several forward/backward passes: config 5's batch of 8 items
# on fewer than 8 GPUs, each item with its own BatchNorm statistics as a DataParallel replica has, main.py:420).  Left to
# autograd, every parameter's gradient is added into .grad by its own 5-microsecond launch per pass (640 per item) and every
# weight-gradient kernel is preceded by its own memset: 39 ms of a 540 ms step in launches that move no data to speak of
# (profiles/r03/train_step_kernels_before_accumulator.log).  Here a pass's gradients are taken from .grad with ONE
# multi-tensor add, and the weight-gradient kernels write into slices of one arena zeroed by ONE memset per pass.
_ARENA = None


class _GradArena:
    def __init__(self):
        self.buf, self.used, self.want = {}, {}, {}

    def reset(self):
        """Start of a pass: zero what the last pass used; grow to what it asked for."""
        for dev in list(self.buf) + [d for d in self.want if d not in self.buf]:
            want = self.want.get(dev, 0)
            if dev not in self.buf or self.buf[dev].numel() < want:
                self.buf[dev] = torch.zeros(int(want * 1.05) + 1024, dtype=torch.float32, device=dev)
            elif self.used.get(dev, 0):
                self.buf[dev][: self.used[dev]].zero_()
            self.used[dev], self.want[dev] = 0, 0

    def take(self, n, device):
        n4 = (n + 3) // 4 * 4  # 16-byte aligned slices
        self.want[device] = self.want.get(device, 0) + n4
        buf, off = self.buf.get(device), self.used.get(device, 0)
        if buf is None or off + n4 > buf.numel():
            return None  # (first pass, or a pass larger than the last: the caller allocates and zeroes its own)
        self.used[device] = off + n4
        return buf[off : off + n]


class MicroBatchGradients:
    """Sums the gradients of several forward/backward passes into the parameters' .grad with one multi-tensor add per pass.

        acc = MicroBatchGradients(model.parameters())
        acc.begin(len(items))                # instead of optimizer.zero_grad()
        for k, item in enumerate(items):
            if k == len(items) - 1: acc.before_last_backward()   # (needed under DistributedDataParallel: its all-reduce
            loss(item).backward()                                #  must see the sum — the last pass accumulates in .grad)
            if k < len(items) - 1: acc.after_backward()
        acc.finish(); optimizer.step()

    Same sum as autograd's own accumulation (fp32 adds in the same order per parameter)."""

    def __init__(self, params, single_pass_arena=False):
        self.params = [p for p in params if p.requires_grad]
        self.acc = None
        self.seen = False
        self.arena = _GradArena()
        # single_pass_arena (round 6): a step of ONE pass also takes its weight gradients as slices of the arena — one memset per step
        # instead of one per convolution (110 of a one-item config-5 step's launches).  The gradients then ARE arena slices until the
        # next begin() drops them: for loops that read .grad only between backward and the optimizer step, on one rank (DDP's buckets
        # want gradients that own their storage)
        self.single_pass_arena = bool(single_pass_arena)
        self.passes = 1

    def begin(self, passes):
        """passes: forward/backward passes of this optimizer step (1: nothing to accumulate; the arena stays off unless single_pass_arena)."""
        global _ARENA
        for p in self.params:
            p.grad = None
        live = [a for a in (self.acc or []) if a is not None]
        if live:
            torch._foreach_zero_(live)
        self.seen = False
        self.passes = passes
        on = passes > 1 or self.single_pass_arena
        _ARENA = self.arena if on else None
        if on:
            self.arena.reset()

    def after_backward(self):
        if self.acc is None:
            self.acc = [None] * len(self.params)
        src, dst = [], []
        for i, p in enumerate(self.params):
            if p.grad is None:
                continue
            if self.acc[i] is None:  # own storage (a gradient may be a slice of the arena, which the next pass zeroes)
                self.acc[i] = torch.zeros_like(p.grad)
            dst.append(self.acc[i])
            src.append(p.grad)
            p.grad = None
        if src:
            torch._foreach_add_(dst, src)
        self.seen = True
        self.arena.reset()

    def before_last_backward(self):
        """Hand the sum so far to autograd: the next backward accumulates into it (and DDP all-reduces the total)."""
        if self.seen:
            for p, a in zip(self.params, self.acc):
                if a is not None:
                    p.grad = a

    def finish(self):
        global _ARENA
        arena_on, _ARENA = _ARENA is not None, None
        if not arena_on or self.passes == 1:  # (a single pass: every gradient owns its storage — or, single_pass_arena, its slice until begin())
            return
        # the gradients the last pass produced as arena slices live on in .grad only where autograd accumulated into the
        # accumulator's own tensors; a parameter first seen in the last pass keeps an arena slice: give it its own storage
        if self.acc is None:
            self.acc = [None] * len(self.params)
        for i, p in enumerate(self.params):
            if p.grad is not None and self.acc[i] is None:
                p.grad = p.grad.clone()


# ------------------------------------------------------------------------------------------------------------------------
# Convolutions of the training step on the split-plane MFMA kernel (csrc/conv_x3.hip, IO32 form): the forward of every
# Conv3d(bias=False) with channel counts in multiples of 8, and the input gradient of the stride-1 ones (a convolution of
# dy with the flipped, transposed filter).  fp32 tensors in and out; 2^-22 (forward, fp16 planes) / 2^-16 (dgrad, bf16
# planes: gradients need fp32's exponent range) per product instead of the fp32 MFMA's rate of 1/16 of the bf16 pipe.
# The weight gradient is csrc/wgrad_x3.hip (bf16 planes, transposing LDS stage, split over positions with fp32 atomics);
# the strided layers' input gradients are `_dgrad_strided` below; the stems have kernels of their own (`_StemX3`).
_TABS, _PLANES = {}, {}
_PLANES_MULTI = 1      # stale planes are re-made by ONE launch over every cached weight (0: one by one, as rounds 3-5 did; tests)
_PLANES_GEN = [0]      # invalidate_weight_cache() bumps it
_PLANE_TABLES = {}     # device -> (key, job table, blk2job, blocks) of the last multi-job launch
_PLANE_TABLES_CAPTURED = []  # tables a stream capture has launched with (kept alive for the graphs that replay them)
_PLANE_REFRESH = {}    # device -> {"id", "event", "stream", "waited": {stream handle: id}}


class _PlaneEntry:
    """A cached set of weight planes: `value` is what the lookup returns; `jobs` describes the launches that made its tensors
    (None: not re-makeable in place — torch-assembled forms), as dicts {"off": byte offset of the source in the owner's storage,
    "fields": the AvtPlaneJob fields after `w` up to `blk0`, "sel": taps, "blocks": grid size}."""
    __slots__ = ("ref", "version", "gen", "value", "jobs", "sig")

    def __init__(self, owner, value, jobs=None):
        self.ref, self.version, self.gen, self.value, self.jobs = weakref.ref(owner), owner._version, _PLANES_GEN[0], value, jobs
        self.sig = (tuple(owner.shape), tuple(owner.stride()), owner.dtype)  # what the jobs' sizes and offsets were derived from

    def stale(self, owner):
        return self.version != owner._version or self.gen != _PLANES_GEN[0]


def _job_rows(w, owner, hi, lo, ws, f16, rows, k, idx_map=None):
    kind = 0 if idx_map is None else 2
    return {"off": w.data_ptr() - owner.data_ptr(), "blocks": int(rows),
            "fields": (hi.data_ptr(), lo.data_ptr(), ws.data_ptr() if ws is not None else 0, idx_map.data_ptr() if idx_map is not None else 0,
                       kind, int(bool(f16)), int(rows), int(k), 0, 0, 0, 0, 0, 0),
            "sel": (), "keep": (hi, lo, ws, idx_map)}


def _job_transposed(w, owner, hi, lo, cout, taps, cin, sel):
    gx, gy = (cin + 31) // 32, (cout + 31) // 32
    return {"off": w.data_ptr() - owner.data_ptr(), "blocks": gx * gy * len(sel),
            "fields": (hi.data_ptr(), lo.data_ptr(), 0, 0, 1, 0, 0, 0, int(cout), int(taps), int(cin), len(sel), gx, gy),
            "sel": tuple(int(v) for v in sel), "keep": (hi, lo)}


def _refresh_planes(device):
    """Re-make EVERY stale cached plane set of `device` whose recipe is known, in place, with one launch (ops.weight_planes_multi);
    -> False when there was nothing to do.  Other streams order themselves behind the launch in _plane_lookup (one event)."""
    import struct

    from . import ops
    todo = []
    for ent in list(_PLANES.values()):
        owner = ent.ref()
        if owner is not None and ent.jobs and owner.device == device and ent.stale(owner):
            if ent.sig != (tuple(owner.shape), tuple(owner.stride()), owner.dtype):
                ent.jobs = None  # (`param.data = other layout`: the recipe is void; the lookup misses and the planes are made anew)
                continue
            todo.append((ent, owner))
    if not todo:
        return False
    recs = tuple((owner.data_ptr() + j["off"],) + j["fields"] + j["sel"] for ent, owner in todo for j in ent.jobs)
    tab = _PLANE_TABLES.get(device)
    if tab is None or tab[0] != recs:  # (the same weights at the same addresses every step: built once; a HIP graph captures its pointers)
        if torch.cuda.is_current_stream_capturing():
            return False  # (a capture whose warm-up did not see this set of weights: no host-to-device copy inside it — one by one, as before)
        raw, blk, b0 = [], [], 0
        for n, (ent, owner) in enumerate(todo):
            for j in ent.jobs:
                sel = j["sel"] + (0,) * (32 - len(j["sel"]))
                raw.append(struct.pack("<5Q12i32i", owner.data_ptr() + j["off"], *j["fields"][:4], *j["fields"][4:], b0, 0, *sel))
                blk.append(np.full(j["blocks"], len(raw) - 1, dtype=np.int32))
                b0 += j["blocks"]
        raw = b"".join(raw)
        assert len(raw) == len(blk) * ops.weight_planes_job_bytes(), "AvtPlaneJob layout"
        tab = (recs, torch.from_numpy(np.frombuffer(raw, dtype=np.uint8).copy()).to(device), torch.from_numpy(np.concatenate(blk)).to(device), b0)
        _PLANE_TABLES[device] = tab
    if torch.cuda.is_current_stream_capturing() and not any(t is tab for t in _PLANE_TABLES_CAPTURED):
        _PLANE_TABLES_CAPTURED.append(tab)  # a HIP graph holds this table's address: it outlives every later rebuild
    with torch.no_grad():
        ops.weight_planes_multi(tab[1], tab[2], tab[3])
    for ent, owner in todo:
        ent.version, ent.gen = owner._version, _PLANES_GEN[0]
    cur = torch.cuda.current_stream(device)
    r = _PLANE_REFRESH.setdefault(device, {"id": 0, "waited": {}})
    r["id"] += 1
    r["event"] = torch.cuda.Event()
    r["event"].record(cur)
    r["stream"] = cur
    r["waited"] = {}
    CALLS["planes_multi"] += 1
    return True


def _plane_lookup(key, owner):
    """The cached planes under `key` if they are `owner`'s and current — made current, together with every other stale entry, by one
    launch when only the optimizer step lies between (same tensors, new values) — else None."""
    ent = _PLANES.get(key)
    if ent is None or ent.ref() is not owner:
        return None
    if ent.stale(owner):
        if not (_PLANES_MULTI and ent.jobs and owner.is_cuda and _refresh_planes(owner.device)) or ent.stale(owner):
            return None
    r = _PLANE_REFRESH.get(owner.device) if owner.is_cuda else None
    if r is not None and r.get("event") is not None and ent.jobs:  # the launch that wrote these planes ran on another stream: behind it, once
        raw = _RAW_STREAM(owner.device.index) if _RAW_STREAM is not None else torch.cuda.current_stream(owner.device).cuda_stream
        if raw != r["stream"].cuda_stream and r["waited"].get(raw) != r["id"]:
            torch.cuda.current_stream(owner.device).wait_event(r["event"])
            r["waited"][raw] = r["id"]
    return ent.value


def _plane_store(key, owner, value, jobs=None):
    if len(_PLANES) > 4096:  # entries of models that are gone
        for k in [k for k, v in _PLANES.items() if v.ref() is None]:
            del _PLANES[k]
    _PLANES[key] = _PlaneEntry(owner, value, jobs)


def _ktab(cin, kernel, h, w, ldi, device):
    from . import ops
    key = (cin, tuple(kernel), h, w, ldi, str(device))
    tab = _TABS.get(key)
    if tab is None:
        tab = torch.from_numpy(ops.conv3d_ktab(cin, kernel, h, w, ldi)).to(device)
        _TABS[key] = tab
    return tab


def _weight_planes(weight, transposed):
    """(hi, lo, wscale) planes of a Conv3d weight in the kernel's [Cout, taps * Cin] order, cached until the optimizer
    changes the tensor.  transposed: the dgrad filter W'[ci, co, flipped taps], as bf16 planes."""
    from . import ops
    from .fused_slowfast import split_planes
    # keyed on the tensor that OWNS the storage: conv2d hands over `conv.weight.unsqueeze(2)`, a fresh view per call, which never
    # hit the cache and left an entry per call behind; a view shares its base's version counter
    owner = weight._base if weight._base is not None else weight
    key = (id(owner), bool(transposed), tuple(weight.shape))
    # (the entry remembers WHICH tensor it was made from: an id — like an address — can be reused after the first one died)
    hit = _plane_lookup(key, owner)
    if hit is not None:
        return hit
    jobs = None
    with torch.no_grad():
        w = weight.detach()
        fast = (_PLANES_HIP and w.dtype == torch.float32 and w.dim() == 5 and w.shape[1] % 8 == 0 and w.shape[0] % 2 == 0 and
                w.is_contiguous(memory_format=torch.channels_last_3d) and w.shape[2] * w.shape[3] * w.shape[4] <= 32)
        if fast:  # two launches per convolution and step instead of ~35 torch ops (the memory of a channels-last weight is
            #       already [cout][taps][cin], the kernels' K order)
            cout, cin = w.shape[0], w.shape[1]
            taps = w.shape[2] * w.shape[3] * w.shape[4]
            rows = w.permute(0, 2, 3, 4, 1).reshape(cout, taps * cin)  # a view
            if transposed:
                sel = list(range(taps - 1, -1, -1))
                planes = ops.weight_planes_t_f32(rows.view(cout, taps, cin), sel) + (None,)
                jobs = [_job_transposed(rows, owner, planes[0], planes[1], cout, taps, cin, sel)]
            else:
                planes = ops.weight_planes_f32(rows, ops.X3_F16)
                jobs = [_job_rows(rows, owner, planes[0], planes[1], planes[2], True, cout, taps * cin)]
        else:
            w = w.float()
            if w.shape[1] % 8:  # the stems' 3 input channels, padded with zero taps
                w = torch.cat([w, w.new_zeros((w.shape[0], 8 - w.shape[1] % 8) + tuple(w.shape[2:]))], 1)
            if transposed:
                w = w.flip(2, 3, 4).transpose(0, 1)
            wt = w.permute(0, 2, 3, 4, 1).reshape(w.shape[0], -1)
            if transposed:
                hi, lo = split_planes(wt, ops.X3_BF16)
                planes = (hi, lo, None)
            else:  # fp16 planes: rows scaled by a power of two into [2^9, 2^10), undone on the accumulator (fused_slowfast.FusedConv)
                mx = wt.abs().amax(dim=1).clamp_min(1e-30)
                sc = torch.pow(2.0, 9.0 - torch.floor(torch.log2(mx)))
                hi, lo = split_planes(wt * sc.view(-1, 1), ops.X3_F16)
                planes = (hi, lo, (1.0 / sc).float().contiguous())
    _plane_store(key, owner, planes, jobs)
    return planes


# ---- few-channel layers: g adjacent pixels along W as g * C channels (round 4; the inference path's group_weights_w in the training
# step).  The fast pathway's 8-channel layers ran their forward / input gradient at 0.8-2.3 TB/s on the general tile (rows of 32
# bytes, a 32-wide MFMA tile that is 3/4 padding); the same bytes viewed as [.., W / g, g * C] rows with block-Toeplitz weights along W
# give 4 x fewer, 4 x fuller rows.  The weight gradient keeps the plain form (grouping would multiply its flops by g * g).
_GROUP = 1
_PW_F32 = 1         # pointwise layers (forward / stride-1 input gradient) on the streaming fp32-in / fp32-out kernel (csrc/pw_x3.hip)
_GROUP_IDX = {}


def _group_factor(cin, cout, kernel, stride, pad, w):
    if _PW_F32 and tuple(kernel) == (1, 1, 1) and tuple(stride) == (1, 1, 1) and tuple(pad) == (0, 0, 0):
        from . import ops
        if ops.pw_x3_f32_supported(cin, cout):
            return 1  # (a pointwise layer the streaming kernel takes as it is: _conv_x3_rows)
    if not _GROUP or tuple(stride) != (1, 1, 1) or pad[2] != kernel[2] // 2 or kernel[2] % 2 == 0 or cin % 8 or cout % 8:
        return 1
    small = min(cin, cout)
    g = 4 if small <= 16 else (2 if small <= 32 else 1)
    if kernel[2] == 1 and kernel[1] == 1 and kernel[0] > 1:  # temporal taps only: grouping just fills the tile; stop at 32 outputs
        while g > 1 and g * cout > 32:
            g //= 2
    while g > 1 and w % g:
        g //= 2
    return g


_GROUP_MAPS = {}  # (shape, strides, g, transposed, device) -> int32 [rows, cols] storage offsets of the grouped rows (-1: a zero tap)
_GROUP_GATHER = 1  # the grouped planes by ONE gather launch per weight (0: the torch assembly of round 4, for tests)


def _grouped_rows(w, g, transposed):
    """The pixel-grouped row form of a [cout, cin, kt, kh, kw] tensor (the weight itself, or — round 6 — the tensor of its storage
    offsets): rows [(po, n)][(dt, dh, dg)][(pi, c)], block-Toeplitz along W over groups of g pixels; `pad` fills the taps no original
    tap meets.  -> rows [g * co, kt * kh * kwg * g * ci], (kt, kh, kwg), rg."""
    pad = 0 if w.is_floating_point() else -1
    if transposed:
        w = w.flip(2, 3, 4).transpose(0, 1)  # the input gradient as a convolution of dY: filter [cin][cout][flipped taps]
    co, ci, kt, kh, kw = w.shape
    r = kw // 2
    rg = -(-r // g)
    kwg = 1 + 2 * rg
    ikey = (g, kw, str(w.device))
    idx = _GROUP_IDX.get(ikey)
    if idx is None:  # tap of the ORIGINAL filter that (output pixel po, input pixel pi, group tap dg) meets; kw = the zero tap
        po, pi, dg = torch.meshgrid(torch.arange(g), torch.arange(g), torch.arange(kwg), indexing="ij")
        dw = g * (dg - rg) + pi - po + r
        idx = torch.where((dw >= 0) & (dw < kw), dw, torch.full_like(dw, kw)).to(w.device)
        _GROUP_IDX[ikey] = idx
    w_ext = torch.cat([w, w.new_full((co, ci, kt, kh, 1), pad)], 4)
    wg = w_ext[..., idx]  # [co, ci, kt, kh, po, pi, dg]
    rows = wg.permute(4, 0, 2, 3, 6, 5, 1).reshape(g * co, kt * kh * kwg * g * ci).contiguous()
    return rows, (kt, kh, kwg), rg


def _grouped_planes(weight, g, transposed, plane_dtype):
    """Planes of the pixel-grouped form of a Conv3d weight (transposed: of the input gradient's filter W'[ci][flipped taps][co])
    -> (hi, lo, wscale | None), (kt, kh, kwg), rg.  The grouped rows are a fixed GATHER of the weight's elements: the map of storage
    offsets is built once per (shape, layout, g, direction) by running the assembly on the offsets themselves, and every step is one
    launch (ops.weight_planes_gather_f32) instead of a flip, a cat, an index, a permute and a copy in front of the split."""
    from . import ops
    key = (id(weight), "group", g, bool(transposed))
    hit = _plane_lookup(key, weight)
    if hit is not None:
        return hit
    jobs = None
    with torch.no_grad():
        w = weight.detach()
        dense = w.dtype == torch.float32 and (w.is_contiguous() or w.is_contiguous(memory_format=torch.channels_last_3d))
        if _GROUP_GATHER and dense and w.is_cuda and w.numel() < (1 << 31):
            mkey = (tuple(w.shape), tuple(w.stride()), g, bool(transposed), str(w.device))
            ent = _GROUP_MAPS.get(mkey)
            if ent is None:
                offs = torch.arange(w.numel(), dtype=torch.int32).as_strided(tuple(w.shape), tuple(w.stride()))  # element -> storage offset
                rows, kern, rg = _grouped_rows(offs, g, transposed)
                ent = (rows.to(torch.int32).contiguous().to(w.device), kern, rg)
                _GROUP_MAPS[mkey] = ent
            pl = ops.weight_planes_gather_f32(w, ent[0], plane_dtype)
            planes = (pl, ent[1], ent[2])
            jobs = [_job_rows(w, weight, pl[0], pl[1], pl[2], plane_dtype == ops.X3_F16, ent[0].shape[0], ent[0].shape[1], idx_map=ent[0])]
        else:
            rows, kern, rg = _grouped_rows(w.float(), g, transposed)
            planes = (ops.weight_planes_f32(rows, plane_dtype), kern, rg)
    _plane_store(key, weight, planes, jobs)
    return planes


_STAT_GROUP_COLS = 128  # widest pixel-grouped output (g * cout columns) whose BatchNorm statistics ride on the epilogue: the copies of
#                         a channel must share one N tile (csrc/conv_args.h stat_fold_store; conv_x3.hip rejects wider forms)
_LAST_STATS = None  # (output data_ptr, workspace, rows of partials per group, groups, channels) of the last convolution that left
#                     BatchNorm statistics behind: handed from inside the autograd Function to conv3d(), which tags the output


def _bwd_bn_args(h):
    return (h.x, h.mean, h.invstd, h.weight, None if h.mask is not None else h.bias, h.mask, h.relu)


def _conv_x3_rows(x, planes, plane_dtype, cin, cout, kernel, stride, pad, add=None, group=None, stats=0, bwd_bn=None):
    """x [B, Cin, T, H, W] channels-last fp32 -> [B, Cout, To, Ho, Wo] channels-last fp32 (+ add, same shape, in the epilogue).
    stats = n > 0: the output feeds a train-mode BatchNorm of n replica groups — the kernel's epilogue leaves its statistics'
    partial sums (ops.conv3d_igemm_x3_f32_stats / ops.pw_x3_f32_stats; recorded in _LAST_STATS)."""
    global _LAST_STATS
    from . import ops
    b, _, t, h, w = x.shape
    od = [(n + 2 * p - k) // s_ + 1 for n, p, k, s_ in zip((t, h, w), pad, kernel, stride)]
    y = torch.empty((b, cout, od[0], od[1], od[2]), dtype=torch.float32, device=x.device, memory_format=torch.channels_last_3d)
    stats = int(stats) if (stats and add is None and (cout & (cout - 1)) == 0 and cout >= 8 and b % int(stats) == 0) else 0
    if bwd_bn is not None:
        # an input gradient that is the output gradient of the BatchNorm `bwd_bn` (a _BnHandle): masked on the way out, that
        # BatchNorm's backward statistics summed in the epilogue (ops.conv3d_igemm_x3_f32_bwdstats / ops.pw_x3_f32_bwdstats; the
        # 256 x 256 tile does not: st is None, the plain launch below runs)
        st = None
        addr = None if add is None else add.permute(0, 2, 3, 4, 1)
        if group is not None:
            g, gk, rg = group
            # (pixel-grouped statistics fold a channel's pixel copies inside ONE N tile of <= 128 columns: wider forms keep the pass)
            st = None if g * cout > _STAT_GROUP_COLS else ops.conv3d_igemm_x3_f32_bwdstats(x.permute(0, 2, 3, 4, 1), planes[0], planes[1], y.permute(0, 2, 3, 4, 1),
                                                  _ktab(g * cin, gk, h, w // g, g * cin, x.device), (b, t, h, w // g), g * cin, g * cout, gk,
                                                  (pad[0], pad[1], rg), plane_dtype, _bwd_bn_args(bwd_bn), bwd_bn.groups, cout, add=addr)
        elif (_PW_F32 and tuple(kernel) == (1, 1, 1) and tuple(pad) == (0, 0, 0) and x.numel() // cin < (1 << 31) - 16 and
              ops.pw_x3_f32_supported(cin, cout)):
            st = ops.pw_x3_f32_bwdstats(x.permute(0, 2, 3, 4, 1), cin, planes[0], planes[1], y.permute(0, 2, 3, 4, 1), cout, plane_dtype,
                                        _bwd_bn_args(bwd_bn), bwd_bn.groups, add=addr)
            if st is not None:
                CALLS["pw_f32"] += 1
        else:
            st = ops.conv3d_igemm_x3_f32_bwdstats(x.permute(0, 2, 3, 4, 1), planes[0], planes[1], y.permute(0, 2, 3, 4, 1),
                                                  _ktab(cin, kernel, h, w, cin, x.device), (b, t, h, w), cin, cout, kernel, pad, plane_dtype,
                                                  _bwd_bn_args(bwd_bn), bwd_bn.groups, cout, add=addr)
        if st is not None:
            CALLS["dgrad_bwdstats"] += 1
            _BWD_STATS[y.data_ptr()] = (st[0], st[1], bwd_bn, y._version)
            return y
    if group is not None:  # (g, grouped kernel, rg): the same memory as [.., w / g, g * C] rows, block-Toeplitz planes
        g, gk, rg = group
        if stats and g * cout <= _STAT_GROUP_COLS:
            ws, pre_rows = ops.conv3d_igemm_x3_f32_stats(x.permute(0, 2, 3, 4, 1), planes[0], planes[1], planes[2], y.permute(0, 2, 3, 4, 1),
                                                         _ktab(g * cin, gk, h, w // g, g * cin, x.device), (b, t, h, w // g), g * cin, g * cout,
                                                         gk, (1, 1, 1), (pad[0], pad[1], rg), g * cin, g * cout, plane_dtype, stats, cout)
            _LAST_STATS = (y.data_ptr(), ws, pre_rows, stats, cout)
            return y
        ops.conv3d_igemm_x3_f32(x.permute(0, 2, 3, 4, 1), planes[0], planes[1], planes[2], y.permute(0, 2, 3, 4, 1),
                                _ktab(g * cin, gk, h, w // g, g * cin, x.device), (b, t, h, w // g), g * cin, g * cout, gk, (1, 1, 1),
                                (pad[0], pad[1], rg), g * cin, g * cout, plane_dtype, add=None if add is None else add.permute(0, 2, 3, 4, 1))
        return y
    if (_PW_F32 and tuple(kernel) == (1, 1, 1) and tuple(stride) == (1, 1, 1) and tuple(pad) == (0, 0, 0) and x.numel() // cin < (1 << 31) - 16 and
            ops.pw_x3_f32_supported(cin, cout)):
        # the pointwise layers with few K-steps and many outputs (the bottlenecks' expanding convolutions, the input gradients of the
        # reducing ones) on the streaming kernel: the 128 x 128 tile spent 55 % of a workgroup in its prologue + epilogue on them
        CALLS["pw_f32"] += 1
        if stats:
            st = ops.pw_x3_f32_stats(x.permute(0, 2, 3, 4, 1), cin, planes[0], planes[1], planes[2], y.permute(0, 2, 3, 4, 1), cout,
                                     plane_dtype, stats)
            if st is not None:
                _LAST_STATS = (y.data_ptr(), st[0], st[1], stats, cout)
                return y
        ops.pw_x3_f32(x.permute(0, 2, 3, 4, 1), cin, planes[0], planes[1], planes[2], y.permute(0, 2, 3, 4, 1), cout, plane_dtype,
                      add=None if add is None else add.permute(0, 2, 3, 4, 1))
        return y
    if stats:
        ws, pre_rows = ops.conv3d_igemm_x3_f32_stats(x.permute(0, 2, 3, 4, 1), planes[0], planes[1], planes[2], y.permute(0, 2, 3, 4, 1),
                                                     _ktab(cin, kernel, h, w, cin, x.device), (b, t, h, w), cin, cout, kernel, stride, pad,
                                                     cin, cout, plane_dtype, stats, cout)
        _LAST_STATS = (y.data_ptr(), ws, pre_rows, stats, cout)
        return y
    ops.conv3d_igemm_x3_f32(x.permute(0, 2, 3, 4, 1), planes[0], planes[1], planes[2], y.permute(0, 2, 3, 4, 1),
                            _ktab(cin, kernel, h, w, cin, x.device), (b, t, h, w), cin, cout, kernel, stride, pad, cin, cout, plane_dtype,
                            add=None if add is None else add.permute(0, 2, 3, 4, 1))
    return y


def _tag_stats(y):
    """The statistics the convolution that produced y left behind (if it did) become an attribute of y: bn_act picks them up."""
    global _LAST_STATS
    st, _LAST_STATS = _LAST_STATS, None
    if st is not None and st[0] == y.data_ptr():
        y._avt_stats = st[1:]
    return y


class _ConvX3(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, stride, padding, stats=0, bn_in=None):
        from . import ops
        cout, cin = weight.shape[0], weight.shape[1]
        kernel = tuple(weight.shape[2:])
        ctx.conf = (stride, padding, kernel, cin, cout)
        ctx.bn_in = bn_in  # the fused BatchNorm whose output x is (a _BnHandle), or None: see _conv_backward
        xin = x
        if cin % 8:  # stem: [B, 3, T, H, W] -> 8 channels-last channels, the last 5 zero (weights padded to match)
            xin = torch.empty((x.shape[0], 8) + tuple(x.shape[2:]), dtype=x.dtype, device=x.device, memory_format=torch.channels_last_3d)
            xin[:, :cin] = x
            xin[:, cin:] = 0
        # (the stems keep the padded channels-last copy: their weight gradient reads it 4 channels at a time)
        ctx.save_for_backward(xin if (cin % 8 and (_STEM_WGRAD_X3 or cin != 3)) else x, weight)
        CALLS["conv_fwd_x3"] += 1
        g = _group_factor(xin.shape[1], cout, kernel, stride, padding, xin.shape[4]) if xin is x else 1
        if g > 1:
            planes, gk, rg = _grouped_planes(weight, g, False, ops.X3_F16)
            return _conv_x3_rows(xin, planes, ops.X3_F16, cin, cout, kernel, stride, padding, group=(g, gk, rg), stats=stats)
        return _conv_x3_rows(xin, _weight_planes(weight, False), ops.X3_F16, xin.shape[1], cout, kernel, stride, padding, stats=stats)

    @staticmethod
    def backward(ctx, dy):
        return _conv_backward(ctx, dy, None) + (None, None, None, None)


def _conv_backward(ctx, dy, dalias):
    """(dx, dw) of the convolution; dalias = a gradient that reached the input by another path (conv3d_fork), summed into dx
    in the stride-1 kernel's epilogue instead of by a separate pass.  (Weight gradients on a side stream of their own — nothing in the
    backward waits for a dW — were measured three times and not kept: round 5 at 8 items per rank, -4 ... -11 %; round 6 at one item
    per rank, eager 234 -> 227 clips/s, inside the replayed HIP graph 303 -> 280: profiles/r06/train_wgrad_side_stream_ab_not_kept.log)"""
    from . import ops
    x, weight = ctx.saved_tensors
    stride, padding, kernel, cin, cout = ctx.conf
    dy = dy.contiguous(memory_format=torch.channels_last_3d)
    dw = _conv_wgrad(ctx, dy, x, weight, stride, padding, kernel, cin, cout) if ctx.needs_input_grad[1] else None
    return _conv_dgrad(ctx, dy, dalias, x, weight, stride, padding, kernel, cin, cout), dw


def _conv_wgrad(ctx, dy, x, weight, stride, padding, kernel, cin, cout):
    """dW of the convolution (split-plane weight-gradient kernels; MIOpen for the shapes they do not cover)."""
    from . import ops
    dw = None
    taps = kernel[0] * kernel[1] * kernel[2]
    if (_WGRAD_X3 and cin % 8 == 0 and taps <= 28 and max(x.numel(), dy.numel()) < (1 << 31) - 64 and
            weight.is_contiguous(memory_format=torch.channels_last_3d)):
        CALLS["wgrad_x3"] += 1
        dims = (x.shape[0], x.shape[2], x.shape[3], x.shape[4])
        flat = _ARENA.take(weight.numel(), weight.device) if _ARENA is not None else None
        if flat is not None:  # a zeroed slice of the micro-batch's gradient arena: no memset launch per convolution
            # (the weight's OWN strides, size-1 dimensions included: torch's multi-tensor optimizer kernels take their fast path only for
            #  gradients whose strides equal the parameters' — a permuted view differs in the stride of a one-tap dimension, and a
            #  foreach SGD then ran one launch per tensor: +4.8 ms on a 47 ms step, profiles/r06/train_one_item_arena_sgd_ab.log)
            dw = torch.as_strided(flat, tuple(weight.shape), tuple(weight.stride()))
            ops.conv3d_wgrad_x3_sub_f32(dy.permute(0, 2, 3, 4, 1), x.permute(0, 2, 3, 4, 1), flat, dims, cin, cout, kernel, stride,
                                        padding, (0, 0, 0), cin, cout, taps * cin, False)
        else:
            dw = torch.empty_like(weight)  # channels-last strides: memory [cout][kt][kh][kw][cin], the kernel's order
            ops.conv3d_wgrad_x3_f32(dy.permute(0, 2, 3, 4, 1), x.permute(0, 2, 3, 4, 1), dw.permute(0, 2, 3, 4, 1), dims, cin, cout,
                                    kernel, stride, padding, cin, cout)
    elif (_WGRAD_X3 and cin % 8 and cin != 3 and x.shape[1] == 8 and taps <= 28 and max(x.numel(), dy.numel()) < (1 << 31) - 64):
        # a few-channel first layer (VGGish: 1 mel channel) on its zero-padded 8-channel copy; the padded taps' gradient is dropped
        CALLS["wgrad_x3"] += 1
        dims = (x.shape[0], x.shape[2], x.shape[3], x.shape[4])
        dw8 = torch.empty((cout, 8) + tuple(kernel), dtype=torch.float32, device=dy.device, memory_format=torch.channels_last_3d)
        ops.conv3d_wgrad_x3_f32(dy.permute(0, 2, 3, 4, 1), x.permute(0, 2, 3, 4, 1), dw8.permute(0, 2, 3, 4, 1), dims, 8, cout, kernel,
                                stride, padding, 8, cout)
        dw = torch.empty_like(weight)
        dw.copy_(dw8[:, :cin])
    elif (_WGRAD_X3 and _STEM_WGRAD_X3 and cin == 3 and x.shape[1] == 8 and kernel[1] * kernel[2] <= 49 and
          max(x.numel(), dy.numel()) < (1 << 31) - 64):
        # the stems: 3 input channels travel as 4 of the forward's 8-channel padded clip; a [kt,7,7] filter is kt slices of
        # 49 taps, each a weight-gradient problem of its own with the temporal padding shifted by its frame tap
        kt, kh, kw = kernel
        acc = torch.zeros((cout, kt, kh * kw, 4), dtype=torch.float32, device=dy.device)
        xr, dyr = x.permute(0, 2, 3, 4, 1), dy.permute(0, 2, 3, 4, 1)
        dims = (x.shape[0], x.shape[2], x.shape[3], x.shape[4])
        for dt in range(kt):
            ops.conv3d_wgrad_x3_sub_f32(dyr, xr, acc[:, dt], dims, 4, cout, (1, kh, kw), stride, (padding[0] - dt, padding[1], padding[2]),
                                        (dy.shape[2], dy.shape[3], dy.shape[4]), 8, cout, kt * kh * kw * 4, False)
        CALLS["wgrad_stem_x3"] += 1
        dw = torch.empty_like(weight)
        dw.copy_(acc.view(cout, kt, kh, kw, 4)[..., :3].permute(0, 4, 1, 2, 3))
    else:  # (stems when AVT_TRAIN_STEM_WGRAD_X3=0, filters of more than 28 taps): MIOpen
        if x.shape[1] != weight.shape[1]:  # the padded copy was saved: MIOpen wants the 3-channel clip
            x = x[:, : weight.shape[1]].contiguous(memory_format=torch.channels_last_3d)
        CALLS["miopen_wgrad"] += 1
        dw = torch.ops.aten.convolution_backward(dy, x, weight, None, stride, padding, (1, 1, 1), False, (0, 0, 0), 1,
                                                 [False, True, False])[1]
    return dw


def _conv_dgrad(ctx, dy, dalias, x, weight, stride, padding, kernel, cin, cout):
    from . import ops
    dx = None
    if ctx.needs_input_grad[0]:
        if dalias is not None:
            dalias = dalias.contiguous(memory_format=torch.channels_last_3d)
        if (cin % 8 == 0 and stride == (1, 1, 1) and all(2 * p == k - 1 for p, k in zip(padding, kernel)) and
                dy.numel() < (1 << 30) - 64):
            CALLS["dgrad_x3"] += 1
            hb = getattr(ctx, "bn_in", None)  # x is a fused BatchNorm's output: dx is that BatchNorm's output gradient
            ctx.bn_in = None  # (the handle holds that BatchNorm's input and mask: not past this backward)
            if hb is not None and not (_EPI_BWD and hb.c == cin and tuple(hb.x.shape) == tuple(x.shape) and x.shape[0] % hb.groups == 0):
                hb = None
            g = _group_factor(cout, cin, kernel, (1, 1, 1), padding, dy.shape[4])
            if g > 1:
                planes, gk, rg = _grouped_planes(weight, g, True, ops.X3_BF16)
                dx = _conv_x3_rows(dy, planes, ops.X3_BF16, cout, cin, kernel, (1, 1, 1), padding, add=dalias, group=(g, gk, rg), bwd_bn=hb)
            else:
                dx = _conv_x3_rows(dy, _weight_planes(weight, True), ops.X3_BF16, cout, cin, kernel, (1, 1, 1), padding, add=dalias,
                                   bwd_bn=hb)
        else:
            dx = _dgrad_strided(dy, weight, stride, padding, kernel, tuple(x.shape)) if (_DGRAD_S_X3 and cin % 8 == 0) else None
            if dx is not None:
                CALLS["dgrad_strided_x3"] += 1
            else:
                CALLS["miopen_dgrad"] += 1
                dx = torch.ops.aten.convolution_backward(dy, x, weight, None, stride, padding, (1, 1, 1), False, (0, 0, 0), 1,
                                                         [True, False, False])[0]
            if dalias is not None:
                dx = dx + dalias
    return dx


def _stride_classes(k, s, p, x, y):
    """One dimension of a strided convolution's input gradient: for every residue r of the input index mod s, the taps that
    reach it (descending, so that the offsets into dY ascend), the padding in front of dY and the number of input positions
    -> [(r, taps, pad_before, count)]; None when a class would need dY shifted the other way (not a SlowFast shape)."""
    out = []
    for r in range(s):
        taps = list(range((r + p) % s, k, s))
        offs = [(r + p - d) // s for d in taps]  # descending with d ascending
        if offs and min(offs) > 0:
            return None
        out.append((r, taps[::-1], -min(offs) if offs else 0, max(0, -(-(x - r) // s))))
    return out


def _dgrad_strided(dy, weight, stride, padding, kernel, xshape):
    """Input gradient of a STRIDED convolution as convolutions of dY, one per residue class of the input position modulo the
    stride (a transposed convolution decomposed so that no zero is multiplied): class r gathers the taps d = (r + p) mod s,
    + s, ... and writes the positions s i + r through the kernel's output-row remap.  SlowFast's three kinds: [1,3,3] stride
    (1,2,2) (4 classes of 1, 2, 2, 4 taps), 1x1x1 stride (1,2,2) (one class; the other positions are zeros), [7,1,1] stride
    (4,1,1) (4 temporal classes of 1, 2, 2, 2 taps).  -> dx, or None outside that domain (the caller falls back to MIOpen)."""
    from . import ops
    from .fused_slowfast import split_planes
    b, cin, t, h, w = xshape
    cout = weight.shape[0]
    st, sh, sw = stride
    if not ((st == 1 and sh == sw and sh > 1) or (sh == 1 and sw == 1 and st > 1)) or cin % 8 or cout % 8:
        return None
    if (st > 1 and t % st) or (sh > 1 and (h % sh or w % sw)) or dy.numel() >= (1 << 30) - 64:
        return None
    to, ho, wo = dy.shape[2], dy.shape[3], dy.shape[4]
    if st > 1 and to * st != t:
        return None
    key = (id(weight), "strided", tuple(stride), tuple(padding))
    classes = _plane_lookup(key, weight)
    if classes is None:
        per_dim = [_stride_classes(k, s_, p, x, y) for k, s_, p, x, y in zip(kernel, stride, padding, (t, h, w), (to, ho, wo))]
        if any(c is None for c in per_dim):
            return None
        classes, jobs = [], []
        with torch.no_grad():
            wd = weight.detach()
            fast = (_PLANES_HIP and wd.dtype == torch.float32 and wd.is_contiguous(memory_format=torch.channels_last_3d) and
                    cout % 2 == 0)
            wf = None if fast else wd.float()  # [cout, cin, kt, kh, kw]
            w3 = wd.permute(0, 2, 3, 4, 1).reshape(cout, kernel[0] * kernel[1] * kernel[2], cin) if fast else None  # a view
            for rt, dt, pbt, nt in per_dim[0]:
                for rh, dh, pbh, nh in per_dim[1]:
                    for rw, dw, pbw, nw in per_dim[2]:
                        if not (dt and dh and dw):
                            classes.append(((rt, rh, rw), None))
                            continue
                        if fast:  # one launch: the class's taps (row-major over its (dt, dh, dw) lists) gathered and transposed
                            sel = [(a * kernel[1] + b_) * kernel[2] + c_ for a in dt for b_ in dh for c_ in dw]
                            hi, lo = ops.weight_planes_t_f32(w3, sel)
                            jobs.append(_job_transposed(w3, weight, hi, lo, cout, kernel[0] * kernel[1] * kernel[2], cin, sel))
                        else:
                            sub = wf[:, :, dt][:, :, :, dh][:, :, :, :, dw]                  # [cout, cin, |dt|, |dh|, |dw|]
                            wt = sub.transpose(0, 1).permute(0, 2, 3, 4, 1).reshape(cin, -1)   # rows ci, K = (taps, co)
                            hi, lo = split_planes(wt, ops.X3_BF16)
                        classes.append(((rt, rh, rw), (hi, lo, (len(dt), len(dh), len(dw)), (pbt, pbh, pbw), (nt, nh, nw))))
        _plane_store(key, weight, classes, jobs if fast else None)
    dx = torch.empty((b, cin, t, h, w), dtype=torch.float32, device=dy.device, memory_format=torch.channels_last_3d)
    if any(c[1] is None for c in classes):  # classes no tap reaches (a 1x1x1 filter at stride 2: three of four) are zeros
        dx.zero_()
    rows = dx.permute(0, 2, 3, 4, 1).reshape(-1, cin)  # (a view: channels-last memory)
    dyr = dy.permute(0, 2, 3, 4, 1)
    for (rt, rh, rw), c in classes:
        if c is None:
            continue
        hi, lo, kern, pad_b, n = c
        if n[0] * n[1] * n[2] == 0:
            continue
        if st > 1:   # temporal classes: frame i_t of class rt is frame st * i_t + rt: a grid of st * h rows per group of st frames
            out_rows, first = (1, st * h, w), rt * h * w
        else:        # spatial classes: position (i_h, i_w) of class (rh, rw) is (sh * i_h + rh, sw * i_w + rw)
            out_rows, first = (sh, h, w), rh * w + rw
        ops.conv3d_igemm_x3_f32_ex(dyr, hi, lo, None, rows[first:], _ktab(cout, kern, ho, wo, cout, dy.device), (b, to, ho, wo), cout,
                                   cin, kern, pad_b, n, cout, cin, ops.X3_BF16, out_rows=out_rows)
    return dx


class _ConvX3Fork(torch.autograd.Function):
    """conv(x) AND x itself (an alias): the block's shortcut takes the alias, so the gradient that comes back through the
    shortcut arrives HERE and is added in the epilogue of this convolution's input-gradient kernel — autograd would
    otherwise sum the two contributions to x with a separate elementwise pass (5.5 % of the step's device time)."""

    @staticmethod
    def forward(ctx, x, weight, stride, padding, stats=0, bn_in=None):
        y = _ConvX3.forward(ctx, x, weight, stride, padding, stats, bn_in)
        return y, x.view_as(x)

    @staticmethod
    def backward(ctx, dy, dalias):
        if dy is None:  # (the convolution's output was not used: only the alias carries a gradient)
            return (dalias if ctx.needs_input_grad[0] else None), None, None, None, None, None
        return _conv_backward(ctx, dy, dalias) + (None, None, None, None)


# ------------------------------------------------------------------------------------------------------------------------
# The stems: Conv3d(3, C, [kt,7,7], stride [1,2,2], pad [kt//2,3,3]) on the raw clip.  As an implicit GEMM over 8 padded
# channels the fast stem's forward ran 4.7 ms and its weight gradient (five [1,7,7] slices of wgrad_x3) 9.5 ms of a 95 ms item
# (profiles/r03/train_layers_before_stem.log): both gather every pixel once per tap.  The patch-resident kernels stage the
# input rows once: the forward is the inference stem kernel (pixel-pair form, time-grouped for the 8-channel fast stem, fp16
# planes) writing fp32, the weight gradient csrc/stem_train.hip (bf16 planes).  The clip is re-split into planes in each
# direction (a 12-byte read and a 16-byte write per pixel) instead of kept: the saved tensor is the caller's clip itself.
_STEM_IMAGES = {}


def _stem_tgroup(cout, t):
    """Output frames computed together as channels of one MFMA tile (fused_slowfast.stem_conv): 32 / cout, or None."""
    if cout % 32 == 0:
        return 1
    g = 32 // cout if 32 % cout == 0 else 0
    return g if g > 1 and t % g == 0 else None


def stem_patch_ok(x, conv):
    from . import ops
    if not (_STEM_PATCH and conv.in_channels == 3 and x.dim() == 5 and x.shape[1] == 3):
        return False
    kt = conv.kernel_size[0]
    _, _, t, h, w = x.shape
    return (tuple(conv.kernel_size[1:]) == (7, 7) and tuple(conv.stride) == (1, 2, 2) and tuple(conv.padding) == (kt // 2, 3, 3) and
            w % 2 == 0 and _stem_tgroup(conv.out_channels, t) is not None and
            bool(_lib.lib().avt_stem_conv_supported(h, w // 2, conv.out_channels * _stem_tgroup(conv.out_channels, t))) and
            ops.stem_wgrad_x3_supported(h, w // 2, conv.out_channels, kt) and x.numel() // 3 * 16 < (1 << 32) - 64)


def _stem_weight_image(weight, g):
    """(hi, lo, wscale) of a stem weight [C, 3, kt, 7, 7] in the LDS image order of csrc/stem_conv.hip: pixel-pair taps
    (column tap k -> pair tap (k + 1) // 2, pixel (k + 1) % 2; the first is a structural zero), g output frames as g * C
    channels over kt + g - 1 frame taps, fp16 planes of rows scaled into [2^9, 2^10).  Cached until the tensor changes."""
    from . import ops
    from .fused_slowfast import split_planes, stem_lds_image
    key = (id(weight), g)
    hit = _STEM_IMAGES.get(key)
    if hit is not None and hit[0]() is weight and hit[1] == weight._version:
        return hit[2]
    with torch.no_grad():
        w = weight.detach().float()
        c, _, kt = w.shape[0], w.shape[1], w.shape[2]
        w8 = w.new_zeros((c, kt, 7, 8, 4))
        w8[:, :, :, 1:, :3] = w.permute(0, 2, 3, 4, 1)
        wg = w.new_zeros((g, c, kt + g - 1, 7, 8, 4))
        for j in range(g):
            wg[j, :, j : j + kt] = w8
        wt = wg.reshape(g * c, -1)
        mx = wt.abs().amax(dim=1).clamp_min(1e-30)
        sc = torch.pow(2.0, 9.0 - torch.floor(torch.log2(mx)))
        hi, lo = split_planes(wt * sc.view(-1, 1), ops.X3_F16)
        fm = g == 4 and c == 8  # frame-major tiles: the kernel skips the frame taps a tile never meets
        img = (stem_lds_image(hi, kt + g - 1, fm), stem_lds_image(lo, kt + g - 1, fm), (1.0 / sc).float().contiguous(), 2 if fm else 0)
    if len(_STEM_IMAGES) > 64:
        for k in [k for k, v in _STEM_IMAGES.items() if v[0]() is None]:
            del _STEM_IMAGES[k]
    _STEM_IMAGES[key] = (weakref.ref(weight), weight._version, img)
    return img


class _StemX3(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight):
        from . import ops
        b, _, t, h, w = x.shape
        c, kt = weight.shape[0], weight.shape[2]
        g = _stem_tgroup(c, t)
        xh, xl = ops.clip_planes_f32(x, ops.X3_F16)
        wh, wl, wscale, fpt = _stem_weight_image(weight, g)
        y = torch.empty((b, c, t, h // 2, w // 2), dtype=torch.float32, device=x.device, memory_format=torch.channels_last_3d)
        CALLS["conv_fwd_x3"] += 1
        CALLS["stem_fwd_patch"] += 1
        ops.stem_conv_x3_f32(xh, xl, wh, wl, wscale, y, b, t, h, w // 2, g * c, kt + g - 1, g, kt // 2, g, ops.X3_F16, frames_per_tile=fpt)
        ctx.save_for_backward(x, weight)
        return y

    @staticmethod
    def backward(ctx, dy):
        from . import ops
        x, weight = ctx.saved_tensors
        if not ctx.needs_input_grad[1]:
            return None, None
        b, _, t, h, w = x.shape
        c, kt = weight.shape[0], weight.shape[2]
        dy = dy.contiguous(memory_format=torch.channels_last_3d)
        xh, xl = ops.clip_planes_f32(x, ops.X3_BF16)
        CALLS["wgrad_stem_patch"] += 1
        dwp = ops.stem_wgrad_x3(xh, xl, dy, b, t, h, w // 2, c, kt, kt // 2)  # [c, kt, 7, 4 pair taps, 2 pixels x 4 channels]
        dw = torch.empty_like(weight)
        dw.copy_(dwp.view(c, kt, 7, 8, 4)[:, :, :, 1:, :3].permute(0, 4, 1, 2, 3))  # column tap k = 2 * pair + pixel - 1
        return None, dw


class _MaxPoolHW(torch.autograd.Function):
    """MaxPool3d((1,3,3),(1,2,2),(0,1,1)) on channels-last fp32 (csrc/stem_train.hip): the forward records which tap held each
    maximum (4 bits per element), the backward gathers the gradient from it."""

    @staticmethod
    def forward(ctx, x, cat=None):
        b, c, t, h, w = x.shape
        ho, wo = (h - 1) // 2 + 1, (w - 1) // 2 + 1
        if cat is None:
            y, ldy = torch.empty((b, c, t, ho, wo), dtype=torch.float32, device=x.device, memory_format=torch.channels_last_3d), 0
        else:  # the first c channels of a concatenation buffer's rows (join_channels)
            y, ldy = _alias(cat[0], cat[1], c), cat[0].shape[1]
        tap = torch.empty(y.numel() // 2, dtype=torch.uint8, device=x.device)
        CALLS["maxpool_hip"] += 1
        _lib.check(_lib.lib().avt_maxpool_train_fwd(_p(x), _p(y), _p(tap), b * t, h, w, c, ldy, _stream()), "avt_maxpool_train_fwd")
        ctx.save_for_backward(tap)
        ctx.dims = (b, c, t, h, w)
        return y

    @staticmethod
    def backward(ctx, dy):
        (tap,) = ctx.saved_tensors
        b, c, t, h, w = ctx.dims
        ld_dy = _row_ld(dy)
        if ld_dy is None:
            dy, ld_dy = dy.contiguous(memory_format=torch.channels_last_3d), c
        dx = torch.empty((b, c, t, h, w), dtype=torch.float32, device=dy.device, memory_format=torch.channels_last_3d)
        _lib.check(_lib.lib().avt_maxpool_train_bwd(_p(dy), _p(tap), _p(dx), b * t, h, w, c, ld_dy, _stream()), "avt_maxpool_train_bwd")
        return dx, None


def max_pool_hw(x, pool, cat_extra=0):
    """pool(x) for the stems' nn.MaxPool3d((1,3,3),(1,2,2),(0,1,1)): the HIP pair above on channels-last fp32 device tensors
    that carry a gradient in train mode, the module itself otherwise.  cat_extra: as in bn_act (the HIP path only)."""
    if (_FUSED and x.is_cuda and x.dtype == torch.float32 and x.dim() == 5 and x.requires_grad and x.shape[1] % 4 == 0 and
            x.is_contiguous(memory_format=torch.channels_last_3d) and tuple(pool.kernel_size) == (1, 3, 3) and
            tuple(pool.stride) == (1, 2, 2) and tuple(pool.padding) == (0, 1, 1) and not pool.ceil_mode and
            pool.dilation in (1, (1, 1, 1)) and x.numel() // 4 < (1 << 32)):
        if not cat_extra or not _JOIN:
            return _MaxPoolHW.apply(x)
        b, c, t, h, w = x.shape
        cat = (_cat_buffer((b, c, t, (h - 1) // 2 + 1, (w - 1) // 2 + 1), c + int(cat_extra), x.device), 0)
        y = _MaxPoolHW.apply(x, cat)
        y._avt_cat = cat
        return y
    return pool(x)


def conv_fusable(x, conv):
    """(in_channels == 3: the stems — the clip is zero-padded to 8 channels for the forward; it needs no input gradient.)"""
    return (_CONV_X3 and conv.training and x.is_cuda and x.dtype == torch.float32 and x.dim() == 5 and conv.bias is None and
            conv.groups == 1 and tuple(conv.dilation) == (1, 1, 1) and conv.out_channels % 8 == 0 and
            (conv.in_channels % 8 == 0 or (conv.in_channels == 3 and not x.requires_grad)) and
            max(conv.kernel_size) <= 8 and conv.weight.dtype == torch.float32 and conv.padding_mode == "zeros" and
            not isinstance(conv.padding, str) and x.numel() < (1 << 30) - 64 and
            conv.weight.is_contiguous(memory_format=torch.channels_last_3d) and  # the model was put in the training layout
            (max(conv.kernel_size) > 1 or x.is_contiguous(memory_format=torch.channels_last_3d)))  # (a 1x1x1 weight is both layouts)


def conv2d(x, conv):
    """conv(x) for a Conv2d module (VGGish, models/audio_models/vggish.py:15-33, in the m=2 training branch, models.py:343-345,
    405-407): the same split-plane kernels, the image as a one-frame clip — NHWC rows are NDHWC rows with T = 1 — forward,
    input gradient and weight gradient; the bias is added by torch.  The module's weight is put in channels-last memory on first
    use (its [cout][kh][kw][cin] order is the kernels' K order).  Stock op otherwise (eval mode, CPU, other dtypes)."""
    ok = (_CONV_X3 and conv.training and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and conv.groups == 1 and
          tuple(conv.dilation) == (1, 1) and conv.out_channels % 8 == 0 and conv.weight.dtype == torch.float32 and
          conv.padding_mode == "zeros" and not isinstance(conv.padding, str) and max(conv.kernel_size) <= 5 and
          (conv.in_channels % 8 == 0 or (conv.in_channels < 8 and not x.requires_grad)) and x.numel() < (1 << 30) - 64)
    if not ok:
        return conv(x)
    if not conv.weight.is_contiguous(memory_format=torch.channels_last):
        # (main.py puts the model in the training layout BEFORE the optimizer / DDP buckets exist: training_layout(); this is the
        #  guard for callers that did not — a layout change, not a value change)
        conv.weight.data = conv.weight.data.contiguous(memory_format=torch.channels_last)
    x5 = x.contiguous(memory_format=torch.channels_last).unsqueeze(2)   # [n, C, 1, H, W]: channels_last_3d strides
    w5 = conv.weight.unsqueeze(2)                                       # [cout, cin, 1, kh, kw], a view (autograd folds it back)
    y = _ConvX3.apply(x5, w5, (1,) + tuple(conv.stride), (0,) + tuple(conv.padding)).squeeze(2)
    return y if conv.bias is None else y + conv.bias.view(1, -1, 1, 1)


def training_layout(model):
    """The model in the layout the hand-written training passes read: channels_last_3d Conv3d weights / activations, channels_last
    Conv2d weights (VGGish) — once, at setup, before optimizers or DDP buckets are built over the parameters."""
    model = model.to(memory_format=torch.channels_last_3d)
    with torch.no_grad():
        for m in model.modules():
            if isinstance(m, torch.nn.Conv2d) and not m.weight.is_contiguous(memory_format=torch.channels_last):
                m.weight.data = m.weight.data.contiguous(memory_format=torch.channels_last)
    return model


def _stat_groups(x, bn):
    """Replica groups of the train-mode BatchNorm `bn` that consumes a convolution's output (conv3d(..., stats=bn)), or 0 when the
    fused BatchNorm pass will not run on it (eval mode, switched off, a channel count outside its domain)."""
    if bn is None or not (_EPI_STATS and _FUSED and bn.training and bn.weight is not None and bn.weight.dtype == torch.float32):
        return 0
    groups = _BN_GROUPS if _BN_GROUPS > 1 else 1
    return groups if x.shape[0] % groups == 0 else 0


def conv3d_fork(x, conv, stats=None):
    """-> (conv(x), x'): x' is x for every purpose but autograd's — hand it to the OTHER consumers of x (the block's shortcut
    or projection) and their gradient is summed into conv's input gradient inside its kernel.  Stock path: (conv(x), x).
    stats = the BatchNorm module conv(x) feeds (see conv3d)."""
    if not _FORK:
        return conv3d(x, conv, stats), x
    if not conv_fusable(x, conv) or conv.in_channels % 8:
        return conv(x), x
    x = x.contiguous(memory_format=torch.channels_last_3d)
    y, alias = _ConvX3Fork.apply(x, conv.weight, tuple(conv.stride), tuple(conv.padding), _stat_groups(x, stats),
                                 getattr(x, "_avt_bn", None) if _EPI_BWD else None)
    return _tag_stats(y), alias


def conv3d(x, conv, stats=None):
    """conv(x) for a Conv3d module: the split-plane MFMA kernels in train mode on fp32 device tensors of a model in the
    training layout (forward, stride-1 input gradient, weight gradient), the module itself otherwise.
    stats = the nn.BatchNorm3d the output goes into next (bn_act(conv3d(x, conv, stats=bn), bn)): the convolution's epilogue sums
    the statistics that BatchNorm needs while the tile is still in the LDS, and bn_act skips its pass over the output."""
    if not conv_fusable(x, conv):
        return conv(x)
    if conv.in_channels == 3 and stem_patch_ok(x, conv):
        return _StemX3.apply(x, conv.weight)
    if conv.in_channels % 8 and not x.is_contiguous(memory_format=torch.channels_last_3d):
        pass  # (the stem's padded copy is built channels-last from any layout)
    else:
        x = x.contiguous(memory_format=torch.channels_last_3d)  # no-op inside the network; a clip handed over as NCDHW is transposed once
    return _tag_stats(_ConvX3.apply(x, conv.weight, tuple(conv.stride), tuple(conv.padding), _stat_groups(x, stats),
                                    getattr(x, "_avt_bn", None) if _EPI_BWD else None))
