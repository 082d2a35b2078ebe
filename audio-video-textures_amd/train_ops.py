"""Training-step fusion on the MI355X: train-mode BatchNorm3d + shortcut add + ReLU as one HIP pass in
each direction (csrc/bn_train.hip), as a torch.autograd.Function the SlowFast modules call in train mode.

The convolutions of the training step stay MIOpen's (forward / dgrad / wgrad through autograd); what this removes is the
third of the step that is not convolution: MIOpenBatchNormFwdTrainSpatial, MIOpenBatchNormBwdSpatial and the separate
add / ReLU / ReLU-backward passes (profiles/r02/train_fp32_steady_state_kernels.log)."""
import ctypes as C
import os

import torch
import torch.nn.functional as F

from . import _lib

_FUSED = int(os.environ.get("AVT_FUSED_BN", "1"))


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _rows(x):
    """(m, c) of a [B,C,T,H,W] tensor whose memory is channels-last rows [B*T*H*W, C]."""
    return x.numel() // x.shape[1], x.shape[1]


def _workspace(m, c, device):
    """Scratch for the per-workgroup partial sums and the per-channel coefficients (sized by the library)."""
    n = _lib.lib().avt_bn_train_ws_bytes(m, c)
    if n < 0:
        raise ValueError("bn_train: %d rows x %d channels is outside the kernel's domain" % (m, c))
    return torch.empty(n, dtype=torch.uint8, device=device)


def fusable(x, bn, res=None):
    c = x.shape[1]
    return (_FUSED and bn.training and x.is_cuda and x.dtype == torch.float32 and x.dim() == 5 and c >= 8 and (c & (c - 1)) == 0 and
            x.is_contiguous(memory_format=torch.channels_last_3d) and bn.weight is not None and bn.weight.dtype == torch.float32 and
            (res is None or (res.shape == x.shape and res.dtype == torch.float32 and
                             res.is_contiguous(memory_format=torch.channels_last_3d))))


class _BNAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, res, relu, momentum, eps):
        m, c = _rows(x)
        y = torch.empty_like(x)  # keeps the channels-last strides
        ws = _workspace(m, c, x.device)
        save_mean = torch.empty(c, dtype=torch.float32, device=x.device)
        save_invstd = torch.empty(c, dtype=torch.float32, device=x.device)
        _lib.check(_lib.lib().avt_bn_train_fwd(_p(x), _p(res), _p(y), m, c, _p(weight), _p(bias), float(eps), float(momentum),
                                               1 if relu else 0, _p(ws), ws.numel(), _p(save_mean), _p(save_invstd),
                                               _p(running_mean), _p(running_var), _stream()), "avt_bn_train_fwd")
        ctx.save_for_backward(x, y if relu else None, weight, save_mean, save_invstd)
        ctx.has_res = res is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, weight, save_mean, save_invstd = ctx.saved_tensors
        m, c = _rows(x)
        dy = dy.contiguous(memory_format=torch.channels_last_3d)
        dx = torch.empty_like(x)
        dres = torch.empty_like(x) if ctx.has_res else None
        ws = _workspace(m, c, x.device)
        dgamma = torch.empty(c, dtype=torch.float32, device=x.device)
        dbeta = torch.empty(c, dtype=torch.float32, device=x.device)
        _lib.check(_lib.lib().avt_bn_train_bwd(_p(dy), _p(y), _p(x), m, c, _p(weight), _p(save_mean), _p(save_invstd), _p(ws), ws.numel(),
                                               _p(dx), _p(dres), _p(dgamma), _p(dbeta), _stream()), "avt_bn_train_bwd")
        return dx, dgamma, dbeta, None, None, dres, None, None, None


def bn_act(x, bn, res=None, relu=True):
    """act(bn(x) [+ res]) for a BatchNorm3d module `bn`: the fused HIP pass in train mode on channels-last fp32 device tensors,
    the stock torch ops otherwise (eval mode, other layouts / dtypes, CPU) — same result, same running-statistics update."""
    if not fusable(x, bn, res):
        y = bn(x)
        if res is not None:
            y = y + res
        return F.relu(y) if relu else y
    momentum = bn.momentum
    if bn.track_running_stats and bn.num_batches_tracked is not None:
        bn.num_batches_tracked.add_(1)
        if momentum is None:  # cumulative moving average
            momentum = 1.0 / float(bn.num_batches_tracked)
    rm, rv = (bn.running_mean, bn.running_var) if bn.track_running_stats else (None, None)
    return _BNAct.apply(x, bn.weight, bn.bias, rm, rv, res, relu, 0.0 if momentum is None else momentum, bn.eps)
