"""Torch-tensor wrappers over the C ABI (include/avt.h).

Plumbing only: tensors supply device memory and the current HIP stream; every
operator below is one call into libavt_hip.so.  CPU tensors are rejected —
there is no fallback path.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib

SIM_BF16, SIM_BF16X3, SIM_F32 = 0, 1, 2
_PREC = {"bf16": SIM_BF16, "bf16x3": SIM_BF16X3, "f32": SIM_F32}
FAST_T, SLOW_T, SLOTS = 32, 8, 40


def _dev(t, name, dtype=None, contiguous=True):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise _lib.AvtError("%s must be a device (HIP) tensor; the hot path has no CPU fallback" % name)
    if dtype is not None and t.dtype != dtype:
        raise _lib.AvtError("%s must be %s, got %s" % (name, dtype, t.dtype))
    if contiguous and not t.is_contiguous():
        raise _lib.AvtError("%s must be contiguous" % name)
    return t


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


_RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", None)  # the current stream's handle without building a Stream object


def _stream():
    # (one call per launch, ~2900 a step at one item per rank — a host-bound step: the raw getter is ~1.5 us cheaper than
    #  torch.cuda.current_stream().cuda_stream)
    if _RAW_STREAM is not None:
        return C.c_void_p(_RAW_STREAM(torch.cuda.current_device()))
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


_SIDE = {}


def side_streams(device, n):
    """The first n of the process-wide side streams of `device` (created on first use, never more than asked for in total): every
    multi-stream path of the package — the synthesis engine's q / t encoder streams, the training step's query-encoder stream —
    draws from this one list, so that a given role always runs on the same stream (and hardware queue) whatever ran before it."""
    dev = torch.device(device)
    key = (dev.type, dev.index if dev.index is not None else torch.cuda.current_device())
    lst = _SIDE.setdefault(key, [])
    while len(lst) < n:
        lst.append(torch.cuda.Stream(device=dev))
    return lst[:n]


def device_check():
    buf = C.create_string_buffer(64)
    _lib.check(_lib.lib().avt_device_check(buf, 64), "avt_device_check")
    return buf.value.decode()


# ---- clip_pack ---------------------------------------------------------------
def clip_sample_table(win_len):
    """(fast_idx[32], slow_idx[8]) — torch.linspace(0, W-1, 32).long() and its 8-subsample."""
    fast = np.empty(FAST_T, np.int32)
    slow = np.empty(SLOW_T, np.int32)
    _lib.check(_lib.lib().avt_clip_sample_table(int(win_len), fast.ctypes.data, slow.ctypes.data),
               "avt_clip_sample_table")
    return fast, slow


def clip_pack_plan(win_start, win_len, n_frames):
    """Host CSR plan frame -> (window, slot) destinations."""
    win_start = np.ascontiguousarray(win_start, np.int32)
    n_win = win_start.shape[0]
    off = np.empty(n_frames + 1, np.int32)
    slot = np.empty(max(n_win * SLOTS, 1), np.int32)
    _lib.check(_lib.lib().avt_clip_pack_plan(win_start.ctypes.data, n_win, int(win_len), int(n_frames),
                                             off.ctypes.data, slot.ctypes.data), "avt_clip_pack_plan")
    return off, slot[: n_win * SLOTS]


class SplitClip:
    """A packed clip batch in the split-plane ("x3") format: two 16-bit planes hi / lo of identical geometry
    [n, T, hw, hw, 4], value = hi + lo (include/avt.h).  Quacks like the tensor the bf16 path passes around."""

    def __init__(self, hi, lo, plane_dtype):
        self.hi, self.lo, self.plane_dtype = hi, lo, plane_dtype

    @property
    def shape(self):
        return self.hi.shape

    def chunk(self, parts):
        return [SplitClip(h, l, self.plane_dtype) for h, l in zip(self.hi.chunk(parts), self.lo.chunk(parts))]

    def record_stream(self, st):
        self.hi.record_stream(st)
        self.lo.record_stream(st)

    def float(self):
        dt = torch.float16 if self.plane_dtype == X3_F16 else torch.bfloat16
        return self.hi.view(dt).float() + self.lo.view(dt).float()


class FrameClip:
    """A batch of clips as a TABLE of distinct packed frames plus an index (round 4): planes hi / lo [F, hw, hw, 4] (F distinct
    source frames, resized / normalised once each) and idx [n, T] int32 — clip i's frame t is table frame idx[i, t].  Overlapping
    windows (W / S of their frames each) and the fast pathway's repeated frames share table rows: 0.55 GB instead of 5.3 GB per
    166 windows at W = 20, S = 4.  Quacks like SplitClip ([n, T, hw, hw, 4]); the patch-resident stem kernel reads it through the
    index (avt_stem_conv_x3 frame_idx), anything else asks for .dense()."""

    def __init__(self, hi, lo, idx, plane_dtype, start=None, win_len=None):
        self.hi, self.lo, self.idx, self.plane_dtype = hi, lo, idx, plane_dtype
        # when the index is a regular sampling of windows (clip_pack_frames): clip i = table frames start[i] + sample_table(win_len)
        self.start, self.win_len = start, win_len

    @property
    def shape(self):
        return (self.idx.shape[0], self.idx.shape[1]) + tuple(self.hi.shape[1:])

    @property
    def table_frames(self):
        return int(self.hi.shape[0])

    def chunk(self, parts):
        starts = self.start.chunk(parts) if self.start is not None else [None] * parts
        return [FrameClip(self.hi, self.lo, i.contiguous(), self.plane_dtype, s, self.win_len) for i, s in zip(self.idx.chunk(parts), starts)]

    def record_stream(self, st):
        for t_ in (self.hi, self.lo, self.idx, self.start):
            if t_ is not None:
                t_.record_stream(st)

    def dense(self):
        """-> the SplitClip [n, T, hw, hw, 4] this table stands for (a gather: shapes the table kernels do not cover)."""
        flat = self.idx.reshape(-1).long()
        shp = self.shape
        return SplitClip(self.hi.index_select(0, flat).view(shp), self.lo.index_select(0, flat).view(shp), self.plane_dtype)

    def float(self):
        return self.dense().float()


X3_BF16, X3_F16 = 0, 1
_X3 = {"bf16x3": X3_BF16, "f16x3": X3_F16}


def clip_pack(frames_u8, win_start, win_len, out_hw=224, mean=0.45, std=0.225, bgr=True,
              dtype=torch.bfloat16, plan=None, layout="ncthw", planes=None):
    """frames_u8 [F,H,W,3] uint8 RGB (device) -> slow [n,3,8,hw,hw], fast [n,3,32,hw,hw] (layout "ncthw"),
    or slow [n,8,hw,hw,4], fast [n,32,hw,hw,4] bf16 with a zero 4th channel (layout "ndhwc4", MFMA stem);
    planes = "bf16x3" | "f16x3" (ndhwc4 only): SplitClip pairs for the contract-grade encoder."""
    _dev(frames_u8, "frames_u8", torch.uint8)
    assert frames_u8.dim() == 4 and frames_u8.shape[3] == 3
    n_frames, h, w, _ = frames_u8.shape
    win_start = np.ascontiguousarray(win_start, np.int32)
    n_win = win_start.shape[0]
    if plan is None:
        off, slot = clip_pack_plan(win_start, win_len, n_frames)
        plan = (torch.from_numpy(off).to(frames_u8.device), torch.from_numpy(slot).to(frames_u8.device))
    d_off, d_slot = plan
    if layout == "ndhwc4":
        if dtype != torch.bfloat16:
            raise _lib.AvtError("clip_pack: layout ndhwc4 is bf16 only")
        mk = lambda t_: torch.empty((n_win, t_, out_hw, out_hw, 4), dtype=dtype, device=frames_u8.device)
        if planes is not None:
            pd = _X3[planes]
            slow, fast = SplitClip(mk(SLOW_T), mk(SLOW_T), pd), SplitClip(mk(FAST_T), mk(FAST_T), pd)
            _lib.check(_lib.lib().avt_clip_pack_u8_ndhwc4_x3(_p(frames_u8), n_frames, h, w, _p(d_off), _p(d_slot), n_win,
                                                             int(out_hw), float(mean), float(std), 1 if bgr else 0,
                                                             _p(slow.hi), _p(slow.lo), _p(fast.hi), _p(fast.lo), pd,
                                                             _stream()), "avt_clip_pack_u8_ndhwc4_x3")
            return slow, fast
        slow, fast = mk(SLOW_T), mk(FAST_T)
        _lib.check(_lib.lib().avt_clip_pack_u8_ndhwc4(_p(frames_u8), n_frames, h, w, _p(d_off), _p(d_slot), n_win,
                                                      int(out_hw), float(mean), float(std), 1 if bgr else 0, _p(slow),
                                                      _p(fast), _stream()), "avt_clip_pack_u8_ndhwc4")
        return slow, fast
    if dtype not in (torch.bfloat16, torch.float32):
        raise _lib.AvtError("clip_pack: dtype must be bfloat16 or float32")
    slow = torch.empty((n_win, 3, SLOW_T, out_hw, out_hw), dtype=dtype, device=frames_u8.device)
    fast = torch.empty((n_win, 3, FAST_T, out_hw, out_hw), dtype=dtype, device=frames_u8.device)
    _lib.check(_lib.lib().avt_clip_pack_u8(_p(frames_u8), n_frames, h, w, _p(d_off), _p(d_slot), n_win, int(out_hw),
                                           float(mean), float(std), 1 if bgr else 0, _p(slow), _p(fast),
                                           1 if dtype == torch.bfloat16 else 0, _stream()), "avt_clip_pack_u8")
    return slow, fast


def clip_pack_frames(frames_u8, win_start, win_len, out_hw=224, mean=0.45, std=0.225, bgr=True, planes="f16x3"):
    """clip_pack for the contract-grade encoder as a frame TABLE: every frame of frames_u8 [F,H,W,3] is resized / normalised /
    split ONCE into planes [F, hw, hw, 4] (the ndhwc4 kernel with one destination per frame), and the windows' temporal
    sampling (linspace(0, W-1, 32).long() and its 8-subsample) becomes two index arrays -> (slow FrameClip [n,8,..], fast
    FrameClip [n,32,..]) sharing the table."""
    _dev(frames_u8, "frames_u8", torch.uint8)
    assert frames_u8.dim() == 4 and frames_u8.shape[3] == 3
    n_frames, h, w, _ = frames_u8.shape
    win_start = np.ascontiguousarray(win_start, np.int64)
    if win_start.size and (win_start.min() < 0 or win_start.max() + win_len > n_frames):
        raise _lib.AvtError("clip_pack_frames: a window leaves the %d frames" % n_frames)
    pd = _X3[planes]
    dev = frames_u8.device
    f = np.arange(n_frames, dtype=np.int32)
    off = np.arange(n_frames + 1, dtype=np.int32)            # one destination per frame ...
    slot = (f // SLOW_T) * SLOTS + f % SLOW_T               # ... slot f of a "slow" tensor [ceil(F/8), 8, hw, hw, 4] = table row f
    n_grp = -(-n_frames // SLOW_T)
    hi = torch.empty((n_grp * SLOW_T, out_hw, out_hw, 4), dtype=torch.bfloat16, device=dev)
    lo = torch.empty_like(hi)
    d_off, d_slot = torch.from_numpy(off).to(dev, non_blocking=True), torch.from_numpy(slot.astype(np.int32)).to(dev, non_blocking=True)
    _lib.check(_lib.lib().avt_clip_pack_u8_ndhwc4_x3(_p(frames_u8), n_frames, h, w, _p(d_off), _p(d_slot), n_grp, int(out_hw),
                                                     float(mean), float(std), 1 if bgr else 0, _p(hi), _p(lo), _p(hi), _p(lo), pd,
                                                     _stream()), "avt_clip_pack_u8_ndhwc4_x3")
    fast_off, slow_off = clip_sample_table(win_len)
    idx_f = torch.from_numpy((win_start[:, None] + fast_off[None, :]).astype(np.int32)).to(dev, non_blocking=True)
    idx_s = torch.from_numpy((win_start[:, None] + slow_off[None, :]).astype(np.int32)).to(dev, non_blocking=True)
    start = torch.from_numpy(win_start.astype(np.int32)).to(dev, non_blocking=True)
    hi, lo = hi[:n_frames], lo[:n_frames]
    return FrameClip(hi, lo, idx_s, pd, start, int(win_len)), FrameClip(hi, lo, idx_f, pd, start, int(win_len))


def clip_pack_gather(frames_u8, win_start_dev, win_len, out_hw=224, mean=0.45, std=0.225, bgr=True, dtype=torch.float32):
    """clip_pack for scattered windows whose starts are a DEVICE int32 tensor (training batches): no host plan."""
    _dev(frames_u8, "frames_u8", torch.uint8)
    _dev(win_start_dev, "win_start", torch.int32)
    n_frames, h, w, _ = frames_u8.shape
    n_win = int(win_start_dev.numel())
    if dtype not in (torch.bfloat16, torch.float32):
        raise _lib.AvtError("clip_pack_gather: dtype must be bfloat16 or float32")
    slow = torch.empty((n_win, 3, SLOW_T, out_hw, out_hw), dtype=dtype, device=frames_u8.device)
    fast = torch.empty((n_win, 3, FAST_T, out_hw, out_hw), dtype=dtype, device=frames_u8.device)
    _lib.check(_lib.lib().avt_clip_pack_gather_u8(_p(frames_u8), n_frames, h, w, _p(win_start_dev), n_win, int(win_len),
                                                  int(out_hw), float(mean), float(std), 1 if bgr else 0, _p(slow), _p(fast),
                                                  1 if dtype == torch.bfloat16 else 0, _stream()), "avt_clip_pack_gather_u8")
    return slow, fast


def negative_sample(mt_state, idx, n_len, n_negs):
    """mt_state int32/uint32-bits [625] device tensor (np.random.get_state() key + pos), advanced in place; idx int64 [B]
    device -> negatives int32 [B, n_negs] (dataset.py:181-190 semantics, NumPy's stream)."""
    _dev(mt_state, "mt_state", torch.int32)
    _dev(idx, "idx", torch.int64)
    out = torch.empty((idx.numel(), int(n_negs)), dtype=torch.int32, device=idx.device)
    _lib.check(_lib.lib().avt_negative_sample_mt19937(_p(mt_state), _p(idx), int(idx.numel()), int(n_len), int(n_negs),
                                                      _p(out), _stream()), "avt_negative_sample_mt19937")
    return out


# ---- l2norm --------------------------------------------------------------------
def l2norm_rows(x0, x1=None, eps=1e-12, want_f32=True, want_split=False):
    """y = [x0|x1] / max(||.||, eps) row-wise -> (y_f32 | None, y_hi | None, y_lo | None)."""
    _dev(x0, "x0", torch.float32)
    n, d0 = x0.shape
    d1 = 0
    if x1 is not None:
        _dev(x1, "x1", torch.float32)
        assert x1.shape[0] == n
        d1 = x1.shape[1]
    d = d0 + d1
    y = torch.empty((n, d), dtype=torch.float32, device=x0.device) if want_f32 else None
    hi = torch.empty((n, d), dtype=torch.bfloat16, device=x0.device) if want_split else None
    lo = torch.empty((n, d), dtype=torch.bfloat16, device=x0.device) if want_split else None
    _lib.check(_lib.lib().avt_l2norm_rows(_p(x0), d0, _p(x1), d1, n, float(eps), _p(y), _p(hi), _p(lo), _stream()),
               "avt_l2norm_rows")
    return y, hi, lo


# ---- similarity -----------------------------------------------------------------
SIM_XL = True   # the bf16x3 similarity on the encoder's 256 x 256 LDS-DMA tile where the shape allows (avt_gemm_nt_x3_f32out)
_GEMM_TABS = {}


def _gemm_ktab(k, lda, device):
    key = (int(k), int(lda), str(device))
    tab = _GEMM_TABS.get(key)
    if tab is None:
        tab = torch.from_numpy(conv3d_ktab(int(k), (1, 1, 1), 1, 1, int(lda))).to(device)
        _GEMM_TABS[key] = tab
    return tab


def sim_gemm_nt(q, t, temp, precision="f32", q_lo=None, t_lo=None, out=None, nq_total=None):
    """out[i,j] = <q_i, t_j> / temp.  precision: "f32" (exact, canonical) | "bf16" | "bf16x3".
    nq_total: query rows of the WHOLE build when q is one rank's row block (default: max(rows of q, rows of t) — in a sharded build
    every rank holds all of T).  The bf16x3 tile is chosen from it, not from this call's rows: the two tiles accumulate in different
    orders (3-4e-6 apart), and a score must not depend on how many ranks built the matrix."""
    prec = _PREC[precision]
    nq_all = max(int(q.shape[0]), int(t.shape[0])) if nq_total is None else int(nq_total)
    if (SIM_XL and prec == SIM_BF16X3 and q_lo is not None and t_lo is not None and q.dim() == 2 and t.shape[0] % 256 == 0 and
            q.shape[1] % 32 == 0 and 256 <= q.shape[1] <= 8192 and (SIM_XL == "always" or -(-nq_all // 256) * (t.shape[0] // 256) >= 192) and
            (out is None or out.stride(0) % 4 == 0)):
        # (>= 192 tiles of the whole build: three quarters of a round of the 256 CUs — 4096^2 is exactly one round; at 2048^2 (64
        #  tiles) the 128 x 128 tile's 256 workgroups win: tools/probe_sim_xl.py)
        for name, v in (("q", q), ("q_lo", q_lo), ("t", t), ("t_lo", t_lo)):
            _dev(v, name, torch.bfloat16)
        nq, d = q.shape
        nt = t.shape[0]
        if out is None:
            out = torch.empty((nq, nt), dtype=torch.float32, device=q.device)
        else:
            _dev(out, "out", torch.float32, contiguous=False)
        _lib.check(_lib.lib().avt_gemm_nt_x3_f32out(_p(q), _p(q_lo), d, _p(t), _p(t_lo), _p(out), out.stride(0), nq, nt, d, float(temp),
                                                    _p(_gemm_ktab(d, d, q.device)), X3_BF16, _stream()), "avt_gemm_nt_x3_f32out")
        return out
    want = torch.float32 if prec == SIM_F32 else torch.bfloat16
    _dev(q, "q", want)
    _dev(t, "t", want)
    nq, d = q.shape
    nt = t.shape[0]
    assert t.shape[1] == d
    if prec == SIM_BF16X3:
        _dev(q_lo, "q_lo", torch.bfloat16)
        _dev(t_lo, "t_lo", torch.bfloat16)
    if out is None:
        out = torch.empty((nq, nt), dtype=torch.float32, device=q.device)
    else:
        _dev(out, "out", torch.float32)
    _lib.check(_lib.lib().avt_sim_gemm_nt(_p(q), _p(q_lo), _p(t), _p(t_lo), nq, nt, d, float(temp), prec, _p(out),
                                          out.stride(0), _stream()), "avt_sim_gemm_nt")
    return out


# ---- transition select ------------------------------------------------------------
def row_transition(sim, q_ids=None, sim_a=None, alpha=0.5, threshold=0.0, cap=64):
    """validate.py:524-572 for every row -> dict(idx, seg, p, cnt, stats) (device tensors)."""
    _dev(sim, "sim", torch.float32)
    nq, nt = sim.shape
    n_seg = 0
    if q_ids is not None:
        _dev(q_ids, "q_ids", torch.int64)
        n_seg = nt
    if sim_a is not None:
        _dev(sim_a, "sim_a", torch.float32)
        assert sim_a.shape == sim.shape
    dev = sim.device
    idx = torch.full((nq, cap), -1, dtype=torch.int32, device=dev)
    seg = torch.full((nq, cap), -1, dtype=torch.int32, device=dev)
    p = torch.zeros((nq, cap), dtype=torch.float32, device=dev)
    cnt = torch.zeros((nq,), dtype=torch.int32, device=dev)
    stats = torch.zeros((nq, 4), dtype=torch.float32, device=dev)
    _lib.check(_lib.lib().avt_row_transition(_p(sim), nq, nt, sim.stride(0), _p(q_ids), n_seg, _p(sim_a),
                                             sim_a.stride(0) if sim_a is not None else 0, float(alpha),
                                             float(threshold), int(cap), _p(idx), _p(seg), _p(p), _p(cnt), _p(stats),
                                             _stream()), "avt_row_transition")
    return dict(idx=idx, seg=seg, p=p, cnt=cnt, stats=stats)


def row_topk(sim, k, self_col=None):
    _dev(sim, "sim", torch.float32)
    nq, nt = sim.shape
    if self_col is not None:
        _dev(self_col, "self_col", torch.int64)
    idx = torch.empty((nq, k), dtype=torch.int32, device=sim.device)
    val = torch.empty((nq, k), dtype=torch.float32, device=sim.device)
    _lib.check(_lib.lib().avt_row_topk(_p(sim), nq, nt, sim.stride(0), _p(self_col), int(k), _p(idx), _p(val),
                                       _stream()), "avt_row_topk")
    return idx, val


# ---- InfoNCE cross-entropy ------------------------------------------------------------
def softmax_ce_fwd(logits, label=None):
    _dev(logits, "logits", torch.float32)
    b, c = logits.shape
    if label is not None:
        _dev(label, "label", torch.int64)
    loss = torch.empty((b,), dtype=torch.float32, device=logits.device)
    prob = torch.empty((b, c), dtype=torch.float32, device=logits.device)
    _lib.check(_lib.lib().avt_softmax_ce_fwd(_p(logits), b, c, _p(label), _p(loss), _p(prob), _stream()),
               "avt_softmax_ce_fwd")
    return loss, prob


def softmax_ce_bwd(prob, label=None, scale=1.0):
    _dev(prob, "prob", torch.float32)
    b, c = prob.shape
    d = torch.empty_like(prob)
    _lib.check(_lib.lib().avt_softmax_ce_bwd(_p(prob), _p(label), b, c, float(scale), _p(d), _stream()),
               "avt_softmax_ce_bwd")
    return d


# ---- encoder convolutions (implicit GEMM on MFMA) ----------------------------------------
def conv3d_ktab(cin, kernel, h, w, ldi):
    """Host table of per-K-chunk tap offsets for avt_conv3d_igemm_bf16 (int32 [n_entries, 2])."""
    kt, kh, kw = kernel
    n_entries = 8 * ((kt * kh * kw * cin + 63) // 64) + 2  # + the 16 zero bytes the kernel reads for OOB chunks
    tab = np.empty((n_entries, 2), np.int32)
    _lib.check(_lib.lib().avt_conv3d_ktab(int(cin), kt, kh, kw, int(h), int(w), int(ldi), tab.ctypes.data, n_entries),
               "avt_conv3d_ktab")
    return tab


def conv3d_wfrag_supported(cin, cout, kernel):
    return bool(_lib.lib().avt_conv3d_igemm_wfrag_supported(int(cin), int(cout), *[int(k) for k in kernel]))


def conv3d_igemm(x_ptr, wt, bias, res_ptr, out_ptr, ktab, dims, cin, cout, kernel, stride, pad, ldi, ldo, ldr, relu,
                 out_dims=(0, 0, 0), out_rows=None, wfrag=None):
    """Raw launch: x_ptr/res_ptr/out_ptr are device addresses (int) of bf16 NDHWC rows; dims = (B,T,H,W).
    out_rows = (stride, H, W): output position (f, ho, wo) goes to row (f*H + stride*ho)*W + stride*wo of the output
    buffer (avt_conv3d_igemm_rows_bf16)."""
    b, t, h, w = dims
    if wfrag is not None:  # (fragment-order weights [tiles, nup, 2, 64, 8]: the XB tile, weights bypass the LDS)
        _dev(wfrag, "wfrag", torch.bfloat16)
        orr = out_rows if out_rows is not None else (1, 0, 0)
        _lib.check(_lib.lib().avt_conv3d_igemm_wfrag_bf16(C.c_void_p(x_ptr), _p(wt), _p(bias),
                                                          C.c_void_p(res_ptr) if res_ptr else None, C.c_void_p(out_ptr),
                                                          _p(ktab), b, t, h, w, int(cin), int(cout), *kernel, *stride, *pad,
                                                          *out_dims, int(ldi), int(ldo), int(ldr), 1 if relu else 0,
                                                          int(orr[0]), int(orr[1]), int(orr[2]), _p(wfrag),
                                                          int(wfrag.shape[1]), _stream()),
                   "avt_conv3d_igemm_wfrag_bf16")
        return
    if out_rows is not None:
        _lib.check(_lib.lib().avt_conv3d_igemm_rows_bf16(C.c_void_p(x_ptr), _p(wt), _p(bias),
                                                         C.c_void_p(res_ptr) if res_ptr else None, C.c_void_p(out_ptr),
                                                         _p(ktab), b, t, h, w, int(cin), int(cout), *kernel, *stride, *pad,
                                                         *out_dims, int(ldi), int(ldo), int(ldr), 1 if relu else 0,
                                                         int(out_rows[0]), int(out_rows[1]), int(out_rows[2]), _stream()),
                   "avt_conv3d_igemm_rows_bf16")
        return
    _lib.check(_lib.lib().avt_conv3d_igemm_bf16(C.c_void_p(x_ptr), _p(wt), _p(bias),
                                                C.c_void_p(res_ptr) if res_ptr else None, C.c_void_p(out_ptr), _p(ktab),
                                                b, t, h, w, int(cin), int(cout), *kernel, *stride, *pad, *out_dims,
                                                int(ldi), int(ldo), int(ldr), 1 if relu else 0, _stream()),
               "avt_conv3d_igemm_bf16")


def conv3d_igemm_x3(x_ptrs, wt_hi, wt_lo, bias, res_ptrs, out_ptrs, ktab, dims, cin, cout, kernel, stride, pad, ldi, ldo,
                    ldr, relu, plane_dtype, wscale=None, out_dims=(0, 0, 0), out_rows=None, wblk=False):
    """Contract-grade (split-plane) convolution: x_ptrs / res_ptrs / out_ptrs = (hi, lo) device addresses of the two
    16-bit planes (res_ptrs None = no residual); wt_hi / wt_lo bf16-typed [Cout, K] planes — wblk: in K-blocked order
    [K / 32, Cout, 32] (avt_conv3d_igemm_x3_wblk: layers the XL tile runs); see include/avt.h."""
    b, t, h, w = dims
    _dev(wt_hi, "wt_hi", torch.bfloat16)
    _dev(wt_lo, "wt_lo", torch.bfloat16)
    orr = out_rows if out_rows is not None else (1, 0, 0)
    rh, rl = res_ptrs if res_ptrs is not None else (0, 0)
    fn = _lib.lib().avt_conv3d_igemm_x3_wblk if wblk else _lib.lib().avt_conv3d_igemm_x3
    _lib.check(fn(C.c_void_p(x_ptrs[0]), C.c_void_p(x_ptrs[1]), _p(wt_hi), _p(wt_lo), _p(bias),
                                              C.c_void_p(rh) if rh else None, C.c_void_p(rl) if rl else None,
                                              C.c_void_p(out_ptrs[0]), C.c_void_p(out_ptrs[1]), _p(ktab), b, t, h, w,
                                              int(cin), int(cout), *kernel, *stride, *pad, *out_dims, int(ldi), int(ldo),
                                              int(ldr), int(relu), int(orr[0]), int(orr[1]), int(orr[2]),
                                              int(plane_dtype), _p(wscale), _stream()), "avt_conv3d_igemm_x3_wblk" if wblk else "avt_conv3d_igemm_x3")


def conv3d_igemm_x3_f32(x, wt_hi, wt_lo, wscale, out, ktab, dims, cin, cout, kernel, stride, pad, ldi, ldo, plane_dtype, add=None):
    """fp32 rows in / fp32 rows out on the split-plane kernel (training forward / stride-1 dgrad); add = fp32 rows [M, cout]
    summed into the result; see include/avt.h."""
    b, t, h, w = dims
    _dev(x, "x", torch.float32)
    _dev(out, "out", torch.float32)
    _dev(wt_hi, "wt_hi", torch.bfloat16)
    _dev(wt_lo, "wt_lo", torch.bfloat16)
    if add is not None:
        _dev(add, "add", torch.float32)
    _lib.check(_lib.lib().avt_conv3d_igemm_x3_f32(_p(x), _p(wt_hi), _p(wt_lo), _p(wscale), _p(add), _p(out), _p(ktab), int(b), int(t),
                                                  int(h), int(w), int(cin), int(cout), *[int(k) for k in kernel],
                                                  *[int(v) for v in stride], *[int(v) for v in pad], int(ldi), int(ldo),
                                                  int(cout), int(plane_dtype), _stream()), "avt_conv3d_igemm_x3_f32")


def conv3d_igemm_x3_f32_stats(x, wt_hi, wt_lo, wscale, out, ktab, dims, cin, cout, kernel, stride, pad, ldi, ldo, plane_dtype, groups, stat_c):
    """conv3d_igemm_x3_f32 that also leaves the train-mode BatchNorm statistics of its output behind (include/avt.h): -> (ws, pre_rows),
    the BatchNorm workspace with the per-tile partial sums at its head, for avt_bn_train_fwd_pre."""
    b, t, h, w = dims
    _dev(x, "x", torch.float32)
    _dev(out, "out", torch.float32)
    _dev(wt_hi, "wt_hi", torch.bfloat16)
    _dev(wt_lo, "wt_lo", torch.bfloat16)
    k = int(kernel[0]) * int(kernel[1]) * int(kernel[2]) * int(cin)
    m = out.numel() // int(cout)
    rows = _lib.lib().avt_conv3d_igemm_x3_f32_stat_rows(int(cout), k, int(m), int(groups))
    if rows <= 0:
        raise _lib.AvtError("conv3d_igemm_x3_f32_stats: %d rows do not split into %d groups" % (m, groups))
    pre_rows = rows * max(1, int(stat_c) // 1024)
    ws = torch.empty(_lib.lib().avt_bn_train_ws_bytes_pre(int(stat_c), int(groups), pre_rows), dtype=torch.uint8, device=x.device)
    _lib.check(_lib.lib().avt_conv3d_igemm_x3_f32_stats(_p(x), _p(wt_hi), _p(wt_lo), _p(wscale), _p(out), _p(ktab), int(b), int(t), int(h),
                                                        int(w), int(cin), int(cout), *[int(v) for v in kernel], *[int(v) for v in stride],
                                                        *[int(v) for v in pad], int(ldi), int(ldo), int(plane_dtype), _p(ws), int(groups),
                                                        int(stat_c), _stream()), "avt_conv3d_igemm_x3_f32_stats")
    return ws, pre_rows


def pw_x3_f32_supported(k, n):
    return bool(_lib.lib().avt_pw_x3_f32_supported(int(k), int(n)))


def pw_x3_f32(x, k, wt_hi, wt_lo, wscale, out, n, plane_dtype, add=None):
    """A 1x1x1 / stride 1 convolution on fp32 channels-last rows in the streaming form (csrc/pw_x3.hip, training): x [.., k] -> out [.., n],
    wt_hi / wt_lo the plain [n, k] planes of weight_planes_f32 / weight_planes_t_f32, add = fp32 rows summed into the result."""
    _dev(x, "x", torch.float32)
    _dev(out, "out", torch.float32)
    _dev(wt_hi, "wt_hi", torch.bfloat16)
    _dev(wt_lo, "wt_lo", torch.bfloat16)
    if add is not None:
        _dev(add, "add", torch.float32)
    m = x.numel() // k
    _lib.check(_lib.lib().avt_pw_x3_f32(_p(x), int(k), int(k), _p(wt_hi), _p(wt_lo), _p(wscale), _p(add), int(n), _p(out), int(n), int(n),
                                        int(m), int(plane_dtype), _stream()), "avt_pw_x3_f32")


def conv3d_igemm_x3_f32_bwdstats(dy, wt_hi, wt_lo, out, ktab, dims, cin, cout, kernel, pad, plane_dtype, bn, groups, stat_c, add=None):
    """The stride-1 input gradient (a convolution of dy [.., cin] with the transposed filter -> out [.., cout]) that is the OUTPUT
    gradient of a train-mode BatchNorm: out = mask * (conv + add), and the BatchNorm's backward statistics as per-tile partials
    (include/avt.h).  bn = (x, save_mean, save_invstd, gamma, beta | None, mask | None, relu).  -> (ws, pre_rows) for
    avt_bn_train_bwd_pre, or None where the kernel does not apply (the 256 x 256 tile's layers): nothing was launched."""
    b, t, h, w = dims
    _dev(dy, "dy", torch.float32)
    _dev(out, "out", torch.float32)
    k = int(kernel[0]) * int(kernel[1]) * int(kernel[2]) * int(cin)
    m = out.numel() // int(cout)
    rows = _lib.lib().avt_conv3d_igemm_x3_f32_bwdstats_rows(int(cout), k, int(m), int(groups))
    if rows <= 0:
        return None
    bx, mean, invstd, gamma, beta, mask, relu = bn
    pre_rows = rows * max(1, int(stat_c) // 1024)
    ws = torch.empty(_lib.lib().avt_bn_train_ws_bytes_pre(int(stat_c), int(groups), pre_rows), dtype=torch.uint8, device=dy.device)
    _lib.check(_lib.lib().avt_conv3d_igemm_x3_f32_bwdstats(
        _p(dy), _p(wt_hi), _p(wt_lo), None, _p(add), _p(out), _p(ktab), int(b), int(t), int(h), int(w), int(cin), int(cout),
        *[int(v) for v in kernel], *[int(v) for v in pad], int(cin), int(cout), int(cout), int(plane_dtype), _p(bx), _p(mean), _p(invstd),
        _p(gamma), _p(beta), _p(mask), int(relu), _p(ws), int(groups), int(stat_c), _stream()), "avt_conv3d_igemm_x3_f32_bwdstats")
    return ws, pre_rows


def pw_x3_f32_bwdstats(dy, k, wt_hi, wt_lo, out, n, plane_dtype, bn, groups, add=None):
    """The streaming pointwise form of conv3d_igemm_x3_f32_bwdstats: dy [.., k] -> out [.., n] = mask * (dy W + add), the BatchNorm's
    backward statistics as one row of partials per wave and group.  -> (ws, pre_rows), or None outside the kernel's domain."""
    _dev(dy, "dy", torch.float32)
    _dev(out, "out", torch.float32)
    m = dy.numel() // k
    rows = _lib.lib().avt_pw_x3_f32_bwdstats_rows(int(k), int(n), int(m), int(groups))
    if rows <= 0 or plane_dtype != X3_BF16:
        return None
    bx, mean, invstd, gamma, beta, mask, relu = bn
    pre_rows = rows * max(1, int(n) // 1024)
    ws = torch.empty(_lib.lib().avt_bn_train_ws_bytes_pre(int(n), int(groups), pre_rows), dtype=torch.uint8, device=dy.device)
    _lib.check(_lib.lib().avt_pw_x3_f32_bwdstats(_p(dy), int(k), int(k), _p(wt_hi), _p(wt_lo), _p(add), int(n), _p(out), int(n), int(n), int(m),
                                                 int(plane_dtype), _p(bx), _p(mean), _p(invstd), _p(gamma), _p(beta), _p(mask),
                                                 1 if relu else 0, _p(ws), int(groups), _stream()), "avt_pw_x3_f32_bwdstats")
    return ws, pre_rows


def pw_x3_f32_stats(x, k, wt_hi, wt_lo, wscale, out, n, plane_dtype, groups):
    """pw_x3_f32 that also leaves the train-mode BatchNorm statistics of its output behind: -> (ws, pre_rows) as conv3d_igemm_x3_f32_stats,
    or None when the layer / group count is outside the kernel's domain (the caller runs the plain form)."""
    _dev(x, "x", torch.float32)
    _dev(out, "out", torch.float32)
    _dev(wt_hi, "wt_hi", torch.bfloat16)
    _dev(wt_lo, "wt_lo", torch.bfloat16)
    m = x.numel() // k
    rows = _lib.lib().avt_pw_x3_f32_stat_rows(int(k), int(n), int(m), int(groups))
    if rows <= 0 or n & (n - 1):
        return None
    pre_rows = rows * max(1, int(n) // 1024)
    ws = torch.empty(_lib.lib().avt_bn_train_ws_bytes_pre(int(n), int(groups), pre_rows), dtype=torch.uint8, device=x.device)
    _lib.check(_lib.lib().avt_pw_x3_f32_stats(_p(x), int(k), int(k), _p(wt_hi), _p(wt_lo), _p(wscale), _p(out), int(n), int(n), int(m),
                                              int(plane_dtype), _p(ws), int(groups), _stream()), "avt_pw_x3_f32_stats")
    return ws, pre_rows


def conv3d_igemm_x3_f32_ex(x, wt_hi, wt_lo, wscale, out, ktab, dims, cin, cout, kernel, pad, out_dims, ldi, ldo, plane_dtype,
                           out_rows=(1, 0, 0)):
    """conv3d_igemm_x3_f32 at stride 1 with an explicit output extent and the output-row remap (out_rows = (stride, grid h, grid w));
    `out` is the first element the class writes (a view into the full gradient); see include/avt.h."""
    b, t, h, w = dims
    _dev(x, "x", torch.float32)
    _dev(wt_hi, "wt_hi", torch.bfloat16)
    _dev(wt_lo, "wt_lo", torch.bfloat16)
    _lib.check(_lib.lib().avt_conv3d_igemm_x3_f32_ex(_p(x), _p(wt_hi), _p(wt_lo), _p(wscale), C.c_void_p(out.data_ptr()), _p(ktab), int(b),
                                                     int(t), int(h), int(w), int(cin), int(cout), *[int(k) for k in kernel],
                                                     *[int(v) for v in pad], *[int(v) for v in out_dims], int(ldi), int(ldo),
                                                     int(out_rows[0]), int(out_rows[1]), int(out_rows[2]), int(plane_dtype), _stream()),
               "avt_conv3d_igemm_x3_f32_ex")


def weight_planes_f32(w2d, plane_dtype):
    """[cout, k] fp32 rows (a channels-last Conv3d weight viewed as rows) -> (hi, lo, wscale): fp16 planes row-scaled into
    [2^9, 2^10) with wscale = 1 / scale, or unscaled bf16 planes with wscale None (csrc/stem_train.hip)."""
    _dev(w2d, "w", torch.float32)
    cout, k = w2d.shape
    hi = torch.empty((cout, k), dtype=torch.bfloat16, device=w2d.device)
    lo = torch.empty_like(hi)
    ws = torch.empty(cout, dtype=torch.float32, device=w2d.device) if plane_dtype == X3_F16 else None
    _lib.check(_lib.lib().avt_weight_planes_f32(_p(w2d), int(cout), int(k), _p(hi), _p(lo), _p(ws), int(plane_dtype), _stream()),
               "avt_weight_planes_f32")
    return hi, lo, ws


def weight_planes_gather_f32(w, idx_map, plane_dtype):
    """w: a dense fp32 weight (any layout; read through its storage), idx_map int32 [rows, k] of storage offsets (-1: zero) ->
    (hi, lo, wscale | None) like weight_planes_f32 on the gathered rows (csrc/stem_train.hip)."""
    _dev(w, "w", torch.float32, contiguous=False)
    _dev(idx_map, "idx_map", torch.int32)
    rows, k = idx_map.shape
    hi = torch.empty((rows, k), dtype=torch.bfloat16, device=w.device)
    lo = torch.empty_like(hi)
    ws = torch.empty(rows, dtype=torch.float32, device=w.device) if plane_dtype == X3_F16 else None
    _lib.check(_lib.lib().avt_weight_planes_gather_f32(C.c_void_p(w.data_ptr()), _p(idx_map), int(rows), int(k), _p(hi), _p(lo), _p(ws),
                                                       int(plane_dtype), _stream()), "avt_weight_planes_gather_f32")
    return hi, lo, ws


def conv_x3_set_small_tile(on):
    """The IO32 convolutions' 64-row tile at small batches on / off (csrc/conv_x3.hip io32_tile_rows) -> the previous setting."""
    return int(_lib.lib().avt_conv_x3_set_small_tile(int(bool(on))))


def weight_planes_job_bytes():
    return int(_lib.lib().avt_weight_planes_job_bytes())


def weight_planes_multi(jobs, blk2job, nblocks):
    """Every job of a DEVICE table of AvtPlaneJob (include/avt.h) in one launch: the planes of all of a step's weights, in place
    (train_ops._refresh_planes builds the table from the single launches' own arguments)."""
    assert jobs.is_cuda and blk2job.is_cuda and blk2job.dtype == torch.int32 and blk2job.numel() == nblocks
    _lib.check(_lib.lib().avt_weight_planes_multi(_p(jobs), _p(blk2job), int(nblocks), _stream()), "avt_weight_planes_multi")


def weight_planes_t_f32(w3d, sel):
    """[cout, taps, cin] fp32 -> (hi, lo) bf16 planes [cin, len(sel) * cout] with out[ci][a][co] = w[co][sel[a]][ci]: the input
    gradient's filter (sel = all taps reversed) or one residue class of a strided layer's (csrc/stem_train.hip)."""
    _dev(w3d, "w", torch.float32)
    cout, taps, cin = w3d.shape
    hi = torch.empty((cin, len(sel) * cout), dtype=torch.bfloat16, device=w3d.device)
    lo = torch.empty_like(hi)
    arr = (C.c_int32 * len(sel))(*[int(v) for v in sel])
    _lib.check(_lib.lib().avt_weight_planes_t_f32(_p(w3d), int(cout), int(taps), int(cin), arr, len(sel), _p(hi), _p(lo), _stream()),
               "avt_weight_planes_t_f32")
    return hi, lo


def conv3d_wgrad_x3_f32(dy, x, dw, dims, cin, cout, kernel, stride, pad, ldx, ldy):
    """dw [cout, taps, cin] fp32 = weight gradient of the convolution (csrc/wgrad_x3.hip); dy / x fp32 NDHWC rows; dims = x's (B,T,H,W)."""
    b, t, h, w = dims
    _dev(dy, "dy", torch.float32)
    _dev(x, "x", torch.float32)
    _dev(dw, "dw", torch.float32)
    _lib.check(_lib.lib().avt_conv3d_wgrad_x3_f32(_p(dy), _p(x), _p(dw), int(b), int(t), int(h), int(w), int(cin), int(cout),
                                                  *[int(k) for k in kernel], *[int(v) for v in stride], *[int(v) for v in pad],
                                                  int(ldx), int(ldy), _stream()), "avt_conv3d_wgrad_x3_f32")


def conv3d_wgrad_x3_sub_f32(dy, x, dw, dims, cin, cout, kernel, stride, pad, out_dims, ldx, ldy, ldw, zero_dw):
    """The general weight-gradient entry (4-channel inputs, up to 49 taps, explicit output extent, any padding offset, dW a
    column slice of a longer filter): the SlowFast stems; see include/avt.h.  dy / x / dw are device tensors (dw may be a view)."""
    b, t, h, w = dims
    _dev(dy, "dy", torch.float32)
    _dev(x, "x", torch.float32)
    _lib.check(_lib.lib().avt_conv3d_wgrad_x3_sub_f32(_p(dy), _p(x), C.c_void_p(dw.data_ptr()), int(b), int(t), int(h), int(w),
                                                      int(cin), int(cout), *[int(k) for k in kernel], *[int(v) for v in stride],
                                                      *[int(v) for v in pad], *[int(v) for v in out_dims], int(ldx), int(ldy),
                                                      int(ldw), int(bool(zero_dw)), _stream()), "avt_conv3d_wgrad_x3_sub_f32")


def stem_conv_x3(x_ptrs, wt_hi, wt_lo, bias, wscale, out_ptrs, batch, t, h, pw, cout, kt, st, pt, plane_dtype, relu=True,
                 frames_per_tile=0, frame_idx=None, table_frames=0):
    """stem_conv on plane pairs (contract-grade mode); wt_hi / wt_lo = fused_slowfast.stem_lds_image of each weight plane
    (frames_per_tile = 2: its frame-major form for the 4-frame x 8-channel time-grouped stem).  frame_idx (int32 [batch * t],
    device): x_ptrs is a table of `table_frames` distinct frames and clip b's frame t is table frame frame_idx[b * t_total + t]."""
    _dev(wt_hi, "wt_hi", torch.bfloat16)
    _dev(wt_lo, "wt_lo", torch.bfloat16)
    if frame_idx is not None:
        _dev(frame_idx, "frame_idx", torch.int32)
        if frame_idx.numel() != batch * t:
            raise _lib.AvtError("stem_conv_x3: frame_idx has %d entries, expected batch * t = %d" % (frame_idx.numel(), batch * t))
    _lib.check(_lib.lib().avt_stem_conv_x3(C.c_void_p(x_ptrs[0]), C.c_void_p(x_ptrs[1]), _p(wt_hi), _p(wt_lo), _p(bias),
                                           _p(wscale), C.c_void_p(out_ptrs[0]), C.c_void_p(out_ptrs[1]), int(batch), int(t),
                                           int(h), int(pw), int(cout), int(kt), int(st), int(pt), int(bool(relu)),
                                           int(plane_dtype), int(frames_per_tile), _p(frame_idx), int(table_frames), _stream()),
               "avt_stem_conv_x3")


def lateral_x3_supported(cin, cout, kt):
    return bool(_lib.lib().avt_lateral_x3_supported(int(cin), int(cout), int(kt)))


def lateral_x3(x_ptrs, ldx, cin, w_hi, w_lo, bias, wscale, y_ptrs, ldy, cout, batch, t, hw, kt, st, pt, relu, plane_dtype):
    """Conv3d [kt,1,1] stride (st,1,1) pad (pt,0,0) + bias (+ ReLU) on plane pairs as one streaming pass (csrc/pw_x3.hip's
    temporal-tap form): SlowFast's lateral connections; w_* = fused_slowfast.pack_pw_planes of the conv's [cout -> 32 k, kt * cin] rows."""
    _dev(w_hi, "w_hi", torch.bfloat16)
    _dev(w_lo, "w_lo", torch.bfloat16)
    _lib.check(_lib.lib().avt_lateral_x3(C.c_void_p(x_ptrs[0]), C.c_void_p(x_ptrs[1]), int(ldx), int(cin), _p(w_hi), _p(w_lo), _p(bias),
                                         _p(wscale), C.c_void_p(y_ptrs[0]), C.c_void_p(y_ptrs[1]), int(ldy), int(cout), int(batch),
                                         int(t), int(hw), int(kt), int(st), int(pt), int(relu), int(plane_dtype), _stream()),
               "avt_lateral_x3")


def stem_conv_x3_merged(x_ptrs, wt_hi, wt_lo, bias, wscale, out_ptrs, batch, t, h, pw, cout, kt, st, pt, plane_dtype, tap_frames,
                        tap_tiles, ktm, table_frames, relu=True):
    """The time-grouped fast stem over a frame table with the taps of one source frame merged (include/avt.h)."""
    _dev(wt_hi, "wt_hi", torch.bfloat16)
    _dev(wt_lo, "wt_lo", torch.bfloat16)
    _dev(tap_frames, "tap_frames", torch.int32)
    _dev(tap_tiles, "tap_tiles", torch.int32)
    _lib.check(_lib.lib().avt_stem_conv_x3_merged(C.c_void_p(x_ptrs[0]), C.c_void_p(x_ptrs[1]), _p(wt_hi), _p(wt_lo), _p(bias),
                                                  _p(wscale), C.c_void_p(out_ptrs[0]), C.c_void_p(out_ptrs[1]), int(batch), int(t),
                                                  int(h), int(pw), int(cout), int(kt), int(st), int(pt), int(bool(relu)),
                                                  int(plane_dtype), _p(tap_frames), _p(tap_tiles), int(ktm), int(table_frames),
                                                  _stream()), "avt_stem_conv_x3_merged")


def clip_planes_f32(x, plane_dtype):
    """[B, 3, T, H, W] fp32 device tensor (any strides) -> (hi, lo) planes [B, T, H, W, 4] typed bfloat16 (raw bits), the 4th
    channel zero: what the patch-resident stem kernels read as pixel pairs [B, T, H, W/2, 8]."""
    _dev(x, "x", torch.float32, contiguous=False)
    if x.dim() != 5 or x.shape[1] != 3:
        raise _lib.AvtError("clip_planes_f32: expected [B, 3, T, H, W], got %s" % (tuple(x.shape),))
    b, _, t, h, w = x.shape
    hi = torch.empty((b, t, h, w, 4), dtype=torch.bfloat16, device=x.device)
    lo = torch.empty_like(hi)
    sb, sc, st, sh, sw = x.stride()
    _lib.check(_lib.lib().avt_clip_planes_f32(_p(x), b, t, h, w, sb, sc, st, sh, sw, _p(hi), _p(lo), int(plane_dtype), _stream()),
               "avt_clip_planes_f32")
    return hi, lo


def stem_conv_x3_f32(x_hi, x_lo, wt_hi, wt_lo, wscale, out, batch, t, h, pw, cout, kt, st, pt, tgroup, plane_dtype, frames_per_tile=0):
    """The stems' training forward: stem_conv_x3 with fp32 NDHWC output [batch, to*tgroup, h/2, pw, cout/tgroup], no bias / ReLU."""
    _dev(wt_hi, "wt_hi", torch.bfloat16)
    _dev(wt_lo, "wt_lo", torch.bfloat16)
    _dev(out, "out", torch.float32, contiguous=False)
    _lib.check(_lib.lib().avt_stem_conv_x3_f32(_p(x_hi), _p(x_lo), _p(wt_hi), _p(wt_lo), _p(wscale), _p(out), int(batch), int(t),
                                               int(h), int(pw), int(cout), int(kt), int(st), int(pt), int(tgroup),
                                               int(plane_dtype), int(frames_per_tile), _stream()), "avt_stem_conv_x3_f32")


def stem_wgrad_x3_supported(h, pw, cout, kt):
    return bool(_lib.lib().avt_stem_wgrad_x3_supported(int(h), int(pw), int(cout), int(kt)))


def stem_wgrad_x3(x_hi, x_lo, dy, batch, t, h, pw, cout, kt, pt):
    """The stems' weight gradient in the pixel-pair form -> dw [cout, kt, 7, 4, 8] fp32 (see include/avt.h); x_* = bf16 planes
    of clip_planes_f32, dy = fp32 NDHWC rows [batch, to, h/2, pw, cout] (dense)."""
    _dev(dy, "dy", torch.float32, contiguous=False)
    dw = torch.zeros((cout, kt, 7, 4, 8), dtype=torch.float32, device=dy.device)
    _lib.check(_lib.lib().avt_stem_wgrad_x3(_p(x_hi), _p(x_lo), _p(dy), _p(dw), int(batch), int(t), int(h), int(pw), int(cout),
                                            int(kt), int(pt), _stream()), "avt_stem_wgrad_x3")
    return dw


def pw_x3_supported(k, n):
    return bool(_lib.lib().avt_pw_x3_supported(int(k), int(n)))


def pw_x3(x_ptrs, ldx, k, w_hi, w_lo, bias, wscale, res_ptrs, ldr, y_ptrs, ldy, n, m, relu, plane_dtype):
    """Streaming pointwise layer on plane pairs (csrc/pw_x3.hip); w_hi / w_lo = fused_slowfast.pack_pw_planes(...)."""
    _dev(w_hi, "w_hi", torch.bfloat16)
    _dev(w_lo, "w_lo", torch.bfloat16)
    rh, rl = res_ptrs if res_ptrs is not None else (0, 0)
    _lib.check(_lib.lib().avt_pw_x3(C.c_void_p(x_ptrs[0]), C.c_void_p(x_ptrs[1]), int(ldx), int(k), _p(w_hi), _p(w_lo),
                                    _p(bias), _p(wscale), C.c_void_p(rh) if rh else None, C.c_void_p(rl) if rl else None,
                                    int(ldr), C.c_void_p(y_ptrs[0]), C.c_void_p(y_ptrs[1]), int(ldy), int(n), int(m),
                                    1 if relu else 0, int(plane_dtype), _stream()), "avt_pw_x3")


def pw_chain_x3_supported(k1, n1, n2):
    return bool(_lib.lib().avt_pw_chain_x3_supported(int(k1), int(n1), int(n2)))


def pw_chain_x3(x_ptrs, ldx, k1, w1, bias1, wscale1, res_ptrs, ldr, y_ptrs, ldy, n1, relu1, w2, bias2, wscale2, z_ptrs, ldz, n2, m,
                plane_dtype):
    """y = act(W1 x + b1 [+ res]), z = relu(W2 y + b2) in one pass on plane pairs (csrc/pw_x3.hip); w1 / w2 = (hi, lo) fragment
    planes (fused_slowfast.pack_pw_planes)."""
    rh, rl = res_ptrs if res_ptrs is not None else (0, 0)
    _lib.check(_lib.lib().avt_pw_chain_x3(C.c_void_p(x_ptrs[0]), C.c_void_p(x_ptrs[1]), int(ldx), int(k1), _p(w1[0]), _p(w1[1]),
                                          _p(bias1), _p(wscale1), C.c_void_p(rh) if rh else None, C.c_void_p(rl) if rl else None,
                                          int(ldr), C.c_void_p(y_ptrs[0]), C.c_void_p(y_ptrs[1]), int(ldy), int(n1),
                                          1 if relu1 else 0, _p(w2[0]), _p(w2[1]), _p(bias2), _p(wscale2), C.c_void_p(z_ptrs[0]),
                                          C.c_void_p(z_ptrs[1]), int(ldz), int(n2), int(m), int(plane_dtype), _stream()),
               "avt_pw_chain_x3")


def maxpool_hw3s2_x3(x_ptrs, out_ptrs, bt, h, w, c, ldi, ldo, plane_dtype, tgroup=1, frame_idx=None):
    """frame_idx (int32 [bt], device): output frame b pools input frame frame_idx[b] (a table of distinct frames)."""
    if frame_idx is not None:
        _dev(frame_idx, "frame_idx", torch.int32)
        if frame_idx.numel() != bt:
            raise _lib.AvtError("maxpool_hw3s2_x3: frame_idx has %d entries, expected %d" % (frame_idx.numel(), bt))
    _lib.check(_lib.lib().avt_maxpool_hw3s2_ndhwc_x3(C.c_void_p(x_ptrs[0]), C.c_void_p(x_ptrs[1]), C.c_void_p(out_ptrs[0]),
                                                     C.c_void_p(out_ptrs[1]), int(bt), int(h), int(w), int(c), int(ldi),
                                                     int(ldo), int(tgroup), int(plane_dtype), _p(frame_idx), _stream()),
               "avt_maxpool_hw3s2_ndhwc_x3")


def maxpool_hw2s2_x3(x_ptrs, out_ptrs, bt, h, w, c, ldi, ldo, plane_dtype):
    """MaxPool2d(2, 2), floor mode, on plane pairs (NHWC rows): the contract-grade VGGish."""
    _lib.check(_lib.lib().avt_maxpool_hw2s2_ndhwc_x3(C.c_void_p(x_ptrs[0]), C.c_void_p(x_ptrs[1]), C.c_void_p(out_ptrs[0]),
                                                     C.c_void_p(out_ptrs[1]), int(bt), int(h), int(w), int(c), int(ldi),
                                                     int(ldo), int(plane_dtype), _stream()), "avt_maxpool_hw2s2_ndhwc_x3")


def mean_positions_x3(x_ptrs, batch, p, c, ldi, out, col0, plane_dtype):
    _dev(out, "out", torch.float32)
    _lib.check(_lib.lib().avt_mean_positions_x3(C.c_void_p(x_ptrs[0]), C.c_void_p(x_ptrs[1]), int(batch), int(p), int(c),
                                                int(ldi), C.c_void_p(out.data_ptr() + 4 * int(col0)), int(out.shape[1]),
                                                int(plane_dtype), _stream()), "avt_mean_positions_x3")


def maxpool_hw3s2(x_ptr, out_ptr, bt, h, w, c, ldi, ldo, tgroup=1):
    """MaxPool3d((1,3,3),(1,2,2),(0,1,1)) on NDHWC bf16 rows (raw device addresses)."""
    _lib.check(_lib.lib().avt_maxpool_hw3s2_ndhwc_bf16(C.c_void_p(x_ptr), C.c_void_p(out_ptr), int(bt), int(h), int(w),
                                                       int(c), int(ldi), int(ldo), int(tgroup), _stream()),
               "avt_maxpool_hw3s2_ndhwc_bf16")


def stem_conv_supported(h, pw, cout):
    return bool(_lib.lib().avt_stem_conv_supported(int(h), int(pw), int(cout)))


def stem_conv(x_ptr, wt, bias, out_ptr, batch, t, h, pw, cout, kt, st, pt, relu=True):
    """SlowFast stem in pixel-pair form, input patch resident in LDS (csrc/stem_conv.hip); raw device addresses for the
    activations, wt = fused_slowfast.stem_lds_image(packed weights) bf16, bias [cout] fp32."""
    _dev(wt, "wt", torch.bfloat16)
    _dev(bias, "bias", torch.float32)
    _lib.check(_lib.lib().avt_stem_conv_bf16(C.c_void_p(x_ptr), _p(wt), _p(bias), C.c_void_p(out_ptr), int(batch), int(t),
                                             int(h), int(pw), int(cout), int(kt), int(st), int(pt), int(bool(relu)),
                                             _stream()), "avt_stem_conv_bf16")


def stem_conv_pool(x_ptr, wt, bias, out_ptr, batch, t, h, pw, cout, kt, st, pt, tgroup, ldo):
    """stem_conv + ReLU + MaxPool3d((1,3,3),(1,2,2),(0,1,1)) in one kernel; out = pooled rows (stride ldo elements)."""
    _dev(wt, "wt", torch.bfloat16)
    _dev(bias, "bias", torch.float32)
    _lib.check(_lib.lib().avt_stem_conv_pool_bf16(C.c_void_p(x_ptr), _p(wt), _p(bias), C.c_void_p(out_ptr), int(batch),
                                                  int(t), int(h), int(pw), int(cout), int(kt), int(st), int(pt),
                                                  int(tgroup), int(ldo), _stream()), "avt_stem_conv_pool_bf16")


def bottleneck_fused_supported(c, w):
    return bool(_lib.lib().avt_bottleneck_fused_supported(int(c), int(w)))


def bottleneck_fused(x_ptr, out_ptr, packed, batch, t, h, w, c, tchunk=8):
    """One fast-pathway identity bottleneck (a [3,1,1] -> b [1,3,3] -> c [1,1,1] + x, ReLUs, BN folded) in one kernel;
    packed = fused_slowfast.pack_bottleneck(...) = (wa, ba, wb, bb, wc, bc) device tensors; raw activation addresses."""
    wa, ba, wb, bb, wc, bc = packed
    _lib.check(_lib.lib().avt_bottleneck_fused_bf16(C.c_void_p(x_ptr), C.c_void_p(out_ptr), _p(wa), _p(ba), _p(wb), _p(bb),
                                                    _p(wc), _p(bc), int(batch), int(t), int(h), int(w), int(c),
                                                    int(tchunk), _stream()), "avt_bottleneck_fused_bf16")


def conv3d_igemm_x3_xl_picked(cout, k, m):
    return bool(_lib.lib().avt_conv3d_igemm_x3_xl_picked(int(cout), int(k), int(m)))


def conv33_x3_supported(cin, cout):
    return bool(_lib.lib().avt_conv33_x3_supported(int(cin), int(cout)))


def conv33_x3(x_ptrs, packed, out_ptrs, batch, t, h, w, ldi, ldo, plane_dtype, relu=True):
    """[1,3,3] 64 -> 64 stride-1 conv + BN + ReLU on plane pairs, activations as direct MFMA operands (csrc/conv33_x3.hip);
    packed = fused_slowfast.pack_c33_x3(...) = (wfrag, coef)."""
    wfrag, coef = packed
    _dev(wfrag, "wfrag", torch.bfloat16)
    _dev(coef, "coef", torch.float32)
    _lib.check(_lib.lib().avt_conv33_x3(C.c_void_p(x_ptrs[0]), C.c_void_p(x_ptrs[1]), _p(wfrag), _p(coef), C.c_void_p(out_ptrs[0]),
                                        C.c_void_p(out_ptrs[1]), int(batch), int(t), int(h), int(w), int(ldi), int(ldo),
                                        int(bool(relu)), int(plane_dtype), _stream()), "avt_conv33_x3")


def res2_x3_supported(c, cm, w):
    return bool(_lib.lib().avt_res2_x3_supported(int(c), int(cm), int(w)))


def res2_x3(x_ptrs, ldi, out_ptrs, ldo, packed, batch, t, h, w, plane_dtype):
    """One identity bottleneck of the slow pathway's res2 stage (256 -> 64 -> 64 -> 256 at width 56) on plane pairs in ONE kernel
    (csrc/res2_x3.hip): the a output in an LDS ring, b's in registers, the weights streamed from L2; packed =
    fused_slowfast.pack_res2_x3(...) = (wfrag, coef); raw plane addresses, row pitches in elements."""
    wfrag, coef = packed
    _dev(wfrag, "wfrag", torch.bfloat16)
    _dev(coef, "coef", torch.float32)
    if wfrag.numel() * 2 != _lib.lib().avt_res2_x3_wfrag_bytes():
        raise _lib.AvtError("res2_x3: wfrag holds %d bytes, the kernel streams %d" % (wfrag.numel() * 2, _lib.lib().avt_res2_x3_wfrag_bytes()))
    _lib.check(_lib.lib().avt_res2_x3(C.c_void_p(x_ptrs[0]), C.c_void_p(x_ptrs[1]), C.c_void_p(out_ptrs[0]), C.c_void_p(out_ptrs[1]),
                                      _p(wfrag), _p(coef), int(batch), int(t), int(h), int(w), int(ldi), int(ldo), int(plane_dtype),
                                      _stream()), "avt_res2_x3")


def bneck_x3_supported(cin, c, w):
    return bool(_lib.lib().avt_bneck_x3_supported(int(cin), int(c), int(w)))


def bneck_x3(x_ptrs, out_ptrs, packed, batch, t, h, w, cin, c, plane_dtype, tchunk=16):
    """One fast-pathway bottleneck (identity, or res2's 8-channel first block with its shortcut conv) on plane pairs in one
    kernel (csrc/bneck_x3.hip); packed = fused_slowfast.pack_bottleneck_x3(...) = (wfrag, coef); raw plane addresses."""
    wfrag, coef = packed
    _dev(wfrag, "wfrag", torch.bfloat16)
    _dev(coef, "coef", torch.float32)
    _lib.check(_lib.lib().avt_bneck_x3(C.c_void_p(x_ptrs[0]), C.c_void_p(x_ptrs[1]), C.c_void_p(out_ptrs[0]),
                                       C.c_void_p(out_ptrs[1]), _p(wfrag), _p(coef), int(batch), int(t), int(h), int(w),
                                       int(cin), int(c), int(tchunk), int(plane_dtype), _stream()), "avt_bneck_x3")


def mean_positions(x_ptr, batch, p, c, ldi, out, col0=0):
    """Head average pool: bf16 rows [batch*p, ldi] (raw address) -> out[:, col0:col0+c] fp32 (device tensor [batch, D])."""
    _dev(out, "out", torch.float32)
    _lib.check(_lib.lib().avt_mean_positions_bf16(C.c_void_p(x_ptr), int(batch), int(p), int(c), int(ldi),
                                                  C.c_void_p(out.data_ptr() + 4 * int(col0)), int(out.shape[1]), _stream()),
               "avt_mean_positions_bf16")


def bottleneck_first_supported(cin, c, w):
    return bool(_lib.lib().avt_bottleneck_first_supported(int(cin), int(c), int(w)))


def bottleneck_first(x_ptr, out_ptr, packed, batch, t, h, w, cin, c, tchunk=8):
    """First fast-pathway block of res2 (8 -> 32 channels, shortcut conv) in one kernel; packed =
    fused_slowfast.pack_bottleneck(..., shortcut=(wsc, bsc)) = (wa, ba, wb, bb, wc, bc, wsc)."""
    wa, ba, wb, bb, wc, bc, wsc = packed
    _lib.check(_lib.lib().avt_bottleneck_first_bf16(C.c_void_p(x_ptr), C.c_void_p(out_ptr), _p(wa), _p(ba), _p(wb), _p(bb),
                                                    _p(wc), _p(wsc), _p(bc), int(batch), int(t), int(h), int(w), int(cin),
                                                    int(c), int(tchunk), _stream()), "avt_bottleneck_first_bf16")


def conv33_c64_supported(cin, cout, w):
    return bool(_lib.lib().avt_conv33_c64_supported(int(cin), int(cout), int(w)))


def conv33_c64(x_ptr, wb, bias, out_ptr, batch, t, h, w, ldo, relu=True):
    """[1,3,3] 64 -> 64 conv with the input strip resident in LDS; wb = fused_slowfast.pack_c33(...)."""
    _dev(wb, "wb", torch.bfloat16)
    _dev(bias, "bias", torch.float32)
    _lib.check(_lib.lib().avt_conv33_c64_bf16(C.c_void_p(x_ptr), _p(wb), _p(bias), C.c_void_p(out_ptr), int(batch), int(t),
                                              int(h), int(w), int(ldo), int(bool(relu)), _stream()), "avt_conv33_c64_bf16")


def pw_chain_supported(k1, n1, n2, has_res, k2x=0):
    return bool(_lib.lib().avt_pw_chain_supported(int(k1), int(n1), int(n2), int(bool(has_res)), int(k2x)))


def pw_chain(x1_ptr, ldx, k1, w1, b1, res_ptr, ldr, y_ptr, ldy, n1, w2, b2, z_ptr, ldz, n2, m, x2_ptr=0, ldx2=0, k2x=0):
    """y = relu(W1 x1 + b1 [+ res]); z = relu(W2 [y | x2] + b2) in one pass (csrc/pw_chain.hip); w1, w2 =
    fused_slowfast.pack_pw(...) fragments; res_ptr 0 = no residual; x2_ptr 0 = the second layer reads y only."""
    _dev(w1, "w1", torch.bfloat16)
    _dev(w2, "w2", torch.bfloat16)
    _dev(b1, "b1", torch.float32)
    _dev(b2, "b2", torch.float32)
    _lib.check(_lib.lib().avt_pw_chain_bf16(C.c_void_p(x1_ptr), int(ldx), int(k1), _p(w1), _p(b1),
                                            C.c_void_p(res_ptr) if res_ptr else None, int(ldr), C.c_void_p(y_ptr), int(ldy),
                                            int(n1), C.c_void_p(x2_ptr) if x2_ptr else None, int(ldx2), int(k2x), _p(w2),
                                            _p(b2), C.c_void_p(z_ptr), int(ldz), int(n2), int(m), _stream()),
               "avt_pw_chain_bf16")


def maxpool_hw2s2(x_ptr, out_ptr, bt, h, w, c, ldi, ldo):
    """MaxPool2d(2, 2), floor mode, on NHWC bf16 rows (raw device addresses) — VGGish (audio_models/vggish.py:15-33)."""
    _lib.check(_lib.lib().avt_maxpool_hw2s2_ndhwc_bf16(C.c_void_p(x_ptr), C.c_void_p(out_ptr), int(bt), int(h), int(w),
                                                       int(c), int(ldi), int(ldo), _stream()),
               "avt_maxpool_hw2s2_ndhwc_bf16")


# ---- classic baseline -----------------------------------------------------------------------
def pairwise_l2(x):
    """x [n, d] fp32 (device) -> D1 [n, n] fp32, D1[i, j] = ||x_i - x_j||_2 (computeD1.py:47-96)."""
    _dev(x, "x", torch.float32)
    n, d = x.shape
    out = torch.empty((n, n), dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().avt_pairwise_l2_f32(_p(x), int(n), int(d), _p(out), _stream()), "avt_pairwise_l2_f32")
    return out


def diag_filter(d1, w):
    """D2[i, j] = sum_k w[k] D1[i+k, j+k] (computeD2.py:21-52) on device matrices."""
    _dev(d1, "d1", torch.float32)
    _dev(w, "w", torch.float32)
    n, fs = d1.shape[0], int(w.numel())
    out = torch.empty((n - fs + 1, n - fs + 1), dtype=torch.float32, device=d1.device)
    _lib.check(_lib.lib().avt_diag_filter_f32(_p(d1), int(n), _p(w), fs, _p(out), _stream()), "avt_diag_filter_f32")
    return out


def q_learning_supported(n):
    return bool(_lib.lib().avt_q_learning_supported(int(n)))


def q_learning(d3, alpha, tol=10e-3, max_iter=1000):
    """The future-cost sweeps of q_learning.py:27-68 on d3 = D2**p (device, square, <= 200 rows) -> (D3_new, sweeps)."""
    _dev(d3, "d3", torch.float32)
    n = d3.shape[0]
    assert d3.shape[1] == n
    out = torch.empty_like(d3)
    iters = torch.zeros(1, dtype=torch.int32, device=d3.device)
    _lib.check(_lib.lib().avt_q_learning_f32(_p(d3), int(n), float(alpha), float(tol), int(max_iter), _p(out), _p(iters),
                                             _stream()), "avt_q_learning_f32")
    return out, iters


# ---- audio front-end ------------------------------------------------------------------------
def logmel(wave, window, melmat, hop, fft_len, log_offset):
    """wave [n] fp32/fp64, window [win] fp64, melmat [fft_len/2+1, n_mel] fp64 (device) -> log-mel [n_frames, n_mel]
    fp64 (mel_features.py:176-205)."""
    if wave.dtype not in (torch.float32, torch.float64):
        raise _lib.AvtError("logmel: waveform must be float32 or float64, got %s" % wave.dtype)
    _dev(wave, "wave", wave.dtype)
    _dev(window, "window", torch.float64)
    _dev(melmat, "melmat", torch.float64)
    win, n_mel = int(window.numel()), int(melmat.shape[1])
    if melmat.shape[0] != fft_len // 2 + 1:
        raise _lib.AvtError("logmel: melmat has %d rows, fft_len/2+1 = %d" % (melmat.shape[0], fft_len // 2 + 1))
    n = int(wave.numel())
    n_frames = 1 + (n - win) // hop if n >= win else 0
    out = torch.empty((n_frames, n_mel), dtype=torch.float64, device=wave.device)
    _lib.check(_lib.lib().avt_logmel_f64(_p(wave), int(wave.dtype == torch.float64), n, _p(window), win, int(hop),
                                         int(fft_len), _p(melmat), n_mel, float(log_offset), _p(out), _stream()),
               "avt_logmel_f64")
    return out


def logmel_examples(lm, ex_len, ex_hop):
    """log-mel [n_frames, n_mel] fp64 -> examples [n_ex, ex_len, n_mel] fp32 (vggish_utils.py:60-68 + the cast of
    validate.py:160-161)."""
    _dev(lm, "logmel", torch.float64)
    n_frames, n_mel = int(lm.shape[0]), int(lm.shape[1])
    n_ex = 1 + (n_frames - ex_len) // ex_hop if n_frames >= ex_len else 0
    out = torch.empty((n_ex, ex_len, n_mel), dtype=torch.float32, device=lm.device)
    _lib.check(_lib.lib().avt_logmel_examples_f32(_p(lm), n_frames, n_mel, int(ex_len), int(ex_hop), _p(out),
                                                  _stream()), "avt_logmel_examples_f32")
    return out


# ---- fused training branch ----------------------------------------------------------------
def infonce_fwd(q, t, temp, eps=1e-12):
    """q [b,d], t [b,n,d] fp32 -> (logits [b,n], inv_q [b], inv_t [b,n])."""
    _dev(q, "q", torch.float32)
    _dev(t, "t", torch.float32)
    b, n, d = t.shape
    logits = torch.empty((b, n), dtype=torch.float32, device=q.device)
    inv_q = torch.empty((b,), dtype=torch.float32, device=q.device)
    inv_t = torch.empty((b, n), dtype=torch.float32, device=q.device)
    _lib.check(_lib.lib().avt_infonce_fwd(_p(q), _p(t), b, n, d, float(temp), float(eps), _p(logits), _p(inv_q), _p(inv_t),
                                          _stream()), "avt_infonce_fwd")
    return logits, inv_q, inv_t


def infonce_bwd(q, t, logits, dlogits, inv_q, inv_t, temp):
    b, n, d = t.shape
    dq = torch.empty_like(q)
    dt = torch.empty_like(t)
    _lib.check(_lib.lib().avt_infonce_bwd(_p(q), _p(t), _p(logits), _p(_dev(dlogits.contiguous(), "dlogits", torch.float32)),
                                          _p(inv_q), _p(inv_t), b, n, d, float(temp), _p(dq), _p(dt), _stream()),
               "avt_infonce_bwd")
    return dq, dt


# ---------------------------------------------------------------- SuperSloMo interpolation passes (csrc/interp.hip)
def _mean3(mean):
    return (C.c_float * 3)(*[float(m) for m in mean])


def interp_pack_pair(frame0, frame1, mean, img, x_ptrs, plane_dtype):
    """Two uint8 [H,W,3] device frames -> img [2,H,W,4] fp32 (x / 255 - mean) and flowComp's input planes [H*W, 8]."""
    _dev(frame0, "frame0", torch.uint8)
    _dev(frame1, "frame1", torch.uint8)
    _dev(img, "img", torch.float32)
    h, w = frame0.shape[0], frame0.shape[1]
    _lib.check(_lib.lib().avt_interp_pack_pair_u8(_p(frame0), _p(frame1), int(h), int(w), _mean3(mean), _p(img), C.c_void_p(x_ptrs[0]),
                                                  C.c_void_p(x_ptrs[1]), int(plane_dtype), _stream()), "avt_interp_pack_pair_u8")


def avgpool2_x3(x_ptrs, dims, c, ldi, y_ptrs, ldo, plane_dtype):
    b, h, w = dims
    _lib.check(_lib.lib().avt_avgpool2_x3(C.c_void_p(x_ptrs[0]), C.c_void_p(x_ptrs[1]), int(b), int(h), int(w), int(c), int(ldi),
                                          C.c_void_p(y_ptrs[0]), C.c_void_p(y_ptrs[1]), int(ldo), int(plane_dtype), _stream()),
               "avt_avgpool2_x3")


def upsample2_bilinear_x3(x_ptrs, dims, c, ldi, y_ptrs, ldo, plane_dtype):
    b, h, w = dims
    _lib.check(_lib.lib().avt_upsample2_bilinear_x3(C.c_void_p(x_ptrs[0]), C.c_void_p(x_ptrs[1]), int(b), int(h), int(w), int(c),
                                                    int(ldi), C.c_void_p(y_ptrs[0]), C.c_void_p(y_ptrs[1]), int(ldo), int(plane_dtype),
                                                    _stream()), "avt_upsample2_bilinear_x3")


def interp_mid_input(img, flow_ptrs, h, w, sf, x_ptrs, ft, plane_dtype):
    _dev(img, "img", torch.float32)
    _dev(ft, "ft", torch.float32)
    _lib.check(_lib.lib().avt_interp_mid_input(_p(img), C.c_void_p(flow_ptrs[0]), C.c_void_p(flow_ptrs[1]), int(h), int(w), int(sf),
                                               C.c_void_p(x_ptrs[0]), C.c_void_p(x_ptrs[1]), _p(ft), int(plane_dtype), _stream()),
               "avt_interp_mid_input")


def interp_final(img, ft, o_ptrs, h, w, sf, mean, out, plane_dtype):
    _dev(img, "img", torch.float32)
    _dev(ft, "ft", torch.float32)
    _dev(out, "out", torch.uint8)
    _lib.check(_lib.lib().avt_interp_final_u8(_p(img), _p(ft), C.c_void_p(o_ptrs[0]), C.c_void_p(o_ptrs[1]), int(h), int(w), int(sf),
                                              _mean3(mean), _p(out), int(plane_dtype), _stream()), "avt_interp_final_u8")
