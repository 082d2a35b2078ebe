"""ctypes binding of oracle/avt_oracle.c (the C half of the CPU oracle).

TEST INFRASTRUCTURE ONLY — see the header of avt_oracle.c.  Imported by
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never by the
product package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libavt_oracle.so")


def build(force=False):
    src = os.path.join(_HERE, "avt_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _SO


def _load():
    build()
    try:
        return C.CDLL(_SO)
    except OSError:
        build(force=True)
        return C.CDLL(_SO)


_lib = _load()
_f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
_u16p = np.ctypeslib.ndpointer(np.uint16, flags="C_CONTIGUOUS")


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def threads():
    return int(_lib.avt_oracle_threads())


def set_threads(n):
    """OpenMP threads of the C oracle (bench.py's cpu_baseline picks the fastest count for the host)."""
    _lib.avt_oracle_set_threads(int(n))


def l2norm_rows(x0, x1=None, eps=1e-12, want_split=True):
    """-> (y_f32, y_hi(uint16 bf16 bits), y_lo)"""
    x0 = np.ascontiguousarray(x0, np.float32)
    n, d0 = x0.shape
    d1 = 0
    if x1 is not None:
        x1 = np.ascontiguousarray(x1, np.float32)
        d1 = x1.shape[1]
    y = np.empty((n, d0 + d1), np.float32)
    hi = np.empty((n, d0 + d1), np.uint16) if want_split else None
    lo = np.empty((n, d0 + d1), np.uint16) if want_split else None
    _lib.avt_oracle_l2norm_rows(_ptr(x0), C.c_int(d0), _ptr(x1), C.c_int(d1), C.c_int64(n),
                                C.c_float(eps), _ptr(y), _ptr(hi), _ptr(lo))
    return y, hi, lo


def sim_f32(q, t, temp, naive=False):
    q = np.ascontiguousarray(q, np.float32)
    t = np.ascontiguousarray(t, np.float32)
    nq, d = q.shape
    nt = t.shape[0]
    assert t.shape[1] == d
    out = np.empty((nq, nt), np.float32)
    fn = _lib.avt_oracle_sim_f32_naive if naive else _lib.avt_oracle_sim_f32
    fn(_ptr(q), _ptr(t), C.c_int64(nq), C.c_int64(nt), C.c_int(d), C.c_float(temp), _ptr(out),
       C.c_int64(nt))
    return out


def sim_bf16(qh, ql, th, tl, temp, x3):
    nq, d = qh.shape
    nt = th.shape[0]
    out = np.empty((nq, nt), np.float32)
    _lib.avt_oracle_sim_bf16(_ptr(qh), _ptr(ql), _ptr(th), _ptr(tl), C.c_int64(nq), C.c_int64(nt),
                             C.c_int(d), C.c_float(temp), C.c_int(1 if x3 else 0), _ptr(out),
                             C.c_int64(nt))
    return out


def target_order(q, n_seg):
    ids = np.empty(n_seg, np.int64)
    ln = C.c_int64(0)
    _lib.avt_oracle_target_order(C.c_int64(q), C.c_int64(n_seg), _ptr(ids), C.byref(ln))
    return ids[: ln.value].copy()


def row_transition(sim, q_ids=None, n_seg=0, sim_a=None, alpha=0.5, threshold=0.0, cap=None):
    """-> dict(idx, seg, p, cnt, stats); rows of idx/seg/p are padded with -1/-1/0."""
    sim = np.ascontiguousarray(sim, np.float32)
    nq, nt = sim.shape
    if cap is None:
        cap = nt
    if q_ids is not None:
        q_ids = np.ascontiguousarray(q_ids, np.int64)
        n_seg = n_seg or nt
    if sim_a is not None:
        sim_a = np.ascontiguousarray(sim_a, np.float32)
    idx = np.full((nq, cap), -1, np.int32)
    seg = np.full((nq, cap), -1, np.int32)
    p = np.zeros((nq, cap), np.float32)
    cnt = np.zeros(nq, np.int32)
    stats = np.zeros((nq, 4), np.float32)
    _lib.avt_oracle_row_transition(_ptr(sim), C.c_int64(nq), C.c_int64(nt), C.c_int64(nt),
                                   _ptr(q_ids), C.c_int64(n_seg), _ptr(sim_a), C.c_int64(nt),
                                   C.c_float(alpha), C.c_float(threshold), C.c_int(cap), _ptr(idx),
                                   _ptr(seg), _ptr(p), _ptr(cnt), _ptr(stats))
    return dict(idx=idx, seg=seg, p=p, cnt=cnt, stats=stats)


def row_topk(sim, k, self_col=None):
    sim = np.ascontiguousarray(sim, np.float32)
    nq, nt = sim.shape
    if self_col is not None:
        self_col = np.ascontiguousarray(self_col, np.int64)
    idx = np.empty((nq, k), np.int32)
    val = np.empty((nq, k), np.float32)
    _lib.avt_oracle_row_topk(_ptr(sim), C.c_int64(nq), C.c_int64(nt), C.c_int64(nt), _ptr(self_col),
                             C.c_int(k), _ptr(idx), _ptr(val))
    return idx, val


def softmax_ce_fwd(logits, label=None):
    logits = np.ascontiguousarray(logits, np.float32)
    b, c = logits.shape
    if label is not None:
        label = np.ascontiguousarray(label, np.int64)
    loss = np.empty(b, np.float32)
    prob = np.empty((b, c), np.float32)
    _lib.avt_oracle_softmax_ce_fwd(_ptr(logits), C.c_int64(b), C.c_int64(c), _ptr(label), _ptr(loss),
                                   _ptr(prob))
    return loss, prob


def softmax_ce_bwd(prob, label=None, scale=1.0):
    prob = np.ascontiguousarray(prob, np.float32)
    b, c = prob.shape
    if label is not None:
        label = np.ascontiguousarray(label, np.int64)
    d = np.empty((b, c), np.float32)
    _lib.avt_oracle_softmax_ce_bwd(_ptr(prob), _ptr(label), C.c_int64(b), C.c_int64(c),
                                   C.c_float(scale), _ptr(d))
    return d


def bf16_bits_to_f32(u16):
    return (u16.astype(np.uint32) << 16).view(np.float32)
