"""Python half of the CPU oracle: restatement of the reference's host-side logic
and of the torch ops the hot path calls, in plain numpy / CPU torch.

TEST INFRASTRUCTURE ONLY — imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg; never by the product package.

All citations are relative to /root/reference/contrastive_video_textures/.
Pinned by tests/test_oracle_golden.py against tests/golden/*.npz (vectors made
by tools/gen_golden.py from the reference's own code run in the build container).

PARITY UNPINNED: `process_cv2_inputs` / SlowFast live in the third-party
`slowfast` package that the reference neither vendors nor version-pins
(models/models.py:18-20, 365-367); `pack_clip` below follows upstream PySlowFast
(slowfast/visualization/utils.py process_cv2_inputs + pack_pathway_output:
NUM_FRAMES 32, ALPHA 4, mean 0.45, std 0.225) with the /255 already applied by
the caller as validate.py:121 does.  No reference test pins it.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

FAST_T, SLOW_T, ALPHA = 32, 8, 4


# ---- clip preprocessing (models.py:364-383, validate.py:120-125, 333-344) -------
def sample_indices(win_len):
    fast = torch.linspace(0, win_len - 1, FAST_T).long()
    pick = torch.linspace(0, FAST_T - 1, FAST_T // ALPHA).long()
    return fast.numpy(), fast[pick].numpy()


def pack_clip(frames_u8, start, win_len, out_hw=224, mean=0.45, std=0.225, bgr=True):
    """frames_u8: uint8 [F,H,W,3] RGB (numpy or CPU tensor) -> (slow [3,8,hw,hw], fast [3,32,hw,hw]) fp32."""
    fr = torch.as_tensor(frames_u8)[start : start + win_len]
    x = fr.float() / 255  # validate.py:121
    if bgr:
        x = x[:, :, :, [2, 1, 0]]  # validate.py:124-125
    x = (x - mean) / std  # tensor_normalize
    x = x.permute(3, 0, 1, 2)  # T H W C -> C T H W
    fast_idx = torch.linspace(0, x.shape[1] - 1, FAST_T).long()
    fast = torch.index_select(x, 1, fast_idx)
    slow = torch.index_select(fast, 1, torch.linspace(0, fast.shape[1] - 1, fast.shape[1] // ALPHA).long())
    # models.py:369-376: F.interpolate(item.squeeze(0), size=(S,S), mode="bilinear") on [C,T,H,W]
    slow = F.interpolate(slow, size=(out_hw, out_hw), mode="bilinear")
    fast = F.interpolate(fast, size=(out_hw, out_hw), mode="bilinear")
    return slow, fast


# ---- chunking helpers (utils/utils.py:208-260) -----------------------------------
def split_into_batches(x, max_segments):
    """x [1,N,...] -> ([ceil(N/m), m, ...] zero padded, N)   (utils.py:208-230)"""
    x = np.asarray(x)
    assert x.shape[0] == 1
    n = x.shape[1]
    nb = math.ceil(n / max_segments)
    out = np.zeros((nb, max_segments) + x.shape[2:], x.dtype)
    for b in range(nb):
        lo = b * max_segments
        hi = min(lo + max_segments, n)
        out[b, : hi - lo] = x[0, lo:hi]
    return out, n


def split_into_overlapping_segments(x, max_segments, W, S):
    """x [N,...] -> ([batch, m*S+W, ...] zero padded, N)   (utils.py:233-260; chunk start uses m-1 [quirk Q4])"""
    x = np.asarray(x)
    n = x.shape[0]
    total = math.ceil((n - W) / S)
    chunk = max_segments * S + W
    nb = math.ceil(total / max_segments)
    out = np.zeros((nb, chunk) + x.shape[1:], x.dtype)
    for b in range(nb):
        lo = b * S * (max_segments - 1)
        hi = min(lo + chunk, n)
        if hi > lo:
            out[b, : hi - lo] = x[lo:hi]
    return out, n


# ---- stitch-loop index logic (validate.py:188-195, 369-391, 442-493) -------------
def num_segments(n_frames, W, S):
    return math.floor((n_frames - W) / S)  # validate.py:189


def target_segment_ids(q_id, L):
    """validate.py:369-378: [pos] + every segment except q and pos, ascending."""
    pos = min(q_id + 1, L - 1)
    mask = np.ones(L, bool)
    mask[[q_id, pos]] = False
    return np.concatenate((np.array([pos]), np.arange(L)[mask]))


def target_frame_ids(seg_ids, W, S):
    """validate.py:380-388: concatenated frame ranges, order-preserving unique."""
    ids = np.concatenate([np.arange(i * S, i * S + W) for i in seg_ids])
    _, first = np.unique(ids, return_index=True)
    return ids[np.sort(first)]


def compat_window_frames(q_id, n_frames, W, S, mbs, n_gpus=1):
    """Frame ids each OUTPUT slot of the reference's validate() row actually scores [quirks Q3/Q4].

    Returns (frames [len(target_segment_ids), W] int64 with -1 = zero padding,
             seg_ids = os_ids_t the labels the reference attaches to those slots).
    Follows validate.py:369-395 (t_video, chunking), models.py:358-367 (re-windowing of a
    chunk at stride S) and validate.py:442-493/522 (grouping by n_gpus, num_valid slice).
    """
    L = num_segments(n_frames, W, S)
    seg_ids = target_segment_ids(q_id, L)
    fids = target_frame_ids(seg_ids, W, S)
    chunks, _ = split_into_overlapping_segments(fids + 1, mbs, W, S)  # +1 so that padding (0) -> -1
    chunks = chunks.astype(np.int64) - 1
    n_out = len(seg_ids)
    out = np.full((n_out, W), -1, np.int64)
    num_valid = n_out
    n_calls = math.ceil(len(chunks) / n_gpus)
    for itr in range(n_calls):
        group = chunks[itr * n_gpus : itr * n_gpus + n_gpus]
        # each replica re-windows row 0 of its chunk: windows i*S : i*S+W, i < mbs (models.py:358-367)
        wins = np.stack([np.stack([c[i * S : i * S + W] for i in range(mbs)]) for c in group])
        flat = wins.reshape(-1, W)
        take = min(num_valid, n_gpus * mbs)
        take = min(take, flat.shape[0])
        lo = itr * n_gpus * mbs
        if take > 0:
            out[lo : lo + take] = flat[:take]
        num_valid -= mbs * n_gpus
    return out, seg_ids


def frame_bookkeeping(q_id, p_q_id, W, S):
    """validate.py:580-615: frame ids appended for the chosen segment, and whether it was a jump."""
    if p_q_id == -1:
        return np.arange(q_id * S, q_id * S + W), False
    ids = np.arange(q_id * S + (W - S), q_id * S + W)
    return ids, q_id != p_q_id + 1


def row_postprocess(output, threshold, output_a=None, alpha=0.5):
    """validate.py:524-572 on one row with torch CPU ops, exactly as written there.

    Returns dict(p_pre (after /sum and blend), ce, p_post, choices, entropy)."""
    out = torch.as_tensor(np.array(output, np.float32)).clone()
    out /= out.sum()
    if output_a is not None:
        oa = torch.as_tensor(np.array(output_a, np.float32)).clone()
        oa /= oa.sum()
        out = alpha * out + (1 - alpha) * oa
    pre = out.clone()
    ce = torch.nn.CrossEntropyLoss()(out.unsqueeze(0), torch.zeros(1, dtype=torch.long))
    out[out < (out.max() - threshold * out.max())] = 0.0
    out[torch.nonzero(out).view(-1)] /= out.sum()
    nz = out.nonzero().view(-1)
    ent = abs(out[nz].log().mean(-1))
    return dict(p_pre=pre.numpy(), ce=float(ce), p_post=out.numpy(), choices=nz.numpy(), entropy=float(ent))


def stitch_walk(row_fn, n_frames, W, S, max_length, q_id=10, seed=None, rng=None):
    """The serial remainder of validate(): validate.py:324, 570-572, 580-615, 685.

    row_fn(q_id) -> (choices (positions, ascending), os_ids_t).  Consumes exactly one
    np.random.choice per step from the LEGACY global-style RandomState."""
    rng = rng or np.random.RandomState(seed)
    new_ids, p_q, jumps, steps = [], -1, 0, []
    while len(new_ids) < max_length:
        choices, os_ids = row_fn(q_id)
        rdm = rng.choice(np.asarray(choices))
        q_id = int(os_ids[rdm])
        ids, jump = frame_bookkeeping(q_id, p_q, W, S)
        jumps += int(jump)
        new_ids.extend(int(i) for i in ids)
        steps.append(q_id)
        p_q = q_id
    return new_ids, steps, jumps


# ---- classic video textures, config 1 (baselines/classic_video_textures) -------------
def classic_d1_p1(frames, sigma_factor):
    """computeD1.py:47-96 (RGB path) + :240-247: D1[i,j]=||f_i-f_j||_2, P1 shifted/normalised."""
    f = torch.as_tensor(np.asarray(frames)).float().reshape(len(frames), -1)
    d1 = torch.cdist(f.double(), f.double()).float()
    nz = torch.nonzero(d1).size(0)
    sigma = sigma_factor * (d1.sum() / nz)
    p1 = torch.exp(-d1 / sigma)
    p1 = torch.cat((p1[1:, :], p1[-1, :].unsqueeze(0)), dim=0)
    p1 = p1 / p1.sum(1, keepdim=True)
    return d1.numpy(), p1.numpy(), float(sigma)


def classic_d2(d1, sigma_factor, filter_size=16):
    """computeD2.py:21-52: diagonal binomial filter, valid conv, P2."""
    d1 = torch.as_tensor(d1)
    w = torch.tensor(np.diag((np.poly1d([0.5, 0.5]) ** (filter_size - 1)).coeffs), dtype=torch.float32)
    d2 = F.conv2d(d1.view(1, 1, *d1.shape), w.view(1, 1, filter_size, filter_size))
    d2 = d2.view(d2.shape[2], d2.shape[3])
    nz = torch.nonzero(d2).size(0)
    sigma = sigma_factor * (d2.sum() / nz)
    p2 = torch.exp(-d2 / sigma)
    p2 = torch.cat((p2[1:, :], p2[-1, :].unsqueeze(0)), dim=0)
    p2 = p2 / p2.sum(1, keepdim=True)
    return d2.numpy(), p2.numpy(), float(sigma)


# ---- reference-compat stitch rows at oracle level (CPU torch encoders + C oracle) ------
def pack_clip_ids(frames01_bgr, ids, out_hw, mean=0.45, std=0.225):
    """Window given by explicit frame ids (-1 = the zero frame split_into_overlapping_segments pads with).
    frames01_bgr: float [F,H,W,3] already /255 and BGR (validate.py:120-125)."""
    ids = np.asarray(ids)
    fr = torch.zeros((len(ids),) + tuple(frames01_bgr.shape[1:]), dtype=torch.float32)
    ok = ids >= 0
    fr[torch.from_numpy(ok)] = frames01_bgr[torch.from_numpy(ids[ok])]
    x = ((fr - mean) / std).permute(3, 0, 1, 2)
    fast = torch.index_select(x, 1, torch.linspace(0, x.shape[1] - 1, FAST_T).long())
    slow = torch.index_select(fast, 1, torch.linspace(0, fast.shape[1] - 1, fast.shape[1] // ALPHA).long())
    slow = F.interpolate(slow, size=(out_hw, out_hw), mode="bilinear")
    fast = F.interpolate(fast, size=(out_hw, out_hw), mode="bilinear")
    return slow, fast


class CompatOracle:
    """What the reference's validate() computes per step, with every distinct window encoded once.

    q_enc / t_enc: SlowFast-contract modules ([slow, fast] -> [B,D]); vgg: audio module or None.
    Embeddings are cached by window frame-id tuple; rows are produced by the C oracle
    (l2norm -> canonical fp32 sim -> /temp)."""

    def __init__(self, video_u8, W, S, mbs, n_gpus, img_size, q_enc, t_enc, temp, audio_eg=None, vgg=None,
                 driving_eg=None):
        from . import cref

        self.cref = cref
        v = torch.as_tensor(video_u8).float() / 255
        self.frames = v[:, :, :, [2, 1, 0]]
        self.F = len(v)
        self.W, self.S, self.mbs, self.G, self.hw, self.temp = W, S, mbs, n_gpus, img_size, temp
        self.q_enc, self.t_enc, self.vgg = q_enc, t_enc, vgg
        self.L = num_segments(self.F, W, S)
        self.audio_eg = audio_eg
        self.driving_eg = driving_eg
        self.cache_t, self.cache_q = {}, {}
        self.a_feat = None
        if vgg is not None and audio_eg is not None:
            with torch.no_grad():
                self.a_feat = vgg(torch.as_tensor(audio_eg[: self.L]).float()).numpy()
                self.d_feat = vgg(torch.as_tensor(driving_eg).float()).numpy() if driving_eg is not None else None

    def _embed(self, enc, cache, keys):
        miss = [k for k in dict.fromkeys(keys) if k not in cache]
        for i in range(0, len(miss), 64):
            part = miss[i : i + 64]
            packs = [pack_clip_ids(self.frames, np.array(k), self.hw) for k in part]
            with torch.no_grad():
                e = enc([torch.stack([p[0] for p in packs]), torch.stack([p[1] for p in packs])]).numpy()
            for k, row in zip(part, e):
                cache[k] = row
        return np.stack([cache[k] for k in keys])

    def row(self, q_id, step):
        """-> (raw logits [n_out], raw audio logits or None, os_ids_t)"""
        wins, seg_ids = compat_window_frames(q_id, self.F, self.W, self.S, self.mbs, self.G)
        tv = self._embed(self.t_enc, self.cache_t, [tuple(w) for w in wins])
        qv = self._embed(self.q_enc, self.cache_q, [tuple(range(q_id * self.S, q_id * self.S + self.W))])
        ta = qa = None
        if self.a_feat is not None:
            mx = self.a_feat.shape[0] - 1
            ta = self.a_feat[np.minimum(seg_ids, mx)]  # validate.py:398-401
            qa = self.a_feat[min(q_id, mx)][None]  # validate.py:346
        qn, _, _ = self.cref.l2norm_rows(qv, qa, want_split=False)
        tn, _, _ = self.cref.l2norm_rows(tv, ta, want_split=False)
        out = self.cref.sim_f32(qn, tn, self.temp)[0]
        out_a = None
        if self.driving_eg is not None:  # models.py:424-439 (VGG branch), driving example `step` (validate.py:417)
            dn, _, _ = self.cref.l2norm_rows(self.d_feat[step][None], want_split=False)
            sn, _, _ = self.cref.l2norm_rows(ta, want_split=False)
            out_a = self.cref.sim_f32(dn, sn, self.temp)[0]
        return out, out_a, seg_ids


# ---- audio front-end (utils/mel_features.py:21-205, utils/vggish_utils.py:27-69, utils/vggish_params.py:27-38) ----
def log_mel(wave, win=400, hop=160, n_mel=64, sr=16000.0, lo_hz=125.0, hi_hz=7500.0, offset=0.01):
    """float64 [n_frames, n_mel]: frame (tail dropped, :21-45) * periodic Hann (:41-43) -> |DFT_512| (:48-68; written
    as an explicit DFT matrix product, the definition np.fft.rfft implements) -> HTK mel matrix (:117-173, DC bin
    zeroed :172) -> log(. + 0.01) (:176-205).  Pinned by G6 (the reference's own waveform_to_examples output)."""
    x = np.asarray(wave, dtype=np.float64)
    fft_len = 2 ** int(math.ceil(math.log(win) / math.log(2.0)))
    n_frames = 1 + (len(x) - win) // hop if len(x) >= win else 0
    nb = fft_len // 2 + 1
    if n_frames == 0:
        return np.zeros((0, n_mel))
    idx = np.arange(n_frames)[:, None] * hop + np.arange(win)[None, :]
    fr = x[idx] * (0.5 - 0.5 * np.cos(2 * np.pi / win * np.arange(win)))
    ang = -2.0 * np.pi * ((np.arange(win)[:, None] * np.arange(nb)[None, :]) % fft_len) / fft_len
    spec = np.abs(fr @ np.cos(ang) + 1j * (fr @ np.sin(ang)))
    mel_of = lambda hz: 1127.0 * np.log(1.0 + hz / 700.0)
    spec_mel = mel_of(np.linspace(0.0, sr / 2.0, nb))
    edges = np.linspace(mel_of(lo_hz), mel_of(hi_hz), n_mel + 2)
    m = np.zeros((nb, n_mel))
    for i in range(n_mel):
        lo, ce, hi = edges[i : i + 3]
        m[:, i] = np.maximum(0.0, np.minimum((spec_mel - lo) / (ce - lo), (hi - spec_mel) / (hi - ce)))
    m[0, :] = 0.0
    return np.log(spec @ m + offset)


def logmel_examples(lm, ex_len=100, ex_hop=10):
    """vggish_utils.py:60-68 framing: [n_ex, ex_len, n_mel] float64 (incomplete tail dropped)."""
    n = 1 + (len(lm) - ex_len) // ex_hop if len(lm) >= ex_len else 0
    return np.stack([lm[e * ex_hop : e * ex_hop + ex_len] for e in range(n)]) if n else np.zeros((0, ex_len, lm.shape[1]))
