"""How far two encoder implementations move the texture: score deviation, survivor sets, stitched frames.

The reference's encoders compute in fp32 (contrastive_video_textures/models/models.py:335, 399); its survivor cut
`p < max - th*max` (validate.py:554) and uniform draw (validate.py:570-572) turn score noise into different stitch
indices.  `compare_tables` measures exactly that between two pairs of embedding tables built FROM THE SAME FRAMES
(e.g. the bf16 MFMA encoder and an fp32 encoder): everything downstream of the tables is the product path
(HIP l2norm, exact-fp32 MFMA similarity, HIP row select, host RNG walk)."""
import numpy as np
import torch

from . import ops, texture


def walk_from_survivors(idx, seg, cnt, n_frames, W, S, max_length, q_id=10, rng=None):
    """The serial stitch walk (validate.py:324, 570-615) over precomputed survivor lists: idx/seg [N, cap] positions /
    segment ids in the reference's [pos]+others order, cnt [N].  One rng.choice per step, like the reference.
    -> (frame ids, chosen segments)."""
    rng = np.random if rng is None else rng
    frames, chosen, p_q = [], [], -1
    while len(frames) < max_length:
        k = int(cnt[q_id])
        if k > idx.shape[1]:
            raise ValueError("row %d has %d survivors, more than the %d kept" % (q_id, k, idx.shape[1]))
        pos = rng.choice(idx[q_id, :k])  # validate.py:570: uniform over the survivors' positions
        q_id = int(seg[q_id, int(np.nonzero(idx[q_id, :k] == pos)[0][0])])
        frames.extend(range(q_id * S, q_id * S + W) if p_q == -1 else range(q_id * S + (W - S), q_id * S + W))
        chosen.append(q_id)
        p_q = q_id
    return frames, chosen


def _build(qv, tv, temp):
    qn, _, _ = ops.l2norm_rows(qv.float().contiguous())
    tn, _, _ = ops.l2norm_rows(tv.float().contiguous())
    return ops.sim_gemm_nt(qn, tn, temp, "f32")


def compare_tables(qa, ta, qb, tb, temp, W, S, thresholds=(0.0, 0.3), walk_frames=600, seeds=(7, 8, 9), q_id=10):
    """Tables A (qa, ta) against tables B (qb, tb), fp32 [N, D] device tensors of the SAME windows.  -> dict with
    max/mean |score_A - score_B|, the relative embedding error, and per threshold: fraction of rows with identical
    survivor sets, fraction of identical stitch steps and whether the frames lists agree, under fixed host RNG seeds."""
    n = qa.shape[0]
    sa, sb = _build(qa, ta, temp), _build(qb, tb, temp)
    d = (sa - sb).abs()
    out = {"windows": int(n), "max_abs_dscore": float(d.max()), "mean_abs_dscore": float(d.mean()),
           "rel_embedding_err_q": float(((qa - qb).norm(dim=1) / qb.norm(dim=1)).max()),
           "rel_embedding_err_t": float(((ta - tb).norm(dim=1) / tb.norm(dim=1)).max()),
           "score_spread": float(sb.max() - sb.min()), "thresholds": {}}
    q_ids = torch.arange(n, device=qa.device, dtype=torch.int64)
    n_frames = n * S + W
    for th in thresholds:
        ra = ops.row_transition(sa, q_ids=q_ids, threshold=th, cap=n)
        rb = ops.row_transition(sb, q_ids=q_ids, threshold=th, cap=n)
        same = (ra["cnt"] == rb["cnt"]) & (ra["seg"] == rb["seg"]).all(dim=1)
        host = [{k: r[k].cpu().numpy() for k in ("idx", "seg", "cnt")} for r in (ra, rb)]
        steps_same = steps = 0
        lists_same = 0
        for seed in seeds:
            fa, ca = walk_from_survivors(host[0]["idx"], host[0]["seg"], host[0]["cnt"], n_frames, W, S, walk_frames,
                                         q_id=min(q_id, n - 1), rng=np.random.RandomState(seed))
            fb, cb = walk_from_survivors(host[1]["idx"], host[1]["seg"], host[1]["cnt"], n_frames, W, S, walk_frames,
                                         q_id=min(q_id, n - 1), rng=np.random.RandomState(seed))
            m = min(len(ca), len(cb))
            agree = np.asarray(ca[:m]) == np.asarray(cb[:m])
            first_diff = int(np.argmin(agree)) if not agree.all() else m
            steps_same += first_diff
            steps += m
            lists_same += int(fa == fb)
        # threshold 0.0 keeps the candidates that TIE with the row maximum (validate.py:553).  Where the fp32 reference itself has
        # exact ties (two candidates with bit-identical scores: e.g. windows whose features are all dead embed to one constant
        # vector), which of them another arithmetic calls equal is decided in the last bit: such rows are counted apart, and on
        # them the question is whether A's survivors are a SUBSET of the reference's tie set (A picked among the tied ones)
        tie_rows = np.nonzero(host[1]["cnt"] > 1)[0] if th == 0.0 else np.zeros(0, dtype=np.int64)
        same_h = same.cpu().numpy()
        no_tie = np.ones(n, dtype=bool)
        no_tie[tie_rows] = False
        subset = sum(1 for r in tie_rows if set(host[0]["seg"][r, : host[0]["cnt"][r]].tolist()) <= set(host[1]["seg"][r, : host[1]["cnt"][r]].tolist()))
        # ... and where the survivors differ although the reference has no exact tie: how far apart, IN THE REFERENCE, are its own
        # maximum and the worst candidate the other arithmetic kept?  (|A - B| <= d everywhere bounds this gap by 2 d: a row can only
        # differ where the reference itself separates the candidates by less than the two arithmetics' distance)
        gap = 0.0
        if th == 0.0:
            sb_h = None
            for r in np.nonzero(~same_h)[0]:
                if sb_h is None:
                    sb_h = sb.cpu().numpy()
                ka, kb = int(host[0]["cnt"][r]), int(host[1]["cnt"][r])
                if ka and kb:
                    ref_max = float(sb_h[r, host[1]["seg"][r, :kb]].max())
                    gap = max(gap, ref_max - float(sb_h[r, host[0]["seg"][r, :ka]].min()))
        ties = {} if th != 0.0 else {
            "max_ref_gap_on_differing_rows": gap,
            "rows_with_exact_ties_ref": int(len(tie_rows)),
            "rows_identical_survivors_outside_tie_rows": float(same_h[no_tie].mean()) if no_tie.any() else 1.0,
            "tie_rows_survivors_subset_of_ref_ties": "%d/%d" % (subset, len(tie_rows))}
        out["thresholds"]["%.1f" % th] = {
            **ties,
            "rows_identical_survivors": float(same.float().mean()),
            "mean_survivors": float(rb["cnt"].float().mean()),
            "walk_steps_identical_before_first_divergence": steps_same / max(steps, 1),
            "frames_lists_identical": "%d/%d" % (lists_same, len(seeds))}
    return out
