"""Classic (Schoedl-style) video textures — BASELINE config 1, CPU plumbing only.

`compute_D1_device` runs the distance matrix on the GPU (csrc/pairwise.hip); the rest restates
baselines/classic_video_textures/computeD1.py:47-96 + :240-247 (pairwise L2 D1, sigma,
P1 shifted by one row and row-normalised), computeD2.py:21-52 (diagonal binomial filter) and
q_learning.py:27-68 (future-cost iteration) on CPU torch, without the `.cuda()` calls and the missing
`utils`/`models` modules that make the shipped scripts unrunnable.  Not a GPU target.
"""
import copy

import numpy as np
import torch
import torch.nn.functional as F


def compute_D1(frames, sigma_factor, batch_size=128):
    """frames [N,H,W,C] -> (D1 [N,N], P1 [N,N], sigma).  Tiled like the reference's `slow` path so that the
    [bs, bs, H*W*C] difference tensor, not an N^2 one, is materialised."""
    f = torch.as_tensor(frames).float().reshape(len(frames), -1)
    n = len(f)
    d1 = torch.ones((n, n))
    for i in range(0, n, batch_size):
        a = f[i : i + batch_size]
        for j in range(0, n, batch_size):
            b = f[j : j + batch_size]
            d1[i : i + batch_size, j : j + batch_size] = torch.norm(a.unsqueeze(1) - b.unsqueeze(0), dim=2)
    nz = torch.nonzero(d1).size(0)
    sigma = sigma_factor * (d1.sum() / nz)
    p1 = torch.exp(-d1 / sigma)
    p1 = torch.cat((p1[1:, :], p1[-1, :].unsqueeze(0)), dim=0)
    p1 = p1 / p1.sum(1, keepdim=True)
    return d1, p1, sigma


def compute_D1_device(frames, sigma_factor, device="cuda"):
    """compute_D1 on the MI355X: the N x N distance matrix on the hand-written pairwise kernel (csrc/pairwise.hip), the
    N x N post-processing (sigma, exp, shift, row-normalise: computeD1.py:240-247) as device tensor ops."""
    from . import ops

    f = torch.as_tensor(frames).float().reshape(len(frames), -1).contiguous().to(device)
    d1 = ops.pairwise_l2(f)
    nz = torch.nonzero(d1).size(0)
    sigma = sigma_factor * (d1.sum() / nz)
    p1 = torch.exp(-d1 / sigma)
    p1 = torch.cat((p1[1:, :], p1[-1, :].unsqueeze(0)), dim=0)
    p1 = p1 / p1.sum(1, keepdim=True)
    return d1, p1, sigma


def _p_from_d(d, sigma_factor):
    """sigma, exp, one-row shift, row-normalise (computeD1.py:240-247; the same tail in computeD2.py and q_learning.py)."""
    nz = torch.nonzero(d).size(0)
    sigma = sigma_factor * (d.sum() / nz)
    p = torch.exp(-d / sigma)
    p = torch.cat((p[1:, :], p[-1, :].unsqueeze(0)), dim=0)
    return p / p.sum(1, keepdim=True), sigma


def compute_D2_device(d1, sigma_factor, filter_size=16):
    """compute_D2 on the MI355X: the diagonal binomial filter as fs taps per output (csrc/classic.hip), not a dense
    [fs, fs] conv2d over a diagonal kernel (computeD2.py:21-52)."""
    from . import ops

    w = torch.tensor((np.poly1d([0.5, 0.5]) ** (filter_size - 1)).coeffs, dtype=torch.float32, device=d1.device)
    d2 = ops.diag_filter(d1.contiguous(), w)
    p2, sigma = _p_from_d(d2, sigma_factor)
    return d2, p2, sigma, torch.diag(w)


def q_learning_device(d2, sigma_factor, p=0.7, alpha=0.997, thresholding=0.75, max_iter=1000):
    """q_learning on the MI355X: every sweep inside ONE kernel launch with the matrix resident in LDS (csrc/classic.hip)."""
    from . import ops

    d3_new, _ = ops.q_learning((d2 ** p).contiguous(), alpha, 10e-3, max_iter)
    p3, sigma = _p_from_d(d3_new, sigma_factor)
    p3_new = p3.clone()
    cut = p3_new.max(dim=1, keepdim=True)[0]
    p3_new[p3_new < (cut - thresholding * cut)] = 0.0
    return d3_new, p3, p3_new, sigma


def compute_D2(d1, sigma_factor, filter_size=16, stride=1):
    w = torch.tensor(np.diag((np.poly1d([0.5, 0.5]) ** (filter_size - 1)).coeffs), dtype=torch.float32)
    d2 = F.conv2d(d1.view(1, 1, *d1.shape), w.view(1, 1, filter_size, filter_size), stride=stride)
    d2 = d2.view(d2.shape[2], d2.shape[3])
    nz = torch.nonzero(d2).size(0)
    sigma = sigma_factor * (d2.sum() / nz)
    p2 = torch.exp(-d2 / sigma)
    p2 = torch.cat((p2[1:, :], p2[-1, :].unsqueeze(0)), dim=0)
    p2 = p2 / p2.sum(1, keepdim=True)
    return d2, p2, sigma, w


def q_learning(d2, sigma_factor, p=0.7, alpha=0.997, thresholding=0.75, max_iter=1000):
    d3 = d2 ** p
    d3_new = copy.deepcopy(d3)
    n = d3.shape[0]
    off = ~torch.eye(n, d3.shape[1], dtype=torch.bool)
    eps, it = 10000.0, 0
    while eps > 10e-3 and it < max_iter:
        d3_old = copy.deepcopy(d3_new)
        mins = d3_old[off].view(n, -1).min(dim=1)[0]  # computed once per sweep, as the reference does per row
        for i in range(n - 1, 0, -1):
            d3_new[i] = d3[i] + alpha * mins
        eps = float(((d3_new - d3_old) ** 2).mean())
        it += 1
    nz = torch.nonzero(d3_new).size(0)
    sigma = sigma_factor * (d3_new.sum() / nz)
    p3 = torch.exp(-d3_new / sigma)
    p3 = torch.cat((p3[1:, :], p3[-1, :].unsqueeze(0)), dim=0)
    p3 = p3 / p3.sum(1, keepdim=True)
    p3_new = copy.deepcopy(p3)
    for i in range(len(p3_new)):
        p3_new[i][p3_new[i] < (p3_new[i].max() - thresholding * p3_new[i].max())] = 0.0
    return d3_new, p3, p3_new, sigma


def random_walk(p, n_steps, start=0, rng=None):
    """Frame sequence by sampling the transition matrix row by row (video_textures.py:32-241, core loop)."""
    rng = rng or np.random
    p = np.asarray(p, np.float64)
    seq, cur = [start], start
    for _ in range(n_steps - 1):
        row = p[cur] / p[cur].sum()
        cur = int(rng.choice(len(row), p=row))
        seq.append(cur)
    return seq
