"""Host helpers with the reference's names (contrastive_video_textures/utils/utils.py).

Only `split_into_batches` / `split_into_overlapping_segments` / `combine_batches` sit on the stitch
path (utils.py:192-260); they are pure index arithmetic kept for callers that still chunk by hand.
The MI355X stitch path itself never materialises these zero-padded copies (see texture.py)."""
import math
import os
import shutil
import subprocess

import torch


class AverageMeter(object):
    """Running average (utils.py:7-40)."""

    def __init__(self):
        self.reset()

    def reset(self):
        self.val = self.avg = self.sum = self.count = 0

    def update(self, val, n=1):
        self.val = val
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count


def combine_batches(tensor, num_valid):
    """[num_gpus, n, ...] -> [1, num_valid, ...] (utils.py:192-205)."""
    g, n = tensor.shape[:2]
    assert num_valid <= g * n
    return tensor.reshape(1, g * n, *tensor.shape[2:])[0:1, :num_valid]


def split_into_batches(tensor, max_segments_per_gpu):
    """[1, N, ...] -> ([ceil(N/m), m, ...] zero padded, N) (utils.py:208-230)."""
    assert tensor.size(0) == 1
    n = tensor.size(1)
    nb = math.ceil(n / max_segments_per_gpu)
    out = torch.zeros(nb, max_segments_per_gpu, *tensor.shape[2:], dtype=tensor.dtype, device=tensor.device)
    flat = out.view(nb * max_segments_per_gpu, *tensor.shape[2:])
    flat[:n] = tensor[0]
    return out, n


def split_into_overlapping_segments(tensor, max_segments_per_gpu, W, S):
    """[N, ...] -> ([batch, m*S+W, ...] zero padded, N).  Chunk c starts at frame c*S*(m-1): the
    reference's off-by-one [quirk Q4] (utils.py:255) is part of the drop-in behaviour."""
    n = tensor.size(0)
    total = math.ceil((n - W) / S)
    chunk = max_segments_per_gpu * S + W
    nb = math.ceil(total / max_segments_per_gpu)
    out = torch.zeros(nb, chunk, *tensor.shape[1:], dtype=tensor.dtype, device=tensor.device)
    for b in range(nb):
        lo = b * S * (max_segments_per_gpu - 1)
        hi = min(lo + chunk, n)
        if hi > lo:
            out[b, : hi - lo] = tensor[lo:hi]
    return out, n


def save_videos(frames_dir, outfile, fps, interpolation=False, audio_w=None, SF=5, audio_file="",
                audio_file_intp="", frames_dir_intp=None, outfile_intp=None):
    """PNG folder -> mp4 through an ffmpeg subprocess when one is installed (utils.py:43-189 does the
    same; muxing is outside the hot path).  Returns False, loudly, when ffmpeg is absent."""
    if shutil.which("ffmpeg") is None:
        print("save_videos: ffmpeg not found; frames left in {}".format(frames_dir))
        return False
    cmd = ["ffmpeg", "-y", "-framerate", str(fps), "-i", os.path.join(frames_dir, "%04d.png")]
    if audio_file:
        cmd += ["-i", audio_file, "-shortest"]
    cmd += ["-pix_fmt", "yuv420p", outfile]
    subprocess.call(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return True


def save_video_raw(frames_u8, outfile, fps, audio_file=""):
    """Frame tensor -> mp4 WITHOUT the PNG round trip of the reference (validate.py:789-872 writes every frame as a PNG,
    utils.py:43-189 lets ffmpeg read them back): uint8 [n, H, W, 3] RGB frames (host or device; gathered from the resident
    video by index) are piped to ffmpeg's stdin as rawvideo.  Without an ffmpeg binary the frames are kept losslessly as
    <outfile>.npz (video, fps) — the same container read_video() accepts — and False is returned."""
    import numpy as np
    import torch

    v = torch.as_tensor(frames_u8)
    if v.dtype != torch.uint8 or v.dim() != 4 or v.shape[3] != 3:
        raise ValueError("save_video_raw expects uint8 [n, H, W, 3]")
    arr = v.contiguous().cpu().numpy()
    if shutil.which("ffmpeg") is None:
        # (stored, not deflated: zlib took 4 of the 9.5 s of an at-size validate() run — tools/r04_runs/gpu_r04_e2e_profile.sh)
        np.savez(os.path.splitext(outfile)[0] + ".npz", video=arr, fps=float(fps))
        print("save_video_raw: ffmpeg not found; wrote {}.npz".format(os.path.splitext(outfile)[0]))
        return False
    n, h, w, _ = arr.shape
    cmd = ["ffmpeg", "-y", "-f", "rawvideo", "-pix_fmt", "rgb24", "-s", "{}x{}".format(w, h), "-framerate", str(fps), "-i", "-"]
    if audio_file:
        cmd += ["-i", audio_file, "-shortest"]
    cmd += ["-pix_fmt", "yuv420p", outfile]
    proc = subprocess.Popen(cmd, stdin=subprocess.PIPE, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    proc.stdin.write(arr.tobytes())
    proc.stdin.close()
    return proc.wait() == 0
