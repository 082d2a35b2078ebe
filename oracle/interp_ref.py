"""ORACLE (test infrastructure, never imported by the product): CPU restatement of the reference's SuperSloMo
interpolation at jumps — contrastive_video_textures/interpolate.py:75-147 (`interpolate.forward`), :49-71
(`modify_frames`' ToTensor + Normalize), models/slowmo.py:10-208 (`UNet`), :211-284 (`backWarp`) — as plain torch fp32
functions over state dicts with the reference's parameter names.  Pinned by tests/golden/g10_interp.npz, which
tools/gen_golden.py produced by running the reference's own UNet / backWarp / interpolate.forward in this container
(torchvision is absent there, so ToTensor / Normalize / ToPILImage are restated from torchvision's published
behaviour: x / 255, (x - mean) / std, x.mul(255).byte())."""
import torch
import torch.nn.functional as F

MEAN = (0.429, 0.431, 0.397)  # interpolate.py:51


def _lrelu(x):
    return F.leaky_relu(x, negative_slope=0.1)


def _conv(sd, name, x):
    w = sd[name + ".weight"]
    return F.conv2d(x, w, sd[name + ".bias"], stride=1, padding=(w.shape[-1] - 1) // 2)


def unet(sd, x):
    """slowmo.py:179-208"""
    x = _lrelu(_conv(sd, "conv1", x))
    skips = [_lrelu(_conv(sd, "conv2", x))]
    for i in range(1, 6):  # down blocks, slowmo.py:49-72
        y = F.avg_pool2d(skips[-1], 2)
        y = _lrelu(_conv(sd, "down%d.conv1" % i, y))
        skips.append(_lrelu(_conv(sd, "down%d.conv2" % i, y)))
    x = skips.pop()
    for i in range(1, 6):  # up blocks, slowmo.py:111-135
        x = F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=False)
        x = _lrelu(_conv(sd, "up%d.conv1" % i, x))
        x = _lrelu(_conv(sd, "up%d.conv2" % i, torch.cat((x, skips.pop()), 1)))
    return _lrelu(_conv(sd, "conv3", x))


def backwarp(img, flow):
    """slowmo.py:251-284 (grid_sample defaults of this torch: bilinear, zeros, align_corners=False)"""
    _, _, h, w = img.shape
    gy, gx = torch.meshgrid(torch.arange(h), torch.arange(w), indexing="ij")
    x = gx.unsqueeze(0).float() + flow[:, 0]
    y = gy.unsqueeze(0).float() + flow[:, 1]
    grid = torch.stack((2 * (x / w - 0.5), 2 * (y / h - 0.5)), dim=3)
    return F.grid_sample(img, grid, mode="bilinear", padding_mode="zeros", align_corners=False)


def to_tensor(frame_u8):
    """ToTensor + Normalize(MEAN, 1) of one [H,W,3] uint8 frame -> [3,H,W]   (interpolate.py:51-61)"""
    x = frame_u8.permute(2, 0, 1).float().div(255)
    return (x - torch.tensor(MEAN).view(3, 1, 1)) / torch.ones(3).view(3, 1, 1)


def to_u8(x):
    """revNormalize + ToPILImage of one [3,H,W] tensor -> [H,W,3] uint8   (interpolate.py:57-61)"""
    x = (x - (-torch.tensor(MEAN)).view(3, 1, 1)) / torch.ones(3).view(3, 1, 1)
    return x.mul(255).byte().permute(1, 2, 0).contiguous()


def interpolate_pair(sd_fc, sd_at, frame0_u8, frame1_u8, sf, return_float=False):
    """interpolate.forward for frames whose extents are multiples of 32 -> uint8 [sf-1, H, W, 3]."""
    with torch.no_grad():
        i0, i1 = to_tensor(frame0_u8).unsqueeze(0), to_tensor(frame1_u8).unsqueeze(0)
        flow = unet(sd_fc, torch.cat((i0, i1), 1))
        f01, f10 = flow[:, :2], flow[:, 2:]
        out, outf = [], []
        for k in range(1, sf):
            t = float(k) / sf
            temp = -t * (1 - t)
            ft0 = temp * f01 + (t * t) * f10
            ft1 = ((1 - t) * (1 - t)) * f01 + temp * f10
            g0, g1 = backwarp(i0, ft0), backwarp(i1, ft1)
            o = unet(sd_at, torch.cat((i0, i1, f01, f10, ft1, ft0, g1, g0), 1))
            ft0f, ft1f = o[:, :2] + ft0, o[:, 2:4] + ft1
            v0 = torch.sigmoid(o[:, 4:5])
            v1 = 1 - v0
            g0f, g1f = backwarp(i0, ft0f), backwarp(i1, ft1f)
            w0, w1 = 1 - t, t
            p = (w0 * v0 * g0f + w1 * v1 * g1f) / (w0 * v0 + w1 * v1)
            outf.append(p[0])
            out.append(to_u8(p[0]))
        return (torch.stack(out), torch.stack(outf)) if return_float else torch.stack(out)
