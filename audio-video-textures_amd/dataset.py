"""`AudioVideoSegments` — the training / validation dataset plugin, drop-in for
contrastive_video_textures/dataset/dataset.py:24-253 (same constructor, __len__, item tuples, and the
same NumPy RNG consumption in the negative sampling, :183-190).

Host side by design: it only picks indices and slices the video held in RAM.  `segment_plan(idx)` exposes
the window starts of an item so a trainer can pack them on the MI355X with ops.clip_pack instead of the
per-item CPU preprocessing of dataset.py:145-209.
"""
import copy
import math
import os

import numpy as np
import torch
import torch.nn.functional as F
from torch.utils.data import Dataset

from .audio_frontend import waveform_to_examples
from .models import process_cv2_inputs
from .validate import read_audio, read_video


class AudioVideoSegments(Dataset):
    def __init__(self, args, video_name, split="train", video=None, audio=None):
        self.vdata, self.adata = args.vdata, args.adata
        self.video_name, self.split = video_name, split
        self.n_negs = args.n_negs
        self.crop_size = self.img_size = args.img_size
        self.enc_arch = args.enc_arch
        if video is None:
            self.video_filename = os.path.join(self.vdata, "{}.mp4".format(video_name))
            self.video_u8, fps = read_video(self.video_filename)
        else:
            self.video_u8, fps = torch.as_tensor(video[0]), video[1]
        self.fps = fps
        if self.enc_arch != "slowfast":
            # dataset.py:44-58 (ToPILImage/Resize/ToTensor/Normalize); resize by antialiased bilinear
            mean = torch.tensor([0.4345, 0.4051, 0.3775]).view(1, 3, 1, 1)
            std = torch.tensor([0.2768, 0.2713, 0.2737]).view(1, 3, 1, 1)
            v = self.video_u8.permute(0, 3, 1, 2).float() / 255
            if v.shape[-1] != args.img_size or v.shape[-2] != args.img_size:
                v = F.interpolate(v, size=(args.img_size, args.img_size), mode="bilinear", antialias=True)
            self.video = (v - mean) / std
        else:
            self.video = (self.video_u8.float() / 255)[:, :, :, [2, 1, 0]]  # dataset.py:68-73: 0-1, BGR
        print("Frame shape: ", self.video[0].shape)
        # Window of 0.5 seconds, stride of 0.2 seconds — overrides -w/-stride [quirk Q10] (dataset.py:78-80)
        args.window = math.ceil(self.fps / 2)
        args.stride = math.ceil(self.fps / 5)
        print("Stride {} Window {}".format(args.stride, args.window))
        self.stride, self.window = args.stride, args.window
        if self.adata is None and audio is None:
            # dummy audio (dataset.py:87-93); consumes the torch RNG in this order
            self.audio_w = torch.rand(len(self.video) * 10)
            self.apf = 10
            self.audio_eg = torch.rand((math.floor((len(self.video) - self.window) / self.stride)), 10)
        else:
            if audio is None:
                path = os.path.join(self.adata, "{}.wav".format(video_name))
                assert os.path.exists(path) or os.path.exists(os.path.splitext(path)[0] + ".npz"), \
                    "No audio found at {}".format(path)
                audio = read_audio(path)
            self.audio_w, self.sr = audio
            self.apf = math.floor(self.sr / self.fps)
            self.audio_w = np.asarray(self.audio_w)[: len(self.video) * self.apf]
            self.audio_eg = torch.from_numpy(waveform_to_examples(self.audio_w, self.sr)).unsqueeze(dim=1).float()
            self.audio_w = torch.tensor(self.audio_w)

    def __len__(self):
        n = math.floor((len(self.video) - self.window) / self.stride)
        return n - 1 if self.split == "train" else n

    def sample_ids(self, idx):
        """Segment ids of an item: (positive id, negative ids) with dataset.py:128-139, 181-190 semantics:
        n_negs drawn without replacement, then the first len(hard) overwritten by the 8 temporal neighbours
        (idx-4..idx-1, idx+2..idx+5) that fall in [0, len] — duplicates possible [quirk]."""
        n = self.__len__()
        if self.split == "train":
            pos = idx + 1
            neg_ids = np.arange(n + 1)
            mask = np.ones(n + 1, dtype=bool)
            mask[[idx, idx + 1]] = False
        else:
            pos = (idx + 1) % n
            neg_ids = np.arange(n)
            mask = np.ones(n, dtype=bool)
            mask[[idx, pos]] = False
        neg = neg_ids[mask, ...]
        if self.split == "train":
            neg = np.random.choice(neg, self.n_negs, replace=False)
            hard = np.array([idx - 4, idx - 3, idx - 2, idx - 1, idx + 2, idx + 3, idx + 4, idx + 5])
            hard = hard[hard >= 0]
            hard = hard[hard <= n]
            neg[: len(hard)] = hard
        return pos, neg

    def segment_plan(self, idx):
        """Window start frames (query, [positive] + negatives) for device-side packing."""
        pos, neg = self.sample_ids(idx)
        return idx * self.stride, np.concatenate(([pos], neg)) * self.stride

    def _clip(self, start):
        S, W = self.stride, self.window
        if self.enc_arch != "slowfast":
            return self.video[start : start + W]
        return [F.interpolate(item.squeeze(0), size=(self.img_size, self.img_size), mode="bilinear")
                for item in process_cv2_inputs(self.video[start : start + W])]

    def __getitem__(self, idx):
        S, W = self.stride, self.window
        pos, neg = self.sample_ids(idx)
        q_v = self._clip(idx * S)
        q_aw = self.audio_w[idx * S * self.apf : (idx * S + W) * self.apf]
        q_ae = self.audio_eg[idx]
        t_v = [self._clip(pos * S)] + [self._clip(int(i) * S) for i in neg]
        t_aw = [self.audio_w[pos * S * self.apf : (pos * S + W) * self.apf]]
        t_aw += [self.audio_w[int(i) * S * self.apf : (int(i) * S + W) * self.apf] for i in neg]
        pos_ae = idx + 1 if self.split == "train" else pos  # dataset.py:179 uses idx+1
        t_ae = [self.audio_eg[pos_ae]] + list(self.audio_eg[neg])
        if self.enc_arch == "slowfast":
            t_v = copy.deepcopy([torch.stack([x[k] for x in t_v]) for k in range(2)])
        else:
            t_v = torch.stack(t_v)
        t_aw, t_ae = torch.stack(t_aw), torch.stack(t_ae)
        if self.split == "train":
            return (q_v, q_aw, q_ae, t_v, t_aw, t_ae)
        ordering = torch.cat((torch.tensor([pos]), torch.tensor(neg)))
        return (q_v, q_aw, q_ae, t_v, t_aw, t_ae, idx, ordering)


class DeviceSegmentBatcher:
    """Training batches assembled ON the MI355X — the video part of `AudioVideoSegments.__getitem__` + default collate
    (dataset.py:121-253) without its per-item CPU preprocessing (dataset.py:145-209: process_cv2_inputs + interpolate per
    segment in DataLoader workers) and without a host round trip per step:

      * the uint8 video (and the log-mel examples) stay resident in HBM;
      * negatives are drawn by the device MT19937 kernel from NumPy's own stream (ops.negative_sample: the state of
        np.random is uploaded once by `seed_from_numpy()` and can be handed back by `sync_to_numpy()`), with the
        reference's hard-negative overwrite (dataset.py:183-190);
      * query / positive / negative windows are packed by the gather kernel (ops.clip_pack_gather), starts read on device.

    batch(idx) -> (q_frames, t_frames, q_audio_eg, t_audio_eg) with the shapes the DataLoader would deliver:
    q_frames [slow [B,3,8,hw,hw], fast [B,3,32,hw,hw]], t_frames [slow [B,1+negs,3,8,hw,hw], fast [B,1+negs,3,32,hw,hw]]."""

    def __init__(self, dataset, device, dtype=torch.float32):
        from . import ops

        if dataset.enc_arch != "slowfast" or dataset.split != "train":
            raise ValueError("DeviceSegmentBatcher packs SlowFast training items")
        self.ops, self.ds, self.dev, self.dtype = ops, dataset, torch.device(device), dtype
        self.frames = dataset.video_u8.to(self.dev).contiguous()
        self.audio_eg = dataset.audio_eg.to(self.dev) if dataset.audio_eg.dim() == 4 else None
        self.state = None
        self._gauss = (0, 0.0)
        # segment ids the sampler can produce are 0 .. len(ds) (positive = idx + 1, negatives from [0, len]); the gather kernel
        # reads its starts on the device and CLAMPS frame ids (a bad id cannot fault, but would silently repeat edge frames),
        # so the one check that every id is in range is made here, once
        need = len(dataset) * dataset.stride + dataset.window
        if need > self.frames.shape[0]:
            raise ValueError("DeviceSegmentBatcher: segment %d needs frames up to %d, the video has %d"
                             % (len(dataset), need, self.frames.shape[0]))

    def seed_from_numpy(self):
        st = np.random.get_state()
        self._gauss = (int(st[3]), float(st[4]))  # a cached gaussian of the host stream survives the round trip
        words = np.concatenate([np.asarray(st[1], np.uint32), np.array([st[2]], np.uint32)])
        self.state = torch.from_numpy(words.view(np.int32).copy()).to(self.dev)
        return self

    def sync_to_numpy(self):
        """Hands the advanced stream back to np.random (one D2H of 2.5 KB; only needed when host code draws next)."""
        words = self.state.cpu().numpy().view(np.uint32)
        np.random.set_state(("MT19937", words[:624].copy(), int(words[624]), self._gauss[0], self._gauss[1]))

    def sample(self, idx):
        """idx int64 [B] -> (positive ids [B], negative ids [B, n_negs]) on the device."""
        if self.state is None:
            self.seed_from_numpy()
        idx = torch.as_tensor(idx, dtype=torch.int64).to(self.dev).contiguous()
        neg = self.ops.negative_sample(self.state, idx, len(self.ds), self.ds.n_negs)
        return idx + 1, neg

    def batch(self, idx):
        ds, S, W = self.ds, self.ds.stride, self.ds.window
        idx = torch.as_tensor(idx, dtype=torch.int64).to(self.dev).contiguous()
        b = idx.numel()
        pos, neg = self.sample(idx)
        tgt = torch.cat((pos.view(b, 1), neg.to(torch.int64)), 1)  # [B, 1 + negs] segment ids
        starts = (torch.cat((idx, tgt.reshape(-1))) * S).to(torch.int32).contiguous()
        slow, fast = self.ops.clip_pack_gather(self.frames, starts, W, out_hw=ds.img_size, dtype=self.dtype)
        hw, n = ds.img_size, tgt.shape[1]
        q_frames = [slow[:b], fast[:b]]
        t_frames = [slow[b:].view(b, n, 3, 8, hw, hw), fast[b:].view(b, n, 3, 32, hw, hw)]
        q_ae = t_ae = None
        if self.audio_eg is not None:
            q_ae = self.audio_eg[idx]
            t_ae = self.audio_eg[tgt.reshape(-1)].view(b, n, *self.audio_eg.shape[1:])
        return q_frames, t_frames, q_ae, t_ae
