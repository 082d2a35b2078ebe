"""`train()` — one epoch of InfoNCE SGD, drop-in for contrastive_video_textures/train.py:39-210
(same signature, same meters and prints; tensorboard images are skipped when no logger is given).
The loss runs on the HIP softmax-CE kernels (models.InfoNCECriterion)."""
import time
from collections import OrderedDict

import torch

from . import train_ops
from ._lib import AvtError
from .models import InfoNCECriterion
from .utils import AverageMeter


def to_cuda(item):
    if isinstance(item[0], list):
        return [[x.cuda() for x in y] for y in item]
    elif isinstance(item, list):
        return [x.cuda() for x in item]
    return item.cuda()


def train(train_loader, model, optimizer, args, epoch, tb_logger=None):
    batch_time, data_time, losses = AverageMeter(), AverageMeter(), AverageMeter()
    model.train()
    if not next(model.parameters()).is_cuda:
        raise AvtError("train(): model must be on the MI355X (model.cuda()); the loss runs on the HIP kernels, no CPU fallback")
    criterion = InfoNCECriterion()
    # --train_graph: the device side of a step (forward, loss, backward, optimizer) captured once per batch shape as a HIP graph and
    # replayed (train_ops.GraphedStep) — for small batches, whose ~3400 launches the host issues slower than the device runs them
    # (config 5 on 8 GPUs: one item per rank).  One process only: a DDP all-reduce is not captured.
    graphed = model.__dict__.setdefault("_avt_graphed_steps", {})  # (kept on the model: train() is called once per epoch)
    if graphed.get("optimizer") is not optimizer:
        graphed.clear()
        graphed["optimizer"] = optimizer
    use_graph = bool(getattr(args, "train_graph", 0)) and not (torch.distributed.is_available() and torch.distributed.is_initialized()
                                                               and torch.distributed.get_world_size() > 1)
    end = time.time()
    for i, batch_data in enumerate(train_loader):
        q_frames, q_audio_wav, q_audio_eg, t_frames, t_audio_wav, t_audio_eg = batch_data
        q_frames, t_frames = to_cuda(q_frames), to_cuda(t_frames)
        q_audio_eg, t_audio_eg = q_audio_eg.cuda(), t_audio_eg.cuda()
        data_time.update(time.time() - end)

        batch_size = (q_frames[0] if isinstance(q_frames, list) else q_frames).shape[0]
        if use_graph:
            flat = (list(q_frames) if isinstance(q_frames, list) else [q_frames]) + (list(t_frames) if isinstance(t_frames, list) else [t_frames]) + \
                   [q_audio_eg, t_audio_eg]
            key = tuple((tuple(v.shape), v.dtype) for v in flat)
            ent = graphed.get(key)
            if ent is None:
                static = [v.clone() for v in flat]
                nq = len(q_frames) if isinstance(q_frames, list) else 0
                nt = len(t_frames) if isinstance(t_frames, list) else 0

                def device_step(static=static, nq=nq, nt=nt, batch_size=batch_size):
                    qf = static[:nq] if nq else static[0]
                    tf = static[max(nq, 1) : max(nq, 1) + nt] if nt else static[max(nq, 1)]
                    return _eager_step(model, optimizer, criterion, args, qf, tf, static[-2], static[-1], batch_size)

                ent = graphed[key] = (static, train_ops.GraphedStep(device_step, static[0].device, warmup=2))
                # (the warm-up steps and the capture consumed this batch: three optimizer steps on it, like a first batch seen thrice)
            else:
                for dst, src in zip(ent[0], flat):
                    dst.copy_(src, non_blocking=True)
            loss = ent[1]()
            losses.update(loss.item(), batch_size)
        else:
            loss = _eager_step(model, optimizer, criterion, args, q_frames, t_frames, q_audio_eg, t_audio_eg, batch_size)
            losses.update(loss.item(), batch_size)

        batch_time.update(time.time() - end)
        end = time.time()
        if i % args.print_freq == 0:
            print("Epoch: [{0}][{1}/{2}]\t"
                  "Time {batch_time.val:.3f} ({batch_time.avg:.3f})\t"
                  "Data {data_time.val:.3f} ({data_time.avg:.3f})\t"
                  "Loss {loss.val:.4f} ({loss.avg:.4f})".format(epoch, i, len(train_loader), batch_time=batch_time,
                                                                data_time=data_time, loss=losses))
        if tb_logger is not None and i % args.log_freq == 0:
            logs = OrderedDict()
            logs["Train_IterLoss"] = losses.val
            iter_count = epoch * len(train_loader) + i
            for key, value in logs.items():
                tb_logger.log_scalar(value, key, iter_count)
            tb_logger.flush()
    return losses.avg


def _eager_step(model, optimizer, criterion, args, q_frames, t_frames, q_audio_eg, t_audio_eg, batch_size):
    """One optimizer step of train(): forward (train.py:114-116), InfoNCE + CE, backward, step -> the loss tensor."""
    groups = getattr(args, "bn_replicas", 1)
    groups = batch_size if groups < 0 else groups
    if groups > 1 and batch_size % groups:  # (a short last batch: say so instead of changing the BatchNorm semantics silently)
        if not getattr(train, "_warned_groups", False):
            print("train(): --bn_replicas %d does not divide a batch of %d items: that batch is normalised as ONE group "
                  "(use a batch size the replicas divide, or drop_last)" % (groups, batch_size))
            train._warned_groups = True
        groups = 1
    with train_ops.bn_replicas(groups):
        output = model(q_frames, t_frames, q_audio_eg=q_audio_eg, t_audio_eg=t_audio_eg)  # train.py:114-116
    labels = torch.zeros(batch_size, dtype=torch.long, device=output.device)  # positives at column 0
    loss = criterion(output, labels).mean()
    optimizer.zero_grad()
    loss.backward()
    optimizer.step()
    train_ops.invalidate_weight_cache()  # (optimizers that update through .data do not bump Tensor._version)
    return loss.detach()
