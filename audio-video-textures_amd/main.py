"""CLI / per-video driver, drop-in for contrastive_video_textures/main.py:41-548: every flag of the
reference parser with the same names, defaults and types; `main(args, video_name, itr=0)`; the per-video
loop with the fps -> window/stride override (main.py:511-516) and default checkpoint name (:520-534).

Additions (all optional): --stitch_mode {compat,aligned}, --ref_num_gpus, --enc_dtype {fp32,bf16},
--enc_batch, --vcam (the flag validate.py:299 reads but the reference never defines [quirk Q2]).
Multi-GPU: one process per GPU under torch.distributed.run instead of torch.nn.DataParallel (main.py:420).
"""
import argparse
import math
import os
import shutil

import torch

from . import dist as avt_dist
from .dataset import AudioVideoSegments
from .logger import Logger
from .models import ContrastivePredictionTemporal, ModelBuilder3D
from .train import train
from .validate import read_video, validate
from .vggish import VGGish


def build_parser():
    parser = argparse.ArgumentParser(description="PyTorch Video Textures (MI355X-native hot path)")
    a = parser.add_argument
    a("--enc_arch", "-ea", metavar="ARCH", default="resnet18", help="model architecture")
    a("--model_type", "-m", default=1, type=int, help="(1) Video Textures (2) Audio Video Textures")
    a("--vdata", "-vdata", default=None, type=str, help="Path to video dataset")
    a("--adata", "-adata", default=None, type=str, help="Path to audio")
    a("--pdata", "-pdata", default=None, type=str, help="Path to poses")
    a("--fdata", "-fdata", default=None, type=str, help="Path to flow")
    a("--dadata", "-dadata", default="audio/target", type=str, help="Path to driving audio dataset")
    a("--video_list", "-vl", default=None, type=str, nargs="+", help="list of input videos")
    a("--fps", "-fps", default=30, type=int, help="frame rate of input video")
    a("--subsample_rate", "-subsample", default=1, type=int, help="rate for subsampling the video")
    a("--temp", "-temp", default=0.1, type=float, help="Temperature value")
    a("--threshold", "-th", default=0.0, type=float, help="Threshold value")
    a("--l2", "-l2", default=True, action="store_false", help="To use l2 norm or not")
    a("--interpolation", "-nintp", default=True, action="store_false", help="Interpolate frames at eval")
    a("--img_size", "-size", default=224, type=int, help="resize image to this size")
    a("--n_negs", "-negs", default=20, type=int, help="Number negative frames to use when training")
    a("--window", "-w", default=20, type=int, help="Size of temporal window")
    a("--train_stride", "-train_stride", default=4, type=int, help="Stride length")
    a("--stride", "-stride", default=4, type=int, help="Stride length")
    a("--new_video_length", "-nvl", default=30, type=int, help="Length of new video")
    a("--alpha", "-alpha", default=0.5, type=float, help="alpha for validation to control driving audio")
    a("--SF", "-SF", default=5, type=int, help="slomo factor N")
    a("--train_layout", default="ndhwc", choices=["ndhwc", "ncdhw"],
      help="memory layout of the encoders in training: ndhwc = channels_last_3d (+ fused BatchNorm passes), ncdhw = torch default")
    a("--bn_replicas", default=1, type=int,
      help="train(): normalise the rank's batch as this many equal groups of items, each with its own BatchNorm statistics — "
           "what the reference's DataParallel gives every GPU's share of the batch (main.py:420), running statistics from "
           "the first group only as DataParallel keeps replica 0's buffers; 1 = over the rank's whole batch; -1 = one group "
           "per item (batch 8 on 8 GPUs in the reference)")
    a("--train_graph", default=0, type=int, choices=[0, 1],
      help="1: train() captures the device side of a step (forward, loss, backward, optimizer) once per batch shape as a HIP graph and "
           "replays it (train_ops.GraphedStep) — for small batches whose launches the host issues slower than the device runs them; one "
           "process only; the first batch of a shape also serves the two warm-up steps (not in the reference)")
    a("--train_conv", default="x3", choices=["x3", "fp32"],
      help="arithmetic of the training convolutions (with --train_layout ndhwc): x3 = split-plane MFMA kernels, fp32 "
           "accumulation, forward 2^-22 / gradients 2^-16 per product (default; train_ops.py); fp32 = MIOpen's fp32 "
           "convolutions, the reference's arithmetic (train.py:114-141)")
    a("--slomo_ckpt", default="ckpt/SuperSloMo.ckpt", type=str,
      help="SuperSloMo checkpoint (validate.py:183 hard-codes this path); 'random' = seeded weights; missing file = cuts")
    a("-long", "--long", dest="long", default=False, action="store_true", help="unused in the reference")
    a("-fb", "--frames_bar", dest="frames_bar", default=False, action="store_true", help="Visualize transitions.")
    a("--epochs", default=60, type=int, metavar="N", help="number of total epochs to run")
    a("--size", default=224, type=int, metavar="N", help="primary image input size")
    a("--start_epoch", default=None, type=int, metavar="N", help="manual epoch number (useful on restarts)")
    a("--batch_size", "-bs", default=32, type=int, metavar="N", help="mini-batch size (default: 32)")
    a("--mini_batchsize", "-mbs", default=150, type=int, help="mini-batch size for target frames")
    a("--lr", "-lr", default=10e-3, type=float, metavar="LR", help="initial learning rate")
    a("--lr_steps", default=30, type=int, metavar="LRSteps", help="epochs to decay learning rate by 10")
    a("--momentum", default=0.9, type=float, metavar="M", help="momentum")
    a("--weight_decay", "--wd", default=0.0001, type=float, metavar="W", help="weight decay (default: 1e-4)")
    a("--workers", "-j", default=4, type=int, metavar="N", help="number of data loading workers")
    a("--print_freq", "-p", default=5, type=int, metavar="N", help="print frequency")
    a("--log_freq", "-lf", default=10, type=int, metavar="N", help="frequency to write in tensorboard")
    a("--resume", default="", type=str, metavar="PATH", help="path to latest checkpoint (default: none)")
    a("-e", "--evaluate", dest="evaluate", action="store_true", help="evaluate model on validation set")
    a("-da", "--driving_audio", default=None, type=str, nargs="+", help="list of target audios")
    a("-daf", "--da_feats", default="VGG", type=str, help="type of feats for audio conditioning")
    a("-daf_resume", "--daf_resume", default="", type=str, nargs="+", help="List of paths to best VideoForAudio ckpt")
    a("-ve", "--visualize_evaluate", dest="visualize_evaluate", action="store_true",
      help="evaluate model on validation set and visualize logits")
    a("-vf", "--val_freq", default=5, type=int, metavar="VF", help="frequency to call validate during train)")
    a("--logdir", default="./logs", help="folder to output tensorboard logs")
    a("--logname", default="exp", help="name of the experiment for checkpoints and logs")
    a("-rf", "--results_folder", default="results", type=str, help="folder for result videos")
    a("--ckpt", default="./ckpt", help="folder to output checkpoints")
    # --- additions of the MI355X build ---
    a("--stitch_mode", default="compat", choices=["compat", "aligned"],
      help="compat: the shipped reference's window/label map; aligned: the N x N matrix the labels claim")
    a("--ref_num_gpus", default=None, type=int, help="reference GPU count to emulate in compat mode")
    a("--enc_dtype", default="fp32", choices=["fp32", "bf16", "bf16x3", "f16x3"],
      help="encoder arithmetic at -e: fp32 (default; on the MFMA kernels = the contract-grade split-plane mode f16x3), "
           "bf16 (fast path: 5x the throughput, scores off by up to 1e-1), or a split-plane mode by name")
    a("--sim_precision", default="f32", choices=["f32", "bf16x3", "bf16"],
      help="similarity arithmetic of the SHARDED aligned build (world > 1): f32 = exact, bit-identical to one GPU (all-gathers "
           "fp32 rows); bf16x3 = bf16 hi/lo planes through the MFMA bf16 pipe (all-gathers the planes, scores within 5e-6); "
           "bf16 = one plane (outside the 1e-3 score contract)")
    a("--enc_batch", default=249, type=int,
      help="windows per encoder batch (83 k: whole rounds of the 256 x 256 tile on the 256 CUs; 166 measured +2 %% over 83, "
           "249 +1.3-2 %% more; ~37 GB of activations at 224^2).  Cut to texture.max_enc_batch(img_size) — the kernels' signed "
           "32-bit element offsets allow 267 clips at 224^2, 204 at 256^2; the bf16 path's dense clips 166 at 224^2")
    a("--enc_impl", default="auto", choices=["auto", "mfma", "module"],
      help="SlowFast at -e: hand-written MFMA convolutions (auto/mfma) or the nn.Module on MIOpen (module)")
    a("--dump_png", default=False, action="store_true",
      help="also write the reference's per-frame PNG folder (validate.py:789-792); default: frames go to the encoder directly")
    a("--vcam", default=False, action="store_true", help="defined for validate.py:299; CAM dumps are out of scope")
    return parser


parser = build_parser()


def main(args, video_name, itr=0):
    best_loss = 1000000
    rank, world, local = avt_dist.init_from_env()
    device = torch.device("cuda", local)
    torch.cuda.set_device(device)
    if not args.evaluate and not args.visualize_evaluate:
        dataset_train = AudioVideoSegments(args, video_name, split="train")
        sampler = torch.utils.data.distributed.DistributedSampler(dataset_train) if world > 1 else None
        train_loader = torch.utils.data.DataLoader(dataset_train, batch_size=max(args.batch_size // world, 1),
                                                   shuffle=sampler is None, sampler=sampler,
                                                   num_workers=args.workers, drop_last=True)
    print("=> creating model '{}'".format(args.model_type))
    builder = ModelBuilder3D()
    q_image_enc_model, fc_dim = builder.build_network(arch=args.enc_arch, img_size=args.size, window=args.window)
    t_image_enc_model, fc_dim = builder.build_network(arch=args.enc_arch, img_size=args.size, window=args.window)
    audio_enc_model = VGGish()
    if os.path.isfile("pytorch_vggish.pth"):  # main.py:338 loads it unconditionally
        audio_enc_model.load_state_dict(torch.load("pytorch_vggish.pth", map_location="cpu"))
    else:
        print("pytorch_vggish.pth not found in cwd: VGGish keeps its random init")
    model = ContrastivePredictionTemporal(q_image_enc_model, t_image_enc_model, audio_enc_model, args.model_type,
                                          fc_dim, args.temp, args.window, args.stride, args.threshold,
                                          mini_batchsize=args.mini_batchsize, enc_arch=args.enc_arch,
                                          img_size=args.img_size)
    if args.resume:
        assert os.path.isfile(args.resume), "No checkpoint found at '{}'".format(args.resume)
        print("=> loading checkpoint '{}'".format(args.resume))
        checkpoint = torch.load(args.resume, map_location="cpu")
        if args.start_epoch is None:
            args.start_epoch = checkpoint["epoch"]
        best_loss = checkpoint["best_loss"]
        model.load_state_dict(checkpoint["state_dict"])
        print("=> loaded checkpoint '{}' (epoch {})".format(args.resume, checkpoint["epoch"]))
    os.makedirs("./ckpt", exist_ok=True)
    tag = "{}_model_{}_vd_{}_vn_{}_bs_{}_".format(args.logname, args.model_type, os.path.split(args.vdata)[-1],
                                                  video_name, args.batch_size)
    if not args.evaluate:
        tag += "negs_{}_".format(args.n_negs)
    logname = tag + "w_{}_stride_{}_temp_{}_th_{}_enca_{}_subr_{}_eval_{}".format(
        args.window, args.stride, args.temp, args.threshold, args.enc_arch, args.subsample_rate,
        args.evaluate or args.visualize_evaluate)
    if args.evaluate and args.driving_audio is not None:
        logname += "alpha_{}_daf_{}".format(args.alpha, args.da_feats)
    if args.start_epoch is None:
        args.start_epoch = 0

    model = model.to(device)
    if args.evaluate and args.enc_dtype == "bf16" and args.enc_impl == "module":
        for enc in (model.q_encoder, model.t_encoder):
            enc.to(torch.bfloat16).to(memory_format=torch.channels_last_3d)
    if not args.evaluate and getattr(args, "train_layout", "ndhwc") == "ndhwc":
        # training layout on the MI355X: channels-last convolution weights (MIOpen's fwd / dgrad / wgrad then run without
        # layout transposes) and the fused train-mode BatchNorm + shortcut + ReLU passes of csrc/bn_train.hip, which work
        # on channels-last rows (train_ops.bn_act; fp32 step 95 -> 130 clips/s).  The BatchNorm passes compute
        # what stock fp32 BatchNorm computes (fp64 statistics); the CONVOLUTIONS' arithmetic is --train_conv's choice: x3 (the
        # default) is split-plane MFMA, not bit-for-bit fp32 — printed so that a log says which one trained the model.
        from . import train_ops

        model = train_ops.training_layout(model)
        mode = train_ops.set_conv_mode(getattr(args, "train_conv", "x3"))
        if rank == 0:
            print("training convolutions: %s" % ("split-plane MFMA (x3: fp16 planes forward 2^-22, bf16 planes gradients 2^-16, "
                                                  "fp32 accumulation)" if mode == "x3" else "MIOpen fp32"))
    if world > 1 and not args.evaluate:  # weights resident per rank, gradients all-reduced over RCCL
        model = wrap_ddp(model, device, local)
    torch.backends.cudnn.benchmark = True
    tb_logdir = os.path.join(args.logdir, logname)
    os.makedirs(tb_logdir, exist_ok=True)
    tb_logger = Logger(tb_logdir) if rank == 0 else None

    if args.evaluate:
        # aligned mode: every rank takes part (its block of windows / rows, dist.sharded_survivors); compat mode keeps
        # the reference's per-step window map, which is one rank's work
        if rank == 0 or (world > 1 and args.stitch_mode == "aligned"):
            validate(model, args, video_name=video_name, tb_logger=tb_logger, model_type=args.model_type, itr=itr)
        return
    optimizer = torch.optim.SGD(params=model.parameters(), lr=args.lr, momentum=args.momentum,
                                weight_decay=args.weight_decay)
    scheduler = torch.optim.lr_scheduler.StepLR(optimizer, step_size=args.lr_steps)
    print("Training for {} epochs.".format(args.epochs - args.start_epoch))
    for epoch in range(args.start_epoch, args.epochs):
        if world > 1:
            train_loader.sampler.set_epoch(epoch)
        loss = train(train_loader, model, optimizer, args, epoch, tb_logger)
        is_best = loss < best_loss
        best_loss = min(loss, best_loss)
        if rank == 0:
            net = model.module if hasattr(model, "module") else model
            save_checkpoint({"epoch": epoch + 1, "arch": args.enc_arch, "state_dict": net.state_dict(),
                             "best_loss": best_loss}, is_best, os.path.join(args.ckpt, logname))
        scheduler.step()
        if loss < 0.07:
            print("Loss {}. Stopping at epoch {}.".format(loss, epoch))
            break


def wrap_ddp(model, device, local):
    """DistributedDataParallel over RCCL in place of the reference's DataParallel (main.py:420): weights resident per rank,
    gradients all-reduced.  Modules that exist only so reference checkpoints load by key and are never applied — q_a_mlp /
    t_a_mlp (models.py:267-284) and VGGish's fc stack (vggish.py:45) — are frozen first: DDP's reducer would otherwise wait
    for gradients that never come (DataParallel tolerated them), and ~300 M dead parameters would be all-reduced every
    step.  find_unused_parameters covers plugin encoders with dead parameters of their own."""
    net = model
    for name in ("q_a_mlp", "t_a_mlp"):
        if hasattr(net, name):
            getattr(net, name).requires_grad_(False)
    for name in ("q_a_encoder", "t_a_encoder"):
        enc = getattr(net, name, None)
        if isinstance(enc, VGGish) and hasattr(enc, "fc"):
            enc.fc.requires_grad_(False)
    ddp = torch.nn.parallel.DistributedDataParallel(model, device_ids=[local] if device.type == "cuda" else None,
                                                    find_unused_parameters=True)
    if device.type == "cuda":
        ddp.register_comm_hook(None, _allreduce_after_every_stream)
    return ddp


def _allreduce_after_every_stream(_state, bucket):
    """DDP's default all-reduce (mean over ranks), started only after EVERY stream of the training forward has produced its
    part of the bucket: the query encoder runs on a side stream (models.ContrastivePredictionTemporal.forward) and autograd
    replays its backward there, while the reducer orders the collective after the stream of the LAST gradient only."""
    import torch.distributed as dist
    from . import models
    buf = bucket.buffer()
    cur = torch.cuda.current_stream(buf.device)
    for s in models.training_side_streams(buf.device) + [torch.cuda.default_stream(buf.device)]:
        if s != cur:
            cur.wait_stream(s)
    buf.div_(dist.get_world_size())
    return dist.all_reduce(buf, async_op=True).get_future().then(lambda f: f.value()[0])


def save_checkpoint(state, is_best, filename):
    torch.save(state, filename + "_latest.pth.tar")
    if is_best:
        shutil.copyfile(filename + "_latest.pth.tar", filename + "_best.pth.tar")


def cli(argv=None):
    args = parser.parse_args(argv)
    print(args)
    assert os.path.exists(args.vdata), "No videos found at {}".format(args.vdata)
    if args.adata is not None and os.path.exists(args.adata):
        print("Audio found at {}".format(args.adata))
    if args.video_list is None:
        args.video_list = sorted([f.split(".")[0] for f in sorted(os.listdir(args.vdata)) if not f.startswith(".")])
    for itr, video_name in enumerate(args.video_list):
        args.results_folder = "results_{}".format(video_name)
        if args.evaluate or args.visualize_evaluate:
            _, fps = read_video(os.path.join(args.vdata, "{}.mp4".format(video_name)))
            if fps:
                args.fps = fps
            print("Frame rate: ", args.fps)
            args.window = math.ceil(args.fps / 2)  # [quirk Q10] overrides -w / -stride (main.py:515-516)
            args.stride = math.ceil(args.fps / 5)
            print("Stride {} Window {}".format(args.stride, args.window))
            if args.resume == "":
                args.resume = ("ckpt/exp_model_{}_vd_{}_vn_{}_bs_{}_negs_{}_w_{}_"
                               "stride_{}_temp_0.1_th_0.0_enca_{}_subr_{}_eval_False_best.pth.tar".format(
                                   args.model_type, os.path.split(args.vdata)[-1], video_name, args.batch_size,
                                   args.n_negs, args.window, args.stride, args.enc_arch, args.subsample_rate))
            assert os.path.isfile(args.resume), "No checkpoint found at '{}'".format(args.resume)
            print("=> loading checkpoint '{}'".format(args.resume))
            if args.driving_audio is not None:
                args.results_folder += "_target_{}_{}".format(
                    video_name, os.path.split(args.driving_audio[itr])[-1].split(".")[0])
        print("Starting video {}".format(video_name))
        main(args, video_name, itr)


if __name__ == "__main__":
    cli()
