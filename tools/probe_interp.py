"""Per-jump cost of the SuperSloMo interpolation (avtex.slowmo.Interpolator) on the device against the oracle's fp32 torch
CPU run of the same networks, at the frame sizes of the bench (128^2) and of the encoders (224^2).
    python tools/probe_interp.py [SF]"""
import sys
import time

import torch

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from interp_weights import frame_pair, unet_state  # noqa: E402
from oracle import interp_ref  # noqa: E402

from avtex import slowmo  # noqa: E402


def main():
    sf = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    dev = "cuda:0"
    fc, at = unet_state(6, 4, 1, head_gain=20.0), unet_state(20, 5, 2, head_gain=5.0)
    for h, w in ((128, 128), (224, 224), (512, 512)):
        f0, f1 = frame_pair(1, h, w)
        it = slowmo.Interpolator(h, w, sf, dev)
        it.flow_comp.load_state_dict(fc)
        it.arb_time.load_state_dict(at)
        a, b = f0.to(dev), f1.to(dev)
        for _ in range(3):
            out = it(a, b)
        torch.cuda.synchronize()
        t0 = time.time()
        n = 20
        for _ in range(n):
            out = it(a, b)
        torch.cuda.synchronize()
        gpu_ms = (time.time() - t0) / n * 1e3
        cpu_ms = None
        if h <= 224:
            t0 = time.time()
            ref = interp_ref.interpolate_pair(fc, at, f0, f1, sf)
            cpu_ms = (time.time() - t0) * 1e3
            d = (out.cpu().int() - ref.int()).abs()
            agree = "max |diff| %d on %.5f of the pixels" % (int(d.max()), float((d > 0).float().mean()))
        # 2 * MACs of both UNets: flowComp once, ArbTime (sf - 1) times
        def flops(cin, cout):
            w_ = (32, 64, 128, 256, 512, 512)
            f = (cin * 32 * 49 + 32 * 32 * 49) * h * w
            for i, k in enumerate((5, 3, 3, 3, 3)):
                hw = (h >> (i + 1)) * (w >> (i + 1))
                f += (w_[i] * w_[i + 1] + w_[i + 1] * w_[i + 1]) * k * k * hw
            for i, (a_, b_) in enumerate(((512, 512), (512, 256), (256, 128), (128, 64), (64, 32))):
                hw = (h >> (4 - i)) * (w >> (4 - i))
                f += (a_ * b_ + 2 * b_ * b_) * 9 * hw
            return 2.0 * (f + 32 * cout * 9 * h * w)
        fl = flops(6, 4) + (sf - 1) * flops(20, 5)
        print("%dx%d SF=%d: %.2f ms per jump on the device (%.1f TFLOP/s algorithmic)%s" % (
            h, w, sf, gpu_ms, fl / gpu_ms / 1e9,
            "" if cpu_ms is None else "; oracle on %d CPU threads %.0f ms (x%.0f); %s" % (torch.get_num_threads(), cpu_ms, cpu_ms / gpu_ms, agree)))


if __name__ == "__main__":
    main()
