"""Where does the whole-network gradient gap between the fused train-mode BatchNorm passes and the stock torch ops come
from?  Ground truth = the same step in fp64 with the stock ops.  Prints, per fp32 variant (stock NCDHW, stock channels-last,
fused channels-last), the distance of its gradients to the fp64 ones, and for the fused run the per-call forward error of
every bn_act against the stock ops on the SAME input (so a wrong layer shape shows up by itself).
    python tools/probe_bn_train.py [hw] [batch]"""
import copy
import sys

import torch
import torch.nn as nn
import torch.nn.functional as F

sys.path.insert(0, ".")
from avtex import slowfast, train_ops  # noqa: E402


def main():
    hw = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    batch = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    dev = "cuda:0"
    torch.manual_seed(3)
    net = slowfast.SlowFast()
    for m in net.modules():
        if isinstance(m, nn.BatchNorm3d):
            nn.init.uniform_(m.weight, 0.5, 1.5)
            nn.init.uniform_(m.bias, -0.2, 0.2)
    state = copy.deepcopy(net.state_dict())
    clip = torch.randn(batch, 3, 32, hw, hw)

    run = [None]  # running statistics of the last step

    def step(dtype, cl, fused, spy=None):
        n = slowfast.SlowFast()
        n.load_state_dict(state)
        n = n.to(dev, dtype).train()
        fast = clip.to(dev, dtype)
        slow = fast[:, :, ::4].contiguous()
        if cl:
            n = n.to(memory_format=torch.channels_last_3d)
            fast = fast.contiguous(memory_format=torch.channels_last_3d)
            slow = slow.contiguous(memory_format=torch.channels_last_3d)
        train_ops._FUSED = 1 if fused else 0
        train_ops._CONV_X3 = 1 if fused == 2 else 0
        out = n([slow, fast])
        loss = (out.double() ** 2).mean()
        loss.backward()
        torch.cuda.synchronize()
        run[0] = {k: v.double().cpu() for k, v in n.state_dict().items() if "running" in k}
        return float(loss.detach()), {k: p.grad.double().cpu() for k, p in n.named_parameters() if p.grad is not None}

    l64, g64 = step(torch.float64, False, False)
    r64 = run[0]
    print("fp64 loss %.9f" % l64)
    for name, cl, fused in (("fp32 stock NCDHW", False, 0), ("fp32 stock channels-last", True, 0), ("fp32 FUSED BatchNorm channels-last", True, 1),
                            ("fp32 FUSED BatchNorm + x3 convolutions", True, 2)):
        l, g = step(torch.float32, cl, fused)
        rows = []
        num = den = 0.0
        for k in g64:
            d = float((g[k] - g64[k]).norm())
            r = float(g64[k].norm())
            rows.append((d / (r + 1e-30), k, r))
            num += d * d
            den += r * r
        rows.sort(reverse=True)
        print("%-28s loss rel err %.2e   all-gradient rel err %.2e   worst parameters:" % (name, abs(l - l64) / abs(l64), (num / den) ** 0.5))
        for r in rows[:5]:
            print("      %.3e  %-40s |g64| %.3e" % r)
        rr = sorted(((float((run[0][k] - r64[k]).abs().max()), float((run[0][k] - r64[k]).abs().max()) / (float(r64[k].abs().max()) + 1e-30), k,
                      float(r64[k].abs().max())) for k in r64), reverse=True)
        print("    running statistics after the step against fp64: worst abs err / same rel to the tensor's max / key / max|fp64|")
        for r in rr[:4]:
            print("      %.3e  %.3e  %-44s %.3e" % r)

    # per-call forward check inside a fused run
    calls = []
    real = train_ops.bn_act

    def spy(x, bn, res=None, relu=True):
        ok = train_ops.fusable(x, bn, res)
        ref_bn = copy.deepcopy(bn)
        y = real(x, bn, res=res, relu=relu)
        with torch.no_grad():
            e = ref_bn(x.double().to(memory_format=torch.contiguous_format)) if False else None
            rb = copy.deepcopy(ref_bn).double()
            z = rb(x.detach().double())
            if res is not None:
                z = z + res.detach().double()
            if relu:
                z = F.relu(z)
            s = ref_bn(x.detach())
            if res is not None:
                s = s + res.detach()
            if relu:
                s = F.relu(s)
            scale = float(z.abs().max()) + 1e-30
            calls.append((float((y.detach().double() - z).abs().max()) / scale, float((s.double() - z).abs().max()) / scale, ok,
                          tuple(x.shape), res is not None, relu, float(rb.running_var.min())))
        return y

    slowfast.bn_act = spy
    step(torch.float32, True, 2)
    slowfast.bn_act = real
    print("bn_act calls: %d, fused %d" % (len(calls), sum(c[2] for c in calls)))
    calls.sort(reverse=True)
    print("  worst forward errors against fp64 stock ops on the same input (fused | fp32 stock | fused? shape res relu min-var):")
    for c in calls[:8]:
        print("     %.2e | %.2e | %s %s res=%s relu=%s minvar=%.2e" % c)


if __name__ == "__main__":
    main()
