"""Time of csrc/wgrad_x3.hip on four representative SlowFast layers (run it against libraries built with
-DAVT_WGRAD_DBG_CONST=0/1/2/4 via AVT_HIP_LIB to see which phase the time hangs on; profiles/r02/probe_wgrad_phases.log)."""
import sys, time, torch
sys.path.insert(0, ".")
from avtex import ops
dev = "cuda:0"
BATCH = int(sys.argv[1]) if len(sys.argv) > 1 else 15  # clips per launch (120 = a rank's items as one batch)
for (cin, cout, k, s, p, xs) in [(1024, 256, (3, 1, 1), (1, 1, 1), (1, 0, 0), (BATCH, 1024, 8, 14, 14)),
                                 (256, 256, (1, 3, 3), (1, 1, 1), (0, 1, 1), (BATCH, 256, 8, 14, 14)),
                                 (64, 64, (1, 3, 3), (1, 1, 1), (0, 1, 1), (BATCH, 64, 8, 56, 56)),
                                 (256, 1024, (1, 1, 1), (1, 1, 1), (0, 0, 0), (BATCH, 256, 8, 14, 14))]:
    x = torch.randn(xs, device=dev).contiguous(memory_format=torch.channels_last_3d)
    w = torch.randn(cout, cin, *k, device=dev).contiguous(memory_format=torch.channels_last_3d)
    y = torch.nn.functional.conv3d(x, w, stride=s, padding=p)
    dy = torch.randn_like(y).contiguous(memory_format=torch.channels_last_3d)
    dw = torch.empty_like(w)
    f = lambda: ops.conv3d_wgrad_x3_f32(dy.permute(0, 2, 3, 4, 1), x.permute(0, 2, 3, 4, 1), dw.permute(0, 2, 3, 4, 1), (xs[0], xs[2], xs[3], xs[4]), cin, cout, k, s, p, cin, cout)
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(10): f()
    torch.cuda.synchronize(); ms = (time.time() - t0) / 10 * 1e3
    fl = 2.0 * y.numel() * cin * k[0] * k[1] * k[2]
    print("cin%d cout%d k%s: %.3f ms %.0f TF/s" % (cin, cout, k, ms, fl / ms / 1e9))
