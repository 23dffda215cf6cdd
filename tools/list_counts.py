import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from connecting_the_dots_amd import _lib, torchext as te
from tests import workloads
H, W, D, N = 432, 512, 128, 16
L = _lib.lib()
fr = torch.from_numpy(np.stack([workloads.uniform_frame(1234 + i, H, W) for i in range(N)])).cuda()
pat = torch.from_numpy(workloads.syn_dot_pattern(H, W, seed=42)[None, None]).cuda()
x, _ = te.lcn(fr, 5, 0.05); p, _ = te.lcn(pat, 5, 0.05); p = p[0].contiguous()
ws = torch.empty(L.ctd_xcorrvol_argmax_workspace_bytes(N, 1, H, W, D, 9, 1), dtype=torch.uint8, device="cuda")
vol = torch.empty((N, D, H, W), device="cuda")
idx = torch.empty((N, H, W), dtype=torch.int64, device="cuda"); best = torch.empty((N, H, W), device="cuda")
st = L.ctd_xcorrvol_argmax_f32(x.data_ptr(), p.data_ptr(), 0, vol.data_ptr(), idx.data_ptr(), best.data_ptr(), N, 1, H, W, D, 9, 1,
                               1e-5, ws.data_ptr(), ws.numel(), 0, torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
al = lambda v, a: (v + a - 1) // a * a
Dpad = (D + 15) // 16 * 16 + 32; xoff = Dpad + 3; W1 = al(W + 4 + xoff, 4); Wp = al(W + 8, 4)
n0 = al(N * H * Wp * 4, 256); n1 = al(H * W1 * 4, 256)
c = ws[3 * n0 + 3 * n1: 3 * n0 + 3 * n1 + 16].view(torch.int32).tolist()
print("listed frame windows %d of %d, listed pattern windows %d of %d, listed run rows %d of %d" % (c[0], N * H * W, c[1], H * W1, c[2], H))
