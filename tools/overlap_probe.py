"""Can the volume kernel (write-bound) and a volume sweep (read-bound) of another frame chunk overlap on two streams?"""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, time
from connecting_the_dots_amd import torchext as te
import bench
frames, pattern = bench.make_inputs(16, 0, torch.device("cuda"))
pl, _ = te.lcn(pattern, 5, 0.05); pl = pl[0].contiguous()
x, _ = te.lcn(frames, 5, 0.05)
vol_prev = te.xcorrvol_batch(x, pl, 128, 9, algo="fast")
sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
def seq():
    v = te.xcorrvol_batch(x, pl, 128, 9, algo="fast"); te.argmax_disp(vol_prev)
def par():
    with torch.cuda.stream(sA): v = te.xcorrvol_batch(x, pl, 128, 9, algo="fast")
    with torch.cuda.stream(sB): te.argmax_disp(vol_prev)
def only_vol(): te.xcorrvol_batch(x, pl, 128, 9, algo="fast")
def only_arg(): te.argmax_disp(vol_prev)
for name, fn in (("volume only", only_vol), ("argmax sweep only", only_arg), ("sequential", seq), ("two streams", par)):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): fn()
    torch.cuda.synchronize(); print("%-20s %.3f ms" % (name, (time.perf_counter() - t0) / 10 * 1e3))
