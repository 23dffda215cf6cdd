"""Does alternating steps over two streams hide the latency-bound small kernels of one step under the other's big ones?"""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, time
os.environ["CTD_NCC_ALGO"] = "fast"
from connecting_the_dots_amd import torchext as te
import bench
frames, pattern = bench.make_inputs(16, 0, torch.device("cuda"))
pl, _ = te.lcn(pattern, 5, 0.05); pl = pl[0].contiguous()
def step():
    x, _ = te.lcn(frames, 5, 0.05)
    return te.xcorrvol_argmax(x, pl, 128, 9, return_volume=True)
def run(nstreams, K=40):
    streams = [torch.cuda.Stream() for _ in range(nstreams)]
    keep = [None] * nstreams
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(K):
        with torch.cuda.stream(streams[i % nstreams]):
            keep[i % nstreams] = step()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / K * 1e3
for n in (1, 2, 3, 1, 2, 3):
    run(n, 6)
    print("streams %d: %.3f ms per step" % (n, run(n)))
