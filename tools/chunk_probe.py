"""Does ranking a volume right after it was written (per chunk of frames) hit in the memory-side cache?"""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, time
from connecting_the_dots_amd import torchext as te
import bench
chunk = int(sys.argv[1])
frames, pattern = bench.make_inputs(16, 0, torch.device("cuda"))
pl, _ = te.lcn(pattern, 5, 0.05); pl = pl[0].contiguous()
x, _ = te.lcn(frames, 5, 0.05)
def run():
    for c in range(0, 16, chunk):
        te.xcorrvol_argmax(x[c:c + chunk], pl, 128, 9, algo="fast", return_volume=True)
for _ in range(3): run()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): run()
torch.cuda.synchronize(); print("chunk", chunk, "ms per 16 frames", (time.perf_counter() - t0) / 10 * 1e3)
