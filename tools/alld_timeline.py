"""Per-chunk timeline of the all-D kernel from the -DCTD_STAMPS build (tools/build_variant.sh stamps -DCTD_STAMPS):
    python tools/alld_timeline.py tools/variants/libctd_stamps.so [norank|rank|plain] [D]
Every wavefront notes the shader clock when it ARRIVES at a chunk barrier of one pass and when it LEAVES it.  From that:
  busy(wave, chunk)  = arrive(chunk) - leave(chunk - 1)      cycles the wave needed for its chunk
  wait(wave, chunk)  = leave(chunk) - arrive(chunk)          cycles it sat at the barrier
  last(chunk)        = the wave that arrived last (who the others waited for)
printed per wave (mean over chunks and workgroups) with the SIMD class of the wave (wave % 4) and how often it was last."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import workloads
from connecting_the_dots_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
from connecting_the_dots_amd import torchext as te
mode = sys.argv[2] if len(sys.argv) > 2 else "rank"
D = int(sys.argv[3]) if len(sys.argv) > 3 else 128
L = _lib.lib()
raw = ctypes.CDLL(_lib.LIB_PATH)
H, W, N = 432, 512, 16
fr = torch.from_numpy(np.stack([workloads.uniform_frame(1234 + i, H, W) for i in range(N)])).cuda()
pat = torch.from_numpy(workloads.syn_dot_pattern(H, W, seed=42)[None, None]).cuda()
x, _ = te.lcn(fr, 5, 0.05)
p, _ = te.lcn(pat, 5, 0.05)
p = p[0].contiguous()

def call():
    if mode == "rank":
        return te.xcorrvol_argmax(x, p, D, 9, return_volume=True, algo="fast")
    if mode == "norank":
        return te.xcorrvol_argmax(x, p, D, 9, algo="fast")
    return te.xcorrvol_batch(x, p, D, 9, algo="fast")

for _ in range(int(os.environ.get("CTD_WARM_CALLS", "600"))):     # sustained clocks first
    v = call()
torch.cuda.synchronize()
wpw, chunks, spass = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
raw.ctd_debug_stamp_layout(ctypes.byref(wpw), ctypes.byref(chunks), ctypes.byref(spass))
n_wg = 256
buf = np.zeros(n_wg * wpw.value, dtype=np.uint32)
raw.ctd_debug_read_stamps.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
assert raw.ctd_debug_read_stamps(buf.ctypes.data, buf.nbytes) == 0
buf = buf.reshape(n_wg, wpw.value)
hdr, st = buf[:, :8], buf[:, 8:].reshape(n_wg, 16, chunks.value, 2).astype(np.int64)
n_chunks = int(hdr[0, 3])
dt_clk = (hdr[:, 4].astype(np.int64) - hdr[:, 0].astype(np.int64)) & 0xffffffff
dt_real = (hdr[:, 5].astype(np.int64) - hdr[:, 1].astype(np.int64)) & 0xffffffff
print("mode %s  D %d  stamped pass %d  chunks per pass %d" % (mode, D, spass.value, n_chunks))
print("workgroup lifetime: %.0f cycles (median), %.1f us by the 100 MHz counter -> shader clock %.0f MHz" % (
    np.median(dt_clk), np.median(dt_real) / 100.0, np.median(dt_clk / np.maximum(dt_real, 1)) * 100.0))
print("XCC ids seen:", sorted(set(int(v) for v in hdr[:, 2])))
n = min(n_chunks, chunks.value)
arr, lea = st[:, :, :n, 0], st[:, :, :n, 1]
busy = (arr[:, :, 1:] - lea[:, :, :-1]) & 0xffffffff          # [wg][wave][chunk 1..]
wait = (lea[:, :, 1:] - arr[:, :, 1:]) & 0xffffffff
period = (lea[:, :, 1:] - lea[:, :, :-1]) & 0xffffffff
ref = lea[:, :, :-1].min(axis=1, keepdims=True)
last = np.argmax((arr[:, :, 1:] - ref) & 0xffffffff, axis=1)        # [wg][chunk]: the wave that arrived last
print("chunk period (barrier to barrier): mean %.0f cycles, median %.0f, p10 %.0f, p90 %.0f" % (
    period.mean(), np.median(period), np.percentile(period, 10), np.percentile(period, 90)))
print("%-5s %-5s %10s %10s %10s %8s" % ("wave", "simd", "busy", "wait", "busy/per", "last %"))
for w in range(16):
    role = "load" if w == 15 else ""
    print("%-5d %-5d %10.0f %10.0f %10.3f %8.1f %s" % (w, w % 4, busy[:, w].mean(), wait[:, w].mean(),
                                                      busy[:, w].mean() / period[:, w].mean(), 100.0 * (last == w).mean(), role))
by_simd = [np.mean([busy[:, w].mean() for w in range(16) if w % 4 == c]) for c in range(4)]
print("mean busy by SIMD class (wave % 4):", ["%.0f" % v for v in by_simd])
# row position inside the pass: warm-up chunks (no output rows) vs output chunks
pc = period.mean(axis=(0, 1))
print("period by chunk index:", " ".join("%.0f" % v for v in pc))
