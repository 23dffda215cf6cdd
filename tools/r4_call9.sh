cd $GRAFT_REPO_ROOT
python -m pytest tests/test_lcn_gpu.py -x -q 2>&1 | tail -5
python - <<'PY'
import sys, time, numpy as np, torch
sys.path.insert(0, '.')
from connecting_the_dots_amd import torchext as te
x = torch.rand(16, 1, 432, 512, device='cuda')
for algo in ('exact', 'fast', 'exact', 'fast'):
    for _ in range(200): te.lcn(x, 5, 0.05, algo=algo)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(200): te.lcn(x, 5, 0.05, algo=algo)
    torch.cuda.synchronize(); print(algo, '%.2f us per call' % ((time.perf_counter() - t0) / 200 * 1e6))
PY
