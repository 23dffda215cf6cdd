"""Import shim for the reference's Python loss modules (THIS container only).

TEST INFRASTRUCTURE ONLY — used by tests/golden/make_golden.py to capture
input/output vectors from the real `model/networks.py`.  /root/reference does
not exist on the GPU box; nothing at run time depends on this file.

The reference's `torchext/functions.py:2-3` imports `ext_cpu` / `ext_cuda` from
its own (read-only, unbuilt) package directory, so a synthetic `torchext`
package is registered whose `ext_cpu` is the module built by build_ref.py and
whose `ext_cuda` is an empty stub; `functions.py` and `modules.py` are then
executed from their real paths.  `TimedModule.forward` calls
`torch.cuda.synchronize()` unconditionally (model/networks.py:19-22), which is
made a no-op on this GPU-less host.
"""
import importlib.util
import os
import sys
import types

REF_ROOT = "/root/reference"


def available():
    return os.path.isdir(os.path.join(REF_ROOT, "torchext"))


def load():
    """Returns (torchext_module, networks_module) backed by the real reference."""
    import torch
    from . import build_ref

    ext_cpu = build_ref.load()
    if ext_cpu is None:
        raise RuntimeError("reference tree not present")

    pkg_dir = os.path.join(REF_ROOT, "torchext")
    pkg = types.ModuleType("torchext")
    pkg.__path__ = [pkg_dir]
    sys.modules["torchext"] = pkg
    sys.modules["torchext.ext_cpu"] = ext_cpu
    pkg.ext_cpu = ext_cpu
    stub = types.ModuleType("torchext.ext_cuda")
    sys.modules["torchext.ext_cuda"] = stub
    pkg.ext_cuda = stub

    for name in ("functions", "modules"):
        spec = importlib.util.spec_from_file_location(
            "torchext." + name, os.path.join(pkg_dir, name + ".py"))
        mod = importlib.util.module_from_spec(spec)
        sys.modules["torchext." + name] = mod
        spec.loader.exec_module(mod)
        setattr(pkg, name, mod)
        for k, v in vars(mod).items():
            if not k.startswith("_"):
                setattr(pkg, k, v)

    if not torch.cuda.is_available():
        torch.cuda.synchronize = lambda *a, **k: None

    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    import matplotlib
    matplotlib.use("Agg")
    from model import networks
    return pkg, networks
