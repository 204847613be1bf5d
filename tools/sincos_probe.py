"""Accuracy of the three device sin/cos flavours vs float64 (GPU box)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from adaptive_optics_gym_amd import _lib
lib = _lib.load()
n = 1 << 22
g = torch.Generator("cuda").manual_seed(0)
for span in (0.5, 4.0, 40.0):
    u = (torch.rand(n, device="cuda", generator=g) * 2 - 1) * span
    ud = u.double()
    rs, rc = torch.sin(2 * np.pi * ud), torch.cos(2 * np.pi * ud)
    for fl, name in ((0, "poly"), (1, "hw"), (2, "hwraw")):
        s = torch.empty_like(u); c = torch.empty_like(u)
        _lib.check(lib.aog_selftest_sincos(C.c_void_p(u.data_ptr()), C.c_void_p(s.data_ptr()), C.c_void_p(c.data_ptr()), n, fl, None))
        torch.cuda.synchronize()
        es, ec = s.double() - rs, c.double() - rc
        print(f"span +-{span:5.1f} rev {name:6s}: max|ds|={es.abs().max():.3e} max|dc|={ec.abs().max():.3e} "
              f"mean ds={es.mean():+.3e} mean dc={ec.mean():+.3e} rms={es.pow(2).mean().sqrt():.3e} "
              f"corr(ds,sin)={(es*rs).mean():+.3e} corr(dc,cos)={(ec*rc).mean():+.3e}", flush=True)
