"""Per-level error of nof_hash_encode_bwd against the oracle (random points and points along rays); run on the GPU box."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests import util as U
from oracle import nof_oracle as O
from bundlesdf_amd import lib as nof

def run(tag, pts, L, T, finest):
    g, geo = U.make_grids(nof, L=L, T=T, finest=finest)
    B = len(pts)
    torch.manual_seed(0)
    table = (torch.rand(geo.n_entries, 2) * 2 - 1) * 0.1
    tt = table.clone().requires_grad_(True)
    ref = O.hash_encode(torch.from_numpy((pts + 1) / 2), tt, geo)
    torch.manual_seed(1)
    dy = torch.randn(B, L * 2)
    ref.backward(dy)
    dfeat = dy.reshape(B, L, 2).permute(1, 0, 2).contiguous().cuda()
    gtab = torch.zeros(geo.n_entries, 2, device='cuda')
    nof.call('nof_hash_encode_bwd', C.byref(g), U.dev(pts), table.cuda(), dfeat, gtab, None, B)
    torch.cuda.synchronize()
    err = (gtab.cpu() - tt.grad).abs()
    off = list(geo.offsets) + [geo.n_entries]
    print(tag, 'max err', float(err.max()), 'ref max', float(tt.grad.abs().max()))
    for l in range(L):
        e = err[off[l]:off[l + 1]]
        print(f'  level {l} size {off[l+1]-off[l]} err {float(e.max()):.3e} sum got {float(gtab[off[l]:off[l+1]].sum()):.5f} ref {float(tt.grad[off[l]:off[l+1]].sum()):.5f}')

rng = np.random.default_rng(0)
pts = U.test_points(3000, seed=16)
run('random', pts, 16, 14, 256)
o = rng.uniform(-0.9, 0.9, size=(64, 1, 3)); d = rng.normal(size=(64, 1, 3)); d /= np.linalg.norm(d, axis=-1, keepdims=True)
t = np.linspace(0, 0.8, 192)[None, :, None]
rays = (o + d * t).reshape(-1, 3).astype(np.float32)
run('rays', rays, 16, 19, 256)
