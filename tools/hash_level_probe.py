"""Per-level cost of the hash-gradient scatter on a real training batch (run on the GPU box):
times nof_hash_encode_bwd on single-level grids cut out of the cfg2 grid, on the samples bench.py's runner draws."""
import sys, os, ctypes as C, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import numpy as np
import bench
from bundlesdf_amd import lib

ap = argparse.ArgumentParser()
ap.add_argument('--keyframes', type=int, default=16)
a = ap.parse_args()
args = argparse.Namespace(keyframes=a.keyframes, height=480, width=640, rays=4096, log2_T=19, mlp='baseline', precision='bf16')
torch.cuda.set_device(0)
runner, cfg = bench.build_runner(args, 0, 1, torch.device('cuda', 0))
fld = runner.field
for _ in range(20):
    runner.train_loop(); runner.global_step += 1
torch.cuda.synchronize()
R, S = 4096, 192
b = fld._buffers(R, S)
B = R * S
pts, dfeat = b['pts_w'], b['dfeat']
g = fld.grid
valid = b['valid'].float().mean().item()
print(f'B={B} valid={valid:.3f}')
x01 = (pts.view(-1, 3) + 1) * 0.5


def timeit(fn, n=5):
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))


gt = torch.zeros(fld.n_entries, 2, device='cuda')
print('whole bwd (with dpts): %.1f us' % (1e3 * timeit(lambda: lib.call('nof_hash_encode_bwd', C.byref(g), pts, fld.table, dfeat, gt, b['dpts'], B))))
print('whole bwd (no dpts):   %.1f us' % (1e3 * timeit(lambda: lib.call('nof_hash_encode_bwd', C.byref(g), pts, fld.table, dfeat, gt, None, B))))
tot = 0.0
for l in range(g.L):
    g1 = lib.NofHashGrid()
    g1.L, g1.C = 2, 2                      # make_hash_grid's floor; level 1 is a copy with an empty-gradient slice
    for k in (0, 1):
        g1.scale[k], g1.resolution[k], g1.size[k], g1.hashed[k] = g.scale[l], g.resolution[l], g.size[l], g.hashed[l]
        g1.offset[k] = 0 if k == 0 else g.size[l]
    d1 = torch.zeros(2, B, 2, device='cuda')
    d1[0] = dfeat.view(g.L, B, 2)[l]
    d1[1] = dfeat.view(g.L, B, 2)[l]
    gt1 = torch.zeros(2 * g.size[l], 2, device='cuda')
    ms = timeit(lambda: lib.call('nof_hash_encode_bwd', C.byref(g1), pts, fld.table, d1, gt1, None, B))
    # distinct entries touched at this level (cells of the samples' base corner)
    pos = x01 * g.scale[l] + 0.5
    cell = pos.floor().long().clamp_(0, g.resolution[l])
    key = (cell[:, 0] + cell[:, 1] * 1024 + cell[:, 2] * 1024 * 1024)
    nuniq = torch.unique(key).numel()
    tot += ms / 2
    print(f'level {l:2d} res {g.resolution[l]:4d} size {g.size[l]:7d} hashed {g.hashed[l]}  2 copies: {ms * 1e3:7.1f} us  -> {ms * 500:6.1f} us/level   '
          f'distinct base cells {nuniq:7d} ({B / nuniq:6.1f} samples/cell)')
print(f'sum over levels {tot * 1e3:.1f} us')
