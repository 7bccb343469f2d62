"""Times nof_hash_encode_bwd and its level groups on a real cfg2 training batch (run on the GPU box).

Round-2 measurements made with temporary knobs in the scatter kernel (16-keyframe pool, fp16x3; whole call 497 us):
  no emission at all 459 us (the kernel is NOT bound by the atomic rate: removing every atomic saves 8 %);
  scatter set-up + LDS staging only ~100 us for the 15 levels; run sums + row de-duplication + emission ~310 us;
  without the row de-duplication (more atomics) 807 us -- the atomic count matters as soon as it grows;
  2 / 3 / 4 persistent workgroups per CU: 755 / 585 / 484 us (latency-bound: time ~ 1 / resident waves; 5 do not fit the LDS);
  back to back instead of overlapped with dL/dx and the level-0 kernel: 640 us;
  prefetching the next tile's point and gradient, and dropping the per-tile 64-bit division: no change;
  run sums in one LDS round trip (16 independent loads per lane + three shuffles for runs that cross a quarter) with a
  pairwise fold instead of the owner chain: 570 us (more atomics than the chain); with the owner chain kept: 549 us."""
import sys, os, ctypes as C, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import numpy as np
import bench
from bundlesdf_amd import lib

args = argparse.Namespace(keyframes=16, height=480, width=640, rays=4096, log2_T=19, mlp='baseline', precision='fp16x3', finest=256)
torch.cuda.set_device(0)
runner, cfg = bench.build_runner(args, 0, 1, torch.device('cuda', 0))
fld = runner.field
for _ in range(20):
    runner.train_loop(); runner.global_step += 1
torch.cuda.synchronize()
R, S = 4096, 192
b = fld._buffers(R, S)
B = R * S
g = fld.grid


def timeit(fn, n=7):
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return float(np.median(ts)) * 1e3


gt = torch.zeros(fld.n_entries, 2, device='cuda')
tag = 'default'
full = timeit(lambda: lib.call('nof_hash_encode_bwd', C.byref(g), b['pts_w'], fld.table, b['dfeat'], gt, b['dpts'], B))
nodx = timeit(lambda: lib.call('nof_hash_encode_bwd', C.byref(g), b['pts_w'], fld.table, b['dfeat'], gt, None, B))
fine = timeit(lambda: lib.call('nof_hash_encode_bwd_levels', C.byref(g), b['pts_w'], fld.table, b['dfeat'], gt, None, 9, 16, B))
mid = timeit(lambda: lib.call('nof_hash_encode_bwd_levels', C.byref(g), b['pts_w'], fld.table, b['dfeat'], gt, None, 1, 9, B))
l0 = timeit(lambda: lib.call('nof_hash_encode_bwd_levels', C.byref(g), b['pts_w'], fld.table, b['dfeat'], gt, None, 0, 1, B))
print(f'[{tag}] whole call {full:.0f} us | without dL/dx {nodx:.0f} | levels 9-15 (hashed) {fine:.0f} | levels 1-8 (dense) {mid:.0f} | level 0 (LDS) {l0:.0f}')
