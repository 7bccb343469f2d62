"""Times nof_hash_encode_bwd and its level groups on a real cfg2 training batch (run on the GPU box).

Round-2 measurements, made with temporary knobs in the scatter kernel (16-keyframe pool, fp16x3):
 first kernel of the round (all 64 lanes' products parked in LDS, run sums + row de-duplication there; whole call 497 us):
  rocprofv3: ~1000 VALU instructions per (64 samples, level) tile = ~300 us of VALU issue; 2 / 3 / 4 persistent workgroups per
  CU: 755 / 585 / 484 us; prefetching the next tile, one-round-trip LDS sums, a pairwise fold: no gain.
 rewritten kernel (run totals by a DPP segmented scan in registers, geometric chain links, ~400 VALU per tile):
  no emission at all 175 us | everything but the atomics 216 us (plain stores instead of atomics: 226) | with atomics 444 us
  (profiles/r02_d_hash_probe_a.txt, _b.txt): the kernel is now bound by the ATOMIC RATE of the memory side;
  the emission repeated 4x: + 268 us per pass for the hashed levels 9-15, + 117 us for the dense levels 1-8 = 385 us of atomic
  time per pass for 6.74 M line requests (tools/scatter_requests.py) = 17.5 G/s in situ (tools/atomic_probe.py: 20.8 G/s);
  persistent workgroups per CU 1 / 2 / 3 / 4 / 8: whole call 646 / 439 / 474 / 525 / 537 us, bench step 1.187 / 0.960 / 0.999 /
  - / 1.062 ms (profiles/r02_d_hash_probe_c.txt): two per CU saturate the atomic units, more only take L2 bandwidth from
  k_hash_dx beside it.
 tools/overlap_probe.py: the scatter does not overlap with the gather / MLP kernels of another ray chunk on a second stream
  (both = sum of the two, whatever the workgroup count): atomics occupy the L2 channels everything else needs, so pipelining
  the step over ray chunks buys nothing."""
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
if os.environ.get('NOF_DUMP_PTS'):
    os.makedirs('gpurun_out', exist_ok=True)
    np.save('gpurun_out/cfg2_pts_w_512rays.npy', b['pts_w'][:512 * S].cpu().numpy())
