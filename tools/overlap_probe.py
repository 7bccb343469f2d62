"""Can the atomic-bound table scatter run beside the MLP / gather kernels of ANOTHER ray chunk?  Times, on a real cfg2 batch:
the scatter alone, [hash fwd + MLP fwd + MLP bwd] alone, and both at once on two streams (independent buffers)."""
import sys, os, ctypes as C, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import bench
from bundlesdf_amd import lib

args = argparse.Namespace(keyframes=16, height=480, width=640, rays=4096, log2_T=19, mlp='baseline', precision='fp16x3', finest=256)
torch.cuda.set_device(0)
runner, cfg = bench.build_runner(args, 0, 1, torch.device('cuda', 0))
fld = runner.field
for _ in range(10):
    runner.train_loop(); runner.global_step += 1
torch.cuda.synchronize()
R, S = 4096, 192
b = fld._buffers(R, S)
B = R * S
g = fld.grid
gt = torch.zeros(fld.n_entries, 2, device='cuda')
b2 = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in b.items()}
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

def scatter(st, Bn):
  with torch.cuda.stream(st):
    lib.call('nof_hash_encode_bwd', C.byref(g), b['pts_w'], fld.table, b['dfeat'], gt, b['dpts'], Bn)

def rest(st, Bn):
  with torch.cuda.stream(st):
    lib.call('nof_hash_encode_fwd', C.byref(g), b2['pts_w'], fld.table, b2['feat'], Bn)
    lib.call('nof_mlp_fwd', C.byref(fld.desc), fld.packed, b2['feat'], fld.L, b2['view'], S, b2['raw'], b2['sig'], Bn)
    lib.call('nof_mlp_bwd', C.byref(fld.desc), fld.packed, b2['feat'], fld.L, b2['view'], S, b2['draw'], b2['sig'], b2['dsig'],
             b2['dfeat'], b2['dview'], b2['partials'], Bn)

def timeit(fn, n=7):
    ts = []
    for _ in range(n):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return float(np.median(ts)) * 1e3

def both(Bn):
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur); s2.wait_stream(cur)
    scatter(s1, Bn); rest(s2, Bn)
    cur.wait_stream(s1); cur.wait_stream(s2)

cur = torch.cuda.current_stream()
for Bn in (B, B // 2, B // 4):
    a = timeit(lambda: scatter(cur, Bn)); r = timeit(lambda: rest(cur, Bn)); c = timeit(lambda: both(Bn))
    print(f'B={Bn}: scatter alone {a:.0f} us | hash fwd + MLP fwd + bwd alone {r:.0f} us | both on two streams {c:.0f} us (sum {a + r:.0f})')
