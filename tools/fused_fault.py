"""nof_encode_mlp_fwd of the library NOF_LIB names (an A/B build of bundlesdf_amd/build.py:build_variant) against the two-launch forward
of the same library, bit for bit, on a step's own ray-ordered sample points: differing samples over `reps` launches, by lane quarter.
    NOF_LIB=bundlesdf_amd/ab_g2.so python tools/fused_fault.py 30"""
import sys, os, ctypes as C, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bundlesdf_amd import lib
from tests.test_gpu_step import _pair
from tests import util as U
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
tag = os.path.basename(os.environ.get('NOF_LIB', 'libnof_hip.so'))
for ns, nc, R in ((3, 2, 2048), (2, 3, 2048)):
    cfg, fld, orc, batch, rng = _pair(lib, 'fp16x3', 0, ns, nc, R=R)
    Ns, Na = cfg['N_samples'], cfg['N_samples_around_depth']
    S, B = Ns + Na, R * (Ns + Na)
    u1, u2 = rng.random((R, Ns)).astype(np.float32), rng.random((R, Na)).astype(np.float32)
    fld.fused_forward = False
    b = fld.train_step(U.dev(batch), None, R, U.dev(u1), U.dev(u2), do_step=False)
    torch.cuda.synchronize()
    raw_ref = b['raw'].clone()
    want_q = b['feat'].permute(1, 0, 2).reshape(B, 32).to(torch.float16)
    featq = torch.zeros(B * 32, dtype=torch.int16, device='cuda')
    raw, sig = torch.zeros(B, 4, device='cuda'), torch.zeros(B, 16, dtype=torch.int16, device='cuda')
    hist, total, launches_hit, cols = np.zeros(4, int), 0, 0, np.zeros(32, int)
    for rep in range(reps):
        raw.zero_(); featq.zero_()
        lib.call('nof_encode_mlp_fwd', C.byref(fld.grid), C.byref(fld.desc), fld.packed, fld.table, b['pts_w'], b['view'], S, raw, sig, featq, B)
        torch.cuda.synchronize()
        badq = featq.view(torch.float16).reshape(B, 32) != want_q
        bad = (raw != raw_ref).any(-1) | badq.any(-1)
        idx = torch.nonzero(bad).reshape(-1).cpu().numpy()
        total += idx.size
        launches_hit += idx.size > 0
        hist += np.bincount((idx % 64) // 16, minlength=4)
        cols += badq.sum(0).cpu().numpy()
    print(f'{tag:18s} ({ns},{nc}): {total:6d} differing samples, {launches_hit:3d} of {reps} launches x {B} samples hit; by lane quarter {hist.tolist()}; '
          f'feature columns {np.nonzero(cols)[0].tolist()}', flush=True)
