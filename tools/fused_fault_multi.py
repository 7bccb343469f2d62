"""nof_encode_mlp_fwd of several A/B builds (bundlesdf_amd/ab_*.so, linked with -Bsymbolic) in ONE process, each against the two-launch
forward of the regular library, bit for bit, on a step's own ray-ordered sample points.
    python tools/fused_fault_multi.py REPS lib1.so lib2.so ..."""
import sys, os, ctypes as C, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bundlesdf_amd import lib
from tests.test_gpu_step import _pair
from tests import util as U
reps, paths = int(sys.argv[1]), sys.argv[2:]
libs = []
for p in paths:
    so = C.CDLL(os.path.abspath(p))
    so.nof_encode_mlp_fwd.argtypes, so.nof_encode_mlp_fwd.restype = lib._SIGNATURES['nof_encode_mlp_fwd']
    libs.append((os.path.basename(p), so))
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
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for tag, so in libs:
        hist, total, hit, cols = np.zeros(4, int), 0, 0, np.zeros(32, int)
        for rep in range(reps):
            raw.zero_(); featq.zero_()
            rc = so.nof_encode_mlp_fwd(C.byref(fld.grid), C.byref(fld.desc), fld.packed.data_ptr(), fld.table.data_ptr(), b['pts_w'].data_ptr(),
                                       b['view'].data_ptr(), S, raw.data_ptr(), sig.data_ptr(), featq.data_ptr(), B, st)
            assert rc == 0, rc
            torch.cuda.synchronize()
            badq = featq.view(torch.float16).reshape(B, 32) != want_q
            bad = (raw != raw_ref).any(-1) | badq.any(-1)
            idx = torch.nonzero(bad).reshape(-1).cpu().numpy()
            total += idx.size; hit += idx.size > 0
            hist += np.bincount((idx % 64) // 16, minlength=4)
            cols += badq.sum(0).cpu().numpy()
        print(f'{tag:26s} ({ns},{nc}): {total:6d} differing samples, {hit:3d} of {reps} launches x {B} hit; by lane quarter {hist.tolist()}; feature columns {np.nonzero(cols)[0].tolist()}', flush=True)
