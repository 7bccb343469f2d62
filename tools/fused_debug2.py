"""which FEATURES are wrong in a failing lane quarter, and what do they hold instead?"""
import sys, ctypes as C, numpy as np, torch
sys.path.insert(0, '.')
from bundlesdf_amd import lib
from tests.test_gpu_step import _pair
from tests import util as U
ns, nc, R = 3, 2, 2048
cfg, fld, orc, batch, rng = _pair(lib, 'fp16x3', 0, ns, nc, R=R)
Ns, Na = cfg['N_samples'], cfg['N_samples_around_depth']
S = Ns + Na
B = R * S
u1, u2 = rng.random((R, Ns)).astype(np.float32), rng.random((R, Na)).astype(np.float32)
fld.fused_forward = False
b = fld.train_step(U.dev(batch), None, R, U.dev(u1), U.dev(u2), do_step=False)
torch.cuda.synchronize()
want = b['feat'].permute(1, 0, 2).reshape(B, 32).clone()
featq = torch.zeros(B * 32 + B * 64, dtype=torch.int16, device='cuda')
raw = torch.zeros(B, 4, device='cuda'); sig = torch.zeros(B, 16, dtype=torch.int16, device='cuda')
shown = 0
for rep in range(30):
    featq.zero_()
    lib.call('nof_encode_mlp_fwd', C.byref(fld.grid), C.byref(fld.desc), fld.packed, fld.table, b['pts_w'], b['view'], S, raw, sig, featq, B)
    torch.cuda.synchronize()
    dbg = featq[B * 32:].view(torch.float32).reshape(B, 32)
    bad = (dbg != want)
    rows = torch.nonzero(bad.any(-1)).reshape(-1)
    if rows.numel() == 0:
        continue
    print(f'rep {rep}: {rows.numel()} bad samples; bad feature columns histogram {bad.sum(0).tolist()}')
    for r in rows[:3].tolist():
        cols = torch.nonzero(bad[r]).reshape(-1).tolist()
        print(f'  sample {r} (pair {r // 64}, lane {r % 64}): wrong features {cols}')
        for c in cols[:6]:
            g, w = dbg[r, c].item(), want[r, c].item()
            # is the wrong value some other sample's value of the same feature (which sample?), or another feature of this sample?
            same_col = torch.nonzero(want[:, c] == g).reshape(-1).tolist()[:4]
            same_row = torch.nonzero(want[r] == g).reshape(-1).tolist()
            print(f'    feature {c}: got {g:.6e} want {w:.6e}; equals feature {c} of samples {same_col} / features {same_row} of this sample')
    shown += 1
    if shown >= 3:
        break
