"""fused vs two-launch forward on a step's own sample points: how many samples differ, where (lane, wave), is it repeatable?"""
import sys, ctypes as C, numpy as np, torch
sys.path.insert(0, '.')
from bundlesdf_amd import lib
from tests.test_gpu_step import _pair
from tests import util as U
for ns, nc, R in ((3, 2, 256), (3, 2, 2048), (2, 3, 2048)):
    cfg, fld, orc, batch, rng = _pair(lib, 'fp16x3', 0, ns, nc, R=R)
    Ns, Na = cfg['N_samples'], cfg['N_samples_around_depth']
    S = Ns + Na
    B = R * S
    u1, u2 = rng.random((R, Ns)).astype(np.float32), rng.random((R, Na)).astype(np.float32)
    fld.fused_forward = False
    b = fld.train_step(U.dev(batch), None, R, U.dev(u1), U.dev(u2), do_step=False)
    torch.cuda.synchronize()
    raw_ref, sig_ref, feat_ref = b['raw'].clone(), b['sig'].clone(), b['feat'].clone()
    want_q = feat_ref.permute(1, 0, 2).reshape(B, 32).to(torch.float16)
    featq = torch.zeros(B * 32 + B * 64, dtype=torch.int16, device='cuda')      # [B,32] operand copy + (debug build) [B,32] fp32 features
    raw = torch.zeros(B, 4, device='cuda')
    sig = torch.zeros(B, 16, dtype=torch.int16, device='cuda')
    hist = np.zeros(64, int)
    for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 12):
        raw.zero_(); featq.zero_()
        lib.call('nof_encode_mlp_fwd', C.byref(fld.grid), C.byref(fld.desc), fld.packed, fld.table, b['pts_w'], b['view'], S, raw, sig, featq, B)
        torch.cuda.synchronize()
        bad = (raw != raw_ref).any(-1)
        badq = (featq[:B * 32].view(torch.float16).reshape(B, 32) != want_q).any(-1)
        dbg = featq[B * 32:].view(torch.float32).reshape(B, 32)
        badd = (dbg != feat_ref.permute(1, 0, 2).reshape(B, 32)).any(-1)
        idx = torch.nonzero(bad).reshape(-1).cpu().numpy()
        hist += np.bincount(idx % 64, minlength=64)
        print(f'({ns},{nc}) R={R} rep {rep}: raw differs on {int(bad.sum())} samples, featq on {int(badq.sum())}, pre-LDS features on {int(badd.sum())} (same samples: {bool((badd == badq).all())}); max |d raw| {(raw - raw_ref).abs().max().item():.2e}; '
              f'pairs {sorted(set((idx // 64).tolist()))[:6]} lanes {sorted(set((idx % 64).tolist()))[:4]}..')
    print('   by lane quarter', hist.reshape(4, 16).sum(1).tolist())
