"""Fixed cost of the merged MLP backward launch (k_mlp_bwd_both): the same call over the batch's real work list, over an EMPTY list
(weight images in, accumulators flushed, nothing in between) and over every tile -- per-launch time from events around 50 calls."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np, torch
from bundlesdf_amd import lib
from tests.test_gpu_step import _pair
from tests import util as U
lib.load()
cfg, fld, orc, batch, rng = _pair(lib, 'fp16x3', ns=3, nc=2, R=256)
R0 = batch.shape[0]
R = 4096
pool = U.dev(np.ascontiguousarray(np.tile(batch, (R // R0 + 1, 1))[:R]))
for _ in range(30):
    fld.train_step(pool, None, R, seed=1)
torch.cuda.synchronize()
S = cfg['N_samples'] + cfg['N_samples_around_depth']
B = R * S
b = fld._buffers(R, S)
tiles = b['tiles']
def run(tag, lst, n=50):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    for it in range(n + 5):
        if it == 5:
            ev[0].record()
        lib.call('nof_mlp_bwd_featq', C.byref(fld.desc), fld.packed, b['featq'], fld.L, b['view'], S, b['draw'], b['sig'], b['dsig'],
                 b['dfeat'], b['dview'], b['partials'], lst, B)
    ev[1].record()
    torch.cuda.synchronize()
    head = lst[:16].view(torch.int32)[:2].tolist() if lst is not None else None
    print(f'{tag:>28}: {ev[0].elapsed_time(ev[1]) / n * 1e3:7.1f} us per launch  (listed tiles, tiles: {head})')
run('the step\'s own list', tiles)
empty = torch.zeros_like(tiles)
lib.call('nof_tile_list_build', torch.zeros(B, 4, device='cuda'), B, 0, empty)
run('empty list', empty)
full = torch.zeros_like(tiles)
lib.call('nof_tile_list_build', None, B, 1, full)
run('every tile', full)
