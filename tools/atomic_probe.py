"""fp32 atomic throughput on MI355X for the address patterns the hash scatter could use (run on the GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bundlesdf_amd import build as _build
_probe = _build.load_probe()          # the test-only probe library (csrc/test/nof_probe.hip), not libnof_hip.so
def _atomic_probe(variant, idx, table, n):
    assert _probe.nof_atomic_probe(variant, idx.data_ptr(), table.data_ptr(), n, torch.cuda.current_stream().cuda_stream) == 0, _probe.nof_probe_last_error()
n = 1 << 24
T = 1 << 19
g = torch.Generator(device='cuda').manual_seed(0)
idx_rand = torch.randint(0, T, (n,), device='cuda', generator=g, dtype=torch.int64).to(torch.int32)
idx_sorted = idx_rand.view(-1, 64).sort(dim=1)[0].reshape(-1).contiguous()
# "ray-like": runs of nearby entries
idx_hot = (torch.randint(0, 512, (n,), device='cuda', generator=g, dtype=torch.int64)).to(torch.int32)
table = torch.zeros(T, 2, device='cuda')
def run(name, variant, idx):
    for rep in range(3):
        table.zero_()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); _atomic_probe(variant, idx, table, n); b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b)
    per = 1 if variant == 3 else 2
    print(f'{name:46s} {ms:8.3f} ms  {n * per / ms / 1e6:8.1f} G atomics/s  sum={table.sum().item():.0f}')
run('0 x,y separate instr, random entries', 0, idx_rand)
run('1 adjacent-lane (x,y) pairs, random entries', 1, idx_rand)
run('2 x,y separate instr, sequential entries', 2, idx_rand)
run('3 x only, random entries', 3, idx_rand)
run('4 x,y separate, per-wave sorted entries', 0, idx_sorted)
run('5 x,y separate, 512 hot entries', 0, idx_hot)
run('6 same address from lanes 16 apart (16 lines/instr)', 3, idx_rand) if False else None
def run1(name, variant, idx):
    for rep in range(3):
        table.zero_()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); _atomic_probe(variant, idx, table, n); b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b)
    print(f'{name:58s} {ms:8.3f} ms  {n / ms / 1e6:8.1f} G lane-ops/s  {n / 4 / ms / 1e6:8.1f} G lines/s  sum={table.sum().item():.0f}')
run1('6 same ADDRESS, lanes 16 apart, 16 lines/instr', 6, idx_rand)
run1('7 same line diff words, lanes 16 apart, 16 lines/instr', 7, idx_rand)
run1('8 same line diff words, adjacent lanes, 16 lines/instr', 8, idx_rand)
run1('9 plain 4-byte stores, random entries (64 lines/instr)', 9, idx_rand)

# round 2: does the place where the atomic executes depend on instruction flags / on keeping a line inside one XCD?
def run2(name, variant):
    for rep in range(3):
        table.zero_()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); _atomic_probe(variant, idx_rand, table, n); b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b)
    print(f'{name:62s} {ms:8.3f} ms  {n / ms / 1e6:8.1f} G line-requests/s  sum={table.sum().item():.0f}')
for part, pn in enumerate(['random entries over the whole table', 'entries folded into the XCD-owned eighth (XCC_ID)', 'entries folded by blockIdx % 8']):
    for flag, fn in enumerate(['plain', 'nt', 'sc1', 'sc0 (returning)']):
        run2(f'{10 + 4 * part + flag} x only, {pn}, {fn}', 10 + 4 * part + flag)
for v, nm in ((22, 'u32 add'), (23, 'u64 add'), (24, 'f64 add'), (25, 'packed f16 add')):
    run2(f'{v} random entries, {nm}', v)

# gather rate of 8-byte rows: does a run of lanes reading the SAME row cost as much as distinct rows?
tg = torch.zeros((1 << 20) + n, device='cuda')          # 2^19 rows x 2 floats + one result per lane
def run3(name, variant):
    for rep in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); _atomic_probe(variant, idx_rand, tg, n); b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b)
    print(f'{name:62s} {ms:8.3f} ms  {8 * n / ms / 1e6:8.1f} G lane-gathers/s')
run3('30 8 gathers per lane, every lane its own random row', 30)
run3('31 8 gathers per lane, 8 adjacent lanes read the same row', 31)
run3('32 8 gathers, one lane of every 8 active', 32)
