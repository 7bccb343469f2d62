"""A known byte count for calibrating FETCH_SIZE / WRITE_SIZE (rocprofv3 --pmc): 146 MB (= 36.5 M floats, the size of one flat
cfg2 buffer) copied device to device a few times, with 16-byte accesses (torch's vectorised copy) -- and the library's own Adam pass
over buffers of the same size (16 B read + 16 B written per parameter)."""
import sys, ctypes as C, torch
sys.path.insert(0, '.')
from bundlesdf_amd import lib
n = 36_500_000
src, dst = torch.randn(n, device='cuda'), torch.empty(n, device='cuda')
for _ in range(5):
    dst.copy_(src)
bufs = [torch.randn(n, device='cuda').abs_() for _ in range(4)]
for _ in range(5):
    lib.call('nof_adam_step', *bufs, n, n, C.c_float(0.01), C.c_float(0.003), C.c_float(0.9), C.c_float(0.999), C.c_float(1e-15), 7, None)
torch.cuda.synchronize()
print('copied', n * 4, 'bytes x 5; adam over', n, 'parameters x 5')
