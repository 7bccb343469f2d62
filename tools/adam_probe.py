import sys, ctypes as C, torch
sys.path.insert(0, '.')
from bundlesdf_amd import lib
for n in (12_700_000, 59_000_000):
    bufs = [torch.randn(n + 8, device='cuda').abs_() for _ in range(4)]
    args = (C.c_float(0.01), C.c_float(0.003), C.c_float(0.9), C.c_float(0.999), C.c_float(1e-15), 7, None)
    for name, views in (('float4', [x[:n] for x in bufs]), ('scalar', [x[k:k + n] for k, x in enumerate(bufs)])):
        for _ in range(3): lib.call('nof_adam_step', *views, n, n, *args)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20): lib.call('nof_adam_step', *views, n, n, *args)
        b.record(); torch.cuda.synchronize()
        ms = a.elapsed_time(b) / 20
        print(f'n={n} {name}: {ms*1e3:.1f} us  {n*32/ms/1e9:.2f} TB/s')
