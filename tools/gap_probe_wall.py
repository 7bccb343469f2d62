"""Without a profiler: what a dependent launch costs on this queue.  N pairs of (kernel A: streams 32 / 128 MB, kernel B: one word)
back to back, wall clock per pair against A's own duration measured with ONE pair of events around a long run of A alone; then the
same with a timing event recorded after every pair.  (tools/gap_probe.py reads the same kernels' gaps from a rocprofv3 trace.)"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from bundlesdf_amd import build
so = C.CDLL(build.build_probe(verbose=False))
so.nof_gap_probe.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]
word = torch.zeros(16, device='cuda')
src = torch.randn(128 * 2 ** 20 // 4, device='cuda')
dst = torch.empty_like(src)
st = torch.cuda.current_stream().cuda_stream
N = 400
for mb in (8, 32, 128):
    for with_events in (0, 1):
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(N)]
        for _ in range(20):
            so.nof_gap_probe(dst.data_ptr(), src.data_ptr(), mb * 2 ** 20 // 16, 0, 2048, word.data_ptr(), st)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(N):
            so.nof_gap_probe(dst.data_ptr(), src.data_ptr(), mb * 2 ** 20 // 16, 0, 2048, word.data_ptr(), st)
            if with_events:
                evs[i].record()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f'{mb:4d} MB per pair, {"an event after every pair" if with_events else "no events":>26}: {(t2 - t0) / N * 1e6:7.2f} us per pair '
              f'(host enqueue {(t1 - t0) / N * 1e6:5.2f} us per pair)')
