"""Driver of pk_swap_repro.hip (run on the GPU box).  Per instruction form: executions whose result differs from the arithmetic the
instruction is defined as, with and without MFMA chains in the other waves of every SIMD, by lane quarter."""
import ctypes as C, os, subprocess, sys
import numpy as np, torch
HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, '..', '..', 'bundlesdf_amd', 'libpk_swap.so')
subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off', '-fno-slp-vectorize', '-shared', '-fPIC',
                       os.path.join(HERE, 'pk_swap_repro.hip'), '-o', SO])
so = C.CDLL(SO)
so.pk_run.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
blocks, iters = 512, 200
n = blocks * 12 * 64
sink = torch.zeros(768, device='cuda')
FORMS = {0: 'v_pk_mul_f32 D, D, B op_sel:[0,1]', 1: 'v_pk_mul_f32 D, D, B op_sel:[0,1] op_sel_hi:[1,0]', 2: 'v_pk_add_f32 D, D, B op_sel:[0,1]',
         3: 'v_pk_fma_f32 D, D, B, C op_sel:[0,1,0]', 4: 'v_pk_mul_f32 D, D, B op_sel:[1,0]', 5: 'v_pk_mul_f32 D, D, B', 6: 'v_pk_mul_f32 D, A, B op_sel:[0,1]'}
f32 = np.float32
lane = np.arange(64)
a_x = (f32(1.5) + f32(0.001) * lane.astype(f32)).astype(f32)
def b_hi(w):                                              # the kernel's b.y of wave w, per lane
    return ((f32(0.75) - f32(0.01) * (lane & 3).astype(f32)).astype(f32) + f32(0.03125) * f32(w + 1)).astype(f32)
for form, text in FORMS.items():
    for mfma_iters, pad in ((0, 0), (24, 0), (1, 0), (24, 1), (24, 4), (24, 16)) if form == 0 else ((0, 0), (24, 0)):
        tot = np.zeros(3, np.int64); hist = np.zeros(4, np.int64); hit = 0; other = 0; firsts = 0; which = 0
        for rep in range(reps):
            counts = torch.zeros(n, 4, dtype=torch.int32, device='cuda')
            by_index = torch.zeros(n, dtype=torch.int32, device='cuda')
            assert so.pk_run(form, blocks, counts.data_ptr(), sink.data_ptr(), iters, mfma_iters, pad, by_index.data_ptr(), None) == 0
            torch.cuda.synchronize()
            which |= int(np.bitwise_or.reduce(by_index.cpu().numpy().view(np.uint32)))
            c = counts.cpu().numpy().reshape(blocks, 12, 64, 4)
            tot += c[..., :3].astype(np.int64).sum((0, 1, 2)); hit += bool(c[..., 0].any())
            hist += c[..., 0].astype(np.int64).reshape(-1, 4, 16).sum((0, 2))
            if form in (0, 1, 6):                         # is a wrong low product a.x * (ANOTHER wave's b.y)?
                blk, w, l = np.nonzero(c[..., 1])
                got = c[blk, w, l, 3].view(f32)
                firsts += got.size
                cand = np.stack([(a_x[l] * b_hi(x)[l]).astype(f32) for x in range(12)], 1)          # [events, 12]
                other += int(((cand == got[:, None]) & (np.arange(12)[None] != w[:, None])).any(1).sum())
        print(f'{text:50s} MFMAs per chain {mfma_iters:2d}, s_nop 15 x {pad:2d} behind it: wrong {tot[0]:8d} of {reps * n * iters * 16:.1e} (low {tot[1]}, high {tot[2]}); launches hit {hit}/{reps}; '
              f'by lane quarter {hist.tolist()}; wrong executions after a chain (bit i = the i-th) {which:#06x}' + (f'; of {firsts} first wrong low products {other} equal a.x * b.hi of ANOTHER wave of the workgroup' if firsts else ''), flush=True)
