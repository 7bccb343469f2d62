"""The hardware behaviour behind round 4's "quarter-wave fault" of the fused forward, as a stand-alone kernel:
tools/repro/pk_swap_repro.hip.  On gfx950 (MI355X, ROCm 7.2) a packed-fp32 VALU instruction that reads SOURCE 1 through op_sel = 1
(the source's high register feeds the low result),

    v_pk_mul_f32 vD, vA, vB op_sel:[0,1]         (likewise v_pk_add_f32, v_pk_fma_f32; in place or not; with or without op_sel_hi)

delivers a wrong LOW result in lanes 48-63 when another wave of the SIMD executes an MFMA at that moment: 1.6 % of the executions under
a steady MFMA load, none when no MFMA is issued; op_sel on source 0 and plain packed instructions are never wrong
(profiles/r05_i_fault_pk_forms.txt).  clang emits the form from its SLP vectoriser only, so the library is built with
-fno-slp-vectorize and tests/test_capi.py scans the built code for it.

Collected FIRST (tests/conftest.py).  What is asserted is what the product relies on: without MFMAs every form is exact; under MFMA
load the forms the library may contain (plain, op_sel on source 0) are exact.  What the source-1 forms do on this box is printed
(every box of round 5: wrong only in the last quarter-wave, only in the low result -- the signature the round-4 bisect found); a
different signature is reported as a warning, not as a failure: the product contains none of these forms."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, 'tools', 'repro', 'pk_swap_repro.hip')
FORMS = {0: 'v_pk_mul_f32 D, D, B op_sel:[0,1]', 1: 'v_pk_mul_f32 D, D, B op_sel:[0,1] op_sel_hi:[1,0]', 2: 'v_pk_add_f32 D, D, B op_sel:[0,1]',
         3: 'v_pk_fma_f32 D, D, B, C op_sel:[0,1,0]', 4: 'v_pk_mul_f32 D, D, B op_sel:[1,0]', 5: 'v_pk_mul_f32 D, D, B', 6: 'v_pk_mul_f32 D, A, B op_sel:[0,1]',
         7: 'v_fma_mixlo_f16 D, H, S, X op_sel_hi:[1,0,0]', 8: 'v_fma_mixhi_f16 D, H, S, X op_sel_hi:[1,0,0]'}
SAFE = (4, 5, 7, 8)        # (7, 8: round 6 -- the operand split's residual is one v_fma_mixlo / mixhi_f16 per element)


def test_packed_fp32_source1_op_sel_under_mfma_load(nof, tmp_path):
    so_path = str(tmp_path / 'libpk_swap.so')
    subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off', '-fno-slp-vectorize', '-shared',
                           '-fPIC', SRC, '-o', so_path])
    so = C.CDLL(so_path)
    so.pk_run.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    blocks, iters = 512, 100
    n = blocks * 12 * 64
    sink = torch.zeros(768, device='cuda')

    def run(form, mfma_iters):
        counts = torch.zeros(n, 4, dtype=torch.int32, device='cuda')
        by_index = torch.zeros(n, dtype=torch.int32, device='cuda')
        assert so.pk_run(form, blocks, counts.data_ptr(), sink.data_ptr(), iters, mfma_iters, 0, by_index.data_ptr(), None) == 0
        torch.cuda.synchronize()
        return counts.cpu().numpy().reshape(blocks * 12, 64, 4).astype(np.int64)

    for form, text in FORMS.items():
        quiet = run(form, 0)
        assert quiet[..., 0].sum() == 0, f'{text}: wrong results without any MFMA in the kernel'
        loaded = run(form, 24)
        wrong = int(loaded[..., 0].sum())
        by_quarter = loaded[..., 0].reshape(-1, 4, 16).sum((0, 2)).tolist()
        print(f'{text:52s} under MFMA load: {wrong:9d} wrong of {n * iters * 16:.1e} executions; by lane quarter {by_quarter}; '
              f'low {int(loaded[..., 1].sum())}, high {int(loaded[..., 2].sum())}')
        if form in SAFE:
            assert wrong == 0, f'{text}: a form the library may contain is wrong under MFMA load'
        elif not (by_quarter[0] == by_quarter[1] == by_quarter[2] == 0 and int(loaded[..., 2].sum()) == 0):
            # (observed on every box of this round: only lanes 48-63, only the low result.  Another signature is news about the
            # hardware, not a defect of the product -- which contains none of these forms --: reported, the suite goes on)
            import warnings
            warnings.warn(f'{text}: wrong results outside the last quarter-wave / in the high result: {by_quarter}')
