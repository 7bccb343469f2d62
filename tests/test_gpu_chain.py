"""Both links of the parity chain in ONE record.

`pytest -m gpu` (what the driver runs on the MI355X box) covers HIP <-> oracle and HIP <-> reference fixture; the
oracle <-> reference link lives in the CPU tests (`-m "not gpu"`: tests/test_oracle.py against reference-executed golden
vectors, tests/test_ref_native.py against the reference's own gridencoder.cu / common.cu compiled as host C++).  This wrapper runs
those CPU tests on the GPU box too, so that the driver's GPU record shows the oracle pinned on the same machine, with the same
NumPy / torch / libm, that the HIP comparisons ran on -- and fails if the compiled reference library did not travel."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_is_pinned_on_this_box():
    from oracle import ref_native as RN
    assert RN.load() is not None, 'oracle/_ref/libnof_ref.so is missing: the compiled reference kernels did not travel'
    r = subprocess.run([sys.executable, '-m', 'pytest', '-q', '-x', '-p', 'no:cacheprovider', 'tests/test_oracle.py',
                        'tests/test_ref_native.py', 'tests/test_host_logic.py', 'tests/test_scene_io.py', 'tests/test_mesh.py'],
                       cwd=ROOT, capture_output=True, text=True, timeout=600)
    tail = '\n'.join(r.stdout.splitlines()[-15:])
    print(tail)
    assert r.returncode == 0, tail + '\n' + r.stderr[-2000:]
    assert ' skipped' not in r.stdout.splitlines()[-1], tail           # nothing may hide behind a skip here
