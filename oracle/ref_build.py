"""Recipe for oracle/_ref/libnof_ref.so -- TEST INFRASTRUCTURE.

Compiles the reference's OWN native arithmetic for the hot path as host C++ with g++, from the sources where they lie
under /root/reference (never copied into this repository):

    mycuda/torch_ngp_grid_encoder/gridencoder.cu   fast_hash, get_grid_index, kernel_grid, kernel_grid_backward,
                                                   kernel_input_backward, launchers, grid_encode_forward/backward
    mycuda/common.cu                               sample_rays_uniform_occupied_voxels_kernel,
                                                   postprocessOctreeRayTracingKernel, rayColorToTextureImageKernel

How: `-I oracle/ref_shim` puts host stand-ins in front of <cuda.h>, <cuda_fp16.h>, <cuda_runtime.h>, <ATen/...>,
<torch/...> and "Eigen/Dense" (oracle/ref_shim/cuda_host_shim.h: a launch is a nested loop over blockIdx/threadIdx on
one host thread, atomicAdd is `+=` in launch order, at::Half is a software binary16).  Two things in the files are not
C++ and are rewritten IN A TEMPORARY COPY that is deleted after the compile:
  * `kernel<<<grid, block>>>(args);`  ->  `{ dim3 g = grid; dim3 b = block; cuda_host::run(g, b, [&]{ kernel(args); }); }`
  * the sampler's error path `while (1){};` (common.cu:71,92: prints ERROR and hangs the GPU) -> `throw cuda_host::spin();`
Nothing else is touched; the reference's entry points are then exposed to ctypes by oracle/ref_shim/ref_capi_*.inc.

The output (a .so only) goes to oracle/_ref/, which is git-ignored and travels to the GPU box with the snapshot.
/root/reference does not exist there: tests use the prebuilt .so, or skip the reference-compiled checks when absent.

    python -m oracle.ref_build [--force]
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
SHIM = os.path.join(HERE, 'ref_shim')
OUT_DIR = os.path.join(HERE, '_ref')
LIB = os.path.join(OUT_DIR, 'libnof_ref.so')
REF_ROOT = os.environ.get('NOF_REFERENCE_ROOT', '/root/reference')
UNITS = [('mycuda/torch_ngp_grid_encoder/gridencoder.cu', 'ref_capi_grid.inc'),
         ('mycuda/common.cu', 'ref_capi_common.inc')]


def _split_top(s):
    """Split `grid, block` at the top-level comma (braces/parentheses nest)."""
    depth = 0
    for i, ch in enumerate(s):
        if ch in '({[':
            depth += 1
        elif ch in ')}]':
            depth -= 1
        elif ch == ',' and depth == 0:
            return s[:i].strip(), s[i + 1:].strip()
    raise ValueError(f'launch configuration without a block size: {s!r}')


def rewrite_launches(src):
    out, pos, n = [], 0, 0
    pat = re.compile(r'([A-Za-z_]\w*(?:\s*<[^<>;(){}]*>)?)\s*<<<(.*?)>>>\s*\(', re.S)
    while True:
        m = pat.search(src, pos)
        if not m:
            out.append(src[pos:])
            break
        depth, j = 1, m.end()
        while depth:
            depth += {'(': 1, ')': -1}.get(src[j], 0)
            j += 1
        assert src[j] == ';', src[m.start():j + 1]
        grid, block = _split_top(m.group(2))
        out.append(src[pos:m.start()])
        out.append(f'{{ dim3 nof_g__ = {grid}; dim3 nof_b__ = {block}; '
                   f'cuda_host::run(nof_g__, nof_b__, [&]{{ {m.group(1)}({src[m.end():j - 1]}); }}); }}')
        pos, n = j + 1, n + 1
    return ''.join(out), n


def available():
    return os.path.exists(LIB)


def can_build():
    return all(os.path.exists(os.path.join(REF_ROOT, u)) for u, _ in UNITS) and shutil.which('g++') is not None


def build(force=False, verbose=False):
    """Returns the path of the library, building it when the reference tree is present; None when it cannot be built."""
    deps = [os.path.join(SHIM, f) for f in ('cuda_host_shim.h', 'Eigen/Dense', 'ref_capi_grid.inc', 'ref_capi_common.inc')]
    deps.append(os.path.abspath(__file__))
    if not can_build():
        return LIB if available() else None
    deps += [os.path.join(REF_ROOT, u) for u, _ in UNITS]
    if not force and available() and all(os.path.getmtime(d) <= os.path.getmtime(LIB) for d in deps):
        return LIB
    os.makedirs(OUT_DIR, exist_ok=True)
    tmp = tempfile.mkdtemp(prefix='nof_ref_')
    try:
        objs = []
        for unit, capi in UNITS:
            path = os.path.join(REF_ROOT, unit)
            src, n = rewrite_launches(open(path).read())
            src, n_spin = re.subn(r'while\s*\(1\)\s*\{\s*\}\s*;', 'throw cuda_host::spin();', src)
            if verbose:
                print(f'{unit}: {n} launches rewritten, {n_spin} spin loops -> throw', flush=True)
            stem = os.path.splitext(os.path.basename(unit))[0]
            tu = os.path.join(tmp, stem + '_host.cpp')
            with open(tu, 'w') as f:
                f.write(f'#line 1 "{path}"\n{src}\n#include "{os.path.join(SHIM, capi)}"\n')
            obj = os.path.join(tmp, stem + '.o')
            cmd = ['g++', '-std=c++17', '-O2', '-fPIC', '-ffp-contract=off', '-fno-fast-math', '-w',
                   '-I', SHIM, '-I', os.path.dirname(path), '-c', tu, '-o', obj]
            if verbose:
                print(' '.join(cmd), flush=True)
            subprocess.check_call(cmd)
            objs.append(obj)
        subprocess.check_call(['g++', '-shared', '-fPIC', '-o', LIB] + objs)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose=True))
