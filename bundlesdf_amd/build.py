"""Builds libnof_hip.so (the C-ABI library of include/nof_hip.h) for gfx950 with hipcc, in-tree.

    python -m bundlesdf_amd.build [--force]

hipcc cross-compiles without a GPU.  Object files are cached under bundlesdf_amd/csrc/build/ and are
rebuilt when their source (or a header) is newer.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
OBJ = os.path.join(CSRC, 'build')
LIB = os.path.join(HERE, 'libnof_hip.so')
SOURCES = ['nof_capi.hip', 'nof_hash.hip', 'nof_trace.hip', 'nof_loss.hip', 'nof_adam_tail.hip', 'nof_pose.hip', 'nof_mlp.hip', 'nof_mlp_wide.hip', 'nof_mesh.hip', 'nof_texture.hip']
HEADERS = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h')) + [os.path.join(HERE, '..', 'include', 'nof_hip.h')]
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-ffp-contract=off', '-munsafe-fp-atomics',
         '-Wno-unused-result', '-Wno-pass-failed', '-fno-slp-vectorize']
# -fno-slp-vectorize: on gfx950 a packed-fp32 VALU instruction whose SOURCE 1 is read through op_sel = 1 (its high register feeds the
# low result: `v_pk_mul_f32 vD, vA, vB op_sel:[0,1]`, likewise v_pk_add_f32 / v_pk_fma_f32) returns a wrong low result in lanes 48-63
# whenever another wave of the SIMD is executing an MFMA at that moment -- 1.6 % of the executions under a steady MFMA load, none
# without MFMAs (tools/repro/pk_swap_repro.*, profiles/r05_i_fault_pk_forms.txt; DESIGN 2.10: this is what corrupted the fused forward
# with two levels in flight in round 4).  clang emits the form only from its SLP vectoriser (shuffles folded into op_sel); without it
# the built library contains none (tools/pk_opsel_scan.py; tests/test_capi.py scans every library build() produces).
# NOTE: `-mllvm -amdgpu-mfma-vgpr-form` (keeps MFMA results in VGPRs: -23 % instructions in k_mlp_bwd) MISCOMPILES
# k_mlp_bwd<16-bit, 3, 2> with this ROCm 7.2 clang (weight gradients wrong, data gradients right; tests/test_gpu_ops.py
# caught it) -- do not enable it.
# nof_mlp.hip: without -fno-honor-nans every fmaxf(h, 0) of the ReLUs is TWO v_max_f32 (IEEE maxnum quiets signalling NaNs
# first); the file has no NaN-dependent logic.
EXTRA = {'nof_mlp.hip': ['-fno-honor-nans'], 'nof_mlp_wide.hip': ['-fno-honor-nans']}


def hipcc():
    for c in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', 'hipcc'):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError('hipcc not found')


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def _digest(paths, flags):
    """sha256 over the CONTENT of the sources, the headers and the flags: what a library was built from.  Kept beside the library
    as <lib>.stamp (git-ignored, travels with the gpurun snapshot): file times do not survive a snapshot, content does, so a
    library that lags its sources is rebuilt wherever build() runs -- here and on the GPU box (round 4 shipped a stale test library
    that way)."""
    import hashlib
    h = hashlib.sha256(' '.join(flags).encode())
    for p in sorted(os.path.abspath(x) for x in paths):
        h.update(os.path.basename(p).encode())
        h.update(open(p, 'rb').read())
    return h.hexdigest()


def _fresh(lib, digest):
    try:
        return os.path.exists(lib) and open(lib + '.stamp').read().strip() == digest
    except OSError:
        return False


def _stamp(lib, digest):
    with open(lib + '.stamp', 'w') as f:
        f.write(digest + '\n')


class _Lock:
    """one builder at a time (the ranks of a torchrun launch all call lib.load(), which builds a library that lags its sources)"""

    def __enter__(self):
        import fcntl
        self.f = open(os.path.join(HERE, '.build.lock'), 'w')
        fcntl.flock(self.f, fcntl.LOCK_EX)
        return self

    def __exit__(self, *a):
        import fcntl
        fcntl.flock(self.f, fcntl.LOCK_UN)
        self.f.close()


def build(force=False, verbose=True):
    srcs = [os.path.join(CSRC, x) for x in SOURCES if os.path.exists(os.path.join(CSRC, x))]
    digest = _digest(srcs + HEADERS, FLAGS + [f'{k}:{v}' for k, v in sorted(EXTRA.items())])
    if not force and _fresh(LIB, digest):
        return LIB                  # e.g. on the GPU box: the snapshot ships the .so and its stamp but not the object cache
    with _Lock():
        if not force and _fresh(LIB, digest):                         # (another process built it while this one waited)
            return LIB
        return _build_locked(force, verbose, digest)


def _build_locked(force, verbose, digest):
    os.makedirs(OBJ, exist_ok=True)
    cc = hipcc()
    objs, procs = [], []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        if not os.path.exists(s):
            continue
        o = os.path.join(OBJ, src.replace('.hip', '.o'))
        objs.append(o)
        od = _digest([s] + HEADERS, FLAGS + EXTRA.get(src, []))       # (an object is as old as its source, the headers AND the flags)
        if force or not _fresh(o, od):
            cmd = [cc] + FLAGS + EXTRA.get(src, []) + ['-x', 'hip', '-c', s, '-o', o]
            if verbose:
                print(' '.join(cmd), flush=True)
            procs.append((src, subprocess.Popen(cmd), o, od))
    failed = [src for src, p, _, _ in procs if p.wait() != 0]
    if failed:
        raise RuntimeError(f'hipcc failed for {failed}')
    for _, _, o, od in procs:
        _stamp(o, od)
    # (always link here: the library's stamp did not match, whatever the objects' file times say -- one second)
    cmd = [cc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.check_call(cmd)
    _stamp(LIB, digest)
    return LIB


PERTURB_LIB = os.path.join(HERE, 'libnof_hash_perturb.so')


def build_perturb(force=False, verbose=True):
    """Test build of the hash kernels with -DNOF_AGG_PERTURB (extra live registers across the scatter's hand-written DPP block:
    a different register allocation around it).  tests/test_gpu_ops.py runs the table scatter through it."""
    srcs = [os.path.join(CSRC, x) for x in ('nof_hash.hip', 'nof_capi.hip')]
    digest = _digest(srcs + HEADERS, FLAGS + ['perturb'])
    if not force and _fresh(PERTURB_LIB, digest):
        return PERTURB_LIB
    # -Bsymbolic: this library's references bind to its OWN kernels and helpers even when libnof_hip.so (same symbol names) is
    # already loaded in the process -- without it the dynamic linker would hand it the first library's kernel stubs
    cmd = [hipcc()] + FLAGS + ['-DNOF_AGG_PERTURB=1', '-shared', '-Wl,-Bsymbolic', '-x', 'hip'] + srcs + ['-o', PERTURB_LIB]
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.check_call(cmd)
    _stamp(PERTURB_LIB, digest)
    return PERTURB_LIB


PROBE_LIB = os.path.join(HERE, 'libnof_probe.so')


def build_probe(force=False, verbose=True):
    """TEST-ONLY library: the raw MFMA tile probe (tests/test_gpu_ops.py::test_mfma_operand_layout) and the atomic-rate probe
    (tools/atomic_probe.py) -- csrc/test/nof_probe.hip.  Not part of libnof_hip.so, not declared in include/nof_hip.h."""
    srcs = [os.path.join(CSRC, 'test', 'nof_probe.hip')]
    digest = _digest(srcs + HEADERS, FLAGS + ['probe'])
    if not force and _fresh(PROBE_LIB, digest):
        return PROBE_LIB
    cmd = [hipcc()] + FLAGS + ['-shared', '-Wl,-Bsymbolic', '-x', 'hip'] + srcs + ['-o', PROBE_LIB]
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.check_call(cmd)
    _stamp(PROBE_LIB, digest)
    return PROBE_LIB


def load_probe():
    """ctypes handle of the probe library with its two signatures set (builds it when it lags its source)."""
    import ctypes as C
    so = C.CDLL(build_probe(verbose=False))
    so.nof_mfma_probe.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]
    so.nof_atomic_probe.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
    so.nof_mfma_probe.restype = so.nof_atomic_probe.restype = C.c_int
    so.nof_probe_last_error.restype = C.c_char_p
    return so


def build_variant(name, defines, sources=('nof_mlp.hip',), verbose=True):
    """Development aid for A/B runs on the GPU box: the library with `sources` recompiled under extra -D flags, linked with the
    regular objects of everything else, as bundlesdf_amd/ab_<name>.so (git-ignored; travels with the gpurun snapshot).  Loaded
    instead of libnof_hip.so when the environment names it (NOF_LIB, read by lib.py -- the Python host, not the C library)."""
    build(verbose=verbose)
    cc = hipcc()
    objs = []
    for src in SOURCES:
        o = os.path.join(OBJ, src.replace('.hip', '.o'))
        if src in sources:
            o = os.path.join(OBJ, f'ab_{name}_' + src.replace('.hip', '.o'))
            cmd = ([cc] + FLAGS + EXTRA.get(src, []) + [d if d.startswith('-') else f'-D{d}' for d in defines] +       # (a leading '-': a raw compiler flag)
                   ['-x', 'hip', '-c', os.path.join(CSRC, src), '-o', o])
            if verbose:
                print(' '.join(cmd), flush=True)
            subprocess.check_call(cmd)
        objs.append(o)
    out = os.path.join(HERE, f'ab_{name}.so')
    subprocess.check_call([cc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', out] + objs)
    return out


if __name__ == '__main__':
    if '--variant' in sys.argv:                      # python -m bundlesdf_amd.build --variant NAME [--sources a.hip,b.hip] DEFINE[=V] ...
        i = sys.argv.index('--variant')
        rest = sys.argv[i + 2:]
        srcs = ('nof_mlp.hip',)
        if rest and rest[0] == '--sources':
            srcs, rest = tuple(rest[1].split(',')), rest[2:]
        print(build_variant(sys.argv[i + 1], rest, sources=srcs))
    else:
        print(build(force='--force' in sys.argv))
