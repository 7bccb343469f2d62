"""Scans the gfx950 code objects bundled in a BUILT library for packed-fp32 instructions whose SOURCE 1 delivers its HIGH register to the
LOW result (op_sel[1] = 1):
    v_pk_mul_f32 vD, vA, vB op_sel:[0,1]
The form returned, now and then and in lanes 48-63 only, a wrong low/high product while other waves of the SIMD issued MFMAs (gfx950,
ROCm 7.2: DESIGN 2.10, tools/repro/pk_swap_repro.*, profiles/r05_*_fault_*.txt).  clang's SLP vectoriser emits it; nothing hand-written
does.  No GPU needed:      python tools/pk_opsel_scan.py [path/to/lib.so]      -> offending instructions per kernel (exit 1 if any)
tests/test_capi.py runs scan() on every library build() produces."""
import os
import re
import struct
import subprocess
import sys
import tempfile

OBJDUMP = '/opt/rocm/lib/llvm/bin/llvm-objdump'
MAGIC = b'__CLANG_OFFLOAD_BUNDLE__'
PK = re.compile(r'\b(v_pk_(?:mul|add|fma)_f32)\s+(.*)$')        # the packed-fp32 arithmetic forms of gfx950 (the ones the reproducer measured)


def code_objects(lib_path):
    """the gfx950 code objects of the library's (uncompressed) clang offload bundles; raises if the library holds no such bundle --
    a compressed bundle (magic CCOB) or a changed layout must not read as 'nothing found'"""
    d = open(lib_path, 'rb').read()
    if MAGIC not in d:
        raise RuntimeError(f'{lib_path}: no uncompressed clang offload bundle' + (' (compressed bundle: CCOB)' if b'CCOB' in d else ''))
    pos = 0
    while True:
        i = d.find(MAGIC, pos)
        if i < 0:
            return
        pos = i + len(MAGIC)
        n, p = struct.unpack_from('<Q', d, i + 24)[0], i + 32
        for _ in range(n):
            off, size, tl = struct.unpack_from('<QQQ', d, p)
            triple = d[p + 24:p + 24 + tl].decode()
            p += 24 + tl
            if 'gfx950' in triple and size:
                yield d[i + off:i + off + size]


def hazardous(operands):
    """source 1 (source 2 of an fma as well) read through op_sel = 1: its high register feeds the low result"""
    m = re.search(r'op_sel:\[([01,]+)\]', operands)
    if not m:
        return False
    sel = [int(x) for x in m.group(1).split(',')]
    return any(sel[1:])


def scan(lib_path, stats=None):
    """[(kernel symbol, instruction text)] of every hazardous instruction in the library's device code.  `stats` (a dict) receives
    what was actually looked at -- code_objects, kernels, instructions, mfma, packed_f32 -- so that a caller can tell an empty result
    from a scan that saw nothing."""
    hits = []
    st = dict(code_objects=0, kernels=0, instructions=0, mfma=0, packed_f32=0)
    with tempfile.TemporaryDirectory() as tmp:
        for k, co in enumerate(code_objects(lib_path)):
            path = os.path.join(tmp, f'{k}.co')
            open(path, 'wb').write(co)
            txt = subprocess.run([OBJDUMP, '-d', '--no-show-raw-insn', path], capture_output=True, text=True, check=True).stdout
            sym = None
            st['code_objects'] += 1
            for ln in txt.splitlines():
                m = re.match(r'^[0-9a-f]+ <(\S+)>:', ln)
                if m:
                    sym = m.group(1)
                    st['kernels'] += 1
                    continue
                if sym is None or not ln.startswith('\t'):
                    continue
                st['instructions'] += 1
                st['mfma'] += 'v_mfma_' in ln
                m = PK.search(ln)
                if m:
                    st['packed_f32'] += 1
                    if hazardous(m.group(2)):
                        hits.append((sym, m.group(0).strip()))
    if stats is not None:
        stats.update(st)
    return hits


if __name__ == '__main__':
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'bundlesdf_amd', 'libnof_hip.so')
    st = {}
    hits = scan(lib, st)
    print(st)
    per = {}
    for sym, ins in hits:
        per.setdefault(sym, []).append(ins)
    for sym, ins in sorted(per.items()):
        print(f'{len(ins):4d}  {sym[:100]}')
        for i in sorted(set(ins))[:4]:
            print('        ', i)
    print(f'{len(hits)} packed-fp32 instructions with op_sel on source 1 in {len(per)} kernels of {lib}')
    sys.exit(1 if hits or not st['instructions'] else 0)
