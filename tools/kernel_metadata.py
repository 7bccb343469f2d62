"""Registers, scratch, LDS and workgroup size of every gfx950 kernel in the BUILT library, from the code objects bundled in it
(clang offload bundles, uncompressed; AMDGPU metadata note through llvm-readelf).  No GPU, no recompilation:
    python tools/kernel_metadata.py [path/to/libnof_hip.so]
tests/test_capi.py uses read() to keep the step's kernels free of scratch memory (tools/kernel_resources.py gives the same
numbers by recompiling each source with -save-temps)."""
import os
import re
import struct
import subprocess
import sys
import tempfile

READELF = '/opt/rocm/lib/llvm/bin/llvm-readelf'
MAGIC = b'__CLANG_OFFLOAD_BUNDLE__'


def read(lib_path):
    """[(demangled kernel name without 'void ', {metadata key: value string})]"""
    d = open(lib_path, 'rb').read()
    pos, rows = 0, []
    with tempfile.TemporaryDirectory() as tmp:
        while True:
            i = d.find(MAGIC, pos)
            if i < 0:
                break
            pos = i + len(MAGIC)
            n, p = struct.unpack_from('<Q', d, i + 24)[0], i + 32
            for _ in range(n):
                off, size, tl = struct.unpack_from('<QQQ', d, p)
                triple = d[p + 24:p + 24 + tl].decode()
                p += 24 + tl
                if 'gfx950' not in triple or not size:
                    continue
                co = os.path.join(tmp, 'k.co')
                open(co, 'wb').write(d[i + off:i + off + size])
                notes = subprocess.run([READELF, '--notes', co], capture_output=True, text=True, check=True).stdout
                for blk in notes.split('  - .agpr_count:')[1:]:
                    md = {k: v for k, v in re.findall(r'\.(\w+):\s+(\S+)\n', blk)}
                    md['agpr_count'] = blk.split('\n')[0].strip()
                    rows.append(md)
    assert rows, 'no gfx950 code object found in ' + lib_path
    names = subprocess.run(['c++filt'] + [r['name'] for r in rows], capture_output=True, text=True).stdout.splitlines()
    return [(n.replace('void ', ''), r) for n, r in zip(names, rows)]


if __name__ == '__main__':
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(here, 'bundlesdf_amd', 'libnof_hip.so')
    print(f"{'kernel':64s} {'vgpr':>5s} {'agpr':>5s} {'sgpr':>5s} {'scratch':>8s} {'vspill':>6s} {'lds':>7s} {'wg':>5s}")
    for name, md in sorted(read(path), key=lambda x: x[0]):
        print(f"{name.split('(')[0][:64]:64s} {md['vgpr_count']:>5s} {md['agpr_count']:>5s} {md['sgpr_count']:>5s} "
              f"{md['private_segment_fixed_size']:>8s} {md['vgpr_spill_count']:>6s} {md['group_segment_fixed_size']:>7s} "
              f"{md['max_flat_workgroup_size']:>5s}")
