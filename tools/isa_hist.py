"""Instruction histogram per kernel of a HIP source compiled for gfx950: python tools/isa_hist.py file.hip [extra flags]"""
import re, collections, subprocess, sys
src = sys.argv[1]
cmd = ['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-ffp-contract=off', '-munsafe-fp-atomics',
       '-I', 'include', '-x', 'hip', '-S', '--cuda-device-only', src, '-o', '/tmp/isa_hist.s'] + sys.argv[2:]
subprocess.run(cmd, capture_output=True)
txt = open('/tmp/isa_hist.s').read()
for m in re.finditer(r'^(_Z\w+):[^\n]*\n(.*?)s_endpgm', txt, re.S | re.M):
    name, body = m.group(1), m.group(2)
    ops = collections.Counter()
    for line in body.splitlines():
        line = line.strip()
        if not line or line.startswith(('.', ';', '//')) or line.endswith(':'):
            continue
        ops[line.split()[0]] += 1
    name = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
    print(re.sub(r'\(.*', '', name)[:50], sum(ops.values()), dict(ops.most_common(16)))
