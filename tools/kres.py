"""Prints one line per kernel from hipcc -Rpass-analysis=kernel-resource-usage: python tools/kres.py file.hip"""
import re, subprocess, sys
src = sys.argv[1]
cmd = ['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-ffp-contract=off',
       '-munsafe-fp-atomics', '-Wno-pass-failed', '-x', 'hip', '-c', src, '-o', '/tmp/kres.o',
       '-Rpass-analysis=kernel-resource-usage'] + sys.argv[2:]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = {}
for line in out.splitlines():
    m = re.search(r'remark: .*?:\d+:\d+:\s+(.*?)\s*\[-Rpass', line) or re.search(r'remark:\s+(.*?)\s*\[-Rpass', line)
    if not m: 
        if 'error' in line: print(line)
        continue
    t = m.group(1)
    if t.startswith('Function Name:'):
        if cur: print(cur)
        name = subprocess.run(['c++filt', t.split(':',1)[1].strip()], capture_output=True, text=True).stdout.strip()
        cur = {'name': re.sub(r'\(.*', '', name)[:60]}
    else:
        k, _, v = t.partition(':')
        if k.strip() in ('VGPRs', 'AGPRs', 'SGPRs', 'ScratchSize [bytes/lane]', 'Occupancy [waves/SIMD]', 'LDS Size [bytes/block]'):
            cur[k.strip().split(' ')[0]] = v.strip()
if cur: print(cur)
