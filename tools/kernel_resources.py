"""Registers, scratch and LDS of every kernel of the library, from the compiler's own output (no GPU needed):
    python tools/kernel_resources.py [nof_mlp.hip ...]        (default: every csrc/*.hip)
Compiles each source with the flags of bundlesdf_amd/build.py plus -save-temps and reads the .amdhsa_* directives.  Scratch > 0
means spills: in the matrix-core kernels they cost time (a spilled address register is reloaded behind s_waitcnt vmcnt(0)) and, as
long as inline-assembly VALU instructions were around, they came with miscomputed gradients."""
import glob, os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'bundlesdf_amd', 'csrc')
srcs = [os.path.join(CSRC, a) for a in sys.argv[1:]] or sorted(glob.glob(os.path.join(CSRC, '*.hip')))
with tempfile.TemporaryDirectory() as tmp:
    for src in srcs:
        obj = os.path.join(tmp, os.path.basename(src) + '.o')
        cmd = ['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off', '-munsafe-fp-atomics',
               '-I' + os.path.join(ROOT, 'include'), '-I' + CSRC, '-c', src, '-o', obj, '-save-temps=obj']
        subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
        asm = glob.glob(os.path.join(tmp, '*gfx950*.s'))[0]
        text = open(asm).read()
        os.remove(asm)
        print(f'== {os.path.basename(src)}')
        print(f"{'kernel':70s} {'vgpr':>5s} {'sgpr':>5s} {'scratch':>8s} {'lds':>7s}")
        for m in re.finditer(r'\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel', text, re.S):
            body = m.group(2)
            get = lambda key: int(re.search(r'\.amdhsa_' + key + r' (\d+)', body).group(1))
            name = subprocess.run(['c++filt', m.group(1)], capture_output=True, text=True).stdout.strip() or m.group(1)
            name = re.sub(r'\(.*', '', name).replace('void ', '')
            print(f"{name[:70]:70s} {get('next_free_vgpr'):5d} {get('next_free_sgpr'):5d} {get('private_segment_fixed_size'):8d} "
                  f"{get('group_segment_fixed_size'):7d}")
