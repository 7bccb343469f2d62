#!/bin/bash
# A/B build of libnof_hip.so from an EDITED device assembly of nof_mlp.hip (instruction-level bisects):
#   tools/asm_variant.sh NAME "<defines/flags>" EDIT.py     -> bundlesdf_amd/ab_NAME.so
# EDIT.py is run as `python EDIT.py in.s out.s`.  Steps: device code to .s, edit, assemble, lld, bundle, host compile with the
# bundle embedded, link with the regular objects of the other sources (bundlesdf_amd/csrc/build/).
set -e
NAME=$1; DEFS=$2; EDIT=$3
R=$(cd "$(dirname "$0")/.." && pwd); W=/tmp/isa/asm_$NAME; mkdir -p $W
LLVM=/opt/rocm/lib/llvm/bin
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -Wno-unused-result -Wno-pass-failed -fno-honor-nans $DEFS"
SRC=$R/bundlesdf_amd/csrc/nof_mlp.hip
[ -f $W/dev.s ] || /opt/rocm/bin/hipcc $FLAGS --cuda-device-only -S -x hip $SRC -o $W/dev.s 2>/dev/null
python $EDIT $W/dev.s $W/dev_edit.s
$LLVM/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c $W/dev_edit.s -o $W/dev.o
$LLVM/lld -flavor gnu -m elf64_amdgpu --no-undefined -shared -o $W/dev.out $W/dev.o
$LLVM/clang-offload-bundler -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950 -input=/dev/null -input=$W/dev.out -output=$W/dev.hipfb
/opt/rocm/bin/hipcc $FLAGS --cuda-host-only -Xclang -fcuda-include-gpubinary -Xclang $W/dev.hipfb -x hip -c $SRC -o $W/host.o 2>/dev/null
OBJS=""; for s in nof_capi nof_hash nof_trace nof_loss nof_pose nof_mesh nof_texture; do OBJS="$OBJS $R/bundlesdf_amd/csrc/build/$s.o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,-Bsymbolic -o $R/bundlesdf_amd/ab_$NAME.so $W/host.o $OBJS
echo $R/bundlesdf_amd/ab_$NAME.so
