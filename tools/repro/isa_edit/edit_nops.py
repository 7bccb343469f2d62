"""usage: edit_nops.py MODE in.s out.s -- insert s_nop around instructions of the ENCODE region (before the first v_mfma) of every
k_enc_mlp_fwd kernel.  MODE: after_vmov | after_vpk | before_vpk | after_valu_all | pk_block_wait"""
import re, sys
mode, src, dst = sys.argv[1], sys.argv[2], sys.argv[3]
lines = open(src).read().split('\n')
out, inside, enc, n = [], False, False, 0
for ln in lines:
    if re.match(r'^_Z13k_enc_mlp_fwd\S+:', ln):
        inside, enc = True, True
    if inside and 'v_mfma' in ln:
        enc = False
    if inside and 's_endpgm' in ln:
        inside = False
    ins = ln.strip().split(' ')[0] if ln.startswith('\t') else ''
    if inside and enc and ins:
        if mode == 'before_vpk' and ins.startswith('v_pk_'):
            out.append('\ts_nop 4'); n += 1
        out.append(ln)
        if mode == 'after_vmov' and ins.startswith('v_mov_b32'):
            out.append('\ts_nop 1'); n += 1
        if mode == 'after_vpk' and ins.startswith('v_pk_'):
            out.append('\ts_nop 1'); n += 1
        if mode == 'after_valu_all' and ins.startswith('v_') and not ins.startswith('v_readfirstlane'):
            out.append('\ts_nop 0'); n += 1
        continue
    out.append(ln)
print(mode, 'inserted', n, file=sys.stderr)
open(dst, 'w').write('\n'.join(out))
