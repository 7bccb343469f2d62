"""usage: edit_scalarize.py MODE in.s out.s -- rewrite v_pk_{mul,add}_f32 of the ENCODE region (before the first v_mfma) of every
k_enc_mlp_fwd kernel into two scalar VOP3 instructions with the same arithmetic (a temp VGPR is added to the kernel).
MODE: all | opsel (op_sel: present) | opselhi (only op_sel_hi non-default) | plain | mul | add | none"""
import re, sys
mode, src, dst = sys.argv[1], sys.argv[2], sys.argv[3]
lines = open(src).read().split('\n')

def half(tok, h):
    m = re.match(r'([vs])\[(\d+):(\d+)\]$', tok)
    if m:
        return f'{m.group(1)}{int(m.group(2)) + h}'
    return tok                                            # constant / literal: the same value for both halves

def parse_mods(rest):
    mods = {'op_sel': [0, 0], 'op_sel_hi': [1, 1], 'neg_lo': [0, 0], 'neg_hi': [0, 0]}
    for k, a, b in re.findall(r'(op_sel_hi|op_sel|neg_lo|neg_hi):\[(\d),(\d)\]', rest):
        mods[k] = [int(a), int(b)]
    return mods

out, inside, enc, n, kname, temp = [], False, False, 0, None, {}
i = 0
for ln in lines:
    m = re.match(r'^(_Z13k_enc_mlp_fwd\S+):', ln)
    if m:
        inside, enc, kname = True, True, m.group(1)
    if inside and 'v_mfma' in ln:
        enc = False
    if inside and 's_endpgm' in ln:
        inside = False
    mm = re.match(r'^\tv_pk_(mul|add)_f32 (v\[\d+:\d+\]), ([^,]+), (\S+)(.*)$', ln)
    if inside and enc and mm:
        kind, D, A, B, rest = mm.groups()
        B = B.rstrip(',')
        mods = parse_mods(rest)
        has_opsel = 'op_sel:' in rest
        has_opselhi = mods['op_sel_hi'] != [1, 1]
        const_ops = [not re.match(r'[vs]\[', t) for t in (A, B)]
        # a constant operand is printed with op_sel_hi 0 for it: ignore that when classifying
        eff_hi_nondefault = any(mods['op_sel_hi'][k] != 1 and not const_ops[k] for k in range(2))
        cls = 'opsel' if has_opsel else ('opselhi' if eff_hi_nondefault else 'plain')
        crossed_inplace = any((tok == D and mods['op_sel'][k] == 1 and mods['op_sel_hi'][k] == 0) for k, tok in enumerate((A, B)))
        take = mode == 'all' or mode == cls or mode == kind or (mode == 'crossed' and crossed_inplace) or (mode == 'opsel_not_crossed' and cls == 'opsel' and not crossed_inplace) or (mode == 'bcast_src0' and has_opsel and mods['op_sel'][0] == 1 and mods['op_sel_hi'][0] == 1 and not const_ops[0]) or (mode == 'bcast_src1' and has_opsel and mods['op_sel'][1] == 1 and mods['op_sel_hi'][1] == 1 and not const_ops[1])
        if take:
            t = temp.setdefault(kname, None)
            op = 'v_mul_f32_e64' if kind == 'mul' else 'v_add_f32_e64'
            def operand(tok, h, neg):
                s = half(tok, h)
                return ('-' + s) if neg else s
            lo = f'{operand(A, mods["op_sel"][0], mods["neg_lo"][0])}, {operand(B, mods["op_sel"][1], mods["neg_lo"][1])}'
            hi = f'{operand(A, mods["op_sel_hi"][0] if not const_ops[0] else 0, mods["neg_hi"][0])}, {operand(B, mods["op_sel_hi"][1] if not const_ops[1] else 0, mods["neg_hi"][1])}'
            out.append(f'\t{op} vTEMP_{kname}, {hi}')
            out.append(f'\t{op} {half(D, 0)}, {lo}')
            out.append(f'\tv_mov_b32_e32 {half(D, 1)}, vTEMP_{kname}')
            n += 1
            continue
    out.append(ln)
text = '\n'.join(out)
# give every rewritten kernel a temp VGPR: bump its register counts
for k in temp:
    i0 = text.index('\n' + k + ':')
    j0 = text.index('.end_amdhsa_kernel', i0)
    blk = text[i0:j0]
    nv = int(re.search(r'\.amdhsa_next_free_vgpr (\d+)', blk).group(1))
    acc = int(re.search(r'\.amdhsa_accum_offset (\d+)', blk).group(1))
    assert nv <= acc + 0 or True
    new = (max(nv, acc) + 4) // 4 * 4
    blk = re.sub(r'\.amdhsa_next_free_vgpr \d+', f'.amdhsa_next_free_vgpr {new}', blk)
    blk = re.sub(r'\.amdhsa_accum_offset \d+', f'.amdhsa_accum_offset {new}', blk)
    blk = blk.replace(f'vTEMP_{k}', f'v{max(nv, acc)}')
    text = text[:i0] + blk + text[j0:]
print(mode, 'rewritten', n, 'kernels', len(temp), file=sys.stderr)
open(dst, 'w').write(text)
