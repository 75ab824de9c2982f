"""Classifies the VALU instructions of the trace kernel's loop into full-rate / half-rate classes (measured by
scripts/microbench/valu_rate.hip on MI355X) and prints a weighted cost per basic-block region."""
import re, sys, glob
f = glob.glob(sys.argv[1] + '/*gfx950*.s')[0]
t = open(f).read()
name = '_ZN4blok12_GLOBAL__N_112trace_kernelILNS_7RayModeE0EEEvNS_9TraceArgsE'
a = t.index('\n' + name + ':'); b = t.index('.Lfunc_end', a)
lines = [l.strip() for l in t[a:b].split('\n') if l.strip() and not l.strip().startswith(';')]
FULL = ('v_fma_f32', 'v_mul_f32', 'v_sub_f32', 'v_subrev_f32', 'v_add_f32', 'v_mov_b32', 'v_add_u32', 'v_sub_u32', 'v_subrev_u32',
        'v_and_b32', 'v_or_b32', 'v_xor_b32', 'v_not_b32', 'v_pk_add_f32', 'v_pk_mul_f32', 'v_pk_fma_f32', 'v_mac_f32', 'v_fmac_f32')
start = next(i for i, l in enumerate(lines) if 'Loop Header' in l)
end = next(i for i in range(start, len(lines)) if lines[i].startswith('s_branch') or (lines[i].startswith('s_cbranch') and i > start + 150))
# the loop ends at the backward branch to the header label
hdr = lines[start].split(':')[0]
ends = [i for i, l in enumerate(lines) if hdr in l and i > start and ('s_branch' in l or 's_cbranch' in l)]
loop = lines[start:(ends[-1] if ends else end) + 1]
full = half = salu = other = 0
from collections import Counter
c = Counter()
for l in loop:
    op = l.split()[0]
    if op.startswith('v_'):
        base = re.sub(r'_e(32|64)$', '', op)
        if base in FULL: full += 1
        else: half += 1; c[base] += 1
    elif op.startswith('s_'): salu += 1
    else: other += 1
print(f"loop: {len(loop)} lines; VALU full-rate {full}, half-rate {half} -> cost {full + 2 * half} full-rate slots; SALU {salu}; other {other}")
print("half-rate ops:", c.most_common())
