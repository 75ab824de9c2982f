"""Where are a kernel's scratch (spill) instructions?  Reads hipcc -S output, takes one function, finds its loops (backward branches) and prints,
per loop nest, the instruction count and the scratch / LDS / global memory instructions inside.  usage: isa_loops.py file.s substring_of_name"""
import re, sys
lines = open(sys.argv[1]).read().split('\n')
name = sys.argv[2]
start = next(i for i, l in enumerate(lines) if re.match(r'^_Z\w*' + name + r'\w*:', l))
end = next(i for i in range(start, len(lines)) if lines[i].startswith('.Lfunc_end'))
body = lines[start:end]
labels = {m.group(1): i for i, l in enumerate(body) if (m := re.match(r'^(\.LBB\d+_\d+):', l))}
loops = []
for i, l in enumerate(body):
    m = re.match(r'\s+s_cbranch_\w+\s+(\.LBB\d+_\d+)|\s+s_branch\s+(\.LBB\d+_\d+)', l)
    if m:
        tgt = m.group(1) or m.group(2)
        if tgt in labels and labels[tgt] < i: loops.append((labels[tgt], i))
DS = r'\sds_'
def is_insn(l): return l.startswith('\t') and not l.strip().startswith(('.', ';'))
def count(a, b, pat): return sum(1 for l in body[a:b + 1] if is_insn(l) and re.search(pat, l))
print(f"{name}: {sum(1 for l in body if is_insn(l))} instructions, scratch {count(0, len(body) - 1, 'scratch_')}, loops {len(loops)}")
for a, b in sorted(loops, key=lambda x: (x[0], -x[1])):
    depth = sum(1 for c, d in loops if c <= a and d >= b) - 1
    print(f"{'  ' * depth}loop lines {a}-{b}: {count(a, b, '.')} instr, scratch_load {count(a, b, 'scratch_load')}, scratch_store {count(a, b, 'scratch_store')}, ds {count(a, b, DS)}, global {count(a, b, 'global_|buffer_')}, v_readlane/writelane {count(a, b, 'v_readlane|v_writelane')}")
