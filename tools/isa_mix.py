"""Instruction mix of one kernel of a hipcc -S listing, per basic block / barrier-delimited phase:
   python tools/isa_mix.py <listing.s> <substring of the mangled kernel name>
(how the per-tile prologue / staging / epilogue instruction counts quoted in DESIGN.md were taken)"""
import re, sys
src = open(sys.argv[1]).read().split('\n')
key = sys.argv[2]
i0 = next(i for i, l in enumerate(src) if key in l and re.match(r'^_Z\S+:', l))
i1 = next(i for i in range(i0, len(src)) if src[i].strip().startswith('.Lfunc_end'))
body = src[i0:i1]
def cat(op):
    if op.startswith('v_mfma'): return 'mfma'
    if op.startswith('v_'): return 'valu'
    if op.startswith('s_waitcnt'): return 'wait'
    if op.startswith('s_barrier'): return 'bar'
    if op.startswith('s_cbranch') or op.startswith('s_branch'): return 'br'
    if op.startswith('s_'): return 'salu'
    if op.startswith('ds_'): return 'lds'
    if op.startswith(('buffer_', 'global_', 'scratch_', 'flat_')): return 'vmem'
    return 'other'
seg = []; cur = {'start': 0, 'label': 'entry', 'c': {}}
for n, l in enumerate(body):
    s = l.strip()
    m = re.match(r'^(\.LBB\d+_\d+):', s)
    if m:
        seg.append(cur); cur = {'start': n, 'label': m.group(1), 'c': {}}
        continue
    if not s or s[0] in ';.': continue
    op = s.split()[0]
    c = cat(op)
    cur['c'][c] = cur['c'].get(c, 0) + 1
    if c == 'br': cur['c']['->'] = s.split()[-1]
    if c == 'bar':
        seg.append(cur); cur = {'start': n, 'label': '  (after barrier)', 'c': {}}
seg.append(cur)
tot = {}
for s in seg:
    for k, v in s['c'].items():
        if k != '->': tot[k] = tot.get(k, 0) + v
    if sum(v for k, v in s['c'].items() if k != '->') > 3: print(f"{s['start']:6d} {s['label']:22s} {s['c']}")
print('total', tot)
