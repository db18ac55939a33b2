"""Basic-block map of one kernel in a hipcc -S dump: per block, instruction-class counts and branch targets (to find the hot loop).
usage: python tools/isa_cfg.py file.s kernel_substring"""
import re, sys
lines = open(sys.argv[1]).read().split('\n'); key = sys.argv[2]
start = next(i for i, l in enumerate(lines) if l.startswith('_Z') and key in l and ':' in l)
end = next(i for i in range(start, len(lines)) if 's_endpgm' in lines[i])
name = 'entry'; cur = []; blocks = []
for l in lines[start + 1:end + 1]:
    t = l.strip()
    m = re.match(r'^(\.LBB\d+_\d+):', t)
    if m: blocks.append((name, cur)); name = m.group(1); cur = []; continue
    if not t or t.startswith(';') or t.startswith('.'): continue
    cur.append(t)
blocks.append((name, cur))
for n, b in blocks:
    br = [x for x in b if x.startswith(('s_cbranch', 's_branch'))]
    c = lambda pre: sum(1 for x in b if x.startswith(pre))
    print(f"{n:10s} n={len(b):4d} v={c('v_') - c('v_mfma'):3d} mfma={c('v_mfma'):3d} s={c('s_'):3d} (wait {c('s_waitcnt')}, nop {c('s_nop')}) ds={c('ds_'):3d} g={c('global_'):3d}  ",
          ' | '.join(x.replace('s_cbranch_', '').replace('s_branch', 'br') for x in br))
