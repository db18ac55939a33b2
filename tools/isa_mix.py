"""Instruction-class map of one kernel in a hipcc -S dump: per basic block, counts and a compressed class string.
usage: python tools/isa_mix.py file.s kernel_substring [--seq]"""
import re, sys
path, key = sys.argv[1], sys.argv[2]
seq = "--seq" in sys.argv
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l.split(":")[0] and ":" in l)
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
def cls(op):
    if op.startswith("v_mfma"): return "M"
    if op.startswith(("v_exp", "v_rcp", "v_rsq", "v_log", "v_sqrt")): return "t"
    if op.startswith("v_pk_") and op.endswith("f32"): return "P"
    if op.startswith("v_pk_"): return "p"
    if op.startswith("v_cvt"): return "c"
    if op.startswith(("v_permlane", "v_mov_b32_dpp", "v_readlane", "v_readfirstlane", "v_writelane")) or "dpp" in op: return "x"
    if op.startswith("v_accvgpr"): return "a"
    if op.startswith("v_"): return "v"
    if op.startswith("ds_"): return "d"
    if op.startswith(("global_", "buffer_", "scratch_", "flat_")): return "g"
    if op.startswith("s_waitcnt"): return "w"
    if op.startswith("s_barrier"): return "B"
    if op.startswith("s_nop"): return "n"
    if op.startswith("s_"): return "s"
    return "?"
blocks, cur, name = [], [], "entry"
for l in lines[start + 1:end + 1]:
    t = l.strip()
    if not t or t.startswith(";") or t.startswith("."):
        m = re.match(r"^(\.LBB\d+_\d+):", t)
        if m:
            blocks.append((name, cur)); cur, name = [], m.group(1)
        continue
    full = t.split()[0]
    c = cls(full)
    if "dpp" in t and c == "v": c = "x"
    cur.append((c, t))
blocks.append((name, cur))
tot = {}
for name, b in blocks:
    if len(b) < 40: 
        for c, _ in b: tot[c] = tot.get(c, 0) + 1
        continue
    cnt = {}
    for c, _ in b: cnt[c] = cnt.get(c, 0) + 1; tot[c] = tot.get(c, 0) + 1
    print(f"{name:12s} n={len(b):5d} ", " ".join(f"{k}:{v}" for k, v in sorted(cnt.items())))
    if seq:
        s = "".join(c for c, _ in b)
        for i in range(0, len(s), 160): print("      ", s[i:i + 160])
print("total", " ".join(f"{k}:{v}" for k, v in sorted(tot.items())))
