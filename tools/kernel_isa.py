"""Disassemble the gfx950 code objects embedded in libuitk.so and answer questions about a kernel's ISA (build-time check, no GPU):

    python tools/kernel_isa.py                        # table: kernel, VGPRs, spills, scratch, MFMA / VALU counts, scratch in loops
    python tools/kernel_isa.py --phases 'uit_encoder_kernel<0, 8, false'   # instruction mix between the barriers of the block loop

Used by tests/test_build_checks.py: the block loop of the encoder instantiations the product launches by default must contain no
scratch_ instruction (a spill reload is a vector-memory load: the compiler waits for it with s_waitcnt vmcnt(0), which drains the
weight ring's LDS-DMAs in flight -- DESIGN.md 4.2.1).

How: llvm-objcopy dumps the .hip_fatbin section, which holds one clang offload bundle per translation unit; each is unbundled for
hipv4-amdgcn-amd-amdhsa--gfx950 and disassembled with llvm-objdump.  A "loop" is the span of a backward branch.
"""
import re
import subprocess
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
LLVM = Path("/opt/rocm/lib/llvm/bin")
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
TARGET = "hipv4-amdgcn-amd-amdhsa--gfx950"


def code_objects(lib, workdir):
    fat = Path(workdir) / "fat.bin"
    subprocess.check_call([str(LLVM / "llvm-objcopy"), f"--dump-section=.hip_fatbin={fat}", str(lib), str(Path(workdir) / "discard.so")])
    data = fat.read_bytes()
    starts = [m.start() for m in re.finditer(re.escape(MAGIC), data)]
    out = []
    for i, st in enumerate(starts):
        part = Path(workdir) / f"bundle{i}.bin"
        part.write_bytes(data[st:starts[i + 1] if i + 1 < len(starts) else len(data)])
        co = Path(workdir) / f"dev{i}.co"
        r = subprocess.run([str(LLVM / "clang-offload-bundler"), "--unbundle", "--type=o", f"--input={part}", f"--targets={TARGET}",
                            f"--output={co}"], capture_output=True, text=True)
        if r.returncode == 0 and co.exists() and co.stat().st_size > 0:
            out.append(co)
    return out


def demangle(names):
    r = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
    return r.stdout.splitlines()


def kernels(lib=None):
    """-> {demangled kernel name: {"insts": [(addr, text)], "meta": {...}}} for every kernel of the library"""
    lib = Path(lib or ROOT / "uit_mobile_amd" / "libuitk.so")
    res = {}
    with tempfile.TemporaryDirectory() as wd:
        for co in code_objects(lib, wd):
            dis = subprocess.run([str(LLVM / "llvm-objdump"), "-d", "--no-show-raw-insn", str(co)], capture_output=True, text=True).stdout
            notes = subprocess.run([str(LLVM / "llvm-readelf"), "--notes", str(co)], capture_output=True, text=True).stdout
            meta = {}
            for blk in notes.split("- .agpr_count:")[1:]:
                d = dict(re.findall(r"\.(\w+):\s+(\S+)", "agpr_count:" + blk))
                if "symbol" in d:
                    meta[d["symbol"].strip("'").removesuffix(".kd")] = d
            cur, sym = None, None
            for line in dis.splitlines():
                m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
                if m:
                    sym = m.group(1)
                    cur = res.setdefault(sym, {"insts": [], "meta": meta.get(sym, {})}) if sym in meta else None
                    continue
                if cur is None:
                    continue
                m = re.match(r"^\s+(\S.*?)\s*// ([0-9A-Fa-f]+):", line)
                if m:
                    cur["insts"].append((int(m.group(2), 16), m.group(1)))
    names = list(res)
    return {d: res[n] for n, d in zip(names, demangle(names))}


def loops(insts):
    """spans (start_addr, end_addr) of backward branches, outermost first"""
    addr_of = {}
    spans = []
    for a, t in insts:
        m = re.match(r"s_cbranch_\w+ (\d+)|s_branch (\d+)", t)
        if m:
            # llvm-objdump prints the branch target as a relative dword offset; resolve: target = addr + 4 + simm16 * 4
            off = int(m.group(1) or m.group(2))
            if off >= 0x8000:
                off -= 0x10000
            tgt = a + 4 + 4 * off
            if tgt <= a:
                spans.append((tgt, a))
    return sorted(spans, key=lambda s: s[0] - s[1])


def summary(k):
    insts = k["insts"]
    lp = loops(insts)
    outer = lp[0] if lp else None
    in_outer = [t for a, t in insts if outer and outer[0] <= a <= outer[1]]
    cnt = lambda pred, seq: sum(1 for t in seq if pred(t))
    return {"vgpr": int(k["meta"].get("vgpr_count", -1)), "agpr": int(k["meta"].get("agpr_count", -1)),
            "vgpr_spill": int(k["meta"].get("vgpr_spill_count", -1)), "sgpr_spill": int(k["meta"].get("sgpr_spill_count", -1)),
            "scratch_bytes": int(k["meta"].get("private_segment_fixed_size", -1)), "instructions": len(insts),
            "mfma": cnt(lambda t: t.startswith("v_mfma"), [t for _, t in insts]),
            "scratch_insts": cnt(lambda t: t.startswith("scratch_"), [t for _, t in insts]),
            "outer_loop_instructions": len(in_outer), "outer_loop_mfma": cnt(lambda t: t.startswith("v_mfma"), in_outer),
            "outer_loop_scratch_insts": cnt(lambda t: t.startswith("scratch_"), in_outer)}


def phases(k):
    """instruction mix of the outermost loop, cut at its s_barrier instructions"""
    insts = k["insts"]
    lp = loops(insts)
    if not lp:
        return []
    lo, hi = lp[0]
    out, cur = [], {"mfma": 0, "valu": 0, "trans": 0, "lds": 0, "vmem": 0, "salu": 0, "wait": 0, "other": 0}
    for a, t in insts:
        if not (lo <= a <= hi):
            continue
        op = t.split()[0]
        if op == "s_barrier":
            out.append(cur)
            cur = dict.fromkeys(cur, 0)
            continue
        if op.startswith("v_mfma"):
            cur["mfma"] += 1
        elif re.match(r"v_(exp|log|rcp|rsq|sqrt|sin|cos)_", op):
            cur["trans"] += 1
        elif op.startswith("v_"):
            cur["valu"] += 1
        elif op.startswith("ds_"):
            cur["lds"] += 1
        elif op.startswith(("global_", "buffer_", "scratch_", "flat_")):
            cur["vmem"] += 1
        elif op in ("s_waitcnt", "s_nop"):
            cur["wait"] += 1
        elif op.startswith("s_"):
            cur["salu"] += 1
        else:
            cur["other"] += 1
    out.append(cur)
    return out


# ---- MFMA hazards the compiler cannot see (round 6) -----------------------------------------------------------------------------------
# hipcc pads every hazard between instructions IT emits, but an inline-asm statement is opaque to its hazard recogniser: an asm VALU
# instruction (v_max3_f32 / v_max_f32 / v_pk_max_f16 ... written out because the compiler's own form costs more instructions) whose OUTPUT
# register the allocator places on a register that an MFMA issued just before it still has to write -- a dead lane of a ones-row MFMA's
# destination, say -- is overwritten by the MFMA's late write-back: `v_mfma v[102:105], ...` ; `v_max3_f32 v104, ...` one instruction later
# was the f16ax BNeckAttention race of round 5 (profiles/round6/f16ax_race.log): the softmax maximum of one query tile came out as a row
# sum.  The same holds for an asm instruction that READS an MFMA result too early (round 3's hmax).  mfma_hazards() walks a kernel's
# instruction stream with the wait-state rules of the LLVM gfx950 hazard recogniser (XDL write -> VALU / LDS / VMEM read or VALU write:
# passes + 3 (+ 1 on gfx950 for more than 2 passes), passes + 2 for the fp32-input MFMAs; SrcC read -> VALU write: passes - 1) and reports every violation -- compiler
# code never violates them, so whatever it finds involves an asm statement.
_MFMA_PASSES = [(r"v_mfma_f32_4x4x", 2), (r"v_mfma_f32_16x16x(16|32)_?(f16|bf16)", 4), (r"v_mfma_f32_16x16x16bf16_1k", 4),
                (r"v_mfma_f32_32x32x", 8), (r"v_mfma_scale_f32_16x16x128", 4), (r"v_mfma_f32_16x16x4_?f32", 8)]


def _regs(tok):
    """VGPR numbers named by one operand token ('v12', 'v[8:11]', otherwise nothing)"""
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", tok)
    return {int(m.group(1))} if m else set()


def mfma_hazards(insts):
    """-> [(addr, instruction, reason)]: reads / writes of an MFMA's destination, and VALU writes of its SrcC, inside the wait states the
    hardware needs and nobody padded.  The fall-through order of the whole kernel, then every loop's back edge: the instructions in front
    of a backward branch followed by the ones at its target (an MFMA at the end of a loop body against an asm statement at its head)."""
    out = _mfma_hazards_linear(insts)
    addr_index = {a: i for i, (a, _) in enumerate(insts)}
    seen = {(a, why) for a, _, why in out}
    for i, (a, t) in enumerate(insts):
        m = re.match(r"s_cbranch_\w+ (\d+)|s_branch (\d+)", t)
        if not m:
            continue
        off = int(m.group(1) or m.group(2))
        if off >= 0x8000:
            off -= 0x10000
        tgt = a + 4 + 4 * off
        if tgt > a or tgt not in addr_index:
            continue
        j = addr_index[tgt]
        tail, head = insts[max(0, i - 16):i], insts[j:j + 16]
        for b, u, why in _mfma_hazards_linear(tail + head):
            if any(b == hb for hb, _ in head) and "wait states after" in why and (b, why) not in seen:
                # only pairs that straddle the edge: the producer must sit in the tail
                if any(prod in why for _, prod in tail):
                    out.append((b, u, why + "  [across the back edge at %#x]" % a))
                    seen.add((b, why))
    return out


def _mfma_hazards_linear(insts):
    live = []                                                   # [dst set, srcC set, passes, age in wait states, text]
    valu = []                                                   # recent VALU writes: [dst set, age, is transcendental, text]
    out = []
    for a, t in insts:
        op, _, rest = t.partition(" ")
        ops = [x.strip() for x in re.split(r",(?![^\[]*\])", rest)] if rest else []
        ops = [re.split(r"\s", o)[0] for o in ops if o]
        ws = int(ops[0], 0) + 1 if op == "s_nop" and ops else 1
        if op.startswith("v_mfma"):
            passes = next((p for pat, p in _MFMA_PASSES if re.match(pat, op)), 4)
            if op.startswith("v_mfma_scale_f32_16x16x128"):         # MX operands: fp8 on either side (format 0 / 1, the default) = 8 passes, fp6 x fp4 = 4
                fmt = lambda key: int(re.search(key + r":(\d+)", t).group(1)) if re.search(key + r":(\d+)", t) else 0
                passes = 8 if (fmt("cbsz") <= 1 or fmt("blgp") <= 1) else 4
            dst, srcc = _regs(ops[0]), (_regs(ops[3]) if len(ops) > 3 else set())
            for e in live:
                e[3] += ws
            live.append([dst, srcc - dst, passes, 0, t])
            live = [e for e in live if e[3] < 24]
            for e in valu:
                e[1] += ws
            valu = [e for e in valu if e[1] < 3]
            continue
        reads, writes = set(), set()
        if op.startswith("v_"):
            two_dst = op.startswith("v_permlane") and "swap" in op
            for i, o in enumerate(ops):
                (writes if i == 0 or (two_dst and i == 1) else reads).update(_regs(o))
            if two_dst:
                reads |= writes
            if op.startswith(("v_cmp", "v_cmpx")):                  # destination is a mask, every VGPR operand is a source
                reads |= writes
                writes = set()
        elif op.startswith(("ds_", "global_", "buffer_", "flat_", "scratch_", "exp")):
            is_load = ("load" in op or "read" in op) and not op.startswith("global_load_lds")
            for i, o in enumerate(ops):
                if not (is_load and i == 0):
                    reads.update(_regs(o))                            # addresses and store data (a load's destination is not padded by anyone: LDS latency)
        # two more rules of the gfx950 recogniser an asm statement can break: a VALU write needs 2 wait states before v_permlane*_swap
        # reads the register, and the result of a transcendental (v_rcp / v_exp / v_log / v_rsq / v_sqrt / v_sin / v_cos) 1 wait state
        # before another kind of VALU instruction reads it
        if op.startswith("v_") and not op.startswith("v_mfma"):
            is_swap = op.startswith("v_permlane") and "swap" in op
            is_trans = re.match(r"v_(exp|log|rcp|rsq|sqrt|sin|cos)_", op) is not None
            for dst, age, trans, vt in valu:
                if is_swap and age < 2 and (reads & dst):
                    out.append((a, t, f"reads v{sorted(reads & dst)} {age} wait states after the VALU write `{vt}` (v_permlane*_swap needs 2)"))
                elif trans and not is_trans and age < 1 and (reads & dst):
                    out.append((a, t, f"reads v{sorted(reads & dst)} right behind the transcendental `{vt}` (needs 1 wait state)"))
        if reads or writes:
            for dst, srcc, passes, age, mt in live:
                sgemm = re.match(r"v_mfma_f32_\d+x\d+x\d+_?(16b_)?f32", mt.split()[0]) is not None     # fp32 inputs: not an XDL op
                need = passes + 2 if sgemm else passes + 3 + (1 if passes != 2 else 0)
                if age < need and ((reads | writes) & dst):
                    kind = "writes" if writes & dst else "reads"
                    out.append((a, t, f"{kind} v{sorted((reads | writes) & dst)} {age} wait states after `{mt}` (needs {need})"))
                elif op.startswith("v_") and age < passes - 1 and (writes & srcc):
                    out.append((a, t, f"writes v{sorted(writes & srcc)} {age} wait states after `{mt}` reads it as SrcC (needs {passes - 1})"))
        for e in live:
            e[3] += ws
        live = [e for e in live if e[3] < 24]
        for e in valu:
            e[1] += ws
        valu = [e for e in valu if e[1] < 3]
        if op.startswith("v_") and writes:
            valu.append([set(writes), 0, re.match(r"v_(exp|log|rcp|rsq|sqrt|sin|cos)_", op) is not None, t])
    return out


def main():
    ks = kernels()
    if len(sys.argv) > 1 and sys.argv[1] == "--hazards":        # python tools/kernel_isa.py --hazards [library]
        ks = kernels(sys.argv[2]) if len(sys.argv) > 2 else ks
        n = 0
        for name, k in sorted(ks.items()):
            for a, t, why in mfma_hazards(k["insts"]):
                print(f"{name[:90]}  {a:#x}: {t}   <- {why}")
                n += 1
        print(f"{n} MFMA hazard violation(s) in {len(ks)} kernels")
        return
    if len(sys.argv) > 2 and sys.argv[1] == "--phases":
        for name, k in ks.items():
            if name.startswith(sys.argv[2]):
                print(name)
                tot = {}
                for i, ph in enumerate(phases(k)):
                    print(f"  segment {i:2d}: " + "  ".join(f"{key} {v:4d}" for key, v in ph.items()))
                    for key, v in ph.items():
                        tot[key] = tot.get(key, 0) + v
                print("  total     : " + "  ".join(f"{key} {v:4d}" for key, v in tot.items()),
                      f"  VALU(+trans)/MFMA {(tot['valu'] + tot['trans']) / max(tot['mfma'], 1):.2f}")
        return
    for name, k in sorted(ks.items()):
        s = summary(k)
        print(f"{name[:110]:110s} vgpr {s['vgpr']:3d} agpr {s['agpr']:3d} spill v{s['vgpr_spill']:3d} s{s['sgpr_spill']:3d} scratch {s['scratch_bytes']:4d} B  "
              f"insts {s['instructions']:6d} mfma {s['mfma']:5d} scratch_insts {s['scratch_insts']:3d} | outer loop: {s['outer_loop_instructions']:6d} insts, "
              f"{s['outer_loop_mfma']:5d} mfma, {s['outer_loop_scratch_insts']:3d} scratch")


if __name__ == "__main__":
    main()
