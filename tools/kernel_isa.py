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


def main():
    ks = kernels()
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
