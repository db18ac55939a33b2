"""CPU: checks on the BUILT library's gfx950 ISA (tools/kernel_isa.py: .hip_fatbin -> code objects -> llvm-objdump), no GPU.

The encoder's weight ring counts vmcnt: a compiler-generated scratch reload inside the block loop is a vector-memory load the compiler
waits for with s_waitcnt vmcnt(0), which drains every weight DMA in flight (1.56 ms instead of 0.47 when it happened).
So the instantiations the product launches by default must keep every loop scratch-free, whatever the compiler version does."""
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "tools"))
import kernel_isa  # noqa: E402

pytestmark = pytest.mark.skipif(not (kernel_isa.LLVM / "llvm-objdump").exists() or not (ROOT / "uit_mobile_amd" / "libuitk.so").exists(),
                                reason="llvm-objdump / built library not available")


@pytest.fixture(scope="module")
def kernels():
    return kernel_isa.kernels()


def _find(kernels, prefix):
    hits = [k for k in kernels if k.startswith(prefix)]
    assert len(hits) == 1, (prefix, hits)
    return kernels[hits[0]]


@pytest.mark.parametrize("inst,max_spill", [("<0, 8, false, false, 0, 0, 0>", 0),      # f16 / f16c, full clips, mean pooling: THE bench kernel
                                            ("<4, 8, false, false, 0, 0, 0>", 16),     # f16l6 (round 3's parity mode): spills in prologue / epilogue only
                                            ("<3, 8, false, false, 0, 0, 0>", 8),      # f16x2
                                            # the full-width-attention factories (round 4: head 0's attention output waits in LDS, the
                                            # probabilities are packed before the V stage: 117 spilled registers -> none in the block loop)
                                            ("<0, 8, false, false, 1, 0, 0>", 0),      # audio_transformer_h128_*_relu, f16 / f16c
                                            ("<0, 8, false, false, 1, 1, 0>", -16),    # audio_transformer_h128_* (GELU): <= 16 spilled; the packed-f32
                                            # erf polynomial (-7 % encoder time) leaves two loop-invariant 64-bit values that are RELOADED
                                            # inside the loop (no stores): negative = allow up to two scratch loads there
                                            ("<0, 4, false, false, 0, 0, 0>", 64)])    # the small-batch shape (4-wave workgroups): loop scratch-free
def test_default_encoder_loops_are_scratch_free(kernels, inst, max_spill):
    k = _find(kernels, f"void uitk::uit_encoder_kernel{inst}")
    s = kernel_isa.summary(k)
    assert s["mfma"] > 450 and s["vgpr"] <= 256
    loop_loads_ok = 2 if max_spill < 0 else 0
    max_spill = abs(max_spill)
    assert s["vgpr_spill"] <= max_spill, s
    # the block loop = the loop that holds most of the kernel's MFMAs (the epilogue's head-tile loop and the prologue may spill)
    mf = lambda lo, hi: sum(1 for a, t in k["insts"] if lo <= a <= hi and t.startswith("v_mfma"))
    lo, hi = max(kernel_isa.loops(k["insts"]), key=lambda sp: (mf(*sp), sp[1] - sp[0]))
    assert mf(lo, hi) >= 200, mf(lo, hi)
    in_loop = [(hex(a), t) for a, t in k["insts"] if t.startswith("scratch_") and lo <= a <= hi]
    assert len(in_loop) <= loop_loads_ok and not [t for _, t in in_loop if t.startswith("scratch_store")], in_loop
    if max_spill == 0:
        assert s["scratch_insts"] == 0 and s["scratch_bytes"] == 0, s


def test_frontend_kernel_has_no_vector_spills(kernels):
    for inst in ("<false, true, false>", "<true, true, false>"):           # f32 / int16 input, patch layout: what uitk_frontend launches
        s = kernel_isa.summary(_find(kernels, f"void uitk::mel_frontend_kernel{inst}"))
        assert s["vgpr_spill"] == 0 and s["scratch_insts"] == 0 and s["mfma"] >= 52, s


def _block_loop(k):
    """the loop that holds most of the kernel's MFMAs (the epilogue's head-tile loop and the prologue are other loops)"""
    mf = lambda lo, hi: sum(1 for a, t in k["insts"] if lo <= a <= hi and t.startswith("v_mfma"))
    lo, hi = max(kernel_isa.loops(k["insts"]), key=lambda sp: (mf(*sp), sp[1] - sp[0]))
    return lo, hi, mf(lo, hi)


# Round 6: with honest register lifetimes for the asm statements (test_no_mfma_hazard_that_the_compiler_cannot_see) the two most
# register-starved instantiations -- full-width attention + GELU on SHORT clips -- reloaded six values inside the block loop until gelu4
# ran its two halves one after the other (a scheduling barrier; same encoder time: profiles/round6/gelu_seq_ab.log).  No allowance left.
ROUND6_LOOP_SCRATCH = {}


def test_every_encoder_instantiation_has_a_scratch_free_block_loop(kernels):
    """Round 5: EVERY uit_encoder_kernel instantiation the product can launch -- short clips, GELU, full-width
    attention, the poolings, bf16 / fp8 / f16x2 / f16l6, 4-wave workgroups: all of the library's encoder kernels -- keeps its block
    loop free of scratch instructions (round 4 checked six full-clip shapes; the default family on SHORT clips had two in-loop
    reloads, the short-clip GELU / full-attention builds 13-41).  What did it: the short-clip key masks became the score MFMAs'
    start values rebuilt per block from a 48-bit scalar column mask (-14 vector registers), the GELU constants moved into scalar
    register pairs, the register-starved builds issue their DMAs in the scalar-base form and re-derive lane indices at their uses."""
    enc = {n: k for n, k in kernels.items() if n.startswith("void uitk::uit_encoder_kernel<")}
    assert len(enc) >= 60, len(enc)
    bad, spills = [], {}
    for name, k in sorted(enc.items()):
        lo, hi, n_mfma = _block_loop(k)
        assert n_mfma >= 150, (name, n_mfma)
        in_loop = [(hex(a), t) for a, t in k["insts"] if t.startswith("scratch_") and lo <= a <= hi]
        inst = name.split("uit_encoder_kernel")[1].split("(")[0]
        if len(in_loop) > ROUND6_LOOP_SCRATCH.get(inst, 0):
            bad.append((name, in_loop))
        s = kernel_isa.summary(k)
        assert s["vgpr"] <= 256, name
        spills[inst] = s["vgpr_spill"]
    assert not bad, bad
    # vector spills outside the loop (prologue / epilogue): none in the 8-wave default family; bounded everywhere
    worst = max(spills.items(), key=lambda kv: kv[1])
    print("vector registers spilled outside the block loop:", {k: v for k, v in spills.items() if v})
    assert spills["<0, 8, false, false, 0, 0, 0>"] == 0 and spills["<0, 8, true, false, 0, 0, 0>"] == 0
    assert worst[1] <= 20, worst                      # (round 5: 13; the 4-wave fp8 short-clip shape 19 with round 6's honest asm lifetimes)


def test_design_md_kernel_table_matches_the_binary():
    """Round 5: the kernel table of DESIGN.md is GENERATED from the built library (tools/design_tables.py) and must
    be the one the current binary gives -- registers, spills, MFMA and scratch counts cannot drift from the document again."""
    import design_tables
    text = (ROOT / "DESIGN.md").read_text()
    assert design_tables.BEGIN in text and design_tables.END in text
    have = text[text.index(design_tables.BEGIN):text.index(design_tables.END) + len(design_tables.END)]
    want = design_tables.isa_half()
    assert have == want, "DESIGN.md's generated kernel table is stale: run `python tools/design_tables.py --write`"


def test_no_mfma_hazard_that_the_compiler_cannot_see(kernels):
    """Round 6, the root cause of round 5's f16ax BNeckAttention race: an inline-asm VALU instruction is invisible to hipcc's hazard
    recogniser.  With a free "=v" output the register allocator placed `v_max3_f32 v104` one instruction behind `v_mfma v[102:105]`
    (rows 1..3 of the ones-row MFMA are dead on arrival): the MFMA's late write-back replaced the softmax maximum of a query tile by a row
    sum -- NaN or 2.4e-3 on logits, depending on timing between the waves.  The library that shipped in round 5 had 271 such sites
    (asm v_max / v_max3, the asm v_mbcnt pair, the GELU's packed asm) in 43 kernels, all latent but that one.  kernel_isa.mfma_hazards
    applies the recogniser's wait-state rules to the shipped ISA; every kernel must come out clean."""
    bad = [(name, hex(a), t, why) for name, k in kernels.items() for a, t, why in kernel_isa.mfma_hazards(k["insts"])]
    assert not bad, bad[:10]


def test_hazard_checker_sees_the_pattern_of_the_f16ax_race():
    """the checker itself, on the instruction pair of the failing build (and on its padded form)"""
    mk = lambda lines: [(4 * i, t) for i, t in enumerate(lines)]
    race = mk(["v_mfma_f32_16x16x16_f16 v[102:105], v[116:117], v[72:73], v[68:71]", "v_max3_f32 v104, v121, v124, v195"])
    assert len(kernel_isa.mfma_hazards(race)) == 1 and "writes v[104]" in kernel_isa.mfma_hazards(race)[0][2]
    padded = mk(["v_mfma_f32_16x16x16_f16 v[102:105], v[116:117], v[72:73], v[68:71]", "s_nop 7", "v_max3_f32 v104, v121, v124, v195"])
    assert kernel_isa.mfma_hazards(padded) == []
    srcc = mk(["v_mfma_f32_16x16x32_f16 v[246:249], v[190:193], v[126:129], v[232:235]", "v_pk_mul_f32 v[232:233], v[220:221], s[44:45]"])
    assert "SrcC" in kernel_isa.mfma_hazards(srcc)[0][2]
    early_read = mk(["v_mfma_f32_16x16x32_f16 v[0:3], v[8:11], v[12:15], v[0:3]", "s_nop 3", "v_cvt_pk_f16_f32 v20, v0, v1"])
    assert len(kernel_isa.mfma_hazards(early_read)) == 1
    chain = mk(["v_mfma_f32_16x16x32_f16 v[0:3], v[8:11], v[12:15], v[0:3]", "v_mfma_f32_16x16x32_f16 v[0:3], v[8:11], v[12:15], v[0:3]"])
    assert kernel_isa.mfma_hazards(chain) == []                    # an accumulate chain is the hardware's business
    swap = mk(["v_max_f32_e32 v1, v2, v3", "v_mov_b32_e32 v4, v1", "v_permlane16_swap_b32_e32 v1, v4"])
    assert any("permlane" in w for _, _, w in kernel_isa.mfma_hazards(swap))
    trans = mk(["v_rcp_f32_e32 v1, v2", "v_pk_mul_f32 v[4:5], v[0:1], s[2:3]"])
    assert any("transcendental" in w for _, _, w in kernel_isa.mfma_hazards(trans))
    # across a loop's back edge: the MFMA at the end of the body against the asm statement at its head
    loop = mk(["v_max3_f32 v104, v121, v124, v195", "v_mfma_f32_16x16x16_f16 v[102:105], v[116:117], v[72:73], v[68:71]", "s_cbranch_scc1 65533"])
    assert any("back edge" in w for _, _, w in kernel_isa.mfma_hazards(loop))
    padded_loop = mk(["v_max3_f32 v104, v121, v124, v195", "v_mfma_f32_16x16x16_f16 v[102:105], v[116:117], v[72:73], v[68:71]", "s_nop 7", "s_cbranch_scc1 65532"])
    assert kernel_isa.mfma_hazards(padded_loop) == []
