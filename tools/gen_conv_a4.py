#!/usr/bin/env python3
"""Generates goal_force_amd/csrc/gf_conv_a4_loop.inc: the K loop of the 4-wave direct 3x3x3 convolution of the Wan VAE's 192- and
384-channel levels (conv_a4_kernel, gf_conv_a4.hip) as ONE inline-asm statement with hand-allocated registers.

    python tools/gen_conv_a4.py            # rewrites the .inc (committed; the build does not run this)

The loop is the 4-wave GEMM's (tools/gen_gemm_a4.py: one wave per SIMD, accumulators in AGPRs, every LDS read, LDS-DMA piece,
counted wait and barrier at a fixed MFMA slot) with two changes:

  * the C tile is 256 rows x 192 columns (a wave = 128 x 96 = 8 x 6 MFMA tiles, 96 MFMAs per 64-wide K tile instead of 128):
    Cout = 192 and 384 are whole tiles (the 256-wide tile would run a quarter of its MFMAs on padding);
  * the A operand is the activation itself.  It lives in a ZERO-BORDERED buffer [2 + T, H + 2, W + 2, C] (two history frames in
    front: the causal padding; one pixel of zeros around every frame: the spatial padding), and GEMM row m is the padded position
    m = (t (H + 2) + y) (W + 2) + x.  Tap (dt, dy, dx) of row m is then the buffer row m + (dt (H + 2) + dy) (W + 2) + dx: a constant
    row shift per tap, no border test, no gather — the A tile of a K tile is 256 CONSECUTIVE buffer rows x 64 channels, fetched by
    the same `buffer_load ... lds` pieces as a plain GEMM's.  Along K = (dt, dy, dx, cin) the source advances by 128 bytes per K
    tile inside a pixel row's three taps (3 C contiguous channels), by `jr` at the end of such a run and by `jf` at the end of a
    frame's three runs: ten scalar instructions per K tile keep the two counters (no table, no memory access).
    Rows of the padded border are computed like any other and dropped by the epilogue (2.6 % of the rows at 120 x 208).

Slots of one iteration (96 MFMAs; tile t multiplies from registers, tile t+1 sits in the other LDS stage, tile t+2 is in flight):
  0..7    the 8 A fragments of k-sub-step 1 of tile t; lgkmcnt(0); barrier B1 at 11 -> the stage's A region is dead;
  12..40  one staging piece of A(t+2) every 4 MFMAs (8 pieces), the first six followed by one B fragment read of sub-step 1;
  46      barrier B2 -> the 6 pieces of B(t+2) (47, 51, .. 67; the one that would fall between the counted wait and its barrier moves);
  64      s_waitcnt vmcnt(n): everything older than this iteration's pieces has landed; barrier B3 -> the 8 + 6 fragments of
          k-sub-step 0 of tile t+1 from the other stage, one read every 2 slots.
"""
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "goal_force_amd", "csrc", "gf_conv_a4_loop.inc")

NI, NJ = 8, 6                                         # 16-row blocks of A, 16-column blocks of W per wave
NS = 2 * NI * NJ                                      # MFMA slots per K tile (two k-sub-steps of 32)
PA, PB = 8, 6                                         # staging pieces per wave and K tile: A 256 rows, W 192 rows (8 rows per piece)
# ---- register plan -------------------------------------------------------------------------------------------------
A_K0, B_K0, A_K1, B_K1 = 128, 160, 192, 224          # fragment i of a set: v[base + 4 i : base + 4 i + 3]
SRD_A, SRD_B = 60, 64                                 # s[60:63], s[64:67]
SOFF_A, SOFF_B = 36, 44                               # s[36:43], s[44:49]: row-group offsets of this wave's pieces
S_M0SAVE, S_CNT, S_WR, S_T, S_NRA, S_NRB = 52, 53, 54, 55, 56, 57
S_R, S_STEPA, S_Q, S_J, S_RUN, S_JR, S_JF = 58, 59, 68, 69, 70, 71, 72
STAGE = 65536
B_TILE = 32768
CLOBBER_S = list(range(36, 60)) + list(range(60, 73))
V_FIRST = 128                                         # v[128:255] are the loop's fragment registers
PIECE_STEP = 0x1000                                   # LDS bytes between a wave's consecutive pieces (the four waves interleaved)
FRAG_STEP = 2048                                      # LDS bytes between the fragments of consecutive 16-row blocks
# schedule parameters (the shipped values are the defaults; tools/conv_a4_variants.sh sweeps them into build/variants/)
DS = int(os.environ.get("CONV_A4_DS", "4"))           # MFMA slots between staging pieces
WAIT_SLOT = int(os.environ.get("CONV_A4_WAIT", "64")) # slot of the counted wait for tile t+1
RD_STEP = int(os.environ.get("CONV_A4_RDSTEP", "2"))  # slots between the sub-step-0 fragment reads behind it


def advance_a():
    """The A source's step to the next K tile: 128 bytes inside a run of 3 C / 64 tiles (a pixel row's three taps are contiguous),
    `jr` at the end of a run (next pixel row), `jf` at the end of a frame's third run (next frame).  R = tiles left in the run,
    Q = runs left in the frame."""
    return [f"s_sub_u32 s{S_R}, s{S_R}, 1", f"s_cmp_eq_u32 s{S_Q}, 1", f"s_cselect_b32 s{S_J}, s{S_JF}, s{S_JR}",
            f"s_cmp_eq_u32 s{S_R}, 0", f"s_cselect_b32 s{S_STEPA}, s{S_J}, 128", f"s_cselect_b32 s{S_R}, s{S_RUN}, s{S_R}",
            f"s_cselect_b32 s{S_T}, 1, 0", f"s_sub_u32 s{S_Q}, s{S_Q}, s{S_T}", f"s_cmp_eq_u32 s{S_Q}, 0",
            f"s_cselect_b32 s{S_Q}, 3, s{S_Q}"]


ADVANCE_V = [f"v_add_u32 %[voffA], s{S_STEPA}, %[voffA]", "v_add_u32 %[voffB], 0x80, %[voffB]"]


def v4(base, i):
    return f"v[{base + 4 * i}:{base + 4 * i + 3}]"


def acc(i, j):
    b = (i * 8 + j) * 4                                # the GEMM kernel's numbering (j < 6 used): a4_acc<(i * 8 + j) * 4 + r>
    return f"a[{b}:{b + 3}]"


def mfma(half, g):
    i, j = g // NJ, g % NJ
    a, b = (A_K0, B_K0) if half == 0 else (A_K1, B_K1)
    return f"v_mfma_f32_16x16x32_bf16 {acc(i, j)}, {v4(b, j)}, {v4(a, i)}, {acc(i, j)}"


def dma(which, p, back_to_back=False):
    srd, soff, voff = (SRD_A, SOFF_A, "%[voffA]") if which == 0 else (SRD_B, SOFF_B, "%[voffB]")
    return [f"buffer_load_dwordx4 {voff}, s[{srd}:{srd + 3}], s{soff + p} offen lds", f"s_add_u32 m0, m0, {PIECE_STEP:#x}"] + \
        (["s_nop 0"] if back_to_back else [])


def dma_tile(fill=None):
    """The 8 + 6 pieces of one tile (prologue): back to back, or each followed by its share of other work."""
    out = [f"s_mov_b32 m0, s{S_WR}", "s_nop 0"]
    for p in range(PA):
        out += dma(0, p, fill is None) + ([] if fill is None else fill[p])
    out += [f"s_add_u32 m0, s{S_WR}, {B_TILE}", "s_nop 0"]
    for p in range(PB):
        out += dma(1, p, fill is None) + ([] if fill is None else fill[PA + p])
    out += advance_a() + ["s_nop 0"] + ADVANCE_V
    return out


def rd(dst_base, i, addr, extra=0):
    off = i * FRAG_STEP + extra
    return f"ds_read_b128 {v4(dst_base, i)}, {addr}" + (f" offset:{off}" if off else "")


def gen():
    L = []
    # ---- prologue ------------------------------------------------------------------------------------------------------
    L += [f"s_mov_b32 s{S_M0SAVE}, m0"]
    L += [f"s_mov_b32 s{SRD_A}, %[aLo]", f"s_mov_b32 s{SRD_A + 1}, %[aHi]", f"s_mov_b32 s{SRD_A + 2}, %[nrA]",
          f"s_mov_b32 s{SRD_A + 3}, 0x00020000",
          f"s_mov_b32 s{SRD_B}, %[bLo]", f"s_mov_b32 s{SRD_B + 1}, %[bHi]", f"s_mov_b32 s{SRD_B + 2}, %[nrB]",
          f"s_mov_b32 s{SRD_B + 3}, 0x00020000",
          f"s_mov_b32 s{S_NRA}, %[nrA]", f"s_mov_b32 s{S_NRB}, %[nrB]", f"s_mov_b32 s{S_CNT}, %[nk]",
          f"s_mov_b32 s{S_WR}, %[ldsW]",
          f"s_mov_b32 s{S_RUN}, %[run]", f"s_mov_b32 s{S_R}, %[run]", f"s_mov_b32 s{S_Q}, 3",
          f"s_mov_b32 s{S_JR}, %[jr]", f"s_mov_b32 s{S_JF}, %[jf]"]
    L += [f"s_mov_b32 s{SOFF_A}, %[soA]", f"s_mov_b32 s{SOFF_B}, %[soB]"]
    for p in range(1, PA):
        L += [f"s_add_u32 s{SOFF_A + p}, s{SOFF_A + p - 1}, %[stA]"]
    for p in range(1, PB):
        L += [f"s_add_u32 s{SOFF_B + p}, s{SOFF_B + p - 1}, %[stB]"]
    L += dma_tile()                                                   # tile 0 -> stage 0
    L += [f"s_xor_b32 s{S_WR}, s{S_WR}, {STAGE}",
          f"s_cmp_gt_u32 s{S_CNT}, 1", f"s_cselect_b32 s{SRD_A + 2}, s{S_NRA}, 0", f"s_cselect_b32 s{SRD_B + 2}, s{S_NRB}, 0",
          "s_nop 1"]
    # tile 1 -> stage 1 (zeros past K), the accumulators zeroed between its pieces (under the latency of tile 0)
    regs = [(i * 8 + j) * 4 + r for i in range(NI) for j in range(NJ) for r in range(4)]
    n = PA + PB
    fills = [[f"v_accvgpr_write_b32 a{x}, 0" for x in regs[q * len(regs) // n:(q + 1) * len(regs) // n]] for q in range(n)]
    L += dma_tile(fills)
    L += [f"s_xor_b32 s{S_WR}, s{S_WR}, {STAGE}"]
    L += [f"s_waitcnt vmcnt({PA + PB})", "s_barrier"]
    for i in range(NI):
        L.append(rd(A_K0, i, "%[rdA0]"))
    for j in range(NJ):
        L.append(rd(B_K0, j, "%[rdB0]"))
    L += ["s_waitcnt lgkmcnt(0)"]

    # ---- the loop ------------------------------------------------------------------------------------------------------
    ev = {}                      # MFMA slot -> instructions issued right after it

    def at(slot, *ins):
        assert 0 <= slot < NS, slot
        ev.setdefault(slot, []).extend(ins)

    # tile t+2 exists iff remaining > 2
    at(0, f"s_cmp_gt_u32 s{S_CNT}, 2", f"s_cselect_b32 s{SRD_A + 2}, s{S_NRA}, 0")
    at(1, f"s_cselect_b32 s{SRD_B + 2}, s{S_NRB}, 0")
    b1 = NI + 3                                           # B1: behind the last sub-step-1 read of A
    for i in range(NI):
        at(i, rd(A_K1, i, "%[rdA1]"))
    at(b1 - 2, f"s_mov_b32 m0, s{S_WR}")
    at(b1 - 1, "s_waitcnt lgkmcnt(0)")
    at(b1, "s_barrier")
    a_slots = [b1 + 1 + DS * p for p in range(PA)]
    b2 = max(a_slots[-1] + 3, a_slots[NJ - 1] + 4) if DS != 4 else a_slots[-1] + 6     # B2: behind the last A piece and the last sub-step-1 read of B
    b_slots = [b2 + 1 + DS * p for p in range(PB)]
    b_slots = [x + 2 if x in (WAIT_SLOT, WAIT_SLOT + 1) else x for x in b_slots]    # not between the counted wait and its barrier
    for p in range(PA):
        at(a_slots[p], dma(0, p)[0])
        if p < NJ:
            at(a_slots[p] + 1, rd(B_K1, p, "%[rdB1]"))
        if p < PA - 1:
            at(a_slots[p] + 1, dma(0, p)[1])              # the M0 step: one MFMA behind the piece that still has to read M0
    assert a_slots[NJ - 1] + 1 < b2 - 1 and a_slots[NJ - 1] + 1 < NI * NJ - 2, "the sub-step-1 reads of B must land before the second half starts"
    assert b2 - 2 > a_slots[-1], "M0 moves to the W region only behind the last A piece"
    at(b2 - 2, f"s_add_u32 m0, s{S_WR}, {B_TILE}")
    at(b2 - 1, "s_waitcnt lgkmcnt(0)")
    at(b2, "s_barrier")
    if b2 - 1 >= NI * NJ:                                  # the barrier sits in the second half: its MFMAs need the fragments earlier
        at(NI * NJ - 2, "s_waitcnt lgkmcnt(0)")
    for p in range(PB):
        at(b_slots[p], dma(1, p)[0])
        if p < PB - 1:
            at(b_slots[p] + 1, dma(1, p)[1])
    last_piece = b_slots[-1]
    n_before = sum(1 for x in a_slots + b_slots if x < WAIT_SLOT)
    assert a_slots[-1] < WAIT_SLOT and NI * NJ <= WAIT_SLOT, "the counted wait sits behind the A pieces and behind the first half's MFMAs"
    # the read addresses move to the other stage between the last sub-step-1 read and the first sub-step-0 read of the next tile
    at(NI * NJ + 2, "v_xor_b32 %[rdA0], 0x10000, %[rdA0]", "v_xor_b32 %[rdA1], 0x10000, %[rdA1]")
    at(NI * NJ + 3, "v_xor_b32 %[rdB0], 0x10000, %[rdB0]", "v_xor_b32 %[rdB1], 0x10000, %[rdB1]")
    at(WAIT_SLOT, f"s_waitcnt vmcnt({n_before})")
    at(WAIT_SLOT + 1, "s_barrier")
    for i in range(NI):
        at(WAIT_SLOT + 2 + RD_STEP * i, rd(A_K0, i, "%[rdA0]"))
    for j in range(NJ):
        at(WAIT_SLOT + 2 + RD_STEP * NI + RD_STEP * j, rd(B_K0, j, "%[rdB0]"))
    assert WAIT_SLOT + 2 + RD_STEP * NI + RD_STEP * (NJ - 1) <= NS - 3
    adv = last_piece + 1                                   # the staging position moves on only behind the iteration's last piece
    at(adv, f"s_xor_b32 s{S_WR}, s{S_WR}, {STAGE}", *advance_a()[:3])
    at(adv + 1, *advance_a()[3:7])
    at(adv + 2, *advance_a()[7:])
    at(adv + 4, *ADVANCE_V)
    assert adv + 4 < NS - 4
    at(NS - 4, f"s_sub_u32 s{S_CNT}, s{S_CNT}, 1")
    at(NS - 3, f"s_cmp_eq_u32 s{S_CNT}, 0")
    at(NS - 2, "s_waitcnt lgkmcnt(0)")
    L.append("1:")
    for s in range(NS):
        L.append(mfma(s // (NI * NJ), s % (NI * NJ)))
        L += ev.get(s, [])
    L += ["s_cbranch_scc0 1b"]
    # ---- drain: the last two iterations staged zero tiles; they must have landed (and every wave must be past its reads)
    # before the epilogue reuses LDS.  MFMA results need 4 passes + margin before v_accvgpr_read.
    L += ["s_waitcnt vmcnt(0)", "s_nop 7", "s_nop 7", f"s_mov_b32 m0, s{S_M0SAVE}", "s_barrier"]
    return L, dict(a_slots=a_slots, b_slots=b_slots, b1=b1, b2=b2, n_before=n_before, adv=adv)


def check(lines):
    """Replay the slot plan: register sets are not overwritten while MFMAs still read them, SCC is not clobbered between a
    compare and its selects, and the loop-closing compare is the last SCC writer before the branch."""
    loop = lines[lines.index("1:") + 1:lines.index("s_cbranch_scc0 1b")]
    slot = -1
    last_read_of = {}                     # fragment register -> last MFMA slot that reads it
    writes = []                           # (slot, first register) of ds_read destinations
    for ins in loop:
        if ins.startswith("v_mfma"):
            slot += 1
            ops = ins.split(None, 1)[1].split(", ")
            for o in ops[1:3]:
                lo = int(o[2:o.index(":")])
                last_read_of[lo] = slot
        elif ins.startswith("ds_read_b128"):
            writes.append((slot, int(ins.split()[1][2:].split(":")[0])))
    half = NI * NJ
    for s, r in writes:
        sub1 = r >= A_K1                   # sub-step-1 sets are read by the second half's MFMAs, sub-step-0 sets by the first half's
        if sub1:
            assert s < half, ("a sub-step-1 fragment must be loaded during the first half", s, r)
        else:
            assert s >= last_read_of[r], ("a sub-step-0 fragment is overwritten while the first half still reads it", s, r)
    scc_writers = ("s_cmp", "s_sub_u32", "s_add_u32", "s_xor_b32", "s_lshl")
    tail = [i for i in loop if i.startswith(scc_writers) or i.startswith("s_cselect")]
    assert tail[-1].startswith("s_cmp_eq_u32 s%d, 0" % S_CNT), tail[-3:]
    pend = None
    for ins in loop:
        if ins.startswith("s_cmp"):
            pend = ins
        elif ins.startswith("s_cselect"):
            assert pend is not None, ("select without a live compare", ins)
        elif ins.startswith(scc_writers):
            pend = None
    assert sum(1 for i in loop if i.startswith("buffer_load_dwordx4")) == PA + PB
    assert sum(1 for i in loop if i.startswith("ds_read_b128")) == 2 * (NI + NJ)


def emit(name, lines, plan):
    n_mfma = sum(1 for l in lines if l.startswith("v_mfma"))
    assert n_mfma == NS, n_mfma
    body = "\n".join(f'    "{l}\\n\\t"' for l in lines)
    vclob = ", ".join(f'"v{r}"' for r in range(V_FIRST, 256))
    aclob = ", ".join(f'"a{r}"' for r in range(256))
    sclob = ", ".join(f'"s{r}"' for r in CLOBBER_S)
    text = f"""// GENERATED by tools/gen_conv_a4.py — do not edit.  The K loop of conv_a4_kernel as one asm statement ({NS} MFMAs per K tile).
// slot plan: A pieces {plan['a_slots']}, W pieces {plan['b_slots']}, barriers {plan['b1']} / {plan['b2']} / {WAIT_SLOT + 1}, s_waitcnt vmcnt({plan['n_before']}) at {WAIT_SLOT}.
// operands: voffA/voffB (per-lane source byte offsets; A advances by 128 / jr / jf per K tile, W by 128), rdA0/rdA1/rdB0/rdB1 (LDS
// fragment read addresses of k-sub-steps 0/1, stage toggled by XOR 0x10000), aLo/aHi/nrA, bLo/bHi/nrB (tile row base + valid
// bytes), soA/stA, soB/stB (this wave's first row-group offset and the 32-row stride, bytes), ldsW (this wave's LDS write base in
// stage 0), nk (K tiles >= 1), run (K tiles per contiguous run = 3 C / 64), jr / jf (A's byte step at the end of a run / of a
// frame's third run).  Accumulators are left in a[(i*8+j)*4 + r] = C[16 i + lane%16][16 j + 4 (lane/16) + r], i < 8, j < 6.
#define {name}(voffA, voffB, rdA0, rdA1, rdB0, rdB1, aLo, aHi, nrA, bLo, bHi, nrB, soA, stA, soB, stB, ldsW, nk, run, jr, jf) \\
    asm volatile( \\
{body.replace(chr(10), " " + chr(92) + chr(10))} \\
        : [voffA] "+v"(voffA), [voffB] "+v"(voffB), [rdA0] "+v"(rdA0), [rdA1] "+v"(rdA1), [rdB0] "+v"(rdB0), [rdB1] "+v"(rdB1) \\
        : [aLo] "s"(aLo), [aHi] "s"(aHi), [nrA] "s"(nrA), [bLo] "s"(bLo), [bHi] "s"(bHi), [nrB] "s"(nrB), [soA] "s"(soA), \\
          [stA] "s"(stA), [soB] "s"(soB), [stB] "s"(stB), [ldsW] "s"(ldsW), [nk] "s"(nk), [run] "s"(run), [jr] "s"(jr), [jf] "s"(jf) \\
        : "memory", "scc", "vcc", {sclob}, \\
          {vclob}, \\
          {aclob})
"""
    return text


def main():
    lines, plan = gen()
    check(lines)
    text = emit("GF_CONV_A4_LOOP_ASM", lines, plan)
    out = os.environ.get("CONV_A4_OUT", OUT)
    with open(out, "w") as f:
        f.write(text)
    print(f"wrote {out}: {plan}")


if __name__ == "__main__":
    main()
