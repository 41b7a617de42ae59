#!/usr/bin/env python3
"""Generates goal_force_amd/csrc/gf_gemm_a4_loop.inc: the K loop of the 4-wave bf16 GEMM (gemm_a4_kernel, gf_gemm.hip)
as ONE inline-asm statement with hand-allocated registers.

    python tools/gen_gemm_a4.py            # rewrites the .inc (committed; the build does not run this)

Why asm: the loop keeps 256 fp32 accumulators in a[0:255] and 128 fragment registers in v[128:255] live; hipcc spills
at that pressure (gemm_sl_kernel), and the schedule below needs every LDS read, LDS-DMA piece, counted wait
and barrier at a fixed MFMA slot.

Per workgroup: 256 threads = 4 waves (wm, wn) in 2 x 2, C tile 256 x 256, K step 64; wave = 128 x 128 of C = 8 x 8 MFMA tiles
of v_mfma_f32_16x16x32_bf16 (operands swapped: D = W_frag x A_frag, a lane holds 4 consecutive n of one m).
LDS: 2 stages x (A tile 256 rows x 128 B | B tile 256 rows x 128 B) = 128 KiB, 16-byte chunk c of row r at chunk c ^ (r & 7).
Staging: buffer_load_dwordx4 ... offen lds (1 KiB = 8 rows per wave instruction; piece p of wave w = rows 32 p + 8 w .. + 7), descriptor in s[60:63] / s[64:67], the
piece's row group as an SGPR soffset, the K position in the per-lane voffset: no vector address arithmetic per piece.

Pipeline (tile t multiplies from registers while tile t+1 sits in LDS stage (t+1)&1 and tile t+2 is in flight), MFMA slots 0..127:
  slots 0..7   the 8 A fragments of k-sub-step 1 of tile t (one read per slot); lgkmcnt(0); barrier B1 at slot 11 -> stage t&1's A
               region is dead;
  slots 12..   ONE staging piece every 6 MFMAs for the rest of the iteration: the 8 pieces of A(t+2) (12, 18, .. 54), each followed
               by one B fragment read of sub-step 1; lgkmcnt(0), barrier B2 at slot 60 -> the 8 pieces of B(t+2) (61, 67, .. 103);
  slot 88      s_waitcnt vmcnt(13): this iteration's 13 pieces issued so far may stay in flight, everything older — all of tile
               t+1, including the three pieces the previous iteration issued behind ITS wait — has landed; barrier B3 -> the 16
               fragments of k-sub-step 0 of tile t+1 from the other stage (one read every 2 slots) between the last three pieces.
  The spacing of the pieces is the single most important number of the schedule (gemm_ab_sched4/5/6.log, 8 interleaved rounds,
  bit-identical outputs): with all 16 pieces packed into slots 20..69 (one every 3 MFMAs, the round's first version) F->D ran at
  1.41 PFLOP/s; every 4: 1.46; every 5 with B1 moved up to slot 11: 1.50; every 6 (this): **1.52** (D->F 1.50 -> 1.53, D->D 1.50 ->
  1.535 = the vendor kernel); every 2: 1.34.  A burst of LDS-DMA instructions stalls the issuing wave and the three that meet it
  at the next barrier; the vendor kernel spreads its pieces the same way (its counted wait is vmcnt(13) too).  Second: the M0 step
  that follows a piece sits one MFMA behind it, not directly behind it (the piece still has to read M0): another +1.3 % on all
  three shapes (gemm_ab_sched7/8.log; 2, 3 or 5 MFMAs behind: the same).
Tiles past K are staged with num_records = 0 (reads return 0, no memory traffic), so the loop needs no peeled tail.

Measured and NOT kept (same box, one process, tools/gemm_ab.py; logs profiles/r02/gemm_ab_*.txt|log, write-up EXPERIMENTS.md):
  * rings of 3 A stages + 2 B stages in all 160 KiB of LDS (A tile t+3 / B tile t+2 in flight, loop unrolled 6 x): F->D +2.3 %,
    but D->D -4 % and D->F -5.5 %;
  * L2 warm-up loads, one 128-byte line per LANE six K tiles ahead: 1.43 -> 1.08 PFLOP/s (64 line look-ups per instruction
    in the texture addresser);
  * un-permuted source chunks (what a padded instead of XOR-swizzled LDS image would fetch): no difference;
  * two barriers per iteration instead of three (all 16 k-sub-step-1 reads, one barrier, then the 16 pieces two MFMAs apart):
    4 % slower on all three shapes — although a timing-only build WITHOUT barriers is 2.4 / 3.7 / 11 % faster (D->F / D->D /
    F->D) while one without the counted vmcnt wait gains nothing: the loss is waves waiting for each other, not for memory;
  * a PERSISTENT kernel (one workgroup per CU walking the tiles) that requests the next C tile's first two K tiles before the
    epilogue of the current one (two asm statements per tile, 8 KiB of epilogue scratch per wave behind the stages, epilogue in
    four passes): correct, bit-identical, and exactly as fast as one workgroup per tile (±1 %, gemm_ab_persistent_prefetch.txt) —
    the ~12 us per tile outside the K loop are not request latency or workgroup turnover;
  * a start-up skew per workgroup on top of it (all CUs otherwise reach their epilogues together): no gain at 32 ns - 0.25 us per
    workgroup, slower beyond (gemm_ab_startup_skew.txt);
  * after reading the vendor kernel's loop once more (same 128 MFMA / 32 reads / 16 pieces / 3 barriers, its third barrier at slot
    105 with three pieces behind it, pieces of the four waves interleaved 8 rows at a time): the wait at slot 94 / 100 with the
    B fragments read first (A4_WAIT_SLOT, A4_B_FIRST), the first MFMA source constant over 8 MFMAs (A4_J_OUTER) — each within
    +-1 % on all three shapes with the pieces still packed (gemm_ab_sched1/jouter.log); the interleaved row
    map, re-measured over 8 interleaved rounds: +0.5 / +1.1 / +1.3 % and KEPT (gemm_ab_rowmap2.log); an LDS image of 16-byte K-chunk PLANES staged one row per lane (64 lines per instruction, 7 of 8 pieces L1 hits):
    bit-identical and 2.2 x SLOWER (gemm_ab_planes.log); GROUP_M 16: -14 %, 4: +-2 %.
"""
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "goal_force_amd", "csrc", "gf_gemm_a4_loop.inc")

# ---- register plan -------------------------------------------------------------------------------------------------
A_K0, B_K0, A_K1, B_K1 = 128, 160, 192, 224          # fragment i of a set: v[base + 4 i : base + 4 i + 3]
SRD_A, SRD_B = 60, 64                                 # s[60:63], s[64:67]
SOFF_A, SOFF_B = 36, 44                               # s[36:43], s[44:51]: row-group offsets of this wave's 8 pieces
S_M0SAVE, S_CNT, S_WR, S_NRA, S_NRB = 52, 53, 54, 56, 57
S_POS, S_STEP, S_NK, S_WRAP = 58, 59, 68, 69          # staged K tile position (wraps at nk), voffset step of the next advance
STAGE = 65536
B_TILE = 32768
CLOBBER_S = list(range(36, 60)) + list(range(60, 70))


def advance_k():
    """Scalar half of moving the staging position one K tile on: 128 bytes, or back to the row start at the wrap."""
    return [f"s_add_u32 s{S_POS}, s{S_POS}, 1", f"s_cmp_eq_u32 s{S_POS}, s{S_NK}",
            f"s_cselect_b32 s{S_STEP}, s{S_WRAP}, 128", f"s_cselect_b32 s{S_POS}, 0, s{S_POS}"]


ADVANCE_V = [f"v_add_u32 %[voffA], s{S_STEP}, %[voffA]", f"v_add_u32 %[voffB], s{S_STEP}, %[voffB]",
             f"v_add_u32 %[pfA], s{S_STEP}, %[pfA]", f"v_add_u32 %[pfB], s{S_STEP}, %[pfB]"]
WARM = False          # L2 warm-up loads: measured 1.43 -> 1.08 PFLOP/s (one 128-byte line per LANE costs the texture addresser 64
                      # line look-ups per instruction); kept for the record, not generated
WAIT_SLOT = int(os.environ.get("A4_WAIT_SLOT", "88"))
# LDS bytes between a wave's consecutive pieces: 0x1000 = the four waves' pieces interleaved (piece p of wave w = rows 32 p + 8 w ..,
# gf_gemm.hip GF_A4_ROWMAP=1: the four waves walk down the tile together; +0.5 / +1.1 / +1.3 % on D->D / D->F / F->D against each
# wave staging its own 64 consecutive rows, 0x400 with GF_A4_ROWMAP=0)
PIECE_STEP = int(os.environ.get("A4_PIECE_STEP", "0x1000"), 0)
FRAG_STEP = 2048                                      # LDS bytes between the fragments of consecutive 16-row blocks
M0_LATE = int(os.environ.get("A4_M0_LATE", "1"))      # MFMA slots between an LDS-DMA piece and the M0 step that follows it (0: right behind it)
B_FIRST = os.environ.get("A4_B_FIRST", "0") == "1"    # experiment: after the wait read B(t+1) sub-step 0 first, A rows 3..7 early in the next iteration
MERGE_B12 = False     # True: one barrier (instead of two) between the k-sub-step-1 reads and the staging of tile t+2 — measured 4 % SLOWER
PF_TILES = 6          # L2 warm-up distance beyond the staged tile (K tiles); the instruction offset field holds <= 31
V_DUMMY = 126         # v[126:127]: destinations of the warm-up loads (never read)


def warm(which):
    """One 4-byte load per lane = one 128-byte line per row of this wave's 64 staging rows, PF_TILES K tiles ahead of the
    staging position: pulls the lines the LDS-DMA will ask for from HBM / MALL into this XCD's L2 early, so that the DMA's
    latency is an L2 hit (the two LDS stages give the DMA itself only ~1.3 iterations of lead)."""
    srd, soff, voff = (SRD_A, SOFF_A, "%[pfA]") if which == 0 else (SRD_B, SOFF_B, "%[pfB]")
    return f"buffer_load_dword v{V_DUMMY + which}, {voff}, s[{srd}:{srd + 3}], s{soff} offen offset:{PF_TILES * 128}"


def v4(base, i):
    return f"v[{base + 4 * i}:{base + 4 * i + 3}]"


def acc(i, j):
    b = (i * 8 + j) * 4
    return f"a[{b}:{b + 3}]"


J_OUTER = os.environ.get("A4_J_OUTER", "0") == "1"    # experiment: the MFMA's FIRST source operand (the W fragment) constant over 8 MFMAs instead of the second


def mfma(half, g):
    i, j = (g & 7, g >> 3) if J_OUTER else (g >> 3, g & 7)
    a, b = (A_K0, B_K0) if half == 0 else (A_K1, B_K1)
    return f"v_mfma_f32_16x16x32_bf16 {acc(i, j)}, {v4(b, j)}, {v4(a, i)}, {acc(i, j)}"


def dma(which, p, back_to_back=False):
    """One 1-KiB piece + the M0 step to the next piece's LDS address (an M0 write needs one wait state before the next LDS-DMA:
    in the loop the next piece is several MFMAs away, in the prologue an s_nop pads it)."""
    srd, soff, voff = (SRD_A, SOFF_A, "%[voffA]") if which == 0 else (SRD_B, SOFF_B, "%[voffB]")
    return [f"buffer_load_dwordx4 {voff}, s[{srd}:{srd + 3}], s{soff + p} offen lds", f"s_add_u32 m0, m0, {PIECE_STEP:#x}"] + \
        (["s_nop 0"] if back_to_back else [])


def dma_tile(fill=None):
    """16 pieces of one tile (prologue): back to back, or — `fill`, 16 lists of instructions — each followed by its share of other
    work (tile 1 is staged between the zeroing of the accumulators: nothing waits for it yet, and a burst of LDS-DMA instructions
    is expensive, see the module docstring)."""
    out = [f"s_mov_b32 m0, s{S_WR}", "s_nop 0"]
    for p in range(8):
        out += dma(0, p, fill is None) + ([] if fill is None else fill[p])
    out += [f"s_add_u32 m0, s{S_WR}, {B_TILE}", "s_nop 0"]
    for p in range(8):
        out += dma(1, p, fill is None) + ([] if fill is None else fill[8 + p])
    out += advance_k() + ADVANCE_V
    return out


def rd(dst_base, i, addr, extra=0):
    off = i * FRAG_STEP + extra
    return f"ds_read_b128 {v4(dst_base, i)}, {addr}" + (f" offset:{off}" if off else "")


def gen(whatif=0):
    """whatif (timing-only builds, wrong results): 1 = no barriers / counted waits in the loop, 2 = the K position never advances
    (every tile re-reads the first one: all staging hits L2), 4 = no staging instructions in the loop."""
    L = []
    # ---- prologue ------------------------------------------------------------------------------------------------------
    L += [f"s_mov_b32 s{S_M0SAVE}, m0"]
    L += [f"s_mov_b32 s{SRD_A}, %[aLo]", f"s_mov_b32 s{SRD_A + 1}, %[aHi]", f"s_mov_b32 s{SRD_A + 2}, %[nrA]",
          f"s_mov_b32 s{SRD_A + 3}, 0x00020000",
          f"s_mov_b32 s{SRD_B}, %[bLo]", f"s_mov_b32 s{SRD_B + 1}, %[bHi]", f"s_mov_b32 s{SRD_B + 2}, %[nrB]",
          f"s_mov_b32 s{SRD_B + 3}, 0x00020000",
          f"s_mov_b32 s{S_NRA}, %[nrA]", f"s_mov_b32 s{S_NRB}, %[nrB]", f"s_mov_b32 s{S_CNT}, %[nk]",
          f"s_mov_b32 s{S_WR}, %[ldsW]",
          # staggered start: this tile's K loop begins at K tile k0 and wraps (the sum over k is rotated, not changed)
          f"s_mov_b32 s{S_NK}, %[nk]", f"s_mov_b32 s{S_POS}, %[k0]", f"s_sub_u32 s{S_WRAP}, 128, %[kb]",
          f"s_lshl_b32 s{S_STEP}, %[k0], 7", "s_nop 0",
          f"v_add_u32 %[voffA], s{S_STEP}, %[voffA]", f"v_add_u32 %[voffB], s{S_STEP}, %[voffB]",
          f"v_add_u32 %[pfA], s{S_STEP}, %[pfA]", f"v_add_u32 %[pfB], s{S_STEP}, %[pfB]"]
    L += [f"s_mov_b32 s{SOFF_A}, %[soA]", f"s_mov_b32 s{SOFF_B}, %[soB]"]
    for p in range(1, 8):
        L += [f"s_add_u32 s{SOFF_A + p}, s{SOFF_A + p - 1}, %[stA]", f"s_add_u32 s{SOFF_B + p}, s{SOFF_B + p - 1}, %[stB]"]
    L += dma_tile()                                                   # tile 0 -> stage 0
    L += [f"s_xor_b32 s{S_WR}, s{S_WR}, {STAGE}",
          f"s_cmp_gt_u32 s{S_CNT}, 1", f"s_cselect_b32 s{SRD_A + 2}, s{S_NRA}, 0", f"s_cselect_b32 s{SRD_B + 2}, s{S_NRB}, 0",
          "s_nop 1"]
    if os.environ.get("A4_PROLOGUE_DENSE", "0") == "1":               # the round's first version: 32 pieces back to back, then the zeroing
        L += dma_tile()                                               # tile 1 -> stage 1 (zeros past K)
        L += [f"s_xor_b32 s{S_WR}, s{S_WR}, {STAGE}"]
        for r in range(256):
            L.append(f"v_accvgpr_write_b32 a{r}, 0")
    else:
        # tile 1 -> stage 1 (zeros past K), one piece per 16 accumulator registers zeroed (under the latency of tile 0)
        L += dma_tile([[f"v_accvgpr_write_b32 a{16 * q + r}, 0" for r in range(16)] for q in range(16)])
        L += [f"s_xor_b32 s{S_WR}, s{S_WR}, {STAGE}"]
    L += ["s_waitcnt vmcnt(16)", "s_barrier"]
    for i in range(8):
        L.append(rd(A_K0, i, "%[rdA0]"))
    for j in range(8):
        L.append(rd(B_K0, j, "%[rdB0]"))
    L += ["s_waitcnt lgkmcnt(0)"]

    # ---- the loop ------------------------------------------------------------------------------------------------------
    ev = {}                      # MFMA slot (0..127) -> instructions issued right after it

    def at(slot, *ins):
        ev.setdefault(slot, []).extend(ins)

    # tile t+2 exists iff remaining > 2
    at(0, f"s_cmp_gt_u32 s{S_CNT}, 2", f"s_cselect_b32 s{SRD_A + 2}, s{S_NRA}, 0")
    at(1, f"s_cselect_b32 s{SRD_B + 2}, s{S_NRB}, 0")
    last_piece, n_before = 70, 16
    if MERGE_B12:
        # ONE barrier for both operands: all 16 k-sub-step-1 reads first, then the 16 pieces of tile t+2.  (Three barriers per
        # iteration cost the F->D shape 11 % in waves waiting for each other: tools/gemm_a4_whatif.py, whatif 128.)
        for i in range(8):
            at(2 * i, rd(A_K1, i, "%[rdA1]"))
            at(16 + 2 * i, rd(B_K1, i, "%[rdB1]"))
        at(33, f"s_mov_b32 m0, s{S_WR}")
        at(34, "s_waitcnt lgkmcnt(0)")
        at(35, "s_barrier")
        for p in range(8):
            at(36 + 2 * p, *dma(0, p))
        at(52, f"s_add_u32 m0, s{S_WR}, {B_TILE}")
        for p in range(8):
            at(54 + 2 * p, *dma(1, p))
    else:
        R1 = int(os.environ.get("A4_RD1_STRIDE", "1"))       # MFMA slots between the sub-step-1 reads of A (1 shipped: B1 at slot 11)
        b1 = 8 * R1 + 3                                       # B1: behind the last of them (19 at R1 = 2)
        for i in range(8):
            at(R1 * i, rd(A_K1, i, "%[rdA1]"))
        at(b1 - 2, f"s_mov_b32 m0, s{S_WR}")
        at(b1 - 1, "s_waitcnt lgkmcnt(0)")
        at(b1, "s_barrier")
        DS = int(os.environ.get("A4_DMA_STRIDE", "6"))        # MFMA slots between staging pieces (6 shipped: see the module docstring)
        b2 = b1 + 1 + DS * 7 + 1 + 5                           # B2: behind the last sub-step-1 read of B (47 at R1 = 2, DS = 3)
        a_slots = [b1 + 1 + DS * p for p in range(8)]
        b_slots = [b2 + 1 + DS * p for p in range(8)]
        b_slots = [x + 2 if x in (WAIT_SLOT, WAIT_SLOT + 1) else x for x in b_slots]    # not between the counted wait and its barrier
        for p in range(8):
            at(a_slots[p], dma(0, p)[0])
            at(a_slots[p] + 1, rd(B_K1, p, "%[rdB1]"))
            if p < 7:
                at(a_slots[p] + M0_LATE, dma(0, p)[1])        # the M0 step: not glued to the piece that still has to read M0
        at(b2 - 2, f"s_add_u32 m0, s{S_WR}, {B_TILE}")
        at(b2 - 1, "s_waitcnt lgkmcnt(0)")
        at(b2, "s_barrier")
        for p in range(8):
            at(b_slots[p], dma(1, p)[0])
            if p < 7:
                at(b_slots[p] + M0_LATE, dma(1, p)[1])
        last_piece = b_slots[7]
        # pieces of this iteration that are issued before the counted wait: everything older than them (= all of tile t+1, including
        # the pieces the previous iteration issued behind ITS wait) has landed once vmcnt has dropped to their number
        n_before = sum(1 for x in a_slots + b_slots if x < WAIT_SLOT)
        assert last_piece <= 119 and a_slots[7] < WAIT_SLOT, "staging must end before the loop counter's scalar compare"
    at(70, "v_xor_b32 %[rdA0], 0x10000, %[rdA0]", "v_xor_b32 %[rdA1], 0x10000, %[rdA1]")
    at(71, "v_xor_b32 %[rdB0], 0x10000, %[rdB0]", "v_xor_b32 %[rdB1], 0x10000, %[rdB1]")
    at(72, f"s_xor_b32 s{S_WR}, s{S_WR}, {STAGE}")
    # the wait for tile t+1 sits as late as the 16 fragment reads behind it allow: every slot it moves back is lead time for
    # the LDS-DMA (two LDS stages leave it ~1.3 iterations between issue and this wait)
    at(WAIT_SLOT, "s_waitcnt vmcnt(18)" if (WARM and not (whatif & 8)) else f"s_waitcnt vmcnt({n_before})")
    at(WAIT_SLOT + 1, "s_barrier")
    if B_FIRST:
        # the first 8 MFMAs of an iteration need all of B's sub-step-0 fragments but only A row 0; A row i is first used at slot
        # 8 i.  So: B and A rows 0..2 of tile t+1 behind the wait, A rows 3..7 of the CURRENT tile in slots 1..9 of the iteration
        # (the prologue has loaded them already for tile 0: the first iteration re-reads the same values).
        tail = [(B_K0, j, "%[rdB0]") for j in range(8)] + [(A_K0, i, "%[rdA0]") for i in range(3)]
        span = 125 - (WAIT_SLOT + 2)
        for n, (base, i, addr) in enumerate(tail):
            at(WAIT_SLOT + 2 + (n * span) // (len(tail) - 1), rd(base, i, addr))
        for n, i in enumerate(range(3, 8)):
            at(1 + 2 * n, rd(A_K0, i, "%[rdA0]"))
        # (the lgkmcnt(0) before barrier B1 at slot 18 covers them: their first use is slot 24)
    else:
        for i in range(8):
            at(WAIT_SLOT + 2 + 2 * i, rd(A_K0, i, "%[rdA0]"))
        for j in range(8):
            at(WAIT_SLOT + 18 + 2 * j, rd(B_K0, j, "%[rdB0]"))
    adv = max(76, (last_piece + 1) if not MERGE_B12 else 76)   # the staging position moves on only behind the iteration's last piece
    at(adv, *advance_k()[:2])
    at(adv + 1, *advance_k()[2:])
    at(adv + 4, *ADVANCE_V)
    at(124, f"s_sub_u32 s{S_CNT}, s{S_CNT}, 1")
    at(125, f"s_cmp_eq_u32 s{S_CNT}, 0")
    at(126, "s_waitcnt lgkmcnt(0)")
    L.append("1:")
    for s in range(128):
        L.append(mfma(s >> 6, s & 63))
        for ins in ev.get(s, []):
            if (whatif & 1) and (ins == "s_barrier" or ins.startswith("s_waitcnt vmcnt")):
                continue
            if (whatif & 64) and ins.startswith("s_waitcnt vmcnt"):
                continue
            if (whatif & 128) and ins == "s_barrier":
                continue
            if (whatif & 2) and ins.startswith("v_add_u32 %[voff"):
                continue
            if (whatif & 4) and ins.startswith("buffer_load_dwordx4"):
                continue
            L.append(ins)
    L += ["s_cbranch_scc0 1b"]
    # ---- drain: the last two iterations staged zero tiles; they must have landed (and every wave must be past its reads)
    # before the epilogue reuses LDS.  MFMA results need 4 passes + margin before v_accvgpr_read.
    L += ["s_waitcnt vmcnt(0)", "s_nop 7", "s_nop 7", f"s_mov_b32 m0, s{S_M0SAVE}", "s_barrier"]
    return L


def emit(name, lines):
    n_mfma = sum(1 for l in lines if l.startswith("v_mfma"))
    assert n_mfma == 128, n_mfma
    body = "\n".join(f'    "{l}\\n\\t"' for l in lines)
    vclob = ", ".join(f'"v{r}"' for r in range(V_DUMMY, 256))
    aclob = ", ".join(f'"a{r}"' for r in range(256))
    sclob = ", ".join(f'"s{r}"' for r in CLOBBER_S)
    text = f"""// GENERATED by tools/gen_gemm_a4.py — do not edit.  The K loop of gemm_a4_kernel as one asm statement.
// operands: voffA/voffB (per-lane source byte offsets, advanced by 128 per K tile), pfA/pfB (row-per-lane offsets of the L2
// warm-up loads, advanced alike), rdA0/rdA1/rdB0/rdB1 (LDS fragment read
// addresses of k-sub-steps 0/1, stage toggled by XOR 0x10000), aLo/aHi/nrA, bLo/bHi/nrB (tile row base + valid bytes),
// soA/stA, soB/stB (this wave's first row-group offset and the 8-row stride, bytes), ldsW (this wave's LDS write base in
// stage 0), nk (K tiles >= 1), k0 (first K tile of this workgroup's rotated K loop, < nk), kb (K in bytes).  Accumulators are left in a[0:255]: a[(i*8+j)*4 + r] = C[16 i + lane%16][16 j + 4 (lane/16) + r].
#define {name}(voffA, voffB, pfA, pfB, rdA0, rdA1, rdB0, rdB1, aLo, aHi, nrA, bLo, bHi, nrB, soA, stA, soB, stB, ldsW, nk, k0, kb) \\
    asm volatile( \\
{body.replace(chr(10), " " + chr(92) + chr(10))} \\
        : [voffA] "+v"(voffA), [voffB] "+v"(voffB), [pfA] "+v"(pfA), [pfB] "+v"(pfB), [rdA0] "+v"(rdA0), [rdA1] "+v"(rdA1), [rdB0] "+v"(rdB0), [rdB1] "+v"(rdB1) \\
        : [aLo] "s"(aLo), [aHi] "s"(aHi), [nrA] "s"(nrA), [bLo] "s"(bLo), [bHi] "s"(bHi), [nrB] "s"(nrB), [soA] "s"(soA), \\
          [stA] "s"(stA), [soB] "s"(soB), [stB] "s"(stB), [ldsW] "s"(ldsW), [nk] "s"(nk), [k0] "s"(k0), [kb] "s"(kb) \\
        : "memory", "scc", "vcc", {sclob}, \\
          {vclob}, \\
          {aclob})
"""
    return text


def main():
    text = emit("GF_A4_LOOP_ASM", gen())
    if os.environ.get("A4_WHATIF_VARIANTS", "0") == "1":
        # timing-only variants behind -DGF_A4_WHATIF (tools/gemm_a4_whatif.py): generated for the experimental tree only
        # (tools/experimental_tree.sh); the product's .inc holds the one loop that ships
        text += "#ifdef GF_A4_WHATIF\n" + "".join(emit(f"GF_A4_LOOP_ASM_W{w}", gen(w)) for w in (1, 2, 4, 5, 64, 128)) + "#endif\n"
    out = os.environ.get("A4_OUT", OUT)
    with open(out, "w") as f:
        f.write(text)
    print(f"wrote {out}")


if __name__ == "__main__":
    main()
