#!/usr/bin/env python3
"""Generates goal_force_amd/csrc/gf_gemm_a4f8_loop.inc: the K loop of the 4-wave FP8 GEMM (gemm_a4_kernel<EPI, true>,
gf_gemm.hip; the reference's fp8_linear contract, diffsynth/vram_management/layers.py:115-151) as ONE inline-asm statement.

    python tools/gen_gemm_a4f8.py           # rewrites the .inc (committed; the build does not run this) and runs the checker

Same frame as the bf16 loop (tools/gen_gemm_a4.py): 256 threads = 4 waves (wm, wn) in 2 x 2 on a 256 x 256 C tile, 128 x 128 of C
per wave = 8 x 8 MFMA tiles, the 256 fp32 accumulators in a[0:255], LDS = 2 stages x (A tile | B tile) of 256 rows x 128 B with
16-byte chunk c of row r at chunk c ^ (r & 7), staging by `buffer_load_dwordx4 ... offen lds` (1 KiB = 8 rows per wave
instruction; piece p of wave w = rows 32 p + 8 w .. + 7).  What differs:

  * operands are OCP e4m3 bytes: a 128-byte tile row is 128 K elements = ONE k step of v_mfma_f32_16x16x128_f8f6f4 (32 cycles per
    MFMA against 16 for 16x16x32 bf16 at four times the K: twice the bf16 rate).  A K tile is 64 MFMAs per wave, 32 ds_read_b128,
    16 staging pieces — per CYCLE the same LDS / L2 rates as the bf16 loop, per FLOP half of them.
  * a lane's 32 operand bytes are chunk fq and chunk 4 + fq of its row (fq = lane >> 4), NOT the 32 consecutive bytes 32 fq ..:
    A and W fragments follow the same rule, so every product still meets its partner, and the two ds_read_b128 of a fragment are
    exactly the bf16 kernel's conflict-free k-sub-step reads (32 consecutive bytes under the XOR swizzle put rows r and r + 8 of
    the (lanes 0-3, 12-15, 20-27) group on the same banks: a 2-way conflict on every read).
  * a whole K tile's fragments are 128 registers, so register double buffering is by HALF TILES: four sets of 32 registers
    (4 fragments x 8) XA, YA, B0, B1 = A rows lo / hi, W rows lo / hi of the wave's 128 x 128; the tile's 64 MFMAs run as four
    quadrants  q1 (Alo, Blo) 0-15, q2 (Ahi, Blo) 16-31, q3 (Ahi, Bhi) 32-47, q4 (Alo, Bhi) 48-63.  B0 is free after q2 and takes
    Blo of tile t+1 in slots 32-39; the set that held Ahi is free after q3 and takes Alo of tile t+1 in slots 48-55 — so the two A
    sets swap roles every tile and the loop body is TWO tiles (stage 0, stage 1: static LDS addresses, no toggling).
  * staging follows the read order, by halves: the LDS regions of the lo halves (A rows of pieces 0, 1, 4, 5; likewise W) of a
    stage are dead once the previous iteration has read them (slots 32-55), the hi halves after this iteration's slots 0-15.  So an
    iteration issues the 8 lo pieces of tile t+2 in its first half and the 8 hi pieces in its second half, ~3.5 MFMAs (112 cycles)
    apart, and every piece has >= 66 MFMA slots (2100 cycles) before the counted wait that needs it.
  * TWO barriers per K tile (the bf16 loop: three), each paired with a counted wait that leaves a full tile (16 pieces) in flight:
      M (slot 29): hi reads of tile t done by every wave  +  lo(t+1) landed  -> hi pieces of t+2 may overwrite, lo(t+1) readable;
      E (slot 59): lo(t+1) reads done by every wave       +  hi(t+1) landed  -> lo pieces of t+3 may overwrite, hi(t+1) readable.
  * nk odd: the body's second tile multiplies a tile staged with num_records = 0 (zeros): no peeled tail.

The generator carries a symbolic checker (check()): it replays prologue + three bodies, tags every LDS half-region and register
fragment with (tile, operand, half) and asserts that every MFMA sees the fragments of ITS tile, that every ds_read happens behind
a counted wait + barrier covering the pieces of what it reads, and that every piece is issued behind a barrier that follows the
last read of what it overwrites.
"""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "goal_force_amd", "csrc", "gf_gemm_a4f8_loop.inc")

# ---- register plan -------------------------------------------------------------------------------------------------
SET_A = (128, 160)                                    # the two A sets; fragment f of a set: v[base + 8 f : base + 8 f + 7]
SET_B = (192, 224)                                    # B0 (W rows lo), B1 (W rows hi)
V_RD = 120                                            # v[120:127]: fragment read addresses, (A c0, A c1, B c0, B c1) x stage
SRD_A, SRD_B = 60, 64                                 # s[60:63], s[64:67]
SOFF_A, SOFF_B = 36, 44                               # s[36:43], s[44:51]: row-group offsets of this wave's 8 pieces
S_M0SAVE, S_CNT, S_WR, S_NRA, S_NRB = 52, 53, 54, 56, 57
S_POS, S_STEP, S_NK, S_WRAP = 58, 59, 68, 69
STAGE = 65536
B_TILE = 32768
PIECE_STEP = 0x1000                                   # LDS bytes between a wave's consecutive pieces (GF_A4_ROWMAP = 1)
FRAG_STEP = 2048                                      # LDS bytes between the fragments of consecutive 16-row blocks
CLOBBER_S = list(range(36, 60)) + list(range(60, 70))
LO_P, HI_P = (0, 1, 4, 5), (2, 3, 6, 7)               # pieces (32-row groups of the 256-row tile) of the waves' lo / hi 64-row halves

BF16 = os.environ.get("A4F8_BF16", "0") == "1"        # the SAME schedule for bf16 operands (a K tile = 64 elements): every fp8 MFMA
                                                      # becomes the two 16x16x32 bf16 MFMAs of k-sub-steps 0 / 1 -> gf_gemm_a4h_loop.inc
SCALED = os.environ.get("A4F8_SCALED", "0") == "1"    # 1: v_mfma_scale_... with unit E8M0 scales in a VGPR (16-byte encoding)
M_SLOT = int(os.environ.get("A4F8_M_SLOT", "28"))     # counted wait of barrier M (the barrier one slot later)
E_SLOT = int(os.environ.get("A4F8_E_SLOT", "58"))
LO_SLOTS = [int(x) for x in os.environ.get("A4F8_LO_SLOTS", "2,5,9,12,16,19,23,26").split(",")]
HI_SLOTS = [int(x) for x in os.environ.get("A4F8_HI_SLOTS", "31,34,38,41,45,48,52,55").split(",")]
M0_LATE = 1                                           # MFMA slots between a piece and the M0 write for the next one


def v8(base, f):
    return f"v[{base + 8 * f}:{base + 8 * f + 7}]"


def acc(i, j):
    b = (i * 8 + j) * 4
    return f"a[{b}:{b + 3}]"


def mfma(i, j, aset, bset):
    """D = W_frag x A_frag (operands swapped as in the bf16 kernel: a lane ends up with 4 consecutive n of one m)."""
    a, b = v8(aset, i & 3), v8(bset, j & 3)
    if BF16:
        ra, rb = aset + 8 * (i & 3), bset + 8 * (j & 3)
        return [f"v_mfma_f32_16x16x32_bf16 {acc(i, j)}, v[{rb + 4 * ks}:{rb + 4 * ks + 3}], v[{ra + 4 * ks}:{ra + 4 * ks + 3}], {acc(i, j)}"
                for ks in range(2)]
    if SCALED:
        return f"v_mfma_scale_f32_16x16x128_f8f6f4 {acc(i, j)}, {b}, {a}, {acc(i, j)}, v{V_RD - 1}, v{V_RD - 1} op_sel_hi:[0,0,0]"
    return f"v_mfma_f32_16x16x128_f8f6f4 {acc(i, j)}, {b}, {a}, {acc(i, j)}"


def rd_addr(op, chunk, stage):
    return f"v{V_RD + stage * 4 + op * 2 + chunk}"


def rd(setbase, f, op, half, chunk, stage):
    """One ds_read_b128: registers 4 chunk .. 4 chunk + 3 of fragment f of a set <- chunk (4 chunk + fq) of 16-row block 4 half + f."""
    r0 = setbase + 8 * f + 4 * chunk
    off = (4 * half + f) * FRAG_STEP
    return f"ds_read_b128 v[{r0}:{r0 + 3}], {rd_addr(op, chunk, stage)}" + (f" offset:{off}" if off else "")


def advance_k():
    return [f"s_add_u32 s{S_POS}, s{S_POS}, 1", f"s_cmp_eq_u32 s{S_POS}, s{S_NK}",
            f"s_cselect_b32 s{S_STEP}, s{S_WRAP}, 128", f"s_cselect_b32 s{S_POS}, 0, s{S_POS}"]


ADVANCE_V = [f"v_add_u32 %[voffA], s{S_STEP}, %[voffA]", f"v_add_u32 %[voffB], s{S_STEP}, %[voffB]"]


def piece(op, p):
    srd, soff, voff = (SRD_A, SOFF_A, "%[voffA]") if op == 0 else (SRD_B, SOFF_B, "%[voffB]")
    return f"buffer_load_dwordx4 {voff}, s[{srd}:{srd + 3}], s{soff + p} offen lds"


def m0_for(op, p, stage):
    return f"s_add_u32 m0, s{S_WR}, {stage * STAGE + op * B_TILE + p * PIECE_STEP:#x}"


def half_pieces(half, first_op):
    """The 8 pieces of one half of a tile in issue order: the operand that is read first is staged first."""
    ps = LO_P if half == 0 else HI_P
    return [(first_op, p) for p in ps] + [(1 - first_op, p) for p in ps]


def tile_events(par, whatif=0):
    """Events of one K tile of parity `par` (= its LDS stage): slot -> instructions issued right after that slot's MFMA."""
    ev = {}

    def at(slot, *ins):
        ev.setdefault(slot, []).extend(ins)

    st, nst = par, 1 - par
    XA, YA = SET_A[par], SET_A[1 - par]               # XA holds Alo(t) at tile entry; YA takes Ahi(t), then Alo(t+1)
    B0, B1 = SET_B
    # tile t+2 exists iff more than two tiles remain (s_cnt = tiles remaining at the start of the body; its second tile: one fewer)
    at(0, f"s_cmp_gt_u32 s{S_CNT}, {2 + par}", f"s_cselect_b32 s{SRD_A + 2}, s{S_NRA}, 0", f"s_cselect_b32 s{SRD_B + 2}, s{S_NRB}, 0")
    # hi halves of tile t: Ahi -> YA (slots 0-7, needed by q2 at 16), Bhi -> B1 (slots 8-15, needed by q3 at 32)
    n = 0
    for f in range(4):
        for c in range(2):
            at(n, rd(YA, f, 0, 1, c, st))
            at(8 + n, rd(B1, f, 1, 1, c, st))
            n += 1
    at(15, "s_waitcnt lgkmcnt(8)")
    # lo pieces of tile t+2 -> stage st (its lo regions were read in the previous iteration's slots 32-55, barrier E since)
    lo = half_pieces(0, 1)                            # W first: Blo is what the next iteration reads first (slot 32)
    at(0, m0_for(*lo[0], st))
    for k, (op, p) in enumerate(lo):
        at(LO_SLOTS[k], piece(op, p))
        if k + 1 < 8:
            at(LO_SLOTS[k] + M0_LATE, m0_for(*lo[k + 1], st))
    assert LO_SLOTS[7] < M_SLOT and LO_SLOTS[0] >= 1
    # barrier M: every wave has read the hi halves of tile t; lo(t+1) landed (16 younger pieces may stay in flight: hi(t+1), lo(t+2))
    at(M_SLOT, "s_waitcnt vmcnt(16)")
    at(M_SLOT + 1, "s_waitcnt lgkmcnt(0)", "s_barrier")
    hi = half_pieces(1, 0)                            # A first: Ahi is read first (slot 0 of the iteration after next)
    at(M_SLOT + 1, m0_for(*hi[0], st))
    for k, (op, p) in enumerate(hi):
        at(HI_SLOTS[k], piece(op, p))
        if k + 1 < 8:
            at(HI_SLOTS[k] + M0_LATE, m0_for(*hi[k + 1], st))
    assert HI_SLOTS[0] > M_SLOT + 1 and HI_SLOTS[7] < E_SLOT
    # lo halves of tile t+1 from the other stage: Blo -> B0 (free after q2), Alo -> YA (free after q3)
    n = 0
    for f in range(4):
        for c in range(2):
            at(32 + n, rd(B0, f, 1, 0, c, nst))
            at(48 + n, rd(YA, f, 0, 0, c, nst))
            n += 1
    # barrier E: every wave has read lo(t+1); hi(t+1) landed (16 younger pieces in flight: lo(t+2), hi(t+2))
    at(E_SLOT, "s_waitcnt vmcnt(16)")
    at(E_SLOT + 1, "s_waitcnt lgkmcnt(0)", "s_barrier")
    # the staging position moves one K tile on: behind this tile's last piece, before the next tile's first
    at(E_SLOT + 2, *advance_k()[:2])
    at(E_SLOT + 3, *advance_k()[2:])
    at(E_SLOT + 4, *ADVANCE_V)
    assert E_SLOT + 4 <= 63 and HI_SLOTS[7] < E_SLOT + 2
    if par == 1:
        at(61, f"s_sub_u32 s{S_CNT}, s{S_CNT}, 2")
        at(63, f"s_cmp_gt_i32 s{S_CNT}, 0")
    # the 64 MFMAs: quadrants (Alo, Blo), (Ahi, Blo), (Ahi, Bhi), (Alo, Bhi); A block outer, W block inner
    order = []
    for (ah, bh) in ((0, 0), (1, 0), (1, 1), (0, 1)):
        for i in range(4):
            for j in range(4):
                order.append((4 * ah + i, 4 * bh + j, XA if ah == 0 else YA, B0 if bh == 0 else B1))
    out = []
    if BF16:
        # per A block: k-sub-step 0 of its four accumulators, then k-sub-step 1 (an accumulator's two MFMAs four apart); a "slot" of
        # the schedule = two MFMAs = the 32 cycles of one fp8 MFMA
        flat = []
        for g in range(0, 64, 4):
            pairs = [mfma(*order[g + q]) for q in range(4)]
            flat += [pr[0] for pr in pairs] + [pr[1] for pr in pairs]
        slots = [flat[2 * s: 2 * s + 2] for s in range(64)]
    else:
        slots = [[mfma(*order[s])] for s in range(64)]
    for s in range(64):
        out += slots[s]
        for ins in ev.get(s, []):
            if (whatif & 1) and (ins == "s_barrier" or ins.startswith("s_waitcnt vmcnt")):
                continue
            if (whatif & 4) and ins.startswith("buffer_load_dwordx4"):
                continue
            out.append(ins)
    return out


def prologue():
    L = [f"s_mov_b32 s{S_M0SAVE}, m0"]
    L += [f"s_mov_b32 s{SRD_A}, %[aLo]", f"s_mov_b32 s{SRD_A + 1}, %[aHi]", f"s_mov_b32 s{SRD_A + 2}, %[nrA]",
          f"s_mov_b32 s{SRD_A + 3}, 0x00020000",
          f"s_mov_b32 s{SRD_B}, %[bLo]", f"s_mov_b32 s{SRD_B + 1}, %[bHi]", f"s_mov_b32 s{SRD_B + 2}, %[nrB]",
          f"s_mov_b32 s{SRD_B + 3}, 0x00020000",
          f"s_mov_b32 s{S_NRA}, %[nrA]", f"s_mov_b32 s{S_NRB}, %[nrB]", f"s_mov_b32 s{S_CNT}, %[nk]",
          f"s_mov_b32 s{S_WR}, %[ldsW]",
          # staggered start: this tile's K loop begins at K tile k0 and wraps (the sum over k is rotated, not changed)
          f"s_mov_b32 s{S_NK}, %[nk]", f"s_mov_b32 s{S_POS}, %[k0]", f"s_sub_u32 s{S_WRAP}, 128, %[kb]",
          f"s_lshl_b32 s{S_STEP}, %[k0], 7", "s_nop 0",
          f"v_add_u32 %[voffA], s{S_STEP}, %[voffA]", f"v_add_u32 %[voffB], s{S_STEP}, %[voffB]"]
    L += [f"s_mov_b32 s{SOFF_A}, %[soA]", f"s_mov_b32 s{SOFF_B}, %[soB]"]
    for p in range(1, 8):
        L += [f"s_add_u32 s{SOFF_A + p}, s{SOFF_A + p - 1}, %[stA]", f"s_add_u32 s{SOFF_B + p}, s{SOFF_B + p - 1}, %[stB]"]
    # fragment read addresses of both stages: (A c0, A c1, B c0, B c1)
    for stg in range(2):
        for k, name in enumerate(("rdA0", "rdA1", "rdB0", "rdB1")):
            L.append(f"v_add_u32 v{V_RD + stg * 4 + k}, {stg * STAGE:#x}, %[{name}]" if stg else f"v_mov_b32 v{V_RD + k}, %[{name}]")
    if SCALED:
        L.append(f"v_mov_b32 v{V_RD - 1}, 0x7f7f7f7f")
    # tile 0 -> stage 0 back to back (its latency is the kernel's start-up), in the loop's order: lo halves, then hi halves
    for (op, p) in half_pieces(0, 1) + half_pieces(1, 0):
        L += [m0_for(op, p, 0), "s_nop 0", piece(op, p)]
    L += advance_k() + ADVANCE_V
    L += [f"s_cmp_gt_u32 s{S_CNT}, 1", f"s_cselect_b32 s{SRD_A + 2}, s{S_NRA}, 0", f"s_cselect_b32 s{SRD_B + 2}, s{S_NRB}, 0", "s_nop 1"]
    # tile 1 -> stage 1 (zeros past K), one piece per 16 accumulator registers zeroed (under the latency of tile 0)
    for q, (op, p) in enumerate(half_pieces(0, 1) + half_pieces(1, 0)):
        L += [m0_for(op, p, 1), "s_nop 0", piece(op, p)]
        L += [f"v_accvgpr_write_b32 a{16 * q + r}, 0" for r in range(16)]
    L += advance_k() + ADVANCE_V
    L += ["s_waitcnt vmcnt(16)", "s_barrier"]
    n = 0
    for f in range(4):
        for c in range(2):
            L.append(rd(SET_A[0], f, 0, 0, c, 0))
    for f in range(4):
        for c in range(2):
            L.append(rd(SET_B[0], f, 1, 0, c, 0))
    # every wave has read its lo halves of tile 0 before the body's first piece (lo of tile 2) overwrites them
    L += ["s_waitcnt lgkmcnt(0)", "s_barrier"]
    return L


def gen(whatif=0):
    L = prologue()
    L.append("1:")
    L += tile_events(0, whatif)
    L += tile_events(1, whatif)
    L += ["s_cbranch_scc1 1b"]
    # drain: the last iterations staged zero tiles; they must have landed (and every wave must be past its reads) before the
    # epilogue reuses LDS.  The 8-pass MFMA's results need their wait states before v_accvgpr_read.
    L += ["s_waitcnt vmcnt(0)", "s_nop 7", "s_nop 7", "s_nop 7", f"s_mov_b32 m0, s{S_M0SAVE}", "s_barrier"]
    return L


# ---- symbolic checker ----------------------------------------------------------------------------------------------------
def check(lines, nbodies=3):
    """Replays prologue + nbodies bodies of ONE wave (all waves run the same stream, so 'every wave has done X' == 'a barrier
    follows X in this stream').  State: LDS half-regions (stage, op, half) -> (tile, landed-and-visible?), register fragments ->
    (tile, op, block), outstanding pieces in issue order."""
    i1 = lines.index("1:")
    pro, body = lines[:i1], [l for l in lines[i1 + 1:] if not l.startswith("s_cbranch")]
    stream = pro + body * nbodies
    region = {}          # (stage, op, half) -> dict(tile=, visible=bool, pending_reads=bool(read since last barrier))
    frag = {}            # first register of a 4-register half fragment -> (tile, op, block, chunk)
    outstanding = []     # pieces in issue order: (stage, op, half, tile)
    landed_unbarriered = []   # pieces this wave knows landed, not yet published by a barrier
    reads_unbarriered = set() # regions read since the last (lgkmcnt(0) + barrier)
    reads_unwaited = []       # (regs first, n-th read) issued, not yet covered by an lgkmcnt wait
    issued_per_region = {}
    m0 = None
    tile_of_stage = {0: 0, 1: 1}      # tile currently being staged INTO each stage: advanced when its 16th piece is issued
    pieces_in_tile = {0: 0, 1: 0}
    mf_count = 0
    ks_seen = {}
    lgkm_clean = True
    scc_from_cmp = False
    for ins in stream + ["s_cbranch_scc1 1b"]:
        # SCC: every s_cselect / s_cbranch must consume the flag of an s_cmp, not of an s_add / s_sub / s_lshl in between
        if ins.startswith(("s_cselect", "s_cbranch_scc")):
            assert scc_from_cmp, f"{ins}: SCC was overwritten since the compare"
        elif ins.startswith("s_cmp"):
            scc_from_cmp = True
        elif ins.startswith(("s_add_u32", "s_sub_u32", "s_lshl_b32", "s_xor_b32", "s_and_b32", "s_or_b32")):
            scc_from_cmp = False
        m = re.match(r"s_add_u32 m0, s\d+, (0x[0-9a-f]+)", ins)
        if m:
            m0 = int(m.group(1), 16)
            continue
        if ins.startswith("buffer_load_dwordx4"):
            stage, rem = divmod(m0, STAGE)
            op, rem = divmod(rem, B_TILE)
            p = rem // PIECE_STEP
            assert (ins.find("voffA") >= 0) == (op == 0), (ins, m0)
            assert int(re.search(r"s(\d+) offen", ins).group(1)) == (SOFF_A if op == 0 else SOFF_B) + p, (ins, m0)
            half = 0 if p in LO_P else 1
            key = (stage, op, half)
            assert key not in reads_unbarriered, f"piece overwrites {key} before a barrier closed its reads"
            t = tile_of_stage[stage]
            old = region.get(key)
            if old is not None and old["tile"] != t:
                assert old["read_done"], f"piece of tile {t} overwrites {key} holding unread tile {old['tile']}"
            if old is None or old["tile"] != t:
                region[key] = dict(tile=t, visible=False, count=0, read_done=False)
            outstanding.append(key)
            pieces_in_tile[stage] += 1
            if pieces_in_tile[stage] == 16:
                pieces_in_tile[stage] = 0
                tile_of_stage[stage] += 2
            m0 = None   # every piece needs its own M0
            continue
        m = re.match(r"s_waitcnt vmcnt\((\d+)\)", ins)
        if m:
            keep = int(m.group(1))
            while len(outstanding) > keep:
                landed_unbarriered.append(outstanding.pop(0))
            continue
        m = re.match(r"s_waitcnt lgkmcnt\((\d+)\)", ins)
        if m:
            keep = int(m.group(1))
            while len(reads_unwaited) > keep:
                r0, tag = reads_unwaited.pop(0)
                frag[r0] = tag
            lgkm_clean = keep == 0
            continue
        if ins == "s_barrier":
            assert not reads_unwaited, "barrier with LDS reads in flight (the region could be overwritten under them)"
            for key in landed_unbarriered:
                region[key]["count"] += 1
                if region[key]["count"] == 4:      # a half region = 4 pieces per wave (2 row groups x ... ) -> see below
                    region[key]["visible"] = True
            landed_unbarriered = []
            reads_unbarriered = set()
            continue
        m = re.match(r"ds_read_b128 v\[(\d+):\d+\], v(\d+)(?: offset:(\d+))?", ins)
        if m:
            r0, va, off = int(m.group(1)), int(m.group(2)), int(m.group(3) or 0)
            stage, k = divmod(va - V_RD, 4)
            op, chunk = divmod(k, 2)
            block = off // FRAG_STEP
            half = block // 4
            key = (stage, op, half)
            assert region[key]["visible"], f"read of {key} (tile {region[key]['tile']}) before its pieces are landed + barriered"
            assert (r0 - 128) % 8 == 4 * chunk, ins
            for (rr, _) in reads_unwaited:
                assert rr != r0
            reads_unwaited.append((r0, (region[key]["tile"], op, block, chunk)))
            frag.pop(r0, None)                      # the old contents are gone once the read is issued
            reads_unbarriered.add(key)
            region[key]["reads"] = region[key].get("reads", 0) + 1
            if region[key]["reads"] == 8:
                region[key]["read_done"] = True
            continue
        m = re.match(r"v_mfma\S* a\[(\d+):\d+\], v\[(\d+):(\d+)\], v\[(\d+):\d+\], a\[(\d+):", ins)
        if m:
            a0, vb, vbe, va, c0 = (int(x) for x in m.groups())
            assert a0 == c0
            i, j = divmod(a0 // 4, 8)
            per_tile = 128 if BF16 else 64
            tile = mf_count // per_tile
            mf_count += 1
            chunks = (0, 1) if vbe - vb == 7 else ((vb % 8) // 4,)
            if BF16:      # an accumulator takes k-sub-step 0 of a tile before k-sub-step 1 (the bf16 kernels' summation order)
                assert ks_seen.get((a0, tile), -1) == chunks[0] - 1, f"acc({i},{j}) tile {tile}: k-sub-steps out of order"
                ks_seen[(a0, tile)] = chunks[0]
            for (base, op, blk) in ((va - va % 8, 0, i), (vb - vb % 8, 1, j)):
                for chunk in chunks:
                    got = frag.get(base + 4 * chunk)
                    assert got == (tile, op, blk, chunk), f"MFMA #{mf_count - 1} acc({i},{j}) of tile {tile}: v{base + 4 * chunk} holds {got}"
            continue
    assert mf_count == (128 if BF16 else 64) * 2 * nbodies
    return True


def emit(name, lines):
    n_mfma = sum(1 for l in lines if l.startswith("v_mfma"))
    assert n_mfma == (256 if BF16 else 128), n_mfma
    body = "\n".join(f'    "{l}\\n\\t"' for l in lines)
    vclob = ", ".join(f'"v{r}"' for r in range(V_RD - 1, 256))
    aclob = ", ".join(f'"a{r}"' for r in range(256))
    sclob = ", ".join(f'"s{r}"' for r in CLOBBER_S)
    text = f"""// GENERATED by tools/gen_gemm_a4f8.py — do not edit.  The K loop of gemm_a4_kernel<EPI, FP8 = true> as one asm statement.
// operands: voffA/voffB (per-lane source byte offsets, advanced by 128 per K tile), rdA0/rdA1/rdB0/rdB1 (LDS fragment read
// addresses of chunk fq / chunk 4 + fq in stage 0), aLo/aHi/nrA, bLo/bHi/nrB (tile row base + valid bytes), soA/stA, soB/stB
// (this wave's first row-group offset and the 32-row stride, bytes), ldsW (this wave's LDS write base in stage 0), nk (K tiles of
// 128 bytes, >= 1), k0 (first K tile of this workgroup's rotated K loop, < nk), kb (K in bytes).
// Accumulators are left in a[0:255]: a[(i*8+j)*4 + r] = C[16 i + lane%16][16 j + 4 (lane/16) + r].
#define {name}(voffA, voffB, rdA0, rdA1, rdB0, rdB1, aLo, aHi, nrA, bLo, bHi, nrB, soA, stA, soB, stB, ldsW, nk, k0, kb) \\
    asm volatile( \\
{body.replace(chr(10), " " + chr(92) + chr(10))} \\
        : [voffA] "+v"(voffA), [voffB] "+v"(voffB) \\
        : [rdA0] "v"(rdA0), [rdA1] "v"(rdA1), [rdB0] "v"(rdB0), [rdB1] "v"(rdB1), \\
          [aLo] "s"(aLo), [aHi] "s"(aHi), [nrA] "s"(nrA), [bLo] "s"(bLo), [bHi] "s"(bHi), [nrB] "s"(nrB), [soA] "s"(soA), \\
          [stA] "s"(stA), [soB] "s"(soB), [stB] "s"(stB), [ldsW] "s"(ldsW), [nk] "s"(nk), [k0] "s"(k0), [kb] "s"(kb) \\
        : "memory", "scc", "vcc", {sclob}, \\
          {vclob}, \\
          {aclob})
"""
    return text


def main():
    lines = gen()
    check(lines)
    if BF16:
        text = emit("GF_A4H_LOOP_ASM", lines).replace("gemm_a4_kernel<EPI, FP8 = true>", "gemm_a4_kernel<EPI> (bf16, half-tile schedule)")
    else:
        text = emit("GF_A4F8_LOOP_ASM", lines)
        # timing-only variants behind -DGF_A4_WHATIF (never in the shipped library): 1 = no barriers / counted waits, 4 = no staging
        text += "#ifdef GF_A4_WHATIF\n" + "".join(emit(f"GF_A4F8_LOOP_ASM_W{w}", gen(w)) for w in (1, 4)) + "#endif\n"
    out = os.environ.get("A4F8_OUT", OUT.replace("a4f8", "a4h") if BF16 else OUT)
    with open(out, "w") as f:
        f.write(text)
    print(f"wrote {out} (schedule checked)")


if __name__ == "__main__":
    main()
