#!/usr/bin/env python3
"""TIMING-ONLY probe (round 4): the steady phase of a flash-attention forward with 64 query rows per wave at ONE wave per SIMD,
as a hand-scheduled asm loop — the variant round 2 built compiler-scheduled (18.7 ms on zeros) and the round-3 verdict asked to
try "through the generated-asm route".  Results are WRONG by construction (no rescale path, no first / last tile handling, no
normalisation); what it measures is whether a lone wave can issue this instruction stream — real LDS-DMA staging of K / V^T tiles,
real fragment reads, the softmax's exp2 / pack / max instructions on real scores, one barrier per tile — fast enough to beat
the shipped two-waves-per-SIMD kernel (tools/probes/attn_nq4_whatif.hip, tools/attn_nq4_probe.sh).

    python3 tools/gen_attn_nq4.py > build/attn_nq4_loop.inc

Per 64-key tile and wave: 64 PV + 8 row-sum + 64 QK^T MFMAs (v_mfma_f32_16x16x32_bf16), 32 ds_read_b128 (each fragment feeds the
wave's four 16-query blocks), 8 LDS-DMA pieces, 64 v_exp_f32, 32 v_cvt_pk_bf16_f32, 32 v_max3_f32: ONE non-MFMA vector
instruction per MFMA slot.  Registers (hand-allocated):
  a[0:127]   O^T accumulators [qb][db]        a[128:143] row sums [qb]        a[144:207] Q fragments [qb][ks] (MFMA B operand)
  v[16:79], v[80:143] S(tile parity 0 / 1) [qb][kb]      v[144:175], v[176:207] P(parity) [kk][qb] packed bf16 pairs
  v[208:223] fragment ring (4)   v[224:231] exp2 results (2 pairs)   v[232:247] -max splats [qb]   v[248:251] ones fragment
  v[252:255] running-max watch [qb];  v0..v15 are the compiler's (the operands below).
"""
import os
import sys

OUT = []
NO_VALU, NO_DMA, NO_READS = (os.environ.get(k, "0") == "1" for k in ("NQ4_NO_VALU", "NQ4_NO_DMA", "NQ4_NO_READS"))   # what-if builds
# NQ4_STAGE=reg: K / V^T tiles staged through REGISTERS instead of LDS-DMA: 8 plain buffer_load_dwordx4 into spare AGPRs a[208:239]
# early in the phase, 8 ds_write_b128 from them late in the phase (a lone wave stalls ~100 cycles on every LDS-DMA piece: 27 % of
# the loop; profiles/r04/attn_nq4_probe.log).  Slots of the loads / of the LDS writes: NQ4_LD0, NQ4_LDS (first, spacing), NQ4_WR0, NQ4_WRS.
STAGE_REG = os.environ.get("NQ4_STAGE", "dma") == "reg"
LD0, LDSP, WR0, WRS = (int(os.environ.get(k, d)) for k, d in (("NQ4_LD0", "2"), ("NQ4_LDS", "4"), ("NQ4_WR0", "100"), ("NQ4_WRS", "4")))


def emit(s):
    OUT.append(s)


def O(qb, db):
    b = (qb * 8 + db) * 4
    return f"a[{b}:{b + 3}]"


def RS(qb):
    b = 128 + qb * 4
    return f"a[{b}:{b + 3}]"


def Q(qb, ks):
    b = 144 + (qb * 4 + ks) * 4
    return f"a[{b}:{b + 3}]"


def S(buf, qb, kb, j=None):
    b = 16 + buf * 64 + (qb * 4 + kb) * 4
    return f"v[{b}:{b + 3}]" if j is None else f"v{b + j}"


def P(buf, kk, qb, w=None):
    b = 144 + buf * 32 + (kk * 4 + qb) * 4
    return f"v[{b}:{b + 3}]" if w is None else f"v{b + w}"


def RING(f):
    b = 208 + (f % 4) * 4
    return f"v[{b}:{b + 3}]"


def NEGM(qb):
    b = 232 + qb * 4
    return f"v[{b}:{b + 3}]"


ONES = "v[248:251]"


def MX(qb):
    return f"v{252 + qb}"


def frag_read(f, par):
    rd = 1 - par                                   # this phase reads the buffers the previous phase's DMA filled
    if f < 16:
        kk, db = f >> 3, f & 7
        return f"ds_read_b128 {RING(f)}, %[voff{kk}] offset:{db * 2048 + rd * 16384}"
    ks, kb = (f - 16) >> 2, (f - 16) & 3
    return f"ds_read_b128 {RING(f)}, %[koff{ks}] offset:{kb * 4096 + rd * 16384}"


def first_use_slot(f):
    return 4 * f if f < 16 else 72 + 4 * (f - 16)


def phase(par):
    """One tile phase; par = parity of tile p: S(p) in S[par], P(p) -> P[par]; PV(p-1) reads P[1-par]; QK(p+1) -> S[1-par]."""
    # VALU program: 32 pairs x (exp a, exp b, max3, cvt)
    valu = []
    for i in range(32):
        e0, e1 = 2 * i, 2 * i + 1
        qb, kb, j0 = e0 // 16, (e0 // 4) % 4, e0 % 4
        t0, t1 = 224 + (i % 2) * 2, 225 + (i % 2) * 2
        w = (kb & 1) * 2 + (j0 >> 1)
        valu.append(f"v_exp_f32 v{t0}, {S(par, qb, kb, j0)}")
        valu.append(f"v_exp_f32 v{t1}, {S(par, qb, kb, j0 + 1)}")
        valu.append(f"v_max3_f32 {MX(qb)}, {MX(qb)}, {S(par, qb, kb, j0)}, {S(par, qb, kb, j0 + 1)}")
        valu.append(f"v_cvt_pk_bf16_f32 {P(par, kb >> 1, qb, w)}, v{t0}, v{t1}")
    # reads: fragment f is requested 8 slots before its first use (frags 0..2 at the top of the phase)
    reads = {}
    for f in range(32):
        s = max(first_use_slot(f) - 8, -1)
        reads.setdefault(s, []).append(f)
    issued = []                                    # fragments requested so far, in order (LDS returns in order)
    for f in reads.get(-1, []):
        if not NO_READS:
            emit(frag_read(f, par))
        issued.append(f)
    dma = {10: ("v", 0), 26: ("v", 1), 42: ("v", 2), 58: ("v", 3), 74: ("k", 0), 90: ("k", 1), 106: ("k", 2), 122: ("k", 3)}
    pieces = [("v", 0), ("v", 1), ("v", 2), ("v", 3), ("k", 0), ("k", 1), ("k", 2), ("k", 3)]
    loads = {LD0 + LDSP * i: i for i in range(8)} if STAGE_REG else {}
    writes = {WR0 + WRS * i: i for i in range(8)} if STAGE_REG else {}
    if STAGE_REG:
        dma = {}
    for s in range(136):
        for f in reads.get(s, []):
            if f not in issued:
                if not NO_READS:
                    emit(frag_read(f, par))
                issued.append(f)
        # the MFMA of this slot
        if s < 64:
            f, qb = s // 4, s % 4
            kk, db = f >> 3, f & 7
            if s % 4 == 0 and not NO_READS:
                emit(f"s_waitcnt lgkmcnt({len(issued) - 1 - issued.index(f)})")
            emit(f"v_mfma_f32_16x16x32_bf16 {O(qb, db)}, {RING(f)}, {P(1 - par, kk, qb)}, {O(qb, db)}")
        elif s < 72:
            kk, qb = (s - 64) // 4, s % 4
            emit(f"v_mfma_f32_16x16x32_bf16 {RS(qb)}, {ONES}, {P(1 - par, kk, qb)}, {RS(qb)}")
        else:
            i = s - 72
            f, qb = 16 + i // 4, i % 4
            ks, kb = (f - 16) >> 2, (f - 16) & 3
            if i % 4 == 0 and not NO_READS:
                emit(f"s_waitcnt lgkmcnt({len(issued) - 1 - issued.index(f)})")
            c = NEGM(qb) if ks == 0 else S(1 - par, qb, kb)
            emit(f"v_mfma_f32_16x16x32_bf16 {S(1 - par, qb, kb)}, {RING(f)}, {Q(qb, ks)}, {c}")
        if s < len(valu) and not NO_VALU:
            emit(valu[s])
        if s in dma and not NO_DMA:
            which, jj = dma[s]
            if which == "v":       # V^T tile p -> V buffer par
                emit(f"s_add_u32 s76, s71, s{80 + jj}")
                emit(f"s_add_u32 m0, s54, 0x{32768 + par * 16384 + jj * 4096:x}")
                emit("s_nop 0")
                emit("buffer_load_dwordx4 %[vtoff], s[64:67], s76 offen lds")
            else:                  # K tile p + 2 -> K buffer par
                emit(f"s_add_u32 s76, s70, s{72 + jj}")
                emit(f"s_add_u32 m0, s54, 0x{par * 16384 + jj * 4096:x}")
                emit("s_nop 0")
                emit("buffer_load_dwordx4 %[ksoff], s[60:63], s76 offen lds")
        if s in loads and not NO_DMA:
            i = loads[s]
            which, jj = pieces[i]
            if which == "v":
                emit(f"s_add_u32 s76, s71, s{80 + jj}")
                emit(f"buffer_load_dwordx4 a[{208 + 4 * i}:{211 + 4 * i}], %[vtoff], s[64:67], s76 offen")
            else:
                emit(f"s_add_u32 s76, s70, s{72 + jj}")
                emit(f"buffer_load_dwordx4 a[{208 + 4 * i}:{211 + 4 * i}], %[ksoff], s[60:63], s76 offen")
        if s in writes and not NO_DMA:
            i = writes[s]
            which, jj = pieces[i]
            emit(f"s_waitcnt vmcnt({7 - i})")          # loads return in order: piece i has landed when 7 - i younger ones are still out
            emit(f"ds_write_b128 %[wdst], a[{208 + 4 * i}:{211 + 4 * i}] offset:{(32768 if which == 'v' else 0) + par * 16384 + jj * 4096}")
            issued.append(("w", i))                    # LDS writes count in lgkmcnt with the reads, in order
        if s == 130:
            emit("s_add_u32 s70, s70, %[kstep]")      # next tile's K / V^T source offsets
            emit("s_add_u32 s71, s71, 128")
    emit("s_waitcnt vmcnt(0) lgkmcnt(0)" if STAGE_REG else "s_waitcnt vmcnt(0)")
    emit("s_barrier")


def main():
    emit("s_mov_b32 s52, m0")
    # descriptors: K rows [kv_len, k_stride] (num_records = bytes that exist: tiles past the end read zeros), V^T [heads*128, kv_pad]
    for i, n in enumerate(("kLo", "kHi", "kNr")):
        emit(f"s_mov_b32 s{60 + i}, %[{n}]")
    emit("s_mov_b32 s63, 0x00020000")
    for i, n in enumerate(("vLo", "vHi", "vNr")):
        emit(f"s_mov_b32 s{64 + i}, %[{n}]")
    emit("s_mov_b32 s67, 0x00020000")
    emit("s_mov_b32 s54, %[ldsW]")
    emit("s_mov_b32 s53, %[npairs]")
    emit("s_mov_b32 s72, 0")
    emit("s_mov_b32 s80, 0")
    for jj in range(1, 4):
        emit(f"s_add_u32 s{72 + jj}, s{71 + jj}, %[kpiece]")
        emit(f"s_add_u32 s{80 + jj}, s{79 + jj}, %[vpiece]")
    # Q fragments: 16 x global_load_dwordx4 into v[16:79], then into a[144:207]
    for qb in range(4):
        for ks in range(4):
            b = 16 + (qb * 4 + ks) * 4
            emit(f"global_load_dwordx4 v[{b}:{b + 3}], %[qoff], %[qptr] offset:{ks * 64}")
        if qb < 3:
            emit("v_add_u32 %[qoff], %[qstep], %[qoff]")
    # zero O, row sums; -max splats 0; ones fragment (bf16 1.0 in row 0 only is the kernel's; here every lane: timing only)
    for i in range(144):
        emit(f"v_accvgpr_write_b32 a{i}, 0")
    for i in range(232, 248):
        emit(f"v_mov_b32 v{i}, 0")
    for i in range(248, 252):
        emit(f"v_mov_b32 v{i}, 0x3f803f80")
    for i in range(252, 256):
        emit(f"v_mov_b32 v{i}, 0")
    for i in range(80, 208):
        emit(f"v_mov_b32 v{i}, 0")
    emit("s_waitcnt vmcnt(0)")
    for i in range(64):
        emit(f"v_accvgpr_write_b32 a{144 + i}, v{16 + i}")
    for i in range(16, 80):
        emit(f"v_mov_b32 v{i}, 0")
    # prologue staging: K tile 1 -> K buffer 1, V^T tile 0 -> V buffer 1 (what phase 0 reads); phase p stages V^T(p), K(p+2)
    emit("s_mov_b32 s70, %[kstep]")
    emit("s_mov_b32 s71, 0")
    for jj in range(4):
        emit(f"s_add_u32 s76, s70, s{72 + jj}")
        emit(f"s_add_u32 m0, s54, 0x{16384 + jj * 4096:x}")
        emit("s_nop 0")
        emit("buffer_load_dwordx4 %[ksoff], s[60:63], s76 offen lds")
        emit(f"s_add_u32 s76, s71, s{80 + jj}")
        emit(f"s_add_u32 m0, s54, 0x{32768 + 16384 + jj * 4096:x}")
        emit("s_nop 0")
        emit("buffer_load_dwordx4 %[vtoff], s[64:67], s76 offen lds")
    emit("s_add_u32 s70, s70, %[kstep]")               # K tile 2 comes next
    emit("s_waitcnt vmcnt(0)")
    emit("s_barrier")
    emit("1:")
    phase(0)
    phase(1)
    emit("s_sub_u32 s53, s53, 1")
    emit("s_cmp_eq_u32 s53, 0")
    emit("s_cbranch_scc0 1b")
    emit("s_nop 7")
    emit("s_nop 7")
    emit("s_mov_b32 m0, s52")
    body = " \\\n".join(f'    "{l}\\n\\t"' for l in OUT)
    clob = ['"memory"', '"scc"', '"vcc"'] + [f'"s{i}"' for i in list(range(52, 55)) + list(range(60, 68)) + list(range(70, 77)) + list(range(80, 84))]
    clob += [f'"v{i}"' for i in range(16, 256)] + [f'"a{i}"' for i in range(256)]
    print("// GENERATED by tools/gen_attn_nq4.py — do not edit.  TIMING-ONLY steady phase of a 64-query-rows-per-wave attention forward.")
    print("#define GF_NQ4_LOOP_ASM(koff0, koff1, koff2, koff3, voff0, voff1, ksoff, vtoff, qoff, qstep, qptr, kLo, kHi, kNr, vLo, vHi, vNr, "
          "ldsW, npairs, kstep, kpiece, vpiece, wdst) \\")
    print("    asm volatile( \\")
    print(body + " \\")
    print('        : [qoff] "+v"(qoff) \\')
    print('        : [koff0] "v"(koff0), [koff1] "v"(koff1), [koff2] "v"(koff2), [koff3] "v"(koff3), [voff0] "v"(voff0), [voff1] "v"(voff1), '
          '[ksoff] "v"(ksoff), [vtoff] "v"(vtoff), [wdst] "v"(wdst), [qstep] "s"(qstep), [qptr] "s"(qptr), [kLo] "s"(kLo), [kHi] "s"(kHi), [kNr] "s"(kNr), '
          '[vLo] "s"(vLo), [vHi] "s"(vHi), [vNr] "s"(vNr), [ldsW] "s"(ldsW), [npairs] "s"(npairs), [kstep] "s"(kstep), [kpiece] "s"(kpiece), '
          '[vpiece] "s"(vpiece) \\')
    print("        : " + ", ".join(clob) + ")")
    mf = sum(1 for l in OUT if l.startswith("v_mfma"))
    print(f"// loop body: {mf} MFMA lines in all (2 phases + none in the prologue), {len(OUT)} instructions", file=sys.stderr)


if __name__ == "__main__":
    main()
