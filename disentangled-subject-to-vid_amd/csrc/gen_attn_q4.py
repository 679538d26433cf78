#!/usr/bin/env python3
"""Generator of the hand-scheduled body of attn_q4 (csrc/attention_q4.hip): writes attn_q4_body.inc -- ONE asm statement that
takes a work item from "K / V^T tiles 0-3 / 0-1 staged, Q fragments loaded" to "O^T accumulators and row sums complete" -- and
attn_q4_regs.h (the physical-register constraints and the clobber list of that statement).  Run by build.py when the
generated files are older than this script; the outputs are committed.

Why generated asm: with more than 256 registers per lane hipcc selects the AGPR form for every MFMA builtin and copies the
scores to VGPRs for the softmax, and its allocator does not fit S (128) + -m (32) + P (32) into the 256 VGPRs without
shuttling tuples through AGPRs; the loop is issue-bound (about six fillers per MFMA), so every extra instruction is time.
A first version kept the rare paths in C++ and left the asm at the checks: the register shuffling and scratch spills hipcc put
around each exit cost 3.8 ms of a 10.9 ms launch, so everything between the prologue's DMA and the epilogue is in here.

Two forms from one generator (Q4_JB = 32-row blocks per wave): JB = 2 -> attn_q4, four waves x 64 rows, one wave per SIMD, a K / V^T
fragment feeds two MFMAs; JB = 1 -> attn_q8, eight waves x 32 rows, two waves per SIMD running the SAME fine-grained stream
(MFMA, five fillers, MFMA ...) side by side -- a lone wave issues one VALU per 4.9 cycles (8.9 for v_exp_f32), so at head
dimension 64 (16 MFMA = 512 matrix-pipe cycles against 80 softmax VALU per 32 rows) one wave per SIMD is issue-bound.
Register file (per lane; numbers for JB = 2, the JB = 1 map is the same list packed):
  v[0:127]    S^T accumulators st[buf][j][kb] (16 each): buf = tile parity, j = 32-row block, kb = 32-key half
  v[128:159]  -m of the rows of block j (16 copies: the C operand of the first MFMA of a chain)
  v[160:191]  P as bf16 pairs pk[j][s] (4 each): B operand of the P.V step s (16 keys)
  v[192:199]  exp2 results in flight, two groups of four
  v[200:207]  row sums ps[set][j][x] of the tile whose P is being built (set = tile parity)
  v[208:215]  IN: fragment address of k-step kk in slot 0 (4), staging lane offsets K piece 0 / 1, V^T piece 0 / 1
  v[216:217]  OUT: running row sums l[j];  v[218:229] scratch of the rare paths
  a[0:63]     OUT: O^T accumulators ot[j][db];  a[64:95] IN: Q fragments qf[j][kk];  a[96:127] K fragments;  a[128:159] V^T fragments
  s[36:37] IN: K source, s[38:39] IN: V^T source (next tile to stage: K(4), V^T(2), clamped)
  s44 IN: nt, s45 IN: K tile stride (bytes), s46 IN: Ntok, s47 IN: LDS address of the wave's first piece in slot 0
  s40 t, s41 end of the current phase, s42 / s43 K / V^T source advance per iteration, s48 threshold (f32 bits), s49 return position,
  s[50:51] scratch, s52 OUT: slow paths taken
Phases: A = iterations t < nt - 5 (threshold 2^64: a check fires only for a genuine slow path); B = the last five (threshold -1:
every check enters the rare-path handler, which applies the staging clamp, the tail mask and the real threshold).
"""
import os
import sys

JB = 2  # set by main() per output
# F8 (set by main() for attn_q4f_body.inc, JB = 2 only): Q and K are MX e4m3 images (32-element blocks along the head dimension, one E8M0 scale
# each; produced by qk_quant_mx_k, elementwise.hip) and S^T = K.Q^T is ONE v_mfma_scale_f32_32x32x64_f8f6f4 per (row block, 32-key half):
# 4 MFMA of 64 cycles per tile instead of 16 of 32.  No reference code (the reference has no fp8 path: parity unpinned); P.V stays bf16.
#   K tile in LDS: 64 keys x 64 B (4 KiB = ONE 1-KiB piece per wave), 16-byte chunks XOR-ed with (key >> 2) & 3 on the source address; a lane's
#   operand = chunks hi and 2 + hi of its row (the instruction's logical blocks are bytes 16 b .. 16 b + 15 of both lane halves, tools/probes/mfma_mx.hip),
#   i.e. two ds_read_b128 per 32-key half: v[KIN + 0 / 1] = addresses of the two chunks in slot 0 (v[VIN + 0 .. 3] stay the V^T fragment addresses).
#   K block scales: one dword per lane and tile ([tile][lane] in memory: byte kb = scale of (key 32 kb + (lane & 31), block lane >> 5)), loaded
#   straight into a ring of four VGPRs three tiles ahead (the end-of-iteration vmcnt(4) covers it as it covers the LDS-DMA pieces);
#   Q block scales: v[QS + j], byte 0.
F8 = False
# P16 (set by main() for attn_q4h_body.inc / attn_q4fh_body.inc, JB = 2 only): P and V^T in fp16 instead of bf16 -- the kernel is bound by what ONE
# wave can issue (tools/probes/gap_order.hip, tools/q4_ablate.sh: every instruction of the stream costs its issue slot), the packed fp16 forms issue
# at the plain VALU rate (tools/probes/filler_price.hip; v_pk_add_f32 and every dot2 form block the matrix pipe), and with P as fp16 PAIRS the row
# sums are v_pk_add_f16 on the registers the P.V MFMA reads: 15 packed adds + 4 flush instructions per row block and KV tile instead of 32 adds + 2.
#   * ps[set][j][0] is a PACKED fp16 accumulator (two partial sums of 16 probabilities); seg 2 adds its halves to the fp32 running sum l;
#   * fp16 ends at 65504, so the deferred maximum falls from 2^64 to 2^14 on a partial sum: a row keeps its adopted maximum until a later score
#     exceeds it by ~14 in the exp2 domain (9.7 natural units) -- then the slow path re-adopts the true maximum (an overflowing exp2 becomes +inf
#     in the conversion and trips the same check before P.V or l see the tile).  Below the adopted maximum fp16 reaches 2^-24 (subnormals): a key
#     is dropped only below 2^-25 of the weight of the row's maximum key, the whole tail of N <= 2^17 such keys weighs < 2^-8 of that key alone;
#   * P keeps 11 significant bits (bf16: 8), V^T is converted bf16 -> fp16 by the transpose pass (exact above 2^-14).
P16 = False
# H16 (attn_q4hh_body.inc, JB = 2, with P16): q and k are fp16 too (the fp16 model dtype, src/inference.py:191) -- QK^T on v_mfma_f32_32x32x16_f16.  The
# register map, the schedule and every other instruction are attn_q4h's: staging and fragment reads move 16-bit elements whatever they encode.
H16 = False
# Deferred maximum: a row keeps the maximum its first tile adopted until a partial row sum of a later tile exceeds 2^64, i.e. until some
# p = exp2(s - m) does -- fp32 and bf16 share the exponent range, sums and P.V stay below 2^64 * N * |v| << 2^127, and every quantity is
# scale-free, so nothing is lost by letting m lag (keys 2^126 below the adopted maximum flush to zero, as they would below any maximum).
# Round 2 used 2^13: with a score spread of 4 / 8 / 12 (natural units) 0.9 / 6 / 11 % of the (wave, tile) pairs took the slow path and
# a launch cost 3 / 15 / 27 % more (profiles/r03_attn_slow_path.txt); at 2^64 the scores must rise 44 above the first tile's maximum.
THR_BITS = 0x5F800000


def thr_a():   # phase A threshold: 2^64 as fp32, or (P16) 2^14 as fp16 -- a partial sum of 16 probabilities
    return "0x7400" if P16 else f"0x{THR_BITS:08x}"


def thr_b():   # phase B: -1, every check fires
    return "0xbc00" if P16 else "0xbf800000"


def cmp_thr(s_thr):   # vcc <- v[VS] beyond the threshold in SGPR s_thr (or NaN)
    return f"v_cmp_ngt_f16 vcc, s{s_thr}, {vr(VS)}" if P16 else f"v_cmp_nge_f32 vcc, s{s_thr}, {vr(VS)}"
ORDER = os.environ.get("Q4_ORDER", "")  # placement experiments
READPOS = os.environ.get("Q4_READPOS", "first")  # fragment read first in its gap: -3 % against last (A/B, profiles/r03_attn_q4_placement.txt)
READS = os.environ.get("Q4_READS", "")
DMAPOS = os.environ.get("Q4_DMAPOS", "")
SPLIT = os.environ.get("Q4_SPLIT", "")
# Row sums: "add" = four v_add_f32 per four scores on the fp32 exp2 results (product).  "dot" = v_dot2c_f32_bf16 acc, <1.0 | 1.0>, pk:
# ONE instruction adds the two bf16 values of a packed P register (the numbers the P.V MFMA multiplies) to the row sum -- 8 softmax
# VALU per four scores instead of 10, results correct (harness: 46 checks ok) -- but the dot2 issues slower than the two adds it
# replaces: 7.95 ms against 7.40 at C3 on the same box (tools/q4_ablate.sh build "dot=Q4_SUM=dot" "add=Q4_SUM=add"), so it stays an option.
# "pk" = v_pk_add_f32 on the accumulator pair (two adds per instruction, fp32 as now): 8.15 ms against 7.43.  Both packed forms cost more
# issue time than the two plain VALU they replace (round 2 measured 21.5 cycles per v_pk_*_f32 beside an MFMA stream): on this part the
# softmax stream cannot be shortened by wider VALU instructions.
SUM = os.environ.get("Q4_SUM", "add")
ABLATE = set(filter(None, os.environ.get("Q4_ABLATE", "").split(",")))  # timing experiments only (results are wrong)
S_KPTR, S_VPTR, S_T, S_END, S_KADV, S_VADV, S_NT, S_KSTR, S_NTOK, S_M0W, S_THR, S_RET, S_X0, S_X1, S_CNT, S_ONES = 36, 38, 40, 41, 42, 43, 44, 45, 46, 47, 48, 49, 50, 51, 52, 53
S_KSB, S_KSX, S_NTM1, S_KSA = 54, 56, 57, 58  # F8: IN K block-scale base (64-bit); scratch; nt - 1; address of the tile being loaded (64-bit)


def layout(jb, f8=False, p16=False, h16=False):
    global JB, ST, NEGM, PK, TMP, PS, VIN, LRUN, VS, OT, QF, KF, VF, NM, F8, KS, QS, KIN, P16, H16
    H16 = h16
    JB = jb
    F8 = f8
    P16 = p16
    ST, NEGM, PK, TMP = 0, 64 * jb, 80 * jb, 96 * jb
    PS = TMP + 8
    VIN = PS + 4 * jb
    LRUN = VIN + 8
    VS = LRUN + 2          # scratch: 12 registers
    OT, QF, KF = 0, 32 * jb, 48 * jb
    VF = KF + 32
    NM = 8 * jb            # MFMAs per segment
    KS = VS + 12           # F8: ring of four K block-scale registers (tile t -> KS + (t & 3)); IN: the first three
    QS = KS + 4            # F8: IN Q block scales of row block j
    KIN = QS + 2           # F8: IN [0 / 1] address of chunks hi / 2 + hi of the lane's key in slot 0 (e4m3 K tile), [2] 4 * lane (scale dword of a tile)


def vr(base, n=1):
    return f"v{base}" if n == 1 else f"v[{base}:{base + n - 1}]"


def ar(base, n=1):
    return f"a{base}" if n == 1 else f"a[{base}:{base + n - 1}]"


def st(buf, j, kb):
    return ST + 32 * JB * buf + 32 * j + 16 * kb


def ps(sset, j, x):
    return PS + 2 * JB * sset + 2 * j + x


def pk(j, s):
    return PK + 16 * j + 4 * s


def tmp(g, x):
    return TMP + 4 * (g & 1) + x


def soft_stream(sb, kb, sset):
    """40 JB ops turning half kb of the scores in st[sb] into P: groups g of four values (row block g >> 2, accumulator elements
    4 (g & 3) ..); exp2 of group g interleaved with the row-sum adds and the two v_cvt_pk of group g - 1."""
    def exp(g, x):
        return f"v_exp_f32 {vr(tmp(g, x))}, {vr(st(sb, g >> 2, kb) + (g & 3) * 4 + x)}"

    def add(g, x):
        acc = vr(ps(sset, g >> 2, x & 1))
        if kb == 0 and (g & 3) == 0 and x < 2:  # first touch of this accumulator in the tile: no reset needed
            return f"v_mov_b32 {acc}, {vr(tmp(g, x))}"
        return f"v_add_f32 {acc}, {acc}, {vr(tmp(g, x))}"

    def cvt(g, c):
        e0 = (g & 3) * 4
        dst = pk(g >> 2, kb * 2 + (e0 >> 3)) + ((e0 & 7) >> 1) + c
        return f"v_cvt_pk_bf16_f32 {vr(dst)}, {vr(tmp(g, 2 * c))}, {vr(tmp(g, 2 * c + 1))}"

    def dot(g, c):
        e0 = (g & 3) * 4
        src = pk(g >> 2, kb * 2 + (e0 >> 3)) + ((e0 & 7) >> 1) + c
        acc = vr(ps(sset, g >> 2, c))
        if kb == 0 and (g & 3) == 0:  # first touch of this accumulator in the tile
            return f"v_dot2_f32_bf16 {acc}, {vr(src)}, s{S_ONES}, 0"
        return f"v_dot2c_f32_bf16 {acc}, s{S_ONES}, {vr(src)}"

    def pkadd(g, h):  # two row-sum adds in one v_pk_add_f32: accumulators ps[.][j][0:1] += exp2 results tmp[g][2h : 2h + 1]
        acc = vr(ps(sset, g >> 2, 0), 2)
        if kb == 0 and (g & 3) == 0 and h == 0:
            return f"v_pk_add_f32 {acc}, {vr(tmp(g, 0), 2)}, 0"
        return f"v_pk_add_f32 {acc}, {acc}, {vr(tmp(g, 2 * h), 2)}"

    def cvth(g, c):
        e0 = (g & 3) * 4
        dst = pk(g >> 2, kb * 2 + (e0 >> 3)) + ((e0 & 7) >> 1) + c
        return f"v_cvt_pk_f16_f32 {vr(dst)}, {vr(tmp(g, 2 * c))}, {vr(tmp(g, 2 * c + 1))}"

    def hadd(g, c):  # the packed pair just converted joins the row block's packed fp16 accumulator (None: first register of the tile, folded into the next add)
        e0 = (g & 3) * 4
        src = pk(g >> 2, kb * 2 + (e0 >> 3)) + ((e0 & 7) >> 1) + c
        acc = vr(ps(sset, g >> 2, c))   # two packed accumulators per row block: a dependent v_pk_add_f16 four instructions behind, not two
        if kb == 0 and (g & 3) == 0:
            return f"v_mov_b32 {acc}, {vr(src)}"
        return f"v_pk_add_f16 {acc}, {acc}, {vr(src)}"

    ng = 4 * JB
    if P16:
        # Four per gap, two exp2 + one cvt_pk_f16 + its packed add wherever possible (tools/probes/gap_order.hip: a gap of [e e c a] costs its
        # issue slots, 4 exp2 in one gap cost 8 cycles more).  Segment 1 (kb = 1): [e e e e] [e e c h] x 14 [c h c h].  Segment 2 (kb = 0) converts
        # into P registers that P.V MFMAs of the same segment still read -- pk[j][s] is read last by MFMA 4 s + 2 j + 1 -- so its conversions run
        # TWO groups behind the exp2: [e e e e] [e e e e] [c h e e] x 12 [c h c h] x 2, which writes pk[j][s] from gap 4 s + 8 j + 2 on (asserted):
        # at least one whole gap after the last reader, the distance the bf16 stream keeps.  The two-group TMP ring suffices: a gap converts
        # tmp[g - 2][2 c, 2 c + 1] before its exp2 overwrite the same two registers.
        if kb == 1:
            ops = [exp(0, x) for x in range(4)]
            for g in range(1, ng):
                ops += [exp(g, 0), exp(g, 1), cvth(g - 1, 0), hadd(g - 1, 0), exp(g, 2), exp(g, 3), cvth(g - 1, 1), hadd(g - 1, 1)]
            ops += [cvth(ng - 1, 0), hadd(ng - 1, 0), cvth(ng - 1, 1), hadd(ng - 1, 1)]
            ops += [f"v_pk_add_f16 {vr(ps(sset, j, 0))}, {vr(ps(sset, j, 0))}, {vr(ps(sset, j, 1))}" for j in range(JB)]  # the tile's sums in ps[.][j][0]
        else:
            ops = [exp(0, x) for x in range(4)] + [exp(1, x) for x in range(4)]
            for g in range(2, ng):
                ops += [cvth(g - 2, 0), hadd(g - 2, 0), exp(g, 0), exp(g, 1), cvth(g - 2, 1), hadd(g - 2, 1), exp(g, 2), exp(g, 3)]
            for g in (ng - 2, ng - 1):
                ops += [cvth(g, 0), hadd(g, 0), cvth(g, 1), hadd(g, 1)]
            for pos, o in enumerate(ops):
                if o.startswith("v_cvt_pk_f16_f32"):
                    r = int(o.split()[1].rstrip(",")[1:]) - PK
                    jj, ss = r // 16, (r % 16) // 4
                    assert pos // 4 >= 4 * ss + 2 * jj + 2, (o, pos)
        assert len(ops) == 32 * JB + (JB if kb == 1 else 0)
        return ops
    if SUM == "pk":
        ops = [exp(0, x) for x in range(4)]
        for g in range(1, ng):
            ops += [exp(g, 0), exp(g, 1), pkadd(g - 1, 0), cvt(g - 1, 0), exp(g, 2), exp(g, 3), pkadd(g - 1, 1), cvt(g - 1, 1)]
        ops += [pkadd(ng - 1, 0), cvt(ng - 1, 0), pkadd(ng - 1, 1), cvt(ng - 1, 1)]
        assert len(ops) == 32 * JB
        return ops
    if SUM == "dot":  # per group of four scores: exp2 x 4 of this group, cvt_pk x 2 + dot2 x 2 of the group before
        ops = [exp(0, x) for x in range(4)]
        for g in range(1, ng):
            ops += [exp(g, 0), exp(g, 1), cvt(g - 1, 0), dot(g - 1, 0), exp(g, 2), exp(g, 3), cvt(g - 1, 1), dot(g - 1, 1)]
        ops += [cvt(ng - 1, 0), dot(ng - 1, 0), cvt(ng - 1, 1), dot(ng - 1, 1)]
        assert len(ops) == 32 * JB
        return ops
    if ORDER == "uniform":  # every gap of five = [exp, add, exp, add, cvt]: two exp2 per gap; adds / cvt belong to the group before
        ops = []
        for g in range(ng + 1):
            for h in range(2):
                if g < ng:
                    ops.append(exp(g, 2 * h))
                if g > 0:
                    ops.append(add(g - 1, 2 * h))
                if g < ng:
                    ops.append(exp(g, 2 * h + 1))
                if g > 0:
                    ops.append(add(g - 1, 2 * h + 1))
                    ops.append(cvt(g - 1, h))
    elif ORDER == "explast":  # per pair of gaps: [add add cvt cvt exp][add add exp exp exp]: exp2 at the END of a gap
        ops = [exp(0, x) for x in range(4)]
        for g in range(1, ng):
            ops += [add(g - 1, 0), add(g - 1, 1), cvt(g - 1, 0), add(g - 1, 2), add(g - 1, 3), cvt(g - 1, 1), exp(g, 0), exp(g, 1), exp(g, 2), exp(g, 3)]
        ops += [add(ng - 1, x) for x in range(4)] + [cvt(ng - 1, 0), cvt(ng - 1, 1)]
    else:
        ops = [exp(0, x) for x in range(4)]
        for g in range(1, ng):
            ops += [exp(g, 0), exp(g, 1), add(g - 1, 0), add(g - 1, 1), exp(g, 2), exp(g, 3), add(g - 1, 2), add(g - 1, 3),
                    cvt(g - 1, 0), cvt(g - 1, 1)]
        ops += [add(ng - 1, x) for x in range(4)] + [cvt(ng - 1, 0), cvt(ng - 1, 1)]
    assert len(ops) == 40 * JB
    return ops


def frag_read(dst_base, i, slot, is_v):
    # fragment i = (k-step i >> 1, 32-row half i & 1) of the K (V^T) tile in ring slot `slot`
    off = slot * 16384 + (8192 if is_v else 0) + (i & 1) * 4096
    return f"ds_read_b128 {ar(dst_base + 4 * i, 4)}, {vr(VIN + (i >> 1))} offset:{off}"


def frag_read_k8(i, slot):
    # F8: fragment i = (32-key half i >> 1, chunk pair member i & 1) of the e4m3 K tile in ring slot `slot`
    kb, c = i >> 1, i & 1
    return f"ds_read_b128 {ar(KF + 8 * kb + 4 * c, 4)}, {vr(KIN + c)} offset:{slot * 16384 + kb * 2048}"


def qk_mfma_f8(buf, i, ring):
    j, kb = i % JB, i // JB
    sel = f" op_sel:[{kb},0,0] op_sel_hi:[0,0,0]"
    return (f"v_mfma_scale_f32_32x32x64_f8f6f4 {vr(st(buf, j, kb), 16)}, {ar(KF + 8 * kb, 8)}, {ar(QF + 8 * j, 8)}, {vr(NEGM + 16 * j, 16)}, "
            f"{vr(KS + ring)}, {vr(QS + j)}{sel}")


def qk_mfma(buf, i):
    j, kb, kk = i % JB, (i // JB) & 1, i // (2 * JB)
    d = vr(st(buf, j, kb), 16)
    c = vr(NEGM + 16 * j, 16) if kk == 0 else d
    return f"v_mfma_f32_32x32x16_{'f16' if H16 else 'bf16'} {d}, {ar(KF + 4 * (kk * 2 + kb), 4)}, {ar(QF + 16 * j + 4 * kk, 4)}, {c}"


def pv_mfma(i):
    db, j, s = i & 1, (i >> 1) % JB, i // (2 * JB)
    d = ar(OT + 32 * j + 16 * db, 16)
    return f"v_mfma_f32_32x32x16_{'f16' if P16 else 'bf16'} {d}, {ar(VF + 4 * (s * 2 + db), 4)}, {vr(pk(j, s), 4)}, {d}"


def max_ps(emit, cur):
    """v[VS] = maximum of the partial row sums of tile set `cur`"""
    if P16:  # two packed fp16 accumulators: v[VS] (low half) = the largest of the four partial sums (an overflow is +inf)
        assert JB == 2
        emit(f"v_pk_max_f16 {vr(VS)}, {vr(ps(cur, 0, 0))}, {vr(ps(cur, 1, 0))}")
        emit(f"v_max_f16_sdwa {vr(VS)}, {vr(VS)}, {vr(VS)} dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1")
        return
    regs = [ps(cur, j, x) for j in range(JB) for x in range(2)]
    if len(regs) == 2:
        emit(f"v_max_f32 {vr(VS)}, {vr(regs[0])}, {vr(regs[1])}")
    else:
        emit(f"v_max3_f32 {vr(VS)}, {vr(regs[0])}, {vr(regs[1])}, {vr(regs[2])}")
        emit(f"v_max_f32 {vr(VS)}, {vr(VS)}, {vr(regs[3])}")


def slow_path(emit, cur, first):
    """raise the running maximum of the row blocks to the true maximum of the tile in st[cur], rescale everything at the old
    scale, redo P and the row sums of the whole tile; S of the next tile (other buffer) was computed against the old maximum and
    is shifted as well.  first: tile 0 adopts its maximum (O and l are zero, exp2(-d) could overflow) and only the first half of
    P(0) is built: segment 1 of iteration 0 builds the second half, as for every tile"""
    nxt = cur ^ 1
    vA, vB, vD, vAl, T = VS, VS + 1, VS + 2, VS + 3, VS + 4  # T: 8 temporaries
    for j in range(JB):
        s = [st(cur, j, 0) + e for e in range(16)] + [st(cur, j, 1) + e for e in range(16)]
        emit(f"v_max3_f32 {vr(vA)}, {vr(s[0])}, {vr(s[1])}, {vr(s[2])}")
        for x in range(3, 31, 2):
            emit(f"v_max3_f32 {vr(vA)}, {vr(vA)}, {vr(s[x])}, {vr(s[x + 1])}")
        emit(f"v_max_f32 {vr(vA)}, {vr(vA)}, {vr(s[31])}")
        emit(f"v_mov_b32 {vr(vB)}, {vr(vA)}")
        emit("s_nop 1")
        emit(f"v_permlane32_swap_b32 {vr(vA)}, {vr(vB)}")  # vA = lower half's value in both halves, vB = upper half's
        emit("s_nop 1")
        emit(f"v_max_f32 {vr(vD)}, {vr(vA)}, {vr(vB)}")
        if not first:
            emit(f"v_max_f32 {vr(vD)}, 0, {vr(vD)}")
            emit(f"v_exp_f32_e64 {vr(vAl)}, -{vr(vD)}")
            emit("s_nop 0")
            emit(f"v_mul_f32 {vr(LRUN + j)}, {vr(LRUN + j)}, {vr(vAl)}")
            for e0 in range(0, 32, 4):
                for k in range(4):
                    emit(f"v_accvgpr_read_b32 {vr(T + k)}, {ar(OT + 32 * j + e0 + k)}")
                for k in range(4):
                    emit(f"v_mul_f32 {vr(T + k)}, {vr(T + k)}, {vr(vAl)}")
                for k in range(4):
                    emit(f"v_accvgpr_write_b32 {ar(OT + 32 * j + e0 + k)}, {vr(T + k)}")
        for e in range(16):
            emit(f"v_sub_f32 {vr(NEGM + 16 * j + e)}, {vr(NEGM + 16 * j + e)}, {vr(vD)}")
        for x in range(32):
            emit(f"v_sub_f32 {vr(s[x])}, {vr(s[x])}, {vr(vD)}")
        nexp = 16 if first else 32
        for x0 in range(0, nexp, 4):  # four exp2, then their sums and packs (a transcendental's result is not read by the next instruction)
            for k in range(4):
                emit(f"v_exp_f32 {vr(T + k)}, {vr(s[x0 + k])}")
            emit("s_nop 0")
            kb, e = x0 >> 4, x0 & 15
            dst = pk(j, kb * 2 + (e >> 3)) + ((e & 7) >> 1)
            if P16:  # against the true maximum every probability is <= 1: no overflow.  The tile's sums are rebuilt in ps[.][j][0]; after the
                # first tile's adoption segment 1 continues into both accumulators, so the second one starts at zero
                acc = vr(ps(cur, j, 0))
                if first and x0 == 0:
                    emit(f"v_mov_b32 {vr(ps(cur, j, 1))}, 0")
                emit(f"v_cvt_pk_f16_f32 {vr(dst)}, {vr(T)}, {vr(T + 1)}")
                emit(f"v_cvt_pk_f16_f32 {vr(dst + 1)}, {vr(T + 2)}, {vr(T + 3)}")
                emit("s_nop 0")
                if x0 == 0:
                    emit(f"v_pk_add_f16 {acc}, {vr(dst)}, {vr(dst + 1)}")
                else:
                    emit(f"v_pk_add_f16 {acc}, {acc}, {vr(dst)}")
                    emit(f"v_pk_add_f16 {acc}, {acc}, {vr(dst + 1)}")
                continue
            for k in range(4):
                acc = vr(ps(cur, j, k & 1))
                emit(f"v_mov_b32 {acc}, {vr(T + k)}" if x0 == 0 and k < 2 else f"v_add_f32 {acc}, {acc}, {vr(T + k)}")
            emit(f"v_cvt_pk_bf16_f32 {vr(dst)}, {vr(T)}, {vr(T + 1)}")
            emit(f"v_cvt_pk_bf16_f32 {vr(dst + 1)}, {vr(T + 2)}, {vr(T + 3)}")
        if not first:
            for kb in range(2):
                for e in range(16):
                    r = st(nxt, j, kb) + e
                    emit(f"v_sub_f32 {vr(r)}, {vr(r)}, {vr(vD)}")


def mask_tile(emit, buf, s_kv0):
    """keys >= Ntok of the tile in st[buf] (first key in SGPR s_kv0) get score -inf; a clamped duplicate past the end is masked
    completely.  Element e of half kb is key kv0 + 32 kb + (e & 3) + 8 (e >> 2) + 4 (lane >> 5)."""
    vL, vBase, vInf = VS + 4, VS + 5, VS + 6
    emit(f"v_mbcnt_lo_u32_b32 {vr(vL)}, -1, 0")
    emit(f"v_mbcnt_hi_u32_b32 {vr(vL)}, -1, {vr(vL)}")
    emit(f"v_lshrrev_b32 {vr(vL)}, 5, {vr(vL)}")
    emit(f"v_lshlrev_b32 {vr(vL)}, 2, {vr(vL)}")
    emit(f"s_sub_u32 s{S_X1}, s{S_NTOK}, s{s_kv0}")  # keys of this tile below Ntok (<= 0: none)
    emit(f"v_sub_u32 {vr(vBase)}, s{S_X1}, {vr(vL)}")
    emit(f"v_mov_b32 {vr(vInf)}, 0xff800000")
    for kb in range(2):
        for e in range(16):
            c = kb * 32 + (e & 3) + 8 * (e >> 2)
            emit(f"v_cmp_ge_i32 vcc, {c}, {vr(vBase)}")
            for j in range(JB):
                r = st(buf, j, kb) + e
                emit(f"v_cndmask_b32 {vr(r)}, {vr(r)}, {vr(vInf)}, vcc")


def gen():
    L = []
    ctx = set()

    def emit(ln):
        op = ln.split()[0]
        if "nodma" in ABLATE and op == "global_load_lds_dwordx4":
            return
        if "noread" in ABLATE and op == "ds_read_b128":
            return
        if "nosoft" in ABLATE and op in ("v_exp_f32", "v_add_f32", "v_mov_b32", "v_cvt_pk_bf16_f32", "v_cvt_pk_f16_f32", "v_pk_add_f16"):
            return
        if "movexp" in ABLATE and op == "v_exp_f32":
            ln = ln.replace("v_exp_f32", "v_mov_b32")
        if "nobar" in ABLATE and op == "s_barrier":
            return
        if "nowait" in ABLATE and ln.startswith("s_waitcnt lgkmcnt"):
            return
        if "nomfma" in ABLATE and op.startswith("v_mfma"):
            return
        if "noadd" in ABLATE and (op == "v_add_f32" or (op == "v_mov_b32" and "main" in ctx)):  # row sums off the VALU (timing of a sum-by-MFMA form)
            return
        L.append(ln)

    def soft_lo(i):  # first softmax op of gap i: five per gap (four with the dot2 row sums), or (SPLIT = "64") six in gaps without a fragment read and four in those with one
        if SUM in ("dot", "pk") or P16:
            return 4 * i
        if SPLIT == "64" and JB == 2:
            return 5 * i + (i & 1)
        return 5 * i

    # where the extra fillers of a segment go (gap index -> instruction), besides five softmax ops per gap
    gap_m0 = [3, 9] if JB == 2 else [2]     # M0 <- LDS address of the wave's piece p
    gap_dma = [5, 11] if JB == 2 else [4]   # the piece's LDS-DMA (at least one instruction after the M0 write)
    read_gaps = range(0, NM, JB) if READS != "front" else range(8)            # eight fragment reads per segment

    emit(f"; ---- attn_q{8 // JB} body (generated by gen_attn_q4.py; do not edit)")
    # ---------------- state
    for r in range(32 * JB):
        emit(f"v_accvgpr_write_b32 {ar(OT + r)}, 0")
    for r in range(16 * JB):
        emit(f"v_mov_b32 {vr(NEGM + r)}, 0")
    for j in range(JB):
        emit(f"v_mov_b32 {vr(LRUN + j)}, 0")
    emit(f"s_mov_b32 s{S_T}, 0")
    emit(f"s_mov_b32 s{S_CNT}, 0")               # OUT: slow paths taken by this wave (diagnostics)
    emit(f"s_mov_b32 s{S_THR}, {thr_a()}")      # 2^64 (P16: 2^14 as fp16)
    emit(f"s_mov_b32 s{S_ONES}, 0x3f803f80")         # bf16 (1.0, 1.0)
    emit(f"s_mov_b32 s{S_KADV}, s{S_KSTR}")
    emit(f"s_mov_b32 s{S_VADV}, 128")
    if F8:
        emit(f"s_sub_u32 s{S_NTM1}, s{S_NT}, 1")
    emit(f"s_sub_u32 s{S_END}, s{S_NT}, 5")      # phase A ends at nt - 5 ...
    emit(f"s_cmp_gt_i32 s{S_END}, 0")
    emit("s_cbranch_scc1 L_q4_pa_%=")
    emit(f"s_mov_b32 s{S_END}, s{S_NT}")         # ... or there is none: phase B from the start
    emit(f"s_mov_b32 s{S_THR}, {thr_b()}")
    emit(f"s_mov_b32 s{S_KADV}, 0")
    emit("L_q4_pa_%=:")
    # ---------------- prologue: K(0) fragments, S(0) -> st[0], K(1) fragments, tail mask, adoption of tile 0's maxima
    if F8:
        for i in range(4):
            emit(frag_read_k8(i, 0))
        emit("s_waitcnt lgkmcnt(0)")
        for i in range(2 * JB):
            emit(qk_mfma_f8(0, i, 0))
        for i in range(4):
            emit(frag_read_k8(i, 1))
        emit("s_nop 15")
    else:
        for i in range(8):
            emit(frag_read(KF, i, 0, False))
        emit("s_waitcnt lgkmcnt(0)")
        for i in range(NM):
            emit(qk_mfma(0, i))
        for i in range(8):
            emit(frag_read(KF, i, 1, False))
    emit("s_nop 15")
    emit("s_nop 15")
    emit(f"s_cmp_ge_u32 s{S_NTOK}, 64")
    emit("s_cbranch_scc1 L_q4_nomask0_%=")
    emit(f"s_mov_b32 s{S_X0}, 0")
    mask_tile(emit, 0, S_X0)
    emit("L_q4_nomask0_%=:")
    slow_path(emit, 0, True)
    emit("s_branch L_q4_e0_%=")
    # ---------------- main loop, unrolled over four tiles (U = t mod 4: st buffers and LDS slots are immediates)
    ctx.add("main")
    for U in range(4):
        cur, nxt = U & 1, (U & 1) ^ 1
        # segment 1: S(t+1) = K(t+1).Q^T -> st[nxt]; second half of P(t); V^T(t) fragments; K(t+4) pieces
        emit(f"L_q4_e{U}_%=:")
        emit("s_waitcnt lgkmcnt(0)")
        soft = soft_stream(cur, 1, cur)
        for i in range(NM):
            first, last = [], []
            if i in read_gaps:
                (first if READPOS == "first" else last).append(frag_read(VF, list(read_gaps).index(i), U, True))
            for p in range(1 if F8 else JB):  # F8: an e4m3 K tile is one piece per wave
                if i == gap_m0[p]:
                    last.append(f"s_add_u32 m0, s{S_M0W}, {U * 16384 + p * 4096}")
                if i == gap_dma[p]:
                    (first if DMAPOS == "first" else last).append(f"global_load_lds_dwordx4 {vr(VIN + 4 + p)}, s[{S_KPTR}:{S_KPTR + 1}]")
            if F8 and i == gap_m0[1]:   # block scales of K tile min(t + 3, nt - 1) -> ring register (t + 3) & 3
                last += [f"s_add_u32 s{S_KSX}, s{S_T}, 3", f"s_min_u32 s{S_KSX}, s{S_KSX}, s{S_NTM1}", f"s_lshl_b32 s{S_KSX}, s{S_KSX}, 8"]
            if F8 and i == gap_m0[1] + 1:
                last += [f"s_add_u32 s{S_KSA}, s{S_KSB}, s{S_KSX}", f"s_addc_u32 s{S_KSA + 1}, s{S_KSB + 1}, 0"]
            if F8 and i == gap_dma[1]:
                last.append(f"global_load_dword {vr(KS + ((U + 3) & 3))}, {vr(KIN + 2)}, s[{S_KSA}:{S_KSA + 1}]")
            if i == NM - 2:
                last += [f"s_add_u32 s{S_KPTR}, s{S_KPTR}, s{S_KADV}", f"s_addc_u32 s{S_KPTR + 1}, s{S_KPTR + 1}, 0"]
            if F8:  # four 64-cycle MFMAs, one per four gaps of the softmax stream
                head = [qk_mfma_f8(nxt, i // 4, (U + 1) & 3)] if i % 4 == 0 else []
            else:
                head = [qk_mfma(nxt, i)]
            for ln in head + first + soft[soft_lo(i):(soft_lo(i + 1) if i + 1 < NM else len(soft))] + last:
                emit(ln)
        # check of tile t: the partial row sums against the threshold (any lane)
        max_ps(emit, cur)
        emit(cmp_thr(S_THR))  # row sum > threshold, or NaN
        emit(f"s_cbranch_vccnz L_q4_x{U}_%=")
        # segment 2: O += V^T(t).P(t); first half of P(t+1); K(t+2) fragments; V^T(t+2) pieces; l += row sums of t
        emit(f"L_q4_e{4 + U}_%=:")
        emit("s_waitcnt lgkmcnt(0)")
        soft = soft_stream(nxt, 0, nxt)
        slot2 = (U + 2) & 3
        ladd = {1: (0, 0), 7: (0, 1), 13: (1, 0), 15: (1, 1)} if JB == 2 else {1: (0, 0), 7: (0, 1)}
        for i in range(NM):
            first, last = [], []
            if F8:
                if i in (0, 2, 4, 6):
                    (first if READPOS == "first" else last).append(frag_read_k8(i >> 1, slot2))
            elif i in read_gaps:
                (first if READPOS == "first" else last).append(frag_read(KF, list(read_gaps).index(i), slot2, False))
            if P16:  # l += the two halves of the packed fp16 accumulator of tile t (complete since segment 1): one instruction per gap
                jj, st4 = i >> 3, (i & 7) >> 1
                if (i & 1) == 1:
                    acc, t0 = vr(ps(cur, jj, 0)), vr(VS + 10 + (st4 & 1))
                    last.append({0: f"v_cvt_f32_f16 {t0}, {acc}",
                                 1: f"v_cvt_f32_f16_sdwa {t0}, {acc} dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1",
                                 2: f"v_add_f32 {vr(LRUN + jj)}, {vr(LRUN + jj)}, {vr(VS + 10)}",
                                 3: f"v_add_f32 {vr(LRUN + jj)}, {vr(LRUN + jj)}, {vr(VS + 11)}"}[st4])
            elif i in ladd:
                jj, x = ladd[i]
                last.append(f"v_add_f32 {vr(LRUN + jj)}, {vr(LRUN + jj)}, {vr(ps(cur, jj, x))}")
            for p in range(JB):
                if i == gap_m0[p]:
                    last.append(f"s_add_u32 m0, s{S_M0W}, {slot2 * 16384 + 8192 + p * 4096}")
                if i == gap_dma[p]:
                    (first if DMAPOS == "first" else last).append(f"global_load_lds_dwordx4 {vr(VIN + 6 + p)}, s[{S_VPTR}:{S_VPTR + 1}]")
            extra = []
            if "summfma" in ABLATE and (i & 1) == 1:  # timing only: one more MFMA per (row block, k-step), as a sum-by-MFMA form would issue
                jx = (i >> 1) % JB
                extra = [f"v_mfma_f32_32x32x16_bf16 {ar(VF + 32 + 16 * jx, 16)}, {ar(VF + 64, 4)}, {vr(pk(jx, i // (2 * JB)), 4)}, {ar(VF + 32 + 16 * jx, 16)}"]
            for ln in [pv_mfma(i)] + first + soft[soft_lo(i):soft_lo(i + 1)] + last + extra:
                emit(ln)
        emit(f"s_add_u32 s{S_VPTR}, s{S_VPTR}, s{S_VADV}")
        emit(f"s_addc_u32 s{S_VPTR + 1}, s{S_VPTR + 1}, 0")
        emit(f"s_waitcnt vmcnt({2 * JB})")
        emit("s_barrier")
        emit(f"s_add_u32 s{S_T}, s{S_T}, 1")
        emit(f"s_cmp_lt_u32 s{S_T}, s{S_END}")
        emit("s_cbranch_scc0 L_q4_phase_%=")
        if U == 3:
            emit("s_branch L_q4_e0_%=")
    ctx.discard("main")
    # ---------------- end of a phase: A -> B (threshold -1, no more K advance), or finished
    emit("L_q4_phase_%=:")
    emit(f"s_cmp_ge_u32 s{S_T}, s{S_NT}")
    emit("s_cbranch_scc1 L_q4_done_%=")
    emit(f"s_mov_b32 s{S_END}, s{S_NT}")
    emit(f"s_mov_b32 s{S_THR}, {thr_b()}")
    emit(f"s_mov_b32 s{S_KADV}, 0")
    emit(f"s_and_b32 s{S_X0}, s{S_T}, 3")
    for U in range(1, 4):
        emit(f"s_cmp_eq_u32 s{S_X0}, {U}")
        emit(f"s_cbranch_scc1 L_q4_e{U}_%=")
    emit("s_branch L_q4_e0_%=")
    # ---------------- rare-path handlers: the check of position U fired (st[U & 1] = S(t), st[~U & 1] = S(t+1))
    for U in range(4):
        emit(f"L_q4_x{U}_%=:")
        emit(f"s_mov_b32 s{S_RET}, {U}")
        emit(f"s_branch L_q4_rare{U & 1}_%=")
    for cur in range(2):
        nxt = cur ^ 1
        emit(f"L_q4_rare{cur}_%=:")
        emit("s_nop 15")  # MFMA results of segment 1 are read below
        emit("s_nop 15")
        emit(f"s_add_u32 s{S_X0}, s{S_T}, 3")      # V^T(t+2) is staged by this iteration: the source advances while t + 3 < nt
        emit(f"s_cmp_lt_u32 s{S_X0}, s{S_NT}")
        emit(f"s_cselect_b32 s{S_VADV}, 128, 0")
        emit(f"s_add_u32 s{S_X0}, s{S_T}, 1")      # tail mask of S(t+1): first key 64 (t+1); needed when it ends past Ntok
        emit(f"s_lshl_b32 s{S_X0}, s{S_X0}, 6")
        emit(f"s_add_u32 s{S_X1}, s{S_X0}, 64")
        emit(f"s_cmp_gt_u32 s{S_X1}, s{S_NTOK}")
        emit(f"s_cbranch_scc0 L_q4_nomask{1 + cur}_%=")
        mask_tile(emit, nxt, S_X0)
        emit(f"L_q4_nomask{1 + cur}_%=:")
        max_ps(emit, cur)
        emit(f"s_mov_b32 s{S_X0}, {thr_a()}")
        emit(cmp_thr(S_X0))
        emit(f"s_cbranch_vccz L_q4_ret{cur}_%=")
        emit(f"s_add_u32 s{S_CNT}, s{S_CNT}, 1")
        slow_path(emit, cur, False)
        emit(f"L_q4_ret{cur}_%=:")
        emit("s_nop 7")
        for U in (cur, cur + 2):
            emit(f"s_cmp_eq_u32 s{S_RET}, {U}")
            emit(f"s_cbranch_scc1 L_q4_e{4 + U}_%=")
        emit("s_trap 2")
    emit("L_q4_done_%=:")
    emit("s_waitcnt vmcnt(0)")
    emit("s_nop 15")  # the epilogue reads the O^T accumulators next
    emit("s_nop 15")
    return L


def main():
    here = os.environ.get("S2V_GEN_OUT") or os.path.dirname(os.path.abspath(__file__))  # S2V_GEN_OUT: tests/test_host_cpu.py regenerates into a scratch directory
    with open(os.path.join(here, "attn_q4_regs.h"), "w") as f:
        f.write("// generated by gen_attn_q4.py: the physical registers the bodies of attn_q4 (JB = 2) / attn_q8 (JB = 1) own\n#pragma once\n")
        for jb, name, f8, p16, h16 in ((2, "Q4", False, False, False), (1, "Q8", False, False, False), (2, "Q4F", True, False, False), (2, "Q4H", False, True, False),
                                       (2, "Q4FH", True, True, False), (2, "Q4HH", False, True, True)):
            layout(jb, f8, p16, h16)
            with open(os.path.join(here, f"attn_{name.lower()}_body.inc"), "w") as g:
                for ln in gen():
                    g.write('"' + ln + '\\n\\t"\n')
            clob = [f"v{r}" for r in list(range(0, VIN)) + list(range(VS, VS + 12))] + [f"a{r}" for r in range(KF, VF + 32)]
            clob += [f"s{r}" for r in (S_T, S_END, S_KADV, S_VADV, S_THR, S_RET, S_X0, S_X1, S_ONES)]
            extra = []
            if f8:
                clob += [f"s{r}" for r in (S_KSX, S_NTM1, S_KSA, S_KSA + 1)]
                extra = [("KS", "v", KS, 4), ("QS", "v", QS, 2), ("KIN", "v", KIN, 4), ("KSB", "s", S_KSB, 2)]  # KS is in / out ("+"): the body reloads the ring
            for nm, cls, base, n in [("VIN", "v", VIN, 8), ("LRUN", "v", LRUN, 2), ("QF", "a", QF, (8 if f8 else 16) * jb), ("PTR", "s", S_KPTR, 4),
                                     ("SIN", "s", S_NT, 4)] + extra + [(f"OT{j}", "a", OT + 32 * j, 32) for j in range(jb)]:
                f.write(f'#define {name}_{nm} "{{{cls}[{base}:{base + n - 1}]}}"\n')
            f.write(f'#define {name}_CNT "{{s{S_CNT}}}"\n')
            f.write(f"#define {name}_CLOBBERS " + ", ".join(f'"{c}"' for c in clob) + ', "vcc", "scc", "m0", "memory"\n')


if __name__ == "__main__":
    main()
