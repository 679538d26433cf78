#!/usr/bin/env python3
"""Generator of gemm_g4t (csrc/gemm_g4t.hip): the four-wave 256 x 256 GEMM of gen_gemm_g4.py as a PERSISTENT kernel whose epilogue is
trickled through the MFMA gaps of the NEXT tile's K loop.  Writes gemm_g4t_body<EPI>.inc (ONE asm statement = the whole walk of a
workgroup over its tiles) and gemm_g4t_regs.h.  Run by build.py when stale; the outputs are committed.

Why (profiles/r03_pmc_sq.md): gemm_g4 runs its K loop within 4-19 % of the matrix pipe's floor, but one workgroup
per CU (512 registers per wave, 128 KiB of LDS) means that while a tile's epilogue runs -- 17.8 % of an FF1 tile: bias + GELU is VALU
issue on ONE wave per SIMD, 256 outputs per lane at ~5 cycles per instruction -- the matrix pipe idles, and so it does during the ~3 us
of every tile's prologue (first operands in flight).  The K loop itself carries 1.1 filler instructions per MFMA where the pipe hides
about five.

Structure of one tile iteration (register map below):
  1. the tile's record (operand / output pointers, made by the C++ preamble) from LDS into SGPRs;
  2. [tile i-1 exists] its bias, 16 x 8 bytes per lane, requested into the (idle) fragment registers;
  3. barrier (every wave is done reading tile i-1's operand stages), then the prologue LDS-DMA of tile i: K-tile 0 and the A half of
     K-tile 1 -- 24 pieces in flight;
  4. [tile i-1 exists] the DRAIN, hidden under that DMA latency: accumulators of tile i-1 -> + bias -> v_cvt_pk_bf16_f32 -> the 128
     packed registers E (the linear's bf16 output, exactly the value the C++ epilogue rounds to);
  5. the K loop of tile i; its first TK + 1 K-tiles are unrolled and carry the TRICKLE of tile i-1 behind their MFMAs (TPS issue slots
     per MFMA): per 32-row x 64-column unit [GELU in place on E], 8 ds_write_b64 into the wave's 4-KiB patch (rows of 128 B, 16-byte
     chunks XOR-swizzled by row & 7), 4 ds_read_b128 row-major, 4 full-line global_store_dwordx4.  LDS ops of a wave complete in order
     and the loop already waits lgkmcnt(0) at every step; stores are only issued in steps 3 / 0 / 1 so that the loop's own vmcnt(0) of
     step 3 (for the LDS-DMA) never waits for a young store;
  6. the generic loop and the three tail K-tiles of gen_gemm_g4.py, unchanged.
The LAST tile of a workgroup leaves its accumulators to the C++ epilogue (gemm_epi.h), as gemm_g4 does for every tile.
Arithmetic is that of gemm_epi.h instruction for instruction (y = acc + bias, one v_cvt_pk_bf16_f32; GELU = x * rcp(1 + exp2(x *
fma(x * x, k1, k0))) on the rounded value): results are bit-identical to gemm_g4 (tests/test_gpu_gemm_schedules.py).

Round 5, gemm_g4t_body_qknorm.inc: the fused QKV projection (EPI_BIAS_QKNORM).  Tiles whose previous tile held v heads take the bias trickle above;
q / k tiles take qk_program(): the previous tile's rows get their per-head LayerNorm + affine + rotary embedding in the READ-BACK layout (lane = row,
octet -- the layout and order of operations of gemm_epi.h, hence the same bits) between the patch read-back and the stores, ~165 VALU per row, 32 rows
per lane, 32 K-tiles at three slots per MFMA.  Registers: the E registers of units already written to the patch (unit 0: the row's values + temporaries;
units 1, 2: a four-row ring of rotary values requested three rows ahead); the lane constants, LayerNorm parameters and position arithmetic come from an LDS
block and v99's upper bits because no SGPR or VGPR is left to pass them in.  Vector-memory instructions (rotary loads, stores) leave only in the loop's store
steps: met elsewhere they are parked and the program runs on; consumers wait with vmcnt(number of vector-memory instructions issued since), which the
generator counts (one in-order counter over loads, LDS-DMA and stores on gfx9).

Hazards handled by hand (the assembler inserts nothing): v_exp / v_rcp results are never consumed by the next instruction (gfx950 trans-use
hazard: the two chains of a packed pair are interleaved); SGPRs written by v_readfirstlane reach memory instructions only through s_mov /
s_add; (qknorm) two wait states between a VALU write and a DPP read of the register and between v_cmp and the v_cndmask that reads its mask, one after
v_sqrt / v_rcp; MFMA results are read (v_accvgpr_read) a barrier and 24 DMA issues after the last MFMA.

Registers.  a[0:255] accumulators.  v[0:63] fragments (bias + drain temporaries between tiles); v[64:79] IN fragment addresses; v[80:95] IN
staging offsets; v96 IN store offset of the lane ((lane >> 3) * ldc + (lane & 7) * 8) * 2; v97 IN patch write address of the lane;
v98 IN patch read address; v99 IN bias offset (hi * 8); v[100:227] E; v[228:243] R (read-back); v[244:253] GELU temporaries; v254 IN LDS
address of the workgroup's next tile record; v255 GELU k0.
s[36:39] operand pointers; s40 IN M0 base of the wave's pieces; s41 loop counter; s42 IN (nT - 4) / 2; s43 IN tiles of this workgroup;
s[44:47] previous tile's bias / C pointers; s48 IN 16 * ldc (bytes of 8 output rows); s49 have_prev; s[50:51] store address; s52
0xffff0000; s54 IN the wave's C offset; s55 IN the wave's bias offset; s[56:57] IN bias; s[58:61] this tile's C / bias pointers; s62
scratch (GELU k1 during the trickle); s[64:71] the record.
"""
import os
import sys

EPI = {"bias": 0, "gelu": 1, "qknorm": 4}
# Stores and the loop's vmcnt: vmcnt counts loads, LDS-DMA and stores in issue order (gfx9: one counter, in-order), and a store is only
# counted down when L2 has acknowledged it -- measured: with stores anywhere in the K-tile and the loop's vmcnt(0) of step 3, the 128 KiB
# of C stores of a tile cost 0.12 ms of a 2.17-ms FF1 launch (tools/g4t_ablate.sh nostore).  So stores go out in steps 1-2 only, BEHIND
# the K-tile's W pieces (step 0), and the wait of step 3 is vmcnt(number of those stores): every LDS-DMA piece has landed, the young
# stores stay in flight and have a whole further K-tile to complete (the next wait covers them).
STORE_STEPS = tuple(int(x) for x in os.environ.get("G4T_STORE_STEPS", "1,2").split(","))
COUNTED = os.environ.get("G4T_COUNTED", "1") == "1"
SPREAD = int(os.environ.get("G4T_SPREAD", "1"))
TPS = float(os.environ.get("G4T_TPS", "3"))   # trickle issue slots per MFMA (a transcendental counts two, SALU a half)
ABLATE = set(filter(None, os.environ.get("G4T_ABLATE", "").split(",")))
TPS_QK = float(os.environ.get("G4T_TPS_QK", "3"))   # the same for the q/k-norm trickle
QK_DIST = int(os.environ.get("G4T_QK_DIST", "64"))  # MFMAs between the request of a row's rotary values and their first use (below: the trickle idles)

FRAG, VADDR, VOFF = 0, 64, 80
V_ST, V_DSW, V_DSR, V_BOFF, E0, R0, G0, V_TAB, V_K0 = 96, 97, 98, 99, 100, 228, 244, 254, 255
S_A, S_W, S_M0W, S_CNT, S_CNT0, S_TILES = 36, 38, 40, 41, 42, 43
S_PB, S_PC, S_LDC8, S_HAVE, S_ST, S_MASK, S_WCOFF, S_WBOFF, S_BIAS, S_C, S_CB, S_TMP, S_REC = 44, 46, 48, 49, 50, 52, 54, 55, 56, 58, 60, 62, 64
# qknorm (few SGPRs: every VGPR is taken, so the compiler has nowhere to spill scalars it keeps across the statement): s[72:73] rotary table,
# s[74:77] tokens per sample, text length, 1 / tokens (float), eps -- read once from the LDS constant block behind the tile records (the C++
# preamble writes it: these six dwords, then the LayerNorm weights / biases of q and k, 4 x 128 B) --, s78 previous tile's (m0 << 2 | kind),
# s79 this tile's, s80 IN LDS address of the constant block; the rotary predicates of the four ring slots live in the record registers
# s[64:71], idle during a K loop.  v99 IN carries (wm * 128 + (lane >> 3)) << 8 above the bias offset.
S_CS, S_TOK, S_TEXT, S_INVTOK, S_EPS, S_PREV7, S_CUR7, S_CB0, S_SM = 72, 74, 75, 76, 77, 78, 79, 80, 64
QK_CONST_BYTES, QK_LN_OFF = 64 + 512, 64
A_STRIDE, W_BASE = 65536, 32768
PATCH_BASE, PATCH_WAVE, TABLE_BASE, TABLE_BYTES = 131072, 4096, 131072 + 16384, 2048
GELU_K0, GELU_K1 = 0xc0135761, 0xbdd2d3e7   # -log2(e) * 2 sqrt(2 / pi) and that times 0.044715 (common.h gelu_tanh_fast), as hipcc encodes them


def vr(b, n=1):
    return f"v{b}" if n == 1 else f"v[{b}:{b + n - 1}]"


def ar(b, n):
    return f"a[{b}:{b + n - 1}]"


def wf(buf, i):
    return FRAG + 32 * buf + 4 * i


def af(buf, j):
    return FRAG + 32 * buf + 16 + 4 * j


def vaddr(is_w, g, s):
    return VADDR + (8 if is_w else 0) + 4 * g + s


def ereg(i, j, rq, h):
    """packed pair (columns i*32 + 8rq + 4hi + 2h, +1; row j*32 + fr) of the wave tile"""
    return E0 + (i * 4 + j) * 8 + rq * 2 + h


# ------------------------------------------------------------------------------------------------------------------ the trickle
class Trickle:
    """a linear program pulled a few issue slots at a time.  Items are tuples (text, cost, kind, ...).  Markers: ("mark") remembers how
    many lgkmcnt(0) waits the loop has emitted so far -- placed right behind a group of ds_reads; ("lgkm") may only be passed once the
    loop has emitted a LATER wait, i.e. the reads have returned; "store" instructions are not issued in step 2 (too close to the loop's
    vmcnt(0) of step 3).
    The q/k-norm program adds (vmcnt is ONE in-order counter over loads, LDS-DMA pieces and stores, so the generator can count):
      ([lines], cost, "vmgroup", tag, n)  n vector-memory instructions with their address arithmetic, emitted as a block in a store
                                          step; met in another step the block is PARKED (the program goes on) and leaves first thing in
                                          the next store step
      ("", 0, "fence")                    passes once nothing is parked (registers of parked stores are about to be reused)
      ("", 0, "vmwait", tag)              the data of group `tag`: s_waitcnt vmcnt(number of vector-memory instructions issued since),
                                          nothing when a step-3 wait of the loop already covered it
      ("", 0, "dist", tag, n)             passes once n MFMAs have been emitted since group `tag` left (the wait above then finds the
                                          data there: a waiting wave issues no MFMA either)"""

    def __init__(self, prog):
        self.prog, self.pos, self.syncs, self.mark = prog, 0, 0, -1
        self.young_stores = 0   # stores issued since the last LDS-DMA piece: the loop's vmcnt wait may leave exactly these in flight
        self.vmops, self.vm_done, self.vm_at = 0, 0, {}
        self.mfmas, self.mfma_at = 0, {}
        self.parked = []

    def done(self):
        return self.pos >= len(self.prog) and not self.parked

    def sync(self):
        self.syncs += 1

    def _group(self, emit, item):
        for ln in item[0]:
            emit(ln)
        self.vmops += item[4]
        self.young_stores += item[4]
        self.vm_at[item[3]] = self.vmops - 1
        self.mfma_at[item[3]] = self.mfmas

    def pull(self, emit, step, budget):
        while self.parked and step in STORE_STEPS and budget > 0:
            item = self.parked.pop(0)
            self._group(emit, item)
            budget -= item[1]
        while budget > 0 and self.pos < len(self.prog):
            item = self.prog[self.pos]
            ins, cost, kind = item[0], item[1], item[2]
            if kind == "mark":
                self.mark = self.syncs
                self.pos += 1
                continue
            if kind == "lgkm":
                if self.syncs <= self.mark:
                    return           # the reads behind the mark have not been waited for yet
                self.pos += 1
                continue
            if kind == "vmgroup":
                self.pos += 1
                if step in STORE_STEPS and not self.parked:
                    self._group(emit, item)
                    budget -= cost
                else:
                    self.parked.append(item)
                continue
            if kind == "fence":
                if self.parked:
                    return
                self.pos += 1
                continue
            if kind == "dist":
                if item[3] not in self.mfma_at or self.mfmas - self.mfma_at[item[3]] < item[4]:
                    return
                self.pos += 1
                continue
            if kind == "vmwait":
                if item[3] not in self.vm_at:
                    return           # still parked
                idx = self.vm_at[item[3]]
                if idx >= self.vm_done:
                    n = self.vmops - idx - 1
                    assert n <= 63, "vmcnt field"
                    emit(f"s_waitcnt vmcnt({n})")
                self.pos += 1
                continue
            if kind == "store":
                if step not in STORE_STEPS:
                    return
                self.young_stores += 1
                self.vmops += 1
            emit(ins)
            self.pos += 1
            budget -= cost


def gelu_pair_code(r0, r1):
    """GELU in place on two packed registers (four values), the four chains interleaved"""
    x = [G0 + k for k in range(4)]
    t = [G0 + 4 + k for k in range(4)]
    out = []
    out.append((f"v_lshlrev_b32 {vr(x[0])}, 16, {vr(r0)}", 1))
    out.append((f"v_and_b32 {vr(x[1])}, s{S_MASK}, {vr(r0)}", 1))
    out.append((f"v_lshlrev_b32 {vr(x[2])}, 16, {vr(r1)}", 1))
    out.append((f"v_and_b32 {vr(x[3])}, s{S_MASK}, {vr(r1)}", 1))
    for k in range(4):
        out.append((f"v_mul_f32 {vr(t[k])}, {vr(x[k])}, {vr(x[k])}", 1))
    for k in range(4):
        out.append((f"v_fma_f32 {vr(t[k])}, {vr(t[k])}, s{S_TMP}, {vr(V_K0)}", 1))   # s62 holds k1 during the trickle
    for k in range(4):
        out.append((f"v_mul_f32 {vr(t[k])}, {vr(t[k])}, {vr(x[k])}", 1))
    for k in range(4):
        out.append((f"v_exp_f32 {vr(t[k])}, {vr(t[k])}", 2))
    for k in range(4):
        out.append((f"v_add_f32 {vr(t[k])}, 1.0, {vr(t[k])}", 1))
    for k in range(4):
        out.append((f"v_rcp_f32 {vr(t[k])}, {vr(t[k])}", 2))
    for k in range(4):
        out.append((f"v_mul_f32 {vr(x[k])}, {vr(t[k])}, {vr(x[k])}", 1))
    out.append((f"v_cvt_pk_bf16_f32 {vr(r0)}, {vr(x[0])}, {vr(x[1])}", 1))
    out.append((f"v_cvt_pk_bf16_f32 {vr(r1)}, {vr(x[2])}, {vr(x[3])}", 1))
    return [(i, c, None) for i, c in out]


def trickle_program(epi):
    """units u = (row block j, column half ih) of the wave tile, software-pipelined: GELU(u) | patch writes + read-back(u) | GELU(u + 1) |
    stores(u) | patch writes + read-back(u + 1) | ...  -- the read-back of a unit has a whole GELU block to return in"""
    units = [(j, ih) for j in range(4) for ih in range(2)]
    P = []
    if epi == "gelu":
        P.append((f"s_mov_b32 s{S_TMP}, 0x{GELU_K1:08x}", 0.5, None))

    def gelu(u):
        j, ih = u
        out = []
        if epi == "gelu" and "nogelu" not in ABLATE:
            for i in (2 * ih, 2 * ih + 1):
                for rq in range(4):
                    out += gelu_pair_code(ereg(i, j, rq, 0), ereg(i, j, rq, 1))
        return out

    def patch(u):
        j, ih = u
        out = []
        for i in (2 * ih, 2 * ih + 1):
            for rq in range(4):
                c = (i & 1) * 4 + rq
                out.append((f"v_xor_b32 {vr(G0 + 8)}, {c << 4}, {vr(V_DSW)}", 1, None))
                out.append((f"ds_write_b64 {vr(G0 + 8)}, {vr(ereg(i, j, rq, 0), 2)}", 1, None))
        for k in range(4):
            out.append((f"ds_read_b128 {vr(R0 + 4 * k, 4)}, {vr(V_DSR)} offset:{k * 1024}", 1, None))
        out.append(("", 0, "mark"))
        return out

    def stores(u):
        j, ih = u
        out = [("", 0, "lgkm")]
        # row block j of the previous tile's wave tile: C + j * 4 * (8 rows); the column half is the instruction's immediate
        out.append((f"s_mul_i32 s{S_ST}, s{S_LDC8}, {4 * j}", 0.5, None))
        out.append((f"s_add_u32 s{S_ST}, s{S_PC}, s{S_ST}", 0.5, None))
        out.append((f"s_addc_u32 s{S_ST + 1}, s{S_PC + 1}, 0", 0.5, None))
        for k in range(4):
            if "nostore" not in ABLATE:
                out.append((f"global_store_dwordx4 {vr(V_ST)}, {vr(R0 + 4 * k, 4)}, s[{S_ST}:{S_ST + 1}] offset:{ih * 128}", 1, "store"))
            if k < 3:
                out.append((f"s_add_u32 s{S_ST}, s{S_ST}, s{S_LDC8}", 0.5, None))
                out.append((f"s_addc_u32 s{S_ST + 1}, s{S_ST + 1}, 0", 0.5, None))
        return out

    def spread(work, st):
        """the four stores of the previous unit (each with its address arithmetic) spaced evenly through a block of VALU work: a CU's
        write path takes ~18 B / clk, and a store issued into a full queue holds its wave -- the only wave of its SIMD -- MFMAs included"""
        groups, cur = [], []
        for item in st:
            cur.append(item)
            if item[2] == "store":
                groups.append(cur)
                cur = []
        if cur:
            groups[-1] += cur
        if not work or SPREAD == 0:
            return work + st
        out, n = [], len(work)
        cuts = [n * g // len(groups) for g in range(len(groups))]
        for idx, item in enumerate(work):
            while cuts and idx == cuts[0]:
                out += groups.pop(0)
                cuts.pop(0)
            out.append(item)
        for g in groups:
            out += g
        return out

    P += gelu(units[0]) + patch(units[0])
    for n in range(1, len(units)):
        P += spread(gelu(units[n]), stores(units[n - 1])) + patch(units[n])
    P += stores(units[-1])
    return P


# ------------------------------------------------------------------------------------------------------------------ the q/k-norm trickle
def eblock(i, j):
    return E0 + (i * 4 + j) * 8


QK_UNITS = [(j, ih) for j in range(4) for ih in range(2)]
QK_WQ, QK_BQ, QK_ROW, QK_C16, QK_TMP = G0, G0 + 4, G0 + 8, G0 + 9, V_K0
QK_X = eblock(0, 0)                                             # unit (0, 0)'s registers, free once its patch is written: the row's 8 values
QK_T = eblock(1, 0)                                             # ... and 8 temporaries
QK_SLOT = [eblock(2, 0), eblock(3, 0), eblock(0, 1), eblock(1, 1)]  # rotary ring: units (0, 1) and (1, 0) -- 8 registers (4 cos | 4 sin) per row
DPP_STEPS = ("quad_perm:[1,0,3,2]", "quad_perm:[2,3,0,1]", "row_half_mirror")


def qk_program():
    """LayerNorm(64) + affine + rotary embedding of the previous tile's q or k heads (gemm_epi.h, EPI_BIAS_QKNORM), in the read-back layout
    and in gemm_epi.h's order of operations: lane = (row 8 k + (lane >> 3) of the 32-row unit, octet c16 = lane & 7 of the head's 64
    columns).  Per row: sum (8 in the lane, then the 8 lanes of the row by three DPP adds), mean, centred squares the same way,
    1 / sqrt(var + eps) as the compiler's correctly rounded sqrt and division (the no-scaling paths: var + eps >= 1e-6, 1 / sqrt within
    [1e-4, 1e3]), (d * rstd) * w + b rounded to bf16, the rotary pair on the ROUNDED values for video rows (v_cndmask by the row's
    predicate), packed into the read-back registers, stored.  Rows n = 4 u + k of unit u; rotary values of row n + 4 (two rows ahead
    while only two ring slots are free) are requested when row n is done."""
    X = [QK_X + e for e in range(8)]
    TS, TM, TA, TB, TC, TD, TE, TF = (QK_T + e for e in range(8))
    P = []

    def I(text, cost=1.0):
        P.append((text, cost, None))

    def nop(n):
        P.append((f"s_nop {n}", 0.5, None))

    def slot_of(n):
        return (n & 1) if n < 4 else (n & 3)

    def patch_write(u):
        j, ih = QK_UNITS[u]
        for i in (2 * ih, 2 * ih + 1):
            for rq in range(4):
                c = (i & 1) * 4 + rq
                I(f"v_xor_b32 {vr(QK_TMP)}, {c << 4}, {vr(V_DSW)}")
                I(f"ds_write_b64 {vr(QK_TMP)}, {vr(ereg(i, j, rq, 0), 2)}")

    def read_back(u):
        for k in range(4):
            I(f"ds_read_b128 {vr(R0 + 4 * k, 4)}, {vr(V_DSR)} offset:{k * 1024}")
        P.append(("", 0, "mark"))

    def request(n):
        """position of row n -> rotary predicate (SGPR pair of the slot) and table offset -> two loads"""
        j, ih = QK_UNITS[n >> 2]
        k = n & 3
        sl = slot_of(n)
        c = [QK_SLOT[sl] + e for e in range(8)]
        sm = f"s[{S_SM + 2 * sl}:{S_SM + 2 * sl + 1}]"
        I(f"v_add_u32 {vr(c[7])}, {32 * j + 8 * k}, {vr(QK_ROW)}")                   # m
        I(f"v_cvt_f32_i32 {vr(c[4])}, {vr(c[7])}")
        I(f"v_mul_f32 {vr(c[4])}, s{S_INVTOK}, {vr(c[4])}")
        I(f"v_cvt_i32_f32 {vr(c[4])}, {vr(c[4])}")                                   # sample index (maybe one off)
        I(f"v_mul_lo_u32 {vr(c[4])}, {vr(c[4])}, s{S_TOK}", 2)
        I(f"v_sub_u32 {vr(c[7])}, {vr(c[7])}, {vr(c[4])}")                           # r = m - b * tok
        I(f"v_cmp_gt_i32 vcc, 0, {vr(c[7])}")
        I(f"v_add_u32 {vr(c[5])}, s{S_TOK}, {vr(c[7])}")
        nop(1)
        I(f"v_cndmask_b32 {vr(c[7])}, {vr(c[7])}, {vr(c[5])}, vcc")
        I(f"v_cmp_le_i32 vcc, s{S_TOK}, {vr(c[7])}")
        I(f"v_subrev_u32 {vr(c[5])}, s{S_TOK}, {vr(c[7])}")
        nop(1)
        I(f"v_cndmask_b32 {vr(c[7])}, {vr(c[7])}, {vr(c[5])}, vcc")
        I(f"v_cmp_le_i32 {sm}, s{S_TEXT}, {vr(c[7])}")                               # video row: rotary
        I(f"v_subrev_u32 {vr(c[7])}, s{S_TEXT}, {vr(c[7])}")
        nop(1)
        I(f"v_cndmask_b32 {vr(c[7])}, 0, {vr(c[7])}, {sm}")
        I(f"v_lshl_add_u32 {vr(c[7])}, {vr(c[7])}, 8, {vr(QK_C16)}")                 # 256 B per position + 16 B per octet
        if "qk_noload" in ABLATE:   # timing ablation (wrong results): no rotary loads
            P.append(([], 0, "vmgroup", ("cs", n), 0))
            return
        P.append(([f"global_load_dwordx4 {vr(c[0], 4)}, {vr(c[7])}, s[{S_CS}:{S_CS + 1}]",
                   f"global_load_dwordx4 {vr(c[4], 4)}, {vr(c[7])}, s[{S_CS}:{S_CS + 1}] offset:128"], 2, "vmgroup", ("cs", n), 2))

    def dpp_sum(reg):
        for st in DPP_STEPS:
            nop(1)
            I(f"v_add_f32_dpp {vr(reg)}, {vr(reg)}, {vr(reg)} {st} row_mask:0xf bank_mask:0xf bound_ctrl:1")

    def compute(n):
        if "qk_nocompute" in ABLATE:   # timing ablation (wrong results): the rows go out as they were read back
            P.append(("", 0, "vmwait", ("cs", n)))
            return
        k = n & 3
        sl = slot_of(n)
        c = [QK_SLOT[sl] + e for e in range(8)]
        sm = f"s[{S_SM + 2 * sl}:{S_SM + 2 * sl + 1}]"
        r = [R0 + 4 * k + e for e in range(4)]
        for i in range(4):
            I(f"v_lshlrev_b32 {vr(X[2 * i])}, 16, {vr(r[i])}")
            I(f"v_and_b32 {vr(X[2 * i + 1])}, s{S_MASK}, {vr(r[i])}")
        I(f"v_add_f32 {vr(TS)}, 0, {vr(X[0])}")
        for e in range(1, 8):
            I(f"v_add_f32 {vr(TS)}, {vr(TS)}, {vr(X[e])}")
        dpp_sum(TS)
        I(f"v_mul_f32 {vr(TM)}, 0x3c800000, {vr(TS)}")                               # mean
        for e in range(8):
            I(f"v_sub_f32 {vr(X[e])}, {vr(X[e])}, {vr(TM)}")
        I(f"v_mul_f32 {vr(TS)}, {vr(X[0])}, {vr(X[0])}")
        I(f"v_mul_f32 {vr(TA)}, {vr(X[1])}, {vr(X[1])}")
        I(f"v_add_f32 {vr(TS)}, {vr(TS)}, {vr(TA)}")
        for e in range(2, 8):
            I(f"v_mul_f32 {vr(TA)}, {vr(X[e])}, {vr(X[e])}")
            I(f"v_add_f32 {vr(TS)}, {vr(TA)}, {vr(TS)}")
        dpp_sum(TS)
        I(f"v_mul_f32 {vr(TS)}, 0x3c800000, {vr(TS)}")
        I(f"v_add_f32 {vr(TS)}, s{S_EPS}, {vr(TS)}")                                 # var + eps
        # sqrt, correctly rounded (the compiler's expansion of sqrtf without its denormal-range scaling)
        I(f"v_sqrt_f32 {vr(TA)}, {vr(TS)}", 2)
        nop(0)
        I(f"v_add_u32 {vr(TB)}, -1, {vr(TA)}")
        I(f"v_fma_f32 {vr(TC)}, -{vr(TB)}, {vr(TA)}, {vr(TS)}")
        I(f"v_cmp_ge_f32 vcc, 0, {vr(TC)}")
        I(f"v_add_u32 {vr(TD)}, 1, {vr(TA)}")
        nop(0)
        I(f"v_cndmask_b32 {vr(TB)}, {vr(TA)}, {vr(TB)}, vcc")
        I(f"v_fma_f32 {vr(TC)}, -{vr(TD)}, {vr(TA)}, {vr(TS)}")
        I(f"v_cmp_lt_f32 vcc, 0, {vr(TC)}")
        nop(1)
        I(f"v_cndmask_b32 {vr(TA)}, {vr(TB)}, {vr(TD)}, vcc")                        # s = sqrt(var + eps)
        # 1 / s, correctly rounded (v_div_scale / v_div_fmas / v_div_fixup are identities in this range)
        I(f"v_rcp_f32 {vr(TB)}, {vr(TA)}", 2)
        nop(0)
        I(f"v_fma_f32 {vr(TC)}, -{vr(TA)}, {vr(TB)}, 1.0")
        I(f"v_fmac_f32 {vr(TB)}, {vr(TC)}, {vr(TB)}")                                # r1
        I(f"v_mul_f32 {vr(TC)}, 1.0, {vr(TB)}")                                      # q0
        I(f"v_fma_f32 {vr(TD)}, -{vr(TA)}, {vr(TC)}, 1.0")
        I(f"v_fmac_f32 {vr(TC)}, {vr(TD)}, {vr(TB)}")                                # q1
        I(f"v_fma_f32 {vr(TD)}, -{vr(TA)}, {vr(TC)}, 1.0")
        I(f"v_fma_f32 {vr(TM)}, {vr(TD)}, {vr(TB)}, {vr(TC)}")                       # rstd
        P.append(("", 0, "dist", ("cs", n), QK_DIST))
        P.append(("", 0, "vmwait", ("cs", n)))
        for i in range(4):
            x0, x1 = X[2 * i], X[2 * i + 1]
            I(f"v_mul_f32 {vr(x0)}, {vr(x0)}, {vr(TM)}")
            I(f"v_mul_f32 {vr(x1)}, {vr(x1)}, {vr(TM)}")
            I(f"v_lshlrev_b32 {vr(TA)}, 16, {vr(QK_WQ + i)}")
            I(f"v_and_b32 {vr(TB)}, s{S_MASK}, {vr(QK_WQ + i)}")
            I(f"v_mul_f32 {vr(x0)}, {vr(x0)}, {vr(TA)}")
            I(f"v_mul_f32 {vr(x1)}, {vr(x1)}, {vr(TB)}")
            I(f"v_lshlrev_b32 {vr(TA)}, 16, {vr(QK_BQ + i)}")
            I(f"v_and_b32 {vr(TB)}, s{S_MASK}, {vr(QK_BQ + i)}")
            I(f"v_add_f32 {vr(x0)}, {vr(x0)}, {vr(TA)}")
            I(f"v_add_f32 {vr(x1)}, {vr(x1)}, {vr(TB)}")
            I(f"v_cvt_pk_bf16_f32 {vr(TC)}, {vr(x0)}, {vr(x1)}")                     # the normalised pair, rounded
            I(f"v_lshlrev_b32 {vr(TA)}, 16, {vr(TC)}")
            I(f"v_and_b32 {vr(TB)}, s{S_MASK}, {vr(TC)}")
            I(f"v_mul_f32 {vr(TD)}, {vr(c[i])}, {vr(TA)}")
            I(f"v_mul_f32 {vr(TE)}, {vr(c[4 + i])}, {vr(TB)}")
            I(f"v_sub_f32 {vr(TD)}, {vr(TD)}, {vr(TE)}")                             # x0 cos - x1 sin
            I(f"v_mul_f32 {vr(TE)}, {vr(c[i])}, {vr(TB)}")
            I(f"v_mul_f32 {vr(TF)}, {vr(c[4 + i])}, {vr(TA)}")
            I(f"v_add_f32 {vr(TE)}, {vr(TE)}, {vr(TF)}")                             # x1 cos + x0 sin
            I(f"v_cvt_pk_bf16_f32 {vr(TD)}, {vr(TD)}, {vr(TE)}")
            I(f"v_cndmask_b32 {vr(r[i])}, {vr(TC)}, {vr(TD)}, {sm}")

    def store(n):
        j, ih = QK_UNITS[n >> 2]
        k = n & 3
        P.append(([f"s_mul_i32 s{S_ST}, s{S_LDC8}, {4 * j + k}", f"s_add_u32 s{S_ST}, s{S_PC}, s{S_ST}", f"s_addc_u32 s{S_ST + 1}, s{S_PC + 1}, 0",
                   f"global_store_dwordx4 {vr(V_ST)}, {vr(R0 + 4 * k, 4)}, s[{S_ST}:{S_ST + 1}] offset:{ih * 128}"], 2.5, "vmgroup", ("st", n), 1))

    # lane constants, LayerNorm parameters of the head kind (q: 0, k: 1) of the previous tile from the LDS constant block
    I(f"v_mbcnt_lo_u32_b32 {vr(QK_TMP)}, -1, 0")
    I(f"v_mbcnt_hi_u32_b32 {vr(QK_TMP)}, -1, {vr(QK_TMP)}")
    I(f"v_and_b32 {vr(QK_C16)}, 7, {vr(QK_TMP)}")
    I(f"v_lshlrev_b32 {vr(QK_C16)}, 4, {vr(QK_C16)}")
    I(f"v_lshrrev_b32 {vr(QK_ROW)}, 8, {vr(V_BOFF)}")                                # wm * 128 + (lane >> 3)
    I(f"s_lshr_b32 s{S_TMP}, s{S_PREV7}, 2", 0.5)
    I(f"v_add_u32 {vr(QK_ROW)}, s{S_TMP}, {vr(QK_ROW)}")                             # first row of the lane in the wave tile, as a token row
    I(f"s_and_b32 s{S_TMP}, s{S_PREV7}, 1", 0.5)
    I(f"s_lshl_b32 s{S_TMP}, s{S_TMP}, 7", 0.5)
    I(f"s_add_u32 s{S_TMP}, s{S_TMP}, s{S_CB0}", 0.5)
    I(f"v_add_u32 {vr(QK_TMP)}, s{S_TMP}, {vr(QK_C16)}")
    I(f"ds_read_b128 {vr(QK_WQ, 4)}, {vr(QK_TMP)} offset:{QK_LN_OFF}")
    I(f"ds_read_b128 {vr(QK_BQ, 4)}, {vr(QK_TMP)} offset:{QK_LN_OFF + 256}")
    patch_write(0)
    read_back(0)
    patch_write(1)
    request(0)
    request(1)
    P.append(("", 0, "lgkm"))
    for n in range(32):
        u, k = n >> 2, n & 3
        if k == 0 and u > 0:
            P.append(("", 0, "fence"))        # the stores of unit u - 1 have left: its read-back registers are free
            read_back(u)
            if u + 1 < 8:
                patch_write(u + 1)
            if u == 1:                        # unit 2's registers are free now: ring slots 2, 3
                request(6)
                request(7)
            P.append(("", 0, "lgkm"))
        compute(n)
        nxt = n + 2 if n < 4 else n + 4
        if nxt < 32 and nxt not in (6, 7):
            request(nxt)
        store(n)
    return P


# ------------------------------------------------------------------------------------------------------------------ the K loop
def ktile(emit, g, first=False, dma_w=True, dma_a=True, last=False, trick=None):
    for s in range(4):
        cur, nxt = s & 1, (s & 1) ^ 1
        if s == 3 and not last:
            young = trick.young_stores if (trick and COUNTED) else 0
            emit(f"s_waitcnt vmcnt({young}) lgkmcnt(0)")
            emit("s_barrier")
            if trick:
                trick.vm_done = trick.vmops - trick.young_stores   # everything but the young stores has landed
                trick.young_stores = 0
        else:
            emit("s_waitcnt lgkmcnt(0)")
        if trick:
            trick.sync()
        for k in range(16):
            i, j = k >> 2, k & 3
            acc = ar(64 * i + 16 * j, 16)
            c = "0" if (first and s == 0) else acc
            emit(f"v_mfma_f32_32x32x16_bf16 {acc}, {vr(wf(cur, i), 4)}, {vr(af(cur, j), 4)}, {c}")
            if trick:
                trick.mfmas += 1
            if k < 8 and not (last and s == 3):
                gs, ss = (g, s + 1) if s < 3 else (g ^ 1, 0)
                if k < 4:
                    emit(f"ds_read_b128 {vr(wf(nxt, k), 4)}, {vr(vaddr(True, gs, ss))} offset:{k * 4096}")
                else:
                    emit(f"ds_read_b128 {vr(af(nxt, k - 4), 4)}, {vr(vaddr(False, gs, ss))} offset:{(k - 4) * 4096}")
            p = k >> 1
            if s == 0 and dma_w:
                if k & 1 == 0:
                    emit(f"s_add_u32 m0, s{S_M0W}, {(g ^ 1) * 65536 + 32768 + p * 4096}")
                else:
                    emit(f"global_load_lds_dwordx4 {vr(VOFF + 8 + p)}, s[{S_W}:{S_W + 1}]")
                    if trick:
                        trick.vmops += 1
            if s == 3 and dma_a:
                if k & 1 == 0:
                    emit(f"s_add_u32 m0, s{S_M0W}, {g * A_STRIDE + p * 4096}")
                else:
                    emit(f"global_load_lds_dwordx4 {vr(VOFF + p)}, s[{S_A}:{S_A + 1}]")
                    if trick:
                        trick.vmops += 1
            if trick and not trick.done():
                trick.pull(emit, s, trick.tps if hasattr(trick, 'tps') else TPS)
        if s == 0 and dma_w:
            emit(f"s_add_u32 s{S_W}, s{S_W}, 128")
            emit(f"s_addc_u32 s{S_W + 1}, s{S_W + 1}, 0")
        if s == 3 and dma_a:
            emit(f"s_add_u32 s{S_A}, s{S_A}, 128")
            emit(f"s_addc_u32 s{S_A + 1}, s{S_A + 1}, 0")


def drain(emit):
    """accumulators of the previous tile + bias -> E.  Bias (packed bf16, 4 consecutive columns per (i, rq)) sits in v[0:31]; v[32:47]
    take the accumulator block, v[48:63] the 16 fp32 bias values of the current i."""
    T, B = 32, 48
    for i in range(4):
        for rq in range(4):
            q = i * 4 + rq
            emit(f"v_lshlrev_b32 {vr(B + 4 * rq)}, 16, {vr(2 * q)}")
            emit(f"v_and_b32 {vr(B + 4 * rq + 1)}, s{S_MASK}, {vr(2 * q)}")
            emit(f"v_lshlrev_b32 {vr(B + 4 * rq + 2)}, 16, {vr(2 * q + 1)}")
            emit(f"v_and_b32 {vr(B + 4 * rq + 3)}, s{S_MASK}, {vr(2 * q + 1)}")
        for j in range(4):
            for e in range(16):
                emit(f"v_accvgpr_read_b32 {vr(T + e)}, a{64 * i + 16 * j + e}")
            for e in range(16):
                emit(f"v_add_f32 {vr(T + e)}, {vr(T + e)}, {vr(B + e)}")
            for rq in range(4):
                for h in range(2):
                    emit(f"v_cvt_pk_bf16_f32 {vr(ereg(i, j, rq, h))}, {vr(T + 4 * rq + 2 * h)}, {vr(T + 4 * rq + 2 * h + 1)}")


def frag_reads(emit):
    for n in range(8):
        if n < 4:
            emit(f"ds_read_b128 {vr(wf(0, n), 4)}, {vr(vaddr(True, 0, 0))} offset:{n * 4096}")
        else:
            emit(f"ds_read_b128 {vr(af(0, n - 4), 4)}, {vr(vaddr(False, 0, 0))} offset:{(n - 4) * 4096}")


def gen(epi):
    L = []

    def emit(ln):
        if ln:
            L.append(ln)

    emit(f"; ---- gemm_g4t<{epi}>: persistent tile walk with trickled epilogue (generated by gen_gemm_g4t.py; do not edit)")
    emit(f"s_mov_b32 s{S_HAVE}, 0")
    emit(f"s_mov_b32 s{S_MASK}, 0xffff0000")
    emit(f"v_mov_b32 {vr(V_K0)}, 0x{GELU_K0:08x}")
    if epi == "qknorm":
        emit(f"v_mov_b32 {vr(R0 + 8)}, s{S_CB0}")
        emit(f"ds_read_b128 {vr(R0, 4)}, {vr(R0 + 8)}")
        emit(f"ds_read_b64 {vr(R0 + 4, 2)}, {vr(R0 + 8)} offset:16")
        emit("s_waitcnt lgkmcnt(0)")
        for k in range(6):
            emit(f"v_readfirstlane_b32 s{S_CS + k}, {vr(R0 + k)}")
        emit(f"s_mov_b32 s{S_CUR7}, 0")
    emit("L_t_tile_%=:")
    # 1. record -> SGPRs
    emit(f"ds_read_b128 {vr(R0, 4)}, {vr(V_TAB)}")
    emit(f"ds_read_b128 {vr(R0 + 4, 4)}, {vr(V_TAB)} offset:16")
    emit("s_waitcnt lgkmcnt(0)")
    for k in range(7):
        emit(f"v_readfirstlane_b32 s{S_REC + k}, {vr(R0 + k)}")
    if epi == "qknorm":
        emit(f"s_mov_b32 s{S_PREV7}, s{S_CUR7}")
        emit(f"v_readfirstlane_b32 s{S_CUR7}, {vr(R0 + 7)}")
    emit(f"v_add_u32 {vr(V_TAB)}, 32, {vr(V_TAB)}")
    emit(f"s_mov_b64 s[{S_A}:{S_A + 1}], s[{S_REC}:{S_REC + 1}]")
    emit(f"s_mov_b64 s[{S_W}:{S_W + 1}], s[{S_REC + 2}:{S_REC + 3}]")
    emit(f"s_add_u32 s{S_C}, s{S_REC + 4}, s{S_WCOFF}")
    emit(f"s_addc_u32 s{S_C + 1}, s{S_REC + 5}, 0")
    emit(f"s_add_u32 s{S_CB}, s{S_BIAS}, s{S_REC + 6}")
    emit(f"s_addc_u32 s{S_CB + 1}, s{S_BIAS + 1}, 0")
    emit(f"s_add_u32 s{S_CB}, s{S_CB}, s{S_WBOFF}")
    emit(f"s_addc_u32 s{S_CB + 1}, s{S_CB + 1}, 0")
    emit(f"s_mov_b32 s{S_CNT}, s{S_CNT0}")
    # 2. bias of the previous tile
    emit(f"s_cmp_eq_u32 s{S_HAVE}, 0")
    emit("s_cbranch_scc1 L_t_nobias_%=")
    boff = V_BOFF
    if epi == "qknorm":
        boff = R0 + 8
        emit(f"v_and_b32 {vr(boff)}, 0xff, {vr(V_BOFF)}")
    for i in range(4):
        for rq in range(4):
            q = i * 4 + rq
            emit(f"global_load_dwordx2 {vr(2 * q, 2)}, {vr(boff)}, s[{S_PB}:{S_PB + 1}] offset:{i * 64 + rq * 16}")
    emit("L_t_nobias_%=:")
    # 3. barrier + prologue DMA
    emit("s_barrier")
    for p in range(8):
        emit(f"s_add_u32 m0, s{S_M0W}, {p * 4096}")
        emit("s_nop 0")
        emit(f"global_load_lds_dwordx4 {vr(VOFF + p)}, s[{S_A}:{S_A + 1}]")
    for p in range(8):
        emit(f"s_add_u32 m0, s{S_M0W}, {W_BASE + p * 4096}")
        emit("s_nop 0")
        emit(f"global_load_lds_dwordx4 {vr(VOFF + 8 + p)}, s[{S_W}:{S_W + 1}]")
    emit(f"s_add_u32 s{S_A}, s{S_A}, 128")
    emit(f"s_addc_u32 s{S_A + 1}, s{S_A + 1}, 0")
    emit(f"s_add_u32 s{S_W}, s{S_W}, 128")
    emit(f"s_addc_u32 s{S_W + 1}, s{S_W + 1}, 0")
    for p in range(8):
        emit(f"s_add_u32 m0, s{S_M0W}, {A_STRIDE + p * 4096}")
        emit("s_nop 0")
        emit(f"global_load_lds_dwordx4 {vr(VOFF + p)}, s[{S_A}:{S_A + 1}]")
    emit(f"s_add_u32 s{S_A}, s{S_A}, 128")
    emit(f"s_addc_u32 s{S_A + 1}, s{S_A + 1}, 0")
    emit(f"s_cmp_eq_u32 s{S_HAVE}, 0")
    emit("s_cbranch_scc1 L_t_first_%=")
    # 4. drain under the DMA latency
    emit("s_waitcnt vmcnt(24)")
    if "nodrain" not in ABLATE:
        drain(emit)
    emit("s_waitcnt vmcnt(8)")
    emit("s_barrier")
    frag_reads(emit)
    # 5. K-tiles 0 .. TK with the trickle
    def unrolled(prog, tps):
        trick = Trickle(prog if "notrickle" not in ABLATE else [])
        trick.tps = tps
        ktile(emit, 0, first=True, trick=trick)
        tk = 0
        while not trick.done() or tk % 2:
            tk += 1
            ktile(emit, tk & 1, trick=trick)
            assert tk < 46, "trickle does not fit"
        emit(f"s_sub_u32 s{S_CNT}, s{S_CNT}, {tk // 2}")
        emit("s_branch L_t_loop_%=")
        return tk

    if epi == "qknorm":   # the previous tile: q or k heads (kind 0 / 1: the q/k-norm program) or v heads (kind 2: bias only)
        emit(f"s_bitcmp1_b32 s{S_PREV7}, 1")
        emit("s_cbranch_scc1 L_t_plain_%=")
        tk = unrolled(qk_program(), TPS_QK)
        emit("L_t_plain_%=:")
        tk = max(tk, unrolled(trickle_program("bias"), TPS))
    else:
        tk = unrolled(trickle_program(epi), TPS)
    emit("L_t_first_%=:")
    emit("s_waitcnt vmcnt(8)")
    emit("s_barrier")
    frag_reads(emit)
    ktile(emit, 0, first=True)
    # 6. generic loop + tail
    emit("L_t_loop_%=:")
    emit(f"s_cmp_eq_u32 s{S_CNT}, 0")
    emit("s_cbranch_scc1 L_t_tail_%=")
    ktile(emit, 1)
    ktile(emit, 0)
    emit(f"s_sub_u32 s{S_CNT}, s{S_CNT}, 1")
    emit("s_branch L_t_loop_%=")
    emit("L_t_tail_%=:")
    ktile(emit, 1)
    ktile(emit, 0, dma_a=False, dma_w=True)
    ktile(emit, 1, dma_w=False, dma_a=False, last=True)
    emit(f"s_mov_b64 s[{S_PB}:{S_PB + 1}], s[{S_CB}:{S_CB + 1}]")
    emit(f"s_mov_b64 s[{S_PC}:{S_PC + 1}], s[{S_C}:{S_C + 1}]")
    emit(f"s_mov_b32 s{S_HAVE}, 1")
    emit(f"s_sub_u32 s{S_TILES}, s{S_TILES}, 1")
    emit(f"s_cmp_eq_u32 s{S_TILES}, 0")
    emit("s_cbranch_scc0 L_t_tile_%=")
    emit("s_waitcnt vmcnt(0)")
    emit("s_nop 15")
    emit("s_nop 15")
    return L, tk


def main():
    here = os.environ.get("S2V_GEN_OUT") or os.path.dirname(os.path.abspath(__file__))  # S2V_GEN_OUT: tests/test_host_cpu.py regenerates into a scratch directory
    tks = {}
    for epi in EPI:
        body, tk = gen(epi)
        tks[epi] = tk
        with open(os.path.join(here, f"gemm_g4t_body_{epi}.inc"), "w") as f:
            for ln in body:
                f.write('"' + ln + '\\n\\t"\n')
    vclob = [f"v{r}" for r in range(0, 64)] + [f"v{r}" for r in range(E0, 256) if r != V_TAB]
    sclob = [f"s{r}" for r in (S_CNT, S_HAVE, S_ST, S_ST + 1, S_MASK, S_C, S_C + 1, S_CB, S_CB + 1, S_TMP)] + [f"s{S_REC + k}" for k in range(8)] + [f"s{S_PB + k}" for k in range(4)]
    sclob_qk = [f"s{S_CS + k}" for k in range(8)]
    with open(os.path.join(here, "gemm_g4t_regs.h"), "w") as f:
        f.write("// generated by gen_gemm_g4t.py: register constraints, LDS map and unroll depth of gemm_g4t\n#pragma once\n")
        f.write(f"#define G4T_LDS_BYTES {TABLE_BASE + TABLE_BYTES}\n#define G4T_PATCH_BASE {PATCH_BASE}\n#define G4T_PATCH_WAVE {PATCH_WAVE}\n#define G4T_TABLE_BASE {TABLE_BASE}\n#define G4T_TABLE_RECORDS {TABLE_BYTES // 32}\n")
        f.write(f"#define G4T_A_STRIDE {A_STRIDE}\n#define G4T_W_BASE {W_BASE}\n#define G4T_W_STRIDE 65536\n")
        for epi, tk in tks.items():
            f.write(f"#define G4T_TK_{epi.upper()} {tk}  // K-tiles (after K-tile 0) that carry the trickle: nT >= TK + 4\n")
        for k in range(8):
            f.write(f'#define G4T_ACC{k} "{{a[{32 * k}:{32 * k + 31}]}}"\n')
        f.write(f'#define G4T_VADDR "{{v[{VADDR}:{VADDR + 15}]}}"\n#define G4T_VOFF "{{v[{VOFF}:{VOFF + 15}]}}"\n#define G4T_VLANE "{{v[{V_ST}:{V_BOFF}]}}"\n#define G4T_VTAB "{{v{V_TAB}}}"\n')
        f.write(f'#define G4T_SIN0 "{{s{S_M0W}}}"\n#define G4T_SIN1 "{{s[{S_CNT0}:{S_TILES}]}}"\n#define G4T_SIN2 "{{s{S_LDC8}}}"\n#define G4T_SIN3 "{{s[{S_WCOFF}:{S_WBOFF}]}}"\n#define G4T_SIN4 "{{s[{S_BIAS}:{S_BIAS + 1}]}}"\n#define G4T_PTR "{{s[{S_A}:{S_A + 3}]}}"\n')
        f.write("#define G4T_CLOBBERS " + ", ".join(f'"{c}"' for c in vclob + sclob) + ', "vcc", "scc", "m0", "memory"\n')
        f.write(f'#define G4T_QK_CB "{{s{S_CB0}}}"\n#define G4T_QK_CONST_BASE {TABLE_BASE + TABLE_BYTES}\n#define G4T_QK_CONST_BYTES {QK_CONST_BYTES}\n#define G4T_QK_LN_OFF {QK_LN_OFF}\n')
        f.write("#define G4T_QK_CLOBBERS " + ", ".join(f'"{c}"' for c in sclob_qk) + "\n")
    if "-v" in sys.argv:
        for epi in EPI:
            print(epi, "TK", tks[epi], "trickle instructions", len(trickle_program(epi)))


if __name__ == "__main__":
    main()
