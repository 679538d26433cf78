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

Hazards handled by hand (the assembler inserts nothing): v_exp / v_rcp results are never consumed by the next instruction (gfx950 trans-use
hazard: the two chains of a packed pair are interleaved); SGPRs written by v_readfirstlane reach memory instructions only through s_mov /
s_add; MFMA results are read (v_accvgpr_read) a barrier and 24 DMA issues after the last MFMA.

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

EPI = {"bias": 0, "gelu": 1}
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

FRAG, VADDR, VOFF = 0, 64, 80
V_ST, V_DSW, V_DSR, V_BOFF, E0, R0, G0, V_TAB, V_K0 = 96, 97, 98, 99, 100, 228, 244, 254, 255
S_A, S_W, S_M0W, S_CNT, S_CNT0, S_TILES = 36, 38, 40, 41, 42, 43
S_PB, S_PC, S_LDC8, S_HAVE, S_ST, S_MASK, S_WCOFF, S_WBOFF, S_BIAS, S_C, S_CB, S_TMP, S_REC = 44, 46, 48, 49, 50, 52, 54, 55, 56, 58, 60, 62, 64
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
    """a linear program pulled a few issue slots at a time.  Markers: ("mark") remembers how many lgkmcnt(0) waits the loop has emitted
    so far -- placed right behind a group of ds_reads; ("lgkm") may only be passed once the loop has emitted a LATER wait, i.e. the
    reads have returned; "store" instructions are not issued in step 2 (too close to the loop's vmcnt(0) of step 3)."""

    def __init__(self, prog):
        self.prog, self.pos, self.syncs, self.mark = prog, 0, 0, -1
        self.young_stores = 0   # stores issued since the last LDS-DMA piece: the loop's vmcnt wait may leave exactly these in flight

    def done(self):
        return self.pos >= len(self.prog)

    def sync(self):
        self.syncs += 1

    def pull(self, emit, step, budget):
        while budget > 0 and not self.done():
            ins, cost, kind = self.prog[self.pos]
            if kind == "mark":
                self.mark = self.syncs
                self.pos += 1
                continue
            if kind == "lgkm":
                if self.syncs <= self.mark:
                    return           # the reads behind the mark have not been waited for yet
                self.pos += 1
                continue
            if kind == "store":
                if step not in STORE_STEPS:
                    return
                self.young_stores += 1
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


# ------------------------------------------------------------------------------------------------------------------ the K loop
def ktile(emit, g, first=False, dma_w=True, dma_a=True, last=False, trick=None):
    for s in range(4):
        cur, nxt = s & 1, (s & 1) ^ 1
        if s == 3 and not last:
            young = trick.young_stores if (trick and COUNTED) else 0
            emit(f"s_waitcnt vmcnt({young}) lgkmcnt(0)")
            emit("s_barrier")
            if trick:
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
            if s == 3 and dma_a:
                if k & 1 == 0:
                    emit(f"s_add_u32 m0, s{S_M0W}, {g * A_STRIDE + p * 4096}")
                else:
                    emit(f"global_load_lds_dwordx4 {vr(VOFF + p)}, s[{S_A}:{S_A + 1}]")
            if trick and not trick.done():
                trick.pull(emit, s, TPS)
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
    emit("L_t_tile_%=:")
    # 1. record -> SGPRs
    emit(f"ds_read_b128 {vr(R0, 4)}, {vr(V_TAB)}")
    emit(f"ds_read_b128 {vr(R0 + 4, 4)}, {vr(V_TAB)} offset:16")
    emit("s_waitcnt lgkmcnt(0)")
    for k in range(7):
        emit(f"v_readfirstlane_b32 s{S_REC + k}, {vr(R0 + k)}")
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
    for i in range(4):
        for rq in range(4):
            q = i * 4 + rq
            emit(f"global_load_dwordx2 {vr(2 * q, 2)}, {vr(V_BOFF)}, s[{S_PB}:{S_PB + 1}] offset:{i * 64 + rq * 16}")
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
    trick = Trickle(trickle_program(epi) if "notrickle" not in ABLATE else [])
    ktile(emit, 0, first=True, trick=trick)
    tk = 0
    while not trick.done() or tk % 2:
        tk += 1
        ktile(emit, tk & 1, trick=trick)
        assert tk < 40, "trickle does not fit"
    emit(f"s_sub_u32 s{S_CNT}, s{S_CNT}, {tk // 2}")
    emit("s_branch L_t_loop_%=")
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
    here = os.path.dirname(os.path.abspath(__file__))
    tks = {}
    for epi in EPI:
        body, tk = gen(epi)
        tks[epi] = tk
        with open(os.path.join(here, f"gemm_g4t_body_{epi}.inc"), "w") as f:
            for ln in body:
                f.write('"' + ln + '\\n\\t"\n')
    vclob = [f"v{r}" for r in range(0, 64)] + [f"v{r}" for r in range(E0, 256) if r != V_TAB]
    sclob = [f"s{r}" for r in (S_CNT, S_HAVE, S_ST, S_ST + 1, S_MASK, S_C, S_C + 1, S_CB, S_CB + 1, S_TMP)] + [f"s{S_REC + k}" for k in range(8)] + [f"s{S_PB + k}" for k in range(4)]
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
    if "-v" in sys.argv:
        for epi in EPI:
            print(epi, "TK", tks[epi], "trickle instructions", len(trickle_program(epi)))


if __name__ == "__main__":
    main()
