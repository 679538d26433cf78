#!/usr/bin/env python3
"""Generator of the K loop of gemm_g4f (csrc/gemm_g4f.hip): the four-wave 256 x 256 loop of gen_gemm_g4.py on e4m3 operands and
v_mfma_scale_f32_32x32x64_f8f6f4 (BASELINE configs[4]: fp8 weights; no reference code -- the reference has no fp8 path, parity unpinned).
Writes gemm_g4f_body_a3.inc (unit block scales: per-token / per-channel scales are applied by the epilogue), gemm_g4f_body_mx.inc (A is an
MX image: one E8M0 scale per row and 32 elements, GemmArgs::mx_a_s) and gemm_g4f_regs.h.  Run by build.py when stale; outputs committed.

A K-tile is 128 BYTES of K as in the bf16 loop (same rows, same swizzle, same 16 LDS-DMA pieces per wave), i.e. 128 e4m3 elements, and
2048 matrix-pipe cycles: TWO steps of 16 MFMA (64 cycles each: twice the flops of the bf16 instruction).  A lane's 32-byte operand of step
s is the two 16-byte chunks 4 s + hi and 4 s + 2 + hi of its row (gemm.hip, gemm_bf16_pp64<.., FP8>: the instruction's logical K order
makes the memory-contiguous 32-element blocks 2 s and 2 s + 1 its two scale blocks, lane (row, hi) supplying the scale of block 2 s + hi).
Per step: 16 MFMA, 16 ds_read_b128 (the next step's eight fragments), 8 LDS-DMA pieces.

"a3": the schedule of gen_gemm_g4.py's product loop -- three A stages, A pieces of K-tile t+2 in step 0, vmcnt(8) + barrier at step 1, W
pieces of K-tile t+2 behind the barrier; LDS [A0 | A1 | A2 | W0 | W1] x 32 KiB.
"mx": the same loop; the block scales do not pass through LDS (the three-stage map leaves no room, and a first two-stage form with the
scales staged beside the tiles ran at 3.0-3.3 k cycles per K-tile): the scale array is laid out K-tile major with the rows of a 128-row half
permuted (kernels.h GemmArgs::mx_a_s) so that the four dwords a lane needs -- rows j * 32 + fr of its four A fragments -- are 16
consecutive bytes: ONE global_load_dwordx4 per wave and K-tile, requested a K-tile ahead in front of the A pieces (the vmcnt(8) of step 1
covers it), shifted by 8 hi at the start of its K-tile so that byte 0 / byte 2 (op_sel_hi) are the blocks 2 s + hi of steps 0 / 1.
Registers: a[0:255] acc[i][j] at 64 i + 16 j; v[0:127] fragments [buffer][W 0-3 | A 0-3] x 8; v[128:143] IN fragment addresses [A | W][stage
parity][2 s + kk]; v[144:159] IN staging offsets [A | W][piece]; v[160:163] stage-0 A addresses (a3); v164 IN 8 hi; v165 IN byte offset of the lane's four scale dwords
inside a K-tile's scale row (mx); v[168:175] block scales [K-tile parity][j]; v176 unit scales;
s[36:37] / s[38:39] A / W source; s40 IN LDS address of the wave's piece 0; s41 IN pairs of K-tiles in the loop = (nT - 4) / 2; s42 / s43 / s48
A stage rotation (a3); s[44:45] IN scale source (mx), s47 IN bytes between the scale rows of consecutive K-tiles (mx); nT = K / 128 even, >= 4.
"""
import os

FRAG, VADDR, VOFF, ABASE, V_SH, V_SOFF, V_SRD, V_SB, V_UNIT = 0, 128, 144, 160, 164, 165, 166, 168, 176
S_A, S_W, S_M0W, S_CNT, S_ANEXT, S_ADMA, S_AM0, S_SC, S_SM0 = 36, 38, 40, 41, 42, 43, 48, 44, 46


def vr(b, n=1):
    return f"v{b}" if n == 1 else f"v[{b}:{b + n - 1}]"


def ar(b, n):
    return f"a[{b}:{b + n - 1}]"


def wf(buf, i):
    return FRAG + 64 * buf + 8 * i


def af(buf, j):
    return FRAG + 64 * buf + 32 + 8 * j


def vaddr(is_w, g, x):
    return VADDR + (8 if is_w else 0) + 4 * g + x


class Map:
    def __init__(self, mx):
        self.mx = mx
        self.A_STRIDE, self.W_BASE, self.W_STRIDE, self.S_BASE, self.LDS = 32768, 98304, 32768, 0, 163840


def mfma(emit, M, t_par, s, k, first):
    i, j = k >> 2, k & 3
    acc = ar(64 * i + 16 * j, 16)
    c = "0" if (first and s == 0) else acc
    sb = vr(V_SB + 4 * t_par + j) if M.mx else vr(V_UNIT)
    hi = 1 if (M.mx and s == 1) else 0
    emit(f"v_mfma_scale_f32_32x32x64_f8f6f4 {acc}, {vr(wf(s, i), 8)}, {vr(af(s, j), 8)}, {c}, {vr(V_UNIT)}, {sb} op_sel_hi:[0,{hi},0]")


def frag_read(emit, buf, n, kk, g, s):
    """chunk kk of fragment n (0-3 W, 4-7 A) of step s of the K-tile whose fragment-address set is g -> buffer buf"""
    if n < 4:
        emit(f"ds_read_b128 {vr(wf(buf, n) + 4 * kk, 4)}, {vr(vaddr(True, g, 2 * s + kk))} offset:{n * 4096}")
    else:
        emit(f"ds_read_b128 {vr(af(buf, n - 4) + 4 * kk, 4)}, {vr(vaddr(False, g, 2 * s + kk))} offset:{(n - 4) * 4096}")


def ktile(emit, M, g, first=False, dma=True, last=False):
    """K-tile t (parity g).  dma: it stages K-tile t+2 (A pieces in step 0, W pieces in step 1 behind the barrier).  mx: the lane's four
    scale dwords of K-tile t+1 are requested (ONE global_load_dwordx4: the scale array is laid out for it, kernels.h) ahead of the A pieces,
    so the vmcnt(8) of step 1 covers them; they are shifted by 8 hi at the start of K-tile t+1."""
    mx = M.mx
    for s in range(2):
        if s == 1 and not last:
            emit(f"s_waitcnt vmcnt({8 if dma else 0}) lgkmcnt(0)")
            emit("s_barrier")
        else:
            emit("s_waitcnt lgkmcnt(0)")
        if mx and s == 0:  # byte 0 / 2 = blocks hi / 2 + hi of the K-tile
            for j in range(4):
                emit(f"v_lshrrev_b32 {vr(V_SB + 4 * g + j)}, {vr(V_SH)}, {vr(V_SB + 4 * g + j)}")
            emit("s_nop 1")  # VALU result -> scale operand of the next MFMA
        if s == 0 and dma:
            emit(f"s_add_u32 s{S_AM0}, s{S_M0W}, s{S_ADMA}")
        for k in range(16):
            mfma(emit, M, g, s, k, first)
            n, kk = k >> 1, k & 1
            if s == 0:
                frag_read(emit, 1, n, kk, g, 1)              # this K-tile's step 1
            elif not last:
                frag_read(emit, 0, n, kk, g ^ 1, 0)          # the next K-tile's step 0 (behind the barrier)
            p = k >> 1
            if mx and s == 0 and k == 0 and not last:        # scales of K-tile t+1, older than every A piece of this step
                emit(f"global_load_dwordx4 {vr(V_SB + 4 * (g ^ 1), 4)}, {vr(V_SOFF)}, s[{S_SC}:{S_SC + 1}]")
            if s == 0 and dma:   # A piece p of K-tile t+2 -> A stage (t+2) % 3
                emit(f"s_add_u32 m0, s{S_AM0}, {p * 4096}" if k & 1 == 0 else f"global_load_lds_dwordx4 {vr(VOFF + p)}, s[{S_A}:{S_A + 1}]")
            if s == 0 and not last and k >= 12:  # A fragment addresses of K-tile t+1 (set g^1)
                emit(f"v_add_u32 {vr(vaddr(False, g ^ 1, k - 12))}, s{S_ANEXT}, {vr(ABASE + k - 12)}")
            if s == 1 and dma:   # W piece p of K-tile t+2 -> W stage g
                emit(f"s_add_u32 m0, s{S_M0W}, {M.W_BASE + g * M.W_STRIDE + p * 4096}" if k & 1 == 0 else f"global_load_lds_dwordx4 {vr(VOFF + 8 + p)}, s[{S_W}:{S_W + 1}]")
        if s == 0 and dma:
            emit(f"s_add_u32 s{S_A}, s{S_A}, 128")
            emit(f"s_addc_u32 s{S_A + 1}, s{S_A + 1}, 0")
        if s == 0 and not last:
            emit(f"s_mov_b32 s{S_ANEXT}, s{S_ADMA}")
            emit(f"s_add_u32 s{S_ADMA}, s{S_ADMA}, {M.A_STRIDE}")
            emit(f"s_cmp_ge_u32 s{S_ADMA}, {3 * M.A_STRIDE}")
            emit(f"s_cselect_b32 s{S_ADMA}, 0, s{S_ADMA}")
            if mx:
                emit(f"s_add_u32 s{S_SC}, s{S_SC}, s{S_SM0 + 1}")
                emit(f"s_addc_u32 s{S_SC + 1}, s{S_SC + 1}, 0")
        if s == 1 and dma:
            emit(f"s_add_u32 s{S_W}, s{S_W}, 128")
            emit(f"s_addc_u32 s{S_W + 1}, s{S_W + 1}, 0")


def prologue(emit, M):
    def pieces(is_w, stage):
        for p in range(8):
            base = (M.W_BASE + stage * M.W_STRIDE) if is_w else stage * M.A_STRIDE
            emit(f"s_add_u32 m0, s{S_M0W}, {base + p * 4096}")
            emit("s_nop 0")
            emit(f"global_load_lds_dwordx4 {vr(VOFF + (8 if is_w else 0) + p)}, s[{S_W if is_w else S_A}:{(S_W if is_w else S_A) + 1}]")
        sp = S_W if is_w else S_A
        emit(f"s_add_u32 s{sp}, s{sp}, 128")
        emit(f"s_addc_u32 s{sp + 1}, s{sp + 1}, 0")

    emit(f"v_mov_b32 {vr(V_UNIT)}, 0x7f7f7f7f")
    if M.mx:   # K-tile 0's scale dwords first (K-tile 1's follow in step 0 of K-tile 0)
        emit(f"global_load_dwordx4 {vr(V_SB, 4)}, {vr(V_SOFF)}, s[{S_SC}:{S_SC + 1}]")
        emit(f"s_add_u32 s{S_SC}, s{S_SC}, s{S_SM0 + 1}")
        emit(f"s_addc_u32 s{S_SC + 1}, s{S_SC + 1}, 0")
    # K-tiles 0 and 1 whole; K-tile 2's A half follows in step 0 of K-tile 0
    pieces(False, 0); pieces(True, 0); pieces(False, 1); pieces(True, 1)
    emit(f"s_mov_b32 s{S_ANEXT}, {M.A_STRIDE}")
    emit(f"s_mov_b32 s{S_ADMA}, {2 * M.A_STRIDE}")
    for x in range(4):
        emit(f"v_mov_b32 {vr(ABASE + x)}, {vr(vaddr(False, 0, x))}")
    emit("s_waitcnt vmcnt(16)")
    emit("s_barrier")
    for n in range(8):
        for kk in range(2):
            frag_read(emit, 0, n, kk, 0, 0)


def gen(mx):
    M = Map(mx)
    L = []
    emit = L.append
    emit(f"; ---- gemm_g4f K loop, {'mx' if mx else 'a3'} (generated by gen_gemm_g4f.py; do not edit)")
    prologue(emit, M)
    ktile(emit, M, 0, first=True)
    emit("L_g4f_loop_%=:")
    emit(f"s_cmp_eq_u32 s{S_CNT}, 0")
    emit("s_cbranch_scc1 L_g4f_tail_%=")
    ktile(emit, M, 1)
    ktile(emit, M, 0)
    emit(f"s_sub_u32 s{S_CNT}, s{S_CNT}, 1")
    emit("s_branch L_g4f_loop_%=")
    emit("L_g4f_tail_%=:")
    ktile(emit, M, 1)                                         # K-tile nT-3: the last one that stages (all of K-tile nT-1)
    ktile(emit, M, 0, dma=False)                              # K-tile nT-2 (mx: still requests the scales of K-tile nT-1)
    ktile(emit, M, 1, dma=False, last=True)
    emit("s_waitcnt vmcnt(0)")
    emit("s_nop 15")
    emit("s_nop 15")
    return L, M


def main():
    here = os.environ.get("S2V_GEN_OUT") or os.path.dirname(os.path.abspath(__file__))  # S2V_GEN_OUT: tests/test_host_cpu.py regenerates into a scratch directory
    maps = {}
    for mx in (False, True):
        body, M = gen(mx)
        maps[mx] = M
        with open(os.path.join(here, f"gemm_g4f_body_{'mx' if mx else 'a3'}.inc"), "w") as f:
            for ln in body:
                f.write('"' + ln + '\\n\\t"\n')
    with open(os.path.join(here, "gemm_g4f_regs.h"), "w") as f:
        f.write("// generated by gen_gemm_g4f.py: register constraints and LDS maps of gemm_g4f\n#pragma once\n")
        for mx, tag in ((False, "A3"), (True, "MX")):
            M = maps[mx]
            f.write(f"#define G4F_{tag}_LDS_BYTES {M.LDS}\n#define G4F_{tag}_A_STRIDE {M.A_STRIDE}\n#define G4F_{tag}_W_BASE {M.W_BASE}\n#define G4F_{tag}_W_STRIDE {M.W_STRIDE}\n#define G4F_{tag}_S_BASE {M.S_BASE}\n")
        for k in range(8):
            f.write(f'#define G4F_ACC{k} "{{a[{32 * k}:{32 * k + 31}]}}"\n')
        f.write(f'#define G4F_VADDR "{{v[{VADDR}:{VADDR + 15}]}}"\n#define G4F_VOFF "{{v[{VOFF}:{VOFF + 15}]}}"\n#define G4F_VMX "{{v[{V_SH}:{V_SH + 3}]}}"\n')
        f.write(f'#define G4F_PTR "{{s[{S_A}:{S_A + 3}]}}"\n#define G4F_SIN "{{s[{S_M0W}:{S_M0W + 1}]}}"\n#define G4F_SSC "{{s[{S_SC}:{S_SC + 1}]}}"\n#define G4F_SSM0 "{{s[{S_SM0}:{S_SM0 + 1}]}}"\n')
        clob = [f"v{r}" for r in range(0, 128)] + [f"v{r}" for r in range(ABASE, V_SH)] + [f"v{r}" for r in range(V_SB, V_UNIT + 1)] + [f"s{S_ANEXT}", f"s{S_ADMA}", f"s{S_AM0}"]
        f.write("#define G4F_CLOBBERS " + ", ".join(f'"{c}"' for c in clob) + ', "vcc", "scc", "m0", "memory"\n')


if __name__ == "__main__":
    main()
