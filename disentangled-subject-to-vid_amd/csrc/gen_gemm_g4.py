#!/usr/bin/env python3
"""Generator of the K loop of gemm_g4 (csrc/gemm_g4.hip): writes gemm_g4_body.inc -- ONE asm statement that takes a 256 x 256 output
tile from "nothing staged" to "accumulators complete" -- and gemm_g4_regs.h (its register constraints, clobbers and LDS size).  Run by
build.py when the outputs are older than this script; the outputs are committed.

Shape of the kernel: four waves (2 x 2), ONE per SIMD, 128 x 128 wave tiles (256 accumulator registers in the AGPR half of the file),
K-tiles of 64 bf16 = 128-byte LDS rows in two 64-KiB stages ([A 256 rows | W 256 rows] each), operands by LDS-DMA.  A wave tile of
128 x 128 reads 8 fragments per 16 MFMA where the eight-wave 128 x 64 tiling of gemm_bf16_pp64 reads 12, and with one wave per SIMD
nothing arbitrates: the stream below IS the schedule.  Per K-tile and wave: 64 MFMA (2065 matrix-pipe cycles), 32 ds_read_b128,
16 LDS-DMA pieces + 16 M0 writes, 4 SALU of pointer arithmetic, ONE barrier -- 1.1 fillers per MFMA where the
pipe hides ~5.
Why asm (as attention_q4): with more than 256 registers hipcc selects AGPR-form MFMAs and moves operands through v_accvgpr copies, puts
s_nop / s_waitcnt where its hazard model wants them and the 64-bit-vaddr form of the LDS-DMA inside loops; the round-2 C++ kernel of
this shape (gemm_q4, diagnostics library) ran 2300-2500 cycles per K-tile.

What the measurements said (tools/stall_g4.py, tools/g4_ablate.sh; cycles per K-tile, floor 2065):
  * without the LDS-DMA instructions the loop runs AT the floor (2063.5) with every read, wait and barrier in place: all overhead is
    DMA -- its issue cost or its latency;
  * pieces one per two MFMAs over the two steps after the stage's barrier (this schedule, without the prefetch): 2136 (QKV), 2177 (FF1),
    2250 (out), 2400 (FF2); without the barrier (waits kept) FF2 drops to 2092: waves wait at the barrier for data that had
    1500-2000 cycles to arrive;
  * all 16 pieces right behind the barrier (earliest possible issue): 2770 -- back-to-back LDS-DMA instructions stall the issuing wave
    for hundreds of cycles: pieces must be SPREAD;
  * a ring of five K32 stages (64-byte rows, pieces four tiles ahead, one per four MFMAs): 2410-2750 -- a piece then touches sixteen
    half cache lines instead of eight whole ones.
  * an L2 prefetch (G4_PF > 0: PF K-tiles ahead each wave touches the 128 cache lines of its own future pieces with two
    global_load_dword, one lane per line, result discarded): 2470-3070 -- a load with 64 distinct lines per instruction occupies the
    CU's address path longer than the latency it was meant to hide.
  * register-staged operands (G4_STAGE=reg: global_load_dwordx4 of K-tile t+2 into one of two 64-VGPR sets, ds_write_b128 of K-tile t+1
    behind a counted vmcnt -- 1-1.5 K-tiles of lead instead of 0.75-1; results bit-identical): 3020-3440 with the 16 loads and the 16
    writes dense in one step each, 2340-2480 with both spread one per two MFMAs (G4_REGV=2) against 2154-2453 for this schedule on the
    same box.  Writes alone cost 215 cycles per K-tile (13 per ds_write_b128: the LDS-DMA writes LDS without holding the wave), and
    with the writes removed the loop still waits for the loads (FF2: 2790): operands that miss the XCD's L2 take longer than one K-tile
    to arrive, a third register set does not fit (292 VGPRs), so more lead needs a third LDS stage, which 160 KiB only has for one
    operand.
So the schedule below stands: 3-4 % faster than gemm_bf16_pp64 on the four C3 shapes (profiles/r03_gemm_g4.txt); what is left above the
floor is the latency of operands that come from beyond the XCD's L2, which two stages cannot cover.

Schedule of K-tile t (stage g = t & 1), four steps s of 16 MFMA (acc[i][j] += W-fragment i x A-fragment j of k-step s):
  every step : first 8 MFMA slots carry the 8 fragment reads of the NEXT step (step 3: of step 0 of K-tile t+1, other stage)
  step 0     : W pieces of K-tile t+1 -> stage g^1 (its W region was last read in step 2 of K-tile t-1)
  step 1     : (G4_PF > 0 only) the two prefetch loads of K-tile t+PF
  step 3     : s_waitcnt vmcnt(0) lgkmcnt(0) (vmcnt(2) with the prefetch) + s_barrier first, then A pieces of K-tile t+2 -> stage g (free since that barrier)
RAW: K-tile t+1 (A issued in step 3 of t-1, W in step 0 of t) is awaited by every wave's vmcnt before the barrier of step 3 of t; its first read is that step's prefetch.  WAR: the last reads of stage g's
A (W) region are the prefetch of step 2 of K-tile t (step 2 of t-1 for stage g^1's W region), completed (lgkmcnt(0)) before the barrier
that precedes the overwriting DMA.
Registers: a[0:255] acc[i][j] at 64 i + 16 j (OUT); v[0:63] fragments [buffer][W 0-3 | A 0-3]; v[64:79] IN fragment addresses
[A | W][stage][step]; v[80:95] IN staging offsets [A | W][piece]; v[96:97] IN prefetch offsets A / W (lane = one cache line of the wave's
pieces); v[98:99] prefetch results (never read); s[36:37] / s[38:39] IN next A / W K-tile to stage; s40 IN LDS address of the wave's
piece 0 of A in stage 0; s41 IN number of [odd, even] K-tile pairs of the loop = (nT - 2 - max(PF, 2)) / 2; nT = K / 64 even, >= 4.
"""
import os

ACC, FRAG, VADDR, VOFF, VPF, VPFD = 0, 0, 64, 80, 96, 98
S_A, S_W, S_M0W, S_CNT = 36, 38, 40, 41
ABLATE = set(filter(None, os.environ.get("G4_ABLATE", "").split(",")))
STAGE = os.environ.get("G4_STAGE", "dma")  # "dma": LDS-DMA pieces one K-tile ahead; "reg": global_load -> VGPR two K-tiles ahead, ds_write (below)
VREG, VWR = 100, 228                       # "reg": two sets of 16 pieces x 4 VGPRs (v[100:227]); ds_write addresses of stage 0 / 1 (v228, v229)
# G4_W3 = 1: THREE stages for the W operand (LDS: [A0 | A1 | W0 | W1 | W2] x 32 KiB = the whole 160 KiB).  The W pieces of K-tile t+2 are issued
# in step 0 of K-tile t (stage (t+2) % 3, last read in step 2 of K-tile t-1) -- 1.75-2 K-tiles of lead where two stages give W 0.75-1 (A has
# 1.25: its pieces follow the barrier of step 3).  The code stays unrolled by two (A parity); the W stage is run-time state: s42 / s43 = LDS
# offset of the W stage K-tile t+1 reads / K-tile t's DMA writes, and the four W fragment addresses of K-tile t+1 (register set (t+1) & 1)
# are rebuilt in step 1 of K-tile t from the stage-0 addresses (v[100:103]).  The step-3 wait becomes vmcnt(8): the W pieces issued in
# this K-tile may stay in flight.  Built, bit-identical to the two-stage loop, and measured LEVEL with it on the same box (profiles/r03_gemm_g4_w3.txt:
# QKV 2152 vs 2155 cycles per K-tile, out 2282 vs 2286, FF1 2188 vs 2203, FF2 2459 vs 2459): the W lead is not what the loop waits for.
# Kept as an option (default off: the body is byte-identical to the round's product loop).
W3 = os.environ.get("G4_W3", "0") == "1"
# G4_A3 = 1: THREE stages for the A operand AND the issue order swapped (LDS: [A0 | A1 | A2 | W0 | W1] x 32 KiB).  vmcnt counts in issue order,
# so an operand only gains lead if what is issued BEHIND it may stay in flight at the wait: the W pieces of K-tile t+2 go out in step 3 of K-tile
# t (right behind the barrier that frees their stage), the A pieces of K-tile t+2 in step 0 of K-tile t (stage (t+2) % 3, free since the barrier
# of K-tile t-1), and the wait of step 3 is vmcnt(8): K-tile t+1 has landed, the A pieces of K-tile t+2 are still on their way.  A then has
# 1.75 K-tiles to arrive (two stages: 1.0), W 1.0 (0.75).  The A stage is run-time state as the W stage is under G4_W3 (same registers).
# Measured (profiles/r04_gemm_g4_a3.txt, same box, cycles per K-tile, floor 2065): QKV 2153 -> 2130, out 2267 -> 2177, FF1 2193 -> 2158,
# FF2 2439 -> 2141 (2.309 -> 2.160 ms, 1256 -> 1343 TFLOP/s); bit-identical.  The round-3 G4_W3 (three W stages) gave nothing because the A
# pieces were issued behind the W pieces' wait: under an in-order counter the operand that is issued LAST before the wait gets the lead.
# The activations (A) are the operand that streams from HBM (FF2: 0.94 GB); the weights come back from the MALL.
A3 = os.environ.get("G4_A3", "1") == "1" and os.environ.get("G4_W3", "0") != "1" and os.environ.get("G4_STAGE", "dma") == "dma"  # the product loop since round 4
assert not (A3 and W3)
S_WNEXT, S_WDMA, S_WM0, WBASE = 42, 43, 48, 100
A_STRIDE, W_BASE, W_STRIDE = (32768, 65536, 32768) if W3 else (32768, 98304, 32768) if A3 else (65536, 32768, 65536)
A_POL = (" " + os.environ["G4_A_POL"]) if os.environ.get("G4_A_POL") else ""   # cache-policy bits of the A / W pieces (nt, sc0, sc1): an experiment switch
W_POL = (" " + os.environ["G4_W_POL"]) if os.environ.get("G4_W_POL") else ""
PF = int(os.environ.get("G4_PF", "0"))  # K-tiles between the L2 prefetch of a tile and its staging (0: none -- the default, see above); even


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


def ktile(emit, g, first=False, dma_w=True, dma_a=True, last=False, prefetch=False):
    if A3:
        return ktile_a3(emit, g, first, dma_w, dma_a, last)
    for s in range(4):
        cur, nxt = s & 1, (s & 1) ^ 1
        if s == 3 and not last:
            emit(f"s_waitcnt vmcnt({8 if (W3 and dma_w) else 2 if (prefetch and PF) else 0}) lgkmcnt(0)")
            emit("s_barrier")
        else:
            emit("s_waitcnt lgkmcnt(0)")
        if W3 and s == 0 and dma_w:
            emit(f"s_add_u32 s{S_WM0}, s{S_M0W}, s{S_WDMA}")
        for k in range(16):
            i, j = k >> 2, k & 3
            acc = ar(ACC + 64 * i + 16 * j, 16)
            c = "0" if (first and s == 0) else acc
            emit(f"v_mfma_f32_32x32x16_bf16 {acc}, {vr(wf(cur, i), 4)}, {vr(af(cur, j), 4)}, {c}")
            if k < 8 and not (last and s == 3):
                gs, ss = (g, s + 1) if s < 3 else (g ^ 1, 0)
                if k < 4:
                    emit(f"ds_read_b128 {vr(wf(nxt, k), 4)}, {vr(vaddr(True, gs, ss))} offset:{k * 4096}")
                else:
                    emit(f"ds_read_b128 {vr(af(nxt, k - 4), 4)}, {vr(vaddr(False, gs, ss))} offset:{(k - 4) * 4096}")
            p = k >> 1
            if s == 0 and dma_w:  # W piece p of K-tile t+1 -> stage g^1 (W3: of K-tile t+2 -> stage (t+2) % 3)
                if k & 1 == 0:
                    if W3:
                        emit(f"s_add_u32 m0, s{S_WM0}, {W_BASE + p * 4096}")
                    else:
                        emit(f"s_add_u32 m0, s{S_M0W}, {(g ^ 1) * 65536 + 32768 + p * 4096}")
                else:
                    emit(f"global_load_lds_dwordx4 {vr(VOFF + 8 + p)}, s[{S_W}:{S_W + 1}]{W_POL}")
            if W3 and s == 1 and not last and k >= 12:  # W fragment addresses of K-tile t+1 (set g^1, idle since step 3 of K-tile t-1)
                emit(f"v_add_u32 {vr(vaddr(True, g ^ 1, k - 12))}, s{S_WNEXT}, {vr(WBASE + k - 12)}")
            if s == 3 and dma_a:  # A piece p of K-tile t+2 -> stage g
                if k & 1 == 0:
                    emit(f"s_add_u32 m0, s{S_M0W}, {g * A_STRIDE + p * 4096}")
                else:
                    emit(f"global_load_lds_dwordx4 {vr(VOFF + p)}, s[{S_A}:{S_A + 1}]{A_POL}")
            if s == 1 and prefetch and PF:  # the A pointer is at K-tile t+2, the W pointer (advanced in step 0) too
                if k == 3:
                    emit(f"global_load_dword {vr(VPFD)}, {vr(VPF)}, s[{S_A}:{S_A + 1}] offset:{(PF - 2) * 128}")
                if k == 11:
                    emit(f"global_load_dword {vr(VPFD + 1)}, {vr(VPF + 1)}, s[{S_W}:{S_W + 1}] offset:{(PF - 2) * 128}")
        if s == 0 and dma_w:
            emit(f"s_add_u32 s{S_W}, s{S_W}, 128")
            emit(f"s_addc_u32 s{S_W + 1}, s{S_W + 1}, 0")
        if W3 and s == 1 and not last:  # rotate the W stages: K-tile t+2 reads what this K-tile's DMA wrote
            emit(f"s_mov_b32 s{S_WNEXT}, s{S_WDMA}")
            emit(f"s_add_u32 s{S_WDMA}, s{S_WDMA}, {W_STRIDE}")
            emit(f"s_cmp_ge_u32 s{S_WDMA}, {3 * W_STRIDE}")
            emit(f"s_cselect_b32 s{S_WDMA}, 0, s{S_WDMA}")
        if s == 3 and dma_a:
            emit(f"s_add_u32 s{S_A}, s{S_A}, 128")
            emit(f"s_addc_u32 s{S_A + 1}, s{S_A + 1}, 0")


def ktile_a3(emit, g, first, dma_w, dma_a, last):
    """K-tile t under G4_A3 (stage parity g = t & 1 for W and for the fragment-address register set; the A stage is run-time):
    step 0: A pieces of K-tile t+2 -> A stage s43;  step 1: A fragment addresses of K-tile t+1 = s42 + stage-0 addresses, then rotate;
    step 3: vmcnt(8) [the A pieces just issued stay in flight] + barrier, then W pieces of K-tile t+2 -> W stage g"""
    for s in range(4):
        cur, nxt = s & 1, (s & 1) ^ 1
        if s == 3 and not last:
            emit(f"s_waitcnt vmcnt({8 if dma_a else 0}) lgkmcnt(0)")
            emit("s_barrier")
        else:
            emit("s_waitcnt lgkmcnt(0)")
        if s == 0 and dma_a:
            emit(f"s_add_u32 s{S_WM0}, s{S_M0W}, s{S_WDMA}")
        for k in range(16):
            i, j = k >> 2, k & 3
            acc = ar(ACC + 64 * i + 16 * j, 16)
            c = "0" if (first and s == 0) else acc
            emit(f"v_mfma_f32_32x32x16_bf16 {acc}, {vr(wf(cur, i), 4)}, {vr(af(cur, j), 4)}, {c}")
            if k < 8 and not (last and s == 3):
                gs, ss = (g, s + 1) if s < 3 else (g ^ 1, 0)
                if k < 4:
                    emit(f"ds_read_b128 {vr(wf(nxt, k), 4)}, {vr(vaddr(True, gs, ss))} offset:{k * 4096}")
                else:
                    emit(f"ds_read_b128 {vr(af(nxt, k - 4), 4)}, {vr(vaddr(False, gs, ss))} offset:{(k - 4) * 4096}")
            p = k >> 1
            if s == 0 and dma_a:  # A piece p of K-tile t+2 -> A stage (t+2) % 3
                if k & 1 == 0:
                    emit(f"s_add_u32 m0, s{S_WM0}, {p * 4096}")
                else:
                    emit(f"global_load_lds_dwordx4 {vr(VOFF + p)}, s[{S_A}:{S_A + 1}]{A_POL}")
            if s == 1 and not last and k >= 12:  # A fragment addresses of K-tile t+1 (set g^1, idle since step 3 of K-tile t-1)
                emit(f"v_add_u32 {vr(vaddr(False, g ^ 1, k - 12))}, s{S_WNEXT}, {vr(WBASE + k - 12)}")
            if s == 3 and dma_w:  # W piece p of K-tile t+2 -> W stage g (free since this step's barrier)
                if k & 1 == 0:
                    emit(f"s_add_u32 m0, s{S_M0W}, {W_BASE + g * W_STRIDE + p * 4096}")
                else:
                    emit(f"global_load_lds_dwordx4 {vr(VOFF + 8 + p)}, s[{S_W}:{S_W + 1}]{W_POL}")
        if s == 0 and dma_a:
            emit(f"s_add_u32 s{S_A}, s{S_A}, 128")
            emit(f"s_addc_u32 s{S_A + 1}, s{S_A + 1}, 0")
        if s == 1 and not last:  # rotate the A stages: K-tile t+2 reads what this K-tile's DMA wrote
            emit(f"s_mov_b32 s{S_WNEXT}, s{S_WDMA}")
            emit(f"s_add_u32 s{S_WDMA}, s{S_WDMA}, {A_STRIDE}")
            emit(f"s_cmp_ge_u32 s{S_WDMA}, {3 * A_STRIDE}")
            emit(f"s_cselect_b32 s{S_WDMA}, 0, s{S_WDMA}")
        if s == 3 and dma_w:
            emit(f"s_add_u32 s{S_W}, s{S_W}, 128")
            emit(f"s_addc_u32 s{S_W + 1}, s{S_W + 1}, 0")


REGV = int(os.environ.get("G4_REGV", "0"))  # "reg" placement: 0 = loads dense in step 0, writes dense in step 2; 1 = loads one per two MFMAs over steps 0-1; 2 = + writes over steps 1-2


def ktile_reg(emit, g, first=False, loads=True, writes=True, last=False):
    """register-staged K-tile t (stage g = t & 1): the operands of K-tile t+2 are requested in step 0 (steps 0-1) into register set g
    (free since its ds_write in K-tile t-1), the operands of K-tile t+1 (register set g^1, requested one K-tile ago) are written to
    stage g^1 in step 2 (steps 1-2) behind a counted vmcnt: a load has 1-1.5 K-tiles to arrive where an LDS-DMA piece had 0.75-1."""
    wsteps = (1, 2) if REGV >= 2 else (2,)
    for s in range(4):
        cur, nxt = s & 1, (s & 1) ^ 1
        if s == 3 and not last:
            emit("s_waitcnt lgkmcnt(0)")
            emit("s_barrier")
        elif s == wsteps[0] and writes:
            newer = 0 if not loads else (16 if REGV == 0 else 8 * s)  # loads of K-tile t+2 already issued in this tile
            emit(f"s_waitcnt vmcnt({newer}) lgkmcnt(0)")
        else:
            emit("s_waitcnt lgkmcnt(0)")
        for k in range(16):
            i, j = k >> 2, k & 3
            acc = ar(ACC + 64 * i + 16 * j, 16)
            c = "0" if (first and s == 0) else acc
            emit(f"v_mfma_f32_32x32x16_bf16 {acc}, {vr(wf(cur, i), 4)}, {vr(af(cur, j), 4)}, {c}")
            if k < 8 and not (last and s == 3):
                gs, ss = (g, s + 1) if s < 3 else (g ^ 1, 0)
                if k < 4:
                    emit(f"ds_read_b128 {vr(wf(nxt, k), 4)}, {vr(vaddr(True, gs, ss))} offset:{k * 4096}")
                else:
                    emit(f"ds_read_b128 {vr(af(nxt, k - 4), 4)}, {vr(vaddr(False, gs, ss))} offset:{(k - 4) * 4096}")
            pl = k if (REGV == 0 and s == 0) else (s * 8 + (k >> 1) if (REGV >= 1 and s < 2 and (k & 1)) else -1)
            if loads and pl >= 0:     # piece pl of K-tile t+2 (A pieces 0-7, W pieces 0-7) -> register set g
                src = f"{vr(VOFF + pl)}, s[{S_A}:{S_A + 1}]" if pl < 8 else f"{vr(VOFF + pl)}, s[{S_W}:{S_W + 1}]"
                emit(f"global_load_dwordx4 {vr(VREG + 64 * g + 4 * pl, 4)}, {src}")
            pw = k if (REGV < 2 and s == 2) else ((s - 1) * 8 + (k >> 1) if (REGV >= 2 and s in (1, 2) and not (k & 1)) else -1)
            if writes and pw >= 0:    # piece pw of K-tile t+1: register set g^1 -> stage g^1
                off = pw * 4096 if pw < 8 else 32768 + (pw - 8) * 4096
                emit(f"ds_write_b128 {vr(VWR + (g ^ 1))}, {vr(VREG + 64 * (g ^ 1) + 4 * pw, 4)} offset:{off}")
        if loads and s == (0 if REGV == 0 else 1):
            for sp in (S_A, S_W):
                emit(f"s_add_u32 s{sp}, s{sp}, 128")
                emit(f"s_addc_u32 s{sp + 1}, s{sp + 1}, 0")


def gen_reg():
    L = []

    def emit(ln):
        op = ln.split()[0]
        if "nodma" in ABLATE and op in ("global_load_dwordx4", "ds_write_b128"):
            return
        if ("noload" in ABLATE and op == "global_load_dwordx4") or ("nowrite" in ABLATE and op == "ds_write_b128"):
            return
        if "noread" in ABLATE and op == "ds_read_b128":
            return
        if "nobar" in ABLATE and op == "s_barrier":
            return
        L.append(ln)

    emit("; ---- gemm_g4 K loop, register-staged operands (generated by gen_gemm_g4.py; do not edit)")
    emit(f"v_add_u32 {vr(VWR)}, s{S_M0W}, {vr(V_SK)}")          # LDS address of the lane's 16 bytes of the wave's piece 0 of A, stage 0
    emit(f"v_add_u32 {vr(VWR + 1)}, 0x10000, {vr(VWR)}")        # ... stage 1
    for t in range(2):                                          # K-tiles 0 and 1 -> register sets 0 and 1
        for k in range(16):
            src = f"{vr(VOFF + k)}, s[{S_A}:{S_A + 1}]" if k < 8 else f"{vr(VOFF + k)}, s[{S_W}:{S_W + 1}]"
            emit(f"global_load_dwordx4 {vr(VREG + 64 * t + 4 * k, 4)}, {src}")
        for sp in (S_A, S_W):
            emit(f"s_add_u32 s{sp}, s{sp}, 128")
            emit(f"s_addc_u32 s{sp + 1}, s{sp + 1}, 0")
    emit("s_waitcnt vmcnt(16)")
    for k in range(16):
        off = k * 4096 if k < 8 else 32768 + (k - 8) * 4096
        emit(f"ds_write_b128 {vr(VWR)}, {vr(VREG + 4 * k, 4)} offset:{off}")
    emit("s_waitcnt lgkmcnt(0)")
    emit("s_barrier")
    for n in range(8):  # fragments of step 0 of K-tile 0
        if n < 4:
            emit(f"ds_read_b128 {vr(wf(0, n), 4)}, {vr(vaddr(True, 0, 0))} offset:{n * 4096}")
        else:
            emit(f"ds_read_b128 {vr(af(0, n - 4), 4)}, {vr(vaddr(False, 0, 0))} offset:{(n - 4) * 4096}")
    # K-tile 0, s41 pairs [odd, even] (t = 1 .. nT-4), then t = nT-3 (the last that requests operands), nT-2 (still writes nT-1), nT-1
    ktile_reg(emit, 0, first=True)
    emit("L_g4_loop_%=:")
    emit(f"s_cmp_eq_u32 s{S_CNT}, 0")
    emit("s_cbranch_scc1 L_g4_nopf_%=")
    ktile_reg(emit, 1)
    ktile_reg(emit, 0)
    emit(f"s_sub_u32 s{S_CNT}, s{S_CNT}, 1")
    emit("s_branch L_g4_loop_%=")
    emit("L_g4_nopf_%=:")
    ktile_reg(emit, 1)
    ktile_reg(emit, 0, loads=False)
    ktile_reg(emit, 1, loads=False, writes=False, last=True)
    emit("s_nop 15")  # the epilogue reads the accumulators next
    emit("s_nop 15")
    emit(f"s_cmp_lt_u32 s{S_SK + 2}, 2")
    emit("s_cbranch_scc1 L_g4_end_%=")
    for ln in gen_sk_store(f"s{S_SK}", f"s{S_SK + 1}", f"v{V_SK}"):
        emit(ln)
    emit("L_g4_end_%=:")
    return L


def gen():
    L = []

    def emit(ln):
        op = ln.split()[0]
        if "nodma" in ABLATE and op == "global_load_lds_dwordx4":
            return
        if "noread" in ABLATE and op == "ds_read_b128":
            return
        if "nobar" in ABLATE and op == "s_barrier":
            return
        L.append(ln)

    emit("; ---- gemm_g4 K loop (generated by gen_gemm_g4.py; do not edit)")
    if A3:
        return gen_a3(emit, L)
    # prologue: K-tile 0 whole -> stage 0, A half of K-tile 1 -> stage 1 (its W half follows in step 0 of K-tile 0)
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
    if W3:  # K-tile 1's W half too (stage 1); K-tile 2's follows in step 0 of K-tile 0 (stage 2)
        assert PF == 0
        for p in range(8):
            emit(f"s_add_u32 m0, s{S_M0W}, {W_BASE + W_STRIDE + p * 4096}")
            emit("s_nop 0")
            emit(f"global_load_lds_dwordx4 {vr(VOFF + 8 + p)}, s[{S_W}:{S_W + 1}]")
        emit(f"s_add_u32 s{S_W}, s{S_W}, 128")
        emit(f"s_addc_u32 s{S_W + 1}, s{S_W + 1}, 0")
        emit(f"s_mov_b32 s{S_WNEXT}, {W_STRIDE}")
        emit(f"s_mov_b32 s{S_WDMA}, {2 * W_STRIDE}")
        for x in range(4):
            emit(f"v_mov_b32 {vr(WBASE + x)}, {vr(vaddr(True, 0, x))}")
    for k in range(2, PF):  # the lines of K-tiles 2 .. PF-1 (pointer + (k - 2) * 128: the A pointer is at K-tile 2, the W pointer at 1)
        emit(f"global_load_dword {vr(VPFD)}, {vr(VPF)}, s[{S_A}:{S_A + 1}] offset:{(k - 2) * 128}")
        emit(f"global_load_dword {vr(VPFD + 1)}, {vr(VPF + 1)}, s[{S_W}:{S_W + 1}] offset:{(k - 1) * 128}")
    emit(f"s_waitcnt vmcnt({16 if W3 else 8 + 2 * max(0, PF - 2)})")  # K-tile 0 landed; the A half of K-tile 1 (and the prefetches) stay in flight (W3: all of K-tile 1)
    emit("s_barrier")
    for n in range(8):  # fragments of step 0 of K-tile 0
        if n < 4:
            emit(f"ds_read_b128 {vr(wf(0, n), 4)}, {vr(vaddr(True, 0, 0))} offset:{n * 4096}")
        else:
            emit(f"ds_read_b128 {vr(af(0, n - 4), 4)}, {vr(vaddr(False, 0, 0))} offset:{(n - 4) * 4096}")
    # K-tile t prefetches K-tile t+PF (valid while t + PF <= nT - 1).  Sequence: FIRST (t = 0), then pairs [odd, even] with prefetch
    # (s41 of them: t = 1 .. nT-PF-1), then PF/2 - 1 pairs and one odd tile without prefetch (t = nT-PF .. nT-3), then the two last tiles
    emit("; K-tile 0")
    ktile(emit, 0, first=True, prefetch=True)
    emit("L_g4_loop_%=:")
    emit(f"s_cmp_eq_u32 s{S_CNT}, 0")
    emit("s_cbranch_scc1 L_g4_nopf_%=")
    ktile(emit, 1, prefetch=True)
    ktile(emit, 0, prefetch=True)
    emit(f"s_sub_u32 s{S_CNT}, s{S_CNT}, 1")
    emit("s_branch L_g4_loop_%=")
    emit("L_g4_nopf_%=:")
    for _ in range(max(0, PF // 2 - 1)):
        ktile(emit, 1)
        ktile(emit, 0)
    ktile(emit, 1)                            # K-tile nT-3: the last one that stages both halves
    ktile(emit, 0, dma_a=False, dma_w=not W3) # K-tile nT-2: still stages the W half of K-tile nT-1 (W3: that went out with K-tile nT-3)
    ktile(emit, 1, dma_w=False, dma_a=False, last=True)
    emit("s_waitcnt vmcnt(0)")
    emit("s_nop 15")  # the epilogue reads the accumulators next
    emit("s_nop 15")
    # split K: this workgroup's partial tile goes to its slot (the stores read a[...] here, inside the statement that produced them:
    # as operands of a second statement the compiler copied all 256 accumulators out and back, with spills)
    emit(f"s_cmp_lt_u32 s{S_SK + 2}, 2")
    emit("s_cbranch_scc1 L_g4_end_%=")
    for ln in gen_sk_store(f"s{S_SK}", f"s{S_SK + 1}", f"v{V_SK}"):
        emit(ln)
    emit("L_g4_end_%=:")
    return L


def gen_a3(emit, L):
    """prologue of the G4_A3 loop: K-tiles 0 and 1 whole (A0, W0, A1, W1: 32 pieces); K-tile 2's A half follows in step 0 of K-tile 0"""
    assert PF == 0
    for t in range(2):
        for p in range(8):
            emit(f"s_add_u32 m0, s{S_M0W}, {t * A_STRIDE + p * 4096}")
            emit("s_nop 0")
            emit(f"global_load_lds_dwordx4 {vr(VOFF + p)}, s[{S_A}:{S_A + 1}]")
        for p in range(8):
            emit(f"s_add_u32 m0, s{S_M0W}, {W_BASE + t * W_STRIDE + p * 4096}")
            emit("s_nop 0")
            emit(f"global_load_lds_dwordx4 {vr(VOFF + 8 + p)}, s[{S_W}:{S_W + 1}]")
        for sp in (S_A, S_W):
            emit(f"s_add_u32 s{sp}, s{sp}, 128")
            emit(f"s_addc_u32 s{sp + 1}, s{sp + 1}, 0")
    emit(f"s_mov_b32 s{S_WNEXT}, {A_STRIDE}")
    emit(f"s_mov_b32 s{S_WDMA}, {2 * A_STRIDE}")
    for x in range(4):
        emit(f"v_mov_b32 {vr(WBASE + x)}, {vr(vaddr(False, 0, x))}")
    emit("s_waitcnt vmcnt(16)")  # K-tile 0 landed; K-tile 1 stays in flight
    emit("s_barrier")
    for n in range(8):
        if n < 4:
            emit(f"ds_read_b128 {vr(wf(0, n), 4)}, {vr(vaddr(True, 0, 0))} offset:{n * 4096}")
        else:
            emit(f"ds_read_b128 {vr(af(0, n - 4), 4)}, {vr(vaddr(False, 0, 0))} offset:{(n - 4) * 4096}")
    emit("; K-tile 0")
    ktile(emit, 0, first=True)
    emit("L_g4_loop_%=:")
    emit(f"s_cmp_eq_u32 s{S_CNT}, 0")
    emit("s_cbranch_scc1 L_g4_nopf_%=")
    ktile(emit, 1)
    ktile(emit, 0)
    emit(f"s_sub_u32 s{S_CNT}, s{S_CNT}, 1")
    emit("s_branch L_g4_loop_%=")
    emit("L_g4_nopf_%=:")
    ktile(emit, 1)                                   # K-tile nT-3: the last one that stages (A and W of K-tile nT-1)
    ktile(emit, 0, dma_a=False, dma_w=False)         # K-tile nT-2
    ktile(emit, 1, dma_w=False, dma_a=False, last=True)
    emit("s_waitcnt vmcnt(0)")
    emit("s_nop 15")
    emit("s_nop 15")
    emit(f"s_cmp_lt_u32 s{S_SK + 2}, 2")
    emit("s_cbranch_scc1 L_g4_end_%=")
    for ln in gen_sk_store(f"s{S_SK}", f"s{S_SK + 1}", f"v{V_SK}"):
        emit(ln)
    emit("L_g4_end_%=:")
    return L


# ---------------------------------------------------------------------------------------------------------------------------------
# Split K (gemm_g4.hip, GemmArgs::splitk): the hand-over of the fp32 partial tiles.  Layout of one partial: [wave][64 register quads]
# [lane] x 16 bytes -- quad q of a wave = accumulator registers a[4q : 4q + 3], 1 KiB per wave and instruction.  Every access is sc1
# (device scope: written through / never served from an XCD's L2), so no L2 write-back or invalidate is needed around the arrival
# counter.  Operands: %[lo] / %[hi] = the wave's base address (SGPRs), %[voff] = lane * 16, %[ns] = number of partials (sum only).
SK_S0, SK_V0, SK_ZSTRIDE = 50, 100, 262144  # scratch SGPR pair, first of 4 x 32 scratch VGPRs, bytes between the partials of a tile


S_SK, V_SK = 44, 98  # main statement: s[44:47] = {partial slot address lo, hi, number of splits, -}, v98 = lane * 16


def gen_sk_store(lo, hi, voff):
    out = []
    for k in range(8):
        out.append(f"s_add_u32 s{SK_S0}, {lo}, {k * 8192 + 4096}")
        out.append(f"s_addc_u32 s{SK_S0 + 1}, {hi}, 0")
        for g in range(8):
            out.append(f"global_store_dwordx4 {voff}, a[{32 * k + 4 * g}:{32 * k + 4 * g + 3}], s[{SK_S0}:{SK_S0 + 1}] offset:{g * 1024 - 4096} sc1")
    out.append("s_waitcnt vmcnt(0)")  # the partial tile is in memory before the arrival is counted
    return out


def gen_sk_sum():
    """a[...] = partial 0 + partial 1 (+ partial 2 (+ partial 3)), in that order whoever runs it; the partial of the running workgroup
    (%[z]) is taken from its registers instead of memory (same bits, a third less to fetch at three splits)"""
    out = []
    for k in range(8):
        out.append(f"s_add_u32 s{SK_S0}, %[lo], {k * 8192 + 4096}")
        out.append(f"s_addc_u32 s{SK_S0 + 1}, %[hi], 0")
        for zz in range(4):
            if zz >= 2:
                out.append(f"s_cmp_lt_u32 %[ns], {zz + 1}")
                out.append(f"s_cbranch_scc1 L_sk_w{k}_%=")
            if zz:
                out.append(f"s_add_u32 s{SK_S0}, s{SK_S0}, {SK_ZSTRIDE}")
                out.append(f"s_addc_u32 s{SK_S0 + 1}, s{SK_S0 + 1}, 0")
            out.append(f"s_cmp_eq_u32 %[z], {zz}")
            out.append(f"s_cbranch_scc1 L_sk_o{k}_{zz}_%=")
            for g in range(8):
                v = SK_V0 + 32 * zz + 4 * g
                out.append(f"global_load_dwordx4 v[{v}:{v + 3}], %[voff], s[{SK_S0}:{SK_S0 + 1}] offset:{g * 1024 - 4096} sc1")
            out.append(f"s_branch L_sk_n{k}_{zz}_%=")
            out.append(f"L_sk_o{k}_{zz}_%=:")
            for i in range(32):
                out.append(f"v_accvgpr_read_b32 v{SK_V0 + 32 * zz + i}, a{32 * k + i}")
            out.append(f"L_sk_n{k}_{zz}_%=:")
        out.append(f"L_sk_w{k}_%=:")
        out.append("s_waitcnt vmcnt(0)")
        for zz in range(1, 4):
            if zz >= 2:
                out.append(f"s_cmp_lt_u32 %[ns], {zz + 1}")
                out.append(f"s_cbranch_scc1 L_sk_d{k}_%=")
            for i in range(32):
                out.append(f"v_add_f32 v{SK_V0 + i}, v{SK_V0 + i}, v{SK_V0 + 32 * zz + i}")
        out.append(f"L_sk_d{k}_%=:")
        for i in range(32):
            out.append(f"v_accvgpr_write_b32 a{32 * k + i}, v{SK_V0 + i}")
    return out


def main():
    here = os.environ.get("S2V_GEN_OUT") or os.path.dirname(os.path.abspath(__file__))  # S2V_GEN_OUT: tests/test_host_cpu.py regenerates into a scratch directory
    body = gen_reg() if STAGE == "reg" else gen()
    with open(os.path.join(here, "gemm_g4_body.inc"), "w") as f:
        for ln in body:
            f.write('"' + ln + '\\n\\t"\n')
    # the fp16 model dtype (round 5): the same loop on v_mfma_f32_32x32x16_f16 -- staging, swizzle and fragment reads move 16-bit elements
    # whatever they encode, so the mnemonic is the only difference
    with open(os.path.join(here, "gemm_g4_body_f16.inc"), "w") as f:
        for ln in body:
            f.write('"' + ln.replace("v_mfma_f32_32x32x16_bf16", "v_mfma_f32_32x32x16_f16") + '\\n\\t"\n')
    for name, body in (("gemm_g4_sk_sum.inc", gen_sk_sum()),):
        with open(os.path.join(here, name), "w") as f:
            for ln in body:
                f.write('"' + ln + '\\n\\t"\n')
    clob = [f"v{r}" for r in range(0, 64)] + ([f"v{r}" for r in range(VREG, VWR + 2)] if STAGE == "reg" else [f"v{VPFD}", f"v{VPFD + 1}"])
    if W3 or A3:
        clob += [f"v{WBASE + x}" for x in range(4)] + [f"s{S_WNEXT}", f"s{S_WDMA}", f"s{S_WM0}"]
    with open(os.path.join(here, "gemm_g4_regs.h"), "w") as f:
        f.write("// generated by gen_gemm_g4.py: the physical registers the K loop of gemm_g4 owns, and its LDS size\n#pragma once\n")
        f.write(f"#define G4_PF {PF}\n#define G4_LDS_BYTES {163840 if (W3 or A3) else 131072}\n")
        f.write(f"#define G4_A_STRIDE {A_STRIDE}\n#define G4_W_BASE {W_BASE}\n#define G4_W_STRIDE {W_STRIDE}  // LDS map of the operand stages: A stage g at g * A_STRIDE, W stage h at W_BASE + h * W_STRIDE\n")
        for k in range(8):
            f.write(f'#define G4_ACC{k} "{{a[{32 * k}:{32 * k + 31}]}}"\n')
        f.write(f'#define G4_VADDR "{{v[{VADDR}:{VADDR + 15}]}}"\n#define G4_VOFF "{{v[{VOFF}:{VOFF + 15}]}}"\n#define G4_VPF "{{v[{VPF}:{VPF + 1}]}}"\n')
        f.write(f'#define G4_PTR "{{s[{S_A}:{S_A + 3}]}}"\n#define G4_SIN "{{s[{S_M0W}:{S_M0W + 1}]}}"\n')
        f.write(f'#define G4_SK "{{s[{S_SK}:{S_SK + 3}]}}"\n#define G4_VSK "{{v{V_SK}}}"\n')
        f.write("#define G4_SK_CLOBBERS " + ", ".join(f'"v{r}"' for r in range(SK_V0, SK_V0 + 128)) + f', "s{SK_S0}", "s{SK_S0 + 1}", "scc", "memory"\n')
        f.write("#define G4_CLOBBERS " + ", ".join(f'"{c}"' for c in clob) + f', "s{SK_S0}", "s{SK_S0 + 1}", "vcc", "scc", "m0", "memory"\n')


if __name__ == "__main__":
    main()
