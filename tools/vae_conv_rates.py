"""Per-convolution MFMA rates of the untiled VAE decode (13 x 60 x 90 latents): zips the convolutions' (M, N, K) in launch order (diagnostics
library, S2V_VAE_CONV_LOG) with the GEMM kernels of a rocprofv3 kernel trace of the same process.
    S2V_LIB=disentangled-subject-to-vid_amd/libs2v_hip_diag.so S2V_VAE_CONV_LOG=/tmp/convs.txt rocprofv3 --kernel-trace --output-format csv -d /tmp/vt -o t -- \
        python tools/vae_profile_probe.py once;  python tools/vae_conv_rates.py /tmp/convs.txt /tmp/vt"""
import collections, csv, glob, os, sys
PEAK_TF = 2500.0
log = []
for ln in open(sys.argv[1]):
    f = ln.split()
    log.append((int(f[0]), int(f[1]), int(f[2]), int(f[3]), int(f[4]), f[5] if len(f) > 5 else "conv"))
tr = glob.glob(os.path.join(sys.argv[2], "**", "*kernel_trace.csv"), recursive=True)[0]
rows = sorted(csv.DictReader(open(tr)), key=lambda r: int(r["Start_Timestamp"]))
gemms = [r for r in rows if r["Kernel_Name"].startswith(("void gemm_bf16_pp64", "void gemm_bf16_stag", "void gemm_g4", "void gemm_bf16_128", "void gemm_bf16_w8"))]
mf = [c for c in log if c[4]]
print(f"{len(mf)} MFMA GEMM-shaped launches logged ({sum(1 for c in mf if c[5] == 'conv')} convolutions, {sum(1 for c in mf if c[5] != 'conv')} 1 x 1 shortcuts), "
      f"{len(gemms)} GEMM kernels in the trace")
# Round 5: the log now holds EVERY GEMM-shaped launch (convolutions and the 1 x 1 shortcut GEMMs) in launch order, so entry i IS kernel i of the
# trace -- no searching by grid size (round 4 matched a shortcut GEMM to a convolution whose grid it happened to share: rows at 6-7 PFLOP/s).  The
# grid is still checked; a count mismatch or an entry whose grid fits no tiling of its (M, N) is reported, never silently paired.
if len(mf) != len(gemms):
    print(f"WARNING: {len(mf)} logged launches vs {len(gemms)} traced GEMM kernels: the zip below is truncated to the shorter list and may be shifted")
agg = collections.defaultdict(lambda: [0.0, 0.0, 0])
bad = 0
for (M, N, K, epi, _, kind), r in zip(mf, gemms):
    wg = int(r.get("Grid_Size_X", r.get("Grid_Size", 0))) // max(1, int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 1))))
    name = r["Kernel_Name"].split("(")[0]
    t256 = ((M + 255) // 256) * ((N + 255) // 256)
    fits = wg in (t256, ((M + 255) // 256) * ((N + 127) // 128), ((M + 127) // 128) * ((N + 127) // 128), (t256 + 7) // 8 * 8) or wg <= 256
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
    if not fits:
        bad += 1
        print(f"MISMATCH: logged {kind} M={M} N={N} K={K} against {name[5:40]} with {wg} workgroups -- dropped")
        continue
    a = agg[(name[5:40], kind, M, N, K)]
    a[0] += 2.0 * M * N * K; a[1] += d; a[2] += 1
tot_f = sum(a[0] for a in agg.values()); tot_t = sum(a[1] for a in agg.values())
print(f"{'kernel':36s} {'kind':8s} {'M':>8s} {'N':>5s} {'K':>6s} {'n':>4s} {'ms each':>8s} {'TFLOP/s':>8s} {'share':>6s}")
for (name, kind, M, N, K), (f, t, n) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    rate = f / t / 1e12
    flag = "  <-- ABOVE THE DENSE PEAK: pairing error" if rate > PEAK_TF else ""
    print(f"{name:36s} {kind:8s} {M:8d} {N:5d} {K:6d} {n:4d} {t / n * 1e3:8.3f} {rate:8.1f} {t / tot_t:6.1%}{flag}")
print(f"all paired launches: {tot_f / 1e12:.1f} TFLOP in {tot_t * 1e3:.1f} ms = {tot_f / tot_t / 1e12:.1f} TFLOP/s; {bad} dropped as mismatched")
