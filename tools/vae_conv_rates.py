"""Per-convolution MFMA rates of the untiled VAE decode (13 x 60 x 90 latents): zips the convolutions' (M, N, K) in launch order (diagnostics
library, S2V_VAE_CONV_LOG) with the GEMM kernels of a rocprofv3 kernel trace of the same process.
    S2V_LIB=disentangled-subject-to-vid_amd/libs2v_hip_diag.so S2V_VAE_CONV_LOG=/tmp/convs.txt rocprofv3 --kernel-trace --output-format csv -d /tmp/vt -o t -- \
        python tools/vae_profile_probe.py once;  python tools/vae_conv_rates.py /tmp/convs.txt /tmp/vt"""
import collections, csv, glob, os, sys
convs = [tuple(int(x) for x in ln.split()) for ln in open(sys.argv[1])]
tr = glob.glob(os.path.join(sys.argv[2], "**", "*kernel_trace.csv"), recursive=True)[0]
rows = sorted(csv.DictReader(open(tr)), key=lambda r: int(r["Start_Timestamp"]))
gemms = [r for r in rows if r["Kernel_Name"].startswith(("void gemm_bf16_pp64", "void gemm_bf16_stag", "void gemm_g4", "void gemm_bf16_128"))]
mf = [c for c in convs if c[4]]
print(f"{len(mf)} MFMA convolutions logged, {len(gemms)} GEMM launches in the trace (1x1 shortcuts are GEMMs too)")
# the 1x1 shortcut GEMMs are launched between convs: match by order using grid size = tiles of (M, N)
agg = collections.defaultdict(lambda: [0.0, 0.0, 0])
gi = 0
for (M, N, K, epi, _) in mf:
    tiles256 = ((M + 255) // 256) * ((N + 255) // 256)
    while gi < len(gemms):
        r = gemms[gi]; gi += 1
        wg = int(r.get("Grid_Size_X", r.get("Grid_Size", 0))) // max(1, int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 1))))
        name = r["Kernel_Name"].split("(")[0]
        ok = wg in (tiles256, ((M + 255) // 256) * ((N + 127) // 128), ((M + 127) // 128) * ((N + 127) // 128), (tiles256 + 7) // 8 * 8)
        if ok or "conv" in name:
            d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
            a = agg[(name[5:40], M, N, K)]
            a[0] += 2.0 * M * N * K; a[1] += d; a[2] += 1
            break
tot_f = sum(a[0] for a in agg.values()); tot_t = sum(a[1] for a in agg.values())
print(f"{'kernel':36s} {'M':>8s} {'N':>5s} {'K':>6s} {'n':>4s} {'ms each':>8s} {'TFLOP/s':>8s} {'share':>6s}")
for (name, M, N, K), (f, t, n) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{name:36s} {M:8d} {N:5d} {K:6d} {n:4d} {t / n * 1e3:8.3f} {f / t / 1e12:8.1f} {t / tot_t:6.1%}")
print(f"all matched convolutions: {tot_f / 1e12:.1f} TFLOP in {tot_t * 1e3:.1f} ms = {tot_f / tot_t / 1e12:.1f} TFLOP/s")
