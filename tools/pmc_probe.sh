#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/probe
mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT/t -o t -- python tools/probe_blaslt.py > $OUT/t.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT -d $OUT/a -o a -- python tools/probe_blaslt.py > $OUT/a.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum SQ_INSTS_VMEM_RD SQ_INSTS_VALU -d $OUT/b -o b -- python tools/probe_blaslt.py > $OUT/b.log 2>&1
python - $OUT <<'PY'
import sqlite3, sys, glob
out = sys.argv[1]
for db in glob.glob(f"{out}/t/*.db"):
    c = sqlite3.connect(db)
    cols = [d[1] for d in c.execute("pragma table_info(kernels)")]
    want = [x for x in ("name","grid_size_x","grid_size","workgroup_size_x","workgroup_size","lds_block_size","vgpr_count","accum_vgpr_count","sgpr_count","scratch_size","duration") if x in cols]
    seen=set()
    for r in c.execute(f"select {','.join(want)} from kernels"):
        if r[0] in seen or 'elementwise' in r[0] or 'distribution' in r[0]: continue
        seen.add(r[0]); print(dict(zip(want, [str(x)[:200] for x in r])))
for sub in "ab":
    for db in glob.glob(f"{out}/{sub}/*.db"):
        c = sqlite3.connect(db)
        q = "select kernel_name, counter_name, count(*), avg(value) from counters_collection group by kernel_name, counter_name"
        for n, ctr, cnt, avg in c.execute(q):
            if 'elementwise' in n or 'distribution' in n or 'fill' in n.lower(): continue
            print(f"{n[:40]:40s} {ctr:30s} n={cnt:3d} avg={avg:16.1f}")
PY
