"""Summarise rocprofv3 (rocpd sqlite) outputs of tools/profile.sh into small CSVs under profiles/.
usage: python tools/summarize_prof.py gpurun_out/prof_r01 r01"""
import os
import sqlite3
import sys

src, tag = sys.argv[1], sys.argv[2]
os.makedirs("profiles", exist_ok=True)


def short(n):
    n = n.replace("void ", "")
    return n.split("(")[0][:70]


db = os.path.join(src, "trace", "trace_results.db")
if os.path.exists(db):
    c = sqlite3.connect(db)
    rows = list(c.execute("select name,total_calls,total_duration,average,percentage from top_kernels order by total_duration desc limit 25"))
    with open(f"profiles/{tag}_kernel_stats.csv", "w") as f:
        f.write("kernel,calls,total_ms,avg_ms,percent\n")
        for n, calls, tot, avg, pct in rows:
            f.write(f"\"{short(n)}\",{calls},{tot/1e3:.1f},{avg/1e3:.2f},{pct:.2f}\n")
    print("wrote", f"profiles/{tag}_kernel_stats.csv")
for sub, ctr in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    dbs = [os.path.join(src, sub, x) for x in os.listdir(os.path.join(src, sub))] if os.path.isdir(os.path.join(src, sub)) else []
    for db in dbs:
        c = sqlite3.connect(db)
        q = ("select kernel_name, count(*), avg(value), sum(value) from counters_collection where counter_name=? "
             "group by kernel_name order by sum(value) desc limit 20")
        with open(f"profiles/{tag}_{sub}.csv", "w") as f:
            f.write(f"kernel,launches,mean_{ctr}_KB_per_launch_raw,total_{ctr}_KB_raw\n")
            for n, cnt, avg, tot in c.execute(q, (ctr,)):
                f.write(f"\"{short(n)}\",{cnt},{avg:.1f},{tot:.1f}\n")
        print("wrote", f"profiles/{tag}_{sub}.csv")
