"""rocprofv3 --kernel-trace --stats SQLite output -> text summary (per-kernel calls / avg / min / max / total).
    python tools/prof_summary.py gpurun_out/prof/bench_results.db > profiles/<name>.txt"""
import sqlite3, sys
con = sqlite3.connect(sys.argv[1])
rows = con.execute("select name, count(*), avg(duration), min(duration), max(duration), sum(duration), max(vgpr_count), "
                   "max(accum_vgpr_count), max(scratch_size), max(lds_size), max(grid_x), max(workgroup_x) "
                   "from kernels group by name order by sum(duration) desc").fetchall()
tot = sum(r[5] for r in rows)
print(f"# {sys.argv[1]}  (durations in microseconds; rocprofv3 --kernel-trace --stats)")
print(f"{'kernel':72s} {'calls':>6s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'total_ms':>9s} {'%':>6s} {'vgpr':>5s} {'agpr':>5s} {'scratch':>7s} {'lds':>7s} {'grid':>9s} {'wg':>5s}")
for r in rows:
    print(f"{r[0][:72]:72s} {r[1]:6d} {r[2]/1e3:10.1f} {r[3]/1e3:10.1f} {r[4]/1e3:10.1f} {r[5]/1e6:9.2f} {100*r[5]/tot:6.1f} {r[6]:5d} {r[7]:5d} {r[8]:7d} {r[9]:7d} {r[10]:9d} {r[11]:5d}")
