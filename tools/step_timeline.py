"""Timeline of ONE optimisation step from a rocprofv3 --kernel-trace database: start / end / duration (us, relative to the step's
first launch) and stream of every kernel between two launches of the step's first kernel (the ray marcher -- or the k_mlp_pack right in
front of it, which carries the pose update since round 4: since round 6 the optimiser launch leaves the next step's operand image
and a step starts at its marcher; k_pose_fwd in older traces).
    python tools/step_timeline.py <results.db> [step index, default 240]"""
import sqlite3, sys
con = sqlite3.connect(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 240
rows = con.execute("select name, start, end, stream_id from kernels order by start").fetchall()
idx = [i - 1 if i > 0 and 'k_mlp_pack' in rows[i - 1][0] else i for i, r in enumerate(rows) if 'k_raymarch_wave' in r[0]]
if len(idx) < 3:
    idx = [i for i, r in enumerate(rows) if 'k_mlp_pack' in r[0]]
if len(idx) < 3:
    idx = [i for i, r in enumerate(rows) if r[0].startswith('k_pose_fwd')]
n = min(n, len(idx) - 2)
i0, i1 = idx[n], idx[n + 1]
t0 = rows[i0][1]
print(f"# step {n} of {sys.argv[1]}: start_us end_us dur_us stream kernel")
for r in rows[i0:i1]:
    print(f"{(r[1] - t0) / 1e3:8.1f} {(r[2] - t0) / 1e3:8.1f} {(r[2] - r[1]) / 1e3:7.1f}  s{r[3]}  {r[0][:60]}")
print(f"# next step starts at {(rows[i1][1] - t0) / 1e3:.1f} us")
