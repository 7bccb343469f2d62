"""Is a launch gap the host being late?  From a rocprofv3 --kernel-trace --hip-runtime-trace database: for every kernel of one
step, when its hipLaunchKernel call RETURNED on the host relative to when the previous kernel on the same stream ENDED on the device.
    python tools/launch_lag.py <results.db> [step index]"""
import sqlite3, sys
con = sqlite3.connect(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
names = [r[0] for r in con.execute("select name from sqlite_master where type in ('table','view') order by name")]
print('# tables/views:', ' '.join(names))
def cols(t):
    return [r[1] for r in con.execute(f"pragma table_info('{t}')")]
for t in ('kernels', 'regions', 'regions_and_samples', 'top'):
    if t in names:
        print(f'# {t}:', cols(t))
kc = cols('kernels')
rows = con.execute("select * from kernels order by start").fetchall()
ki = {c: i for i, c in enumerate(kc)}
idx = [i for i, r in enumerate(rows) if r[ki['name']].startswith('k_pose_fwd')]
n = min(n, len(idx) - 2)
i0, i1 = idx[n], idx[n + 1]
t0 = rows[i0][ki['start']]
# host side: launch API regions, matched to dispatches through the correlation / stack id when the view has one, else by order
reg = None
if 'regions' in names:
    rc = cols('regions')
    ri = {c: i for i, c in enumerate(rc)}
    reg = [r for r in con.execute("select * from regions order by start") if 'LaunchKernel' in str(r[ri['name']]) or 'ModuleLaunch' in str(r[ri['name']])]
    print(f'# {len(reg)} launch API regions, {len(rows)} kernel dispatches')
key = next((c for c in ('stack_id', 'correlation_id', 'corr_id') if c in ki), None)
by = {}
if reg is not None and key is not None and key in ri:
    by = {r[ri[key]]: r for r in reg}
last_end = {}
print('#  start_us   end_us  gap_us  host_return_us  host_lag_us(+ = API returned AFTER the previous kernel on the stream ended)  stream kernel')
for j, r in enumerate(rows[i0:i1 + 1]):
    st = r[ki['stream_id']] if 'stream_id' in ki else 0
    s, e = r[ki['start']], r[ki['end']]
    api = by.get(r[ki[key]]) if by else (reg[i0 + j] if reg is not None and len(reg) == len(rows) else None)
    prev = last_end.get(st)
    gap = (s - prev) / 1e3 if prev is not None else float('nan')
    if api is not None:
        ret = api[ri['end']]
        lag = (ret - prev) / 1e3 if prev is not None else float('nan')
        print(f"{(s - t0) / 1e3:9.1f} {(e - t0) / 1e3:8.1f} {gap:7.1f} {(ret - t0) / 1e3:12.1f} {lag:10.1f}   s{st} {r[ki['name']][:50]}")
    else:
        print(f"{(s - t0) / 1e3:9.1f} {(e - t0) / 1e3:8.1f} {gap:7.1f} {'?':>12} {'?':>10}   s{st} {r[ki['name']][:50]}")
    last_end[st] = e
