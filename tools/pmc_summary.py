"""Per-kernel averages of rocprofv3 --pmc counters (SQLite output): python tools/pmc_summary.py <dir> [<dir> ...]"""
import glob, os, sqlite3, sys
for d in sys.argv[1:]:
    for db in glob.glob(os.path.join(d, '**', '*.db'), recursive=True):
        con = sqlite3.connect(db)
        rows = con.execute("select name, counter_name, count(*), avg(counter_value), avg(duration) from pmc_events "
                           "where (name like '%k_%' or name like '%copy%' or name like '%Copy%') group by name, counter_name "
                           "order by avg(duration) desc").fetchall()
        print(f'== {db}')
        print(f"{'kernel':56s} {'counter':26s} {'n':>4s} {'avg_value':>16s} {'avg_us':>9s}")
        for r in rows:
            print(f"{r[0].split('(')[0][:56]:56s} {r[1]:26s} {r[2]:4d} {r[3]:16.4g} {r[4] / 1e3:9.1f}")
