"""Per-kernel averages of rocprofv3 --pmc counter CSVs: python tools/pmc_summary.py <dir> [<dir> ...]"""
import csv, glob, os, sys, collections
for d in sys.argv[1:]:
    files = glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True)
    print(f'== {d}: {len(files)} file(s)')
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in files:
        with open(f) as fh:
            for row in csv.DictReader(fh):
                k = row.get('Kernel_Name', '?')
                if not k.startswith(('void k_', 'k_')):
                    continue
                k = k.split('(')[0][:70]
                acc[k][row.get('Counter_Name')].append(float(row.get('Counter_Value', 0)))
    for k in sorted(acc):
        print('  ', k, {c: f'{sum(v) / len(v):.4g} (n={len(v)})' for c, v in acc[k].items()})
