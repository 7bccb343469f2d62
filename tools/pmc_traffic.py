"""profiles/pmc_traffic.json from the FETCH_SIZE / WRITE_SIZE passes of tools/gpu_evidence.sh (leg "traffic") (run on the GPU box, where the
rocprofv3 databases are): HBM-side bytes per launch of every C-ABI entry point bench.py can name as dominant."""
import glob, json, os, sqlite3, sys

out_path = sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/pmc_traffic.json'
source = sys.argv[2] if len(sys.argv) > 2 else 'the FETCH_SIZE / WRITE_SIZE summary of the same round in profiles/'


def base(name):
    return name.replace('void ', '').split('(')[0].split('<')[0].strip()


def per_kernel(dbdir, counter):
    """{kernel base name: average counter value per launch}: from the pass's database on the GPU box, or -- the databases are deleted
    once summarised -- from the committed summary text (tools/pmc_summary.py's table) given as the second argument"""
    dbs = glob.glob(f'gpurun_out/{dbdir}/*.db')
    out = {}
    if dbs:
        rows = sqlite3.connect(dbs[0]).execute(
            "select name, avg(counter_value) from pmc_events where counter_name=? group by name", (counter,)).fetchall()
        for n, v in rows:
            out[base(n)] = out.get(base(n), 0.0) + v      # (template instances of one kernel never share a run of the bench)
        return out
    txt = source if os.path.exists(source) else os.path.join('profiles', source)
    for ln in open(txt):
        f = ln.rstrip().split()
        if len(f) >= 5 and f[-4] == counter:
            out[base(' '.join(f[:-4]))] = float(f[-2])
    return out


fetch, write = per_kernel('pmc_fetch', 'FETCH_SIZE'), per_kernel('pmc_write', 'WRITE_SIZE')
# kernels of each C-ABI entry point by base name, the FIRST alternative that ran (keys = the tags NeuralObjectField._call times its
# launches under: what bench.py names as the dominant entry).  Since round 6 the scatter launch k_hash_bwd_agg_dx also computes dL/dx,
# k_hash_bwd_lds carries the row reduction and the pose rows, the MLP backward is k_mlp_bwd_both and the optimiser launch k_adam_tail;
# the older kernels still run in the bench's A/B phases (a few launches) and must not be added on top.
groups = {'hash_bwd[table+table_lds]': [['k_hash_bwd_agg_dx', 'k_hash_bwd_lds'], ['k_hash_bwd_agg', 'k_hash_bwd_lds']],
          'hash_bwd[input]': [['k_hash_dx']], 'nof_hash_encode_fwd': [['k_hash_fwd']],
          'nof_mlp_bwd_tiles': [['k_mlp_bwd_both'], ['k_mlp_bwd_color', 'k_mlp_bwd_sigma']], 'nof_mlp_fwd': [['k_mlp_fwd']],
          'nof_encode_mlp_fwd': [['k_enc_mlp_fwd']], 'nof_adam_step': [['k_adam_tail'], ['k_adam']]}


def total(table, alternatives):
    for names in alternatives:
        if names[0] in table:
            return sum(table.get(n, 0.0) for n in names)
    return 0.0
out = {'_note': 'HBM-side bytes per launch from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, KiB * 1024), cfg2 batch '
                '(4096 rays x 192 samples, L=16, T=2^19), summed over the kernels of each C-ABI entry point. RAW counter values: '
                'on gfx950 FETCH_SIZE under-reports wide coalesced streaming reads by 2x (MI355X_MICROARCH.md, HBM section); for '
                '8-byte gathers and atomics it is uncalibrated. Source: profiles/' + source + ' (backward over the work list of non-zero '
                'tiles, 200 warm-up steps)',
       'workload': f"cfg2 batch, bench.py default precision (fp16x3), {os.environ.get('KF', '64')}-keyframe pool"}
for k, names in groups.items():
    f = total(fetch, names) * 1024
    w = total(write, names) * 1024
    out[k] = {'fetch_bytes': f, 'write_bytes': w, 'traffic_bytes': f + w}
json.dump(out, open(out_path, 'w'), indent=1)
print(json.dumps({k: v['traffic_bytes'] for k, v in out.items() if isinstance(v, dict)}))
