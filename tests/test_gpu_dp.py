"""The N > 1 launch path of bench.py on ONE GPU: two processes started by torch.distributed.run share cuda:0 and talk
through gloo (RCCL refuses two ranks on one device), so everything except the transport is the product path: keyframe
sharding by rank, all-gather of the pose table / octree cloud, the per-step all-reduce of the flat gradient buffer through
the grad_sync hook, barrier + MAX-over-ranks timing, rank-0 JSON line.  After the synchronised steps both replicas must
hold bit-identical parameters."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(overlap, port, backend='gloo', ranks=2, force='0', steps=6, payload='fp32', mode='allreduce', inject=None, precision=None):
    env = dict(os.environ, NOF_DIST_BACKEND=backend, HSA_ENABLE_IPC_MODE_LEGACY='0', NOF_DP_OVERLAP=overlap, NOF_DP_FORCE=force,
               NOF_DP_PAYLOAD=payload, NOF_DP_MODE=mode)
    if inject is not None:
        env['NOF_DP_INJECT_OVERFLOW'] = inject
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(ranks), '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.join(ROOT, 'bench.py'), '--gpus', str(ranks), '--steps', str(steps), '--warmup', '2',
           '--keyframes', '3', '--no-cpu-baseline', '--settle', '0', '--round-steps', '0']
    if precision is not None:
        cmd += ['--precision', precision]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    line = [l for l in out.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(line) == 1, out.stdout[-2000:]                          # rank 0 only
    return json.loads(line[0])


def test_sharded_optimiser_two_ranks_one_gpu_gloo(nof):
    """GradSync mode 'zero1' through bench.py's own launch path, two ranks sharing the one GPU (gloo transport): the replicas end
    with bit-identical parameters (every rank receives every shard) that equal the all-reduce form's."""
    d = _run('1', 29541, mode='zero1')
    d0 = _run('0', 29542)
    assert d['dp_mode'] == 'zero1' and d['dp_param_checksum_spread'] == 0.0 and d['flags'] == 0
    assert abs(d['param_checksum'] - d0['param_checksum']) <= 2e-5 * d0['param_checksum']
    assert d['collectives_per_step'] == 2


def test_bench_two_ranks_one_gpu_gloo(nof):
    d = _run('1', 29533)               # bucketed: fine hash levels reduced asynchronously beside the rest of the backward
    d0 = _run('0', 29534)              # one blocking all-reduce of the whole buffer
    assert d0['dp_param_checksum_spread'] == 0.0
    # element-wise sums: bucketing cannot change the result beyond the atomics' summation order inside each rank -- which Adam
    # (eps 1e-15) turns into +-lr on entries whose gradient is rounding noise: a few dozen of 9 M parameters over the 8 steps the
    # checksum is taken after (bench.py takes it right after the timed region: over the 26 steps of the whole run one MLP weight
    # waking up on noise sent 3 of 16 identical runs 4e-4 away from the other 13, profiles/r04_q_rccl_states.txt)
    assert abs(d['param_checksum'] - d0['param_checksum']) <= 2e-5 * d0["param_checksum"]
    assert d['n_gpus'] == 2 and d['scaling'] == 'weak' and d['config']['parallelism'] == 'dp2'
    assert d['flags'] == 0 and d['loss'] == d['loss']                  # finite
    assert d['dp_param_checksum_spread'] == 0.0
    assert d['value'] > 0 and abs(d['value'] - 2 * 4096 * 192 * 1e3 / d['ms_per_step']) < 1e-3 * d['value']
    # two collectives per step in the bucketed form (fine levels + MLP early, everything else in one trailing call), one otherwise;
    # the trailing one carries a copy of the [features | poses] tail in the headroom in front of the gradient buffer
    assert d['collectives_per_step'] == 2 and d0['collectives_per_step'] == 1
    assert d['allreduce_bytes_per_step'] >= d0['allreduce_bytes_per_step'] > 4 * 9_000_000


def test_touched_row_exchange_two_ranks_one_gpu_gloo(nof):
    """GradSync mode 'rows' (round 6) through bench.py's own launch path, two ranks sharing the one GPU: only the
    table rows some rank touched travel (bitmap all-gather + all-reduce of the union's rows), everything behind the table dense.
    Replicas bit-identical, parameters equal to the dense all-reduce form's, and far fewer bytes handed to the collectives."""
    d = _run('1', 29543, mode='rows', steps=8)
    d0 = _run('0', 29544, steps=8)
    assert d['dp_mode'] == 'rows' and d['dp_param_checksum_spread'] == 0.0 and d['flags'] == 0
    assert abs(d['param_checksum'] - d0['param_checksum']) <= 2e-5 * d0['param_checksum']
    assert d['collectives_per_step'] == 3                              # bitmap all-gather, the union's rows, the dense tail
    assert 0 < d['dp_rows_per_step'] < 0.6 * d['dp_table_rows']
    assert d['allreduce_bytes_per_step'] < 0.6 * d0['allreduce_bytes_per_step']
    print(f"touched-row exchange, 2 ranks: {d['dp_rows_per_step']} of {d['dp_table_rows']} rows, {d['allreduce_bytes_per_step'] / 1e6:.2f} MB "
          f"per step vs {d0['allreduce_bytes_per_step'] / 1e6:.2f} MB dense; {d['ms_per_step']:.3f} vs {d0['ms_per_step']:.3f} ms/step")


@pytest.mark.parametrize("overlap,payload,mode", [('1', 'fp32', 'allreduce'), ('1', 'bf16', 'allreduce'), ('0', 'fp32', 'allreduce'),
                                                  ('1', 'fp32', 'zero1'), ('1', 'fp32', 'rows')])
def test_overflow_on_one_rank_is_skipped_by_both(nof, overlap, payload, mode):
    """the bucketed step ran the first slice's share of Adam before the ranks had agreed on the step's overflow flag -- a
    rank that overflowed alone skipped the slice while the other applied a summed gradient that held its inf (NaN weights, replicas
    apart).  Rank 1's fp16 loss scale is raised by 2^40 for step 3 (bench.py: NOF_DP_INJECT_OVERFLOW): its MLP weight gradient is not
    finite in that step, rank 0's is.  Every form of the exchange must end with bit-identical, finite replicas that both skipped
    that step (the skipped step's mark is the sticky bit the host polls: it must be up on rank 0, which did not overflow itself)."""
    port = 29550 + 4 * ['allreduce', 'zero1', 'rows'].index(mode) + 2 * int(overlap) + (payload == 'bf16')
    d = _run(overlap, port, payload=payload, mode=mode, inject='1:3', precision='fp16x3')
    assert d['dp_param_checksum_spread'] == 0.0, d['dp_param_checksum_spread']
    assert d['param_checksum'] == d['param_checksum'] and d['param_checksum'] < 1e30 and d['loss'] == d['loss']
    assert d['flags'] & 12, d['flags']                                  # the skipped step left its mark on rank 0 as well
    clean = _run(overlap, port + 20, payload=payload, mode=mode, precision='fp16x3')
    assert clean['flags'] == 0
    # one step of eight was skipped: the parameters moved less far than in the clean run (Adam's first steps are +-lr each), and
    # nothing blew up
    assert 0.5 * clean['param_checksum'] < d['param_checksum'] < 1.5 * clean['param_checksum']
    assert d['param_checksum'] != clean['param_checksum']
    assert d['param_checksum_parts']['mlp'] == d['param_checksum_parts']['mlp'] and d['param_checksum_parts']['mlp'] < 1e9


def test_bf16_payload_loss_drift_over_50_steps(nof):
    """GradSync(payload='bf16'): the fine hash levels' slice of the gradient travels as bfloat16 (opt-in).  Two ranks, 52 steps
    each way: the replicas stay bit-identical (every rank receives the same rounded sum), half the table bytes go out, and the
    loss after the run is the fp32 run's within a few per cent."""
    d = _run('1', 29541, steps=50, payload='bf16')
    d0 = _run('1', 29542, steps=50)
    assert d['dp_payload'] == 'bf16' and d0['dp_payload'] == 'fp32'
    assert d['dp_param_checksum_spread'] == 0.0 and d['flags'] == 0
    assert d['collectives_per_step'] == 2 and d0['collectives_per_step'] == 2
    assert d['allreduce_bytes_per_step'] < 0.65 * d0['allreduce_bytes_per_step']
    print(f"loss after 52 steps on 2 ranks: bf16 payload {d['loss']:.6f}, fp32 {d0['loss']:.6f}; "
          f"bytes per step {d['allreduce_bytes_per_step']} vs {d0['allreduce_bytes_per_step']}")
    assert abs(d['loss'] - d0['loss']) <= 0.03 * abs(d0['loss'])
    assert abs(d['param_checksum'] - d0['param_checksum']) <= 1e-2 * d0['param_checksum']


def test_rccl_calls_of_the_bucketed_step_one_rank(nof):
    """RCCL on the one GPU a test box has: a one-rank 'nccl' process group whose collectives are forced on (NOF_DP_FORCE).  Every
    all-reduce is then an identity performed by RCCL on its own stream -- which is what this exercises: asynchronous collectives on
    slices of the flat gradient buffer and on the headroom in front of it, the stream hand-over at start() / finish(), the
    communication fields of the bench line.  The parameters must come out as with ONE blocking all-reduce of the whole buffer."""
    d = _run('1', 29537, backend='nccl', ranks=1, force='1')        # two asynchronous collectives per step, Adam in three ranges
    d1 = _run('0', 29539, backend='nccl', ranks=1, force='1')       # one blocking all-reduce of the whole buffer, one Adam
    d0 = _run('1', 29538, backend='nccl', ranks=1)                  # no collective at all (and bench.py's captured-step leg: more steps)
    assert d['collectives_per_step'] == 2 and d1['collectives_per_step'] == 1 and d0['collectives_per_step'] == 0
    assert d['allreduce_bytes_per_step'] >= d1['allreduce_bytes_per_step'] > 4 * 9_000_000 and d0['allreduce_bytes_per_step'] == 0
    assert d['exposed_comm_ms'] is not None and d['exposed_comm_ms'] >= 0 and d0['exposed_comm_ms'] is None
    assert d['flags'] == 0 and d['loss'] == d['loss']
    assert abs(d['param_checksum'] - d1['param_checksum']) <= 2e-5 * d1['param_checksum'], (d['param_checksum_parts'], d1['param_checksum_parts'], d0['param_checksum_parts'])
    # the sharded optimiser (reduce-scatter -> Adam on the rank's shard -> all-gather of the parameters; at one rank the shard is
    # everything and both collectives are identities performed by RCCL): same parameters again
    dz = _run('1', 29540, backend='nccl', ranks=1, force='1', mode='zero1')
    assert dz['dp_mode'] == 'zero1' and dz['collectives_per_step'] == 2 and dz['flags'] == 0
    assert dz['allreduce_bytes_per_step'] >= 2 * 4 * 9_000_000
    assert abs(dz['param_checksum'] - d1['param_checksum']) <= 2e-5 * d1['param_checksum']
    print(f"one-rank RCCL: bucketed {d['ms_per_step']:.3f} ms/step (exposed {d['exposed_comm_ms']:.3f} ms), blocking {d1['ms_per_step']:.3f}, "
          f"sharded optimiser {dz['ms_per_step']:.3f} (exposed {dz['exposed_comm_ms']:.3f}), no collectives {d0['ms_per_step']:.3f}")


def test_bench_two_ranks_two_gpus_rccl(nof):
    """The same launch over RCCL ('nccl' backend, one rank per GPU, xGMI): the transport the driver's scaling run uses.  Needs two
    visible devices -- the test box has one, where this skips; on a multi-GPU node it checks what the gloo variant checks, plus the
    communication fields of the bench line."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip(f'{torch.cuda.device_count()} GPU visible: RCCL refuses two ranks on one device')
    d = _run('1', 29535, backend='nccl')
    d0 = _run('0', 29536, backend='nccl')
    assert d['dp_param_checksum_spread'] == 0.0 and d0['dp_param_checksum_spread'] == 0.0
    assert abs(d['param_checksum'] - d0['param_checksum']) <= 2e-5 * d0['param_checksum']
    assert d['n_gpus'] == 2 and d['collectives_per_step'] == 2 and d0['collectives_per_step'] == 1
    assert d['allreduce_bytes_per_step'] >= d0['allreduce_bytes_per_step'] > 4 * 9_000_000
    assert d['exposed_comm_ms'] is not None and d['exposed_comm_ms'] >= 0
