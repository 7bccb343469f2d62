#!/usr/bin/env python
"""Benchmark of the Neural Object Field hot path on MI355X.

    python bench.py [--gpus N --steps K --warmup W]

N > 1: one rank per GPU over RCCL.  Started plainly (`python bench.py --gpus N`, no WORLD_SIZE in the environment) it
launches its N ranks itself through torch.distributed.run and fails loudly when fewer than N GPUs are visible; started by
torch.distributed.run (the driver's way) it is one of the ranks.

A step = one full train_loop iteration (batch draw -> occupancy trace -> z sampling -> hash encode -> SDF/colour MLPs ->
compositing + losses -> backward -> [RCCL gradient all-reduce] -> Adam) over a synthetic 640x480 RGBD keyframe pool
resident in HBM.  Workload = BASELINE.json configs[1]: 64 keyframes per GPU, 4096 rays/step, hash L=16 T=2^19 (base 16
-> finest 256), SDF-MLP 3x64 + colour-MLP 2x64, 128+64 samples per ray, 16-bit MFMA operands: fp16 (the reference's
autocast type) with the hi+lo operand split in the forward kernels -- the mode whose SDF/colour outputs are within 1e-3 of
the fp32 oracle (tests/test_gpu_step.py) -- and a loss-scaled fp16 backward.  Prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

def log(msg):
    print(f'[bench {time.strftime("%H:%M:%S")}] {msg}', file=sys.stderr, flush=True)


PRECISION_NOTE = {
    'fp16x3': 'fp16 MFMA operands (hi+lo operand split = 3 MFMAs per product in the forward kernels, loss-scaled fp16 backward)',
    'bf16x3': 'bf16 MFMA operands (hi+lo operand split in the forward kernels, bf16 backward)',
    'bf16': 'bf16 MFMA', 'fp16': 'fp16 MFMA (loss-scaled backward)', 'fp32': 'exact fp32 MFMA'}
# (num_layers, num_layers_color, hidden): BASELINE cfg2's network, the reference's own (nerf_runner.py:221), BASELINE cfg5's
MLP_SHAPES = {'baseline': (3, 2, 64), 'reference': (2, 3, 64), 'cfg5': (4, 4, 128)}
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
L2_PEAK_GBS = 34500.0          # MI355X_MICROARCH.md "L2 (per XCD)": ~34.5 TB/s aggregate over the 8 XCDs' 4 MiB L2s
MFMA_BF16_PEAK_TF = 2500.0     # dense bf16 MFMA


def workload_cfg(args):
    from bundlesdf_amd.config import default_cfg
    return default_cfg(n_step=100000, N_rand=args.rays, num_levels=16, log2_hashmap_size=args.log2_T, finest_res=args.finest,
                       base_res=16, N_samples=128, N_samples_around_depth=64, far=1.0, frame_features=0,
                       save_octree_clouds=False, i_print=10 ** 9, i_weights=10 ** 9)


def build_runner(args, rank, world, device):
    import torch.distributed as dist
    from bundlesdf_amd import synthetic
    from bundlesdf_amd.nerf_runner import NerfRunner
    F_local = args.keyframes
    pool = synthetic.make_pool(n_frames=F_local, H=args.height, W=args.width, fx=600.0 * args.width / 640.0, seed=0,
                               frame_offset=rank * F_local, n_total=F_local * world, analytic_bounds=True)
    cfg = workload_cfg(args)
    cfg.update(sc_factor=pool['sc_factor'], translation=pool['translation'])
    poses, cloud = pool['poses'], pool['pcd_normalized']
    if world > 1:               # identical pose table and octree cloud on every rank
        p = torch.from_numpy(poses).to(device)
        ps = [torch.empty_like(p) for _ in range(world)]
        dist.all_gather(ps, p)
        poses = torch.cat(ps, 0).cpu().numpy()
        c = torch.from_numpy(cloud).to(device)
        cs = [torch.empty_like(c) for _ in range(world)]
        dist.all_gather(cs, c)
        cloud = torch.cat(cs, 0).cpu().numpy()
    sync = None
    # RCCL sum; gradients are pre-scaled by 1/world_size on the device.  A sum over one rank is the identity: a world-size-1
    # launch issues no collective at all -- unless NOF_DP_FORCE=1 asks for them (tests/test_gpu_dp.py: the RCCL calls of the
    # bucketed step, exercised on the one GPU a test box has)
    if dist.is_initialized() and (world > 1 or os.environ.get('NOF_DP_FORCE') == '1'):
        from bundlesdf_amd.dist import GradSync
        # bucketed: the fine hash levels' slice is reduced beside the rest of the backward; NOF_DP_PAYLOAD=bf16: that slice travels
        # as bfloat16 (opt-in: it changes the gradient by 2^-8 relative per entry; fp32, the default, changes nothing)
        # NOF_DP_MODE=zero1: reduce-scatter -> Adam on 1/world of the flat buffers -> all-gather of the parameters (opt-in)
        sync = GradSync(payload=os.environ.get('NOF_DP_PAYLOAD', 'fp32'), mode=os.environ.get('NOF_DP_MODE', 'allreduce'))
        if os.environ.get('NOF_DP_OVERLAP', '1') == '0' and sync.mode == 'allreduce':
            sync = sync.__call__            # one blocking all-reduce of the whole buffer
    ns, nc, hidden = MLP_SHAPES[args.mlp]
    precision = args.precision
    if hidden != 64 or ns > 3 or nc > 3:        # the wide kernels have no operand split: fp16x3 / bf16x3 run as fp16 / bf16 there
        precision = {'fp16x3': 'fp16', 'bf16x3': 'bf16'}.get(precision, precision)
    import contextlib
    with contextlib.redirect_stdout(sys.stderr):    # the runner prints like the reference's does; stdout carries the ONE result line
        runner = NerfRunner(cfg, pool['rgbs'], depths=pool['depths'], masks=pool['masks'], normal_maps=None, poses=poses,
                            K=pool['K'], build_octree_pcd=synthetic.PointCloud(cloud), precision=precision, n_sigma=ns,
                            n_color=nc, world_size=world, rank=rank, grad_sync=sync, frame_offset=rank * F_local, hidden=hidden)
    return runner, cfg


def cpu_baseline(seconds=20.0, rays_per_step=4096, log2_T=19, mlp='baseline', finest=256):
    """The oracle (CPU PyTorch fp32 restatement of nerf_runner's step = "nerf_runner's CPU PyTorch path": the reference itself
    has none) on the SAME step as the GPU workload -- same rays per step, samples per ray, hash grid and MLP shape -- for a
    bounded number of steps on this box's host cores.  The pool holds 4 keyframes instead of 64: the cost of a step does not
    depend on how many rows the ray table has."""
    from bundlesdf_amd import synthetic
    from bundlesdf_amd.config import default_cfg
    from bundlesdf_amd.rays import make_frame_rays
    from oracle import nof_oracle as O
    # threads actually used: all host cores up to 32 (the oracle's small tensor ops stop scaling, and oversubscribing a
    # 100+-core host makes every op slower)
    ncores = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(ncores)
    log(f'cpu_baseline: {ncores} threads of {os.cpu_count()} cores')
    pool = synthetic.make_pool(n_frames=4, H=480, W=640, seed=0, analytic_bounds=True)
    cfg = default_cfg(n_step=500, N_rand=rays_per_step, num_levels=16, log2_hashmap_size=log2_T, finest_res=finest, base_res=16,
                      N_samples=128, N_samples_around_depth=64, far=1.0, sc_factor=pool['sc_factor'],
                      translation=pool['translation'], use_octree=1)
    occ, occ_l, max_level, level = O.build_occupancy(pool['pcd_normalized'], cfg)

    def trace_fn(o, d):
        return O.trace_rays(occ_l, o, d)[2] > 0
    rows = []
    for f in range(4):
        r = make_frame_rays(f, pool['rgbs'][f], pool['depths'][f], pool['masks'][f], pool['poses'][f], pool['K'], cfg)
        sel = np.random.default_rng(f).choice(len(r), size=min(len(r), 2 * rays_per_step), replace=False)
        r = r[sel]
        o = np.tile(pool['poses'][f][:3, 3], (len(r), 1)).astype(np.float32)
        unit = r[:, :3] / np.linalg.norm(r[:, :3], axis=-1, keepdims=True)
        d = (pool['poses'][f][:3, :3] @ unit.T).T.astype(np.float32)
        rows.append(r[trace_fn(o, d)])
    rays = np.concatenate(rows, 0).astype(np.float32)
    torch.manual_seed(0)
    geo = O.HashGeometry(16, 2, 16, log2_T, finest)
    ns, nc, hidden = MLP_SHAPES[mlp]
    field = O.OracleField(cfg, geo, O.FieldShape(num_layers=ns, num_layers_color=nc, hidden_dim=hidden, hidden_dim_color=hidden),
                          4, pool['poses'], occ_l)
    rng = np.random.default_rng(0)
    R, S = rays_per_step, 192
    times = []
    t_start = time.time()
    it = 0
    while True:
        ids = rng.choice(len(rays), size=R, replace=False)
        u1, u2 = rng.random((R, 128)).astype(np.float32), rng.random((R, 64)).astype(np.float32)
        t0 = time.time()
        field.train_step(rays[ids], u1, u2)
        dt = time.time() - t0
        if it >= 1:
            times.append(dt)
        log(f'cpu_baseline: step {it} took {dt:.2f}s')
        it += 1
        if (time.time() - t_start > seconds and len(times) >= 1) or len(times) >= 20:
            break
    med = float(np.median(times))
    return {"value": R * S / med, "unit": "ray-samples/s", "cores": ncores, "cores_total": os.cpu_count(), "kind": "port",
            "iters_per_s": 1.0 / med,
            "sample": f"{len(times)} timed steps (1 warm-up) of oracle/nof_oracle.py OracleField.train_step on the workload's own "
                      f"step: {R} rays x {S} samples, L=16 T=2^{log2_T}, MLP SDF {ns}x{hidden} + colour {nc}x{hidden}, fp32, rays of 4 "
                      f"keyframes 640x480, torch threads={ncores}"}


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks (one per GPU, RCCL) through torch.distributed.run
    and hand its exit status on.  Rank 0 prints the JSON line on the inherited stdout."""
    import socket
    import subprocess
    n_dev = torch.cuda.device_count()
    if n_dev < args.gpus:
        raise SystemExit(f'bench.py: --gpus {args.gpus} but only {n_dev} GPU(s) are visible; refusing to run fewer ranks '
                         f'than asked for (the result would claim n_gpus={args.gpus})')
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}',
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    log('self-launch: ' + ' '.join(cmd))
    raise SystemExit(subprocess.call(cmd, env=env))


def extra_configs():
    """Sub-records of the one line, so that the driver's run carries the other single-GPU BASELINE.json configurations: short runs
    of this same script in child processes (16 keyframes: a step does not depend on how many rows the ray table has).
      cfg4  configs[3]: 8192 rays per step, finest 512 + the 512^3 dense extraction (query + device marching cubes)
      cfg5  configs[4]: 1280x720 frames, T = 2^22, MLP 4x128 + 4x128, fp16, 16 384 rays per step
      cfg1_shapes  configs[0]'s shapes (1024 rays, T = 2^14, the reference's 2+3-layer network, 4 keyframes) on the GPU
      cfg2_bf16    the headline workload with bfloat16 operands (hi + lo split)"""
    import subprocess
    common = ['--keyframes', '16', '--steps', '20', '--warmup', '10', '--settle', '0', '--round-steps', '0', '--no-cpu-baseline',
              '--no-extra-configs']
    runs = [('cfg4', ['--rays', '8192', '--finest', '512', '--extract', '512']),
            # (cfg5 also times the same 20 steps again 100 steps into the run -- `ms_per_step_settled` -- where the zero-gradient
            # fraction has stopped moving: steps 10-30 from a fresh field still scatter 1.5x the settled number of rows)
            ('cfg5', ['--mlp', 'cfg5', '--rays', '16384', '--log2_T', '22', '--finest', '512', '--width', '1280', '--height', '720',
                      '--precision', 'fp16', '--settle', '100']),
            # configs[0]'s shapes on the GPU: the same workload the cpu_baseline leg of a `--rays 1024 --log2_T 14 --mlp reference`
            # run times ("also run the GPU path at cfg1 shapes for an apples-to-apples ratio"); 4 keyframes as BASELINE says
            # ... and the CPU leg at these shapes in the same sub-record (BASELINE configs[0] is "the reference's own CPU-runnable
            # case"): `cpu_baseline` of the child = the oracle's step at 1024 rays, T = 2^14, 2+3 layers, for ~8 s
            ('cfg1_shapes', ['--rays', '1024', '--log2_T', '14', '--mlp', 'reference', '--keyframes', '4', '--with-cpu-baseline',
                             '--cpu-seconds', '8']),
            # configs[1] names bf16: the headline runs the reference's own autocast type (fp16, operands split hi + lo); the same
            # workload with bfloat16 operands split the same way (parity-tested at full size like fp16x3)
            ('cfg2_bf16', ['--precision', 'bf16x3'])]
    keep = ('value', 'unit', 'ms_per_step', 'steps', 'warmup', 'dtype', 'ms_per_step_dense_backward', 'zero_grad_sample_fraction',
            'ms_per_step_settled', 'settled_after_steps', 'zero_grad_sample_fraction_settled',
            'train_iters_per_sec', 'loss', 'flags', 'step_ms_spread', 'extraction', 'cpu_baseline', 'ms_per_step_p50_timed')
    out = []
    for name, extra in runs:
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__)] + common + extra, capture_output=True, text=True, timeout=600)
            d = json.loads(r.stdout.strip().splitlines()[-1])
            rec = {"name": name, **{k: d.get(k) for k in keep if k in d}, "workload": d['config']['workload'].split(';')[0],
                   "roofline": {k: d['roofline'].get(k) for k in ('kernel', 'bound', 'frac', 'avg_ms')} if d.get('roofline') else None}
            if d.get('cpu_baseline'):                 # SURVEY 8d's apples-to-apples ratio: the same shapes on both sides
                rec["gpu_over_cpu"] = d['value'] / d['cpu_baseline']['value']
            out.append(rec)
        except Exception as ex:                       # a sub-record must never cost the main line
            out.append({"name": name, "error": repr(ex)[:300]})
    return out


def time_extraction(runner, n):
    """BASELINE cfg4's renderer half: the dense SDF query of an n^3 grid over the bounding box (octree-masked, fused encode + sigma
    net) and the device iso-surface extraction (marching cubes), on the field as trained so far; a few hundred more steps first
    if no surface exists yet."""
    import time as _t
    from bundlesdf_amd.mesh_gpu import marching_cubes_lewiner_gpu as marching_cubes_gpu     # (the runner's default extractor: skimage's method)
    fld = runner.field
    bounds = np.array(runner.cfg['bounding_box']).reshape(2, 3)
    axes = [np.linspace(bounds[0, d], bounds[1, d], n + 1)[:-1] + 0.5 * (bounds[1, d] - bounds[0, d]) / n for d in range(3)]
    rec = None
    for attempt in range(3):
        try:
            v = f = None
            for rep in range(3):                      # (the third call re-uses the first call's pinned staging buffers, released here)
                v = f = None
                torch.cuda.synchronize(); t0 = _t.perf_counter()
                vol = fld.query_sdf_grid(*axes)
                torch.cuda.synchronize(); t1 = _t.perf_counter()
                v, f = marching_cubes_gpu(vol, 0.0)
                t2 = _t.perf_counter()
            occ = float((vol != 1.0).float().mean().item())
            rec = {"grid": n, "voxels": n ** 3, "octree_fraction": round(occ, 4), "query_ms": round((t1 - t0) * 1e3, 3),
                   "marching_cubes_ms_incl_d2h": round((t2 - t1) * 1e3, 3), "vertices": int(len(v)), "triangles": int(len(f)),
                   "trained_steps": int(fld.global_step)}
            break
        except ValueError:                            # no level set yet: train on
            for _ in range(200):
                runner.train_loop()
                runner.global_step += 1
    return rec or {"grid": n, "error": "no iso-surface after training"}


ATOMIC_PEAK_G = 20.8        # G line requests/s: tools/atomic_probe.py on MI355X (profiles/r02_d_atomic_probe.txt)


def atomic_roofline(fld, runner, R, S, dom_ms):
    """Atomic line requests of the last batch's table scatter (tools/scatter_requests.py on its sample points) / launch time."""
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import scatter_requests as SR
    b = fld._buffers(R, S)
    n_rays = min(R, 1024)
    pts = b['pts_w'][:n_rays * S].cpu().numpy()
    g = fld.grid
    lds = [l for l in range(fld.L) if int(g.size[l]) * 8 <= 48 * 1024]            # accumulated in LDS, flushed once (nof_hash.hip)
    levels = [l for l in range(fld.L) if l not in lds]
    # [L, n]: the samples that reach the scatter = those of the listed tiles (dfeat of the others is not even written) ...
    live = (b['draw'][:n_rays * S] != 0).any(-1).reshape(-1, 32).any(-1).repeat_interleave(32)
    nonzero = ((b['dfeat'][:, :n_rays * S] != 0).any(-1) & live[None]).cpu().numpy()   # ... and exact zeros emit nothing
    per = SR.count_requests(pts, list(g.scale), list(g.resolution), list(g.offset), list(g.size), list(g.hashed), levels,
                            nonzero=nonzero)
    req = sum(per.values()) * (R / n_rays)
    req += sum(64 * min(int(g.size[l]) // 8, 1 << 30) for l in lds)                 # upper bound of the LDS levels' flush
    ach = req / (dom_ms * 1e-3) / 1e9
    return {"line_requests": int(req), "achieved": ach, "peak": ATOMIC_PEAK_G, "unit": "G line-requests/s",
            "frac": ach / ATOMIC_PEAK_G, "sample": f"counted on {n_rays} of the {R} rays of the last batch, scaled; zero-gradient samples "
                                                                   f"({1.0 - float(nonzero.any(0).mean()):.2f} of them) emit nothing",
            "floor_ms": req / (ATOMIC_PEAK_G * 1e9) * 1e3}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--keyframes', type=int, default=64)
    ap.add_argument('--rays', type=int, default=4096)
    ap.add_argument('--log2_T', type=int, default=19)
    ap.add_argument('--height', type=int, default=480)
    ap.add_argument('--width', type=int, default=640)
    ap.add_argument('--precision', default='fp16x3', choices=['fp16x3', 'bf16x3', 'bf16', 'fp16', 'fp32'])
    ap.add_argument('--mlp', default='baseline', choices=['baseline', 'reference', 'cfg5'],
                    help='baseline: SDF 3x64 + colour 2x64 (BASELINE.json cfg2); reference: NeRFSmall(2,3) nerf_runner.py:221; '
                         'cfg5: SDF 4x128 + colour 4x128 (BASELINE.json cfg5, with --rays 16384 --log2_T 22 --width 1280 --height 720 '
                         '--precision fp16)')
    ap.add_argument('--finest', type=int, default=256, help='finest hash resolution (256: cfg1-3; 512: cfg4/5)')
    ap.add_argument('--settle', type=int, default=200, help='steps into the run at which the K steps are timed AGAIN for ms_per_step_settled (the zero-gradient fraction has settled by then); 0: skip.  `value` is always the K steps right after the W warm-up steps')
    ap.add_argument('--round-steps', type=int, default=501, help='steps of the whole-round measurement from a fresh field (the reference\'s N_iters = n_step + 1 = 501); 0: skip')
    ap.add_argument('--preroll', type=int, default=0, help='forward-only batches before the warm-up (no training state is touched); measured to change nothing (DESIGN 5), off by default')
    ap.add_argument('--unfused', action='store_true', help='training forward as nof_hash_encode_fwd + nof_mlp_fwd (fp32 embedding in HBM) instead of the fused nof_encode_mlp_fwd')
    ap.add_argument('--no-extra-configs', action='store_true', help='skip the BASELINE cfg4 / cfg5 sub-records (N = 1 default run only)')
    ap.add_argument('--extract', type=int, default=0, help='after the timed steps: time the dense SDF query + marching cubes of an N^3 grid (BASELINE cfg4: 512)')
    ap.add_argument('--one-stream', type=int, default=None, choices=[0, 1],
                    help='the backward tail as one chain with the scatter and dL/dx in one launch (field.one_stream_backward)')
    ap.add_argument('--fused-tail', type=int, default=None, choices=[0, 1],
                    help='the optimiser launch with the pose sums and the next step\'s operand image + pose table inside (field.fused_tail)')
    ap.add_argument('--mlp-bwd-one-launch', type=int, default=None, choices=[0, 1],
                    help='the 16-bit MLP backward\'s two halves as one launch (field.mlp_bwd_one_launch)')
    ap.add_argument('--march-ahead', type=int, default=None, choices=[0, 1],
                    help='the next batch\'s ray marcher inside the optimiser launch (field.march_ahead; measured slower, off by default)')
    ap.add_argument('--event-stride', type=int, default=4,
                    help='timed region: HIP events around the dominant launch in every Nth step only (a timing event recorded on the '
                         'stream costs the queue a ~6 us bubble: profiles/r06_am_gap_probe.txt)')
    ap.add_argument('--marker-stride', type=int, default=4, help='timed region: one step-boundary event every Nth step')
    ap.add_argument('--host-trace', action='store_true', help='diagnosis: host time of every C-ABI call of the first three timed steps (log)')
    ap.add_argument('--scatter-wgs', type=int, default=0, help='persistent workgroups per CU of the table scatter (0 = library default)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--with-cpu-baseline', action='store_true', help='overrides --no-cpu-baseline (the cfg1-shapes sub-record)')
    ap.add_argument('--trace-kernel', type=int, default=None, help='ray marcher of the step (NofSampleCfg.marcher): 0 = one lane per ray (walk), 1 = one wave per ray (the default)')
    ap.add_argument('--cpu-seconds', type=float, default=25.0)
    args = ap.parse_args()

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        return self_launch(args)
    import torch.distributed as dist
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X (no CPU fallback for the product path)')
    dev_index = local_rank % torch.cuda.device_count()      # (== local_rank on a real node; the 1-GPU gloo test wraps around)
    torch.cuda.set_device(dev_index)
    device = torch.device('cuda', dev_index)
    if world > 1 or ('RANK' in os.environ and 'MASTER_PORT' in os.environ):     # launched by torch.distributed.run
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        backend = os.environ.get('NOF_DIST_BACKEND', 'nccl')                    # 'nccl' is RCCL; 'gloo' only for the 1-GPU test
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=device)
        else:
            dist.init_process_group(backend)
    if world != args.gpus:
        raise SystemExit(f'bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: the two must agree (n_gpus in the result is WORLD_SIZE)')

    log(f'rank {rank}/{world}: building the keyframe pool and ray table')
    runner, cfg = build_runner(args, rank, world, device)
    fld = runner.field
    fld.scatter_wgs_per_cu = args.scatter_wgs
    log(f'pool ready: {runner.rays.shape[0]} rays, level {fld.level}, table {fld.n_entries} rows')
    R, S = args.rays, cfg['N_samples'] + cfg['N_samples_around_depth']
    B = R * S

    if args.one_stream is not None:
        fld.one_stream_backward = bool(args.one_stream)
    if args.fused_tail is not None:
        fld.fused_tail = bool(args.fused_tail)
    if args.march_ahead is not None:
        fld.march_ahead = bool(args.march_ahead)
    if args.mlp_bwd_one_launch is not None:
        fld.mlp_bwd_one_launch = bool(args.mlp_bwd_one_launch)
    if args.unfused:
        fld.fused_forward = False
        fld.fused_forward_wide = False
    if args.trace_kernel is not None:
        fld.marcher = 1 - int(args.trace_kernel)             # (the option's 1 = wave per ray = lib.MARCHER_WAVE = 0)
    fwd_name = ('nof_encode_mlp_fwd' if fld.fused_forward else 'nof_encode_mlp_wide_fwd' if fld.wide and fld.fused_forward_wide
                else 'nof_hash_encode_fwd')     # the launch that holds the hash lookup

    def zero_fraction():
        """ray-samples of the last batch whose loss gradient is exactly zero (one device reduction + host sync: outside timing)"""
        return float((fld._buffers(R, S)['draw'] == 0).all(-1).float().mean().item())

    trace_steps = os.environ.get('NOF_BENCH_TRACE_STEPS') == '1'     # diagnosis: loss and parameter sums after every step (host syncs)

    # test hook (tests/test_gpu_dp.py): NOF_DP_INJECT_OVERFLOW="rank:step" -- on that rank, in that step (0-based, warm-up included),
    # the fp16 loss scale of the MLP backward is raised by 2^40, so its weight gradient overflows on THAT rank only
    inject = os.environ.get('NOF_DP_INJECT_OVERFLOW')
    inject = tuple(int(x) for x in inject.split(':')) if inject else None

    def step():
        """one iteration of NerfRunner.train()"""
        if inject is not None and inject == (rank, int(fld.global_step)):
            keep, fld._scale_backoff = fld._scale_backoff, fld._scale_backoff - 40
            runner.train_loop()
            fld._scale_backoff = keep
        else:
            runner.train_loop()
        runner.global_step += 1
        if trace_steps:
            print(f"[step {fld.global_step}] loss {fld.losses()['loss']:.7f} table {float(fld.table.double().abs().sum().item()):.4f} "
                  f"mlp {float(fld.mlp.double().abs().sum().item()):.5f} tiles {fld.backward_tiles} ids {runner.data_loader.batch_ray_ids[:4].tolist()} "
                  f"pos {runner.data_loader.pos} lr {fld.learning_rates()} flags {int(fld.flags[0].item())} scale {fld.desc.grad_scale} "
                  f"trunc {fld.truncation():.6f} pose {float(fld.pose.double().abs().sum().item()):.6f} "
                  f"m {float(fld.exp_avg.double().abs().sum().item()):.6f} v {float(fld.exp_avg_sq.double().sum().item()):.8f} "
                  f"g {float(fld.grads.double().abs().sum().item()):.6f} "
                  + " ".join(f"{k} {float(v.double().abs().sum().item()):.7f}" for k, v in
                             (('tf', fld.tf), ('packed', fld.packed.view(torch.int16).float()), ('raw', fld._buffers(R, S)['raw']),
                              ('rgb', fld._buffers(R, S)['rgb_map']), ('z', fld._buffers(R, S)['z_vals']), ('pts', fld._buffers(R, S)['pts_w']),
                              ('draw', fld._buffers(R, S)['draw']), ('dfeat', fld._buffers(R, S)['dfeat']),
                              ('dview', fld._buffers(R, S)['dview']), ('gray', fld._buffers(R, S)['g_ray']),
                              ('lossrows', fld._buffers(R, S)['loss_rows']), ('view', fld._buffers(R, S)['view'])))
                  + f" ntiles {int(fld._buffers(R, S)['tiles'][:4].view(torch.int32)[0].item())} lossvec {fld.loss_out.tolist()}",
                  file=sys.stderr, flush=True)

    # ---- optional pre-roll: forward-only batches (NerfRunner.render_images' path), no parameter / optimiser / loader / RNG state touched.
    # Built on the suspicion that the driver's 20 timed steps after 5 warm-up steps (0.52 ms) ran on a GPU whose clocks had not ramped;
    # measured, it changes nothing (0.5232 without, 0.5215 / 0.516 with 300 batches): those are the first steps of a fresh field, where
    # half of the samples still carry a loss gradient (0.50 -> 0.65 zero-gradient fraction over these 20 steps) and the backward kernels
    # have 1.5x the settled work.  Off by default.
    if args.preroll > 0:
        ids0 = torch.arange(R, device=device) % runner.rays.shape[0]
        for _ in range(args.preroll):
            fld.render_batch(runner.rays, ids0, R)
        torch.cuda.synchronize()
    # ---- warm-up with every launch bracketed by events: finds the dominant kernel --------------------------------
    fld.profile = {}
    sync = runner.grad_sync if hasattr(runner.grad_sync, 'finish') else None
    dp_mode = getattr(runner.grad_sync, 'mode', None) if runner.grad_sync is not None else None
    # The LAST two warm-up steps run right in front of the timed region's barrier, after this analysis (and after the interpreter's
    # garbage collection): the first timed step of rounds 3-5 took 0.63-0.65 ms where the others take 0.44 -- the HOST needed 0.29-0.43
    # ms to enqueue it instead of 0.08 (a full heap walk and an idle wait had just emptied the CPU's caches), and the device waited for
    # its launches (--host-trace; profiles/r06_ap_first_step.txt).  W warm-up steps in all, as before.
    n_prof = args.warmup - 2 if args.warmup >= 4 else args.warmup
    for _ in range(n_prof):
        step()
    torch.cuda.synchronize()
    log('warm-up (profiled part) done')
    # median over the profiled warm-up steps, the first two left out (the first one if there are only three): a kernel's first launch
    # loads its code object (with the mean, a 5-microsecond kernel that happened to go first looked like the longest launch of a
    # 5-step warm-up)
    ktimes = fld.kernel_times_ms(stat='median', skip=2 if n_prof > 3 else max(n_prof - 2, 0))
    # (no warm-up step to look at: the launch that is the longest in every committed trace of this workload)
    dominant = max(ktimes, key=ktimes.get) if ktimes else 'hash_bwd[table+table_lds]'
    fld.profile = {dominant: []} if dominant else None          # timed region: only the dominant kernel keeps its events
    fld.profile_only = dominant
    # A timing event recorded on the stream is not free under the profiler: the queue idles ~5.8 us around each (tools/gap_probe.py,
    # profiles/r06_am_gap_probe.txt: dependent kernels run back to back without one, 5.8 us apart with one between them).  Rounds 3-5
    # recorded five per step inside the timed region (two around the dominant launch, two around the forward, one step marker):
    # 29 us of every step in a rocprofv3 trace -- exactly the "gaps" of the profiled timelines of those rounds.  WITHOUT the profiler
    # the cost is small (profiles/r06_an_events.txt: 3 events per step against 0.75: 0.445-0.450 against 0.437-0.451 ms), but it is
    # not work of the step: the dominant launch keeps its events on every `event_stride`-th step of the region, the hash lookup
    # (north_star: ">= 40 % HBM roofline for hash lookup") is timed in the spread pass right behind the region, the step markers
    # come every `marker_stride` steps.
    ev_stride = max(int(args.event_stride), 1) if args.steps >= 4 * max(int(args.event_stride), 1) else 1    # (>= 4 samples)
    mk_stride = max(int(args.marker_stride), 1) if args.steps >= 4 * max(int(args.marker_stride), 1) else 1
    fld.profile_stride = ev_stride
    if sync is not None:
        sync.timing, sync.timed_steps = [], 0

    def barrier():
        if dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize()

    def timed(n, pre=2):
        # `pre` UNTIMED steps between the garbage collection and the barrier (see the warm-up above: the host enqueues its first
        # step after a heap walk three to five times slower than the others); the main region's are its last warm-up steps.
        # (no cyclic garbage collection inside the timed region: a generation-2 pass of the interpreter is milliseconds of host
        # time, i.e. ten steps' worth.  One driver-style run in eleven read 0.73 instead of 0.47-0.48 ms/step -- a one-off host stall
        # of unknown origin, profiles/r05_final4_*, r05_x_headline_repeat.txt -- and this is the one such pause the script can rule
        # out; collected right before instead)
        import gc
        gc.collect()
        gc.disable()
        marks = []
        ms_ = mk_stride if n == args.steps else 1
        at = [i for i in range(n + 1) if i % ms_ == 0 or i == n]       # step boundaries that get a marker (0 and n always)
        evs = {i: torch.cuda.Event(enable_timing=True) for i in at}    # (created outside the timed region)
        try:                                                           # (an exception in step() must not leave the collector off)
            for _ in range(pre):
                step()
            barrier()
            t0 = time.perf_counter()
            evs[0].record()
            for i in range(n):
                if args.host_trace and n == args.steps and i < 3:
                    from bundlesdf_amd import lib as _lib
                    real, rec = _lib.call, []

                    def traced(name, *a, **k):
                        ta = time.perf_counter()
                        real(name, *a, **k)
                        rec.append((name, (ta - t0) * 1e6, (time.perf_counter() - ta) * 1e6))
                    _lib.call = traced
                    ts = time.perf_counter()
                    step()
                    _lib.call = real
                    log(f'host trace, timed step {i}: step() {1e6 * (time.perf_counter() - ts):.0f} us; calls (name, start us after t0, host us): '
                        + ', '.join(f'{nm} {a_:.0f} {d_:.0f}' for nm, a_, d_ in rec))
                    if i + 1 in evs:
                        evs[i + 1].record()
                    marks.append(time.perf_counter())
                    continue
                step()
                if i + 1 in evs:
                    evs[i + 1].record()                                # (a marker on the step's stream: ~1 us of host time, a ~6 us bubble in the queue)
                marks.append(time.perf_counter())                      # (50 ns: when the host finished enqueueing the step)
            barrier()
            dt = time.perf_counter() - t0
        finally:
            gc.enable()
        # DEVICE-side duration of every step of the timed region (event to event on the step's stream): their median and maximum
        # make a one-off stall inside the region visible in the record itself (round 5: one driver-style run in eleven read a MEAN of
        # 0.73 instead of 0.47 ms)
        dv = np.array([evs[a].elapsed_time(evs[b]) / (b - a) for a, b in zip(at[:-1], at[1:])]) if n else np.zeros(0)
        timed.device_intervals = ({"p50": float(np.median(dv)), "max": float(dv.max()), "n": int(len(dv)), "steps_per_interval": ms_,
                                   "all": [round(float(x), 4) for x in dv]} if len(dv) else None)
        # ... and the HOST's enqueue intervals (the host runs ahead of the device: these say how long a step takes to ENQUEUE)
        iv = np.diff(np.array([t0] + marks)) * 1e3
        timed.host_intervals = {"p50": float(np.median(iv)), "max": float(iv.max()), "n": int(len(iv))} if len(iv) else None
        if world > 1:
            t = torch.tensor([dt], device=device, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    # ---- THE timed region: exactly K steps right after the W warm-up steps (the contract) ----------------------------------
    # The step's cost depends on how many ray-samples carry a loss gradient (the backward runs over their tiles only); that
    # fraction moves over the first few hundred steps of a run (0.49 -> 0.67 at cfg2).  `value` is what these K steps deliver;
    # the same K steps once the fraction has settled, with every tile listed, and the mean over a whole reference round
    # (501 steps from a fresh field, config.yml:2) are reported beside it under their own names.
    zero_first = zero_fraction() if args.warmup > 0 else None
    dt = timed(args.steps, pre=args.warmup - n_prof)
    timed_intervals, host_intervals = timed.device_intervals, timed.host_intervals
    log(f'timed region done: {dt / args.steps * 1e3:.3f} ms/step; device step intervals {timed_intervals}; host enqueue intervals {host_intervals}')
    # The parameters after exactly W + K steps: what the tests compare between the forms of the data-parallel step.  Taken HERE and
    # not at the end of the run: Adam with eps = 1e-15 (the reference's) turns the first rounding-noise gradient of a so far dead
    # weight into a full +-lr step, and an MLP weight that wakes up that way moves everything downstream -- 3 of 16 otherwise
    # identical 26-step runs ended 4e-4 away from the other 13 in sum |p| (all 3 in the same place), parting at step 14
    # (profiles/r04_q_rccl_states.txt); after 8 steps the runs agree to 1e-6.
    checksum = float(fld.params.double().abs().sum().item())
    # (per segment of the flat buffer: tells which of table / MLP / frame features / poses two runs disagree on)
    checksum_parts = {k: float(getattr(fld, k).double().abs().sum().item()) for k in ('table', 'mlp', 'feat', 'pose')}
    checksum_parts['steps_taken'] = int(fld.global_step)
    checksum_parts['adam_steps'] = int(fld.adam_steps)
    checksum_parts['loss_scale_backoff'] = int(fld._scale_backoff)       # > 0: a step overflowed in the 16-bit backward and was skipped
    zero_last = zero_fraction()
    kt = fld.kernel_times_ms()
    dom_ms = kt.get(dominant) if dominant else None
    hash_fwd_ms = kt.get(fwd_name)                # (the forward IS the dominant launch; otherwise timed in the spread pass below)
    exposed_comm_ms = None
    if sync is not None and sync.timing:
        torch.cuda.synchronize()
        exposed_comm_ms = float(np.sum([a.elapsed_time(b) for a, b in sync.timing])) / max(sync.timed_steps, 1)
        sync.timing = None
    fld.profile, fld.profile_only, fld.profile_also, fld.profile_stride = None, None, None, 1
    timed_events_per_step = (2.0 / ev_stride if dominant else 0.0) + 1.0 / mk_stride
    # ---- spread of the step time: K more steps with one event after each (device-side durations, no host sync in between) ----
    # (the same steps also time the forward -- the hash lookup's own roofline entry -- with two events per step)
    if dominant != fwd_name:
        fld.profile, fld.profile_only = {fwd_name: []}, fwd_name
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    barrier()
    evs[0].record()
    for i in range(args.steps):
        step()
        evs[i + 1].record()
    barrier()
    per_step = np.array([evs[i].elapsed_time(evs[i + 1]) for i in range(args.steps)])
    if dominant != fwd_name:
        hash_fwd_ms = fld.kernel_times_ms().get(fwd_name)
        fld.profile, fld.profile_only = None, None
    # (the events are recorded when the step's last launch retires; a step that found the queue empty shows its host time instead)
    spread = {"p10": float(np.percentile(per_step, 10)), "p50": float(np.percentile(per_step, 50)),
              "p90": float(np.percentile(per_step, 90)), "n": int(args.steps)} if args.steps >= 5 else None
    # ---- the same K steps once the zero-gradient fraction has settled (>= --settle steps into the run) ----
    settled_ms, zero_settled, settled_after = None, None, None
    if args.settle > 0:
        more = max(0, args.settle - fld.global_step)
        for _ in range(more):
            step()
        settled_after = fld.global_step
        zero_settled = zero_fraction()
        settled_ms = timed(args.steps) / args.steps * 1e3
        log(f'settled ({settled_after} steps into the run, zero-gradient fraction {zero_settled:.3f}): {settled_ms:.3f} ms/step')
    # The same K steps with EVERY tile in the backward's work list (nothing skipped, same kernels): what the step costs when no
    # loss gradient is zero -- the sparsity-independent figure.
    fld.backward_tiles = 'all'
    timed(2, pre=0)
    dense_ms = timed(args.steps) / args.steps * 1e3
    fld.backward_tiles = 'list'
    log(f'dense backward (every tile listed): {dense_ms:.3f} ms/step')
    # cfg hip_graph = True replays the step as ONE captured HIP graph (NerfRunner.train_loop / GraphedStep): one chain -- the table
    # scatter and dL/dx as two roles of one launch, the pose rows as a passenger of the LDS levels' launch (round 6) -- instead of
    # the eager step's two streams.  Same K steps, captured, for the record:
    graph_ms, graph_alt_ms = None, None
    if not dist.is_initialized() or world == 1:
        runner.cfg['hip_graph'] = True              # opt-in (the product default is the eager two-stream step)
        for _ in range(4):
            step()
        if getattr(runner, '_graph', None) is not None:
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                step()
            torch.cuda.synchronize()
            graph_ms = (time.perf_counter() - t1) / args.steps * 1e3
            log(f'captured-step mode: {graph_ms:.3f} ms/step')
        # the same with the eager step's two branches kept inside the graph (round 5's captured step): fork / join against none
        keep_modes = (fld.graph_fork, fld.one_stream_backward)
        fld.graph_fork, fld.one_stream_backward = True, False
        runner._graph = None
        for _ in range(4):
            step()
        if getattr(runner, '_graph', None) is not None:
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                step()
            torch.cuda.synchronize()
            graph_alt_ms = (time.perf_counter() - t1) / args.steps * 1e3
            log(f'captured-step mode, two branches inside the graph: {graph_alt_ms:.3f} ms/step')
        fld.graph_fork, fld.one_stream_backward = keep_modes
        runner.cfg['hip_graph'] = False
        runner._graph = None
    extraction = time_extraction(runner, args.extract) if args.extract > 0 and world == 1 else None
    flags = int(fld.flags[0].item())
    losses = fld.losses()
    dp_spread = None
    if dist.is_initialized():               # replicas must hold bit-identical parameters after K synchronised steps
        chk = torch.stack([fld.params.double().sum(), fld.params.double().abs().sum()]).to(device)
        allc = [torch.empty_like(chk) for _ in range(dist.get_world_size())]
        dist.all_gather(allc, chk)
        allc = torch.stack(allc)
        dp_spread = float((allc.max(0).values - allc.min(0).values).abs().max().item())
    # ---- one whole round of the reference: N_iters = n_step + 1 = 501 steps from a FRESH field (config.yml:2, nerf_runner.py:160,
    #      855-863) with its learning-rate schedule; mean step time over the round, early (denser) steps included ----
    round_ms, round_zero = None, None
    if args.round_steps > 0:
        keep_step = runner.cfg['n_step']
        runner.cfg['n_step'] = args.round_steps - 1
        runner.N_iters = args.round_steps
        runner.create_nerf()
        runner.create_optimizer()
        runner.global_step = 0
        fld_r = runner.field
        fld_r.scatter_wgs_per_cu = args.scatter_wgs
        fld_r.fused_forward = fld.fused_forward
        fld_r.fused_forward_wide = fld.fused_forward_wide
        fld_r.one_stream_backward = fld.one_stream_backward
        fld_r.fused_tail = fld.fused_tail
        fld_r.mlp_bwd_one_launch = fld.mlp_bwd_one_launch
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.round_steps):
            step()
        barrier()
        dt_r = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt_r], device=device, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt_r = float(t.item())
        round_ms = dt_r / args.round_steps * 1e3
        round_zero = float((fld_r._buffers(R, S)['draw'] == 0).all(-1).float().mean().item())
        runner.cfg['n_step'] = keep_step
        log(f'reference round ({args.round_steps} steps from a fresh field): {round_ms:.3f} ms/step, final loss {fld_r.losses()["loss"]:.5f}')

    if rank == 0:
        ms = dt / args.steps * 1e3
        it_s = args.steps / dt
        value = world * B * it_s
        # algorithmic work per launch of each kernel (DESIGN.md "Kernels")
        zero_frac = zero_last
        n_mlp = fld.n_mlp
        fl_fwd = 2.0 * (n_mlp - sum(o for o, _ in fld.layer_dims))      # 2*MAC per sample
        fl_net = [2.0 * sum(o * i for o, i in fld.layer_dims[:fld.n_sigma]), 2.0 * sum(o * i for o, i in fld.layer_dims[fld.n_sigma:])]
        gather_bytes = B * (16 * 8 * 2 * 4 + 12)                          # L * 2^D * C * 4 table bytes + the point
        hash_fwd_bytes = gather_bytes + B * 16 * 2 * 4                    # + the fp32 embedding written by the stand-alone encoder
        # the fused forward writes raw (16 B), the sigma hand-off (32 B) and the operand-precision feature copy (64 B) per sample
        fused_bytes = gather_bytes + B * (16 + 32 + 64)
        work = {
            'nof_hash_encode_fwd': ('hbm', hash_fwd_bytes),
            'nof_encode_mlp_fwd': ('hbm', fused_bytes),
            # dfeat read (L*C*4 = 64 B at the reference's fp16 gradient... 128 B for this fp32 dfeat; 8d prices 64) +
            # atomic read-modify-write of 8 corners x 2 channels per level (2*L*8*2*4 = 2048 B) = 2112 B per sample, both only
            # for the samples the work list keeps: the others are not read and would add 0 -- pricing them would credit work that
            # is not done (frac > 1).  This is the table scatter (the run-merged global atomics of the large levels + the
            # LDS-accumulated small level behind them); dL/dx (k_hash_dx, 1048 B/sample) runs beside it on the step's second
            # stream and is its own entry.
            # (+ the MLP backward's partial rows, whose reduction rides inside the LDS levels' launch in the one-GPU step)
            'hash_bwd[table+table_lds]': ('hbm', B * (1.0 - zero_frac) * 2112.0 +
                                          (fld.nblk * fld.n_mlp * 4.0 if not fld.wide and not fld.eikonal and runner.grad_sync is None else 0.0)),
            'hash_bwd[input]': ('hbm', B * (1.0 - zero_frac) * (16 * 2 * 4 + 16 * 8 * 2 * 4 + 12) + B * 12),
            'nof_mlp_fwd': ('mfma', B * fl_fwd),
            'nof_mlp_bwd_tiles': ('mfma', B * (1.0 - zero_frac) * 3.0 * fl_fwd),
            'nof_mlp_wide_fwd': ('mfma', B * fl_fwd),
            # the wide backward (one kernel per network, round 6): forward recompute + data gradients + weight gradients of the listed
            # tiles = 3 x the forward's MACs (the transposing MFMAs that feed the weight gradient are not counted as useful work)
            'nof_mlp_wide_bwd': ('mfma', B * (1.0 - zero_frac) * 3.0 * fl_fwd),
            'nof_adam_step': ('hbm', fld.n_total * 32.0),
        }
        traffic = None          # HBM bytes per launch from the committed PMC passes (profiles/pmc_traffic.json), same workload only
        hash_fwd_traffic = None
        try:
            if args.mlp == 'baseline' and R == 4096 and args.log2_T == 19:
                pm = json.load(open(os.path.join(ROOT, 'profiles', 'pmc_traffic.json')))
                traffic = pm.get(dominant, {}).get('traffic_bytes')
                hash_fwd_traffic = pm.get(fwd_name, {}).get('traffic_bytes')
        except Exception:
            traffic = None
        src = ("profiles/pmc_traffic.json (rocprofv3 --pmc passes of this workload, committed; not re-measured in this run)")
        roof = None
        if dominant in work and dom_ms:
            kind, amount = work[dominant]
            if kind == 'hbm':
                ach = amount / (dom_ms * 1e-3) / 1e9
                roof = {"kernel": dominant, "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": ach / HBM_PEAK_GBS, "traffic": traffic, "algorithmic_bytes": amount, "avg_ms": dom_ms,
                        "traffic_source": src if traffic else None}
                if dominant == 'hash_bwd[table+table_lds]':
                    # what actually bounds the scatter: the memory side retires ~20.8 G atomic LINE REQUESTS per second
                    # (tools/atomic_probe.py; same for every scope, cache flag and data type), and the launch needs one request per
                    # 64-byte line per atomic instruction.  The requests of THIS batch are counted from its own sample points.
                    try:
                        at = atomic_roofline(fld, runner, R, S, dom_ms)
                        # the ceiling of this launch IS the atomic request rate (counter traffic is 0.14x the algorithmic bytes: the
                        # read-modify-write happens memory-side): the line's primary figures are against it, the HBM pricing of
                        # SURVEY 8d's algorithmic bytes is kept under "hbm"
                        roof = {"kernel": dominant, "bound": "atomic", "achieved": at["achieved"], "peak": at["peak"], "unit": at["unit"],
                                "frac": at["frac"], "traffic": traffic, "avg_ms": dom_ms, "line_requests": at["line_requests"],
                                "floor_ms": at["floor_ms"], "sample": at["sample"], "traffic_source": src if traffic else None,
                                "call": ("the call's two launches: { large levels' scatter | dL/dx } and { LDS levels | MLP row reduction | "
                                         "per-ray pose rows } (round 6: one chain); avg_ms is the whole call, the requests are the scatter's"
                                         if fld.one_stream_backward else "scatter launch + LDS-level launch (dL/dx beside them on a second stream)"),
                                "peak_source": "tools/atomic_probe.py on MI355X: 20.8 G fp32-atomic line requests/s, whatever the scope, "
                                               "cache flags or data type (profiles/r02_d_atomic_probe.txt)",
                                "hbm": {"achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                                        "algorithmic_bytes": amount,
                                        "traffic_frac_of_peak": (traffic / (dom_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic else None}}
                    except Exception as ex:          # measurement aid only
                        roof["atomic"] = {"error": repr(ex)}
            else:
                ach = amount / (dom_ms * 1e-3) / 1e12
                roof = {"kernel": dominant, "bound": "mfma", "achieved": ach, "peak": MFMA_BF16_PEAK_TF, "unit": "TFLOP/s",
                        "frac": ach / MFMA_BF16_PEAK_TF, "traffic": traffic, "algorithmic_flop": amount, "avg_ms": dom_ms}
        elif dominant:
            roof = {"kernel": dominant, "bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None,
                    "traffic": None, "avg_ms": dom_ms}
        if roof is not None and hash_fwd_ms:
            # The hash lookup, three ways: SURVEY 8d's algorithmic bytes count every corner read as memory traffic (L2 hits
            # included: against HBM this fraction can exceed 1, so it is also put against the L2's own rate, where the 36 MB table
            # mostly lives), and the memory-side figure is what the PMC counters saw leave / enter HBM.
            fwd_bytes = fused_bytes if fld.fused_forward else hash_fwd_bytes
            ach = fwd_bytes / (hash_fwd_ms * 1e-3) / 1e9
            roof["hash_fwd"] = {"kernel": fwd_name, "avg_ms": hash_fwd_ms, "algorithmic_bytes": fwd_bytes,
                                "achieved_algorithmic": ach, "frac_algorithmic": ach / HBM_PEAK_GBS,
                                "frac_of_l2_peak": ach / L2_PEAK_GBS, "l2_peak": L2_PEAK_GBS,
                                "traffic": hash_fwd_traffic,
                                "frac_memory_side": (hash_fwd_traffic / (hash_fwd_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if hash_fwd_traffic else None,
                                "traffic_source": src if hash_fwd_traffic else None, "unit": "GB/s", "peak": HBM_PEAK_GBS,
                                "note": "bound by the vector-memory instruction rate (92 gather instructions per 64 samples), not by HBM"}
            if fld.fused_forward:                       # the same launch holds both MLPs: its matrix-core side
                roof["hash_fwd"]["mlp_tflops_useful"] = B * fl_fwd / (hash_fwd_ms * 1e-3) / 1e12
                roof["hash_fwd"]["mlp_frac_of_mfma_peak"] = roof["hash_fwd"]["mlp_tflops_useful"] / MFMA_BF16_PEAK_TF
        shape_key = (args.keyframes, R, args.log2_T, args.mlp, args.width, args.height)
        cfg_name = {(64, 4096, 19, 'baseline', 640, 480): 'cfg2' if world == 1 else 'cfg3',
                    (4, 1024, 14, 'reference', 640, 480): 'cfg1 shapes',
                    (64, 16384, 22, 'cfg5', 1280, 720): 'cfg5 (per GPU)'}.get(shape_key, 'custom')
        if R == 8192 and args.finest == 512 and args.mlp == 'baseline':
            cfg_name = 'cfg4 shapes'
        elif args.mlp == 'cfg5' and R == 16384 and args.log2_T == 22:
            cfg_name = 'cfg5 shapes'  if args.keyframes != 64 else cfg_name
        out = {
            "metric": "ray_samples_per_sec", "value": value, "unit": "ray-samples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": {'fp16x3': 'fp16', 'bf16x3': 'bf16'}.get(runner.precision, runner.precision), "data": "synthetic",
            "config": {"workload": f"{cfg_name}: {args.keyframes} synthetic {args.width}x{args.height} RGBD keyframes per GPU, "
                                   f"{R} rays/step x {S} samples, hash L=16 T=2^{args.log2_T} base16->{args.finest}, "
                                   f"MLP SDF {MLP_SHAPES[args.mlp][0]}x{MLP_SHAPES[args.mlp][2]} + colour "
                                   f"{MLP_SHAPES[args.mlp][1]}x{MLP_SHAPES[args.mlp][2]}, "
                                   f"{PRECISION_NOTE[runner.precision]}, fp32 table/accumulators/Adam; every sample runs "
                                   f"the forward and the loss, the backward runs over the work list of the 32-sample tiles that hold a "
                                   f"non-zero loss gradient (the others add exactly nothing: same sums; fraction of zero samples in "
                                   f"zero_grad_sample_fraction, the step with every tile listed in ms_per_step_dense_backward); "
                                   f"`value` = the {args.steps} steps right after the {args.warmup} warm-up steps of a fresh field "
                                   f"-- the most expensive steps of a round: half of the samples still carry a loss gradient there; the "
                                   f"same steps later in the run: ms_per_step_settled; a whole {args.round_steps}-step reference "
                                   f"round from a fresh field: round_ms_per_step",
                       "rays_per_step": R, "samples_per_ray": S, "keyframes_per_gpu": args.keyframes,
                       "pool_rays": int(runner.rays.shape[0]), "parallelism": f"dp{world}",
                       "forward": ("fused encode+MLP (nof_encode_mlp_fwd)" if fld.fused_forward else
                                   "encode + sigma net in one launch, colour net behind it (nof_encode_mlp_wide_fwd)" if fld.wide and fld.fused_forward_wide
                                   else "nof_hash_encode_fwd + nof_mlp_fwd")},
            # the timed region's per-step DEVICE durations (event to event; median / largest): a mean far above the median = a one-off
            # stall inside the region, not the steady step.  host_enqueue_ms_p50: how long the host takes to enqueue a step
            "ms_per_step_p50_timed": timed_intervals["p50"] if timed_intervals else None,
            "ms_step_max_timed": timed_intervals["max"] if timed_intervals else None,
            "timing_events": {"per_event_queue_bubble_us_under_rocprofv3": 5.8, "source": "tools/gap_probe.py, profiles/r06_am_gap_probe.txt "
                              "(without the profiler: profiles/r06_an_events.txt)",
                              "dominant_launch_event_stride": ev_stride, "step_marker_stride": mk_stride,
                              "events_per_timed_step": timed_events_per_step},
            "host_enqueue_ms_p50": host_intervals["p50"] if host_intervals else None,
            "train_iters_per_sec": it_s * 1.0, "captured_step_ms_per_step": graph_ms, "captured_step_two_branches_ms_per_step": graph_alt_ms,
            # forward-only batches run before the warm-up steps (no parameter / optimiser / loader / RNG state touched)
            "preroll_forward_batches": args.preroll,
            # device-side duration of single steps (K more steps, one event after each): p10 / p50 / p90
            "step_ms_spread": spread,
            # the reference's unit of work: one round = N_iters steps from a fresh field (config.yml:2); mean over the whole round
            "round_steps": args.round_steps, "round_ms_per_step": round_ms,
            "round_ray_samples_per_sec": (world * B / (round_ms * 1e-3)) if round_ms else None,
            "round_zero_grad_sample_fraction_last_step": round_zero,
            # the same K steps `settled_after_steps` steps into the run, where the zero-gradient fraction has stopped moving
            "ms_per_step_settled": settled_ms, "settled_after_steps": settled_after, "zero_grad_sample_fraction_settled": zero_settled,
            # same run, same kernels, every tile of the batch in the backward's work list (no sparsity): DESIGN 2.9
            "ms_per_step_dense_backward": dense_ms, "value_dense_backward": world * B / (dense_ms * 1e-3),
            "kernel_ms_warmup": {k: round(v, 4) for k, v in sorted(ktimes.items(), key=lambda kv: -kv[1])},
            "valid_sample_fraction": losses['n_valid_samples'] / B,     # samples inside [-1,1]^3 (the rest still run the MLPs)
            # ray-samples whose loss gradient is exactly zero in the last batch (background rays, saturated free-space samples):
            # the backward skips 32- / 64-sample tiles made of them (exact: they contribute nothing)
            "zero_grad_sample_fraction": zero_frac,
            "zero_grad_sample_fraction_first_timed_step": zero_first, "zero_grad_sample_fraction_last_timed_step": zero_last,
            "loss": losses['loss'], "flags": flags, "dp_param_checksum_spread": dp_spread, "param_checksum": checksum,
            "param_checksum_parts": checksum_parts,
            # data parallel (N > 1): gradient bytes each rank hands to RCCL per step, in how many collectives, and how long the
            # step's stream waited for them after the backward (events around GradSync.finish: what did not hide)
            "allreduce_bytes_per_step": (sync.bytes_step if sync is not None else (fld.n_total * 4 if runner.grad_sync is not None else 0)),
            "collectives_per_step": (sync.collectives_step if sync is not None else (1 if runner.grad_sync is not None else 0)),
            "exposed_comm_ms": exposed_comm_ms, "dp_payload": getattr(sync, 'payload', None), "dp_mode": dp_mode,
            # GradSync mode 'rows': table rows that travelled in the last step (the union of the ranks' non-zero rows) of how many
            "dp_rows_per_step": getattr(sync, 'rows_step', None), "dp_table_rows": int(fld.n_table // 2),
            "roofline": roof,
        }
        if extraction is not None:
            out["extraction"] = extraction
        default_workload = shape_key == (64, 4096, 19, 'baseline', 640, 480)
        if world == 1 and default_workload and not args.no_extra_configs and not dist.is_initialized():
            out["extra_configs"] = extra_configs()
        if (not args.no_cpu_baseline or args.with_cpu_baseline) and world == 1:          # the CPU leg is timed on rank 0 at N = 1 only
            out["cpu_baseline"] = cpu_baseline(args.cpu_seconds, R, args.log2_T, args.mlp, args.finest)
        print(json.dumps(out), flush=True)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
