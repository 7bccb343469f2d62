"""N > 1 path on CPU: two gloo ranks (one process each, rendezvous on 127.0.0.1) exercise the data-parallel plumbing of
bundlesdf_amd/dist.py with the CPU oracle standing in for the device step:
  * keyframe sharding + all-gather give every rank the identical pose table / octree cloud;
  * summed gradients pre-scaled by 1/world equal ONE process stepping on the concatenated batch."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import nof_oracle as O


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _scene(R):
    from tests.test_gpu_ops import _scene as sc
    cfg, occ, c2w, batch = sc(None, R=R, level=4, seed=1)
    cfg.update(N_samples=16, N_samples_around_depth=8, num_levels=4, log2_hashmap_size=10, finest_res=64)
    return cfg, occ, c2w, batch


def _field(cfg, occ, c2w, scale):
    torch.manual_seed(0)
    geo = O.HashGeometry(cfg['num_levels'], 2, cfg['base_res'], cfg['log2_hashmap_size'], cfg['finest_res'])
    shape = O.FieldShape(input_ch=geo.out_dim)
    tab = (torch.rand(geo.n_entries, 2) * 2 - 1) * 0.05
    return O.OracleField(cfg, geo, shape, c2w.shape[0], c2w, occ, table=tab)


def _worker(rank, world, port, R, out):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from bundlesdf_amd import dist as D
    r, w, _ = D.init_from_env(torch.device('cpu'))
    assert (r, w) == (rank, world)
    cfg, occ, c2w, batch = _scene(R * world)
    # --- sharding + all-gather of per-rank metadata ---
    lo, hi = D.shard_frames(c2w.shape[0] + 1, rank, world)
    assert (lo, hi) == ((0, 3) if rank == 0 else (3, 6))
    poses_all = D.all_gather_cat(torch.from_numpy(c2w[:4] + rank))
    assert poses_all.shape[0] == 8 and torch.equal(poses_all[4:], torch.from_numpy(c2w[:4] + 1))
    # --- gradient averaging == single process on the concatenated batch ---
    fld = _field(cfg, occ, c2w, 1.0 / world)
    rng = np.random.default_rng(7)
    u1 = rng.random((R * world, cfg['N_samples'])).astype(np.float32)
    u2 = rng.random((R * world, cfg['N_samples_around_depth'])).astype(np.float32)
    sl = slice(rank * R, (rank + 1) * R)
    res = fld.train_step(batch[sl], u1[sl], u2[sl], do_step=False)
    flat = torch.cat([g.reshape(-1) for g in res['grads']]) / world          # NofLossCfg.grad_scale = 1/world
    sync = D.make_grad_sync()
    assert sync is not None
    bucketed = flat.clone()
    bucketed_local = flat.clone()                        # this rank's own (pre-scaled) gradient, for the sharded-optimiser check below
    sync(flat)                                           # one blocking all-reduce of the whole buffer
    # the bucketed asynchronous form the device step uses: TWO collectives -- the middle slice (fine levels + MLP) first, then
    # [copy of the tail | head slice] as one contiguous range (the tail rides in headroom in front of the buffer): same bits
    a, b = flat.numel() // 3, 2 * flat.numel() // 3
    nt = flat.numel() - b
    store = torch.zeros(nt + flat.numel())
    store[nt:] = bucketed
    view = store[nt:]
    sync.start(view[a:b])
    store[:nt] = view[b:]
    sync.start(store[:nt + a])
    sync.start(view[:0])                                 # empty slices are skipped
    # the step runs the first slice's share of Adam between the two waits: that slice is final after finish_first()
    assert len(sync.pending) == 2
    sync.finish_first()
    assert len(sync.pending) == 1 and torch.equal(view[a:b], flat[a:b])
    sync.finish()
    view[b:] = store[:nt]
    assert not sync.pending and torch.equal(view, flat)
    assert sync.collectives_step == 2 and sync.bytes_step == 4 * (b - a + nt + a) and sync.timed_steps == 1
    # bf16 payload: the slice is rounded to bfloat16, summed in bfloat16 and written back -- the same bits on every rank
    n = 4099
    mine = [((torch.arange(n) % 97).float() - 48.0) * (1e-3 * (r + 1)) + 1e-5 * (r + 1) for r in range(world)]
    buf = mine[rank].clone()
    sb = D.GradSync(payload='bf16')
    sb.start(buf[3:], compressed=True)
    sb.start(buf[:3])                                    # (uncompressed slices of the same step stay fp32)
    assert len(sb.pending) == 2
    sb.finish_first()
    want = mine[0][3:].bfloat16()
    for r in range(1, world):
        want = want + mine[r][3:].bfloat16()
    assert torch.equal(buf[3:], want.float())
    sb.finish()
    assert torch.equal(buf[:3], sum(m[:3] for m in mine)) and sb.bytes_step == 2 * (n - 3) + 4 * 3 and sb.collectives_step == 2
    assert (buf[3:] - sum(m[3:] for m in mine)).abs().max() <= 2.0 ** -7 * sum(m[3:] for m in mine).abs().max()
    plain = mine[rank].clone()
    sp = D.GradSync()                                    # payload fp32: compressed=True changes nothing
    sp.start(plain[3:], compressed=True)
    sp.finish()
    assert torch.equal(plain[3:], sum(m[3:] for m in mine)) and sp.bytes_step == 4 * (n - 3)
    # sharded optimiser (GradSync mode 'zero1'): reduce-scatter -> Adam on this rank's 1/world of the flat buffers ->
    # all-gather of the parameters == all-reduce -> Adam on everything, BIT for bit, on every rank
    n = flat.numel()
    n_pad, shard, lo, hi = D.GradSync.shard_range(n)
    assert n_pad % (4 * world) == 0 and shard * world == n_pad and 0 <= n_pad - n < 4 * world
    assert D.GradSync.shard_range(10, 1, 2) == (16, 8, 8, 10) and D.GradSync.shard_range(10, 1, 4)[2:] == (4, 8)
    assert D.GradSync.shard_range(3, 3, 4)[2:] == (3, 3)             # a rank can own nothing but padding
    p0 = torch.linspace(-1, 1, n)
    pa, ma, va = O.adam_reference_step(p0.numpy(), flat.numpy(), np.zeros(n, np.float32), np.zeros(n, np.float32), 1, np.float32(0.01))
    gz, pz = torch.zeros(n_pad), torch.zeros(n_pad)
    gz[:n], pz[:n] = bucketed_local, p0
    sz = D.make_grad_sync(mode='zero1')
    assert sz.mode == 'zero1'
    sz.reduce_scatter_(gz)
    assert torch.equal(gz[lo:hi], flat[lo:hi])                      # my shard holds the sum the all-reduce gave
    ps, ms, vs = O.adam_reference_step(p0[lo:hi].numpy(), gz[lo:hi].numpy(), np.zeros(hi - lo, np.float32), np.zeros(hi - lo, np.float32),
                                       1, np.float32(0.01))
    pz[lo:hi] = torch.from_numpy(ps)
    pz[:lo] = float('nan')                                          # what other ranks own is theirs to deliver
    pz[hi:n] = float('nan')
    sz.all_gather_(pz)
    sz.end_step()
    assert torch.equal(pz[:n], torch.from_numpy(pa))
    assert sz.collectives_step == 2 and sz.bytes_step == 2 * 4 * n_pad and sz.timed_steps == 1
    # touched-row exchange (GradSync mode 'rows'): a table gradient that is zero except in the rows a rank's samples hit -- bitmap
    # all-gather, OR, all-reduce of the union's rows -- equals the dense all-reduce BIT for bit, and moves the union's rows only
    nrow = 1003                                          # (not a multiple of 8: the bitmap's last byte is partial)
    gens = [torch.Generator().manual_seed(100 + r) for r in range(world)]
    tabs = []
    for r in range(world):
        t = torch.zeros(nrow, 2)
        hit = torch.randperm(nrow, generator=gens[r])[:nrow // 10]
        t[hit] = torch.randn(hit.numel(), 2, generator=gens[r])
        t[hit[:5], 1] = 0.0                              # rows with one zero entry still count as touched
        t[hit[5:8]] = 0.0                                # ... and a hit that summed to exactly zero is simply not sent
        tabs.append(t)
    tail = [torch.randn(37, generator=gens[r]) for r in range(world)]
    mine = torch.cat([tabs[rank].reshape(-1), tail[rank]])
    dense = mine.clone()
    dist.all_reduce(dense)
    sr = D.make_grad_sync(mode='rows')
    assert sr.mode == 'rows'
    k = sr.exchange_rows_(mine[:2 * nrow], 2)
    sr.exchange_dense_(mine[2 * nrow:])
    sr.end_step()
    union = torch.zeros(nrow, dtype=torch.bool)
    for t in tabs:
        union |= (t != 0).any(1)
    assert k == int(union.sum()) == sr.rows_step and 0 < k < nrow // 4
    assert torch.equal(mine, dense)
    assert sr.collectives_step == 3 and sr.bytes_step == (nrow + 7) // 8 + 8 * k + 4 * 37
    empty = torch.zeros(2 * 64)                          # nothing touched anywhere: the bitmap travels, no row does
    assert sr.exchange_rows_(empty, 2) == 0 and not empty.any()
    sr.end_step()
    assert sr.collectives_step == 1
    # only the skip bit (4) travels; a rank's other bits -- 1: a ray exceeded max_hits, 8: the sticky mark of an EARLIER skipped step
    # -- neither hide another rank's skip bit (MAX over the whole word would let 9 beat 4) nor spread
    fl = torch.tensor([4 if rank == 1 else 9, 7, 0, 0], dtype=torch.int32)
    sz.max_flags_(fl)
    assert int(fl[0]) == (4 if rank == 1 else 13) and int(fl[1]) == 7  # one rank's overflow is everybody's skipped step
    if rank == 0:
        out.put(flat.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_gradient_sum_equals_single_process_batch():
    R, world = 24, 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, R, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    cfg, occ, c2w, batch = _scene(R * world)
    fld = _field(cfg, occ, c2w, 1.0)
    rng = np.random.default_rng(7)
    u1 = rng.random((R * world, cfg['N_samples'])).astype(np.float32)
    u2 = rng.random((R * world, cfg['N_samples_around_depth'])).astype(np.float32)
    ref = fld.train_step(batch, u1, u2, do_step=False)
    want = torch.cat([g.reshape(-1) for g in ref['grads']]).numpy()
    assert np.abs(got - want).max() < 1e-5 * max(1.0, np.abs(want).max())


def test_single_process_helpers_are_identity():
    from bundlesdf_amd import dist as D
    t = torch.arange(6.0).reshape(2, 3)
    assert torch.equal(D.all_gather_cat(t), t)
    assert D.make_grad_sync() is None
    assert D.shard_frames(512, 3, 8) == (192, 256)


def test_bench_refuses_to_run_fewer_ranks_than_asked():
    """`python bench.py --gpus N` without a launcher starts its N ranks itself -- and exits loudly, before any measurement, when
    fewer than N GPUs are visible (here: none), instead of silently benchmarking a smaller world and reporting n_gpus = N; under a
    launcher a WORLD_SIZE that disagrees with --gpus is an error too."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        pytest.skip('two GPUs are visible')
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT')}
    p = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0'],
                       capture_output=True, text=True, env=env, timeout=300)
    assert p.returncode != 0 and 'GPU(s) are visible' in p.stderr and '--gpus 2' in p.stderr, p.stderr[-500:]
    assert p.stdout.strip() == ''                                     # no result line
    if not torch.cuda.is_available():
        return
    env2 = dict(env, WORLD_SIZE='1', RANK='0', LOCAL_RANK='0')
    p = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0'],
                       capture_output=True, text=True, env=env2, timeout=300)
    assert p.returncode != 0 and 'WORLD_SIZE=1' in p.stderr


def _rows_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from bundlesdf_amd import dist as D
    D.init_from_env(torch.device('cpu'))
    nrow = 4099
    g = torch.Generator().manual_seed(7 + rank)
    t = torch.zeros(nrow, 2)
    hit = torch.randperm(nrow, generator=g)[:nrow // 8]
    t[hit] = torch.randn(hit.numel(), 2, generator=g)
    mine = t.reshape(-1).clone()
    dense = mine.clone()
    dist.all_reduce(dense)
    sr = D.make_grad_sync(mode='rows')
    k = sr.exchange_rows_(mine, 2)
    sr.end_step()
    masks = [torch.zeros(nrow, dtype=torch.bool) for _ in range(world)]
    masks[rank] = (t != 0).any(1)
    allm = torch.stack(masks).to(torch.uint8)
    dist.all_reduce(allm)
    assert k == int((allm.sum(0) > 0).sum())
    # more than two ranks: the union's rows are summed by an all-reduce over another buffer, so the order of a row's terms may differ
    # from the dense call's -- equal to rounding, identical on every rank, and exactly zero wherever no rank touched a row
    assert torch.allclose(mine, dense, rtol=1e-6, atol=1e-7)
    assert not mine.view(-1, 2)[allm.sum(0) == 0].any()
    every = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(every, mine)
    assert all(torch.equal(e, every[0]) for e in every)
    if rank == 0:
        out.put((k, sr.bytes_step))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_touched_row_exchange_three_ranks():
    """GradSync mode 'rows' at a world size that is neither 2 nor a power of two: the union of three ranks' rows travels, every
    rank ends with the same bits, untouched rows stay exactly zero, and the sum equals the dense all-reduce's to rounding."""
    world = 3
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rows_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    k, nbytes = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert 4099 // 8 < k < 3 * (4099 // 8) + 1 and nbytes == (4099 + 7) // 8 + 8 * k
