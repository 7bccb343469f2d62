"""Operator-level parity: every C-ABI entry point of libnof_hip.so against the CPU oracle (oracle/nof_oracle.py)
on seeded inputs.  Bit-exact for index / interval / z work, stated tolerances for floating point."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import nof_oracle as O
from tests import util as U

pytestmark = pytest.mark.gpu


def cpu(t):
    return t.detach().cpu().numpy()


# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("precision,K,tol", [(0, 32, 1e-5), (1, 64, 2e-2), (2, 64, 2e-3)])
def test_mfma_operand_layout(nof, precision, K, tol):
    """A[32,K] @ B[K,32] through the exact operand / accumulator layouts nof_mlp uses (asymmetric inputs)."""
    rng = np.random.default_rng(precision)
    A = rng.normal(size=(32, K)).astype(np.float32)
    Bm = rng.normal(size=(K, 32)).astype(np.float32)
    D = torch.zeros(32, 32, device='cuda')
    from bundlesdf_amd import build as Bld
    probe = Bld.load_probe()                  # test-only library (csrc/test/nof_probe.hip over nof_mfma_dev.h): not in libnof_hip.so
    dA, dB = U.dev(A), U.dev(Bm)
    rc = probe.nof_mfma_probe(precision, dA.data_ptr(), dB.data_ptr(), D.data_ptr(), K, torch.cuda.current_stream().cuda_stream)
    assert rc == 0, probe.nof_probe_last_error()
    torch.cuda.synchronize()
    ref = A.astype(np.float64) @ Bm.astype(np.float64)
    err = np.abs(cpu(D) - ref).max() / np.abs(ref).max()
    assert err < tol, err


# ------------------------------------------------------------------------------------------------
def test_scatter_survives_a_different_register_allocation(nof):
    """libnof_hash_perturb.so = nof_hash.hip built with -DNOF_AGG_PERTURB: two dozen extra values live across the scatter's
    hand-written DPP scan block, so every register the block touches sits somewhere else.  The table gradient of ray-coherent
    samples (long runs, chains across cells: the paths the block serves) must equal the regular build's up to the atomics' order."""
    import os
    from bundlesdf_amd import build as B
    path = B.build_perturb(verbose=False)          # no-op when the library's stamp matches its sources' content, else a rebuild
    assert os.path.exists(path), f'{path} is missing: __graft_entry__.build() builds it'
    alt = C.CDLL(path)
    fn = alt.nof_hash_encode_bwd
    fn.restype = C.c_int
    fn.argtypes = [C.POINTER(nof.NofHashGrid)] + [C.c_void_p] * 5 + [C.c_int64, C.c_void_p]
    g, geo = U.make_grids(nof, L=16, T=19, finest=256)
    R, S = 256, 192
    Bn = R * S
    gen = torch.Generator(device='cuda').manual_seed(2)
    o = torch.randn(R, 3, device='cuda', generator=gen)
    o = o / o.norm(dim=1, keepdim=True) * 1.6
    d = (torch.rand(R, 3, device='cuda', generator=gen) - 0.5) * 1.2 - o
    d = d / d.norm(dim=1, keepdim=True)
    t = torch.linspace(0.55, 2.4, S, device='cuda')[None, :, None] + torch.rand(R, S, 1, device='cuda', generator=gen) * 0.004
    pts = (o[:, None, :] + t * d[:, None, :]).reshape(Bn, 3).contiguous()
    table = (torch.rand(geo.n_entries, 2, device='cuda', generator=gen) - 0.5) * 0.2
    dfeat = torch.randn(16, Bn, 2, device='cuda', generator=gen)
    want = torch.zeros(geo.n_entries, 2, device='cuda')
    nof.call('nof_hash_encode_bwd', C.byref(g), pts, table, dfeat, want, None, Bn)
    got = torch.zeros(geo.n_entries, 2, device='cuda')
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    rc = fn(C.byref(g), C.c_void_p(pts.data_ptr()), C.c_void_p(table.data_ptr()), C.c_void_p(dfeat.data_ptr()),
            C.c_void_p(got.data_ptr()), None, Bn, st)
    torch.cuda.synchronize()
    assert rc == 0
    assert (got - want).abs().max().item() <= 2e-5 * want.abs().max().item()
    off = geo.offsets
    for l in range(16):                                          # level by level: a coarse level cannot hide a fine one
        a, b = got[off[l]:off[l + 1]], want[off[l]:off[l + 1]]
        assert (a - b).norm().item() <= 1e-5 * b.norm().item(), l


@pytest.mark.parametrize("L,T,finest", [(16, 14, 256), (4, 22, 128), (16, 19, 512)])
def test_hash_forward_backward(nof, L, T, finest):
    g, geo = U.make_grids(nof, L=L, T=T, finest=finest)
    B = 3000
    pts = U.test_points(B, seed=L)
    torch.manual_seed(0)
    table = (torch.rand(geo.n_entries, 2) * 2 - 1) * 0.1
    d_pts, d_table = U.dev(pts), table.cuda()
    feat = torch.empty(L, B, 2, device='cuda')
    nof.call('nof_hash_encode_fwd', C.byref(g), d_pts, d_table, feat, B)
    idx = torch.empty(B, L, 8, dtype=torch.int32, device='cuda')
    nof.call('nof_hash_corner_indices', C.byref(g), d_pts, idx, B)
    torch.cuda.synchronize()

    # integer part: bit-exact table rows
    x01 = ((pts + np.float32(1.0)) * np.float32(0.5)).astype(np.float32)
    oob = ((x01 < 0) | (x01 > 1)).any(-1)
    ref_idx = O.hash_corner_indices(x01, geo)
    got_idx = cpu(idx).astype(np.int64)
    assert (got_idx[oob] == -1).all()
    assert np.array_equal(got_idx[~oob], ref_idx[~oob])

    # floating point part
    tp = torch.from_numpy(pts).requires_grad_(True)
    tt = table.clone().requires_grad_(True)
    ref = O.hash_encode((tp + 1) / 2, tt, geo)
    got = cpu(feat).transpose(1, 0, 2).reshape(B, L * 2)
    assert np.abs(got - ref.detach().numpy()).max() < 2e-6
    assert (got[oob] == 0).all()

    # backward: table grads (atomic scatter) and input grads
    torch.manual_seed(1)
    dy = torch.randn(B, L * 2)
    ref.backward(dy)
    dfeat = dy.reshape(B, L, 2).permute(1, 0, 2).contiguous().cuda()
    gtab = torch.zeros(geo.n_entries, 2, device='cuda')
    dpts = torch.full((B, 3), 7.0, device='cuda')
    nof.call('nof_hash_encode_bwd', C.byref(g), d_pts, d_table, dfeat, gtab, dpts, B)
    torch.cuda.synchronize()
    gt_ref = tt.grad.numpy()
    assert np.abs(cpu(gtab) - gt_ref).max() < 1e-4 * max(1.0, np.abs(gt_ref).max())
    dp_ref = tp.grad.numpy()
    assert np.abs(cpu(dpts) - dp_ref).max() < 2e-4 * max(1.0, np.abs(dp_ref).max())


# ------------------------------------------------------------------------------------------------
def test_pose_forward_backward(nof):
    F = 9
    rng = np.random.default_rng(3)
    pose = rng.normal(size=(F, 6)).astype(np.float32) * 0.8
    pose[1] = 0.0                       # |w|^2 below the 1e-4 clamp
    pose[2, 3:] = 1e-3
    c2w = np.tile(np.eye(4, dtype=np.float32), (F, 1, 1))
    for f in range(F):
        q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
        c2w[f, :3, :3] = q
        c2w[f, :3, 3] = rng.normal(size=3)
    mt, mr = 0.105, 20.0
    tf = torch.empty(F, 12, device='cuda')
    nof.call('nof_pose_fwd', U.dev(pose), U.dev(c2w.reshape(F, 16)), C.c_float(mt), C.c_float(mr / 180 * np.pi), tf, F)
    tp = torch.from_numpy(pose).requires_grad_(True)
    Ts = O.pose_matrices(tp, mt, mr)
    ref = (Ts @ torch.from_numpy(c2w))[:, :3, :4]
    torch.cuda.synchronize()
    assert np.abs(cpu(tf).reshape(F, 3, 4) - ref.detach().numpy()).max() < 2e-6
    G = torch.from_numpy(rng.normal(size=(F, 3, 4)).astype(np.float32))
    (Ts[:, :3, :4] * G).sum().backward()
    gp = torch.zeros(F, 6, device='cuda')
    nof.call('nof_pose_bwd', U.dev(pose), G.reshape(F, 12).cuda().contiguous(), C.c_float(mt), C.c_float(mr / 180 * np.pi), gp, F)
    torch.cuda.synchronize()
    ref_g = tp.grad.numpy()
    assert np.abs(cpu(gp) - ref_g).max() < 2e-4 * max(1.0, np.abs(ref_g).max())
    assert (cpu(gp)[0] == 0).all()


# ------------------------------------------------------------------------------------------------
def _build_occ(nof, occ, level):
    n = occ.shape[0]
    bits = torch.zeros((n ** 3 + 31) // 32, dtype=torch.int32, device='cuda')
    coords = U.dev(U.occ_to_coords(occ))
    nof.call('nof_occgrid_build', coords, coords.shape[0], level, level, bits)
    return bits


@pytest.mark.parametrize("level,fill", [(4, 0.15), (5, 0.05), (3, 0.5), (6, 0.02)])
def test_trace_bit_identical(nof, level, fill):
    n = 1 << level
    occ = U.random_occ(n, fill, seed=level)
    bits = _build_occ(nof, occ, level)
    R = 600
    o, d = U.random_rays(R, seed=level)
    # axis aligned / diagonal / grazing / inside-origin rays (SURVEY 8c golden list)
    o[0], d[0] = [-3, 0.01, 0.02], [1, 0, 0]
    o[1], d[1] = [0.3, 3, -0.2], [0, -1, 0]
    o[2], d[2] = [-2, -2, -2], np.array([1, 1, 1]) / np.sqrt(3)
    o[3], d[3] = [-3, 1.0, 0.0], [1, 0, 0]                      # grazing the +y face
    o[4], d[4] = [0.01, 0.02, 0.03], [0, 0, 1]                  # origin inside the cube
    o[5], d[5] = [-3, 0.5, 0.5], [1, 0, 0]                      # along a cell boundary plane
    o[6], d[6] = [3, 3, 3], [1, 0, 0]                           # misses everything
    o, d = o.astype(np.float32), d.astype(np.float32)
    H = 3 * n + 2
    tio = torch.empty(R, H, 2, device='cuda')
    cid = torch.empty(R, H, dtype=torch.int32, device='cuda')
    nh = torch.empty(R, dtype=torch.int32, device='cuda')
    flags = torch.zeros(4, dtype=torch.int32, device='cuda')
    nof.call('nof_trace_rays', bits, level, U.dev(o), U.dev(d), R, H, tio, cid, nh, flags)
    torch.cuda.synchronize()
    r_tio, r_cid, r_nh = O.trace_rays(occ, o, d, max_hits=H)
    assert cpu(flags)[0] == 0
    assert np.array_equal(cpu(nh), r_nh)
    assert np.array_equal(cpu(cid), r_cid)                       # bit-identical ray-hit indices
    assert np.array_equal(cpu(tio).view(np.uint32), r_tio.view(np.uint32))   # bit-identical intervals
    # occupancy query
    q = np.random.default_rng(0).uniform(-1.1, 1.1, size=(5000, 3)).astype(np.float32)
    inside = torch.empty(5000, dtype=torch.uint8, device='cuda')
    nof.call('nof_occgrid_query', bits, level, U.dev(q), inside, 5000)
    torch.cuda.synchronize()
    qi = np.floor(np.clip(n * (q + 1.0) / 2.0, 0, n - 1.0)).astype(np.int64)
    assert np.array_equal(cpu(inside).astype(bool), occ[qi[:, 0], qi[:, 1], qi[:, 2]])


def _scene(nof, R=300, level=4, seed=0, n_frames=5):
    """Random occupancy + camera-like rays with depths; returns everything the sampler needs."""
    rng = np.random.default_rng(seed)
    n = 1 << level
    occ = np.zeros((n, n, n), bool)
    c = (np.indices((n, n, n)).transpose(1, 2, 3, 0) + 0.5) / n * 2 - 1
    occ[np.linalg.norm(c, axis=-1) < 0.55] = True
    cfg = O.default_cfg(sc_factor=5.0, N_samples=64, N_samples_around_depth=32, far=1.0, num_levels=16,
                        log2_hashmap_size=14, finest_res=256)
    F = n_frames
    c2w = np.tile(np.eye(4, dtype=np.float32), (F, 1, 1))
    for f in range(F):
        pos = rng.normal(size=3)
        pos = pos / np.linalg.norm(pos) * 3.0
        zax = pos / np.linalg.norm(pos)                              # GL camera looks along -z
        xax = np.cross([0, 0, 1.0], zax)
        xax /= np.linalg.norm(xax)
        yax = np.cross(zax, xax)
        c2w[f, :3, 0], c2w[f, :3, 1], c2w[f, :3, 2], c2w[f, :3, 3] = xax, yax, zax, pos
    batch = np.zeros((R, 12), np.float32)
    batch[:, 0] = rng.uniform(-0.2, 0.2, R)
    batch[:, 1] = rng.uniform(-0.2, 0.2, R)
    batch[:, 2] = -1.0
    batch[:, 3:6] = rng.uniform(0, 1, (R, 3))
    batch[:, 6] = rng.uniform(2.4, 3.2, R)                          # depth (normalised units)
    batch[:, 7] = 1
    batch[:, 8] = rng.integers(0, F, R)
    batch[::7, 6] = 99 * cfg['sc_factor']                           # BAD_DEPTH rays (type stays 0: uncertain free space)
    batch[5::11, 9] = 1
    batch[:, 10], batch[:, 11] = 1.0, 5.0
    return cfg, occ, c2w, batch


@pytest.mark.parametrize("level,fill", [(2, 0.6), (4, 0.2), (5, 0.08), (6, 0.03)])
def test_wave_ray_marcher_equals_the_walk(nof, level, fill):
    """marcher = NOF_MARCHER_WAVE: one wave per ray, the ray's cells from the ranks of its plane crossings instead of a walk
    (k_batch_trace_wave; NumPy statement of the algorithm: tools/dda_closed_form.py, checked against the walk on the CPU).  Every
    output of nof_batch_trace -- gathered rows, ray setup, view rows, intervals, cell ids, hit counts, the overflow flag -- must be
    the walk kernel's, bit for bit: random rays through random poses, rays built to tie (lattice origins, axis-parallel and diagonal
    directions, identity pose), frame features in the view rows, a hit-list capacity small enough to overflow, no cell ids."""
    n = 1 << level
    rng = np.random.default_rng(70 + level)
    occ = U.random_occ(n, fill, seed=80 + level)
    bits = _build_occ(nof, occ, level)
    F, ff = 24, 2
    cs = 2.0 / n
    # frames 0..11: random rigid poses around the grid; 12..23: identity rotation, the translation a grid vertex moved a whole number
    # of cells back along the ray direction of that frame's rays
    tf = np.zeros((F, 12), np.float32)
    dirs = [(1, 0, 0), (0, -1, 0), (0, 0, 1), (1, 1, 0), (1, -1, 0), (0, 1, 1), (1, 1, 1), (-1, 1, -1), (1, 2, 0), (2, 1, 3), (1, 2, 4), (4, 4, 1)]
    for f in range(12):
        q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
        tf[f].reshape(3, 4)[:, :3] = q
        tf[f].reshape(3, 4)[:, 3] = rng.normal(size=3) * 1.2
    for f in range(12, 24):
        dv = np.array(dirs[f - 12], np.float64)
        tf[f].reshape(3, 4)[:, :3] = np.eye(3)
        tf[f].reshape(3, 4)[:, 3] = rng.integers(0, n + 1, size=3) * cs - 1.0 - dv / np.abs(dv).max() * rng.integers(1, 2 * n) * cs
    R = 3000
    batch = rng.normal(size=(R, 12)).astype(np.float32)
    fid = rng.integers(0, F, size=R)
    batch[:, 8] = fid
    for r in range(R):
        if fid[r] >= 12:
            batch[r, :3] = np.array(dirs[fid[r] - 12], np.float32) * np.float32(rng.uniform(0.5, 2.0))
        else:
            tgt = rng.uniform(-0.9, 0.9, size=3)                    # towards the grid, in the frame's camera coordinates
            batch[r, :3] = tf[fid[r]].reshape(3, 4)[:, :3].T @ (tgt - tf[fid[r]].reshape(3, 4)[:, 3])
    feat = rng.normal(size=(F, ff)).astype(np.float32)
    ids = torch.from_numpy(rng.permutation(R)).cuda()
    outs = {}
    for kind in (0, 1):                                            # (0 = the walk, 1 = the wave kernel: the keys below)
        for H, want_cells in ((3 * n + 2, True), (5, False)):
            o = dict(batch=torch.full((R, 12), 7.0, device='cuda'), o_w=torch.empty(R, 3, device='cuda'), d_w=torch.empty(R, 3, device='cuda'),
                     view=torch.empty(R, 16, device='cuda'), tio=torch.full((R, H, 2), 7.0, device='cuda'),
                     cid=torch.full((R, H), 7, dtype=torch.int32, device='cuda') if want_cells else None,
                     nh=torch.empty(R, dtype=torch.int32, device='cuda'), flags=torch.zeros(4, dtype=torch.int32, device='cuda'))
            nof.call('nof_batch_trace', U.dev(batch), ids, U.dev(tf), U.dev(feat), ff, 3, bits, level, R, H,
                     nof.MARCHER_WAVE if kind == 1 else nof.MARCHER_WALK, o['batch'], o['o_w'], o['d_w'],
                     o['view'], o['tio'], o['cid'], o['nh'], o['flags'])
            torch.cuda.synchronize()
            outs[(kind, H)] = o
    for H in (3 * n + 2, 5):
        a, b = outs[(0, H)], outs[(1, H)]
        for k in ('batch', 'o_w', 'd_w', 'view', 'tio', 'nh', 'flags'):
            assert torch.equal(a[k].view(torch.int32) if a[k].dtype == torch.float32 else a[k],
                               b[k].view(torch.int32) if b[k].dtype == torch.float32 else b[k]), (H, k)
        if a['cid'] is not None:
            assert torch.equal(a['cid'], b['cid'])
    full = outs[(1, 3 * n + 2)]
    assert int(full['flags'][0]) == 0 and int(outs[(1, 5)]['flags'][0]) == (1 if int(full['nh'].max()) > 5 else 0)
    assert int(full['nh'].sum()) > R // 2, 'the rays should hit something'
    tied = torch.from_numpy(fid >= 12).cuda()[ids]
    assert int(full['nh'][tied].sum()) > 0


@pytest.mark.parametrize("level", [4, 6])
def test_fused_raymarch_sample_equals_the_two_launches(nof, level):
    """nof_raymarch_sample with the wave-per-ray kernel is ONE launch (k_raymarch_wave<true>: the wave that enumerated a ray's cells
    places its samples); with NofSampleCfg.marcher = NOF_MARCHER_WALK it is the walk kernel followed by k_sample_points.  Same bits in every output
    -- intervals, hit counts, z, sample points, validity, flags (incl. the skipped-step mark 4 -> 8 the sampler half turns over) --
    with injected uniforms, with Philox, and with perturb=False; rays with usable and unusable depth."""
    cfg, occ, c2w, batch = _scene(nof, level=level, R=2000, seed=3)
    R, F = batch.shape[0], c2w.shape[0]
    bits = _build_occ(nof, occ, level)
    rng = np.random.default_rng(2)
    pose = (rng.normal(size=(F, 6)) * 0.3).astype(np.float32)
    tf = torch.empty(F, 12, device='cuda')
    nof.call('nof_pose_fwd', U.dev(pose), U.dev(c2w.reshape(F, 16)), C.c_float(cfg['max_trans'] * cfg['sc_factor']),
             C.c_float(cfg['max_rot'] / 180 * np.pi), tf, F)
    Ns, Na = cfg['N_samples'], cfg['N_samples_around_depth']
    S, H = Ns + Na, 3 * (1 << level) + 2
    u_occ, u_dep = U.dev(rng.random((R, Ns)).astype(np.float32)), U.dev(rng.random((R, Na)).astype(np.float32))
    ids = torch.from_numpy(rng.permutation(R)).cuda()
    trunc = O.get_truncation(cfg, 0)
    for uo, ud, det in ((u_occ, u_dep, 0), (None, None, 0), (None, None, 1)):
        res = {}
        for kind in (0, 1):
            sc = nof.NofSampleCfg(Ns, Na, cfg['near'] * cfg['sc_factor'], cfg['far'] * cfg['sc_factor'], trunc, cfg['neg_trunc_ratio'], 77, 5,
                                  None, det, nof.MARCHER_WAVE if kind == 1 else nof.MARCHER_WALK)
            o = dict(batch=torch.empty(R, 12, device='cuda'), o_w=torch.empty(R, 3, device='cuda'), d_w=torch.empty(R, 3, device='cuda'),
                     view=torch.empty(R, 16, device='cuda'), tio=torch.empty(R, H, 2, device='cuda'),
                     cid=torch.empty(R, H, dtype=torch.int32, device='cuda'), nh=torch.empty(R, dtype=torch.int32, device='cuda'),
                     z=torch.full((R, S), 7.0, device='cuda'), pts=torch.full((R * S, 3), 7.0, device='cuda'),
                     valid=torch.full((R * S,), 7, dtype=torch.uint8, device='cuda'),
                     flags=torch.tensor([4, 0, 0, 0], dtype=torch.int32, device='cuda'))
            nof.call('nof_raymarch_sample', C.byref(sc), U.dev(batch), ids, tf, None, 0, 3, bits, level, R, H, uo, ud, o['batch'], o['o_w'],
                     o['d_w'], o['view'], o['tio'], o['cid'], o['nh'], o['z'], o['pts'], o['valid'], o['flags'])
            torch.cuda.synchronize()
            res[kind] = o
        a, b = res[0], res[1]
        for k in a:
            x, y = (a[k].view(torch.int32), b[k].view(torch.int32)) if a[k].dtype == torch.float32 else (a[k], b[k])
            assert torch.equal(x, y), (det, uo is None, k)
        assert int(b['flags'][0]) == 8 and int(b['nh'].max()) > 0 and bool((b['valid'] <= 1).all())
        assert float(b['z'].max()) > 0 and bool(torch.isfinite(b['pts']).all())


def test_sample_points_bit_identical(nof):
    level = 4
    cfg, occ, c2w, batch = _scene(nof, level=level)
    R, F = batch.shape[0], c2w.shape[0]
    bits = _build_occ(nof, occ, level)
    rng = np.random.default_rng(1)
    pose = (rng.normal(size=(F, 6)) * 0.3).astype(np.float32)
    mt, mr = cfg['max_trans'] * cfg['sc_factor'], cfg['max_rot']
    tf = torch.empty(F, 12, device='cuda')
    nof.call('nof_pose_fwd', U.dev(pose), U.dev(c2w.reshape(F, 16)), C.c_float(mt), C.c_float(mr / 180 * np.pi), tf, F)
    H = 3 * (1 << level) + 2
    d_batch = torch.empty(R, 12, device='cuda')
    o_w = torch.empty(R, 3, device='cuda')
    d_w = torch.empty(R, 3, device='cuda')
    view = torch.empty(R, 16, device='cuda')
    tio = torch.empty(R, H, 2, device='cuda')
    cid = torch.empty(R, H, dtype=torch.int32, device='cuda')
    nh = torch.empty(R, dtype=torch.int32, device='cuda')
    flags = torch.zeros(4, dtype=torch.int32, device='cuda')
    ids = torch.arange(R, device='cuda').flip(0).contiguous()
    nof.call('nof_batch_trace', U.dev(batch), ids, tf, None, 0, 3, bits, level, R, H, nof.MARCHER_WAVE, d_batch, o_w, d_w, view, tio, cid,
             nh, flags)
    torch.cuda.synchronize()
    bb = batch[::-1].copy()
    assert np.array_equal(cpu(d_batch), bb)
    # ray setup vs oracle (fp tolerance), trace vs oracle on the SAME o,d (bit-exact)
    field = O.OracleField(cfg, O.HashGeometry(16, 2, 16, 14, 256), O.FieldShape(), F, c2w, occ, pose=pose)
    Ns, Na = cfg['N_samples'], cfg['N_samples_around_depth']
    u_occ = rng.random((R, Ns)).astype(np.float32)
    u_dep = rng.random((R, Na)).astype(np.float32)
    z_ref, tr = field.trace_and_sample(torch.from_numpy(bb), u_occ, u_dep)
    assert np.abs(cpu(o_w) - tr['rays_o_w'].numpy()).max() < 1e-5
    assert np.abs(cpu(d_w) - tr['viewdirs_w'].numpy()).max() < 1e-6
    sh_ref = O.sh_encode(tr['viewdirs_w'], 3).numpy()
    assert np.abs(cpu(view)[:, :9] - sh_ref).max() < 1e-5 and (cpu(view)[:, 9:] == 0).all()
    r_tio, r_cid, r_nh = O.trace_rays(occ, cpu(o_w), cpu(d_w), max_hits=H)
    assert np.array_equal(cpu(nh), r_nh) and np.array_equal(cpu(cid), r_cid)
    assert np.array_equal(cpu(tio).view(np.uint32), r_tio.view(np.uint32))

    trunc = O.get_truncation(cfg, 0)
    sc = nof.NofSampleCfg(Ns, Na, cfg['near'] * cfg['sc_factor'], cfg['far'] * cfg['sc_factor'], trunc,
                          cfg['neg_trunc_ratio'], 1234, 0)
    S = Ns + Na
    z = torch.empty(R, S, device='cuda')
    pts = torch.empty(R * S, 3, device='cuda')
    valid = torch.empty(R * S, dtype=torch.uint8, device='cuda')
    nof.call('nof_sample_points', C.byref(sc), d_batch, tf, tio, nh, R, H, U.dev(u_occ), U.dev(u_dep), z, pts, valid, flags)
    torch.cuda.synchronize()
    assert cpu(flags)[0] == 0
    # z from the oracle sampler fed with the GPU's own (bit-identical) intervals
    z_same = O.sample_z(r_tio, (bb[:, 2] / np.sqrt((bb[:, :3] ** 2).sum(-1, dtype=np.float32))).astype(np.float32),
                        bb[:, 6], cfg, trunc, u_occ, u_dep)
    assert np.array_equal(cpu(z).view(np.uint32), z_same.view(np.uint32))          # bit-identical z samples
    assert np.abs(cpu(z) - z_ref.numpy()).max() < 1e-4
    # points + validity
    with torch.no_grad():
        fw = field.forward(torch.from_numpy(bb), torch.from_numpy(cpu(z)))
    assert np.abs(cpu(pts).reshape(R, S, 3) - fw['pts_w'].numpy()).max() < 1e-5
    v_ref = fw['valid_samples'].numpy().reshape(-1)
    assert (cpu(valid).astype(bool) != v_ref).mean() < 1e-3
    # Philox path: in range, deterministic, different per step
    z1 = torch.empty_like(z); z2 = torch.empty_like(z); z3 = torch.empty_like(z)
    nof.call('nof_sample_points', C.byref(sc), d_batch, tf, tio, nh, R, H, None, None, z1, pts, valid, flags)
    nof.call('nof_sample_points', C.byref(sc), d_batch, tf, tio, nh, R, H, None, None, z2, pts, valid, flags)
    sc.step = 1
    nof.call('nof_sample_points', C.byref(sc), d_batch, tf, tio, nh, R, H, None, None, z3, pts, valid, flags)
    torch.cuda.synchronize()
    assert torch.equal(z1, z2) and not torch.equal(z1, z3)
    has = cpu(nh) > 0
    assert np.isfinite(cpu(z1)).all() and (cpu(z1)[has] > 0).all()


# ------------------------------------------------------------------------------------------------
def _mlp_setup(nof, ns, nc, ff, L, precision, seed=0):
    torch.manual_seed(seed)
    n_view = 9 + ff
    shape = O.FieldShape(input_ch=2 * L, input_ch_views=n_view, num_layers=ns, num_layers_color=nc)
    params = O.init_mlp_params(shape)
    for W, b in params:                                  # make biases / weights less tame than the default init
        b.add_(torch.randn_like(b) * 0.1)
    desc, dims = nof.make_mlp_desc(ns, nc, 2 * L, n_view, precision)
    flat = torch.cat([torch.cat([W.reshape(-1), b.reshape(-1)]) for W, b in params])
    assert flat.numel() == desc.n_params == shape.n_params()
    return shape, params, desc, flat


def _pack(nof, desc, flat):
    packed = torch.empty(int(nof.load().nof_mlp_packed_bytes(C.byref(desc))), dtype=torch.uint8, device='cuda')
    nof.call('nof_mlp_pack', C.byref(desc), flat.cuda(), packed)
    return packed


# max |err| / max |ref| of the forward outputs against the PURE fp32 oracle.  3 / 4 = fp16 / bf16 with the hi+lo operand split
# in the forward kernels (the default mode): far inside north_star's 1e-3
TOL = {0: 2e-5, 1: 3e-2, 2: 4e-3, 3: 1e-4, 4: 4e-4}
# Gradients are checked against the oracle evaluated with the SAME operand rounding as the kernel (16-bit GEMM operands,
# fp32 accumulate = the reference's autocast path): against a pure-fp32 oracle a ReLU whose pre-activation rounds across 0
# flips one unit's whole gradient path (measured: fp16 1.6e-2 L2 / 8.6e-2 max, bf16 ~1e-1), which says nothing about the kernel.
ODT = {0: None, 1: torch.bfloat16, 2: torch.float16, 3: torch.float16, 4: torch.bfloat16}   # rounding model of the BACKWARD
TOL_L2 = {0: 2e-5, 1: 1.5e-2, 2: 2e-3, 3: 2e-3, 4: 1.5e-2}     # ||err|| / ||ref||
TOL_MAX = {0: 1e-4, 1: 8e-2, 2: 1e-2, 3: 1e-2, 4: 8e-2}      # max |err| / max |ref|


def rel_l2(got, ref):
    return float(np.linalg.norm((got - ref).ravel()) / np.linalg.norm(ref.ravel()))


def rel_max(got, ref):
    return float(np.abs(got - ref).max() / np.abs(ref).max())


def worst_elementwise(got, ref, rtol=1e-3, atol=1e-5):
    """north_star's "within 1e-3 rel" PER ELEMENT: max over the elements of |err| / (rtol |ref| + atol);
    <= 1 means every single value is within rtol of its reference value, with a small absolute floor for values near zero
    ("mixed abs/rel").  The max-norm figure `rel_max` stays beside it for the record."""
    got, ref = np.asarray(got, dtype=np.float64), np.asarray(ref, dtype=np.float64)
    return float((np.abs(got - ref) / (rtol * np.abs(ref) + atol)).max()) if ref.size else 0.0


@pytest.mark.parametrize("ns,nc,ff,L", [(2, 3, 0, 16), (3, 2, 2, 16), (2, 3, 2, 4), (2, 2, 0, 16), (3, 3, 0, 16)])
@pytest.mark.parametrize("precision", [0, 1, 2, 3, 4])
def test_mlp_forward(nof, ns, nc, ff, L, precision):
    shape, params, desc, flat = _mlp_setup(nof, ns, nc, ff, L, precision)
    R, S = 37, 40
    B = R * S                                            # not a multiple of 32: exercises the tail tile
    torch.manual_seed(5)
    feat = torch.randn(B, 2 * L) * 0.5
    view = torch.zeros(R, 16)
    view[:, :9 + ff] = torch.randn(R, 9 + ff)
    x = torch.cat([feat, view[:, :9 + ff].repeat_interleave(S, 0)], -1)
    ref = O.mlp_forward(shape, params, x).detach().numpy()
    ref_m = O.mlp_forward(shape, params, x, None if precision >= 3 else ODT[precision]).detach().numpy()
    d_feat = feat.reshape(B, L, 2).permute(1, 0, 2).contiguous().cuda()
    raw = torch.zeros(B, 4, device='cuda')
    packed = _pack(nof, desc, flat)
    nof.call('nof_mlp_fwd', C.byref(desc), packed, d_feat, L, view.cuda(), S, raw, None, B)
    sdf = torch.zeros(B, device='cuda')
    nof.call('nof_mlp_sdf', C.byref(desc), packed, d_feat, L, sdf, B)
    torch.cuda.synchronize()
    scale = np.abs(ref).max()
    assert np.abs(cpu(raw) - ref).max() / scale < TOL[precision]
    assert np.abs(cpu(sdf) - ref[:, 3]).max() / scale < TOL[precision]
    # same operand rounding as the kernel: what is left is accumulation order, which can still move an activation across
    # a 16-bit rounding boundary (one ulp of one operand)
    assert np.abs(cpu(raw) - ref_m).max() / scale < {0: 2e-5, 1: 4e-3, 2: 5e-4, 3: 1e-4, 4: 4e-4}[precision]


@pytest.mark.parametrize("ns,nc,ff,L", [(2, 3, 0, 16), (3, 2, 2, 16), (2, 3, 2, 4), (3, 3, 0, 16)])
@pytest.mark.parametrize("precision,split", [(0, False), (1, False), (2, False), (1, True), (2, True), (3, True), (4, True)])
def test_mlp_backward(nof, ns, nc, ff, L, precision, split):
    """split=True: the two-kernel path (colour net, sigma net) fed by the forward kernel's sigma-head output;
    split=False: the fused kernel.  Both must match the oracle."""
    shape, params, desc, flat = _mlp_setup(nof, ns, nc, ff, L, precision, seed=2)
    R, S = 21, 48                                        # S not a multiple of 32: tiles straddle rays
    B = R * S
    torch.manual_seed(6)
    feat = (torch.randn(B, 2 * L) * 0.5).requires_grad_(True)
    view_t = torch.randn(R, 9 + ff).requires_grad_(True)
    ps = [[W.clone().requires_grad_(True), b.clone().requires_grad_(True)] for W, b in params]
    x = torch.cat([feat, view_t.repeat_interleave(S, 0)], -1)
    out = O.mlp_forward(shape, ps, x, ODT[precision], split_forward=precision >= 3)
    draw = torch.randn(B, 4)
    (out * draw).sum().backward()
    view = torch.zeros(R, 16)
    view[:, :9 + ff] = view_t.detach()
    d_feat = feat.detach().reshape(B, L, 2).permute(1, 0, 2).contiguous().cuda()
    nblk = nof.load().nof_mlp_bwd_blocks()
    dfeat = torch.full((L, B, 2), 3.0, device='cuda')
    dview = torch.zeros(R, 16, device='cuda')
    partials = torch.full((nblk, desc.n_params), 5.0, device='cuda')
    packed = _pack(nof, desc, flat)
    sig = dsig = None
    if (ns, nc) == (3, 3) and precision != 0 and not split:
        # the fused kernel's fragments + operand slots for six 16-bit layers exceed the 160 KB LDS: loud error, not a crash
        with pytest.raises(nof.NofError):
            nof.call('nof_mlp_bwd', C.byref(desc), packed, d_feat, L, view.cuda(), S, draw.cuda(), None, None, dfeat, dview, partials, B)
        return
    if split:
        sig = torch.zeros(B, 16, dtype=torch.int16, device='cuda')
        dsig = torch.zeros(B, 16, dtype=torch.int16, device='cuda')
        raw = torch.zeros(B, 4, device='cuda')
        nof.call('nof_mlp_fwd', C.byref(desc), packed, d_feat, L, view.cuda(), S, raw, sig, B)
    nof.call('nof_mlp_bwd', C.byref(desc), packed, d_feat, L, view.cuda(), S, draw.cuda(), sig, dsig, dfeat, dview, partials, B)
    gflat = torch.zeros(desc.n_params, device='cuda')
    nof.call('nof_reduce_partials', partials, nblk, desc.n_params, gflat, None)
    torch.cuda.synchronize()
    ref_df = feat.grad.numpy()
    got_df = cpu(dfeat).transpose(1, 0, 2).reshape(B, 2 * L)
    ref_dv = view_t.grad.numpy()
    ref_g = torch.cat([torch.cat([W.grad.reshape(-1), b.grad.reshape(-1)]) for W, b in ps]).numpy()
    got_g = cpu(gflat)
    report = {'dfeat': (rel_l2(got_df, ref_df), rel_max(got_df, ref_df)),
              'dview': (rel_l2(cpu(dview)[:, :9 + ff], ref_dv), rel_max(cpu(dview)[:, :9 + ff], ref_dv))}
    for l in range(ns + nc):                             # per layer so that a small layer cannot hide behind a big one
        lo, hi = desc.w_off[l], desc.b_off[l] + desc.out_dim[l]
        report[f'layer{l}'] = (rel_l2(got_g[lo:hi], ref_g[lo:hi]), rel_max(got_g[lo:hi], ref_g[lo:hi]))
    print('mlp_bwd precision', precision, {k: (f'{a:.2e}', f'{b:.2e}') for k, (a, b) in report.items()})
    assert (cpu(dview)[:, 9 + ff:] == 0).all()
    for k, (l2, mx) in report.items():
        assert l2 < TOL_L2[precision] and mx < TOL_MAX[precision], (k, l2, mx)


@pytest.mark.parametrize("ns,nc,hidden,ff,L", [(4, 4, 128, 0, 16), (2, 3, 128, 2, 16), (4, 4, 64, 0, 16), (3, 4, 128, 2, 4)])
@pytest.mark.parametrize("precision", [1, 2])
def test_mlp_wide_forward_backward(nof, ns, nc, hidden, ff, L, precision):
    """The wide / deep shapes (BASELINE cfg5: SDF 4x128 + colour 4x128, fp16) through nof_mlp_wide_fwd / _sdf / _bwd: outputs,
    dfeat, dview and every layer's dW / db against the oracle with the same 16-bit operand rounding."""
    torch.manual_seed(3)
    n_view = 9 + ff
    shape = O.FieldShape(input_ch=2 * L, input_ch_views=n_view, num_layers=ns, hidden_dim=hidden, num_layers_color=nc,
                         hidden_dim_color=hidden)
    params = O.init_mlp_params(shape)
    for W, b in params:
        b.add_(torch.randn_like(b) * 0.1)
    desc, dims = nof.make_mlp_desc(ns, nc, 2 * L, n_view, precision, hidden=hidden)
    flat = torch.cat([torch.cat([W.reshape(-1), b.reshape(-1)]) for W, b in params])
    assert flat.numel() == desc.n_params == shape.n_params()
    R, S = 23, 48                                        # B not a multiple of 32, S not a multiple of 32
    B = R * S
    feat = (torch.randn(B, 2 * L) * 0.5).requires_grad_(True)
    view_t = torch.randn(R, n_view).requires_grad_(True)
    ps = [[W.clone().requires_grad_(True), b.clone().requires_grad_(True)] for W, b in params]
    x = torch.cat([feat, view_t.repeat_interleave(S, 0)], -1)
    ref32 = O.mlp_forward(shape, params, x.detach()).detach().numpy()
    out = O.mlp_forward(shape, ps, x, ODT[precision])
    draw = torch.randn(B, 4)
    (out * draw).sum().backward()
    ref_m = out.detach().numpy()
    view = torch.zeros(R, 16)
    view[:, :n_view] = view_t.detach()
    d_feat = feat.detach().reshape(B, L, 2).permute(1, 0, 2).contiguous().cuda()
    packed = _pack(nof, desc, flat)
    ws = torch.zeros(int(nof.load().nof_mlp_wide_workspace_bytes(C.byref(desc), B)), dtype=torch.uint8, device='cuda')
    raw = torch.zeros(B, 4, device='cuda')
    nof.call('nof_mlp_wide_fwd', C.byref(desc), packed, d_feat, L, view.cuda(), S, raw, ws, B)
    sdf = torch.zeros(B, device='cuda')
    nof.call('nof_mlp_wide_sdf', C.byref(desc), packed, d_feat, L, sdf, B)
    with pytest.raises(nof.NofError):                    # the narrow entry points name the wide ones instead of mis-computing
        nof.call('nof_mlp_fwd', C.byref(desc), packed, d_feat, L, view.cuda(), S, raw, None, B)
    desc3, _ = nof.make_mlp_desc(ns, nc, 2 * L, n_view, {1: 4, 2: 3}[precision], hidden=hidden)
    with pytest.raises(nof.NofError, match='operand split'):     # 'fp16x3' / 'bf16x3' do not exist on the wide path: refused, not downgraded
        nof.call('nof_mlp_wide_fwd', C.byref(desc3), packed, d_feat, L, view.cuda(), S, raw, ws, B)
    rows = nof.load().nof_mlp_wide_partial_rows()
    dfeat = torch.full((L, B, 2), 3.0, device='cuda')
    dview = torch.zeros(R, 16, device='cuda')
    partials = torch.full((rows, desc.n_params), 5.0, device='cuda')
    nof.call('nof_mlp_wide_bwd', C.byref(desc), packed, d_feat, L, view.cuda(), S, draw.cuda(), ws, dfeat, dview, partials, B)
    gflat = torch.zeros(desc.n_params, device='cuda')
    nof.call('nof_reduce_partials', partials, rows, desc.n_params, gflat, None)
    torch.cuda.synchronize()
    scale = np.abs(ref32).max()
    e32, em = np.abs(cpu(raw) - ref32).max() / scale, np.abs(cpu(raw) - ref_m).max() / scale
    print(f'wide ({ns},{nc},{hidden}) precision {precision}: forward vs fp32 {e32:.2e}, vs same rounding {em:.2e}')
    assert e32 < TOL[precision] and em < {1: 6e-3, 2: 8e-4}[precision]
    assert np.abs(cpu(sdf) - ref32[:, 3]).max() / scale < TOL[precision]
    ref_df, ref_dv = feat.grad.numpy(), view_t.grad.numpy()
    got_df = cpu(dfeat).transpose(1, 0, 2).reshape(B, 2 * L)
    ref_g = torch.cat([torch.cat([W.grad.reshape(-1), b.grad.reshape(-1)]) for W, b in ps]).numpy()
    got_g = cpu(gflat)
    report = {'dfeat': (rel_l2(got_df, ref_df), rel_max(got_df, ref_df)),
              'dview': (rel_l2(cpu(dview)[:, :n_view], ref_dv), rel_max(cpu(dview)[:, :n_view], ref_dv))}
    for l in range(ns + nc):
        lo, hi = desc.w_off[l], desc.b_off[l] + desc.out_dim[l]
        report[f'layer{l}'] = (rel_l2(got_g[lo:hi], ref_g[lo:hi]), rel_max(got_g[lo:hi], ref_g[lo:hi]))
    print('wide bwd', {k: (f'{a:.2e}', f'{b:.2e}') for k, (a, b) in report.items()})
    assert (cpu(dview)[:, n_view:] == 0).all()
    for k, (l2, mx) in report.items():
        assert l2 < 1.5 * TOL_L2[precision] and mx < 1.5 * TOL_MAX[precision], (k, l2, mx)


def test_mlp_backward_fp16_loss_scale(nof):
    """Loss gradients of a 1/(R*S)-normalised loss are ~1e-7: as fp16 MFMA operands they are subnormal or zero (binary16's
    smallest normal is 6.1e-5, its subnormal step 6e-8).  NofMlpDesc.grad_scale multiplies them by a power of two where
    they enter the kernel and divides every fp32 output by it (the reference's GradScaler, nerf_runner.py:159,756-761):
    with it the fp16 backward matches the oracle as well as at unit scale; without it it does not."""
    ns, nc, ff, L, precision = 3, 2, 0, 16, 2
    shape, params, desc, flat = _mlp_setup(nof, ns, nc, ff, L, precision, seed=2)
    R, S = 21, 48
    B = R * S
    torch.manual_seed(6)
    feat = (torch.randn(B, 2 * L) * 0.5).requires_grad_(True)
    view_t = torch.randn(R, 9).requires_grad_(True)
    ps = [[W.clone().requires_grad_(True), b.clone().requires_grad_(True)] for W, b in params]
    out = O.mlp_forward(shape, ps, torch.cat([feat, view_t.repeat_interleave(S, 0)], -1), torch.float16)
    draw = torch.randn(B, 4) * 3e-7
    (out * draw).sum().backward()
    ref_df = feat.grad.numpy()
    ref_g = torch.cat([torch.cat([W.grad.reshape(-1), b.grad.reshape(-1)]) for W, b in ps]).numpy()
    view = torch.zeros(R, 16)
    view[:, :9] = view_t.detach()
    d_feat = feat.detach().reshape(B, L, 2).permute(1, 0, 2).contiguous().cuda()
    nblk = nof.load().nof_mlp_bwd_blocks()
    packed = _pack(nof, desc, flat)
    errs = {}
    for scale in (0.0, 65536.0):
        desc.grad_scale = scale
        dfeat = torch.zeros(L, B, 2, device='cuda')
        dview = torch.zeros(R, 16, device='cuda')
        partials = torch.zeros(nblk, desc.n_params, device='cuda')
        sig = torch.zeros(B, 16, dtype=torch.int16, device='cuda')
        dsig = torch.zeros(B, 16, dtype=torch.int16, device='cuda')
        raw = torch.zeros(B, 4, device='cuda')
        nof.call('nof_mlp_fwd', C.byref(desc), packed, d_feat, L, view.cuda(), S, raw, sig, B)
        nof.call('nof_mlp_bwd', C.byref(desc), packed, d_feat, L, view.cuda(), S, draw.cuda(), sig, dsig, dfeat, dview, partials, B)
        gflat = torch.zeros(desc.n_params, device='cuda')
        nof.call('nof_reduce_partials', partials, nblk, desc.n_params, gflat, None)
        torch.cuda.synchronize()
        errs[scale] = (rel_l2(cpu(dfeat).transpose(1, 0, 2).reshape(B, 2 * L), ref_df), rel_l2(cpu(gflat), ref_g))
    print('fp16 backward, |draw| ~ 3e-7: rel-L2 (dfeat, dW) without / with the loss scale:', errs)
    assert max(errs[65536.0]) < TOL_L2[2]
    assert min(errs[0.0]) > 10 * TOL_L2[2]                 # the unscaled fp16 backward is NOT usable at this magnitude


# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("fs_rgb", [0.0, 0.5])
def test_composite_loss(nof, fs_rgb):
    cfg, occ, c2w, batch = _scene(nof, R=200)
    cfg['fs_rgb_weight'] = fs_rgb
    R = batch.shape[0]
    S = 96
    rng = np.random.default_rng(4)
    trunc = O.get_truncation(cfg, 0)
    depth = batch[:, 6:7]
    z = (np.where(depth > 50, 2.8, depth) + rng.uniform(-3 * trunc, 3 * trunc, (R, S))).astype(np.float32)
    raw = torch.from_numpy(rng.normal(size=(R, S, 4)).astype(np.float32) * 1.5).requires_grad_(True)
    valid = torch.from_numpy(rng.random((R, S)) > 0.1)
    valid[3] = False
    tb = torch.from_numpy(batch)
    rgb_map, w = O.raw2outputs(raw, torch.from_numpy(z), tb[:, 6], valid, cfg, trunc)
    out = O.losses(rgb_map, raw, torch.from_numpy(z), valid, tb, cfg, trunc)
    out['loss'].backward()
    lc = nof.NofLossCfg(trunc, cfg['neg_trunc_ratio'], cfg['sdf_lambda'], cfg['near'] * cfg['sc_factor'],
                        cfg['far'] * cfg['sc_factor'], cfg['rgb_weight'], cfg['fs_weight'], cfg['trunc_weight'],
                        cfg['empty_weight'], cfg['fs_sdf'], cfg['fs_rgb_weight'], cfg['first_frame_weight'], 1.0)
    g_rgb = torch.empty(R, 3, device='cuda')
    g_w = torch.empty(R, S, device='cuda')
    g_draw = torch.empty(R, S, 4, device='cuda')
    g_loss = torch.zeros(8, device='cuda')
    g_rows = torch.empty(R, 8, device='cuda')
    nof.call('nof_composite_loss', C.byref(lc), raw.detach().cuda(), U.dev(z), valid.to(torch.uint8).cuda(), tb.cuda(),
             R, S, g_rgb, g_w, g_draw, g_rows, g_loss)
    torch.cuda.synchronize()
    assert np.abs(cpu(g_rgb) - rgb_map.detach().numpy()).max() < 2e-6
    assert np.abs(cpu(g_w) - w.detach().numpy()).max() < 2e-6
    ref_d = raw.grad.numpy()
    assert np.abs(cpu(g_draw) - ref_d).max() < 1e-5 * np.abs(ref_d).max() + 1e-9
    L = cpu(g_loss)
    assert abs(L[0] - out['loss'].item()) < 1e-4 * abs(out['loss'].item())
    assert abs(L[1] - out['rgb_loss'].item()) < 1e-4 * abs(out['rgb_loss'].item()) + 1e-9
    assert abs(L[2] - out['fs_loss'].item()) < 1e-4 * abs(out['fs_loss'].item()) + 1e-9
    assert abs(L[3] - out['sdf_loss'].item()) < 1e-4 * abs(out['sdf_loss'].item()) + 1e-9


# ------------------------------------------------------------------------------------------------
def test_adam_matches_torch(nof):
    n, nb = 10000, 9000
    torch.manual_seed(0)
    p0 = torch.randn(n)
    pa = p0[:nb].clone().requires_grad_(True)
    pb = p0[nb:].clone().requires_grad_(True)
    opt = torch.optim.Adam([{'params': [pa], 'lr': 0.01}, {'params': [pb], 'lr': 0.003}], betas=(0.9, 0.999), eps=1e-15)
    p, m, v = p0.clone().cuda(), torch.zeros(n, device='cuda'), torch.zeros(n, device='cuda')
    for step in range(1, 6):
        g = torch.randn(n) * (10.0 ** -step)
        g[::5] = 0
        pa.grad, pb.grad = g[:nb].clone(), g[nb:].clone()
        opt.step()
        gd = g.clone().cuda()
        nof.call('nof_adam_step', p, gd, m, v, n, nb, C.c_float(0.01), C.c_float(0.003), C.c_float(0.9), C.c_float(0.999),
                 C.c_float(1e-15), step, None)
        torch.cuda.synchronize()
        assert (gd == 0).all()
        ref = torch.cat([pa.detach(), pb.detach()]).numpy()
        assert np.abs(cpu(p) - ref).max() < 2e-6


@pytest.mark.parametrize('off,n,nb', [(0, 10000, 9001), (1, 4099, 4098), (2, 5, 3), (3, 8190, 0), (1, 2, 2), (0, 3, 3)])
def test_adam_ranges_are_bit_identical(nof, off, n, nb):
    """The update is element-wise: a range that starts off a 16-byte boundary, ends inside a 16-byte group, or whose learning-rate
    boundary falls inside one gives the bits of the element-by-element form (here: each entry updated by its own 1-entry call
    would be slow, so the check is against a call on buffers with a DIFFERENT alignment per array, which takes the scalar path),
    and entries outside the range are untouched.  The split optimiser step of field.py (table first, the rest later) rests on it."""
    torch.manual_seed(n + off)
    N = n + 16
    base = [torch.randn(N, device='cuda') for _ in range(4)]
    base[3].abs_()                                               # exp_avg_sq >= 0
    if n > 1000:
        # the 16-byte path leaves out stores that would write back the same bits (a zero gradient is not zeroed again; an entry
        # whose gradient AND moments are all-zero bits is a fixed point).  The scalar path below does the arithmetic and stores
        # everything: equal bits prove the shortcut, including -0.0 (not all-zero bits: takes the arithmetic path) and NaN
        base[1][off + 100:off + 400] = 0                         # gradient zero, moments not
        for k in (1, 2, 3):
            base[k][off + 500:off + 900] = 0                     # never-touched entries
        base[2][off + 600] = -0.0
        base[1][off + 700] = -0.0
        base[3][off + 800] = -0.0
        base[0][off + 520] = -0.0                                # a parameter that is -0 stays -0
        base[0][off + 530] = float('nan')
    a = [x.clone() for x in base]
    args = (C.c_float(0.01), C.c_float(0.003), C.c_float(0.9), C.c_float(0.999), C.c_float(1e-15), 7, None)
    nof.call('nof_adam_step', *[x[off:off + n] for x in a], n, nb, *args)
    # scalar path: the four arrays at four different offsets from a 16-byte boundary
    b = [torch.zeros(N + 4, device='cuda') for _ in range(4)]
    for k, (x, y) in enumerate(zip(b, base)):
        x[k:k + n] = y[off:off + n]
    nof.call('nof_adam_step', *[x[k:k + n] for k, x in enumerate(b)], n, nb, *args)
    torch.cuda.synchronize()
    for k, (x, y, z) in enumerate(zip(a, b, base)):
        assert torch.equal(x[off:off + n].view(torch.int32), y[k:k + n].view(torch.int32)), k         # bits (NaN, -0 included)
        assert torch.equal(x[:off], z[:off]) and torch.equal(x[off + n:], z[off + n:]), k
    assert (a[1][off:off + n] == 0).all()


@pytest.mark.parametrize('R,S,F,ff', [(4096, 192, 64, 0), (1000, 64, 7, 2), (33, 32, 3, 16), (257, 96, 300, 1)])
def test_pose_gradients_slot_sums_equal_batch_search(nof, R, S, F, ff):
    """nof_pose_grad_accum + nof_pose_reduce_bwd in their two forms: per-ray rows searched by frame (no atomics, fixed order) and
    the rows added on the fly to 16 partial sums per frame (what the step uses).  Same per-ray rows bit for bit; per-frame sums,
    pose and frame-feature gradients equal up to fp32 summation order; dview rows and the slots come back zeroed."""
    rng = np.random.default_rng(R + F)
    B = R * S
    batch = rng.normal(size=(R, 12)).astype(np.float32)
    batch[:, 8] = rng.integers(0, F, size=R)
    batch[:5, 8] = 0                                                   # frame 0: no pose correction, but frame features
    pose = (rng.normal(size=(F, 6)) * 0.1).astype(np.float32)
    c2w = np.tile(np.eye(4, dtype=np.float32), (F, 1, 1)) + rng.normal(size=(F, 4, 4)).astype(np.float32) * 0.1
    mt, mr = 0.02, np.float32(np.deg2rad(5.0))
    tf = torch.empty(F, 12, device='cuda')
    nof.call('nof_pose_fwd', U.dev(pose), U.dev(c2w.reshape(F, 16)), C.c_float(mt), C.c_float(mr), tf, F)
    dpts = U.dev(rng.normal(size=(B, 3)).astype(np.float32) * 1e-3)
    dview0 = rng.normal(size=(R, 16)).astype(np.float32)
    z = U.dev(rng.random((R, S)).astype(np.float32) + 0.2)
    out = {}
    for mode in ('search', 'slots'):
        dview = U.dev(dview0)
        g_ray = torch.full((R, 12), 7.0, device='cuda')
        gp, gf, gd = torch.zeros(F, 6, device='cuda'), torch.zeros(F, max(ff, 1), device='cuda'), torch.zeros(F, 12, device='cuda')
        slots = torch.zeros(F * 16 * 28, device='cuda') if mode == 'slots' else None
        nof.call('nof_pose_grad_accum', dpts, dview, U.dev(batch), z, U.dev(c2w.reshape(F, 16)), tf, ff, 3 if ff <= 7 else 1, R, S,
                 g_ray, slots)                                     # (view row = [ff features | SH]: 9 SH coefficients need ff <= 7)
        if mode == 'slots':
            nof.call('nof_pose_reduce_bwd', U.dev(pose), None, None, None, R, ff, C.c_float(mt), C.c_float(mr), gp,
                     gf if ff else None, gd, F, 0, slots)
        else:
            nof.call('nof_pose_reduce_bwd', U.dev(pose), g_ray, dview, U.dev(batch), R, ff, C.c_float(mt), C.c_float(mr), gp,
                     gf if ff else None, gd, F, 1, None)
        torch.cuda.synchronize()
        assert (dview == 0).all()
        if slots is not None:
            assert (slots == 0).all()
        out[mode] = [cpu(x) for x in (g_ray, gd, gp, gf)]
    a, b = out['search'], out['slots']
    assert np.array_equal(a[0], b[0])                                  # the per-ray rows do not depend on the mode
    assert (a[0][:5] == 0).all()
    # reference for the per-frame sums: float64 over the rows
    want = np.zeros((F, 12))
    np.add.at(want, batch[:, 8].astype(int), a[0].astype(np.float64))
    scale = np.abs(want).max() + 1e-12
    for k, name in ((1, 'g_delta'), (2, 'grad_pose'), (3, 'grad_feat')):
        s = max(np.abs(a[k]).max(), 1e-12)
        assert np.abs(a[k] - b[k]).max() <= 2e-5 * s, name
    assert np.abs(a[1] - want).max() <= 2e-5 * scale and np.abs(b[1] - want).max() <= 2e-5 * scale
    if ff:
        wf = np.zeros((F, ff))
        np.add.at(wf, batch[:, 8].astype(int), dview0[:, :ff].astype(np.float64))
        assert np.abs(b[3] - wf).max() <= 2e-5 * (np.abs(wf).max() + 1e-12)


@pytest.mark.parametrize("ns,nc,ff,L,T,finest", [(3, 2, 0, 16, 19, 256), (2, 3, 2, 16, 14, 256), (2, 2, 0, 8, 12, 128), (3, 3, 0, 16, 19, 512),
                                                  (2, 3, 0, 4, 14, 64)])
@pytest.mark.parametrize("precision", [1, 2, 3, 4])
def test_fused_encode_mlp_forward_equals_the_two_launches(nof, ns, nc, ff, L, T, finest, precision):
    """nof_encode_mlp_fwd (hash encode in the MLP forward kernel, the embedding never in HBM) against nof_hash_encode_fwd +
    nof_mlp_fwd on the same points: the encode is the same arithmetic per level and the chain the same MFMA sequence, so raw and
    the sigma hand-off must be BIT-identical; `featq` must be the fp32 embedding rounded to the operand type in operand order;
    and the backward fed from featq (nof_mlp_bwd_featq) must give the bits of the backward fed from the fp32 embedding.
    Geometries: dense + hashed levels, the 257 / 513 resolution quirk (finest 512), L < 8 (upper lane half idle), ragged B."""
    shape, params, desc, flat = _mlp_setup(nof, ns, nc, ff, L, precision, seed=3)
    g, geo = U.make_grids(nof, L=L, T=T, finest=finest)
    R, S = 37, 40                                        # B = 1480: not a multiple of 64 (a ragged tile pair) nor of 32
    B = R * S
    rng = np.random.default_rng(7)
    pts = rng.uniform(-1, 1, size=(B, 3)).astype(np.float32)
    pts[:6] = U.test_points(6)                           # corners, centre, out of range, an ulp outside
    pts[100:140] = pts[100] + np.linspace(0, 1e-3, 40, dtype=np.float32)[:, None]    # a run inside one cell, like samples of a ray
    table = (rng.uniform(-1, 1, size=(geo.n_entries, 2)) * 0.3).astype(np.float32)
    view = torch.zeros(R, 16)
    view[:, :9 + ff] = torch.randn(R, 9 + ff)
    d_pts, d_table, d_view = U.dev(pts), U.dev(table), view.cuda()
    packed = _pack(nof, desc, flat)
    feat = torch.empty(L, B, 2, device='cuda')
    raw_a, raw_b = torch.zeros(B, 4, device='cuda'), torch.ones(B, 4, device='cuda')
    sig_a, sig_b = torch.zeros(B, 16, dtype=torch.int16, device='cuda'), torch.ones(B, 16, dtype=torch.int16, device='cuda')
    featq = torch.full((B, 32), 7, dtype=torch.int16, device='cuda')
    nof.call('nof_hash_encode_fwd', C.byref(g), d_pts, d_table, feat, B)
    nof.call('nof_mlp_fwd', C.byref(desc), packed, feat, L, d_view, S, raw_a, sig_a, B)
    nof.call('nof_encode_mlp_fwd', C.byref(g), C.byref(desc), packed, d_table, d_pts, d_view, S, raw_b, sig_b, featq, B)
    torch.cuda.synchronize()
    assert torch.isfinite(raw_a).all()
    assert torch.equal(raw_a, raw_b), (raw_a - raw_b).abs().max().item()
    assert torch.equal(sig_a, sig_b)
    # featq[b][hi][r] = operand-type rounding of feature 16 hi + r of sample b
    odt = torch.bfloat16 if precision in (1, 4) else torch.float16
    want = feat.permute(1, 0, 2).reshape(B, 32 if L == 16 else 2 * L)
    if L < 16:
        want = torch.cat([want, torch.zeros(B, 32 - 2 * L, device='cuda')], 1)
    assert torch.equal(featq.view(odt), want.to(odt))
    # without the copy (featq = NULL) the outputs are the same
    raw_c = torch.zeros(B, 4, device='cuda')
    nof.call('nof_encode_mlp_fwd', C.byref(g), C.byref(desc), packed, d_table, d_pts, d_view, S, raw_c, None, None, B)
    torch.cuda.synchronize()
    assert torch.equal(raw_c, raw_a)
    # the backward from featq == the backward from the fp32 embedding (S >= 32 is required there)
    draw = torch.randn(B, 4, device='cuda')
    draw[64:256] = 0                                     # some all-zero tiles
    nblk = nof.load().nof_mlp_bwd_blocks()
    outs = []
    for use_q in (False, True):
        dsig = torch.zeros(B, 16, dtype=torch.int16, device='cuda')
        dfeat = torch.full((L, B, 2), 3.0, device='cuda')
        dview = torch.zeros(R, 16, device='cuda')
        partials = torch.full((nblk, desc.n_params), 5.0, device='cuda')
        if use_q:
            nof.call('nof_mlp_bwd_featq', C.byref(desc), packed, featq, L, d_view, S, draw, sig_a, dsig, dfeat, dview, partials, None, B)
        else:
            nof.call('nof_mlp_bwd_tiles', C.byref(desc), packed, feat, L, d_view, S, draw, sig_a, dsig, dfeat, dview, partials, None, B)
        torch.cuda.synchronize()
        outs.append((dfeat, partials, dsig))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]) and torch.equal(outs[0][2], outs[1][2])
    # round 6: nof_mlp_bwd_featq runs both networks' halves as ONE launch where their workgroup shapes agree (two colour layers);
    # nof_mlp_bwd_featq_two_launches keeps them apart: the same bits, over the whole batch and over a work list
    tl = torch.zeros(int(nof.load().nof_tile_list_bytes(B)), dtype=torch.uint8, device='cuda')
    nof.call('nof_tile_list_build', draw, B, 0, tl)
    for lst in (None, tl):
        res = []
        for entry in ('nof_mlp_bwd_featq', 'nof_mlp_bwd_featq_two_launches'):
            dsig = torch.zeros(B, 16, dtype=torch.int16, device='cuda')
            dfeat = torch.full((L, B, 2), 3.0, device='cuda')
            dview = torch.zeros(R, 16, device='cuda')
            partials = torch.full((nblk, desc.n_params), 5.0, device='cuda')
            nof.call(entry, C.byref(desc), packed, featq, L, d_view, S, draw, sig_a, dsig, dfeat, dview, partials, lst, B)
            torch.cuda.synchronize()
            res.append((dfeat, partials, dsig, dview))
        assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1]) and torch.equal(res[0][2], res[1][2])
        assert (res[0][3] - res[1][3]).abs().max().item() <= 1e-5 * res[1][3].abs().max().item()      # (atomics: order only)
        if lst is None:
            assert torch.equal(res[0][0], outs[1][0])


@pytest.mark.parametrize("ns,nc,hidden,ff,L,T,finest", [(4, 4, 128, 0, 16, 19, 512), (2, 3, 128, 2, 16, 14, 256), (4, 4, 64, 0, 16, 20, 512),
                                                         (3, 4, 128, 2, 4, 14, 64)])
@pytest.mark.parametrize("precision", [1, 2])
def test_wide_fused_encode_forward_equals_the_two_launches(nof, ns, nc, hidden, ff, L, T, finest, precision):
    """nof_encode_mlp_wide_fwd (round 6: hash encode inside the wide sigma forward, colour net behind it) against
    nof_hash_encode_fwd + nof_mlp_wide_fwd on the same points: raw and the sigma hand-off in the workspace BIT-identical, featq =
    the fp32 embedding rounded to the operand type in operand order, and nof_mlp_wide_bwd_parts fed from featq gives the bits of
    the backward fed from the fp32 embedding.  Ragged B (not a multiple of 64 nor 32), dense + hashed levels, the 513 quirk, L < 8."""
    torch.manual_seed(11)
    n_view = 9 + ff
    shape = O.FieldShape(input_ch=2 * L, input_ch_views=n_view, num_layers=ns, hidden_dim=hidden, num_layers_color=nc,
                         hidden_dim_color=hidden)
    params = O.init_mlp_params(shape)
    for W, b in params:
        b.add_(torch.randn_like(b) * 0.1)
    desc, dims = nof.make_mlp_desc(ns, nc, 2 * L, n_view, precision, hidden=hidden)
    flat = torch.cat([torch.cat([W.reshape(-1), b.reshape(-1)]) for W, b in params])
    g, geo = U.make_grids(nof, L=L, T=T, finest=finest)
    R, S = 37, 40
    B = R * S
    rng = np.random.default_rng(7)
    pts = rng.uniform(-1, 1, size=(B, 3)).astype(np.float32)
    pts[:6] = U.test_points(6)
    pts[100:140] = pts[100] + np.linspace(0, 1e-3, 40, dtype=np.float32)[:, None]
    table = (rng.uniform(-1, 1, size=(geo.n_entries, 2)) * 0.3).astype(np.float32)
    view = torch.zeros(R, 16)
    view[:, :n_view] = torch.randn(R, n_view)
    d_pts, d_table, d_view = U.dev(pts), U.dev(table), view.cuda()
    packed = _pack(nof, desc, flat)
    nbytes = int(nof.load().nof_mlp_wide_workspace_bytes(C.byref(desc), B))
    ws_a = torch.zeros(nbytes, dtype=torch.uint8, device='cuda')
    ws_b = torch.full((nbytes,), 9, dtype=torch.uint8, device='cuda')
    feat = torch.empty(L, B, 2, device='cuda')
    raw_a, raw_b = torch.zeros(B, 4, device='cuda'), torch.ones(B, 4, device='cuda')
    featq = torch.full((B, 32), 7, dtype=torch.int16, device='cuda')
    nof.call('nof_hash_encode_fwd', C.byref(g), d_pts, d_table, feat, B)
    nof.call('nof_mlp_wide_fwd', C.byref(desc), packed, feat, L, d_view, S, raw_a, ws_a, B)
    nof.call('nof_encode_mlp_wide_fwd', C.byref(g), C.byref(desc), packed, d_table, d_pts, d_view, S, raw_b, ws_b, featq, B)
    torch.cuda.synchronize()
    assert torch.isfinite(raw_a).all()
    assert torch.equal(raw_a, raw_b), (raw_a - raw_b).abs().max().item()
    odt = torch.bfloat16 if precision == 1 else torch.float16
    want = feat.permute(1, 0, 2).reshape(B, 2 * L)
    if L < 16:
        want = torch.cat([want, torch.zeros(B, 32 - 2 * L, device='cuda')], 1)
    assert torch.equal(featq.view(odt), want.to(odt))
    raw_c = torch.zeros(B, 4, device='cuda')
    nof.call('nof_encode_mlp_wide_fwd', C.byref(g), C.byref(desc), packed, d_table, d_pts, d_view, S, raw_c, ws_b, None, B)
    torch.cuda.synchronize()
    assert torch.equal(raw_c, raw_a)
    # the backward: from the fp32 embedding + the two-launch forward's workspace, and from featq + the fused forward's workspace
    draw = torch.randn(B, 4, device='cuda')
    draw[64:256] = 0
    rows = nof.load().nof_mlp_wide_partial_rows()
    outs = []
    for use_q in (False, True):
        dfeat = torch.full((L, B, 2), 3.0, device='cuda')
        dview = torch.zeros(R, 16, device='cuda')
        partials = torch.full((rows, desc.n_params), 5.0, device='cuda')
        nof.call('nof_mlp_wide_bwd_parts', C.byref(desc), packed, None if use_q else feat, featq if use_q else None, L, d_view, S, draw,
                 ws_b if use_q else ws_a, dfeat, dview, partials, None, 3, B)
        gflat = torch.zeros(desc.n_params, device='cuda')
        nof.call('nof_reduce_partials', partials, rows, desc.n_params, gflat, None)
        torch.cuda.synchronize()
        outs.append((dfeat, dview, gflat))
    for a, b in zip(*outs):
        assert torch.equal(a, b), (a - b).abs().max().item()


@pytest.mark.parametrize("precision", ['fp16x3', 'bf16x3'])
def test_fused_forward_is_repeatable(nof, precision):
    """Round 4's fault (profiles/r04_fused_forward_race.txt): with more than one level's gathers in flight the fused forward returned,
    in a few 16-sample groups per million samples and differently on every run, wrong features of one level in lanes 48-63.  Root
    cause (round 5, tests/test_gpu_erratum.py): the SLP vectoriser had blended that level with `v_pk_mul_f32 ... op_sel:[0,1]`, a form
    gfx950 executes wrongly in its last quarter-wave while another wave of the SIMD runs an MFMA; the library is built without that
    vectoriser now.  Kept as the product-level guard: ten launches over 196 608 ray-ordered samples (consecutive lanes in the same
    cells, like a training batch) must give the two-launch path's bits every time -- raw, the sigma hand-off and the
    operand-precision features -- in both 16-bit operand types."""
    from tests.test_gpu_step import _pair
    R = 2048
    for ns, nc in ((3, 2), (2, 3)):
        cfg, fld, orc, batch, rng = _pair(nof, precision, 0, ns, nc, R=R)
        qt = torch.float16 if precision == 'fp16x3' else torch.bfloat16
        Ns, Na = cfg['N_samples'], cfg['N_samples_around_depth']
        S = Ns + Na
        B = R * S
        u1, u2 = rng.random((R, Ns)).astype(np.float32), rng.random((R, Na)).astype(np.float32)
        fld.fused_forward = False
        b = fld.train_step(U.dev(batch), None, R, U.dev(u1), U.dev(u2), do_step=False)
        torch.cuda.synchronize()
        raw_ref, sig_ref = b['raw'].clone(), b['sig'].clone()
        want_q = b['feat'].permute(1, 0, 2).reshape(B, 32).to(qt)
        featq = torch.zeros(B, 32, dtype=torch.int16, device='cuda')
        raw = torch.zeros(B, 4, device='cuda')
        sig = torch.zeros(B, 16, dtype=torch.int16, device='cuda')
        for rep in range(10):
            raw.zero_()
            featq.zero_()
            nof.call('nof_encode_mlp_fwd', C.byref(fld.grid), C.byref(fld.desc), fld.packed, fld.table, b['pts_w'], b['view'], S, raw,
                     sig, featq, B)
            torch.cuda.synchronize()
            bad = int((raw != raw_ref).any(-1).sum())
            assert bad == 0, f'({ns},{nc}) run {rep}: {bad} of {B} samples differ from the two-launch path'
            assert torch.equal(sig, sig_ref) and torch.equal(featq.view(qt), want_q)
