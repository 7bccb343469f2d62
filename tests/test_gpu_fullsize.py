"""Parity at BASELINE.json's full sizes (cfg2: 4096 rays x 192 samples = 786 432 samples, L = 16, T = 2^19, base 16 -> 256)
through size-independent properties -- the oracle is far too slow there:

  * hash forward of a constant table is that constant (trilinear weights are a partition of unity), out-of-range -> 0;
  * hash backward conserves mass: per level and channel, sum of the table gradient == sum of dfeat over in-range samples
    (a checksum over 201 M atomic contributions, through the LDS-privatised, run-merged and row-de-duplicated paths);
  * input gradient of a table that is constant is zero;
  * every kernel is tile-independent: the MLP forward of the whole batch equals the concatenation of two halves bit for
    bit, and the split backward equals the fused backward;
  * ray marching: intervals sorted and disjoint, samples inside the traced span, stratified samples ordered;
  * a whole training step at cfg2 shapes is finite, decreases the loss and leaves flags == 0.
"""
import ctypes as C

import numpy as np
import pytest
import torch

from tests import util as U
from tests.test_gpu_ops import _mlp_setup, _pack

pytestmark = pytest.mark.gpu

R, S, L, T = 4096, 192, 16, 19
B = R * S


def _ray_like_points(seed=0, R=R):
    """samples along rays through the cube (consecutive samples of a ray are spatially coherent, like training batches),
    a few of them outside [-1,1]"""
    B = R * S
    g = torch.Generator(device='cuda').manual_seed(seed)
    o = torch.randn(R, 3, device='cuda', generator=g)
    o = o / o.norm(dim=1, keepdim=True) * 1.6
    tgt = (torch.rand(R, 3, device='cuda', generator=g) - 0.5) * 1.2
    d = tgt - o
    d = d / d.norm(dim=1, keepdim=True)
    t = torch.linspace(0.55, 2.4, S, device='cuda')[None, :, None] + torch.rand(R, S, 1, device='cuda', generator=g) * 0.004
    return (o[:, None, :] + t * d[:, None, :]).reshape(B, 3).contiguous()


@pytest.mark.parametrize("R,T,finest", [(4096, 19, 256),        # BASELINE cfg2 / cfg3 (per GPU)
                                        (8192, 19, 512),        # cfg4: 8192 rays per step, finest 512
                                        (16384, 22, 512)])      # cfg5: 16384 rays, T = 2^22 (237 MB table), finest 512
def test_hash_partition_of_unity_and_mass_conservation(nof, R, T, finest):
    B = R * S
    g, geo = U.make_grids(nof, L=L, T=T, finest=finest)
    pts = _ray_like_points(R=R)
    inside = ((pts >= -1) & (pts <= 1)).all(1)
    assert 0.3 < inside.float().mean().item() < 0.99
    table = torch.empty(geo.n_entries, 2, device='cuda')
    table[:, 0] = 0.375
    table[:, 1] = -1.25
    feat = torch.empty(L, B, 2, device='cuda')
    nof.call('nof_hash_encode_fwd', C.byref(g), pts, table, feat, B)
    f = feat[:, inside]
    assert (f[..., 0] - 0.375).abs().max().item() < 1e-6 and (f[..., 1] + 1.25).abs().max().item() < 2e-6
    assert (feat[:, ~inside] == 0).all()

    gen = torch.Generator(device='cuda').manual_seed(3)
    dfeat = torch.randn(L, B, 2, device='cuda', generator=gen)
    gtab = torch.zeros(geo.n_entries, 2, device='cuda')
    dpts = torch.full((B, 3), 9.0, device='cuda')
    nof.call('nof_hash_encode_bwd', C.byref(g), pts, table, dfeat, gtab, dpts, B)
    torch.cuda.synchronize()
    off = geo.offsets
    for l in range(L):
        got = gtab[off[l]:off[l + 1]].double().sum(0)
        ref = dfeat[l, inside].double().sum(0)
        scale = dfeat[l, inside].double().abs().sum(0)
        assert ((got - ref).abs() / scale).max().item() < 2e-6, (l, got, ref)
    # constant table: every corner difference is zero, so dL/dx is exactly zero (and untouched for nothing: overwritten)
    assert (dpts == 0).all()
    # the scatter is a pure function of its inputs up to summation order
    gtab2 = torch.zeros_like(gtab)
    nof.call('nof_hash_encode_bwd', C.byref(g), pts, table, dfeat, gtab2, None, B)
    assert (gtab2 - gtab).abs().max().item() < 1e-3 * gtab.abs().max().item()


@pytest.mark.parametrize("precision", [1, 2])
def test_mlp_tile_independence_and_split_equals_fused(nof, precision):
    ns, nc = 3, 2                                       # BASELINE cfg2 shape: SDF 3x64 + colour 2x64
    shape, params, desc, flat = _mlp_setup(nof, ns, nc, 0, L, precision, seed=4)
    packed = _pack(nof, desc, flat)
    gen = torch.Generator(device='cuda').manual_seed(5)
    feat = torch.randn(L, B, 2, device='cuda', generator=gen) * 0.3
    view = torch.zeros(R, 16, device='cuda')
    view[:, :9] = torch.randn(R, 9, device='cuda', generator=gen)
    raw = torch.empty(B, 4, device='cuda')
    sig = torch.zeros(B, 16, dtype=torch.int16, device='cuda')
    nof.call('nof_mlp_fwd', C.byref(desc), packed, feat, L, view, S, raw, sig, B)
    half = B // 2
    rawh = torch.empty(B, 4, device='cuda')
    for lo in (0, half):
        fh = feat[:, lo:lo + half].contiguous()
        nof.call('nof_mlp_fwd', C.byref(desc), packed, fh, L, view[lo // S:], S, rawh[lo:lo + half], None, half)
    assert torch.equal(raw, rawh) and torch.isfinite(raw).all()

    draw = torch.randn(B, 4, device='cuda', generator=gen)
    nblk = nof.load().nof_mlp_bwd_blocks()
    out = []
    for split in (False, True):
        dfeat = torch.empty(L, B, 2, device='cuda')
        dview = torch.zeros(R, 16, device='cuda')
        partials = torch.empty(nblk, desc.n_params, device='cuda')
        dsig = torch.zeros(B, 16, dtype=torch.int16, device='cuda')
        nof.call('nof_mlp_bwd', C.byref(desc), packed, feat, L, view, S, draw, sig if split else None, dsig if split else None,
                 dfeat, dview, partials, B)
        gflat = torch.zeros(desc.n_params, device='cuda')
        nof.call('nof_reduce_partials', partials, nblk, desc.n_params, gflat, None)
        out.append((dfeat, dview, gflat))
    torch.cuda.synchronize()
    (df0, dv0, g0), (df1, dv1, g1) = out
    assert torch.equal(df0, df1)                                       # same MFMA chains, same operand roundings
    assert (dv0 - dv1).abs().max().item() <= 1e-3 * dv0.abs().max().item()        # atomics: summation order only
    assert (g0 - g1).abs().max().item() <= 2e-4 * g0.abs().max().item()           # different tile -> wave assignment


def test_full_size_step_properties(nof):
    from bundlesdf_amd import synthetic
    from bundlesdf_amd.config import default_cfg
    from bundlesdf_amd.nerf_runner import NerfRunner
    pool = synthetic.make_pool(n_frames=6, H=480, W=640, fx=600.0, seed=0, analytic_bounds=True)
    cfg = default_cfg(n_step=1000, N_rand=R, num_levels=L, log2_hashmap_size=T, finest_res=256, base_res=16, far=1.0,
                      sc_factor=pool['sc_factor'], translation=pool['translation'])
    runner = NerfRunner(cfg, pool['rgbs'], depths=pool['depths'], masks=pool['masks'], normal_maps=None, poses=pool['poses'],
                        K=pool['K'], build_octree_pcd=synthetic.PointCloud(pool['pcd_normalized']), precision='bf16',
                        n_sigma=3, n_color=2)
    fld = runner.field
    runner.train_loop()
    first = fld.losses()
    b = fld._buffers(R, S)
    nh = b['n_hits'].long()
    tio = b['t_in_out']
    H = tio.shape[1]
    live = torch.arange(H, device='cuda')[None, :] < nh[:, None]
    tin, tout = tio[..., 0], tio[..., 1]
    assert (tout[live] > tin[live]).all() and (tin[live] > 0).all()
    nxt_ok = (tin[:, 1:] >= tout[:, :-1]) | ~live[:, 1:]
    assert nxt_ok.all()                                                # front to back, disjoint
    assert (tio[~live] == 0).all()                                     # zero padding (common.cu:158)
    z = b['z_vals']
    Ns = cfg['N_samples']
    hit = nh > 0
    zu = z[hit][:, :Ns]
    assert (zu[:, 1:] >= zu[:, :-1]).all()                             # stratified over the union of intervals: ordered
    assert (zu > 0).all() and (zu <= tout[hit].max(dim=1, keepdim=True).values + 1e-5).all()     # inside the traced span
    assert torch.isfinite(b['raw']).all() and torch.isfinite(fld.grads).all()
    for _ in range(60):
        runner.global_step += 1
        runner.train_loop()
    last = fld.losses()
    assert int(fld.flags[0].item()) == 0
    assert np.isfinite(last['loss']) and last['loss'] < 0.8 * first['loss'], (first, last)
    assert first['n_valid_samples'] > 0.2 * B


# ----------------------------------------------------------------------------------------------------------------------
# One WHOLE step at BASELINE.json's full sizes against the oracle: real rays of the synthetic keyframe pool, so the gradient
# scatter's run-merge / row de-duplication / LDS-privatised paths are value-checked on ray-coherent samples, not only by mass
# conservation.
#   cfg2  4096 rays x 192 samples, L = 16, T = 2^19, finest 256, SDF 3x64 + colour 2x64            (~20 s of oracle per case)
#   cfg4  8192 rays, finest 512 (the float32 resolution quirk of levels 12 / 15: 257 / 513), same network   (~45 s)
#   cfg5  T = 2^22 (237 MB table), finest 512, SDF 4x128 + colour 4x128, fp16 operands, all 16 384 rays of the configuration: the
#         oracle runs them as four chunks of 4096 (rays are independent; every loss term is a mean over the batch, so the
#         chunks' losses and gradients add up with weight R_chunk / R)                                     (~3 min of oracle)
CASES = {
    'cfg2': dict(R=4096, T=19, finest=256, ns=3, nc=2, hidden=64),
    'cfg4': dict(R=8192, T=19, finest=512, ns=3, nc=2, hidden=64),
    'cfg5': dict(R=16384, T=22, finest=512, ns=4, nc=4, hidden=128),         # ALL of the configuration's rays (oracle in 4 chunks)
    'cfg5_quarter': dict(R=4096, T=22, finest=512, ns=4, nc=4, hidden=128),
}
ORACLE_CHUNK = 4096          # rays per oracle call: rays are independent and every loss term is a mean over the batch


def _oracle_step_chunked(orc, batch, u_occ, u_dep):
    """OracleField.train_step(do_step=False) over the batch in chunks of ORACLE_CHUNK rays: per-ray results concatenated, losses and
    gradients combined with weight R_chunk / R (every term is a .mean() over R or R*S: nerf_runner.py:700, nerf_helpers.py:389-395)."""
    R = batch.shape[0]
    if R <= ORACLE_CHUNK:
        return orc.train_step(batch, u_occ, u_dep, do_step=False)
    parts = []
    for i in range(0, R, ORACLE_CHUNK):
        sl = slice(i, i + ORACLE_CHUNK)
        p = orc.train_step(batch[sl], u_occ[sl], u_dep[sl], do_step=False)
        w = batch[sl].shape[0] / R
        parts.append(dict(
            z_vals=p['z_vals'], trace={k: p['trace'][k] for k in ('n_hits', 'cell_ids', 'rays_o_w', 'viewdirs_w')},
            fwd={k: p['fwd'][k].detach() for k in ('raw', 'valid_samples')},
            losses={k: float(v) * w for k, v in p['losses'].items() if k in ('loss', 'rgb_loss', 'fs_loss', 'sdf_loss')},
            grads=[None if g is None else g * w for g in p['grads']]))
        del p
    Hc = max(q['trace']['cell_ids'].shape[1] for q in parts)
    pad = lambda c: np.pad(c, ((0, 0), (0, Hc - c.shape[1])), constant_values=-1)
    return dict(
        z_vals=torch.cat([q['z_vals'] for q in parts], 0),
        trace=dict(n_hits=np.concatenate([q['trace']['n_hits'] for q in parts]),
                   cell_ids=np.concatenate([pad(q['trace']['cell_ids']) for q in parts], 0),
                   rays_o_w=torch.cat([q['trace']['rays_o_w'] for q in parts], 0),
                   viewdirs_w=torch.cat([q['trace']['viewdirs_w'] for q in parts], 0)),
        fwd={k: torch.cat([q['fwd'][k] for q in parts], 0) for k in ('raw', 'valid_samples')},
        losses={k: sum(q['losses'][k] for q in parts) for k in parts[0]['losses']},
        grads=[None if g[0] is None else sum(g) for g in zip(*[q['grads'] for q in parts])])


@pytest.mark.parametrize("case,precision", [('cfg2', 'fp32'), ('cfg2', 'fp16x3'), ('cfg2', 'bf16x3'), ('cfg4', 'fp16x3'),
                                            ('cfg5', 'fp16'), ('cfg5_quarter', 'fp16x3')])
def test_fullsize_step_matches_oracle(nof, case, precision):
    from bundlesdf_amd import synthetic
    from bundlesdf_amd.config import default_cfg
    from bundlesdf_amd.nerf_runner import NerfRunner
    from oracle import nof_oracle as O
    from tests.test_gpu_ops import rel_l2, rel_max, worst_elementwise, ODT
    from bundlesdf_amd.field import PRECISIONS
    c = CASES[case]
    R, T, ns, nc, hidden = c['R'], c['T'], c['ns'], c['nc'], c['hidden']
    pool = synthetic.make_pool(n_frames=6, H=480, W=640, fx=600.0, seed=0, analytic_bounds=True)
    cfg = default_cfg(n_step=1000, N_rand=R, num_levels=L, log2_hashmap_size=T, finest_res=c['finest'], base_res=16, far=1.0,
                      sc_factor=pool['sc_factor'], translation=pool['translation'])
    runner = NerfRunner(cfg, pool['rgbs'], depths=pool['depths'], masks=pool['masks'], normal_maps=None, poses=pool['poses'],
                        K=pool['K'], build_octree_pcd=synthetic.PointCloud(pool['pcd_normalized']), precision=precision,
                        n_sigma=ns, n_color=nc, hidden=hidden)
    fld = runner.field
    F = fld.F
    rng = np.random.default_rng(11)
    table0 = (rng.uniform(-1, 1, size=(fld.n_entries, 2)) * 0.05).astype(np.float32)
    pose0 = (rng.normal(size=(F, 6)) * 0.1).astype(np.float32)
    fld.load_parameters(table=table0, pose=pose0)
    ids = runner.data_loader.next_ids()
    batch = runner.rays[ids].cpu().numpy()
    Ns, Na = cfg['N_samples'], cfg['N_samples_around_depth']
    u_occ = rng.random((R, Ns)).astype(np.float32)
    u_dep = rng.random((R, Na)).astype(np.float32)
    b = fld.train_step(runner.rays, ids, R, U.dev(u_occ), U.dev(u_dep), do_step=False, want_cells=True)
    torch.cuda.synchronize()
    assert int(fld.flags[0].item()) == 0

    occ, occ_l, max_level, level = O.build_occupancy(pool['pcd_normalized'], cfg)
    assert level == fld.level
    geo = O.HashGeometry(L, 2, cfg['base_res'], T, cfg['finest_res'])
    shape = O.FieldShape(input_ch=2 * L, input_ch_views=9, num_layers=ns, num_layers_color=nc, hidden_dim=hidden,
                         hidden_dim_color=hidden)
    mlp = [[W.clone(), bb.clone()] for W, bb in fld.mlp_state()]
    # The yardstick of the OUTPUTS is the pure fp32 oracle whenever the forward claims fp32-class results (fp32, the split
    # precisions); a plain 16-bit forward is compared with the oracle that rounds its operands the same way (the reference's own
    # autocast path deviates from its fp32 path by 1e-4 ... 4e-3, tests/test_ref_native.py) and its distance to fp32 is printed.
    plain16 = precision in ('fp16', 'bf16')
    orc = O.OracleField(cfg, geo, shape, F, pool['poses'], occ_l, table=table0, mlp=mlp, pose=pose0,
                        operand_dtype=ODT[PRECISIONS[precision]] if plain16 else None)
    cpu = lambda t: t.detach().cpu().numpy()
    # The index work is discontinuous in the ray, so both programs must start from the same ray BITS.  What the oracle is handed is
    # the product's pose table -- F x 12 floats, tf = Delta(pose) c2w, asserted against the oracle's own (pytorch3d-style se3 exp in
    # torch: libm's sin / cos / tanh differ from the device's in the last bit) -- and nothing per ray: the oracle composes the world
    # rays itself, in the product's documented float32 order (OracleField.pose_table), and the rays' bits are then ASSERTED equal
    # (round 5 allowed 0.1 % of the rays to differ with torch's batched matmul order and checked strictly only on the device's rays).
    tf_dev = cpu(fld.tf).reshape(F, 3, 4)
    with torch.no_grad():
        tf_orc = orc.frame_tf()[:, :3, :4].numpy()
    assert np.abs(tf_dev - tf_orc).max() < 2e-6, np.abs(tf_dev - tf_orc).max()
    orc.pose_table = tf_dev
    ref = _oracle_step_chunked(orc, batch, u_occ, u_dep)

    # ---- index work: every ray's hit count and cell list, bit for bit (north_star: "bit-identical occupancy/ray-hit indices") ----
    ro, vd = cpu(b['rays_o_w']), cpu(b['viewdirs_w'])
    assert np.array_equal(ro.view(np.uint32), ref['trace']['rays_o_w'].numpy().view(np.uint32))
    assert np.array_equal(vd.view(np.uint32), ref['trace']['viewdirs_w'].numpy().view(np.uint32))
    nh_ref, cid_ref = ref['trace']['n_hits'], ref['trace']['cell_ids']
    H = cid_ref.shape[1]
    nh, cid = cpu(b['n_hits']), cpu(b['cell_ids'])[:, :H]
    same = (nh == nh_ref) & (cid == cid_ref).all(axis=1)
    assert same.all(), (int((~same).sum()), R)
    z, z_ref = cpu(b['z_vals']), ref['z_vals'].numpy()
    assert np.array_equal(z.view(np.uint32), z_ref.view(np.uint32)), float(np.abs(z - z_ref).max())
    # ---- the same once more with the tracer and the sampler called directly on the device's rays: interval bits too, and the
    #      padding (Utils.py:443-475, common.cu:41-167, nerf_runner.py:67-87,979-1011).
    tio_s, cid_s, nh_s = O.trace_rays(occ_l, ro, vd)
    Hs = cid_s.shape[1]
    assert np.array_equal(nh, nh_s)
    full_cid = cpu(b['cell_ids'])
    assert np.array_equal(full_cid[:, :Hs], cid_s) and (full_cid[:, Hs:] == -1).all()
    assert np.array_equal(cpu(b['t_in_out'])[:, :Hs].view(np.uint32), tio_s.view(np.uint32))
    # (the camera-frame unit direction's z, which scales the intervals: float32 throughout like the device -- and like the reference
    # on its GPU; torch's CPU norm accumulates the squares in float64, which parts a few rays' z by an ulp)
    dcam = batch[:, 0:3].astype(np.float32)
    nrm = np.sqrt((dcam[:, 0] * dcam[:, 0] + dcam[:, 1] * dcam[:, 1]) + dcam[:, 2] * dcam[:, 2]).astype(np.float32)
    vz = (dcam[:, 2] / nrm).astype(np.float32)
    z_s = np.asarray(O.sample_z(tio_s, vz, batch[:, 6], cfg, O.get_truncation(cfg, 0), u_occ, u_dep), np.float32)
    z_bad = z.view(np.uint32) != z_s.view(np.uint32)
    assert not z_bad.any(), (int(z_bad.sum()), int(z_bad.any(1).sum()), float(np.abs(z - z_s).max()))
    print(f'fullsize {case} {precision}: world rays composed by the oracle from the pose table (max |tf - oracle tf| '
          f'{np.abs(tf_dev - tf_orc).max():.1e}): ray bits, n_hits, cell ids, interval bits and z bits all identical ({R} rays, {int(nh.sum())} hits)')
    # ---- outputs: north_star's bar, SDF / colour within 1e-3 (max-norm) on the samples both sides call valid ----
    v_ref = ref['fwd']['valid_samples'].numpy()
    v_got = cpu(b['valid']).reshape(R, S).astype(bool)
    assert (v_got != v_ref).mean() < 1e-3
    both = v_got & v_ref & same[:, None]
    assert both.mean() > 0.3
    raw_ref = ref['fwd']['raw'].detach().numpy()
    raw = cpu(b['raw']).reshape(R, S, 4)
    err_rgb, err_sdf = rel_max(raw[both][:, :3], raw_ref[both][:, :3]), rel_max(raw[both][:, 3], raw_ref[both][:, 3])
    # ... and per element: |err| <= 1e-3 |ref| + 1e-5 for EVERY colour / SDF value (a plain 16-bit forward is measured against
    # the oracle with the same operand rounding: what it can be held to per element is that rounding's own noise floor)
    ew_rgb, ew_sdf = worst_elementwise(raw[both][:, :3], raw_ref[both][:, :3]), worst_elementwise(raw[both][:, 3], raw_ref[both][:, 3])
    print(f'fullsize {case} {precision}: colour rel-max {err_rgb:.2e}, sdf rel-max {err_sdf:.2e}, valid fraction {both.mean():.3f}; '
          f'per element (|err| / (1e-3 |ref| + 1e-5)) colour {ew_rgb:.3f}, sdf {ew_sdf:.3f}')
    assert err_rgb < 1e-3 and err_sdf < 1e-3
    if not plain16 and not fld.wide:                 # (the wide path has no operand split: its 'x3' forward is a plain 16-bit one)
        assert ew_rgb <= 1.0 and ew_sdf <= 1.0, (ew_rgb, ew_sdf)
    if plain16:
        with torch.no_grad():
            orc32 = O.OracleField(cfg, geo, shape, F, pool['poses'], occ_l, table=table0, mlp=mlp, pose=pose0)
            raw32 = np.concatenate([orc32.forward(torch.from_numpy(batch[i:i + ORACLE_CHUNK]), ref['z_vals'][i:i + ORACLE_CHUNK])['raw'].numpy()
                                    for i in range(0, R, ORACLE_CHUNK)], 0)
        e32_rgb, e32_sdf = rel_max(raw[both][:, :3], raw32[both][:, :3]), rel_max(raw[both][:, 3], raw32[both][:, 3])
        print(f'fullsize {case} {precision}: vs the PURE fp32 oracle colour {e32_rgb:.2e}, sdf {e32_sdf:.2e}')
        if fld.wide:          # north_star's 1e-3 (max-norm) holds against pure fp32 too on the wide path, without an operand split
            assert e32_rgb < 1e-3 and e32_sdf < 1e-3, (e32_rgb, e32_sdf)
    Lo = fld.losses()
    for k in ('loss', 'rgb_loss', 'fs_loss', 'sdf_loss'):
        r = float(ref['losses'][k])
        assert abs(Lo[k] - r) <= 1e-3 * abs(r) + 1e-7, (k, Lo[k], r)
    # ---- gradients: table (201 M scattered contributions at cfg2), MLP layers, poses.  fp32: against the oracle as is; 16-bit
    #      backward (the reference's autocast): looser, the backward rounds its operands to the 16-bit type ----
    tight = precision == 'fp32'
    bf = 2.0 if precision.startswith('bf16') else 1.0             # bfloat16 backward: 8 mantissa bits
    names = ['table'] + [f'mlp{i}' for i in range(2 * (ns + nc))] + ['pose']
    g_ref = dict(zip(names, ref['grads']))
    gt = cpu(fld._seg(fld.grads, 'table')).reshape(-1, 2)
    gt_ref = g_ref['table'].numpy()
    e_tab = rel_l2(gt, gt_ref)
    print(f'fullsize {case} {precision}: table-gradient rel-L2 {e_tab:.2e}, touched rows {int((gt_ref != 0).any(1).sum())}')
    assert e_tab < (1e-3 if tight else 3e-2 * bf)
    for lvl in range(L):                                           # per level, so that a coarse level cannot hide a fine one
        lo, hi = int(fld.offsets[lvl]), int(fld.offsets[lvl + 1])
        assert rel_l2(gt[lo:hi], gt_ref[lo:hi]) < (2e-3 if tight else 5e-2 * bf), lvl
    gm = cpu(fld._seg(fld.grads, 'mlp'))
    gm_ref = torch.cat([g.reshape(-1) for n, g in g_ref.items() if n.startswith('mlp')]).numpy()
    for l in range(ns + nc):
        lo, hi = fld.desc.w_off[l], fld.desc.b_off[l] + fld.desc.out_dim[l]
        assert rel_l2(gm[lo:hi], gm_ref[lo:hi]) < (2e-3 if tight else 5e-2 * bf), (l, rel_l2(gm[lo:hi], gm_ref[lo:hi]))
    gp = cpu(fld._seg(fld.grads, 'pose')).reshape(-1, 6)
    assert rel_l2(gp, g_ref['pose'].numpy()) < (1e-2 if tight else 5e-2 * bf), rel_l2(gp, g_ref['pose'].numpy())
