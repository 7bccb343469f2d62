"""The work list of the backward (NofTileList: north_star's per-wavefront compaction).

  * the list itself -- count, ascending tile ids, flags -- is exactly what NumPy derives from dL/draw, through both builders
    (the loss kernel's fused flags for S % 32 == 0, the extra pass otherwise) and the `all` form;
  * every backward entry point computes the same result over the list as over the whole batch: the sums only lose terms that
    are exactly zero.  (Whole steps against the oracle run over the list by default: tests/test_gpu_step.py, test_gpu_fullsize.py.)
"""
import ctypes as C

import numpy as np
import pytest
import torch

from tests import util as U
from tests.test_gpu_ops import _mlp_setup, _pack, _scene

pytestmark = pytest.mark.gpu


def cpu(t):
    return t.detach().cpu().numpy()


def _parse(nof, tl, B):
    nt = (B + 31) // 32
    words = 4 + ((nt + 1 + 3) & ~3)
    raw = cpu(tl)
    head = raw[:16].view(np.uint32)
    tiles = raw[16:16 + 4 * (nt + 1)].view(np.uint32)
    flags = raw[4 * words:4 * words + nt]
    assert int(nof.load().nof_tile_list_bytes(B)) == raw.size == 4 * words + ((nt + 15) & ~15)
    return int(head[0]), int(head[1]), tiles, flags


def _sparse_draw(R, S, seed, p_ray=0.5, p_front=0.4):
    """dL/draw with the structure of a training batch: whole rays without a loss term, the first part of the others zero"""
    rng = np.random.default_rng(seed)
    draw = rng.normal(size=(R, S, 4)).astype(np.float32) * 1e-3
    draw[rng.random(R) < p_ray] = 0
    for r in range(R):
        draw[r, :int(S * p_front * rng.random())] = 0
    draw[rng.random((R, S)) < 0.05] = 0                       # scattered single zeros inside live tiles
    return draw.reshape(R * S, 4)


@pytest.mark.parametrize("R,S", [(64, 192), (33, 96), (37, 40), (5, 33), (1, 32)])
def test_tile_list_matches_numpy(nof, R, S):
    B = R * S
    draw = _sparse_draw(R, S, seed=R)
    nt = (B + 31) // 32
    pad = np.zeros((nt * 32, 4), np.float32)
    pad[:B] = draw
    want = (pad.reshape(nt, 32 * 4) != 0).any(1)
    tl = torch.zeros(int(nof.load().nof_tile_list_bytes(B)), dtype=torch.uint8, device='cuda')
    tl.fill_(0xAB)                                             # nothing may depend on the buffer's previous content
    nof.call('nof_tile_list_build', U.dev(draw), B, 0, tl)
    torch.cuda.synchronize()
    n, ntl, tiles, flags = _parse(nof, tl, B)
    assert ntl == nt and n == int(want.sum())
    assert np.array_equal(tiles[:n], np.nonzero(want)[0]) and np.array_equal(flags.astype(bool), want)
    if n & 1:
        assert tiles[n] == nt                                  # an odd list ends in a tile that does not exist
    nof.call('nof_tile_list_build', None, B, 1, tl)
    torch.cuda.synchronize()
    n, ntl, tiles, flags = _parse(nof, tl, B)
    assert n == nt and np.array_equal(tiles[:n], np.arange(nt)) and flags.all()


@pytest.mark.parametrize("S", [96, 64, 40])
def test_loss_kernel_emits_the_list_of_its_own_draw(nof, S):
    """nof_composite_loss_fwd_bwd: the list equals the one built from the dL/draw it wrote (fused flags when S % 32 == 0)."""
    cfg, occ, c2w, batch = _scene(nof, R=300)
    from bundlesdf_amd import lib
    R = batch.shape[0]
    B = R * S
    rng = np.random.default_rng(3)
    raw = rng.normal(size=(B, 4)).astype(np.float32)
    z = np.sort(rng.uniform(2.0, 3.6, size=(R, S)).astype(np.float32), axis=1)
    valid = (rng.random(B) > 0.1).astype(np.uint8)
    sc = cfg['sc_factor']
    lc = lib.NofLossCfg(cfg['trunc'] * sc, cfg['neg_trunc_ratio'], cfg['sdf_lambda'], cfg['near'] * sc, cfg['far'] * sc,
                        cfg['rgb_weight'], cfg['fs_weight'], cfg['trunc_weight'], cfg['empty_weight'], cfg['fs_sdf'], 0.0,
                        cfg['first_frame_weight'], 1.0)
    rgb, draw, rows = torch.empty(R, 3, device='cuda'), torch.empty(B, 4, device='cuda'), torch.empty(R, 8, device='cuda')
    loss = torch.zeros(8, device='cuda')
    tl = torch.full((int(nof.load().nof_tile_list_bytes(B)),), 0xCD, dtype=torch.uint8, device='cuda')
    nof.call('nof_composite_loss_fwd_bwd', C.byref(lc), U.dev(raw), U.dev(z), U.dev(valid), U.dev(batch), R, S, rgb, None, draw,
             rows, loss, tl)
    torch.cuda.synchronize()
    d = cpu(draw)
    nt = (B + 31) // 32
    pad = np.zeros((nt * 32, 4), np.float32)
    pad[:B] = d
    want = (pad.reshape(nt, 128) != 0).any(1)
    assert 0.05 < want.mean() < 0.95                            # the scene has rays with and without loss terms
    n, ntl, tiles, flags = _parse(nof, tl, B)
    assert n == int(want.sum()) and np.array_equal(tiles[:n], np.nonzero(want)[0]) and np.array_equal(flags.astype(bool), want)
    # and the loss terms did not move: same call without a list
    loss2 = torch.zeros(8, device='cuda')
    draw2 = torch.empty_like(draw)
    nof.call('nof_composite_loss', C.byref(lc), U.dev(raw), U.dev(z), U.dev(valid), U.dev(batch), R, S, rgb, None, draw2, rows, loss2)
    torch.cuda.synchronize()
    assert torch.equal(draw, draw2) and torch.equal(loss, loss2)


@pytest.mark.parametrize("R,S", [(2100, 192), (4099, 192), (16384, 192), (20001, 40), (9000, 96)])
def test_loss_kernel_list_over_several_workgroups(nof, R, S):
    """round 6: the scan of the flags is shared by one workgroup per ~6000 tiles (2 / 4 / 16 / 4 / 4 here; each counts the listed
    tiles in front of its range by itself): the list is the ascending list of the tiles with a non-zero row of the dL/draw the
    call wrote, the head holds their number, an odd list ends in the sentinel -- sizes with a partial last word and a partial last
    16-byte group, fused flags (S % 32 == 0) and the separate flag pass (S = 40)."""
    cfg, occ, c2w, batch0 = _scene(nof, R=300)
    from bundlesdf_amd import lib
    batch = np.ascontiguousarray(np.tile(batch0, ((R + batch0.shape[0] - 1) // batch0.shape[0], 1))[:R])
    B = R * S
    rng = np.random.default_rng(R)
    raw = rng.normal(size=(B, 4)).astype(np.float32)
    z = np.sort(rng.uniform(2.0, 3.6, size=(R, S)).astype(np.float32), axis=1)
    valid = (rng.random(B) > 0.1).astype(np.uint8)
    sc = cfg['sc_factor']
    lc = lib.NofLossCfg(cfg['trunc'] * sc, cfg['neg_trunc_ratio'], cfg['sdf_lambda'], cfg['near'] * sc, cfg['far'] * sc,
                        cfg['rgb_weight'], cfg['fs_weight'], cfg['trunc_weight'], cfg['empty_weight'], cfg['fs_sdf'], 0.0,
                        cfg['first_frame_weight'], 1.0)
    rgb, draw, rows = torch.empty(R, 3, device='cuda'), torch.empty(B, 4, device='cuda'), torch.empty(R, 8, device='cuda')
    loss = torch.zeros(8, device='cuda')
    tl = torch.full((int(nof.load().nof_tile_list_bytes(B)),), 0xCD, dtype=torch.uint8, device='cuda')
    nof.call('nof_composite_loss_fwd_bwd', C.byref(lc), U.dev(raw), U.dev(z), U.dev(valid), U.dev(batch), R, S, rgb, None, draw,
             rows, loss, tl)
    torch.cuda.synchronize()
    nt = (B + 31) // 32
    assert nt // 6144 >= 2
    pad = np.zeros((nt * 32, 4), np.float32)
    pad[:B] = cpu(draw)
    want = (pad.reshape(nt, 128) != 0).any(1)
    assert 0.05 < want.mean() < 0.95
    n, ntl, tiles, flags = _parse(nof, tl, B)
    assert ntl == nt and n == int(want.sum())
    assert np.array_equal(tiles[:n], np.nonzero(want)[0]) and np.array_equal(flags.astype(bool), want)
    if n & 1:
        assert tiles[n] == nt
    # the loss terms: the row sum of the same call without a list
    loss2 = torch.zeros(8, device='cuda')
    nof.call('nof_composite_loss', C.byref(lc), U.dev(raw), U.dev(z), U.dev(valid), U.dev(batch), R, S, rgb, None, draw, rows, loss2)
    torch.cuda.synchronize()
    assert torch.equal(loss, loss2)


@pytest.mark.parametrize("ns,nc,precision", [(3, 2, 3), (2, 3, 2), (2, 3, 0), (3, 3, 1)])
@pytest.mark.parametrize("R,S", [(40, 192), (21, 48)])
def test_mlp_backward_over_the_list_equals_whole_batch(nof, ns, nc, precision, R, S):
    L, ff = 16, 2
    shape, params, desc, flat = _mlp_setup(nof, ns, nc, ff, L, precision, seed=4)
    B = R * S
    torch.manual_seed(8)
    feat = (torch.randn(L, B, 2) * 0.5).cuda()
    view = torch.zeros(R, 16)
    view[:, :9 + ff] = torch.randn(R, 9 + ff)
    view = view.cuda()
    draw = U.dev(_sparse_draw(R, S, seed=S + ns))
    packed = _pack(nof, desc, flat)
    split = precision != 0
    sig = torch.zeros(B, 16, dtype=torch.int16, device='cuda') if split else None
    raw = torch.zeros(B, 4, device='cuda')
    nof.call('nof_mlp_fwd', C.byref(desc), packed, feat, L, view, S, raw, sig, B)
    nblk = nof.load().nof_mlp_bwd_blocks()
    tl = torch.zeros(int(nof.load().nof_tile_list_bytes(B)), dtype=torch.uint8, device='cuda')
    nof.call('nof_tile_list_build', draw, B, 0, tl)
    out = {}
    for mode in ('whole', 'list', 'all'):
        dsig = torch.zeros(B, 16, dtype=torch.int16, device='cuda') if split else None
        dfeat = torch.full((L, B, 2), 3.0, device='cuda')
        dview = torch.zeros(R, 16, device='cuda')
        partials = torch.full((nblk, desc.n_params), 5.0, device='cuda')
        if mode == 'all':
            nof.call('nof_tile_list_build', None, B, 1, tl)
        nof.call('nof_mlp_bwd_tiles', C.byref(desc), packed, feat, L, view, S, draw, sig, dsig, dfeat, dview, partials,
                 None if mode == 'whole' else tl, B)
        g = torch.zeros(desc.n_params, device='cuda')
        nof.call('nof_reduce_partials', partials, nblk, desc.n_params, g, None)
        torch.cuda.synchronize()
        out[mode] = (cpu(dfeat), cpu(dview), cpu(g))
    nt = (B + 31) // 32
    pad = np.zeros((nt * 32, 4), np.float32)
    pad[:B] = cpu(draw)
    live = np.repeat((pad.reshape(nt, 128) != 0).any(1), 32)[:B]
    assert 0.1 < live.mean() < 0.9
    df_w, dv_w, g_w = out['whole']
    assert (df_w[:, ~live] == 0).all()                          # whole batch: zero tiles are written as zeros
    for mode in ('list', 'all'):
        df, dv, g = out[mode]
        sel = live if mode == 'list' else np.ones(B, bool)
        assert np.array_equal(df[:, sel], df_w[:, sel]), mode   # per-sample results do not depend on which wave computed them
        if mode == 'list':
            assert (df[:, ~live] == 3.0).all()                  # unlisted tiles are not touched
        # the weight gradient and dview are sums over tiles in a different order: equal up to fp32 summation order
        assert np.abs(g - g_w).max() <= 2e-5 * np.abs(g_w).max(), mode
        assert np.abs(dv - dv_w).max() <= 2e-5 * max(np.abs(dv_w).max(), 1e-30), mode


@pytest.mark.parametrize("ns,nc,hidden,precision", [(4, 4, 128, 2), (2, 3, 128, 1), (4, 4, 64, 2)])
def test_wide_backward_over_the_list_equals_whole_batch(nof, ns, nc, hidden, precision):
    from oracle import nof_oracle as O
    L, ff, R, S = 16, 2, 40, 192
    B = R * S
    torch.manual_seed(9)
    shape = O.FieldShape(input_ch=2 * L, input_ch_views=9 + ff, num_layers=ns, hidden_dim=hidden, num_layers_color=nc,
                         hidden_dim_color=hidden)
    params = O.init_mlp_params(shape)
    desc, dims = nof.make_mlp_desc(ns, nc, 2 * L, 9 + ff, precision, hidden=hidden)
    desc.grad_scale = 1024.0 if precision == 2 else 0.0
    flat = torch.cat([torch.cat([W.reshape(-1), b.reshape(-1)]) for W, b in params])
    feat = (torch.randn(L, B, 2) * 0.5).cuda()
    view = torch.zeros(R, 16)
    view[:, :9 + ff] = torch.randn(R, 9 + ff)
    view = view.cuda()
    draw = U.dev(_sparse_draw(R, S, seed=ns + hidden))
    packed = _pack(nof, desc, flat)
    ws = torch.zeros(int(nof.load().nof_mlp_wide_workspace_bytes(C.byref(desc), B)), dtype=torch.uint8, device='cuda')
    raw = torch.zeros(B, 4, device='cuda')
    nof.call('nof_mlp_wide_fwd', C.byref(desc), packed, feat, L, view, S, raw, ws, B)
    rows = nof.load().nof_mlp_wide_partial_rows()
    tl = torch.zeros(int(nof.load().nof_tile_list_bytes(B)), dtype=torch.uint8, device='cuda')
    nof.call('nof_tile_list_build', draw, B, 0, tl)
    out = {}
    for mode in ('whole', 'list'):
        dfeat = torch.full((L, B, 2), 3.0, device='cuda')
        dview = torch.zeros(R, 16, device='cuda')
        partials = torch.full((rows, desc.n_params), 5.0, device='cuda')
        nof.call('nof_mlp_wide_bwd_tiles', C.byref(desc), packed, feat, L, view, S, draw, ws, dfeat, dview, partials,
                 None if mode == 'whole' else tl, B)
        g = torch.zeros(desc.n_params, device='cuda')
        nof.call('nof_reduce_partials', partials, rows, desc.n_params, g, None)
        torch.cuda.synchronize()
        out[mode] = (cpu(dfeat), cpu(dview), cpu(g))
    nt = (B + 31) // 32
    pad = np.zeros((nt * 32, 4), np.float32)
    pad[:B] = cpu(draw)
    live = np.repeat((pad.reshape(nt, 128) != 0).any(1), 32)[:B]
    (df_w, dv_w, g_w), (df, dv, g) = out['whole'], out['list']
    assert (df_w[:, ~live] == 0).all() and (df[:, ~live] == 3.0).all()
    assert np.array_equal(df[:, live], df_w[:, live])
    assert np.abs(g - g_w).max() <= 2e-5 * np.abs(g_w).max()
    assert np.abs(dv - dv_w).max() <= 2e-5 * max(np.abs(dv_w).max(), 1e-30)


@pytest.mark.parametrize("T,finest,R,S", [(19, 256, 512, 192), (14, 128, 33, 96), (22, 512, 256, 192)])
def test_hash_backward_over_the_list_equals_whole_batch(nof, T, finest, R, S):
    from tests.test_gpu_fullsize import _ray_like_points
    L = 16
    B = R * S
    g, geo = U.make_grids(nof, L=L, T=T, finest=finest)
    gen = torch.Generator(device='cuda').manual_seed(5)
    o = torch.randn(R, 3, device='cuda', generator=gen)
    o = o / o.norm(dim=1, keepdim=True) * 1.6
    tgt = (torch.rand(R, 3, device='cuda', generator=gen) - 0.5) * 1.2
    d = tgt - o
    d = d / d.norm(dim=1, keepdim=True)
    t = torch.linspace(0.55, 2.4, S, device='cuda')[None, :, None] + torch.rand(R, S, 1, device='cuda', generator=gen) * 0.004
    pts = (o[:, None, :] + t * d[:, None, :]).reshape(B, 3).contiguous()
    table = (torch.rand(geo.n_entries, 2, device='cuda', generator=gen) - 0.5) * 0.2
    draw = _sparse_draw(R, S, seed=T)
    nt = (B + 31) // 32
    pad = np.zeros((nt * 32, 4), np.float32)
    pad[:B] = draw
    live = torch.from_numpy(np.repeat((pad.reshape(nt, 128) != 0).any(1), 32)[:B]).cuda()
    dfeat = torch.randn(L, B, 2, device='cuda', generator=gen)
    dfeat_garbage = dfeat.clone()
    dfeat_garbage[:, ~live] = float('nan')                      # what the list path must never read
    dfeat[:, ~live] = 0
    tl = torch.zeros(int(nof.load().nof_tile_list_bytes(B)), dtype=torch.uint8, device='cuda')
    nof.call('nof_tile_list_build', U.dev(draw), B, 0, tl)
    g_w = torch.zeros(geo.n_entries, 2, device='cuda')
    dp_w = torch.full((B, 3), 7.0, device='cuda')
    nof.call('nof_hash_encode_bwd', C.byref(g), pts, table, dfeat, g_w, dp_w, B)
    g_l = torch.zeros(geo.n_entries, 2, device='cuda')
    dp_l = torch.full((B, 3), 7.0, device='cuda')
    # the three kernels as the training step launches them: the big levels on one stream, the rest beside them
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        nof.call('nof_hash_encode_bwd_parts', C.byref(g), pts, table, dfeat_garbage, None, None, g_l, dp_l, 0, L, tl,
                 nof.HASH_BWD_INPUT | nof.HASH_BWD_TABLE_SMALL, 0, B)
    nof.call('nof_hash_encode_bwd_parts', C.byref(g), pts, table, dfeat_garbage, None, None, g_l, dp_l, 0, L, tl,
             nof.HASH_BWD_TABLE_BIG, 0, B)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    assert torch.isfinite(g_l).all() and torch.isfinite(dp_l).all()
    assert torch.equal(dp_l, dp_w)                              # per-sample gathers: bit-equal, zeros on the unlisted tiles
    scale = g_w.abs().max().item()
    assert (g_l - g_w).abs().max().item() <= 2e-5 * scale       # atomics in a different order
    off = geo.offsets
    for l in range(L):                                          # per level: mass conservation of the listed path
        got = g_l[off[l]:off[l + 1]].double().sum(0)
        want = g_w[off[l]:off[l + 1]].double().sum(0)
        assert ((got - want).abs() / g_w[off[l]:off[l + 1]].double().abs().sum(0).clamp_min(1e-30)).max().item() < 1e-6, l
    # the large levels' scatter and dL/dx as two roles of ONE launch (round 6, NOF_HASH_BWD_MERGE_INPUT), with and without a list:
    # dL/dx bit-equal, the table gradient up to the atomics' order
    for lst, df_in in ((tl, dfeat_garbage), (None, dfeat)):
        g_m = torch.zeros(geo.n_entries, 2, device='cuda')
        dp_m = torch.full((B, 3), 7.0, device='cuda')
        nof.call('nof_hash_encode_bwd_parts', C.byref(g), pts, table, df_in, None, None, g_m, dp_m, 0, L, lst,
                 nof.HASH_BWD_ALL | nof.HASH_BWD_MERGE_INPUT, 0, B)
        torch.cuda.synchronize()
        assert torch.equal(dp_m, dp_w)
        assert (g_m - g_w).abs().max().item() <= 2e-5 * scale
    # the table levels with the MLP backward's row reduction riding in the LDS levels' launch (what the step calls): the same table
    # gradient, and the reduction's own result bit for bit (fixed summation order per column and row range; two atomics per column
    # onto zeros commute) -- incl. the overflow flag for a non-finite column
    rows, cols = 512, 10755
    partials = torch.randn(rows, cols, device='cuda', generator=gen)
    partials[7, 123] = float('inf')
    want_mlp = torch.zeros(cols, device='cuda')
    want_flags = torch.zeros(4, dtype=torch.int32, device='cuda')
    nof.call('nof_reduce_partials', partials, rows, cols, want_mlp, want_flags)
    for parts in (nof.HASH_BWD_TABLE_BIG | nof.HASH_BWD_TABLE_SMALL, nof.HASH_BWD_TABLE_BIG):     # (the second: no launch to ride in)
        g_m = torch.zeros(geo.n_entries, 2, device='cuda')
        got_mlp = torch.zeros(cols, device='cuda')
        got_flags = torch.zeros(4, dtype=torch.int32, device='cuda')
        nof.call('nof_hash_encode_bwd_parts_reduce', C.byref(g), pts, table, dfeat_garbage, None, None, g_m, None, 0, L, tl, parts, 0, B,
                 partials, rows, cols, got_mlp, got_flags)
        torch.cuda.synchronize()
        assert torch.equal(got_mlp.view(torch.int32), want_mlp.view(torch.int32)) and int(got_flags[0]) == int(want_flags[0]) == 4
        if parts & nof.HASH_BWD_TABLE_SMALL:
            assert (g_m - g_w).abs().max().item() <= 2e-5 * scale


def test_step_over_the_list_equals_step_without(nof):
    """whole optimisation steps: backward_tiles 'list' (default), 'all' and 'off' give the same gradients and the same parameters
    after several Adam steps (up to summation order)."""
    from tests.test_gpu_step import _pair
    res = {}
    for mode in ('off', 'list', 'all'):
        cfg, fld, orc, batch, rng = _pair(nof, 'fp16x3', ff=2, ns=3, nc=2, R=256)
        fld.backward_tiles = mode
        R = batch.shape[0]
        pool = U.dev(batch)
        for it in range(4):
            u1 = rng.random((R, cfg['N_samples'])).astype(np.float32)
            u2 = rng.random((R, cfg['N_samples_around_depth'])).astype(np.float32)
            fld.train_step(pool, None, R, U.dev(u1), U.dev(u2), do_step=(it < 3))
        torch.cuda.synchronize()
        res[mode] = (cpu(fld.grads).copy(), cpu(fld.params).copy(), fld.losses()['loss'])
    g0, p0, l0 = res['off']
    for mode in ('list', 'all'):
        g, p, l = res[mode]
        assert abs(l - l0) <= 1e-6 * abs(l0)
        # after three Adam steps the parameters differ by what summation order does to a gradient near zero (eps = 1e-15:
        # Adam moves such an entry by ~lr whatever its size): bound the fraction
        assert (np.abs(p - p0) > 0.05 * cfg['lrate']).mean() < 2e-3, mode
        assert np.linalg.norm(g - g0) <= 2e-3 * np.linalg.norm(g0), mode


@pytest.mark.parametrize("ff,tail,R", [(0, False, 256), (0, True, 256), (2, True, 256), (0, True, 203)])
def test_one_stream_backward_equals_two_streams(nof, ff, tail, R):
    """round 6: the backward tail as ONE chain -- { large levels' scatter | dL/dx } as roles of one launch, { LDS levels | MLP row
    reduction | per-ray pose rows } as roles of the next (nof_hash_encode_bwd_step) -- against the two-stream tail: a first step's
    gradients (pose and frame-feature rows included; dL/dx itself is bit-equal in test_hash_backward_over_the_list_equals_whole_batch),
    then the parameters after three Adam steps.  tail: the one-chain side also with the optimiser launch that carries the per-frame
    pose sums and the next step's operand image + pose table (nof_adam_step_tail; with frame features it must fall back by itself)."""
    from tests.test_gpu_step import _pair
    res = {}
    for one in (False, True):
        cfg, fld, orc, batch, rng = _pair(nof, 'fp16x3', ff=ff, ns=2, nc=3, R=R)      # (203 rays: ragged last workgroups of every role)
        fld.one_stream_backward = one
        fld.fused_tail = one and tail
        R = batch.shape[0]
        pool = U.dev(batch)
        first = None
        for it in range(4):
            u1 = rng.random((R, cfg['N_samples'])).astype(np.float32)
            u2 = rng.random((R, cfg['N_samples_around_depth'])).astype(np.float32)
            fld.train_step(pool, None, R, U.dev(u1), U.dev(u2), do_step=(0 < it < 4))
            if it == 0:
                torch.cuda.synchronize()
                first = {k: cpu(fld._seg(fld.grads, k)).copy() for k in ('table', 'mlp', 'pose') + (('feat',) if ff else ())}
                fld.grads.zero_()
        torch.cuda.synchronize()
        assert fld._buffers(R, cfg['N_samples'] + cfg['N_samples_around_depth'])['dview'].abs().max().item() == 0   # re-zeroed for the next step
        assert (fld._tail_step == fld.global_step) == (one and tail and ff == 0)
        res[one] = (first, cpu(fld.params).copy(), fld.losses()['loss'])
    (g0, p0, l0), (g1, p1, l1) = res[False], res[True]
    assert abs(l1 - l0) <= 1e-6 * abs(l0)
    for k in g0:
        assert np.abs(g0[k]).max() > 0, k
        assert np.linalg.norm(g1[k] - g0[k]) <= 1e-4 * np.linalg.norm(g0[k]), k
    assert (np.abs(p1 - p0) > 0.05 * cfg['lrate']).mean() < 2e-3
