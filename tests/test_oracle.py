"""CPU tests: the oracle (oracle/nof_oracle.py) and the product's host logic against golden vectors produced by the
reference's OWN code (tests/golden/make_golden.py -> reference_vectors.npz).  Nothing here reads /root/reference."""
import os

import numpy as np
import pytest
import torch

from oracle import nof_oracle as O

G = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'reference_vectors.npz'), allow_pickle=False)


def test_sh_encoder_matches_reference():
    d = torch.from_numpy(G['sh_dirs'])
    assert np.abs(O.sh_encode(d, 3).numpy() - G['sh3']).max() < 1e-6
    assert np.abs(O.sh_encode(d, 4).numpy() - G['sh4']).max() < 1e-6


@pytest.mark.parametrize("tag,nl,nlc,ichv", [('ref', 2, 3, 9), ('base', 3, 2, 11)])
def test_nerfsmall_init_and_forward_match_reference(tag, nl, nlc, ichv):
    shape = O.FieldShape(input_ch=32, input_ch_views=ichv, num_layers=nl, num_layers_color=nlc)
    torch.manual_seed(123)
    params = O.init_mlp_params(shape)                    # same RNG consumption as NeRFSmall.__init__
    flat = torch.cat([torch.cat([W.reshape(-1), b.reshape(-1)]) for W, b in params]).numpy()
    assert np.array_equal(flat, G[f'mlp_{tag}_flat'])    # bit-identical initial weights (incl. the 0.1 SDF bias)
    x = torch.from_numpy(G[f'mlp_{tag}_x'])
    assert np.abs(O.mlp_forward(shape, params, x).numpy() - G[f'mlp_{tag}_y']).max() < 1e-6
    assert np.abs(O.mlp_forward_sdf(shape, params, x[:, :32]).numpy() - G[f'mlp_{tag}_sdf']).max() < 1e-6


def test_product_init_consumes_rng_like_reference():
    """bundlesdf_amd.field.NeuralObjectField.init_parameters draws the MLP exactly like NeRFSmall (CPU only part)."""
    from bundlesdf_amd import lib
    torch.manual_seed(123)
    chunks = []
    desc, dims = lib.make_mlp_desc(2, 3, 32, 9)
    for l, (o, i) in enumerate(dims):
        lin = torch.nn.Linear(i, o, bias=True)
        if l == 1:
            torch.nn.init.constant_(lin.bias, 0.1)
        chunks += [lin.weight.detach().reshape(-1), lin.bias.detach().reshape(-1)]
    assert np.array_equal(torch.cat(chunks).numpy(), G['mlp_ref_flat'])
    assert desc.n_params == len(G['mlp_ref_flat']) == 9107


@pytest.mark.parametrize("N", [64, 128, 2, 33])
def test_sample_rays_uniform_bit_exact(N):
    z = O.sample_rays_uniform(N, G['sru_near'], G['sru_far'], G[f'sru_{N}_u'])
    assert np.array_equal(z.view(np.uint32), G[f'sru_{N}_z'].view(np.uint32))


def test_linspace_is_torch_linspace():
    for N in (2, 3, 32, 33, 64, 128, 192, 256, 320):
        assert np.array_equal(O.linspace01(N), torch.linspace(0, 1, N).numpy())


def test_truncation_and_lr_schedule():
    keys, vals = G['step_cfg_keys'], G['step_cfg_vals']
    cfg = O.default_cfg(**{k: (int(v) if float(v).is_integer() and k not in ('sc_factor',) else float(v)) for k, v in zip(keys, vals)})
    assert abs(O.get_truncation(cfg, 37) - float(G['trunc_const'])) < 1e-12
    assert abs(O.get_truncation(dict(cfg, trunc_decay_type='linear', trunc_start=0.03), 37) - float(G['trunc_linear'])) < 1e-12
    assert abs(O.get_truncation(dict(cfg, trunc_decay_type='exp', trunc_start=0.03), 37) - float(G['trunc_exp'])) < 1e-12
    lr = 0.01 * 0.1 ** (40.0 / (cfg['n_step'] + 1))
    assert abs(lr - float(G['lr_step40'])) < 1e-15


def _step_cfg():
    keys, vals = G['step_cfg_keys'], G['step_cfg_vals']
    cfg = O.default_cfg()
    for k, v in zip(keys, vals):
        k = str(k)
        cfg[k] = int(v) if isinstance(cfg.get(k, 0.5), (int, bool)) and float(v).is_integer() else float(v)
    return cfg


def _unflatten(shape, flat):
    s, c = shape.layer_dims()
    out, o = [], 0
    for (od, idim) in s + c:
        W = torch.from_numpy(flat[o:o + od * idim].reshape(od, idim).copy())
        o += od * idim
        b = torch.from_numpy(flat[o:o + od].copy())
        o += od
        out.append([W, b])
    return out


def test_full_step_matches_reference_driven_run():
    """OracleField.train_step == the reference's own train_loop/render_rays/run_network/raw2outputs/get_sdf_loss executed on
    CPU (with only the hash encoder, octree tracer, sampler kernel and se3_exp_map injected): z samples, raw, weights,
    rgb_map, total loss and the gradient of EVERY parameter group."""
    cfg = _step_cfg()
    occ, c2w, batch = G['step_occ'], G['step_c2w'], G['step_batch']
    geo = O.HashGeometry(cfg['num_levels'], 2, cfg['base_res'], cfg['log2_hashmap_size'], cfg['finest_res'])
    shape = O.FieldShape(input_ch=geo.out_dim, input_ch_views=9 + cfg['frame_features'])
    fld = O.OracleField(cfg, geo, shape, c2w.shape[0], c2w, occ, table=G['step_table'], mlp=_unflatten(shape, G['step_mlp_flat']),
                        pose=G['step_pose'], feat=G['step_feat'])
    fld.global_step = 1
    assert np.abs(O.pose_matrices(fld.pose, cfg['max_trans'] * cfg['sc_factor'], cfg['max_rot']).detach().numpy()
                  - G['step_pose_mats']).max() < 1e-6
    out = fld.train_step(batch, G['step_u_occ'], G['step_u_dep'], do_step=False)
    assert np.abs(out['z_vals'].numpy() - G['step_z']).max() < 2e-6
    assert np.array_equal(out['fwd']['valid_samples'].numpy(), G['step_valid'])
    v = G['step_valid']
    assert np.abs(out['fwd']['raw'].detach().numpy() - G['step_raw'])[v].max() < 2e-5
    assert np.abs(out['fwd']['weights'].detach().numpy() - G['step_weights']).max() < 1e-6
    assert np.abs(out['fwd']['rgb_map'].detach().numpy() - G['step_rgb_map']).max() < 1e-6
    assert abs(float(out['losses']['loss']) - float(G['step_loss'])) < 2e-5 * abs(float(G['step_loss']))
    n = int(G['step_n_grads'])
    assert len(out['grads']) == n
    for i in range(n):
        ref, got = G[f'step_grad_{i}'], out['grads'][i].numpy()
        assert np.abs(got - ref).max() < 1e-4 * max(np.abs(ref).max(), 1e-6) + 1e-7, i


def test_render_images_matches_reference_driven_run():
    """OracleField.render_rays_image == the reference's own render_images (nerf_runner.py:586-637: render -> batchify_rays in
    chunks of N_rand -> render_rays with perturb=False; depth = z at the first SDF sign change, far*sc for rays without one)
    executed on CPU by tests/golden/make_golden.py on the step fixture's scene, with the sdf bias shifted so that rays cross zero."""
    cfg = dict(_step_cfg(), N_rand=int(G['render_n_rand']))
    occ, c2w, batch = G['step_occ'], G['step_c2w'], G['step_batch']
    geo = O.HashGeometry(cfg['num_levels'], 2, cfg['base_res'], cfg['log2_hashmap_size'], cfg['finest_res'])
    shape = O.FieldShape(input_ch=geo.out_dim, input_ch_views=9 + cfg['frame_features'])
    fld = O.OracleField(cfg, geo, shape, c2w.shape[0], c2w, occ, table=G['step_table'], mlp=_unflatten(shape, G['render_mlp_flat']),
                        pose=G['step_pose'], feat=G['step_feat'])
    fld.global_step = 1
    img_i = int(G['render_img_i'])
    rows = batch[batch[:, 8] == img_i]
    out = fld.render_rays_image(rows)
    assert out['raw'].shape == G['render_raw'].shape and cfg['N_rand'] < rows.shape[0]      # several chunks
    assert np.abs(out['z_vals'].numpy() - G['render_z']).max() < 2e-6
    assert np.array_equal(out['valid_samples'].numpy(), G['render_valid'])
    assert np.abs(out['raw'].numpy() - G['render_raw']).max() < 2e-5
    # scatter like the reference (:616-634) and compare the images
    H, W = (int(x) for x in G['render_HW'])
    X = rows[:, 0:3].astype(np.float64).copy()
    X[:, [1, 2]] = -X[:, [1, 2]]
    proj = (G['render_K'] @ X.T).T
    uvs = (proj / proj[:, 2].reshape(-1, 1)).round().astype(int)
    rgb_full, depth_full = np.zeros((H, W, 3)), np.zeros((H, W))
    rgb_full[uvs[:, 1], uvs[:, 0]] = out['rgb_map'].numpy()
    depth_full[uvs[:, 1], uvs[:, 0]] = out['depth'].numpy()
    far = cfg['far'] * cfg['sc_factor']
    assert np.array_equal(depth_full == far, G['render_depth_full'] == far) and (G['render_depth_full'] == far).any()
    assert np.abs(depth_full - G['render_depth_full']).max() < 2e-6
    assert np.abs(rgb_full - G['render_rgb_full']).max() < 1e-6


def test_unperturbed_sampler_is_the_linspace():
    """perturb=False (nerf_runner.py:78 is not entered): z = near (1 - t) + far t with torch's own linspace, no clip."""
    near = np.array([[0.5], [1.0], [2.0]], np.float32)
    far = np.array([[0.5], [3.0], [2.5]], np.float32)
    for N in (2, 32, 64, 128):
        t = torch.linspace(0., 1., steps=N).reshape(1, -1)
        want = (torch.from_numpy(near) * (1. - t) + torch.from_numpy(far) * t).numpy()
        assert np.array_equal(O.sample_rays_uniform(N, near, far, None), want)


def test_raw_at_invalid_samples_is_mlp_of_zero_features():
    """invalid samples still run through both MLPs with zero hash features (nerf_runner.py:1247,1289-1294)."""
    v = G['step_valid']
    if v.all():
        pytest.skip('no invalid sample in the golden batch')
    cfg = _step_cfg()
    geo = O.HashGeometry(cfg['num_levels'], 2, cfg['base_res'], cfg['log2_hashmap_size'], cfg['finest_res'])
    shape = O.FieldShape(input_ch=geo.out_dim, input_ch_views=9 + cfg['frame_features'])
    fld = O.OracleField(cfg, geo, shape, G['step_c2w'].shape[0], G['step_c2w'], G['step_occ'], table=G['step_table'],
                        mlp=_unflatten(shape, G['step_mlp_flat']), pose=G['step_pose'], feat=G['step_feat'])
    fld.global_step = 1
    with torch.no_grad():
        fw = fld.forward(torch.from_numpy(G['step_batch']), torch.from_numpy(G['step_z']))
    assert np.abs(fw['raw'].numpy() - G['step_raw'])[~v].max() < 2e-5


def test_run_network_points_is_the_reference_run_of_run_network():
    """OracleField.run_network_points (what mesh_vertex_color_from_network calls, nerf_runner.py:1412-1424: run_network on free
    points with an identity transform, one view direction, one frame's latent code) pinned on the REFERENCE-executed fixture: for
    every ray of the golden batch its sample points, its world view direction and its frame reproduce the raw outputs the
    reference's own run_network produced for that ray (`step_raw`), valid and invalid samples alike."""
    cfg = _step_cfg()
    geo = O.HashGeometry(cfg['num_levels'], 2, cfg['base_res'], cfg['log2_hashmap_size'], cfg['finest_res'])
    shape = O.FieldShape(input_ch=geo.out_dim, input_ch_views=9 + cfg['frame_features'])
    fld = O.OracleField(cfg, geo, shape, G['step_c2w'].shape[0], G['step_c2w'], G['step_occ'], table=G['step_table'],
                        mlp=_unflatten(shape, G['step_mlp_flat']), pose=G['step_pose'], feat=G['step_feat'])
    fld.global_step = 1
    batch = torch.from_numpy(G['step_batch'])
    with torch.no_grad():
        fw = fld.forward(batch, torch.from_numpy(G['step_z']))
        rays_d = batch[:, 0:3]
        viewdirs = rays_d / rays_d.norm(dim=-1, keepdim=True)
        fid = batch[:, 8].long()
        dirs_w = (fld.frame_tf()[fid][:, :3, :3] @ viewdirs[:, :, None])[:, :, 0]
    worst, n_out = 0.0, 0
    for r in range(0, batch.shape[0], max(1, batch.shape[0] // 24)):          # two dozen rays of different frames
        pts = fw['pts_w'][r].numpy()
        got = fld.run_network_points(pts, viewdir=tuple(float(x) for x in dirs_w[r]), frame_id=int(fid[r])).numpy()
        worst = max(worst, float(np.abs(got - G['step_raw'][r]).max()))
        n_out += int((np.abs(pts) > 1).any(1).sum())
    assert worst < 2e-5, worst
    assert n_out > 0 or G['step_valid'].all()                               # (points outside the unit cube were among them)


# ---- product host logic (bundlesdf_amd/*.py, CPU parts) against the same reference vectors -----------------------
def test_camera_rays_and_ray_box():
    from bundlesdf_amd.nerf_helpers import get_camera_rays_np, ray_box_intersection_batch
    K = G['K'] * np.array([[0.1], [0.1], [1]])
    assert np.array_equal(get_camera_rays_np(48, 64, K), G['cam_rays'])
    tmin, tmax = ray_box_intersection_batch(G['rb_o'], G['rb_d'], np.array([[-1, -1, -1], [1, 1, 1]]))
    assert np.allclose(tmin.numpy(), G['rb_tmin'], rtol=0, atol=1e-12) and np.allclose(tmax.numpy(), G['rb_tmax'], rtol=0, atol=1e-12)


def test_near_far_filter():
    from bundlesdf_amd.rays import compute_near_far_and_filter_rays
    got = compute_near_far_and_filter_rays(G['nf_cam'], G['nf_rays'].copy(), {'bounding_box': [[-1, -1, -1], [1, 1, 1]]})
    assert got.shape == G['nf_out'].shape and np.allclose(got, G['nf_out'], rtol=0, atol=1e-12)


def test_dataloader_order_matches_reference():
    from bundlesdf_amd.rays import DataLoader
    torch.manual_seed(0)
    dl = DataLoader(rays=torch.arange(1000).float().reshape(-1, 1), batch_size=300)
    for k in range(9):
        next(dl)
        assert np.array_equal(dl.batch_ray_ids.numpy(), G['dl_seq'][k]), k


def test_dilate_mask_is_cv2_dilate():
    """cv2.dilate with a k x k ones kernel: dst(y,x) = max src(y+dy, x+dx), dy,dx in [-k//2, k-1-k//2] (anchor k//2)."""
    from bundlesdf_amd.rays import dilate_mask
    rng = np.random.default_rng(0)
    m = (rng.random((37, 53)) > 0.97).astype(np.uint8)
    m[0, 0] = m[-1, -1] = m[5, 52] = 1
    for k in (1, 2, 3, 6, 7, 10):
        ref = np.zeros_like(m)
        ys, xs = np.nonzero(m)
        for y, x in zip(ys, xs):                       # a set pixel reaches y - (k-1-k//2) .. y + k//2
            ref[max(0, y - (k - 1 - k // 2)):y + k // 2 + 1, max(0, x - (k - 1 - k // 2)):x + k // 2 + 1] = 1
        assert np.array_equal(dilate_mask(m, k), ref), k


# ---- properties of the oracle itself ------------------------------------------------------------------------------
def test_hash_geometry_matches_survey_table_sizes():
    assert O.HashGeometry(16, 2, 16, 14, 256).n_entries == 242808          # cfg1
    assert O.HashGeometry(16, 2, 16, 19, 256).n_entries == 4555440         # cfg2
    assert O.HashGeometry(16, 2, 16, 22, 512).n_entries == 29580312        # cfg5


def _dda_numpy(occ, o, d):
    """independent numpy mirror of the product's DDA walk (bundlesdf_amd/csrc/nof_trace.hip:trace_one)"""
    f32 = np.float32
    n = occ.shape[0]
    cs = f32(2.0) / f32(n)
    out = []
    for r in range(o.shape[0]):
        zero = np.abs(d[r]) < f32(1e-20)
        with np.errstate(divide='ignore', over='ignore'):
            inv = np.where(zero, f32(0), f32(1.0) / np.where(zero, f32(1), d[r])).astype(f32)
        step = np.where(d[r] > 0, 1, -1)

        def slab(a, i):
            lo, hi = f32(i) * cs - f32(1), f32(i + 1) * cs - f32(1)
            if zero[a]:
                ins = lo <= o[r, a] < hi
                return (f32(-np.inf), f32(np.inf)) if ins else (f32(np.inf), f32(-np.inf))
            t0, t1 = f32((lo - o[r, a]) * inv[a]), f32((hi - o[r, a]) * inv[a])
            return min(t0, t1), max(t0, t1)
        tenter, texit = f32(0), f32(np.inf)
        for a in range(3):
            if zero[a]:
                if not (-1 <= o[r, a] < 1):
                    texit = f32(-np.inf)
            else:
                t0, t1 = f32((f32(-1) - o[r, a]) * inv[a]), f32((f32(1) - o[r, a]) * inv[a])
                tenter, texit = max(tenter, min(t0, t1)), min(texit, max(t0, t1))
        hits = []
        if tenter <= texit:
            c = [0, 0, 0]
            for a in range(3):
                p = o[r, a] if zero[a] else f32(o[r, a] + f32(tenter * d[r, a]))
                i = int(np.floor(f32(f32(p + f32(1)) / cs)))
                i = min(max(i, 0), n - 1)
                if not zero[a]:
                    for _ in range(4):
                        tmin, tmax = slab(a, i)
                        if tmax < tenter and 0 <= i + step[a] < n:
                            i += step[a]
                        elif tmin > tenter and 0 <= i - step[a] < n:
                            i -= step[a]
                        else:
                            break
                c[a] = i
            for _ in range(3 * n + 8):
                sl = [slab(a, c[a]) for a in range(3)]
                tin = max(sl[0][0], sl[1][0], sl[2][0], f32(0))
                tout = min(sl[0][1], sl[1][1], sl[2][1])
                if tin <= tout and occ[c[0], c[1], c[2]]:
                    if tin == 0 or tout == 0:
                        break
                    if not abs(f32(tout - tin)) < f32(1e-4):
                        hits.append(((c[0] * n + c[1]) * n + c[2], tin, tout))
                a = 0
                if sl[1][1] < sl[a][1]:
                    a = 1
                if sl[2][1] < sl[a][1]:
                    a = 2
                if zero[a]:
                    break
                c[a] += step[a]
                if c[a] < 0 or c[a] >= n:
                    break
        out.append(hits)
    return out


@pytest.mark.parametrize("level,fill", [(2, 0.6), (4, 0.25), (6, 0.03)])
def test_dda_closed_form_enumeration_equals_the_walk(level, fill):
    """tools/dda_closed_form.py (groundwork for a lane-parallel ray marcher): the cells of a ray enumerated from the three
    sorted lists of exit times -- one independent computation per crossing -- are the walk's cells, in the walk's order, with the
    walk's float32 intervals: random rays, and rays built to tie (lattice origins, axis-parallel / face-diagonal / space-diagonal /
    small-integer-ratio directions through cell corners and edges, rays inside the grid, rays that miss)."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools'))
    from dda_closed_form import enumerate_cells
    from tests import util as U
    n = 1 << level
    occ = U.random_occ(n, fill, seed=40 + level)
    o, d = U.random_rays(120, seed=50 + level)
    rng = np.random.default_rng(60 + level)
    cs = 2.0 / n
    extra_o, extra_d = [], []
    dirs = [(1, 0, 0), (0, -1, 0), (0, 0, 1), (1, 1, 0), (1, -1, 0), (0, 1, 1), (1, 1, 1), (-1, 1, -1), (1, 2, 0), (2, 1, 3), (1, 2, 4),
            (-3, 1, 2), (1, 1, 2), (4, 4, 1)]
    for dv in dirs:
        dv = np.array(dv, np.float64)
        for _ in range(6):
            lat = rng.integers(0, n + 1, size=3) * cs - 1.0                    # a grid vertex: the ray runs through corners / edges
            back = rng.integers(1, 3 * n)
            extra_o.append(lat - dv / np.abs(dv).max() * back * cs)            # start a whole number of cells back (often outside)
            extra_d.append(dv / np.linalg.norm(dv))
        extra_o.append(np.array([0.5 * cs - 1.0, 0.5 * cs - 1.0, 0.5 * cs - 1.0]) + rng.integers(0, n, size=3) * cs)   # cell centre, inside
        extra_d.append(dv / np.linalg.norm(dv))
    extra_o += [[-3, 5, 0], [0.2, 0.3, -4], [1.0, 1.0, 1.0], [-1.0, -1.0, -1.0]]                                       # misses / corners
    extra_d += [[1, 0, 0], [0, 0, -1], [-1, -1, -1], [1, 1, 1]]
    extra_d[-2] = list(np.array(extra_d[-2]) / np.sqrt(3)); extra_d[-1] = list(np.array(extra_d[-1]) / np.sqrt(3))
    o = np.concatenate([o, np.array(extra_o)]).astype(np.float32)
    d = np.concatenate([d, np.array(extra_d)]).astype(np.float32)
    walk = _dda_numpy(occ, o, d)
    n_hits = n_tied = 0
    for r in range(len(o)):
        got = enumerate_cells(occ, o[r], d[r])
        assert len(got) == len(walk[r]), (r, o[r], d[r], len(got), len(walk[r]))
        for k, ((c, a, b), (c2, a2, b2)) in enumerate(zip(got, walk[r])):
            assert c == c2 and np.float32(a) == np.float32(a2) and np.float32(b) == np.float32(b2), (r, k)
        n_hits += len(got)
        n_tied += int(r >= 120)
    assert n_hits > 50 and n_tied > 80


@pytest.mark.parametrize("level,fill", [(3, 0.4), (4, 0.2), (5, 0.05)])
def test_dda_walk_equals_bruteforce_definition(level, fill):
    """the traversal order the HIP kernel uses visits exactly the cells of the brute-force slab definition"""
    from tests import util as U
    n = 1 << level
    occ = U.random_occ(n, fill, seed=10 + level)
    o, d = U.random_rays(150, seed=20 + level)
    o[0], d[0] = [-3, 0.013, 0.021], [1, 0, 0]
    o[1], d[1] = [-2, -2, -2], np.array([1, 1, 1], np.float32) / np.sqrt(np.float32(3))
    o[2], d[2] = [-3, 1.0, 0.0], [1, 0, 0]
    o[3], d[3] = [-3, 0.5, 0.5], [1, 0, 0]
    o[4], d[4] = [0.011, 0.022, 0.033], [0, 0, 1]
    o, d = o.astype(np.float32), d.astype(np.float32)
    tio, cid, nh = O.trace_rays(occ, o, d)
    walk = _dda_numpy(occ, o, d)
    for r in range(len(o)):
        assert len(walk[r]) == nh[r], r
        for k, (c, a, b) in enumerate(walk[r]):
            assert c == cid[r, k] and np.float32(a) == tio[r, k, 0] and np.float32(b) == tio[r, k, 1], (r, k)


def test_sampler_edge_cases():
    """zero-length clipped boxes, rays without hits, BAD_DEPTH rays (SURVEY 8c golden list iii)"""
    cfg = O.default_cfg(sc_factor=5.0, N_samples=16, N_samples_around_depth=8, far=1.0)
    tio = np.zeros((4, 3, 2), np.float32)
    tio[0, :2] = [[2.0, 2.5], [2.6, 3.4]]           # second box is cut to zero length by depth + trunc
    tio[1, :1] = [[1.0, 1.2]]
    tio[3, :3] = [[1.0, 1.5], [1.5, 2.0], [2.5, 2.75]]
    depth = np.array([2.3, 99 * 5.0, 2.0, 2.6], np.float32)
    vz = np.ones(4, np.float32)
    rng = np.random.default_rng(0)
    u1, u2 = rng.random((4, 16)).astype(np.float32), rng.random((4, 8)).astype(np.float32)
    trunc = O.get_truncation(cfg)
    z = O.sample_z(tio, vz, depth, cfg, trunc, u1, u2)
    assert (z[2, :16] == 0).all()                                        # no box -> zeros (common.cu:54)
    assert (z[0, :16] >= 2.0).all() and (z[0, :16] <= np.float32(2.3) + np.float32(trunc)).all()
    assert (np.diff(z[:, :16], axis=1) >= 0).all()                      # stratified samples stay ordered inside a part
    assert ((z[1, 16:] >= 1.0) & (z[1, 16:] <= 1.2)).all()              # BAD_DEPTH ray re-samples the occupied boxes
    inside = ((z[3, :16] >= 1.0) & (z[3, :16] <= 2.0)) | ((z[3, :16] >= 2.5) & (z[3, :16] <= 2.75))
    assert inside.all()


def test_se3_exp_is_a_rigid_transform_and_matches_series():
    xi = torch.tensor([[0.1, -0.2, 0.3, 0.0, 0.0, 0.0], [0.0, 0.0, 0.0, 0.3, -0.2, 0.1], [0.05, 0.02, -0.01, 1e-3, 0, 0]])
    T = O.se3_exp(xi)
    Rm = T[:, :3, :3]
    assert torch.allclose(Rm @ Rm.transpose(1, 2), torch.eye(3).expand(3, 3, 3), atol=1e-6)
    assert torch.allclose(T[0, :3, 3], xi[0, :3], atol=1e-6)            # w -> 0: t = u (up to the 1e-4 clamp's O(theta^2))
    M = torch.zeros(4, 4, dtype=torch.float64)
    w, u = xi[1, 3:].double(), xi[1, :3].double()
    M[:3, :3] = torch.tensor([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    M[:3, 3] = u
    assert torch.allclose(torch.matrix_exp(M).float(), T[1], atol=1e-6)  # closed form == matrix exponential


def test_adam_restatement_equals_torch():
    rng = np.random.default_rng(0)
    p0 = rng.normal(size=100)
    p = torch.tensor(p0, dtype=torch.float64, requires_grad=True)
    opt = torch.optim.Adam([p], lr=0.01, betas=(0.9, 0.999), eps=1e-15)
    pp, m, v = p0.copy(), np.zeros(100), np.zeros(100)
    for t in range(1, 5):
        g = rng.normal(size=100) * 10.0 ** -t
        p.grad = torch.tensor(g)
        opt.step()
        pp, m, v = O.adam_reference_step(pp, g, m, v, t, 0.01)
        assert np.abs(pp - p.detach().numpy()).max() < 1e-12
