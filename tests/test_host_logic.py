"""Host-side logic of the product that needs no GPU: the level table handed to the kernels, the MLP descriptor / parameter
layout, configuration validation, schedules, the NumPy ray-pool helpers and the mesh container."""
import ctypes as C

import numpy as np
import pytest
import torch

from bundlesdf_amd import lib
from bundlesdf_amd.config import default_cfg, validate_cfg
from oracle import nof_oracle as O


@pytest.mark.parametrize("L,T,base,finest", [(16, 19, 16, 256), (16, 14, 16, 256), (16, 22, 16, 512), (4, 22, 16, 128),
                                              (8, 14, 16, 128), (16, 19, 16, 512)])
def test_level_table_matches_oracle_and_reference_rule(L, T, base, finest):
    """grid.py:110,127-134 (allocation, float64) + gridencoder.cu:154-156 (indexing constants, float32): the struct the
    kernels receive and the oracle's geometry must be the same numbers, bit for bit."""
    g, offsets, n, pls = lib.make_hash_grid(L, 2, base, T, finest)
    geo = O.HashGeometry(n_levels=L, level_dim=2, base_resolution=base, log2_hashmap_size=T, desired_resolution=finest)
    assert n == geo.n_entries and np.array_equal(offsets, geo.offsets)
    for l in range(L):
        assert np.float32(g.scale[l]) == geo.scale[l] and g.resolution[l] == geo.resolution[l]
        assert g.size[l] == geo.size[l] and g.offset[l] == geo.offsets[l] and bool(g.hashed[l]) == bool(geo.hashed[l])
        assert g.size[l] % 8 == 0 and g.size[l] <= 2 ** T
        if g.hashed[l]:
            assert g.size[l] == 2 ** T                    # the kernels mask instead of dividing for hashed levels
    # the documented cfg2 / cfg5 table sizes
    if (L, T, finest) == (16, 19, 256):
        assert n == 4555440
    if (L, T, finest) == (16, 22, 512):
        assert n == 29580312


@pytest.mark.parametrize("ns,nc,ff", [(2, 3, 0), (3, 2, 2), (2, 2, 0), (3, 3, 4)])
def test_mlp_descriptor_is_pytorch_parameter_order(ns, nc, ff):
    """NeRFSmall.parameters() order: sigma_net W,b per layer, then color_net W,b (nerf_helpers.py:243-321)."""
    n_view = 9 + ff
    desc, dims = lib.make_mlp_desc(ns, nc, 32, n_view, 1)
    shape = O.FieldShape(input_ch=32, input_ch_views=n_view, num_layers=ns, num_layers_color=nc)
    s, c = shape.layer_dims()
    assert [tuple(d) for d in dims] == [tuple(x) for x in list(s) + list(c)]
    off = 0
    for l, (o, i) in enumerate(dims):
        assert desc.w_off[l] == off and desc.out_dim[l] == o and desc.in_dim[l] == i
        off += o * i
        assert desc.b_off[l] == off
        off += o
    assert desc.n_params == off == shape.n_params()
    assert dims[ns - 1][0] == 16 and dims[-1][0] == 3 and dims[ns][1] == n_view + 15


def test_unsupported_configuration_is_rejected_loudly():
    validate_cfg(default_cfg())
    validate_cfg(default_cfg(eikonal_weight=0.1))            # implemented (nof_eikonal)
    for key, val in (('depth_weight', 1.0), ('N_importance', 64), ('N_samples_around_depth', 0),
                     ('use_viewdirs', 0), ('raw_noise_std', 1.0), ('feature_grid_dim', 4)):
        with pytest.raises(NotImplementedError, match=key.split('_')[0]):
            validate_cfg(default_cfg(**{key: val}))


def test_runner_needs_the_gpu_and_the_library():
    """no CPU fallback: constructing the runner without an MI355X is an error, not a slow path"""
    if torch.cuda.is_available():
        pytest.skip('a GPU is present')
    from bundlesdf_amd.nerf_runner import NerfRunner
    from bundlesdf_amd import synthetic
    pool = synthetic.make_pool(n_frames=1, H=24, W=32, fx=30.0, seed=0, pose_noise=False)
    cfg = default_cfg(sc_factor=pool['sc_factor'], translation=pool['translation'])
    with pytest.raises(lib.NofError, match='no CPU'):
        NerfRunner(cfg, pool['rgbs'], depths=pool['depths'], masks=pool['masks'], normal_maps=None, poses=pool['poses'],
                   K=pool['K'], build_octree_pcd=synthetic.PointCloud(pool['pcd_normalized']))


def test_dilate_mask_window_convention():
    """cv2.dilate with an even k x k kernel anchors at k//2: window offsets -k//2 .. k-1-k//2 (nerf_runner.py:275-283 uses
    100 and 60); checked against a brute-force loop on an asymmetric mask."""
    from bundlesdf_amd.rays import dilate_mask
    rng = np.random.default_rng(0)
    m = (rng.random((23, 31)) < 0.03).astype(np.uint8)
    for k in (2, 3, 6, 9):
        ref = np.zeros_like(m)
        lo, hi = -(k // 2), k - 1 - k // 2
        for v in range(m.shape[0]):
            for u in range(m.shape[1]):
                v0, v1 = max(v + lo, 0), min(v + hi, m.shape[0] - 1)
                u0, u1 = max(u + lo, 0), min(u + hi, m.shape[1] - 1)
                ref[v, u] = m[v0:v1 + 1, u0:u1 + 1].max()
        assert np.array_equal(dilate_mask(m, k), ref), k
    assert np.array_equal(dilate_mask(m, 1), m)


def test_data_loader_epochs():
    """nerf_runner.py:90-107: consecutive slices of one permutation; a new permutation when fewer than batch+1 ids remain"""
    from bundlesdf_amd.rays import DataLoader
    torch.manual_seed(0)
    rays = torch.arange(10 * 12, dtype=torch.float32).reshape(10, 12)
    dl = DataLoader(rays, batch_size=4)
    first = dl.ids.clone()
    a, b = dl.next_ids(), dl.next_ids()
    assert torch.equal(a, first[:4]) and torch.equal(b, first[4:8])
    c = dl.next_ids()                                    # 2 left < 4 + 1 -> reshuffle, the tail is dropped
    assert dl.pos == 4 and torch.equal(c, dl.ids[:4]) and sorted(dl.ids.tolist()) == list(range(10))
    assert torch.equal(next(dl), rays[dl.batch_ray_ids])


def test_learning_rate_and_truncation_schedules():
    """schedule_lr every 10 steps after step 0 (nerf_runner.py:579-583,762-763); get_truncation (:663-676)"""
    from bundlesdf_amd.field import NeuralObjectField

    class Stub:
        learning_rates = NeuralObjectField.learning_rates
        truncation = NeuralObjectField.truncation
    s = Stub()
    s.cfg = default_cfg(n_step=100, lrate=0.01, lrate_pose=0.02, decay_rate=0.1, trunc=0.01, trunc_start=0.04, sc_factor=2.0)
    s.N_iters = 101
    for step, g in ((0, 0), (10, 0), (11, 10), (20, 10), (21, 20), (100, 90)):
        s.global_step = step
        k = 1.0 if g == 0 else 0.1 ** (g / 101)
        lr, lrp = s.learning_rates()
        assert abs(lr - 0.01 * k) < 1e-12 and abs(lrp - 0.02 * k) < 1e-12, step
    s.global_step = 50
    assert s.truncation() == pytest.approx(0.01 * 2.0)
    s.cfg['trunc_decay_type'] = 'linear'
    assert s.truncation() == pytest.approx((0.04 - 0.03 * 0.5) * 2.0)
    s.cfg['trunc_decay_type'] = 'exp'
    lamb = np.log(0.01 / 0.04) / 25
    assert s.truncation() == pytest.approx(max(0.04 * np.exp(50 * lamb), 0.01) * 2.0)


def test_mesh_container_operations(tmp_path):
    from bundlesdf_amd.mesh import Mesh, largest_component
    v = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1], [5, 5, 5], [6, 5, 5], [5, 6, 5]], dtype=np.float64)
    f = np.array([[0, 1, 2], [0, 1, 3], [0, 2, 3], [1, 2, 3], [4, 5, 6]])
    m = Mesh(v, f)
    big = largest_component(m)
    assert len(big.faces) == 4 and len(big.vertices) == 4
    T = np.eye(4)
    T[:3, 3] = [1, 2, 3]
    m2 = m.copy().apply_transform(T)
    assert np.allclose(m2.vertices, v + [1, 2, 3]) and np.allclose(m.vertices, v)
    for ext in ('obj', 'ply'):
        p = m.export(str(tmp_path / f'm.{ext}'))
        txt = open(p).read()
        assert ('f 1 2 3' in txt) if ext == 'obj' else ('element face 5' in txt)
    dup = Mesh(np.concatenate([v, v[:1]]), np.concatenate([f, [[7, 1, 2]]]))
    dup.merge_vertices()
    dup.remove_duplicate_faces()
    assert len(dup.vertices) == 7 and len(dup.faces) == 5


# ---- plugin-surface helpers pinned against the reference's own code (tests/golden/make_golden_plugin.py) ----------------
def _plugin_golden():
    import os
    return np.load(os.path.join(os.path.dirname(__file__), 'golden', 'plugin_vectors.npz'))


def test_constants_match_reference():
    from bundlesdf_amd import nerf_helpers as H
    g = _plugin_golden()
    assert H.BAD_DEPTH == float(g['BAD_DEPTH']) and np.array_equal(np.asarray(H.BAD_COLOR, dtype=np.float64), g['BAD_COLOR'])
    assert np.array_equal(np.asarray(H.glcam_in_cvcam, dtype=np.float64), g['glcam_in_cvcam'])


def test_preprocess_data_matches_reference():
    from bundlesdf_amd.nerf_helpers import preprocess_data
    g = _plugin_golden()
    r = preprocess_data(g['pp_rgbs'].copy(), g['pp_depths'].copy(), g['pp_masks'].copy(), g['pp_normals'].copy(),
                        g['pp_poses'].copy(), float(g['pp_sc']), g['pp_tr'])
    for k, v in zip(('rgbs', 'depths', 'masks', 'normals', 'poses'), r):
        ref = g['pp_out_' + k]
        assert np.asarray(v).shape == ref.shape and np.asarray(v).dtype == ref.dtype, k
        assert np.array_equal(np.asarray(v), ref), k


def test_pose_hand_back_and_mesh_to_real_world_match_reference():
    """bundlesdf.py:231-235: optimised poses back in real-world OpenCV convention, mesh back to metres"""
    from bundlesdf_amd.nerf_helpers import get_optimized_poses_in_real_world, mesh_to_real_world
    from bundlesdf_amd.mesh import Mesh
    g = _plugin_golden()

    class Poses:
        def get_matrices(self, ids):
            return torch.as_tensor(g['gp_delta'])[torch.as_tensor(ids).long()]
    opt, offset = get_optimized_poses_in_real_world(g['gp_poses'].copy(), Poses(), float(g['pp_sc']), g['pp_tr'])
    assert opt.dtype == g['gp_out'].dtype and np.array_equal(opt, g['gp_out'])
    assert np.array_equal(np.asarray(offset), g['gp_offset'])
    m = mesh_to_real_world(Mesh(g['mw_v'].copy(), np.zeros((0, 3), dtype=np.int64)), np.asarray(offset, dtype=np.float64),
                           g['pp_tr'], float(g['pp_sc']))
    assert np.allclose(np.asarray(m.vertices), g['mw_out'], rtol=0, atol=1e-12)


@pytest.mark.parametrize("tag", ['a', 'b'])
def test_octree_cells_match_reference_build_octree(tag):
    """the points the reference's build_octree hands to kaolin (dilated cell centres, clipped) and its max_level, from a
    reference-driven run of the method itself; the product's cell list must be exactly their quantisation"""
    from bundlesdf_amd.rays import octree_cells
    g = _plugin_golden()
    sv, dil, rv, scf = g[f'oct_{tag}_cfg']
    cfg = dict(octree_smallest_voxel_size=float(sv), octree_dilate_size=float(dil), octree_raytracing_voxel_size=float(rv),
               sc_factor=float(scf))
    q, centres, max_level, level = octree_cells(g[f'oct_{tag}_cloud'], cfg)
    ref = g[f'oct_{tag}_pts']
    assert max_level == int(g[f'oct_{tag}_max_level']) and level == int(np.floor(np.log2(2.0 / (rv * scf))))
    key = lambda a: a[np.lexsort((a[:, 2], a[:, 1], a[:, 0]))]
    assert centres.dtype == ref.dtype and np.array_equal(key(centres), key(ref))
    n = 2 ** max_level
    q_ref = np.floor(np.clip(n * (ref.astype(np.float32) + 1.0) / 2.0, 0, n - 1.0)).astype(np.int32)
    assert np.array_equal(np.unique(q, axis=0), np.unique(q_ref, axis=0))


@pytest.mark.parametrize("valid_only", [1, 0])
def test_make_frame_rays_matches_reference_driven_run(valid_only):
    """the reference's NerfRunner.make_frame_rays executed on CPU (only cv2.dilate stubbed, octree filter off) against
    bundlesdf_amd/rays.py: same rays, same order, same 12 float64 columns (mask dilation 100 / 60//down, ray types, depth
    validity, bounding-box near / far)"""
    from bundlesdf_amd.rays import make_frame_rays
    g = _plugin_golden()
    cfg = default_cfg(near=0.1, far=1.0, sc_factor=float(g['mfr_sc']), down_scale_ratio=4, rays_valid_depth_only=valid_only,
                      use_octree=0, bounding_box=[[-1, -1, -1], [1, 1, 1]])
    for fid in range(2):
        ref = g[f'mfr_{valid_only}_{fid}']
        got = make_frame_rays(fid, g['mfr_images'][fid], g['mfr_depths'][fid], g['mfr_masks'][fid], g['mfr_poses'][fid],
                              g['mfr_K'], cfg)
        assert got.shape == ref.shape and got.dtype == ref.dtype, (got.shape, ref.shape)
        assert np.array_equal(got[:, :10], ref[:, :10])
        assert np.allclose(got[:, 10:], ref[:, 10:], rtol=1e-12, atol=0)


@pytest.mark.parametrize("tag,ns,nc,n_view", [('ref', 2, 3, 9), ('base', 3, 2, 11)])
def test_checkpoint_layout_is_the_reference_state_dict(tag, ns, nc, n_view):
    """bundlesdf_amd/checkpoint.py against NeRFSmall.state_dict() of the reference (key names, order, shapes, values) and
    its forward: a 'model' entry written here loads into the reference and vice versa"""
    from bundlesdf_amd.checkpoint import mlp_flat_from_state, mlp_state_from_flat
    g = _plugin_golden()
    desc, dims = lib.make_mlp_desc(ns, nc, 32, n_view, 1)
    flat = torch.from_numpy(g[f'sd_{tag}_flat'])
    state = mlp_state_from_flat(flat, dims, ns)
    assert list(state.keys()) == [str(k) for k in g[f'sd_{tag}_keys']]
    assert [list(v.shape) + [0] * (2 - v.dim()) for v in state.values()] == g[f'sd_{tag}_shapes'].tolist()
    assert np.array_equal(torch.cat([v.reshape(-1) for v in state.values()]).numpy(), g[f'sd_{tag}_cat'])
    assert torch.equal(mlp_flat_from_state(state, dims, ns), flat)
    # the flat layout is what the oracle (and the kernels' weight packer) consume: same network function as the reference
    shape = O.FieldShape(input_ch=32, input_ch_views=n_view, num_layers=ns, num_layers_color=nc)
    params, off = [], 0
    for o, i in dims:
        W = flat[off:off + o * i].reshape(o, i); off += o * i
        b = flat[off:off + o]; off += o
        params.append((W, b))
    y = O.mlp_forward(shape, params, torch.from_numpy(g[f'sd_{tag}_x'])).detach().numpy()
    assert np.abs(y - g[f'sd_{tag}_y']).max() < 1e-6
    bad = dict(state)
    bad['sigma_net.0.weight'] = bad['sigma_net.0.weight'][:, :16]
    with pytest.raises(ValueError):
        mlp_flat_from_state(bad, dims, ns)
    bad = dict(state)
    bad['color_net.9.weight'] = torch.zeros(1)
    with pytest.raises(ValueError):
        mlp_flat_from_state(bad, dims, ns)


def test_reference_checkpoint_carries_a_loadable_adam_state():
    """to_reference_checkpoint's 'optimizer' entry loads into a torch.optim.Adam built the way the reference's
    create_optimizer builds it (nerf_runner.py:492-504: embeddings, NeRFSmall parameters, feature array | pose array), which
    is what its load_weights does unconditionally (nerf_runner.py:544) -- and reads back into the flat moment buffers."""
    from bundlesdf_amd import checkpoint as ck

    class CpuField:                       # the attributes checkpoint.py uses of a NeuralObjectField, on the CPU
        def __init__(self):
            self.n_entries, self.F, self.ff, self.n_sigma = 96, 3, 2, 2
            self.desc, self.layer_dims = lib.make_mlp_desc(2, 3, 32, 9 + self.ff, 1)
            self.optimize_poses = True
            self.offsets = np.array([0, 40, 96])
            self.n_table, self.n_mlp = 2 * self.n_entries, self.desc.n_params
            self.n_feat, self.n_pose = self.F * self.ff, self.F * 6
            self.n_basic = self.n_table + self.n_mlp + self.n_feat
            n = self.n_basic + self.n_pose
            g = torch.Generator().manual_seed(0)
            self.params, self.grads = torch.randn(n, generator=g), torch.zeros(n)
            self.exp_avg, self.exp_avg_sq = torch.randn(n, generator=g), torch.rand(n, generator=g)
            self.global_step = 37
            self.adam_steps = 37                  # the age of the Adam moments (bias correction): what 'optimizer' carries
            self._packed_step = 0

        def learning_rates(self):
            return 0.004, 0.002

        def load_parameters(self, table=None, mlp=None, feat=None, pose=None):
            for name, val in (('table', table), ('mlp', mlp), ('feat', feat), ('pose', pose)):
                if val is not None:
                    self._seg(self.params, name).copy_(torch.as_tensor(val).reshape(-1))

    from bundlesdf_amd.field import NeuralObjectField
    CpuField._seg = NeuralObjectField._seg
    for name in ('table', 'mlp', 'feat', 'pose'):
        setattr(CpuField, name, property(lambda s, n=name: s._seg(s.params, n)))
    f = CpuField()
    data = ck.to_reference_checkpoint(f, global_step=37)
    # what the reference does with it: the same modules' parameters in the same order, then load_state_dict
    ref_params = [torch.nn.Parameter(torch.zeros(f.n_entries, 2))]
    for o, i in f.layer_dims:
        ref_params += [torch.nn.Parameter(torch.zeros(o, i)), torch.nn.Parameter(torch.zeros(o))]
    ref_params.append(torch.nn.Parameter(torch.zeros(f.F, f.ff)))
    pose = torch.nn.Parameter(torch.zeros(f.F, 6))
    opt = torch.optim.Adam([{'name': 'basic', 'params': ref_params, 'lr': 0.01}, {'name': 'pose_array', 'params': [pose], 'lr': 0.01}],
                           betas=(0.9, 0.999), weight_decay=0, eps=1e-15)
    opt.load_state_dict(data['optimizer'])
    assert [g['name'] for g in opt.param_groups] == ['basic', 'pose_array']
    assert opt.param_groups[0]['lr'] == 0.004 and opt.param_groups[1]['lr'] == 0.002 and opt.param_groups[0]['eps'] == 1e-15
    assert torch.equal(opt.state[ref_params[0]]['exp_avg'].reshape(-1), f._seg(f.exp_avg, 'table'))
    assert torch.equal(opt.state[pose]['exp_avg_sq'].reshape(-1), f._seg(f.exp_avg_sq, 'pose'))
    assert float(opt.state[pose]['step']) == 37.0
    got = torch.cat([opt.state[p]['exp_avg'].reshape(-1) for p in ref_params + [pose]])
    assert torch.equal(got, f.exp_avg)                   # every moment, in the flat buffer's own order
    # ... and back: a fresh field takes parameters, moments and the step count from the reference-format file
    g = CpuField()
    g.params.zero_(), g.exp_avg.zero_(), g.exp_avg_sq.zero_()
    g.global_step = g.adam_steps = 0
    assert ck.load_reference_checkpoint(g, data) == 37   # the runner's iteration count: the caller sets its schedules from it
    assert g.adam_steps == 37
    assert torch.equal(g.params, f.params) and torch.equal(g.exp_avg, f.exp_avg) and torch.equal(g.exp_avg_sq, f.exp_avg_sq)


# ------------------------------------------------------------------------------------------------
def test_scatter_chain_rules_partition_exactly():
    """The table-gradient scatter (k_hash_bwd_agg, nof_hash.hip) emits a grid vertex once per CHAIN of consecutive runs whose
    cells contain it: corner (m, k) hands its run total on when the vertex also belongs to the next run's cell, otherwise it
    collects runs m-1, m-2, ... while their cells contain the vertex and they handed it on.  Restated here with the same
    masks / corner shifts (tools/scatter_requests.shared_corners) and checked against a direct per-vertex sum on tiles with
    face-, edge- and corner-adjacent cells, revisited cells, out-of-range lanes and 32-run window edges: every contribution
    must be emitted exactly once, whatever the adjacency pattern."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools'))
    import scatter_requests as SR
    NONE = int(SR.NONE)

    def shared(key, other):
        m = int(SR.shared_corners(np.array([key], np.int64), np.array([other], np.int64))[0])
        d = [((other >> (10 * a)) & 1023) - ((key >> (10 * a)) & 1023) for a in range(3)]
        return m, d[0] + 2 * d[1] + 4 * d[2]                          # corner k of `key` is corner k - shift of `other`

    def corner_vertex(key, k):
        return ((key & 1023) + (k & 1), ((key >> 10) & 1023) + ((k >> 1) & 1), ((key >> 20) & 1023) + (k >> 2))

    def kernel_rules(keys, vals, window=32):
        out = {}
        runs = []                                                     # (first lane, last lane)
        for i in range(64):
            if keys[i] != NONE and (i == 0 or keys[i - 1] != keys[i]):
                j = i
                while j < 63 and keys[j + 1] == keys[i]:
                    j += 1
                runs.append((i, j))
        info = []
        for m, (h, t) in enumerate(runs):
            key = keys[h]
            knext = keys[t + 1] if t < 63 else NONE
            kp1 = keys[h - 1] if h > 0 else NONE
            kp2 = NONE
            if kp1 != NONE:
                h1 = runs[m - 1][0]
                kp2 = keys[h1 - 1] if h1 > 0 else NONE
            ho, _ = shared(key, knext) if knext != NONE else (0, 0)
            t1, s1 = shared(key, kp1) if kp1 != NONE else (0, 0)
            t2, s2 = shared(key, kp2) if kp2 != NONE else (0, 0)
            info.append([key, ho, t1, t1 & t2, s1, s2, vals[h:t + 1].sum(0)])
        for base in range(0, len(info), window):
            W = [list(x) for x in info[base:base + window]]
            W[-1][1] = 0                                              # a window edge breaks the chains
            W[0][2] = W[0][3] = 0
            if len(W) > 1:
                W[1][3] = 0
            for m, (key, ho, t1, t2, s1, s2, tot) in enumerate(W):
                for k in range(8):
                    if (ho >> k) & 1:
                        continue
                    acc = tot[k]
                    if (t1 >> k) & 1:
                        acc += W[m - 1][6][k - s1]
                        if (t2 >> k) & 1:
                            acc += W[m - 2][6][k - s2]
                            v = corner_vertex(key, k)
                            for j in range(m - 3, -1, -1):
                                c = corner_vertex(W[j][0], 0)
                                d = [v[a] - c[a] for a in range(3)]
                                if not all(0 <= x < 2 for x in d):
                                    break
                                kk = d[0] | d[1] << 1 | d[2] << 2
                                if not (W[j][1] >> kk) & 1:
                                    break
                                acc += W[j][6][kk]
                    v = corner_vertex(key, k)
                    out[v] = out.get(v, 0.0) + acc
        return out

    def direct(keys, vals):
        out = {}
        for i in range(64):
            if keys[i] != NONE:
                for k in range(8):
                    v = corner_vertex(keys[i], k)
                    out[v] = out.get(v, 0.0) + vals[i, k]
        return out

    rng = np.random.default_rng(5)
    for trial in range(60):
        # a walk through cells: stay, step to a face / edge / corner neighbour, jump, or leave the box for a lane
        c = rng.integers(2, 60, 3)
        keys = []
        p_stay = [0.0, 0.5, 0.8][trial % 3]                           # runs of 1 (two windows), ~2 and ~5 lanes
        for lane in range(64):
            u = rng.random()
            if u < 0.04:
                keys.append(NONE)
                continue
            if u > p_stay + 0.04:
                if rng.random() < 0.1:
                    c = rng.integers(2, 60, 3)
                else:
                    c = c + rng.integers(-1, 2, 3)
            keys.append(int(c[0]) | int(c[1]) << 10 | int(c[2]) << 20)
        vals = rng.normal(size=(64, 8))
        a, b = kernel_rules(keys, vals), direct(keys, vals)
        assert set(a) == set(b)
        assert max(abs(a[v] - b[v]) for v in b) < 1e-9


def test_scatter_request_counter_zero_gradient_rules():
    """tools/scatter_requests.py (bench.py's roofline.atomic): a mask of all-True gradients changes nothing, all-False emits nothing,
    and zeroing whole rays removes exactly those rays' requests (rays = multiples of 64 samples here, so tiles do not mix)."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools'))
    import scatter_requests as SR
    from oracle import nof_oracle as O
    geo = O.HashGeometry(16, 2, 16, 19, 256)
    rng = np.random.default_rng(3)
    n_rays, S = 24, 192
    o = rng.uniform(-0.8, 0.8, size=(n_rays, 1, 3))
    d = rng.normal(size=(n_rays, 1, 3))
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    pts = (o + d * np.linspace(0, 0.7, S)[None, :, None]).reshape(-1, 3).astype(np.float32)
    N = len(pts)
    args = (pts, geo.scale, geo.resolution, geo.offsets[:-1], geo.size, geo.hashed, range(1, 16))
    base = SR.count_requests(*args)
    assert SR.count_requests(*args, nonzero=np.ones((16, N), bool)) == base
    assert sum(SR.count_requests(*args, nonzero=np.zeros((16, N), bool)).values()) == 0
    keep = np.ones((n_rays, S), bool)
    keep[::2] = False                                                    # every other ray has no gradient at all
    half = SR.count_requests(*args, nonzero=np.broadcast_to(keep.reshape(-1), (16, N)))
    only = SR.count_requests(pts.reshape(n_rays, S, 3)[1::2].reshape(-1, 3), *args[1:])
    assert half == only
    assert 0 < sum(half.values()) < sum(base.values())


def test_row_bitmap_round_trip():
    """GradSync mode 'rows' (bundlesdf_amd/dist.py): the bitmap of a table gradient's non-zero rows and its inverse -- rows with one
    zero entry, NaN / inf rows (a non-finite gradient must travel: the ranks agree on the skipped step from the SUMMED gradient), sizes
    that are not a multiple of 8."""
    import torch
    from bundlesdf_amd.dist import GradSync
    for n in (1, 7, 8, 9, 64, 1003):
        g = torch.Generator().manual_seed(n)
        rows = torch.zeros(n, 2)
        hit = torch.randperm(n, generator=g)[:max(1, n // 3)]
        rows[hit] = torch.randn(hit.numel(), 2, generator=g)
        rows[hit[0], 1] = 0.0
        if hit.numel() >= 3:
            rows[hit[1]] = torch.tensor([float('nan'), 0.0])
            rows[hit[2]] = torch.tensor([0.0, float('inf')])
        bits = GradSync._row_bitmap(rows)
        assert bits.dtype == torch.uint8 and bits.numel() == (n + 7) // 8
        idx = GradSync._bitmap_rows(bits, n)
        want = torch.nonzero((rows != 0).any(1)).reshape(-1)
        assert torch.equal(idx, want)
        assert set(want.tolist()) <= set(hit.tolist())


def test_operand_image_backward_position_is_the_inverse_of_the_packing_map():
    """nof_adam_step_tail (csrc/nof_adam_tail_dev.h) updates a weight where the FORWARD operand image holds it and stores the new value
    at its place in the BACKWARD image as well, through the closed-form inverse of k_mlp_pack's second mapping (csrc/nof_mlp.hip):
    for the fw element (pair (p, q), register r, lane (hi, i)) that is the bw element at pair (q, p), register r' and lane (hi', i')
    with nloc(hi', r') = i and i' = nloc(hi, r).  Restated here in Python for every network shape the library accepts: both images
    hold every weight exactly once, and the closed form lands on the bw element that holds the same weight."""
    import itertools

    def nloc(hi, r):
        return (r & 3) + 8 * (r >> 2) + 4 * hi

    def check(n_sigma, n_color, hidden, in_feat, n_view, geo):
        nl = n_sigma + n_color
        in_dim, out_dim = [], []
        for l in range(nl):
            in_dim.append(in_feat if l == 0 else (n_view + geo if l == n_sigma else hidden))
            out_dim.append(1 + geo if l == n_sigma - 1 else (3 if l == nl - 1 else hidden))
        qn = lambda l: 1 if l == 0 else (2 if l == n_sigma else hidden // 32)
        pn = lambda l: (out_dim[l] + 31) // 32

        def inmap(l, q, hi, r):                                   # csrc/nof_mlp_dev.h:inmap
            if l == 0:
                c = 16 * hi + r
                return c if c < in_feat else -1
            if l == n_sigma:
                if q == 0:
                    o = nloc(hi, r)
                    return n_view + o - 1 if 1 <= o <= geo else -1
                u = 8 * hi + r
                return u if (r < 8 and u < n_view) else -1
            c = 32 * q + nloc(hi, r)
            return c if c < in_dim[l] else -1
        for l in range(nl):
            fw, bw = {}, {}
            for pair, r, lane in itertools.product(range(pn(l) * qn(l)), range(16), range(64)):
                hi, i = lane >> 5, lane & 31
                p, q = pair // qn(l), pair % qn(l)
                row, col = 32 * p + i, inmap(l, q, hi, r)
                if row < out_dim[l] and col >= 0:
                    fw[(p * qn(l) + q, r, lane)] = (row, col)
                q2, p2 = pair // pn(l), pair % pn(l)
                row2, col2 = 32 * p2 + nloc(hi, r), inmap(l, q2, (i >> 2) & 1, (i & 3) + 4 * (i >> 3))
                if row2 < out_dim[l] and col2 >= 0:
                    bw[(q2 * pn(l) + p2, r, lane)] = (row2, col2)
            every = sorted(itertools.product(range(out_dim[l]), range(in_dim[l])))
            assert sorted(fw.values()) == every and sorted(bw.values()) == every, (n_sigma, n_color, hidden, l)
            where = {v: k for k, v in bw.items()}
            for (pr, r, lane), rc in fw.items():
                hi, i = lane >> 5, lane & 31
                p, q = pr // qn(l), pr % qn(l)
                assert where[rc] == (q * pn(l) + p, (i & 3) + 4 * (i >> 3), 32 * ((i >> 2) & 1) + nloc(hi, r)), (l, pr, r, lane)

    for shape in ((2, 3, 64, 32, 9, 15), (3, 2, 64, 32, 9, 15), (2, 2, 64, 32, 11, 15), (3, 3, 64, 24, 9, 15), (4, 4, 128, 32, 9, 15),
                  (2, 3, 128, 32, 11, 15)):
        check(*shape)
