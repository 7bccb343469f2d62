"""End-to-end on the MI355X through the reference's plugin surface (the call sequence of bundlesdf.py:run_nerf,
:219-241): NerfRunner(...) -> train() -> get_optimized_poses_in_real_world -> extract_mesh -> mesh_to_real_world ->
add_new_frames(...) -> train().  The data set is the synthetic ellipsoid pool, so the mesh is checked by the Chamfer
distance to the analytic surface (stand-in for the 'milk' sequence, whose data is not available offline)."""
import numpy as np
import pytest
import torch
from scipy.spatial import cKDTree

pytestmark = pytest.mark.gpu


def _ellipsoid_points(semi, n=20000, seed=0):
    rng = np.random.default_rng(seed)
    p = rng.normal(size=(n, 3))
    p /= np.linalg.norm(p, axis=1, keepdims=True)
    return p * semi


def chamfer(a, b):
    """Utils.py:268-273 (mutual mean nearest-neighbour distance)"""
    d1, _ = cKDTree(a).query(b)
    d2, _ = cKDTree(b).query(a)
    print(f'chamfer: gt->mesh mean {d1.mean() * 1e3:.2f} mm p95 {np.percentile(d1, 95) * 1e3:.2f} max {d1.max() * 1e3:.2f} | '
          f'mesh->gt mean {d2.mean() * 1e3:.2f} mm p95 {np.percentile(d2, 95) * 1e3:.2f} max {d2.max() * 1e3:.2f}')
    return 0.5 * (d1.mean() + d2.mean())


def _surface_samples(mesh, n=20000):
    v, f = np.asarray(mesh.vertices), np.asarray(mesh.faces)
    a, b, c = v[f[:, 0]], v[f[:, 1]], v[f[:, 2]]
    area = 0.5 * np.linalg.norm(np.cross(b - a, c - a), axis=1)
    rng = np.random.default_rng(0)
    idx = rng.choice(len(area), size=n, p=area / area.sum())
    r1, r2 = np.sqrt(rng.random(n)), rng.random(n)
    return (1 - r1)[:, None] * a[idx] + (r1 * (1 - r2))[:, None] * b[idx] + (r1 * r2)[:, None] * c[idx]


@pytest.mark.parametrize("precision", ['fp16', 'bf16'])
def test_runner_surface_end_to_end(nof, precision):
    from bundlesdf_amd import synthetic
    from bundlesdf_amd.config import default_cfg
    from bundlesdf_amd.nerf_runner import NerfRunner, get_optimized_poses_in_real_world, mesh_to_real_world
    pool = synthetic.make_pool(n_frames=10, H=240, W=320, fx=300.0, seed=1)
    cfg = default_cfg(n_step=600, N_rand=2048, num_levels=16, log2_hashmap_size=17, finest_res=256, far=1.0,
                      sc_factor=pool['sc_factor'], translation=pool['translation'], frame_features=2)
    n0 = 8
    runner = NerfRunner(cfg, pool['rgbs'][:n0], depths=pool['depths'][:n0], masks=pool['masks'][:n0], normal_maps=None,
                        poses=pool['poses'][:n0].copy(), K=pool['K'], build_octree_pcd=synthetic.PointCloud(pool['pcd_normalized']),
                        precision=precision)
    assert runner.rays.shape[1] == 12 and runner.rays.shape[0] > 20000
    assert (runner.rays[:, 9] == 0).all() and (runner.rays[:, 8] < n0).all()
    runner.train_loop()
    first = runner.field.losses()['loss']
    runner.global_step += 1
    runner.train()
    last = runner.field.losses()
    print('runner losses', first, last)
    assert np.isfinite(last['loss']) and last['loss'] < 0.8 * first and last['sdf_loss'] < 1.0, (first, last)
    assert int(runner.field.flags[0].item()) == 0

    poses_opt, offset = get_optimized_poses_in_real_world(pool['poses'][:n0].copy(), runner.models['pose_array'],
                                                          cfg['sc_factor'], cfg['translation'])
    assert poses_opt.shape == (n0, 4, 4) and np.isfinite(poses_opt).all()
    # frame 0 is the anchor: its optimised real-world pose equals the input pose (converted back to OpenCV)
    cam0 = pool['poses_gt'][0] @ np.diag([1.0, -1.0, -1.0, 1.0])
    assert np.abs(poses_opt[0] - cam0).max() < 1e-4
    # pose noise of +-5 mm / 2 deg was injected on frames > 0: optimisation must not make things worse on average
    gt = pool['poses_gt'][:n0] @ np.diag([1.0, -1.0, -1.0, 1.0])
    noisy = pool['poses'][:n0].copy()
    noisy[:, :3, 3] = noisy[:, :3, 3] / cfg['sc_factor'] - cfg['translation']
    noisy = noisy @ np.diag([1.0, -1.0, -1.0, 1.0])
    err_before = np.linalg.norm(noisy[1:, :3, 3] - gt[1:, :3, 3], axis=1).mean()
    err_after = np.linalg.norm(poses_opt[1:, :3, 3] - gt[1:, :3, 3], axis=1).mean()
    print('pose translation error before/after (m):', err_before, err_after)
    # corrections are bounded by max_trans / max_rot (tanh parametrisation, nerf_helpers.py:147-149)
    delta = runner.models['pose_array'].get_matrices(np.arange(n0)).cpu().numpy()
    assert np.abs(delta[:, :3, 3]).max() <= cfg['max_trans'] * cfg['sc_factor'] * 1.8 and np.isfinite(delta).all()
    assert np.abs(delta[0] - np.eye(4)).max() == 0

    mesh = runner.extract_mesh(isolevel=0, voxel_size=0.004)
    assert mesh is not None and len(mesh.vertices) > 500 and len(mesh.faces) > 1000
    mesh = mesh_to_real_world(mesh, pose_offset=offset, translation=cfg['translation'], sc_factor=cfg['sc_factor'])
    from bundlesdf_amd.mesh import largest_component
    chamfer(_surface_samples(mesh), _ellipsoid_points(pool['semi_axes']))          # raw mesh (printed for the record)
    mesh = largest_component(mesh)                                                 # bundlesdf.py:748-760
    cd = chamfer(_surface_samples(mesh), _ellipsoid_points(pool['semi_axes']))
    print(f'Chamfer distance {cd * 100:.3f} cm, V={len(mesh.vertices)} F={len(mesh.faces)}')
    assert cd < 0.003, f'Chamfer distance {cd * 100:.3f} cm'         # 3 mm at 4 mm voxels, 1 mm depth noise, 8 views

    # growing the pool: images of the NEW frames, poses of ALL frames (bundlesdf.py:223)
    runner.add_new_frames(pool['rgbs'][n0:], pool['depths'][n0:], pool['masks'][n0:], None, pool['poses'].copy(),
                          new_pcd=synthetic.PointCloud(pool['pcd_normalized']), reuse_weights=False)
    assert runner.field.F == 10 and (runner.rays[:, 8].max() == 9)
    runner.cfg['n_step'] = 150
    runner.N_iters = 151
    runner.train()
    assert np.isfinite(runner.field.losses()['loss'])
    m2 = runner.extract_mesh(isolevel=0, voxel_size=0.006, return_sigma=True)
    assert m2 is not None and m2[1].ndim == 3


def test_add_new_frames_on_a_sharded_pool(nof):
    """add_new_frames under keyframe-sharded data parallel (nerf_runner.py:352-433 has no such mode: SURVEY 8e): two ranks'
    runners on the one GPU (no process group needed: the ray tables are what is checked).  Every rank is handed the same new
    frames and keeps its share; rays stay addressed by GLOBAL frame id, and the ranks' tables together are exactly the table of
    one process that owns every frame."""
    from bundlesdf_amd import synthetic
    from bundlesdf_amd.config import default_cfg
    from bundlesdf_amd.nerf_runner import NerfRunner
    pool = synthetic.make_pool(n_frames=9, H=120, W=160, fx=150.0, seed=2)
    n0, world = 4, 2

    def make(rank, world_size):
        cfg = default_cfg(n_step=10, N_rand=512, num_levels=8, log2_hashmap_size=14, finest_res=128, far=1.0,
                          sc_factor=pool['sc_factor'], translation=pool['translation'], frame_features=2)
        lo, hi = (0, n0) if world_size == 1 else (rank * 2, rank * 2 + 2)
        return NerfRunner(cfg, pool['rgbs'][lo:hi], depths=pool['depths'][lo:hi], masks=pool['masks'][lo:hi], normal_maps=None,
                          poses=pool['poses'][:n0].copy(), K=pool['K'], build_octree_pcd=synthetic.PointCloud(pool['pcd_normalized']),
                          precision='fp16x3', world_size=world_size, rank=rank, frame_offset=lo)
    single = make(0, 1)
    ranks = [make(r, world) for r in range(world)]
    feat_before = [r.field.feat.clone() for r in ranks]
    new = slice(n0, 9)
    for r in [single] + ranks:
        r.add_new_frames(pool['rgbs'][new], pool['depths'][new], pool['masks'][new], None, pool['poses'].copy(),
                         new_pcd=synthetic.PointCloud(pool['pcd_normalized']), reuse_weights=True)
    assert list(ranks[0].frame_ids) == [0, 1, 4, 5, 6] and list(ranks[1].frame_ids) == [2, 3, 7, 8]
    assert all(r.field.F == 9 for r in ranks) and single.field.F == 9
    for r, fb in zip(ranks, feat_before):                                  # reuse_weights: the old GLOBAL rows keep their features
        assert torch.equal(r.field.feat.view(9, 2)[:n0], fb.view(n0, 2))
    rows = torch.cat([r.rays for r in ranks]).cpu().numpy()
    want = single.rays.cpu().numpy()
    assert rows.shape == want.shape
    key = lambda a: a[np.lexsort(a.T[::-1])]
    assert np.array_equal(key(rows), key(want))                            # the same rays, each on exactly one rank
    for r in ranks:                                                        # and a step runs on the grown shard
        r.train_loop()
        assert np.isfinite(r.field.losses()['loss']) and int(r.field.flags[0].item()) == 0
    with pytest.raises(ValueError):
        ranks[0].add_new_frames(pool['rgbs'][new], pool['depths'][new], pool['masks'][new], None, pool['poses'].copy(),
                                new_pcd=synthetic.PointCloud(pool['pcd_normalized']))


def test_checkpoint_roundtrip(nof, tmp_path):
    from tests.test_gpu_step import _pair
    from tests import util as U
    cfg, fld, orc, batch, rng = _pair(nof, 'bf16', R=128)
    pool = U.dev(batch)
    for _ in range(3):
        fld.train_step(pool, None, 128, seed=1)
    snap = {k: getattr(fld, k).clone() for k in ('params', 'exp_avg', 'exp_avg_sq')}
    step, adam = fld.global_step, fld.adam_steps
    fld.train_step(pool, None, 128, seed=1)
    after = fld.params.clone()
    for k, v in snap.items():
        getattr(fld, k).copy_(v)
    fld.global_step, fld.adam_steps = step, adam
    fld.grads.zero_()
    fld.train_step(pool, None, 128, seed=1)
    torch.cuda.synchronize()
    # same state + same Philox (seed, step) -> same parameters up to atomic summation order
    assert (fld.params - after).abs().max().item() < 2e-2 and ((fld.params - after).abs() > 1e-4).float().mean().item() < 1e-2


def test_reference_format_checkpoint_roundtrip(nof, tmp_path):
    """save_weights(reference_format=True) writes the reference's layout ('model' / 'embed_fn' / 'pose_array' /
    'feature_array' / 'optimizer' state_dicts, nerf_runner.py:546-566); load_weights reads it back into a fresh field: same
    parameters, same Adam moments and step count."""
    from tests.test_gpu_step import _pair
    from tests import util as U
    from bundlesdf_amd.checkpoint import to_reference_checkpoint, load_reference_checkpoint
    cfg, fld, orc, batch, rng = _pair(nof, 'bf16', ff=2, R=128)
    pool = U.dev(batch)
    for _ in range(3):
        fld.train_step(pool, None, 128, seed=1)
    ck = to_reference_checkpoint(fld, 3)
    p = str(tmp_path / 'ref_format.pth')
    torch.save(ck, p)
    ck2 = torch.load(p)
    assert set(ck2) >= {'global_step', 'model', 'embed_fn', 'pose_array', 'feature_array', 'optimizer'}
    assert ck2['embed_fn']['embeddings'].shape == (fld.n_entries, 2) and ck2['pose_array']['data'].shape == (fld.F, 6)
    before, m, v = fld.params.clone(), fld.exp_avg.clone(), fld.exp_avg_sq.clone()
    fld.params.zero_()
    fld.global_step = fld.adam_steps = 0
    assert load_reference_checkpoint(fld, ck2) == 3
    torch.cuda.synchronize()
    assert torch.equal(fld.params, before)
    # the Adam moments come back with THEIR step count (bias correction); the schedule step is the caller's (ck['global_step'])
    assert torch.equal(fld.exp_avg, m) and torch.equal(fld.exp_avg_sq, v) and fld.adam_steps == 3 and float(m.abs().sum()) > 0
    # a checkpoint without optimiser state: parameters at iteration 3, moments of age 0 (a bias correction for step 4 on
    # zero moments would shrink the first updates by 1 - beta^4)
    ck3 = {k: v for k, v in ck2.items() if k != 'optimizer'}
    ck3['embed_fn'] = ck['embed_fn']
    assert load_reference_checkpoint(fld, ck3) == 3 and fld.adam_steps == 0 and float(fld.exp_avg.abs().sum()) == 0
    ck2['embed_fn']['embeddings'] = ck2['embed_fn']['embeddings'][:-8]
    with pytest.raises(ValueError):
        load_reference_checkpoint(fld, ck2)


def test_global_refine_from_a_tracker_directory(nof, tmp_path):
    """SURVEY.md 8f ranks 3-4 end to end: synthetic keyframes laid out as the tracker's output directory (keyframes.yml,
    color_segmented / depth_filtered (uint16 mm) / mask PNGs, cam_K.txt, ob_in_cam) -> bundlesdf_amd.global_refine.run_global_nerf
    (scene bounds + fusion, preprocess, NerfRunner.train, pose hand-back, extract_mesh, biggest component, texture bake, back to
    the real world) -> textured_mesh.obj; checked by the Chamfer distance to the analytic surface."""
    from bundlesdf_amd import synthetic
    from bundlesdf_amd.config import default_cfg
    from bundlesdf_amd.data_reader import write_tracker_output
    from bundlesdf_amd.global_refine import run_global_nerf
    rng = np.random.default_rng(3)
    F, H, W = 10, 240, 320
    K = np.array([[300.0, 0, 160.0], [0, 300.0, 120.0], [0, 0, 1]])
    cams = synthetic.fibonacci_sphere(F, 0.6)
    rgbs, depths, masks, cam_in_obs = [], [], [], []
    for i in range(F):
        T = synthetic.look_at_cv(cams[i])
        rgb, depth, mask = synthetic.render_frame(T, K, H, W, rng)
        rgbs.append(rgb.astype(np.uint8)), depths.append(depth), masks.append(mask), cam_in_obs.append(T)
    dd = str(tmp_path / 'track_out')
    write_tracker_output(dd, rgbs, depths, masks, K, np.array(cam_in_obs))
    cfg = default_cfg(n_step=500, N_rand=2048, num_levels=16, log2_hashmap_size=17, finest_res=256, far=1.0, frame_features=2,
                      mesh_resolution=0.004, n_train_image=500)
    out = run_global_nerf(dd, cfg, get_texture=True, tex_res=1024)
    assert out['textured'] is True
    mesh = out['mesh']
    import os
    for f in ('final/nerf/config.yml', 'final/nerf/normalization.yml', 'final/nerf/naive_fusion_biggest_cluster.ply',
              'final/nerf/trainval_poses.txt', 'mesh_cleaned.obj', 'textured_mesh.obj', 'textured_mesh.mtl', 'textured_mesh.png'):
        assert os.path.getsize(f'{dd}/{f}') > 0, f
    assert 0.5 < cfg['sc_factor'] * 0.24 / 2 / 0.9 < 1.3             # scale from the fused cloud: largest extent 0.24 m -> 0.9 of [-1,1]
    assert mesh.uv is not None and mesh.texture.shape == (1024, 1024, 3)
    cd = chamfer(_surface_samples(mesh), _ellipsoid_points(synthetic.SEMI_AXES))
    print(f'global refine from the tracker directory: Chamfer {cd * 1e3:.2f} mm, V={len(mesh.vertices)} F={len(mesh.faces)}, '
          f'texture coverage {mesh.texture_coverage:.3f}')
    assert cd < 0.003
    assert out['optimized_cvcam_in_obs'].shape == (F, 4, 4)
    assert np.abs(out['optimized_cvcam_in_obs'][0] - np.array(cam_in_obs)[0]).max() < 1e-3     # the anchor frame is handed back unchanged
