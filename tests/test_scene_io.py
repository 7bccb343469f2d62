"""scene bounds / point-cloud fusion (tool.py) and the on-disk formats (data_reader.py, the tracker's output
directory) without open3d / cv2: bundlesdf_amd/scene.py against a reference-driven run of tool.py
(tests/golden/make_golden_scene.py), bundlesdf_amd/data_reader.py by round trips through the reference's layouts."""
import os

import numpy as np
import pytest

from bundlesdf_amd import data_reader as DR
from bundlesdf_amd import scene

G = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'scene_vectors.npz'))


def test_depth2xyzmap_and_worker_match_the_reference():
    assert np.array_equal(scene.depth2xyzmap(G['depths'][0], G['K']), G['xyz0'])
    pts, colors = scene.compute_scene_bounds_worker(None, G['K'], G['glcams'][1], True, G['rgbs'][1], G['depths'][1], G['masks'][1])
    assert np.array_equal(pts, G['worker_pts']) and np.array_equal(colors, G['worker_colors'])
    empty = scene.compute_scene_bounds_worker(None, G['K'], G['glcams'][1], True, G['rgbs'][1], G['depths'][1] * 0, G['masks'][1])
    assert empty is None                                            # tool.py:56-57


def test_translation_and_scale_match_the_reference():
    t, s, keep = scene.compute_translation_scales(G['cts_pts'], cluster=True, eps=0.06, min_samples=1)
    assert np.array_equal(t, G['cts_t']) and s == float(G['cts_s']) and np.array_equal(keep, G['cts_keep'])
    assert 300 < keep.sum() < 400                                   # the small far cluster is dropped
    t2, s2, _ = scene.compute_translation_scales(G['cts_pts'], cluster=False)
    assert np.array_equal(t2, G['cts_t_nocluster']) and s2 == float(G['cts_s_nocluster'])


def test_compute_scene_bounds_matches_the_reference(tmp_path):
    sc, tr, real, norm = scene.compute_scene_bounds(None, G['glcams'], G['K'], use_mask=True, base_dir=str(tmp_path), rgbs=G['rgbs'],
                                                    depths=G['depths'], masks=G['masks'], cluster=True, eps=0.01, min_samples=5)
    assert sc == float(G['csb_sc']) and np.array_equal(tr, G['csb_tr'])
    assert np.array_equal(real.points, G['csb_real']) and np.array_equal(norm.points, G['csb_norm'])
    assert np.array_equal(norm.colors, G['csb_norm_colors'])
    assert np.abs(norm.points).max() <= 0.9 + 1e-9                   # 0.9 of the unit cube (tool.py:37)
    for f in ('naive_fusion.ply', 'naive_fusion_biggest_cluster.ply', 'normalization.yml'):
        assert os.path.getsize(tmp_path / f) > 0
    import yaml
    n = yaml.safe_load(open(tmp_path / 'normalization.yml'))
    assert n['sc_factor'] == sc and n['translation_cvcam'] == list(tr)
    # re-used normalisation: the keep mask is |p_n| < 1 instead of the cluster (tool.py:108-113; bundlesdf.py:696-705)
    sc3, tr3, _, norm3 = scene.compute_scene_bounds(None, G['glcams'], G['K'], use_mask=True, base_dir=None, rgbs=G['rgbs'],
                                                    depths=G['depths'], masks=G['masks'], cluster=True, eps=0.01, min_samples=5,
                                                    sc_factor=float(G['csb_reuse_sc']), translation_cvcam=G['csb_reuse_tr'])
    assert sc3 == float(G['csb_reuse_sc']) and np.array_equal(norm3.points, G['csb_reuse_norm'])
    assert len(norm3.points) < len(norm.points) and np.abs(norm3.points).max() < 1


def test_point_cloud_filters():
    """the two open3d operations as documented: voxel means on the grid anchored at min_bound - v/2, statistical outlier removal"""
    pts = np.array([[0.0, 0, 0], [0.004, 0, 0], [0.011, 0, 0], [0.5, 0.5, 0.5]])
    cols = np.array([[1.0, 0, 0], [0, 1, 0], [0, 0, 1], [1, 1, 1]])
    d = scene.PointCloud(pts, cols).voxel_down_sample(0.01)
    # origin = -0.005: voxels [-.005,.005) and [.005,.015) along x
    assert len(d.points) == 3
    i = np.argsort(d.points[:, 0])
    assert np.allclose(d.points[i][0], [0.002, 0, 0]) and np.allclose(d.colors[i][0], [0.5, 0.5, 0])
    rng = np.random.default_rng(0)
    blob = rng.normal(size=(300, 3)) * 0.01
    far = np.array([[1.0, 1.0, 1.0], [-1.0, 0.5, 0.2]])
    kept, ind = scene.PointCloud(np.concatenate([blob, far])).remove_statistical_outlier(nb_neighbors=30, std_ratio=2.0)
    assert 300 not in ind and 301 not in ind and len(ind) >= 280
    assert np.array_equal(kept.points, np.concatenate([blob, far])[ind])


def test_resize_follows_cv2_inter_nearest():
    img = np.arange(6 * 8).reshape(6, 8)
    small = DR.resize_nearest(img, 4, 3)                             # (W, H) like cv2.resize
    assert small.shape == (3, 4) and np.array_equal(small, img[[0, 2, 4]][:, [0, 2, 4, 6]])
    big = DR.resize_nearest(img, 16, 12)
    assert np.array_equal(big[::2, ::2], img) and np.array_equal(big[1::2, 1::2], img)
    assert DR.resize_nearest(img, 8, 6) is img


def test_capture_directory_round_trip(tmp_path):
    """write_capture -> YcbineoatReader (BundleTrack/scripts/data_reader.py:21-105): ids, intrinsics, colour, uint16-mm depth,
    masks (3-channel masks collapse to one), hand masks, ground-truth poses, downscale"""
    rng = np.random.default_rng(1)
    F, H, W = 3, 24, 32
    rgbs = rng.integers(0, 255, size=(F, H, W, 3), dtype=np.uint8)
    depths = np.round(rng.uniform(0.2, 1.5, size=(F, H, W)), 3)
    depths[:, :3] = 0
    masks = rng.random((F, H, W)) > 0.5
    K = np.array([[100.0, 0, 16], [0, 100.0, 12], [0, 0, 1]])
    poses = [np.eye(4) * (i + 1) for i in range(F)]
    vd = str(tmp_path / 'milk')
    ids = DR.write_capture(vd, rgbs, depths, masks, K, poses=poses)
    os.makedirs(f'{vd}/masks_hand')
    hand = np.zeros((H, W), np.uint8)
    hand[5:9, 5:9] = 255
    DR.write_png(f'{vd}/masks_hand/{ids[1]}.png', hand)
    DR.write_png(f'{vd}/masks/{ids[2]}.png', np.repeat((masks[2] * 255).astype(np.uint8)[..., None], 3, -1))     # RGB mask
    r = DR.YcbineoatReader(vd)
    assert len(r) == F and r.id_strs == ids and (r.H, r.W) == (H, W) and np.array_equal(r.K, K) and r.get_video_name() == 'milk'
    for i in range(F):
        assert np.array_equal(r.get_color(i), rgbs[i])
        assert np.abs(r.get_depth(i) - depths[i]).max() < 5.1e-4     # millimetre quantisation
        assert np.array_equal(r.get_mask(i) > 0, masks[i])
        assert np.array_equal(r.get_gt_pose(i), poses[i])
    assert r.get_mask(2).ndim == 2 and r.get_mask(2).dtype == np.uint8
    assert r.get_occ_mask(1)[6, 6] == 1 and r.get_occ_mask(1).sum() == 16 and r.get_occ_mask(0).sum() == 0
    assert np.array_equal(r.get_xyz_map(0), scene.depth2xyzmap(r.get_depth(0), K))
    half = DR.YcbineoatReader(vd, shorter_side=12)
    assert (half.H, half.W) == (12, 16) and np.allclose(half.K[:2], K[:2] * 0.5)
    assert np.array_equal(half.get_color(0), rgbs[0][::2, ::2])


def test_tracker_output_round_trip(tmp_path):
    """what run_global_nerf reads (bundlesdf.py:640-688): cam_K.txt, ob_in_cam/*, <last>/keyframes.yml, color_segmented /
    depth_filtered / mask PNGs"""
    rng = np.random.default_rng(2)
    F, H, W = 4, 12, 16
    rgbs = rng.integers(0, 255, size=(F, H, W, 3), dtype=np.uint8)
    depths = np.round(rng.uniform(0.2, 1.5, size=(F, H, W)), 3)
    masks = rng.random((F, H, W)) > 0.3
    K = np.array([[50.0, 0, 8], [0, 50.0, 6], [0, 0, 1]])
    cam_in_obs = np.tile(np.eye(4), (F, 1, 1))
    cam_in_obs[:, :3, 3] = rng.normal(size=(F, 3))
    dd = str(tmp_path / 'out')
    ids = DR.write_tracker_output(dd, rgbs, depths, masks, K, cam_in_obs)
    t = DR.TrackerOutput(dd)
    assert t.last_stamp == ids[-1] and t.keys == [f'keyframe_{s}' for s in ids] and np.array_equal(t.K, K)
    d = t.load()
    assert d['frame_ids'] == ids and np.allclose(d['cam_in_obs'], cam_in_obs, atol=1e-6)
    assert np.array_equal(d['rgbs'], rgbs) and np.abs(d['depths'] - depths).max() < 5.1e-4
    assert np.array_equal(d['masks'] > 0, masks)
    sel = t.select(2, rng=np.random.default_rng(0))
    assert sel[0] == t.keys[0] or t.keys[0] in sel
    assert len(set(sel)) == len(sel) <= 3 and t.select(10) == t.keys
