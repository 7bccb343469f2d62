"""Scene bounds / point-cloud fusion in front of the Neural Object Field: what the reference's
tool.py does with open3d + cv2 + joblib, here with NumPy / SciPy / scikit-learn only, so that a capture directory can be
fed to NerfRunner without open3d or cv2.

    depth2xyzmap                 Utils.py:219-231
    find_biggest_cluster         tool.py:18-25     (sklearn DBSCAN, the same call)
    compute_translation_scales   tool.py:28-39
    compute_scene_bounds_worker  tool.py:42-64
    compute_scene_bounds         tool.py:67-132

`PointCloud` carries the handful of open3d.geometry.PointCloud operations those functions and NerfRunner use
(`points`, `colors`, `voxel_down_sample`, `remove_statistical_outlier`, `transform`, `+=`).  open3d itself is third-party and
absent here; its two geometric filters are restated from its documented behaviour:
  * voxel_down_sample(v): points are binned by floor((p - (min_bound - v/2)) / v) and every occupied voxel yields the MEAN of
    its points (and of their colours);
  * remove_statistical_outlier(k, r): d_i = mean distance of point i to its k nearest neighbours (the point itself included,
    as open3d's KNN search returns it); points with d_i > mean(d) + r * std(d) (sample standard deviation) are dropped.
The glue around them (validity masks, camera convention, normalisation, keep masks) is pinned against a reference-driven run
of tool.py (tests/golden/make_golden_scene.py -> tests/golden/scene_vectors.npz).
"""
import copy
import logging
import os

import numpy as np

from .nerf_helpers import glcam_in_cvcam


class PointCloud:
    def __init__(self, points=None, colors=None):
        self.points = np.zeros((0, 3)) if points is None else np.asarray(points, dtype=np.float64).reshape(-1, 3)
        self.colors = None if colors is None else np.asarray(colors, dtype=np.float64).reshape(-1, 3)

    def __len__(self):
        return len(self.points)

    def __iadd__(self, other):
        if self.colors is not None and other.colors is not None:
            self.colors = np.concatenate([self.colors, other.colors], 0)
        else:
            self.colors = None
        self.points = np.concatenate([self.points, other.points], 0)
        return self

    def voxel_down_sample(self, voxel_size):
        if len(self.points) == 0:
            return PointCloud(self.points.copy(), None if self.colors is None else self.colors.copy())
        origin = self.points.min(axis=0) - voxel_size * 0.5
        key = np.floor((self.points - origin) / voxel_size).astype(np.int64)
        _, first, inv = np.unique(key, axis=0, return_index=True, return_inverse=True)
        inv = inv.reshape(-1)
        n = len(first)
        cnt = np.bincount(inv, minlength=n).astype(np.float64)

        def mean(a):
            return np.stack([np.bincount(inv, weights=a[:, k], minlength=n) / cnt for k in range(3)], -1)
        return PointCloud(mean(self.points), None if self.colors is None else mean(self.colors))

    def remove_statistical_outlier(self, nb_neighbors, std_ratio):
        from scipy.spatial import cKDTree
        n = len(self.points)
        if n == 0:
            return PointCloud(), np.zeros(0, dtype=np.int64)
        k = min(nb_neighbors, n)
        d, _ = cKDTree(self.points).query(self.points, k=k, workers=-1)
        d = d.reshape(n, -1)
        avg = d.mean(axis=1)
        if n < 2:
            keep = np.ones(n, dtype=bool)
        else:
            keep = avg < avg.mean() + std_ratio * avg.std(ddof=1)
        ind = np.flatnonzero(keep)
        return PointCloud(self.points[ind], None if self.colors is None else self.colors[ind]), ind

    def transform(self, tf):
        tf = np.asarray(tf, dtype=np.float64)
        self.points = self.points @ tf[:3, :3].T + tf[:3, 3]
        return self


def to_cloud(points, colors=None):
    """toOpen3dCloud (Utils.py:207-216): colours above 1 are taken as 0..255."""
    if colors is not None:
        colors = np.asarray(colors, dtype=np.float64)
        if colors.size and colors.max() > 1:
            colors = colors / 255.0
    return PointCloud(np.asarray(points, dtype=np.float64), colors)


def depth2xyzmap(depth, K):
    """Utils.py:219-231: back-projected points [H,W,3] float32 in the OpenCV camera frame, 0 where depth < 0.1."""
    invalid_mask = depth < 0.1
    H, W = depth.shape[:2]
    vs, us = np.meshgrid(np.arange(0, H), np.arange(0, W), sparse=False, indexing='ij')
    zs = depth.reshape(-1)
    xs = (us.reshape(-1) - K[0, 2]) * zs / K[0, 0]
    ys = (vs.reshape(-1) - K[1, 2]) * zs / K[1, 1]
    xyz_map = np.stack((xs, ys, zs), 1).reshape(H, W, 3).astype(np.float32)
    xyz_map[invalid_mask] = 0
    return xyz_map.astype(np.float32)


def find_biggest_cluster(pts, eps=0.06, min_samples=1):
    """tool.py:18-25."""
    from sklearn.cluster import DBSCAN
    dbscan = DBSCAN(eps=eps, min_samples=min_samples, n_jobs=-1)
    dbscan.fit(pts)
    ids, cnts = np.unique(dbscan.labels_, return_counts=True)
    best_id = ids[cnts.argsort()[-1]]
    keep_mask = dbscan.labels_ == best_id
    return pts[keep_mask], keep_mask


def compute_translation_scales(pts, max_dim=2, cluster=True, eps=0.06, min_samples=1):
    """tool.py:28-39: translation = -centre of the bounding box, sc_factor = 0.9 * max_dim / largest extent."""
    if cluster:
        pts, keep_mask = find_biggest_cluster(pts, eps, min_samples)
    else:
        keep_mask = np.ones((len(pts)), dtype=bool)
    max_xyz = pts.max(axis=0)
    min_xyz = pts.min(axis=0)
    center = (max_xyz + min_xyz) / 2
    sc_factor = max_dim / (max_xyz - min_xyz).max()
    sc_factor *= 0.9
    translation_cvcam = -center
    return translation_cvcam, sc_factor, keep_mask


def compute_scene_bounds_worker(color_file, K, glcam_in_world, use_mask, rgb=None, depth=None, mask=None):
    """tool.py:42-64: masked back-projection of one frame -> 1 cm voxel grid -> statistical outlier removal -> world frame."""
    if rgb is None:
        from .data_reader import read_depth_png, read_png
        rgb = read_png(color_file)[..., :3]
        depth = read_depth_png(color_file.replace('images', 'depth_filtered'))
        if use_mask and mask is None:
            mask = read_png(color_file.replace('images', 'masks'))
    xyz_map = depth2xyzmap(depth, K)
    valid = depth >= 0.1
    if use_mask:
        valid = valid & (mask > 0)
    pts = xyz_map[valid].reshape(-1, 3)
    if len(pts) == 0:
        return None
    colors = rgb[valid].reshape(-1, 3)
    pcd = to_cloud(pts, colors)
    pcd = pcd.voxel_down_sample(0.01)
    pcd, ind = pcd.remove_statistical_outlier(nb_neighbors=30, std_ratio=2.0)
    cam_in_world = glcam_in_world @ glcam_in_cvcam
    pcd.transform(cam_in_world)
    return pcd.points.copy(), pcd.colors.copy()


def make_normalisation(translation_cvcam, sc_factor):
    """p_n = (p + translation) * sc_factor as a 4x4 (tool.py:98-104)."""
    tf = np.eye(4)
    tf[:3, 3] = translation_cvcam
    tf1 = np.eye(4)
    tf1[:3, :3] *= sc_factor
    return tf1 @ tf


def compute_scene_bounds(color_files, glcam_in_worlds, K, use_mask=True, base_dir=None, rgbs=None, depths=None, masks=None,
                         cluster=True, translation_cvcam=None, sc_factor=None, eps=0.06, min_samples=1, write_files=True):
    """tool.py:67-132.  Returns (sc_factor, translation_cvcam, pcd_real_scale, pcd_normalised); writes naive_fusion.ply,
    naive_fusion_biggest_cluster.ply and normalization.yml into base_dir like the reference (write_files=False skips that)."""
    assert color_files is None or rgbs is None
    if base_dir is None and color_files is not None:
        base_dir = os.path.dirname(color_files[0]) + '/../'
    args = []
    if rgbs is not None:
        for i in range(len(rgbs)):
            args.append((None, K, glcam_in_worlds[i], use_mask, rgbs[i], depths[i], masks[i]))
    else:
        for i in range(len(color_files)):
            args.append((color_files[i], K, glcam_in_worlds[i], use_mask))
    logging.info("compute_scene_bounds_worker start")
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=10) as ex:                     # joblib.Parallel(n_jobs=10, prefer="threads"), tool.py:82
        ret = list(ex.map(lambda a: compute_scene_bounds_worker(*a), args))
    logging.info("compute_scene_bounds_worker done")
    pcd_all = None
    for r in ret:
        if r is None:
            continue
        if pcd_all is None:
            pcd_all = to_cloud(r[0], r[1])
        else:
            pcd_all += to_cloud(r[0], r[1])
    pcd = pcd_all.voxel_down_sample(eps / 5)
    logging.info("merge pcd")
    write = write_files and base_dir is not None
    if write:
        os.makedirs(base_dir, exist_ok=True)
        write_ply(f'{base_dir}/naive_fusion.ply', pcd)
    pts = pcd.points.copy()
    if translation_cvcam is None:
        translation_cvcam, sc_factor, keep_mask = compute_translation_scales(pts, cluster=cluster, eps=eps, min_samples=min_samples)
        tf = make_normalisation(translation_cvcam, sc_factor)
    else:
        tf = make_normalisation(translation_cvcam, sc_factor)
        tmp = copy.deepcopy(pcd)
        tmp.transform(tf)
        keep_mask = (np.abs(tmp.points) < 1).all(axis=-1)
    logging.info("compute_translation_scales done")
    pcd = to_cloud(pts[keep_mask], pcd.colors[keep_mask])
    if write:
        write_ply(f"{base_dir}/naive_fusion_biggest_cluster.ply", pcd)
    pcd_real_scale = copy.deepcopy(pcd)
    print(f'translation_cvcam={translation_cvcam}, sc_factor={sc_factor}')
    if write:
        import yaml
        with open(f'{base_dir}/normalization.yml', 'w') as ff:
            yaml.safe_dump({'translation_cvcam': np.asarray(translation_cvcam).tolist(), 'sc_factor': float(sc_factor)}, ff)
    pcd.transform(tf)
    return sc_factor, translation_cvcam, pcd_real_scale, pcd


def write_ply(path, pcd):
    """ASCII PLY of a point cloud (what o3d.io.write_point_cloud leaves for inspection)."""
    n = len(pcd.points)
    has_c = pcd.colors is not None and len(pcd.colors) == n
    with open(path, 'w') as f:
        f.write('ply\nformat ascii 1.0\n')
        f.write(f'element vertex {n}\nproperty double x\nproperty double y\nproperty double z\n')
        if has_c:
            f.write('property uchar red\nproperty uchar green\nproperty uchar blue\n')
        f.write('end_header\n')
        c = (np.clip(pcd.colors, 0, 1) * 255).astype(np.uint8) if has_c else None
        for i in range(n):
            p = pcd.points[i]
            f.write(f'{p[0]:.9g} {p[1]:.9g} {p[2]:.9g}' + (f' {c[i, 0]} {c[i, 1]} {c[i, 2]}' if has_c else '') + '\n')
