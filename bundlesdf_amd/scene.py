"""Scene bounds and point-cloud fusion in front of the Neural Object Field, with NumPy / SciPy /
scikit-learn only, so that a capture directory reaches NerfRunner without open3d or cv2.

What has to come out (the behaviour of the reference's tool.py:18-132, pinned bit for bit by tests/golden/scene_vectors.npz, which
a reference-driven run of tool.py produced -- tests/golden/make_golden_scene.py):

  * every keyframe's masked depth is back-projected (Utils.py:219-231), thinned on a 1 cm voxel grid, cleaned of statistical
    outliers (30 neighbours, 2 sigma) and moved to the world frame through its OpenGL camera pose;
  * the frames' clouds are concatenated in frame order and thinned again on a grid of eps / 5;
  * the normalisation p_n = (p + translation) * sc_factor puts the biggest DBSCAN cluster of that cloud into 0.9 of the cube
    [-1, 1]^3 (or a given normalisation is re-used and the cloud is cut to the open cube);
  * naive_fusion.ply, naive_fusion_biggest_cluster.ply and normalization.yml are left in the base directory.

Layout of this module (its own, not the reference's): `PointCloud` is the small stand-in for the open3d cloud type the callers
see; `back_project` / `frame_cloud` build one frame's cloud; `Normalisation` owns translation, scale and their 4x4; `FusedScene`
collects frames and derives bounds.  The functions named like the reference's (`compute_scene_bounds`, ...) are the plugin
surface bundlesdf.py calls (bundlesdf.py:148-170,696-705) and only assemble those pieces.

open3d itself is third-party and absent here; its two geometric filters are restated from its documented behaviour:
  * voxel_down_sample(v): points are binned by floor((p - (min_bound - v/2)) / v) and every occupied voxel yields the MEAN of
    its points (and of their colours);
  * remove_statistical_outlier(k, r): d_i = mean distance of point i to its k nearest neighbours (the point itself included,
    as open3d's KNN search returns it); points with d_i > mean(d) + r * std(d) (sample standard deviation) are dropped.
"""
import logging
import os
from concurrent.futures import ThreadPoolExecutor
from dataclasses import dataclass

import numpy as np

from .nerf_helpers import glcam_in_cvcam

FRAME_VOXEL = 0.01          # per-frame thinning grid [m]
OUTLIER_NEIGHBOURS = 30
OUTLIER_SIGMA = 2.0
MIN_DEPTH = 0.1             # [m] below this a depth pixel is invalid (Utils.py:221)
CUBE_FILL = 0.9             # the biggest cluster spans this fraction of [-1, 1]


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

    def clone(self):
        return PointCloud(self.points.copy(), None if self.colors is None else self.colors.copy())

    def select(self, keep):
        return PointCloud(self.points[keep], None if self.colors is None else self.colors[keep])

    def voxel_down_sample(self, voxel_size):
        if len(self.points) == 0:
            return self.clone()
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
        return self.select(ind), ind

    def transform(self, tf):
        tf = np.asarray(tf, dtype=np.float64)
        self.points = self.points @ tf[:3, :3].T + tf[:3, 3]
        return self


def to_cloud(points, colors=None):
    """A cloud from raw arrays; colours given in 0..255 are brought to 0..1 (what the reference's toOpen3dCloud does)."""
    if colors is not None:
        colors = np.asarray(colors, dtype=np.float64)
        if colors.size and colors.max() > 1:
            colors = colors / 255.0
    return PointCloud(np.asarray(points, dtype=np.float64), colors)


# ---- one frame ------------------------------------------------------------------------------------------------------------
def back_project(depth, K):
    """Pinhole back-projection of a depth image: [H,W,3] float32 points in the OpenCV camera frame, zero where depth < 0.1 m.
    Evaluated per pixel as ((u - cx) * z / fx, (v - cy) * z / fy, z) in float64, then rounded once to float32 -- the arithmetic
    the fixture pins."""
    depth = np.asarray(depth)
    rows, cols = depth.shape[:2]
    u = np.arange(cols).reshape(1, cols)
    v = np.arange(rows).reshape(rows, 1)
    z = depth.reshape(rows, cols)
    out = np.empty((rows, cols, 3), dtype=np.float32)
    out[..., 0] = (u - K[0, 2]) * z / K[0, 0]
    out[..., 1] = (v - K[1, 2]) * z / K[1, 1]
    out[..., 2] = z
    out[z < MIN_DEPTH] = 0
    return out


depth2xyzmap = back_project          # the name the reference's callers use (Utils.py:219)


def frame_cloud(rgb, depth, mask, K, glcam_in_world):
    """One keyframe's contribution to the fused cloud, in world coordinates; None when no pixel survives."""
    usable = depth >= MIN_DEPTH
    if mask is not None:
        usable &= mask > 0
    if not usable.any():
        return None
    cloud = to_cloud(back_project(depth, K)[usable], rgb[usable])
    cloud = cloud.voxel_down_sample(FRAME_VOXEL)
    cloud, _ = cloud.remove_statistical_outlier(OUTLIER_NEIGHBOURS, OUTLIER_SIGMA)
    return cloud.transform(glcam_in_world @ glcam_in_cvcam)          # OpenGL camera pose -> OpenCV camera in world


def _load_frame(color_file, want_mask):
    """the tracker's file layout: images/<id>.png beside depth_filtered/<id>.png (uint16 mm) and masks/<id>.png"""
    from .data_reader import read_depth_png, read_png
    rgb = read_png(color_file)[..., :3]
    depth = read_depth_png(color_file.replace('images', 'depth_filtered'))
    mask = read_png(color_file.replace('images', 'masks')) if want_mask else None
    return rgb, depth, mask


# ---- normalisation ------------------------------------------------------------------------------------------------------------
def largest_cluster(points, eps=0.06, min_samples=1):
    """(points of the most populated DBSCAN cluster, membership mask)."""
    from sklearn.cluster import DBSCAN
    labels = DBSCAN(eps=eps, min_samples=min_samples, n_jobs=-1).fit(points).labels_
    names, sizes = np.unique(labels, return_counts=True)
    member = labels == names[np.flatnonzero(sizes == sizes.max())[-1]]
    return points[member], member


@dataclass
class Normalisation:
    """p_n = (p + translation) * scale."""
    translation: np.ndarray
    scale: float

    @classmethod
    def fit(cls, points, max_dim=2.0):
        lo, hi = points.min(axis=0), points.max(axis=0)
        scale = max_dim / (hi - lo).max()
        return cls(-((hi + lo) / 2), scale * CUBE_FILL)

    def matrix(self):
        shift, grow = np.eye(4), np.eye(4)
        shift[:3, 3] = self.translation
        grow[:3, :3] *= self.scale
        return grow @ shift

    def inside_unit_cube(self, points):
        return (np.abs(PointCloud(points).transform(self.matrix()).points) < 1).all(axis=-1)


# ---- the fused scene ------------------------------------------------------------------------------------------------------------
class FusedScene:
    def __init__(self, K, use_mask=True):
        self.K, self.use_mask = K, use_mask
        self._jobs = []

    def add_arrays(self, rgb, depth, mask, glcam_in_world):
        self._jobs.append(lambda: frame_cloud(rgb, depth, mask if self.use_mask else None, self.K, glcam_in_world))

    def add_file(self, color_file, glcam_in_world):
        def job():
            rgb, depth, mask = _load_frame(color_file, self.use_mask)
            return frame_cloud(rgb, depth, mask, self.K, glcam_in_world)
        self._jobs.append(job)

    def fuse(self, voxel, workers=10):
        """all frames (a thread pool runs them; results keep frame order) -> one thinned cloud"""
        with ThreadPoolExecutor(max_workers=workers) as pool:
            parts = [c for c in pool.map(lambda job: job(), self._jobs) if c is not None]
        merged = parts[0].clone()
        for c in parts[1:]:
            merged += c
        return merged.voxel_down_sample(voxel)


def write_normalisation(path, norm):
    import yaml
    with open(path, 'w') as f:
        yaml.safe_dump({'translation_cvcam': np.asarray(norm.translation).tolist(), 'sc_factor': float(norm.scale)}, f)


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


# ---- plugin surface: the names and signatures bundlesdf.py / tool.py users call --------------------------------------------
def find_biggest_cluster(pts, eps=0.06, min_samples=1):
    return largest_cluster(pts, eps, min_samples)


def compute_translation_scales(pts, max_dim=2, cluster=True, eps=0.06, min_samples=1):
    """-> (translation_cvcam, sc_factor, keep_mask)"""
    if cluster:
        kept, member = largest_cluster(pts, eps, min_samples)
    else:
        kept, member = pts, np.ones(len(pts), dtype=bool)
    norm = Normalisation.fit(kept, max_dim)
    return norm.translation, norm.scale, member


def compute_scene_bounds_worker(color_file, K, glcam_in_world, use_mask, rgb=None, depth=None, mask=None):
    """-> (points, colours) of one frame in the world frame, or None"""
    if rgb is None:
        rgb, depth, file_mask = _load_frame(color_file, use_mask and mask is None)
        mask = file_mask if mask is None else mask
    cloud = frame_cloud(rgb, depth, mask if use_mask else None, K, glcam_in_world)
    return None if cloud is None else (cloud.points.copy(), cloud.colors.copy())


def make_normalisation(translation_cvcam, sc_factor):
    return Normalisation(np.asarray(translation_cvcam), sc_factor).matrix()


def compute_scene_bounds(color_files, glcam_in_worlds, K, use_mask=True, base_dir=None, rgbs=None, depths=None, masks=None,
                         cluster=True, translation_cvcam=None, sc_factor=None, eps=0.06, min_samples=1, write_files=True):
    """-> (sc_factor, translation_cvcam, cloud at real scale, cloud normalised).  Frames come either as files (`color_files`)
    or as arrays (`rgbs`, `depths`, `masks`); a given (translation_cvcam, sc_factor) is re-used instead of being fitted."""
    assert color_files is None or rgbs is None
    scene = FusedScene(K, use_mask)
    if rgbs is not None:
        for rgb, depth, mask, pose in zip(rgbs, depths, masks, glcam_in_worlds):
            scene.add_arrays(rgb, depth, mask, pose)
    else:
        for path, pose in zip(color_files, glcam_in_worlds):
            scene.add_file(path, pose)
        if base_dir is None:
            base_dir = os.path.dirname(color_files[0]) + '/../'
    out_dir = base_dir if write_files else None
    if out_dir is not None:
        os.makedirs(out_dir, exist_ok=True)

    logging.info('scene bounds: fusing %d frames', len(scene._jobs))
    fused = scene.fuse(voxel=eps / 5)
    if out_dir is not None:
        write_ply(os.path.join(out_dir, 'naive_fusion.ply'), fused)

    if translation_cvcam is None:
        _, member = largest_cluster(fused.points, eps, min_samples) if cluster else (None, np.ones(len(fused), dtype=bool))
        norm = Normalisation.fit(fused.points[member])
    else:
        norm = Normalisation(translation_cvcam, sc_factor)
        member = norm.inside_unit_cube(fused.points)
    real = fused.select(member)
    logging.info('scene bounds: translation %s, scale %s, %d of %d points kept', norm.translation, norm.scale, len(real), len(fused))
    if out_dir is not None:
        write_ply(os.path.join(out_dir, 'naive_fusion_biggest_cluster.ply'), real)
        write_normalisation(os.path.join(out_dir, 'normalization.yml'), norm)
    return norm.scale, norm.translation, real, real.clone().transform(norm.matrix())
