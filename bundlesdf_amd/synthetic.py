"""Synthetic posed RGBD keyframe pools: an analytic ellipsoid with a procedural
texture seen from cameras on a view sphere.  Used by bench.py, __graft_entry__.smoke() and the tests --
there is no network for datasets, and the reference ships no data.

Everything returned is in the form bundlesdf.py hands to NerfRunner AFTER preprocess_data
(nerf_helpers.py:218-240): rgbs [F,H,W,3] float32 in [0,1], depths [F,H,W,1] * sc_factor (invalid = 99*sc),
masks [F,H,W,1] uint8, poses [F,4,4] OpenGL cam-in-object normalised, K [3,3], plus the normalised point
cloud for the octree.
"""
import numpy as np

from .nerf_helpers import preprocess_data, BAD_DEPTH, glcam_in_cvcam

SEMI_AXES = np.array([0.06, 0.09, 0.12])


def fibonacci_sphere(n, radius):
    i = np.arange(n) + 0.5
    phi = np.arccos(1 - 2 * i / n)
    theta = np.pi * (1 + 5 ** 0.5) * i
    return radius * np.stack([np.cos(theta) * np.sin(phi), np.sin(theta) * np.sin(phi), np.cos(phi)], -1)


def look_at_cv(cam_pos, target=np.zeros(3)):
    """OpenCV camera (x right, y down, z forward) at cam_pos looking at target; returns cam_in_ob 4x4."""
    z = target - cam_pos
    z = z / np.linalg.norm(z)
    up = np.array([0.0, 0.0, 1.0]) if abs(z[2]) < 0.95 else np.array([0.0, 1.0, 0.0])
    x = np.cross(z, up)
    x /= np.linalg.norm(x)
    y = np.cross(z, x)
    T = np.eye(4)
    T[:3, 0], T[:3, 1], T[:3, 2], T[:3, 3] = x, y, z, cam_pos
    return T


def render_frame(cam_in_ob, K, H, W, rng, depth_noise=0.001):
    """Analytic ray/ellipsoid intersection -> (rgb uint8-range float [H,W,3], depth [H,W] metres, mask [H,W] uint8)."""
    u, v = np.meshgrid(np.arange(W, dtype=np.float64), np.arange(H, dtype=np.float64), indexing='xy')
    d_cam = np.stack([(u - K[0, 2]) / K[0, 0], (v - K[1, 2]) / K[1, 1], np.ones_like(u)], -1)
    Rm, t = cam_in_ob[:3, :3], cam_in_ob[:3, 3]
    d = d_cam @ Rm.T
    os_ = t / SEMI_AXES
    ds = d / SEMI_AXES
    a = (ds * ds).sum(-1)
    b = 2 * (ds * os_).sum(-1)
    c = (os_ * os_).sum() - 1
    disc = b * b - 4 * a * c
    hit = disc > 0
    tt = np.where(hit, (-b - np.sqrt(np.maximum(disc, 0))) / (2 * a), 0.0)
    hit &= tt > 0
    p = t + d * tt[..., None]
    rgb = 0.5 + 0.5 * np.stack([np.sin(40 * p[..., 0] + 1.0) * np.cos(25 * p[..., 1]),
                                np.sin(30 * p[..., 1] + 2.0) * np.cos(35 * p[..., 2]),
                                np.sin(45 * p[..., 2] + 0.5) * np.cos(20 * p[..., 0])], -1)
    depth = np.where(hit, tt + rng.normal(0, depth_noise, tt.shape), 0.0)     # z-depth: d_cam z component is 1
    rgb = np.where(hit[..., None], rgb * 255.0, 0.0)
    return rgb.astype(np.float32), depth.astype(np.float32), hit.astype(np.uint8)


def make_pool(n_frames=4, H=480, W=640, fx=600.0, seed=0, pose_noise=True, view_radius=0.6, max_cloud=20000,
              frame_offset=0, n_total=None, analytic_bounds=False):
    """Returns dict(rgbs, depths, masks, poses, K, sc_factor, translation, pcd_normalized, poses_gt).
    frame_offset / n_total select a shard [frame_offset, frame_offset + n_frames) of an n_total-camera lattice (data-
    parallel ranks); analytic_bounds takes sc_factor/translation from the known object extent so that every rank
    normalises identically without a collective."""
    rng = np.random.default_rng(seed + 1000003 * frame_offset)
    K = np.array([[fx, 0, W / 2], [0, fx, H / 2], [0, 0, 1]], dtype=np.float64)
    n_total = n_total or n_frames
    cams = fibonacci_sphere(n_total, view_radius)[frame_offset:frame_offset + n_frames]
    rgbs, depths, masks, cam_in_obs = [], [], [], []
    cloud = []
    for f in range(n_frames):
        T = look_at_cv(cams[f])
        rgb, depth, mask = render_frame(T, K, H, W, rng)
        rgbs.append(rgb)
        depths.append(depth)
        masks.append(mask)
        cam_in_obs.append(T)
        vs, us = np.nonzero(mask)
        sel = rng.choice(len(vs), size=min(len(vs), max(1, max_cloud // n_frames)), replace=False)
        z = depth[vs[sel], us[sel]].astype(np.float64)
        pc = np.stack([(us[sel] - K[0, 2]) / K[0, 0] * z, (vs[sel] - K[1, 2]) / K[1, 1] * z, z], -1)
        cloud.append(pc @ T[:3, :3].T + T[:3, 3])
    cam_in_obs = np.array(cam_in_obs)
    cloud = np.concatenate(cloud, 0)
    # compute_translation_scales (tool.py:28-39): max_dim 2, x0.9; x0.7 as in bundlesdf.py:151
    mx, mn = cloud.max(0), cloud.min(0)
    if analytic_bounds:
        mx, mn = SEMI_AXES.copy(), -SEMI_AXES.copy()
    translation = -(mx + mn) / 2
    sc_factor = float(2.0 / (mx - mn).max() * 0.9 * 0.7)
    poses_gt = cam_in_obs @ glcam_in_cvcam
    poses = poses_gt.copy()
    if pose_noise and n_frames > 1:                      # +-5 mm, +-2 deg on frames > 0 (exercises PoseArray)
        for f in range(0 if frame_offset > 0 else 1, n_frames):
            w = rng.normal(size=3)
            w = w / np.linalg.norm(w) * np.deg2rad(rng.uniform(0, 2))
            th = np.linalg.norm(w)
            Kx = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
            dR = np.eye(3) + np.sin(th) / max(th, 1e-12) * Kx + (1 - np.cos(th)) / max(th * th, 1e-12) * Kx @ Kx
            dT = np.eye(4)
            dT[:3, :3] = dR
            dT[:3, 3] = rng.uniform(-0.005, 0.005, 3)
            poses[f] = dT @ poses[f]
    rgbs, depths, masks = np.array(rgbs), np.array(depths), np.array(masks)
    rgbs, depths, masks, _, poses = preprocess_data(rgbs, depths, masks, None, poses.copy(), sc_factor, translation)
    pcd = ((cloud + translation) * sc_factor).astype(np.float32)
    return dict(rgbs=rgbs, depths=depths.astype(np.float32), masks=masks, poses=poses.astype(np.float32), K=K,
                sc_factor=sc_factor, translation=translation, pcd_normalized=pcd, poses_gt=poses_gt,
                semi_axes=SEMI_AXES.copy())


class PointCloud:
    """Minimal stand-in for the open3d PointCloud NerfRunner receives (`.points`, `.voxel_down_sample`,
    nerf_runner.py:127,376)."""

    def __init__(self, points):
        self.points = np.asarray(points, dtype=np.float64)

    def voxel_down_sample(self, voxel_size):
        key = np.floor(self.points / voxel_size).astype(np.int64)
        _, inv = np.unique(key, axis=0, return_inverse=True)
        inv = inv.reshape(-1)
        n = inv.max() + 1
        out = np.zeros((n, 3))
        cnt = np.bincount(inv, minlength=n)
        for k in range(3):
            out[:, k] = np.bincount(inv, weights=self.points[:, k], minlength=n) / cnt
        return PointCloud(out)
