"""Host-side helpers of the Neural Object Field plugin surface (what `from nerf_runner import *` must bring into
bundlesdf.py): constants, preprocessing, camera rays, ray/AABB near-far, pose hand-back.

Each function states the reference lines whose behaviour it reproduces; the code is written from that behaviour.
"""
import logging
import random

import numpy as np
import torch

BAD_DEPTH = 99          # Utils.py:34
BAD_COLOR = 128         # Utils.py:35

# OpenGL camera in OpenCV camera (Utils.py:37-40)
glcam_in_cvcam = np.diag([1.0, -1.0, -1.0, 1.0])

logging.basicConfig(level=logging.INFO, format='[%(filename)s] %(message)s')


def set_seed(random_seed):
    """Utils.py:71-75."""
    np.random.seed(random_seed)
    random.seed(random_seed)
    torch.manual_seed(random_seed)


def to_homo(pts):
    """Utils.py:235-241: append a column of ones."""
    assert pts.ndim == 2, f'pts.shape: {pts.shape}'
    return np.concatenate((pts, np.ones((pts.shape[0], 1))), axis=-1)


def transform_pts(pts, tf):
    """Utils.py:253-257: R p + t over leading batch dims."""
    return (tf[..., :-1, :-1] @ pts[..., None] + tf[..., :-1, -1:])[..., 0]


def preprocess_data(rgbs, depths, masks, normal_maps, poses, sc_factor, translation):
    """nerf_helpers.py:218-240.  In place like the reference: depth < 0.1 m or outside the mask becomes BAD_DEPTH,
    background colour BAD_COLOR, colours /255, depths and pose translations move to the normalised object space."""
    depths[depths < 0.1] = BAD_DEPTH
    if masks is not None:
        rgbs[masks == 0] = BAD_COLOR
        depths[masks == 0] = BAD_DEPTH
        if normal_maps is not None:
            normal_maps[..., [1, 2]] *= -1
            normal_maps[masks == 0] = 0
        masks = masks[..., None]
    rgbs = (rgbs / 255.0).astype(np.float32)
    depths *= sc_factor
    depths = depths[..., None]
    poses[:, :3, 3] += translation
    poses[:, :3, 3] *= sc_factor
    return rgbs, depths, masks, normal_maps, poses


def get_camera_rays_np(H, W, K):
    """nerf_helpers.py:358-363: per-pixel OpenGL ray directions ((u-cx)/fx, -(v-cy)/fy, -1), float32 pixel grid."""
    i, j = np.meshgrid(np.arange(W, dtype=np.float32), np.arange(H, dtype=np.float32), indexing='xy')
    return np.stack([(i - K[0, 2]) / K[0, 0], -(j - K[1, 2]) / K[1, 1], -np.ones_like(i)], axis=-1)


def ray_box_intersection_batch(origins, dirs, bounds):
    """nerf_helpers.py:403-446: slab test of rays against one AABB, entry clamped at 0, misses -> (-1,-1).
    origins/dirs [N,3], bounds [2,3]; directions are normalised first (+1e-10)."""
    origins = torch.as_tensor(origins)
    dirs = torch.as_tensor(dirs)
    bounds = torch.as_tensor(bounds)
    dirs = dirs / (torch.norm(dirs, dim=-1, keepdim=True) + 1e-10)
    inv = 1 / dirs
    lo, hi = bounds[0], bounds[1]
    neg = inv < 0

    def axis(k, clamp):
        near_plane = torch.where(neg[:, k], hi[k], lo[k])
        far_plane = torch.where(neg[:, k], lo[k], hi[k])
        tn = (near_plane - origins[:, k]) * inv[:, k]
        if clamp:
            tn = torch.where(tn < 0, torch.zeros_like(tn), tn)
        tf_ = (far_plane - origins[:, k]) * inv[:, k]
        return tn, tf_

    tmin, tmax = axis(0, True)
    tymin, tymax = axis(1, True)
    ishit = ~((tmin > tymax) | (tymin > tmax))
    tmin = torch.where(tymin > tmin, tymin, tmin)
    tmax = torch.where(tymax < tmax, tymax, tmax)
    tzmin, tzmax = axis(2, True)
    ishit = ishit & ~((tmin > tzmax) | (tzmin > tmax))
    tmin = torch.where(tzmin > tmin, tzmin, tmin)
    tmax = torch.where(tzmax < tmax, tzmax, tmax)
    minus = -torch.ones_like(tmin)
    return torch.where(ishit, tmin, minus), torch.where(ishit, tmax, minus)


def get_optimized_poses_in_real_world(poses_normalized, pose_array, sc_factor, translation):
    """Utils.py:479-505: apply the learnt corrections, undo the normalisation, re-anchor on frame 0 and return
    OpenCV cam-in-object poses plus the frame-0 offset."""
    original = poses_normalized.copy()
    original[:, :3, 3] /= sc_factor
    original[:, :3, 3] -= translation
    tf = pose_array.get_matrices(np.arange(len(poses_normalized))).reshape(-1, 4, 4).data.cpu().numpy()
    optimized = np.array(tf @ poses_normalized).astype(np.float32)
    optimized[:, :3, 3] /= sc_factor
    optimized[:, :3, 3] -= translation
    offset = np.linalg.inv(optimized[0].copy()) @ original[0]
    for i in range(len(optimized)):
        optimized[i] = (optimized[i] @ offset) @ glcam_in_cvcam
    return optimized, offset


def mesh_to_real_world(mesh, pose_offset, translation, sc_factor):
    """Utils.py:508-514."""
    mesh.vertices = mesh.vertices / sc_factor - np.array(translation).reshape(1, 3)
    mesh.apply_transform(pose_offset)
    return mesh
