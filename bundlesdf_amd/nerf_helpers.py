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
    """What nerf_helpers.py:218-240 does to the keyframes before they reach NerfRunner, in place on the arrays it is given:
    invalid depth (< 0.1 m, or outside the mask) -> BAD_DEPTH, background colour -> BAD_COLOR, colours to [0,1] float32,
    metric depths and camera positions into the normalised object space p_n = (p + translation) * sc_factor, normals to the
    OpenGL camera frame.  Returns (rgbs, depths [..,1], masks [..,1], normal_maps, poses)."""
    too_close = depths < 0.1
    depths[too_close] = BAD_DEPTH
    if masks is not None:
        background = masks == 0
        rgbs[background] = BAD_COLOR
        depths[background] = BAD_DEPTH
        if normal_maps is not None:
            normal_maps[..., [1, 2]] *= -1                       # OpenCV -> OpenGL camera axes
            normal_maps[background] = 0
        masks = masks[..., None]
    rgbs = (rgbs / 255.0).astype(np.float32)
    depths *= sc_factor
    cam_centres = poses[:, :3, 3]                                # a view: the poses are normalised in place
    cam_centres += translation
    cam_centres *= sc_factor
    return rgbs, depths[..., None], masks, normal_maps, poses


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


def _to_metric(T, sc_factor, translation):
    """camera positions of normalised poses [n,4,4] back to metres, in place: p = p_n / sc_factor - translation"""
    T[:, :3, 3] /= sc_factor
    T[:, :3, 3] -= translation
    return T


def get_optimized_poses_in_real_world(poses_normalized, pose_array, sc_factor, translation):
    """The pose hand-back of Utils.py:479-505: learnt correction x input pose per keyframe (normalised object space), both
    sets taken back to metres, every optimised pose re-expressed so that keyframe 0 keeps its input pose (the corrections
    share a gauge freedom with the object frame), OpenGL -> OpenCV camera.  Returns (cam_in_ob [n,4,4] float32, offset)."""
    n = len(poses_normalized)
    before = _to_metric(poses_normalized.copy(), sc_factor, translation)
    delta = pose_array.get_matrices(np.arange(n)).reshape(-1, 4, 4).data.cpu().numpy()
    after = _to_metric(np.array(delta @ poses_normalized).astype(np.float32), sc_factor, translation)
    offset = np.linalg.inv(after[0].copy()) @ before[0]          # what maps the optimised frame 0 onto the input frame 0
    for i in range(n):
        after[i] = (after[i] @ offset) @ glcam_in_cvcam
    return after, offset


def mesh_to_real_world(mesh, pose_offset, translation, sc_factor):
    """Utils.py:508-514: vertices from the normalised object space back to metres, then the frame-0 offset of the pose
    hand-back, so that the mesh and the handed-back poses share one object frame."""
    metric = mesh.vertices / sc_factor
    mesh.vertices = metric - np.array(translation).reshape(1, 3)
    mesh.apply_transform(pose_offset)
    return mesh
