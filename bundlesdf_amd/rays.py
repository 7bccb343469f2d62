"""Ray-pool construction on the host: per-keyframe ray table, mask dilation, near/far
against the bounding box, octree-miss filter, depth denoising, and the epoch batch order.

Reference behaviour reproduced (file:line into /root/reference):
  make_frame_rays                     nerf_runner.py:246-316
  compute_near_far_and_filter_rays    nerf_runner.py:39-65
  cloud-based depth denoise           nerf_runner.py:178-195, 411-428
  DataLoader                          nerf_runner.py:90-107
Column layout of a ray row (no normal maps): dir 0-2, rgb 3-5, depth 6, mask 7, frame 8, type 9, near 10, far 11.
"""
import logging

import numpy as np
import torch
from scipy import ndimage
from scipy.spatial import cKDTree

from .nerf_helpers import BAD_DEPTH, get_camera_rays_np, ray_box_intersection_batch, to_homo

RAY_DIR, RAY_RGB, RAY_DEPTH, RAY_MASK, RAY_FRAME, RAY_TYPE, RAY_NEAR, RAY_FAR = [0, 1, 2], [3, 4, 5], 6, 7, 8, 9, 10, 11


def dilate_mask(mask, k):
    """cv2.dilate(mask, np.ones((k,k)), iterations=1) without cv2: the default anchor of an even k x k kernel is
    (k//2, k//2), i.e. the window covers offsets -k//2 ... k-1-k//2 on both axes, borders ignored.
    scipy's maximum_filter centres an even window the same way (origin 0) and the footprint is separable."""
    if k <= 1:
        return mask.copy()
    return ndimage.maximum_filter(mask, size=(k, k), mode='constant', cval=0)


def compute_near_far_and_filter_rays(cam_in_world, rays, cfg):
    """nerf_runner.py:39-65: rays of one keyframe against cfg['bounding_box'] in world space.  Rays that miss the box are
    dropped; the survivors get two more columns, the camera-frame depth |z| at which they enter and leave it (the ray
    directions are un-normalised with z = -1, so depth = |t * unit_direction_z|)."""
    rays = rays.reshape(-1, rays.shape[-1])
    d_cam = rays[:, :3]
    unit_z = (d_cam / np.linalg.norm(d_cam, axis=-1).reshape(-1, 1))[:, 2]
    d_world = (cam_in_world[:3, :3] @ d_cam.T).T
    centre = (cam_in_world @ to_homo(np.zeros(d_world.shape)).T).T[:, :3]        # the camera centre, once per ray
    box = np.array(cfg['bounding_box']).reshape(2, 3)
    t_in, t_out = (t.numpy() for t in ray_box_intersection_batch(centre, d_world, box))
    hit = t_in >= 0                                              # misses come back as -1
    depth_in = np.abs(unit_z * t_in)[hit].reshape(-1, 1)
    depth_out = np.abs(unit_z * t_out)[hit].reshape(-1, 1)
    return np.concatenate((rays[hit], depth_in, depth_out), axis=-1)


def make_frame_rays(frame_id, image, depth, mask_in, pose, K, cfg, occ_mask=None, trace_fn=None):
    """nerf_runner.py:246-316 for one keyframe.  image [H,W,3], depth [H,W,1], mask_in [H,W,1];
    trace_fn(rays_o_world f32 [n,3], rays_d_world f32 [n,3]) -> bool [n] (hits an occupied voxel) replaces the
    octree_m.ray_trace(...)[0] > 0 test (:302-314)."""
    H, W = image.shape[:2]
    mask = mask_in[..., 0].copy()
    rays = get_camera_rays_np(H, W, K)
    rays = np.concatenate([rays, image, depth, mask_in > 0, frame_id * np.ones(depth.shape)], -1)
    near_sc, far_sc = cfg['near'] * cfg['sc_factor'], cfg['far'] * cfg['sc_factor']
    invalid_depth = ((depth[..., 0] < near_sc) | (depth[..., 0] > far_sc)) & (mask > 0)
    ray_types = np.zeros((H, W, 1))
    ray_types[invalid_depth] = 1                                # 1 = masked pixel without a usable depth
    rays = np.concatenate((rays, ray_types), axis=-1)
    n = rays.shape[-1]
    down = int(cfg['down_scale_ratio'])
    k = 100 if frame_id == 0 else 60 // down                    # first-frame mask is trusted (:275-283)
    mask = dilate_mask(mask.astype(np.uint8), k)
    if occ_mask is not None:
        mask[occ_mask > 0] = 0
    if cfg['rays_valid_depth_only']:
        mask[invalid_depth] = 0
    vs, us = np.where(mask > 0)
    cur = rays[vs, us].reshape(-1, n)
    cur = cur[cur[:, RAY_TYPE] == 0]
    cur = compute_near_far_and_filter_rays(pose, cur, cfg)
    if cfg['use_octree'] and trace_fn is not None and len(cur):
        o = (pose @ to_homo(np.zeros((len(cur), 3))).T).T[:, :3]
        unit = cur[:, :3] / np.linalg.norm(cur[:, :3], axis=-1).reshape(-1, 1)
        d = (pose[:3, :3] @ unit.T).T
        cur = cur[trace_fn(o.astype(np.float32), d.astype(np.float32))]
    return cur


def denoise_rays(rays, poses, cloud_pts, cfg):
    """nerf_runner.py:178-195: a masked ray whose back-projected point is farther than 2 cm (scaled) from the
    octree cloud loses its depth (-> BAD_DEPTH, type 1) and, with type-0-only pools, is dropped."""
    sc = cfg['sc_factor']
    m = (rays[:, RAY_MASK] > 0) & (rays[:, RAY_DEPTH] <= cfg['far'] * sc)
    sub = rays[m]
    pts = sub[:, RAY_DIR] * sub[:, RAY_DEPTH].reshape(-1, 1)
    fid = sub[:, RAY_FRAME].astype(int)
    pts_w = (poses[fid] @ to_homo(pts)[..., None])[:, :3, 0]
    dists, _ = cKDTree(cloud_pts).query(pts_w, k=1, workers=-1)
    bad = dists > 0.02 * sc
    bad_ids = np.arange(len(rays))[m][bad]
    rays[bad_ids, RAY_DEPTH] = BAD_DEPTH * sc
    rays[bad_ids, RAY_TYPE] = 1
    logging.info(f"bad_mask#={bad.sum()}")
    return rays[rays[:, RAY_TYPE] == 0]


def octree_cells(points, cfg):
    """build_octree's host arithmetic (nerf_runner.py:443-476): occupied cells at max_level = 27-neighbour dilation of the
    cloud's cells, `radius` times; centres clipped to [-1,1] and re-quantised the way kaolin's quantize_points does
    (floor(clamp(n (x+1)/2, 0, n-1))).  Returns (cells [P,3] int32 at max_level, centres [P,3] float32 = the points the
    reference hands to OctreeManager, max_level, ray-tracing level)."""
    sv = cfg['octree_smallest_voxel_size'] * cfg['sc_factor']
    max_level = int(np.ceil(np.log2(2.0 / sv)))
    vs = 2.0 / (2 ** max_level)
    radius = max(1, int(np.ceil(cfg['octree_dilate_size'] / cfg['octree_smallest_voxel_size'])))
    logging.info(f"Octree voxel dilate_radius:{radius}")
    pts = np.asarray(points, dtype=np.float32)
    assert pts.min() >= -1 and pts.max() <= 1
    coords = np.floor((pts + 1) / np.float32(vs)).astype(np.int64)
    shifts = np.array([[dx, dy, dz] for dx in (-1, 0, 1) for dy in (-1, 0, 1) for dz in (-1, 0, 1)], dtype=np.int64)
    for _ in range(radius):
        coords = np.unique((coords[None] + shifts[:, None]).reshape(-1, 3), axis=0)
    n = 2 ** max_level
    centres = np.clip(((coords + 0.5) * vs - 1).astype(np.float32), -1, 1)
    q = np.floor(np.clip(n * (centres + 1.0) / 2.0, 0, n - 1.0)).astype(np.int32)
    rv = cfg['octree_raytracing_voxel_size'] * cfg['sc_factor']
    level = int(np.floor(np.log2(2.0 / rv)))
    return q, centres, max_level, level


class DataLoader:
    """nerf_runner.py:90-107: epoch permutation from torch.randperm (CPU generator); a batch is the next
    `batch_size` ids, and when fewer than batch_size+1 remain the permutation is redrawn (the tail is dropped).
    The ray table stays resident in HBM; only the ids are uploaded, once per epoch."""

    def __init__(self, rays, batch_size):
        self.rays = rays
        self.batch_size = batch_size
        self.pos = 0
        self._new_epoch()

    def _new_epoch(self):
        self.ids = torch.randperm(len(self.rays))
        self.ids_dev = self.ids.to(self.rays.device) if self.rays.is_cuda else self.ids

    def next_ids(self):
        if self.pos + self.batch_size < len(self.ids):
            a = self.pos
            self.pos += self.batch_size
        else:
            self._new_epoch()
            a = 0
            self.pos = self.batch_size
        self.batch_ray_ids = self.ids[a:a + self.batch_size]
        return self.ids_dev[a:a + self.batch_size]

    def __next__(self):
        return self.rays[self.next_ids()]
