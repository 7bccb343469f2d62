"""Ray-pool construction on the MI355X: the device counterpart of bundlesdf_amd/rays.py
(make_frame_rays + denoise_rays; reference nerf_runner.py:246-316, :39-65, :178-195, nerf_helpers.py:403-446).

Per keyframe: upload image / depth / mask, separable mask dilation, one float64 pass over the pixels (ray direction, validity,
slab test against the bounding box, octree-miss test with the training tracer's DDA), then for the whole set of frames a
brute-force nearest-cloud-point test and a stable compaction (order = frame, then row-major pixel, like np.where).  torch
supplies only the exclusive scan and the buffers.  rays.py stays the NumPy restatement the tests compare this against.
"""
import ctypes as C

import numpy as np
import torch

from . import lib


def _cmp_threshold(arr_dtype, value):
    """the constant NumPy would compare an array of `arr_dtype` against (NEP 50: Python floats are weak, NumPy scalars not)"""
    dt = np.result_type(arr_dtype, value)
    return float(np.asarray(value, dtype=dt))


def frame_rays_device(field, frame_ids, images, depths, masks, poses, K, cfg, frame_offset=0, occ_masks=None,
                      cloud_pts=None, global_ids=None):
    """[N,12] float32 CUDA tensor of the rays make_frame_rays (+ denoise_rays when cloud_pts is given) would return for
    the local frames `frame_ids` (global id = global_ids[local] when given, else local + frame_offset)."""
    dev = field.device
    H, W = images[0].shape[:2]
    npx = H * W
    use_octree = bool(cfg['use_octree']) and field.occ_bits is not None
    down = int(cfg['down_scale_ratio'])
    near_sc, far_sc = cfg['near'] * cfg['sc_factor'], cfg['far'] * cfg['sc_factor']
    bounds = np.array(cfg['bounding_box'], dtype=np.float64).reshape(2, 3)
    rows_all, keep_all = [], []
    tmp = torch.empty(npx, dtype=torch.uint8, device=dev)
    for i in frame_ids:
        g = int(global_ids[i]) if global_ids is not None else i + frame_offset
        img = torch.from_numpy(np.ascontiguousarray(images[i], dtype=np.float32)).to(dev)
        dep = torch.from_numpy(np.ascontiguousarray(depths[i][..., 0], dtype=np.float32)).to(dev)
        m_in = torch.from_numpy(np.ascontiguousarray(masks[i][..., 0]).astype(np.uint8)).to(dev)
        k = 100 if g == 0 else 60 // down                        # nerf_runner.py:275-283
        if k > 1:
            m_sel = torch.empty(npx, dtype=torch.uint8, device=dev)
            lib.call('nof_mask_dilate', m_in, H, W, int(k), tmp, m_sel)
        else:
            m_sel = m_in
        occ = None
        if occ_masks is not None:
            occ = torch.from_numpy(np.ascontiguousarray(occ_masks[i]).reshape(H, W).astype(np.uint8)).to(dev)
        c = lib.NofFrameRaysCfg()
        c.fx, c.fy, c.cx, c.cy = float(K[0, 0]), float(K[1, 1]), float(K[0, 2]), float(K[1, 2])
        c.near_thr = _cmp_threshold(depths[i].dtype, near_sc)
        c.far_thr = _cmp_threshold(depths[i].dtype, far_sc)
        for a in range(3):
            c.box_lo[a], c.box_hi[a] = bounds[0, a], bounds[1, a]
        P = np.asarray(poses[g], dtype=np.float64).reshape(-1)
        for a in range(16):
            c.pose[a] = P[a]
        c.frame_id, c.valid_depth_only = int(g), int(bool(cfg['rays_valid_depth_only']))
        rows = torch.empty(npx, 12, device=dev)
        keep = torch.empty(npx, dtype=torch.uint8, device=dev)
        lib.call('nof_frame_rays', C.byref(c), img, dep, m_in, m_sel, occ, field.occ_bits if use_octree else None,
                 int(field.level or 0), H, W, rows, keep)
        rows_all.append(rows)
        keep_all.append(keep)
    if not rows_all:
        return torch.empty(0, 12, device=dev)
    rows = torch.cat(rows_all, 0)
    keep = torch.cat(keep_all, 0)
    N = rows.shape[0]
    if cloud_pts is not None and len(cloud_pts):
        sc = cfg['sc_factor']
        poses_d = torch.from_numpy(np.ascontiguousarray(poses, dtype=np.float64).reshape(-1, 16)).to(dev)
        cloud_d = torch.from_numpy(np.ascontiguousarray(cloud_pts, dtype=np.float64)).to(dev)
        lib.call('nof_cloud_filter', rows, N, keep, poses_d, cloud_d, int(cloud_d.shape[0]), C.c_double(float(cfg['far'] * sc)),
                 C.c_double(float(0.02 * sc)))
    incl = torch.cumsum(keep, 0, dtype=torch.int64)
    total = int(incl[-1].item())
    out = torch.empty(total, 12, device=dev)
    if total:
        lib.call('nof_compact_rows', rows, keep, (incl - keep).contiguous(), N, out)
    return out
