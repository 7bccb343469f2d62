#!/usr/bin/env python
"""Generates tests/golden/scene_vectors.npz by EXECUTING the reference's own scene-bounds code on CPU (tool.py:18-132,
Utils.py:207-231), cut out of the read-only mount with `ast` like make_golden.py does -- nothing is copied into this repository.

    python tests/golden/make_golden_scene.py        (needs /root/reference; run in the build container)

open3d is absent here: the reference's `toOpen3dCloud` / `o3d.geometry.PointCloud` calls run on bundlesdf_amd.scene.PointCloud
(the restatement of open3d's voxel_down_sample / remove_statistical_outlier / transform), so the fixture pins the reference's
OWN logic around those calls -- depth2xyzmap, validity masks, camera convention, DBSCAN biggest cluster (real scikit-learn),
translation / scale, the keep mask of the re-used normalisation, the order of operations -- not open3d's filters themselves.
"""
import copy
import logging
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_golden import cut                     # noqa: E402
from bundlesdf_amd import scene, synthetic       # noqa: E402
from bundlesdf_amd.nerf_helpers import glcam_in_cvcam   # noqa: E402


def main():
    import joblib
    from sklearn.cluster import DBSCAN

    class Vec:
        @staticmethod
        def Vector3dVector(x):
            return np.asarray(x)

    class Cloud(scene.PointCloud):
        def __init__(self):
            super().__init__()

        def voxel_down_sample(self, v):
            return self._wrap(super().voxel_down_sample(v))

        def remove_statistical_outlier(self, nb_neighbors, std_ratio):
            p, ind = super().remove_statistical_outlier(nb_neighbors, std_ratio)
            return self._wrap(p), ind

        @staticmethod
        def _wrap(p):
            c = Cloud()
            c.points, c.colors = p.points, p.colors
            return c

    o3d = types.SimpleNamespace(geometry=types.SimpleNamespace(PointCloud=Cloud), utility=Vec,
                                io=types.SimpleNamespace(write_point_cloud=lambda *a, **k: None))

    class Yaml:
        @staticmethod
        def dump(obj, f):
            pass
    ns = dict(np=np, o3d=o3d, joblib=joblib, DBSCAN=DBSCAN, logging=logging, copy=copy, os=os, yaml=Yaml,
              glcam_in_cvcam=glcam_in_cvcam, open=lambda *a, **k: open(os.devnull, 'w'))
    for s in cut('Utils.py', ['toOpen3dCloud', 'depth2xyzmap']):
        exec(s, ns)
    for s in cut('tool.py', ['find_biggest_cluster', 'compute_translation_scales', 'compute_scene_bounds_worker',
                             'compute_scene_bounds']):
        exec(s, ns)

    rng = np.random.default_rng(0)
    F, H, W = 5, 120, 160
    K = np.array([[150.0, 0, 80.0], [0, 150.0, 60.0], [0, 0, 1]])
    cams = synthetic.fibonacci_sphere(F, 0.5)
    rgbs, depths, masks, glcams = [], [], [], []
    for i in range(F):
        cam_in_ob = synthetic.look_at_cv(cams[i])
        rgb, depth, mask = synthetic.render_frame(cam_in_ob, K, H, W, rng)
        depth = depth.copy()
        depth[5:9, 5:9] = 0.45                                   # a stray blob outside the mask / far from the object
        rgbs.append((rgb * 255 if rgb.max() <= 1 else rgb).astype(np.uint8))
        depths.append(depth)
        masks.append(mask)
        glcams.append(cam_in_ob @ glcam_in_cvcam)
    rgbs, depths, masks, glcams = np.array(rgbs), np.array(depths), np.array(masks), np.array(glcams)
    out = dict(K=K, rgbs=rgbs, depths=depths, masks=masks, glcams=glcams)
    out['xyz0'] = ns['depth2xyzmap'](depths[0], K)
    w = ns['compute_scene_bounds_worker'](None, K, glcams[1], True, rgbs[1], depths[1], masks[1])
    out['worker_pts'], out['worker_colors'] = w
    pts = rng.normal(size=(400, 3)) * 0.03
    pts[:40] += 0.5                                              # a second, smaller cluster
    t, s, keep = ns['compute_translation_scales'](pts, cluster=True, eps=0.06, min_samples=1)
    out['cts_pts'], out['cts_t'], out['cts_s'], out['cts_keep'] = pts, t, np.float64(s), keep
    t2, s2, keep2 = ns['compute_translation_scales'](pts, cluster=False)
    out['cts_t_nocluster'], out['cts_s_nocluster'] = t2, np.float64(s2)
    sc, tr, real, norm = ns['compute_scene_bounds'](None, glcams, K, use_mask=True, base_dir='/tmp', rgbs=rgbs, depths=depths,
                                                    masks=masks, cluster=True, eps=0.01, min_samples=5)
    out['csb_sc'], out['csb_tr'] = np.float64(sc), np.asarray(tr)
    out['csb_real'], out['csb_norm'], out['csb_norm_colors'] = np.asarray(real.points), np.asarray(norm.points), np.asarray(norm.colors)
    # re-used normalisation (run_global_nerf: bundlesdf.py:696-705)
    sc3, tr3, real3, norm3 = ns['compute_scene_bounds'](None, glcams, K, use_mask=True, base_dir='/tmp', rgbs=rgbs, depths=depths,
                                                        masks=masks, cluster=True, eps=0.01, min_samples=5, sc_factor=sc * 1.3,
                                                        translation_cvcam=np.asarray(tr) + 0.004)
    out['csb_reuse_sc'], out['csb_reuse_tr'], out['csb_reuse_norm'] = np.float64(sc3), np.asarray(tr3), np.asarray(norm3.points)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'scene_vectors.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, os.path.getsize(path) // 1024, 'KiB', {k: np.asarray(v).shape for k, v in out.items() if k.startswith('csb')})


if __name__ == '__main__':
    main()
