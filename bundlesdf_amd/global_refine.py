"""The Neural-Object-Field half of BundleSdf.run_global_nerf (bundlesdf.py:636-766) on top of the drop-in NerfRunner, with the
pieces either side of the hot path that SURVEY.md 8f ranks 3-4 name: reading the tracker's output directory (or a raw capture
through a reader), scene bounds / fusion, preprocess_data, training, pose hand-back, mesh extraction, biggest component,
texture bake, back to the real world, export.  No cv2 / open3d / trimesh / pyrender.

    python -m bundlesdf_amd.global_refine --debug_dir <tracker out dir> [--video_dir <capture>] [--texture] [--n_step 2000]

`cfg_nerf` is the reference's config.yml dict (bundlesdf_amd.config.default_cfg / load_yaml); the overrides of
run_custom.py:121-134 (global refine) are the caller's to apply, exactly as in the reference.
"""
import copy
import logging
import os

import numpy as np
import torch

from .data_reader import TrackerOutput
from .mesh import Mesh, largest_component
from .nerf_helpers import get_optimized_poses_in_real_world, glcam_in_cvcam, mesh_to_real_world, preprocess_data
from .scene import compute_scene_bounds


def run_global_nerf(debug_dir, cfg_nerf, reader=None, get_texture=False, tex_res=1024, runner_kwargs=None):
    """Returns dict(mesh, optimized_cvcam_in_obs, offset, runner, frame_ids).  Files written like the reference:
    <debug_dir>/final/nerf/{config.yml, trainval_poses.txt, normalization.yml, naive_fusion*.ply}, <debug_dir>/<last id>/
    poses_after_nerf.txt, <debug_dir>/mesh_cleaned.obj, <debug_dir>/textured_mesh.obj."""
    import yaml
    from .nerf_runner import NerfRunner
    cfg = cfg_nerf
    trk = TrackerOutput(debug_dir)
    K = trk.K.copy()
    keys = trk.select(cfg['n_train_image']) if 'n_train_image' in cfg else trk.keys
    data = trk.load(keys)
    frame_ids, cam_in_obs = data['frame_ids'], data['cam_in_obs']
    out_dir = f"{debug_dir}/final/nerf"
    # a previous global refine's own final/nerf/config.yml would otherwise be the last match of the glob below ('final' sorts
    # after the numeric stamps) and its normalisation silently re-used; the reference empties the directory first (bundlesdf.py:665)
    import shutil
    shutil.rmtree(out_dir, ignore_errors=True)
    os.makedirs(out_dir, exist_ok=True)
    cfg['save_dir'] = cfg.get('save_dir') or out_dir
    os.makedirs(cfg['save_dir'], exist_ok=True)
    if reader is not None:                                          # full-resolution raw images (bundlesdf.py:673-679)
        K = reader.K.copy()
        idx = [reader.id_strs.index(f) for f in frame_ids]
        rgbs = np.array([reader.get_color(i)[..., :3] for i in idx])
        depths = np.array([reader.get_depth(i) for i in idx])
        masks = np.array([reader.get_mask(i) for i in idx])
    else:
        cfg['down_scale_ratio'] = 1                                 # images were down-scaled by the tracker (bundlesdf.py:681)
        rgbs, depths, masks = data['rgbs'], data['depths'], data['masks']
    glcam_in_obs = cam_in_obs @ glcam_in_cvcam
    sc_prev = tr_prev = None                                        # re-use the online rounds' normalisation (bundlesdf.py:696-700)
    import glob
    files = sorted(glob.glob(f"{debug_dir}/**/nerf/config.yml", recursive=True))
    if files:
        prev = yaml.safe_load(open(files[-1]))
        if prev and prev.get('sc_factor') is not None and prev.get('translation') is not None:
            sc_prev, tr_prev = float(prev['sc_factor']), np.array(prev['translation'], dtype=np.float64)
    sc_factor, translation, pcd_real_scale, pcd_normalized = compute_scene_bounds(
        None, glcam_in_obs, K, use_mask=True, base_dir=cfg['save_dir'], rgbs=rgbs, depths=depths, masks=masks, cluster=True,
        eps=0.01, min_samples=5, sc_factor=sc_prev, translation_cvcam=tr_prev)
    cfg['sc_factor'] = float(sc_factor)
    cfg['translation'] = np.asarray(translation)
    rgbs_raw = rgbs.copy()
    rgbs_p, depths_p, masks_p, _, poses = preprocess_data(rgbs.astype(np.float32), depths=depths.astype(np.float64).copy(),
                                                          masks=masks.copy(), normal_maps=None, poses=glcam_in_obs.copy(),
                                                          sc_factor=cfg['sc_factor'], translation=cfg['translation'])
    cfg['sampled_frame_ids'] = np.arange(len(rgbs_p))
    # the NORMALISED OpenGL poses: the reference's preprocess_data shifts / scales its `poses` argument in place, and that array is
    # what it then writes (bundlesdf.py:711-715, nerf_helpers.py:238-240)
    np.savetxt(f"{cfg['save_dir']}/trainval_poses.txt", poses.reshape(-1, 4))
    nerf = NerfRunner(cfg, rgbs_p, depths=depths_p.astype(np.float32), masks=masks_p, normal_maps=None, occ_masks=None,
                      poses=poses.astype(np.float32), K=K, build_octree_pcd=pcd_normalized, **(runner_kwargs or {}))
    logging.info('Start training')
    nerf.train()
    optimized_cvcam_in_obs, offset = get_optimized_poses_in_real_world(poses, nerf.models['pose_array'], cfg['sc_factor'],
                                                                       cfg['translation'])
    with open(f"{out_dir}/config.yml", 'w') as ff:
        tmp = {k: (v.tolist() if isinstance(v, np.ndarray) else v) for k, v in copy.copy(cfg).items()}
        yaml.safe_dump(tmp, ff)
    os.makedirs(f"{debug_dir}/{frame_ids[-1]}", exist_ok=True)
    np.savetxt(f"{debug_dir}/{frame_ids[-1]}/poses_after_nerf.txt", np.array(optimized_cvcam_in_obs).reshape(-1, 4))
    mesh, sigma, query_pts = nerf.extract_mesh(voxel_size=cfg['mesh_resolution'], isolevel=0, return_sigma=True)
    m = Mesh(np.asarray(mesh.vertices), np.asarray(mesh.faces))
    m.merge_vertices()
    m = largest_component(m)                                        # trimesh_split + biggest piece (bundlesdf.py:748-760)
    m = Mesh(np.asarray(m.vertices), np.asarray(m.faces))
    m.export(f'{debug_dir}/mesh_cleaned.obj')
    textured = False
    if get_texture:
        # the per-triangle atlas needs a few texels per triangle: a 2 mm mesh (~200 k triangles) does not fit 1024^2, so the
        # size follows the face count; and a failed bake must not cost the trained field -- the untextured mesh is exported
        res = m.atlas_resolution(tex_res)
        if res != tex_res:
            logging.info(f'texture: {len(m.faces)} triangles need a {res}^2 atlas (asked for {tex_res}^2)')
        try:
            m = nerf.mesh_texture_from_train_images(m, rgbs_raw=rgbs_raw.astype(np.float32), train_texture=False, tex_res=res)
            textured = True
        except (ValueError, torch.cuda.OutOfMemoryError) as e:       # the two expected failures: atlas does not fit / device memory
            logging.error(f'texture bake failed ({e!r}): exporting the UNTEXTURED mesh (result["textured"] is False)')
    m = mesh_to_real_world(m, pose_offset=offset, translation=cfg['translation'], sc_factor=cfg['sc_factor'])
    m.export(f'{debug_dir}/textured_mesh.obj')
    return dict(mesh=m, textured=textured, optimized_cvcam_in_obs=optimized_cvcam_in_obs, offset=offset, runner=nerf,
                frame_ids=frame_ids)


if __name__ == '__main__':
    import argparse
    from .config import default_cfg
    from .data_reader import YcbineoatReader
    ap = argparse.ArgumentParser()
    ap.add_argument('--debug_dir', required=True)
    ap.add_argument('--video_dir', default=None)
    ap.add_argument('--texture', action='store_true')
    ap.add_argument('--tex_res', type=int, default=1024)
    ap.add_argument('--n_step', type=int, default=2000)
    a = ap.parse_args()
    # run_custom.py:121-134 (global refine of custom data)
    cfg = default_cfg(n_step=a.n_step, N_samples=64, N_samples_around_depth=256, num_levels=16, finest_res=256, frame_features=2,
                      rgb_weight=100, fs_sdf=0.1, mesh_resolution=0.002, n_train_image=500, first_frame_weight=1, far=1.0)
    out = run_global_nerf(a.debug_dir, cfg, reader=YcbineoatReader(a.video_dir) if a.video_dir else None, get_texture=a.texture,
                          tex_res=a.tex_res)
    print('mesh', out['mesh'].vertices.shape, out['mesh'].faces.shape, '->', f'{a.debug_dir}/textured_mesh.obj')
