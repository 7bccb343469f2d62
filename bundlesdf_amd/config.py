"""Neural Object Field hyper-parameters: the keys of the reference's config.yml (config.yml:2-102) that the hot path reads,
with that file's default values.  Drivers override entries exactly like run_custom.py:50-62,121-134 do."""
import numpy as np


def default_cfg(**over):
    cfg = dict(n_step=500, N_rand=2048, lrate=0.01, lrate_pose=0.01, decay_rate=0.1, amp=True,
               N_samples=128, N_samples_around_depth=64, N_importance=0, perturb=1, use_viewdirs=1,
               i_embed=1, i_embed_views=2, multires=8, multires_views=3, feature_grid_dim=2,
               raw_noise_std=0, finest_res=128, base_res=16, num_levels=4, log2_hashmap_size=22,
               use_octree=1, first_frame_weight=10, denoise_depth_use_octree_cloud=True,
               octree_embed_base_voxel_size=0.02, octree_smallest_voxel_size=0.02,
               octree_raytracing_voxel_size=0.02, octree_dilate_size=0.02, down_scale_ratio=1,
               bounding_box=[[-1, -1, -1], [1, 1, 1]], use_mask=1, dilate_mask_size=0,
               rays_valid_depth_only=True, near=0.1, far=2, rgb_weight=10, depth_weight=0, trunc=0.01,
               trunc_start=0.01, sdf_lambda=5, neg_trunc_ratio=1, trunc_decay_type='', fs_weight=100,
               empty_weight=0.01, fs_rgb_weight=0, trunc_weight=6000, frame_features=0, optimize_poses=1,
               pose_reg_weight=0, eikonal_weight=0, feature_reg_weight=0.1, fs_sdf=0.001,
               mesh_resolution=0.005, max_trans=0.02, max_rot=20, save_octree_clouds=False,
               tv_loss_weight=0, no_batching=0, chunk=99999999999, netchunk=6553600,
               i_print=999999, i_img=999999, i_weights=999999, i_mesh=999999, i_pose=999999,
               sc_factor=1.0, translation=np.zeros(3), save_dir=None, datadir=None)
    cfg.update(over)
    return cfg


def load_yaml(path, **over):
    import yaml
    with open(path) as f:
        cfg = yaml.safe_load(f)
    cfg.update(over)
    return cfg


def validate_cfg(cfg):
    """Options of the reference's config the MI355X path does not implement are rejected loudly, never ignored.
    (All of them are 0 / off in the reference's own config.yml; two are dead code there)
    eikonal_weight > 0 IS implemented (nof_eikonal): the reference's train_loop path for it is dead code (nerf_runner.py:686,
    1297-1302), so the term uses the one meaningful normal of the reference, d sdf / d x of run_network_density (:1342-1345)."""
    bad = []
    if float(cfg.get('depth_weight', 0)) > 0:
        bad.append("depth_weight > 0 (references an undefined name in the reference, nerf_runner.py:718)")
    if int(cfg.get('N_importance', 0)) > 0:
        bad.append("N_importance > 0 (broken in the reference, nerf_runner.py:1106)")
    if int(cfg.get('N_samples_around_depth', 0)) < 2:
        bad.append("N_samples_around_depth < 2 (the reference requires the depth-guided branch, nerf_runner.py:1081)")
    if not int(cfg.get('use_viewdirs', 1)):
        bad.append("use_viewdirs = 0")
    if float(cfg.get('raw_noise_std', 0)) > 0:
        bad.append("raw_noise_std > 0")
    if int(cfg.get('feature_grid_dim', 2)) != 2:
        bad.append("feature_grid_dim != 2")
    if bad:
        raise NotImplementedError('Neural Object Field (MI355X): unsupported configuration: ' + '; '.join(bad))
