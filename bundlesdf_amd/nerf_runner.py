"""Drop-in `nerf_runner` module for NVlabs/BundleSDF on MI355X: same constructor, methods and attributes that
bundlesdf.py drives, with the whole optimisation step running in hand-written HIP kernels behind
the C ABI of libnof_hip.so (include/nof_hip.h).

Plugin surface kept (call sites in /root/reference/bundlesdf.py):
    NerfRunner(cfg, images, depths=, masks=, normal_maps=, poses=, K=, occ_masks=, build_octree_pcd=)   :219,225,724
    .add_new_frames(rgbs, depths, masks, normal_maps, poses, occ_masks=, new_pcd=, reuse_weights=False)   :223
    .train()                                                                                              :228,726
    .extract_mesh(isolevel=0, voxel_size=, return_sigma=)                                                 :234,747
    .mesh_texture_from_train_images(mesh, rgbs_raw=, train_texture=False, tex_res=)                       :763
    .models['pose_array'].get_matrices(ids)   (via get_optimized_poses_in_real_world, Utils.py:491)       :231
    .cfg['translation'], .cfg['sc_factor']                                                                :235
and the names `from nerf_runner import *` must provide (preprocess_data, get_optimized_poses_in_real_world,
mesh_to_real_world, glcam_in_cvcam, BAD_DEPTH, set_seed).
Beside them, the methods of the reference class that its own scripts use: render_images / the i_img canvas (nerf_runner.py:586-637,
768-791), save_weights / load_weights (:528-577), mesh_vertex_color_from_network (:1412-1429).
There is no CPU fallback: constructing a runner without the HIP library or without a GPU raises.
"""
import logging
import os

import numpy as np
import torch

from . import lib
from .checkpoint import load_reference_checkpoint, to_reference_checkpoint
from .field import GraphedStep, NeuralObjectField
from .mesh_gpu import marching_cubes_gpu, marching_cubes_lewiner_gpu, marching_tetrahedra_gpu
from .mesh import make_mesh, marching_tetrahedra
from .nerf_helpers import *          # noqa: F401,F403  (re-exported on purpose, like the reference module does)
from .nerf_helpers import set_seed, get_optimized_poses_in_real_world, mesh_to_real_world, glcam_in_cvcam
from .rays import DataLoader, denoise_rays, make_frame_rays, octree_cells
from .rays_gpu import frame_rays_device

__all__ = ['NerfRunner', 'PoseArrayView', 'preprocess_data', 'get_optimized_poses_in_real_world', 'mesh_to_real_world',
           'glcam_in_cvcam', 'BAD_DEPTH', 'BAD_COLOR', 'set_seed', 'get_camera_rays_np', 'ray_box_intersection_batch',
           'to_homo', 'transform_pts']


class PoseArrayView:
    """What callers use of models['pose_array'] (nerf_helpers.py:127-154): `.data` [F,6] and `.get_matrices(ids)`
    -> [n,4,4] float32 CUDA tensor of the learnt corrections (identity for frame 0), evaluated by nof_pose_fwd."""

    def __init__(self, field):
        self._f = field

    @property
    def data(self):
        return self._f.pose.view(self._f.F, 6)

    def get_matrices(self, ids):
        import ctypes as C
        f = self._f
        if not torch.is_tensor(ids):
            ids = torch.tensor(np.asarray(ids)).long()
        eye = torch.eye(4, device=f.device).reshape(1, 16).repeat(f.F, 1).contiguous()
        tf = torch.empty(f.F, 12, device=f.device)
        lib.call('nof_pose_fwd', f.pose if f.optimize_poses else None, eye, C.c_float(f.max_trans), C.c_float(f.max_rot),
                 tf, f.F)
        Ts = torch.eye(4, device=f.device).reshape(1, 4, 4).repeat(f.F, 1, 1)
        Ts[:, :3, :4] = tf.view(f.F, 3, 4)
        return Ts[ids.to(f.device)]


class NerfRunner:
    def __init__(self, cfg, images, depths, masks, normal_maps, poses, K, _run=None, occ_masks=None,
                 build_octree_pcd=None, precision=None, n_sigma=2, n_color=3, world_size=1, rank=0, grad_sync=None,
                 frame_offset=0, hidden=64):
        if not torch.cuda.is_available():
            raise lib.NofError('NerfRunner needs an MI355X: there is no CPU path for the Neural Object Field')
        lib.load()
        set_seed(0)
        self.cfg = cfg
        self.cfg['tv_loss_weight'] = eval(str(self.cfg.get('tv_loss_weight', 0)))
        self._run = _run
        self.images, self.depths, self.masks = images, depths, masks
        self.poses = poses
        self.normal_maps = None          # normal maps are never used on the hot path (bundlesdf.py passes None)
        self.occ_masks = occ_masks
        self.K = K.copy()
        self.mesh = None
        self.N_iters = self.cfg['n_step'] + 1
        self.build_octree_pts = np.asarray(build_octree_pcd.points).copy()
        # amp: true in config.yml selects the 16-bit MFMA path: fp16 = the reference's autocast operand type, forward with the
        # hi/lo operand split (outputs within 1e-3 of fp32), backward plain fp16 with a loss scale like its GradScaler
        self.precision = precision or cfg.get('mfma_precision', 'fp16x3' if cfg.get('amp', True) else 'fp32')
        self.n_sigma, self.n_color, self.hidden = n_sigma, n_color, hidden
        self.world_size, self.rank, self.grad_sync = world_size, rank, grad_sync
        # data parallel: `images/depths/masks` hold this rank's keyframes, which are frames frame_offset.. of `poses`
        self.frame_offset = int(frame_offset)
        # local keyframe i of this rank is global frame frame_ids[i] (a row of `poses` / of the pose and feature arrays); a
        # contiguous shard at construction, whatever add_new_frames appends afterwards
        self.frame_ids = self.frame_offset + np.arange(len(images), dtype=np.int64)
        self.device = torch.device('cuda')

        r = int(cfg['down_scale_ratio'])
        self.down_scale = np.ones(2, dtype=np.float32)
        if r != 1:                                   # strided subsampling, no interpolation (nerf_runner.py:129-148)
            H, W = images[0].shape[:2]
            self.images, self.depths, self.masks = images[:, ::r, ::r], depths[:, ::r, ::r], masks[:, ::r, ::r]
            if occ_masks is not None:
                self.occ_masks = occ_masks[:, ::r, ::r]
            self.H, self.W = self.images.shape[1:3]
            self.cfg['dilate_mask_size'] = int(self.cfg['dilate_mask_size'] // r)
            self.K[0] *= float(self.W) / W
            self.K[1] *= float(self.H) / H
            self.down_scale = np.array([float(self.W) / W, float(self.H) / H])
        self.H, self.W = self.images[0].shape[:2]

        self.field = None
        self.create_nerf()
        if self.cfg['use_octree']:
            self.build_octree()
        self.global_step = 0
        print("sc_factor", self.cfg['sc_factor'])
        print("translation", self.cfg['translation'])
        self.rays = self._frame_rays_tensor(range(len(self.masks)))
        print("rays", self.rays.shape)
        self.data_loader = DataLoader(rays=self.rays, batch_size=self.cfg['N_rand'])

    # ---- model ---------------------------------------------------------------------------------------
    def create_nerf(self, device=None):
        """nerf_runner.py:204-242: fresh hash grid, SDF/colour MLPs, frame features, pose corrections (and,
        because the optimiser state lives in the same flat buffers, create_optimizer :492-504)."""
        old = self.field
        self.field = NeuralObjectField(self.cfg, len(self.poses), self.poses, precision=self.precision,
                                       n_sigma=self.n_sigma, n_color=self.n_color, world_size=self.world_size,
                                       rank=self.rank, hidden=self.hidden)
        if old is not None and old.occ_bits is not None:
            self.field.occ_bits, self.field.level, self.field.max_level = old.occ_bits, old.level, old.max_level
            self.field.max_hits = old.max_hits
        self.models = {'pose_array': PoseArrayView(self.field) if self.cfg['optimize_poses'] else None,
                       'field': self.field}

    def create_optimizer(self):
        self.field.exp_avg.zero_()
        self.field.exp_avg_sq.zero_()
        self.field.grads.zero_()
        self.field.global_step = 0
        self.field.adam_steps = 0
        self.field._packed_step = None

    # ---- occupancy ---------------------------------------------------------------------------------------
    def build_octree(self):
        """nerf_runner.py:436-489: occupied cells = 27-neighbour dilation of the cloud's cells at max_level; the ray
        tracing level is floor(log2(2/(octree_raytracing_voxel_size*sc)))."""
        cfg = self.cfg
        q, centres, max_level, level = octree_cells(self.build_octree_pts, cfg)
        self.octree_levels = (max_level, level)
        self.field.set_occupancy(q, max_level, level)
        if cfg.get('save_octree_clouds', False) and cfg.get('save_dir'):
            os.makedirs(cfg['save_dir'], exist_ok=True)
            make_mesh(centres, np.zeros((0, 3), dtype=np.int64)).export(f"{cfg['save_dir']}/build_octree_cloud_dilated.ply")

    def _trace_hits(self, o, d):
        _, _, nh = self.field.trace(torch.from_numpy(o).to(self.device), torch.from_numpy(d).to(self.device))
        return (nh > 0).cpu().numpy()

    def _frame_rays_tensor(self, frame_ids):
        """[N,12] float32 CUDA tensor of the rays of the local frames `frame_ids`: built on the device
        (bundlesdf_amd/rays_gpu.py) unless cfg['device_ray_pool'] is false, in which case the NumPy path of rays.py (what
        the reference does on the host) is used and uploaded."""
        if self.cfg.get('device_ray_pool', True):
            cloud = self.build_octree_pts if self.cfg['denoise_depth_use_octree_cloud'] else None
            return frame_rays_device(self.field, list(frame_ids), self.images, self.depths, self.masks, self.poses, self.K,
                                     self.cfg, occ_masks=self.occ_masks, cloud_pts=cloud, global_ids=self.frame_ids)
        return torch.tensor(self._frame_rays(frame_ids), dtype=torch.float, device=self.device)

    def _frame_rays(self, frame_ids):
        rays_ = []
        for i in frame_ids:
            occ = self.occ_masks[i] if self.occ_masks is not None else None
            g = int(self.frame_ids[i])
            rays_.append(make_frame_rays(g, self.images[i], self.depths[i], self.masks[i], self.poses[g], self.K, self.cfg,
                                         occ_mask=occ, trace_fn=self._trace_hits if self.cfg['use_octree'] else None))
        rays = np.concatenate(rays_, axis=0)
        if self.cfg['denoise_depth_use_octree_cloud']:
            logging.info("denoise cloud")
            rays = denoise_rays(rays, self.poses, self.build_octree_pts, self.cfg)
        return rays

    # ---- growing the keyframe pool (nerf_runner.py:352-433) ----------------------------------------------
    def add_new_frames(self, images, depths, masks, normal_maps, poses, occ_masks=None, new_pcd=None, reuse_weights=False):
        """`poses` holds ALL frames (old + new), the image arrays the new frames only (bundlesdf.py:223).  Data parallel
        (world_size > 1): every rank is handed the same new frames -- global ids n_old .. n_old + n_new - 1 -- and keeps a
        contiguous share of them (dist.shard_frames); its local -> global frame map `frame_ids` grows accordingly, so rays,
        pose corrections and frame features stay addressed by global frame id on every rank."""
        n_old_total = self.field.F
        new_global = np.arange(n_old_total, n_old_total + len(images), dtype=np.int64)
        if len(poses) != n_old_total + len(images):
            raise ValueError(f'add_new_frames: {len(poses)} poses for {n_old_total} old + {len(images)} new frames')
        if self.world_size > 1:
            from .dist import shard_frames
            lo, hi = shard_frames(len(images), self.rank, self.world_size)
            images, depths, masks, new_global = images[lo:hi], depths[lo:hi], masks[lo:hi], new_global[lo:hi]
            if occ_masks is not None:
                occ_masks = occ_masks[lo:hi]
        prev = len(self.images)
        r = int(self.cfg['down_scale_ratio'])
        images, depths, masks = images[:, ::r, ::r], depths[:, ::r, ::r], masks[:, ::r, ::r]
        if occ_masks is not None:
            self.occ_masks = np.concatenate((self.occ_masks, occ_masks[:, ::r, ::r]), axis=0)
        self.images = np.concatenate((self.images, images), axis=0)
        self.depths = np.concatenate((self.depths, depths), axis=0)
        self.masks = np.concatenate((self.masks, masks), axis=0)
        self.frame_ids = np.concatenate((self.frame_ids, new_global))
        self.poses = poses.copy()
        old = self.field
        # new frame count -> new pose/feature arrays; reuse_weights keeps table + MLPs (+ old frame features)
        self.create_nerf()
        if reuse_weights:
            self.field.load_parameters(table=old.table, mlp=old.mlp)
            if old.ff > 0:
                f = self.field.feat.view(self.field.F, old.ff)
                f[:old.F] = old.feat.view(old.F, old.ff)             # rows are GLOBAL frames: the old ones keep their features
        if self.cfg['use_octree']:
            pcd = new_pcd.voxel_down_sample(0.005)
            self.build_octree_pts = np.asarray(pcd.points).copy()
            self.build_octree()
        self.create_optimizer()
        self.global_step = 0
        if len(self.masks) > prev:                                   # (a rank's share of the new frames can be empty)
            self.rays = torch.cat((self.rays, self._frame_rays_tensor(range(prev, len(self.masks)))), dim=0)
        self.data_loader = DataLoader(rays=self.rays, batch_size=self.cfg['N_rand'])

    # ---- training (nerf_runner.py:679-763, 855-863) --------------------------------------------------------
    def train_loop(self, ids=None):
        """One optimisation step on the next batch of the loader (the reference passes the gathered rows; here the
        rows are gathered on the device from the resident pool)."""
        next_ids = None
        if ids is None and not self.field.march_ahead:
            ids = self.data_loader.next_ids()
        elif ids is None:
            # the loader runs one batch ahead: this step's optimiser launch marches the NEXT batch's rays (field.march_ahead, off by default)
            dl = self.data_loader
            ahead = getattr(self, '_ids_ahead', None)
            if ahead is not None and ahead[2] is dl:
                ids, host_ids = ahead[0], ahead[1]
            else:
                ids = dl.next_ids()
                host_ids = dl.batch_ray_ids
            next_ids = dl.next_ids()
            self._ids_ahead = (next_ids, dl.batch_ray_ids, dl)
            dl.batch_ray_ids = host_ids                    # (what the loader reports: the batch being trained)
        seed = self.cfg.get('seed', 0) + 7919 * self.rank
        f = self.field
        # captured-step mode (cfg hip_graph, default OFF since the eager step overlaps its two backward chains on two streams and is
        # 5-9 % faster than the single captured chain; the capture frees the host): after a few eager steps (module load, buffers, the hash backward's side
        # stream) the whole step is replayed as one HIP graph; eager otherwise (data parallel, per-kernel timing, truncation decay)
        graph_ok = (self.cfg.get('hip_graph', False) and self.grad_sync is None and f.profile is None
                    and not self.cfg.get('trunc_decay_type', ''))
        g = getattr(self, '_graph', None)
        if g is not None and (g.field is not f or g.R != ids.shape[0] or not g.usable() or not graph_ok):
            g = self._graph = None
        # (the device step state drives schedule AND bias correction from one counter: no capture while the Adam moments are
        # not as old as the schedule step, e.g. after a checkpoint without optimiser state)
        if (graph_ok and g is None and getattr(self, '_eager_steps', 0) >= 3 and getattr(self, '_graph_field', None) is f
                and f.adam_steps == f.global_step):
            g = GraphedStep(f, self.rays, ids.shape[0], seed)
            g = self._graph = g if g.usable() else None
        if g is not None:
            g(ids)
        else:
            if getattr(self, '_graph_field', None) is not f:
                self._graph_field, self._eager_steps = f, 0
            self._eager_steps += 1
            f.train_step(self.rays, ids, ids.shape[0], seed=seed, grad_sync=self.grad_sync, next_ids=next_ids)
        if self.global_step % self.cfg['i_print'] == 0 and self.global_step > 0:
            if self.field.poll_flags() & 4:
                logging.warning('non-finite weight gradient in the fp16 backward: that step was skipped, loss scale halved')
            m = self.field.losses()
            logging.info(f"Iter: {self.global_step}, " + ", ".join(f"{k}: {v:.7f}" for k, v in m.items()))
        if self.global_step % self.cfg['i_weights'] == 0 and self.global_step > 0 and self.cfg.get('save_dir'):
            self.save_weights(os.path.join(self.cfg['save_dir'], 'model_latest.pth'))
        if self.global_step % self.cfg['i_img'] == 0 and self.global_step > 0 and self.cfg.get('save_dir'):
            self.save_image_canvas(f"{self.cfg['save_dir']}/image_step_{self.global_step:07d}.png")

    # ---- forward image renderer (nerf_runner.py:586-637, used by train_loop :768-791) -----------------------
    @torch.no_grad()
    def render_images(self, img_i, cur_rays=None):
        """Colour and depth of keyframe `img_i` (a GLOBAL frame id, as stored in the ray table) rendered through the field.
        Every ray of that frame in the pool goes through the forward half of the step with perturb=False, N_rand rays at a time
        (the reference sets chunk = N_rand, :595-598); depth = z at the first SDF sign change, far*sc_factor for rays without one
        (:604-612); results are scattered back to the H x W pixel each ray came from (:616-634, last ray wins on a shared pixel).
        Returns (rgb [H,W,3], depth [H,W], ray_mask [H,W,3] uint8, gt_rgb [H,W,3], gt_depth [H,W], extras) like the reference;
        extras holds the per-ray arrays 'raw' [n,S,4], 'z_vals' [n,S], 'valid_samples' [n,S], 'rgb_map' [n,3], 'depth' [n]."""
        f = self.field
        if cur_rays is None:
            sel = torch.nonzero(self.rays[:, 8] == float(img_i)).reshape(-1)
            pool = self.rays
        else:                                            # (the reference's own cur_rays path dies on an unbound name, :594)
            pool = torch.as_tensor(cur_rays, dtype=torch.float32, device=self.device).contiguous()
            sel = torch.arange(pool.shape[0], device=self.device)
        n = int(sel.numel())
        S = self.cfg['N_samples'] + self.cfg['N_samples_around_depth']
        chunk = int(self.cfg['N_rand'])
        keys = ('rgb_map', 'depth', 'raw', 'z_vals', 'valid')
        parts = {k: [] for k in keys}
        for i in range(0, n, chunk):
            ids = sel[i:i + chunk].contiguous()
            R = int(ids.numel())
            b = f.render_batch(pool, ids, R)
            for k in keys:
                parts[k].append(b[k].clone())
        cat = {k: (torch.cat(v, 0) if v else torch.zeros(0, device=self.device)) for k, v in parts.items()}
        rows = pool[sel].cpu().numpy()
        rgb = cat['rgb_map'].cpu().numpy().reshape(-1, 3)
        depth = cat['depth'].cpu().numpy().reshape(-1)
        H, W = self.H, self.W
        rgb_full = np.zeros((H, W, 3), dtype=float)
        depth_full = np.zeros((H, W), dtype=float)
        ray_mask_full = np.zeros((H, W, 3), dtype=np.uint8)
        gt_rgb_full = np.zeros((H, W, 3), dtype=float)
        gt_depth_full = np.zeros((H, W), dtype=float)
        if n:
            X = rows[:, 0:3].copy()
            X[:, [1, 2]] = -X[:, [1, 2]]
            projected = (self.K @ X.T).T
            uvs = (projected / projected[:, 2].reshape(-1, 1)).round().astype(int)
            ray_type = rows[:, 9]
            good, unc = uvs[ray_type == 0], uvs[ray_type == 1]
            ray_mask_full[good[:, 1], good[:, 0]] = [255, 0, 0]
            ray_mask_full[unc[:, 1], unc[:, 0]] = [0, 255, 0]
            rgb_full[uvs[:, 1], uvs[:, 0]] = rgb
            depth_full[uvs[:, 1], uvs[:, 0]] = depth
            gt_rgb_full[uvs[:, 1], uvs[:, 0]] = rows[:, 3:6]
            gt_depth_full[uvs[:, 1], uvs[:, 0]] = rows[:, 6]
        extras = {'raw': cat['raw'].reshape(n, S, 4), 'z_vals': cat['z_vals'].reshape(n, S),
                  'valid_samples': cat['valid'].reshape(n, S).bool(), 'rgb_map': cat['rgb_map'].reshape(n, 3), 'depth': cat['depth']}
        return rgb_full, depth_full, ray_mask_full, gt_rgb_full, gt_depth_full, extras

    def save_image_canvas(self, path):
        """train_loop's i_img block (nerf_runner.py:768-791): up to six keyframes (every len/5-th + the last), one row each:
        [render | gt | depth | gt depth | ray-type mask over the render], written as one PNG."""
        from .data_reader import write_png
        ids = sorted(int(i) for i in torch.unique(self.rays[:, 8]).cpu().numpy().astype(int).tolist())
        last = ids[-1]
        ids = ids[::max(1, len(ids) // 5)]
        if last not in ids:
            ids.append(last)
        to8b = lambda x: (255 * np.clip(x, 0, 1)).astype(np.uint8)
        far = self.cfg['far'] * self.cfg['sc_factor']
        canvas = []
        for frame_idx in ids:
            rgb, depth, ray_mask, gt_rgb, gt_depth, _ = self.render_images(frame_idx)
            mask_vis = np.clip((rgb * 255 * 0.2 + ray_mask * 0.8).astype(np.uint8), 0, 255)
            gt_depth = np.clip(gt_depth, self.cfg['near'] * self.cfg['sc_factor'], far)
            depth_vis = np.tile(np.concatenate((to8b(depth / far), to8b(gt_depth / far)), axis=1)[..., None], (1, 1, 3))
            canvas.append(np.concatenate((to8b(np.concatenate((rgb, gt_rgb), axis=1)), depth_vis, mask_vis), axis=1))
        os.makedirs(os.path.dirname(path) or '.', exist_ok=True)
        write_png(path, np.concatenate(canvas, axis=0).astype(np.uint8))
        return path

    def train(self):
        set_seed(0)
        for it in range(self.N_iters):
            if it % max(1, self.N_iters // 10) == 0:
                logging.info(f'train progress {it}/{self.N_iters}')
            self.train_loop()
            self.global_step += 1
        torch.cuda.synchronize()
        flags = self.field.poll_flags()
        if flags:
            logging.warning(f"nof flags={flags}: 1 = ray exceeded max_hits, 2 = inconsistent sample walk (common.cu:66-72), "
                            f"4 = non-finite weight gradient in the 16-bit backward (loss scale halved)")

    def get_truncation(self):
        return self.field.truncation()

    # ---- renderer side (nerf_runner.py:1351-1409) ----------------------------------------------------------
    @torch.no_grad()
    def extract_mesh(self, level=None, voxel_size=0.003, isolevel=0.0, return_sigma=False):
        voxel_size *= self.cfg['sc_factor']
        bounds = np.array(self.cfg['bounding_box']).reshape(2, 3)
        tx = np.arange(bounds[0, 0] + 0.5 * voxel_size, bounds[1, 0], voxel_size)
        ty = np.arange(bounds[0, 1] + 0.5 * voxel_size, bounds[1, 1], voxel_size)
        tz = np.arange(bounds[0, 2] + 0.5 * voxel_size, bounds[1, 2], voxel_size)
        f = self.field
        # fused dense query on the device: voxel centres, octree mask, hash encode, sigma net (nerf_runner.py:1363-1386)
        sigma_dev = f.query_sdf_grid(tx, ty, tz, outside_value=1.0, use_octree=f.occ_bits is not None)
        logging.info(f'query grid:{tuple(sigma_dev.shape)}, valid:{int((sigma_dev != 1.0).sum().item())}')
        logging.info('Running iso-surface extraction')
        # the reference's skimage call (:1388-1394) with skimage's default method: marching cubes with Lewiner's topological
        # disambiguation -- skimage's triangles, one for one.  cfg mesh_extractor: 'cubes' = classic marching cubes (same vertices on
        # the grid edges, one fixed tiling per sign configuration), 'tetrahedra' = marching tetrahedra (~4x the triangles)
        kind = self.cfg.get('mesh_extractor', 'lewiner')
        if kind not in ('lewiner', 'cubes', 'tetrahedra'):             # (a configuration error is not an empty level set: raised, not logged)
            raise ValueError(f"mesh_extractor must be 'lewiner', 'cubes' or 'tetrahedra', not {kind!r}")
        extract = {'lewiner': marching_cubes_lewiner_gpu, 'cubes': marching_cubes_gpu, 'tetrahedra': marching_tetrahedra_gpu}[kind]
        try:
            vertices, triangles = extract(sigma_dev, isolevel)
        except Exception as e:
            logging.info(f"ERROR Marching Cubes {e}")
            return None
        logging.info(f'done V:{vertices.shape}, F:{triangles.shape}')
        step = np.array([tx[-1] - tx[0], ty[-1] - ty[0], tz[-1] - tz[0]]) / np.array([[len(tx) - 1, len(ty) - 1, len(tz) - 1]])
        offset = np.array([tx[0], ty[0], tz[0]])
        vertices = step.reshape(1, 3) * vertices + offset.reshape(1, 3)
        mesh = make_mesh(vertices, triangles)
        if return_sigma:
            gx, gy, gz = np.meshgrid(tx, ty, tz, indexing='ij')
            query_pts = torch.tensor(np.stack([gx, gy, gz], -1).astype(np.float32).reshape(-1, 3), device=self.device)
            return mesh, sigma_dev.cpu().numpy(), query_pts
        return mesh

    @torch.no_grad()
    def mesh_vertex_color_from_network(self, mesh):
        """nerf_runner.py:1412-1429: the colour net at every vertex (normalised space), zero view direction, frame 0's latent code;
        vertex colours = sigmoid of the logits, truncated to 8 bits.  One fused encode + MLP launch per million vertices."""
        pts = torch.from_numpy(np.ascontiguousarray(mesh.vertices, dtype=np.float32)).to(self.device)
        raw = self.field.query_network(pts, viewdir=(0.0, 0.0, 0.0), frame_id=0)
        colors = (torch.sigmoid(raw[:, :3]) * 255).to(torch.uint8).cpu().numpy()
        try:
            import trimesh
            if isinstance(mesh, trimesh.Trimesh):
                mesh.visual = trimesh.visual.ColorVisuals(mesh=mesh, face_colors=None, vertex_colors=colors)
                return mesh
        except ImportError:
            pass
        mesh.vertex_colors = colors
        return mesh

    @torch.no_grad()
    def mesh_texture_from_train_images(self, mesh, rgbs_raw, train_texture=False, tex_res=1024):
        """nerf_runner.py:1468-1542: project the raw training images onto the (normalised-space) mesh and average them per
        texel.  Per keyframe the reference renders the mesh's depth with pyrender, takes trimesh's closest point / triangle
        of every masked pixel, interpolates its UV (common.rayColorToTextureImageCUDA) and lets every texel take ONE colour
        per frame; here that is nof_texture_bake_frame (z-buffer rasteriser + barycentric UV + first-pixel-per-texel).  The UV
        parameterisation is a per-triangle atlas (Mesh.unwrap) instead of trimesh's xatlas unwrap.  Returns a Mesh with
        `.uv` and `.texture` (export('x.obj') writes .obj/.mtl/.png); like the reference, the texture image is flipped so that
        v grows upwards."""
        import ctypes as C
        from .mesh import Mesh
        assert len(self.images) == len(rgbs_raw)
        f = self.field
        dev = self.device
        ids = torch.as_tensor(self.frame_ids, device=dev)
        tf = f.c2w.view(-1, 4, 4)[ids]
        if self.models['pose_array'] is not None:
            tf = self.models['pose_array'].get_matrices(ids) @ tf
        tf = tf.cpu().numpy().astype(np.float64)
        m = Mesh(np.asarray(mesh.vertices), np.asarray(mesh.faces))
        m.merge_vertices()
        m.remove_duplicate_faces()
        m = m.unwrap(tex_res)
        H, W = tex_res, tex_res
        uvs_tex = torch.from_numpy((m.uv * np.array([W - 1, H - 1]).reshape(1, 2)).astype(np.float32)).to(dev).contiguous()
        verts = torch.from_numpy(m.vertices.astype(np.float32)).to(dev).contiguous()
        faces = torch.from_numpy(m.faces.astype(np.int64)).to(dev).contiguous()
        tex = torch.zeros(H, W, 3, device=dev)
        wtex = torch.zeros(H, W, device=dev)
        zbuf = torch.empty(self.H * self.W, dtype=torch.int64, device=dev)
        owner = torch.empty(H * W, dtype=torch.int32, device=dev)
        K4 = (C.c_float * 4)(float(self.K[0, 0]), float(self.K[1, 1]), float(self.K[0, 2]), float(self.K[1, 2]))
        min_depth = 0.1 * self.cfg['sc_factor']
        gl_in_cv_inv = np.linalg.inv(glcam_in_cvcam)
        for i in range(len(rgbs_raw)):
            cvcam_in_ob = tf[i] @ gl_in_cv_inv                      # nerf_runner.py:1501
            ob_in_cam = np.linalg.inv(cvcam_in_ob)[:3, :4].astype(np.float32)
            mask = torch.from_numpy(np.ascontiguousarray(self.masks[i].reshape(self.H, self.W) > 0).astype(np.uint8)).to(dev)
            rgb = torch.from_numpy(np.ascontiguousarray(rgbs_raw[i][..., :3], dtype=np.float32)).to(dev).contiguous()
            assert rgb.shape[:2] == (self.H, self.W), 'rgbs_raw must be the images the runner was trained on'
            lib.call('nof_texture_bake_frame', (C.c_float * 12)(*ob_in_cam.reshape(-1)), K4, self.H, self.W, verts, faces,
                     faces.shape[0], uvs_tex, mask, rgb, C.c_float(min_depth), tex_res, zbuf, owner, tex, wtex)
        img = torch.where(wtex[..., None] > 0, tex / wtex[..., None].clamp_min(1), torch.zeros_like(tex))
        img = np.clip(img.cpu().numpy(), 0, 255).astype(np.uint8)[::-1].copy()      # nerf_runner.py:1537-1540
        m.texture = img
        m.texture_coverage = float((wtex > 0).float().mean().item())
        return m

    # ---- checkpoint (nerf_runner.py:528-577) ----------------------------------------------------------------
    def save_weights(self, out_file, models=None, reference_format=False):
        """native: the flat buffers + Adam moments + occupancy bitfield; reference_format=True: the reference's own layout
        (nerf_runner.py:546-566: 'model' / 'embed_fn' / 'pose_array' / 'feature_array' state_dicts), loadable by it"""
        f = self.field
        f.gather_optimizer_state()               # (sharded optimiser: every rank's Adam moments, not only this rank's shard)
        if reference_format:
            torch.save(to_reference_checkpoint(f, self.global_step), out_file)
        else:
            torch.save({'global_step': self.global_step, 'params': f.params.cpu(), 'exp_avg': f.exp_avg.cpu(),
                        'exp_avg_sq': f.exp_avg_sq.cpu(), 'field_step': f.global_step, 'adam_steps': f.adam_steps,
                        'octree': (f.occ_bits.cpu() if f.occ_bits is not None else None, f.level, f.max_level)}, out_file)
        print('Saved checkpoints at', out_file)

    def load_weights(self, ckpt_path):
        """accepts both layouts: this implementation's, and checkpoints written by the reference's save_weights (parameters
        and Adam state; its kaolin octree bytes have no counterpart here)"""
        ck = torch.load(ckpt_path)
        f = self.field
        if 'params' not in ck and 'model' in ck:
            self.global_step = load_reference_checkpoint(f, ck)      # (sets f.adam_steps: the age of the Adam moments it found)
            f.global_step = self.global_step                         # schedules follow the runner's iteration count
            return
        f._packed_step = None                   # the MFMA weight image belongs to the old parameters
        f.grads.zero_()
        f.params.copy_(ck['params'].to(f.device))
        f.exp_avg.copy_(ck['exp_avg'].to(f.device))
        f.exp_avg_sq.copy_(ck['exp_avg_sq'].to(f.device))
        f.global_step = ck['field_step']
        f.adam_steps = ck.get('adam_steps', ck['field_step'])
        self.global_step = ck['global_step']
