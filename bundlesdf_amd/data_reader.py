"""On-disk formats either side of the Neural Object Field, read with PIL / NumPy / PyYAML only.

    YcbineoatReader     the custom-data capture layout  <dir>/rgb/*.png  depth/*.png (uint16 mm)  masks/*.png  cam_K.txt
                        [annotated_poses/*  masks_hand/*  masks_hand_right/*]   (BundleTrack/scripts/data_reader.py:21-105,
                        readme.md:59-65); same attributes and methods, same resize convention (cv2.INTER_NEAREST)
    TrackerOutput       what the tracker leaves for the global refine and what run_global_nerf reads (bundlesdf.py:640-688,
                        Bundler.cpp:959-1084): cam_K.txt, ob_in_cam/<id>.txt, <last id>/keyframes.yml (keyframe_<id>: {cam_in_ob:
                        [16 floats]}), color_segmented/<id>.png, depth_filtered/<id>.png (uint16 mm), mask/<id>.png
    write_capture / write_tracker_output   the inverse, used to lay synthetic data out in those formats
"""
import glob
import logging
import os

import numpy as np


def read_png(path):
    """PNG -> array as stored: [H,W,3|4] uint8 (RGB order, like imageio), [H,W] uint8 or [H,W] uint16."""
    from PIL import Image
    with Image.open(path) as im:
        if im.mode in ('I;16', 'I;16B', 'I'):
            return np.array(im).astype(np.uint16)
        if im.mode == 'P':
            im = im.convert('RGB')
        return np.array(im)


def read_depth_png(path):
    """uint16 millimetres -> float64 metres (cv2.imread(path, -1) / 1e3)."""
    return read_png(path).astype(np.float64) / 1e3


def write_png(path, arr):
    from PIL import Image
    arr = np.asarray(arr)
    if arr.dtype == np.uint16:
        Image.fromarray(arr.astype(np.uint16)).save(path)                # 16-bit greyscale
    else:
        Image.fromarray(arr.astype(np.uint8)).save(path)


def resize_nearest(img, W, H):
    """cv2.resize(img, (W, H), interpolation=cv2.INTER_NEAREST): dst[y, x] = src[floor(y * H0/H), floor(x * W0/W)]."""
    H0, W0 = img.shape[:2]
    if (H0, W0) == (H, W):
        return img
    ys = np.minimum(np.floor(np.arange(H) * (H0 / H)).astype(np.int64), H0 - 1)
    xs = np.minimum(np.floor(np.arange(W) * (W0 / W)).astype(np.int64), W0 - 1)
    return img[ys][:, xs]


class YcbineoatReader:
    """Reader of the custom-data capture layout.  Attributes callers use: video_dir, color_files, id_strs, K (already scaled),
    H, W (already scaled), downscale, gt_pose_files."""

    def __init__(self, video_dir, downscale=1, shorter_side=None):
        def listing(sub, pattern):
            return sorted(glob.glob(os.path.join(video_dir, sub, pattern)))
        self.video_dir = video_dir
        self.color_files = listing('rgb', '*.png')
        self.gt_pose_files = listing('annotated_poses', '*')
        self.id_strs = [os.path.splitext(os.path.basename(f))[0] for f in self.color_files]
        full_h, full_w = read_png(self.color_files[0]).shape[:2]
        # `shorter_side` overrides `downscale`: the factor that brings min(H, W) to it; sizes truncate like int()
        self.downscale = downscale if shorter_side is None else shorter_side / min(full_h, full_w)
        self.H, self.W = int(full_h * self.downscale), int(full_w * self.downscale)
        intrinsics = np.loadtxt(os.path.join(video_dir, 'cam_K.txt')).reshape(3, 3)
        intrinsics[:2] *= self.downscale                              # fx, fy, cx, cy scale with the image; the last row stays
        self.K = intrinsics

    def get_video_name(self):
        return self.video_dir.split('/')[-1]

    def __len__(self):
        return len(self.color_files)

    def get_gt_pose(self, i):
        try:
            return np.loadtxt(self.gt_pose_files[i]).reshape(4, 4)
        except Exception:
            logging.info("GT pose not found, return None")
            return None

    def get_color(self, i):
        return resize_nearest(read_png(self.color_files[i]), self.W, self.H)

    def get_mask(self, i):
        mask = read_png(self.color_files[i].replace('rgb', 'masks'))
        if len(mask.shape) == 3:
            mask = (mask.sum(axis=-1) > 0).astype(np.uint8)
        return resize_nearest(mask, self.W, self.H)

    def get_depth(self, i):
        return resize_nearest(read_depth_png(self.color_files[i].replace('rgb', 'depth')), self.W, self.H)

    def get_xyz_map(self, i):
        from .scene import depth2xyzmap
        return depth2xyzmap(self.get_depth(i), self.K)

    def get_occ_mask(self, i):
        occ_mask = None
        for sub in ('masks_hand', 'masks_hand_right'):
            f = self.color_files[i].replace('rgb', sub)
            if os.path.exists(f):
                m = read_png(f)
                m = (m.sum(axis=-1) if m.ndim == 3 else m) > 0
                occ_mask = m if occ_mask is None else (occ_mask | m)
        if occ_mask is None:
            return np.zeros((self.H, self.W), dtype=np.uint8)
        return resize_nearest(occ_mask, self.W, self.H).astype(np.uint8)


class TrackerOutput:
    """The tracker's debug directory as run_global_nerf consumes it (bundlesdf.py:640-688)."""

    def __init__(self, debug_dir):
        import yaml
        self.debug_dir = debug_dir
        self.K = np.loadtxt(f'{debug_dir}/cam_K.txt').reshape(3, 3)
        stamps = sorted(glob.glob(f"{debug_dir}/ob_in_cam/*"))
        if not stamps:
            raise FileNotFoundError(f'{debug_dir}/ob_in_cam is empty')
        self.last_stamp = os.path.basename(stamps[-1]).replace('.txt', '')
        with open(f'{debug_dir}/{self.last_stamp}/keyframes.yml') as f:
            self.keyframes = yaml.safe_load(f)
        self.keys = list(self.keyframes.keys())

    def select(self, n_train_image, rng=np.random):
        """bundlesdf.py:648-651: the first keyframe plus a random subset when there are more than n_train_image."""
        keys = self.keys
        if len(keys) > n_train_image:
            keys = [keys[0]] + list(rng.choice(keys, n_train_image, replace=False))
            keys = list(set(keys))
        return keys

    def load(self, keys=None):
        """-> dict(frame_ids, cam_in_obs [F,4,4] OpenCV cam-in-object, rgbs [F,H,W,3] uint8, depths [F,H,W] metres, masks [F,H,W])"""
        keys = self.keys if keys is None else keys
        frame_ids = [k.replace('keyframe_', '') for k in keys]
        cam_in_obs = np.array([np.array(self.keyframes[k]['cam_in_ob']).reshape(4, 4) for k in keys])
        rgbs, depths, masks = [], [], []
        for fid in frame_ids:
            rgb_file = f"{self.debug_dir}/color_segmented/{fid}.png"
            rgbs.append(read_png(rgb_file)[..., :3])
            depths.append(read_depth_png(rgb_file.replace('color_segmented', 'depth_filtered')))
            masks.append(read_png(rgb_file.replace('color_segmented', 'mask')))
        return dict(frame_ids=frame_ids, cam_in_obs=cam_in_obs, rgbs=np.array(rgbs), depths=np.array(depths), masks=np.array(masks))


def write_capture(video_dir, rgbs, depths, masks, K, id_strs=None, poses=None):
    """rgbs [F,H,W,3] uint8, depths [F,H,W] metres, masks [F,H,W] -> the YcbineoatReader layout."""
    for sub in ('rgb', 'depth', 'masks'):
        os.makedirs(f'{video_dir}/{sub}', exist_ok=True)
    np.savetxt(f'{video_dir}/cam_K.txt', np.asarray(K).reshape(3, 3))
    F = len(rgbs)
    id_strs = id_strs or [f'{i:07d}' for i in range(F)]
    for i, s in enumerate(id_strs):
        write_png(f'{video_dir}/rgb/{s}.png', np.asarray(rgbs[i], np.uint8))
        write_png(f'{video_dir}/depth/{s}.png', np.round(np.asarray(depths[i]) * 1e3).astype(np.uint16))
        write_png(f'{video_dir}/masks/{s}.png', (np.asarray(masks[i]) > 0).astype(np.uint8) * 255)
    if poses is not None:
        os.makedirs(f'{video_dir}/annotated_poses', exist_ok=True)
        for s, p in zip(id_strs, poses):
            np.savetxt(f'{video_dir}/annotated_poses/{s}.txt', np.asarray(p).reshape(4, 4))
    return id_strs


def write_tracker_output(debug_dir, rgbs, depths, masks, K, cam_in_obs, id_strs=None):
    """the files Bundler::saveNewframeResult leaves (Bundler.cpp:959-1084) for the given keyframes"""
    import yaml
    F = len(rgbs)
    id_strs = id_strs or [f'{i:07d}' for i in range(F)]
    for sub in ('ob_in_cam', 'color_segmented', 'depth_filtered', 'mask', id_strs[-1]):
        os.makedirs(f'{debug_dir}/{sub}', exist_ok=True)
    np.savetxt(f'{debug_dir}/cam_K.txt', np.asarray(K).reshape(3, 3))
    node = {}
    for i, s in enumerate(id_strs):
        write_png(f'{debug_dir}/color_segmented/{s}.png', np.asarray(rgbs[i], np.uint8))
        write_png(f'{debug_dir}/depth_filtered/{s}.png', np.round(np.asarray(depths[i]) * 1e3).astype(np.uint16))
        write_png(f'{debug_dir}/mask/{s}.png', (np.asarray(masks[i]) > 0).astype(np.uint8) * 255)
        np.savetxt(f'{debug_dir}/ob_in_cam/{s}.txt', np.linalg.inv(np.asarray(cam_in_obs[i]).reshape(4, 4)))
        node[f'keyframe_{s}'] = {'cam_in_ob': [float(v) for v in np.asarray(cam_in_obs[i], np.float32).reshape(-1)]}
    with open(f'{debug_dir}/{id_strs[-1]}/keyframes.yml', 'w') as f:
        yaml.safe_dump(node, f, default_flow_style=None, sort_keys=False)
    return id_strs
