"""End-to-end quality at the benchmark configuration (BASELINE cfg2: 64 synthetic 640x480 keyframes, 4096 rays x 192
samples, L=16 T=2^19, SDF 3x64 + colour 2x64): train, extract the mesh, Chamfer distance to the analytic ellipsoid, pose error
before / after -- one JSON line per --precision (same seed, same data: the exact-fp32 MFMA mode beside the 16-bit modes shows what
the operand precision does to the CONVERGED field).  Run on the GPU box."""
import sys, os, json, time, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from scipy.spatial import cKDTree
import bench
from bundlesdf_amd import synthetic
from bundlesdf_amd.mesh import largest_component
from bundlesdf_amd.nerf_runner import get_optimized_poses_in_real_world, mesh_to_real_world

ap = argparse.ArgumentParser()
ap.add_argument('--steps', type=int, default=3000)
ap.add_argument('--keyframes', type=int, default=64)
ap.add_argument('--voxel', type=float, default=0.002)
ap.add_argument('--precision', default='fp16x3')
a = ap.parse_args()
args = argparse.Namespace(keyframes=a.keyframes, height=480, width=640, rays=4096, log2_T=19, mlp='baseline', precision=a.precision,
                          finest=256)
torch.cuda.set_device(0)
t0 = time.time()
runner, cfg = bench.build_runner(args, 0, 1, torch.device('cuda', 0))
cfg['n_step'] = a.steps
runner.cfg['n_step'] = a.steps
runner.N_iters = a.steps + 1
t1 = time.time()
runner.train_loop(); first = runner.field.losses(); runner.global_step += 1
torch.cuda.synchronize(); t2 = time.time()
for _ in range(a.steps - 1):
    runner.train_loop(); runner.global_step += 1
torch.cuda.synchronize(); t3 = time.time()
last = runner.field.losses()
pool = synthetic.make_pool(n_frames=a.keyframes, H=480, W=640, fx=600.0, seed=0, analytic_bounds=True)
poses_opt, offset = get_optimized_poses_in_real_world(pool['poses'].copy(), runner.models['pose_array'], cfg['sc_factor'], cfg['translation'])
gt = pool['poses_gt'] @ np.diag([1.0, -1.0, -1.0, 1.0])
noisy = pool['poses'].copy(); noisy[:, :3, 3] = noisy[:, :3, 3] / cfg['sc_factor'] - cfg['translation']; noisy = noisy @ np.diag([1.0, -1.0, -1.0, 1.0])
e0 = np.linalg.norm(noisy[1:, :3, 3] - gt[1:, :3, 3], axis=1).mean()
e1 = np.linalg.norm(poses_opt[1:, :3, 3] - gt[1:, :3, 3], axis=1).mean()
t4 = time.time()
mesh = runner.extract_mesh(isolevel=0, voxel_size=a.voxel)
torch.cuda.synchronize(); t5 = time.time()
mesh = largest_component(mesh_to_real_world(mesh, pose_offset=offset, translation=cfg['translation'], sc_factor=cfg['sc_factor']))
v, f = np.asarray(mesh.vertices), np.asarray(mesh.faces)
aa, bb, cc = v[f[:, 0]], v[f[:, 1]], v[f[:, 2]]
area = 0.5 * np.linalg.norm(np.cross(bb - aa, cc - aa), axis=1)
rng = np.random.default_rng(0)
idx = rng.choice(len(area), size=40000, p=area / area.sum())
r1, r2 = np.sqrt(rng.random(40000)), rng.random(40000)
ms = (1 - r1)[:, None] * aa[idx] + (r1 * (1 - r2))[:, None] * bb[idx] + (r1 * r2)[:, None] * cc[idx]
p = rng.normal(size=(40000, 3)); p /= np.linalg.norm(p, axis=1, keepdims=True); gs = p * pool['semi_axes']
d1, _ = cKDTree(ms).query(gs); d2, _ = cKDTree(gs).query(ms)
print(json.dumps(dict(precision=a.precision, steps=a.steps, keyframes=a.keyframes, setup_s=round(t1 - t0, 2), train_s=round(t3 - t2, 3),
                      ms_per_step=round((t3 - t2) / (a.steps - 1) * 1e3, 4), loss_first=first['loss'], loss_last=last['loss'],
                      sdf_loss_last=last['sdf_loss'], pose_err_before_mm=round(e0 * 1e3, 3), pose_err_after_mm=round(e1 * 1e3, 3),
                      extract_s=round(t5 - t4, 3), voxel_mm=a.voxel * 1e3, V=len(v), F=len(f),
                      chamfer_mm=round(0.5 * (d1.mean() + d2.mean()) * 1e3, 4), gt_to_mesh_mm=round(d1.mean() * 1e3, 4),
                      mesh_to_gt_mm=round(d2.mean() * 1e3, 4), flags=int(runner.field.flags[0].item()))))
