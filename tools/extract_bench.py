"""Renderer-side timing on the GPU box (BASELINE cfg4: dense extraction up to 512^3): fused SDF-grid query + device
marching tetrahedra on the field bench.py trains (a few hundred steps so that a surface exists)."""
import sys, os, argparse, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from bundlesdf_amd.mesh_gpu import marching_tetrahedra_gpu

ap = argparse.ArgumentParser()
ap.add_argument('--keyframes', type=int, default=16)
ap.add_argument('--train-steps', type=int, default=300)
ap.add_argument('--sizes', type=int, nargs='+', default=[128, 256, 512])
a = ap.parse_args()
args = argparse.Namespace(keyframes=a.keyframes, height=480, width=640, rays=4096, log2_T=19, mlp='baseline', precision='bf16')
torch.cuda.set_device(0)
runner, cfg = bench.build_runner(args, 0, 1, torch.device('cuda', 0))
for _ in range(a.train_steps):
    runner.train_loop(); runner.global_step += 1
torch.cuda.synchronize()
fld = runner.field
bounds = np.array(cfg['bounding_box']).reshape(2, 3)
out = []
for n in a.sizes:
    axes = [np.linspace(bounds[0, d], bounds[1, d], n + 1)[:-1] + 0.5 * (bounds[1, d] - bounds[0, d]) / n for d in range(3)]
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        vol = fld.query_sdf_grid(*axes)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        v, f = marching_tetrahedra_gpu(vol, 0.0)
        t2 = time.perf_counter()
    occ = float((vol != 1.0).float().mean().item())
    rec = dict(grid=n, voxels=n ** 3, occupied_frac=round(occ, 4), query_ms=round((t1 - t0) * 1e3, 3),
               occupied_voxels_per_s=round(occ * n ** 3 / (t1 - t0)), extract_ms=round((t2 - t1) * 1e3, 3), V=len(v), F=len(f))
    print(json.dumps(rec), flush=True)
    del vol
