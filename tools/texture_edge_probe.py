"""Which pixels of tests/test_gpu_texture.py::test_texture_bake_frame_matches_oracle get another triangle from the device rasteriser than
from the float64 ray caster of oracle/texture_oracle.py?  For every such pixel: the float64 barycentric coordinates of the pixel ray on
BOTH triangles and the depth along the ray (development probe, round 6: VERDICT r5 weak 1d)."""
import ctypes as C
import sys, os
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bundlesdf_amd import lib as nof
from bundlesdf_amd.mesh import Mesh
from bundlesdf_amd.synthetic import look_at_cv
from oracle import texture_oracle as TO
from tests import util as U
from tests.test_gpu_texture import _ellipsoid_mesh


def bary(o, d, A, B, Cc):
    e1, e2 = B - A, Cc - A
    p = np.cross(d, e2); det = p @ e1
    tv = o - A
    u = (p @ tv) / det
    q = np.cross(tv, e1)
    w = (d @ q) / det
    t = (e2 @ q) / det
    return u, w, 1 - u - w, t


nof.load()
V, F = _ellipsoid_mesh()
T, H, W = 96, 60, 80
m = Mesh(V, F).unwrap(T)
verts, faces = m.vertices.astype(np.float32), m.faces
uvs_tex = (m.uv * (T - 1)).astype(np.float32)
K = np.array([[90.0, 0, 40.0], [0, 90.0, 30.0], [0, 0, 1]])
rng = np.random.default_rng(1)
tex = torch.zeros(T, T, 3, device='cuda'); wtex = torch.zeros(T, T, device='cuda')
zbuf = torch.empty(H * W, dtype=torch.int64, device='cuda'); owner = torch.empty(T * T, dtype=torch.int32, device='cuda')
tex_o, wtex_o = np.zeros((T, T, 3)), np.zeros((T, T))
K4 = (C.c_float * 4)(90.0, 90.0, 40.0, 30.0)
for cam in ([1.6, 0.2, 0.3], [-0.4, 1.5, -0.5]):
    cam_in_ob = look_at_cv(np.array(cam)); ob_in_cam = np.linalg.inv(cam_in_ob)
    rgb = rng.integers(0, 255, size=(H, W, 3)).astype(np.float32)
    mask = (rng.random((H, W)) > 0.1).astype(np.uint8)
    nof.call('nof_texture_bake_frame', (C.c_float * 12)(*ob_in_cam[:3, :4].astype(np.float32).reshape(-1)), K4, H, W,
             U.dev(verts), U.dev(faces), faces.shape[0], U.dev(uvs_tex), U.dev(mask), U.dev(rgb), C.c_float(0.05), T, zbuf, owner, tex, wtex)
    torch.cuda.synchronize()
    tri, depth = TO.bake_frame(ob_in_cam, K, H, W, verts, faces, uvs_tex, mask, rgb, 0.05, T, tex_o, wtex_o)
    z = zbuf.cpu().numpy().astype(np.uint64)
    got = np.where(z == np.uint64(0xFFFFFFFFFFFFFFFF), -1, (z & np.uint64(0xFFFFFFFF)).astype(np.int64)).reshape(H, W)
    R, t = ob_in_cam[:3, :3], ob_in_cam[:3, 3]
    o = -R.T @ t
    bad = np.argwhere(got != tri)
    print(f'camera {cam}: {len(bad)} of {H * W} pixels differ ({(tri >= 0).sum()} on the mesh)')
    v64 = verts.astype(np.float64)
    for (y, x) in bad:
        d = R.T @ np.array([(x - K[0, 2]) / K[0, 0], (y - K[1, 2]) / K[1, 1], 1.0])
        row = f'  pixel ({y},{x}) oracle tri {tri[y, x]} device tri {got[y, x]}'
        for name, f in (('oracle', tri[y, x]), ('device', got[y, x])):
            if f >= 0:
                u, w, s, tt = bary(o, d, *(v64[faces[f, k]] for k in range(3)))
                row += f' | {name}: min bary {min(u, w, s):+.2e} depth {tt:.5f}'
        print(row)
w_got, t_got = wtex.cpu().numpy(), tex.cpu().numpy()
cover = (w_got > 0) | (wtex_o > 0)
agree = (w_got == wtex_o) & (np.abs(t_got - tex_o).max(-1) < 1e-3)
print(f'texels: {cover.sum()} covered, {int((~agree[cover]).sum())} differ (weights differ in {(w_got != wtex_o)[cover].sum()}, colours in {(np.abs(t_got - tex_o).max(-1) >= 1e-3)[cover].sum()})')
for (v, u) in np.argwhere(cover & ~agree)[:40]:
    print(f'  texel ({v},{u}): device w {w_got[v, u]} colour {t_got[v, u]} | oracle w {wtex_o[v, u]} colour {tex_o[v, u]}')
