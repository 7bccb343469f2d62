"""Golden vectors for the plugin-surface helpers the tracking loop calls around the Neural Object Field
(bundlesdf.py:148-170,231-235): the reference's OWN functions are cut out of the read-only mount with `ast` and executed on
CPU (pure NumPy, no CUDA):

    preprocess_data                       nerf_helpers.py:218-240
    get_optimized_poses_in_real_world     Utils.py:479-505
    mesh_to_real_world                    Utils.py:508-514
    NerfRunner.build_octree               nerf_runner.py:436-489 (reference-driven: kaolin's OctreeManager stubbed)
    NerfRunner.make_frame_rays            nerf_runner.py:246-316 (reference-driven: cv2.dilate stubbed, use_octree off)
    NeRFSmall.state_dict()                nerf_helpers.py:243-294: key names, order, shapes (the checkpoint layout)
    glcam_in_cvcam, BAD_DEPTH, BAD_COLOR  Utils.py:34-40

Run here (needs /root/reference):  python tests/golden/make_golden_plugin.py  ->  tests/golden/plugin_vectors.npz
The fixture travels; the generator does not need to."""
import ast
import os
import textwrap

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = '/root/reference'


def cut(path, names):
    src = open(os.path.join(REF, path)).read()
    tree = ast.parse(src)
    out = []
    for n in tree.body:
        if isinstance(n, ast.FunctionDef) and n.name in names:
            out.append('\n'.join(src.splitlines()[n.lineno - 1:n.end_lineno]))
    assert len(out) == len(names), (path, names)
    return out


def constants(path, names):
    """top-level `NAME = <literal expression>` assignments"""
    src = open(os.path.join(REF, path)).read()
    out = {}
    for n in ast.parse(src).body:
        if isinstance(n, ast.Assign) and len(n.targets) == 1 and isinstance(n.targets[0], ast.Name) and n.targets[0].id in names:
            out[n.targets[0].id] = '\n'.join(src.splitlines()[n.lineno - 1:n.end_lineno])
    assert set(out) == set(names), (path, names, list(out))
    return out


class PoseStub:
    """stands in for PoseArray: get_matrices(ids) -> torch [n,4,4] (nerf_helpers.py:143-154)"""

    def __init__(self, mats):
        self.mats = torch.as_tensor(mats)

    def get_matrices(self, ids):
        return self.mats[torch.as_tensor(ids).long()]


class MeshStub:
    def __init__(self, v):
        self.vertices = v

    def apply_transform(self, T):
        self.vertices = self.vertices @ T[:3, :3].T + T[:3, 3]


def main():
    ns = {'np': np, 'torch': torch}
    for code in constants('Utils.py', ['BAD_DEPTH', 'BAD_COLOR', 'glcam_in_cvcam']).values():
        exec(code, ns)
    for code in cut('nerf_helpers.py', ['preprocess_data']) + cut('Utils.py', ['get_optimized_poses_in_real_world', 'mesh_to_real_world']):
        exec(code, ns)
    rng = np.random.default_rng(0)
    out = {'BAD_DEPTH': np.float64(ns['BAD_DEPTH']), 'BAD_COLOR': np.asarray(ns['BAD_COLOR'], dtype=np.float64),
           'glcam_in_cvcam': np.asarray(ns['glcam_in_cvcam'], dtype=np.float64)}
    # ---- preprocess_data ----
    N, H, W = 3, 6, 8
    rgbs = rng.integers(0, 256, size=(N, H, W, 3)).astype(np.float32)
    depths = rng.uniform(0.0, 1.5, size=(N, H, W)).astype(np.float32)
    depths[0, 0, :3] = 0.05                                   # below the 0.1 m validity threshold
    masks = (rng.random((N, H, W)) < 0.7).astype(np.uint8)
    normals = rng.normal(size=(N, H, W, 3)).astype(np.float32)
    poses = np.tile(np.eye(4), (N, 1, 1))
    poses[:, :3, 3] = rng.normal(size=(N, 3))
    sc, tr = 3.7, np.array([0.01, -0.02, 0.3])
    out.update(pp_rgbs=rgbs.copy(), pp_depths=depths.copy(), pp_masks=masks.copy(), pp_normals=normals.copy(), pp_poses=poses.copy(),
               pp_sc=np.float64(sc), pp_tr=tr)
    r = ns['preprocess_data'](rgbs.copy(), depths.copy(), masks.copy(), normals.copy(), poses.copy(), sc, tr)
    for k, v in zip(('rgbs', 'depths', 'masks', 'normals', 'poses'), r):
        out['pp_out_' + k] = np.asarray(v)
    # ---- get_optimized_poses_in_real_world ----
    F = 5
    pn = np.tile(np.eye(4), (F, 1, 1))
    for i in range(F):
        w = rng.normal(size=3)
        w /= np.linalg.norm(w)
        th = rng.uniform(0, 1.0)
        Kx = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
        pn[i, :3, :3] = np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * Kx @ Kx
        pn[i, :3, 3] = rng.uniform(-0.6, 0.6, 3)
    delta = np.tile(np.eye(4), (F, 1, 1)).astype(np.float32)
    delta[1:, :3, 3] = rng.uniform(-0.02, 0.02, (F - 1, 3))
    for i in range(1, F):
        a = rng.uniform(-0.05, 0.05)
        delta[i, :2, :2] = [[np.cos(a), -np.sin(a)], [np.sin(a), np.cos(a)]]
    opt, offset = ns['get_optimized_poses_in_real_world'](pn.copy(), PoseStub(delta), sc, tr)
    out.update(gp_poses=pn, gp_delta=delta, gp_out=np.asarray(opt), gp_offset=np.asarray(offset))
    # ---- mesh_to_real_world ----
    V = rng.normal(size=(50, 3))
    m = ns['mesh_to_real_world'](MeshStub(V.copy()), np.asarray(offset, dtype=np.float64), tr, sc)
    out.update(mw_v=V, mw_out=np.asarray(m.vertices))
    # ---- NerfRunner.build_octree (nerf_runner.py:436-489), reference-driven: the method itself runs on CPU with
    #      .cuda() patched to identity and a stub OctreeManager that records what kaolin would have been given ----
    src = open(os.path.join(REF, 'nerf_runner.py')).read()
    tree = ast.parse(src)
    cls = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == 'NerfRunner')
    fn = next(n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name == 'build_octree')
    code = textwrap.dedent('\n'.join(src.splitlines()[fn.lineno - 1:fn.end_lineno]))
    captured = {}

    class OctreeManager:
        def __init__(self, pts, max_level):
            captured['pts'], captured['max_level'] = pts.detach().cpu().numpy().copy(), int(max_level)

        def draw_boxes(self, **kw):
            pass
    import logging
    ns2 = {'np': np, 'torch': torch, 'logging': logging, 'OctreeManager': OctreeManager}
    exec(code, ns2)
    torch.Tensor.cuda = lambda self, *a, **k: self
    for tag, (npts, sv, dil, rv, scf) in {'a': (400, 0.02, 0.02, 0.02, 5.25), 'b': (150, 0.01, 0.03, 0.04, 3.1)}.items():
        cloud = (rng.normal(size=(npts, 3)) * 0.25).clip(-1, 1).astype(np.float64)
        cloud[0] = [1.0, -1.0, 0.999]                          # corner: the dilation leaves the cube and is clipped back

        class Stub:
            pass
        st = Stub()
        st.cfg = dict(save_octree_clouds=False, save_dir=None, octree_smallest_voxel_size=sv, octree_dilate_size=dil,
                      octree_raytracing_voxel_size=rv, sc_factor=scf)
        st.build_octree_pts, st._run = cloud, None
        ns2['build_octree'](st)
        out[f'oct_{tag}_cfg'] = np.array([sv, dil, rv, scf])
        out[f'oct_{tag}_cloud'] = cloud
        out[f'oct_{tag}_pts'] = captured['pts']
        out[f'oct_{tag}_max_level'] = np.int64(captured['max_level'])
    # ---- NerfRunner.make_frame_rays (nerf_runner.py:246-316), reference-driven: the method and its helpers run on CPU.
    #      cv2.dilate is the one stub (cv2 is not installed): all-ones k x k kernel anchored at k//2, borders ignored, written
    #      here as a plain double loop; use_octree is off (the kaolin tracer cannot run here) ----
    import types
    import sys
    sys.path.insert(0, HERE)
    import make_golden as MG
    ns3 = MG.build_namespace()

    def cv2_dilate(mask, kernel, iterations=1):
        k = kernel.shape[0]
        lo, hi = -(k // 2), k - 1 - k // 2
        Hh, Ww = mask.shape
        outm = np.zeros_like(mask)
        for v in range(Hh):
            v0, v1 = max(v + lo, 0), min(v + hi, Hh - 1)
            for u in range(Ww):
                u0, u1 = max(u + lo, 0), min(u + hi, Ww - 1)
                outm[v, u] = mask[v0:v1 + 1, u0:u1 + 1].max()
        return outm
    ns3['cv2'] = types.SimpleNamespace(dilate=cv2_dilate)
    code = textwrap.dedent('\n'.join(src.splitlines()[
        next(n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name == 'make_frame_rays').lineno - 1:
        next(n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name == 'make_frame_rays').end_lineno]))
    exec(code, ns3)
    Hh, Ww, Fn = 36, 48, 2
    imgs = rng.random((Fn, Hh, Ww, 3)).astype(np.float32)
    msk = np.zeros((Fn, Hh, Ww, 1), dtype=np.uint8)
    msk[0, 10:22, 14:30] = 1
    msk[1, 5:30, 20:40] = 1
    scf = 2.5
    dep = (rng.uniform(0.3, 0.9, (Fn, Hh, Ww, 1)) * scf).astype(np.float32)
    dep[0, 12:14, 16:20] = 99 * scf                            # invalid depth inside the mask -> ray type 1
    dep[1, 6, 21] = 0.01 * scf
    Kc = np.array([[40.0, 0, 24.0], [0, 40.0, 18.0], [0, 0, 1]])
    ps = np.tile(np.eye(4), (Fn, 1, 1))
    ps[0, :3, 3] = [0.1, -0.2, 1.6]
    ps[1, :3, :3] = pn[1, :3, :3]
    ps[1, :3, 3] = [-0.3, 0.2, 1.4]

    class Stub2:
        pass
    for valid_only in (1, 0):
        st = Stub2()
        st.images, st.depths, st.masks, st.normal_maps, st.occ_masks, st.poses, st.K, st.H, st.W = imgs, dep, msk, None, None, ps, Kc, Hh, Ww
        st.cfg = dict(near=0.1, far=1.0, sc_factor=scf, down_scale_ratio=4, rays_valid_depth_only=valid_only, use_octree=0,
                      bounding_box=[[-1, -1, -1], [1, 1, 1]])
        for fid in range(Fn):
            out[f'mfr_{valid_only}_{fid}'] = np.asarray(ns3['make_frame_rays'](st, fid))
    out.update(mfr_images=imgs, mfr_depths=dep, mfr_masks=msk, mfr_poses=ps, mfr_K=Kc, mfr_sc=np.float64(scf))
    # ---- NeRFSmall.state_dict(): the key names / shapes / order a reference checkpoint carries (nerf_runner.py:546-550) ----
    for tag, (nl, nlc, ichv) in {'ref': (2, 3, 9), 'base': (3, 2, 11)}.items():
        torch.manual_seed(3)
        net = ns3['NeRFSmall'](num_layers=nl, hidden_dim=64, geo_feat_dim=15, num_layers_color=nlc, hidden_dim_color=64,
                               input_ch=32, input_ch_views=ichv)
        sd = net.state_dict()
        out[f'sd_{tag}_keys'] = np.array(list(sd.keys()))
        out[f'sd_{tag}_flat'] = torch.cat([p.detach().reshape(-1) for p in net.parameters()]).numpy()
        out[f'sd_{tag}_cat'] = torch.cat([sd[k].reshape(-1) for k in sd.keys()]).numpy()     # state_dict order
        out[f'sd_{tag}_shapes'] = np.array([list(sd[k].shape) + [0] * (2 - sd[k].dim()) for k in sd.keys()])
        xin = torch.from_numpy(rng.normal(size=(7, 32 + ichv)).astype(np.float32))
        out[f'sd_{tag}_x'] = xin.numpy()
        out[f'sd_{tag}_y'] = net(xin).detach().numpy()
    np.savez_compressed(os.path.join(HERE, 'plugin_vectors.npz'), **out)
    print('wrote', len(out), 'arrays')


if __name__ == '__main__':
    main()
