"""CPU restatement (NumPy, float64) of one keyframe of the texture bake -- TEST INFRASTRUCTURE (see oracle/__init__.py).

NerfRunner.mesh_texture_from_train_images, nerf_runner.py:1499-1535: the mesh's depth is rendered from the keyframe's pose
(pyrender there; here brute-force ray / triangle intersection, nearest hit), pixels that are masked and whose rendered depth
is >= 0.1*sc_factor take the visible surface point and its triangle (trimesh.proximity.closest_point of the back-projected
depth there; the intersection point here -- the same point for a closed mesh), the point's UV is the barycentric blend of the
triangle's vertex UVs (calculateBarycentricCoordinate3DKernel, common.cu:171-185), rounded half-to-even (torch.round) to a
texel; every texel takes ONE colour per frame -- the first pixel in row-major order -- with weight 1 (:1527-1535).  The
reference flattens texels with (W-1) and decodes with (W-1) (:1528,:1532); that quirk is kept.
"""
import numpy as np


def bake_frame(ob_in_cam, K, H, W, verts, faces, uvs_tex, mask, rgb, min_depth, tex_res, tex, wtex):
    """Accumulates into tex [T,T,3], wtex [T,T] (float64).  Returns (tri_id [H,W] (-1 = no hit), depth [H,W])."""
    R, t = ob_in_cam[:3, :3].astype(np.float64), ob_in_cam[:3, 3].astype(np.float64)
    v = verts.astype(np.float64)
    A, B, C = v[faces[:, 0]], v[faces[:, 1]], v[faces[:, 2]]
    o = -R.T @ t
    ys, xs = np.meshgrid(np.arange(H), np.arange(W), indexing='ij')
    dc = np.stack([(xs - K[0, 2]) / K[0, 0], (ys - K[1, 2]) / K[1, 1], np.ones_like(xs, dtype=np.float64)], -1).reshape(-1, 3)
    d = dc @ R                                               # R^T dc per row
    e1, e2 = B - A, C - A
    tri = -np.ones(H * W, dtype=np.int64)
    depth = np.full(H * W, np.inf)
    loc = np.zeros((H * W, 3))
    for f in range(len(faces)):                              # Moeller-Trumbore against every pixel ray
        p = np.cross(d, e2[f])
        det = p @ e1[f]
        ok = np.abs(det) > 1e-18
        inv = np.where(ok, 1.0 / np.where(ok, det, 1.0), 0.0)
        tv = o - A[f]
        u = (p @ tv) * inv
        q = np.cross(tv, e1[f])
        w = (d @ q) * inv
        tt = (e2[f] @ q) * inv
        hit = ok & (u >= 0) & (w >= 0) & (u + w <= 1) & (tt > 0)
        z = tt                                               # dc_z == 1: the ray parameter is the camera-space depth
        better = hit & (z < depth)
        depth[better] = z[better]
        tri[better] = f
        loc[better] = o + tt[better, None] * d[better]
    valid = (tri >= 0) & (mask.reshape(-1) > 0) & (depth >= min_depth)
    T = tex_res
    taken = set()
    for pix in np.flatnonzero(valid):                        # row-major: the first pixel of a texel owns it
        f = tri[pix]
        a, b, c, p = A[f], B[f], C[f], loc[pix]
        nrm = np.cross(b - c, b - a)
        area = nrm @ np.cross(b - a, c - a)
        w0 = nrm @ np.cross(b - p, c - p) / area
        w1 = nrm @ np.cross(c - p, a - p) / area
        w2 = 1 - w0 - w1
        uv = uvs_tex[faces[f, 0]] * w0 + uvs_tex[faces[f, 1]] * w1 + uvs_tex[faces[f, 2]] * w2
        iu, iv = int(np.rint(uv[0])), int(np.rint(uv[1]))
        if iu < 0 or iv < 0 or iu >= T or iv >= T:
            continue
        flat = iv * (T - 1) + iu
        du, dv = flat % (T - 1), flat // (T - 1)
        if dv >= T or (dv, du) in taken:
            continue
        taken.add((dv, du))
        tex[dv, du] += rgb.reshape(-1, 3)[pix]
        wtex[dv, du] += 1
    return tri.reshape(H, W), np.where(np.isfinite(depth), depth, 0).reshape(H, W)
