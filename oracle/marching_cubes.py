"""Classic marching cubes (vertices on grid edges by linear interpolation, one or more polygons per sign configuration) as a
NumPy restatement -- TEST INFRASTRUCTURE (see oracle/__init__.py).

The reference extracts its mesh with skimage.measure.marching_cubes(sigma, isolevel) on the host (nerf_runner.py:1388-1394), whose
default method is Lewiner's variant: THAT is restated in oracle/marching_cubes_lewiner.py and pinned on scikit-image's own outputs.
This file is the CLASSIC algorithm (one fixed tiling per corner-sign configuration; scikit-image's method='lorensen' family), kept as
the yardstick of the product's classic extractor (cfg mesh_extractor: 'cubes').  It is fully defined by its construction, which this
file restates without a hand-typed case table: for every one of the 256 corner-sign configurations the iso-polygons are DERIVED --
on each cube face the crossed edges are joined by segments (two crossed edges: one segment; four, the ambiguous face: two
segments that cut off the inside corners, a rule that depends on that face's corner signs only and is therefore applied
identically by the two cubes sharing the face, which is what makes the surface watertight), the segments are chained into
closed loops, every loop is fan-triangulated.  Against scikit-image 0.18.3 (tests/golden/mc_skimage_vectors.npz): the same
vertices as its classic method on every volume of the fixture, the same polygons with other diagonals (54 % of the triangles
identical); Lewiner's variant differs from both in the ambiguous cells (other tilings, extra centre vertices).

It is the yardstick of the product's marching-cubes extractor (nof_mc_*, bundlesdf_amd/mesh_gpu.py: the same vertices and the
same triangles, tests/test_gpu_mesh.py; the product derives its own case table by a different construction -- directed face
segments, cycles of the resulting permutation -- and tests/test_mesh.py compares the two tables case by case), and it bounds what
the marching-TETRAHEDRA option (nof_mt_*, bundlesdf_amd/mesh.py) may differ from a marching-cubes surface by (symmetric Hausdorff
distance below half a voxel on noisy SDFs).
"""
import numpy as np

_C = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [1, 1, 0], [0, 0, 1], [1, 0, 1], [0, 1, 1], [1, 1, 1]], dtype=np.int64)   # corner = x + 2y + 4z
_EDGES = [(a, b) for a in range(8) for b in range(a + 1, 8) if bin(a ^ b).count('1') == 1]                                  # 12 cube edges
_EDGE_ID = {e: i for i, e in enumerate(_EDGES)}
# faces as corner cycles (consecutive corners share a cube edge)
_FACES = [(0, 1, 3, 2), (4, 5, 7, 6), (0, 1, 5, 4), (2, 3, 7, 6), (0, 2, 6, 4), (1, 3, 7, 5)]


def _eid(a, b):
    return _EDGE_ID[(min(a, b), max(a, b))]


def _build_table():
    """case -> [T,3] cube-edge ids.  Loops are found as UNDIRECTED chains of face segments and then given their direction by
    geometry: a loop's edges each join one inside and one outside corner, the sum of those (outside - inside) vectors is the
    side the surface normal must point to, and the loop's area vector (Newell's formula over the edge midpoints) is made to
    agree with it -- normals point from the inside (value < iso) to the outside.  Every loop starts at its smallest edge id,
    loops are ordered by that id, and are fan-triangulated from it: a canonical, oriented table."""
    mid = np.array([(_C[a] + _C[b]) / 2.0 for a, b in _EDGES])
    table = []
    for case in range(256):
        inside = [(case >> c) & 1 for c in range(8)]
        segs = []
        for f in _FACES:
            s = [inside[c] for c in f]
            crossed = [k for k in range(4) if s[k] != s[(k + 1) % 4]]     # face edge k joins corners k, k+1
            e = lambda k: _eid(f[k], f[(k + 1) % 4])
            if len(crossed) == 2:
                segs.append((e(crossed[0]), e(crossed[1])))
            elif len(crossed) == 4:                                        # ambiguous face: cut off each inside corner
                for k in range(4):
                    if s[k]:
                        segs.append((e((k - 1) % 4), e(k)))
        loops, left = [], list(segs)
        while left:
            a, b = left.pop()
            loop = [a, b]
            while loop[-1] != loop[0]:
                for i, (p, q) in enumerate(left):
                    if p == loop[-1] or q == loop[-1]:
                        loop.append(q if p == loop[-1] else p)
                        left.pop(i)
                        break
                else:
                    raise AssertionError(f'open loop in case {case}')
            loops.append(loop[:-1])
        canon = []
        for lp in loops:
            out_dir = np.zeros(3)
            for eid in lp:
                a, b = _EDGES[eid]
                out_dir += (_C[b] - _C[a]) * (1.0 if inside[a] else -1.0)
            pts = mid[lp]
            area = sum(np.cross(pts[i], pts[(i + 1) % len(lp)]) for i in range(len(lp)))
            assert abs(float(area @ out_dir)) > 1e-9, case
            if float(area @ out_dir) < 0:
                lp = lp[::-1]
            k = lp.index(min(lp))
            canon.append(lp[k:] + lp[:k])
        canon.sort(key=lambda lp: lp[0])
        tris = []
        for lp in canon:
            for i in range(1, len(lp) - 1):
                tris.append((lp[0], lp[i], lp[i + 1]))
        table.append(np.array(tris, dtype=np.int64).reshape(-1, 3))
    return table


_TABLE = _build_table()
_EA = np.array([a for a, b in _EDGES])
_EB = np.array([b for a, b in _EDGES])


def marching_cubes(vol, iso=0.0):
    """vol [nx,ny,nz] -> (vertices [V,3] in index coordinates, faces [T,3]); 'inside' = value < iso.  Triangles are oriented
    with their normals from the inside to the outside (see _build_table); vertices are sorted by their edge key
    lo * (nx ny nz) + hi, faces are listed case by case (compare them as a set of rows)."""
    vol = np.asarray(vol, dtype=np.float64)
    nx, ny, nz = vol.shape
    inside = vol < iso
    case = np.zeros((nx - 1, ny - 1, nz - 1), dtype=np.int64)
    for c, (dx, dy, dz) in enumerate(_C):
        case |= inside[dx:nx - 1 + dx, dy:ny - 1 + dy, dz:nz - 1 + dz].astype(np.int64) << c
    act = np.argwhere((case > 0) & (case < 255))
    if len(act) == 0:
        raise ValueError('Surface level must be within volume data range.')
    cs = case[act[:, 0], act[:, 1], act[:, 2]]
    lin = lambda p: (p[..., 0] * ny + p[..., 1]) * nz + p[..., 2]
    flat = vol.reshape(-1)
    keys_a, keys_b = [], []
    for k in np.unique(cs):
        tri = _TABLE[k]
        if len(tri) == 0:
            continue
        cells = act[cs == k]                                               # [n,3]
        ca = lin(cells[:, None, None, :] + _C[_EA[tri]][None])             # [n,T,3] first corner of each triangle-corner's edge
        cb = lin(cells[:, None, None, :] + _C[_EB[tri]][None])
        keys_a.append(ca.reshape(-1, 3))
        keys_b.append(cb.reshape(-1, 3))
    A, Bv = np.concatenate(keys_a, 0), np.concatenate(keys_b, 0)
    npts = nx * ny * nz
    uniq, inv = np.unique((A * npts + Bv).reshape(-1), return_inverse=True)
    faces = inv.reshape(-1, 3)
    ua, ub = uniq // npts, uniq % npts
    fa, fb = flat[ua], flat[ub]
    t = np.where(fb != fa, (iso - fa) / np.where(fb != fa, fb - fa, 1.0), 0.5)
    unlin = lambda i: np.stack([i // (ny * nz), (i // nz) % ny, i % nz], -1).astype(np.float64)
    verts = unlin(ua) + t[:, None] * (unlin(ub) - unlin(ua))
    faces = faces[(faces[:, 0] != faces[:, 1]) & (faces[:, 1] != faces[:, 2]) & (faces[:, 0] != faces[:, 2])]
    return verts, faces


def surface_samples(verts, faces, per_unit_area=40.0, seed=0):
    """area-weighted surface samples (about per_unit_area points per unit of index-space area) + the vertices themselves"""
    a, b, c = verts[faces[:, 0]], verts[faces[:, 1]], verts[faces[:, 2]]
    area = 0.5 * np.linalg.norm(np.cross(b - a, c - a), axis=1)
    n = max(int(area.sum() * per_unit_area), 1000)
    rng = np.random.default_rng(seed)
    f = rng.choice(len(area), size=n, p=area / area.sum())
    r1, r2 = np.sqrt(rng.random(n)), rng.random(n)
    pts = (1 - r1)[:, None] * a[f] + (r1 * (1 - r2))[:, None] * b[f] + (r1 * r2)[:, None] * c[f]
    return np.concatenate([pts, verts], 0)


def hausdorff(v1, f1, v2, f2):
    """symmetric Hausdorff distance between two triangle surfaces, estimated from dense samples (slight over-estimate by the
    sampling pitch)"""
    from scipy.spatial import cKDTree
    s1, s2 = surface_samples(v1, f1, seed=1), surface_samples(v2, f2, seed=2)
    d12, _ = cKDTree(s2).query(s1)
    d21, _ = cKDTree(s1).query(s2)
    return float(max(d12.max(), d21.max())), float(0.5 * (d12.mean() + d21.mean()))
