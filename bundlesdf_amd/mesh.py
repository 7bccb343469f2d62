"""Iso-surface extraction from the dense SDF grid and a minimal mesh container.

The reference extracts with skimage.measure.marching_cubes on the host (nerf_runner.py:1388-1394) and wraps the result
in trimesh.Trimesh (:1404); neither package exists in this image.  The SDF grid itself is produced on the GPU
(NeuralObjectField.query_sdf_grid) and the surface is extracted there as well (mesh_gpu.py): marching cubes by default -- this
file derives its case table, `mc_case_table` -- or marching tetrahedra (6 tetrahedra per cell around the main diagonal; the
NumPy version below is what the GPU kernels are compared with).
"""
import numpy as np

# ---- marching cubes: the case table ---------------------------------------------------------------------------------------------
# corner c = x + 2 y + 4 z of the cell; edge e joins corners MC_EDGES[e] = (a, b), a < b, in lexicographic order:
# (0,1) (0,2) (0,4) (1,3) (1,5) (2,3) (2,6) (3,7) (4,5) (4,6) (5,7) (6,7)
MC_EDGES = tuple((a, a | (1 << d)) for a in range(8) for d in range(3) if not a & (1 << d))


def mc_case_table():
    """[256, 16] int8: row `case` (bit c = corner c is inside, value < iso) = [number of triangles T <= 5, 3 T cube-edge ids].

    Derived, not typed in.  On every cube face the iso-contour is drawn as DIRECTED segments between the crossed face edges, with
    the inside corners on the segment's right as seen from outside the cube: two crossings give one segment; four (the ambiguous
    face) give two segments that cut off the inside corners -- a rule that only looks at that face's corner signs, so the two
    cells sharing the face draw the same contour and the surface is watertight.  Every crossed cube edge then has exactly one
    segment arriving and one leaving: the segments are a permutation of the crossed edges, its cycles are the polygons, already
    oriented with their normals from the inside to the outside.  A polygon starts at its smallest edge id (polygons in the
    order of those ids) and is cut into a triangle fan from there."""
    edge_id = {e: i for i, e in enumerate(MC_EDGES)}
    faces = []                                                       # corner cycles, counter-clockwise seen from outside
    for axis in range(3):
        u, v = 1 << ((axis + 1) % 3), 1 << ((axis + 2) % 3)          # (axis, u, v) right-handed
        for side in (0, 1):
            base = side << axis
            cyc = (base, base | u, base | u | v, base | v)           # counter-clockwise seen from +axis ...
            faces.append(cyc if side else cyc[::-1])                 # ... which is outside only for the far face
    table = np.zeros((256, 16), dtype=np.int8)
    for case in range(256):
        nxt = {}
        for cyc in faces:
            ins = [(case >> c) & 1 for c in cyc]
            fe = [edge_id[tuple(sorted((cyc[k], cyc[(k + 1) % 4])))] for k in range(4)]     # face edge k joins corners k, k+1
            cross = [k for k in range(4) if ins[k] != ins[(k + 1) % 4]]
            if len(cross) == 2:
                k1, k2 = cross
                # the corners k1+1 .. k2 lie to the right of the chord from edge k1 to edge k2
                a, b = (fe[k1], fe[k2]) if ins[k2] else (fe[k2], fe[k1])
                assert a not in nxt
                nxt[a] = b
            elif len(cross) == 4:
                for k in range(4):
                    if ins[k]:
                        assert fe[k - 1] not in nxt
                        nxt[fe[k - 1]] = fe[k]
        tris, left = [], set(nxt)
        while left:
            start = min(left)
            poly, e = [], start
            while True:
                poly.append(e)
                left.discard(e)
                e = nxt[e]
                if e == start:
                    break
            tris += [(poly[0], poly[i], poly[i + 1]) for i in range(1, len(poly) - 1)]
        assert len(tris) <= 5
        table[case, 0] = len(tris)
        table[case, 1:1 + 3 * len(tris)] = np.asarray(tris, dtype=np.int8).reshape(-1)
    return table


_CORNERS = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [1, 1, 0], [0, 0, 1], [1, 0, 1], [0, 1, 1], [1, 1, 1]], dtype=np.int64)
_TETS = np.array([[0, 1, 3, 7], [0, 1, 5, 7], [0, 2, 3, 7], [0, 2, 6, 7], [0, 4, 5, 7], [0, 4, 6, 7]], dtype=np.int64)


# Lewiner's lookup tables in the order include/nof_hip.h lists them (NofMclLuts.off[t] = byte offset of table t in the packed buffer)
LEWINER_TABLE_ORDER = ('CASES', 'TILING1', 'TILING2', 'TILING3_1', 'TILING3_2', 'TILING4_1', 'TILING4_2', 'TILING5', 'TILING6_1_1',
                       'TILING6_1_2', 'TILING6_2', 'TILING7_1', 'TILING7_2', 'TILING7_3', 'TILING7_4_1', 'TILING7_4_2', 'TILING8', 'TILING9',
                       'TILING10_1_1', 'TILING10_1_1_', 'TILING10_1_2', 'TILING10_2', 'TILING10_2_', 'TILING11', 'TILING12_1_1',
                       'TILING12_1_1_', 'TILING12_1_2', 'TILING12_2', 'TILING12_2_', 'TILING13_1', 'TILING13_1_', 'TILING13_2', 'TILING13_2_',
                       'TILING13_3', 'TILING13_3_', 'TILING13_4', 'TILING13_5_1', 'TILING13_5_2', 'TILING14', 'TEST3', 'TEST4', 'TEST6',
                       'TEST7', 'TEST10', 'TEST12', 'TEST13', 'SUBCONFIG13')


def lewiner_lut_pack():
    """(packed int8 [n] numpy, offsets int32 [47]) of the lookup tables of Lewiner et al. 2003 for nof_mcl_* -- the tables of the
    paper's companion code (LookUpTable.h), stored as plain int8 arrays in bundlesdf_amd/lewiner_luts.npz (how they got there:
    tools/make_lewiner_luts.py).  Shapes are checked against what the device code strides by."""
    import os
    L = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'lewiner_luts.npz'))
    shapes = {'CASES': (256, 2), 'TILING1': (16, 3), 'TILING2': (24, 6), 'TILING3_1': (24, 6), 'TILING3_2': (24, 12), 'TILING4_1': (8, 6),
              'TILING4_2': (8, 18), 'TILING5': (48, 9), 'TILING6_1_1': (48, 9), 'TILING6_1_2': (48, 27), 'TILING6_2': (48, 15),
              'TILING7_1': (16, 9), 'TILING7_2': (16, 3, 15), 'TILING7_3': (16, 3, 27), 'TILING7_4_1': (16, 15), 'TILING7_4_2': (16, 27),
              'TILING8': (6, 6), 'TILING9': (8, 12), 'TILING10_1_1': (6, 12), 'TILING10_1_1_': (6, 12), 'TILING10_1_2': (6, 24),
              'TILING10_2': (6, 24), 'TILING10_2_': (6, 24), 'TILING11': (12, 12), 'TILING12_1_1': (24, 12), 'TILING12_1_1_': (24, 12),
              'TILING12_1_2': (24, 24), 'TILING12_2': (24, 24), 'TILING12_2_': (24, 24), 'TILING13_1': (2, 12), 'TILING13_1_': (2, 12),
              'TILING13_2': (2, 6, 18), 'TILING13_2_': (2, 6, 18), 'TILING13_3': (2, 12, 30), 'TILING13_3_': (2, 12, 30),
              'TILING13_4': (2, 4, 36), 'TILING13_5_1': (2, 4, 18), 'TILING13_5_2': (2, 4, 30), 'TILING14': (12, 12), 'TEST3': (24,),
              'TEST4': (8,), 'TEST6': (48, 3), 'TEST7': (16, 5), 'TEST10': (6, 3), 'TEST12': (24, 4), 'TEST13': (2, 7), 'SUBCONFIG13': (64,)}
    parts, offs, pos = [], [], 0
    for name in LEWINER_TABLE_ORDER:
        t = np.ascontiguousarray(L[name]).astype(np.int8)
        assert t.shape == shapes[name], (name, t.shape)
        offs.append(pos)
        parts.append(t.reshape(-1))
        pos += t.size
    return np.concatenate(parts), np.array(offs, dtype=np.int32)


def marching_tetrahedra(vol, iso=0.0):
    """vol [nx,ny,nz] float -> (vertices [V,3] in index coordinates, faces [T,3] int64).  Raises ValueError when
    the level set is empty (skimage raises too: the caller maps that to `None`, nerf_runner.py:1390-1394)."""
    vol = np.asarray(vol, dtype=np.float32)
    nx, ny, nz = vol.shape
    inside = vol < iso
    c = inside[:-1, :-1, :-1].astype(np.int8)
    tot = np.zeros_like(c)
    for dx, dy, dz in _CORNERS:
        tot += inside[dx:nx - 1 + dx, dy:ny - 1 + dy, dz:nz - 1 + dz]
    act = np.argwhere((tot > 0) & (tot < 8))
    if len(act) == 0:
        raise ValueError('Surface level must be within volume data range.')
    lin = lambda p: (p[..., 0] * ny + p[..., 1]) * nz + p[..., 2]
    corner_ids = lin(act[:, None, :] + _CORNERS[None])                  # [A,8]
    flat = vol.reshape(-1)
    tv = corner_ids[:, _TETS].reshape(-1, 4)                            # [A*6,4] vertex ids of each tetrahedron
    tf = flat[tv]
    tin = tf < iso
    cnt = tin.sum(1)
    tris_a, tris_b, tris_d = [], [], []                                 # triangles as edge endpoints (a_k, b_k), k = 0..2
    unlin = lambda i: np.stack([i // (ny * nz), (i // nz) % ny, i % nz], -1).astype(np.float64)

    def emit(sel, e, d):
        tris_a.append(np.stack([sel[:, e[0][0]], sel[:, e[1][0]], sel[:, e[2][0]]], 1))
        tris_b.append(np.stack([sel[:, e[0][1]], sel[:, e[1][1]], sel[:, e[2][1]]], 1))
        tris_d.append(d)                                                # inside -> outside direction of the tetrahedron

    for want in (1, 3):                                                 # one vertex on its own side -> one triangle
        m = cnt == want
        if m.any():
            v, s = tv[m], tin[m] if want == 1 else ~tin[m]
            lone = s.argmax(1)
            order = (lone[:, None] + np.arange(4)[None]) % 4
            vs = np.take_along_axis(v, order, 1)                        # vs[:,0] is the lone vertex
            d = (unlin(vs[:, 1]) + unlin(vs[:, 2]) + unlin(vs[:, 3])) / 3.0 - unlin(vs[:, 0])
            emit(vs, [(0, 1), (0, 2), (0, 3)], d if want == 1 else -d)
    m = cnt == 2                                                        # two / two -> a quad (two triangles)
    if m.any():
        v, s = tv[m], tin[m]
        order = np.argsort(~s, axis=1, kind='stable')                   # inside vertices first
        vs = np.take_along_axis(v, order, 1)                            # (i0, i1, o0, o1)
        d = (unlin(vs[:, 2]) + unlin(vs[:, 3])) / 2.0 - (unlin(vs[:, 0]) + unlin(vs[:, 1])) / 2.0
        emit(vs, [(0, 2), (0, 3), (1, 3)], d)
        emit(vs, [(0, 2), (1, 3), (1, 2)], d)
    A = np.concatenate(tris_a, 0)
    Bv = np.concatenate(tris_b, 0)
    lo, hi = np.minimum(A, Bv), np.maximum(A, Bv)
    npts = nx * ny * nz
    keys = lo * npts + hi
    uniq, inv = np.unique(keys.reshape(-1), return_inverse=True)
    faces = inv.reshape(-1, 3)
    ua, ub = uniq // npts, uniq % npts
    fa, fb = flat[ua].astype(np.float64), flat[ub].astype(np.float64)
    t = np.where(fb != fa, (iso - fa) / np.where(fb != fa, fb - fa, 1.0), 0.5)
    verts = unlin(ua) + t[:, None] * (unlin(ub) - unlin(ua))
    # orient every triangle so that its normal points from the inside (value < iso) vertices of its tetrahedron to the
    # outside ones, i.e. out of the object for an SDF (a purely local rule: the GPU extractor applies the same one)
    D = np.concatenate(tris_d, 0)
    nrm = np.cross(verts[faces[:, 1]] - verts[faces[:, 0]], verts[faces[:, 2]] - verts[faces[:, 0]])
    flip = (nrm * D).sum(1) < 0
    faces[flip] = faces[flip][:, [0, 2, 1]]
    faces = faces[(faces[:, 0] != faces[:, 1]) & (faces[:, 1] != faces[:, 2]) & (faces[:, 0] != faces[:, 2])]
    return verts, faces.astype(np.int64)


class Mesh:
    """The subset of trimesh.Trimesh that bundlesdf.py / Utils.py touch on the object extract_mesh returns
    (Utils.py:512-513, bundlesdf.py:748-766): .vertices (assignable), .faces, apply_transform, merge_vertices,
    remove_duplicate_faces, export('*.obj'|'*.ply'), copy."""

    def __init__(self, vertices, faces, process=False, uv=None, texture=None):
        self.vertices = np.asarray(vertices, dtype=np.float64)
        self.faces = np.asarray(faces, dtype=np.int64)
        self.uv = None if uv is None else np.asarray(uv, dtype=np.float64)      # per-vertex texture coordinates in [0,1]
        self.texture = texture                                                   # [H,W,3] uint8, row 0 = top (image order)
        self.vertex_colors = None                                                # [V,3] uint8 (mesh_vertex_color_from_network)

    def apply_transform(self, T):
        T = np.asarray(T, dtype=np.float64)
        self.vertices = self.vertices @ T[:3, :3].T + T[:3, 3]
        return self

    def merge_vertices(self):
        v, inv = np.unique(np.round(self.vertices, 10), axis=0, return_inverse=True)
        self.vertices, self.faces = v, inv.reshape(-1)[self.faces]

    def remove_duplicate_faces(self):
        _, idx = np.unique(np.sort(self.faces, 1), axis=0, return_index=True)
        self.faces = self.faces[np.sort(idx)]

    def copy(self):
        return Mesh(self.vertices.copy(), self.faces.copy(), uv=None if self.uv is None else self.uv.copy(), texture=self.texture)

    def unwrap(self, tex_res=1024):
        """A UV parameterisation (what trimesh's mesh.unwrap() = xatlas gives the reference, nerf_runner.py:1484): here a
        per-triangle atlas -- every triangle gets its own half of a square cell of the texture, vertices are duplicated per
        face (3F vertices), cells are inset by one texel so that neighbouring triangles never share texels."""
        F = len(self.faces)
        n = int(np.ceil(np.sqrt((F + 1) // 2))) or 1
        cell = (tex_res - 1) / n                                   # texels per cell side
        if cell < 4:
            raise ValueError(f'{F} triangles do not fit a {tex_res}^2 texture (cell {cell:.2f} texels): raise tex_res '
                             f'(Mesh.atlas_resolution gives the size that fits) or simplify')
        k = np.arange(F)
        c, upper = k // 2, (k % 2).astype(bool)
        cx, cy = (c % n) * cell, (c // n) * cell
        m = 1.0                                                    # inset (texels)
        lo = np.stack([np.stack([cx + m, cy + m], -1), np.stack([cx + cell - 2.5 * m, cy + m], -1),
                       np.stack([cx + m, cy + cell - 2.5 * m], -1)], 1)
        up = np.stack([np.stack([cx + cell - m, cy + cell - m], -1), np.stack([cx + 2.5 * m, cy + cell - m], -1),
                       np.stack([cx + cell - m, cy + 2.5 * m], -1)], 1)
        uv_tex = np.where(upper[:, None, None], up, lo)            # [F,3,2] texel coordinates
        verts = self.vertices[self.faces].reshape(-1, 3)
        faces = np.arange(3 * F, dtype=np.int64).reshape(F, 3)
        return Mesh(verts, faces, uv=uv_tex.reshape(-1, 2) / (tex_res - 1))

    def atlas_resolution(self, wanted=1024, texels_per_cell=8, limit=16384):
        """The texture size `unwrap` needs for this face count: `wanted` when every triangle's cell gets at least
        `texels_per_cell` texels there, else the next multiple of 256 that does (capped).  The reference's xatlas unwrap has no
        such limit; a 2 mm mesh of a hand-sized object has ~200 k triangles, which a 1024^2 per-triangle atlas cannot hold."""
        n = int(np.ceil(np.sqrt((len(self.faces) + 1) // 2))) or 1
        need = n * texels_per_cell + 1
        return int(min(max(wanted, (need + 255) // 256 * 256), limit))

    @property
    def face_normals(self):
        v = self.vertices
        n = np.cross(v[self.faces[:, 1]] - v[self.faces[:, 0]], v[self.faces[:, 2]] - v[self.faces[:, 0]])
        return n / (np.linalg.norm(n, axis=1, keepdims=True) + 1e-30)

    def sample(self, n, seed=0):
        """Area-weighted surface samples (for Chamfer evaluation)."""
        rng = np.random.default_rng(seed)
        v = self.vertices
        a, b, c = v[self.faces[:, 0]], v[self.faces[:, 1]], v[self.faces[:, 2]]
        area = 0.5 * np.linalg.norm(np.cross(b - a, c - a), axis=1)
        f = rng.choice(len(area), size=n, p=area / area.sum())
        r1, r2 = np.sqrt(rng.random(n)), rng.random(n)
        return (1 - r1)[:, None] * a[f] + (r1 * (1 - r2))[:, None] * b[f] + (r1 * r2)[:, None] * c[f]

    def export(self, path):
        path = str(path)
        if path.endswith('.ply'):
            col = self.vertex_colors is not None and len(self.vertex_colors) == len(self.vertices)
            with open(path, 'w') as f:
                f.write(f'ply\nformat ascii 1.0\nelement vertex {len(self.vertices)}\nproperty float x\nproperty float y\n'
                        f'property float z\n' + ('property uchar red\nproperty uchar green\nproperty uchar blue\n' if col else '') +
                        f'element face {len(self.faces)}\nproperty list uchar int vertex_indices\nend_header\n')
                if col:
                    for v, c in zip(self.vertices, np.asarray(self.vertex_colors, dtype=np.uint8)):
                        f.write('%.7f %.7f %.7f %d %d %d\n' % (v[0], v[1], v[2], c[0], c[1], c[2]))
                else:
                    np.savetxt(f, self.vertices, fmt='%.7f')
                np.savetxt(f, np.concatenate([np.full((len(self.faces), 1), 3), self.faces], 1), fmt='%d')
        elif self.uv is not None:                                 # textured OBJ: .obj + .mtl + .png, like trimesh's exporter
            base = path[:-4]
            name = base.split('/')[-1]
            with open(path, 'w') as f:
                f.write(f'mtllib {name}.mtl\nusemtl material_0\n')
                np.savetxt(f, self.vertices, fmt='v %.7f %.7f %.7f')
                np.savetxt(f, self.uv, fmt='vt %.7f %.7f')
                fi = self.faces + 1
                np.savetxt(f, np.stack([fi[:, 0], fi[:, 0], fi[:, 1], fi[:, 1], fi[:, 2], fi[:, 2]], 1), fmt='f %d/%d %d/%d %d/%d')
            with open(base + '.mtl', 'w') as f:
                f.write(f'newmtl material_0\nKa 1 1 1\nKd 1 1 1\nKs 0 0 0\nmap_Kd {name}.png\n')
            if self.texture is not None:
                from PIL import Image
                Image.fromarray(np.asarray(self.texture, np.uint8)).save(base + '.png')
        else:
            with open(path, 'w') as f:
                np.savetxt(f, self.vertices, fmt='v %.7f %.7f %.7f')
                np.savetxt(f, self.faces + 1, fmt='f %d %d %d')
        return path


def largest_component(mesh):
    """Keep the connected component with the most faces (what run_global_nerf does with trimesh_split before export,
    bundlesdf.py:748-760): unsupervised space inside the object can carry small closed level sets."""
    from scipy.sparse import coo_matrix
    from scipy.sparse.csgraph import connected_components
    v, f = np.asarray(mesh.vertices), np.asarray(mesh.faces)
    n = len(v)
    e = np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]], 0)
    g = coo_matrix((np.ones(len(e), dtype=np.int8), (e[:, 0], e[:, 1])), shape=(n, n))
    _, lab = connected_components(g, directed=False)
    fl = lab[f[:, 0]]
    keep = fl == np.bincount(fl).argmax()
    used = np.unique(f[keep])
    remap = -np.ones(n, dtype=np.int64)
    remap[used] = np.arange(len(used))
    return make_mesh(v[used], remap[f[keep]])


def make_mesh(vertices, faces):
    """trimesh.Trimesh(vertices, faces, process=False) when trimesh is importable (the reference's return type,
    nerf_runner.py:1404), else the minimal container above."""
    try:
        import trimesh
        return trimesh.Trimesh(vertices, faces, process=False)
    except ImportError:
        return Mesh(vertices, faces)
