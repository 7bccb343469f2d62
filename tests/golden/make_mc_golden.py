"""Golden vectors of the mesh extractor the reference calls: skimage.measure.marching_cubes(volume, level) (nerf_runner.py:1388-1394;
scikit-image is unpinned in the reference's docker file; its default since 0.17 is method='lewiner').

scikit-image is not importable by this repository's interpreter, but the build container carries an Anaconda Python with
scikit-image 0.18.3 -- run this script with THAT interpreter:

    /opt/conda/bin/python3.9 tests/golden/make_mc_golden.py        ->  tests/golden/mc_skimage_vectors.npz

Contents: (1) every non-trivial corner-sign configuration of a single 2x2x2 cell with K sets of random magnitudes (the ambiguous
configurations take different tilings depending on the magnitudes: face tests, interior tests), (2) a few small volumes (sphere, two
touching blobs, smooth and rough random fields, a noisy SDF).  For each: the volume and skimage's vertices / faces as returned
(gradient_direction='descent', allow_degenerate=True, the defaults the reference leaves in place).  Also the classic variant
(method='lorensen') of the volumes, for reference."""
import os
import warnings

import numpy as np

warnings.filterwarnings('ignore')
import skimage                                                        # noqa: E402
from scipy.ndimage import gaussian_filter                             # noqa: E402
from skimage import measure                                           # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
K = 24
rng = np.random.default_rng(20250924)
cells, cverts, cnv, cfaces, cnf = [], [], [], [], []
for case in range(1, 255):
    sign = np.array([1.0 if (case >> p) & 1 else -1.0 for p in range(8)])
    for k in range(K):
        mag = np.exp(rng.uniform(np.log(0.03), np.log(1.0), size=8))
        vol = (sign * mag).astype(np.float32).reshape(2, 2, 2)        # [i, j, k]: corner p = i * 4 + j * 2 + k
        v, f, _, _ = measure.marching_cubes(vol, 0.0)
        pv = np.full((16, 3), np.nan, np.float32)
        pf = np.full((16, 3), -1, np.int32)
        pv[:len(v)] = v
        pf[:len(f)] = f
        cells.append(vol.reshape(8)); cverts.append(pv); cnv.append(len(v)); cfaces.append(pf); cnf.append(len(f))
out = dict(cell_values=np.array(cells), cell_verts=np.array(cverts), cell_nverts=np.array(cnv, np.int32), cell_faces=np.array(cfaces),
           cell_nfaces=np.array(cnf, np.int32), skimage_version=np.array(skimage.__version__))

# (1b) many more magnitude sets for the configurations whose tiling depends on the magnitudes (Lewiner's cases 3, 4, 6, 7, 10, 12, 13: which tiling a cell gets
# depends on its face tests and interior test), stored compactly: a triangle = three cube-edge ids (0..11; 12 = the centre vertex of
# the 'c' tilings), read off the returned vertex positions -- a vertex on a cube edge has two integral coordinates.
CORNER = np.array([(0, 0, 0), (0, 0, 1), (0, 1, 1), (0, 1, 0), (1, 0, 0), (1, 0, 1), (1, 1, 1), (1, 1, 0)])
EDGE = [(0, 1), (1, 2), (2, 3), (3, 0), (4, 5), (5, 6), (6, 7), (7, 4), (0, 4), (1, 5), (2, 6), (3, 7)]


def edge_of(p):
    integral = [abs(x - round(x)) < 1e-9 for x in p]
    if sum(integral) < 2:
        return 12
    for e, (a, b) in enumerate(EDGE):
        ax = int(np.nonzero(CORNER[a] != CORNER[b])[0][0])
        if all(integral[d] and round(p[d]) == CORNER[a][d] for d in range(3) if d != ax):
            return e
    raise ValueError(p)


def n_components(case):
    """the tiling depends on the magnitudes <=> some face of the cube has its two positive corners on a diagonal (Lewiner's cases
    3, 6, 7, 10, 12, 13), or the two minority corners are opposite ends of a body diagonal (case 4: interior test only)"""
    pos = [(case >> q) & 1 for q in range(8)]
    faces = [(0, 1, 2, 3), (4, 5, 6, 7), (0, 1, 5, 4), (1, 2, 6, 5), (2, 3, 7, 6), (3, 0, 4, 7)]
    if any(pos[f[0]] == pos[f[2]] and pos[f[1]] == pos[f[3]] and pos[f[0]] != pos[f[1]] for f in faces):
        return True
    minority = [q for q in range(8) if pos[q] == (1 if sum(pos) <= 4 else 0)]
    return len(minority) == 2 and int(np.abs(CORNER[minority[0]] - CORNER[minority[1]]).sum()) == 3


xv, xt, xn = [], [], []
for case in range(1, 255):
    # (corner p of Lewiner's numbering sits at array offset CORNER[p])
    if not n_components(case):
        continue
    pos = [(case >> q) & 1 for q in range(8)]
    k_sets = 4000 if sum(pos) == 4 and all(pos[q] != pos[r] for q, r in EDGE) else 300          # the two 'case 13' configurations
    for k in range(k_sets):
        mag = np.exp(rng.uniform(np.log(0.02), np.log(1.0), size=8))
        vol = np.zeros((2, 2, 2), np.float32)
        for q in range(8):
            vol[tuple(CORNER[q])] = (1.0 if pos[q] else -1.0) * mag[q]
        v, f, _, _ = measure.marching_cubes(vol, 0.0)
        eid = np.array([edge_of(p) for p in v.astype(np.float64)], np.int8)
        tri = np.full((12, 3), -1, np.int8)
        tri[:len(f)] = eid[f]
        xv.append(vol.reshape(8)); xt.append(tri); xn.append(len(f))
# (1c) the same configurations with TINY magnitudes (3e-4 ... 3e-3): the face tests' determinants A C - B D are then 1e-8 ... 1e-5,
# on both sides of FLT_EPSILON -- where the paper's companion code has a special branch (`return face >= 0`) and scikit-image has none.
n_tiny = 0
for case in range(1, 255):
    if not n_components(case):
        continue
    pos = [(case >> q) & 1 for q in range(8)]
    for k in range(40):
        mag = np.exp(rng.uniform(np.log(3e-4), np.log(3e-3), size=8))
        vol = np.zeros((2, 2, 2), np.float32)
        for q in range(8):
            vol[tuple(CORNER[q])] = (1.0 if pos[q] else -1.0) * mag[q]
        v, f, _, _ = measure.marching_cubes(vol, 0.0)
        eid = np.array([edge_of(p) for p in v.astype(np.float64)], np.int8)
        tri = np.full((12, 3), -1, np.int8)
        tri[:len(f)] = eid[f]
        xv.append(vol.reshape(8)); xt.append(tri); xn.append(len(f))
        n_tiny += 1
out.update(amb_values=np.array(xv), amb_tris=np.array(xt), amb_ntris=np.array(xn, np.int8))
print('ambiguous-configuration cells', len(xv), 'of which with tiny magnitudes', n_tiny)
# (1d) every configuration with magnitudes CLOSE TO THE ISO VALUE (1e-6 ... 1e-3), vertex POSITIONS kept: scikit-image places an edge
# vertex with weights 1 / (eps + |value|) where its eps is np.spacing(1.0) = 2.2e-16 (its Cython source calls it FLT_EPSILON), i.e. an
# exact linear interpolation; with binary32's 1.19e-7 in its place the vertex moves by 1e-4 ... 1e-2 of a voxel at these magnitudes
# (ADVICE r5).  Its own generator: the arrays above keep their bytes.
rng_d = np.random.default_rng(20260930)
nv_, nvv, nnv, nf_, nnf = [], [], [], [], []
for case in range(1, 255):
    sign = np.array([1.0 if (case >> p) & 1 else -1.0 for p in range(8)])
    for k in range(6):
        mag = np.exp(rng_d.uniform(np.log(1e-6), np.log(1e-3), size=8))
        vol = (sign * mag).astype(np.float32).reshape(2, 2, 2)
        v, f, _, _ = measure.marching_cubes(vol, 0.0)
        pv = np.full((16, 3), np.nan, np.float32)
        pf = np.full((16, 3), -1, np.int32)
        pv[:len(v)] = v
        pf[:len(f)] = f
        nv_.append(vol.reshape(8)); nvv.append(pv); nnv.append(len(v)); nf_.append(pf); nnf.append(len(f))
out.update(near_values=np.array(nv_), near_verts=np.array(nvv), near_nverts=np.array(nnv, np.int32), near_faces=np.array(nf_),
           near_nfaces=np.array(nnf, np.int32))
print('near-iso cells', len(nv_))
n = 20
x, y, z = np.mgrid[-1:1:n * 1j, -1:1:n * 1j, -1:1:n * 1j]
vols = {
    'sphere': np.sqrt(x * x + y * y + z * z) - 0.6,
    'blobs': np.minimum(np.sqrt((x + 0.33) ** 2 + y * y + z * z), np.sqrt((x - 0.33) ** 2 + (y - 0.05) ** 2 + z * z)) - 0.34,
    'smooth_noise': gaussian_filter(rng.normal(size=(16, 16, 16)), 1.2),
    'rough_noise': gaussian_filter(rng.normal(size=(12, 12, 12)), 0.6),
    'sdf_noisy': np.sqrt(x * x + y * y + z * z) - 0.6 + gaussian_filter(rng.normal(size=(n, n, n)), 1.0) * 0.15,
    'slab': (np.abs(z) - 0.11 + 0.05 * np.sin(5 * x) * np.cos(4 * y))[:, :, 4:16],
}
for name, vol in vols.items():
    vol = vol.astype(np.float32)
    out[f'vol_{name}'] = vol
    for method in ('lewiner', 'lorensen'):
        v, f, _, _ = measure.marching_cubes(vol, 0.0) if method == 'lewiner' else measure.marching_cubes(vol, 0.0, method=method)
        out[f'{method}_{name}_v'], out[f'{method}_{name}_f'] = v.astype(np.float32), f.astype(np.int32)
np.savez_compressed(os.path.join(HERE, 'mc_skimage_vectors.npz'), **out)
print('cells', len(cells), 'max verts', max(cnv), 'max faces', max(cnf), {k: v.shape for k, v in out.items() if k.endswith('_f')})
