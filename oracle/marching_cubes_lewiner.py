"""Marching cubes with Lewiner's topological disambiguation, as scikit-image implements it -- TEST INFRASTRUCTURE (oracle/__init__.py).

The reference extracts its mesh with `skimage.measure.marching_cubes(sigma, isolevel)` (nerf_runner.py:1388-1394), whose default
method is 'lewiner': T. Lewiner, H. Lopes, A. Vieira, G. Tavares, "Efficient implementation of Marching Cubes' cases with
topological guarantees", JGT 8(2), 2003.  scikit-image is a third-party dependency that is absent from /root/reference (its
docker file installs it unpinned, docker/dockerfile:99).  This file restates the published algorithm: the 33 topological cases
behind the 256 corner-sign configurations, chosen per cell by the paper's face tests and interior test, with the paper's own lookup
tables (`oracle/lewiner_luts.npz` = LookUpTable.h of the paper's companion code, read out of scikit-image 0.18.3's copy by
tools/make_lewiner_luts.py), plus the two places where scikit-image departs from the companion code -- where it puts an edge
vertex and the extra centre vertex of the 'c' tilings (inverse-|value| weights) -- because scikit-image is what the reference calls.

PINNED on scikit-image itself: tests/golden/mc_skimage_vectors.npz holds skimage 0.18.3's output for every corner-sign configuration
of a single cell with 24 sets of random magnitudes and for six small volumes (tests/golden/make_mc_golden.py, run with the build
container's Anaconda interpreter); tests/test_mesh.py requires this file's triangles to be skimage's, one for one.

Cube conventions (Lewiner's): corner p of a cell at array index (i, j, k) -- x = the LAST array axis, as in scikit-image --
    0:(i,j,k) 1:(i,j,k+1) 2:(i,j+1,k+1) 3:(i,j+1,k) 4:(i+1,j,k) 5:(i+1,j,k+1) 6:(i+1,j+1,k+1) 7:(i+1,j+1,k)
edges 0:(0,1) 1:(1,2) 2:(2,3) 3:(3,0) 4:(4,5) 5:(5,6) 6:(6,7) 7:(7,4) 8:(0,4) 9:(1,5) 10:(2,6) 11:(3,7); "edge" 12 = the centre vertex.
"""
import os

import numpy as np

_L = {k: v.astype(np.int64) for k, v in np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'lewiner_luts.npz')).items()}
# scikit-image 0.18.3's Cython source names this FLT_EPSILON but defines it as np.spacing(1.0) = 2.2e-16 (double precision): its
# edge vertices are exact linear interpolations down to |value| ~ 1e-15 (pinned by the near-iso cells of the fixture, whose
# POSITIONS are compared at 2e-7 voxel; with binary32's 1.19e-7 here a vertex between -1e-4 and 3e-4 sat at 0.25015 instead of 0.25)
FLT_EPSILON = float(np.spacing(1.0))
# corner p -> (di, dj, dk)
CORNER = np.array([(0, 0, 0), (0, 0, 1), (0, 1, 1), (0, 1, 0), (1, 0, 0), (1, 0, 1), (1, 1, 1), (1, 1, 0)], dtype=np.int64)
EDGE = np.array([(0, 1), (1, 2), (2, 3), (3, 0), (4, 5), (5, 6), (6, 7), (7, 4), (0, 4), (1, 5), (2, 6), (3, 7)], dtype=np.int64)


def _test_face(c, face):
    """Lewiner's test_face: the sign of A C - B D on the ambiguous face decides whether its positive corners are joined."""
    f = abs(face)
    A, B, C, D = {1: (c[0], c[4], c[5], c[1]), 2: (c[1], c[5], c[6], c[2]), 3: (c[2], c[6], c[7], c[3]), 4: (c[3], c[7], c[4], c[0]),
                  5: (c[0], c[3], c[2], c[1]), 6: (c[4], c[7], c[6], c[5])}[f]
    # (the companion code answers `face >= 0` when |A C - B D| < FLT_EPSILON; scikit-image 0.18.3 has no such branch -- cells with
    # determinants of 1e-9 ... 1e-7 in tests/golden/mc_skimage_vectors.npz follow the sign rule)
    return face * A * (A * C - B * D) >= 0


def _test_interior(c, s, case, config, subconfig):
    """Lewiner's test_interior: are the two positive (or negative) diagonal components joined through the cell's interior?"""
    if case in (4, 10):
        a = (c[4] - c[0]) * (c[6] - c[2]) - (c[7] - c[3]) * (c[5] - c[1])
        b = c[2] * (c[4] - c[0]) + c[0] * (c[6] - c[2]) - c[1] * (c[7] - c[3]) - c[3] * (c[5] - c[1])
        t = -b / (2 * a) if a != 0 else np.inf
        if not (0 <= t <= 1):
            return s > 0
        At = c[0] + (c[4] - c[0]) * t
        Bt = c[3] + (c[7] - c[3]) * t
        Ct = c[2] + (c[6] - c[2]) * t
        Dt = c[1] + (c[5] - c[1]) * t
    else:
        edge = {6: lambda: _L['TEST6'][config][2], 7: lambda: _L['TEST7'][config][4], 12: lambda: _L['TEST12'][config][3],
                13: lambda: _L['TILING13_5_1'][config][subconfig][0]}[case]()
        # the reference edge (p, q) and the three edges parallel to it, in the cyclic order Lewiner's code walks them
        par = {0: (0, 1, 3, 2, 7, 6, 4, 5), 1: (1, 2, 0, 3, 4, 7, 5, 6), 2: (2, 3, 1, 0, 5, 4, 6, 7), 3: (3, 0, 2, 1, 6, 5, 7, 4),
               4: (4, 5, 0, 1, 3, 2, 7, 6), 5: (5, 6, 1, 2, 0, 3, 4, 7), 6: (6, 7, 2, 3, 1, 0, 5, 4), 7: (7, 4, 3, 0, 2, 1, 6, 5),
               8: (0, 4, 3, 7, 2, 6, 1, 5), 9: (1, 5, 0, 4, 3, 7, 2, 6), 10: (2, 6, 1, 5, 0, 4, 3, 7), 11: (3, 7, 2, 6, 1, 5, 0, 4)}[int(edge)]
        p, q, b0, b1, c0, c1, d0, d1 = par
        t = c[p] / (c[p] - c[q])
        At = 0.0
        Bt = c[b0] + (c[b1] - c[b0]) * t
        Ct = c[c0] + (c[c1] - c[c0]) * t
        Dt = c[d0] + (c[d1] - c[d0]) * t
    test = (1 if At >= 0 else 0) + (2 if Bt >= 0 else 0) + (4 if Ct >= 0 else 0) + (8 if Dt >= 0 else 0)
    if test in (0, 1, 2, 3, 4, 6, 8, 9, 12):
        return s > 0
    if test in (7, 11, 13, 14, 15):
        return s < 0
    # (the companion code compares the determinant with FLT_EPSILON and falls through to `return s < 0` when that test fails;
    # scikit-image 0.18.3 compares with ZERO and answers False there whatever the sign of s -- both established on its outputs for
    # cells of ordinary and of tiny magnitudes, tests/golden/mc_skimage_vectors.npz -- and scikit-image is what the reference calls)
    if test == 5:
        return At * Ct - Bt * Dt < 0 and s > 0
    return At * Ct - Bt * Dt >= 0 and s > 0                             # test == 10


def cell_tiling(c):
    """c [8]: corner values minus the iso value, Lewiner's corner order.  Returns (tiling row of edge ids, number of triangles)."""
    idx = 0
    for p in range(8):
        if c[p] > 0:
            idx |= 1 << p
    case, config = int(_L['CASES'][idx][0]), int(_L['CASES'][idx][1])
    T = _L
    tf = lambda f: _test_face(c, int(f))
    ti = lambda s, sub=0: _test_interior(c, int(s), case, config, sub)
    if case == 0:
        return np.zeros(0, np.int64), 0
    if case == 1:
        return T['TILING1'][config], 1
    if case == 2:
        return T['TILING2'][config], 2
    if case == 3:
        return (T['TILING3_2'][config], 4) if tf(T['TEST3'][config]) else (T['TILING3_1'][config], 2)
    if case == 4:
        return (T['TILING4_1'][config], 2) if ti(T['TEST4'][config]) else (T['TILING4_2'][config], 6)
    if case == 5:
        return T['TILING5'][config], 3
    if case == 6:
        if tf(T['TEST6'][config][0]):
            return T['TILING6_2'][config], 5
        if ti(T['TEST6'][config][1]):
            return T['TILING6_1_1'][config], 3
        return T['TILING6_1_2'][config], 9
    if case == 7:
        sub = (1 if tf(T['TEST7'][config][0]) else 0) + (2 if tf(T['TEST7'][config][1]) else 0) + (4 if tf(T['TEST7'][config][2]) else 0)
        if sub == 0:
            return T['TILING7_1'][config], 3
        if sub in (1, 2, 4):
            return T['TILING7_2'][config][{1: 0, 2: 1, 4: 2}[sub]], 5
        if sub in (3, 5, 6):
            return T['TILING7_3'][config][{3: 0, 5: 1, 6: 2}[sub]], 9
        return (T['TILING7_4_2'][config], 9) if ti(T['TEST7'][config][3]) else (T['TILING7_4_1'][config], 5)
    if case == 8:
        return T['TILING8'][config], 2
    if case == 9:
        return T['TILING9'][config], 4
    if case in (10, 12):
        n = str(case)
        if tf(T['TEST' + n][config][0]):
            if tf(T['TEST' + n][config][1]):
                return T['TILING' + n + '_1_1_'][config], 4
            return T['TILING' + n + '_2'][config], 8
        if tf(T['TEST' + n][config][1]):
            return T['TILING' + n + '_2_'][config], 8
        if ti(T['TEST' + n][config][2]):
            return T['TILING' + n + '_1_1'][config], 4
        return T['TILING' + n + '_1_2'][config], 8
    if case == 11:
        return T['TILING11'][config], 4
    if case == 13:
        sub = sum((1 << b) for b in range(6) if tf(T['TEST13'][config][b]))
        k = int(T['SUBCONFIG13'][sub])
        if k == 0:
            return T['TILING13_1'][config], 4
        if 1 <= k <= 6:
            return T['TILING13_2'][config][k - 1], 6
        if 7 <= k <= 18:
            return T['TILING13_3'][config][k - 7], 10
        if 19 <= k <= 22:
            return T['TILING13_4'][config][k - 19], 12
        if 23 <= k <= 26:
            s = k - 23
            return (T['TILING13_5_1'][config][s], 6) if ti(T['TEST13'][config][6], s) else (T['TILING13_5_2'][config][s], 10)
        if 27 <= k <= 38:
            return T['TILING13_3_'][config][k - 27], 10
        if 39 <= k <= 44:
            return T['TILING13_2_'][config][k - 39], 6
        if k == 45:
            return T['TILING13_1_'][config], 4
        raise ValueError(f'impossible case 13 subconfiguration {sub}')
    if case == 14:
        return T['TILING14'][config], 4
    raise ValueError(case)


def _edge_point(c, e):
    """scikit-image's vertex on cube edge e: the two end points weighted by 1 / (eps + |value|) (= linear interpolation up to eps)"""
    p, q = EDGE[e]
    w1, w2 = 1.0 / (FLT_EPSILON + abs(c[p])), 1.0 / (FLT_EPSILON + abs(c[q]))
    if tuple(CORNER[p]) > tuple(CORNER[q]):                            # (from the end point with the smaller grid index: an edge shared by
        p, q, w1, w2 = q, p, w2, w1                                    #  four cells gets the same float64 position from each of them)
    return CORNER[p] + (CORNER[q] - CORNER[p]) * (w2 / (w1 + w2))


def _centre_point(c):
    """scikit-image's centre vertex: the eight corners weighted by 1 / (eps + |value|)"""
    w = 1.0 / (FLT_EPSILON + np.abs(np.asarray(c, np.float64)))
    return (CORNER * w[:, None]).sum(0) / w.sum()


def marching_cubes(vol, iso=0.0):
    """vertices [V,3] float64 in array-index coordinates (i, j, k), faces [F,3] int64; oriented like scikit-image's default
    (gradient_direction='descent').  Vertices are welded per grid edge (centre vertices per cell); faces in cell order."""
    vol = np.asarray(vol, np.float32)
    ni, nj, nk = vol.shape
    keys, verts, faces = {}, [], []
    for i in range(ni - 1):
        for j in range(nj - 1):
            blk = vol[i:i + 2, j:j + 2]
            lo, hi = blk[:, :, :-1], blk[:, :, 1:]
            mn = np.minimum(lo, hi).min((0, 1)) if False else None
            for k in range(nk - 1):
                c = np.array([float(vol[i + d[0], j + d[1], k + d[2]]) - iso for d in CORNER], np.float64)
                if (c > 0).all() or not (c > 0).any():
                    continue
                row, nt = cell_tiling(c)
                for t in range(nt):
                    tri = []
                    for e in row[3 * t:3 * t + 3]:
                        e = int(e)
                        if e == 12:
                            key = ('c', i, j, k)
                            if key not in keys:
                                keys[key] = len(verts)
                                verts.append(np.array([i, j, k], np.float64) + _centre_point(c))
                        else:
                            p, q = EDGE[e]
                            a = (i + CORNER[p][0], j + CORNER[p][1], k + CORNER[p][2])
                            b = (i + CORNER[q][0], j + CORNER[q][1], k + CORNER[q][2])
                            key = (min(a, b), max(a, b))
                            if key not in keys:
                                keys[key] = len(verts)
                                verts.append(np.array([i, j, k], np.float64) + _edge_point(c, e))
                        tri.append(keys[key])
                    faces.append(tri[::-1])                              # ('descent': scikit-image flips the companion code's winding)
    return np.array(verts, np.float64).reshape(-1, 3), np.array(faces, np.int64).reshape(-1, 3)
