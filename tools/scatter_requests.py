"""How many atomic line requests does k_hash_bwd_agg issue for a batch of sample points?  A NumPy restatement of the kernel's
run / chain / emission rules (nof_hash.hip), used by bench.py to price the scatter against the measured atomic rate of the
memory side (tools/atomic_probe.py: ~20.8 G line requests/s on MI355X, whatever the scope, flags or data type).

A request = one 64-byte line touched by one atomic instruction (lanes of an instruction that fall into the same line merge).
The kernel emits, per (64 samples, level) tile, the runs of equal cells four at a time; a corner whose grid vertex also belongs
to the next run's cell is handed on and not emitted (chains end at 32-run window edges and at out-of-range samples)."""
import numpy as np

NONE = np.int64(-1)
P1, P2 = np.int64(2654435761), np.int64(805459861)
M32 = np.int64(0xFFFFFFFF)


def _axis(key, d):
    return (key >> (10 * d)) & 1023


def shared_corners(key, other):
    """8-bit masks: corners of cell `key` that are vertices of cell `other` (0 where other is NONE or not touching)."""
    m = np.full(key.shape, 0xFF, dtype=np.int64)
    ok = other != NONE
    for d, (lo, hi) in enumerate(((0x55, 0xAA), (0x33, 0xCC), (0x0F, 0xF0))):
        delta = _axis(other, d) - _axis(key, d)
        ok &= np.abs(delta) <= 1
        m &= np.where(delta == 0, 0xFF, np.where(delta == 1, hi, lo))
    return np.where(ok, m, 0)


def level_rows(vx, vy, vz, res, size, hashed):
    if hashed:
        idx = (vx ^ ((vy * P1) & M32) ^ ((vz * P2) & M32)) & M32
    else:
        r1 = np.int64(res + 1)
        idx = (vx + vy * r1 + vz * r1 * r1) & M32
    return idx % np.int64(size)


def count_requests(pts_w, scale, resolution, offset, size, hashed, levels, window=32, nonzero=None):
    """pts_w [N,3] float32 (N a multiple of 64, consecutive samples of rays); per-level constants as in NofHashGrid.
    nonzero: optional [L, N] bool, False where the level's feature gradient of a sample is exactly zero -- the kernel skips tiles
    without a non-zero gradient and does not emit vertex totals that are exactly zero (taken here as: every run of the vertex's
    chain, two runs back at most, has only zero gradients).  Returns dict(level -> requests) for the given levels."""
    pts = np.asarray(pts_w, np.float32)
    N = len(pts) // 64 * 64
    pts = pts[:N]
    x01 = ((pts + np.float32(1)) * np.float32(0.5)).astype(np.float32)
    oob = ((x01 < 0) | (x01 > 1)).any(-1)
    idx = np.arange(N)
    tile, lane = idx // 64, idx % 64
    out = {}
    for l in levels:
        pos = x01 * np.float32(scale[l]) + np.float32(0.5)
        g = np.floor(pos).astype(np.int64)
        key = np.where(oob, NONE, g[:, 0] | (g[:, 1] << 10) | (g[:, 2] << 20))
        valid = key != NONE
        prev = np.concatenate([[NONE], key[:-1]]); prev[lane == 0] = NONE
        nxt = np.concatenate([key[1:], [NONE]]); nxt[lane == 63] = NONE
        head, tail = valid & (prev != key), valid & (nxt != key)
        rk = key[tail]                                                   # one entry per run, in order
        rt, rnext = tile[tail], nxt[tail]
        first = np.concatenate([[True], rt[1:] != rt[:-1]])              # slot of the run inside its tile
        start = np.maximum.accumulate(np.where(first, np.arange(len(rt)), 0))
        slot = np.arange(len(rt)) - start
        last = np.concatenate([rt[1:] != rt[:-1], [True]])
        hand_on = shared_corners(rk, rnext)
        hand_on[(slot % window == window - 1) | last] = 0
        group = rt * 64 + slot // 4                                      # one atomic instruction = 4 runs of one tile
        dead = None
        if nonzero is not None:
            nz = np.asarray(nonzero[l][:N], bool) & valid
            hpos, tpos = np.flatnonzero(head), np.flatnonzero(tail)
            csum = np.concatenate([[0], np.cumsum(nz)])
            rz = (csum[tpos + 1] - csum[hpos]) == 0                      # run has only zero gradients
            # the run a corner collects from: lane-adjacent previous run(s) of the same tile and window
            adj1 = np.zeros(len(rk), bool); adj1[1:] = (hpos[1:] == tpos[:-1] + 1) & (rt[1:] == rt[:-1]) & (slot[1:] % window != 0)
            prev1 = np.concatenate([[NONE], rk[:-1]]); prev1[~adj1] = NONE
            adj2 = np.zeros(len(rk), bool); adj2[2:] = adj1[2:] & adj1[1:-1] & (slot[2:] % window != 1)
            prev2 = np.concatenate([[NONE, NONE], rk[:-2]]); prev2[~adj2] = NONE
            take1 = shared_corners(rk, prev1)
            take2 = take1 & shared_corners(rk, prev2)
            rz1 = np.concatenate([[True], rz[:-1]]); rz2 = np.concatenate([[True, True], rz[:-2]])
            dead = [rz & ((((take1 >> k) & 1) == 0) | rz1) & ((((take2 >> k) & 1) == 0) | rz2) for k in range(8)]
        keys = []
        for k in range(8):
            emit = ((hand_on >> k) & 1) == 0
            if dead is not None:
                emit &= ~dead[k]
            rows = level_rows(_axis(rk, 0) + (k & 1), _axis(rk, 1) + ((k >> 1) & 1), _axis(rk, 2) + (k >> 2),
                              int(resolution[l]), int(size[l]), bool(hashed[l]))
            line = (np.int64(offset[l]) + rows) >> 3
            keys.append((group[emit] << 24) | line[emit])
        out[l] = int(len(np.unique(np.concatenate(keys))))
    return out
