"""CPU oracle (PyTorch fp32 + NumPy) of the Neural Object Field hot path.

TEST INFRASTRUCTURE: see oracle/__init__.py for who may import this.

All citations are into /root/reference/ (NVlabs/BundleSDF @ 2025-01-03).  Nothing here
reads that tree at run time.

Conventions: R rays, S = N_samples + N_samples_around_depth samples per ray, L hash
levels, C = 2 features per level, F keyframes.
"""
import math
import numpy as np
import torch
import torch.nn.functional as Fnn

f32 = np.float32


# --------------------------------------------------------------------------------------
# a8  multires hash grid  (mycuda/torch_ngp_grid_encoder/grid.py, gridencoder.cu)
# --------------------------------------------------------------------------------------
class HashGeometry:
    """Level table of the hash grid.

    Allocation follows grid.py:110,127-134 (float64 host arithmetic):
      per_level_scale = exp2(log2(desired/base)/(L-1)); resolution_i = ceil(base*s^i);
      params_i = min(2^log2_T, (resolution_i+1)^3) rounded up to a multiple of 8.
    Indexing constants follow gridencoder.cu:154-156 (float32):
      S = (float)log2(per_level_scale); scale = exp2f(level*S)*H - 1; resolution = ceil(scale)+1.
    The reference evaluates exp2f on the device; here (and in the HIP product) the
    per-level (scale, resolution) pairs are evaluated ONCE on the host with a correctly
    rounded exp2f and handed to the kernels, so oracle, product and the reference's kernel
    compiled as host code (oracle/_ref) share bit-identical level constants.
    """

    def __init__(self, n_levels=16, level_dim=2, base_resolution=16, log2_hashmap_size=19,
                 desired_resolution=256, input_dim=3):
        assert level_dim == 2 and input_dim == 3
        self.L, self.C, self.D = int(n_levels), int(level_dim), 3
        self.H = int(base_resolution)
        self.log2_T = int(log2_hashmap_size)
        self.per_level_scale = np.exp2(np.log2(desired_resolution / base_resolution) / (n_levels - 1))
        offsets, offset = [], 0
        max_params = 2 ** log2_hashmap_size
        for i in range(n_levels):
            resolution = int(np.ceil(base_resolution * self.per_level_scale ** i))
            params_in_level = min(max_params, (resolution + 1) ** input_dim)
            params_in_level = int(np.ceil(params_in_level / 8) * 8)
            offsets.append(offset)
            offset += params_in_level
        offsets.append(offset)
        self.offsets = np.array(offsets, dtype=np.int64)
        self.n_entries = int(offset)
        S = f32(np.log2(self.per_level_scale))           # grid.py:45 -> `const float S`
        self.S = S
        lv = np.arange(n_levels, dtype=np.uint32).astype(f32)
        # exp2f correctly rounded (float64 exp2, one rounding): bit-equal to glibc's exp2f, i.e. to gridencoder.cu:155
        # compiled as host code (tests/test_ref_native.py::test_level_constants_bitwise); NumPy's own float32 exp2
        # is 1 ulp off on some levels
        self.scale = (np.exp2((lv * S).astype(f32).astype(np.float64)).astype(f32) * f32(self.H) - f32(1.0)).astype(f32)
        self.resolution = (np.ceil(self.scale).astype(np.int64) + 1)
        self.size = (self.offsets[1:] - self.offsets[:-1]).astype(np.int64)
        # dense-vs-hash decision of get_grid_index (gridencoder.cu:67-80)
        self.hashed = np.zeros(n_levels, dtype=bool)
        for l in range(n_levels):
            stride, d = 1, 0
            while d < 3 and stride <= self.size[l]:
                stride *= int(self.resolution[l]) + 1
                d += 1
            self.hashed[l] = stride > self.size[l]

    @property
    def out_dim(self):
        return self.L * self.C


_PRIMES = (1, 2654435761, 805459861)
_M32 = 0xFFFFFFFF


def grid_index(geo, level, pg):
    """get_grid_index (gridencoder.cu:66-83) for integer corner coords pg [...,3] (int64)."""
    size = int(geo.size[level])
    res1 = int(geo.resolution[level]) + 1
    x, y, z = pg[..., 0], pg[..., 1], pg[..., 2]
    if geo.hashed[level]:
        idx = ((x * _PRIMES[0]) & _M32) ^ ((y * _PRIMES[1]) & _M32) ^ ((z * _PRIMES[2]) & _M32)
    else:
        idx = (x + y * res1 + z * res1 * res1) & _M32
    return idx % size


def hash_encode(x01, table, geo):
    """kernel_grid (gridencoder.cu:107-200) as differentiable torch ops.

    x01 [B,3] float32 in [0,1] (module input is (x+1)/2, grid.py:160); table [N,2] float32.
    Returns [B, L*C] level-major/channel-minor (grid.py:64).  Autograd of this function
    reproduces kernel_grid_backward (:250-336, scatter-add of w*grad) and
    kernel_input_backward (:340-365, dy_dx = scale*sum w'(f_right-f_left), :202-245).
    Out-of-range points give zeros and zero gradients (:128-152, :276-281).
    """
    B = x01.shape[0]
    oob = ((x01 < 0) | (x01 > 1)).any(dim=-1)
    xs = torch.where(oob[:, None], torch.zeros_like(x01), x01)
    outs = []
    for l in range(geo.L):
        scale = float(geo.scale[l])
        pos = xs * scale + 0.5
        pg_f = torch.floor(pos)
        frac = pos - pg_f
        pg = pg_f.detach().long()
        acc = torch.zeros(B, geo.C, dtype=x01.dtype)
        for idx in range(8):
            w = torch.ones(B, dtype=x01.dtype)
            corner = []
            for d in range(3):
                if (idx >> d) & 1:
                    w = w * frac[:, d]
                    corner.append(pg[:, d] + 1)
                else:
                    w = w * (1 - frac[:, d])
                    corner.append(pg[:, d])
            index = grid_index(geo, l, torch.stack(corner, -1)) + int(geo.offsets[l])
            acc = acc + w[:, None] * table[index]
        outs.append(torch.where(oob[:, None], torch.zeros_like(acc), acc))
    return torch.stack(outs, dim=1).reshape(B, geo.L * geo.C)


def hash_corner_indices(x01, geo):
    """Integer part only: [B, L, 8] int64 absolute table rows (bit-exact check of the HIP indices)."""
    x = x01.astype(f32)
    out = np.zeros((x.shape[0], geo.L, 8), dtype=np.int64)
    for l in range(geo.L):
        pos = (x * geo.scale[l]).astype(f32) + f32(0.5)
        pg = np.floor(pos).astype(np.int64)
        for idx in range(8):
            c = pg + np.array([(idx >> d) & 1 for d in range(3)], dtype=np.int64)
            out[:, l, idx] = grid_index(geo, l, torch.from_numpy(c)).numpy() + geo.offsets[l]
    return out


# --------------------------------------------------------------------------------------
# a9  spherical harmonics of the world view direction (nerf_helpers.py:22-105)
# --------------------------------------------------------------------------------------
SH_C0 = 0.28209479177387814
SH_C1 = 0.4886025119029199
SH_C2 = [1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396]
SH_C3 = [-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154,
         -0.4570457994644658, 1.445305721320277, -0.5900435899266435]


def sh_encode(dirs, degree=3):
    """SHEncoder.forward (nerf_helpers.py:67-105), degree <= 4."""
    x, y, z = dirs.unbind(-1)
    res = [torch.full_like(x, SH_C0)]
    if degree > 1:
        res += [-SH_C1 * y, SH_C1 * z, -SH_C1 * x]
    if degree > 2:
        xx, yy, zz = x * x, y * y, z * z
        xy, yz, xz = x * y, y * z, x * z
        res += [SH_C2[0] * xy, SH_C2[1] * yz, SH_C2[2] * (2.0 * zz - xx - yy), SH_C2[3] * xz, SH_C2[4] * (xx - yy)]
    if degree > 3:
        res += [SH_C3[0] * y * (3 * xx - yy), SH_C3[1] * xy * z, SH_C3[2] * y * (4 * zz - xx - yy),
                SH_C3[3] * z * (2 * zz - 3 * xx - 3 * yy), SH_C3[4] * x * (4 * zz - xx - yy),
                SH_C3[5] * z * (xx - yy), SH_C3[6] * x * (xx - 3 * yy)]
    return torch.stack(res, dim=-1)


# --------------------------------------------------------------------------------------
# a6  per-frame pose corrections (nerf_helpers.py:127-154 + pytorch3d se3_exp_map)
# --------------------------------------------------------------------------------------
def _hat(v):
    z = torch.zeros_like(v[:, 0])
    return torch.stack([torch.stack([z, -v[:, 2], v[:, 1]], -1),
                        torch.stack([v[:, 2], z, -v[:, 0]], -1),
                        torch.stack([-v[:, 1], v[:, 0], z], -1)], dim=1)


def se3_exp(log_transform, eps=1e-4):
    """pytorch3d.transforms.se3_exp_map restated (third-party, absent here: docker/dockerfile:79
    `pytorch3d@stable`, unpinned).  Published algorithm: theta = sqrt(clamp(|w|^2, eps));
    R = I + sin(theta)/theta K + (1-cos theta)/theta^2 K^2;  V = I + (1-cos theta)/theta^2 K +
    (theta - sin theta)/theta^3 K^2;  t = V u.  Returned here in COLUMN convention [[R,t],[0,1]],
    i.e. what nerf_helpers.py:150 holds after its `.permute(0,2,1)`.
    """
    u, w = log_transform[:, :3], log_transform[:, 3:]
    nrms = (w * w).sum(1)
    th = torch.clamp(nrms, eps).sqrt()
    K = _hat(w)
    K2 = torch.bmm(K, K)
    inv = 1.0 / th
    fac1 = inv * th.sin()
    fac2 = inv * inv * (1.0 - th.cos())
    eye = torch.eye(3, dtype=w.dtype)[None]
    Rm = fac1[:, None, None] * K + fac2[:, None, None] * K2 + eye
    V = eye + K * ((1 - torch.cos(th)) / (th ** 2))[:, None, None] + K2 * ((th - torch.sin(th)) / (th ** 3))[:, None, None]
    t = torch.bmm(V, u[:, :, None])[:, :, 0]
    T = torch.zeros(w.shape[0], 4, 4, dtype=w.dtype)
    T[:, :3, :3] = Rm
    T[:, :3, 3] = t
    T[:, 3, 3] = 1.0
    return T


def pose_matrices(pose_data, max_trans, max_rot_deg):
    """PoseArray.get_matrices for ALL frames (nerf_helpers.py:143-154): frame 0 is identity."""
    theta = torch.tanh(pose_data)
    trans = theta[:, :3] * max_trans
    rot = theta[:, 3:6] * max_rot_deg / 180.0 * np.pi
    Ts = se3_exp(torch.cat((trans, rot), dim=-1))
    eye = torch.eye(4, dtype=pose_data.dtype)[None]
    mask = torch.ones(pose_data.shape[0], dtype=torch.bool)
    mask[0] = False
    return torch.where(mask[:, None, None], Ts, eye.expand_as(Ts))


# --------------------------------------------------------------------------------------
# a10  NeRFSmall (nerf_helpers.py:243-321), parameterised like the constructor
# --------------------------------------------------------------------------------------
class FieldShape:
    def __init__(self, input_ch=32, input_ch_views=9, num_layers=2, hidden_dim=64, geo_feat_dim=15,
                 num_layers_color=3, hidden_dim_color=64):
        self.input_ch, self.input_ch_views = input_ch, input_ch_views
        self.num_layers, self.hidden_dim, self.geo_feat_dim = num_layers, hidden_dim, geo_feat_dim
        self.num_layers_color, self.hidden_dim_color = num_layers_color, hidden_dim_color

    def layer_dims(self):
        """[(out,in)] for sigma_net then color_net, exactly nerf_helpers.py:255-294
        (note color hidden->hidden uses hidden_dim, :283)."""
        s, c = [], []
        for l in range(self.num_layers):
            i = self.input_ch if l == 0 else self.hidden_dim
            o = 1 + self.geo_feat_dim if l == self.num_layers - 1 else self.hidden_dim
            s.append((o, i))
        for l in range(self.num_layers_color):
            i = self.input_ch_views + self.geo_feat_dim if l == 0 else self.hidden_dim
            o = 3 if l == self.num_layers_color - 1 else self.hidden_dim_color
            c.append((o, i))
        return s, c

    def n_params(self):
        s, c = self.layer_dims()
        return sum(o * i + o for o, i in s + c)


def init_mlp_params(shape):
    """Same construction order and initialisers as NeRFSmall.__init__ (nn.Linear default init,
    last sigma bias = 0.1, nerf_helpers.py:267-272,290), consuming the torch CPU RNG identically."""
    s, c = shape.layer_dims()
    params = []
    for k, (o, i) in enumerate(s):
        lin = torch.nn.Linear(i, o, bias=True)
        params.append([lin.weight.detach().clone(), lin.bias.detach().clone()])
    params[len(s) - 1][1].fill_(0.1)
    for (o, i) in c:
        lin = torch.nn.Linear(i, o, bias=True)
        params.append([lin.weight.detach().clone(), lin.bias.detach().clone()])
    return params


def _round_ste(t, dtype):
    """Round to a 16-bit operand type with a straight-through gradient (what autocast's casts do to values)."""
    if dtype is None:
        return t
    return t + (t.detach().to(dtype).float() - t.detach())


def _linear(h, W, b, operand_dtype):
    return Fnn.linear(_round_ste(h, operand_dtype), _round_ste(W, operand_dtype), b)


def mlp_forward(shape, params, x, operand_dtype=None, split_forward=False):
    """NeRFSmall.forward (nerf_helpers.py:305-321). x = [hash(input_ch) | views(input_ch_views)];
    returns [rgb_raw(3), sdf(1)].  operand_dtype=torch.float16/bfloat16 restates the autocast path the reference
    trains with (nerf_runner.py:1289-1294: 16-bit GEMM operands, fp32 accumulate, fp32 bias).

    split_forward (with an operand_dtype): the model of the product's 'fp16x3' / 'bf16x3' modes -- VALUES of the outputs and
    of the sigma head's hand-off to the colour net are the fp32 ones (the forward kernels carry operands as hi + lo), while
    the GRADIENT flows through the 16-bit-rounded network (the backward kernels recompute with plain 16-bit operands)."""
    if split_forward and operand_dtype is not None:
        with torch.no_grad():
            exact = mlp_forward(shape, params, x, None)
            h32 = x[:, :shape.input_ch]
            for l in range(shape.num_layers):
                h32 = _linear(h32, params[l][0], params[l][1], None)
                if l != shape.num_layers - 1:
                    h32 = torch.relu(h32)
    ns, nc = shape.num_layers, shape.num_layers_color
    h = x[:, :shape.input_ch]
    views = x[:, shape.input_ch:]
    for l in range(ns):
        W, b = params[l]
        h = _linear(h, W, b, operand_dtype)
        if l != ns - 1:
            h = torch.relu(h)
    if split_forward and operand_dtype is not None:
        h = h + (h32 - h).detach()                       # the hand-off carries the exact head output (then rounded as an operand)
    sigma, geo = h[:, 0], h[:, 1:]
    h = torch.cat([views, geo], dim=-1)
    for l in range(nc):
        W, b = params[ns + l]
        h = _linear(h, W, b, operand_dtype)
        if l != nc - 1:
            h = torch.relu(h)
    out = torch.cat([h, sigma[:, None]], dim=-1)
    if split_forward and operand_dtype is not None:
        out = out + (exact - out).detach()
    return out


def mlp_forward_sdf(shape, params, feat, operand_dtype=None):
    """NeRFSmall.forward_sdf (nerf_helpers.py:296-302)."""
    h = feat
    for l in range(shape.num_layers):
        W, b = params[l]
        h = _linear(h, W, b, operand_dtype)
        if l != shape.num_layers - 1:
            h = torch.relu(h)
    return h[:, 0]


# --------------------------------------------------------------------------------------
# a3  occupancy grid = what the kaolin SPC octree encodes (nerf_runner.py:436-476, Utils.py:360-373)
# --------------------------------------------------------------------------------------
def octree_levels(cfg):
    """max_level (build) and level (ray tracing): nerf_runner.py:444-447, 483-484."""
    sv = cfg['octree_smallest_voxel_size'] * cfg['sc_factor']
    max_level = int(np.ceil(np.log2(2.0 / sv)))
    rv = cfg['octree_raytracing_voxel_size'] * cfg['sc_factor']
    level = int(np.floor(np.log2(2.0 / rv)))
    return max_level, level


def build_occupancy(pts, cfg):
    """Occupied cells at max_level after 27-neighbour dilation, then the coarser `level` grid.

    nerf_runner.py:449-465 (dilate), :464-465 (centres clipped to [-1,1]); kaolin
    quantize_points(x, level) = floor(clamp(2^level*(x+1)/2, 0, 2^level-1)) (third-party, unpinned);
    a level-l cell is occupied iff any max_level descendant is.
    Returns (occ_max [n,n,n] bool, occ_level [m,m,m] bool, max_level, level) indexed [x,y,z].
    """
    max_level, level = octree_levels(cfg)
    vs = 2.0 / (2 ** max_level)
    dilate_radius = max(1, int(np.ceil(cfg['octree_dilate_size'] / cfg['octree_smallest_voxel_size'])))
    coords = np.floor((np.asarray(pts, dtype=np.float32) + 1) / np.float32(vs)).astype(np.int64)
    shifts = np.array([[dx, dy, dz] for dx in (-1, 0, 1) for dy in (-1, 0, 1) for dz in (-1, 0, 1)], dtype=np.int64)
    for _ in range(dilate_radius):
        coords = np.unique((coords[None] + shifts[:, None]).reshape(-1, 3), axis=0)
    centres = np.clip(((coords + 0.5) * vs - 1).astype(np.float32), -1, 1)
    n = 2 ** max_level
    q = np.floor(np.clip(n * (centres + 1.0) / 2.0, 0, n - 1.0)).astype(np.int64)
    occ = np.zeros((n, n, n), dtype=bool)
    occ[q[:, 0], q[:, 1], q[:, 2]] = True
    m = 2 ** level
    sh = max_level - level
    if sh >= 0:
        occ_l = occ.reshape(m, 1 << sh, m, 1 << sh, m, 1 << sh).any(axis=(1, 3, 5))
    else:
        raise ValueError("ray tracing level deeper than the octree")
    return occ, occ_l, max_level, level


# --------------------------------------------------------------------------------------
# a4  ray / occupied-voxel intersection  (kaolin unbatched_raytrace + common.cu:129-149)
# --------------------------------------------------------------------------------------
ZERO_DIR = f32(1e-20)
MIN_LEN = f32(1e-4)     # common.cu:142


def _axis_slabs(o_a, d_a, n):
    """Per-axis [tmin,tmax] of every cell index 0..n-1 for rays o_a,d_a [R] -> two [R,n] float32 arrays."""
    cs = f32(2.0) / f32(n)
    idx = np.arange(n, dtype=np.float32)
    lo = idx * cs - f32(1.0)
    hi = (idx + f32(1.0)) * cs - f32(1.0)
    zero = np.abs(d_a) < ZERO_DIR
    with np.errstate(divide='ignore', invalid='ignore', over='ignore'):
        inv = (f32(1.0) / np.where(zero, f32(1.0), d_a)).astype(f32)
        t0 = ((lo[None] - o_a[:, None]) * inv[:, None]).astype(f32)
        t1 = ((hi[None] - o_a[:, None]) * inv[:, None]).astype(f32)
    tmin = np.minimum(t0, t1)
    tmax = np.maximum(t0, t1)
    inside = (lo[None] <= o_a[:, None]) & (o_a[:, None] < hi[None])
    tmin = np.where(zero[:, None], np.where(inside, f32(-np.inf), f32(np.inf)), tmin).astype(f32)
    tmax = np.where(zero[:, None], np.where(inside, f32(np.inf), f32(-np.inf)), tmax).astype(f32)
    return tmin, tmax


def trace_rays(occ_l, rays_o, rays_d, max_hits=None):
    """OctreeManager.ray_trace (Utils.py:443-475) on the dense level-l occupancy grid -- DEFINITION.

    For every occupied cell: slab test in float32, t_in = max(tmin_x,tmin_y,tmin_z,0),
    t_out = min(tmax_x,tmax_y,tmax_z); the cell is crossed iff t_in <= t_out.  Crossed cells are
    ordered front-to-back (t_in, then t_out, then cell id) and filtered exactly like
    postprocessOctreeRayTracingKernel (common.cu:137-147): stop at the first entry whose in or
    out is 0, skip in>out and |out-in|<1e-4.  kaolin itself is third-party, absent and unpinned
    (docker/dockerfile:84): "parity unpinned" for this function; the geometric definition above is
    what "bit-identical ray-hit indices" is checked against.

    Returns (t_in_out [R,H,2] float32 zero-padded, cell_ids [R,H] int32 (-1 pad), n_hits [R] int32)
    with H = max hits (or max_hits).  cell id = (x*n + y)*n + z.
    """
    flat_ray, flat_io, flat_cid = trace_rays_flat(occ_l, rays_o, rays_d)
    R = np.asarray(rays_o).shape[0]
    return postprocess_hits(flat_ray, flat_io, R, max_hits=max_hits, cell_ids=flat_cid)


def trace_rays_flat(occ_l, rays_o, rays_d):
    """The flat list kaolin's unbatched_raytrace hands back (Utils.py:457), per the geometric definition of trace_rays:
    ray_index [M] int64 (ascending), depth_in_out [M,2] float32 front-to-back within a ray, cell id [M] int64 --
    every occupied cell whose slab test passes, BEFORE the filters of common.cu:137-147."""
    o = np.ascontiguousarray(rays_o, dtype=f32)
    d = np.ascontiguousarray(rays_d, dtype=f32)
    R = o.shape[0]
    n = occ_l.shape[0]
    cells = np.argwhere(occ_l)                                     # [M,3] sorted by cell id
    ids = ((cells[:, 0] * n + cells[:, 1]) * n + cells[:, 2]).astype(np.int64)
    fr, fio, fc = [], [], []
    for r0 in range(0, R, 256):
        r1 = min(R, r0 + 256)
        sl = [_axis_slabs(o[r0:r1, a], d[r0:r1, a], n) for a in range(3)]
        tin = np.maximum(np.maximum(sl[0][0][:, cells[:, 0]], sl[1][0][:, cells[:, 1]]), sl[2][0][:, cells[:, 2]])
        tin = np.maximum(tin, f32(0.0))
        tout = np.minimum(np.minimum(sl[0][1][:, cells[:, 0]], sl[1][1][:, cells[:, 1]]), sl[2][1][:, cells[:, 2]])
        hit = tin <= tout
        for r in range(r1 - r0):
            h = np.nonzero(hit[r])[0]
            h = h[np.lexsort((ids[h], tout[r, h], tin[r, h]))]
            fr.append(np.full(len(h), r0 + r, dtype=np.int64))
            fio.append(np.stack([tin[r, h], tout[r, h]], -1).astype(f32).reshape(-1, 2))
            fc.append(ids[h])
    if not fr:
        return np.zeros(0, np.int64), np.zeros((0, 2), f32), np.zeros(0, np.int64)
    return np.concatenate(fr), np.concatenate(fio), np.concatenate(fc)


def postprocess_hits(ray_index, depth_in_out, N_rays, max_hits=None, cell_ids=None):
    """Utils.py:466-470 + postprocessOctreeRayTracingKernel (common.cu:129-149): per hit ray copy its run of (in,out)
    pairs into a zero-padded [N_rays, H, 2] table; stop at the first entry whose in or out is 0, skip in>out and
    |out-in|<1e-4.  With max_hits=None, H is the reference's max_intersections (longest run BEFORE filtering,
    Utils.py:467) when cell_ids is None -- the exact shape the reference returns -- or the longest kept run when the
    caller also wants cell ids (trace_rays).  PINNED against the reference's kernel compiled as host code
    (tests/test_ref_native.py)."""
    ray_index = np.asarray(ray_index, np.int64)
    dio = np.asarray(depth_in_out, f32).reshape(-1, 2)
    M = ray_index.shape[0]
    starts = np.flatnonzero(np.concatenate([[True], ray_index[1:] != ray_index[:-1]])) if M else np.zeros(0, np.int64)
    ends = np.concatenate([starts[1:], [M]]) if M else starts
    kept = {}
    H_ref = int((ends - starts).max()) if M else 1
    H_kept = 0
    for s0, s1 in zip(starts, ends):
        r = int(ray_index[s0])
        keep_t, keep_c = [], []
        for i in range(s0, s1):
            a, b = dio[i, 0], dio[i, 1]
            if a == 0 or b == 0:
                break
            if a > b:
                continue
            if abs(b - a) < MIN_LEN:
                continue
            keep_t.append((a, b))
            keep_c.append(-1 if cell_ids is None else cell_ids[i])
        kept[r] = (keep_t, keep_c)           # a ray id that re-appears later overwrites (the kernel would too)
        H_kept = max(H_kept, len(keep_t))
    H = max_hits if max_hits is not None else (H_ref if cell_ids is None else H_kept)
    H = max(H, 1)
    tio = np.zeros((N_rays, H, 2), dtype=f32)
    cid = -np.ones((N_rays, H), dtype=np.int32)
    nh = np.zeros(N_rays, dtype=np.int32)
    for r, (kt, kc) in kept.items():
        m = min(len(kt), H)
        nh[r] = m
        if m:
            tio[r, :m] = np.array(kt[:m], dtype=f32)
            cid[r, :m] = np.array(kc[:m], dtype=np.int32)
    if cell_ids is None:
        return tio
    return tio, cid, nh


# --------------------------------------------------------------------------------------
# a5  z sampling (nerf_runner.py:67-87, 979-1011, 1063-1081; common.cu:41-105)
# --------------------------------------------------------------------------------------
def linspace01(N):
    """torch.linspace(0,1,N) in float32 (pinned against torch itself in tests/test_oracle.py):
    start + i*step for i < N/2 and fma(-step, N-1-i, end) above (one rounding: ATen's vectorised CPU
    kernel and the nvcc-contracted CUDA kernel both fuse the multiply-add)."""
    step = f32(1.0) / f32(N - 1)
    i = np.arange(N)
    lo = (i.astype(f32) * step).astype(f32)
    hi = (1.0 - np.float64(step) * (N - 1 - i).astype(np.float64)).astype(f32)   # exact product, one rounding
    return np.where(i < N // 2, lo, hi).astype(f32)


def sample_rays_uniform(N, near, far, u):
    """sample_rays_uniform (nerf_runner.py:67-87), perturb=True with injected uniforms u [R,N]; u = None: perturb=False
    (render_images, :597) -- the linspace itself, lines :78-85 (jitter and clip) are skipped."""
    near = near.reshape(-1, 1).astype(f32)
    far = far.reshape(-1, 1).astype(f32)
    t = linspace01(N).reshape(1, -1)
    z = (near * (f32(1.0) - t)).astype(f32) + (far * t).astype(f32)
    z = z.astype(f32)
    if u is None:
        return z
    mids = (f32(0.5) * (z[:, 1:] + z[:, :-1]).astype(f32)).astype(f32)
    upper = np.concatenate([mids, z[:, -1:]], -1)
    lower = np.concatenate([z[:, :1], mids], -1)
    z = (lower + ((upper - lower).astype(f32) * u.astype(f32)).astype(f32)).astype(f32)
    return np.clip(z, near, far).astype(f32)


def walk_boxes(z_in_out, z_cont, return_spin=False):
    """sample_rays_uniform_occupied_voxels_kernel (common.cu:41-105): map a distance along the
    concatenated occupied length back into the per-voxel intervals.  Where the reference prints
    an error and spins forever (:66-72,:87-93: the remaining distance exceeds eps=1e-4 past the last
    box, or the very first box is a terminator) this returns the end of the last valid box and, with
    return_spin, a [R,N] bool mask of those samples (the HIP product raises its error flag there).
    PINNED against the reference's kernel compiled as host code (tests/test_ref_native.py)."""
    R, N = z_cont.shape
    Hn = z_in_out.shape[1]
    out = np.zeros((R, N), dtype=f32)
    spin = np.zeros((R, N), dtype=bool)
    eps = f32(1e-4)
    for r in range(R):
        if z_in_out[r, 0, 0] == 0:
            continue
        for s in range(N):
            zr = f32(z_cont[r, s])
            ib = 0
            while True:
                if ib >= Hn or z_in_out[r, ib, 0] == 0:
                    out[r, s] = z_in_out[r, max(ib - 1, 0), 1]
                    spin[r, s] = not (zr <= eps and (ib >= Hn or ib >= 1))
                    break
                bl = f32(z_in_out[r, ib, 1] - z_in_out[r, ib, 0])
                if zr <= bl:
                    out[r, s] = f32(z_in_out[r, ib, 0] + zr)
                    break
                zr = f32(zr - bl)
                ib += 1
    return (out, spin) if return_spin else out


def sample_occupied(t_in_out, viewdir_cam_z, N, u, depths=None, trunc=None, near_sc=None, far_sc=None):
    """NerfRunner.sample_rays_uniform_occupied_voxels (nerf_runner.py:979-1011)."""
    z = (t_in_out * np.abs(viewdir_cam_z).reshape(-1, 1, 1).astype(f32)).astype(f32)
    if depths is not None:
        dep = depths.reshape(-1, 1).astype(f32)
        valid = (dep >= f32(near_sc)) & (dep <= f32(far_sc))
        valid = valid & (z > 0).all(axis=-1)
        cap = (dep + f32(trunc)).astype(f32)[:, :, None]
        zc = np.minimum(np.maximum(z, f32(0.0)), cap)
        z = np.where(valid[:, :, None], zc, z).astype(f32)
    lens = (z[:, :, 1] - z[:, :, 0]).astype(f32)
    total = np.zeros(z.shape[0], dtype=f32)
    for k in range(z.shape[1]):                                   # sequential fp32 sum (definition)
        total = (total + lens[:, k]).astype(f32)
    z_cont = sample_rays_uniform(N, np.zeros_like(total), total, u)
    return walk_boxes(z, z_cont), z_cont, z


def sample_z(t_in_out, viewdir_cam_z, depth, cfg, trunc, u_occ, u_dep):
    """z_vals [R, N_samples + N_samples_around_depth] of render_rays (nerf_runner.py:1061-1081).
    u_occ [R,N_samples], u_dep [R,N_around] are the uniforms torch.rand would have produced."""
    Ns, Na = cfg['N_samples'], cfg['N_samples_around_depth']
    near_sc = f32(cfg['near'] * cfg['sc_factor'])
    far_sc = f32(cfg['far'] * cfg['sc_factor'])
    depth = depth.astype(f32)
    z_occ, _, _ = sample_occupied(t_in_out, viewdir_cam_z, Ns, u_occ, depths=depth, trunc=trunc,
                                  near_sc=near_sc, far_sc=far_sc)
    valid = (depth >= near_sc) & (depth <= far_sc)
    near_d = (depth - f32(trunc)).astype(f32)
    far_d = (depth + f32(f32(trunc) * f32(cfg['neg_trunc_ratio']))).astype(f32)
    z_a = sample_rays_uniform(Na, near_d, far_d, u_dep)
    if (~valid).any():
        z_inv, _, _ = sample_occupied(t_in_out, viewdir_cam_z, Na, u_dep, depths=None)   # (u_dep None: perturb=False)
        z_a = np.where(valid[:, None], z_a, z_inv).astype(f32)
    return np.concatenate([z_occ, z_a], axis=-1).astype(f32)


# --------------------------------------------------------------------------------------
# a11/a12  compositing + losses (nerf_runner.py:1132-1169, 679-752; nerf_helpers.py:367-399)
# --------------------------------------------------------------------------------------
def get_truncation(cfg, global_step=0):
    """NerfRunner.get_truncation (nerf_runner.py:663-676)."""
    if cfg.get('trunc_decay_type', '') == 'linear':
        t = cfg['trunc_start'] - (cfg['trunc_start'] - cfg['trunc']) * float(global_step) / cfg['n_step']
    elif cfg.get('trunc_decay_type', '') == 'exp':
        lamb = np.log(cfg['trunc'] / cfg['trunc_start']) / (cfg['n_step'] / 4)
        t = max(cfg['trunc_start'] * np.exp(global_step * lamb), cfg['trunc'])
    else:
        t = cfg['trunc']
    return t * cfg['sc_factor']


def raw2outputs(raw, z_vals, depth, valid_samples, cfg, truncation):
    """NerfRunner.raw2outputs (nerf_runner.py:1132-1169): depth-guided weights (the `sdf`
    argument of sdf2weights is unused, :1152-1161)."""
    depth = depth.view(-1, 1)
    sfd = (depth - z_vals) / truncation
    w = torch.sigmoid(sfd * cfg['sdf_lambda']) * torch.sigmoid(-sfd * cfg['sdf_lambda'])
    invalid = (depth > cfg['far'] * cfg['sc_factor']).reshape(-1)
    mask = (z_vals - depth <= truncation * cfg['neg_trunc_ratio']) & (z_vals - depth >= -truncation)
    w = torch.where(invalid[:, None], torch.zeros_like(w), w * mask)
    w = w / (w.sum(dim=-1, keepdim=True) + 1e-10)
    rgb = torch.sigmoid(raw[..., :3])
    w = torch.where(valid_samples, w, torch.zeros_like(w))
    rgb_map = torch.sum(w[..., None] * rgb, -2)
    return rgb_map, w


def losses(rgb_map, raw, z_vals, valid_samples, batch, cfg, truncation, first_frame_weight=None):
    """train_loop loss assembly (nerf_runner.py:680-727) + get_sdf_loss (nerf_helpers.py:382-399).
    batch columns: dir 0-2, rgb 3-5, depth 6, mask 7, frame 8, type 9, near 10, far 11."""
    target_s, target_d = batch[:, 3:6], batch[:, 6]
    frame_ids, ray_type = batch[:, 8], batch[:, 9]
    sdf = raw[..., -1]
    R, S = sdf.shape
    valid_rays = (valid_samples > 0).any(dim=-1) & (ray_type == 0)
    ray_w = torch.ones(R, dtype=torch.float32)
    ray_w[frame_ids == 0] = cfg['first_frame_weight']
    ray_w = ray_w * valid_rays
    sw = ray_w.view(R, 1).expand(-1, S) * valid_samples
    img_loss = ((rgb_map - target_s) ** 2 * ray_w.view(-1, 1)).mean()
    rgb_loss = cfg['rgb_weight'] * img_loss
    sw = torch.where((ray_type == 1)[:, None], torch.zeros_like(sw), sw)
    td = target_d.reshape(-1, 1).expand(-1, S)
    far_sc = cfg['far'] * cfg['sc_factor']
    near_sc = cfg['near'] * cfg['sc_factor']
    valid_depth = (td >= near_sc) & (td <= far_sc)
    front = z_vals < td - truncation
    back = z_vals > td + truncation * cfg['neg_trunc_ratio']
    sdf_mask = ((1.0 - front.float()) * (1.0 - back.float()) * valid_depth).bool()
    m = (td > far_sc) & (sdf < cfg['fs_sdf'])
    fs_loss = torch.mean(((sdf - cfg['fs_sdf']) * m) ** 2 * sw) * 0.5
    m = front & (td <= far_sc) & (sdf < 1)
    empty_loss = torch.mean(torch.abs(sdf - 1) * m * sw) * cfg['empty_weight']
    fs_loss = fs_loss + empty_loss
    sdf_loss = torch.mean(((z_vals + sdf * truncation) * sdf_mask - td * sdf_mask) ** 2 * sw) * 0.5
    fs_loss = fs_loss * cfg['fs_weight']
    sdf_loss = sdf_loss * cfg['trunc_weight']
    out = {'rgb_loss': rgb_loss, 'fs_loss': fs_loss, 'sdf_loss': sdf_loss}
    loss = rgb_loss + fs_loss + sdf_loss
    if cfg.get('fs_rgb_weight', 0) > 0:
        fs_rgb = (((torch.sigmoid(raw[..., :3]) - 1) * front[..., None]) ** 2 * sw[..., None]).mean()
        loss = loss + fs_rgb * cfg['fs_rgb_weight']
        out['fs_rgb_loss'] = fs_rgb
    out['loss'] = loss
    return out


# --------------------------------------------------------------------------------------
# the whole field + one optimisation step (nerf_runner.py:679-763, 1014-1129, 1227-1304)
# --------------------------------------------------------------------------------------
class OracleField:
    def __init__(self, cfg, geo, shape, n_frames, c2w, occ_l, table=None, mlp=None, pose=None, feat=None,
                 operand_dtype=None, split_forward=False):
        self.cfg, self.geo, self.shape, self.F = cfg, geo, shape, n_frames
        self.operand_dtype = operand_dtype
        self.split_forward = split_forward
        self.c2w = torch.as_tensor(c2w, dtype=torch.float32)
        self.occ_l = occ_l
        ff = cfg.get('frame_features', 0)
        self.table = (torch.empty(geo.n_entries, 2).uniform_(-1e-4, 1e-4) if table is None
                      else torch.as_tensor(table, dtype=torch.float32).clone()).requires_grad_(True)
        mlp = init_mlp_params(shape) if mlp is None else mlp
        self.mlp = [[torch.as_tensor(W).clone().float().requires_grad_(True),
                     torch.as_tensor(b).clone().float().requires_grad_(True)] for W, b in mlp]
        self.feat = None
        if ff > 0:
            self.feat = (torch.normal(0, 1, size=[n_frames, ff]).float() if feat is None
                         else torch.as_tensor(feat).float().clone()).requires_grad_(True)
        self.pose = None
        if cfg.get('optimize_poses', 1):
            self.pose = (torch.zeros(n_frames, 6) if pose is None else torch.as_tensor(pose).float().clone()).requires_grad_(True)
        self.global_step = 0
        self.pose_table = None      # optional [F,3,4]: trace_and_sample composes the world rays from it, in the product's float32 order
        self.N_iters = cfg['n_step'] + 1
        self._make_optimizer()

    def _make_optimizer(self):
        """create_optimizer (nerf_runner.py:492-504): torch.optim.Adam itself (the reference's dependency)."""
        basic = [self.table] + [p for wb in self.mlp for p in wb]
        if self.feat is not None:
            basic.append(self.feat)
        groups = [{'name': 'basic', 'params': basic, 'lr': self.cfg['lrate']}]
        if self.pose is not None:
            groups.append({'name': 'pose_array', 'params': [self.pose], 'lr': self.cfg['lrate_pose']})
        self.optimizer = torch.optim.Adam(groups, betas=(0.9, 0.999), weight_decay=0, eps=1e-15)
        self.init_lrs = [g['lr'] for g in groups]

    def frame_tf(self):
        tf = self.c2w
        if self.pose is not None:
            Ts = pose_matrices(self.pose, self.cfg['max_trans'] * self.cfg['sc_factor'], self.cfg['max_rot'])
            tf = Ts @ tf
        return tf

    def trace_and_sample(self, batch, u_occ, u_dep):
        """render_rays up to z_vals (nerf_runner.py:1044-1081), no_grad."""
        cfg = self.cfg
        with torch.no_grad():
            rays_d = batch[:, 0:3]
            viewdirs = rays_d / rays_d.norm(dim=-1, keepdim=True)
            fid = batch[:, 8].long()
            if getattr(self, 'pose_table', None) is not None:
                # The world-frame ray in the product's DOCUMENTED float32 order (csrc/nof_trace.hip: k_raymarch_wave / k_batch_trace),
                # from a given pose table tf [F,3,4] = Delta(pose) c2w (the product's own, whose distance to frame_tf() the caller
                # asserts): the reference composes the same quantities with a batched matmul whose summation order is the BLAS's;
                # the index work that follows (cells, intervals, z) is discontinuous in the ray, so the two programs must agree on
                # the ray's BITS before their indices can be compared one for one (tests/test_gpu_fullsize.py).
                T = np.asarray(self.pose_table, dtype=f32).reshape(-1, 3, 4)[fid.numpy()]
                r = batch[:, 0:3].numpy().astype(f32)
                nrm = np.sqrt(((r[:, 0] * r[:, 0] + r[:, 1] * r[:, 1]).astype(f32) + r[:, 2] * r[:, 2]).astype(f32)).astype(f32)
                v = (r / nrm[:, None]).astype(f32)
                d = ((T[:, :, 0] * v[:, 0:1] + T[:, :, 1] * v[:, 1:2]).astype(f32) + T[:, :, 2] * v[:, 2:3]).astype(f32)
                rays_o_w, viewdirs_w = torch.from_numpy(np.ascontiguousarray(T[:, :, 3])), torch.from_numpy(d)
                viewdirs = torch.from_numpy(v)
            else:
                tf = self.frame_tf()[fid]
                rays_o_w = tf[:, :3, 3]
                viewdirs_w = (tf[:, :3, :3] @ viewdirs[:, :, None])[:, :, 0]
            tio, cid, nh = trace_rays(self.occ_l, rays_o_w.numpy(), viewdirs_w.numpy())
            trunc = get_truncation(cfg, self.global_step)
            z = sample_z(tio, viewdirs[:, 2].numpy(), batch[:, 6].numpy(), cfg, trunc,
                         None if u_occ is None else np.asarray(u_occ, dtype=f32),
                         None if u_dep is None else np.asarray(u_dep, dtype=f32))
        return torch.from_numpy(z), dict(t_in_out=tio, cell_ids=cid, n_hits=nh, rays_o_w=rays_o_w, viewdirs_w=viewdirs_w)

    def forward(self, batch, z_vals):
        """run_network + raw2outputs (nerf_runner.py:1083-1088, 1227-1304); amp disabled (fp32)."""
        cfg = self.cfg
        R, S = z_vals.shape
        rays_d = batch[:, 0:3]
        viewdirs = rays_d / rays_d.norm(dim=-1, keepdim=True)
        fid = batch[:, 8].long()
        tf = self.frame_tf()[fid]
        pts = rays_d[:, None, :] * z_vals[:, :, None]
        pts_w = (tf[:, None, :3, :3] @ pts[..., None])[..., 0] + tf[:, None, :3, 3]
        flat = pts_w.reshape(-1, 3)
        eik = float(cfg.get('eikonal_weight', 0)) > 0
        if eik and not flat.requires_grad:
            flat = flat.detach().requires_grad_(True)
        valid = (torch.abs(flat) <= 1).all(dim=-1)
        emb = torch.zeros(flat.shape[0], self.geo.out_dim)
        x01 = (flat[valid] + 1) / 2
        emb[valid] = hash_encode(x01, self.table, self.geo)
        parts = [emb]
        if self.feat is not None:
            parts.append(self.feat[fid][:, None].expand(-1, S, -1).reshape(R * S, -1))
        dirs_w = (tf[:, :3, :3] @ viewdirs[:, :, None])[:, :, 0]
        sh = sh_encode(dirs_w, cfg['multires_views'])
        parts.append(sh[:, None].expand(-1, S, -1).reshape(R * S, -1))
        raw = mlp_forward(self.shape, self.mlp, torch.cat(parts, -1), self.operand_dtype, self.split_forward).reshape(R, S, 4)
        normals = None
        if eik:
            # the field's normal = d sdf / d x_world with grad_outputs = 1: the only meaningful normal path of the reference
            # (run_network_density, nerf_runner.py:1342-1345; train_loop's own is dead code: :686, :1297-1302 give zeros).
            # create_graph: the eikonal term is differentiated again (table, sigma net, poses).
            (normals,) = torch.autograd.grad(raw[..., 3].sum(), flat, create_graph=True, allow_unused=True)
            normals = torch.zeros_like(flat) if normals is None else normals
        valid = valid.view(R, S)
        trunc = get_truncation(cfg, self.global_step)
        rgb_map, w = raw2outputs(raw, z_vals, batch[:, 6], valid, cfg, trunc)
        return dict(raw=raw, rgb_map=rgb_map, weights=w, valid_samples=valid, pts_w=pts_w, normals=normals)

    def loss(self, batch, z_vals, fwd):
        cfg = self.cfg
        trunc = get_truncation(cfg, self.global_step)
        out = losses(fwd['rgb_map'], fwd['raw'], z_vals, fwd['valid_samples'], batch, cfg, trunc)
        loss = out['loss']
        if self.feat is not None:
            loss = loss + cfg['feature_reg_weight'] * (self.feat ** 2).mean()       # nerf_runner.py:745-747
        if self.pose is not None and cfg.get('pose_reg_weight', 0) > 0:
            loss = loss + cfg['pose_reg_weight'] * self.pose[1:].norm()             # :749-752
        if float(cfg.get('eikonal_weight', 0)) > 0:                                 # :734-738
            sdf = fwd['raw'][..., 3].reshape(-1)
            eikonal = ((torch.norm(fwd['normals'][sdf < 1], dim=-1) - 1) ** 2).mean() * cfg['eikonal_weight']
            out['eikonal_loss'] = eikonal
            loss = loss + eikonal
        out['loss'] = loss
        return out

    def all_params(self):
        ps = [self.table] + [p for wb in self.mlp for p in wb]
        if self.feat is not None:
            ps.append(self.feat)
        if self.pose is not None:
            ps.append(self.pose)
        return ps

    def train_step(self, batch, u_occ, u_dep, do_step=True):
        """train_loop (nerf_runner.py:679-763) with GradScaler disabled."""
        batch = torch.as_tensor(batch, dtype=torch.float32)
        z_vals, tr = self.trace_and_sample(batch, u_occ, u_dep)
        fwd = self.forward(batch, z_vals)
        out = self.loss(batch, z_vals, fwd)
        self.optimizer.zero_grad()
        out['loss'].backward()
        grads = [None if p.grad is None else p.grad.detach().clone() for p in self.all_params()]
        if do_step:
            self.optimizer.step()
            if self.global_step % 10 == 0 and self.global_step > 0:
                self.schedule_lr()
            self.global_step += 1
        return dict(z_vals=z_vals, trace=tr, fwd=fwd, losses=out, grads=grads)

    def render_rays_image(self, rays, chunk=None):
        """render_images up to the per-ray results (nerf_runner.py:586-613): the rays of one keyframe through render ->
        batchify_rays (chunk = N_rand, :595-598) -> render_rays with perturb=False and the rays' own depth, then
        depth = z_vals at the first SDF sign change (torch.argmax of the mask: index 0 when there is none), far*sc_factor for rays
        whose neighbouring SDF products are all > 0 (:604-612).  Returns rgb [n,3], depth [n] and the extras the reference keeps."""
        cfg = self.cfg
        rays = torch.as_tensor(rays, dtype=torch.float32)
        chunk = int(chunk or cfg['N_rand'])
        outs = dict(rgb_map=[], raw=[], z_vals=[], valid_samples=[], cell_ids=[], n_hits=[])
        with torch.no_grad():
            for i in range(0, rays.shape[0], chunk):
                b = rays[i:i + chunk]
                z_vals, tr = self.trace_and_sample(b, None, None)
                fwd = self.forward(b, z_vals)
                outs['rgb_map'].append(fwd['rgb_map']); outs['raw'].append(fwd['raw']); outs['z_vals'].append(z_vals)
                outs['valid_samples'].append(fwd['valid_samples'])
                outs['cell_ids'].append(tr['cell_ids']); outs['n_hits'].append(tr['n_hits'])
            cell_w = max(c.shape[1] for c in outs['cell_ids'])
            cells = np.concatenate([np.pad(c, ((0, 0), (0, cell_w - c.shape[1])), constant_values=-1) for c in outs['cell_ids']], 0)
            n_hits = np.concatenate(outs['n_hits'], 0)
            out = {k: torch.cat(v, 0) for k, v in outs.items() if k not in ('cell_ids', 'n_hits')}
            sdf = out['raw'][..., -1]
            signs = sdf[:, 1:] * sdf[:, :-1]
            empty_rays = (signs > 0).all(dim=-1)
            inds = torch.argmax((signs < 0).float(), axis=1)[..., None]
            depth = torch.gather(out['z_vals'], dim=1, index=inds)
            depth[empty_rays] = cfg['far'] * cfg['sc_factor']
        out.update(depth=depth.reshape(-1), cell_ids=cells, n_hits=n_hits)
        return out

    def schedule_lr(self):
        """nerf_runner.py:579-583."""
        for i, g in enumerate(self.optimizer.param_groups):
            g['lr'] = self.init_lrs[i] * (self.cfg['decay_rate'] ** (float(self.global_step) / self.N_iters))

    def query_sdf(self, pts):
        """run_network_density (nerf_runner.py:1307-1347): clip to [-1,1], hash, sigma_net only."""
        with torch.no_grad():
            x = torch.clip(torch.as_tensor(pts, dtype=torch.float32), -1, 1)
            feat = hash_encode((x + 1) / 2, self.table, self.geo)
            return mlp_forward_sdf(self.shape, self.mlp, feat, self.operand_dtype)


    def run_network_points(self, pts, viewdir=(0.0, 0.0, 0.0), frame_id=0):
        """run_network as mesh_vertex_color_from_network calls it (nerf_runner.py:1412-1424 -> :1226-1294): free-standing points,
        identity transform, ONE view direction (the reference passes the zero vector), the latent code of ONE frame (it passes
        frame 0).  Points outside [-1,1]^3 keep a zero embedding (:1246-1257).  raw [N,4]."""
        with torch.no_grad():
            x = torch.as_tensor(pts, dtype=torch.float32)
            valid = (torch.abs(x) <= 1).all(dim=-1)
            emb = torch.zeros(x.shape[0], self.geo.out_dim)
            emb[valid] = hash_encode((x[valid] + 1) / 2, self.table, self.geo)
            parts = [emb]
            if self.feat is not None:
                parts.append(self.feat[int(frame_id)][None].expand(x.shape[0], -1))
            sh = sh_encode(torch.tensor([list(viewdir)], dtype=torch.float32), self.cfg['multires_views'])
            parts.append(sh.expand(x.shape[0], -1))
            return mlp_forward(self.shape, self.mlp, torch.cat(parts, -1), self.operand_dtype, self.split_forward)


def adam_reference_step(p, g, m, v, t, lr, b1=0.9, b2=0.999, eps=1e-15):
    """torch.optim.Adam single-tensor update restated (what the HIP nof_adam_step must equal);
    t is the 1-based step count."""
    m = b1 * m + (1 - b1) * g
    v = b2 * v + (1 - b2) * g * g
    bc1 = 1 - b1 ** t
    bc2 = 1 - b2 ** t
    denom = np.sqrt(v) / math.sqrt(bc2) + eps
    p = p - (lr / bc1) * m / denom
    return p, m, v


def default_cfg(**over):
    """config.yml:2-102 keys that the hot path reads, with the file's default values."""
    cfg = dict(n_step=500, N_rand=2048, lrate=0.01, lrate_pose=0.01, decay_rate=0.1, amp=True,
               N_samples=128, N_samples_around_depth=64, N_importance=0, perturb=1, use_viewdirs=1,
               i_embed=1, i_embed_views=2, multires=8, multires_views=3, feature_grid_dim=2,
               raw_noise_std=0, finest_res=128, base_res=16, num_levels=4, log2_hashmap_size=22,
               use_octree=1, first_frame_weight=10, denoise_depth_use_octree_cloud=True,
               octree_embed_base_voxel_size=0.02, octree_smallest_voxel_size=0.02,
               octree_raytracing_voxel_size=0.02, octree_dilate_size=0.02, down_scale_ratio=1,
               bounding_box=[[-1, -1, -1], [1, 1, 1]], use_mask=1, dilate_mask_size=0,
               rays_valid_depth_only=True, near=0.1, far=2, rgb_weight=10, depth_weight=0, trunc=0.01,
               trunc_start=0.01, sdf_lambda=5, neg_trunc_ratio=1, trunc_decay_type='', fs_weight=100,
               empty_weight=0.01, fs_rgb_weight=0, trunc_weight=6000, frame_features=0, optimize_poses=1,
               pose_reg_weight=0, eikonal_weight=0, feature_reg_weight=0.1, fs_sdf=0.001,
               mesh_resolution=0.005, max_trans=0.02, max_rot=20, save_octree_clouds=False,
               tv_loss_weight=0, no_batching=0, chunk=99999999999, netchunk=6553600,
               i_print=999999, i_img=999999, i_weights=999999, i_mesh=999999, i_pose=999999,
               sc_factor=1.0, translation=np.zeros(3), save_dir='/tmp/nof', datadir='/tmp/nof')
    cfg.update(over)
    return cfg
