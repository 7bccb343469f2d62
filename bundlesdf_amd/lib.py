"""ctypes binding of libnof_hip.so (include/nof_hip.h).

The product path has NO fallback: if the shared library is missing or a call fails, an exception is
raised.  Tensors are torch CUDA tensors used purely as device memory; every call is enqueued on
torch's current stream.
"""
import ctypes as C
import os

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# NOF_LIB: a development override for A/B builds of the same C ABI (bundlesdf_amd/build.py:build_variant); unset in every product run
LIB_PATH = os.environ.get('NOF_LIB') or os.path.join(_HERE, 'libnof_hip.so')

NOF_MAX_LEVELS = 16
NOF_MAX_LAYERS = 8
RAY_COLS = 12
VIEW_COLS = 16
HASH_BWD_TABLE_BIG, HASH_BWD_TABLE_SMALL, HASH_BWD_INPUT, HASH_BWD_ALL, HASH_BWD_MERGE_INPUT, HASH_BWD_NEW_BATCH = 1, 2, 4, 7, 8, 16
MARCHER_WAVE, MARCHER_WALK = 0, 1


class NofError(RuntimeError):
    pass


class NofHashGrid(C.Structure):
    _fields_ = [('L', C.c_int32), ('C', C.c_int32),
                ('scale', C.c_float * NOF_MAX_LEVELS),
                ('resolution', C.c_uint32 * NOF_MAX_LEVELS),
                ('offset', C.c_uint32 * NOF_MAX_LEVELS),
                ('size', C.c_uint32 * NOF_MAX_LEVELS),
                ('hashed', C.c_uint32 * NOF_MAX_LEVELS)]


class NofFrameRaysCfg(C.Structure):
    _fields_ = [('fx', C.c_double), ('fy', C.c_double), ('cx', C.c_double), ('cy', C.c_double),
                ('near_thr', C.c_double), ('far_thr', C.c_double),
                ('box_lo', C.c_double * 3), ('box_hi', C.c_double * 3), ('pose', C.c_double * 16),
                ('frame_id', C.c_int32), ('valid_depth_only', C.c_int32)]


class NofSampleCfg(C.Structure):
    _fields_ = [('n_samples', C.c_int32), ('n_around', C.c_int32),
                ('near_sc', C.c_float), ('far_sc', C.c_float), ('trunc', C.c_float), ('neg_trunc_ratio', C.c_float),
                ('seed', C.c_uint64), ('step', C.c_uint32), ('d_step', C.c_void_p), ('deterministic', C.c_int32),
                ('marcher', C.c_int32)]


class NofPoseAccum(C.Structure):
    """nof_pose_grad_accum's arguments as a struct (include/nof_hip.h): nof_hash_encode_bwd_step carries that kernel as a passenger"""
    _fields_ = [('dpts', C.c_void_p), ('dview', C.c_void_p), ('batch', C.c_void_p), ('z_vals', C.c_void_p), ('c2w', C.c_void_p),
                ('tf', C.c_void_p), ('ff', C.c_int32), ('sh_degree', C.c_int32), ('R', C.c_int64), ('S', C.c_int32),
                ('g_ray', C.c_void_p), ('frame_slots', C.c_void_p)]


class NofAdamTail(C.Structure):
    """nof_adam_step_tail's neighbours (include/nof_hip.h): the pose gradient sums in front of Adam, the operand image and the pose
    table of the next step behind it"""
    _fields_ = [('desc', C.c_void_p), ('packed', C.c_void_p), ('mlp_off', C.c_int64), ('n_mlp', C.c_int64), ('pose_off', C.c_int64),
                ('F', C.c_int32), ('max_trans', C.c_float), ('max_rot', C.c_float), ('c2w', C.c_void_p), ('tf', C.c_void_p),
                ('frame_slots', C.c_void_p)]


class NofMarchNext(C.Structure):
    """nof_adam_step_tail_march: the NEXT batch's nof_raymarch_sample, carried by this step's optimiser launch (include/nof_hip.h)"""
    _fields_ = [('cfg', C.c_void_p), ('pool', C.c_void_p), ('ids', C.c_void_p), ('occ_bits', C.c_void_p), ('sh_degree', C.c_int32),
                ('level', C.c_int32), ('max_hits', C.c_int32), ('reserved', C.c_int32), ('R', C.c_int64), ('batch', C.c_void_p),
                ('rays_o_w', C.c_void_p), ('viewdirs_w', C.c_void_p), ('view', C.c_void_p), ('t_in_out', C.c_void_p),
                ('n_hits', C.c_void_p), ('z_vals', C.c_void_p), ('pts_w', C.c_void_p), ('valid', C.c_void_p), ('flags', C.c_void_p)]


NOF_MCL_TABLES = 47


class NofMclLuts(C.Structure):
    _fields_ = [('off', C.c_int32 * NOF_MCL_TABLES)]


class NofMlpDesc(C.Structure):
    _fields_ = [('n_sigma', C.c_int32), ('n_color', C.c_int32), ('hidden', C.c_int32), ('in_feat', C.c_int32),
                ('n_view', C.c_int32), ('geo', C.c_int32),
                ('w_off', C.c_int32 * NOF_MAX_LAYERS), ('b_off', C.c_int32 * NOF_MAX_LAYERS),
                ('out_dim', C.c_int32 * NOF_MAX_LAYERS), ('in_dim', C.c_int32 * NOF_MAX_LAYERS),
                ('n_params', C.c_int32), ('precision', C.c_int32), ('grad_scale', C.c_float)]


class NofLossCfg(C.Structure):
    _fields_ = [(k, C.c_float) for k in ('trunc', 'neg_trunc_ratio', 'sdf_lambda', 'near_sc', 'far_sc', 'rgb_weight',
                                         'fs_weight', 'trunc_weight', 'empty_weight', 'fs_sdf', 'fs_rgb_weight',
                                         'first_frame_weight', 'grad_scale')]


_P = C.c_void_p
_I64, _I32, _F = C.c_int64, C.c_int32, C.c_float
_SIGNATURES = {
    'nof_version': ([], C.c_int),
    'nof_hash_encode_fwd': ([C.POINTER(NofHashGrid), _P, _P, _P, _I64, _P], C.c_int),
    'nof_hash_encode_bwd': ([C.POINTER(NofHashGrid), _P, _P, _P, _P, _P, _I64, _P], C.c_int),
    'nof_hash_encode_bwd_levels': ([C.POINTER(NofHashGrid), _P, _P, _P, _P, _P, _I32, _I32, _I64, _P], C.c_int),
    'nof_hash_corner_indices': ([C.POINTER(NofHashGrid), _P, _P, _I64, _P], C.c_int),
    'nof_pose_fwd': ([_P, _P, _F, _F, _P, _I32, _P], C.c_int),
    'nof_pose_bwd': ([_P, _P, _F, _F, _P, _I32, _P], C.c_int),
    'nof_occgrid_build': ([_P, _I64, _I32, _I32, _P, _P], C.c_int),
    'nof_occgrid_query': ([_P, _I32, _P, _P, _I64, _P], C.c_int),
    'nof_trace_rays': ([_P, _I32, _P, _P, _I64, _I32, _P, _P, _P, _P, _P], C.c_int),
    'nof_batch_trace': ([_P, _P, _P, _P, _I32, _I32, _P, _I32, _I64, _I32, _I32, _P, _P, _P, _P, _P, _P, _P, _P, _P], C.c_int),
    'nof_sample_points': ([C.POINTER(NofSampleCfg), _P, _P, _P, _P, _I64, _I32, _P, _P, _P, _P, _P, _P, _P], C.c_int),
    'nof_mlp_packed_bytes': ([C.POINTER(NofMlpDesc)], C.c_int64),
    'nof_mlp_pack': ([C.POINTER(NofMlpDesc), _P, _P, _P], C.c_int),
    'nof_mlp_pack_pose': ([C.POINTER(NofMlpDesc), _P, _P, _P, _P, _F, _F, _P, _I32, _P], C.c_int),
    'nof_mlp_fwd': ([C.POINTER(NofMlpDesc), _P, _P, _I32, _P, _I32, _P, _P, _I64, _P], C.c_int),
    'nof_encode_mlp_fwd': ([C.POINTER(NofHashGrid), C.POINTER(NofMlpDesc), _P, _P, _P, _P, _I32, _P, _P, _P, _I64, _P], C.c_int),
    'nof_mlp_bwd_featq': ([C.POINTER(NofMlpDesc), _P, _P, _I32, _P, _I32, _P, _P, _P, _P, _P, _P, _P, _I64, _P], C.c_int),
    'nof_mlp_bwd_featq_two_launches': ([C.POINTER(NofMlpDesc), _P, _P, _I32, _P, _I32, _P, _P, _P, _P, _P, _P, _P, _I64, _P], C.c_int),
    'nof_mlp_bwd_blocks': ([], C.c_int),
    'nof_mlp_bwd': ([C.POINTER(NofMlpDesc), _P, _P, _I32, _P, _I32, _P, _P, _P, _P, _P, _P, _I64, _P], C.c_int),
    'nof_reduce_partials': ([_P, _I32, _I32, _P, _P, _P], C.c_int),
    'nof_mlp_sdf': ([C.POINTER(NofMlpDesc), _P, _P, _I32, _P, _I64, _P], C.c_int),
    'nof_composite_loss': ([C.POINTER(NofLossCfg), _P, _P, _P, _P, _I64, _I32, _P, _P, _P, _P, _P, _P], C.c_int),
    'nof_pose_grad_accum': ([_P, _P, _P, _P, _P, _P, _I32, _I32, _I64, _I32, _P, _P, _P], C.c_int),
    'nof_pose_reduce_bwd': ([_P, _P, _P, _P, _I64, _I32, _F, _F, _P, _P, _P, _I32, _I32, _P, _P], C.c_int),
    'nof_pose_reg': ([_P, _P, _I32, _F, _F, _P, _P], C.c_int),
    'nof_small_regs': ([_P, _P, _I32, _F, _F, _P], C.c_int),
    'nof_adam_step': ([_P, _P, _P, _P, _I64, _I64, _F, _F, _F, _F, _F, _I32, _P, _P], C.c_int),
    'nof_adam_step_tail': ([_P, _P, _P, _P, _I64, _I64, _F, _F, _F, _F, _F, _I32, _P, C.POINTER(NofAdamTail), _P], C.c_int),
    'nof_adam_step_tail_march': ([_P, _P, _P, _P, _I64, _I64, _F, _F, _F, _F, _F, _I32, _P, C.POINTER(NofAdamTail),
                                  C.POINTER(NofMarchNext), _P, C.c_uint32, _P], C.c_int),
    'nof_adam_step_tail_dyn': ([_P, _P, _P, _P, _I64, _I64, _P, _F, _F, _F, _I32, _F, _F, _F, _P, C.POINTER(NofAdamTail), _P, _P], C.c_int),
    'nof_grad_check': ([_P, _I64, _P, _P], C.c_int),
    'nof_render_depth': ([_P, _P, _I64, _I32, _F, _P, _P], C.c_int),
    'nof_step_state_advance': ([_P, _F, _F, _F, _I32, _F, _F, _I32, _P], C.c_int),
    'nof_adam_step_dyn': ([_P, _P, _P, _P, _I64, _I64, _P, _F, _F, _F, _P, _P], C.c_int),
    'nof_raymarch_sample': ([C.POINTER(NofSampleCfg), _P, _P, _P, _P, _I32, _I32, _P, _I32, _I64, _I32, _P, _P,
                             _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P], C.c_int),
    'nof_composite_loss_fwd_bwd': ([C.POINTER(NofLossCfg), _P, _P, _P, _P, _I64, _I32, _P, _P, _P, _P, _P, _P, _P], C.c_int),
    'nof_tile_list_bytes': ([_I64], C.c_int64),
    'nof_tile_list_build': ([_P, _I64, _I32, _P, _P], C.c_int),
    'nof_mlp_bwd_tiles': ([C.POINTER(NofMlpDesc), _P, _P, _I32, _P, _I32, _P, _P, _P, _P, _P, _P, _P, _I64, _P], C.c_int),
    'nof_hash_encode_bwd_parts': ([C.POINTER(NofHashGrid), _P, _P, _P, _P, _P, _P, _P, _I32, _I32, _P, _I32, _I32, _I64, _P], C.c_int),
    'nof_hash_encode_bwd_parts_reduce': ([C.POINTER(NofHashGrid), _P, _P, _P, _P, _P, _P, _P, _I32, _I32, _P, _I32, _I32, _I64,
                                          _P, _I32, _I32, _P, _P, _P], C.c_int),
    'nof_hash_encode_bwd_step': ([C.POINTER(NofHashGrid), _P, _P, _P, _P, _P, _P, _P, _I32, _I32, _P, _I32, _I32, _I64,
                                          _P, _I32, _I32, _P, _P, C.POINTER(NofPoseAccum), _P], C.c_int),
    'nof_mlp_bwd_workspace_bytes': ([C.POINTER(NofMlpDesc)], C.c_int64),
    'nof_sdf_grid_query': ([C.POINTER(NofHashGrid), C.POINTER(NofMlpDesc), _P, _P, _P, _I32, _P, _P, _P, _I32, _I32, _I32,
                            _F, _P, _P], C.c_int),
    'nof_mt_count': ([_P, _I32, _I32, _I32, _F, _P, _P], C.c_int),
    'nof_mt_emit': ([_P, _I32, _I32, _I32, _F, _P, _P, _P], C.c_int),
    'nof_mc_count': ([_P, _I32, _I32, _I32, _F, _P, _P, _P], C.c_int),
    'nof_mc_emit': ([_P, _I32, _I32, _I32, _F, _P, _P, _P, _P], C.c_int),
    'nof_mt_vertices': ([_P, _I32, _I32, _I32, _F, _P, _I64, _P, _P], C.c_int),
    'nof_mcl_count': ([_P, _I32, _I32, _I32, _F, _P, _P, _P, _P], C.c_int),
    'nof_mcl_emit': ([_P, _I32, _I32, _I32, _F, _P, _P, _P, _P, _P], C.c_int),
    'nof_mcl_count_blocks': ([_P, _I32, _I32, _I32, _F, _P, _P, _P, _P], C.c_int),
    'nof_mcl_emit_blocks': ([_P, _I32, _I32, _I32, _F, _P, _P, _P, _P, _P], C.c_int),
    'nof_mcl_vertices': ([_P, _I32, _I32, _I32, _F, _P, _I64, _P, _P], C.c_int),
    'nof_mask_dilate': ([_P, _I32, _I32, _I32, _P, _P, _P], C.c_int),
    'nof_frame_rays': ([C.POINTER(NofFrameRaysCfg), _P, _P, _P, _P, _P, _P, _I32, _I32, _I32, _P, _P, _P], C.c_int),
    'nof_cloud_filter': ([_P, _I64, _P, _P, _P, _I64, C.c_double, C.c_double, _P], C.c_int),
    'nof_compact_rows': ([_P, _P, _P, _I64, _P, _P], C.c_int),
    'nof_bary_uv': ([_P, _P, _P, _P, _P, _I64, _P, _P], C.c_int),
    'nof_hash_encode_bwd_eik': ([C.POINTER(NofHashGrid), _P, _P, _P, _P, _P, _P, _P, _I32, _I32, _I64, _P], C.c_int),
    'nof_eikonal': ([C.POINTER(NofMlpDesc), _P, C.POINTER(NofHashGrid), _P, _P, _P, _P, _F, _F, _P, _P, _P, _P, _I64, _P], C.c_int),
    'nof_mlp_wide_workspace_bytes': ([C.POINTER(NofMlpDesc), _I64], C.c_int64),
    'nof_mlp_wide_partial_rows': ([], C.c_int),
    'nof_mlp_wide_fwd': ([C.POINTER(NofMlpDesc), _P, _P, _I32, _P, _I32, _P, _P, _I64, _P], C.c_int),
    'nof_mlp_wide_sdf': ([C.POINTER(NofMlpDesc), _P, _P, _I32, _P, _I64, _P], C.c_int),
    'nof_mlp_wide_bwd': ([C.POINTER(NofMlpDesc), _P, _P, _I32, _P, _I32, _P, _P, _P, _P, _P, _I64, _P], C.c_int),
    'nof_mlp_wide_bwd_tiles': ([C.POINTER(NofMlpDesc), _P, _P, _I32, _P, _I32, _P, _P, _P, _P, _P, _P, _I64, _P], C.c_int),
    'nof_encode_mlp_wide_fwd': ([C.POINTER(NofHashGrid), C.POINTER(NofMlpDesc), _P, _P, _P, _P, _I32, _P, _P, _P, _I64, _P], C.c_int),
    'nof_mlp_wide_bwd_parts': ([C.POINTER(NofMlpDesc), _P, _P, _P, _I32, _P, _I32, _P, _P, _P, _P, _P, _P, _I32, _I64, _P], C.c_int),
    'nof_texture_bake_frame': ([_P, _P, _I32, _I32, _P, _P, _I64, _P, _P, _P, _F, _I32, _P, _P, _P, _P, _P], C.c_int),
}
OPTIONAL = set()
ABI_VERSION = 122           # include/nof_hip.h: NOF_ABI_VERSION

_lib = None


def exported_symbols():
    return sorted(_SIGNATURES) + ['nof_last_error']


def load():
    """dlopen libnof_hip.so; raises NofError (never falls back) when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.environ.get('NOF_LIB') and not os.environ.get('NOF_NO_REBUILD') and os.path.isdir(os.path.join(_HERE, 'csrc')):
        # a library that lags its sources (content digest, build.py) is rebuilt before it is loaded: what runs is what is in the tree.
        # Where that is not possible (no hipcc on the box, a read-only package directory, a compile error in a tree someone is
        # editing) an EXISTING library is still loaded -- with a warning that names it stale; only a missing library raises.
        from . import build as _build
        try:
            _build.build(verbose=False)
        except Exception as e:                                         # noqa: BLE001 (hipcc missing, lock file not writable, compile error)
            if not os.path.exists(LIB_PATH):
                raise
            import warnings
            warnings.warn(f'{LIB_PATH} could not be checked against / rebuilt from its sources ({type(e).__name__}: {e}); '
                          'loading the existing library, which may be STALE', RuntimeWarning)
    if not os.path.exists(LIB_PATH):
        raise NofError(f'{LIB_PATH} is missing: run `python -m bundlesdf_amd.build` (hipcc --offload-arch=gfx950). '
                       'There is no CPU fallback for the Neural Object Field hot path.')
    lib = C.CDLL(LIB_PATH)
    lib.nof_last_error.restype = C.c_char_p
    lib.nof_last_error.argtypes = []
    for name, (args, res) in _SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            if name in OPTIONAL:
                continue
            raise NofError(f'{LIB_PATH} does not export {name}; rebuild it')
        fn.argtypes = args
        fn.restype = res
    # the header this table was written against (include/nof_hip.h: NOF_ABI_VERSION): a library of another ABI would take these
    # argument lists apart differently
    if lib.nof_version() != ABI_VERSION:
        raise NofError(f'{LIB_PATH} has ABI version {lib.nof_version()}, this binding was written for {ABI_VERSION}; rebuild it')
    _lib = lib
    return lib


def _stream():
    return torch.cuda.current_stream().cuda_stream


_Tensor = torch.Tensor


def call(name, *args, stream=None):
    """One C-ABI call.  Tensors go in as their device address (contiguous CUDA tensors only), everything else as given; the
    last argument of every entry point is the HIP stream: `stream` (a raw handle) or torch's current stream.  This runs ~20 times
    per optimisation step, where the host is within 10 % of being the bottleneck (tools/host_probe.py): no per-argument wrapper
    objects, no stream lookup when the caller already knows its stream."""
    fn = _fns.get(name)
    if fn is None:
        fn = _fns[name] = getattr(load(), name)
    conv = []
    for a in args:
        if isinstance(a, _Tensor):
            if not (a.is_cuda and a.is_contiguous()):
                raise NofError(f'{name}: device pointer arguments must be contiguous CUDA tensors')
            conv.append(a.data_ptr())
        else:
            conv.append(a)
    rc = fn(*conv, torch.cuda.current_stream().cuda_stream if stream is None else stream)
    if rc != 0:
        raise NofError(f'{name} failed with code {rc}: {load().nof_last_error().decode()}')


_fns = {}


# ---------------------------------------------------------------------------------------------------
def make_hash_grid(n_levels, level_dim, base_resolution, log2_hashmap_size, desired_resolution):
    """Host-side level table: allocation per grid.py:110,127-134 (float64), indexing constants per
    gridencoder.cu:154-156 evaluated in float32.  Returns (NofHashGrid, offsets int64 [L+1], n_entries)."""
    if level_dim != 2:
        raise NofError('feature_grid_dim must be 2')
    if n_levels > NOF_MAX_LEVELS or n_levels < 2:
        raise NofError(f'num_levels must be in [2,{NOF_MAX_LEVELS}]')
    per_level_scale = np.exp2(np.log2(desired_resolution / base_resolution) / (n_levels - 1))
    offsets, offset = [], 0
    max_params = 2 ** log2_hashmap_size
    for i in range(n_levels):
        resolution = int(np.ceil(base_resolution * per_level_scale ** i))
        params_in_level = min(max_params, (resolution + 1) ** 3)
        params_in_level = int(np.ceil(params_in_level / 8) * 8)
        offsets.append(offset)
        offset += params_in_level
    offsets.append(offset)
    g = NofHashGrid()
    g.L, g.C = n_levels, level_dim
    S = np.float32(np.log2(per_level_scale))
    for l in range(n_levels):
        # exp2f(level*S) correctly rounded (float64 exp2 then one rounding == glibc exp2f; NumPy's float32 exp2 is 1 ulp off)
        e = np.float32(np.exp2(np.float64(np.float32(np.float32(l) * S))))
        scale = np.float32(e * np.float32(base_resolution) - np.float32(1.0))
        res = int(np.ceil(scale)) + 1
        size = offsets[l + 1] - offsets[l]
        stride, d = 1, 0
        while d < 3 and stride <= size:
            stride *= res + 1
            d += 1
        g.scale[l] = float(scale)
        g.resolution[l] = res
        g.offset[l] = offsets[l]
        g.size[l] = size
        g.hashed[l] = 1 if stride > size else 0
    return g, np.array(offsets, dtype=np.int64), int(offset), float(per_level_scale)


def make_mlp_desc(n_sigma, n_color, in_feat, n_view, precision=1, hidden=64, geo=15):
    """Layer table in PyTorch parameter order (NeRFSmall.parameters(): sigma_net W,b ... color_net W,b)."""
    d = NofMlpDesc()
    d.n_sigma, d.n_color, d.hidden, d.in_feat, d.n_view, d.geo = n_sigma, n_color, hidden, in_feat, n_view, geo
    d.precision = precision
    dims = []
    for l in range(n_sigma):
        dims.append((1 + geo if l == n_sigma - 1 else hidden, in_feat if l == 0 else hidden))
    for l in range(n_color):
        dims.append((3 if l == n_color - 1 else hidden, n_view + geo if l == 0 else hidden))
    off = 0
    for l, (o, i) in enumerate(dims):
        d.w_off[l] = off
        off += o * i
        d.b_off[l] = off
        off += o
        d.out_dim[l], d.in_dim[l] = o, i
    d.n_params = off
    return d, dims
