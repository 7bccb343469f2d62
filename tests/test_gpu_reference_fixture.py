"""The HIP path against the REFERENCE's own run, with no oracle in between.

tests/golden/reference_vectors.npz holds one optimisation step of the reference's train_loop / render_rays / run_network /
raw2outputs / get_sdf_loss (nerf_runner.py:679-763, executed on CPU by tests/golden/make_golden.py with the reference's own
compiled gridencoder.cu / common.cu kernels underneath): inputs (`step_*` batch, uniforms, parameters) and what the
reference produced from them (z samples, validity, raw, weights, rgb_map, the total loss and the gradient of EVERY parameter
group).  tests/test_oracle.py::test_full_step_matches_reference_driven_run pins the oracle on the same arrays; this file feeds
them to NeuralObjectField.train_step, so HIP <-> reference is one link, not two."""
import ctypes as C

import numpy as np
import pytest
import torch

from tests import util as U
from tests.test_oracle import G, _step_cfg

pytestmark = pytest.mark.gpu


def cpu(t):
    return t.detach().cpu().numpy()


def _field(nof, precision):
    from bundlesdf_amd.field import NeuralObjectField
    cfg = _step_cfg()
    occ, c2w = G['step_occ'], G['step_c2w']
    level = int(np.log2(occ.shape[0]))
    fld = NeuralObjectField(cfg, c2w.shape[0], c2w, precision=precision, n_sigma=2, n_color=3, seed_init=False)
    fld.load_parameters(table=G['step_table'], mlp=G['step_mlp_flat'], feat=G['step_feat'], pose=G['step_pose'])
    fld.set_occupancy(U.occ_to_coords(occ), level, level)
    fld.global_step = 1                          # the fixture was taken at global_step 1 (tests/golden/make_golden.py)
    return cfg, fld


def _rel_max(got, ref):
    return float(np.abs(got - ref).max() / max(np.abs(ref).max(), 1e-30))


def _rel_l2(got, ref):
    return float(np.linalg.norm((got - ref).ravel()) / max(np.linalg.norm(ref.ravel()), 1e-30))


@pytest.mark.parametrize("precision", ['fp32', 'fp16x3', 'bf16x3'])
def test_hip_step_matches_reference_driven_fixture(nof, precision):
    cfg, fld = _field(nof, precision)
    batch = G['step_batch']
    R, S = batch.shape[0], cfg['N_samples'] + cfg['N_samples_around_depth']
    # pose corrections: PoseArray.get_matrices of the reference (nerf_helpers.py:143-154)
    fld.update_poses()
    tf = cpu(fld.tf).reshape(-1, 3, 4)
    delta_c2w = G['step_pose_mats'] @ G['step_c2w']
    assert np.abs(tf - delta_c2w[:, :3, :]).max() < 2e-6
    b = fld.train_step(U.dev(batch), None, R, U.dev(G['step_u_occ']), U.dev(G['step_u_dep']), do_step=False)
    torch.cuda.synchronize()
    assert cpu(fld.flags)[0] == 0
    # ---- sampler (common.cu:41-125 + nerf_runner.py:979-1011,1063-1081) ----
    assert np.abs(cpu(b['z_vals']) - G['step_z']).max() < 2e-5
    v_ref = G['step_valid']
    v_got = cpu(b['valid']).reshape(R, S).astype(bool)
    assert np.array_equal(v_got, v_ref)
    # ---- network outputs: north_star's 1e-3 (max-norm) on SDF / colour, at the valid AND the invalid samples
    #      (an invalid sample is the MLP of zero features, nerf_runner.py:1247,1289-1294) ----
    raw = cpu(b['raw']).reshape(R, S, 4)
    e_rgb, e_sdf = _rel_max(raw[..., :3], G['step_raw'][..., :3]), _rel_max(raw[..., 3], G['step_raw'][..., 3])
    from tests.test_gpu_ops import worst_elementwise
    w_rgb, w_sdf = worst_elementwise(raw[..., :3], G['step_raw'][..., :3]), worst_elementwise(raw[..., 3], G['step_raw'][..., 3])
    print(f'{precision}: colour {e_rgb:.2e}  sdf {e_sdf:.2e} (max-norm, vs the reference run); per element '
          f'|err| / (1e-3 |ref| + 1e-5): colour {w_rgb:.3f}  sdf {w_sdf:.3f}')
    assert e_rgb < 1e-3 and e_sdf < 1e-3
    assert w_rgb <= 1.0 and w_sdf <= 1.0, (w_rgb, w_sdf)             # north_star's 1e-3 for every single value
    # ---- compositing (raw2outputs, nerf_runner.py:1132-1169) ----
    weights = torch.empty(R, S, device='cuda')
    lc = fld._loss_cfg()
    scratch = {k: torch.empty_like(b[k]) for k in ('rgb_map', 'draw', 'loss_rows')}
    nof.call('nof_composite_loss', C.byref(lc), b['raw'], b['z_vals'], b['valid'], b['batch'], R, S, scratch['rgb_map'],
             weights, scratch['draw'], scratch['loss_rows'], None)
    torch.cuda.synchronize()
    assert np.abs(cpu(weights) - G['step_weights']).max() < 2e-5
    assert np.abs(cpu(b['rgb_map']) - G['step_rgb_map']).max() < 1e-4
    # ---- loss (train_loop, nerf_runner.py:680-752): data terms on the device + feature_reg on the host ----
    L = fld.losses()
    loss = L['loss'] + cfg['feature_reg_weight'] * float((cpu(fld.feat) ** 2).mean())
    ref_loss = float(G['step_loss'])
    assert abs(loss - ref_loss) < (2e-4 if precision == 'fp32' else 1e-3) * abs(ref_loss), (loss, ref_loss)
    # ---- gradients of every parameter group, in the reference's optimiser order ----
    n = int(G['step_n_grads'])
    ref = [G[f'step_grad_{i}'] for i in range(n)]
    got = [cpu(fld._seg(fld.grads, 'table')).reshape(-1, 2)]
    gm = cpu(fld._seg(fld.grads, 'mlp'))
    for l, (o, i) in enumerate(fld.layer_dims):
        got.append(gm[fld.desc.w_off[l]:fld.desc.w_off[l] + o * i].reshape(o, i))
        got.append(gm[fld.desc.b_off[l]:fld.desc.b_off[l] + o])
    got.append(cpu(fld._seg(fld.grads, 'feat')).reshape(-1, fld.ff))
    got.append(cpu(fld._seg(fld.grads, 'pose')).reshape(-1, 6))
    assert len(got) == n
    # fp16x3 / bf16x3: the backward runs in the plain 16-bit type like the reference's autocast (bf16: 8 mantissa bits)
    tl2, tmx = {'fp32': (3e-4, 3e-3), 'fp16x3': (6e-3, 5e-2), 'bf16x3': (3e-2, 2e-1)}[precision]
    for i, (g, r) in enumerate(zip(got, ref)):
        assert g.shape == r.shape, i
        if np.abs(r).max() == 0:
            assert np.abs(g).max() == 0, i
            continue
        assert _rel_l2(g, r) < tl2 and _rel_max(g, r) < tmx, (i, _rel_l2(g, r), _rel_max(g, r))
    assert (got[-1][0] == 0).all()                       # frame 0 is the anchor (nerf_helpers.py:152-153)


def test_hip_adam_step_from_reference_gradients(nof):
    """torch.optim.Adam over the reference's gradients == nof_adam_step over the same buffers (nerf_runner.py:492-504)."""
    cfg, fld = _field(nof, 'fp32')
    n = int(G['step_n_grads'])
    flat = np.concatenate([G[f'step_grad_{i}'].reshape(-1) for i in range(n)]).astype(np.float32)
    assert flat.size == fld.n_total
    p0 = cpu(fld.params).copy()
    fld.grads.copy_(torch.from_numpy(flat).cuda())
    fld.global_step = 0
    fld.adam_step()
    torch.cuda.synchronize()
    tp = torch.from_numpy(p0.copy()).requires_grad_(True)
    basic, pose = tp[:fld.n_basic], tp[fld.n_basic:]
    pb, pp = basic.detach().clone().requires_grad_(True), pose.detach().clone().requires_grad_(True)
    opt = torch.optim.Adam([{'params': [pb], 'lr': cfg['lrate']}, {'params': [pp], 'lr': cfg['lrate_pose']}],
                           betas=(0.9, 0.999), eps=1e-15)
    pb.grad, pp.grad = torch.from_numpy(flat[:fld.n_basic].copy()), torch.from_numpy(flat[fld.n_basic:].copy())
    opt.step()
    want = torch.cat([pb.detach(), pp.detach()]).numpy()
    assert np.abs(cpu(fld.params) - want).max() < 2e-6
