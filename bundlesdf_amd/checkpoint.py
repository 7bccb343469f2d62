"""Checkpoint interchange with the reference (nerf_runner.py:526-577).

The reference saves one state_dict per module:
    'model'          NeRFSmall: sigma_net.{2 l}.weight / .bias, color_net.{2 l}.weight / .bias  (Sequential: Linear at even
                     positions, ReLU between; nerf_helpers.py:243-294)
    'embed_fn'       GridEncoder: embeddings [rows, 2] (+ the int32 'offsets' buffer, grid.py:136-141)
    'pose_array'     PoseArray: data [F, 6];   'feature_array'  FeatureArray: data [F, ff]
    'optimizer', 'octree' (kaolin bytes), 'global_step'
Here all parameters live in one flat buffer [table | MLP in PyTorch parameter order | features | poses]; these helpers
convert between the two so that a field trained by either implementation can be loaded by the other -- including the
'optimizer' entry, which the reference's load_weights reads unconditionally (nerf_runner.py:544): it is written as the
state_dict of a torch.optim.Adam built like create_optimizer does (nerf_runner.py:492-504: group 'basic' = embeddings, NeRFSmall
parameters, feature array in that order; group 'pose_array'), with step / exp_avg / exp_avg_sq taken from the flat moment
buffers, and read back the same way.  The kaolin octree bytes are not convertible (no kaolin): the octree is rebuilt from
the point cloud.  Pure tensor shuffling, no GPU needed.
"""
import numpy as np
import torch


def mlp_state_from_flat(flat, layer_dims, n_sigma):
    """flat [n_mlp] (W0, b0, W1, b1, ... in NeRFSmall.parameters() order) -> the reference's 'model' state_dict"""
    flat = torch.as_tensor(flat, dtype=torch.float32).detach().cpu().reshape(-1)
    state, off = {}, 0
    for l, (o, i) in enumerate(layer_dims):
        net, k = ('sigma_net', l) if l < n_sigma else ('color_net', l - n_sigma)
        state[f'{net}.{2 * k}.weight'] = flat[off:off + o * i].reshape(o, i).clone()
        off += o * i
        state[f'{net}.{2 * k}.bias'] = flat[off:off + o].clone()
        off += o
    assert off == flat.numel(), (off, flat.numel())
    return state


def mlp_flat_from_state(state, layer_dims, n_sigma):
    """inverse of mlp_state_from_flat; checks every shape and that no parameter of the state_dict is left over"""
    parts, used = [], set()
    for l, (o, i) in enumerate(layer_dims):
        net, k = ('sigma_net', l) if l < n_sigma else ('color_net', l - n_sigma)
        for name, shape in ((f'{net}.{2 * k}.weight', (o, i)), (f'{net}.{2 * k}.bias', (o,))):
            if name not in state:
                raise KeyError(f"reference checkpoint has no '{name}' (expected NeRFSmall with layers {layer_dims})")
            t = torch.as_tensor(state[name], dtype=torch.float32).detach().cpu()
            if tuple(t.shape) != shape:
                raise ValueError(f"'{name}' is {tuple(t.shape)}, this field expects {shape}")
            parts.append(t.reshape(-1))
            used.add(name)
    extra = set(state) - used
    if extra:
        raise ValueError(f'unexpected parameters in the reference checkpoint: {sorted(extra)}')
    return torch.cat(parts)


def _param_shapes(field):
    """[(segment, offset in the segment, shape)] in the reference optimiser's parameter order, and the index of the first
    parameter of the 'pose_array' group (None without pose optimisation)."""
    out = [('table', 0, (field.n_entries, 2))]
    off = 0
    for o, i in field.layer_dims:
        out.append(('mlp', off, (o, i)))
        off += o * i
        out.append(('mlp', off, (o,)))
        off += o
    if field.ff > 0:
        out.append(('feat', 0, (field.F, field.ff)))
    n_basic = len(out)
    if field.optimize_poses:
        out.append(('pose', 0, (field.F, 6)))
    return out, (n_basic if field.optimize_poses else None)


def optimizer_state_dict(field, lrs=None):
    """state_dict of the torch.optim.Adam the reference builds in create_optimizer (nerf_runner.py:492-504), filled from the
    field's flat Adam moments: what its load_weights hands to optimizer.load_state_dict (nerf_runner.py:544)."""
    shapes, pose_at = _param_shapes(field)
    lr, lr_pose = lrs if lrs is not None else field.learning_rates()
    params = [torch.nn.Parameter(torch.zeros(shape)) for _, _, shape in shapes]
    n_basic = pose_at if pose_at is not None else len(params)
    groups = [{'name': 'basic', 'params': params[:n_basic], 'lr': lr}]
    if pose_at is not None:
        groups.append({'name': 'pose_array', 'params': params[n_basic:], 'lr': lr_pose})
    opt = torch.optim.Adam(groups, betas=(0.9, 0.999), weight_decay=0, eps=1e-15)
    if field.adam_steps > 0:
        for p, (seg, off, shape) in zip(params, shapes):
            n = int(np.prod(shape))
            m = field._seg(field.exp_avg, seg)[off:off + n].detach().cpu().reshape(shape).clone()
            v = field._seg(field.exp_avg_sq, seg)[off:off + n].detach().cpu().reshape(shape).clone()
            opt.state[p] = {'step': torch.tensor(float(field.adam_steps)), 'exp_avg': m, 'exp_avg_sq': v}
    return opt.state_dict()


def load_optimizer_state_dict(field, sd):
    """inverse of optimizer_state_dict: Adam moments (and the step count) of a reference checkpoint -> the flat buffers.
    Returns the step count, or None when the checkpoint's optimiser had not stepped yet."""
    shapes, _ = _param_shapes(field)
    ids = [i for g in sd['param_groups'] for i in g['params']]
    if len(ids) != len(shapes):
        raise ValueError(f"optimizer state has {len(ids)} parameters, this field has {len(shapes)}")
    step = None
    for pid, (seg, off, shape) in zip(ids, shapes):
        st = sd['state'].get(pid)
        if st is None:
            continue
        n = int(np.prod(shape))
        for key, buf in (('exp_avg', field.exp_avg), ('exp_avg_sq', field.exp_avg_sq)):
            t = torch.as_tensor(st[key], dtype=torch.float32)
            if tuple(t.shape) != tuple(shape):
                raise ValueError(f"optimizer state {key} of parameter {pid} is {tuple(t.shape)}, expected {tuple(shape)}")
            field._seg(buf, seg)[off:off + n].copy_(t.reshape(-1).to(buf.device))
        step = int(float(st['step']))
    return step


def to_reference_checkpoint(field, global_step=0):
    """dict in the reference's layout (CPU tensors) from a NeuralObjectField"""
    ck = {'global_step': int(global_step),
          'model': mlp_state_from_flat(field.mlp, field.layer_dims, field.n_sigma),
          'embed_fn': {'embeddings': field.table.detach().cpu().reshape(-1, 2).clone(),
                       'offsets': torch.as_tensor(np.asarray(field.offsets), dtype=torch.int32)},
          'embeddirs_fn': {},
          'optimizer': optimizer_state_dict(field)}
    if field.optimize_poses:
        ck['pose_array'] = {'data': field.pose.detach().cpu().reshape(field.F, 6).clone()}
    if field.ff > 0:
        ck['feature_array'] = {'data': field.feat.detach().cpu().reshape(field.F, field.ff).clone()}
    return ck


def load_reference_checkpoint(field, ck):
    """parameters of a checkpoint written by the reference's save_weights -> this field (Adam moments restart)"""
    emb = torch.as_tensor(ck['embed_fn']['embeddings'], dtype=torch.float32)
    if tuple(emb.shape) != (field.n_entries, 2):
        raise ValueError(f"embed_fn.embeddings is {tuple(emb.shape)}, this hash grid has ({field.n_entries}, 2) rows: "
                         'num_levels / base_res / finest_res / log2_hashmap_size differ')
    if 'offsets' in ck['embed_fn']:
        off = np.asarray(torch.as_tensor(ck['embed_fn']['offsets']).cpu(), dtype=np.int64)
        if not np.array_equal(off, np.asarray(field.offsets, dtype=np.int64)):
            raise ValueError('embed_fn.offsets differ from this hash grid')
    pose = feat = None
    if field.optimize_poses and 'pose_array' in ck:
        pose = torch.as_tensor(ck['pose_array']['data'], dtype=torch.float32)
        if tuple(pose.shape) != (field.F, 6):
            raise ValueError(f"pose_array.data is {tuple(pose.shape)}, expected ({field.F}, 6)")
    if field.ff > 0 and 'feature_array' in ck:
        feat = torch.as_tensor(ck['feature_array']['data'], dtype=torch.float32)
        if tuple(feat.shape) != (field.F, field.ff):
            raise ValueError(f"feature_array.data is {tuple(feat.shape)}, expected ({field.F}, {field.ff})")
    field.load_parameters(table=emb, mlp=mlp_flat_from_state(ck['model'], field.layer_dims, field.n_sigma), feat=feat, pose=pose)
    field.exp_avg.zero_()
    field.exp_avg_sq.zero_()
    field.grads.zero_()
    # Adam's own step count is the age of ITS moments (bias correction), not the runner's iteration count: zero moments restart
    # at 0 whatever global_step says (steps skipped by the reference's GradScaler make the two differ, too)
    field.adam_steps = 0
    if 'optimizer' in ck:
        step = load_optimizer_state_dict(field, ck['optimizer'])
        if step is not None:
            field.adam_steps = int(step)
    return int(ck.get('global_step', 0))
