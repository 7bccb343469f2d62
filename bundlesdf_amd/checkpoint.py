"""Checkpoint interchange with the reference (nerf_runner.py:526-577).

The reference saves one state_dict per module:
    'model'          NeRFSmall: sigma_net.{2 l}.weight / .bias, color_net.{2 l}.weight / .bias  (Sequential: Linear at even
                     positions, ReLU between; nerf_helpers.py:243-294)
    'embed_fn'       GridEncoder: embeddings [rows, 2] (+ the int32 'offsets' buffer, grid.py:136-141)
    'pose_array'     PoseArray: data [F, 6];   'feature_array'  FeatureArray: data [F, ff]
    'optimizer', 'octree' (kaolin bytes), 'global_step'
Here all parameters live in one flat buffer [table | MLP in PyTorch parameter order | features | poses]; these helpers
convert between the two so that a field trained by either implementation can be loaded by the other.  The optimiser state
and the kaolin octree bytes are not convertible (different optimiser object / no kaolin): the octree is rebuilt from the
point cloud, Adam's moments restart.  Pure tensor shuffling, no GPU needed.
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


def to_reference_checkpoint(field, global_step=0):
    """dict in the reference's layout (CPU tensors) from a NeuralObjectField"""
    ck = {'global_step': int(global_step),
          'model': mlp_state_from_flat(field.mlp, field.layer_dims, field.n_sigma),
          'embed_fn': {'embeddings': field.table.detach().cpu().reshape(-1, 2).clone(),
                       'offsets': torch.as_tensor(np.asarray(field.offsets), dtype=torch.int32)},
          'embeddirs_fn': {}}
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
    return int(ck.get('global_step', 0))
