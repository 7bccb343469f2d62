"""NeuralObjectField: device-resident state of the Neural Object Field and one optimisation step through the
C ABI of libnof_hip.so.  PyTorch is used for device memory, streams and torch.distributed only.

HBM layout (all float32, one flat buffer each for params / grads / Adam m / Adam v so that the optimiser is a single
streaming pass and the data-parallel gradient exchange is a single all-reduce):

    [ hash table rows*2 | MLP (PyTorch parameter order) | frame features F*ff | pose corrections F*6 ]
      `------------------------- param group 'basic' -------------------------' `--- 'pose_array' ---'

Step (train_loop nerf_runner.py:679-763), no host synchronisation:
    pose_fwd -> mlp_pack -> raymarch_sample -> hash_fwd -> mlp_fwd -> composite_loss (+ the work list of the backward)
    -> mlp_bwd -> { hash_bwd table | reduce_partials } beside { hash_bwd input / LDS levels -> pose_grad_accum -> pose_reduce_bwd }
    -> [all-reduce] -> adam
The backward runs over a WORK LIST (NofTileList): the 32-sample tiles that hold a non-zero loss gradient, found by the loss kernel
itself and dealt evenly to the persistent waves of every backward kernel -- the same sums as the whole batch at the cost of the
tiles that have any (a third of a settled cfg2 batch).
"""
import ctypes as C
import math

import numpy as np
import torch

from . import lib
from .config import validate_cfg

# MFMA operand types.  'fp16x3' / 'bf16x3': forward kernels carry both operands as hi + lo (three MFMAs per product, fp32-class
# outputs: the mode that meets north_star's 1e-3 on SDF/colour), backward in the plain 16-bit type like the reference's
# autocast path -- with a loss scale for fp16 (NofMlpDesc.grad_scale), like its GradScaler.
PRECISIONS = {'fp32': 0, 'bf16': 1, 'fp16': 2, 'fp16x3': 3, 'bf16x3': 4}
FP16_MODES = (2, 3)


class NeuralObjectField:
    def __init__(self, cfg, n_frames, c2w, device='cuda', precision='bf16', n_sigma=2, n_color=3, seed_init=True,
                 world_size=1, rank=0, hidden=64):
        lib.load()
        self.cfg = cfg
        self.device = torch.device(device)
        self.F = int(n_frames)
        self.ff = int(cfg.get('frame_features', 0))
        validate_cfg(cfg)
        self.sh_degree = int(cfg['multires_views'])
        self.n_view = self.ff + self.sh_degree ** 2
        self.world_size, self.rank = world_size, rank
        self.grid, self.offsets, self.n_entries, self.per_level_scale = lib.make_hash_grid(
            cfg['num_levels'], cfg['feature_grid_dim'], cfg['base_res'], cfg['log2_hashmap_size'], cfg['finest_res'])
        self.L = int(cfg['num_levels'])
        self.n_sigma, self.n_color, self.hidden = n_sigma, n_color, int(hidden)
        # hidden 128 / 4 layers per network (BASELINE cfg5) run through the nof_mlp_wide_* kernels (everything on chip since round 6)
        self.wide = hidden != 64 or n_sigma > 3 or n_color > 3
        prec = PRECISIONS[precision] if isinstance(precision, str) else int(precision)
        if self.wide and prec in (3, 4):
            # the wide kernels have no hi + lo operand split (the C ABI refuses 3 / 4 for these shapes): the step runs in the plain
            # 16-bit type -- the reference's own autocast arithmetic -- and says so (rounds 3-5 downgraded silently inside the library)
            import warnings
            prec = {3: 2, 4: 1}[prec]
            warnings.warn(f"NeuralObjectField: hidden {hidden} / depths ({n_sigma},{n_color}) run without the operand split: precision "
                          f"'{precision}' -> '{ {1: 'bf16', 2: 'fp16'}[prec] }'", stacklevel=2)
        self.precision_effective = {v: k for k, v in PRECISIONS.items()}[prec]
        self.desc, self.layer_dims = lib.make_mlp_desc(n_sigma, n_color, 2 * self.L, self.n_view, prec, hidden=hidden)
        self.eikonal = float(cfg.get('eikonal_weight', 0)) > 0
        if self.eikonal:
            if self.wide:
                raise NotImplementedError('eikonal_weight > 0 with the wide networks (hidden 128 / 4 layers) is not implemented')
            # the second-order term is evaluated with the exact-fp32 MFMA whatever the training precision: its own fragment image
            self.desc32, _ = lib.make_mlp_desc(n_sigma, n_color, 2 * self.L, self.n_view, 0, hidden=hidden)
        self.optimize_poses = bool(cfg.get('optimize_poses', 1))
        self.n_table = self.n_entries * 2
        self.n_mlp = self.desc.n_params
        self.n_feat = self.F * self.ff
        self.n_pose = self.F * 6 if self.optimize_poses else 0
        self.n_basic = self.n_table + self.n_mlp + self.n_feat
        self.n_total = self.n_basic + self.n_pose
        self.max_trans = float(cfg['max_trans'] * cfg['sc_factor'])
        self.max_rot = float(cfg['max_rot'] / 180.0 * np.pi)
        dev = self.device
        # (the flat buffers carry a few floats of padding behind them, so that the sharded-optimiser exchange (GradSync mode
        # 'zero1') can reduce-scatter / all-gather equal shards in place: a multiple of 4 * world entries)
        self._pad = (-self.n_total) % (4 * max(int(world_size), 1))
        self._params_store = torch.zeros(self.n_total + self._pad, device=dev)
        self.params = self._params_store[:self.n_total]
        self._st, self._sh, self._events = None, None, {}     # the step's stream (torch object, raw handle), fork / join events
        self.pose_slots = torch.zeros(max(self.F, 1) * 16 * 28, device=dev)      # [F, NOF_POSE_SLOTS, NOF_POSE_SLOT_W], kept zero between steps
        # gradients: some headroom in FRONT of the flat buffer, so that the data-parallel step can put a copy of what lies behind
        # the table ([MLP | frame features | poses]: tens of KB) next to the coarse table levels and reduce both in one collective
        # (train_step, `bucketed`)
        self._n_tail = self.n_feat + self.n_pose
        self._head = (self._n_tail + self.n_mlp + 63) // 64 * 64
        self._grads_store = torch.zeros(self._head + self.n_total + self._pad, device=dev)
        self.grads = self._grads_store[self._head:self._head + self.n_total]
        self.exp_avg = torch.zeros(self.n_total, device=dev)
        self.exp_avg_sq = torch.zeros(self.n_total, device=dev)
        self.c2w = torch.as_tensor(np.asarray(c2w, dtype=np.float32)).reshape(self.F, 16).to(dev).contiguous()
        self.tf = torch.zeros(self.F, 12, device=dev)
        self.g_delta = torch.zeros(self.F, 12, device=dev)
        self.loss_out = torch.zeros(8, device=dev)
        self.flags = torch.zeros(4, dtype=torch.int32, device=dev)
        self.occ_bits = None
        self.level = None
        self.global_step = 0         # drives the schedules (learning rate, truncation) and the sampler's Philox step
        self.adam_steps = 0          # optimiser steps behind the Adam moments (bias correction): equal to global_step unless a
                                     # checkpoint brought parameters without moments, or moments of a different age
        self._scale_backoff = 0      # halvings of the fp16 loss scale after a reported overflow (poll_flags)
        self._bufs = {}
        self.nblk = lib.load().nof_mlp_wide_partial_rows() if self.wide else lib.load().nof_mlp_bwd_blocks()
        self.packed = torch.empty(int(lib.load().nof_mlp_packed_bytes(C.byref(self.desc))), dtype=torch.uint8, device=dev)
        if self.eikonal:
            self.packed32 = torch.empty(int(lib.load().nof_mlp_packed_bytes(C.byref(self.desc32))), dtype=torch.uint8, device=dev)
        self._packed_step = None     # optimiser step the fragment image was built for
        self.profile = None          # dict name -> [(start_event, end_event)] when per-kernel timing is on (bench.py)
        self.profile_only = None
        self.profile_also = None     # a second entry point that keeps its events beside profile_only
        self.profile_stride = 1      # events only in steps whose index is a multiple of this (an event record costs the queue ~6 us)
        # backward over the work list of non-zero tiles ('list'), over every tile through the same code path ('all': what the
        # dense-backward figure of bench.py measures), or without a list ('off': every tile, zero tiles skipped in place)
        self.backward_tiles = 'list'
        # the training forward as ONE launch with the embedding kept on chip (nof_encode_mlp_fwd; 64-wide networks, 16-bit operand
        # types); False: the two launches nof_hash_encode_fwd + nof_mlp_fwd with the fp32 [L,B,2] embedding in HBM between them
        self.fused_forward = not self.wide and self.desc.precision != 0
        # the wide networks' counterpart (round 6): encode + sigma net in one launch, colour net behind it; the backward reads the
        # operand-precision embedding `featq`.  False: nof_hash_encode_fwd + nof_mlp_wide_fwd with the fp32 embedding in HBM
        self.fused_forward_wide = self.wide
        # the large levels' scatter and dL/dx as two roles of ONE launch, the per-ray pose rows as a passenger of the LDS levels'
        # launch on the same stream: no side stream, no fork, no join (round 6).  Measured at the driver's invocation, three
        # alternating pairs on one box (profiles/r06_y_one_stream_driver.txt): eager 0.471-0.474 against 0.475-0.479 ms with the two
        # streams, settled 0.409-0.413 against 0.409-0.418, captured 0.400-0.416 against 0.437-0.449; cfg5 2.88 / 2.87 captured
        # against 2.90 / 3.05 (r06_x_one_stream.txt).  False: the two-stream tail of rounds 3-5; None: in a captured step only.
        # (The bucketed data-parallel exchange keeps the two streams: it needs the fine levels' slice early.)
        self.one_stream_backward = True
        # single-GPU step with the reference's defaults (poses optimised, no frame features, no pose regulariser): the optimiser
        # launch also sums the frames' pose gradients in front of its update and leaves the NEXT step's MFMA operand image and pose
        # table behind it (nof_adam_step_tail, round 6) -- two launches and their gaps less per step; False: the three calls
        self.fused_tail = True
        self.mlp_bwd_one_launch = True   # the 64-wide 16-bit backward's colour and sigma halves as one launch (two colour layers)
        # train_step(next_ids=...): the optimiser launch of a step also carries the NEXT batch's ray marcher (nof_adam_step_tail_march,
        # round 6) -- a latency chain per ray beside Adam's streaming; the next step then starts at its forward.  Needs fused_tail.
        # OFF: measured 0.475-0.482 against 0.455-0.459 ms at the driver's invocation -- the merged launch takes 86-88 us where
        # Adam's takes 37 and the marcher's 20, fences and wait or not (profiles/r06_ai_tail_driver.txt, r06_ak_march_x.txt): the
        # marcher's dependent loads crawl under Adam's streaming (round 3 found the same with the prologue on a side stream).
        self.march_ahead = False
        self._marched = None         # (step, ids tensor, R, pool pointer, seed) of the batch the last optimiser launch marched
        self._new_batch_pending = False
        self._tf_epoch = 0           # running target of the device counter the marcher's workgroups wait on (F more per launch)
        self._tf_epoch_dev = torch.zeros(1, dtype=torch.int32, device=dev)
        self._tail_step = None       # optimiser step whose operand image AND pose table the last nof_adam_step_tail left
        self._tail_done = torch.zeros(1040, dtype=torch.int32, device=dev)     # nof_adam_step_tail_dyn's counter (zero between launches)
        self._tail_plan = False      # inside a train_step: this step's optimiser launch will be nof_adam_step_tail
        self.graph_fork = False           # a captured step (GraphedStep) is ONE chain (True: it keeps the backward's two branches)
        self.marcher = lib.MARCHER_WAVE   # NofSampleCfg.marcher: the ray marcher of nof_raymarch_sample (lib.MARCHER_WALK: the per-lane walk)
        self.scatter_wgs_per_cu = 0                  # persistent workgroups per CU of the table scatter (0 = the library's default)
        self._state = None           # NofStepState on the device (captured-step mode): see sync_step_state / GraphedStep
        if seed_init:
            self.init_parameters()

    # ---- per-kernel timing with events on the launch stream -------------------------------------
    def _call(self, name, *args, tag=None):
        """one C-ABI call on the step's stream (`_on`; torch's current stream outside a step); with per-kernel timing on,
        bracketed by events recorded on that same stream under `tag` (default: the entry point's name)"""
        prof = self.profile
        key = tag or name
        if (prof is None or (self.profile_only is not None and key != self.profile_only and key != self.profile_also)
                or (self.profile_stride > 1 and self.global_step % self.profile_stride != 0)):
            return lib.call(name, *args, stream=self._sh)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        st = self._st if self._st is not None else torch.cuda.current_stream()
        a.record(st)
        lib.call(name, *args, stream=self._sh)
        b.record(st)
        prof.setdefault(key, []).append((a, b))

    # ---- which stream the step's launches go to --------------------------------------------------------------------------
    # The host needs ~0.45 ms to enqueue a 0.5 ms step (tools/host_probe.py), so the step does not ask torch for its current
    # stream at every launch, does not enter torch's stream context manager to launch on another one, and re-uses its fork / join
    # events: `_on(stream)` switches where `_call` launches (torch operations inside such a block still need torch's own context).
    class _On:
        __slots__ = ('f', 'st', 'prev')

        def __init__(self, f, st):
            self.f, self.st = f, st

        def __enter__(self):
            f = self.f
            self.prev = (f._st, f._sh)
            f._st, f._sh = self.st, self.st.cuda_stream

        def __exit__(self, *exc):
            self.f._st, self.f._sh = self.prev

    def _on(self, stream):
        return NeuralObjectField._On(self, stream)

    def _after(self, waiter, src, key):
        """`waiter` waits for everything enqueued on `src` so far (wait_stream with a re-used event)"""
        ev = self._events.get(key)
        if ev is None:
            ev = self._events[key] = torch.cuda.Event()
        ev.record(src)
        waiter.wait_event(ev)

    def kernel_times_ms(self, stat='mean', skip=0):
        """launch duration per timed call (ms; `stat`: 'mean' or 'median' over the recorded launches, the first `skip` of each
        left out: a kernel's first launch includes loading its code object -- a call that was made no more than `skip` times is
        not a per-step launch and is left out altogether, e.g. the packing launch of a run's first step when every later step's
        operand image comes from the optimiser launch before it); the events are recorded on the stream the launch goes to."""
        if not self.profile:
            return {}
        torch.cuda.synchronize()
        f = np.median if stat == 'median' else np.mean
        out = {}
        for k, v in self.profile.items():
            t = [a.elapsed_time(b) for a, b in v]
            t = t[skip:] if skip else t
            if t:
                out[k] = float(f(t))
        return out

    # ---- views into the flat buffers ------------------------------------------------------------
    def _seg(self, buf, which):
        a = {'table': 0, 'mlp': self.n_table, 'feat': self.n_table + self.n_mlp, 'pose': self.n_basic}[which]
        n = {'table': self.n_table, 'mlp': self.n_mlp, 'feat': self.n_feat, 'pose': self.n_pose}[which]
        return buf[a:a + n]

    table = property(lambda s: s._seg(s.params, 'table'))
    mlp = property(lambda s: s._seg(s.params, 'mlp'))
    feat = property(lambda s: s._seg(s.params, 'feat'))
    pose = property(lambda s: s._seg(s.params, 'pose'))

    def init_parameters(self):
        """Same initialisers, in the same order, on the torch CPU generator as create_nerf (nerf_runner.py:204-242):
        GridEncoder U(-1e-4,1e-4) (grid.py:146-148), NeRFSmall's nn.Linear defaults with the SDF bias at 0.1
        (nerf_helpers.py:267-272,290), FeatureArray N(0,1) (:119), PoseArray zeros (:139)."""
        table = torch.empty(self.n_entries, 2).uniform_(-1e-4, 1e-4)
        chunks = []
        for l, (o, i) in enumerate(self.layer_dims):
            lin = torch.nn.Linear(i, o, bias=True)
            if l == self.n_sigma - 1:
                torch.nn.init.constant_(lin.bias, 0.1)
            chunks += [lin.weight.detach().reshape(-1), lin.bias.detach().reshape(-1)]
        # NOTE: nn.init.constant_ consumes no random numbers, so doing it inline keeps the stream identical.
        mlp = torch.cat(chunks)
        self.load_parameters(table=table, mlp=mlp,
                             feat=torch.normal(0, 1, size=[self.F, self.ff]).float() if self.ff > 0 else None,
                             pose=torch.zeros(self.F, 6) if self.optimize_poses else None)

    def load_parameters(self, table=None, mlp=None, feat=None, pose=None):
        self._packed_step = None
        for name, val in (('table', table), ('mlp', mlp), ('feat', feat), ('pose', pose)):
            if val is not None:
                seg = self._seg(self.params, name)
                seg.copy_(torch.as_tensor(val, dtype=torch.float32).reshape(-1).to(self.device))

    def mlp_state(self):
        """[(W [out,in], b [out])] views on the host (PyTorch state_dict order)."""
        flat = self.mlp.detach().cpu()
        out = []
        for l, (o, i) in enumerate(self.layer_dims):
            W = flat[self.desc.w_off[l]:self.desc.w_off[l] + o * i].reshape(o, i)
            b = flat[self.desc.b_off[l]:self.desc.b_off[l] + o]
            out.append((W, b))
        return out

    # ---- occupancy ---------------------------------------------------------------------------------
    def set_occupancy(self, coords_max_level, max_level, level):
        """coords [P,3] int occupied cells at max_level (dilated) -> bitfield of the ray-tracing level."""
        n = 1 << level
        self._marched = None
        self.level, self.max_level = int(level), int(max_level)
        self.occ_bits = torch.zeros((n ** 3 + 31) // 32, dtype=torch.int32, device=self.device)
        coords = torch.as_tensor(np.ascontiguousarray(coords_max_level, dtype=np.int32)).to(self.device)
        lib.call('nof_occgrid_build', coords, coords.shape[0], max_level, level, self.occ_bits)
        self.max_hits = 3 * n + 2

    def trace(self, rays_o, rays_d, want_cells=False):
        R = rays_o.shape[0]
        tio = torch.empty(R, self.max_hits, 2, device=self.device)
        nh = torch.empty(R, dtype=torch.int32, device=self.device)
        cid = torch.empty(R, self.max_hits, dtype=torch.int32, device=self.device) if want_cells else None
        lib.call('nof_trace_rays', self.occ_bits, self.level, rays_o.contiguous(), rays_d.contiguous(), R, self.max_hits,
                 tio, cid, nh, self.flags)
        return tio, cid, nh

    # ---- schedules ---------------------------------------------------------------------------------
    def truncation(self):
        """get_truncation (nerf_runner.py:663-676)."""
        cfg = self.cfg
        kind = cfg.get('trunc_decay_type', '')
        if kind == 'linear':
            t = cfg['trunc_start'] - (cfg['trunc_start'] - cfg['trunc']) * float(self.global_step) / cfg['n_step']
        elif kind == 'exp':
            lamb = np.log(cfg['trunc'] / cfg['trunc_start']) / (cfg['n_step'] / 4)
            t = max(cfg['trunc_start'] * np.exp(self.global_step * lamb), cfg['trunc'])
        else:
            t = cfg['trunc']
        return float(t * cfg['sc_factor'])

    def learning_rates(self):
        """schedule_lr is applied after steps g with g % 10 == 0, g > 0 (nerf_runner.py:762-763,579-583); so the step
        with index s uses the rate set at the last such g < s."""
        s = self.global_step
        g = 0 if s <= 10 else ((s - 1) // 10) * 10
        k = 1.0 if g == 0 else self.cfg['decay_rate'] ** (float(g) / (self.cfg['n_step'] + 1))   # N_iters, nerf_runner.py:160
        return self.cfg['lrate'] * k, self.cfg['lrate_pose'] * k

    # ---- buffers --------------------------------------------------------------------------------------
    def _buffers(self, R, S):
        key = (R, S)
        if key not in self._bufs:
            d, B = self.device, R * S
            e = lambda *s, dt=torch.float32: torch.empty(*s, dtype=dt, device=d)
            self._bufs[key] = dict(
                batch=e(R, 12), rays_o_w=e(R, 3), viewdirs_w=e(R, 3), view=e(R, 16), t_in_out=e(R, self.max_hits, 2),
                n_hits=e(R, dt=torch.int32), z_vals=e(R, S), pts_w=e(B, 3), valid=e(B, dt=torch.uint8),
                feat=None, featq=None,                  # the embedding: fp32 [L,B,2], or the fused forward's operand-precision copy [B,32]
                raw=e(B, 4), draw=e(B, 4), dfeat=e(self.L, B, 2), dview=torch.zeros(R, 16, device=d), dpts=e(B, 3),
                rgb_map=e(R, 3), partials=e(self.nblk, self.n_mlp), loss_rows=e(R, 8), g_ray=e(R, 12),
                # sigma-head output / its gradient in MFMA operand precision: the hand-off of the split MLP backward
                sig=e(B, 16, dt=torch.int16) if self.desc.precision != 0 and not self.wide else None,
                dsig=e(B, 16, dt=torch.int16) if self.desc.precision != 0 and not self.wide else None,
                wide_ws=e(int(lib.load().nof_mlp_wide_workspace_bytes(C.byref(self.desc), B)), dt=torch.uint8) if self.wide else None,
                tiles=e(int(lib.load().nof_tile_list_bytes(B)), dt=torch.uint8),
                # eikonal option: d sdf / d feature, dE/dn, and the sigma layers' weight-gradient rows (colour entries stay zero)
                geik=e(self.L, B, 2) if self.eikonal else None, dedn=e(B, 3) if self.eikonal else None,
                partials_e=torch.zeros(self.nblk, self.n_mlp, device=d) if self.eikonal else None)
        return self._bufs[key]

    def _side_stream(self):
        if getattr(self, '_side', None) is None:
            self._side = torch.cuda.Stream(device=self.device)
        return self._side

    def _aux_stream(self):
        if getattr(self, '_aux', None) is None:
            self._aux = torch.cuda.Stream(device=self.device)
        return self._aux

    def _sample_cfg(self, seed, step, dyn=False, deterministic=False):
        cfg = self.cfg
        return lib.NofSampleCfg(cfg['N_samples'], cfg['N_samples_around_depth'], cfg['near'] * cfg['sc_factor'],
                                cfg['far'] * cfg['sc_factor'], self.truncation(), cfg['neg_trunc_ratio'], seed, step,
                                self._state.data_ptr() if dyn else None, 1 if deterministic else 0, self.marcher)

    # ---- device-resident step state (NofStepState): what makes a captured step replayable ----------------------
    def sync_step_state(self):
        """device state := this field's global_step (and the Adam / schedule constants of the next optimiser step)"""
        if self._state is None:
            self._state = torch.zeros(4, dtype=torch.int32, device=self.device)
        cfg = self.cfg
        lib.call('nof_step_state_advance', self._state, C.c_float(cfg['lrate']), C.c_float(cfg['lrate_pose']),
                 C.c_float(cfg['decay_rate']), int(cfg['n_step']) + 1, C.c_float(0.9), C.c_float(0.999), int(self.global_step))

    def _set_grad_scale(self, B):
        """fp16 operands only: loss scale of the MLP backward = the power of two nearest below B/16, at most 2^16 (the
        reference's GradScaler starts at 2^16, nerf_runner.py:159).  Every loss term is a mean over R*S = B samples, so the
        loss gradient is O(weight / B): the scale brings it to O(weight / 16), far inside binary16's normal range and
        three orders of magnitude below its maximum.  Applied and removed inside nof_mlp_bwd; exact (power of two)."""
        if self.desc.precision in FP16_MODES:
            e = int(min(16, max(0, math.floor(math.log2(max(B, 16) / 16.0))))) - self._scale_backoff
            self.desc.grad_scale = float(2 ** max(e, 0))

    def poll_flags(self):
        """Host side of the device flags (ONE sync; the runner calls it at its print interval and at the end of training, never
        inside a step).  A non-finite weight gradient out of the 16-bit backward raises bit 2 (value 4) for the step it happened
        in: that step's Adam launches see it and skip the update on the device, like the reference's GradScaler.step
        (nerf_runner.py:756-761), and the next batch's sampler launch turns it into the sticky bit 3 (value 8).  Here -- the other
        half of GradScaler.update -- the loss scale of the following steps is halved and the sticky bit cleared.  Returned: the
        flags as found, with a skipped step reported as 4; the other bits (1 = a ray exceeded max_hits, 2 = inconsistent sample
        walk) as they are."""
        v = int(self.flags[0].item())
        if v & 12:
            self._scale_backoff += 1
            self.flags[0] = v & ~12                                  # (.item() synchronised: no step is in flight)
            v = (v & ~12) | 4
        return v

    def _loss_cfg(self):
        cfg = self.cfg
        return lib.NofLossCfg(self.truncation(), cfg['neg_trunc_ratio'], cfg['sdf_lambda'], cfg['near'] * cfg['sc_factor'],
                              cfg['far'] * cfg['sc_factor'], cfg['rgb_weight'], cfg['fs_weight'], cfg['trunc_weight'],
                              cfg['empty_weight'], cfg['fs_sdf'], cfg.get('fs_rgb_weight', 0), cfg['first_frame_weight'],
                              1.0 / self.world_size)

    # ---- forward pieces -----------------------------------------------------------------------------------
    def pack_weights(self, force=False):
        """fp32 PyTorch-layout MLP parameters -> MFMA fragment image (once per optimiser step; `force`: a captured step must
        contain the launch whatever the host-side cache says)."""
        if force or self._packed_step != self.global_step:
            self._call('nof_mlp_pack', C.byref(self.desc), self.mlp, self.packed)
            if self.eikonal:
                self._call('nof_mlp_pack', C.byref(self.desc32), self.mlp, self.packed32)
            self._packed_step = self.global_step

    def update_poses(self):
        self._call('nof_pose_fwd', self.pose if self.optimize_poses else None, self.c2w, C.c_float(self.max_trans),
                 C.c_float(self.max_rot), self.tf, self.F)

    def _prologue(self, b, pool, ids, R, u_occ, u_dep, seed, want_cells, dyn, deterministic=False):
        """What a step needs before it touches the hash table: the MFMA fragment image of the MLPs (5 us on the main stream:
        packing it on the side stream cost more in the cross-stream hand-over than the overlap returned), the pose corrections,
        occupancy ray marching + stratified sampling + sample points of the batch (nerf_runner.py:1044-1060 via :918-1000).
        (Running the NEXT batch's prologue at the end of a step, on the side stream beside Adam -- it needs only the few KB of
        poses / features / MLPs, updated first -- was built and measured: 0.526 vs 0.515 ms/step.  The fork and the join cost what
        the overlap returns, and the ray marcher's dependent loads slow down under Adam's streaming.)"""
        fresh = self._packed_step == self.global_step == self._tail_step   # (nothing replaced the parameters since the last optimiser launch)
        if (dyn and not self._tail_plan) or self._packed_step != self.global_step:
            # pose table + MFMA fragment image in ONE launch (two 6-microsecond kernels before: the pose update is one short
            # dependent chain per frame and rides as one extra workgroup of the packing launch)
            self._call('nof_mlp_pack_pose', C.byref(self.desc), self.mlp, self.packed, self.pose if self.optimize_poses else None,
                       self.c2w, C.c_float(self.max_trans), C.c_float(self.max_rot), self.tf, self.F)
            if self.eikonal:
                self._call('nof_mlp_pack', C.byref(self.desc32), self.mlp, self.packed32)
            self._packed_step = self.global_step
        elif self._tail_step != self.global_step:      # (the last optimiser launch left this step's pose table as well: fused_tail)
            self.update_poses()
        cid = None
        if want_cells:
            cid = b.setdefault('cell_ids', torch.empty(R, self.max_hits, dtype=torch.int32, device=self.device))
        mk, self._marched = self._marched, None
        if (mk is not None and self._tail_plan and not dyn and not deterministic and not want_cells and u_occ is None and u_dep is None
                and mk[0] == self.global_step and ids is not None and mk[1].data_ptr() == ids.data_ptr() and mk[2] == R == ids.shape[0]
                and mk[3] == pool.data_ptr() and mk[4] == seed and fresh):
            # the previous step's optimiser launch marched this batch (nof_adam_step_tail_march); what the marcher does for the
            # device flags when a batch starts rides in this step's merged scatter launch instead (HASH_BWD_NEW_BATCH)
            self._new_batch_pending = True
            return
        sc = self._sample_cfg(seed, self.global_step, dyn, deterministic)
        self._call('nof_raymarch_sample', C.byref(sc), pool, ids, self.tf, self.feat if self.ff > 0 else None, self.ff,
                   self.sh_degree, self.occ_bits, self.level, R, self.max_hits, u_occ, u_dep, b['batch'], b['rays_o_w'],
                   b['viewdirs_w'], b['view'], b['t_in_out'], cid, b['n_hits'], b['z_vals'], b['pts_w'], b['valid'],
                   self.flags)

    def forward_batch(self, pool, ids, R, u_occ=None, u_dep=None, seed=0, want_cells=False, dyn=False, deterministic=False):
        """render_rays up to raw (nerf_runner.py:1044-1088); returns the buffer dict.  `deterministic`: perturb=False."""
        cfg = self.cfg
        S = cfg['N_samples'] + cfg['N_samples_around_depth']
        b = self._buffers(R, S)
        self._prologue(b, pool, ids, R, u_occ, u_dep, seed, want_cells, dyn, deterministic)
        B = R * S
        if self.fused_forward:
            if b['featq'] is None:
                b['featq'] = torch.empty(B, 32, dtype=torch.int16, device=self.device)
            self._call('nof_encode_mlp_fwd', C.byref(self.grid), C.byref(self.desc), self.packed, self.table, b['pts_w'], b['view'], S,
                       b['raw'], b['sig'], b['featq'], B)
            return b, S
        if self.wide and self.fused_forward_wide:
            if b['featq'] is None:
                b['featq'] = torch.empty(B, 32, dtype=torch.int16, device=self.device)
            self._call('nof_encode_mlp_wide_fwd', C.byref(self.grid), C.byref(self.desc), self.packed, self.table, b['pts_w'], b['view'], S,
                       b['raw'], b['wide_ws'], b['featq'], B)
            return b, S
        if b['feat'] is None:
            b['feat'] = torch.empty(self.L, B, 2, device=self.device)
        self._call('nof_hash_encode_fwd', C.byref(self.grid), b['pts_w'], self.table, b['feat'], B)
        if self.wide:
            self._call('nof_mlp_wide_fwd', C.byref(self.desc), self.packed, b['feat'], self.L, b['view'], S, b['raw'], b['wide_ws'], B)
        else:
            self._call('nof_mlp_fwd', C.byref(self.desc), self.packed, b['feat'], self.L, b['view'], S, b['raw'], b['sig'], B)
        return b, S

    def train_step(self, pool, ids, R, u_occ=None, u_dep=None, seed=0, do_step=True, want_cells=False,
                   grad_sync=None, dyn=False, next_ids=None):
        """next_ids (optional): the ids of the batch the NEXT train_step call will be given (same pool, R and seed): where it can,
        this step's optimiser launch marches that batch's rays beside Adam and the next call starts at its forward (march_ahead).
        The step's buffer dict is then overwritten with the next batch's rays / samples by the time this call's launches end."""
        with self._on(torch.cuda.current_stream()):
            return self._train_step(pool, ids, R, u_occ, u_dep, seed, do_step, want_cells, grad_sync, dyn, next_ids)

    def _train_step(self, pool, ids, R, u_occ, u_dep, seed, do_step, want_cells, grad_sync, dyn, next_ids=None):
        """One train_loop iteration (nerf_runner.py:679-763).  `grad_sync(flat_grads)` is the data-parallel hook
        (RCCL all-reduce); gradients are already scaled by 1/world_size.  dyn=True: the per-step scalars (Philox step, Adam
        step sizes) are read from the device-resident NofStepState instead of being passed by value, and the state is advanced
        by the step's last launch -- the form GraphedStep captures."""
        cfg = self.cfg
        # (decided before the prologue: a captured step whose optimiser launch leaves the next operand image packs none itself)
        one_stream = self.one_stream_backward if self.one_stream_backward is not None else (dyn and not self.graph_fork)
        tail = self._tail_plan = bool(one_stream and do_step and self._tail_ok(grad_sync))
        self._new_batch_pending = False
        try:
            b, S = self.forward_batch(pool, ids, R, u_occ, u_dep, seed, want_cells, dyn)
        finally:
            self._tail_plan = False
        new_batch = self._new_batch_pending        # (this batch was marched by the previous optimiser launch: see _prologue)
        self._new_batch_pending = False
        B = R * S
        lc = self._loss_cfg()
        # the work list of the backward; the eikonal term has a gradient at every sample, so it takes none
        tiles = b['tiles'] if self.backward_tiles != 'off' and not self.eikonal else None
        self._call('nof_composite_loss_fwd_bwd', C.byref(lc), b['raw'], b['z_vals'], b['valid'], b['batch'], R, S, b['rgb_map'],
                   None, b['draw'], b['loss_rows'], self.loss_out, tiles)
        if tiles is not None and self.backward_tiles == 'all':
            self._call('nof_tile_list_build', None, B, 1, tiles)
        self._set_grad_scale(B)                    # (dview is zero: allocated so, and re-zeroed after its last use in every step)
        # wide networks (hidden 128 / 4 layers): one kernel per network, colour then sigma, forward recomputed and the weight
        # gradient accumulated on chip (round 6; rounds 3-5: two data kernels + eight weight-gradient passes on a third stream)
        if self.wide:
            fused = self.fused_forward_wide                               # (the embedding the forward of THIS batch left: featq or feat)
            self._call('nof_mlp_wide_bwd_parts', C.byref(self.desc), self.packed, None if fused else b['feat'],
                       b['featq'] if fused else None, self.L, b['view'], S, b['draw'],
                       b['wide_ws'], b['dfeat'], b['dview'], b['partials'], tiles, 3, B, tag='nof_mlp_wide_bwd')
        elif self.fused_forward:
            # (both networks' halves in ONE launch where their workgroup shapes agree, round 6; mlp_bwd_one_launch = False: two)
            self._call('nof_mlp_bwd_featq' if self.mlp_bwd_one_launch else 'nof_mlp_bwd_featq_two_launches', C.byref(self.desc), self.packed, b['featq'], self.L, b['view'], S, b['draw'], b['sig'],
                       b['dsig'], b['dfeat'], b['dview'], b['partials'], tiles, B, tag='nof_mlp_bwd_tiles')
        else:
            self._call('nof_mlp_bwd_tiles', C.byref(self.desc), self.packed, b['feat'], self.L, b['view'], S, b['draw'], b['sig'],
                       b['dsig'], b['dfeat'], b['dview'], b['partials'], tiles, B)
        geik = dedn = None
        if self.eikonal:
            # nerf_runner.py:734-738: mean over the samples with sdf < 1 (their number stays on the device)
            n_sel = (b['raw'][:, 3] < 1).sum().float().reshape(1)
            self._call('nof_eikonal', C.byref(self.desc32), self.packed32, C.byref(self.grid), self.table, b['pts_w'], b['valid'],
                       n_sel, C.c_float(cfg['eikonal_weight']), C.c_float(1.0 / self.world_size), b['geik'], b['dedn'],
                       b['partials_e'], self.loss_out, B)
            geik, dedn = b['geik'], b['dedn']
        dpts = b['dpts'] if self.optimize_poses else None
        gtab = self._seg(self.grads, 'table')
        hashed = [l for l in range(self.L) if self.grid.hashed[l]]
        split = hashed[0] if hashed and 0 < hashed[0] < self.L else None
        # (a captured step is one chain on one stream: no bucketed exchange inside it)
        bucketed = (grad_sync is not None and hasattr(grad_sync, 'start') and split is not None and not dyn
                    and getattr(grad_sync, 'mode', 'allreduce') == 'allreduce')
        BIG, SMALL, INPUT, ALL = lib.HASH_BWD_TABLE_BIG, lib.HASH_BWD_TABLE_SMALL, lib.HASH_BWD_INPUT, lib.HASH_BWD_ALL
        adam_done = None                               # flat entries [adam_done) that have had their Adam update already

        def reduce_mlp():
            self._call('nof_reduce_partials', b['partials'], self.nblk, self.n_mlp, self._seg(self.grads, 'mlp'), self.flags)
            if self.eikonal:
                self._call('nof_reduce_partials', b['partials_e'], self.nblk, self.n_mlp, self._seg(self.grads, 'mlp'), self.flags)

        def hash_bwd(parts, lo, hi, with_reduce=False):
            """the kernels `parts` of the hash backward for the table levels [lo, hi), on the current stream (the library owns no
            stream: what runs beside what is decided here).  with_reduce: followed by the MLP backward's row reduction, which then
            rides inside the launch of the LDS-accumulated levels instead of being the step's next launch."""
            if not self.optimize_poses:
                parts &= ~INPUT
            if parts:
                tag = 'hash_bwd[' + '+'.join(n for n, m in (('table', BIG), ('table_lds', SMALL), ('input', INPUT)) if parts & m) + ']'
                if with_reduce:
                    self._call('nof_hash_encode_bwd_parts_reduce', C.byref(self.grid), b['pts_w'], self.table, b['dfeat'], geik, dedn,
                               gtab, dpts, lo, hi, tiles, parts, self.scatter_wgs_per_cu, B, b['partials'], self.nblk, self.n_mlp,
                               self._seg(self.grads, 'mlp'), self.flags, tag=tag)
                    return
                self._call('nof_hash_encode_bwd_parts', C.byref(self.grid), b['pts_w'], self.table, b['dfeat'], geik, dedn, gtab,
                           dpts, lo, hi, tiles, parts, self.scatter_wgs_per_cu, B, tag=tag)
            elif with_reduce:
                reduce_mlp()

        def pose_kernels():
            if self.optimize_poses:
                # per-ray rows, added on the fly to 16 partial sums per frame (one atomic instruction per ray; dview rows zeroed
                # for the next step's atomics); the per-frame kernel adds a frame's partial sums instead of searching the batch
                # for the frame's rays (25 -> 8 us on the step's second chain)
                self._call('nof_pose_grad_accum', b['dpts'], b['dview'], b['batch'], b['z_vals'], self.c2w, self.tf, self.ff,
                           self.sh_degree, R, S, b['g_ray'], self.pose_slots)
                self._call('nof_pose_reduce_bwd', self.pose, None, None, None, R, self.ff, C.c_float(self.max_trans),
                           C.c_float(self.max_rot), self._seg(self.grads, 'pose'),
                           self._seg(self.grads, 'feat') if self.ff > 0 else None, None, self.F, 0, self.pose_slots)
            elif self.ff > 0:
                self._call('nof_pose_reduce_bwd', None, None, b['dview'], b['batch'], R, self.ff,
                           C.c_float(self.max_trans), C.c_float(self.max_rot), None, self._seg(self.grads, 'feat'), None, self.F, 1,
                           None)                                       # (zeroes dview for the next step's atomics)
            else:
                with torch.cuda.stream(self._st):
                    b['dview'].zero_()

        one_stream = one_stream and not bucketed and self.optimize_poses and not self.eikonal
        assert one_stream or not tail
        # (tail: the optimiser launch takes the per-frame sums and the next step's prologue along, see fused_tail)
        if one_stream:
            # ONE chain of three launches: { large levels' scatter | dL/dx } as roles of one launch (NOF_HASH_BWD_MERGE_INPUT),
            # { LDS level | MLP row reduction | per-ray pose rows } as roles of the next (nof_hash_encode_bwd_step), the per-frame sums
            pa = lib.NofPoseAccum(b['dpts'].data_ptr(), b['dview'].data_ptr(), b['batch'].data_ptr(), b['z_vals'].data_ptr(),
                                  self.c2w.data_ptr(), self.tf.data_ptr(), self.ff, self.sh_degree, R, S, b['g_ray'].data_ptr(),
                                  self.pose_slots.data_ptr())
            self._call('nof_hash_encode_bwd_step', C.byref(self.grid), b['pts_w'], self.table, b['dfeat'], None, None, gtab, dpts,
                       0, self.L, tiles, ALL | lib.HASH_BWD_MERGE_INPUT | (lib.HASH_BWD_NEW_BATCH if new_batch else 0),
                       self.scatter_wgs_per_cu, B, b['partials'], self.nblk,
                       self.n_mlp, self._seg(self.grads, 'mlp'), self.flags, C.byref(pa), tag='hash_bwd[table+table_lds]')
            if not tail:
                self._call('nof_pose_reduce_bwd', self.pose, None, None, None, R, self.ff, C.c_float(self.max_trans),
                           C.c_float(self.max_rot), self._seg(self.grads, 'pose'),
                           self._seg(self.grads, 'feat') if self.ff > 0 else None, None, self.F, 0, self.pose_slots)
        elif dyn and not self.graph_fork:
            # captured step as ONE chain
            reduce_mlp()
            hash_bwd(ALL, 0, self.L)
            pose_kernels()
        else:
            # (dyn and graph_fork: the capture follows the fork / join events below, so the captured step keeps the two branches)
            # The table scatter of the large levels (atomics that execute memory-side) is the longest launch of the backward; the
            # input gradient and the pose / frame-feature gradients that hang off it are independent of it and run beside it on a
            # second stream (fork / join by events); the LDS-accumulated small levels and the MLP row reduction follow it.
            main = self._st
            side = self._side_stream()
            self._after(side, main, 'fork')
            if bucketed:
                # data parallel: the fine (hashed) levels first; their slice [rows of level `split` .., MLP] of the flat gradient
                # buffer (80 % of its bytes at cfg2) is all-reduced while the coarse levels, dL/dx and the pose kernels still run.
                # With a bf16 payload (GradSync.payload) that slice is the table rows alone, rounded to bfloat16 on their way out,
                # and the MLP rows travel in fp32 with the trailing collective.
                a = 2 * int(self.offsets[split])
                compressed = getattr(grad_sync, 'payload', 'fp32') == 'bf16'
                first_hi = self.n_table if compressed else self.n_table + self.n_mlp
                reduce_mlp()
                hash_bwd(BIG | SMALL, split, self.L)
                grad_sync.start(self.grads[a:first_hi], compressed=compressed)
                with self._on(side):
                    hash_bwd(INPUT, 0, self.L)
                    pose_kernels()
                hash_bwd(BIG | SMALL, 0, split)
            else:
                # measured chains at cfg2 over the work list: { table scatter 95 us (beside dL/dx), LDS levels 30, row reduction 10 }
                # | { dL/dx 125 us (beside the scatter), pose kernels 35 }
                with self._on(side):
                    hash_bwd(INPUT, 0, self.L)
                    pose_kernels()
                # (the LDS levels on a third stream beside both: no gain; the two chains swapped between the streams: 2 % slower; the
                # MLP row reduction at the HEAD of the second chain: settled 0.417-0.422 vs 0.411)
                if not self.eikonal:
                    hash_bwd(BIG | SMALL, 0, self.L, with_reduce=True)
                else:
                    hash_bwd(BIG | SMALL, 0, self.L)
                    reduce_mlp()
                # (Adam is element-wise and could start per range as soon as a range's gradient is final -- the table's share right
                # here, the rest at the end of the side stream.  Measured: 0.518 vs 0.521 ms at cfg2 (noise), 4.9-5.0 vs 4.7-4.8 ms
                # at cfg5, where it takes HBM bandwidth from the weight-gradient passes that are the critical path: not done)
            self._after(main, side, 'join')
        if self.optimize_poses and float(cfg.get('pose_reg_weight', 0)) > 0:
            self._call('nof_pose_reg', self.pose, self._seg(self.grads, 'pose'), self.F, C.c_float(cfg['pose_reg_weight']),
                       C.c_float(1.0 / self.world_size), self.loss_out)
        if self.ff > 0:
            self._call('nof_small_regs', self.feat, self._seg(self.grads, 'feat'), self.n_feat,
                       C.c_float(cfg['feature_reg_weight']), C.c_float(1.0 / self.world_size))
        zero1 = grad_sync is not None and getattr(grad_sync, 'mode', '') == 'zero1'
        if zero1:
            # sharded optimiser: reduce-scatter the flat gradient, Adam on this rank's 1/world of the flat buffers,
            # all-gather the parameters.  Every rank ends the step with the same parameter bits (they all receive every shard);
            # the Adam moments of a rank are current on its own shard only (gather_optimizer_state before a checkpoint).
            from .dist import GradSync
            padded_g = self._grads_store[self._head:]
            n_pad, shard, lo, hi = GradSync.shard_range(self.n_total)
            assert padded_g.numel() == n_pad == self._params_store.numel(), 'field built for another world size'
            # a non-finite partial sum on any rank: every rank skips (its own shard) -- the skip bit is agreed on first, in every
            # precision (nof_reduce_partials raises it whatever the operand type)
            self._call('nof_grad_check', self._seg(self.grads, 'mlp'), self.n_mlp, self.flags)
            grad_sync.max_flags_(self.flags)
            grad_sync.reduce_scatter_(padded_g)
            if do_step:
                if hi > lo:
                    self.adam_step(dyn, lo, hi, advance=False)                  # (zeroes the gradient of [lo, hi))
                with torch.cuda.stream(self._st):
                    padded_g[:lo].zero_()
                    padded_g[hi:].zero_()
                grad_sync.all_gather_(self._params_store)
                self.global_step += 1
                self.adam_steps += 1
                self._zero1_shard = (lo, hi)
            grad_sync.end_step()
            return b
        if grad_sync is not None and getattr(grad_sync, 'mode', '') == 'rows':
            # touched-row exchange (dist.GradSync mode 'rows'): the table as the union of the ranks' non-zero rows, the rest dense
            with torch.cuda.stream(self._st):
                grad_sync.exchange_rows_(self.grads[:self.n_table], 2)
                grad_sync.exchange_dense_(self.grads[self.n_table:])
            grad_sync.end_step()
        elif bucketed:
            # ONE more collective: everything that is not in flight yet.  [0, a) and what lies behind the first slice are not
            # contiguous in the flat buffer, so a copy of the latter (tens of KB) rides in the headroom in front of it
            h = self._head
            rest = self.grads[first_hi:]
            nr = rest.numel()
            if nr:
                self._grads_store[h - nr:h].copy_(rest)
            grad_sync.start(self._grads_store[h - nr:h + a])
            if do_step and not dyn and hasattr(grad_sync, 'finish_first') and not compressed:
                # Adam is element-wise: the first slice's share of it runs while the trailing collective is on the wire.  Adam
                # skips on the step's flag (bit 2), and at this point a rank's flag only knows its OWN partial sums: a rank that
                # overflowed alone would skip this slice while the others applied a summed gradient that holds its inf.  The
                # first slice carries the MLP rows, SUMMED by now: the check on them raises the flag on every rank alike (a
                # non-finite partial makes the sum non-finite everywhere) before the first Adam launch reads it.  With a bf16
                # payload the MLP rows travel in the trailing call, so there is no early slice (Adam after the check below).
                grad_sync.finish_first()
                self._call('nof_grad_check', self._seg(self.grads, 'mlp'), self.n_mlp, self.flags)
                self.adam_step(False, a, first_hi, advance=False)
                adam_done = (a, first_hi)
            grad_sync.finish()
            if nr:
                rest.copy_(self._grads_store[h - nr:h])
        elif grad_sync is not None:
            grad_sync(self.grads)
        if grad_sync is not None and self.world_size > 1:
            # every rank must skip the same steps: the non-finite check again, on the SUMMED weight gradient -- in every precision
            # (nof_reduce_partials raises the skip bit for a bf16 / fp32 backward too, and Adam honours it)
            self._call('nof_grad_check', self._seg(self.grads, 'mlp'), self.n_mlp, self.flags)
        if do_step:
            if adam_done is not None:                                  # (data parallel: everything on either side of the early slice)
                self.adam_step(dyn, 0, adam_done[0], advance=False)
                self.adam_step(dyn, adam_done[1])
            elif tail:
                ahead = None
                if (next_ids is not None and self.march_ahead and not dyn and self.marcher == lib.MARCHER_WAVE and self.level <= 5
                        and self.max_hits <= 196 and next_ids.shape[0] == R and not cfg.get('trunc_decay_type', '')):
                    ahead = (pool, next_ids, R, seed, b)
                self.adam_step_tail(dyn, ahead)
            else:
                self.adam_step(dyn)
        return b

    def adam_step(self, dyn=False, lo=0, hi=None, advance=True):
        """Adam over the flat entries [lo, hi) (default: all); `advance` = this call completes the optimiser step."""
        hi = self.n_total if hi is None else hi
        n, nb = hi - lo, min(max(self.n_basic - lo, 0), hi - lo)
        bufs = [x[lo:hi] for x in (self.params, self.grads, self.exp_avg, self.exp_avg_sq)]
        if dyn:
            cfg = self.cfg
            self._call('nof_adam_step_dyn', *bufs, n, nb, self._state, C.c_float(0.9), C.c_float(0.999), C.c_float(1e-15), self.flags)
            if advance:
                self._call('nof_step_state_advance', self._state, C.c_float(cfg['lrate']), C.c_float(cfg['lrate_pose']),
                           C.c_float(cfg['decay_rate']), int(cfg['n_step']) + 1, C.c_float(0.9), C.c_float(0.999), -1)
        else:
            lr, lr_pose = self.learning_rates()
            self._call('nof_adam_step', *bufs, n, nb, C.c_float(lr), C.c_float(lr_pose), C.c_float(0.9), C.c_float(0.999),
                       C.c_float(1e-15), self.adam_steps + 1, self.flags,
                       tag=None if lo == 0 and hi >= self.n_table else 'nof_adam_step[rest]')
        if advance:
            self.global_step += 1
            self.adam_steps += 1

    def _tail_ok(self, grad_sync=None):
        """may the optimiser launch carry the pose sums and the next step's prologue (fused_tail)?  Single GPU and the reference's
        defaults.  (The operand image it updates was packed for the current parameters by the step's own prologue -- or by the
        previous step's optimiser launch: its structural zeros are never rewritten.)"""
        return (self.fused_tail and grad_sync is None and self.world_size == 1 and self.optimize_poses and not self.eikonal
                and self.ff == 0 and float(self.cfg.get('pose_reg_weight', 0)) == 0)

    def adam_step_tail(self, dyn=False, ahead=None):
        """nof_pose_reduce_bwd + Adam over everything + nof_mlp_pack_pose for the next step (+ nof_step_state_advance in a captured
        step), as ONE launch (nof_adam_step_tail / nof_adam_step_tail_dyn); ahead = (pool, ids, R, seed, buffers): + the ray marcher
        of the next batch (nof_adam_step_tail_march)"""
        t = lib.NofAdamTail(C.addressof(self.desc), self.packed.data_ptr(), self.n_table, self.n_mlp, self.n_basic, self.F,
                            self.max_trans, self.max_rot, self.c2w.data_ptr(), self.tf.data_ptr(), self.pose_slots.data_ptr())
        if dyn:
            cfg = self.cfg
            self._call('nof_adam_step_tail_dyn', self.params, self.grads, self.exp_avg, self.exp_avg_sq, self.n_total, self.n_basic,
                       self._state, C.c_float(cfg['lrate']), C.c_float(cfg['lrate_pose']), C.c_float(cfg['decay_rate']),
                       int(cfg['n_step']) + 1, C.c_float(0.9), C.c_float(0.999), C.c_float(1e-15), self.flags, C.byref(t),
                       self._tail_done, tag='nof_adam_step')
        elif ahead is not None:
            pool, ids, R, seed, b = ahead
            lr, lr_pose = self.learning_rates()
            sc = self._sample_cfg(seed, self.global_step + 1)                  # (the next step's Philox counter)
            nx = lib.NofMarchNext(C.addressof(sc), pool.data_ptr(), ids.data_ptr(), self.occ_bits.data_ptr(), self.sh_degree,
                                  self.level, self.max_hits, 0, R, b['batch'].data_ptr(), b['rays_o_w'].data_ptr(),
                                  b['viewdirs_w'].data_ptr(), b['view'].data_ptr(), b['t_in_out'].data_ptr(), b['n_hits'].data_ptr(),
                                  b['z_vals'].data_ptr(), b['pts_w'].data_ptr(), b['valid'].data_ptr(), self.flags.data_ptr())
            self._tf_epoch = (self._tf_epoch + self.F) & 0xffffffff
            self._call('nof_adam_step_tail_march', self.params, self.grads, self.exp_avg, self.exp_avg_sq, self.n_total, self.n_basic,
                       C.c_float(lr), C.c_float(lr_pose), C.c_float(0.9), C.c_float(0.999), C.c_float(1e-15), self.adam_steps + 1,
                       self.flags, C.byref(t), C.byref(nx), self._tf_epoch_dev, self._tf_epoch, tag='nof_adam_step')
            self._marched = (self.global_step + 1, ids, R, pool.data_ptr(), seed)
        else:
            lr, lr_pose = self.learning_rates()
            self._call('nof_adam_step_tail', self.params, self.grads, self.exp_avg, self.exp_avg_sq, self.n_total, self.n_basic,
                       C.c_float(lr), C.c_float(lr_pose), C.c_float(0.9), C.c_float(0.999), C.c_float(1e-15), self.adam_steps + 1,
                       self.flags, C.byref(t), tag='nof_adam_step')
        self.global_step += 1
        self.adam_steps += 1
        self._packed_step = self._tail_step = self.global_step

    def gather_optimizer_state(self):
        """after steps with the sharded optimiser (GradSync mode 'zero1') a rank's Adam moments are current on its own shard only:
        all-gather them (a checkpoint needs all of them).  No-op otherwise."""
        import torch.distributed as dist
        if getattr(self, '_zero1_shard', None) is None or not dist.is_initialized():
            return
        from .dist import GradSync
        n_pad, shard, lo, hi = GradSync.shard_range(self.n_total)
        for buf in (self.exp_avg, self.exp_avg_sq):
            full = torch.zeros(n_pad, device=self.device)
            full[lo:hi] = buf[lo:hi]
            dist.all_gather_into_tensor(full, full[dist.get_rank() * shard:(dist.get_rank() + 1) * shard].clone())
            buf.copy_(full[:self.n_total])

    # ---- renderer side ------------------------------------------------------------------------------------
    def render_batch(self, pool, ids, R, want_cells=False):
        """The forward half of the step for R rays of `pool` with perturb=False, as render_images runs it (nerf_runner.py:595-612
        -> render -> batchify_rays -> render_rays -> raw2outputs): ray marching, UNPERTURBED z samples, encode, both MLPs,
        depth-guided compositing, and the depth read off the first SDF sign change.  No gradient, no optimiser state touched.
        Returns the step's buffer dict with `rgb_map` [R,3], `depth` [R], `raw`, `z_vals`, `valid` (views: copy before the next call)."""
        with self._on(torch.cuda.current_stream()):
            b, S = self.forward_batch(pool, ids, R, want_cells=want_cells, deterministic=True)
            lc = self._loss_cfg()
            self._call('nof_composite_loss', C.byref(lc), b['raw'], b['z_vals'], b['valid'], b['batch'], R, S, b['rgb_map'],
                       None, b['draw'], None, None)
            if 'depth' not in b:
                b['depth'] = torch.empty(R, device=self.device)
            self._call('nof_render_depth', b['raw'], b['z_vals'], R, S, C.c_float(self.cfg['far'] * self.cfg['sc_factor']), b['depth'])
        return b

    def view_row(self, viewdir=(0.0, 0.0, 0.0), frame_id=0):
        """[16] = [frame features | SH(viewdir) | 0]: the per-ray row the colour net consumes (nerf_runner.py:1270-1286), for ONE
        world direction and ONE frame's latent code -- what k_sample_points writes per ray, built here for free-standing queries."""
        x, y, z = (np.float32(v) for v in viewdir)
        sh = [np.float32(0.28209479177387814)]
        if self.sh_degree > 1:
            c1 = np.float32(0.4886025119029199)
            sh += [-c1 * y, c1 * z, -c1 * x]
        if self.sh_degree > 2:
            c = [np.float32(v) for v in (1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396)]
            sh += [c[0] * (x * y), c[1] * (y * z), c[2] * (np.float32(2) * z * z - x * x - y * y), c[3] * (x * z), c[4] * (x * x - y * y)]
        assert self.sh_degree <= 3 and self.ff + len(sh) <= 16
        row = torch.zeros(16, device=self.device)
        if self.ff > 0:
            row[:self.ff] = self.feat.view(self.F, self.ff)[int(frame_id)]
        row[self.ff:self.ff + len(sh)] = torch.tensor(np.array(sh, dtype=np.float32), device=self.device)
        return row

    def query_network(self, pts, viewdir=(0.0, 0.0, 0.0), frame_id=0, chunk=None):
        """run_network on free-standing points (nerf_runner.py:1226-1294 the way mesh_vertex_color_from_network :1412-1424 calls it:
        identity transform, one view direction, one frame's latent code): raw [N,4] = (colour logits, sdf).  Points outside [-1,1]^3
        get a zero embedding like the reference's valid_samples (:1246-1257)."""
        pts = pts.to(self.device, torch.float32).contiguous()
        self.pack_weights()
        N = pts.shape[0]
        chunk = chunk or (1 << 18 if self.wide else 1 << 20)          # (the wide forward stages ~1.5 KB of activations per point)
        raw = torch.empty(N, 4, device=self.device)
        view = self.view_row(viewdir, frame_id).view(1, 16).contiguous()
        for i in range(0, N, chunk):
            n = min(chunk, N - i)
            p = pts[i:i + n]
            if self.fused_forward:
                lib.call('nof_encode_mlp_fwd', C.byref(self.grid), C.byref(self.desc), self.packed, self.table, p, view, n, raw[i:i + n],
                         None, None, n)
                continue
            feat = torch.empty(self.L, n, 2, device=self.device)
            lib.call('nof_hash_encode_fwd', C.byref(self.grid), p, self.table, feat, n)
            if self.wide:
                ws = torch.empty(int(lib.load().nof_mlp_wide_workspace_bytes(C.byref(self.desc), n)), dtype=torch.uint8, device=self.device)
                lib.call('nof_mlp_wide_fwd', C.byref(self.desc), self.packed, feat, self.L, view, n, raw[i:i + n], ws, n)
            else:
                lib.call('nof_mlp_fwd', C.byref(self.desc), self.packed, feat, self.L, view, n, raw[i:i + n], None, n)
        return raw

    def query_sdf(self, pts, chunk=1 << 22):
        """run_network_density (nerf_runner.py:1307-1347): clip to [-1,1], hash encode, sigma_net -> sdf [N]."""
        pts = torch.clip(pts.to(self.device, torch.float32), -1, 1).contiguous()
        self.pack_weights()
        N = pts.shape[0]
        out = torch.empty(N, device=self.device)
        for i in range(0, N, chunk):
            n = min(chunk, N - i)
            feat = torch.empty(self.L, n, 2, device=self.device)
            p = pts[i:i + n].contiguous()
            lib.call('nof_hash_encode_fwd', C.byref(self.grid), p, self.table, feat, n)
            lib.call('nof_mlp_wide_sdf' if self.wide else 'nof_mlp_sdf', C.byref(self.desc), self.packed, feat, self.L, out[i:i + n], n)
        return out

    def query_sdf_grid(self, tx, ty, tz, outside_value=1.0, use_octree=True):
        """extract_mesh's dense query (nerf_runner.py:1351-1386) fused into one launch: sdf [nx,ny,nz] on the device for the
        voxel centres (tx[i], ty[j], tz[k]); voxels outside the octree get `outside_value`."""
        self.pack_weights()
        ax = [torch.as_tensor(np.asarray(a, dtype=np.float32)).to(self.device).contiguous() for a in (tx, ty, tz)]
        nx, ny, nz = (int(a.numel()) for a in ax)
        out = torch.empty(nx, ny, nz, device=self.device)
        occ = self.occ_bits if use_octree else None
        if self.wide:
            # no fused grid kernel for the wide networks: the octree mask of EVERY voxel centre in one nof_occgrid_query (slabs of
            # x only bound the scratch for the point list), ONE compaction (the only host synchronisation: the number of voxels
            # inside), then hash encode + sigma net of the voxels inside in large chunks (query_sdf), scattered back -- all on the
            # device.  (Round 3 looped over x slabs in Python with a host synchronisation per slab.)
            out.fill_(outside_value)
            flat = out.view(-1)
            yz = torch.stack(torch.meshgrid(ax[1], ax[2], indexing='ij'), -1).reshape(-1, 2)
            slab = max(1, (1 << 24) // max(yz.shape[0], 1))               # x planes per scratch buffer (~16 M points)
            idx_parts, pts_parts = [], []
            for i0 in range(0, nx, slab):
                xs = ax[0][i0:i0 + slab]
                pts = torch.cat([xs.repeat_interleave(yz.shape[0]).unsqueeze(1), yz.repeat(xs.numel(), 1)], -1).contiguous()
                if occ is not None:
                    inside = torch.empty(pts.shape[0], dtype=torch.uint8, device=self.device)
                    lib.call('nof_occgrid_query', occ, self.level, pts, inside, pts.shape[0])
                    sel = torch.nonzero(inside).reshape(-1)                  # (device-side compaction; its size is read below, once per slab group)
                    idx_parts.append(sel + i0 * yz.shape[0])
                    pts_parts.append(pts[sel])
                else:
                    idx_parts.append(torch.arange(pts.shape[0], device=self.device) + i0 * yz.shape[0])
                    pts_parts.append(pts)
            idx = torch.cat(idx_parts)
            if idx.numel():
                flat[idx] = self.query_sdf(torch.cat(pts_parts))
            return out
        lib.call('nof_sdf_grid_query', C.byref(self.grid), C.byref(self.desc), self.packed, self.table, occ, self.level,
                 ax[0], ax[1], ax[2], nx, ny, nz, C.c_float(outside_value), out)
        return out

    def losses(self):
        """dict of the last step's loss terms (one host sync)."""
        v = self.loss_out.cpu().numpy()
        return dict(loss=float(v[0]), rgb_loss=float(v[1]), fs_loss=float(v[2]), sdf_loss=float(v[3]),
                    fs_rgb_loss=float(v[4]), n_valid_samples=float(v[5]), n_valid_rays=float(v[6]), eikonal_loss=float(v[7]))


class GraphedStep:
    """One optimisation step captured into a HIP graph and replayed: 20-odd launches, two fills and the side-stream fork/join
    of the hash backward become one graph launch.  What changes from step to step lives in device memory: the batch ids in a
    static buffer (copied in before each replay), the Philox step / Adam step sizes / scheduled learning rates in the field's
    NofStepState (advanced by the last launch of the captured step).  Not used with a data-parallel gradient sync, with
    per-kernel event timing, or with a truncation schedule (the truncation is baked into the captured launches)."""

    def __init__(self, field, pool, R, seed):
        self.field, self.R = field, R
        self.ids = torch.zeros(R, dtype=torch.int64, device=field.device)
        # every persistent buffer of the step exists BEFORE the capture starts: allocated inside it they would live in the graph's
        # private pool (and be reused by eager steps after the graph is dropped)
        field._buffers(R, field.cfg['N_samples'] + field.cfg['N_samples_around_depth'])
        field._side_stream()
        field.sync_step_state()
        # a captured step whose optimiser launch leaves the NEXT step's operand image and pose table (nof_adam_step_tail_dyn) holds
        # no packing launch: the first replay starts from what is built here
        field.pack_weights(force=True)
        field.update_poses()
        field._tail_step = field.global_step
        step0, adam0 = field.global_step, field.adam_steps
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            with torch.cuda.graph(self.graph, stream=s):
                field.train_step(pool, self.ids, R, seed=seed, dyn=True)
        torch.cuda.current_stream().wait_stream(s)
        self.has_tail = field._tail_step == field.global_step and field.global_step == step0 + 1   # (set by adam_step_tail)
        field.global_step, field.adam_steps = step0, adam0       # capturing executes nothing: only the host-side counters moved
        field._packed_step = field._tail_step = step0 if self.has_tail else None
        self.trunc = field.truncation()
        self.grad_scale = float(field.desc.grad_scale)           # baked into the captured launches

    def usable(self):
        f = self.field
        # (the device step state drives the schedule AND Adam's bias correction from one counter)
        f._set_grad_scale(self.R * (f.cfg['N_samples'] + f.cfg['N_samples_around_depth']))
        return (f.profile is None and f.truncation() == self.trunc and f.adam_steps == f.global_step
                and float(f.desc.grad_scale) == self.grad_scale)   # (a loss-scale back-off since the capture: capture again)

    def __call__(self, ids):
        f = self.field
        if self.has_tail and not (f._packed_step == f._tail_step == f.global_step):
            f.pack_weights(force=True)         # (the parameters were replaced since the last replay: the graph itself packs nothing)
            f.update_poses()
        self.ids.copy_(ids)
        self.graph.replay()
        f.global_step += 1
        f.adam_steps += 1
        # without the tail: the fragment image inside the graph belongs to the parameters before this step
        f._packed_step = f._tail_step = f.global_step if self.has_tail else None
