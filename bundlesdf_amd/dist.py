"""Data-parallel plumbing: one process per GPU, torch.distributed ('nccl' == RCCL over xGMI on ROCm; 'gloo' in the CPU
tests).  Keyframes are sharded by rank; the hash table, MLPs, the FULL pose/feature arrays and the occupancy grid are
replicated.  Each rank draws N_rand rays from its own pool, so the global batch is world*N_rand; every loss term is a
mean over rays/samples, hence gradients are pre-scaled by 1/world on the device (NofLossCfg.grad_scale) and SUMMED here:
the result equals one process stepping on the concatenated batch.  The exchange is the all-reduce of the flat gradient buffer
[table | MLP | features | poses], issued as TWO collectives per step: the fine hash levels + MLP slice (80 % of
the bytes) as soon as it is final, overlapped with the rest of the backward, and everything else in one trailing call."""
import os

import torch
import torch.distributed as dist


def init_from_env(device=None):
    """(rank, world, local_rank) from the torchrun environment; initialises the process group when world > 1."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        if device is not None and device.type == 'cuda':
            dist.init_process_group('nccl', device_id=device)
        else:
            dist.init_process_group('gloo')
    return rank, world, local_rank


def all_gather_cat(t):
    """concatenate equally-shaped tensors of all ranks along dim 0 (identical result on every rank)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return t
    parts = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(parts, t.contiguous())
    return torch.cat(parts, 0)


class GradSync:
    """Hook for NeuralObjectField.train_step: sums the (pre-scaled) flat gradients of all ranks.

    Called as a function it is one blocking all-reduce of the whole buffer.  `start(slice)` / `finish()` is the bucketed
    form the step uses: the fine hash levels' slice (80 % of the bytes) is reduced asynchronously -- torch's process group runs
    the collective on its own stream, ordered after the launches already enqueued on the current one -- while the scatter
    of the coarse levels and the pose kernels still run; `finish_first()` / `finish()` make the current stream wait before the
    slice's share of Adam.  Sums are
    element-wise, so bucketing does not change a single bit of the result."""

    def __init__(self, payload='fp32', mode='allreduce'):
        assert payload in ('fp32', 'bf16') and mode in ('allreduce', 'zero1', 'rows')
        # 'rows' (round 6; SURVEY 8e "per level"): only the table rows SOME rank touched in this step travel.  A step's table
        # gradient is zero except in the rows its samples' corners hit (cfg2, settled: one row in ten on a rank); the ranks
        # all-gather a bitmap of their non-zero rows (1 bit per row), OR them, and all-reduce the union's rows only -- the other
        # rows are zero on every rank, so the result equals the dense all-reduce's entry for entry (at world size 2 bit for bit: a
        # two-term sum does not depend on its order; beyond that up to the ring's summation order, like any all-reduce).  Everything
        # behind the table (MLP, frame features, poses) goes dense in a second call.  One host sync per step (the union's size is
        # the collective's size); nothing overlaps the backward.  Opt-in like the other two: the union grows with the world size
        # (1 - 0.9^N: 19 % of the rows at N = 2, 57 % at N = 8), so what it saves shrinks where the wire matters most.
        # 'zero1' (SURVEY 8e): reduce-scatter of the flat gradient -> every rank runs Adam on ITS 1/world of the flat buffers ->
        # all-gather of the parameters.  The same bytes on the wire as the all-reduce (which is a reduce-scatter + an all-gather
        # of GRADIENTS), the optimiser pass cut to 1/world per rank; the exchange sits between the backward and the next forward
        # with nothing to hide behind, which is why it is opt-in (DESIGN 6 has the cost model).  field.py drives it.
        self.mode = mode
        self.payload = payload       # 'bf16': slices started with compressed=True travel as bfloat16 (half the bytes on the wire)
        self.pending = []            # [(work, fp32 slice to refill from its staging buffer or None, staging buffer)]
        self.bytes_step = 0          # payload bytes handed to all-reduce in the last step (per rank)
        self.collectives_step = 0
        self._bytes = self._n = 0
        self._staging = None
        self.timing = None           # list of (event before a wait, event after it): bench.py's exposed-communication time
        self.timed_steps = 0

    def __call__(self, flat):
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        self.bytes_step, self.collectives_step = flat.numel() * flat.element_size(), 1

    # ---- 'zero1' ------------------------------------------------------------------------------------------------------
    @staticmethod
    def shard_range(n_total, rank=None, world=None):
        """(n_padded, shard, lo, hi): the flat buffers are padded to a multiple of 4 * world floats (Adam moves 16 bytes per lane);
        rank r owns entries [r * shard, (r + 1) * shard) of the padded buffer = [lo, hi) of the real one."""
        world = dist.get_world_size() if world is None else world
        rank = dist.get_rank() if rank is None else rank
        q = 4 * world
        n_pad = (n_total + q - 1) // q * q
        shard = n_pad // world
        lo = min(rank * shard, n_total)
        return n_pad, shard, lo, min(lo + shard, n_total)

    def reduce_scatter_(self, padded):
        """in place: this rank's shard of `padded` [n_pad] becomes the sum over the ranks (the rest of the buffer is left as it is)"""
        world, rank = dist.get_world_size(), dist.get_rank()
        shard = padded.numel() // world
        ev = self._ev0()
        dist.reduce_scatter_tensor(padded[rank * shard:(rank + 1) * shard], padded, op=dist.ReduceOp.SUM)
        self._ev1(ev)
        self._bytes += padded.numel() * padded.element_size()
        self._n += 1

    def all_gather_(self, padded):
        """in place: every rank's shard of `padded` is distributed to all ranks"""
        world, rank = dist.get_world_size(), dist.get_rank()
        shard = padded.numel() // world
        ev = self._ev0()
        dist.all_gather_into_tensor(padded, padded[rank * shard:(rank + 1) * shard])
        self._ev1(ev)
        self._bytes += padded.numel() * padded.element_size()
        self._n += 1

    # ---- 'rows' -------------------------------------------------------------------------------------------------------
    @staticmethod
    def _row_bitmap(rows):
        """[n, w] float rows -> uint8 [ceil(n / 8)]: bit (i & 7) of byte (i >> 3) = row i has a non-zero entry"""
        nz = (rows != 0).any(1)
        n = nz.numel()
        pad = (-n) % 8
        if pad:
            nz = torch.cat([nz, nz.new_zeros(pad)])
        w = torch.tensor([1, 2, 4, 8, 16, 32, 64, 128], dtype=torch.uint8, device=rows.device)
        return (nz.view(-1, 8).to(torch.uint8) * w).sum(1, dtype=torch.uint8)

    @staticmethod
    def _bitmap_rows(bits, n):
        """the inverse: indices (int64, ascending) of the set bits among the first n"""
        sh = torch.arange(8, dtype=torch.uint8, device=bits.device)
        on = ((bits[:, None] >> sh) & 1).reshape(-1)[:n]
        return torch.nonzero(on).reshape(-1)

    def exchange_rows_(self, table_grad, width=2):
        """in place: `table_grad` (flat fp32, rows of `width` entries) becomes the sum over the ranks, moving only the rows that are
        non-zero on at least one rank.  Blocking; returns the number of rows that travelled."""
        rows = table_grad.view(-1, width)
        n = rows.shape[0]
        world = dist.get_world_size()
        ev = self._ev0()
        mine = self._row_bitmap(rows)
        every = torch.empty(world * mine.numel(), dtype=torch.uint8, device=mine.device)
        dist.all_gather_into_tensor(every, mine)
        every = every.view(world, -1)
        union = every[0]
        for r in range(1, world):
            union = union | every[r]
        idx = self._bitmap_rows(union, n)                       # (torch.nonzero: the step's one host sync in this mode)
        k = int(idx.numel())
        self._bytes += mine.numel()                             # (payload handed to the collectives per rank, like the other modes)
        self._n += 1
        if k:
            vals = rows.index_select(0, idx)
            dist.all_reduce(vals, op=dist.ReduceOp.SUM)
            rows.index_copy_(0, idx, vals)
            self._bytes += vals.numel() * vals.element_size()
            self._n += 1
        self._ev1(ev)
        self.rows_step = k
        return k

    def exchange_dense_(self, part):
        """blocking all-reduce of a slice, counted into the step's totals (the 'rows' mode's second call: everything behind the table)"""
        if part.numel():
            ev = self._ev0()
            dist.all_reduce(part, op=dist.ReduceOp.SUM)
            self._ev1(ev)
            self._bytes += part.numel() * part.element_size()
            self._n += 1

    def max_flags_(self, flags):
        """the ranks agree on the SKIP predicate of the device flag word (bit 2 of flags[0]: this step's gradient is not finite -- a
        skipped step must be skipped by everyone).  Only that bit travels: MAX over a bit mask is not a bitwise OR (a rank holding
        only bit 3 would beat a rank holding bit 2), and the other bits are per-rank diagnostics that stay what they were."""
        skip = flags[0:1] & 4
        dist.all_reduce(skip, op=dist.ReduceOp.MAX)
        flags[0:1] |= skip

    def end_step(self):
        self.timed_steps += 1
        self.bytes_step, self.collectives_step, self._bytes, self._n = self._bytes, self._n, 0, 0

    def _ev0(self):
        if self.timing is not None and torch.cuda.is_available():
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
            return ev
        return None

    def _ev1(self, ev):
        if ev is not None:
            ev[1].record()
            self.timing.append(ev)

    def start(self, part, compressed=False):
        """asynchronous all-reduce of a slice of the flat fp32 gradient buffer.  compressed (and payload 'bf16'): the slice is
        rounded to bfloat16 into a staging buffer, summed in bfloat16 -- every rank receives the same bits, so replicas stay
        identical -- and written back to the slice when the collective is waited for.  Meant for the table slice only: Adam
        divides a gradient by its own running magnitude, so a 2^-8 relative error per entry moves no parameter by more than that
        fraction of a step (tests/test_gpu_dp.py measures the loss drift over 50 steps)."""
        if not part.numel():
            return
        if compressed and self.payload == 'bf16':
            if self._staging is None or self._staging.numel() < part.numel() or self._staging.device != part.device:
                self._staging = torch.empty(part.numel(), dtype=torch.bfloat16, device=part.device)
            st = self._staging[:part.numel()]
            st.copy_(part)
            self.pending.append((dist.all_reduce(st, op=dist.ReduceOp.SUM, async_op=True), part, st))
            self._bytes += st.numel() * st.element_size()
        else:
            self.pending.append((dist.all_reduce(part, op=dist.ReduceOp.SUM, async_op=True), None, None))
            self._bytes += part.numel() * part.element_size()
        self._n += 1

    def _wait(self, works):
        ev = None
        if self.timing is not None and torch.cuda.is_available():
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        for w, part, st in works:
            w.wait()
            if part is not None:
                part.copy_(st)
        if ev is not None:
            ev[1].record()
            self.timing.append(ev)

    def finish_first(self):
        """the current stream waits for the OLDEST collective in flight only (the step then runs that slice's share of Adam
        while the trailing collective is still on the wire)"""
        if self.pending:
            self._wait([self.pending.pop(0)])

    def finish(self):
        """the current stream waits for every collective in flight; what of them did not hide behind the backward is the time
        between the event pairs recorded around the waits (when `timing` is a list and the tensors live on a GPU:
        `timed_steps` counts the steps they belong to)"""
        self._wait(self.pending)
        self.pending = []
        self.timed_steps += 1
        self.bytes_step, self.collectives_step, self._bytes, self._n = self._bytes, self._n, 0, 0


def make_grad_sync(overlap=True, payload='fp32', mode='allreduce'):
    """GradSync when a process group with more than one rank exists, else None."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return None
    g = GradSync(payload, mode)
    return g if overlap or mode in ('zero1', 'rows') else g.__call__


def shard_frames(n_total, rank, world):
    """contiguous keyframe shard [lo, hi) of this rank."""
    per = (n_total + world - 1) // world
    lo = min(rank * per, n_total)
    return lo, min(lo + per, n_total)
