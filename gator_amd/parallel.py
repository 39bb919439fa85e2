"""Multi-GPU execution of the path: samples are independent, so the batch is sharded contiguously across
ranks (one process per GPU, weights replicated) and the ONLY collective is

  mode='gather' (BASELINE config 4): an all-gather of the predicted vertices [B/N, 6890, 3] (+ pose3d [B/N, J, 3]) over
                xGMI -- RCCL via torch.distributed's "nccl" backend;
  mode='eval'   (BASELINE config 5): joints are regressed and the MPJPE / PA-MPJPE sums formed on the device
                (gator_amd.eval), and only those few scalars are all-reduced -- no rank ever holds another rank's meshes and no
                mesh goes to the host (lib/core/base.py:219-237 copies every mesh to the host twice).

The reference has no multi-GPU counterpart (no torch.distributed anywhere); correctness criterion: the
gathered tensor equals the single-GPU output of the concatenated batch bit for bit (same kernels, per-sample arithmetic
independent of B), resp. the reduced sums equal the single-process sums.

One code path for every backend.  The chunk plan, the output-buffer rotation and the gather itself are the same Python for a
HIP device ("nccl") and for host tensors ("gloo", the CPU tests); the only difference is that on a device the collective is
issued on a side stream.

Overlap and buffer ownership.  The collective of step k runs on the side stream and the compute stream does not wait for
it, so it hides behind step k+1's kernels (xGMI is point-to-point, 7 links x ~153 GB/s per GPU: a ring all-gather of
21 MB/rank per 256 samples costs ~1 ms at 8 GPUs, the same order as the compute).  Outputs rotate through `depth` buffers
(default 2).  Ordering is by streams, never by host waits:
  * `step()` returns the buffers of THIS step and sets `last_event`; the consumer calls `wait()` (current stream waits
    for the gather) before reading them;
  * the buffers returned by step k are overwritten by the gather of step k+depth, which the side stream starts only after
    everything the CALLING stream had queued when step k+depth was called (wait_stream), i.e. after every read the consumer
    queued on that stream in between.  A consumer on another stream must synchronise with the calling stream itself;
  * the backlog is bounded: before step k issues its kernels, the calling stream waits for the collective of step k - depth
    (the one whose output buffer step k re-uses).  If the gather is slower than the compute, the compute therefore runs at the
    gather's rate `depth` steps ahead instead of queueing collectives (and the temporaries pinned for them) without limit.
    `max_inflight` records the largest number of collectives that were ever outstanding (tests assert <= depth).

Encoder choice.  The library picks the GAT encoder kernel from the batch size of each call (gator_set_encoder); the two kernels
agree to fp32 rounding, not bit for bit.  With more than one rank that would make a sample's last bits depend on how many ranks
the batch was cut into, so a sharded run pins ONE kernel for all its calls: the one the library would use for the local batch
(`encoder='auto'` here), or the caller's choice.  Run the single-process reference with the same pin to compare bit for bit."""
import contextlib

import torch

from ._lib import DeferredDeviceStatus


class ShardedForward:
    def __init__(self, model, world_size=1, rank=0, dist=None, micro_batch=None, always_gather=False, mode='gather', depth=2,
                 metrics_fn=None, encoder='auto'):
        if mode not in ('gather', 'eval'):
            raise ValueError("mode must be 'gather' or 'eval'")
        self.model, self.world, self.rank, self.dist = model, world_size, rank, dist
        self.micro = micro_batch
        self.always_gather = always_gather      # run the collective even for one rank (exercises RCCL on a 1-GPU box)
        self.mode = mode
        self.depth = max(1, int(depth))
        self.metrics_fn = metrics_fn            # eval mode: (verts, pose3d, target, sample slice) -> 1-D tensor of partial sums
        self._bufs = {}
        self._stage = {}
        self._stage_free = {}                   # staging slot -> event after which it may be written again
        self._comm_stream = None
        self._flip = 0
        self._target = None
        self._regressor = None
        self._regressor_dense = None
        self._eval_joints = None
        self._last = None                       # (kind, tensors) of the last collective, for comm_only()
        self.last_event = None
        self._inflight = []                     # completion events (None on the host) of the collectives not yet waited for
        self.max_inflight = 0
        if encoder not in ('auto', 'sample', 'tiled'):
            raise ValueError("encoder must be 'auto', 'sample' or 'tiled'")
        self.encoder = encoder
        self._pinned = None

    TILED_MIN_BATCH = 1025                      # fallback only (a model without encoder_for_batch / no context yet): the library's default rule

    def _pin_encoder(self, calls_batch, device=None):
        """world > 1: one encoder kernel for every call of this run (see the module docstring)."""
        if self._pinned is not None or not hasattr(self.model, 'set_encoder'):
            return
        want = self.encoder
        if want == 'auto':
            if self.world <= 1 and not self.always_gather:
                self._pinned = 'auto'
                return
            ask = getattr(self.model, 'encoder_for_batch', None)      # the library's own decision (its switches, the device's CU count)
            if ask is not None and device is not None and device.type == 'cuda' and hasattr(self.model, '_context'):
                self.model._context(device)                           # the decision needs the packed context (created on first use anyway)
            want = ask(calls_batch) if ask is not None else None
            if want is None:
                want = 'tiled' if calls_batch >= self.TILED_MIN_BATCH else 'sample'
        self.model.set_encoder(want)
        self._pinned = want

    def _throttle(self, device):
        """Before a step queues work: at most `depth` collectives outstanding (the oldest one wrote the buffer this step re-uses)."""
        while len(self._inflight) >= self.depth:
            ev = self._inflight.pop(0)
            if ev is not None:
                torch.cuda.current_stream(device).wait_event(ev)

    def _issued(self, event):
        self._inflight.append(event)
        self.max_inflight = max(self.max_inflight, len(self._inflight))

    # ---- shared plan: chunks of the local batch, rotating output buffers ------------------------------------------------
    def _plan(self, B):
        mb = self.micro or B
        return [(s, min(B, s + mb)) for s in range(0, B, mb)]

    def _buffers(self, B, J, device):
        self._flip = (self._flip + 1) % self.depth
        key = (B, J, str(device), self._flip)
        if key not in self._bufs:
            self._bufs[key] = (torch.empty((self.world * B, 6890, 3), device=device, dtype=torch.float32),
                               torch.empty((self.world * B, J, 3), device=device, dtype=torch.float32))
        return self._bufs[key]

    def _staging(self, n, J, device, slot=0):
        """[world, n, ...] landing tiles of a micro-batch's collective; two slots in turn, so that the forward of chunk k + 1 can
        write its slot while the gather of chunk k still runs on the side stream."""
        key = (n, J, str(device), slot & 1)
        if key not in self._stage:
            self._stage[key] = (torch.empty((self.world, n, 6890, 3), device=device, dtype=torch.float32),
                                torch.empty((self.world, n, J, 3), device=device, dtype=torch.float32))
        return key, self._stage[key]

    def _side(self, device):
        """Context in which collectives are issued: the side stream on a HIP device, nothing on the host."""
        if device.type != 'cuda':
            return contextlib.nullcontext(), None
        if self._comm_stream is None:
            self._comm_stream = torch.cuda.Stream(device=device)
        cur = torch.cuda.current_stream(device)
        self._comm_stream.wait_stream(cur)      # after the producing kernels AND after every queued read of the buffer we reuse
        return torch.cuda.stream(self._comm_stream), self._comm_stream

    def _landing(self, s, e, B, J, gv, gp, slot):
        """Where this rank's rows [s,e) must lie for the collective to run IN PLACE (send buffer = this rank's slot of the receive
        buffer, RCCL's in-place all-gather): its slice of the output when the chunk is the whole local batch, its slot of a
        [world, n, ...] staging tile for a micro-batch.  The forward writes there directly: no fresh output allocation and no
        21 - 85 MB copy of the local shard into the gather buffer.  -> (receive verts, receive pose3d, my verts, my pose3d, key)"""
        if s == 0 and e == B:
            return gv, gp, gv[self.rank * B:(self.rank + 1) * B], gp[self.rank * B:(self.rank + 1) * B], None
        key, (sv, sp) = self._staging(e - s, J, gv.device, slot)
        return sv.view(-1, 6890, 3), sp.view(-1, J, 3), sv[self.rank], sp[self.rank], key

    def _gather_chunk(self, verts, pose3d, s, e, B, gv, gp, slot=0):
        """All ranks' rows [s,e) of their local batches -> rows r*B+s .. r*B+e of the replicated outputs."""
        dev, J = verts.device, pose3d.shape[1]
        ctx, side = self._side(dev)
        rv, rp, mv, mp, key = self._landing(s, e, B, J, gv, gp, slot)
        with ctx:
            # whole local batch: rank-major concatenation IS the output layout; micro-batch: one gather into the staging tile, then
            # one strided copy.  `verts` is normally `mv` itself (the forward wrote its rank's slot); a model without `out=` hands
            # back its own tensor and the collective copies it in.
            self.dist.all_gather_into_tensor(rv, verts)
            self.dist.all_gather_into_tensor(rp, pose3d)
            if key is not None:
                gv.view(self.world, B, 6890, 3)[:, s:e].copy_(rv.view(self.world, e - s, 6890, 3))
                gp.view(self.world, B, J, 3)[:, s:e].copy_(rp.view(self.world, e - s, J, 3))
                if side is not None:
                    self._stage_free[key] = side.record_event()      # the next writer of this staging slot waits for this
        if side is not None and verts.data_ptr() != mv.data_ptr():
            verts.record_stream(side)
            pose3d.record_stream(side)
        return side

    def wait(self):
        """Make the current stream wait for the collective of the last step (call before consuming its outputs)."""
        if self.last_event is not None:
            torch.cuda.current_stream().wait_event(self.last_event)

    @staticmethod
    def _forward(fn, *a, **k):
        """-> (outputs, deferred report or None).  A module with on_device_status = 'raise' hands an EARLIER forward's device status to the
        caller as DeferredDeviceStatus (GATOR_EDEVICE_DEFERRED, include/gator_hip.h) on a call that WAS queued normally: the exception
        carries that call's outputs, this rank goes on with them -- the other ranks are already in the collective -- and the report
        is raised once every collective of the step is issued.  (The default, 'heal', never raises: gator_amd/models/_base.py.)"""
        try:
            return fn(*a, **k), None
        except DeferredDeviceStatus as ex:
            if ex.outputs is None:
                raise
            return ex.outputs, ex

    # ---- mode 'gather' ---------------------------------------------------------------------------------------------------
    def step(self, pose2d_shard):
        """pose2d_shard [B_local, J, 2] on this rank's GPU.
        gather: -> (verts [world*B_local,6890,3], pose3d [world*B_local,J,3]) replicated on every rank (rank r's samples at
                rows r*B_local ...);
        eval:   -> 1-D tensor of the globally reduced error sums (see set_eval)."""
        if self.mode == 'eval':
            return self._step_eval(pose2d_shard)
        if self.dist is None or (self.world == 1 and not self.always_gather):
            return self.model(pose2d_shard)
        B, J = pose2d_shard.shape[0], pose2d_shard.shape[1]
        self._pin_encoder(min(B, self.micro or B), pose2d_shard.device)
        self._throttle(pose2d_shard.device)
        gv, gp = self._buffers(B, J, pose2d_shard.device)
        side, chunks, deferred = None, [], None
        in_place = bool(getattr(self.model, 'supports_out', False))
        for i, (s, e) in enumerate(self._plan(B)):
            if in_place:                        # the forward writes this rank's slot of the collective's receive buffer
                _, _, mv, mp, key = self._landing(s, e, B, J, gv, gp, i)
                free = self._stage_free.pop(key, None)
                if free is not None:            # a staging slot is re-used: its last collective + copy-out must be done first
                    torch.cuda.current_stream(pose2d_shard.device).wait_event(free)
                (verts, pose3d), ex = self._forward(self.model, pose2d_shard[s:e], out=(mv, mp))
            else:
                (verts, pose3d), ex = self._forward(self.model, pose2d_shard[s:e])
                verts, pose3d = verts.contiguous(), pose3d.contiguous()
            deferred = deferred or ex
            side = self._gather_chunk(verts, pose3d, s, e, B, gv, gp, i)
            chunks.append((verts, pose3d, s, e, i))
        self._last = ('gather', (chunks, B, gv, gp))
        # the compute stream does NOT wait: the next step overlaps this gather
        self.last_event = side.record_event() if side is not None else None
        self._issued(self.last_event)
        if deferred is not None:
            raise deferred
        return gv, gp

    # ---- mode 'eval' -----------------------------------------------------------------------------------------------------
    def set_eval(self, j_regressor, target_joints, eval_joints=None):
        """j_regressor: dense [n_joint, 6890] (J_regressor_h36m, lib/core/base.py:221); target_joints: this rank's ground-truth
        joints [B_local, n_joint, 3] in mm on the same device as the inputs."""
        self._target = target_joints
        self._eval_joints = eval_joints
        self._regressor_dense = j_regressor
        self._regressor = None

    def _joint_metrics(self, joints_m, target, sl):
        """[sum MPJPE, sum PA-MPJPE, samples] from regressed joints in metres (gator_amd.eval, all on the device)."""
        from . import eval as ev
        kw = {} if self._eval_joints is None else {'eval_joints': self._eval_joints}
        err = ev.joint_errors(joints_m, target[sl], pred_scale=1000.0, **kw)     # metres -> mm (lib/core/base.py:219); one launch
        n = err.shape[0]
        if getattr(self, '_count', None) is None or self._count[0] != n or self._count[1].device != err.device:
            self._count = (n, torch.tensor([float(n)], device=err.device, dtype=torch.float64))
        return torch.cat([err.double().sum(0), self._count[1]])

    def _device_metrics(self, verts, pose3d, target, sl):
        """Same from materialised vertices (models without the fused joint-regression epilogue)."""
        from . import eval as ev
        if self._regressor is None:
            self._regressor = ev.JointRegressor(self._regressor_dense, verts.device)
        joints = self._regressor(verts) * 1000.0
        tgt = target[sl]
        kw = {} if self._eval_joints is None else {'eval_joints': self._eval_joints}
        n = joints.shape[0]
        return torch.stack([ev.mpjpe(joints, tgt, **kw) * n, ev.pa_mpjpe(joints, tgt, **kw) * n,
                            torch.tensor(float(n), device=verts.device)]).double()

    def _step_eval(self, pose2d_shard):
        fn = self.metrics_fn or self._device_metrics
        # the HIP model regresses the joints in the vertex GEMM's epilogue: no vertex is ever written (gator_forward_joints_f32)
        fused = self.metrics_fn is None and hasattr(self.model, 'forward_joints') and self._regressor_dense is not None
        if fused and getattr(self.model, '_jreg', None) is None:
            self.model.set_joint_regressor(self._regressor_dense)
        B = pose2d_shard.shape[0]
        self._pin_encoder(min(B, self.micro or B), pose2d_shard.device)
        self._throttle(pose2d_shard.device)
        acc, deferred = None, None
        for s, e in self._plan(B):
            if fused:
                (joints, pose3d), ex = self._forward(self.model.forward_joints, pose2d_shard[s:e])
                part = self._joint_metrics(joints, self._target, slice(s, e))
            else:
                (verts, pose3d), ex = self._forward(self.model, pose2d_shard[s:e])
                part = fn(verts, pose3d, self._target, slice(s, e))
            deferred = deferred or ex
            acc = part if acc is None else acc + part
        if self.dist is not None and (self.world > 1 or self.always_gather):
            ctx, side = self._side(acc.device)
            with ctx:
                self.dist.all_reduce(acc)
            if side is not None:
                acc.record_stream(side)
                self.last_event = side.record_event()
            self._issued(self.last_event if side is not None else None)
            self._last = ('reduce', (acc,))
        if deferred is not None:
            raise deferred
        return acc

    # ---- measurement helper ---------------------------------------------------------------------------------------------
    def comm_only(self):
        """Re-issue the last step's collective alone (bench.py: what the collective costs without compute to hide behind)."""
        if self._last is None:
            return None
        kind, t = self._last
        if kind == 'gather':
            chunks, B, gv, gp = t
            for verts, pose3d, s, e, i in chunks:
                land = self._landing(s, e, B, pose3d.shape[1], gv, gp, i)
                if not (s == 0 and e == B) and verts.data_ptr() == land[2].data_ptr():
                    # in-place micro-batch: a later chunk has re-used the staging slot; put this rank's rows back first -- behind the
                    # last collective + copy-out that used the slot (as step() does before it lets a forward write there)
                    free = self._stage_free.pop(land[4], None)
                    if free is not None:
                        torch.cuda.current_stream(verts.device).wait_event(free)
                    verts.copy_(gv.view(self.world, B, 6890, 3)[self.rank, s:e])
                    pose3d.copy_(gp.view(self.world, B, -1, 3)[self.rank, s:e])
                side = self._gather_chunk(verts, pose3d, s, e, B, gv, gp, i)
        else:
            ctx, side = self._side(t[0].device)
            with ctx:
                self.dist.all_reduce(t[0])
        if side is not None:
            self.last_event = side.record_event()
        return None
