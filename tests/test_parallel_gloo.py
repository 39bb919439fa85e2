"""N>1 path on CPU: world_size-2 gloo run of gator_amd.parallel.ShardedForward -- the SAME chunk plan, buffer rotation and
collective calls the device path runs (only the side stream is absent on the host) -- with a deterministic stand-in for the
per-rank forward.  Criterion: gathered == single-process output of the concatenated batch, bit for bit, with and
without micro-batching (incl. a ragged last chunk), over several steps of the rotating output buffers; eval mode: the
all-reduced error sums == the single-process sums."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _fake_forward(x):                       # per-sample, batch-independent: like the real path
    B, J = x.shape[:2]
    base = x.reshape(B, -1).sum(1)
    verts = base[:, None, None] + torch.arange(6890 * 3, dtype=torch.float32).reshape(1, 6890, 3) * 1e-3
    return verts, x.repeat(1, 1, 2)[:, :, :3].contiguous()


class _FakeModel:                           # the module surface ShardedForward touches: __call__(x, out=) and set_encoder
    def __init__(self, supports_out=True):
        self.encoder_calls = []
        self.supports_out = supports_out     # like gator_amd.models.GATOR: the forward writes into the caller's buffers
        self.out_ptrs = []

    def __call__(self, x, out=None):
        v, p = _fake_forward(x)
        if out is not None:
            assert out[0].is_contiguous() and out[1].is_contiguous() and out[0].shape == v.shape and out[1].shape == p.shape
            out[0].copy_(v)
            out[1].copy_(p)
            self.out_ptrs.append(out[0].data_ptr())
            v, p = out
        if self.report_next:                 # like gator_forward_f32 after an earlier bad forward with on_device_status = 'raise': the work is
            self.report_next = False         # queued, THEN the report is raised with the call's outputs attached (gator_amd/models/_base.py: _run)
            from gator_amd._lib import DeferredDeviceStatus
            ex = DeferredDeviceStatus('gator_forward_f32 failed (-8): an earlier forward on this ctx produced non-finite ...')
            ex.reason, ex.outputs = 2, (v, p)
            raise ex
        return v, p

    report_next = False

    def set_encoder(self, mode):
        self.encoder_calls.append(mode)


def _fake_metrics(verts, pose3d, target, sl):      # per-sample partial sums, like gator_amd.eval on the device
    t = target[sl]
    return torch.stack([(verts[:, :17].double() - t.double()).abs().sum(), pose3d.double().sum(),
                        torch.tensor(float(verts.shape[0]), dtype=torch.float64)])


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, micro, mode, q, in_place=True):
    os.environ['MASTER_ADDR'], os.environ['MASTER_PORT'] = '127.0.0.1', str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from gator_amd.parallel import ShardedForward
    n = 6
    ok = True
    if mode == 'deferred':                   # one rank's forward reports an earlier call's device status: nobody may hang in the collective
        from gator_amd._lib import DeferredDeviceStatus
        model = _FakeModel(in_place)
        run = ShardedForward(model, world, rank, dist, micro_batch=micro)
        for step in range(3):
            torch.manual_seed(step)
            full = torch.randn(world * n, 17, 2)
            model.report_next = rank == 1 and step == 1
            raised = False
            try:
                gv, gp = run.step(full[rank * n:(rank + 1) * n])
            except DeferredDeviceStatus as ex:
                raised = ex.code == -8
                gv, gp = run._last[1][2], run._last[1][3]
            rv, rp = _fake_forward(full)
            ok = ok and raised == (rank == 1 and step == 1)
            ok = ok and bool(torch.equal(gv, rv)) and bool(torch.equal(gp, rp))     # every rank, incl. the reporting one, holds the full gather
    elif mode == 'deferred_eval':            # the same in evaluation mode: the reporting rank still joins the all-reduce
        from gator_amd._lib import DeferredDeviceStatus
        model = _FakeModel(False)
        torch.manual_seed(0)
        tgt = torch.randn(world * n, 17, 3)
        run = ShardedForward(model, world, rank, dist, micro_batch=micro, mode='eval', metrics_fn=_fake_metrics)
        run.set_eval(None, tgt[rank * n:(rank + 1) * n])
        for step in range(3):
            torch.manual_seed(step + 1)
            full = torch.randn(world * n, 17, 2)
            model.report_next = rank == 1 and step == 1
            raised = False
            try:
                got = run.step(full[rank * n:(rank + 1) * n])
            except DeferredDeviceStatus:
                raised = True
                got = run._last[1][0]
            v, p = _fake_forward(full)
            want = _fake_metrics(v, p, tgt, slice(0, world * n))
            ok = ok and raised == (rank == 1 and step == 1)
            ok = ok and bool(torch.allclose(got, want, rtol=1e-12, atol=0))
    elif mode == 'gather':
        model = _FakeModel(in_place)
        run = ShardedForward(model, world, rank, dist, micro_batch=micro)
        kept = []
        for step in range(5):                # five steps: the two output buffers rotate; step k's result survives step k+1
            torch.manual_seed(step)
            full = torch.randn(world * n, 17, 2)
            gv, gp = run.step(full[rank * n:(rank + 1) * n])
            rv, rp = _fake_forward(full)
            ok = ok and bool(torch.equal(gv, rv)) and bool(torch.equal(gp, rp))
            if in_place and micro is None:   # the forward wrote this rank's slice of the gather buffer itself: no second copy of the shard
                ok = ok and model.out_ptrs[-1] == gv[rank * n:].data_ptr()
            elif in_place:                   # micro-batches land in this rank's slot of a [world, n, ...] staging tile
                ok = ok and len(model.out_ptrs) == 2 * (step + 1) and all(
                    any(t[0][rank].data_ptr() == ptr for t in run._stage.values()) for ptr in model.out_ptrs[-2:])
            if kept:
                ok = ok and bool(torch.equal(kept[-1][0], kept[-1][1]))       # previous step's buffer not overwritten yet
                ok = ok and kept[-1][0].data_ptr() != gv.data_ptr()
            kept.append((gv, rv))
        run.comm_only()
        ok = ok and bool(torch.equal(gv, rv))
        ok = ok and 1 <= run.max_inflight <= run.depth          # the backlog of collectives is bounded by the buffer rotation
        ok = ok and model.encoder_calls == ['sample']            # world > 1: ONE encoder for every call (small local batch -> per-sample kernel)
    else:
        torch.manual_seed(0)
        full = torch.randn(world * n, 17, 2)
        tgt = torch.randn(world * n, 17, 3)
        run = ShardedForward(_fake_forward, world, rank, dist, micro_batch=micro, mode='eval', metrics_fn=_fake_metrics)
        run.set_eval(None, tgt[rank * n:(rank + 1) * n])
        got = run.step(full[rank * n:(rank + 1) * n])
        v, p = _fake_forward(full)
        want = _fake_metrics(v, p, tgt, slice(0, world * n))
        ok = bool(torch.allclose(got, want, rtol=1e-12, atol=0)) and float(got[2]) == world * n
    q.put((rank, ok))
    dist.barrier()
    dist.destroy_process_group()


def _run(micro, mode='gather', in_place=True):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, micro, mode, q, in_place)) for r in range(2)]
    for p in ps:
        p.start()
    res = [q.get(timeout=120) for _ in ps]
    for p in ps:
        p.join(60)
    assert sorted(r[0] for r in res) == [0, 1]
    assert all(r[1] for r in res), res


def test_allgather_full_batch():
    _run(None)


def test_allgather_microbatched_ragged():
    _run(4)                                  # 6 samples per rank in chunks of 4 + 2


def test_deferred_device_status_does_not_strand_the_other_ranks():
    _run(None, 'deferred')
    _run(4, 'deferred')


def test_deferred_device_status_model_without_out_argument_and_eval_mode():
    _run(4, 'deferred', in_place=False)      # every branch of step() keeps the reporting rank in the collective
    _run(4, 'deferred_eval')


def test_allgather_model_without_out_argument():
    _run(None, in_place=False)               # a model that returns its own tensors: the collective copies the shard in
    _run(4, in_place=False)


def test_eval_mode_allreduce_only():
    _run(4, 'eval')


def test_single_rank_passthrough():
    from gator_amd.parallel import ShardedForward
    x = torch.randn(3, 17, 2)
    m = _FakeModel()
    v, p = ShardedForward(m, 1, 0, None).step(x)
    assert torch.equal(v, _fake_forward(x)[0])
    assert m.encoder_calls == []             # one rank: the library's own per-call choice stays


def test_encoder_pin_follows_the_local_batch():
    """world > 1: the pinned kernel is the one the library would use for the calls this run makes (local batch, or the micro-batch),
    decided once; an explicit choice wins."""
    from gator_amd.parallel import ShardedForward
    for kw, B, want in (({}, 1025, 'tiled'), ({}, 1024, 'sample'), ({'micro_batch': 256}, 2048, 'sample'), ({'encoder': 'sample'}, 4096, 'sample')):
        m = _FakeModel()
        run = ShardedForward(m, 8, 3, object(), **kw)
        run._pin_encoder(min(B, run.micro or B))
        run._pin_encoder(7)
        assert m.encoder_calls == [want], (kw, B, m.encoder_calls)
