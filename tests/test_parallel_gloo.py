"""N>1 path on CPU: world_size-2 gloo run of the batch-sharding + vertex all-gather logic (gator_amd/parallel.py) with a
deterministic stand-in for the per-rank forward.  Criterion: gathered == single-process output of the
concatenated batch, bit for bit, with and without micro-batching."""
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


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, micro, q):
    os.environ['MASTER_ADDR'], os.environ['MASTER_PORT'] = '127.0.0.1', str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from gator_amd.parallel import ShardedForward
    torch.manual_seed(0)
    full = torch.randn(world * 6, 17, 2)
    shard = full[rank * 6:(rank + 1) * 6]
    gv, gp = ShardedForward(_fake_forward, world, rank, dist, micro_batch=micro).step(shard)
    rv, rp = _fake_forward(full)
    q.put((rank, bool(torch.equal(gv, rv)), bool(torch.equal(gp, rp))))
    dist.barrier()
    dist.destroy_process_group()


def _run(micro):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, micro, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = [q.get(timeout=120) for _ in ps]
    for p in ps:
        p.join(60)
    assert sorted(r[0] for r in res) == [0, 1]
    assert all(r[1] and r[2] for r in res), res


def test_allgather_full_batch():
    _run(None)


def test_allgather_microbatched():
    _run(4)


def test_single_rank_passthrough():
    from gator_amd.parallel import ShardedForward
    x = torch.randn(3, 17, 2)
    v, p = ShardedForward(_fake_forward, 1, 0, None).step(x)
    assert torch.equal(v, _fake_forward(x)[0])
