"""The device side of gator_amd.parallel on the real backend (RCCL = torch.distributed "nccl").

* one rank (always runs on the 1-GPU box): the side-stream all_gather_into_tensor / micro-batched staging path and the
  eval-mode all-reduce, against the plain forward;
* two ranks (runs when the box has >= 2 GPUs, skipped otherwise): two spawned processes, one per GPU; every rank's gathered
  tensors must equal the single-process forward of the concatenated batch BIT FOR BIT, over several pipelined steps of the
  rotating output buffers; and `bench.py --gpus 2`, started directly, must launch its own ranks and report n_gpus == 2."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

from gator_amd import synthetic
from gator_amd.parallel import ShardedForward
from tests.helpers import build_model

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.fixture(scope='module')
def rccl_group():
    import torch.distributed as dist
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    dist.init_process_group('nccl', init_method='tcp://127.0.0.1:%d' % _free_port(), rank=0, world_size=1,
                            device_id=torch.device('cuda', 0))
    yield dist
    dist.destroy_process_group()


@pytest.mark.parametrize('micro', [None, 16])
def test_single_rank_rccl_gather_equals_forward(rccl_group, micro):
    z, m = build_model('h36m17_bn', 'fused')
    x = torch.from_numpy(synthetic.synthetic_pose2d(40, 17, seed=3)).cuda()
    ref_v, ref_p = m(x)
    run = ShardedForward(m, 1, 0, rccl_group, micro_batch=micro, always_gather=True)
    for _ in range(3):                      # exercises the rotating outputs and the un-awaited side stream
        gv, gp = run.step(x)
    run.wait()
    torch.cuda.synchronize()
    assert torch.equal(gv, ref_v) and torch.equal(gp, ref_p)
    run.comm_only()
    run.wait()
    torch.cuda.synchronize()
    assert torch.equal(gv, ref_v)


def test_single_rank_eval_mode_matches_eval_module(rccl_group):
    from gator_amd import eval as ev
    z, m = build_model('coco19_alpha', 'fused')
    x = torch.from_numpy(synthetic.synthetic_pose2d(48, 19, seed=4)).cuda()
    tgt = torch.from_numpy(np.random.RandomState(5).randn(48, 17, 3).astype(np.float32) * 150).cuda()
    jr = synthetic.load_j_regressors()['h36m']
    run = ShardedForward(m, 1, 0, rccl_group, micro_batch=20, always_gather=True, mode='eval')
    run.set_eval(jr, tgt)
    got = run.step(x)
    run.wait()
    verts, _ = m(x)
    joints = ev.JointRegressor(jr, x.device)(verts) * 1000.0
    want = torch.stack([ev.mpjpe(joints, tgt) * 48, ev.pa_mpjpe(joints, tgt) * 48]).double()
    torch.cuda.synchronize()
    assert float(got[2]) == 48
    assert torch.allclose(got[:2], want, rtol=1e-5), (got, want)


def test_native_rccl_allgather_one_rank():
    """gator_allgather_verts (RCCL resolved at run time from the process, no torch.distributed): one rank = a copy."""
    from gator_amd.comm import NativeComm
    comm = NativeComm.create(0, 1, lambda b: b)
    v = torch.randn(5, 6890, 3, device='cuda')
    p = torch.randn(5, 17, 3, device='cuda')
    gv, gp = comm.allgather(v, p)
    torch.cuda.synchronize()
    assert torch.equal(gv, v) and torch.equal(gp, p)
    gv2 = comm.allgather(v)
    torch.cuda.synchronize()
    assert torch.equal(gv2, v)
    comm.close()


_TWO_RANK = r'''
import os, sys, json, torch, torch.distributed as dist
sys.path.insert(0, %(root)r)
rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
torch.cuda.set_device(rank)
dev = torch.device('cuda', rank)
dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
from gator_amd import synthetic
from gator_amd.parallel import ShardedForward
from tests.helpers import build_model
z, m = build_model('h36m17_bn', 'fused', device=dev)
n = 24
ok = True
for micro in (None, 10):
    run = ShardedForward(m, world, rank, dist, micro_batch=micro)
    prev = None
    for step in range(3):
        full = torch.from_numpy(synthetic.synthetic_pose2d(world * n, 17, seed=50 + step)).to(dev)
        gv, gp = run.step(full[rank * n:(rank + 1) * n])        # gather of this step overlaps the reference forward below
        rv, rp = m(full)                                        # single-process forward of the concatenated batch
        if prev is not None:                                    # the previous step's buffers are still intact
            ok = ok and bool(torch.equal(prev[0], prev[1]))
        run.wait()
        ok = ok and bool(torch.equal(gv, rv)) and bool(torch.equal(gp, rp))
        prev = (gv, rv)
    ok = ok and 1 <= run.max_inflight <= run.depth              # bounded backlog of collectives
    ok = ok and run._pinned == 'sample'                         # one encoder kernel for every call of a multi-rank run
# evaluation mode (BASELINE config 5): on-device joint regression + error sums, ONE all-reduce of three doubles
import numpy as np
jr = synthetic.load_j_regressors()['h36m']
full = torch.from_numpy(synthetic.synthetic_pose2d(world * n, 17, seed=77)).to(dev)
tgt = torch.from_numpy(np.random.RandomState(5).randn(world * n, 17, 3).astype(np.float32) * 150).to(dev)
erun = ShardedForward(m, world, rank, dist, micro_batch=10, mode='eval')
erun.set_eval(jr, tgt[rank * n:(rank + 1) * n])
for step in range(3):
    got = erun.step(full[rank * n:(rank + 1) * n])
erun.wait()
single = ShardedForward(m, 1, 0, None, mode='eval')
single.set_eval(jr, tgt)
want = single.step(full)
torch.cuda.synchronize()
ok = ok and float(got[2]) == world * n and bool(torch.allclose(got, want, rtol=1e-9, atol=0))
ok = ok and erun.max_inflight <= erun.depth
# the same gather through the C ABI's own RCCL communicator (gator_allgather_verts); the id travels over torch's object broadcast
from gator_amd.comm import NativeComm
def _bcast(b):
    box = [b]
    dist.broadcast_object_list(box, src=0)
    return box[0]
comm = NativeComm.create(rank, world, _bcast)
full = torch.from_numpy(synthetic.synthetic_pose2d(world * n, 17, seed=99)).to(dev)
lv, lp = m(full[rank * n:(rank + 1) * n])
nv, npp = comm.allgather(lv, lp)
rv, rp = m(full)
torch.cuda.synchronize()
ok = ok and bool(torch.equal(nv, rv)) and bool(torch.equal(npp, rp))
comm.close()
torch.cuda.synchronize()
flag = torch.tensor([1.0 if ok else 0.0], device=dev)
dist.all_reduce(flag, op=dist.ReduceOp.MIN)
if rank == 0:
    print(json.dumps({'ok': bool(flag.item() == 1.0), 'world': world}))
dist.barrier()
dist.destroy_process_group()
'''


def _need_two():
    if torch.cuda.device_count() < 2:
        pytest.skip('needs >= 2 GPUs (the gpurun box has one; covered on CPU by tests/test_parallel_gloo.py)')


def test_two_rank_rccl_gather_is_bitwise_single_process():
    _need_two()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    # torch.distributed.run wants a script path: write the rank program next to the test outputs
    path = os.path.join(ROOT, 'gpurun_out', '_two_rank_prog.py')
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, 'w') as f:
        f.write(_TWO_RANK % {'root': ROOT})
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), path]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1]
    res = json.loads(line)
    assert res == {'ok': True, 'world': 2}, res


def test_bench_launches_its_own_ranks():
    _need_two()
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '1', '--blocks', '2',
                        '--batch', '64'], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1])
    assert line['n_gpus'] == 2 and line['config']['parallelism'] == 'dp2'
    assert 'comm' in line and line['comm']['collective_ms'] > 0
    # the BASELINE presets: config 4 (gather) and config 5 (evaluation mode, all-reduce only)
    for cfg, mode in (('4', 'gather'), ('5', 'eval')):
        r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--config', cfg, '--steps', '2', '--warmup', '1',
                            '--blocks', '2', '--batch', '96'], capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1])
        assert line['n_gpus'] == 2 and line['config']['mode'] == mode and line['config']['batch_per_gpu'] == 96
