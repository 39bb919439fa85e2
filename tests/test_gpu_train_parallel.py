"""Data-parallel training (SURVEY 8e applied to the training row): every rank steps on its shard, ONE all-reduce of the flat
gradient buffer per step (RCCL = torch.distributed "nccl"), then the same Adam launch on every rank.
* one rank (always runs): the all-reduce path is the identity - weights equal the single-process step bit for bit;
* two ranks (when >= 2 GPUs are visible): two spawned processes on half batches with dropout off end with weights equal, to fp32
  rounding, to one process stepping on the whole batch (mean of the shard gradients = gradient of the mean loss; BatchNorm-free
  variant so that no statistic couples the samples)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

from gator_amd import synthetic
from gator_amd.train import model as M
from tests.test_gpu_train_step import batch_of, make_trainer

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_single_rank_allreduce_step_equals_plain_step():
    import torch.distributed as dist
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    own = not dist.is_initialized()
    if own:
        dist.init_process_group('nccl', init_method='tcp://127.0.0.1:%d' % _free_port(), rank=0, world_size=1, device_id=torch.device('cuda', 0))
    try:
        z, m, tr_a, _ = make_trainer('coco19_alpha', rates=M.Rates(), seed=9)
        z, m, tr_b, _ = make_trainer('coco19_alpha', rates=M.Rates(), seed=9, dist=dist)
        x, tg = batch_of(z, 6, shift=5)
        for _ in range(2):
            la, _ = tr_a.step(x, tg)
            lb, _ = tr_b.step(x, tg)
        assert float(la) == float(lb)
        assert torch.equal(tr_a.params.flat.detach(), tr_b.params.flat.detach())
    finally:
        if own:
            dist.destroy_process_group()


_TWO_RANK = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, %(root)r)
rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
torch.cuda.set_device(rank)
dev = torch.device('cuda', rank)
dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
import numpy as np
from gator_amd import synthetic
from gator_amd.train import model as M
from gator_amd.train.trainer import Trainer
from tests.helpers import build_model, load_golden
name = 'coco19_alpha'                                     # LayerNorm head: no batch statistic couples the samples
z = load_golden('train_' + name)
zz, m = build_model(name, 'fused', device=dev)
seed, J, n = int(z['seed']), 19, 4
base = synthetic.make_base_data(seed)
jreg = synthetic.load_j_regressors()['h36m'].astype(np.float32)
faces = synthetic.synthetic_faces(seed)
x = torch.from_numpy(synthetic.synthetic_pose2d(world * n, J, 77)).to(dev)
tg = {k: torch.from_numpy(v).to(dev) for k, v in synthetic.training_targets(world * n, J, base, jreg, 77).items()}
for k in ('mesh_valid', 'lift_pose3d_valid'):
    tg[k].fill_(1.0)                                      # equal-sized shards with equal masks: mean of shard losses = whole-batch loss
whole = Trainer.from_module(m, faces, jreg, device=dev, rates=M.Rates(0.0), lr=1e-5)
shard = Trainer.from_module(m, faces, jreg, device=dev, rates=M.Rates(0.0), lr=1e-5, dist=dist)
whole.epoch = shard.epoch = 16
sl = slice(rank * n, (rank + 1) * n)
for it in range(2):
    whole.step(x, tg)
    shard.step(x[sl], {k: v[sl] for k, v in tg.items()})
d = float((whole.params.flat.detach() - shard.params.flat.detach()).abs().max())
moved = float((whole.params.flat.detach() - Trainer.from_module(m, faces, jreg, device=dev).params.flat.detach()).abs().max())
flat = shard.params.flat.detach().clone()
other = [torch.empty_like(flat) for _ in range(world)]
dist.all_gather(other, flat)
same = all(bool(torch.equal(o, flat)) for o in other)     # replicas stay bit-identical
print('RESULT', rank, d, moved, same, flush=True)
dist.destroy_process_group()
sys.exit(0 if (d <= 0.05 * moved and same and moved > 0) else 1)
'''


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason='needs two GPUs (the gpurun box has one)')
def test_two_rank_data_parallel_training_matches_whole_batch(tmp_path):
    script = tmp_path / 'two_rank_train.py'
    script.write_text(_TWO_RANK % {'root': ROOT})
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
                        '--master-port', str(_free_port()), str(script)], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
