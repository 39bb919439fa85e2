"""The device side of gator_amd.parallel on the real backend: a ONE-rank RCCL ("nccl") process group on the GPU box runs
the same side-stream all_gather_into_tensor / micro-batched all_gather code as N ranks do (the N>1 sharding logic itself is
covered on CPU by test_parallel_gloo.py).  Gathered output must equal the plain forward bit for bit."""
import os

import numpy as np
import pytest
import torch

from gator_amd import synthetic
from gator_amd.parallel import ShardedForward
from tests.helpers import build_model

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def rccl_group():
    import torch.distributed as dist
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    dist.init_process_group('nccl', init_method='tcp://127.0.0.1:29653', rank=0, world_size=1,
                            device_id=torch.device('cuda', 0))
    yield dist
    dist.destroy_process_group()


@pytest.mark.parametrize('micro', [None, 16])
def test_single_rank_rccl_gather_equals_forward(rccl_group, micro):
    z, m = build_model('h36m17_bn', 'fused')
    x = torch.from_numpy(synthetic.synthetic_pose2d(40, 17, seed=3)).cuda()
    ref_v, ref_p = m(x)
    run = ShardedForward(m, 1, 0, rccl_group, micro_batch=micro, always_gather=True)
    for _ in range(3):                      # exercises the double-buffered outputs and the un-awaited side stream
        gv, gp = run.step(x)
    run.wait()
    torch.cuda.synchronize()
    assert torch.equal(gv, ref_v) and torch.equal(gp, ref_p)
