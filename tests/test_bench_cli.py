"""bench.py's launch contract, without a GPU: `--gpus N` started directly spawns N ranks as a CHILD torch.distributed.run
(before anything touches the GPU); started under torch.distributed.run with a different WORLD_SIZE it refuses."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_gpus_must_match_world_size():
    env = dict(os.environ, RANK='0', LOCAL_RANK='0', WORLD_SIZE='2', MASTER_ADDR='127.0.0.1', MASTER_PORT='29999')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '4'], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 2 and 'WORLD_SIZE=2' in r.stderr


def test_direct_start_spawns_ranks(tmp_path):
    """On this GPU-less container the ranks fail at their `assert torch.cuda.is_available()`; what is checked is that the
    parent started N of them through torch.distributed.run and relayed a non-zero exit code, not an n_gpus=1 line."""
    env = dict(os.environ)
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE'):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0', '--blocks', '1'],
                       env=env, capture_output=True, text=True, timeout=300)
    import torch
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        assert r.returncode == 0 and '"n_gpus": 2' in r.stdout
    else:
        assert r.returncode != 0
        assert '"n_gpus": 1' not in r.stdout
        assert 'bench.py needs a HIP device' in r.stderr or 'ChildFailedError' in r.stderr or 'AssertionError' in r.stderr
