"""The four MDR stages as one persistent launch (k_mdr_persist, mdr_fused.hip) against the four per-stage launches.

Both forms run the same tile body, so they must agree BIT FOR BIT at every batch size -- including the ones where the library's own
rule would not pick the persistent form -- and the persistent form must not depend on how many samples share an XCD's queue."""
import numpy as np
import pytest
import torch

from gator_amd import synthetic
from tests.helpers import build_model

pytestmark = pytest.mark.gpu


def _forward(monkeypatch, mode, name, x):
    if mode is None:
        monkeypatch.delenv('GATOR_MDR_PERSIST', raising=False)
    else:
        monkeypatch.setenv('GATOR_MDR_PERSIST', mode)
    z, m = build_model(name, 'fused')              # a fresh module -> a fresh context, which reads the switch
    v, p = m(x)
    v2, p2 = m(x)                                  # second call: tickets and completion counts must have been reset
    torch.cuda.synchronize()
    assert torch.equal(v, v2) and torch.equal(p, p2)
    return v, p, m


@pytest.mark.timeout(300)
@pytest.mark.parametrize('name,J,B', [('h36m17_bn', 17, 1), ('h36m17_bn', 17, 7), ('h36m17_bn', 17, 100), ('h36m17_bn', 17, 256),
                                      ('coco19_alpha', 19, 300), ('h36m17_bn', 17, 1100)])
def test_persistent_launch_is_bitwise_the_four_launches(monkeypatch, name, J, B):
    x = torch.from_numpy(synthetic.synthetic_pose2d(B, J, seed=11)).cuda()
    v0, p0, _ = _forward(monkeypatch, '0', name, x)
    v1, p1, m1 = _forward(monkeypatch, '1', name, x)
    va, pa, _ = _forward(monkeypatch, None, name, x)
    assert torch.isfinite(v1).all()
    assert torch.equal(v0, v1) and torch.equal(p0, p1)
    assert torch.equal(v0, va) and torch.equal(p0, pa)
    # a sample's result does not depend on the batch (and so on the queue) it went through; from 1024 samples on the ENCODER is
    # chosen by batch size (tests/test_gpu_tiled.py), so the check stops there
    if B >= 1024:
        return
    idx = sorted({0, B // 2, B - 1})
    vs, ps = m1(x[idx].contiguous())
    assert torch.equal(vs, v1[idx]) and torch.equal(ps, p1[idx])


@pytest.mark.timeout(300)
def test_persistent_stage_entry_point_and_changing_batches(monkeypatch):
    """gator_mdr_forward (pose_combine in, k_mdr_joint resets the counters) and a context whose batch size changes from call to call."""
    monkeypatch.setenv('GATOR_MDR_PERSIST', '1')
    z, m = build_model('h36m17_bn', 'fused')
    monkeypatch.setenv('GATOR_MDR_PERSIST', '0')
    z0, m0 = build_model('h36m17_bn', 'fused')
    g = torch.Generator().manual_seed(3)
    for B in (40, 3, 260, 40):
        pc = torch.randn(B, 17, 133, generator=g).cuda()
        assert torch.equal(m.pose2mesh(pc), m0.pose2mesh(pc)), B
        x = torch.from_numpy(synthetic.synthetic_pose2d(B, 17, seed=B)).cuda()
        a, b = m(x), m0(x)
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]), B


@pytest.mark.timeout(300)
def test_persistent_launch_survives_random_batch_sizes(monkeypatch):
    """400 back-to-back forwards with the persistent launch forced on and batch sizes drawn from 1..699 (every queue shape: empty
    XCD queues, ragged last tickets, counters re-zeroed by the joint-token kernel for a different B each time).  A tripped poll
    budget would surface as NaN vertices (k_mdr_head poisons its output), a broken dependency as a mismatch with the four launches."""
    monkeypatch.setenv('GATOR_MDR_PERSIST', '1')
    z, m = build_model('h36m17_bn', 'fused')
    monkeypatch.setenv('GATOR_MDR_PERSIST', '0')
    z0, m0 = build_model('h36m17_bn', 'fused')
    g = torch.Generator().manual_seed(0)
    xs = {}
    for it in range(400):
        B = int(torch.randint(1, 700, (1,), generator=g))
        if B not in xs:
            xs[B] = torch.from_numpy(synthetic.synthetic_pose2d(B, 17, seed=B)).cuda()
        v, p = m(xs[B])
        if it % 40 == 0:
            v0, p0 = m0(xs[B])
            assert torch.isfinite(v).all() and torch.equal(v, v0) and torch.equal(p, p0), (it, B)
    torch.cuda.synchronize()
    assert torch.isfinite(v).all()


_ADOPT_CHILD = r"""
import sys, torch
sys.path.insert(0, '.')
from gator_amd import synthetic
from tests.helpers import build_model
z, m = build_model('h36m17_bn', 'fused')
for B in (3, 40, 97):
    x = torch.from_numpy(synthetic.synthetic_pose2d(B, 17, seed=B)).cuda()
    v, p = m(x)
    torch.cuda.synchronize()
    m.device_status()
    torch.save((v.cpu(), p.cpu()), sys.argv[1] + '.%d' % B)
print('ok')
"""


@pytest.mark.timeout(600)
def test_queues_of_xcds_without_a_workgroup_are_adopted(tmp_path):
    """A persistent launch whose grid leaves XCDs EMPTY (1, 3, 5 or 13 workgroups for 8 queues: what a CU mask or reserved CUs would do):
    every queue is still drained -- by the one XCD that claims it -- and the result is bitwise the four launches'.  Before the
    ownership protocol the samples of an empty XCD were silently never computed."""
    import os, subprocess, sys
    outs = {}
    for tag, env in (('ref', {'GATOR_MDR_PERSIST': '0'}), ('g1', {'GATOR_MDR_PERSIST': '1', 'GATOR_MDR_PERSIST_GRID': '1'}),
                     ('g3', {'GATOR_MDR_PERSIST': '1', 'GATOR_MDR_PERSIST_GRID': '3'}), ('g5', {'GATOR_MDR_PERSIST': '1', 'GATOR_MDR_PERSIST_GRID': '5'}),
                     ('g13', {'GATOR_MDR_PERSIST': '1', 'GATOR_MDR_PERSIST_GRID': '13'})):
        path = str(tmp_path / tag)
        r = subprocess.run([sys.executable, '-c', _ADOPT_CHILD, path], env=dict(os.environ, **env), capture_output=True, text=True, timeout=500)
        assert r.returncode == 0 and 'ok' in r.stdout, (tag, r.stdout[-500:], r.stderr[-2000:])
        outs[tag] = {B: torch.load(path + '.%d' % B) for B in (3, 40, 97)}
    for tag in ('g1', 'g3', 'g5', 'g13'):
        for B in (3, 40, 97):
            assert torch.isfinite(outs[tag][B][0]).all(), (tag, B)
            assert torch.equal(outs[tag][B][0], outs['ref'][B][0]) and torch.equal(outs[tag][B][1], outs['ref'][B][1]), (tag, B)
