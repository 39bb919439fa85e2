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


@pytest.mark.timeout(600)
def test_unserved_queue_is_loud_and_the_ctx_falls_back(monkeypatch):
    """A persistent launch whose grid leaves XCDs EMPTY (3 workgroups for 8 queues: what a CU mask or reserved CUs would do): the samples
    of an unserved queue are never computed.  That must not be silent (round-3 advice): their vertices are NaN, the next call on the
    ctx reports GATOR_EDEVICE once, and from then on the ctx runs the four-launch form -- bitwise the right results."""
    monkeypatch.setenv('GATOR_MDR_PERSIST', '0')
    z, ref = build_model('h36m17_bn', 'fused')
    x = torch.from_numpy(synthetic.synthetic_pose2d(40, 17, seed=40)).cuda()
    want_v, want_p = ref(x)
    monkeypatch.setenv('GATOR_MDR_PERSIST', '1')
    monkeypatch.setenv('GATOR_MDR_PERSIST_GRID', '3')
    z, m = build_model('h36m17_bn', 'fused')
    m.on_device_status = 'raise'
    v, p = m(x)
    torch.cuda.synchronize()
    bad = ~torch.isfinite(v).reshape(40, -1).all(1)
    assert bad.any() and not bad.all()                      # at most three of the eight queues were served
    assert torch.equal(v[~bad], want_v[~bad])               # what WAS computed is right
    with pytest.raises(RuntimeError, match='persistent MDR launch'):
        m.device_status()
    v2, p2 = m(x)                                           # the ctx fell back to four launches
    torch.cuda.synchronize()
    m.device_status()
    assert torch.equal(v2, want_v) and torch.equal(p2, want_p)


@pytest.mark.timeout(300)
@pytest.mark.parametrize('name,J,B', [('h36m17_bn', 17, 37), ('coco19_alpha', 19, 256)])
def test_head_from_partial_sums_agrees_with_the_whole_head(monkeypatch, name, J, B):
    """Round 6: the head's Conv1d leaves the stage-3 tiles as partial sums (double, fixed association) and k_mdr_head_finish adds the 14 of a sample;
    GATOR_MDR_HEAD_PARTIALS=0 is the whole head in k_mdr_head (double sums over the sample in another order).  Same products, both exact in double:
    the two forms may differ in the last place of the fp32 results only."""
    x = torch.from_numpy(synthetic.synthetic_pose2d(B, J, seed=5)).cuda()
    monkeypatch.setenv('GATOR_MDR_HEAD_PARTIALS', '0')
    z, m0 = build_model(name, 'fused')
    v0, p0 = m0(x)
    monkeypatch.delenv('GATOR_MDR_HEAD_PARTIALS')
    z, m1 = build_model(name, 'fused')
    v1, p1 = m1(x)
    torch.cuda.synchronize()
    assert torch.equal(p0, p1)
    assert torch.isfinite(v1).all()
    assert float((v0 - v1).abs().max()) <= 5e-7            # metres: vertices are O(1) m, one fp32 ulp is 1.2e-7 (measured: 2.4e-7, 64 % of the values bit-equal); the bar of the path is 1e-6 m
    assert not torch.equal(v0, v1)                          # the switch is read at gator_create: if it were ignored the comparison above would be vacuous (drop this line should the forms ever agree bit for bit)
