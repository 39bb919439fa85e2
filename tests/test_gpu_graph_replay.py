"""hipGraph replay inside the library (gator_set_graph_replay, fused_api.hip: fused_forward_graph): the second time the same forward
(batch, input / output tensors) is seen it is captured, from then on it is one graph launch -- the same kernels, so the same bits."""
import pytest
import torch

from gator_amd import synthetic
from tests.helpers import build_model

pytestmark = pytest.mark.gpu


def _x(B, J, seed):
    return torch.from_numpy(synthetic.synthetic_pose2d(B, J, seed=seed)).cuda()


@pytest.mark.timeout(300)
@pytest.mark.parametrize('name,J', [('h36m17_bn', 17), ('coco19_alpha', 19)])
def test_graph_replay_is_bitwise_and_follows_the_input(name, J):
    z, ref = build_model(name, 'fused')
    z, m = build_model(name, 'fused')
    m.set_graph_replay(True)                               # before the first forward: applied when the context is created
    B = 256
    x = _x(B, J, 1)
    out = (torch.empty(B, 6890, 3, device='cuda'), torch.empty(B, J, 3, device='cuda'))
    want = ref(x)
    for it in range(4):                                    # direct, capture + launch, replay, replay
        v, p = m(x, out=out)
        torch.cuda.synchronize()
        assert v.data_ptr() == out[0].data_ptr()
        assert torch.equal(v, want[0]) and torch.equal(p, want[1]), it
    assert m.graph_launches() == 3
    x2 = _x(B, J, 2)
    x.copy_(x2)                                            # same tensors, new contents: the replay reads them
    want2 = ref(x2)
    v, p = m(x, out=out)
    torch.cuda.synchronize()
    assert torch.equal(v, want2[0]) and torch.equal(p, want2[1])
    assert m.graph_launches() == 4
    m.device_status()


@pytest.mark.timeout(600)
def test_graph_replay_slots_eviction_and_switching_off():
    z, ref = build_model('h36m17_bn', 'fused')
    z, m = build_model('h36m17_bn', 'fused')
    m(_x(4, 17, 0))                                        # context exists: the switch goes to the library at once
    assert m.set_graph_replay(True) == 0
    m(_x(1100, 17, 9))                                     # the workspace has its final size (a re-allocation retires every captured forward)
    cases = {}
    for B in (3, 64, 130, 256, 300, 1100):                 # both encoders; persistent and four-launch MDR; fewer keys than slots
        x = _x(B, 17, B)
        out = (torch.empty(B, 6890, 3, device='cuda'), torch.empty(B, 17, 3, device='cuda'))
        cases[B] = (x, out, ref(x))
    for rnd in range(3):                                   # direct, capture + launch, replay
        for B, (x, out, want) in cases.items():
            v, p = m(x, out=out)
            torch.cuda.synchronize()
            assert torch.equal(v, want[0]) and torch.equal(p, want[1]), (rnd, B)
    assert m.graph_launches() == 2 * len(cases)
    for B in range(10, 19):                                # eighteen more keys (fresh output tensors each call) than the eight slots hold: keys that were
                                                           # never captured are evicted first (no device-wide wait), the six captured forwards stay
        xs = _x(B, 17, B)
        for it in range(2):
            v, p = m(xs, out=(torch.empty(B, 6890, 3, device='cuda'), torch.empty(B, 17, 3, device='cuda')))
    x, out, want = cases[256]
    n = m.graph_launches()
    for it in range(3):                                    # seen again: its capture survived, three replays
        v, p = m(x, out=out)
        torch.cuda.synchronize()
        assert torch.equal(v, want[0]) and torch.equal(p, want[1])
    assert m.graph_launches() == n + 3
    for B in range(20, 29):                                # nine more CAPTURED keys: now the oldest captures are dropped
        xs, outs = _x(B, 17, B), (torch.empty(B, 6890, 3, device='cuda'), torch.empty(B, 17, 3, device='cuda'))
        for it in range(2):
            m(xs, out=outs)
    n = m.graph_launches()
    for it in range(3):                                    # direct, capture + launch, replay
        v, p = m(x, out=out)
        torch.cuda.synchronize()
        assert torch.equal(v, want[0]) and torch.equal(p, want[1])
    assert m.graph_launches() == n + 2
    # a forward without out= allocates fresh outputs: whatever the allocator hands back, the result is right
    for it in range(4):
        v, p = m(x)
        assert torch.equal(v, want[0])
    # taps and profiling bypass the graph
    m.enable_block_taps(True)
    n = m.graph_launches()
    v, p = m(x, out=out)
    assert torch.equal(v, want[0]) and m.graph_launches() == n
    m.enable_block_taps(False)
    m.set_graph_replay(False)
    n = m.graph_launches()
    v, p = m(x, out=out)
    torch.cuda.synchronize()
    assert torch.equal(v, want[0]) and m.graph_launches() == n


@pytest.mark.timeout(300)
def test_graph_replay_on_a_side_stream():
    z, ref = build_model('h36m17_bn', 'fused')
    z, m = build_model('h36m17_bn', 'fused')
    m.set_graph_replay(True)
    B = 256
    x = _x(B, 17, 5)
    want = ref(x)
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    out = (torch.empty(B, 6890, 3, device='cuda'), torch.empty(B, 17, 3, device='cuda'))
    with torch.cuda.stream(s):
        for it in range(4):
            v, p = m(x, out=out)
        s.synchronize()
    assert torch.equal(v, want[0]) and torch.equal(p, want[1])
    assert m.graph_launches() == 3


@pytest.mark.timeout(300)
def test_graph_replay_steps_aside_for_a_caller_that_captures():
    """torch.cuda.CUDAGraph around a module whose library-level replay is on: the forward is captured by the caller as plain launches."""
    z, ref = build_model('h36m17_bn', 'fused')
    z, m = build_model('h36m17_bn', 'fused')
    m.set_graph_replay(True)
    B = 64
    x = _x(B, 17, 7)
    want = ref(x)
    out = (torch.empty(B, 6890, 3, device='cuda'), torch.empty(B, 17, 3, device='cuda'))
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        m(x, out=out)
        m(x, out=out)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    n = m.graph_launches()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        m(x, out=out)
    out[0].zero_()
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(out[0], want[0]) and torch.equal(out[1], want[1])
    assert m.graph_launches() == n                         # nothing was replayed by the library inside the caller's capture
