"""Full-size checks at the bench configuration (BASELINE configs[1]: B=256, J=17, fp32) and at a sharded shape.

Size-independent properties of the path (samples are independent):
  * permutation equivariance and batch-size independence, BITWISE (same kernels, per-sample arithmetic does not depend on
    the batch size or on the position in the batch) -- this is also the correctness criterion of the multi-GPU all-gather:
    concatenated shard outputs == single-process output;
  * determinism (no atomics, fixed reduction order): two runs are bit-identical;
  * accuracy: max |verts - fp64 oracle| <= 1e-3 mm over all 256 x 6890 x 3 coordinates."""
import numpy as np
import pytest
import torch

from gator_amd import synthetic
from tests.helpers import build_model, oracle_setup

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def setup():
    z, m = build_model('h36m17_bn', 'fused')
    x = torch.from_numpy(synthetic.synthetic_pose2d(256, 17, seed=1000)).cuda()     # bench.py's rank-0 input
    v, p = m(x)
    torch.cuda.synchronize()
    return m, x, v.clone(), p.clone()


def test_b256_accuracy_vs_fp64_oracle(setup):
    from oracle import gator_oracle as go
    m, x, v, p = setup
    zz, c, sd = oracle_setup('h36m17_bn')
    ref, rp = go.gator_forward(sd, c, x.cpu(), torch.float64)
    err = np.abs(v.cpu().numpy().astype(np.float64) - ref.numpy()) * 1e3
    print('\n[B=256] max %.3e mm, mean %.3e mm over %d coordinates' % (err.max(), err.mean(), err.size))
    assert err.max() <= 1e-3
    assert np.abs(p.cpu().numpy() - rp.numpy()).max() <= 1e-3


@pytest.mark.timeout(900)
@pytest.mark.parametrize('name,J', [('h36m17_bn', 17), ('coco19_alpha', 19)])
def test_b4096_shipped_policy_every_coordinate_of_512_samples(name, J):
    """B = 4 096 under the SHIPPED policy (no switch, no pin: two full rounds of the sample-tiled encoder + the remainder on the
    one-sample-per-workgroup one), both variants: every coordinate of 512 oracle-checked samples -- 256 from the tiled rounds, 256
    from the remainder -- within 1e-3 mm of the fp64 oracle, and not noisier than the reference's own fp32 arithmetic on the same
    samples (the oracle in fp32: same ops, same order).  The statistics over 16k samples x 3 weight draws: profiles/r04_error_budget.md."""
    from oracle import gator_oracle as go
    z, m = build_model(name, 'fused')
    x = torch.from_numpy(synthetic.synthetic_pose2d(4096, J, seed=4096 + J)).cuda()
    v, p = m(x)
    torch.cuda.synchronize()
    m.device_status()
    assert torch.isfinite(v).all()
    idx = list(range(0, 256)) + list(range(4096 - 256, 4096))
    zz, c, sd = oracle_setup(name)
    xs = x[idx].cpu()
    errs, refs = [], []
    for lo in range(0, 512, 128):
        r64, rp = go.gator_forward(sd, c, xs[lo:lo + 128], torch.float64)
        r32, _ = go.gator_forward(sd, c, xs[lo:lo + 128], torch.float32)
        errs.append(np.abs(v[idx[lo:lo + 128]].cpu().numpy().astype(np.float64) - r64.numpy()) * 1e3)
        refs.append(np.abs(r32.numpy().astype(np.float64) - r64.numpy()) * 1e3)
        assert np.abs(p[idx[lo:lo + 128]].cpu().numpy() - rp.numpy()).max() <= 1e-3
    e, r = np.concatenate(errs), np.concatenate(refs)
    rms = lambda a: float(np.sqrt((a ** 2).mean()))
    print('\n[%s B=4096, 512 samples, %d coordinates] ours: max %.3e rms %.3e mm (tiled rounds %.3e, remainder %.3e) ; reference fp32 arithmetic: max %.3e rms %.3e mm'
          % (name, e.size, e.max(), rms(e), e[:256].max(), e[256:].max(), r.max(), rms(r)))
    assert e.max() <= 1e-3
    assert rms(e) <= rms(r) and e.max() <= 1.25 * r.max()


def test_deterministic_and_batch_independent(setup):
    m, x, v, p = setup
    v2, p2 = m(x)
    assert torch.equal(v2, v) and torch.equal(p2, p)                      # run-to-run bitwise
    perm = torch.randperm(256, generator=torch.Generator().manual_seed(0)).cuda()
    vp, pp = m(x[perm])
    assert torch.equal(vp, v[perm]) and torch.equal(pp, p[perm])          # permutation equivariance, bitwise
    for lo, hi in ((0, 1), (37, 38), (100, 133), (128, 256), (5, 250)):   # ragged sub-batches == slices of the full batch
        vs, ps = m(x[lo:hi].contiguous())
        assert torch.equal(vs, v[lo:hi]) and torch.equal(ps, p[lo:hi]), (lo, hi)


def test_shard_concat_equals_full(setup):
    """What the 8-GPU all-gather assembles (rank r owns rows r*B/N...) equals the single-GPU output bit for bit."""
    m, x, v, p = setup
    for n in (2, 4, 8):
        sh = 256 // n
        parts = [m(x[r * sh:(r + 1) * sh].contiguous()) for r in range(n)]
        assert torch.equal(torch.cat([a for a, _ in parts]), v)
        assert torch.equal(torch.cat([b for _, b in parts]), p)


def test_large_batch_b2048_j19(monkeypatch):
    """Config-3/4 shapes (B=2048 per GPU, J=19) on the one-sample-per-workgroup encoder (GATOR_GAT_TILED=0: bitwise batch
    invariance at any size): finite, bitwise-consistent with small batches, sampled accuracy.  The shipped large-batch policy
    (sample-tiled encoder) is covered by tests/test_gpu_tiled.py."""
    from oracle import gator_oracle as go
    monkeypatch.setenv('GATOR_GAT_TILED', '0')
    z, m = build_model('coco19_alpha', 'fused')
    zz, c, sd = oracle_setup('coco19_alpha')
    x = torch.from_numpy(synthetic.synthetic_pose2d(2048, 19, seed=77)).cuda()
    v, p = m(x)
    assert torch.isfinite(v).all()
    idx = [0, 1, 511, 1024, 2047]
    vs, ps = m(x[idx].contiguous())
    assert torch.equal(vs, v[idx]) and torch.equal(ps, p[idx])
    ref, _ = go.gator_forward(sd, c, x[idx].cpu(), torch.float64)
    assert np.abs(vs.cpu().numpy() - ref.numpy()).max() * 1e3 <= 1e-3


def test_subbatch_streams_bitwise(setup):
    """gator_config.subbatch_streams = 2 (two half-batches on two internal streams, forked from / joined to the caller's
    stream by events) returns exactly the single-stream result, for even and odd splits and when called repeatedly."""
    m, x, v, p = setup
    z, m2 = build_model('h36m17_bn', 'fused')
    m2.subbatch_streams = 2
    for _ in range(2):
        v2, p2 = m2(x)
        assert torch.equal(v2, v) and torch.equal(p2, p)
    for n in (255, 129, 128):
        vs, ps = m2(x[:n].contiguous())
        assert torch.equal(vs, v[:n]) and torch.equal(ps, p[:n]), n
    vs, ps = m2(x[:64].contiguous())            # below the threshold: plain path, same context
    assert torch.equal(vs, v[:64]) and torch.equal(ps, p[:64])


def test_forward_is_graph_capturable_and_replays_bitwise(setup):
    """The steady-state forward issues no allocation, no host synchronisation and no host-side read of device memory (the status
    word is host memory), so it can be captured into a hipGraph on a side stream and replayed: same bits as the eager call, also
    after the input buffer was overwritten in place (the graph reads the buffer, not a copy)."""
    m, x, v, p = setup
    xs = x.clone()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2):                       # warm-up on the capture stream (workspace, contexts)
            m(xs)
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        vg, pg = m(xs)
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(vg, v) and torch.equal(pg, p)
    perm = torch.randperm(256, generator=torch.Generator().manual_seed(1)).cuda()
    xs.copy_(x[perm])
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(vg, v[perm]) and torch.equal(pg, p[perm])
    m.device_status()
