"""Bit-identity of the fp32 forward against recorded digests (tests/golden/fp32_digests.json, written by tools/ab_digest.py on a GPU box).
Round 4 made several changes that were meant to be schedules only and checked them by hand with tools/ab_digest.py; this is that check
as a test: per-sample kernel and sample-tiled kernel, persistent and four-launch MDR forms (B = 5 / 256 / 700), both golden variants.
A change that is meant to move bits re-records the file (and re-runs the error budget); any other change must leave it green."""
import json
import os

import pytest

pytestmark = pytest.mark.gpu
PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'fp32_digests.json')


def test_fp32_forward_digests_match_the_recorded_build():
    if not os.path.exists(PATH):
        pytest.skip('no recorded digests (tools/ab_digest.py --write tests/golden/fp32_digests.json on a GPU box)')
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from tools.ab_digest import digests
    want = json.load(open(PATH))['digests']
    got = digests()
    bad = {k: (got.get(k), h) for k, h in want.items() if got.get(k) != h}
    assert not bad, 'outputs moved bits against tests/golden/fp32_digests.json: %s' % bad


def test_byte_lo_weight_stream_equals_the_three_plane_stream_bit_for_bit():
    """k_gat8 streams its weights with the lo plane as one byte per weight by default (gat_roles.hip: H3B); GATOR_GAT8_LOBYTE=0 keeps
    the three fp16 planes.  The switch is read once per process, so the other form runs in a child process; the digests (which cover
    B = 5 / 256 / 700 on k_gat8) must be equal.  (Which kernel ran is visible in profiles/r05_kernel_stats_*: k_gat8<true, 10, false, true>.)"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = {}
    for lb in ('1', '0'):
        env = dict(os.environ, GATOR_GAT8_LOBYTE=lb, GATOR_GAT8_TAIL='0')      # (the fused tail exists on the byte-lo stream's kernel only: both runs keep the two tail launches)
        r = subprocess.run([sys.executable, os.path.join(root, 'tools', 'ab_digest.py'), 'lobyte' + lb], cwd=root, env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        out[lb] = sorted(ln.split()[1:] for ln in r.stdout.splitlines() if ln.startswith('lobyte'))
        assert len(out[lb]) >= 8
    assert out['1'] == out['0']
