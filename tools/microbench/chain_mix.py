#!/usr/bin/env python3
"""Why does a trivial kernel cost 4.5 - 5 us inside the training step and 1.9 us in a homogeneous graph chain?  Chains of 400 launches on [1088, 128] fp32
tensors (the encoder's token rows), replayed as a hipGraph:
  same    : one kernel, independent data         (k_t_unary_flat, x -> o)
  dep     : one kernel, each launch reads what the previous wrote (ping-pong)
  mix     : four different kernels in rotation (unary, binary, layernorm forward, column reduction), independent data
  mixdep  : the rotation with real dependencies (what the step looks like)
    python tools/microbench/chain_mix.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gator_amd.train import ops     # noqa: E402


def bench(name, body, n=400):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        body(4)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            body(n)
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        g.replay()
    torch.cuda.synchronize()
    print('%-8s %.2f us per launch' % (name, (time.perf_counter() - t0) / 50 / n * 1e6), flush=True)


def main():
    R, C = 1088, 128
    x = torch.randn(R, C, device='cuda'); y = torch.randn(R, C, device='cuda'); o = torch.empty(R, C, device='cuda'); o2 = torch.empty(R, C, device='cuda')
    w = torch.ones(C, device='cuda'); b = torch.zeros(C, device='cuda')
    col = torch.empty(C, device='cuda')

    def same(n):
        for _ in range(n):
            ops.raw_unary(1, x, 1.0, 0.5, out=o)

    def dep(n):
        a, c = o, o2
        ops.raw_unary(1, x, 1.0, 0.5, out=a)
        for _ in range(n):
            ops.raw_unary(1, a, 1.0, 0.5, out=c)
            a, c = c, a

    def rot(n, chained):
        a = x
        with torch.no_grad():
            for i in range(n // 4):
                t1 = ops.raw_unary(1, a if chained else x, 1.0, 0.5)
                t2 = ops.raw_binary(0, t1 if chained else x, y)
                t3 = ops.layernorm(t2 if chained else x, w, b, 1e-5, 0)
                ops.raw_sum(t3 if chained else x, (0,), out=col)
                a = t3

    bench('same', same)
    bench('dep', dep)
    bench('mix', lambda n: rot(n, False))
    bench('mixdep', lambda n: rot(n, True))


if __name__ == '__main__':
    main()
