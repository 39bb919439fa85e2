#!/usr/bin/env python3
"""Does a hipGraph run independent branches side by side on this stack, and what does a fork / join inside a graph cost?
(Training row: the step is one chain of ~680 small launches, each on a few CUs; the weight-gradient GEMMs and the MGCN / attention
branches of a block are independent of the chain next to them.)

Chains of N tiny kernels (torch add_ on a 64 K-element tensor, ~2 us of work), captured (a) as one chain of 2N, (b) as two chains of N
on two streams forked and joined once, (c) forked and joined every `seg` kernels.  Replayed 50 times; us per replay.
    python tools/microbench/graph_branches.py"""
import time
import torch


def build(n, mode, seg, size):
    a = torch.zeros(size, device='cuda'); b = torch.zeros(size, device='cuda')
    s1 = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        cur = torch.cuda.current_stream()
        if mode == 'chain':
            for _ in range(n):
                a.add_(1.0); b.add_(1.0)
        else:
            k = n if mode == 'fork1' else seg
            for _ in range(n // k):
                s1.wait_stream(cur)
                with torch.cuda.stream(s1):
                    for _ in range(k):
                        b.add_(1.0)
                for _ in range(k):
                    a.add_(1.0)
                cur.wait_stream(s1)
    return g, a, b


def main():
    for size in (1 << 16, 1 << 22):
        for mode, seg in (('chain', 0), ('fork1', 0), ('forkseg', 20), ('forkseg', 5), ('forkseg', 1)):
            n = 200
            g, a, b = build(n, mode, seg, size)
            for _ in range(5):
                g.replay()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(50):
                g.replay()
            torch.cuda.synchronize()
            us = (time.perf_counter() - t0) / 50 * 1e6
            assert float(a[0]) == float(b[0])
            print('elements %8d  %-8s seg %3d : %8.1f us per replay of 2 x %d kernels  (%.2f us per kernel)' % (size, mode, seg, us, n, us / (2 * n)), flush=True)


if __name__ == '__main__':
    main()
