"""What the collective of BASELINE config 4 costs the compute kernels of ONE rank -- measured on one GPU, without a node (round-5 review item 8).

Config 4: B = 8192 sharded over 8 ranks, 1024 samples per rank and step, RCCL all-gather of [B, 6890, 3] vertices (+ pose3d): every rank
sends its 85 MB shard to 7 peers and receives 7 x 85 MB.  RCCL moves that data with ordinary kernels -- one workgroup per channel -- on the
same CUs and at the same power budget as the forward, whose dominant kernels follow the energy of their instructions (DESIGN 4c').  This
script runs the rank's per-step compute (the fused forward, B = 1024, J = 17, fp32) alone, then beside a side-stream kernel of N workgroups
that reads the 85 MB shard and writes it 7 times (gator_emulate_gather_traffic: the device-side traffic of the rank's share of the gather),
N in {16, 32, 64}, and reports the compute slowdown, the side kernel's own time (alone and beside the forward) and what that leaves of the
overlap.  It cannot see xGMI itself (link latency, the peers' pace); it bounds what the CHANNEL COUNT costs the compute.

    python tools/contention_model.py [--batch 1024] [--steps 20] [--blocks 7] [--out profiles/r06_contention_model.json]"""
import argparse
import ctypes
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=1024)
    ap.add_argument('--joints', type=int, default=17)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--blocks', type=int, default=7)
    ap.add_argument('--peers', type=int, default=7)
    ap.add_argument('--out', default=None)
    a = ap.parse_args()
    import numpy as np
    import torch
    import bench
    from gator_amd import _lib, synthetic
    dev = torch.device('cuda', 0)
    B, J = a.batch, a.joints
    model, base, alpha = bench.build_model(J, 'fused', dev)
    x = torch.from_numpy(synthetic.synthetic_pose2d(B, J, seed=4)).to(dev)
    out = (torch.empty(B, 6890, 3, device=dev), torch.empty(B, J, 3, device=dev))
    shard_bytes = B * 6890 * 3 * 4
    scratch = torch.empty(a.peers * B * 6890 * 3, device=dev, dtype=torch.float32)
    lib = _lib.load()
    side = torch.cuda.Stream(device=dev)

    def traffic(n_wg):
        _lib.check(lib.gator_emulate_gather_traffic(out[0].data_ptr(), scratch.data_ptr(), shard_bytes, a.peers, n_wg,
                                                    ctypes.c_void_p(side.cuda_stream)), 'gator_emulate_gather_traffic')

    def block(fn):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / a.steps * 1e3

    def med(fn):
        return float(np.median([block(fn) for _ in range(a.blocks)]))

    for _ in range(5):
        model(x, out=out)
    compute_ms = med(lambda: model(x, out=out))
    rows = []
    for n_wg in (16, 32, 64, 128):
        traffic(n_wg)
        torch.cuda.synchronize()
        alone_ms = med(lambda: traffic(n_wg))

        def both():
            # step k's gather (side stream, behind step k's forward) beside step k + 1's forward: the overlap ShardedForward runs
            side.wait_stream(torch.cuda.current_stream(dev))
            traffic(n_wg)
            model(x, out=out)
        both()
        both_ms = med(both)
        # the forward's own duration beside the traffic: HIP events on the compute stream
        evs = []
        for _ in range(a.steps):
            side.wait_stream(torch.cuda.current_stream(dev))
            traffic(n_wg)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            model(x, out=out)
            e1.record()
            evs.append((e0, e1))
        torch.cuda.synchronize()
        fwd_ms = float(np.median([p.elapsed_time(q) for p, q in evs]))
        moved = shard_bytes * (1 + a.peers)
        rows.append({'workgroups': n_wg, 'traffic_alone_ms': round(alone_ms, 4), 'traffic_alone_GBps': round(moved / alone_ms / 1e6, 1),
                     'step_with_traffic_ms': round(both_ms, 4), 'forward_beside_traffic_ms': round(fwd_ms, 4),
                     'compute_slowdown': round(fwd_ms / compute_ms, 4), 'step_over_compute': round(both_ms / compute_ms, 4)})
    res = {'what': 'per-rank compute of BASELINE config 4 (B = %d, J = %d, fp32) beside the device-side traffic of its all-gather share '
                   '(read %.1f MB, write %d x %.1f MB per step) from N workgroups on a side stream' % (B, J, shard_bytes / 1e6, a.peers, shard_bytes / 1e6),
           'compute_alone_ms': round(compute_ms, 4), 'ingress_GBps_needed_to_hide': round(shard_bytes * a.peers / compute_ms / 1e6, 1),
           'rows': rows, 'clocks': bench.gpu_clocks(0)}
    best = min(rows, key=lambda r: r['step_with_traffic_ms'])
    res['favoured_channels'] = best['workgroups']
    text = json.dumps(res, indent=1)
    print(text)
    if a.out:
        with open(a.out, 'w') as f:
            f.write(text + '\n')


if __name__ == '__main__':
    main()
