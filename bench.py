#!/usr/bin/env python3
"""bench.py -- meshes/sec of the GATOR forward (GAT encoder + MDR head -> 6890 vertices) on N MI355X.

  python bench.py --gpus N --steps K --warmup W        (N>1: launched by torch.distributed.run, one rank per GPU)

A "step" = one GATOR.forward over a batch of B=256 synthetic Human3.6M 17-joint poses per GPU (BASELINE.json configs[1];
weak scaling: every rank owns its own 256 samples), inputs resident in HBM, followed -- for N>1 -- by the RCCL all-gather
of the predicted vertices [N*256, 6890, 3] over xGMI.  Rank 0 prints ONE JSON line.

roofline  : the dominant kernel of the forward (largest share of device time), timed live with HIP events recorded by the
            library on the launch stream (gator_profile_*); algorithmic (fp32) FLOPs per stage from SURVEY Appendix D.
            `peak` is the MFMA peak of the pipe that kernel computes on: the fused kernels run their products on the bf16
            MFMA with every fp32 operand split exactly into three bf16 planes and six partial products per fp32 product
            (x3_common.h), so their ceiling is the dense bf16 peak / 6 = 416.7 TFLOP/s of fp32-equivalent work; a stage
            switched back to the fp32-input MFMA (GATOR_*_X3=0) is priced against 157.3 TFLOP/s.
cpu_baseline: the oracle (torch-CPU restatement of the reference forward, kind "port") timed on this box's host cores on a
            bounded sample (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

PEAK_F32_TFLOPS = 157.3          # MI355X_MICROARCH.md: fp32 vector == fp32-input MFMA peak
PEAK_BF16_TFLOPS = 2500.0        # MI355X_MICROARCH.md: dense bf16 MFMA peak
X3_PRODUCTS = 6                  # bf16 MFMA partial products per fp32 product on the split-precision path (x3_common.h)
PEAK_X3_TFLOPS = round(PEAK_BF16_TFLOPS / X3_PRODUCTS, 1)
# which switch moves a stage back to the fp32-input MFMA kernels (read by the library when the context is created)
STAGE_X3_SWITCH = {'gat': 'GATOR_GAT_X3', 'mdr_layer0': 'GATOR_MDR_X3', 'mdr_layer': 'GATOR_MDR_X3', 'mdr_attn_head': 'GATOR_MDR_X3',
                   'upsample': 'GATOR_UPSAMPLE_X3'}


def stage_pipe(stage, impl):
    """-> (pipe name, peak TFLOP/s of fp32-equivalent work) for a profiled stage."""
    sw = STAGE_X3_SWITCH.get(stage)
    if impl == 'fused' and sw is not None and os.environ.get(sw, '1') != '0':
        return 'bf16 MFMA, split precision (3 planes, %d partial products per fp32 product)' % X3_PRODUCTS, PEAK_X3_TFLOPS
    if stage == 'upsample_bf16':
        return 'bf16 MFMA', PEAK_BF16_TFLOPS
    return 'fp32-input MFMA', PEAK_F32_TFLOPS
FLOPS_PER_MESH = {17: 4.10e8, 19: 4.18e8}      # dense algorithmic count
# algorithmic MFLOP per mesh per stage, J=17
# mdr_layer  = one middle LBF launch: 431x431 attention core of layer l-1 (47.6) + its out-proj (3.5) + cross-attn/Mlp of
#              layer l (37.5) + q/k/v in-proj of layer l (10.6) = 99.2 ; mdr_layer0 = tokenise + the last two items
STAGE_MFLOP = {'upsample_bf16': 53.45, 'gat': 56.66, 'mdr_layer0': 48.7, 'mdr_layer': 99.2, 'mdr_attn_head': 51.4, 'mdr_head': 1.3, 'upsample': 53.45}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=30)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=256, help='samples per GPU per step')
    ap.add_argument('--joints', type=int, default=17)
    ap.add_argument('--impl', default=os.environ.get('GATOR_AMD_IMPL', 'fused'))
    ap.add_argument('--precision', default='f32', choices=['f32', 'bf16'], help='bf16: vertex regressor on bf16 MFMA (config 3)')
    ap.add_argument('--subbatch-variant', action='store_true', help='also time the sub-batch-streams=2 mode (extra key)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-seconds', type=float, default=15.0)
    return ap.parse_args()


def build_model(J, impl, device):
    import scipy.sparse as sps
    from gator_amd import models, synthetic
    alpha = J == 19
    base = synthetic.make_base_data(0)
    sk17 = ((0, 7), (7, 8), (8, 9), (9, 10), (8, 11), (11, 12), (12, 13), (8, 14), (14, 15), (15, 16), (0, 1), (1, 2), (2, 3),
            (0, 4), (4, 5), (5, 6), (1, 4), (2, 5), (3, 6), (14, 11), (15, 12), (16, 13))
    sk19 = ((1, 2), (0, 1), (0, 2), (2, 4), (1, 3), (6, 8), (8, 10), (5, 7), (7, 9), (12, 14), (14, 16), (11, 13), (13, 15),
            (17, 11), (17, 12), (17, 18), (18, 5), (18, 6), (18, 0), (3, 4), (5, 6), (7, 8), (9, 10), (11, 12), (13, 14), (15, 16))
    adj = np.eye(J)
    for a, b in (sk17 if J == 17 else sk19):
        adj[a, b] = adj[b, a] = 1
    m = models.GATOR.get_model(J, 128, 6, [None, sps.csr_matrix(adj)], 1, torch.Tensor(synthetic.model_j_regressor(J)),
                               base_data=base, alpha=alpha)
    sd = m.state_dict()
    w = synthetic.seeded_state_dict(synthetic.shapes_of(sd), base['rs'])
    sd.update({k: torch.from_numpy(v) for k, v in w.items()})
    m.load_state_dict(sd)
    m.impl = impl
    return m.to(device).eval(), base, alpha


STAGE_KERNEL = {'gat': 'k_gat<true>', 'mdr_layer0': 'k_mdr_layer<0, true>', 'mdr_layer': 'k_mdr_layer<1, true>',
                'mdr_attn_head': 'k_mdr_layer<2, true>', 'upsample': 'k_upsample_x3'}


def pmc_traffic(stage, B):
    """HBM-side bytes per launch of the stage's kernel from the committed rocprofv3 PMC passes (FETCH_SIZE x2 as the gfx950
    guide prescribes, + WRITE_SIZE), collected with this same command at B=256; None for other batch sizes / kernels."""
    path = os.path.join(ROOT, 'profiles', 'r01_pmc_summary_B256.json')
    if B != 256 or not os.path.exists(path):
        return None
    for row in json.load(open(path)):
        if row['kernel'] == STAGE_KERNEL.get(stage):
            return int((row['fetch_MB_corrected'] + row['write_MB']) * 1048576)
    return None


def cpu_baseline(model, base, alpha, J, seconds):
    """Oracle fp32 on the host cores: bounded sample of the same workload (synthetic poses, same weights).  The thread count
    is swept (all cores is pathological for these tiny tensors on a 100+-core host) and the best rate is reported."""
    from gator_amd import synthetic
    from oracle import gator_oracle as go
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    c = go.Consts(J, synthetic.model_j_regressor(J), base, alpha)
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cands = sorted({n for n in (8, 16, 32, 64) if n <= avail}) or [avail]   # >64 threads: minutes per forward
    best, sample, cores = 0.0, '', 1
    t_end = time.time() + seconds
    per = seconds / (len(cands) * 2)
    for nt in cands:
        torch.set_num_threads(nt)
        for B in (64, 256):
            x = torch.from_numpy(synthetic.synthetic_pose2d(B, J, seed=1))
            t_stop = time.time() + per
            go.gator_forward(sd, c, x, torch.float32)                      # warm-up
            ts = []
            while len(ts) < 5 and (time.time() < t_stop or len(ts) < 1):
                t0 = time.perf_counter()
                go.gator_forward(sd, c, x, torch.float32)
                ts.append(time.perf_counter() - t0)
            rate = B / float(np.median(ts))
            if rate > best:
                best, cores = rate, nt
                sample = 'B=%d x %d timed forwards (median), fp32 torch-CPU, %d of %d host threads (best of sweep %s)' % (
                    B, len(ts), nt, avail, cands)
        if time.time() > t_end + seconds:
            break
    return {'value': round(best, 1), 'unit': 'meshes/sec', 'cores': int(cores), 'kind': 'port', 'sample': sample}


def main():
    a = parse()
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    assert torch.cuda.is_available(), 'bench.py needs a HIP device (there is no CPU path)'
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)   # backend "nccl" == RCCL on ROCm
    from gator_amd import synthetic
    from gator_amd.parallel import ShardedForward
    J, B = a.joints, a.batch
    model, base, alpha = build_model(J, a.impl, dev)
    model.precision = a.precision
    runner = ShardedForward(model, world, rank, dist)
    x = torch.from_numpy(synthetic.synthetic_pose2d(B, J, seed=1000 + rank)).to(dev)     # this rank's shard, resident in HBM

    def sync():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        out = runner.step(x)
    sync()
    model.profile(4)          # HIP-event brackets on every 4th timed step (the brackets themselves cost ~3 % of a step)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        out = runner.step(x)
    sync()
    dt = time.perf_counter() - t0
    prof = model.profile_read()
    model.profile(0)
    if dist is not None:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    assert out[0].shape == (B * world, 6890, 3)
    if rank == 0:
        ms = dt / a.steps * 1e3
        value = B * world * a.steps / dt
        roof = None
        if prof:
            name, (tot_ms, calls) = max(prof.items(), key=lambda kv: kv[1][0])
            avg_s = tot_ms / calls * 1e-3
            mflop = STAGE_MFLOP.get(name.split(':')[0], None)
            if mflop is not None and avg_s > 0:
                ach = mflop * 1e6 * B / avg_s / 1e12
                traffic = pmc_traffic(name, B)
                pipe, peak = stage_pipe(name.split(':')[0], a.impl)
                roof = {'bound': 'mfma', 'kernel': name, 'achieved': round(ach, 2), 'peak': peak, 'unit': 'TFLOP/s',
                        'frac': round(ach / peak, 4), 'traffic': traffic, 'avg_launch_ms': round(avg_s * 1e3, 4), 'pipe': pipe,
                        'stages_ms': {k: round(v[0] / v[1], 4) for k, v in prof.items()}}
        if roof is None:   # no per-kernel events available (bring-up path): price the whole forward
            ach = FLOPS_PER_MESH[J] * value / world / 1e12
            roof = {'bound': 'mfma', 'kernel': 'whole forward', 'achieved': round(ach, 2), 'peak': PEAK_F32_TFLOPS,
                    'unit': 'TFLOP/s', 'frac': round(ach / PEAK_F32_TFLOPS, 4), 'traffic': None}
        line = {'metric': 'meshes/sec (B=256, J=17) GATOR forward', 'value': round(value, 1), 'unit': 'meshes/sec',
                'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup, 'ms_per_step': round(ms, 4), 'higher_is_better': True,
                'scaling': 'weak', 'vs_baseline': None, 'dtype': a.precision, 'data': 'synthetic',
                'config': {'workload': 'B=%d synthetic Human3.6M %d-joint poses per GPU, GAT+MDR forward fp32%s'
                           % (B, J, ', RCCL all-gather of [%d,6890,3] vertices' % (B * world) if world > 1 else ''),
                           'batch_per_gpu': B, 'num_joint': J, 'impl': a.impl, 'parallelism': 'dp%d' % world},
                'roofline': roof}
        if world == 1 and B >= 128 and a.subbatch_variant:
            # same workload with the library's sub-batch pipelining (two half-batches on two streams; bit-identical results).
            # Reported beside the headline, not as it: concurrent streams make per-kernel durations (and so `roofline`) ambiguous.
            m2, _, _ = build_model(J, a.impl, dev)
            m2.precision = a.precision
            m2.subbatch_streams = 2
            for _ in range(a.warmup):
                m2(x)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(a.steps):
                m2(x)
            torch.cuda.synchronize()
            dt2 = time.perf_counter() - t1
            line['subbatch_streams_2'] = {'value': round(B * a.steps / dt2, 1), 'ms_per_step': round(dt2 / a.steps * 1e3, 4)}
        if world == 1 and not a.no_cpu_baseline:
            line['cpu_baseline'] = cpu_baseline(model, base, alpha, J, a.cpu_seconds)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
