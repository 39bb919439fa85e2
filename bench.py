#!/usr/bin/env python3
"""bench.py -- meshes/sec of the GATOR forward (GAT encoder + MDR head -> 6890 vertices) on N MI355X.

  python bench.py --gpus N --steps K --warmup W

N = 1 runs in this process.  N > 1: one rank per GPU over RCCL.  When the script is started directly (no RANK in the
environment) it starts `python -m torch.distributed.run --nproc-per-node N bench.py ...` as a CHILD process -- before
anything in this process has touched the GPU -- relays rank 0's JSON line and exits with the child's code; when it is
started by torch.distributed.run (RANK/LOCAL_RANK/WORLD_SIZE set) it is one of the ranks, and --gpus must equal WORLD_SIZE.

A "step" = one GATOR.forward over a batch of B synthetic poses per GPU (default: B=256 Human3.6M 17-joint, BASELINE.json
configs[1]; weak scaling: every rank owns its own B samples), inputs resident in HBM, followed -- for N>1 -- by the RCCL
all-gather of the predicted vertices [N*B, 6890, 3] over xGMI.  Rank 0 prints ONE JSON line.

Timing: W warm-up steps, then R blocks (default 11) of EXACTLY K steps, each block bracketed by barrier + synchronize on
both sides and reduced with MAX over ranks; the line reports the MEDIAN block (`ms_per_step` x `steps` = that block).

roofline  : the dominant kernel of the forward (largest share of device time), timed live with HIP events recorded by the
            library on the launch stream (gator_profile_*); algorithmic (fp32) FLOPs per stage from SURVEY Appendix D.
            `peak` is the ceiling of the arithmetic that kernel EXECUTES: every product runs on the 16-bit MFMA (2.5 PFLOP/s dense)
            from split fp32 operands (x3_common.h) -- token-wise linears on four partial products (weights exact on three fp16
            planes, activations rounded to two), attention cores and the vertex regressor on three (both operands on two planes) --
            so a stage's ceiling is its FLOPs / sum(part_i x products_i / 2500) (STAGE_PRODUCTS below; 714 TFLOP/s of fp32-equivalent
            work for the MDR layers); a stage switched to the exact three-plane bf16 split (six products) is priced against 416.7, one
            switched back to the fp32-input MFMA (GATOR_*_X3=0) against 157.3 TFLOP/s.
            `traffic_ratio` = HBM-side bytes of the WHOLE forward per mesh (committed PMC digest of this command: FETCH_SIZE x 2 +
            WRITE_SIZE over every launch of a step) / the 83 020 compulsory bytes of SURVEY 8(d); `frac_of_dense_16bit_peak_2500` =
            executed MFMA FLOP of the dominant kernel (PMC instruction counts) / live duration / 2.5 PFLOP/s.
variants   : (N = 1) the same workload in the same process with the library's A/B switches: no rounded operand anywhere
            (GATOR_MDR_X3=1), every product on the fp32-input MFMA, the previous encoder kernel; and the headline re-measured with
            the variants' protocol.  clocks: sclk / mclk levels and power cap from sysfs.
parity     : (N = 1, with the cpu_baseline leg) max |verts - fp64 oracle| in mm of the first 32 samples of the TIMED batch, measured
            after the timed region; the statistical margin over >= 16k samples is tests/error_budget.py -> profiles/r04_error_budget.*.
--config   : BASELINE.json presets (2: B=256 J=17 fp32; 3: B=2048 J=19 bf16; 4: 1024 per GPU + all-gather, 8 GPUs = B 8192; 5: evaluation
            mode, all-reduce only); `config.baseline_config` names the BASELINE config a run is.
cpu_baseline: the oracle (torch-CPU restatement of the reference forward, kind "port") timed on this box's host cores with
            BASELINE.md section 3's protocol (B in {16,64,256}, 3 warm-up + 10 timed, median per B, best B), rank 0, N=1 only.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_TFLOPS = 157.3          # MI355X_MICROARCH.md: fp32 vector == fp32-input MFMA peak
PEAK_BF16_TFLOPS = 2500.0        # MI355X_MICROARCH.md: dense bf16 MFMA peak
X3_PRODUCTS = 6                  # bf16 MFMA partial products per fp32 product on the split-precision path (x3_common.h)
PEAK_X3_TFLOPS = round(PEAK_BF16_TFLOPS / X3_PRODUCTS, 1)
# which switch moves a stage back to the fp32-input MFMA kernels (read by the library when the context is created)
STAGE_X3_SWITCH = {'gat': 'GATOR_GAT_X3', 'gat_tail': None, 'mdr_layer0': 'GATOR_MDR_X3', 'mdr_layer': 'GATOR_MDR_X3',
                   'mdr_attn_head': 'GATOR_MDR_X3', 'mdr_layers': 'GATOR_MDR_X3', 'upsample': 'GATOR_UPSAMPLE_X3'}
FLOPS_PER_MESH = {17: 4.10e8, 19: 4.18e8}      # dense algorithmic count
BYTES_PER_MESH = {17: 83020, 19: 83060}        # compulsory HBM bytes (pose2d in, vertices + pose3d out)
# algorithmic MFLOP per mesh per stage, J=17
# mdr_layer  = one middle LBF launch: 431x431 attention core of layer l-1 (47.6) + its out-proj (3.5) + cross-attn/Mlp of
#              layer l (37.5) + q/k/v in-proj of layer l (10.6) = 99.2 ; mdr_layer0 = tokenise + the last two items
# mdr_layers = the four of them in one persistent launch (k_mdr_persist, the default): 48.7 + 2 x 99.2 + 51.4
STAGE_MFLOP = {'upsample_bf16': 53.45, 'gat': 56.66, 'mdr_layer0': 48.7, 'mdr_layer': 99.2, 'mdr_attn_head': 51.4, 'mdr_head': 1.3,
               'mdr_layers': 298.5,
               'upsample': 53.45}
# inter-kernel operand bytes per mesh the dominant kernels are DESIGNED to move: the residual tile set
# vf (14 tiles x 2 blocks x 4 KiB) and the Q/K/V tile sets, read and/or written once
_VF, _QKV = 14 * 2 * 4096, 3 * 14 * 2 * 4096          # Q/K/V as two fp16 planes: 4 KiB tiles (6 KiB when GATOR_MDR_X3=1)
STAGE_BYTES = {'mdr_layer0': _VF + _QKV, 'mdr_layer': 2 * (_VF + _QKV), 'mdr_attn_head': _VF + _QKV + 431 * 32 * 4 + 431 * 64 * 4,
               'mdr_layers': 6 * (_VF + _QKV) + 431 * 32 * 4 + 431 * 64 * 4,
               'upsample': 3 * 431 * 2 * 3 + 6890 * 3 * 4, 'gat': 136 + 17 * 128 * 4 + 204 + 12 * 4096}
# kernel of a profiled stage in the PMC digest: (name prefix, characters that may follow it) -- 'k_gat8<...' but not k_gat_joint / k_gat_lifter
STAGE_KERNEL = {'gat': ('k_gat8', '<'), 'mdr_layer0': ('k_mdr_layer<0,', None), 'mdr_layer': ('k_mdr_layer<1,', None),
                'mdr_attn_head': ('k_mdr_layer<2,', None), 'mdr_layers': ('k_mdr_persist', '<'), 'upsample': ('k_upsample_x2', '<(')}
# BASELINE.json `configs`, 1-based as VERDICT.md numbers them (config 1 is the reference's own CPU demo): per-GPU presets
BASELINE_CONFIGS = {
    2: dict(batch=256, joints=17, precision='f32', mode='gather', gpus=1,
            name='configs[1]: B=256 synthetic Human3.6M 17-joint poses, GAT+MDR forward, 1xMI355X fp32'),
    3: dict(batch=2048, joints=19, precision='bf16', mode='gather', gpus=1,
            name='configs[2]: B=2048 COCO 19-joint poses, full GATOR forward bf16, 1xMI355X (MFMA vertex regressor)'),
    4: dict(batch=1024, joints=17, precision='f32', mode='gather', gpus=8,
            name='configs[3]: B=8192 Human3.6M poses sharded over 8xMI355X (1024 per GPU), RCCL all-gather of 6890x3 vertices over xGMI'),
    5: dict(batch=1024, joints=19, precision='f32', mode='eval', gpus=8,
            name='configs[4]: 3DPW path_3dpw graph (19 input joints), 8xMI355X end-to-end inference with MPJPE/PA-MPJPE eval (all-reduce only)'),
}


def baseline_config_of(a, world):
    """Which BASELINE config a run is (exactly, or as a scaled-down instance of it), for the line's `config.baseline_config`."""
    if a.config:
        c = BASELINE_CONFIGS[a.config]
        exact = (a.batch, a.joints, a.precision, a.mode, world) == (c['batch'], c['joints'], c['precision'], c['mode'], c['gpus'])
        return ('config %d = %s' % (a.config, c['name'])) + ('' if exact else
                ' -- run here with batch %d per GPU on %d GPU(s)' % (a.batch, world))
    for k, c in BASELINE_CONFIGS.items():
        if (a.batch, a.joints, a.precision, a.mode) == (c['batch'], c['joints'], c['precision'], c['mode']) and (world == c['gpus'] or c['gpus'] == 1):
            return ('config %d = %s' % (k, c['name'])) + ('' if world == c['gpus'] else ' -- weak-scaled to %d GPUs (same batch per GPU)' % world)
    return 'none (free-form run: batch %d per GPU, J=%d, %s, mode %s, %d GPU(s))' % (a.batch, a.joints, a.precision, a.mode, world)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=30)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--blocks', type=int, default=11, help='timed blocks of --steps steps each (a minimum: see --min-timed-s); the median block is reported')
    ap.add_argument('--min-timed-s', type=float, default=3.0,
                    help='keep adding timed blocks of --steps steps until the timed region holds at least this many seconds of forwards '
                         '(N = 1; with ranks the block count must agree, so it is fixed from the first blocks by rank 0 and broadcast); 0 = exactly --blocks')
    ap.add_argument('--config', type=int, default=0, choices=[0, 2, 3, 4, 5],
                    help='BASELINE.json preset (1-based as in VERDICT.md): 2 = B=256 J=17 fp32, 3 = B=2048 J=19 bf16, 4 = 1024 per GPU J=17 + '
                         'all-gather (8 GPUs = B 8192), 5 = evaluation mode J=19 (all-reduce only); --batch/--joints/--precision/--mode override')
    ap.add_argument('--batch', type=int, default=None, help='samples per GPU per step (default 256, or the preset)')
    ap.add_argument('--joints', type=int, default=None)
    ap.add_argument('--impl', default=os.environ.get('GATOR_AMD_IMPL', 'fused'))
    ap.add_argument('--precision', default=None, choices=['f32', 'bf16'],
                    help='bf16 = gator_forward_bf16, the 16-bit operand mode of BASELINE config 3: MDR layers on one fp16 activation plane (dtype f16 in the line)')
    ap.add_argument('--mode', default=None, choices=['gather', 'eval'],
                    help='N>1: all-gather the vertices (config 4) or the all-reduce-only evaluation mode (config 5)')
    ap.add_argument('--subbatch-variant', action='store_true', help='also time the sub-batch-streams=2 mode (extra key)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-variants', action='store_true', help='skip the A/B variants measured beside the headline (N = 1 only)')
    ap.add_argument('--no-config3', action='store_true', help='skip the `config3` sub-record (BASELINE configs[2] beside the default headline, N = 1 only)')
    a = ap.parse_args()
    preset = BASELINE_CONFIGS.get(a.config, dict(batch=256, joints=17, precision='f32', mode='gather'))
    for k in ('batch', 'joints', 'precision', 'mode'):
        if getattr(a, k) is None:
            setattr(a, k, preset[k])
    return a


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_ranks(a):
    """Started directly with --gpus N > 1: run the N ranks as a child torch.distributed.run (this process never initialises
    the GPU, and nothing is exec'd over a GPU-touched process), relay its output, return its exit code."""
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')      # dmabuf IPC only on this driver (RCCL needs it)
    env.setdefault('NCCL_DEBUG', 'WARN')                   # no version banner on stdout (the result line must be the last one)
    env.setdefault('OMP_NUM_THREADS', '8')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(a.gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in p.stdout:
        s = ln.strip()
        if s.startswith('{') and '"metric"' in s:
            line = s
        else:
            sys.stderr.write(ln)
    rc = p.wait()
    if line is not None:
        print(line, flush=True)
    elif rc == 0:
        sys.stderr.write('bench.py: the ranks exited without printing a result line\n')
        rc = 1
    return rc


# How many 16-bit MFMA partial products each fp32 product of a stage costs (x3_common.h), as (MFLOP, products) parts, for the
# default configuration (GATOR_MDR_X3=2, GATOR_UPSAMPLE_X3=2); every other split-precision stage runs on six (exact three-plane
# bf16 split).  MDR stages: the 431x431 self-attention core (47.6 MFLOP per layer) and the J-joint cross-attention (1.9 per layer) on
# two fp16 planes = 3 products; the token-wise linears with activations on two planes and weights exact on three = 4 products.
_SA, _CA = 47.6, 1.9
STAGE_PRODUCTS = {'mdr_layer0': [(_CA, 3), (48.7 - _CA, 4)],
                  'mdr_layer': [(_SA + _CA, 3), (99.2 - _SA - _CA, 4)],
                  'mdr_attn_head': [(_SA, 3), (51.4 - _SA, 4)],
                  'mdr_layers': [(3 * _SA + 3 * _CA, 3), (298.5 - 3 * _SA - 3 * _CA, 4)],
                  'upsample': [(53.45, 3)],
                  # k_gat8 (B <= 1024): the 16 token-wise linears per block on 4 products; of the J x J operators the attention
                  # (QK^T + PV, 0.89 MFLOP) on two fp16 planes = 3 products, the hop aggregations (0.50) against exact 0/1 fp16 masks = 2,
                  # the MGCN adjacency product (0.89) on the fp32-input MFMA (= 16 in units of the 16-bit rate)
                  'gat': [(54.4, 4), (0.89, 3), (0.50, 2), (56.66 - 54.4 - 0.89 - 0.50, 16)]}
# BASELINE config 3 (--precision bf16 = gator_forward_bf16, round 5): the MDR layers on ONE fp16 activation plane -- attention cores one
# product, token-wise linears two (weights on two planes) -- encoder and vertex regressor as in the fp32 configuration
STAGE_PRODUCTS_C3 = {'mdr_layer0': [(_CA, 1), (48.7 - _CA, 2)], 'mdr_layer': [(_SA + _CA, 1), (99.2 - _SA - _CA, 2)],
                     'mdr_attn_head': [(_SA, 1), (51.4 - _SA, 2)], 'mdr_layers': [(3 * _SA + 3 * _CA, 1), (298.5 - 3 * _SA - 3 * _CA, 2)]}
STAGE_PRODUCTS_C3['upsample'] = [(53.45, 2)]                                   # weights one plane, coarse vertices two
STAGE_PRODUCTS_C3['gat'] = [(54.4, 2), (0.89, 3), (0.50, 2), (56.66 - 54.4 - 0.89 - 0.50, 16)]      # k_gat8's token-wise linears on two products
PEAK_X2_TFLOPS = round(PEAK_BF16_TFLOPS / 3, 1)


def stage_pipe(stage, impl, precision='f32'):
    """-> (pipe name, peak TFLOP/s of fp32-equivalent work) for a profiled stage.  A stage that mixes split forms is priced
    against the time-weighted ceiling of the arithmetic it executes: total / sum(part_i / (2500 / products_i))."""
    sw = STAGE_X3_SWITCH.get(stage)
    mode = os.environ.get(sw, '2' if sw in ('GATOR_MDR_X3', 'GATOR_UPSAMPLE_X3') else '1') if sw else '0'
    if impl == 'fused' and sw is not None and mode != '0':
        parts = STAGE_PRODUCTS.get(stage) if mode == '2' else None
        c3 = precision == 'bf16' and mode == '2'
        # config 3: each stage follows its own switch (fused_api.hip: GATOR_C3_MDR / GATOR_C3_UPSAMPLE_W1 / GATOR_C3_ENCODER, all default on)
        c3_switch = 'GATOR_C3_UPSAMPLE_W1' if stage == 'upsample' else 'GATOR_C3_MDR'
        if c3 and stage in STAGE_PRODUCTS_C3 and stage != 'gat' and os.environ.get(c3_switch, '1') != '0':
            parts = STAGE_PRODUCTS_C3[stage]
        if stage == 'gat':      # the four-product form is k_gat8's (GATOR_GAT8_H4, default on); k_gat / k_gat_tiled run six
            g8 = os.environ.get('GATOR_GAT8', '1') != '0' and os.environ.get('GATOR_GAT8_H4', '1') != '0'
            c3e = precision == 'bf16' and os.environ.get('GATOR_C3_ENCODER', '1') != '0'
            parts = (STAGE_PRODUCTS_C3['gat'] if c3e else STAGE_PRODUCTS['gat']) if g8 else None
        if parts:
            tot = sum(m for m, _ in parts)
            peak = tot / sum(m * k / PEAK_BF16_TFLOPS for m, k in parts)
            return ('16-bit MFMA, split precision: ' + ' + '.join('%.1f MFLOP on %d partial products' % (m, k) for m, k in parts)), round(peak, 1)
        return 'bf16 MFMA, split precision (3 planes, %d partial products per fp32 product)' % X3_PRODUCTS, PEAK_X3_TFLOPS
    if stage == 'upsample_bf16':
        return 'bf16 MFMA', PEAK_BF16_TFLOPS
    return 'fp32-input MFMA', PEAK_F32_TFLOPS


def build_model(J, impl, device):
    import numpy as np
    import scipy.sparse as sps
    import torch
    from gator_amd import models, synthetic
    alpha = J == 19
    base = synthetic.make_base_data(0)
    sk17 = ((0, 7), (7, 8), (8, 9), (9, 10), (8, 11), (11, 12), (12, 13), (8, 14), (14, 15), (15, 16), (0, 1), (1, 2), (2, 3),
            (0, 4), (4, 5), (5, 6), (1, 4), (2, 5), (3, 6), (14, 11), (15, 12), (16, 13))
    sk19 = ((1, 2), (0, 1), (0, 2), (2, 4), (1, 3), (6, 8), (8, 10), (5, 7), (7, 9), (12, 14), (14, 16), (11, 13), (13, 15),
            (17, 11), (17, 12), (17, 18), (18, 5), (18, 6), (18, 0), (3, 4), (5, 6), (7, 8), (9, 10), (11, 12), (13, 14), (15, 16))
    adj = np.eye(J)
    for p, q in (sk17 if J == 17 else sk19):
        adj[p, q] = adj[q, p] = 1
    m = models.GATOR.get_model(J, 128, 6, [None, sps.csr_matrix(adj)], 1, torch.Tensor(synthetic.model_j_regressor(J)),
                               base_data=base, alpha=alpha)
    sd = m.state_dict()
    w = synthetic.seeded_state_dict(synthetic.shapes_of(sd), base['rs'])
    sd.update({k: torch.from_numpy(v) for k, v in w.items()})
    m.load_state_dict(sd)
    m.impl = impl
    return m.to(device).eval(), base, alpha


def pmc_digest(B):
    """The committed rocprofv3 PMC digest of this same command at B=256 (tools/profile_round.sh + tools/pmc_digest.py), newest
    round first (profiles/rNN_pmc_summary_B256.json, highest NN); None for other batch sizes."""
    if B != 256:
        return None, None
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r[0-9][0-9]_pmc_summary_B256.json')), reverse=True):
        name = os.path.basename(path)
        prov = os.path.join(ROOT, 'profiles', name.replace('_pmc_summary_B256.json', '_pmc_provenance.json'))
        commit = json.load(open(prov)).get('commit') if os.path.exists(prov) else None
        return json.load(open(path)), (name + (' @ ' + commit if commit else ''))
    return None, None


def pmc_row(rows, stage):
    want, follow = STAGE_KERNEL.get(stage, ('', None))
    for row in rows or []:
        k = row['kernel']
        if want and k.startswith(want) and (follow is None or len(k) == len(want) or k[len(want)] in follow):
            return row
    return None


def pmc_traffic(rows, stage):
    """HBM-side bytes per launch of the stage's kernel: FETCH_SIZE x2 (as the gfx950 guide prescribes) + WRITE_SIZE."""
    row = pmc_row(rows, stage)
    if row is None or 'fetch_MB_corrected' not in row:
        return None
    return int((row['fetch_MB_corrected'] + row.get('write_MB', 0.0)) * 1048576)


def pmc_forward_bytes(rows):
    """HBM-side bytes of ONE whole forward (every launch of every kernel of a step) from the digest: the figure SURVEY 8(d)'s
    83 kB per mesh of compulsory traffic is to be compared with.  A step = one launch of the vertex-regressor kernel."""
    if not rows:
        return None
    steps = max([r.get('launches', 0) for r in rows if r['kernel'].startswith('k_upsample')] or [0])
    if steps <= 0:
        return None
    tot = 0.0
    for r in rows:
        if 'fetch_MB_corrected' in r:
            tot += (r['fetch_MB_corrected'] + r.get('write_MB', 0.0)) * 1048576 * r.get('launches', steps) / steps
    return int(tot)


def gpu_clocks(index=0):
    """Shader / memory clock levels and the power cap of the device as sysfs reports them after the timed region (box-to-box spread of
    one binary is 280-340k meshes/s: this makes a line comparable).  Read from files: a GPU-initialised process must not start
    another program on this pool, so no rocm-smi.  None where a file is not readable by this user."""
    import glob
    out = {}
    cards = sorted(glob.glob('/sys/class/drm/card[0-9]*/device/pp_dpm_sclk'))
    if not cards:
        return {'error': 'no /sys/class/drm/card*/device/pp_dpm_sclk'}
    dev = os.path.dirname(cards[min(index, len(cards) - 1)])

    def levels(name):
        try:
            rows = [ln.strip() for ln in open(os.path.join(dev, name)) if ln.strip()]
        except OSError:
            return None
        cur = [r for r in rows if r.endswith('*')]
        return {'current': cur[0].rstrip('*').strip() if cur else None, 'levels': [r.rstrip('*').strip() for r in rows]}

    out['sclk'], out['mclk'] = levels('pp_dpm_sclk'), levels('pp_dpm_mclk')
    for key, pat in (('power_cap_w', 'hwmon/hwmon*/power1_cap'), ('power_w', 'hwmon/hwmon*/power1_average'), ('power_input_w', 'hwmon/hwmon*/power1_input')):
        f = glob.glob(os.path.join(dev, pat))
        try:
            out[key] = round(int(open(f[0]).read().strip()) / 1e6, 1) if f else None
        except (OSError, ValueError):
            out[key] = None
    return out


def cpu_baseline(model, base, alpha, J):
    """Oracle fp32 on the host cores, BASELINE.md section 3: B in {16, 64, 256}, 3 warm-up + 10 timed forwards each, median
    meshes/s per B, best B reported.  Thread count: `torch.set_num_threads(os.cpu_count())` is pathological on the GPU box's
    100+-thread host (1.2 meshes/s at 256 threads: these are tiny tensors), so the count is picked once by a short probe
    (one B=64 forward each at 8/16/32 threads) and stated."""
    import numpy as np
    import torch
    from gator_amd import synthetic
    from oracle import gator_oracle as go
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    c = go.Consts(J, synthetic.model_j_regressor(J), base, alpha)
    go.KEEP_ATTENTION_MAPS = True       # as the reference's modules do between forwards (oracle/gator_oracle.py); released below
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cands = sorted({n for n in (8, 16, 32) if n <= avail}) or [avail]
    xp = torch.from_numpy(synthetic.synthetic_pose2d(64, J, seed=1))
    probe = {}
    with torch.no_grad():
        for nt in cands:
            torch.set_num_threads(nt)
            go.gator_forward(sd, c, xp, torch.float32)
            t0 = time.perf_counter()
            go.gator_forward(sd, c, xp, torch.float32)
            probe[nt] = time.perf_counter() - t0
        nt = min(probe, key=probe.get)
        torch.set_num_threads(nt)
        per_b = {}
        for B in (16, 64, 256):
            x = torch.from_numpy(synthetic.synthetic_pose2d(B, J, seed=1))
            for _ in range(3):
                go.gator_forward(sd, c, x, torch.float32)
            ts = []
            for _ in range(10):
                t0 = time.perf_counter()
                go.gator_forward(sd, c, x, torch.float32)
                ts.append(time.perf_counter() - t0)
            per_b[B] = B / float(np.median(ts))
    go.KEEP_ATTENTION_MAPS = False
    go._ATTN_KEPT.clear()
    bb = max(per_b, key=per_b.get)
    legit = None      # is the port a fair stand-in?  measured in the dev container against the imported reference (tools/cpu_port_vs_reference.py)
    try:
        import glob
        rec = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'profiles', 'r[0-9][0-9]_cpu_port_vs_reference.json')))[-1]      # the newest round's record
        with open(rec) as fh:
            d = json.load(fh)
        legit = {'oracle_over_reference_by_batch': {k: v['oracle_over_reference'] for k, v in d['per_batch'].items()}, 'threads': d['threads'],
                 'where': 'dev container (8 shared vCPUs), separate processes, profiles/%s' % os.path.basename(rec).replace('.json', '.txt')}
    except Exception:
        pass
    return {'value': round(per_b[bb], 1), 'unit': 'meshes/sec', 'cores': int(nt), 'kind': 'port', 'port_vs_reference': legit,
            'sample': 'B in {16,64,256} x (3 warm-up + 10 timed forwards), median per B, best B=%d; fp32 torch-CPU oracle, %d of %d '
                      'host threads (probe over %s)' % (bb, nt, avail, cands),
            'per_batch': {str(k): round(v, 1) for k, v in per_b.items()}}


def parity_check(model, base, alpha, J, x, verts, n=32, bar_mm=1e-3):
    """The checker leg beside cpu_baseline (never timed, never the product path): the fp64 oracle on the first `n` samples of the
    batch that was timed, against the vertices the timed forward wrote for them."""
    import numpy as np
    import torch
    from gator_amd import synthetic
    from oracle import gator_oracle as go
    n = min(n, x.shape[0])
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    c = go.Consts(J, synthetic.model_j_regressor(J), base, alpha)
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    ref, _ = go.gator_forward(sd, c, x[:n].cpu(), torch.float64)
    err = np.abs(verts[:n].cpu().numpy().astype(np.float64) - ref.numpy()) * 1e3
    return {'max_err_mm': float('%.3e' % err.max()), 'rms_err_mm': float('%.3e' % np.sqrt((err ** 2).mean())), 'samples': int(n),
            'coordinates': int(err.size), 'against': 'fp64 oracle (torch-CPU restatement of the reference forward), first %d samples of the timed batch' % n,
            'bar_mm': bar_mm}


def main():
    a = parse()
    in_group = 'RANK' in os.environ and 'WORLD_SIZE' in os.environ
    if a.gpus > 1 and not in_group:
        sys.exit(launch_ranks(a))
    world = int(os.environ.get('WORLD_SIZE', '1')) if in_group else 1
    if world != a.gpus:
        sys.stderr.write('bench.py: --gpus %d but WORLD_SIZE=%d; start it as `python bench.py --gpus N` or with '
                         'torch.distributed.run --nproc-per-node N\n' % (a.gpus, world))
        sys.exit(2)
    rank = int(os.environ.get('RANK', '0')) if in_group else 0
    local = int(os.environ.get('LOCAL_RANK', '0')) if in_group else 0

    import numpy as np
    import torch
    assert torch.cuda.is_available(), 'bench.py needs a HIP device (there is no CPU path)'
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    dist = None
    force_dist = os.environ.get('GATOR_BENCH_FORCE_DIST') == '1'      # exercise the N>1 code path (RCCL, side stream, `comm`) with one rank
    if world > 1 or force_dist:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', str(_free_port()))
        os.environ.setdefault('NCCL_DEBUG', 'WARN')        # keep RCCL's version banner off stdout (also when the driver launches the ranks)
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)   # backend "nccl" == RCCL on ROCm
    from gator_amd import synthetic
    from gator_amd.parallel import ShardedForward
    J, B = a.joints, a.batch
    model, base, alpha = build_model(J, a.impl, dev)
    model.precision = a.precision
    runner = ShardedForward(model, world, rank, dist, mode=a.mode, always_gather=force_dist)
    x = torch.from_numpy(synthetic.synthetic_pose2d(B, J, seed=1000 + rank)).to(dev)     # this rank's shard, resident in HBM
    target = None
    if a.mode == 'eval':      # config 5: synthetic ground-truth joints for the on-device MPJPE / PA-MPJPE sums
        target = torch.from_numpy(np.random.RandomState(7 + rank).randn(B, 17, 3).astype(np.float32) * 200).to(dev)
        runner.set_eval(synthetic.load_j_regressors()['h36m'], target)

    def sync():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def block(fn, n):
        sync()
        t0 = time.perf_counter()
        for _ in range(n):
            out = fn()
        sync()
        dt = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt, out

    for _ in range(a.warmup):
        out = runner.step(x)
    sync()
    model.profile(4)          # HIP-event brackets on every 4th timed step (the brackets themselves cost ~3 % of a step)
    dts = []
    for _ in range(max(1, a.blocks)):
        dt, out = block(lambda: runner.step(x), a.steps)
        dts.append(dt)
    # The timed region is made long enough for an outside sampler (the driver's SMI poll) to see a busy GPU: more blocks of EXACTLY
    # --steps steps, same protocol, until it holds --min-timed-s seconds.  Every rank runs the same count (dts are MAX-reduced, so equal).
    if a.min_timed_s > 0:
        more = int(min(20000, max(0.0, a.min_timed_s - sum(dts)) / max(float(np.median(dts)), 1e-6) + 0.999))
        for _ in range(more):
            dt, out = block(lambda: runner.step(x), a.steps)
            dts.append(dt)
    prof = model.profile_read()
    model.profile(0)
    dt = float(np.median(dts))
    if a.mode == 'gather':
        assert out[0].shape == (B * world, 6890, 3)
    comm = None
    if world > 1 or force_dist:   # outside the timed region: what the collective costs alone, and how much of it the overlap hides
        dt_c, _ = block(lambda: model(x), a.steps)
        dt_g, _ = block(lambda: runner.comm_only(), a.steps)
        c_ms, g_ms, t_ms = dt_c / a.steps * 1e3, dt_g / a.steps * 1e3, dt / a.steps * 1e3
        hidden = max(0.0, c_ms + g_ms - t_ms)
        comm = {'compute_ms': round(c_ms, 4), 'collective_ms': round(g_ms, 4), 'step_ms': round(t_ms, 4),
                'overlap_frac': round(min(1.0, hidden / max(min(c_ms, g_ms), 1e-9)), 3),
                'collective': 'all_gather [%d,6890,3]+[%d,%d,3] f32' % (B * world, B * world, J) if a.mode == 'gather'
                else 'all_reduce of 4 error sums',
                'bytes_per_rank': int(B * (6890 * 3 + J * 3) * 4) if a.mode == 'gather' else 32}
        # what the overlap needs from xGMI: every rank receives the other ranks' shards within one step's compute time
        comm['ingress_GBps_needed_to_hide'] = round(comm['bytes_per_rank'] * (world - 1) / max(c_ms, 1e-9) / 1e6, 1)
    if rank == 0:
        ms = dt / a.steps * 1e3
        value = B * world * a.steps / dt
        per_gpu_tf = FLOPS_PER_MESH.get(J, 4.10e8) * value / world / 1e12
        roof = None
        if prof:
            name, (tot_ms, calls) = max(prof.items(), key=lambda kv: kv[1][0])
            avg_s = tot_ms / calls * 1e-3
            stage = name.split(':')[0]
            mflop = STAGE_MFLOP.get(stage, None)
            if mflop is not None and avg_s > 0:
                ach = mflop * 1e6 * B / avg_s / 1e12
                rows, digest = pmc_digest(B)
                traffic = pmc_traffic(rows, stage)
                pipe, peak = stage_pipe(stage, a.impl, a.precision)
                designed = STAGE_BYTES.get(stage)
                fwd_bytes = pmc_forward_bytes(rows)
                row = pmc_row(rows, stage)
                executed = None
                if row and 'mfma_bf16_insts' in row:       # what the matrix pipe really executed per launch (instruction counts from the PMC pass)
                    executed = (row['mfma_bf16_insts'] + row.get('mfma_f16_insts', 0)) * 32768.0 + row.get('mfma_f32_insts', 0) * 4096.0
                roof = {'bound': 'mfma', 'kernel': name, 'achieved': round(ach, 2), 'peak': peak, 'unit': 'TFLOP/s',
                        'frac': round(ach / peak, 4), 'frac_of_all_bf16x3_ceiling_416.7': round(ach / PEAK_X3_TFLOPS, 4),
                        # executed MFMA FLOP (all planes, padding included) / live duration / 2.5 PFLOP/s dense 16-bit peak
                        'frac_of_dense_16bit_peak_2500': round(executed / avg_s / 1e12 / PEAK_BF16_TFLOPS, 4) if executed else None,
                        'speedup_over_fp32_mfma_roof_157.3': round(ach / PEAK_F32_TFLOPS, 3),
                        'traffic': traffic,
                        # the algorithmic bytes of the path are its compulsory HBM traffic, 83 kB per mesh for the WHOLE
                        # forward; the kernels exchange operand tiles through L2 / Infinity Cache on top of that (counted at the
                        # L2's memory side, so cache-resident re-reads are included)
                        'whole_forward_traffic_bytes_per_mesh': int(fwd_bytes / B) if fwd_bytes else None,
                        'algorithmic_bytes_per_mesh': BYTES_PER_MESH.get(J),
                        'traffic_ratio': round(fwd_bytes / B / BYTES_PER_MESH[J], 1) if (fwd_bytes and J in BYTES_PER_MESH) else None,
                        'kernel_designed_operand_bytes': int(designed * B) if designed else None,
                        'kernel_traffic_over_designed': round(traffic / (designed * B), 3) if (traffic and designed) else None,
                        'pmc_digest': digest,
                        'avg_launch_ms': round(avg_s * 1e3, 4), 'pipe': pipe,
                        'stages_ms': {k: round(v[0] / v[1], 4) for k, v in prof.items()},
                        'stages_note': 'HIP-event brackets on every 4th timed step; a bracketed step runs ~2 % slower, so the stages sum to slightly more than ms_per_step'}
        if roof is None:   # no per-kernel events available (bring-up path): price the whole forward
            roof = {'bound': 'mfma', 'kernel': 'whole forward', 'achieved': round(per_gpu_tf, 2), 'peak': PEAK_F32_TFLOPS,
                    'unit': 'TFLOP/s', 'frac': round(per_gpu_tf / PEAK_F32_TFLOPS, 4), 'traffic': None}
        jset = {17: 'Human3.6M 17-joint', 19: 'COCO 19-joint'}.get(J, '%d-joint' % J)
        prec = 'fp32' if a.precision == 'f32' else '16-bit operand mode (activations of the encoder and MDR layers on one fp16 plane, weights on two; vertex regressor weights on one)'
        tail = ''
        if world > 1:
            tail = (', RCCL all-gather of [%d,6890,3] vertices' % (B * world)) if a.mode == 'gather' else \
                ', on-device joint regression + MPJPE/PA-MPJPE sums, RCCL all-reduce only'
        line = {'metric': 'meshes/sec (B=%d, J=%d) GATOR forward' % (B, J), 'value': round(value, 1), 'unit': 'meshes/sec',
                'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup, 'ms_per_step': round(ms, 4), 'higher_is_better': True,
                'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32' if a.precision == 'f32' else 'f16', 'data': 'synthetic',
                # what `dtype` covers: fp32 in / out / accumulation; the products run on the 16-bit MFMA with split operands (weights exact on
                # three planes; activations, attention operands and the vertex regressor's operands on two = rounded to 22 bits).  The
                # build with no rounded operand is the `exact_split` entry of `variants`; parity of both: tests/test_gpu_x3.py.
                'arithmetic': ('fp32 values, split-precision 16-bit MFMA products: weights exact (3 planes), activations / attention / vertex-regressor operands rounded to 22 bits (2 planes); measured error of this run: `parity`'
                               if a.precision == 'f32' else
                               'gator_forward_bf16 (BASELINE config 3): fp32 in / out / accumulate / softmax / norms / GELU / residual stream; the token-wise linears of the encoder and of the three MDR layers take their '
                               'activations - and the MDR attention cores their Q, K, V and probabilities - as ONE fp16 plane (weights on two); the vertex regressor takes its weights as one plane and the coarse '
                               'vertices as two; head features, J x J attention of the encoder, lifter and tokenisers keep two planes / fp32 (profiles/r05_emulate_16bit.txt); measured error of this run: `parity` (bar: 1 mm max, 0.2 mm rms)'),
                'config': {'workload': 'B=%d synthetic %s poses per GPU, GAT+MDR forward %s%s' % (B, jset, prec, tail),
                           'baseline_config': baseline_config_of(a, world),
                           'batch_per_gpu': B, 'global_batch': B * world, 'num_joint': J, 'impl': a.impl, 'parallelism': 'dp%d' % world,
                           'mode': a.mode},
                'timing': {'blocks': len(dts), 'reported': 'median block',
                           'timed_region_s': round(sum(dts), 3),
                           'block_ms': [round(d * 1e3, 3) for d in dts] if len(dts) <= 32 else None,
                           'block_ms_quantiles': {q: round(float(np.quantile(dts, float(q))) * 1e3, 3) for q in ('0.0', '0.05', '0.25', '0.5', '0.75', '0.95', '1.0')},
                           'min_ms_per_step': round(min(dts) / a.steps * 1e3, 4), 'max_ms_per_step': round(max(dts) / a.steps * 1e3, 4)},
                'whole_forward': {'flop_per_mesh': FLOPS_PER_MESH.get(J), 'achieved_tflops_per_gpu': round(per_gpu_tf, 2),
                                  'frac_of_fp32_peak_157.3': round(per_gpu_tf / PEAK_F32_TFLOPS, 4),
                                  'frac_of_split_precision_peak_416.7': round(per_gpu_tf / PEAK_X3_TFLOPS, 4),
                                  'compulsory_bytes_per_mesh': BYTES_PER_MESH.get(J)},
                'roofline': roof}
        line['clocks'] = gpu_clocks(local)
        if comm is not None:
            line['comm'] = comm
        if world == 1 and not a.no_variants and a.impl == 'fused' and a.mode == 'gather':
            # The same workload, same process, same box, with the library's A/B switches (read when a context is created).  The
            # headline's 431x431 attention core and its vertex regressor ROUND their operands to two fp16 planes (22 bits; output
            # parity demonstrated in tests/test_gpu_x3.py); `exact_split` is the build without any rounded operand, `fp32_mfma` every product on the
            # fp32-input MFMA, `k_gat` the previous one-sample-per-workgroup encoder, `four MDR launches` the per-stage form of the
            # MDR layers (bitwise the same results as the persistent launch).
            variants = {}
            vlist = (('headline, re-measured with the variants\' protocol (5 blocks, later in the run: clocks drift)', {}),
                               ('round-5 form of the path: two tail launches, whole head in k_mdr_head (GATOR_GAT8_TAIL=0 GATOR_MDR_HEAD_PARTIALS=0)',
                                {'GATOR_GAT8_TAIL': '0', 'GATOR_MDR_HEAD_PARTIALS': '0'}),
                               ('exact_split (GATOR_MDR_X3=1 GATOR_UPSAMPLE_X3=1 GATOR_GAT8_H4=0 GATOR_GAT_TILED_H4=0)',
                                {'GATOR_MDR_X3': '1', 'GATOR_UPSAMPLE_X3': '1', 'GATOR_GAT8_H4': '0', 'GATOR_GAT_TILED_H4': '0'}),
                               ('six-product encoder (GATOR_GAT8_H4=0)', {'GATOR_GAT8_H4': '0'}),
                               ('three-plane vertex regressor (GATOR_UPSAMPLE_X3=1)', {'GATOR_UPSAMPLE_X3': '1'}),
                               ('fp32_mfma (GATOR_GAT_X3=0 GATOR_MDR_X3=0 GATOR_UPSAMPLE_X3=0)', {'GATOR_GAT_X3': '0', 'GATOR_MDR_X3': '0', 'GATOR_UPSAMPLE_X3': '0'}),
                               ('four MDR launches instead of the persistent one (GATOR_MDR_PERSIST=0)', {'GATOR_MDR_PERSIST': '0'}),
                               ('k_gat encoder (GATOR_GAT8=0)', {'GATOR_GAT8': '0'}))
            if a.precision != 'f32':      # config 3: what the 16-bit mode buys -- the fp32 build at the same shape, same box, same process
                vlist = (vlist[0],
                         ('fp32 build at this shape (gator_forward_f32)', {'_precision': 'f32'}),
                         ('16-bit MDR layers only (GATOR_C3_ENCODER=0 GATOR_C3_UPSAMPLE_W1=0)', {'GATOR_C3_ENCODER': '0', 'GATOR_C3_UPSAMPLE_W1': '0'}),
                         ('round 4 form of config 3: bf16 vertex regressor only (GATOR_C3_MDR=0 GATOR_C3_ENCODER=0 GATOR_C3_UPSAMPLE_BF16=1)',
                          {'GATOR_C3_MDR': '0', 'GATOR_C3_ENCODER': '0', 'GATOR_C3_UPSAMPLE_BF16': '1'}),
                         ('four MDR launches instead of the persistent one (GATOR_MDR_PERSIST=0)', {'GATOR_MDR_PERSIST': '0'}))
            for vname, env in vlist:
                env = dict(env)
                vprec = env.pop('_precision', a.precision)
                old = {k: os.environ.get(k) for k in env}
                os.environ.update(env)
                try:
                    mv, _, _ = build_model(J, a.impl, dev)
                    mv.precision = vprec
                    for _ in range(a.warmup):
                        mv(x)
                    dv = sorted(block(lambda: mv(x), a.steps)[0] for _ in range(5))[2]
                    variants[vname] = {'value': round(B * a.steps / dv, 1), 'ms_per_step': round(dv / a.steps * 1e3, 4)}
                    del mv
                finally:
                    for k, v in old.items():
                        if v is None:
                            os.environ.pop(k, None)
                        else:
                            os.environ[k] = v
            # the headline forward captured once into a hipGraph (torch.cuda.CUDAGraph around model(x)) and replayed: what a caller with a
            # fixed batch shape gets by removing the host side of the four launches (results are bitwise the eager ones: tests/test_gpu_fullsize.py)
            try:
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    model(x)
                torch.cuda.current_stream().wait_stream(side)
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph, stream=side):
                    model(x)
                for _ in range(a.warmup):
                    graph.replay()
                dv = sorted(block(lambda: graph.replay(), a.steps)[0] for _ in range(5))[2]
                variants['headline replayed from a hipGraph (torch.cuda.CUDAGraph)'] = {'value': round(B * a.steps / dv, 1), 'ms_per_step': round(dv / a.steps * 1e3, 4)}
                del graph
            except Exception as e:       # a box whose runtime refuses the capture does not invalidate the line
                variants['headline replayed from a hipGraph (torch.cuda.CUDAGraph)'] = {'error': str(e)[:200]}
            # the same through the library's own switch (gator_set_graph_replay: the forward is captured the second time it is seen with the
            # same batch and tensors, then one hipGraphLaunch per call) -- what a caller of the C ABI gets without torch
            try:
                mg, _, _ = build_model(J, a.impl, dev)
                mg.precision = a.precision
                mg.set_graph_replay(True)
                og = (torch.empty(B, 6890, 3, device=dev), torch.empty(B, J, 3, device=dev))
                for _ in range(max(a.warmup, 3)):
                    mg(x, out=og)
                dv = sorted(block(lambda: mg(x, out=og), a.steps)[0] for _ in range(5))[2]
                variants['headline with the library replaying the forward from a hipGraph (gator_set_graph_replay)'] = {
                    'value': round(B * a.steps / dv, 1), 'ms_per_step': round(dv / a.steps * 1e3, 4), 'graph_launches': mg.graph_launches()}
                del mg
            except Exception as e:
                variants['headline with the library replaying the forward from a hipGraph (gator_set_graph_replay)'] = {'error': str(e)[:200]}
            line['variants'] = variants
            f32v = variants.get('fp32 build at this shape (gator_forward_f32)')
            if f32v and 'value' in f32v:      # config 3's own bar (round-4 review): >= 1.6 x the fp32 build on the same box
                line['vs_fp32_build_same_box'] = round(value / f32v['value'], 3)
        if world == 1 and a.config == 0 and not a.no_config3 and a.impl == 'fused' and a.mode == 'gather' and (B, J, a.precision) == (256, 17, 'f32'):
            # BASELINE configs[2] (B = 2048 COCO 19-joint, 16-bit operand mode) beside the default headline, so that the driver's own run carries it
            # (round-5 review): same process, same box, the variants' protocol (median of 5 blocks of --steps steps), the fp32 build at the same
            # shape next to it, parity of 8 samples of the timed batch against the fp64 oracle.  `python bench.py --config 3` is the full line.
            try:
                c3 = BASELINE_CONFIGS[3]
                B3, J3 = c3['batch'], c3['joints']
                m3, base3, alpha3 = build_model(J3, a.impl, dev)
                x3 = torch.from_numpy(synthetic.synthetic_pose2d(B3, J3, seed=1003)).to(dev)
                rec = {'workload': c3['name'], 'dtype': 'f16', 'protocol': 'median of 5 blocks of %d steps after %d warm-up steps' % (a.steps, a.warmup)}
                for prec, key in (('bf16', None), ('f32', 'fp32_build_same_shape')):
                    m3.precision = prec
                    for _ in range(a.warmup):
                        o3 = m3(x3)
                    dv = sorted(block(lambda: m3(x3), a.steps)[0] for _ in range(5))[2]
                    r = {'value': round(B3 * a.steps / dv, 1), 'unit': 'meshes/sec', 'ms_per_step': round(dv / a.steps * 1e3, 4)}
                    if key is None:
                        rec.update(r)
                        if not a.no_cpu_baseline:
                            m3.precision = 'bf16'
                            rec['parity'] = parity_check(m3, base3, alpha3, J3, x3, m3(x3)[0], n=8, bar_mm=1.0)
                    else:
                        rec[key] = r
                rec['vs_fp32_build_same_box'] = round(rec['value'] / rec['fp32_build_same_shape']['value'], 3)
                line['config3'] = rec
                del m3, x3
            except Exception as e:       # the sub-record must never cost the headline
                line['config3'] = {'error': str(e)[:300]}
        if world == 1 and B >= 128 and a.subbatch_variant:
            # same workload with the library's sub-batch pipelining (two half-batches on two streams; bit-identical results).
            # Reported beside the headline, not as it: concurrent streams make per-kernel durations (and so `roofline`) ambiguous.
            m2, _, _ = build_model(J, a.impl, dev)
            m2.precision = a.precision
            m2.subbatch_streams = 2
            for _ in range(a.warmup):
                m2(x)
            d2 = sorted(block(lambda: m2(x), a.steps)[0] for _ in range(5))[2]
            line['subbatch_streams_2'] = {'value': round(B * a.steps / d2, 1), 'ms_per_step': round(d2 / a.steps * 1e3, 4)}
        if world == 1 and not a.no_cpu_baseline:
            if a.mode == 'gather':
                line['parity'] = parity_check(model, base, alpha, J, x, out[0], bar_mm=1e-3 if a.precision == 'f32' else 1.0)
            line['cpu_baseline'] = cpu_baseline(model, base, alpha, J)
        result = json.dumps(line)
    else:
        result = None
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if result is not None:
        # RCCL writes its version banner to C stdio, which is flushed at exit - AFTER a Python print.  Flush it first so that the
        # result is the LAST line on stdout.
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(result, flush=True)


if __name__ == '__main__':
    main()
