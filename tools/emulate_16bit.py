"""CPU study for BASELINE config 3 ("full GATOR forward bf16"): which operands of the forward tolerate ONE 16-bit plane?

The fp64 oracle is run under a TorchFunctionMode that rounds the operands of every product (linear, attention core, adjacency /
hop aggregation, the two convolutions) to a chosen number of significant bits (11 = one fp16 plane, 8 = one bf16 plane, 22 = the
two-plane fp16 split the fp32 configuration ships; exponent range unlimited: the kernels pre-scale their operands into range).
Everything else (accumulation, softmax, norms, GELU, residual stream) stays fp64, so the figures are the floor of what a kernel
with that operand form and fp32 accumulation can reach.  Output: max / rms vertex error (mm) and the mean joint error of the
regressed joints (mm) against the clean fp64 forward, per policy.

    python tools/emulate_16bit.py [--samples 256] [--variant coco19_alpha] [--policies all]

Dev-container tool (imports the oracle); results under profiles/r05_emulate_16bit.txt.
"""
import argparse
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F
from torch.overrides import TorchFunctionMode

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gator_amd import synthetic          # noqa: E402
from oracle import gator_oracle as go    # noqa: E402
from tests.helpers import oracle_setup   # noqa: E402


def rnd(x, bits):
    """round to `bits` significant bits (round-to-nearest-even on the significand), exponent unlimited"""
    if bits is None or bits >= 53:
        return x
    m, e = torch.frexp(x)
    s = float(2 ** bits)
    return torch.ldexp(torch.round(m * s) / s, e)


class Policy:
    """bits per operand class; None = exact.  Classes:
    lin_a / lin_w   token-wise linears (F.linear and the x @ W forms of GraphLinear / MGCN)
    qk              both operands of every Q.K^T            pv_p / pv_v   probabilities / values of every P.V
    adj             the J x J adjacency and hop-mask aggregations of the encoder (both operands; the 0/1 masks are exact anyway)
    head            softmax(A) @ B of the MDR head and bias_conv1d
    up_a / up_w     the vertex regressor (upsample_conv): coarse vertices / weights
    """
    KEYS = ('lin_a', 'lin_w', 'qk', 'pv_p', 'pv_v', 'adj', 'head', 'up_a', 'up_w',
            'gat_a', 'mdr_a', 'hd_a', 'mlp_a',     # optional overrides of lin_a by site: encoder / MDR layers / head + lifter + tokenisers / MLP hidden (fc2 input)
            'gat_core')                             # optional override of qk / pv_p / pv_v for the encoder's J x J attention (head dim 16)

    def __init__(self, name, **kw):
        self.name = name
        self.b = {k: kw.get(k, 'x' if k in ('gat_a', 'mdr_a', 'hd_a', 'mlp_a', 'gat_core') else None) for k in self.KEYS}


class Emu(TorchFunctionMode):
    def __init__(self, pol):
        super().__init__()
        self.p = pol.b

    def act_bits(self, w):
        p = self.p
        o, i = int(w.shape[0]), int(w.shape[1])
        site = 'hd_a'
        if (o, i) in ((256, 64), (64, 256), (512, 128), (128, 512)) and i in (256, 512):
            site = 'mlp_a'
        elif i in (128, 144, 512) and o in (384, 128, 16, 512):
            site = 'gat_a'
        elif i in (64, 256) and o in (64, 256):
            site = 'mdr_a'
        return p[site] if p.get(site, 'x') != 'x' else p['lin_a']

    def __torch_function__(self, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        p = self.p
        if func is F.linear:
            x, w = args[0], args[1]
            rest = args[2:]
            return func(rnd(x, self.act_bits(w)), rnd(w, p['lin_w']), *rest, **kwargs)
        if func in (torch.matmul, torch.Tensor.matmul, torch.Tensor.__matmul__, torch.bmm, torch.Tensor.bmm):
            a, b = args[0], args[1]
            if a.dim() == 4 and b.dim() == 4:                 # attention cores [B,H,N,d] @ [B,H,d,M] / [B,H,N,M] @ [B,H,M,d]
                gat = a.shape[1] == 8 and p.get('gat_core', 'x') != 'x'      # the encoder has 8 heads, the MDR layers 2
                if a.shape[-1] in (16, 32) and b.shape[-2] == a.shape[-1] and b.shape[-1] != a.shape[-1]:
                    return func(rnd(a, p['gat_core'] if gat else p['qk']), rnd(b, p['gat_core'] if gat else p['qk']))
                return func(rnd(a, p['gat_core'] if gat else p['pv_p']), rnd(b, p['gat_core'] if gat else p['pv_v']))
            if a.dim() == 3 and b.dim() == 3 and a.shape[-1] == 20:      # head: softmax(mat_A) @ mat_B
                return func(rnd(a, p['head']), rnd(b, p['head']))
            if a.dim() <= 3 and a.shape[-1] == a.shape[-2] and a.shape[-1] in (17, 19):   # adjacency / hop masks [J,J] @ [B,J,C]
                return func(rnd(a, p['adj']), rnd(b, p['adj']))
            if b.dim() == 2 or (a.dim() == 3 and a.shape[0] == 1):      # x @ W (MGCN) / W[None] @ x (GraphLinear)
                wa = a.dim() == 3 and a.shape[0] == 1
                return func(rnd(a, p['lin_w'] if wa else p['lin_a']), rnd(b, p['lin_a'] if wa else p['lin_w']))
            raise RuntimeError('unclassified product %s %s' % (tuple(a.shape), tuple(b.shape)))
        if func is F.conv1d:
            x, w = args[0], args[1]
            rest = args[2:]
            if w.shape[0] == 6890:
                return func(rnd(x, p['up_a']), rnd(w, p['up_w']), *rest, **kwargs)
            return func(rnd(x, p['head']), rnd(w, p['head']), *rest, **kwargs)
        return func(*args, **kwargs)


def policies():
    F16, BF, X2 = 11, 8, 22
    allk = lambda b: {k: b for k in Policy.KEYS if (not k.endswith('_a') or k in ('lin_a', 'up_a')) and k != 'gat_core'}
    P = [
        Policy('shipped fp32 config: every operand 22 bits (weights of linears exact)', **dict(allk(X2), lin_w=None)),
        Policy('ALL operands one fp16 plane (11 bits)', **allk(F16)),
        Policy('ALL operands one bf16 plane (8 bits)', **allk(BF)),
        Policy('activations fp16, all weights exact', lin_a=F16, qk=F16, pv_p=F16, pv_v=F16, adj=F16, head=F16, up_a=F16),
        Policy('only the vertex regressor: both operands fp16', up_a=F16, up_w=F16),
        Policy('only the vertex regressor: weights fp16, coarse vertices 22 bits', up_a=X2, up_w=F16),
        Policy('only the vertex regressor: both bf16 (the round-4 "config 3")', up_a=BF, up_w=BF),
        Policy('only attention cores fp16 (Q, K, P, V)', qk=F16, pv_p=F16, pv_v=F16),
        Policy('only P and V fp16', pv_p=F16, pv_v=F16),
        Policy('only P fp16', pv_p=F16),
        Policy('only Q, K fp16', qk=F16),
        Policy('only linears: activations fp16, weights exact', lin_a=F16),
        Policy('only linears: activations 22 bits, weights fp16', lin_a=X2, lin_w=F16),
        Policy('only linears: both fp16', lin_a=F16, lin_w=F16),
        Policy('cores fp16 + regressor weights fp16 (a 22), rest 22 / exact weights', **dict(allk(X2), lin_w=None, qk=F16, pv_p=F16, pv_v=F16, up_w=F16)),
        Policy('P, V fp16 + regressor weights fp16 (a 22), rest 22 / exact weights', **dict(allk(X2), lin_w=None, pv_p=F16, pv_v=F16, up_w=F16)),
        Policy('P, V fp16, rest 22 / exact weights', **dict(allk(X2), lin_w=None, pv_p=F16, pv_v=F16)),
        Policy('P fp16 only, rest 22 / exact weights', **dict(allk(X2), lin_w=None, pv_p=F16)),
        Policy('linear weights 22 bits (two planes) instead of exact, rest shipped', **dict(allk(X2))),
        # config-3 candidates: base = activations fp16 in linears + cores fp16; weights 22 bits; regressor two-plane (22 | 22)
        Policy('C3a: lin acts fp16, cores fp16, weights 22, regressor 22|22', **dict(allk(X2), lin_a=F16, qk=F16, pv_p=F16, pv_v=F16)),
        Policy('C3b: = C3a but encoder linears keep 22-bit activations', **dict(allk(X2), lin_a=F16, gat_a=X2, qk=F16, pv_p=F16, pv_v=F16)),
        Policy('C3c: = C3a but MDR-layer linears keep 22-bit activations', **dict(allk(X2), lin_a=F16, mdr_a=X2, qk=F16, pv_p=F16, pv_v=F16)),
        Policy('C3d: = C3a but head / lifter / tokenisers keep 22-bit activations', **dict(allk(X2), lin_a=F16, hd_a=X2, qk=F16, pv_p=F16, pv_v=F16)),
        Policy('C3e: = C3a but MLP hidden (fc2 input) keeps 22 bits', **dict(allk(X2), lin_a=F16, mlp_a=X2, qk=F16, pv_p=F16, pv_v=F16)),
        Policy('C3f: only MLP hidden fp16 + cores fp16, rest 22', **dict(allk(X2), mlp_a=F16, qk=F16, pv_p=F16, pv_v=F16)),
        Policy('C3g: = C3a + regressor weights fp16 (a 22)', **dict(allk(X2), lin_a=F16, qk=F16, pv_p=F16, pv_v=F16, up_w=F16)),
        Policy('C3h: = C3a with bf16 instead of fp16', **dict(allk(X2), lin_a=BF, qk=BF, pv_p=BF, pv_v=BF)),
        # what round 5 ships as gator_forward_bf16: encoder and MDR-layer linears on one activation plane, MDR attention cores on one plane, head
        # features / lifter / tokenisers and the encoder's J x J attention on two (or exact), weights 22 bits, regressor: weights one plane, coarse vertices two
        Policy('C3 shipped: linears (encoder + MDR) acts fp16, MDR cores fp16, head 22, regressor w fp16 | a 22', **dict(allk(X2), lin_a=F16, hd_a=X2, qk=F16, pv_p=F16, pv_v=F16, gat_core=X2, up_w=F16)),
    ]
    return P


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--samples', type=int, default=256)
    ap.add_argument('--variant', default='coco19_alpha')
    ap.add_argument('--seed', type=int, default=None)
    ap.add_argument('--only', default='')
    a = ap.parse_args()
    torch.set_num_threads(os.cpu_count())
    z, c, sd = oracle_setup(a.variant, seed=a.seed)
    x = torch.from_numpy(synthetic.synthetic_pose2d(a.samples, c.J, seed=11))
    ref, _ = go.gator_forward(sd, c, x, torch.float64)
    ref32, _ = go.gator_forward(sd, c, x, torch.float32)
    jr = torch.from_numpy(np.asarray(synthetic.model_j_regressor(17), np.float64))      # the h36m evaluation regressor (lib/core/base.py:221)

    def report(name, v):
        e = (v.double() - ref).abs() * 1e3
        j0, j1 = torch.matmul(jr[None], v.double() * 1e3), torch.matmul(jr[None], ref * 1e3)
        j0, j1 = j0 - j0[:, :1], j1 - j1[:, :1]
        mpj = (j0 - j1).norm(dim=2).mean()
        print('%-82s max %9.3e  rms %9.3e  joint err %9.3e mm' % (name, float(e.max()), float((e ** 2).mean().sqrt()), float(mpj)), flush=True)

    print('variant %s, %d samples, vs clean fp64 oracle' % (a.variant, a.samples))
    report('reference arithmetic (fp32 oracle)', ref32)
    for pol in policies():
        if a.only and a.only not in pol.name:
            continue
        with Emu(pol):
            v, _ = go.gator_forward(sd, c, x, torch.float64)
        report(pol.name, v)


if __name__ == '__main__':
    main()
