"""GATBlock of the training step as ONE launch per direction (gator_t_gat_block_fwd / _bwd, csrc/train_gat.inc): a workgroup owns
a sample and walks the block's operations itself.  Same arithmetic, same Philox masks and the same parameter / gradient plumbing as
the block composed from primitives (train/model.py: gat_block), which stays as the cross-check."""
import ctypes

import torch

from .. import _lib
from . import ops

C, H, HID, X1 = 128, 8, 512, 16


def _ptr(t):
    return t.data_ptr() if t is not None else None


class _GatBlock(torch.autograd.Function):
    NPARAM = 21

    @staticmethod
    def forward(ctx, x, hop_bias, adj, consts, gen, rates6, training, *params):
        B, J, _ = x.shape
        dev = x.device
        e = lambda *shape: torch.empty(shape, device=dev, dtype=torch.float32)
        a = _lib.GatBlock()
        a.B, a.J = B, J
        xc = ops._contig(x)
        hb, ad = ops._contig(hop_bias), ops._contig(adj)
        ins = [xc, hb, ad, consts.m1, consts.m2] + [ops._contig(p) for p in params]
        for name, t in zip(_lib.GatBlock.IN, ins):
            setattr(a, name, _ptr(t))
        saved = dict(y=e(B, J, C), qkv=e(B, J, 3 * C), P=e(B, H, J, J), a0=e(B, J, C), h0=e(B, J, C), h1=e(B, J, C), s=e(B, J, C), cat=e(B, J, C + X1),
                     x1=e(B, J, C), y2=e(B, J, C), hpre=e(B, J, HID), hd=e(B, J, HID), x2=e(B, J, C), stats=e(B, J, 4))
        scratch = dict(t0=e(B, J, C), t1=e(B, J, C), t2=e(B, J, X1), t3=e(B, H, J, J))
        for k, t in list(saved.items()) + list(scratch.items()):
            setattr(a, k, _ptr(t))
        a.seed = gen.seed
        a.counter = gen.counter_ptr()
        offs = []
        for i, r in enumerate(rates6):                      # offsets drawn in the composed block's order, only for active sites
            on = training and r > 0.0
            offs.append(gen.next_offset() if on else 0)
            a.off[i] = offs[-1]
            a.rate[i] = float(r) if on else 0.0
        _lib.check(_lib.load().gator_t_gat_block_fwd(ctypes.byref(a), ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)), 'gator_t_gat_block_fwd')
        ctx.args, ctx.ins, ctx.saved, ctx.params = a, ins, saved, params
        ctx.slots = [ops.grad_slot(p) for p in params]
        return saved['x2']

    @staticmethod
    def backward(ctx, g):
        a, saved, params, slots = ctx.args, ctx.saved, ctx.params, ctx.slots
        B, J = a.B, a.J
        R = B * J
        dev = g.device
        e = lambda *shape: torch.empty(shape, device=dev, dtype=torch.float32)
        gc = ops._contig(g)
        out = dict(dx=e(B, J, C), dqkv=e(B, J, 3 * C), da1=e(B, J, C), dh0=e(B, J, C), dh1=e(B, J, C), dl0=e(B, J, C), dl1=e(B, J, X1), dxf=e(B, J, C),
                   dhpre=e(B, J, HID), dm=e(B, J, C), dgout=e(B, J, C), pm=e(B, J, C), dadj=e(B, J, J), dS=e(B, H, J, J), ln1_gw=e(B, J, C),
                   ln1_gb=e(B, J, C), ln2_gw=e(B, J, C), ln2_gb=e(B, J, C), u0=e(B, J, C), u1=e(1), u2=e(1), u3=e(B, J, HID), u4=e(B, H, J, J))
        a.dx2 = gc.data_ptr()
        for k, t in out.items():
            setattr(a, k, t.data_ptr())
        scratch_pd = e(B, H, J, J)
        a.t3 = scratch_pd.data_ptr()
        _lib.check(_lib.load().gator_t_gat_block_bwd(ctypes.byref(a), ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)), 'gator_t_gat_block_bwd')
        # ---- parameter gradients: the products and column sums join the grouped launch when they land in the flat gradient buffer
        grads = [None] * len(params)
        ones = lambda n: ops._one(dev).as_strided((1, 1, 1, n), (0, 0, 0, 0))
        r2 = lambda t: t.reshape(1, 1, R, t.shape[-1])

        def weight(i_w, i_b, x_rows, dy_rows):            # F.linear weight [N_out, K_in] (+ bias): dW = dY^T X, db = colsum(dY)
            w, ws, bs = params[i_w], slots[i_w], (slots[i_b] if i_b is not None else None)
            if ws is not None and (i_b is None or bs is not None):
                ops.Deferred.add(r2(dy_rows).transpose(2, 3), r2(x_rows), ws.view(1, 1, w.shape[0], w.shape[1]), bs)
                grads[i_w] = ws
                if i_b is not None:
                    grads[i_b] = bs
            else:
                gb = e(w.shape[0]) if i_b is not None else None
                grads[i_w] = ops.raw_gemm(r2(dy_rows).transpose(2, 3), r2(x_rows), a_rowsum=gb).reshape(w.shape)
                if i_b is not None:
                    grads[i_b] = gb

        def colsum(i_p, rows2d):                          # parameter gradient = column sums of a [rows, n] matrix
            sl = slots[i_p]
            n = rows2d.shape[1]
            if sl is not None:
                ops.Deferred.add(ones(rows2d.shape[0]), rows2d.reshape(1, 1, rows2d.shape[0], n), sl.view(1, 1, 1, n), None)
                grads[i_p] = sl
            else:
                grads[i_p] = ops.raw_sum(rows2d, [0]).reshape(params[i_p].shape)

        s_ = saved
        colsum(0, out['ln1_gw'].view(R, C)); colsum(1, out['ln1_gb'].view(R, C))
        weight(2, 3, s_['y'], out['dqkv'])
        weight(4, 5, s_['a0'], out['da1'])
        # gcn.W [2, C_in, C_out] used as x @ W: dW[k] = y^T dh_k
        wsl = slots[6]
        if wsl is not None:
            ops.Deferred.add(r2(s_['y']).transpose(2, 3), r2(out['dh0']), wsl.narrow(0, 0, 1).view(1, 1, C, C), None)
            ops.Deferred.add(r2(s_['y']).transpose(2, 3), r2(out['dh1']), wsl.narrow(0, 1, 1).view(1, 1, C, C), None)
            grads[6] = wsl
        else:
            gw = e(2, C, C)
            ops.raw_gemm(r2(s_['y']).transpose(2, 3), r2(out['dh0']), out=gw.narrow(0, 0, 1).view(1, 1, C, C))
            ops.raw_gemm(r2(s_['y']).transpose(2, 3), r2(out['dh1']), out=gw.narrow(0, 1, 1).view(1, 1, C, C))
            grads[6] = gw
        colsum(7, out['pm'].view(B, J * C))
        colsum(8, out['dgout'].view(R, C))
        weight(9, 10, s_['s'], out['dl0'])
        weight(11, 12, s_['s'], out['dl1'])
        weight(13, 14, s_['cat'], out['dxf'])
        colsum(15, out['ln2_gw'].view(R, C)); colsum(16, out['ln2_gb'].view(R, C))
        weight(17, 18, s_['y2'], out['dhpre'])
        weight(19, 20, s_['hd'], out['dm'])
        ctx.keep = (out, scratch_pd, gc)                   # operands of the queued products stay alive until the backward ends
        dhop = ops.raw_sum(out['dS'], [0]) if ctx.needs_input_grad[1] else None
        dadj = ops.raw_sum(out['dadj'], [0]) if ctx.needs_input_grad[2] else None
        return (out['dx'], dhop, dadj, None, None, None, None) + tuple(grads)


PARAM_KEYS = ('norm1.weight', 'norm1.bias', 'attn.qkv.weight', 'attn.qkv.bias', 'attn.proj.weight', 'attn.proj.bias', 'gcn.W', 'gcn.M', 'gcn.bias',
              'x_feat.linears.0.weight', 'x_feat.linears.0.bias', 'x_feat.linears.1.weight', 'x_feat.linears.1.bias', 'x_feat.linearback.weight',
              'x_feat.linearback.bias', 'norm2.weight', 'norm2.bias', 'mlp.fc1.weight', 'mlp.fc1.bias', 'mlp.fc2.weight', 'mlp.fc2.bias')


def gat_block(P, c, x, bias, i, gen, rates, training=True, p='pose_lifter.'):
    from .model import sym_adjacency
    b = p + 'blocks.%d.' % i
    adj = sym_adjacency(c, P[b + 'gcn.adj2'])
    rates6 = (rates.gat_attn, rates.gat_proj, rates.gat_path[i], rates.gat_mlp, rates.gat_mlp, rates.gat_path[i])
    return _GatBlock.apply(x, bias, adj, c, gen, rates6, training, *[P[b + k] for k in PARAM_KEYS])
