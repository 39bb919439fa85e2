"""Differentiable fp32 primitives of the training row on the HIP kernels of include/gator_train.h.

torch is the plumbing here: tensors own device memory, views (reshape / permute / expand) are metadata, and
``torch.autograd.Function`` orders the backward calls.  Every arithmetic operation - forward and backward - is a kernel of
libgator_hip.so; nothing in this module calls an aten compute op on the data path, and there is no CPU path: inputs must live on
a HIP device.  The reference gets the same operations from aten through autograd (lib/core/base.py:135-153)."""
import ctypes

import torch

from .. import _lib

_I64x4 = ctypes.c_int64 * 4
_I32x4 = ctypes.c_int32 * 4
_I64x2 = ctypes.c_int64 * 2
ADD, SUB, MUL, DIV = 0, 1, 2, 3
U_AFFINE, U_GELU, U_DGELU, U_EXP, U_RSQRT, U_SQRT, U_RECIP, U_ABS, U_SIGN, U_POWBASE, U_SQUARE, U_GT = range(12)


def _stream(t):
    return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def _need_device(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError('gator_amd.train: tensors must live on a HIP device (there is no CPU path)')
        if t is not None and t.dtype != torch.float32:
            raise RuntimeError('gator_amd.train: float32 only, got %s' % t.dtype)


def _v4(t):
    """(shape4, stride4) of a tensor with <= 4 dims, left-padded."""
    d = t.dim()
    if d > 4:
        raise ValueError('gator_amd.train: at most 4 dims (got %d): reshape first' % d)
    return [1] * (4 - d) + list(t.shape), [0] * (4 - d) + list(t.stride())


def _call(name, *args):
    _lib.check(getattr(_lib.load(), name)(*args), name)


# ------------------------------------------------------------------------------------------------ raw (non-differentiable) kernels
def raw_binary(op, a, b, out=None):
    shape = torch.broadcast_shapes(a.shape, b.shape)
    ae, be = a.expand(shape), b.expand(shape)
    if out is None:
        out = torch.empty(shape, device=a.device, dtype=torch.float32)
    n4, sa = _v4(ae)
    _, sb = _v4(be)
    _, so = _v4(out)
    _call('gator_t_binary', op, ae.data_ptr(), _I64x4(*sa), be.data_ptr(), _I64x4(*sb), out.data_ptr(), _I64x4(*so), _I64x4(*n4), _stream(a))
    return out


def raw_unary(op, x, p0=0.0, p1=0.0, out=None):
    if out is None:
        out = torch.empty(x.shape, device=x.device, dtype=torch.float32)
    n4, sx = _v4(x)
    _, so = _v4(out)
    _call('gator_t_unary', op, x.data_ptr(), _I64x4(*sx), out.data_ptr(), _I64x4(*so), _I64x4(*n4), float(p0), float(p1), _stream(x))
    return out


def zeros(shape, device):
    """A zero-filled tensor without an aten fill kernel: (x > +inf) is 0 for every bit pattern of the fresh memory."""
    out = torch.empty(shape, device=device, dtype=torch.float32)
    if out.numel():
        raw_unary(U_GT, out.view(-1), float('inf'), 0.0, out=out.view(-1))
    return out


def raw_sum(x, dims, keepdim=False, out=None, accumulate=False):
    dims = sorted(d % x.dim() for d in dims)
    n4, sx = _v4(x)
    pad = 4 - x.dim()
    red = [0] * 4
    for d in dims:
        red[pad + d] = 1
    kept = [1 if (i in dims) else x.shape[i] for i in range(x.dim())]
    if out is None:
        out = torch.empty(kept, device=x.device, dtype=torch.float32)
    lib = _lib.load()
    nbytes = int(lib.gator_t_reduce_ws_bytes(_I64x4(*n4), _I32x4(*red)))
    ws = torch.empty(max(nbytes, 8), device=x.device, dtype=torch.uint8)
    _call('gator_t_reduce_sum', x.data_ptr(), _I64x4(*sx), _I64x4(*n4), _I32x4(*red), out.data_ptr(), int(accumulate), ws.data_ptr(), _stream(x))
    if not keepdim:
        out = out.reshape([x.shape[i] for i in range(x.dim()) if i not in dims])
    return out


def sum_to(g, shape):
    """Reduce a broadcast gradient back to `shape` (left-padded broadcasting rules)."""
    shape = list(shape)
    if list(g.shape) == shape:
        return g
    lead = g.dim() - len(shape)
    dims = list(range(lead)) + [lead + i for i, n in enumerate(shape) if n == 1 and g.shape[lead + i] != 1]
    if not dims:
        return g.reshape(shape)
    return raw_sum(g, dims, keepdim=True).reshape(shape)


def raw_gemm(a, b, out=None, bias=None, alpha=1.0, accumulate=False, a_rowsum=None):
    """a [n1,n2,M,K] x b [n1,n2,K,N] (any strides, stride 0 = broadcast) -> out [n1,n2,M,N].  a_rowsum [M]: also alpha * a.sum(K)."""
    n1, n2, M, K = a.shape
    N = b.shape[3]
    if out is None:
        out = torch.empty((n1, n2, M, N), device=a.device, dtype=torch.float32)
    sa, sb, so = a.stride(), b.stride(), out.stride()
    ksplit, ws = 1, None
    tiles = ((M + 63) // 64) * ((N + 63) // 64)
    if n1 * n2 == 1 and K >= 512 and tiles < 256:          # few output tiles, long K: slices of >= 128 fill the chip
        ksplit = max(1, min(64, K // 128, 1024 // tiles))
        if ksplit > 1:
            ws = torch.empty(ksplit * (M * N + M), device=a.device, dtype=torch.float32)
    _call('gator_t_gemm', a.data_ptr(), b.data_ptr(), out.data_ptr(), M, N, K, _I64x2(sa[2], sa[3]), _I64x2(sb[2], sb[3]), _I64x2(so[2], so[3]),
          n1, n2, _I64x2(sa[0], sa[1]), _I64x2(sb[0], sb[1]), _I64x2(so[0], so[1]), bias.data_ptr() if bias is not None else None, float(alpha),
          int(accumulate), ksplit, ws.data_ptr() if ws is not None else None, a_rowsum.data_ptr() if a_rowsum is not None else None, _stream(a))
    return out


def grad_slot(p, part=None):
    """The slice of the flat gradient buffer that belongs to parameter view `p` (set by optim.FlatParams.views), or None.

    A slot-aware op WRITES its weight gradient into that slice (deferred, accumulate = 0) and hands the slice itself back to autograd,
    so a view may feed exactly ONE such op per step (or several that each own a disjoint `part` of it, like the two rows of the MGCN
    weight): a second writer would overwrite the first, and autograd would sum an unwritten buffer.  That used to hold by convention;
    it is checked here - the views are made afresh every step, so the marks do not outlive it."""
    sl = getattr(p, '_gslot', None)
    if sl is None:
        return None
    used = p.__dict__.setdefault('_gslot_used', set())
    if None in used or part in used or (part is None and used):
        raise RuntimeError('gator_amd.train: a parameter view feeds two gradient-slot-aware ops in one step (its slice of the flat gradient '
                           'would be written twice); route one of the uses through a plain op or give each its own part')
    used.add(part)
    return sl


def raw_copy(x):
    """Contiguous copy through the library's strided copy kernel."""
    if x.dim() > 4:
        raise ValueError('raw_copy: at most 4 dims')
    return raw_unary(U_AFFINE, x, 1.0, 0.0)


def _contig(x):
    return x if x.is_contiguous() else raw_copy(x)


# ------------------------------------------------------------------------------------------------ deferred weight gradients
_ONES = {}


def _one(dev):
    """A device scalar 1.0, used with stride 0 as the all-ones operand that turns a column sum into a row of a grouped GEMM."""
    t = _ONES.get(dev)
    if t is None:
        t = _ONES[dev] = raw_unary(U_AFFINE, zeros((1,), dev), 0.0, 1.0)
    return t


class Deferred:
    """Weight / bias gradients are LEAVES of the backward graph: nothing reads them before the optimiser.  Backward nodes queue
    them here instead of launching each between two links of the dependent activation-gradient chain, and `flush()` - called by
    the last backward node, the parameter scatter - runs the whole list as ONE grouped GEMM launch (+ one split-K finish):
    ~70 small products that fill the chip together instead of ~230 latency-bound launches on the critical path."""
    enabled = True
    queue = []
    retired = []
    pool = None        # pinned host memory for the problem tables of CAPTURED steps (a replayed graph re-reads its table); allocated
    pool_used = 0      # outside any capture (page-locking is not capturable) and carved up without reuse

    @staticmethod
    def add(a4, b4, out4, rowsum):
        Deferred.queue.append((a4, b4, out4, rowsum))

    @staticmethod
    def flush(dev):
        """Runs the queued products of device `dev` (two trainers on two devices interleaved in one process keep their lists apart)."""
        dev = torch.device(dev)
        mine = [e for e in Deferred.queue if e[2].device == dev]
        if not mine:
            return
        Deferred.queue[:] = [e for e in Deferred.queue if e[2].device != dev]
        run_group([(a4, b4, out4, rowsum, None) for (a4, b4, out4, rowsum) in mine], dev)


def run_group(problems, dev):
    """One grouped GEMM launch (+ one split-K finish) for a list of independent unbatched products
    (a4 [1,1,M,K], b4 [1,1,K,N], out4 [1,1,M,N], a_rowsum [M] or None, bias [N] or None)."""
    n = len(problems)
    nbytes = n * ctypes.sizeof(_lib.GemmProblem)
    capturing = torch.cuda.is_current_stream_capturing()
    D = Deferred
    if not capturing and (D.pool is None or D.pool_used + (64 << 10) > D.pool.numel()):
        if D.pool is not None:
            D.retired.append(D.pool)                   # captured graphs still read their tables from it
        D.pool, D.pool_used = torch.empty(1 << 20, dtype=torch.uint8, pin_memory=True), 0
    if capturing:
        if D.pool is None or D.pool_used + nbytes > D.pool.numel():
            raise RuntimeError('gator_amd.train: run one eager step before capturing (pinned table pool) / pool exhausted')
        arr = (_lib.GemmProblem * n).from_address(D.pool.data_ptr() + D.pool_used)
        D.pool_used += (nbytes + 255) // 256 * 256
    else:
        arr = (_lib.GemmProblem * n)()                 # pageable: the upload is staged before the call returns
    for p, (a4, b4, out4, rowsum, bias) in zip(arr, problems):
        M, K, N = a4.shape[2], a4.shape[3], b4.shape[3]
        tiles = ((M + 63) // 64) * ((N + 63) // 64)
        p.A, p.B, p.C = a4.data_ptr(), b4.data_ptr(), out4.data_ptr()
        p.a_rowsum = rowsum.data_ptr() if rowsum is not None else None
        p.bias = bias.data_ptr() if bias is not None else None
        p.M, p.N, p.K = M, N, K
        p.ksplit = max(1, min(64, K // 128, 1024 // tiles)) if (K >= 512 and tiles < 256) else 1
        p.stride_a[0], p.stride_a[1] = a4.stride(2), a4.stride(3)
        p.stride_b[0], p.stride_b[1] = b4.stride(2), b4.stride(3)
        p.stride_c[0], p.stride_c[1] = out4.stride(2), out4.stride(3)
        p.alpha, p.accumulate = 1.0, 0
    lib = _lib.load()
    ws_floats = int(lib.gator_t_gemm_grouped_prepare(arr, n))
    if ws_floats < 0:
        raise RuntimeError('gator_t_gemm_grouped_prepare rejected the problem list')
    ws = torch.empty(max(ws_floats, 1), device=dev, dtype=torch.float32)
    table = torch.empty(nbytes, device=dev, dtype=torch.uint8)
    _call('gator_t_gemm_grouped', arr, n, table.data_ptr(), ws.data_ptr(), ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream))


# ------------------------------------------------------------------------------------------------ differentiable ops
class _Binary(torch.autograd.Function):
    @staticmethod
    def forward(ctx, op, a, b):
        _need_device(a, b)
        ctx.op, ctx.sa, ctx.sb = op, a.shape, b.shape
        ctx.save_for_backward(a if op in (MUL, DIV) else None, b if op in (MUL, DIV) else None)
        return raw_binary(op, a, b)

    @staticmethod
    def backward(ctx, g):
        op = ctx.op
        a, b = ctx.saved_tensors
        ga = gb = None
        if op == ADD:
            ga, gb = g, g
        elif op == SUB:
            ga = g
            gb = raw_unary(U_AFFINE, g, -1.0, 0.0) if ctx.needs_input_grad[2] else None
        elif op == MUL:
            ga = raw_binary(MUL, g, b) if ctx.needs_input_grad[1] else None
            gb = raw_binary(MUL, g, a) if ctx.needs_input_grad[2] else None
        else:
            ga = raw_binary(DIV, g, b) if ctx.needs_input_grad[1] else None
            if ctx.needs_input_grad[2]:                       # -g a / b^2
                gb = raw_unary(U_AFFINE, raw_binary(DIV, raw_binary(MUL, g, raw_binary(DIV, a, b)), b), -1.0, 0.0)
        ga = sum_to(ga, ctx.sa) if (ga is not None and ctx.needs_input_grad[1]) else None
        gb = sum_to(gb, ctx.sb) if (gb is not None and ctx.needs_input_grad[2]) else None
        return None, ga, gb


def add(a, b):
    return _Binary.apply(ADD, a, b)


def sub(a, b):
    return _Binary.apply(SUB, a, b)


def mul(a, b):
    return _Binary.apply(MUL, a, b)


def div(a, b):
    return _Binary.apply(DIV, a, b)


class _Unary(torch.autograd.Function):
    @staticmethod
    def forward(ctx, op, x, p0, p1):
        _need_device(x)
        y = raw_unary(op, x, p0, p1)
        ctx.op, ctx.p0 = op, p0
        ctx.save_for_backward(x if op in (U_GELU, U_ABS, U_SQUARE) else None, y if op in (U_EXP, U_RSQRT, U_SQRT, U_RECIP, U_POWBASE) else None)
        return y

    @staticmethod
    def backward(ctx, g):
        import math
        op = ctx.op
        x, y = ctx.saved_tensors
        if op == U_AFFINE:
            gx = raw_unary(U_AFFINE, g, ctx.p0, 0.0)
        elif op == U_GELU:
            gx = raw_binary(MUL, g, raw_unary(U_DGELU, x))
        elif op == U_EXP:
            gx = raw_binary(MUL, g, y)
        elif op == U_RSQRT:                                    # -0.5 y^3
            gx = raw_binary(MUL, g, raw_unary(U_AFFINE, raw_binary(MUL, raw_binary(MUL, y, y), y), -0.5, 0.0))
        elif op == U_SQRT:                                     # 0.5 / y
            gx = raw_binary(DIV, raw_unary(U_AFFINE, g, 0.5, 0.0), y)
        elif op == U_RECIP:                                    # -y^2
            gx = raw_binary(MUL, g, raw_unary(U_AFFINE, raw_binary(MUL, y, y), -1.0, 0.0))
        elif op == U_ABS:
            gx = raw_binary(MUL, g, raw_unary(U_SIGN, x))
        elif op == U_POWBASE:                                  # ln(p0) y
            gx = raw_binary(MUL, g, raw_unary(U_AFFINE, y, math.log(ctx.p0), 0.0))
        elif op == U_SQUARE:
            gx = raw_binary(MUL, g, raw_unary(U_AFFINE, x, 2.0, 0.0))
        else:
            raise RuntimeError('no gradient for unary op %d' % op)
        return None, gx, None, None


def affine(x, scale=1.0, shift=0.0):
    return _Unary.apply(U_AFFINE, x, float(scale), float(shift))


def gelu(x):
    return _Unary.apply(U_GELU, x, 0.0, 0.0)


def exp(x):
    return _Unary.apply(U_EXP, x, 0.0, 0.0)


def rsqrt(x):
    return _Unary.apply(U_RSQRT, x, 0.0, 0.0)


def sqrt(x):
    return _Unary.apply(U_SQRT, x, 0.0, 0.0)


def recip(x):
    return _Unary.apply(U_RECIP, x, 0.0, 0.0)


def abs_(x):
    return _Unary.apply(U_ABS, x, 0.0, 0.0)


def square(x):
    return _Unary.apply(U_SQUARE, x, 0.0, 0.0)


def pow_base(base, x):
    return _Unary.apply(U_POWBASE, x, float(base), 0.0)


class _Sum(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, dims, keepdim):
        _need_device(x)
        ctx.shape, ctx.dims, ctx.keepdim = x.shape, [d % x.dim() for d in dims], keepdim
        return raw_sum(x, dims, keepdim)

    @staticmethod
    def backward(ctx, g):
        if not ctx.keepdim:
            kept = [1 if i in ctx.dims else n for i, n in enumerate(ctx.shape)]
            g = g.reshape(kept)
        return raw_copy(g.expand(ctx.shape)), None, None


def sum_(x, dims, keepdim=False):
    return _Sum.apply(x, tuple(dims), keepdim)


def mean(x, dims, keepdim=False):
    n = 1
    for d in dims:
        n *= x.shape[d]
    return affine(sum_(x, dims, keepdim), 1.0 / n)


def _as4(t, batch_shape):
    """[..., R, C] with <= 2 leading batch dims -> expanded 4-D view [n1, n2, R, C]."""
    lead = list(t.shape[:-2])
    if len(lead) > 2:
        raise ValueError('matmul: at most two batch dims')
    t4 = t.reshape([1] * (2 - len(lead)) + list(t.shape)) if len(lead) < 2 else t
    return t4.expand(list(batch_shape) + list(t.shape[-2:]))


class _MatMul(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b, alpha):
        _need_device(a, b)
        la, lb = list(a.shape[:-2]), list(b.shape[:-2])
        bs = list(torch.broadcast_shapes(tuple(la), tuple(lb)))
        bs4 = [1] * (2 - len(bs)) + bs
        a4, b4 = _as4(a, bs4), _as4(b, bs4)
        ctx.alpha, ctx.bs, ctx.sa, ctx.sb = alpha, bs, a.shape, b.shape
        ctx.save_for_backward(a, b)
        out = raw_gemm(a4, b4, alpha=alpha)
        return out.reshape(bs + [a.shape[-2], b.shape[-1]])

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        bs4 = [1] * (2 - len(ctx.bs)) + list(ctx.bs)
        g4 = _as4(g, bs4)
        ga = gb = None
        if ctx.needs_input_grad[0]:
            ga = raw_gemm(g4, _as4(b, bs4).transpose(2, 3), alpha=ctx.alpha).reshape(list(ctx.bs) + list(ctx.sa[-2:]))
            ga = sum_to(ga, ctx.sa)
        if ctx.needs_input_grad[1]:
            gb = raw_gemm(_as4(a, bs4).transpose(2, 3), g4, alpha=ctx.alpha).reshape(list(ctx.bs) + list(ctx.sb[-2:]))
            gb = sum_to(gb, ctx.sb)
        return ga, gb, None


def matmul(a, b, alpha=1.0):
    return _MatMul.apply(a, b, float(alpha))


class _Linear(torch.autograd.Function):
    """y = x W^T + b over the last dim (F.linear): one GEMM on the flattened rows; dW is one split-K GEMM over all rows."""

    @staticmethod
    def forward(ctx, x, w, b):
        _need_device(x, w, b)
        x2 = _contig(x).reshape(1, 1, -1, x.shape[-1])
        w4 = w.reshape(1, 1, w.shape[0], w.shape[1])
        ctx.save_for_backward(x2, w4)
        ctx.xshape, ctx.has_b = x.shape, b is not None
        ctx.wslot, ctx.bslot = grad_slot(w), grad_slot(b) if b is not None else None
        y = raw_gemm(x2, w4.transpose(2, 3), bias=b)
        return y.reshape(list(x.shape[:-1]) + [w.shape[0]])

    @staticmethod
    def backward(ctx, g):
        x2, w4 = ctx.saved_tensors
        g2 = _contig(g).reshape(1, 1, -1, g.shape[-1])
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = raw_gemm(g2, w4).reshape(ctx.xshape)
        need_b = ctx.has_b and ctx.needs_input_grad[2]
        if ctx.needs_input_grad[1]:                      # dW = dY^T X; the bias gradient (row sums of dY^T) rides on the same launch
            gb = (ctx.bslot if ctx.bslot is not None else torch.empty(w4.shape[2], device=g.device, dtype=torch.float32)) if need_b else None
            if Deferred.enabled and ctx.wslot is not None and (gb is None or ctx.bslot is not None):
                Deferred.add(g2.transpose(2, 3), x2, ctx.wslot.view(w4.shape), gb)       # lands in the flat gradient buffer: grouped launch
                gw = ctx.wslot
            else:
                gw = raw_gemm(g2.transpose(2, 3), x2, a_rowsum=gb, out=ctx.wslot.view(w4.shape) if ctx.wslot is not None else None)
                gw = ctx.wslot if ctx.wslot is not None else gw.reshape(w4.shape[2], w4.shape[3])
        elif need_b:
            gb = raw_sum(g2.reshape(-1, g2.shape[-1]), [0])
        return gx, gw, gb


def linear(x, w, b=None):
    return _Linear.apply(x, w, b)


class _LinearGroup(torch.autograd.Function):
    """Several F.linear layers with independent inputs / weights in ONE grouped launch per direction (forward products; backward
    activation gradients) - q / k / v projections, the two hop linears of X_Feat ...  Weight gradients go to the deferred group."""

    @staticmethod
    def forward(ctx, n, *args):                           # args = x_0, w_0, b_0, x_1, w_1, b_1, ...
        xs, ws, bs = args[0::3], args[1::3], args[2::3]
        _need_device(*xs)
        x2s = [_contig(x).reshape(1, 1, -1, x.shape[-1]) for x in xs]
        w4s = [w.reshape(1, 1, w.shape[0], w.shape[1]) for w in ws]
        outs = [torch.empty((1, 1, x2.shape[2], w4.shape[2]), device=x2.device, dtype=torch.float32) for x2, w4 in zip(x2s, w4s)]
        run_group([(x2, w4.transpose(2, 3), o, None, b) for x2, w4, o, b in zip(x2s, w4s, outs, bs)], x2s[0].device)
        ctx.save_for_backward(*(x2s + w4s))
        ctx.n, ctx.xshapes, ctx.has_b = n, [x.shape for x in xs], [b is not None for b in bs]
        ctx.wslots = [grad_slot(w) for w in ws]
        ctx.bslots = [grad_slot(b) if b is not None else None for b in bs]
        return tuple(o.reshape(list(sh[:-1]) + [w4.shape[2]]) for o, sh, w4 in zip(outs, ctx.xshapes, w4s))

    @staticmethod
    def backward(ctx, *gs):
        n = ctx.n
        x2s, w4s = ctx.saved_tensors[:n], ctx.saved_tensors[n:]
        g2s = [_contig(g).reshape(1, 1, -1, g.shape[-1]) for g in gs]
        grads = [None] * (3 * n)
        gxs = [torch.empty(x2.shape, device=x2.device, dtype=torch.float32) if ctx.needs_input_grad[1 + 3 * i] else None for i, x2 in enumerate(x2s)]
        probs = [(g2s[i], w4s[i], gxs[i], None, None) for i in range(n) if gxs[i] is not None]
        if probs:
            run_group(probs, g2s[0].device)
        for i in range(n):
            if gxs[i] is not None:
                grads[3 * i] = gxs[i].reshape(ctx.xshapes[i])
            w4 = w4s[i]
            need_b = ctx.has_b[i] and ctx.needs_input_grad[3 + 3 * i]
            if ctx.needs_input_grad[2 + 3 * i]:
                gb = (ctx.bslots[i] if ctx.bslots[i] is not None else torch.empty(w4.shape[2], device=w4.device, dtype=torch.float32)) if need_b else None
                if Deferred.enabled and ctx.wslots[i] is not None and (gb is None or ctx.bslots[i] is not None):
                    Deferred.add(g2s[i].transpose(2, 3), x2s[i], ctx.wslots[i].view(w4.shape), gb)
                    grads[3 * i + 1] = ctx.wslots[i]
                else:
                    grads[3 * i + 1] = raw_gemm(g2s[i].transpose(2, 3), x2s[i], a_rowsum=gb).reshape(w4.shape[2], w4.shape[3])
                grads[3 * i + 2] = gb
            elif need_b:
                grads[3 * i + 2] = raw_sum(g2s[i].reshape(-1, g2s[i].shape[-1]), [0])
        return (None,) + tuple(grads)


def linear_group(items):
    """[(x, w, b or None), ...] -> [x_i W_i^T + b_i]: independent linears as one launch"""
    flat = []
    for x, w, b in items:
        flat += [x, w, b]
    return list(_LinearGroup.apply(len(items), *flat))


class _XW(torch.autograd.Function):
    """x @ W[k] for a stacked weight W [n, C_in, C_out] (MGCN's `torch.matmul(input, self.W[k])`, modules.py:244-245).  The weight
    gradient x^T dY lands in W's slice of the flat gradient buffer through the grouped launch (no split / copy chain).  A model
    that uses W must use its slice 0 (the slot is reported to autograd by that use)."""

    @staticmethod
    def forward(ctx, x, W, k):
        _need_device(x, W)
        x2 = _contig(x).reshape(1, 1, -1, x.shape[-1])
        wk = W.narrow(0, k, 1).reshape(1, 1, W.shape[1], W.shape[2])
        ctx.save_for_backward(x2, wk)
        ctx.xshape, ctx.k, ctx.wshape = x.shape, k, W.shape
        sl = grad_slot(W, part=k)
        ctx.wslot = sl.narrow(0, k, 1).view(1, 1, W.shape[1], W.shape[2]) if sl is not None else None
        ctx.full = sl
        return raw_gemm(x2, wk).reshape(list(x.shape[:-1]) + [W.shape[2]])

    @staticmethod
    def backward(ctx, g):
        x2, wk = ctx.saved_tensors
        g2 = _contig(g).reshape(1, 1, -1, g.shape[-1])
        gx = raw_gemm(g2, wk.transpose(2, 3)).reshape(ctx.xshape) if ctx.needs_input_grad[0] else None
        gW = None
        if ctx.needs_input_grad[1]:
            if Deferred.enabled and ctx.wslot is not None:
                Deferred.add(x2.transpose(2, 3), g2, ctx.wslot, None)
                # every slice writes into the SAME slot tensor: hand it to autograd once (slice 0), or the engine would add the
                # tensor to itself for each further use of W
                gW = ctx.full if ctx.k == 0 else None
            else:
                gW = zeros(ctx.wshape, g.device)
                raw_gemm(x2.transpose(2, 3), g2, out=gW.narrow(0, ctx.k, 1).view(1, 1, ctx.wshape[1], ctx.wshape[2]))
        return gx, gW, None


def xw(x, W, k):
    return _XW.apply(x, W, int(k))


class _Softmax(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        _need_device(x)
        xc = _contig(x)
        p = torch.empty_like(xc)
        n = xc.shape[-1]
        _call('gator_t_softmax_fwd', xc.data_ptr(), xc.numel() // n, n, p.data_ptr(), _stream(x))
        ctx.save_for_backward(p)
        return p

    @staticmethod
    def backward(ctx, g):
        p, = ctx.saved_tensors
        gc = _contig(g)
        dx = torch.empty_like(p)
        n = p.shape[-1]
        _call('gator_t_softmax_bwd', p.data_ptr(), gc.data_ptr(), p.numel() // n, n, dx.data_ptr(), _stream(p))
        return dx


def softmax(x):
    """softmax over the last dim"""
    return _Softmax.apply(x)


class _LayerNorm(torch.autograd.Function):
    """y = LN(x) and, with residual=True, a second output that IS x (the skip connection): the backward then forms
    d x = LN-backward(d y) + d skip in the LayerNorm kernel itself instead of a fork + add launch."""

    @staticmethod
    def forward(ctx, x, w, b, eps, mode, residual=False):
        _need_device(x, w, b)
        xc = _contig(x)
        n = xc.shape[-1]
        rows = xc.numel() // n
        y = torch.empty_like(xc)
        mean = torch.empty(rows, device=x.device, dtype=torch.float32)
        rinv = torch.empty(rows, device=x.device, dtype=torch.float32)
        _call('gator_t_layernorm_fwd', xc.data_ptr(), rows, n, w.data_ptr() if w is not None else None, b.data_ptr() if b is not None else None,
              float(eps), int(mode), y.data_ptr(), mean.data_ptr(), rinv.data_ptr(), _stream(x))
        ctx.save_for_backward(xc, mean, rinv, w)
        ctx.eps, ctx.mode, ctx.has_b = eps, mode, b is not None
        ctx.wslot, ctx.bslot = grad_slot(w) if w is not None else None, grad_slot(b) if b is not None else None
        ctx.residual = residual
        if residual:
            ctx.set_materialize_grads(False)
            return y, xc.view_as(xc)
        return y

    @staticmethod
    def backward(ctx, g, gres=None):
        xc, mean, rinv, w = ctx.saved_tensors
        if g is None:                                      # only the skip connection carried a gradient
            return (gres, None, None, None, None, None)
        gc = _contig(g)
        add = _contig(gres) if gres is not None else None
        n = xc.shape[-1]
        rows = xc.numel() // n
        dx = torch.empty_like(xc)
        need_w = w is not None and ctx.needs_input_grad[1]
        dyx = torch.empty_like(xc) if need_w else None
        _call('gator_t_layernorm_bwd', gc.data_ptr(), xc.data_ptr(), mean.data_ptr(), rinv.data_ptr(), w.data_ptr() if w is not None else None, rows, n,
              float(ctx.eps), int(ctx.mode), dx.data_ptr(), dyx.data_ptr() if need_w else None, add.data_ptr() if add is not None else None, _stream(xc))
        gw = gb = None
        grouped = Deferred.enabled
        ones = _one(xc.device).as_strided((1, 1, 1, rows), (0, 0, 0, 0)) if grouped else None      # column sums as 1 x rows products
        if need_w:
            if grouped and ctx.wslot is not None:
                Deferred.add(ones, dyx.view(1, 1, rows, n), ctx.wslot.view(1, 1, 1, n), None)
                gw = ctx.wslot
            else:
                gw = raw_sum(dyx.reshape(rows, n), [0], keepdim=True, out=ctx.wslot.view(1, n) if ctx.wslot is not None else None).reshape(n)
        if ctx.has_b and ctx.needs_input_grad[2]:
            if grouped and ctx.bslot is not None:
                Deferred.add(ones, gc.view(1, 1, rows, n), ctx.bslot.view(1, 1, 1, n), None)
                gb = ctx.bslot
            else:
                gb = raw_sum(gc.reshape(rows, n), [0], keepdim=True, out=ctx.bslot.view(1, n) if ctx.bslot is not None else None).reshape(n)
        return dx, gw, gb, None, None, None


def layernorm(x, w=None, b=None, eps=1e-5, mode=0):
    """mode 0: nn.LayerNorm over the last dim; mode 1: lib/models/vanilla_transformer_encoder.py:31-34 (unbiased std, eps on the std)."""
    return _LayerNorm.apply(x, w, b, float(eps), int(mode))


def layernorm_skip(x, w=None, b=None, eps=1e-5, mode=0):
    """(LN(x), x): the pre-norm residual pattern `res = x; y = norm(x)` with the two gradients of x summed inside the LayerNorm backward"""
    return _LayerNorm.apply(x, w, b, float(eps), int(mode), True)


class Generator:
    """Philox stream of the dropout masks: (seed, running offset).  Every mask draws a fresh offset, so a step is reproducible
    from (seed, step) and no two sites share random numbers.  With a device step counter (`device_steps(dev)`) the offset's high
    word comes from device memory: `begin_step()` resets the site index and advances the counter with a kernel, so a training
    step captured in a hipGraph draws new masks at every replay."""

    def __init__(self, seed=0):
        self.seed, self.offset, self.counter = int(seed), 0, None

    def device_steps(self, device):
        self.counter = torch.zeros(1, device=device, dtype=torch.int64)
        return self

    def begin_step(self):
        if self.counter is not None:
            self.offset = 0
            _call('gator_t_step_advance', self.counter.data_ptr(), ctypes.c_void_p(torch.cuda.current_stream(self.counter.device).cuda_stream))

    def next_offset(self):
        self.offset += 1
        return self.offset

    def counter_ptr(self):
        return self.counter.data_ptr() if self.counter is not None else None


class _Dropout(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, rate, seed, offset, counter):
        _need_device(x)
        xc = _contig(x)
        out = torch.empty_like(xc)
        mask = torch.empty(xc.shape, device=x.device, dtype=torch.uint8)
        _call('gator_t_dropout', xc.data_ptr(), xc.numel(), float(rate), ctypes.c_uint64(seed), ctypes.c_uint64(offset), counter, out.data_ptr(),
              mask.data_ptr(), _stream(x))
        ctx.save_for_backward(mask)
        ctx.scale = 1.0 / (1.0 - rate)
        return out

    @staticmethod
    def backward(ctx, g):
        mask, = ctx.saved_tensors
        gc = _contig(g)
        out = torch.empty_like(gc)
        _call('gator_t_mask_scale', gc.data_ptr(), mask.data_ptr(), gc.numel(), float(ctx.scale), out.data_ptr(), _stream(gc))
        return out, None, None, None, None


def dropout(x, rate, gen, training=True):
    """nn.Dropout: identity when not training or rate == 0."""
    if not training or rate <= 0.0:
        return x
    return _Dropout.apply(x, float(rate), gen.seed, gen.next_offset(), gen.counter_ptr())


def drop_path(x, rate, gen, training=True):
    """timm DropPath (lib/models/GAT.py:25, MDR.py:57): one keep decision per SAMPLE, kept samples scaled by 1/(1-rate)."""
    if not training or rate <= 0.0:
        return x
    B = x.shape[0]
    factor = torch.empty(B, device=x.device, dtype=torch.float32)
    mask = torch.empty(B, device=x.device, dtype=torch.uint8)
    _call('gator_t_dropout', None, B, float(rate), ctypes.c_uint64(gen.seed), ctypes.c_uint64(gen.next_offset()), gen.counter_ptr(), factor.data_ptr(),
          mask.data_ptr(), _stream(x))
    return mul(x, factor.reshape([B] + [1] * (x.dim() - 1)))


class _Attention(torch.autograd.Function):
    """o = dropout(softmax(scale q k^T)) v per head, q / k / v / o [B, T, H*32] (gator_t_attn_fwd / _bwd): no [B,H,T,T] tensor is
    ever written; the masks are those the composed form (softmax -> dropout -> matmul) draws from the same generator state."""

    @staticmethod
    def forward(ctx, q, k, v, heads, scale, rate, seed, offset, counter):
        _need_device(q, k, v)
        q, k, v = _contig(q), _contig(k), _contig(v)
        B, T, HD = q.shape
        o = torch.empty_like(q)
        lse = torch.empty((B, heads, T), device=q.device, dtype=torch.float32)
        _call('gator_t_attn_fwd', q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), lse.data_ptr(), B, heads, T, k.shape[1], HD // heads, float(scale),
              float(rate), ctypes.c_uint64(seed), ctypes.c_uint64(offset), counter, _stream(q))
        ctx.save_for_backward(q, k, v, o, lse)
        ctx.cfg = (heads, scale, rate, seed, offset, counter)
        return o

    @staticmethod
    def backward(ctx, g):
        q, k, v, o, lse = ctx.saved_tensors
        heads, scale, rate, seed, offset, counter = ctx.cfg
        B, T, HD = q.shape
        gc = _contig(g)
        Tk = k.shape[1]
        # few keys, many queries (the cross-attention): the key-owning waves would walk all query tiles alone - deal the tiles out
        qsplit = min((T + 31) // 32, 16) if Tk <= 128 and T >= 4 * Tk else 1
        dq = torch.empty_like(q)
        dk = torch.empty((qsplit,) + tuple(k.shape), device=q.device, dtype=torch.float32)
        dv = torch.empty((qsplit,) + tuple(v.shape), device=q.device, dtype=torch.float32)
        dsum = torch.empty_like(lse)
        _call('gator_t_attn_bwd', q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), lse.data_ptr(), gc.data_ptr(), dq.data_ptr(), dk.data_ptr(),
              dv.data_ptr(), dsum.data_ptr(), B, heads, T, Tk, HD // heads, float(scale), float(rate), ctypes.c_uint64(seed), ctypes.c_uint64(offset), counter,
              qsplit, _stream(q))
        if qsplit > 1:
            dk, dv = raw_sum(dk.view(qsplit, -1), [0]).view(k.shape), raw_sum(dv.view(qsplit, -1), [0]).view(v.shape)
        else:
            dk, dv = dk[0], dv[0]
        return dq, dk, dv, None, None, None, None, None, None


def attention(q, k, v, heads, scale, rate=0.0, gen=None, training=True):
    """q [B, T, heads*32], k, v [B, Tk, heads*32] -> [B, T, heads*32]"""
    on = training and rate > 0.0
    return _Attention.apply(q, k, v, int(heads), float(scale), float(rate) if on else 0.0, gen.seed if on else 0, gen.next_offset() if on else 0,
                            gen.counter_ptr() if on else None)


class _AttentionSmall(torch.autograd.Function):
    """The GAT encoder's attention (gator_t_attn_small_fwd / _bwd): qkv [B,J,3*H*16] and the additive bias [H,J,J] -> [B,J,H*16]."""

    @staticmethod
    def forward(ctx, qkv, bias, heads, scale, rate, seed, offset, counter):
        _need_device(qkv, bias)
        qkv, bias = _contig(qkv), _contig(bias)
        B, J, C3 = qkv.shape
        C = C3 // 3
        o = torch.empty((B, J, C), device=qkv.device, dtype=torch.float32)
        P = torch.empty((B, heads, J, J), device=qkv.device, dtype=torch.float32)
        _call('gator_t_attn_small_fwd', qkv.data_ptr(), bias.data_ptr(), o.data_ptr(), P.data_ptr(), B, heads, J, C // heads, float(scale), float(rate),
              ctypes.c_uint64(seed), ctypes.c_uint64(offset), counter, _stream(qkv))
        ctx.save_for_backward(qkv, bias, P)
        ctx.cfg = (heads, scale, rate, seed, offset, counter)
        return o

    @staticmethod
    def backward(ctx, g):
        qkv, bias, P = ctx.saved_tensors
        heads, scale, rate, seed, offset, counter = ctx.cfg
        B, J, C3 = qkv.shape
        gc = _contig(g)
        dqkv = torch.empty_like(qkv)
        dS = torch.empty_like(P)
        _call('gator_t_attn_small_bwd', qkv.data_ptr(), bias.data_ptr(), P.data_ptr(), gc.data_ptr(), dqkv.data_ptr(), dS.data_ptr(), B, heads, J,
              C3 // 3 // heads, float(scale), float(rate), ctypes.c_uint64(seed), ctypes.c_uint64(offset), counter, _stream(qkv))
        dbias = raw_sum(dS, [0]) if ctx.needs_input_grad[1] else None
        return dqkv, dbias, None, None, None, None, None, None


def attention_small(qkv, bias, heads, scale, rate=0.0, gen=None, training=True):
    """qkv [B,J,3*heads*16] (the GAT in-projection's output), bias [heads,J,J] -> attention output [B,J,heads*16]"""
    on = training and rate > 0.0
    return _AttentionSmall.apply(qkv, bias, int(heads), float(scale), float(rate) if on else 0.0, gen.seed if on else 0, gen.next_offset() if on else 0,
                                 gen.counter_ptr() if on else None)


class _Mgcn(torch.autograd.Function):
    """gator_t_mgcn_fwd / _bwd: (h0, h1 [B,J,C], adj [J,J], M [J,C], bias [C]) -> [B,J,C]"""

    @staticmethod
    def forward(ctx, h0, h1, adj, M, bias):
        _need_device(h0, h1, adj, M, bias)
        h0, h1, adj, Mc, bc = _contig(h0), _contig(h1), _contig(adj), _contig(M), _contig(bias)
        B, J, C = h0.shape
        out = torch.empty_like(h0)
        _call('gator_t_mgcn_fwd', h0.data_ptr(), h1.data_ptr(), adj.data_ptr(), Mc.data_ptr(), bc.data_ptr(), out.data_ptr(), B, J, C, _stream(h0))
        ctx.save_for_backward(h0, h1, adj, Mc)
        ctx.mslot, ctx.bslot = grad_slot(M), grad_slot(bias)
        return out

    @staticmethod
    def backward(ctx, g):
        h0, h1, adj, M = ctx.saved_tensors
        B, J, C = h0.shape
        gc = _contig(g)
        dh0, dh1, pm = torch.empty_like(h0), torch.empty_like(h0), torch.empty_like(h0)
        dadj = torch.empty((B, J, J), device=g.device, dtype=torch.float32)
        _call('gator_t_mgcn_bwd', h0.data_ptr(), h1.data_ptr(), adj.data_ptr(), M.data_ptr(), gc.data_ptr(), dh0.data_ptr(), dh1.data_ptr(), pm.data_ptr(),
              dadj.data_ptr(), B, J, C, _stream(g))
        ones = lambda n: _one(g.device).as_strided((1, 1, 1, n), (0, 0, 0, 0))
        if Deferred.enabled and ctx.mslot is not None:
            Deferred.add(ones(B), pm.view(1, 1, B, J * C), ctx.mslot.view(1, 1, 1, J * C), None)
            gM = ctx.mslot
        else:
            gM = raw_sum(pm, [0])
        if Deferred.enabled and ctx.bslot is not None:
            Deferred.add(ones(B * J), gc.view(1, 1, B * J, C), ctx.bslot.view(1, 1, 1, C), None)
            gb = ctx.bslot
        else:
            gb = raw_sum(gc.reshape(B * J, C), [0])
        return dh0, dh1, raw_sum(dadj, [0]), gM, gb


def mgcn(h0, h1, adj, M, bias):
    return _Mgcn.apply(h0, h1, adj, M, bias)


class _DropFused(torch.autograd.Function):
    """out = res + DropPath(dropout(act(x))) in one launch per direction (gator_t_drop_fused): the tail of every residual branch."""

    @staticmethod
    def forward(ctx, x, res, gelu, rate, seed, offset, path_rate, path_offset, counter):
        _need_device(x, res)
        xc = _contig(x)
        rc = _contig(res) if res is not None else None
        n, per = xc.numel(), xc.numel() // xc.shape[0]
        out = torch.empty_like(xc)
        mask = torch.empty(xc.shape, device=x.device, dtype=torch.uint8) if (rate > 0 and offset) else None
        fac = torch.empty(xc.shape[0], device=x.device, dtype=torch.float32) if (path_rate > 0 and path_offset) else None
        _call('gator_t_drop_fused', xc.data_ptr(), rc.data_ptr() if rc is not None else None, n, per, int(gelu), float(rate), ctypes.c_uint64(seed),
              ctypes.c_uint64(offset), float(path_rate), ctypes.c_uint64(path_offset), counter, out.data_ptr(), mask.data_ptr() if mask is not None else None,
              fac.data_ptr() if fac is not None else None, _stream(x))
        ctx.save_for_backward(xc if gelu else None, mask, fac)
        ctx.cfg = (gelu, rate, res is not None)
        return out

    @staticmethod
    def backward(ctx, g):
        x, mask, fac = ctx.saved_tensors
        gelu, rate, has_res = ctx.cfg
        gc = _contig(g)
        if not gelu and mask is None and fac is None:
            dx = gc
        else:
            dx = torch.empty_like(gc)
            _call('gator_t_drop_fused_bwd', gc.data_ptr(), x.data_ptr() if x is not None else None, mask.data_ptr() if mask is not None else None,
                  fac.data_ptr() if fac is not None else None, gc.numel(), gc.numel() // gc.shape[0], int(gelu), float(rate), dx.data_ptr(), _stream(gc))
        return dx, (gc if has_res else None), None, None, None, None, None, None, None


def drop_fused(x, res=None, gelu=False, rate=0.0, path_rate=0.0, gen=None, training=True):
    """res + drop_path(dropout(gelu?(x), rate), path_rate) - offsets drawn in that order, as the composed chain draws them."""
    d_on = training and rate > 0.0
    p_on = training and path_rate > 0.0
    off = gen.next_offset() if d_on else 0
    poff = gen.next_offset() if p_on else 0
    if not gelu and not d_on and not p_on:
        return x if res is None else add(res, x)
    return _DropFused.apply(x, res, bool(gelu), float(rate) if d_on else 0.0, gen.seed if (d_on or p_on) else 0, off, float(path_rate) if p_on else 0.0, poff,
                            gen.counter_ptr() if (d_on or p_on) else None)


class _BatchNormTrain(torch.autograd.Function):
    """nn.BatchNorm1d(C) in training mode on x [B,C,L] (gator_t_batchnorm_fwd / _bwd); the running statistics are updated in place."""

    @staticmethod
    def forward(ctx, x, w, b, run_mean, run_var, eps, momentum):
        _need_device(x, w, b)
        xc = _contig(x)
        B, C, L = xc.shape
        y = torch.empty_like(xc)
        mean = torch.empty(C, device=x.device, dtype=torch.float32)
        rinv = torch.empty(C, device=x.device, dtype=torch.float32)
        _call('gator_t_batchnorm_fwd', xc.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), mean.data_ptr(), rinv.data_ptr(),
              run_mean.data_ptr() if run_mean is not None else None, run_var.data_ptr() if run_var is not None else None, B, C, L, float(eps),
              float(momentum), _stream(x))
        ctx.save_for_backward(xc, w, mean, rinv)
        ctx.wslot, ctx.bslot = grad_slot(w), grad_slot(b)
        return y

    @staticmethod
    def backward(ctx, g):
        xc, w, mean, rinv = ctx.saved_tensors
        B, C, L = xc.shape
        gc = _contig(g)
        dx = torch.empty_like(xc)
        dw = ctx.wslot if ctx.wslot is not None else torch.empty(C, device=g.device, dtype=torch.float32)
        db = ctx.bslot if ctx.bslot is not None else torch.empty(C, device=g.device, dtype=torch.float32)
        _call('gator_t_batchnorm_bwd', gc.data_ptr(), xc.data_ptr(), w.data_ptr(), mean.data_ptr(), rinv.data_ptr(), dx.data_ptr(), dw.data_ptr(),
              db.data_ptr(), B, C, L, _stream(g))
        return dx, dw, db, None, None, None, None


def batchnorm_train(x, w, b, run_mean=None, run_var=None, eps=1e-5, momentum=0.1):
    return _BatchNormTrain.apply(x, w, b, run_mean, run_var, float(eps), float(momentum))


class _Contig(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return raw_copy(x)

    @staticmethod
    def backward(ctx, g):
        return g


def contiguous(x):
    return x if x.is_contiguous() else _Contig.apply(x)


class _Cat(torch.autograd.Function):
    @staticmethod
    def forward(ctx, dim, *ts):
        _need_device(*ts)
        dim = dim % ts[0].dim()
        ctx.dim, ctx.sizes = dim, [t.shape[dim] for t in ts]
        shape = list(ts[0].shape)
        shape[dim] = sum(ctx.sizes)
        out = torch.empty(shape, device=ts[0].device, dtype=torch.float32)
        o = 0
        for t in ts:
            raw_unary(U_AFFINE, t, 1.0, 0.0, out=out.narrow(dim, o, t.shape[dim]))
            o += t.shape[dim]
        return out

    @staticmethod
    def backward(ctx, g):
        outs, o = [], 0
        for n in ctx.sizes:
            outs.append(g.narrow(ctx.dim, o, n))
            o += n
        return (None,) + tuple(outs)


def cat(ts, dim):
    return _Cat.apply(dim, *ts)


class _Narrow(torch.autograd.Function):
    """x.narrow(dim, start, length) as a view; the backward writes the gradient into a zero tensor with library kernels."""

    @staticmethod
    def forward(ctx, x, dim, start, length):
        ctx.shape, ctx.dim, ctx.start, ctx.length = x.shape, dim, start, length
        return x.narrow(dim, start, length)

    @staticmethod
    def backward(ctx, g):
        out = zeros(ctx.shape, g.device)
        raw_unary(U_AFFINE, g, 1.0, 0.0, out=out.narrow(ctx.dim, ctx.start, ctx.length))
        return out, None, None, None


def narrow(x, dim, start, length):
    return _Narrow.apply(x, dim % x.dim(), start, length)


class _Reshape(torch.autograd.Function):
    """x.reshape(shape) whose backward re-lays a non-contiguous gradient out with the library's copy kernel (autograd's own
    reshape backward would call aten::clone for e.g. the head-split views of q, k, v)."""

    @staticmethod
    def forward(ctx, x, shape):
        ctx.shape = x.shape
        return _contig(x).reshape(shape) if not x.is_contiguous() and not _viewable(x, shape) else x.reshape(shape)

    @staticmethod
    def backward(ctx, g):
        return _contig(g).reshape(ctx.shape), None


def _viewable(x, shape):
    try:
        x.view(shape)
        return True
    except RuntimeError:
        return False


def reshape(x, *shape):
    return _Reshape.apply(x, tuple(shape))


class _Split(torch.autograd.Function):
    """x split into consecutive pieces along `dim` (views); the backward lays the pieces' gradients side by side in ONE buffer
    (a copy per piece - no zero-filled full-size tensors and no additions, unlike narrow + fork)."""

    @staticmethod
    def forward(ctx, x, dim, sizes):
        ctx.set_materialize_grads(False)
        ctx.shape, ctx.dim, ctx.sizes = x.shape, dim, sizes
        outs, o = [], 0
        for n in sizes:
            outs.append(x.narrow(dim, o, n))
            o += n
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gs):
        dev = next(g.device for g in gs if g is not None)
        out = torch.empty(ctx.shape, device=dev, dtype=torch.float32)
        o = 0
        for g, n in zip(gs, ctx.sizes):
            dst = out.narrow(ctx.dim, o, n)
            if g is None:
                raw_unary(U_GT, dst, float('inf'), 0.0, out=dst)
            else:
                raw_unary(U_AFFINE, g, 1.0, 0.0, out=dst)
            o += n
        return out, None, None


def split(x, dim, sizes):
    return _Split.apply(x, dim % x.dim(), tuple(sizes))


class _Fork(torch.autograd.Function):
    """n aliases of x whose gradients are summed by the library (instead of autograd's own accumulation with aten adds)."""

    @staticmethod
    def forward(ctx, x, n):
        ctx.set_materialize_grads(False)                 # an alias used only detached has no gradient: do not zero-fill one
        return tuple(x.view_as(x) for _ in range(n))

    @staticmethod
    def backward(ctx, *gs):
        gs = [g for g in gs if g is not None]
        acc = gs[0]
        rest = gs[1:]
        while rest:                                      # up to four gradients per launch, summed left to right
            take, rest = rest[:3], rest[3:]
            if all(t.is_contiguous() and t.shape == acc.shape for t in take) and acc.is_contiguous():
                out = torch.empty_like(acc)
                ptr = [t.data_ptr() for t in take] + [None] * (3 - len(take))
                _call('gator_t_add_n', acc.data_ptr(), ptr[0], ptr[1], ptr[2], out.data_ptr(), acc.numel(), _stream(acc))
                acc = out
            else:
                for t in take:
                    acc = raw_binary(ADD, acc, t)
        return acc, None


def fork(x, n=2):
    return _Fork.apply(x, n)
