"""torch.autograd.Function wrappers around the C ABI (include/rlt_hip.h).

PyTorch here is plumbing: it owns device buffers, the stream and the autograd tape.  Every
arithmetic step of the hot path - forward AND backward - is a call into librlt_hip.so.
Activations are "position-major": a (S*B, E) matrix whose row index is s*B + b.
"""
import contextvars
import math

import torch
from torch.autograd import Function

from . import native as N
from .native import call, ptr, query, stream, workspace


def _empty(shape, like):
    return torch.empty(shape, dtype=torch.float32, device=like.device)


class KernelTimer:
    """Optional HIP-event timing of named launches on the current stream (bench.py's live roofline
    measurement).  Disabled unless `KernelTimer.active` is set to an instance."""
    active = None

    def __init__(self):
        self.events = {}

    def reset(self):
        self.events = {}

    def timed(self, name, fn):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        self.events.setdefault(name, []).append((a, b))

    def summary(self):
        """name -> (launches, mean milliseconds); call after torch.cuda.synchronize()."""
        return {k: (len(v), sum(a.elapsed_time(b) for a, b in v) / len(v)) for k, v in self.events.items()}


_SEED_COUNTER = [0]
_SEED_STREAM = [0]
KERNEL_LEVEL_ENCODER = [False]      # tests: run the encoder layer launch by launch from the host (EncoderLayerKernelsFn)


def set_seed_stream(stream_id):
    """Data-parallel runs: every rank has the same torch seed (identical initial weights) but must draw its own
    dropout masks - the rank is mixed into the seed as a stream id."""
    _SEED_STREAM[0] = int(stream_id)


def next_seed():
    """A fresh 32-bit dropout seed: a function of torch's seed, the stream id (rank) and a call counter, so that
    torch.manual_seed(...) makes a training run reproducible."""
    _SEED_COUNTER[0] += 1
    x = (torch.initial_seed() * 0x9E3779B97F4A7C15 + _SEED_COUNTER[0] * 0xD1B54A32D192ED03
         + _SEED_STREAM[0] * 0xA24BAED4963EE407) & 0xFFFFFFFFFFFFFFFF
    x ^= x >> 29
    return int((x * 0xBF58476D1CE4E5B9 >> 32) & 0xFFFFFFFF)


# the scope is a context variable: per thread (autograd runs backward nodes on its own threads) and per asyncio task, like the
# thread-local scope of the library's own entry points (include/rlt_hip.h)
_CALL_PRECISION = contextvars.ContextVar("rlt_call_precision", default=N.PRECISION_DEFAULT)


class precision:
    """`with ops.precision('fp32'):` - every library call issued inside (the forward of the modules called there) runs in
    that mode, whatever the process default (native.set_precision) is; each tape node remembers the mode of its forward and
    gives it to its backward (the stash layout depends on it).  Re-entrant: nested scopes restore the outer one; scopes of
    different threads do not see each other."""

    def __init__(self, mode):
        self.code = N.PRECISION_DEFAULT if mode is None else N.precision_code(mode)
        self._tokens = []

    def __enter__(self):
        self._tokens.append(_CALL_PRECISION.set(self.code))
        return self

    def __exit__(self, *exc):
        _CALL_PRECISION.reset(self._tokens.pop())
        return False


def current_precision():
    """The RLT_PRECISION_* code a forward issued now runs in: the enclosing `precision(...)` scope's, else the process
    default read once, so that forward and backward of a tape node agree even if the default is changed in between."""
    c = _CALL_PRECISION.get()
    return c if c >= 0 else int(N.load().rlt_get_precision())


def _launch(name, fn):
    t = KernelTimer.active
    if t is None:
        fn()
    else:
        t.timed(name, fn)


# ------------------------------------------------------------------------------ raw helpers
def gemm(ta, tb, M, Nn, K, A, lda, B, ldb, C, ldc, bias=None, bias2=None, flags=0, a_off=0, b_off=0, c_off=0,
         relu_mask=None, ldmask=0, colsum_a=None, mask_scale=1.0, drop_p=0.0, seed=0, prec=None):
    """C[M,N] (+)= op(A) op(B) (+bias); *_off are element offsets into the tensors.
    relu_mask: C = mask > 0 ? C*mask_scale : 0 (fused ReLU/dropout backward); colsum_a (M): row sums of
    op(A) (ta=1 only); drop_p/seed: dropout on the output."""
    ws_bytes = query("rlt_gemm_workspace", ta, tb, M, Nn, K)
    ws = workspace(ws_bytes, C.device) if ws_bytes else None
    esz = 4
    call("rlt_gemm_ex", ta, tb, M, Nn, K,
         N.c_void_p(A.data_ptr() + a_off * esz), lda, N.c_void_p(B.data_ptr() + b_off * esz), ldb,
         N.c_void_p(C.data_ptr() + c_off * esz), ldc, ptr(bias), ptr(bias2), flags,
         ptr(relu_mask), ldmask, mask_scale, ptr(colsum_a), drop_p, seed, ptr(ws), ws_bytes,
         current_precision() if prec is None else prec, stream())


def gemm_bits(ta, tb, M, Nn, K, A, lda, B, ldb, C, ldc, bias=None, flags=0, bits_out=None, bits_in=None, mask_scale=1.0,
              drop_p=0.0, seed=0, prec=None):
    """rlt_gemm_bits: ReLU (+ dropout) forward that also emits a 1-bit mask of the surviving elements (bits_out,
    int32 (ceil(M/32), N): `alloc_relu_bits(M, N, device)`), or the masked backward product that consumes it (bits_in,
    mask_scale = 1/(1-p))."""
    call("rlt_gemm_bits", ta, tb, M, Nn, K, ptr(A), lda, ptr(B), ldb, ptr(C), ldc, ptr(bias), flags, drop_p, seed,
         ptr(bits_out), ptr(bits_in), mask_scale, current_precision() if prec is None else prec, stream())


def alloc_relu_bits(M, Nn, device):
    """Buffer for the 1-bit mask of rlt_gemm_bits: packed along rows, bit (row & 31) of word [(row >> 5), col]."""
    return torch.zeros(((M + 31) // 32, Nn), dtype=torch.int32, device=device)


def unpack_relu_bits(bits, M):
    """(ceil(M/32), N) int32 -> (M, N) bool (tests)."""
    sh = torch.arange(32, dtype=torch.int32, device=bits.device).view(1, 32, 1)
    return (((bits.unsqueeze(1) >> sh) & 1).reshape(-1, bits.shape[1])[:M]).bool()


def colsum(X, ldx, T, Nn, out, accumulate=0, x_off=0):
    ws_bytes = query("rlt_colsum_workspace", T, Nn)
    ws = workspace(ws_bytes, X.device)
    call("rlt_colsum", N.c_void_p(X.data_ptr() + 4 * x_off), ldx, T, Nn, ptr(out), accumulate, ptr(ws), ws_bytes, stream())


# ------------------------------------------------------------------------------ Linear
class LinearFn(Function):
    """y = x W^T + b (optional fused ReLU).  x (T,K), W (N,K), b (N)."""

    @staticmethod
    def forward(ctx, x, w, b, relu):
        T, K = x.shape
        Nn = w.shape[0]
        y = _empty((T, Nn), x)
        ctx.prec = current_precision()
        gemm(0, 1, T, Nn, K, x, K, w, K, y, Nn, bias=b, flags=N.GEMM_RELU if relu else 0, prec=ctx.prec)
        ctx.relu = relu
        ctx.save_for_backward(x, w, y if relu else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, y = ctx.saved_tensors
        T, K = x.shape
        Nn = w.shape[0]
        dy = N.f32c(dy)
        if ctx.relu:
            dy = dy.clone()
            call("rlt_relu_bwd", ptr(dy), ptr(y), dy.numel(), stream())
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = _empty((T, K), x)
            gemm(0, 0, T, K, Nn, dy, Nn, w, K, dx, K, prec=ctx.prec)
        if ctx.needs_input_grad[1]:
            dw = _empty((Nn, K), x)
            db = _empty((Nn,), x) if ctx.needs_input_grad[2] else None
            gemm(1, 0, Nn, K, T, dy, Nn, x, K, dw, K, colsum_a=db, prec=ctx.prec)      # db rides on the dW product
        elif ctx.needs_input_grad[2]:
            db = _empty((Nn,), x)
            colsum(dy, Nn, T, Nn, db)
        return dx, dw, db, None


def linear(x, w, b, relu=False):
    return LinearFn.apply(x, w, b, relu)


class FFNFn(Function):
    """y = relu(x W1^T + b1) W2^T + b2 (the encoder layer's feed-forward block) as one tape node:
    backward fuses the ReLU mask into the dH product and the bias gradients into the dW products."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, drop_p=0.0, seed=0):
        T, E = x.shape
        Fh = w1.shape[0]
        h = _empty((T, Fh), x)          # relu output, already dropped when drop_p > 0
        ctx.prec = pr = current_precision()
        gemm(0, 1, T, Fh, E, x, E, w1, E, h, Fh, bias=b1, flags=N.GEMM_RELU, drop_p=drop_p, seed=seed, prec=pr)
        y = _empty((T, w2.shape[0]), x)
        gemm(0, 1, T, w2.shape[0], Fh, h, Fh, w2, Fh, y, w2.shape[0], bias=b2, prec=pr)
        ctx.save_for_backward(x, w1, w2, h)
        ctx.drop_p = drop_p
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w1, w2, h = ctx.saved_tensors
        T, E = x.shape
        Fh, Eo = w1.shape[0], w2.shape[0]
        dy = N.f32c(dy)
        pr = ctx.prec
        dw2, db2 = _empty((Eo, Fh), x), _empty((Eo,), x)
        gemm(1, 0, Eo, Fh, T, dy, Eo, h, Fh, dw2, Fh, colsum_a=db2, prec=pr)
        dh = _empty((T, Fh), x)
        # dH = (dY W2) * (H > 0) [/ (1-p)]: a dropped element has H == 0, so one mask covers ReLU and dropout
        gemm(0, 0, T, Fh, Eo, dy, Eo, w2, Fh, dh, Fh, relu_mask=h, ldmask=Fh, mask_scale=1.0 / (1.0 - ctx.drop_p), prec=pr)
        dw1, db1 = _empty((Fh, E), x), _empty((Fh,), x)
        gemm(1, 0, Fh, E, T, dh, Fh, x, E, dw1, E, colsum_a=db1, prec=pr)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = _empty((T, E), x)
            gemm(0, 0, T, E, Fh, dh, Fh, w1, E, dx, E, prec=pr)
        return dx, dw1, db1, dw2, db2, None, None


def ffn(x, w1, b1, w2, b2, drop_p=0.0):
    return FFNFn.apply(x, w1, b1, w2, b2, drop_p, next_seed() if drop_p > 0 else 0)


# ------------------------------------------------------------------------------ residual + LayerNorm
class AddLayerNormFn(Function):
    @staticmethod
    def forward(ctx, x, r, gamma, beta, eps, drop_p=0.0, seed=0):
        T, E = x.shape
        y = _empty((T, E), x)
        stats = _empty((T, 2), x)
        call("rlt_add_layernorm_fwd", ptr(x), ptr(r), ptr(gamma), ptr(beta), T, E, eps, drop_p, seed,
             ptr(y), ptr(stats), stream())
        ctx.save_for_backward(x, r, gamma, stats)
        ctx.drop = (drop_p, seed)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, r, gamma, stats = ctx.saved_tensors
        T, E = x.shape
        dy = N.f32c(dy)
        drop_p, seed = ctx.drop
        dz = _empty((T, E), x)
        dr = _empty((T, E), x) if drop_p > 0 else None
        dgamma, dbeta = _empty((E,), x), _empty((E,), x)
        ws_bytes = query("rlt_add_layernorm_bwd_workspace", T, E)
        ws = workspace(ws_bytes, x.device)
        call("rlt_add_layernorm_bwd", ptr(x), ptr(r), ptr(gamma), ptr(stats), ptr(dy), T, E, drop_p, seed,
             ptr(dz), ptr(dr), ptr(dgamma), ptr(dbeta), 0, ptr(ws), ws_bytes, stream())
        return dz, (dz if dr is None else dr), dgamma, dbeta, None, None, None


def add_layernorm(x, r, gamma, beta, eps=1e-5, drop_p=0.0):
    """LayerNorm(x + dropout(r)) * gamma + beta."""
    return AddLayerNormFn.apply(x, r, gamma, beta, eps, drop_p, next_seed() if drop_p > 0 else 0)


# ------------------------------------------------------------------------------ list-axis attention
class ListAttentionFn(Function):
    """qkv (S*B, 3E) -> concatenated heads (S*B, E); attention over the B lists at each position."""

    @staticmethod
    def forward(ctx, qkv, S, B, H, drop_p=0.0, seed=0):
        E = qkv.shape[1] // 3
        HD = E // H
        out = _empty((S * B, E), qkv)
        lse = _empty((S, H, B), qkv)
        ctx.prec = pr = current_precision()
        img_bytes = query("rlt_list_attention_fwd_workspace", S, B, H, HD, drop_p, pr)
        images = workspace(img_bytes, qkv.device) if img_bytes else None     # pre-split tile records / images of this call
        _launch("attn_fwd", lambda: call("rlt_list_attention_fwd", ptr(qkv), S, B, H, HD, drop_p, seed,
                                         ptr(out), ptr(lse), ptr(images), img_bytes, pr, stream()))
        ctx.dims = (S, B, H, HD)
        ctx.drop = (drop_p, seed)
        # kept for the backward only where it reads them (split-bf16 records); the images of the pipelined bf16x6 forward kernels
        # are scratch of the call above - several GB per layer at the benchmark sizes
        ctx.images = images if images is not None and N.load().rlt_list_attention_images_retained(S, B, H, HD, pr) else None
        ctx.save_for_backward(qkv, out, lse)
        return out

    @staticmethod
    def backward(ctx, dout):
        qkv, out, lse = ctx.saved_tensors
        S, B, H, HD = ctx.dims
        images = ctx.images
        dout = N.f32c(dout)
        dqkv = torch.empty_like(qkv)
        pr = ctx.prec
        drop_p, seed = ctx.drop
        ws_bytes = query("rlt_list_attention_bwd_workspace", S, B, H, HD, drop_p, pr)
        ws = workspace(ws_bytes, qkv.device)
        call("rlt_list_attention_bwd_prepare", ptr(out), ptr(dout), ptr(lse), S, B, H, HD, drop_p, ptr(images), ptr(ws), ws_bytes, pr, stream())
        _launch("attn_bwd_dkv", lambda: call("rlt_list_attention_bwd_dkv", ptr(qkv), ptr(dout), ptr(lse), ptr(images), ptr(ws), ws_bytes,
                                             S, B, H, HD, drop_p, seed, ptr(dqkv), pr, stream()))
        _launch("attn_bwd_dq", lambda: call("rlt_list_attention_bwd_dq", ptr(qkv), ptr(dout), ptr(lse), ptr(images), ptr(ws), ws_bytes,
                                            S, B, H, HD, drop_p, seed, ptr(dqkv), pr, stream()))
        ctx.images = None
        return dqkv, None, None, None, None, None


def list_attention(qkv, S, B, H, drop_p=0.0):
    return ListAttentionFn.apply(qkv, S, B, H, drop_p, next_seed() if drop_p > 0 else 0)


# ------------------------------------------------------------------------------ whole encoder layer
class EncoderLayerFn(Function):
    """One post-norm nn.TransformerEncoderLayer (list-axis attention, ReLU FFN) as ONE tape node on the path-level
    entry points rlt_encoder_layer_fwd / _bwd: the launch sequence (in_proj, attention, out_proj, residual+LayerNorm,
    the FFN pair with its 1-bit ReLU mask, residual+LayerNorm), the stash layout and the in-place accumulation of the
    two fan-out gradients live in the library (csrc/path.hip); here: buffers and the tape."""

    @staticmethod
    def forward(ctx, x, in_w, in_b, out_w, out_b, n1_w, n1_b, w1, b1, w2, b2, n2_w, n2_b, S, B, H, eps, drop_p, seeds):
        T, E = x.shape
        FF = w1.shape[0]
        if T != S * B or in_w.shape != (3 * E, E) or w2.shape != (E, FF):
            raise RuntimeError(f"encoder layer: x {tuple(x.shape)} does not match S*B = {S * B} / the layer's weights")
        weights = (in_w, in_b, out_w, out_b, n1_w, n1_b, w1, b1, w2, b2, n2_w, n2_b)
        pr = current_precision()
        stash_bytes = query("rlt_workspace_bytes", N.OP_ENCODER_STASH, S, B, E, H, FF, 0, pr)
        ws_bytes = query("rlt_workspace_bytes", N.OP_ENCODER_FWD_WS, S, B, E, H, FF, 1 if drop_p > 0 else 0, pr)
        stash = N.byte_buffer(stash_bytes, x.device)
        ws = N.byte_buffer(ws_bytes, x.device)
        y = _empty((T, E), x)
        sarr = (N.c_uint32 * 4)(*seeds)
        wp = N.encoder_ptrs(weights)
        _launch("encoder_fwd", lambda: call("rlt_encoder_layer_fwd", ptr(x), N.ctypes.byref(wp), S, B, E, H, FF, eps, drop_p, sarr,
                                            ptr(y), ptr(stash), stash_bytes, ptr(ws), ws_bytes, pr, stream()))
        ctx.cfg = (S, B, E, H, FF, eps, drop_p, tuple(seeds), stash_bytes, pr)
        ctx.stash = stash
        ctx.save_for_backward(x, *weights)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, *weights = ctx.saved_tensors
        S, B, E, H, FF, eps, drop_p, seeds, stash_bytes, pr = ctx.cfg
        if ctx.stash is None:
            raise RuntimeError("EncoderLayerFn.backward frees its stash and can run only once")
        dy = N.f32c(dy)
        dx = torch.empty_like(x)
        grads = [torch.empty_like(w) for w in weights]
        ws_bytes = query("rlt_workspace_bytes", N.OP_ENCODER_BWD_WS, S, B, E, H, FF, 1 if drop_p > 0 else 0, pr)
        ws = N.byte_buffer(ws_bytes, x.device)
        sarr = (N.c_uint32 * 4)(*seeds)
        wp, gp = N.encoder_ptrs(weights), N.encoder_ptrs(grads)
        _launch("encoder_bwd", lambda: call("rlt_encoder_layer_bwd", ptr(x), N.ctypes.byref(wp), S, B, E, H, FF, eps, drop_p, sarr,
                                            ptr(dy), ptr(ctx.stash), stash_bytes, ptr(dx), N.ctypes.byref(gp),
                                            ptr(ws), ws_bytes, pr, stream()))
        ctx.stash = None
        return (dx, *grads, None, None, None, None, None, None)


class EncoderLayerKernelsFn(Function):
    """The encoder layer composed from the KERNEL-level entry points, launch by launch, on the host: the same launches
    in the same order as rlt_encoder_layer_fwd / _bwd (csrc/path.hip) make inside the library.  Not the product path -
    it exists so that bench.py can bracket the three attention launches of a REAL training step with HIP events
    (`KernelTimer.active`), and as a cross-check of the path-level composition in the tests.

    Same kernels as LinearFn / ListAttentionFn / AddLayerNormFn / FFNFn chained by hand; what the single node buys is
    the two fan-out points of the layer (x feeds in_proj and the first residual; LN1's output feeds the FFN and the
    second residual): the dX products of in_proj and linear1 accumulate straight into the residual gradient
    (RLT_GEMM_ACCUMULATE) instead of autograd materialising both and adding them in a separate pass (two 1.26 GB
    adds per layer at B=4096)."""

    @staticmethod
    def forward(ctx, *args):
        ctx.prec = current_precision()
        with precision(ctx.prec):          # every gemm() below reads the scope
            return EncoderLayerKernelsFn._forward(ctx, *args)

    @staticmethod
    def backward(ctx, dy):
        with precision(ctx.prec):
            return EncoderLayerKernelsFn._backward(ctx, dy)

    @staticmethod
    def _forward(ctx, x, in_w, in_b, out_w, out_b, n1_w, n1_b, w1, b1, w2, b2, n2_w, n2_b, S, B, H, eps, drop_p, seeds):
        T, E = x.shape
        HD = E // H
        Fh = w1.shape[0]
        pr = ctx.prec
        s_attn, s_ln1, s_ffn, s_ln2 = seeds
        qkv = _empty((T, 3 * E), x)
        gemm(0, 1, T, 3 * E, E, x, E, in_w, E, qkv, 3 * E, bias=in_b)
        att = _empty((T, E), x)
        lse = _empty((S, H, B), x)
        img_bytes = query("rlt_list_attention_fwd_workspace", S, B, H, HD, drop_p, pr)
        images = workspace(img_bytes, x.device) if img_bytes else None
        _launch("attn_fwd", lambda: call("rlt_list_attention_fwd", ptr(qkv), S, B, H, HD, drop_p, s_attn,
                                         ptr(att), ptr(lse), ptr(images), img_bytes, pr, stream()))
        if images is not None and not N.load().rlt_list_attention_images_retained(S, B, H, HD, pr):
            images = None                                    # scratch of the forward call: not kept for the backward
        proj = _empty((T, E), x)
        gemm(0, 1, T, E, E, att, E, out_w, E, proj, E, bias=out_b)
        h1 = _empty((T, E), x)
        st1 = _empty((T, 2), x)
        call("rlt_add_layernorm_fwd", ptr(x), ptr(proj), ptr(n1_w), ptr(n1_b), T, E, eps, drop_p, s_ln1, ptr(h1), ptr(st1), stream())
        hid = _empty((T, Fh), x)
        relu_bits = None
        if Fh % 32 == 0:
            # 1-bit mask (passed the ReLU and kept by the dropout) for the backward dH product, which then reads
            # T*Fh/8 bytes instead of the 4*T*Fh of `hid`
            relu_bits = alloc_relu_bits(T, Fh, x.device)
            gemm_bits(0, 1, T, Fh, E, h1, E, w1, E, hid, Fh, bias=b1, flags=N.GEMM_RELU, bits_out=relu_bits,
                      drop_p=drop_p, seed=s_ffn)
        else:
            gemm(0, 1, T, Fh, E, h1, E, w1, E, hid, Fh, bias=b1, flags=N.GEMM_RELU, drop_p=drop_p, seed=s_ffn)
        ff = _empty((T, E), x)
        gemm(0, 1, T, E, Fh, hid, Fh, w2, Fh, ff, E, bias=b2)
        y = _empty((T, E), x)
        st2 = _empty((T, 2), x)
        call("rlt_add_layernorm_fwd", ptr(h1), ptr(ff), ptr(n2_w), ptr(n2_b), T, E, eps, drop_p, s_ln2, ptr(y), ptr(st2), stream())
        ctx.cfg = (S, B, H, HD, eps, drop_p, seeds)
        ctx.images = images
        ctx.relu_bits = relu_bits
        ctx.save_for_backward(x, in_w, out_w, n1_w, w1, w2, n2_w, qkv, att, lse, proj, st1, h1, hid, ff, st2)
        return y

    @staticmethod
    def _backward(ctx, dy):
        x, in_w, out_w, n1_w, w1, w2, n2_w, qkv, att, lse, proj, st1, h1, hid, ff, st2 = ctx.saved_tensors
        S, B, H, HD, eps, drop_p, (s_attn, s_ln1, s_ffn, s_ln2) = ctx.cfg
        pr = ctx.prec
        T, E = x.shape
        Fh = w1.shape[0]
        dy = N.f32c(dy)

        def ln_bwd(xx, rr, gamma, stats, dyy, seed):
            dz = _empty((T, E), x)
            dr = _empty((T, E), x) if drop_p > 0 else None
            dg, db = _empty((E,), x), _empty((E,), x)
            ws_bytes = query("rlt_add_layernorm_bwd_workspace", T, E)
            ws = workspace(ws_bytes, x.device)
            call("rlt_add_layernorm_bwd", ptr(xx), ptr(rr), ptr(gamma), ptr(stats), ptr(dyy), T, E, drop_p, seed,
                 ptr(dz), ptr(dr), ptr(dg), ptr(db), 0, ptr(ws), ws_bytes, stream())
            return dz, (dz if dr is None else dr), dg, db

        # LN2: dz2 = gradient of h1 through the residual, dr2 = gradient of the FFN branch output
        dz2, dr2, dn2_w, dn2_b = ln_bwd(h1, ff, n2_w, st2, dy, s_ln2)
        dw2, db2 = _empty((E, Fh), x), _empty((E,), x)
        gemm(1, 0, E, Fh, T, dr2, E, hid, Fh, dw2, Fh, colsum_a=db2)
        dhid = _empty((T, Fh), x)
        if ctx.relu_bits is not None:
            gemm_bits(0, 0, T, Fh, E, dr2, E, w2, Fh, dhid, Fh, bits_in=ctx.relu_bits, mask_scale=1.0 / (1.0 - drop_p))
            ctx.relu_bits = None
        else:
            gemm(0, 0, T, Fh, E, dr2, E, w2, Fh, dhid, Fh, relu_mask=hid, ldmask=Fh, mask_scale=1.0 / (1.0 - drop_p))
        dw1, db1 = _empty((Fh, E), x), _empty((Fh,), x)
        gemm(1, 0, Fh, E, T, dhid, Fh, h1, E, dw1, E, colsum_a=db1)
        gemm(0, 0, T, E, Fh, dhid, Fh, w1, E, dz2, E, flags=N.GEMM_ACCUMULATE)       # dh1 = dz2 + dhid W1, in place
        del dhid
        # LN1
        dz1, dr1, dn1_w, dn1_b = ln_bwd(x, proj, n1_w, st1, dz2, s_ln1)
        dw_o, db_o = _empty((E, E), x), _empty((E,), x)
        gemm(1, 0, E, E, T, dr1, E, att, E, dw_o, E, colsum_a=db_o)
        datt = _empty((T, E), x)
        gemm(0, 0, T, E, E, dr1, E, out_w, E, datt, E)
        # attention
        images = ctx.images
        dqkv = torch.empty_like(qkv)
        ws_bytes = query("rlt_list_attention_bwd_workspace", S, B, H, HD, drop_p, pr)
        ws = workspace(ws_bytes, x.device)
        call("rlt_list_attention_bwd_prepare", ptr(att), ptr(datt), ptr(lse), S, B, H, HD, drop_p, ptr(images), ptr(ws), ws_bytes, pr, stream())
        _launch("attn_bwd_dkv", lambda: call("rlt_list_attention_bwd_dkv", ptr(qkv), ptr(datt), ptr(lse), ptr(images), ptr(ws), ws_bytes,
                                             S, B, H, HD, drop_p, s_attn, ptr(dqkv), pr, stream()))
        _launch("attn_bwd_dq", lambda: call("rlt_list_attention_bwd_dq", ptr(qkv), ptr(datt), ptr(lse), ptr(images), ptr(ws), ws_bytes,
                                            S, B, H, HD, drop_p, s_attn, ptr(dqkv), pr, stream()))
        ctx.images = None
        # in_proj
        dw_in, db_in = _empty((3 * E, E), x), _empty((3 * E,), x)
        gemm(1, 0, 3 * E, E, T, dqkv, 3 * E, x, E, dw_in, E, colsum_a=db_in)
        gemm(0, 0, T, E, 3 * E, dqkv, 3 * E, in_w, E, dz1, E, flags=N.GEMM_ACCUMULATE)    # dx = dz1 + dqkv W_in, in place
        return (dz1, dw_in, db_in, dw_o, db_o, dn1_w, dn1_b, dw1, db1, dw2, db2, dn2_w, dn2_b,
                None, None, None, None, None, None)


def encoder_layer(x, layer, S, B, H, drop_p=0.0, eps=1e-5):
    """`layer`: a ParamTree mirror of nn.TransformerEncoderLayer."""
    att = layer.self_attn
    seeds = tuple(next_seed() for _ in range(4)) if drop_p > 0 else (0, 0, 0, 0)
    fn = EncoderLayerKernelsFn if (KernelTimer.active is not None or KERNEL_LEVEL_ENCODER[0]) else EncoderLayerFn
    return fn.apply(x, att.in_proj_weight, att.in_proj_bias, att.out_proj.weight, att.out_proj.bias,
                                layer.norm1.weight, layer.norm1.bias, layer.linear1.weight, layer.linear1.bias,
                                layer.linear2.weight, layer.linear2.bias, layer.norm2.weight, layer.norm2.bias,
                                S, B, H, eps, drop_p, seeds)


def encoder_ffn_hidden(y):
    """Tests only: the FFN hidden activation (T, FF) inside the stash of the encoder-layer tape node `y` (or the node
    that produced the tensor `y`); offsets: enc_stash() in csrc/path.hip; valid until that node's backward has run."""
    node = y.grad_fn if torch.is_tensor(y) else y
    if type(node).__name__ != "EncoderLayerFnBackward" or node.stash is None:
        raise RuntimeError("not the output of a live encoder-layer tape node")
    S, B, E, H, FF = node.cfg[:5]
    T, rup = S * B, lambda n: (n + 255) // 256 * 256
    off = rup(T * 3 * E * 4) + rup(T * E * 4) + rup(S * H * B * 4) + rup(T * E * 4) + rup(T * 2 * 4) + rup(T * E * 4)
    return node.stash[off:off + T * FF * 4].view(torch.float32).view(T, FF)


# ------------------------------------------------------------------------------ 2-layer BiLSTM (H = 128)
class BiLSTMFn(Function):
    """nn.LSTM(input, hidden, num_layers=2, batch_first=True, bidirectional=True) on position-major input x (S*B, I) ->
    (S*B, 2*hidden), ONE tape node on the path-level entry points: hidden 128 (what the reference hard-codes outside
    MMOECut's `encoding_size` argument) on the persistent recurrence kernels (rlt_bilstm_fwd / _bwd, csrc/path.hip),
    any other hidden size on the general step-by-step form (rlt_bilstm_generic_fwd / _bwd, csrc/lstm_generic.hip)."""

    @staticmethod
    def forward(ctx, x, S, B, *w):          # w: 16 tensors, per layer w_ih_f, w_hh_f, b_ih_f, b_hh_f, w_ih_r, w_hh_r, b_ih_r, b_hh_r
        T, I = x.shape
        if len(w) != 16:
            raise RuntimeError("BiLSTMFn takes the 16 parameters of a 2-layer bidirectional LSTM")
        Hd = w[1].shape[1]
        if w[0].shape != (4 * Hd, I) or w[4].shape != (4 * Hd, I) or w[8].shape != (4 * Hd, 2 * Hd) or w[1].shape != (4 * Hd, Hd) or T != S * B:
            raise RuntimeError(f"BiLSTM input has {I} features x {T} rows; the layers were built for "
                               f"{w[0].shape[1]} features, hidden {Hd} and S*B = {S * B} rows")
        fast = Hd == 128
        pr = current_precision()
        if fast:
            stash_bytes = query("rlt_workspace_bytes", N.OP_BILSTM_STASH, S, B, I, 0, 0, 0, pr)
            ws_bytes = query("rlt_workspace_bytes", N.OP_BILSTM_WS, S, B, I, 0, 0, 0, pr)
        else:
            stash_bytes = query("rlt_bilstm_generic_bytes", 1, S, B, I, Hd)
            ws_bytes = query("rlt_bilstm_generic_bytes", 0, S, B, I, Hd)
        stash = N.byte_buffer(stash_bytes, x.device)
        ws = N.byte_buffer(ws_bytes, x.device)
        h = _empty((T, 2 * Hd), x)
        wp = N.lstm_ptrs([w[0:8], w[8:16]])
        if fast:
            _launch("bilstm_fwd", lambda: call("rlt_bilstm_fwd", ptr(x), I, wp, S, B, ptr(h), ptr(stash), stash_bytes,
                                               ptr(ws), ws_bytes, pr, stream()))
        else:
            call("rlt_bilstm_generic_fwd", ptr(x), I, Hd, wp, S, B, ptr(h), ptr(stash), stash_bytes, ptr(ws), ws_bytes, pr, stream())
        ctx.cfg = (S, B, I, Hd, stash_bytes, ws_bytes, pr)
        ctx.stash = stash
        ctx.save_for_backward(x, h, *w)
        return h

    @staticmethod
    def backward(ctx, dh):
        if ctx.stash is None:
            raise RuntimeError("BiLSTMFn.backward overwrites its stash in place and can run only once")
        x, h, *w = ctx.saved_tensors
        S, B, I, Hd, stash_bytes, ws_bytes, pr = ctx.cfg
        dh = N.f32c(dh)
        grads = [torch.empty_like(t) for t in w]
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        ws = N.byte_buffer(ws_bytes, x.device)
        wp, gp = N.lstm_ptrs([w[0:8], w[8:16]]), N.lstm_ptrs([grads[0:8], grads[8:16]])
        if Hd == 128:
            _launch("bilstm_bwd", lambda: call("rlt_bilstm_bwd", ptr(x), I, wp, ptr(h), ptr(dh), S, B, ptr(ctx.stash), stash_bytes,
                                               ptr(dx), gp, ptr(ws), ws_bytes, pr, stream()))
        else:
            call("rlt_bilstm_generic_bwd", ptr(x), I, Hd, wp, ptr(h), ptr(dh), S, B, ptr(ctx.stash), stash_bytes,
                 ptr(dx), gp, ptr(ws), ws_bytes, pr, stream())
        ctx.stash = None
        return (dx, None, None, *grads)


def bilstm(x, params, S, B):
    """`params`: a ParamTree mirror of the reference's nn.LSTM (state_dict names weight_ih_l0 ... bias_hh_l1_reverse)."""
    names = ("weight_ih", "weight_hh", "bias_ih", "bias_hh")
    w = [getattr(params, f"{n}_l{layer}{suffix}") for layer in (0, 1) for suffix in ("", "_reverse") for n in names]
    return BiLSTMFn.apply(x, S, B, *w)


# ------------------------------------------------------------------------------ layout
class ToPositionMajorFn(Function):
    """(B,S,F) -> (S*B,F)."""

    @staticmethod
    def forward(ctx, x):
        B, S, F = x.shape
        out = _empty((S * B, F), x)
        call("rlt_to_position_major", ptr(x), B, S, F, ptr(out), stream())
        ctx.dims = (B, S, F)
        return out

    @staticmethod
    def backward(ctx, dout):
        B, S, F = ctx.dims
        dout = N.f32c(dout)
        dx = _empty((B, S, F), dout)
        call("rlt_from_position_major", ptr(dout), B, S, F, ptr(dx), stream())
        return dx


def to_position_major(x):
    return ToPositionMajorFn.apply(x)


class ChoopyEmbedFn(Function):
    """cat(score, position_encoding) in position-major layout: (B,S,1),(S,E-1) -> (S*B,E)."""

    @staticmethod
    def forward(ctx, score, pe):
        B, S = score.shape[0], score.shape[1]
        E = pe.shape[1] + 1
        out = _empty((S * B, E), score)
        call("rlt_choopy_embed", ptr(score), ptr(pe), B, S, E, ptr(out), stream())
        ctx.dims = (B, S, E)
        return out

    @staticmethod
    def backward(ctx, dout):
        B, S, E = ctx.dims
        dout = N.f32c(dout)
        dpe = _empty((S, E - 1), dout)
        # dPE[s] = sum over the B rows of position s, columns 1..E-1
        call("rlt_segment_colsum", N.c_void_p(dout.data_ptr() + 4), E, S, B, E - 1, ptr(dpe), E - 1, 0, stream())
        dscore = None
        if ctx.needs_input_grad[0]:
            col = dout[:, 0].contiguous().view(S, B, 1)
            dscore = _empty((B, S, 1), dout)
            call("rlt_from_position_major", ptr(col), B, S, 1, ptr(dscore), stream())
        return dscore, dpe


def choopy_embed(score, pe):
    return ChoopyEmbedFn.apply(score, pe)


# ------------------------------------------------------------------------------ heads
class HeadsFn(Function):
    """n Linear(E,1) heads + {softmax over positions | sigmoid | identity}: x (S*B,E) -> (n,B,S)."""

    @staticmethod
    def forward(ctx, x, w, b, kinds, S, B):
        n, E = w.shape
        out = _empty((n, B, S), x)
        karr = (N.c_int * n)(*kinds)
        call("rlt_heads_fwd", ptr(x), ptr(w), ptr(b), karr, n, S, B, E, ptr(out), stream())
        ctx.meta = (tuple(kinds), S, B)
        ctx.save_for_backward(x, w, out)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, w, out = ctx.saved_tensors
        kinds, S, B = ctx.meta
        n, E = w.shape
        dout = N.f32c(dout)
        dx = _empty((S * B, E), x)
        dw, db = _empty((n, E), x), _empty((n,), x)
        karr = (N.c_int * n)(*kinds)
        ws_bytes = query("rlt_heads_bwd_workspace", n, S, B, E)
        ws = workspace(ws_bytes, x.device)
        call("rlt_heads_bwd", ptr(x), ptr(w), karr, n, ptr(out), ptr(dout), S, B, E, ptr(dx), 0, ptr(dw), ptr(db),
             ptr(ws), ws_bytes, stream())
        return dx, dw, db, None, None, None


def heads(x, weights, biases, kinds, S, B):
    """weights: list of (1,E) parameters, biases: list of (1,) parameters (glue: tiny cat)."""
    w = torch.cat([wi.reshape(1, -1) for wi in weights], dim=0) if len(weights) > 1 else weights[0].reshape(1, -1)
    b = torch.cat([bi.reshape(1) for bi in biases], dim=0) if len(biases) > 1 else biases[0].reshape(1)
    out = HeadsFn.apply(x, w, b, kinds, S, B)
    return [out[i].unsqueeze(2) for i in range(len(kinds))]          # each (B,S,1)


# ------------------------------------------------------------------------------ MMOE
class MMOEGateFn(Function):
    """gates (n_tasks,B,n_e) = softmax_e(flatten_s(h[b]) @ w_gate[t])."""

    @staticmethod
    def forward(ctx, h, S, B, *w_gates):
        C = h.shape[1]
        nt, ne = len(w_gates), w_gates[0].shape[1]
        gates = _empty((nt, B, ne), h)
        warr = N.pointer_array(w_gates)
        call("rlt_mmoe_gate_fwd", ptr(h), warr, nt, ne, S, B, C, ptr(gates), stream())
        ctx.dims = (S, B, C, nt, ne)
        ctx.save_for_backward(h, gates, *w_gates)
        return gates

    @staticmethod
    def backward(ctx, dgates):
        h, gates, *w_gates = ctx.saved_tensors
        S, B, C, nt, ne = ctx.dims
        dgates = N.f32c(dgates)
        dh = torch.empty_like(h)
        dws = [torch.empty_like(w) for w in w_gates]
        ws_bytes = query("rlt_mmoe_gate_bwd_workspace", nt, ne, S, B, C)
        ws = workspace(ws_bytes, h.device)
        call("rlt_mmoe_gate_bwd", ptr(h), N.pointer_array(w_gates), ptr(gates), ptr(dgates), nt, ne, S, B, C,
             ptr(dh), 0, N.pointer_array(dws), ptr(ws), ws_bytes, stream())
        return (dh, None, None, *dws)


class MMOEMixFn(Function):
    """mixed (n_tasks,S*B,E) = sum_e gates[t][b][e] * expert_e."""

    @staticmethod
    def forward(ctx, gates, S, B, *experts):
        nt, _, ne = gates.shape
        E = experts[0].shape[1]
        mixed = _empty((nt, S * B, E), gates)
        call("rlt_mmoe_mix_fwd", N.pointer_array(experts), ptr(gates), nt, ne, S, B, E, ptr(mixed), stream())
        ctx.dims = (S, B, E, nt, ne)
        ctx.save_for_backward(gates, *experts)
        return mixed

    @staticmethod
    def backward(ctx, dmixed):
        gates, *experts = ctx.saved_tensors
        S, B, E, nt, ne = ctx.dims
        dmixed = N.f32c(dmixed)
        dex = [torch.empty_like(e) for e in experts]
        dgates = torch.empty_like(gates)
        call("rlt_mmoe_mix_bwd", N.pointer_array(experts), ptr(gates), ptr(dmixed), nt, ne, S, B, E,
             N.pointer_array(dex), ptr(dgates), stream())
        return (dgates, None, None, *dex)


# ------------------------------------------------------------------------------ losses
_COEF_CACHE = {}


def dcg_coef(S, device):
    """log2(j+2) table exactly as the reference builds it (utils/metrics.py:7), fp32 on device."""
    key = (S, str(device))
    if key not in _COEF_CACHE:
        _COEF_CACHE[key] = torch.tensor([math.log(j + 2, 2) for j in range(S)], dtype=torch.float32, device=device)
    return _COEF_CACHE[key]


_DCG_TABLE_CACHE = {}


def dcg_table(device):
    """The float64 DCG coefficient table of rlt_loss_metrics (1 / log2(j + 2) and its prefix sums), filled once per device by
    rlt_dcg_table_init on the current stream: caller memory, the library keeps no state of its own."""
    device = torch.device(device)
    key = device.index if device.index is not None else torch.cuda.current_device()      # 'cuda' and 'cuda:0' are one device
    if key not in _DCG_TABLE_CACHE:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("the DCG table must exist before a hipGraph capture starts: call ops.dcg_table(device) once, eagerly")
        nbytes = query("rlt_dcg_table_bytes")
        with torch.cuda.device(key):
            t = torch.empty((nbytes // 8,), dtype=torch.float64, device=torch.device("cuda", key))
            call("rlt_dcg_table_init", ptr(t), nbytes, stream())
            # filled once, read from whatever stream a later loss call runs on: nothing orders those streams behind this one
            torch.cuda.current_stream().synchronize()
        _DCG_TABLE_CACHE[key] = t
    return _DCG_TABLE_CACHE[key]


def _fused_metric_buffers(B, dev):
    return (torch.empty((B,), dtype=torch.int32, device=dev), torch.empty((B,), dtype=torch.float64, device=dev),
            torch.empty((B,), dtype=torch.float64, device=dev), torch.empty((2,), dtype=torch.float64, device=dev))


class RewardLossFn(Function):
    """Fused reward matrix + {Choopy | AttnCut | KL | JS} loss; returns a 0-d tensor.  `penalty`: the DCG gain numerator of
    a non-relevant document (Metric_for_Loss.dcg's argument, utils/metrics.py:94).  With `with_metrics` the same kernel
    pass also yields the cut metrics of run.py:141-145 and the call returns (loss, k (B) int32, sums (2) float64 =
    [sum F1@k, sum DCG@k])."""

    @staticmethod
    def forward(ctx, p, labels, metric, kind, tau, penalty=-1.0, with_metrics=False, metric_penalty=-1.0):
        B, S = labels.shape
        coef = dcg_coef(S, p.device) if metric == N.METRIC_DCG else None
        per_list = _empty((B,), p)
        loss = _empty((1,), p)
        dp = _empty((B, S), p)
        ctx.save_for_backward(dp)
        ctx.pshape = p.shape
        if not with_metrics:
            call("rlt_reward_loss_ex", ptr(p), ptr(labels), ptr(coef), B, S, metric, penalty, kind, tau,
                 ptr(per_list), ptr(loss), ptr(dp), stream())
            return loss.reshape(())
        k, f1, dcg, sums = _fused_metric_buffers(B, p.device)
        ws_bytes = query("rlt_loss_metrics_workspace", B)
        ws = workspace(ws_bytes, p.device)
        call("rlt_loss_metrics", ptr(p), ptr(labels), ptr(coef), B, S, metric, penalty, kind, tau, metric_penalty,
             ptr(per_list), ptr(loss), ptr(dp), ptr(k), ptr(f1), ptr(dcg), ptr(sums), ptr(dcg_table(p.device)), ptr(ws), ws_bytes, stream())
        ctx.mark_non_differentiable(k, sums)
        return loss.reshape(()), k, sums

    @staticmethod
    def backward(ctx, go, *_unused):
        (dp,) = ctx.saved_tensors
        g = dp.clone()
        go = N.f32c(go).reshape(1)
        call("rlt_scale", ptr(g), ptr(go), g.numel(), stream())
        return g.view(ctx.pshape), None, None, None, None, None, None, None


class WassDistLossFn(Function):
    """utils/losses.py:236-311: Sinkhorn forward recorded in a workspace, explicit reverse sweep for the gradient."""

    @staticmethod
    def forward(ctx, p, labels, eps, max_iter, thresh):
        B, S = labels.shape
        loss = _empty((1,), p)
        ws_bytes = query("rlt_wass_loss_workspace", B, max_iter)
        ws = workspace(ws_bytes, p.device)
        call("rlt_wass_loss_fwd", ptr(p), ptr(labels), B, S, eps, max_iter, thresh, ptr(loss), ptr(ws), ws_bytes, stream())
        ctx.cfg = (B, S, eps, max_iter, ws_bytes)
        ctx.ws = ws
        ctx.save_for_backward(p, labels)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, go):
        p, labels = ctx.saved_tensors
        B, S, eps, max_iter, ws_bytes = ctx.cfg
        dp = _empty((B, S), p)
        go = N.f32c(go).reshape(1)
        call("rlt_wass_loss_bwd", ptr(p), ptr(labels), ptr(go), B, S, eps, max_iter, ptr(ctx.ws), ws_bytes, ptr(dp), stream())
        ctx.ws = None
        return dp, None, None, None, None


class PairSoftmaxFn(Function):
    """BiCut's two-class head: position-major logits (S*B, 2) -> dropout -> softmax over the classes -> (B,S,2)."""

    @staticmethod
    def forward(ctx, z, S, B, drop_p=0.0, seed=0):
        out = _empty((B, S, 2), z)
        call("rlt_pair_softmax_fwd", ptr(z), B, S, drop_p, seed, ptr(out), stream())
        ctx.cfg = (S, B, drop_p, seed)
        ctx.save_for_backward(out)
        return out

    @staticmethod
    def backward(ctx, dout):
        (out,) = ctx.saved_tensors
        S, B, drop_p, seed = ctx.cfg
        dz = _empty((S * B, 2), out)
        call("rlt_pair_softmax_bwd", ptr(out), ptr(N.f32c(dout)), B, S, drop_p, seed, ptr(dz), stream())
        return dz, None, None, None, None


def pair_softmax(z, S, B, drop_p=0.0):
    return PairSoftmaxFn.apply(z, S, B, drop_p, next_seed() if drop_p > 0 else 0)


class BiCutLossFn(Function):
    """utils/losses.py:11-45 with its gradient, one launch."""

    @staticmethod
    def forward(ctx, out, labels, nci, alpha, r):
        B, S = labels.shape
        per_list = _empty((B,), out)
        loss = _empty((1,), out)
        dout = _empty((B, S, 2), out)
        call("rlt_bicut_loss", ptr(out), ptr(labels), B, S, int(nci), alpha, r, ptr(per_list), ptr(loss), ptr(dout), stream())
        ctx.save_for_backward(dout)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, go):
        (dout,) = ctx.saved_tensors
        g = dout.clone()
        go = N.f32c(go).reshape(1)
        call("rlt_scale", ptr(g), ptr(go), g.numel(), stream())
        return g, None, None, None, None


class MtCutLossFn(Function):
    """cut JS loss + w_r * rerank hinge + w_c * BCE, one tape node (utils/losses.py:180-191).  `with_metrics`: the cut term's
    kernel pass also yields the cut metrics; returns (loss, k, sums) as RewardLossFn does."""

    @staticmethod
    def forward(ctx, cut_p, rerank, cls, labels, metric, tau, w_r, w_c, margin, with_metrics=False):
        B, S = labels.shape
        dev = cut_p.device
        coef = dcg_coef(S, dev) if metric == N.METRIC_DCG else None
        per_list = _empty((B,), cut_p)
        cut = _empty((1,), cut_p)
        dp = _empty((B, S), cut_p)
        if with_metrics:
            k, f1, dcg, sums = _fused_metric_buffers(B, dev)
            lws_bytes = query("rlt_loss_metrics_workspace", B)
            lws = workspace(lws_bytes, dev)
            call("rlt_loss_metrics", ptr(cut_p), ptr(labels), ptr(coef), B, S, metric, -1.0, N.LOSS_JS, tau, -1.0,
                 ptr(per_list), ptr(cut), ptr(dp), ptr(k), ptr(f1), ptr(dcg), ptr(sums), ptr(dcg_table(dev)), ptr(lws), lws_bytes, stream())
        else:
            call("rlt_reward_loss", ptr(cut_p), ptr(labels), ptr(coef), B, S, metric, N.LOSS_JS, tau,
                 ptr(per_list), ptr(cut), ptr(dp), stream())
        terms = _empty((4,), cut_p)
        ws_bytes = query("rlt_mt_terms_workspace", B, S)
        ws = workspace(ws_bytes, dev)
        call("rlt_mt_terms", ptr(rerank), ptr(cls), ptr(labels), B, S, margin, ptr(terms), ptr(ws), ws_bytes, stream())
        # loss = cut + w_r * hinge + w_c * bce, in the reference's order of additions
        xs, ws_ = [cut], [1.0]
        if rerank is not None:
            xs.append(terms[0:1]); ws_.append(w_r)
        if cls is not None:
            xs.append(terms[1:2]); ws_.append(w_c)
        loss = _empty((1,), cut_p)
        warr = (N.c_float * len(ws_))(*ws_)
        call("rlt_weighted_sum", N.pointer_array(xs), warr, len(xs), ptr(loss), stream())
        ctx.save_for_backward(dp, cls, labels, terms)
        ctx.meta = (w_r, w_c, rerank is not None, cut_p.shape, None if rerank is None else rerank.shape,
                    None if cls is None else cls.shape)
        if with_metrics:
            ctx.mark_non_differentiable(k, sums)
            return loss.reshape(()), k, sums
        return loss.reshape(())

    @staticmethod
    def backward(ctx, go, *_unused):
        dp, cls, labels, terms = ctx.saved_tensors
        w_r, w_c, has_rr, pshape, rshape, cshape = ctx.meta
        B, S = labels.shape
        go = N.f32c(go).reshape(1)
        g = dp.clone()
        call("rlt_scale", ptr(g), ptr(go), g.numel(), stream())
        d_rr = _empty((B, S), dp) if has_rr else None
        d_cl = _empty((B, S), dp) if cls is not None else None
        if has_rr or cls is not None:
            call("rlt_mt_terms_bwd", ptr(cls), ptr(labels), ptr(terms), B, S, w_r, w_c, ptr(go), ptr(d_rr), ptr(d_cl), stream())
        return (g.view(pshape), None if d_rr is None else d_rr.view(rshape), None if d_cl is None else d_cl.view(cshape),
                None, None, None, None, None, None, None)


class RerankLossFn(Function):
    """Stand-alone RerankLoss (utils/losses.py:99-141)."""

    @staticmethod
    def forward(ctx, score, labels, margin):
        B, S = labels.shape
        terms = _empty((4,), score)
        ws_bytes = query("rlt_mt_terms_workspace", B, S)
        ws = workspace(ws_bytes, score.device)
        call("rlt_mt_terms", ptr(score), None, ptr(labels), B, S, margin, ptr(terms), ptr(ws), ws_bytes, stream())
        ctx.save_for_backward(labels, terms)
        ctx.sshape = score.shape
        return terms[0].clone()

    @staticmethod
    def backward(ctx, go):
        labels, terms = ctx.saved_tensors
        B, S = labels.shape
        go = N.f32c(go).reshape(1)
        d = _empty((B, S), labels)
        call("rlt_mt_terms_bwd", None, ptr(labels), ptr(terms), B, S, 1.0, 0.0, ptr(go), ptr(d), None, stream())
        return d.view(ctx.sshape), None, None


# ------------------------------------------------------------------------------ metrics
def cut_metrics(p, labels, k_in=None, penalty=-1.0):
    """Per-list (k, F1@k, DCG@k) on device; p (B,S) or (B,S,1), labels (B,S).  Returns tensors."""
    B, S = labels.shape
    dev = labels.device
    k, f1, dcg, sums = _fused_metric_buffers(B, dev)
    call("rlt_cut_metrics_ex", ptr(p), ptr(labels), ptr(k_in), B, S, float(penalty), ptr(k), ptr(f1), ptr(dcg), ptr(sums), stream())
    return k, f1, dcg, sums


def reward_matrix(labels, metric, tau=1.0, want_q=False, penalty=-1.0):
    B, S = labels.shape
    coef = dcg_coef(S, labels.device) if metric == N.METRIC_DCG else None
    r = _empty((B, S), labels)
    q = _empty((B, S), labels) if want_q else None
    call("rlt_reward_matrix_ex", ptr(labels), ptr(coef), B, S, metric, float(penalty), tau, ptr(r), ptr(q), stream())
    return (r, q) if want_q else r
