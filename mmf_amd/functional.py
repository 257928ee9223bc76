"""Autograd surface of the MI355X kernels: `torch.autograd.Function`s whose forward AND backward
are hand-written HIP kernels called through the C ABI (mmf_amd/_native.py -> libmmf_amd.so).

Activations travel in bf16 (HBM-resident, token-major `[B*S, H]`), parameters stay fp32 masters
with the reference's names and shapes; each Function takes the bf16 *shadow* of its weights (see
`ShadowCache`) next to the fp32 parameter so that autograd still delivers fp32 parameter gradients.
Nothing here falls back to eager PyTorch math: torch is used for allocation, views and autograd
bookkeeping only.

Reference ops replaced (paths relative to the reference root):
  LinearFn                  nn.Linear                                      (hf_layers.py:169-180, visual_bert.py:330)
  SelfAttentionFn           BertSelfAttentionJit.forward                   (hf_layers.py:161-213)
  DenseDropoutResidualLNFn  HF BertSelfOutput / BertOutput                 (call sites hf_layers.py:248,290)
  DenseGeluFn               HF BertIntermediate                            (call site hf_layers.py:289)
  FeedForwardFn             BertIntermediate + BertOutput fused            (hf_layers.py:289-290)
  TransformerLayerFn        BertLayerJit.forward, one autograd node        (hf_layers.py:255-292)
  LayerNormFn               nn.LayerNorm                                   (visual_bert.py:328)
  VisioLinguisticEmbeddingsFn  BertVisioLinguisticEmbeddings.forward       (embeddings.py:423-459)
  GatherRowsFn              torch.gather + Dropout of the `vqa` pooler     (visual_bert.py:389-400)
  LogitBCEFn                LogitBinaryCrossEntropy                        (losses.py:246-251)
  MaskedLMHeadFn            tied decoder + CrossEntropyLoss(ignore_index=-1) of VisualBERTForPretraining  (visual_bert.py:267-277)
  MaskedRegionHeadFn        image-prediction decoder + masked KLDivLoss of ViLBERTForPretraining  (vilbert.py:846-858, 1150-1157)
"""
import contextlib
import math
import os
import weakref

import torch

from mmf_amd import _native as nat
from mmf_amd import _ops_native

# With the native operator library loaded (mmf_amd/_ops_native.py) the dropout keys, the weight-shadow cache and the deferred LayerNorm
# reductions live in libmmf_amd_ops.so; the classes below then forward to it, so that the C++ autograd nodes and the Python ones
# (the models that are not ported to C++, the experiment hooks) share ONE state.
NATIVE = _ops_native.NATIVE

BF16 = torch.bfloat16
F32 = torch.float32


# ---------------------------------------------------------------------------------------------
# dropout keys
# ---------------------------------------------------------------------------------------------
class _DropoutKeys:
    """Per-site dropout keys.

    Eager mode: key = mix(seed, Philox offset) of the device's default torch generator, and the offset is advanced
    like any torch random op would: `torch.manual_seed(s)` reproduces the masks, `torch.cuda.get_rng_state()`
    checkpoints them; backward re-uses the key saved by forward.

    Graph mode (`with dropout_keys.graph_mode(seed_tensor)` around capture): site keys are a fixed sequence baked
    into the captured kernels and a device word (`seed_tensor`, int32[1]) is mixed in at run time;
    `nat.seed_advance(seed_tensor)` captured at the head of the step makes every replay draw fresh masks."""

    def __init__(self):
        self.seed_tensor = None
        self.counter = 0

    @staticmethod
    def _mix(a, b):
        x = (a * 0x9E3779B97F4A7C15 + (b + 1) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
        x ^= x >> 31
        return (x * 0x94D049BB133111EB >> 16) & 0xFFFFFFFF

    def next(self):
        if NATIVE:
            key, seed = torch.ops.mmf_amd._drop_next()
            return key, (seed[0] if seed else None)
        if self.seed_tensor is not None:
            self.counter += 1
            return self._mix(0x5EED, self.counter), self.seed_tensor
        gen = torch.cuda.default_generators[torch.cuda.current_device()]
        seed, off = gen.initial_seed(), gen.get_offset()
        gen.set_offset(off + 4)
        return self._mix(seed, off), None

    def graph_mode(self, seed_tensor):
        keys = self

        class _Ctx:
            def __enter__(self_):
                keys.seed_tensor, keys.counter = seed_tensor, 0
                if NATIVE:
                    torch.ops.mmf_amd._drop_graph_mode(seed_tensor)

            def __exit__(self_, *exc):
                keys.seed_tensor = None
                if NATIVE:
                    torch.ops.mmf_amd._drop_graph_mode(None)

        return _Ctx()


dropout_keys = _DropoutKeys()


def make_drop(p, training):
    if not training or p is None or p <= 0.0:
        return nat.NO_DROP
    key, seed = dropout_keys.next()
    return nat.drop_cfg(p, key, seed)


# ---------------------------------------------------------------------------------------------
# bf16 shadows of fp32 master parameters
# ---------------------------------------------------------------------------------------------
# Transposed weight twins for the input-gradient GEMMs (on; `DGRAD_NT` below).  dX = dY W reads W k-major; with
# a W^T twin both operands are row operands and the launch can use the forward-form kernels.  On the 128x128 kernel that bought
# nothing inside the step (round 1: 11.62 vs 11.53-11.62 ms), but it makes the narrow-output dgrads eligible for the 256x96 wide
# tile, which does pay (see _twin_pays).  Cost: one bf16 transpose per twin and optimizer step (mmf_transpose_bf16_multi).
DGRAD_NT = True      # (a module attribute, not an environment switch: tests/test_dgrad_nt_gpu.py flips it to compare with the k-major form)


def _twin_pays(out_width):
    """Input-gradient GEMMs whose OUTPUT is narrow (a multiple of 96 up to 1152 columns: 768 here) run on the 256x96 wide tile
    when both operands are row operands — 2.15 ms instead of 2.42 ms per step for the QKV / FFN-up / out-proj dgrads of VisualBERT
    VQA2 (same-box A/B, profiles/r02_bench_line.json) — so their weights keep a transposed twin.  Round 3: wide outputs (FFN-down's
    dgrad, 3072 columns, with the saved-gelu' multiply in its epilogue) take the 256x128 wide tile the same way: 52 us against 59-60
    us for the k-major W on the 128x128 kernel (tools/gemm_ab.py, profiles/r03_gemm_ab_ks.txt); the twin refresh is one transpose
    launch per optimizer step for all twins together."""
    return (out_width % 96 == 0 and out_width <= 1152) or out_width % 128 == 0


class ShadowCache:
    """bf16 copies of fp32 parameters, refreshed (one cast kernel) whenever the parameter's version
    counter or storage changes.  Several parameters can share one contiguous shadow (Q|K|V).

    What does NOT refresh a shadow: writes that bypass the version counter — `p.data.copy_(...)`, `p.data.mul_(...)` (EMA
    averaging, manual re-initialisation after the first forward) or a foreign kernel writing through `data_ptr()`.  After such
    an edit call `mmf_amd.functional.shadows.clear()` (every shadow is re-cast on its next use).  `load_state_dict`, torch
    optimizers and in-place ops on the parameter itself bump the version and are picked up automatically; the fused `adam_w`
    rewrites parameter and shadow in the same kernel."""

    def __init__(self):
        self._store = {}  # id(head parameter) -> (signature, buffer, dtype); entry dies with the parameter
        self._slot = {}   # id(parameter) -> (head key, first row) for bf16 shadows
        self._t = {}      # head key -> (signature at transposition, W^T buffer): transposed twins for the dgrad GEMMs
        self._by_ptr = {}  # data_ptr of a bf16 shadow buffer -> head key

    def clear(self):
        """Forget every shadow: the next use re-casts (used before hipGraph capture)."""
        if NATIVE:
            torch.ops.mmf_amd._shadow_set_twins(DGRAD_NT)      # (the module-level A/B switch is handed over at these sync points)
            torch.ops.mmf_amd._shadow_clear()
        self._store.clear()
        self._slot.clear()
        self._t.clear()
        self._by_ptr.clear()

    def transposed(self, w16):
        """W^T [in, out] (bf16) of the weight shadow `w16` [out, in], or None when `w16` is not a whole tracked shadow (or
        its dimensions are not multiples of 64, or DGRAD_NT is off).  Built on first use, re-built when the shadow's
        signature changed (a re-cast after `load_state_dict` / a torch optimizer), and kept current by
        `refresh_transposed()` when the fused optimizer updates parameters and shadows in place."""
        if NATIVE:
            torch.ops.mmf_amd._shadow_set_twins(DGRAD_NT)
            t = torch.ops.mmf_amd._shadow_transposed(w16)
            return t[0] if t else None
        if not DGRAD_NT or w16.dim() != 2:
            return None
        key = self._by_ptr.get(w16.data_ptr())
        ent = self._store.get(key) if key is not None else None
        if ent is None or ent[2] != BF16 or ent[1].data_ptr() != w16.data_ptr() or ent[1].shape != w16.shape:
            return None
        R, Cn = ent[1].shape
        if R % 64 or Cn % 64:
            return None
        t = self._t.get(key)
        if t is not None and t[0] == ent[0]:
            return t[1]
        tbuf = t[1] if (t is not None and tuple(t[1].shape) == (Cn, R)) else torch.empty(Cn, R, dtype=BF16, device=w16.device)
        nat.transpose_multi([(ent[1], tbuf)])
        self._t[key] = (ent[0], tbuf)
        return tbuf

    def refresh_transposed(self, only=None, skip=None):
        """Re-transpose every live twin from its (already updated) shadow: called by the fused optimizer's step.  `only` / `skip`:
        ids of the head parameters whose twins to refresh / leave alone (the optimizer-in-backward refreshes a layer's twins with
        that layer's update and the final step skips them).  With the native library: the head parameters themselves."""
        if NATIVE:
            torch.ops.mmf_amd._shadow_refresh_transposed(list(only) if only is not None else [], only is not None,
                                                         list(skip) if skip is not None else [], skip is not None)
            return
        only = None if only is None else {id(p) for p in only}
        skip = None if skip is None else {id(p) for p in skip}
        pairs = [(self._store[k][1], t[1]) for k, t in self._t.items() if k in self._store and self._store[k][0] == t[0]
                 and (only is None or k in only) and (skip is None or k not in skip)]
        if pairs:
            nat.transpose_multi(pairs)

    def slot(self, p):
        """The mirror rows of parameter `p` (bf16 weight shadow, or its slice of a packed fp32 Q|K|V bias) if it has an
        up-to-date one, else None.  The fused optimizer writes the new value there in the same pass that updates the
        fp32 master: no re-cast / re-pack next step, and the update stays visible to a replayed hipGraph."""
        if NATIVE:
            t = torch.ops.mmf_amd._shadow_slot(p)
            return t[0] if t else None
        ent = self._slot.get(id(p))
        if ent is None:
            return None
        key, r0 = ent
        st = self._store.get(key)
        if st is None:
            return None
        for q, ver, ptr in st[0]:
            if q == id(p) and (ver != p._version or ptr != p.data_ptr()):
                return None
        return st[1][r0:r0 + p.shape[0]]

    def get(self, *params, dtype=BF16):
        if NATIVE:
            return torch.ops.mmf_amd._shadow_get(list(params), dtype != BF16)
        head = params[0]
        key = id(head)
        ent = self._store.get(key)
        sig = tuple((id(p), p._version, p.data_ptr()) for p in params)
        if ent is not None and ent[0] == sig and ent[2] == dtype:
            return ent[1]
        if ent is None:
            weakref.finalize(head, self._store.pop, key, None)
            weakref.finalize(head, self._t.pop, key, None)
        rows = sum(p.shape[0] for p in params)
        shape = (rows,) + tuple(head.shape[1:])
        buf = ent[1] if ent is not None and ent[1].shape == shape and ent[2] == dtype else torch.empty(
            shape, dtype=dtype, device=head.device)
        r = 0
        for p in params:
            n = p.shape[0]
            src = p.detach()
            if not src.is_contiguous():
                src = src.contiguous()
            if dtype == BF16:
                nat.cast_f32_to_bf16(src, buf[r:r + n])
            else:
                buf[r:r + n].copy_(src)
            if dtype == BF16 or len(params) > 1:   # a lone fp32 "shadow" would just be the parameter itself
                self._slot[id(p)] = (key, r)
            r += n
        self._store[key] = (sig, buf, dtype)
        if dtype == BF16:
            self._by_ptr[buf.data_ptr()] = key
        return buf


shadows = ShadowCache()


def _as_bf16_2d(x):
    """Token-major bf16 view [rows, features] of an activation (casts fp32 inputs once)."""
    x2 = x.reshape(-1, x.shape[-1])
    if x2.dtype != BF16:
        if x2.dtype != F32:
            x2 = x2.float()
        out = torch.empty(x2.shape, dtype=BF16, device=x2.device)
        nat.cast_f32_to_bf16(x2.contiguous(), out)
        return out
    return x2 if x2.is_contiguous() else x2.contiguous()


def _pad8(n):
    return (n + 7) // 8 * 8


def _feature_rows(feats, rows, D):
    """Pre-extracted region / grid features [*, D] as the A operand of their projection GEMM.  fp32 features (what MMF's feature
    readers hand over) are cast to bf16 ONCE — one pass, fp32 read once, bf16 written once — so that the projection and, in backward,
    its weight gradient run on the LDS-DMA bf16 path; the fp32-operand GEMM converts in staging registers and ran VisualBERT's
    26 MB feature block at ~0.4 TB/s (66-72 us forward + 35 us weight gradient per step).  Same rounding either way."""
    f2 = feats.reshape(rows, D)
    if f2.dtype not in (F32, BF16):
        f2 = f2.float()
    f2 = f2.contiguous()
    if f2.dtype == F32 and _FEATS_CAST and D % 8 == 0 and not feats.requires_grad:
        f16 = torch.empty(rows, D, dtype=BF16, device=f2.device)
        nat.cast_f32_to_bf16(f2, f16)
        return f16
    return f2


def _grad_bf16(g, cols):
    """bf16, contiguous, ld == cols (cols % 8 == 0) version of an incoming gradient."""
    g2 = g.reshape(-1, g.shape[-1])
    if g2.dtype == BF16 and g2.is_contiguous():
        return g2
    return _as_bf16_2d(g2)


# ---------------------------------------------------------------------------------------------
# shared forward/backward pieces
# ---------------------------------------------------------------------------------------------
_FUSED_DB = True     # the bias gradient rides on the weight-gradient GEMM (tests switch it off to compare)
_FEATS_CAST = True   # region features cast to bf16 once (see VisioLinguisticEmbeddingsFn)


_WGRAD_DEFER_MIN_ROWS = 64      # token rows below which a weight gradient is not worth queueing (the heads: a handful of rows, their own skinny paths)


def _linear_bwd(dy, ldy, x, w16, M, N, K, need_dx=True, dx_resid=None, act_aux=None, want_db=False, defer=False):
    """dy [M,N] bf16 (row stride ldy, pad columns zero), x [M,K] bf16, w16 [N,K].
    Returns (dx [M,K] bf16 or None, dW [N,K] fp32) and, with `want_db`, the bias gradient [N] fp32 as a third value —
    carried by the weight-gradient GEMM itself when that runs split-K (one extra MFMA per A fragment against a ones
    operand, no separate pass over dy), else by the column-sum kernels.
    `defer`: the caller's weight is an encoder-layer-internal matrix that receives exactly ONE gradient contribution per backward, so
    inside `wgrad_defer()` its dW / db may be returned unfilled and written by a later grouped launch.  Nodes whose weight can be
    shared with another node (a decoder tied to an embedding table, M4C's classifier that is also PrevPredEmbeddings' lookup table,
    plain `LinearFn`) never pass it: autograd sums the contributions of a shared parameter as soon as the second one arrives — before any
    flush — and would add an unfilled buffer."""
    dev = dy.device
    dx = None
    if need_dx:
        dx = torch.empty(M, K, dtype=BF16, device=dev)
        wt = shadows.transposed(w16) if (N % 8 == 0 and _twin_pays(K)) else None
        if wt is not None:     # dX = dY (W^T)^T with two row operands (see DGRAD_NT)
            nat.gemm(dy, wt, dx, M, K, N, ldy, N, K, resid=dx_resid, ldr=K, act=2 if act_aux is not None else 0, aux=act_aux)
        else:
            nat.gemm(dy, w16, dx, M, K, N, ldy, K, K, b_kmajor=True, resid=dx_resid, ldr=K,
                     act=2 if act_aux is not None else 0, aux=act_aux)
    if (defer and wgrad_defer.active and x.dtype == BF16 and dy.dtype == BF16 and x.stride(1) == 1 and M >= _WGRAD_DEFER_MIN_ROWS
            and wgrad_defer.first_use(w16)):
        # the weight gradient (and the bias gradient riding on it) joins a grouped launch that runs when eight problems are waiting or the
        # deferral block ends: dW / db are returned now and FILLED then (see _WgradDefer)
        prob, dw, db = _wgrad_problem(dy, ldy, x, M, N, K, want_db)
        wgrad_defer.push(prob, (dy, x, dw, db))
        return (dx, dw, db) if want_db else (dx, dw)
    dw = torch.empty(N, K, dtype=F32, device=dev)
    fused = want_db and x.dtype == BF16 and _FUSED_DB and nat.gemm_rowsum_supported(N, K, M)
    db = torch.empty(N, dtype=F32, device=dev) if fused else None
    nat.gemm(dy, x, dw, N, K, M, ldy, x.stride(0), K, a_kmajor=True, b_kmajor=True, rowsum_out=db)
    if not want_db:
        return dx, dw
    if db is None:
        db = _colsum(dy, ldy, M, N)
    return dx, dw, db


def _colsum(x, ld, rows, N):
    out = torch.empty(N, dtype=F32, device=x.device)
    ws = torch.empty(nat.colsum_ws_floats(N), dtype=F32, device=x.device)
    nat.colsum(x, ld, 1, rows, 0, N, out, 0.0, ws)
    return out


# ---------------------------------------------------------------------------------------------
# Linear
# ---------------------------------------------------------------------------------------------
class LinearFn(torch.autograd.Function):
    """y = x W^T + b.  x [*, K] (bf16 or fp32), weight [N, K] fp32 master + bf16 shadow, y bf16 or fp32."""

    @staticmethod
    def forward(ctx, x, weight, bias, w16, out_f32):
        if x.dtype == F32 and not x.requires_grad:
            x2 = x.reshape(-1, x.shape[-1]).contiguous()   # raw fp32 features: converted while staged by the GEMM
        else:
            x2 = _as_bf16_2d(x)
        M, K = x2.shape
        N = weight.shape[0]
        y = torch.empty(M, N, dtype=F32 if out_f32 else BF16, device=x2.device)
        nat.gemm(x2, w16, y, M, N, K, K, K, N, bias=bias.detach() if bias is not None else None)
        ctx.save_for_backward(x2, w16)
        ctx.dims = (M, N, K, x.shape, bias is not None)
        return y.view(*x.shape[:-1], N)

    @staticmethod
    def backward(ctx, gy):
        x2, w16 = ctx.saved_tensors
        M, N, K, xshape, has_bias = ctx.dims
        ldy = _pad8(N)
        g2 = gy.reshape(M, N)
        if g2.dtype == BF16 and N == ldy and g2.is_contiguous():
            dy = g2
        else:
            dy = torch.empty(M, ldy, dtype=BF16, device=gy.device)
            if g2.dtype == BF16:
                g2 = g2.float()
            nat.cast2d_f32_to_bf16(g2.contiguous(), N, dy, ldy, M, N)
        if has_bias:
            dx, dw, db = _linear_bwd(dy, ldy, x2, w16, M, N, K, need_dx=ctx.needs_input_grad[0], want_db=True)
        else:
            (dx, dw), db = _linear_bwd(dy, ldy, x2, w16, M, N, K, need_dx=ctx.needs_input_grad[0]), None
        return (dx.view(xshape) if dx is not None else None), dw, db, None, None


def linear(x, weight, bias, out_f32=False):
    return LinearFn.apply(x, weight, bias, shadows.get(weight), out_f32)


# ---------------------------------------------------------------------------------------------
# self attention (QKV projection + fused attention)
# ---------------------------------------------------------------------------------------------
# Keep an fp32 copy of the attention output for the backward's delta = rowsum(dO o O): with the bf16 O the rows of dS sum
# to ~2^-9 |dO||O| instead of 0 and the common component of K / Q (their biases) leaks into dQ / dK.  Costs one extra
# [tokens, H] fp32 write in forward and read in backward per layer (~0.7 % of a VisualBERT step).
EXACT_ATTENTION_DELTA = True


def _o32(M, H, dev, need):
    return torch.empty(M, H, dtype=F32, device=dev) if (need and EXACT_ATTENTION_DELTA) else None


def _keep_bits(B, heads, Sq, Sk, head_dim, drop, dev, need):
    """The table through which the attention forward hands its dropout decisions to the backward (mmf_attn_desc.keep_bits) where the backward is the
    one-pass kernel — head_dim 64 up to 256 positions, head_dim 128 up to 128: every model of the path at BASELINE's shapes — else None (both directions
    hash)."""
    if not (need and drop[1]):
        return None
    words = nat.attention_keep_bits_words(B, heads, Sq, Sk, head_dim)
    return torch.empty(words, dtype=torch.int32, device=dev) if words else None


class PrefixLMMask:
    """What M4C's multimodal transformer hands its encoder as a [B, 1, L, L] tensor (MMT.forward, mmf/models/m4c.py:424-440),
    in the form the fused attention kernel takes it: the additive key mask [B, L] fp32 plus the number of trailing
    decoding positions that see each other causally (`mmf_attn_desc.causal_tail`).  Pass it wherever an encoder accepts
    `attention_mask`."""
    __slots__ = ("key_mask", "causal_tail")

    def __init__(self, key_mask, causal_tail):
        self.key_mask, self.causal_tail = key_mask, int(causal_tail)


def _split_mask(mask):
    if isinstance(mask, PrefixLMMask):
        return mask.key_mask, mask.causal_tail
    return mask, 0


def _attn_fwd(x2, wqkv16, bqkv, mask_add, B, S, heads, drop, need_bwd=True, tail=0, qk_gate=None, site=0):
    M, H = x2.shape
    dev = x2.device
    qkv = torch.empty(M, 3 * H, dtype=BF16, device=dev)
    nat.gemm(x2, wqkv16, qkv, M, 3 * H, H, H, H, 3 * H, bias=bqkv, debug_flags=nat.gemm_site(site))
    if qk_gate is not None:       # ViLBERT dynamic_attention: per-sample column gates on the Q|K columns (vilbert.py:211-212)
        nat.rowgroup_scale(qkv, 3 * H, qk_gate, B, S, 2 * H)
    ctxt = torch.empty(M, H, dtype=BF16, device=dev)
    lse = torch.empty(B, heads, S, dtype=F32, device=dev)
    scale = 1.0 / math.sqrt(H // heads)
    o32 = _o32(M, H, dev, need_bwd)
    kb = _keep_bits(B, heads, S, S, H // heads, drop, dev, need_bwd)
    nat.attention_fwd(qkv, qkv[:, H:], qkv[:, 2 * H:], 3 * H, 3 * H, 3 * H, mask_add, ctxt, H, lse, B, heads, S, S, scale, drop,
                      head_dim=H // heads, ctx_f32=o32, causal_tail=tail, keep_bits=kb)
    return qkv, ctxt, lse, o32, kb


def _attn_bwd(dctx, x2, qkv, ctxt, lse, wqkv16, mask_add, B, S, heads, drop, dx_resid=None, need_dx=True, o32=None, tail=0,
              qk_gate=None, kb=None):
    """Returns (dx, dW_qkv, db_qkv) — and the gradient of `qk_gate` as a fourth element when a gate was applied."""
    M, H = x2.shape
    dev = x2.device
    dqkv = torch.empty(M, 3 * H, dtype=BF16, device=dev)
    delta = torch.empty(B, heads, S, dtype=F32, device=dev)
    scale = 1.0 / math.sqrt(H // heads)
    nat.attention_bwd(qkv, qkv[:, H:], qkv[:, 2 * H:], 3 * H, 3 * H, 3 * H, mask_add, ctxt, H, lse, B, heads, S, S, scale,
                      dctx, dqkv, dqkv[:, H:], dqkv[:, 2 * H:], delta, drop, head_dim=H // heads, ctx_f32=o32, causal_tail=tail, keep_bits=kb)
    if qk_gate is None:
        return _linear_bwd(dqkv, 3 * H, x2, wqkv16, M, 3 * H, H, need_dx=need_dx, dx_resid=dx_resid, want_db=True, defer=True)
    dgate = torch.empty_like(qk_gate)
    nat.rowgroup_scale_bwd(dqkv, qkv, 3 * H, qk_gate, dgate, B, S, 2 * H)      # dqkv becomes the gradient of the un-gated projection
    return _linear_bwd(dqkv, 3 * H, x2, wqkv16, M, 3 * H, H, need_dx=need_dx, dx_resid=dx_resid, want_db=True, defer=True) + (dgate,)


class SelfAttentionFn(torch.autograd.Function):
    """BertSelfAttentionJit.forward: returns the context layer [B, S, H] (bf16)."""

    @staticmethod
    def forward(ctx, x, wq, bq, wk, bk, wv, bv, wqkv16, bqkv, mask_add, heads, drop):
        B, S, H = x.shape
        x2 = _as_bf16_2d(x)
        mask_add, tail = _split_mask(mask_add)
        qkv, ctxt, lse, o32, kb = _attn_fwd(x2, wqkv16, bqkv, mask_add, B, S, heads, drop, any(ctx.needs_input_grad), tail)
        ctx.save_for_backward(x2, qkv, ctxt, lse, wqkv16, mask_add, o32, kb)
        ctx.meta = (B, S, H, heads, drop, tail)
        return ctxt.view(B, S, H)

    @staticmethod
    def backward(ctx, g):
        x2, qkv, ctxt, lse, wqkv16, mask_add, o32, kb = ctx.saved_tensors
        B, S, H, heads, drop, tail = ctx.meta
        dx, dw, db = _attn_bwd(_grad_bf16(g, H), x2, qkv, ctxt, lse, wqkv16, mask_add, B, S, heads, drop,
                               need_dx=ctx.needs_input_grad[0], o32=o32, tail=tail, kb=kb)
        return ((dx.view(B, S, H) if dx is not None else None), dw[:H], db[:H], dw[H:2 * H], db[H:2 * H], dw[2 * H:], db[2 * H:],
                None, None, None, None, None)


# ---------------------------------------------------------------------------------------------
# dense -> dropout -> (+ residual) -> LayerNorm      (BertSelfOutput / BertOutput)
# ---------------------------------------------------------------------------------------------
def _ddrln_fwd(h2, resid2, w16, bias, gamma, beta, eps, drop, site=0):
    M, K = h2.shape
    N = w16.shape[0]
    dev = h2.device
    y = torch.empty(M, N, dtype=BF16, device=dev)
    nat.gemm(h2, w16, y, M, N, K, K, K, N, bias=bias, resid=resid2, ldr=N, drop=drop, debug_flags=nat.gemm_site(site))
    out = torch.empty(M, N, dtype=BF16, device=dev)
    mean = torch.empty(M, dtype=F32, device=dev)
    rstd = torch.empty(M, dtype=F32, device=dev)
    nat.layernorm_fwd(y, gamma, beta, out, mean, rstd, M, N, eps)
    return out, y, mean, rstd


class _LnDefer:
    """Opt-in: inside `with ln_defer():` LayerNorm backwards leave the column sums behind their dgamma / dbeta as per-workgroup
    partials and ONE multi-tensor launch finishes all of them when the block ends (26 reductions per VisualBERT step become one).
    The returned dgamma / dbeta tensors are NOT valid before that: only for callers that own the whole backward and consume the
    parameter gradients after it (the graphed steps in mmf_amd/utils/graph.py)."""

    def __init__(self):
        self.active = False
        self.pending = []

    @contextlib.contextmanager
    def __call__(self):
        old, self.active = self.active, True
        if NATIVE:
            torch.ops.mmf_amd._ln_defer_set(True)
        try:
            yield
        finally:
            self.active = old
            if NATIVE:
                torch.ops.mmf_amd._ln_defer_set(old)
            if not old:
                self.flush()

    def flush(self):
        if NATIVE:
            torch.ops.mmf_amd._ln_defer_flush()       # the reductions the C++ autograd nodes deferred
        if self.pending:
            items, self.pending = self.pending, []
            nat.layernorm_bwd_reduce_multi(items)


ln_defer = _LnDefer()


def _ln_bwd(dy, y, mean, rstd, gamma, drop, want_dbias):
    M, N = y.shape
    dev = y.device
    dx = torch.empty(M, N, dtype=BF16, device=dev)
    dlin = torch.empty(M, N, dtype=BF16, device=dev) if drop[1] else None
    dgamma = torch.empty(N, dtype=F32, device=dev)
    dbeta = torch.empty(N, dtype=F32, device=dev)
    dbias = torch.empty(N, dtype=F32, device=dev) if want_dbias else None
    ws = torch.empty(nat.layernorm_bwd_ws_floats(N), dtype=F32, device=dev)
    if ln_defer.active and not want_dbias and nat.layernorm_bwd_deferrable(M, N):
        nat.layernorm_bwd(dy, y, mean, rstd, gamma, dx, dlin, drop, None, None, None, 0, ws, M, N)
        ln_defer.pending.append((ws, M, N, dgamma, dbeta))
    else:
        nat.layernorm_bwd(dy, y, mean, rstd, gamma, dx, dlin, drop, dgamma, dbeta, dbias, 0, ws, M, N)
    return dx, (dlin if dlin is not None else dx), dgamma, dbeta, dbias


def _ln_bwd_din(dy, y, mean, rstd, gamma, in_drop):
    """Backward of dropout(LayerNorm(y)) with the dropout backward folded into the load of dy (embeddings.py:343-345); returns (dx, dgamma, dbeta)."""
    M, N = y.shape
    dev = y.device
    dx = torch.empty(M, N, dtype=BF16, device=dev)
    dgamma = torch.empty(N, dtype=F32, device=dev)
    dbeta = torch.empty(N, dtype=F32, device=dev)
    ws = torch.empty(nat.layernorm_bwd_ws_floats(N), dtype=F32, device=dev)
    if ln_defer.active and nat.layernorm_bwd_deferrable(M, N):
        nat.layernorm_bwd_din(dy, y, mean, rstd, gamma, dx, in_drop, None, None, 0, ws, M, N)
        ln_defer.pending.append((ws, M, N, dgamma, dbeta))
    else:
        nat.layernorm_bwd_din(dy, y, mean, rstd, gamma, dx, in_drop, dgamma, dbeta, 0, ws, M, N)
    return dx, dgamma, dbeta


class DenseDropoutResidualLNFn(torch.autograd.Function):
    """LayerNorm(dropout(h W^T + b) + residual)."""

    @staticmethod
    def forward(ctx, h, resid, weight, bias, gamma, beta, w16, eps, drop):
        h2 = _as_bf16_2d(h)
        r2 = _as_bf16_2d(resid)
        out, y, mean, rstd = _ddrln_fwd(h2, r2, w16, bias.detach(), gamma.detach(), beta.detach(), eps, drop)
        ctx.save_for_backward(h2, y, mean, rstd, w16, gamma.detach())
        ctx.meta = (h.shape, resid.shape, drop)
        return out.view(resid.shape)

    @staticmethod
    def backward(ctx, g):
        h2, y, mean, rstd, w16, gamma = ctx.saved_tensors
        hshape, rshape, drop = ctx.meta
        M, N = y.shape
        K = h2.shape[1]
        # (the bias gradient of the output projection is a column sum of dlin, the A operand of its weight-gradient GEMM: it rides on that GEMM —
        # the fused layer node does the same — so the LayerNorm backward carries no third column sum and its reduction can be deferred: ln_defer)
        dx, dlin, dgamma, dbeta, _ = _ln_bwd(_grad_bf16(g, N), y, mean, rstd, gamma, drop, False)
        dh, dw, dbias = _linear_bwd(dlin if dlin is not None else dx, N, h2, w16, M, N, K, need_dx=ctx.needs_input_grad[0], want_db=True, defer=True)
        return ((dh.view(hshape) if dh is not None else None), dx.view(rshape), dw, dbias, dgamma, dbeta, None, None, None)


# ---------------------------------------------------------------------------------------------
# dense -> GELU      (BertIntermediate)
# ---------------------------------------------------------------------------------------------
class DenseGeluFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, w16):
        x2 = _as_bf16_2d(x)
        M, K = x2.shape
        N = w16.shape[0]
        u = torch.empty(M, N, dtype=BF16, device=x2.device)
        hh = torch.empty(M, N, dtype=BF16, device=x2.device)
        nat.gemm(x2, w16, hh, M, N, K, K, K, N, bias=bias.detach(), act=1, U=u)
        ctx.save_for_backward(x2, u, w16)
        ctx.xshape = x.shape
        return hh.view(*x.shape[:-1], N)

    @staticmethod
    def backward(ctx, g):
        x2, u, w16 = ctx.saved_tensors
        M, K = x2.shape
        N = w16.shape[0]
        du = torch.empty(M, N, dtype=BF16, device=x2.device)
        nat.gelu_bwd(_grad_bf16(g, N), u, du)
        dx, dw, db = _linear_bwd(du, N, x2, w16, M, N, K, need_dx=ctx.needs_input_grad[0], want_db=True)
        return (dx.view(ctx.xshape) if dx is not None else None), dw, db, None


# ---------------------------------------------------------------------------------------------
# fused transformer sub-blocks (what BertLayerJit.forward actually runs)
# ---------------------------------------------------------------------------------------------
class AttentionBlockFn(torch.autograd.Function):
    """BertAttentionJit.forward (hf_layers.py:233-252): self-attention + BertSelfOutput, with the
    residual-gradient add fused into the QKV dgrad epilogue."""

    @staticmethod
    def forward(ctx, x, wq, bq, wk, bk, wv, bv, wo, bo, gamma, beta, wqkv16, bqkv, wo16, mask_add, heads, eps, drop_attn, drop_hid,
                qk_gate=None):
        """`qk_gate` (fp32 [B, 2H] or None): ViLBERT's dynamic_attention gates on the queries and keys (vilbert.py:199-212)."""
        B, S, H = x.shape
        x2 = _as_bf16_2d(x)
        mask_add, tail = _split_mask(mask_add)
        gate = None if qk_gate is None else qk_gate.detach().float().contiguous()
        qkv, ctxt, lse, o32, kb = _attn_fwd(x2, wqkv16, bqkv, mask_add, B, S, heads, drop_attn, any(ctx.needs_input_grad), tail, gate)
        out, y, mean, rstd = _ddrln_fwd(ctxt, x2, wo16, bo.detach(), gamma.detach(), beta.detach(), eps, drop_hid)
        ctx.save_for_backward(x2, qkv, ctxt, lse, y, mean, rstd, wqkv16, wo16, gamma.detach(), mask_add, o32, gate, kb)
        ctx.meta = (B, S, H, heads, drop_attn, drop_hid, tail)
        return out.view(B, S, H)

    @staticmethod
    def backward(ctx, g):
        x2, qkv, ctxt, lse, y, mean, rstd, wqkv16, wo16, gamma, mask_add, o32, gate, kb = ctx.saved_tensors
        B, S, H, heads, drop_attn, drop_hid, tail = ctx.meta
        M = B * S
        # (the bias gradient of the output projection is a column sum of dlin, the A operand of its weight-gradient GEMM: it rides on that GEMM —
        # the fused layer node does the same — so the LayerNorm backward carries no third column sum and its reduction can be deferred: ln_defer)
        dres, dlin, dgamma, dbeta, _ = _ln_bwd(_grad_bf16(g, H), y, mean, rstd, gamma, drop_hid, False)
        dctx, dwo, dbo = _linear_bwd(dlin if dlin is not None else dres, H, ctxt, wo16, M, H, H, want_db=True, defer=True)
        res = _attn_bwd(dctx, x2, qkv, ctxt, lse, wqkv16, mask_add, B, S, heads, drop_attn, dx_resid=dres, o32=o32, tail=tail, kb=kb,
                        qk_gate=gate)
        dx, dwqkv, dbqkv = res[:3]
        return (dx.view(B, S, H), dwqkv[:H], dbqkv[:H], dwqkv[H:2 * H], dbqkv[H:2 * H], dwqkv[2 * H:], dbqkv[2 * H:],
                dwo, dbo, dgamma, dbeta, None, None, None, None, None, None, None, None, res[3] if gate is not None else None)


class MaskedMeanFn(torch.autograd.Function):
    """(x * mask.unsqueeze(-1)).sum(1) / mask.sum(1, keepdim=True): the text pooling of ViLBERT's dynamic_attention
    (vilbert.py:204-205).  x [B, T, H], mask [B, T] (0 / 1) -> fp32 [B, H]."""

    @staticmethod
    def forward(ctx, x, mask):
        B, T, H = x.shape
        x3 = x.detach()
        x3 = (x3 if x3.dtype == BF16 else x3.to(BF16)).contiguous()
        m = mask.detach().reshape(B, T).float().contiguous()
        pool = torch.empty(B, H, dtype=F32, device=x.device)
        nat.masked_mean_fwd(x3, m, pool, B, T, H)
        ctx.save_for_backward(m)
        ctx.meta = (B, T, H, x.dtype)
        return pool

    @staticmethod
    def backward(ctx, g):
        m, = ctx.saved_tensors
        B, T, H, dtype = ctx.meta
        dx = torch.empty(B, T, H, dtype=BF16, device=g.device)
        nat.masked_mean_bwd(g.float().contiguous(), m, dx, B, T, H)
        return (dx if dtype == BF16 else dx.to(dtype)), None


class DynamicGateFn(torch.autograd.Function):
    """ViLBERT `dynamic_attention` gates (mmf/models/vilbert.py:206-209): cat([1 + sigmoid(zq), 1 + sigmoid(zk)], dim=1) as ONE fp32
    [B, 2 C] tensor — the two halves are written in place by the sigmoid kernel (no ATen sigmoid / add / cat), backward dz = dgate s (1 - s)."""

    @staticmethod
    def forward(ctx, zq, zk):
        B, Cn = zq.shape
        q = zq.float().contiguous(); k = zk.float().contiguous()
        gate = torch.empty(B, 2 * Cn, dtype=F32, device=zq.device)
        nat.gate_sigmoid_fwd(q, gate, 0, B, Cn)
        nat.gate_sigmoid_fwd(k, gate, Cn, B, Cn)
        ctx.save_for_backward(gate)
        ctx.meta = (B, Cn, zq.dtype, zk.dtype)
        return gate

    @staticmethod
    def backward(ctx, g):
        (gate,) = ctx.saved_tensors
        B, Cn, dq_t, dk_t = ctx.meta
        g = g.float().contiguous()
        dzq = torch.empty(B, Cn, dtype=F32, device=g.device); dzk = torch.empty(B, Cn, dtype=F32, device=g.device)
        nat.gate_sigmoid_bwd(g, gate, 0, dzq, B, Cn)
        nat.gate_sigmoid_bwd(g, gate, Cn, dzk, B, Cn)
        return dzq.to(dq_t), dzk.to(dk_t)


class FeedForwardFn(torch.autograd.Function):
    """BertIntermediate + BertOutput (hf_layers.py:289-290): GELU in the up-projection epilogue,
    dropout + residual in the down-projection epilogue, GELU' in the down-projection dgrad epilogue,
    residual-gradient add in the up-projection dgrad epilogue."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, gamma, beta, w1_16, w2_16, eps, drop_hid):
        shape = x.shape
        x2 = _as_bf16_2d(x)
        M, H = x2.shape
        I = w1_16.shape[0]
        dev = x2.device
        u = torch.empty(M, I, dtype=BF16, device=dev)
        hh = torch.empty(M, I, dtype=BF16, device=dev)
        nat.gemm(x2, w1_16, hh, M, I, H, H, H, I, bias=b1.detach(), act=1, U=u)
        out, y, mean, rstd = _ddrln_fwd(hh, x2, w2_16, b2.detach(), gamma.detach(), beta.detach(), eps, drop_hid)
        ctx.save_for_backward(x2, u, hh, y, mean, rstd, w1_16, w2_16, gamma.detach())
        ctx.meta = (shape, drop_hid)
        return out.view(shape)

    @staticmethod
    def backward(ctx, g):
        x2, u, hh, y, mean, rstd, w1_16, w2_16, gamma = ctx.saved_tensors
        shape, drop_hid = ctx.meta
        M, H = x2.shape
        I = w1_16.shape[0]
        # (the bias gradient of the output projection is a column sum of dlin, the A operand of its weight-gradient GEMM: it rides on that GEMM —
        # the fused layer node does the same — so the LayerNorm backward carries no third column sum and its reduction can be deferred: ln_defer)
        dres, dlin, dgamma, dbeta, _ = _ln_bwd(_grad_bf16(g, H), y, mean, rstd, gamma, drop_hid, False)
        du, dw2, db2 = _linear_bwd(dlin if dlin is not None else dres, H, hh, w2_16, M, H, I, act_aux=u, want_db=True, defer=True)        # du = (dlin W2) * gelu'(u), gelu' saved by the forward
        dx, dw1, db1 = _linear_bwd(du, I, x2, w1_16, M, I, H, dx_resid=dres, want_db=True, defer=True)     # dx = du W1 + dres
        return dx.view(shape), dw1, db1, dw2, db2, dgamma, dbeta, None, None, None, None


def _dgrad(dy, ldy, w16, M, N, K, dx_resid=None, act_aux=None, site=0):
    """dX [M, K] = dY [M, N] W [N, K] (W k-major: no transposed copy), residual-gradient add / saved-GELU' multiply fused.
    `site`: the call's MMF_SITE_* tag (include/mmf_amd.h: names the call for the per-site store policy, like the native layer node's)."""
    dx = torch.empty(M, K, dtype=BF16, device=dy.device)
    wt = shadows.transposed(w16) if (N % 8 == 0 and _twin_pays(K)) else None
    if wt is not None:      # two row operands (W^T twin): the forward-form kernels, i.e. the 256x96 wide tile for these widths
        nat.gemm(dy, wt, dx, M, K, N, ldy, N, K, resid=dx_resid, ldr=K, act=2 if act_aux is not None else 0, aux=act_aux, debug_flags=nat.gemm_site(site))
    else:
        nat.gemm(dy, w16, dx, M, K, N, ldy, K, K, b_kmajor=True, resid=dx_resid, ldr=K, act=2 if act_aux is not None else 0, aux=act_aux,
                 debug_flags=nat.gemm_site(site))
    return dx


def _wgrad_problem(dy, ldy, x, M, N, K, want_db):
    """Deferred weight gradient dW [N, K] = dY^T X (+ bias gradient = column sums of dY) as a `nat.gemm_grouped` problem."""
    dw = torch.empty(N, K, dtype=F32, device=dy.device)
    db = torch.empty(N, dtype=F32, device=dy.device) if want_db else None
    prob = dict(A=dy, B=x, C_out=dw, M=N, N=K, K=M, lda=ldy, ldb=x.stride(0), ldc=K, a_kmajor=True, b_kmajor=True, rowsum_out=db)
    return prob, dw, db


class _WgradDefer:
    """Opt-in: inside `with wgrad_defer():` the weight gradients of the encoder-internal autograd nodes that are NOT the fused encoder layer
    (ViLBERT's connection layers: bi-attention, two output blocks, two feed-forward blocks — the nodes that pass `defer=True` to `_linear_bwd`;
    never a node whose weight may be tied to another node's: `LinearFn`, the vocabulary / classifier heads) are not launched one by one — each a split-K GEMM over a few
    dozen tiles plus the kernel that sums its slabs — but queued and launched eight at a time as ONE grouped GEMM (`nat.gemm_grouped`, the
    launch the fused layer node uses for its own four): every tile reduces over all token rows itself, no slabs, and the launch fills the chip.
    Problems that are whole 256 x 128 tiles with a 64-aligned token count go to the wide-tile queue, the others to the 128-row one.  The
    returned dW / db tensors are NOT valid before the flush: only for callers that own the whole backward and read the parameter gradients
    after it (the graphed steps in mmf_amd/utils/graph.py), like `ln_defer`."""

    def __init__(self):
        self.active = False
        self.queues = {}      # (wide-tile eligible, stream) -> list of (problem, tensors kept alive)
        self.seen = set()     # weights (their bf16 shadows) that already have a queued gradient in this deferral block
        self.streams = {}     # stream handle -> torch stream object of the queues above (a mid-backward flush orders itself behind them)
        self.enabled = True

    @contextlib.contextmanager
    def __call__(self):
        old, self.active = self.active, self.enabled
        try:
            yield
        finally:
            self.active = old
            if not old:
                self.flush()

    @staticmethod
    def _wide(prob):
        return prob["M"] % 256 == 0 and prob["N"] % 128 == 0 and prob["K"] % 64 == 0 and prob["K"] >= 192

    def first_use(self, w16):
        """True when `w16`'s weight has no gradient queued yet.  A weight that comes a second time in one backward (a layer applied twice:
        tied layers) must not be queued again — autograd sums the two contributions the moment the second node returns — so everything
        queued is launched now (the first contribution is filled, in stream order, before that sum) and the caller computes this one
        immediately."""
        key = w16.data_ptr()
        if key in self.seen:
            self.flush(mid_backward=True)
            self.seen.add(key)      # (a third application is immediate as well)
            return False
        self.seen.add(key)
        return True

    def push(self, prob, keep):
        # One queue per stream: a full queue is launched on the stream its problems were produced on (ViLBERT runs its visual stream on a
        # side HIP stream; autograd replays a node on its forward stream).  What is left at the end is launched by `flush` on the caller's
        # stream, after the autograd engine has joined every stream the backward pass used.
        key = (self._wide(prob), torch.cuda.current_stream().cuda_stream if torch.cuda.is_available() else 0)      # (no GPU: the host-logic dry runs)
        if torch.cuda.is_available():
            self.streams[key[1]] = torch.cuda.current_stream()
        q = self.queues.setdefault(key, [])
        q.append((prob, keep))
        if len(q) == nat.GEMM_GROUP_MAX:
            self._launch(q)

    def _launch(self, q):
        if q:
            nat.gemm_grouped([p for p, _ in q])
            del q[:]

    def flush(self, mid_backward=False):
        """`mid_backward` (first_use: a tied weight met in the middle of backward): the other streams are still running, so the caller's stream first
        waits for the stream that produced a foreign queue's dy / x - at the end of backward the autograd engine has already joined them."""
        cur = torch.cuda.current_stream().cuda_stream if torch.cuda.is_available() else 0
        for (_, stream), q in self.queues.items():
            if q and stream != cur and mid_backward and stream in self.streams:
                torch.cuda.current_stream().wait_stream(self.streams[stream])
            if q and stream != cur and not torch.cuda.is_current_stream_capturing():
                # leftovers of another stream's queue (ViLBERT's visual stream) run on the caller's stream: their buffers belong to that
                # stream's pool of the caching allocator, which must not hand them out again while this launch / the optimizer that reads
                # dW, db on the caller's stream is pending
                for _, keep in q:
                    for t in keep:
                        if t is not None:
                            t.record_stream(torch.cuda.current_stream())
            self._launch(q)
        self.queues.clear()
        self.streams.clear()
        self.seen.clear()


wgrad_defer = _WgradDefer()


class TransformerLayerFn(torch.autograd.Function):
    """BertLayerJit.forward (hf_layers.py:255-292) as ONE autograd node: the attention sub-layer (AttentionBlockFn) followed by
    the feed-forward sub-layer (FeedForwardFn), same kernels and same saved tensors.  What the fusion buys is in backward:
    the four weight gradients of the layer (dW_qkv, dW_o, dW_1, dW_2, with the two bias gradients that are column sums of
    GEMM operands) leave the dgrad chain and run at its end as ONE grouped launch (`nat.gemm_grouped`): 432 output tiles at
    the VisualBERT VQA2 shape = one round of the chip's 512 workgroup slots, each tile reducing over all 7296 tokens itself —
    no split-K slabs, no slab-reduction kernels."""

    @staticmethod
    def forward(ctx, x, wq, bq, wk, bk, wv, bv, wo, bo, g1, be1, w1, b1, w2, b2, g2, be2,
                wqkv16, bqkv, wo16, w1_16, w2_16, mask_add, heads, eps1, eps2, drop_attn, drop_hid1, drop_hid2):
        B, S, H = x.shape
        x2 = _as_bf16_2d(x)
        M = B * S
        I = w1_16.shape[0]
        dev = x2.device
        mask_add, tail = _split_mask(mask_add)
        qkv, ctxt, lse, o32, kb = _attn_fwd(x2, wqkv16, bqkv, mask_add, B, S, heads, drop_attn, any(ctx.needs_input_grad), tail, site=nat.SITE_QKV_FWD)
        a_out, y1, mean1, rstd1 = _ddrln_fwd(ctxt, x2, wo16, bo.detach(), g1.detach(), be1.detach(), eps1, drop_hid1, site=nat.SITE_ATTN_OUT_FWD)
        u = torch.empty(M, I, dtype=BF16, device=dev)
        hh = torch.empty(M, I, dtype=BF16, device=dev)
        nat.gemm(a_out, w1_16, hh, M, I, H, H, H, I, bias=b1.detach(), act=1, U=u, debug_flags=nat.gemm_site(nat.SITE_FFN_UP_FWD))
        out, y2, mean2, rstd2 = _ddrln_fwd(hh, a_out, w2_16, b2.detach(), g2.detach(), be2.detach(), eps2, drop_hid2, site=nat.SITE_FFN_DOWN_FWD)
        ctx.save_for_backward(x2, qkv, ctxt, lse, y1, mean1, rstd1, a_out, u, hh, y2, mean2, rstd2, wqkv16, wo16, w1_16, w2_16,
                              g1.detach(), g2.detach(), mask_add, o32, kb)
        ctx.meta = (B, S, H, I, heads, drop_attn, drop_hid1, drop_hid2, tail)
        return out.view(B, S, H)

    @staticmethod
    def backward(ctx, g):
        (x2, qkv, ctxt, lse, y1, mean1, rstd1, a_out, u, hh, y2, mean2, rstd2, wqkv16, wo16, w1_16, w2_16, g1, g2, mask_add,
         o32, kb) = ctx.saved_tensors
        B, S, H, I, heads, drop_attn, drop_hid1, drop_hid2, tail = ctx.meta
        M = B * S
        dev = x2.device
        # feed-forward sub-layer
        # (the bias gradients of the two output projections are column sums of dlin2 / dlin1, i.e. of the A operands of their
        # weight-gradient GEMMs: the grouped launch below delivers them, the LayerNorm backward does not have to)
        dres2, dlin2, dg2, dbe2, _ = _ln_bwd(_grad_bf16(g, H), y2, mean2, rstd2, g2, drop_hid2, False)
        du = _dgrad(dlin2, H, w2_16, M, H, I, act_aux=u, site=nat.SITE_FFN_DOWN_DGRAD)                  # (dlin2 W2) * gelu'(u)
        da = _dgrad(du, I, w1_16, M, I, H, dx_resid=dres2, site=nat.SITE_FFN_UP_DGRAD)     # du W1 + dres2  = gradient of the attention block's output
        # attention sub-layer
        dres1, dlin1, dg1, dbe1, _ = _ln_bwd(da, y1, mean1, rstd1, g1, drop_hid1, False)
        dctx = _dgrad(dlin1, H, wo16, M, H, H, site=nat.SITE_ATTN_OUT_DGRAD)
        dqkv = torch.empty(M, 3 * H, dtype=BF16, device=dev)
        delta = torch.empty(B, heads, S, dtype=F32, device=dev)
        scale = 1.0 / math.sqrt(H // heads)
        nat.attention_bwd(qkv, qkv[:, H:], qkv[:, 2 * H:], 3 * H, 3 * H, 3 * H, mask_add, ctxt, H, lse, B, heads, S, S, scale,
                          dctx, dqkv, dqkv[:, H:], dqkv[:, 2 * H:], delta, drop_attn, head_dim=H // heads, ctx_f32=o32, causal_tail=tail, keep_bits=kb)
        dx = _dgrad(dqkv, 3 * H, wqkv16, M, 3 * H, H, dx_resid=dres1, site=nat.SITE_QKV_DGRAD) if ctx.needs_input_grad[0] else None
        # the four weight gradients, one launch
        p_1, dw1, db1 = _wgrad_problem(du, I, a_out, M, I, H, True)
        p_2, dw2, db2 = _wgrad_problem(dlin2, H, hh, M, H, I, True)
        p_q, dwqkv, dbqkv = _wgrad_problem(dqkv, 3 * H, x2, M, 3 * H, H, True)
        p_o, dwo, dbo = _wgrad_problem(dlin1, H, ctxt, M, H, H, True)
        nat.gemm_grouped([p_1, p_2, p_q, p_o])
        return ((dx.view(B, S, H) if dx is not None else None),
                dwqkv[:H], dbqkv[:H], dwqkv[H:2 * H], dbqkv[H:2 * H], dwqkv[2 * H:], dbqkv[2 * H:], dwo, dbo, dg1, dbe1,
                dw1, db1, dw2, db2, dg2, dbe2) + (None,) * 13


# ---------------------------------------------------------------------------------------------
# LayerNorm
# ---------------------------------------------------------------------------------------------
class LayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        x2 = _as_bf16_2d(x)
        M, N = x2.shape
        out = torch.empty(M, N, dtype=BF16, device=x2.device)
        mean = torch.empty(M, dtype=F32, device=x2.device)
        rstd = torch.empty(M, dtype=F32, device=x2.device)
        nat.layernorm_fwd(x2, gamma.detach(), beta.detach(), out, mean, rstd, M, N, eps)
        ctx.save_for_backward(x2, mean, rstd, gamma.detach())
        ctx.xshape = x.shape
        return out.view(x.shape)

    @staticmethod
    def backward(ctx, g):
        x2, mean, rstd, gamma = ctx.saved_tensors
        dx, _, dgamma, dbeta, _ = _ln_bwd(_grad_bf16(g, x2.shape[1]), x2, mean, rstd, gamma, nat.NO_DROP, False)
        return dx.view(ctx.xshape), dgamma, dbeta, None


# ---------------------------------------------------------------------------------------------
# embeddings
# ---------------------------------------------------------------------------------------------
class VisioLinguisticEmbeddingsFn(torch.autograd.Function):
    """BertVisioLinguisticEmbeddings.forward (embeddings.py:423-459, image_text_alignment=None)."""

    @staticmethod
    def forward(ctx, input_ids, token_type_ids, feats, vtype, word, pos, typ, ln_w, ln_b, typ_vis, pos_vis, proj_w, proj_b,
                proj_w16, eps, drop, pad_idx=None, align=None):
        B, T = input_ids.shape
        H = word.shape[1]
        R = 0 if feats is None else feats.shape[1]
        S = T + R
        dev = word.device
        y = torch.empty(B * S, H, dtype=BF16, device=dev)
        ids = input_ids.contiguous()
        seg = token_type_ids.contiguous()
        nat.embed_text_fwd(ids, seg, word.detach(), pos.detach(), typ.detach(), y, B, T, S, H)
        f2 = None
        if R:
            D = feats.shape[2]
            f2 = _feature_rows(feats, B * R, D)
            vt = vtype.reshape(B * R).contiguous()
            if align is not None:       # image_text_alignment (embeddings.py:373-397): mean text-position row of the aligned words, per region
                al = align.reshape(B * R, -1).contiguous()
                if al.dtype != torch.int64:
                    al = al.long()
                addend = torch.empty(B * R, H, dtype=F32, device=dev)
                nat.align_pos_fwd(al, pos.detach(), typ_vis.detach(), vt, addend, B * R, al.shape[1], H)
                rows = torch.arange(B * R, dtype=torch.int64, device=dev)
                nat.gemm(f2, proj_w16, y, B * R, H, D, D, D, H, bias=proj_b.detach(), coladd=pos_vis.detach()[0],
                         rowtab=addend, rowidx=rows, rowtab_ld=H, grp=(R, T, T))
            else:
                al = None
                nat.gemm(f2, proj_w16, y, B * R, H, D, D, D, H, bias=proj_b.detach(), coladd=pos_vis.detach()[0],
                         rowtab=typ_vis.detach(), rowidx=vt, rowtab_ld=H, grp=(R, T, T))
        else:
            vt = al = None
        out = torch.empty(B * S, H, dtype=BF16, device=dev)
        mean = torch.empty(B * S, dtype=F32, device=dev)
        rstd = torch.empty(B * S, dtype=F32, device=dev)
        if drop[1] and nat.layernorm_dropout_fusable(H):      # LayerNorm + nn.Dropout (embeddings.py:343-345) as one launch, same bits as the two below
            nat.layernorm_dropout_fwd(y, ln_w.detach(), ln_b.detach(), out, mean, rstd, B * S, H, eps, drop)
        else:
            nat.layernorm_fwd(y, ln_w.detach(), ln_b.detach(), out, mean, rstd, B * S, H, eps)
            if drop[1]:
                out2 = torch.empty_like(out)
                nat.dropout(out, out2, drop)
                out = out2
        ctx.save_for_backward(ids, seg, f2, vt, y, mean, rstd, ln_w.detach(), proj_w16, al)
        ctx.meta = (B, T, R, S, H, drop, word.shape[0], pos.shape[0], typ.shape[0], typ_vis.shape[0], pos_vis.shape[0])
        ctx.pad_idx = -1 if pad_idx is None else int(pad_idx)
        return out.view(B, S, H)

    @staticmethod
    def backward(ctx, g):
        ids, seg, f2, vt, y, mean, rstd, ln_w, proj_w16, al = ctx.saved_tensors
        B, T, R, S, H, drop, V, P, NT, NTV, PV = ctx.meta
        dev = y.device
        dy = _grad_bf16(g, H)
        if drop[1] and nat.layernorm_dropout_fusable(H):      # dropout backward applied while the LayerNorm backward loads dy (same bits as the two launches)
            dpre, dgamma, dbeta = _ln_bwd_din(dy, y, mean, rstd, ln_w, drop)
        else:
            if drop[1]:
                d2 = torch.empty_like(dy)
                nat.dropout(dy, d2, drop)
                dy = d2
            dpre, _, dgamma, dbeta, _ = _ln_bwd(dy, y, mean, rstd, ln_w, nat.NO_DROP, False)
        # ONE zero fill for the five table gradients (row blocks of one buffer: each gradient is a contiguous [rows, H] view of it)
        tabs = torch.zeros(V + P + NT + ((NTV + PV) if R else 0), H, dtype=F32, device=dev)
        dword, dpos, dtyp = tabs[:V], tabs[V:V + P], tabs[V + P:V + P + NT]
        nat.rows_scatter_add(dpre, H, B, T, S, ids, T, 0, 0, dword, H, 0, ctx.pad_idx)   # padding_idx rows get no gradient
        dtyp_vis = dpos_vis = dproj_w = dproj_b = None
        if R:
            dtyp_vis, dpos_vis = tabs[V + P + NT:V + P + NT + NTV], tabs[V + P + NT + NTV:]
        # text positions, text token types, visual token types, the visual position row: one pass over dpre (two launches instead of seven)
        nat.embed_tables_bwd(dpre, H, B, T, R, seg, vt, 0, dpos, dtyp, dtyp_vis, dpos_vis, H)
        if R:
            vis = dpre[T:]  # row (b, r) of the visual block lives at dpre[b*S + T + r]
            if al is not None:          # the aligned words' TEXT position rows collect the regions' gradients / count
                nat.align_pos_bwd(vis, H, B, R, S, al, dpos, al.shape[1], H)
            dvis = torch.empty(B * R, H, dtype=BF16, device=dev)
            nat.copy_rows(vis, S, dvis, R, B, R, H)
            D = f2.shape[1]
            dproj_w = torch.empty(H, D, dtype=F32, device=dev)
            if f2.dtype == BF16:      # the bias gradient = row sums of the GEMM's A operand: delivered by the same launch
                dproj_b = torch.empty(H, dtype=F32, device=dev)
                nat.gemm(dvis, f2, dproj_w, H, D, B * R, H, D, D, a_kmajor=True, b_kmajor=True, rowsum_out=dproj_b)
            else:                     # (fp32 features staged by the GEMM: its row-sum form takes bf16 operands)
                nat.gemm(dvis, f2, dproj_w, H, D, B * R, H, D, D, a_kmajor=True, b_kmajor=True)
                dproj_b = _colsum(dvis, H, B * R, H)
        return (None, None, None, None, dword, dpos, dtyp, dgamma, dbeta, dtyp_vis, dpos_vis, dproj_w, dproj_b, None, None, None, None, None)


# ---------------------------------------------------------------------------------------------
# pooled-token gather (+ dropout) and the loss
# ---------------------------------------------------------------------------------------------
class GatherRowsFn(torch.autograd.Function):
    """out[b] = dropout(x[b, index[b]]): torch.gather + nn.Dropout of visual_bert.py:389-400."""

    @staticmethod
    def forward(ctx, x, index, drop):
        B, S, H = x.shape
        x2 = _as_bf16_2d(x)
        out = torch.empty(B, H, dtype=BF16, device=x2.device)
        idx = index.contiguous()
        nat.gather_rows(x2, idx, out, B, S, H, drop)
        ctx.save_for_backward(idx)
        ctx.meta = (B, S, H, drop)
        return out

    @staticmethod
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        B, S, H, drop = ctx.meta
        dx = torch.zeros(B * S, H, dtype=BF16, device=g.device)
        nat.scatter_rows(_grad_bf16(g, H), idx, dx, B, S, H, drop)
        return dx.view(B, S, H), None, None


class LogitBCEFn(torch.autograd.Function):
    """mean(BCEWithLogits(scores, targets)) * num_labels   (losses.py:246-251)."""

    @staticmethod
    def forward(ctx, scores, targets):
        B, N = scores.shape
        s = scores.float().contiguous()
        t = targets.float().contiguous()
        loss = torch.empty(1, dtype=F32, device=s.device)
        nat.bce_logits_fwd(s, t, loss, B, N)
        ctx.save_for_backward(s, t)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        s, t = ctx.saved_tensors
        B, N = s.shape
        ldd = _pad8(N)
        d16 = torch.empty(B, ldd, dtype=BF16, device=s.device)
        nat.bce_logits_bwd(s, t, g.float().reshape(1).contiguous(), d16, ldd, B, N)
        d = torch.empty(B, N, dtype=F32, device=s.device)
        nat.cast2d_bf16_to_f32(d16, ldd, d, N, B, N)
        return d, None


# ---------------------------------------------------------------------------------------------
# MMBT pieces (mmf/models/mmbt.py)
# ---------------------------------------------------------------------------------------------
class DropoutFn(torch.autograd.Function):
    """nn.Dropout on a bf16 activation with the counter-hash mask (same mask regenerated in backward)."""

    @staticmethod
    def forward(ctx, x, drop):
        x2 = _as_bf16_2d(x)
        y = torch.empty_like(x2)
        nat.dropout(x2, y, drop)
        ctx.drop = drop
        return y.view(x.shape)

    @staticmethod
    def backward(ctx, g):
        g2 = _grad_bf16(g, g.shape[-1])
        d = torch.empty_like(g2)
        nat.dropout(g2, d, ctx.drop)
        return d.view(g.shape), None


class LinearTanhFn(torch.autograd.Function):
    """HF BertPooler body: tanh(x W^T + b), tanh fused in the GEMM epilogue (act = 3)."""

    @staticmethod
    def forward(ctx, x, weight, bias, w16):
        x2 = _as_bf16_2d(x)
        M, K = x2.shape
        N = w16.shape[0]
        y = torch.empty(M, N, dtype=BF16, device=x2.device)
        nat.gemm(x2, w16, y, M, N, K, K, K, N, bias=bias.detach(), act=3)
        ctx.save_for_backward(x2, y, w16)
        ctx.xshape = x.shape
        return y.view(*x.shape[:-1], N)

    @staticmethod
    def backward(ctx, g):
        x2, y, w16 = ctx.saved_tensors
        M, K = x2.shape
        N = w16.shape[0]
        dpre = torch.empty(M, N, dtype=BF16, device=x2.device)
        nat.tanh_bwd(_grad_bf16(g, N), y, dpre)
        dx, dw, db = _linear_bwd(dpre, N, x2, w16, M, N, K, need_dx=ctx.needs_input_grad[0], want_db=True)
        return (dx.view(ctx.xshape) if dx is not None else None), dw, db, None


def mmbt_modal_types(modal_type, B, L, N, s0, dev, pd, td):
    """The token-type ids of MMBT's modal block (ModalEmbeddings.forward, mmbt.py:117-127) in the forms the fused embedding stage takes them.
    `modal_type`: a Python int or a one-element device tensor — ONE type for the whole block, what MMBTBase derives from the batch's segment ids
    (mmbt.py:385-410): its row rides in the projection GEMM's epilogue as a column vector — or a [B, L] tensor of per-position ids (a caller's
    own `modal_token_type_ids`): each projected feature's position row and type row are then summed into one per-row table for the same epilogue.
    Returns (start-token types [B, 1], end-token types [B, 1], coladd, rowtab, rowidx, types of all L modal rows [B, L], feature types [B * N] or None)."""
    H = td.shape[1]
    posidx = (torch.arange(N, device=dev, dtype=torch.int64) + s0).repeat(B)
    if isinstance(modal_type, torch.Tensor) and modal_type.numel() > 1:
        ids = modal_type.detach().to(device=dev, dtype=torch.int64).reshape(B, L).contiguous()
        feat = ids[:, s0:s0 + N].reshape(-1).contiguous()
        rowtab = pd.index_select(0, posidx).add_(td.index_select(0, feat)).contiguous()
        return (ids[:, :1].contiguous(), ids[:, L - 1:].contiguous(), None, rowtab, torch.arange(B * N, device=dev, dtype=torch.int64), ids, feat)
    if isinstance(modal_type, torch.Tensor):      # (a one-element device tensor stays on the device: no host read-back, capturable)
        mt = modal_type.detach().reshape(1).to(device=dev, dtype=torch.int64)
    else:
        mt = torch.full((1,), int(modal_type), dtype=torch.int64, device=dev)
    mtype = mt.reshape(1, 1).expand(B, 1).contiguous()
    return mtype, mtype, td.index_select(0, mt).reshape(H), pd, posidx, mt.reshape(1, 1).expand(B, L).contiguous(), None


class MMBTEmbeddingsFn(torch.autograd.Function):
    """ModalEmbeddings.forward (mmbt.py:84-129) and BertEmbeddingsJit.forward for the text (hf_layers.py:108-135),
    concatenated modal-first (mmbt.py:225), in one buffer: [start token | N projected features | end token | T text].
    The two LayerNorm + dropout passes of the reference share their parameters, so one pass over all rows is the
    same computation."""

    @staticmethod
    def forward(ctx, feats, input_ids, start_tok, end_tok, text_type_ids, modal_type, word, pos, typ, ln_w, ln_b, proj_w, proj_b,
                proj_w16, eps, drop, pad_idx=None):
        B, N, D = feats.shape
        T = input_ids.shape[1]
        H = word.shape[1]
        s0 = 1 if start_tok is not None else 0
        L = N + s0 + (1 if end_tok is not None else 0)
        S = L + T
        dev = word.device
        y = torch.empty(B * S, H, dtype=BF16, device=dev)
        wd, pd, td = word.detach(), pos.detach(), typ.detach()
        mt_st, mt_en, coladd, rowtab, rowidx, mtype_rows, _ = mmbt_modal_types(modal_type, B, L, N, s0, dev, pd, td)
        ids = input_ids.contiguous()
        seg = text_type_ids.contiguous()
        st = start_tok.reshape(B, 1).contiguous() if start_tok is not None else None
        en = end_tok.reshape(B, 1).contiguous() if end_tok is not None else None
        if st is not None:
            nat.embed_text_fwd(st, mt_st, wd, pd, td, y, B, 1, S, H, 0, 0)
        if en is not None:
            nat.embed_text_fwd(en, mt_en, wd, pd, td, y, B, 1, S, H, s0 + N, s0 + N)
        nat.embed_text_fwd(ids, seg, wd, pd, td, y, B, T, S, H, L, 0)
        f2 = _feature_rows(feats, B * N, D)
        nat.gemm(f2, proj_w16, y, B * N, H, D, D, D, H, bias=proj_b.detach(), coladd=coladd, rowtab=rowtab,
                 rowidx=rowidx, rowtab_ld=H, grp=(N, S - N, s0))
        out = torch.empty(B * S, H, dtype=BF16, device=dev)
        mean = torch.empty(B * S, dtype=F32, device=dev)
        rstd = torch.empty(B * S, dtype=F32, device=dev)
        nat.layernorm_fwd(y, ln_w.detach(), ln_b.detach(), out, mean, rstd, B * S, H, eps)
        if drop[1]:
            out2 = torch.empty_like(out)
            nat.dropout(out, out2, drop)
            out = out2
        ctx.save_for_backward(ids, seg, st, en, f2, y, mean, rstd, ln_w.detach(), proj_w16, mtype_rows)
        ctx.meta = (B, N, T, S, L, s0, H, drop, word.shape[0], pos.shape[0], typ.shape[0])
        ctx.pad_idx = -1 if pad_idx is None else int(pad_idx)
        return out.view(B, S, H)

    @staticmethod
    def backward(ctx, g):
        ids, seg, st, en, f2, y, mean, rstd, ln_w, proj_w16, mtype_rows = ctx.saved_tensors
        B, N, T, S, L, s0, H, drop, V, P, NT = ctx.meta
        dev = y.device
        dy = _grad_bf16(g, H)
        if drop[1]:
            d2 = torch.empty_like(dy)
            nat.dropout(dy, d2, drop)
            dy = d2
        dpre, _, dgamma, dbeta, _ = _ln_bwd(dy, y, mean, rstd, ln_w, nat.NO_DROP, False)
        dword = torch.zeros(V, H, dtype=F32, device=dev)
        nat.rows_scatter_add(dpre[L:], H, B, T, S, ids, T, 0, 0, dword, H, 0, ctx.pad_idx)
        if st is not None:
            nat.rows_scatter_add(dpre, H, B, 1, S, st, 1, 0, 0, dword, H, 0, ctx.pad_idx)
        if en is not None:
            nat.rows_scatter_add(dpre[s0 + N:], H, B, 1, S, en, 1, 0, 0, dword, H, 0, ctx.pad_idx)
        dpos = torch.zeros(P, H, dtype=F32, device=dev)
        nat.rows_scatter_add(dpre, H, B, L, S, None, 0, 1, 0, dpos, H, 0)        # modal block: position = row index
        nat.rows_scatter_add(dpre[L:], H, B, T, S, None, 0, 1, 0, dpos, H, 0)    # text: positions restart at 0
        dtyp = torch.zeros(NT, H, dtype=F32, device=dev)
        nat.rows_scatter_add(dpre[L:], H, B, T, S, seg, T, 0, 0, dtyp, H, 1)
        nat.rows_scatter_add(dpre, H, B, L, S, mtype_rows, L, 0, 0, dtyp, H, 1)
        dvis = dpre.view(B, S, H)[:, s0:s0 + N, :].contiguous().view(B * N, H)
        D = f2.shape[1]
        dproj_w = torch.empty(H, D, dtype=F32, device=dev)
        nat.gemm(dvis, f2, dproj_w, H, D, B * N, H, D, D, a_kmajor=True, b_kmajor=True)
        dproj_b = _colsum(dvis, H, B * N, H)
        dfeats = None
        if ctx.needs_input_grad[0]:        # a trainable modal encoder ahead of the projection (finetune_faster_rcnn_fpn_fc7): dX = dY W
            dfeats = _dgrad(dvis, H, proj_w16, B * N, H, D).view(B, N, D)
        return (dfeats, None, None, None, None, None, dword, dpos, dtyp, dgamma, dbeta, dproj_w, dproj_b, None, None, None, None)


class CrossEntropyFn(torch.autograd.Function):
    """nn.CrossEntropyLoss(ignore_index) over [B, C] fp32 logits (MMF `cross_entropy`, losses.py:595-602)."""

    @staticmethod
    def forward(ctx, scores, targets, ignore_index):
        B, Cn = scores.shape
        s = scores.float().contiguous()
        t = targets.contiguous()
        loss = torch.empty(1, dtype=F32, device=s.device)
        count = torch.empty(1, dtype=F32, device=s.device)
        nat.cross_entropy_fwd(s, t, loss, count, B, Cn, ignore_index)
        ctx.save_for_backward(s, t, count)
        ctx.ignore_index = ignore_index
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        s, t, count = ctx.saved_tensors
        B, Cn = s.shape
        d = torch.empty(B, Cn, dtype=F32, device=s.device)
        nat.cross_entropy_bwd(s, t, count, g.float().reshape(1).contiguous(), d, B, Cn, ctx.ignore_index)
        return d, None, None


class MaskedLMHeadFn(torch.autograd.Function):
    """The decoder + loss of VisualBERTForPretraining (mmf/models/visual_bert.py:267-277): prediction scores = h W_word^T + b over every
    position of the joint sequence (HF BertLMPredictionHead.decoder, weight tied to the word embeddings, :227-235), then
    nn.CrossEntropyLoss(ignore_index) over [B * S, vocab].  Returns (loss, logits [B, S, vocab] fp32).

    ONE autograd node so that backward never materialises an fp32 [B * S, vocab] gradient: the loss kernel keeps each row's
    log-sum-exp and the backward kernel writes gloss / count * (softmax - onehot) straight into the zero-padded bf16 operand
    of the decoder's input- and weight-gradient GEMMs (ignored rows — the visual positions and ~85 % of the text — are zeros).
    `logits` is returned for the output dict (as the reference does) and marked non-differentiable: only the loss carries
    gradient, which is all MMF's trainer ever differentiates."""

    @staticmethod
    def forward(ctx, x, weight, bias, w16, labels, ignore_index):
        x2 = _as_bf16_2d(x)
        M, K = x2.shape
        N = weight.shape[0]
        dev = x2.device
        logits = torch.empty(M, N, dtype=F32, device=dev)
        nat.gemm(x2, w16, logits, M, N, K, K, K, N, bias=bias.detach())
        lab = labels.reshape(M).contiguous()
        lse = torch.empty(M, dtype=F32, device=dev)
        rowloss = torch.empty(M, dtype=F32, device=dev)
        loss = torch.empty(1, dtype=F32, device=dev)
        count = torch.empty(1, dtype=F32, device=dev)
        nat.vocab_cross_entropy_fwd(logits, lab, lse, rowloss, loss, count, M, N, ignore_index)
        ctx.save_for_backward(x2, w16, logits, lab, lse, count)
        ctx.meta = (M, N, K, x.shape, ignore_index)
        out = logits.view(*x.shape[:-1], N)
        ctx.mark_non_differentiable(out)
        return loss[0], out

    @staticmethod
    def backward(ctx, gloss, _glogits):
        x2, w16, logits, lab, lse, count = ctx.saved_tensors
        M, N, K, xshape, ignore_index = ctx.meta
        ldd = _pad8(N)
        d = torch.empty(M, ldd, dtype=BF16, device=x2.device)
        nat.vocab_cross_entropy_bwd(logits, lab, lse, count, gloss.float().reshape(1).contiguous(), d, ldd, M, N, ignore_index)
        dx, dw, db = _linear_bwd(d, ldd, x2, w16, M, N, K, need_dx=ctx.needs_input_grad[0], want_db=True)
        return (dx.view(xshape) if dx is not None else None), dw, db, None, None, None


class MaskedRegionHeadFn(torch.autograd.Function):
    """The decoder + loss of ViLBERT's masked-region classification (mmf/models/vilbert.py:846-858 BertImagePredictionHead.decoder,
    :1150-1157 `visual_target: 0`): prediction_scores_v = h W^T + b over every region, then KLDivLoss(log_softmax(scores), target)
    summed over the regions with image_label == 1 and divided by their number.  Returns (loss, scores [B, R, v_target_size] fp32).
    One autograd node, like MaskedLMHeadFn: backward writes gloss / count * (softmax * sum(target) - target) directly as the
    zero-padded bf16 operand of the decoder's gradient GEMMs; `scores` is returned non-differentiable."""

    @staticmethod
    def forward(ctx, x, weight, bias, w16, target, row_label):
        x2 = _as_bf16_2d(x)
        M, K = x2.shape
        N = weight.shape[0]
        dev = x2.device
        logits = torch.empty(M, N, dtype=F32, device=dev)
        nat.gemm(x2, w16, logits, M, N, K, K, K, N, bias=bias.detach())
        tgt = target.reshape(M, N)
        tgt = (tgt if tgt.dtype == F32 else tgt.float()).contiguous()
        lab = row_label.reshape(M).contiguous()
        if lab.dtype != torch.int64:
            lab = lab.long()
        lse = torch.empty(M, dtype=F32, device=dev)
        tsum = torch.empty(M, dtype=F32, device=dev)
        rowloss = torch.empty(M, dtype=F32, device=dev)
        loss = torch.empty(1, dtype=F32, device=dev)
        count = torch.empty(1, dtype=F32, device=dev)
        nat.soft_target_kl_fwd(logits, tgt, lab, lse, tsum, rowloss, loss, count, M, N)
        ctx.save_for_backward(x2, w16, logits, tgt, lab, lse, tsum, count)
        ctx.meta = (M, N, K, x.shape)
        out = logits.view(*x.shape[:-1], N)
        ctx.mark_non_differentiable(out)
        return loss[0], out

    @staticmethod
    def backward(ctx, gloss, _glogits):
        x2, w16, logits, tgt, lab, lse, tsum, count = ctx.saved_tensors
        M, N, K, xshape = ctx.meta
        ldd = _pad8(N)
        d = torch.empty(M, ldd, dtype=BF16, device=x2.device)
        nat.soft_target_kl_bwd(logits, tgt, lab, lse, tsum, count, gloss.float().reshape(1).contiguous(), d, ldd, M, N)
        dx, dw, db = _linear_bwd(d, ldd, x2, w16, M, N, K, need_dx=ctx.needs_input_grad[0], want_db=True)
        return (dx.view(xshape) if dx is not None else None), dw, db, None, None, None


class MaskedRegionRegressionFn(torch.autograd.Function):
    """The decoder + loss of ViLBERT's masked-region REGRESSION (`visual_target: 1`, mmf/models/vilbert.py:1074-1075, 1139-1148):
    prediction_scores_v = h W^T + b (BertImagePredictionHead.decoder, :846-858), nn.MSELoss(reduction="none") against the region targets,
    summed over the regions with image_label == 1 and divided by max(number of their elements, 1).  Returns (loss, scores fp32); one node:
    the loss kernel's backward writes the zero-padded bf16 operand of the decoder's gradient GEMMs (unlabelled rows are zeros)."""

    @staticmethod
    def forward(ctx, x, weight, bias, w16, target, row_label):
        x2 = _as_bf16_2d(x)
        M, K = x2.shape
        N = weight.shape[0]
        dev = x2.device
        pred = torch.empty(M, N, dtype=F32, device=dev)
        nat.gemm(x2, w16, pred, M, N, K, K, K, N, bias=bias.detach())
        tgt = target.reshape(M, N)
        tgt = (tgt if tgt.dtype == F32 else tgt.float()).contiguous()
        lab = row_label.reshape(M).contiguous()
        if lab.dtype != torch.int64:
            lab = lab.long()
        loss = torch.empty(1, dtype=F32, device=dev)
        count = torch.empty(1, dtype=F32, device=dev)
        nat.mse_fwd(pred, tgt, loss, M, N, row_label=lab, count=count)
        ctx.save_for_backward(x2, w16, pred, tgt, lab, count)
        ctx.meta = (M, N, K, x.shape)
        out = pred.view(*x.shape[:-1], N)
        ctx.mark_non_differentiable(out)
        return loss[0], out

    @staticmethod
    def backward(ctx, gloss, _gscores):
        x2, w16, pred, tgt, lab, count = ctx.saved_tensors
        M, N, K, xshape = ctx.meta
        ldd = _pad8(N)
        d = torch.empty(M, ldd, dtype=BF16, device=x2.device)
        nat.mse_bwd(pred, tgt, gloss.float().reshape(1).contiguous(), d, ldd, M, N, row_label=lab, count=count)
        dx, dw, db = _linear_bwd(d, ldd, x2, w16, M, N, K, need_dx=ctx.needs_input_grad[0], want_db=True)
        return (dx.view(xshape) if dx is not None else None), dw, db, None, None, None


class ExpandBatchFn(torch.autograd.Function):
    """ViLBERT's `in_batch_pairs` / `fast_mode` batch expansion (mmf/models/vilbert.py:678-725) of a bf16 activation [Bs, L, H] -> [reps * Bs, L, H]:
    mode 0 `x.unsqueeze(0).expand(reps, ...)` (pair (i, j) takes sample j), mode 1 `x.unsqueeze(1).expand(.., reps, ...)` (pair (i, j) takes sample i).
    Backward: the sum over the broadcast index (fp32 accumulation)."""

    @staticmethod
    def forward(ctx, x, reps, mode):
        Bs, L, H = x.shape
        x2 = _as_bf16_2d(x)
        out = torch.empty(reps * Bs * L, H, dtype=BF16, device=x2.device)
        nat.expand_batch(x2, out, Bs, reps, L * H, mode)
        ctx.meta = (Bs, L, H, reps, mode)
        return out.view(reps * Bs, L, H)

    @staticmethod
    def backward(ctx, g):
        Bs, L, H, reps, mode = ctx.meta
        g2 = _grad_bf16(g, H)
        dx = torch.empty(Bs * L, H, dtype=BF16, device=g2.device)
        nat.reduce_batch(g2, dx, Bs, reps, L * H, mode)
        return dx.view(Bs, L, H), None, None


class MaskedRegionNCEFn(torch.autograd.Function):
    """The decoder + loss of ViLBERT's masked-region NCE form (`visual_target: 2`, mmf/models/vilbert.py:1158-1227): prediction_scores_v =
    h W^T + b, each masked region's prediction scored against its own target feature and K sampled negatives (the batched product of :1221),
    CrossEntropyLoss against class 0.  `neg_index` int64 [B, R, K]: flat indices into the B * R regions (:1176, :1200).  Returns (loss, scores
    fp32); the loss kernel's backward writes the zero-padded bf16 operand of the decoder's gradient GEMMs."""

    @staticmethod
    def forward(ctx, x, weight, bias, w16, target, row_label, neg_index):
        x2 = _as_bf16_2d(x)
        M, K = x2.shape
        N = weight.shape[0]
        dev = x2.device
        pred = torch.empty(M, N, dtype=F32, device=dev)
        nat.gemm(x2, w16, pred, M, N, K, K, K, N, bias=bias.detach())
        tgt = target.reshape(M, N)
        tgt = (tgt if tgt.dtype == F32 else tgt.float()).contiguous()
        lab = row_label.reshape(M).contiguous()
        if lab.dtype != torch.int64:
            lab = lab.long()
        neg = neg_index.reshape(M, -1).contiguous()
        if neg.dtype != torch.int64:
            neg = neg.long()
        NK = neg.shape[1]
        scores = torch.empty(M, NK + 1, dtype=F32, device=dev)
        lse = torch.empty(M, dtype=F32, device=dev); rowloss = torch.empty(M, dtype=F32, device=dev)
        loss = torch.empty(1, dtype=F32, device=dev); count = torch.empty(1, dtype=F32, device=dev)
        nat.nce_fwd(pred, tgt, neg, lab, scores, lse, rowloss, loss, count, M, N, NK)
        ctx.save_for_backward(x2, w16, tgt, neg, lab, scores, lse, count)
        ctx.meta = (M, N, K, NK, x.shape)
        out = pred.view(*x.shape[:-1], N)
        ctx.mark_non_differentiable(out)
        return loss[0], out

    @staticmethod
    def backward(ctx, gloss, _gscores):
        x2, w16, tgt, neg, lab, scores, lse, count = ctx.saved_tensors
        M, N, K, NK, xshape = ctx.meta
        ldd = _pad8(N)
        d = torch.empty(M, ldd, dtype=BF16, device=x2.device)
        nat.nce_bwd(tgt, neg, lab, scores, lse, count, gloss.float().reshape(1).contiguous(), d, ldd, M, N, NK)
        dx, dw, db = _linear_bwd(d, ldd, x2, w16, M, N, K, need_dx=ctx.needs_input_grad[0], want_db=True)
        return (dx.view(xshape) if dx is not None else None), dw, db, None, None, None, None


class TiedRegressionMSEFn(torch.autograd.Function):
    """MRFR's last two lines (mmf/models/transformers/heads/mrfr.py:85-90): prediction = h W + b with the TIED image-embedding
    weight W [hidden, img_dim] (UNITERImageEmbeddings.img_linear.weight applied transposed), loss = mean squared error against the targets.
    The projection is the GEMM's NN form (W read k-major: no transposed copy) with fp32 output; the loss kernel's backward writes the
    bf16 operand of the input-gradient (NT against W) and weight-gradient (TN: dW = h^T d) GEMMs; the bias gradient is its column sum."""

    @staticmethod
    def forward(ctx, h, weight, bias, w16, targets):
        h2 = _as_bf16_2d(h)
        n, K = h2.shape
        D = weight.shape[1]
        t = targets.float().contiguous()
        loss = torch.empty(1, dtype=F32, device=h2.device)
        pred = torch.empty(max(n, 1), D, dtype=F32, device=h2.device)
        if n:
            nat.gemm(h2, w16, pred, n, D, K, K, D, D, b_kmajor=True, bias=bias.detach())
            nat.mse_fwd(pred, t, loss, n, D)
        else:
            loss.fill_(float("nan"))          # (F.mse_loss of an empty tensor)
        ctx.save_for_backward(h2, w16, pred, t)
        ctx.meta = (h.shape, n, K, D)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        h2, w16, pred, t = ctx.saved_tensors
        hshape, n, K, D = ctx.meta
        dev = h2.device
        if not n:
            return torch.zeros(hshape, dtype=BF16, device=dev), torch.zeros(K, D, dtype=F32, device=dev), torch.zeros(D, dtype=F32, device=dev), None, None
        ldd = _pad8(D)
        d16 = torch.empty(n, ldd, dtype=BF16, device=dev)
        nat.mse_bwd(pred, t, g.float().reshape(1).contiguous(), d16, ldd, n, D)
        dh = torch.empty(n, K, dtype=BF16, device=dev)
        nat.gemm(d16, w16, dh, n, K, D, ldd, D, K)                                  # dh[m, k] = sum_n d[m, n] W[k, n]
        dw = torch.empty(K, D, dtype=F32, device=dev)
        nat.gemm(h2, d16, dw, K, D, n, K, ldd, D, a_kmajor=True, b_kmajor=True)     # dW[k, n] = sum_m h[m, k] d[m, n]
        db = _colsum(d16, ldd, n, D)
        return dh.view(hshape), dw, db, None, None


class WordRegionAlignmentFn(torch.autograd.Function):
    """WRA.forward (mmf/models/transformers/heads/wra.py:36-83) over `optimal_transport_dist` (mmf/modules/ot.py:87-110): returns
    (loss, per-sample OT distance).  One workgroup per sample: cosine cost, 50 IPOT steps with the plan in LDS, trace(C T); the plan is
    a constant of the backward pass (ot.py:38 `@torch.no_grad()`), which differentiates the cost matrix and the two normalisations."""

    @staticmethod
    def forward(ctx, sequence_output, txt_len, img_len, txt_pad, img_pad, labels):
        B, S, H = sequence_output.shape
        seq = _as_bf16_2d(sequence_output)
        M, N = int(txt_len), int(img_len)
        dev = seq.device
        tp = txt_pad.reshape(B, M).float().contiguous()
        ip = img_pad.reshape(B, N).float().contiguous()
        lab = labels.reshape(B).long().contiguous()
        xinv = torch.empty(B, M, dtype=F32, device=dev); yinv = torch.empty(B, N, dtype=F32, device=dev)
        plan = torch.empty(B, N, M, dtype=F32, device=dev); cost = torch.empty(B, M, N, dtype=F32, device=dev)
        dist = torch.empty(B, dtype=F32, device=dev)
        loss = torch.empty(1, dtype=F32, device=dev); count = torch.empty(1, dtype=F32, device=dev)
        nat.wra_fwd(seq, H, B, S, H, M, N, tp, ip, lab, xinv, yinv, plan, cost, dist, loss, count)
        ctx.save_for_backward(seq, tp, ip, lab, xinv, yinv, plan, cost, dist, count)
        ctx.meta = (B, S, H, M, N, sequence_output.shape)
        ctx.mark_non_differentiable(dist)
        return loss[0], dist

    @staticmethod
    def backward(ctx, g, _gdist):
        seq, tp, ip, lab, xinv, yinv, plan, cost, dist, count = ctx.saved_tensors
        B, S, H, M, N, shape = ctx.meta
        dseq = (torch.zeros if S > M + N else torch.empty)(B * S, H, dtype=BF16, device=seq.device)
        nat.wra_bwd(seq, H, B, S, H, M, N, tp, ip, lab, xinv, yinv, plan, cost, dist, g.float().reshape(1).contiguous(), count, dseq, H)
        return dseq.view(shape), None, None, None, None, None


class AttachedZeroFn(torch.autograd.Function):
    """A scalar 0 that depends on its inputs and hands each of them an all-zero gradient: what the reference's
    `nan_to_num(cross_entropy(<zero rows>))` is to autograd (mlm.py:89-94) when a batch has no masked token.  The value never
    reads the inputs, so NaN/inf in them cannot leak into the loss."""

    @staticmethod
    def forward(ctx, *tensors):
        ctx.meta = [(t.shape, t.dtype, t.device) for t in tensors]
        return torch.zeros((), dtype=torch.float32, device=tensors[0].device)

    @staticmethod
    def backward(ctx, g):
        return tuple(torch.zeros(s, dtype=d, device=dev) for s, d, dev in ctx.meta)


class TakeRowsFn(torch.autograd.Function):
    """out[r] = x[idx[r]] for a list of DISTINCT row indices: the `sequence_output[masked_tokens, :]` compaction of the MLM
    transformer head (mmf/models/transformers/heads/mlm.py:80-83) — only the masked positions go through the vocabulary
    projection.  Backward puts each gradient row back (the other rows are zero)."""

    @staticmethod
    def forward(ctx, x, idx):
        x2 = _as_bf16_2d(x)
        M, H = x2.shape
        n = idx.numel()
        out = torch.empty(n, H, dtype=BF16, device=x2.device)
        if n:
            nat.gather_rows2(x2, x2[:0], idx.contiguous(), out, n, H)
        ctx.save_for_backward(idx)
        ctx.meta = (x.shape, M, H, n)
        return out

    @staticmethod
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        xshape, M, H, n = ctx.meta
        d32 = torch.zeros(M, H, dtype=F32, device=g.device)
        if n:
            nat.rows_scatter_add(_grad_bf16(g, H), H, 1, n, 0, idx.contiguous(), n, 0, 0, d32, H, 0)
        dx = torch.empty(M, H, dtype=BF16, device=g.device)
        nat.cast_f32_to_bf16(d32, dx)
        return dx.view(xshape), None


# ---------------------------------------------------------------------------------------------
# MMF Transformer pieces (mmf/models/transformers/backends/huggingface.py)
# ---------------------------------------------------------------------------------------------
class AddPosTypeFn(torch.autograd.Function):
    """total = tok + pos_emb(arange(L)) + token_type_embeddings(segment_ids)   (huggingface.py:147-155) for a modality
    whose token embedding is a projection (`x` [B, L, H] bf16).  Either table may be None."""

    @staticmethod
    def forward(ctx, x, seg, pos, typ):
        B, L, H = x.shape
        x2 = _as_bf16_2d(x)
        y = torch.empty(B * L, H, dtype=BF16, device=x2.device)
        sg = seg.contiguous() if (seg is not None and typ is not None) else None
        nat.rows_add_embed(x2, sg, pos.detach() if pos is not None else None, typ.detach() if sg is not None else None, y, B, L, L, H)
        ctx.save_for_backward(sg)
        ctx.meta = (B, L, H, None if pos is None else pos.shape[0], None if typ is None else typ.shape[0])
        return y.view(B, L, H)

    @staticmethod
    def backward(ctx, g):
        (sg,) = ctx.saved_tensors
        B, L, H, P, NT = ctx.meta
        g2 = _grad_bf16(g, H)
        dpos = dtyp = None
        if P is not None:
            dpos = torch.zeros(P, H, dtype=F32, device=g2.device)
            nat.rows_scatter_add(g2, H, B, L, L, None, 0, 1, 0, dpos, H, 0)
        if NT is not None and sg is not None:
            dtyp = torch.zeros(NT, H, dtype=F32, device=g2.device)
            nat.rows_scatter_add(g2, H, B, L, L, sg, L, 0, 0, dtyp, H, 1)
        return g2.view(B, L, H), None, dpos, dtyp


class ConcatRowsFn(torch.autograd.Function):
    """torch.cat(list_embeddings, dim=1) (huggingface.py:159) of [B, L_m, H] bf16 blocks as strided HIP copies."""

    @staticmethod
    def forward(ctx, *xs):
        B, _, H = xs[0].shape
        lens = [int(x.shape[1]) for x in xs]
        S = sum(lens)
        out = torch.empty(B, S, H, dtype=BF16, device=xs[0].device)
        off = 0
        for x, L in zip(xs, lens):
            nat.copy_rows(_as_bf16_2d(x), L, out.view(B * S, H)[off:], S, B, L, H)
            off += L
        ctx.meta = (B, S, H, lens)
        return out

    @staticmethod
    def backward(ctx, g):
        B, S, H, lens = ctx.meta
        g2 = _grad_bf16(g, H)
        outs, off = [], 0
        for L in lens:
            d = torch.empty(B * L, H, dtype=BF16, device=g2.device)
            nat.copy_rows(g2[off:], S, d, L, B, L, H)
            outs.append(d.view(B, L, H))
            off += L
        return tuple(outs)


# ---------------------------------------------------------------------------------------------
# ViLBERT pieces (mmf/models/vilbert.py)
# ---------------------------------------------------------------------------------------------
class BiAttentionFn(torch.autograd.Function):
    """BertBiAttention.forward (vilbert.py:388-475): text queries over image keys/values (context_layer1, image mask,
    dropout1) and image queries over text keys/values (context_layer2, text mask, dropout2).  Each stream's Q|K|V is one
    packed GEMM; the two cross attentions read the other stream's K and V straight out of its packed buffer."""

    @staticmethod
    def forward(ctx, img, txt, q1w, q1b, k1w, k1b, v1w, v1b, q2w, q2b, k2w, k2b, v2w, v2b, w1_16, b1, w2_16, b2,
                img_mask_add, txt_mask_add, heads, drop1, drop2):
        """q1w .. v2b are the reference's six separate projections (autograd leaves); w*_16 / b* their packed shadows."""
        B, R, VH = img.shape
        T, H = txt.shape[1], txt.shape[2]
        BH = w1_16.shape[0] // 3
        hd = BH // heads
        i2, t2 = _as_bf16_2d(img), _as_bf16_2d(txt)
        dev = i2.device
        qkv1 = torch.empty(B * R, 3 * BH, dtype=BF16, device=dev)
        qkv2 = torch.empty(B * T, 3 * BH, dtype=BF16, device=dev)
        nat.gemm(i2, w1_16, qkv1, B * R, 3 * BH, VH, VH, VH, 3 * BH, bias=b1)
        nat.gemm(t2, w2_16, qkv2, B * T, 3 * BH, H, H, H, 3 * BH, bias=b2)
        scale = 1.0 / math.sqrt(hd)
        ctx1 = torch.empty(B * T, BH, dtype=BF16, device=dev)
        lse1 = torch.empty(B, heads, T, dtype=F32, device=dev)
        need = any(ctx.needs_input_grad)
        o1, o2 = _o32(B * T, BH, dev, need), _o32(B * R, BH, dev, need)
        kb1, kb2 = _keep_bits(B, heads, T, R, hd, drop1, dev, need), _keep_bits(B, heads, R, T, hd, drop2, dev, need)
        nat.attention_fwd(qkv2, qkv1[:, BH:], qkv1[:, 2 * BH:], 3 * BH, 3 * BH, 3 * BH, img_mask_add, ctx1, BH, lse1, B, heads, T, R,
                          scale, drop1, head_dim=hd, ctx_f32=o1, keep_bits=kb1)
        ctx2 = torch.empty(B * R, BH, dtype=BF16, device=dev)
        lse2 = torch.empty(B, heads, R, dtype=F32, device=dev)
        nat.attention_fwd(qkv1, qkv2[:, BH:], qkv2[:, 2 * BH:], 3 * BH, 3 * BH, 3 * BH, txt_mask_add, ctx2, BH, lse2, B, heads, R, T,
                          scale, drop2, head_dim=hd, ctx_f32=o2, keep_bits=kb2)
        ctx.save_for_backward(i2, t2, qkv1, qkv2, ctx1, ctx2, lse1, lse2, w1_16, w2_16, img_mask_add, txt_mask_add, o1, o2, kb1, kb2)
        ctx.meta = (B, R, T, VH, H, BH, heads, drop1, drop2)
        return ctx1.view(B, T, BH), ctx2.view(B, R, BH)

    @staticmethod
    def backward(ctx, g1, g2):
        i2, t2, qkv1, qkv2, ctx1, ctx2, lse1, lse2, w1_16, w2_16, img_mask_add, txt_mask_add, o1, o2, kb1, kb2 = ctx.saved_tensors
        B, R, T, VH, H, BH, heads, drop1, drop2 = ctx.meta
        hd = BH // heads
        dev = i2.device
        scale = 1.0 / math.sqrt(hd)
        dqkv1 = torch.empty(B * R, 3 * BH, dtype=BF16, device=dev)
        dqkv2 = torch.empty(B * T, 3 * BH, dtype=BF16, device=dev)
        delta1 = torch.empty(B, heads, T, dtype=F32, device=dev)
        delta2 = torch.empty(B, heads, R, dtype=F32, device=dev)
        nat.attention_bwd(qkv2, qkv1[:, BH:], qkv1[:, 2 * BH:], 3 * BH, 3 * BH, 3 * BH, img_mask_add, ctx1, BH, lse1, B, heads, T, R,
                          scale, _grad_bf16(g1, BH), dqkv2, dqkv1[:, BH:], dqkv1[:, 2 * BH:], delta1, drop1, head_dim=hd, ctx_f32=o1, keep_bits=kb1)
        nat.attention_bwd(qkv1, qkv2[:, BH:], qkv2[:, 2 * BH:], 3 * BH, 3 * BH, 3 * BH, txt_mask_add, ctx2, BH, lse2, B, heads, R, T,
                          scale, _grad_bf16(g2, BH), dqkv1, dqkv2[:, BH:], dqkv2[:, 2 * BH:], delta2, drop2, head_dim=hd, ctx_f32=o2, keep_bits=kb2)
        dimg, dw1, db1 = _linear_bwd(dqkv1, 3 * BH, i2, w1_16, B * R, 3 * BH, VH, want_db=True, defer=True)
        dtxt, dw2, db2 = _linear_bwd(dqkv2, 3 * BH, t2, w2_16, B * T, 3 * BH, H, want_db=True, defer=True)
        return (dimg.view(B, R, VH), dtxt.view(B, T, H),
                dw1[:BH], db1[:BH], dw1[BH:2 * BH], db1[BH:2 * BH], dw1[2 * BH:], db1[2 * BH:],
                dw2[:BH], db2[:BH], dw2[BH:2 * BH], db2[BH:2 * BH], dw2[2 * BH:], db2[2 * BH:],
                None, None, None, None, None, None, None, None, None)


class ImageFeatureEmbeddingsFn(torch.autograd.Function):
    """BertImageFeatureEmbeddings.forward (vilbert.py:904-913): LayerNorm(Linear(features) + Linear(5-d location)),
    dropout.  The 5-wide location operand is zero-padded to 8 columns (16-byte rows for the GEMM loader) and its GEMM
    adds the feature projection in the epilogue."""

    @staticmethod
    def forward(ctx, feats, loc, w_img, b_img, w_loc, b_loc, ln_w, ln_b, w_img16, eps, drop):
        B, R, D = feats.shape
        VH = w_img16.shape[0]
        M = B * R
        dev = w_img.device
        f2 = _feature_rows(feats, M, D)
        KL = loc.shape[-1]
        KP = _pad8(KL)
        l8 = torch.empty(M, KP, dtype=BF16, device=dev)
        nat.cast2d_f32_to_bf16(loc.reshape(M, KL).float().contiguous(), KL, l8, KP, M, KL)
        wl8 = torch.empty(VH, KP, dtype=BF16, device=dev)
        nat.cast2d_f32_to_bf16(w_loc.detach().contiguous(), KL, wl8, KP, VH, KL)
        y0 = torch.empty(M, VH, dtype=BF16, device=dev)
        nat.gemm(f2, w_img16, y0, M, VH, D, D, D, VH, bias=b_img.detach())
        y = torch.empty(M, VH, dtype=BF16, device=dev)
        nat.gemm(l8, wl8, y, M, VH, KP, KP, KP, VH, bias=b_loc.detach(), resid=y0, ldr=VH)
        out = torch.empty(M, VH, dtype=BF16, device=dev)
        mean = torch.empty(M, dtype=F32, device=dev)
        rstd = torch.empty(M, dtype=F32, device=dev)
        nat.layernorm_fwd(y, ln_w.detach(), ln_b.detach(), out, mean, rstd, M, VH, eps)
        if drop[1]:
            o2 = torch.empty_like(out)
            nat.dropout(out, o2, drop)
            out = o2
        ctx.save_for_backward(f2, l8, y, mean, rstd, ln_w.detach())
        ctx.meta = (B, R, D, VH, KL, KP, drop)
        return out.view(B, R, VH)

    @staticmethod
    def backward(ctx, g):
        f2, l8, y, mean, rstd, ln_w = ctx.saved_tensors
        B, R, D, VH, KL, KP, drop = ctx.meta
        M = B * R
        dev = y.device
        dy = _grad_bf16(g, VH)
        if drop[1]:
            d2 = torch.empty_like(dy)
            nat.dropout(dy, d2, drop)
            dy = d2
        dpre, _, dgamma, dbeta, _ = _ln_bwd(dy, y, mean, rstd, ln_w, nat.NO_DROP, False)
        dw_img = torch.empty(VH, D, dtype=F32, device=dev)
        nat.gemm(dpre, f2, dw_img, VH, D, M, VH, D, D, a_kmajor=True, b_kmajor=True)
        dwl8 = torch.empty(VH, KP, dtype=F32, device=dev)
        nat.gemm(dpre, l8, dwl8, VH, KP, M, VH, KP, KP, a_kmajor=True, b_kmajor=True)
        db = _colsum(dpre, VH, M, VH)
        return (None, None, dw_img, db, dwl8[:, :KL].contiguous(), db.clone(), dgamma, dbeta, None, None, None)


class EltwiseMulFn(torch.autograd.Function):
    """pooled_output_t * pooled_output_v (vilbert.py:1318)."""

    @staticmethod
    def forward(ctx, a, b):
        a2, b2 = _as_bf16_2d(a), _as_bf16_2d(b)
        out = torch.empty_like(a2)
        nat.eltwise(0, a2, b2, out)
        ctx.save_for_backward(a2, b2)
        return out.view(a.shape)

    @staticmethod
    def backward(ctx, g):
        a2, b2 = ctx.saved_tensors
        g2 = _grad_bf16(g, g.shape[-1])
        da, db = torch.empty_like(a2), torch.empty_like(b2)
        nat.eltwise(0, g2, b2, da)
        nat.eltwise(0, g2, a2, db)
        return da.view(g.shape), db.view(g.shape)


class ReluFn(torch.autograd.Function):
    """nn.ReLU of the ViLBERT poolers (vilbert.py:803,818)."""

    @staticmethod
    def forward(ctx, x):
        x2 = _as_bf16_2d(x)
        y = torch.empty_like(x2)
        nat.eltwise(1, x2, None, y)
        ctx.save_for_backward(y)
        return y.view(x.shape)

    @staticmethod
    def backward(ctx, g):
        (y,) = ctx.saved_tensors
        g2 = _grad_bf16(g, g.shape[-1])
        d = torch.empty_like(y)
        nat.eltwise(2, g2, y, d)
        return d.view(g.shape)


# ---------------------------------------------------------------------------------------------
# UNITER pieces (mmf/models/uniter.py)
# ---------------------------------------------------------------------------------------------
class FeatureTableAddFn(torch.autograd.Function):
    """img_feat + mask_embedding(img_masks) (uniter.py:74-78): fp32 region features plus a table row picked per region, written
    once as bf16 (the conversion the feature GEMM would do anyway).  `padding_idx` rows of the table receive no gradient."""

    @staticmethod
    def forward(ctx, feats, idx, table, padding_idx):
        B, R, D = feats.shape
        f2 = feats.reshape(B * R, D).float().contiguous()
        ix = idx.reshape(B * R).contiguous() if idx is not None else None
        y = torch.empty(B * R, D, dtype=BF16, device=table.device)
        nat.rows_add_table_f32(f2, ix, table.detach() if ix is not None else None, y, B * R, D)
        ctx.save_for_backward(ix)
        ctx.meta = (B, R, D, table.shape[0], padding_idx)
        return y.view(B, R, D)

    @staticmethod
    def backward(ctx, g):
        (ix,) = ctx.saved_tensors
        B, R, D, NT, pad = ctx.meta
        if ix is None:
            return None, None, None, None
        g2 = _grad_bf16(g, D)
        dt = torch.zeros(NT, D, dtype=F32, device=g2.device)
        nat.rows_scatter_add(g2, D, B, R, R, ix.view(B, R), R, 0, 0, dt, D, 1)     # deterministic two-bucket column sums
        if pad is not None:
            dt[pad].zero_()
        return None, None, dt, None


class AddFn(torch.autograd.Function):
    """a + b on bf16 activations (transformed_im + transformed_pos, uniter.py:82)."""

    @staticmethod
    def forward(ctx, a, b):
        a2, b2 = _as_bf16_2d(a), _as_bf16_2d(b)
        out = torch.empty_like(a2)
        nat.eltwise(3, a2, b2, out)
        return out.view(a.shape)

    @staticmethod
    def backward(ctx, g):
        return g, g


class SmallKLinearFn(torch.autograd.Function):
    """nn.Linear over a handful of input features (UNITER's 7-d box geometry, uniter.py:64,81): the operand is zero-padded to a
    multiple of 8 columns so that its rows are 16-byte aligned for the GEMM loader.  The input gets no gradient."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        K = x.shape[-1]
        KP = _pad8(K)
        N = weight.shape[0]
        x2 = x.reshape(-1, K).float().contiguous()
        M = x2.shape[0]
        dev = weight.device
        x8 = torch.empty(M, KP, dtype=BF16, device=dev)
        nat.cast2d_f32_to_bf16(x2, K, x8, KP, M, K)
        w8 = torch.empty(N, KP, dtype=BF16, device=dev)
        nat.cast2d_f32_to_bf16(weight.detach().contiguous(), K, w8, KP, N, K)
        y = torch.empty(M, N, dtype=BF16, device=dev)
        nat.gemm(x8, w8, y, M, N, KP, KP, KP, N, bias=bias.detach())
        ctx.save_for_backward(x8)
        ctx.meta = (x.shape, M, N, K, KP)
        return y.view(*x.shape[:-1], N)

    @staticmethod
    def backward(ctx, g):
        (x8,) = ctx.saved_tensors
        xshape, M, N, K, KP = ctx.meta
        dy = _grad_bf16(g, N)
        dw8 = torch.empty(N, KP, dtype=F32, device=dy.device)
        nat.gemm(dy, x8, dw8, N, KP, M, N, KP, KP, a_kmajor=True, b_kmajor=True)
        return None, dw8[:, :K].contiguous(), _colsum(dy, N, M, N)


class PairHalvesFn(torch.autograd.Function):
    """nlvr2: pooled outputs of the two images of a sample side by side, [2B, H] -> [B, 2H] =
    cat(x[:B], x[B:], dim=1) (visual_bert.py:369-374, vilbert.py:1322-1323 equivalent), as two strided row copies."""

    @staticmethod
    def forward(ctx, x):
        x2 = _as_bf16_2d(x)
        B2, H = x2.shape
        B = B2 // 2
        out = torch.empty(B, 2 * H, dtype=BF16, device=x2.device)
        o2 = out.view(2 * B, H)
        nat.copy_rows(x2, 1, o2, 2, B, 1, H)              # first image  -> columns [0, H)
        nat.copy_rows(x2[B:], 1, o2[1:], 2, B, 1, H)      # second image -> columns [H, 2H)
        ctx.meta = (B, H)
        return out

    @staticmethod
    def backward(ctx, g):
        B, H = ctx.meta
        g2 = _grad_bf16(g, 2 * H).view(2 * B, H)
        dx = torch.empty(2 * B, H, dtype=BF16, device=g2.device)
        nat.copy_rows(g2, 2, dx, 1, B, 1, H)
        nat.copy_rows(g2[1:], 2, dx[B:], 1, B, 1, H)
        return dx


# ---------------------------------------------------------------------------------------------
# M4C pieces (mmf/models/m4c.py)
# ---------------------------------------------------------------------------------------------
class L2NormRowsFn(torch.autograd.Function):
    """F.normalize(x, dim=-1) (m4c.py:195) -> bf16.  x: fp32 input features or a bf16 activation."""

    @staticmethod
    def forward(ctx, x):
        D = x.shape[-1]
        x2 = x.reshape(-1, D)
        if x2.dtype not in (F32, BF16):
            x2 = x2.float()
        x2 = x2.contiguous()
        rows = x2.shape[0]
        y = torch.empty(rows, D, dtype=BF16, device=x2.device)
        inv = torch.empty(rows, dtype=F32, device=x2.device)
        nat.l2norm_rows_fwd(x2, D, y, D, inv, rows, D)
        ctx.save_for_backward(y, inv)
        ctx.xshape = x.shape
        return y.view(x.shape)

    @staticmethod
    def backward(ctx, g):
        y, inv = ctx.saved_tensors
        rows, D = y.shape
        g2 = _grad_bf16(g, D)
        dx = torch.empty(rows, D, dtype=BF16, device=y.device)
        nat.l2norm_rows_bwd(g2, D, y, D, inv, dx, D, rows, D)
        return dx.view(ctx.xshape)


class OcrFeatureConcatFn(torch.autograd.Function):
    """cat([normalize(fasttext), normalize(phoc), normalize(fc7), zeros(order vectors)], -1) of m4c.py:211-237 written once
    as bf16 rows padded to a multiple of 8 columns (16-byte rows for the GEMM loader).  Only the appearance feature
    (`fc7`, bf16 activation) carries a gradient; the two text features are inputs."""

    @staticmethod
    def forward(ctx, fasttext, phoc, fc7, order_dim):
        B, N, _ = fasttext.shape
        rows = B * N
        d0, d1, d2 = fasttext.shape[-1], phoc.shape[-1], fc7.shape[-1]
        K = d0 + d1 + d2 + int(order_dim)
        KP = _pad8(K)
        dev = fc7.device
        out = torch.zeros(rows, KP, dtype=BF16, device=dev)
        inv = torch.empty(3, rows, dtype=F32, device=dev)
        f0 = fasttext.reshape(rows, d0).float().contiguous()
        f1 = phoc.reshape(rows, d1).float().contiguous()
        f2 = _as_bf16_2d(fc7)
        nat.l2norm_rows_fwd(f0, d0, out, KP, inv[0], rows, d0)
        nat.l2norm_rows_fwd(f1, d1, out[:, d0:], KP, inv[1], rows, d1)
        nat.l2norm_rows_fwd(f2, d2, out[:, d0 + d1:], KP, inv[2], rows, d2)
        ctx.save_for_backward(out, inv)
        ctx.meta = (B, N, d0 + d1, d2, KP)
        return out.view(B, N, KP)

    @staticmethod
    def backward(ctx, g):
        out, inv = ctx.saved_tensors
        B, N, off, d2, KP = ctx.meta
        rows = B * N
        g2 = _grad_bf16(g, KP)
        dx = torch.empty(rows, d2, dtype=BF16, device=out.device)
        nat.l2norm_rows_bwd(g2[:, off:], KP, out[:, off:], KP, inv[2], dx, d2, rows, d2)
        return None, None, dx.view(B, N, d2), None


class PaddedLinearFn(torch.autograd.Function):
    """nn.Linear whose input width K is not a multiple of 8 (M4C's 3002-wide OCR feature, m4c.py:243): `x` arrives as bf16
    rows already zero-padded to KP = round_up(K, 8) columns (OcrFeatureConcatFn); the weight is padded the same way while
    it is converted to bf16."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        N, K = weight.shape
        KP = x.shape[-1]
        if KP != _pad8(K):
            raise ValueError("PaddedLinearFn: input has %d columns, expected round_up(%d, 8)" % (KP, K))
        x2 = _as_bf16_2d(x)
        M = x2.shape[0]
        dev = weight.device
        w8 = torch.empty(N, KP, dtype=BF16, device=dev)
        nat.cast2d_f32_to_bf16(weight.detach().contiguous(), K, w8, KP, N, K)
        y = torch.empty(M, N, dtype=BF16, device=dev)
        nat.gemm(x2, w8, y, M, N, KP, KP, KP, N, bias=bias.detach())
        ctx.save_for_backward(x2, w8)
        ctx.meta = (x.shape, M, N, K, KP)
        return y.view(*x.shape[:-1], N)

    @staticmethod
    def backward(ctx, g):
        x2, w8 = ctx.saved_tensors
        xshape, M, N, K, KP = ctx.meta
        dy = _grad_bf16(g, N)
        dx, dw8, db = _linear_bwd(dy, N, x2, w8, M, N, KP, need_dx=ctx.needs_input_grad[0], want_db=True)
        return (dx.view(xshape) if dx is not None else None), dw8[:, :K].contiguous(), db


class ParamRowsFn(torch.autograd.Function):
    """A weight matrix used as an activation: `fixed_ans_emb = self.classifier.module.weight` (m4c.py:268).  Forward hands
    out the bf16 shadow the GEMMs already keep; backward converts the row gradients to the parameter's fp32."""

    @staticmethod
    def forward(ctx, weight):
        w16 = shadows.get(weight)
        return w16.view(w16.shape)

    @staticmethod
    def backward(ctx, g):
        g2 = _grad_bf16(g, g.shape[-1])
        d = torch.empty(g2.shape, dtype=F32, device=g2.device)
        nat.cast_bf16_to_f32(g2, d)
        return d


class PrevPredGatherFn(torch.autograd.Function):
    """_batch_gather(cat([ans_emb.expand(B), ocr_emb], 1), prev_inds) of PrevPredEmbeddings.forward (m4c.py:526-528) without
    the [B, V + N, H] concatenation: one two-source row gather; backward scatter-adds (repeated indices, e.g. <pad>,
    collide) into an fp32 [V + B*N, H] buffer that splits into the two gradients."""

    @staticmethod
    def forward(ctx, ans, ocr, prev_inds):
        V, H = ans.shape
        B, N, _ = ocr.shape
        T = prev_inds.shape[1]
        a2, o2 = _as_bf16_2d(ans), _as_bf16_2d(ocr)
        batch = torch.arange(B, device=prev_inds.device, dtype=torch.int64).unsqueeze(1) * N
        flat = torch.where(prev_inds < V, prev_inds, prev_inds + batch).contiguous()     # OCR row (b, i) -> V + b*N + i
        out = torch.empty(B * T, H, dtype=BF16, device=a2.device)
        nat.gather_rows2(a2, o2, flat, out, B * T, H)
        ctx.save_for_backward(flat)
        ctx.meta = (V, B, N, T, H)
        return out.view(B, T, H)

    @staticmethod
    def backward(ctx, g):
        (flat,) = ctx.saved_tensors
        V, B, N, T, H = ctx.meta
        g2 = _grad_bf16(g, H)
        d = torch.zeros(V + B * N, H, dtype=F32, device=g2.device)
        nat.rows_scatter_add(g2, H, B, T, T, flat, T, 0, 0, d, H, 0)
        d16 = torch.empty(V + B * N, H, dtype=BF16, device=g2.device)
        nat.cast_f32_to_bf16(d, d16)
        return d16[:V], d16[V:].view(B, N, H), None


class M4CScoresFn(torch.autograd.Function):
    """M4C._forward_output (m4c.py:275-283): fixed-vocabulary scores `classifier(dec)`, the OCR pointer scores
    `OcrPtrNet(dec, ocr, ocr_mask)` (:474-493) and their concatenation, written by the two producers straight into one
    fp32 [B, T, V + N] buffer (the classifier GEMM with ldc = V + N, the pointer kernel at column V)."""

    @staticmethod
    def forward(ctx, dec, ocr, cls_w, cls_b, q_w, q_b, k_w, k_b, ocr_mask_add, cls_w16, q_w16, k_w16):
        B, T, H = dec.shape
        N = ocr.shape[1]
        V, HQ = cls_w.shape[0], q_w.shape[0]
        dec2, ocr2 = _as_bf16_2d(dec), _as_bf16_2d(ocr)
        dev = dec2.device
        out = torch.empty(B * T, V + N, dtype=F32, device=dev)
        nat.gemm(dec2, cls_w16, out, B * T, V, H, H, H, V + N, bias=cls_b.detach())
        q = torch.empty(B * T, HQ, dtype=BF16, device=dev)
        k = torch.empty(B * N, HQ, dtype=BF16, device=dev)
        nat.gemm(dec2, q_w16, q, B * T, HQ, H, H, H, HQ, bias=q_b.detach())
        nat.gemm(ocr2, k_w16, k, B * N, HQ, H, H, H, HQ, bias=k_b.detach())
        scale = 1.0 / math.sqrt(HQ)
        nat.ptr_scores_fwd(q, k, ocr_mask_add, out[:, V:], V + N, B, T, N, HQ, scale)
        ctx.save_for_backward(dec2, ocr2, q, k, cls_w16, q_w16, k_w16)
        ctx.meta = (B, T, N, H, V, HQ, scale)
        return out.view(B, T, V + N)

    @staticmethod
    def backward(ctx, g):
        dec2, ocr2, q, k, cls_w16, q_w16, k_w16 = ctx.saved_tensors
        B, T, N, H, V, HQ, scale = ctx.meta
        M = B * T
        dev = dec2.device
        g2 = g.reshape(M, V + N)
        if g2.dtype != F32:
            g2 = g2.float()
        g2 = g2.contiguous()
        ldv = _pad8(V)
        dfix = torch.empty(M, ldv, dtype=BF16, device=dev)
        nat.cast2d_f32_to_bf16(g2, V + N, dfix, ldv, M, V)
        ddec, dcw, dcb = _linear_bwd(dfix, ldv, dec2, cls_w16, M, V, H, want_db=True)
        dq = torch.empty(M, HQ, dtype=BF16, device=dev)
        dk = torch.empty(B * N, HQ, dtype=BF16, device=dev)
        nat.ptr_scores_bwd(g2[:, V:], V + N, q, k, dq, dk, B, T, N, HQ, scale)
        ddec, dqw, dqb = _linear_bwd(dq, HQ, dec2, q_w16, M, HQ, H, dx_resid=ddec, want_db=True)
        docr, dkw, dkb = _linear_bwd(dk, HQ, ocr2, k_w16, B * N, HQ, H, want_db=True)
        return ddec.view(B, T, H), docr.view(B, N, H), dcw, dcb, dqw, dqb, dkw, dkb, None, None, None, None


class DecodingBCEWithMaskFn(torch.autograd.Function):
    """M4CDecodingBCEWithMaskLoss.forward (mmf/modules/losses.py:581-592): sum(BCEWithLogits(scores, targets) * loss_mask) /
    max(sum(loss_mask), 1).  Returns a 1-element tensor like the reference."""

    @staticmethod
    def forward(ctx, scores, targets, loss_mask):
        assert scores.dim() == 3 and loss_mask.dim() == 2
        B, T, Cn = scores.shape
        s = scores.float().contiguous()
        t = targets.float().contiguous()
        w = loss_mask.float().contiguous()
        loss = torch.empty(1, dtype=F32, device=s.device)
        count = torch.empty(1, dtype=F32, device=s.device)
        nat.bce_rowmask_fwd(s, t, w, loss, count, B * T, Cn)
        ctx.save_for_backward(s, t, w, count)
        return loss

    @staticmethod
    def backward(ctx, g):
        s, t, w, count = ctx.saved_tensors
        B, T, Cn = s.shape
        d = torch.empty(B, T, Cn, dtype=F32, device=s.device)
        nat.bce_rowmask_bwd(s, t, w, count, g.float().reshape(1).contiguous(), d, B * T, Cn)
        return d, None, None


class SplitRowsFn(torch.autograd.Function):
    """The slices `mmt_seq_output[:, a:b]` of MMT.forward (m4c.py:446-449) as strided HIP copies: [B, S, H] -> one
    contiguous [B, L_i, H] block per entry of `lens` (sum(lens) == S).  Backward writes the block gradients back into
    one [B, S, H] buffer; blocks nobody used stay zero."""

    @staticmethod
    def forward(ctx, x, lens):
        B, S, H = x.shape
        lens = [int(l) for l in lens]
        assert sum(lens) == S
        x2 = _as_bf16_2d(x)
        outs, off = [], 0
        for L in lens:
            d = torch.empty(B * L, H, dtype=BF16, device=x2.device)
            nat.copy_rows(x2[off:], S, d, L, B, L, H)
            outs.append(d.view(B, L, H))
            off += L
        ctx.meta = (B, S, H, lens)
        ctx.set_materialize_grads(False)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gs):
        B, S, H, lens = ctx.meta
        dev = next(g.device for g in gs if g is not None)
        dx = torch.zeros(B * S, H, dtype=BF16, device=dev)
        off = 0
        for g, L in zip(gs, lens):
            if g is not None:
                nat.copy_rows(_grad_bf16(g, H), L, dx[off:], S, B, L, H)
            off += L
        return dx.view(B, S, H), None
