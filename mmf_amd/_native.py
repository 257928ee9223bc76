"""ctypes binding of libmmf_amd.so (the gfx950 kernels + C ABI declared in include/mmf_amd.h).

This is the only place Python touches the native library.  There is NO fallback: if the shared
library is missing or a call fails, an exception is raised — the product path never silently
degrades to eager PyTorch or to the CPU oracle.

PyTorch is used here purely as plumbing: device memory (`tensor.data_ptr()`), the current HIP
stream, and dtype bookkeeping.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# MMF_AMD_LIB: an instrumented development build of the same library (python -m mmf_amd.csrc.build --tag probe), never a fallback
LIB_PATH = os.environ.get("MMF_AMD_LIB") or os.path.join(_HERE, "libmmf_amd.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "mmf_amd.h")


class NativeLibraryError(RuntimeError):
    pass


class GemmDesc(C.Structure):
    _fields_ = [
        ("A", C.c_void_p), ("B", C.c_void_p), ("C", C.c_void_p),
        ("M", C.c_int), ("N", C.c_int), ("K", C.c_int),
        ("lda", C.c_int), ("ldb", C.c_int), ("ldc", C.c_int),
        ("a_kmajor", C.c_int), ("b_kmajor", C.c_int),
        ("a_f32", C.c_int), ("b_f32", C.c_int), ("out_f32", C.c_int),
        ("beta", C.c_float),
        ("bias", C.c_void_p), ("coladd", C.c_void_p),
        ("rowtab", C.c_void_p), ("rowidx", C.c_void_p), ("rowtab_ld", C.c_int),
        ("act", C.c_int), ("U", C.c_void_p), ("aux", C.c_void_p),
        ("resid", C.c_void_p), ("ldr", C.c_int),
        ("drop_key", C.c_uint32), ("drop_thr16", C.c_uint32), ("drop_scale", C.c_float), ("drop_seed", C.c_void_p),
        ("grp_in", C.c_int), ("grp_pad", C.c_int), ("grp_off", C.c_int),
        ("splitk_ws", C.c_void_p), ("splitk_ws_bytes", C.c_int64), ("rowsum_out", C.c_void_p), ("debug_flags", C.c_int),
    ]


class AttnDesc(C.Structure):
    _fields_ = [
        ("q", C.c_void_p), ("k", C.c_void_p), ("v", C.c_void_p),
        ("ldq", C.c_int), ("ldk", C.c_int), ("ldv", C.c_int),
        ("mask", C.c_void_p), ("ctx", C.c_void_p), ("ldo", C.c_int), ("lse", C.c_void_p),
        ("B", C.c_int), ("heads", C.c_int), ("Sq", C.c_int), ("Sk", C.c_int),
        ("scale", C.c_float),
        ("drop_key", C.c_uint32), ("drop_thr16", C.c_uint32), ("drop_scale", C.c_float), ("drop_seed", C.c_void_p),
        ("head_dim", C.c_int), ("ctx_f32", C.c_void_p), ("causal_tail", C.c_int),
        ("q_batch_rows", C.c_int), ("kv_batch_rows", C.c_int), ("mask_batch_stride", C.c_int), ("mask_query_stride", C.c_int),
        ("keep_bits", C.c_void_p), ("keep_lanes", C.c_void_p), ("mask_head_stride", C.c_int),
    ]


ATTN_DRAW_MAX = 48


class AttnDrawSite(C.Structure):
    _fields_ = [
        ("drop_key", C.c_uint32), ("drop_thr16", C.c_uint32), ("drop_seed", C.c_void_p),
        ("B", C.c_int), ("heads", C.c_int), ("Sq", C.c_int), ("Sk", C.c_int), ("head_dim", C.c_int),
        ("keep_bits", C.c_void_p), ("keep_lanes", C.c_void_p),
    ]


MT_MAX = 52


class AdamWMultiDesc(C.Structure):
    _fields_ = [
        ("n", C.c_int),
        ("p", C.c_void_p * MT_MAX), ("g", C.c_void_p * MT_MAX), ("m", C.c_void_p * MT_MAX), ("v", C.c_void_p * MT_MAX),
        ("p16", C.c_void_p * MT_MAX), ("p32", C.c_void_p * MT_MAX), ("numel", C.c_int64 * MT_MAX), ("lr", C.c_float * MT_MAX), ("wd", C.c_float * MT_MAX),
        ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float),
        ("step", C.c_int), ("correct_bias", C.c_int), ("mode", C.c_int),
        ("grad_scale", C.c_float), ("norm_sq", C.c_void_p), ("max_norm", C.c_float), ("dev_state", C.c_void_p),
        ("g_bf16_mask", C.c_uint64),
    ]


class WraDesc(C.Structure):
    _fields_ = [("seq", C.c_void_p), ("ld", C.c_int), ("B", C.c_int), ("S", C.c_int), ("H", C.c_int), ("M", C.c_int), ("N", C.c_int),
                ("txt_pad", C.c_void_p), ("img_pad", C.c_void_p), ("label", C.c_void_p), ("xinv", C.c_void_p), ("yinv", C.c_void_p),
                ("plan", C.c_void_p), ("cost", C.c_void_p), ("dist", C.c_void_p), ("beta", C.c_float), ("eps", C.c_float), ("iterations", C.c_int)]


class LnReduceList(C.Structure):
    _fields_ = [("n", C.c_int), ("partials", C.c_void_p * MT_MAX), ("rows", C.c_int * MT_MAX), ("H", C.c_int * MT_MAX),
                ("dgamma", C.c_void_p * MT_MAX), ("dbeta", C.c_void_p * MT_MAX)]


class TensorList(C.Structure):
    _fields_ = [("n", C.c_int), ("ptr", C.c_void_p * MT_MAX), ("numel", C.c_int64 * MT_MAX)]


class TransposeList(C.Structure):
    _fields_ = [("n", C.c_int), ("src", C.c_void_p * MT_MAX), ("dst", C.c_void_p * MT_MAX), ("rows", C.c_int * MT_MAX),
                ("cols", C.c_int * MT_MAX)]


class AttnBwdDesc(C.Structure):
    _fields_ = [
        ("f", AttnDesc), ("dctx", C.c_void_p), ("dq", C.c_void_p), ("dk", C.c_void_p), ("dv", C.c_void_p),
        ("delta", C.c_void_p),
    ]


_lib = None


def lib():
    """Load (once) and return the native library; raise loudly when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NativeLibraryError(
            "mmf_amd native library not found at %s. Build it with `python -m mmf_amd.csrc.build` "
            "(or `python -c 'import __graft_entry__ as g; g.build()'`). There is no fallback path." % LIB_PATH
        )
    try:
        L = C.CDLL(LIB_PATH)
    except OSError as e:  # pragma: no cover - environment specific
        raise NativeLibraryError("failed to load %s: %s" % (LIB_PATH, e))
    L.mmf_amd_last_error.restype = C.c_char_p
    L.mmf_amd_target.restype = C.c_char_p
    L.mmf_amd_abi_version.restype = C.c_int
    if L.mmf_amd_abi_version() != 1:
        raise NativeLibraryError("ABI version mismatch: library %d, binding 1" % L.mmf_amd_abi_version())
    _lib = L
    if os.environ.get("MMF_AMD_TUN"):            # generic A/B switch: "id:value,id:value" (include/mmf_amd.h MMF_TUN_*)
        for kv in os.environ["MMF_AMD_TUN"].split(","):
            k, v = kv.split(":")
            L.mmf_amd_set_tunable(int(k), int(v, 0))
    return L


def _check(rc, what):
    if rc != 0:
        raise NativeLibraryError("%s failed (rc=%d): %s" % (what, rc, lib().mmf_amd_last_error().decode()))


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    """Device pointer of a tensor (or None)."""
    if t is None:
        return None
    return C.c_void_p(t.data_ptr())


def _req(t, dtype, name):
    if t is None:
        return
    if not t.is_cuda:
        raise NativeLibraryError("%s must live in HBM (cuda/hip tensor), got %s" % (name, t.device))
    if t.dtype != dtype:
        raise NativeLibraryError("%s must be %s, got %s" % (name, dtype, t.dtype))


NO_DROP = (0, 0, 1.0, None)


def drop_cfg(p, key, seed=None):
    """(key, thr16, scale, seed_tensor) for dropout probability p; thr16 == 0 disables dropout.  `seed` is an
    optional 1-element int32 device tensor mixed into the key at run time (hipGraph replays)."""
    if p is None or p <= 0.0:
        return NO_DROP
    thr = int(round(p * 65536.0))
    thr = max(1, min(thr, 65535))
    return int(key) & 0xFFFFFFFF, thr, 1.0 / (1.0 - thr / 65536.0), seed


def _drop4(drop):
    if len(drop) == 3:
        return drop[0], drop[1], drop[2], None
    return drop[0], drop[1], drop[2], _p(drop[3])


# --------------------------------------------------------------------------------------------
# GEMM
# --------------------------------------------------------------------------------------------
def _gemm_desc(A, B, C_out, M, N, K, lda, ldb, ldc, a_kmajor=False, b_kmajor=False, beta=0.0, bias=None, coladd=None,
               rowtab=None, rowidx=None, rowtab_ld=0, act=0, U=None, aux=None, resid=None, ldr=0, drop=NO_DROP,
               grp=(0, 0, 0), debug_flags=0, rowsum_out=None, d=None):
    d = GemmDesc() if d is None else d
    d.debug_flags = int(debug_flags)
    d.A, d.B, d.C = _p(A), _p(B), _p(C_out)
    d.M, d.N, d.K = M, N, K
    d.lda, d.ldb, d.ldc = lda, ldb, ldc
    d.a_kmajor, d.b_kmajor = int(a_kmajor), int(b_kmajor)
    d.a_f32 = int(A.dtype == torch.float32)
    d.b_f32 = int(B.dtype == torch.float32)
    d.out_f32 = int(C_out.dtype == torch.float32)
    for t, n in ((A, "A"), (B, "B")):
        if t.dtype not in (torch.bfloat16, torch.float32):
            raise NativeLibraryError("gemm operand %s must be bf16 or fp32" % n)
        if not t.is_cuda:
            raise NativeLibraryError("gemm operand %s must live in HBM (cuda/hip tensor), got %s" % (n, t.device))
    if C_out.dtype not in (torch.bfloat16, torch.float32):
        raise NativeLibraryError("gemm output must be bf16 or fp32")
    _req(bias, torch.float32, "bias"); _req(coladd, torch.float32, "coladd"); _req(rowtab, torch.float32, "rowtab")
    _req(rowidx, torch.int64, "rowidx"); _req(U, torch.bfloat16, "U"); _req(aux, torch.bfloat16, "aux")
    _req(resid, torch.bfloat16, "resid")
    d.beta = beta
    d.bias, d.coladd, d.rowtab, d.rowidx, d.rowtab_ld = _p(bias), _p(coladd), _p(rowtab), _p(rowidx), rowtab_ld
    d.act, d.U, d.aux = act, _p(U), _p(aux)
    d.resid, d.ldr = _p(resid), ldr
    d.drop_key, d.drop_thr16, d.drop_scale, d.drop_seed = _drop4(drop)
    d.grp_in, d.grp_pad, d.grp_off = grp
    if rowsum_out is not None:
        _req(rowsum_out, torch.float32, "rowsum_out")
        d.rowsum_out = _p(rowsum_out)
    return d


def gemm(A, B, C_out, M, N, K, lda, ldb, ldc, a_kmajor=False, b_kmajor=False, beta=0.0, bias=None, coladd=None,
         rowtab=None, rowidx=None, rowtab_ld=0, act=0, U=None, aux=None, resid=None, ldr=0, drop=NO_DROP,
         grp=(0, 0, 0), debug_flags=0, rowsum_out=None):
    """`rowsum_out` (fp32 [M], weight-gradient form): also returns sum_k A[k][m], the bias gradient (with or without split-K)."""
    d = _gemm_desc(A, B, C_out, M, N, K, lda, ldb, ldc, a_kmajor, b_kmajor, beta, bias, coladd, rowtab, rowidx, rowtab_ld, act, U,
                   aux, resid, ldr, drop, grp, debug_flags, rowsum_out)
    if d.out_f32 and a_kmajor and b_kmajor and bias is None and resid is None and act == 0:
        sp = lib().mmf_gemm_splitk_splits(M, N, K)
        if sp > 1:
            ws = torch.empty(sp * M * (N + 1), dtype=torch.float32, device=C_out.device)
            d.splitk_ws, d.splitk_ws_bytes = _p(ws), ws.numel() * 4
    elif rowsum_out is None and not (debug_flags & (1 << 18)):      # skinny problems (the heads): K-slices over the chip, epilogue on the slab sums
        sp = lib().mmf_gemm_skinny_splits(M, N, K, 1 if a_kmajor else 0)
        if sp > 1:
            ws = torch.empty(sp * M * ((N + 7) // 8 * 8), dtype=torch.float32, device=C_out.device)
            d.splitk_ws, d.splitk_ws_bytes = _p(ws), ws.numel() * 4
    _check(lib().mmf_gemm_bf16(C.byref(d), _stream()), "mmf_gemm_bf16")


GEMM_GROUP_MAX = 8
GEMM_NO_SKINNY = 1 << 18     # mmf_gemm_desc.debug_flags: never the skinny split-K path (M <= 64, K >= 1536), whatever workspace is offered


def gemm_grouped(problems):
    """Several GEMMs of one operand layout in ONE launch (mmf_gemm_bf16_grouped): `problems` is a list of dicts holding the
    arguments of `gemm` (positional ones under their names A, B, C_out, M, N, K, lda, ldb, ldc).  No split-K; every problem
    keeps its own epilogue and `rowsum_out`."""
    n = len(problems)
    if not 1 <= n <= GEMM_GROUP_MAX:
        raise NativeLibraryError("gemm_grouped takes 1..%d problems, got %d" % (GEMM_GROUP_MAX, n))
    arr = (GemmDesc * n)()
    for i, kw in enumerate(problems):
        _gemm_desc(d=arr[i], **kw)
    _check(lib().mmf_gemm_bf16_grouped(arr, n, _stream()), "mmf_gemm_bf16_grouped")


class LnBwdDesc(C.Structure):
    """include/mmf_amd.h mmf_ln_bwd_desc"""
    _fields_ = [("dy", C.c_void_p), ("x", C.c_void_p), ("mean", C.c_void_p), ("rstd", C.c_void_p), ("gamma", C.c_void_p), ("dx", C.c_void_p),
                ("dlin", C.c_void_p), ("drop_key", C.c_uint32), ("drop_thr16", C.c_uint32), ("drop_scale", C.c_float), ("drop_seed", C.c_void_p),
                ("partials", C.c_void_p), ("rows", C.c_int), ("H", C.c_int)]


def gemm_grouped_ln(problems, dy, x, mean, rstd, gamma, dx, dlin, drop, partials, rows, H):
    """`gemm_grouped(problems)` and the deferred LayerNorm backward `layernorm_bwd(dy, x, ..., None, None, None, 0, partials, rows, H)` that does not depend on
    it, in ONE launch where the tiles leave CUs idle (mmf_gemm_bf16_grouped_ln); otherwise one after the other.  Same bits either way."""
    n = len(problems)
    if not 1 <= n <= GEMM_GROUP_MAX:
        raise NativeLibraryError("gemm_grouped_ln takes 1..%d problems, got %d" % (GEMM_GROUP_MAX, n))
    arr = (GemmDesc * n)()
    for i, kw in enumerate(problems):
        _gemm_desc(d=arr[i], **kw)
    for t, nm in ((dy, "dy"), (x, "x"), (dx, "dx"), (dlin, "dlin")):
        _req(t, torch.bfloat16, nm)
    for t, nm in ((mean, "mean"), (rstd, "rstd"), (gamma, "gamma"), (partials, "partials")):
        _req(t, torch.float32, nm)
    k, t, sc, sd = _drop4(drop)
    ptr = lambda v: None if v is None else v.data_ptr()
    d = LnBwdDesc(ptr(dy), ptr(x), ptr(mean), ptr(rstd), ptr(gamma), ptr(dx), ptr(dlin), k, t, sc, None if len(drop) == 3 or drop[3] is None else drop[3].data_ptr(),
                  ptr(partials), int(rows), int(H))
    _check(lib().mmf_gemm_bf16_grouped_ln(arr, n, C.byref(d), _stream()), "mmf_gemm_bf16_grouped_ln")


# call-site tags of the encoder layer's GEMMs (include/mmf_amd.h MMF_SITE_*, MMF_GEMM_SITE): they name a call for the per-site store policy
SITE_QKV_FWD, SITE_ATTN_OUT_FWD, SITE_FFN_UP_FWD, SITE_FFN_DOWN_FWD, SITE_FFN_DOWN_DGRAD, SITE_FFN_UP_DGRAD, SITE_ATTN_OUT_DGRAD, SITE_QKV_DGRAD = range(1, 9)


def gemm_site(s):
    """`debug_flags` value that tags a GEMM call as site `s` (0: untagged)."""
    return (int(s) & 15) << 20


def gemm_last_kernel():
    """Family / tile of the kernel the last gemm / gemm_grouped call launched (bench.py labels its roofline with it)."""
    f = lib().mmf_gemm_last_kernel
    f.restype = C.c_char_p
    return f().decode()


def gemm_probe_log():
    """Host-side log of the launches probed since the last `gemm_set_probe(buf)`: list of (launch id, layout, M, N, K); layout bit 0 / 1 = A / B
    k-major, bit 2 = grouped launch (then M, N = problems, tiles)."""
    arr = (C.c_int64 * (5 * 4096))()
    f = lib().mmf_gemm_probe_log
    f.restype = C.c_int
    n = min(int(f(arr, C.c_int(4096))), 4096)
    return [tuple(int(arr[i * 5 + j]) for j in range(5)) for i in range(n)]


def gemm_set_probe(buf):
    """Development aid: `buf` = zeroed int64 device tensor of 8 * (1 + capacity) words (or None to switch the probe off);
    while set, every GEMM workgroup appends a timeline record (see gemm.hip Probe)."""
    if buf is None:
        _check(lib().mmf_gemm_set_probe(None, C.c_int64(0)), "mmf_gemm_set_probe")
    else:
        _req(buf, torch.int64, "probe buffer")
        _check(lib().mmf_gemm_set_probe(_p(buf), C.c_int64(buf.numel() // 8 - 1)), "mmf_gemm_set_probe")


def attention_set_probe(buf):
    """Development aid (library built with -DMMF_ATTN_PROBE only): zeroed int64 device tensor of 12 * (1 + capacity) words, or None."""
    if buf is None:
        _check(lib().mmf_attention_set_probe(None, C.c_int64(0)), "mmf_attention_set_probe")
    else:
        _req(buf, torch.int64, "probe buffer")
        _check(lib().mmf_attention_set_probe(_p(buf), C.c_int64(buf.numel() // 12 - 1)), "mmf_attention_set_probe")


def gemm_rowsum_supported(M, N, K):
    """True when a weight-gradient GEMM of this shape can carry the bias gradient (`rowsum_out`): always, since the row sums
    are written directly when the launch does not split K."""
    return True


# --------------------------------------------------------------------------------------------
# attention
# --------------------------------------------------------------------------------------------
def attention_keep_bits_words(B, heads, Sq, Sk, head_dim=64):
    """int32 words of the dropout keep-bit table the forward can hand to the backward for this shape (`keep_bits=`), 0 = these kernels take none."""
    f = lib().mmf_attention_keep_bits_words
    f.restype = C.c_int64
    return int(f(int(B), int(heads), int(Sq), int(Sk), int(head_dim)))


def attention_keep_lanes_words(B, heads, Sq, Sk, head_dim=64):
    """int32 words of the lane-major table `attention_draw_keep_bits` writes for the forward (`keep_lanes=`), 0 where `attention_keep_bits_words` is 0."""
    f = lib().mmf_attention_keep_lanes_words
    f.restype = C.c_int64
    return int(f(int(B), int(heads), int(Sq), int(Sk), int(head_dim)))


def attention_draw_keep_bits(sites, seed_offset=0):
    """The attention-dropout decisions of several attention calls in ONE launch, ahead of the kernels that use them.  `sites`: an iterable of
    (drop, B, heads, Sq, Sk, head_dim, keep_bits, keep_lanes) with `drop` as `drop_cfg` returns it and the two int32 tables sized by
    `attention_keep_bits_words` / `attention_keep_lanes_words`.  `seed_offset` is added to the seed word (1 = the next step's decisions)."""
    sites = list(sites)
    arr = (AttnDrawSite * max(1, len(sites)))()
    for d, (drop, B, heads, Sq, Sk, hd, kb, kl) in zip(arr, sites):
        _req(kb, torch.int32, "keep_bits"); _req(kl, torch.int32, "keep_lanes")
        d.drop_key, d.drop_thr16, _, d.drop_seed = _drop4(drop)
        d.B, d.heads, d.Sq, d.Sk, d.head_dim = int(B), int(heads), int(Sq), int(Sk), int(hd)
        d.keep_bits, d.keep_lanes = _p(kb), _p(kl)
    _check(lib().mmf_attention_draw_keep_bits(arr, len(sites), C.c_uint32(int(seed_offset)), _stream()), "mmf_attention_draw_keep_bits")


def _attn_desc(q, k, v, ldq, ldk, ldv, mask, ctx, ldo, lse, B, heads, Sq, Sk, scale, drop, head_dim=64, ctx_f32=None,
               causal_tail=0, q_batch_rows=0, kv_batch_rows=0, mask_batch_stride=0, keep_bits=None, keep_lanes=None):
    d = AttnDesc()
    _req(keep_bits, torch.int32, "keep_bits"); _req(keep_lanes, torch.int32, "keep_lanes")
    d.keep_bits, d.keep_lanes = _p(keep_bits), _p(keep_lanes)
    # a 3-D mask [B, Sq, Sk] is a materialised additive mask per (query, key) pair (mmf_attn_desc.mask_query_stride); 4-D [B, heads, Sq, Sk]: one such
    # mask per head (mmf_attn_desc.mask_head_stride); 2-D: the key mask [B, Sk]
    d.mask_query_stride, d.mask_head_stride = 0, 0
    if mask is not None and mask.dim() == 4:
        if tuple(mask.shape) != (B, heads, Sq, Sk) or mask.stride(3) != 1:
            raise NativeLibraryError("a per-head attention mask must be [B, heads, Sq, Sk] with contiguous rows, got %s" % (tuple(mask.shape),))
        d.mask_query_stride, d.mask_head_stride = int(mask.stride(2)), int(mask.stride(1))
        mask_batch_stride = int(mask.stride(0)) if (B > 1 and mask.stride(0) != heads * mask.stride(1)) else 0       # (0 = the dense default; a custom value is forward only)
    elif mask is not None and mask.dim() == 3:
        d.mask_query_stride = int(mask.stride(1))
    if d.mask_query_stride and not d.mask_head_stride and not mask_batch_stride:
        if tuple(mask.shape) != (B, Sq, Sk) or mask.stride(2) != 1:
            raise NativeLibraryError("a per-query attention mask must be [B, Sq, Sk] with contiguous rows, got %s" % (tuple(mask.shape),))
        mask_batch_stride = int(mask.stride(0)) if (B > 1 and mask.stride(0) != Sq * mask.stride(1)) else 0       # (the stride of a size-1 batch dimension is arbitrary)
    d.causal_tail = int(causal_tail)
    d.q_batch_rows, d.kv_batch_rows, d.mask_batch_stride = int(q_batch_rows), int(kv_batch_rows), int(mask_batch_stride)
    for t, n in ((q, "q"), (k, "k"), (v, "v"), (ctx, "ctx")):
        _req(t, torch.bfloat16, n)
    _req(mask, torch.float32, "mask"); _req(lse, torch.float32, "lse")
    d.q, d.k, d.v = _p(q), _p(k), _p(v)
    d.ldq, d.ldk, d.ldv = ldq, ldk, ldv
    d.mask, d.ctx, d.ldo, d.lse = _p(mask), _p(ctx), ldo, _p(lse)
    d.B, d.heads, d.Sq, d.Sk = B, heads, Sq, Sk
    d.scale = scale
    d.drop_key, d.drop_thr16, d.drop_scale, d.drop_seed = _drop4(drop)
    d.head_dim = head_dim
    _req(ctx_f32, torch.float32, "ctx_f32")
    d.ctx_f32 = _p(ctx_f32)
    return d


def attention_fwd(q, k, v, ldq, ldk, ldv, mask, ctx, ldo, lse, B, heads, Sq, Sk, scale, drop=NO_DROP, head_dim=64, ctx_f32=None,
                  causal_tail=0, q_batch_rows=0, kv_batch_rows=0, mask_batch_stride=0, keep_bits=None, keep_lanes=None):
    """`q_batch_rows` / `kv_batch_rows` / `mask_batch_stride` (forward only): q, k / v and the mask may live inside longer
    per-sample buffers (a K|V cache); 0 = the dense defaults Sq / Sk / Sk.  `keep_bits`: int32 [attention_keep_bits_words(...)], the forward
    writes its dropout decisions there for `attention_bwd(..., keep_bits=)` (which then does not hash them again).  `keep_lanes`: the decisions
    were drawn ahead by `attention_draw_keep_bits` (same drop, same shape): the forward reads them and writes no table."""
    d = _attn_desc(q, k, v, ldq, ldk, ldv, mask, ctx, ldo, lse, B, heads, Sq, Sk, scale, drop, head_dim, ctx_f32, causal_tail,
                   q_batch_rows, kv_batch_rows, mask_batch_stride, keep_bits, keep_lanes)
    _check(lib().mmf_attention_fwd(C.byref(d), _stream()), "mmf_attention_fwd")


def attention_bwd(q, k, v, ldq, ldk, ldv, mask, ctx, ldo, lse, B, heads, Sq, Sk, scale, dctx, dq, dk, dv, delta,
                  drop=NO_DROP, head_dim=64, ctx_f32=None, causal_tail=0, keep_bits=None):
    d = AttnBwdDesc()
    d.f = _attn_desc(q, k, v, ldq, ldk, ldv, mask, ctx, ldo, lse, B, heads, Sq, Sk, scale, drop, head_dim, ctx_f32, causal_tail, keep_bits=keep_bits)
    for t, n in ((dctx, "dctx"), (dq, "dq"), (dk, "dk"), (dv, "dv")):
        _req(t, torch.bfloat16, n)
    _req(delta, torch.float32, "delta")
    d.dctx, d.dq, d.dk, d.dv, d.delta = _p(dctx), _p(dq), _p(dk), _p(dv), _p(delta)
    _check(lib().mmf_attention_bwd(C.byref(d), _stream()), "mmf_attention_bwd")


# --------------------------------------------------------------------------------------------
# row kernels
# --------------------------------------------------------------------------------------------
def layernorm_fwd(x, gamma, beta, y, mean, rstd, rows, H, eps):
    _req(x, torch.bfloat16, "x"); _req(y, torch.bfloat16, "y")
    _req(gamma, torch.float32, "gamma"); _req(beta, torch.float32, "beta")
    _req(mean, torch.float32, "mean"); _req(rstd, torch.float32, "rstd")
    _check(lib().mmf_layernorm_fwd(_p(x), _p(gamma), _p(beta), _p(y), _p(mean), _p(rstd), rows, H, C.c_float(eps), _stream()),
           "mmf_layernorm_fwd")


def layernorm_dropout_fusable(H):
    return bool(lib().mmf_layernorm_dropout_fusable(int(H)))


def layernorm_dropout_fwd(x, gamma, beta, y, mean, rstd, rows, H, eps, drop):
    """y = dropout(LayerNorm(x)) in one launch (embeddings.py:343-345), bit-identical to layernorm_fwd + dropout."""
    _req(x, torch.bfloat16, "x"); _req(y, torch.bfloat16, "y")
    _req(gamma, torch.float32, "gamma"); _req(beta, torch.float32, "beta")
    _req(mean, torch.float32, "mean"); _req(rstd, torch.float32, "rstd")
    k, t, sc, sd = _drop4(drop)
    _check(lib().mmf_layernorm_dropout_fwd(_p(x), _p(gamma), _p(beta), _p(y), _p(mean), _p(rstd), rows, H, C.c_float(eps), C.c_uint32(k), C.c_uint32(t),
                                           C.c_float(sc), sd, _stream()), "mmf_layernorm_dropout_fwd")


def layernorm_bwd_din(dy, x, mean, rstd, gamma, dx, in_drop, dgamma, dbeta, accumulate, partials, rows, H):
    """layernorm_bwd of a LayerNorm whose OUTPUT went through dropout `in_drop`: the mask is applied to dy while it is loaded."""
    for t, n in ((dy, "dy"), (x, "x"), (dx, "dx")):
        _req(t, torch.bfloat16, n)
    for t, n in ((mean, "mean"), (rstd, "rstd"), (gamma, "gamma"), (dgamma, "dgamma"), (dbeta, "dbeta"), (partials, "partials")):
        _req(t, torch.float32, n)
    k, t, sc, sd = _drop4(in_drop)
    _check(lib().mmf_layernorm_bwd_din(_p(dy), _p(x), _p(mean), _p(rstd), _p(gamma), _p(dx), C.c_uint32(k), C.c_uint32(t), C.c_float(sc), sd, _p(dgamma),
                                       _p(dbeta), int(accumulate), _p(partials), rows, H, _stream()), "mmf_layernorm_bwd_din")


def layernorm_bwd_deferrable(rows, H):
    return bool(lib().mmf_layernorm_bwd_deferrable(int(rows), int(H)))


def layernorm_bwd_reduce_multi(items):
    """items: (partials, rows, H, dgamma, dbeta) of LayerNorm backwards launched with dgamma = dbeta = dbias = None."""
    for i0 in range(0, len(items), MT_MAX):
        chunk = items[i0:i0 + MT_MAX]
        d = LnReduceList()
        d.n = len(chunk)
        for i, (ws, rows, H, dg, db) in enumerate(chunk):
            for t, n in ((ws, "partials"), (dg, "dgamma"), (db, "dbeta")):
                _req(t, torch.float32, n)
            d.partials[i], d.rows[i], d.H[i], d.dgamma[i], d.dbeta[i] = ws.data_ptr(), int(rows), int(H), dg.data_ptr(), db.data_ptr()
        _check(lib().mmf_layernorm_bwd_reduce_multi(C.byref(d), _stream()), "mmf_layernorm_bwd_reduce_multi")


def layernorm_bwd_ws_floats(H):
    return lib().mmf_layernorm_bwd_ws_floats(H)


def layernorm_bwd(dy, x, mean, rstd, gamma, dx, dlin, drop, dgamma, dbeta, dbias, accumulate, partials, rows, H):
    for t, n in ((dy, "dy"), (x, "x"), (dx, "dx"), (dlin, "dlin")):
        _req(t, torch.bfloat16, n)
    for t, n in ((mean, "mean"), (rstd, "rstd"), (gamma, "gamma"), (dgamma, "dgamma"), (dbeta, "dbeta"), (dbias, "dbias"),
                 (partials, "partials")):
        _req(t, torch.float32, n)
    k, t, sc, sd = _drop4(drop)
    _check(lib().mmf_layernorm_bwd(_p(dy), _p(x), _p(mean), _p(rstd), _p(gamma), _p(dx), _p(dlin), C.c_uint32(k),
                                   C.c_uint32(t), C.c_float(sc), sd, _p(dgamma), _p(dbeta), _p(dbias),
                                   int(accumulate), _p(partials), rows, H, _stream()), "mmf_layernorm_bwd")


def embed_text_fwd(ids, seg, word, pos, typ, y, B, T, S, H, row0=0, pos0=0):
    _req(ids, torch.int64, "ids"); _req(seg, torch.int64, "seg"); _req(y, torch.bfloat16, "y")
    for t, n in ((word, "word"), (pos, "pos"), (typ, "type")):
        _req(t, torch.float32, n)
    _check(lib().mmf_embed_text_fwd(_p(ids), _p(seg), _p(word), _p(pos), _p(typ), _p(y), B, T, S, H, row0, pos0,
                                    int(word.shape[0]), int(pos.shape[0]), int(typ.shape[0]), _stream()), "mmf_embed_text_fwd")


def align_pos_fwd(align, pos, typ, typ_idx, out, rows, A, H):
    """out[r] = mean of pos[align[r, a]] over align != -1 (+ typ[typ_idx[r]]): the image_text_alignment position term, embeddings.py:373-397."""
    _req(align, torch.int64, "align"); _req(typ_idx, torch.int64, "typ_idx")
    for t, n in ((pos, "pos"), (typ, "typ"), (out, "out")):
        _req(t, torch.float32, n)
    _check(lib().mmf_align_pos_fwd(_p(align), _p(pos), _p(typ), _p(typ_idx), _p(out), rows, A, H, int(pos.shape[0]),
                                   0 if typ is None else int(typ.shape[0]), _stream()), "mmf_align_pos_fwd")


def align_pos_bwd(dvis, ld, nb, rpb, bstride, align, dpos, A, H):
    _req(dvis, torch.bfloat16, "dvis"); _req(align, torch.int64, "align"); _req(dpos, torch.float32, "dpos")
    _check(lib().mmf_align_pos_bwd(_p(dvis), ld, nb, rpb, bstride, _p(align), _p(dpos), A, H, int(dpos.shape[0]), _stream()), "mmf_align_pos_bwd")


def take_index_error():
    """True if an embedding gather / scatter-add met an index outside its table since the last call (the offending rows
    were skipped, nothing was read or written out of bounds); clears the flag.  Synchronises: call between steps."""
    rc = lib().mmf_amd_take_index_error()
    if rc < 0:
        _check(1, "mmf_amd_take_index_error")
    return bool(rc)


def rows_add_embed(x, seg, pos, typ, y, B, L, S, H, row0=0, pos0=0):
    _req(x, torch.bfloat16, "x"); _req(y, torch.bfloat16, "y"); _req(seg, torch.int64, "seg")
    _req(pos, torch.float32, "pos"); _req(typ, torch.float32, "type")
    _check(lib().mmf_rows_add_embed(_p(x), _p(seg), _p(pos), _p(typ), _p(y), B, L, S, H, row0, pos0, _stream()), "mmf_rows_add_embed")


def rows_add_table_f32(x, idx, table, y, rows, D):
    _req(x, torch.float32, "x"); _req(idx, torch.int64, "idx"); _req(table, torch.float32, "table"); _req(y, torch.bfloat16, "y")
    _check(lib().mmf_rows_add_table_f32(_p(x), _p(idx), _p(table), _p(y), rows, D, _stream()), "mmf_rows_add_table_f32")


def copy_rows(src, src_bstride, dst, dst_bstride, nb, rpb, H):
    _req(src, torch.bfloat16, "src"); _req(dst, torch.bfloat16, "dst")
    _check(lib().mmf_copy_rows_bf16(_p(src), src_bstride, _p(dst), dst_bstride, nb, rpb, H, _stream()), "mmf_copy_rows_bf16")


def rows_scatter_add(x, ld, nb, rpb, bstride, idx, idx_ld, per_pos, idx_base, out, H, few_buckets, skip_bucket=-1):
    _req(x, torch.bfloat16, "x"); _req(idx, torch.int64, "idx"); _req(out, torch.float32, "out")
    ws = None
    if few_buckets:
        ws = torch.empty(lib().mmf_rows_scatter_add_ws_floats(H), dtype=torch.float32, device=out.device)
    _check(lib().mmf_rows_scatter_add(_p(x), ld, nb, rpb, bstride, _p(idx), idx_ld, int(per_pos), idx_base, _p(out), H,
                                      int(few_buckets), int(out.shape[0]), _p(ws), int(-1 if skip_bucket is None else skip_bucket), _stream()),
           "mmf_rows_scatter_add")


def embed_tables_bwd(x, ld, B, T, R, seg, vt, pos0, dpos, dtyp, dtyp_vis, dpos_vis, H):
    """The four small table gradients of BertVisioLinguisticEmbeddings' backward in two launches (see mmf_embed_tables_bwd); outputs are added to."""
    _req(x, torch.bfloat16, "x"); _req(seg, torch.int64, "seg"); _req(vt, torch.int64, "vt")
    for t, n in ((dpos, "dpos"), (dtyp, "dtyp"), (dtyp_vis, "dtyp_vis"), (dpos_vis, "dpos_vis")):
        _req(t, torch.float32, n)
    ws = torch.empty(lib().mmf_embed_tables_bwd_ws_floats(T + R, H), dtype=torch.float32, device=x.device)
    rows = lambda t: 0 if t is None else int(t.shape[0])
    _check(lib().mmf_embed_tables_bwd(_p(x), ld, B, T, R, _p(seg), _p(vt), pos0, _p(dpos), rows(dpos), _p(dtyp), rows(dtyp), _p(dtyp_vis), rows(dtyp_vis),
                                      _p(dpos_vis), H, _p(ws), _stream()), "mmf_embed_tables_bwd")


def gate_sigmoid_fwd(z, gate, col0, B, Cn):
    """gate[:, col0:col0 + Cn] = 1 + sigmoid(z)  (ViLBERT dynamic_attention, vilbert.py:206-209); z fp32 [B, Cn], gate fp32 [B, ldg]."""
    _req(z, torch.float32, "z"); _req(gate, torch.float32, "gate")
    _check(lib().mmf_gate_sigmoid_fwd(_p(z), _p(gate), gate.stride(0), col0, B, Cn, _stream()), "mmf_gate_sigmoid_fwd")


def gate_sigmoid_bwd(dgate, gate, col0, dz, B, Cn):
    _req(dgate, torch.float32, "dgate"); _req(gate, torch.float32, "gate"); _req(dz, torch.float32, "dz")
    if dgate.stride(0) != gate.stride(0):
        raise NativeLibraryError("gate_sigmoid_bwd: dgate and gate must share their row stride")
    _check(lib().mmf_gate_sigmoid_bwd(_p(dgate), _p(gate), gate.stride(0), col0, _p(dz), B, Cn, _stream()), "mmf_gate_sigmoid_bwd")


def masked_mean_fwd(x, mask, pool, B, T, H):
    _req(x, torch.bfloat16, "x"); _req(mask, torch.float32, "mask"); _req(pool, torch.float32, "pool")
    _check(lib().mmf_masked_mean_fwd(_p(x), _p(mask), _p(pool), B, T, H, _stream()), "mmf_masked_mean_fwd")


def masked_mean_bwd(dpool, mask, dx, B, T, H):
    _req(dpool, torch.float32, "dpool"); _req(mask, torch.float32, "mask"); _req(dx, torch.bfloat16, "dx")
    _check(lib().mmf_masked_mean_bwd(_p(dpool), _p(mask), _p(dx), B, T, H, _stream()), "mmf_masked_mean_bwd")


def rowgroup_scale(x, ld, gate, groups, rows_per_group, cols):
    """x[g * rows_per_group + r, :cols] *= gate[g] in place (bf16 rows of leading dimension ld, gate fp32 [groups, cols])."""
    _req(x, torch.bfloat16, "x"); _req(gate, torch.float32, "gate")
    _check(lib().mmf_rowgroup_scale(_p(x), ld, _p(gate), groups, rows_per_group, cols, _stream()), "mmf_rowgroup_scale")


def rowgroup_scale_bwd(dy, y, ld, gate, dgate, groups, rows_per_group, cols):
    for t, n in ((dy, "dy"), (y, "y")):
        _req(t, torch.bfloat16, n)
    _req(gate, torch.float32, "gate"); _req(dgate, torch.float32, "dgate")
    _check(lib().mmf_rowgroup_scale_bwd(_p(dy), _p(y), ld, _p(gate), _p(dgate), groups, rows_per_group, cols, _stream()),
           "mmf_rowgroup_scale_bwd")


def gather_rows(x, index, out, B, S, H, drop=NO_DROP):
    _req(x, torch.bfloat16, "x"); _req(index, torch.int64, "index"); _req(out, torch.bfloat16, "out")
    k, t, sc, sd = _drop4(drop)
    _check(lib().mmf_gather_rows(_p(x), _p(index), _p(out), B, S, H, C.c_uint32(k), C.c_uint32(t), C.c_float(sc), sd, _stream()),
           "mmf_gather_rows")


def scatter_rows(dout, index, dx, B, S, H, drop=NO_DROP):
    _req(dout, torch.bfloat16, "dout"); _req(index, torch.int64, "index"); _req(dx, torch.bfloat16, "dx")
    k, t, sc, sd = _drop4(drop)
    _check(lib().mmf_scatter_rows(_p(dout), _p(index), _p(dx), B, S, H, C.c_uint32(k), C.c_uint32(t), C.c_float(sc), sd, _stream()),
           "mmf_scatter_rows")


def colsum_ws_floats(N):
    return lib().mmf_colsum_ws_floats(N)


def colsum(x, ld, nb, rpb, bstride, N, out, beta, partials):
    _req(x, torch.bfloat16, "x"); _req(out, torch.float32, "out"); _req(partials, torch.float32, "partials")
    _check(lib().mmf_colsum_bf16(_p(x), ld, nb, rpb, bstride, N, _p(out), C.c_float(beta), _p(partials), _stream()),
           "mmf_colsum_bf16")


def cast_f32_to_bf16(src, dst, n=None):
    _req(src, torch.float32, "src"); _req(dst, torch.bfloat16, "dst")
    n = src.numel() if n is None else n
    _check(lib().mmf_cast_f32_to_bf16(_p(src), _p(dst), C.c_int64(n), _stream()), "mmf_cast_f32_to_bf16")


def cast_bf16_to_f32(src, dst, n=None):
    _req(src, torch.bfloat16, "src"); _req(dst, torch.float32, "dst")
    n = src.numel() if n is None else n
    _check(lib().mmf_cast_bf16_to_f32(_p(src), _p(dst), C.c_int64(n), _stream()), "mmf_cast_bf16_to_f32")


def cast2d_f32_to_bf16(src, lds, dst, ldd, rows, cols):
    _req(src, torch.float32, "src"); _req(dst, torch.bfloat16, "dst")
    _check(lib().mmf_cast2d_f32_to_bf16(_p(src), lds, _p(dst), ldd, rows, cols, _stream()), "mmf_cast2d_f32_to_bf16")


def cast2d_bf16_to_f32(src, lds, dst, ldd, rows, cols):
    _req(src, torch.bfloat16, "src"); _req(dst, torch.float32, "dst")
    _check(lib().mmf_cast2d_bf16_to_f32(_p(src), lds, _p(dst), ldd, rows, cols, _stream()), "mmf_cast2d_bf16_to_f32")


def dropout(x, y, drop):
    _req(x, torch.bfloat16, "x"); _req(y, torch.bfloat16, "y")
    k, t, sc, sd = _drop4(drop)
    _check(lib().mmf_dropout_bf16(_p(x), _p(y), C.c_int64(x.numel()), C.c_uint32(k), C.c_uint32(t), C.c_float(sc), sd, _stream()),
           "mmf_dropout_bf16")


def step_advance(seed, state, schedule=0, warmup=0.0, total=0.0):
    """seed_advance and optim_state_advance as one launch (either tensor may be None)."""
    _req(seed, torch.int32, "seed"); _req(state, torch.float32, "state")
    _check(lib().mmf_step_advance(_p(seed), _p(state), int(schedule), C.c_float(warmup), C.c_float(total), _stream()), "mmf_step_advance")


def seed_advance(seed):
    _req(seed, torch.int32, "seed")
    _check(lib().mmf_seed_advance(_p(seed), _stream()), "mmf_seed_advance")


def gelu_bwd(dh, u, du):
    for t, nme in ((dh, "dh"), (u, "u"), (du, "du")):
        _req(t, torch.bfloat16, nme)
    _check(lib().mmf_gelu_bwd_bf16(_p(dh), _p(u), _p(du), C.c_int64(dh.numel()), _stream()), "mmf_gelu_bwd_bf16")


def eltwise(op, a, b, out):
    _req(a, torch.bfloat16, "a"); _req(b, torch.bfloat16, "b"); _req(out, torch.bfloat16, "out")
    _check(lib().mmf_eltwise_bf16(int(op), _p(a), _p(b), _p(out), C.c_int64(a.numel()), _stream()), "mmf_eltwise_bf16")


def tanh_bwd(dy, y, dx):
    for t, nme in ((dy, "dy"), (y, "y"), (dx, "dx")):
        _req(t, torch.bfloat16, nme)
    _check(lib().mmf_tanh_bwd_bf16(_p(dy), _p(y), _p(dx), C.c_int64(dy.numel()), _stream()), "mmf_tanh_bwd_bf16")


# include/mmf_amd.h MMF_TUN_*: the measurement knobs that are left (MMF_AMD_TUN="id:value,..." sets them at load time)
TUN_GEMM_WIDE, TUN_ALT_FORMS, TUN_EPI_NT, TUN_NT_SITE_KEEP, TUN_WGRAD_WIDE, TUN_SC1_SITE, TUN_SCATTER_ATOMIC, TUN_GEMM_PERSIST = 2, 3, 6, 8, 10, 14, 17, 18


def set_tunable(which, value):
    _check(lib().mmf_amd_set_tunable(int(which), int(value)), "mmf_amd_set_tunable")


def visual_masks(input_mask, image_dim, B, T, R, image_mask, attention_mask, vtype, mask_add, pool_index):
    for t, n in ((input_mask, "input_mask"), (image_dim, "image_dim"), (image_mask, "image_mask"), (attention_mask, "attention_mask"), (vtype, "vtype"),
                 (pool_index, "pool_index")):
        _req(t, torch.int64, n)
    _req(mask_add, torch.float32, "mask_add")
    _check(lib().mmf_visual_masks(_p(input_mask), _p(image_dim), B, T, R, _p(image_mask), _p(attention_mask), _p(vtype), _p(mask_add), _p(pool_index),
                                  _stream()), "mmf_visual_masks")


def make_additive_mask(mask, out):
    _req(mask, torch.int64, "mask"); _req(out, torch.float32, "out")
    _check(lib().mmf_make_additive_mask(_p(mask), _p(out), C.c_int64(mask.numel()), _stream()), "mmf_make_additive_mask")


def bce_logits_fwd(scores, targets, loss, B, N):
    for t, n in ((scores, "scores"), (targets, "targets"), (loss, "loss")):
        _req(t, torch.float32, n)
    ws = torch.empty(lib().mmf_bce_logits_ws_floats(), dtype=torch.float32, device=scores.device)
    _check(lib().mmf_bce_logits_fwd(_p(scores), _p(targets), _p(loss), _p(ws), B, N, _stream()), "mmf_bce_logits_fwd")


def bce_logits_bwd(scores, targets, gloss, dscores, ldd, B, N):
    for t, n in ((scores, "scores"), (targets, "targets"), (gloss, "gloss")):
        _req(t, torch.float32, n)
    _req(dscores, torch.bfloat16, "dscores")
    _check(lib().mmf_bce_logits_bwd(_p(scores), _p(targets), _p(gloss), _p(dscores), ldd, B, N, _stream()),
           "mmf_bce_logits_bwd")


def cross_entropy_fwd(logits, labels, loss, count, B, Cn, ignore_index=-100):
    _req(logits, torch.float32, "logits"); _req(labels, torch.int64, "labels"); _req(loss, torch.float32, "loss"); _req(count, torch.float32, "count")
    _check(lib().mmf_cross_entropy_fwd(_p(logits), _p(labels), _p(loss), _p(count), B, Cn, ignore_index, _stream()), "mmf_cross_entropy_fwd")


def cross_entropy_bwd(logits, labels, count, gloss, dlogits, B, Cn, ignore_index=-100):
    _req(logits, torch.float32, "logits"); _req(labels, torch.int64, "labels"); _req(dlogits, torch.float32, "dlogits")
    _check(lib().mmf_cross_entropy_bwd(_p(logits), _p(labels), _p(count), _p(gloss), _p(dlogits), B, Cn, ignore_index, _stream()),
           "mmf_cross_entropy_bwd")


def vocab_cross_entropy_fwd(logits, labels, lse, rowloss, loss, count, R, Cn, ignore_index=-1):
    """Masked-LM cross-entropy over [R, Cn] fp32 logits (contiguous rows): per-row log-sum-exp / loss, mean loss, count."""
    for t, n in ((logits, "logits"), (lse, "lse"), (rowloss, "rowloss"), (loss, "loss"), (count, "count")):
        _req(t, torch.float32, n)
    _req(labels, torch.int64, "labels")
    _check(lib().mmf_vocab_cross_entropy_fwd(_p(logits), logits.stride(0), _p(labels), _p(lse), _p(rowloss), _p(loss), _p(count), R, Cn,
                                             ignore_index, _stream()), "mmf_vocab_cross_entropy_fwd")


def vocab_cross_entropy_bwd(logits, labels, lse, count, gloss, dlogits, ldd, R, Cn, ignore_index=-1):
    """dlogits: bf16 [R, ldd], ldd = round_up(Cn, 8): the zero-padded GEMM operand."""
    for t, n in ((logits, "logits"), (lse, "lse"), (count, "count"), (gloss, "gloss")):
        _req(t, torch.float32, n)
    _req(labels, torch.int64, "labels"); _req(dlogits, torch.bfloat16, "dlogits")
    _check(lib().mmf_vocab_cross_entropy_bwd(_p(logits), logits.stride(0), _p(labels), _p(lse), _p(count), _p(gloss), _p(dlogits), ldd, R, Cn,
                                             ignore_index, _stream()), "mmf_vocab_cross_entropy_bwd")


def soft_target_kl_fwd(logits, target, row_label, lse, tsum, rowloss, loss, count, R, Cn):
    """Masked soft-target KL (ViLBERT masked-region loss): logits / target fp32 [R, Cn], row_label int64 [R] (1 = counted)."""
    for t, n in ((logits, "logits"), (target, "target"), (lse, "lse"), (tsum, "tsum"), (rowloss, "rowloss"), (loss, "loss"), (count, "count")):
        _req(t, torch.float32, n)
    _req(row_label, torch.int64, "row_label")
    _check(lib().mmf_soft_target_kl_fwd(_p(logits), logits.stride(0), _p(target), target.stride(0), _p(row_label), _p(lse), _p(tsum),
                                        _p(rowloss), _p(loss), _p(count), R, Cn, _stream()), "mmf_soft_target_kl_fwd")


def soft_target_kl_bwd(logits, target, row_label, lse, tsum, count, gloss, dlogits, ldd, R, Cn):
    for t, n in ((logits, "logits"), (target, "target"), (lse, "lse"), (tsum, "tsum"), (count, "count"), (gloss, "gloss")):
        _req(t, torch.float32, n)
    _req(row_label, torch.int64, "row_label"); _req(dlogits, torch.bfloat16, "dlogits")
    _check(lib().mmf_soft_target_kl_bwd(_p(logits), logits.stride(0), _p(target), target.stride(0), _p(row_label), _p(lse), _p(tsum),
                                        _p(count), _p(gloss), _p(dlogits), ldd, R, Cn, _stream()), "mmf_soft_target_kl_bwd")


# --------------------------------------------------------------------------------------------
# fp32-accurate forward path (mmf_amd/csrc/fp32_path.hip)
# --------------------------------------------------------------------------------------------
def gemm_f32(A, B, C_out, M, N, K, lda, ldb, ldc, bias=None, coladd=None, rowtab=None, rowidx=None, rowtab_ld=0, act=0, resid=None,
             ldr=0, grp=(0, 0, 0), a_kmajor=False, b_kmajor=False, U=None, aux=None, drop=NO_DROP, beta=0.0, split_k=False):
    """C = epilogue(A B^T) on the fp32-input MFMA, everything fp32.  Layouts: forward (A [M, K], B [N, K]), dgrad (b_kmajor: B [K, N]),
    weight gradient (a_kmajor and b_kmajor: A [K, M], B [K, N]; `split_k` lets the library split the long contraction over slabs)."""
    for t, n in ((A, "A"), (B, "B"), (C_out, "C"), (resid, "resid"), (bias, "bias"), (coladd, "coladd"), (rowtab, "rowtab"), (U, "U"), (aux, "aux")):
        _req(t, torch.float32, n)
    _req(rowidx, torch.int64, "rowidx")
    d = GemmDesc()
    d.A, d.B, d.C = _p(A), _p(B), _p(C_out)
    d.M, d.N, d.K = M, N, K
    d.lda, d.ldb, d.ldc = lda, ldb, ldc
    d.a_kmajor, d.b_kmajor = int(a_kmajor), int(b_kmajor)
    d.a_f32 = d.b_f32 = d.out_f32 = 1
    d.beta = beta
    d.bias, d.coladd, d.rowtab, d.rowidx, d.rowtab_ld = _p(bias), _p(coladd), _p(rowtab), _p(rowidx), rowtab_ld
    d.act = act
    d.U, d.aux = _p(U), _p(aux)
    d.resid, d.ldr = _p(resid), ldr
    d.drop_key, d.drop_thr16, d.drop_scale, d.drop_seed = _drop4(drop)
    d.grp_in, d.grp_pad, d.grp_off = grp
    if split_k:
        sp = lib().mmf_gemm_f32_splits(M, N, K)
        if sp > 1:
            ws = torch.empty(sp * M * N, dtype=torch.float32, device=C_out.device)
            d.splitk_ws, d.splitk_ws_bytes = _p(ws), ws.numel() * 4
    _check(lib().mmf_gemm_f32(C.byref(d), _stream()), "mmf_gemm_f32")


def _attn_f32_desc(q, k, v, ldq, ldk, ldv, mask, ctx, ldo, B, heads, Sq, Sk, scale, head_dim, causal_tail, lse, drop):
    for t, n in ((q, "q"), (k, "k"), (v, "v"), (ctx, "ctx"), (mask, "mask"), (lse, "lse")):
        _req(t, torch.float32, n)
    d = AttnDesc()
    d.q, d.k, d.v = _p(q), _p(k), _p(v)
    d.ldq, d.ldk, d.ldv = ldq, ldk, ldv
    d.mask, d.ctx, d.ldo, d.lse = _p(mask), _p(ctx), ldo, _p(lse)
    d.B, d.heads, d.Sq, d.Sk = B, heads, Sq, Sk
    d.scale = scale
    d.drop_key, d.drop_thr16, d.drop_scale, d.drop_seed = _drop4(drop)
    d.head_dim = head_dim
    d.causal_tail = causal_tail
    # a 3-D mask [B, Sq, Sk] is a materialised additive mask per (query, key) pair (mmf_attn_desc.mask_query_stride), as on the bf16 path
    if mask is not None and mask.dim() == 3:
        if tuple(mask.shape) != (B, Sq, Sk) or mask.stride(2) != 1:
            raise NativeLibraryError("a per-query attention mask must be [B, Sq, Sk] with contiguous rows, got %s" % (tuple(mask.shape),))
        d.mask_query_stride = int(mask.stride(1))
        d.mask_batch_stride = int(mask.stride(0))
    elif mask is not None and mask.dim() == 4:      # one [Sq, Sk] mask per head (mmf_attn_desc.mask_head_stride)
        if tuple(mask.shape) != (B, heads, Sq, Sk) or mask.stride(3) != 1:
            raise NativeLibraryError("a per-head attention mask must be [B, heads, Sq, Sk] with contiguous rows, got %s" % (tuple(mask.shape),))
        d.mask_query_stride, d.mask_head_stride, d.mask_batch_stride = int(mask.stride(2)), int(mask.stride(1)), int(mask.stride(0))
    return d


def attention_f32_fwd(q, k, v, ldq, ldk, ldv, mask, ctx, ldo, B, heads, Sq, Sk, scale, head_dim=64, causal_tail=0, lse=None, drop=NO_DROP):
    """`lse` (fp32 [B, heads, Sq], optional) receives the row statistic mmf_attention_f32_bwd needs; `drop`: probability dropout (training)."""
    d = _attn_f32_desc(q, k, v, ldq, ldk, ldv, mask, ctx, ldo, B, heads, Sq, Sk, scale, head_dim, causal_tail, lse, drop)
    _check(lib().mmf_attention_f32_fwd(C.byref(d), _stream()), "mmf_attention_f32_fwd")


def attention_f32_bwd(q, k, v, ldq, ldk, ldv, mask, ctx, ldo, lse, B, heads, Sq, Sk, scale, dctx, dq, dk, dv, delta, head_dim=64, causal_tail=0,
                      drop=NO_DROP):
    d = AttnBwdDesc()
    d.f = _attn_f32_desc(q, k, v, ldq, ldk, ldv, mask, ctx, ldo, B, heads, Sq, Sk, scale, head_dim, causal_tail, lse, drop)
    for t, n in ((dctx, "dctx"), (dq, "dq"), (dk, "dk"), (dv, "dv"), (delta, "delta")):
        _req(t, torch.float32, n)
    d.dctx, d.dq, d.dk, d.dv, d.delta = _p(dctx), _p(dq), _p(dk), _p(dv), _p(delta)
    _check(lib().mmf_attention_f32_bwd(C.byref(d), _stream()), "mmf_attention_f32_bwd")


def layernorm_f32_fwd_stats(x, gamma, beta, y, mean, rstd, rows, H, eps):
    for t, n in ((x, "x"), (gamma, "gamma"), (beta, "beta"), (y, "y"), (mean, "mean"), (rstd, "rstd")):
        _req(t, torch.float32, n)
    _check(lib().mmf_layernorm_f32_fwd_stats(_p(x), _p(gamma), _p(beta), _p(y), _p(mean), _p(rstd), rows, H, C.c_float(eps), _stream()),
           "mmf_layernorm_f32_fwd_stats")


def layernorm_f32_bwd(dy, x, mean, rstd, gamma, dx, dgamma, dbeta, rows, H):
    for t, n in ((dy, "dy"), (x, "x"), (mean, "mean"), (rstd, "rstd"), (gamma, "gamma"), (dx, "dx"), (dgamma, "dgamma"), (dbeta, "dbeta")):
        _req(t, torch.float32, n)
    ws = torch.empty(lib().mmf_layernorm_f32_bwd_blocks(rows) * 2 * H, dtype=torch.float32, device=dy.device)
    _check(lib().mmf_layernorm_f32_bwd(_p(dy), _p(x), _p(mean), _p(rstd), _p(gamma), _p(dx), _p(dgamma), _p(dbeta), _p(ws), rows, H, _stream()),
           "mmf_layernorm_f32_bwd")


def colsum_f32(x, ld, rows, N, out, accumulate=False):
    _req(x, torch.float32, "x"); _req(out, torch.float32, "out")
    ws = torch.empty(lib().mmf_colsum_f32_slices(rows) * N, dtype=torch.float32, device=x.device)
    _check(lib().mmf_colsum_f32(_p(x), ld, rows, N, _p(out), int(accumulate), _p(ws), _stream()), "mmf_colsum_f32")


def dropout_f32(x, y, drop):
    _req(x, torch.float32, "x"); _req(y, torch.float32, "y")
    key, thr, scale, seed = _drop4(drop)
    _check(lib().mmf_dropout_f32(_p(x), _p(y), C.c_long(x.numel()), key, thr, C.c_float(scale), seed, _stream()), "mmf_dropout_f32")


def expand_batch(x, out, Bs, reps, n, mode):
    """bf16 or fp32 activations (both tensors of one dtype): the library's bf16 / f32 entry."""
    if x.dtype == torch.float32:
        _req(x, torch.float32, "x"); _req(out, torch.float32, "out")
        _check(lib().mmf_expand_batch_f32(_p(x), _p(out), C.c_int64(Bs), C.c_int64(reps), C.c_int64(n), mode, _stream()), "mmf_expand_batch_f32")
        return
    _req(x, torch.bfloat16, "x"); _req(out, torch.bfloat16, "out")
    _check(lib().mmf_expand_batch_bf16(_p(x), _p(out), C.c_int64(Bs), C.c_int64(reps), C.c_int64(n), mode, _stream()), "mmf_expand_batch_bf16")


def reduce_batch(g, dx, Bs, reps, n, mode):
    if g.dtype == torch.float32:
        _req(g, torch.float32, "g"); _req(dx, torch.float32, "dx")
        _check(lib().mmf_reduce_batch_f32(_p(g), _p(dx), C.c_int64(Bs), C.c_int64(reps), C.c_int64(n), mode, _stream()), "mmf_reduce_batch_f32")
        return
    _req(g, torch.bfloat16, "g"); _req(dx, torch.bfloat16, "dx")
    _check(lib().mmf_reduce_batch_bf16(_p(g), _p(dx), C.c_int64(Bs), C.c_int64(reps), C.c_int64(n), mode, _stream()), "mmf_reduce_batch_bf16")


def nce_fwd(pred, target, neg, label, scores, lse, rowloss, loss, count, M, N, K):
    for t, n in ((pred, "pred"), (target, "target"), (scores, "scores"), (lse, "lse"), (rowloss, "rowloss"), (loss, "loss"), (count, "count")):
        _req(t, torch.float32, n)
    _req(neg, torch.int64, "neg"); _req(label, torch.int64, "label")
    _check(lib().mmf_nce_fwd(_p(pred), _p(target), _p(neg), _p(label), _p(scores), _p(lse), _p(rowloss), _p(loss), _p(count), M, N, K, _stream()), "mmf_nce_fwd")


def nce_bwd(target, neg, label, scores, lse, count, gloss, dpred, ldd, M, N, K):
    for t, n in ((target, "target"), (scores, "scores"), (lse, "lse"), (count, "count"), (gloss, "gloss")):
        _req(t, torch.float32, n)
    _req(neg, torch.int64, "neg"); _req(label, torch.int64, "label"); _req(dpred, torch.bfloat16, "dpred")
    _check(lib().mmf_nce_bwd(_p(target), _p(neg), _p(label), _p(scores), _p(lse), _p(count), _p(gloss), _p(dpred), ldd, M, N, K, _stream()), "mmf_nce_bwd")


def vocab_cross_entropy_f32_bwd(logits, labels, lse, count, gloss, dlogits, ldd, R, Cn, ignore_index=-1):
    for t, n in ((logits, "logits"), (lse, "lse"), (count, "count"), (gloss, "gloss"), (dlogits, "dlogits")):
        _req(t, torch.float32, n)
    _req(labels, torch.int64, "labels")
    _check(lib().mmf_vocab_cross_entropy_f32_bwd(_p(logits), logits.stride(0), _p(labels), _p(lse), _p(count), _p(gloss), _p(dlogits), ldd, R, Cn,
                                                 ignore_index, _stream()), "mmf_vocab_cross_entropy_f32_bwd")


def bce_logits_f32_bwd(scores, targets, gloss, dscores, B, N):
    for t, n in ((scores, "scores"), (targets, "targets"), (gloss, "gloss"), (dscores, "dscores")):
        _req(t, torch.float32, n)
    _check(lib().mmf_bce_logits_f32_bwd(_p(scores), _p(targets), _p(gloss), _p(dscores), B, N, _stream()), "mmf_bce_logits_f32_bwd")


def scatter_add_rows_f32(g, ld, rows, H, idx, out, ldo, grp=(0, 0, 0), dst_stride=0, skip=-1):
    """out[idx[r] + r * dst_stride] += g[src(r)], src(r) = (r // grp[0]) * grp[1] + grp[2] + r % grp[0] when grp[0] > 0 (else r)."""
    _req(g, torch.float32, "g"); _req(out, torch.float32, "out"); _req(idx, torch.int64, "idx")
    _check(lib().mmf_scatter_add_rows_f32(_p(g), ld, rows, H, grp[0], grp[1], grp[2], _p(idx), C.c_long(dst_stride), C.c_long(skip),
                                          int(out.shape[0]), _p(out), ldo, _stream()), "mmf_scatter_add_rows_f32")


def layernorm_f32_fwd(x, gamma, beta, y, rows, H, eps):
    for t, n in ((x, "x"), (gamma, "gamma"), (beta, "beta"), (y, "y")):
        _req(t, torch.float32, n)
    _check(lib().mmf_layernorm_f32_fwd(_p(x), _p(gamma), _p(beta), _p(y), rows, H, C.c_float(eps), _stream()), "mmf_layernorm_f32_fwd")


def embed_text_f32_fwd(ids, seg, word, pos, typ, y, B, T, S, H, row0=0, pos0=0):
    _req(ids, torch.int64, "ids"); _req(seg, torch.int64, "seg"); _req(y, torch.float32, "y")
    for t, n in ((word, "word"), (pos, "pos"), (typ, "type")):
        _req(t, torch.float32, n)
    _check(lib().mmf_embed_text_f32_fwd(_p(ids), _p(seg), _p(word), _p(pos), _p(typ), _p(y), B, T, S, H, row0, pos0,
                                        int(word.shape[0]), int(pos.shape[0]), int(typ.shape[0]), _stream()), "mmf_embed_text_f32_fwd")


def rows_add_embed_f32(x, seg, pos, typ, y, B, L, S, H, row0=0, pos0=0):
    _req(x, torch.float32, "x"); _req(y, torch.float32, "y"); _req(seg, torch.int64, "seg")
    _req(pos, torch.float32, "pos"); _req(typ, torch.float32, "type")
    _check(lib().mmf_rows_add_embed_f32(_p(x), _p(seg), _p(pos), _p(typ), _p(y), B, L, S, H, row0, pos0, _stream()), "mmf_rows_add_embed_f32")


def l2norm_rows_f32(x, ldx, y, ldy, rows, D, eps=1e-12):
    _req(x, torch.float32, "x"); _req(y, torch.float32, "y")
    _check(lib().mmf_l2norm_rows_f32(_p(x), ldx, _p(y), ldy, rows, D, C.c_float(eps), _stream()), "mmf_l2norm_rows_f32")


def gather_rows2_f32(a, b, idx, out, n, H):
    for t, nm in ((a, "a"), (b, "b"), (out, "out")):
        _req(t, torch.float32, nm)
    _req(idx, torch.int64, "idx")
    _check(lib().mmf_gather_rows2_f32(_p(a), C.c_int64(a.shape[0]), _p(b), C.c_int64(b.shape[0]), _p(idx), _p(out), n, H, _stream()), "mmf_gather_rows2_f32")


def ptr_scores_f32(q, k, mask_add, out, ldo, B, T, N, HQ, scale):
    for t, nm in ((q, "q"), (k, "k"), (mask_add, "mask_add"), (out, "out")):
        _req(t, torch.float32, nm)
    _check(lib().mmf_ptr_scores_f32(_p(q), _p(k), _p(mask_add), _p(out), ldo, B, T, N, HQ, C.c_float(scale), _stream()), "mmf_ptr_scores_f32")


def pad_rows_f32(src, K, dst, KP, rows):
    _req(src, torch.float32, "src"); _req(dst, torch.float32, "dst")
    _check(lib().mmf_pad_rows_f32(_p(src), K, _p(dst), KP, rows, _stream()), "mmf_pad_rows_f32")


def eltwise_f32(op, a, b, y):
    _req(a, torch.float32, "a"); _req(b, torch.float32, "b"); _req(y, torch.float32, "y")
    _check(lib().mmf_eltwise_f32(op, _p(a), _p(b), _p(y), C.c_long(a.numel()), _stream()), "mmf_eltwise_f32")


def masked_mean_f32(x, mask, pool, B, T, H):
    _req(x, torch.float32, "x"); _req(mask, torch.float32, "mask"); _req(pool, torch.float32, "pool")
    _check(lib().mmf_masked_mean_f32(_p(x), _p(mask), _p(pool), B, T, H, _stream()), "mmf_masked_mean_f32")


def rowgroup_scale_f32(x, ld, gate, groups, rows_per_group, Cn):
    _req(x, torch.float32, "x"); _req(gate, torch.float32, "gate")
    _check(lib().mmf_rowgroup_scale_f32(_p(x), ld, _p(gate), groups, rows_per_group, Cn, _stream()), "mmf_rowgroup_scale_f32")


# ---- fp32 backwards of the operators round 5 added to mmf_amd.fp32_training() (gate_ops.hip, m4c_ops.hip, rowops.hip) -----------------
def segment_sum_rows_f32(sorted_ids, perm, rows, out):
    """out[id] = sum over the segment of equal ids (stable-sorted) of rows[perm[j]], in sorted order, no atomics (see include/mmf_amd.h)."""
    _req(sorted_ids, torch.int64, "sorted_ids"); _req(perm, torch.int64, "perm"); _req(rows, torch.float32, "rows"); _req(out, torch.float32, "out")
    M, H = rows.shape
    _check(lib().mmf_segment_sum_rows_f32(_p(sorted_ids), _p(perm), _p(rows), _p(out), M, H, C.c_int64(out.shape[0]), _stream()), "mmf_segment_sum_rows_f32")


def slice_rows_f32(src, ld_src, K, dst, KP, rows):
    """dst[r, :KP] = src[r * ld_src + :K] followed by zeros (a column slice of wider fp32 rows as a 16-byte-row GEMM operand)."""
    _req(src, torch.float32, "src"); _req(dst, torch.float32, "dst")
    _check(lib().mmf_slice_rows_f32(_p(src), ld_src, K, _p(dst), KP, rows, _stream()), "mmf_slice_rows_f32")


def masked_mean_f32_bwd(dpool, mask, dx, B, T, H):
    _req(dpool, torch.float32, "dpool"); _req(mask, torch.float32, "mask"); _req(dx, torch.float32, "dx")
    _check(lib().mmf_masked_mean_f32_bwd(_p(dpool), _p(mask), _p(dx), B, T, H, _stream()), "mmf_masked_mean_f32_bwd")


def rowgroup_scale_f32_bwd(dy, y, ld, gate, dgate, groups, rows_per_group, Cn):
    """In place: dy becomes the gradient of the un-gated rows; dgate[g][c] = sum_r dy * x."""
    for t, n in ((dy, "dy"), (y, "y"), (gate, "gate"), (dgate, "dgate")):
        _req(t, torch.float32, n)
    _check(lib().mmf_rowgroup_scale_f32_bwd(_p(dy), _p(y), ld, _p(gate), _p(dgate), groups, rows_per_group, Cn, _stream()), "mmf_rowgroup_scale_f32_bwd")


def align_pos_f32_bwd(dvis, ld, nb, rpb, bstride, align, dpos, A, H):
    _req(dvis, torch.float32, "dvis"); _req(align, torch.int64, "align"); _req(dpos, torch.float32, "dpos")
    _check(lib().mmf_align_pos_f32_bwd(_p(dvis), ld, nb, rpb, bstride, _p(align), _p(dpos), A, H, int(dpos.shape[0]), _stream()), "mmf_align_pos_f32_bwd")


def soft_target_kl_f32_bwd(logits, target, row_label, lse, tsum, count, gloss, dlogits, ldd, R, Cn):
    for t, n in ((logits, "logits"), (target, "target"), (lse, "lse"), (tsum, "tsum"), (count, "count"), (gloss, "gloss"), (dlogits, "dlogits")):
        _req(t, torch.float32, n)
    _req(row_label, torch.int64, "row_label")
    _check(lib().mmf_soft_target_kl_f32_bwd(_p(logits), logits.stride(0), _p(target), target.stride(0), _p(row_label), _p(lse), _p(tsum),
                                            _p(count), _p(gloss), _p(dlogits), ldd, R, Cn, _stream()), "mmf_soft_target_kl_f32_bwd")


def mse_f32_bwd(pred, target, gloss, dpred, ldd, rows, cols, row_label=None, count=None):
    for t, n in ((pred, "pred"), (target, "target"), (gloss, "gloss"), (count, "count"), (dpred, "dpred")):
        _req(t, torch.float32, n)
    _req(row_label, torch.int64, "row_label")
    _check(lib().mmf_mse_f32_bwd(_p(pred), pred.stride(0), _p(target), target.stride(0), _p(row_label), _p(count), _p(gloss), _p(dpred), ldd, rows, cols,
                                 _stream()), "mmf_mse_f32_bwd")


def nce_f32_bwd(target, neg, label, scores, lse, count, gloss, dpred, ldd, M, N, K):
    for t, n in ((target, "target"), (scores, "scores"), (lse, "lse"), (count, "count"), (gloss, "gloss"), (dpred, "dpred")):
        _req(t, torch.float32, n)
    _req(neg, torch.int64, "neg"); _req(label, torch.int64, "label")
    _check(lib().mmf_nce_f32_bwd(_p(target), _p(neg), _p(label), _p(scores), _p(lse), _p(count), _p(gloss), _p(dpred), ldd, M, N, K, _stream()), "mmf_nce_f32_bwd")


def l2norm_rows_f32_bwd(g, ldg, y, ldy, x, ldx, dx, lddx, rows, D, eps=1e-12):
    """dx = (g - y <g, y>) / max(||x||, eps): autograd of F.normalize on fp32 rows (the factor is recomputed from x)."""
    for t, n in ((g, "g"), (y, "y"), (x, "x"), (dx, "dx")):
        _req(t, torch.float32, n)
    inv = torch.empty(rows, dtype=torch.float32, device=g.device)
    _check(lib().mmf_l2norm_rows_f32_bwd(_p(g), ldg, _p(y), ldy, _p(x), ldx, _p(inv), _p(dx), lddx, rows, D, C.c_float(eps), _stream()),
           "mmf_l2norm_rows_f32_bwd")


def ptr_scores_f32_bwd(dscores, ldd, q, k, dq, dk, B, T, N, HQ, scale):
    for t, n in ((dscores, "dscores"), (q, "q"), (k, "k"), (dq, "dq"), (dk, "dk")):
        _req(t, torch.float32, n)
    _check(lib().mmf_ptr_scores_f32_bwd(_p(dscores), ldd, _p(q), _p(k), _p(dq), _p(dk), B, T, N, HQ, C.c_float(scale), _stream()), "mmf_ptr_scores_f32_bwd")


def gather_rows_f32(x, index, out, B, S, H):
    _req(x, torch.float32, "x"); _req(index, torch.int64, "index"); _req(out, torch.float32, "out")
    _check(lib().mmf_gather_rows_f32(_p(x), _p(index), _p(out), B, S, H, _stream()), "mmf_gather_rows_f32")


# --------------------------------------------------------------------------------------------
# UNITER pretraining heads (mmf_amd/csrc/uniter_ops.hip)
# --------------------------------------------------------------------------------------------
def mse_fwd(pred, target, loss, rows, cols, row_label=None, count=None):
    """loss[0] = mean((pred - target)^2); pred / target fp32 [rows, cols] (row strides from the tensors).  With `row_label` (int64 [rows]):
    the sum over the rows with label 1 divided by max(their element count, 1), which is written to `count`."""
    for t, n in ((pred, "pred"), (target, "target"), (loss, "loss"), (count, "count")):
        _req(t, torch.float32, n)
    _req(row_label, torch.int64, "row_label")
    ws = torch.empty(lib().mmf_mse_ws_floats(), dtype=torch.float32, device=pred.device)
    _check(lib().mmf_mse_fwd(_p(pred), pred.stride(0), _p(target), target.stride(0), _p(row_label), _p(loss), _p(count), _p(ws), rows, cols, _stream()),
           "mmf_mse_fwd")


def mse_bwd(pred, target, gloss, dpred, ldd, rows, cols, row_label=None, count=None):
    for t, n in ((pred, "pred"), (target, "target"), (gloss, "gloss"), (count, "count")):
        _req(t, torch.float32, n)
    _req(dpred, torch.bfloat16, "dpred"); _req(row_label, torch.int64, "row_label")
    _check(lib().mmf_mse_bwd(_p(pred), pred.stride(0), _p(target), target.stride(0), _p(row_label), _p(count), _p(gloss), _p(dpred), ldd, rows, cols,
                             _stream()), "mmf_mse_bwd")


def _wra_desc(seq, ld, B, S, H, M, N, txt_pad, img_pad, label, xinv, yinv, plan, cost, dist):
    _req(seq, torch.bfloat16, "seq"); _req(label, torch.int64, "label")
    for t, n in ((txt_pad, "txt_pad"), (img_pad, "img_pad"), (xinv, "xinv"), (yinv, "yinv"), (plan, "plan"), (cost, "cost"), (dist, "dist")):
        _req(t, torch.float32, n)
    d = WraDesc()
    d.seq, d.ld, d.B, d.S, d.H, d.M, d.N = _p(seq), ld, B, S, H, M, N
    d.txt_pad, d.img_pad, d.label = _p(txt_pad), _p(img_pad), _p(label)
    d.xinv, d.yinv, d.plan, d.cost, d.dist = _p(xinv), _p(yinv), _p(plan), _p(cost), _p(dist)
    d.beta, d.eps, d.iterations = 0.5, 1e-5, 50
    return d


def wra_fwd(seq, ld, B, S, H, M, N, txt_pad, img_pad, label, xinv, yinv, plan, cost, dist, loss, count):
    """UNITER word-region alignment (heads/wra.py + modules/ot.py): per-sample OT distance by 50 IPOT steps, signed mean."""
    _req(loss, torch.float32, "loss"); _req(count, torch.float32, "count")
    d = _wra_desc(seq, ld, B, S, H, M, N, txt_pad, img_pad, label, xinv, yinv, plan, cost, dist)
    _check(lib().mmf_wra_fwd(C.byref(d), _p(loss), _p(count), _stream()), "mmf_wra_fwd")


def wra_bwd(seq, ld, B, S, H, M, N, txt_pad, img_pad, label, xinv, yinv, plan, cost, dist, gloss, count, dseq, ldd):
    _req(gloss, torch.float32, "gloss"); _req(count, torch.float32, "count"); _req(dseq, torch.bfloat16, "dseq")
    d = _wra_desc(seq, ld, B, S, H, M, N, txt_pad, img_pad, label, xinv, yinv, plan, cost, dist)
    _check(lib().mmf_wra_bwd(C.byref(d), _p(gloss), _p(count), _p(dseq), ldd, _stream()), "mmf_wra_bwd")


# --------------------------------------------------------------------------------------------
# M4C kernels (mmf_amd/csrc/m4c_ops.hip)
# --------------------------------------------------------------------------------------------
def l2norm_rows_fwd(x, ldx, y, ldy, inv_norm, rows, D, eps=1e-12):
    """y[r, :D] = x[r, :D] / max(||x[r, :D]||, eps); x fp32 or bf16; y bf16 (may be a column slice of a wider row)."""
    if x.dtype not in (torch.float32, torch.bfloat16):
        raise NativeLibraryError("l2norm_rows_fwd: x must be fp32 or bf16")
    _req(x, x.dtype, "x"); _req(y, torch.bfloat16, "y"); _req(inv_norm, torch.float32, "inv_norm")
    _check(lib().mmf_l2norm_rows_fwd(_p(x), int(x.dtype == torch.float32), ldx, _p(y), ldy, _p(inv_norm), rows, D, C.c_float(eps),
                                     _stream()), "mmf_l2norm_rows_fwd")


def l2norm_rows_bwd(g, ldg, y, ldy, inv_norm, dx, lddx, rows, D):
    _req(g, torch.bfloat16, "g"); _req(y, torch.bfloat16, "y"); _req(dx, torch.bfloat16, "dx"); _req(inv_norm, torch.float32, "inv_norm")
    _check(lib().mmf_l2norm_rows_bwd(_p(g), ldg, _p(y), ldy, _p(inv_norm), _p(dx), lddx, rows, D, _stream()), "mmf_l2norm_rows_bwd")


def gather_rows2(a, b, idx, out, n, H):
    """out[r] = idx[r] < len(a) ? a[idx[r]] : b[idx[r] - len(a)]; a, b bf16 [*, H]."""
    _req(a, torch.bfloat16, "a"); _req(b, torch.bfloat16, "b"); _req(idx, torch.int64, "idx"); _req(out, torch.bfloat16, "out")
    _check(lib().mmf_gather_rows2(_p(a), C.c_int64(a.shape[0]), _p(b), C.c_int64(0 if b is None else b.shape[0]), _p(idx), _p(out), n, H,
                                  _stream()), "mmf_gather_rows2")


def ptr_scores_fwd(q, k, mask_add, out, ldo, B, T, N, HQ, scale):
    _req(q, torch.bfloat16, "q"); _req(k, torch.bfloat16, "k"); _req(mask_add, torch.float32, "mask_add"); _req(out, torch.float32, "out")
    _check(lib().mmf_ptr_scores_fwd(_p(q), _p(k), _p(mask_add), _p(out), ldo, B, T, N, HQ, C.c_float(scale), _stream()), "mmf_ptr_scores_fwd")


def ptr_scores_bwd(dscores, ldd, q, k, dq, dk, B, T, N, HQ, scale):
    _req(dscores, torch.float32, "dscores")
    for t, nme in ((q, "q"), (k, "k"), (dq, "dq"), (dk, "dk")):
        _req(t, torch.bfloat16, nme)
    _check(lib().mmf_ptr_scores_bwd(_p(dscores), ldd, _p(q), _p(k), _p(dq), _p(dk), B, T, N, HQ, C.c_float(scale), _stream()),
           "mmf_ptr_scores_bwd")


def bce_rowmask_fwd(scores, targets, row_weight, loss, count, rows, N):
    for t, nme in ((scores, "scores"), (targets, "targets"), (row_weight, "row_weight"), (loss, "loss"), (count, "count")):
        _req(t, torch.float32, nme)
    ws = torch.empty(lib().mmf_bce_rowmask_ws_floats(), dtype=torch.float32, device=scores.device)
    _check(lib().mmf_bce_rowmask_fwd(_p(scores), _p(targets), _p(row_weight), _p(loss), _p(count), _p(ws), rows, N, _stream()),
           "mmf_bce_rowmask_fwd")


def bce_rowmask_bwd(scores, targets, row_weight, count, gloss, dscores, rows, N):
    for t, nme in ((scores, "scores"), (targets, "targets"), (row_weight, "row_weight"), (count, "count"), (gloss, "gloss"),
                   (dscores, "dscores")):
        _req(t, torch.float32, nme)
    _check(lib().mmf_bce_rowmask_bwd(_p(scores), _p(targets), _p(row_weight), _p(count), _p(gloss), _p(dscores), rows, N, _stream()),
           "mmf_bce_rowmask_bwd")


def adamw_step(p, g, m, v, p16, n, seg_end, seg_wd, nseg, lr, beta1, beta2, eps, step, correct_bias, mode, grad_scale):
    for t, nme in ((p, "p"), (g, "g"), (m, "m"), (v, "v"), (seg_wd, "seg_wd")):
        _req(t, torch.float32, nme)
    _req(p16, torch.bfloat16, "p16"); _req(seg_end, torch.int64, "seg_end")
    _check(lib().mmf_adamw_step(_p(p), _p(g), _p(m), _p(v), _p(p16), C.c_int64(n), _p(seg_end), _p(seg_wd), nseg,
                                C.c_float(lr), C.c_float(beta1), C.c_float(beta2), C.c_float(eps), int(step),
                                int(correct_bias), int(mode), C.c_float(grad_scale), _stream()), "mmf_adamw_step")


def adamw_multi(items, beta1, beta2, eps, step, correct_bias, mode, grad_scale=1.0, norm_sq=None, max_norm=0.0, dev_state=None):
    """items: list of (p, g, m, v, mirror or None, lr, wd); fp32 contiguous tensors, any number (launched MT_MAX at a time); `g` may
    also be bf16 (a wire buffer of the data-parallel step, read directly).  `mirror` is the bf16 weight shadow or the fp32
    packed-bias slice that must follow the parameter."""
    for i0 in range(0, len(items), MT_MAX):
        chunk = items[i0:i0 + MT_MAX]
        d = AdamWMultiDesc()
        d.n = len(chunk)
        mask = 0
        for i, (p, g, m, v, p16, lr, wd) in enumerate(chunk):
            if g.dtype == torch.bfloat16:
                mask |= 1 << i
            elif g.dtype != torch.float32:
                raise NativeLibraryError("adamw_multi: gradients must be fp32 or bf16, got %s" % g.dtype)
            if g.numel() != p.numel() or not g.is_contiguous():
                raise NativeLibraryError("adamw_multi: gradient must be contiguous with the parameter's element count")
            d.p[i], d.g[i], d.m[i], d.v[i] = p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr()
            d.p16[i] = p16.data_ptr() if (p16 is not None and p16.dtype == torch.bfloat16) else None
            d.p32[i] = p16.data_ptr() if (p16 is not None and p16.dtype == torch.float32) else None
            d.numel[i], d.lr[i], d.wd[i] = p.numel(), lr, wd
        d.beta1, d.beta2, d.eps = beta1, beta2, eps
        d.step, d.correct_bias, d.mode = int(step), int(correct_bias), int(mode)
        d.grad_scale = grad_scale
        d.norm_sq = norm_sq.data_ptr() if norm_sq is not None else None
        d.max_norm = max_norm
        d.dev_state = dev_state.data_ptr() if dev_state is not None else None
        d.g_bf16_mask = mask
        _check(lib().mmf_adamw_multi(C.byref(d), _stream()), "mmf_adamw_multi")


def optim_state_advance(state, schedule=0, warmup_steps=0.0, total_steps=0.0):
    _req(state, torch.float32, "state")
    _check(lib().mmf_optim_state_advance(_p(state), int(schedule), C.c_float(warmup_steps), C.c_float(total_steps), _stream()),
           "mmf_optim_state_advance")


def l2norm_sq_multi(tensors, out):
    """out[0] = sum over all tensors of sum(x^2) (fp32 tensors), deterministic."""
    _req(out, torch.float32, "out")
    for i0 in range(0, len(tensors), MT_MAX):
        chunk = tensors[i0:i0 + MT_MAX]
        d = TensorList()
        d.n = len(chunk)
        for i, t in enumerate(chunk):
            _req(t, torch.float32, "tensor")
            d.ptr[i], d.numel[i] = t.data_ptr(), t.numel()
        ws = torch.empty(lib().mmf_l2norm_sq_ws_floats(C.byref(d)), dtype=torch.float32, device=out.device)
        _check(lib().mmf_l2norm_sq_multi(C.byref(d), _p(out), int(i0 > 0), _p(ws), _stream()), "mmf_l2norm_sq_multi")


class OffsetList(C.Structure):
    _fields_ = [("off", C.c_int64 * MT_MAX)]


def pack_f32_multi(tensors, offsets, dst, scale=1.0):
    """dst[offsets[t] : offsets[t] + tensors[t].numel()] = scale * tensors[t] (fp32 sources; dst fp32 or bf16), MT_MAX tensors per launch."""
    if dst.dtype not in (torch.float32, torch.bfloat16):
        raise NativeLibraryError("pack_f32_multi: dst must be fp32 or bf16")
    for i0 in range(0, len(tensors), MT_MAX):
        chunk = tensors[i0:i0 + MT_MAX]
        d, o = TensorList(), OffsetList()
        d.n = len(chunk)
        for i, t in enumerate(chunk):
            _req(t, torch.float32, "tensor")
            if not t.is_contiguous():
                raise NativeLibraryError("pack_f32_multi: contiguous gradients expected")
            d.ptr[i], d.numel[i], o.off[i] = t.data_ptr(), t.numel(), int(offsets[i0 + i])
        _check(lib().mmf_pack_f32_multi(C.byref(d), C.byref(o), _p(dst), int(dst.dtype == torch.bfloat16), C.c_float(scale), _stream()),
               "mmf_pack_f32_multi")


def transpose_multi(pairs):
    """dst = src^T for every (src bf16 [R, C], dst bf16 [C, R]) pair; R, C multiples of 64; MT_MAX matrices per launch."""
    for i0 in range(0, len(pairs), MT_MAX):
        chunk = pairs[i0:i0 + MT_MAX]
        d = TransposeList()
        d.n = len(chunk)
        for i, (src, dst) in enumerate(chunk):
            _req(src, torch.bfloat16, "src"); _req(dst, torch.bfloat16, "dst")
            if not (src.is_contiguous() and dst.is_contiguous()) or tuple(dst.shape) != (src.shape[1], src.shape[0]):
                raise NativeLibraryError("transpose_multi: contiguous [R, C] -> [C, R] pairs expected")
            d.src[i], d.dst[i], d.rows[i], d.cols[i] = src.data_ptr(), dst.data_ptr(), src.shape[0], src.shape[1]
        _check(lib().mmf_transpose_bf16_multi(C.byref(d), _stream()), "mmf_transpose_bf16_multi")


def probe_mfma16(a, b, d):
    _check(lib().mmf_probe_mfma16(_p(a), _p(b), _p(d), _stream()), "mmf_probe_mfma16")


def probe_mfma32(a, b, d):
    _check(lib().mmf_probe_mfma32(_p(a), _p(b), _p(d), _stream()), "mmf_probe_mfma32")


def probe_tr16(img, addr, out):
    _check(lib().mmf_probe_tr16(_p(img), _p(addr), _p(out), _stream()), "mmf_probe_tr16")
