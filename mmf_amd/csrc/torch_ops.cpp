// mmf_amd :: native PyTorch operator library  (libmmf_amd_ops.so;  torch.ops.load_library(...) -> torch.ops.mmf_amd.*)
//
// SURVEY.md section 8(b), last row: MMF has no FFI on this path, so what a native replacement exports is PyTorch's custom-op ABI — a
// shared library of `mmf_amd::name(Tensor ...) -> Tensor` operators over at::Tensor on the CURRENT HIP stream, with registered
// autograd, TORCH_CHECK argument errors, no hidden synchronisation and no allocation outside the caching allocator.  This file is
// that library: TORCH_LIBRARY(mmf_amd) + torch::autograd::Function nodes whose forward AND backward enqueue the hand-written gfx950
// kernels of libmmf_amd.so through the C ABI of include/mmf_amd.h.  A scripted / saved VisualBERT (the reference's own test scripts
// its model: tests/models/test_visual_bert.py:40-49) runs after `torch.ops.load_library` alone, and the eager loop MMF's trainer
// drives (`model(batch)`; `loss.backward()`, mmf/trainers/core/training_loop.py:199-231) costs ~60 host calls per step instead
// of ~450 ctypes launches.
//
//   operator                              replaces (reference file:line)
//   mmf_amd::transformer_layer            BertLayerJit.forward                        mmf/modules/hf_layers.py:255-292
//   mmf_amd::visio_linguistic_embeddings  BertVisioLinguisticEmbeddings.forward       mmf/modules/embeddings.py:423-459
//   mmf_amd::additive_mask                (1 - mask) * -10000                          mmf/models/visual_bert.py:94-106
//   mmf_amd::gather_rows                  `vqa` pooling gather + dropout              mmf/models/visual_bert.py:389-400
//   mmf_amd::dense_gelu                   HF BertIntermediate / head transform dense   hf_layers.py:289, visual_bert.py:328
//   mmf_amd::layer_norm                   nn.LayerNorm                                 visual_bert.py:328
//   mmf_amd::linear                       nn.Linear                                    visual_bert.py:330
//   mmf_amd::linear_tanh                  HF BertPooler                                visual_bert.py:146
//   mmf_amd::dropout                      nn.Dropout                                   visual_bert.py:400
//   mmf_amd::pair_halves                  nlvr2 pooled-output pairing                  visual_bert.py:369-374
//   mmf_amd::logit_bce                    LogitBinaryCrossEntropy                      mmf/modules/losses.py:225-251
//   mmf_amd::masked_lm_head               tied decoder + masked-LM CrossEntropyLoss     mmf/models/visual_bert.py:267-277
//   mmf_amd::masked_region_head           image-prediction decoder + masked KLDivLoss  mmf/models/vilbert.py:846-858, 1150-1157
//
// State the operators need lives here, not in Python: the bf16 weight shadows (+ W^T twins) of the fp32 master parameters, the
// per-site dropout keys (torch's Philox offset in eager mode, a device seed word under hipGraph capture) and the deferred LayerNorm
// parameter-gradient reductions.  The Python package drives the same state through the `_`-prefixed service operators at the end of
// this file (mmf_amd/functional.py is then a thin proxy), so both always see ONE cache.
//
// Python-only modes stay reachable: with the fp32-accurate forward path switched on (`mmf_amd.fp32_inference()`), or with one of the
// opt-in experiment hooks of mmf_amd/utils/graph.py active, an operator forwards to its `_py_<name>` twin, whose schema is defined
// here and whose kernel the Python package registers (torch.library IMPL).  Without the package those modes cannot be entered.
#include <ATen/ATen.h>
#include <ATen/core/Generator.h>
#include <c10/hip/HIPStream.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>
#include <torch/autograd.h>
#include <torch/library.h>

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <optional>
#include <set>
#include <string>
#include <unordered_map>
#include <vector>

#include "mmf_amd.h"

namespace {

using at::Tensor;
using std::optional;
using torch::autograd::AutogradContext;
using torch::autograd::variable_list;

// ---------------------------------------------------------------------------------------------------------------------------------
// plumbing
// ---------------------------------------------------------------------------------------------------------------------------------
inline void* sp() { return reinterpret_cast<void*>(c10::hip::getCurrentHIPStream().stream()); }
inline void* P(const Tensor& t) { return t.defined() ? t.data_ptr() : nullptr; }
inline const float* PF(const Tensor& t) { return t.defined() ? t.data_ptr<float>() : nullptr; }

#define MMF_RC(call, what) TORCH_CHECK((call) == 0, what, " failed: ", mmf_amd_last_error())

inline void req(const Tensor& t, at::ScalarType dt, const char* name) {
    if (!t.defined()) return;
    TORCH_CHECK(t.is_cuda(), "mmf_amd: `", name, "` must live in HBM (got a ", t.device(), " tensor); there is no CPU path");
    TORCH_CHECK(t.scalar_type() == dt, "mmf_amd: `", name, "` must be ", dt, ", got ", t.scalar_type());
    TORCH_CHECK(t.is_contiguous() || t.stride(-1) == 1, "mmf_amd: `", name, "` must be row-contiguous");
}
inline Tensor empty_bf16(at::IntArrayRef s, const Tensor& like) { return at::empty(s, like.options().dtype(at::kBFloat16)); }
inline Tensor empty_f32(at::IntArrayRef s, const Tensor& like) { return at::empty(s, like.options().dtype(at::kFloat)); }
inline int pad8(int64_t n) { return (int)((n + 7) / 8 * 8); }

// token-major bf16 view [rows, features] of an activation (fp32 inputs are cast once)
Tensor as_bf16_2d(const Tensor& x) {
    Tensor x2 = x.reshape({-1, x.size(-1)});
    if (x2.scalar_type() != at::kBFloat16) {
        if (x2.scalar_type() != at::kFloat) x2 = x2.to(at::kFloat);
        x2 = x2.contiguous();
        req(x2, at::kFloat, "x");
        Tensor out = empty_bf16(x2.sizes(), x2);
        MMF_RC(mmf_cast_f32_to_bf16(x2.data_ptr<float>(), out.data_ptr(), x2.numel(), sp()), "mmf_cast_f32_to_bf16");
        return out;
    }
    req(x2, at::kBFloat16, "x");
    return x2.is_contiguous() ? x2 : x2.contiguous();
}
// bf16, contiguous, ld == cols version of an incoming gradient
Tensor grad_bf16(const Tensor& g, int64_t cols) {
    Tensor g2 = g.reshape({-1, cols});
    if (g2.scalar_type() == at::kBFloat16 && g2.is_contiguous()) return g2;
    return as_bf16_2d(g2);
}

// ---------------------------------------------------------------------------------------------------------------------------------
// dropout keys  (mmf_amd/functional.py::_DropoutKeys keeps the documentation; this is the implementation both sides use)
// ---------------------------------------------------------------------------------------------------------------------------------
struct Drop {
    uint32_t key = 0, thr16 = 0;
    float scale = 1.f;
    Tensor seed;       // optional int32[1] device word mixed into the key at run time (hipGraph replays)
    const uint32_t* seed_ptr() const { return seed.defined() ? reinterpret_cast<const uint32_t*>(seed.data_ptr()) : nullptr; }
    bool on() const { return thr16 != 0; }
};
struct DropKeys {
    Tensor graph_seed;
    int64_t counter = 0;
    static uint32_t mix(uint64_t a, uint64_t b) {
        uint64_t x = a * 0x9E3779B97F4A7C15ull + (b + 1) * 0xBF58476D1CE4E5B9ull;
        x ^= x >> 31;
        return (uint32_t)(((x * 0x94D049BB133111EBull) >> 16) & 0xFFFFFFFFull);
    }
    uint32_t next(Tensor& seed_out) {
        if (graph_seed.defined()) {
            ++counter;
            seed_out = graph_seed;
            return mix(0x5EED, (uint64_t)counter);
        }
        at::Generator gen = at::globalContext().defaultGenerator(c10::Device(c10::kCUDA, c10::hip::current_device()));
        std::lock_guard<std::mutex> lock(gen.mutex());
        const uint64_t seed = gen.current_seed(), off = gen.get_offset();
        gen.set_offset(off + 4);       // advanced like any torch random op: torch.manual_seed reproduces the masks
        seed_out = Tensor();
        return mix(seed, off);
    }
};
DropKeys g_keys;

Drop make_drop(double p, bool training) {
    Drop d;
    if (!training || !(p > 0.0)) return d;
    long thr = std::lround(p * 65536.0);
    thr = thr < 1 ? 1 : (thr > 65535 ? 65535 : thr);
    d.key = g_keys.next(d.seed);
    d.thr16 = (uint32_t)thr;
    d.scale = (float)(1.0 / (1.0 - (double)thr / 65536.0));
    return d;
}
// (key, thr16, scale-bits) <-> IValue for AutogradContext::saved_data
c10::IValue drop_pack(const Drop& d) {
    uint32_t sb;
    std::memcpy(&sb, &d.scale, 4);
    return c10::IValue(std::vector<int64_t>{(int64_t)d.key, (int64_t)d.thr16, (int64_t)sb});
}
Drop drop_unpack(const c10::IValue& v, const Tensor& seed) {
    auto l = v.toIntVector();
    Drop d;
    d.key = (uint32_t)l[0]; d.thr16 = (uint32_t)l[1];
    const uint32_t sb = (uint32_t)l[2];
    std::memcpy(&d.scale, &sb, 4);
    d.seed = seed;
    return d;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// bf16 shadows of fp32 master parameters (+ transposed twins for the input-gradient GEMMs)
// ---------------------------------------------------------------------------------------------------------------------------------
struct Sig {
    const void* impl; int64_t version; const void* ptr;
    bool operator==(const Sig& o) const { return impl == o.impl && version == o.version && ptr == o.ptr; }
};
struct ShadowEntry {
    std::vector<Sig> sig;
    Tensor buf;
    bool f32 = false;
    std::vector<c10::weak_intrusive_ptr<c10::TensorImpl>> owners;   // keeps the owners' addresses from being reused while cached
    Tensor twin;
    std::vector<Sig> twin_sig;
};
class Shadows {
public:
    void clear() { store_.clear(); slot_.clear(); by_ptr_.clear(); }

    Tensor get(const std::vector<Tensor>& params, bool want_f32) {
        TORCH_CHECK(!params.empty(), "mmf_amd: shadow of no parameter");
        purge_dead();
        const void* key = params[0].unsafeGetTensorImpl();
        std::vector<Sig> sig;
        sig.reserve(params.size());
        for (const Tensor& p : params) sig.push_back(Sig{p.unsafeGetTensorImpl(), (int64_t)p._version(), p.data_ptr()});
        auto it = store_.find(key);
        if (it != store_.end() && it->second.sig == sig && it->second.f32 == want_f32) return it->second.buf;
        int64_t rows = 0;
        for (const Tensor& p : params) rows += p.size(0);
        std::vector<int64_t> shape(params[0].sizes().begin(), params[0].sizes().end());
        shape[0] = rows;
        ShadowEntry& e = store_[key];
        const auto dt = want_f32 ? at::kFloat : at::kBFloat16;
        if (!e.buf.defined() || e.buf.sizes() != at::IntArrayRef(shape) || e.buf.scalar_type() != dt) {
            if (e.buf.defined()) by_ptr_.erase(e.buf.data_ptr());
            e.buf = at::empty(shape, params[0].options().dtype(dt).requires_grad(false));
            e.twin = Tensor();
        }
        int64_t r = 0;
        e.owners.clear();
        for (const Tensor& p : params) {
            const int64_t n = p.size(0);
            Tensor src = p.detach();
            if (!src.is_contiguous()) src = src.contiguous();
            req(src, at::kFloat, "parameter");
            Tensor dst = e.buf.narrow(0, r, n);
            if (!want_f32) { MMF_RC(mmf_cast_f32_to_bf16(src.data_ptr<float>(), dst.data_ptr(), src.numel(), sp()), "mmf_cast_f32_to_bf16"); }
            else dst.copy_(src);
            if (!want_f32 || params.size() > 1) slot_[p.unsafeGetTensorImpl()] = {key, r};
            e.owners.emplace_back(c10::weak_intrusive_ptr<c10::TensorImpl>(p.getIntrusivePtr()));
            r += n;
        }
        e.sig = std::move(sig);
        e.f32 = want_f32;
        if (!want_f32) by_ptr_[e.buf.data_ptr()] = key;
        return e.buf;
    }

    // mirror rows of parameter p (bf16 weight shadow, or its slice of a packed fp32 Q|K|V bias) if up to date, else undefined
    Tensor slot(const Tensor& p) {
        auto it = slot_.find(p.unsafeGetTensorImpl());
        if (it == slot_.end()) return Tensor();
        auto st = store_.find(it->second.first);
        if (st == store_.end()) return Tensor();
        for (const Sig& s : st->second.sig)
            if (s.impl == p.unsafeGetTensorImpl() && (s.version != (int64_t)p._version() || s.ptr != p.data_ptr())) return Tensor();
        return st->second.buf.narrow(0, it->second.second, p.size(0));
    }

    // W^T [in, out] (bf16) of the weight shadow w16 [out, in] when it is a whole tracked shadow with dims % 64 == 0
    bool twins_on = true;      // (tests/test_dgrad_nt_gpu.py flips it through _shadow_set_twins: functional.DGRAD_NT)
    Tensor transposed(const Tensor& w16) {
        if (!twins_on || !w16.defined() || w16.dim() != 2) return Tensor();
        auto k = by_ptr_.find(w16.data_ptr());
        if (k == by_ptr_.end()) return Tensor();
        auto st = store_.find(k->second);
        if (st == store_.end()) return Tensor();
        ShadowEntry& e = st->second;
        if (e.f32 || e.buf.data_ptr() != w16.data_ptr() || e.buf.sizes() != w16.sizes()) return Tensor();
        const int64_t R = e.buf.size(0), C = e.buf.size(1);
        if (R % 64 || C % 64) return Tensor();
        if (e.twin.defined() && e.twin_sig == e.sig) return e.twin;
        if (!e.twin.defined() || e.twin.size(0) != C || e.twin.size(1) != R) e.twin = at::empty({C, R}, e.buf.options());
        mmf_transpose_list l;
        l.n = 1; l.src[0] = e.buf.data_ptr(); l.dst[0] = e.twin.data_ptr(); l.rows[0] = (int)R; l.cols[0] = (int)C;
        MMF_RC(mmf_transpose_bf16_multi(&l, sp()), "mmf_transpose_bf16_multi");
        e.twin_sig = e.sig;
        return e.twin;
    }

    // re-transpose every live twin from its (already updated in place) shadow; `only` / `skip`: head parameters to restrict to / leave out
    void refresh_transposed(const std::vector<Tensor>& only, bool has_only, const std::vector<Tensor>& skip, bool has_skip) {
        auto in = [](const std::vector<Tensor>& v, const void* k) {
            for (const Tensor& t : v) if (t.unsafeGetTensorImpl() == k) return true;
            return false;
        };
        mmf_transpose_list l;
        l.n = 0;
        for (auto& kv : store_) {
            ShadowEntry& e = kv.second;
            if (!e.twin.defined() || !(e.twin_sig == e.sig)) continue;
            if (has_only && !in(only, kv.first)) continue;
            if (has_skip && in(skip, kv.first)) continue;
            l.src[l.n] = e.buf.data_ptr(); l.dst[l.n] = e.twin.data_ptr(); l.rows[l.n] = (int)e.buf.size(0); l.cols[l.n] = (int)e.buf.size(1);
            if (++l.n == MMF_MT_MAX) { MMF_RC(mmf_transpose_bf16_multi(&l, sp()), "mmf_transpose_bf16_multi"); l.n = 0; }
        }
        if (l.n) MMF_RC(mmf_transpose_bf16_multi(&l, sp()), "mmf_transpose_bf16_multi");
    }

private:
    void purge_dead() {
        if (++calls_ % 4096) return;
        for (auto it = store_.begin(); it != store_.end();) {
            bool dead = it->second.owners.empty();
            for (auto& w : it->second.owners) dead = dead || w.expired();
            if (dead) {
                if (it->second.buf.defined()) by_ptr_.erase(it->second.buf.data_ptr());
                for (auto s = slot_.begin(); s != slot_.end();) s = (s->second.first == it->first) ? slot_.erase(s) : std::next(s);
                it = store_.erase(it);
            } else ++it;
        }
    }
    std::unordered_map<const void*, ShadowEntry> store_;
    std::unordered_map<const void*, std::pair<const void*, int64_t>> slot_;
    std::unordered_map<const void*, const void*> by_ptr_;
    uint64_t calls_ = 0;
};
Shadows g_shadows;

// Input-gradient GEMMs run on the wide forward-form tiles when their weight keeps a W^T twin (mmf_amd/functional.py::_twin_pays)
inline bool twin_pays(int64_t out_width) { return (out_width % 96 == 0 && out_width <= 1152) || out_width % 128 == 0; }

// ---------------------------------------------------------------------------------------------------------------------------------
// kernel call helpers
// ---------------------------------------------------------------------------------------------------------------------------------
struct Gemm {
    mmf_gemm_desc d;
    Gemm(const Tensor& A, const Tensor& B, const Tensor& C, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldc) {
        std::memset(&d, 0, sizeof(d));
        TORCH_CHECK(A.is_cuda() && B.is_cuda() && C.is_cuda(), "mmf_amd: GEMM operands must live in HBM; there is no CPU path");
        d.A = A.data_ptr(); d.B = B.data_ptr(); d.C = C.data_ptr();
        d.M = (int)M; d.N = (int)N; d.K = (int)K; d.lda = (int)lda; d.ldb = (int)ldb; d.ldc = (int)ldc;
        d.a_f32 = A.scalar_type() == at::kFloat; d.b_f32 = B.scalar_type() == at::kFloat; d.out_f32 = C.scalar_type() == at::kFloat;
        d.drop_scale = 1.f;
    }
    Gemm& kmajor(bool a, bool b) { d.a_kmajor = a; d.b_kmajor = b; return *this; }
    Gemm& bias(const Tensor& t) { req(t, at::kFloat, "bias"); d.bias = PF(t); return *this; }
    Gemm& coladd(const Tensor& t) { req(t, at::kFloat, "coladd"); d.coladd = PF(t); return *this; }
    Gemm& rowtab(const Tensor& tab, const Tensor& idx, int ld) {
        req(tab, at::kFloat, "rowtab"); req(idx, at::kLong, "rowidx");
        d.rowtab = PF(tab); d.rowidx = idx.data_ptr<int64_t>(); d.rowtab_ld = ld; return *this;
    }
    Gemm& act(int a, const Tensor& U = Tensor(), const Tensor& aux = Tensor()) { d.act = a; d.U = P(U); d.aux = P(aux); return *this; }
    Gemm& resid(const Tensor& r, int64_t ldr) { if (r.defined()) { req(r, at::kBFloat16, "resid"); d.resid = r.data_ptr(); d.ldr = (int)ldr; } return *this; }
    Gemm& drop(const Drop& dr) { d.drop_key = dr.key; d.drop_thr16 = dr.thr16; d.drop_scale = dr.scale; d.drop_seed = dr.seed_ptr(); return *this; }
    Gemm& grp(int in, int pad, int off) { d.grp_in = in; d.grp_pad = pad; d.grp_off = off; return *this; }
    Gemm& rowsum(const Tensor& t) { if (t.defined()) d.rowsum_out = t.data_ptr<float>(); return *this; }
    Gemm& site(int s) { d.debug_flags |= MMF_GEMM_SITE(s); return *this; }      // names the call for MMF_TUN_NT_SITE_KEEP (no effect by itself)
    void run() {
        Tensor ws;
        if (d.out_f32 && d.a_kmajor && d.b_kmajor && !d.bias && !d.resid && d.act == 0) {
            const int sp_ = mmf_gemm_splitk_splits(d.M, d.N, d.K);
            if (sp_ > 1) {
                ws = at::empty({(int64_t)sp_ * d.M * (d.N + 1)}, at::TensorOptions().dtype(at::kFloat).device(c10::Device(c10::kCUDA, c10::hip::current_device())));
                d.splitk_ws = ws.data_ptr(); d.splitk_ws_bytes = ws.numel() * 4;
            }
        }
        if (!ws.defined() && !d.rowsum_out) {      // skinny problems (the heads): K-slices over the chip, the epilogue runs on the slab sums
            const int sp_ = mmf_gemm_skinny_splits(d.M, d.N, d.K, d.a_kmajor);
            if (sp_ > 1) {
                ws = at::empty({(int64_t)sp_ * d.M * ((d.N + 7) / 8 * 8)}, at::TensorOptions().dtype(at::kFloat).device(c10::Device(c10::kCUDA, c10::hip::current_device())));
                d.splitk_ws = ws.data_ptr(); d.splitk_ws_bytes = ws.numel() * 4;
            }
        }
        MMF_RC(mmf_gemm_bf16(&d, sp()), "mmf_gemm_bf16");
    }
};

Tensor colsum(const Tensor& x, int64_t ld, int64_t rows, int64_t N) {
    Tensor out = empty_f32({N}, x), ws = empty_f32({(int64_t)mmf_colsum_ws_floats((int)N)}, x);
    MMF_RC(mmf_colsum_bf16(x.data_ptr(), (int)ld, 1, (int)rows, 0, (int)N, out.data_ptr<float>(), 0.f, ws.data_ptr<float>(), sp()), "mmf_colsum_bf16");
    return out;
}

// dX [M, K] = dY [M, N] W [N, K], residual-gradient add / saved-gelu' multiply fused
Tensor dgrad(const Tensor& dy, int64_t ldy, const Tensor& w16, int64_t M, int64_t N, int64_t K, const Tensor& dx_resid = Tensor(),
             const Tensor& act_aux = Tensor(), int site = 0) {
    Tensor dx = empty_bf16({M, K}, dy);
    Tensor wt = (N % 8 == 0 && twin_pays(K)) ? g_shadows.transposed(w16) : Tensor();
    if (wt.defined()) Gemm(dy, wt, dx, M, K, N, ldy, N, K).resid(dx_resid, K).act(act_aux.defined() ? 2 : 0, Tensor(), act_aux).site(site).run();
    else Gemm(dy, w16, dx, M, K, N, ldy, K, K).kmajor(false, true).resid(dx_resid, K).act(act_aux.defined() ? 2 : 0, Tensor(), act_aux).site(site).run();
    return dx;
}
struct LinBwd { Tensor dx, dw, db; };
// dy [M,N] bf16 (row stride ldy, pad columns zero), x [M,K] bf16 (or fp32 features), w16 [N,K]
LinBwd linear_bwd(const Tensor& dy, int64_t ldy, const Tensor& x, const Tensor& w16, int64_t M, int64_t N, int64_t K, bool need_dx,
                  const Tensor& dx_resid, const Tensor& act_aux, bool want_db) {
    LinBwd r;
    if (need_dx) r.dx = dgrad(dy, ldy, w16, M, N, K, dx_resid, act_aux);
    r.dw = empty_f32({N, K}, dy);
    const bool fused = want_db && x.scalar_type() == at::kBFloat16;
    if (fused) r.db = empty_f32({N}, dy);
    Gemm(dy, x, r.dw, N, K, M, ldy, x.stride(0), K).kmajor(true, true).rowsum(r.db).run();
    if (want_db && !r.db.defined()) r.db = colsum(dy, ldy, M, N);
    return r;
}

// ---- LayerNorm backward with deferred parameter-gradient reductions ---------------------------------------------------------------
struct LnPending { Tensor ws; int rows, H; Tensor dgamma, dbeta; };
bool g_ln_defer = false;
bool g_wgrad_hold = true;             // A/B switch of the held weight gradients (_wgrad_hold_set)
std::vector<LnPending> g_ln_pending;
// A fused layer's grouped weight gradients held back until the NEXT LayerNorm backward of the same stream (inside a deferral block only: the caller owns the
// whole backward and reads parameter gradients after `_ln_defer_flush`): that LayerNorm — the first kernel of the layer below, independent of the weight
// gradients — rides on the CUs the gradient tiles leave idle (mmf_gemm_bf16_grouped_ln).  Anything else that ends the wait launches the group plainly.
struct WgradHeld { mmf_gemm_desc g[4]; std::vector<Tensor> keep; void* stream; };
std::vector<WgradHeld> g_wgrad_held;
int64_t g_wgrad_joint = 0;                 // joint launches so far (tests read it through _wgrad_joint_launches)
std::set<const void*> g_wgrad_seen;       // weights with a gradient produced in this deferral block (a layer applied twice is never held: autograd sums at once)
void wgrad_release(void* stream, bool all) {
    for (size_t i = 0; i < g_wgrad_held.size();) {
        if (all || g_wgrad_held[i].stream == stream) {
            if (all && g_wgrad_held[i].stream != sp()) {
                // another stream's group (ViLBERT's visual stream) ends the block on the caller's stream, after the autograd engine has joined the streams: its
                // buffers belong to that stream's pool of the caching allocator, which must not hand them out again while this launch is pending
                hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
                (void)hipStreamIsCapturing(reinterpret_cast<hipStream_t>(sp()), &st);
                if (st == hipStreamCaptureStatusNone)
                    for (auto& t : g_wgrad_held[i].keep) t.record_stream(c10::hip::getCurrentHIPStreamMasqueradingAsCUDA());
            }
            MMF_RC(mmf_gemm_bf16_grouped(g_wgrad_held[i].g, 4, all ? sp() : stream), "mmf_gemm_bf16_grouped");
            g_wgrad_held.erase(g_wgrad_held.begin() + i);
        } else ++i;
    }
}
void ln_flush() {
    wgrad_release(nullptr, true);
    g_wgrad_seen.clear();
    mmf_ln_reduce_list l;
    l.n = 0;
    auto go = [&]() { if (l.n) { MMF_RC(mmf_layernorm_bwd_reduce_multi(&l, sp()), "mmf_layernorm_bwd_reduce_multi"); l.n = 0; } };
    for (auto& p : g_ln_pending) {
        l.partials[l.n] = p.ws.data_ptr<float>(); l.rows[l.n] = p.rows; l.H[l.n] = p.H;
        l.dgamma[l.n] = p.dgamma.data_ptr<float>(); l.dbeta[l.n] = p.dbeta.data_ptr<float>();
        if (++l.n == MMF_MT_MAX) go();
    }
    go();
    g_ln_pending.clear();
}
struct LnBwd { Tensor dx, dlin, dgamma, dbeta, dbias; };
LnBwd ln_bwd(const Tensor& dy, const Tensor& y, const Tensor& mean, const Tensor& rstd, const Tensor& gamma, const Drop& drop, bool want_dbias) {
    const int64_t M = y.size(0), N = y.size(1);
    LnBwd r;
    r.dx = empty_bf16({M, N}, y);
    Tensor dlin = drop.on() ? empty_bf16({M, N}, y) : Tensor();
    r.dgamma = empty_f32({N}, y); r.dbeta = empty_f32({N}, y);
    if (want_dbias) r.dbias = empty_f32({N}, y);
    Tensor ws = empty_f32({(int64_t)mmf_layernorm_bwd_ws_floats((int)N)}, y);
    const bool defer = g_ln_defer && !want_dbias && mmf_layernorm_bwd_deferrable((int)M, (int)N);
    WgradHeld* held = nullptr;
    for (auto& h : g_wgrad_held) if (h.stream == sp()) held = &h;
    if (held && defer) {
        mmf_ln_bwd_desc ld;
        ld.dy = dy.data_ptr(); ld.x = y.data_ptr(); ld.mean = PF(mean); ld.rstd = PF(rstd); ld.gamma = PF(gamma); ld.dx = r.dx.data_ptr(); ld.dlin = P(dlin);
        ld.drop_key = drop.key; ld.drop_thr16 = drop.thr16; ld.drop_scale = drop.scale; ld.drop_seed = drop.seed_ptr(); ld.partials = ws.data_ptr<float>();
        ld.rows = (int)M; ld.H = (int)N;
        MMF_RC(mmf_gemm_bf16_grouped_ln(held->g, 4, &ld, sp()), "mmf_gemm_bf16_grouped_ln");
        ++g_wgrad_joint;
        g_wgrad_held.erase(g_wgrad_held.begin() + (held - g_wgrad_held.data()));
    } else {
        if (held) wgrad_release(sp(), false);
        MMF_RC(mmf_layernorm_bwd(dy.data_ptr(), y.data_ptr(), PF(mean), PF(rstd), PF(gamma), r.dx.data_ptr(), P(dlin), drop.key, drop.thr16, drop.scale,
                                 drop.seed_ptr(), defer ? nullptr : r.dgamma.data_ptr<float>(), defer ? nullptr : r.dbeta.data_ptr<float>(),
                                 defer ? nullptr : (float*)P(r.dbias), 0, ws.data_ptr<float>(), (int)M, (int)N, sp()), "mmf_layernorm_bwd");
    }
    if (defer) g_ln_pending.push_back(LnPending{ws, (int)M, (int)N, r.dgamma, r.dbeta});
    r.dlin = dlin.defined() ? dlin : r.dx;
    return r;
}

// backward of dropout(LayerNorm(y)) with the dropout backward folded into the load of dy (embeddings.py:343-345)
LnBwd ln_bwd_din(const Tensor& dy, const Tensor& y, const Tensor& mean, const Tensor& rstd, const Tensor& gamma, const Drop& in_drop) {
    const int64_t M = y.size(0), N = y.size(1);
    LnBwd r;
    r.dx = empty_bf16({M, N}, y);
    r.dgamma = empty_f32({N}, y); r.dbeta = empty_f32({N}, y);
    Tensor ws = empty_f32({(int64_t)mmf_layernorm_bwd_ws_floats((int)N)}, y);
    const bool defer = g_ln_defer && mmf_layernorm_bwd_deferrable((int)M, (int)N);
    MMF_RC(mmf_layernorm_bwd_din(dy.data_ptr(), y.data_ptr(), PF(mean), PF(rstd), PF(gamma), r.dx.data_ptr(), in_drop.key, in_drop.thr16, in_drop.scale,
                                 in_drop.seed_ptr(), defer ? nullptr : r.dgamma.data_ptr<float>(), defer ? nullptr : r.dbeta.data_ptr<float>(), 0,
                                 ws.data_ptr<float>(), (int)M, (int)N, sp()), "mmf_layernorm_bwd_din");
    if (defer) g_ln_pending.push_back(LnPending{ws, (int)M, (int)N, r.dgamma, r.dbeta});
    r.dlin = r.dx;
    return r;
}

// dense -> dropout -> (+ residual) -> LayerNorm   (HF BertSelfOutput / BertOutput)
struct Ddrln { Tensor out, y, mean, rstd; };
Ddrln ddrln_fwd(const Tensor& h2, const Tensor& resid2, const Tensor& w16, const Tensor& bias, const Tensor& gamma, const Tensor& beta, double eps,
                const Drop& drop, int site = 0) {
    const int64_t M = h2.size(0), K = h2.size(1), N = w16.size(0);
    Ddrln r;
    r.y = empty_bf16({M, N}, h2);
    Gemm(h2, w16, r.y, M, N, K, K, K, N).bias(bias).resid(resid2, N).drop(drop).site(site).run();
    r.out = empty_bf16({M, N}, h2); r.mean = empty_f32({M}, h2); r.rstd = empty_f32({M}, h2);
    req(gamma, at::kFloat, "LayerNorm.weight"); req(beta, at::kFloat, "LayerNorm.bias");
    MMF_RC(mmf_layernorm_fwd(r.y.data_ptr(), PF(gamma), PF(beta), r.out.data_ptr(), r.mean.data_ptr<float>(), r.rstd.data_ptr<float>(), (int)M, (int)N,
                             (float)eps, sp()), "mmf_layernorm_fwd");
    return r;
}

void attn_desc(mmf_attn_desc& d, const Tensor& qkv, int64_t H, const Tensor& mask, const Tensor& ctx, const Tensor& lse, const Tensor& o32, int64_t B,
               int64_t heads, int64_t S, const Drop& drop, int64_t tail) {
    std::memset(&d, 0, sizeof(d));
    char* base = reinterpret_cast<char*>(qkv.data_ptr());
    d.q = base; d.k = base + 2 * H; d.v = base + 4 * H;       // bf16: H elements = 2H bytes
    d.ldq = d.ldk = d.ldv = (int)(3 * H);
    if (mask.defined()) {
        req(mask, at::kFloat, "attention mask"); d.mask = mask.data_ptr<float>();
        if (mask.dim() == 4) {      // one [S, S] mask per (sample, head), [B, heads, S, S] (mmf_attn_desc.mask_head_stride)
            TORCH_CHECK(mask.size(0) == B && mask.size(1) == heads && mask.size(2) == S && mask.size(3) == S && mask.is_contiguous(),
                        "mmf_amd: a per-head attention mask must be a contiguous [B, heads, S, S] tensor");
            d.mask_query_stride = (int)S;
            d.mask_head_stride = (int)(S * S);
        } else if (mask.dim() == 3) {      // materialised additive mask per (query, key) pair, [B, S, S] (mmf_attn_desc.mask_query_stride)
            TORCH_CHECK(mask.size(0) == B && mask.size(1) == S && mask.size(2) == S && mask.is_contiguous(), "mmf_amd: a per-query attention mask must be a contiguous [B, S, S] tensor");
            d.mask_query_stride = (int)S;
        } else {
            TORCH_CHECK(mask.numel() == B * S, "mmf_amd: the additive key mask must hold B * S entries");
        }
    }
    d.ctx = ctx.data_ptr(); d.ldo = (int)H; d.lse = lse.data_ptr<float>();
    d.B = (int)B; d.heads = (int)heads; d.Sq = d.Sk = (int)S;
    d.scale = (float)(1.0 / std::sqrt((double)(H / heads)));
    d.drop_key = drop.key; d.drop_thr16 = drop.thr16; d.drop_scale = drop.scale; d.drop_seed = drop.seed_ptr();
    d.head_dim = (int)(H / heads);
    d.ctx_f32 = o32.defined() ? o32.data_ptr<float>() : nullptr;
    d.causal_tail = (int)tail;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// autograd nodes
// ---------------------------------------------------------------------------------------------------------------------------------
// BertLayerJit.forward (hf_layers.py:255-292) as ONE node: attention sub-layer + feed-forward sub-layer; in backward the four weight
// gradients (with the bias gradients that are column sums of their A operands) leave the dgrad chain as ONE grouped launch.
struct TransformerLayerFn : public torch::autograd::Function<TransformerLayerFn> {
    static Tensor forward(AutogradContext* ctx, const Tensor& x, const Tensor& wq, const Tensor& bq, const Tensor& wk, const Tensor& bk, const Tensor& wv,
                          const Tensor& bv, const Tensor& wo, const Tensor& bo, const Tensor& g1, const Tensor& be1, const Tensor& w1, const Tensor& b1,
                          const Tensor& w2, const Tensor& b2, const Tensor& g2, const Tensor& be2, const Tensor& wqkv16, const Tensor& bqkv,
                          const Tensor& wo16, const Tensor& w1_16, const Tensor& w2_16, const optional<Tensor>& mask_opt, int64_t heads, double eps1,
                          double eps2, Drop drop_attn, Drop drop_hid1, Drop drop_hid2, int64_t tail, bool need_bwd) {
        TORCH_CHECK(x.dim() == 3, "mmf_amd::transformer_layer: hidden states must be [B, S, H]");
        const int64_t B = x.size(0), S = x.size(1), H = x.size(2), M = B * S, I = w1_16.size(0);
        TORCH_CHECK(H % heads == 0 && (H / heads == 64 || H / heads == 128), "mmf_amd::transformer_layer: head_dim must be 64 or 128");
        Tensor x2 = as_bf16_2d(x);
        Tensor mask = mask_opt.has_value() ? *mask_opt : Tensor();
        // attention
        Tensor qkv = empty_bf16({M, 3 * H}, x2);
        Gemm(x2, wqkv16, qkv, M, 3 * H, H, H, H, 3 * H).bias(bqkv).site(MMF_SITE_QKV_FWD).run();
        Tensor ctxt = empty_bf16({M, H}, x2), lse = empty_f32({B, heads, S}, x2);
        Tensor o32 = need_bwd ? empty_f32({M, H}, x2) : Tensor();
        // the forward's dropout decisions as a bit table for the backward, where this shape's kernels take one (mmf_attn_desc.keep_bits)
        Tensor kb;
        if (need_bwd && drop_attn.on()) {
            const int64_t words = mmf_attention_keep_bits_words((int)B, (int)heads, (int)S, (int)S, (int)(H / heads));
            if (words > 0) kb = at::empty({words}, x2.options().dtype(at::kInt));
        }
        mmf_attn_desc ad;
        attn_desc(ad, qkv, H, mask, ctxt, lse, o32, B, heads, S, drop_attn, tail);
        ad.keep_bits = kb.defined() ? reinterpret_cast<uint32_t*>(kb.data_ptr<int32_t>()) : nullptr;
        MMF_RC(mmf_attention_fwd(&ad, sp()), "mmf_attention_fwd");
        Ddrln a = ddrln_fwd(ctxt, x2, wo16, bo, g1, be1, eps1, drop_hid1, MMF_SITE_ATTN_OUT_FWD);
        // feed-forward
        Tensor u = empty_bf16({M, I}, x2), hh = empty_bf16({M, I}, x2);
        Gemm(a.out, w1_16, hh, M, I, H, H, H, I).bias(b1).act(1, u).site(MMF_SITE_FFN_UP_FWD).run();
        Ddrln f = ddrln_fwd(hh, a.out, w2_16, b2, g2, be2, eps2, drop_hid2, MMF_SITE_FFN_DOWN_FWD);
        ctx->save_for_backward({x2, qkv, ctxt, lse, a.y, a.mean, a.rstd, a.out, u, hh, f.y, f.mean, f.rstd, wqkv16, wo16, w1_16, w2_16, g1.detach(),
                                g2.detach(), mask, o32, drop_attn.seed, drop_hid1.seed, drop_hid2.seed, kb});
        ctx->saved_data["dims"] = std::vector<int64_t>{B, S, H, I, heads, tail};
        ctx->saved_data["da"] = drop_pack(drop_attn); ctx->saved_data["d1"] = drop_pack(drop_hid1); ctx->saved_data["d2"] = drop_pack(drop_hid2);
        return f.out.view({B, S, H});
    }

    static variable_list backward(AutogradContext* ctx, variable_list grads) {
        auto sv = ctx->get_saved_variables();
        const Tensor &x2 = sv[0], &qkv = sv[1], &ctxt = sv[2], &lse = sv[3], &y1 = sv[4], &mean1 = sv[5], &rstd1 = sv[6], &a_out = sv[7], &u = sv[8],
                     &hh = sv[9], &y2 = sv[10], &mean2 = sv[11], &rstd2 = sv[12], &wqkv16 = sv[13], &wo16 = sv[14], &w1_16 = sv[15], &w2_16 = sv[16],
                     &g1 = sv[17], &g2 = sv[18], &mask = sv[19], &o32 = sv[20];
        auto dims = ctx->saved_data["dims"].toIntVector();
        const int64_t B = dims[0], S = dims[1], H = dims[2], I = dims[3], heads = dims[4], tail = dims[5], M = B * S;
        const Drop drop_attn = drop_unpack(ctx->saved_data["da"], sv[21]), drop_hid1 = drop_unpack(ctx->saved_data["d1"], sv[22]),
                   drop_hid2 = drop_unpack(ctx->saved_data["d2"], sv[23]);
        // feed-forward sub-layer (bias gradients of the two output projections ride on the grouped weight-gradient launch)
        LnBwd l2 = ln_bwd(grad_bf16(grads[0], H), y2, mean2, rstd2, g2, drop_hid2, false);
        Tensor du = dgrad(l2.dlin, H, w2_16, M, H, I, Tensor(), u, MMF_SITE_FFN_DOWN_DGRAD);      // (dlin2 W2) * gelu'(u)
        Tensor da = dgrad(du, I, w1_16, M, I, H, l2.dx, Tensor(), MMF_SITE_FFN_UP_DGRAD);         // du W1 + dres2
        // attention sub-layer
        LnBwd l1 = ln_bwd(da, y1, mean1, rstd1, g1, drop_hid1, false);
        Tensor dctx = dgrad(l1.dlin, H, wo16, M, H, H, Tensor(), Tensor(), MMF_SITE_ATTN_OUT_DGRAD);
        Tensor dqkv = empty_bf16({M, 3 * H}, x2), delta = empty_f32({B, heads, S}, x2);
        mmf_attn_bwd_desc bd;
        attn_desc(bd.f, qkv, H, mask, ctxt, lse, o32, B, heads, S, drop_attn, tail);
        bd.f.keep_bits = sv[24].defined() ? reinterpret_cast<uint32_t*>(sv[24].data_ptr<int32_t>()) : nullptr;
        char* dbase = reinterpret_cast<char*>(dqkv.data_ptr());
        bd.dctx = dctx.data_ptr(); bd.dq = dbase; bd.dk = dbase + 2 * H; bd.dv = dbase + 4 * H; bd.delta = delta.data_ptr<float>();
        MMF_RC(mmf_attention_bwd(&bd, sp()), "mmf_attention_bwd");
        Tensor dx = ctx->needs_input_grad(0) ? dgrad(dqkv, 3 * H, wqkv16, M, 3 * H, H, l1.dx, Tensor(), MMF_SITE_QKV_DGRAD) : Tensor();
        // the four weight gradients, one launch
        Tensor dw1 = empty_f32({I, H}, x2), db1 = empty_f32({I}, x2), dw2 = empty_f32({H, I}, x2), db2 = empty_f32({H}, x2);
        Tensor dwqkv = empty_f32({3 * H, H}, x2), dbqkv = empty_f32({3 * H}, x2), dwo = empty_f32({H, H}, x2), dbo = empty_f32({H}, x2);
        mmf_gemm_desc g[4];
        g[0] = Gemm(du, a_out, dw1, I, H, M, I, H, H).kmajor(true, true).rowsum(db1).d;
        g[1] = Gemm(l2.dlin, hh, dw2, H, I, M, H, I, I).kmajor(true, true).rowsum(db2).d;
        g[2] = Gemm(dqkv, x2, dwqkv, 3 * H, H, M, 3 * H, H, H).kmajor(true, true).rowsum(dbqkv).d;
        g[3] = Gemm(l1.dlin, ctxt, dwo, H, H, M, H, H, H).kmajor(true, true).rowsum(dbo).d;
        // inside a deferral block the launch waits for the next LayerNorm backward of this stream (WgradHeld); a layer applied twice launches at once
        if (g_ln_defer && g_wgrad_hold && g_wgrad_seen.insert(wqkv16.data_ptr()).second) {
            wgrad_release(sp(), false);
            WgradHeld h;
            std::memcpy(h.g, g, sizeof(g));
            h.keep = {du, a_out, dw1, db1, l2.dlin, hh, dw2, db2, dqkv, x2, dwqkv, dbqkv, l1.dlin, ctxt, dwo, dbo};
            h.stream = sp();
            g_wgrad_held.push_back(std::move(h));
        } else {
            wgrad_release(sp(), false);
            MMF_RC(mmf_gemm_bf16_grouped(g, 4, sp()), "mmf_gemm_bf16_grouped");
        }
        variable_list out(31);
        if (dx.defined()) out[0] = dx.view({B, S, H});
        out[1] = dwqkv.narrow(0, 0, H); out[2] = dbqkv.narrow(0, 0, H); out[3] = dwqkv.narrow(0, H, H); out[4] = dbqkv.narrow(0, H, H);
        out[5] = dwqkv.narrow(0, 2 * H, H); out[6] = dbqkv.narrow(0, 2 * H, H); out[7] = dwo; out[8] = dbo; out[9] = l1.dgamma; out[10] = l1.dbeta;
        out[11] = dw1; out[12] = db1; out[13] = dw2; out[14] = db2; out[15] = l2.dgamma; out[16] = l2.dbeta;
        return out;
    }
};

struct LayerNormFn : public torch::autograd::Function<LayerNormFn> {
    static Tensor forward(AutogradContext* ctx, const Tensor& x, const Tensor& gamma, const Tensor& beta, double eps) {
        Tensor x2 = as_bf16_2d(x);
        const int64_t M = x2.size(0), N = x2.size(1);
        Tensor out = empty_bf16({M, N}, x2), mean = empty_f32({M}, x2), rstd = empty_f32({M}, x2);
        req(gamma, at::kFloat, "weight"); req(beta, at::kFloat, "bias");
        MMF_RC(mmf_layernorm_fwd(x2.data_ptr(), PF(gamma), PF(beta), out.data_ptr(), mean.data_ptr<float>(), rstd.data_ptr<float>(), (int)M, (int)N, (float)eps,
                                 sp()), "mmf_layernorm_fwd");
        ctx->save_for_backward({x2, mean, rstd, gamma.detach()});
        ctx->saved_data["shape"] = x.sizes().vec();
        return out.view(x.sizes());
    }
    static variable_list backward(AutogradContext* ctx, variable_list grads) {
        auto sv = ctx->get_saved_variables();
        LnBwd r = ln_bwd(grad_bf16(grads[0], sv[0].size(1)), sv[0], sv[1], sv[2], sv[3], Drop(), false);
        return {r.dx.view(ctx->saved_data["shape"].toIntVector()), r.dgamma, r.dbeta, Tensor()};
    }
};

// y = x W^T + b.  x [*, K] (bf16, or raw fp32 features converted while staged), weight [N, K] fp32 master + bf16 shadow.
struct LinearFn : public torch::autograd::Function<LinearFn> {
    static Tensor forward(AutogradContext* ctx, const Tensor& x, const Tensor& weight, const optional<Tensor>& bias, const Tensor& w16, bool out_f32,
                          int64_t act /* 0 | 1 gelu | 3 tanh */) {
        Tensor x2;
        if (x.scalar_type() == at::kFloat && !x.requires_grad() && act == 0) { x2 = x.reshape({-1, x.size(-1)}).contiguous(); req(x2, at::kFloat, "x"); }
        else x2 = as_bf16_2d(x);
        const int64_t M = x2.size(0), K = x2.size(1), N = weight.size(0);
        TORCH_CHECK(weight.dim() == 2 && weight.size(1) == K, "mmf_amd::linear: weight must be [out, ", K, "], got ", weight.sizes());
        Tensor y = at::empty({M, N}, x2.options().dtype(out_f32 ? at::kFloat : at::kBFloat16)), u;
        Gemm gm(x2, w16, y, M, N, K, K, K, N);
        if (bias.has_value() && bias->defined()) gm.bias(bias->detach());
        if (act == 1) { u = empty_bf16({M, N}, x2); gm.act(1, u); }
        else if (act == 3) gm.act(3);
        gm.run();
        ctx->save_for_backward({x2, w16, act == 1 ? u : (act == 3 ? y : Tensor())});
        ctx->saved_data["meta"] = std::vector<int64_t>{M, N, K, bias.has_value() && bias->defined(), act};
        ctx->saved_data["shape"] = x.sizes().vec();
        std::vector<int64_t> os(x.sizes().begin(), x.sizes().end() - 1);
        os.push_back(N);
        return y.view(os);
    }
    static variable_list backward(AutogradContext* ctx, variable_list grads) {
        auto sv = ctx->get_saved_variables();
        const Tensor &x2 = sv[0], &w16 = sv[1], &aux = sv[2];
        auto m = ctx->saved_data["meta"].toIntVector();
        const int64_t M = m[0], N = m[1], K = m[2], act = m[4];
        const bool has_bias = m[3] != 0;
        const int ldy = pad8(N);
        Tensor g2 = grads[0].reshape({M, N}), dy;
        if (g2.scalar_type() == at::kBFloat16 && N == ldy && g2.is_contiguous()) dy = g2;
        else {
            dy = empty_bf16({M, (int64_t)ldy}, x2);
            Tensor gf = (g2.scalar_type() == at::kFloat ? g2 : g2.to(at::kFloat)).contiguous();
            MMF_RC(mmf_cast2d_f32_to_bf16(gf.data_ptr<float>(), (int)N, dy.data_ptr(), ldy, (int)M, (int)N, sp()), "mmf_cast2d_f32_to_bf16");
        }
        if (act == 1) {          // HF BertIntermediate activation: du = dh * gelu'(u)
            Tensor du = empty_bf16({M, N}, x2);
            MMF_RC(mmf_gelu_bwd_bf16(dy.data_ptr(), aux.data_ptr(), du.data_ptr(), dy.numel(), sp()), "mmf_gelu_bwd_bf16");
            dy = du;
        } else if (act == 3) {   // HF BertPooler: dpre = dy * (1 - y^2)
            Tensor dp = empty_bf16({M, N}, x2);
            MMF_RC(mmf_tanh_bwd_bf16(dy.data_ptr(), aux.data_ptr(), dp.data_ptr(), dy.numel(), sp()), "mmf_tanh_bwd_bf16");
            dy = dp;
        }
        LinBwd r = linear_bwd(dy, ldy, x2, w16, M, N, K, ctx->needs_input_grad(0), Tensor(), Tensor(), has_bias);
        return {r.dx.defined() ? r.dx.view(ctx->saved_data["shape"].toIntVector()) : Tensor(), r.dw, has_bias ? r.db : Tensor(), Tensor(), Tensor(), Tensor()};
    }
};

// out[b] = dropout(x[b, index[b]]): torch.gather + nn.Dropout of visual_bert.py:389-400
struct GatherRowsFn : public torch::autograd::Function<GatherRowsFn> {
    static Tensor forward(AutogradContext* ctx, const Tensor& x, const Tensor& index, Drop drop) {
        TORCH_CHECK(x.dim() == 3, "mmf_amd::gather_rows: x must be [B, S, H]");
        const int64_t B = x.size(0), S = x.size(1), H = x.size(2);
        Tensor x2 = as_bf16_2d(x), idx = index.contiguous();
        req(idx, at::kLong, "index");
        TORCH_CHECK(idx.numel() == B, "mmf_amd::gather_rows: one index per batch row");
        Tensor out = empty_bf16({B, H}, x2);
        MMF_RC(mmf_gather_rows(x2.data_ptr(), idx.data_ptr<int64_t>(), out.data_ptr(), (int)B, (int)S, (int)H, drop.key, drop.thr16, drop.scale, drop.seed_ptr(),
                               sp()), "mmf_gather_rows");
        ctx->save_for_backward({idx, drop.seed});
        ctx->saved_data["dims"] = std::vector<int64_t>{B, S, H};
        ctx->saved_data["drop"] = drop_pack(drop);
        return out;
    }
    static variable_list backward(AutogradContext* ctx, variable_list grads) {
        auto sv = ctx->get_saved_variables();
        auto d = ctx->saved_data["dims"].toIntVector();
        const Drop drop = drop_unpack(ctx->saved_data["drop"], sv[1]);
        Tensor g = grad_bf16(grads[0], d[2]);
        Tensor dx = at::empty({d[0] * d[1], d[2]}, g.options());
        if (d[2] % 8 == 0) {        // one pass writes the selected rows AND the zeros
            MMF_RC(mmf_scatter_rows_full(g.data_ptr(), sv[0].data_ptr<int64_t>(), dx.data_ptr(), (int)d[0], (int)d[1], (int)d[2], drop.key, drop.thr16,
                                         drop.scale, drop.seed_ptr(), sp()), "mmf_scatter_rows_full");
        } else {
            dx.zero_();
            MMF_RC(mmf_scatter_rows(g.data_ptr(), sv[0].data_ptr<int64_t>(), dx.data_ptr(), (int)d[0], (int)d[1], (int)d[2], drop.key, drop.thr16, drop.scale,
                                    drop.seed_ptr(), sp()), "mmf_scatter_rows");
        }
        return {dx.view({d[0], d[1], d[2]}), Tensor(), Tensor()};
    }
};

struct DropoutFn : public torch::autograd::Function<DropoutFn> {
    static Tensor forward(AutogradContext* ctx, const Tensor& x, Drop drop) {
        Tensor x2 = as_bf16_2d(x), y = at::empty_like(x2);
        MMF_RC(mmf_dropout_bf16(x2.data_ptr(), y.data_ptr(), x2.numel(), drop.key, drop.thr16, drop.scale, drop.seed_ptr(), sp()), "mmf_dropout_bf16");
        ctx->save_for_backward({drop.seed});
        ctx->saved_data["drop"] = drop_pack(drop);
        return y.view(x.sizes());
    }
    static variable_list backward(AutogradContext* ctx, variable_list grads) {
        const Drop drop = drop_unpack(ctx->saved_data["drop"], ctx->get_saved_variables()[0]);
        Tensor g2 = grad_bf16(grads[0], grads[0].size(-1)), d = at::empty_like(g2);
        MMF_RC(mmf_dropout_bf16(g2.data_ptr(), d.data_ptr(), g2.numel(), drop.key, drop.thr16, drop.scale, drop.seed_ptr(), sp()), "mmf_dropout_bf16");
        return {d.view(grads[0].sizes()), Tensor()};
    }
};

// nlvr2: [2B, H] -> [B, 2H] = cat(x[:B], x[B:], dim=1) (visual_bert.py:369-374) as two strided row copies
struct PairHalvesFn : public torch::autograd::Function<PairHalvesFn> {
    static Tensor forward(AutogradContext* ctx, const Tensor& x) {
        Tensor x2 = as_bf16_2d(x);
        const int64_t B2 = x2.size(0), H = x2.size(1), B = B2 / 2;
        TORCH_CHECK(B2 % 2 == 0 && H % 8 == 0, "mmf_amd::pair_halves: [2B, H] with H % 8 == 0");
        Tensor out = empty_bf16({B, 2 * H}, x2);
        char* o = reinterpret_cast<char*>(out.data_ptr());
        const char* s = reinterpret_cast<const char*>(x2.data_ptr());
        MMF_RC(mmf_copy_rows_bf16(s, 1, o, 2, (int)B, 1, (int)H, sp()), "mmf_copy_rows_bf16");
        MMF_RC(mmf_copy_rows_bf16(s + B * H * 2, 1, o + H * 2, 2, (int)B, 1, (int)H, sp()), "mmf_copy_rows_bf16");
        ctx->saved_data["dims"] = std::vector<int64_t>{B, H};
        return out;
    }
    static variable_list backward(AutogradContext* ctx, variable_list grads) {
        auto d = ctx->saved_data["dims"].toIntVector();
        const int64_t B = d[0], H = d[1];
        Tensor g2 = grad_bf16(grads[0], 2 * H), dx = empty_bf16({2 * B, H}, g2);
        const char* s = reinterpret_cast<const char*>(g2.data_ptr());
        char* o = reinterpret_cast<char*>(dx.data_ptr());
        MMF_RC(mmf_copy_rows_bf16(s, 2, o, 1, (int)B, 1, (int)H, sp()), "mmf_copy_rows_bf16");
        MMF_RC(mmf_copy_rows_bf16(s + H * 2, 2, o + B * H * 2, 1, (int)B, 1, (int)H, sp()), "mmf_copy_rows_bf16");
        return {dx};
    }
};

// BertVisioLinguisticEmbeddings.forward (embeddings.py:423-459, image_text_alignment = None)
static const bool g_feats_cast = [] { const char* e = std::getenv("MMF_AMD_FEATS_CAST"); return !(e && e[0] == '0'); }();
struct VisioLinguisticEmbeddingsFn : public torch::autograd::Function<VisioLinguisticEmbeddingsFn> {
    static Tensor forward(AutogradContext* ctx, const Tensor& input_ids, const Tensor& token_type_ids, const optional<Tensor>& feats_o,
                          const optional<Tensor>& vtype_o, const Tensor& word, const Tensor& pos, const Tensor& typ, const Tensor& ln_w, const Tensor& ln_b,
                          const Tensor& typ_vis, const Tensor& pos_vis, const Tensor& proj_w, const Tensor& proj_b, const Tensor& proj_w16, double eps,
                          Drop drop, int64_t pad_idx, const optional<Tensor>& align_o) {
        TORCH_CHECK(input_ids.dim() == 2, "mmf_amd::visio_linguistic_embeddings: input_ids must be [B, T]");
        const int64_t B = input_ids.size(0), T = input_ids.size(1), H = word.size(1);
        const bool vis = feats_o.has_value() && feats_o->defined() && vtype_o.has_value() && vtype_o->defined();
        const int64_t R = vis ? feats_o->size(1) : 0, S = T + R;
        Tensor y = empty_bf16({B * S, H}, word);
        Tensor ids = input_ids.contiguous(), seg = token_type_ids.contiguous();
        req(ids, at::kLong, "input_ids"); req(seg, at::kLong, "token_type_ids");
        for (const Tensor* t : {&word, &pos, &typ}) req(*t, at::kFloat, "embedding table");
        TORCH_CHECK(seg.numel() == B * T, "mmf_amd::visio_linguistic_embeddings: token_type_ids must match input_ids");
        MMF_RC(mmf_embed_text_fwd(ids.data_ptr<int64_t>(), seg.data_ptr<int64_t>(), PF(word), PF(pos), PF(typ), y.data_ptr(), (int)B, (int)T, (int)S, (int)H, 0, 0,
                                  (int)word.size(0), (int)pos.size(0), (int)typ.size(0), sp()), "mmf_embed_text_fwd");
        Tensor f2, vt, al;
        if (R) {
            const Tensor& feats = *feats_o;
            const int64_t D = feats.size(2);
            f2 = feats.reshape({B * R, D});
            if (f2.scalar_type() != at::kFloat && f2.scalar_type() != at::kBFloat16) f2 = f2.to(at::kFloat);
            f2 = f2.contiguous();
            if (f2.scalar_type() == at::kFloat && g_feats_cast && D % 8 == 0 && !feats.requires_grad()) {   // fp32 features cast to bf16 ONCE
                Tensor f16 = empty_bf16({B * R, D}, f2);
                MMF_RC(mmf_cast_f32_to_bf16(f2.data_ptr<float>(), f16.data_ptr(), f2.numel(), sp()), "mmf_cast_f32_to_bf16");
                f2 = f16;
            }
            vt = vtype_o->reshape({B * R}).contiguous();
            req(vt, at::kLong, "visual_embeddings_type");
            if (align_o.has_value() && align_o->defined()) {
                // image_text_alignment (embeddings.py:373-397): the mean text-position row of the words aligned with each region (+ its visual
                // token-type row) as one fp32 addend row per region, gathered by the projection GEMM's epilogue
                al = align_o->reshape({B * R, -1}).contiguous();
                if (al.scalar_type() != at::kLong) al = al.to(at::kLong);
                req(al, at::kLong, "image_text_alignment");
                req(typ_vis, at::kFloat, "token_type_embeddings_visual");
                Tensor addend = empty_f32({B * R, H}, y);
                MMF_RC(mmf_align_pos_fwd(al.data_ptr<int64_t>(), PF(pos), PF(typ_vis), vt.data_ptr<int64_t>(), addend.data_ptr<float>(), (int)(B * R), (int)al.size(1),
                                         (int)H, (int)pos.size(0), (int)typ_vis.size(0), sp()), "mmf_align_pos_fwd");
                Tensor rows = at::arange(B * R, vt.options());
                Gemm(f2, proj_w16, y, B * R, H, D, D, D, H).bias(proj_b.detach()).coladd(pos_vis.detach()[0]).rowtab(addend, rows, (int)H).grp((int)R, (int)T, (int)T).run();
            } else {
                Gemm(f2, proj_w16, y, B * R, H, D, D, D, H).bias(proj_b.detach()).coladd(pos_vis.detach()[0]).rowtab(typ_vis.detach(), vt, (int)H).grp((int)R, (int)T, (int)T).run();
            }
        }
        Tensor out = empty_bf16({B * S, H}, y), mean = empty_f32({B * S}, y), rstd = empty_f32({B * S}, y);
        req(ln_w, at::kFloat, "LayerNorm.weight"); req(ln_b, at::kFloat, "LayerNorm.bias");
        if (drop.on() && mmf_layernorm_dropout_fusable((int)H)) {      // LayerNorm + nn.Dropout (embeddings.py:343-345) as one launch, same bits as the two below
            MMF_RC(mmf_layernorm_dropout_fwd(y.data_ptr(), PF(ln_w), PF(ln_b), out.data_ptr(), mean.data_ptr<float>(), rstd.data_ptr<float>(), (int)(B * S), (int)H,
                                             (float)eps, drop.key, drop.thr16, drop.scale, drop.seed_ptr(), sp()), "mmf_layernorm_dropout_fwd");
        } else {
            MMF_RC(mmf_layernorm_fwd(y.data_ptr(), PF(ln_w), PF(ln_b), out.data_ptr(), mean.data_ptr<float>(), rstd.data_ptr<float>(), (int)(B * S), (int)H,
                                     (float)eps, sp()), "mmf_layernorm_fwd");
            if (drop.on()) {
                Tensor o2 = at::empty_like(out);
                MMF_RC(mmf_dropout_bf16(out.data_ptr(), o2.data_ptr(), out.numel(), drop.key, drop.thr16, drop.scale, drop.seed_ptr(), sp()), "mmf_dropout_bf16");
                out = o2;
            }
        }
        ctx->save_for_backward({ids, seg, f2, vt, y, mean, rstd, ln_w.detach(), proj_w16, drop.seed, al});
        ctx->saved_data["dims"] = std::vector<int64_t>{B, T, R, S, H, word.size(0), pos.size(0), typ.size(0), typ_vis.size(0), pos_vis.size(0), pad_idx};
        ctx->saved_data["drop"] = drop_pack(drop);
        return out.view({B, S, H});
    }
    static void scatter(const void* x, int64_t ld, int64_t nb, int64_t rpb, int64_t bstride, const Tensor& idx, int64_t idx_ld, int per_pos, const Tensor& out,
                        int64_t H, int few, int64_t skip) {
        Tensor ws;
        if (few) ws = empty_f32({(int64_t)mmf_rows_scatter_add_ws_floats((int)H)}, out);
        MMF_RC(mmf_rows_scatter_add(x, (int)ld, (int)nb, (int)rpb, (int)bstride, idx.defined() ? idx.data_ptr<int64_t>() : nullptr, (int)idx_ld, per_pos, 0,
                                    out.data_ptr<float>(), (int)H, few, (int)out.size(0), ws.defined() ? ws.data_ptr<float>() : nullptr, (int)skip, sp()),
               "mmf_rows_scatter_add");
    }
    static variable_list backward(AutogradContext* ctx, variable_list grads) {
        auto sv = ctx->get_saved_variables();
        const Tensor &ids = sv[0], &seg = sv[1], &f2 = sv[2], &vt = sv[3], &y = sv[4], &mean = sv[5], &rstd = sv[6], &ln_w = sv[7];
        auto d = ctx->saved_data["dims"].toIntVector();
        const int64_t B = d[0], T = d[1], R = d[2], S = d[3], H = d[4], V = d[5], Pn = d[6], NT = d[7], NTV = d[8], PV = d[9], pad = d[10];
        const Drop drop = drop_unpack(ctx->saved_data["drop"], sv[9]);
        Tensor dy = grad_bf16(grads[0], H);
        LnBwd l;
        if (drop.on() && mmf_layernorm_dropout_fusable((int)H)) {      // dropout backward applied while the LayerNorm backward loads dy (same bits as the two launches)
            l = ln_bwd_din(dy, y, mean, rstd, ln_w, drop);
        } else {
            if (drop.on()) {
                Tensor d2 = at::empty_like(dy);
                MMF_RC(mmf_dropout_bf16(dy.data_ptr(), d2.data_ptr(), dy.numel(), drop.key, drop.thr16, drop.scale, drop.seed_ptr(), sp()), "mmf_dropout_bf16");
                dy = d2;
            }
            l = ln_bwd(dy, y, mean, rstd, ln_w, Drop(), false);
        }
        const Tensor& dpre = l.dx;
        auto f32o = dpre.options().dtype(at::kFloat);
        // ONE zero fill for the five table gradients (row blocks of one buffer: each gradient is a contiguous [rows, H] view of it)
        Tensor tabs = at::zeros({V + Pn + NT + (R ? NTV + PV : 0), H}, f32o);
        Tensor dword = tabs.narrow(0, 0, V), dpos = tabs.narrow(0, V, Pn), dtyp = tabs.narrow(0, V + Pn, NT);
        scatter(dpre.data_ptr(), H, B, T, S, ids, T, 0, dword, H, 0, pad);      // padding_idx rows get no gradient
        Tensor dtyp_vis, dpos_vis, dproj_w, dproj_b;
        if (R) {
            dtyp_vis = tabs.narrow(0, V + Pn + NT, NTV);
            dpos_vis = tabs.narrow(0, V + Pn + NT + NTV, PV);
        }
        {   // text positions, text token types, visual token types, the visual position row: one pass over dpre (two launches instead of seven)
            Tensor ws = empty_f32({(int64_t)mmf_embed_tables_bwd_ws_floats((int)S, (int)H)}, tabs);
            MMF_RC(mmf_embed_tables_bwd(dpre.data_ptr(), (int)H, (int)B, (int)T, (int)R, seg.data_ptr<int64_t>(), R ? vt.data_ptr<int64_t>() : nullptr, 0,
                                        dpos.data_ptr<float>(), (int)Pn, dtyp.data_ptr<float>(), (int)NT, R ? dtyp_vis.data_ptr<float>() : nullptr, (int)NTV,
                                        R ? dpos_vis.data_ptr<float>() : nullptr, (int)H, ws.data_ptr<float>(), sp()), "mmf_embed_tables_bwd");
        }
        if (R) {
            const char* vis = reinterpret_cast<const char*>(dpre.data_ptr()) + T * H * 2;   // row (b, r) of the visual block lives at dpre[b*S + T + r]
            if (sv[10].defined())       // the aligned words' TEXT position rows collect the regions' gradients / count
                MMF_RC(mmf_align_pos_bwd(vis, (int)H, (int)B, (int)R, (int)S, sv[10].data_ptr<int64_t>(), dpos.data_ptr<float>(), (int)sv[10].size(1), (int)H, (int)Pn,
                                         sp()), "mmf_align_pos_bwd");
            Tensor dvis = empty_bf16({B * R, H}, dpre);
            MMF_RC(mmf_copy_rows_bf16(vis, (int)S, dvis.data_ptr(), (int)R, (int)B, (int)R, (int)H, sp()), "mmf_copy_rows_bf16");
            const int64_t D = f2.size(1);
            dproj_w = at::empty({H, D}, f32o);
            if (f2.scalar_type() == at::kBFloat16) {      // the bias gradient = row sums of the GEMM's A operand: delivered by the same launch
                dproj_b = at::empty({H}, f32o);
                Gemm(dvis, f2, dproj_w, H, D, B * R, H, D, D).kmajor(true, true).rowsum(dproj_b).run();
            } else {                                     // (fp32 features staged by the GEMM: its row-sum form takes bf16 operands)
                Gemm(dvis, f2, dproj_w, H, D, B * R, H, D, D).kmajor(true, true).run();
                dproj_b = colsum(dvis, H, B * R, H);
            }
        }
        return {Tensor(), Tensor(), Tensor(), Tensor(), dword, dpos, dtyp, l.dgamma, l.dbeta, dtyp_vis, dpos_vis, dproj_w, dproj_b, Tensor(), Tensor(), Tensor(), Tensor(),
                Tensor()};
    }
};

// mean(BCEWithLogits(scores, targets)) * num_labels   (mmf/modules/losses.py:246-251)
struct LogitBCEFn : public torch::autograd::Function<LogitBCEFn> {
    static Tensor forward(AutogradContext* ctx, const Tensor& scores, const Tensor& targets) {
        TORCH_CHECK(scores.dim() == 2 && targets.sizes() == scores.sizes(), "mmf_amd::logit_bce: scores and targets must both be [B, N]");
        const int64_t B = scores.size(0), N = scores.size(1);
        Tensor s = scores.to(at::kFloat).contiguous(), t = targets.to(at::kFloat).contiguous();
        req(s, at::kFloat, "scores"); req(t, at::kFloat, "targets");
        Tensor loss = empty_f32({1}, s), ws = empty_f32({(int64_t)mmf_bce_logits_ws_floats()}, s);
        MMF_RC(mmf_bce_logits_fwd(PF(s), PF(t), loss.data_ptr<float>(), ws.data_ptr<float>(), (int)B, (int)N, sp()), "mmf_bce_logits_fwd");
        ctx->save_for_backward({s, t});
        return loss[0];
    }
    static variable_list backward(AutogradContext* ctx, variable_list grads) {
        auto sv = ctx->get_saved_variables();
        const int64_t B = sv[0].size(0), N = sv[0].size(1);
        const int ldd = pad8(N);
        Tensor g = grads[0].to(at::kFloat).reshape({1}).contiguous();
        Tensor d = empty_f32({B, N}, sv[0]);        // fp32 directly (one launch; the classifier's backward rounds it to its bf16 GEMM operand)
        (void)ldd;
        MMF_RC(mmf_bce_logits_f32_bwd(PF(sv[0]), PF(sv[1]), PF(g), d.data_ptr<float>(), (int)B, (int)N, sp()), "mmf_bce_logits_f32_bwd");
        return {d, Tensor()};
    }
};

// The decoder + loss of the pretraining heads as ONE node each, so that backward never materialises an fp32 [rows, classes] gradient:
// the loss kernel keeps each row's log-sum-exp and its backward writes the bf16 operand of the decoder's dgrad / wgrad GEMMs directly.
//   kind 0: masked LM — prediction scores = h W^T + b, nn.CrossEntropyLoss(ignore_index) (mmf/models/visual_bert.py:267-277; HF BertLMPredictionHead)
//   kind 1: masked region classification — KLDivLoss(log_softmax(scores), target) over the rows with label 1 / their number
//           (mmf/models/vilbert.py:846-858, 1150-1157, `visual_target: 0`)
// Returns (loss, scores [*, classes] fp32); the scores are non-differentiable (only the loss carries gradient).
struct PretrainHeadFn : public torch::autograd::Function<PretrainHeadFn> {
    static variable_list forward(AutogradContext* ctx, const Tensor& x, const Tensor& weight, const Tensor& bias, const Tensor& w16, const Tensor& labels,
                                 const optional<Tensor>& target, int64_t ignore_index, int64_t kind) {
        Tensor x2 = as_bf16_2d(x);
        const int64_t M = x2.size(0), K = x2.size(1), N = weight.size(0);
        TORCH_CHECK(weight.dim() == 2 && weight.size(1) == K, "mmf_amd pretraining head: decoder weight must be [classes, ", K, "]");
        Tensor logits = empty_f32({M, N}, x2);
        Gemm(x2, w16, logits, M, N, K, K, K, N).bias(bias.detach()).run();
        Tensor lab = labels.reshape({M}).contiguous();
        if (lab.scalar_type() != at::kLong) lab = lab.to(at::kLong);
        req(lab, at::kLong, "labels");
        Tensor lse = empty_f32({M}, x2), rowloss = empty_f32({M}, x2), loss = empty_f32({1}, x2), count = empty_f32({1}, x2), tgt, tsum;
        if (kind == 0) {
            MMF_RC(mmf_vocab_cross_entropy_fwd(PF(logits), (int)N, lab.data_ptr<int64_t>(), lse.data_ptr<float>(), rowloss.data_ptr<float>(), loss.data_ptr<float>(),
                                               count.data_ptr<float>(), (int)M, (int)N, (int)ignore_index, sp()), "mmf_vocab_cross_entropy_fwd");
        } else {
            TORCH_CHECK(target.has_value() && target->numel() == M * N, "mmf_amd::masked_region_head: target must be [rows, classes]");
            tgt = target->reshape({M, N}).to(at::kFloat).contiguous();
            req(tgt, at::kFloat, "target");
            tsum = empty_f32({M}, x2);
            MMF_RC(mmf_soft_target_kl_fwd(PF(logits), (int)N, PF(tgt), (int)N, lab.data_ptr<int64_t>(), lse.data_ptr<float>(), tsum.data_ptr<float>(),
                                          rowloss.data_ptr<float>(), loss.data_ptr<float>(), count.data_ptr<float>(), (int)M, (int)N, sp()), "mmf_soft_target_kl_fwd");
        }
        ctx->save_for_backward({x2, w16, logits, lab, lse, count, tgt, tsum});
        ctx->saved_data["meta"] = std::vector<int64_t>{M, N, K, ignore_index, kind};
        ctx->saved_data["shape"] = x.sizes().vec();
        std::vector<int64_t> os(x.sizes().begin(), x.sizes().end() - 1);
        os.push_back(N);
        Tensor out = logits.view(os);
        ctx->mark_non_differentiable({out});
        return {loss[0], out};
    }
    static variable_list backward(AutogradContext* ctx, variable_list grads) {
        auto sv = ctx->get_saved_variables();
        const Tensor &x2 = sv[0], &w16 = sv[1], &logits = sv[2], &lab = sv[3], &lse = sv[4], &count = sv[5], &tgt = sv[6], &tsum = sv[7];
        auto m = ctx->saved_data["meta"].toIntVector();
        const int64_t M = m[0], N = m[1], K = m[2], ignore_index = m[3], kind = m[4];
        const int ldd = pad8(N);
        Tensor d = empty_bf16({M, (int64_t)ldd}, x2), g = grads[0].to(at::kFloat).reshape({1}).contiguous();
        if (kind == 0) {
            MMF_RC(mmf_vocab_cross_entropy_bwd(PF(logits), (int)N, lab.data_ptr<int64_t>(), PF(lse), PF(count), PF(g), d.data_ptr(), ldd, (int)M, (int)N,
                                               (int)ignore_index, sp()), "mmf_vocab_cross_entropy_bwd");
        } else {
            MMF_RC(mmf_soft_target_kl_bwd(PF(logits), (int)N, PF(tgt), (int)N, lab.data_ptr<int64_t>(), PF(lse), PF(tsum), PF(count), PF(g), d.data_ptr(), ldd, (int)M,
                                          (int)N, sp()), "mmf_soft_target_kl_bwd");
        }
        LinBwd r = linear_bwd(d, ldd, x2, w16, M, N, K, ctx->needs_input_grad(0), Tensor(), Tensor(), true);
        return {r.dx.defined() ? r.dx.view(ctx->saved_data["shape"].toIntVector()) : Tensor(), r.dw, r.db, Tensor(), Tensor(), Tensor(), Tensor(), Tensor()};
    }
};

// ---------------------------------------------------------------------------------------------------------------------------------
// operators
// ---------------------------------------------------------------------------------------------------------------------------------
// Modes in which an operator forwards to its Python-implemented `_py_` twin: bit 0 = the fp32-accurate forward path is on, bit 1 = an
// opt-in experiment hook (weight gradients on a side stream, optimizer in backward) is active.
int64_t g_py_mode = 0;
template <typename Sig, typename... A>
auto call_py(const char* name, A&&... a) {
    static std::unordered_map<std::string, c10::OperatorHandle> handles;
    auto it = handles.find(name);
    if (it == handles.end())
        it = handles.emplace(name, c10::Dispatcher::singleton().findSchemaOrThrow((std::string("mmf_amd::_py_") + name).c_str(), "")).first;
    return it->second.typed<Sig>().call(std::forward<A>(a)...);
}

// VisualBERT.forward's input massaging as one launch: (image_mask, attention_mask, visual_embeddings_type, additive mask, pooling index)
std::tuple<Tensor, Tensor, Tensor, Tensor, Tensor> op_visual_masks(const Tensor& input_mask, const optional<Tensor>& image_dim, int64_t R) {
    TORCH_CHECK(input_mask.is_cuda(), "mmf_amd::visual_masks: the mask must live in HBM; there is no CPU path");
    TORCH_CHECK(input_mask.dim() == 2 && R > 0, "mmf_amd::visual_masks: input_mask must be [B, T] and R > 0");
    Tensor im = input_mask.to(at::kLong).contiguous();
    const int64_t B = im.size(0), T = im.size(1);
    Tensor dim;
    if (image_dim.has_value() && image_dim->defined()) {
        dim = image_dim->to(at::kLong).reshape({-1}).contiguous();
        TORCH_CHECK(dim.numel() == B && dim.is_cuda(), "mmf_amd::visual_masks: image_dim must hold one region count per sample, in HBM");
    }
    auto lo = im.options();
    Tensor image_mask = at::empty({B, R}, lo), attention_mask = at::empty({B, T + R}, lo), vtype = at::empty({B, R}, lo), pool = at::empty({B}, lo);
    Tensor mask_add = at::empty({B, T + R}, lo.dtype(at::kFloat));
    MMF_RC(mmf_visual_masks(im.data_ptr<int64_t>(), dim.defined() ? dim.data_ptr<int64_t>() : nullptr, (int)B, (int)T, (int)R, image_mask.data_ptr<int64_t>(),
                            attention_mask.data_ptr<int64_t>(), vtype.data_ptr<int64_t>(), mask_add.data_ptr<float>(), pool.data_ptr<int64_t>(), sp()),
           "mmf_visual_masks");
    return {image_mask, attention_mask, vtype, mask_add, pool};
}

Tensor op_additive_mask(const Tensor& mask) {
    TORCH_CHECK(mask.is_cuda(), "mmf_amd::additive_mask: the mask must live in HBM; there is no CPU path");
    Tensor am = mask.contiguous();
    if (am.scalar_type() != at::kLong) am = am.to(at::kLong);
    Tensor out = at::empty(am.sizes(), am.options().dtype(at::kFloat));
    MMF_RC(mmf_make_additive_mask(am.data_ptr<int64_t>(), out.data_ptr<float>(), am.numel(), sp()), "mmf_make_additive_mask");
    return out;
}

using VleSig = Tensor(const Tensor&, const Tensor&, const optional<Tensor>&, const optional<Tensor>&, const Tensor&, const Tensor&, const Tensor&, const Tensor&,
                      const Tensor&, const Tensor&, const Tensor&, const Tensor&, const Tensor&, double, double, bool, int64_t, const optional<Tensor>&);
Tensor op_vle(const Tensor& input_ids, const Tensor& token_type_ids, const optional<Tensor>& vis, const optional<Tensor>& vis_type, const Tensor& word,
              const Tensor& pos, const Tensor& typ, const Tensor& ln_w, const Tensor& ln_b, const Tensor& typ_vis, const Tensor& pos_vis, const Tensor& proj_w,
              const Tensor& proj_b, double eps, double p, bool training, int64_t pad_idx, const optional<Tensor>& align) {
    if (g_py_mode & 1) return call_py<VleSig>("visio_linguistic_embeddings", input_ids, token_type_ids, vis, vis_type, word, pos, typ, ln_w, ln_b, typ_vis, pos_vis,
                                              proj_w, proj_b, eps, p, training, pad_idx, align);
    const bool have = vis.has_value() && vis->defined() && vis_type.has_value() && vis_type->defined();
    Tensor w16 = have ? g_shadows.get({proj_w}, false) : Tensor();
    return VisioLinguisticEmbeddingsFn::apply(input_ids, token_type_ids, have ? vis : optional<Tensor>(), have ? vis_type : optional<Tensor>(), word, pos, typ,
                                              ln_w, ln_b, typ_vis, pos_vis, proj_w, proj_b, w16, eps, make_drop(p, training), pad_idx,
                                              have ? align : optional<Tensor>());
}

using LayerSig = Tensor(const Tensor&, const Tensor&, const Tensor&, const Tensor&, const Tensor&, const Tensor&, const Tensor&, const Tensor&, const Tensor&,
                        const Tensor&, const Tensor&, const Tensor&, const Tensor&, const Tensor&, const Tensor&, const Tensor&, const Tensor&,
                        const optional<Tensor>&, int64_t, double, double, double, double, double, bool, int64_t);
Tensor op_transformer_layer(const Tensor& x, const Tensor& wq, const Tensor& bq, const Tensor& wk, const Tensor& bk, const Tensor& wv, const Tensor& bv,
                            const Tensor& wo, const Tensor& bo, const Tensor& ln1_w, const Tensor& ln1_b, const Tensor& w1, const Tensor& b1, const Tensor& w2,
                            const Tensor& b2, const Tensor& ln2_w, const Tensor& ln2_b, const optional<Tensor>& mask_add, int64_t heads, double eps1, double eps2,
                            double p_attn, double p_hid1, double p_hid2, bool training, int64_t causal_tail) {
    if (g_py_mode) return call_py<LayerSig>("transformer_layer", x, wq, bq, wk, bk, wv, bv, wo, bo, ln1_w, ln1_b, w1, b1, w2, b2, ln2_w, ln2_b, mask_add, heads, eps1,
                                            eps2, p_attn, p_hid1, p_hid2, training, causal_tail);
    Tensor wqkv16 = g_shadows.get({wq, wk, wv}, false), bqkv = g_shadows.get({bq, bk, bv}, true);
    Tensor wo16 = g_shadows.get({wo}, false), w1_16 = g_shadows.get({w1}, false), w2_16 = g_shadows.get({w2}, false);
    const Drop da = make_drop(p_attn, training), d1 = make_drop(p_hid1, training), d2 = make_drop(p_hid2, training);
    bool any_grad = false;
    for (const Tensor* t : {&x, &wq, &bq, &wk, &bk, &wv, &bv, &wo, &bo, &ln1_w, &ln1_b, &w1, &b1, &w2, &b2, &ln2_w, &ln2_b}) any_grad = any_grad || t->requires_grad();
    const bool need_bwd = at::GradMode::is_enabled() && any_grad;
    return TransformerLayerFn::apply(x, wq, bq, wk, bk, wv, bv, wo, bo, ln1_w, ln1_b, w1, b1, w2, b2, ln2_w, ln2_b, wqkv16, bqkv, wo16, w1_16, w2_16, mask_add,
                                     heads, eps1, eps2, da, d1, d2, causal_tail, need_bwd);
}

using LinSig = Tensor(const Tensor&, const Tensor&, const optional<Tensor>&, bool);
Tensor op_linear(const Tensor& x, const Tensor& weight, const optional<Tensor>& bias, bool out_f32) {
    if (g_py_mode & 1) return call_py<LinSig>("linear", x, weight, bias, out_f32);
    return LinearFn::apply(x, weight, bias, g_shadows.get({weight}, false), out_f32, 0);
}
using Lin3Sig = Tensor(const Tensor&, const Tensor&, const Tensor&);
Tensor op_dense_gelu(const Tensor& x, const Tensor& weight, const Tensor& bias) {
    if (g_py_mode & 1) return call_py<Lin3Sig>("dense_gelu", x, weight, bias);
    return LinearFn::apply(x, weight, optional<Tensor>(bias), g_shadows.get({weight}, false), false, 1);
}
Tensor op_linear_tanh(const Tensor& x, const Tensor& weight, const Tensor& bias) {
    if (g_py_mode & 1) return call_py<Lin3Sig>("linear_tanh", x, weight, bias);
    return LinearFn::apply(x, weight, optional<Tensor>(bias), g_shadows.get({weight}, false), false, 3);
}
using LnSig = Tensor(const Tensor&, const Tensor&, const Tensor&, double);
Tensor op_layer_norm(const Tensor& x, const Tensor& weight, const Tensor& bias, double eps) {
    if (g_py_mode & 1) return call_py<LnSig>("layer_norm", x, weight, bias, eps);
    return LayerNormFn::apply(x, weight, bias, eps);
}
using GrSig = Tensor(const Tensor&, const Tensor&, double, bool);
Tensor op_gather_rows(const Tensor& x, const Tensor& index, double p, bool training) {
    if (g_py_mode & 1) return call_py<GrSig>("gather_rows", x, index, p, training);
    return GatherRowsFn::apply(x, index, make_drop(p, training));
}
using DrSig = Tensor(const Tensor&, double, bool);
Tensor op_dropout(const Tensor& x, double p, bool training) {
    if (g_py_mode & 1) return call_py<DrSig>("dropout", x, p, training);
    const Drop d = make_drop(p, training);
    if (!d.on()) return x;
    return DropoutFn::apply(x, d);
}
using PhSig = Tensor(const Tensor&);
Tensor op_pair_halves(const Tensor& x) {
    if (g_py_mode & 1) return call_py<PhSig>("pair_halves", x);
    return PairHalvesFn::apply(x);
}
Tensor op_logit_bce(const Tensor& scores, const Tensor& targets) { return LogitBCEFn::apply(scores, targets); }
using MlmSig = std::tuple<Tensor, Tensor>(const Tensor&, const Tensor&, const Tensor&, const Tensor&, int64_t);
std::tuple<Tensor, Tensor> op_masked_lm_head(const Tensor& x, const Tensor& weight, const Tensor& bias, const Tensor& labels, int64_t ignore_index) {
    if (g_py_mode & 1) return call_py<MlmSig>("masked_lm_head", x, weight, bias, labels, ignore_index);
    auto r = PretrainHeadFn::apply(x, weight, bias, g_shadows.get({weight}, false), labels, optional<Tensor>(), ignore_index, 0);
    return {r[0], r[1]};
}
using MrhSig = std::tuple<Tensor, Tensor>(const Tensor&, const Tensor&, const Tensor&, const Tensor&, const Tensor&);
std::tuple<Tensor, Tensor> op_masked_region_head(const Tensor& x, const Tensor& weight, const Tensor& bias, const Tensor& target, const Tensor& row_label) {
    if (g_py_mode & 1) return call_py<MrhSig>("masked_region_head", x, weight, bias, target, row_label);
    auto r = PretrainHeadFn::apply(x, weight, bias, g_shadows.get({weight}, false), row_label, optional<Tensor>(target), -1, 1);
    return {r[0], r[1]};
}

// ---- service operators: the Python package drives the state above through these -------------------------------------------------
Tensor svc_shadow_get(at::TensorList params, bool as_f32) { return g_shadows.get(params.vec(), as_f32); }
std::vector<Tensor> svc_shadow_slot(const Tensor& p) { Tensor t = g_shadows.slot(p); return t.defined() ? std::vector<Tensor>{t} : std::vector<Tensor>{}; }
std::vector<Tensor> svc_shadow_transposed(const Tensor& w16) { Tensor t = g_shadows.transposed(w16); return t.defined() ? std::vector<Tensor>{t} : std::vector<Tensor>{}; }
void svc_shadow_clear() { g_shadows.clear(); }
bool svc_shadow_set_twins(bool on) { const bool old = g_shadows.twins_on; g_shadows.twins_on = on; return old; }
void svc_shadow_refresh_transposed(at::TensorList only, bool has_only, at::TensorList skip, bool has_skip) {
    g_shadows.refresh_transposed(only.vec(), has_only, skip.vec(), has_skip);
}
void svc_drop_graph_mode(const optional<Tensor>& seed) { g_keys.graph_seed = seed.has_value() ? *seed : Tensor(); g_keys.counter = 0; }
std::tuple<int64_t, std::vector<Tensor>> svc_drop_next() {
    Tensor seed;
    const uint32_t k = g_keys.next(seed);
    return {(int64_t)k, seed.defined() ? std::vector<Tensor>{seed} : std::vector<Tensor>{}};
}
// The fused AdamW launches of one optimizer step (mmf/modules/optimizers.py:8-17, transformers.AdamW): the descriptor of every launch is filled
// here instead of field by field through ctypes (0.4 ms of host time per step for the 200 tensors of VisualBERT, plus one operator call per
// parameter for its shadow slot), so MMF's own eager loop (training_loop.py:199-231) stays ahead of the GPU.  Mirrors (bf16 weight shadow or
// packed fp32 bias slice) are looked up per parameter, with the version check of Shadows::slot.
void svc_adamw_step(at::TensorList p, at::TensorList g, at::TensorList m, at::TensorList v, at::ArrayRef<double> lr, at::ArrayRef<double> wd,
                    double beta1, double beta2, double eps, int64_t step, bool correct_bias, int64_t mode, double grad_scale,
                    const c10::optional<Tensor>& norm_sq, double max_norm, const c10::optional<Tensor>& dev_state) {
    const size_t n = p.size();
    TORCH_CHECK(g.size() == n && m.size() == n && v.size() == n && lr.size() == n && wd.size() == n, "mmf_amd: adamw_step takes lists of one length");
    for (size_t i0 = 0; i0 < n; i0 += MMF_MT_MAX) {
        mmf_adamw_multi_desc d;
        std::memset(&d, 0, sizeof(d));
        d.n = (int)std::min<size_t>(MMF_MT_MAX, n - i0);
        for (int i = 0; i < d.n; ++i) {
            const Tensor &pp = p[i0 + i], &gg = g[i0 + i];
            TORCH_CHECK(pp.scalar_type() == at::kFloat && pp.is_contiguous() && m[i0 + i].is_contiguous() && v[i0 + i].is_contiguous(),
                        "mmf_amd: adamw_step wants contiguous fp32 parameters and moments");
            TORCH_CHECK((gg.scalar_type() == at::kFloat || gg.scalar_type() == at::kBFloat16) && gg.is_contiguous() && gg.numel() == pp.numel(),
                        "mmf_amd: adamw_step: gradients must be contiguous fp32 or bf16 with the parameter's element count");
            if (gg.scalar_type() == at::kBFloat16) d.g_bf16_mask |= (uint64_t)1 << i;
            d.p[i] = pp.data_ptr(); d.g[i] = gg.data_ptr(); d.m[i] = m[i0 + i].data_ptr(); d.v[i] = v[i0 + i].data_ptr();
            const Tensor mirror = g_shadows.slot(pp);
            if (mirror.defined()) (mirror.scalar_type() == at::kBFloat16 ? d.p16[i] : d.p32[i]) = mirror.data_ptr();
            d.numel[i] = pp.numel(); d.lr[i] = (float)lr[i0 + i]; d.wd[i] = (float)wd[i0 + i];
        }
        d.beta1 = (float)beta1; d.beta2 = (float)beta2; d.eps = (float)eps;
        d.step = (int)step; d.correct_bias = correct_bias ? 1 : 0; d.mode = (int)mode;
        d.grad_scale = (float)grad_scale;
        d.norm_sq = (norm_sq.has_value() && norm_sq->defined()) ? norm_sq->data_ptr<float>() : nullptr;
        d.max_norm = (float)max_norm;
        d.dev_state = (dev_state.has_value() && dev_state->defined()) ? dev_state->data_ptr<float>() : nullptr;
        MMF_RC(mmf_adamw_multi(&d, sp()), "mmf_adamw_multi");
    }
}
bool svc_ln_defer_set(bool on) { const bool old = g_ln_defer; g_ln_defer = on; return old; }
void svc_ln_defer_flush() { ln_flush(); }
int64_t svc_wgrad_joint_launches() { return g_wgrad_joint; }
bool svc_wgrad_hold_set(bool on) { const bool old = g_wgrad_hold; g_wgrad_hold = on; return old; }
int64_t svc_set_py_mode(int64_t mode) { const int64_t old = g_py_mode; g_py_mode = mode; return old; }

}  // namespace

TORCH_LIBRARY(mmf_amd, m) {
    m.def("additive_mask(Tensor mask) -> Tensor");
    m.def("visual_masks(Tensor input_mask, Tensor? image_dim, int R) -> (Tensor, Tensor, Tensor, Tensor, Tensor)");
    m.def("visio_linguistic_embeddings(Tensor input_ids, Tensor token_type_ids, Tensor? visual_embeddings, Tensor? visual_embeddings_type, "
          "Tensor word, Tensor pos, Tensor typ, Tensor ln_w, Tensor ln_b, Tensor typ_vis, Tensor pos_vis, Tensor proj_w, Tensor proj_b, "
          "float eps, float p, bool training, int pad_idx, Tensor? image_text_alignment=None) -> Tensor");
    m.def("transformer_layer(Tensor x, Tensor wq, Tensor bq, Tensor wk, Tensor bk, Tensor wv, Tensor bv, Tensor wo, Tensor bo, "
          "Tensor ln1_w, Tensor ln1_b, Tensor w1, Tensor b1, Tensor w2, Tensor b2, Tensor ln2_w, Tensor ln2_b, Tensor? mask_add, "
          "int heads, float eps1, float eps2, float p_attn, float p_hid1, float p_hid2, bool training, int causal_tail) -> Tensor");
    m.def("linear(Tensor x, Tensor weight, Tensor? bias, bool out_f32) -> Tensor");
    m.def("layer_norm(Tensor x, Tensor weight, Tensor bias, float eps) -> Tensor");
    m.def("dense_gelu(Tensor x, Tensor weight, Tensor bias) -> Tensor");
    m.def("linear_tanh(Tensor x, Tensor weight, Tensor bias) -> Tensor");
    m.def("gather_rows(Tensor x, Tensor index, float p, bool training) -> Tensor");
    m.def("dropout(Tensor x, float p, bool training) -> Tensor");
    m.def("pair_halves(Tensor x) -> Tensor");
    m.def("logit_bce(Tensor scores, Tensor targets) -> Tensor");
    m.def("masked_lm_head(Tensor x, Tensor weight, Tensor bias, Tensor labels, int ignore_index) -> (Tensor, Tensor)");
    m.def("masked_region_head(Tensor x, Tensor weight, Tensor bias, Tensor target, Tensor row_label) -> (Tensor, Tensor)");
    m.def("_py_masked_lm_head(Tensor x, Tensor weight, Tensor bias, Tensor labels, int ignore_index) -> (Tensor, Tensor)");
    m.def("_py_masked_region_head(Tensor x, Tensor weight, Tensor bias, Tensor target, Tensor row_label) -> (Tensor, Tensor)");
    // Python-implemented twins (fp32-accurate forward path, opt-in experiment hooks)
    m.def("_py_visio_linguistic_embeddings(Tensor input_ids, Tensor token_type_ids, Tensor? visual_embeddings, Tensor? visual_embeddings_type, "
          "Tensor word, Tensor pos, Tensor typ, Tensor ln_w, Tensor ln_b, Tensor typ_vis, Tensor pos_vis, Tensor proj_w, Tensor proj_b, "
          "float eps, float p, bool training, int pad_idx, Tensor? image_text_alignment=None) -> Tensor");
    m.def("_py_transformer_layer(Tensor x, Tensor wq, Tensor bq, Tensor wk, Tensor bk, Tensor wv, Tensor bv, Tensor wo, Tensor bo, "
          "Tensor ln1_w, Tensor ln1_b, Tensor w1, Tensor b1, Tensor w2, Tensor b2, Tensor ln2_w, Tensor ln2_b, Tensor? mask_add, "
          "int heads, float eps1, float eps2, float p_attn, float p_hid1, float p_hid2, bool training, int causal_tail) -> Tensor");
    m.def("_py_linear(Tensor x, Tensor weight, Tensor? bias, bool out_f32) -> Tensor");
    m.def("_py_layer_norm(Tensor x, Tensor weight, Tensor bias, float eps) -> Tensor");
    m.def("_py_dense_gelu(Tensor x, Tensor weight, Tensor bias) -> Tensor");
    m.def("_py_linear_tanh(Tensor x, Tensor weight, Tensor bias) -> Tensor");
    m.def("_py_gather_rows(Tensor x, Tensor index, float p, bool training) -> Tensor");
    m.def("_py_dropout(Tensor x, float p, bool training) -> Tensor");
    m.def("_py_pair_halves(Tensor x) -> Tensor");
    // service operators
    m.def("_shadow_get(Tensor[] params, bool as_f32) -> Tensor");
    m.def("_shadow_slot(Tensor p) -> Tensor[]");
    m.def("_shadow_transposed(Tensor w16) -> Tensor[]");
    m.def("_shadow_clear() -> ()");
    m.def("_adamw_step(Tensor[] p, Tensor[] g, Tensor[] m, Tensor[] v, float[] lr, float[] wd, float beta1, float beta2, float eps, int step, "
          "bool correct_bias, int mode, float grad_scale, Tensor? norm_sq, float max_norm, Tensor? dev_state) -> ()");
    m.def("_shadow_set_twins(bool on) -> bool");
    m.def("_shadow_refresh_transposed(Tensor[] only, bool has_only, Tensor[] skip, bool has_skip) -> ()");
    m.def("_drop_graph_mode(Tensor? seed) -> ()");
    m.def("_drop_next() -> (int, Tensor[])");
    m.def("_ln_defer_set(bool on) -> bool");
    m.def("_ln_defer_flush() -> ()");
    m.def("_wgrad_hold_set(bool on) -> bool");
    m.def("_wgrad_joint_launches() -> int");
    m.def("_set_py_mode(int mode) -> int");
}

TORCH_LIBRARY_IMPL(mmf_amd, CompositeImplicitAutograd, m) {
    m.impl("additive_mask", op_additive_mask);
    m.impl("visual_masks", op_visual_masks);
    m.impl("visio_linguistic_embeddings", op_vle);
    m.impl("transformer_layer", op_transformer_layer);
    m.impl("linear", op_linear);
    m.impl("layer_norm", op_layer_norm);
    m.impl("dense_gelu", op_dense_gelu);
    m.impl("linear_tanh", op_linear_tanh);
    m.impl("gather_rows", op_gather_rows);
    m.impl("dropout", op_dropout);
    m.impl("pair_halves", op_pair_halves);
    m.impl("logit_bce", op_logit_bce);
    m.impl("masked_lm_head", op_masked_lm_head);
    m.impl("masked_region_head", op_masked_region_head);
    m.impl("_shadow_get", svc_shadow_get);
    m.impl("_shadow_slot", svc_shadow_slot);
    m.impl("_shadow_transposed", svc_shadow_transposed);
    m.impl("_shadow_clear", svc_shadow_clear);
    m.impl("_adamw_step", svc_adamw_step);
    m.impl("_shadow_set_twins", svc_shadow_set_twins);
    m.impl("_shadow_refresh_transposed", svc_shadow_refresh_transposed);
    m.impl("_drop_graph_mode", svc_drop_graph_mode);
    m.impl("_drop_next", svc_drop_next);
    m.impl("_ln_defer_set", svc_ln_defer_set);
    m.impl("_ln_defer_flush", svc_ln_defer_flush);
    m.impl("_wgrad_hold_set", svc_wgrad_hold_set);
    m.impl("_wgrad_joint_launches", svc_wgrad_joint_launches);
    m.impl("_set_py_mode", svc_set_py_mode);
}
