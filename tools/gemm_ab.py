"""Same-process, interleaved A/B of GEMM kernel variants on the VisualBERT VQA2 layer shapes WITH their real epilogues, plus a
correctness check of every variant against an fp32 torch product of the same bf16 operands.

    python tools/gemm_ab.py [--tun ID:V1,V2,...] [--set ID:V] [--M rows] [--rounds R] [--iters I] [shape-name-filter ...]

Default: tunable MMF_TUN_GEMM_PERSIST (18) in {-1, 0}: the one-tile kernels against the persistent kernel where the rule takes it.  Every (shape, variant) is timed `rounds` times, `iters` launches each, the
variants interleaved inside a round (cdna_hip_programming.md section 5.4 rule 24); the table prints median / min per variant."""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from mmf_amd import _native as nat

M = 7296
# name, N, K, epilogue kind
SHAPES = [
    ("qkv_fwd", 2304, 768, "bias"),
    ("out_fwd", 768, 768, "bias_drop_resid"),
    ("ffn1_fwd", 3072, 768, "gelu"),
    ("ffn2_fwd", 768, 3072, "bias_drop_resid"),
    ("ffn2_dgrad", 3072, 768, "gelup"),
    ("ffn1_dgrad", 768, 3072, "resid"),
    ("out_dgrad", 768, 768, "plain"),
    ("qkv_dgrad", 768, 2304, "resid"),
]


def make(name, N, K, kind, dev="cuda"):
    g = torch.Generator(device=dev).manual_seed(hash(name) % 1000)
    A = (torch.randn(M, K, device=dev, generator=g) * 0.5).bfloat16()
    B = (torch.randn(N, K, device=dev, generator=g) * 0.05).bfloat16()
    C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    bias = torch.randn(N, device=dev, generator=g) * 0.1
    resid = torch.randn(M, N, device=dev, generator=g).bfloat16()
    aux = torch.rand(M, N, device=dev, generator=g).bfloat16()
    U = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    ref = A.float() @ B.float().t()
    kw = {}
    if kind == "bias":
        kw = dict(bias=bias); ref = ref + bias
    elif kind == "bias_drop_resid":      # timing with dropout on; the check runs without (the mask is a hash of the element index)
        kw = dict(bias=bias, resid=resid, ldr=N); ref = ref + bias + resid.float()
    elif kind == "gelu":
        kw = dict(bias=bias, act=1, U=U); ref = torch.nn.functional.gelu(ref + bias)
    elif kind == "gelup":
        kw = dict(act=2, aux=aux); ref = ref * aux.float()
    elif kind == "resid":
        kw = dict(resid=resid, ldr=N); ref = ref + resid.float()
    timed_kw = dict(kw)
    if kind == "bias_drop_resid":
        timed_kw["drop"] = nat.drop_cfg(0.1, 12345, None) if hasattr(nat, "drop_cfg") else nat.NO_DROP
    return A, B, C, kw, timed_kw, ref, K, N


def main():
    args = sys.argv[1:]
    tun, vals, rounds, iters, filt, fixed = 18, [-1, 0], 7, 20, [], []
    i = 0
    while i < len(args):
        if args[i] == "--tun":
            t, v = args[i + 1].split(":"); tun, vals = int(t), [int(x) for x in v.split(",")]; i += 2
        elif args[i] == "--set":          # a fixed tunable for the whole run, e.g. --set 2:3 (force the 256x128 wide tile)
            t, v = args[i + 1].split(":"); fixed.append((int(t), int(v))); i += 2
        elif args[i] == "--rounds":
            rounds = int(args[i + 1]); i += 2
        elif args[i] == "--iters":
            iters = int(args[i + 1]); i += 2
        elif args[i] == "--shapes":       # "name:N:K:kind,..." instead of the VisualBERT layer table (kinds: bias, bias_drop_resid, gelu, gelup, resid, plain)
            global SHAPES
            SHAPES = [(a, int(b), int(c), d) for a, b, c, d in (x.split(":") for x in args[i + 1].split(","))]; i += 2
        elif args[i] == "--M":            # token rows (default 7296 = 32 x 228; a trimmed batch of 24 + 100 positions: 3968)
            global M
            M = int(args[i + 1]); i += 2
        else:
            filt.append(args[i]); i += 1
    L = nat.lib()
    for t, v in fixed:
        L.mmf_amd_set_tunable(t, v)
    tot = {v: 0.0 for v in vals}
    flops = 0.0
    for name, N, K, kind in SHAPES:
        if filt and not any(f in name for f in filt):
            continue
        A, B, C, kw, tkw, ref, K, N = make(name, N, K, kind)
        errs, kern, dd = {}, {}, {}
        first_drop = None
        for v in vals:
            L.mmf_amd_set_tunable(tun, v)
            C.fill_(float("nan"))
            if "U" in kw:
                kw["U"].fill_(float("nan"))
            nat.gemm(A, B, C, M, N, K, K, K, N, **kw)
            torch.cuda.synchronize()
            kern[v] = nat.gemm_last_kernel()
            errs[v] = float((C.float() - ref).abs().max() / ref.abs().max())
            if "U" in kw:      # the saved gelu' against autograd's
                x = (A.float() @ B.float().t() + kw["bias"]).requires_grad_(True)
                g, = torch.autograd.grad(torch.nn.functional.gelu(x).sum(), x)
                errs[v] = max(errs[v], float((kw["U"].float() - g).abs().max()))
            # with the timed epilogue (dropout on): every variant must draw the same mask -> outputs equal up to the summation order
            C.fill_(float("nan"))
            nat.gemm(A, B, C, M, N, K, K, K, N, **tkw)
            torch.cuda.synchronize()
            if first_drop is None:
                first_drop = C.clone(); dd[v] = 0.0
            else:
                dd[v] = float((C.float() - first_drop.float()).abs().max())
        times = {v: [] for v in vals}
        for r in range(rounds):
            for v in vals:
                L.mmf_amd_set_tunable(tun, v)
                for _ in range(3):
                    nat.gemm(A, B, C, M, N, K, K, K, N, **tkw)
                e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(iters):
                    nat.gemm(A, B, C, M, N, K, K, K, N, **tkw)
                e1.record(); torch.cuda.synchronize()
                times[v].append(e0.elapsed_time(e1) / iters * 1e3)
        fl = 2.0 * M * N * K
        flops += fl
        row = "%-11s N=%4d K=%4d " % (name, N, K)
        for v in vals:
            med = statistics.median(times[v]); tot[v] += med
            row += "| v=%d %-22s med %6.1f us min %6.1f (%4.0f TF) relerr %.1e dvs0 %.1e " % (v, kern[v][-18:], med, min(times[v]), fl / med / 1e6, errs[v], dd[v])
        print(row, flush=True)
    print("sum of medians: " + "  ".join("v=%d %.1f us (%.0f TF)" % (v, tot[v], flops / tot[v] / 1e6) for v in vals))
    L.mmf_amd_set_tunable(tun, 0)


if __name__ == "__main__":
    main()
