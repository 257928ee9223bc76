"""Turn gpurun_out/profiles_<tag>/ (written by tools/make_profiles.sh on the GPU box) into the committed
summaries under profiles/: per-kernel stats, PMC-derived HBM traffic per launch (FETCH_SIZE doubled as
MI355X_MICROARCH.md prescribes for wide coalesced reads on gfx950), SQ counters of the GEMM kernels."""
import collections, csv, glob, json, os, re, sys

_DEMANGLED = {}


def demangle(name):
    if name.startswith("_Z") and name not in _DEMANGLED:
        import subprocess
        try:
            filt = "/opt/rocm/lib/llvm/bin/llvm-cxxfilt"       # (binutils' c++filt does not know the bf16 mangling DF16b)
            _DEMANGLED[name] = subprocess.run([filt if os.path.exists(filt) else "c++filt", name], capture_output=True, text=True).stdout.strip() or name
        except OSError:
            _DEMANGLED[name] = name
    return _DEMANGLED.get(name, name)


def anon_name(name):
    """kernel<ints> out of a mangled anonymous-namespace symbol the installed demanglers cannot read (bf16 = DF16b)."""
    m = re.match(r"_ZN12_GLOBAL__N_1(\d+)", name)
    if not m:
        return None
    n = int(m.group(1)); p0 = m.end()
    fname, rest = name[p0:p0 + n], name[p0 + n:]
    targs = re.findall(r"L[ib](\d+)E", rest.split("EEv")[0]) if rest.startswith("I") else []
    return fname + ("<%s>" % ", ".join(targs) if targs else "")


def short(name):
    v = variant(name)
    if v is not None and ("grouped" in v or "wide" in v):
        return v.split(" | ")[1].replace(" ", "<", 1) + ">"
    if "gemm_bf16_kernelI" not in name:
        name = demangle(name)
        if name.startswith("_ZN12_GLOBAL__N_1"):
            return anon_name(name) or name[:80]
    name = name.replace("(anonymous namespace)::", "")
    m = re.search(r"gemm_bf16_kernelI(DF16b|f)(DF16b|f)Lb(\d)ELb(\d)ELb(\d)ELi(\d)ELi(\d+)E(?:Li(\d)ELb(\d)E)?", name)
    if m:
        a, b, ak, bk, rg, nwn, bn, ks, rs = m.groups()
        extra = ("" if ks in (None, "1") else ",Ksplit") + ("" if rs in (None, "0") else ",+bias-grad")
        return "gemm_bf16_kernel<A=%s,B=%s,%s%s,%s,%dwaves,BN=%s%s>" % ("bf16" if a != "f" else "f32", "bf16" if b != "f" else "f32",
            "T" if ak == "1" else "N", "N" if bk == "1" else "T", "ragged" if rg == "1" else "full", 2 * int(nwn), bn, extra)
    m = re.search(r"gemm_bf16_kernel<([^>]*)>", name)
    if m:
        return "gemm_bf16_kernel<%s>" % m.group(1)
    return re.sub(r"\(.*", "", name)[:80]

OTHER_FAMILIES = [("attn_fwd8_kernel", "attention fwd | attn_fwd8_kernel"), ("attn_fwd_kernel", "attention fwd | attn_fwd_kernel"),
                  ("attn_bwd_fused_kernel", "attention bwd | attn_bwd_fused_kernel"), ("attn_bwd_dq_kernel", "attention bwd | attn_bwd_dq_kernel"),
                  ("attn_bwd_dkv_kernel", "attention bwd | attn_bwd_dkv_kernel"), ("ln_bwd_h_kernel", "layernorm bwd | ln_bwd_h_kernel"),
                  ("ln_fwd_h_kernel", "layernorm fwd | ln_fwd_h_kernel"), ("adamw_multi_kernel", "adamw | adamw_multi_kernel"),
                  ("transpose_multi_kernel", "weight twins | transpose_multi_kernel"), ("scatter_add_unique_kernel", "word-embedding gradient | scatter_add_unique_kernel"),
                  ("ln_bwd_reduce_multi_kernel", "layernorm bwd | ln_bwd_reduce_multi_kernel"), ("splitk_epilogue_kernel", "gemm skinny | splitk_epilogue_kernel"),
                  ("splitk_reduce_kernel", "gemm split-K | splitk_reduce_kernel")]


def variant(name):
    """The label bench.py's KernelProbe gives the launches of this kernel ("gemm <form> | <kernel family and tile>[ ksplit]
    [ (ragged / small)]"), so that profiles/pmc_traffic.json can be looked up with the bench line's dominant-kernel label."""
    if "gemm_wide_grouped_ln_kernel" in name:
        return "gemm TN grouped wgrad + LayerNorm bwd rider | gemm_wide_grouped_ln_kernel 256x128"
    if "gemm_wide_grouped_kernel" in name:
        return "gemm TN grouped wgrad | gemm_wide_grouped_kernel 256x128"
    if "gemm_bf16_grouped_kernel" in name:
        return "gemm TN grouped wgrad | gemm_bf16_grouped_kernel 128x128"
    m = re.search(r"gemm_persist_kernel<(\d+), (\d+)", name) or re.search(r"gemm_persist_kernelILi(\d+)ELi(\d+)E", name)
    if m:
        return "gemm NT | gemm_persist_kernel %sx%s" % (m.group(1), m.group(2))
    m = re.search(r"gemm_wide_kernel<(\d+), (\d+)", name) or re.search(r"gemm_wide_kernelILi(\d+)ELi(\d+)E", name)
    if m:
        return "gemm NT | gemm_wide_kernel %sx%s" % (m.group(1), m.group(2))
    m = re.search(r"gemm_bf16_kernelI(?:DF16b|f)(?:DF16b|f)Lb(\d)ELb(\d)ELb(\d)ELi(\d)ELi(\d+)ELi(\d)E", name)
    if not m:
        # every other kernel family that takes a visible share of the step, under the name bench.py's roofline block uses for it
        for key, label in OTHER_FAMILIES:
            if key in name:
                return label
        return None
    ak, bk, rg, nwn, bn, ks = m.groups()
    return "gemm %s%s | gemm_bf16_kernel 128x%s%s%s" % ("T" if ak == "1" else "N", "N" if bk == "1" else "T", bn,
                                                       " ksplit" if ks == "2" else "", " (ragged / small)" if rg == "1" else "")

def main(tag):
    src = "gpurun_out/profiles_%s" % tag
    os.makedirs("profiles", exist_ok=True)
    # ---- kernel stats
    rows = list(csv.DictReader(open(glob.glob(src + "/trace/*kernel_trace.csv")[0])))
    agg = collections.defaultdict(list)
    for r in rows:
        agg[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    tot = sum(sum(v) for v in agg.values())
    nsteps = max(1, sum(1 for r in rows if "embed_text_kernel" in r["Kernel_Name"]))      # one text-embedding launch per step
    lines = ["# rocprofv3 --kernel-trace --stats of: python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-graph  (%d steps incl. warm-up, the eager-enqueue leg and 1 instrumented)" % nsteps,
             "%-8s %-12s %-10s %-10s %-10s %-7s %s" % ("calls", "total_us", "avg_us", "min_us", "max_us", "pct", "kernel")]
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        lines.append("%-8d %-12.1f %-10.2f %-10.2f %-10.2f %-7.2f %s" % (len(v), sum(v), sum(v) / len(v), min(v), max(v), 100 * sum(v) / tot, k))
    lines.append("TOTAL kernel time %.1f us over %d dispatches, %d steps (%.2f ms of kernel time per step; optimizer-state zero-fills of the two optimizers included)" % (tot, len(rows), nsteps, tot / nsteps / 1e3))
    open("profiles/%s_kernel_stats.txt" % tag, "w").write("\n".join(lines) + "\n")
    raw = glob.glob(src + "/trace/*kernel_stats.csv")       # rocprofv3's own --stats table, verbatim
    if raw:
        open("profiles/%s_rocprofv3_kernel_stats.csv" % tag, "w").write(open(raw[0]).read())
    # ---- traffic
    traffic = {}
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    for which in ("fetch", "write"):
        f = glob.glob(src + "/pmc_%s/*counter_collection.csv" % which)
        if not f:
            continue
        for r in csv.DictReader(open(f[0])):
            v = variant(r["Kernel_Name"])
            if v:
                per[v][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for v, d in per.items():
        fetch_kb = sum(d.get("FETCH_SIZE", [0])) / max(1, len(d.get("FETCH_SIZE", [1])))
        write_kb = sum(d.get("WRITE_SIZE", [0])) / max(1, len(d.get("WRITE_SIZE", [1])))
        traffic[v] = {"bytes_per_launch": int((2.0 * fetch_kb + write_kb) * 1024), "fetch_size_kb_raw": round(fetch_kb, 1),
                      "write_size_kb_raw": round(write_kb, 1), "launches": len(d.get("FETCH_SIZE", [])),
                      "note": "FETCH_SIZE doubled (gfx950 counts 128-B requests as 64 B for wide coalesced reads)"}
    if traffic:      # a trace-only refresh keeps the PMC summaries of the last full run
        import subprocess, time
        head = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
        flat = {k: v["bytes_per_launch"] for k, v in traffic.items()}
        flat["_collected"] = "%s, tree at %s, tools/make_profiles.sh %s" % (time.strftime("%Y-%m-%d"), head or "?", tag)
        json.dump(flat, open("profiles/pmc_traffic.json", "w"), indent=1)
        json.dump(traffic, open("profiles/%s_pmc_traffic_detail.json" % tag, "w"), indent=1)
    # ---- SQ counters
    f = glob.glob(src + "/pmc_sq/*counter_collection.csv")
    if f:
        sq = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f[0])):
            v = variant(r["Kernel_Name"])
            if v:
                sq[v][r["Counter_Name"]].append(float(r["Counter_Value"]))
        out = {}
        for v, d in sq.items():
            a = {c: sum(x) / len(x) for c, x in d.items()}
            wc = a.get("SQ_WAVE_CYCLES", 1.0)
            out[v] = {"avg": {c: round(x, 1) for c, x in a.items()},
                      "wait_any_frac": round(a.get("SQ_WAIT_ANY", 0) / wc, 3), "wait_inst_frac": round(a.get("SQ_WAIT_INST_ANY", 0) / wc, 3),
                      "active_frac": round(a.get("SQ_ACTIVE_INST_ANY", 0) / wc, 3),
                      "lds_bank_conflict_per_lds_cycle": round(a.get("SQ_LDS_BANK_CONFLICT", 0) / max(1.0, a.get("SQ_LDS_IDX_ACTIVE", 1)), 4)}
        json.dump(out, open("profiles/%s_pmc_sq.json" % tag, "w"), indent=1)
    print(open("profiles/%s_kernel_stats.txt" % tag).read()[:3000])
    print(json.dumps(traffic, indent=1))

if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "r01")
