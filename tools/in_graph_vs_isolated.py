"""Every kernel of one encoder layer inside the REPLAYED hipGraph of the training step beside the same GEMMs timed in isolation on warm operands in
the same gpurun call.  Inputs (written by `tools/make_profiles.sh TAG` + the two extra commands named in profiles/README.md):

    gpurun_out/profiles_TAG/graph/g_kernel_trace.csv     rocprofv3 --kernel-trace -- python tools/graph_gaps.py run
    gpurun_out/profiles_TAG/gemm_isolated.log            python tools/gemm_ab.py --tun 18:0 --rounds 5

    python tools/in_graph_vs_isolated.py TAG > profiles/TAG_in_graph_vs_isolated.txt"""
import csv
import re
import statistics
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
base = "gpurun_out/profiles_%s/" % tag
rows = []
with open(base + "graph/g_kernel_trace.csv") as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
cuts = [i for i, r in enumerate(rows) if ("seed_advance_kernel" in r[2] or "step_advance_kernel" in r[2])]


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    m = re.match(r"_ZN12_GLOBAL__N_1(\d+)", n)
    if m:
        k = int(m.group(1)); n = n[m.end():m.end() + k]
    return re.sub(r"\(.*", "", n)[:64]


reps = [rows[a:b] for a, b in zip(cuts[-10:-1], cuts[-9:])]
n = min(len(r) for r in reps)
seq = [(short(reps[0][i][2]), statistics.median((r[i][1] - r[i][0]) / 1e3 for r in reps)) for i in range(n)]
per = statistics.median((r[-1][1] - r[0][0]) / 1e3 for r in reps)
print("# The replayed hipGraph of the training step, kernel by kernel (median over 9 replays), for one encoder layer forward and backward, beside the")
print("# SAME kernels timed in isolation on warm operands in the same gpurun call (tools/gemm_ab.py --tun 18:0).  Replay period %.1f us, %d kernels." % (per, n))
iso = {}
for l in open(base + "gemm_isolated.log"):
    m = re.match(r"(\w+)\s+N=\s*(\d+) K=\s*(\d+) \| v=0 (\S+ \S+)\s+med\s+([\d.]+) us", l)
    if m:
        iso[m.group(1)] = float(m.group(5))
names_f = ["layernorm fwd (the previous layer's output LN)", "qkv_fwd", "attention fwd", "out_fwd", "layernorm fwd", "ffn1_fwd", "ffn2_fwd"]
idx = [i for i, k in enumerate(seq) if "attn_fwd8" in k[0]]
i0 = idx[5] - 2
print("\n## forward, layer 5                              in graph    isolated (warm)   kernel")
for nm, k in zip(names_f, seq[i0:i0 + 7]):
    print("%-48s %7.1f us   %s   %s" % (nm, k[1], ("%7.1f us" % iso[nm]) if nm in iso else "      -   ", k[0]))
names_b = ["layernorm bwd", "ffn2_dgrad", "ffn1_dgrad", "layernorm bwd", "out_dgrad", "attention bwd", "qkv_dgrad", "grouped weight gradients (4 GEMMs)"]
if any("grouped_ln" in k[0] for k in seq):      # the first LayerNorm backward of a layer rides on the weight-gradient launch of the layer above (gemm_wide_grouped_ln_kernel)
    names_b = ["weight gradients of the layer above + layernorm bwd"] + names_b[1:7] + ["weight gradients + layernorm bwd of the layer below"]
idx = [i for i, k in enumerate(seq) if "attn_bwd_fused" in k[0]]
i0 = idx[5] - 5
print("\n## backward, one layer                           in graph    isolated (warm)   kernel")
for nm, k in zip(names_b, seq[i0:i0 + 8]):
    print("%-48s %7.1f us   %s   %s" % (nm, k[1], ("%7.1f us" % iso[nm]) if nm in iso else "      -   ", k[0]))
tot = {}
for k in seq:
    tot.setdefault(k[0], [0, 0.0]); tot[k[0]][0] += 1; tot[k[0]][1] += k[1]
print("\n## whole replay by kernel")
for k, v in sorted(tot.items(), key=lambda kv: -kv[1][1])[:18]:
    print("%4d x %8.1f us = %8.1f us  %s" % (v[0], v[1] / v[0], v[1], k))
