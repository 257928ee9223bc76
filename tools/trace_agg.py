"""Aggregate a rocprofv3 --kernel-trace CSV by kernel: python tools/trace_agg.py <t_kernel_trace.csv> [steps]."""
import collections
import csv
import re
import sys


def short(n):
    n = n.replace("void ", "").replace("(anonymous namespace)::", "")
    m = re.match(r"_ZN12_GLOBAL__N_1\d+([a-z0-9_]+?)I(.*)", n)
    if m:
        return m.group(1) + "<" + m.group(2)[:44] + ">"
    return re.sub(r"\(.*", "", n)[:70]


def main(path, steps=8):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        agg[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    tot = sum(sum(v) for v in agg.values())
    print("calls/step  us/step   avg_us   min_us   max_us    pct  kernel")
    for n, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        if sum(v) / tot < 0.001:
            continue
        print("%9.1f %9.1f %8.2f %8.2f %8.2f %5.1f%%  %s" % (len(v) / steps, sum(v) / steps, sum(v) / len(v), min(v), max(v), 100 * sum(v) / tot, n))
    print("total %.3f ms/step over %d dispatches" % (tot / steps / 1e3, sum(len(v) for v in agg.values())))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 8)
