"""Aggregate a rocprofv3 --kernel-trace CSV by kernel: python tools/trace_agg.py <t_kernel_trace.csv> [steps]
   python tools/trace_agg.py <csv> --periods N [marker]: only the LAST N periods of the trace, a period running from one group of `marker` launches
   (default adamw_multi_kernel: the tail of a training step) to the next — i.e. the replayed graph without the warm-up / capture passes; also prints
   the wall-clock length of a period and the sum of kernel time inside it (two streams overlap: the sum may exceed the period)."""
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


def periods(path, n, marker="adamw_multi_kernel"):
    rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])) for r in csv.DictReader(open(path))))
    ends = [i for i in range(len(rows)) if marker in rows[i][2] and (i + 1 == len(rows) or marker not in rows[i + 1][2])]
    ends = [e for j, e in enumerate(ends) if j == 0 or e - ends[j - 1] > 8]       # (one boundary per step: the last launch of the marker group)
    cuts = ends[-(n + 1):]
    if len(cuts) < 2:
        raise SystemExit("fewer than two '%s' groups in the trace" % marker)
    sel = rows[cuts[0] + 1: cuts[-1] + 1]
    k = len(cuts) - 1
    agg = collections.defaultdict(list)
    for s0, e0, name in sel:
        agg[name].append((e0 - s0) / 1e3)
    tot = sum(sum(v) for v in agg.values())
    print("last %d periods: %.3f ms per period (wall), %.3f ms of kernel time, %.1f launches per period" % (k, (rows[cuts[-1]][1] - rows[cuts[0]][1]) / k / 1e6, tot / k / 1e3, len(sel) / k))
    print("calls/step  us/step   avg_us   min_us   max_us    pct  kernel")
    for name, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        if sum(v) / tot < 0.001:
            continue
        print("%9.1f %9.1f %8.2f %8.2f %8.2f %5.1f%%  %s" % (len(v) / k, sum(v) / k, sum(v) / len(v), min(v), max(v), 100 * sum(v) / tot, name))


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[2] == "--periods":
        periods(sys.argv[1], int(sys.argv[3]), *(sys.argv[4:5]))
    else:
        main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 8)
