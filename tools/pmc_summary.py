"""Average PMC counter values per kernel (name, grid) from a rocprofv3 --pmc csv."""
import collections, csv, sys
def main(path, filt="gemm"):
    rows = list(csv.DictReader(open(path)))
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    for r in rows:
        k = (r["Kernel_Name"].replace("(anonymous namespace)::", "")[:70], r["Grid_Size"])
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for k, d in agg.items():
        if filt not in k[0]:
            continue
        avg = {c: sum(v) / len(v) for c, v in d.items()}
        print(k, "avg_us=%.1f" % (sum(dur[k]) / len(dur[k])))
        print("    " + "  ".join("%s=%.4g" % (c, v) for c, v in sorted(avg.items())))
if __name__ == "__main__":
    main(*sys.argv[1:])
