#!/bin/bash
# Same-box A/B of the experiment build with non-temporal epilogue stores against the regular library:
# step time twice each, then the FETCH_SIZE pass of both (HBM bytes fetched per launch of every GEMM family).
#   build container:  MMF_AMD_EXTRA_HIPCC_FLAGS=-DMMF_EPI_NT_STORES python -m mmf_amd.csrc.build --tag nt
#   GPU box:          bash tools/nt_stores_ab.sh          (writes gpurun_out/nt_ab/)
export TMPDIR=/tmp
out=gpurun_out/nt_ab
mkdir -p $out
NT=$PWD/mmf_amd/libmmf_amd.nt.so
[ -f "$NT" ] || { echo "build the experiment library first (see the header of this script)"; exit 1; }
line() { python -c "import sys,json; l=[x for x in sys.stdin if x.startswith('{')]; d=json.loads(l[-1]); print('$1', d['ms_per_step'], d['value'], d['roofline']['all_gemm']['tflops'])"; }
for i in 1 2; do
  python bench.py --no-cpu-baseline 2>/dev/null | line "regular" | tee -a $out/ab.log
  MMF_AMD_LIB=$NT python bench.py --no-cpu-baseline 2>/dev/null | line "nt-stores" | tee -a $out/ab.log
done
CMD="python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-graph"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/fetch_regular -o p -- $CMD > $out/fetch_regular.log 2>&1
MMF_AMD_LIB=$NT rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/fetch_nt -o p -- $CMD > $out/fetch_nt.log 2>&1
python - <<'PY'
import csv, glob, collections
for tag in ("regular", "nt"):
    f = glob.glob("gpurun_out/nt_ab/fetch_%s/*counter_collection.csv" % tag)
    if not f:
        continue
    by = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f[0])):
        if r.get("Counter_Name") == "FETCH_SIZE" and "gemm" in r["Kernel_Name"]:
            k = "wide" if "wide" in r["Kernel_Name"] else ("grouped" if "grouped" in r["Kernel_Name"] else "128x128")
            by[k][0] += 1; by[k][1] += float(r["Counter_Value"])
    print(tag, {k: "%.1f MB/launch (x2 correction applied)" % (2 * v[1] * 1024 / v[0] / 1e6) for k, v in by.items()})
PY
