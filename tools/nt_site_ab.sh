#!/bin/bash
# Same-box A/B of the per-call-site exception to the non-temporal epilogue stores (MMF_TUN_NT_SITE_KEEP).
# Round 3 made EVERY GEMM output non-temporal (8.62 -> 8.41 ms): the operand panels stay in L2.  But an output that is the very next
# kernel's input then comes back from HBM: in situ the FFN-down forward's K-loop runs 0.76 us per step against 0.63 on warm operands and
# the narrow GEMMs' epilogues 7.1 us against 2.8 (profiles/r02_gemm_timeline_wide.txt vs r02_wide_gemm_ablation.txt) - 80 us per layer,
# 0.96 ms per step, between the isolated and the in-situ GEMM times.  This script measures, interleaved on ONE box, which tagged outputs
# (include/mmf_amd.h MMF_SITE_*: 1 QKV -> attention, 2 out-proj -> LayerNorm, 3 FFN-up -> FFN-down's A operand, 4 FFN-down -> LayerNorm,
# 5 du -> FFN-up dgrad, 6 da -> LayerNorm backward, 7 dctx -> attention backward, 8 dx -> the layer below) are better stored temporally.
#   GPU box:  bash tools/nt_site_ab.sh [quick]          (writes gpurun_out/nt_site_ab/ab.log; ~20 s per line)
# Bake the winner into the default of MMF_TUN_NT_SITE_KEEP (mmf_amd/csrc/lib.hip g_tun) and record the table in DESIGN.md section 6.
export TMPDIR=/tmp
out=gpurun_out/nt_site_ab
mkdir -p $out
line() { python -c "import sys,json; l=[x for x in sys.stdin if x.startswith('{')]; d=json.loads(l[-1]); g=d['roofline']['all_gemm']; print('$1', d['ms_per_step'], d['value'], g['tflops'], d['roofline']['attention'])"; }
run() { MMF_AMD_NT_SITE_KEEP=$1 python bench.py --no-cpu-baseline --no-fp32 2>/dev/null | line "keep=$1" | tee -a $out/ab.log; }
# masks: bit s = site s.  0 = the default; 0x1fe = every tagged output temporal (close to the round-2 policy, U and fp32 outputs stay NT)
if [ "$1" = "quick" ]; then MASKS="0 0x08 0x28 0x2a 0x1fe 0"; else MASKS="0 0x02 0x04 0x08 0x10 0x20 0x40 0x80 0x100 0x28 0x2a 0xaa 0x1fe 0"; fi
for m in $MASKS; do run $m; done
for m in $MASKS; do run $m; done
sort -k2 -n $out/ab.log | head -8
# second question on the same box: the GELU up-projection on a wide tile (MMF_TUN_GELU_WIDE; 70.7 vs 60.5 us before the non-temporal stores)
for g in 0 3 1 0 3 1; do MMF_AMD_GELU_WIDE=$g python bench.py --no-cpu-baseline --no-fp32 2>/dev/null | line "gelu_wide=$g" | tee -a $out/gelu.log; done
