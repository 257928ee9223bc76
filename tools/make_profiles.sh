#!/bin/bash
# Run on the GPU box: kernel-trace stats + HBM-traffic PMC passes of the default bench command, then the graph replay trace, the isolated GEMM
# timings and the default bench line of the same tree on the same box.
# Outputs go to gpurun_out/profiles_$1/ ; tools/collect_profiles.py turns them into profiles/*.
tag=${1:-r01}
out=gpurun_out/profiles_$tag
export TMPDIR=/tmp
mkdir -p $out
CMD="python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-graph --no-fp32"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o t -- $CMD > $out/trace.log 2>&1
[ "$2" = "trace-only" ] && { grep -h metric $out/trace.log | cut -c1-300; exit 0; }
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -o p -- $CMD > $out/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -o p -- $CMD > $out/pmc_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $out/pmc_sq -o p -- $CMD > $out/pmc_sq.log 2>&1
# the replayed hipGraph of the step kernel by kernel, the layer GEMMs in isolation on the same box, and the default bench line
# (tools/graph_gaps.py report, tools/in_graph_vs_isolated.py and tools/collect_profiles.py turn these into profiles/$tag_*)
(cd /tmp; rocprofv3 --kernel-trace --output-format csv -d $OLDPWD/$out/graph -o g -- python $OLDPWD/tools/graph_gaps.py run) > $out/graph.log 2>&1
python tools/gemm_ab.py --tun 18:0 --rounds 5 > $out/gemm_isolated.log 2>&1
python bench.py > $out/bench_line.json 2> $out/bench.err
grep -h metric $out/trace.log | cut -c1-300
ls $out/*
