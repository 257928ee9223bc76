#!/bin/bash
# Separate-process A/B of HIP runtime knobs on the graphed steps (same box, back to back; profiles/r05_experiments.txt sections 12-13).
#   bash tools/env_knobs_ab.sh            VisualBERT headline     bash tools/env_knobs_ab.sh vilbert     a --config model (parallel branches in its graph)
CFG=${1:-}
run() { env "$@" MMF_AMD_BENCH_NO_EAGER=1 python bench.py ${CFG:+--config $CFG} --steps 30 --warmup 5 --no-cpu-baseline --no-fp32 --no-other-configs 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-45s %.3f ms (event median %s)' % ('$*', d['ms_per_step'], d.get('ms_per_step_event_median')))"; }
for i in 1 2; do
run A=0
run HIP_FORCE_DEV_KERNARG=1
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run DEBUG_HIP_FORCE_GRAPH_QUEUES=1
run DEBUG_HIP_FORCE_GRAPH_QUEUES=4
run DEBUG_HIP_FORCE_GRAPH_QUEUES=4 DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
done
