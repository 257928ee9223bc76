#!/bin/bash
# Same-box A/B of bench.py under several settings of ONE environment variable:  bash tools/env_ab.sh VAR "v1 v2 ..." [rounds]
# Prints ms_per_step, samples/s, all-GEMM TFLOP/s and the attention / LayerNorm times per setting, interleaved over the rounds.
# (Separate processes: +-0.03 ms between identical runs.  For anything a tunable of the library can express prefer tools/step_ab.py: one process,
# one captured hipGraph per setting, +-0.01 ms.)
export TMPDIR=/tmp
var=$1; vals=$2; rounds=${3:-2}
for r in $(seq $rounds); do
  for v in $vals; do
    env $var=$v python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
l=[x for x in sys.stdin if x.startswith('{')]; d=json.loads(l[-1]); r=d['roofline']
a=r.get('attention',{}); ln=r.get('layernorm',{})
print('$var=$v', d['ms_per_step'], d['value'], r['all_gemm']['tflops'], 'attn', a.get('attention fwd',{}).get('ms'), a.get('attention bwd',{}).get('ms'), 'ln', ln.get('layernorm fwd',{}).get('ms'), ln.get('layernorm bwd',{}).get('ms'), 'fam', {k.split('| ')[1]: v['ms'] for k,v in r['all_gemm']['variants'].items() if v['ms']>0.3})"
  done
done
