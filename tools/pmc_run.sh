#!/bin/bash
# usage: tools/pmc_run.sh <outdir> <counters...> -- <command...>
out=$1; shift
ctrs=()
while [ "$1" != "--" ]; do ctrs+=("$1"); shift; done
shift
export TMPDIR=/tmp
mkdir -p "$out"
rocprofv3 --kernel-trace --pmc "${ctrs[@]}" --output-format csv -d "$out" -o p -- "$@" > "$out/run.log" 2>&1
