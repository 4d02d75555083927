#!/bin/bash
# usage: tools/ab_r6.sh out lib... ("default" = the shipped library); workloads in $AB_WORKLOADS (one per line), two passes
out=$1; shift
mkdir -p "$(dirname "$out")"
: > "$out"
for pass in 1 2; do
for L in "$@"; do
  if [ "$L" != default ]; then export LTO_HIP_LIB=$PWD/$L; else unset LTO_HIP_LIB; fi
  echo "LIB=$L pass $pass" >> "$out"
  while IFS= read -r W; do
    [ -z "$W" ] && continue
    python bench.py $W --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1])
print('  %-70s ms_per_step %.5f' % ('''$W''', d['ms_per_step']))
" >> "$out"
  done <<< "$AB_WORKLOADS"
done
done
cat "$out"
