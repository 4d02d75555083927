#!/bin/bash
# usage: ab.sh lib...   (empty string = default)
mkdir -p gpurun_out/r04p
for L in "$@"; do
  if [ "$L" != default ]; then export LTO_HIP_LIB=$PWD/$L; else unset LTO_HIP_LIB; fi
  echo "LIB=$L"
  for W in "--workload c3" "--workload c4 --steps 10 --warmup 3" "--workload c5 --steps 20 --warmup 3" "--workload c5_stm --steps 10 --warmup 2" "--workload hbm --ndim 12 --segments 1048576 --steps 20 --warmup 3" "--ndim 12 --method rkf78" "--ndim 14 --method dop853 --steps 50"; do
    python bench.py $W --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('  %-70s ms_per_step %.5f' % ('$W', d['ms_per_step']))
"
  done
done
