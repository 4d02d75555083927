#!/bin/bash
# reference-integrator sweeps (k_indirect_coop2, k_indirect_defect4) with several builds: tools/ab_dop853.sh default build/liblto_x.so ...
for L in "$@"; do
  if [ "$L" != default ]; then export LTO_HIP_LIB=$PWD/$L; else unset LTO_HIP_LIB; fi
  python bench.py --ndim 12 --method dop853 --no-cpu-baseline --steps 100 --warmup 20 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('%-32s STM sweep %.2f us' % ('$L', d['ms_per_step']*1e3), end='')
"
  python bench.py --workload c2_defect --ndim 12 --method dop853 --no-cpu-baseline --steps 100 --warmup 20 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('   defect sweep %.2f us' % (d['ms_per_step']*1e3))
"
done
