#!/bin/bash
# contract sweep (14-dim and 12-dim legs) with several builds of the library: tools/ab_c2.sh default build/liblto_x.so ...
for L in "$@"; do
  if [ "$L" != default ]; then export LTO_HIP_LIB=$PWD/$L; else unset LTO_HIP_LIB; fi
  python bench.py --no-cpu-baseline --steps 200 --warmup 20 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('%-32s 14-dim %.2f us  12-dim %.2f us  parity %s' % ('$L', d['ms_per_step']*1e3, d['reference_system_12dim']['ms_per_step']*1e3, json.dumps(d.get('parity_vs_oracle'))[:160]))
"
done
