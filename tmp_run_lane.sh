mkdir -p gpurun_out/r05d
for L in default build/liblto_lane_ilp.so; do
  if [ "$L" != default ]; then export LTO_HIP_LIB=$PWD/$L; else unset LTO_HIP_LIB; fi
  for rep in 1 2; do
  python bench.py --workload c4 --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('$L c4 ms_per_step %.4f kernel_ms %.4f frac %.3f %s' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['config'].get('stm_kernel')))
"
  done
done
unset LTO_HIP_LIB
for seg in 65536 131072; do python bench.py --workload c4 --segments $seg --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('c4 $seg segs ms_per_step %.4f %s' % (d['ms_per_step'], d['config'].get('stm_kernel')))
"; done
