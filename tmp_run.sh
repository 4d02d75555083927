for L in default build/liblto_stream_nt.so build/liblto_stream_wpe1.so build/liblto_stream_ntwpe1.so; do
  if [ "$L" != default ]; then export LTO_HIP_LIB=$PWD/$L; else unset LTO_HIP_LIB; fi
  for rep in 1 2; do
  python bench.py --workload hbm --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('$L hbm ms_per_step %.4f kernel_ms %.4f frac %.3f' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac']))
"
  done
done
