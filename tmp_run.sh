mkdir -p gpurun_out/r05e
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r05e/pytest_gpu.log 2>&1
tail -4 gpurun_out/r05e/pytest_gpu.log
for rep in 1 2; do python bench.py --workload c4 --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('c4 ms_per_step %.4f kernel_ms %.4f frac %.3f %s' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['config'].get('stm_kernel')))
"; done
