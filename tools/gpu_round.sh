#!/bin/bash
# Standard GPU-box series: parity tests, bench, rocprof kernel trace + PMC passes.  Outputs -> gpurun_out/.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out/${1:-run}
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > "$OUT/pytest_gpu.log"
cat "$OUT/pytest_gpu.log" | tail -8
python bench.py 2>&1 | tail -3 > "$OUT/bench_c2.json"; cat "$OUT/bench_c2.json"
python bench.py --workload hbm --ndim 12 --no-cpu-baseline 2>&1 | tail -1 > "$OUT/bench_hbm.json"; cat "$OUT/bench_hbm.json"
python bench.py --workload c3 --cpu-seconds 5 2>&1 | tail -1 > "$OUT/bench_c3.json"; cat "$OUT/bench_c3.json"
python bench.py --ndim 12 --cpu-seconds 5 2>&1 | tail -1 > "$OUT/bench_c2_ndim12.json"; cat "$OUT/bench_c2_ndim12.json"
python bench.py --ndim 12 --method dop853 --no-cpu-baseline --steps 50 2>&1 | tail -1 > "$OUT/bench_c2_dop853.json"; cat "$OUT/bench_c2_dop853.json"
python bench.py --ndim 12 --method rkf78 --no-cpu-baseline 2>&1 | tail -1 > "$OUT/bench_c2_rkf78x4.json"; cat "$OUT/bench_c2_rkf78x4.json"
python bench.py --kernel 1 --no-cpu-baseline 2>&1 | tail -1 > "$OUT/bench_c2_per_lane.json"; cat "$OUT/bench_c2_per_lane.json"
python bench.py --workload c4 --no-cpu-baseline --steps 20 --warmup 3 2>&1 | tail -1 > "$OUT/bench_c4.json"; cat "$OUT/bench_c4.json"
python bench.py --workload c5 --no-cpu-baseline --steps 20 --warmup 3 2>&1 | tail -1 > "$OUT/bench_c5.json"; cat "$OUT/bench_c5.json"
python bench.py --workload c5_stm --no-cpu-baseline --steps 10 --warmup 2 2>&1 | tail -1 > "$OUT/bench_c5_stm.json"; cat "$OUT/bench_c5_stm.json"
# kernel trace + stats of the contract command
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_c2" -- python bench.py --steps 50 --warmup 5 --no-cpu-baseline > "$OUT/prof_c2.log" 2>&1
# PMC passes, one counter group per run (FETCH_SIZE and WRITE_SIZE do not fit one pass)
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline > "$OUT/pmc_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline > "$OUT/pmc_write.log" 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d "$OUT/pmc_sq" -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline > "$OUT/pmc_sq.log" 2>&1
find "$OUT" -name "*.csv" | head -30
ls -la "$OUT"
