set -u
OUT=gpurun_out/r06z
mkdir -p $OUT
export TMPDIR=/tmp
python bench.py --workload c4 --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null > $OUT/bench_c4.json; tail -c 300 $OUT/bench_c4.json; echo
rm -rf $OUT/prof_c4 $OUT/c4
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_c4 -- python bench.py --workload c4 --steps 10 --warmup 2 --no-cpu-baseline > $OUT/prof_c4.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/c4/pmc_fetch -- python bench.py --workload c4 --steps 5 --warmup 2 --no-cpu-baseline > $OUT/pmc_fetch_c4.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/c4/pmc_write -- python bench.py --workload c4 --steps 5 --warmup 2 --no-cpu-baseline > $OUT/pmc_write_c4.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d $OUT/c4/pmc_sq -- python bench.py --workload c4 --steps 5 --warmup 2 --no-cpu-baseline > $OUT/pmc_sq_c4.log 2>&1
T0=$SECONDS; python bench.py --gpus 1 --steps 20 --warmup 5 --detail $OUT/bench_c2_verbose.json 2>$OUT/bench_c2_driver.err > $OUT/bench_c2_driver.json; echo "driver line: $((SECONDS - T0)) s wall, $(tail -1 $OUT/bench_c2_driver.json | wc -c) bytes"
python tools/probe_calibration.py 2>/dev/null > $OUT/probe_calibration.txt; cat $OUT/probe_calibration.txt
find $OUT -name "*.csv" | wc -l
