#!/bin/bash
# Standard GPU-box series: parity tests, bench lines, rocprof kernel traces + PMC passes.  Outputs -> gpurun_out/<tag>/.
# usage: tools/gpu_round.sh <tag> [bench|prof|pmc|all]   (one gpurun call holds 20 minutes: bench + prof in one call, pmc in another)
set -u
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out/${1:-run}
PART=${2:-all}
mkdir -p "$OUT"
export TMPDIR=/tmp
last() { tail -1 "$1" | cut -c1-400; }
if [ "$PART" = all ] || [ "$PART" = bench ]; then
# progress goes straight into the log (a pipe into tail would hold it back until the end: the GPU box kills a run that stays silent for 7 minutes)
timeout -k 10 900 python -m pytest tests -m gpu -x -q > "$OUT/pytest_gpu.log" 2>&1
tail -3 "$OUT/pytest_gpu.log"
T0=$SECONDS; python bench.py --gpus 1 --steps 20 --warmup 5 --detail "$OUT/bench_c2_verbose.json" 2>"$OUT/bench_c2_driver.err" > "$OUT/bench_c2_driver.json"; echo "driver line: $((SECONDS - T0)) s wall, $(tail -1 "$OUT/bench_c2_driver.json" | wc -c) bytes"; last "$OUT/bench_c2_driver.json"
python bench.py --no-cpu-baseline --no-configs 2>/dev/null > "$OUT/bench_c2_200.json"; last "$OUT/bench_c2_200.json"
python bench.py --workload hbm --ndim 12 --segments 1048576 --no-cpu-baseline --live-traffic on --steps 20 --warmup 3 2>/dev/null > "$OUT/bench_hbm.json"; last "$OUT/bench_hbm.json"
python bench.py --workload hbm --ndim 14 --segments 1048576 --no-cpu-baseline --live-traffic on --steps 20 --warmup 3 2>/dev/null > "$OUT/bench_hbm14.json"; last "$OUT/bench_hbm14.json"   # round 6: the 14-dim form of the whole-segment one-step kernel
python bench.py --segments 8192 --no-cpu-baseline 2>/dev/null > "$OUT/bench_c2_8192.json"; last "$OUT/bench_c2_8192.json"    # 32-segment pipeline (AUTO above one round)
python bench.py --workload c3 --cpu-seconds 5 2>/dev/null > "$OUT/bench_c3.json"; last "$OUT/bench_c3.json"
python bench.py --ndim 12 --method dop853 --no-cpu-baseline --steps 50 2>/dev/null > "$OUT/bench_c2_dop853.json"; last "$OUT/bench_c2_dop853.json"
python bench.py --ndim 14 --method dop853 --no-cpu-baseline --steps 50 2>/dev/null > "$OUT/bench_c2_ndim14_dop853.json"; last "$OUT/bench_c2_ndim14_dop853.json"   # round 6: the 7 + 7 two-lane form
python bench.py --ndim 12 --method rkf78 --no-cpu-baseline 2>/dev/null > "$OUT/bench_c2_rkf78x4.json"; last "$OUT/bench_c2_rkf78x4.json"
python bench.py --workload c4 --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null > "$OUT/bench_c4.json"; last "$OUT/bench_c4.json"
python bench.py --workload c5 --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null > "$OUT/bench_c5.json"; last "$OUT/bench_c5.json"
python bench.py --workload c5_stm --no-cpu-baseline --steps 10 --warmup 2 2>/dev/null > "$OUT/bench_c5_stm.json"; last "$OUT/bench_c5_stm.json"
python bench.py --workload newton --cpu-seconds 6 2>/dev/null > "$OUT/bench_newton.json"; last "$OUT/bench_newton.json"
python tools/probe_calibration.py 2>/dev/null > "$OUT/probe_calibration.txt"; cat "$OUT/probe_calibration.txt"
python tools/kernel_resources.py > "$OUT/kernel_resources.txt"; head -1 "$OUT/kernel_resources.txt"
fi
if [ "$PART" = all ] || [ "$PART" = prof ]; then
# kernel traces + stats.  The device ramps its clocks over the first ~300 contract launches (91 -> 79 us per launch): the c2
# and c3 traces time enough steps (4 000 / 2 000) for the average over ALL launches of the trace to be the ramped duration.
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_c2" -- python bench.py --steps 4000 --warmup 5 --no-cpu-baseline --no-configs > "$OUT/prof_c2.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_c2_8192" -- python bench.py --segments 8192 --steps 2000 --warmup 5 --no-cpu-baseline > "$OUT/prof_c2_8192.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_c3" -- python bench.py --workload c3 --steps 2000 --warmup 5 --no-cpu-baseline > "$OUT/prof_c3.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_c4" -- python bench.py --workload c4 --steps 10 --warmup 2 --no-cpu-baseline > "$OUT/prof_c4.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_c2_dop853" -- python bench.py --ndim 12 --method dop853 --steps 30 --warmup 5 --no-cpu-baseline > "$OUT/prof_c2_dop853.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_c2_ndim14_dop853" -- python bench.py --ndim 14 --method dop853 --steps 30 --warmup 5 --no-cpu-baseline > "$OUT/prof_c2_ndim14_dop853.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_c5_stm" -- python bench.py --workload c5_stm --steps 5 --warmup 2 --no-cpu-baseline > "$OUT/prof_c5_stm.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_c5" -- python bench.py --workload c5 --steps 20 --warmup 3 --no-cpu-baseline > "$OUT/prof_c5.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_newton" -- python bench.py --workload newton --pmc-child --steps 100 --warmup 5 > "$OUT/prof_newton.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_hbm" -- python bench.py --workload hbm --ndim 12 --segments 1048576 --steps 20 --warmup 3 --no-cpu-baseline > "$OUT/prof_hbm.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_hbm14" -- python bench.py --workload hbm --ndim 14 --segments 1048576 --steps 20 --warmup 3 --no-cpu-baseline > "$OUT/prof_hbm14.log" 2>&1
fi
if [ "$PART" = all ] || [ "$PART" = pmc ]; then
# PMC passes for EVERY workload that has a roofline row, one counter group per run (FETCH_SIZE and WRITE_SIZE do not fit one
# pass; the program itself right after `--`).  Layout: $OUT/<key>/pmc_{fetch,write,sq}; key = bench.py's pmc_key().
pmc_passes() {   # pmc_passes <key> <bench.py arguments...>
  local KEY=$1; shift
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/$KEY/pmc_fetch" -- python bench.py "$@" --steps 5 --warmup 2 --no-cpu-baseline > "$OUT/pmc_fetch_$KEY.log" 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/$KEY/pmc_write" -- python bench.py "$@" --steps 5 --warmup 2 --no-cpu-baseline > "$OUT/pmc_write_$KEY.log" 2>&1
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d "$OUT/$KEY/pmc_sq" -- python bench.py "$@" --steps 5 --warmup 2 --no-cpu-baseline > "$OUT/pmc_sq_$KEY.log" 2>&1
  echo "pmc $KEY done"
}
pmc_passes c2 --no-configs                                   # also holds the 12-dim leg (pmc_c2_ndim12) and the reference-integrator leg
pmc_passes c2_8192 --segments 8192
pmc_passes c3 --workload c3
pmc_passes c4 --workload c4
pmc_passes c5 --workload c5
pmc_passes c5_stm --workload c5_stm
pmc_passes c2_ndim12_dop853 --ndim 12 --method dop853
pmc_passes c2_dop853 --ndim 14 --method dop853
pmc_passes hbm_ndim12 --workload hbm --ndim 12 --segments 1048576
pmc_passes hbm --workload hbm --ndim 14 --segments 1048576
pmc_passes newton --workload newton --pmc-child      # the Newton iteration's kernels: k_bvp_chunk / _tail / _backchunk, the sweeps, the norms
fi
find "$OUT" -name "*.csv" | wc -l
# condense into profiles/ (tag = $1):
#   python tools/summarize_profile.py $OUT <tag> c2 "k_indirect_pipe8<14"
#   python tools/summarize_profile.py $OUT <tag> c2_ndim12 "k_indirect_pipe8<12" c2
#   python tools/summarize_profile.py $OUT <tag> c3 "k_direct_jacobian_pipe<6"
#   (tools/condense_round.sh <run> <tag> runs all of these)
#   python tools/summarize_profile.py $OUT <tag> c4 "k_indirect_lane"
#   python tools/summarize_profile.py $OUT <tag> c5 "k_indirect_defect4"
#   python tools/summarize_profile.py $OUT <tag> c5_stm "k_indirect_coop2"
#   python tools/summarize_profile.py $OUT <tag> c2_ndim12_dop853 "k_indirect_coop2" c2_dop853
#   python tools/summarize_profile.py $OUT <tag> hbm_ndim12 "k_indirect_stream"
#   python tools/summarize_profile.py $OUT <tag> newton_bvp_chunk "k_bvp_chunk<12, true" newton      (likewise _tail, _backchunk, _chunk_rhs)
