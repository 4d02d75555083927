#!/bin/bash
# Condense a tools/gpu_round.sh series (gpurun_out/<run>/) into the committed evidence under profiles/ with prefix <tag>:
# bench lines, the GPU suite's log, kernel-trace summaries and the PMC summaries of every roofline row.
# usage: tools/condense_round.sh <run> <tag>      e.g. tools/condense_round.sh r04z r04z
set -eu
cd "$(dirname "$0")/.."
RUN=gpurun_out/$1
TAG=$2
for f in "$RUN"/bench_*.json; do cp "$f" "profiles/${TAG}_$(basename "$f")"; done
[ -f "$RUN/kernel_resources.txt" ] && cp "$RUN/kernel_resources.txt" "profiles/${TAG}_kernel_resources.txt"
cp "$RUN/pytest_gpu.log" "profiles/${TAG}_pytest_gpu.log"
[ -f "$RUN/probe_calibration.txt" ] && cp "$RUN/probe_calibration.txt" "profiles/${TAG}_probe_calibration.txt"
S="python tools/summarize_profile.py $RUN $TAG"
q() { "$@" > /dev/null; }
q $S c2 "k_indirect_pipe8<14"
q $S c2_ndim12 "k_indirect_pipe8<12" c2
q $S c2_8192 "k_indirect_pipe32<14"
q $S c3 "k_direct_jacobian_pipe<6"
q $S c4 "k_indirect_lane"
q $S c5 "k_indirect_defect4<1, 12>"
q $S c5_stm "k_indirect_coop2<" - "k_node_records"
q $S c2_ndim12_dop853 "k_indirect_coop2<" c2_dop853
q $S c2_dop853 "k_indirect_coop2_14" c2_ndim14_dop853
q $S hbm_ndim12 "k_indirect_stream<12" hbm
q $S hbm "k_indirect_stream<14" hbm14
q $S newton_bvp_chunk_first "k_bvp_chunk<12, true" newton
q $S newton_bvp_chunk "k_bvp_chunk<12, false" newton
q $S newton_bvp_tail "k_bvp_tail<12" newton
q $S newton_bvp_backchunk "k_bvp_backchunk<12" newton
q $S newton_bvp_chunk_rhs "k_bvp_chunk_rhs<12, true" newton
q $S newton_stm_sweep "k_indirect_coop2<" newton
q $S newton_defect_sweep "k_indirect_defect4" newton
ls profiles/${TAG}_* | wc -l
