#!/bin/bash
# Copies the summaries of one tools/gpu_round.sh call (gpurun_out/<tag>/) into profiles/<tag>_* (tracked): the evidence DESIGN.md cites.
# usage: bash tools/collect_round.sh <tag>
set -e
T=${1:?tag}
S=gpurun_out/$T
P=profiles/$T
cp $S/bench.json ${P}_bench.json
cp $S/bench_detail_extra.json ${P}_bench_detail_extra.json
cp $S/kernel_stats.csv ${P}_bench_kernel_stats.csv
cp $S/kernel_stats_serial.csv ${P}_bench_kernel_stats_serial.csv
cp $S/pmc_traffic.json ${P}_pmc_traffic.json
cp $S/pmc_traffic.json profiles/pmc_traffic.json
cp $S/sq/valu_issue.json profiles/valu_issue.json
cp $S/sq_counters.log ${P}_sq_counters.txt
cp $S/valu_rates.txt ${P}_valu_rates.txt
cp $S/fast_mix.txt ${P}_fast_mix.txt
cp $(ls $S/lbaprof/*/*kernel_stats.csv | head -1) ${P}_lba_kernel_stats.csv
cp $(ls $S/libaprof/*/*kernel_stats.csv | head -1) ${P}_inertial_ba_kernel_stats.csv
cp $(ls $S/lbabatchprof/*/*kernel_stats.csv | head -1) ${P}_lba_batch32_kernel_stats.csv
cp $(ls $S/projprof/*/*kernel_stats.csv | head -1) ${P}_proj_kernel_stats.csv
cp $(ls $S/projprof_seq/*/*kernel_stats.csv | head -1) ${P}_proj_sequential_kernel_stats.csv
cat $S/latency_b1.log $S/pi_latency.log $S/latency_matcher.log 2>/dev/null | grep -v amdgpu.ids > ${P}_latencies.txt
cp $S/lba_pmc.log ${P}_lba_mfma_counters.txt
tail -3 $S/pytest_gpu.log > ${P}_pytest_gpu.txt
ls -la ${P}_*
