#!/bin/bash
# Everything measured on the GPU box in one gpurun call: parity tests, bench, rocprofv3 kernel stats, PMC traffic.
# usage (from the repo root on the box):  bash tools/gpu_round.sh <tag>
set -o pipefail
TAG=${1:-rXX}
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout -k 10 420 python -m pytest tests -m gpu -x -q < /dev/null > $OUT/pytest_gpu.log 2>&1; rc=$?
tail -3 $OUT/pytest_gpu.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "pytest timed out: stopping"; exit 1; fi
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python bench.py --steps 5 --warmup 1 --no-cpu > $OUT/bench_prof.json 2> $OUT/bench_prof.err || { echo "rocprof stats failed"; exit 1; }
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python bench.py --steps 3 --warmup 1 --no-cpu --no-lba > /dev/null 2> $OUT/pmc_fetch.err || { echo "pmc fetch failed"; exit 1; }
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python bench.py --steps 3 --warmup 1 --no-cpu --no-lba > /dev/null 2> $OUT/pmc_write.err || { echo "pmc write failed"; exit 1; }
python tools/collect_pmc.py $OUT/pmc_fetch $OUT/pmc_write 256 > $OUT/pmc_traffic.log && cp profiles/pmc_traffic.json $OUT/pmc_traffic.json
# SQ counters of every kernel of a step (+ LDS / memory instruction counts) -> profiles/valu_issue.json (bench.py: roofline.valu_issue_frac)
bash tools/pmc_sq.sh $TAG/sq > $OUT/sq_counters.log 2>&1 || echo "sq counter pass failed"
# the bench line LAST of the three: roofline.traffic / valu_issue_frac come from the counter passes above (stamped with these kernel sources)
timeout -k 10 400 python bench.py > $OUT/bench.json 2> $OUT/bench.err || { echo "bench failed"; tail -5 $OUT/bench.err; exit 1; }
cp gpurun_out/bench_detail.json $OUT/bench_detail.json 2>/dev/null
cp $(ls $OUT/prof/*/*kernel_stats.csv | head -1) $OUT/kernel_stats.csv
# the same command with the serial schedule (ORBX_SERIAL=1: one launch per stage on one stream): per-kernel averages that can be
# compared with the HIP-event stage times of the bench line (in the production schedule FAST and the octree are several
# launches that run beside other kernels)
ORBX_SERIAL=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_serial -- python bench.py --steps 5 --warmup 1 --no-cpu --no-lba > $OUT/bench_prof_serial.json 2> $OUT/bench_prof_serial.err || echo "serial rocprof stats failed"
cp $(ls $OUT/prof_serial/*/*kernel_stats.csv | head -1) $OUT/kernel_stats_serial.csv 2>/dev/null
timeout -k 10 200 python bench.py --streams 4 --no-cpu --no-lba > $OUT/bench_streams4.json 2> $OUT/bench_streams4.err || echo "streams-4 bench failed"
# per-kernel times of the two BA solvers alone (profiles/<tag>_lba_kernel_stats.csv, <tag>_inertial_ba_kernel_stats.csv)
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/lbaprof -- python tools/lba_prof.py 5 > $OUT/lbaprof.log 2>&1 || echo "lba profile failed"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/libaprof -- python tools/liba_prof.py 5 > $OUT/libaprof.log 2>&1 || echo "inertial ba profile failed"
# single-call latencies (what Tracking / LocalMapping make): extract for small batches, per-frame inertial optimisation, LocalInertialBA
timeout -k 10 200 python tools/latency_b1.py > $OUT/latency_b1.log 2>&1 || echo "latency_b1 failed"
timeout -k 10 200 python tools/pi_latency.py > $OUT/pi_latency.log 2>&1 || echo "pi_latency failed"
timeout -k 10 200 python tools/latency_matcher.py > $OUT/latency_matcher.log 2>&1 || echo "latency_matcher failed"
bash tools/pmc_lba.sh $TAG/lba_pmc > $OUT/lba_pmc.log 2>&1 || echo "lba mfma counter pass failed"
# the probes behind the issue-cost figures
(/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 tools/probes/valu_rates.hip -o /tmp/valu_rates && timeout -k 10 120 /tmp/valu_rates > $OUT/valu_rates.txt 2>&1) || echo "valu_rates probe failed"
(/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 tools/probes/fast_mix.hip -o /tmp/fast_mix && timeout -k 10 120 /tmp/fast_mix > $OUT/fast_mix.txt 2>&1) || echo "fast_mix probe failed"
# many windows per launch, and the block-parallel projection search (256 frames x 900 points)
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/lbabatchprof -- python tools/lba_batch_prof.py 32 3 > $OUT/lba_batch_prof.log 2>&1 < /dev/null || echo "lba batch profile failed"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/projprof -- python tools/proj_batch_prof.py 256 3 > $OUT/proj_prof.log 2>&1 < /dev/null || echo "projection profile failed"
ORBM_PROJ_SEQUENTIAL=1 timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/projprof_seq -- python tools/proj_batch_prof.py 256 3 > $OUT/proj_prof_seq.log 2>&1 < /dev/null || echo "sequential projection profile failed"
timeout -k 10 400 python bench.py --extra > $OUT/bench_extra.json 2> $OUT/bench_extra.err < /dev/null && cp gpurun_out/bench_detail.json $OUT/bench_detail_extra.json || echo "bench --extra failed"
python - <<PY
import json
d = json.load(open("$OUT/bench.json"))
print("frames/s %.0f  x%.0f vs cpu  roofline %s frac %.4f pipeline frac %.4f" % (d["value"], d.get("speedup_vs_cpu_1core", 0), d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"].get("pipeline_frac", 0)))
print({k: round(v, 3) for k, v in d["roofline"]["stage_ms"].items()})
l = d["lba"]
print("lba iters/s %.0f  x%.1f vs cpu  ms/iteration %.3f ms/trial %.3f  roofline %s" % (l["value"], l.get("speedup_vs_cpu_1core", 0), l["ms_per_iteration"], l["ms_per_trial"], l["roofline"]))
t = json.load(open("$OUT/pmc_traffic.json"))
print("traffic MB/launch", {k: round(v / 1e6, 1) for k, v in t.items() if not k.startswith("_")})
print("line length", len(open("$OUT/bench.json").read()))
for f in ("latency_b1.log", "pi_latency.log"):
    try:
        print("".join(l for l in open("$OUT/" + f) if ("ms per" in l or "liba_solve" in l)).rstrip())
    except OSError:
        pass
PY
