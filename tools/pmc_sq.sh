#!/bin/bash
# SQ counters of every kernel of one bench step (GPU box): where do the waves spend their cycles?
# usage: bash tools/pmc_sq.sh <tag>
set -o pipefail
OUT=gpurun_out/${1:-sq}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_BUSY_CYCLES \
    --output-format csv -d $OUT/pmc_sq -- python bench.py --steps 2 --warmup 1 --no-cpu --no-lba > /dev/null 2> $OUT/pmc_sq.err || { echo "pmc sq failed"; tail -5 $OUT/pmc_sq.err; exit 1; }
python - <<PY
import csv, glob, collections
f = glob.glob("$OUT/pmc_sq/*/*counter_collection.csv")[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0][:40]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVE_CYCLES": cnt[k] += 1
for k, v in acc.items():
    wc = max(v["SQ_WAVE_CYCLES"], 1)
    print("%-40s n=%3d wave_cycles/launch %.3g  wait_any %.0f%%  wait_inst %.0f%%  active %.0f%%  valu %.0f%%  lds %.0f%%  valu_insts/launch %.3g" % (
        k, cnt[k], wc / max(cnt[k], 1), 100 * v["SQ_WAIT_ANY"] / wc, 100 * v["SQ_WAIT_INST_ANY"] / wc, 100 * v["SQ_ACTIVE_INST_ANY"] / wc,
        100 * v["SQ_ACTIVE_INST_VALU"] / wc, 100 * v["SQ_ACTIVE_INST_LDS"] / wc, v["SQ_INSTS_VALU"] / max(cnt[k], 1)))
PY
