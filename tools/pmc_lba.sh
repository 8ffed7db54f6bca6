#!/bin/bash
# Matrix-pipe counters of the LocalBA kernels (GPU box): how much of k_chol_step's time the f64 MFMA pipe is busy.
# usage: bash tools/pmc_lba.sh <tag>      (counters in their own pass: no trace domains besides the kernel dispatch records)
set -o pipefail
OUT=gpurun_out/${1:-lba_pmc}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z0-9_]*MFMA[A-Z0-9_]*" | sort -u > $OUT/mfma_counters_available.txt
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU \
    --output-format csv -d $OUT/pmc -- python tools/lba_prof.py 3 > $OUT/pmc.log 2> $OUT/pmc.err || { echo "pmc pass failed"; tail -5 $OUT/pmc.err; exit 1; }
python - <<PY
import csv, glob, collections
f = glob.glob("$OUT/pmc/*/*counter_collection.csv")[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")[:34]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVE_CYCLES": cnt[k] += 1
lines = []
for k, v in sorted(acc.items()):
    if not k.startswith("lba::"): continue
    n = max(cnt[k], 1)
    # SQ_VALU_MFMA_BUSY_CYCLES counts cycles, SQ_BUSY_CYCLES quad-cycles per SE-aggregated SQ (MI355X_MICROARCH.md): report the raw ratio too
    lines.append("%-34s n=%4d  mfma_mops_f64/launch %10.0f  insts_mfma/launch %8.0f  insts_valu/launch %10.0f  mfma_busy_cycles/launch %10.0f  sq_busy_cycles/launch %10.0f  wave_cycles/launch %10.0f" % (
        k, n, v["SQ_INSTS_VALU_MFMA_MOPS_F64"] / n, v["SQ_INSTS_MFMA"] / n, v["SQ_INSTS_VALU"] / n, v["SQ_VALU_MFMA_BUSY_CYCLES"] / n, v["SQ_BUSY_CYCLES"] / n, v["SQ_WAVE_CYCLES"] / n))
open("$OUT/lba_mfma_counters.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
