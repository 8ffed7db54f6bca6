#!/bin/bash
# SQ counters of every kernel of one bench step (GPU box): where do the waves spend their cycles?
# usage: bash tools/pmc_sq.sh <tag>
set -o pipefail
OUT=gpurun_out/${1:-sq}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_BUSY_CYCLES \
    --output-format csv -d $OUT/pmc_sq -- python bench.py --steps 2 --warmup 1 --no-cpu --no-lba > /dev/null 2> $OUT/pmc_sq.err || { echo "pmc sq failed"; tail -5 $OUT/pmc_sq.err; exit 1; }
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_INST_LEVEL_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM_RD \
    --output-format csv -d $OUT/pmc_sq2 -- python bench.py --steps 2 --warmup 1 --no-cpu --no-lba > /dev/null 2> $OUT/pmc_sq2.err || echo "second counter pass (LDS / memory instructions) failed"
python - <<PY
import csv, glob, collections, json, sys
sys.path.insert(0, "tools")
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
# second pass: LDS / scalar / memory instruction counts per launch
f2 = glob.glob("$OUT/pmc_sq2/*/*counter_collection.csv")
acc2 = collections.defaultdict(lambda: collections.defaultdict(float)); cnt2 = collections.Counter()
if f2:
    for r in csv.DictReader(open(f2[0])):
        k = r["Kernel_Name"].split("(")[0][:40]
        acc2[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_INSTS_LDS": cnt2[k] += 1
    for k, v in acc2.items():
        n = max(cnt2[k], 1)
        print("%-40s per launch: lds_insts %.3g  salu %.3g  vmem %.3g  smem %.3g  lds_bank_conflict_cycles %.3g  lds_idx_active %.3g" % (
            k, v["SQ_INSTS_LDS"] / n, v["SQ_INSTS_SALU"] / n, v["SQ_INSTS_VMEM"] / n, v["SQ_INSTS_SMEM"] / n, v["SQ_LDS_BANK_CONFLICT"] / n, v["SQ_LDS_IDX_ACTIVE"] / n))
# profiles/valu_issue.json: wave-instructions per STEP of every kernel, stamped with the kernel sources they were counted on
# (bench.py turns them into roofline.valu_issue_frac with the per-instruction issue cost of tools/probes/valu_rates.hip)
from collect_pmc import kernel_sources_stamp
steps = max(cnt.get("orbx::k_orient_desc", 1), 1)
out = {"_note": "SQ_INSTS_VALU / SQ_INSTS_LDS wave-instructions per bench step (B=256 frames), tools/pmc_sq.sh", "_kernel_sources_sha": kernel_sources_stamp(),
       "_ns_per_wave_instruction_and_simd": 1.80, "_ns_from": "profiles/r03_a_valu_rates.txt / r03_a_fast_mix.txt: v_perm / v_pk_* / v_min3 / v_max3 class and the stage-1 mix at 4-8 waves per SIMD",
       "_simds": 1024}
for k, v in acc.items():
    name = k.split("::")[-1].split("<")[0]
    out[name] = {"valu": v["SQ_INSTS_VALU"] / steps, "lds": acc2.get(k, {}).get("SQ_INSTS_LDS", 0.0) / max(steps, 1) if k in acc2 else None}
json.dump(out, open("profiles/valu_issue.json", "w"), indent=1)
PY
cp profiles/valu_issue.json $OUT/valu_issue.json 2>/dev/null
