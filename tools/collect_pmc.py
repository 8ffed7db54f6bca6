#!/usr/bin/env python3
"""Turns two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs as MI355X_MICROARCH.md prescribes) of
`bench.py --no-cpu --no-lba` into profiles/pmc_traffic.json: HBM bytes per step (= per launch, summed over the launches of kernels that run more than once per step) of every extractor kernel.

Units / corrections (MI355X_MICROARCH.md, HBM section): FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE
reports exactly 1/2 of the bytes of a wide coalesced read (factor 2), WRITE_SIZE reads the bytes exactly for wide stores.
Round 1 calibrated both on k_copy_level0 of this pipeline (known byte count): fetch factor 1.9993, write factor 1.0000
(profiles/r01_n_pmc_traffic.json); that kernel no longer runs when the input is read in place, so the guide's factors are
applied directly.  The file records raw counters, the factors and the corrected per-launch bytes."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def per_kernel(dirname, counter):
    files = glob.glob(os.path.join(dirname, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        raise SystemExit("no counter_collection.csv under " + dirname)
    acc = defaultdict(list)
    for f in files:
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != counter:
                continue
            name = r["Kernel_Name"].split("(")[0].split("::")[-1]
            acc[name].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}, {k: len(v) for k, v in acc.items()}


def kernel_sources_sha(root=ROOT, group="orbx"):
    """sha256 over one group of kernel sources (`orbx`: the extractor's kernels, `orbm`: the matcher's): the stamp that ties a counter
    pass to the kernels it measured (bench.py prints `traffic: null` when the stamp of profiles/pmc_traffic.json is not the running
    sources').  Per group, so that an edit of the matcher does not void the extractor's counters and vice versa."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(root, "orb_slam3-1_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.startswith(group + "_") and name.endswith((".hip", ".inc", ".h")):
            h.update(name.encode()); h.update(open(os.path.join(d, name), "rb").read())
    return h.hexdigest()[:16]


def kernel_sources_stamp(root=ROOT):
    return {g: kernel_sources_sha(root, g) for g in ("orbx", "orbm")}


def stamp_matches(stamp, kernel, root=ROOT):
    """is a profile stamped `stamp` valid for `kernel` (k_bow* lives in orbm_*, every other kernel of the step in orbx_*)?"""
    g = "orbm" if kernel.replace("k_", "").startswith("bow") else "orbx"
    return isinstance(stamp, dict) and stamp.get(g) == kernel_sources_sha(root, g)


def main():
    fetch_dir, write_dir, batch = sys.argv[1], sys.argv[2], int(sys.argv[3])
    fetch, nf = per_kernel(fetch_dir, "FETCH_SIZE")
    write, nw = per_kernel(write_dir, "WRITE_SIZE")
    cal_f, cal_w = 2.0, 1.0
    if "k_copy_level0" in fetch and "k_copy_level0" in write:      # still launched for unaligned inputs: a known byte count
        known = batch * 640 * 480
        cal_f = known / (fetch["k_copy_level0"] * 1024.0)
        cal_w = known / (write["k_copy_level0"] * 1024.0)
    out = {"_note": "HBM bytes per launch (B=%d frames); raw counters in KiB" % batch,
           "_source": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE passes of `bench.py --no-cpu --no-lba` (tools/gpu_round.sh)",
           "_calibration": {"fetch_factor": cal_f, "write_factor": cal_w}, "_raw_kib": {},
           "_kernel_sources_sha": kernel_sources_stamp()}
    for k in sorted(set(fetch) | set(write)):
        if not k.startswith("k_"):
            continue
        f, w = fetch.get(k, 0.0), write.get(k, 0.0)
        out["_raw_kib"][k] = {"FETCH_SIZE": f, "WRITE_SIZE": w, "dispatches": nf.get(k, 0)}
        out[k[2:]] = f * 1024.0 * cal_f + w * 1024.0 * cal_w
    # kernels launched several times per step (k_resize once per lower level, k_octree and k_fast_strips two or three times in the
    # large-batch schedule): report bytes per STEP
    steps = max(nf.get("k_orient_desc", nf.get("k_blur", 1)), 1)      # kernels that are one launch per step in every schedule
    for k in list(out):
        if k.startswith("_"):
            continue
        per_step = nf.get("k_" + k, steps) / float(steps)
        if per_step > 1.01:
            out[k] = out[k] * per_step
            out["_raw_kib"]["k_" + k]["launches_per_step"] = per_step
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    json.dump(out, open(path, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
