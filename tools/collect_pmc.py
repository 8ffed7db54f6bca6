#!/usr/bin/env python3
"""Turns two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs as MI355X_MICROARCH.md prescribes) of
`bench.py --no-cpu --no-lba` into profiles/pmc_traffic.json: HBM bytes per launch of every extractor kernel.

Units / corrections (MI355X_MICROARCH.md, HBM section): FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE
reports 1/2 of the bytes of a wide coalesced read, and other access widths are uncalibrated.  We therefore calibrate on
a kernel of this very pipeline whose byte count is known exactly: k_copy_level0 reads B*640*480 bytes and writes the same.
The file records raw counters, the calibration factors and the corrected per-launch bytes."""
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


def main():
    fetch_dir, write_dir, batch = sys.argv[1], sys.argv[2], int(sys.argv[3])
    fetch, nf = per_kernel(fetch_dir, "FETCH_SIZE")
    write, nw = per_kernel(write_dir, "WRITE_SIZE")
    known = batch * 640 * 480
    cal_f = known / (fetch["k_copy_level0"] * 1024.0)
    cal_w = known / (write["k_copy_level0"] * 1024.0)
    out = {"_note": "HBM bytes per launch (B=%d frames); raw counters in KiB; calibrated on k_copy_level0 (known %d B read, %d B written)" % (batch, known, known),
           "_calibration": {"fetch_factor": cal_f, "write_factor": cal_w}, "_raw_kib": {}}
    for k in sorted(set(fetch) | set(write)):
        if not k.startswith("k_"):
            continue
        f, w = fetch.get(k, 0.0), write.get(k, 0.0)
        out["_raw_kib"][k] = {"FETCH_SIZE": f, "WRITE_SIZE": w, "dispatches": nf.get(k, 0)}
        out[k[2:]] = f * 1024.0 * cal_f + w * 1024.0 * cal_w
    # k_resize is launched once per level: report the sum over the 7 levels as one "resize" stage
    if "resize" in out:
        out["resize"] = out["resize"] * 7
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    json.dump(out, open(path, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
