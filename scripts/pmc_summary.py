#!/usr/bin/env python3
"""Average the rocprofv3 --pmc passes (csv) per kernel and derive the HBM traffic per launch.

usage: pmc_summary.py <dir with pass sub-directories> <out.json> [steps]
steps (optional): how many steps (frames / training steps, warm-up included) the profiled command ran - adds "_hbm_gb_per_step" =
sum over all kernels of HBM bytes per launch x launches / steps.
HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: both counters are in KB, and on gfx950 FETCH_SIZE counts
64-byte requests as 32 bytes (MI355X_MICROARCH.md, HBM / rocprofv3 section), hence the factor 2 on the fetch side.
"""
import collections
import csv
import glob
import json
import re
import sys

MODES = {"0": "full", "1": "forward", "2": "reverse", "3": "train"}


def kname(raw):
    k = raw.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").strip()
    m = re.match(r"k_field16<(\d)>", k)
    return f"k_field16<{MODES[m.group(1)]}>" if m else k


def main(root, out, steps=None):
    res = collections.defaultdict(dict)
    for f in sorted(glob.glob(root + "/**/*counter_collection.csv", recursive=True)):
        agg = collections.defaultdict(lambda: collections.defaultdict(float))
        n = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = kname(r["Kernel_Name"])
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            n[(k, r["Counter_Name"])] += 1
        for k in agg:
            for c, v in agg[k].items():
                res[k][c] = v / n[(k, c)]
                res[k]["launches_" + c] = n[(k, c)]
    keep = {}
    for k, v in res.items():
        if not k.startswith("k_"):
            continue
        d = {c: x for c, x in v.items() if not c.startswith("launches_")}
        d["launches_sampled"] = int(min(x for c, x in v.items() if c.startswith("launches_")))
        if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
            d["hbm_bytes_per_launch"] = (2.0 * d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024.0
        if "TCC_HIT" in d and d.get("TCC_REQ"):
            d["l2_hit_rate"] = d["TCC_HIT"] / d["TCC_REQ"]
        if "SQ_VALU_MFMA_BUSY_CYCLES" in d and d.get("GRBM_GUI_ACTIVE"):
            # MFMA busy cycles are summed over the 1024 SIMDs (= SQ_INSTS_MFMA x 32 cycles for the 32x32x16 f16 op);
            # GRBM_GUI_ACTIVE is summed over the 8 XCDs
            d["mfma_busy_frac"] = d["SQ_VALU_MFMA_BUSY_CYCLES"] / (d["GRBM_GUI_ACTIVE"] / 8.0 * 256 * 4)
        keep[k] = d
    if steps:
        tot = sum(v.get("hbm_bytes_per_launch", 0.0) * v["launches_sampled"] for v in keep.values())
        keep["_hbm_gb_per_step"] = tot / float(steps) / 1e9
        keep["_steps"] = int(steps)
    json.dump(keep, open(out, "w"), indent=1, sort_keys=True)
    keep = {k: v for k, v in keep.items() if isinstance(v, dict)}
    for k, v in sorted(keep.items(), key=lambda kv: -kv[1].get("hbm_bytes_per_launch", 0)):
        print(k, {c: "%.4g" % x for c, x in v.items()})


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else None)
