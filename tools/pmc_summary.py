"""Summarise rocprofv3 --pmc passes of tools/bench_conv.py into the JSON bench.py reads for `roofline.traffic`.

    python tools/pmc_summary.py [--src <kernel source file> ...] [--grid-max N] <kernel substring> <out.json> <counter_collection.csv> [...]

--src (repeatable; paths relative to the repository root): the sources the measured kernel is compiled from.  Their SHA-256 goes into
the summary as "kernel_sources"; bench.py reports `traffic: null, traffic_source: "stale: ..."` once one of them no longer matches
(a committed counter value must not outlive the kernel it was measured on).

Each CSV is one `rocprofv3 --pmc ... --output-format csv` pass (counters are collected in separate passes, as
MI355X_MICROARCH.md prescribes).  Values are averaged over the dispatches of the named kernel.  HBM bytes per launch =
(2 * FETCH_SIZE + WRITE_SIZE) KB * 1024: on gfx950 FETCH_SIZE counts 64 B per 128-byte request (same guide).
"""
import collections
import csv
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def source_hashes(paths):
    return {p: hashlib.sha256(open(os.path.join(ROOT, p), "rb").read()).hexdigest() for p in paths}


def main():
    argv = sys.argv[1:]
    srcs, grid_max = [], None
    while argv and argv[0] in ("--src", "--grid-max"):
        if argv[0] == "--src":
            srcs.append(argv[1])
        else:
            grid_max = int(argv[1])   # only dispatches of at most this many work-items (one kernel, launches of several sizes in one run)
        argv = argv[2:]
    kernel, out = argv[0], argv[1]
    vals = collections.defaultdict(list)
    grids = collections.Counter()   # work-items per dispatch of the named kernel (all of them, before --grid-max), from the first pass
    durations = []   # of the dispatches as they ran UNDER the counters (serialised, other clocks than in the step): for GRBM cycles / time
    for path in argv[2:]:
        per_dispatch = collections.defaultdict(dict)
        seen = set()
        count_grids = not grids
        seen_g = set()
        for r in csv.DictReader(open(path)):
            if count_grids and kernel in r["Kernel_Name"] and r["Dispatch_Id"] not in seen_g:
                seen_g.add(r["Dispatch_Id"])
                grids[int(r["Grid_Size"])] += 1
            if kernel in r["Kernel_Name"] and (grid_max is None or int(r["Grid_Size"]) <= grid_max):
                per_dispatch[r["Dispatch_Id"]].setdefault(r["Counter_Name"], 0.0)
                per_dispatch[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
                if r["Counter_Name"] == "GRBM_GUI_ACTIVE" and r["Dispatch_Id"] not in seen and r.get("Start_Timestamp") and r.get("End_Timestamp"):
                    seen.add(r["Dispatch_Id"])
                    durations.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        for d in per_dispatch.values():
            for k, v in d.items():
                vals[k].append(v)
    res = {"kernel": kernel, "dispatches_averaged": {k: len(v) for k, v in vals.items()}}
    for k, v in vals.items():
        res[k] = sum(v) / len(v)
    if "FETCH_SIZE" in res and "WRITE_SIZE" in res:
        res["hbm_bytes_per_launch"] = int((2 * res["FETCH_SIZE"] + res["WRITE_SIZE"]) * 1024)
        res["note"] = "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 counts 64 B per 128-B request); (2*FETCH_SIZE + WRITE_SIZE) KB * 1024"
    if "SQ_VALU_MFMA_BUSY_CYCLES" in res and "GRBM_GUI_ACTIVE" in res:
        # GRBM_GUI_ACTIVE is summed over the 8 XCDs, the SQ counter over all 1024 SIMDs: 128 SIMDs per XCD-cycle
        res["mfma_util_at_clock"] = round(res["SQ_VALU_MFMA_BUSY_CYCLES"] / (res["GRBM_GUI_ACTIVE"] * 128), 4)
    if durations and "GRBM_GUI_ACTIVE" in res:
        res["duration_under_counters_ns"] = sum(durations) / len(durations)
        res["clock_ghz_under_counters"] = round(res["GRBM_GUI_ACTIVE"] / 8 / res["duration_under_counters_ns"], 3)   # GRBM: summed over 8 XCDs
    res["grid_sizes_seen"] = {str(k): v for k, v in sorted(grids.items())}
    if grid_max is not None:
        res["grid_max"] = grid_max
    if srcs:
        res["kernel_sources"] = source_hashes(srcs)
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
