#!/usr/bin/env python3
"""Aggregate rocprofv3 passes of one bench step into profiles/<round>/pmc_traffic_<key>.json.

    python3 tools/pmc_traffic.py <key> <windows_in_step> <dir with stats/ pmc_FETCH_SIZE/ pmc_WRITE_SIZE/> <out.json> ["how" text]

Inputs: three separate runs of the same command (the microarchitecture guide: FETCH_SIZE and WRITE_SIZE do not fit one pass, and PMC passes
are collected with --kernel-trace only) -- `--kernel-trace --stats` (durations), `--pmc FETCH_SIZE`, `--pmc WRITE_SIZE`.  Corrections as the guide
prescribes for gfx950: FETCH_SIZE (KB) reports half of the bytes of wide coalesced reads -> doubled; WRITE_SIZE (KB) exact."""
import csv
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def binary_identity():
    """sha256 of the library the passes ran on + the commit it was built from (build.py leaves build_info.json next to the .so; the GPU box has
    no .git).  bench_common.measured_traffic quotes a traffic file only for the library with this hash."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("_bc", os.path.join(ROOT, "unsupervised-probing-using-generative-diffusion-models_amd", "bench_common.py"))
    bc = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bc)
    info = {}
    try:
        info = json.load(open(os.path.join(ROOT, "unsupervised-probing-using-generative-diffusion-models_amd", "build_info.json")))
    except (OSError, ValueError):
        pass
    return bc.lib_sha256(), info.get("commit"), info.get("dirty"), bc.csrc_sha256(ROOT)


def short(name):
    name = name.replace("(anonymous namespace)::", "")
    name = re.sub(r"^void ", "", name)
    return name.split("(")[0].strip()


def counters(d, counter):
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    per = {}
    for f in files:
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                per.setdefault(short(r["Kernel_Name"]), []).append(float(r["Counter_Value"]))
    return per


def durations(d):
    files = glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)
    per = {}
    for f in files:
        for r in csv.DictReader(open(f)):
            per[short(r["Name"])] = {"calls": int(r["Calls"]), "total_ms": float(r["TotalDurationNs"]) / 1e6, "avg_us": float(r["AverageNs"]) / 1e3,
                                     "share_pct": float(r["Percentage"])}
    return per


def main():
    key, windows, root, out = sys.argv[1], int(sys.argv[2]), sys.argv[3], sys.argv[4]
    how = sys.argv[5] if len(sys.argv) > 5 else ""
    fetch, write, dur = counters(os.path.join(root, "pmc_FETCH_SIZE"), "FETCH_SIZE"), counters(os.path.join(root, "pmc_WRITE_SIZE"), "WRITE_SIZE"), durations(os.path.join(root, "stats"))
    kernels, total = {}, 0.0
    for k in sorted(set(fetch) | set(write)):
        nf, nw = max(len(fetch.get(k, [])), 1), max(len(write.get(k, [])), 1)
        fb = 2.0 * 1024.0 * sum(fetch.get(k, [])) / nf     # per launch; gfx950: FETCH_SIZE tallies 128-byte requests at 64 bytes
        wb = 1024.0 * sum(write.get(k, [])) / nw
        # the step's launches of this kernel come from the stats pass: the PMC passes may run a SHORTER plan of the same launches (c3:
        # MPV_BENCH_DTS_STEPS) -- bytes per launch do not depend on the plan length, the number of launches does
        calls = dur[k]["calls"] if k in dur else max(len(fetch.get(k, [])), len(write.get(k, [])))
        rec = {"launches": calls, "pmc_launches": max(len(fetch.get(k, [])), len(write.get(k, []))), "fetch_bytes_per_launch": fb,
               "write_bytes_per_launch": wb, "hbm_bytes": (fb + wb) * calls}
        if k in dur and dur[k]["total_ms"] > 0:
            rec.update(avg_us=dur[k]["avg_us"], share_pct=dur[k]["share_pct"], GBps=(fb + wb) / (dur[k]["avg_us"] * 1e-6) / 1e9,
                       frac_of_8TBps=(fb + wb) / (dur[k]["avg_us"] * 1e-6) / 8e12)
        kernels[k] = rec
        total += rec["hbm_bytes"]
    top = sorted(kernels.items(), key=lambda kv: -kv[1].get("share_pct", 0.0))
    sha, commit, dirty, src = binary_identity()
    res = {"_how": how, "_lib_sha256": sha, "_csrc_sha256": src, "_commit": commit, "_commit_dirty": dirty, "_correction": "MI355X_MICROARCH.md HBM section: FETCH_SIZE (KB) x 1024 x 2 (gfx950 reports half of wide coalesced reads), "
                                       "WRITE_SIZE (KB) x 1024 exact; separate --pmc passes, durations from a third --kernel-trace --stats pass",
           "windows_in_step": windows, "hbm_bytes_total": total, "bytes_per_window": total / max(windows, 1),
           "dominant_kernel": top[0][0] if top else None, "kernels": dict(top)}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps({k: v for k, v in res.items() if k != "kernels"}, indent=1))
    for k, v in top[:12]:
        print("%-48s launches %6d  %8.1f MB/launch  %6.1f us  %6.0f GB/s  %5.1f %%" % (k[:48], v["launches"], v["hbm_bytes"] / max(v["launches"], 1) / 1e6,
                                                                                        v.get("avg_us", 0), v.get("GBps", 0), v.get("share_pct", 0)))


if __name__ == "__main__":
    main()
