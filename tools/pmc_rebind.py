#!/usr/bin/env python3
"""Re-stamp profiles/<round>/pmc_traffic_*.json with the hash of the CURRENT kernel sources -- only for files whose `_lib_sha256` equals the sha256 of
the in-tree libmpv_hip.so, i.e. when the sources were edited in a way that leaves the built library bit-identical (comments, compiled-out probes).
A file measured on another binary is left alone (and stays stale for bench_common.measured_traffic).

    python tools/pmc_rebind.py profiles/r04
"""
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mpv_amd  # noqa: E402,F401
from mpv_amd import bench_common as B  # noqa: E402


def main():
    lib, src = B.lib_sha256(), B.csrc_sha256()
    for f in sorted(glob.glob(os.path.join(sys.argv[1], "pmc_traffic_*.json"))):
        d = json.load(open(f))
        if d.get("_lib_sha256") != lib:
            print("%-40s measured on another binary: left alone" % os.path.basename(f))
            continue
        if d.get("_csrc_sha256") != src:
            d["_csrc_sha256_measured"] = d.get("_csrc_sha256")
            d["_csrc_sha256"] = src
            json.dump(d, open(f, "w"), indent=1)
            print("%-40s source hash re-stamped (library bit-identical)" % os.path.basename(f))
        else:
            print("%-40s up to date" % os.path.basename(f))


if __name__ == "__main__":
    main()
