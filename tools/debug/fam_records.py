import sys, json, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from mpv_amd import bench_families as B
dev = torch.device("cuda", 0)
for fn in (B.record_nsdiff_spatial, B.record_tmdm):
    r = fn(dev, bench.log, cpu=True)
    print(json.dumps({k: r[k] for k in ("value", "ms_per_step", "roofline", "cpu_baseline")}, indent=0)[:1500])
