#!/usr/bin/env python3
"""Derived utilisation figures of the chain kernel from the raw counters of tools/run_pmc_util.sh (gpurun_out/pmc_util.json):
    python tools/pmc_util_derive.py gpurun_out/pmc_util.json profiles/r03/pmc_util_f16x3.json [windows=181]
Same derivations as profiles/r01/pmc_util_f16x3.json: one 181-window launch = W*100*100*32*50/32 tile-steps on 1024 SIMDs."""
import json
import sys

raw = json.load(open(sys.argv[1]))
W = int(sys.argv[3]) if len(sys.argv) > 3 else 181
tile_steps = W * 100 * 100 * 32 * 50 // 32
simd_cycles = raw["GRBM_GUI_ACTIVE"] / 8 * 1024
d = {
    "tile_steps_per_launch": tile_steps,
    "simd_cycles_per_tile_step": simd_cycles / tile_steps,
    "mfma_per_tile_step": raw["SQ_INSTS_MFMA"] / tile_steps * 4 / 4,
    "valu_incl_mfma_per_tile_step": raw["SQ_INSTS_VALU"] / tile_steps,
    "transcendental_per_tile_step": raw.get("SQ_INSTS_VALU_TRANS_F32", 0) / tile_steps,
    "lds_instr_per_tile_step": raw.get("SQ_INSTS_LDS", 0) / tile_steps,
    "cycles_per_mfma (SQ_VALU_MFMA_BUSY_CYCLES/SQ_INSTS_MFMA)": raw["SQ_VALU_MFMA_BUSY_CYCLES"] / raw["SQ_INSTS_MFMA"],
    "MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 * 1024 SIMDs)": raw["SQ_VALU_MFMA_BUSY_CYCLES"] / simd_cycles,
    "VALU issue active = 4*SQ_ACTIVE_INST_VALU / SIMD cycles": 4 * raw["SQ_ACTIVE_INST_VALU"] / simd_cycles,
    "mean waves per SIMD = 4*SQ_WAVE_CYCLES / SIMD cycles": 4 * raw["SQ_WAVE_CYCLES"] / simd_cycles,
    "GRBM_GUI_ACTIVE cycles per XCD": raw["GRBM_GUI_ACTIVE"] / 8,
}
out = {"_how": "rocprofv3 --kernel-trace --pmc <set> in three separate passes over `python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-other-modes` "
               "(tools/run_pmc_util.sh, tools/pmc_util_derive.py); counters of nsdiff_chain_f16_kernel<1,true,3,true>, one %d-window launch, summed over the device" % W,
       "raw": raw, "derived": d}
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps(d, indent=1))
