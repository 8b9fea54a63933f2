"""Build libmpv_hip.so (the gfx950 HIP kernels + C ABI) in-tree with hipcc.

    python -m mpv_amd.build            # or: python <package>/build.py

hipcc cross-compiles for gfx950 without a GPU.  The .so is git-ignored but travels with
the working tree to the GPU box.
"""
import os
import shutil
import subprocess
import sys
from pathlib import Path

PKG = Path(__file__).resolve().parent
ROOT = PKG.parent
SOURCES = ["mpv_api.hip", "nsdiff_chain.hip", "nsdiff_chain_f16.hip", "sigma_head.hip", "mpv_reduce.hip", "biomass_sim.hip"]
LIB = PKG / "libmpv_hip.so"


def hipcc_path() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and Path(cand).exists():
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or put /opt/rocm/bin on PATH)")


def stale() -> bool:
    if not LIB.exists():
        return True
    deps = [PKG / "csrc" / s for s in SOURCES] + [PKG / "csrc" / "mpv_common.h", PKG / "csrc" / "chain_common.h", ROOT / "include" / "mpv_hip.h"]
    return any(d.stat().st_mtime > LIB.stat().st_mtime for d in deps)


def build(force: bool = False, verbose: bool = False) -> Path:
    if not force and not stale():
        return LIB
    # -fno-slp-vectorize: packed-f32 VALU (v_pk_fma_f32 ...) issues at half rate beside MFMAs on gfx950
    # (tools/ubench/pipe_overlap4: 4 v_pk_fma per MFMA gap = 68 cycles vs 8 v_fma = 48)
    cmd = [hipcc_path(), "-O3", "--offload-arch=gfx950", "-std=c++17", "-fno-slp-vectorize", "-fPIC", "-shared",
           "-I", str(ROOT / "include"), "-I", str(PKG / "csrc"), "-o", str(LIB)]
    cmd += [str(PKG / "csrc" / s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd), flush=True)
    proc = subprocess.run(cmd, capture_output=True, text=True)
    if proc.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + proc.stdout + proc.stderr)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
