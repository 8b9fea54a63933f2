"""Build libmpv_hip.so (the gfx950 HIP kernels + C ABI) in-tree with hipcc.

    python -m mpv_amd.build            # or: python <package>/build.py [--force]

hipcc cross-compiles for gfx950 without a GPU.  Every csrc/*.hip is compiled to an object under build/ (in parallel,
only when it or a header changed) and the objects are linked into the .so, which is git-ignored but travels with the
working tree to the GPU box.
"""
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

PKG = Path(__file__).resolve().parent
ROOT = PKG.parent
CSRC = PKG / "csrc"
OBJ = PKG / "build"
LIB = PKG / "libmpv_hip.so"
# -fno-slp-vectorize: packed-f32 VALU (v_pk_fma_f32 ...) issues at half rate beside MFMAs on gfx950
# (tools/ubench/pipe_overlap4: 4 v_pk_fma per MFMA gap = 68 cycles vs 8 v_fma = 48)
FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fno-slp-vectorize", "-fPIC"]
# per-file additions.  dts.hip: MFMA results stay in VGPRs -- the short 16x16 products of the attention / MLP kernels are consumed by the
# VALU at once, and with accumulation registers as destination every tile paid 8-16 v_accvgpr_read/write (attention forward: 89 -> 77
# instructions per key tile)
EXTRA = {"dts.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"], "stg.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"]}   # (DiffSTG: 8.43 -> 8.58 windows/s)


def sources():
    return sorted(CSRC.glob("*.hip"))


def headers():
    return sorted(CSRC.glob("*.h")) + [ROOT / "include" / "mpv_hip.h"]


def hipcc_path() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and Path(cand).exists():
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or put /opt/rocm/bin on PATH)")


def _newer(path: Path, than: Path) -> bool:
    return (not than.exists()) or path.stat().st_mtime > than.stat().st_mtime


def stale() -> bool:
    return any(_newer(d, LIB) for d in sources() + headers())


def _compile(src: Path, obj: Path, verbose: bool):
    cmd = [hipcc_path(), *FLAGS, *EXTRA.get(src.name, []), "-I", str(ROOT / "include"), "-I", str(CSRC), "-c", str(src), "-o", str(obj)]
    if verbose:
        print(" ".join(cmd), flush=True)
    proc = subprocess.run(cmd, capture_output=True, text=True)
    if proc.returncode != 0:
        raise RuntimeError("hipcc failed on {}:\n{}{}".format(src.name, proc.stdout, proc.stderr))


def build(force: bool = False, verbose: bool = False) -> Path:
    if not force and not stale():
        return LIB
    OBJ.mkdir(exist_ok=True)
    hdr_time = max(h.stat().st_mtime for h in headers())
    jobs = []
    for src in sources():
        obj = OBJ / (src.stem + ".o")
        if force or _newer(src, obj) or obj.stat().st_mtime < hdr_time:
            jobs.append((src, obj))
    with ThreadPoolExecutor(max_workers=min(6, max(1, len(jobs)))) as pool:
        list(pool.map(lambda j: _compile(j[0], j[1], verbose), jobs))
    cmd = [hipcc_path(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(LIB)] + [str(OBJ / (s.stem + ".o")) for s in sources()]
    if verbose:
        print(" ".join(cmd), flush=True)
    proc = subprocess.run(cmd, capture_output=True, text=True)
    if proc.returncode != 0:
        raise RuntimeError("link failed:\n" + proc.stdout + proc.stderr)
    _write_build_info()
    return LIB


def _write_build_info():
    """build_info.json beside the library (git-ignored, travels to the GPU box, which has no .git): the commit the sources were at, whether the
    tree was dirty, the library's sha256.  tools/pmc_traffic.py copies commit + hash into every PMC traffic file."""
    import hashlib
    import json
    info = {"lib_sha256": hashlib.sha256(LIB.read_bytes()).hexdigest(), "commit": None, "dirty": None}
    try:
        info["commit"] = subprocess.run(["git", "-C", str(ROOT), "rev-parse", "HEAD"], capture_output=True, text=True, timeout=10).stdout.strip() or None
        info["dirty"] = bool(subprocess.run(["git", "-C", str(ROOT), "status", "--porcelain", "--", str(CSRC), str(ROOT / "include")],
                                            capture_output=True, text=True, timeout=10).stdout.strip())
    except (OSError, subprocess.SubprocessError):
        pass
    (PKG / "build_info.json").write_text(json.dumps(info, indent=1))


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
