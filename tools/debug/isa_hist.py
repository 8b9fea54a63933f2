#!/usr/bin/env python3
"""Opcode histogram of the step body (largest basic block) of one nsdiff_chain_f16_kernel instantiation.
usage: python tools/debug/isa_hist.py [-DFLAG ...] [--kernel ILi1ELb1ELi3ELb1E] [--dump file]"""
import re, subprocess, sys, tempfile, collections
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
PKG = ROOT / "unsupervised-probing-using-generative-diffusion-models_amd"
flags = [a for a in sys.argv[1:] if a.startswith("-D")]
kern = "ILi1ELb1ELi3ELb1E"
dump = None
for i, a in enumerate(sys.argv):
    if a == "--kernel": kern = sys.argv[i + 1]
    if a == "--dump": dump = sys.argv[i + 1]
with tempfile.TemporaryDirectory() as tmp:
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-fno-slp-vectorize", "-fPIC", "-I", str(ROOT / "include"),
           "-I", str(PKG / "csrc"), "-c", str(PKG / "csrc" / "nsdiff_chain_f16.hip"), "-o", tmp + "/f16.o", "-save-temps"] + flags
    subprocess.run(cmd, cwd=tmp, check=True, capture_output=True)
    asm = next(Path(tmp).glob("*gfx950.s")).read_text()
m = re.search(r"^(_ZN\S*nsdiff_chain_f16_kernel%s\S*):" % kern, asm, re.M)
body = asm[m.end(): asm.index(".Lfunc_end", m.end())].split("\n")
blocks, cur = {}, None
for ln in body:
    lab = re.match(r"^(\.LBB\d+_\d+):", ln)
    if lab:
        cur = lab.group(1); blocks[cur] = []
    elif cur and ln.startswith("\t") and not ln.startswith(("\t.", "\t;")):
        blocks[cur].append(ln.strip())
hot = max(blocks, key=lambda k: len(blocks[k]))
ins = blocks[hot]
if dump: Path(dump).write_text("\n".join(ins))
h = collections.Counter(l.split()[0] for l in ins)
size = 0
print("step body %s: %d instructions" % (hot, len(ins)))
for k, v in h.most_common(60): print("%6d %s" % (v, k))
