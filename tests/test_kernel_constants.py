"""The polynomial constants of the DiffusionTS kernels' elementary functions, checked on the CPU.

`erf_abs7` (csrc/dts_kernels.h) and `cos_medium` (csrc/dts_small.h) replace library routines inside GPU kernels; the GPU parity tests see them only
through whole-network tolerances.  Here the constants are parsed out of the sources and the same arithmetic is replayed in numpy float32, so a
mistyped digit fails a CPU test with the function's own error bound (erf: 2e-7 absolute on the real line; cos: 3e-7 absolute for |x| < 2^13)."""
import math
import re

import numpy as np
from scipy.special import erf

from conftest import ROOT

CSRC = ROOT / "unsupervised-probing-using-generative-diffusion-models_amd" / "csrc"
F = np.float32


def _body(path, name):
    src = path.read_text()
    at = src.index(name + "(float x) {")
    return src[at: src.index("\n}\n", at)]


def _floats(text):
    return [float(t.rstrip("f")) for t in re.findall(r"(?<![\w.])-?\d+\.\d+(?:e[+-]?\d+)?f", text)]


def test_erf_abs7_is_accurate_to_its_stated_bound():
    body = _body(CSRC / "dts_kernels.h", "erf_abs7")
    coef = [c for c in _floats(body) if abs(c) not in (4.0, 1.0)]
    assert len(coef) == 9 and abs(coef[-1] + 1.627907258) < 1e-6, coef          # degree 9, no constant term, leading-order -2/sqrt(pi)/ln2
    x = np.concatenate([np.linspace(-6.0, 6.0, 400001), np.array([0.0, 1e-8, -1e-8, 1e-3, 3.9, 4.0, 50.0, -50.0])]).astype(F)
    t = np.minimum(np.abs(x), F(4.0))
    p = np.full_like(t, F(coef[0]))
    for c in coef[1:]:
        p = (p * t + F(c)).astype(F)                                             # (fused on the GPU: one rounding less per step)
    got = np.copysign(F(1.0) - np.exp2((p * t).astype(F)).astype(F), x).astype(np.float64)
    err = np.abs(got - erf(x.astype(np.float64)))
    assert err.max() < 2e-7, err.max()
    assert got[np.abs(x) >= 4.0].tolist() == np.sign(x[np.abs(x) >= 4.0]).tolist()


def test_cos_medium_reduction_constants_and_polynomials():
    body = _body(CSRC / "dts_small.h", "cos_medium")
    c = _floats(body)
    two_over_pi, c1, c2, c3 = c[0], c[1], c[2], c[3]
    assert abs(two_over_pi - 2.0 / math.pi) < 1e-8
    assert abs((c1 + c2 + c3) - math.pi / 2.0) < 1e-16 and F(c1) == c1            # three-term split of pi/2, leading term exact in fp32
    cos_c = [c[4], c[5], c[6]]
    sin_c = [c[9], c[10], c[11]]
    assert [round(v, 12) for v in (c[7], c[8])] == [-0.5, 1.0]
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.uniform(-8192.0, 8192.0, 400000), np.linspace(-700.0, 700.0, 200001)]).astype(F)
    k = np.rint(x * F(two_over_pi)).astype(F)
    r = x.astype(np.float64) - k.astype(np.float64) * c1                         # fused multiply-adds: exact product, one rounding each
    r = r.astype(F).astype(np.float64) - k.astype(np.float64) * c2
    r = (r.astype(F).astype(np.float64) - k.astype(np.float64) * c3).astype(F)
    z = (r * r).astype(F)
    cs = ((z * z) * ((F(cos_c[0]) * z + F(cos_c[1])) * z + F(cos_c[2])) + (F(-0.5) * z + F(1.0))).astype(F)
    sn = ((((F(sin_c[0]) * z + F(sin_c[1])) * z + F(sin_c[2])) * z) * r + r).astype(F)
    q = k.astype(np.int64)
    v = np.where(q & 1, sn, cs)
    got = np.where((q + 1) & 2, -v, v).astype(np.float64)
    err = np.abs(got - np.cos(x.astype(np.float64)))
    assert err.max() < 3e-7, err.max()
