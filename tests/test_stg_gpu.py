"""GPU parity tests (-m gpu) of the DiffSTG HIP path (BASELINE configs[4]) through the C ABI, against
fixtures recorded from the reference's UGnet / sampler (tests/golden/make_golden_diffstg.py: everything but the graph convolution
is the reference's arithmetic; that layer is the restated ResGatedGraphConv, parity unpinned).  Tolerance 1e-4, elementwise."""
import types

import numpy as np
import pytest
import torch

from test_diffstg_oracle import case, state
from test_gpu_parity import dev, rel_err

pytestmark = pytest.mark.gpu
T = torch.from_numpy
RTOL = 1e-4


@pytest.fixture(scope="module")
def z():
    from conftest import GOLDEN
    return np.load(GOLDEN / "diffstg_cases.npz")


def build(z, tag, **over):
    from mpv_amd.models.models import diffusion_models
    net, V = case(z, tag)
    m = diffusion_models("DiffSTG", dict(net, device=dev(), **over))
    m.load_state_dict(state(z, tag, net), strict=True)
    return m.to(dev()).eval(), net, V


@pytest.mark.parametrize("gemm", ["f16x3", "f32"])
@pytest.mark.parametrize("tag", ["tiny", "ddpm", "c5"])
def test_ugnet_forward_matches_reference(z, tag, gemm):
    m, net, V = build(z, tag, gemm_mode=gemm)
    ctx = m.native_context()
    ctx.stg_set_graph(T(z["edge_%s" % tag]), V)
    x, xm = T(z["x_%s" % tag]).to(dev()), T(z["xm_%s" % tag]).to(dev())
    for t in (1, net["diffusion_steps"] // 2, net["diffusion_steps"]):
        eps = ctx.stg_forward(x, xm, t)
        err = rel_err(eps, z["eps_%s_t%d" % (tag, t)])
        print("DiffSTG UGnet %s t=%d rel err %.3e" % (tag, t, err))
        assert err < RTOL


@pytest.mark.parametrize("tag", ["tiny", "ddpm", "c5"])
def test_downsampling_kernel_forced_on_small_shapes(z, tag, monkeypatch):
    """csrc/stg_down_mfma.h takes the k = T + 1 "downsampling" products only at K >= 512 by itself; MPV_STG_DOWN=1 sends every product with
    N <= 160 and no epilogue through it (K down to 24, N not a multiple of 16, row counts that leave most of a 128-row block empty), =0 none.
    Both against the reference's fixtures, and against each other at half the tolerance (two fp32-equivalent split-f16 kernels that sum K in a
    different order, seen through the whole network: 1.3e-5 measured on the c5 fixture)."""
    m, net, V = build(z, tag, gemm_mode="f16x3")
    ctx = m.native_context()
    ctx.stg_set_graph(T(z["edge_%s" % tag]), V)
    x, xm = T(z["x_%s" % tag]).to(dev()), T(z["xm_%s" % tag]).to(dev())
    t = net["diffusion_steps"] // 2
    out = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("MPV_STG_DOWN", mode)
        out[mode] = ctx.stg_forward(x, xm, t).clone()
        err = rel_err(out[mode], z["eps_%s_t%d" % (tag, t)])
        print("DiffSTG UGnet %s MPV_STG_DOWN=%s rel err %.3e" % (tag, mode, err))
        assert err < RTOL
    monkeypatch.delenv("MPV_STG_DOWN")
    assert rel_err(out["1"], out["0"].cpu().numpy()) < 5e-5


@pytest.mark.parametrize("tag", ["tiny", "ddpm", "c5"])
def test_evaluation_step_matches_reference(z, tag):
    """DiffSTG.evaluation_step (parallel graph copies, sequential passes, cosine / linear time pairs, DDIM with the DDPM last
    step) with the reference's recorded normal draws injected in their order."""
    m, net, V = build(z, tag)
    data = types.SimpleNamespace(x=T(z["batch_%s" % tag]), edge_index=T(z["edge_%s" % tag]), num_nodes=V)
    outs, truth = m.evaluation_step(data, noise=T(z["eval_%s_noise" % tag]))
    want = z["eval_%s_out" % tag]
    assert truth is None and outs.device.type == "cpu" and tuple(outs.shape) == tuple(want.shape)
    err = rel_err(outs, want)
    mpv_g, mpv_w = outs.var(dim=-1, unbiased=False).mean().item(), T(want).var(dim=-1, unbiased=False).mean().item()
    print("DiffSTG evaluation_step %s rel err %.3e, MPV %.6g vs %.6g" % (tag, err, mpv_g, mpv_w))
    assert err < RTOL and abs(mpv_g - mpv_w) <= RTOL * abs(mpv_w)


def test_torch_cpu_stream_philox_mode_and_window_batching(z):
    m, net, V = build(z, "tiny", noise_source="torch_cpu")
    edge = T(z["edge_tiny"])
    data = types.SimpleNamespace(x=T(z["batch_tiny"]), edge_index=edge, num_nodes=V)
    torch.manual_seed(31)
    outs, _ = m.evaluation_step(data)
    assert rel_err(outs, z["eval_tiny_out"]) < RTOL                    # the reference's own seed replays its run
    hist = torch.stack([T(z["batch_tiny"]), T(z["batch_tiny"]) * 0.5 + 0.2]).to(dev())
    torch.manual_seed(4)
    both = m.sample_windows(hist, edge)
    torch.manual_seed(4)
    one = [m.sample_windows(hist[w:w + 1], edge) for w in range(2)]
    assert rel_err(both[0], one[0][0]) < 1e-6 and rel_err(both[1], one[1][0]) < 1e-6
    mp, _, _ = build(z, "tiny")
    torch.manual_seed(5)
    a = mp.sample_windows(hist, edge)
    torch.manual_seed(5)
    b = mp.sample_windows(hist, edge)
    c = mp.sample_windows(hist, edge)
    assert torch.equal(a, b) and not torch.equal(a, c) and bool(torch.isfinite(a).all())
    ctx = mp.native_context()
    mpv, pm, _ = ctx.reduce(a[:, :, -net["T_p"]:].contiguous())
    assert bool(torch.isfinite(mpv).all()) and float(mpv.min()) > 0


def test_stg_c_abi_error_paths(z):
    import ctypes as C
    from mpv_amd import _native
    lib = _native.load_library()
    ctx = _native.Context(dev())
    x = torch.zeros(6, 20, 1, device=dev())
    assert lib.mpv_stg_forward(ctx._h, C.c_void_p(x.data_ptr()), C.c_void_p(x.data_ptr()), 6, 1, C.c_void_p(x.data_ptr())) < 0
    assert b"mpv_stg_load" in lib.mpv_last_error(ctx._h)
    m, net, V = build(z, "tiny")
    c = m.native_context()
    assert lib.mpv_stg_forward(c._h, C.c_void_p(x.data_ptr()), C.c_void_p(x.data_ptr()), 6, 1, C.c_void_p(x.data_ptr())) < 0   # no graph yet
    c.stg_set_graph(T(z["edge_tiny"]), V)
    assert lib.mpv_stg_forward(c._h, C.c_void_p(x.data_ptr()), C.c_void_p(x.data_ptr()), 5, 1, C.c_void_p(x.data_ptr())) < 0   # rows % V
    with pytest.raises(_native.MpvError):
        c.stg_set_graph(torch.tensor([[0, 9], [1, 0]]), V)                                                      # node out of range
