"""GPU parity tests (-m gpu) of the NsDiff_spatial HIP path through the C ABI (mpv_nss_*), against fixtures
recorded from the reference's NsDiff_net_spatial / UGnet / NsDiff_model_spatial.evaluation_step (tests/golden/make_golden_nsdiff_spatial.py:
everything but the graph convolution and f(x) is the reference's arithmetic).  Tolerance 1e-4, elementwise."""
import types

import numpy as np
import pytest
import torch

from test_gpu_parity import dev, rel_err
from test_nsdiff_spatial_oracle import case, state

pytestmark = pytest.mark.gpu
T = torch.from_numpy
RTOL = 1e-4


@pytest.fixture(scope="module")
def z():
    from conftest import GOLDEN
    return np.load(GOLDEN / "nsdiff_spatial_cases.npz")


def build(z, tag, **over):
    from mpv_amd.models.models import diffusion_models
    net, V = case(z, tag)
    m = diffusion_models("NsDiff_spatial", dict(net, device=dev(), **over), train_model_select="NsDiff_model")
    m.load_state_dict(state(z, tag, net), strict=True)
    return m.to(dev()).eval(), net, V


@pytest.mark.parametrize("gemm", ["f16x3", "f32"])
@pytest.mark.parametrize("tag", ["tiny", "f2", "wide"])
def test_network_matches_reference(z, tag, gemm):
    from oracle import nsdiff_spatial_oracle as O
    m, net, V = build(z, tag, gemm_mode=gemm)
    ctx = m.native_context()
    P = net["parallel_sample"]
    edge = T(z["edge_%s" % tag])
    # the reference hands the network the copy-major duplicated edge index over the P*V rows; the library takes the ONE graph and offsets the copies
    ctx.nss_set_graph(edge, V)
    y, y0, gx = (T(z["%s_%s" % (k, tag)]).to(dev()) for k in ("y", "y0", "gx"))
    for t in (0, net["diffusion_steps"] // 2, net["diffusion_steps"] - 1):
        eps, sig = ctx.nss_forward(y, y0, gx, t)
        e1, e2 = rel_err(eps, z["eps_%s_t%d" % (tag, t)]), rel_err(sig, z["sig_%s_t%d" % (tag, t)])
        print("NsDiff_spatial UGnet %s t=%d (%s): eps rel err %.3e, sigma rel err %.3e" % (tag, t, gemm, e1, e2))
        assert e1 < RTOL and e2 < RTOL


@pytest.mark.parametrize("tag", ["tiny", "f2", "wide"])
def test_evaluation_step_matches_reference(z, tag):
    m, net, V = build(z, tag)
    data = types.SimpleNamespace(x=T(z["batch_%s" % tag]), edge_index=T(z["edge_%s" % tag]), num_nodes=V)
    outs, truth = m.evaluation_step(data, y0hat=T(z["y0eval_%s" % tag]), noise=T(z["eval_%s_noise" % tag]))
    want = z["eval_%s_out" % tag]
    assert truth is None and outs.device.type == "cpu" and tuple(outs.shape) == tuple(want.shape)
    err = rel_err(outs, want)
    mpv_g, mpv_w = outs.var(dim=-1, unbiased=False).mean().item(), T(want).var(dim=-1, unbiased=False).mean().item()
    print("NsDiff_spatial evaluation_step %s rel err %.3e, MPV %.6g vs %.6g" % (tag, err, mpv_g, mpv_w))
    assert err < RTOL and abs(mpv_g - mpv_w) <= RTOL * abs(mpv_w)


def test_torch_cpu_stream_philox_mode_and_window_batching(z):
    m, net, V = build(z, "tiny", noise_source="torch_cpu")
    edge, y0 = T(z["edge_tiny"]), T(z["y0eval_tiny"])
    data = types.SimpleNamespace(x=T(z["batch_tiny"]), edge_index=edge, num_nodes=V)
    torch.manual_seed(31)
    outs, _ = m.evaluation_step(data, y0hat=y0)
    assert rel_err(outs, z["eval_tiny_out"]) < RTOL                    # the reference's own seed replays its run
    hist = torch.stack([T(z["batch_tiny"]), T(z["batch_tiny"]) * 0.5 + 0.2]).to(dev())
    y0w = torch.stack([y0, y0 * 0.5 - 0.1])
    torch.manual_seed(4)
    both = m.sample_windows(hist, edge, y0w)
    torch.manual_seed(4)
    one = [m.sample_windows(hist[w:w + 1], edge, y0w[w:w + 1]) for w in range(2)]
    assert rel_err(both[0], one[0][0]) < 1e-6 and rel_err(both[1], one[1][0]) < 1e-6
    mp, _, _ = build(z, "tiny")
    a = mp.sample_windows(hist, edge, y0w, seed=5)
    b = mp.sample_windows(hist, edge, y0w, seed=5)
    c = mp.sample_windows(hist, edge, y0w, seed=6)
    assert torch.equal(a, b) and not torch.equal(a, c) and bool(torch.isfinite(a).all())
    mpv, pm, _ = mp.native_context().reduce(a)
    assert bool(torch.isfinite(mpv).all()) and float(mpv.min()) > 0


def test_window_blocks_do_not_change_the_samples(z, monkeypatch):
    """mpv_nss_sample runs a long window list in blocks of at most rows_per_pass node rows (ABI v3): Philox keys and injected draws are
    addressed by (window, chunk, position), so one window per pass, two per pass and all five at once give the same bits."""
    m, net, V = build(z, "tiny")
    edge, y0 = T(z["edge_tiny"]), T(z["y0eval_tiny"])
    g = torch.Generator().manual_seed(8)
    scale = 1.0 + 0.2 * torch.randn(5, 1, 1, 1, generator=g)
    hist = (T(z["batch_tiny"]).unsqueeze(0) * scale).to(dev())
    y0w = y0.unsqueeze(0) * scale
    PV = net["parallel_sample"] * V
    whole = m.sample_windows(hist, edge, y0w, seed=11)
    for rows in (1, PV, 2 * PV + 1):                                                # 1 row: still one whole window per pass
        monkeypatch.setenv("MPV_NSS_ROWS", str(rows))
        assert torch.equal(m.sample_windows(hist, edge, y0w, seed=11), whole), "rows_per_pass = %d changed the samples" % rows
    monkeypatch.delenv("MPV_NSS_ROWS")
    mt, _, _ = build(z, "tiny", noise_source="torch_cpu")
    torch.manual_seed(12)
    whole = mt.sample_windows(hist, edge, y0w)
    monkeypatch.setenv("MPV_NSS_ROWS", str(2 * PV))
    torch.manual_seed(12)
    assert torch.equal(mt.sample_windows(hist, edge, y0w), whole)


def test_nss_c_abi_error_paths(z):
    import ctypes as C
    from mpv_amd import _native
    lib = _native.load_library()
    ctx = _native.Context(dev())
    x = torch.zeros(12, 8, 1, device=dev())
    p = C.c_void_p(x.data_ptr())
    assert lib.mpv_nss_forward(ctx._h, p, p, p, 12, 1, p, p) < 0
    assert b"mpv_nss_load" in lib.mpv_last_error(ctx._h)
    m, net, V = build(z, "tiny")
    c = m.native_context()
    assert lib.mpv_nss_forward(c._h, p, p, p, 12, 1, p, p) < 0                       # no graph yet
    c.nss_set_graph(T(z["edge_tiny"]), V)
    assert lib.mpv_nss_forward(c._h, p, p, p, 11, 1, p, p) < 0                       # rows % V
    assert lib.mpv_nss_forward(c._h, p, p, p, 12, net["diffusion_steps"], p, p) < 0  # t out of range
    with pytest.raises(_native.MpvError):
        c.nss_set_graph(torch.tensor([[0, 9], [1, 0]]), V)                           # node out of range
    with pytest.raises(_native.MpvError):
        c.nss_sample(torch.zeros(1, V, 8, 1, device=dev()), torch.ones(1, V, 8, 1, device=dev()), T(z["edge_tiny"]), 3, 2)      # S % P
    d = _native.NssDesc(10, 1, 4, 4, 5, 0)                                           # pred_len not a multiple of 4
    assert lib.mpv_nss_load(c._h, C.byref(d), None, 0, None, None) < 0
