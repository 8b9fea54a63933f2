"""Debug aid: DiffusionTS subset replay against the oracle over batch shapes (V nodes x par samples) and plan lengths."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
from oracle import diffusionts_oracle as O
from mpv_amd.models.models import diffusion_models
from test_gpu_parity import rel_err
dev = torch.device("cuda", 0)
torch.set_num_threads(16)


def run(V, par, steps, ns=2, nv=2):
    net = dict(device=dev, scaler_type="StandardScaler", task_model="DiffusionTS", dataset_nf=1, windows=100, pred_len=100, d_model=64, n_heads=4,
               n_layer_enc=3, n_layer_dec=6, timesteps=100, diffusion_steps=steps, n_z_samples=par, parallel_sample=par)
    torch.manual_seed(123)
    model = diffusion_models("DiffusionTS", net).to(dev).eval()
    g = torch.Generator().manual_seed(77)
    hist = (torch.randn(V, 100, 1, generator=g) * 0.4).clamp(-1, 1)
    plan = model._plan()
    draws = 1 + sum(2 + int(r[6]) for r in plan.tolist() if r[1] >= 0)
    noise = torch.randn(1, draws, 1, par * V, 200, 1, generator=g)
    out = model.sample_windows(hist.unsqueeze(0).to(dev), noise=noise)[0].cpu()
    flat = out.permute(0, 3, 1, 2).reshape(V * par, 100, 1)
    ns, nv = min(ns, par), min(nv, V)
    rows = torch.tensor([s * V + v for s in range(ns) for v in range(nv)])
    it = iter(noise[0, :, 0, rows].unbind(0))
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    ocfg = O.DTSConfig(100, 100, 1, 64, 3, 6, 4, 100, steps, n_z_samples=ns, parallel_sample=ns)
    tb = O.schedule_tables(100)
    x = torch.cat([hist[:nv], torch.zeros(nv, 100, 1)], dim=1).repeat(ns, 1, 1)
    mask = torch.cat([torch.ones(100, 1, dtype=torch.bool), torch.zeros(100, 1, dtype=torch.bool)]).expand(ns * nv, -1, -1)
    want = O.fast_sample_infill(sd, ocfg, tb, x * mask, mask, lambda s: next(it), batch_of=V * par)[:, -100:, :]
    print("V=%4d par=%3d steps=%3d  draws=%3d  subset(%dx%d) rel err %.3e" % (V, par, steps, draws, ns, nv, rel_err(flat[rows], want)), flush=True)
    del model


for V, par in ((2, 2), (5, 3), (3, 5), (16, 8), (100, 32)):
    for steps in (1, 2, 5):
        run(V, par, steps)
