"""RCCL on the one GPU a test box has: a world of ONE under backend "nccl" (= RCCL on ROCm) with MPV_FORCE_DIST=1 drives every
collective helper of `mpv_amd.distributed`, the harness's window-sharded and sample-split paths and bench.py's all-gather step
through their DEVICE branches (`dist.all_gather_into_tensor` / `broadcast` / `all_reduce` on HBM tensors), and checks the results
bit for bit against the same calls under gloo and against the unsharded run.

Every scenario runs in a fresh child process started before this process touches the GPU state of the child (nothing is exec'd
from a GPU-initialised process; the child is a plain `python child.py`)."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]

CHILD = r'''
import json, os, sys
sys.path.insert(0, %(root)r)
os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
os.environ["MPV_FORCE_DIST"] = "1"
os.environ["MASTER_ADDR"] = "127.0.0.1"
os.environ["MASTER_PORT"] = %(port)r
import numpy as np
import torch
import torch.distributed as dist
from mpv_amd import distributed as D

backend = sys.argv[1]
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
if backend == "nccl":
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
else:
    dist.init_process_group("gloo", rank=0, world_size=1)
assert D.active() and D.world_size() == 1 and D.rank() == 0
out = {"backend": dist.get_backend()}
g = torch.Generator().manual_seed(9)
a = torch.rand(7, generator=g).to(dev)
b = torch.rand(7, generator=g).to(dev)
fa, fb = D.gather_curve(7, 0, 7, a, b)
out["curve"] = [fa.tolist(), fb.tolist()]
assert np.array_equal(fa, a.cpu().numpy()) and np.array_equal(fb, b.cpu().numpy())
w = torch.rand(7, 3, 5, 1, 4, generator=g).to(dev)
fw = D.gather_windows(7, 0, 7, w)
assert fw.device.type == "cpu" and torch.equal(fw, w.cpu())
out["windows_sum"] = float(fw.double().sum())
s = torch.rand(2, 3, 5, 1, 8, generator=g).to(dev)
fs = D.gather_samples(s)
assert torch.equal(fs, s.cpu())
mean = s.double().mean(dim=-1)
m2 = ((s.double() - mean.unsqueeze(-1)) ** 2).sum(dim=-1)
var, mu = D.merge_row_moments(mean, m2, 8)
assert torch.equal(var.cpu(), (m2 / 8).cpu()) and torch.equal(mu.cpu(), mean.cpu())
out["var_sum"] = float(var.cpu().sum())            # (summed on the host: the comparison is of the gathered values, not of a device sum)
assert D.broadcast_int(123456789012345) == 123456789012345
D.barrier()

# the harness itself: window-sharded run_evaluation_cache (curve + samples + cache) under the forced group vs. the plain run
from oracle import nsdiff_oracle as O
from mpv_amd.models.models import diffusion_models
from mpv_amd.evaluation_and_analysis import diffusion_model_uncertainy as H
sd = O.random_state_dict(1, 10, 40, 24, 16, seed=11)
net = dict(device=dev, scaler_type="StandardScaler", task_model="NsDiff_model_variants", rolling_length=16, diffusion_schedule="linear",
           diffusion_steps=10, beta_start=1e-4, beta_end=0.02, n_z_samples=8, parallel_sample=4, windows=40, pred_len=24, dataset_nf=1,
           noise_source="torch_cpu")
model = diffusion_models("NsDiff_model_variants", net, train_model_select="cond_var").to(dev)
model.load_state_dict(sd, strict=True)
model.eval()
series = torch.cumsum(torch.randn(5, 70, 1, generator=g) * 0.1, dim=1)
windows, _ = H.build_sliding_windows(series, torch.arange(70.0), 40, 10)
import tempfile
with tempfile.TemporaryDirectory() as td:
    torch.manual_seed(3)
    got = H.run_evaluation_cache(model, windows, 24, os.path.join(td, "c.pt"), dev, force_recompute=True)
    cached = torch.load(os.path.join(td, "c.pt"))
    out["ews"] = [float(v) for v in got.ews]
    out["cache_len"] = len(cached)
    out["samples_sum"] = float(torch.stack(list(got)).double().sum())
    os.environ["MPV_FORCE_DIST"] = "0"
    assert not D.active()
    torch.manual_seed(3)
    ref = H.run_evaluation_cache(model, windows, 24, os.path.join(td, "d.pt"), dev, force_recompute=True)
    assert np.array_equal(np.asarray(ref.ews), np.asarray(got.ews)), (ref.ews, got.ews)
    assert all(torch.equal(x, y) for x, y in zip(ref, got))
    os.environ["MPV_FORCE_DIST"] = "1"
dist.destroy_process_group()
print("RESULT " + json.dumps(out))
'''


def _run_child(tmp_path, backend, port):
    script = tmp_path / ("child_%s.py" % backend)
    script.write_text(CHILD % {"root": str(ROOT), "port": str(port)})
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("NCCL_DEBUG", None)
    proc = subprocess.run([sys.executable, str(script), backend], capture_output=True, text=True, timeout=600, env=env)
    assert proc.returncode == 0, proc.stdout[-3000:] + "\n" + proc.stderr[-3000:]
    line = [ln for ln in proc.stdout.splitlines() if ln.startswith("RESULT ")][-1]
    return json.loads(line[len("RESULT "):])


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.gpu
def test_rccl_world1_device_branches_equal_gloo(tmp_path):
    nccl = _run_child(tmp_path, "nccl", _free_port())
    gloo = _run_child(tmp_path, "gloo", _free_port())
    assert nccl.pop("backend") == "nccl" and gloo.pop("backend") == "gloo"
    assert nccl == gloo                                        # every gathered value identical, bit for bit (floats printed by repr)
    assert nccl["cache_len"] == 4


@pytest.mark.gpu
def test_bench_forced_rccl_step(tmp_path):
    """bench.py's own collective step (`MPV_BENCH_FORCE_DIST=1`): init_process_group("nccl", device_id=...), the device-side
    all_gather_into_tensor of the curve, barrier and the MAX all-reduce of the elapsed time, on a cut-down trajectory."""
    env = dict(os.environ, MPV_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", LOCAL_RANK="0",
               WORLD_SIZE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("NCCL_DEBUG", None)
    proc = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--steps", "1", "--warmup", "0", "--max-windows", "6", "--no-cpu-baseline",
                           "--no-other-modes"], capture_output=True, text=True, timeout=600, env=env, cwd=str(ROOT))
    assert proc.returncode == 0, proc.stderr[-3000:]
    rec = json.loads([ln for ln in proc.stdout.splitlines() if ln.startswith("{")][-1])
    assert rec["n_gpus"] == 1 and rec["value"] > 0 and rec["config"]["truncated_for_profiling"] is True
    assert rec["collective"] == {"backend": "nccl", "forced_world_of_one": True}
