import sys, time, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from mpv_amd import synthetic, _native
from mpv_amd.evaluation_and_analysis import diffusion_model_uncertainy as H
dev = torch.device("cuda", 0)
CFG = bench.CFG
ctx0 = _native.Context(dev)
series, _ = synthetic.biomass_trajectory_device(ctx0, CFG["nodes"], CFG["ticks"], seed=1000, sampling_stride=100)
sampled = series.cpu().contiguous()
model = bench.build_model(dev, sampled)
ctx = model.native_context()
sd = sampled.to(dev)
mean_dev, std_dev = model.scaler_mean.float().contiguous(), model.scaler_std.float().contiguous()
W = 181
def sync(): torch.cuda.synchronize(dev)
for rep in range(3):
    sync(); t0 = time.perf_counter()
    gx = ctx.sigma_head(sd, W, 0, 5, mean_dev, std_dev, add_eps=model.EPS); sync(); t1 = time.perf_counter()
    _, mpv, pm, _ = ctx.sample(gx, None, 32, 32, seed=rep, gemm="f16x3", summary=True, want_samples=False, std=std_dev, mean=mean_dev); t2 = time.perf_counter()
    sync(); t3 = time.perf_counter()
    c = torch.stack([mpv, pm]).cpu(); t4 = time.perf_counter()
    print("sigma %.2f ms | sample call returns after %.2f ms | kernel done +%.2f ms | stack+cpu %.2f ms | total %.2f" % ((t1-t0)*1e3, (t2-t1)*1e3, (t3-t2)*1e3, (t4-t3)*1e3, (t4-t0)*1e3))
ctx.timing_enable(True); ctx.timing()
sync(); t0 = time.perf_counter()
for i in range(3):
    gx = ctx.sigma_head(sd, W, 0, 5, mean_dev, std_dev, add_eps=model.EPS)
    _, mpv, pm, _ = ctx.sample(gx, None, 32, 32, seed=i, gemm="f16x3", summary=True, want_samples=False, std=std_dev, mean=mean_dev)
    c = torch.stack([mpv, pm]).cpu()
sync(); dt = (time.perf_counter() - t0) / 3
print("per step %.2f ms, kernel avg %.2f ms" % (dt * 1e3, ctx.timing()[0]))
