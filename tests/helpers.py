"""Shared test helpers: write a model directory in the reference's on-disk layout."""
import yaml

def write_model_dir(path, ckpt, yaml_net, select="cond_var"):
    from mpv_amd.models.models import diffusion_models
    from mpv_amd.utils.utils import save_checkpoint
    path.mkdir(parents=True, exist_ok=True)
    net_param = dict(yaml_net)
    model = diffusion_models(net_param["task_model"], dict(net_param, device="cpu"), train_model_select=select)
    model.load_state_dict(ckpt, strict=True)
    save_checkpoint(str(path), "model_trained", model, net_param)
    cfg = {"dataset": {"windows": net_param["windows"], "pred_len": net_param["pred_len"], "sampling_t": 10},
           "net": net_param, "train": {"train_model_select": select}}
    (path / "model_trained.yaml").write_text(yaml.safe_dump(cfg))
    return cfg


WOFX_NET = dict(device="cuda", scaler_type="StandardScaler", task_model="NsDiff_model_variants", rolling_length=100,
                diffusion_schedule="linear", diffusion_steps=20, beta_start=1e-4, beta_end=0.02, n_z_samples=100,
                parallel_sample=100, windows=200, pred_len=200, dataset_nf=2)


