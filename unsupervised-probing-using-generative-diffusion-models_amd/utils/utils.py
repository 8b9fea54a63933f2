"""Checkpoint I/O of the hot path (mirrors reference utils/utils.py:611-622, 660-689)."""
import os

import torch

from ..models.models import diffusion_models


def save_checkpoint(path: str, model_name: str, model, net_param):
    """{'net_param': dict, 'state_dict': OrderedDict} in one torch.save file (reference :611-622)."""
    torch.save({"net_param": net_param, "state_dict": model.state_dict()}, os.path.join(path, model_name))


# Optional callable(model) run after construction and BEFORE the state dict is loaded -- the place to attach a
# caller-supplied f(x) backbone (`model.attach_backbone(...)`) so that its `cond_pred_model.*` checkpoint entries load too.
# Module-level so that the harness entry points (which load models internally) pick it up without new arguments.
MODEL_HOOK = None


def load_diffusion_model(path: str, device, infer_para=None, dataparallel=True, **kwargs):
    """Checkpoint file -> (model on `device`, merged net_param).

    `infer_para` is merged into net_param BEFORE the model is constructed, `module.` prefixes
    left by DataParallel training are stripped, the state dict is loaded strictly
    (reference :660-689).  `MODEL_HOOK` (above) or `model_hook=` runs between construction and loading.
    """
    with open(path, "rb") as f:
        state = torch.load(f, map_location="cpu")
    net_param = state["net_param"]
    if infer_para is not None:
        net_param.update(infer_para)
    state_dict = state["state_dict"]
    if not torch.cuda.device_count() > 1 or not dataparallel:
        state_dict = {k.replace("module.", ""): v for k, v in state_dict.items()}
    net_param["device"] = device
    model = diffusion_models(task_model=net_param["task_model"], net_param=net_param,
                             train_model_select=kwargs["train_model_select"]).to(device)
    hook = kwargs.get("model_hook") or MODEL_HOOK
    if hook is not None:
        hook(model)
    model.load_state_dict(state_dict, strict=True)
    return model.to(device), net_param
