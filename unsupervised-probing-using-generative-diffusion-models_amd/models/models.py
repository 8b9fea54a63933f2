"""Model factory -- the operator-API entry point (mirrors reference models/models.py:5-32)."""


def diffusion_models(task_model, net_param, **kwargs):
    if task_model == "NsDiff":
        from .Diffusion_model.NsDiff.NsDiff_model import NsDiff_model
        return NsDiff_model(net_param=net_param, train_model_select=kwargs["train_model_select"],
                            pretrain_f_path=net_param.get("pretrain_f_path") or None,
                            pretrain_g_path=net_param.get("pretrain_g_path") or None)
    if task_model == "NsDiff_model_variants":
        from .Diffusion_model.NsDiff.NsDiff_model import NsDiff_model_variants
        return NsDiff_model_variants(net_param=net_param, train_model_select=kwargs["train_model_select"])
    if task_model == "TMDM":
        from .Diffusion_model.TMDM.tmdm_adapter import TMDM_model
        return TMDM_model(net_param=net_param)
    if task_model == "DiffusionTS":
        from .Diffusion_model.DiffusionTS.DiffusionTS_model import DiffusionTS_model
        return DiffusionTS_model(net_param=net_param)
    if task_model == "DiffSTG":
        from .Diffusion_model.DiffSTG.graph_diffusion_model import DiffSTG
        return DiffSTG(net_param=net_param)
    if task_model == "NsDiff_spatial":
        from .Diffusion_model.NsDiff.NsDiff_model import NsDiff_model_spatial
        return NsDiff_model_spatial(net_param=net_param, train_model_select=kwargs["train_model_select"],
                                    pretrain_f_path=net_param.get("pretrain_f_path") or None,
                                    pretrain_g_path=net_param.get("pretrain_g_path") or None)
    raise ValueError("the definition  don't exit\n\tyou can define it before using it")
