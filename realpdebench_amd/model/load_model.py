"""Model registry with the reference's signature (realpdebench/model/load_model.py:4):
``load_model(train_dataset, device='cpu', **kwargs)`` reads ``train_dataset[0]`` for the shapes and
switches on ``kwargs['model_name']``.  Only the MI355X-native hot-path models are registered."""
import logging


def load_model(train_dataset, device="cpu", **kwargs):
    model_name = kwargs["model_name"]
    input, target = train_dataset[0]        # T, S, S, C   (load_model.py:7-9)
    input_shape = tuple(input.shape)
    output_shape = tuple(target.shape)
    logging.info(f"Loading model {model_name} with input shape {input_shape} and output shape {output_shape}")
    if model_name == "fno":
        from .fno import FNO3d
        model = FNO3d(
            modes1=kwargs["modes1"],
            modes2=kwargs["modes2"],
            modes3=kwargs["modes3"],
            n_layers=kwargs["n_layers"],
            width=kwargs["width"],
            shape_in=input_shape,
            shape_out=output_shape,
        ).to(device)
    elif model_name == "transolver":
        from .transolver import Transolver
        model = Transolver(                                   # load_model.py:145-152
            space_dim=kwargs["space_dim"], n_layers=kwargs["n_layers"], n_hidden=kwargs["n_hidden"],
            n_head=kwargs["n_head"], H=kwargs["H"], W=kwargs["W"], D=kwargs["D"], Time_Input=False, unified_pos=False,
            fun_dim=kwargs["fun_dim"], out_dim=kwargs["out_dim"], ref=kwargs["ref"], dropout=kwargs["dropout"],
            act=kwargs["act"], mlp_ratio=kwargs["mlp_ratio"], slice_num=kwargs["slice_num"],
        ).to(device)
    elif model_name == "galerkin_transformer":
        from .galerkin_transformer import GalerkinTransformer3d
        kwargs["node_feats"] = input_shape[-1]                # load_model.py:77-91
        kwargs["n_targets"] = output_shape[-1]
        kwargs["shape_in"] = input_shape
        kwargs["shape_out"] = output_shape
        kwargs.pop("config", None)
        model = GalerkinTransformer3d(**kwargs).to(device)
    elif model_name == "unet":
        from .unet import Unet3d
        model = Unet3d(dim=input_shape[1], out_channels=output_shape[-1], dim_mults=kwargs["dim_mults"],   # load_model.py:47-58
                       channels=input_shape[-1], in_time=input_shape[0], out_time=output_shape[0]).to(device)
    elif model_name == "dpot":
        from .dpot import DPOT
        keys = ("img_size", "in_channels", "out_channels", "in_timesteps", "out_timesteps", "patch_size", "embed_dim", "depth",
                "n_blocks", "modes", "mlp_ratio", "out_layer_dim", "normalize", "act", "time_agg", "n_cls", "model_type",
                "checkpoint_path")                            # load_model.py:108-131
        model = DPOT(shape_in=input_shape, shape_out=output_shape, **{k: kwargs[k] for k in keys}).to(device)
    else:
        raise ValueError(f"Model {model_name} not supported by the MI355X backend "
                         "(supported: fno, transolver, galerkin_transformer, unet, dpot)")
    return model
