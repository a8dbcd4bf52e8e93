"""Importable alias of the ``1xgpt_amd`` package (a Python identifier cannot start with a digit; the reference's own
distribution name is ``onexgpt``): ``from onexgpt_amd import STMaskGIT, GenieConfig``."""
import importlib as _il

_pkg = _il.import_module("1xgpt_amd")
config = _il.import_module("1xgpt_amd.config")
GenieConfig = config.GenieConfig


def __getattr__(name):
    lazy = {"STMaskGIT": "st_mask_git", "GenieEvaluator": "evaluate", "RawTokenDataset": "data", "VQModel": "magvit2",
            "HipDecoder": "magvit2", "HipEncoder": "magvit2", "generate_frames": "generate",
            "generate_frames_cached": "generate", "AvgMetric": "eval_utils", "compute_loss": "eval_utils"}
    if name in lazy:
        return getattr(_il.import_module("1xgpt_amd." + lazy[name]), name)
    try:
        return _il.import_module("1xgpt_amd." + name)
    except ModuleNotFoundError as e:
        raise AttributeError(name) from e
