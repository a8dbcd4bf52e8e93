"""1xgpt_amd -- MI355X-native (gfx950) GENIE forward / MaskGIT sampling path.

The directory name starts with a digit, so import it with
``importlib.import_module("1xgpt_amd")`` (or through the ``onexgpt_amd`` alias package).
"""
from .config import GenieConfig  # noqa: F401
