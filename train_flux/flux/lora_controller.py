"""Alias: the implementation lives in reflectionflow_amd/flux/lora_controller.py."""
from reflectionflow_amd.flux.lora_controller import *  # noqa: F401,F403
from reflectionflow_amd.flux import lora_controller as _impl

globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("__")})
