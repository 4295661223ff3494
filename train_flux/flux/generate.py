"""Alias: the implementation lives in reflectionflow_amd/flux/generate.py."""
from reflectionflow_amd.flux.generate import *  # noqa: F401,F403
from reflectionflow_amd.flux import generate as _impl

globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("__")})
