"""Alias: the implementation lives in reflectionflow_amd/flux/pipeline_tools.py."""
from reflectionflow_amd.flux.pipeline_tools import *  # noqa: F401,F403
from reflectionflow_amd.flux import pipeline_tools as _impl

globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("__")})
