"""reflectionflow_amd -- MI355X-native hot path of ReflectionFlow's FLUX denoise loop.

Only the path named by BASELINE.json's north_star lives here: HIP kernels + C ABI
(`csrc/`, `librf_flux.so`), their ctypes binding (`_lib`, `ops`), the engine that
sequences a whole denoise (`engine`), the host-side mirror of the reference's
`train_flux/flux` API (`flux/`) and the candidate-parallel search drivers (`tts/`).
"""
__version__ = "0.1.0"
