"""LoRA gating context managers -- same names and semantics as the reference's
train_flux/flux/lora_controller.py:5-75 (temporarily scale PEFT LoRA layers, restore on exit).

On the fused HIP path the per-token-group gating that `enable_lora` expresses (image/text rows
without LoRA, condition rows with it, unless model_config["latent_lora"]) is applied inside the
kernels via K-segments; these context managers remain for leaf-level callers and API parity.
"""
from typing import Any, List, Optional, Type

from .modules import BaseTunerLayer


class enable_lora:
    def __init__(self, lora_modules: List[BaseTunerLayer], activated: bool) -> None:
        self.activated = activated
        if activated:
            return
        self.lora_modules = [m for m in lora_modules if isinstance(m, BaseTunerLayer)]
        self.scales = [{a: m.scaling[a] for a in m.active_adapters} for m in self.lora_modules]

    def __enter__(self) -> None:
        if self.activated:
            return
        for m in self.lora_modules:
            m.scale_layer(0)

    def __exit__(self, exc_type: Optional[Type[BaseException]], exc_val: Optional[BaseException],
                 exc_tb: Optional[Any]) -> None:
        if self.activated:
            return
        for saved, m in zip(self.scales, self.lora_modules):
            for a in m.active_adapters:
                m.scaling[a] = saved[a]


class set_lora_scale:
    def __init__(self, lora_modules: List[BaseTunerLayer], scale: float) -> None:
        self.lora_modules = [m for m in lora_modules if isinstance(m, BaseTunerLayer)]
        self.scales = [{a: m.scaling[a] for a in m.active_adapters} for m in self.lora_modules]
        self.scale = scale

    def __enter__(self) -> None:
        for m in self.lora_modules:
            m.scale_layer(self.scale)

    def __exit__(self, exc_type: Optional[Type[BaseException]], exc_val: Optional[BaseException],
                 exc_tb: Optional[Any]) -> None:
        for saved, m in zip(self.scales, self.lora_modules):
            for a in m.active_adapters:
                m.scaling[a] = saved[a]
