"""`Condition` with the reference's constructor and `encode` contract
(train_flux/flux/condition.py:24-132): holds the condition image, its type and the RoPE
`position_delta`; `encode(pipe)` returns (tokens [B,S_c,64], ids [S_c,3], type_id [S_c,1]).

Also accepts pre-encoded tokens (`tokens=`) so the search runner can hand a candidate's latents to
the next round without a PNG -> VAE round trip; that is an addition, not a change of the
reference call sites (tts_reflectionflow.py:273-279 keeps working)."""
from typing import Optional, Tuple

import torch

from .pipeline_tools import encode_images

condition_dict = {"depth": 0, "canny": 1, "subject": 4, "coloring": 6, "deblurring": 7, "depth_pred": 8,
                  "fill": 9, "sr": 10, "cartoon": 11, "cot": 12}

_PASS_THROUGH = ("subject", "fill", "cartoon", "sr", "cot", "depth_pred")


class Condition(object):
    def __init__(self, condition_type: str, raw_img=None, condition=None, mask=None, position_delta=None,
                 tokens: Optional[torch.Tensor] = None, ids: Optional[torch.Tensor] = None) -> None:
        self.condition_type = condition_type
        if condition_type not in condition_dict:
            raise NotImplementedError(f"Condition type {condition_type} not implemented")
        assert raw_img is not None or condition is not None or tokens is not None
        assert mask is None, "Mask not supported yet"
        self.tokens, self.ids = tokens, ids
        if tokens is not None:
            assert ids is not None, "pre-encoded condition tokens need their position ids"
            self.condition = None
        elif raw_img is not None:
            self.condition = self.get_condition(condition_type, raw_img)
        else:
            self.condition = condition
        self.position_delta = position_delta
        self.generator = None     # optional torch.Generator for the VAE posterior sample in encode() (None: global RNG, as the reference)

    def with_generator(self, generator):
        """A shallow copy that samples the VAE posterior from `generator` (the image / tokens are shared, not copied)."""
        import copy
        c = copy.copy(self)
        c.generator = generator
        return c

    def get_condition(self, condition_type: str, raw_img):
        if condition_type in _PASS_THROUGH:
            return raw_img.convert("RGB") if hasattr(raw_img, "convert") else raw_img
        if condition_type == "coloring":
            return raw_img.convert("L").convert("RGB")
        if condition_type == "deblurring":
            from PIL import ImageFilter
            return raw_img.convert("RGB").filter(ImageFilter.GaussianBlur(10)).convert("RGB")
        # depth needs a HF depth model, canny needs OpenCV: neither is on the ReflectionFlow path ("cot")
        raise NotImplementedError(f"building a '{condition_type}' condition from a raw image needs an external "
                                  "model/library that is outside the denoise hot path; pass `condition=` instead")

    @property
    def type_id(self) -> int:
        return condition_dict[self.condition_type]

    @classmethod
    def get_type_id(cls, condition_type: str) -> int:
        return condition_dict[condition_type]

    def encode(self, pipe, empty: bool = False) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        if self.tokens is not None:
            tokens, ids = self.tokens, self.ids.clone()
        else:
            # NB the reference encodes the real condition even when empty=True (condition.py:114-121)
            tokens, ids = encode_images(pipe, self.condition) if self.generator is None else \
                encode_images(pipe, self.condition, generator=self.generator)
        if self.position_delta is None and self.condition_type == "subject" and self.condition is not None:
            self.position_delta = [0, -self.condition.size[0] // 16]
        if self.position_delta is not None:
            ids[:, 1] += self.position_delta[0]
            ids[:, 2] += self.position_delta[1]
        type_id = torch.ones_like(ids[:, :1]) * self.type_id
        return tokens, ids, type_id
