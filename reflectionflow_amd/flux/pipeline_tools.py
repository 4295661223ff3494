"""`encode_images` / `prepare_text_input` with the reference's names and signatures
(train_flux/flux/pipeline_tools.py:7-30, :33-52).  VAE and text encoders stay PyTorch(-ROCm)
modules supplied by the pipeline object; this file only moves data between them and the
packed-token layout."""
from torch import Tensor


def encode_images(pipeline, images: Tensor, generator=None):
    """`generator` (not in the reference signature, default None = the reference's global-RNG draw): the torch.Generator the
    VAE posterior sample is drawn from."""
    images = pipeline.image_processor.preprocess(images)
    images = images.to(pipeline.device).to(pipeline.dtype)
    dist = pipeline.vae.encode(images).latent_dist
    images = dist.sample() if generator is None else dist.sample(generator=generator)
    images = (images - pipeline.vae.config.shift_factor) * pipeline.vae.config.scaling_factor
    images_tokens = pipeline._pack_latents(images, *images.shape)
    images_ids = pipeline._prepare_latent_image_ids(images.shape[0], images.shape[2], images.shape[3],
                                                    pipeline.device, pipeline.dtype)
    if images_tokens.shape[1] != images_ids.shape[0]:   # diffusers >= 0.32 id convention
        images_ids = pipeline._prepare_latent_image_ids(images.shape[0], images.shape[2] // 2, images.shape[3] // 2,
                                                        pipeline.device, pipeline.dtype)
    return images_tokens, images_ids


def prepare_text_input(pipeline, prompts, max_sequence_length=512, prompts_2=None):
    prompt_embeds, pooled_prompt_embeds, text_ids = pipeline.encode_prompt(
        prompt=prompts, prompt_2=prompts_2, prompt_embeds=None, pooled_prompt_embeds=None, device=pipeline.device,
        num_images_per_prompt=1, max_sequence_length=max_sequence_length, lora_scale=None)
    return prompt_embeds, pooled_prompt_embeds, text_ids
