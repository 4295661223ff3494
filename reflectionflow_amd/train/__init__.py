"""Training path of the same blocks: LoRA reflection tuning as in train_flux/train/model.py:164-238.

  kernels.py  tensor-level wrappers over the rf_*_bwd / rf_qkv_train_* entry points of librf_flux.so
  blocks.py   DoubleStream / SingleStream blocks as torch.autograd.Functions: the forward runs the training form of the
              block on the HIP kernels and keeps only the block inputs; the backward RECOMPUTES the block (the
              gradient-checkpoint branch of train_flux/flux/transformer.py:139-157) and back-propagates through it on
              the HIP kernels -- dX through the frozen weights, dA / dB of the LoRA factors, the modulation gradients
  step.py     the flow-matching training step (x_t = (1 - t) x_0 + t x_1, MSE against x_1 - x_0) and the all-reduce of
              the LoRA gradients over the data-parallel ranks (RCCL on the GPU node, gloo in the CPU tests)

No CPU fallback: every tensor must be bf16 on a HIP device.
"""
