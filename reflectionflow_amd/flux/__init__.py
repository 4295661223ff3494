"""Host-side mirror of the reference's `train_flux/flux` package (same module and function names:
block, transformer, generate, condition, lora_controller, pipeline_tools).  Kept import-light,
like the reference's empty __init__."""
