"""Import-path compatibility: `train_flux.flux.*` resolves to the MI355X implementation in
`reflectionflow_amd.flux.*`, so scripts written against the reference's package layout
(`from train_flux.flux.generate import generate`, ...) run unchanged."""
