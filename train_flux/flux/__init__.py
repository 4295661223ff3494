"""Aliases of reflectionflow_amd.flux (see train_flux/__init__.py)."""
