"""PyTorch-facing pieces (device noise generators)."""
