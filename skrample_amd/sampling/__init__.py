"""Sampler layer: fp64 host algebra that compiles every solver step to one fused HIP launch."""

from . import functional, interface, lazy, models, program, structured, tableaux, traits  # noqa: F401
