"""skr_oracle -- CPU restatement of the skrample sampler-step hot path.

*** TEST INFRASTRUCTURE.  NOT PRODUCT CODE. ***

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import this package, and only as the *checker* of the HIP path -- never as the thing measured
or shipped.  Nothing under ``skrample_amd/`` imports it (tests/test_boundary.py enforces this).

What it is: a plain numpy / torch-CPU restatement, written as flat functions, of the algorithm in
the reference (Beinsezii/skrample @ 0.8.0-dev).  Each function cites the reference file:line it
follows.  It keeps the reference's *operation order* on tensors (separate mul/add/sub/div passes,
``math.sumprod`` left-to-right accumulation starting from int 0) so that it is bit-comparable with
the reference's fp32 PyTorch path.

Pinning: checked in ``tests/test_oracle_*.py`` against
  * the reference's own committed goldens (``tests/golden/reference_kats.json`` holds the numbers
    from reference tests/self_sampling.py:57-82, tests/self_scheduling.py:30-45,
    tests/miscellaneous.py:11),
  * fixtures produced by importing the reference itself in the build container
    (``tools/make_golden.py`` -> ``tests/golden/*.npz``).
"""

from . import noise, predictors, rk, samplers, scalars, schedules, wrapper  # noqa: F401
