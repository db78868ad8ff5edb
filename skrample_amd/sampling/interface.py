"""Drive a structured (one-step-per-call) sampler with the functional, closure-based protocol.

`StructuredFunctionalAdapter(sampler).sample_model(x, model, transform, schedule, steps, ...)` is the canonical
denoise loop of the reference (`skrample/sampling/interface.py:13-59`): evaluate the network at every schedule
point, hand `(sample, prediction, step, noise)` to the sampler, keep as much history as the sampler asks for.
Each loop iteration costs exactly one fused kernel launch besides the network call.
"""

from __future__ import annotations

from dataclasses import dataclass

from ..common import DeltaPoint, Point, Step
from ..scheduling import SkrampleSchedule
from .functional import FunctionalSampler
from .models import DiffusionModel
from .structured import SampleInput, SKSamples, StructuredSampler

_CLEAN = Point(0, 0, 1)


@dataclass(frozen=True)
class StructuredFunctionalAdapter(FunctionalSampler):
    sampler: StructuredSampler

    def remove_noise(self, sample, noise, point: Point):
        return self.sampler.remove_noise(sample, noise, point)

    def add_noise(self, sample, noise, point: Point):
        return self.sampler.add_noise(sample, noise, point)

    def sample_model(self, sample, model, model_transform: DiffusionModel, schedule: SkrampleSchedule, steps: int, include: slice = slice(None), rng=None, callback=None):
        table = schedule.schedule(steps)
        total = len(table)
        window = self.sampler.require_previous
        draw = rng if (rng and self.sampler.require_noise) else None
        history: list[SKSamples] = []
        for index in range(total)[include]:
            here = table[index]
            step = Step.from_int(index, total)
            seen_by_model = self.sampler.scale_input(sample, here)
            packed = SampleInput(sample, model(seen_by_model, *here), step, draw(step) if draw else None)
            result = self.sampler.sample_packed(packed, model_transform, schedule, previous=history)
            if window > 0:
                history = (history + [result])[-window:]
            sample = result.final
            if callback:
                callback(sample, index, DeltaPoint(here, table[index + 1] if index + 1 < total else _CLEAN))
        return sample
