"""Adapter that drives a structured sampler with the functional (closure) protocol -- the canonical
denoise loop (reference `skrample/sampling/interface.py:13-59`)."""

from __future__ import annotations

import dataclasses

from .. import scheduling
from ..common import DeltaPoint, Point, Step
from . import functional, models, structured


@dataclasses.dataclass(frozen=True)
class StructuredFunctionalAdapter(functional.FunctionalSampler):
    sampler: structured.StructuredSampler

    def add_noise(self, sample, noise, point: Point):
        return self.sampler.add_noise(sample, noise, point)

    def remove_noise(self, sample, noise, point: Point):
        return self.sampler.remove_noise(sample, noise, point)

    def sample_model(self, sample, model, model_transform: models.DiffusionModel, schedule: scheduling.SkrampleSchedule, steps: int, include: slice = slice(None), rng=None, callback=None):
        history: list[structured.SKSamples] = []
        points = schedule.schedule(steps)
        keep = self.sampler.require_previous
        wants_noise = self.sampler.require_noise
        for n, point in list(enumerate(points))[include]:
            step = Step.from_int(n, len(points))
            record = self.sampler.sample_packed(
                structured.SampleInput(
                    sample=sample,
                    prediction=model(self.sampler.scale_input(sample, point), *point),
                    step=step,
                    noise=rng(step) if rng and wants_noise else None,
                ),
                model_transform,
                schedule,
                previous=history,
            )
            if keep > 0:
                history.append(record)
                history = history[max(len(history) - keep, 0) :]
            sample = record.final
            if callback:
                callback(sample, n, DeltaPoint(point, points[n + 1] if n + 1 < len(points) else Point(0, 0, 1)))
        return sample
