"""Scheduler-shaped driver in the reference's op order -- the CPU baseline that bench.py times and
the tensor-level checker for the HIP wrapper.  Follows reference skrample/diffusers.py:312-346
(get_step_noise), :494-599 (SkrampleWrapperScheduler.set_timesteps/step) and :798-873 (RK step).
"""

from __future__ import annotations

import torch

from . import noise as N
from . import samplers as A
from .rk import InsideOutRK
from .scalars import stp_from_int
from .schedules import Sched, space_regularize


class StepDriver:
    """Equivalent of SkrampleWrapperScheduler for one (sampler cfg, schedule, predictor).

    `noise_kind` in {"random","offset","pyramid","colored"}; `seeds` = one int per batch item
    (the reference takes torch.Generators; their only use is as per-sample randn streams).
    `mimic_copies=True` also performs the tensor deep-copies the reference incurs through
    dataclasses.asdict (structured.py:113-125) so that timing is a faithful baseline.
    """

    def __init__(self, cfg: dict, sched: Sched, pred="eps", compute=torch.float32, noise_kind="random", noise_kw=None, invert=False, mimic_copies=False):
        self.cfg, self.sched, self.pred, self.compute = cfg, sched, pred, compute
        self.noise_kind, self.noise_kw = noise_kind, dict(noise_kw or {})
        self.invert, self.mimic_copies = invert, mimic_copies
        self.set_timesteps(50)

    def set_timesteps(self, steps: int) -> None:
        "diffusers.py:494-538 (dynamic Karras/FlowShift replacement is the caller's business here)"
        self.steps = steps
        self.table = self.sched.schedule_np(steps)
        self.previous: list[A.Rec] = []
        self._gens = None

    @property
    def timesteps(self) -> torch.Tensor:
        return torch.from_numpy(self.table[:, 0])

    @property
    def sigmas(self) -> torch.Tensor:
        "diffusers.py:266-270"
        s = torch.from_numpy(space_regularize(self.sched.space, self.table[:, 1]))
        return torch.cat([s, torch.zeros([1], dtype=s.dtype)])

    def _noise(self, step, sample: torch.Tensor, seeds) -> torch.Tensor:
        "diffusers.py:312-346 + noise.py:438-446: one generator per batch item, stacked, cast"
        if self._gens is None:
            self._gens = [N.torch_draws(torch.Generator().manual_seed(int(s))) for s in seeds]
        unit = tuple(sample.shape[1:])
        kind, kw = self.noise_kind, self.noise_kw
        outs = []
        for randn, rand1 in self._gens:
            if kind == "random":
                outs.append(N.random_noise(unit, randn))
            elif kind == "offset":
                outs.append(N.offset_noise(unit, randn, **kw))
            elif kind == "pyramid":
                outs.append(N.pyramid_noise(unit, randn, rand1, **kw))
            elif kind == "colored":
                outs.append(N.colored_noise(unit, randn, step, **kw))
            else:
                raise KeyError(kind)
        return torch.stack(outs).to(dtype=self.compute or sample.dtype, device=sample.device)

    def step(self, model_output: torch.Tensor, timestep, sample: torch.Tensor, seeds=None, noise=None):
        "diffusers.py:550-599.  `noise` (optional) injects the step noise instead of drawing it."
        if self.invert:
            model_output = -model_output
        t = timestep if isinstance(timestep, (int, float)) else timestep.item()
        idx = self.table[:, 0].tolist().index(t)
        step = stp_from_int(idx, len(self.table))
        if A.require_noise(self.cfg):
            if noise is None:
                noise = self._noise(step, sample, seeds)
            else:
                noise = noise.to(dtype=self.compute or sample.dtype)
        else:
            noise = None
        cur = A.Rec(sample.to(dtype=self.compute), model_output.to(dtype=self.compute), step, noise)
        rec = A.sample_packed(self.cfg, cur, self.pred, self.sched, self.previous)
        if self.mimic_copies:
            rec = A.Rec(rec.sample.clone(), rec.prediction.clone(), rec.step, None if rec.noise is None else rec.noise.clone(), rec.final)
        self.previous.append(rec)
        self.previous = self.previous[max(len(self.previous) - A.require_previous(self.cfg), 0) :]
        return rec.final.to(dtype=model_output.dtype), rec.prediction.to(dtype=model_output.dtype)


class RKDriver:
    "Equivalent of RKUltraWrapperScheduler (fixed tableau) -- diffusers.py:602-963"

    def __init__(self, tab, sched: Sched, pred="eps", deriv="data", eta: float = 0, compute=torch.float32, invert=False):
        self.tab, self.sched, self.pred, self.deriv, self.eta, self.compute, self.invert = tab, sched, pred, deriv, eta, compute, invert

    def set_timesteps(self, steps: int) -> None:
        self.machine = InsideOutRK(self.tab, self.sched, steps, self.pred, self.deriv, self.eta)
        p0 = self.sched.point(0)
        pts = [p for p in self.machine.all_points if abs(p.t - p0.t) > 1e-8 and abs(p.s - p0.s) > 1e-8]  # :652-666
        self.table = pts or list(self.machine.all_points)

    @property
    def timesteps(self) -> torch.Tensor:
        return torch.tensor([p.t for p in self.table], dtype=torch.float64)

    def step(self, model_output: torch.Tensor, timestep, sample: torch.Tensor, noise_fn=None):
        if self.invert:
            model_output = -model_output
        t = timestep if isinstance(timestep, (int, float)) else timestep.item()
        assert t == self.machine.all_points[self.machine.index].t
        out = self.machine.feed(sample, model_output, noise_fn, cast=lambda v: v.to(dtype=self.compute))
        return out.to(dtype=model_output.dtype)
