"""Whole-loop capture: record an N-step sampling loop (network + fused sampler steps) into one HIP graph.

Every launch of this engine is capture-safe by construction (`skr_step_launch` only enqueues a kernel on the
caller's stream: no allocation, no synchronisation, no host read-back; per-sample seeds and all operands are
read from device memory at run time), so a sampling loop whose network is itself capturable can be recorded
with `torch.cuda.graph` and replayed with new inputs / new seeds at ~10 us of host cost for the whole loop.
This removes the per-step Python + launch overhead that dominates small batches.

`indexed=True` additionally moves every step's scalars (coefficients, zeta, Philox stream ids, conversion constants) out
of the frozen kernel arguments into a device-resident table (`skr_step_launch_indexed`, include/skrample_hip.h): the kernels
read row `index + k` when they run.  The captured loop then serves ANY schedule of its length -- other sigmas, flow shift
(`mu`), begin index, stochasticity -- by rewriting the table (`CapturedLoop.retarget`, a dry run of the new scheduler on one
sample plus one small copy; no re-capture), and keeps several schedules resident at once, selected per replay by moving the
device-resident index (`loop(latents, slot=k)`).  The network's timestep argument is data as well: with `device_timesteps`
(the default of an indexed capture) `model(x, t)` receives 0-d elements of ONE device tensor, as a diffusers pipeline's
`for t in scheduler.timesteps` does, and re-targeting / switching slots rewrites that tensor -- a host float would be frozen into
the captured network.  The reference redoes this work on the host every step, with a device sync
(skrample/diffusers.py:565-567, skrample/scheduling.py:51-62, skrample/sampling/interface.py:34-59).
"""

from __future__ import annotations

from collections import OrderedDict
from typing import Callable, Sequence

import torch


class CapturedLoop:
    "replayable sampling loop: `out = loop(initial_latents, seeds=None)`"

    def __init__(self, graph: torch.cuda.CUDAGraph, static_in: torch.Tensor, static_out: torch.Tensor, seeds_dev: torch.Tensor | None, rows=None, runner=None,
                 static_times: torch.Tensor | None = None):
        self.graph, self.static_in, self.static_out, self.seeds_dev = graph, static_in, static_out, seeds_dev
        self.rows, self._runner = rows, runner  # _hip.IndexedRows of an indexed capture; runner(wrapper, x) = the captured loop body
        self._filled = {0}  # table slots that hold a schedule (the capture itself fills slot 0; the others start as zero rows)
        # device_timesteps: the tensor whose elements the captured network reads as `t`, and each slot's values for it
        self.static_times = static_times
        self._times = {0: static_times.detach().cpu().clone()} if static_times is not None else {}
        self._times_slot = 0

    @property
    def slots(self) -> int:
        return self.rows.slots if self.rows is not None else 1

    def retarget(self, wrapper, slot: int = 0) -> None:
        """Load the step scalars of `wrapper` (same sampler structure and number of steps as the captured one, any schedule /
        shift / begin index / stochasticity) into table slot `slot`: a dry run of its loop on one sample fills the rows, one
        small host-to-device copy publishes them.  The graph itself is untouched."""
        from . import _hip

        if self.rows is None:
            raise ValueError("capture the loop with indexed=True to re-target it")
        if not 0 <= slot < self.rows.slots:
            raise ValueError(f"slot {slot} outside 0..{self.rows.slots - 1}")
        self.rows.begin("refill", slot)
        _hip.indexed = self.rows
        try:
            self._runner(wrapper, self.static_in[:1].clone())
        finally:
            _hip.indexed = None
        if self.rows.cursor != self.rows.length:
            raise _hip.SkrampleHipError(f"the new schedule issued {self.rows.cursor} launches, the captured loop has {self.rows.length}: re-capture")
        self.rows.upload(slot)
        self._filled.add(slot)
        if self.static_times is not None:
            times = wrapper.timesteps.detach().to("cpu", self.static_times.dtype)
            if times.shape != self.static_times.shape:
                raise _hip.SkrampleHipError(f"the new schedule has {times.numel()} timesteps, the captured loop {self.static_times.numel()}: re-capture")
            self._times[slot] = times.clone()
            if slot == self._times_slot:
                self.static_times.copy_(self._times[slot])

    def __call__(self, latents: torch.Tensor, seeds: Sequence[int] | None = None, slot: int | None = None) -> torch.Tensor:
        self.static_in.copy_(latents)
        if slot is not None:
            if self.rows is None or not 0 <= slot < self.rows.slots:
                raise ValueError("no such schedule slot")
            if slot not in self._filled:
                raise ValueError(f"schedule slot {slot} has never been loaded: call retarget(wrapper, slot={slot}) first (its rows are all zero)")
            self.rows.index_dev.fill_(slot * self.rows.length)  # the device-resident step index: row = index + position in the loop
            if self.static_times is not None and slot != self._times_slot:
                self.static_times.copy_(self._times[slot])  # what the network reads as `t` (stream-ordered ahead of the replay)
                self._times_slot = slot
        if seeds is not None:
            if self.seeds_dev is None:
                raise ValueError("this loop draws no noise")
            from .pytorch.noise import seeds_tensor

            self.seeds_dev.copy_(seeds_tensor([int(s) & 0xFFFFFFFFFFFFFFFF for s in seeds], self.seeds_dev.device))
        self.graph.replay()
        return self.static_out.clone()


class CapturedLoops:
    """Captured sampling loops for ANY run length: `out = loops(latents, steps, seeds=None)`.

    A run's length shapes the loop itself (number of launches, the multistep ramp-up, the operands of every step), so a graph
    serves one length; the reference's loop takes `steps` per call (skrample/sampling/interface.py:34-59).  This keeps one
    `CapturedLoop` per length, recorded the first time that length is asked for (one eager warm-up + one capture, tens of
    milliseconds) and replayed afterwards; the `keep` most recently used lengths stay resident, the oldest graph and its static
    buffers are dropped beyond that.  `make_wrapper()` returns a fresh scheduler wrapper per capture (a graph reads its wrapper's
    device buffers, so lengths cannot share one); the other arguments are `capture_sampling_loop`'s."""

    def __init__(self, make_wrapper: Callable[[], object], model: Callable[[torch.Tensor, torch.Tensor], torch.Tensor], example: torch.Tensor, seeds: Sequence[int] | None = None,
                 keep: int = 8, **capture_options):
        if keep < 1:
            raise ValueError("keep at least one captured length")
        self._make, self._model, self._example, self._seeds, self._options = make_wrapper, model, example, seeds, capture_options
        self.keep = keep
        self._loops: OrderedDict[int, CapturedLoop] = OrderedDict()
        self.captures = 0  # how many graphs were recorded so far (a replay of a resident length adds none)

    def loop(self, steps: int) -> CapturedLoop:
        "the captured loop of this run length (recorded now if it is not resident)"
        steps = int(steps)
        if steps < 1:
            raise ValueError(f"a sampling loop has at least one step, not {steps}")
        found = self._loops.get(steps)
        if found is None:
            found = capture_sampling_loop(self._make(), self._model, self._example, steps, seeds=self._seeds, **self._options)
            self.captures += 1
            self._loops[steps] = found
            while len(self._loops) > self.keep:
                self._loops.popitem(last=False)
        self._loops.move_to_end(steps)
        return found

    @property
    def resident(self) -> tuple[int, ...]:
        "run lengths whose graphs are resident, least recently used first"
        return tuple(self._loops)

    def __call__(self, latents: torch.Tensor, steps: int, seeds: Sequence[int] | None = None, slot: int | None = None) -> torch.Tensor:
        return self.loop(steps)(latents, seeds, slot)


def capture_sampling_loop(wrapper, model: Callable[[torch.Tensor, torch.Tensor], torch.Tensor], example: torch.Tensor, steps: int, seeds: Sequence[int] | None = None, warmup: int = 2,
                          indexed: bool = False, slots: int = 4, device_timesteps: bool | None = None) -> CapturedLoop:
    """Capture `for t in wrapper.timesteps: x = wrapper.step(model(x, t), t, x)` for `steps` steps.

    `wrapper` is any scheduler wrapper of skrample_amd.diffusers; `model(x, t)` must be capturable (pure device
    work); `example` fixes shape/dtype/device.  Warm-up passes run eagerly first so that every step program is
    lowered and every allocation pattern is known before capture.

    `device_timesteps` (default: the value of `indexed`): `t` is a 0-d element of the scheduler's device-resident `timesteps`
    tensor -- located by the wrapper through its storage offset, no read-back -- instead of a host float; the network then
    follows a re-targeted schedule, because the tensor's contents are replaced with the rows.
    """
    from .pytorch import noise as _noise

    _noise._private_vectors[0] += 1  # the loop's seed vector is overwritten in place by replays with new seeds: it is this loop's alone
    try:
        return _capture(wrapper, model, example, steps, seeds, warmup, indexed, slots, device_timesteps)
    finally:
        _noise._private_vectors[0] -= 1


def _capture(wrapper, model, example: torch.Tensor, steps: int, seeds, warmup: int, indexed: bool, slots: int, device_timesteps) -> CapturedLoop:
    dev = example.device
    static_in = example.clone()
    gen = list(seeds) if seeds is not None else None
    if device_timesteps is None:
        device_timesteps = indexed

    wrapper.set_timesteps(steps)
    static_times = None
    if device_timesteps:
        if wrapper.timesteps.device != dev:
            wrapper.set_timesteps(steps, device=dev)
        static_times = wrapper.timesteps  # (the scheduler remembers what it handed out for as long as this tensor lives)
        times = [static_times[i] for i in range(static_times.numel())]
    else:
        times = wrapper.timesteps.tolist()

    def run(x):
        wrapper.reset_run()  # same schedule, same device seed vector; history and draw counter rewound
        for t in times:
            x = wrapper.step(model(x, t), t, x, generator=gen, return_dict=False)[0]
        return x

    def run_other(other, x):  # the same loop body on another scheduler instance (re-targeting dry run, one sample)
        other.set_timesteps(steps, device=dev) if device_timesteps else other.set_timesteps(steps)
        sub = gen[: x.shape[0]] if gen is not None else None
        ts = other.timesteps
        for t in ([ts[i] for i in range(ts.numel())] if device_timesteps else ts.tolist()):
            x = other.step(model(x, t), t, x, generator=sub, return_dict=False)[0]
        return x

    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        for _ in range(max(warmup, 1)):
            run(static_in)
    torch.cuda.current_stream(dev).wait_stream(side)
    torch.cuda.synchronize(dev)
    if hasattr(wrapper, "noise_quiesced"):
        wrapper.noise_quiesced()  # the warm-up may have drawn noise ahead on the wrapper's side stream; the device is idle now

    rows = None
    if indexed:
        from . import _hip

        rows = _hip.IndexedRows(dev, slots=slots)
        _hip.indexed = rows
        try:
            rows.begin("record")
            run(static_in)  # recording pass: one row per launch, in launch order
            torch.cuda.synchronize(dev)
            rows.finish_recording()
            rows.begin("emit")
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                static_out = run(static_in)
        finally:
            _hip.indexed = None
        if rows.cursor != rows.length:
            raise _hip.SkrampleHipError("the captured loop issued a different number of launches than the recording pass")
    else:
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            static_out = run(static_in)
    seeds_dev = getattr(getattr(wrapper, "_noise_generator", None), "_seeds", None)
    from .pytorch.noise import forget_seed_vector

    forget_seed_vector(seeds_dev)  # the graph reads this very buffer and replays may overwrite it: no other run may share it from now on
    return CapturedLoop(graph, static_in, static_out, seeds_dev, rows, run_other, static_times)
