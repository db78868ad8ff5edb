"""Whole-loop capture: record an N-step sampling loop (network + fused sampler steps) into one HIP graph.

Every launch of this engine is capture-safe by construction (`skr_step_launch` only enqueues a kernel on the
caller's stream: no allocation, no synchronisation, no host read-back; per-sample seeds and all operands are
read from device memory at run time), so a sampling loop whose network is itself capturable can be recorded
with `torch.cuda.graph` and replayed with new inputs / new seeds at ~10 us of host cost for the whole loop.
This removes the per-step Python + launch overhead that dominates small batches.
"""

from __future__ import annotations

from typing import Callable, Sequence

import torch


class CapturedLoop:
    "replayable sampling loop: `out = loop(initial_latents, seeds=None)`"

    def __init__(self, graph: torch.cuda.CUDAGraph, static_in: torch.Tensor, static_out: torch.Tensor, seeds_dev: torch.Tensor | None):
        self.graph, self.static_in, self.static_out, self.seeds_dev = graph, static_in, static_out, seeds_dev

    def __call__(self, latents: torch.Tensor, seeds: Sequence[int] | None = None) -> torch.Tensor:
        self.static_in.copy_(latents)
        if seeds is not None:
            if self.seeds_dev is None:
                raise ValueError("this loop draws no noise")
            from .pytorch.noise import seeds_tensor

            self.seeds_dev.copy_(seeds_tensor([int(s) & 0xFFFFFFFFFFFFFFFF for s in seeds], self.seeds_dev.device))
        self.graph.replay()
        return self.static_out.clone()


def capture_sampling_loop(wrapper, model: Callable[[torch.Tensor, torch.Tensor], torch.Tensor], example: torch.Tensor, steps: int, seeds: Sequence[int] | None = None, warmup: int = 2) -> CapturedLoop:
    """Capture `for t in wrapper.timesteps: x = wrapper.step(model(x, t), t, x)` for `steps` steps.

    `wrapper` is any scheduler wrapper of skrample_amd.diffusers; `model(x, t)` must be capturable (pure device
    work); `example` fixes shape/dtype/device.  Warm-up passes run eagerly first so that every step program is
    lowered and every allocation pattern is known before capture.
    """
    dev = example.device
    static_in = example.clone()
    gen = list(seeds) if seeds is not None else None

    wrapper.set_timesteps(steps)
    times = wrapper.timesteps.tolist()

    def run(x):
        wrapper.reset_run()  # same schedule, same device seed vector; history and draw counter rewound
        for t in times:
            x = wrapper.step(model(x, t), t, x, generator=gen, return_dict=False)[0]
        return x

    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        for _ in range(max(warmup, 1)):
            run(static_in)
    torch.cuda.current_stream(dev).wait_stream(side)
    torch.cuda.synchronize(dev)

    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        static_out = run(static_in)
    seeds_dev = getattr(getattr(wrapper, "_noise_generator", None), "_seeds", None)
    return CapturedLoop(graph, static_in, static_out, seeds_dev)
