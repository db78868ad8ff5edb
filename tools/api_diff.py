#!/usr/bin/env python3
"""Build container only: reflect over the imported reference (tools/ref_loader.py) and over skrample_amd and report every
public name, dataclass field / default and method parameter list that the reference has and this package lacks.
An empty report is the claim made in DESIGN.md section 1."""
import dataclasses
import importlib
import inspect
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [HERE, os.path.dirname(HERE)]
import ref_loader  # noqa: E402

ref_loader.install()
PAIRS = [
    ("skrample.common", "skrample_amd.common"), ("skrample.scheduling", "skrample_amd.scheduling"), ("skrample.sampling.models", "skrample_amd.sampling.models"),
    ("skrample.sampling.structured", "skrample_amd.sampling.structured"), ("skrample.sampling.functional", "skrample_amd.sampling.functional"),
    ("skrample.sampling.interface", "skrample_amd.sampling.interface"), ("skrample.sampling.traits", "skrample_amd.sampling.traits"),
    ("skrample.sampling.tableaux", "skrample_amd.sampling.tableaux"), ("skrample.pytorch.noise", "skrample_amd.pytorch.noise"), ("skrample.diffusers", "skrample_amd.diffusers"),
]
IMPORTED = {"TYPE_CHECKING", "abc", "dataclasses", "functional", "models", "scheduling", "structured", "common", "annotations", "Sequence"}


def fields(c):
    try:
        return {f.name: (f.default if f.default is not dataclasses.MISSING else "<factory-or-required>") for f in dataclasses.fields(c)}
    except TypeError:
        return None


def params(f):
    try:
        return list(inspect.signature(f).parameters)
    except (TypeError, ValueError):
        return None


problems = 0
for ref_name, mine_name in PAIRS:
    R, M = importlib.import_module(ref_name), importlib.import_module(mine_name)
    for name in dir(R):
        if name.startswith("_") or name in IMPORTED:
            continue
        obj = getattr(R, name)
        if inspect.ismodule(obj) and not hasattr(M, name):
            continue
        if getattr(obj, "__module__", ref_name).split(".")[0] != "skrample":
            continue
        if not hasattr(M, name):
            print(f"{ref_name}.{name}: missing")
            problems += 1
            continue
        mine = getattr(M, name)
        if inspect.isclass(obj):
            fr, fm = fields(obj), fields(mine)
            if fr is not None:
                for k, v in fr.items():
                    if fm is None or k not in fm:
                        print(f"{ref_name}.{name}: field {k} missing")
                        problems += 1
                    elif isinstance(v, (int, float, str, bool, tuple, type(None))) and v != fm[k]:
                        print(f"{ref_name}.{name}.{k}: default {v!r} vs {fm[k]!r}")
                        problems += 1
            for attr in dir(obj):
                if attr.startswith("_"):
                    continue
                if not hasattr(mine, attr):
                    print(f"{ref_name}.{name}.{attr}: missing")
                    problems += 1
                    continue
                ra, ma = getattr(obj, attr, None), getattr(mine, attr, None)
                if callable(ra) and callable(ma) and not inspect.isclass(ra):
                    pr, pm = params(ra), params(ma)
                    if pr and pm and not set(pr) <= set(pm) and pr != ["args", "kwds"]:
                        print(f"{ref_name}.{name}.{attr}: parameters {pr} vs {pm}")
                        problems += 1
        elif callable(obj):
            pr, pm = params(obj), params(mine)
            if pr and pm and pr != pm:
                print(f"{ref_name}.{name}: parameters {pr} vs {pm}")
                problems += 1
print("differences:", problems)
sys.exit(1 if problems else 0)
