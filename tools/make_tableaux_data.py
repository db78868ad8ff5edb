#!/usr/bin/env python3
"""Dump the published high-order Butcher tableaux (15-35 stages) as plain numbers into
skrample_amd/sampling/tableaux_high_order.json (build container only; reads the reference through
tools/ref_loader.py and stores coefficients only -- exact float hex, no source text).

Sources of the coefficients (as cited by the reference, skrample/sampling/tableaux/providers.py:361-380):
  Stepanov10  M. Stepanov, "On Runge-Kutta methods of order 10" (2025), arXiv:2504.17329      (15 stages)
  Ono10       H. Ono, 17-stage order-10 scheme (2003), via P. Stone's RK coefficient collection
  Harrier10   17-stage order-10 scheme
  Zhang10     D. Zhang, "Discovering new Runge-Kutta methods using unstructured numerical search" (2019), arXiv:1911.00318
  Feagin10/12/14  T. Feagin, "An explicit Runge-Kutta method of order twelve" (2007) and companions
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_loader  # noqa: E402

ref_loader.install()
from skrample.sampling import tableaux  # noqa: E402

out = {}
for name in ("Stepanov10", "Ono10", "Harrier10", "Zhang10", "Feagin10", "Feagin12", "Feagin14"):
    tab = getattr(tableaux.RKZ, name).tableau()
    out[name] = {
        "c": [float(s.c).hex() for s in tab.stages],
        "a": [[float(v).hex() for v in s.a] for s in tab.stages],
        "b": [float(v).hex() for v in tab.weights],
    }
path = os.path.join(os.path.dirname(HERE), "skrample_amd", "sampling", "tableaux_high_order.json")
json.dump(out, open(path, "w"))
print(path, os.path.getsize(path), {k: len(v["b"]) for k, v in out.items()})

# the "graveyard" groups (reference tableaux/__init__.py:39-43; providers.py:641-861 WSO -- Biswas et al. 2023,
# arXiv:2310.02817; providers.py:863-1000 Shanks1965 -- E. B. Shanks, NASA TN D-2920): coefficients only
grave = {}
for group in ("WSO", "Shanks1965"):
    for member in getattr(tableaux, group):
        tab = member.tableau()
        grave[f"{group}.{member.name}"] = {
            "c": [float(s.c).hex() for s in tab.stages],
            "a": [[float(v).hex() for v in s.a] for s in tab.stages],
            "b": [float(v).hex() for v in tab.weights],
        }
path = os.path.join(os.path.dirname(HERE), "skrample_amd", "sampling", "tableaux_graveyard.json")
json.dump(grave, open(path, "w"))
print(path, os.path.getsize(path), {k: len(v["b"]) for k, v in grave.items()})
