#!/usr/bin/env python3
"""Re-create every fixture in this directory from the reference (build container only: needs /root/reference).
The generator itself lives in tools/make_golden.py (+ tools/ref_loader.py, the in-memory import hook)."""
import os
import runpy

runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tools", "make_golden.py"), run_name="__main__")
