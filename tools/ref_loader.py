"""Container-only loader that imports the read-only reference (/root/reference, Python >= 3.12
syntax) under this image's Python 3.10 so that golden fixtures can be generated from the
reference's own arithmetic.

Nothing from the reference is stored: source text is read, rewritten in memory (PEP 695 syntax
de-sugared) and exec'd.  Only the numbers it produces are written to tests/golden/ by
tools/make_golden.py.  This module never travels to the GPU box in a usable form
(/root/reference does not exist there) and is imported by nothing under skrample_amd/.

Rewrite rules:
  1. ``type X[...] = ...``        -> ``X = typing.Any``  (statement may span lines)
  2. ``def name[...](``           -> ``def name(``
  3. ``class Name[...](Base[...])`` -> every ``[...]`` group in the class header removed
  4. ``from __future__ import annotations`` prepended
Shims: math.sumprod (3.12), enum.StrEnum (3.11), typing.Self (3.11).
"""

from __future__ import annotations

import enum
import importlib.abc
import importlib.util
import math
import os
import re
import sys
import typing
from fractions import Fraction

REFERENCE_ROOT = os.environ.get("SKR_REFERENCE_ROOT", "/root/reference")


def _sumprod(p, q):
    """CPython 3.12 math.sumprod: exact (once-rounded) for all-float operands,
    otherwise ``total = 0; total = total + p_i * q_i`` left to right."""
    p, q = list(p), list(q)
    if len(p) != len(q):
        raise ValueError("Inputs are not the same length")
    if all(type(x) in (int, float) for x in p) and all(type(x) in (int, float) for x in q):
        if all(math.isfinite(x) for x in (*p, *q)):
            return float(sum((Fraction(a) * Fraction(b) for a, b in zip(p, q)), Fraction(0)))
        return math.fsum(a * b for a, b in zip(p, q))
    total = 0
    for a, b in zip(p, q):
        total = total + a * b
    return total


def install_shims() -> None:
    if not hasattr(math, "sumprod"):
        math.sumprod = _sumprod  # type: ignore[attr-defined]
    if not hasattr(enum, "StrEnum"):

        class StrEnum(str, enum.Enum):
            def __str__(self) -> str:
                return str(self.value)

            @staticmethod
            def _generate_next_value_(name, start, count, last_values):  # noqa: ARG004
                return name.lower()

        enum.StrEnum = StrEnum  # type: ignore[attr-defined]
    if not hasattr(typing, "Self"):
        typing.Self = typing.Any  # type: ignore[attr-defined]


def _strip_brackets(text: str, start: int) -> tuple[str, int]:
    """Remove the balanced ``[...]`` group that opens at text[start]; returns (new_text, pos)."""
    depth = 0
    i = start
    while i < len(text):
        if text[i] == "[":
            depth += 1
        elif text[i] == "]":
            depth -= 1
            if depth == 0:
                return text[:start] + text[i + 1 :], start
        i += 1
    raise SyntaxError("unbalanced brackets")


def desugar(source: str) -> str:
    out_lines: list[str] = []
    lines = source.split("\n")
    i = 0
    while i < len(lines):
        line = lines[i]
        m = re.match(r"^(\s*)type\s+([A-Za-z_]\w*)(\[.*?\])?\s*=", line)
        if m:
            # swallow continuation lines until brackets/parens balance
            stmt = line
            while stmt.count("(") + stmt.count("[") > stmt.count(")") + stmt.count("]"):
                i += 1
                stmt += "\n" + lines[i]
            out_lines.append(f"{m.group(1)}{m.group(2)} = __import__('typing').Any")
            i += 1
            continue
        m = re.match(r"^(\s*)def\s+([A-Za-z_]\w*)\[", line)
        if m:
            pos = m.end() - 1
            line, _ = _strip_brackets(line, pos)
        m = re.match(r"^(\s*)class\s+([A-Za-z_]\w*)", line)
        if m and line.rstrip().endswith(":"):
            head = line
            while "[" in head:
                head, _ = _strip_brackets(head, head.index("["))
            line = head
        out_lines.append(line)
        i += 1
    return "from __future__ import annotations\n" + "\n".join(out_lines)


class _RefFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def __init__(self, root: str) -> None:
        self.root = root

    def _path(self, fullname: str) -> tuple[str, bool] | None:
        rel = fullname.replace(".", "/")
        pkg = os.path.join(self.root, rel, "__init__.py")
        mod = os.path.join(self.root, rel + ".py")
        if os.path.isfile(pkg):
            return pkg, True
        if os.path.isfile(mod):
            return mod, False
        if os.path.isdir(os.path.join(self.root, rel)):
            return os.path.join(self.root, rel), True  # namespace-ish package without __init__
        return None

    def find_spec(self, fullname, path=None, target=None):  # noqa: ARG002
        if fullname != "skrample" and not fullname.startswith("skrample."):
            return None
        found = self._path(fullname)
        if found is None:
            return None
        p, is_pkg = found
        spec = importlib.util.spec_from_loader(fullname, self, is_package=is_pkg)
        spec.origin = p
        if is_pkg:
            spec.submodule_search_locations = [p if os.path.isdir(p) else os.path.dirname(p)]
        return spec

    def create_module(self, spec):  # noqa: ARG002
        return None

    def exec_module(self, module) -> None:
        origin = module.__spec__.origin
        if os.path.isdir(origin):
            return
        with open(origin, encoding="utf-8") as fh:
            text = desugar(fh.read())
        module.__file__ = origin
        exec(compile(text, origin, "exec"), module.__dict__)  # noqa: S102


def install(root: str = REFERENCE_ROOT) -> None:
    if not os.path.isdir(os.path.join(root, "skrample")):
        raise FileNotFoundError(f"reference not present at {root} (it exists only in the build container)")
    install_shims()
    if not any(isinstance(f, _RefFinder) for f in sys.meta_path):
        sys.meta_path.insert(0, _RefFinder(root))


if __name__ == "__main__":
    install()
    import skrample.common as c  # noqa: PLC0415

    print("bashforth(4) =", c.bashforth(4))
