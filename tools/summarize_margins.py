#!/usr/bin/env python3
"""Condense the measured parity margins of one GPU test run into profiles/r05_parity_margins.txt.

    SKR_PARITY_MARGINS=/tmp/margins.jsonl python -m pytest tests -m gpu -q      (every parity comparison appends what it measured)
    python tools/summarize_margins.py /tmp/margins.jsonl profiles/r05_parity_margins.txt

One line per (family, measure): how many comparisons, the median, the 99th percentile and the maximum that was MEASURED, the bar the
test asserts, and the test that produced the maximum -- the record the tolerances written in tests/ are justified against."""
import collections
import json
import sys


def main(src: str, dst: str) -> None:
    groups: dict = collections.defaultdict(list)
    for line in open(src):
        r = json.loads(line)
        groups[(r["family"], r["measure"], r["bar"])].append((r["value"], r["test"]))
    lines = [
        "# measured parity margins of `SKR_PARITY_MARGINS=... pytest tests -m gpu` on one MI355X (tools/collect_r05.sh margins; tools/summarize_margins.py)",
        "# family | measure | comparisons | median | p99 | MAX | bar asserted | share of the bar used by the max | test of the max",
    ]
    for (family, measure, bar), vals in sorted(groups.items(), key=lambda kv: (kv[0][0], kv[0][1])):
        vs = sorted(v for v, _ in vals)
        top = max(vals, key=lambda t: t[0])
        med, p99 = vs[len(vs) // 2], vs[min(len(vs) - 1, int(0.99 * len(vs)))]
        used = "-" if not bar else f"{top[0] / bar:.2f}"
        lines.append(f"{family} | {measure} | {len(vs)} | {med:.3g} | {p99:.3g} | {top[0]:.3g} | {bar if bar is not None else '-'} | {used} | {top[1].split('::', 1)[-1][:110]}")
    open(dst, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
