"""Butcher tableaux for the explicit Runge-Kutta samplers (host-side constants only: the kernel
sees a row of <= 35 weights).

Names follow reference `skrample/sampling/tableaux/` (common.py: Stage/Tableau/EmbeddedTableau :7-24,
validate_tableau :145-156; providers.py: generator functions :15-127, provider enums :173-958) so that
`tableaux.RKE5.CashKarp`, `tableaux.RK2.Mid`, ... resolve as they do there.  The coefficients
themselves are the published ones (citations on each family); they are written here as exact
rationals / surds and evaluated once at import.

The 10th-14th order tables with 15-35 stages (RKZ.Stepanov10, Ono10, Harrier10, Zhang10, Feagin10/12/14) are
pure data: their published coefficients live in `tableaux_high_order.json` (exact float hex; produced by
tools/make_tableaux_data.py, which also lists the papers they come from).
"""

from __future__ import annotations

import dataclasses
import enum
import json
import math
import os
from fractions import Fraction
from typing import NamedTuple, Protocol, Sequence


class Stage(NamedTuple):
    c: float
    a: tuple[float, ...]


class Tableau(NamedTuple):
    stages: tuple[Stage, ...]
    weights: tuple[float, ...]


class EmbeddedTableau(NamedTuple):
    stages: tuple[Stage, ...]
    weights: tuple[float, ...]
    error_weights: tuple[float, ...]

    def unembed(self) -> Tableau:
        return Tableau(self.stages, self.weights)


TableauType = "Tableau | EmbeddedTableau"


def validate_tableau(tab, tolerance: float = 1e-12) -> Exception | None:
    "row i has i coefficients summing to c_i; every weight row has one weight per stage and sums to 1"
    for i, stage in enumerate(tab.stages):
        if len(stage.a) != i:
            return IndexError(f"stage {i} has {len(stage.a)} coefficients: {stage}")
        if abs(stage.c - math.fsum(stage.a)) > tolerance:
            return ValueError(f"stage {i}: c={stage.c} but row sums to {math.fsum(stage.a)}")
    for row in tab[1:]:
        if len(row) != len(tab.stages):
            return IndexError(f"{len(row)} weights for {len(tab.stages)} stages")
        if abs(1 - math.fsum(row)) > tolerance:
            return ValueError(f"weights sum to {math.fsum(row)}")
    return None


def serialize(tab) -> list[float]:
    "flat [c..., a row by row..., b...] (what the reference's ButcherCoeffs.serialize yields)"
    return [*(s.c for s in tab.stages), *(v for s in tab.stages for v in s.a), *tab.weights]


def _q(text: str) -> float:
    "exact rational 'p/q' -> correctly rounded float (identical to python's p / q on ints)"
    return float(Fraction(text))


def _rational(rows: Sequence[str], *weight_rows: str):
    """build a tableau from strings of rationals; row i = 'a_i0 a_i1 ...' (c_i is the row sum as
    published, recomputed exactly in rational arithmetic)"""
    stages = []
    for row in rows:
        coeffs = [Fraction(tok) for tok in row.split()]
        stages.append(Stage(float(sum(coeffs, Fraction(0))) if coeffs else 0, tuple(float(c) for c in coeffs)))
    weights = [tuple(float(Fraction(tok)) for tok in w.split()) for w in weight_rows]
    if len(weights) == 1:
        return Tableau(tuple(stages), weights[0])
    return EmbeddedTableau(tuple(stages), weights[0], weights[1])


# ---------------------------------------------------------------------------------------------------
# parametric families
# ---------------------------------------------------------------------------------------------------
def rk2_tableau(c1: float) -> Tableau:
    "all 2-stage order-2 methods: b = (1 - 1/(2c), 1/(2c))"
    return Tableau((Stage(0.0, ()), Stage(c1, (c1,))), (1 - 1 / (2 * c1), 1 / (2 * c1)))


def rk3_tableau(c1: float, c2: float) -> Tableau:
    "generic 3-stage order-3 family in its two nodes"
    return Tableau(
        (
            Stage(0.0, ()),
            Stage(c1, (c1,)),
            Stage(c2, (c2 / c1 * ((c2 - 3 * c1 * (1 - c1)) / (3 * c1 - 2)), -c2 / c1 * ((c2 - c1) / (3 * c1 - 2)))),
        ),
        (1 - (3 * c1 + 3 * c2 - 2) / (6 * c1 * c2), (3 * c2 - 2) / (6 * c1 * (c2 - c1)), (2 - 3 * c1) / (6 * c2 * (c2 - c1))),
    )


def rk4_tableau(c1: float, c2: float) -> Tableau:
    "generic 4-stage order-4 family with c3 = 1 (not defined at the classic c1 = c2 = 1/2)"
    det = 6 * c1 * c2 - 4 * (c1 + c2) + 3
    b2 = (2 * c2 - 1) / (12 * c1 * (c2 - c1) * (1 - c1))
    b3 = (2 * c1 - 1) / (12 * c2 * (c1 - c2) * (1 - c2))
    b4 = det / (12 * (1 - c1) * (1 - c2))
    a32 = c2 * (c1 - c2) / (2 * c1 * (2 * c1 - 1))
    a42 = ((4 * c2**2 - 5 * c2 - c1 + 2) * (1 - c1)) / (2 * c1 * (c1 - c2) * det)
    a43 = ((2 * c1 - 1) * (1 - c1) * (1 - c2)) / (c2 * (c1 - c2) * det)
    return Tableau(
        (Stage(0.0, ()), Stage(c1, (c1,)), Stage(c2, (c2 - a32, a32)), Stage(1.0, (1 - a42 - a43, a42, a43))),
        (1 - b2 - b3 - b4, b2, b3, b4),
    )


def ees25_tableau(x: float) -> Tableau:
    "EES(2,5;x): explicit & effectively symmetric, 3 stages (arXiv 2507.21006)"
    c1 = (1 + 2 * x) / (4 * (1 - x))
    return Tableau(
        (
            Stage(0.0, ()),
            Stage(c1, (c1,)),
            Stage(3 / (4 * (1 - x)), ((4 * x - 1) ** 2 / (4 * (x - 1) * (1 - 4 * x**2)), (1 - x) / (1 - 4 * x**2))),
        ),
        (x, 1 / 2, 1 / 2 - x),
    )


def ees27_tableau(x: float) -> Tableau:
    "EES(2,7;x): 4 stages (arXiv 2507.21006; a42 as printed in its tableau (8.6))"
    s2 = math.sqrt(2)
    big_a = (2 * x + s2) / ((2 * x - 1) * (-2 * x - s2 + 1))
    big_b = 1 / ((2 * x - 1) * (1 - s2 - 2 * x) * (2 - s2 - 2 * x))
    row2 = ((-2 + s2 * (1 - 2 * x)) / (4 * (x - 1)),)
    row3 = ((((2 * x + s2 - 2) * (4 * x + s2 - 2)) / (4 * s2 * (x - 1))) * big_a, (0.5 * (-1 + s2)) * big_a)
    row4 = (
        ((2 * x - s2) * (-40 * x**4 + (80 - 40 * s2) * x**3 - (88 - 60 * s2) * x**2 + (48 - 34 * s2) * x + 7 * s2 - 10))
        / (4 * (x - 1) * (2 * x**2 - 1))
        * big_b,
        (2 - s2) * x * (x - 1) * (4 * x + s2 - 2) * big_b,
        ((2 - s2) * (2 * x - s2) * (2 + s2 - 2 * x) * (x - 1) * (2 * x - 1)) / (4 * (2 * x**2 - 1) * (2 * x**2 - 4 * x + 1)),
    )
    return Tableau(
        (Stage(0.0, ()), Stage(math.fsum(row2), row2), Stage(math.fsum(row3), row3), Stage(math.fsum(row4), row4)),
        (x, 1 / 2 * (2 - s2) - (1 - s2) * x, (1 - s2) * (x - 1), 1 / 2 * (2 - s2) - x),
    )


def shu_osher_tableau(alphas: Sequence[Sequence[float]], betas: Sequence[Sequence[float]]) -> Tableau:
    """Shu-Osher form (u_i = sum_k alpha_ik u_k + dt beta_ik F(u_k)) -> Butcher form:
    a_ij = beta_(i-1)j + sum_{k>j} alpha_(i-1)k a_kj, the last row giving b."""
    s = len(alphas)
    a = [[0.0] * n for n in range(s)]

    def row(i: int, j: int) -> float:
        return math.fsum((betas[i][j], *(alphas[i][k] * a[k][j] for k in range(j + 1, i + 1))))

    for i in range(1, s):
        for j in range(i):
            a[i][j] = row(i - 1, j)
    b = [row(s - 1, j) for j in range(s)]
    return Tableau(tuple(Stage(math.fsum(r), tuple(r)) for r in a), tuple(b))


@dataclasses.dataclass(frozen=True)
class ButcherCoeffs:
    """Mutable scratch form of a tableau: nodes c, strictly lower-triangular rows a (row n has n entries) and weights b,
    optionally 1-indexed with a dummy leading entry (papers that number stages from 1).  Interface of the reference's
    tableaux/common.py:30-125 (empty / compute_c / compose / decompose / deserialize / serialize / from_shu_osher)."""

    one_index: bool
    c: list
    a: list
    b: list

    @classmethod
    def empty(cls, stages: int, fill: float = -math.inf, one_index: bool = False) -> "ButcherCoeffs":
        n = stages + int(one_index)
        c = [fill] * n
        c[int(one_index)] = 0  # the first stage sits at the step start
        return cls(one_index, c=c, a=[[fill] * k for k in range(n)], b=[fill] * n)

    def compute_c(self) -> None:
        "row-sum condition: c_i = sum_j a_ij"
        self.c[:] = [math.fsum(row) for row in self.a]

    def compose(self) -> Tableau:
        k = int(self.one_index)
        return Tableau(tuple(Stage(c, tuple(row[k:])) for c, row in zip(self.c[k:], self.a[k:], strict=True)), tuple(self.b[k:]))

    @classmethod
    def decompose(cls, tableau: Tableau) -> "ButcherCoeffs":
        return cls(False, c=[st.c for st in tableau.stages], a=[list(st.a) for st in tableau.stages], b=list(tableau.weights))

    @classmethod
    def deserialize(cls, coeffs: Sequence[float], stages: int, compute_c: bool = False, b_last: bool = True) -> "ButcherCoeffs":
        "flat list: [c (unless compute_c)] [b if not b_last] [a rows 1..] [b if b_last]"
        t = cls.empty(stages)
        n_a = sum(len(row) for row in t.a)
        if len(coeffs) != (0 if compute_c else stages) + stages + n_a:
            raise AssertionError(f"{len(coeffs)} coefficients for {stages} stages")
        it = iter(coeffs)
        if not compute_c:
            t.c[:] = [next(it) for _ in range(stages)]
        if not b_last:
            t.b[:] = [next(it) for _ in range(stages)]
        for row in t.a[1:]:
            row[:] = [next(it) for _ in row]
        if compute_c:
            t.compute_c()
        if b_last:
            t.b[:] = [next(it) for _ in range(stages)]
        return t

    def serialize(self) -> list:
        return [*self.c, *(v for row in self.a for v in row), *self.b]

    @classmethod
    def from_shu_osher(cls, alphas: Sequence[Sequence[float]], betas: Sequence[Sequence[float]]) -> "ButcherCoeffs":
        return cls.decompose(shu_osher_tableau(alphas, betas))


# ---------------------------------------------------------------------------------------------------
# providers
# ---------------------------------------------------------------------------------------------------
class TableauProvider(Protocol):
    def tableau(self): ...

    def pretty(self) -> str:
        return pretty_tableau(self.tableau())


class _Pretty:
    "mix-in for the dataclass providers: the text rendering every provider offers"

    def pretty(self) -> str:
        return pretty_tableau(self.tableau())


class _EnumProvider(enum.Enum):
    def tableau(self):
        return self.value

    def pretty(self) -> str:
        return pretty_tableau(self.value, str(self))


@dataclasses.dataclass(frozen=True)
class CustomTableau(_Pretty):
    custom: object

    def tableau(self):
        return self.custom


@dataclasses.dataclass(frozen=True)
class RK2Custom(_Pretty):
    c1: float = 1.0

    def tableau(self) -> Tableau:
        return rk2_tableau(self.c1)


@dataclasses.dataclass(frozen=True)
class RK3Custom(_Pretty):
    c1: float = 1 / 2
    c2: float = 1.0

    def tableau(self) -> Tableau:
        return rk3_tableau(self.c1, self.c2)


@dataclasses.dataclass(frozen=True)
class RK4Custom(_Pretty):
    c1: float = 1 / 3
    c2: float = 2 / 3

    def tableau(self) -> Tableau:
        return rk4_tableau(self.c1, self.c2)


_S2, _S5, _S21 = math.sqrt(2), math.sqrt(5), math.sqrt(21)


@enum.unique
class RK1(_EnumProvider):
    Euler = Tableau((Stage(0, ()),), (1,))


@enum.unique
class RK2(_EnumProvider):
    Mid = rk2_tableau(1 / 2)
    Ralston = rk2_tableau(2 / 3)
    Golden = rk2_tableau((1 + _S5) / 4)
    EES5_SYM = ees25_tableau(1 / 4)
    EES5_MIN = ees25_tableau(1 / 10)
    EES7_SYM = ees27_tableau(1 / 4 * (2 - _S2))
    EES7_MIN = ees27_tableau(1 / 14 * (5 - 3 * _S2))


@enum.unique
class RK3(_EnumProvider):
    Kutta = rk3_tableau(1 / 2, 1)
    Heun = rk3_tableau(1 / 3, 2 / 3)
    Ralston = rk3_tableau(1 / 2, 3 / 4)  # Ralston 1962, minimum error bound
    Wray = rk3_tableau(8 / 15, 2 / 3)


@enum.unique
class RK4(_EnumProvider):
    Kutta = _rational(["", "1/2", "0 1/2", "0 0 1"], "1/6 1/3 1/3 1/6")
    Eighth = rk4_tableau(1 / 3, 2 / 3)
    Ralston = rk4_tableau(2 / 5, (14 - 3 * _S5) / 16)


def _butcher6() -> Tableau:
    "Butcher 1964, 'On Runge-Kutta processes of high order', fig. 15: 7 stages, order 6"
    lo, hi = (5 - _S5) / 10, (5 + _S5) / 10
    return Tableau(
        (
            Stage(0, ()),
            Stage(lo, (lo,)),
            Stage(hi, (-_S5 / 10, (5 + 2 * _S5) / 10)),
            Stage(lo, ((-15 + 7 * _S5) / 20, (-1 + _S5) / 4, (15 - 7 * _S5) / 10)),
            Stage(hi, ((5 - _S5) / 60, 0, 1 / 6, (15 + 7 * _S5) / 60)),
            Stage(lo, ((5 + _S5) / 60, 0, (9 - 5 * _S5) / 12, 1 / 6, (-5 + 3 * _S5) / 10)),
            Stage(1.0, (1 / 6, 0, (-55 + 25 * _S5) / 12, (-25 - 7 * _S5) / 12, 5 - 2 * _S5, (5 + _S5) / 2)),
        ),
        (1 / 12, 0, 0, 0, 5 / 12, 5 / 12, 1 / 12),
    )


def _frac(text: str) -> float:
    num, _, den = text.partition("/")
    return int(num) / int(den) if den else float(int(num))


def _surd21(text: str) -> float:
    "'p/q' or 'p/q+r/s' / 'p/q-r/s' meaning p/q +- (r/s)*sqrt(21)"
    cut = max(text.rfind("+"), text.rfind("-"))
    if cut <= 0:  # no surd part (a leading '-' is the sign of the rational)
        return _frac(text)
    rational, surd = text[:cut], text[cut + 1 :]
    term = _frac(surd) * _S21
    return _frac(rational) + term if text[cut] == "+" else _frac(rational) - term


def _cv8() -> Tableau:
    "Cooper & Verner 1972, 11 stages, order 8 (coefficients p/q +- (r/s) sqrt(21))"
    rows = [
        "",
        "1/2",
        "1/4 1/4",
        "1/7 -1/14-3/98 3/7+5/49",
        "11/84+1/84 0 2/7+4/63 1/12-1/252",
        "5/48+1/48 0 1/4+1/36 -77/120+7/180 63/80-7/80",
        "5/21-1/42 0 -48/35+92/315 211/30-29/18 -36/5+23/14 9/5-13/35",
        "1/14 0 0 0 1/9-1/42 13/63-1/21 1/9",
        "1/32 0 0 0 91/576-7/192 11/72 -385/1152-25/384 63/128+13/128",
        "1/14 0 0 0 1/9 -733/2205-1/15 515/504+37/168 -51/56-11/56 132/245+4/35",
        "0 0 0 0 -7/3+7/18 -2/5+28/45 -91/24-53/72 301/72+53/72 28/45-28/45 49/18-7/18",
    ]
    nodes = ["0", "1/2", "1/2", "1/2+1/14", "1/2+1/14", "1/2", "1/2-1/14", "1/2-1/14", "1/2", "1/2+1/14", "1"]
    stages = tuple(Stage(_surd21(c), tuple(_surd21(tok) for tok in row.split())) for c, row in zip(nodes, rows))
    return Tableau(stages, (1 / 20, 0, 0, 0, 0, 0, 0, 49 / 180, 16 / 45, 49 / 180, 1 / 20))


def _high_order() -> dict[str, Tableau]:
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tableaux_high_order.json")
    with open(path) as fh:
        raw = json.load(fh)
    h = float.fromhex
    return {
        name: Tableau(tuple(Stage(h(c), tuple(h(v) for v in row)) for c, row in zip(t["c"], t["a"])), tuple(h(v) for v in t["b"]))
        for name, t in raw.items()
    }


_HIGH = _high_order()


class RKZ(_EnumProvider):
    "methods without a clean generic form that need more stages than their order"

    Nystrom5 = _rational(["", "1/3", "4/25 6/25", "1/4 -3 15/4", "2/27 10/9 -50/81 8/81", "2/25 12/25 2/15 8/75 0"], "23/192 0 125/192 0 -27/64 125/192")
    Butcher6 = _butcher6()
    CV8 = _cv8()
    Stepanov10 = _HIGH["Stepanov10"]  # arXiv 2504.17329, 15 stages
    Ono10 = _HIGH["Ono10"]  # H. Ono 2003, 17 stages
    Harrier10 = _HIGH["Harrier10"]
    Zhang10 = _HIGH["Zhang10"]  # arXiv 1911.00318, 16 stages
    Feagin10 = _HIGH["Feagin10"]
    Feagin12 = _HIGH["Feagin12"]  # T. Feagin 2007, 25 stages
    Feagin14 = _HIGH["Feagin14"]  # 35 stages


@enum.unique
class RKE2(_EnumProvider):
    Heun = _rational(["", "1"], "1/2 1/2", "1 0")
    Fehlberg = _rational(["", "1/2", "1/256 255/256"], "1/512 255/256 1/512", "1/256 255/256 0")


@enum.unique
class RKE3(_EnumProvider):
    BogackiShampine = _rational(["", "1/2", "0 3/4", "2/9 1/3 4/9"], "2/9 1/3 4/9 0", "7/24 1/4 1/3 1/8")
    SSPRK3_4 = _rational(["", "1/2", "1/2 1/2", "1/6 1/6 1/6"], "1/6 1/6 1/6 1/2", "1/4 1/4 1/4 1/4")  # arXiv 2104.06836


@enum.unique
class RKE5(_EnumProvider):
    Fehlberg = _rational(
        ["", "1/4", "3/32 9/32", "1932/2197 -7200/2197 7296/2197", "439/216 -8 3680/513 -845/4104", "-8/27 2 -3544/2565 1859/4104 -11/40"],
        "16/135 0 6656/12825 28561/56430 -9/50 2/55",
        "25/216 0 1408/2565 2197/4104 -1/5 0",
    )
    CashKarp = _rational(
        ["", "1/5", "3/40 9/40", "3/10 -9/10 6/5", "-11/54 5/2 -70/27 35/27", "1631/55296 175/512 575/13824 44275/110592 253/4096"],
        "37/378 0 250/621 125/594 0 512/1771",
        "2825/27648 0 18575/48384 13525/55296 277/14336 1/4",
    )
    DormandPrince = _rational(
        [
            "",
            "1/5",
            "3/40 9/40",
            "44/45 -56/15 32/9",
            "19372/6561 -25360/2187 64448/6561 -212/729",
            "9017/3168 -355/33 46732/5247 49/176 -5103/18656",
            "35/384 0 500/1113 125/192 -2187/6784 11/84",
        ],
        "35/384 0 500/1113 125/192 -2187/6784 11/84 0",
        "5179/57600 0 7571/16695 393/640 -92097/339200 187/2100 1/40",
    )


def _ssp(alpha_text: str, beta_text: str) -> Tableau:
    "Ruuth 2006 optimal SSP methods, given in Shu-Osher form; ';' separates rows"
    parse = lambda text: [[float(tok) for tok in row.split()] for row in text.strip().split(";")]  # noqa: E731
    return shu_osher_tableau(parse(alpha_text), parse(beta_text))


@enum.unique
class SSP(_EnumProvider):
    "Ruuth, 'Global optimization of explicit strong-stability-preserving Runge-Kutta methods' (2006)"

    RK3_3 = rk3_tableau(1, 1 / 2)
    RK3_5 = _ssp(
        "1; 0 1; 0.355909775063327 0 0.644090224936674; 0.367933791638137 0 0 0.632066208361863; 0 0 0.237593836598569 0 0.762406163401431",
        "0.377268915331368; 0 0.377268915331368; 0 0 0.242995220537396; 0 0 0 0.238458932846290; 0 0 0 0 0.287632146308408",
    )
    RK3_6 = _ssp(
        "1; 0 1; 0 0 1; 0.476769811285196 0.098511733286064 0 0.424718455428740; 0 0 0 0 1; 0 0 0.155221702560091 0 0 0.844778297439909",
        "0.284220721334261; 0 0.284220721334261; 0 0 0.284220721334261; 0 0 0 0.120713785765930; 0 0 0 0 0.284220721334261; 0 0 0 0 0 0.240103497065900",
    )
    RK3_7 = _ssp(
        "1; 0 1; 0 0 1; 0.184962588071072 0 0 0.815037411928928; 0.180718656570380 0.314831034403793 0 0 0.504450309025826; 0 0 0 0 0 1;"
        " 0 0 0 0.120199000000000 0 0 0.879801000000000",
        "0.233213863663009; 0 0.233213863663009; 0 0 0.233213863663009; 0 0 0 0.190078023865845; 0 0 0 0 0.117644805593912;"
        " 0 0 0 0 0 0.233213863663009; 0 0 0 0 0 0 0.205181790464579",
    )
    RK3_8 = _ssp(
        "1; 0 1; 0 0 1; 0 0 0 1; 0.421366967085359 0.005949401107575 0 0 0.572683631807067; 0 0.004254010666365 0 0 0 0.995745989333635;"
        " 0 0 0.104380143093325 0.243265240906726 0 0 0.652354615999950; 0 0 0 0 0 0 0 1",
        "0.195804015330143; 0 0.195804015330143; 0 0 0.195804015330143; 0 0 0 0.195804015330143; 0 0 0 0 0.112133754621673;"
        " 0 0 0 0 0 0.194971062960412; 0 0 0 0 0 0 0.127733653231944; 0 0 0 0 0 0 0 0.195804015330143",
    )
    RK4_5 = _ssp(
        "1; 0.444370493651235 0.555629506348765; 0.620101851488403 0 0.379898148511597; 0.178079954393132 0 0 0.821920045606868;"
        " 0 0 0.517231671970585 0.096059710526147 0.386708617503269",
        "0.391752226571890; 0 0.368410593050371; 0 0 0.251891774271694; 0 0 0 0.544974750228521; 0 0 0 0.063692468666290 0.226007483236906",
    )
    RK5_10 = _ssp(
        "1; 0.258168167463650 0.741831832536350; 0 0.037493531856076 0.962506468143924; 0.595955269449077 0 0.404044730550923 0;"
        " 0.331848124368345 0 0 0.008466192609453 0.659685683022202; 0.086976414344414 0 0 0 0 0.913023585655586;"
        " 0.075863700003186 0 0.267513039663395 0 0 0 0.656623260333419; 0.005212058095597 0 0 0.407430107306541 0 0 0 0.587357834597862;"
        " 0.122832051947995 0 0 0 0 0 0 0 0.877167948052005;"
        " 0.075346276482673 0.000425904246091 0 0 0 0.064038648145995 0.354077936287492 0 0 0.506111234837749",
        "0.173586107937995; 0 0.218485490268790; 0 0.011042654588541 0.283478934653295; 0 0 0.118999896166647 0;"
        " 0.025030881091201 0 0 -0.002493476502164 0.194291675763785; 0 0 0 0 0 0.268905157462563;"
        " 0 0 0.066115378914543 0 0 0 0.193389726166555; 0 0 0 -0.119996962708895 0 0 0 0.172989562899406;"
        " 0.000000000000035 0 0 0 0 0 0 0 0.258344898092277;"
        " 0.016982542367506 0 0 0 0 0.018860764424857 0.098896719553054 0 0 0.149060685217562",
    )


def _from_table(path_name: str, prefix: str) -> dict[str, Tableau]:
    "coefficient tables stored as exact float hex (tools/make_tableaux_data.py): members `prefix.<name>`"
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), path_name)) as fh:
        raw = json.load(fh)
    h = float.fromhex
    return {
        key.split(".", 1)[1]: Tableau(tuple(Stage(h(c), tuple(h(v) for v in row)) for c, row in zip(t["c"], t["a"])), tuple(h(v) for v in t["b"]))
        for key, t in raw.items()
        if key.startswith(prefix + ".")
    }


# The reference's "graveyard" (tableaux/__init__.py:39-43): published methods it keeps but does not recommend.
WSO = _EnumProvider("WSO", _from_table("tableaux_graveyard.json", "WSO"))
WSO.__doc__ = """Methods with a higher weak stage order, named STAGES_ORDER_WSO (Biswas et al. 2023, "Explicit Runge-Kutta
methods that alleviate order reduction", arXiv:2310.02817; reference providers.py:641-861)."""
Shanks1965 = _EnumProvider("Shanks1965", _from_table("tableaux_graveyard.json", "Shanks1965"))
Shanks1965.__doc__ = """E. B. Shanks, "Higher order approximations of Runge-Kutta type", NASA TN D-2920 (1965); RK5_5, RK6_6, RK7_7
and RK8_10 only approximate their nominal orders (reference providers.py:863-1000)."""

BUILTIN_TABLEAUX: Sequence = [*RK1, *RK2, *RK3, *RK4, *RKZ, *SSP]
BUILTIN_EMBEDDED_TABLEAU: Sequence = [*RKE2, *RKE3, *RKE5]
GRAVEYARD: Sequence = [*WSO, *Shanks1965]


def pretty_tableau(tab, label: str | None = None) -> str:
    def cell(x: float) -> str:
        return f"{'+' if x >= 0 else '-'}{float(round(abs(x), 4)): <6}"

    body = [f"{cell(s.c)} | {' '.join(cell(v) for v in s.a)}" for s in tab.stages]
    foot = ["        | " + " ".join(cell(v) for v in row) for row in tab[1:]]
    width = max(len(line) for line in (*body, *foot))
    head = [label.rjust((width + len(label)) // 2)] if label is not None else []
    return "\n".join([*head, *body, "-" * width, *foot])


# the reference splits this module into a package (tableaux.common / tableaux.providers); both names resolve here
import sys as _sys  # noqa: E402

common = providers = _sys.modules[__name__]
