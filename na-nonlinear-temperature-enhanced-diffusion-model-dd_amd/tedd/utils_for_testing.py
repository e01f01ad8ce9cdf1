"""Rate report used by the convergence tests (reference src/utils_for_testing.py:65-167)."""
from __future__ import annotations

import math
from typing import List

import numpy as np


def observed_rates_report(errors: List[float], *, expected_rate: float = 2.0, tolerance: float = 0.1,
                          cmp_type: str = "least", halt_print: bool = False) -> list:
    """3-point observed rates log2((E_k - E_{k+1}) / (E_{k+1} - E_{k+2})) for errors ordered from
    the coarsest to the finest level.  Asserts on the finest rate: at least
    `expected_rate - tolerance` for cmp_type "least", within `tolerance` of `expected_rate` for
    "equal".  A non-finite final rate is tolerated only when some error is below 1e-15 (the
    discretisation is then exact to rounding)."""
    if cmp_type not in ("equal", "least"):
        raise ValueError(f"cmp_type must be 'equal' or 'least', not {cmp_type}")
    say = (lambda *a: None) if halt_print else print
    floor = 1e-16
    rates = []
    say("\nObserved Rates (3-point formula):")
    for k in range(len(errors) - 2):
        num = errors[k] - errors[k + 1]
        den = errors[k + 1] - errors[k + 2]
        rate = math.log2(num / den) if (num > floor and den > floor) else float("nan")
        rates.append(rate)
        say(f"    Levels {k},{k + 1},{k + 2}: log2({num:.3e} / {den:.3e}) = {rate:.3f}")
    assert rates, "Not enough refinement levels."
    final = rates[-1]
    if not np.isfinite(final) and any(abs(e) < 1e-15 for e in errors):
        say("Non-finite final rate with an effectively zero error present. Not failing.")
        return rates
    assert np.isfinite(final), f"Final rate is not finite ({final})."
    say(f"  Final observed rate (3-point): {final:.3f}")
    if cmp_type == "least":
        assert final >= expected_rate - tolerance, f"Observed rate {final:.3f} not at least {expected_rate:.1f}"
    else:
        assert np.isclose(final, expected_rate, atol=tolerance), \
            f"Observed rate {final:.3f} not close to expected {expected_rate:.1f}"
    return rates
