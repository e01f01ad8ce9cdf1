"""Observed-convergence-rate helper of the reference's src/cvg_studies_base.py:25-106."""
from __future__ import annotations

import math
from typing import List, NamedTuple, Tuple

import numpy as np


class _RateStatus(NamedTuple):
    OK: str = "OK"
    INSUFFICIENT_DATA: str = "Insufficient Data"
    ZERO_DENOMINATOR_ZERO_NUMERATOR: str = "Differences near zero (converged/stalled?)"
    ZERO_DENOMINATOR_NONZERO_NUMERATOR: str = "Unstable rate (denominator near zero)"
    NON_POSITIVE_RATIO: str = "Non-positive ratio (convergence issue?)"
    ERROR_INCREASING: str = "Error increasing significantly"


RateStatus = _RateStatus()


def calculate_observed_rates(errors: List[float], refinement_factor: float = 2.0) -> List[Tuple[float, str]]:
    """3-point observed rates: log_r((E_k - E_{k+1}) / (E_{k+1} - E_{k+2})) for errors ordered from
    the coarsest to the finest level; each rate comes with a status string."""
    assert len(errors) >= 3, "At least 3 error values are required for rate calculation."
    assert refinement_factor > 1.0, "Refinement factor must be > 1.0"
    assert all(e >= 0 for e in errors), "All error values must be positive for rate calculation."
    log_r = math.log(refinement_factor)
    tiny = np.finfo(float).eps
    out = []
    for k in range(len(errors) - 2):
        num = errors[k] - errors[k + 1]
        den = errors[k + 1] - errors[k + 2]
        rate, status = np.nan, RateStatus.OK
        if den < 0:
            status = RateStatus.ERROR_INCREASING
        elif num <= 0:
            status = RateStatus.NON_POSITIVE_RATIO
        else:
            if abs(den) < tiny:
                status = (RateStatus.ZERO_DENOMINATOR_ZERO_NUMERATOR if abs(num) < tiny
                          else RateStatus.ZERO_DENOMINATOR_NONZERO_NUMERATOR)
            ratio = num / den if den != 0 else np.inf
            assert ratio > 0
            rate = math.log(ratio) / log_r
        out.append((rate, status))
    return out
