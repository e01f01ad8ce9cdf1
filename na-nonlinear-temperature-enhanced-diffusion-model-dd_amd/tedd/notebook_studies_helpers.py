"""Helpers the convergence-study notebooks call around MMSTrial.run_for_errors
(reference src/notebook_studies_helpers.py:13-211): the rate report that does not raise, the two figures,
and the template-notebook cell replacement.  matplotlib is imported when a figure is asked for; a notebook
file is read and written as the JSON document it is."""
from __future__ import annotations

import json
from typing import List, Tuple

import numpy as np

from . import prob1base as p1
from .utils_for_testing import observed_rates_report


def report_on_rates(errors: List[float], *, expected_rate: float = 2.0, tolerance: float = 0.1,
                    cmp_type: str = "least", title: str = "Observed Rates") -> Tuple[List[float], bool]:
    """observed_rates_report with its assertion turned into the second return value: (rates, True) when
    the finest rate meets `expected_rate` (cmp_type "least" or "equal", within `tolerance`), ([], False)
    otherwise (reference src/notebook_studies_helpers.py:13-59)."""
    print(f"\n{title}:")
    print("-" * len(title))
    verdict = f"expected rate of {expected_rate:.1f} (within tolerance ±{tolerance:.1f})"
    try:
        rates = observed_rates_report(errors=errors, expected_rate=expected_rate, tolerance=tolerance,
                                      cmp_type=cmp_type, halt_print=False)
    except Exception as err:
        print(f"❌ Error in rate calculation: {err}")
        print(f"❌ Does not match {verdict}")
        return [], False
    print(f"Final rate: {rates[-1]:.3f}")
    print(f"✅ Matches {verdict}")
    return rates, True


def plot_errors_and_rates(errors: List[float], rates: List[float], x_values: List, x_label: str, title: str,
                          expected_rate: float = None, log_scale: bool = True):
    """Errors over the refinement parameter (log-log by default) beside the 3-point rates, which belong to
    the inner refinement levels (reference src/notebook_studies_helpers.py:62-122)."""
    import matplotlib.pyplot as plt
    _fig, (left, right) = plt.subplots(1, 2, figsize=(15, 6))
    left.plot(x_values, errors, "o-", linewidth=2, markersize=8)
    if log_scale:
        left.set_xscale("log")
        left.set_yscale("log")
    left.set(xlabel=x_label, ylabel="Error", title=f"{title} - Error Values")
    left.grid(True, which="both", ls="-")
    if rates:
        right.plot(x_values[1:-1], rates, "o-", linewidth=2, markersize=8)
        if expected_rate:
            right.axhline(y=expected_rate, color="r", linestyle="--", label=f"Expected Rate = {expected_rate}")
            right.legend()
        right.set(xlabel=x_label, ylabel="Convergence Rate", title=f"{title} - Convergence Rates")
        right.grid(True)
    else:
        right.text(0.5, 0.5, "Not enough data points\nto calculate rates", ha="center", va="center", fontsize=14)
    plt.tight_layout()
    plt.show()


def visualize_mms_solution(mms_case, time_point=0.1):
    """Filled contours of the five exact fields at one time and, in the sixth panel, the regularised
    Heaviside function for eta = 10, 50, 100 (reference src/notebook_studies_helpers.py:125-178)."""
    import matplotlib.pyplot as plt
    xx, yy = mms_case.grid.xx, mms_case.grid.yy
    _fig, axs = plt.subplots(2, 3, figsize=(18, 10))
    panels = (("T", "T (Temperature)", "hot"), ("cp", "cp (Product Concentration)", "viridis"),
              ("cl", "cl (Liquid Concentration)", "Blues"), ("cd", "cd (Dissolved Concentration)", "Greens"),
              ("cs", "cs (Solid Concentration)", "Purples"))
    for ax, (var, label, cmap) in zip(axs.flat, panels):
        image = ax.contourf(xx, yy, getattr(mms_case, var)(time_point, xx, yy), cmap=cmap)
        plt.colorbar(image, ax=ax)
        ax.set_title(label)
    x = np.linspace(-1, 1, 1000)
    for eta in (10, 50, 100):
        axs[1, 2].plot(x, p1.heaviside_regularized(x, eta), label=f"η = {eta}")
    axs[1, 2].set_title("Regularized Heaviside Function")
    axs[1, 2].legend()
    axs[1, 2].grid(True)
    plt.tight_layout()
    plt.suptitle(f"MMS Solution at time t = {time_point}", fontsize=16)
    plt.subplots_adjust(top=0.93)
    plt.show()


def modify_notebook_cell(template_path: str, output_path: str, cell_index: int, new_cell_content: str):
    """Copies the notebook `template_path` to `output_path` with the source of cell `cell_index` (0-based)
    replaced (reference src/notebook_studies_helpers.py:183-211).  Works on the notebook's JSON directly
    (format 4: top-level "cells", each with a "source" string or list of lines)."""
    with open(template_path, encoding="utf-8") as f:
        notebook = json.load(f)
    cells = notebook["cells"]
    assert 0 <= cell_index < len(cells), "Cell index out of bounds."
    cells[cell_index]["source"] = new_cell_content.splitlines(keepends=True)
    with open(output_path, "w", encoding="utf-8") as f:
        json.dump(notebook, f, indent=1, ensure_ascii=False)
        f.write("\n")
