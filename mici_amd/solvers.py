"""Solver *selectors* with the reference's names (mici/solvers.py).  The iterations themselves run
inside the HIP kernels; these objects only name which device routine and which norm to use, and
carry the reference's default tolerances (solvers.py:47-54, 346-357)."""

from __future__ import annotations


class _Selector:
    def __init__(self, name, code):
        self.name, self.code = name, code

    def __repr__(self):
        return f"<mici_amd.solvers.{self.name}>"

    def __call__(self, *a, **k):
        raise TypeError(
            f"{self.name} is a device-routine selector: the iteration runs inside the HIP "
            "kernels (mm_implicit_leapfrog / mm_constrained_leapfrog) and cannot be called on host arrays"
        )


maximum_norm = _Selector("maximum_norm", 0)      # solvers.py:25-27
euclidean_norm = _Selector("euclidean_norm", 1)  # solvers.py:20-22

solve_fixed_point_direct = _Selector("solve_fixed_point_direct", 0)          # solvers.py:47-94
solve_fixed_point_steffensen = _Selector("solve_fixed_point_steffensen", 1)  # solvers.py:97-154

solve_projection_onto_manifold_newton = _Selector(                            # solvers.py:346-469
    "solve_projection_onto_manifold_newton", 0)

FIXED_POINT_DEFAULTS = dict(convergence_tol=1e-9, divergence_tol=1e10, max_iters=100,
                            norm=maximum_norm)
solve_projection_onto_manifold_quasi_newton = _Selector(                      # solvers.py:195-343
    "solve_projection_onto_manifold_quasi_newton", 1)
solve_projection_onto_manifold_newton_with_line_search = _Selector(           # solvers.py:472-614
    "solve_projection_onto_manifold_newton_with_line_search", 2)

PROJECTION_DEFAULTS = dict(constraint_tol=1e-9, position_tol=1e-8, divergence_tol=1e10,
                           max_iters=50, norm=maximum_norm, max_line_search_iters=10)


def norm_code(norm):
    if isinstance(norm, _Selector) and norm.name in ("maximum_norm", "euclidean_norm"):
        return norm.code
    name = getattr(norm, "__name__", None)  # accept the reference's own function objects
    if name == "maximum_norm":
        return 0
    if name == "euclidean_norm":
        return 1
    raise ValueError("norm must be maximum_norm or euclidean_norm (device built-ins)")


def fp_solver_code(solver):
    name = solver.name if isinstance(solver, _Selector) else getattr(solver, "__name__", None)
    if name == "solve_fixed_point_direct":
        return 0
    if name == "solve_fixed_point_steffensen":
        return 1
    raise ValueError("fixed_point_solver must be solve_fixed_point_direct or _steffensen")


def proj_solver_code(solver):
    name = solver.name if isinstance(solver, _Selector) else getattr(solver, "__name__", None)
    codes = {"solve_projection_onto_manifold_newton": 0,
             "solve_projection_onto_manifold_quasi_newton": 1,
             "solve_projection_onto_manifold_newton_with_line_search": 2}
    if name in codes:
        return codes[name]
    raise ValueError("projection_solver must be one of the three solve_projection_onto_manifold_* "
                     "selectors")
