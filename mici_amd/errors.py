"""Exception types: same names and hierarchy as the reference's ``mici/errors.py:6-35`` so that
``except IntegratorError`` in mici.transitions (transitions.py:292-295, 670-672) behaves the same."""


class Error(RuntimeError):
    """Base class for errors."""


class IntegratorError(Error):
    """Error raised when integrator step fails."""


class NonReversibleStepError(IntegratorError):
    """Error raised when integrator step fails reversibility check."""


class ConvergenceError(IntegratorError):
    """Error raised when solver fails to converge within allowed iterations."""


class LinAlgError(Error):
    """Error raised when a matrix operation raises a linear algebra error."""


class HamiltonianDivergenceError(IntegratorError):
    """Error raised when integration of Hamiltonian dynamics diverges."""


class AdaptationError(Error):
    """Error raised when adaptation of transition parameters fails."""


class ReadOnlyStateError(Error):
    """Error raised when writing to attributes of read-only chain state."""


class DeviceError(Error):
    """The HIP library is missing, failed to load, or a device call failed (no CPU fallback)."""


# per-chain status codes of include/mici_amd.h -> the exception the reference would have raised
ST_OK, ST_DIVERGED, ST_MAX_ITERS, ST_SOLVER_LINALG, ST_NON_REVERSIBLE, ST_LINALG = range(6)

_STATUS_EXC = {
    ST_DIVERGED: (ConvergenceError, "Solver diverged."),
    ST_MAX_ITERS: (ConvergenceError, "Solver did not converge within the allowed iterations."),
    ST_SOLVER_LINALG: (ConvergenceError, "Linear algebra error inside iterative solver."),
    ST_NON_REVERSIBLE: (NonReversibleStepError, "Non-reversible step."),
    ST_LINALG: (LinAlgError, "Matrix is not finite or Cholesky factorisation failed."),
}


def raise_for_status(status):
    """Raise the reference-equivalent exception for a non-zero per-chain status code."""
    status = int(status)
    if status == ST_OK:
        return
    exc, msg = _STATUS_EXC.get(status, (IntegratorError, f"Unknown failure status {status}."))
    raise exc(msg)
