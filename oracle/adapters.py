"""NumPy restatement of the reference's dual-averaging step-size adapter
(/root/reference/src/mici/adapters.py:174-389) on top of oracle/transitions.py.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): the checker for mici_amd.adapters; pinned to the reference
by tests/golden/adapt_*.npz (tools/gen_golden.py)."""

from math import exp, log

import numpy as np

from . import integrators as orc


class AdaptationError(RuntimeError):
    pass


def find_init_step_size(ad, q, p, direction, max_iters=100):
    """_find_and_set_init_step_size (adapters.py:271-344): double / halve from 1 until |delta h| over ONE step
    crosses log 2; a failed step counts as "too big" from then on."""
    h_init = ad.h(q, p)
    if np.isnan(h_init):
        raise AdaptationError("Hamiltonian evaluating to NaN at initial state.")
    step_size = 1.0
    threshold = log(2)
    too_big = False
    for s in range(max_iters):
        q1, p1, status, n_done = ad.steps(q, p, direction * step_size, 1)
        if status != orc.ST_OK:
            too_big = True
            step_size /= 2
            continue
        delta_h = abs(h_init - ad.h(q1, p1))
        if s == 0 or np.isnan(delta_h):
            too_big = bool(np.isnan(delta_h) or delta_h > threshold)
        if (too_big and delta_h <= threshold) or (not too_big and delta_h > threshold):
            return step_size
        if too_big:
            step_size /= 2
        else:
            step_size *= 2
    raise AdaptationError(f"Could not find reasonable initial step size in {max_iters} iterations "
                          f"(final step size {step_size}).")


def initial_state(init_step_size, log_step_size_reg_target=None):
    return dict(iter=0, smoothed_log_step_size=0.0, adapt_stat_error=0.0,
                log_step_size_reg_target=log(10 * init_step_size) if log_step_size_reg_target is None
                else log_step_size_reg_target)


def update(state, accept_stat, adapt_stat_target=0.8, log_step_size_reg_coefficient=0.05,
           iter_decay_coeff=0.75, iter_offset=10):
    """DualAveragingStepSizeAdapter.update (adapters.py:346-368); returns the step size to use next."""
    state["iter"] += 1
    error_weight = 1 / (iter_offset + state["iter"])
    state["adapt_stat_error"] *= 1 - error_weight
    state["adapt_stat_error"] += error_weight * (adapt_stat_target - accept_stat)
    smoothing_weight = (1 / state["iter"]) ** iter_decay_coeff
    log_step_size = state["log_step_size_reg_target"] - (
        state["adapt_stat_error"] * state["iter"] ** 0.5 / log_step_size_reg_coefficient)
    state["smoothed_log_step_size"] *= 1 - smoothing_weight
    state["smoothed_log_step_size"] += smoothing_weight * log_step_size
    return exp(log_step_size)


def finalize(states):
    """adapters.py:370-389: one chain -> exp(smoothed); several -> arithmetic mean of the per-chain step sizes
    (arithmetic_mean_log_step_size_reducer, adapters.py:126-135)."""
    if isinstance(states, dict):
        return exp(states["smoothed_log_step_size"])
    return float(np.mean([exp(s["smoothed_log_step_size"]) for s in states]))
