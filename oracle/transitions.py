"""NumPy restatement of the reference's momentum refresh and Metropolis static-integration transition
(/root/reference/src/mici/transitions.py:129-142, 275-315, 318-352) on top of the oracle integrators.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): the checker for mm_metropolis_accept and
mici_amd.transitions; pinned to the reference by tests/golden/transition_*.npz (tools/gen_golden.py).

Randomness is passed in: ``z`` is the standard-normal vector the reference draws in
``system.sample_momentum(state, rng)`` and ``u`` the uniform it draws in the accept step - which it draws
ONLY when the trajectory raised no IntegratorError (`not integration_error and rng.uniform() < accept_prob`,
transitions.py:309), so a replay must consume its recorded uniforms the same way."""

import numpy as np

from . import integrators as orc


class Adapter:
    """h(q, p), sample_momentum(q, z), steps(q, p, dt, n) -> (q, p, status, n_done) of one oracle system."""

    def __init__(self, h, sample_momentum, steps):
        self.h, self.sample_momentum, self.steps = h, sample_momentum, steps


def euclid_adapter(system, free_coefficients=None, initial_h1_flow_step=True):
    def steps(q, p, dt, n):
        if free_coefficients is None:
            q2, p2 = orc.leapfrog_steps(system, q, p, dt, n)
        else:
            q2, p2 = orc.composition_steps(system, q, p, dt, n, free_coefficients, initial_h1_flow_step)
        return q2, p2, orc.ST_OK, n

    return Adapter(system.h, lambda q, z: system.msqrt(z), steps)


def riemann_adapter(system, **kw):
    def h(q, p):
        try:
            with np.errstate(all="ignore"):
                return system.h(orc._State(q, p))
        except (orc.LinAlgError, ValueError):
            return np.nan

    return Adapter(h, lambda q, z: system.sample_momentum(orc._State(q, None), z),
                   lambda q, p, dt, n: orc.implicit_leapfrog_steps(system, q, p, dt, n, **kw))


def constrained_adapter(system, **kw):
    def sample_momentum(q, z):  # systems.py:614-616: sample, then project onto the cotangent space
        return system.project_onto_cotangent_space(system.msqrt(z), system.constraint.jacob_constr(q))

    return Adapter(system.h, sample_momentum,
                   lambda q, p, dt, n: orc.constrained_leapfrog_steps(system, q, p, dt, n, **kw))


def metropolis_static_transition(ad, q, p, direction, step_size, n_step, draw_uniform):
    """MetropolisIntegrationTransition._sample_n_step (transitions.py:275-315).  ``draw_uniform`` is called
    at most once, exactly when the reference calls ``rng.uniform()``.  Returns the new (q, p, dir) and the
    statistics the reference records."""
    h_init = ad.h(q, p)
    q_p, p_p, status, n_done = ad.steps(q, p, direction * step_size, n_step)
    integration_error = status != orc.ST_OK
    if n_done > 0:  # `state_p is not state`: at least one step completed
        h_diff = h_init - ad.h(q_p, p_p)
        accept_prob = 0.0 if np.isnan(h_diff) else float(np.exp(min(0.0, h_diff)))
    else:
        accept_prob = 0.0
    stats = dict(
        n_step=int(n_done), metrop_accept_prob=accept_prob,
        accept_stat=0.0 if integration_error else accept_prob,
        convergence_error=status in (orc.ST_DIVERGED, orc.ST_MAX_ITERS, orc.ST_SOLVER_LINALG),
        non_reversible_step=status == orc.ST_NON_REVERSIBLE, status=int(status), accepted=False,
    )
    if not integration_error and draw_uniform() < accept_prob:
        q, p = q_p, p_p  # proposal dir was negated (involution) and is negated again below
        stats["accepted"] = True
    else:
        direction = -direction
    return np.array(q, dtype=np.float64), np.array(p, dtype=np.float64), int(direction), stats


def correlated_momentum(ad, q, p, z, coeff):
    """CorrelatedMomentumTransition.sample (transitions.py:185-197): ``z`` is consumed only when the reference
    draws (coeff != 0 or no momentum yet)."""
    if p is None or coeff == 1:
        return ad.sample_momentum(q, z)
    if coeff != 0:
        return p * (1.0 - coeff**2) ** 0.5 + coeff * ad.sample_momentum(q, z)
    return p
