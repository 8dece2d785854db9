"""Chain output in the layout ArviZ consumes (reference mici/interop.py:31-143).

The reference converts ``sample_chains`` output - ``traces`` / ``stats`` dictionaries of per-chain arrays - into an
``arviz.InferenceData`` (ArviZ < 1.0) or an ``xarray.DataTree`` (ArviZ >= 1.0) with two groups:

  posterior     every traced variable, stacked to ``[chain, draw, *variable_shape]``
  sample_stats  the transition statistics stacked to ``[chain, draw]`` under ArviZ's names
                (``n_step`` -> ``n_steps``, ``accept_stat`` -> ``acceptance_rate``; the traced Hamiltonian and log
                density, if present under ``energy_key`` / ``lp_key``, are copied in as ``energy`` / ``lp``)

``arviz_layout`` builds exactly those two dictionaries (NumPy only) from either form this package produces: the
reference's per-chain lists, or the batched ``[draw, chain, ...]`` arrays that the device-resident transitions and the
RCCL trace gather deliver.  ``convert_to_inference_data`` / ``convert_to_data_tree`` hand them to ArviZ with the
reference's version checks; ArviZ itself is imported lazily and is not a dependency of the path."""

from __future__ import annotations

import numpy as np

# statistic names of mici.transitions -> ArviZ's sample_stats names (interop.py:38-39)
STAT_RENAMES = {"n_step": "n_steps", "accept_stat": "acceptance_rate"}


def per_chain_lists(batched):
    """``{key: array[draw, chain, ...]}`` (batched device output) -> ``{key: [array[draw, ...] per chain]}``
    (the reference's ``sample_chains`` form)."""
    out = {}
    for key, arr in batched.items():
        arr = np.asarray(arr)
        if arr.ndim < 2:
            raise ValueError(f"{key}: batched arrays are [draw, chain, ...], got shape {arr.shape}")
        out[key] = [np.ascontiguousarray(arr[:, c]) for c in range(arr.shape[1])]
    return out


def _as_lists(data):
    first = next(iter(data.values()), None)
    if first is None or isinstance(first, (list, tuple)):
        return {k: list(v) for k, v in data.items()}
    return per_chain_lists(data)


def preprocess_stats(traces, stats, energy_key="energy", lp_key="lp"):
    """Statistics dictionary under ArviZ's variable names (interop.py:31-45); the input is not modified."""
    out = dict(stats)
    for old, new in STAT_RENAMES.items():
        out[new] = out.pop(old)  # KeyError if absent, as in the reference
    if energy_key is not None and energy_key in traces:
        out["energy"] = traces[energy_key]
    if lp_key is not None and lp_key in traces:
        out["lp"] = traces[lp_key]
    return out


def stack_chains(data):
    """``{key: [per-chain arrays]}`` -> ``{key: array[chain, draw, ...]}`` (interop.py:48-51)."""
    return {key: np.stack(per_chain) for key, per_chain in data.items()}


def arviz_layout(traces, stats, energy_key="energy", lp_key="lp"):
    """The ``posterior`` and ``sample_stats`` groups as dictionaries of ``[chain, draw, ...]`` arrays."""
    traces, stats = _as_lists(traces), _as_lists(stats)
    return {"posterior": stack_chains(traces),
            "sample_stats": stack_chains(preprocess_stats(traces, stats, energy_key, lp_key))}


def _arviz():
    try:
        import arviz
    except ImportError as e:  # the reference has the same hard requirement (interop.py:84)
        raise ImportError("ArviZ is needed for this conversion; `arviz_layout` gives the same data as "
                          "plain dictionaries of [chain, draw, ...] arrays") from e
    return arviz


def _major(version):
    return int(str(version).split(".")[0])


def convert_to_inference_data(traces, stats, energy_key="energy", lp_key="lp"):
    """``arviz.InferenceData`` with ``posterior`` and ``sample_stats`` groups (interop.py:54-93; ArviZ < 1.0)."""
    arviz = _arviz()
    if _major(arviz.__version__) >= 1:
        raise RuntimeError("InferenceData was removed in ArviZ v1.0+ in favour of xarray.DataTree")
    traces, stats = _as_lists(traces), _as_lists(stats)
    sample_stats = preprocess_stats(traces, stats, energy_key, lp_key)
    return arviz.InferenceData(posterior=arviz.dict_to_dataset(traces), sample_stats=arviz.dict_to_dataset(sample_stats))


def convert_to_data_tree(traces, stats, energy_key="energy", lp_key="lp"):
    """``xarray.DataTree`` with the same two groups (interop.py:96-143; ArviZ >= 1.0)."""
    arviz = _arviz()
    if _major(arviz.__version__) < 1:
        raise RuntimeError("xarray.DataTree support requires ArviZ v1.0+")
    return arviz.from_dict(arviz_layout(traces, stats, energy_key, lp_key))
