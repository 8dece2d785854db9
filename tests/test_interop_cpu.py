"""CPU: the ArviZ-layout exporter (mici_amd/interop.py) against tests/golden/interop_arviz_layout.npz, recorded from
the reference's own ``_preprocess_stats`` + ``_stack_arrays`` (mici/interop.py:31-51) - everything the reference's
``convert_to_inference_data`` / ``convert_to_data_tree`` do before handing the two dictionaries to ArviZ."""

import numpy as np
import pytest

from conftest import load_golden

from mici_amd import interop


def _inputs(g):
    traces = {k: list(g[f"in_trace_{k}"]) for k in g["trace_keys"]}
    stats = {k: list(g[f"in_stat_{k}"]) for k in g["stat_in_keys"]}
    return traces, stats


@pytest.mark.parametrize("tag,keys", [("default", ("energy", "lp")), ("nokeys", (None, None)), ("absent", ("h", "logp"))])
def test_layout_matches_reference_fixture(tag, keys):
    g = load_golden("interop_arviz_layout")
    traces, stats = _inputs(g)
    before = {k: [a.copy() for a in v] for k, v in stats.items()}
    lay = interop.arviz_layout(traces, stats, *keys)
    assert sorted(lay["posterior"]) == sorted(g["trace_keys"])
    for k in g["trace_keys"]:
        assert np.array_equal(lay["posterior"][k], g[f"post_{k}"])
        assert lay["posterior"][k].shape[:2] == (3, 7)  # [chain, draw, ...]
    assert sorted(lay["sample_stats"]) == list(g[f"{tag}_stat_keys"])
    for k in g[f"{tag}_stat_keys"]:
        got, want = lay["sample_stats"][k], g[f"{tag}_stat_{k}"]
        assert got.dtype == want.dtype and np.array_equal(got, want), k
    assert "n_step" not in lay["sample_stats"] and "accept_stat" not in lay["sample_stats"]
    assert sorted(stats) == sorted(before)  # the caller's dictionary is not renamed in place


def test_batched_device_output_gives_the_same_layout():
    """[draw, chain, ...] arrays (what the device-resident transitions and the trace gather deliver) == per-chain lists"""
    g = load_golden("interop_arviz_layout")
    traces, stats = _inputs(g)
    bt = {k: np.stack(v, axis=1) for k, v in traces.items()}
    bs = {k: np.stack(v, axis=1) for k, v in stats.items()}
    a, b = interop.arviz_layout(traces, stats), interop.arviz_layout(bt, bs)
    for grp in ("posterior", "sample_stats"):
        assert sorted(a[grp]) == sorted(b[grp])
        for k in a[grp]:
            assert np.array_equal(a[grp][k], b[grp][k])


def test_missing_statistics_and_missing_arviz_fail_like_the_reference():
    with pytest.raises(KeyError):  # the reference pops "n_step" / "accept_stat" unconditionally
        interop.arviz_layout({"pos": [np.zeros((2, 1))]}, {"n_step": [np.zeros(2)]})
    try:
        import arviz  # noqa: F401
    except ImportError:
        with pytest.raises(ImportError):
            interop.convert_to_data_tree({"pos": [np.zeros((2, 1))]}, {"n_step": [np.zeros(2)], "accept_stat": [np.zeros(2)]})
