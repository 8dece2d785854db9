"""Host-side trace writer vs the reference's on-disk format (samplers.py:80-131, 231-301): file names, dtypes,
shapes and default fill values recorded from the reference in tests/golden/tracefmt_reference.npz.  The
transition's statistic_types are read from mici_amd.transitions without touching a device."""

import numpy as np

from conftest import load_golden
from mici_amd import traces


class _StatTypes:  # statistic_types of MetropolisStaticIntegrationTransition, no device needed
    def __init__(self):
        from mici_amd.transitions import MetropolisStaticIntegrationTransition as T
        t = T.__new__(T)
        T.__init__(t, system=None, integrator=None, n_step=2)
        self.statistic_types = t.statistic_types


def test_file_names_match_reference():
    g = load_golden("tracefmt_reference")
    for i, key in enumerate(g["keys"]):
        mine = [p.name for p in traces.generate_memmap_filenames("/tmp/x", "trace", str(key), range(3))]
        assert mine == [str(s) for s in g[f"names_{i}"]]


def test_writer_creates_the_reference_files(tmp_path):
    g = load_golden("tracefmt_reference")
    w = traces.MemmapTraceWriter(tmp_path, 2, 5, {"pos": np.zeros(4), "count": 3, "flag": np.array(True)},
                                 {"integration_transition": _StatTypes(), "momentum_transition": type("M", (), {"statistic_types": None})()})
    files = sorted(p.name for p in tmp_path.iterdir())
    assert files == [str(s) for s in g["files"]]
    for k, dt, init, nd in zip(g["trace_keys"], g["trace_dtypes"], g["trace_init"], g["trace_ndim"]):
        m = w.traces[str(k)][0]
        assert str(m.dtype) == str(dt) and m.ndim == int(nd) and m.shape[0] == 5
        first = float(np.asarray(m).ravel()[0])
        assert (np.isnan(first) and np.isnan(init)) or first == init
    for k, dt, default in zip(g["stat_names"], g["stat_dtypes"], g["stat_defaults"]):
        m = w.stats["integration_transition"][str(k)][1]
        assert str(m.dtype) == str(dt) and m.shape == (5,)
        first = float(m[0])
        assert (np.isnan(first) and np.isnan(default)) or first == default
    # batched writes land in the per-chain files; undeclared statistics are ignored
    w.write(2, {"pos": np.arange(8.0).reshape(2, 4), "count": [7, 9], "flag": [True, False]},
            {"integration_transition": {"accept_stat": [0.25, 0.5], "n_step": [3, 4], "accepted": [1, 0],
                                        "convergence_error": [False, True], "step_size": 0.1}})
    w.flush()
    tr, st = w.file_paths()
    assert np.array_equal(np.load(tr["pos"][1])[2], np.arange(4.0, 8.0))
    assert np.load(tr["count"][0])[2] == 7 and np.isnan(np.load(tr["pos"][0])[0]).all()
    assert np.load(st["integration_transition"]["accept_stat"][1])[2] == 0.5
    assert np.load(st["integration_transition"]["n_step"][0])[2] == 3
    assert np.load(st["integration_transition"]["n_step"][0])[0] == -1
    assert bool(np.load(st["integration_transition"]["convergence_error"][1])[2]) is True
    assert np.load(st["integration_transition"]["step_size"][1])[2] == 0.1
