"""Chain traces and statistics on disk in the reference's format (mici/samplers.py:80-138, 231-301): one
``.npy`` memory map per chain and per traced variable / transition statistic,

    {dir}/trace_{chain}_{key}.npy            shape (n_iter, *value_shape), NaN- (inexact) or 0-initialised
    {dir}/stats_{chain}_{trans_key}_{stat}.npy   shape (n_iter,), dtype / default from ``statistic_types``

so that a run on the GPUs can be post-processed by the reference's tooling (``sample_chains(...,
force_memmap=True)`` produces exactly these files).  Host-side only: the arrays written here are what the
RCCL trace gather (or a DeviceBatch download) delivered."""

from __future__ import annotations

from pathlib import Path

import numpy as np


_KEEP = frozenset("._- ")


def get_valid_filename(string):
    """File-name form of a trace / statistic key: every character that is neither alphanumeric nor one of
    ``. _ -`` or a space is dropped (the rule of samplers.py:80-93, pinned by tests/golden/tracefmt_reference.npz)."""
    kept = [ch for ch in str(string) if ch.isalnum() or ch in _KEEP]
    return "".join(kept)


def generate_memmap_filenames(dir_path, prefix, key, indices):
    """``{dir}/{prefix}_{index}_{key}.npy`` for every chain index (the naming of samplers.py:96-105)."""
    stem = get_valid_filename(key)
    root = Path(dir_path)
    return [root / "{}_{}_{}.npy".format(prefix, index, stem) for index in indices]


def open_new_memmap(file_path, shape, default_val, dtype):
    """Create a ``.npy`` file as a writable memory map, every entry set to ``default_val`` (what
    samplers.py:108-131 produces)."""
    dims = (int(shape),) if np.isscalar(shape) else tuple(int(n) for n in shape)
    mm = np.lib.format.open_memmap(str(file_path), mode="w+", dtype=np.dtype(dtype), shape=dims)
    mm.fill(default_val)
    return mm


class MemmapTraceWriter:
    """Per-chain memory-mapped traces and statistics of N chains advanced in lock-step.

    ``trace_values``: dict key -> example value of ONE chain (fixes shape and dtype, as ``trace_func(state)``
    does in the reference); ``transitions``: dict trans_key -> object with ``statistic_types`` (may be None)."""

    def __init__(self, dir_path, n_chain, n_iter, trace_values, transitions=None, chain_offset=0):
        self.dir_path = Path(dir_path)
        self.dir_path.mkdir(parents=True, exist_ok=True)
        self.n_chain, self.n_iter = int(n_chain), int(n_iter)
        chains = range(chain_offset, chain_offset + self.n_chain)
        self.traces, self.stats = {}, {}
        for key, val in trace_values.items():
            array_val = np.array(val) if np.isscalar(val) else np.asarray(val)
            init = np.nan if np.issubdtype(array_val.dtype, np.inexact) else 0
            self.traces[key] = [
                open_new_memmap(f, (self.n_iter, *array_val.shape), init, array_val.dtype)
                for f in generate_memmap_filenames(self.dir_path, "trace", key, chains)]
        for trans_key, transition in (transitions or {}).items():
            if getattr(transition, "statistic_types", None) is None:
                continue
            self.stats[trans_key] = {
                key: [open_new_memmap(f, self.n_iter, val, dtype)
                      for f in generate_memmap_filenames(self.dir_path, "stats", f"{trans_key}_{key}", chains)]
                for key, (dtype, val) in transition.statistic_types.items()}

    def write(self, sample_index, trace_arrays=None, stat_arrays=None):
        """``trace_arrays``: key -> array [N, *shape]; ``stat_arrays``: trans_key -> {stat -> array [N]}
        (statistics the transition does not declare in ``statistic_types`` are ignored)."""
        for key, arr in (trace_arrays or {}).items():
            arr = np.asarray(arr)
            for c in range(self.n_chain):
                self.traces[key][c][sample_index] = arr[c]
        for trans_key, stats in (stat_arrays or {}).items():
            for key, arr in stats.items():
                if key in self.stats.get(trans_key, {}):
                    arr = np.broadcast_to(np.asarray(arr), (self.n_chain,))
                    for c in range(self.n_chain):
                        self.stats[trans_key][key][c][sample_index] = arr[c]

    def flush(self):
        for maps in self.traces.values():
            for m in maps:
                m.flush()
        for per_trans in self.stats.values():
            for maps in per_trans.values():
                for m in maps:
                    m.flush()

    def file_paths(self):
        """The same nested structure as the reference hands back (`_memmaps_to_file_paths`, samplers.py:134-157)."""
        return ({k: [Path(m.filename) for m in v] for k, v in self.traces.items()},
                {t: {k: [Path(m.filename) for m in v] for k, v in s.items()} for t, s in self.stats.items()})
