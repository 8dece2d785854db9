"""ChainState: the (pos, mom, dir) triple integrators step (reference states.py:160-305).

The device path needs no per-state memoisation cache - what the reference memoises between
sub-steps (gradient at step end, metric across momentum iterations) is kept in
registers / LDS inside the kernels - so this class is only the variable container with the same
surface: attribute access, ``copy()``, ``in``, and independence of copies
(reference tests/test_states.py:130-353)."""

from __future__ import annotations

import copy as _copy

from .errors import ReadOnlyStateError


class ChainState:
    def __init__(self, *, _read_only=False, **variables):
        for name in variables:
            if name.startswith("_") or name == "copy":
                raise ValueError(f"invalid state variable name {name!r}")
        self.__dict__["_variables"] = variables
        self.__dict__["_read_only"] = _read_only

    def __getattr__(self, name):
        variables = self.__dict__.get("_variables", {})
        if name in variables:
            return variables[name]
        raise AttributeError(f"'{type(self).__name__}' object has no attribute '{name}'")

    def __setattr__(self, name, value):
        if self._read_only:
            raise ReadOnlyStateError("ChainState instance is read-only.")
        if name in self._variables:
            self._variables[name] = value
        else:
            super().__setattr__(name, value)

    def __contains__(self, name):
        return name in self._variables

    def copy(self, *, read_only=False):
        return type(self)(
            _read_only=read_only,
            **{name: _copy.copy(val) for name, val in self._variables.items()},
        )

    def __getstate__(self):
        return {"variables": self._variables, "read_only": self._read_only}

    def __setstate__(self, state):
        self.__dict__["_variables"] = state["variables"]
        self.__dict__["_read_only"] = state["read_only"]

    def __repr__(self):
        body = ",\n ".join(f"{k}={v}" for k, v in self._variables.items())
        return f"{type(self).__name__}(\n {body})"
