"""Import shim for the upstream reference (THIS CONTAINER ONLY).

The reference at /root/reference needs `numba`, `gym` and `h5py`, none of which
is installed here.  Every `@jit` body in the reference is plain Python/NumPy, so
an identity `numba.jit` gives the same integer semantics.  This
module injects three stub modules into `sys.modules` and puts /root/reference on
`sys.path`.  It is tool code of this repo: nothing from /root/reference is copied
and nothing here ships to the GPU box (the reference does not exist there).

Usage:
    from tools.oracle.ref_stubs import import_reference
    ref = import_reference()          # namespace with the reference's modules
"""
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("STRATEGO_REFERENCE_ROOT", "/root/reference")


class _TypeStub:
    """Stands in for numba type objects: callable and subscriptable, returns itself."""

    def __call__(self, *a, **k):
        return self

    def __getitem__(self, item):
        return self


def _install_stubs():
    if "numba" not in sys.modules:
        numba = types.ModuleType("numba")

        def jit(*args, **kwargs):
            # used both as @jit(signature, nopython=True, ...) -> decorator factory
            def deco(fn):
                return fn

            return deco

        numba.jit = jit
        numba.njit = jit
        for name in ("int64", "boolean", "float32", "float64", "int32"):
            setattr(numba, name, _TypeStub())
        nb_types = types.ModuleType("numba.types")
        nb_types.UniTuple = lambda *a, **k: _TypeStub()
        numba.types = nb_types
        sys.modules["numba"] = numba
        sys.modules["numba.types"] = nb_types

    if "gym" not in sys.modules:
        gym = types.ModuleType("gym")
        spaces = types.ModuleType("gym.spaces")

        class Discrete:
            def __init__(self, n):
                self.n = int(n)

        class Box:
            def __init__(self, low=None, high=None, shape=None, dtype=None):
                self.low, self.high, self.shape, self.dtype = low, high, shape, dtype

        class Dict(dict):
            def __init__(self, spaces_dict=None):
                super().__init__(spaces_dict or {})
                self.spaces = dict(spaces_dict or {})

        spaces.Discrete, spaces.Box, spaces.Dict = Discrete, Box, Dict
        gym.spaces = spaces
        sys.modules["gym"] = gym
        sys.modules["gym.spaces"] = spaces

    if "h5py" not in sys.modules:
        sys.modules["h5py"] = types.ModuleType("h5py")


def reference_available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "stratego_env"))


def import_reference():
    """Return a namespace exposing the reference modules used by the oracle tools."""
    if not reference_available():
        raise RuntimeError("reference tree not found at %s (only present in the build container)" % REFERENCE_ROOT)
    _install_stubs()
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    import stratego_env as ref_pkg  # noqa: the REFERENCE package, not this repo's
    from stratego_env.game import stratego_procedural_impl as impl
    from stratego_env.game import stratego_procedural_env as penv
    from stratego_env.game import util as util
    from stratego_env.game import config as config
    from stratego_env.game import enums as enums
    from stratego_env import stratego_multiagent_env as maenv

    ns = types.SimpleNamespace(pkg=ref_pkg, impl=impl, penv=penv, util=util, config=config, enums=enums, maenv=maenv)
    return ns
