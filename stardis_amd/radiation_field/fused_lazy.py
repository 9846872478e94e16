"""The lazily materialised outputs of the fused call: `opacities_dict` entries that form on first read, `total_alphas` and
`I_nus` that stay on the device until somebody looks, and the per-process bound on the device memory such fields keep
(DEVICE_BUDGET_BYTES on the public module `stardis_amd.radiation_field.fused`)."""
import numpy as np

from stardis_amd.radiation_field.opacities import Opacities


class _Thunk:
    """A dictionary entry that has not been asked for yet."""

    __slots__ = ("make",)

    def __init__(self, make):
        self.make = make


class LazyOpacitiesDict(dict):
    """`opacities_dict` whose array entries materialise on first read.  Every read path goes through __getitem__ (a
    trivially overridden __iter__ keeps dict(d) / d.copy() off CPython's raw-value fast path)."""

    def __getitem__(self, key):
        value = dict.__getitem__(self, key)
        if isinstance(value, _Thunk):
            value = value.make()
            dict.__setitem__(self, key, value)
        return value

    def __iter__(self):
        return dict.__iter__(self)

    def get(self, key, default=None):
        return self[key] if key in self else default

    def _all(self):
        for key in dict.keys(self):
            self[key]

    def items(self):
        self._all()
        return dict.items(self)

    def values(self):
        self._all()
        return dict.values(self)

    def copy(self):
        self._all()
        return dict(dict.items(self))

    def pop(self, key, *default):
        if key in self:
            self[key]
        return dict.pop(self, key, *default)


# Device memory a fused field keeps for its lazy entries (total, line plane, broadening tables, staged inputs, tracked
# intensities: up to ~1 GB at 1e6 lines) is BOUNDED per process: fields are remembered weakly in creation order, and when the
# bytes they hold exceed DEVICE_BUDGET_BYTES the oldest ones are released — their entries materialise on the host first (what
# the reference would hold anyway), then the device twins go back to the context's pool.  A caller that keeps many outputs
# (model grids, fits) therefore runs out of nothing the reference would not run out of.  release_device() does it by hand.
_LIVE = []  # [(weakref to the field, bytes)], oldest first


def _budget():
    from stardis_amd.radiation_field import fused

    return fused.DEVICE_BUDGET_BYTES


def _enforce_budget(new_bytes):
    import weakref  # noqa: F401

    alive = [(r, b) for r, b in _LIVE if r() is not None and getattr(r().opacities, "_device_bytes", 0)]
    _LIVE[:] = alive
    total = sum(b for _, b in alive) + new_bytes
    while alive and total > _budget():
        ref, b = alive.pop(0)
        field = ref()
        if field is not None:
            release_device(field)
        total -= b
    _LIVE[:] = alive


def release_device(field, materialize=True):
    """Drop the device memory a fused RadiationField holds.  materialize=True (default) first forms every lazy entry on the host
    — opacities_dict, total_alphas, I_nus — so that nothing is lost; False discards what has not been read (the entries then
    read as the general path would recompute them is NOT attempted: they raise)."""
    opac = field.opacities
    if not isinstance(opac, FusedOpacities):
        return
    if materialize:
        opac.opacities_dict._all()
        opac.total_alphas  # noqa: B018
        if getattr(field, "_I_dev", None) is not None:
            field.I_nus  # noqa: B018
    else:
        def gone():
            raise RuntimeError("this entry was released with release_device(materialize=False) before it was read")
        for key in list(dict.keys(opac.opacities_dict)):
            if isinstance(dict.__getitem__(opac.opacities_dict, key), _Thunk):
                dict.__setitem__(opac.opacities_dict, key, _Thunk(gone))
        opac._discarded = opac._total_host is None
    opac._total_twin = None
    opac._total_dev = None
    opac._resident = {}
    opac._device_bytes = 0
    if getattr(field, "_I_dev", None) is not None:
        field._I_dev = None
    field._device_blob = None


class FusedOpacities(Opacities):
    """Opacities whose `total_alphas` lives on the device until read (opacities/base.py:4-28 keeps a host array)."""

    def __init__(self, shape):
        self.opacities_dict = LazyOpacitiesDict()
        self._resident = {}
        self._total_dev = None
        self._total_host = None
        self._total_twin = None  # device plane written by the fused step
        self._shape = shape
        self._device_bytes = 0

    @property
    def total_alphas(self):
        if self._total_host is None and getattr(self, "_discarded", False):
            raise RuntimeError("total_alphas was released with release_device(materialize=False) before it was read")
        if self._total_host is None:
            self._total_host = self._total_twin.numpy() if self._total_twin is not None else np.zeros(self._shape)  # (no twin: nothing was computed — np.zeros like the reference's constructor)
            if self._total_twin is not None:
                self._total_dev = (self._total_host.copy(), self._total_twin)
        return self._total_host

    @total_alphas.setter
    def total_alphas(self, value):
        self._total_host = value

    def total_alphas_device(self, ctx):
        if self._total_host is None and self._total_twin is not None:
            return self._total_twin
        return super().total_alphas_device(ctx)


_TRACKED_CLASSES = {}


def _tracked_class(field_cls):
    """field_cls with `I_nus` (radiation_field/base.py:64-68) materialising from the device on first read: the array is
    N_theta times the size of F_nu (68 MB at 7634 frequencies, 20 angles), and most callers never look at it."""
    cls = _TRACKED_CLASSES.get(field_cls)
    if cls is None:
        def get(self):
            if self._I_host is None:
                self._I_host = self._I_dev.numpy()
            return self._I_host

        def put(self, value):
            self._I_host = value

        cls = _TRACKED_CLASSES[field_cls] = type("Fused" + field_cls.__name__, (field_cls,), {"I_nus": property(get, put), "_I_host": None, "_I_dev": None})
    return cls
