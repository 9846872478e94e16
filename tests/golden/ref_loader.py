"""Load the reference's hot-path modules un-jitted (THIS container only).

Test tooling, never shipped or imported by the product or by anything that runs
on the GPU box.  Run under ``/opt/conda/bin/python3.9`` (has astropy 4.3.1);
numba / tardis are absent, so they are replaced by inert stand-ins *in
sys.modules only* and the reference source is executed by CPython from where it
lies under ``/root/reference``.  Nothing from the reference is copied.

Recipe follows SURVEY.md Appendix A.
"""
import functools
import importlib
import sys
import types
import warnings

import numpy as np

REF_ROOT = "/root/reference/stardis"

_NUMPY_GONE = (
    "asscalar alen msort sometrue alltrue product cumproduct round_ float_ complex_ unicode_ in1d "
    "trapz row_stack find_common_type cast source safe_eval who set_string_function lookfor deprecate "
    "byte_bounds issubclass_ issctype maximum_sctype obj2sctype sctype2char sctypes issubsctype "
    "set_numeric_ops compare_chararrays fastCopyAndTranspose recfromcsv recfromtxt mat asfarray "
    "get_array_wrap DataSource nbytes disp add_newdoc_ufunc tracemalloc_domain"
).split()


def _strip_subclasses(fn):
    """numba hands plain ndarrays to a jitted function (astropy Quantity loses its unit)."""

    @functools.wraps(fn)
    def call(*args, **kw):
        args = [
            np.asarray(a) if isinstance(a, np.ndarray) and type(a) is not np.ndarray else a
            for a in args
        ]
        return fn(*args, **kw)

    return call


def _passthrough_decorator(*a, **k):
    if len(a) == 1 and callable(a[0]) and not k:
        return _strip_subclasses(a[0])
    return lambda f: _strip_subclasses(f)


def _vectorize_decorator(*a, **k):
    if len(a) == 1 and callable(a[0]) and not k:
        return np.vectorize(a[0])
    return lambda f: np.vectorize(f)


def _module(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


_loaded = None


def load():
    """Returns a namespace with the reference modules: ob, vg, br, rt, bb, ut, rf, op, mk."""
    global _loaded
    if _loaded is not None:
        return _loaded
    warnings.filterwarnings("ignore")
    for n in _NUMPY_GONE:
        if not hasattr(np, n):
            setattr(np, n, (lambda *a, **k: None))

    cuda = _module("numba.cuda", is_available=lambda: False, jit=_passthrough_decorator, grid=lambda n: 0)
    _module(
        "numba",
        njit=_passthrough_decorator,
        jit=_passthrough_decorator,
        vectorize=_vectorize_decorator,
        prange=range,
        get_thread_id=lambda: 0,
        set_num_threads=lambda n: None,
        config=types.SimpleNamespace(NUMBA_DEFAULT_NUM_THREADS=1),
        cuda=cuda,
    )

    def species(text):
        """Stand-in for tardis.util.base.species_string_to_tuple (TARDIS is not on this box): element symbol, then the
        spectroscopic stage as a Roman numeral or in digits — "H I" and "H 1" are both (1, 0)."""
        symbol, stage = text.split()
        z = {"h": 1, "he": 2}[symbol.lower()]
        number = int(stage) if stage.isdigit() else {"i": 1, "ii": 2, "iii": 3}[stage.lower()]
        return z, number - 1

    for n in ("tardis", "tardis.util", "tardis.io", "tardis.model", "tardis.model.matter"):
        _module(n)
    _module(
        "tardis.util.base",
        species_string_to_tuple=species,
        element_symbol2atomic_number=None,
        atomic_number2element_symbol=None,
    )
    _module("tardis.io.util", HDFWriterMixin=type("HDFWriterMixin", (), {}))
    _module(
        "tardis.model.matter.composition",
        Composition=type("Composition", (), {"__init__": lambda self, *a, **k: None}),
    )

    pkgs = {
        "stardis": "",
        "stardis.radiation_field": "/radiation_field",
        "stardis.radiation_field.opacities": "/radiation_field/opacities",
        "stardis.radiation_field.opacities.opacities_solvers": "/radiation_field/opacities/opacities_solvers",
        "stardis.radiation_field.radiation_field_solvers": "/radiation_field/radiation_field_solvers",
        "stardis.radiation_field.source_functions": "/radiation_field/source_functions",
        "stardis.io": "/io",
        "stardis.io.model": "/io/model",
        "stardis.model": "/model",
        "stardis.model.geometry": "/model/geometry",
    }
    for name, sub in pkgs.items():
        m = types.ModuleType(name)
        m.__path__ = [REF_ROOT + sub]
        sys.modules[name] = m

    imp = importlib.import_module
    ns = types.SimpleNamespace(
        ob=imp("stardis.radiation_field.opacities.opacities_solvers.base"),
        vg=imp("stardis.radiation_field.opacities.opacities_solvers.voigt"),
        br=imp("stardis.radiation_field.opacities.opacities_solvers.broadening"),
        ut=imp("stardis.radiation_field.opacities.opacities_solvers.util"),
        rt=imp("stardis.radiation_field.radiation_field_solvers.base"),
        bb=imp("stardis.radiation_field.source_functions.blackbody"),
        op=imp("stardis.radiation_field.opacities.base"),
        mk=imp("stardis.io.model.marcs"),
    )
    # radiation_field/base.py imports names from the package __init__ files, which the
    # synthetic package objects above do not execute; provide them before importing it.
    sys.modules["stardis.radiation_field.opacities"].Opacities = ns.op.Opacities
    sys.modules["stardis.radiation_field.opacities.opacities_solvers"].calc_alphas = ns.ob.calc_alphas
    sys.modules["stardis.radiation_field.radiation_field_solvers"].raytrace = ns.rt.raytrace
    ns.rf = imp("stardis.radiation_field.base")
    _loaded = ns
    return ns


_plasma = None


def load_plasma():
    """The reference's plasma property classes (stardis/plasma/base.py, molecules.py).  TARDIS is absent: its plasma
    base classes are replaced by empty stand-ins in sys.modules, which is enough because every `calculate` method
    used here is a self-contained numpy/pandas/astropy function of its arguments."""
    global _plasma
    if _plasma is not None:
        return _plasma
    load()
    blank = type("ProcessingPlasmaProperty", (), {})
    for n in ("tardis.plasma", "tardis.plasma.properties", "tardis.opacities"):
        _module(n)
    _module("tardis.plasma.base", BasePlasma=type("BasePlasma", (), {}))
    _module("tardis.plasma.properties.base", ProcessingPlasmaProperty=blank, DataFrameInput=type("DataFrameInput", (), {}))
    _module(
        "tardis.plasma.properties.property_collections",
        **{k: [] for k in ("basic_inputs basic_properties lte_excitation_properties lte_ionization_properties "
                           "non_nlte_properties helium_lte_properties").split()},
    )
    _module("tardis.opacities.tau_sobolev", TauSobolev=type("TauSobolev", (), {}))
    sys.modules["tardis"].plasma = sys.modules["tardis.plasma"]
    m = types.ModuleType("stardis.plasma")
    m.__path__ = [REF_ROOT + "/plasma"]
    sys.modules["stardis.plasma"] = m
    _plasma = types.SimpleNamespace(
        mol=importlib.import_module("stardis.plasma.molecules"),
        base=importlib.import_module("stardis.plasma.base"),
    )
    return _plasma
