"""Host-side pieces of the fused drop-in path that need no GPU: the lazily materialised `opacities_dict` behaves like the plain
dict the reference fills (opacities_solvers/base.py:655-738), the per-object cache keys on identity, the table row order is the
reference's `_in_grid` (:392-397: sort_values("nu") then nu.between(min, max))."""
import numpy as np
import pandas as pd
import pytest

from stardis_amd.radiation_field import fused


def test_lazy_dict_materialises_once_and_behaves_like_a_dict():
    calls = []

    def make(tag, value):
        def run():
            calls.append(tag)
            return value
        return fused._Thunk(run)

    d = fused.LazyOpacitiesDict()
    a, b = np.arange(3.0), np.arange(4.0)
    dict.__setitem__(d, "alpha_bf", make("bf", a))
    dict.__setitem__(d, "alpha_electron", 0)
    dict.__setitem__(d, "alpha_line_at_nu", make("line", b))
    assert list(d.keys()) == ["alpha_bf", "alpha_electron", "alpha_line_at_nu"] and len(d) == 3 and "alpha_bf" in d and not calls
    assert d["alpha_electron"] == 0 and not calls
    assert d["alpha_bf"] is a and d["alpha_bf"] is a and calls == ["bf"]  # once
    assert d.get("missing", 7) == 7 and d.get("alpha_line_at_nu") is b and calls == ["bf", "line"]
    d2 = fused.LazyOpacitiesDict()
    dict.__setitem__(d2, "x", make("x", a))
    dict.__setitem__(d2, "y", make("y", b))
    calls.clear()
    assert [k for k, _ in d2.items()] == ["x", "y"] and all(isinstance(v, np.ndarray) for v in d2.values()) and calls == ["x", "y"]
    d3 = fused.LazyOpacitiesDict()
    dict.__setitem__(d3, "x", make("x3", a))
    plain = dict(d3)  # the copy constructor must not leak a thunk
    assert plain["x"] is a and isinstance(d3.copy()["x"], np.ndarray)
    d4 = fused.LazyOpacitiesDict()
    dict.__setitem__(d4, "x", make("x4", a))
    assert d4.pop("x") is a and "x" not in d4 and d4.pop("x", None) is None
    d4["z"] = b  # plain assignment keeps working (calc_alphas of the general path writes entries this way)
    assert d4["z"] is b


def test_memo_is_keyed_on_object_identity():
    fused.clear_cache()
    builds = []
    t1, t2 = pd.Series([1.0, 2.0]), pd.Series([1.0, 2.0])
    f = lambda obj: fused._memo("t", (obj,), "k", lambda: builds.append(1) or len(builds))  # noqa: E731
    assert f(t1) == 1 and f(t1) == 1 and f(t2) == 2 and f(t1) == 1 and builds == [1, 1]  # equal values, different objects
    assert fused._memo("t", (t1,), "other", lambda: 99) == 99  # the extra key is part of the key
    for k in range(40):  # least recently used entries go first; the cache stays small
        fused._memo("fill", (pd.Series([float(k)]),), None, lambda: k)
    assert len(fused._MEMO) <= fused._MEMO_MAX
    fused.clear_cache()
    assert not fused._MEMO


def test_memo_is_bounded_by_bytes_too(monkeypatch):
    """Sixteen entries of sorted 10^6-line tables would be gigabytes of host memory: the cache also counts the numpy bytes of what
    it holds and drops the least recently used entries beyond MEMO_MAX_BYTES (the newest always stays)."""
    fused.clear_cache()
    monkeypatch.setattr(fused, "MEMO_MAX_BYTES", 3000)
    keys = [pd.Series([float(k)]) for k in range(5)]
    for k, obj in enumerate(keys):
        fused._memo("big", (obj,), None, lambda: {"a": np.zeros(100), "b": (np.zeros(25), [np.zeros(0)])})  # 1000 bytes each
    assert len(fused._MEMO) == 3 and sum(e[2] for e in fused._MEMO.values()) == 3000
    fused._memo("huge", (keys[0],), None, lambda: np.zeros(1000))  # larger than the bound on its own: it stays, alone
    assert len(fused._MEMO) == 1
    fused.clear_cache()


def test_memo_notices_an_edit_in_place():
    """A table edited IN PLACE between two calls is the same object: the cheap content fingerprint next to the identity (shape,
    first, last and summed value) makes the cached derivation miss, as the reference — which recomputes everything — would see
    the edit (round-3 advisor finding)."""
    fused.clear_cache()
    table = pd.DataFrame({"nu": [3.0, 1.0, 2.0], "A_ul": [1e7, 2e7, 3e7]})
    n = []
    f = lambda: fused._memo("edit", (table,), None, lambda: n.append(1) or float(table["nu"].sum()))  # noqa: E731
    assert f() == 6.0 and f() == 6.0 and len(n) == 1
    table.loc[1, "nu"] = 10.0
    assert f() == 15.0 and len(n) == 2
    table.drop(index=0, inplace=True)  # rows dropped in place: another shape
    assert f() == 12.0 and len(n) == 3
    fused.clear_cache()


def test_sorted_line_tables_follow_pandas_in_grid_order():
    rng = np.random.default_rng(5)
    n, nd = 300, 7
    nu = rng.uniform(4.0e14, 5.0e14, n)
    nu[10] = nu[200]  # a tie: both tables must break it the same way
    lines = pd.DataFrame(dict(nu=nu, atomic_number=rng.choice([1, 26], n), ion_number=rng.integers(0, 2, n), ionization_energy=rng.uniform(1, 2, n),
                              level_energy_upper=rng.uniform(0.5, 1.5, n), level_energy_lower=rng.uniform(0, 0.5, n), A_ul=rng.uniform(1e6, 1e9, n)))
    alpha = pd.DataFrame(rng.uniform(0, 1, (n, nd)), columns=np.arange(nd))
    alpha["nu"] = nu
    masses = pd.Series([1.0, 56.0], index=pd.Index([1, 26], name="atomic_number"))
    tab = fused._sorted_line_tables(lines, alpha, masses, False)
    grid = np.linspace(4.9e14, 4.2e14, 50)
    lo, hi = grid.min(), grid.max()
    ref_l = lines.sort_values("nu")
    ref_l = ref_l[ref_l.nu.between(lo, hi)]
    ref_a = alpha.sort_values("nu")
    ref_a = ref_a[ref_a.nu.between(lo, hi)].drop(labels="nu", axis=1).to_numpy()
    i0, i1 = np.searchsorted(tab["nu"], lo, "left"), np.searchsorted(tab["nu"], hi, "right")
    assert np.array_equal(tab["nu"][i0:i1], ref_l.nu.to_numpy()) and np.array_equal(tab["a_ul"][i0:i1], ref_l.A_ul.to_numpy())
    assert np.array_equal(tab["alphas"][i0:i1], ref_a)
    assert np.array_equal(tab["mass"][i0:i1], masses.loc[ref_l.atomic_number].to_numpy())
    lines.loc[3, "nu"] = np.nan
    assert fused._sorted_line_tables(lines, alpha, masses, False) is None  # pandas' NaN handling: general path
    with pytest.raises(KeyError):
        fused._mass_of(masses, np.array([1, 6]))
