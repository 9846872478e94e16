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
    """A table edited IN PLACE between two calls is the same object: the digest of its values kept beside the identity makes the
    cached derivation miss, as the reference — which recomputes everything (radiation_field/base.py:71-117) — would see the edit.
    ANY value counts (round-4 review: the three sampled frequencies of the earlier fingerprint let most edits through)."""
    fused.clear_cache()
    table = pd.DataFrame({"nu": [3.0, 1.0, 2.0], "A_ul": [1e7, 2e7, 3e7]})
    n = []
    f = lambda: fused._memo("edit", (table,), None, lambda: n.append(1) or float(table["nu"].sum() + table["A_ul"].sum()))  # noqa: E731
    assert f() == 6.0 + 6e7 and f() == 6.0 + 6e7 and len(n) == 1
    table.loc[1, "nu"] = 10.0
    assert f() == 15.0 + 6e7 and len(n) == 2
    table.iloc[1, 1] *= 2.0  # a value the old fingerprint never looked at
    assert f() == 15.0 + 8e7 and f() == 15.0 + 8e7 and len(n) == 3
    table.drop(index=0, inplace=True)  # rows dropped in place: another shape
    assert f() == 12.0 + 7e7 and len(n) == 4
    dens = pd.Series(np.arange(5.0))
    g = lambda: fused._memo("series", (dens,), None, lambda: n.append(1) or float(dens.sum()))  # noqa: E731
    assert g() == 10.0 and g() == 10.0
    dens *= 2.0
    assert g() == 20.0
    dens.iloc[2] = 0.0
    assert g() == 16.0
    frame = pd.DataFrame(np.ones((4, 3)))
    h = lambda: fused._memo("frame", (frame,), None, lambda: float(frame.to_numpy().sum()))  # noqa: E731
    assert h() == 12.0
    frame.iloc[:, 0] *= 2.0
    assert h() == 16.0
    frame[1] = 5.0  # a column replaced
    assert h() == 32.0
    fused.clear_cache()


def test_memo_private_objects_are_identity_only_and_cache_can_be_switched_off(monkeypatch):
    fused.clear_cache()
    own = pd.Series([1.0, 2.0])
    n = []
    f = lambda: fused._memo("own", (own,), None, lambda: n.append(1) or float(own.sum()), private=(0,))  # noqa: E731
    assert f() == 3.0
    own.iloc[0] = 5.0  # this module's own derivations are never edited; if one were, it would not be looked at
    assert f() == 3.0 and len(n) == 1
    monkeypatch.setattr(fused, "CACHE", False)
    assert f() == 7.0 and f() == 7.0 and len(n) == 3
    fused.clear_cache()


def test_witness_of_object_columns_and_unknown_objects():
    a = pd.DataFrame({"molecule": ["TiO", "H2O", "TiO"], "nu": [1.0, 2.0, 3.0]})
    w0 = fused._witness(a)
    assert w0 == fused._witness(a)
    a.loc[1, "molecule"] = "CO"
    assert w0 != fused._witness(a)
    assert fused._witness(object()) != fused._witness(object())  # nothing to read: never a hit
    idx = pd.MultiIndex.from_tuples([(1, 0), (2, 0)])
    assert fused._witness(idx) == fused._witness(idx) and fused._witness(idx) != fused._witness(idx.copy())


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


def test_witnesses_are_per_call_and_per_thread():
    """The once-per-call witness table of try_fused is thread-local: a call on another thread neither sees nor clears it."""
    import threading

    from stardis_amd.radiation_field import fused_cache as FC

    a = np.arange(8.0)
    seen_in_thread, errors = [], []

    def other():
        try:
            assert getattr(FC._CALL, "seen", None) is None  # nothing leaks across threads
            with FC.one_call():
                FC._witness_once(a)
                seen_in_thread.append(len(FC._CALL.seen))
        except Exception as e:  # noqa: BLE001
            errors.append(e)

    with FC.one_call():
        w = FC._witness_once(a)
        t = threading.Thread(target=other)
        t.start(), t.join()
        assert not errors and seen_in_thread == [1]
        assert FC._CALL.seen and FC._witness_once(a) is w  # still this call's table, untouched by the other thread's exit
        a[0] = 5.0
        assert FC._witness_once(a) is w  # within ONE call an object is read once
    assert getattr(FC._CALL, "seen", None) is None
    assert FC._witness_once(a) != w  # outside a call: always the current content
