"""Small host-side helpers the hot path's callers need and TARDIS normally provides."""
import re

_SYMBOLS = (
    "H He Li Be B C N O F Ne Na Mg Al Si P S Cl Ar K Ca Sc Ti V Cr Mn Fe Co Ni Cu Zn Ga Ge As Se Br Kr Rb Sr Y Zr "
    "Nb Mo Tc Ru Rh Pd Ag Cd In Sn Sb Te I Xe Cs Ba La Ce Pr Nd Pm Sm Eu Gd Tb Dy Ho Er Tm Yb Lu Hf Ta W Re Os Ir "
    "Pt Au Hg Tl Pb Bi Po At Rn Fr Ra Ac Th Pa U"
).split()
_Z = {s.lower(): i + 1 for i, s in enumerate(_SYMBOLS)}
_ROMAN = {"I": 1, "V": 5, "X": 10, "L": 50}


def _roman(text):
    total, prev = 0, 0
    for ch in reversed(text.upper()):
        v = _ROMAN[ch]
        total += -v if v < prev else v
        prev = max(prev, v)
    return total


def species_string_to_tuple(species):
    """'H I' / 'Fe II' / 'Si 2' -> (atomic_number, ion_number); same contract as tardis.util.base.species_string_to_tuple,
    which opacities_solvers/util.py:6,156 calls: the second token is the SPECTROSCOPIC stage whether it is written as a
    Roman numeral or in digits — 'Si II' and 'Si 2' are both (14, 1), 'Si IX' is (14, 8) — so a configuration key such as
    `H_1` names neutral hydrogen."""
    m = re.match(r"^\s*([A-Za-z]+)[\s_]*([IVXLivxl]+|\d+)\s*$", species)
    if not m:
        raise ValueError(f"cannot parse species string {species!r}")
    sym, ion = m.groups()
    if sym.lower() not in _Z:
        raise ValueError(f"unknown element symbol in {species!r}")
    ion_number = (int(ion) if ion.isdigit() else _roman(ion)) - 1
    z = _Z[sym.lower()]
    if ion_number < 0 or ion_number > z:
        raise ValueError(f"species {species!r} has an impossible ionisation stage")
    return z, ion_number
