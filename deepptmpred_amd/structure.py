"""Structure features of predict.py:172-321 without PyRosetta or Biopython.

``StructureCalculator`` has the interface of the reference's ``PyRosettaCalculator``
(``calculate_features(residue_number) -> 8 floats``, attributes ``plDDT_values`` and
``res_sasa``) and the same feature order (predict.py:308-318):

    [sasa, phi, psi, H, E, L, local_plDDT, avg_plDDT]      raw A^2, degrees, 0..100

What is restated exactly, with the reference's quirks kept:
  * ``plDDT_values`` = the B-factor of every atom of the file in file order
    (predict.py:258-261); ``local_plDDT`` = ``plDDT_values[residue_number - 1]``, i.e. the
    list of ATOMS indexed by a RESIDUE number (predict.py:299-303); ``avg_plDDT`` = mean over
    atoms (predict.py:304);
  * secondary structure: the reference reads ``pose.secstruct(i)`` of a pose that never had
    DSSP run on it (predict.py:203, 293), which Rosetta initialises to 'L' for every
    residue, so the one-hot is always [0, 0, 1];
  * phi / psi in degrees from the backbone atoms, 0.0 where the torsion is undefined (first
    residue's phi, last residue's psi, chain breaks), Rosetta's convention;
  * residue numbers are pose numbers: 1..N in file order;
  * a failing residue gives [0, 0, 0, 0, 0, 1, 85, 85] (predict.py:319-321).
What is NOT reproducible: SASA.  Rosetta's ``calc_per_atom_sasa`` (probe 1.4 A) works on
a pose with hydrogens built, its own 162-dot sphere and its legacy radii table; here it is
a Shrake-Rupley surface over the heavy atoms of the file with united-atom radii, same
probe.  Values track Rosetta's but do not match digit for digit (parity unpinned).
"""
from __future__ import annotations

import math
import os
from typing import Dict, List, Optional, Tuple

import numpy as np

PROBE_RADIUS = 1.4            # predict.py:232
N_SPHERE_POINTS = 162         # dots per atom, the count Rosetta's sasa code uses
FAILED_ROW = np.array([0.0, 0.0, 0.0, 0, 0, 1, 85.0, 85.0])   # predict.py:321

# united-atom (hydrogens folded in) van der Waals radii, Angstrom
_RADIUS_BY_ELEMENT = {"C": 1.87, "N": 1.65, "O": 1.40, "S": 1.85, "P": 1.90, "SE": 1.90, "H": 0.0}
_CARBONYL_OR_AROMATIC_C = 1.76
_AROMATIC_RES = {"PHE", "TYR", "TRP", "HIS"}


def _atom_radius(resname: str, atom: str, element: str) -> float:
    if element == "C":
        if atom == "C":
            return _CARBONYL_OR_AROMATIC_C
        if resname in _AROMATIC_RES and atom not in ("CA", "CB"):
            return _CARBONYL_OR_AROMATIC_C
        if (resname, atom) in (("ASP", "CG"), ("ASN", "CG"), ("GLU", "CD"), ("GLN", "CD"), ("ARG", "CZ")):
            return _CARBONYL_OR_AROMATIC_C
    return _RADIUS_BY_ELEMENT.get(element, 1.80)


def sphere_points(n: int = N_SPHERE_POINTS) -> np.ndarray:
    """n points on the unit sphere, golden-spiral placement (deterministic)."""
    k = np.arange(n, dtype=np.float64) + 0.5
    z = 1.0 - 2.0 * k / n
    r = np.sqrt(np.maximum(0.0, 1.0 - z * z))
    phi = k * (math.pi * (3.0 - math.sqrt(5.0)))
    return np.stack([r * np.cos(phi), r * np.sin(phi), z], axis=1)


class PdbAtoms:
    """ATOM/HETATM records of the first model of a PDB file, column-parsed."""

    def __init__(self, path: str):
        names: List[str] = []
        resnames: List[str] = []
        keys: List[Tuple[str, str]] = []
        elements: List[str] = []
        xyz: List[Tuple[float, float, float]] = []
        bfac: List[float] = []
        hetero: List[bool] = []
        seen = set()
        with open(path, "r") as fh:
            for line in fh:
                rec = line[:6]
                if rec.startswith("ENDMDL"):
                    break
                if not (rec.startswith("ATOM") or rec.startswith("HETATM")):
                    continue
                name = line[12:16].strip()
                key = (line[21], line[22:27])
                if (key, name) in seen:          # alternate locations: first one kept, counted once
                    continue
                seen.add((key, name))
                names.append(name)
                resnames.append(line[17:20].strip())
                keys.append(key)
                xyz.append((float(line[30:38]), float(line[38:46]), float(line[46:54])))
                bfac.append(float(line[60:66]) if len(line) >= 66 and line[60:66].strip() else 0.0)
                el = line[76:78].strip().upper() if len(line) >= 78 else ""
                if not el:
                    el = "".join(c for c in name if c.isalpha())[:1].upper()
                elements.append(el)
                hetero.append(rec.startswith("HETATM"))
        self.names = names
        self.resnames = resnames
        self.elements = elements
        self.xyz = np.asarray(xyz, dtype=np.float64).reshape(-1, 3)
        self.bfactors = np.asarray(bfac, dtype=np.float64)
        self.hetero = np.asarray(hetero, dtype=bool)
        # pose numbering: residues 1..N in file order
        self.residue_index = np.zeros(len(names), dtype=np.int64)
        self.residue_keys: List[Tuple[str, str]] = []
        self.residue_names: List[str] = []
        last = None
        for i, k in enumerate(keys):
            if k != last:
                self.residue_keys.append(k)
                self.residue_names.append(resnames[i])
                last = k
            self.residue_index[i] = len(self.residue_keys)
        self.n_residues = len(self.residue_keys)


def dihedral_deg(p0, p1, p2, p3) -> float:
    """IUPAC dihedral angle p0-p1-p2-p3 in degrees, (-180, 180]."""
    b0 = p0 - p1
    b1 = p2 - p1
    b2 = p3 - p2
    b1n = b1 / np.linalg.norm(b1)
    v = b0 - np.dot(b0, b1n) * b1n
    w = b2 - np.dot(b2, b1n) * b1n
    x = np.dot(v, w)
    y = np.dot(np.cross(b1n, v), w)
    return float(np.degrees(np.arctan2(y, x)))


def backbone_torsions(atoms: PdbAtoms, max_peptide_bond: float = 2.0) -> Tuple[np.ndarray, np.ndarray]:
    """phi[i], psi[i] for pose residues 1..N (index 0 unused).  0.0 where undefined."""
    n = atoms.n_residues
    bb: List[Dict[str, np.ndarray]] = [dict() for _ in range(n + 1)]
    for i, name in enumerate(atoms.names):
        if name in ("N", "CA", "C") and not atoms.hetero[i]:
            bb[atoms.residue_index[i]].setdefault(name, atoms.xyz[i])
    phi = np.zeros(n + 1)
    psi = np.zeros(n + 1)

    def bonded(a, b):
        return ("C" in bb[a] and "N" in bb[b] and atoms.residue_keys[a - 1][0] == atoms.residue_keys[b - 1][0]
                and np.linalg.norm(bb[a]["C"] - bb[b]["N"]) < max_peptide_bond)

    for r in range(1, n + 1):
        if not all(k in bb[r] for k in ("N", "CA", "C")):
            continue
        if r > 1 and bonded(r - 1, r):
            phi[r] = dihedral_deg(bb[r - 1]["C"], bb[r]["N"], bb[r]["CA"], bb[r]["C"])
        if r < n and bonded(r, r + 1):
            psi[r] = dihedral_deg(bb[r]["N"], bb[r]["CA"], bb[r]["C"], bb[r + 1]["N"])
    return phi, psi


def shrake_rupley(xyz: np.ndarray, radii: np.ndarray, probe: float = PROBE_RADIUS,
                  n_points: int = N_SPHERE_POINTS, backend: str = "numpy", device: int = 0) -> np.ndarray:
    """Solvent-accessible area per atom (A^2): fraction of the n_points test points on the
    sphere of radius r_i + probe that lie outside every neighbour's expanded sphere.

    ``backend="hip"`` runs the same count on the GPU (``dptm_sasa`` in libdptm.so, fp64, same
    operation order: bit-identical areas) and raises if the library or the GPU is missing;
    ``"numpy"`` is the host implementation, the counterpart of the reference's host-side call."""
    if backend == "hip":
        from . import _native
        return _native.sasa(xyz, radii, probe, sphere_points(n_points), device)
    if backend != "numpy":
        raise ValueError("backend must be 'numpy' or 'hip'")
    from scipy.spatial import cKDTree
    n = len(xyz)
    area = np.zeros(n)
    if n == 0:
        return area
    unit = sphere_points(n_points)
    ext = radii + probe
    tree = cKDTree(xyz)
    reach = float(ext.max())
    for i in range(n):
        if radii[i] <= 0.0:
            continue
        nb = np.asarray(tree.query_ball_point(xyz[i], ext[i] + reach), dtype=np.int64)
        nb = nb[nb != i]
        if nb.size:
            d = np.linalg.norm(xyz[nb] - xyz[i], axis=1)
            nb = nb[(d < ext[i] + ext[nb]) & (radii[nb] > 0.0)]
        pts = xyz[i] + ext[i] * unit
        if nb.size:
            d2 = ((pts[:, None, :] - xyz[nb][None, :, :]) ** 2).sum(axis=2)
            exposed = (d2 >= (ext[nb] ** 2)[None, :]).all(axis=1)
            frac = exposed.sum() / float(n_points)
        else:
            frac = 1.0
        area[i] = 4.0 * math.pi * ext[i] ** 2 * frac
    return area


class StructureCalculator:
    """Stand-in for ``PyRosettaCalculator`` (predict.py:172-321)."""

    def __init__(self, pdb_path: Optional[str] = None, verbose: bool = False, sasa_backend: str = "numpy", device: int = 0):
        self.pose = None                    # the parsed atoms (``PdbAtoms``), None when no file was given
        self.plDDT_values: List[float] = []
        self.res_sasa: Dict[int, Tuple[float, float, float]] = {}
        self._phi = np.zeros(1)
        self._psi = np.zeros(1)
        if not pdb_path:
            return
        if not os.path.exists(pdb_path):
            raise FileNotFoundError(f"PDB file does not exist: {pdb_path}")     # predict.py:194-195
        atoms = PdbAtoms(pdb_path)
        self.pose = atoms
        radii = np.array([_atom_radius(atoms.resnames[i], atoms.names[i], atoms.elements[i])
                          for i in range(len(atoms.names))])
        # the GPU backend fails loudly (missing library, no device); only the host computation gets the
        # reference's "SASA failed -> zeros" treatment
        gpu_area = shrake_rupley(atoms.xyz, radii, backend="hip", device=device) if sasa_backend == "hip" else None
        try:
            per_atom = gpu_area if gpu_area is not None else shrake_rupley(atoms.xyz, radii, backend=sasa_backend)
            per_res = np.bincount(atoms.residue_index, weights=per_atom, minlength=atoms.n_residues + 1)
            for r in range(1, atoms.n_residues + 1):
                total = float(per_res[r])
                self.res_sasa[r] = (total, total * 0.3, total * 0.7)          # predict.py:237-241
        except Exception:                                                       # predict.py:250-252
            self.res_sasa = {r: (0.0, 0.0, 0.0) for r in range(1, atoms.n_residues + 1)}
        self.plDDT_values = [float(b) for b in atoms.bfactors]                 # predict.py:258-261
        self._phi, self._psi = backbone_torsions(atoms)
        if verbose:
            print(f"Total number of residues: {atoms.n_residues}")
            print(f"Extracted {len(self.plDDT_values)} B-factors (plDDT)")
            print(f"Average plDDT: {np.mean(self.plDDT_values):.2f}")

    def total_residue(self) -> int:
        return 0 if self.pose is None else self.pose.n_residues

    def phi(self, residue_number: int) -> float:
        return float(self._phi[residue_number])

    def psi(self, residue_number: int) -> float:
        return float(self._psi[residue_number])

    @staticmethod
    def secstruct(residue_number: int) -> str:
        """'L' for every residue: what ``pose.secstruct`` holds when DSSP was never run."""
        return "L"

    def calculate_features(self, residue_number: int) -> np.ndarray:
        try:
            if self.pose is None:
                raise ValueError("no PDB structure loaded")
            if residue_number < 1 or residue_number > self.total_residue():
                raise ValueError(f"residue {residue_number} out of range [1, {self.total_residue()}]")
            total_sasa, _, _ = self.res_sasa.get(residue_number, (0.0, 0.0, 0.0))
            ss = self.secstruct(residue_number)
            onehot = [1, 0, 0] if ss == "H" else ([0, 1, 0] if ss == "E" else [0, 0, 1])
            pl = self.plDDT_values
            local = pl[residue_number - 1] if pl and residue_number <= len(pl) else 85.0
            avg = float(np.mean(pl)) if pl else 85.0
            return np.array([total_sasa, self.phi(residue_number), self.psi(residue_number),
                             onehot[0], onehot[1], onehot[2], local, avg])
        except Exception:
            return FAILED_ROW.copy()

    def features(self, positions) -> np.ndarray:
        """Rows for many residues at once, ``float64[N, 8]``."""
        return np.stack([self.calculate_features(int(p)) for p in positions]) if len(positions) else np.zeros((0, 8))
