"""Structure-feature stage (predict.py:172-321 without PyRosetta): the vectorised product code
against the brute-force oracle's fixtures, and the reference's own printed numbers for AKT1."""
import os

import numpy as np
import pytest

from deepptmpred_amd import structure as S
from deepptmpred_amd.predictor import DEFAULT_STRUCT_ROW, PredictConfig, PTMPredictDataLoader


@pytest.fixture(scope="module")
def fx(golden_dir):
    return np.load(os.path.join(golden_dir, "structure_akt1.npz"))


@pytest.fixture(scope="module")
def head(golden_dir):
    return S.StructureCalculator(os.path.join(golden_dir, "akt1_head60.pdb"))


def test_head60_sasa_matches_bruteforce_oracle(head, fx):
    got = np.array([0.0] + [head.res_sasa[r][0] for r in range(1, 61)])
    assert np.abs(got - fx["head_sasa_per_residue"]).max() < 1e-9
    # the 30 % / 70 % backbone / side-chain split of predict.py:238-241
    t, bb, sc = head.res_sasa[7]
    assert bb == pytest.approx(0.3 * t) and sc == pytest.approx(0.7 * t)


def test_head60_torsions(head, fx):
    phi = np.array([head.phi(r) for r in range(1, 61)])
    psi = np.array([head.psi(r) for r in range(1, 61)])
    assert np.abs(phi - fx["head_phi"][1:]).max() < 1e-9
    assert np.abs(psi - fx["head_psi"][1:]).max() < 1e-9
    assert head.phi(1) == 0.0 and head.psi(60) == 0.0       # undefined at the termini
    # the excerpt ends mid-chain, so up to residue 59 the torsions are those of the whole protein
    assert np.abs(phi[:59] - fx["full_phi"][1:60]).max() < 1e-9
    assert np.abs(psi[:59] - fx["full_psi"][1:60]).max() < 1e-9


def test_feature_row_layout_and_defaults(head):
    row = head.calculate_features(10)
    assert row.shape == (8,) and row.dtype == np.float64
    assert row[0] == head.res_sasa[10][0] and row[1] == head.phi(10) and row[2] == head.psi(10)
    assert list(row[3:6]) == [0, 0, 1]                        # secstruct is 'L' without DSSP (predict.py:293)
    assert row[7] == pytest.approx(np.mean(head.plDDT_values))
    # residue number indexes the per-ATOM B-factor list (predict.py:299-303)
    assert row[6] == head.plDDT_values[9]
    for bad in (0, 61, -3):
        assert np.array_equal(head.calculate_features(bad), [0, 0, 0, 0, 0, 1, 85.0, 85.0])   # predict.py:321
    assert np.array_equal(S.StructureCalculator(None).calculate_features(1), [0, 0, 0, 0, 0, 1, 85.0, 85.0])
    with pytest.raises(FileNotFoundError):
        S.StructureCalculator("/nonexistent/file.pdb")


def test_known_answers_of_the_reference_printout(fx):
    """predict.ipynb cell 4 for AF-P31749-F1-model_v4.pdb: 480 residues, 3919 B-factors,
    'Average plDDT: 82.95'; and the atom-indexed local plDDT of residue 129 is atom #129's 64.70."""
    b = fx["full_bfactors"]
    assert len(b) == 3919 and len(fx["full_phi"]) - 1 == 480
    assert "%.2f" % b.mean() == "82.95"
    assert b[129 - 1] == 64.70


def test_sasa_properties():
    # an isolated atom exposes its whole expanded sphere; two coincident atoms each keep nothing of the
    # other's interior; far-apart atoms do not interact
    r = np.array([1.87])
    one = S.shrake_rupley(np.zeros((1, 3)), r)
    assert one[0] == pytest.approx(4 * np.pi * (1.87 + 1.4) ** 2)
    far = S.shrake_rupley(np.array([[0.0, 0, 0], [50.0, 0, 0]]), np.array([1.87, 1.4]))
    assert far[0] == pytest.approx(one[0])
    near = S.shrake_rupley(np.array([[0.0, 0, 0], [1.5, 0, 0]]), np.array([1.87, 1.87]))
    assert 0 < near[0] < one[0] and near[0] == pytest.approx(near[1], rel=0.05)
    assert S.shrake_rupley(np.zeros((0, 3)), np.zeros(0)).shape == (0,)


def test_dihedral_sign_and_range():
    p = [np.array(v, dtype=float) for v in ([1, 0, 0], [0, 0, 0], [0, 0, 1], [0, 1, 1])]
    assert S.dihedral_deg(*p) == pytest.approx(90.0)
    p[3] = np.array([0.0, -1, 1])
    assert S.dihedral_deg(*p) == pytest.approx(-90.0)
    p[3] = np.array([-1.0, 0, 1])
    assert abs(S.dihedral_deg(*p)) == pytest.approx(180.0)


def test_loader_uses_structure_file(golden_dir, tmp_path):
    cfg = PredictConfig("phosphorylation", str(tmp_path), model_path="unused")
    dl = PTMPredictDataLoader(cfg)
    seq = "M" * 60
    pos = [3, 17, 60]
    _, st_default, _, _ = dl.prepare_protein_data("X", seq, pos)
    assert np.array_equal(st_default, np.tile(DEFAULT_STRUCT_ROW, (3, 1)))      # predict.py:518-526
    _, st, _, _ = dl.prepare_protein_data("X", seq, pos, pdb_path=os.path.join(golden_dir, "akt1_head60.pdb"))
    assert st.shape == (3, 8) and np.array_equal(st[:, 5], [1, 1, 1]) and (st[:, 0] > 0).all()
    assert np.array_equal(st[1], dl.rosetta_calc.calculate_features(17))
