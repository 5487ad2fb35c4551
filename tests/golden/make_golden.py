"""Regenerates the committed fixtures under tests/golden/.

Run from the repo root IN THE BUILD CONTAINER (it reads the reference's shipped data
files under /root/reference, which do not exist on the GPU box):

    python tests/golden/make_golden.py

Outputs
  synthetic_s7.npz   64 synthetic sites (seed 1234) + weights seed 7: inputs, fp64 oracle
                     probabilities and per-stage activations.
  proteins.npz       the two shipped proteins (AKT1 P31749, tau P10636): sequence, target
                     positions, the float32 ESM rows of those positions (the shipped
                     float64 tables are float32-exact), and the expected padded windows.
Everything numeric comes from oracle/ptm_oracle.py (the CPU restatement); the ESM rows and
sequences are data files of the reference (pred/custom_esm/*.npz), not code.
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from deepptmpred_amd import spec  # noqa: E402
from oracle import ptm_oracle as O  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def synthetic():
    w = spec.synthetic_weights(7)
    x = spec.synthetic_inputs(64, 1234)
    # force the edge cases: all-pad window, left pad, right pad, unknown->20 already as 20
    x[0][0, :] = 20; x[1][0, :] = 20; x[2][0, :] = 20
    x[2][1, :25] = 20; x[1][1, :16] = 20; x[0][1, :10] = 20
    x[2][2, 26:] = 20; x[1][2, 17:] = 20; x[0][2, 11:] = 20
    st = {}
    with np.errstate(over="ignore"):
        p = O.forward(w, x, np.float64, st)
    keep = {"stage_" + k.replace(".", "_"): v.astype(np.float64) for k, v in st.items()
            if k.endswith(".pool") or k in ("struct", "esm", "merged", "trunk")}
    np.savez_compressed(os.path.join(HERE, "synthetic_s7.npz"), tok21=x[0], tok33=x[1], tok51=x[2],
                        struct=x[3], esm=x[4], proba=p, **keep)
    print("synthetic_s7: p1 range", p[:, 1].min(), p[:, 1].max())


def proteins():
    out = {}
    meta = {}
    for pid, targets in (("P31749", {"phospho": ["S", "T"], "ubiq": ["K"]}), ("P10636", {"phospho": ["S", "T"]})):
        d = np.load(os.path.join(REF, "pred", "custom_esm", f"{pid}_full_esm.npz"))
        seq = str(d["sequence"])
        feats = d["features"]
        assert np.array_equal(feats.astype(np.float32).astype(np.float64), feats)
        meta[pid] = {"sequence": seq, "esm_rows": int(feats.shape[0]), "sets": {}}
        for tag, aa in targets.items():
            pos = O.target_positions(seq, aa)
            rows = O.prepare_esm_features(feats, pos).astype(np.float32)
            out[f"{pid}_{tag}_positions"] = np.asarray(pos, dtype=np.int32)
            out[f"{pid}_{tag}_esm"] = rows
            wins = {}
            for w in (21, 33, 51):
                ws = O.get_sequence_windows(seq, pos, w)
                out[f"{pid}_{tag}_tok{w}"] = O.encode_sequences(ws).astype(np.uint8)
                wins[str(w)] = ws
            meta[pid]["sets"][tag] = {"n": len(pos), "windows": wins}
    np.savez_compressed(os.path.join(HERE, "proteins.npz"), **out)
    with open(os.path.join(HERE, "proteins.json"), "w") as fh:
        json.dump(meta, fh)
    print({k: {t: v["n"] for t, v in m["sets"].items()} for k, m in meta.items()})


def structure():
    """Structure-feature fixtures from the reference's shipped AlphaFold model of AKT1
    (data/AF-P31749-F1-model_v4.pdb, a data file): the ATOM records of residues 1..60 as a small
    PDB, the brute-force oracle's SASA / torsions for it, and for the whole file the B-factor
    list, the oracle's torsions and the numbers predict.ipynb prints (480 residues, 3919
    B-factors, average plDDT 82.95)."""
    from deepptmpred_amd import structure as S
    from oracle import structure_oracle as SO
    src = os.path.join(REF, "data", "AF-P31749-F1-model_v4.pdb")
    head = os.path.join(HERE, "akt1_head60.pdb")
    with open(src) as fin, open(head, "w") as fout:
        for line in fin:
            if line.startswith("ATOM") and int(line[22:26]) <= 60:
                fout.write(line)
        fout.write("END\n")
    atoms = SO.read_atoms(head)
    xyz = np.array([a[4] for a in atoms])
    radii = np.array([S._atom_radius(a[1], a[0], a[6]) for a in atoms])
    per_atom = SO.sasa_per_atom(xyz, radii, S.sphere_points(S.N_SPHERE_POINTS), S.PROBE_RADIUS)
    res_of_atom = []
    last, idx = None, 0
    for a in atoms:
        if (a[2], a[3]) != last:
            idx += 1
            last = (a[2], a[3])
        res_of_atom.append(idx)
    sasa_res = np.bincount(np.array(res_of_atom), weights=per_atom, minlength=61)
    phi_h, psi_h = SO.torsions(atoms)
    full = SO.read_atoms(src)
    phi_f, psi_f = SO.torsions(full)
    bf = np.array([a[5] for a in full])
    np.savez_compressed(os.path.join(HERE, "structure_akt1.npz"), head_sasa_per_residue=sasa_res,
                        head_phi=np.array(phi_h), head_psi=np.array(psi_h), full_phi=np.array(phi_f),
                        full_psi=np.array(psi_f), full_bfactors=bf)
    print("structure:", len(atoms), "atoms in head60;", len(full), "atoms,", len(phi_f) - 1, "residues, mean B", bf.mean())


if __name__ == "__main__":
    synthetic()
    proteins()
    structure()
