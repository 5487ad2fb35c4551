"""Host side of ``predict.py``: configuration, feature preparation around the scoring
call and the result table.  Same class and method names as the reference
(pred/train_PTM/predict.py:115-169 ``PredictConfig``, 437-560 ``PTMPredictDataLoader``,
563-612 ``PTMPredictor``), so a script written against them keeps working; the model
call goes to the MI355X library.

Structure rows (predict.py:172-321, PyRosetta in the reference) come from
``deepptmpred_amd.structure.StructureCalculator`` when a PDB path is given, from a
caller-supplied ``struct_features`` array or ``struct_calculator`` object
(``calculate_features(position) -> 8 floats``) otherwise; with none of them every site gets
the reference's default row (predict.py:524-526).
"""
from __future__ import annotations

import os
import re
from typing import List, Optional, Sequence

import numpy as np

from .model import Model, load_model
from .spec import LETTER_CODES, PAD_CODE, PTM_AA_MAP

DEFAULT_STRUCT_ROW = np.array([0.5, 0.0, 0.0, 0, 1, 0, 85.0, 85.0])  # predict.py:526

ONE_LETTER = {  # predict.py:72-78
    "ALA": "A", "ARG": "R", "ASN": "N", "ASP": "D", "CYS": "C", "GLN": "Q", "GLU": "E", "GLY": "G",
    "HIS": "H", "ILE": "I", "LEU": "L", "LYS": "K", "MET": "M", "PHE": "F", "PRO": "P", "SER": "S",
    "THR": "T", "TRP": "W", "TYR": "Y", "VAL": "V", "SEC": "U", "PYL": "O", "ASX": "B", "GLX": "Z",
    "XAA": "X", "XLE": "J",
}


def extract_protein_id_from_pdb_path(pdb_path: str) -> str:
    """predict.py:47-69: UniProt id from an AlphaFold file name."""
    filename = os.path.basename(pdb_path)
    for pat in (r"AF-([A-Za-z0-9]+)-F\d+-model_v\d+", r"AF_([A-Za-z0-9]+)_F\d+_model_v\d+", r"^([A-Za-z0-9]{5,6})\.pdb$"):
        m = re.search(pat, filename)
        if m:
            return m.group(1)
    raise ValueError(f"cannot extract a protein id from file name '{filename}'")


def extract_sequence_from_pdb(pdb_path: str, chain_id: Optional[str] = None) -> str:
    """predict.py:80-112 without Biopython: one letter per residue of the ATOM records
    (hetero residues skipped, unknown names -> 'X'); one chain, or all chains as
    ``A:SEQ/B:SEQ``.  Only the first MODEL is read."""
    sequences: dict = {}
    seen = set()
    with open(pdb_path, "r") as fh:
        for line in fh:
            rec = line[:6]
            if rec.startswith("ENDMDL"):
                break
            if not rec.startswith("ATOM"):
                continue
            chain = line[21]
            key = (chain, line[22:27])          # residue number + insertion code
            if key in seen:
                continue
            seen.add(key)
            sequences.setdefault(chain, []).append(ONE_LETTER.get(line[17:20].strip(), "X"))
    seqs = {k: "".join(v) for k, v in sequences.items()}
    if chain_id is not None:
        return seqs.get(chain_id, "")
    return "/".join(f"{k}:{v}" for k, v in seqs.items())


class PredictConfig:
    """predict.py:115-169, same attribute names.  ``model_path`` may point at a Keras
    ``.h5`` checkpoint or a flat ``.dptm`` file."""

    def __init__(self, ptm_type: str = "phosphorylation", project_root: Optional[str] = None,
                 model_path: Optional[str] = None):
        self.ptm_aa_map = {k: list(v) for k, v in PTM_AA_MAP.items()}
        self.ptm_type = ptm_type
        self.target_aa = self.ptm_aa_map.get(ptm_type, [])
        if not self.target_aa:
            raise ValueError(f"Unsupported PTM type: {ptm_type}")
        if project_root is None:
            raise ValueError("project_root must be provided to PredictConfig")
        self.root_dir = project_root
        self.esm_dir = os.path.join(self.root_dir, "pred", "train_PTM")
        self.custom_esm_dir = os.path.join(self.root_dir, "pred", "custom_esm")
        self.model_path = model_path or os.path.join(self.esm_dir, "model", f"models_{ptm_type}_esm2",
                                                     "ptm_data_210_39_64_best_model.h5")
        self.esm_dim = 1280
        self.struct_features = ["sasa", "phi", "psi", "secstruct", "local_plDDT", "avg_plDDT"]
        self.struct_feature_dim = 8
        self.embed_dim = 64
        self.num_heads = 4
        self.ff_dim = 256
        self.window_sizes = [21, 33, 51]


class PTMPredictDataLoader:
    """predict.py:437-560."""

    def __init__(self, config: PredictConfig, struct_calculator=None, sasa_backend: str = "numpy", device: int = 0):
        self.config = config
        self.rosetta_calc = struct_calculator
        self.sasa_backend = sasa_backend      # "hip": surface areas on the GPU (bit-identical to "numpy")
        self.device = device

    def init_pyrosetta(self, pdb_path):
        """predict.py:442-445 (name kept): lazily builds the structure calculator for a PDB file."""
        if self.rosetta_calc is None:
            from .structure import StructureCalculator
            self.rosetta_calc = StructureCalculator(pdb_path, sasa_backend=self.sasa_backend, device=self.device)

    def load_custom_esm(self, protein_id: str):
        path = os.path.join(self.config.custom_esm_dir, f"{protein_id}_full_esm.npz")
        if not os.path.exists(path):
            return None
        try:
            return np.load(path)["features"]
        except Exception:
            return None

    def prepare_protein_data(self, protein_id, protein_sequence, positions=None, pdb_path=None):
        if positions is None:
            positions = [i + 1 for i, aa in enumerate(protein_sequence) if aa in self.config.target_aa]
        if pdb_path:
            self.init_pyrosetta(pdb_path)
        esm = self._prepare_esm_features(protein_id, protein_sequence, positions)
        st = self._generate_structural_features(positions)
        x_seq = [self._encode_sequences(self._get_sequence_windows(protein_sequence, positions, w))
                 for w in self.config.window_sizes]
        return x_seq, st, esm, positions

    def _prepare_esm_features(self, protein_id, protein_sequence, positions):
        table = self.load_custom_esm(protein_id)
        out = np.zeros((len(positions), self.config.esm_dim))
        if table is not None:
            for i, pos in enumerate(positions):
                if pos - 1 < len(table):
                    out[i] = table[pos - 1]
        return out

    def _generate_structural_features(self, positions):
        feats = np.zeros((len(positions), self.config.struct_feature_dim))
        for i, pos in enumerate(positions):
            row = None
            if self.rosetta_calc is not None:
                try:
                    row = self.rosetta_calc.calculate_features(pos)
                except Exception:
                    row = None
            feats[i] = DEFAULT_STRUCT_ROW if row is None else row
        return feats

    @staticmethod
    def _get_sequence_windows(sequence: str, positions: Sequence[int], window_size: int) -> List[str]:
        half = (window_size - 1) // 2
        out = []
        for pos in positions:
            left = sequence[max(0, pos - 1 - half):pos - 1]
            right = sequence[pos:min(len(sequence), pos + half)]
            out.append("*" * max(0, half - len(left)) + left + sequence[pos - 1] + right + "*" * max(0, half - len(right)))
        return out

    @staticmethod
    def _encode_sequences(sequences: Sequence[str]) -> np.ndarray:
        if not sequences:
            return np.zeros((0, 0), dtype=int)
        enc = np.zeros((len(sequences), len(sequences[0])), dtype=int)
        for i, s in enumerate(sequences):
            enc[i] = [LETTER_CODES.get(c, PAD_CODE) for c in s]
        return enc


class PTMPredictor:
    """predict.py:563-612.  ``predict_ptm_sites`` returns a pandas DataFrame with the
    reference's columns: protein_id, position, residue, probability, prediction."""

    def __init__(self, config: PredictConfig, model: Optional[Model] = None, struct_calculator=None, device: int = 0):
        self.config = config
        # the predictor owns a GPU model, so the structure stage uses the GPU for its surface areas too
        self.data_loader = PTMPredictDataLoader(config, struct_calculator, sasa_backend="hip", device=device)
        self.model = model if model is not None else load_model(config.model_path, device=device)

    def predict_ptm_sites(self, protein_id, protein_sequence, positions=None, pdb_path=None, struct_features=None,
                          fused: bool = True):
        import pandas as pd
        if positions is None:
            positions = [i + 1 for i, aa in enumerate(protein_sequence) if aa in self.config.target_aa]
        positions = list(positions)
        if struct_features is None:
            if pdb_path:
                self.data_loader.init_pyrosetta(pdb_path)
            struct_features = self.data_loader._generate_structural_features(positions)
        if fused and hasattr(self.model, "predict_protein"):
            # windows and ESM rows assembled on the GPU (same indexing as predict.py:487-501, 529-560)
            table = self.data_loader.load_custom_esm(protein_id)
            proba = self.model.predict_protein(protein_sequence, positions, struct_features, table)[:, 1]
        else:
            x_seq, _, esm, positions = self.data_loader.prepare_protein_data(protein_id, protein_sequence, positions, pdb_path)
            proba = self.model.predict(x_seq + [struct_features, esm])[:, 1]
        rows = [{"protein_id": protein_id, "position": pos, "residue": protein_sequence[pos - 1],
                 "probability": p, "prediction": 1 if p >= 0.5 else 0} for pos, p in zip(positions, proba)]
        return pd.DataFrame(rows, columns=["protein_id", "position", "residue", "probability", "prediction"])


    # a scoring call of predict_ptm_sites_batch holds at most this many bytes of ESM tables / this many sites on the host
    BATCH_TABLE_BYTES = 2 << 30
    BATCH_SITES = 1 << 20

    def predict_ptm_sites_batch(self, proteins):
        """The loop a proteome run makes around ``predict_ptm_sites`` (predict.py:584-612) as a few large scoring
        calls instead of one per protein.  ``proteins``: iterable of dicts with ``protein_id``, ``protein_sequence``
        and optionally ``positions``, ``pdb_path``, ``struct_features`` (same meaning as the arguments of
        ``predict_ptm_sites``).  Proteins are gathered until their ESM tables reach ``BATCH_TABLE_BYTES`` (or their
        sites ``BATCH_SITES``), scored in ONE native call and released, so the host never holds more than one such
        group of tables however long the input is (20,000 proteins' float tables are ~50 GB).  Returns one DataFrame
        with the reference's columns, proteins in input order."""
        import pandas as pd
        cols = ["protein_id", "position", "residue", "probability", "prediction"]
        frames, jobs, meta = [], [], []
        held_bytes = held_sites = 0

        def flush():
            nonlocal jobs, meta, held_bytes, held_sites
            if not jobs:
                return
            probas = self.model.predict_proteins(jobs)
            rows = [{"protein_id": pid, "position": pos, "residue": seq[pos - 1], "probability": pr,
                     "prediction": 1 if pr >= 0.5 else 0}
                    for (pid, seq, positions), proba in zip(meta, probas) for pos, pr in zip(positions, proba[:, 1])]
            frames.append(pd.DataFrame(rows, columns=cols))
            jobs, meta, held_bytes, held_sites = [], [], 0, 0

        for p in proteins:
            pid, seq = p["protein_id"], p["protein_sequence"]
            positions = p.get("positions")
            if positions is None:
                positions = [i + 1 for i, aa in enumerate(seq) if aa in self.config.target_aa]
            positions = list(positions)
            st = p.get("struct_features")
            if st is None:
                loader = self.data_loader
                if p.get("pdb_path"):          # a structure calculator is per protein
                    loader = PTMPredictDataLoader(self.config, None, self.data_loader.sasa_backend, self.data_loader.device)
                    loader.init_pyrosetta(p["pdb_path"])
                st = loader._generate_structural_features(positions)
            table = self.data_loader.load_custom_esm(pid)
            jobs.append((seq, positions, st, table))
            meta.append((pid, seq, positions))
            held_bytes += 0 if table is None else int(np.asarray(table).nbytes)
            held_sites += len(positions)
            if held_bytes >= self.BATCH_TABLE_BYTES or held_sites >= self.BATCH_SITES:
                flush()
        flush()
        return pd.concat(frames, ignore_index=True) if frames else pd.DataFrame([], columns=cols)


def run_protein(ptm_type: str, project_root: str, pdb_path: str, model_path: Optional[str] = None, out_dir: Optional[str] = None):
    """What ``python predict.py`` does for one PDB file (predict.py:615-659): returns the
    DataFrame and writes ``results/{id}_{ptm}_predictions.csv``."""
    cfg = PredictConfig(ptm_type, project_root, model_path)
    predictor = PTMPredictor(cfg)
    pid = extract_protein_id_from_pdb_path(pdb_path)
    seq = extract_sequence_from_pdb(pdb_path, chain_id="A")
    positions = [i + 1 for i, aa in enumerate(seq) if aa in cfg.target_aa]
    df = predictor.predict_ptm_sites(pid, seq, positions, pdb_path=pdb_path)
    out_dir = out_dir or os.path.join(cfg.root_dir, "results")
    os.makedirs(out_dir, exist_ok=True)
    path = os.path.join(out_dir, f"{pid}_{ptm_type}_predictions.csv")
    df.to_csv(path, index=False)
    return df, path


def run_all_ptm_types(project_root: str, pdb_path: str, model_paths: Optional[dict] = None, out_dir: Optional[str] = None,
                      ptm_types: Optional[Sequence[str]] = None, device: int = 0):
    """The 17-type sweep of the notebook menu (predict.ipynb cell 5; type -> residues from predict.py:119-137)
    for one protein: the sequence, the structure rows and the ESM table are prepared once, each PTM
    type scores its own target residues with its own weight file.  Types whose weight file is missing
    are skipped and reported.  Returns ``({ptm_type: DataFrame}, [skipped types])``."""
    pid = extract_protein_id_from_pdb_path(pdb_path)
    seq = extract_sequence_from_pdb(pdb_path, chain_id="A")
    calc = None
    results, skipped = {}, []
    out_dir = out_dir or os.path.join(project_root, "results")
    for ptm_type in (ptm_types or list(PTM_AA_MAP)):
        cfg = PredictConfig(ptm_type, project_root, (model_paths or {}).get(ptm_type))
        if not os.path.exists(cfg.model_path):
            skipped.append(ptm_type)
            continue
        predictor = PTMPredictor(cfg, struct_calculator=calc, device=device)
        positions = [i + 1 for i, aa in enumerate(seq) if aa in cfg.target_aa]
        df = predictor.predict_ptm_sites(pid, seq, positions, pdb_path=pdb_path)
        calc = predictor.data_loader.rosetta_calc          # structure features are per protein, not per type
        predictor.model.close()
        os.makedirs(out_dir, exist_ok=True)
        df.to_csv(os.path.join(out_dir, f"{pid}_{ptm_type}_predictions.csv"), index=False)
        results[ptm_type] = df
    return results, skipped
