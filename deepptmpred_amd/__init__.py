"""MI355X-native scoring path for DeepPTMPred's PTM-site classifier.

Public surface (same names as the reference's call sites use):
``load_model``, ``create_ptm_model``, ``Model.predict``.
"""
from .spec import DEFAULT_SPEC, ModelSpec, PTM_AA_MAP  # noqa: F401
from .model import Model, create_ptm_model, encode_sequence, load_model  # noqa: F401
