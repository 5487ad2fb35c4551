"""ctypes binding of libdptm.so (C ABI in include/dptm.h).

There is no CPU fallback: if the shared library is missing or does not export the
expected symbols, importing a scorer fails loudly with ``NativeLibraryError``.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import numpy as np

from .spec import DEFAULT_SPEC, ModelSpec

_LIB_NAME = "libdptm.so"
# DPTM_LIBRARY overrides the in-tree path (used to A/B kernel builds); still the same C ABI
_LIB_PATH = os.environ.get("DPTM_LIBRARY") or os.path.join(os.path.dirname(os.path.abspath(__file__)), _LIB_NAME)

DPTM_OK, DPTM_EINVAL, DPTM_EHIP, DPTM_ENOMEM = 0, -1, -2, -3
PROF_NAMES = ("assemble", "branch21", "branch33", "branch51", "struct", "esm1", "esm2",
              "trunk1", "trunk2", "head")

PRECISIONS = {"f32": 0, "bf16": 1, "f16": 2, "f16x3": 3}

EXPORTS = (
    "dptm_weight_count", "dptm_create", "dptm_destroy", "dptm_score", "dptm_score_protein",
    "dptm_score_device", "dptm_score_protein_device", "dptm_assemble_device", "dptm_sync",
    "dptm_score_proteins", "dptm_score_proteins_device", "dptm_assemble_proteins_device",
    "dptm_score_device_noalloc", "dptm_noalloc_max_sites",
    "dptm_profile_begin", "dptm_profile_end", "dptm_chunk_sites", "dptm_last_error",
    "dptm_set_precision", "dptm_convert_rows", "dptm_score_rows16", "dptm_trainer_create", "dptm_trainer_destroy", "dptm_trainer_forward_backward",
    "dptm_trainer_forward_backward_rows",
    "dptm_trainer_grad_buffer", "dptm_trainer_param_floats", "dptm_trainer_apply",
    "dptm_trainer_get_weights", "dptm_trainer_get_grads", "dptm_trainer_get_proba",
    "dptm_trainer_get_optimizer_state", "dptm_trainer_set_optimizer_state",
    "dptm_trainer_profile_begin", "dptm_trainer_profile_end", "dptm_trainer_loss_buffer", "dptm_trainer_proba_buffer", "dptm_trainer_stream",
    "dptm_trainer_stage_loss", "dptm_trainer_lanes",
    "dptm_trainer_last_error", "dptm_trainer_set_allreduce", "dptm_sasa", "dptm_sasa_last_error",
)

# int (*dptm_allreduce_fn)(void* user, float* buf, int64_t count, void* stream)
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p)


class NativeLibraryError(RuntimeError):
    pass


class DptmSpec(C.Structure):
    _fields_ = [
        ("n_windows", C.c_int32), ("window_sizes", C.c_int32 * 3), ("max_features", C.c_int32),
        ("embed_dim", C.c_int32), ("num_heads", C.c_int32), ("ff_dim", C.c_int32),
        ("esm_dim", C.c_int32), ("struct_feature_dim", C.c_int32), ("struct_units", C.c_int32),
        ("esm_units1", C.c_int32), ("esm_units2", C.c_int32), ("trunk_units1", C.c_int32),
        ("trunk_units2", C.c_int32), ("n_classes", C.c_int32),
    ]

    @classmethod
    def from_spec(cls, spec: ModelSpec) -> "DptmSpec":
        s = cls()
        ws = list(spec.window_sizes)[:3] + [0] * (3 - min(3, len(spec.window_sizes)))
        s.n_windows = len(spec.window_sizes)
        s.window_sizes = (C.c_int32 * 3)(*ws)
        s.max_features = spec.max_features
        s.embed_dim = spec.embed_dim
        s.num_heads = spec.num_heads
        s.ff_dim = spec.ff_dim
        s.esm_dim = spec.esm_dim
        s.struct_feature_dim = spec.struct_feature_dim
        s.struct_units = spec.struct_units
        s.esm_units1 = spec.esm_units1
        s.esm_units2 = spec.esm_units2
        s.trunk_units1 = spec.trunk_units1
        s.trunk_units2 = spec.trunk_units2
        s.n_classes = spec.n_classes
        return s


class TrainOpts(C.Structure):
    _fields_ = [
        ("seed", C.c_uint32), ("rate_embed", C.c_float), ("rate_conv", C.c_float),
        ("rate_tb", C.c_float), ("rate_dense", C.c_float), ("n_global", C.c_int64),
        ("n_pos_global", C.c_int64), ("l2", C.c_float), ("l2_scale", C.c_float),
    ]


_lib: Optional[C.CDLL] = None


def library_path() -> str:
    return _LIB_PATH


def load_library() -> C.CDLL:
    """dlopen libdptm.so and declare its prototypes.  Raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        raise NativeLibraryError(
            f"{_LIB_PATH} is missing: build it with `make -C deepptmpred_amd/csrc` "
            "(or `python -c 'import __graft_entry__ as g; g.build()'`). There is no CPU fallback.")
    # PyTorch wheels bundle their own libamdhip64.so.7 / libhsa-runtime64.so.1.  A process can drive the
    # GPU through one ROCm runtime only: if libdptm.so pulled in the system copy first, a later
    # `import torch` would load the bundled one as a second runtime and find no GPU.  Loading torch
    # first makes libdptm.so bind to the runtime torch uses (same sonames), so device pointers of torch
    # tensors are valid in both.  DPTM_NO_TORCH_PRELOAD=1 skips this for torch-free processes.
    if not os.environ.get("DPTM_NO_TORCH_PRELOAD"):
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    try:
        lib = C.CDLL(_LIB_PATH)
    except OSError as e:  # pragma: no cover - depends on the machine
        raise NativeLibraryError(f"cannot load {_LIB_PATH}: {e}") from e
    missing = [n for n in EXPORTS if not hasattr(lib, n)]
    if missing:
        raise NativeLibraryError(f"{_LIB_PATH} lacks symbols: {missing}")
    vp, i64, u8p, f32p, i32p = C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p
    lib.dptm_weight_count.restype = C.c_int64
    lib.dptm_weight_count.argtypes = [C.POINTER(DptmSpec)]
    lib.dptm_create.restype = C.c_int
    lib.dptm_create.argtypes = [vp, C.c_size_t, C.POINTER(DptmSpec), C.c_int, C.POINTER(vp)]
    lib.dptm_destroy.restype = None
    lib.dptm_destroy.argtypes = [vp]
    lib.dptm_score.restype = C.c_int
    lib.dptm_score.argtypes = [vp, u8p, u8p, u8p, f32p, f32p, i64, f32p]
    lib.dptm_score_rows16.restype = C.c_int
    lib.dptm_score_rows16.argtypes = [vp, u8p, u8p, u8p, f32p, vp, i64, f32p]
    lib.dptm_score_protein.restype = C.c_int
    lib.dptm_score_protein.argtypes = [vp, u8p, i64, f32p, i64, i32p, f32p, i64, f32p]
    lib.dptm_score_device.restype = C.c_int
    lib.dptm_score_device.argtypes = [vp, u8p, u8p, u8p, f32p, f32p, i64, f32p, vp]
    lib.dptm_score_device_noalloc.restype = C.c_int
    lib.dptm_score_device_noalloc.argtypes = [vp, u8p, u8p, u8p, f32p, f32p, i64, f32p, vp]
    lib.dptm_noalloc_max_sites.restype = C.c_int64
    lib.dptm_noalloc_max_sites.argtypes = []
    lib.dptm_score_protein_device.restype = C.c_int
    lib.dptm_score_protein_device.argtypes = [vp, u8p, i64, f32p, i64, i32p, f32p, i64, f32p, vp]
    lib.dptm_assemble_device.restype = C.c_int
    lib.dptm_assemble_device.argtypes = [vp, u8p, i64, i64, i32p, i64, u8p, u8p, u8p, i32p, vp]
    lib.dptm_score_proteins.restype = C.c_int
    lib.dptm_score_proteins.argtypes = [vp, u8p, vp, f32p, vp, i64, i32p, i32p, f32p, i64, f32p]
    lib.dptm_score_proteins_device.restype = C.c_int
    lib.dptm_score_proteins_device.argtypes = [vp, u8p, vp, f32p, vp, i64, i32p, i32p, f32p, i64, f32p, vp]
    lib.dptm_assemble_proteins_device.restype = C.c_int
    lib.dptm_assemble_proteins_device.argtypes = [vp, u8p, vp, vp, i32p, i32p, i64, u8p, u8p, u8p, i32p, vp]
    lib.dptm_sync.restype = C.c_int
    lib.dptm_sync.argtypes = [vp]
    lib.dptm_profile_begin.restype = C.c_int
    lib.dptm_profile_begin.argtypes = [vp]
    lib.dptm_profile_end.restype = C.c_int
    lib.dptm_profile_end.argtypes = [vp, C.c_void_p, C.c_void_p]
    lib.dptm_set_precision.restype = C.c_int
    lib.dptm_set_precision.argtypes = [vp, C.c_int]
    lib.dptm_convert_rows.restype = C.c_int
    lib.dptm_convert_rows.argtypes = [f32p, i64, C.c_int, vp]
    lib.dptm_chunk_sites.restype = C.c_int64
    lib.dptm_chunk_sites.argtypes = [vp]
    lib.dptm_last_error.restype = C.c_char_p
    lib.dptm_last_error.argtypes = []
    lib.dptm_trainer_create.restype = C.c_int
    lib.dptm_trainer_create.argtypes = [vp, C.c_size_t, C.POINTER(DptmSpec), C.c_int, i64, C.POINTER(vp)]
    lib.dptm_trainer_destroy.restype = None
    lib.dptm_trainer_destroy.argtypes = [vp]
    lib.dptm_trainer_forward_backward.restype = C.c_int
    lib.dptm_trainer_forward_backward.argtypes = [vp, u8p, u8p, u8p, f32p, f32p, i32p, i64,
                                                  C.POINTER(TrainOpts), C.POINTER(C.c_float), vp]
    lib.dptm_trainer_forward_backward_rows.restype = C.c_int
    lib.dptm_trainer_forward_backward_rows.argtypes = [vp, vp, vp, vp, vp, vp, vp, i64, vp, i64, C.POINTER(TrainOpts), vp, vp, vp]
    lib.dptm_trainer_grad_buffer.restype = C.c_int
    lib.dptm_trainer_grad_buffer.argtypes = [vp, C.POINTER(vp), C.POINTER(i64)]
    lib.dptm_trainer_param_floats.restype = C.c_int64
    lib.dptm_trainer_param_floats.argtypes = [vp]
    lib.dptm_trainer_apply.restype = C.c_int
    lib.dptm_trainer_apply.argtypes = [vp, C.c_float, C.c_float, C.c_float, C.c_float, vp]
    for name in ("dptm_trainer_get_weights", "dptm_trainer_get_grads"):
        getattr(lib, name).restype = C.c_int
        getattr(lib, name).argtypes = [vp, f32p]
    lib.dptm_trainer_get_proba.restype = C.c_int
    lib.dptm_trainer_get_proba.argtypes = [vp, f32p, i64]
    lib.dptm_trainer_get_optimizer_state.restype = C.c_int
    lib.dptm_trainer_get_optimizer_state.argtypes = [vp, f32p, f32p, C.POINTER(i64)]
    lib.dptm_trainer_set_optimizer_state.restype = C.c_int
    lib.dptm_trainer_set_optimizer_state.argtypes = [vp, f32p, f32p, i64]
    lib.dptm_trainer_loss_buffer.restype = C.c_int
    lib.dptm_trainer_loss_buffer.argtypes = [vp, C.POINTER(vp)]
    lib.dptm_trainer_proba_buffer.restype = C.c_int
    lib.dptm_trainer_proba_buffer.argtypes = [vp, C.POINTER(vp)]
    lib.dptm_trainer_stage_loss.restype = C.c_int
    lib.dptm_trainer_stage_loss.argtypes = [vp, C.POINTER(i64), vp]
    lib.dptm_trainer_lanes.restype = C.c_int
    lib.dptm_trainer_lanes.argtypes = [vp, C.POINTER(C.c_int32)]
    lib.dptm_trainer_stream.restype = C.c_int
    lib.dptm_trainer_stream.argtypes = [vp, C.POINTER(vp)]
    lib.dptm_trainer_profile_begin.restype = C.c_int
    lib.dptm_trainer_profile_begin.argtypes = [vp]
    lib.dptm_trainer_profile_end.restype = C.c_int
    lib.dptm_trainer_profile_end.argtypes = [vp, f32p]
    lib.dptm_trainer_set_allreduce.restype = C.c_int
    lib.dptm_trainer_set_allreduce.argtypes = [vp, ALLREDUCE_FN, vp]
    lib.dptm_trainer_last_error.restype = C.c_char_p
    lib.dptm_trainer_last_error.argtypes = []
    lib.dptm_sasa.restype = C.c_int
    lib.dptm_sasa.argtypes = [vp, vp, i64, C.c_double, vp, C.c_int32, C.c_int32, vp]
    lib.dptm_sasa_last_error.restype = C.c_char_p
    lib.dptm_sasa_last_error.argtypes = []
    _lib = lib
    return lib


def convert_rows(a: np.ndarray, mode: str) -> np.ndarray:
    """fp32 array -> uint16 array of the same shape holding the bfloat16 / float16 bit patterns the library's host
    pipelines send over PCIe in the "bf16" / "f16" modes (round to nearest even)."""
    lib = load_library()
    a = np.ascontiguousarray(a, dtype=np.float32)
    out = np.empty(a.shape, dtype=np.uint16)
    _check(lib, lib.dptm_convert_rows(_ptr(a), a.size, PRECISIONS[mode], _ptr(out)), "dptm_convert_rows")
    return out


def _check(lib, rc: int, what: str) -> None:
    if rc == DPTM_OK:
        return
    msg = (lib.dptm_last_error() or b"").decode("utf-8", "replace")
    if rc == DPTM_EINVAL:
        raise ValueError(f"{what}: {msg}")
    if rc == DPTM_ENOMEM:
        raise MemoryError(f"{what}: {msg}")
    raise RuntimeError(f"{what}: {msg}")


def _ptr(a: np.ndarray) -> C.c_void_p:
    return C.c_void_p(a.ctypes.data)


def _buffer(a, dtype, shape, what: str) -> np.ndarray:
    """The C side reads raw memory: refuse anything but a C-contiguous array of exactly this dtype and
    shape (``None`` = any length in that axis) instead of letting it read past the end."""
    if not isinstance(a, np.ndarray) or a.dtype != np.dtype(dtype) or not a.flags.c_contiguous:
        raise ValueError(f"{what}: expected a C-contiguous numpy array of dtype {np.dtype(dtype).name}")
    if a.ndim != len(shape) or any(w is not None and a.shape[i] != w for i, w in enumerate(shape)):
        raise ValueError(f"{what}: expected shape {tuple('n' if w is None else w for w in shape)}, got {a.shape}")
    return a


def sasa(xyz: np.ndarray, radii: np.ndarray, probe: float, unit_points: np.ndarray, device: int = 0) -> np.ndarray:
    """Per-atom solvent-accessible area on the GPU (``dptm_sasa``); float64 in, float64 out."""
    lib = load_library()
    xyz = np.ascontiguousarray(xyz, dtype=np.float64).reshape(-1, 3)
    radii = np.ascontiguousarray(radii, dtype=np.float64)
    unit_points = np.ascontiguousarray(unit_points, dtype=np.float64).reshape(-1, 3)
    if len(radii) != len(xyz):
        raise ValueError("sasa: xyz and radii differ in length")
    out = np.zeros(len(xyz), dtype=np.float64)
    rc = lib.dptm_sasa(_ptr(xyz), _ptr(radii), len(xyz), float(probe), _ptr(unit_points), len(unit_points), device, _ptr(out))
    if rc != DPTM_OK:
        msg = (lib.dptm_sasa_last_error() or b"").decode("utf-8", "replace")
        raise (ValueError if rc == DPTM_EINVAL else RuntimeError)(msg)
    return out


class Scorer:
    """One ``dptm_handle``: a weight set resident on one GPU."""

    def __init__(self, blob: np.ndarray, spec: ModelSpec = DEFAULT_SPEC, device: int = 0):
        self._lib = load_library()
        self._h = C.c_void_p()
        self.spec = spec
        self.device = device
        blob = np.ascontiguousarray(blob, dtype=np.float32)
        cs = DptmSpec.from_spec(spec)
        rc = self._lib.dptm_create(_ptr(blob), blob.nbytes, C.byref(cs), device, C.byref(self._h))
        _check(self._lib, rc, "dptm_create")

    def close(self) -> None:
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.dptm_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    @property
    def chunk_sites(self) -> int:
        return int(self._lib.dptm_chunk_sites(self._h))

    # ---- host-buffer calls (numpy in, numpy out)
    def score(self, tok21, tok33, tok51, strct, esm) -> np.ndarray:
        n = len(tok21)
        ws = self.spec.window_sizes
        for a, w, name in ((tok21, ws[0], "tok21"), (tok33, ws[1], "tok33"), (tok51, ws[2], "tok51")):
            _buffer(a, np.uint8, (n, w), f"dptm_score: {name}")
        _buffer(strct, np.float32, (n, self.spec.struct_feature_dim), "dptm_score: struct")
        _buffer(esm, np.float32, (n, self.spec.esm_dim), "dptm_score: esm")
        out = np.empty((n, 2), dtype=np.float32)
        rc = self._lib.dptm_score(self._h, _ptr(tok21), _ptr(tok33), _ptr(tok51), _ptr(strct),
                                  _ptr(esm), n, _ptr(out))
        _check(self._lib, rc, "dptm_score")
        return out

    def score_rows16(self, tok21, tok33, tok51, strct, esm16) -> np.ndarray:
        """``score`` with the ESM rows kept as the 16-bit operand values of the scorer's bf16 / f16 mode
        (``convert_rows``): uint16 ``[n][1280]``."""
        n = len(tok21)
        ws = self.spec.window_sizes
        for a, w, name in ((tok21, ws[0], "tok21"), (tok33, ws[1], "tok33"), (tok51, ws[2], "tok51")):
            _buffer(a, np.uint8, (n, w), f"dptm_score_rows16: {name}")
        _buffer(strct, np.float32, (n, self.spec.struct_feature_dim), "dptm_score_rows16: struct")
        _buffer(esm16, np.uint16, (n, self.spec.esm_dim), "dptm_score_rows16: esm")
        out = np.empty((n, 2), dtype=np.float32)
        rc = self._lib.dptm_score_rows16(self._h, _ptr(tok21), _ptr(tok33), _ptr(tok51), _ptr(strct),
                                         C.c_void_p(esm16.ctypes.data), n, _ptr(out))
        _check(self._lib, rc, "dptm_score_rows16")
        return out

    def score_protein(self, codes, esm_table, positions, strct) -> np.ndarray:
        n = len(positions)
        _buffer(codes, np.uint8, (None,), "dptm_score_protein: codes")
        _buffer(positions, np.int32, (n,), "dptm_score_protein: positions")
        _buffer(strct, np.float32, (n, self.spec.struct_feature_dim), "dptm_score_protein: struct")
        if esm_table is not None and len(esm_table):
            _buffer(esm_table, np.float32, (None, self.spec.esm_dim), "dptm_score_protein: esm table")
        out = np.empty((n, 2), dtype=np.float32)
        rows = 0 if esm_table is None else len(esm_table)
        tp = C.c_void_p(None) if esm_table is None or rows == 0 else _ptr(esm_table)
        rc = self._lib.dptm_score_protein(self._h, _ptr(codes), len(codes), tp, rows,
                                          _ptr(positions), _ptr(strct), n, _ptr(out))
        _check(self._lib, rc, "dptm_score_protein")
        return out

    def score_proteins(self, codes, seq_offsets, esm_tables, esm_offsets, positions, site_protein, strct) -> np.ndarray:
        """Sites of many proteins in one call (``dptm_score_proteins``); see include/dptm.h for the layout."""
        n = len(positions)
        n_prot = len(seq_offsets) - 1
        _buffer(codes, np.uint8, (None,), "dptm_score_proteins: codes")
        _buffer(seq_offsets, np.int64, (n_prot + 1,), "dptm_score_proteins: seq_offsets")
        _buffer(esm_offsets, np.int64, (n_prot + 1,), "dptm_score_proteins: esm_offsets")
        _buffer(positions, np.int32, (n,), "dptm_score_proteins: positions")
        _buffer(site_protein, np.int32, (n,), "dptm_score_proteins: site_protein")
        _buffer(strct, np.float32, (n, self.spec.struct_feature_dim), "dptm_score_proteins: struct")
        if n_prot < 0 or (n_prot >= 0 and len(codes) != int(seq_offsets[-1])):
            raise ValueError("dptm_score_proteins: seq_offsets[-1] must equal len(codes)")
        rows = int(esm_offsets[-1])
        tp = C.c_void_p(None)
        if rows:
            _buffer(esm_tables, np.float32, (rows, self.spec.esm_dim), "dptm_score_proteins: esm tables")
            tp = _ptr(esm_tables)
        out = np.empty((n, 2), dtype=np.float32)
        rc = self._lib.dptm_score_proteins(self._h, _ptr(codes), _ptr(seq_offsets), tp, _ptr(esm_offsets), n_prot,
                                           _ptr(positions), _ptr(site_protein), _ptr(strct), n, _ptr(out))
        _check(self._lib, rc, "dptm_score_proteins")
        return out

    # ---- device-pointer calls (integers = device addresses, e.g. torch data_ptr())
    def score_proteins_device(self, codes, seq_offsets, esm_tables, esm_offsets, n_proteins, positions, site_protein,
                              strct, n, out, stream=0) -> None:
        rc = self._lib.dptm_score_proteins_device(self._h, codes, seq_offsets, esm_tables or None, esm_offsets, n_proteins,
                                                  positions, site_protein, strct, n, out, C.c_void_p(stream or None))
        _check(self._lib, rc, "dptm_score_proteins_device")

    def assemble_proteins_device(self, codes, seq_offsets, esm_offsets, positions, site_protein, n, tok21, tok33, tok51,
                                 esm_row, stream=0) -> None:
        rc = self._lib.dptm_assemble_proteins_device(self._h, codes, seq_offsets, esm_offsets, positions, site_protein, n,
                                                     tok21, tok33, tok51, esm_row, C.c_void_p(stream or None))
        _check(self._lib, rc, "dptm_assemble_proteins_device")

    def score_device(self, tok21, tok33, tok51, strct, esm, n, out, stream=0) -> None:
        rc = self._lib.dptm_score_device(self._h, tok21, tok33, tok51, strct, esm, n, out,
                                         C.c_void_p(stream or None))
        _check(self._lib, rc, "dptm_score_device")

    def score_device_noalloc(self, tok21, tok33, tok51, strct, esm, n, out, stream=0) -> None:
        """``score_device`` on the library's static workspace: allocates nothing, at most
        ``dptm_noalloc_max_sites()`` sites per call."""
        rc = self._lib.dptm_score_device_noalloc(self._h, tok21, tok33, tok51, strct, esm, n, out,
                                                 C.c_void_p(stream or None))
        _check(self._lib, rc, "dptm_score_device_noalloc")

    def score_protein_device(self, codes, L, esm_table, esm_rows, positions, strct, n, out,
                             stream=0) -> None:
        rc = self._lib.dptm_score_protein_device(self._h, codes, L, esm_table, esm_rows, positions,
                                                 strct, n, out, C.c_void_p(stream or None))
        _check(self._lib, rc, "dptm_score_protein_device")

    def assemble_device(self, codes, L, esm_rows, positions, n, tok21, tok33, tok51, esm_row,
                        stream=0) -> None:
        rc = self._lib.dptm_assemble_device(self._h, codes, L, esm_rows, positions, n, tok21, tok33,
                                            tok51, esm_row, C.c_void_p(stream or None))
        _check(self._lib, rc, "dptm_assemble_device")

    def sync(self) -> None:
        _check(self._lib, self._lib.dptm_sync(self._h), "dptm_sync")

    def set_precision(self, mode) -> None:
        """mode: 'f32' | 'bf16' | 'f16' | 'f16x3' (or the DPTM_PREC_* integer)."""
        if isinstance(mode, str):
            if mode.lower() not in PRECISIONS:
                raise ValueError("unknown precision %r (one of %s)" % (mode, ", ".join(PRECISIONS)))
            mode = PRECISIONS[mode.lower()]
        _check(self._lib, self._lib.dptm_set_precision(self._h, int(mode)), "dptm_set_precision")

    def profile_begin(self) -> None:
        _check(self._lib, self._lib.dptm_profile_begin(self._h), "dptm_profile_begin")

    def profile_end(self):
        ms = np.zeros(len(PROF_NAMES), dtype=np.float32)
        cnt = np.zeros(len(PROF_NAMES), dtype=np.int64)
        _check(self._lib, self._lib.dptm_profile_end(self._h, _ptr(ms), _ptr(cnt)), "dptm_profile_end")
        return {k: (float(m), int(c)) for k, m, c in zip(PROF_NAMES, ms, cnt)}


def _check_t(lib, rc: int, what: str) -> None:
    if rc == DPTM_OK:
        return
    msg = (lib.dptm_trainer_last_error() or lib.dptm_last_error() or b"").decode("utf-8", "replace")
    if rc == DPTM_EINVAL:
        raise ValueError(f"{what}: {msg}")
    if rc == DPTM_ENOMEM:
        raise MemoryError(f"{what}: {msg}")
    raise RuntimeError(f"{what}: {msg}")


class Trainer:
    """One ``dptm_trainer``: a trainable replica on one GPU (device-pointer interface)."""

    def __init__(self, blob: np.ndarray, spec: ModelSpec = DEFAULT_SPEC, device: int = 0,
                 max_batch: int = 64):
        self._lib = load_library()
        self._h = C.c_void_p()
        self.spec = spec
        self.max_batch = int(max_batch)
        self.n_canonical = int(np.asarray(blob).size)
        blob = np.ascontiguousarray(blob, dtype=np.float32)
        cs = DptmSpec.from_spec(spec)
        rc = self._lib.dptm_trainer_create(_ptr(blob), blob.nbytes, C.byref(cs), device, max_batch,
                                           C.byref(self._h))
        _check_t(self._lib, rc, "dptm_trainer_create")

    def close(self) -> None:
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.dptm_trainer_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    def set_allreduce(self, fn) -> None:
        """``fn(dev_ptr: int, count: int, stream: int)`` must sum ``count`` floats at ``dev_ptr`` over
        all replicas, ordered on the HIP stream ``stream`` (synchronised BatchNorm); ``None`` turns it
        off.  An exception raised inside ``fn`` aborts the step and is re-raised by forward_backward."""
        self._ar_error = None
        if fn is None:
            self._ar_cb = ALLREDUCE_FN()            # NULL function pointer
        else:
            def trampoline(_user, buf, count, stream):
                try:
                    fn(int(buf or 0), int(count), int(stream or 0))
                    return 0
                except BaseException as e:           # must not propagate into the C frame
                    self._ar_error = e
                    return -1
            self._ar_cb = ALLREDUCE_FN(trampoline)   # kept alive as long as the trainer uses it
        _check_t(self._lib, self._lib.dptm_trainer_set_allreduce(self._h, self._ar_cb, None),
                 "dptm_trainer_set_allreduce")

    def forward_backward(self, tok21, tok33, tok51, strct, esm, labels, n, opts: TrainOpts,
                         want_loss: bool = True, stream=0):
        loss = C.c_float(0.0)
        rc = self._lib.dptm_trainer_forward_backward(
            self._h, tok21, tok33, tok51, strct, esm, labels, n, C.byref(opts),
            C.byref(loss) if want_loss else None, C.c_void_p(stream or None))
        err, self._ar_error = getattr(self, "_ar_error", None), None
        if err is not None:
            raise err
        _check_t(self._lib, rc, "dptm_trainer_forward_backward")
        return float(loss.value) if want_loss else None

    def forward_backward_rows(self, tok21, tok33, tok51, strct, esm, labels, n_data: int, rows, n: int, opts: TrainOpts,
                              loss_dev_out=0, proba_dev_out=0, stream=0) -> None:
        """The step on rows ``rows[0..n)`` (device int64) of a device-resident data set; loss and class probabilities
        are left at the given device addresses (0: not wanted).  All arguments but ``n_data`` / ``n`` are device addresses."""
        p = lambda a: C.c_void_p(a or None)
        rc = self._lib.dptm_trainer_forward_backward_rows(
            self._h, p(tok21), p(tok33), p(tok51), p(strct), p(esm), p(labels), n_data, p(rows), n, C.byref(opts),
            p(loss_dev_out), p(proba_dev_out), p(stream))
        err, self._ar_error = getattr(self, "_ar_error", None), None
        if err is not None:
            raise err
        _check_t(self._lib, rc, "dptm_trainer_forward_backward_rows")

    def loss_buffer(self) -> int:
        """Device address of the step's loss (one float32, written by forward_backward in stream order)."""
        p = C.c_void_p()
        _check_t(self._lib, self._lib.dptm_trainer_loss_buffer(self._h, C.byref(p)), "dptm_trainer_loss_buffer")
        return int(p.value)

    def proba_buffer(self) -> int:
        """Device address of the step's class probabilities ([n][2] float32, the training-mode outputs that
        forward_backward writes in stream order)."""
        p = C.c_void_p()
        _check_t(self._lib, self._lib.dptm_trainer_proba_buffer(self._h, C.byref(p)), "dptm_trainer_proba_buffer")
        return int(p.value)

    def stage_loss(self, stream=0) -> int:
        """Copies the step's loss into the spare float of the gradient buffer (in stream order) and returns that
        float's index: the one gradient all-reduce of a data-parallel step then sums the loss shares as well."""
        slot = C.c_int64(-1)
        _check_t(self._lib, self._lib.dptm_trainer_stage_loss(self._h, C.byref(slot), C.c_void_p(stream or None)),
                 "dptm_trainer_stage_loss")
        return int(slot.value)

    def lanes_independent(self) -> int:
        """How many of the step's four streams sit on hardware queues of their own (4 = all concurrent)."""
        v = C.c_int32(0)
        _check_t(self._lib, self._lib.dptm_trainer_lanes(self._h, C.byref(v)), "dptm_trainer_lanes")
        return int(v.value)

    def stream(self) -> int:
        """Handle (hipStream_t) of the stream the step's main lane runs on; work enqueued on it and calls that pass
        it as their stream need no hand-over between streams.  Owned by the trainer."""
        p = C.c_void_p()
        _check_t(self._lib, self._lib.dptm_trainer_stream(self._h, C.byref(p)), "dptm_trainer_stream")
        return int(p.value)

    def grad_buffer(self):
        p = C.c_void_p()
        n = C.c_int64()
        _check_t(self._lib, self._lib.dptm_trainer_grad_buffer(self._h, C.byref(p), C.byref(n)),
                 "dptm_trainer_grad_buffer")
        return int(p.value), int(n.value)

    def apply(self, lr: float, weight_decay: float, grad_scale: float = 1.0, stat_scale: float = 1.0,
              stream=0) -> None:
        rc = self._lib.dptm_trainer_apply(self._h, lr, weight_decay, grad_scale, stat_scale,
                                          C.c_void_p(stream or None))
        _check_t(self._lib, rc, "dptm_trainer_apply")

    def get_weights(self) -> np.ndarray:
        out = np.empty(self.n_canonical, dtype=np.float32)
        _check_t(self._lib, self._lib.dptm_trainer_get_weights(self._h, _ptr(out)), "dptm_trainer_get_weights")
        return out

    def get_grads(self) -> np.ndarray:
        out = np.empty(self.n_canonical, dtype=np.float32)
        _check_t(self._lib, self._lib.dptm_trainer_get_grads(self._h, _ptr(out)), "dptm_trainer_get_grads")
        return out

    def get_optimizer_state(self):
        """(m, v, step): Adam moments in the canonical blob order and the number of steps taken."""
        m = np.empty(self.n_canonical, dtype=np.float32)
        v = np.empty(self.n_canonical, dtype=np.float32)
        step = C.c_int64(0)
        _check_t(self._lib, self._lib.dptm_trainer_get_optimizer_state(self._h, _ptr(m), _ptr(v), C.byref(step)),
                 "dptm_trainer_get_optimizer_state")
        return m, v, int(step.value)

    def set_optimizer_state(self, m: np.ndarray, v: np.ndarray, step: int) -> None:
        m = np.ascontiguousarray(m, dtype=np.float32)
        v = np.ascontiguousarray(v, dtype=np.float32)
        if m.size != self.n_canonical or v.size != self.n_canonical:
            raise ValueError("optimizer moments must hold one float per weight of the canonical blob")
        _check_t(self._lib, self._lib.dptm_trainer_set_optimizer_state(self._h, _ptr(m), _ptr(v), int(step)),
                 "dptm_trainer_set_optimizer_state")

    PROFILE_KERNELS = ("conv_a_fwd", "conv_b_fwd", "tb_fwd", "tb_bwd", "conv_b_bwd", "conv_a_bwd")

    def profile_begin(self) -> None:
        """Arms the per-kernel clocks: until profile_end() the sequence branches run one after another on one stream."""
        _check_t(self._lib, self._lib.dptm_trainer_profile_begin(self._h), "dptm_trainer_profile_begin")

    def profile_end(self) -> dict:
        """{"tb_bwd51": ms, ...} for the sequence-branch launches of the last step."""
        ms = np.zeros(18, dtype=np.float32)
        _check_t(self._lib, self._lib.dptm_trainer_profile_end(self._h, _ptr(ms)), "dptm_trainer_profile_end")
        return {f"{k}{w}": float(ms[b * 6 + i]) for b, w in enumerate((21, 33, 51)) for i, k in enumerate(self.PROFILE_KERNELS)}

    def get_proba(self, n: int) -> np.ndarray:
        out = np.empty((n, 2), dtype=np.float32)
        _check_t(self._lib, self._lib.dptm_trainer_get_proba(self._h, _ptr(out), n), "dptm_trainer_get_proba")
        return out
