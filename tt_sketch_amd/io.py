"""FROSTT ``.tns`` ingestion: text, one nonzero per line, 1-based indices
followed by the value, optionally gzip-compressed -- the format the reference's
``scripts/frostt.py:51-65`` parses line by line in Python.  Here the file is parsed in one
vectorised pass and handed to ``SparseTensor`` (indices (d, nnz) int64, entries (nnz,))."""
from __future__ import annotations

import gzip
import io
from typing import Optional, Tuple

import numpy as np

from .tensor import SparseTensor


def read_tns(path: str, shape: Optional[Tuple[int, ...]] = None) -> SparseTensor:
    """Load a FROSTT tensor.  ``shape`` defaults to the largest index per mode; lines starting
    with ``#`` and blank lines are skipped.  Raises ``ValueError`` on ragged lines, indices < 1 or
    indices outside ``shape``."""
    opener = gzip.open if str(path).endswith(".gz") else open
    with opener(path, "rb") as f:
        raw = f.read()
    lines = [ln for ln in raw.splitlines() if ln.strip() and not ln.lstrip().startswith(b"#")]
    if not lines:
        if shape is None:
            raise ValueError("empty .tns file and no shape given")
        d = len(shape)
        return SparseTensor(shape, np.zeros((d, 0), dtype=np.int64), np.zeros(0))
    width = len(lines[0].split())
    if width < 2:
        raise ValueError("a .tns line needs at least one index and a value")
    try:
        table = np.loadtxt(io.BytesIO(b"\n".join(lines)), dtype=np.float64, ndmin=2)
    except ValueError as exc:
        raise ValueError(f"ragged or malformed .tns file: {exc}") from None
    if table.shape[1] != width:
        raise ValueError("ragged .tns file")
    idx = table[:, :-1]
    if np.any(idx != np.floor(idx)) or np.any(idx < 1):
        raise ValueError(".tns indices must be integers >= 1")
    indices = idx.astype(np.int64).T - 1
    entries = np.ascontiguousarray(table[:, -1])
    if shape is None:
        shape = tuple(int(m) + 1 for m in indices.max(axis=1))
    else:
        shape = tuple(int(n) for n in shape)
        if len(shape) != indices.shape[0]:
            raise ValueError(f"shape has {len(shape)} modes, file has {indices.shape[0]}")
        if np.any(indices.max(axis=1) >= np.array(shape)):
            raise ValueError("index outside the given shape")
    return SparseTensor(shape, np.ascontiguousarray(indices), entries)
