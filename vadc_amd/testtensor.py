"""Reader/writer for the `.testtensor` container used by vadc for weights and test fixtures.

Format (little-endian, packed) as consumed by the reference loader
(/root/reference/tensor.h:97-102, 201-253) and produced by its writer
(/root/reference/utils.py:7-53):

    int32 version (=1), int32 count
    count x { int32 name_len, name_len bytes UTF-8 (no NUL) }
    count x { int32 ndim, int32 dims[ndim], int32 size, int32 nbytes, nbytes of float32 }

Tensors are addressed BY POSITION by every consumer; names are informational.
"""
from __future__ import annotations

import struct
from typing import List, Sequence, Tuple

import numpy as np

VERSION = 1


def loads(raw: bytes) -> List[Tuple[str, np.ndarray]]:
    off = 0
    version, count = struct.unpack_from("<ii", raw, off)
    off += 8
    if version != VERSION:
        raise ValueError(f"unsupported .testtensor version {version}")
    if count <= 0:
        raise ValueError("empty .testtensor")
    names = []
    for _ in range(count):
        (n,) = struct.unpack_from("<i", raw, off)
        off += 4
        names.append(raw[off:off + n].decode("utf-8"))
        off += n
    out = []
    for name in names:
        (ndim,) = struct.unpack_from("<i", raw, off)
        off += 4
        dims = struct.unpack_from(f"<{ndim}i", raw, off)
        off += 4 * ndim
        size, nbytes = struct.unpack_from("<ii", raw, off)
        off += 8
        if nbytes != size * 4 or int(np.prod(dims, dtype=np.int64)) != size:
            raise ValueError(f"tensor {name!r}: inconsistent header dims={dims} size={size} nbytes={nbytes}")
        arr = np.frombuffer(raw, dtype="<f4", count=size, offset=off).reshape(dims).copy()
        off += nbytes
        out.append((name, arr))
    if off != len(raw):
        raise ValueError(f"trailing bytes in .testtensor: parsed {off} of {len(raw)}")
    return out


def load(path: str) -> List[Tuple[str, np.ndarray]]:
    with open(path, "rb") as f:
        return loads(f.read())


def dumps(tensors: Sequence[Tuple[str, np.ndarray]]) -> bytes:
    parts = [struct.pack("<ii", VERSION, len(tensors))]
    for name, _ in tensors:
        b = name.encode("utf-8")
        parts.append(struct.pack("<i", len(b)))
        parts.append(b)
    for _, arr in tensors:
        a = np.ascontiguousarray(arr, dtype="<f4")
        parts.append(struct.pack("<i", a.ndim))
        parts.append(struct.pack(f"<{a.ndim}i", *a.shape))
        parts.append(struct.pack("<ii", a.size, a.nbytes))
        parts.append(a.tobytes())
    return b"".join(parts)


def dump(path: str, tensors: Sequence[Tuple[str, np.ndarray]]) -> None:
    with open(path, "wb") as f:
        f.write(dumps(tensors))
