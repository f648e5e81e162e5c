"""Flat binary "compiled model" blob shared by the CPU oracle and the HIP engine.

Layout (little endian), mirrored by ``include/flybody_engine.h``:
    char     magic[4] = "FBM1"
    uint32   narr
    narr x { char name[40]; uint32 dtype (0 = f64, 1 = i32); uint32 ndim; uint32 shape[4];
             uint64 offset (from blob start, 8-byte aligned); uint64 nbytes }
    data region
String-typed entries of the compiled model (names, notes) are not part of the blob.
"""
from __future__ import annotations

import struct
from typing import Dict

import numpy as np

_ENTRY = struct.Struct('<40sII4IQQ')
MAGIC = b'FBM1'


def pack_model(m: Dict[str, np.ndarray]) -> bytes:
    items = []
    for k, v in m.items():
        v = np.asarray(v)
        if v.dtype.kind in 'US' or v.dtype == object:
            continue
        if v.dtype.kind in 'iub':
            arr = np.ascontiguousarray(v, dtype='<i4'); dt = 1
        else:
            arr = np.ascontiguousarray(v, dtype='<f8'); dt = 0
        if arr.ndim > 4:
            raise ValueError(k)
        if len(k) > 39:
            raise ValueError(f'name too long: {k}')
        items.append((k, dt, arr))
    head = 8 + _ENTRY.size * len(items)
    off = (head + 7) // 8 * 8
    table = bytearray()
    data = bytearray()
    for k, dt, arr in items:
        shape = list(arr.shape) + [0] * (4 - arr.ndim)
        nb = arr.nbytes
        table += _ENTRY.pack(k.encode(), dt, arr.ndim, *shape, off + len(data), nb)
        data += arr.tobytes()
        data += b'\0' * ((-len(data)) % 8)
    blob = MAGIC + struct.pack('<I', len(items)) + bytes(table)
    blob += b'\0' * (off - len(blob)) + bytes(data)
    return blob


def load_npz(path: str) -> Dict[str, np.ndarray]:
    with np.load(path, allow_pickle=False) as z:
        return {k: z[k] for k in z.files}
