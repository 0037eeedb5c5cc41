"""File formats of the posterior surfaces.

``saveBin``/``loadBin``: the ``.lab`` format PhoneRecog reads -- ``uint32 rows, uint32 cols, float32[rows*cols]`` row-major
(reference scripts/util/kw_utils.py:4-22; reader ``CData::LoadBin``, recog_src/source/data.cpp:23-44).
"""
from __future__ import annotations

import numpy as np


def saveBin(filename, x):
    x = np.asarray(x)
    dims = np.zeros(2, dtype=np.uint32)
    if x.ndim == 1:
        dims[:] = (x.shape[0], 1)  # 1-D input is written as [n, 1]
    elif x.ndim == 2:
        dims[:] = x.shape
    else:
        raise ValueError("saveBin takes 1-D or 2-D arrays")
    with open(filename, "wb") as fid:
        fid.write(dims.tobytes())
        fid.write(np.ascontiguousarray(x).tobytes())


def loadBin(filename, dtype=np.float32):
    with open(filename, "rb") as fid:
        dims = np.frombuffer(fid.read(8), dtype=np.uint32)
        x = np.frombuffer(fid.read(), dtype=dtype)
    return x.reshape(dims) if dims[1] > 1 else x.copy()
