"""Operators with more than 2^31 stored entries (64-bit row offsets, include/saamge_amd.h: *_64 entry points).

The reference's HYPRE_Int is 32-bit; it reaches such operators only split over MPI ranks (BASELINE config 5, Q2
elasticity on 96^3 elements: 4.2e9 entries).  Here one GPU holds the operator, so every CSR / SELL offset inside the
library is 64-bit.  The test multiplies a banded matrix with 2.2e9 entries, built on the device, by a vector and
compares with the closed form of the same sum evaluated by torch -- through the CSR kernel and through the SELL-64
copy the level operators use.  (About 60 GB of device memory.)"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _banded(n, w, dev):
    import torch
    rowptr = torch.arange(n + 1, dtype=torch.int64, device=dev) * w
    col = torch.empty(n * w, dtype=torch.int32, device=dev)
    val = torch.empty(n * w, dtype=torch.float64, device=dev)
    offs = torch.arange(w, dtype=torch.int64, device=dev)
    step = 1 << 16
    for r0 in range(0, n, step):
        r1 = min(n, r0 + step)
        rows = torch.arange(r0, r1, dtype=torch.int64, device=dev)
        c = (rows[:, None] + offs[None, :] * 3) % n                       # w distinct columns per row
        v = ((rows[:, None] * 7 + offs[None, :] * 13) % 17 + 1).double() / 17.0
        c, order = torch.sort(c, dim=1)                                    # ascending columns inside a row
        col[r0 * w:r1 * w] = c.reshape(-1).int()
        val[r0 * w:r1 * w] = torch.gather(v, 1, order).reshape(-1)
    return rowptr, col, val


def _expected(n, w, x, dev):
    import torch
    rows = torch.arange(n, dtype=torch.int64, device=dev)
    y = torch.zeros(n, dtype=torch.float64, device=dev)
    for k in range(w):
        y += ((rows * 7 + k * 13) % 17 + 1).double() / 17.0 * x[(rows + 3 * k) % n]
    return y


@pytest.mark.parametrize("sell", [False, True])
def test_spmv_beyond_2_31_entries(sell):
    import torch
    from saamge_amd import capi
    dev = "cuda:0"
    n, w = 1_250_048, 1_800                      # 2.25e9 entries; n a multiple of 64
    assert n * w > 2 ** 31
    rowptr, col, val = _banded(n, w, dev)
    x = torch.cos(torch.arange(n, dtype=torch.float64, device=dev) * 1e-3)
    y = torch.zeros(n, dtype=torch.float64, device=dev)
    old = capi.set_options(spmv_sell=1 if sell else 0)
    try:
        capi.spmv_raw(n, n, rowptr, col, val, x, y)
    finally:
        capi.set_options(spmv_sell=old.spmv_sell)
    torch.cuda.synchronize()
    ref = _expected(n, w, x, dev)
    err = float(torch.max(torch.abs(y - ref)) / torch.max(torch.abs(ref)))
    assert err < 1e-12, err          # (fp64 sums of 1800 terms in different orders)
    # the last rows use the entries beyond offset 2^31
    assert float(torch.max(torch.abs(y[-64:] - ref[-64:]))) < 1e-9
    del rowptr, col, val, x, y, ref
    torch.cuda.empty_cache()
    capi.release_cached_memory()
