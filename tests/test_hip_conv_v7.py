"""conv_v7_kernel (the persistent form of the dominant convolution, use_conv_v7.hip; dispatched only with
use_set_option("conv_v7_min_units", n > 0)) against conv_v4_kernel in the single-convolution harness (use_conv_bench): same operator,
same K order - the stored outputs must be bit-identical; with a residual (served as an identity shortcut: one exact MFMA product per
element instead of a VALU add) a handful of elements may differ by one ulp of the storage type.  GroupNorm totals: conv_v7 sums the
stored values on the matrix pipe (squares rounded to bf16 per element), conv_v4 the fp32 values - equal to ~1e-4 of the largest total."""
import ctypes as C

import numpy as np
import pytest

from universal_speech_enhancement_amd import _lib
from universal_speech_enhancement_amd._lib import UseConvCase

pytestmark = pytest.mark.gpu


def _run(case, variant, B, dtype):
    H, W, C0, C1, Cout, XC0, XC1, act, gn, temb, res = case
    c = UseConvCase(B, H, W, C0, C1, Cout, XC0, XC1, act, gn, temb, res, 1, dtype, variant, 1)
    out = np.empty((B, H, W, Cout), np.float32)
    st = np.empty((B, Cout, 2), np.float32)
    ms, fl = C.c_double(), C.c_double()
    rc = _lib.lib().use_conv_bench(C.byref(c), out.ctypes.data_as(C.c_void_p), st.ctypes.data_as(C.c_void_p), C.byref(ms), C.byref(fl))
    assert rc == 0, _lib.lib().use_last_error()
    return out, st


# name: (H, W, C0, C1, Cout, XC0, XC1, act, gn, temb, res), batch
CASES = {
    "one tile, plain":               ((16, 32, 128, 0, 128, 0, 0, 0, 0, 0, 0), 1),
    "one tile, GN + SiLU + temb":    ((16, 32, 128, 0, 128, 0, 0, 1, 1, 1, 0), 1),
    "2x2 tiles, two channel blocks": ((32, 64, 128, 0, 256, 0, 0, 1, 1, 1, 0), 1),
    "4x4 tiles, concat input, B=3":  ((64, 128, 128, 64, 128, 0, 0, 1, 1, 1, 0), 3),
    "fused shortcut over a concat":  ((32, 64, 128, 0, 128, 128, 128, 1, 1, 0, 0), 2),
    "shortcut 384, B=3":             ((64, 128, 128, 0, 128, 256, 128, 1, 1, 0, 0), 3),
    "two channel blocks + shortcut": ((32, 64, 128, 128, 256, 256, 0, 1, 1, 1, 0), 2),
    "residual":                      ((64, 128, 128, 0, 128, 0, 0, 1, 1, 1, 1), 3),
}


@pytest.mark.parametrize("dtype", [1, 2])
@pytest.mark.parametrize("name", list(CASES))
def test_conv_v7_equals_conv_v4(name, dtype):
    case, B = CASES[name]
    a, sa = _run(case, 4, B, dtype)
    b, sb = _run(case, 8, B, dtype)
    assert np.isfinite(b).all()
    diff = np.abs(a - b)
    if case[10]:                                             # residual: rare one-ulp differences
        ulp = np.maximum(np.abs(a), 1e-30) * (2.0 ** -7 if dtype == 1 else 2.0 ** -10)
        assert (diff <= ulp).all() and (diff > 0).mean() < 1e-4, (name, float(diff.max()), float((diff > 0).mean()))
    else:
        assert diff.max() == 0.0, (name, float(diff.max()))
    rel = float(np.abs(sa - sb).max() / np.abs(sa).max())
    assert rel < 1e-3, (name, rel)
