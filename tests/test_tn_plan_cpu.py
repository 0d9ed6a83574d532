"""Host logic of umoe_tiled_gemm_tn (no GPU): the K split the library chooses for the weight-gradient products of a training step
(model fitted on MI355X: 20 us + 0.55 us per 32-row K tile per round of workgroups, reduction at 5 TB/s; csrc/umoe_tgemm_tn.hip)."""
import ctypes as C

import pytest

from unimoe_audio_amd import _lib as L


def _split(groups, ldo, k_split=-1):
    arr = (L.TnGroup * len(groups))()
    for i, g in enumerate(groups):
        for k, v in g.items():
            setattr(arr[i], k, v)
    a = L.TGemmTnArgs(groups=C.cast(arr, C.c_void_p), num_groups=len(groups), ldo=ldo, k_split=k_split)
    return L.lib().umoe_tiled_gemm_tn_split(C.byref(a))


@pytest.mark.parametrize("name,groups,ldo,expect", [
    ("QKV dW, 80 tiles", [dict(m=2560, n=2048, k=6240)], 2048, 3),
    ("o_proj dW, 64 tiles", [dict(m=2048, n=2048, k=6240)], 2048, 4),
    ("shared gate|up dW, 176 tiles", [dict(m=1376, n=2048, k=6240, out_row_base=i * 1376) for i in range(4)], 2048, 1),
    ("shared down dW, 96 tiles", [dict(m=2048, n=1376, k=6240, out_row_base=i * 2048) for i in range(2)], 1376, 2),
    ("codec head dW, 392 tiles", [dict(m=12324, n=2048, k=6240)], 2048, 1),
    ("router gate dW, 8 tiles", [dict(m=16, n=2048, k=6240)], 2048, 8),
    ("short contraction: no split below 16 K tiles per part", [dict(m=256, n=256, k=600)], 256, 1),
])
def test_k_split_choice(monkeypatch, name, groups, ldo, expect):
    monkeypatch.setenv("UMOE_FAKE_CUS", "256")
    assert _split(groups, ldo) == expect, name


def test_k_split_needs_one_dense_slab_and_static_windows():
    """Per-group output pointers, column windows of a wider output or device-side windows: never split (the reduction walks ONE slab)."""
    assert _split([dict(m=2048, n=2048, k=6240, out_col_off=8)], 4096) == 1
    assert _split([dict(m=2048, n=1024, k=6240)], 2048) == 1
    assert _split([dict(m=2048, n=2048, k=6240)], 2048, k_split=5) == 5
    assert _split([dict(m=2048, n=2048, k=6240)], 2048, k_split=0) == 1
