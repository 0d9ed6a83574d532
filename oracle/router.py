"""ctypes front-end of oracle/router_oracle.c (CPU restatement of the Top-P router).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py -- never by the product package.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "_build", "liboracle_router.so")
    src = os.path.join(_HERE, "router_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build())
        _LIB.umoe_oracle_router.restype = ctypes.c_int
        _LIB.umoe_oracle_exp_det.restype = ctypes.c_float
        _LIB.umoe_oracle_exp_det.argtypes = [ctypes.c_float]
        _LIB.umoe_oracle_dispatch.restype = ctypes.c_int
    return _LIB


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None else None


def route(logits: torch.Tensor, n_dyn: int, n_real: int, n_fix: int, top_p: float, fixed_top_k: int = 0,
          jitter_eps: float = 0.01, attn_mask: torch.Tensor | None = None) -> dict:
    """logits [S, n_dyn+n_fix], dtype bfloat16 or float32 (decides the arithmetic type T)."""
    assert logits.dim() == 2 and logits.shape[1] == n_dyn + n_fix
    S, E = logits.shape
    is_bf16 = logits.dtype == torch.bfloat16
    if is_bf16:
        raw = logits.contiguous().view(torch.int16).numpy().copy()
    else:
        assert logits.dtype == torch.float32
        raw = logits.contiguous().numpy().copy()
    am = None
    if attn_mask is not None:
        am = np.ascontiguousarray(attn_mask.reshape(-1).to(torch.uint8).numpy())
        assert am.shape[0] == S
    top_k = np.zeros(S, np.int64)
    sel = np.zeros((S, n_dyn), np.int32)
    mask = np.zeros((S, E), np.int32)
    route_w = np.zeros((S, n_dyn), np.float32)
    global_w = np.zeros((S, E), np.float32)
    moe_w = np.zeros((S, n_real), np.float32)
    rc = lib().umoe_oracle_router(
        _p(raw), ctypes.c_int(int(is_bf16)), ctypes.c_int(S), ctypes.c_int(n_dyn), ctypes.c_int(n_real),
        ctypes.c_int(n_fix), ctypes.c_float(top_p), ctypes.c_int(int(fixed_top_k)), ctypes.c_double(jitter_eps),
        _p(am), _p(top_k), _p(sel), _p(mask), _p(route_w), _p(global_w), _p(moe_w))
    if rc != 0:
        raise ValueError(f"umoe_oracle_router failed rc={rc}")
    dt = logits.dtype
    return dict(
        top_k=torch.from_numpy(top_k),
        sel=torch.from_numpy(sel),
        expert_mask=torch.from_numpy(mask),
        routing_weights=torch.from_numpy(route_w).to(dt),
        global_weight=torch.from_numpy(global_w).to(dt),
        moe_weight=torch.from_numpy(moe_w).to(dt),
    )


def dispatch(expert_mask: torch.Tensor, n_real: int) -> dict:
    """Per-expert token lists (ascending token index) from the 0/1 mask."""
    m = np.ascontiguousarray(expert_mask.to(torch.int32).numpy())
    S, ld = m.shape
    counts = np.zeros(n_real, np.int32)
    offsets = np.zeros(n_real + 1, np.int32)
    slot_token = np.zeros(max(1, S * n_real), np.int32)
    slot_of = np.zeros((S, n_real), np.int32)
    total = lib().umoe_oracle_dispatch(_p(m), ctypes.c_int(S), ctypes.c_int(ld), ctypes.c_int(n_real), _p(counts),
                                       _p(offsets), _p(slot_token), _p(slot_of))
    return dict(counts=torch.from_numpy(counts), offsets=torch.from_numpy(offsets),
                slot_token=torch.from_numpy(slot_token[:total].copy()), slot_of=torch.from_numpy(slot_of), total=total)


def exp_det(x: float) -> float:
    return float(lib().umoe_oracle_exp_det(ctypes.c_float(x)))
