"""Task lists of the one-launch expert-parallel MoE half (unimoe_audio_amd/csrc/umoe_moe_ep.hip, epf_plan) -- host logic, no GPU:
every unit of every phase (shared gate/up pairs, local gate/up pairs, local down blocks, shared down blocks) is covered exactly once,
every task has a size the kernel has a pass shape for, riders come first, publishes / the count-in sit behind the phase they close,
and the lists fit their fixed length, for the ep sizes and the workgroup counts the engine uses (256 = one rank per card; 128 / 64 /
32 = 2 / 4 / 8 ranks sharing one card in the tests)."""
import ctypes as C

import numpy as np
import pytest

from unimoe_audio_amd import _lib

MAXT = 24
A, PUB_A, B, PUB_B, CC, SIG_C, D, TILE, ROUTER = 1, 2, 3, 4, 5, 6, 7, 8, 9


def plan(n_wg, R, S=16, Dm=2048, I_dyn=2752, I_sh=1376, n_fix=2):
    L = _lib.lib()
    L.umoe_moe_ep_plan_probe.argtypes = [C.c_int] * 8 + [C.POINTER(C.c_uint32), C.c_int]
    out = (C.c_uint32 * (2 + n_wg * MAXT))()
    assert L.umoe_moe_ep_plan_probe(n_wg, R, 8 // R, S, Dm, I_dyn, I_sh, n_fix, out, len(out)) == 0
    a = np.frombuffer(out, dtype=np.uint32).copy()
    return bool(a[0]), int(a[1]), a[2:].reshape(n_wg, MAXT)


@pytest.mark.parametrize("n_wg,R", [(256, 2), (256, 4), (256, 8), (120, 2), (60, 4), (30, 8), (240, 8), (200, 2)])
def test_ep_task_lists_cover_every_unit_once(n_wg, R):
    ok, n_cwg, t = plan(n_wg, R)
    assert ok
    E_loc, S = 8 // R, 16
    halves = 2 if R == 8 else 1                                       # at 8 row tiles a down pass takes four of them
    cover = {A: np.zeros(2 * 86, int), B: np.zeros((E_loc, halves, 172), int), CC: np.zeros((E_loc, halves, 128), int), D: np.zeros((2, 128), int)}
    legal_b = {2: {1, 2, 3, 4}, 4: {1, 2}, 8: {1, 2}}[R]
    RT = 2 * R if n_wg >= 128 else R                                  # with roles: re-lay riders [0, R), push riders [R, 2R)
    legal_c = {1, 2, 3, 4}
    counted = 0
    for w in range(n_wg):
        kinds = []
        assert t[w, MAXT - 1] == 0                                   # the terminator survives
        for word in t[w]:
            if word == 0:
                break
            kind, grp, first, n = int(word >> 28), int((word >> 24) & 15), int((word >> 8) & 0xffff), int(word & 255)
            kinds.append(kind)
            if kind == A:
                assert 1 <= n <= 7
                cover[A][first:first + n] += 1
            elif kind == B:
                assert n in legal_b and first + n <= 172 and (grp & 3) < E_loc and (grp >> 2) < halves
                cover[B][grp & 3, grp >> 2, first:first + n] += 1
            elif kind == CC:
                assert n in legal_c and first + n <= 128 and (grp & 3) < E_loc and (grp >> 2) < halves
                cover[CC][grp & 3, grp >> 2, first:first + n] += 1
            elif kind == D:
                assert 1 <= n <= 10 and first + n <= 128
                cover[D][grp, first:first + n] += 1
            elif kind == TILE:
                assert (w == grp < R and n == (2 if n_wg >= 128 else 3)) or (n_wg >= 128 and R <= w < RT and grp == w - R and n == 1)
            elif kind == ROUTER:
                assert RT <= w < RT + S and first == w - RT
        # order: rider | A .. publish | B .. publish | C .. count-in, with D behind C (every workgroup takes every phase: D hides the
        # return flight) or in front of it (roles: D only needs A, C hangs on everybody's B)
        core = [k for k in kinds if k not in (TILE, ROUTER)]
        assert kinds[:len(kinds) - len(core)] == [k for k in kinds if k in (TILE, ROUTER)]
        rank = {A: 0, PUB_A: 1, B: 2, PUB_B: 3, CC: 4, SIG_C: 5, D: 6 if n_wg < 128 else 3.5}
        assert [rank[k] for k in core] == sorted(rank[k] for k in core)
        if n_wg >= 128:
            assert not (B in core and D in core)                     # two roles
        assert (A in core) == (PUB_A in core) and (B in core) == (PUB_B in core) and (CC in core) == (SIG_C in core)
        assert core.count(PUB_A) <= 1 and core.count(PUB_B) <= 1 and core.count(SIG_C) <= 1
        counted += SIG_C in core
    for k, c in cover.items():
        assert (c == 1).all(), (k, np.argwhere(c != 1)[:5])
    assert counted == n_cwg > 0


def test_ep_plan_rejects_shapes_without_pass_shapes():
    assert not plan(20, 8)[0]            # fewer workgroups than riders
    assert not plan(256, 8, Dm=4096)[0]  # the tile riders' normalisation is written for D = 2048
