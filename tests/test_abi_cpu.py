"""CPU suite: the C-ABI library loads and exports every symbol include/umoe.h declares (no compute calls)."""
import ctypes
import os
import re

from conftest import ROOT


def _declared():
    txt = open(os.path.join(ROOT, "include", "umoe.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    names = re.findall(r"\b(umoe_[a-z0-9_]+)\s*\(", txt)
    return sorted(set(names))


def test_library_builds_and_exports_every_declared_symbol():
    from unimoe_audio_amd import _lib
    so = _lib.build()
    L = ctypes.CDLL(so)
    decl = _declared()
    assert len(decl) >= 25
    missing = [n for n in decl if not hasattr(L, n)]
    assert missing == []
    assert set(_lib.EXPORTS) <= set(decl)
    assert _lib.lib().umoe_abi_version() == 1
    assert _lib.lib().umoe_packed_elems(33, 64) == 48 * 64


def test_product_has_no_cpu_fallback():
    """CPU tensors must raise, never silently compute."""
    import pytest
    import torch
    from unimoe_audio_amd import _lib, ops
    with pytest.raises(_lib.UmoeError):
        ops.pack_weight(torch.zeros(16, 32, dtype=torch.bfloat16))
    from unimoe_audio_amd.config import UniMoEAudioConfig
    from unimoe_audio_amd.dcmoe import UniMoEAudioSparseMoeBlock
    blk = UniMoEAudioSparseMoeBlock(UniMoEAudioConfig.tiny())
    with pytest.raises(_lib.UmoeError):
        blk(torch.zeros(1, 2, 128, dtype=torch.bfloat16), None, None)


def test_product_never_imports_oracle():
    for dp, _, fs in os.walk(os.path.join(ROOT, "unimoe_audio_amd")):
        for f in fs:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dp, f)).read()
                assert "from oracle" not in src and "import oracle" not in src, f
                assert "#include \"../../oracle" not in src and "router_oracle" not in src.replace("oracle/router_oracle.c", ""), f
