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


def test_no_mfma_is_predicated_through_exec(tmp_path):
    """MFMA ignores EXEC: a guard around an MFMA must be a scalar (wave-uniform) branch.  Compile the MFMA kernels to assembly
    and check that no v_mfma directly follows an s_and_saveexec (scripts/scan_mfma_exec.py; the hazard doubled the last k-step of
    partial chunks in the weight-streaming GEMM before the slice bounds were pinned into SGPRs)."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "scripts"))
    from scan_mfma_exec import scan
    csrc = os.path.join(root, "unimoe_audio_amd", "csrc")
    hipcc = "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not found")
    procs = []
    for f in ("umoe_gemm", "umoe_moe_flat", "umoe_tgemm", "umoe_attn", "umoe_attn_bwd"):
        out = str(tmp_path / (f + ".s"))
        procs.append((f, out, subprocess.Popen([hipcc, "-O3", "-std=c++17", "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt",
                                                "--offload-arch=gfx950", "-I" + os.path.join(root, "include"), "-I" + csrc, "--cuda-device-only", "-S",
                                                os.path.join(csrc, f + ".hip"), "-o", out], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)))
    for f, out, pr in procs:
        assert pr.wait(timeout=600) == 0, f
        n, bad = scan(out)
        assert n > 0 and not bad, (f, bad[:3])


def test_ctypes_mirrors_have_the_size_and_offsets_of_the_c_structs(tmp_path):
    """unimoe_audio_amd/_lib.py mirrors every argument struct of include/umoe.h by hand: compile a probe with gcc and compare
    sizeof and the offset of every field (the field NAMES must exist in the C struct too)."""
    import ctypes as C
    import subprocess
    from unimoe_audio_amd import _lib as L
    pairs = L.STRUCT_MIRRORS
    assert len(pairs) == 16
    for cname, cls in pairs.items():                       # the built library agrees too (checked again at every load)
        assert int(L.lib().umoe_struct_size(cname.encode())) == C.sizeof(cls), cname
    assert int(L.lib().umoe_struct_size(b"no_such_struct")) == 0
    src = ['#include <stdio.h>', '#include <stddef.h>', '#include "umoe.h"', 'int main(void) {']
    for cname, cls in pairs.items():
        src.append(f'  printf("{cname} %zu\\n", sizeof({cname}));')
        for fname, _ in cls._fields_:
            src.append(f'  printf("{cname}.{fname} %zu\\n", offsetof({cname}, {fname}));')
    src += ['  return 0;', '}']
    cfile, exe = tmp_path / "probe.c", tmp_path / "probe"
    cfile.write_text("\n".join(src))
    subprocess.check_call(["gcc", "-I" + os.path.join(ROOT, "include"), str(cfile), "-o", str(exe)])
    got = dict(line.split() for line in subprocess.check_output([str(exe)], text=True).splitlines())
    for cname, cls in pairs.items():
        assert int(got[cname]) == C.sizeof(cls), (cname, got[cname], C.sizeof(cls))
        for fname, _ in cls._fields_:
            assert int(got[f"{cname}.{fname}"]) == getattr(cls, fname).offset, (cname, fname)


def test_restaging_barriers_behind_transposing_reads_are_pinned():
    """ds_read_b64_tr_b16 (__builtin_amdgcn_ds_read_tr16_b64): hipcc (ROCm 7.2) emitted such reads BEHIND the __syncthreads() that
    ends an iteration, where the next tile's stores race with them (umoe_attn_bwd.hip header).  The fix is structural: in front of
    every barrier behind which a tile read by transposing reads is restaged, the accumulators those reads feed pass through a
    volatile asm (TR_PIN8 / the acc_o pin), which cannot move across s_barrier.  This test keeps the structure: in the two sources
    that use the builtin, every `lstore(buf ^ 1)` (the restage) is followed by a pin before its `__syncthreads()`."""
    import re
    csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "unimoe_audio_amd", "csrc")
    found = 0
    for f in ("umoe_attn.hip", "umoe_attn_bwd.hip"):
        src = open(os.path.join(csrc, f)).read()
        assert "__builtin_amdgcn_ds_read_tr16_b64" in src
        for m in re.finditer(r"lstore\(buf \^ 1\);", src):
            tail = src[m.end(): m.end() + 700]
            sync = tail.index("__syncthreads()")
            assert ("TR_PIN8(" in tail[:sync]) or ('"+v"(acc_o[0])' in tail[:sync]), (f, src[: m.start()].count("\n") + 1)
            found += 1
    assert found >= 3


def test_integration_doc_router_stub_mirrors_the_whole_struct():
    """INTEGRATION.md shows the binding a reference maintainer would write; a mirror SHORTER than umoe_router_args would let the library
    read past it (trailing pointer fields), so the documented field list must be the one of unimoe_audio_amd/_lib.py, in order."""
    import re
    from unimoe_audio_amd import _lib as L
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    block = doc[doc.index("class RouterArgs(C.Structure)"):doc.index("def router(hidden_states")]
    names = re.findall(r'\("(\w+)",\s*C\.c_', block)
    assert names == [n for n, _ in L.RouterArgs._fields_], (names, [n for n, _ in L.RouterArgs._fields_])
