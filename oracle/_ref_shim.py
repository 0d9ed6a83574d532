"""Import shims that let the reference's Python modules load in the BUILD CONTAINER.

TEST INFRASTRUCTURE ONLY -- used by oracle/gen_golden.py to produce tests/golden/*.npz.
The reference (/root/reference) never travels to the GPU box and is never copied here;
this file only fabricates the *third-party* names the reference imports and that are
absent offline (deepspeed, dac, audiotools, torchaudio, ...).  The stand-ins mirror what
the reference itself does in single-process mode: `_AllToAll.forward` is the identity
(reference utils/UniMoE_Audio_utils.py:332-335,429).

Nothing here is imported by the product package, the tests or the bench.
"""
from __future__ import annotations

import importlib.util
import os
import sys
import types

import torch

REF = os.environ.get("UMOE_REFERENCE", "/root/reference")


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    m.__spec__ = importlib.machinery.ModuleSpec(name, loader=None)
    sys.modules[name] = m
    return m


def install_stubs():
    import importlib.machinery  # noqa: F401
    # transformers probes for deepspeed with find_spec: import what we need first.
    import transformers  # noqa: F401
    from transformers.activations import ACT2FN  # noqa: F401
    import transformers.models.qwen2_5_vl.modeling_qwen2_5_vl as q

    if not hasattr(q, "Qwen2RMSNorm"):  # renamed in transformers 5.x; same math
        q.Qwen2RMSNorm = q.Qwen2_5_VLRMSNorm

    class _A2A(torch.autograd.Function):
        @staticmethod
        def forward(ctx, group, x):
            return x.contiguous()

        @staticmethod
        def backward(ctx, g):
            return None, g

    def gumbel_rsample(shape, device):
        g = torch.distributions.gumbel.Gumbel(torch.tensor(0.0, device=device), torch.tensor(1.0, device=device))
        return g.rsample(shape)

    class _Base(torch.nn.Module):
        pass

    ds = _mod("deepspeed")
    ds.__path__ = []
    ds.comm = _mod("deepspeed.comm", ProcessGroup=object, ReduceOp=types.SimpleNamespace(AVG=0, MAX=1))
    utils = _mod("deepspeed.utils", log_dist=lambda *a, **k: None)
    utils.__path__ = []
    utils.groups = _mod("deepspeed.utils.groups")
    ds.utils = utils
    _mod("deepspeed.utils.timer", SynchronizedWallClockTimer=type("SynchronizedWallClockTimer", (), {}))
    moe = _mod("deepspeed.moe")
    moe.__path__ = []
    ds.moe = moe
    moe.sharded_moe = _mod(
        "deepspeed.moe.sharded_moe",
        FIRST_ALLTOALL_TIMER="1st_a2a",
        MOE_TIMER="moe",
        SECOND_ALLTOALL_TIMER="2nd_a2a",
        _AllToAll=_A2A,
        einsum=torch.einsum,
        gumbel_rsample=gumbel_rsample,
        MOELayer=type("MOELayer", (_Base,), {}),
        _capacity=None,
        _one_hot_to_float=None,
        top2gating=None,
    )
    moe.experts = _mod("deepspeed.moe.experts", Experts=type("Experts", (_Base,), {}))
    moe.layer = _mod("deepspeed.moe.layer", MoE=type("MoE", (_Base,), {}))

    # codec / audio / vision third-party names (never executed by the golden generator)
    _mod("dac", DAC=object, utils=types.SimpleNamespace(download=None))
    _mod("audiotools", AudioSignal=object)
    ta = _mod("torchaudio", save=None)
    ta.transforms = _mod("torchaudio.transforms", Resample=object)
    tv = _mod("torchvision")
    tv.__path__ = []
    tv.transforms = _mod("torchvision.transforms", InterpolationMode=types.SimpleNamespace(BICUBIC=3))
    tv.transforms.functional = _mod("torchvision.transforms.functional")
    mp = _mod("moviepy")
    mp.__path__ = []
    _mod("moviepy.video").__path__ = []
    _mod("moviepy.video.io").__path__ = []
    _mod("moviepy.video.io.VideoFileClip", VideoFileClip=object)
    _mod("qwen_vl_utils", smart_resize=None)
    try:
        import PIL  # noqa: F401
    except Exception:
        pil = _mod("PIL")
        pil.__path__ = []
        _mod("PIL.Image")
        pil.Image = sys.modules["PIL.Image"]


def _load(pkg_name, mod_name, path):
    spec = importlib.util.spec_from_file_location(f"{pkg_name}.{mod_name}", path)
    m = importlib.util.module_from_spec(spec)
    sys.modules[f"{pkg_name}.{mod_name}"] = m
    spec.loader.exec_module(m)
    return m


def load_reference(full: bool = True):
    """Returns a namespace with .core, .moe_utils and (if `full`) .utils, .model, .mod."""
    install_stubs()
    ns = types.SimpleNamespace()
    pkg = _mod("refpkg")
    pkg.__path__ = [os.path.join(REF, "utils")]
    # dependency-free twin of compress/decompress (byte-identical modulo comments)
    ns.moe_utils = _load("refpkg", "MoE_utils", os.path.join(REF, "UniMoEV2-Preview/training/Models/MoE_utils.py"))
    if full:
        ns.utils = _load("refpkg", "UniMoE_Audio_utils", os.path.join(REF, "utils/UniMoE_Audio_utils.py"))
    else:
        sys.modules["refpkg.UniMoE_Audio_utils"] = ns.moe_utils
    ns.core = _load("refpkg", "UniMoE_Audio_core", os.path.join(REF, "utils/UniMoE_Audio_core.py"))
    if full:
        ns.model = _load("refpkg", "UniMoE_Audio_model", os.path.join(REF, "utils/UniMoE_Audio_model.py"))
        try:
            ns.mod = _load("refpkg", "UniMoE_Audio_mod", os.path.join(REF, "utils/UniMoE_Audio_mod.py"))
        except Exception as e:  # pragma: no cover - optional
            ns.mod = None
            ns.mod_error = repr(e)
    return ns


if __name__ == "__main__":
    ns = load_reference(True)
    print("core:", ns.core.UniMoEAudioSparseMoeBlock)
    print("utils:", ns.utils.DecoderOutput)
    print("model:", ns.model.UniAudioRVQQwen2_5VLMoEForConditionalGeneration)
    print("mod:", ns.mod, getattr(ns, "mod_error", ""))
