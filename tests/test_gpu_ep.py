"""GPU tests of the expert-parallel decode engine (umoe_engine_cfg.ep_size > 1; reference AudioMOELayer.forward with its two
all-to-alls, utils/UniMoE_Audio_core.py:446-493).  The bar is bit-identity with ep_size 1: every (expert, 16-row tile) product is
computed by the same kernel instantiation with the same K split, the owner combines in ascending expert order.
The ranks run as CHILD processes (this pytest process has initialised the GPU and must not be replaced; children are fine):
 * virtual ranks: N engines of one process on N streams (GPU_MAX_HW_QUEUES=8 so that no two streams share a hardware queue --
   a receive that spins in front of the peer's send on the same queue would only end by its timeout);
 * processes: one engine per process, exchange regions mapped through HIP IPC, control plane on gloo -- the deployment path,
   rehearsed with both ranks on this one GPU."""
import os
import subprocess
import sys

import pytest
import torch

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _run(cmd, env_extra=None, timeout=600):
    env = dict(os.environ)
    env.update(env_extra or {})
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    return p.returncode, p.stdout + p.stderr


@pytest.mark.parametrize("ranks,graph,flat", [(2, 0, 1), (4, 1, 1), (8, 1, 1), (2, 1, 0), (4, 0, 0)])
def test_ep_virtual_ranks_bit_identical(ranks, graph, flat):
    """2 / 4 / 8 ranks on one card (8 = BASELINE configs[3]'s own shape: one local expert per rank), eager and hipGraph replay, the MoE
    half as ONE launch per rank (umoe_moe_ep.hip; each rank's launch takes 1/N of the CUs) and as the launch-per-kernel exchange
    (UMOE_EP_FLAT=0): logits of every step and all tokens bit-identical to N independent ep_size 1 engines."""
    assert torch.cuda.is_available()
    rc, out = _run([sys.executable, "scripts/ep_virtual.py", str(ranks), "2", "5", str(graph)],
                   {"GPU_MAX_HW_QUEUES": "32", "UMOE_EP_FLAT": str(flat)})
    assert rc == 0 and "BIT-IDENTICAL" in out, out[-3000:]


def test_ep_two_processes_hip_ipc_bit_identical():
    assert torch.cuda.is_available()
    import random
    port = random.randint(20000, 40000)
    rc, out = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                    "--master-port", str(port), "scripts/ep_multiproc.py", "2", "5"])
    assert rc == 0 and "BIT-IDENTICAL" in out, out[-3000:]


def test_ep_sharded_weights_two_processes_bit_identical():
    """Every rank holds ONLY its local experts (core.py:505): packed local experts in the engine, prompt through the module-level forward
    with the blocks' exchange between the ranks, KV cache handed to the engine, expert-parallel decode -- tokens and logits bit-identical
    to ep_size 1 on the full model, routed-expert bytes per rank = 1 / N (scripts/ep_sharded_multiproc.py)."""
    assert torch.cuda.is_available()
    import random
    port = random.randint(20000, 40000)
    rc, out = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                    "--master-port", str(port), "scripts/ep_sharded_multiproc.py", "2", "5"])
    assert rc == 0 and "SHARDED-BIT-IDENTICAL" in out, out[-3000:]


def test_ep_training_step_two_processes_vs_single_block():
    """Expert-parallel training of one DCMoE block (ep_size 2, two processes, the exchange as part of the HIP block's forward and backward,
    core.py:455-488 under autograd) against the ep_size 1 block: outputs, input gradients, gate / shared gradients and the local experts'
    weight gradients summed over both ranks' rows (scripts/ep_train_multiproc.py)."""
    assert torch.cuda.is_available()
    import random
    port = random.randint(20000, 40000)
    rc, out = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                    "--master-port", str(port), "scripts/ep_train_multiproc.py"])
    assert rc == 0 and "EP-TRAIN-OK" in out, out[-3000:]


def test_ep_loopback_emulation_runs_clean():
    """One rank of an 8-rank job in loopback (bench.py --ep-emulate): every receive is satisfied by the engine's own sends, no
    timeout, and the sampler still produces valid codes."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_gpu_engine import build, prompt, small_cfg
    from unimoe_audio_amd.codec_utils import prepare_audio_prompt
    from unimoe_audio_amd.ep import EpLink
    from unimoe_audio_amd.model import DecodeEngine
    dev = torch.device("cuda:0")
    cfg = small_cfg(hidden_size=2048, num_attention_heads=16, num_key_value_heads=2, dynamic_intermediate_size=2752,
                    shared_intermediate_size=1376)
    m, _ = build(cfg, 1, 0.02)
    gm = m.to(dev)
    B, T, MAXT = 8, 12, 46
    ids, am, codec = prompt(cfg, B, T, 2, [3, 0, 1, 0] + [0] * 12)
    pre, psteps = prepare_audio_prompt(cfg, [None] * B)
    eng = DecodeEngine(gm, B, Lmax=T + MAXT + 8, Tmax=MAXT + 64, ep=EpLink(0, 8, "loopback"))
    x = gm.calculate_input_embedding(ids.to(dev), codec.to(dev))
    eng.prefill(x.reshape(-1, cfg.hidden_size).contiguous(), am.to(dev))
    eng.start_decode(pre, psteps, MAXT, 6, cfg_scale=3.0, temperature=1.2, top_p=0.95, top_k=45, eos_mul=0.8, do_sample=True, seed=1)
    for s in range(6):
        eng.step(use_graph=(s >= 2))
    assert eng.ep_error() == 0
    pred = eng.copy_buffer("pred", torch.int64, (B, cfg.codec_channels)).cpu()
    assert int(pred.min()) >= 0 and int(pred.max()) < cfg.codec_vocab_size
    eng.close()


def test_ep_engine_rejects_bad_geometry():
    import ctypes as C
    from unimoe_audio_amd import _lib as L
    from unimoe_audio_amd.config import UniMoEAudioConfig
    cfg = UniMoEAudioConfig()
    sec = list(cfg.mrope_section)
    for ep_size, rank, rows in ((3, 0, 16), (2, 2, 16), (2, 0, 18)):
        c = L.EngineCfg(hidden=cfg.hidden_size, layers=1, heads=cfg.num_attention_heads, kv_heads=cfg.num_key_value_heads,
                        head_dim=cfg.head_dim, n_dyn=cfg.num_dyn, n_real=cfg.mlp_dynamic_expert_num, n_fix=cfg.mlp_fixed_expert_num,
                        inter_dyn=cfg.dynamic_intermediate_size, inter_shared=cfg.shared_intermediate_size,
                        codec_channels=cfg.codec_channels, codec_vocab=cfg.codec_vocab_size, eos=cfg.codec_eos_value,
                        pad=cfg.codec_pad_value, bos=cfg.codec_bos_value, mrope0=sec[0], mrope1=sec[1], mrope2=sec[2],
                        rms_eps=cfg.rms_norm_eps, top_p=0.7, fixed_top_k=0, jitter_eps=0.01, rows=rows, Lmax=64, Tmax=64,
                        attn_splits=1, ep_rank=rank, ep_size=ep_size)
        h = C.c_void_p()
        assert L.lib().umoe_engine_create(C.byref(c), C.byref(h)) != 0
        assert b"expert parallel" in L.lib().umoe_last_error()
