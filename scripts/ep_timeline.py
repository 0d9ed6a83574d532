"""Diagnostic (not a test, not the product path): per-WORKGROUP timeline of the one-launch expert-parallel MoE half (umoe_moe_ep.hip) of the
LAST layer of a decode step, one rank of an N-rank job in loopback.  Uses the instrumented build (make tl).  Stamp 0 = entry, stamp k + 1
= end of task k of the workgroup's list, 15 = exit; prints, per task kind, when the tasks of that kind END (us after the earliest entry).
  python scripts/ep_timeline.py [ep=8]"""
import os, sys, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "unimoe_audio_amd", "csrc")
subprocess.check_call(["make", "-C", CSRC, "tl", "-j8", "-s"])
os.environ["UMOE_HIP_LIB"] = os.path.join(CSRC, "libumoe_hip_tl.so")
sys.path.insert(0, ROOT)
import ctypes as C
import numpy as np
import torch
import bench
from unimoe_audio_amd import _lib
from unimoe_audio_amd.codec_utils import prepare_audio_prompt
from unimoe_audio_amd.ep import EpLink
from unimoe_audio_amd.model import DecodeEngine

EP = int(sys.argv[1]) if len(sys.argv) > 1 else 8
LAYERS = int(os.environ.get("TL_LAYERS", "6"))
REPS = int(os.environ.get("TL_REPS", "8"))
dev = torch.device("cuda:0")
L = _lib.lib()
NK, NL = 16, 64
CTR = NL * NK * 16
tl = torch.zeros(CTR + 8 + 1024 * 12, dtype=torch.int64, device=dev)
tl[:CTR].view(NL, NK, 16)[:, :, 0] = (1 << 62)
tl[CTR + 1] = 16
for name in ("gemm", "router", "attn", "misc"):
    fn = getattr(L, "umoe_tl_set_" + name)
    fn.argtypes = [C.c_void_p]
    fn.restype = C.c_int
    assert fn(tl.data_ptr()) == 0, name
L.umoe_moe_ep_stamps.argtypes = [C.c_void_p]
assert L.umoe_moe_ep_stamps(None) == 0
import argparse
args = argparse.Namespace(layers=LAYERS, codec_channels=0, prompt=300)
cfg = bench.make_cfg(args)
model, _ = bench.build_model(cfg, dev)
B, T = 8, 300
eng = DecodeEngine(model, B, Lmax=T + 200, Tmax=200, ep=EpLink(0, EP, "loopback"))
ids, am, codec = bench.synth_prompt(cfg, B, T, dev)
x = model.calculate_input_embedding(ids, codec)
eng.prefill(x.reshape(-1, cfg.hidden_size).contiguous(), am)
pre, psteps = prepare_audio_prompt(cfg, [None] * B)
eng.start_decode(pre, psteps, 100, 100, cfg_scale=3.0, temperature=1.2, top_p=0.95, top_k=45, eos_mul=0.8, do_sample=True, seed=1)
MAXT = 24
plan = (C.c_uint32 * (2 + 256 * MAXT))()
L.umoe_moe_ep_plan_probe.argtypes = [C.c_int] * 8 + [C.POINTER(C.c_uint32), C.c_int]
n_wg = min(eng.info("n_cu"), 256)
assert L.umoe_moe_ep_plan_probe(n_wg, EP, 8 // EP, 2 * B, cfg.hidden_size, cfg.dynamic_intermediate_size, cfg.shared_intermediate_size, 2, plan, len(plan)) == 0
tasks = np.frombuffer(plan, dtype=np.uint32)[2:].reshape(256, MAXT)[:n_wg]
buf = np.zeros((256, 16), dtype=np.uint64)
for r in range(REPS):
    eng.step(True)
    torch.cuda.synchronize()
    assert L.umoe_moe_ep_stamps(buf.ctypes.data_as(C.c_void_p)) == 0
assert eng.ep_error() == 0
d = buf[:n_wg].astype(np.int64)
t0 = d[:, 0].min()
us = np.where(d > 0, (d - t0) / 100.0, np.nan)
names = {1: "A shared gate/up", 2: "publish A", 3: "B local gate/up", 4: "publish B", 5: "C local down", 6: "count-in", 7: "D shared down", 8: "tile rider", 9: "router rider"}
ends = {k: [] for k in names}
for w in range(n_wg):
    for k in range(min(MAXT, 14)):
        word = int(tasks[w, k])
        if word == 0:
            break
        ends[word >> 28].append(us[w, k + 1])
def q(v):
    v = np.sort(np.array(v)[~np.isnan(v)])
    if len(v) == 0:
        return "-"
    return " ".join(f"{v[int(p * (len(v) - 1))]:6.2f}" for p in (0, .1, .5, .9, 1.0)) + f"  (n={len(v)})"
print(f"## one-launch MoE half, ep {EP} loopback, {n_wg} workgroups, last layer of {LAYERS}; task ENDS in us after the earliest entry (min p10 p50 p90 max)")
print(f"   entry               {q(us[:, 0])}")
for k in (8, 9, 1, 2, 3, 4, 5, 6, 7):
    print(f"   {names[k]:18s}  {q(ends[k])}")
print(f"   exit                {q(us[:, 15])}")
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(dict(us=np.nan_to_num(us).tolist(), tasks=tasks.tolist()), open(os.path.join(ROOT, "gpurun_out", f"ep{EP}_timeline.json"), "w"))
