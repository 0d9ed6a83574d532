"""Diagnostic (not a test, not the product path): in-kernel timeline of one decode layer at the headline shapes.

Loads the instrumented build (make -C unimoe_audio_amd/csrc tl -> libumoe_hip_tl.so, -DUMOE_TIMELINE), runs a
LAYERS-layer full-size model through the captured decode graph and prints, per kernel class of the LAST layer's
launches, first-workgroup entry / last entry / last exit and the marks of workgroup (0,0,0), in microseconds relative
to the QKV kernel's first entry.  wall_clock64() ticks at 100 MHz (10 ns).  Caches are flushed between replays
(512 MiB write) so weights stream cold, as they do in the 36-layer loop.
"""
import os, sys, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "unimoe_audio_amd", "csrc")
subprocess.check_call(["make", "-C", CSRC, "tl", "-j4", "-s"])
os.environ["UMOE_HIP_LIB"] = os.path.join(CSRC, "libumoe_hip_tl.so")
sys.path.insert(0, ROOT)
import ctypes as C
import torch
import bench
from unimoe_audio_amd import _lib
from unimoe_audio_amd.codec_utils import prepare_audio_prompt
from unimoe_audio_amd.config import UniMoEAudioConfig
from unimoe_audio_amd.model import UniAudioRVQQwen2_5VLMoEForConditionalGeneration

LAYERS = int(os.environ.get("TL_LAYERS", "1"))
REPS = int(os.environ.get("TL_REPS", "24"))
dev = torch.device("cuda:0")
# the timeline buffer must be installed before ANY instrumented kernel runs (null pointer otherwise)
L = _lib.lib()
NK = 16
NL = 64
tl = torch.zeros(NL * NK * 16 + 8, dtype=torch.int64, device=dev)
for name in ("gemm", "router", "attn", "misc"):
    fn = getattr(L, "umoe_tl_set_" + name)
    fn.argtypes = [C.c_void_p]
    fn.restype = C.c_int
    assert fn(tl.data_ptr()) == 0, name
cfg = UniMoEAudioConfig()
cfg.num_hidden_layers = LAYERS
torch.set_default_dtype(torch.bfloat16)
with torch.device(dev):
    model = UniAudioRVQQwen2_5VLMoEForConditionalGeneration(cfg)
torch.set_default_dtype(torch.float32)
model.init_synthetic(1234).eval()
B, T = 8, 300
eng = model.engine(B, T, 256, attn_splits=8)
ids, am, codec = bench.synth_prompt(cfg, B, T, dev)
x = model.calculate_input_embedding(ids, codec)
eng.prefill(x.reshape(-1, cfg.hidden_size).contiguous(), am)
pre, psteps = prepare_audio_prompt(cfg, [None] * B)
eng.start_decode(pre, psteps, 256, 256, cfg_scale=3.0, temperature=1.2, top_p=0.95, top_k=45, eos_mul=0.8, do_sample=True, seed=1)

trash = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
init = torch.zeros(NL, NK, 16, dtype=torch.int64)
init[:, :, 0] = (1 << 62)
tailw = torch.zeros(8, dtype=torch.int64)
tailw[1] = 2 * B          # combine workgroups per launch (one per row)
init = torch.cat([init.reshape(-1), tailw]).to(dev)
for _ in range(4):
    eng.step(True)
rows = []
for r in range(REPS):
    if LAYERS == 1:
        trash.fill_(r & 255)
    tl.copy_(init)
    torch.cuda.synchronize()
    eng.step(True)
    torch.cuda.synchronize()
    rows.append(tl.cpu()[: NL * NK * 16].reshape(NL, NK, 16).clone())
allr = torch.stack(rows).double()           # [REPS, NL, NK, 16]
names = {0: "qkv", 1: "o_proj", 2: "gate_up", 3: "down", 5: "router", 6: "dispatch", 7: "attn", 8: "attn_combine", 9: "combine"}
order = [0, 7, 8, 1, 5, 6, 2, 3, 9]
marks = {3: "wg0", 4: "m4", 5: "m5", 6: "m6", 7: "m7", 8: "m8"}
lays = [0] if LAYERS == 1 else list(range(1, LAYERS))      # layer 0 follows the embedding, not a combine
print(f"# {LAYERS}-layer model, {REPS} replays; median over replays" + (" and layers 1.." if LAYERS > 1 else "") +
      "; us relative to the QKV kernel's first workgroup entry of the same layer (wall_clock64, 10 ns ticks)")
print("| kernel | first entry | last entry | last exit | in-kernel span | boundary before | wg0: entry, marks... |")
print("|---|---|---|---|---|---|---|")
out = {}
prev_exit = None
for k in order:
    rel = torch.stack([allr[:, l, k, :] - allr[:, l, 0, 0:1] for l in lays], 1) / 100.0     # [REPS, nl, 16] us
    med = rel.reshape(-1, 16).median(0).values
    valid = allr[:, lays[0], k, :].median(0).values > 0
    ms = " ".join(f"{marks.get(j, 'm%d' % j)}={med[j]:.2f}" for j in range(3, 10) if valid[j])
    gap = "" if prev_exit is None else f"{med[0] - prev_exit:.2f}"
    print(f"| {names[k]} | {med[0]:.2f} | {med[1]:.2f} | {med[2]:.2f} | {med[2] - med[0]:.2f} | {gap} | {ms} |")
    prev_exit = float(med[2])
    out[names[k]] = [round(float(v), 2) for v in med[:10]]
if LAYERS > 1:
    per_layer = (allr[:, 2:LAYERS, 0, 0] - allr[:, 1:LAYERS - 1, 0, 0]).reshape(-1).median() / 100.0
    print(f"layer period (QKV entry to next QKV entry): {float(per_layer):.2f} us")
    out["layer_period_us"] = round(float(per_layer), 2)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", f"timeline_L{LAYERS}.json"), "w"), indent=1)
