"""Diagnostic (not a test, not the product path): per-WORKGROUP timeline of one kernel class of one decode layer.

TL_KID = kernel id (2 gate/up, 3 down, 0 qkv, 1 o_proj, 7 attention, 9 combine), TL_LAYER = layer.  Uses the instrumented build
(make tl).  Prints the distribution over workgroups of entry, marks and exit (us relative to the earliest entry), per XCC the
mean finish time, and writes gpurun_out/timeline_wgs_<kid>.json."""
import os, sys, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "unimoe_audio_amd", "csrc")
subprocess.check_call(["make", "-C", CSRC, "tl", "-j8", "-s"])
os.environ["UMOE_HIP_LIB"] = os.path.join(CSRC, "libumoe_hip_tl.so")
sys.path.insert(0, ROOT)
import ctypes as C
import torch
import bench
from unimoe_audio_amd import _lib
from unimoe_audio_amd.codec_utils import prepare_audio_prompt
from unimoe_audio_amd.config import UniMoEAudioConfig
from unimoe_audio_amd.model import UniAudioRVQQwen2_5VLMoEForConditionalGeneration

LAYERS = int(os.environ.get("TL_LAYERS", "4"))
KIDS = [int(k) for k in os.environ.get("TL_KID", "2,3").split(",")]
LAYER = int(os.environ.get("TL_LAYER", "2"))
REPS = int(os.environ.get("TL_REPS", "12"))
dev = torch.device("cuda:0")
L = _lib.lib()
NK, NL = 16, 64
CTR = NL * NK * 16
tl = torch.zeros(CTR + 8 + 1024 * 12, dtype=torch.int64, device=dev)
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
for _ in range(4):
    eng.step(True)
res = {}
for kid in KIDS:
    init = torch.zeros(NL, NK, 16, dtype=torch.int64)
    init[:, :, 0] = (1 << 62)
    tailw = torch.zeros(8 + 1024 * 12, dtype=torch.int64)
    tailw[1] = 2 * B
    tailw[2] = kid + 1
    tailw[3] = LAYER
    init = torch.cat([init.reshape(-1), tailw]).to(dev)
    runs = []
    for r in range(REPS):
        tl.copy_(init)
        torch.cuda.synchronize()
        eng.step(True)
        torch.cuda.synchronize()
        d = tl.cpu()[CTR + 8:].reshape(1024, 12)
        runs.append(d[d[:, 9] == 1].clone())
    d = runs[-1]
    n = d.shape[0]
    t0 = int(d[:, 0].min())
    ent = (d[:, 0] - t0).double() / 100
    ex = (d[:, 7] - t0).double() / 100
    xcc = (d[:, 8] >> 32) & 0xf
    print(f"## kernel id {kid}, layer {LAYER}: {n} workgroups recorded (last of {REPS} replays)")
    def q(v):
        v = v.sort().values
        return " ".join(f"{float(v[int(p * (len(v) - 1))]):.2f}" for p in (0, .1, .5, .9, 1.0))
    print(f"entry  (min p10 p50 p90 max): {q(ent)}")
    for k in range(1, 7):
        m = d[:, k]
        ok = m > 0
        if ok.any():
            print(f"mark m{k + 3} ({int(ok.sum())} wgs): {q((m[ok] - t0).double() / 100)}")
    print(f"exit   (min p10 p50 p90 max): {q(ex)}")
    print(f"span per workgroup (exit - entry): {q(ex - ent)}")
    for xc in sorted(set(xcc.tolist())):
        sel = xcc == xc
        print(f"  xcc {xc}: {int(sel.sum())} wgs, entry mean {float(ent[sel].mean()):.2f}, exit mean {float(ex[sel].mean()):.2f} max {float(ex[sel].max()):.2f}")
    # all replays: last exit - first entry
    sp = [float((r[:, 7].max() - r[:, 0].min())) / 100 for r in runs if r.shape[0]]
    print("in-kernel span over replays:", " ".join(f"{v:.1f}" for v in sp))
    res[kid] = dict(entry=ent.tolist(), exit=ex.tolist(), xcc=xcc.tolist(), marks=((d[:, 1:7] - t0).double() / 100).tolist())
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "timeline_wgs.json"), "w"))
