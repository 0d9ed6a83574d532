"""Diagnostic (not a test, not the product path): per-WORKGROUP timeline of the flat expert launch (umoe_moe_flat.hip) of the LAST layer of a
decode step.  Uses the instrumented build (make tl: the product kernel carries no stamp).  Stamps (us after the earliest entry):
  0 entry | 1 first weight chunk requested | 2 rider flags seen | 3 rows staged | 4 gate/up stream end (this wave) | 5 all waves |
  6 published | per down slice s (7 + 4 s ..): producers' flags seen, rows staged, stream end, all waves | 15 exit
Prints quantiles over the workgroups and the plan class of the slowest ones; writes gpurun_out/flat_timeline.json."""
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
from unimoe_audio_amd.config import UniMoEAudioConfig
from unimoe_audio_amd.model import UniAudioRVQQwen2_5VLMoEForConditionalGeneration

LAYERS = int(os.environ.get("TL_LAYERS", "6"))
REPS = int(os.environ.get("TL_REPS", "8"))
dev = torch.device("cuda:0")
L = _lib.lib()
# the instrumented build stamps EVERY kernel class into one buffer (scripts/timeline_wgs.py): it must exist even though only the
# flat launch's own stamps are read here
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
L.umoe_moe_flat_stamps.argtypes = [C.c_void_p]
assert L.umoe_moe_flat_stamps(None) == 0          # enable the stamps (allocates: must happen outside the graph capture)
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
for _ in range(6):
    eng.step(True)
plan = (C.c_double * (3 + 9 * 256))()
L.umoe_moe_flat_plan_probe.argtypes = [C.c_int] * 7 + [C.POINTER(C.c_double), C.c_int]
assert L.umoe_moe_flat_plan_probe(256, 2 * B, cfg.hidden_size, cfg.dynamic_intermediate_size, cfg.shared_intermediate_size, 8, 2, plan, len(plan)) == 0
rows = [[int(v) for v in plan[3 + 9 * j: 12 + 9 * j]] for j in range(256)]
buf = np.zeros((256, 16), dtype=np.uint64)
L.umoe_moe_flat_stamps.argtypes = [C.c_void_p]
runs = []
for r in range(REPS):
    eng.step(True)
    torch.cuda.synchronize()
    assert L.umoe_moe_flat_stamps(buf.ctypes.data_as(C.c_void_p)) == 0
    runs.append(buf.copy())
d = runs[-1].astype(np.int64)
t0 = d[:, 0].min()
us = np.where(d > 0, (d - t0) / 100.0, np.nan)
names = ["entry", "w0 requested", "rider flags seen", "rows staged", "gu stream end", "gu all waves", "published",
         "dn0 flags seen", "dn0 staged", "dn0 stream end", "dn0 all waves", "dn1 flags seen", "dn1 staged", "dn1 stream end", "dn1 all waves", "exit"]
def q(v):
    v = np.sort(v[~np.isnan(v)])
    if len(v) == 0:
        return "-"
    return " ".join(f"{v[int(p * (len(v) - 1))]:6.2f}" for p in (0, .1, .5, .9, 1.0)) + f"  (n={len(v)})"
print(f"## flat expert launch, last layer of {LAYERS}, last of {REPS} replays; us after the earliest entry (min p10 p50 p90 max)")
for k, nm in enumerate(names):
    print(f"{k:2d} {nm:18s} {q(us[:, k])}")
print("in-kernel span over replays:", " ".join(f"{(r[:, 15].max() - r[:, 0].min()) / 100.0:.1f}" for r in runs))
ex = us[:, 15]
order = np.argsort(-ex)
print("slowest workgroups: wg exit | pairs rider | slices (group, first, blocks) | gu published, dn0 flags seen, dn0 staged")
for j in order[:12]:
    r = rows[j]
    print(f"  {j:3d} {ex[j]:6.2f} | {r[1]} {r[2]} | {r[3:6]} {r[6:9]} | {us[j, 6]:.2f} {us[j, 7]:.2f} {us[j, 8]:.2f}")
print("fastest:")
for j in order[-6:]:
    r = rows[j]
    print(f"  {j:3d} {ex[j]:6.2f} | {r[1]} {r[2]} | {r[3:6]} {r[6:9]} | {us[j, 6]:.2f} {us[j, 7]:.2f} {us[j, 8]:.2f}")
# bytes per workgroup vs time
kib = np.array([r[1] * 128 + sum(r[5 + 3 * k] * (43 if r[3 + 3 * k] < 2 else 86) for k in range(2)) for r in rows], dtype=float)
print(f"KiB per workgroup: min {kib.min():.0f} mean {kib.mean():.0f} max {kib.max():.0f}; streaming rate while staged..exit: "
      f"{np.nanmean((kib - 112) * 1024 / ((ex - us[:, 3]) * 1e-6)) / 1e9:.1f} GB/s per CU")
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(dict(us=np.nan_to_num(us).tolist(), plan=rows), open(os.path.join(ROOT, "gpurun_out", "flat_timeline.json"), "w"))
