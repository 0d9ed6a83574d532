"""BASELINE configs[2] measurement (not the headline bench): forward + backward of the full 36-layer model + codec head +
shifted per-channel CE + aux loss at batch 4 x 1560 tokens (30 s of 50 Hz codec frames + text), bf16, 1 x MI355X.
Layer math only (no optimizer step), as SURVEY.md 8(d) defines config 3.  Prints one JSON line."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unimoe_audio_amd.config import UniMoEAudioConfig
from unimoe_audio_amd.model import UniAudioRVQQwen2_5VLMoEForConditionalGeneration
from unimoe_audio_amd import train as TR

layers = int(os.environ.get("TB_LAYERS", "36"))
B, T = int(os.environ.get("TB_BATCH", "4")), int(os.environ.get("TB_T", "1560"))
steps = int(os.environ.get("TB_STEPS", "5"))   # (the second step builds the transposed-weight cache: a median of 3 can land on it)
dev = torch.device("cuda:0")
cfg = UniMoEAudioConfig()
cfg.num_hidden_layers = layers
torch.set_default_dtype(torch.bfloat16)
with torch.device(dev):
    model = UniAudioRVQQwen2_5VLMoEForConditionalGeneration(cfg)
torch.set_default_dtype(torch.float32)
model.init_synthetic(1234).train()
for p in model.parameters():
    p.requires_grad_(True)
g = torch.Generator().manual_seed(11)
n_text = 40
ids = torch.randint(0, 151643, (B, T), generator=g)
ids[:, n_text:] = cfg.codec_placeholder_value
codec = torch.randint(0, 1024, (B * (T - n_text), cfg.codec_channels), generator=g)
labels = torch.randint(0, 1024, (B, T, cfg.codec_channels), generator=g)
labels[:, :n_text] = -100
am = torch.ones(B, T, dtype=torch.long)


def step():
    for p in model.parameters():
        p.grad = None
    loss, closs, aux = TR.forward_train(model, ids, codec, am, labels)
    loss.backward()
    return float(loss.detach())


t0 = time.perf_counter()
l0 = step()
torch.cuda.synchronize()
t_first = time.perf_counter() - t0
ts = []
for _ in range(steps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    step()
    torch.cuda.synchronize()
    ts.append(time.perf_counter() - t0)
dt = sorted(ts)[len(ts) // 2]
tf = []
with torch.no_grad():
    for _ in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        model.forward(input_ids=ids, codec_input_ids=codec, attention_mask=am, codec_labels=labels, labels=labels)
        torch.cuda.synchronize()
        tf.append(time.perf_counter() - t0)
print(json.dumps({"workload": f"BASELINE configs[2]: fwd+bwd, {layers} layers, batch {B} x {T} tokens", "tokens_per_step": B * T,
                  "step_ms": round(dt * 1e3, 1), "tokens_per_s": round(B * T / dt, 1), "first_step_ms": round(t_first * 1e3, 1),
                  "all_steps_ms": [round(v * 1e3, 1) for v in ts],
                  "forward_only_ms": round(min(tf) * 1e3, 1), "loss": round(l0, 4),
                  "max_mem_GiB": round(torch.cuda.max_memory_allocated() / 2 ** 30, 1),
                  "weights_unchanged_between_steps": True,
                  "note": "layer math only, no optimizer step (SURVEY 8d config 3): the weights keep their versions, so from the third step on "
                          "the transposed weight copies of the input-gradient GEMMs are reused (ops._WT_CACHE, the gradient-accumulation "
                          "case); UMOE_WT_CACHE=0 rebuilds them every step as a trainer stepping after every backward would"}))
