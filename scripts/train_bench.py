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


k_real = [0.0]


def step():
    for p in model.parameters():
        p.grad = None
    loss, closs, aux, routing = TR.forward_train(model, ids, codec, am, labels, return_routing=True)
    loss.backward()
    if k_real[0] == 0.0:      # logged routing (first step only): routed REAL experts per token (no null / shared experts), mean over layers
        k_real[0] = float(torch.stack([m[1][..., : cfg.mlp_dynamic_expert_num].float().sum(-1).mean() for m in routing]).mean())
    return float(loss.detach())


def step_flops(k):
    """Algorithmic FLOPs of one forward + backward (3 x forward; the attention recompute of the flash backward is not counted):
    per token and layer QKV + o_proj + causal attention (T/2 keys on average) + gate + k routed experts + the shared experts, + codec head."""
    D, H, KV, hd = cfg.hidden_size, cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim
    per_tok_layer = 2.0 * D * (H * hd + 2 * KV * hd) + 2.0 * H * hd * D + 4.0 * H * hd * (T / 2.0) + 2.0 * D * cfg.num_experts \
        + k * 6.0 * cfg.dynamic_intermediate_size * D + cfg.mlp_fixed_expert_num * 6.0 * cfg.shared_intermediate_size * D
    head = 2.0 * D * cfg.codec_channels * cfg.codec_vocab_size
    return 3.0 * B * T * (layers * per_tok_layer + head)


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
# a trainer steps the optimizer after every backward: the same steps with every parameter updated in place (untimed) in between --
# nothing is kept across steps that depends on the weights (round 2 kept transposed weight copies; the input gradients read the weights
# as stored now), so this is the same number
tn = []
for _ in range(int(os.environ.get("TB_NOCACHE_STEPS", "3"))):
    with torch.no_grad():
        for p in model.parameters():
            p.add_(0.0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    step()
    torch.cuda.synchronize()
    tn.append(time.perf_counter() - t0)
dtn = sorted(tn)[len(tn) // 2] if tn else None
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
                  "step_ms_weights_change_every_step": round(dtn * 1e3, 1) if dtn else None,
                  "forward_only_ms": round(min(tf) * 1e3, 1), "loss": round(l0, 4),
                  "max_mem_GiB": round(torch.cuda.max_memory_allocated() / 2 ** 30, 1),
                  "k_real": round(k_real[0], 3), "step_tflop": round(step_flops(k_real[0]) / 1e12, 2),
                  "roofline": {"bound": "mfma", "achieved": round(step_flops(k_real[0]) / dt / 1e12, 1), "peak": 2500.0, "unit": "TFLOP/s",
                               "frac": round(step_flops(k_real[0]) / dt / 2.5e15, 4),
                               "note": "whole step: algorithmic FLOPs (3 x forward, logged k_real routed experts per token) / step time; "
                                       "peak = dense bf16 MFMA (MI355X_MICROARCH.md)"},
                  "note": "layer math only, no optimizer step (SURVEY 8d config 3); step_ms_weights_change_every_step = the same step after an "
                          "in-place update of every parameter (nothing weight-dependent is cached across steps)"}))
