"""Expert-parallel TRAINING step of one DCMoE block with N processes (gloo control plane; device slabs staged through the host, which is
what ep._a2a does under gloo; a real job runs RCCL) against the ep_size = 1 block holding all experts, in every process:
  forward output of the own rows, gradient of the own rows, gradients of the gate / shared experts (own rows only: the data-parallel
  reduction is the trainer's), gradients of the LOCAL experts' weights = sum over every rank's rows (they arrive through the backward
  of the exchange).
usage: python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 scripts/ep_train_multiproc.py"""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from unimoe_audio_amd.config import UniMoEAudioConfig
    from unimoe_audio_amd.dcmoe import UniMoEAudioSparseMoeBlock
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    ndev = torch.cuda.device_count()
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")) % ndev)
    torch.cuda.set_device(dev)
    dist.init_process_group("gloo")
    S, D, Id, Is = 192, 256, 128, 64
    kw = dict(hidden_size=D, dynamic_intermediate_size=Id, shared_intermediate_size=Is, router_jitter_noise=0.01, input_jitter_noise=0.0)
    torch.manual_seed(7)
    full = UniMoEAudioSparseMoeBlock(UniMoEAudioConfig(ep_size=1, **kw))
    with torch.no_grad():
        for p in full.parameters():
            p.normal_(0, 0.05)
    full = full.to(dev, torch.bfloat16).train()
    part = UniMoEAudioSparseMoeBlock(UniMoEAudioConfig(ep_size=world, **kw)).to(dev, torch.bfloat16).train()
    part.dynamic_real_moe.set_deepspeed_parallelism(ep_group=dist.group.WORLD)
    n_real = full.mlp_dynamic_real_expert_num
    E_loc = n_real // world
    with torch.no_grad():
        part.gate.weight.copy_(full.gate.weight)
        for q in range(E_loc):
            for a, b in zip(part._experts()[q].parameters(), full._experts()[rank * E_loc + q].parameters()):
                a.copy_(b)
        for a, b in zip(part.fixed_real_moe.parameters(), full.fixed_real_moe.parameters()):
            a.copy_(b)
    xs, gs = [], []
    for r in range(world):
        g = torch.Generator().manual_seed(50 + r)
        xs.append(torch.randn(1, S, D, generator=g).to(torch.bfloat16).to(dev))
        gs.append(torch.randn(1, S, D, generator=g).to(torch.bfloat16).to(dev))
    # ---- expert parallel
    for p in part.parameters():
        p.requires_grad_(True)
    x = xs[rank].clone().requires_grad_(True)
    out = part(x, None, None)
    (out[0].float() * gs[rank].float()).sum().backward()
    # ---- reference: the full block on EVERY rank's rows
    ref_out, ref_dx, ref_g = None, None, {}
    dw_sum = [[None, None, None] for _ in range(n_real)]
    for r in range(world):
        for p in full.parameters():
            p.requires_grad_(True)
            p.grad = None
        xr = xs[r].clone().requires_grad_(True)
        o = full(xr, None, None)
        (o[0].float() * gs[r].float()).sum().backward()
        if r == rank:
            ref_out, ref_dx = o[0].detach(), xr.grad.detach()
            assert torch.equal(o[3], out[3]), "routing masks differ"
            ref_g = dict(gate=full.gate.weight.grad.clone(), shared=[p.grad.clone() for p in full.fixed_real_moe.parameters()])
        for e in range(n_real):
            for k, p in enumerate(full._experts()[e].parameters()):
                dw_sum[e][k] = p.grad.float().clone() if dw_sum[e][k] is None else dw_sum[e][k] + p.grad.float()

    def rel(a, b):
        return float((a.float() - b.float()).norm() / (b.float().norm() + 1e-12))
    errs = {"out": rel(out[0], ref_out), "dx": rel(x.grad, ref_dx), "d_gate": rel(part.gate.weight.grad, ref_g["gate"])}
    errs["d_shared"] = max(rel(a.grad, b) for a, b in zip(part.fixed_real_moe.parameters(), ref_g["shared"]))
    errs["d_experts"] = max(rel(p.grad, dw_sum[rank * E_loc + q][k]) for q in range(E_loc) for k, p in enumerate(part._experts()[q].parameters()))
    ok = errs["out"] < 1e-6 and errs["dx"] < 2e-2 and errs["d_gate"] < 2e-2 and errs["d_shared"] < 1e-6 and errs["d_experts"] < 2e-2
    t = torch.tensor([1.0 if ok else 0.0])
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    print(f"rank {rank}: " + " ".join(f"{k}={v:.2e}" for k, v in errs.items()), flush=True)
    if rank == 0:
        print("EP-TRAIN-OK" if float(t) == 1.0 else "EP-TRAIN-MISMATCH", flush=True)
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if float(t) == 1.0 else 1)


if __name__ == "__main__":
    main()
