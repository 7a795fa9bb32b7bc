"""Time one XML training step (BASELINE.json configs[4]: batch 128 video+sub, bf16) on the HIP kernels.

    python tools/bench_train.py [--bsz 128] [--ctx-l 100] [--hidden 768] [--dtype bf16] [--steps 10]
    python tools/bench_train.py --gpus N        (data parallel, RCCL all-reduce: starts the N ranks itself)
    python -m torch.distributed.run --nproc-per-node N ... tools/bench_train.py --gpus N     (same job under torchrun)

Synthetic features of TVR shape (video 3072-d, subtitle / query 768-d), random-init weights.  Prints one JSON line:
ms per step split into forward / backward / all-reduce / optimizer, and pairs (query + video) per second.
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def train_flops_per_sample(lc, lq, hidden, dv, ds, dq, cross=True):
    """Algorithmic flops of ONE (query, video+sub) training sample, forward + backward = 3 x forward:
    context branch 2 L H (Dv + Ds) + 44 L H^2 + 24 L^2 H (two input projections, 2 x (encoder1 + cross
    attention + encoder2)), query branch 2 Lq H Dq + 8 Lq H^2 + 4 Lq^2 H (+ the two H x H query linears), and the in-batch
    similarity / span contractions (negligible: listed for completeness by the caller)."""
    ctx = 2.0 * lc * hidden * (dv + ds) + 44.0 * lc * hidden ** 2 + 24.0 * lc ** 2 * hidden
    qry = 2.0 * lq * hidden * dq + 8.0 * lq * hidden ** 2 + 4.0 * lq ** 2 * hidden + 4.0 * hidden ** 2
    return 3.0 * (ctx + qry)


def run(bsz=128, ctx_l=100, desc_l=30, hidden=768, dv=3072, ds=768, dtype="bf16", steps=10, warmup=3, rank=0, world=1,
        dev=None, graph=False):
    """Times `steps` training steps (forward, backward, [all-reduce], BertAdam) on this rank; returns the result dict.
    graph=True (one GPU): the whole iteration replayed as ONE HIP graph (train.GraphedTrainStep)."""
    import torch.distributed as dist
    from tvretrieval_amd.model_xml import XML, xml_base_config
    from tvretrieval_amd.train import BertAdam, GradientReducer, allreduce_gradients, xml_forward_train
    dev = dev or torch.device("cuda", torch.cuda.current_device())
    cfg = dict(xml_base_config)
    cfg.update(visual_input_size=dv, sub_input_size=ds, query_input_size=ds, hidden_size=hidden,
               max_ctx_l=ctx_l, max_desc_l=desc_l, lw_st_ed=0.01)
    dt = torch.bfloat16 if dtype == "bf16" else torch.float32
    torch.manual_seed(1234)
    model = XML(cfg, compute_dtype=dt).to(dev)
    named = list(model.named_parameters())
    no_decay = ["bias", "LayerNorm.bias", "LayerNorm.weight"]
    opt = BertAdam([{"params": [p for n, p in named if not any(nd in n for nd in no_decay)], "weight_decay": 0.01},
                    {"params": [p for n, p in named if any(nd in n for nd in no_decay)], "weight_decay": 0.0}],
                   lr=1e-4, warmup=0.01, t_total=10000)
    if world > 1:      # gradient buckets are all-reduced under backward (xml_rccl_allreduce_avg_f32 on a side stream)
        GradientReducer(opt)
    g = torch.Generator().manual_seed(77 + rank)
    n, lc, lq = bsz, ctx_l, desc_l
    lens = torch.randint(lc // 2, lc + 1, (n,), generator=g)
    lens[0] = lc
    qlens = torch.randint(5, lq + 1, (n,), generator=g)
    qlens[0] = lq
    mk = lambda ls, l: (torch.arange(l)[None] < ls[:, None]).float()                 # noqa: E731

    def feats(l, d, m):
        x = torch.nn.functional.normalize(torch.randn(n, l, d, generator=g), dim=-1) * m[:, :, None]
        return x.to(dev)
    vm, qm = mk(lens, lc), mk(qlens, lq)
    st = torch.stack([torch.randint(0, int(x), (1,), generator=g)[0] for x in lens])
    ed = torch.stack([torch.randint(int(s), int(x), (1,), generator=g)[0] for s, x in zip(st, lens)])
    batch = dict(query_feat=feats(lq, ds, qm), query_mask=qm.to(dev), video_feat=feats(lc, dv, vm),
                 video_mask=vm.to(dev), sub_feat=feats(lc, ds, vm), sub_mask=vm.to(dev),
                 st_ed_indices=torch.stack([st, ed], 1).to(dev))

    if graph:
        from tvretrieval_amd.train import GraphedTrainStep
        step = GraphedTrainStep(model, opt, batch)      # (world > 1: the reducer's bucketed all-reduces are captured too)
        for _ in range(warmup):
            step(None)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        t0 = time.perf_counter()
        losses = []
        for _ in range(steps):
            loss, parts = step(None)
            if os.environ.get("XML_TRAIN_NO_LOSS_SYNC"):      # A/B only: what the per-step host round trip costs
                losses.append(loss)
            else:
                losses.append(float(loss))      # the reference logs the loss every step: one synchronisation per step
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / steps * 1e3
        losses = [float(v) for v in losses]
        n_param = sum(p.numel() for p in model.parameters())
        flops = train_flops_per_sample(ctx_l, desc_l, hidden, dv, ds, ds) * bsz
        peak = 2500.0 if dtype == "bf16" else 157.3
        tflops = flops / (wall * 1e-3) / 1e12
        return dict(metric="xml_train_step", ms_per_step=round(wall, 3), pairs_per_s=round(bsz * world / wall * 1e3, 1),
                    n_gpus=world,
                    dtype=dtype, mode="one HIP graph per step (train.GraphedTrainStep)", flops_per_step=flops,
                    tflops=round(tflops, 1), frac_of_mfma_peak=round(tflops / peak, 4),
                    config=dict(bsz_per_gpu=bsz, ctx_l=ctx_l, desc_l=desc_l, hidden=hidden, dv=dv, params=n_param),
                    loss_first=round(losses[0], 4), loss_last=round(losses[-1], 4))
    ev = lambda: torch.cuda.Event(enable_timing=True)      # noqa: E731
    acc = dict(fwd=0.0, bwd=0.0, allreduce=0.0, optim=0.0)
    losses = []
    wall0 = None
    for it in range(warmup + steps):
        if it == warmup:
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            wall0 = time.perf_counter()
        e = [ev() for _ in range(5)]
        e[0].record()
        loss, parts = xml_forward_train(model, **batch)
        e[1].record()
        opt.zero_grad()
        loss.backward()
        e[2].record()
        allreduce_gradients(opt)
        e[3].record()
        opt.step()
        e[4].record()
        if it >= warmup:
            torch.cuda.synchronize()
            for k, i in (("fwd", 0), ("bwd", 1), ("allreduce", 2), ("optim", 3)):
                acc[k] += e[i].elapsed_time(e[i + 1])
            losses.append(parts["loss_overall"])
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    wall = (time.perf_counter() - wall0) / steps * 1e3
    n_param = sum(p.numel() for p in model.parameters())
    flops = train_flops_per_sample(lc, lq, hidden, dv, ds, ds) * bsz
    peak = 2500.0 if dtype == "bf16" else 157.3
    tflops = flops / (wall * 1e-3) / 1e12
    return dict(metric="xml_train_step", ms_per_step=round(wall, 3),
                pairs_per_s=round(bsz * world / wall * 1e3, 1), n_gpus=world, dtype=dtype,
                breakdown_ms={k: round(v / steps, 3) for k, v in acc.items()},
                flops_per_step=flops, tflops=round(tflops, 1), frac_of_mfma_peak=round(tflops / peak, 4),
                config=dict(bsz_per_gpu=bsz, ctx_l=lc, desc_l=lq, hidden=hidden, dv=dv, params=n_param),
                loss_first=round(losses[0], 4), loss_last=round(losses[-1], 4))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--bsz", type=int, default=128)
    ap.add_argument("--ctx-l", type=int, default=100)
    ap.add_argument("--desc-l", type=int, default=30)
    ap.add_argument("--hidden", type=int, default=768)
    ap.add_argument("--dv", type=int, default=3072)
    ap.add_argument("--ds", type=int, default=768)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--graph", action="store_true", help="replay the whole step as one HIP graph (one GPU)")
    ap.add_argument("--separate-dropout", action="store_true", help="A/B: dropout sites as their own launches (train.FUSE_DROPOUT = False)")
    ap.add_argument("--no-shadows", action="store_true", help="A/B: convert / transpose every weight at its use (train.SHADOW_WEIGHTS = False)")
    ap.add_argument("--separate-loss-tail", action="store_true", help="A/B: autograd.FUSED_LOSS_TAIL = False")
    ap.add_argument("--separate-kv", action="store_true", help="A/B: train.FUSE_KV = False")
    ap.add_argument("--no-res-qkv", action="store_true", help="A/B: train.RESIDUAL_THROUGH_QKV = False (autograd sums a block input's two gradients itself)")
    ap.add_argument("--query-first", action="store_true", help="A/B: train.QUERY_FIRST = True (query encoder in front of the context branches)")
    ap.add_argument("--one-stream", action="store_true", help="A/B: train.PARALLEL_BRANCHES = False (video and subtitle branches on one stream)")
    a = ap.parse_args()
    from tvretrieval_amd import launch
    if a.separate_loss_tail:
        import tvretrieval_amd.autograd as AG
        AG.FUSED_LOSS_TAIL = False
    if a.separate_kv:
        import tvretrieval_amd.train as TR
        TR.FUSE_KV = False
    if a.one_stream:
        import tvretrieval_amd.train as TR
        TR.PARALLEL_BRANCHES = False
    if a.no_res_qkv:
        import tvretrieval_amd.train as TR
        TR.RESIDUAL_THROUGH_QKV = False
    if a.query_first:
        import tvretrieval_amd.train as TR
        TR.QUERY_FIRST = True
    if a.no_shadows:
        import tvretrieval_amd.train as TR
        TR.SHADOW_WEIGHTS = False
    if a.separate_dropout:
        import tvretrieval_amd.train as TR
        TR.FUSE_DROPOUT = False
    if a.gpus > 1 and not launch.under_launcher():
        rc = launch.spawn_local_ranks(os.path.abspath(__file__), sys.argv[1:], a.gpus)
        sys.exit(rc)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)
    res = run(a.bsz, a.ctx_l, a.desc_l, a.hidden, a.dv, a.ds, a.dtype, a.steps, a.warmup, rank, world, dev, graph=a.graph)
    if rank == 0:
        print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
