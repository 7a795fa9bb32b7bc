"""Two (or more) ranks on ONE GPU, gloo between them, real HIP kernels: started by tests/test_gpu_dist.py through
tvretrieval_amd.launch.spawn_local_ranks.  Every rank encodes its shard of a small corpus; the sharded VCMR pass (both
merge schemes, with and without query chunks) must equal the single-process search over the whole corpus bit for bit,
and the data-parallel gradient average must equal the mean of the per-rank gradients."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
from conftest import load_golden  # noqa: E402


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tvretrieval_amd import dist as xd
    from tvretrieval_amd import inference as inf
    from tvretrieval_amd.model_xml import XML
    d, cfg, sd = load_golden("xml_video_sub_cross_h128")
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)       # noqa: E731
    vf, vm, sf, sm = T(d["video_feat"]), T(d["video_mask"]), T(d["sub_feat"]), T(d["sub_mask"])
    # a corpus of 4 x the fixture's videos (rolled copies, so that the shards differ) -- enough for every rank
    reps = 4
    vf, sf = torch.cat([vf.roll(i, 0) * (1 + 0.05 * i) for i in range(reps)]), torch.cat([sf.roll(i, 0) for i in range(reps)])
    vm, sm = torch.cat([vm.roll(i, 0) for i in range(reps)]), torch.cat([sm.roll(i, 0) for i in range(reps)])
    nv = vf.shape[0]
    qf, qm = T(d["query_feat"]), T(d["query_mask"])
    for dtype in (torch.float32, torch.bfloat16):
        m = XML(cfg, compute_dtype=dtype)
        m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
        m = m.to(dev).eval()
        lo, hi = xd.shard_range(nv, rank, world)
        with torch.no_grad():
            full = inf.build_corpus_index(m, [(vf, vm, sf, sm)])
            want = inf.vcmr_search(m, full, qf, qm, max_vcmr_video=6, max_before_nms=50)
            shard = inf.build_corpus_index(m, [(vf[lo:hi], vm[lo:hi], sf[lo:hi], sm[lo:hi])], video_offset=lo, n_total=nv,
                                           l_ref=full.l_ref)
            got = xd.sharded_vcmr_search(m, shard, qf, qm, max_vcmr_video=6, max_before_nms=50)          # video-owner rerank
            xd.replicate_rerank_features(shard)
            got2 = xd.sharded_vcmr_search(m, shard, qf, qm, max_vcmr_video=6, max_before_nms=50)         # query-owner rerank
            got3 = xd.sharded_vcmr_search(m, shard, qf, qm, max_vcmr_video=6, max_before_nms=50, n_chunks=2)
            own = xd.sharded_vcmr_search(m, shard, qf, qm, max_vcmr_video=6, max_before_nms=50, gather_results=False)
        torch.cuda.synchronize()
        for name, g in (("video-owner", got), ("query-owner", got2), ("chunked", got3)):
            for k in ("top_scores", "top_indices", "flat_scores", "flat_indices"):
                assert torch.equal(g[k], want[k]), (rank, str(dtype), name, k)
        idx = own["query_index"]
        for k in ("top_scores", "top_indices", "flat_scores", "flat_indices"):
            assert torch.equal(own[k], want[k][idx.to(want[k].device)]), (rank, str(dtype), "owner slice", k)
    # exact-rank mode, sharded: every shard runs the bf16 filter / f32 re-score / certificate chain on its own videos and
    # hands its EXACT local top-k to the owner's merge -- the lists are the plain f32 single-process lists (ties at f32
    # rounding aside: here, with a handful of candidates per shard, bit for bit except the video scores' last bits)
    # ... in both exact-rank pipelines: the f32 model (bf16 filter, exact-f32 re-score) and the ops.F16S model (split-f16
    # re-score and ConvSE, second tier on the device; feat2 replicas are SplitRows)
    from tvretrieval_amd import ops as hops
    m32 = XML(cfg, compute_dtype=torch.float32)
    m32.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    m32 = m32.to(dev).eval()
    with torch.no_grad():
        full = inf.build_corpus_index(m32, [(vf, vm, sf, sm)])
        want = inf.vcmr_search(m32, full, qf, qm, max_vcmr_video=6, max_before_nms=50)
    for xdt in (torch.float32, hops.F16S):
        m = XML(cfg, compute_dtype=xdt)
        m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
        m = m.to(dev).eval()
        lo, hi = xd.shard_range(nv, rank, world)
        with torch.no_grad():
            shard = inf.build_corpus_index(m, [(vf[lo:hi], vm[lo:hi], sf[lo:hi], sm[lo:hi])], video_offset=lo, n_total=nv,
                                           l_ref=full.l_ref, exact_filter=True)
            assert shard.exact.mode == ("f16s" if xdt is hops.F16S else "f32")
            shard.exact.n_candidates = 8                    # fewer candidates than local videos: the filter really filters
            assert shard.n_videos > 8
            got = xd.sharded_vcmr_search(m, shard, qf, qm, max_vcmr_video=6, max_before_nms=50)          # video-owner rerank
            xd.replicate_rerank_features(shard)
            got2 = xd.sharded_vcmr_search(m, shard, qf, qm, max_vcmr_video=6, max_before_nms=50)         # query-owner rerank
        torch.cuda.synchronize()
        for name, g in (("exact video-owner", got), ("exact query-owner", got2)):
            if xdt is hops.F16S:
                # split-f16 scores are f32-GRADE, not the f32 MFMA's bits: identical lists up to groups of scores tied to
                # f32 rounding (the rule of tests/test_gpu_split16.py), and the two merge schemes agree bit for bit
                from oracle.listcmp import moment_keys, tie_aware_equal
                gi, wi = g["top_indices"].cpu().numpy(), want["top_indices"].cpu().numpy()
                tie_aware_equal(gi, g["top_scores"].cpu().numpy(), wi, want["top_scores"].cpu().numpy(), gi.shape[1] - 1, 2e-5,
                                name + " videos")
                same = np.nonzero((gi == wi).all(1))[0]
                assert len(same) >= gi.shape[0] - 1, (rank, name)
                l_ = full.l_ref
                tie_aware_equal(moment_keys(g["flat_indices"].cpu().numpy(), gi, l_)[same], g["flat_scores"].cpu().numpy()[same],
                                moment_keys(want["flat_indices"].cpu().numpy(), wi, l_)[same],
                                want["flat_scores"].cpu().numpy()[same], 40, 5e-5, name + " moments")
                for k in ("top_indices", "top_scores", "flat_indices", "flat_scores"):
                    assert torch.equal(g[k], got[k]), (rank, name, k)
                continue
            assert torch.equal(g["top_indices"], want["top_indices"]), (rank, str(xdt), name, "top_indices")
            assert torch.equal(g["flat_indices"], want["flat_indices"]), (rank, str(xdt), name, "flat_indices")
            assert torch.allclose(g["top_scores"], want["top_scores"], rtol=2e-5, atol=0), (rank, str(xdt), name)
            assert torch.allclose(g["flat_scores"], want["flat_scores"], rtol=5e-5, atol=0), (rank, str(xdt), name)
    # data-parallel gradient average over gloo (flat buffer on the GPU)
    from tvretrieval_amd.train import allreduce_gradients

    class Holder(object):
        pass
    h = Holder()
    g = torch.Generator(device=dev).manual_seed(100 + rank)
    h.flat_g = torch.randn(1 << 18, device=dev, generator=g)
    mine = h.flat_g.clone()
    parts = [torch.empty_like(mine.cpu()) for _ in range(world)]
    dist.all_gather(parts, mine.cpu())
    allreduce_gradients(h, bucket_bytes=1 << 18)
    torch.cuda.synchronize()
    assert torch.allclose(h.flat_g.cpu(), torch.stack(parts).mean(0), atol=1e-6)
    dist.barrier()
    if rank == 0:
        print("TWO_RANKS_ONE_GPU_OK world=%d" % world)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
