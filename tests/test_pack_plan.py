"""Host-side layout logic of the length-bucketed K6 corpus image (tvretrieval_amd.ops.PackPlan): pure index arithmetic,
runs on CPU tensors."""
import numpy as np
import torch

from tvretrieval_amd import ops


def _masks(lens):
    return (torch.arange(128)[None] < torch.as_tensor(lens)[:, None]).float()


def test_pack_plan_layout():
    rng = np.random.default_rng(0)
    lens = np.concatenate([rng.integers(1, 33, 13), rng.integers(33, 65, 57), rng.integers(65, 129, 9), [32, 64, 128, 1, 33, 65]])
    rng.shuffle(lens)
    nv = len(lens)
    m_video = _masks(lens)
    m_sub = _masks(np.maximum(lens - rng.integers(0, 3, nv), 1))      # a modality that is sometimes shorter
    plan = ops.PackPlan([m_video, m_sub])
    ids = plan.slot_ids.numpy()
    rm = plan.row_map.numpy()
    assert rm.shape[0] == plan.n_tiles * 256 and ids.shape == (2 * plan.n_tiles, 4)
    # every video exactly once; bucket by padded length; per-bucket ascending ids
    flat = ids[ids >= 0]
    assert sorted(flat.tolist()) == list(range(nv))
    w128, w64 = 2 * plan.ct128, 2 * plan.ct64
    assert (ids[:w128, 1:] == -1).all() and (ids[w128:w64, 2:] == -1).all()
    for rng_w, lo, hi in ((slice(0, w128), 64, 128), (slice(w128, w64), 32, 64), (slice(w64, None), 0, 32)):
        v = ids[rng_w][ids[rng_w] >= 0]
        assert ((lens[v] > lo) & (lens[v] <= hi)).all()
        assert (np.diff(v) > 0).all()
    # row_map: wave tile w, sub-slot j of padded length lp covers columns [j*lp, (j+1)*lp) = clips 0..lp-1 of its video
    for w in range(2 * plan.n_tiles):
        lp = 128 if w < w128 else 64 if w < w64 else 32
        for j in range(128 // lp):
            seg = rm[w * 128 + j * lp: w * 128 + (j + 1) * lp]
            v = ids[w, j]
            assert (seg == (-1 if v < 0 else v * 128 + np.arange(lp))).all()
    assert plan.padded_clips == int(sum(128 if x > 64 else 64 if x > 32 else 32 for x in lens))
    # packed mask bits = the masks of the packed columns
    for m in (m_video, m_sub):
        bits = plan.mask_bits(m).numpy().astype(np.int64) & 0xffffffff
        cols = np.where(rm >= 0, m.reshape(-1).numpy()[np.maximum(rm, 0)], 0).reshape(-1, 4, 32)
        want = (cols.astype(np.int64) << np.arange(32)).sum(-1)
        assert (bits == want).all()


def test_pack_plan_not_used_for_full_length_or_soft_masks():
    full = torch.ones(6, 128)
    assert ops.q2c_pack_plan([full, full]) is None
    soft = _masks([40, 50, 60, 70])
    soft[0, 3] = 0.5
    assert ops.q2c_pack_plan([soft]) is None
    long_only = _masks([100, 128, 90, 70])
    assert ops.q2c_pack_plan([long_only]) is None          # nothing to gain: every video needs the 128 bucket
    assert ops.q2c_pack_plan([_masks([100, 20, 90, 70])]) is not None
