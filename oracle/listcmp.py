"""TEST INFRASTRUCTURE (like everything under oracle/): tie-aware comparison of ranked lists.

torch.topk / torch.sort leave the order of equal scores unspecified (xml/inference.py:347,381 rely on them), and a
score computed by two correct fp32 implementations differs in the last bits.  Two ranked lists are therefore "the same"
when they agree position by position except inside groups of scores that are tied to within rounding."""
import numpy as np


def tie_aware_equal(got_keys, got_scores, want_keys, want_scores, k, rtol, what="list"):
    """got_*: (rows, >= k); want_*: (rows, >= k) -- pass a few entries MORE than k on the want side so that a swap across
    the list boundary can be recognised.  got[:k] must equal want[:k] position by position, except where the entry `got`
    placed there is found in want's list with a score within rtol of the score want has at that position, and got's own
    score agrees with it.  Returns the number of positions that differed (all of them justified ties); raises
    AssertionError on the first unjustified difference."""
    got_keys, want_keys = np.asarray(got_keys), np.asarray(want_keys)
    got_scores, want_scores = np.asarray(got_scores), np.asarray(want_scores)
    n_diff = 0
    for q in range(len(got_keys)):
        g, w = got_keys[q][:k], want_keys[q]
        bad = np.nonzero(g != w[:k])[0]
        n_diff += len(bad)
        for i in bad:
            j = np.nonzero(w == g[i])[0]
            assert len(j) == 1, (what, "entry not in the reference list at all", q, int(i), int(g[i]))
            ref_here = want_scores[q][i]
            assert abs(want_scores[q][j[0]] - ref_here) <= rtol * abs(ref_here), \
                (what, "order differs beyond rounding", q, int(i), float(want_scores[q][j[0]]), float(ref_here))
            assert abs(got_scores[q][i] - want_scores[q][j[0]]) <= rtol * abs(ref_here), (what, "score", q, int(i))
    return n_diff


def moment_keys(flat, top_indices, l_ref):
    """(video id, st, ed) packed into one integer per moment, decoded through the list's OWN top-video order
    (flat = (r * l_ref + st) * l_ref + ed, xml/inference.py:423-431); -1 stays -1."""
    flat = np.asarray(flat).astype(np.int64)
    top = np.asarray(top_indices).astype(np.int64)
    ll = l_ref * l_ref
    ok = flat >= 0
    r = np.where(ok, flat // ll, 0)
    vid = np.take_along_axis(top, np.clip(r, 0, top.shape[1] - 1), 1)
    return np.where(ok, vid * ll + flat % ll, -1)
