// Host-side post-processing behind the same C ABI ("next" row 8f-1): greedy temporal NMS per video + re-ranking.
//   reference: temporal_non_maximum_suppression   utils/temporal_nms.py:25-74   (pure-Python list popping)
//              filter_vcmr_by_nms                 baselines/clip_alignment_with_language/inference.py:189-225
//              post_processing_svmr_nms           baselines/clip_alignment_with_language/inference.py:247-265
// This runs on the host CPU on purpose: it is O(200^2) scalar work per query on lists that are already on the
// host (the step after the all-gather); inputs are doubles because the reference operates on Python floats.
// The *_batched entries take the (Nq, n) arrays the engine's K10 epilogue produces (one D2H per batch) and spread the
// queries over host threads; the per-query entries are the same core on one row.
#include <algorithm>
#include <atomic>
#include <numeric>
#include <thread>
#include <vector>

#include "../../include/xmlhip.h"

namespace {

inline double tiou(double s0, double e0, double s1, double e1) {
  const double inter = std::max(0.0, std::min(e0, e1) - std::max(s0, s1));
  const double uni = std::max(e0, e1) - std::min(s0, s1);      // hull, as in the reference
  return uni == 0 ? 0.0 : inter / uni;
}

struct Scratch {
  std::vector<int> order;                                             // nms_core
  std::vector<char> dead;
  std::vector<int> by_vid, group_first, members, merged, kept, idx;   // the row drivers
};

// temporal_non_maximum_suppression on the entries idx[0..m) of (st, ed, score): appends the kept entries to `kept` in the
// reference's order.  The reference pops suppressed entries from three parallel lists while it walks them; a dead flag
// visits the same pairs in the same order (an entry suppressed by an earlier head is never compared again there either).
void nms_core(const int* idx, int m, const double* st, const double* ed, const double* score, double thd, int max_after,
              Scratch& s, std::vector<int>& kept) {
  if (m == 1) {                    // "only has one prediction, no need for nms" (also skips the cap, like the reference)
    kept.push_back(idx[0]);
    return;
  }
  s.order.assign(idx, idx + m);
  std::stable_sort(s.order.begin(), s.order.end(), [&](int a, int b) { return score[a] > score[b]; });
  s.dead.assign(m, 0);
  int n_kept = 0;
  for (int h = 0; h < m && n_kept < max_after; ++h) {
    if (s.dead[h]) continue;
    const int head = s.order[h];
    const double hs = st[head], he = ed[head];
    for (int k = h + 1; k < m; ++k)
      if (!s.dead[k] && tiou(hs, he, st[s.order[k]], ed[s.order[k]]) > thd) s.dead[k] = 1;
    kept.push_back(head);
    ++n_kept;
  }
}

// filter_vcmr_by_nms on one row: returns the number of kept entries written to out_index (indices into the row)
int vcmr_row(const int64_t* vid, const double* st, const double* ed, const double* score, int n, double thd, int max_before,
             int max_after, int32_t* out_index, Scratch& s) {
  const int m = std::min(n, max_before);
  if (m <= 0) return 0;
  // group by video in order of first appearance (a dict in the reference): sort positions by (vid, position), then order the
  // groups by their first position
  std::vector<int>& by_vid = s.by_vid;
  by_vid.resize(m);
  std::iota(by_vid.begin(), by_vid.end(), 0);
  std::stable_sort(by_vid.begin(), by_vid.end(), [&](int a, int b) { return vid[a] < vid[b]; });
  // group boundaries in by_vid; groups ordered by their first member (= smallest position, stable sort)
  s.group_first.clear();
  for (int i = 0; i < m; ++i)
    if (i == 0 || vid[by_vid[i]] != vid[by_vid[i - 1]]) s.group_first.push_back(i);
  std::vector<int>& groups = s.group_first;
  std::sort(groups.begin(), groups.end(), [&](int a, int b) { return by_vid[a] < by_vid[b]; });
  s.merged.clear();
  for (int g : groups) {
    int e = g + 1;
    while (e < m && vid[by_vid[e]] == vid[by_vid[g]]) ++e;
    s.members.assign(by_vid.begin() + g, by_vid.begin() + e);
    nms_core(s.members.data(), e - g, st, ed, score, thd, 100, s, s.merged);   // per-video cap = the reference's default
  }
  std::stable_sort(s.merged.begin(), s.merged.end(), [&](int a, int b) { return score[a] > score[b]; });
  const int k = std::min<int>((int)s.merged.size(), max_after);
  for (int i = 0; i < k; ++i) out_index[i] = s.merged[i];
  return k;
}

int svmr_row(const double* st, const double* ed, const double* score, int n, double thd, int max_before, int max_after,
             int32_t* out_index, Scratch& s) {
  const int m = std::min(n, max_before);
  if (m <= 0) return 0;
  s.idx.resize(m);
  std::iota(s.idx.begin(), s.idx.end(), 0);
  s.kept.clear();
  nms_core(s.idx.data(), m, st, ed, score, thd, 100, s, s.kept);
  const int k = std::min<int>((int)s.kept.size(), max_after);
  for (int i = 0; i < k; ++i) out_index[i] = s.kept[i];
  return k;
}

template <typename F> void parallel_rows(int nq, int n_threads, F&& fn) {
  int t = n_threads > 0 ? n_threads : (int)std::thread::hardware_concurrency();
  t = std::max(1, std::min(t, std::min(nq / 64 + 1, 64)));
  if (t == 1) {
    Scratch s;
    for (int q = 0; q < nq; ++q) fn(q, s);
    return;
  }
  std::atomic<int> next(0);
  std::vector<std::thread> pool;
  for (int i = 0; i < t; ++i)
    pool.emplace_back([&]() {
      Scratch s;
      for (;;) {
        const int q0 = next.fetch_add(32);
        if (q0 >= nq) break;
        for (int q = q0; q < std::min(nq, q0 + 32); ++q) fn(q, s);
      }
    });
  for (auto& th : pool) th.join();
}

}  // namespace

extern "C" int xml_nms_vcmr_host(const int64_t* vid, const double* st, const double* ed, const double* score, int n,
                                 double thd, int max_before, int max_after, int32_t* out_index, int32_t* n_out) {
  if (!vid || !st || !ed || !score || !out_index || !n_out || n < 0) return XML_ERR_BAD_ARG;
  Scratch s;
  *n_out = vcmr_row(vid, st, ed, score, n, thd, max_before, max_after, out_index, s);
  return XML_OK;
}

extern "C" int xml_nms_svmr_host(const double* st, const double* ed, const double* score, int n, double thd,
                                 int max_before, int max_after, int32_t* out_index, int32_t* n_out) {
  if (!st || !ed || !score || !out_index || !n_out || n < 0) return XML_ERR_BAD_ARG;
  Scratch s;
  *n_out = svmr_row(st, ed, score, n, thd, max_before, max_after, out_index, s);
  return XML_OK;
}

extern "C" int xml_nms_vcmr_batched_host(const int64_t* vid, const double* st, const double* ed, const double* score,
                                         const int32_t* count, int nq, int64_t ld, double thd, int max_before,
                                         int max_after, int32_t* out_index, int64_t ld_out, int32_t* out_count,
                                         int n_threads) {
  if (!vid || !st || !ed || !score || !count || !out_index || !out_count || nq < 0 || ld < 0 || max_after < 0 ||
      ld_out < max_after)
    return XML_ERR_BAD_ARG;
  std::atomic<int> bad(0);
  parallel_rows(nq, n_threads, [&](int q, Scratch& s) {
    const int n = count[q];
    if (n < 0 || n > ld) { bad = 1; out_count[q] = 0; return; }
    const int64_t o = (int64_t)q * ld;
    out_count[q] = vcmr_row(vid + o, st + o, ed + o, score + o, n, thd, max_before, max_after, out_index + (int64_t)q * ld_out, s);
  });
  return bad ? XML_ERR_BAD_ARG : XML_OK;
}

extern "C" int xml_nms_svmr_batched_host(const double* st, const double* ed, const double* score, const int32_t* count,
                                         int nq, int64_t ld, double thd, int max_before, int max_after,
                                         int32_t* out_index, int64_t ld_out, int32_t* out_count, int n_threads) {
  if (!st || !ed || !score || !count || !out_index || !out_count || nq < 0 || ld < 0 || max_after < 0 || ld_out < max_after)
    return XML_ERR_BAD_ARG;
  std::atomic<int> bad(0);
  parallel_rows(nq, n_threads, [&](int q, Scratch& s) {
    const int n = count[q];
    if (n < 0 || n > ld) { bad = 1; out_count[q] = 0; return; }
    const int64_t o = (int64_t)q * ld;
    out_count[q] = svmr_row(st + o, ed + o, score + o, n, thd, max_before, max_after, out_index + (int64_t)q * ld_out, s);
  });
  return bad ? XML_ERR_BAD_ARG : XML_OK;
}
