// Host-side post-processing behind the same C ABI ("next" row 8f-1): greedy temporal NMS per video + re-ranking.
//   reference: temporal_non_maximum_suppression   utils/temporal_nms.py:25-74   (pure-Python list popping)
//              filter_vcmr_by_nms                 baselines/clip_alignment_with_language/inference.py:189-225
//              post_processing_svmr_nms           baselines/clip_alignment_with_language/inference.py:247-265
// This runs on the host CPU on purpose: it is O(200^2) scalar work per query on lists that are already on the
// host (the step after the all-gather); inputs are doubles because the reference operates on Python floats.
#include <algorithm>
#include <numeric>
#include <unordered_map>
#include <vector>

#include "../../include/xmlhip.h"

namespace {

inline double tiou(double s0, double e0, double s1, double e1) {
  const double inter = std::max(0.0, std::min(e0, e1) - std::max(s0, s1));
  const double uni = std::max(e0, e1) - std::min(s0, s1);      // hull, as in the reference
  return uni == 0 ? 0.0 : inter / uni;
}

// indices `idx` (into st/ed/score) -> kept indices, reference order
std::vector<int> nms(const std::vector<int>& idx, const double* st, const double* ed, const double* score, double thd,
                     int max_after) {
  if (idx.size() == 1) return idx;
  std::vector<int> alive(idx);
  std::stable_sort(alive.begin(), alive.end(), [&](int a, int b) { return score[a] > score[b]; });
  std::vector<int> kept;
  while (alive.size() > 1 && (int)kept.size() < max_after) {
    const int head = alive[0];
    std::vector<int> rest;
    rest.reserve(alive.size());
    for (size_t k = 1; k < alive.size(); ++k)
      if (!(tiou(st[head], ed[head], st[alive[k]], ed[alive[k]]) > thd)) rest.push_back(alive[k]);
    kept.push_back(head);
    alive.swap(rest);
  }
  if ((int)kept.size() < max_after && !alive.empty()) kept.push_back(alive[0]);
  return kept;
}

}  // namespace

extern "C" int xml_nms_vcmr_host(const int64_t* vid, const double* st, const double* ed, const double* score, int n,
                                 double thd, int max_before, int max_after, int32_t* out_index, int32_t* n_out) {
  if (!vid || !st || !ed || !score || !out_index || !n_out || n < 0) return XML_ERR_BAD_ARG;
  const int m = std::min(n, max_before);
  std::vector<int64_t> group_order;
  std::unordered_map<int64_t, std::vector<int>> groups;
  for (int i = 0; i < m; ++i) {
    auto it = groups.find(vid[i]);
    if (it == groups.end()) {
      group_order.push_back(vid[i]);
      groups[vid[i]] = {i};
    } else {
      it->second.push_back(i);
    }
  }
  std::vector<int> merged;
  for (int64_t v : group_order) {
    const std::vector<int> k = nms(groups[v], st, ed, score, thd, 100);   // per-video cap = the reference's default
    merged.insert(merged.end(), k.begin(), k.end());
  }
  std::stable_sort(merged.begin(), merged.end(), [&](int a, int b) { return score[a] > score[b]; });
  const int k = std::min<int>((int)merged.size(), max_after);
  for (int i = 0; i < k; ++i) out_index[i] = merged[i];
  *n_out = k;
  return XML_OK;
}

extern "C" int xml_nms_svmr_host(const double* st, const double* ed, const double* score, int n, double thd,
                                 int max_before, int max_after, int32_t* out_index, int32_t* n_out) {
  if (!st || !ed || !score || !out_index || !n_out || n < 0) return XML_ERR_BAD_ARG;
  const int m = std::min(n, max_before);
  std::vector<int> idx(m);
  std::iota(idx.begin(), idx.end(), 0);
  std::vector<int> kept = m ? nms(idx, st, ed, score, thd, 100) : idx;
  const int k = std::min<int>((int)kept.size(), max_after);
  for (int i = 0; i < k; ++i) out_index[i] = kept[i];
  *n_out = k;
  return XML_OK;
}
