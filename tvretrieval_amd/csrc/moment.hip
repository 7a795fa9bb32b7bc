// K9 + K10 (+ K11): banded moment candidates and per-query top-n.
//   reference: einsum("qvm,qv,qvn->qvmn") * min/max-length mask, flat descending sort, [:max_before_nms]
//              xml/inference.py:365-386, generate_min_max_length_mask :170-192, index decode :423-431
//              SVMR variant: get_svmr_res_from_st_ed_probs :195-241 + utils/tensor_utils.py:133-141
//
// The reference materialises (Nq, 100, L, L) products and fully sorts 1.64 M values per query.  Only the band
// min_l <= j-i < max_l can be non-zero, so one workgroup per query works on the k*L*(max_l-min_l) band candidates
// from register-resident rows and keeps the best n_out, exactly, without ever sorting more than a few hundred values:
//   1. Wave w owns the ACTIVE pairs (w != 0) w, w+4, ...; lane l owns start clips l and l+64 and holds st*w and the
//      pair's end probabilities for them in registers; (st*w)*ed is the reference's einsum order.
//   2. row maxima m(r,i) = max_d score(r,i,i+d) = (st*w)[i] * max_d ed[i+d] (exact for non-negative inputs: rounding
//      is monotone); the sliding-window maximum of ed is a few wave shuffles.  Nothing of a pair is staged in LDS
//      (24 KiB per workgroup, four per CU); the maxima are simply recomputed in the expansion pass.
//   3. a lower bound lb of the n_out-th best score: ONE histogram pass over the row maxima on bits [30:20] (wave-aggregated
//      LDS adds, wave-parallel suffix scan) -- the lower edge of the bin that holds the n_out-th largest row maximum.
//   4. expand rows with m(r,i) >= lb; candidates >= lb go to an LDS list (wave-aggregated append).
//      If the list overflows, lb is raised to the n_out-th largest of the stored entries (a subset, hence still a
//      valid lower bound) and the expansion repeats; if that separates nothing, to the exact n_out-th largest candidate
//      (radix select over all band candidates); if more candidates than the list holds TIE with it: those above first,
//      then as many ties as fit.
//   5. the list in (score desc, flat index asc) order -- each entry ranked against the list (<= 3 entries per thread),
//      bitonic sort for longer lists -- emit n_out.
// Zero products (masked clips, skipped pairs) are not candidates: the reference's order among zeros is unspecified.
// This is LDS/latency-bound integer-ish work; it is not reshaped into a GEMM.
#include "band.h"

#ifdef XML_DEBUG_VARIANTS
__device__ unsigned long long g_k9_stats[8];      // dbg 63: [0] expansion attempts, [1] live rows of the first attempt, [2] queries
                                                  // with > 1024 live rows, [3] list entries at the end, [4] queries
extern "C" int xml_debug_read_k9_stats(unsigned long long* host_out, int reset) {
  if (hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_k9_stats), sizeof(g_k9_stats)) != hipSuccess) return -4;
  if (reset) {
    unsigned long long z[8] = {};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_k9_stats), z, sizeof(z)) != hipSuccess) return -4;
  }
  return 0;
}
#endif
static constexpr int MT_CAP = 2048;   // LDS candidate list capacity
static constexpr int MT_PPW = 32;     // pairs per wave -> kpairs <= 128 (pair weights live in two lane registers)

struct MomentShared {
  uint32_t prefix, need, cnt, flag;
};

// wave-aggregated histogram increment: one LDS atomic per distinct bin per wave
__device__ __forceinline__ void hist_add(uint32_t* hist, uint32_t bin, bool active, int lane) {
  unsigned long long todo = __ballot(active);
  while (todo) {
    const int leader = __ffsll((long long)todo) - 1;
    const uint32_t lb = __shfl(bin, leader, 64);
    const unsigned long long same = __ballot(active && bin == lb) & todo;
    if (lane == leader) atomicAdd(&hist[lb], (uint32_t)__popcll(same));
    todo &= ~same;
  }
}

// Block-wide radix select of the `need`-th largest non-zero key (MSB first, 11/11/10 bits).
// each(f): calls f(key, valid) the SAME number of times in every lane of a wave (valid = false for padding).
// Returns T (0 when fewer than `need` non-zero keys exist: everything non-zero qualifies).
template <typename Each>
__device__ uint32_t block_radix_select(Each each, uint32_t need, uint32_t* hist /*2048*/, MomentShared* sh) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int shifts[3] = {21, 10, 0};
  const uint32_t widths[3] = {11, 11, 10};
  if (tid == 0) { sh->prefix = 0; sh->need = need; sh->flag = 0; }
  uint32_t mask = 0;
  for (int pass = 0; pass < 3; ++pass) {
    const int shift = shifts[pass];
    const uint32_t bins = 1u << widths[pass];
    for (int i = tid; i < 2048; i += (int)blockDim.x) hist[i] = 0;
    __syncthreads();
    const uint32_t prefix = sh->prefix;
    if (sh->flag) break;                                   // not enough keys: T = 0
    each([&](uint32_t key, bool valid) {
      const bool act = valid && key != 0 && (key & mask) == prefix;
      hist_add(hist, (key >> shift) & (bins - 1), act, lane);
    });
    __syncthreads();
    if (tid < 64) {                                        // wave 0: suffix scan over the bins, 32 bins per lane
      const uint32_t nd = sh->need;
      const int per = (int)bins / 64;
      uint32_t local = 0;
      for (int b = 0; b < per; ++b) local += hist[lane * per + b];
      uint32_t suffix = local;                             // inclusive suffix sum over lanes >= lane
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const uint32_t v = __shfl_down(suffix, o, 64);
        if (lane + o < 64) suffix += v;
      }
      const uint32_t above = suffix - local;               // keys in bins owned by higher lanes
      if (suffix >= nd && above < nd) {                    // the crossing bin is mine
        uint32_t acc = above;
        int b = per - 1;
        for (; b > 0; --b) {
          if (acc + hist[lane * per + b] >= nd) break;
          acc += hist[lane * per + b];
        }
        sh->need = nd - acc;
        sh->prefix = prefix | ((uint32_t)(lane * per + b) << shift);
      }
      if (lane == 0 && suffix < nd) sh->flag = 1;          // fewer than `need` keys in total
    }
    mask |= (bins - 1) << shift;
    __syncthreads();
  }
  __syncthreads();
  const uint32_t T = sh->flag ? 0u : sh->prefix;
  __syncthreads();
  return T;
}

// NT threads per workgroup, NW = NT / 64 waves.  NT = 256: the throughput shape (thousands of queries, four workgroups per CU).
// NT = 1024: few queries (a 50-query batch on a 256-CU chip: K9 was the longest kernel of the batch) -- the workgroup's time
// is then the LATENCY of its phases, and sixteen waves walk a query's pairs in two rounds of loads instead of seven, expand the
// live rows in three steps instead of twelve.  (Several workgroups per query on shares of the pairs + a merge of their exact
// part lists was built twice and is slower than one workgroup: each part needs ITS OWN n_out-th bound, which prunes far less
// than the query's, and the merge is one more sort -- 50.8 + 13.7 us and 46.8 + 29.3 us against 63.4 us.)
template <int NT>
__global__ __launch_bounds__(NT) void moment_topk_kernel(const float* __restrict__ st, const float* __restrict__ ed,
                                                         const float* __restrict__ w, float* __restrict__ out_score,
                                                         int32_t* __restrict__ out_flat, int kpairs, int lpad,
                                                         int l_ref, int min_l, int max_l, int n_out,
                                                         const float* __restrict__ summ,
                                                         const int32_t* __restrict__ pair_vid,
                                                         const int32_t* __restrict__ vid_len, int dbg) {
  constexpr int NW = NT / 64;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int q = blockIdx.x;
  unsigned long long* s_list = reinterpret_cast<unsigned long long*>(smem);                 // [MT_CAP]
  uint32_t* s_hist = reinterpret_cast<uint32_t*>(s_list + MT_CAP);                          // [2048]
  __shared__ MomentShared sh;
  // ragged corpora: entries >= the video's valid length are exact zeros that K7 did not write (xml_convse_rerank_ex with the
  // same vid_len) -- they are not read here.  s_len[r] = valid clips of pair r (l_ref without vid_len, 0 for skipped pairs).
  __shared__ int s_len[4 * MT_PPW];
  if (tid < 4 * MT_PPW) {
    int n = l_ref;
    if (vid_len && tid < kpairs) {
      const int pv = pair_vid[(int64_t)q * kpairs + tid];
      n = pv >= 0 ? max(0, min(vid_len[pv], l_ref)) : 0;
    }
    s_len[tid] = n;
  }

  const float* gst = st + (int64_t)q * kpairs * lpad;
  const float* ged = ed + (int64_t)q * kpairs * lpad;
  const float* gw = w ? w + (int64_t)q * kpairs : nullptr;

  // ---- 1 + 2. row maxima, in registers ---------------------------------------------------------------------------
  // Wave w owns pairs w, w + NW, ...; lane l owns start clips l and l + 64 of a pair: a = st * w for its two clips and the
  // pair's end probabilities e (lane l: clips l, l + 64) live in registers.  With a, e >= 0 the row maximum
  // max_d (a * e[i + d]) equals a * max_d e[i + d] exactly (rounding is monotone), and the sliding-window maximum of e
  // over [i + min_l, i + max_l) is 3-5 wave shuffles (doubling window widths, one overlapping step, the min_l offset)
  // instead of 16 LDS reads per row -- and nothing of the pair has to be staged in LDS: 24 KiB per workgroup instead
  // of 75, four workgroups per CU instead of two.  The expansion pass re-reads the few live (row, end clip) entries
  // from L2.
  const float w_lo = gw ? (lane < kpairs ? gw[lane] : 0.f) : 1.f;        // pair weights: one vector load,
  const float w_hi = gw ? (lane + 64 < kpairs ? gw[lane + 64] : 0.f) : 1.f;  // broadcast later with a lane read
  auto pair_w = [&](int r) -> float { return __shfl(r < 64 ? w_lo : w_hi, r & 63, 64); };
  const int band = max_l - min_l;
  const int n_t = wave < kpairs ? (kpairs - wave + NW - 1) / NW : 0;      // pairs of this wave: r = wave + NW t
  for (int i = tid; i < 2048; i += NT) s_hist[i] = 0;
  if (tid == 0) { sh.prefix = 0; sh.flag = 0; }
  __syncthreads();

  // raw (st_lo, st_hi, ed_lo, ed_hi) of pair r for this lane's two start clips
  auto pair_load = [&](int r, float (&raw)[4]) {
    const float* sp = gst + r * lpad;
    const float* ep = ged + r * lpad;
    const int len_r = s_len[r];
    const bool in_lo = lane < len_r, in_hi = lane + 64 < len_r;
    raw[0] = in_lo ? sp[lane] : 0.f;
    raw[1] = in_hi ? sp[lane + 64] : 0.f;
    raw[2] = in_lo ? ep[lane] : 0.f;
    raw[3] = in_hi ? ep[lane + 64] : 0.f;
  };
  // (a_lo, a_hi) = st * w and the row maxima (a * window maximum of ed) from the raw values
  auto pair_eval = [&](const float (&raw)[4], float wv, float& a_lo, float& a_hi, float& m_lo, float& m_hi) {
    a_lo = raw[0] * wv;
    a_hi = raw[1] * wv;
    float lo = raw[2], hi = raw[3];
    band_window_max(lo, hi, band, min_l, lane);            // band.h (shared with K7's candidate summaries)
    m_lo = fmaxf(a_lo * lo, 0.f);
    m_hi = fmaxf(a_hi * hi, 0.f);
  };
  auto pair_rows = [&](int r, float wv, float& a_lo, float& a_hi, float& m_lo, float& m_hi) {
    float raw[4];
    pair_load(r, raw);
    pair_eval(raw, wv, a_lo, a_hi, m_lo, m_hi);
  };
  // walk this wave's active pairs, FOUR at a time: the 16 loads of a group are issued before the first one is used
  // (one pair per iteration meant one memory round trip per pair: 25 in a row per pass at k = 100)
  auto for_pairs = [&](auto&& body) {
    for (int t0 = 0; t0 < n_t; t0 += 4) {
      float raw[4][4], wv4[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int r = wave + (t0 + g) * NW;
        wv4[g] = (t0 + g < n_t) ? pair_w(r) : 0.f;
        if (wv4[g] != 0.f) pair_load(r, raw[g]);
        else raw[g][0] = raw[g][1] = raw[g][2] = raw[g][3] = 0.f;
      }
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        if (wv4[g] == 0.f) continue;                       // w == 0: skipped pair (other rank / padding)
        float a2[2], m2[2];
        pair_eval(raw[g], wv4[g], a2[0], a2[1], m2[0], m2[1]);
        body(wave + (t0 + g) * NW, a2, m2);
      }
    }
  };

  // NT = 1024: a wave owns at most 8 pairs -- their (st * w, row maximum) values stay in REGISTERS from the first walk on, and
  // the expansion pass replays them instead of fetching the pairs' rows a second time (one round of dependent loads less on
  // the latency path of a 50-query batch).  With 4 waves a wave owns up to 32 pairs: 128 registers, or the LDS table that
  // was tried in round 5 and cost a workgroup per CU -- the 256-thread form walks twice.
  constexpr bool KEEP = NT >= 1024;
  constexpr int KT = KEEP ? (4 * MT_PPW + NW - 1) / NW : 1;
  float k_a[KT][2], k_m[KT][2];
  if constexpr (KEEP) {
    float raw[KT][4], wv[KT];
#pragma unroll
    for (int t = 0; t < KT; ++t) {
      wv[t] = t < n_t ? pair_w(wave + t * NW) : 0.f;
      if (wv[t] != 0.f) pair_load(wave + t * NW, raw[t]);
      else raw[t][0] = raw[t][1] = raw[t][2] = raw[t][3] = 0.f;
    }
#pragma unroll
    for (int t = 0; t < KT; ++t) {
      k_a[t][0] = k_a[t][1] = k_m[t][0] = k_m[t][1] = 0.f;
      if (wv[t] != 0.f) pair_eval(raw[t], wv[t], k_a[t][0], k_a[t][1], k_m[t][0], k_m[t][1]);     // (wave-uniform)
    }
  }
  // the active pairs of this wave: from the registers (KEEP) or by walking their rows
  auto walk_pairs = [&](auto&& body) {
    if constexpr (KEEP) {
#pragma unroll
      for (int t = 0; t < KT; ++t)
        if (t < n_t) body(wave + t * NW, k_a[t], k_m[t]);        // (skipped pairs hold zeros: no bin, no live row)
    } else {
      for_pairs(body);
    }
  };

  // ---- 2+3. ONE histogram pass over the row maxima on bits [30:20] (8 exponent + 3 mantissa bits); the lower edge
  //           of the bin holding the n_out-th largest row maximum is a lower bound of the n_out-th best score ------
  if (summ) {
    // K7 already took the XML_MOMENT_SUMM largest row maxima of every pair while the pair's rows were in its registers
    // (xml_convse_rerank_ex): the histogram is built from those kpairs x 8 values instead of a pass over the st / ed rows
    // (1 GB at the TVR shape).  Every value is the best candidate of a DISTINCT row, so the lower edge of the bin that holds
    // the n_out-th largest of them is still a valid lower bound of the n_out-th best score -- a weaker one than the bound
    // from all rows when one pair owns more than 8 of the best rows, which costs list refinements below, never exactness.
    const float* gs = summ + (int64_t)q * kpairs * XML_MOMENT_SUMM;
    for (int i = tid; i < kpairs * XML_MOMENT_SUMM; i += NT) {
      const int r = i / XML_MOMENT_SUMM;
      const bool on = gw ? gw[r] != 0.f : true;            // skipped pairs (weight 0) have no summary
      const uint32_t key = on ? __float_as_uint(gs[i]) : 0u;
      if (key != 0u && !(key & 0x80000000u)) atomicAdd(&s_hist[key >> 20], 1u);
    }
  } else {
  walk_pairs([&](int r, const float (&a2)[2], const float (&m2)[2]) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const float m = m2[h];
      const uint32_t key = __float_as_uint(m);
      const uint32_t bin = key >> 20;                      // sign bit is 0: < 2048
      const bool act = key != 0;
      const unsigned long long bal = __ballot(act);
      if (bal == 0) continue;
      const uint32_t first = __shfl(bin, __ffsll((long long)bal) - 1, 64);
      if (__all(!act || bin == first)) {                   // whole wave in one bin (flat distributions): one add
        if (lane == __ffsll((long long)bal) - 1) atomicAdd(&s_hist[first], (uint32_t)__popcll(bal));
      } else if (act) {
        atomicAdd(&s_hist[bin], 1u);
      }
    }
  });
  }
  __syncthreads();
  if (tid < 64) {                                          // wave 0: suffix scan, 32 bins per lane
    uint32_t local = 0;
    for (int b = 0; b < 32; ++b) local += s_hist[lane * 32 + b];
    uint32_t suffix = local;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t v = __shfl_down(suffix, o, 64);
      if (lane + o < 64) suffix += v;
    }
    const uint32_t above = suffix - local, nd = (uint32_t)n_out;
    if (suffix >= nd && above < nd) {
      uint32_t acc = above;
      int b = 31;
      for (; b > 0; --b) {
        if (acc + s_hist[lane * 32 + b] >= nd) break;
        acc += s_hist[lane * 32 + b];
      }
      sh.prefix = (uint32_t)(lane * 32 + b) << 20;
    }
  }
  __syncthreads();
  uint32_t lb = max(sh.prefix, 1u);                        // fewer than n_out positive rows: keep everything > 0
  __syncthreads();
  if (dbg == 61) return;                                   // (debug build, tools/bench_k9.py: phase timing by early exit)

  // ---- 4. expand rows whose maximum reaches lb into the list (raise lb and repeat on overflow) -------------
  // (Measured at 10 000 queries x 100 pairs, flat random-init distributions: 1.00 attempts, ~220 live rows, ~250-340 list
  // entries per query -- the bound is tight; the expansion's time is its walk over the pairs' rows.  Keeping the first walk's
  // row maxima in LDS (25 KB more: three workgroups per CU instead of four) to skip that walk made the kernel 0.65 -> 0.97 ms.)
  uint32_t cnt = 0;
  [[maybe_unused]] uint32_t dbg_attempts = 0, dbg_rows0 = 0;
  int mode = 0;              // 0: candidates >= lb;  1: candidates > lb;  2: candidates == lb, appended behind those of mode 1
  bool exact_done = false;
  auto takes = [&](uint32_t key) -> bool { return mode == 0 ? key >= lb : (mode == 1 ? key > lb : key == lb); };
  for (int attempt = 0; attempt < 12; ++attempt) {
    if (tid == 0) {
      if (mode != 2) sh.cnt = 0;
      sh.need = 0;
    }
    __syncthreads();
    // 4a. rows whose maximum reaches lb -> compact list (row id, st*w) in the histogram area (free between step 3
    //     and the overflow refinement).  Expanding rows in place costs ~14 ballot/atomic iterations per 64-row block
    //     with typically 1-2 live rows in it: that was most of this kernel's time.
    unsigned long long* s_rows = reinterpret_cast<unsigned long long*>(s_hist);   // [1024]: (a bits << 32) | row id
    walk_pairs([&](int r, const float (&a2)[2], const float (&m2)[2]) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int i = lane + h * 64;
        const float a = a2[h];
        const bool row_on = (i < l_ref) && __float_as_uint(m2[h]) >= lb;
        const unsigned long long bal = __ballot(row_on);
        if (bal == 0) continue;
        uint32_t base = 0;
        const int leader = __ffsll((long long)bal) - 1;
        if (lane == leader) base = atomicAdd(&sh.need, (uint32_t)__popcll(bal));
        base = __shfl(base, leader, 64);
        const uint32_t slot = base + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
        if (row_on && slot < 1024u)
          s_rows[slot] = ((unsigned long long)__float_as_uint(a) << 32) | (unsigned long long)(uint32_t)(r * l_ref + i);
      }
    });
    __syncthreads();
    const uint32_t n_rows = sh.need;
    if (attempt == 0) dbg_rows0 = n_rows;
    dbg_attempts = attempt + 1;
    if (n_rows <= 1024u) {
      // 4b. one (row, offset) item per lane: every lane of every wave has work
      const int total = (int)n_rows * band;
      const int total_r = (total + NT - 1) & ~(NT - 1);
      for (int idx = tid; idx < total_r; idx += NT) {
        bool take = false;
        uint32_t key = 0, flat = 0;
        if (idx < total) {
          const int row = idx / band, d = idx - row * band;
          const unsigned long long e = s_rows[row];
          const int ri = (int)(uint32_t)(e & 0xffffffffull);       // r * l_ref + i
          const int r = ri / l_ref, i = ri - r * l_ref;
          const int j = i + min_l + d;
          if (j < s_len[r]) {
            key = __float_as_uint(__uint_as_float((uint32_t)(e >> 32)) * ged[r * lpad + j]);
            take = takes(key);
            flat = (uint32_t)(ri * l_ref + j);
          }
        }
        const unsigned long long bal = __ballot(take);
        if (bal) {
          uint32_t base = 0;
          const int leader = __ffsll((long long)bal) - 1;
          if (lane == leader) base = atomicAdd(&sh.cnt, (uint32_t)__popcll(bal));
          base = __shfl(base, leader, 64);
          const uint32_t slot = base + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
          if (take && slot < (uint32_t)MT_CAP)
            s_list[slot] = ((unsigned long long)key << 32) | (unsigned long long)(0xffffffffu - flat);
        }
      }
    } else {
      // more than 1024 live rows (n_out close to the number of rows): expand in place
#pragma unroll 2
    for (int t = 0; t < n_t; ++t) {
      const int r = wave + t * NW;
      const float wv = pair_w(r);
      if (wv == 0.f) continue;
      float a2[2], m2[2];
      pair_rows(r, wv, a2[0], a2[1], m2[0], m2[1]);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int i = lane + h * 64;
        const float a = a2[h];
        const bool row_on = (i < l_ref) && __float_as_uint(m2[h]) >= lb;
        if (!__any(row_on)) continue;
        const int jend = row_on ? min(s_len[r], i + max_l) : 0;
        for (int d = min_l; d < max_l; ++d) {                     // uniform trip count; lanes predicate themselves
          const int j = i + d;
          const bool ok = row_on && j < jend;
          const uint32_t key = ok ? __float_as_uint(a * ged[r * lpad + j]) : 0u;
          const bool take = ok && takes(key);
          const unsigned long long bal = __ballot(take);
          if (bal) {
            uint32_t base = 0;
            const int leader = __ffsll((long long)bal) - 1;
            if (lane == leader) base = atomicAdd(&sh.cnt, (uint32_t)__popcll(bal));
            base = __shfl(base, leader, 64);
            const uint32_t slot = base + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
            if (take && slot < (uint32_t)MT_CAP)
              s_list[slot] = ((unsigned long long)key << 32) |
                             (unsigned long long)(0xffffffffu - (uint32_t)((r * l_ref + i) * l_ref + j));
          }
        }
      }
    }
    }
    __syncthreads();
    cnt = sh.cnt;
    __syncthreads();
    if (mode == 1) { mode = 2; continue; }     // everything above the tie value is in (fewer than n_out entries): now the ties
    if (mode == 2 || cnt <= (uint32_t)MT_CAP) break;
    // overflow: the n_out-th largest of the MT_CAP stored entries is a tighter valid lower bound
    const uint32_t t_c = block_radix_select(
        [&](auto f) {
          for (int i = tid; i < MT_CAP; i += NT) f((uint32_t)(s_list[i] >> 32), true);
        },
        (uint32_t)n_out, s_hist, &sh);
    if (t_c > lb) { lb = t_c; continue; }
    // No progress.  WHICH candidates were stored is a race between the waves (sixteen of them: the best pair's entries can be
    // a sixteenth of the list), so the stored subset need not separate anything.  The exact n_out-th largest candidate, by a
    // radix select over ALL band candidates >= lb (three more walks over the pairs: degenerate inputs only -- flat
    // probabilities with one boosted pair, tests/test_gpu_kernels.py::test_moment_topk_flat_distribution_fallback):
    if (!exact_done) {
      exact_done = true;
      const uint32_t lb0 = lb;
      const uint32_t t_x = block_radix_select(
          [&](auto f) {
            for (int t = 0; t < n_t; ++t) {
              const int r = wave + t * NW;
              const float wv = pair_w(r);
              float a2[2] = {0.f, 0.f}, m2[2];
              if (wv != 0.f) pair_rows(r, wv, a2[0], a2[1], m2[0], m2[1]);
#pragma unroll
              for (int h = 0; h < 2; ++h) {
                const int i = lane + h * 64;
                for (int d = min_l; d < max_l; ++d) {
                  const int j = i + d;
                  const bool ok = wv != 0.f && i < l_ref && j < s_len[r];
                  const uint32_t key = ok ? __float_as_uint(a2[h] * ged[r * lpad + j]) : 0u;
                  f(key, ok && key >= lb0);
                }
              }
            }
          },
          (uint32_t)n_out, s_hist, &sh);
      if (t_x > lb) { lb = t_x; continue; }
    }
    // lb IS the n_out-th best score and more than MT_CAP candidates reach it: fewer than n_out lie above it -- those first,
    // then as many of the exact ties as the list holds (the reference's order among equal scores is unspecified)
    mode = 1;
  }
  cnt = min(cnt, (uint32_t)MT_CAP);
#ifdef XML_DEBUG_VARIANTS
  if (dbg == 63 && tid == 0) {
    atomicAdd(&g_k9_stats[0], (unsigned long long)dbg_attempts);
    atomicAdd(&g_k9_stats[1], (unsigned long long)dbg_rows0);
    atomicAdd(&g_k9_stats[2], (unsigned long long)(dbg_rows0 > 1024u));
    atomicAdd(&g_k9_stats[3], (unsigned long long)cnt);
    atomicAdd(&g_k9_stats[4], 1ull);
  }
#endif
  if (dbg == 62) return;

  // ---- 5. order the list on (score desc, flat asc) and emit n_out ---------------------------------------------------
  // The keys are distinct (the flat index is part of them), so the position of an entry is the number of entries above it.
  // Up to three entries per thread (the usual list: 250-340 entries for 200 results) are RANKED against the whole list --
  // every lane reads the same LDS word, a broadcast -- and stored at their rank: ~cnt short iterations and no barrier,
  // where a bitonic network over the next power of two is 45-66 barrier-separated stages.  Same order, same output.
  if (cnt <= 3u * NT) {
    unsigned long long key[3];
    uint32_t rank[3] = {0u, 0u, 0u};
#pragma unroll
    for (int e = 0; e < 3; ++e) key[e] = (uint32_t)(tid + e * NT) < cnt ? s_list[tid + e * NT] : ~0ull;
    const int n = (int)cnt;
#pragma unroll 4
    for (int j = 0; j < n; ++j) {
      const unsigned long long v = s_list[j];
#pragma unroll
      for (int e = 0; e < 3; ++e) rank[e] += v > key[e] ? 1u : 0u;
    }
#pragma unroll
    for (int e = 0; e < 3; ++e) {
      if ((uint32_t)(tid + e * NT) < cnt && rank[e] < (uint32_t)n_out) {
        out_score[(int64_t)q * n_out + rank[e]] = __uint_as_float((uint32_t)(key[e] >> 32));
        out_flat[(int64_t)q * n_out + rank[e]] = (int32_t)(0xffffffffu - (uint32_t)(key[e] & 0xffffffffull));
      }
    }
    for (int i = (int)cnt + tid; i < n_out; i += NT) {        // fewer candidates than results: (0, -1) rows
      out_score[(int64_t)q * n_out + i] = 0.f;
      out_flat[(int64_t)q * n_out + i] = -1;
    }
    return;
  }
  // longer lists (n_out close to the number of positive candidates, exact ties at the bound): bitonic sort, padded with zeros
  int npow = 256;
  while ((uint32_t)npow < cnt) npow <<= 1;
  for (int i = (int)cnt + tid; i < npow; i += NT) s_list[i] = 0ull;
  __syncthreads();
  for (int size = 2; size <= npow; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int i = tid; i < (npow >> 1); i += NT) {
        const int lo = ((i / stride) * stride << 1) + (i % stride);
        const int hi = lo + stride;
        const unsigned long long a = s_list[lo], b = s_list[hi];
        const bool desc = (lo & size) == 0;
        if (desc ? (a < b) : (a > b)) { s_list[lo] = b; s_list[hi] = a; }
      }
      __syncthreads();
    }
  }
  for (int i = tid; i < n_out; i += NT) {
    float sc = 0.f;
    int32_t flat = -1;
    if (i < npow) {
      const unsigned long long c = s_list[i];
      if (c != 0ull) {
        sc = __uint_as_float((uint32_t)(c >> 32));
        flat = (int32_t)(0xffffffffu - (uint32_t)(c & 0xffffffffull));
      }
    }
    out_score[(int64_t)q * n_out + i] = sc;
    out_flat[(int64_t)q * n_out + i] = flat;
  }
}

extern "C" int xml_moment_topk(const float* st, const float* ed, const float* w, float* out_score, int32_t* out_flat,
                               int nq, int kpairs, int lpad, int l_ref, int min_l, int max_l, int n_out,
                               xml_stream_t stream) {
  return xml_moment_topk_ex(st, ed, w, nullptr, nullptr, nullptr, out_score, out_flat, nq, kpairs, lpad, l_ref, min_l, max_l,
                            n_out, stream);
}

extern "C" int xml_moment_topk_ex(const float* st, const float* ed, const float* w, const float* summ,
                                  const int32_t* pair_vid, const int32_t* vid_len, float* out_score, int32_t* out_flat, int nq,
                                  int kpairs, int lpad, int l_ref, int min_l, int max_l, int n_out, xml_stream_t stream) {
  XML_ENTER();
  if (!st || !ed || !out_score || !out_flat || nq <= 0 || kpairs <= 0 || lpad <= 0 || l_ref <= 0 || n_out <= 0)
    return XML_ERR_BAD_ARG;
  if ((pair_vid == nullptr) != (vid_len == nullptr)) return XML_ERR_BAD_ARG;
  if (l_ref > lpad || min_l < 0 || max_l <= min_l) return XML_ERR_BAD_ARG;
  if (n_out > 1024 || lpad > 128 || kpairs > 4 * MT_PPW) return XML_ERR_UNSUPPORTED;   // (st, ed: probabilities, >= 0)
  const size_t lds = (size_t)MT_CAP * 8 + 2048 * 4;
  // fewer queries than half the CUs and enough pairs to give sixteen waves work: the latency shape (see the kernel)
  if (nq <= 128 && kpairs >= 32)
    hipLaunchKernelGGL(moment_topk_kernel<1024>, dim3(nq), dim3(1024), lds, (hipStream_t)stream, st, ed, w, out_score, out_flat,
                       kpairs, lpad, l_ref, min_l, max_l, n_out, summ, pair_vid, vid_len, (int)g_q2c_ablation);
  else
    hipLaunchKernelGGL(moment_topk_kernel<256>, dim3(nq), dim3(256), lds, (hipStream_t)stream, st, ed, w, out_score, out_flat,
                       kpairs, lpad, l_ref, min_l, max_l, n_out, summ, pair_vid, vid_len, (int)g_q2c_ablation);
  XML_CHECK_LAUNCH();
  return XML_OK;
}
