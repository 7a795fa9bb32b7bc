// Banded row maxima of the moment scorer, shared by K9 (moment.hip) and K7's candidate summaries (convse.hip).
//   reference: generate_min_max_length_mask  xml/inference.py:170-192  (a moment (i, j) counts when min_l <= j - i < max_l)
// A 128-clip row lives in a wave as (lo: clip lane, hi: clip lane + 64).  With non-negative inputs the best score of a start
// clip i, max_d (a[i] * e[i + d]), equals a[i] * max_d e[i + d] exactly (rounding is monotone), and the sliding-window maximum
// of e over [i + min_l, i + max_l) is a few wave shuffles: doubling window widths, one overlapping step, the min_l offset.
#pragma once
#include "common.h"

// x[c + s] for the sequence held as (lo, hi); zero beyond the end
__device__ __forceinline__ void band_shifted(float lo, float hi, int s, int lane, float& olo, float& ohi) {
  const int s1 = s & 63;
  const int srcl = (lane + s1) & 63;
  const bool wrap = lane + s1 >= 64;
  const float from_lo = __shfl(lo, srcl, 64), from_hi = __shfl(hi, srcl, 64);
  if (s < 64) { olo = wrap ? from_hi : from_lo; ohi = wrap ? 0.f : from_hi; }
  else { olo = wrap ? 0.f : from_hi; ohi = 0.f; }
}

// (lo, hi) <- max over d in [min_l, min_l + band) of e[c + d]     (band = max_l - min_l >= 1)
__device__ __forceinline__ void band_window_max(float& lo, float& hi, int band, int min_l, int lane) {
  int p = 1;
  while (2 * p <= band) {                                // window width p -> 2 p
    float slo, shi;
    band_shifted(lo, hi, p, lane, slo, shi);
    lo = fmaxf(lo, slo); hi = fmaxf(hi, shi);
    p <<= 1;
  }
  if (p < band) {                                        // two overlapping windows of width p cover width band
    float slo, shi;
    band_shifted(lo, hi, band - p, lane, slo, shi);
    lo = fmaxf(lo, slo); hi = fmaxf(hi, shi);
  }
  if (min_l > 0) band_shifted(lo, hi, min_l, lane, lo, hi);
}
