/*
 * strand_oracle.c -- CPU restatement of c_utils.filter_strand_list_segments
 * (/root/reference/c_utils/c_utils.pyx:83-127), the Cython helper the Stage-III smoothness loss
 * calls (loss/losses.py:195).  TEST INFRASTRUCTURE / CPU BASELINE ONLY.
 *
 * Pinned: tests/golden/ref_python_pins.npz holds outputs of the reference .pyx itself (built with
 * the image's Cython by oracle/build_ref.py into oracle/_ref/) for ragged inputs incl. empty and
 * single-row strands.
 *
 * The reference takes a numpy object array of [n_j,2] int64 arrays; the flat form here is the
 * concatenation `rows[total,2]` plus `offsets[S+1]`.  Output: [pairs,2,2] int64, pairs =
 * sum(max(n_j-1,0)); returns pairs.  Pass out=NULL to only count (first pass, :106-110).
 */
#include <stdint.h>
#include <stddef.h>

int64_t hgs_oracle_filter_strand_segments(int64_t S, const int64_t* offsets, const int64_t* rows, int64_t* out) {
  int64_t total = 0;
  for (int64_t j = 0; j < S; j++) {
    int64_t n = offsets[j + 1] - offsets[j];
    if (n >= 2) total += n - 1;
  }
  if (!out) return total;
  int64_t cur = 0;
  for (int64_t j = 0; j < S; j++) {
    int64_t n = offsets[j + 1] - offsets[j];
    if (n < 2) continue;
    const int64_t* s = rows + 2 * offsets[j];
    for (int64_t i = 0; i < n - 1; i++) { /* :118-124 */
      out[4 * cur + 0] = s[2 * i];
      out[4 * cur + 1] = s[2 * i + 1];
      out[4 * cur + 2] = s[2 * (i + 1)];
      out[4 * cur + 3] = s[2 * (i + 1) + 1];
      cur++;
    }
  }
  return total;
}
