// msda_hip.hip -- multi-scale deformable attention forward/backward for gfx950 (MI355X).
//
// Hand-written HIP; wave = 64 lanes.  C ABI declared in include/msda_hip.h.
// What each kernel replaces in the reference (paths relative to
// mask2former/modeling/pixel_decoder/ops/src/cuda/) is noted at the kernel.
//
// Vocabulary: a *pair* is one (n, q, m) = (image, query, head); a *sample* is one
// (pair, level, point); a pixel-head *row* is the D contiguous channels of
// value[n, s, m, :].
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>

#include "msda_hip.h"

namespace {
#include "msda_common.inc"
#include "msda_forward.inc"
#include "msda_generic.inc"
#include "msda_backward_binned.inc"
#include "msda_fused.inc"

// ---------------------------------------------------------------------------------------
// Host side.
// ---------------------------------------------------------------------------------------
// hipGetLastError is per thread and sticky until read.  An error some other library left pending on this thread
// must neither be reported as a failed MSDA launch nor be consumed here (the application's own check would never
// see it): every entry point notes what is pending when it is entered (check_dims), and a launch has failed when
// the pending error afterwards is a different one.  Nothing here resets the thread's error state.
thread_local hipError_t g_pending = hipSuccess;

int check_dims(const char *who, int N, int S, int M, int D, int L, int Lq, int P) {
  g_pending = hipPeekAtLastError();
  if (N < 0 || S < 0 || M <= 0 || D <= 0 || L <= 0 || Lq < 0 || P <= 0)
    return fail(MSDA_ERR_INVALID_ARGUMENT,
                "%s: bad dimensions batch=%d spatial_size=%d num_heads=%d channels=%d "
                "num_levels=%d num_query=%d num_point=%d",
                who, N, S, M, D, L, Lq, P);
  return MSDA_OK;
}

int check_launch(const char *who) {
  const hipError_t e = hipPeekAtLastError();
  if (e != hipSuccess && e != g_pending)
    return fail(MSDA_ERR_LAUNCH, "%s: kernel launch failed: %s", who, hipGetErrorString(e));
  g_launches.fetch_add(1, std::memory_order_relaxed);
  return MSDA_OK;
}

inline int pick_group(int D) {
  int g = 1;
  while (g < D && g < 64) g <<= 1;
  return g;
}

inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// spatial_size == 0 with queries: no level can lie inside the value tensor, every sample reads zero
int zero_output(const char *who, void *out, size_t bytes, hipStream_t st) {
  const hipError_t e = hipMemsetAsync(out, 0, bytes, st);
  if (e != hipSuccess) return fail(MSDA_ERR_LAUNCH, "%s: hipMemsetAsync(output) failed: %s", who, hipGetErrorString(e));
  return MSDA_OK;
}

template <typename T>
int forward_generic(const T *value, const int64_t *shapes, const int64_t *lsi, const T *loc,
                    const T *aw, int N, int S, int M, int D, int L, int Lq, int P, T *out,
                    hipStream_t st) {
  const size_t npairs = (size_t)N * Lq * M;
  const int G = pick_group(D);
  const size_t nblk = (npairs + (256 / G) - 1) / (256 / G);
  if (nblk > 0x7fffffffull)
    return fail(MSDA_ERR_INVALID_ARGUMENT, "msda_forward: problem too large (%zu blocks)", nblk);
#define MSDA_LAUNCH_FWD(GG)                                                                  \
  case GG:                                                                                   \
    hipLaunchKernelGGL((msda_fwd_generic_kernel<T, GG>), dim3((unsigned)nblk), dim3(256), 0, \
                       st, value, shapes, lsi, loc, aw, S, M, D, L, Lq, P, npairs, out);     \
    break;
  switch (G) {
    MSDA_LAUNCH_FWD(1)
    MSDA_LAUNCH_FWD(2)
    MSDA_LAUNCH_FWD(4)
    MSDA_LAUNCH_FWD(8)
    MSDA_LAUNCH_FWD(16)
    MSDA_LAUNCH_FWD(32)
    MSDA_LAUNCH_FWD(64)
  }
#undef MSDA_LAUNCH_FWD
  return check_launch("msda_forward");
}

template <typename T>
int backward_generic(const T *value, const int64_t *shapes, const int64_t *lsi, const T *loc,
                     const T *aw, const T *grad_out, int N, int S, int M, int D, int L, int Lq,
                     int P, T *grad_value, T *grad_loc, T *grad_aw, hipStream_t st) {
  const size_t npairs = (size_t)N * Lq * M;
  const int G = pick_group(D);
  const size_t nblk = (npairs + (256 / G) - 1) / (256 / G);
  if (nblk > 0x7fffffffull)
    return fail(MSDA_ERR_INVALID_ARGUMENT, "msda_backward: problem too large (%zu blocks)", nblk);
#define MSDA_LAUNCH_BWD(GG)                                                                  \
  case GG:                                                                                   \
    hipLaunchKernelGGL((msda_bwd_generic_kernel<T, GG>), dim3((unsigned)nblk), dim3(256), 0, \
                       st, grad_out, value, shapes, lsi, loc, aw, S, M, D, L, Lq, P, npairs, \
                       grad_value, grad_loc, grad_aw);                                       \
    break;
  switch (G) {
    MSDA_LAUNCH_BWD(1)
    MSDA_LAUNCH_BWD(2)
    MSDA_LAUNCH_BWD(4)
    MSDA_LAUNCH_BWD(8)
    MSDA_LAUNCH_BWD(16)
    MSDA_LAUNCH_BWD(32)
    MSDA_LAUNCH_BWD(64)
  }
#undef MSDA_LAUNCH_BWD
  return check_launch("msda_backward");
}

template <typename T>
int forward_impl(const T *value, const int64_t *shapes, const int64_t *lsi, const T *loc,
                 const T *aw, int N, int S, int M, int D, int L, int Lq, int P, T *out,
                 void *stream);

template <>
int forward_impl<double>(const double *value, const int64_t *shapes, const int64_t *lsi,
                         const double *loc, const double *aw, int N, int S, int M, int D, int L,
                         int Lq, int P, double *out, void *stream) {
  if (int rc = check_dims("msda_forward_f64", N, S, M, D, L, Lq, P)) return rc;
  if ((size_t)N * Lq == 0) return MSDA_OK;
  return forward_generic<double>(value, shapes, lsi, loc, aw, N, S, M, D, L, Lq, P, out,
                                 static_cast<hipStream_t>(stream));
}

// Which D == 32 forward: queries-by-head (default) or, with MSDA_FWD_IMPL=plain, the
// heads-of-a-query kernel (A/B measurements; also the path of L*P > 32 and unaligned tensors).
int fwd_impl_choice() {
  const char *e = getenv("MSDA_FWD_IMPL");
  return (e && strcmp(e, "plain") == 0) ? 0 : 1;
}

// Unit order of the queries-by-head forward: regions of 2^6 runs, runs fastest inside a region (C3: fp32 1.399 ->
// 1.385 ms, bf16 1.011 -> 0.99 ms against heads fastest = 0; 2^2 ... 2^10 measured, profiles/r03/experiments).
// MSDA_FWD_REGION_LOG2 overrides it for A/B runs; read once per process.
uint32_t fwd_region_log2() {
  static const uint32_t v = [] {
    const char *e = getenv("MSDA_FWD_REGION_LOG2");
    const int r = e ? atoi(e) : 6;
    return (uint32_t)(r < 0 ? 0 : (r > 20 ? 20 : r));
  }();
  return v;
}

template <typename VT>
void launch_forward_d32(const VT *value, const int64_t *shapes, const int64_t *lsi, const float *loc,
                        const float *aw, int S, int M, int L, int Lq, int P, size_t npairs, VT *out,
                        hipStream_t st) {
  static const int impl = fwd_impl_choice();
  constexpr uint32_t GP = 64u / FwdLanes<VT>::LPG;
  const size_t lds = (size_t)4 * 2 * GP * (size_t)(L * P + 1) * 16;
  if (impl != 0 && lds <= FWD_QH_MAX_LDS && (reinterpret_cast<uintptr_t>(loc) & 7u) == 0 && aligned16(value) &&
      aligned16(out)) {
    const uint32_t nq = (uint32_t)(npairs / (size_t)M), nunits = ((nq + GP - 1u) / GP) * (uint32_t)M;
    const uint32_t nblk = (nunits + 3u) / 4u;
    hipLaunchKernelGGL((msda_fwd_d32_qh_kernel<VT, false>), dim3(nblk), dim3(256), lds, st, value, shapes, lsi,
                       loc, aw, (const float *)nullptr, S, M, L, Lq, P, nq, nunits, nblk,
                       aligned16(loc) && (reinterpret_cast<uintptr_t>(aw) & 7u) == 0, fwd_div_magic((uint32_t)(L * P)),
                       fwd_div_magic((uint32_t)P), fwd_div_magic32((uint32_t)M, nunits), fwd_div_magic_lq((uint32_t)Lq, nq, GP),
                       fwd_region_log2(), out);
    return;
  }
  const uint32_t nblk = (uint32_t)((npairs + 31) / 32);
  if (P == 4 && aligned16(loc) && aligned16(aw))
    hipLaunchKernelGGL((msda_fwd_d32_kernel<VT, true>), dim3(nblk), dim3(256), 0, st, value, shapes,
                       lsi, loc, aw, S, M, L, Lq, P, (uint32_t)npairs, nblk, out);
  else
    hipLaunchKernelGGL((msda_fwd_d32_kernel<VT, false>), dim3(nblk), dim3(256), 0, st, value, shapes,
                       lsi, loc, aw, S, M, L, Lq, P, (uint32_t)npairs, nblk, out);
}

template <>
int forward_impl<float>(const float *value, const int64_t *shapes, const int64_t *lsi,
                        const float *loc, const float *aw, int N, int S, int M, int D, int L,
                        int Lq, int P, float *out, void *stream) {
  if (int rc = check_dims("msda_forward_f32", N, S, M, D, L, Lq, P)) return rc;
  if ((size_t)N * Lq == 0) return MSDA_OK;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const size_t npairs = (size_t)N * Lq * M;
  const size_t limit = (size_t)1 << 30;  // elements, i.e. 4 GiB of fp32
  const bool small = (size_t)N * S * M * D < limit && npairs * L * P * 2 < limit &&
                     npairs * D < limit;
  if (D == 32 && S > 0 && small && aligned16(value) && aligned16(out)) {
    launch_forward_d32<float>(value, shapes, lsi, loc, aw, S, M, L, Lq, P, npairs, out, st);
    return check_launch("msda_forward_f32");
  }
  return forward_generic<float>(value, shapes, lsi, loc, aw, N, S, M, D, L, Lq, P, out, st);
}

int num_cus() {
  static thread_local int cached_dev = -1, cached = 0;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 256;
  if (dev != cached_dev) {
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0)
      v = 256;
    cached = v;
    cached_dev = dev;
  }
  return cached;
}

template <typename T>
int backward_atomic(const char *who, const T *value, const int64_t *shapes, const int64_t *lsi,
                    const T *loc, const T *aw, const T *grad_out, int N, int S, int M, int D,
                    int L, int Lq, int P, T *grad_value, T *grad_loc, T *grad_aw,
                    hipStream_t st) {
  const size_t vbytes = sizeof(T) * (size_t)N * S * M * D;
  if (vbytes) {
    const hipError_t e = hipMemsetAsync(grad_value, 0, vbytes, st);
    if (e != hipSuccess)
      return fail(MSDA_ERR_LAUNCH, "%s: hipMemsetAsync(grad_value) failed: %s", who,
                  hipGetErrorString(e));
  }
  if ((size_t)N * Lq == 0) return MSDA_OK;
  return backward_generic<T>(value, shapes, lsi, loc, aw, grad_out, N, S, M, D, L, Lq, P,
                             grad_value, grad_loc, grad_aw, st);
}

template <typename T>
int backward_impl(const char *who, const T *value, const int64_t *shapes, const int64_t *lsi,
                  const T *loc, const T *aw, const T *grad_out, int N, int S, int M, int D,
                  int L, int Lq, int P, T *grad_value, T *grad_loc, T *grad_aw, void *stream);

template <>
int backward_impl<double>(const char *who, const double *value, const int64_t *shapes,
                          const int64_t *lsi, const double *loc, const double *aw,
                          const double *grad_out, int N, int S, int M, int D, int L, int Lq,
                          int P, double *grad_value, double *grad_loc, double *grad_aw,
                          void *stream) {
  if (int rc = check_dims(who, N, S, M, D, L, Lq, P)) return rc;
  return backward_atomic<double>(who, value, shapes, lsi, loc, aw, grad_out, N, S, M, D, L, Lq,
                                 P, grad_value, grad_loc, grad_aw,
                                 static_cast<hipStream_t>(stream));
}

// ---- binned backward: shape test, workspace size, launches ----
bool binned_shape(int N, int S, int M, int D, int L, int Lq, int P) {
  if (!(D == 32 && P <= binned::MAXP && L <= binned::MAXL && N > 0 && S > 0 && Lq > 0)) return false;
  const size_t npairs = (size_t)N * Lq * M, limit = (size_t)1 << 30;
  return (size_t)N * S * M * D < limit && npairs * L * P * 2 < limit && npairs * D < limit &&
         (size_t)Lq * M * L * (size_t)(M * L) < ((size_t)1 << 40) &&            // div_magic of the binning kernels
         (size_t)N * Lq < ((size_t)1 << 28) &&                                   // entry = pair-less id << 4 | mask
         npairs * L * binned::MAX_ENTRIES_PER_QUAD < ((size_t)1 << 32);         // 32-bit list positions
}

// Workspace of one group of images, uint32 words: counters[key_cap + 1] + pad, list starts[key_cap + 1] + zero
// word, control words, descriptors[quads], entries[e_cap]; key_cap: one list per (image, head, tile), and a level
// has at most H*W tiles.  Two ways to live with less than the whole batch's worst case (16 entries per quad):
//   * a batch may be taken in several groups of images, one after the other through the same workspace (images
//     are independent);
//   * a group's entry region may be smaller than 16 per quad: its lists are then written and consumed in several
//     ROUNDS (binned.inc), as many as the worst case needs -- model-like locations fill 1.35 entries per quad,
//     uniformly random ones 4.5, so with 5 or more per quad the rounds after the first find nothing to do.
size_t binned_key_cap(int N, int S, int M) { return (size_t)N * M * S; }
size_t align256(size_t b) { return (b + 255) & ~(size_t)255; }
size_t binned_quads(int N, int M, int L, int Lq) { return (size_t)N * Lq * M * L; }
size_t binned_fixed_words(int N, int S, int M, int L, int Lq) {
  return 2 * (binned_key_cap(N, S, M) + 2) + binned::CTRL_WORDS + binned_quads(N, M, L, Lq);
}

// Small batches: fewer than ~8 tiles per workgroup means the tile kernel runs as long as its longest list
// (coarse-level tiles), so those lists are split over workgroups that ADD their partial sums with float atomics
// into a zero-filled fp32 tensor: grad_value itself, or -- bf16 grad_value -- a staging tensor in the workspace.
bool binned_split(int N, int S, int M) {
  const size_t est_items = (size_t)N * M * (size_t)((S + binned::TP - 1) / binned::TP);
  return est_items < (size_t)16 * num_cus();
}
size_t binned_staging_bytes(int N, int S, int M) { return align256((size_t)N * S * M * 32 * sizeof(float)); }
// every list of the group at once: one round
size_t binned_lists_bytes(int N, int S, int M, int L, int Lq) {
  return align256(4 * (binned_fixed_words(N, S, M, L, Lq) + binned::MAX_ENTRIES_PER_QUAD * binned_quads(N, M, L, Lq)));
}
size_t binned_workspace_bytes(int N, int S, int M, int L, int Lq, int P, bool gv16 = false) {
  (void)P;
  size_t bytes = binned_lists_bytes(N, S, M, L, Lq);
  if (gv16 && binned_split(N, S, M)) bytes += binned_staging_bytes(N, S, M);
  return bytes;
}
// four entries per quad for the whole batch -- four rounds, of which model-like locations (1.35 entries per quad)
// use the first; the other three cost two empty launches each: bs 32 at 1024^2 0.37 GB instead of 1.17 GB and
// 2.90 instead of 2.89 ms, bs 2 at 1536^2 0.498 instead of 0.487 ms
constexpr size_t COMPACT_ENTRIES_PER_QUAD = 4;
size_t binned_compact_bytes(int N, int S, int M, int L, int Lq, bool gv16) {
  size_t bytes = align256(4 * (binned_fixed_words(N, S, M, L, Lq) + COMPACT_ENTRIES_PER_QUAD * binned_quads(N, M, L, Lq) +
                               4 * (size_t)Lq));
  if (gv16 && binned_split(N, S, M)) bytes += binned_staging_bytes(N, S, M);
  return bytes;
}
// LDS histogram of the binning passes: one word per list of an image.  The host cannot read the level shapes, so it
// guesses from spatial_size -- levels that are not far from square have about S/256 tiles plus their borders -- and
// takes the small histogram (8 KB: eight workgroups per CU instead of four; scatter pass 135 -> 124 us at C3) when
// twice that fits; a wrong guess costs speed, not results (the kernels fall back to counting in global memory).
uint32_t binned_kcap(int S, int M, int L) {
  const size_t guess = (size_t)M * (2 * (size_t)((S + binned::TP - 1) / binned::TP) + 8 * (size_t)L);
  return guess <= (size_t)binned::KCAP_SMALL ? binned::KCAP_SMALL : binned::KCAP;
}
// the least a group of ONE image can run in: one entry per quad plus the longest list there can be (a list holds
// at most one entry per point of each of the image's queries) -- sixteen rounds
size_t binned_min_bytes(int S, int M, int L, int Lq) {
  return 4 * (binned_fixed_words(1, S, M, L, Lq) + binned_quads(1, M, L, Lq) + 4 * (size_t)Lq);
}

// Entry region and rounds of a group of n images in `bytes` of workspace; e_cap == 0: the group does not fit.
// Rounds: a round that is not the last holds more than e_cap - 4 * Lq entries (the next list did not fit, and no
// list is longer than 4 * Lq), and there are at most 16 per quad in all.
struct BinnedRegion {
  size_t e_cap = 0;
  int rounds = 0;
};
BinnedRegion binned_region(int n, int S, int M, int L, int Lq, size_t bytes) {
  BinnedRegion r;
  const size_t words = bytes / 4, fixed = binned_fixed_words(n, S, M, L, Lq), quads = binned_quads(n, M, L, Lq);
  const size_t full = binned::MAX_ENTRIES_PER_QUAD * quads, longest = 4 * (size_t)Lq;
  if (words < fixed + quads + longest) return r;
  const size_t avail = words - fixed;
  if (avail >= full) {
    r.e_cap = full;
    r.rounds = 1;
  } else {
    r.e_cap = avail;
    r.rounds = (int)((full + (avail - longest) - 1) / (avail - longest));     // <= 16: avail - longest >= quads
  }
  return r;
}
// Images per group (0: not even one fits): the most that need at most four rounds -- three of which are empty
// launches for any real set of locations -- else one image at a time in as many rounds as it takes.
constexpr int COMFORTABLE_ROUNDS = 4;
int binned_group(int N, int S, int M, int L, int Lq, size_t workspace_bytes) {
  for (int g = N; g >= 1; --g) {
    const BinnedRegion r = binned_region(g, S, M, L, Lq, workspace_bytes);
    if (r.e_cap && r.rounds <= COMFORTABLE_ROUNDS) return g;
  }
  return binned_region(1, S, M, L, Lq, workspace_bytes).e_cap ? 1 : 0;
}

template <typename VT, typename GT>
int launch_binned_group(const char *who, const VT *value, const int64_t *shapes,
                        const int64_t *lsi, const float *loc, const float *aw,
                        const VT *grad_out, int N, int S, int M, int L, int Lq, int P,
                        GT *grad_value, float *grad_loc, float *grad_aw, void *workspace, size_t workspace_bytes,
                        hipStream_t st, bool planar = false) {
  constexpr bool GV16 = sizeof(GT) == 2;
  // Small batches split their long lists over workgroups that add into a zero-filled fp32 tensor: grad_value
  // itself, or (bf16 grad_value) a staging tensor at the END of the workspace -- when there is room for it without
  // costing the lists a round; a group that cannot have it runs unsplit (same results).
  bool split = binned_split(N, S, M);
  BinnedRegion reg = binned_region(N, S, M, L, Lq, workspace_bytes);
  size_t staging_off = 0;
  if (split && GV16) {
    const size_t sb = binned_staging_bytes(N, S, M);
    staging_off = workspace_bytes > sb ? (workspace_bytes - sb) & ~(size_t)255 : 0;
    const BinnedRegion with = binned_region(N, S, M, L, Lq, staging_off);
    if (staging_off && with.e_cap && with.rounds <= reg.rounds)
      reg = with;
    else
      split = false;
  }
  if (reg.e_cap == 0)     // launch_backward_binned chose the group to fit
    return fail(MSDA_ERR_INVALID_ARGUMENT, "%s: workspace of %zu bytes is too small for %d image(s)", who, workspace_bytes, N);
  const uint32_t key_cap = (uint32_t)binned_key_cap(N, S, M);
  uint32_t *counts = static_cast<uint32_t *>(workspace);
  uint32_t *offs = counts + (size_t)key_cap + 2;
  uint32_t *zero_word = offs + (size_t)key_cap + 1;
  uint32_t *ctrl = zero_word + 1;
  uint32_t *desc = ctrl + binned::CTRL_WORDS;
  uint32_t *entries = desc + binned_quads(N, M, L, Lq);
  const uint32_t spi = (uint32_t)((size_t)Lq * M * L);          // quads per image
  const uint32_t bpi = (spi + binned::BIN_QB - 1) / binned::BIN_QB;
  const int cus = num_cus();
  hipLaunchKernelGGL(binned::msda_bwd_bin_zero_kernel, dim3(cus), dim3(256), 0, st, shapes, lsi, N, S, M, L,
                     Lq, P, key_cap, GV16 ? 3u : 4u, counts, static_cast<void *>(grad_value), grad_loc, grad_aw);
  if (int rc = check_launch(who)) return rc;
  const uint64_t magic_ml = (((uint64_t)1 << 40) + (uint64_t)(M * L) - 1) / (uint64_t)(M * L);
  const uint64_t magic_l = (((uint64_t)1 << 40) + (uint64_t)L - 1) / (uint64_t)L;
  const uint32_t kcap = binned_kcap(S, M, L);
  // the fused boundary: grad_loc / grad_aw in the scratch layout [M, L, N * Lq, P] (binned.inc, PLANAR)
  const uint32_t planar_nq = planar ? (uint32_t)((size_t)N * Lq) : 0u;
  hipLaunchKernelGGL(binned::msda_bwd_bin_kernel<0>, dim3((uint32_t)N * bpi), dim3(binned::BIN_NT),
                     kcap * 4, st, shapes, lsi, loc, N, S, M, L, Lq, P, spi, bpi, magic_ml, magic_l, key_cap, 0u, kcap,
                     planar_nq, counts, desc, entries, grad_loc, grad_aw);
  if (int rc = check_launch(who)) return rc;
  hipLaunchKernelGGL(binned::msda_bwd_bin_scan_kernel, dim3(1), dim3(1024), 0, st, shapes, lsi, N, S, M, L,
                     key_cap, (uint32_t)reg.e_cap, reg.rounds, counts, offs);
  if (int rc = check_launch(who)) return rc;
  const size_t gv_elems = (size_t)N * S * M * 32;
  float *sums = nullptr;                               // split mode: what the parts of a tile's list add into
  if (split) {
    sums = GV16 ? reinterpret_cast<float *>(static_cast<char *>(workspace) + staging_off)
                : reinterpret_cast<float *>(grad_value);
    const hipError_t e = hipMemsetAsync(sums, 0, sizeof(float) * gv_elems, st);
    if (e != hipSuccess)
      return fail(MSDA_ERR_LAUNCH, "%s: hipMemsetAsync(grad_value) failed: %s", who, hipGetErrorString(e));
  }
  // one scatter pass + one tile-kernel pass per round (one round unless the caller's workspace is short); a round
  // without lists returns at once
  for (int r = 0; r < reg.rounds; ++r) {
    if (reg.rounds == 1)
      hipLaunchKernelGGL(binned::msda_bwd_bin_kernel<1>, dim3((uint32_t)N * bpi), dim3(binned::BIN_NT),
                         kcap * 4, st, shapes, lsi, loc, N, S, M, L, Lq, P, spi, bpi, magic_ml, magic_l, key_cap, 0u,
                         kcap, planar_nq, counts, desc, entries, grad_loc, grad_aw);
    else
      hipLaunchKernelGGL(binned::msda_bwd_bin_kernel<2>, dim3((uint32_t)N * bpi), dim3(binned::BIN_NT),
                         kcap * 4, st, shapes, lsi, loc, N, S, M, L, Lq, P, spi, bpi, magic_ml, magic_l, key_cap,
                         (uint32_t)r, kcap, planar_nq, counts, desc, entries, grad_loc, grad_aw);
    if (int rc = check_launch(who)) return rc;
    uint32_t *work = ctrl + binned::CTRL_WORK + r;
    if constexpr (!GV16) {
      if (planar) {                                      // (fp32 grad_value only: the fused boundary's)
        hipLaunchKernelGGL((binned::msda_bwd_binned_kernel<VT, float, true>), dim3(2 * cus), dim3(binned::NT), 0, st,
                           grad_out, value, shapes, lsi, loc, aw, offs, entries, zero_word, work, ctrl + r,
                           key_cap, N, S, M, L, P, split ? 1 : 0, planar_nq, reinterpret_cast<float *>(grad_value),
                           split ? sums : (float *)nullptr, grad_loc, grad_aw);
        if (int rc = check_launch(who)) return rc;
        continue;
      }
    }
    if (split)
      // every tile of this group adds to, or stores into, the zero-filled fp32 tensor
      hipLaunchKernelGGL((binned::msda_bwd_binned_kernel<VT, float>), dim3(2 * cus), dim3(binned::NT), 0, st,
                         grad_out, value, shapes, lsi, loc, aw, offs, entries, zero_word, work, ctrl + r,
                         key_cap, N, S, M, L, P, 1, 0u, sums, sums, grad_loc, grad_aw);
    else
      hipLaunchKernelGGL((binned::msda_bwd_binned_kernel<VT, GT>), dim3(2 * cus), dim3(binned::NT), 0, st,
                         grad_out, value, shapes, lsi, loc, aw, offs, entries, zero_word, work, ctrl + r,
                         key_cap, N, S, M, L, P, 0, 0u, grad_value, (float *)nullptr, grad_loc, grad_aw);
    if (int rc = check_launch(who)) return rc;
  }
  if (split && GV16) {
    hipLaunchKernelGGL(binned::msda_bwd_round_kernel, dim3(2 * cus), dim3(256), 0, st, sums,
                       reinterpret_cast<bf16_t *>(grad_value), shapes, lsi, N, S, M, L, key_cap,
                       (uint64_t)(gv_elems / 4));
    return check_launch(who);
  }
  return MSDA_OK;
}

template <typename VT, typename GT>
int launch_backward_binned(const char *who, const VT *value, const int64_t *shapes,
                           const int64_t *lsi, const float *loc, const float *aw,
                           const VT *grad_out, int N, int S, int M, int L, int Lq, int P,
                           GT *grad_value, float *grad_loc, float *grad_aw, void *workspace,
                           size_t workspace_bytes, hipStream_t st, bool planar = false) {
  constexpr bool GV16 = sizeof(GT) == 2;
  const uintptr_t amask = GV16 ? 15u : 3u;             // bf16 grad_value: an fp32 staging tensor may sit behind the lists
  const int g = (workspace == nullptr || (reinterpret_cast<uintptr_t>(workspace) & amask))
                    ? 0 : binned_group(N, S, M, L, Lq, workspace_bytes);
  if (g == 0)
    return fail(MSDA_ERR_INVALID_ARGUMENT,
                "%s: needs a %d-byte aligned device workspace of at least %zu bytes (one image at a time; %zu for the "
                "whole batch at once, the fastest), got %zu", who, (int)amask + 1, binned_min_bytes(S, M, L, Lq),
                binned_workspace_bytes(N, S, M, L, Lq, P, GV16), workspace ? workspace_bytes : (size_t)0);
  if (planar && g != N)      // the scratch layout spans the batch; the fused boundary brings the whole batch's workspace
    return fail(MSDA_ERR_INVALID_ARGUMENT, "%s: the fused backward needs the workspace of the whole batch", who);
  const size_t pairs_img = (size_t)Lq * M, samples_img = pairs_img * L * P, rows_img = (size_t)S * M * 32;
  for (int n0 = 0; n0 < N; n0 += g) {
    const int ng = N - n0 < g ? N - n0 : g;
    if (int rc = launch_binned_group<VT, GT>(who, value + n0 * rows_img, shapes, lsi, loc + n0 * samples_img * 2,
                                             aw + n0 * samples_img, grad_out + n0 * pairs_img * 32, ng, S, M, L,
                                             Lq, P, grad_value + n0 * rows_img, grad_loc + n0 * samples_img * 2,
                                             grad_aw + n0 * samples_img, workspace, workspace_bytes, st, planar))
      return rc;
  }
  return MSDA_OK;
}

// Which D == 32 backward: binned (default) or, with MSDA_BWD_IMPL=generic, the atomic kernel
// (A/B measurements).  Read once per process.
int bwd_impl_choice() {
  static const int choice = [] {
    const char *e = getenv("MSDA_BWD_IMPL");
    return (e && strcmp(e, "generic") == 0) ? 0 : 2;
  }();
  return choice;
}

// Workspace from the device's stream-ordered pool when the caller brought none.
struct PoolWorkspace {
  void *ptr = nullptr;
  hipStream_t st;
  explicit PoolWorkspace(hipStream_t s) : st(s) {}
  int get(const char *who, size_t bytes) {
    const hipError_t e = hipMallocAsync(&ptr, bytes, st);
    if (e != hipSuccess) {
      ptr = nullptr;
      return fail(MSDA_ERR_LAUNCH, "%s: hipMallocAsync(%zu) failed: %s", who, bytes, hipGetErrorString(e));
    }
    return MSDA_OK;
  }
  ~PoolWorkspace() {
    if (ptr) (void)hipFreeAsync(ptr, st);
  }
};

template <typename VT, typename GT>
int backward_d32(const char *who, const VT *value, const int64_t *shapes, const int64_t *lsi,
                 const float *loc, const float *aw, const VT *grad_out, int N, int S, int M, int L,
                 int Lq, int P, GT *grad_value, float *grad_loc, float *grad_aw, void *workspace,
                 size_t workspace_bytes, bool own_workspace, hipStream_t st) {
  if (!own_workspace)
    return launch_backward_binned<VT, GT>(who, value, shapes, lsi, loc, aw, grad_out, N, S, M, L, Lq, P,
                                          grad_value, grad_loc, grad_aw, workspace, workspace_bytes, st);
  PoolWorkspace pool(st);
  const size_t need = binned_workspace_bytes(N, S, M, L, Lq, P, sizeof(GT) == 2);
  if (int rc = pool.get(who, need)) return rc;
  return launch_backward_binned<VT, GT>(who, value, shapes, lsi, loc, aw, grad_out, N, S, M, L, Lq, P,
                                        grad_value, grad_loc, grad_aw, pool.ptr, need, st);
}

int backward_f32(const char *who, const float *value, const int64_t *shapes, const int64_t *lsi,
                 const float *loc, const float *aw, const float *grad_out, int N, int S, int M, int D,
                 int L, int Lq, int P, float *grad_value, float *grad_loc, float *grad_aw,
                 void *workspace, size_t workspace_bytes, bool own_workspace, void *stream) {
  if (int rc = check_dims(who, N, S, M, D, L, Lq, P)) return rc;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (binned_shape(N, S, M, D, L, Lq, P) && aligned16(value) && aligned16(grad_out) &&
      aligned16(grad_value) && aligned16(loc) && aligned16(grad_loc) && bwd_impl_choice() != 0)
    return backward_d32<float, float>(who, value, shapes, lsi, loc, aw, grad_out, N, S, M, L, Lq, P,
                                      grad_value, grad_loc, grad_aw, workspace, workspace_bytes,
                                      own_workspace, st);
  return backward_atomic<float>(who, value, shapes, lsi, loc, aw, grad_out, N, S, M, D, L, Lq, P,
                                grad_value, grad_loc, grad_aw, st);
}

// ---- one entry point that allocates nothing (include/msda_hip.h: msda_backward_f32_noalloc) ----
// The sort-by-tile workspace of the binned backward lives in a file-scope device array instead of
// coming from an allocator.  Each GPU the library runs on holds its own copy of the array at its own
// address; host code never uses the array's NAME as a pointer (that is a host-side stand-in, not a
// device address) -- it asks hipGetSymbolAddress once per device, with that device current, and
// passes the returned pointer to the kernels like any other workspace.
constexpr size_t NOALLOC_SCRATCH_BYTES = (size_t)64 << 20;
static __device__ __align__(256) unsigned char g_noalloc_scratch[NOALLOC_SCRATCH_BYTES];

constexpr int NOALLOC_MAX_DEVICES = 64;
std::atomic<void *> g_noalloc_ptr[NOALLOC_MAX_DEVICES];

// Device address of the scratch array on the current device (looked up on the first call per device).
int noalloc_scratch(const char *who, void **out) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return fail(MSDA_ERR_LAUNCH, "%s: hipGetDevice failed: %s", who, hipGetErrorString(e));
  if (dev < 0 || dev >= NOALLOC_MAX_DEVICES)
    return fail(MSDA_ERR_INVALID_ARGUMENT, "%s: device ordinal %d is beyond the %d this entry point tracks", who,
                dev, NOALLOC_MAX_DEVICES);
  void *p = g_noalloc_ptr[dev].load(std::memory_order_acquire);
  if (p == nullptr) {
    e = hipGetSymbolAddress(&p, HIP_SYMBOL(g_noalloc_scratch));
    if (e != hipSuccess || p == nullptr)
      return fail(MSDA_ERR_LAUNCH, "%s: hipGetSymbolAddress(scratch) failed on device %d: %s", who, dev,
                  hipGetErrorString(e));
    g_noalloc_ptr[dev].store(p, std::memory_order_release);   // racing first calls store the same address
  }
  *out = p;
  return MSDA_OK;
}

int backward_f32_noalloc(const char *who, const float *value, const int64_t *shapes, const int64_t *lsi,
                         const float *loc, const float *aw, const float *grad_out, int N, int S, int M, int D,
                         int L, int Lq, int P, float *grad_value, float *grad_loc, float *grad_aw, void *stream) {
  if (int rc = check_dims(who, N, S, M, D, L, Lq, P)) return rc;
  const bool fast = binned_shape(N, S, M, D, L, Lq, P) && aligned16(value) && aligned16(grad_out) &&
                    aligned16(grad_value) && aligned16(loc) && aligned16(grad_loc) && bwd_impl_choice() != 0;
  if (!fast)   // the generic kernels use no scratch buffer at all (one hipMemsetAsync, no allocation)
    return backward_atomic<float>(who, value, shapes, lsi, loc, aw, grad_out, N, S, M, D, L, Lq, P, grad_value,
                                  grad_loc, grad_aw, static_cast<hipStream_t>(stream));
  // the batch goes through the built-in array in groups of as many images as fit (images are independent)
  const size_t need = binned_min_bytes(S, M, L, Lq);
  if (need > NOALLOC_SCRATCH_BYTES)
    return fail(MSDA_ERR_INVALID_ARGUMENT,
                "%s: one image of this problem needs at least %zu bytes of sort workspace, more than the %zu-byte "
                "built-in scratch buffer; use msda_backward_f32_ws with a caller-provided workspace", who, need,
                NOALLOC_SCRATCH_BYTES);
  void *scratch = nullptr;
  if (int rc = noalloc_scratch(who, &scratch)) return rc;
  return launch_backward_binned<float, float>(who, value, shapes, lsi, loc, aw, grad_out, N, S, M, L, Lq, P,
                                              grad_value, grad_loc, grad_aw, scratch, NOALLOC_SCRATCH_BYTES,
                                              static_cast<hipStream_t>(stream));
}

// `grad_value` is fp32, or (gv16) bfloat16 bit patterns
int backward_bf16(const char *who, const uint16_t *value, const int64_t *shapes, const int64_t *lsi,
                  const float *loc, const float *aw, const uint16_t *grad_out, int N, int S, int M,
                  int D, int L, int Lq, int P, void *grad_value, bool gv16, float *grad_loc, float *grad_aw,
                  void *workspace, size_t workspace_bytes, bool own_workspace, void *stream) {
  if (int rc = check_dims(who, N, S, M, D, L, Lq, P)) return rc;
  if (!binned_shape(N, S, M, D, L, Lq, P) || (reinterpret_cast<uintptr_t>(value) & 7u) ||
      (reinterpret_cast<uintptr_t>(grad_out) & 7u) || (reinterpret_cast<uintptr_t>(grad_value) & (gv16 ? 7u : 15u)) ||
      !aligned16(loc) || !aligned16(grad_loc))
    return fail(MSDA_ERR_INVALID_ARGUMENT,
                "%s: needs channels == 32, num_point <= %d, num_levels <= %d, a non-empty batch and "
                "aligned tensors of < 2^30 elements",
                who, binned::MAXP, binned::MAXL);
  const bf16_t *v = reinterpret_cast<const bf16_t *>(value), *go = reinterpret_cast<const bf16_t *>(grad_out);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (gv16)
    return backward_d32<bf16_t, bf16_t>(who, v, shapes, lsi, loc, aw, go, N, S, M, L, Lq, P,
                                        static_cast<bf16_t *>(grad_value), grad_loc, grad_aw, workspace,
                                        workspace_bytes, own_workspace, st);
  return backward_d32<bf16_t, float>(who, v, shapes, lsi, loc, aw, go, N, S, M, L, Lq, P,
                                     static_cast<float *>(grad_value), grad_loc, grad_aw, workspace,
                                     workspace_bytes, own_workspace, st);
}


// ---- row f3: fused boundary (fp32) ----

template <typename VT>
int fused_forward(const char *who, const VT *value, const int64_t *shapes, const int64_t *lsi, const float *refp,
                  const float *off, const float *logits, int N, int S, int M, int D, int L, int Lq,
                  int P, VT *out, void *stream) {
  if (int rc = check_dims(who, N, S, M, D, L, Lq, P)) return rc;
  if ((size_t)N * Lq == 0) return MSDA_OK;
  const size_t npairs = (size_t)N * Lq * M, limit = (size_t)1 << 30;
  const bool small = (size_t)N * S * M * D < limit && npairs * L * P * 2 < limit && npairs * D < limit;
  if (D != 32 || L * P > FWD_FUSED_MAX_LP || !small || !aligned16(value) || !aligned16(out) ||
      (reinterpret_cast<uintptr_t>(off) & 7u) || (reinterpret_cast<uintptr_t>(refp) & 7u))
    return fail(MSDA_ERR_INVALID_ARGUMENT,
                "%s: needs channels == 32, num_levels * num_point <= %d, aligned tensors of < 2^30 "
                "elements (use the unfused entry points otherwise)", who, FWD_FUSED_MAX_LP);
  if (S == 0) return zero_output(who, out, npairs * 32 * sizeof(VT), static_cast<hipStream_t>(stream));
  constexpr uint32_t GP = 64u / FwdLanes<VT>::LPG;
  const size_t lds = (size_t)4 * ((size_t)2 * GP * (size_t)(L * P + 1) * 16 + (((size_t)GP * L * P + 3) / 4) * 16);
  const uint32_t nq = (uint32_t)((size_t)N * Lq), nunits = ((nq + GP - 1u) / GP) * (uint32_t)M;
  const uint32_t nblk = (nunits + 3u) / 4u;
  hipLaunchKernelGGL((msda_fwd_d32_qh_kernel<VT, true>), dim3(nblk), dim3(256), lds,
                     static_cast<hipStream_t>(stream), value, shapes, lsi, off, logits, refp, S, M, L, Lq, P,
                     nq, nunits, nblk, false, fwd_div_magic((uint32_t)(L * P)), fwd_div_magic((uint32_t)P),
                     fwd_div_magic32((uint32_t)M, nunits), fwd_div_magic_lq((uint32_t)Lq, nq, GP), fwd_region_log2(), out);
  return check_launch(who);
}

size_t fused_backward_workspace_bytes(int N, int S, int M, int D, int L, int Lq, int P) {
  const size_t samples = (size_t)N * Lq * M * L * P;
  size_t inner = 0;
  if (binned_shape(N, S, M, D, L, Lq, P) && bwd_impl_choice() != 0)
    inner = binned_workspace_bytes(N, S, M, L, Lq, P);
  // + the backward's own layout of grad_loc / grad_aw (binned.inc, PLANAR), which the epilogue kernel reads back
  const size_t scratch = inner ? align256(samples * 8) + align256(samples * 4) : 0;
  return align256(samples * 8) + align256(samples * 4) + scratch + align256(inner);
}

// `value` / `grad_out` are fp32 or (bf16 == true) bfloat16 bit patterns
int fused_backward(const char *who, bool bf16, const void *value, const int64_t *shapes, const int64_t *lsi, const float *refp,
                       const float *off, const float *logits, const void *grad_out, int N, int S, int M,
                       int D, int L, int Lq, int P, float *grad_value, float *grad_off, float *grad_logits,
                       void *workspace, size_t workspace_bytes, void *stream) {
  if (int rc = check_dims(who, N, S, M, D, L, Lq, P)) return rc;
  const size_t need = fused_backward_workspace_bytes(N, S, M, D, L, Lq, P);
  if (workspace == nullptr || workspace_bytes < need || (reinterpret_cast<uintptr_t>(workspace) & 255u))
    return fail(MSDA_ERR_INVALID_ARGUMENT,
                "%s: needs a 256-byte aligned device workspace of at least %zu bytes (got %zu)", who, need,
                workspace ? workspace_bytes : (size_t)0);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const size_t npairs = (size_t)N * Lq * M, samples = npairs * L * P;
  float *loc = static_cast<float *>(workspace);
  float *aw = reinterpret_cast<float *>(static_cast<char *>(workspace) + align256(samples * 8));
  char *rest = static_cast<char *>(workspace) + align256(samples * 8) + align256(samples * 4);
  // The binned kernels leave grad_loc / grad_aw in their own layout, in scratch, and the epilogue reads them from
  // there: an item of the tile kernel is one (level, head), and in the caller's [N, Lq, M, L, P] order its results
  // are 16 bytes out of every 384 -- 0.4 ms of partial-line write-backs at bs 32 / 1024^2 (experiments README, 7).
  const bool planar = binned_shape(N, S, M, D, L, Lq, P) && bwd_impl_choice() != 0 && M <= FUSED_PB &&   // (the epilogue takes all heads of a query in one workgroup)
                      (bf16 ? (reinterpret_cast<uintptr_t>(value) & 7u) == 0 && (reinterpret_cast<uintptr_t>(grad_out) & 7u) == 0
                            : aligned16(value) && aligned16(grad_out)) && aligned16(grad_value);
  float *gl_scratch = reinterpret_cast<float *>(rest), *ga_scratch = reinterpret_cast<float *>(rest + align256(samples * 8));
  const size_t scratch_bytes = binned_shape(N, S, M, D, L, Lq, P) && bwd_impl_choice() != 0
                                   ? align256(samples * 8) + align256(samples * 4) : 0;
  void *inner = rest + scratch_bytes;
  const size_t inner_bytes = need - align256(samples * 8) - align256(samples * 4) - scratch_bytes;
  const size_t nblk = (npairs + FUSED_PB - 1) / FUSED_PB;
  if (nblk > 0x7fffffffull) return fail(MSDA_ERR_INVALID_ARGUMENT, "%s: problem too large", who);
  if (L * P > FUSED_MAX_LP)
    return fail(MSDA_ERR_INVALID_ARGUMENT, "%s: num_levels * num_point must be <= %d", who, FUSED_MAX_LP);
  if (npairs) {
    hipLaunchKernelGGL(msda_fused_prologue_kernel, dim3((unsigned)nblk), dim3(256), 0, st, shapes, refp, off,
                       logits, M, L, P, npairs, loc, aw);
    if (int rc = check_launch(who)) return rc;
  }
  int rc;
  if (planar)
    rc = bf16 ? launch_backward_binned<bf16_t, float>(who, static_cast<const bf16_t *>(value), shapes, lsi, loc, aw,
                                                      static_cast<const bf16_t *>(grad_out), N, S, M, L, Lq, P,
                                                      grad_value, gl_scratch, ga_scratch, inner, inner_bytes, st, true)
              : launch_backward_binned<float, float>(who, static_cast<const float *>(value), shapes, lsi, loc, aw,
                                                     static_cast<const float *>(grad_out), N, S, M, L, Lq, P,
                                                     grad_value, gl_scratch, ga_scratch, inner, inner_bytes, st, true);
  else
    rc = bf16 ? backward_bf16(who, static_cast<const uint16_t *>(value), shapes, lsi, loc, aw,
                              static_cast<const uint16_t *>(grad_out), N, S, M, D, L, Lq, P, grad_value, false,
                              grad_off, grad_logits, inner, inner_bytes, false, stream)
              : backward_f32(who, static_cast<const float *>(value), shapes, lsi, loc, aw,
                             static_cast<const float *>(grad_out), N, S, M, D, L, Lq, P, grad_value,
                             grad_off, grad_logits, inner, inner_bytes, false, stream);
  if (rc) return rc;
  if (npairs && planar) {
    const uint32_t NQ = (uint32_t)((size_t)N * Lq), qb = (uint32_t)(FUSED_PB / M > 0 ? FUSED_PB / M : 1);
    const size_t lds = (size_t)qb * M * L * P * 3 * sizeof(float);
    hipLaunchKernelGGL(msda_fused_epilogue_planar_kernel, dim3((NQ + qb - 1) / qb), dim3(256), lds, st, shapes, aw, M, L,
                       P, NQ, qb, gl_scratch, ga_scratch, grad_off, grad_logits);
    if (int rc2 = check_launch(who)) return rc2;
  } else if (npairs) {
    hipLaunchKernelGGL(msda_fused_epilogue_kernel, dim3((unsigned)nblk), dim3(256), 0, st, shapes, aw, L, P,
                       npairs, grad_off, grad_logits);
    if (int rc2 = check_launch(who)) return rc2;
  }
  return MSDA_OK;
}

}  // namespace

extern "C" {

int msda_fused_forward_f32(const float *value, const int64_t *spatial_shapes,
                           const int64_t *level_start_index, const float *reference_points,
                           const float *sampling_offsets, const float *attn_logits, int batch,
                           int spatial_size, int num_heads, int channels, int num_levels, int num_query,
                           int num_point, float *output, void *stream) {
  return fused_forward<float>("msda_fused_forward_f32", value, spatial_shapes, level_start_index,
                              reference_points, sampling_offsets, attn_logits, batch, spatial_size, num_heads,
                              channels, num_levels, num_query, num_point, output, stream);
}

int msda_fused_forward_bf16(const uint16_t *value, const int64_t *spatial_shapes,
                            const int64_t *level_start_index, const float *reference_points,
                            const float *sampling_offsets, const float *attn_logits, int batch,
                            int spatial_size, int num_heads, int channels, int num_levels, int num_query,
                            int num_point, uint16_t *output, void *stream) {
  return fused_forward<bf16_t>("msda_fused_forward_bf16", reinterpret_cast<const bf16_t *>(value),
                               spatial_shapes, level_start_index, reference_points, sampling_offsets,
                               attn_logits, batch, spatial_size, num_heads, channels, num_levels, num_query,
                               num_point, reinterpret_cast<bf16_t *>(output), stream);
}

int msda_fused_backward_bf16(const uint16_t *value, const int64_t *spatial_shapes,
                             const int64_t *level_start_index, const float *reference_points,
                             const float *sampling_offsets, const float *attn_logits,
                             const uint16_t *grad_output, int batch, int spatial_size, int num_heads,
                             int channels, int num_levels, int num_query, int num_point, float *grad_value,
                             float *grad_sampling_offsets, float *grad_attn_logits, void *workspace,
                             size_t workspace_bytes, void *stream) {
  return fused_backward("msda_fused_backward_bf16", true, value, spatial_shapes, level_start_index,
                        reference_points, sampling_offsets, attn_logits, grad_output, batch, spatial_size,
                        num_heads, channels, num_levels, num_query, num_point, grad_value,
                        grad_sampling_offsets, grad_attn_logits, workspace, workspace_bytes, stream);
}

size_t msda_fused_backward_workspace_bytes(int batch, int spatial_size, int num_heads, int channels,
                                           int num_levels, int num_query, int num_point) {
  return fused_backward_workspace_bytes(batch, spatial_size, num_heads, channels, num_levels, num_query,
                                        num_point);
}

int msda_fused_backward_f32(const float *value, const int64_t *spatial_shapes,
                            const int64_t *level_start_index, const float *reference_points,
                            const float *sampling_offsets, const float *attn_logits,
                            const float *grad_output, int batch, int spatial_size, int num_heads,
                            int channels, int num_levels, int num_query, int num_point, float *grad_value,
                            float *grad_sampling_offsets, float *grad_attn_logits, void *workspace,
                            size_t workspace_bytes, void *stream) {
  return fused_backward("msda_fused_backward_f32", false, value, spatial_shapes, level_start_index,
                        reference_points, sampling_offsets, attn_logits, grad_output, batch, spatial_size,
                        num_heads, channels, num_levels, num_query, num_point, grad_value,
                        grad_sampling_offsets, grad_attn_logits, workspace, workspace_bytes, stream);
}

#ifdef MSDA_PHASE_STAMPS
// diagnostic build only: read (and clear) the per-phase cycle sums
int msda_debug_phase_cycles(unsigned long long *out16) {
  if (hipDeviceSynchronize() != hipSuccess) return 1;
  if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(binned::g_phase_cycles), 16 * sizeof(unsigned long long)) != hipSuccess) return 1;
  unsigned long long z[16] = {0};
  return hipMemcpyToSymbol(HIP_SYMBOL(binned::g_phase_cycles), z, sizeof(z)) == hipSuccess ? 0 : 1;
}
#endif

int msda_abi_version(void) { return 5; }

const char *msda_last_error(void) { return g_err; }

uint64_t msda_launch_count(void) { return g_launches.load(std::memory_order_relaxed); }

int msda_forward_f32(const float *value, const int64_t *spatial_shapes,
                     const int64_t *level_start_index, const float *sampling_loc,
                     const float *attn_weight, int batch, int spatial_size, int num_heads,
                     int channels, int num_levels, int num_query, int num_point, float *output,
                     void *stream) {
  return forward_impl<float>(value, spatial_shapes, level_start_index, sampling_loc, attn_weight,
                             batch, spatial_size, num_heads, channels, num_levels, num_query,
                             num_point, output, stream);
}

int msda_forward_f64(const double *value, const int64_t *spatial_shapes,
                     const int64_t *level_start_index, const double *sampling_loc,
                     const double *attn_weight, int batch, int spatial_size, int num_heads,
                     int channels, int num_levels, int num_query, int num_point, double *output,
                     void *stream) {
  return forward_impl<double>(value, spatial_shapes, level_start_index, sampling_loc,
                              attn_weight, batch, spatial_size, num_heads, channels, num_levels,
                              num_query, num_point, output, stream);
}

int msda_backward_f32(const float *value, const int64_t *spatial_shapes,
                      const int64_t *level_start_index, const float *sampling_loc,
                      const float *attn_weight, const float *grad_output, int batch,
                      int spatial_size, int num_heads, int channels, int num_levels,
                      int num_query, int num_point, float *grad_value, float *grad_sampling_loc,
                      float *grad_attn_weight, void *stream) {
  return backward_f32("msda_backward_f32", value, spatial_shapes, level_start_index, sampling_loc,
                      attn_weight, grad_output, batch, spatial_size, num_heads, channels, num_levels,
                      num_query, num_point, grad_value, grad_sampling_loc, grad_attn_weight, nullptr,
                      0, true, stream);
}

int msda_backward_f32_ws(const float *value, const int64_t *spatial_shapes,
                         const int64_t *level_start_index, const float *sampling_loc,
                         const float *attn_weight, const float *grad_output, int batch,
                         int spatial_size, int num_heads, int channels, int num_levels,
                         int num_query, int num_point, float *grad_value, float *grad_sampling_loc,
                         float *grad_attn_weight, void *workspace, size_t workspace_bytes,
                         void *stream) {
  return backward_f32("msda_backward_f32_ws", value, spatial_shapes, level_start_index, sampling_loc,
                      attn_weight, grad_output, batch, spatial_size, num_heads, channels, num_levels,
                      num_query, num_point, grad_value, grad_sampling_loc, grad_attn_weight,
                      workspace, workspace_bytes, false, stream);
}

int msda_backward_f32_noalloc(const float *value, const int64_t *spatial_shapes,
                              const int64_t *level_start_index, const float *sampling_loc,
                              const float *attn_weight, const float *grad_output, int batch,
                              int spatial_size, int num_heads, int channels, int num_levels,
                              int num_query, int num_point, float *grad_value, float *grad_sampling_loc,
                              float *grad_attn_weight, void *stream) {
  return backward_f32_noalloc("msda_backward_f32_noalloc", value, spatial_shapes, level_start_index,
                              sampling_loc, attn_weight, grad_output, batch, spatial_size, num_heads,
                              channels, num_levels, num_query, num_point, grad_value, grad_sampling_loc,
                              grad_attn_weight, stream);
}

size_t msda_backward_noalloc_max_workspace_bytes(void) { return NOALLOC_SCRATCH_BYTES; }

size_t msda_backward_workspace_bytes(int batch, int spatial_size, int num_heads, int channels,
                                     int num_levels, int num_query, int num_point) {
  if (!binned_shape(batch, spatial_size, num_heads, channels, num_levels, num_query, num_point) ||
      bwd_impl_choice() == 0)
    return 0;
  return binned_workspace_bytes(batch, spatial_size, num_heads, num_levels, num_query, num_point);
}

int msda_forward_bf16(const uint16_t *value, const int64_t *spatial_shapes,
                      const int64_t *level_start_index, const float *sampling_loc,
                      const float *attn_weight, int batch, int spatial_size, int num_heads,
                      int channels, int num_levels, int num_query, int num_point, uint16_t *output,
                      void *stream) {
  const int N = batch, S = spatial_size, M = num_heads, D = channels, L = num_levels,
            Lq = num_query, P = num_point;
  if (int rc = check_dims("msda_forward_bf16", N, S, M, D, L, Lq, P)) return rc;
  if ((size_t)N * Lq == 0) return MSDA_OK;
  const size_t npairs = (size_t)N * Lq * M, limit = (size_t)1 << 30;
  const bool small = (size_t)N * S * M * D < limit && npairs * L * P * 2 < limit && npairs * D < limit;
  if (D != 32 || !small || (reinterpret_cast<uintptr_t>(value) & 7u) ||
      (reinterpret_cast<uintptr_t>(output) & 7u))
    return fail(MSDA_ERR_INVALID_ARGUMENT,
                "msda_forward_bf16: needs channels == 32, 8-byte aligned tensors of < 2^30 elements");
  if (S == 0) return zero_output("msda_forward_bf16", output, npairs * 32 * sizeof(uint16_t), static_cast<hipStream_t>(stream));
  launch_forward_d32<bf16_t>(reinterpret_cast<const bf16_t *>(value), spatial_shapes,
                             level_start_index, sampling_loc, attn_weight, S, M, L, Lq, P, npairs,
                             reinterpret_cast<bf16_t *>(output), static_cast<hipStream_t>(stream));
  return check_launch("msda_forward_bf16");
}

int msda_backward_bf16(const uint16_t *value, const int64_t *spatial_shapes,
                       const int64_t *level_start_index, const float *sampling_loc,
                       const float *attn_weight, const uint16_t *grad_output, int batch,
                       int spatial_size, int num_heads, int channels, int num_levels,
                       int num_query, int num_point, float *grad_value, float *grad_sampling_loc,
                       float *grad_attn_weight, void *stream) {
  return backward_bf16("msda_backward_bf16", value, spatial_shapes, level_start_index, sampling_loc,
                       attn_weight, grad_output, batch, spatial_size, num_heads, channels,
                       num_levels, num_query, num_point, grad_value, false, grad_sampling_loc,
                       grad_attn_weight, nullptr, 0, true, stream);
}

int msda_backward_bf16_ws(const uint16_t *value, const int64_t *spatial_shapes,
                          const int64_t *level_start_index, const float *sampling_loc,
                          const float *attn_weight, const uint16_t *grad_output, int batch,
                          int spatial_size, int num_heads, int channels, int num_levels,
                          int num_query, int num_point, float *grad_value, float *grad_sampling_loc,
                          float *grad_attn_weight, void *workspace, size_t workspace_bytes,
                          void *stream) {
  return backward_bf16("msda_backward_bf16_ws", value, spatial_shapes, level_start_index,
                       sampling_loc, attn_weight, grad_output, batch, spatial_size, num_heads,
                       channels, num_levels, num_query, num_point, grad_value, false, grad_sampling_loc,
                       grad_attn_weight, workspace, workspace_bytes, false, stream);
}

int msda_backward_bf16_gv16_ws(const uint16_t *value, const int64_t *spatial_shapes,
                               const int64_t *level_start_index, const float *sampling_loc,
                               const float *attn_weight, const uint16_t *grad_output, int batch,
                               int spatial_size, int num_heads, int channels, int num_levels,
                               int num_query, int num_point, uint16_t *grad_value, float *grad_sampling_loc,
                               float *grad_attn_weight, void *workspace, size_t workspace_bytes,
                               void *stream) {
  return backward_bf16("msda_backward_bf16_gv16_ws", value, spatial_shapes, level_start_index,
                       sampling_loc, attn_weight, grad_output, batch, spatial_size, num_heads,
                       channels, num_levels, num_query, num_point, grad_value, true, grad_sampling_loc,
                       grad_attn_weight, workspace, workspace_bytes, false, stream);
}

size_t msda_backward_bf16_gv16_workspace_bytes(int batch, int spatial_size, int num_heads, int channels,
                                               int num_levels, int num_query, int num_point) {
  if (!binned_shape(batch, spatial_size, num_heads, channels, num_levels, num_query, num_point)) return 0;
  return binned_workspace_bytes(batch, spatial_size, num_heads, num_levels, num_query, num_point, true);
}

size_t msda_backward_compact_workspace_bytes(int batch, int spatial_size, int num_heads, int channels,
                                             int num_levels, int num_query, int num_point, int bf16_grad_value) {
  if (!binned_shape(batch, spatial_size, num_heads, channels, num_levels, num_query, num_point) ||
      bwd_impl_choice() == 0)
    return 0;
  return binned_compact_bytes(batch, spatial_size, num_heads, num_levels, num_query, bf16_grad_value != 0);
}

size_t msda_backward_min_workspace_bytes(int batch, int spatial_size, int num_heads, int channels,
                                         int num_levels, int num_query, int num_point) {
  if (!binned_shape(batch, spatial_size, num_heads, channels, num_levels, num_query, num_point) ||
      bwd_impl_choice() == 0)
    return 0;
  // one image at a time, one entry per quad (sixteen rounds); anything more is used
  return binned_min_bytes(spatial_size, num_heads, num_levels, num_query);
}

int msda_backward_f64(const double *value, const int64_t *spatial_shapes,
                      const int64_t *level_start_index, const double *sampling_loc,
                      const double *attn_weight, const double *grad_output, int batch,
                      int spatial_size, int num_heads, int channels, int num_levels,
                      int num_query, int num_point, double *grad_value,
                      double *grad_sampling_loc, double *grad_attn_weight, void *stream) {
  return backward_impl<double>("msda_backward_f64", value, spatial_shapes, level_start_index,
                               sampling_loc, attn_weight, grad_output, batch, spatial_size,
                               num_heads, channels, num_levels, num_query, num_point, grad_value,
                               grad_sampling_loc, grad_attn_weight, stream);
}

}  // extern "C"
