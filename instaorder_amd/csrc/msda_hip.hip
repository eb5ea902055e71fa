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
#include <string.h>

#include <atomic>

#include "msda_hip.h"

namespace {

thread_local char g_err[512] = "";
std::atomic<uint64_t> g_launches{0};

int fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
int fail(int code, const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

// ---------------------------------------------------------------------------------------
// XCD-aware block remap.  Workgroups are dealt round-robin over the 8 XCDs (each with its
// own 4 MiB L2); give every XCD one contiguous run of logical blocks so that neighbouring
// queries -- which sample neighbouring pixel rows -- share an L2.  Bijective for any grid.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t xcd_remap(uint32_t b, uint32_t nblocks) {
  const uint32_t q = nblocks >> 3, r = nblocks & 7u;
  const uint32_t xcd = b & 7u, idx = b >> 3;
  const uint32_t start = xcd < r ? xcd * (q + 1u) : r * (q + 1u) + (xcd - r) * q;
  return start + idx;
}

// ---------------------------------------------------------------------------------------
// Per-sample geometry shared by forward and backward (reference
// ms_deform_im2col_cuda.cuh:38-59 and :290-293): pixel coordinates, the range gate, the
// four corner weights (zeroed for corners outside the level = zero padding) and clamped
// corner indices, so every address formed from them is inside the level.
// ---------------------------------------------------------------------------------------
template <typename T>
struct Corner4 {
  T k1, k2, k3, k4;     // hh*hw, hh*lw, lh*hw, lh*lw, already 0 for out-of-level corners
  T lh, lw, hh, hw;
  int o1, o2, o3, o4;   // pixel index (h*W + w) of each corner, clamped into the level
  bool ok1, ok2, ok3, ok4, in;
};

template <typename T>
__device__ __forceinline__ Corner4<T> corner_setup(T lx, T ly, int H, int W) {
  Corner4<T> c;
  const T fH = (T)H, fW = (T)W;
  const T h_im = ly * fH - (T)0.5;
  const T w_im = lx * fW - (T)0.5;
  c.in = (h_im > (T)-1) && (w_im > (T)-1) && (h_im < fH) && (w_im < fW);
  // Outside the gate (or NaN) every weight is zero; park the coordinates at 0 so the
  // float->int conversion below is always defined.
  const T hs = c.in ? h_im : (T)0, ws = c.in ? w_im : (T)0;
  const T hf = floor(hs), wf = floor(ws);
  c.lh = hs - hf;
  c.lw = ws - wf;
  c.hh = (T)1 - c.lh;
  c.hw = (T)1 - c.lw;
  const int h0 = (int)hf, w0 = (int)wf, h1 = h0 + 1, w1 = w0 + 1;
  const bool h0ok = h0 >= 0, w0ok = w0 >= 0, h1ok = h1 <= H - 1, w1ok = w1 <= W - 1;
  c.ok1 = c.in && h0ok && w0ok;
  c.ok2 = c.in && h0ok && w1ok;
  c.ok3 = c.in && h1ok && w0ok;
  c.ok4 = c.in && h1ok && w1ok;
  c.k1 = c.ok1 ? c.hh * c.hw : (T)0;
  c.k2 = c.ok2 ? c.hh * c.lw : (T)0;
  c.k3 = c.ok3 ? c.lh * c.hw : (T)0;
  c.k4 = c.ok4 ? c.lh * c.lw : (T)0;
  const int h0c = max(h0, 0), w0c = max(w0, 0), h1c = min(h1, H - 1), w1c = min(w1, W - 1);
  c.o1 = h0c * W + w0c;
  c.o2 = h0c * W + w1c;
  c.o3 = h1c * W + w0c;
  c.o4 = h1c * W + w1c;
  return c;
}

// =======================================================================================
// Forward, fp32, D == 32.
// Replaces ms_deformable_im2col_gpu_kernel<float> (ms_deform_im2col_cuda.cuh:242-304).
//
// Lane mapping: 8 lanes own one pair, each lane 4 consecutive channels (one 16-B load
// covers a lane's share of a 128-B pixel-head row), so a wave handles 8 consecutive pairs
// -- with M = 8 that is all heads of one query -- and stores 1 KiB of contiguous output.
// Every corner fetch is a dwordx4 load; no cross-lane traffic is needed at all because a
// lane carries its 4 channels through all L*P samples.  Sampling locations and weights of
// a (pair, level) are fetched as three 16-B loads (P == 4) that the 8 lanes of a pair
// share through the L1 broadcast.
// All offsets are 32-bit byte offsets from the (wave-uniform) tensor base, which lets the
// loads use the SGPR-base + VGPR-offset form (the host checks every tensor is < 4 GiB).
// =======================================================================================
__device__ __forceinline__ float4 ld16(const float *__restrict__ base, uint32_t byte_off) {
  return *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(base) + byte_off);
}

// NS samples of one level: all 4*NS corner rows are requested before the first is used.
template <int NS>
__device__ __forceinline__ void fwd_samples_d32(const float *__restrict__ value, uint32_t lbyte,
                                                uint32_t row_bytes, int H, int W,
                                                const float (&lx)[NS], const float (&ly)[NS],
                                                const float (&a)[NS], float4 &acc) {
  float k[NS][4];
  float4 v[NS][4];
#pragma unroll
  for (int i = 0; i < NS; ++i) {
    const Corner4<float> c = corner_setup<float>(lx[i], ly[i], H, W);
    k[i][0] = c.k1 * a[i];
    k[i][1] = c.k2 * a[i];
    k[i][2] = c.k3 * a[i];
    k[i][3] = c.k4 * a[i];
    v[i][0] = ld16(value, lbyte + (uint32_t)c.o1 * row_bytes);
    v[i][1] = ld16(value, lbyte + (uint32_t)c.o2 * row_bytes);
    v[i][2] = ld16(value, lbyte + (uint32_t)c.o3 * row_bytes);
    v[i][3] = ld16(value, lbyte + (uint32_t)c.o4 * row_bytes);
  }
#pragma unroll
  for (int i = 0; i < NS; ++i) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      acc.x += k[i][j] * v[i][j].x;
      acc.y += k[i][j] * v[i][j].y;
      acc.z += k[i][j] * v[i][j].z;
      acc.w += k[i][j] * v[i][j].w;
    }
  }
}

template <bool P4>
__global__ __launch_bounds__(256) void msda_fwd_f32_d32_kernel(
    const float *__restrict__ value, const int64_t *__restrict__ shapes,
    const int64_t *__restrict__ lsi, const float *__restrict__ loc,
    const float *__restrict__ aw, int S, int M, int L, int Lq, int P, uint32_t npairs,
    uint32_t nblocks, float *__restrict__ out) {
  const uint32_t blk = xcd_remap(blockIdx.x, nblocks);
  const uint32_t pair = blk * 32u + (threadIdx.x >> 3);
  if (pair >= npairs) return;
  const uint32_t sub = threadIdx.x & 7u;
  const uint32_t m = pair % (uint32_t)M;
  const uint32_t n = (pair / (uint32_t)M) / (uint32_t)Lq;
  const uint32_t row_bytes = (uint32_t)M * 128u;
  const uint32_t vbyte = n * (uint32_t)S * row_bytes + m * 128u + sub * 16u;
  const uint32_t sbase = pair * (uint32_t)(L * P);
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int l = 0; l < L; ++l) {
    const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
    const uint32_t lbyte = vbyte + (uint32_t)lsi[l] * row_bytes;
    if (P4) {
      const uint32_t s0 = sbase + (uint32_t)l * 4u;
      const float4 xa = ld16(loc, s0 * 8u);
      const float4 xb = ld16(loc, s0 * 8u + 16u);
      const float4 a4 = ld16(aw, s0 * 4u);
      const float lx[4] = {xa.x, xa.z, xb.x, xb.z};
      const float ly[4] = {xa.y, xa.w, xb.y, xb.w};
      const float a[4] = {a4.x, a4.y, a4.z, a4.w};
      fwd_samples_d32<4>(value, lbyte, row_bytes, H, W, lx, ly, a, acc);
    } else {
      for (int p = 0; p < P; ++p) {
        const uint32_t s = sbase + (uint32_t)(l * P + p);
        const float lx[1] = {loc[2u * s]}, ly[1] = {loc[2u * s + 1u]}, a[1] = {aw[s]};
        fwd_samples_d32<1>(value, lbyte, row_bytes, H, W, lx, ly, a, acc);
      }
    }
  }
  *reinterpret_cast<float4 *>(reinterpret_cast<char *>(out) + (pair * 128u + sub * 16u)) = acc;
}

// =======================================================================================
// Forward, any D, fp32 / fp64.  G lanes (a power of two <= 64) own one pair and stride
// over its channels.  Same reference kernel as above; this is the path ops/test.py's
// D in {30, 64, 71, 1025, 2048, 3096} and every fp64 call take.  64-bit indexing.
// =======================================================================================
template <typename T, int G>
__global__ __launch_bounds__(256) void msda_fwd_generic_kernel(
    const T *__restrict__ value, const int64_t *__restrict__ shapes,
    const int64_t *__restrict__ lsi, const T *__restrict__ loc, const T *__restrict__ aw, int S,
    int M, int D, int L, int Lq, int P, size_t npairs, T *__restrict__ out) {
  const size_t pair = (size_t)blockIdx.x * (256 / G) + threadIdx.x / G;
  if (pair >= npairs) return;
  const int sub = threadIdx.x % G;
  const int m = (int)(pair % (size_t)M);
  const size_t n = (pair / (size_t)M) / (size_t)Lq;
  const size_t row = (size_t)M * D;
  const T *vb = value + n * (size_t)S * row + (size_t)m * D;
  const T *locp = loc + pair * (size_t)(L * P) * 2;
  const T *awp = aw + pair * (size_t)(L * P);
  for (int c0 = sub; c0 < D; c0 += G) {
    T acc = 0;
    for (int l = 0; l < L; ++l) {
      const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
      const T *vl = vb + (size_t)lsi[l] * row + c0;
      for (int p = 0; p < P; ++p) {
        const int s = l * P + p;
        const Corner4<T> c = corner_setup<T>(locp[2 * s], locp[2 * s + 1], H, W);
        const T v1 = vl[(size_t)c.o1 * row], v2 = vl[(size_t)c.o2 * row];
        const T v3 = vl[(size_t)c.o3 * row], v4 = vl[(size_t)c.o4 * row];
        acc += (c.k1 * v1 + c.k2 * v2 + c.k3 * v3 + c.k4 * v4) * awp[s];
      }
    }
    out[pair * (size_t)D + c0] = acc;
  }
}

// =======================================================================================
// Backward, any D, fp32 / fp64.
// Replaces every ms_deformable_col2im_gpu_kernel_* variant
// (ms_deform_im2col_cuda.cuh:306-925) and ms_deform_attn_col2im_bilinear (:92-164).
//
// G lanes own one pair; lane j walks channels j, j+G, ...  Per sample each lane
//   - scatters k_corner * grad_out * attn into grad_value with hardware float atomics
//     (lanes of a pair hit consecutive dwords of one pixel-head row, so with G = 32 one
//     wave-instruction is two full 128-B row segments),
//   - accumulates its share of d/d attn, d/d x, d/d y, which are then summed over the G
//     lanes with xor-shuffles (the reference does a thread-0 serial sum through shared
//     memory behind two __syncthreads, :381-399) and stored by lane 0 of the group.
// =======================================================================================
template <typename T, int G>
__device__ __forceinline__ T group_sum(T v) {
#pragma unroll
  for (int off = G / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, G);
  return v;
}

template <typename T, int G>
__global__ __launch_bounds__(256) void msda_bwd_generic_kernel(
    const T *__restrict__ grad_out, const T *__restrict__ value,
    const int64_t *__restrict__ shapes, const int64_t *__restrict__ lsi,
    const T *__restrict__ loc, const T *__restrict__ aw, int S, int M, int D, int L, int Lq,
    int P, size_t npairs, T *__restrict__ grad_value, T *__restrict__ grad_loc,
    T *__restrict__ grad_aw) {
  size_t pair = (size_t)blockIdx.x * (256 / G) + threadIdx.x / G;
  // keep whole waves alive for the shuffles; surplus groups redo the last pair and skip stores
  const bool live = pair < npairs;
  if (!live) pair = npairs - 1;
  const int sub = threadIdx.x % G;
  const int m = (int)(pair % (size_t)M);
  const size_t n = (pair / (size_t)M) / (size_t)Lq;
  const size_t row = (size_t)M * D;
  const size_t nbase = n * (size_t)S * row + (size_t)m * D;
  const T *go = grad_out + pair * (size_t)D;
  const size_t sbase = pair * (size_t)(L * P);
  for (int l = 0; l < L; ++l) {
    const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
    const size_t lbase = nbase + (size_t)lsi[l] * row;
    const T *vl = value + lbase;
    T *gl = grad_value + lbase;
    for (int p = 0; p < P; ++p) {
      const size_t s = sbase + (size_t)(l * P + p);
      const T a = aw[s];
      const Corner4<T> c = corner_setup<T>(loc[2 * s], loc[2 * s + 1], H, W);
      T ga = 0, gw = 0, gh = 0;
      if (c.in && live) {
        for (int ch = sub; ch < D; ch += G) {
          const T tg = go[ch], tgv = tg * a;
          const T v1 = c.ok1 ? vl[(size_t)c.o1 * row + ch] : (T)0;
          const T v2 = c.ok2 ? vl[(size_t)c.o2 * row + ch] : (T)0;
          const T v3 = c.ok3 ? vl[(size_t)c.o3 * row + ch] : (T)0;
          const T v4 = c.ok4 ? vl[(size_t)c.o4 * row + ch] : (T)0;
          if (c.ok1) atomicAdd(gl + (size_t)c.o1 * row + ch, c.k1 * tgv);
          if (c.ok2) atomicAdd(gl + (size_t)c.o2 * row + ch, c.k2 * tgv);
          if (c.ok3) atomicAdd(gl + (size_t)c.o3 * row + ch, c.k3 * tgv);
          if (c.ok4) atomicAdd(gl + (size_t)c.o4 * row + ch, c.k4 * tgv);
          ga += tg * (c.k1 * v1 + c.k2 * v2 + c.k3 * v3 + c.k4 * v4);
          gw += tgv * (c.hh * (v2 - v1) + c.lh * (v4 - v3));
          gh += tgv * (c.hw * (v3 - v1) + c.lw * (v4 - v2));
        }
      }
      ga = group_sum<T, G>(ga);
      gw = group_sum<T, G>(gw);
      gh = group_sum<T, G>(gh);
      if (sub == 0 && live) {
        grad_aw[s] = ga;
        grad_loc[2 * s] = (T)W * gw;
        grad_loc[2 * s + 1] = (T)H * gh;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------
// Host side.
// ---------------------------------------------------------------------------------------
int check_dims(const char *who, int N, int S, int M, int D, int L, int Lq, int P) {
  if (N < 0 || S < 0 || M <= 0 || D <= 0 || L <= 0 || Lq < 0 || P <= 0)
    return fail(MSDA_ERR_INVALID_ARGUMENT,
                "%s: bad dimensions batch=%d spatial_size=%d num_heads=%d channels=%d "
                "num_levels=%d num_query=%d num_point=%d",
                who, N, S, M, D, L, Lq, P);
  return MSDA_OK;
}

int check_launch(const char *who) {
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess)
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
  if (D == 32 && small && aligned16(value) && aligned16(out)) {
    const uint32_t nblk = (uint32_t)((npairs + 31) / 32);
    if (P == 4 && aligned16(loc) && aligned16(aw))
      hipLaunchKernelGGL(msda_fwd_f32_d32_kernel<true>, dim3(nblk), dim3(256), 0, st, value,
                         shapes, lsi, loc, aw, S, M, L, Lq, P, (uint32_t)npairs, nblk, out);
    else
      hipLaunchKernelGGL(msda_fwd_f32_d32_kernel<false>, dim3(nblk), dim3(256), 0, st, value,
                         shapes, lsi, loc, aw, S, M, L, Lq, P, (uint32_t)npairs, nblk, out);
    return check_launch("msda_forward_f32");
  }
  return forward_generic<float>(value, shapes, lsi, loc, aw, N, S, M, D, L, Lq, P, out, st);
}

template <typename T>
int backward_impl(const char *who, const T *value, const int64_t *shapes, const int64_t *lsi,
                  const T *loc, const T *aw, const T *grad_out, int N, int S, int M, int D,
                  int L, int Lq, int P, T *grad_value, T *grad_loc, T *grad_aw, void *stream) {
  if (int rc = check_dims(who, N, S, M, D, L, Lq, P)) return rc;
  hipStream_t st = static_cast<hipStream_t>(stream);
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

}  // namespace

extern "C" {

int msda_abi_version(void) { return 1; }

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
  return backward_impl<float>("msda_backward_f32", value, spatial_shapes, level_start_index,
                              sampling_loc, attn_weight, grad_output, batch, spatial_size,
                              num_heads, channels, num_levels, num_query, num_point, grad_value,
                              grad_sampling_loc, grad_attn_weight, stream);
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
