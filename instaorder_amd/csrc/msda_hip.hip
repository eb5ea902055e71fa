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
  T h_im, w_im;         // pixel coordinates of the sample (loc * size - 0.5)
  int h0, w0;           // top-left corner (floor of the above; 0 when the gate is closed)
  int o1, o2, o3, o4;   // pixel index (h*W + w) of each corner, clamped into the level
  bool ok1, ok2, ok3, ok4, in;
};

// loc * size - 0.5 as ONE fused multiply-add, so that every kernel (and the binning pass and
// the tile kernel of the backward, which must agree on where a sample lands) gets bit-identical
// coordinates.
__device__ __forceinline__ float pix_coord(float loc, float size) { return __fmaf_rn(loc, size, -0.5f); }
__device__ __forceinline__ double pix_coord(double loc, double size) { return fma(loc, size, -0.5); }

template <typename T>
__device__ __forceinline__ Corner4<T> corner_setup(T lx, T ly, int H, int W) {
  Corner4<T> c;
  const T fH = (T)H, fW = (T)W;
  const T h_im = pix_coord(ly, fH);
  const T w_im = pix_coord(lx, fW);
  c.h_im = h_im;
  c.w_im = w_im;
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
  c.h0 = h0;
  c.w0 = w0;
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
// The kernel asks for 4 waves/SIMD (not 8) so the compiler keeps all 16 corner loads of a level
// in flight at once (+6 % measured).
// All offsets are 32-bit byte offsets from the (wave-uniform) tensor base, which lets the
// loads use the SGPR-base + VGPR-offset form (the host checks every tensor is < 4 GiB).
// =======================================================================================
__device__ __forceinline__ float4 ld16(const float *__restrict__ base, uint32_t byte_off) {
  return *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(base) + byte_off);
}

// bf16 storage (value / output / grad_output of the bf16 variant); arithmetic stays fp32.
struct bf16_t {
  uint16_t bits;
};
__device__ __forceinline__ float bf16_to_f32(uint32_t b) { return __uint_as_float(b << 16); }
__device__ __forceinline__ uint32_t f32_to_bf16(float f) {      // round to nearest even
  const uint32_t u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;   // keep a NaN a NaN
  return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}

// Four consecutive channels of a row.  `f32_off` is the byte offset the element would have in
// an fp32 tensor (4 * element index), so all index arithmetic is shared between the types.
__device__ __forceinline__ float4 ldrow4(const float *__restrict__ base, uint32_t f32_off) {
  return ld16(base, f32_off);
}
__device__ __forceinline__ float4 ldrow4(const bf16_t *__restrict__ base, uint32_t f32_off) {
  const uint2 r = *reinterpret_cast<const uint2 *>(reinterpret_cast<const char *>(base) + (f32_off >> 1));
  return make_float4(bf16_to_f32(r.x & 0xffffu), bf16_to_f32(r.x >> 16), bf16_to_f32(r.y & 0xffffu),
                     bf16_to_f32(r.y >> 16));
}
__device__ __forceinline__ void strow4(float *__restrict__ base, uint32_t f32_off, float4 v) {
  *reinterpret_cast<float4 *>(reinterpret_cast<char *>(base) + f32_off) = v;
}
__device__ __forceinline__ void strow4(bf16_t *__restrict__ base, uint32_t f32_off, float4 v) {
  *reinterpret_cast<uint2 *>(reinterpret_cast<char *>(base) + (f32_off >> 1)) =
      make_uint2(f32_to_bf16(v.x) | (f32_to_bf16(v.y) << 16), f32_to_bf16(v.z) | (f32_to_bf16(v.w) << 16));
}
// A 4-channel piece of a row as it travels in registers between its load and its store to LDS:
// fp32 as it is, bf16 still packed (half the registers), widened only at the store.
template <typename ST> struct RawPiece;
template <> struct RawPiece<float> { typedef float4 type; };
template <> struct RawPiece<bf16_t> { typedef uint2 type; };
__device__ __forceinline__ float4 ld_raw_piece(const float *__restrict__ base, uint32_t f32_off) { return ld16(base, f32_off); }
__device__ __forceinline__ uint2 ld_raw_piece(const bf16_t *__restrict__ base, uint32_t f32_off) {
  return *reinterpret_cast<const uint2 *>(reinterpret_cast<const char *>(base) + (f32_off >> 1));
}
__device__ __forceinline__ float4 widen_piece(float4 v) { return v; }
__device__ __forceinline__ float4 widen_piece(uint2 r) {
  return make_float4(bf16_to_f32(r.x & 0xffffu), bf16_to_f32(r.x >> 16), bf16_to_f32(r.y & 0xffffu),
                     bf16_to_f32(r.y >> 16));
}
// NS samples of one level: all 4*NS corner rows are requested before the first is used.
template <typename VT, int NS>
__device__ __forceinline__ void fwd_samples_d32(const VT *__restrict__ value, uint32_t lbyte,
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
    v[i][0] = ldrow4(value, lbyte + (uint32_t)c.o1 * row_bytes);
    v[i][1] = ldrow4(value, lbyte + (uint32_t)c.o2 * row_bytes);
    v[i][2] = ldrow4(value, lbyte + (uint32_t)c.o3 * row_bytes);
    v[i][3] = ldrow4(value, lbyte + (uint32_t)c.o4 * row_bytes);
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

template <typename VT, bool P4>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void msda_fwd_d32_kernel(
    const VT *__restrict__ value, const int64_t *__restrict__ shapes,
    const int64_t *__restrict__ lsi, const float *__restrict__ loc,
    const float *__restrict__ aw, int S, int M, int L, int Lq, int P, uint32_t npairs,
    uint32_t nblocks, VT *__restrict__ out) {
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
      fwd_samples_d32<VT, 4>(value, lbyte, row_bytes, H, W, lx, ly, a, acc);
    } else {
      for (int p = 0; p < P; ++p) {
        const uint32_t s = sbase + (uint32_t)(l * P + p);
        const float lx[1] = {loc[2u * s]}, ly[1] = {loc[2u * s + 1u]}, a[1] = {aw[s]};
        fwd_samples_d32<VT, 1>(value, lbyte, row_bytes, H, W, lx, ly, a, acc);
      }
    }
  }
  strow4(out, pair * 128u + sub * 16u, acc);
}

// =======================================================================================
// Forward, fp32 arithmetic, D == 32, L*P <= 32: the kernel every InstaFormer shape runs.
// Replaces ms_deformable_im2col_gpu_kernel<float> (ms_deform_im2col_cuda.cuh:242-304).
//
// What the measurements behind it say (tools/microbench_gather.hip, profiles/r01):
//   * the vector L1 path charges per lane-group row touch, ~2 clk per 8 rows of 16 B/lane
//     wave-instruction, whether the rows hit in L1 or in L2 and whether lane groups share
//     a line; a dwordx2 instruction costs as much as a dwordx4 one.  So every lane always
//     moves 16 B: an fp32 row (128 B) takes 8 lanes, a bf16 row (64 B) takes 4.
//   * the kernel above recomputes a sample's geometry in all lanes of its pair and waits for a
//     round trip to HBM for the locations of every level.
// A wave owns GP = 64/LPG CONSECUTIVE QUERIES OF ONE HEAD (not the heads of one query):
// neighbouring queries sample neighbouring, often the same, pixel rows, and in this mapping
// those repeats sit inside one wave-instruction instead of arriving from other waves while the
// line is still being filled (TCP->L2 requests -43 %).
//   phase 1: lane i takes samples i, i + 64, ... of the wave's GP*L*P samples (ONE round trip
//            to HBM for locations and weights), runs corner_setup once per sample and leaves
//            (4 corner weights * attn, 4 row byte offsets) in a wave-private LDS region;
//   phase 2: the LPG lanes of a query walk its samples NS at a time: two broadcast
//            ds_read_b128 per sample, then 4 row loads of 16 B and 4*CH FMAs per lane.
// Per-query LDS regions are padded by 16 B so the broadcast addresses of a ds_read_b128 fall in
// different banks for every L*P.  Dynamic LDS: 4 waves * 2 * GP * (L*P + 1) * 16 bytes.
// =======================================================================================
template <typename VT> struct FwdLanes;
template <> struct FwdLanes<float> { static constexpr int LPG = 8, CH = 4; };
template <> struct FwdLanes<bf16_t> { static constexpr int LPG = 4, CH = 8; };

__device__ __forceinline__ uint4 ld_raw16(const void *__restrict__ base, uint32_t byte_off) {
  return *reinterpret_cast<const uint4 *>(reinterpret_cast<const char *>(base) + byte_off);
}
__device__ __forceinline__ void fma_piece(float (&acc)[4], float k, uint4 r, const float *) {
  acc[0] = fmaf(k, __uint_as_float(r.x), acc[0]);
  acc[1] = fmaf(k, __uint_as_float(r.y), acc[1]);
  acc[2] = fmaf(k, __uint_as_float(r.z), acc[2]);
  acc[3] = fmaf(k, __uint_as_float(r.w), acc[3]);
}
__device__ __forceinline__ void fma_piece(float (&acc)[8], float k, uint4 r, const bf16_t *) {
  acc[0] = fmaf(k, __uint_as_float(r.x << 16), acc[0]);
  acc[1] = fmaf(k, __uint_as_float(r.x & 0xffff0000u), acc[1]);
  acc[2] = fmaf(k, __uint_as_float(r.y << 16), acc[2]);
  acc[3] = fmaf(k, __uint_as_float(r.y & 0xffff0000u), acc[3]);
  acc[4] = fmaf(k, __uint_as_float(r.z << 16), acc[4]);
  acc[5] = fmaf(k, __uint_as_float(r.z & 0xffff0000u), acc[5]);
  acc[6] = fmaf(k, __uint_as_float(r.w << 16), acc[6]);
  acc[7] = fmaf(k, __uint_as_float(r.w & 0xffff0000u), acc[7]);
}
__device__ __forceinline__ uint4 pack_piece(const float (&a)[4], const float *) {
  return make_uint4(__float_as_uint(a[0]), __float_as_uint(a[1]), __float_as_uint(a[2]), __float_as_uint(a[3]));
}
__device__ __forceinline__ uint4 pack_piece(const float (&a)[8], const bf16_t *) {
  return make_uint4(f32_to_bf16(a[0]) | (f32_to_bf16(a[1]) << 16), f32_to_bf16(a[2]) | (f32_to_bf16(a[3]) << 16),
                    f32_to_bf16(a[4]) | (f32_to_bf16(a[5]) << 16), f32_to_bf16(a[6]) | (f32_to_bf16(a[7]) << 16));
}

// NS samples of one query: all 4*NS row pieces are requested before the first is used.
template <typename VT, int NS>
__device__ __forceinline__ void fwd_rows_d32(const VT *__restrict__ value, uint32_t vbyte,
                                             const float4 *__restrict__ sk, const uint4 *__restrict__ so,
                                             float (&acc)[FwdLanes<VT>::CH]) {
  float4 k[NS];
  uint4 v[NS][4];
#pragma unroll
  for (int i = 0; i < NS; ++i) {
    k[i] = sk[i];
    const uint4 o = so[i];
    v[i][0] = ld_raw16(value, vbyte + o.x);
    v[i][1] = ld_raw16(value, vbyte + o.y);
    v[i][2] = ld_raw16(value, vbyte + o.z);
    v[i][3] = ld_raw16(value, vbyte + o.w);
  }
#pragma unroll
  for (int i = 0; i < NS; ++i) {
    fma_piece(acc, k[i].x, v[i][0], value);
    fma_piece(acc, k[i].y, v[i][1], value);
    fma_piece(acc, k[i].z, v[i][2], value);
    fma_piece(acc, k[i].w, v[i][3], value);
  }
}

constexpr size_t FWD_QH_MAX_LDS = 48 * 1024;   // dynamic LDS per block the launcher accepts: L*P <= 47 (fp32), 23 (bf16)

// FUSED (row f3 of SURVEY section 8, an opt-in second boundary): `loc` holds the raw sampling
// OFFSETS and `aw` the attention LOGITS of the caller (ops/modules/ms_deform_attn.py:102-109);
// phase 1 forms loc = reference_point + offset / (W_l, H_l) and the softmax over the L*P logits
// of a (query, head) itself (logits through the wave's LDS region), so the caller's softmax and
// location arithmetic -- one write and one read of both tensors -- disappear.  L*P <= 16.
constexpr int FWD_FUSED_MAX_LP = 16;

template <typename VT, bool FUSED>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void msda_fwd_d32_qh_kernel(
    const VT *__restrict__ value, const int64_t *__restrict__ shapes,
    const int64_t *__restrict__ lsi, const float *__restrict__ loc,
    const float *__restrict__ aw, const float *__restrict__ refp, int S, int M, int L, int Lq, int P,
    uint32_t nq, uint32_t nunits, uint32_t nblocks, bool wide, VT *__restrict__ out) {
  constexpr uint32_t LPG = FwdLanes<VT>::LPG, GP = 64u / LPG, CH = FwdLanes<VT>::CH;
  extern __shared__ float4 s_dyn[];
  const uint32_t LP = (uint32_t)(L * P);
  const uint32_t pstride = LP + 1u;                                  // float4 units per query, padded
  const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
  const uint32_t wave_f4 = 2u * GP * pstride + (FUSED ? (GP * LP + 3u) / 4u : 0u);   // float4 units per wave
  float4 *sk = s_dyn + wave * wave_f4;                               // [GP][pstride] corner weights * attn
  uint4 *so = reinterpret_cast<uint4 *>(sk + GP * pstride);          // [GP][pstride] row byte offsets
  const uint32_t unit = xcd_remap(blockIdx.x, nblocks) * 4u + wave;  // (run of GP queries, head)
  if (unit >= nunits) return;
  const uint32_t run = unit / (uint32_t)M, m = unit - run * (uint32_t)M;
  const uint32_t q0 = run * GP;                                      // first n*Lq + q of the wave
  const uint32_t row_bytes = (uint32_t)M * 32u * (uint32_t)sizeof(VT);
  if constexpr (FUSED) {
    constexpr int FR = (int)(GP * FWD_FUSED_MAX_LP / 64u);            // sample rounds per lane
    float *lg = reinterpret_cast<float *>(so + GP * pstride);       // [GP * LP] logits of the wave
    const uint32_t nsamp = min(GP, nq - q0) * LP;
    float2 offv[FR];
#pragma unroll
    for (int r = 0; r < FR; ++r) {
      const uint32_t i = lane + 64u * (uint32_t)r;
      offv[r] = make_float2(0.f, 0.f);
      if (i < nsamp) {
        const uint32_t g = i / LP, j = i - g * LP;
        const uint32_t s = ((q0 + g) * (uint32_t)M + m) * LP + j;
        offv[r] = *reinterpret_cast<const float2 *>(reinterpret_cast<const char *>(loc) + s * 8u);
        lg[i] = aw[s];
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int r = 0; r < FR; ++r) {
      const uint32_t i = lane + 64u * (uint32_t)r;
      if (i >= nsamp) continue;
      const uint32_t g = i / LP, j = i - g * LP, l = j / (uint32_t)P;
      const float *row = lg + g * LP;
      float mx = row[0];
      for (uint32_t t = 1; t < LP; ++t) mx = fmaxf(mx, row[t]);
      float sum = 0.f;
      for (uint32_t t = 0; t < LP; ++t) sum += expf(row[t] - mx);
      const float a = __fdiv_rn(expf(row[j] - mx), sum);
      const int H = (int)shapes[2u * l], W = (int)shapes[2u * l + 1u];
      const float2 rp = *reinterpret_cast<const float2 *>(refp + 2u * ((q0 + g) * (uint32_t)L + l));
      // exactly the caller's arithmetic: reference + offset / (W, H), one rounding each
      const float2 xy = make_float2(__fadd_rn(rp.x, __fdiv_rn(offv[r].x, (float)W)),
                                    __fadd_rn(rp.y, __fdiv_rn(offv[r].y, (float)H)));
      const uint32_t lrow = (uint32_t)lsi[l];
      const Corner4<float> c = corner_setup<float>(xy.x, xy.y, H, W);
      sk[g * pstride + j] = make_float4(c.k1 * a, c.k2 * a, c.k3 * a, c.k4 * a);
      so[g * pstride + j] = make_uint4((lrow + (uint32_t)c.o1) * row_bytes, (lrow + (uint32_t)c.o2) * row_bytes,
                                       (lrow + (uint32_t)c.o3) * row_bytes, (lrow + (uint32_t)c.o4) * row_bytes);
    }
  } else {
    const uint32_t nsamp = min(GP, nq - q0) * LP;
    auto one = [&](uint32_t g, uint32_t j, float x, float y, float a) {
      const uint32_t l = j / (uint32_t)P;
      const int H = (int)shapes[2u * l], W = (int)shapes[2u * l + 1u];
      const uint32_t lrow = (uint32_t)lsi[l];
      const Corner4<float> c = corner_setup<float>(x, y, H, W);
      sk[g * pstride + j] = make_float4(c.k1 * a, c.k2 * a, c.k3 * a, c.k4 * a);
      so[g * pstride + j] = make_uint4((lrow + (uint32_t)c.o1) * row_bytes, (lrow + (uint32_t)c.o2) * row_bytes,
                                       (lrow + (uint32_t)c.o3) * row_bytes, (lrow + (uint32_t)c.o4) * row_bytes);
    };
    if ((LP & 1u) == 0u && wide) {
      // two samples per lane: one 16-B location load and one 8-B weight load (a wave-instruction
      // costs the vector L1 the same ~16 clk whatever its width: half as many of them)
      for (uint32_t i = 2u * lane; i < nsamp; i += 128u) {
        const uint32_t g = i / LP, j = i - g * LP;
        const uint32_t s = ((q0 + g) * (uint32_t)M + m) * LP + j;
        const float4 xy = ld16(loc, s * 8u);
        const float2 a = *reinterpret_cast<const float2 *>(reinterpret_cast<const char *>(aw) + s * 4u);
        one(g, j, xy.x, xy.y, a.x);
        one(g, j + 1u, xy.z, xy.w, a.y);
      }
    } else {
      for (uint32_t i = lane; i < nsamp; i += 64u) {
        const uint32_t g = i / LP, j = i - g * LP;
        const uint32_t s = ((q0 + g) * (uint32_t)M + m) * LP + j;
        const float2 xy = *reinterpret_cast<const float2 *>(reinterpret_cast<const char *>(loc) + s * 8u);
        one(g, j, xy.x, xy.y, aw[s]);
      }
    }
  }
  // the region is private to the wave: order its LDS writes before the reads below, no s_barrier
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  const uint32_t g = lane / LPG, sub = lane % LPG;
  const uint32_t q = q0 + g;
  if (q >= nq) return;
  const uint32_t n = q / (uint32_t)Lq;
  const uint32_t vbyte = n * (uint32_t)S * row_bytes + m * 32u * (uint32_t)sizeof(VT) + sub * 16u;
  float acc[CH];
#pragma unroll
  for (uint32_t c = 0; c < CH; ++c) acc[c] = 0.f;
  const float4 *pk = sk + g * pstride;
  const uint4 *po = so + g * pstride;
  uint32_t j = 0;
  for (; j + 4u <= LP; j += 4u) fwd_rows_d32<VT, 4>(value, vbyte, pk + j, po + j, acc);
  for (; j < LP; ++j) fwd_rows_d32<VT, 1>(value, vbyte, pk + j, po + j, acc);
  *reinterpret_cast<uint4 *>(reinterpret_cast<char *>(out) + (q * (uint32_t)M + m) * 32u * (uint32_t)sizeof(VT) + sub * 16u) =
      pack_piece(acc, value);
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

// =======================================================================================
// Backward, fp32 arithmetic, D == 32, P <= 4, any Lq: *binned* destination-tiled backward.
// Replaces ms_deformable_col2im_gpu_kernel_shm_blocksize_aware_reduce_v1<float, 32>
// (ms_deform_im2col_cuda.cuh:306-408), the variant every InstaFormer config dispatches to.
//
// Why: scattering grad_value with global float atomics runs at the chip-wide atomic rate
// (~1.35 TB/s measured: 12 ms for bs16 @1024^2), and LDS float atomics are no way out
// (ds_add_f32 measured at ~185 cycles per wave-instruction on gfx950,
// tools/microbench_lds_atomic.hip).  So grad_value is accumulated in REGISTERS, with no
// float atomic anywhere:
//
//   * A work item is one (image n, level l, 16x16-pixel tile of that level, head m).  Its
//     workgroup owns those 256 grad_value rows: thread (pixel, half) keeps 16 channels of
//     one pixel in registers for the whole item and stores them once at the end (plain,
//     coalesced stores; grad_value needs no zero-fill).
//   * A counting sort hands every tile the list of the samples that touch it.  A list entry
//     is one *quad* -- the (up to 4) points of one (image, query, head, level) -- plus a mask
//     of the points that have at least one in-level corner inside the tile; a quad is listed
//     once in every tile its points touch (model-like locations: 1.35 tiles per quad, 3.0 of
//     4 points live per entry).  Any distribution of sampling locations and any Lq run
//     through the same kernels.
//   * The tile kernel takes its list in chunks of 128 entries, four threads (one per point)
//     per entry.  Per chunk: counting sort of the (corner -> pixel) contributions in LDS
//     (integer LDS atomics only), grad_output rows of the entries staged once in LDS, then
//     every pixel's owner threads walk their pixel's bucket: acc += w * grad_output_row.
//   * The tile that holds a sample's top-left in-level corner (its *home*) also computes that
//     sample's grad_sampling_loc / grad_attn_weight: two lanes x 16 channels read the four
//     corner rows from an LDS copy of the (T+1)^2 value tile and the staged grad_output row.
//   * Samples whose gate is closed get their zero gradients from the binning pass.
//
// Kernels, all on the caller's stream, all reading the level shapes on the device:
//   msda_bwd_bin_zero_kernel    clears the per-list counters and the work counter,
//   msda_bwd_bin_kernel<false>  looks at every location once: describes each quad (which of
//                               <= 2x2 tiles its points touch) and counts entries per list --
//                               LDS histogram per workgroup, one global atomic per touched list,
//   msda_bwd_bin_scan_kernel    exclusive prefix sum -> list starts,
//   msda_bwd_bin_kernel<true>   writes the entries to their lists (from the descriptors), and
//                               the zero gradients of gate-closed samples,
//   msda_bwd_binned_kernel      the tile kernel; workgroups take items from a queue ordered
//                               coarse level first (their tiles carry the longest lists).
// Workspace (caller-provided), uint32 words: counters[key_cap + 1], work counter,
// starts[key_cap + 1], a zero word (what prefetches past the end of a list read), descriptors[quads], entries[16 * quads]; key_cap = N*M*S >= number
// of lists (a level has at most H*W tiles), and a quad touches at most 16 tiles (typically
// 1-2: the bound is address space, not traffic).
// =======================================================================================
namespace binned {

// Diagnostic build only (-DMSDA_PHASE_STAMPS): per-phase shader-clock sums of the tile kernel,
// accumulated per wave and added to a device array that no kernel reads.
#ifdef MSDA_PHASE_STAMPS
__device__ unsigned long long g_phase_cycles[16];
#define MSDA_STAMP_DECL unsigned long long st_last = clock64(), st_ph[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#define MSDA_STAMP(i) { const unsigned long long st_now = clock64(); st_ph[i] += st_now - st_last; st_last = st_now; }
#define MSDA_STAMP_FLUSH if ((threadIdx.x & 63) == 0) { for (int i = 0; i < 12; ++i) atomicAdd(&g_phase_cycles[i], st_ph[i]); }
#else
#define MSDA_STAMP_DECL
#define MSDA_STAMP(i)
#define MSDA_STAMP_FLUSH
#endif

// Workgroup-uniform values read from LDS land in VGPRs; pin them to scalar registers.
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }

// Inclusive prefix sum over the 64 lanes of a wave: DPP shifts inside each row of 16, then the
// row totals through scalar registers.
__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v) {
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, true);   // row_shr:1
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, true);   // row_shr:2
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, true);   // row_shr:4
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, true);   // row_shr:8
  const uint32_t t0 = (uint32_t)__builtin_amdgcn_readlane((int)v, 15);
  const uint32_t t1 = (uint32_t)__builtin_amdgcn_readlane((int)v, 31);
  const uint32_t t2 = (uint32_t)__builtin_amdgcn_readlane((int)v, 47);
  const uint32_t row = (threadIdx.x & 63u) >> 4;
  return v + (row > 0u ? t0 : 0u) + (row > 1u ? t1 : 0u) + (row > 2u ? t2 : 0u);
}

// Sum over an aligned lane pair (quad permute [1,0,3,2]), in DPP: no LDS round trip.
__device__ __forceinline__ float sum2(float v) {
  return v + __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
}

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

constexpr int T = 16;
constexpr int TP = T * T;
constexpr int VT = T + 1;
constexpr int ZROW = VT * VT;
constexpr int NT = 512;
constexpr int CE = NT / 4;        // list entries per chunk: four threads (one per point) per entry
constexpr int MAXL = 8;
constexpr int MAXP = 4;           // entry = (n*Lq + q) << 4 | point mask
constexpr int KCAP = 8192;        // lists per image the binning kernels can histogram in LDS
constexpr int BIN_NT = 256;
constexpr int BIN_QPT = 8;        // quads per thread of the binning kernels
constexpr int BIN_QB = BIN_NT * BIN_QPT;
constexpr int MAX_ENTRIES_PER_QUAD = 16;

struct Levels {
  int H[MAXL], W[MAXL], lsi[MAXL], ntx[MAXL], toff[MAXL + 1];
};

// thread 0 of a workgroup fills the table; caller synchronises
__device__ __forceinline__ void load_levels(Levels &lv, const int64_t *__restrict__ shapes,
                                            const int64_t *__restrict__ lsi, int L) {
  int to = 0;
  for (int l = 0; l < L; ++l) {
    const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
    lv.H[l] = H;
    lv.W[l] = W;
    lv.lsi[l] = lsi ? (int)lsi[l] : 0;
    lv.ntx[l] = (W + T - 1) / T;
    lv.toff[l] = to;
    to += ((H + T - 1) / T) * ((W + T - 1) / T);
  }
  lv.toff[L] = to;
}

__device__ __forceinline__ uint64_t count_lists(const int64_t *__restrict__ shapes, int N, int M, int L) {
  int to = 0;
  for (int l = 0; l < L; ++l)
    to += (((int)shapes[2 * l] + T - 1) / T) * (((int)shapes[2 * l + 1] + T - 1) / T);
  return (uint64_t)N * (uint64_t)to * (uint64_t)M;
}

__global__ __launch_bounds__(256) void msda_bwd_bin_zero_kernel(const int64_t *__restrict__ shapes,
                                                                int N, int M, int L, uint32_t key_cap,
                                                                uint32_t *__restrict__ counts) {
  const uint64_t nkeys = count_lists(shapes, N, M, L);
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    counts[(size_t)key_cap + 1] = 0u;         // the tile kernel's work counter
    counts[2 * (size_t)key_cap + 3] = 0u;     // the pad word behind the list starts: the tile kernel's zero word
  }
  if (nkeys > key_cap) return;
  for (uint64_t k = (uint64_t)blockIdx.x * 256u + threadIdx.x; k <= nkeys; k += (uint64_t)gridDim.x * 256u)
    counts[k] = 0u;
}

// Tile rows / columns spanned by the in-level corners of one sample (a == b when both rows,
// or both columns, fall in the same tile or only one is inside the level).  corner_setup's
// own floor is used, so the tile kernel -- which recomputes the geometry from the same
// location -- finds the sample in exactly these tiles.
struct PointTiles {
  int tya, tyb, txa, txb;
  bool in;
};
__device__ __forceinline__ PointTiles point_tiles(float lx, float ly, int H, int W) {
  // the part of corner_setup that decides WHERE a sample lands: same fused multiply-add, same gate,
  // same floor -- bit-identical h0 / w0 -- without the weights
  const float fH = (float)H, fW = (float)W;
  const float h_im = pix_coord(ly, fH), w_im = pix_coord(lx, fW);
  PointTiles t;
  t.in = (h_im > -1.f) && (w_im > -1.f) && (h_im < fH) && (w_im < fW);
  const int h0 = (int)floorf(t.in ? h_im : 0.f), w0 = (int)floorf(t.in ? w_im : 0.f);
  // first / last in-level row and column of the 2 x 2 footprint (h0, w0 >= -1 when the gate is open)
  t.tya = (uint32_t)max(h0, 0) / (uint32_t)T;
  t.tyb = (uint32_t)min(h0 + 1, H - 1) / (uint32_t)T;
  t.txa = (uint32_t)max(w0, 0) / (uint32_t)T;
  t.txb = (uint32_t)min(w0 + 1, W - 1) / (uint32_t)T;
  return t;
}

// i / d for i < 2^32 and d < 2^8, with magic = ceil(2^40 / d) from the host (exact: i * d < 2^40)
__device__ __forceinline__ uint32_t div_magic(uint32_t i, uint64_t magic) {
  return (uint32_t)(((uint64_t)i * magic) >> 40);
}

// Calls f(tile_y, tile_x, mask) once per list the quad belongs to.  When the quad's points
// span at most 2 x 2 tiles (the rule for model-like locations) every tile is listed once with
// the mask of all points touching it; otherwise each point is listed on its own.
template <typename F>
__device__ __forceinline__ void for_each_entry(const PointTiles (&pt)[4], F &&f) {
  int tymin = 1 << 30, tymax = -1, txmin = 1 << 30, txmax = -1;
#pragma unroll
  for (int j = 0; j < 4; ++j)
    if (pt[j].in) {
      tymin = min(tymin, pt[j].tya);
      tymax = max(tymax, pt[j].tyb);
      txmin = min(txmin, pt[j].txa);
      txmax = max(txmax, pt[j].txb);
    }
  if (tymax < 0) return;
  if (tymax - tymin <= 1 && txmax - txmin <= 1) {
    for (int ty = tymin; ty <= tymax; ++ty)
      for (int tx = txmin; tx <= txmax; ++tx) {
        uint32_t mask = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (pt[j].in && (ty == pt[j].tya || ty == pt[j].tyb) && (tx == pt[j].txa || tx == pt[j].txb))
            mask |= 1u << j;
        if (mask) f(ty, tx, mask);
      }
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (pt[j].in)
        for (int ty = pt[j].tya; ty <= pt[j].tyb; ++ty)
          for (int tx = pt[j].txa; tx <= pt[j].txb; ++tx) f(ty, tx, 1u << j);
  }
}

// What the counting pass learnt about a quad, kept for the scatter pass (one uint32 per quad):
// bit 31 clear: the quad spans at most 2 x 2 tiles; bits 0-14 hold the first tile's index in
// its level, bits 15-30 the four 4-bit point masks of tiles (0,0), (0,1), (1,0), (1,1); a
// point in none of them is gate-closed.  Bit 31 set: look at the locations again.
constexpr uint32_t DESC_SLOW = 1u << 31;
constexpr int DESC_MAX_TILE = 1 << 15;

template <bool SCATTER>
__global__ __launch_bounds__(BIN_NT) void msda_bwd_bin_kernel(
    const int64_t *__restrict__ shapes, const float *__restrict__ loc, int N, int M, int L, int Lq,
    int P, uint32_t qpi, uint32_t bpi, uint64_t magic_ml, uint64_t magic_l, uint32_t key_cap,
    uint32_t *__restrict__ counts,
    uint32_t *__restrict__ desc, uint32_t *__restrict__ entries, float *__restrict__ grad_loc,
    float *__restrict__ grad_aw) {
  __shared__ uint32_t s_hist[KCAP];
  __shared__ Levels lv;
  const int tid = threadIdx.x;
  if (tid == 0) load_levels(lv, shapes, nullptr, L);
  __syncthreads();
  const uint32_t keys_img = (uint32_t)lv.toff[L] * (uint32_t)M;
  if ((uint64_t)N * keys_img > key_cap) return;        // shapes break the sum(H*W) == S contract
  const bool use_lds = keys_img <= (uint32_t)KCAP;
  const uint32_t n = blockIdx.x / bpi, base = (blockIdx.x % bpi) * (uint32_t)BIN_QB;
  const uint32_t gk0 = n * keys_img;
  const uint32_t ML = (uint32_t)(M * L);
  const uint32_t q0 = n * qpi;                          // first quad of this image

  auto decode = [&](uint32_t i, uint32_t &q, uint32_t &m, uint32_t &l) {
    q = div_magic(i, magic_ml);
    const uint32_t r = i - q * ML;
    m = div_magic(r, magic_l);
    l = r - m * (uint32_t)L;
  };
  // quad i of the image -> its points' tiles
  auto examine = [&](uint32_t i, uint32_t l, PointTiles (&pt)[4]) {
    const int H = lv.H[l], W = lv.W[l];
    const float *lp = loc + 2u * (q0 + i) * (uint32_t)P;
    if (P == 4) {
      const float4 a = *reinterpret_cast<const float4 *>(lp), b = *reinterpret_cast<const float4 *>(lp + 4);
      pt[0] = point_tiles(a.x, a.y, H, W);
      pt[1] = point_tiles(a.z, a.w, H, W);
      pt[2] = point_tiles(b.x, b.y, H, W);
      pt[3] = point_tiles(b.z, b.w, H, W);
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (j < P) {
          pt[j] = point_tiles(lp[2 * j], lp[2 * j + 1], H, W);
        } else {
          pt[j].in = false;
          pt[j].tya = pt[j].tyb = pt[j].txa = pt[j].txb = 0;
        }
      }
    }
  };
  // f(list key within the image, point mask) for every list of a quad described by `d`
  auto for_each_fast = [&](uint32_t d, uint32_t m, uint32_t l, auto &&f) {
    const uint32_t t0 = (uint32_t)lv.toff[l] + (d & (uint32_t)(DESC_MAX_TILE - 1)), ntx = (uint32_t)lv.ntx[l];
    const uint32_t m00 = (d >> 15) & 15u, m01 = (d >> 19) & 15u, m10 = (d >> 23) & 15u, m11 = (d >> 27) & 15u;
    if (m00) f(t0 * (uint32_t)M + m, m00);
    if (m01) f((t0 + 1u) * (uint32_t)M + m, m01);
    if (m10) f((t0 + ntx) * (uint32_t)M + m, m10);
    if (m11) f((t0 + ntx + 1u) * (uint32_t)M + m, m11);
  };
  auto for_each_slow = [&](uint32_t i, uint32_t m, uint32_t l, uint32_t &closed, auto &&f) {
    PointTiles pt[4];
    examine(i, l, pt);
    const int toff = lv.toff[l], ntx = lv.ntx[l];
    closed = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (j < P && !pt[j].in) closed |= 1u << j;
    for_each_entry(pt, [&](int ty, int tx, uint32_t mask) {
      f((uint32_t)(toff + ty * ntx + tx) * (uint32_t)M + m, mask);
    });
  };
  auto bump = [&](uint32_t key, uint32_t) {
    if (use_lds)
      atomicAdd(&s_hist[key], 1u);
    else
      atomicAdd(&counts[gk0 + key], 1u);
  };

  if (use_lds) {
    for (uint32_t k = tid; k < keys_img; k += BIN_NT) s_hist[k] = 0u;
    __syncthreads();
  }
  if (!SCATTER) {
    // counting pass: look at the locations, describe the quad, count its lists
    for (int r = 0; r < BIN_QPT; ++r) {
      const uint32_t i = base + (uint32_t)(r * BIN_NT + tid);
      if (i >= qpi) continue;
      uint32_t q, m, l;
      decode(i, q, m, l);
      PointTiles pt[4];
      examine(i, l, pt);
      int tymin = 1 << 30, tymax = -1, txmin = 1 << 30, txmax = -1;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (pt[j].in) {
          tymin = min(tymin, pt[j].tya);
          tymax = max(tymax, pt[j].tyb);
          txmin = min(txmin, pt[j].txa);
          txmax = max(txmax, pt[j].txb);
        }
      uint32_t d = 0;
      if (tymax >= 0) {
        const int ntx = lv.ntx[l], t0 = tymin * ntx + txmin;
        if (tymax - tymin <= 1 && txmax - txmin <= 1 && t0 < DESC_MAX_TILE) {
          d = (uint32_t)t0;
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (pt[j].in) {
              // rows / columns of the 2 x 2 tile block the point touches (bit 0: first, bit 1: second)
              const uint32_t rm = (1u << (pt[j].tya - tymin)) | (1u << (pt[j].tyb - tymin));
              const uint32_t cm = (1u << (pt[j].txa - txmin)) | (1u << (pt[j].txb - txmin));
              const uint32_t c2 = (cm & 1u) | ((cm & 2u) << 3);                          // column bit x -> bit 4 * x
              d |= (((rm & 1u) ? c2 : 0u) | ((rm & 2u) ? c2 << 8 : 0u)) << (15 + j);     // tile c = 2 * row + column -> bit 4 * c
            }
          for_each_fast(d, m, l, bump);
        } else {
          d = DESC_SLOW;
          const int toff = lv.toff[l];
          for_each_entry(pt, [&](int ty, int tx, uint32_t mask) {
            bump((uint32_t)(toff + ty * ntx + tx) * (uint32_t)M + m, mask);
          });
        }
      }
      desc[q0 + i] = d;
    }
    if (use_lds) {
      __syncthreads();
      for (uint32_t k = tid; k < keys_img; k += BIN_NT) {
        const uint32_t c = s_hist[k];
        if (c) atomicAdd(&counts[gk0 + k], c);
      }
    }
    return;
  }
  // scatter pass: per-workgroup counts again (from the descriptors), one reservation per list
  // in the global cursors, then the entries
  if (use_lds) {
    for (int r = 0; r < BIN_QPT; ++r) {
      const uint32_t i = base + (uint32_t)(r * BIN_NT + tid);
      if (i >= qpi) continue;
      const uint32_t d = desc[q0 + i];
      if (d == 0u) continue;
      uint32_t q, m, l, closed;
      decode(i, q, m, l);
      if (d & DESC_SLOW)
        for_each_slow(i, m, l, closed, bump);
      else
        for_each_fast(d, m, l, bump);
    }
    __syncthreads();
    for (uint32_t k = tid; k < keys_img; k += BIN_NT) {
      const uint32_t c = s_hist[k];
      if (c) s_hist[k] = atomicAdd(&counts[gk0 + k], c);   // counts holds the cursors by now
    }
    __syncthreads();
  }
  for (int r = 0; r < BIN_QPT; ++r) {
    const uint32_t i = base + (uint32_t)(r * BIN_NT + tid);
    if (i >= qpi) continue;
    const uint32_t d = desc[q0 + i];
    uint32_t q, m, l;
    decode(i, q, m, l);
    const uint32_t ent = (n * (uint32_t)Lq + q) << 4;
    auto put = [&](uint32_t key, uint32_t mask) {
      const uint32_t pos = use_lds ? atomicAdd(&s_hist[key], 1u) : atomicAdd(&counts[gk0 + key], 1u);
      entries[pos] = ent | mask;
    };
    uint32_t closed;
    if (d & DESC_SLOW) {
      for_each_slow(i, m, l, closed, put);
    } else {
      for_each_fast(d, m, l, put);
      closed = ~((d >> 15) | (d >> 19) | (d >> 23) | (d >> 27)) & ((1u << P) - 1u);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if ((closed >> j) & 1u) {                            // gate closed: zero gradients
        const uint32_t s = (q0 + i) * (uint32_t)P + (uint32_t)j;
        grad_aw[s] = 0.f;
        *reinterpret_cast<float2 *>(grad_loc + 2u * s) = make_float2(0.f, 0.f);
      }
  }
}

// One workgroup: offs[k] = sum of counts[0..k), counts[k] = the same (the scatter's cursors).
__global__ __launch_bounds__(1024) void msda_bwd_bin_scan_kernel(const int64_t *__restrict__ shapes,
                                                                 int N, int M, int L, uint32_t key_cap,
                                                                 uint32_t *__restrict__ counts,
                                                                 uint32_t *__restrict__ offs) {
  __shared__ uint32_t s_wave[16];
  const uint64_t nk64 = count_lists(shapes, N, M, L);
  if (nk64 > key_cap) return;
  const uint32_t nkeys = (uint32_t)nk64;
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  // tiles of 4096 consecutive keys (4 per thread), carried sequentially
  uint32_t carry = 0;
  for (uint32_t k0 = 0; k0 < nkeys; k0 += 4096u) {
    const uint32_t k = k0 + 4u * tid;
    uint32_t c[4];
#pragma unroll
    for (uint32_t i = 0; i < 4u; ++i) c[i] = k + i < nkeys ? counts[k + i] : 0u;
    const uint32_t sum = c[0] + c[1] + c[2] + c[3];
    const uint32_t inc = wave_inclusive_scan(sum);
    if (lane == 63u) s_wave[wave] = inc;
    __syncthreads();
    uint32_t wbase = 0, tot = 0;
    for (uint32_t w = 0; w < 16u; ++w) {
      const uint32_t v = s_wave[w];
      if (w < wave) wbase += v;
      tot += v;
    }
    uint32_t ex = carry + wbase + inc - sum;
#pragma unroll
    for (uint32_t i = 0; i < 4u; ++i) {
      if (k + i < nkeys) {
        offs[k + i] = ex;
        counts[k + i] = ex;
      }
      ex += c[i];
    }
    carry += tot;
    __syncthreads();
  }
  if (tid == 0) offs[nkeys] = carry;
}

// One work item of the tile kernel: a (image, level, tile, head) and its list.
struct Item {
  uint32_t n, m, list0, total, nchunks, lrow;   // lrow: value row of the level's first pixel
  int l, H, W, y0, x0, th, tw;
  bool split;                                   // one of several parts of its tile's list
};

constexpr int SPLIT_TARGET_CHUNKS = 6;          // chunks a part of a split list should get
constexpr int SPLIT_MAX_PARTS = 16;

struct Next {          // the part of an Item that fetching its chunks needs
  uint32_t m;
  int l;
  uint32_t list0, total;
};

template <typename ST>
__global__ __launch_bounds__(NT, 4) void msda_bwd_binned_kernel(
    const ST *__restrict__ grad_out, const ST *__restrict__ value,
    const int64_t *__restrict__ shapes, const int64_t *__restrict__ lsi,
    const float *__restrict__ loc, const float *__restrict__ aw,
    const uint32_t *__restrict__ offs, const uint32_t *__restrict__ entries,
    const uint32_t *__restrict__ zero_word, uint32_t *__restrict__ work, uint32_t key_cap, int N, int S,
    int M, int L, int P, int split_enable,
    float *__restrict__ grad_value, float *__restrict__ grad_loc, float *__restrict__ grad_aw) {
  __shared__ float4 s_vt[(ZROW + 1) * 8];   // value tile rows (8 x 16 B each)
  __shared__ float4 s_go[(CE + 1) * 8];     // grad_output rows of this chunk's entries + a zero row
  __shared__ uint2 s_rec[4 * NT + 4];       // a chunk's contributions, packed by pixel: (row byte offset in s_go, weight bits)
  __shared__ float4 s_home[NT];             // home samples: lh, lw, attn, packed geometry
  __shared__ uint32_t s_q[CE];              // n*Lq + q of each slot of the chunk
  __shared__ uint32_t s_cnt[2][TP + 64];    // contributions per pixel, by chunk parity (+ dummies)
  __shared__ uint32_t s_hcount[2];
  __shared__ uint32_t s_pos[2];             // queue positions handed out to this workgroup
  __shared__ int s_ord[MAXL], s_pbase[MAXL + 1], s_parts[MAXL];
  __shared__ Levels lv;

  const int tid = threadIdx.x;
  if (tid == 0) {
    load_levels(lv, shapes, lsi, L);
    s_hcount[0] = s_hcount[1] = 0u;
    // Work queue order: levels by ascending tile count.  Every level receives the same number
    // of samples, so the tiles of a coarse level carry the longest lists; handing those out
    // first lets the light tiles of the fine levels even out the tail.
    for (int i = 0; i < L; ++i) s_ord[i] = i;
    for (int i = 1; i < L; ++i)
      for (int j = i; j > 0; --j) {
        const int a = s_ord[j - 1], b = s_ord[j];
        if (lv.toff[b + 1] - lv.toff[b] < lv.toff[a + 1] - lv.toff[a]) {
          s_ord[j - 1] = b;
          s_ord[j] = a;
        }
      }
    // Small batches (the host sets split_enable and has zero-filled grad_value): the kernel would
    // run as long as its longest list, so the lists of a level whose tiles average more than
    // SPLIT_TARGET_CHUNKS chunks are cut into parts -- queue positions of their own, taken by
    // different workgroups, which ADD their partial tile sums with float atomics.
    int pb = 0;
    for (int i = 0; i < L; ++i) {
      const int l = s_ord[i], tiles = lv.toff[l + 1] - lv.toff[l];
      int parts = 1;
      if (split_enable && (uint64_t)N * (uint32_t)lv.toff[L] * (uint32_t)M <= key_cap) {
        const uint32_t ents = offs[(uint32_t)lv.toff[l + 1] * (uint32_t)M] - offs[(uint32_t)lv.toff[l] * (uint32_t)M];   // image 0
        const uint32_t avg_chunks = ents / ((uint32_t)(tiles * M) * (uint32_t)CE);
        parts = min(max((int)(avg_chunks / SPLIT_TARGET_CHUNKS), 1), SPLIT_MAX_PARTS);
      }
      s_parts[i] = parts;
      s_pbase[i] = pb;
      pb += N * M * tiles * parts;
    }
    s_pbase[L] = pb;
    s_pos[0] = gridDim.x + atomicAdd(work, 1u);          // this workgroup's second item
  }
  if (tid < 8) s_go[CE * 8 + tid] = make_float4(0.f, 0.f, 0.f, 0.f);     // what padding records read
  for (int t = tid; t < 2 * (TP + 64); t += NT) (&s_cnt[0][0])[t] = 0;
  __syncthreads();
  const uint32_t I_img = (uint32_t)uni(lv.toff[L]) * (uint32_t)M;
  if ((uint64_t)N * I_img > key_cap) return;
  const uint32_t total_items = (uint32_t)uni(s_pbase[L]);       // queue positions: items, or parts of items
  if (blockIdx.x >= total_items) return;
  MSDA_STAMP_DECL

  const uint32_t slot = (uint32_t)tid >> 2, p = (uint32_t)tid & 3u;      // exam: entry slot, point
  const uint32_t gslot = (uint32_t)tid >> 3, gsub = (uint32_t)tid & 7u;   // rows gslot, gslot + 64

  // queue position -> item index (the numbering of the lists: image, level, tile, head) and which
  // part of how many of its list
  auto item_at = [&](uint32_t pos, uint32_t &part, uint32_t &parts) -> uint32_t {
    int j = 0;
    while (j + 1 < L && (int)pos >= uni(s_pbase[j + 1])) ++j;
    const int l = uni(s_ord[j]);
    parts = (uint32_t)uni(s_parts[j]);
    const uint32_t rp = pos - (uint32_t)uni(s_pbase[j]);
    const uint32_t r = rp / parts;
    part = rp - r * parts;
    const uint32_t per = (uint32_t)(uni(lv.toff[l + 1]) - uni(lv.toff[l])) * (uint32_t)M;   // items per image
    const uint32_t n = r / per;
    return n * I_img + (uint32_t)uni(lv.toff[l]) * (uint32_t)M + (r - n * per);
  };
  // the slice of an item's list that part `part` of `parts` takes: whole chunks, equal shares
  auto slice = [&](uint32_t part, uint32_t parts, uint32_t &list0, uint32_t &total) {
    if (parts > 1u) {
      const uint32_t nch = (total + (uint32_t)CE - 1u) / (uint32_t)CE;
      const uint32_t cpp = max((nch + parts - 1u) / parts, 1u);
      const uint32_t skip = min(part * cpp * (uint32_t)CE, total);
      list0 += skip;
      total = min(total - skip, cpp * (uint32_t)CE);
    }
  };
  auto decode = [&](uint32_t pos) -> Item {
    uint32_t part, parts;
    const uint32_t item = item_at(pos, part, parts);
    Item it;
    it.n = item / I_img;
    const uint32_t ri = item - it.n * I_img, tl = ri / (uint32_t)M;
    it.m = ri - tl * (uint32_t)M;
    int l = 0;
    while (l + 1 < L && (int)tl >= uni(lv.toff[l + 1])) ++l;
    it.l = l;
    it.H = uni(lv.H[l]);
    it.W = uni(lv.W[l]);
    const int ntx = uni(lv.ntx[l]), tile = (int)tl - uni(lv.toff[l]);
    const int ty = tile / ntx, tx = tile - ty * ntx;
    it.y0 = ty * T;
    it.x0 = tx * T;
    it.th = min(T, it.H - it.y0);
    it.tw = min(T, it.W - it.x0);
    it.lrow = it.n * (uint32_t)S + (uint32_t)uni(lv.lsi[l]);
    it.list0 = (uint32_t)uni((int)offs[item]);
    it.total = (uint32_t)uni((int)offs[item + 1u]) - it.list0;
    it.split = parts > 1u;
    slice(part, parts, it.list0, it.total);
    // an empty list still takes one (empty) chunk: the item has to store its zero rows anyway
    it.nchunks = max((it.total + (uint32_t)CE - 1u) / (uint32_t)CE, 1u);
    return it;
  };
  // what fetching a chunk needs to know of its item
  auto peek = [&](uint32_t pos) -> Next {
    uint32_t part, parts;
    const uint32_t item = item_at(pos, part, parts);
    Next nx;
    const uint32_t ri = item % I_img, tl = ri / (uint32_t)M;
    nx.m = ri - tl * (uint32_t)M;
    int l = 0;
    while (l + 1 < L && (int)tl >= uni(lv.toff[l + 1])) ++l;
    nx.l = l;
    nx.list0 = (uint32_t)uni((int)offs[item]);
    nx.total = (uint32_t)uni((int)offs[item + 1u]) - nx.list0;
    slice(part, parts, nx.list0, nx.total);
    return nx;
  };
  // entries of chunk `c` this thread needs: its exam slot and the two rows it stages.
  // All prefetch loads are UNCONDITIONAL (lanes with nothing to fetch read element 0 and drop it):
  // loads inside exec-masked blocks make the compiler fall back to s_waitcnt vmcnt(0) at the first
  // use, which would expose the latency of the newest requests as well.
  auto load_ids = [&](const Next &it, uint32_t c, uint32_t &ent, uint32_t &r0, uint32_t &r1) {
    const uint32_t e = c * (uint32_t)CE + slot, e0 = c * (uint32_t)CE + gslot, e1 = e0 + 64u;
    // past the end of the list: the workspace's zero word (mask 0: nothing to examine), so the
    // loaded value needs no fix-up -- nothing touches it before its use, a full phase later
    ent = *(e < it.total ? entries + it.list0 + e : zero_word);
    r0 = *(e0 < it.total ? entries + it.list0 + e0 : zero_word);
    r1 = *(e1 < it.total ? entries + it.list0 + e1 : zero_word);
  };
  auto load_sample = [&](const Next &it, uint32_t ent, float2 &xy, float &a) {
    const bool live = (ent >> p) & 1u;
    const uint32_t s = live ? (((ent >> 4) * (uint32_t)M + it.m) * (uint32_t)L + (uint32_t)it.l) * (uint32_t)P + p : 0u;
    xy = *reinterpret_cast<const float2 *>(loc + 2u * s);
    a = aw[s];
  };
  typedef typename RawPiece<ST>::type Raw;
  auto load_rows = [&](const Next &it, uint32_t c, uint32_t r0, uint32_t r1, Raw &g0, Raw &g1) {
    const uint32_t e0 = c * (uint32_t)CE + gslot;
    const uint32_t o0 = e0 < it.total ? ((r0 >> 4) * (uint32_t)M + it.m) * 128u + gsub * 16u : gsub * 16u;
    const uint32_t o1 = e0 + 64u < it.total ? ((r1 >> 4) * (uint32_t)M + it.m) * 128u + gsub * 16u : gsub * 16u;
    g0 = ld_raw_piece(grad_out, o0);
    g1 = ld_raw_piece(grad_out, o1);
  };

  // The block's items and their chunks form one stream.  While a chunk is processed the next
  // one -- of this item or of the block's next item -- is fetched: its entry ids during phase
  // A, its sample (location, weight) and grad_output rows from phase C on.
  uint32_t pos = blockIdx.x, npos = (uint32_t)uni((int)s_pos[0]);
  Item cur = decode(pos);
  uint32_t c_ent = 0, n_ent = 0, n_r0 = 0, n_r1 = 0;
  float2 c_xy = make_float2(0.f, 0.f);
  float c_a = 0.f;
  Raw c_g0 = Raw(), c_g1 = Raw();
  {
    const Next first = {cur.m, cur.l, cur.list0, cur.total};
    uint32_t r0 = 0, r1 = 0;
    load_ids(first, 0, c_ent, r0, r1);
    load_sample(first, c_ent, c_xy, c_a);
    load_rows(first, 0, r0, r1, c_g0, c_g1);
  }
  uint32_t gchunk = 0;                                   // chunks so far: parity of the counters
  for (uint32_t k = 0;; ++k) {
    // position of the item after the next one: asked for now, published in phase C of this
    // item's first chunk, read by everyone when this item is done
    uint32_t fetched = 0;
    if (tid == 0) fetched = gridDim.x + atomicAdd(work, 1u);
    const bool has_next = npos < total_items;
    Next nxt = {0u, 0, 0u, 0u};
    if (has_next) nxt = peek(npos);
    const int H = cur.H, W = cur.W, y0 = cur.y0, x0 = cur.x0, th = cur.th, tw = cur.tw;
    const uint32_t m = cur.m;
    const int l = cur.l;
    float4 acc[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) acc[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    MSDA_STAMP(0)   // item turnover

    for (uint32_t chunk = 0; chunk < cur.nchunks; ++chunk, ++gchunk) {
      const int par = (int)(gchunk & 1u);
      const bool last = chunk + 1u == cur.nchunks;
      const bool nvalid = !last || has_next;             // there is a next chunk to fetch
      const Next ni = last ? nxt : Next{cur.m, cur.l, cur.list0, cur.total};
      const uint32_t nc = last ? 0u : chunk + 1u;
      if (nvalid) load_ids(ni, nc, n_ent, n_r0, n_r1);
      // ---------------- phase A: examine one point of one list entry per thread ----------------
      bool own1 = false, own2 = false, own3 = false, own4 = false, home = false;
      int r1 = 0;
      float4 hrec = make_float4(0.f, 0.f, 0.f, 0.f);   // lh, lw, attn, packed geometry
#ifdef MSDA_PHASE_STAMPS
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      MSDA_STAMP(8)
#endif
      if ((c_ent >> p) & 1u) {
        const float a = c_a;
        const Corner4<float> g = corner_setup<float>(c_xy.x, c_xy.y, H, W);
        const int ly0 = g.h0 - y0, lx0 = g.w0 - x0;       // corner 1 in tile coordinates
        const bool yin0 = (unsigned)ly0 < (unsigned)th, yin1 = (unsigned)(ly0 + 1) < (unsigned)th;
        const bool xin0 = (unsigned)lx0 < (unsigned)tw, xin1 = (unsigned)(lx0 + 1) < (unsigned)tw;
        own1 = g.ok1 && yin0 && xin0;
        own2 = g.ok2 && yin0 && xin1;
        own3 = g.ok3 && yin1 && xin0;
        own4 = g.ok4 && yin1 && xin1;
        r1 = ly0 * T + lx0;
        const int hy = max(g.h0, 0) - y0, hx = max(g.w0, 0) - x0;
        home = g.in && (unsigned)hy < (unsigned)th && (unsigned)hx < (unsigned)tw;
        const uint32_t ok = (g.ok1 ? 1u : 0u) | (g.ok2 ? 2u : 0u) | (g.ok3 ? 4u : 0u) |
                            (g.ok4 ? 8u : 0u);
        const uint32_t pk = (uint32_t)(ly0 + 1) | ((uint32_t)(lx0 + 1) << 5) | (ok << 10) |
                            ((uint32_t)tid << 14);
        hrec = make_float4(g.lh, g.lw, a, __uint_as_float(pk));
      }
      MSDA_STAMP(9)
      // Bucket positions: four unconditional, independent LDS atomics (corners the tile does
      // not own count into a per-lane dummy), so their latencies overlap.
      const int dummy = TP + (tid & 63);
      const uint32_t pos1 = atomicAdd(&s_cnt[par][own1 ? r1 : dummy], 1u);
      const uint32_t pos2 = atomicAdd(&s_cnt[par][own2 ? r1 + 1 : dummy], 1u);
      const uint32_t pos3 = atomicAdd(&s_cnt[par][own3 ? r1 + T : dummy], 1u);
      const uint32_t pos4 = atomicAdd(&s_cnt[par][own4 ? r1 + T + 1 : dummy], 1u);
      MSDA_STAMP(10)
      MSDA_STAMP(1)
      __syncthreads();   // every wave is past phases D/E of the previous chunk
      MSDA_STAMP(2)
      // ---------------- phase C: contributions -> pixel buckets; rows, home list -> LDS ----------------
      s_go[gslot * 8u + gsub] = widen_piece(c_g0);
      s_go[(gslot + 64u) * 8u + gsub] = widen_piece(c_g1);
      if (p == 0u) s_q[slot] = c_ent >> 4;
      if (chunk == 0) {
        // The item's value tile (zeros outside the level).  fp32 rows go global -> LDS directly
        // (one 1-KiB wave-instruction per 64 consecutive pieces, no registers, all five in
        // flight at once); the barrier below waits for them.
        if constexpr (sizeof(ST) == 4) {
#pragma unroll 1
          for (int j = 0; j < 5; ++j) {
            const int t = tid + j * NT, row = t >> 3, sub = t & 7;
            if (t < (ZROW + 1) * 8) {
              const int y = y0 + row / VT, x = x0 + row % VT;
              const bool inl = row < ZROW && y < H && x < W;
              const uint32_t off = ((cur.lrow + (uint32_t)(y * W + x)) * (uint32_t)M + m) * 128u + (uint32_t)sub * 16u;
              if (inl)
                __builtin_amdgcn_global_load_lds(
                    (const __attribute__((address_space(1))) void *)(reinterpret_cast<const char *>(value) + off),
                    (__attribute__((address_space(3))) void *)(&s_vt[t - (tid & 63)]), 16, 0, 0);   // wave base: lane i lands 16*i further
              else
                s_vt[t] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
          }
        } else {
          // bf16 rows are widened on the way: request all five pieces, then convert and store
          uint2 raw[5];
#pragma unroll
          for (int j = 0; j < 5; ++j) {
            const int t = tid + j * NT, row = t >> 3, sub = t & 7;
            const int y = y0 + row / VT, x = x0 + row % VT;
            const bool inl = t < (ZROW + 1) * 8 && row < ZROW && y < H && x < W;
            const uint32_t off = ((cur.lrow + (uint32_t)(y * W + x)) * (uint32_t)M + m) * 64u + (uint32_t)sub * 8u;
            raw[j] = make_uint2(0u, 0u);
            if (inl) raw[j] = *reinterpret_cast<const uint2 *>(reinterpret_cast<const char *>(value) + off);
          }
#pragma unroll
          for (int j = 0; j < 5; ++j) {
            const int t = tid + j * NT;
            if (t < (ZROW + 1) * 8)
              s_vt[t] = make_float4(bf16_to_f32(raw[j].x & 0xffffu), bf16_to_f32(raw[j].x >> 16),
                                    bf16_to_f32(raw[j].y & 0xffffu), bf16_to_f32(raw[j].y >> 16));
          }
        }
      }
      if (chunk == 0 && tid == 0) s_pos[(k + 1u) & 1u] = fetched;
      if (home) s_home[atomicAdd(&s_hcount[par], 1u)] = hrec;
      for (int t = tid; t < TP + 64; t += NT) s_cnt[par ^ 1][t] = 0;   // next chunk's counters
      if (tid == 0) s_hcount[par ^ 1] = 0u;
      MSDA_STAMP(3)
      // Exclusive prefix sum of the 256 pixel counts -> where each pixel's records start in
      // s_rec (packed: a chunk has at most 4 * NT contributions, so the array never overflows
      // however they pile up).  Every wave does the whole scan for itself -- no barrier -- and
      // writes start << 16 | count over the counts; a wave that still reads a count another
      // wave has already replaced finds it in the low bits all the same.
      {
        const int lane = tid & 63;
        uint4 c = *reinterpret_cast<const uint4 *>(&s_cnt[par][4 * lane]);
        c.x &= 0xffffu;
        c.y &= 0xffffu;
        c.z &= 0xffffu;
        c.w &= 0xffffu;
        const uint32_t s1 = c.x, s2 = s1 + c.y, s3 = s2 + c.z, tot = s3 + c.w;
        const uint32_t ex = wave_inclusive_scan(tot) - tot;
        *reinterpret_cast<uint4 *>(&s_cnt[par][4 * lane]) =
            make_uint4((ex << 16) | c.x, ((ex + s1) << 16) | c.y, ((ex + s2) << 16) | c.z,
                       ((ex + s3) << 16) | c.w);
      }
      __builtin_amdgcn_wave_barrier();
      {
        // corner weights (hh*hw, hh*lw, lh*hw, lh*lw) * attn, as corner_setup forms them
        const float hh = 1.f - hrec.x, hw = 1.f - hrec.y;
        const uint32_t rowb = slot * 128u;               // byte offset of the entry's row in s_go
        if (own1) s_rec[(s_cnt[par][r1] >> 16) + pos1] = make_uint2(rowb, __float_as_uint(hh * hw * hrec.z));
        if (own2) s_rec[(s_cnt[par][r1 + 1] >> 16) + pos2] = make_uint2(rowb, __float_as_uint(hh * hrec.y * hrec.z));
        if (own3) s_rec[(s_cnt[par][r1 + T] >> 16) + pos3] = make_uint2(rowb, __float_as_uint(hrec.x * hw * hrec.z));
        if (own4) s_rec[(s_cnt[par][r1 + T + 1] >> 16) + pos4] = make_uint2(rowb, __float_as_uint(hrec.x * hrec.y * hrec.z));
      }
      MSDA_STAMP(4)
      __syncthreads();
      MSDA_STAMP(7)
      if (nvalid) {                                       // consumed by the next chunk
        load_sample(ni, n_ent, c_xy, c_a);
        load_rows(ni, nc, n_r0, n_r1, c_g0, c_g1);
      }
      c_ent = n_ent;
      // ---------------- phase D: every pixel's owners sum their records ----------------
      // Four records per step, one 16-B piece of the four rows at a time, so the four row
      // reads of a piece are in flight together and the next piece's overlap the FMAs.  Records
      // past the pixel's last one (the next pixel's) are read as the zero row with weight 0.
      {
        const int px = tid >> 1, half = tid & 1;
        const uint32_t pc = s_cnt[par][px];
        const uint32_t base = pc >> 16, cnt = pc & 0xffffu;
        const char *gob = reinterpret_cast<const char *>(s_go);
        for (uint32_t i = 0; i < cnt; i += 4u) {
          uint2 r[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            r[j] = s_rec[base + i + (uint32_t)j];
            if (i + (uint32_t)j >= cnt) r[j] = make_uint2((uint32_t)CE * 128u, 0u);
          }
          const float w0 = __uint_as_float(r[0].y), w1 = __uint_as_float(r[1].y);
          const float w2 = __uint_as_float(r[2].y), w3 = __uint_as_float(r[3].y);
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const uint32_t cb = (uint32_t)(half * 4 + ((k + px) & 3)) * 16u;   // this lane's k-th piece (see flush)
            const float4 g0 = *reinterpret_cast<const float4 *>(gob + r[0].x + cb);
            const float4 g1 = *reinterpret_cast<const float4 *>(gob + r[1].x + cb);
            const float4 g2 = *reinterpret_cast<const float4 *>(gob + r[2].x + cb);
            const float4 g3 = *reinterpret_cast<const float4 *>(gob + r[3].x + cb);
            acc[k].x = fmaf(w3, g3.x, fmaf(w2, g2.x, fmaf(w1, g1.x, fmaf(w0, g0.x, acc[k].x))));
            acc[k].y = fmaf(w3, g3.y, fmaf(w2, g2.y, fmaf(w1, g1.y, fmaf(w0, g0.y, acc[k].y))));
            acc[k].z = fmaf(w3, g3.z, fmaf(w2, g2.z, fmaf(w1, g1.z, fmaf(w0, g0.z, acc[k].z))));
            acc[k].w = fmaf(w3, g3.w, fmaf(w2, g2.w, fmaf(w1, g1.w, fmaf(w0, g0.w, acc[k].w))));
          }
        }
      }
      MSDA_STAMP(5)
      // ---------------- phase E: gradients of the samples this tile is home to ----------------
      // Two lanes per sample, 16 channels each: dot the staged grad_output row with the four
      // corner rows of the LDS value tile, pair-sum with one DPP step, lane 0 finishes.
      {
        const uint32_t hc = (uint32_t)uni((int)s_hcount[par]);
        const int half = tid & 1;
        for (uint32_t h = (uint32_t)tid >> 1; h < hc; h += NT / 2) {
          const float4 hr = s_home[h];
          const uint32_t pk = __float_as_uint(hr.w);
          const int by = (int)(pk & 31u) - 1, bx = (int)((pk >> 5) & 31u) - 1;
          const uint32_t ok = (pk >> 10) & 15u, et = pk >> 14;      // et: examining thread
          const int hslot = (int)(et >> 2);
          const int rb = by * VT + bx;
          const int ra = (ok & 1u) ? rb : ZROW, rbb = (ok & 2u) ? rb + 1 : ZROW;
          const int rc = (ok & 4u) ? rb + VT : ZROW, rd = (ok & 8u) ? rb + VT + 1 : ZROW;
          // the four dot products as packed pairs (v_pk_fma_f32 on the register pairs the
          // ds_read_b128 results already sit in)
          v2f p1 = {0.f, 0.f}, p2 = {0.f, 0.f}, p3 = {0.f, 0.f}, p4 = {0.f, 0.f};
          const char *gob = reinterpret_cast<const char *>(s_go) + hslot * 128 + half * 64;
          const char *vtb = reinterpret_cast<const char *>(s_vt) + half * 64;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const int cb = ((k + (int)h) & 3) * 16;
            const v4f g4 = *reinterpret_cast<const v4f *>(gob + cb);
            const v4f v1 = *reinterpret_cast<const v4f *>(vtb + ra * 128 + cb);
            const v4f v2 = *reinterpret_cast<const v4f *>(vtb + rbb * 128 + cb);
            const v4f v3 = *reinterpret_cast<const v4f *>(vtb + rc * 128 + cb);
            const v4f v4 = *reinterpret_cast<const v4f *>(vtb + rd * 128 + cb);
            p1 = __builtin_elementwise_fma(g4.zw, v1.zw, __builtin_elementwise_fma(g4.xy, v1.xy, p1));
            p2 = __builtin_elementwise_fma(g4.zw, v2.zw, __builtin_elementwise_fma(g4.xy, v2.xy, p2));
            p3 = __builtin_elementwise_fma(g4.zw, v3.zw, __builtin_elementwise_fma(g4.xy, v3.xy, p3));
            p4 = __builtin_elementwise_fma(g4.zw, v4.zw, __builtin_elementwise_fma(g4.xy, v4.xy, p4));
          }
          const float d1 = sum2(p1.x + p1.y);
          const float d2 = sum2(p2.x + p2.y);
          const float d3 = sum2(p3.x + p3.y);
          const float d4 = sum2(p4.x + p4.y);
          if (half == 0) {
            const float lh = hr.x, lw = hr.y, a = hr.z, hh = 1.f - lh, hw = 1.f - lw;
            const uint32_t s = ((s_q[hslot] * (uint32_t)M + m) * (uint32_t)L + (uint32_t)l) * (uint32_t)P + (et & 3u);
            grad_aw[s] = hh * hw * d1 + hh * lw * d2 + lh * hw * d3 + lh * lw * d4;
            const float gw = (float)W * a * (hh * (d2 - d1) + lh * (d4 - d3));
            const float gh = (float)H * a * (hw * (d3 - d1) + lw * (d4 - d2));
            *reinterpret_cast<float2 *>(grad_loc + 2u * s) = make_float2(gw, gh);
          }
        }
      }
      MSDA_STAMP(6)
    }
    // ---------------- the tile's grad_value rows ----------------
    if (!cur.split) {                                    // sole owner: plain stores
      const int px = tid >> 1, half = tid & 1;
      const int py = px / T, pxx = px % T;
      if (py < th && pxx < tw) {
        const uint32_t rowi = cur.lrow + (uint32_t)((y0 + py) * W + x0 + pxx);
        float4 *dst = reinterpret_cast<float4 *>(
            reinterpret_cast<char *>(grad_value) +
            ((size_t)(rowi * (uint32_t)M + m) * 128u + (size_t)half * 64u));
#pragma unroll
        for (int k = 0; k < 4; ++k) dst[(k + px) & 3] = acc[k];
      }
    } else if (cur.total > 0u) {
      // One part of the tile's list: ADD to the zero-filled rows.  Float atomics run at full rate
      // only when a wave-instruction covers whole 128-B rows, so the sums go through LDS (the
      // value tile's space) and come back 32 lanes to a row.
      __syncthreads();                                   // phase E of the last chunk has read the value tile
      {
        const int px = tid >> 1, half = tid & 1;
#pragma unroll
        for (int k = 0; k < 4; ++k) s_vt[px * 8 + half * 4 + ((k + px) & 3)] = acc[k];
      }
      __syncthreads();
      const float *sums = reinterpret_cast<const float *>(s_vt);
      const int c = tid & 31;
      for (int px = tid >> 5; px < TP; px += NT / 32) {
        const int py = px / T, pxx = px % T;
        if (py < th && pxx < tw) {
          const uint32_t rowi = cur.lrow + (uint32_t)((y0 + py) * W + x0 + pxx);
          atomicAdd(grad_value + (size_t)(rowi * (uint32_t)M + m) * 32u + (size_t)c, sums[px * 32 + c]);
        }
      }
    }
    if (!has_next) break;
    pos = npos;
    npos = (uint32_t)uni((int)s_pos[(k + 1u) & 1u]);
    cur = decode(pos);
  }
  MSDA_STAMP_FLUSH
}

}  // namespace binned

// =======================================================================================
// Row f3: the op's neighbours in the caller (ops/modules/ms_deform_attn.py:102-109) for the
// BACKWARD of the fused boundary.  One thread per (image, query, head).
//   prologue: loc = reference_point + offset / (W_l, H_l), aw = softmax(logits over L*P) into the
//             workspace, for the backward kernels above (what the unfused caller keeps from its
//             forward);
//   epilogue: in place, grad_loc -> grad_offset = grad_loc / (W_l, H_l) and
//             grad_aw -> grad_logit = aw * (grad_aw - sum_i aw_i * grad_aw_i).
// =======================================================================================
constexpr int FUSED_PB = 64;        // pairs per workgroup of the prologue / epilogue kernels
constexpr int FUSED_MAX_LP = 64;    // L * P they stage in LDS

// Coalesced: a workgroup takes FUSED_PB consecutive pairs = FUSED_PB * L*P consecutive samples;
// the per-pair softmax statistics go through LDS, everything else is one sample per thread.
__global__ __launch_bounds__(256) void msda_fused_prologue_kernel(
    const int64_t *__restrict__ shapes, const float *__restrict__ refp, const float *__restrict__ off,
    const float *__restrict__ logits, int M, int L, int P, size_t npairs, float *__restrict__ loc,
    float *__restrict__ aw) {
  __shared__ float s_lg[FUSED_PB * FUSED_MAX_LP];
  __shared__ float s_mx[FUSED_PB], s_sum[FUSED_PB];
  const int LP = L * P, tid = threadIdx.x;
  const size_t pair0 = (size_t)blockIdx.x * FUSED_PB;
  const int npb = (int)min((size_t)FUSED_PB, npairs - pair0), nsb = npb * LP;
  const size_t s0 = pair0 * (size_t)LP;
  for (int i = tid; i < nsb; i += 256) s_lg[i] = logits[s0 + i];
  __syncthreads();
  if (tid < npb) {
    const float *lg = s_lg + tid * LP;
    float mx = lg[0];
    for (int t = 1; t < LP; ++t) mx = fmaxf(mx, lg[t]);
    float sum = 0.f;
    for (int t = 0; t < LP; ++t) sum += expf(lg[t] - mx);
    s_mx[tid] = mx;
    s_sum[tid] = sum;
  }
  __syncthreads();
  for (int i = tid; i < nsb; i += 256) {
    const int pl = i / LP, j = i - pl * LP, l = j / P;
    const size_t qi = (pair0 + pl) / (size_t)M;
    const float H = (float)shapes[2 * l], W = (float)shapes[2 * l + 1];
    const float2 rp = *reinterpret_cast<const float2 *>(refp + 2 * (qi * L + l));
    const float2 o = *reinterpret_cast<const float2 *>(off + 2 * (s0 + i));
    *reinterpret_cast<float2 *>(loc + 2 * (s0 + i)) =
        make_float2(__fadd_rn(rp.x, __fdiv_rn(o.x, W)), __fadd_rn(rp.y, __fdiv_rn(o.y, H)));
    aw[s0 + i] = __fdiv_rn(expf(s_lg[i] - s_mx[pl]), s_sum[pl]);
  }
}

__global__ __launch_bounds__(256) void msda_fused_epilogue_kernel(
    const int64_t *__restrict__ shapes, const float *__restrict__ aw, int L, int P, size_t npairs,
    float *__restrict__ grad_loc, float *__restrict__ grad_aw) {
  __shared__ float s_prod[FUSED_PB * FUSED_MAX_LP];
  __shared__ float s_dot[FUSED_PB];
  const int LP = L * P, tid = threadIdx.x;
  const size_t pair0 = (size_t)blockIdx.x * FUSED_PB;
  const int npb = (int)min((size_t)FUSED_PB, npairs - pair0), nsb = npb * LP;
  const size_t s0 = pair0 * (size_t)LP;
  for (int i = tid; i < nsb; i += 256) s_prod[i] = aw[s0 + i] * grad_aw[s0 + i];
  __syncthreads();
  if (tid < npb) {
    float dot = 0.f;
    for (int t = 0; t < LP; ++t) dot += s_prod[tid * LP + t];
    s_dot[tid] = dot;
  }
  __syncthreads();
  for (int i = tid; i < nsb; i += 256) {
    const int pl = i / LP, j = i - pl * LP, l = j / P;
    const float H = (float)shapes[2 * l], W = (float)shapes[2 * l + 1];
    // aw * (grad_aw - dot)
    grad_aw[s0 + i] = s_prod[i] - aw[s0 + i] * s_dot[pl];
    float2 *g = reinterpret_cast<float2 *>(grad_loc + 2 * (s0 + i));
    const float2 v = *g;
    *g = make_float2(__fdiv_rn(v.x, W), __fdiv_rn(v.y, H));
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

// Which D == 32 forward: queries-by-head (default) or, with MSDA_FWD_IMPL=plain, the
// heads-of-a-query kernel (A/B measurements; also the path of L*P > 32 and unaligned tensors).
int fwd_impl_choice() {
  const char *e = getenv("MSDA_FWD_IMPL");
  return (e && strcmp(e, "plain") == 0) ? 0 : 1;
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
                       aligned16(loc) && (reinterpret_cast<uintptr_t>(aw) & 7u) == 0, out);
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
  if (D == 32 && small && aligned16(value) && aligned16(out)) {
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

// uint32 words: counters[key_cap + 1] + work counter, list starts[key_cap + 1] + pad, descriptors[quads],
// entries[16 * quads];
// key_cap: one list per (image, head, tile), and a level has at most H*W tiles.
size_t binned_key_cap(int N, int S, int M) { return (size_t)N * M * S; }
size_t binned_workspace_bytes(int N, int S, int M, int L, int Lq, int P) {
  (void)P;
  const size_t words = 2 * (binned_key_cap(N, S, M) + 2) +
                       (size_t)(1 + binned::MAX_ENTRIES_PER_QUAD) * N * Lq * M * L;
  return (words * 4 + 255) & ~(size_t)255;
}

template <typename VT>
int launch_backward_binned(const char *who, const VT *value, const int64_t *shapes,
                           const int64_t *lsi, const float *loc, const float *aw,
                           const VT *grad_out, int N, int S, int M, int L, int Lq, int P,
                           float *grad_value, float *grad_loc, float *grad_aw, void *workspace,
                           size_t workspace_bytes, hipStream_t st) {
  const size_t need = binned_workspace_bytes(N, S, M, L, Lq, P);
  if (workspace == nullptr || workspace_bytes < need || (reinterpret_cast<uintptr_t>(workspace) & 3u))
    return fail(MSDA_ERR_INVALID_ARGUMENT,
                "%s: needs a 4-byte aligned device workspace of at least %zu bytes (got %zu)", who,
                need, workspace ? workspace_bytes : (size_t)0);
  const uint32_t key_cap = (uint32_t)binned_key_cap(N, S, M);
  uint32_t *counts = static_cast<uint32_t *>(workspace);
  uint32_t *work = counts + (size_t)key_cap + 1;
  uint32_t *offs = counts + (size_t)key_cap + 2;
  uint32_t *desc = offs + (size_t)key_cap + 2;
  uint32_t *entries = desc + (size_t)N * Lq * M * L;
  const uint32_t spi = (uint32_t)((size_t)Lq * M * L);          // quads per image
  const uint32_t bpi = (spi + binned::BIN_QB - 1) / binned::BIN_QB;
  const int cus = num_cus();
  hipLaunchKernelGGL(binned::msda_bwd_bin_zero_kernel, dim3(cus), dim3(256), 0, st, shapes, N, M, L,
                     key_cap, counts);
  if (int rc = check_launch(who)) return rc;
  const uint64_t magic_ml = (((uint64_t)1 << 40) + (uint64_t)(M * L) - 1) / (uint64_t)(M * L);
  const uint64_t magic_l = (((uint64_t)1 << 40) + (uint64_t)L - 1) / (uint64_t)L;
  hipLaunchKernelGGL(binned::msda_bwd_bin_kernel<false>, dim3((uint32_t)N * bpi), dim3(binned::BIN_NT),
                     0, st, shapes, loc, N, M, L, Lq, P, spi, bpi, magic_ml, magic_l, key_cap, counts, desc,
                     entries, grad_loc, grad_aw);
  if (int rc = check_launch(who)) return rc;
  hipLaunchKernelGGL(binned::msda_bwd_bin_scan_kernel, dim3(1), dim3(1024), 0, st, shapes, N, M, L,
                     key_cap, counts, offs);
  if (int rc = check_launch(who)) return rc;
  hipLaunchKernelGGL(binned::msda_bwd_bin_kernel<true>, dim3((uint32_t)N * bpi), dim3(binned::BIN_NT),
                     0, st, shapes, loc, N, M, L, Lq, P, spi, bpi, magic_ml, magic_l, key_cap, counts, desc,
                     entries, grad_loc, grad_aw);
  if (int rc = check_launch(who)) return rc;
  // Small batches: fewer than ~8 tiles per workgroup means the kernel runs as long as its longest
  // list (coarse-level tiles), so those lists are split over workgroups that add their partial
  // sums with float atomics into a zero-filled grad_value.
  const size_t est_items = (size_t)N * M * (size_t)((S + binned::TP - 1) / binned::TP);
  const int split_enable = est_items < (size_t)16 * cus ? 1 : 0;
  if (split_enable) {
    const hipError_t e = hipMemsetAsync(grad_value, 0, sizeof(float) * (size_t)N * S * M * 32, st);
    if (e != hipSuccess)
      return fail(MSDA_ERR_LAUNCH, "%s: hipMemsetAsync(grad_value) failed: %s", who, hipGetErrorString(e));
  }
  hipLaunchKernelGGL(binned::msda_bwd_binned_kernel<VT>, dim3(2 * cus), dim3(binned::NT), 0, st,
                     grad_out, value, shapes, lsi, loc, aw, offs, entries, offs + (size_t)key_cap + 1, work,
                     key_cap, N, S, M, L, P, split_enable, grad_value, grad_loc, grad_aw);
  return check_launch(who);
}

// Which D == 32 backward: binned (default) or, with MSDA_BWD_IMPL=generic, the atomic kernel
// (A/B measurements).
int bwd_impl_choice() {
  const char *e = getenv("MSDA_BWD_IMPL");
  return (e && strcmp(e, "generic") == 0) ? 0 : 2;
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

template <typename VT>
int backward_d32(const char *who, const VT *value, const int64_t *shapes, const int64_t *lsi,
                 const float *loc, const float *aw, const VT *grad_out, int N, int S, int M, int L,
                 int Lq, int P, float *grad_value, float *grad_loc, float *grad_aw, void *workspace,
                 size_t workspace_bytes, bool own_workspace, hipStream_t st) {
  if (!own_workspace)
    return launch_backward_binned<VT>(who, value, shapes, lsi, loc, aw, grad_out, N, S, M, L, Lq, P,
                                      grad_value, grad_loc, grad_aw, workspace, workspace_bytes, st);
  PoolWorkspace pool(st);
  const size_t need = binned_workspace_bytes(N, S, M, L, Lq, P);
  if (int rc = pool.get(who, need)) return rc;
  return launch_backward_binned<VT>(who, value, shapes, lsi, loc, aw, grad_out, N, S, M, L, Lq, P,
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
    return backward_d32<float>(who, value, shapes, lsi, loc, aw, grad_out, N, S, M, L, Lq, P,
                               grad_value, grad_loc, grad_aw, workspace, workspace_bytes,
                               own_workspace, st);
  return backward_atomic<float>(who, value, shapes, lsi, loc, aw, grad_out, N, S, M, D, L, Lq, P,
                                grad_value, grad_loc, grad_aw, st);
}

int backward_bf16(const char *who, const uint16_t *value, const int64_t *shapes, const int64_t *lsi,
                  const float *loc, const float *aw, const uint16_t *grad_out, int N, int S, int M,
                  int D, int L, int Lq, int P, float *grad_value, float *grad_loc, float *grad_aw,
                  void *workspace, size_t workspace_bytes, bool own_workspace, void *stream) {
  if (int rc = check_dims(who, N, S, M, D, L, Lq, P)) return rc;
  if (!binned_shape(N, S, M, D, L, Lq, P) || (reinterpret_cast<uintptr_t>(value) & 7u) ||
      (reinterpret_cast<uintptr_t>(grad_out) & 7u) || !aligned16(grad_value) || !aligned16(loc) ||
      !aligned16(grad_loc))
    return fail(MSDA_ERR_INVALID_ARGUMENT,
                "%s: needs channels == 32, num_point <= %d, num_levels <= %d, a non-empty batch and "
                "aligned tensors of < 2^30 elements",
                who, binned::MAXP, binned::MAXL);
  return backward_d32<bf16_t>(who, reinterpret_cast<const bf16_t *>(value), shapes, lsi, loc, aw,
                              reinterpret_cast<const bf16_t *>(grad_out), N, S, M, L, Lq, P,
                              grad_value, grad_loc, grad_aw, workspace, workspace_bytes,
                              own_workspace, static_cast<hipStream_t>(stream));
}


// ---- row f3: fused boundary (fp32) ----
size_t align256(size_t b) { return (b + 255) & ~(size_t)255; }

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
  constexpr uint32_t GP = 64u / FwdLanes<VT>::LPG;
  const size_t lds = (size_t)4 * ((size_t)2 * GP * (size_t)(L * P + 1) * 16 + (((size_t)GP * L * P + 3) / 4) * 16);
  const uint32_t nq = (uint32_t)((size_t)N * Lq), nunits = ((nq + GP - 1u) / GP) * (uint32_t)M;
  const uint32_t nblk = (nunits + 3u) / 4u;
  hipLaunchKernelGGL((msda_fwd_d32_qh_kernel<VT, true>), dim3(nblk), dim3(256), lds,
                     static_cast<hipStream_t>(stream), value, shapes, lsi, off, logits, refp, S, M, L, Lq, P,
                     nq, nunits, nblk, false, out);
  return check_launch(who);
}

size_t fused_backward_workspace_bytes(int N, int S, int M, int D, int L, int Lq, int P) {
  const size_t samples = (size_t)N * Lq * M * L * P;
  size_t inner = 0;
  if (binned_shape(N, S, M, D, L, Lq, P) && bwd_impl_choice() != 0)
    inner = binned_workspace_bytes(N, S, M, L, Lq, P);
  return align256(samples * 8) + align256(samples * 4) + align256(inner);
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
  void *inner = static_cast<char *>(workspace) + align256(samples * 8) + align256(samples * 4);
  const size_t inner_bytes = need - align256(samples * 8) - align256(samples * 4);
  const size_t nblk = (npairs + FUSED_PB - 1) / FUSED_PB;
  if (nblk > 0x7fffffffull) return fail(MSDA_ERR_INVALID_ARGUMENT, "%s: problem too large", who);
  if (L * P > FUSED_MAX_LP)
    return fail(MSDA_ERR_INVALID_ARGUMENT, "%s: num_levels * num_point must be <= %d", who, FUSED_MAX_LP);
  if (npairs) {
    hipLaunchKernelGGL(msda_fused_prologue_kernel, dim3((unsigned)nblk), dim3(256), 0, st, shapes, refp, off,
                       logits, M, L, P, npairs, loc, aw);
    if (int rc = check_launch(who)) return rc;
  }
  const int rc = bf16 ? backward_bf16(who, static_cast<const uint16_t *>(value), shapes, lsi, loc, aw,
                                      static_cast<const uint16_t *>(grad_out), N, S, M, D, L, Lq, P, grad_value,
                                      grad_off, grad_logits, inner, inner_bytes, false, stream)
                      : backward_f32(who, static_cast<const float *>(value), shapes, lsi, loc, aw,
                                     static_cast<const float *>(grad_out), N, S, M, D, L, Lq, P, grad_value,
                                     grad_off, grad_logits, inner, inner_bytes, false, stream);
  if (rc) return rc;
  if (npairs) {
    hipLaunchKernelGGL(msda_fused_epilogue_kernel, dim3((unsigned)nblk), dim3(256), 0, st, shapes, aw, L, P,
                       npairs, grad_off, grad_logits);
    if (int rc = check_launch(who)) return rc;
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

int msda_abi_version(void) { return 3; }

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
                       num_levels, num_query, num_point, grad_value, grad_sampling_loc,
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
                       channels, num_levels, num_query, num_point, grad_value, grad_sampling_loc,
                       grad_attn_weight, workspace, workspace_bytes, false, stream);
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
