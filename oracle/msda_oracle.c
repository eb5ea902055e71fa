/*
 * oracle/msda_oracle.c -- TEST INFRASTRUCTURE ONLY.
 *
 * Scalar CPU restatement of multi-scale deformable attention (MSDA) forward
 * and backward, used as the checker for the HIP kernels.  Nothing in the
 * product path (instaorder_amd/, MultiScaleDeformableAttention/) may import,
 * link or call this file; only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg do.
 *
 * Algorithm followed (reference = /root/reference, paths relative to
 * mask2former/modeling/pixel_decoder/ops/):
 *   - forward  : src/cuda/ms_deform_im2col_cuda.cuh:242-304 (per output element
 *                loop over levels and points) with the 4-corner interpolation of
 *                :38-89; pixel coordinate h_im = loc_y*H - 0.5, w_im = loc_x*W - 0.5
 *                (:290-291); a sample only contributes when
 *                h_im > -1 && w_im > -1 && h_im < H && w_im < W (:293); each corner
 *                is dropped when it lies outside the level (:61,67,73,79) which is
 *                grid_sample(padding_mode='zeros', align_corners=False) as used by
 *                functions/ms_deform_attn_func.py:52-72.
 *   - backward : src/cuda/ms_deform_im2col_cuda.cuh:92-164 (per corner
 *                grad_value += w_k * grad_out * attn_weight; d/d attn = grad_out *
 *                interpolated value; d/d loc_x = W * grad_w_weight * grad_out * attn,
 *                d/d loc_y = H * grad_h_weight * grad_out * attn), summed over the
 *                D channels of the head (:382-398).
 * Layouts (same as the reference, all contiguous):
 *   value [N,S,M,D]; spatial_shapes int64 [L,2] as (H,W); level_start_index int64 [L];
 *   sampling_loc [N,Lq,M,L,P,2] as (x,y) in [0,1]; attn_weight [N,Lq,M,L,P];
 *   output / grad_output [N,Lq,M*D].
 *
 * Pinned by tests/golden/ (vectors generated from the reference's own
 * ms_deform_attn_core_pytorch + fp64 autograd by tests/golden/make_golden.py).
 *
 * Parallelism: OpenMP over (n, m) so that grad_value rows are private to a
 * thread; results do not depend on the thread count.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

#define MSDA_ORACLE_DEFINE(T, SUFFIX)                                                        \
  void msda_oracle_forward_##SUFFIX(const T *value, const int64_t *shapes,                   \
                                    const int64_t *lsi, const T *loc, const T *aw, int N,    \
                                    int S, int M, int D, int L, int Lq, int P, T *out) {     \
    _Pragma("omp parallel for collapse(2) schedule(static)")                                 \
    for (int n = 0; n < N; ++n)                                                              \
      for (int m = 0; m < M; ++m)                                                            \
        for (int q = 0; q < Lq; ++q) {                                                       \
          const size_t pair = ((size_t)n * Lq + q) * M + m;                                  \
          T *o = out + pair * D;                                                             \
          for (int c = 0; c < D; ++c) o[c] = 0;                                              \
          for (int l = 0; l < L; ++l) {                                                      \
            const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];                    \
            const T *vl = value + ((size_t)n * S + (size_t)lsi[l]) * M * D + (size_t)m * D;  \
            for (int p = 0; p < P; ++p) {                                                    \
              const size_t si = (pair * L + l) * P + p;                                      \
              const T w_im = loc[2 * si] * W - (T)0.5;                                       \
              const T h_im = loc[2 * si + 1] * H - (T)0.5;                                   \
              const T a = aw[si];                                                            \
              if (!(h_im > -1 && w_im > -1 && h_im < H && w_im < W)) continue;               \
              const int h0 = (int)floor(h_im), w0 = (int)floor(w_im);                        \
              const int h1 = h0 + 1, w1 = w0 + 1;                                            \
              const T lh = h_im - h0, lw = w_im - w0, hh = 1 - lh, hw = 1 - lw;              \
              const T k1 = hh * hw, k2 = hh * lw, k3 = lh * hw, k4 = lh * lw;                \
              const T *p1 = (h0 >= 0 && w0 >= 0) ? vl + ((size_t)h0 * W + w0) * M * D : 0;   \
              const T *p2 = (h0 >= 0 && w1 <= W - 1) ? vl + ((size_t)h0 * W + w1) * M * D : 0; \
              const T *p3 = (h1 <= H - 1 && w0 >= 0) ? vl + ((size_t)h1 * W + w0) * M * D : 0; \
              const T *p4 = (h1 <= H - 1 && w1 <= W - 1) ? vl + ((size_t)h1 * W + w1) * M * D : 0; \
              for (int c = 0; c < D; ++c) {                                                  \
                const T v1 = p1 ? p1[c] : 0, v2 = p2 ? p2[c] : 0;                            \
                const T v3 = p3 ? p3[c] : 0, v4 = p4 ? p4[c] : 0;                            \
                o[c] += (k1 * v1 + k2 * v2 + k3 * v3 + k4 * v4) * a;                         \
              }                                                                              \
            }                                                                                \
          }                                                                                  \
        }                                                                                    \
  }                                                                                          \
                                                                                             \
  void msda_oracle_backward_##SUFFIX(const T *value, const int64_t *shapes,                  \
                                     const int64_t *lsi, const T *loc, const T *aw,          \
                                     const T *grad_out, int N, int S, int M, int D, int L,   \
                                     int Lq, int P, T *grad_value, T *grad_loc, T *grad_aw) { \
    memset(grad_value, 0, sizeof(T) * (size_t)N * S * M * D);                                \
    memset(grad_loc, 0, sizeof(T) * (size_t)N * Lq * M * L * P * 2);                         \
    memset(grad_aw, 0, sizeof(T) * (size_t)N * Lq * M * L * P);                              \
    _Pragma("omp parallel for collapse(2) schedule(static)")                                 \
    for (int n = 0; n < N; ++n)                                                              \
      for (int m = 0; m < M; ++m)                                                            \
        for (int q = 0; q < Lq; ++q) {                                                       \
          const size_t pair = ((size_t)n * Lq + q) * M + m;                                  \
          const T *go = grad_out + pair * D;                                                 \
          for (int l = 0; l < L; ++l) {                                                      \
            const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];                    \
            const size_t lbase = ((size_t)n * S + (size_t)lsi[l]) * M * D + (size_t)m * D;   \
            const T *vl = value + lbase;                                                     \
            T *gl = grad_value + lbase;                                                      \
            for (int p = 0; p < P; ++p) {                                                    \
              const size_t si = (pair * L + l) * P + p;                                      \
              const T w_im = loc[2 * si] * W - (T)0.5;                                       \
              const T h_im = loc[2 * si + 1] * H - (T)0.5;                                   \
              const T a = aw[si];                                                            \
              if (!(h_im > -1 && w_im > -1 && h_im < H && w_im < W)) continue;               \
              const int h0 = (int)floor(h_im), w0 = (int)floor(w_im);                        \
              const int h1 = h0 + 1, w1 = w0 + 1;                                            \
              const T lh = h_im - h0, lw = w_im - w0, hh = 1 - lh, hw = 1 - lw;              \
              const T k1 = hh * hw, k2 = hh * lw, k3 = lh * hw, k4 = lh * lw;                \
              const int ok1 = (h0 >= 0 && w0 >= 0), ok2 = (h0 >= 0 && w1 <= W - 1);          \
              const int ok3 = (h1 <= H - 1 && w0 >= 0), ok4 = (h1 <= H - 1 && w1 <= W - 1);  \
              const size_t o1 = ((size_t)h0 * W + w0) * M * D, o2 = ((size_t)h0 * W + w1) * M * D; \
              const size_t o3 = ((size_t)h1 * W + w0) * M * D, o4 = ((size_t)h1 * W + w1) * M * D; \
              T ga = 0, gw = 0, gh = 0;                                                      \
              for (int c = 0; c < D; ++c) {                                                  \
                const T tg = go[c], tgv = tg * a;                                            \
                T ghw = 0, gww = 0, v1 = 0, v2 = 0, v3 = 0, v4 = 0;                          \
                if (ok1) { v1 = vl[o1 + c]; ghw -= hw * v1; gww -= hh * v1; gl[o1 + c] += k1 * tgv; } \
                if (ok2) { v2 = vl[o2 + c]; ghw -= lw * v2; gww += hh * v2; gl[o2 + c] += k2 * tgv; } \
                if (ok3) { v3 = vl[o3 + c]; ghw += hw * v3; gww -= lh * v3; gl[o3 + c] += k3 * tgv; } \
                if (ok4) { v4 = vl[o4 + c]; ghw += lw * v4; gww += lh * v4; gl[o4 + c] += k4 * tgv; } \
                ga += tg * (k1 * v1 + k2 * v2 + k3 * v3 + k4 * v4);                          \
                gw += W * gww * tgv;                                                         \
                gh += H * ghw * tgv;                                                         \
              }                                                                              \
              grad_aw[si] = ga;                                                              \
              grad_loc[2 * si] = gw;                                                         \
              grad_loc[2 * si + 1] = gh;                                                     \
            }                                                                                \
          }                                                                                  \
        }                                                                                    \
  }

MSDA_ORACLE_DEFINE(double, f64)
MSDA_ORACLE_DEFINE(float, f32)
