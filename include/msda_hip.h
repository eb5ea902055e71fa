/*
 * msda_hip.h -- C ABI of libmsda_hip.so: multi-scale deformable attention (MSDA)
 * forward / backward for AMD MI355X (gfx950), hand-written HIP.
 *
 * This is the drop-in boundary for the one native component of the reference
 * (SNU-VGILab/InstaOrder): the two host launchers that its pybind module
 * `MultiScaleDeformableAttention` calls.  Paths below are relative to
 * mask2former/modeling/pixel_decoder/ops/ in the reference.
 *
 * Conventions (all entry points):
 *   - every pointer is a DEVICE pointer to a contiguous array; inputs are read-only;
 *   - layouts are the reference's: value [N,S,M,D], spatial_shapes int64 [L,2] = (H,W),
 *     level_start_index int64 [L], sampling_loc [N,Lq,M,L,P,2] = (x,y) normalised to [0,1],
 *     attn_weight [N,Lq,M,L,P], output / grad_output [N,Lq,M*D];
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream); work is
 *     enqueued asynchronously, nothing synchronises with the host;
 *   - no exceptions; return 0 on success, non-zero on error, and msda_last_error() then
 *     returns a thread-local message;
 *   - no allocation, with one exception: msda_backward_f32 / msda_backward_bf16 on the
 *     channels == 32 fast path take their workspace from the device's stream-ordered pool
 *     (hipMallocAsync / hipFreeAsync on `stream`, no host synchronisation); callers that must
 *     not allocate pass their own workspace to the *_ws variants;
 *   - every output element is written (no pre-zeroing needed by the caller).
 */
#ifndef MSDA_HIP_H_
#define MSDA_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MSDA_OK 0
#define MSDA_ERR_INVALID_ARGUMENT 1
#define MSDA_ERR_LAUNCH 2

/* Library ABI version (bumped when a signature changes or an entry point is added; 4: msda_backward_bf16_gv16_ws,
 * msda_backward_bf16_gv16_workspace_bytes, msda_backward_min_workspace_bytes; 3 had msda_backward_f32_noalloc;
 * 5 adds msda_backward_compact_workspace_bytes and takes short workspaces in rounds). */
int msda_abi_version(void);

/* Thread-local text of the last error returned on this thread ("" if none). */
const char *msda_last_error(void);

/* Number of kernel launches issued by this process through this library since load
 * (tests use it to prove the HIP path ran: the reference's caller swallows every
 * exception and silently falls back to PyTorch, ops/modules/ms_deform_attn.py:116-121). */
uint64_t msda_launch_count(void);

/*
 * Forward.  Replaces ms_deformable_im2col_cuda<float|double>
 * (src/cuda/ms_deform_im2col_cuda.cuh:928-959) as called from
 * ms_deform_attn_cuda_forward (src/cuda/ms_deform_attn_cuda.cu:66-80):
 *   output[n,q,m*D+c] = sum_{l,p} attn_weight[n,q,m,l,p] *
 *                        bilinear(value_l[n,:,m,c], x*W_l-0.5, y*H_l-0.5)   (zero padding)
 */
int msda_forward_f32(const float *value, const int64_t *spatial_shapes,
                     const int64_t *level_start_index, const float *sampling_loc,
                     const float *attn_weight, int batch, int spatial_size, int num_heads,
                     int channels, int num_levels, int num_query, int num_point,
                     float *output, void *stream);

int msda_forward_f64(const double *value, const int64_t *spatial_shapes,
                     const int64_t *level_start_index, const double *sampling_loc,
                     const double *attn_weight, int batch, int spatial_size, int num_heads,
                     int channels, int num_levels, int num_query, int num_point,
                     double *output, void *stream);

/*
 * Backward.  Replaces ms_deformable_col2im_cuda<float|double>
 * (src/cuda/ms_deform_im2col_cuda.cuh:961-1332, every kernel variant it dispatches to)
 * as called from ms_deform_attn_cuda_backward (src/cuda/ms_deform_attn_cuda.cu:136-153).
 * grad_value needs no zero-fill by the caller (the reference's caller does at::zeros_like,
 * ms_deform_attn_cuda.cu:126); grad_sampling_loc and grad_attn_weight are fully written.
 *
 * channels == 32, num_point <= 4, num_levels <= 8 (every InstaFormer configuration) runs the
 * binned destination-tiled kernels, which sort the samples by destination tile in a workspace;
 * other shapes run the generic atomic kernels and need none (both queries then return 0).  The
 * *_ws variants take the workspace from the caller (any device memory, 4-byte aligned, contents
 * irrelevant before and after; calls sharing one workspace must be ordered on one stream); the
 * plain variants take msda_backward_workspace_bytes(...) from the stream-ordered pool.
 * Workspace size: msda_backward_workspace_bytes(...) holds the lists of the whole batch at their
 * worst case (16 entries per (query, head, level); real locations fill 1.4 - 4.5), the fastest:
 * one pass of the kernels.  A caller short of memory may pass any size from
 * msda_backward_min_workspace_bytes(...) upwards; what is there is used in two ways, chosen by the
 * library: the lists are written and consumed in several ROUNDS through a smaller entry region
 * (as many rounds are launched as the worst case needs; a round that finds no lists left returns
 * at once: with 5 or more entries per quad only the first does anything, and the others cost two
 * empty launches each), and the batch is taken in groups of as many images as run in at most four
 * rounds, one group after the other on the stream.  Same results.  The minimum is one image at one
 * entry per quad (sixteen rounds): about a tenth of one image's worst case.
 *
 * Level layout (read on the device; the reference checks none of it).  Every level must be non-empty
 * and lie inside [0, spatial_size); where one does not, the reference reads (forward) and writes
 * (backward) out of bounds, and every entry point here returns ZEROS instead -- the output, or all
 * three gradients -- with rc 0 and without touching memory outside its tensors.  Rows of
 * [0, spatial_size) that belong to no level (spatial_size > sum H*W, or gaps in level_start_index)
 * get zero grad_value, as the reference's at::zeros_like gives them.  Levels that share rows: the
 * channels == 32 backward returns zeros in all three gradients; the other kernels compute (shared
 * rows receive both levels' contributions).
 */
size_t msda_backward_workspace_bytes(int batch, int spatial_size, int num_heads, int channels,
                                     int num_levels, int num_query, int num_point);
size_t msda_backward_min_workspace_bytes(int batch, int spatial_size, int num_heads, int channels,
                                         int num_levels, int num_query, int num_point);
/* A size in between, for callers that care about memory: four entries per quad for the whole batch (four rounds,
 * of which model-like locations, 1.35 entries per quad, use the first).  Measured bs 32 at 1024^2: 0.37 GB instead
 * of 1.17 GB, backward 2.90 instead of 2.89 ms; bs 2 at 1536^2: 0.498 instead of 0.487 ms (the empty rounds' six
 * launches).  bf16_grad_value != 0: for msda_backward_bf16_gv16_ws (adds the fp32 staging tensor of a small
 * batch). */
size_t msda_backward_compact_workspace_bytes(int batch, int spatial_size, int num_heads, int channels,
                                             int num_levels, int num_query, int num_point, int bf16_grad_value);

int msda_backward_f32(const float *value, const int64_t *spatial_shapes,
                      const int64_t *level_start_index, const float *sampling_loc,
                      const float *attn_weight, const float *grad_output, int batch,
                      int spatial_size, int num_heads, int channels, int num_levels,
                      int num_query, int num_point, float *grad_value,
                      float *grad_sampling_loc, float *grad_attn_weight, void *stream);

int msda_backward_f32_ws(const float *value, const int64_t *spatial_shapes,
                         const int64_t *level_start_index, const float *sampling_loc,
                         const float *attn_weight, const float *grad_output, int batch,
                         int spatial_size, int num_heads, int channels, int num_levels,
                         int num_query, int num_point, float *grad_value,
                         float *grad_sampling_loc, float *grad_attn_weight, void *workspace,
                         size_t workspace_bytes, void *stream);

/*
 * msda_backward_f32 for callers that cannot allocate (code under HIP graph capture, tight serving
 * loops, harnesses that forbid hipMalloc): same arguments, same results, and no hipMalloc /
 * hipFree / pool allocation anywhere.  No counterpart in the reference (its launcher allocates
 * through ATen, src/cuda/ms_deform_attn_cuda.cu:126-128).  The sort-by-tile workspace is a
 * 64 MiB array built into the library's code object (each GPU the library runs on holds its copy
 * until the process ends); the batch goes through it in rounds and groups of images as through any
 * short workspace (above: five 1024^2 images per group, one 1536^2 image in two rounds), so any
 * batch size is taken, and only a problem whose ONE image needs more --
 * msda_backward_min_workspace_bytes(...) > msda_backward_noalloc_max_workspace_bytes(): encoder
 * shapes beyond about 3400^2 -- is rejected with MSDA_ERR_INVALID_ARGUMENT (use msda_backward_f32_ws).
 * Shapes off the channels == 32 fast path
 * run the generic kernels, which need no workspace.  All calls on one GPU share that GPU's array:
 * calls that overlap on one GPU (different streams or threads) race; order them on one stream.
 */
int msda_backward_f32_noalloc(const float *value, const int64_t *spatial_shapes,
                              const int64_t *level_start_index, const float *sampling_loc,
                              const float *attn_weight, const float *grad_output, int batch,
                              int spatial_size, int num_heads, int channels, int num_levels,
                              int num_query, int num_point, float *grad_value,
                              float *grad_sampling_loc, float *grad_attn_weight, void *stream);

size_t msda_backward_noalloc_max_workspace_bytes(void);

int msda_backward_f64(const double *value, const int64_t *spatial_shapes,
                      const int64_t *level_start_index, const double *sampling_loc,
                      const double *attn_weight, const double *grad_output, int batch,
                      int spatial_size, int num_heads, int channels, int num_levels,
                      int num_query, int num_point, double *grad_value,
                      double *grad_sampling_loc, double *grad_attn_weight, void *stream);

/*
 * bf16 variant (BASELINE.json configs[2]; not reference behaviour: the reference dispatches
 * float/double only, src/cuda/ms_deform_attn_cuda.cu:69,139, and its caller forces fp32,
 * pixel_decoder/msdeformattn.py:435-441).  value / output / grad_output are bfloat16 bit
 * patterns (uint16_t); sampling locations, attention weights and ALL THREE gradients are fp32;
 * accumulation is fp32.  msda_backward_bf16[_ws] hand grad_value back in fp32 (algorithmic bytes
 * 68.00 B/sample); msda_backward_bf16_gv16_ws hands it back in bfloat16 -- the tile that owns a
 * row rounds its fp32 register sums once, at the store -- which is BASELINE.md's bf16 variant
 * (62.67 B/sample) and what an autograd caller wants (value's dtype, no conversion pass).  Its
 * workspace is msda_backward_bf16_gv16_workspace_bytes(...): the lists, plus -- small batches
 * only, whose long lists are split over workgroups that add in fp32 -- an fp32 staging tensor
 * of grad_value's shape; 16-byte aligned; smaller workspaces down to
 * msda_backward_min_workspace_bytes(...) as above (a group of images whose staging tensor does
 * not fit behind its lists runs unsplit: same results, the long lists are not shared out).  All need channels == 32 (backward also
 * num_point <= 4, num_levels <= 8).  Same kernels as the fp32 fast paths.
 */
int msda_forward_bf16(const uint16_t *value, const int64_t *spatial_shapes,
                      const int64_t *level_start_index, const float *sampling_loc,
                      const float *attn_weight, int batch, int spatial_size, int num_heads,
                      int channels, int num_levels, int num_query, int num_point,
                      uint16_t *output, void *stream);

int msda_backward_bf16(const uint16_t *value, const int64_t *spatial_shapes,
                       const int64_t *level_start_index, const float *sampling_loc,
                       const float *attn_weight, const uint16_t *grad_output, int batch,
                       int spatial_size, int num_heads, int channels, int num_levels,
                       int num_query, int num_point, float *grad_value,
                       float *grad_sampling_loc, float *grad_attn_weight, void *stream);

int msda_backward_bf16_ws(const uint16_t *value, const int64_t *spatial_shapes,
                          const int64_t *level_start_index, const float *sampling_loc,
                          const float *attn_weight, const uint16_t *grad_output, int batch,
                          int spatial_size, int num_heads, int channels, int num_levels,
                          int num_query, int num_point, float *grad_value,
                          float *grad_sampling_loc, float *grad_attn_weight, void *workspace,
                          size_t workspace_bytes, void *stream);

size_t msda_backward_bf16_gv16_workspace_bytes(int batch, int spatial_size, int num_heads, int channels,
                                               int num_levels, int num_query, int num_point);

int msda_backward_bf16_gv16_ws(const uint16_t *value, const int64_t *spatial_shapes,
                               const int64_t *level_start_index, const float *sampling_loc,
                               const float *attn_weight, const uint16_t *grad_output, int batch,
                               int spatial_size, int num_heads, int channels, int num_levels,
                               int num_query, int num_point, uint16_t *grad_value,
                               float *grad_sampling_loc, float *grad_attn_weight, void *workspace,
                               size_t workspace_bytes, void *stream);

/*
 * Fused boundary (opt-in, fp32): the op together with its two
 * elementwise neighbours in the caller, ops/modules/ms_deform_attn.py:102-109 --
 *   attn_weight  = softmax(attn_logits over num_levels * num_point)                  (:103-104)
 *   sampling_loc = reference_points[n,q,None,l,None,:] + sampling_offsets / (W_l, H_l) (:106-109)
 * so the caller passes the raw outputs of its two linears, reference_points [N,Lq,L,2],
 * sampling_offsets [N,Lq,M,L,P,2], attn_logits [N,Lq,M,L*P], and gets gradients with respect to
 * those.  No counterpart in the reference's extension (it takes sampling_loc / attn_weight,
 * src/ms_deform_attn.h:25-32); the drop-in entry points above are unchanged.
 * Forward: channels == 32 and num_levels * num_point <= 16, else MSDA_ERR_INVALID_ARGUMENT (fall
 * back to the unfused entry points).  Backward: any shape msda_backward_f32 takes; the workspace
 * (256-byte aligned, msda_fused_backward_workspace_bytes(...) bytes) holds the recomputed
 * locations and weights, the sort-by-tile lists and (channels == 32) the location / weight
 * gradients in the kernels' own [M, L, N*Lq, P] order, which the epilogue pass turns into the
 * caller's -- 36 bytes per sample plus the lists in all.  No gradient for reference_points is
 * returned (it is sum over heads and points of grad_sampling_offsets * (W_l, H_l)).
 */
int msda_fused_forward_f32(const float *value, const int64_t *spatial_shapes,
                           const int64_t *level_start_index, const float *reference_points,
                           const float *sampling_offsets, const float *attn_logits, int batch,
                           int spatial_size, int num_heads, int channels, int num_levels,
                           int num_query, int num_point, float *output, void *stream);

size_t msda_fused_backward_workspace_bytes(int batch, int spatial_size, int num_heads, int channels,
                                           int num_levels, int num_query, int num_point);

int msda_fused_backward_f32(const float *value, const int64_t *spatial_shapes,
                            const int64_t *level_start_index, const float *reference_points,
                            const float *sampling_offsets, const float *attn_logits,
                            const float *grad_output, int batch, int spatial_size, int num_heads,
                            int channels, int num_levels, int num_query, int num_point,
                            float *grad_value, float *grad_sampling_offsets, float *grad_attn_logits,
                            void *workspace, size_t workspace_bytes, void *stream);

/* bf16 twins of the fused boundary: value / output / grad_output bfloat16 bit patterns, everything
 * else (offsets, logits, reference points, all three gradients) fp32 -- as msda_forward_bf16 /
 * msda_backward_bf16.  Same workspace query. */
int msda_fused_forward_bf16(const uint16_t *value, const int64_t *spatial_shapes,
                            const int64_t *level_start_index, const float *reference_points,
                            const float *sampling_offsets, const float *attn_logits, int batch,
                            int spatial_size, int num_heads, int channels, int num_levels,
                            int num_query, int num_point, uint16_t *output, void *stream);

int msda_fused_backward_bf16(const uint16_t *value, const int64_t *spatial_shapes,
                             const int64_t *level_start_index, const float *reference_points,
                             const float *sampling_offsets, const float *attn_logits,
                             const uint16_t *grad_output, int batch, int spatial_size, int num_heads,
                             int channels, int num_levels, int num_query, int num_point,
                             float *grad_value, float *grad_sampling_offsets, float *grad_attn_logits,
                             void *workspace, size_t workspace_bytes, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* MSDA_HIP_H_ */
