// Micro-benchmark: LDS float atomic add (ds_add_f32) throughput on gfx950 for the access
// shapes the tiled MSDA backward could use.  Build: hipcc -O3 --offload-arch=gfx950 -o mb this.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

constexpr int TILE_ROWS = 256;     // pixels in an accumulator tile
constexpr int ITERS = 2048;

// mode 0: 32 lanes per row, lane = channel (conflict-free by construction)
// mode 1: 8 lanes per row, 4 channels per lane, 4 ds_add per row, rows random, no rotation
// mode 2: same with per-row rotation by (row & 3) dwords
// mode 3: as mode 0 but plain ds_write (store) for comparison
// mode 4: integer ds_add_u32 (no return), one counter per row, 64 random rows per wave-instr
// mode 5: integer ds_add_rtn_u32 (returning), same shape
template <int MODE>
__global__ __launch_bounds__(512) void k(const int *__restrict__ rows, float *out, int nrows_mask) {
  __shared__ float acc[TILE_ROWS * 32];
  for (int i = threadIdx.x; i < TILE_ROWS * 32; i += blockDim.x) acc[i] = 0.f;
  __syncthreads();
  const int t = threadIdx.x;
  float v = 1.0f + t * 1e-6f;
  for (int it = 0; it < ITERS; ++it) {
    if (MODE == 4 || MODE == 5) {
      const int r = rows[((it * 512 + t) + blockIdx.x * 7) & nrows_mask];
      int *cnt = reinterpret_cast<int *>(acc);
      if (MODE == 4) atomicAdd(&cnt[r], 1);
      else v += (float)atomicAdd(&cnt[r], 1);
    } else if (MODE == 0 || MODE == 3) {
      const int r = rows[((it * 16 + (t >> 5)) + blockIdx.x * 7) & nrows_mask];
      const int a = r * 32 + (t & 31);
      if (MODE == 0) atomicAdd(&acc[a], v); else acc[a] = v;
    } else {
      const int r = rows[((it * 64 + (t >> 3)) + blockIdx.x * 7) & nrows_mask];
      const int rot = (MODE == 2) ? (r & 3) : 0;
      const int sub = t & 7;
#pragma unroll
      for (int j = 0; j < 4; ++j) atomicAdd(&acc[r * 32 + ((sub * 4 + j + rot) & 31)], v);
    }
  }
  __syncthreads();
  float s = 0.f;
  for (int i = threadIdx.x; i < TILE_ROWS * 32; i += blockDim.x) s += acc[i];
  if (s == 12345.678f) out[0] = s + v;
}

template <int MODE>
void run(const char *name, const int *drows, float *dout, int mask, double dwords_per_thread_iter) {
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
  const int blocks = 256 * 2;
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(512), 0, 0, drows, dout, mask);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(a));
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(512), 0, 0, drows, dout, mask);
  CHECK(hipEventRecord(b));
  CHECK(hipEventSynchronize(b));
  float ms; CHECK(hipEventElapsedTime(&ms, a, b));
  const double wave_instr = (double)blocks * 8 * ITERS * dwords_per_thread_iter;   // 8 waves per block
  const double per_cu = wave_instr / 256.0;
  printf("%-44s %8.3f ms  %.1f ns per wave-instr per CU  (~%.1f cycles at 2.3 GHz)\n", name, ms,
         ms * 1e6 / per_cu, ms * 1e6 / per_cu * 2.3);
}

int main() {
  const int NR = 1 << 16;
  int *h = (int *)malloc(NR * sizeof(int));
  srand(1);
  for (int i = 0; i < NR; ++i) h[i] = rand() % TILE_ROWS;
  int *d; float *o;
  CHECK(hipMalloc(&d, NR * sizeof(int))); CHECK(hipMalloc(&o, 4));
  CHECK(hipMemcpy(d, h, NR * sizeof(int), hipMemcpyHostToDevice));
  run<0>("ds_add_f32, 32 lanes/row (conflict-free)", d, o, NR - 1, 1);
  run<3>("ds_write_b32, 32 lanes/row", d, o, NR - 1, 1);
  run<1>("ds_add_f32, 8 lanes/row x4, no rotation", d, o, NR - 1, 4);
  run<2>("ds_add_f32, 8 lanes/row x4, rot = row&3", d, o, NR - 1, 4);
  run<4>("ds_add_u32 no-return, random counters", d, o, NR - 1, 1);
  run<5>("ds_add_rtn_u32, random counters", d, o, NR - 1, 1);
  return 0;
}
