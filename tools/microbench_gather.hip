// Micro-benchmark: how fast can a CU gather 128-B rows through the vector L1 (TA/TCP) on gfx950,
// for the lane mappings the MSDA forward could use and for L1-resident vs L2-resident rows?
// Build: hipcc -O3 --offload-arch=gfx950 -o tools/bin/mb_gather tools/microbench_gather.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

constexpr int NLOAD = 16;      // independent row loads in flight per lane group

template <int LPR> struct Vec;
template <> struct Vec<8> { typedef float4 type; };
template <> struct Vec<16> { typedef float2 type; };
template <> struct Vec<32> { typedef float type; };
__device__ __forceinline__ float sum(float4 v) { return v.x + v.y + v.z + v.w; }
__device__ __forceinline__ float sum(float2 v) { return v.x + v.y; }
__device__ __forceinline__ float sum(float v) { return v; }

// LPR lanes own one row per load (128 / LPR bytes per lane).
// PATTERN 0: every lane group draws its own random row of the footprint.
// PATTERN 1: the groups of a wave take consecutive rows starting at a random one (one contiguous
//            64/LPR * 128-B span per wave-instruction).
// PATTERN 2: pairs of neighbouring groups share one random row (half the distinct lines).
template <int LPR, int PATTERN>
__global__ __launch_bounds__(256) void gather(const char *__restrict__ base, uint32_t row_mask,
                                              int iters, float *out) {
  typedef typename Vec<LPR>::type V;
  const uint32_t lane = threadIdx.x & 63u, grp = lane / LPR, sub = lane % LPR;
  const uint32_t wave = (blockIdx.x * 256u + threadIdx.x) >> 6;
  uint32_t h = PATTERN == 0 ? (wave * (64u / LPR) + grp) * 2654435761u + 12345u
             : PATTERN == 1 ? wave * 2654435761u + 12345u
                            : (wave * (64u / LPR) + (grp >> 1)) * 2654435761u + 12345u;
  float acc = 0.f;
  for (int it = 0; it < iters; ++it) {
    V v[NLOAD];
#pragma unroll
    for (int j = 0; j < NLOAD; ++j) {
      h = h * 1664525u + 1013904223u;
      uint32_t row = (h >> 9);
      if (PATTERN == 1) row += grp;
      row &= row_mask;
      v[j] = *reinterpret_cast<const V *>(base + (size_t)row * 128u + sub * (128u / LPR));
    }
#pragma unroll
    for (int j = 0; j < NLOAD; ++j) acc += sum(v[j]);
  }
  if (acc == 12345.678f) out[0] = acc;
}

template <int LPR, int PATTERN>
void run(const char *name, const char *d, float *o, uint32_t rows, double ghz, int cus) {
  const int blocks = cus * 8, iters = 256;   // 8 blocks of 4 waves per CU (all resident at 4 waves/SIMD... 2 rounds)
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
  hipLaunchKernelGGL((gather<LPR, PATTERN>), dim3(blocks), dim3(256), 0, 0, d, rows - 1, iters, o);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(a));
  hipLaunchKernelGGL((gather<LPR, PATTERN>), dim3(blocks), dim3(256), 0, 0, d, rows - 1, iters, o);
  CHECK(hipEventRecord(b));
  CHECK(hipEventSynchronize(b));
  float ms; CHECK(hipEventElapsedTime(&ms, a, b));
  const double row_loads = (double)blocks * 4 * (64 / LPR) * iters * NLOAD;     // row touches
  const double per_cu = row_loads / cus;
  printf("%-58s rows=%7u (%6.0f KiB) %7.3f ms  %5.2f clk/row/CU  %6.2f TB/s\n", name, rows, rows * 128.0 / 1024, ms,
         ms * 1e-3 * ghz * 1e9 / per_cu, row_loads * 128 / (ms * 1e-3) / 1e12);
}

template <int NL, int WPE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void gather_depth(
    const char *__restrict__ base, uint32_t row_mask, int iters, float *out) {
  const uint32_t lane = threadIdx.x & 63u, grp = lane >> 3, sub = lane & 7u;
  const uint32_t wave = (blockIdx.x * 256u + threadIdx.x) >> 6;
  uint32_t h = (wave * 8u + grp) * 2654435761u + 12345u;
  float acc = 0.f;
  for (int it = 0; it < iters; ++it) {
    float4 v[NL];
#pragma unroll
    for (int j = 0; j < NL; ++j) {
      h = h * 1664525u + 1013904223u;
      const uint32_t row = (h >> 9) & row_mask;
      v[j] = *reinterpret_cast<const float4 *>(base + (size_t)row * 128u + sub * 16u);
    }
#pragma unroll
    for (int j = 0; j < NL; ++j) acc += sum(v[j]);
  }
  if (acc == 12345.678f) out[0] = acc;
}

template <int NL, int WPE>
void depth(const char *d, float *o, uint32_t rows, double ghz, int cus) {
  const int blocks = cus * WPE * 4, iters = 4096 / NL;   // WPE waves per SIMD resident, several rounds
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
  hipLaunchKernelGGL((gather_depth<NL, WPE>), dim3(blocks), dim3(256), 0, 0, d, rows - 1, iters, o);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(a));
  hipLaunchKernelGGL((gather_depth<NL, WPE>), dim3(blocks), dim3(256), 0, 0, d, rows - 1, iters, o);
  CHECK(hipEventRecord(b));
  CHECK(hipEventSynchronize(b));
  float ms; CHECK(hipEventElapsedTime(&ms, a, b));
  const double row_loads = (double)blocks * 4 * 8 * iters * NL;
  printf("depth: %2d loads/wave x %d waves/SIMD (%4d wave-loads per CU) rows=%6u  %5.2f clk/row/CU  %6.2f TB/s\n", NL, WPE,
         NL * WPE * 4, rows, ms * 1e-3 * ghz * 1e9 / (row_loads / cus), row_loads * 128 / (ms * 1e-3) / 1e12);
}

// How does the cost of a gather wave-instruction depend on the number of ACTIVE lane groups?
// ACT of the 8 groups of a wave (8 lanes x dwordx4 each) issue the load, the others are masked off.
template <int ACT>
__global__ __launch_bounds__(256) void gather_masked(const char *__restrict__ base, uint32_t row_mask,
                                                     int iters, float *out) {
  const uint32_t lane = threadIdx.x & 63u, grp = lane >> 3, sub = lane & 7u;
  const uint32_t wave = (blockIdx.x * 256u + threadIdx.x) >> 6;
  uint32_t h = (wave * 8u + grp) * 2654435761u + 12345u;
  float acc = 0.f;
  if (grp < (uint32_t)ACT) {
    for (int it = 0; it < iters; ++it) {
      float4 v[NLOAD];
#pragma unroll
      for (int j = 0; j < NLOAD; ++j) {
        h = h * 1664525u + 1013904223u;
        const uint32_t row = (h >> 9) & row_mask;
        v[j] = *reinterpret_cast<const float4 *>(base + (size_t)row * 128u + sub * 16u);
      }
#pragma unroll
      for (int j = 0; j < NLOAD; ++j) acc += sum(v[j]);
    }
  }
  if (acc == 12345.678f) out[0] = acc;
}

template <int ACT>
void masked(const char *d, float *o, uint32_t rows, double ghz, int cus) {
  const int blocks = cus * 8, iters = 256;
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
  hipLaunchKernelGGL((gather_masked<ACT>), dim3(blocks), dim3(256), 0, 0, d, rows - 1, iters, o);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(a));
  hipLaunchKernelGGL((gather_masked<ACT>), dim3(blocks), dim3(256), 0, 0, d, rows - 1, iters, o);
  CHECK(hipEventRecord(b));
  CHECK(hipEventSynchronize(b));
  float ms; CHECK(hipEventElapsedTime(&ms, a, b));
  const double instr = (double)blocks * 4 * iters * NLOAD;
  printf("masked: %d of 8 lane groups active  rows=%6u  %6.2f clk per wave-instruction per CU  (%5.2f clk per active row)\n", ACT, rows,
         ms * 1e-3 * ghz * 1e9 / (instr / cus), ms * 1e-3 * ghz * 1e9 / (instr / cus) / ACT);
}

int main() {
  hipDeviceProp_t p;
  CHECK(hipGetDeviceProperties(&p, 0));
  const double ghz = p.clockRate / 1e6;
  const int cus = p.multiProcessorCount;
  printf("%s: %d CUs, %.2f GHz\n", p.name, cus, ghz);
  const size_t bytes = (size_t)1 << 30;
  char *d; float *o;
  CHECK(hipMalloc(&d, bytes)); CHECK(hipMalloc(&o, 4));
  CHECK(hipMemset(d, 0, bytes));
  const uint32_t sizes[] = {128, 16384};   // 8 KiB, 16 KiB, 128 KiB, 2 MiB, 16 MiB, 512 MiB
  for (uint32_t rows : sizes) {
    run<8, 0>("8 lanes x dwordx4, random row per group", d, o, rows, ghz, cus);
    run<16, 0>("16 lanes x dwordx2, random row per group", d, o, rows, ghz, cus);
    run<32, 0>("32 lanes x dword, random row per group", d, o, rows, ghz, cus);
    run<8, 1>("8 lanes x dwordx4, 8 consecutive rows per wave", d, o, rows, ghz, cus);
    run<8, 2>("8 lanes x dwordx4, neighbouring groups share a row", d, o, rows, ghz, cus);
  }
  for (uint32_t rows : {128u, 16384u}) {
    masked<8>(d, o, rows, ghz, cus); masked<6>(d, o, rows, ghz, cus); masked<4>(d, o, rows, ghz, cus);
    masked<2>(d, o, rows, ghz, cus); masked<1>(d, o, rows, ghz, cus);
  }
  // in-flight depth: waves per SIMD x loads per wave, 8 lanes x dwordx4, random rows
  for (uint32_t rows : {128u, 16384u}) {
    depth<8, 1>(d, o, rows, ghz, cus); depth<16, 1>(d, o, rows, ghz, cus); depth<32, 1>(d, o, rows, ghz, cus);
    depth<8, 2>(d, o, rows, ghz, cus); depth<16, 2>(d, o, rows, ghz, cus); depth<32, 2>(d, o, rows, ghz, cus);
    depth<8, 4>(d, o, rows, ghz, cus); depth<16, 4>(d, o, rows, ghz, cus); depth<32, 4>(d, o, rows, ghz, cus);
    depth<4, 8>(d, o, rows, ghz, cus); depth<8, 8>(d, o, rows, ghz, cus); depth<16, 8>(d, o, rows, ghz, cus);
  }
  return 0;
}
