// Do vector instructions of one wave issue in the shadow of the matrix instructions of the OTHER wave on the same SIMD?
// 512 threads = 8 waves = two per SIMD (waves w and w + 4).  Waves 0-3 run a loop of NM matrix instructions, waves 4-7 a
// loop of NV vector FMAs; each side is timed alone (the other side idle) and together.  Overlap: together = max; none: sum.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/coissue_microbench.hip -o gaudi_amd/coissue_mb && gaudi_amd/coissue_mb
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));

// KIND 0: v_mfma_f32_16x16x32_bf16, dependent chain on one accumulator; 1: the same on four accumulators; 2: v_mfma_f32_16x16x4_f32
template <int KIND>
__global__ __launch_bounds__(512) void k(float* out, unsigned long long* cyc, int iters, int run_m, int run_v, float a, float b) {
  const int wave = threadIdx.x >> 6;
  const bool mside = wave < 4;
  f4 acc[4] = {{0, 0, 0, 1}, {0, 0, 0, 2}, {0, 0, 0, 3}, {0, 0, 0, 4}};
  float v[8] = {a, b, a + 1, b + 1, a + 2, b + 2, a + 3, b + 3};
  bf8 x, y;
#pragma unroll
  for (int i = 0; i < 8; ++i) { x[i] = (__bf16)a; y[i] = (__bf16)b; }
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if (mside) {
    if (run_m) {
#pragma unroll 1
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          if (KIND == 0) acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, acc[0], 0, 0, 0);
          if (KIND == 1) acc[r & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, acc[r & 3], 0, 0, 0);
          if (KIND == 2) acc[r & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[r & 3], 0, 0, 0);
        }
      }
    }
  } else if (run_v) {
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = __builtin_fmaf(v[i], a, b);  // 32 independent-enough v_fma_f32 per iteration
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][3];
#pragma unroll
  for (int i = 0; i < 8; ++i) s += v[i];
  out[threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[wave] = t1 - t0;
}

// Both waves of a SIMD run the SAME interleaved stream: one matrix instruction, then NV vector FMAs, 16 times per iteration.
template <int NV, bool DEP>
__global__ __launch_bounds__(512) void k2(float* out, unsigned long long* cyc, int iters, int waves_active, float a, float b) {
  const int wave = threadIdx.x >> 6;
  f4 acc[4] = {{0, 0, 0, 1}, {0, 0, 0, 2}, {0, 0, 0, 3}, {0, 0, 0, 4}};
  float v[8] = {a, b, a + 1, b + 1, a + 2, b + 2, a + 3, b + 3};
  bf8 x, y;
#pragma unroll
  for (int i = 0; i < 8; ++i) { x[i] = (__bf16)a; y[i] = (__bf16)b; }
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if (wave < waves_active) {
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        acc[DEP ? 0 : (r & 3)] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, acc[DEP ? 0 : (r & 3)], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < NV; ++i) v[(r * NV + i) & 7] = __builtin_fmaf(v[(r * NV + i) & 7], a, b);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][3];
#pragma unroll
  for (int i = 0; i < 8; ++i) s += v[i];
  out[threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[wave] = t1 - t0;
}
template <int NV, bool DEP>
void run2() {
  float* out;
  unsigned long long* cyc;
  (void)hipMalloc(&out, 512 * 4);
  (void)hipMalloc(&cyc, 8 * 8);
  const int iters = 4000;
  double res[2], res4 = 0;
  for (int mode = 0; mode < 2; ++mode) {  // 4 waves (one per SIMD) | 8 waves (two per SIMD)
    for (int rep = 0; rep < 2; ++rep) {
      hipLaunchKernelGGL((k2<NV, DEP>), dim3(1), dim3(512), 0, 0, out, cyc, iters, mode ? 8 : 4, 1.0f, 0.5f);
      (void)hipDeviceSynchronize();
    }
    unsigned long long h[8];
    (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    res[mode] = (double)h[0] / iters;
    res4 = (double)h[4] / iters;
  }
  printf("interleaved stream, 16 x (1 MFMA 16x16x32 bf16 %s + %d v_fma_f32): one wave per SIMD %.1f cycles per iteration | two waves per SIMD: wave 0 %.1f, wave 4 %.1f\n",
         DEP ? "dependent" : "4 accumulators", NV, res[0], res[1], res4);
  (void)hipFree(out);
  (void)hipFree(cyc);
}

template <int KIND>
void run(const char* name) {
  float* out;
  unsigned long long* cyc;
  (void)hipMalloc(&out, 512 * 4);
  (void)hipMalloc(&cyc, 8 * 8);
  const int iters = 4000;
  double res[3][2];
  for (int mode = 0; mode < 3; ++mode) {  // 0: matrix side alone, 1: vector side alone, 2: both
    const int rm = mode != 1, rv = mode != 0;
    for (int rep = 0; rep < 2; ++rep) {
      hipLaunchKernelGGL((k<KIND>), dim3(1), dim3(512), 0, 0, out, cyc, iters, rm, rv, 1.0f, 0.5f);
      (void)hipDeviceSynchronize();
    }
    unsigned long long h[8];
    (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    res[mode][0] = (double)h[0] / iters;  // wave 0 (matrix side), cycles per iteration of 16 matrix instructions
    res[mode][1] = (double)h[4] / iters;  // wave 4 (vector side, same SIMD), cycles per iteration of 32 v_fma_f32
  }
  printf("%s: 16 matrix instructions alone %.1f cycles | 32 v_fma_f32 alone %.1f | together: matrix wave %.1f, vector wave %.1f  (sum %.1f)\n",
         name, res[0][0], res[1][1], res[2][0], res[2][1], res[0][0] + res[1][1]);
  (void)hipFree(out);
  (void)hipFree(cyc);
}

int main() {
  run<0>("16x16x32 bf16, one dependent chain ");
  run<1>("16x16x32 bf16, four accumulators   ");
  run<2>("16x16x4 f32, four accumulators     ");
  run2<0, false>();
  run2<1, false>();
  run2<2, false>();
  run2<3, false>();
  run2<4, false>();
  run2<2, true>();
  run2<4, true>();
  return 0;
}
