// Microbenchmark: what does one CU sustain when 4 waves stream tile-packed weights (1 KiB per wave-load, buffer_load_dwordx4)
// the way node_gemm does?  M matrices of 173 KB are cycled (M=1: L2-resident; M=240: 41 MB working set -> first touch per XCD).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int DEPTH>
__global__ __launch_bounds__(256) void stream(const float* __restrict__ w, int tiles_per_mat, int n_mat, int iters, float* out,
                                              unsigned long long* cyc) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  f4 acc = {0, 0, 0, 0};
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    const float* base = w + (size_t)(it % n_mat) * tiles_per_mat * 256;
    f4 q[DEPTH];
    int t = wave;
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) { q[d] = *(const f4*)(base + (size_t)(t < tiles_per_mat ? t : 0) * 256 + lane * 4); t += 4; }
    for (int i = wave; i < tiles_per_mat; i += 4 * DEPTH) {
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) {
        acc += q[d];
        q[d] = *(const f4*)(base + (size_t)(t < tiles_per_mat ? t : 0) * 256 + lane * 4);
        t += 4;
      }
    }
    __syncthreads();
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * 256 + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int DEPTH>
void run(const float* dw, int tiles, int n_mat, int iters, int blocks, float* dout, unsigned long long* dcyc) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  stream<DEPTH><<<blocks, 256>>>(dw, tiles, n_mat, 2, dout, dcyc);
  hipEventRecord(e0);
  stream<DEPTH><<<blocks, 256>>>(dw, tiles, n_mat, iters, dout, dcyc);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> c(blocks);
  hipMemcpy(c.data(), dcyc, 8 * blocks, hipMemcpyDeviceToHost);
  double bytes = (double)tiles * 1024.0 * iters;
  printf("depth %2d n_mat %3d blocks %3d: %.3f ms  %.1f GB/s per CU  (%.1f B per shader-cycle-counter tick of block 0; ticks %llu)\n", DEPTH,
         n_mat, blocks, ms, bytes / (ms * 1e-3) / 1e9, bytes / (double)c[0], c[0]);
}

int main() {
  const int tiles = 169;  // 208x208 floats = 169 KiB
  const int n_mat = 240;
  float* dw; hipMalloc(&dw, (size_t)n_mat * tiles * 1024);
  hipMemset(dw, 0, (size_t)n_mat * tiles * 1024);
  float* dout; hipMalloc(&dout, 256 * 256 * 4);
  unsigned long long* dcyc; hipMalloc(&dcyc, 8 * 256);
  for (int blocks : {1, 32, 256})
    for (int nm : {1, 240}) {
      run<2>(dw, tiles, nm, 2000, blocks, dout, dcyc);
      run<4>(dw, tiles, nm, 2000, blocks, dout, dcyc);
      run<8>(dw, tiles, nm, 2000, blocks, dout, dcyc);
      run<16>(dw, tiles, nm, 2000, blocks, dout, dcyc);
    }
  return 0;
}
