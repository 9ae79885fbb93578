// Round-trip time of the reverse pass's stash reads (w8_pred.h: 27 KB of node rows per layer and workgroup, read back long after
// the forward pass wrote them) as a function of the stash layout:
//   A  molecule-major (the product's layout up to round 5): address = b * per_molecule + l * per_layer
//   B  layer-major: address = (l * workgroups + b) * per_layer
// Every workgroup (256 of 512 threads, one per CU) first writes its stash (forward order), then spins on ALU work, then reads it back
// layer by layer in reverse order: three float4 per thread and round, all in flight, timed with s_memtime around the wait.
//   hipcc --offload-arch=gfx950 -O3 -o stash_lat_mb tools/stash_latency_microbench.hip && ./stash_lat_mb
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));
#define GL __attribute__((address_space(1)))

constexpr int kThreads = 512, kLayers = 12;
constexpr size_t kNodeF4 = 3 * 572;  // float4 of P, Q, npre for 11 nodes at HP = 208

template <bool NT>
__device__ __forceinline__ f4 ld(const f4* p) {
  if (NT) return __builtin_nontemporal_load((const GL f4*)p);
  return *(const GL f4*)p;
}
template <bool NT>
__device__ __forceinline__ void st4(f4* p, f4 v) {
  if (NT) __builtin_nontemporal_store(v, (GL f4*)p);
  else *(GL f4*)p = v;
}
template <bool NT>
__global__ __launch_bounds__(kThreads) void stash_kernel(float* stash, size_t mol_stride_f4, size_t layer_stride_f4, size_t wg_stride_f4,
                                                         int spin, unsigned long long* out, float* sink) {
  const int tid = threadIdx.x, b = blockIdx.x;
  f4* base = (f4*)stash + (size_t)b * (mol_stride_f4 + wg_stride_f4);
  // forward: write
  for (int l = 0; l < kLayers; ++l) {
    f4* st = base + (size_t)l * layer_stride_f4;
    for (size_t i = tid; i < kNodeF4; i += kThreads) st4<NT>(st + i, (f4){(float)l, (float)i, 1.f, 2.f});
    // a stretch of the edge stash behind it (186 KB per layer in the product), so that the node rows do not sit alone in their pages
    for (size_t i = tid; i < 11648; i += kThreads) st4<NT>(st + kNodeF4 + i, (f4){3.f, 4.f, 5.f, 6.f});
  }
  float acc = 0.f;
  for (int i = 0; i < spin; ++i) acc = __builtin_fmaf(acc, 1.0001f, 0.5f);
  __syncthreads();
  unsigned long long t_sum = 0, t_max = 0;
  for (int l = kLayers - 1; l >= 0; --l) {
    const f4* st = base + (size_t)l * layer_stride_f4;
    const size_t i0 = tid, i1 = tid + kThreads < 572 ? tid + kThreads : 571;
    const unsigned long long t0 = __builtin_readcyclecounter();
    const f4 a0 = ld<NT>(st + i0), a1 = ld<NT>(st + 572 + i0), a2 = ld<NT>(st + 1144 + i0);
    const f4 b0 = ld<NT>(st + i1), b1 = ld<NT>(st + 572 + i1), b2 = ld<NT>(st + 1144 + i1);
    acc += a0[0] + a1[1] + a2[2] + b0[3] + b1[0] + b2[1];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_readcyclecounter();
    t_sum += t1 - t0;
    t_max = t1 - t0 > t_max ? t1 - t0 : t_max;
    for (int i = 0; i < spin / 16; ++i) acc = __builtin_fmaf(acc, 1.0001f, 0.5f);  // (a layer's other work)
    __syncthreads();
  }
  if (tid == 0) {
    out[2 * b] = t_sum;
    out[2 * b + 1] = t_max;
  }
  if (acc == 12345.678f) sink[0] = acc;
}

int main() {
  const int B = 256;
  const size_t per_layer_f4 = kNodeF4 + 11648 + 64;         // ~ 213 KB
  const size_t per_mol_f4 = per_layer_f4 * kLayers + 4096;  // ~ 2.5 MB
  const size_t total = (size_t)B * per_mol_f4 * sizeof(f4);
  float* stash;
  unsigned long long* out;
  float* sink;
  if (hipMalloc(&stash, total) != hipSuccess) return 1;
  hipMalloc(&out, sizeof(unsigned long long) * 2 * B);
  hipMalloc(&sink, 4);
  std::vector<unsigned long long> h(2 * B);
  printf("stash: %.1f MB for %d workgroups, %.1f KB per layer\n", total / 1048576.0, B, per_layer_f4 * 16 / 1024.0);
  for (int nt = 0; nt < 2; ++nt)
  for (int spin : {0, 20000, 200000}) {
    for (int layout = 0; layout < 2; ++layout) {
      // A: molecule-major ; B: layer-major (a workgroup's layers are B * per_layer apart)
      const size_t mol = layout == 0 ? per_mol_f4 : 0, lay = layout == 0 ? per_layer_f4 : per_layer_f4 * B, wg = layout == 0 ? 0 : per_layer_f4;
      for (int rep = 0; rep < 3; ++rep) {
        if (nt) hipLaunchKernelGGL(stash_kernel<true>, dim3(B), dim3(kThreads), 0, 0, stash, mol, lay, wg, spin, out, sink);
        else hipLaunchKernelGGL(stash_kernel<false>, dim3(B), dim3(kThreads), 0, 0, stash, mol, lay, wg, spin, out, sink);
        if (hipDeviceSynchronize() != hipSuccess) return 2;
      }
      hipMemcpy(h.data(), out, sizeof(unsigned long long) * 2 * B, hipMemcpyDeviceToHost);
      double s = 0, m = 0;
      for (int b = 0; b < B; ++b) {
        s += (double)h[2 * b] / kLayers;
        m = h[2 * b + 1] > m ? h[2 * b + 1] : m;
      }
      printf("nt %d spin %6d  layout %s: mean read round trip %.0f cycles per layer (worst single %.0f)\n", nt, spin, layout == 0 ? "A molecule-major" : "B layer-major   ",
             s / B, m);
    }
  }
  return 0;
}
