// What a CU's LDS-DMA path delivers: W waves of a 512-thread workgroup issue 1 KiB (16 B per lane) or 256 B (4 B per lane) loads
// straight into LDS in a loop, nothing else runs.  Forms: global_load_lds_dwordx4 / _dword (FLAT encoding, M0 = LDS base) and
// buffer_load_dwordx4 ... lds (MUBUF).  Compared with plain buffer_load_dwordx4 into registers (what the node GEMMs use).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ldsdma_rate_microbench.hip -o gaudi_amd/ldsdma_mb
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "w8_nodes_role.h"  // tools/experiments (-I tools/experiments)

using gaudi::f4;

// MODE 0: global_load_lds 16 B; 1: global_load_lds 4 B; 2: raw_buffer_load_lds 16 B; 3: raw_buffer_load_b128 into registers
template <int MODE, int F>
__global__ __launch_bounds__(512) void k(const float* w, unsigned wbytes, int waves, int iters, unsigned long long* cyc, float* out) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)w, 0, wbytes, 0x00020000);
  f4 acc = {0.f, 0.f, 0.f, 0.f};
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if (wave < waves) {
    float* dst = smem + wave * 8 * 256;  // 8 slots of 1 KiB per wave
    const unsigned span = wbytes / 4 - 8 * 256 * 8;
    unsigned off = (blockIdx.x * 7919u * 256u + wave * 256u * 97u) % span;
#pragma unroll 1
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const float* src = w + off + u * 256 * waves;
        if (MODE == 0)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + lane * 4), (__attribute__((address_space(3))) void*)(dst + u * 256), 16, 0, 0);
        if (MODE == 1)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + lane), (__attribute__((address_space(3))) void*)(dst + u * 256), 4, 0, 0);
        if (MODE == 2)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(dst + u * 256), 16, lane * 16, (off + u * 256 * waves) * 4, 0, 0);
        if (MODE == 3) {
          const f4 v = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, (off + u * 256 * waves) * 4, 0));
          asm volatile("" ::"v"(v));
        }
        // what a loader wave would do to tell a consumer: 4 = an LDS store behind the counted wait; 5 = an LDS read (poll) behind it; 6 = no wait,
        // no LDS instruction: the flag word itself travels by LDS-DMA (4 bytes from a table of counters, lane 0 only)
        if (MODE == 4 || MODE == 5 || MODE == 6)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + lane * 4), (__attribute__((address_space(3))) void*)(dst + u * 256), 16, 0, 0);
        if (MODE != 6) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(F) : "memory");
        if (MODE == 4 && lane == 0) asm volatile("ds_write_b32 %0, %1" ::"v"(8 * 8 * 1024 + wave * 16), "v"(i * 8 + u) : "memory");
        if (MODE == 5) {
          unsigned v;
          asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(8 * 8 * 1024 + wave * 16) : "memory");
          acc[0] += (float)v;
        }
        if (MODE == 6 && lane == 0)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(w + ((i * 8 + u) & 1023)), (__attribute__((address_space(3))) void*)(smem + 8 * 8 * 256 + wave * 4), 4, 0, 0);
      }
      off = (off + 8 * 256 * waves) % span;
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  __syncthreads();
  out[blockIdx.x * 512 + tid] = smem[tid] + acc[0];
  if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

// the loader program of w8_nodes_role.h alone: waves 4-7 stream matrix after matrix (free-running: GAUDI_ROLE_ABLATE=1), waves 0-3 wait at the end
template <int HP, int R, int F, int VAR>
__global__ __launch_bounds__(512) void kl(const float* w, int nmat, int mats, unsigned long long* cyc, float* out) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  using namespace gaudi;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  w8::RoleState rs;
  w8::role_init<R>(rs, smem, wave, tid);
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  constexpr int MS = (HP / 16) * (HP / 16) * 256;
  if (wave >= 4) {
    const int p = wave & 3;
#pragma unroll 1
    for (int m = 0; m < mats; ++m) {
      if (VAR == 0) w8::role_stream<HP, R, F>(rs, w, (m % nmat) * MS, p, lane, true, 0, w8::RoleGeo<HP>::pairs(p));
      if (VAR == 2 || VAR == 3 || VAR == 5) {
        const float* img = w + 2 * (m % nmat) * MS;
        for (int k = 0; k < 2 * w8::RoleGeo<HP>::pairs(p); ++k) {
          if (VAR == 3) {
            w8::role_issue2<R, F, 0>(rs, img + (k * 4 + p) * 512, lane);
          } else {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(img + (k * 4 + p) * 256 + lane * 4),
                                             (__attribute__((address_space(3))) void*)(rs.ring + (k % R) * 256), 16, 0, 0);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(F) : "memory");
            if (VAR == 2) w8::role_poke(rs.full, (uint32_t)k, lane);
            if (VAR == 5 && lane == 0) *rs.full = (uint32_t)k;
          }
        }
      }
      if (VAR == 4) w8::role_stream<HP, R, F>(rs, w, (m % nmat) * MS, p, lane, false, 0, w8::RoleGeo<HP>::pairs(p));
      if (VAR == 1) {  // the same units, straight: no ring bookkeeping, no publish
        const float* img = w + 2 * (m % nmat) * MS;
        for (int k = 0; k < 2 * w8::RoleGeo<HP>::pairs(p); ++k) {
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(img + (k * 4 + p) * 256 + lane * 4),
                                           (__attribute__((address_space(3))) void*)(rs.ring + (k % R) * 256), 16, 0, 0);
          asm volatile("s_waitcnt vmcnt(%0)" ::"n"(F) : "memory");
        }
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  __syncthreads();
  out[blockIdx.x * 512 + tid] = smem[tid];
  if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}
template <int HP, int R, int F, int VAR>
void run_loader(const char* name, int blocks, int nmat) {
  float *w, *out;
  unsigned long long* cyc;
  const size_t bytes = (size_t)nmat * (HP / 16) * (HP / 16) * 1024 * 2;
  hipMalloc(&w, bytes);
  hipMemset(w, 0, bytes);
  hipMalloc(&out, blocks * 512 * 4);
  hipMalloc(&cyc, blocks * 8 * 8);
  const int mats = 200;
  const size_t lds = gaudi::w8::role_ring_floats(R) * 4;
  hipFuncSetAttribute((const void*)kl<HP, R, F, VAR>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((kl<HP, R, F, VAR>), dim3(blocks), dim3(512), lds, 0, w, nmat, mats, cyc, out);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(blocks * 8);
  hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
  double mx = 0;
  for (int wv = 4; wv < 8; ++wv) mx = std::max(mx, (double)h[wv]);
  printf("%-40s HP=%d ring %d flight %d blocks=%3d matrices=%3d: %7.0f cycles per matrix, %6.1f per unit of the busiest loader\n", name, HP, R, F, blocks, nmat, mx / mats,
         mx / mats / (2 * gaudi::w8::RoleGeo<HP>::pairs(0)));
  fflush(stdout);
  hipFree(w);
  hipFree(out);
  hipFree(cyc);
}

template <int MODE, int F>
void run(const char* name, int waves, int blocks, size_t set_bytes) {
  float *w, *out;
  unsigned long long* cyc;
  hipMalloc(&w, set_bytes);
  hipMemset(w, 0, set_bytes);
  hipMalloc(&out, blocks * 512 * 4);
  hipMalloc(&cyc, blocks * 8 * 8);
  const int iters = 200;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float ms = 0;
  const size_t lds = 8 * 8 * 1024 + 256;
  hipFuncSetAttribute((const void*)k<MODE, F>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k<MODE, F>), dim3(blocks), dim3(512), lds, 0, w, (unsigned)set_bytes, waves, iters, cyc, out);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
  }
  std::vector<unsigned long long> h(blocks * 8);
  hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
  double mx = 0;
  for (int wv = 0; wv < waves; ++wv) mx = std::max(mx, (double)h[wv]);
  const double bytes = (double)waves * iters * 8 * (MODE == 1 ? 256 : 1024);
  printf("%-34s waves=%d in-flight<=%2d blocks=%3d set=%5.1f MB: %7.1f cycles per wave-instruction, %5.1f B/clk per CU, %6.1f GB/s per CU (wall)\n", name, waves, F + 1, blocks,
         set_bytes / 1e6, mx / (iters * 8), bytes / mx, bytes / (ms * 1e-3) / 1e9);
  fflush(stdout);
  hipFree(w);
  hipFree(out);
  hipFree(cyc);
}

int main() {
  run_loader<208, 12, 6, 0>("role_stream (w8_nodes_role.h)", 256, 120);
  run_loader<208, 12, 6, 0>("role_stream (w8_nodes_role.h)", 1, 120);
  run_loader<208, 12, 6, 0>("role_stream (w8_nodes_role.h)", 256, 1);
  run_loader<208, 12, 6, 1>("plain loop, same bytes", 256, 120);
  run_loader<208, 12, 6, 1>("plain loop, same bytes", 256, 1);
  run_loader<208, 24, 12, 1>("plain loop, same bytes", 256, 120);
  run_loader<208, 12, 6, 2>("plain loop + role_poke (asm ds_write)", 256, 120);
  run_loader<208, 12, 6, 5>("plain loop + volatile store", 256, 120);
  run_loader<208, 12, 6, 3>("plain loop of role_issue", 256, 120);
  run_loader<208, 12, 6, 4>("role_stream, ktail = false", 256, 120);
  return 0;
  const size_t big = 40u << 20, small = 2u << 20;
  for (int waves : {4}) {
    run<0, 6>("global_load_lds_dwordx4", waves, 256, big);
    run<2, 6>("buffer_load_dwordx4 lds", waves, 256, big);
    run<1, 6>("global_load_lds_dword", waves, 256, big);
    run<3, 6>("buffer_load_dwordx4 (registers)", waves, 256, big);
  }
  for (int waves : {1, 4}) {
    run<0, 6>("dma x4, L2-resident set", waves, 256, small);
    run<4, 6>("... + ds_write_b32 after the wait", waves, 256, small);
    run<5, 6>("... + ds_read_b32 poll after the wait", waves, 256, small);
    run<6, 6>("... + flag by LDS-DMA, no wait", waves, 256, small);
  }
  run<0, 6>("global_load_lds_dwordx4", 4, 1, big);
  run<0, 6>("global_load_lds_dwordx4", 4, 256, small);
  run<0, 3>("global_load_lds_dwordx4", 4, 256, big);
  run<0, 7>("global_load_lds_dwordx4", 8, 256, big);
  run<2, 7>("buffer_load_dwordx4 lds", 8, 256, big);
  run<3, 7>("buffer_load_dwordx4 (registers)", 8, 256, big);
  run<3, 3>("buffer_load_dwordx4 (registers)", 8, 256, big);
  run<3, 6>("buffer_load_dwordx4 (registers)", 8, 1, big);
  return 0;
}
