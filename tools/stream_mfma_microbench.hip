// Microbenchmark 2: the node-GEMM weight stream WITH its MFMA work (4 waves, UT=4 tiles per chunk per wave, 13 chunks per
// matrix, 240 matrices cycled = 41 MB working set).  SCHED 0: ping-pong sets as node_gemm (B loaded at the top, A in the
// middle); 1: each of four sets refilled right after its MFMAs; 2: loads only (no MFMA); 3: MFMA only (weights loaded once).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f4 mfma1(float a, float b, f4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f4 mfma_bf(u4 a, u4 b, f4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8, a), __builtin_bit_cast(bf8, b), c, 0, 0, 0);
}
constexpr int T = 13;

#ifndef NWV
#define NWV 4
#endif
template <int SCHED, int NW>
__global__ __launch_bounds__(NWV * 64) void k(const float* __restrict__ w, int n_mat, int iters, float* out) {
  constexpr int UT = (T + NW - 1) / NW;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)w, 0, 0x7fffffff, 0x00020000);
  auto ld = [&](int off_floats) { return __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(r, lane * 16, off_floats * 4, 0)); };
  f4 acc[UT];
  for (int u = 0; u < UT; ++u) acc[u] = (f4){0, 0, 0, 0};
  extern __shared__ float stage[];
  const float xb = 1.0f + lane * 1e-6f;
  f4 side[UT];
  for (int u = 0; u < UT; ++u) side[u] = (f4){0, 0, 0, 0};
  auto mmx = [&](const f4 (&a)[UT]) {
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int u = 0; u < UT; ++u) acc[u] = mfma1(a[u][q], xb, acc[u]);
  };
  for (int it = 0; it < iters; ++it) {
    const int base = (it % n_mat) * (T * T * 256);
    auto chunk = [&](int cc) { return base + (cc < T ? cc : T - 1) * (T * 256); };
    int toff[UT];
    for (int u = 0; u < UT; ++u) { const int t = wave + NW * u; toff[u] = (t < T ? t : wave) * 256; }
    f4 a0[UT], a1[UT], b0[UT], b1[UT], c0[UT], c1[UT];
    // SCHED 10: per-wave staging, two sets of two K chunks x UT tiles (1 KiB each)
    float* stA = stage + wave * (4 * UT * 256);
    float* stB = stA + 2 * UT * 256;
    auto dma = [&](float* dst, int cc) {
#pragma unroll
      for (int u = 0; u < UT; ++u)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(w + chunk(cc) + toff[u] + lane * 4),
                                         (__attribute__((address_space(3))) void*)(dst + u * 256), 16, 0, 0);
    };
    if (SCHED == 10) { dma(stA, 0); dma(stA + UT * 256, 1); }
    if (SCHED != 10) for (int u = 0; u < UT; ++u) { a0[u] = ld(chunk(0) + toff[u]); a1[u] = ld(chunk(1) + toff[u]); }
    if (SCHED == 1 || ((SCHED == 6 || SCHED == 7) && wave >= NW / 2))
      for (int u = 0; u < UT; ++u) { b0[u] = ld(chunk(2) + toff[u]); b1[u] = ld(chunk(3) + toff[u]); }
#pragma unroll 1
    for (int cc = 0; cc < 12; cc += 4) {
      if (SCHED == 0 || SCHED == 2) {
#pragma unroll
        for (int u = 0; u < UT; ++u) { b0[u] = ld(chunk(cc + 2) + toff[u]); b1[u] = ld(chunk(cc + 3) + toff[u]); }
        __builtin_amdgcn_sched_barrier(0);
        if (SCHED == 0) { mmx(a0); mmx(a1); } else { for (int u = 0; u < UT; ++u) acc[u] += a0[u] + a1[u]; }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < UT; ++u) { a0[u] = ld(chunk(cc + 4) + toff[u]); a1[u] = ld(chunk(cc + 5) + toff[u]); }
        __builtin_amdgcn_sched_barrier(0);
        if (SCHED == 0) { mmx(b0); mmx(b1); } else { for (int u = 0; u < UT; ++u) acc[u] += b0[u] + b1[u]; }
        __builtin_amdgcn_sched_barrier(0);
      } else if (SCHED == 1) {
        __builtin_amdgcn_sched_barrier(0); mmx(a0); __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < UT; ++u) a0[u] = ld(chunk(cc + 4) + toff[u]);
        __builtin_amdgcn_sched_barrier(0); mmx(a1); __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < UT; ++u) a1[u] = ld(chunk(cc + 5) + toff[u]);
        __builtin_amdgcn_sched_barrier(0); mmx(b0); __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < UT; ++u) b0[u] = ld(chunk(cc + 6) + toff[u]);
        __builtin_amdgcn_sched_barrier(0); mmx(b1); __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < UT; ++u) b1[u] = ld(chunk(cc + 7) + toff[u]);
      } else if (SCHED == 4) {  // concurrency test: MFMAs on static registers while loads stream into others
#pragma unroll
        for (int u = 0; u < UT; ++u) { b0[u] = ld(chunk(cc + 2) + toff[u]); b1[u] = ld(chunk(cc + 3) + toff[u]); }
        __builtin_amdgcn_sched_barrier(0);
        mmx(a0); mmx(a1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < UT; ++u) { c0[u] = ld(chunk(cc + 4) + toff[u]); c1[u] = ld(chunk(cc + 5) + toff[u]); }
        __builtin_amdgcn_sched_barrier(0);
        mmx(a0); mmx(a1);
        __builtin_amdgcn_sched_barrier(0);
        for (int u = 0; u < UT; ++u) side[u] += b0[u] + b1[u] + c0[u] + c1[u];
      } else if (SCHED == 5) {  // one load after every four matrix instructions (no bursts: the texture path drains between them)
        auto mm1u = [&](const f4 (&a)[UT], int u) {
#pragma unroll
          for (int q = 0; q < 4; ++q) acc[u] = mfma1(a[u][q], xb, acc[u]);
        };
#pragma unroll
        for (int u = 0; u < UT; ++u) { b0[u] = ld(chunk(cc + 2) + toff[u]); __builtin_amdgcn_sched_barrier(0); mm1u(a0, u); __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
        for (int u = 0; u < UT; ++u) { b1[u] = ld(chunk(cc + 3) + toff[u]); __builtin_amdgcn_sched_barrier(0); mm1u(a1, u); __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
        for (int u = 0; u < UT; ++u) { a0[u] = ld(chunk(cc + 4) + toff[u]); __builtin_amdgcn_sched_barrier(0); mm1u(b0, u); __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
        for (int u = 0; u < UT; ++u) { a1[u] = ld(chunk(cc + 5) + toff[u]); __builtin_amdgcn_sched_barrier(0); mm1u(b1, u); __builtin_amdgcn_sched_barrier(0); }
      } else if (SCHED == 6 || SCHED == 7) {  // ping-pong, the second wave of every SIMD half an iteration out of phase
        // (7: + the matrix phase runs at raised priority)
        if (wave >= NW / 2) {
          __builtin_amdgcn_sched_barrier(0);
          if (SCHED == 7) __builtin_amdgcn_s_setprio(2);
          mmx(a0); mmx(a1);
          if (SCHED == 7) __builtin_amdgcn_s_setprio(0);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int u = 0; u < UT; ++u) { a0[u] = ld(chunk(cc + 4) + toff[u]); a1[u] = ld(chunk(cc + 5) + toff[u]); }
          __builtin_amdgcn_sched_barrier(0);
          if (SCHED == 7) __builtin_amdgcn_s_setprio(2);
          mmx(b0); mmx(b1);
          if (SCHED == 7) __builtin_amdgcn_s_setprio(0);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int u = 0; u < UT; ++u) { b0[u] = ld(chunk(cc + 6) + toff[u]); b1[u] = ld(chunk(cc + 7) + toff[u]); }
          __builtin_amdgcn_sched_barrier(0);
        } else {
#pragma unroll
          for (int u = 0; u < UT; ++u) { b0[u] = ld(chunk(cc + 2) + toff[u]); b1[u] = ld(chunk(cc + 3) + toff[u]); }
          __builtin_amdgcn_sched_barrier(0);
          if (SCHED == 7) __builtin_amdgcn_s_setprio(2);
          mmx(a0); mmx(a1);
          if (SCHED == 7) __builtin_amdgcn_s_setprio(0);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int u = 0; u < UT; ++u) { a0[u] = ld(chunk(cc + 4) + toff[u]); a1[u] = ld(chunk(cc + 5) + toff[u]); }
          __builtin_amdgcn_sched_barrier(0);
          if (SCHED == 7) __builtin_amdgcn_s_setprio(2);
          mmx(b0); mmx(b1);
          if (SCHED == 7) __builtin_amdgcn_s_setprio(0);
          __builtin_amdgcn_sched_barrier(0);
        }
      } else if (SCHED == 8) {  // ping-pong with the matrix phase at raised priority
#pragma unroll
        for (int u = 0; u < UT; ++u) { b0[u] = ld(chunk(cc + 2) + toff[u]); b1[u] = ld(chunk(cc + 3) + toff[u]); }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(2); mmx(a0); mmx(a1); __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < UT; ++u) { a0[u] = ld(chunk(cc + 4) + toff[u]); a1[u] = ld(chunk(cc + 5) + toff[u]); }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(2); mmx(b0); mmx(b1); __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
      } else if (SCHED == 9) {  // ping-pong with the LOADS at raised priority (requests leave as early as possible)
        __builtin_amdgcn_s_setprio(2);
#pragma unroll
        for (int u = 0; u < UT; ++u) { b0[u] = ld(chunk(cc + 2) + toff[u]); b1[u] = ld(chunk(cc + 3) + toff[u]); }
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        mmx(a0); mmx(a1);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(2);
#pragma unroll
        for (int u = 0; u < UT; ++u) { a0[u] = ld(chunk(cc + 4) + toff[u]); a1[u] = ld(chunk(cc + 5) + toff[u]); }
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        mmx(b0); mmx(b1);
        __builtin_amdgcn_sched_barrier(0);
      } else if (SCHED == 10) {  // weights by LDS-DMA into a per-wave staging area, ds_read_b128 from there (no VGPR write-back
                                 // from the vector-memory path)
        dma(stB, cc + 2); dma(stB + UT * 256, cc + 3);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * UT) : "memory");
#pragma unroll
        for (int u = 0; u < UT; ++u) { a0[u] = *(const f4*)(stA + u * 256 + lane * 4); a1[u] = *(const f4*)(stA + (UT + u) * 256 + lane * 4); }
        __builtin_amdgcn_sched_barrier(0);
        mmx(a0); mmx(a1);
        __builtin_amdgcn_sched_barrier(0);
        dma(stA, cc + 4); dma(stA + UT * 256, cc + 5);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * UT) : "memory");
#pragma unroll
        for (int u = 0; u < UT; ++u) { b0[u] = *(const f4*)(stB + u * 256 + lane * 4); b1[u] = *(const f4*)(stB + (UT + u) * 256 + lane * 4); }
        __builtin_amdgcn_sched_barrier(0);
        mmx(b0); mmx(b1);
        __builtin_amdgcn_sched_barrier(0);
      } else {  // SCHED 3: MFMA only
        mmx(a0); mmx(a1); mmx(a0); mmx(a1);
      }
    }
    if (SCHED == 10) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      for (int u = 0; u < UT; ++u) a0[u] = *(const f4*)(stA + u * 256 + lane * 4);
    }
    if (SCHED == 0 || SCHED == 1 || SCHED >= 5) mmx(a0); else for (int u = 0; u < UT; ++u) acc[u] += a0[u];
    __syncthreads();
  }
  f4 s = acc[0] + side[0];
  for (int u = 1; u < UT; ++u) s += acc[u] + side[u];
  out[blockIdx.x * 512 + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
}

// Pre-split node-GEMM weights (three bf16 pieces = 1.5x the bytes, six piece products on v_mfma_f32_16x16x32_bf16): per
// (tile, 32-input chunk) three 1 KiB loads and six matrix instructions; 7 chunks per H = 208 matrix (the real one: 6.5);
// activations as static registers.  Ping-pong over chunks like the fp32 form.  LOADS / MFMAS select the parts.
template <int NW, bool LOADS, bool MFMAS>
__global__ __launch_bounds__(NWV * 64) void ks(const float* __restrict__ w, int n_mat, int iters, float* out) {
  constexpr int UT = (T + NW - 1) / NW;
  constexpr int NC = 7;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)w, 0, 0x7fffffff, 0x00020000);
  auto ld = [&](int off_floats) { return __builtin_bit_cast(u4, __builtin_amdgcn_raw_buffer_load_b128(r, lane * 16, off_floats * 4, 0)); };
  f4 acc[UT];
  for (int u = 0; u < UT; ++u) acc[u] = (f4){0, 0, 0, 0};
  const u4 bh = (u4){0x3f803f80u + lane, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}, bm = bh + 1u, bl = bh + 2u;
  u4 side = (u4){0, 0, 0, 0};
  struct Set { u4 h[UT], m[UT], l[UT]; };
  auto mm = [&](const Set& a) {
#pragma unroll
    for (int u = 0; u < UT; ++u) {
      acc[u] = mfma_bf(a.l[u], bh, acc[u]);
      acc[u] = mfma_bf(a.h[u], bl, acc[u]);
      acc[u] = mfma_bf(a.m[u], bm, acc[u]);
      acc[u] = mfma_bf(a.m[u], bh, acc[u]);
      acc[u] = mfma_bf(a.h[u], bm, acc[u]);
      acc[u] = mfma_bf(a.h[u], bh, acc[u]);
    }
  };
  auto use = [&](const Set& a) {
    if (MFMAS) mm(a);
    else
      for (int u = 0; u < UT; ++u) side += a.h[u] + a.m[u] + a.l[u];
  };
  for (int it = 0; it < iters; ++it) {
    const int base = (it % n_mat) * (T * T * 256);  // (same footprint per matrix as the fp32 form x 1.5: NC * 16 * 3 units)
    auto load = [&](Set& a, int cc) {
      const int c = cc < NC ? cc : NC - 1;
#pragma unroll
      for (int u = 0; u < UT; ++u) {
        const int t = wave + NW * u;
        const int off = base / 2 * 3 + ((c * 16 + (t < 16 ? t : 0)) * 3) * 256;
        a.h[u] = ld(off); a.m[u] = ld(off + 256); a.l[u] = ld(off + 512);
      }
    };
    Set A, B;
    if (LOADS) load(A, 0); else { for (int u = 0; u < UT; ++u) { A.h[u] = bh; A.m[u] = bm; A.l[u] = bl; } B = A; }
#pragma unroll 1
    for (int cc = 0; cc < NC - 1; cc += 2) {
      if (LOADS) load(B, cc + 1);
      __builtin_amdgcn_sched_barrier(0);
      use(A);
      __builtin_amdgcn_sched_barrier(0);
      if (LOADS) load(A, cc + 2);
      __builtin_amdgcn_sched_barrier(0);
      use(B);
      __builtin_amdgcn_sched_barrier(0);
    }
    use(A);
    __syncthreads();
  }
  f4 s = acc[0];
  for (int u = 1; u < UT; ++u) s += acc[u];
  out[blockIdx.x * 512 + threadIdx.x] = s[0] + s[1] + s[2] + s[3] + (float)(side[0] ^ side[1] ^ side[2] ^ side[3]);
}
template <int NW, bool LOADS, bool MFMAS>
void run_s(const float* dw, int n_mat, int iters, int blocks, float* dout, const char* name) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  ks<NW, LOADS, MFMAS><<<blocks, NW * 64>>>(dw, n_mat, 8, dout);
  (void)hipEventRecord(e0);
  ks<NW, LOADS, MFMAS><<<blocks, NW * 64>>>(dw, n_mat, iters, dout);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  { hipError_t e = hipGetLastError(); if (e != hipSuccess) printf("  launch error: %s\n", hipGetErrorString(e)); }
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  printf("%-34s n_mat %3d blocks %3d: %.2f us per matrix (7 chunks x 16 tiles x 3 KiB = 336 KB split image)\n", name, n_mat, blocks,
         ms * 1e3 / iters);
}

template <int SCHED, int NW>
void run(const float* dw, int n_mat, int iters, int blocks, float* dout, const char* name) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  k<SCHED, NW><<<blocks, NW * 64, 65536>>>(dw, n_mat, 8, dout);
  (void)hipEventRecord(e0);
  k<SCHED, NW><<<blocks, NW * 64, 65536>>>(dw, n_mat, iters, dout);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  { hipError_t e = hipGetLastError(); if (e != hipSuccess) printf("  launch error: %s\n", hipGetErrorString(e)); }
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double bytes = 169.0 * 1024 * iters;
  printf("%-34s n_mat %3d blocks %3d: %.2f us per 169 KB matrix, %.1f GB/s per CU\n", name, n_mat, blocks, ms * 1e3 / iters,
         bytes / (ms * 1e-3) / 1e9);
}

int main() {
  const int n_mat = 240;
  float* dw; (void)hipMalloc(&dw, (size_t)n_mat * 169 * 1024);
  (void)hipMemset(dw, 0, (size_t)n_mat * 169 * 1024);
  float* dout; (void)hipMalloc(&dout, 256 * 512 * 4);
  for (int nm : {1, 120, 240}) {
    run<2, NWV>(dw, nm, 2000, 256, dout, "loads only");
    run<3, NWV>(dw, nm, 2000, 256, dout, "MFMA only");
    run<0, NWV>(dw, nm, 2000, 256, dout, "loads + MFMA, ping-pong");
    run<1, NWV>(dw, nm, 2000, 256, dout, "loads + MFMA, refill-after");
    run<4, NWV>(dw, nm, 2000, 256, dout, "MFMA on static regs + loads elsewhere");
    run<5, NWV>(dw, nm, 2000, 256, dout, "one load per four MFMAs");
    run<6, NWV>(dw, nm, 2000, 256, dout, "ping-pong, second wave out of phase");
    run<7, NWV>(dw, nm, 2000, 256, dout, "out of phase + MFMA phase at prio 2");
    run<8, NWV>(dw, nm, 2000, 256, dout, "ping-pong, MFMA phase at prio 2");
    run<9, NWV>(dw, nm, 2000, 256, dout, "ping-pong, loads at prio 2");
    run<10, NWV>(dw, nm, 2000, 256, dout, "weights by LDS-DMA + ds_read");
    if (nm <= 160) {  // (the split image is 1.5x: 160 matrices fill the allocation)
      run_s<NWV, true, false>(dw, nm, 2000, 256, dout, "pre-split weights: loads only");
      run_s<NWV, false, true>(dw, nm, 2000, 256, dout, "pre-split weights: MFMA only");
      run_s<NWV, true, true>(dw, nm, 2000, 256, dout, "pre-split weights: loads + MFMA");
    }
  }
  return 0;
}
