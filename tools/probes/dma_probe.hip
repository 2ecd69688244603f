// Operand-path bandwidth on gfx950 by level of the hierarchy: every wave reads 1-KiB blocks (64 lanes x 16 B, the
// granule of the bf16 conv kernels' LDS-DMA) at pseudo-random block indices of a region - 3 MB (fits each XCD's 4 MB L2,
// misses the 32 KB L1), 96 MB (Infinity Cache), 3 GB (HBM) - through global_load_lds_dwordx4 or through plain
// global_load_dwordx4 into registers.
//   hipcc -O3 --offload-arch=gfx950 tools/probes/dma_probe.hip -o /tmp/dma_probe && /tmp/dma_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned mix(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}

// the conv kernels' real gather for Cin >= 64: one instruction = 16 rows x 64 B at a 512-byte row stride (half of 16
// 128-byte lines); PAIR issues the other halves of the same lines right behind it
template <bool PAIR>
__global__ __launch_bounds__(256) void probe_rows(const unsigned char* __restrict__ src, unsigned nblocks, int iters, float* sink) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned wid = blockIdx.x * 4 + wave;
  const size_t lane_off = (size_t)(lane >> 2) * 512 + (lane & 3) * 16;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < (PAIR ? 4 : 8); ++k) {
      const size_t base = (size_t)(mix(wid * 7919u + (unsigned)(it * 8 + k)) % nblocks) * 8192 + lane_off;   // 16 rows of 512 B
      __builtin_amdgcn_global_load_lds((gptr_t)(src + base), (lptr_t)(lds + (wave * 8 + k) * 1024), 16, 0, 0);
      if (PAIR)
        __builtin_amdgcn_global_load_lds((gptr_t)(src + base + 64), (lptr_t)(lds + (wave * 8 + 4 + k) * 1024), 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  f32x4 acc = *reinterpret_cast<const f32x4*>(lds + threadIdx.x * 16);
  if (acc[0] == 123.456f) sink[0] = acc[1];
}

template <int MODE>
__global__ __launch_bounds__(256) void probe(const unsigned char* __restrict__ src, unsigned nblocks, int iters, float* sink) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned wid = blockIdx.x * 4 + wave;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
    size_t off[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) off[k] = (size_t)(mix(wid * 7919u + (unsigned)(it * 8 + k)) % nblocks) * 1024 + lane * 16;
    if (MODE == 0) {
#pragma unroll
      for (int k = 0; k < 8; ++k)
        __builtin_amdgcn_global_load_lds((gptr_t)(src + off[k]), (lptr_t)(lds + (wave * 8 + k) * 1024), 16, 0, 0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      f32x4 v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = *reinterpret_cast<const f32x4*>(src + off[k]);
#pragma unroll
      for (int k = 0; k < 8; ++k) acc += v[k];
    }
  }
  if (MODE == 0) acc = *reinterpret_cast<const f32x4*>(lds + threadIdx.x * 16);
  if (acc[0] == 123.456f) sink[0] = acc[1];
}

int main() {
  const size_t total = (size_t)3 << 30;
  unsigned char* buf;
  float* sink;
  if (hipMalloc((void**)&buf, total) != hipSuccess || hipMalloc((void**)&sink, 64) != hipSuccess) return 1;
  (void)hipMemset(buf, 1, total);
  hipEvent_t a, b;
  (void)hipEventCreate(&a);
  (void)hipEventCreate(&b);
  const int iters = 300;
  const size_t regions[3] = {(size_t)3 << 20, (size_t)96 << 20, total};
  for (size_t region : regions)
    for (int wgs_per_cu : {1, 2, 4, 8})
      for (int mode = 0; mode < 2; ++mode) {
        const int grid = 256 * wgs_per_cu;
        const unsigned nblocks = (unsigned)(region / 1024);
        auto launch = [&]() {
          if (mode == 0) hipLaunchKernelGGL(probe<0>, dim3(grid), dim3(256), 32768, 0, buf, nblocks, iters, sink);
          else hipLaunchKernelGGL(probe<1>, dim3(grid), dim3(256), 32768, 0, buf, nblocks, iters, sink);
        };
        launch();
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(a, 0);
        launch();
        (void)hipEventRecord(b, 0);
        (void)hipEventSynchronize(b);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, a, b);
        const double bytes = (double)grid * 4 * iters * 8 * 1024;
        printf("region %5zu MB  %d WG/CU (8 KiB in flight per wave)  %-8s %7.2f TB/s\n", region >> 20, wgs_per_cu,
               mode == 0 ? "lds-dma" : "regs", bytes / (ms * 1e-3) / 1e12);
      }
  // half-line rows (useful bytes counted: 1 KiB per instruction either way)
  for (size_t region : regions)
    for (int wgs_per_cu : {2, 4})
      for (int pair = 0; pair < 2; ++pair) {
        const int grid = 256 * wgs_per_cu;
        const unsigned nblocks = (unsigned)(region / 8192);
        auto launch = [&]() {
          if (pair) hipLaunchKernelGGL(probe_rows<true>, dim3(grid), dim3(256), 32768, 0, buf, nblocks, iters, sink);
          else hipLaunchKernelGGL(probe_rows<false>, dim3(grid), dim3(256), 32768, 0, buf, nblocks, iters, sink);
        };
        launch();
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(a, 0);
        launch();
        (void)hipEventRecord(b, 0);
        (void)hipEventSynchronize(b);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, a, b);
        const double bytes = (double)grid * 4 * iters * 8 * 1024;
        printf("region %5zu MB  %d WG/CU  16 rows x 64 B @ 512 B stride, %s  %7.2f TB/s useful\n", region >> 20, wgs_per_cu,
               pair ? "both halves back to back" : "one half per instruction ", bytes / (ms * 1e-3) / 1e12);
      }
  return 0;
}
