// Probe: can a stream wait (hipStreamWaitValue32) on a flag that a KERNEL of another stream writes from its last
// workgroup, instead of on an event record (a barrier packet that costs the producing stream ~5 us between kernels)?
//   hipcc -O3 --offload-arch=gfx950 tools/probes/waitvalue_probe.hip -o /tmp/waitvalue_probe && /tmp/waitvalue_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <time.h>
#define CK(x) do { hipError_t err_ = (x); if (err_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(err_)); return 1; } } while (0)

__global__ void producer(float* data, int n, unsigned* counter, unsigned* flag, unsigned seq, int spin) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  float v = 0.f;
  for (int k = 0; k < spin; ++k) v = v * 1.0001f + 1.f;
  if (i < n) data[i] = (float)seq + v * 0.f;
  __threadfence();
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned old = atomicAdd(counter, 1u);
    if (old == gridDim.x - 1) {
      *counter = 0;
      __threadfence();
      __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}
__global__ void consumer(const float* data, int n, float expect, int* bad) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && data[i] != expect) atomicAdd(bad, 1);
}
__global__ void filler(float* x, int n, int spin) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  float v = 0.f;
  for (int k = 0; k < spin; ++k) v = v * 1.0001f + 1.f;
  if (i < n) x[i] = v;
}

int main() {
  hipStream_t a, b;
  CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
  const int n = 1 << 22;
  float *data, *scratch;
  unsigned *counter, *flag;
  int* bad;
  CK(hipMalloc(&data, n * sizeof(float)));
  CK(hipMalloc(&scratch, n * sizeof(float)));
  CK(hipMalloc(&counter, 4));
  CK(hipMalloc(&bad, 4));
  CK(hipMemset(counter, 0, 4));
  CK(hipMemset(bad, 0, 4));
  CK(hipExtMallocWithFlags((void**)&flag, 8, hipMallocSignalMemory));
  *reinterpret_cast<volatile unsigned long long*>(flag) = 0;
  CK(hipDeviceSynchronize());
  // correctness: 200 rounds, the consumer on stream b must see round r's data
  for (unsigned r = 1; r <= 200; ++r) {
    hipLaunchKernelGGL(producer, dim3(n / 256), dim3(256), 0, a, data, n, counter, flag, r, 200);
    CK(hipStreamWaitValue32(b, flag, r, hipStreamWaitValueGte, 0xffffffffu));
    hipLaunchKernelGGL(consumer, dim3(n / 256), dim3(256), 0, b, data, n, (float)r, bad);
    // the producer of round r+1 must not overwrite before the consumer has read: order a behind b with an event
    hipEvent_t e;
    CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    CK(hipEventRecord(e, b));
    CK(hipStreamWaitEvent(a, e, 0));
    CK(hipEventDestroy(e));
  }
  CK(hipDeviceSynchronize());
  int hbad = -1;
  CK(hipMemcpy(&hbad, bad, 4, hipMemcpyDeviceToHost));
  printf("wait-value ordering: %d mismatching elements over 200 rounds\n", hbad);
  // cost on the producing stream: 200 back-to-back short kernels with (i) nothing, (ii) an event record between
  // them, (iii) the flag epilogue (the waits sit on the other stream)
  hipEvent_t t0, t1, ev;
  CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
  CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming | hipEventDisableSystemFence));
  for (int mode = 0; mode < 3; ++mode) {
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(t0, a));
      for (unsigned r = 1; r <= 200; ++r) {
        if (mode == 2) {
          hipLaunchKernelGGL(producer, dim3(1024), dim3(256), 0, a, data, 1024 * 256, counter, flag, 1000 * (mode + 1) * (rep + 1) + r, 100);
          CK(hipStreamWaitValue32(b, flag, 1000 * (mode + 1) * (rep + 1) + r, hipStreamWaitValueGte, 0xffffffffu));
          hipLaunchKernelGGL(filler, dim3(64), dim3(256), 0, b, scratch, 64 * 256, 10);
        } else {
          hipLaunchKernelGGL(filler, dim3(1024), dim3(256), 0, a, data, 1024 * 256, 100);
          if (mode == 1) {
            CK(hipEventRecord(ev, a));
            CK(hipStreamWaitEvent(b, ev, 0));
            hipLaunchKernelGGL(filler, dim3(64), dim3(256), 0, b, scratch, 64 * 256, 10);
          }
        }
      }
      CK(hipEventRecord(t1, a));
      CK(hipDeviceSynchronize());
      float ms = 0;
      CK(hipEventElapsedTime(&ms, t0, t1));
      if (rep == 1) printf("mode %d (%s): %.2f us per kernel on the producing stream\n", mode,
                           mode == 0 ? "plain" : mode == 1 ? "event record + wait" : "flag from the last workgroup + wait-value", ms * 1000 / 200);
    }
  }
  // host cost of the calls, and the producing stream's pace when the waits are queued afterwards
  {
    CK(hipDeviceSynchronize());
    auto now = []() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e6 + ts.tv_nsec * 1e-3; };
    CK(hipEventRecord(t0, a));
    double h0 = now();
    for (unsigned r = 1; r <= 200; ++r)
      hipLaunchKernelGGL(producer, dim3(1024), dim3(256), 0, a, data, 1024 * 256, counter, flag, 100000 + r, 100);
    double h1 = now();
    CK(hipEventRecord(t1, a));
    for (unsigned r = 1; r <= 200; ++r) {
      CK(hipStreamWaitValue32(b, flag, 100000 + r, hipStreamWaitValueGte, 0xffffffffu));
      hipLaunchKernelGGL(filler, dim3(64), dim3(256), 0, b, scratch, 64 * 256, 10);
    }
    double h2 = now();
    CK(hipDeviceSynchronize());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, t0, t1));
    printf("producers alone (flag epilogue, waits queued later): %.2f us per kernel; host: %.2f us per launch, %.2f us per wait-value + launch\n",
           ms * 1000 / 200, (h1 - h0) / 200, (h2 - h1) / 200);
    h0 = now();
    for (unsigned r = 1; r <= 200; ++r) {
      CK(hipEventRecord(ev, a));
      CK(hipStreamWaitEvent(b, ev, 0));
    }
    h1 = now();
    CK(hipDeviceSynchronize());
    printf("host: %.2f us per event record + stream wait\n", (h1 - h0) / 200);
  }
  return 0;
}
