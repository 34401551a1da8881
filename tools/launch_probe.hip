// Floors of a host-paced tick on this box (round 5, BASELINE config 5): what one kernel launch, a second launch behind it,
// the runtime's completion wake-up and a flag polled in page-locked memory cost, with kernels that do nothing.
//   hipcc --offload-arch=gfx950 -O2 -o build_variants/launch_probe tools/launch_probe.hip && build_variants/launch_probe
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <time.h>
#include <vector>

static inline double now_us() {
  timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec * 1e6 + ts.tv_nsec * 1e-3;
}

__global__ void k_empty(int *sink) {
  if (sink && threadIdx.x == 1024) *sink = 1;
}

// the last workgroup to finish publishes the tick number in page-locked host memory (system-scope release)
__global__ void k_flag(unsigned *counter, unsigned *host_flag, unsigned seq, unsigned groups, int spin) {
  for (int i = 0; i < spin; ++i) __builtin_amdgcn_s_sleep(8);
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence_system();
    const unsigned done = atomicAdd(counter, 1u) + 1u;
    if (done == groups * seq) __hip_atomic_store(host_flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// a kernel that reads 82 KB from page-locked host memory (a tick's frames) and stores one word per group
__global__ void k_read(const uint4 *host_src, int n16, unsigned *dst) {
  unsigned acc = 0;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += gridDim.x * blockDim.x) {
    const uint4 v = host_src[i];
    acc += v.x ^ v.y ^ v.z ^ v.w;
  }
  if (acc == 0x12345u) dst[blockIdx.x] = acc;
}

struct stats {
  std::vector<double> v;
  void add(double x) { v.push_back(x); }
  void print(const char *name) {
    std::sort(v.begin(), v.end());
    double m = 0;
    for (double x : v) m += x;
    printf("%-64s p50 %7.2f  p99 %7.2f  mean %7.2f us\n", name, v[v.size() / 2], v[v.size() * 99 / 100], m / v.size());
  }
};

// one dependent read over the bus per workgroup (what a tick's control words cost a kernel that waits for them)
__global__ void k_read1(const unsigned *host_src, unsigned *dst) {
  const unsigned v = host_src[blockIdx.x & 15];
  if (v == 0x12345u) dst[blockIdx.x] = v;
}

int main() {
  hipStream_t s;
  hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  unsigned *counter, *flag, *flag_dev, *dst;
  hipMalloc(&counter, 4);
  hipMalloc(&dst, 4096);
  hipMemset(counter, 0, 4);
  hipHostMalloc(&flag, 64);
  *flag = 0;
  hipHostGetDevicePointer((void **)&flag_dev, flag, 0);
  uint4 *hsrc, *hsrc_dev;
  const int n16 = 128 * 640 / 16;
  hipHostMalloc(&hsrc, n16 * 16);
  for (int i = 0; i < n16 * 4; ++i) ((unsigned *)hsrc)[i] = i * 2654435761u;
  hipHostGetDevicePointer((void **)&hsrc_dev, hsrc, 0);
  const int N = 5000;
  for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, s, (int *)nullptr);
  hipStreamSynchronize(s);
  {
    stats a, b;
    for (int i = 0; i < N; ++i) {
      const double t0 = now_us();
      hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, s, (int *)nullptr);
      const double t1 = now_us();
      hipStreamSynchronize(s);
      const double t2 = now_us();
      a.add(t1 - t0);
      b.add(t2 - t0);
    }
    a.print("one empty kernel: the launch call");
    b.print("one empty kernel: launch + hipStreamSynchronize");
  }
  {
    stats a, b;
    for (int i = 0; i < N; ++i) {
      const double t0 = now_us();
      hipLaunchKernelGGL(k_empty, dim3(128), dim3(128), 0, s, (int *)nullptr);
      hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, s, (int *)nullptr);
      const double t1 = now_us();
      hipStreamSynchronize(s);
      const double t2 = now_us();
      a.add(t1 - t0);
      b.add(t2 - t0);
    }
    a.print("two empty kernels: the launch calls");
    b.print("two empty kernels: launches + hipStreamSynchronize");
  }
  unsigned seq = 0;
  for (int spin : {0, 64}) {
    stats a, b;
    for (int i = 0; i < N; ++i) {
      ++seq;
      const double t0 = now_us();
      hipLaunchKernelGGL(k_flag, dim3(256), dim3(256), 0, s, counter, flag_dev, seq, 256u, spin);
      const double t1 = now_us();
      while (__atomic_load_n((volatile unsigned *)flag, __ATOMIC_ACQUIRE) != seq) {
      }
      const double t2 = now_us();
      a.add(t1 - t0);
      b.add(t2 - t0);
    }
    char name[96];
    snprintf(name, sizeof name, "flag kernel (s_sleep x %d): the launch call", spin);
    a.print(name);
    snprintf(name, sizeof name, "flag kernel (s_sleep x %d): launch + poll of the pinned word", spin);
    b.print(name);
    hipStreamSynchronize(s);
    stats c;
    for (int i = 0; i < N; ++i) {
      ++seq;
      const double t0 = now_us();
      hipLaunchKernelGGL(k_flag, dim3(256), dim3(256), 0, s, counter, flag_dev, seq, 256u, spin);
      hipStreamSynchronize(s);
      c.add(now_us() - t0);
    }
    snprintf(name, sizeof name, "flag kernel (s_sleep x %d): launch + hipStreamSynchronize", spin);
    c.print(name);
  }
  {
    stats b;
    for (int i = 0; i < N; ++i) {
      const double t0 = now_us();
      hipLaunchKernelGGL(k_read, dim3(128), dim3(128), 0, s, (const uint4 *)hsrc_dev, n16, dst);
      hipStreamSynchronize(s);
      b.add(now_us() - t0);
    }
    b.print("kernel reading 82 KB of pinned host memory: launch + sync");
  }
  {
    stats a, b;
    for (int i = 0; i < N; ++i) {
      const double t0 = now_us();
      hipLaunchKernelGGL(k_read1, dim3(256), dim3(256), 0, s, (const unsigned *)hsrc_dev, dst);
      hipStreamSynchronize(s);
      b.add(now_us() - t0);
      const double t1 = now_us();
      hipLaunchKernelGGL(k_read1, dim3(256), dim3(256), 0, s, (const unsigned *)counter, dst);
      hipStreamSynchronize(s);
      a.add(now_us() - t1);
    }
    b.print("kernel waiting for ONE word of pinned host memory: launch + sync");
    a.print("the same kernel reading the word from device memory: launch + sync");
  }
  {
    // two launches, the second polled by flag
    stats b;
    for (int i = 0; i < N; ++i) {
      ++seq;
      const double t0 = now_us();
      hipLaunchKernelGGL(k_read, dim3(128), dim3(128), 0, s, (const uint4 *)hsrc_dev, n16, dst);
      hipLaunchKernelGGL(k_flag, dim3(256), dim3(256), 0, s, counter, flag_dev, seq, 256u, 0);
      while (__atomic_load_n((volatile unsigned *)flag, __ATOMIC_ACQUIRE) != seq) {
      }
      b.add(now_us() - t0);
    }
    b.print("read kernel + flag kernel: launches + poll");
    hipStreamSynchronize(s);
  }
  return 0;
}
