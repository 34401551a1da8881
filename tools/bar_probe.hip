// Can the HOST write a tick's samples straight into device memory (large BAR), so that the tick kernel finds them at HBM
// latency instead of waiting 4 us for its own reads over the bus?  Each allocation flavour is tried in a forked child (a
// host store into memory the CPU cannot reach is a SIGSEGV / SIGBUS, not an error code).
//   hipcc --offload-arch=gfx950 -O2 -o build_variants/bar_probe tools/bar_probe.hip && build_variants/bar_probe
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <sys/wait.h>
#include <time.h>
#include <unistd.h>
#include <vector>

static inline double now_us() {
  timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec * 1e6 + ts.tv_nsec * 1e-3;
}

__global__ void k_sum(const uint4 *src, int n16, unsigned *dst) {
  unsigned acc = 0;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += gridDim.x * blockDim.x) {
    const uint4 v = src[i];
    acc += v.x + v.y + v.z + v.w;
  }
  atomicAdd(dst, acc);
}

static int child(int flavour) {
  const size_t bytes = 128 * 640;
  void *p = nullptr;
  hipError_t e = hipSuccess;
  const char *name = "";
  if (flavour == 0) { name = "hipExtMallocWithFlags(hipDeviceMallocFinegrained)"; e = hipExtMallocWithFlags(&p, bytes, hipDeviceMallocFinegrained); }
  if (flavour == 1) { name = "hipExtMallocWithFlags(hipDeviceMallocUncached)"; e = hipExtMallocWithFlags(&p, bytes, hipDeviceMallocUncached); }
  if (flavour == 2) { name = "hipMalloc"; e = hipMalloc(&p, bytes); }
  if (flavour == 3) { name = "hipMallocManaged + hipMemAdviseSetPreferredLocation(device)"; e = hipMallocManaged(&p, bytes);
    if (e == hipSuccess) { hipMemAdvise(p, bytes, hipMemAdviseSetPreferredLocation, 0); hipMemPrefetchAsync(p, bytes, 0, 0); hipDeviceSynchronize(); } }
  if (flavour == 4) { name = "hipHostMalloc (page-locked host memory: today's staging block)"; e = hipHostMalloc(&p, bytes); }
  printf("%s: ", name);
  if (e != hipSuccess) { printf("allocation failed: %s\n", hipGetErrorString(e)); return 0; }
  fflush(stdout);
  std::vector<unsigned> src(bytes / 4);
  unsigned want = 0;
  for (size_t i = 0; i < src.size(); ++i) { src[i] = (unsigned)(i * 2654435761u); want += src[i]; }
  unsigned *dst;
  hipMalloc(&dst, 4);
  hipStream_t s;
  hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  std::vector<double> tw, tk;
  unsigned got = 0;
  for (int it = 0; it < 2000; ++it) {
    hipMemsetAsync(dst, 0, 4, s);
    hipStreamSynchronize(s);
    const double t0 = now_us();
    memcpy(p, src.data(), bytes);  // the host's stores into the buffer (SIGSEGV here = not host-accessible)
    const double t1 = now_us();
    hipLaunchKernelGGL(k_sum, dim3(128), dim3(128), 0, s, (const uint4 *)p, (int)(bytes / 16), dst);
    hipStreamSynchronize(s);
    const double t2 = now_us();
    tw.push_back(t1 - t0);
    tk.push_back(t2 - t1);
    if (it == 1999) hipMemcpy(&got, dst, 4, hipMemcpyDeviceToHost);
    src[it % src.size()] += 1;  // (new data every time: the kernel must see THIS iteration's stores)
    want += 1;
  }
  want -= 1;
  std::sort(tw.begin(), tw.end());
  std::sort(tk.begin(), tk.end());
  printf("host memcpy of 82 KB p50 %.2f us, kernel (launch + sync) p50 %.2f us, kernel saw the data: %s\n", tw[1000], tk[1000],
         got == want ? "yes" : "NO");
  return 0;
}

int main() {
  for (int f = 0; f < 5; ++f) {
    fflush(stdout);
    const pid_t pid = fork();  // (before any HIP call in this process: the child initialises the runtime itself)
    if (pid == 0) {
      const int rc = child(f);
      fflush(stdout);
      _exit(rc);
    }
    int st = 0;
    waitpid(pid, &st, 0);
    if (WIFSIGNALED(st)) printf("host store faulted (signal %d): not reachable from the CPU\n", WTERMSIG(st));
    else if (WEXITSTATUS(st)) printf("child exited with %d\n", WEXITSTATUS(st));
  }
  return 0;
}
