// Development probe: what the SQ counters bench.py's `step_datapath_busy` is made of actually count on gfx950.
//   hipcc --offload-arch=gfx950 -O2 tools/pmc_calib.hip -o tools/build/pmc_calib
//   rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA \
//             SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM -d out -o run --output-format csv -- tools/build/pmc_calib
// Three kernels with a known instruction count per wave, 1,024 workgroups x 256 threads (= one wave per SIMD on every CU,
// one round): calib_valu (N x v_fma_f32), calib_mfma (N x v_mfma_f32_16x16x4_f32), calib_mix (both, interleaved).  The
// program prints the s_memtime cycles a wave spent in its loop; the counters divided by 1,024 SIMDs are compared with them.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define N_VALU 8192
#define N_MFMA 2048

__global__ void calib_valu(long long *cyc, float *sink) {
  float r[8];
  const float x = 1.0f + threadIdx.x * 1e-7f, y = 1.0f - threadIdx.x * 1e-7f;
  for (int i = 0; i < 8; ++i) r[i] = x + i;
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < N_VALU / 8; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(y), "v"(x));
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
  float s = 0;
  for (int i = 0; i < 8; ++i) s += r[i];
  sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void calib_mfma(long long *cyc, float *sink) {
  f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
  const float x = threadIdx.x * 1e-3f, y = 1.0f + threadIdx.x * 1e-4f;
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < N_MFMA / 4; ++i) {
    a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
    a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a1, 0, 0, 0);
    a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a2, 0, 0, 0);
    a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a3, 0, 0, 0);
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
  sink[blockIdx.x * blockDim.x + threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3];
}

// per MFMA two independent v_fma_f32: does the vector work hide behind the matrix instruction or add to it?
__global__ void calib_mix(long long *cyc, float *sink) {
  f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
  float r[8];
  const float x = threadIdx.x * 1e-3f, y = 1.0f + threadIdx.x * 1e-4f;
  for (int i = 0; i < 8; ++i) r[i] = x + i;
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < N_MFMA / 4; ++i) {
    a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[0]) : "v"(y), "v"(x));
    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[1]) : "v"(y), "v"(x));
    a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a1, 0, 0, 0);
    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[2]) : "v"(y), "v"(x));
    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[3]) : "v"(y), "v"(x));
    a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a2, 0, 0, 0);
    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[4]) : "v"(y), "v"(x));
    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[5]) : "v"(y), "v"(x));
    a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a3, 0, 0, 0);
    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[6]) : "v"(y), "v"(x));
    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[7]) : "v"(y), "v"(x));
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
  float s = a0[0] + a1[1] + a2[2] + a3[3];
  for (int i = 0; i < 8; ++i) s += r[i];
  sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
  const int blocks = 1024, threads = 256;
  long long *d_cyc;
  float *d_sink;
  if (hipMalloc((void **)&d_cyc, blocks * 4 * sizeof(long long)) != hipSuccess) return 1;
  if (hipMalloc((void **)&d_sink, blocks * threads * sizeof(float)) != hipSuccess) return 1;
  std::vector<long long> h(blocks * 4);
  struct { const char *name; void (*k)(long long *, float *); int nv, nm; } tests[] = {
      {"calib_valu", calib_valu, N_VALU, 0}, {"calib_mfma", calib_mfma, 0, N_MFMA}, {"calib_mix", calib_mix, 2 * N_MFMA, N_MFMA}};
  for (auto &t : tests) {
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(t.k, dim3(blocks), dim3(threads), 0, 0, d_cyc, d_sink);
    if (hipDeviceSynchronize() != hipSuccess) return 2;
    hipMemcpy(h.data(), d_cyc, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
    double sum = 0;
    for (long long v : h) sum += (double)v;
    // s_memtime tick = shader cycle (MI355X_MICROARCH.md); SQ_ACTIVE_INST_* / SQ_WAVE_CYCLES count quad-cycles
    printf("{\"kernel\": \"%s\", \"v_fma_f32_per_wave\": %d, \"mfma_16x16x4_f32_per_wave\": %d, \"waves\": %d, \"mean_memtime_ticks_per_wave\": %.1f}\n",
           t.name, t.nv, t.nm, blocks * 4, sum / h.size());
  }
  hipFree(d_cyc);
  hipFree(d_sink);
  return 0;
}
