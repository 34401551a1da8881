// Development probe: issue cost of the vector instructions the fp64 front end is made of, on gfx950.
//   hipcc --offload-arch=gfx950 -O2 tools/valu_probe.hip -o tools/build/valu_probe && tools/build/valu_probe
// Each test runs 8 independent register chains x 512 iterations of ONE instruction (inline asm, so the compiler can
// neither fuse nor drop them) in every wave of a workgroup; s_memtime around the loop.  Reported: cycles per
// instruction seen by one wave when it is alone on its SIMD (256 threads = one wave per SIMD) and SIMD cycles per
// instruction when four waves share the SIMD (1024 threads) = the throughput figure.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>

#define REP8(OP)                                                                                                              \
  OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)

#define TEST_D_DD(NAME, INSN)                                                                                                 \
  __global__ void NAME(long long *cyc, double *sink) {                                                                       \
    double r[8];                                                                                                             \
    const double x = 1.0 + threadIdx.x * 1e-9, y = 1.0 - threadIdx.x * 1e-9;                                                 \
    for (int i = 0; i < 8; ++i) r[i] = x + i;                                                                                \
    long long t0 = __builtin_amdgcn_s_memtime();                                                                             \
    for (int it = 0; it < 512; ++it) {                                                                                       \
      _Pragma("unroll") for (int i = 0; i < 8; ++i) asm volatile(INSN " %0, %0, %1" : "+v"(r[i]) : "v"(y));                  \
    }                                                                                                                        \
    long long t1 = __builtin_amdgcn_s_memtime();                                                                             \
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;                                        \
    double s = 0;                                                                                                            \
    for (int i = 0; i < 8; ++i) s += r[i];                                                                                   \
    sink[blockIdx.x * blockDim.x + threadIdx.x] = s;                                                                         \
  }

TEST_D_DD(t_add_f64, "v_add_f64")
TEST_D_DD(t_mul_f64, "v_mul_f64")

__global__ void t_fma_f64(long long *cyc, double *sink) {
  double r[8];
  const double x = 1.0 + threadIdx.x * 1e-9, y = 1.0 - threadIdx.x * 1e-9;
  for (int i = 0; i < 8; ++i) r[i] = x + i;
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < 512; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(r[i]) : "v"(y), "v"(x));
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
  double s = 0;
  for (int i = 0; i < 8; ++i) s += r[i];
  sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

#define TEST_CVT(NAME, INSN, DT, ST)                                                                                          \
  __global__ void NAME(long long *cyc, double *sink) {                                                                       \
    DT r[8];                                                                                                                 \
    ST src[8];                                                                                                               \
    for (int i = 0; i < 8; ++i) src[i] = (ST)(threadIdx.x + i);                                                              \
    for (int i = 0; i < 8; ++i) r[i] = 0;                                                                                    \
    long long t0 = __builtin_amdgcn_s_memtime();                                                                             \
    for (int it = 0; it < 512; ++it) {                                                                                       \
      _Pragma("unroll") for (int i = 0; i < 8; ++i) asm volatile(INSN " %0, %1" : "=v"(r[i]) : "v"(src[i]));                 \
    }                                                                                                                        \
    long long t1 = __builtin_amdgcn_s_memtime();                                                                             \
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;                                        \
    double s = 0;                                                                                                            \
    for (int i = 0; i < 8; ++i) s += (double)r[i];                                                                           \
    sink[blockIdx.x * blockDim.x + threadIdx.x] = s;                                                                         \
  }

TEST_CVT(t_cvt_f64_f32, "v_cvt_f64_f32", double, float)
TEST_CVT(t_cvt_f64_i32, "v_cvt_f64_i32", double, int)
TEST_CVT(t_cvt_f32_f64, "v_cvt_f32_f64", float, double)
TEST_CVT(t_cvt_f32_i32, "v_cvt_f32_i32", float, int)
TEST_CVT(t_sqrt_f32, "v_sqrt_f32", float, float)
TEST_CVT(t_log_f32, "v_log_f32", float, float)
TEST_CVT(t_mov_b32, "v_mov_b32", float, float)

__global__ void t_fma_f32(long long *cyc, double *sink) {
  float r[8];
  const float x = 1.0f + threadIdx.x * 1e-6f, y = 1.0f - threadIdx.x * 1e-6f;
  for (int i = 0; i < 8; ++i) r[i] = x + i;
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < 512; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(y), "v"(x));
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
  double s = 0;
  for (int i = 0; i < 8; ++i) s += r[i];
  sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

typedef float f2 __attribute__((ext_vector_type(2)));
__global__ void t_pk_fma_f32(long long *cyc, double *sink) {
  f2 r[8];
  const f2 x = {1.0f + threadIdx.x * 1e-6f, 0.5f}, y = {1.0f - threadIdx.x * 1e-6f, 0.25f};
  for (int i = 0; i < 8; ++i) r[i] = x + (float)i;
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < 512; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(y), "v"(x));
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
  double s = 0;
  for (int i = 0; i < 8; ++i) s += r[i][0] + r[i][1];
  sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// 64-bit lane exchange as the transform's partner step does it: two ds_bpermute / DPP moves
__global__ void t_dpp_mov64(long long *cyc, double *sink) {
  int r[8];
  for (int i = 0; i < 8; ++i) r[i] = threadIdx.x + i;
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < 512; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(r[i]));
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
  double s = 0;
  for (int i = 0; i < 8; ++i) s += r[i];
  sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

typedef void (*kern)(long long *, double *);
static void run(const char *name, kern k) {
  long long *dc, hc[16];
  double *ds;
  hipMalloc(&dc, 16 * sizeof(long long));
  hipMalloc(&ds, 1024 * sizeof(double));
  double res[2];
  const int threads[2] = {256, 1024};
  for (int c = 0; c < 2; ++c) {
    hipMemset(dc, 0, sizeof(hc));
    hipLaunchKernelGGL(k, dim3(1), dim3(threads[c]), 0, 0, dc, ds);  // warm (instruction cache)
    hipLaunchKernelGGL(k, dim3(1), dim3(threads[c]), 0, 0, dc, ds);
    hipDeviceSynchronize();
    hipMemcpy(hc, dc, sizeof(hc), hipMemcpyDeviceToHost);
    long long mx = 0;
    for (int w = 0; w < threads[c] / 64; ++w) mx = hc[w] > mx ? hc[w] : mx;
    const int waves_per_simd = threads[c] / 256;
    res[c] = (double)mx / (512.0 * 8.0 * waves_per_simd);
  }
  printf("%-16s one wave per SIMD: %5.2f cycles/instruction   four waves per SIMD: %5.2f SIMD cycles/instruction\n", name, res[0], res[1]);
  hipFree(dc);
  hipFree(ds);
}

int main() {
  run("v_add_f64", t_add_f64);
  run("v_mul_f64", t_mul_f64);
  run("v_fma_f64", t_fma_f64);
  run("v_cvt_f64_f32", t_cvt_f64_f32);
  run("v_cvt_f64_i32", t_cvt_f64_i32);
  run("v_cvt_f32_f64", t_cvt_f32_f64);
  run("v_cvt_f32_i32", t_cvt_f32_i32);
  run("v_sqrt_f32", t_sqrt_f32);
  run("v_log_f32", t_log_f32);
  run("v_mov_b32", t_mov_b32);
  run("v_fma_f32", t_fma_f32);
  run("v_pk_fma_f32", t_pk_fma_f32);
  run("v_mov_b32_dpp", t_dpp_mov64);
  return 0;
}
