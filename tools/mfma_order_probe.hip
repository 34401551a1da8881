// Development probe: in which order does v_mfma_f32_16x16x4_f32 add its four products to the accumulator?
//   hipcc --offload-arch=gfx950 -O2 tools/mfma_order_probe.hip -o build_variants/mfma_order_probe && build_variants/mfma_order_probe
// D[i][j] = C[i][j] + sum_k A[i][k] B[k][j] with A in lane k*16 + i, B in lane k*16 + j, D rows 4 g + r in lane g*16 + j.
// On data of wide dynamic range every association of the sum rounds differently: the probe evaluates all 24 orders of a
// chain of fmaf (and the "products first, then a tree" forms) on the host and reports which one reproduces the device bits.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void run(const float *A, const float *B, const float *C, float *D, int n) {
  const int lane = threadIdx.x, j = lane & 15, g = lane >> 4;
  for (int t = 0; t < n; ++t) {
    f32x4 acc;
    for (int r = 0; r < 4; ++r) acc[r] = C[(t * 16 + 4 * g + r) * 16 + j];
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[(t * 4 + g) * 16 + j], B[(t * 4 + g) * 16 + j], acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[(t * 16 + 4 * g + r) * 16 + j] = acc[r];
  }
}

int main() {
  const int n = 256;
  std::mt19937 rng(7);
  std::uniform_real_distribution<float> u(-1.f, 1.f);
  std::uniform_int_distribution<int> e(-12, 12);
  float *hA = new float[n * 64], *hB = new float[n * 64], *hC = new float[n * 256], *hD = new float[n * 256];
  for (int i = 0; i < n * 64; ++i) { hA[i] = std::ldexp(u(rng), e(rng)); hB[i] = std::ldexp(u(rng), e(rng)); }
  for (int i = 0; i < n * 256; ++i) hC[i] = std::ldexp(u(rng), e(rng));
  float *dA, *dB, *dC, *dD;
  hipMalloc(&dA, n * 64 * 4); hipMalloc(&dB, n * 64 * 4); hipMalloc(&dC, n * 256 * 4); hipMalloc(&dD, n * 256 * 4);
  hipMemcpy(dA, hA, n * 64 * 4, hipMemcpyHostToDevice); hipMemcpy(dB, hB, n * 64 * 4, hipMemcpyHostToDevice);
  hipMemcpy(dC, hC, n * 256 * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(run, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD, n);
  hipMemcpy(hD, dD, n * 256 * 4, hipMemcpyDeviceToHost);
  int perm[4] = {0, 1, 2, 3};
  int found = 0;
  do {
    long bad = 0;
    for (int t = 0; t < n; ++t)
      for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j) {
          float acc = hC[(t * 16 + i) * 16 + j];
          for (int q = 0; q < 4; ++q) {
            const int k = perm[q];
            acc = std::fmaf(hA[(t * 4 + k) * 16 + i], hB[(t * 4 + k) * 16 + j], acc);
          }
          bad += std::memcmp(&acc, &hD[(t * 16 + i) * 16 + j], 4) != 0;
        }
    printf("fmaf chain onto C in k order %d %d %d %d: %ld of %d elements differ\n", perm[0], perm[1], perm[2], perm[3], bad, n * 256);
    found += bad == 0;
  } while (std::next_permutation(perm, perm + 4));
  {  // exact sum rounded once (what a fused dot-product unit would give)
    long bad = 0;
    for (int t = 0; t < n; ++t)
      for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j) {
          long double s = hC[(t * 16 + i) * 16 + j];
          for (int k = 0; k < 4; ++k) s += (long double)hA[(t * 4 + k) * 16 + i] * (long double)hB[(t * 4 + k) * 16 + j];
          const float f = (float)s;
          bad += std::memcmp(&f, &hD[(t * 16 + i) * 16 + j], 4) != 0;
        }
    printf("exact sum, one rounding: %ld differ\n", bad);
  }
  printf(found ? "ORDER FOUND\n" : "NO fmaf-chain order matches\n");
  return 0;
}
