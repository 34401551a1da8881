// Development probe: lane maps of v_mfma_f32_4x4x1_16b_f32 on gfx950, found with exact integer data.
//   hipcc --offload-arch=gfx950 -O2 tools/mfma_probe.hip -o tools/build/mfma_probe && tools/build/mfma_probe
// For every lane l the A operand is 1000 + l, the B operand is one-hot in lane q (1.0 in lane q, 0 elsewhere); the
// result registers then show which (lane, reg) received A[l'] * 1: D[lane][reg] = 1000 + l' names the A lane whose
// row meets B lane q's column.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void probe(float *out) {
  const int lane = threadIdx.x;
  for (int q = 0; q < 64; ++q) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_4x4x1f32(1000.f + lane, lane == q ? 1.f : 0.f, acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) out[(q * 64 + lane) * 4 + r] = acc[r];
  }
}

// cycles per instruction, one wave, four independent accumulators, 4096 instructions each
__global__ void rate(long long *cyc, float *sink) {
  f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
  const float x = threadIdx.x * 1e-3f, y = 1.0f + threadIdx.x * 1e-4f;
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < 1024; ++i) {
    a0 = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, a0, 0, 0, 0);
    a1 = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, a1, 0, 0, 0);
    a2 = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, a2, 0, 0, 0);
    a3 = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, a3, 0, 0, 0);
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < 1024; ++i) {
    a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
    a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a1, 0, 0, 0);
    a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a2, 0, 0, 0);
    a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a3, 0, 0, 0);
  }
  long long t2 = __builtin_amdgcn_s_memtime();
  // the mix the fused CRNN kernel issues: 4 x (16x16x4) + 4 x (4x4x1) per operand quad
  for (int i = 0; i < 512; ++i) {
    a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
    a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a1, 0, 0, 0);
    a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(y, x, a0, 0, 0, 0);
    a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(y, x, a1, 0, 0, 0);
    a2 = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, a2, 0, 0, 0);
    a3 = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, a3, 0, 0, 0);
    a2 = __builtin_amdgcn_mfma_f32_4x4x1f32(y, x, a2, 0, 0, 0);
    a3 = __builtin_amdgcn_mfma_f32_4x4x1f32(y, x, a3, 0, 0, 0);
  }
  long long t3 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) {
    cyc[0] = t1 - t0;
    cyc[1] = t2 - t1;
    cyc[2] = t3 - t2;
  }
  sink[threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3];
}

int main() {
  {
    long long *dc, hc[3];
    float *ds;
    hipMalloc(&dc, 3 * sizeof(long long));
    hipMalloc(&ds, 64 * sizeof(float));
    hipLaunchKernelGGL(rate, dim3(1), dim3(64), 0, 0, dc, ds);
    hipMemcpy(hc, dc, sizeof(hc), hipMemcpyDeviceToHost);
    printf("cycles per instruction (s_memtime ticks): 4x4x1_16b %.2f, 16x16x4 %.2f, mix of 4+4 %.2f per 8\n", hc[0] / 4096.0,
           hc[1] / 4096.0, hc[2] / 512.0);
  }
  float *d, *h = new float[64 * 64 * 4];
  hipMalloc(&d, 64 * 64 * 4 * sizeof(float));
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h, d, 64 * 64 * 4 * sizeof(float), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int q = 0; q < 64; ++q)
    for (int l = 0; l < 64; ++l)
      for (int r = 0; r < 4; ++r) {
        const float v = h[(q * 64 + l) * 4 + r];
        // expected map: block = lane / 4; A row i = lane % 4; B col j = lane % 4; D[block][i][j] in lane block*4 + j, reg i
        const float want = (l == q) ? 1000.f + (l / 4) * 4 + r : 0.f;
        if (v != want) {
          if (bad < 20) printf("q=%d lane=%d reg=%d got %.0f want %.0f\n", q, l, r, v, want);
          ++bad;
        }
      }
  printf(bad ? "MAP DIFFERS (%d)\n" : "4x4x1_16b map as expected: A[blk=l/4][i=l%%4], B[blk][j=l%%4], D lane blk*4+j reg i\n", bad);
  return bad != 0;
}
