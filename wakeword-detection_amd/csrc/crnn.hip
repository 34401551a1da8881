// CRNN encode + detect for gfx950 (fp32, MFMA for the dense contractions).
//
// Replaces encode.tflite + detect.tflite of the reference CRNN
// (tf_lite_models/CRNN; architecture wwdetect/CRNN/model.py:21-56; call sites
// spokestack/wakeword/tflite.py:193-231, utils/evaluate_models.py:76-80).
//
//   crnn_fused_kernel  the whole model for one window in one 4-wave workgroup: Conv2D(1->32, 5x20, stride 2x8, SAME,
//                      ReLU) as an implicit GEMM, the layer-1 input projections of both GRU directions (K = 640,
//                      N = 192), both bidirectional GRU layers and the detect head - nothing between the mel window
//                      and the posterior leaves the CU (header of the kernel below).
//   generic path       any other conv geometry (utils/CRNN_files/*_old.tflite): conv_generic_kernel, gemm_nt_kernel
//                      (C = A W^T + b, 64x64x64 tiles, fp32 MFMA, XCD-aware tile order), gru_generic_kernel,
//                      crnn_detect_kernel.
// Round 1's three-kernel chain (conv5x20_kernel -> gemm_nt_kernel -> gru_head_kernel, 52 us per 256 windows against
// 34 us fused) and its bf16x6 projection GEMM are gone; their measurements are in DESIGN.md 7.1.
#include "common.h"
#include "fft_device.h"
#include <type_traits>

#include <cstdlib>

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct win_addr {
  const int64_t *row;    // explicit first mel row per window, or nullptr
  const int32_t *valid;  // explicit valid rows per window, or nullptr
  int64_t row0;
  int hop;
  int valid_const;
  int64_t mel_rows;
};

__device__ __forceinline__ void window_span(const win_addr &wa, int w, int T, int64_t &row, int &valid) {
  if (wa.row && wa.valid) {  // (both tables: the two loads go out together, one round trip at the head of the kernel)
    const int64_t r = wa.row[w];
    const int v = wa.valid[w];
    row = r;
    valid = v;
  } else {
    row = wa.row ? wa.row[w] : wa.row0 + (int64_t)w * wa.hop;
    valid = wa.valid ? wa.valid[w] : wa.valid_const;
  }
  if (valid > T) valid = T;
  if (row + valid > wa.mel_rows) valid = (int)(wa.mel_rows - row);
  if (valid < 0) valid = 0;
}

__device__ __forceinline__ float sigmoid_f(float x) { return 1.0f / (1.0f + expf(-x)); }

// Four mel features at p (zeros unless `in`).  Whether the window's first row is 16-byte aligned is uniform over a workgroup,
// so the staging loops branch on it ONCE, outside (stage_loads below): with the 16-byte and the 4 x 4-byte form behind one
// per-element condition the compiler drained the memory counter after every element's loads - six round trips to L2 in a
// row at the head of crnn_fused_kernel (6.4 k of its 72 k cycles) instead of six loads in flight.
template <bool AL16>
__device__ __forceinline__ float4 ld_mel4(const float *p, bool in) {
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (in) {
    if (AL16) v = *(const float4 *)p;
    else v = make_float4(p[0], p[1], p[2], p[3]);
  }
  return v;
}
// The same from an address that is readable whatever `in` says (the caller clamps it into the buffer): the load is
// unconditional - no exec masking, nothing the next load has to wait for - and `in` only selects between it and zeros.
template <bool AL16>
__device__ __forceinline__ float4 ld_mel4_sel(const float *p_safe, bool in) {
  const float4 v = ld_mel4<AL16>(p_safe, true);
  return make_float4(in ? v.x : 0.f, in ? v.y : 0.f, in ? v.z : 0.f, in ? v.w : 0.f);
}
template <typename F>
__device__ __forceinline__ void stage_loads(bool al16, F &&body) {
  if (al16) body(std::true_type{});
  else body(std::false_type{});
}

// ------------------------------------------------------------------------------------------
// conv
// ------------------------------------------------------------------------------------------
#define CV_ROWS 44   // mel index + PF, 0..43
#define CV_LDT 164   // time index + PT, 0..163 (multiple of 4; 164 % 32 == 4)
#define CV_KB 7      // K = 5*20 = 100 padded to 112 = 7 * 16
#define CV_THREADS 512
// Geometry of every CRNN the reference ships (wwdetect/CRNN/train.py:27-49; checked against the
// blob at model load): compile-time constants turn the index divisions into shifts/multiplies.
#define CV_NMEL 40
#define CV_KF 5
#define CV_KT 20
#define CV_SF 2
#define CV_ST 8
#define CV_PF 1
#define CV_PT 6
#define CV_OF 20
#define CV_OT 19

// ------------------------------------------------------------------------------------------
// GEMM  C[M][N] = A[M][K] * W[N][K]^T + bias[N]
// ------------------------------------------------------------------------------------------
#define GB_M 64
#define GB_N 64
#ifndef GB_K
#define GB_K 64
#endif
#define GB_LD (GB_K + 8)  // 288 B rows at K = 64 (32 mod 64): the operand ds_read_b128s are bank-conflict-free (68 was 2-way conflicted)

struct gemm_args {
  const float *A;
  const float *W;
  const float *bias;
  float *C;
  int M, N, K;
  int n_tiles;  // ceil(N / 64)
};

// 512 threads = 8 waves (2 per SIMD, so one wave's LDS/MFMA latencies are covered by its partner);
// wave (wr, wc) = (wave >> 1, wave & 1) owns rows wr*16..+15, columns wc*32..+31 of the 64x64 tile.
__global__ __launch_bounds__(512) void gemm_nt_kernel(gemm_args g) {
  __shared__ __align__(16) float As[2][GB_M * GB_LD];
  __shared__ __align__(16) float Bs[2][GB_N * GB_LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  // XCD-aware tile order: workgroups go round-robin to the 8 XCDs (id % 8), each with its own L2.  The n-tiles
  // of one m-tile read the same 64 rows of A, so they are given ids of equal residue: A then crosses the fabric
  // once instead of once per n-tile (FETCH_SIZE 28.9 -> see profiles/r01/pmc_traffic.json).
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int mt = (slot / g.n_tiles) * 8 + xcd, nt = slot % g.n_tiles;
  if (mt * GB_M >= g.M) return;
  const int m0 = mt * GB_M, n0 = nt * GB_N;
  constexpr int C4 = GB_K / 4, LROWS = 512 / C4, LH = GB_M / LROWS;  // loader: LH rows per thread, float4 column lc4
  const int lrow = tid / C4, lc4 = tid % C4;
  const int i16 = lane & 15, kk = lane >> 4;

  float4 pa[LH], pb[LH];
  auto gload = [&](int k0) {
#pragma unroll
    for (int h = 0; h < LH; ++h) {
      const int r = lrow + LROWS * h;
      const int gm = m0 + r;
      pa[h] = gm < g.M ? *(const float4 *)(g.A + (size_t)gm * g.K + k0 + lc4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
      const int gn = n0 + r;
      pb[h] = gn < g.N ? *(const float4 *)(g.W + (size_t)gn * g.K + k0 + lc4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto sstore = [&](int buf) {
#pragma unroll
    for (int h = 0; h < LH; ++h) {
      const int r = lrow + LROWS * h;
      *(float4 *)(&As[buf][r * GB_LD + lc4 * 4]) = pa[h];
      *(float4 *)(&Bs[buf][r * GB_LD + lc4 * 4]) = pb[h];
    }
  };

  f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  const int nk = g.K / GB_K;
  gload(0);
  sstore(0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) gload((kt + 1) * GB_K);
#pragma unroll
    for (int kb = 0; kb < GB_K / 16; ++kb) {
      const float4 av = *(const float4 *)(&As[cur][(wr * 16 + i16) * GB_LD + kb * 16 + kk * 4]);
      const float4 b0 = *(const float4 *)(&Bs[cur][(wc * 32 + i16) * GB_LD + kb * 16 + kk * 4]);
      const float4 b1 = *(const float4 *)(&Bs[cur][(wc * 32 + 16 + i16) * GB_LD + kb * 16 + kk * 4]);
      acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, b0.x, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, b1.x, acc[1], 0, 0, 0);
      acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, b0.y, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, b1.y, acc[1], 0, 0, 0);
      acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, b0.z, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, b1.z, acc[1], 0, 0, 0);
      acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, b0.w, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, b1.w, acc[1], 0, 0, 0);
    }
    if (kt + 1 < nk) sstore(cur ^ 1);
    __syncthreads();
  }
#pragma unroll
  for (int ni = 0; ni < 2; ++ni) {
    const int gn = n0 + wc * 32 + ni * 16 + i16;
    const float bv = (gn < g.N) ? g.bias[gn] : 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int gm = m0 + wr * 16 + kk * 4 + r;
      if (gm < g.M && gn < g.N) g.C[(size_t)gm * g.N + gn] = acc[ni][r] + bv;
    }
  }
}

// ------------------------------------------------------------------------------------------
// GRU building blocks (one lane pair per unit: K split over adjacent lanes, joined by one DPP swap)
// ------------------------------------------------------------------------------------------
#define GR_H 32
#define GR_SEQ_LD 68   // seq1 row: 64 values + 4 pad
#define GR_GX_LD 196   // gx2 row: 192 values + 4 pad
#define GR_W1_LD 65

// v_exp_f32 / v_rcp_f32 based gates (~1 ulp each): |error| ~ 2e-7, far inside the 1e-4 budget,
// and several times shorter than the libm expf / tanhf / IEEE-divide sequences that sit on the
// 38-step serial chain.
__device__ __forceinline__ float fast_sigmoid(float x) {
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}
__device__ __forceinline__ float fast_tanh(float x) {
  const float e = __builtin_amdgcn_exp2f(2.8853900817779268f * x);  // exp(2x): inf -> 1, 0 -> -1
  return __fmaf_rn(-2.0f, __builtin_amdgcn_rcpf(1.0f + e), 1.0f);  // spelled out: every kernel rounds it the same way
}
// The GRU cell's last two lines, spelled out (no contraction left to the compiler) so that the vector form (gru_step) and the
// matrix form (gru_tail16_kernel) round identically:  c = tanh(xc + r * hc),  h' = z h + (1 - z) c
__device__ __forceinline__ float gru_candidate(float r, float hc, float xc) { return fast_tanh(__fmaf_rn(r, hc, xc)); }
__device__ __forceinline__ float gru_blend(float z, float h, float c) { return __fmaf_rn(z, h, __fmul_rn(__fsub_rn(1.0f, z), c)); }

__device__ __forceinline__ float swap_pair(float v) {
  // value of the neighbouring lane (lane ^ 1): DPP quad_perm [1,0,3,2]
  return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xF, 0xF, true));
}

typedef float f32x2 __attribute__((ext_vector_type(2)));

struct gru_w {
  f32x2 z[8], r[8], c[8];  // W_h rows of this lane's unit, columns [16*half, 16*half+16), as pairs
  float bz, br, bc;
};

__device__ __forceinline__ void gru_load_w(gru_w &g, const float *wh, const float *bh, int dir, int unit, int half) {
  const float *base = wh + (size_t)dir * 3 * GR_H * GR_H;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float4 z = *(const float4 *)(base + (size_t)(0 * GR_H + unit) * GR_H + half * 16 + q * 4);
    const float4 r = *(const float4 *)(base + (size_t)(1 * GR_H + unit) * GR_H + half * 16 + q * 4);
    const float4 c = *(const float4 *)(base + (size_t)(2 * GR_H + unit) * GR_H + half * 16 + q * 4);
    g.z[2 * q] = (f32x2){z.x, z.y}; g.z[2 * q + 1] = (f32x2){z.z, z.w};
    g.r[2 * q] = (f32x2){r.x, r.y}; g.r[2 * q + 1] = (f32x2){r.z, r.w};
    g.c[2 * q] = (f32x2){c.x, c.y}; g.c[2 * q + 1] = (f32x2){c.z, c.w};
  }
  // the recurrent biases are the initial values of half 0's odd-k chains (gru_step); half 1's chains start from zero
  g.bz = half ? 0.f : bh[dir * 3 * GR_H + unit];
  g.br = half ? 0.f : bh[dir * 3 * GR_H + GR_H + unit];
  g.bc = half ? 0.f : bh[dir * 3 * GR_H + 2 * GR_H + unit];
}

// One GRU step for lane (unit = lane >> 1, half = lane & 1).  hin: this direction's h in LDS.
// Keras GRU v2 (reset_after): z = s(xz + hz), r = s(xr + hr), c = tanh(xc + r * hc), h' = z h + (1-z) c
// ONE association for every kernel that evaluates the cell (this function: crnn_fused_kernel, gru_tail_kernel,
// crnn_stream_kernel, the generic path; gru_tail16_kernel restates it on v_mfma_f32_16x16x4_f32, whose four products are an
// fmaf chain onto the accumulator in k order - tools/mfma_order_probe.hip): a gate's pre-activation is
//     (E0 + O0) + (E1 + O1)
// with four fmaf chains over the units k of the state, eight terms each, k ascending:
//     E0: k = 0, 2, .. 14 starting from the projected input x (z, r; 0 for c)      O0: k = 1, 3, .. 15 starting from b_h
//     E1: k = 16, 18, .. 30 from 0                                                 O1: k = 17, 19, .. 31 from 0
// Here E and O are the two halves of a packed fp32 FMA (v_pk_fma_f32) and the index 0 / 1 is the lane of the pair (half);
// the halves meet through one DPP swap.  The result does not depend on which kernel a window was dispatched to.
template <bool PREFETCH>
__device__ __forceinline__ float gru_step(const gru_w &g, const float *hin, int half, float gz, float gr, float gc, float h_own,
                                          const float *gx_next, float &nz, float &nr, float &nc) {
  const float4 *hp = (const float4 *)(hin + half * 16);
  // all four reads in flight before the first product, in the order of their use (the LDS answers in issue order; left alone
  // the scheduler issues the first-needed quad last)
  float4 hv[4];
  hv[0] = hp[0];
  __builtin_amdgcn_sched_barrier(0);
  hv[1] = hp[1];
  hv[2] = hp[2];
  hv[3] = hp[3];
  __builtin_amdgcn_sched_barrier(0);
  if (PREFETCH) {  // the next step's projected inputs (LDS): behind the h reads
    nz = gx_next[0];
    nr = gx_next[GR_H];
    nc = gx_next[2 * GR_H];
  }
  f32x2 az = {half ? 0.f : gz, g.bz}, ar = {half ? 0.f : gr, g.br}, ac = {0.f, g.bc};
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const f32x2 h0 = {hv[q].x, hv[q].y}, h1 = {hv[q].z, hv[q].w};
    // the three chains strictly round-robin (a dependent packed FMA does not issue back to back; left alone the scheduler
    // runs one chain ahead of the others: +55 cycles per step)
    az = __builtin_elementwise_fma(g.z[2 * q], h0, az); ar = __builtin_elementwise_fma(g.r[2 * q], h0, ar); ac = __builtin_elementwise_fma(g.c[2 * q], h0, ac);
    __builtin_amdgcn_sched_barrier(0);
    az = __builtin_elementwise_fma(g.z[2 * q + 1], h1, az); ar = __builtin_elementwise_fma(g.r[2 * q + 1], h1, ar); ac = __builtin_elementwise_fma(g.c[2 * q + 1], h1, ac);
    __builtin_amdgcn_sched_barrier(0);
  }
  float sz = az.x + az.y, sr = ar.x + ar.y, sc = ac.x + ac.y;
  sz += swap_pair(sz);
  sr += swap_pair(sr);
  sc += swap_pair(sc);
  const float z = fast_sigmoid(sz);
  const float r = fast_sigmoid(sr);
  return gru_blend(z, h_own, gru_candidate(r, sc, gc));
}

// Between the h write of one step and the h reads of the next: LDS instructions of ONE wave are
// executed in issue order, so no s_waitcnt is needed - only the compiler must not reorder them.
__device__ __forceinline__ void wsync_h() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_wave_barrier();
}

// Workgroup barrier between LDS producers and consumers only: waits for this wave's LDS traffic, not for its outstanding
// global loads (__syncthreads() drains the memory counter too - a prefetch issued just before it is then waited for in full)
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

__device__ __forceinline__ void wsync_g() {
  // LDS traffic of one wave is processed in order; wait for it only (not for outstanding
  // global loads, which an acq_rel fence would also drain) and stop compiler reordering
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
}

// ------------------------------------------------------------------------------------------
// crnn_fused_kernel: the whole CRNN for one window in one 4-wave workgroup - the conv output (feat, 48.6 KB per
// window) and the layer-1 input projections (gx1, 14.6 KB) never leave the CU.
//
//   A  stage the 151x40 window transposed into LDS (as conv5x20_kernel)
//   B  conv as implicit GEMM on v_mfma_f32_16x16x4_f32, 6 of the 24 m-tiles per wave -> feat[19][640] in LDS
//   C  layer-1 input projection gx1[19][192] = feat x Wx1^T: the 19 rows are ONE 16-row MFMA tile plus a 3-row
//      remainder on v_mfma_f32_4x4x1_16b_f32 (16 blocks of 4x4x1 at the same 32 MAC/cycle): with the blocks dealt
//      as (k sub-step kk, column quad cg) the B operand is the SAME register as the 16x16x4's
//      (lane = kk*16 + col), the A operand is feat[16 + lane%4][k + 4 kk ..], and the four kk partial sums of a
//      column meet in two cross-lane adds at the very end - 20 rows of matrix time for 19 instead of 32.
//      Each wave owns 3 of the 12 n-tiles over the whole K.  W_x1 (491 KB) streams from L2 straight into registers
//      in B-operand order ([k/4][192][4], one contiguous KB per wave load), three k-steps ahead of its use.
//   D  layer-1 recurrence (wave 0 forward, wave 1 backward); every wave's W_x2 operands (B-operand order) on their way
//      from L2 into registers
//   E  layer-2 input projection, the same 16 + 3 row split, A operand out of LDS
//   F  layer-2 recurrence (waves 2, 3) | waves 0, 1 stage the detect head in the dead feat space
//   G  detect head (wave 0)
//
// One wave per SIMD: v_mfma_f32_16x16x4_f32 reaches its issue rate from a single wave with independent accumulators,
// while two such waves on a SIMD got in each other's way (8-wave form of this kernel: the older wave of a SIMD took
// 14.8k cycles over 9.6k cycles of MFMAs and its partner advanced only once it was alone).
// LDS: the window image (28.9 KB; after B it holds the last six feat rows, then gx, seq1, h, the head's small vectors) + feat
// rows 0..12 (33.7 KB; after C the head's first layer) = 62.6 KB (78.1 KB until the end of round 4: CF_FUSED_SMEM_BYTES below),
// two workgroups per CU - one's recurrences and staging run beside the other's MFMA phases - with room for two front-end
// workgroups beside them.
// ------------------------------------------------------------------------------------------
#define CF_THREADS 256
#define CF_FLD 648  // feat row: 640 + 8; FLD/4 = 162 = 2 (mod 16): the 16-lane groups of the A-operand ds_read_b128 hit 16 distinct slots
#define CF_IMG_FLOATS (CV_ROWS * CV_LDT)
#define CF_FEAT_FLOATS (CV_OT * CF_FLD)
#define CF_SMEM_BYTES ((CF_IMG_FLOATS + CF_FEAT_FLOATS) * 4)
// crnn_fused_kernel keeps only rows 0..12 of feat in the feat region.  Once every wave has read its last image operand the image
// is dead, and the rows the conv produces last go into it as a flat array UNDER gx (which is written last) and clear of seq1 / h
// (which are zeroed first): rows 13..15 as three ordinary feat rows (pitch CF_FLD: the projection's main-tile operand pointer
// advances by the same 16 floats per k-step in every lane, only its base differs), rows 16..18 - the 3-row remainder tile, read
// through a pointer of its own - as a [20][2 k-steps x 3 rows x 16] block behind them (compile-time offsets).
// 78.1 -> 62.6 KB per workgroup: two of them leave two front-end workgroups' 2 x 17 KB free on the CU (timing probes with fewer
// feat rows, same instruction stream: pipelined step 44.7 us at 19 rows, 43.6 at 16, 43.2 at 13, 42.9 at 8: DESIGN.md 7.1).
#define CF_NO_ROW (-0x40000000)                // store offset of a conv row past the window's last one
#define CF_FUSED_FEAT_ROWS 13
#define CF_FUSED_SMEM_BYTES ((CF_IMG_FLOATS + CF_FUSED_FEAT_ROWS * CF_FLD) * 4)
#define CF_ALIAS_ROWS0 24                      // image offset (floats) of feat rows 13..15: (24 - CF_IMG_FLOATS) / 4 = 10 - 2 * 13 (mod 16), i.e. row j
                                               // sits on the 16-byte slot 2 j (mod 16) of the bank cycle like the rows in the feat region
#define CF_ALIAS_REM (CF_ALIAS_ROWS0 + 3 * CF_FLD)  // ... of the remainder block [20][96]
static_assert((((CF_ALIAS_ROWS0 - CF_IMG_FLOATS) / 4 - 2 * 13) % 16 + 16) % 16 == 0 && CF_IMG_FLOATS % 4 == 0, "aliased rows off their bank slots");
static_assert(CV_OT - 16 == 3 && 16 - CF_FUSED_FEAT_ROWS == 3 && CF_ALIAS_REM + 20 * 96 <= 20 * GR_GX_LD /* = CF_SEQ */,
              "the aliased rows must lie under gx");
// offsets (floats) inside the image region once the conv is done
#define CF_GX 0
#define CF_SEQ (20 * GR_GX_LD)
#define CF_HB (CF_SEQ + 32 * GR_SEQ_LD)   // h[layer 2][dir 2][ping-pong 2][32]
#define CF_ENC (CF_HB + 2 * 2 * 2 * GR_H)
#define CF_HID (CF_ENC + 2 * GR_H)
#define CF_W2S (CF_HID + 2 * GR_H)
static_assert(CF_W2S + 8 * 64 <= CF_IMG_FLOATS, "post-conv LDS layout exceeds the image region");
static_assert(64 * GR_W1_LD <= CF_FEAT_FLOATS && 16 * 192 * 4 <= CF_FEAT_FLOATS, "feat region too small for its later tenants");
static_assert(64 * GR_W1_LD <= CF_FUSED_FEAT_ROWS * CF_FLD, "fused kernel's feat region too small for the head's first layer");

struct fused_args {
  const float *mel;
  win_addr wa;
  const float *w4;    // conv weights [112/4][32][4]
  const float *cbias; // [32]
  const float *wx1s;  // W_x1 in B-operand order [640/4][192][4]
  const float *bx1;   // [192]
  const float *wh1, *bh1;
  const float *wx2s;  // W_x2 in B-operand order [64/4][192][4]
  const float *bx2, *wh2, *bh2, *w1, *b1, *w2, *b2;
  float *enc;         // optional [Nw][64]
  float *out;         // [Nw][NOUT]
  int T, NOUT, HEAD;
  long long *stamps;  // development: [blocks][4 waves][10] s_memtime at the phase boundaries (nullptr = off)
  float *gx_out;      // FRONT_ONLY: [Nw][OT][192] layer-1 input projections incl. b_x
  const unsigned short *cwb;   // bf16x3: conv weights, A-operand order [plane 2][k-step 4][m-tile 2][lane 64][8]
  const unsigned short *wx1b;  // bf16x3: W_x1, B-operand order [plane 2][k-step 20][n-tile 12][lane 64][8]
  ww_tick_tag tag;             // streaming ticks: the posterior as a {value, tick number} pair instead of the row of `out` (common.h)
};

// One 8-byte store straight to page-locked host memory (system scope: not parked in L2): the host sees value and tick number
// together or not at all.
__device__ __forceinline__ void tick_tag_store(const ww_tick_tag &t, int w, float v) {
  const unsigned long long word = (unsigned long long)__float_as_uint(v) | ((unsigned long long)t.seq << 32);
  __hip_atomic_store(t.slots + w, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// one direction of one GRU layer over the OT steps held in LDS (gx rows incl. b_x), h ping-pong in hd
template <bool SEQ>
__device__ __forceinline__ float cf_recurrence(const gru_w &g, const float *gxs, float *hd, float *seq1, int dir, int unit, int half) {
  constexpr int H = GR_H, OT = CV_OT;
  const float *gxl = gxs + dir * 3 * H + unit;
  float h_own = 0.f;
  int t = dir ? OT - 1 : 0;
  // a step's projected inputs start two of its chains, so they are fetched a step ahead (behind the h reads, which are the
  // ones the step waits for)
  float gz = gxl[t * GR_GX_LD], gr = gxl[t * GR_GX_LD + H], gc = gxl[t * GR_GX_LD + 2 * H];
  for (int s = 0; s < OT; ++s) {
    const int cur = s & 1;
    const int tn = dir ? (t > 0 ? t - 1 : 0) : (t + 1 < OT ? t + 1 : OT - 1);
    float nz, nr, nc;
    h_own = gru_step<true>(g, hd + cur * H, half, gz, gr, gc, h_own, gxl + tn * GR_GX_LD, nz, nr, nc);
    if (half == 0) {
      hd[(cur ^ 1) * H + unit] = h_own;
      if (SEQ) seq1[t * GR_SEQ_LD + dir * H + unit] = h_own;
    }
    wsync_h();
    gz = nz; gr = nr; gc = nc;
    t = tn;
  }
  return h_own;
}

// one operand quad of the 16 + 3 row product: 3 n-tiles x (16x16x4 on rows 0..15, 4x4x1 on rows 16..18); consecutive
// MFMAs go to different accumulators (dependent-accumulator latency 40 > issue 32)
#define CF_ROUND(av_, rv_, b_, e_)                                                                     \
  acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av_.e_, b_[0].e_, acc[0], 0, 0, 0);                     \
  acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av_.e_, b_[1].e_, acc[1], 0, 0, 0);                     \
  acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(av_.e_, b_[2].e_, acc[2], 0, 0, 0);                     \
  rem[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(rv_.e_, b_[0].e_, rem[0], 0, 0, 0);                       \
  rem[1] = __builtin_amdgcn_mfma_f32_4x4x1f32(rv_.e_, b_[1].e_, rem[1], 0, 0, 0);                       \
  rem[2] = __builtin_amdgcn_mfma_f32_4x4x1f32(rv_.e_, b_[2].e_, rem[2], 0, 0, 0);

// the layer-2 projection's form of it: rows 16..18 on a second 16x16x4 tile (its other thirteen rows read zeros), so that ALL
// nineteen rows are one fmaf chain in the same k order - the association gru_tail16_kernel reproduces step by step
#define CF_ROUND_L2(av_, rv_, b_, e_)                                                                  \
  acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av_.e_, b_[0].e_, acc[0], 0, 0, 0);                     \
  acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av_.e_, b_[1].e_, acc[1], 0, 0, 0);                     \
  acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(av_.e_, b_[2].e_, acc[2], 0, 0, 0);                     \
  rem[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(rv_.e_, b_[0].e_, rem[0], 0, 0, 0);                     \
  rem[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(rv_.e_, b_[1].e_, rem[1], 0, 0, 0);                     \
  rem[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(rv_.e_, b_[2].e_, rem[2], 0, 0, 0);

// development: s_memtime at the phase boundaries (fused_args::stamps; nullptr = off)
#ifdef CF_STAMP_PHASE_A  // development build: slots 6..9 hold the inside of phase A instead of phases D..G
#define CF_STAMP(i_)                                                                                          \
  if ((i_) < 6 && a.stamps && lane == 0) a.stamps[((size_t)blockIdx.x * 4 + wave) * 10 + (i_)] = __builtin_amdgcn_s_memtime();
#define CF_STAMP_A(i_)                                                                                        \
  if (a.stamps && lane == 0) a.stamps[((size_t)blockIdx.x * 4 + wave) * 10 + (i_)] = __builtin_amdgcn_s_memtime();
#else
#define CF_STAMP(i_)                                                                                          \
  if (a.stamps && lane == 0) a.stamps[((size_t)blockIdx.x * 4 + wave) * 10 + (i_)] = __builtin_amdgcn_s_memtime();
#define CF_STAMP_A(i_)
#endif

// Phases D..G of the fused kernels (fp32 and split-bf16 front halves share them): gx1 is in LDS, g holds this wave's
// recurrent weights (waves 0, 1: layer 1; waves 2, 3: layer 2).
__device__ __forceinline__ void cf_phases_d_to_g(const fused_args &a, float *img, float *feat, const gru_w &g, int w) {
  constexpr int H = GR_H;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 15, kk = lane >> 4;
  const int unit = lane >> 1, half = lane & 1, dir = wave & 1;
  float *gxs = img + CF_GX, *seq1 = img + CF_SEQ, *hb = img + CF_HB, *encs = img + CF_ENC, *hid = img + CF_HID, *w2s = img + CF_W2S;
  // ---- D: layer-1 recurrence (waves 0, 1).  Every wave first requests its B operands of the layer-2 projection (W_x2 in
  //      B-operand order, 12 x 16 bytes per lane, L2): they arrive during the recurrence, and phase E reads nothing but
  //      its A operand from LDS (W_x2 used to be staged through the feat space by waves 2, 3; phase E 4.3 k -> 4.0 k cycles for 1.9 k of MFMAs)
  float4 bq2[4][3];
#pragma unroll
  for (int kb = 0; kb < 4; ++kb)
#pragma unroll
    for (int n = 0; n < 3; ++n) bq2[kb][n] = *(const float4 *)(a.wx2s + ((size_t)(kb * 4 + kk) * 192 + (wave * 3 + n) * 16 + j) * 4);
  if (wave < 2) {
    __builtin_amdgcn_s_setprio(3);  // the serial chain issues ahead of a co-resident workgroup's MFMA stream (+1.5 % at scale)
    cf_recurrence<true>(g, gxs, hb + dir * 2 * H, seq1, dir, unit, half);
    __builtin_amdgcn_s_setprio(0);
  }
  __syncthreads();
  CF_STAMP(6)

  // ---- E: layer-2 input projection gx2[t][n] = seq1[t][:] . Wx2[n][:] + bx2[n]  (19 rows, K = 64, N = 192): rows 0..15 as one
  //      MFMA tile, rows 16..18 as a second one (rows 19..31 of seq1 are zeros), 3 n-tiles per wave; every row is the same
  //      fmaf chain from zero: k = 16 kb + 4 kk + e in the order kb, e, kk, the bias added last
  {
    f32x4 acc[3] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    f32x4 rem[3] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    // the biases: requested in front of the products they are added to (asked for where they are used, each of the three
    // was a round trip to L2 of its own at the end of the phase)
    float bb2[3];
#pragma unroll
    for (int n = 0; n < 3; ++n) bb2[n] = a.bx2[(wave * 3 + n) * 16 + j];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
      const float4 av = *(const float4 *)(&seq1[j * GR_SEQ_LD + kb * 16 + kk * 4]);
      const float4 rv = *(const float4 *)(&seq1[(16 + j) * GR_SEQ_LD + kb * 16 + kk * 4]);
      const float4 *b = bq2[kb];
      CF_ROUND_L2(av, rv, b, x) CF_ROUND_L2(av, rv, b, y) CF_ROUND_L2(av, rv, b, z) CF_ROUND_L2(av, rv, b, w)
    }
#pragma unroll
    for (int n = 0; n < 3; ++n) {
      const int col = (wave * 3 + n) * 16 + j;
      const float bb = bb2[n];
#pragma unroll
      for (int r = 0; r < 4; ++r) gxs[(kk * 4 + r) * GR_GX_LD + col] = acc[n][r] + bb;
      if (kk == 0) {
#pragma unroll
        for (int r = 0; r < 3; ++r) gxs[(16 + r) * GR_GX_LD + col] = rem[n][r] + bb;
      }
    }
  }
  __syncthreads();
  CF_STAMP(7)

  // ---- F: layer-2 recurrence (waves 2, 3): only the last state of each direction is kept | waves 0, 1: head -> LDS
  float *w1s = feat;  // [64][GR_W1_LD]
  if (wave >= 2) {
    __builtin_amdgcn_s_setprio(3);  // the serial chain issues ahead of a co-resident workgroup's MFMA stream (+1.5 % at scale)
    const float h_last = cf_recurrence<false>(g, gxs, hb + (2 + dir) * 2 * H, nullptr, dir, unit, half);
    __builtin_amdgcn_s_setprio(0);
    if (half == 0) {
      encs[dir * H + unit] = h_last;
      if (a.enc) a.enc[(size_t)w * 2 * H + dir * H + unit] = h_last;
    }
  } else {
    for (int i = tid; i < 64 * 64; i += 128) w1s[(i >> 6) * GR_W1_LD + (i & 63)] = a.w1[i];
    for (int i = tid; i < a.NOUT * 64; i += 128) w2s[i] = a.w2[i];
  }
  __syncthreads();
  CF_STAMP(8)

  // ---- G: detect head: Dense(64, relu) -> Dense(NOUT) -> sigmoid | softmax   (wave 0)
  if (wave == 0) {
    const float b1v = a.b1[lane], b2v = a.b2[lane < a.NOUT ? lane : 0];  // on their way during the dot products
    __builtin_amdgcn_sched_barrier(0);
    float acc = 0.f;
#pragma unroll 16
    for (int k = 0; k < 2 * H; ++k) acc = fmaf(w1s[lane * GR_W1_LD + k], encs[k], acc);
    hid[lane] = fmaxf(acc + b1v, 0.f);
    wsync_g();
    float y = 0.f;
    if (lane < a.NOUT) {
      for (int k = 0; k < 2 * H; ++k) y = fmaf(w2s[lane * 64 + k], hid[k], y);
      y += b2v;
    }
    float p;
    if (a.HEAD == 0) {
      p = sigmoid_f(y);
    } else {
      float mx = (lane < a.NOUT) ? y : -INFINITY;
      for (int o = 1; o < 8; o <<= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
      float e = (lane < a.NOUT) ? expf(y - mx) : 0.f;
      float sum = e;
      for (int o = 1; o < 8; o <<= 1) sum += __shfl_xor(sum, o);
      p = e / sum;
    }
    if (a.tag.slots) {
      if (lane == a.tag.pidx) tick_tag_store(a.tag, w, p);
    } else if (lane < a.NOUT) {
      a.out[(size_t)w * a.NOUT + lane] = p;
    }
  }
  CF_STAMP(9)
}

template <bool FRONT_ONLY>
__global__ __launch_bounds__(CF_THREADS, 2) void crnn_fused_kernel(fused_args a) {
  extern __shared__ __align__(16) float cf_smem[];
  float *img = cf_smem, *feat = cf_smem + CF_IMG_FLOATS;
  constexpr int H = GR_H, OT = CV_OT;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 15, kk = lane >> 4;
  const int w = blockIdx.x;
  CF_STAMP(0)
  int64_t row;
  int valid;
  window_span(a.wa, w, a.T, row, valid);

  // conv weights for this lane (B operand of mfma 16x16x4: lane (j, kk)); issued first, used after the staging
  float4 wreg[CV_KB][2];
#pragma unroll
  for (int kb = 0; kb < CV_KB; ++kb)
#pragma unroll
    for (int n = 0; n < 2; ++n) wreg[kb][n] = *(const float4 *)(a.w4 + ((size_t)(kb * 4 + kk) * 32 + n * 16 + j) * 4);
  const float cb0 = a.cbias[j], cb1 = a.cbias[16 + j];

  // ---- A: stage the window (loads first, then zero the image, then the transposed scatter)
  int a_off[6], o_off[6][4];
  {
    constexpr int MAXV = 6;  // 6 * 256 float4 >= 151 * 40 / 4
    const float *src = a.mel + row * CV_NMEL;
    const int n = valid * CV_NMEL;
    const bool al16 = ((((uintptr_t)src) & 15) == 0);
    float4 stage[MAXV];
    // n is a multiple of 4: a float4 is inside the window or outside it.  Outside ones are never scattered, so they load the
    // window's last float4 again instead of being masked: six unconditional loads, all in flight at once
    if (n > 0)
      stage_loads(al16, [&](auto al) {
        constexpr bool AL = decltype(al)::value;
#pragma unroll
        for (int q = 0; q < MAXV; ++q) {
          const int i = (q * CF_THREADS + tid) * 4;
          stage[q] = ld_mel4<AL>(src + (i < n ? i : n - 4), true);
        }
      });
    CF_STAMP_A(6)
    for (int i = tid; i < CF_IMG_FLOATS / 4; i += CF_THREADS) ((float4 *)img)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    CF_STAMP_A(7)
    // while the window is on its way: LDS offsets of this lane's conv operands and results for its six m-tiles
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      constexpr int M = CV_OT * CV_OF;
      const int mt = wave + 4 * i;
      int m = mt * 16 + j;
      if (m >= M) m = M - 1;
      const int t = m / CV_OF, f = m - t * CV_OF;
      a_off[i] = (f * CV_SF) * CV_LDT + t * CV_ST;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int mo = mt * 16 + kk * 4 + r;
        const int to = mo / CV_OF, fo = mo - to * CV_OF;
        // offsets from `feat`; rows 13..15 (in tiles 4) and 16..18 (tiles 5) go into the image, below feat: negative offsets.
        // remainder block: k = 32 fo + 16 n + j is k-step 2 fo + n -> row fo of the block, column 48 n + 16 (t - 16) + j
        // (i is a compile-time constant: tiles 0..3 hold rows t <= 12, tile 4 rows 12..15, tile 5 rows 16..18 and, in wave 3, rows past M)
        if (i < 4) o_off[i][r] = to * CF_FLD + fo * 32 + j;
        else if (i == 4) o_off[i][r] = (to < 13 ? to * CF_FLD : CF_ALIAS_ROWS0 - CF_IMG_FLOATS + (to - 13) * CF_FLD) + fo * 32 + j;
        else o_off[i][r] = mo >= M ? CF_NO_ROW : CF_ALIAS_REM - CF_IMG_FLOATS + fo * 96 + 16 * (to - 16) + j;
      }
    }
    __syncthreads();
    CF_STAMP_A(8)
#pragma unroll
    for (int q = 0; q < MAXV; ++q) {
      const int i = (q * CF_THREADS + tid) * 4;
      if (i < n) {  // the four features are four consecutive mel bins of ONE frame (40 = 10 x 4): one division, four rows of the image
        const int it = i / CV_NMEL, im = i - it * CV_NMEL;
        float *d = img + (im + CV_PF) * CV_LDT + it + CV_PT;
        d[0] = stage[q].x;
        d[CV_LDT] = stage[q].y;
        d[2 * CV_LDT] = stage[q].z;
        d[3 * CV_LDT] = stage[q].w;
      }
    }
  }
  CF_STAMP_A(9)
  __syncthreads();
  CF_STAMP(1)
  // (raw buffer loads with a scalar k-step offset would take the 64-bit pointer adds off the vector ALU, but this compiler
  //  lowers __builtin_amdgcn_raw_buffer_load_b128 to a one-dword load on gfx950: not used)
  const float *wb = a.wx1s + ((size_t)kk * 192 + wave * 48 + j) * 4;
  constexpr size_t KS_STRIDE = (size_t)4 * 192 * 4;
  auto w_ld = [&](int ks, int n) { return *(const float4 *)(wb + ks * KS_STRIDE + n * 64); };
  float4 bq[4][3];
  // ---- B: conv -> feat (LDS).  Six m-tiles per wave, software-pipelined by hand: the next tile's A operands are read while
  //      this tile's MFMAs run, and the previous tile's ReLU + store sit in the middle of them.  Vector instructions between
  //      fp32 MFMAs cost matrix time, so the loop holds none it can avoid: operand and store offsets were computed during the
  //      staging (a_off, o_off), the bias is the accumulators' initial value.
  {
    auto load_a = [&](float4(&av)[CV_KB], int i) {
      const float *abase = img + a_off[i];
#pragma unroll
      for (int kb = 0; kb < CV_KB; ++kb) {
        const int k4 = kb * 16 + kk * 4;
        const int kf = k4 / CV_KT, kt = k4 - kf * CV_KT;
        av[kb] = *(const float4 *)(abase + kf * CV_LDT + kt);
      }
    };
    auto store_tile = [&](int i, const f32x4 &r0, const f32x4 &r1) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (i < 5) {
          feat[o_off[i][r]] = relu1(r0[r]);
          feat[o_off[i][r] + 16] = relu1(r1[r]);
        } else if (o_off[i][r] != CF_NO_ROW) {  // (only the last tile of wave 3 has rows past M); the remainder block's channel halves are 48 apart
          feat[o_off[i][r]] = relu1(r0[r]);
          feat[o_off[i][r] + 48] = relu1(r1[r]);
        }
      }
    };
#define CF_CONV_KB(kb_)                                                                              \
  acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i & 1][kb_].x, wreg[kb_][0].x, acc0, 0, 0, 0);       \
  acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i & 1][kb_].x, wreg[kb_][1].x, acc1, 0, 0, 0);       \
  acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i & 1][kb_].y, wreg[kb_][0].y, acc0, 0, 0, 0);       \
  acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i & 1][kb_].y, wreg[kb_][1].y, acc1, 0, 0, 0);       \
  acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i & 1][kb_].z, wreg[kb_][0].z, acc0, 0, 0, 0);       \
  acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i & 1][kb_].z, wreg[kb_][1].z, acc1, 0, 0, 0);       \
  acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i & 1][kb_].w, wreg[kb_][0].w, acc0, 0, 0, 0);       \
  acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i & 1][kb_].w, wreg[kb_][1].w, acc1, 0, 0, 0);
    float4 av[2][CV_KB];
    f32x4 prev0 = {0.f, 0.f, 0.f, 0.f}, prev1 = {0.f, 0.f, 0.f, 0.f}, t4_0 = prev0, t4_1 = prev0;
    load_a(av[0], 0);
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      if (i + 1 < 6) load_a(av[(i + 1) & 1], i + 1);
      __builtin_amdgcn_sched_barrier(0);
      f32x4 acc0 = {cb0, cb0, cb0, cb0}, acc1 = {cb1, cb1, cb1, cb1};
      CF_CONV_KB(0) CF_CONV_KB(1) CF_CONV_KB(2)
      __builtin_amdgcn_sched_barrier(0);
      if (i > 0 && i < 5) store_tile(i - 1, prev0, prev1);  // tiles 4 and 5 hold the rows that go into the image: kept until it is dead
      if (i == 5) { t4_0 = prev0; t4_1 = prev1; }
      __builtin_amdgcn_sched_barrier(0);
      CF_CONV_KB(3) CF_CONV_KB(4) CF_CONV_KB(5) CF_CONV_KB(6)
      prev0 = acc0;
      prev1 = acc1;
    }
    // the projection's first W operands: requested here, in front of the two barriers (they do not wait for global loads)
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
      for (int n = 0; n < 3; ++n) bq[s][n] = w_ld(s, n);
    __builtin_amdgcn_sched_barrier(0);
    lds_barrier();  // every wave has read its last image operand: rows 13..18 of feat may take the image's place
    store_tile(4, t4_0, t4_1);
    store_tile(5, prev0, prev1);
#undef CF_CONV_KB
  }
  CF_STAMP(2)
  lds_barrier();  // feat complete (rows 0..12 in the feat region, rows 13..18 in the image, which holds nothing else any more)
  CF_STAMP(3)
  float *gxs = img + CF_GX, *seq1 = img + CF_SEQ, *hb = img + CF_HB;
  for (int i = tid; i < 32 * GR_SEQ_LD; i += CF_THREADS) seq1[i] = 0.f;  // (behind gx: clear of the aliased rows)
  if (tid < 2 * 2 * 2 * H) hb[tid] = 0.f;

  // ---- C: layer-1 input projection
  const int unit = lane >> 1, half = lane & 1, dir = wave & 1;
  gru_w g;  // waves 0, 1: layer 1; waves 2, 3: layer 2
  {
    const float *a0p = (j < 13 ? feat + j * CF_FLD : img + CF_ALIAS_ROWS0 + (j - 13) * CF_FLD) + kk * 4;  // rows 13..15: in the image
    const int r1 = (lane & 3) < 3 ? (lane & 3) : 2;  // rows 16 + r1; lane % 4 == 3: row 19 does not exist, its sums are never stored
    const float *a1p = img + CF_ALIAS_REM + 16 * r1 + kk * 4;  // the remainder block: k-step ks is its row ks / 2, columns 48 (ks % 2) ..
    f32x4 acc[3] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    f32x4 rem[3] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    // operand ring, fully unrolled so that the slots are plain registers (a rotating copy would have to wait for the
    // load it copies): W three k-steps ahead (L2), feat one step ahead (LDS)
    float4 avq[2], rvq[2];
    avq[0] = *(const float4 *)(a0p);
    rvq[0] = *(const float4 *)(a1p);
    float bv1[3];  // the biases of this lane's three columns: requested now, added 40 k-steps later
#pragma unroll
    for (int n = 0; n < 3; ++n) bv1[n] = a.bx1[wave * 48 + n * 16 + j];
    // recurrent weights: requested here too (48 registers the projection does not need), used a barrier (waves 0, 1) or a whole
    // recurrence (waves 2, 3) later - requested behind the projection they arrived in the first ~800 cycles of phase D
    if (!FRONT_ONLY) gru_load_w(g, wave < 2 ? a.wh1 : a.wh2, wave < 2 ? a.bh1 : a.bh2, dir, unit, half);
#pragma unroll
    for (int ks = 0; ks < 40; ++ks) {
      if (ks + 3 < 40) {
#pragma unroll
        for (int n = 0; n < 3; ++n) bq[(ks + 3) & 3][n] = w_ld(ks + 3, n);
      }
      if (ks + 1 < 40) {
        avq[(ks + 1) & 1] = *(const float4 *)(a0p + (ks + 1) * 16);
        rvq[(ks + 1) & 1] = *(const float4 *)(a1p + ((ks + 1) >> 1) * 96 + ((ks + 1) & 1) * 48);
      }
      __builtin_amdgcn_sched_barrier(0);  // the prefetches stay HERE: sunk to just before their use, the LDS reads cost ~130 cycles per k-step
      const float4 av = avq[ks & 1], rv = rvq[ks & 1];
      const float4 *b = bq[ks & 3];
      CF_ROUND(av, rv, b, x) CF_ROUND(av, rv, b, y) CF_ROUND(av, rv, b, z) CF_ROUND(av, rv, b, w)
      __builtin_amdgcn_sched_barrier(0);
    }
    lds_barrier();  // gx takes the aliased block's place: every wave must have read its last remainder operand out of it
    // the four k sub-steps of a column sit in lanes col, col+16, col+32, col+48
#pragma unroll
    for (int n = 0; n < 3; ++n) {
      const int col = wave * 48 + n * 16 + j;
      const float bv = bv1[n];
#pragma unroll
      for (int r = 0; r < 4; ++r) gxs[(kk * 4 + r) * GR_GX_LD + col] = acc[n][r] + bv;
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        float v = rem[n][i];
        v += __shfl_xor(v, 16);
        v += __shfl_xor(v, 32);
        if (kk == 0) gxs[(16 + i) * GR_GX_LD + col] = v + bv;
      }
    }
  }
  CF_STAMP(4)
  __syncthreads();  // gx complete; nobody reads feat any more
  CF_STAMP(5)
  if (FRONT_ONLY) {
    // large batches: the recurrences run in their own kernel (gru_tail_kernel), where seven windows share a CU and no MFMA
    // stream competes with their serial chains; gx1 (14.6 KB per window) crosses memory instead of feat (48.6 KB)
    float4 *dst = (float4 *)(a.gx_out + (size_t)w * OT * 6 * H);
    for (int q = tid; q < OT * 6 * H / 4; q += CF_THREADS) {
      const int e = q * 4, t = e / (6 * H), c = e - t * 6 * H;
      dst[q] = *(const float4 *)(&gxs[t * GR_GX_LD + c]);
    }
    return;
  }

  cf_phases_d_to_g(a, img, feat, g, w);
}

// ------------------------------------------------------------------------------------------
// crnn_stream_kernel: crnn_fused_kernel for the streaming mode (spokestack/wakeword/tflite.py:187-188: the window slides
// by ONE mel row per posterior).  The conv and the layer-1 input projection act on each of the 19 time positions of a
// window separately, and position t looks at mel rows 8 t - 6 .. 8 t + 13 of the window: positions 1..17 never see the
// window's zero padding, so their projected rows gx1[t] are functions of 20 consecutive mel rows of the STREAM - the row
// that is position 17 of the window ending at stream row r is position 16 of the window ending at r + 8, ... position 1
// at r + 128.  Per stream a ring of WW_STREAM_GXC rows of gx1 (keyed by r mod ring size) keeps them: a new window
// computes THREE positions (0 and 18, which touch the padding, and the new interior one, 17, which it also stores) instead
// of 19, fetches the other sixteen rows (12 KB, L2) and runs the recurrences and the head as crnn_fused_kernel does.
// 1.3 instead of 8.0 MFLOP per posterior; what remains of the kernel's time is the recurrences (phases D..G).
// After a reset every slot holds the row of an all-zero field (the mel history is zeros), which is what the positions
// in front of the stream's first rows must see.  Rows differ from the batch kernel's in the last bits where the batch
// kernel computes a position on the 16-row MFMA tile and this one on the 4x4x1 form (different association of the k sum).
// ------------------------------------------------------------------------------------------
#ifndef CS_DEPTH
#define CS_DEPTH 8
#endif
struct stream_args {
  fused_args f;
  const int32_t *aux;  // [nw] stream * WW_STREAM_GXC + (stream rows incl. this window's newest) % WW_STREAM_GXC
  float *gxc;          // [S][WW_STREAM_GXC][192]
  // FE != 0 - ONE launch per tick: the streaming front end's side (common.h) and the model's filterbank
  ww_tick_fe fe;
  const int *start;
  const float *wpad, *bias;
  int n_mel;
  float floor_v, log_off, scale;
  const double *hann, *tw256, *tw512;
  long long *tstamps;  // development (WWHIP_CF_STAMPS): [workgroups][4 waves][6] s_memtime inside the tick prologue
};
#define CT_STAMP(i_) \
  if (sa.tstamps && lane == 0) sa.tstamps[((size_t)blockIdx.x * 4 + wave) * 6 + (i_)] = __builtin_amdgcn_s_memtime();

// Mel-side LDS of the one-launch tick form, in the part of the feat region the three conv rows leave free (floats from `feat`)
#define CT_X (3 * CF_FLD)                 // [WW_ST_RING] ring | the tick's new samples
#define CT_XS (CT_X + WW_ST_RING)         // [WW_CHUNK] int16: the raw samples
#define CT_WL (CT_XS + WW_CHUNK / 2)      // [3 rounds x 256 x 4] the mel weights [WW_MEL_TAPS][64], padded to whole store rounds
#define CT_MAG (CT_WL + 3 * 256 * 4)      // [2][260] magnitudes of the (at most) two new frames
#define CT_BUF (CT_MAG + 2 * 260 + 8)     // [2][FFT_LD] complex: the transforms' exchange buffers (16-byte aligned)
static_assert(CT_BUF % 4 == 0 && CT_BUF + 2 * FFT_LD * 4 <= CF_FEAT_FLOATS, "tick front end does not fit beside the conv rows");
static_assert(WW_MEL_TAPS * 64 <= 3 * 256 * 4 && WW_CHUNK * 2 == 40 * 16, "tick front end: store rounds");

// FE = 0: the windows of a tick come as descriptor tables behind a front-end kernel of its own (stream_frontend_kernel: two
// launches per tick).  FE = 1 / 2 (fp32 / fp64 transform) - ONE launch per tick (round 5): workgroup 2 s + k is window k of stream
// s's tick and transforms the new frames itself - waves 0 and 1, one frame each, straight into the window image, while the
// other waves stage the rows that were there before; spokestack/wakeword/tflite.py:148-191 for one stream.  Two workgroups of
// one stream run in the same launch, so exactly ONE of them writes the stream's state - the workgroup of the tick's newest
// window (k = n_windows - 1; workgroup 2 s when the tick has no window: the ring still advances, tflite.py:163-168) stores
// both new mel rows, the ring tail and the pre-emphasis carry - and its sibling must still see last tick's: the mel rows it
// reads are not the slots written this tick (the mirrored ring of T + 1 slots leaves exactly those out), the sample ring and
// the carry are kept twice and ping-pong by the stream's state parity.  The sibling transforms frame 0 a second time (same
// instructions, same bits) rather than wait for it.  No kernel boundary, no second launch, no descriptor tables; the
// posteriors go out as {value, tick number} pairs (ww_tick_tag) that the host polls.
template <int FE>
__global__ __launch_bounds__(CF_THREADS, 2) void crnn_stream_kernel(stream_args sa) {
  const fused_args &a = sa.f;
  extern __shared__ __align__(16) float cf_smem[];
  float *img = cf_smem, *feat = cf_smem + CF_IMG_FLOATS;
  constexpr int H = GR_H, RA = WW_STREAM_GXC;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 15, kk = lane >> 4;
  const int w = blockIdx.x;
  CF_STAMP(0)
  float4 wreg[CV_KB][2];
  float cb0, cb1;
  float4 crow0, crow1, crow2;
  int a_off, o_off[4];
  int q0;
  float *cache;
  auto load_conv_w = [&]() {
#pragma unroll
    for (int kb = 0; kb < CV_KB; ++kb)
#pragma unroll
      for (int n = 0; n < 2; ++n) wreg[kb][n] = *(const float4 *)(a.w4 + ((size_t)(kb * 4 + kk) * 32 + n * 16 + j) * 4);
    cb0 = a.cbias[j];
    cb1 = a.cbias[16 + j];
  };
  // ---- the sixteen cached rows (positions 1..16): requested first, parked in LDS once the image is dead
  // (three named registers: as an array they stayed in scratch memory)
  auto cached = [&](int q) {
    const int i = tid + q * CF_THREADS, tr = i / 48, c4 = i - tr * 48;  // 16 rows x 48 float4
    int slot = q0 + 8 * tr;
    slot = slot >= RA ? slot - RA : slot;
    return *(const float4 *)(cache + (size_t)slot * (6 * H) + c4 * 4);
  };
  // the three positions as 60 rows (p, f) of ONE m-tile per wave: p = 0, 1, 2 <-> t = 0, 17, 18
  auto tile_offsets = [&]() {
    constexpr int M = 3 * CV_OF;
    int m = wave * 16 + j;
    m = m < M ? m : M - 1;
    const int p = m / CV_OF, f = m - p * CV_OF, t = p ? 16 + p : 0;
    a_off = (f * CV_SF) * CV_LDT + t * CV_ST;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int mo = wave * 16 + kk * 4 + r, po = mo / CV_OF, fo = mo - po * CV_OF;
      o_off[r] = mo < M ? po * CF_FLD + fo * 32 + j : -1;
    }
  };
  auto scatter = [&](const float4 (&stage)[2], const int (&sidx)[2]) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      if (sidx[q] >= 0) {
        const int it = sidx[q] / 10, im = (sidx[q] - it * 10) * 4;
        const float e[4] = {stage[q].x, stage[q].y, stage[q].z, stage[q].w};
#pragma unroll
        for (int c = 0; c < 4; ++c) img[(im + c + CV_PF) * CV_LDT + it + CV_PT] = e[c];
      }
    }
  };
  if constexpr (FE == 0) {
    // the window's three descriptor words (first mel row, valid rows, cache slot) in ONE load: lanes 0..3 read the two halves of
    // row[w], valid[w] and aux[w] (three arrays, one instruction), the values come back as scalars - read one after the other they
    // were three round trips in a row at the head of every tick's kernel (the streaming path always passes both tables)
    const int *dp = lane == 0 ? (const int *)(a.wa.row + w) : lane == 1 ? (const int *)(a.wa.row + w) + 1
                  : lane == 2 ? (const int *)(a.wa.valid + w) : (const int *)(sa.aux + w);
    const int dv = *dp;
    int64_t row = (int64_t)(((unsigned long long)(unsigned)__builtin_amdgcn_readlane(dv, 1) << 32) | (unsigned)__builtin_amdgcn_readlane(dv, 0));
    int valid = __builtin_amdgcn_readlane(dv, 2);
    const int aux = __builtin_amdgcn_readlane(dv, 3);
    if (valid > a.T) valid = a.T;  // (as window_span)
    if (row + valid > a.wa.mel_rows) valid = (int)(a.wa.mel_rows - row);
    if (valid < 0) valid = 0;
    q0 = aux % RA;
    cache = sa.gxc + (size_t)(aux - q0) * (6 * H);
    load_conv_w();
    crow0 = cached(0); crow1 = cached(1); crow2 = cached(2);
    // ---- A: mel rows 0..13 (position 0) and 130..150 (positions 17, 18) of the window -> the transposed image
    const float *src = a.mel + row * CV_NMEL;  // 160-byte rows of a hipMalloc'ed history: 16-byte aligned
    float4 stage[2];
    int sidx[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int i = tid + q * CF_THREADS;            // 140 + 210 float4
      const int f4 = i < 140 ? i : 1300 + (i - 140);  // float4 index inside the [151][40] window
      const bool in = i < 350 && f4 / 10 < valid;
      sidx[q] = in ? f4 : -1;
      stage[q] = in ? *(const float4 *)(src + (size_t)f4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    for (int i = tid; i < CF_IMG_FLOATS / 4; i += CF_THREADS) ((float4 *)img)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    tile_offsets();
    __syncthreads();
    scatter(stage, sidx);
  } else {
    typedef typename std::conditional<FE == 2, double, float>::type R;
    const ww_tick_fe &fe = sa.fe;
    const int s = w >> 1, k = w & 1;
    // ---- over the bus, together: the tick's 320 samples (40 x 16 bytes) and the stream's control words
    uint4 raw = make_uint4(0u, 0u, 0u, 0u);
    if (tid < 40) raw = ((const uint4 *)(fe.frames + (size_t)s * WW_CHUNK))[tid];
    const int4 cw = ((const int4 *)fe.ctl)[s];
    // ---- device-side inputs that do not depend on the control words, requested while those are crossing the bus: the mel
    // weights, the sample ring (threads 0..127 ask for the copy of parity 0, the others for parity 1: 128 x 4 = 512 > fill),
    // both copies of the carry, the transform's constants (waves 0, 1: one new frame each), the conv weights
    f32x4 wlq[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const int i = tid + q * CF_THREADS;
      wlq[q] = ((const f32x4 *)sa.wpad)[i < WW_MEL_TAPS * 64 / 4 ? i : 0];
    }
    const int mel_st = lane < sa.n_mel ? sa.start[lane] : 0;
    const float mel_bias = lane < sa.n_mel ? sa.bias[lane] : 0.0f;
    const float4 ringq = ((const float4 *)(fe.ring + ((size_t)(tid >> 7) * fe.S + s) * WW_ST_RING))[tid & 127];
    const float carry0 = fe.prev[s], carry1 = fe.prev[fe.S + s];
    fft_consts<R> fc;
    if (wave < 2) fft_load_consts<R>(fc, lane, sa.hann, sa.tw256, sa.tw512);
    load_conv_w();
    // ---- what this workgroup is (uniform over it)
    const int fill = cw.x, nf = cw.y, flags = cw.z, pos = cw.w & 0xffff, rowq = cw.w >> 16;
    const int par = (flags >> 2) & 1;
    CT_STAMP(0)  // the control words are here
    const int np = (flags & 1) ? nf : 0;  // frames are analysed only while the VAD says speech (tflite.py:166)
    if ((flags & 2) || k >= (np > 1 ? np : 1)) return;  // an active stream is not sampled at all (tflite.py:139-140) | no second window
    const bool window = k < np, writer = k + 1 >= np;
    const int nfk = window ? k + 1 : 0;  // window k ends at new frame k: it needs frames 0..k
    const int slots = a.T + 1;
    float4 stage[2];
    int sidx[2] = {-1, -1};
    if (window) {
      // the T rows that end at new row k are the contiguous block that starts at slot (pos + k + 2) % (T + 1) of the mirrored ring
      int b = pos + k + 2;
      b = b >= slots ? b - slots : b;
      q0 = rowq + k + 1;  // rows since the reset incl. this window's newest, mod the cache ring
      q0 = q0 >= RA ? q0 - RA : q0;
      cache = sa.gxc + (size_t)s * RA * (6 * H);
      crow0 = cached(0); crow1 = cached(1); crow2 = cached(2);
      const float *src = a.mel + ((size_t)s * fe.HR + b) * CV_NMEL;
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int i = tid + q * CF_THREADS;            // 140 + 210 float4
        const int f4 = i < 140 ? i : 1300 + (i - 140);  // float4 index inside the [151][40] window
        const bool in = i < 350 && f4 / 10 < a.T - nfk;  // (the rows of this tick come from waves 0, 1)
        sidx[q] = in ? f4 : -1;
        stage[q] = in ? *(const float4 *)(src + (size_t)f4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
    float *fx = feat + CT_X, *fwl = feat + CT_WL, *fmag = feat + CT_MAG;
    short *fxs = (short *)(feat + CT_XS);
    cplx<R> *fbuf = (cplx<R> *)(feat + CT_BUF);
#pragma unroll
    for (int q = 0; q < 3; ++q) ((f32x4 *)fwl)[tid + q * CF_THREADS] = wlq[q];
    if ((tid >> 7) == par) ((float4 *)fx)[tid & 127] = ringq;
    if (tid < 40) ((uint4 *)fxs)[tid] = raw;
    if (window) {
      for (int i = tid; i < CF_IMG_FLOATS / 4; i += CF_THREADS) ((float4 *)img)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      tile_offsets();
    }
    CT_STAMP(1)
    __syncthreads();
    CT_STAMP(2)
    // ---- [ring | new samples]: normalise, clip, pre-emphasise (the arithmetic of stream_frontend_kernel, streams.hip)
    for (int i = tid; i < WW_CHUNK; i += CF_THREADS) {
      float v = __fdiv_rn((float)fxs[i], fe.divisor);
      if (fe.clip) v = fminf(fmaxf(v, -1.0f), 1.0f);
      float p;
      if (i == 0) {
        p = par ? carry1 : carry0;
      } else {
        p = __fdiv_rn((float)fxs[i - 1], fe.divisor);
        if (fe.clip) p = fminf(fmaxf(p, -1.0f), 1.0f);
      }
      fx[fill + i] = (fe.preemph != 0.0f) ? __fsub_rn(v, __fmul_rn(fe.preemph, p)) : v;
    }
    __syncthreads();
    CT_STAMP(3)
    if (writer && tid == 0) {
      float v = __fdiv_rn((float)fxs[WW_CHUNK - 1], fe.divisor);
      if (fe.clip) v = fminf(fmaxf(v, -1.0f), 1.0f);
      fe.prev[(size_t)(par ^ 1) * fe.S + s] = v;  // tflite.py:156-158: the carry is the un-emphasised last sample
    }
    // ---- new frames: wave f transforms frame f, its mel row goes straight into the window image (time T - nfk + f) and,
    // from the writer, into the stream's mirrored ring
    if (wave < nfk) {
      const float *srcx = fx + wave * fe.hop;
      auto x2 = [&](int n) -> float2 { return make_float2(srcx[2 * n], srcx[2 * n + 1]); };
      float *mg = fmag + wave * 260;
      frame_fft_mag<R>(x2, fc, fbuf + wave * FFT_LD, mg, lane);
      const float mv = mel_band(mg, fwl, mel_st, mel_bias, sa.floor_v, sa.log_off, sa.scale, lane);
      if (lane < CV_NMEL) {
        img[(lane + CV_PF) * CV_LDT + (a.T - nfk + wave) + CV_PT] = mv;
        if (writer) {
          int p = pos + wave;  // mirrored ring: the row goes to p % slots and p % slots + slots
          p = p >= slots ? p - slots : p;
          float *h = fe.hist + ((size_t)s * fe.HR + p) * CV_NMEL + lane;
          h[0] = mv;
          h[(size_t)slots * CV_NMEL] = mv;
        }
      }
    }
    if (writer) {  // keep the ring tail (for the next tick: the other copy)
      const int keep = fill + WW_CHUNK - nf * fe.hop;
      float *ring = fe.ring + ((size_t)(par ^ 1) * fe.S + s) * WW_ST_RING;
      for (int i = tid; i < keep; i += CF_THREADS) ring[i] = fx[nf * fe.hop + i];
    }
    CT_STAMP(4)  // (waves 0, 1: the new frames are in the image)
    if (!window) return;  // the tick has no window for this stream: its ring has advanced, that is all
    scatter(stage, sidx);
    CT_STAMP(5)
  }
  __syncthreads();
  CF_STAMP(1)
  // ---- B: conv of the 60 rows -> feat[p][f * 32 + channel]
  {
    float4 av[CV_KB];
    const float *abase = img + a_off;
#pragma unroll
    for (int kb = 0; kb < CV_KB; ++kb) {
      const int k4 = kb * 16 + kk * 4;
      const int kf = k4 / CV_KT, kt = k4 - kf * CV_KT;
      av[kb] = *(const float4 *)(abase + kf * CV_LDT + kt);
    }
    f32x4 acc0 = {cb0, cb0, cb0, cb0}, acc1 = {cb1, cb1, cb1, cb1};
#pragma unroll
    for (int kb = 0; kb < CV_KB; ++kb) {
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[kb].x, wreg[kb][0].x, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[kb].x, wreg[kb][1].x, acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[kb].y, wreg[kb][0].y, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[kb].y, wreg[kb][1].y, acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[kb].z, wreg[kb][0].z, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[kb].z, wreg[kb][1].z, acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[kb].w, wreg[kb][0].w, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[kb].w, wreg[kb][1].w, acc1, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (o_off[r] >= 0) {
        feat[o_off[r]] = relu1(acc0[r]);
        feat[o_off[r] + 16] = relu1(acc1[r]);
      }
    }
  }
  const float *wb = a.wx1s + ((size_t)kk * 192 + wave * 48 + j) * 4;
  constexpr size_t KS_STRIDE = (size_t)4 * 192 * 4;
  auto w_ld = [&](int ks, int n) { return *(const float4 *)(wb + ks * KS_STRIDE + n * 64); };
  // W_x1 ring: with three rows the projection is bound by how many bytes of W are in flight, not by the matrix pipe
  // (3.8 k cycles of MFMAs against 491 KB per workgroup): CS_DEPTH - 1 k-steps ahead instead of the batch kernel's three
  float4 bq[CS_DEPTH][3];
#pragma unroll
  for (int s2 = 0; s2 < CS_DEPTH - 1; ++s2)
#pragma unroll
    for (int n = 0; n < 3; ++n) bq[s2][n] = w_ld(s2, n);
  CF_STAMP(2)
  __syncthreads();  // feat complete; the image is dead from here on
  CF_STAMP(3)
  float *gxs = img + CF_GX, *seq1 = img + CF_SEQ, *hb = img + CF_HB;
  for (int i = tid; i < 32 * GR_SEQ_LD; i += CF_THREADS) seq1[i] = 0.f;
  if (tid < 2 * 2 * 2 * H) hb[tid] = 0.f;
  auto park = [&](int q, const float4 v) {
    const int i = tid + q * CF_THREADS, tr = i / 48, c4 = i - tr * 48;
    *(float4 *)(&gxs[(tr + 1) * GR_GX_LD + c4 * 4]) = v;
  };
  park(0, crow0); park(1, crow1); park(2, crow2);

  // ---- C: layer-1 input projection of the three rows on v_mfma_f32_4x4x1_16b_f32 (the batch kernel's 3-row remainder form)
  const int unit = lane >> 1, half = lane & 1, dir = wave & 1;
  gru_w g;
  {
    const int r1 = lane & 3;
    const float *a1p = feat + (r1 < 3 ? r1 : 2) * CF_FLD + kk * 4;  // lane % 4 == 3: no fourth row, its sums are never stored
    f32x4 rem[3] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    float4 rvq[2];
    rvq[0] = *(const float4 *)(a1p);
    float bv1[3];  // the biases of this lane's three columns: requested now, added when the products are complete
#pragma unroll
    for (int n = 0; n < 3; ++n) bv1[n] = a.bx1[wave * 48 + n * 16 + j];
    gru_load_w(g, wave < 2 ? a.wh1 : a.wh2, wave < 2 ? a.bh1 : a.bh2, dir, unit, half);  // (as crnn_fused_kernel: in front of the projection)
#pragma unroll
    for (int ks = 0; ks < 40; ++ks) {
      if (ks + CS_DEPTH - 1 < 40) {
#pragma unroll
        for (int n = 0; n < 3; ++n) bq[(ks + CS_DEPTH - 1) % CS_DEPTH][n] = w_ld(ks + CS_DEPTH - 1, n);
      }
      if (ks + 1 < 40) rvq[(ks + 1) & 1] = *(const float4 *)(a1p + (ks + 1) * 16);
      __builtin_amdgcn_sched_barrier(0);
      const float4 rv = rvq[ks & 1];
      const float4 *b = bq[ks % CS_DEPTH];
#define CS_ROUND(e_)                                                                 \
  rem[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(rv.e_, b[0].e_, rem[0], 0, 0, 0);     \
  rem[1] = __builtin_amdgcn_mfma_f32_4x4x1f32(rv.e_, b[1].e_, rem[1], 0, 0, 0);     \
  rem[2] = __builtin_amdgcn_mfma_f32_4x4x1f32(rv.e_, b[2].e_, rem[2], 0, 0, 0);
      CS_ROUND(x) CS_ROUND(y) CS_ROUND(z) CS_ROUND(w)
#undef CS_ROUND
      __builtin_amdgcn_sched_barrier(0);
    }
    int nslot = q0 + 128;
    nslot = nslot >= RA ? nslot - RA : nslot;
    float *crow_new = cache + (size_t)nslot * (6 * H);
#pragma unroll
    for (int n = 0; n < 3; ++n) {
      const int col = wave * 48 + n * 16 + j;
      const float bv = bv1[n];
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        float v = rem[n][i];
        v += __shfl_xor(v, 16);
        v += __shfl_xor(v, 32);
        if (kk == 0) {
          gxs[(i ? 16 + i : 0) * GR_GX_LD + col] = v + bv;
          if (i == 1) crow_new[col] = v + bv;  // position 17: position 16 of the window 8 rows on, ... position 1 of the one 128 on
        }
      }
    }
  }
  CF_STAMP(4)
  __syncthreads();  // gx complete; nobody reads feat any more
  CF_STAMP(5)
  cf_phases_d_to_g(a, img, feat, g, w);
}

// ------------------------------------------------------------------------------------------
// crnn_rows_kernel: conv + layer-1 input projection of SIXTEEN time positions per workgroup, for windows that slide over
// one mel sequence with a regular hop (utils/evaluate_models.py:66-73: hop 2).  Positions 1..17 of a window do not see
// its zero padding, so their projected rows are functions of 20 consecutive rows of the sequence and are shared by
// the windows that lie 8, 16, ... rows further on (crnn_stream_kernel is the one-row-at-a-time form of the same fact):
// per window 1 + 2 instead of 19 positions are computed - the new interior field and the two edge positions 0 and 18,
// which are the same conv with the taps over the padding cleared (conv_wL / conv_wR) applied to the sequence's real rows.
// Three position lists ("kinds"): interior fields at stride g = gcd(hop, 8) rows, left edges and right edges at stride hop.
// A workgroup stages the union of its 16 fields (15 stride + 20 <= 140 rows) once, runs the conv as 20 m-tiles
// (position, frequency) and the projection as ONE full 16-row MFMA tile per n-tile - no 3-row remainder, and W_x1 is
// streamed once per 16 rows that are then used by 16 x 17 / 3 windows.  gru_tail_kernel gathers a window's 19 rows.
// ------------------------------------------------------------------------------------------
struct rows_args {
  const float *mel;
  int64_t mel_rows;
  const float *w4[3];  // conv weights per kind: interior, left edge, right edge
  const float *cbias, *wx1s, *bx1;
  float *out[3];       // [count][192] per kind (b_x included)
  int64_t start[3];    // mel row of position 0's field (may lie outside the sequence: rows outside read as zeros)
  int stride[3], count[3], tiles[3];
  const struct rows_tile *desc;  // or: one descriptor per workgroup (several sequences in one buffer: ww_k_crnn_segments_forward)
};
struct rows_tile {
  int64_t start;    // mel row of the tile's first field
  int64_t out_row;  // row of out[kind] its first position goes to
  int32_t stride, count, kind, pad;
};

__global__ __launch_bounds__(CF_THREADS, 2) void crnn_rows_kernel(rows_args a) {
  extern __shared__ __align__(16) float cf_smem[];
  float *img = cf_smem, *feat = cf_smem + CF_IMG_FLOATS;  // feat: [16][CF_FLD]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 15, kk = lane >> 4;
  int kind = 0, stride, np;
  int64_t field0, out_row;
  if (a.desc) {
    const rows_tile d = a.desc[blockIdx.x];
    kind = d.kind; stride = d.stride; np = d.count; field0 = d.start; out_row = d.out_row;
  } else {
    int tile = blockIdx.x;
    if (tile >= a.tiles[0]) { tile -= a.tiles[0]; kind = 1; }
    if (kind == 1 && tile >= a.tiles[1]) { tile -= a.tiles[1]; kind = 2; }
    stride = a.stride[kind];
    const int p0 = tile * 16;
    np = a.count[kind] - p0 < 16 ? a.count[kind] - p0 : 16;
    field0 = a.start[kind] + (int64_t)p0 * stride;
    out_row = p0;
  }
  const float *w4 = a.w4[kind];

  float4 wreg[CV_KB][2];
#pragma unroll
  for (int kb = 0; kb < CV_KB; ++kb)
#pragma unroll
    for (int n = 0; n < 2; ++n) wreg[kb][n] = *(const float4 *)(w4 + ((size_t)(kb * 4 + kk) * 32 + n * 16 + j) * 4);
  const float cb0 = a.cbias[j], cb1 = a.cbias[16 + j];

  // ---- stage the union of the 16 fields: image[(mel + PF)][row - field0]
  int a_off[5], o_off[5][4];
  {
    constexpr int MAXV = 6;  // 6 * 256 float4 >= 140 * 40 / 4
    const int ncols = 15 * stride + CV_KT;
    const bool al16 = ((((uintptr_t)a.mel) & 15) == 0);
    float4 stage[MAXV];
    stage_loads(al16, [&](auto al) {
      constexpr bool AL = decltype(al)::value;
#pragma unroll
      for (int q = 0; q < MAXV; ++q) {
        const int f4 = q * CF_THREADS + tid, it = f4 / 10;
        const int64_t r = field0 + it, rc = r < 0 ? 0 : r < a.mel_rows ? r : a.mel_rows - 1;  // (the host checked: mel_rows >= T)
        stage[q] = ld_mel4_sel<AL>(a.mel + rc * CV_NMEL + (f4 - it * 10) * 4, it < ncols && r >= 0 && r < a.mel_rows);
      }
    });
    for (int i = tid; i < CF_IMG_FLOATS / 4; i += CF_THREADS) ((float4 *)img)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const int mt = wave + 4 * i, m = mt * 16 + j;   // 20 m-tiles: (position, frequency) rows
      const int p = m / CV_OF, f = m - p * CV_OF;
      a_off[i] = (f * CV_SF) * CV_LDT + p * stride;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int mo = mt * 16 + kk * 4 + r, po = mo / CV_OF, fo = mo - po * CV_OF;
        o_off[i][r] = po * CF_FLD + fo * 32 + j;
      }
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < MAXV; ++q) {
      const int f4 = q * CF_THREADS + tid, it = f4 / 10, im = (f4 - it * 10) * 4;
      if (it < ncols) {
        const float e[4] = {stage[q].x, stage[q].y, stage[q].z, stage[q].w};
#pragma unroll
        for (int c = 0; c < 4; ++c) img[(im + c + CV_PF) * CV_LDT + it] = e[c];
      }
    }
  }
  __syncthreads();
  // ---- conv: five m-tiles per wave, pipelined as in crnn_fused_kernel
  {
    auto load_a = [&](float4(&av)[CV_KB], int i) {
      const float *abase = img + a_off[i];
#pragma unroll
      for (int kb = 0; kb < CV_KB; ++kb) {
        const int k4 = kb * 16 + kk * 4;
        const int kf = k4 / CV_KT, kt = k4 - kf * CV_KT;
        av[kb] = *(const float4 *)(abase + kf * CV_LDT + kt);
      }
    };
    auto store_tile = [&](int i, const f32x4 &r0, const f32x4 &r1) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        feat[o_off[i][r]] = relu1(r0[r]);
        feat[o_off[i][r] + 16] = relu1(r1[r]);
      }
    };
#define CR_CONV_KB(kb_)                                                                              \
  acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i & 1][kb_].x, wreg[kb_][0].x, acc0, 0, 0, 0);       \
  acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i & 1][kb_].x, wreg[kb_][1].x, acc1, 0, 0, 0);       \
  acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i & 1][kb_].y, wreg[kb_][0].y, acc0, 0, 0, 0);       \
  acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i & 1][kb_].y, wreg[kb_][1].y, acc1, 0, 0, 0);       \
  acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i & 1][kb_].z, wreg[kb_][0].z, acc0, 0, 0, 0);       \
  acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i & 1][kb_].z, wreg[kb_][1].z, acc1, 0, 0, 0);       \
  acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i & 1][kb_].w, wreg[kb_][0].w, acc0, 0, 0, 0);       \
  acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i & 1][kb_].w, wreg[kb_][1].w, acc1, 0, 0, 0);
    float4 av[2][CV_KB];
    f32x4 prev0 = {0.f, 0.f, 0.f, 0.f}, prev1 = {0.f, 0.f, 0.f, 0.f};
    load_a(av[0], 0);
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      if (i + 1 < 5) load_a(av[(i + 1) & 1], i + 1);
      __builtin_amdgcn_sched_barrier(0);
      f32x4 acc0 = {cb0, cb0, cb0, cb0}, acc1 = {cb1, cb1, cb1, cb1};
      CR_CONV_KB(0) CR_CONV_KB(1) CR_CONV_KB(2)
      __builtin_amdgcn_sched_barrier(0);
      if (i > 0) store_tile(i - 1, prev0, prev1);
      __builtin_amdgcn_sched_barrier(0);
      CR_CONV_KB(3) CR_CONV_KB(4) CR_CONV_KB(5) CR_CONV_KB(6)
      prev0 = acc0;
      prev1 = acc1;
    }
    store_tile(4, prev0, prev1);
#undef CR_CONV_KB
  }
  const float *wb = a.wx1s + ((size_t)kk * 192 + wave * 48 + j) * 4;
  constexpr size_t KS_STRIDE = (size_t)4 * 192 * 4;
  auto w_ld = [&](int ks, int n) { return *(const float4 *)(wb + ks * KS_STRIDE + n * 64); };
  float4 bq[4][3];
#pragma unroll
  for (int s2 = 0; s2 < 3; ++s2)
#pragma unroll
    for (int n = 0; n < 3; ++n) bq[s2][n] = w_ld(s2, n);
  __syncthreads();  // feat complete
  // ---- projection: 16 rows = one MFMA tile per n-tile, 3 n-tiles per wave
  {
    const float *a0p = feat + j * CF_FLD + kk * 4;
    f32x4 acc[3] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    float4 avq[2];
    avq[0] = *(const float4 *)(a0p);
    float bv1[3];  // the biases of this lane's three columns: requested now, added when the products are complete
#pragma unroll
    for (int n = 0; n < 3; ++n) bv1[n] = a.bx1[wave * 48 + n * 16 + j];
#pragma unroll
    for (int ks = 0; ks < 40; ++ks) {
      if (ks + 3 < 40) {
#pragma unroll
        for (int n = 0; n < 3; ++n) bq[(ks + 3) & 3][n] = w_ld(ks + 3, n);
      }
      if (ks + 1 < 40) avq[(ks + 1) & 1] = *(const float4 *)(a0p + (ks + 1) * 16);
      __builtin_amdgcn_sched_barrier(0);
      const float4 av = avq[ks & 1];
      const float4 *b = bq[ks & 3];
#define CR_ROUND(e_)                                                                  \
  acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.e_, b[0].e_, acc[0], 0, 0, 0);    \
  acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.e_, b[1].e_, acc[1], 0, 0, 0);    \
  acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.e_, b[2].e_, acc[2], 0, 0, 0);
      CR_ROUND(x) CR_ROUND(y) CR_ROUND(z) CR_ROUND(w)
#undef CR_ROUND
      __builtin_amdgcn_sched_barrier(0);
    }
    float *out = a.out[kind] + (size_t)out_row * 192;
#pragma unroll
    for (int n = 0; n < 3; ++n) {
      const int col = wave * 48 + n * 16 + j;
      const float bv = bv1[n];
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (kk * 4 + r < np) out[(size_t)(kk * 4 + r) * 192 + col] = acc[n][r] + bv;
    }
  }
}

// ------------------------------------------------------------------------------------------
// crnn_fused_bf16_kernel (WW_PRECISION_BF16X3): phases A..C of crnn_fused_kernel on the bf16 matrix pipe with split
// operands, x = hi + lo (two bf16, 16 mantissa bits) and a*b = ah*bh + ah*bl + al*bh on v_mfma_f32_16x16x32_bf16 with
// fp32 accumulate - the arithmetic of the split-bf16 Wavenet (wavenet.hip).  48 matrix cycles per 16x16x32 products
// against 256 for eight v_mfma_f32_16x16x4_f32, and on a pipe that runs beside the vector ALU (the fp32 MFMA does not).
// Posteriors move by <= 1e-5 against the fp32 kernel (tests/test_gpu_parity.py); fp32 stays the default.
//   A  the window goes to LDS as two bf16 planes [mel + 1][frame + 8] (pairs of frames packed per 32-bit store)
//   B  conv TRANSPOSED: channels are the MFMA rows (weights = A operand), the 380 output positions its columns, so a
//      lane ends up with 4 consecutive channels of one position = one 8-byte store per plane into feat[t][f*32 + c].
//      K = (kf, kt'') with kt padded 20 -> 24 and shifted by 2, so that every group of 8 k is 16 aligned bytes of one
//      image row: 15 groups + 1 zero group = 4 k-steps of 32
//   C  projection: rows 0..15 and 16..18 (+13 clamped) as two MFMA row tiles, 3 n-tiles per wave, W_x1 hi/lo planes
//      streamed from L2 in B-operand order, two k-steps ahead
//   D..G as the fp32 kernel (cf_phases_d_to_g).
// ------------------------------------------------------------------------------------------
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
#define CB_LD 168   // image row, bf16 elements: frame + 8
#define CB_ROWS 43  // mel + 1
#define CB_FLD 648  // feat row, bf16 elements
#define CFB_IMG_FLOATS 7232  // two bf16 image planes (28,896 B) rounded up; the fp32 kernel's image region is 7,216 floats
#define CFB_SMEM_BYTES ((CFB_IMG_FLOATS + CF_FEAT_FLOATS) * 4)
static_assert(2 * CB_ROWS * CB_LD * 2 <= CFB_IMG_FLOATS * 4 && 2 * CV_OT * CB_FLD * 2 <= CF_FEAT_FLOATS * 4 && CF_W2S + 8 * 64 <= CFB_IMG_FLOATS,
              "bf16 planes / post-conv tenants exceed their LDS regions");

// (a, b) -> packed bf16 pairs hi and lo with a ~ hi.x + lo.x, b ~ hi.y + lo.y
__device__ __forceinline__ void split2_pair(float a, float b, unsigned &h, unsigned &l) {
  const bf16x2_t hh = __builtin_convertvector((f32x2_t){a, b}, bf16x2_t);
  h = __builtin_bit_cast(unsigned, hh);
  const bf16x2_t ll = __builtin_convertvector((f32x2_t){a - __uint_as_float(h << 16), b - __uint_as_float(h & 0xffff0000u)}, bf16x2_t);
  l = __builtin_bit_cast(unsigned, ll);
}

template <bool FRONT_ONLY>
__global__ __launch_bounds__(CF_THREADS, 2) void crnn_fused_bf16_kernel(fused_args a) {
  extern __shared__ __align__(16) float cf_smem[];
  float *img = cf_smem, *feat = cf_smem + CFB_IMG_FLOATS;
  unsigned short *imgh = (unsigned short *)img, *imgl = imgh + CB_ROWS * CB_LD;
  unsigned short *fth = (unsigned short *)feat, *ftl = fth + CV_OT * CB_FLD;
  constexpr int H = GR_H, OT = CV_OT, M = CV_OT * CV_OF;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 15, kk = lane >> 4;
  const int w = blockIdx.x;
  CF_STAMP(0)
  int64_t row;
  int valid;
  window_span(a.wa, w, a.T, row, valid);

  // conv weights, A operand (rows = channels): [plane][k-step 4][m-tile 2][lane][8 bf16]
  uint4 wq[2][4][2];
#pragma unroll
  for (int p = 0; p < 2; ++p)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) wq[p][ks][mt] = ((const uint4 *)a.cwb)[((p * 4 + ks) * 2 + mt) * 64 + lane];
  float cb[2][4];  // bias of this lane's channels mt*16 + kk*4 + r
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int r = 0; r < 4; ++r) cb[mt][r] = a.cbias[mt * 16 + kk * 4 + r];

  // ---- A: stage the window as bf16 hi / lo planes; a thread owns (frame pair, mel quad) items
  int b_off[6], f_off[6];  // per position tile of this wave: image offset of (t, f), feat offset of (t, f); -1 past M
  {
    constexpr int NIT = 3;  // 3 * 256 >= 76 frame pairs * 10 mel quads
    const float *src = a.mel + row * CV_NMEL;
    const bool al16 = ((((uintptr_t)src) & 15) == 0);
    float4 lo4[NIT], hi4[NIT];
#pragma unroll
    for (int q = 0; q < NIT; ++q) lo4[q] = hi4[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (valid > 0) stage_loads(al16, [&](auto al) {
      constexpr bool AL = decltype(al)::value;
#pragma unroll
      for (int q = 0; q < NIT; ++q) {
        const int item = q * CF_THREADS + tid, fp = item / 10, mq = item - fp * 10;
        const int f0 = 2 * fp, f1 = 2 * fp + 1;
        lo4[q] = ld_mel4_sel<AL>(src + (f0 < valid ? f0 : valid - 1) * CV_NMEL + mq * 4, f0 < valid);
        hi4[q] = ld_mel4_sel<AL>(src + (f1 < valid ? f1 : valid - 1) * CV_NMEL + mq * 4, f1 < valid);
      }
    });
    for (int i = tid; i < CFB_IMG_FLOATS / 4; i += CF_THREADS) ((float4 *)img)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const int m = (wave + 4 * i) * 16 + j, mc = m < M ? m : M - 1;
      const int t = mc / CV_OF, f = mc - t * CV_OF;
      b_off[i] = (2 * f) * CB_LD + 8 * t;
      f_off[i] = m < M ? t * CB_FLD + f * 32 : -1;
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < NIT; ++q) {
      const int item = q * CF_THREADS + tid, fp = item / 10, mq = item - fp * 10;
      if (2 * fp < a.T) {
        const float x0[4] = {lo4[q].x, lo4[q].y, lo4[q].z, lo4[q].w}, x1[4] = {hi4[q].x, hi4[q].y, hi4[q].z, hi4[q].w};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          unsigned h, l;
          split2_pair(x0[c], x1[c], h, l);
          const int o = (mq * 4 + c + 1) * CB_LD + 2 * fp + 8;  // even: one aligned 32-bit store per plane
          *(unsigned *)(imgh + o) = h;
          *(unsigned *)(imgl + o) = l;
        }
      }
    }
  }
  __syncthreads();
  CF_STAMP(1)

  // ---- B: conv (transposed) -> feat planes
  {
    auto load_x = [&](bf16x8_t(&xh)[4], bf16x8_t(&xl)[4], int i) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        int G = ks * 4 + kk;
        G = G < 15 ? G : 14;  // the 16th group has zero weights: any valid address
        const int kf = G / 3, h8 = (G - kf * 3) * 8;
        const int o = b_off[i] + kf * CB_LD + h8;
        xh[ks] = *(const bf16x8_t *)(imgh + o);
        xl[ks] = *(const bf16x8_t *)(imgl + o);
      }
    };
    auto store_pos = [&](int i, const f32x4 (&r)[2]) {
      if (f_off[i] >= 0) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
          unsigned h0, l0, h1, l1;
          split2_pair(fmaxf(r[mt][0], 0.f), fmaxf(r[mt][1], 0.f), h0, l0);
          split2_pair(fmaxf(r[mt][2], 0.f), fmaxf(r[mt][3], 0.f), h1, l1);
          const int o = f_off[i] + mt * 16 + kk * 4;
          *(uint2 *)(fth + o) = make_uint2(h0, h1);
          *(uint2 *)(ftl + o) = make_uint2(l0, l1);
        }
      }
    };
    bf16x8_t xh[2][4], xl[2][4];
    f32x4 prev[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    load_x(xh[0], xl[0], 0);
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      if (i + 1 < 6) load_x(xh[(i + 1) & 1], xl[(i + 1) & 1], i + 1);
      f32x4 acc[2] = {{cb[0][0], cb[0][1], cb[0][2], cb[0][3]}, {cb[1][0], cb[1][1], cb[1][2], cb[1][3]}};
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
          const bf16x8_t wh = __builtin_bit_cast(bf16x8_t, wq[0][ks][mt]), wl = __builtin_bit_cast(bf16x8_t, wq[1][ks][mt]);
          acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl, xh[i & 1][ks], acc[mt], 0, 0, 0);
          acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xl[i & 1][ks], acc[mt], 0, 0, 0);
          acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xh[i & 1][ks], acc[mt], 0, 0, 0);
        }
      if (i > 0) store_pos(i - 1, prev);
      prev[0] = acc[0];
      prev[1] = acc[1];
    }
    store_pos(5, prev);
  }
  // W_x1 planes, B operand: [plane][k-step 20][n-tile 12][lane][8 bf16]; this wave's n-tiles are 3 wave .. 3 wave + 2
  auto w_ld = [&](int p, int ks, int n) { return ((const uint4 *)a.wx1b)[((size_t)(p * 20 + ks) * 12 + wave * 3 + n) * 64 + lane]; };
  uint4 bq[3][2][3];  // [ring slot][plane][n]
#pragma unroll
  for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int n = 0; n < 3; ++n) bq[s2][p][n] = w_ld(p, s2, n);
  CF_STAMP(2)
  __syncthreads();  // feat complete; the image is dead from here on
  CF_STAMP(3)
  float *gxs = img + CF_GX, *seq1 = img + CF_SEQ, *hb = img + CF_HB;
  for (int i = tid; i < 32 * GR_SEQ_LD; i += CF_THREADS) seq1[i] = 0.f;
  if (tid < 2 * 2 * 2 * H) hb[tid] = 0.f;

  // ---- C: layer-1 input projection
  const int unit = lane >> 1, half = lane & 1, dir = wave & 1;
  gru_w g;
  {
    const int r1 = 16 + j < OT ? 16 + j : OT - 1;  // rows 19..31 of the second row tile do not exist: clamped, never stored
    const unsigned short *a0h = fth + j * CB_FLD + kk * 8, *a0l = ftl + j * CB_FLD + kk * 8;
    const unsigned short *a1h = fth + r1 * CB_FLD + kk * 8, *a1l = ftl + r1 * CB_FLD + kk * 8;
    f32x4 acc[2][3];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int n = 0; n < 3; ++n) acc[mt][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float bv1[3];  // the biases of this lane's three columns: requested now, added when the products are complete
#pragma unroll
    for (int n = 0; n < 3; ++n) bv1[n] = a.bx1[wave * 48 + n * 16 + j];
    if (!FRONT_ONLY) gru_load_w(g, wave < 2 ? a.wh1 : a.wh2, wave < 2 ? a.bh1 : a.bh2, dir, unit, half);  // (in front of the projection)
#pragma unroll
    for (int ks = 0; ks < 20; ++ks) {
      if (ks + 2 < 20) {
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
          for (int n = 0; n < 3; ++n) bq[(ks + 2) % 3][p][n] = w_ld(p, ks + 2, n);
      }
      const bf16x8_t ah[2] = {*(const bf16x8_t *)(a0h + ks * 32), *(const bf16x8_t *)(a1h + ks * 32)};
      const bf16x8_t al[2] = {*(const bf16x8_t *)(a0l + ks * 32), *(const bf16x8_t *)(a1l + ks * 32)};
#pragma unroll
      for (int n = 0; n < 3; ++n) {
        const bf16x8_t bh = __builtin_bit_cast(bf16x8_t, bq[ks % 3][0][n]), bl = __builtin_bit_cast(bf16x8_t, bq[ks % 3][1][n]);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
          acc[mt][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[mt], bh, acc[mt][n], 0, 0, 0);
          acc[mt][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[mt], bl, acc[mt][n], 0, 0, 0);
          acc[mt][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[mt], bh, acc[mt][n], 0, 0, 0);
        }
      }
    }
#pragma unroll
    for (int n = 0; n < 3; ++n) {
      const int col = wave * 48 + n * 16 + j;
      const float bv = bv1[n];
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int t = mt * 16 + kk * 4 + r;
          if (t < OT) gxs[t * GR_GX_LD + col] = acc[mt][n][r] + bv;
        }
    }
  }
  CF_STAMP(4)
  __syncthreads();
  CF_STAMP(5)
  if (FRONT_ONLY) {
    float4 *dst = (float4 *)(a.gx_out + (size_t)w * OT * 6 * H);
    for (int q = tid; q < OT * 6 * H / 4; q += CF_THREADS) {
      const int e = q * 4, t = e / (6 * H), c = e - t * 6 * H;
      dst[q] = *(const float4 *)(&gxs[t * GR_GX_LD + c]);
    }
    return;
  }
  cf_phases_d_to_g(a, img, feat, g, w);
}

// ------------------------------------------------------------------------------------------
// gru_tail_kernel: everything behind the layer-1 input projections for one window in a TWO-wave workgroup (wave 0
// forward, wave 1 backward), for large batches: 22.7 KB of LDS, 128 registers and 2 waves per window put seven windows on a CU, so the
// strictly serial recurrences (38 steps of ~500 cycles) of fourteen waves interleave on the vector ALUs instead of one
// window's sitting beside another's fp32 MFMA stream (which runs on the same datapath: measured 2x slower steps).
// Phases as D..G of crnn_fused_kernel; the layer-2 projection takes its B operands straight from L2.
// ------------------------------------------------------------------------------------------
struct tail_args {
  const float *gx1;   // [Nw][OT][192], or nullptr: rows gathered from the three lists of crnn_rows_kernel
  const float *wh1, *bh1;
  const float *wx2s;  // B-operand order [16][192][4]
  const float *bx2, *wh2, *bh2, *w1, *b1, *w2, *b2;
  float *enc, *out;
  int NOUT, HEAD;
  const float *gxI, *gxL, *gxR;  // interior fields [..][192], left / right edge rows [Nw][192]
  int hop_g, eight_g;            // hop / g and 8 / g: position t of window w is interior field w * hop_g + (t - 1) * eight_g
  const int64_t *iI0;            // or: window w's first interior field explicitly (several sequences)
  const float *wx2;              // W_x2 row-major [2*3H][2H] (gru_tail16_kernel)
};
// LDS (floats): gx [19][196] | seq1 [20][68] (row 19 zero) | h [2][2][2][32] | enc [64] | hid [64] = 21.9 KB: seven per CU whatever
// the allocation granularity (22.7 KB with a 20th gx row was seven only if LDS is handed out in units below 1 KB)
#define GT_SEQ (19 * GR_GX_LD)
#define GT_HB (GT_SEQ + 20 * GR_SEQ_LD)
#define GT_ENC (GT_HB + 2 * 2 * 2 * GR_H)
#define GT_HID (GT_ENC + 2 * GR_H)
#define GT_SMEM_FLOATS (GT_HID + 2 * GR_H)

__global__ __launch_bounds__(128, 4) void gru_tail_kernel(tail_args a) {
  constexpr int H = GR_H, OT = CV_OT;
  __shared__ __align__(16) float sm[GT_SMEM_FLOATS];
  float *gxs = sm, *seq1 = sm + GT_SEQ, *hb = sm + GT_HB, *encs = sm + GT_ENC, *hid = sm + GT_HID;
  const int tid = threadIdx.x, lane = tid & 63, dir = tid >> 6;
  const int j = lane & 15, kk = lane >> 4, unit = lane >> 1, half = lane & 1;
  const int w = blockIdx.x;
  gru_w g;
  gru_load_w(g, a.wh1, a.bh1, dir, unit, half);
  {
    // the window's 19 projected rows -> LDS: all of a thread's 16-byte loads first (unconditional, the last ones clamped), then
    // the stores - as a loop of "load, store" every element was a round trip to L2 of its own, eight in a row per window
    constexpr int NQ = OT * 6 * H / 4, NS = (NQ + 127) / 128;
    int64_t i0w = (int64_t)w * a.hop_g;
    if (!a.gx1 && a.iI0) i0w = a.iI0[w];
    f32x4 st[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const int q = min(tid + 128 * s, NQ - 1), t = q / 48, c4 = q - t * 48;
      const float *row = a.gx1 ? a.gx1 + ((size_t)w * OT + t) * 6 * H
                       : t == 0 ? a.gxL + (size_t)w * 6 * H
                       : t == OT - 1 ? a.gxR + (size_t)w * 6 * H
                                     : a.gxI + ((size_t)i0w + (size_t)(t - 1) * a.eight_g) * 6 * H;
      st[s] = *(const f32x4 *)(row + c4 * 4);
    }
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const int q = tid + 128 * s, t = q / 48, c4 = q - t * 48;
      if (q < NQ) *(f32x4 *)(&gxs[t * GR_GX_LD + c4 * 4]) = st[s];
    }
  }
  for (int i = tid; i < 20 * GR_SEQ_LD; i += 128) seq1[i] = 0.f;
  for (int i = tid; i < 2 * 2 * 2 * H; i += 128) hb[i] = 0.f;
  __syncthreads();
  cf_recurrence<true>(g, gxs, hb + dir * 2 * H, seq1, dir, unit, half);
  __syncthreads();
  // layer-2 input projection: wave d owns n-tiles 6 d .. 6 d + 5, in two passes of three
#pragma unroll 1
  for (int pass = 0; pass < 2; ++pass) {
    const int nt0 = dir * 6 + pass * 3;
    f32x4 acc[3] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    f32x4 rem[3] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    float4 bq[4][3];
#pragma unroll
    for (int kb = 0; kb < 4; ++kb)
#pragma unroll
      for (int n = 0; n < 3; ++n) bq[kb][n] = *(const float4 *)(a.wx2s + ((size_t)(kb * 4 + kk) * 192 + (nt0 + n) * 16 + j) * 4);
    float bb2[3];  // the biases with the weights (asked for where they are added, each was a round trip of its own)
#pragma unroll
    for (int n = 0; n < 3; ++n) bb2[n] = a.bx2[(nt0 + n) * 16 + j];
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
      const float4 av = *(const float4 *)(&seq1[j * GR_SEQ_LD + kb * 16 + kk * 4]);
      const float4 rv = *(const float4 *)(&seq1[(j < 3 ? 16 + j : 19) * GR_SEQ_LD + kb * 16 + kk * 4]);  // row 19 is zero
      const float4 *b = bq[kb];
      CF_ROUND_L2(av, rv, b, x) CF_ROUND_L2(av, rv, b, y) CF_ROUND_L2(av, rv, b, z) CF_ROUND_L2(av, rv, b, w)
    }
    __syncthreads();  // pass 0: both waves are done reading gx1 before anybody overwrites it (gx2 takes its place)
#pragma unroll
    for (int n = 0; n < 3; ++n) {
      const int col = (nt0 + n) * 16 + j;
      const float bb = bb2[n];
#pragma unroll
      for (int r = 0; r < 4; ++r) gxs[(kk * 4 + r) * GR_GX_LD + col] = acc[n][r] + bb;
      if (kk == 0) {
#pragma unroll
        for (int r = 0; r < 3; ++r) gxs[(16 + r) * GR_GX_LD + col] = rem[n][r] + bb;
      }
    }
  }
  // layer-2 recurrent weights: requested only now, when the projection's operands are dead - the kernel fits 128 registers
  // (four waves per SIMD instead of three); one L2 latency per window, partly behind the barrier
  gru_w g2;
  gru_load_w(g2, a.wh2, a.bh2, dir, unit, half);
  __syncthreads();
  const float h_last = cf_recurrence<false>(g2, gxs, hb + (2 + dir) * 2 * H, nullptr, dir, unit, half);
  if (half == 0) {
    encs[dir * H + unit] = h_last;
    if (a.enc) a.enc[(size_t)w * 2 * H + dir * H + unit] = h_last;
  }
  __syncthreads();
  if (dir == 0) {  // detect head, w1 row `lane` straight from L2 (16 KB, shared by every workgroup)
    const float b1v = a.b1[lane], b2v = a.b2[lane < a.NOUT ? lane : 0];
    float acc = 0.f;  // one chain in k order: bit-identical to crnn_fused_kernel's head
    const float4 *wr = (const float4 *)(a.w1 + (size_t)lane * 2 * H);
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const float4 wv = wr[q];
      const float4 ev = *(const float4 *)(&encs[q * 4]);
      acc = fmaf(wv.x, ev.x, acc); acc = fmaf(wv.y, ev.y, acc);
      acc = fmaf(wv.z, ev.z, acc); acc = fmaf(wv.w, ev.w, acc);
    }
    hid[lane] = fmaxf(acc + b1v, 0.f);
    wsync_g();
    float y = 0.f;
    if (lane < a.NOUT) {
      for (int k = 0; k < 2 * H; ++k) y = fmaf(a.w2[lane * 64 + k], hid[k], y);
      y += b2v;
    }
    if (a.HEAD == 0) {
      if (lane < a.NOUT) a.out[(size_t)w * a.NOUT + lane] = sigmoid_f(y);
    } else {
      float mx = (lane < a.NOUT) ? y : -INFINITY;
      for (int o = 1; o < 8; o <<= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
      float e = (lane < a.NOUT) ? expf(y - mx) : 0.f;
      float sum = e;
      for (int o = 1; o < 8; o <<= 1) sum += __shfl_xor(sum, o);
      if (lane < a.NOUT) a.out[(size_t)w * a.NOUT + lane] = e / sum;
    }
  }
}

// ------------------------------------------------------------------------------------------
// gru_tail16_kernel: the same work as gru_tail_kernel for SIXTEEN windows per two-wave workgroup (wave 0 forward, wave 1
// backward), with the recurrent mat-vec of the sixteen windows as one matrix product per step on v_mfma_f32_16x16x4_f32:
//   [16 windows x 32 units] x W_h^T [32 x 96]  =  6 n-tiles (z, r, c  x  unit half) x 8 k-steps = 48 MFMAs.
// A tile's accumulator has its unit (column) on the lane and four windows (rows) in its registers, and so do the tiles
// of the other gates - the gate arithmetic of a (window, unit) pair is lane-local, evaluated ONCE (gru_tail_kernel's lane
// pairs each evaluate a unit's gates twice), the projected inputs gx[window][t][gate unit] are read straight from global
// memory / L2 in that layout (one step ahead), and a step's new state goes through 2 KB of LDS to become the next step's A
// operand (lane = window, its eight k's of every k-step = eight consecutive units: two 16-byte reads).
// Layer 2's input projection (as many MFMAs as all four recurrences together) is computed per step in the same
// accumulators: A = seq1[t] of the sixteen windows (global workspace, written by layer 1), B = this direction's 96 rows of
// W_x2 held in registers (6 tiles x 16 k-steps), issued for step t+1 between writing h(t) and reading it back.
// Every sum is associated exactly as the vector form associates it (gru_step's four chains per gate; phase E's one chain per
// projected value, bias last) - v_mfma_f32_16x16x4_f32 is an fmaf chain in k order - so a window's posterior is the same bits
// whichever tail it was given to (tests/test_gpu_parity.py::test_crnn_large_batch_path).
// ------------------------------------------------------------------------------------------
#define GT16_LD 36   // h exchange row: 32 units + 4
#define GT16_ELD 68  // enc / hid row: 64 + 4
#define GT16_SMEM_BYTES ((2 * 2 * 16 * GT16_LD + 2 * 6 * 2 * 64 * 4) * 4)
struct tail16_args {
  tail_args t;
  float *seq;  // [workgroup][OT][16][64] layer-1 outputs
  int nw;      // windows in this launch (the last workgroup may be partial)
};

__global__ __launch_bounds__(128, 2) void gru_tail16_kernel(tail16_args aa) {
  constexpr int H = GR_H, OT = CV_OT;
  const tail_args &a = aa.t;
  // LDS: the h exchange (9.2 KB) + each wave's recurrent weights as B-operand pages [tile 6][half 2][lane 64] x 16 bytes
  // (24.6 KB: every lane reads back only what it wrote - 48 registers' worth that the four chains per tile need elsewhere);
  // the head's enc / hid rows move into the dead pages at the end.  33.8 KB: four workgroups per CU, as the registers allow
  // (dynamic: the launch asks for MORE than GT16_SMEM_BYTES when its grid does not fill the chip four workgroups deep - launch_tail)
  extern __shared__ __align__(16) float sm16[];
  float (*hs)[2][16 * GT16_LD] = (float (*)[2][16 * GT16_LD])sm16;
  float *encs = sm16 + 2 * 2 * 16 * GT16_LD, *hid = encs + 16 * GT16_ELD;
  static_assert(2 * 16 * GT16_ELD <= 2 * 6 * 2 * 64 * 4, "enc / hid rows do not fit the weight pages");
  const int tid = threadIdx.x, lane = tid & 63, dir = tid >> 6;
  const int c = lane & 15, g = lane >> 4;
  float4 *whl = (float4 *)(sm16 + 2 * 2 * 16 * GT16_LD) + (size_t)dir * 6 * 2 * 64 + lane;  // this lane's slot of page 0
  const int w0 = blockIdx.x * 16;
  float *seq = aa.seq + (size_t)blockIdx.x * OT * 16 * 2 * H;

  // rows of the accumulator tiles = windows 4 g + r (clamped in a partial workgroup: computed twice, stored once)
  int wi[4];
  int64_t i0[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) wi[r] = min(w0 + 4 * g + r, aa.nw - 1);
  if (!a.gx1 && a.iI0) {  // (one uniform branch around four loads in flight; as a per-row ternary they were four round trips in a row)
#pragma unroll
    for (int r = 0; r < 4; ++r) i0[r] = a.iI0[wi[r]];
  } else {
#pragma unroll
    for (int r = 0; r < 4; ++r) i0[r] = a.gx1 ? 0 : (int64_t)wi[r] * a.hop_g;
  }
  auto gx_row = [&](int r, int t) -> const float * {
    if (a.gx1) return a.gx1 + ((size_t)wi[r] * OT + t) * 6 * H;
    if (t == 0) return a.gxL + (size_t)wi[r] * 6 * H;
    if (t == OT - 1) return a.gxR + (size_t)wi[r] * 6 * H;
    return a.gxI + (size_t)(i0[r] + (int64_t)(t - 1) * a.eight_g) * 6 * H;
  };
  // recurrent weights as B operands: tile nt = gate * 2 + unit half.  The eight MFMAs of a tile and step are the four chains
  // of gru_step, two MFMAs each: chain (hh, p) = units 16 hh + p, + 2, .. + 14; its MFMA q takes the k-lanes g = 0..3 <-> unit
  // u = 16 hh + 8 q + 2 g + p.  Operand slot i = 4 hh + 2 q + p; in the LDS exchange row unit u sits at GT16_POS(u) =
  // 8 g + i, so that a lane group reads its eight operands as two 16-byte words.
#define GT16_POS(u_) (8 * (((u_) >> 1) & 3) + 4 * ((u_) >> 4) + 2 * (((u_) >> 3) & 1) + ((u_) & 1))
  float bh[6];
  auto load_wh = [&](const float *whp, const float *bhp) {
#pragma unroll
    for (int nt = 0; nt < 6; ++nt) {
      const int row = (nt >> 1) * H + (nt & 1) * 16 + c;
      const float *p = whp + ((size_t)dir * 3 * H + row) * H + 2 * g;
      float wv[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) wv[i] = p[16 * (i >> 2) + 8 * ((i >> 1) & 1) + (i & 1)];
      whl[(2 * nt) * 64] = make_float4(wv[0], wv[1], wv[2], wv[3]);
      whl[(2 * nt + 1) * 64] = make_float4(wv[4], wv[5], wv[6], wv[7]);
      bh[nt] = bhp[dir * 3 * H + row];
    }
  };
  // pre[nt] = pre-activation of tile nt for the four windows of this lane: (E0 + O0) + (E1 + O1), E0 starting from X0_[nt] (the
  // projected input of the z and r gates; zeros for c), O0 from b_h, the others from zero.  A tile's operands are read from
  // its page while the previous tile's MFMAs run; one tile's chains at a time (all six at once would need 96 registers)
#define GT16_PRE_ALL(X0_)                                                                          \
  {                                                                                                \
    float4 wb[2][2];                                                                               \
    wb[0][0] = whl[0];                                                                             \
    wb[0][1] = whl[64];                                                                            \
    _Pragma("unroll") for (int nt = 0; nt < 6; ++nt) {                                             \
      if (nt + 1 < 6) {                                                                            \
        wb[(nt + 1) & 1][0] = whl[(2 * nt + 2) * 64];                                              \
        wb[(nt + 1) & 1][1] = whl[(2 * nt + 3) * 64];                                              \
      }                                                                                            \
      __builtin_amdgcn_sched_barrier(0);                                                           \
      const float4 b0 = wb[nt & 1][0], b1 = wb[nt & 1][1];                                         \
      f32x4 e0 = nt < 4 ? X0_[nt < 4 ? nt : 0] : zero4, o0 = {bh[nt], bh[nt], bh[nt], bh[nt]};     \
      e0 = __builtin_amdgcn_mfma_f32_16x16x4f32(ha[0], b0.x, e0, 0, 0, 0);                         \
      o0 = __builtin_amdgcn_mfma_f32_16x16x4f32(ha[1], b0.y, o0, 0, 0, 0);                         \
      e0 = __builtin_amdgcn_mfma_f32_16x16x4f32(ha[2], b0.z, e0, 0, 0, 0);                         \
      o0 = __builtin_amdgcn_mfma_f32_16x16x4f32(ha[3], b0.w, o0, 0, 0, 0);                         \
      f32x4 e1 = zero4, o1 = zero4;                                                                \
      e1 = __builtin_amdgcn_mfma_f32_16x16x4f32(ha[4], b1.x, e1, 0, 0, 0);                         \
      o1 = __builtin_amdgcn_mfma_f32_16x16x4f32(ha[5], b1.y, o1, 0, 0, 0);                         \
      e1 = __builtin_amdgcn_mfma_f32_16x16x4f32(ha[6], b1.z, e1, 0, 0, 0);                         \
      o1 = __builtin_amdgcn_mfma_f32_16x16x4f32(ha[7], b1.w, o1, 0, 0, 0);                         \
      pre[nt] = (e0 + o0) + (e1 + o1);                                                             \
      __builtin_amdgcn_sched_barrier(0);                                                           \
    }                                                                                              \
  }
  // the first step: h = 0, the chains are their initial values (0 * w adds nothing, so the products are skipped)
#define GT16_PRE_FIRST(X0_)                                                                        \
  _Pragma("unroll") for (int nt = 0; nt < 6; ++nt) {                                               \
    const f32x4 o0 = {bh[nt], bh[nt], bh[nt], bh[nt]};                                             \
    pre[nt] = ((nt < 4 ? X0_[nt < 4 ? nt : 0] : zero4) + o0) + (zero4 + zero4);                    \
  }
  load_wh(a.wh1, a.bh1);
  for (int i = tid; i < 2 * 2 * 16 * GT16_LD; i += 128) (&hs[0][0][0])[i] = 0.f;
  const int pos0 = GT16_POS(c), pos1 = GT16_POS(16 + c);  // where this lane's two units sit in an exchange row

  // ---- layer 1 ------------------------------------------------------------------------------
  float h_own[4][2];  // [row r][unit half]: h of (window 4 g + r, unit 16 uh + c)
#pragma unroll
  for (int r = 0; r < 4; ++r) h_own[r][0] = h_own[r][1] = 0.f;
  float gxn[4][6];
  {
    const int t = dir ? OT - 1 : 0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float *row = gx_row(r, t) + dir * 3 * H + c;
#pragma unroll
      for (int nt = 0; nt < 6; ++nt) gxn[r][nt] = row[(nt >> 1) * H + (nt & 1) * 16];
    }
  }
  __syncthreads();
#pragma unroll 1
  for (int s = 0; s < OT; ++s) {
    const int t = dir ? OT - 1 - s : s, cur = s & 1;
    f32x4 x0[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) x0[nt] = (f32x4){gxn[0][nt], gxn[1][nt], gxn[2][nt], gxn[3][nt]};
    float gxc[4][2];
#pragma unroll
    for (int r = 0; r < 4; ++r) { gxc[r][0] = gxn[r][4]; gxc[r][1] = gxn[r][5]; }
    if (s + 1 < OT) {  // next step's projected inputs: in flight during this step's products and gates
      const int tn = dir ? t - 1 : t + 1;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float *row = gx_row(r, tn) + dir * 3 * H + c;
#pragma unroll
        for (int nt = 0; nt < 6; ++nt) gxn[r][nt] = row[(nt >> 1) * H + (nt & 1) * 16];
      }
    }
    f32x4 pre[6];
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    if (s > 0) {
      const float4 *hp = (const float4 *)(&hs[dir][cur][c * GT16_LD + 8 * g]);
      const float4 a0 = hp[0], a1 = hp[1];
      const float ha[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
      GT16_PRE_ALL(x0)
    } else {
      GT16_PRE_FIRST(x0)
    }
#pragma unroll
    for (int uh = 0; uh < 2; ++uh)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float z = fast_sigmoid(pre[uh][r]);
        const float rr = fast_sigmoid(pre[2 + uh][r]);
        const float hn = gru_blend(z, h_own[r][uh], gru_candidate(rr, pre[4 + uh][r], gxc[r][uh]));
        h_own[r][uh] = hn;
        hs[dir][cur ^ 1][(4 * g + r) * GT16_LD + (uh ? pos1 : pos0)] = hn;
        seq[((size_t)t * 16 + 4 * g + r) * 2 * H + dir * H + uh * 16 + c] = hn;
      }
    wsync_h();
  }
  __syncthreads();  // seq1 of both directions is in memory (workgroup scope)

  // ---- layer 2: the input projection of step t+1 is issued between writing h(t) and reading it back ---------------------
  // gx2[t][window][n] as phase E of crnn_fused_kernel computes it: ONE fmaf chain from zero over k = 16 kb + 4 kk + e in the
  // order kb, e, kk (kk = the k-lane), the bias added last.  A = seq1[t] of the sixteen windows (global workspace, written by
  // layer 1), B = this direction's 96 rows of W_x2 in registers: slot 4 kb + e of lane group g <-> column 16 kb + 4 g + e
  load_wh(a.wh2, a.bh2);
  float wx[6][16];
  float bx[6];
#pragma unroll
  for (int nt = 0; nt < 6; ++nt) {
    const int row = dir * 3 * H + (nt >> 1) * H + (nt & 1) * 16 + c;
    const float4 *p = (const float4 *)(a.wx2 + (size_t)row * 2 * H + 4 * g);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 v = p[4 * q];
      wx[nt][4 * q] = v.x; wx[nt][4 * q + 1] = v.y; wx[nt][4 * q + 2] = v.z; wx[nt][4 * q + 3] = v.w;
    }
    bx[nt] = a.bx2[row];
  }
  for (int i = tid; i < 2 * 2 * 16 * GT16_LD; i += 128) (&hs[0][0][0])[i] = 0.f;
#pragma unroll
  for (int r = 0; r < 4; ++r) h_own[r][0] = h_own[r][1] = 0.f;
  f32x4 gx2[6];  // projected inputs of the step at hand: tiles 0..3 start the z and r chains, 4..5 are the candidate's x part
  auto project = [&](int t) {
    const float4 *sp = (const float4 *)(seq + ((size_t)t * 16 + c) * 2 * H + 4 * g);
    const float4 q0 = sp[0], q1 = sp[4], q2 = sp[8], q3 = sp[12];
#pragma unroll
    for (int nt = 0; nt < 6; ++nt) gx2[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#define GT16_PROJ(q_, kb_)                                                                           \
  _Pragma("unroll") for (int nt = 0; nt < 6; ++nt) gx2[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(q_.x, wx[nt][4 * kb_ + 0], gx2[nt], 0, 0, 0); \
  _Pragma("unroll") for (int nt = 0; nt < 6; ++nt) gx2[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(q_.y, wx[nt][4 * kb_ + 1], gx2[nt], 0, 0, 0); \
  _Pragma("unroll") for (int nt = 0; nt < 6; ++nt) gx2[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(q_.z, wx[nt][4 * kb_ + 2], gx2[nt], 0, 0, 0); \
  _Pragma("unroll") for (int nt = 0; nt < 6; ++nt) gx2[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(q_.w, wx[nt][4 * kb_ + 3], gx2[nt], 0, 0, 0);
    GT16_PROJ(q0, 0)
    GT16_PROJ(q1, 1)
    GT16_PROJ(q2, 2)
    GT16_PROJ(q3, 3)
#undef GT16_PROJ
#pragma unroll
    for (int nt = 0; nt < 6; ++nt) gx2[nt] += (f32x4){bx[nt], bx[nt], bx[nt], bx[nt]};
  };
  project(dir ? OT - 1 : 0);
  __syncthreads();
#pragma unroll 1
  for (int s = 0; s < OT; ++s) {
    const int t = dir ? OT - 1 - s : s, cur = s & 1;
    f32x4 pre[6];
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    if (s > 0) {
      const float4 *hp = (const float4 *)(&hs[dir][cur][c * GT16_LD + 8 * g]);
      const float4 a0 = hp[0], a1 = hp[1];
      const float ha[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
      GT16_PRE_ALL(gx2)
    } else {
      GT16_PRE_FIRST(gx2)
    }
#pragma unroll
    for (int uh = 0; uh < 2; ++uh)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float z = fast_sigmoid(pre[uh][r]);
        const float rr = fast_sigmoid(pre[2 + uh][r]);
        const float hn = gru_blend(z, h_own[r][uh], gru_candidate(rr, pre[4 + uh][r], gx2[4 + uh][r]));
        h_own[r][uh] = hn;
        hs[dir][cur ^ 1][(4 * g + r) * GT16_LD + (uh ? pos1 : pos0)] = hn;
      }
    if (s + 1 < OT) project(dir ? t - 1 : t + 1);  // between the h write and its read-back: covers the LDS round trip
    wsync_h();
  }
  __syncthreads();  // both waves are done with their weight pages: enc / hid take their place
#pragma unroll
  for (int uh = 0; uh < 2; ++uh)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      encs[(4 * g + r) * GT16_ELD + dir * H + uh * 16 + c] = h_own[r][uh];
      if (a.enc && w0 + 4 * g + r < aa.nw) a.enc[(size_t)(w0 + 4 * g + r) * 2 * H + dir * H + uh * 16 + c] = h_own[r][uh];
    }
  __syncthreads();

  // ---- detect head: wave d takes windows 8 d .. 8 d + 7; lane = hidden unit, its w1 row in registers -----
  {
    float4 w1r[16];
    const float4 *wr = (const float4 *)(a.w1 + (size_t)lane * 2 * H);
#pragma unroll
    for (int q = 0; q < 16; ++q) w1r[q] = wr[q];
    const float b1v = a.b1[lane];
#pragma unroll 1
    for (int wq = dir * 8; wq < dir * 8 + 8; ++wq) {
      float acc = 0.f;  // one chain in k order: bit-identical to crnn_fused_kernel's head
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const float4 ev = *(const float4 *)(&encs[wq * GT16_ELD + q * 4]);
        acc = fmaf(w1r[q].x, ev.x, acc); acc = fmaf(w1r[q].y, ev.y, acc);
        acc = fmaf(w1r[q].z, ev.z, acc); acc = fmaf(w1r[q].w, ev.w, acc);
      }
      hid[wq * GT16_ELD + lane] = fmaxf(acc + b1v, 0.f);
    }
  }
  __syncthreads();
  {
    const int wq = tid >> 3, o = tid & 7;  // 16 windows x up to 8 outputs
    float y = 0.f;
    if (o < a.NOUT) {
      for (int k = 0; k < 2 * H; ++k) y = fmaf(a.w2[o * 64 + k], hid[wq * GT16_ELD + k], y);
      y += a.b2[o];
    }
    const bool live = o < a.NOUT && w0 + wq < aa.nw;
    if (a.HEAD == 0) {
      if (live) a.out[(size_t)(w0 + wq) * a.NOUT + o] = sigmoid_f(y);
    } else {
      float mx = (o < a.NOUT) ? y : -INFINITY;
      for (int d = 1; d < 8; d <<= 1) mx = fmaxf(mx, __shfl_xor(mx, d));
      const float e = (o < a.NOUT) ? expf(y - mx) : 0.f;
      float sum = e;
      for (int d = 1; d < 8; d <<= 1) sum += __shfl_xor(sum, d);
      if (live) a.out[(size_t)(w0 + wq) * a.NOUT + o] = e / sum;
    }
  }
}


#ifndef WW_TAIL16H
#define WW_TAIL16H 0
#endif
#if WW_TAIL16H
// ------------------------------------------------------------------------------------------
// gru_tail16h_kernel (round 6): gru_tail16_kernel's work for TWO groups of sixteen windows per four-wave workgroup, with the
// layer-2 input projection HOISTED out of the 19-step recurrence into a phase of its own:
//   * the recurrent weight pages (24.6 KB of LDS) are shared by the two groups - 43 KB per workgroup instead of 2 x 33.8:
//     three workgroups = twelve waves per CU, THREE waves per SIMD where gru_tail16_kernel has two;
//   * W_x2 (96 registers) is alive only while the projection phase runs: gx2[t] of all 19 steps - the same MFMA chains, bias
//     last - goes to memory in the layout layer 1 reads its projected inputs in (over the group's own rows of gx1 when the
//     launch has them to itself, else into aa.gx2), and the layer-2 recurrence IS the layer-1 loop on those rows;
//   * every sum is associated as before, so a posterior has the same bits whichever tail served it.
// MEASURED AND NOT ADOPTED (profiles/r06/tail16_probes.txt): bit-identical to gru_tail16_kernel, but 16-20 % slower per window at
// three waves per SIMD than that kernel at two (690.9 vs 596.9 us per 49,152 windows, 364.4 vs 360.5 per 24,576): the
// projection's MFMAs no longer cover the h round trip of the recurrence and 29 KB per window take a trip through memory.
// Kept as a development build (-DWW_TAIL16H=1: WW_OPT_CRNN_TAIL_MFMA = 3 selects it for launches that own their gx1 rows; the
// prologue and the head spill ~25 registers at the 168 the occupancy allows - outside the loops, but it is not a shipped kernel).
// ------------------------------------------------------------------------------------------
#define GT16H_HS (2 * 2 * 2 * 16 * GT16_LD)  // floats: h exchange [group 2][direction 2][buffer 2][16 x GT16_LD]
#define GT16H_SMEM_BYTES ((GT16H_HS + 2 * 6 * 2 * 64 * 4) * 4)
struct tail16h_args {
  tail_args t;
  float *seq;   // [group][OT][16][64] layer-1 outputs
  float *gx2;   // [nw][OT][192] projected layer-2 inputs, or nullptr: written over the group's own rows of t.gx1
  int nw;
};

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void gru_tail16h_kernel(tail16h_args aa) {
  constexpr int H = GR_H, OT = CV_OT;
  const tail_args &a = aa.t;
  extern __shared__ __align__(16) float sm16[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, grp = wave >> 1, dir = wave & 1;
  const int c = lane & 15, g = lane >> 4;
  float (*hs)[2][16 * GT16_LD] = (float (*)[2][16 * GT16_LD])(sm16 + grp * 2 * 2 * 16 * GT16_LD);  // this group's exchange
  float *pages = sm16 + GT16H_HS;
  float *encs = pages + grp * 2 * 16 * GT16_ELD, *hid = encs + 16 * GT16_ELD;
  static_assert(2 * 2 * 16 * GT16_ELD <= 2 * 6 * 2 * 64 * 4, "enc / hid rows of two groups do not fit the weight pages");
  float4 *whl = (float4 *)pages + (size_t)dir * 6 * 2 * 64 + lane;  // this lane's slot of page 0 (both groups read the same pages)
  const int gi = blockIdx.x * 2 + grp;  // group number in the launch
  const int w0 = gi * 16;
  float *seq = aa.seq + (size_t)gi * OT * 16 * 2 * H;

  // This form serves launches that own their projected inputs t.gx1 [window][t][192] (the rows of a window are nobody else's:
  // front + tail over explicit windows): a lane's four windows (rows of its accumulator tiles, 4 g + r; clamped in a partial
  // group: computed twice, stored once) as 32-bit element offsets - the index arithmetic of the three gx sources that
  // gru_tail16_kernel carries through its loops costs registers this kernel does not have.
  int ro[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) ro[r] = min(w0 + 4 * g + r, aa.nw - 1) * (OT * 6 * H) + dir * 3 * H + c;
  const float *const g1base = a.gx1;
  // where layer 2's projected inputs go: the same layout, over the group's own rows of gx1 (or aa.gx2 when the caller gave one)
  float *const g2base = aa.gx2 ? aa.gx2 : const_cast<float *>(a.gx1);
  auto gx_row = [&](int r, int t) -> const float * { return g1base + ro[r] + t * (6 * H); };
  auto gx2_row = [&](int r, int t) -> float * { return g2base + ro[r] + t * (6 * H); };
  // (GT16_POS, GT16_PRE_ALL, GT16_PRE_FIRST: gru_tail16_kernel's, still defined)
  float bh[6];
  auto load_wh = [&](const float *whp, const float *bhp) {  // group 0 writes the pages, everybody takes the biases
#pragma unroll
    for (int nt = 0; nt < 6; ++nt) {
      const int row = (nt >> 1) * H + (nt & 1) * 16 + c;
      if (grp == 0) {
        const float *p = whp + ((size_t)dir * 3 * H + row) * H + 2 * g;
        float wv[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) wv[i] = p[16 * (i >> 2) + 8 * ((i >> 1) & 1) + (i & 1)];
        whl[(2 * nt) * 64] = make_float4(wv[0], wv[1], wv[2], wv[3]);
        whl[(2 * nt + 1) * 64] = make_float4(wv[4], wv[5], wv[6], wv[7]);
      }
      bh[nt] = bhp[dir * 3 * H + row];
    }
  };
  const int pos0 = GT16_POS(c), pos1 = GT16_POS(16 + c);
  float h_own[4][2];
  float gxn[4][6];
  // One layer's recurrence over projected inputs in the [window][t][192] layout (LAYER 1: gx_row, from the kernels in front; LAYER
  // 2: gx2_row, from this kernel's projection phase): the loop of gru_tail16_kernel's layer 1, statement for statement.
#define GT16H_RECUR(ROW_, STORE_SEQ_)                                                                         \
  {                                                                                                           \
    _Pragma("unroll") for (int r = 0; r < 4; ++r) h_own[r][0] = h_own[r][1] = 0.f;                             \
    {                                                                                                         \
      const int t = dir ? OT - 1 : 0;                                                                         \
      _Pragma("unroll") for (int r = 0; r < 4; ++r) {                                                         \
        const float *row = ROW_(r, t);                                                      \
        _Pragma("unroll") for (int nt = 0; nt < 6; ++nt) gxn[r][nt] = row[(nt >> 1) * H + (nt & 1) * 16];      \
      }                                                                                                       \
    }                                                                                                         \
    __syncthreads();                                                                                          \
    _Pragma("unroll 1") for (int s = 0; s < OT; ++s) {                                                        \
      const int t = dir ? OT - 1 - s : s, cur = s & 1;                                                        \
      f32x4 x0[4];                                                                                            \
      _Pragma("unroll") for (int nt = 0; nt < 4; ++nt) x0[nt] = (f32x4){gxn[0][nt], gxn[1][nt], gxn[2][nt], gxn[3][nt]}; \
      float gxc[4][2];                                                                                        \
      _Pragma("unroll") for (int r = 0; r < 4; ++r) { gxc[r][0] = gxn[r][4]; gxc[r][1] = gxn[r][5]; }          \
      if (s + 1 < OT) {                                                                                       \
        const int tn = dir ? t - 1 : t + 1;                                                                   \
        _Pragma("unroll") for (int r = 0; r < 4; ++r) {                                                       \
          const float *row = ROW_(r, tn);                                                   \
          _Pragma("unroll") for (int nt = 0; nt < 6; ++nt) gxn[r][nt] = row[(nt >> 1) * H + (nt & 1) * 16];    \
        }                                                                                                     \
      }                                                                                                       \
      const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};                                                               \
      float ha[8];                                                                                            \
      if (s > 0) {                                                                                            \
        const float4 *hp = (const float4 *)(&hs[dir][cur][c * GT16_LD + 8 * g]);                              \
        const float4 a0 = hp[0], a1 = hp[1];                                                                  \
        ha[0] = a0.x; ha[1] = a0.y; ha[2] = a0.z; ha[3] = a0.w; ha[4] = a1.x; ha[5] = a1.y; ha[6] = a1.z; ha[7] = a1.w; \
      }                                                                                                       \
      /* the six tiles in the order z, r, c of unit half 0, then of unit half 1: a half's gates are evaluated as soon as its  \
         three tiles are in (12 pre-activation registers alive instead of 24); a tile's sums are gru_tail16_kernel's */     \
      float4 wb[2][2];                                                                                        \
      if (s > 0) { wb[0][0] = whl[0]; wb[0][1] = whl[64]; }                                                   \
      _Pragma("unroll") for (int uh = 0; uh < 2; ++uh) {                                                      \
        f32x4 pre3[3];                                                                                        \
        _Pragma("unroll") for (int gt = 0; gt < 3; ++gt) {                                                    \
          const int nt = 2 * gt + uh, q = 3 * uh + gt;                /* q: position in the visiting order */ \
          const f32x4 xin = gt < 2 ? x0[nt < 4 ? nt : 0] : zero4;                                             \
          const f32x4 o0i = {bh[nt], bh[nt], bh[nt], bh[nt]};                                                 \
          if (s > 0) {                                                                                        \
            if (q + 1 < 6) {                                                                                  \
              const int ntn = 2 * ((q + 1) % 3) + (q + 1) / 3;                                                \
              wb[(q + 1) & 1][0] = whl[(2 * ntn) * 64];                                                       \
              wb[(q + 1) & 1][1] = whl[(2 * ntn + 1) * 64];                                                   \
            }                                                                                                 \
            __builtin_amdgcn_sched_barrier(0);                                                                \
            const float4 b0 = wb[q & 1][0], b1 = wb[q & 1][1];                                                \
            f32x4 e0 = xin, o0 = o0i;                                                                         \
            e0 = __builtin_amdgcn_mfma_f32_16x16x4f32(ha[0], b0.x, e0, 0, 0, 0);                              \
            o0 = __builtin_amdgcn_mfma_f32_16x16x4f32(ha[1], b0.y, o0, 0, 0, 0);                              \
            e0 = __builtin_amdgcn_mfma_f32_16x16x4f32(ha[2], b0.z, e0, 0, 0, 0);                              \
            o0 = __builtin_amdgcn_mfma_f32_16x16x4f32(ha[3], b0.w, o0, 0, 0, 0);                              \
            f32x4 e1 = zero4, o1 = zero4;                                                                     \
            e1 = __builtin_amdgcn_mfma_f32_16x16x4f32(ha[4], b1.x, e1, 0, 0, 0);                              \
            o1 = __builtin_amdgcn_mfma_f32_16x16x4f32(ha[5], b1.y, o1, 0, 0, 0);                              \
            e1 = __builtin_amdgcn_mfma_f32_16x16x4f32(ha[6], b1.z, e1, 0, 0, 0);                              \
            o1 = __builtin_amdgcn_mfma_f32_16x16x4f32(ha[7], b1.w, o1, 0, 0, 0);                              \
            pre3[gt] = (e0 + o0) + (e1 + o1);                                                                 \
            __builtin_amdgcn_sched_barrier(0);                                                                \
          } else {                                                                                            \
            pre3[gt] = (xin + o0i) + (zero4 + zero4);   /* h = 0: the chains are their initial values */      \
          }                                                                                                   \
        }                                                                                                     \
        _Pragma("unroll") for (int r = 0; r < 4; ++r) {                                                       \
          const float z = fast_sigmoid(pre3[0][r]);                                                           \
          const float rr = fast_sigmoid(pre3[1][r]);                                                          \
          const float hn = gru_blend(z, h_own[r][uh], gru_candidate(rr, pre3[2][r], gxc[r][uh]));             \
          h_own[r][uh] = hn;                                                                                  \
          hs[dir][cur ^ 1][(4 * g + r) * GT16_LD + (uh ? pos1 : pos0)] = hn;                                  \
          if (STORE_SEQ_) seq[((size_t)t * 16 + 4 * g + r) * 2 * H + dir * H + uh * 16 + c] = hn;             \
        }                                                                                                     \
      }                                                                                                       \
      wsync_h();                                                                                              \
    }                                                                                                         \
  }

  // ---- layer 1 ------------------------------------------------------------------------------
  load_wh(a.wh1, a.bh1);
  for (int i = tid; i < GT16H_HS; i += 256) sm16[i] = 0.f;
  GT16H_RECUR(gx_row, true)
  __syncthreads();  // seq1 of both directions is in memory; every wave is done with the layer-1 pages and with its rows of gx1

  // ---- layer 2, projection phase: gx2[t] for all t, one fmaf chain per value over k = 16 kb + 4 kk + e in the order kb, e, kk,
  //      the bias added last (phase E of crnn_fused_kernel); A = seq1[t] of the sixteen windows, B = this direction's 96 rows of W_x2
  load_wh(a.wh2, a.bh2);
  // Three of the six n-tiles at a time (48 registers of W_x2 instead of 96: the phase fits three waves per SIMD without a spill);
  // seq1[t] is read once per half - 2 KB per wave and step out of L2.  Every value is still ONE chain in k order, bias last.
#define GT16H_PROJ_HALF(NT0_)                                                                                   \
  {                                                                                                             \
    float wx[3][16];                                                                                            \
    float bx[3];                                                                                                \
    _Pragma("unroll") for (int i = 0; i < 3; ++i) {                                                             \
      const int nt = (NT0_) + i;                                                                                \
      const int row = dir * 3 * H + (nt >> 1) * H + (nt & 1) * 16 + c;                                          \
      const float4 *p = (const float4 *)(a.wx2 + (size_t)row * 2 * H + 4 * g);                                  \
      _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                           \
        const float4 v = p[4 * q];                                                                              \
        wx[i][4 * q] = v.x; wx[i][4 * q + 1] = v.y; wx[i][4 * q + 2] = v.z; wx[i][4 * q + 3] = v.w;             \
      }                                                                                                         \
      bx[i] = a.bx2[row];                                                                                       \
    }                                                                                                           \
    _Pragma("unroll 1") for (int t = 0; t < OT; ++t) {                                                          \
      const float4 *sp = (const float4 *)(seq + ((size_t)t * 16 + c) * 2 * H + 4 * g);                          \
      const float4 q0 = sp[0], q1 = sp[4], q2 = sp[8], q3 = sp[12];                                             \
      f32x4 gx2[3];                                                                                             \
      _Pragma("unroll") for (int i = 0; i < 3; ++i) gx2[i] = (f32x4){0.f, 0.f, 0.f, 0.f};                       \
      GT16H_PROJ(q0, 0)                                                                                         \
      GT16H_PROJ(q1, 1)                                                                                         \
      GT16H_PROJ(q2, 2)                                                                                         \
      GT16H_PROJ(q3, 3)                                                                                         \
      _Pragma("unroll") for (int i = 0; i < 3; ++i) gx2[i] += (f32x4){bx[i], bx[i], bx[i], bx[i]};              \
      _Pragma("unroll") for (int r = 0; r < 4; ++r) {                                                           \
        float *row = gx2_row(r, t);                                                           \
        _Pragma("unroll") for (int i = 0; i < 3; ++i) row[(((NT0_) + i) >> 1) * H + (((NT0_) + i) & 1) * 16] = gx2[i][r]; \
      }                                                                                                         \
    }                                                                                                           \
  }
#define GT16H_PROJ(q_, kb_)                                                                          \
  _Pragma("unroll") for (int i = 0; i < 3; ++i) gx2[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(q_.x, wx[i][4 * kb_ + 0], gx2[i], 0, 0, 0); \
  _Pragma("unroll") for (int i = 0; i < 3; ++i) gx2[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(q_.y, wx[i][4 * kb_ + 1], gx2[i], 0, 0, 0); \
  _Pragma("unroll") for (int i = 0; i < 3; ++i) gx2[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(q_.z, wx[i][4 * kb_ + 2], gx2[i], 0, 0, 0); \
  _Pragma("unroll") for (int i = 0; i < 3; ++i) gx2[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(q_.w, wx[i][4 * kb_ + 3], gx2[i], 0, 0, 0);
  GT16H_PROJ_HALF(0)
  GT16H_PROJ_HALF(3)
#undef GT16H_PROJ
#undef GT16H_PROJ_HALF
  for (int i = tid; i < GT16H_HS; i += 256) sm16[i] = 0.f;
  __threadfence_block();  // (a lane reads back what it - or its clamped twin with the same values - wrote: program order suffices)
  // ---- layer 2, recurrence: the layer-1 loop on the rows written above
  GT16H_RECUR(gx2_row, false)
  __syncthreads();  // every wave is done with the weight pages: enc / hid take their place
#pragma unroll
  for (int uh = 0; uh < 2; ++uh)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      encs[(4 * g + r) * GT16_ELD + dir * H + uh * 16 + c] = h_own[r][uh];
      if (a.enc && w0 + 4 * g + r < aa.nw) a.enc[(size_t)(w0 + 4 * g + r) * 2 * H + dir * H + uh * 16 + c] = h_own[r][uh];
    }
  __syncthreads();
  // ---- detect head per group: wave `dir` takes windows 8 dir .. 8 dir + 7; lane = hidden unit, its w1 row in registers
  {
    float4 w1r[16];
    const float4 *wr = (const float4 *)(a.w1 + (size_t)lane * 2 * H);
#pragma unroll
    for (int q = 0; q < 16; ++q) w1r[q] = wr[q];
    const float b1v = a.b1[lane];
#pragma unroll 1
    for (int wq = dir * 8; wq < dir * 8 + 8; ++wq) {
      float acc = 0.f;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const float4 ev = *(const float4 *)(&encs[wq * GT16_ELD + q * 4]);
        acc = fmaf(w1r[q].x, ev.x, acc); acc = fmaf(w1r[q].y, ev.y, acc);
        acc = fmaf(w1r[q].z, ev.z, acc); acc = fmaf(w1r[q].w, ev.w, acc);
      }
      hid[wq * GT16_ELD + lane] = fmaxf(acc + b1v, 0.f);
    }
  }
  __syncthreads();
  {
    const int tg = tid & 127, wq = tg >> 3, o = tg & 7;  // per group: 16 windows x up to 8 outputs
    float y = 0.f;
    if (o < a.NOUT) {
      for (int k = 0; k < 2 * H; ++k) y = fmaf(a.w2[o * 64 + k], hid[wq * GT16_ELD + k], y);
      y += a.b2[o];
    }
    const bool live = o < a.NOUT && w0 + wq < aa.nw;
    if (a.HEAD == 0) {
      if (live) a.out[(size_t)(w0 + wq) * a.NOUT + o] = sigmoid_f(y);
    } else {
      float mx = (o < a.NOUT) ? y : -INFINITY;
      for (int d = 1; d < 8; d <<= 1) mx = fmaxf(mx, __shfl_xor(mx, d));
      const float e = (o < a.NOUT) ? expf(y - mx) : 0.f;
      float sum = e;
      for (int d = 1; d < 8; d <<= 1) sum += __shfl_xor(sum, d);
      if (live) a.out[(size_t)(w0 + wq) * a.NOUT + o] = e / sum;
    }
  }
}
#undef GT16H_RECUR
#endif  // WW_TAIL16H

#undef CF_ROUND
#undef CF_ROUND_L2
#undef GT16_PRE_ALL
#undef GT16_PRE_FIRST
#undef GT16_POS

// detect.tflite alone (reference detect_model(x), wakeword/tflite.py:228-229): one wave per row
__global__ __launch_bounds__(64) void crnn_detect_kernel(const float *enc, const float *w1, const float *b1, const float *w2,
                                                         const float *b2, float *out, int NOUT, int HEAD) {
  __shared__ float e[64], hid[64];
  const int lane = threadIdx.x, w = blockIdx.x;
  e[lane] = enc[(size_t)w * 64 + lane];
  __syncthreads();
  float acc = 0.f;
  for (int k = 0; k < 64; ++k) acc = fmaf(w1[lane * 64 + k], e[k], acc);
  hid[lane] = fmaxf(acc + b1[lane], 0.f);
  __syncthreads();
  float y = 0.f;
  if (lane < NOUT) {
    for (int k = 0; k < 64; ++k) y = fmaf(w2[lane * 64 + k], hid[k], y);
    y += b2[lane];
  }
  if (HEAD == 0) {
    if (lane < NOUT) out[(size_t)w * NOUT + lane] = sigmoid_f(y);
  } else {
    float mx = (lane < NOUT) ? y : -INFINITY;
    for (int o = 1; o < 8; o <<= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    float ex = (lane < NOUT) ? expf(y - mx) : 0.f;
    float sum = ex;
    for (int o = 1; o < 8; o <<= 1) sum += __shfl_xor(sum, o);
    if (lane < NOUT) out[(size_t)w * NOUT + lane] = ex / sum;
  }
}

int ww_k_crnn_detect(ww_ctx *ctx, const ww_model *m, const float *d_enc, int nw, float *d_out) {
  if (nw <= 0) return WW_OK;
  const ww_crnn_dev &c = m->crnn;
  ww_launch_scope scope(ctx, "crnn_detect_kernel");
  hipLaunchKernelGGL(crnn_detect_kernel, dim3(nw), dim3(64), 0, ctx->stream, d_enc, c.w1, c.b1, c.w2, c.b2, d_out, c.NOUT, c.HEAD);
  WW_HIP(ctx, hipGetLastError());
  return WW_OK;
}

// ------------------------------------------------------------------------------------------
// Generic geometry (ww_crnn_dev::generic): the older export the reference still ships
// (utils/CRNN_files/{encode,detect}_old.tflite - Conv2D 20x5 over freq x time, stride 8x2, VALID ->
// 74 steps of 3*32 features) and any other conv shape a retrained checkpoint may use.  Correctness path,
// not a tuned one: direct convolution on the vector ALU, the MFMA GEMM above on K padded to 64, one
// workgroup per window walking the recurrence with the step inputs read from global memory.
// ------------------------------------------------------------------------------------------
struct convg_args {
  const float *mel;
  win_addr wa;
  const float *wt;    // [KF*KT][C]
  const float *bias;  // [C]
  float *feat;        // [Nw][OT][FEATP]
  int n_mel, T, C, KF, KT, SF, ST, PF, PT, OF, OT, FEATP;
};

__global__ __launch_bounds__(256) void conv_generic_kernel(convg_args a) {
  extern __shared__ __align__(16) float cg_win[];  // [T][n_mel], rows >= valid zero
  const int tid = threadIdx.x, w = blockIdx.x;
  int64_t row;
  int valid;
  window_span(a.wa, w, a.T, row, valid);
  const float *src = a.mel + row * a.n_mel;
  for (int i = tid; i < a.T * a.n_mel; i += 256) cg_win[i] = i < valid * a.n_mel ? src[i] : 0.f;
  __syncthreads();
  float *dst = a.feat + (size_t)w * a.OT * a.FEATP;
  const int feat_w = a.OF * a.C;
  for (int o = tid; o < a.OT * a.FEATP; o += 256) {
    const int t = o / a.FEATP, col = o - t * a.FEATP;
    float v = 0.f;
    if (col < feat_w) {
      const int f = col / a.C, c = col - f * a.C;
      float acc = 0.f;
      for (int kf = 0; kf < a.KF; ++kf) {
        const int im = f * a.SF - a.PF + kf;
        if (im < 0 || im >= a.n_mel) continue;
        for (int kt = 0; kt < a.KT; ++kt) {
          const int it = t * a.ST - a.PT + kt;
          if (it < 0 || it >= a.T) continue;
          acc = fmaf(cg_win[it * a.n_mel + im], a.wt[(size_t)(kf * a.KT + kt) * a.C + c], acc);
        }
      }
      v = fmaxf(acc + a.bias[c], 0.f);
    }
    dst[o] = v;
  }
}

struct grug_args {
  const float *gx;   // [Nw][OT][2*3H] input projections incl. b_x
  const float *wh, *bh;
  float *seq;        // [Nw][OT][2H] (fwd | bwd) or nullptr
  float *last;       // [Nw][2H] (fwd_last | bwd_last) or nullptr
  int OT;
};

// wave 0 = forward, wave 1 = backward; the next step's three gate inputs are in flight while this step computes
__global__ __launch_bounds__(128) void gru_generic_kernel(grug_args a) {
  constexpr int H = GR_H;
  __shared__ __align__(16) float hbuf[2][2][H];
  const int tid = threadIdx.x, lane = tid & 63, dir = tid >> 6;
  const int unit = lane >> 1, half = lane & 1;
  const int w = blockIdx.x, OT = a.OT;
  gru_w g;
  gru_load_w(g, a.wh, a.bh, dir, unit, half);
  if (tid < 2 * H) hbuf[tid >> 5][0][tid & 31] = 0.f;
  __syncthreads();
  const float *gxl = a.gx + (size_t)w * OT * 6 * H + dir * 3 * H + unit;
  float h_own = 0.f;
  int t = dir ? OT - 1 : 0;
  float gz = gxl[(size_t)t * 6 * H], gr = gxl[(size_t)t * 6 * H + H], gc = gxl[(size_t)t * 6 * H + 2 * H];
  for (int s = 0; s < OT; ++s) {
    const int cur = s & 1;
    const int tn = dir ? t - 1 : t + 1;
    float nz = 0.f, nr = 0.f, nc = 0.f;
    if (s + 1 < OT) {
      nz = gxl[(size_t)tn * 6 * H]; nr = gxl[(size_t)tn * 6 * H + H]; nc = gxl[(size_t)tn * 6 * H + 2 * H];
    }
    float uz, ur, uc;
    h_own = gru_step<false>(g, &hbuf[dir][cur][0], half, gz, gr, gc, h_own, nullptr, uz, ur, uc);
    if (half == 0) {
      hbuf[dir][cur ^ 1][unit] = h_own;
      if (a.seq) a.seq[((size_t)w * OT + t) * 2 * H + dir * H + unit] = h_own;
    }
    wsync_h();
    gz = nz; gr = nr; gc = nc;
    t = tn;
  }
  if (a.last && half == 0) a.last[(size_t)w * 2 * H + dir * H + unit] = h_own;
}

static size_t crnn_generic_workspace(const ww_crnn_dev &c, int nw) {
  const size_t rows = (size_t)nw * c.OT;
  return ww_bump::need(rows * c.FEATP, 4) + ww_bump::need(rows * 6 * c.H, 4) + ww_bump::need(rows * 2 * c.H, 4) +
         ww_bump::need((size_t)nw * 2 * c.H, 4) + 1024;
}

static void launch_gemm(ww_ctx *ctx, const char *name, const float *A, const float *W, const float *bias, float *C, int M, int N, int K) {
  const int n_tiles = (N + GB_N - 1) / GB_N, m_tiles = (M + GB_M - 1) / GB_M;
  gemm_args g = {A, W, bias, C, M, N, K, n_tiles};
  ww_launch_scope scope(ctx, name);
  // ids: 8 XCDs x (m-tiles of that residue, rounded up) x n-tiles; ids whose m-tile is past the end exit at once
  hipLaunchKernelGGL(gemm_nt_kernel, dim3(8 * ((m_tiles + 7) / 8) * n_tiles), dim3(512), 0, ctx->stream, g);
}

static int crnn_forward_generic(ww_ctx *ctx, const ww_model *m, const win_addr &wa, const float *d_mel, int nw, void *ws,
                                float *d_out, float *d_enc) {
  const ww_crnn_dev &c = m->crnn;
  const size_t rows = (size_t)nw * c.OT;
  ww_bump b(ws, ~size_t(0));
  float *feat = b.take<float>(rows * c.FEATP), *gx = b.take<float>(rows * 6 * c.H), *seq = b.take<float>(rows * 2 * c.H);
  float *enc = b.take<float>((size_t)nw * 2 * c.H);
  if (d_enc) enc = d_enc;
  {
    convg_args a = {d_mel, wa, c.conv_wt, c.conv_b, feat, c.n_mel, c.T, c.C, c.KF, c.KT, c.SF, c.ST, c.PF, c.PT, c.OF, c.OT, c.FEATP};
    ww_launch_scope scope(ctx, "conv_generic_kernel");
    hipLaunchKernelGGL(conv_generic_kernel, dim3(nw), dim3(256), (size_t)c.T * c.n_mel * sizeof(float), ctx->stream, a);
  }
  launch_gemm(ctx, "gemm_nt_kernel<gru1,generic>", feat, c.wx1p, c.bx1, gx, (int)rows, 6 * c.H, c.FEATP);
  {
    grug_args a = {gx, c.wh1, c.bh1, seq, nullptr, c.OT};
    ww_launch_scope scope(ctx, "gru_generic_kernel<seq>");
    hipLaunchKernelGGL(gru_generic_kernel, dim3(nw), dim3(128), 0, ctx->stream, a);
  }
  launch_gemm(ctx, "gemm_nt_kernel<gru2,generic>", seq, c.wx2, c.bx2, gx, (int)rows, 6 * c.H, 2 * c.H);
  {
    grug_args a = {gx, c.wh2, c.bh2, nullptr, enc, c.OT};
    ww_launch_scope scope(ctx, "gru_generic_kernel<last>");
    hipLaunchKernelGGL(gru_generic_kernel, dim3(nw), dim3(128), 0, ctx->stream, a);
  }
  WW_HIP(ctx, hipGetLastError());
  return ww_k_crnn_detect(ctx, m, enc, nw, d_out);
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
// Above this many windows per launch the recurrences move to gru_tail_kernel (one more kernel boundary and 14.6 KB per
// window through memory buy a CU-filling mix; below it one workgroup per window end to end is the shorter path).
// (ww_model_set_option(WW_OPT_CRNN_SPLIT_AT): 0 = always fused.)
static int crnn_split_threshold(const ww_model *m) { return m->opt_split_at; }

// More than half a CU's LDS: a workgroup that asks for this much has the CU's LDS to itself (development probes only: asked for
// on crnn_fused_kernel / crnn_stream_kernel<tick> launches of <= 256 workgroups it changed nothing - 33.92 vs 33.90 us, tick
// p50 37.3 vs 37.0 us: those grids already sit one workgroup per CU; profiles/r06/tail16_probes.txt).
#define WW_LDS_ONE_PER_CU (82 * 1024)

// Every CRNN kernel that asks for more than the default 64 KB of dynamic LDS.  The attribute is per device, so it is set
// for the device of every new context (ww_ctx_create, under its device scope) instead of once per process.
int ww_k_crnn_init_device(ww_ctx *ctx) {
  WW_HIP(ctx, hipFuncSetAttribute((const void *)crnn_stream_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, CF_SMEM_BYTES));
  WW_HIP(ctx, hipFuncSetAttribute((const void *)crnn_stream_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, CF_SMEM_BYTES));
  WW_HIP(ctx, hipFuncSetAttribute((const void *)crnn_stream_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, CF_SMEM_BYTES));
  WW_HIP(ctx, hipFuncSetAttribute((const void *)crnn_rows_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, CF_SMEM_BYTES));
  WW_HIP(ctx, hipFuncSetAttribute((const void *)gru_tail16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, WW_LDS_ONE_PER_CU));
  WW_HIP(ctx, hipFuncSetAttribute((const void *)crnn_fused_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, CF_FUSED_SMEM_BYTES));
  WW_HIP(ctx, hipFuncSetAttribute((const void *)crnn_fused_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, CF_FUSED_SMEM_BYTES));
  WW_HIP(ctx, hipFuncSetAttribute((const void *)crnn_fused_bf16_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, CFB_SMEM_BYTES));
  WW_HIP(ctx, hipFuncSetAttribute((const void *)crnn_fused_bf16_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, CFB_SMEM_BYTES));
  return WW_OK;
}

// streaming form (crnn_stream_kernel): standard geometry, fp32 contractions
bool ww_crnn_stream_capable(const ww_model *m) { return m->kind == WW_KIND_CRNN && !m->crnn.generic; }

// ONE launch per tick (crnn_stream_kernel<1 | 2>): 2 S workgroups, the posteriors as tags only
int ww_k_crnn_tick(ww_ctx *ctx, const ww_model *m, const ww_tick_fe &fe, int precise, float *d_gxc, const ww_tick_tag &tag) {
  const ww_crnn_dev &c = m->crnn;
  const ww_filter_dev &f = m->filt;
  if (c.generic || f.n_mel != CV_NMEL || fe.hop != 160)
    return ww_fail(ctx, WW_EINVAL, "one-launch streaming tick: standard conv geometry, 40 mel bands and hop 160 only");
  if (!tag.slots || fe.S <= 0) return ww_fail(ctx, WW_EINVAL, "one-launch streaming tick: no tag slots / no streams");
  stream_args sa = {};
  sa.f = {fe.hist, {nullptr, nullptr, 0, 0, 0, (int64_t)fe.S * fe.HR}, c.conv_w, c.conv_b, c.wx1s, c.bx1, c.wh1, c.bh1, c.wx2s, c.bx2, c.wh2, c.bh2,
          c.w1, c.b1, c.w2, c.b2, nullptr, nullptr, c.T, c.NOUT, c.HEAD, nullptr, nullptr, c.cwb, c.wx1b};
  sa.f.tag = tag;
  sa.gxc = d_gxc;
  sa.fe = fe;
  sa.start = f.start; sa.wpad = f.wpad; sa.bias = f.bias; sa.n_mel = f.n_mel;
  sa.floor_v = f.floor_v; sa.log_off = f.log_off; sa.scale = f.scale;
  sa.hann = f.hann; sa.tw256 = f.tw256; sa.tw512 = f.tw512;
  static const bool want_stamps = getenv("WWHIP_CF_STAMPS") != nullptr;  // development: phase timeline (the call then waits for its kernel)
  const int nwg = 2 * fe.S;
  if (want_stamps) {
    WW_HIP(ctx, hipMalloc((void **)&sa.f.stamps, (size_t)nwg * 64 * sizeof(long long)));
    WW_HIP(ctx, hipMemsetAsync(sa.f.stamps, 0, (size_t)nwg * 64 * sizeof(long long), ctx->stream));
    sa.tstamps = sa.f.stamps + (size_t)nwg * 40;
  }
  {
    ww_launch_scope scope(ctx, "crnn_stream_kernel<tick>");
    if (precise) hipLaunchKernelGGL(crnn_stream_kernel<2>, dim3(nwg), dim3(CF_THREADS), CF_SMEM_BYTES, ctx->stream, sa);
    else hipLaunchKernelGGL(crnn_stream_kernel<1>, dim3(nwg), dim3(CF_THREADS), CF_SMEM_BYTES, ctx->stream, sa);
  }
  WW_HIP(ctx, hipGetLastError());
  if (want_stamps) {
    std::vector<long long> h((size_t)nwg * 64);
    WW_HIP(ctx, hipStreamSynchronize(ctx->stream));
    WW_HIP(ctx, hipMemcpy(h.data(), sa.f.stamps, h.size() * sizeof(long long), hipMemcpyDeviceToHost));
    WW_HIP(ctx, hipFree(sa.f.stamps));
    for (int wv : {0, 3}) {  // wave 0 transforms a frame, wave 3 only stages: mean cycles since the workgroup's entry
      double sum[16] = {0};
      int cnt = 0;
      for (int b = 0; b < nwg; ++b) {
        const long long *c = &h[((size_t)b * 4 + wv) * 10], *t = &h[(size_t)nwg * 40 + ((size_t)b * 4 + wv) * 6];
        if (!c[0] || !c[9]) continue;  // (a workgroup without a window)
        for (int i = 0; i < 6; ++i) sum[i] += (double)(t[i] - c[0]);
        for (int i = 1; i < 10; ++i) sum[5 + i] += (double)(c[i] - c[0]);
        ++cnt;
      }
      fprintf(stderr, "crnn_stream_kernel<tick>, %d window workgroups, wave %d, mean cycles since entry: ctl | pre-barrier barrier normalised "
              "frames staged | image conv conv-barrier proj proj-barrier D E F G:", cnt, wv);
      for (int i = 0; i < 15; ++i) fprintf(stderr, "%s %.0f", i == 1 || i == 6 ? " |" : "", cnt ? sum[i] / cnt : 0.0);
      fprintf(stderr, "\n");
    }
  }
  return WW_OK;
}

int ww_k_crnn_stream_forward(ww_ctx *ctx, const ww_model *m, const float *d_hist, int64_t hist_rows, const int64_t *d_win_row,
                             const int32_t *d_win_valid, const int32_t *d_win_aux, float *d_gxc, int nw, float *d_out,
                             const ww_tick_tag *tag) {
  if (nw <= 0) return WW_OK;
  const ww_crnn_dev &c = m->crnn;
  if (c.generic) return ww_fail(ctx, WW_EINVAL, "streaming CRNN kernel: standard conv geometry only");
  if (!d_win_row || !d_win_valid || !d_win_aux) return ww_fail(ctx, WW_EINVAL, "streaming CRNN kernel: NULL window table");
  stream_args sa = {};
  sa.f = {d_hist, {d_win_row, d_win_valid, 0, 0, 0, hist_rows}, c.conv_w, c.conv_b, c.wx1s, c.bx1, c.wh1, c.bh1, c.wx2s, c.bx2, c.wh2, c.bh2,
          c.w1, c.b1, c.w2, c.b2, nullptr, d_out, c.T, c.NOUT, c.HEAD, nullptr, nullptr, c.cwb, c.wx1b};
  sa.aux = d_win_aux;
  sa.gxc = d_gxc;
  if (tag) sa.f.tag = *tag;
  static const bool want_stamps = getenv("WWHIP_CF_STAMPS") != nullptr;  // development: phase timeline (as ww_k_crnn_forward)
  if (want_stamps) {
    WW_HIP(ctx, hipMalloc((void **)&sa.f.stamps, (size_t)nw * 40 * sizeof(long long)));
    WW_HIP(ctx, hipMemsetAsync(sa.f.stamps, 0, (size_t)nw * 40 * sizeof(long long), ctx->stream));
  }
  {
    ww_launch_scope scope(ctx, "crnn_stream_kernel");
    hipLaunchKernelGGL(crnn_stream_kernel<0>, dim3(nw), dim3(CF_THREADS), CF_SMEM_BYTES, ctx->stream, sa);
  }
  WW_HIP(ctx, hipGetLastError());
  if (want_stamps) {
    std::vector<long long> h((size_t)nw * 40);
    WW_HIP(ctx, hipStreamSynchronize(ctx->stream));
    WW_HIP(ctx, hipMemcpy(h.data(), sa.f.stamps, h.size() * sizeof(long long), hipMemcpyDeviceToHost));
    WW_HIP(ctx, hipFree(sa.f.stamps));
    double sum[10] = {0};
    for (int b = 0; b < nw; ++b)
      for (int i = 0; i < 10; ++i) sum[i] += (double)(h[((size_t)b * 4 + 0) * 10 + i] - h[((size_t)b * 4 + 0) * 10]);
    fprintf(stderr, "crnn_stream_kernel, %d windows, wave 0, mean cycles since entry:", nw);
    for (int i = 0; i < 10; ++i) fprintf(stderr, " %.0f", sum[i] / nw);
    fprintf(stderr, "\n");
  }
  return WW_OK;
}

// Everything behind the layer-1 projections: sixteen windows per workgroup with the recurrent products on the matrix pipe
// (gru_tail16_kernel), or one window per workgroup on the vector ALU (gru_tail_kernel).  The matrix form is a 38-step serial
// chain of ~4 k cycles per step on two waves: it needs >= ~600 workgroups to beat the vector form (tools/tail_sweep.py:
// front + tail per launch 743 vs 745 us at 9,216 windows, 1,204 vs 1,309 at 16,384, but 226 vs 188 at 2,048), so
// WW_OPT_CRNN_TAIL_MFMA = 1 (default) takes it from WW_TAIL16_MIN windows per launch on; 2 = always, 0 = never.
#define WW_TAIL16_MIN 9216
static size_t tail_seq_bytes(int nw) { return ww_bump::need((size_t)((((nw + 15) / 16) + 1) & ~1) * CV_OT * 16 * 2 * GR_H, 4); }  // (an even number of groups: gru_tail16h_kernel takes two per workgroup)
static void launch_tail(ww_ctx *ctx, const ww_model *m, tail_args t, int nw, float *seq) {
  t.wx2 = m->crnn.wx2;
#if WW_TAIL16H
  if (m->opt_tail_mfma == 3 && t.gx1) {  // (round 6 probe) hoisted projection, two groups per workgroup: launches that own their gx1 rows
    tail16h_args ah = {t, seq, nullptr, nw};
    ww_launch_scope scope(ctx, "gru_tail16h_kernel");
    hipLaunchKernelGGL(gru_tail16h_kernel, dim3((unsigned)((nw + 31) / 32)), dim3(256), GT16H_SMEM_BYTES, ctx->stream, ah);
    return;
  }
#endif
  if (m->opt_tail_mfma >= 2 || (m->opt_tail_mfma == 1 && nw >= WW_TAIL16_MIN)) {
    tail16_args a16 = {t, seq, nw};
    ww_launch_scope scope(ctx, "gru_tail16_kernel");
    // Development (round 6 occupancy probe, tools/tail16_occ.py -> profiles/r06/tail16_probes.txt): WWHIP_TAIL16_DEEP = n asks
    // for so much LDS that only n workgroups fit on a CU - the shipped instruction stream at 1 / 2 / 3 / 4 workgroups per CU.
    const int n_wg = (nw + 15) / 16;
    static const int deep = getenv("WWHIP_TAIL16_DEEP") ? atoi(getenv("WWHIP_TAIL16_DEEP")) : 4;
    const size_t lds = deep >= 4 ? (size_t)GT16_SMEM_BYTES : deep == 3 ? (size_t)44 * 1024 : deep == 2 ? (size_t)56 * 1024 : (size_t)WW_LDS_ONE_PER_CU;
    hipLaunchKernelGGL(gru_tail16_kernel, dim3((unsigned)n_wg), dim3(128), lds, ctx->stream, a16);
  } else {
    ww_launch_scope scope(ctx, "gru_tail_kernel");
    hipLaunchKernelGGL(gru_tail_kernel, dim3((unsigned)nw), dim3(128), 0, ctx->stream, t);
  }
}

// Regular sliding windows (hop <= 8 over one mel sequence, every window complete) take crnn_rows_kernel + gru_tail_kernel from
// this many windows on (ww_model_set_option(WW_OPT_CRNN_SLIDE_MIN): 0 = never).
static int crnn_slide_min(const ww_model *m) { return m->opt_slide_min > 0 ? m->opt_slide_min : 0x7fffffff; }
static int gcd8(int hop) { return hop % 8 == 0 ? 8 : hop % 4 == 0 ? 4 : hop % 2 == 0 ? 2 : 1; }

// Several mel sequences in one buffer (the clips of a test set, wwhip/evaluate.py), each slid over with the same hop:
// crnn_rows_kernel by tile descriptors (no tile straddles two sequences), gru_tail_kernel with each window's first
// interior field given explicitly.  seg_row0 / seg_nw are HOST arrays; windows are numbered sequence by sequence.
bool ww_crnn_segments_capable(const ww_model *m, int hop) {
  return m->kind == WW_KIND_CRNN && !m->crnn.generic && hop >= 1 && hop <= 8 &&
         crnn_slide_min(m) != 0x7fffffff;
}

int ww_k_crnn_segments_forward(ww_ctx *ctx, const ww_model *m, const float *d_mel, int64_t mel_rows, const int64_t *seg_row0,
                               const int32_t *seg_nw, int n_seg, int hop, float *d_out) {
  const ww_crnn_dev &c = m->crnn;
  const int g = gcd8(hop);
  // groups of whole sequences of at most ~WW_SEG_GROUP windows bound the workspace
  constexpr int64_t WW_SEG_GROUP = 32768;
  std::vector<rows_tile> tiles;
  std::vector<int64_t> i0;
  int64_t w_done = 0;
  for (int s0 = 0; s0 < n_seg;) {
    tiles.clear();
    i0.clear();
    int64_t nI = 0, nW = 0;
    int s1 = s0;
    for (; s1 < n_seg && (s1 == s0 || nW + seg_nw[s1] <= WW_SEG_GROUP); ++s1) {
      const int nw = seg_nw[s1];
      if (nw < 0) return ww_fail(ctx, WW_EINVAL, "negative window count in sequence %d", s1);
      if (nw == 0) continue;
      const int64_t r0 = seg_row0[s1];
      if (r0 < 0 || r0 + (int64_t)(nw - 1) * hop + c.T > mel_rows)
        return ww_fail(ctx, WW_EINVAL, "sequence %d: windows leave the mel buffer", s1);
      const int64_t n_int = ((int64_t)(nw - 1) * hop + 128) / g + 1;
      for (int64_t p0 = 0; p0 < n_int; p0 += 16)
        tiles.push_back({r0 + 2 + (int64_t)g * p0, nI + p0, g, (int32_t)(n_int - p0 < 16 ? n_int - p0 : 16), 0, 0});
      for (int p0 = 0; p0 < nw; p0 += 16) {
        const int32_t cnt = nw - p0 < 16 ? nw - p0 : 16;
        tiles.push_back({r0 - c.PT + (int64_t)hop * p0, nW + p0, hop, cnt, 1, 0});
        tiles.push_back({r0 + (int64_t)(c.OT - 1) * c.ST - c.PT + (int64_t)hop * p0, nW + p0, hop, cnt, 2, 0});
      }
      for (int k = 0; k < nw; ++k) i0.push_back(nI + (int64_t)k * hop / g);
      nI += n_int;
      nW += nw;
    }
    if (nW > 0) {
      const size_t b_tiles = ww_bump::need(tiles.size() * sizeof(rows_tile), 1), b_i0 = ww_bump::need(i0.size() * 8, 1);
      const size_t b_rows = ww_bump::need((size_t)(nI + 2 * nW) * 6 * c.H, 4);
      int rc = ww_ensure(ctx, ctx->dev, b_tiles + b_i0 + b_rows + tail_seq_bytes((int)nW) + 8192, false);
      if (rc) return rc;
      ww_bump b(ctx->dev.ptr, ctx->dev.cap);
      rows_tile *d_tiles = (rows_tile *)b.take<char>(tiles.size() * sizeof(rows_tile));
      int64_t *d_i0 = b.take<int64_t>(i0.size());
      float *gI = b.take<float>((size_t)nI * 6 * c.H), *gL = b.take<float>((size_t)nW * 6 * c.H), *gR = b.take<float>((size_t)nW * 6 * c.H);
      float *seq = (float *)b.take<char>(tail_seq_bytes((int)nW));
      // the descriptors leave through one of the context's two page-locked buffers: the copies are asynchronous and the call
      // goes on to build the next group (or returns) while this group's kernels run
      const int slot = (int)(ctx->desc_k++ & 1);
      if (!ctx->desc_ev[slot]) WW_HIP(ctx, hipEventCreateWithFlags(&ctx->desc_ev[slot], hipEventDisableTiming));
      if (ctx->desc_busy[slot]) {
        WW_HIP(ctx, hipEventSynchronize(ctx->desc_ev[slot]));
        ctx->desc_busy[slot] = false;
      }
      const size_t n_t = tiles.size() * sizeof(rows_tile), n_i = i0.size() * 8, o_i = (n_t + 63) & ~(size_t)63;
      if ((rc = ww_ensure(ctx, ctx->desc_pin[slot], o_i + n_i, true))) return rc;
      char *hp = (char *)ctx->desc_pin[slot].ptr;
      memcpy(hp, tiles.data(), n_t);
      memcpy(hp + o_i, i0.data(), n_i);
      WW_HIP(ctx, hipMemcpyAsync(d_tiles, hp, n_t, hipMemcpyHostToDevice, ctx->stream));
      WW_HIP(ctx, hipMemcpyAsync(d_i0, hp + o_i, n_i, hipMemcpyHostToDevice, ctx->stream));
      WW_HIP(ctx, hipEventRecord(ctx->desc_ev[slot], ctx->stream));
      ctx->desc_busy[slot] = true;
      rows_args r = {};
      r.mel = d_mel; r.mel_rows = mel_rows;
      r.w4[0] = c.conv_w; r.w4[1] = c.conv_wL; r.w4[2] = c.conv_wR;
      r.cbias = c.conv_b; r.wx1s = c.wx1s; r.bx1 = c.bx1;
      r.out[0] = gI; r.out[1] = gL; r.out[2] = gR;
      r.desc = d_tiles;
      {
        ww_launch_scope scope(ctx, "crnn_rows_kernel");
        hipLaunchKernelGGL(crnn_rows_kernel, dim3((unsigned)tiles.size()), dim3(CF_THREADS), CF_SMEM_BYTES, ctx->stream, r);
      }
      tail_args t = {nullptr, c.wh1, c.bh1, c.wx2s, c.bx2, c.wh2, c.bh2, c.w1, c.b1, c.w2, c.b2, nullptr,
                     d_out + (size_t)w_done * c.NOUT, c.NOUT, c.HEAD, gI, gL, gR, hop / g, 8 / g, d_i0, nullptr};
      launch_tail(ctx, m, t, (int)nW, seq);
      WW_HIP(ctx, hipGetLastError());
      w_done += nW;
    }
    s0 = s1;
  }
  return WW_OK;
}

size_t ww_crnn_workspace(const ww_model *m, int nw, bool sliding) {
  const ww_crnn_dev &c = m->crnn;
  if (c.generic) return crnn_generic_workspace(c, nw);
  const int thr = crnn_split_threshold(m);
  // sliding form: (fields + 2 edge rows per window) x 192 floats; fields <= 7 nw + 130 (hop 7), i.e. never more than the
  // 19 rows per window of the front/tail form
  size_t need = 1024;  // crnn_fused_kernel keeps every intermediate in LDS
  if (sliding && nw >= crnn_slide_min(m)) need = ww_bump::need(((size_t)9 * nw + 160) * 6 * c.H, 4) + tail_seq_bytes(nw) + 2048;
  if (thr > 0 && nw > thr) {
    const size_t split = ww_bump::need((size_t)nw * c.OT * 6 * c.H, 4) + tail_seq_bytes(nw) + 2048;
    need = split > need ? split : need;
  }
  return need;
}

// the one-kernel forms write a tick's tags (cf_phases_d_to_g); the front + tail forms and the generic path do not
bool ww_crnn_forward_tags(const ww_model *m, int nw) {
  return !m->crnn.generic && !(crnn_split_threshold(m) > 0 && nw > crnn_split_threshold(m));
}

int ww_k_crnn_forward(ww_ctx *ctx, const ww_model *m, const float *d_mel, int64_t mel_rows, const int64_t *d_win_row,
                      const int32_t *d_win_valid, int64_t row0, int hop, int valid_const, int nw, void *ws, size_t ws_bytes,
                      float *d_out, float *d_enc, const ww_tick_tag *tag) {
  if (nw <= 0) return WW_OK;
  const ww_crnn_dev &c = m->crnn;
  win_addr wa = {d_win_row, d_win_valid, row0, hop, valid_const, mel_rows};
  // the scratch this launch takes under the model's CURRENT options (ww_model_set_option may have moved the thresholds
  // since the caller sized ws): never past the end of what was handed in
  const bool slide_form = !c.generic && !d_win_row && !d_win_valid && valid_const >= c.T && hop >= 1 && hop <= 8 && nw >= crnn_slide_min(m) &&
                          row0 >= 0 && row0 + (int64_t)(nw - 1) * hop + c.T <= mel_rows;
  {
    size_t need = 0;
    if (c.generic) need = crnn_generic_workspace(c, nw);
    else if (slide_form) need = ww_bump::need(((size_t)9 * nw + 160) * 6 * c.H, 4) + tail_seq_bytes(nw) + 2048;
    else if (crnn_split_threshold(m) > 0 && nw > crnn_split_threshold(m)) need = ww_bump::need((size_t)nw * c.OT * 6 * c.H, 4) + tail_seq_bytes(nw) + 2048;
    if (need > ws_bytes)
      return ww_fail(ctx, WW_EINVAL, "CRNN launch of %d windows needs %zu bytes of scratch, the caller reserved %zu (options changed "
                     "after the buffer was sized?)", nw, need, ws_bytes);
  }
  if (c.generic && tag && tag->slots) return ww_fail(ctx, WW_EINVAL, "tick tags: the generic CRNN path does not write them");
  if (c.generic) return crnn_forward_generic(ctx, m, wa, d_mel, nw, ws, d_out, d_enc);
  fused_args a = {d_mel, wa, c.conv_w, c.conv_b, c.wx1s, c.bx1, c.wh1, c.bh1, c.wx2s, c.bx2, c.wh2, c.bh2,
                  c.w1, c.b1, c.w2, c.b2, d_enc, d_out, c.T, c.NOUT, c.HEAD, nullptr, nullptr, c.cwb, c.wx1b};
  const bool bf16 = m->precision == WW_PRECISION_BF16X3;
  // (also in split-bf16 mode: the mode permits bf16 products, and computing a seventh of them in fp32 is both faster and closer)
  if (slide_form) {
    // windows sliding over one sequence: 1 + 2 positions per window instead of 19 (crnn_rows_kernel)
    const int g = gcd8(hop);
    const int64_t n_int = ((int64_t)(nw - 1) * hop + 128) / g + 1;
    ww_bump b(ws, ws_bytes);
    float *gI = b.take<float>((size_t)n_int * 6 * c.H), *gL = b.take<float>((size_t)nw * 6 * c.H), *gR = b.take<float>((size_t)nw * 6 * c.H);
    float *seq = (float *)b.take<char>(tail_seq_bytes(nw));
    rows_args r = {};
    r.mel = d_mel; r.mel_rows = mel_rows;
    r.w4[0] = c.conv_w; r.w4[1] = c.conv_wL; r.w4[2] = c.conv_wR;
    r.cbias = c.conv_b; r.wx1s = c.wx1s; r.bx1 = c.bx1;
    r.out[0] = gI; r.out[1] = gL; r.out[2] = gR;
    r.start[0] = row0 + 2; r.start[1] = row0 - c.PT; r.start[2] = row0 + (int64_t)(c.OT - 1) * c.ST - c.PT;
    r.stride[0] = g; r.stride[1] = hop; r.stride[2] = hop;
    r.count[0] = (int)n_int; r.count[1] = nw; r.count[2] = nw;
    for (int k = 0; k < 3; ++k) r.tiles[k] = (r.count[k] + 15) / 16;
    {
      ww_launch_scope scope(ctx, "crnn_rows_kernel");
      hipLaunchKernelGGL(crnn_rows_kernel, dim3(r.tiles[0] + r.tiles[1] + r.tiles[2]), dim3(CF_THREADS), CF_SMEM_BYTES, ctx->stream, r);
    }
    tail_args t = {nullptr, c.wh1, c.bh1, c.wx2s, c.bx2, c.wh2, c.bh2, c.w1, c.b1, c.w2, c.b2, d_enc, d_out, c.NOUT, c.HEAD,
                   gI, gL, gR, hop / g, 8 / g, nullptr, nullptr};
    launch_tail(ctx, m, t, nw, seq);
    WW_HIP(ctx, hipGetLastError());
    return WW_OK;
  }
  const int thr = crnn_split_threshold(m);
  if (tag && tag->slots) {
    if (slide_form || (thr > 0 && nw > thr)) return ww_fail(ctx, WW_EINVAL, "tick tags: this launch form does not write them");
    a.tag = *tag;
  }
  if (thr > 0 && nw > thr) {
    ww_bump b(ws, ws_bytes);
    a.gx_out = b.take<float>((size_t)nw * c.OT * 6 * c.H);
    float *seq = (float *)b.take<char>(tail_seq_bytes(nw));
    {
      ww_launch_scope scope(ctx, bf16 ? "crnn_fused_kernel<front,bf16x3>" : "crnn_fused_kernel<front>");
      if (bf16) hipLaunchKernelGGL(crnn_fused_bf16_kernel<true>, dim3(nw), dim3(CF_THREADS), CFB_SMEM_BYTES, ctx->stream, a);
      else hipLaunchKernelGGL(crnn_fused_kernel<true>, dim3(nw), dim3(CF_THREADS), CF_FUSED_SMEM_BYTES, ctx->stream, a);
    }
    tail_args t = {a.gx_out, c.wh1, c.bh1, c.wx2s, c.bx2, c.wh2, c.bh2, c.w1, c.b1, c.w2, c.b2, d_enc, d_out, c.NOUT, c.HEAD, nullptr, nullptr, nullptr, 0, 0, nullptr, nullptr};
    launch_tail(ctx, m, t, nw, seq);
    WW_HIP(ctx, hipGetLastError());
    return WW_OK;
  }
  // development: WWHIP_CF_STAMPS=1 prints the s_memtime stamps of the phase boundaries (first, middle and last workgroup)
  static const bool want_stamps = getenv("WWHIP_CF_STAMPS") != nullptr;
  if (want_stamps) {
    WW_HIP(ctx, hipMalloc((void **)&a.stamps, (size_t)nw * 40 * sizeof(long long)));
    WW_HIP(ctx, hipMemsetAsync(a.stamps, 0, (size_t)nw * 40 * sizeof(long long), ctx->stream));
  }
  {
    ww_launch_scope scope(ctx, bf16 ? "crnn_fused_kernel<bf16x3>" : "crnn_fused_kernel");
    if (bf16) hipLaunchKernelGGL(crnn_fused_bf16_kernel<false>, dim3(nw), dim3(CF_THREADS), CFB_SMEM_BYTES, ctx->stream, a);
    else hipLaunchKernelGGL(crnn_fused_kernel<false>, dim3(nw), dim3(CF_THREADS), CF_FUSED_SMEM_BYTES, ctx->stream, a);
  }
  WW_HIP(ctx, hipGetLastError());
  if (want_stamps) {
    std::vector<long long> h((size_t)nw * 40);
    WW_HIP(ctx, hipStreamSynchronize(ctx->stream));
    WW_HIP(ctx, hipMemcpy(h.data(), a.stamps, h.size() * sizeof(long long), hipMemcpyDeviceToHost));
    WW_HIP(ctx, hipFree(a.stamps));
    for (int b : {0, nw / 2, nw - 1}) {
      for (int wv = 0; wv < 4; ++wv) {
        const long long *t = &h[((size_t)b * 4 + wv) * 10];
        fprintf(stderr, "block %d wave %d:", b, wv);
        for (int i = 0; i < 10; ++i) fprintf(stderr, " %lld", t[i] - t[0]);  // clocks of different XCDs are not comparable
        fprintf(stderr, "\n");
      }
    }
  }
  return WW_OK;
}
