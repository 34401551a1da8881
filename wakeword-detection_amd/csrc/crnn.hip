// CRNN encode + detect for gfx950 (fp32, MFMA for the dense contractions).
//
// Replaces encode.tflite + detect.tflite of the reference CRNN
// (tf_lite_models/CRNN; architecture wwdetect/CRNN/model.py:21-56; call sites
// spokestack/wakeword/tflite.py:193-231, utils/evaluate_models.py:76-80).
//
//   conv5x20_kernel   Conv2D(1->32, 5x20, stride 2x8, SAME, ReLU) as an implicit GEMM on
//                     v_mfma_f32_16x16x4_f32: one workgroup per window, the 151x40 window is
//                     staged once (transposed, zero-bordered) in LDS, conv weights live in
//                     registers, output is written in the [t][f*32+c] order of the
//                     Permute+Reshape (model.py:37-39).
//   gemm_nt_kernel    C[M][N] = A[M][K] * W[N][K]^T + b, 64x64x32 tiles, fp32 MFMA, register
//                     prefetch + double-buffered LDS.  Used for the GRU input projections of
//                     both directions at once (N = 2*3H = 192).
//   gru_kernel        the 19-step recurrence; one wavefront per (window, direction), W_h held
//                     in registers (48 VGPRs per lane, K split over the two lane halves),
//                     h exchanged through LDS.  Layer 2 keeps only the last state and the
//                     backward wave's partner computes the detect head in the same launch.
#include "common.h"

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
  row = wa.row ? wa.row[w] : wa.row0 + (int64_t)w * wa.hop;
  valid = wa.valid ? wa.valid[w] : wa.valid_const;
  if (valid > T) valid = T;
  if (row + valid > wa.mel_rows) valid = (int)(wa.mel_rows - row);
  if (valid < 0) valid = 0;
}

__device__ __forceinline__ float sigmoid_f(float x) { return 1.0f / (1.0f + expf(-x)); }

// ------------------------------------------------------------------------------------------
// conv
// ------------------------------------------------------------------------------------------
#define CV_ROWS 44   // mel index + PF, 0..43
#define CV_LDT 164   // time index + PT, 0..163 (multiple of 4; 164 % 32 == 4)
#define CV_KB 7      // K = 5*20 = 100 padded to 112 = 7 * 16

struct conv_args {
  const float *mel;
  win_addr wa;
  const float *w4;    // [K/4][32][4]  (k-quad major), K padded to 112 with zeros
  const float *bias;  // [32]
  float *feat;        // [Nw][OT][OF*32]
  int n_mel, T, KF, KT, SF, ST, PF, PT, OF, OT;
};

__global__ __launch_bounds__(256) void conv5x20_kernel(conv_args a) {
  __shared__ __align__(16) float img[CV_ROWS * CV_LDT];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int w = blockIdx.x;
  int64_t row;
  int valid;
  window_span(a.wa, w, a.T, row, valid);

  for (int i = tid; i < CV_ROWS * CV_LDT / 4; i += 256) ((float4 *)img)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  __syncthreads();
  {
    const float *src = a.mel + row * a.n_mel;
    const int n = valid * a.n_mel;
    for (int i = tid; i < n; i += 256) {
      int it = i / a.n_mel, im = i - it * a.n_mel;
      img[(im + a.PF) * CV_LDT + it + a.PT] = src[i];
    }
  }
  // conv weights for this lane: B operand of mfma 16x16x4: lane (j = lane&15, kk = lane>>4)
  const int j = lane & 15, kk = lane >> 4;
  float4 wreg[CV_KB][2];
#pragma unroll
  for (int kb = 0; kb < CV_KB; ++kb)
#pragma unroll
    for (int n = 0; n < 2; ++n) wreg[kb][n] = *(const float4 *)(a.w4 + ((size_t)(kb * 4 + kk) * 32 + n * 16 + j) * 4);
  const float b0 = a.bias[j], b1 = a.bias[16 + j];
  __syncthreads();

  const int M = a.OT * a.OF;
  const int n_mt = (M + 15) / 16;
  float *dst = a.feat + (size_t)w * M * 32;
  for (int mt = wave; mt < n_mt; mt += 4) {
    int m = mt * 16 + j;  // A operand row for this lane (i = lane & 15)
    if (m >= M) m = M - 1;
    const int t = m / a.OF, f = m - t * a.OF;
    const float *abase = img + (f * a.SF) * CV_LDT + t * a.ST;
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kb = 0; kb < CV_KB; ++kb) {
      const int k4 = kb * 16 + kk * 4;
      const int kf = k4 / a.KT, kt = k4 - kf * a.KT;
      const float4 av = *(const float4 *)(abase + kf * CV_LDT + kt);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, wreg[kb][0].x, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, wreg[kb][1].x, acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, wreg[kb][0].y, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, wreg[kb][1].y, acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, wreg[kb][0].z, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, wreg[kb][1].z, acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, wreg[kb][0].w, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, wreg[kb][1].w, acc1, 0, 0, 0);
    }
    // D layout: col = lane & 15, row = (lane >> 4) * 4 + r
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int mo = mt * 16 + kk * 4 + r;
      if (mo < M) {
        dst[(size_t)mo * 32 + j] = fmaxf(acc0[r] + b0, 0.f);
        dst[(size_t)mo * 32 + 16 + j] = fmaxf(acc1[r] + b1, 0.f);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// GEMM  C[M][N] = A[M][K] * W[N][K]^T + bias[N]
// ------------------------------------------------------------------------------------------
#define GB_M 64
#define GB_N 64
#define GB_K 32
#define GB_LD 36

struct gemm_args {
  const float *A;
  const float *W;
  const float *bias;
  float *C;
  int M, N, K;
};

__global__ __launch_bounds__(256) void gemm_nt_kernel(gemm_args g) {
  __shared__ __align__(16) float As[2][GB_M * GB_LD];
  __shared__ __align__(16) float Bs[2][GB_N * GB_LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int m0 = blockIdx.x * GB_M, n0 = blockIdx.y * GB_N;
  const int lrow = tid >> 3, lc4 = tid & 7;  // loader: rows lrow, lrow+32; float4 column lc4
  const int i16 = lane & 15, kk = lane >> 4;

  float4 pa[2], pb[2];
  auto gload = [&](int k0) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int r = lrow + 32 * h;
      const int gm = m0 + r;
      pa[h] = gm < g.M ? *(const float4 *)(g.A + (size_t)gm * g.K + k0 + lc4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
      const int gn = n0 + r;
      pb[h] = gn < g.N ? *(const float4 *)(g.W + (size_t)gn * g.K + k0 + lc4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto sstore = [&](int buf) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int r = lrow + 32 * h;
      *(float4 *)(&As[buf][r * GB_LD + lc4 * 4]) = pa[h];
      *(float4 *)(&Bs[buf][r * GB_LD + lc4 * 4]) = pb[h];
    }
  };

  f32x4 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nk = g.K / GB_K;
  gload(0);
  sstore(0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) gload((kt + 1) * GB_K);
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      float4 av[2], bv[2];
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) av[mi] = *(const float4 *)(&As[cur][(wr * 32 + mi * 16 + i16) * GB_LD + kb * 16 + kk * 4]);
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) bv[ni] = *(const float4 *)(&Bs[cur][(wc * 32 + ni * 16 + i16) * GB_LD + kb * 16 + kk * 4]);
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mi].x, bv[ni].x, acc[mi][ni], 0, 0, 0);
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mi].y, bv[ni].y, acc[mi][ni], 0, 0, 0);
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mi].z, bv[ni].z, acc[mi][ni], 0, 0, 0);
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mi].w, bv[ni].w, acc[mi][ni], 0, 0, 0);
        }
    }
    if (kt + 1 < nk) sstore(cur ^ 1);
    __syncthreads();
  }
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      const int gn = n0 + wc * 32 + ni * 16 + i16;
      const float bv = (gn < g.N) ? g.bias[gn] : 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int gm = m0 + wr * 32 + mi * 16 + kk * 4 + r;
        if (gm < g.M && gn < g.N) g.C[(size_t)gm * g.N + gn] = acc[mi][ni][r] + bv;
      }
    }
}

// ------------------------------------------------------------------------------------------
// GRU recurrence (+ detect head for the last layer)
// ------------------------------------------------------------------------------------------
struct gru_args {
  const float *gx;   // [Nw][OT][2*3H]  input projections incl. b_x, cols [dir][gate z,r,h][unit]
  const float *wh;   // [2][3H][H]
  const float *bh;   // [2][3H]
  float *seq;        // layer 1: [Nw][OT][2H] (fwd | bwd); layer 2: nullptr
  float *enc;        // layer 2: optional [Nw][2H] encoder output (fwd_last | bwd_last)
  const float *w1, *b1, *w2, *b2;  // detect head (layer 2)
  float *out;        // [Nw][NOUT]
  int OT, NOUT, HEAD;
};

// H == 32.  Block = 128 threads: wave 0 forward, wave 1 backward, one window per block.
template <bool LAST>
__global__ __launch_bounds__(128) void gru_kernel(gru_args a) {
  constexpr int H = 32;
  __shared__ __align__(16) float hbuf[2][2][H];  // [dir][ping-pong][unit]
  __shared__ float encs[2 * H];
  __shared__ float hid[2 * H];
  const int tid = threadIdx.x, lane = tid & 63, dir = tid >> 6;
  const int unit = lane & 31, half = lane >> 5;
  const int w = blockIdx.x;

  // recurrent weights of this lane: gates z, r, h of `unit`, columns [16*half, 16*half + 16)
  float wz[16], wr_[16], wc[16];
  {
    const float *base = a.wh + (size_t)dir * 3 * H * H;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float4 z = *(const float4 *)(base + (size_t)(0 * H + unit) * H + half * 16 + q * 4);
      float4 r = *(const float4 *)(base + (size_t)(1 * H + unit) * H + half * 16 + q * 4);
      float4 c = *(const float4 *)(base + (size_t)(2 * H + unit) * H + half * 16 + q * 4);
      wz[q * 4 + 0] = z.x; wz[q * 4 + 1] = z.y; wz[q * 4 + 2] = z.z; wz[q * 4 + 3] = z.w;
      wr_[q * 4 + 0] = r.x; wr_[q * 4 + 1] = r.y; wr_[q * 4 + 2] = r.z; wr_[q * 4 + 3] = r.w;
      wc[q * 4 + 0] = c.x; wc[q * 4 + 1] = c.y; wc[q * 4 + 2] = c.z; wc[q * 4 + 3] = c.w;
    }
  }
  const float bz = a.bh[dir * 3 * H + unit], br = a.bh[dir * 3 * H + H + unit], bc = a.bh[dir * 3 * H + 2 * H + unit];
  if (half == 0) hbuf[dir][0][unit] = 0.f;
  float h_own = 0.f;
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();

  const float *gxw = a.gx + (size_t)w * a.OT * 6 * H + dir * 3 * H + unit;
  int t = dir ? a.OT - 1 : 0;
  float gz = gxw[(size_t)t * 6 * H], gr = gxw[(size_t)t * 6 * H + H], gc = gxw[(size_t)t * 6 * H + 2 * H];
  for (int s = 0; s < a.OT; ++s) {
    const int cur = s & 1;
    // prefetch next step's input projection
    const int tn = dir ? t - 1 : t + 1;
    float ngz = 0.f, ngr = 0.f, ngc = 0.f;
    if (s + 1 < a.OT) {
      ngz = gxw[(size_t)tn * 6 * H];
      ngr = gxw[(size_t)tn * 6 * H + H];
      ngc = gxw[(size_t)tn * 6 * H + 2 * H];
    }
    const float4 *hp = (const float4 *)(&hbuf[dir][cur][half * 16]);
    float sz = 0.f, sr = 0.f, sc = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 hv = hp[q];
      sz = fmaf(wz[q * 4 + 0], hv.x, sz); sr = fmaf(wr_[q * 4 + 0], hv.x, sr); sc = fmaf(wc[q * 4 + 0], hv.x, sc);
      sz = fmaf(wz[q * 4 + 1], hv.y, sz); sr = fmaf(wr_[q * 4 + 1], hv.y, sr); sc = fmaf(wc[q * 4 + 1], hv.y, sc);
      sz = fmaf(wz[q * 4 + 2], hv.z, sz); sr = fmaf(wr_[q * 4 + 2], hv.z, sr); sc = fmaf(wc[q * 4 + 2], hv.z, sc);
      sz = fmaf(wz[q * 4 + 3], hv.w, sz); sr = fmaf(wr_[q * 4 + 3], hv.w, sr); sc = fmaf(wc[q * 4 + 3], hv.w, sc);
    }
    sz += __shfl_xor(sz, 32);
    sr += __shfl_xor(sr, 32);
    sc += __shfl_xor(sc, 32);
    // Keras GRU v2 (reset_after): z = s(xz + hz), r = s(xr + hr), c = tanh(xc + r * hc), h' = z*h + (1-z)*c
    const float z = sigmoid_f(gz + (sz + bz));
    const float r = sigmoid_f(gr + (sr + br));
    const float c = tanhf(gc + r * (sc + bc));
    h_own = z * h_own + (1.0f - z) * c;
    if (half == 0) {
      hbuf[dir][cur ^ 1][unit] = h_own;
      if (!LAST) a.seq[((size_t)w * a.OT + t) * 2 * H + dir * H + unit] = h_own;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    gz = ngz; gr = ngr; gc = ngc;
    t = tn;
  }
  if (LAST) {
    if (half == 0) {
      encs[dir * H + unit] = h_own;
      if (a.enc) a.enc[(size_t)w * 2 * H + dir * H + unit] = h_own;
    }
    __syncthreads();
    // detect head: Dense(64, relu) -> Dense(NOUT) -> sigmoid | softmax   (wave 0)
    if (dir == 0) {
      float acc = 0.f;
      const float *wrow = a.w1 + (size_t)lane * 2 * H;
#pragma unroll 8
      for (int k = 0; k < 2 * H; ++k) acc = fmaf(wrow[k], encs[k], acc);
      hid[lane] = fmaxf(acc + a.b1[lane], 0.f);
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
      float y = 0.f;
      if (lane < a.NOUT) {
        const float *w2 = a.w2 + (size_t)lane * 2 * H;
        for (int k = 0; k < 2 * H; ++k) y = fmaf(w2[k], hid[k], y);
        y += a.b2[lane];
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
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
struct crnn_ws {
  float *feat, *gx1, *seq1, *gx2;
};

static crnn_ws carve(const ww_model *m, int nw, void *ws) {
  const ww_crnn_dev &c = m->crnn;
  ww_bump b(ws, ~size_t(0));
  crnn_ws r;
  r.feat = b.take<float>((size_t)nw * c.OT * c.OF * c.C);
  r.gx1 = b.take<float>((size_t)nw * c.OT * 6 * c.H);
  r.seq1 = b.take<float>((size_t)nw * c.OT * 2 * c.H);
  r.gx2 = b.take<float>((size_t)nw * c.OT * 6 * c.H);
  return r;
}

size_t ww_crnn_workspace(const ww_model *m, int nw) {
  const ww_crnn_dev &c = m->crnn;
  return ww_bump::need((size_t)nw * c.OT * c.OF * c.C, 4) + 2 * ww_bump::need((size_t)nw * c.OT * 6 * c.H, 4) +
         ww_bump::need((size_t)nw * c.OT * 2 * c.H, 4) + 1024;
}

int ww_k_crnn_forward(ww_ctx *ctx, const ww_model *m, const float *d_mel, int64_t mel_rows, const int64_t *d_win_row,
                      const int32_t *d_win_valid, int64_t row0, int hop, int valid_const, int nw, void *ws,
                      float *d_out, float *d_enc) {
  if (nw <= 0) return WW_OK;
  const ww_crnn_dev &c = m->crnn;
  crnn_ws s = carve(m, nw, ws);
  win_addr wa = {d_win_row, d_win_valid, row0, hop, valid_const, mel_rows};
  {
    conv_args a = {d_mel, wa, c.conv_w, c.conv_b, s.feat, c.n_mel, c.T, c.KF, c.KT, c.SF, c.ST, c.PF, c.PT, c.OF, c.OT};
    ww_launch_scope scope(ctx, "conv5x20_kernel");
    hipLaunchKernelGGL(conv5x20_kernel, dim3(nw), dim3(256), 0, ctx->stream, a);
  }
  const int M = nw * c.OT;
  {
    gemm_args g = {s.feat, c.wx1, c.bx1, s.gx1, M, 6 * c.H, c.OF * c.C};
    ww_launch_scope scope(ctx, "gemm_nt_kernel<gru1>");
    hipLaunchKernelGGL(gemm_nt_kernel, dim3((M + GB_M - 1) / GB_M, (6 * c.H + GB_N - 1) / GB_N), dim3(256), 0,
                       ctx->stream, g);
  }
  {
    gru_args a = {s.gx1, c.wh1, c.bh1, s.seq1, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, c.OT, c.NOUT, c.HEAD};
    ww_launch_scope scope(ctx, "gru_kernel<seq>");
    hipLaunchKernelGGL((gru_kernel<false>), dim3(nw), dim3(128), 0, ctx->stream, a);
  }
  {
    gemm_args g = {s.seq1, c.wx2, c.bx2, s.gx2, M, 6 * c.H, 2 * c.H};
    ww_launch_scope scope(ctx, "gemm_nt_kernel<gru2>");
    hipLaunchKernelGGL(gemm_nt_kernel, dim3((M + GB_M - 1) / GB_M, (6 * c.H + GB_N - 1) / GB_N), dim3(256), 0,
                       ctx->stream, g);
  }
  {
    gru_args a = {s.gx2, c.wh2, c.bh2, nullptr, d_enc, c.w1, c.b1, c.w2, c.b2, d_out, c.OT, c.NOUT, c.HEAD};
    ww_launch_scope scope(ctx, "gru_kernel<last+head>");
    hipLaunchKernelGGL((gru_kernel<true>), dim3(nw), dim3(128), 0, ctx->stream, a);
  }
  WW_HIP(ctx, hipGetLastError());
  return WW_OK;
}
