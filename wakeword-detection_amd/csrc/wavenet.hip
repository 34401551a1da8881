// Wavenet encode + detect for gfx950 (fp32 MFMA, one persistent workgroup per window).
//
// Replaces encode.tflite + detect.tflite of the reference Wavenet (tf_lite_models/Wavenet;
// architecture wwdetect/wavenet/wavenet_model.py:11-128; call sites
// spokestack/wakeword/tflite.py:205-231, utils/evaluate_models.py:83-86).
//
// A 768-thread workgroup (12 wavefronts, 3 per SIMD) walks all 24 gated residual blocks of one
// 182x40 window without leaving the CU.  Time is the MFMA M dimension: 182 -> 12 tiles of 16
// rows, one tile per wave, so three waves share each SIMD's matrix pipe and cover each other's
// LDS round trips (4 waves x 3 tiles measured 10 % slower).
// The residual stream x[182][16] and the skip accumulator [182][32] never leave registers:
// they sit in the v_mfma_f32_16x16x4_f32 accumulator layout (lane -> column, 4 rows per
// register quad), which is also the layout the next block's epilogue needs.  Only the
// BatchNorm output u (which the dilated taps of OTHER rows read) and the gate product g
// (D-layout -> A-layout transpose) go through LDS:
//     u = x*s + t                      -> LDS (double buffered, 16 zero rows in front = causal pad)
//     [sig|tanh] = u[t-(2-k)d] * Wg    3 taps x 16 ch = K 48, N 32     24 MFMA / 16 rows
//     g = tanh(.)*sigmoid(.)           -> LDS (wave-private tile)
//     [res|skip] = g * Wrs             K 16, N 48                      12 MFMA / 16 rows
//     x += relu(res); skip += relu(skip_b)
// One __syncthreads per block.  The detect head (ReLU, 1x1 32->32 ReLU, 1x1 32->2, max over
// time, softmax) runs in the same launch.
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define WV_T 192      // padded time (12 tiles of 16)
#define WV_C 16
#define WV_S 32
#define WV_PAD 16     // causal zero rows in front of u
#define WV_INLD 48    // staged input row stride (40 mel + zero pad to 3 k-blocks)
#ifndef WV_NW
#define WV_NW 12      // wavefronts per workgroup (3 per SIMD): 12 row tiles of 16 frames, one per wave
#endif
#define WV_MPW (12 / WV_NW)  // row tiles per wave
#define WV_THREADS (WV_NW * 64)

struct win_addr_w {
  const int64_t *row;
  const int32_t *valid;
  int64_t row0;
  int hop;
  int valid_const;
  int64_t mel_rows;
};

struct wave_args {
  const float *mel;
  win_addr_w wa;
  int T, n_mel, NB, NOUT;
  unsigned long long dil4[2];  // dilation of block b: 4 bits at bit 4*(b%16) of dil4[b/16]  (d <= 8, NB <= 32)
  unsigned int has_res_mask;   // bit b: block b has a residual 1x1 conv
  const float *w_in4;   // [3 kb][4 kk][16 col][4 q]   (K = 40 padded to 48)
  const float *b_in;    // [16]
  const float *bn_s, *bn_t;  // [NB][16]
  const float *w_gate4; // [NB][3 kb][4 kk][32 col][4 q]
  const float *b_gate;  // [NB][32]  (sig | tanh)
  const float *w_rs4;   // [NB][4 kk][48 col][4 q]
  const float *b_rs;    // [NB][48]  (res | skip)
  const float *d_w1_4;  // [2 kb][4 kk][32 col][4 q]
  const float *d_b1;    // [32]
  const float *d_w2_4;  // [2 kb][4 kk][16 col][4 q]  (NOUT padded to 16)
  const float *d_b2;    // [16]
  float *out;           // [Nw][NOUT]
  float *enc;           // optional [Nw][T][32]
  const float *enc_in;  // HEAD_ONLY: encoder output to run the detect graph on
};

__device__ __forceinline__ float sigmoid_w(float x) { return 1.0f / (1.0f + expf(-x)); }

__device__ __forceinline__ void wsync() {
  // LDS traffic of one wave is processed in order; wait for it only (not for outstanding
  // global loads, which an acq_rel fence would also drain) and stop compiler reordering
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
}

// v_exp_f32 / v_rcp_f32 gates (~1 ulp each, |error| ~2e-7): the gate activations are 16x2 values
// per row and block; with libm tanhf/expf + IEEE division they cost more VALU time than the
// block's MFMAs.
__device__ __forceinline__ float fast_sigmoid_w(float x) {
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}
__device__ __forceinline__ float fast_tanh_w(float x) {
  const float e = __builtin_amdgcn_exp2f(2.8853900817779268f * x);  // exp(2x): inf -> 1, 0 -> -1
  return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + e);
}

struct wave_blk {
  float4 wg[3][2], wrs[3];
  float bn_s, bn_t, bsig, btanh, bres, bsk0, bsk1;
};

#define MFMA4(acc, av, bv)                                              \
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, bv.x, acc, 0, 0, 0); \
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, bv.y, acc, 0, 0, 0); \
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, bv.z, acc, 0, 0, 0); \
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, bv.w, acc, 0, 0, 0);

__device__ __forceinline__ void wave_blk_load(const wave_args &a, int blk, int j, int kk, wave_blk &p) {
#pragma unroll
  for (int kb = 0; kb < 3; ++kb)
#pragma unroll
    for (int n = 0; n < 2; ++n)
      p.wg[kb][n] = *(const float4 *)(a.w_gate4 + (((size_t)blk * 3 + kb) * 4 + kk) * 32 * 4 + (n * 16 + j) * 4);
#pragma unroll
  for (int n = 0; n < 3; ++n) p.wrs[n] = *(const float4 *)(a.w_rs4 + ((size_t)blk * 4 + kk) * 48 * 4 + (n * 16 + j) * 4);
  p.bn_s = a.bn_s[blk * WV_C + j];
  p.bn_t = a.bn_t[blk * WV_C + j];
  p.bsig = a.b_gate[blk * 32 + j];
  p.btanh = a.b_gate[blk * 32 + 16 + j];
  p.bres = a.b_rs[blk * 48 + j];
  p.bsk0 = a.b_rs[blk * 48 + 16 + j];
  p.bsk1 = a.b_rs[blk * 48 + 32 + j];
}

template <bool HEAD_ONLY>
__global__ __launch_bounds__(WV_THREADS) void wavenet_kernel(wave_args a) {
  // LDS: region A = staged input [192][48] (prologue only), later u[2][208][16] + g[192][16]
  __shared__ __align__(16) float lds[WV_T * WV_INLD > (2 * (WV_T + WV_PAD) * WV_C + WV_T * WV_S) ? WV_T * WV_INLD
                                                                                                  : (2 * (WV_T + WV_PAD) * WV_C + WV_T * WV_S)];
  __shared__ float red[WV_NW][16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 15, kk = lane >> 4;
  const int w = blockIdx.x;
  const int T = a.T;

  f32x4 x[WV_MPW], skip[WV_MPW][2];
  float *ubuf = lds;                                    // [2][WV_T + WV_PAD][16]
  float *gbuf = lds + 2 * (WV_T + WV_PAD) * WV_C;       // [WV_T][16] (wave-private tiles)
  float *hbuf = gbuf;                                   // detect head reuses it as [WV_T][32]
  if (HEAD_ONLY) {
    // detect.tflite alone (reference detect_model(x), wakeword/tflite.py:231): skip sums come from memory
    const float *e = a.enc_in + (size_t)w * T * WV_S;
#pragma unroll
    for (int mi = 0; mi < WV_MPW; ++mi)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int t = (wave * WV_MPW + mi) * 16 + kk * 4 + r;
        skip[mi][0][r] = t < T ? e[(size_t)t * WV_S + j] : 0.f;
        skip[mi][1][r] = t < T ? e[(size_t)t * WV_S + 16 + j] : 0.f;
      }
    (void)x; (void)ubuf;
  } else {
  int64_t row = a.wa.row ? a.wa.row[w] : a.wa.row0 + (int64_t)w * a.wa.hop;
  int valid = a.wa.valid ? a.wa.valid[w] : a.wa.valid_const;
  if (valid > T) valid = T;
  if (row + valid > a.wa.mel_rows) valid = (int)(a.wa.mel_rows - row);
  if (valid < 0) valid = 0;

  // ---- stage the window: in_lds[t][0..47], zero outside [0,valid) x [0,n_mel)
  float *in_lds = lds;
  for (int i = tid; i < WV_T * WV_INLD / 4; i += WV_THREADS) ((float4 *)in_lds)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  __syncthreads();
  {
    const float *src = a.mel + row * a.n_mel;
    const int n = valid * a.n_mel;
    for (int i = tid; i < n; i += WV_THREADS) {
      int t = i / a.n_mel, c = i - t * a.n_mel;
      in_lds[t * WV_INLD + c] = src[i];
    }
  }
  __syncthreads();

  // ---- input 1x1 conv + ReLU -> x in accumulator layout.  m-tile mi of this wave covers rows
  // (wave*3 + mi)*16 .. +15; lane holds rows kk*4 + r, column j.
  {
    float4 bw[3];
#pragma unroll
    for (int kb = 0; kb < 3; ++kb) bw[kb] = *(const float4 *)(a.w_in4 + ((size_t)(kb * 4 + kk) * 16 + j) * 4);
    const float bias = a.b_in[j];
#pragma unroll
    for (int mi = 0; mi < WV_MPW; ++mi) {
      const int t0 = (wave * WV_MPW + mi) * 16;
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kb = 0; kb < 3; ++kb) {
        const float4 av = *(const float4 *)(in_lds + (t0 + j) * WV_INLD + kb * 16 + kk * 4);
        MFMA4(acc, av, bw[kb]);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) x[mi][r] = fmaxf(acc[r] + bias, 0.f);
      skip[mi][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
      skip[mi][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
  }
  __syncthreads();  // in_lds is dead from here on

  // causal zero rows of both u buffers
  for (int i = tid; i < 2 * WV_PAD * WV_C; i += WV_THREADS) {
    int b = i / (WV_PAD * WV_C), o = i - b * (WV_PAD * WV_C);
    ubuf[b * (WV_T + WV_PAD) * WV_C + o] = 0.f;
  }

  // block parameters are prefetched one block ahead (two register sets, loop unrolled by two)
  wave_blk pb[2];
  wave_blk_load(a, 0, j, kk, pb[0]);
  auto run_block = [&](int blk, const wave_blk &P, wave_blk &Pnext) {
    float *u = ubuf + (blk & 1) * (WV_T + WV_PAD) * WV_C + WV_PAD * WV_C;  // row 0 of u
    const int d = (int)((a.dil4[blk >> 4] >> (4 * (blk & 15))) & 15);  // kernel-argument SGPRs, no load
    // ---- BatchNorm affine (wavenet_model.py:57) -> LDS
#pragma unroll
    for (int mi = 0; mi < WV_MPW; ++mi) {
      const int t0 = (wave * WV_MPW + mi) * 16 + kk * 4;
#pragma unroll
      for (int r = 0; r < 4; ++r) u[(t0 + r) * WV_C + j] = x[mi][r] * P.bn_s + P.bn_t;
    }
    // next block's parameters: issued unconditionally (index clamped) - a conditional prefetch makes
    // the compiler drain ALL outstanding loads at the join, i.e. wait for the prefetch it just issued
    wave_blk_load(a, blk + 1 < a.NB ? blk + 1 : blk, j, kk, Pnext);
    const float bsig = P.bsig, btanh = P.btanh, bres = P.bres, bsk0 = P.bsk0, bsk1 = P.bsk1;
    const int has_res = (a.has_res_mask >> blk) & 1;
    __syncthreads();  // u complete (all rows, all waves)

#pragma unroll
    for (int mi = 0; mi < WV_MPW; ++mi) {
      const int t0 = (wave * WV_MPW + mi) * 16;
      f32x4 as = {0.f, 0.f, 0.f, 0.f}, at = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kb = 0; kb < 3; ++kb) {
        // tap kb reads u[t - (2 - kb) * d]; rows < 0 hit the zero pad (d <= 8 -> >= -16)
        const float4 av = *(const float4 *)(u + (t0 + j - (2 - kb) * d) * WV_C + kk * 4);
        MFMA4(as, av, P.wg[kb][0]);
        MFMA4(at, av, P.wg[kb][1]);
      }
      float *gt = gbuf + t0 * WV_C;
#pragma unroll
      for (int r = 0; r < 4; ++r) gt[(kk * 4 + r) * WV_C + j] = fast_tanh_w(at[r] + btanh) * fast_sigmoid_w(as[r] + bsig);
      wsync();
      const float4 gv = *(const float4 *)(gt + j * WV_C + kk * 4);
      f32x4 ar = {0.f, 0.f, 0.f, 0.f}, s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
      MFMA4(ar, gv, P.wrs[0]);
      MFMA4(s0, gv, P.wrs[1]);
      MFMA4(s1, gv, P.wrs[2]);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (has_res) x[mi][r] = fmaxf(ar[r] + bres, 0.f) + x[mi][r];
        skip[mi][0][r] = skip[mi][0][r] + fmaxf(s0[r] + bsk0, 0.f);
        skip[mi][1][r] = skip[mi][1][r] + fmaxf(s1[r] + bsk1, 0.f);
      }
      wsync();
    }
  };
  for (int blk = 0; blk < a.NB; blk += 2) {
    run_block(blk, pb[0], pb[1]);
    if (blk + 1 < a.NB) run_block(blk + 1, pb[1], pb[0]);
  }
  __syncthreads();

  }
  // ---- encoder output (optional) + detect head
  if (a.enc) {
    float *e = a.enc + (size_t)w * T * WV_S;
#pragma unroll
    for (int mi = 0; mi < WV_MPW; ++mi)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int t = (wave * WV_MPW + mi) * 16 + kk * 4 + r;
        if (t < T) {
          e[(size_t)t * WV_S + j] = skip[mi][0][r];
          e[(size_t)t * WV_S + 16 + j] = skip[mi][1][r];
        }
      }
  }
  float4 w1[2][2], w2[2];
#pragma unroll
  for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
    for (int n = 0; n < 2; ++n) w1[kb][n] = *(const float4 *)(a.d_w1_4 + ((size_t)(kb * 4 + kk) * 32 + n * 16 + j) * 4);
    w2[kb] = *(const float4 *)(a.d_w2_4 + ((size_t)(kb * 4 + kk) * 16 + j) * 4);
  }
  const float b1a = a.d_b1[j], b1b = a.d_b1[16 + j], b2 = a.d_b2[j];
  float best = -INFINITY;
#pragma unroll
  for (int mi = 0; mi < WV_MPW; ++mi) {
    const int t0 = (wave * WV_MPW + mi) * 16;
    float *ht = hbuf + t0 * WV_S;  // [16][32] tile, wave-private
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      ht[(kk * 4 + r) * WV_S + j] = fmaxf(skip[mi][0][r], 0.f);
      ht[(kk * 4 + r) * WV_S + 16 + j] = fmaxf(skip[mi][1][r], 0.f);
    }
    wsync();
    f32x4 h0 = {0.f, 0.f, 0.f, 0.f}, h1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      const float4 av = *(const float4 *)(ht + j * WV_S + kb * 16 + kk * 4);
      MFMA4(h0, av, w1[kb][0]);
      MFMA4(h1, av, w1[kb][1]);
    }
    wsync();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      ht[(kk * 4 + r) * WV_S + j] = fmaxf(h0[r] + b1a, 0.f);
      ht[(kk * 4 + r) * WV_S + 16 + j] = fmaxf(h1[r] + b1b, 0.f);
    }
    wsync();
    f32x4 y = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      const float4 av = *(const float4 *)(ht + j * WV_S + kb * 16 + kk * 4);
      MFMA4(y, av, w2[kb]);
    }
    wsync();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int t = t0 + kk * 4 + r;
      if (t < T) best = fmaxf(best, y[r] + b2);
    }
  }
  // GlobalMaxPooling1D over time: reduce over kk (lanes j, j+16, j+32, j+48), then over waves
  best = fmaxf(best, __shfl_xor(best, 16));
  best = fmaxf(best, __shfl_xor(best, 32));
  if (lane < 16) red[wave][lane] = best;
  __syncthreads();
  if (tid < 16) {
    float v = red[0][tid];
#pragma unroll
    for (int q = 1; q < WV_NW; ++q) v = fmaxf(v, red[q][tid]);
    float mx = (tid < a.NOUT) ? v : -INFINITY;
    for (int o = 1; o < 16; o <<= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    float e = (tid < a.NOUT) ? expf(v - mx) : 0.f;
    float sum = e;
    for (int o = 1; o < 16; o <<= 1) sum += __shfl_xor(sum, o);
    if (tid < a.NOUT) a.out[(size_t)w * a.NOUT + tid] = e / sum;
  }
}

size_t ww_wave_workspace(const ww_model *, int) { return 256; }

int ww_k_wave_forward(ww_ctx *ctx, const ww_model *m, const float *d_mel, int64_t mel_rows, const int64_t *d_win_row,
                      const int32_t *d_win_valid, int64_t row0, int hop, int valid_const, int nw, void *, float *d_out,
                      float *d_enc) {
  if (nw <= 0) return WW_OK;
  const ww_wave_dev &v = m->wave;
  wave_args a = {};
  a.mel = d_mel;
  a.wa = {d_win_row, d_win_valid, row0, hop, valid_const, mel_rows};
  a.T = v.T; a.n_mel = v.n_mel; a.NB = v.NB; a.NOUT = v.NOUT;
  if (v.NB > 32) return ww_fail(ctx, WW_EINVAL, "Wavenet with %d blocks: kernel limit 32", v.NB);
  for (int b = 0; b < v.NB; ++b) {
    if (v.dil[b] < 1 || v.dil[b] > 8) return ww_fail(ctx, WW_EINVAL, "dilation %d of block %d outside 1..8", v.dil[b], b);
    a.dil4[b >> 4] |= (unsigned long long)v.dil[b] << (4 * (b & 15));
    if (v.has_res[b]) a.has_res_mask |= 1u << b;
  }
  a.w_in4 = v.w_in; a.b_in = v.b_in; a.bn_s = v.bn_s; a.bn_t = v.bn_t;
  a.w_gate4 = v.w_gate; a.b_gate = v.b_gate; a.w_rs4 = v.w_rs; a.b_rs = v.b_rs;
  a.d_w1_4 = v.d_w1; a.d_b1 = v.d_b1; a.d_w2_4 = v.d_w2; a.d_b2 = v.d_b2;
  a.out = d_out; a.enc = d_enc;
  ww_launch_scope scope(ctx, "wavenet_kernel");
  hipLaunchKernelGGL(wavenet_kernel<false>, dim3(nw), dim3(WV_THREADS), 0, ctx->stream, a);
  WW_HIP(ctx, hipGetLastError());
  return WW_OK;
}

int ww_k_wave_detect(ww_ctx *ctx, const ww_model *m, const float *d_enc, int nw, float *d_out) {
  if (nw <= 0) return WW_OK;
  const ww_wave_dev &v = m->wave;
  wave_args a = {};
  a.T = v.T; a.n_mel = v.n_mel; a.NB = v.NB; a.NOUT = v.NOUT;
  a.d_w1_4 = v.d_w1; a.d_b1 = v.d_b1; a.d_w2_4 = v.d_w2; a.d_b2 = v.d_b2;
  a.out = d_out; a.enc = nullptr; a.enc_in = d_enc;
  ww_launch_scope scope(ctx, "wavenet_detect_kernel");
  hipLaunchKernelGGL(wavenet_kernel<true>, dim3(nw), dim3(WV_THREADS), 0, ctx->stream, a);
  WW_HIP(ctx, hipGetLastError());
  return WW_OK;
}
