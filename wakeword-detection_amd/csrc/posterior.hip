// Posterior smoothing + threshold sweep on the GPU.
//
// Replaces the numeric core of plot_FRR_FAR (utils/evaluate_models.py:185-218):
//   neg' = np.convolve(neg, ones(w)/w, 'same')          fp64, centre of the full convolution:
//          neg'[i] = sum_{j=i-w/2}^{i+(w-1)/2} neg[j] * (1/w)    (w = 30: j in [i-15, i+14])
//   accepts[k] = #(pos > thr[k])                         (fp32 compare: with the NumPy 1.19 the
//                                                         reference pins, a float32 array against
//                                                         a float64 scalar compares in float32)
//   fa[k]      = #{i : neg'[i] > thr[k] and not neg'[i-1] > thr[k]}   (rising edges, fp64)
// The stream is HBM-bound integer/compare work: one pass over neg for the smoothing, one pass
// over neg' for all thresholds at once (thresholds in LDS, per-block counters in LDS, one
// global atomic per (block, threshold)).
#include "common.h"

__global__ __launch_bounds__(256) void smooth_kernel(const float *__restrict__ neg, int64_t n, int win,
                                                     double *__restrict__ out) {
  const double v = 1.0 / (double)win;
  const int64_t shift = (win - 1) / 2;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    if (win <= 0) {
      out[i] = (double)neg[i];
      continue;
    }
    const int64_t hi = i + shift, lo = hi - (win - 1);
    double acc = 0.0;
    for (int64_t j = lo; j <= hi; ++j)
      if (j >= 0 && j < n) acc += (double)neg[j] * v;
    out[i] = acc;
  }
}

#define SW_MAX_THR 1024

__global__ __launch_bounds__(256) void sweep_kernel(const double *__restrict__ sm, int64_t n, const float *__restrict__ pos,
                                                    int64_t n_pos, const double *__restrict__ thr, int n_thr,
                                                    unsigned long long *__restrict__ pos_cnt,
                                                    unsigned long long *__restrict__ fa_cnt) {
  __shared__ double s_thr[SW_MAX_THR];
  __shared__ unsigned int s_fa[SW_MAX_THR];
  __shared__ unsigned int s_pos[SW_MAX_THR];
  for (int k = threadIdx.x; k < n_thr; k += blockDim.x) {
    s_thr[k] = thr[k];
    s_fa[k] = 0;
    s_pos[k] = 0;
  }
  __syncthreads();
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const double cur = sm[i];
    const double prev = i > 0 ? sm[i - 1] : -1.0e300;
    // thresholds are ascending in the reference sweep but need not be: test all
    for (int k = 0; k < n_thr; ++k) {
      const double t = s_thr[k];
      if (cur > t && !(prev > t)) atomicAdd(&s_fa[k], 1u);
    }
  }
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_pos; i += stride) {
    const float p = pos[i];
    for (int k = 0; k < n_thr; ++k)
      if (p > (float)s_thr[k]) atomicAdd(&s_pos[k], 1u);
  }
  __syncthreads();
  for (int k = threadIdx.x; k < n_thr; k += blockDim.x) {
    if (s_fa[k]) atomicAdd(&fa_cnt[k], (unsigned long long)s_fa[k]);
    if (s_pos[k]) atomicAdd(&pos_cnt[k], (unsigned long long)s_pos[k]);
  }
}

int ww_k_far_frr(ww_ctx *ctx, const float *d_pos, int64_t n_pos, const float *d_neg, int64_t n_neg, int win,
                 const double *d_thr, int n_thr, double *d_smoothed, unsigned long long *d_pos_cnt,
                 unsigned long long *d_fa_cnt) {
  if (n_thr > SW_MAX_THR) return ww_fail(ctx, WW_EINVAL, "at most %d thresholds", SW_MAX_THR);
  WW_HIP(ctx, hipMemsetAsync(d_pos_cnt, 0, sizeof(unsigned long long) * n_thr, ctx->stream));
  WW_HIP(ctx, hipMemsetAsync(d_fa_cnt, 0, sizeof(unsigned long long) * n_thr, ctx->stream));
  if (n_neg > 0) {
    int blocks = (int)((n_neg + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    ww_launch_scope scope(ctx, "smooth_kernel");
    hipLaunchKernelGGL(smooth_kernel, dim3(blocks), dim3(256), 0, ctx->stream, d_neg, n_neg, win, d_smoothed);
  }
  {
    int64_t work = n_neg > n_pos ? n_neg : n_pos;
    int blocks = (int)((work + 255) / 256);
    if (blocks < 1) blocks = 1;
    if (blocks > 2048) blocks = 2048;
    ww_launch_scope scope(ctx, "sweep_kernel");
    hipLaunchKernelGGL(sweep_kernel, dim3(blocks), dim3(256), 0, ctx->stream, d_smoothed, n_neg, d_pos, n_pos, d_thr,
                       n_thr, d_pos_cnt, d_fa_cnt);
  }
  WW_HIP(ctx, hipGetLastError());
  return WW_OK;
}

// ---- posterior pick + per-clip reduction (a13 / a14: evaluate_models.py:80,86 take element [1] of a detect row, :98-99 the
// max over a wake-word clip's windows) on the rows the model kernels left in HBM: element `pidx` of rows [n][n_out], either
// as they are (the negative stream: every window counts) or reduced to one maximum per run of windows (seg_offs[s] ..
// seg_offs[s + 1]: the windows of one clip).  4 bytes in per window; one lane per run walks its ~50 windows.
__global__ __launch_bounds__(256) void pick_kernel(const float *__restrict__ rows, int64_t n, int n_out, int pidx,
                                                   const int64_t *__restrict__ seg_offs, int64_t n_seg, float *__restrict__ out) {
  if (!seg_offs) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = rows[i * n_out + pidx];
    return;
  }
  // one wave per run: its ~50 windows are ONE round of loads (a lane per window), not a chain of them
  const int lane = threadIdx.x & 63;
  const int64_t s = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (s >= n_seg) return;
  int64_t lo = seg_offs[s], hi = seg_offs[s + 1];
  lo = lo < 0 ? 0 : lo;
  hi = hi > n ? n : hi;
  float m = -INFINITY;  // (an empty run: the caller never asks for one - np.max of an empty list raises in the reference)
  for (int64_t w = lo + lane; w < hi; w += 64) m = fmaxf(m, rows[w * n_out + pidx]);
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if (lane == 0) out[s] = m;
}

int ww_k_posterior_pick(ww_ctx *ctx, const float *d_rows, int64_t n, int n_out, int pidx, const int64_t *d_seg_offs, int64_t n_seg,
                        float *d_out) {
  const int64_t work = d_seg_offs ? n_seg : n;
  if (work <= 0) return WW_OK;
  const int64_t blocks = d_seg_offs ? (work + 3) / 4 : (work + 255) / 256;
  if (blocks > 0x7fffffff) return ww_fail(ctx, WW_EINVAL, "posterior pick: too many rows for one launch");
  ww_launch_scope scope(ctx, "pick_kernel");
  hipLaunchKernelGGL(pick_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, d_rows, n, n_out, pidx, d_seg_offs, n_seg, d_out);
  WW_HIP(ctx, hipGetLastError());
  return WW_OK;
}

// ---- superframe shortest-path smoothing (wwdetect/wfst.py:17-71; wiring utils/CRNN_files/tflite.py:252-263)
// The reference builds a 2-state x T lattice with pynini and takes the single shortest path in the
// tropical semiring (float32 weights): start arcs cost ln 2 + c[0][p]; an arc into state p at time t
// costs c[t][p], minus the stay bonus (1) when it does not change state; both last states are final.
// c = -ln(posterior).  That is a 2 x T Viterbi recursion; OpenFst relaxes predecessors in state
// order and replaces a distance only when strictly smaller, so ties keep state 0 ('other').
// One thread per superframe; `in` holds either posteriors [n][T][2] or ready-made costs.
#define VT_MAX_T 64
__global__ __launch_bounds__(256) void viterbi2_kernel(const float *__restrict__ in, int64_t n, int T, float stay_bonus,
                                                       int in_is_cost, unsigned char *__restrict__ path,
                                                       unsigned char *__restrict__ wake) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float *p = in + i * T * 2;
  auto cost = [&](int t, int q) { return in_is_cost ? p[t * 2 + q] : -logf(p[t * 2 + q]); };
  unsigned long long back0 = 0, back1 = 0;  // bit t: best predecessor of state 0 / 1 at time t is state 1
  const float init = (float)0.69314718055994530942;  // float32(-log(1/2)) as the Arc weight conversion yields
  // the start arc weight is float32(float64(ln 2) + float64(c)) in the reference (float64 + float32 scalar)
  (void)init;
  float d0 = (float)(0.6931471805599453 + (double)cost(0, 0));
  float d1 = (float)(0.6931471805599453 + (double)cost(0, 1));
  for (int t = 1; t < T; ++t) {
    const float c0 = cost(t, 0), c1 = cost(t, 1);
    const float stay0 = c0 - stay_bonus, stay1 = c1 - stay_bonus;  // float32, as `cost -= 1` on a numpy float32
    const float a00 = d0 + stay0, a10 = d1 + c0;   // into state 0 from 0 / from 1
    const float a01 = d0 + c1, a11 = d1 + stay1;   // into state 1 from 0 / from 1
    float n0 = a00, n1 = a01;
    if (a10 < n0) { n0 = a10; back0 |= 1ull << t; }
    if (a11 < n1) { n1 = a11; back1 |= 1ull << t; }
    d0 = n0;
    d1 = n1;
  }
  int st = d1 < d0 ? 1 : 0;
  int any = 0;
  for (int t = T - 1; t >= 0; --t) {
    path[i * T + t] = (unsigned char)st;
    any |= st;
    st = (int)(((st ? back1 : back0) >> t) & 1ull);
  }
  wake[i] = (unsigned char)any;
}

int ww_k_viterbi2(ww_ctx *ctx, const float *d_in, int64_t n, int T, float stay_bonus, int in_is_cost, unsigned char *d_path,
                  unsigned char *d_wake) {
  if (n <= 0) return WW_OK;
  if (T < 1 || T > VT_MAX_T) return ww_fail(ctx, WW_EINVAL, "superframe length %d outside 1..%d", T, VT_MAX_T);
  ww_launch_scope scope(ctx, "viterbi2_kernel");
  hipLaunchKernelGGL(viterbi2_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, d_in, n, T, stay_bonus,
                     in_is_cost, d_path, d_wake);
  WW_HIP(ctx, hipGetLastError());
  return WW_OK;
}
