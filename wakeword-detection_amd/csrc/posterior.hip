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
