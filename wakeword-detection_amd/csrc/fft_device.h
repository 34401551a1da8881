// Device-side FFT helpers shared by the batch front end (frontend.hip) and the streaming
// front end (streams.hip).  512-point real FFT = 256-point complex radix-4 Stockham FFT per
// wavefront + untangling pass; see frontend.hip for the algorithm notes.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

#define WIN 512
#define NB 257

template <typename R>
struct cplx {
  R re, im;
};

template <typename R>
__device__ __forceinline__ cplx<R> cmul(cplx<R> a, cplx<R> b) {
  return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re};
}

__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}


template <typename R, bool F32IN>
__device__ __forceinline__ float load_sample(const void *tile, int i, float divisor, int clip) {
  if (F32IN) return ((const float *)tile)[i];
  float v = (float)((const int16_t *)tile)[i];
  v = __fdiv_rn(v, divisor);
  if (clip) v = fminf(fmaxf(v, -1.0f), 1.0f);
  return v;
}

// One frame by one wavefront.  x(i) returns the (pre-emphasised) sample i of the frame.
template <typename R, typename XF>
__device__ __forceinline__ void frame_fft_mag(XF x, const double *__restrict__ hann, const cplx<R> *tw256,
                                              const cplx<R> *tw512, cplx<R> *buf, float *mag, int lane) {
  cplx<R> v[4];
  // ---- stage 0 (Ns = 1): inputs straight from the sample tile, twiddles are 1
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    int n = lane + 64 * r;
    double2 h = *(const double2 *)(hann + 2 * n);
    v[r].re = (R)((double)x(2 * n) * h.x);
    v[r].im = (R)((double)x(2 * n + 1) * h.y);
  }
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int Ns = 1 << (2 * s);
    const int k = lane & (Ns - 1);
    if (s > 0) {
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = buf[lane + 64 * r];
      const int step = 64 / Ns;
#pragma unroll
      for (int r = 1; r < 4; ++r) v[r] = cmul(v[r], tw256[k * r * step]);
    }
    cplx<R> t0 = {v[0].re + v[2].re, v[0].im + v[2].im};
    cplx<R> t1 = {v[0].re - v[2].re, v[0].im - v[2].im};
    cplx<R> t2 = {v[1].re + v[3].re, v[1].im + v[3].im};
    cplx<R> t3 = {v[1].im - v[3].im, -(v[1].re - v[3].re)};  // (v1 - v3) * (-i)
    cplx<R> y0 = {t0.re + t2.re, t0.im + t2.im};
    cplx<R> y1 = {t1.re + t3.re, t1.im + t3.im};
    cplx<R> y2 = {t0.re - t2.re, t0.im - t2.im};
    cplx<R> y3 = {t1.re - t3.re, t1.im - t3.im};
    const int j0 = ((lane >> (2 * s)) << (2 * s + 2)) + k;
    wave_sync();  // every lane has finished reading this stage's inputs
    buf[j0] = y0;
    buf[j0 + Ns] = y1;
    buf[j0 + 2 * Ns] = y2;
    buf[j0 + 3 * Ns] = y3;
    wave_sync();
  }
  // ---- real-FFT untangle: X[k] = E[k] + W512^k O[k], k = 0..256
#pragma unroll
  for (int r = 0; r < 5; ++r) {
    int k = lane + 64 * r;
    if (k > 256) break;
    cplx<R> a = buf[k & 255];
    cplx<R> b = buf[(256 - k) & 255];
    R er = (R)0.5 * (a.re + b.re), ei = (R)0.5 * (a.im - b.im);
    R orr = (R)0.5 * (a.im + b.im), oi = (R)-0.5 * (a.re - b.re);
    cplx<R> w = (k < 256) ? tw512[k] : cplx<R>{(R)-1, (R)0};
    R re = er + orr * w.re - oi * w.im;
    R im = ei + orr * w.im + oi * w.re;
    mag[k] = (float)sqrt(re * re + im * im);
  }
  wave_sync();
}

