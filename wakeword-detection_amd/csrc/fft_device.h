// Device-side FFT helpers shared by the batch front end (frontend.hip) and the streaming
// front end (streams.hip).
//
// 512-point real FFT of one frame by one wavefront = 256-point complex radix-4 Stockham FFT
// (4 stages, each lane owns one radix-4 butterfly per stage, stages exchange through a
// per-wave LDS buffer) + the real-FFT untangling pass.  Everything that depends only on the
// lane - the Hann samples it multiplies, its twiddles for every stage and for the untangling
// pass - is loaded ONCE per wavefront into registers (fft_consts) and reused for all frames.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

#define WIN 512
#define NB 257
#define WW_MEL_TAPS 36  // longest band of the shipped filterbank (checked at model load)
#define FFT_LD 288  // per-wave complex buffer: 256 points + 1 pad slot per 8 (bank spreading)

template <typename R>
struct cplx {
  R re, im;
};

template <typename R>
__device__ __forceinline__ cplx<R> cmul(cplx<R> a, cplx<R> b) {
  return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re};
}

__device__ __forceinline__ void wave_sync() {
  // LDS traffic of one wave is processed in order; wait for it only (not for outstanding
  // global loads, which an acq_rel fence would also drain) and stop compiler reordering
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ int fft_slot(int i) { return i + (i >> 3); }

template <typename R>
struct fft_consts {
  double hann[8];       // np.hanning(512)[2n], [2n+1] for n = lane + 64 r
  cplx<R> tw[3][3];     // stage s = 1..3, r = 1..3: e^{-2 pi i (lane % Ns) r / (4 Ns)}
  cplx<R> un[4];        // untangle: e^{-2 pi i k / 512}, k = lane + 64 r
};

template <typename R>
__device__ __forceinline__ void fft_load_consts(fft_consts<R> &c, int lane, const double *__restrict__ hann,
                                                const double *__restrict__ tw256, const double *__restrict__ tw512) {
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int n = lane + 64 * r;
    const double2 h = *(const double2 *)(hann + 2 * n);
    c.hann[2 * r] = h.x;
    c.hann[2 * r + 1] = h.y;
    const double2 u = *(const double2 *)(tw512 + 2 * n);
    c.un[r] = {(R)u.x, (R)u.y};
  }
#pragma unroll
  for (int s = 1; s < 4; ++s) {
    const int Ns = 1 << (2 * s);
    const int k = lane & (Ns - 1);
    const int step = 64 / Ns;
#pragma unroll
    for (int r = 1; r < 4; ++r) {
      const double2 t = *(const double2 *)(tw256 + 2 * (k * r * step));
      c.tw[s - 1][r - 1] = {(R)t.x, (R)t.y};
    }
  }
}

// One frame by one wavefront.  x2(n) returns samples (2n, 2n+1) of the frame as a float2.
// buf: per-wave LDS [FFT_LD] complex; mag: per-wave LDS [>= 257] floats.
template <typename R, typename XF>
__device__ __forceinline__ void frame_fft_mag(XF x2, const fft_consts<R> &c, cplx<R> *buf, float *mag, int lane) {
  cplx<R> v[4];
  // ---- stage 0 (Ns = 1): inputs straight from the sample tile, twiddles are 1.  The Hann
  // product is formed in fp64 like the reference (frame * np.hanning(512), tflite.py:175).
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float2 s = x2(lane + 64 * r);
    v[r].re = (R)((double)s.x * c.hann[2 * r]);
    v[r].im = (R)((double)s.y * c.hann[2 * r + 1]);
  }
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int Ns = 1 << (2 * s);
    const int k = lane & (Ns - 1);
    if (s > 0) {
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = buf[fft_slot(lane + 64 * r)];
#pragma unroll
      for (int r = 1; r < 4; ++r) v[r] = cmul(v[r], c.tw[s - 1][r - 1]);
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
    buf[fft_slot(j0)] = y0;
    buf[fft_slot(j0 + Ns)] = y1;
    buf[fft_slot(j0 + 2 * Ns)] = y2;
    buf[fft_slot(j0 + 3 * Ns)] = y3;
    wave_sync();
  }
  // ---- real-FFT untangle: X[k] = E[k] + W512^k O[k], k = 0..256; |X| in fp32
  // (fp64 power rounded to fp32, then one fp32 sqrt: <= 0.75 ulp from np.abs(...).astype(f32))
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int k = lane + 64 * r;
    const cplx<R> a = buf[fft_slot(k)];
    const cplx<R> b = buf[fft_slot((256 - k) & 255)];
    const R er = (R)0.5 * (a.re + b.re), ei = (R)0.5 * (a.im - b.im);
    const R orr = (R)0.5 * (a.im + b.im), oi = (R)-0.5 * (a.re - b.re);
    const R re = er + orr * c.un[r].re - oi * c.un[r].im;
    const R im = ei + orr * c.un[r].im + oi * c.un[r].re;
    mag[k] = __fsqrt_rn((float)(re * re + im * im));
    if (k == 0) {
      // k = 256: W = -1, Z[256] = Z[0]  ->  X[256] = Re(Z0) - Im(Z0)
      const R re2 = a.re - a.im;
      mag[256] = __fsqrt_rn((float)(re2 * re2));
    }
  }
  wave_sync();
}

// Banded mel filter + log tail of one frame, lane = band.  wl: LDS [WW_MEL_TAPS][64] tap-major
// weights (zero padded), mg: this wave's 257 magnitudes in LDS.
//   y = scale * (ln(max(W x + b, floor)) + log_off)     (filter.tflite ops 0-4)
__device__ __forceinline__ float mel_band(const float *mg, const float *wl, int st, float bias, float floor_v,
                                          float log_off, float scale, int lane) {
  float acc = 0.0f;
#pragma unroll
  for (int i = 0; i < WW_MEL_TAPS; ++i) {
    int k = st + i;
    k = k > 256 ? 256 : k;  // padded taps carry weight 0
    acc = fmaf(wl[i * 64 + lane], mg[k], acc);
  }
  acc += bias;
  acc = fmaxf(acc, floor_v);
  return (logf(acc) + log_off) * scale;
}
