// PCM -> log-mel front end for gfx950.
//
// Replaces the reference's per-sample RingBuffer loop, np.fft.rfft and the filter.tflite
// invoke (utils/tf_lite/filter.py:38-75; spokestack/wakeword/tflite.py:148-191).
//
// Kernel shape: one 256-thread workgroup (4 wavefronts) owns FPB=16 consecutive frames of
// one utterance.  The 512+15*160 samples those frames touch are loaded ONCE with coalesced
// 16-byte loads into LDS (raw int16, or fp32 for the float entry point); every frame is then
// produced by one wavefront: Hann product + a 256-point complex radix-4 Stockham FFT (each
// lane owns one radix-4 butterfly per stage, exchanging through a per-wave LDS buffer),
// the real-FFT untangling pass, |.|, the banded mel filter (one lane per band) and the
// log/affine tail.  The 16x40 mel tile leaves through LDS as coalesced float4 stores.
//
// REAL = double reproduces the reference numerics (the Hann product and FFT run in float64,
// spokestack/wakeword/tflite.py:175-176, then cast to float32); REAL = float is the fast mode.
#include "common.h"

#include "fft_device.h"

#define FPB 16            // frames per block
#define WAVES 4

struct logmel_args {
  const int16_t *pcm;
  const float *f32;
  const int64_t *sample_offs;
  const int64_t *frame_offs;
  int n_utt;
  int hop;
  float divisor;
  int clip;
  float preemph;
  // filter
  const int *start, *len, *woff;
  const float *w, *bias;
  int n_mel, total_taps;
  float floor_v, log_off, scale;
  const double *hann, *tw256, *tw512;
  float *mel;
  // stft-only mode
  const float *frames;
  float *mag_out;
  int64_t n_frames_direct;
};

template <typename R, bool F32IN>
__global__ __launch_bounds__(256) void logmel_kernel(logmel_args a) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int u = blockIdx.y;
  const int64_t s_begin = a.sample_offs[u], s_end = a.sample_offs[u + 1];
  const int64_t n_samples = s_end - s_begin;
  const int64_t nf = n_samples >= WIN ? (n_samples - WIN) / a.hop + 1 : 0;
  const int64_t f0 = (int64_t)blockIdx.x * FPB;
  if (f0 >= nf) return;
  const int nfb = (int)((nf - f0) < FPB ? (nf - f0) : FPB);

  // ---- LDS carve-up
  constexpr int ELT = F32IN ? 4 : 2;
  size_t off = 0;
  cplx<R> *tw256 = (cplx<R> *)(smem + off); off += 256 * sizeof(cplx<R>);
  cplx<R> *tw512 = (cplx<R> *)(smem + off); off += 256 * sizeof(cplx<R>);
  cplx<R> *fbuf = (cplx<R> *)(smem + off); off += WAVES * 256 * sizeof(cplx<R>);
  float *mag = (float *)(smem + off); off += WAVES * 260 * sizeof(float);
  float *mel_tile = (float *)(smem + off); off += FPB * 40 * sizeof(float);
  float *fw = (float *)(smem + off); off += ((a.total_taps + 3) & ~3) * sizeof(float);
  int *fmeta = (int *)(smem + off); off += 3 * 64 * sizeof(int);
  unsigned char *tile = smem + off;  // [tile_cap + 16] elements of ELT bytes

  for (int i = tid; i < 256; i += 256) {
    tw256[i] = {(R)a.tw256[2 * i], (R)a.tw256[2 * i + 1]};
    tw512[i] = {(R)a.tw512[2 * i], (R)a.tw512[2 * i + 1]};
  }
  for (int i = tid; i < a.total_taps; i += 256) fw[i] = a.w[i];
  if (tid < a.n_mel) {
    fmeta[tid] = a.start[tid];
    fmeta[64 + tid] = a.len[tid];
    fmeta[128 + tid] = a.woff[tid];
  }

  // ---- stage the sample tile: element 0 of the tile is the sample BEFORE the first one of
  // frame f0 (pre-emphasis halo; zero at the start of an utterance).
  const int64_t g_first = s_begin + f0 * a.hop;           // first sample of frame f0
  const int n_need = WIN + (nfb - 1) * a.hop;             // samples used by this block
  const int64_t g_lo = g_first - 1;                       // halo sample (may be < s_begin)
  constexpr int VEC = 16 / ELT;                           // elements per 16-byte load
  // tile slot i + shift holds global sample g_lo + i; the tile starts on a 16-byte boundary
  // of the global buffer so that every lane issues aligned 16-byte loads.
  const int shift = (int)(((g_lo % VEC) + VEC) % VEC);
  {
    const int64_t ga = g_lo - shift;                      // multiple of VEC (may be -VEC)
    const int n_vec = (shift + n_need + 1 + VEC - 1) / VEC;
    const int64_t total = a.sample_offs[a.n_utt];
    for (int q = tid; q < n_vec; q += 256) {
      const int64_t g = ga + (int64_t)q * VEC;
      uint4 v = {0u, 0u, 0u, 0u};
      if (g >= 0 && g + VEC <= total) {
        v = F32IN ? *(const uint4 *)(a.f32 + g) : *(const uint4 *)(a.pcm + g);
      } else {
        unsigned char *pv = (unsigned char *)&v;
        for (int e = 0; e < VEC; ++e) {
          const int64_t ge = g + e;
          if (ge >= 0 && ge < total) {
            if (F32IN) ((float *)pv)[e] = a.f32[ge];
            else ((int16_t *)pv)[e] = a.pcm[ge];
          }
        }
      }
      *(uint4 *)(tile + (size_t)q * 16) = v;
    }
  }
  __syncthreads();
  if (tid == 0 && g_lo < s_begin) {                       // utterance start: carry-in is zero
    if (F32IN) ((float *)tile)[shift] = 0.0f;
    else ((int16_t *)tile)[shift] = 0;
  }
  __syncthreads();

  const float alpha = a.preemph;
  cplx<R> *buf = fbuf + wave * 256;
  float *mg = mag + wave * 260;
  for (int f = wave; f < nfb; f += WAVES) {
    const int base = shift + 1 + f * a.hop;  // tile index of the frame's first sample
    auto x = [&](int i) -> float {
      float cur = load_sample<R, F32IN>(tile, base + i, a.divisor, a.clip);
      if (alpha != 0.0f) {
        float prev = load_sample<R, F32IN>(tile, base + i - 1, a.divisor, a.clip);
        // reference: frame -= pre_emphasis * prev  (separate fp32 multiply and subtract)
        cur = __fsub_rn(cur, __fmul_rn(alpha, prev));
      }
      return cur;
    };
    frame_fft_mag<R>(x, a.hann, tw256, tw512, buf, mg, lane);
    // ---- banded mel filter + log tail, one lane per band
    if (lane < a.n_mel) {
      const int st = fmeta[lane], ln = fmeta[64 + lane], wo = fmeta[128 + lane];
      float acc = 0.0f;
      for (int i = 0; i < ln; ++i) acc = fmaf(fw[wo + i], mg[st + i], acc);
      acc += a.bias[lane];
      acc = fmaxf(acc, a.floor_v);
      mel_tile[f * 40 + lane] = (logf(acc) + a.log_off) * a.scale;
    }
    wave_sync();
  }
  __syncthreads();
  // ---- coalesced store of the mel tile
  float *dst = a.mel + (a.frame_offs[u] + f0) * (int64_t)a.n_mel;
  const int n_out = nfb * a.n_mel;
  for (int i = tid; i < n_out; i += 256) dst[i] = mel_tile[i];
}

// STFT magnitude of explicit frames [n][512] -> [n][257]; one wave per frame.
template <typename R>
__global__ __launch_bounds__(256) void stft_mag_kernel(logmel_args a) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  size_t off = 0;
  cplx<R> *tw256 = (cplx<R> *)(smem + off); off += 256 * sizeof(cplx<R>);
  cplx<R> *tw512 = (cplx<R> *)(smem + off); off += 256 * sizeof(cplx<R>);
  cplx<R> *fbuf = (cplx<R> *)(smem + off); off += WAVES * 256 * sizeof(cplx<R>);
  float *mag = (float *)(smem + off);
  for (int i = tid; i < 256; i += 256) {
    tw256[i] = {(R)a.tw256[2 * i], (R)a.tw256[2 * i + 1]};
    tw512[i] = {(R)a.tw512[2 * i], (R)a.tw512[2 * i + 1]};
  }
  __syncthreads();
  const int64_t f = (int64_t)blockIdx.x * WAVES + wave;
  if (f >= a.n_frames_direct) return;
  const float *src = a.frames + f * WIN;
  auto x = [&](int i) -> float { return src[i]; };
  float *mg = mag + wave * 260;
  frame_fft_mag<R>(x, a.hann, tw256, tw512, fbuf + wave * 256, mg, lane);
  float *dst = a.mag_out + f * NB;
  for (int k = lane; k < NB; k += 64) dst[k] = mg[k];
}

template <typename R>
static size_t logmel_smem(int hop, int total_taps, bool f32in) {
  size_t off = 0;
  off += 256 * sizeof(cplx<R>) * 2;
  off += WAVES * 256 * sizeof(cplx<R>);
  off += WAVES * 260 * sizeof(float);
  off += FPB * 40 * sizeof(float);
  off += ((total_taps + 3) & ~3) * sizeof(float);
  off += 3 * 64 * sizeof(int);
  off += (size_t)(WIN + (FPB - 1) * hop + 1 + 16) * (f32in ? 4 : 2);
  return (off + 15) & ~size_t(15);
}

static void fill_filter_args(logmel_args &a, const ww_model *m) {
  const ww_filter_dev &f = m->filt;
  a.start = f.start; a.len = f.len; a.woff = f.woff; a.w = f.w; a.bias = f.bias;
  a.n_mel = f.n_mel; a.total_taps = f.total_taps;
  a.floor_v = f.floor_v; a.log_off = f.log_off; a.scale = f.scale;
  a.hann = f.hann; a.tw256 = f.tw256; a.tw512 = f.tw512;
}

int ww_k_logmel(ww_ctx *ctx, const ww_model *m, const int16_t *d_pcm, const float *d_f32, const int64_t *d_sample_offs,
                const int64_t *d_frame_offs, int n_utt, int64_t total_frames, int64_t max_frames_per_utt,
                const ww_frontend_params *fp, float *d_mel) {
  if (n_utt <= 0 || total_frames <= 0 || max_frames_per_utt <= 0) return WW_OK;
  if (m->filt.n_mel > 40 || m->filt.n_bins != NB) return ww_fail(ctx, WW_EINVAL, "front end expects 257 bins and <= 40 bands");
  if (fp->hop <= 0 || fp->hop > 512) return ww_fail(ctx, WW_EINVAL, "hop %d out of range (1..512)", fp->hop);
  if (n_utt > 65535) return ww_fail(ctx, WW_EINVAL, "at most 65535 utterances per call (got %d)", n_utt);
  logmel_args a = {};
  a.pcm = d_pcm; a.f32 = d_f32; a.sample_offs = d_sample_offs; a.frame_offs = d_frame_offs;
  a.n_utt = n_utt; a.hop = fp->hop; a.divisor = fp->pcm_divisor; a.clip = fp->clip; a.preemph = fp->pre_emphasis;
  a.mel = d_mel;
  fill_filter_args(a, m);
  const bool f32in = d_f32 != nullptr;
  dim3 grid((unsigned)((max_frames_per_utt + FPB - 1) / FPB), (unsigned)n_utt);
  ww_launch_scope scope(ctx, fp->precise ? "logmel_kernel<f64>" : "logmel_kernel<f32>");
  if (fp->precise) {
    size_t sm = logmel_smem<double>(fp->hop, a.total_taps, f32in);
    if (f32in) hipLaunchKernelGGL((logmel_kernel<double, true>), grid, dim3(256), sm, ctx->stream, a);
    else hipLaunchKernelGGL((logmel_kernel<double, false>), grid, dim3(256), sm, ctx->stream, a);
  } else {
    size_t sm = logmel_smem<float>(fp->hop, a.total_taps, f32in);
    if (f32in) hipLaunchKernelGGL((logmel_kernel<float, true>), grid, dim3(256), sm, ctx->stream, a);
    else hipLaunchKernelGGL((logmel_kernel<float, false>), grid, dim3(256), sm, ctx->stream, a);
  }
  WW_HIP(ctx, hipGetLastError());
  return WW_OK;
}

int ww_k_stft_mag(ww_ctx *ctx, const ww_model *m, const float *d_frames, int64_t n, int precise, float *d_mag) {
  if (n <= 0) return WW_OK;
  logmel_args a = {};
  fill_filter_args(a, m);
  a.frames = d_frames; a.mag_out = d_mag; a.n_frames_direct = n;
  dim3 grid((unsigned)((n + WAVES - 1) / WAVES));
  ww_launch_scope scope(ctx, "stft_mag_kernel");
  if (precise) {
    size_t sm = 256 * sizeof(cplx<double>) * 2 + WAVES * 256 * sizeof(cplx<double>) + WAVES * 260 * sizeof(float);
    hipLaunchKernelGGL((stft_mag_kernel<double>), grid, dim3(256), sm, ctx->stream, a);
  } else {
    size_t sm = 256 * sizeof(cplx<float>) * 2 + WAVES * 256 * sizeof(cplx<float>) + WAVES * 260 * sizeof(float);
    hipLaunchKernelGGL((stft_mag_kernel<float>), grid, dim3(256), sm, ctx->stream, a);
  }
  WW_HIP(ctx, hipGetLastError());
  return WW_OK;
}
