// Streaming mode: S concurrent 16 kHz streams advanced 20 ms per tick (BASELINE config 5).
//
// Device-resident replacement for the per-stream state of WakewordTrigger
// (spokestack/wakeword/tflite.py:92-108): sample ring, mel frame window, pre-emphasis carry.
// Per tick and stream the reference runs a 320-iteration Python loop that emits a frame
// whenever 512 samples are buffered and then drops the oldest 160 (tflite.py:163-168); with a
// ring that holds f <= 511 samples before the tick, that is exactly
//     n_frames = (f + 320 >= 512) ? (f + 320 - 512) / 160 + 1 : 0      (0, 1 or 2)
// frames at offsets 0, 160 of the concatenation [ring | new samples], and f' = f + 320 -
// 160 * n_frames samples kept.  The host mirrors f per stream (pure integer bookkeeping) so it
// knows the number of posteriors a tick will produce and can size the launch exactly; all
// sample and mel data stay in HBM.
//
//   stream_frontend_kernel  one workgroup (2 waves) per stream: read the tick's samples and control words from
//                           pinned host memory, normalise + pre-emphasise the 320 new
//                           samples, FFT + mel for each new frame (one wave per frame) when
//                           the stream's is_speech bit is set, keep the ring tail.
//   The mel window of a stream is a MIRRORED ring of R = T + 1 slots (2R rows): row k of the stream's history is written at
//   k % R and at k % R + R, so the latest T rows up to row k are always the contiguous block [(k + 2) % R, +T) that the
//   model kernels' window descriptors want - two 160-byte stores per new frame instead of moving the whole 24-29 KB window
//   every tick (round 1's ping-pong shift).  The one spare slot is what lets BOTH windows of a two-frame tick be evaluated
//   after both rows are in place: the window that ends at row k leaves out exactly the slot of row k + 1.
//   then the regular CRNN / Wavenet kernels run on the compacted list of new windows.
#include "common.h"
#include "fft_device.h"

#include <sched.h>
#include <time.h>

#define ST_RING WW_ST_RING  // 511 + 320 rounded up
#define ST_WL_BYTES (((WW_MEL_TAPS * 64 * 4 + 2047) / 2048) * 2048)  // the mel weights in LDS, padded to whole rounds of 128 x 16 bytes

struct ww_streams {
  ww_ctx *ctx = nullptr;
  const ww_model *model = nullptr;
  int S = 0;
  ww_frontend_params fp = {};
  int T = 0, F = 0, NO = 0, HR = 0;  // HR = history rows per stream = 2 (T + 1) (mirrored ring)
  // device state
  float *ring = nullptr;      // [2][S][ST_RING] (the one-launch tick ping-pongs between the two copies by a stream's state parity; the
  float *hist = nullptr;      //  two-launch forms use the first) | [S][2 (T + 1)][F], mirrored ring
  float *prev = nullptr;      // [2][S] pre-emphasis carry (raw previous sample), likewise
  // per-tick device buffers
  int16_t *d_frames = nullptr;   // [S][320]
  int32_t *d_ctl = nullptr;      // [S][4]: fill, n_frames, flags, pos (history rows written so far, mod T + 1)
  int64_t *d_win_row = nullptr;  // [2S]
  int32_t *d_win_valid = nullptr;
  void *ws = nullptr;
  size_t ws_bytes = 0;           // capacity of ws (grown by ww_stream_step when the model's options ask for more)
  // host mirrors (pinned)
  int32_t *h_ctl = nullptr;
  int64_t *h_win_row = nullptr;
  int32_t *h_win_valid = nullptr;
  float *h_out = nullptr;
  float *h_out_dev = nullptr;   // the device's address of h_out
  int16_t *h_frames = nullptr;
  // the four per-tick inputs live in ONE pinned block (frames | win_row | ctl | win_valid) that the front-end kernel
  // reads over the bus itself - no copy-engine operation on a tick's path (four back-to-back host-to-device copies
  // cost 20 us of a 107 us tick); d_pack mirrors the layout and holds the window descriptors the kernel copies over.
  // The block exists TWICE (round 5): the one-launch tick alternates between the copies - a polled tick returns as soon as its
  // posteriors are in, while workgroups that owe none (a stream without a window only advances its ring) may not have read
  // their samples yet; by the time a copy is written again the kernel of the tick in between has started, i.e. this one ended
  char *h_pack = nullptr, *d_pack = nullptr, *h_pack_dev = nullptr;  // h_pack_dev: the device's address of h_pack
  size_t pack_bytes = 0;  // of ONE copy
  // a tick's posteriors as {value, tick number} pairs the model kernels store straight into page-locked memory (ww_tick_tag):
  // ww_stream_step polls them instead of waiting for the runtime's completion signal
  unsigned long long *h_tag = nullptr, *h_tag_dev = nullptr;  // [2 S]
  unsigned seq = 0;             // tick number (never 0 in a tag)
  unsigned flip = 0;            // which copy of the input block the tick at hand uses
  bool one_launch = false;      // incremental CRNN: front end inside the model kernel's workgroups (crnn_stream_kernel<1 | 2>)
  bool broken = false;          // a tick failed half way (ww_stream_step): no further ticks
  bool poll = false;            // wait by polling the tags (a context that owns its stream; WW_STREAM_SYNC_WAIT turns it off)
  std::vector<int> par;         // a stream's state parity (one-launch form)
  std::vector<int> expect;      // tag slots this tick's posteriors arrive in
  std::vector<int> fill, pos;
  // incremental CRNN (crnn.hip, crnn_stream_kernel): per-stream ring of projected interior rows, the row of an all-zero
  // field every slot holds after a reset, and the number of mel rows since the reset modulo the ring size
  float *gxc = nullptr, *gx_zero = nullptr;  // [S][WW_STREAM_GXC][192], [192]
  int32_t *h_win_aux = nullptr, *d_win_aux = nullptr;
  std::vector<int> rowq;
  bool incremental = false;
  // host timeline of ww_stream_step (ww_stream_timeline): nanoseconds per phase summed over the ticks since the last reset
  uint64_t tl_ns[WW_STREAM_TL_PHASES] = {0};
  int64_t tl_ticks = 0;
  std::vector<uint8_t> stage_flags;  // ww_stream_step_trigger: bit 0 = is_speech, bit 1 = is_active per stream
};

static inline uint64_t st_now_ns() {
  timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (uint64_t)ts.tv_sec * 1000000000ull + (uint64_t)ts.tv_nsec;
}

struct stream_fe_args {
  const int16_t *frames;   // pinned host memory, read over the bus (640 B per stream and tick)
  const int32_t *ctl;      // pinned host memory
  const int64_t *h_row;    // pinned host: this tick's window rows / valid counts, copied to d_row / d_valid here
  const int32_t *h_valid;  //   for the model kernels that follow on the stream
  const int32_t *h_aux;    //   (+ the cache slots of the incremental CRNN kernel)
  int64_t *d_row;
  int32_t *d_valid, *d_aux;
  int nw, S;
  float *ring;
  float *hist;
  float *prev;
  int T, F, HR;
  float divisor;
  int clip;
  float preemph;
  int hop;
  const int *start;
  const float *wpad, *bias;
  int n_mel;
  float floor_v, log_off, scale;
  const double *hann, *tw256, *tw512;
};

template <typename R>
__global__ __launch_bounds__(128) void stream_frontend_kernel(stream_fe_args a) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int s = blockIdx.x;
  // the tick's 320 samples (40 x 16 bytes) are requested from pinned host memory BEFORE the control words are looked at:
  // one trip over the bus instead of two in a row (control words -> branch -> samples)
  static_assert(WW_CHUNK * 2 == 40 * 16, "a tick is 40 sixteen-byte pieces");
  uint4 raw = make_uint4(0u, 0u, 0u, 0u);
  if (tid < 40) raw = ((const uint4 *)(a.frames + (size_t)s * WW_CHUNK))[tid];
  const int4 cw = ((const int4 *)a.ctl)[s];
  const int fill = cw.x, n_frames = cw.y, flags = cw.z, pos = cw.w;
  const bool speech = flags & 1, skip = flags & 2;

  size_t off = 0;
  cplx<R> *fbuf = (cplx<R> *)(smem + off); off += 2 * FFT_LD * sizeof(cplx<R>);
  float *mag = (float *)(smem + off); off += 2 * 260 * sizeof(float);
  float *wl = (float *)(smem + off); off += ST_WL_BYTES;  // (rounded up to whole 128-thread store rounds)
  float *x = (float *)(smem + off); off += ST_RING * sizeof(float);  // [ST_RING]
  short *xs = (short *)(smem + off);  // [WW_CHUNK] raw samples of the tick

  // ---- this tick's window descriptors: host-pinned -> device arrays (read by the model kernels that follow).  A tick has at
  // most two windows per stream, i.e. one descriptor per thread of the first workgroups: requested here with everything else
  // that crosses the bus (unconditionally, from a clamped index), stored below once the loads behind them are out as well -
  // copied in place, the first workgroups waited for the bus before they asked for anything else
  const int di = s * 128 + tid, dic = di < a.nw ? di : 0;
  const int64_t h_row0 = a.h_row[dic];
  const int h_valid0 = a.h_valid[dic], h_aux0 = a.h_aux[dic];
  // device-side inputs that do not depend on the control words are requested before the branch on them, i.e. while
  // the control words and samples are still crossing the bus: mel weights, the whole sample ring (fill <= 511 of its
  // 832 slots are meaningful; the rest is never read), the pre-emphasis carry, the transform's constants
  constexpr int WLQ = ST_WL_BYTES / 16 / 128;
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  f32x4 wlq[WLQ];
#pragma unroll
  for (int q = 0; q < WLQ; ++q) {
    const int i = tid + q * 128;
    wlq[q] = ((const f32x4 *)a.wpad)[i < WW_MEL_TAPS * 64 / 4 ? i : 0];
  }
  const int mel_st = lane < a.n_mel ? a.start[lane] : 0;      // (the mel stage's two per-band scalars: asked for here, not
  const float mel_bias = lane < a.n_mel ? a.bias[lane] : 0.0f;  // behind the transform where they are used)
  float *ring = a.ring + (size_t)s * ST_RING;
  const float4 ringq = ((const float4 *)ring)[tid];  // samples 4 tid .. 4 tid + 3 (128 threads x 4 = 512 >= fill)
  const float carry = a.prev[s];
  fft_consts<R> fc;
  fft_load_consts<R>(fc, lane, a.hann, a.tw256, a.tw512);
  // parked in LDS unconditionally and BEFORE the branch on the control words: behind it, and under a per-piece condition, the
  // compiler moved each load down to its store - five round trips to L2 in a row (the wl region is padded to whole rounds)
#pragma unroll
  for (int q = 0; q < WLQ; ++q) ((f32x4 *)wl)[tid + q * 128] = wlq[q];
  if (di < a.nw) {
    a.d_row[di] = h_row0;
    a.d_valid[di] = h_valid0;
    a.d_aux[di] = h_aux0;
  }
  for (int i = di + a.S * 128; i < a.nw; i += a.S * 128) {  // (more than 128 windows per stream and tick: not a shape the host produces)
    a.d_row[i] = a.h_row[i];
    a.d_valid[i] = a.h_valid[i];
    a.d_aux[i] = a.h_aux[i];
  }
  if (skip) return;  // context.is_active: the frame is not sampled at all (tflite.py:139-140)

  // ---- [ring | new samples] in LDS
  ((float4 *)x)[tid] = ringq;
  if (tid < 40) ((uint4 *)xs)[tid] = raw;
  __syncthreads();
  for (int i = tid; i < WW_CHUNK; i += 128) {
    float v = __fdiv_rn((float)xs[i], a.divisor);
    if (a.clip) v = fminf(fmaxf(v, -1.0f), 1.0f);
    float p;
    if (i == 0) {
      p = carry;
    } else {
      p = __fdiv_rn((float)xs[i - 1], a.divisor);
      if (a.clip) p = fminf(fmaxf(p, -1.0f), 1.0f);
    }
    x[fill + i] = (a.preemph != 0.0f) ? __fsub_rn(v, __fmul_rn(a.preemph, p)) : v;
  }
  __syncthreads();
  if (tid == 0) {
    float v = __fdiv_rn((float)xs[WW_CHUNK - 1], a.divisor);
    if (a.clip) v = fminf(fmaxf(v, -1.0f), 1.0f);
    a.prev[s] = v;  // tflite.py:156-158: carry is the un-emphasised last sample
  }
  // ---- new frames (wave k handles frame k); only analysed while is_speech (tflite.py:166)
  if (speech && wave < n_frames) {
    const float *src = x + wave * a.hop;
    auto x2 = [&](int n) -> float2 { return make_float2(src[2 * n], src[2 * n + 1]); };
    float *mg = mag + wave * 260;
    frame_fft_mag<R>(x2, fc, fbuf + wave * FFT_LD, mg, lane);
    const float mv = mel_band(mg, wl, mel_st, mel_bias, a.floor_v, a.log_off, a.scale, lane);
    if (lane < a.n_mel) {
      const int slots = a.T + 1;
      int p = pos + wave;  // mirrored ring: the row goes to p % slots and p % slots + slots
      p = p >= slots ? p - slots : p;
      float *h = a.hist + ((size_t)s * a.HR + p) * a.F + lane;
      h[0] = mv;
      h[(size_t)slots * a.F] = mv;
    }
  }
  __syncthreads();
  // ---- keep the ring tail
  const int keep = fill + WW_CHUNK - n_frames * a.hop;
  for (int i = tid; i < keep; i += 128) ring[i] = x[n_frames * a.hop + i];
}

__global__ void stream_reset_kernel(float *hist, const int32_t *ids, int S, int HR, int F, float *gxc, const float *gx_zero) {
  const int b = blockIdx.x;
  const int s = ids ? ids[b] : b;
  if (s < 0 || s >= S) return;
  for (int i = threadIdx.x; i < HR * F; i += blockDim.x) hist[(size_t)s * HR * F + i] = 0.f;
  if (gxc) {  // every cached row = the row of an all-zero field (what lies in front of the stream's first mel rows)
    float *c = gxc + (size_t)s * WW_STREAM_GXC * 192;
    for (int i = threadIdx.x; i < WW_STREAM_GXC * 192; i += blockDim.x) c[i] = gx_zero[i % 192];
  }
}

extern "C" {

int ww_stream_destroy(ww_streams *st) {
  WW_GUARD_BEGIN
  if (!st) return WW_OK;
  ww_device_scope dev_scope(st->ctx->device);
  hipStreamSynchronize(st->ctx->stream);
  void *dev[] = {st->ring, st->hist, st->prev, st->d_pack, st->ws, st->gxc, st->gx_zero};
  for (void *p : dev)
    if (p) hipFree(p);
  void *host[] = {st->h_pack, st->h_out, st->h_tag};
  for (void *p : host)
    if (p) hipHostFree(p);
  delete st;
  return WW_OK;
  WW_GUARD_END(nullptr)
}

int ww_stream_create(ww_ctx *ctx, const ww_model *model, int32_t S, const ww_frontend_params *fp, uint32_t flags,
                     ww_streams **out) {
  WW_GUARD_BEGIN
  if (!ctx || !model || !fp || !out) return ww_fail(ctx, WW_EINVAL, "NULL argument");
  *out = nullptr;
  if (flags & ~(uint32_t)(WW_STREAM_FULL_RECOMPUTE | WW_STREAM_TWO_LAUNCH | WW_STREAM_SYNC_WAIT))
    return ww_fail(ctx, WW_EINVAL, "unknown stream flags 0x%x", flags);
  if (S <= 0 || S > 65535) return ww_fail(ctx, WW_EINVAL, "stream count %d out of range (1..65535)", S);
  if (fp->hop != 160) return ww_fail(ctx, WW_EINVAL, "streaming mode supports hop 160 (10 ms @ 16 kHz) only, got %d", fp->hop);
  if (!(fp->pcm_divisor > 0.f)) return ww_fail(ctx, WW_EINVAL, "pcm_divisor must be positive");
  WW_ON_DEVICE(ctx, dev_scope);  // the caller's current device is left as it was
  ww_streams *st = new ww_streams();
  ww_scoped<ww_streams, ww_stream_destroy> own(st);  // (freed on every early return below)
  st->ctx = ctx; st->model = model; st->S = S; st->fp = *fp;
  st->T = model->info.window; st->F = model->info.n_mel; st->NO = model->info.n_out; st->HR = 2 * (st->T + 1);
  const size_t hist_elems = (size_t)S * st->HR * st->F;
  // CRNN, standard geometry: three positions per new window instead of nineteen (crnn_stream_kernel, fp32 contractions - also
  // for a model in split-bf16 mode: a seventh of the products in fp32 is both faster and closer).
  // WW_STREAM_FULL_RECOMPUTE keeps the per-window kernels (every window recomputed from its mel rows).
  st->incremental = ww_crnn_stream_capable(model) && !(flags & WW_STREAM_FULL_RECOMPUTE);
  // ONE launch per tick: the incremental CRNN at any size (tools/stream_forms.py: 7 % faster from 128 to 1,024 streams), the
  // Wavenet while a tick's windows stay within its twelve-wave form (ww_wave_tick_capable: up to 128 streams)
  st->one_launch = !(flags & WW_STREAM_TWO_LAUNCH) &&
                   ((st->incremental && model->filt.n_mel == 40) || ww_wave_tick_capable(model, S));
  // a borrowed stream is the caller's: when the call returns everything enqueued on it has completed, as before
  st->poll = ctx->own_stream && !(flags & WW_STREAM_SYNC_WAIT);
  // model scratch of the per-window kernels (explicit window rows: never the sliding form); the incremental CRNN needs none
  size_t ws_bytes = st->incremental ? 256
                    : (model->kind == WW_KIND_CRNN ? ww_crnn_workspace(model, 2 * S, false) : ww_wave_workspace(model, 2 * S));
  bool ok = hipMalloc((void **)&st->ring, (size_t)2 * S * ST_RING * 4) == hipSuccess &&
            hipMalloc((void **)&st->hist, hist_elems * 4) == hipSuccess &&
            hipMalloc((void **)&st->prev, (size_t)2 * S * 4) == hipSuccess &&
            hipMalloc(&st->ws, ws_bytes) == hipSuccess && ((st->ws_bytes = ws_bytes), true) &&
            hipHostMalloc((void **)&st->h_out, (size_t)2 * S * st->NO * 4) == hipSuccess &&
            hipHostMalloc((void **)&st->h_tag, (size_t)2 * S * 8) == hipSuccess;
  {
    const size_t o_frames = 0, o_row = o_frames + (size_t)S * WW_CHUNK * 2, o_ctl = o_row + (size_t)2 * S * 8,
                 o_valid = o_ctl + (size_t)S * 16, o_aux = o_valid + (size_t)2 * S * 4;
    st->pack_bytes = o_aux + (size_t)2 * S * 4;
    st->pack_bytes = (st->pack_bytes + 255) & ~(size_t)255;
    ok = ok && hipMalloc((void **)&st->d_pack, st->pack_bytes) == hipSuccess &&
         hipHostMalloc((void **)&st->h_pack, 2 * st->pack_bytes) == hipSuccess;
    if (ok) {
      st->h_frames = (int16_t *)(st->h_pack + o_frames); st->d_frames = (int16_t *)(st->d_pack + o_frames);
      st->h_win_row = (int64_t *)(st->h_pack + o_row);   st->d_win_row = (int64_t *)(st->d_pack + o_row);
      st->h_ctl = (int32_t *)(st->h_pack + o_ctl);       st->d_ctl = (int32_t *)(st->d_pack + o_ctl);
      st->h_win_valid = (int32_t *)(st->h_pack + o_valid); st->d_win_valid = (int32_t *)(st->d_pack + o_valid);
      st->h_win_aux = (int32_t *)(st->h_pack + o_aux);     st->d_win_aux = (int32_t *)(st->d_pack + o_aux);
    }
  }
  if (!ok) {
    return ww_fail(ctx, WW_ENOMEM, "cannot allocate state for %d streams", S);
  }
  memset(st->h_tag, 0, (size_t)2 * S * 8);
  if (hipHostGetDevicePointer((void **)&st->h_out_dev, st->h_out, 0) != hipSuccess ||
      hipHostGetDevicePointer((void **)&st->h_tag_dev, st->h_tag, 0) != hipSuccess ||
      hipHostGetDevicePointer((void **)&st->h_pack_dev, st->h_pack, 0) != hipSuccess) {
    return ww_fail(ctx, WW_EHIP, "pinned staging buffers are not visible to the device");
  }
  hipMemsetAsync(st->ring, 0, (size_t)2 * S * ST_RING * 4, ctx->stream);
  hipMemsetAsync(st->hist, 0, hist_elems * 4, ctx->stream);
  hipMemsetAsync(st->prev, 0, (size_t)2 * S * 4, ctx->stream);
  hipMemsetAsync(st->d_pack, 0, st->pack_bytes, ctx->stream);
  memset(st->h_pack, 0, 2 * st->pack_bytes);
  WW_HIP(ctx, hipStreamSynchronize(ctx->stream));
  st->fill.assign(S, 0);
  st->pos.assign(S, 0);
  st->rowq.assign(S, 0);
  st->par.assign(S, 0);
  st->expect.reserve((size_t)2 * S);
  if (st->incremental) {
    if (hipMalloc((void **)&st->gxc, (size_t)S * WW_STREAM_GXC * 192 * 4) != hipSuccess ||
        hipMalloc((void **)&st->gx_zero, 192 * 4) != hipSuccess) {
      return ww_fail(ctx, WW_ENOMEM, "cannot allocate the projected-row cache of %d streams", S);
    }
    // the row of an all-zero field: position 17 of one window over the (all-zero) history of stream 0, which the kernel
    // stores into slot (0 + 128) % ring of stream 0's cache
    hipMemsetAsync(st->gxc, 0, (size_t)WW_STREAM_GXC * 192 * 4, ctx->stream);
    int rc = ww_k_crnn_stream_forward(ctx, model, st->hist, (int64_t)S * st->HR, st->d_win_row, st->d_win_valid, st->d_win_aux,
                                      st->gxc, 1, st->h_out_dev);  // d_pack is zeroed: window row 0, aux 0 (valid 0 = all-zero window)
    if (rc) {
      return rc;
    }
    hipMemcpyAsync(st->gx_zero, st->gxc + (size_t)(128 % WW_STREAM_GXC) * 192, 192 * 4, hipMemcpyDeviceToDevice, ctx->stream);
    hipLaunchKernelGGL(stream_reset_kernel, dim3(S), dim3(256), 0, ctx->stream, st->hist, (const int32_t *)nullptr, S, st->HR, st->F,
                       st->gxc, (const float *)st->gx_zero);
    if (hipStreamSynchronize(ctx->stream) != hipSuccess || hipGetLastError() != hipSuccess) {
      return ww_fail(ctx, WW_EHIP, "streaming CRNN set-up failed");
    }
  }
  *out = own.release();
  return WW_OK;
  WW_GUARD_END(ctx)
}

int ww_stream_reset(ww_streams *st, const int32_t *ids, int32_t n) {
  WW_GUARD_BEGIN
  if (!st) return WW_EINVAL;
  ww_ctx *ctx = st->ctx;
  if (ids && n < 0) return ww_fail(ctx, WW_EINVAL, "negative id count");
  const int count = ids ? n : st->S;
  if (count == 0) return WW_OK;
  WW_ON_DEVICE(ctx, dev_scope);  // the caller's current device is left as it was
  int32_t *d_ids = nullptr;
  if (ids) {
    for (int i = 0; i < n; ++i)
      if (ids[i] < 0 || ids[i] >= st->S) return ww_fail(ctx, WW_EINVAL, "stream id %d out of range", ids[i]);
    // reuse the window-valid buffer (2S ints) as id staging
    if (n > 2 * st->S) return ww_fail(ctx, WW_EINVAL, "more ids than streams");
    memcpy(st->h_win_valid, ids, (size_t)n * 4);
    WW_HIP(ctx, hipMemcpyAsync(st->d_win_valid, st->h_win_valid, (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
    d_ids = st->d_win_valid;
  }
  hipLaunchKernelGGL(stream_reset_kernel, dim3(count), dim3(256), 0, ctx->stream, st->hist, (const int32_t *)d_ids, st->S, st->HR, st->F,
                     st->gxc, (const float *)st->gx_zero);
  WW_HIP(ctx, hipGetLastError());
  WW_HIP(ctx, hipStreamSynchronize(ctx->stream));
  // WakewordTrigger.reset (tflite.py:241-246): sample window emptied, frame window zeroed;
  // _prev_sample is NOT reset by the reference and is not reset here.
  for (int i = 0; i < count; ++i) {
    const int s = ids ? ids[i] : i;
    st->fill[s] = 0;
    st->pos[s] = 0;
    st->rowq[s] = 0;
  }
  return WW_OK;
  WW_GUARD_END(st ? st->ctx : nullptr)
}

// Wait for a tick's posteriors by polling their {value, tick number} pairs in page-locked memory.  The runtime is asked only
// now and then whether the stream has drained (a kernel that died would otherwise leave the host spinning for ever).
// A tick that has not delivered after WW_TICK_TIMEOUT_S seconds (a kernel that neither finishes nor fails: a hung GPU) is an error
// as well - the caller gets WW_EHIP and a bank that refuses further ticks instead of a host thread that spins for ever.
#ifndef WW_TICK_TIMEOUT_S
#define WW_TICK_TIMEOUT_S 10
#endif
static int st_poll_tags(ww_streams *st) {
  ww_ctx *ctx = st->ctx;
  const size_t n = st->expect.size();
  const unsigned seq = st->seq;
  size_t i = 0;
  unsigned spins = 0;
  uint64_t t_slow = 0;  // when the wait left the fast path (first runtime query)
  while (i < n) {
    const unsigned long long v = __atomic_load_n(st->h_tag + st->expect[i], __ATOMIC_ACQUIRE);
    if ((unsigned)(v >> 32) == seq) {
      ++i;
      continue;
    }
    __builtin_ia32_pause();
    if (spins > (1u << 18)) sched_yield();  // (a tick that is milliseconds late - a GPU busy elsewhere - stops pinning the core)
    if ((++spins & 0x3fffu) == 0) {
      const hipError_t q = hipStreamQuery(ctx->stream);
      if (q == hipErrorNotReady) {
        const uint64_t now = st_now_ns();
        if (!t_slow) t_slow = now;
        if (now - t_slow > (uint64_t)WW_TICK_TIMEOUT_S * 1000000000ull)
          return ww_fail(ctx, WW_EHIP, "streaming tick %u did not complete within %d s (posterior slot %d still missing)", seq, WW_TICK_TIMEOUT_S,
                         st->expect[i]);
        continue;
      }
      if (q != hipSuccess) return ww_fail(ctx, WW_EHIP, "streaming tick failed: %s", hipGetErrorString(q));
      for (size_t r = i; r < n; ++r)  // the stream has drained: what will ever arrive has
        if ((unsigned)(__atomic_load_n(st->h_tag + st->expect[r], __ATOMIC_ACQUIRE) >> 32) != seq)
          return ww_fail(ctx, WW_EHIP, "streaming tick %u completed without delivering posterior slot %d", seq, st->expect[r]);
      break;
    }
  }
  return WW_OK;
}

static int stream_step_impl(ww_streams *st, const int16_t *frames, const uint8_t *is_speech, float *post, int32_t *n_post, bool *mutated) {
  ww_ctx *ctx = st->ctx;
  if (!frames || !is_speech || !post || !n_post) return ww_fail(ctx, WW_EINVAL, "NULL argument");
  WW_ON_DEVICE(ctx, dev_scope);  // the caller's current device is left as it was
  *mutated = true;  // from here on the host's mirrors of the streams' state advance
  const int S = st->S, hop = st->fp.hop, R = st->T + 1;
  uint64_t tl[WW_STREAM_TL_PHASES + 1];
  tl[0] = st_now_ns();
  if (++st->seq == 0) st->seq = 1;  // (a tag of tick 0 is "never written")
  // which copy of the page-locked input block this tick uses (the one-launch form alternates, see ww_streams)
  st->flip ^= 1u;  // (a bit of its own: the tick number skips 0 when it wraps)
  const size_t cp = st->one_launch ? (size_t)st->flip * st->pack_bytes : 0;
  int16_t *h_frames = (int16_t *)((char *)st->h_frames + cp);
  int32_t *h_ctl = (int32_t *)((char *)st->h_ctl + cp);
  st->expect.clear();
  int nw = 0;
  for (int s = 0; s < S; ++s) {
    const int flags = is_speech[s] & 3;
    int nf = 0;
    if (!(flags & 2)) {
      const int tot = st->fill[s] + WW_CHUNK;
      nf = tot >= WW_FFT_WINDOW ? (tot - WW_FFT_WINDOW) / hop + 1 : 0;
    }
    const int np = (flags & 1) && !(flags & 2) ? nf : 0;
    n_post[s] = np;
    h_ctl[s * 4 + 0] = st->fill[s];
    h_ctl[s * 4 + 1] = nf;
    if (st->one_launch) {
      // workgroup 2 s + k of the tick's one kernel finds everything else out by itself
      h_ctl[s * 4 + 2] = flags | (st->par[s] << 2);
      h_ctl[s * 4 + 3] = st->pos[s] | (st->rowq[s] << 16);
      for (int k = 0; k < np; ++k) st->expect.push_back(2 * s + k);
      if (!(flags & 2)) st->par[s] ^= 1;
    } else {
      h_ctl[s * 4 + 2] = flags;
      h_ctl[s * 4 + 3] = st->pos[s];
      for (int k = 0; k < np; ++k) {
        // the T rows that end at this tick's new row k are the contiguous block that starts at (pos + k + 2) % R
        st->h_win_row[nw + k] = (int64_t)s * st->HR + (st->pos[s] + k + 2) % R;
        st->h_win_valid[nw + k] = st->T;
        st->h_win_aux[nw + k] = s * WW_STREAM_GXC + (st->rowq[s] + k + 1) % WW_STREAM_GXC;  // rows since the reset incl. this window's newest
        st->expect.push_back(nw + k);
      }
    }
    nw += np;
    if (!(flags & 2)) st->fill[s] = st->fill[s] + WW_CHUNK - nf * hop;
    st->pos[s] = (st->pos[s] + np) % R;
    st->rowq[s] = (st->rowq[s] + np) % WW_STREAM_GXC;
  }
  tl[1] = st_now_ns();
  memcpy(h_frames, frames, (size_t)S * WW_CHUNK * 2);
  tl[2] = st_now_ns();
  const ww_model *m = st->model;
  const ww_filter_dev &f = m->filt;
  // posterior element: width-1 head -> [0]; width-2 head -> [1]  
  const int pidx = st->NO == 1 ? 0 : 1;
  ww_tick_tag tag = {st->h_tag_dev, st->seq, pidx};
  bool tagged = false;
  if (st->one_launch) {
    // ---- ONE launch: front end + model, workgroup 2 s + k = window k of stream s (crnn.hip: crnn_stream_kernel<FE>; wavenet.hip)
    ww_tick_fe fe = {};
    fe.frames = (const int16_t *)(st->h_pack_dev + ((char *)h_frames - st->h_pack));
    fe.ctl = (const int32_t *)(st->h_pack_dev + ((char *)h_ctl - st->h_pack));
    fe.ring = st->ring; fe.prev = st->prev; fe.hist = st->hist;
    fe.S = S; fe.HR = st->HR;
    fe.divisor = st->fp.pcm_divisor; fe.clip = st->fp.clip; fe.preemph = st->fp.pre_emphasis; fe.hop = hop;
    int rc = m->kind == WW_KIND_CRNN ? ww_k_crnn_tick(ctx, m, fe, st->fp.precise, st->gxc, tag) : ww_k_wave_tick(ctx, m, fe, st->fp.precise, tag);
    if (rc) return rc;
    tagged = true;
    tl[3] = st_now_ns();
  } else {
    stream_fe_args a = {};
    // no copy engine on the tick's path: the kernel reads the pinned staging block itself
    a.frames = (const int16_t *)(st->h_pack_dev + ((char *)st->h_frames - st->h_pack));
    a.ctl = (const int32_t *)(st->h_pack_dev + ((char *)st->h_ctl - st->h_pack));
    a.h_row = (const int64_t *)(st->h_pack_dev + ((char *)st->h_win_row - st->h_pack));
    a.h_valid = (const int32_t *)(st->h_pack_dev + ((char *)st->h_win_valid - st->h_pack));
    a.h_aux = (const int32_t *)(st->h_pack_dev + ((char *)st->h_win_aux - st->h_pack));
    a.d_row = st->d_win_row; a.d_valid = st->d_win_valid; a.d_aux = st->d_win_aux; a.nw = nw; a.S = S;
    a.ring = st->ring;
    a.hist = st->hist; a.prev = st->prev;
    a.T = st->T; a.F = st->F; a.HR = st->HR;
    a.divisor = st->fp.pcm_divisor; a.clip = st->fp.clip; a.preemph = st->fp.pre_emphasis; a.hop = hop;
    a.start = f.start; a.wpad = f.wpad; a.bias = f.bias;
    a.n_mel = f.n_mel; a.floor_v = f.floor_v; a.log_off = f.log_off; a.scale = f.scale;
    a.hann = f.hann; a.tw256 = f.tw256; a.tw512 = f.tw512;
    {
      ww_launch_scope scope(ctx, "stream_frontend_kernel");
      if (st->fp.precise) {
        size_t sm = 2 * FFT_LD * 16 + 2 * 260 * 4 + ST_WL_BYTES + ST_RING * 4 + WW_CHUNK * 2;
        hipLaunchKernelGGL((stream_frontend_kernel<double>), dim3(S), dim3(128), sm, ctx->stream, a);
      } else {
        size_t sm = 2 * FFT_LD * 8 + 2 * 260 * 4 + ST_WL_BYTES + ST_RING * 4 + WW_CHUNK * 2;
        hipLaunchKernelGGL((stream_frontend_kernel<float>), dim3(S), dim3(128), sm, ctx->stream, a);
      }
    }
    WW_HIP(ctx, hipGetLastError());
    tl[3] = st_now_ns();
    if (nw && !st->incremental) {
      // the per-window kernels' scratch under the model's options of THIS tick (ww_model_set_option may have lowered the
      // front/tail threshold since the bank was created: the split form then wants nw x 19 x 192 floats)
      const size_t need = m->kind == WW_KIND_CRNN ? ww_crnn_workspace(m, nw, false) : ww_wave_workspace(m, nw);
      if (need > st->ws_bytes) {
        WW_HIP(ctx, hipStreamSynchronize(ctx->stream));
        WW_HIP(ctx, hipFree(st->ws));
        st->ws = nullptr;
        st->ws_bytes = 0;
        const size_t want = m->kind == WW_KIND_CRNN ? ww_crnn_workspace(m, 2 * S, false) : ww_wave_workspace(m, 2 * S);
        if (hipMalloc(&st->ws, want > need ? want : need) != hipSuccess) return ww_fail(ctx, WW_ENOMEM, "cannot grow the model scratch of %d streams", S);
        st->ws_bytes = want > need ? want : need;
      }
    }
    if (nw) {
      // the heads store a tick's few posteriors straight into pinned host memory - as {value, tick number} pairs where the
      // launch form writes them (every one-kernel form), else as rows of h_out: no device-to-host copy (a DMA operation of
      // its own) between the last kernel and the host's wake-up
      tagged = st->poll && (st->incremental || m->kind == WW_KIND_WAVENET || ww_crnn_forward_tags(m, nw));
      const ww_tick_tag *tg = tagged ? &tag : nullptr;
      const float *d_hist = st->hist;
      int rc = st->incremental
                   ? ww_k_crnn_stream_forward(ctx, m, d_hist, (int64_t)S * st->HR, st->d_win_row, st->d_win_valid, st->d_win_aux, st->gxc, nw, st->h_out_dev, tg)
               : m->kind == WW_KIND_CRNN
                   ? ww_k_crnn_forward(ctx, m, d_hist, (int64_t)S * st->HR, st->d_win_row, st->d_win_valid, 0, 0, 0, nw, st->ws, st->ws_bytes, st->h_out_dev, nullptr, tg)
                   : ww_k_wave_forward(ctx, m, d_hist, (int64_t)S * st->HR, st->d_win_row, st->d_win_valid, 0, 0, 0, nw, st->ws, st->ws_bytes, st->h_out_dev, nullptr, tg);
      if (rc) return rc;
    }
  }
  tl[4] = st_now_ns();
  if (tagged && st->poll && nw) {
    int rc = st_poll_tags(st);
    if (rc) return rc;
  } else {
    WW_HIP(ctx, hipStreamSynchronize(ctx->stream));
  }
  tl[5] = st_now_ns();
  int w = 0;
  for (int s = 0; s < S; ++s) {
    post[s * 2] = 0.f;
    post[s * 2 + 1] = 0.f;
    for (int k = 0; k < n_post[s]; ++k, ++w) {
      if (tagged) {
        const unsigned bits = (unsigned)st->h_tag[st->one_launch ? 2 * s + k : w];
        memcpy(&post[s * 2 + k], &bits, 4);
      } else {
        post[s * 2 + k] = st->h_out[(size_t)w * st->NO + pidx];
      }
    }
  }
  tl[6] = st_now_ns();
  for (int i = 0; i < WW_STREAM_TL_PHASES; ++i) st->tl_ns[i] += tl[i + 1] - tl[i];
  ++st->tl_ticks;
  return WW_OK;
}

int ww_stream_step(ww_streams *st, const int16_t *frames, const uint8_t *is_speech, float *post, int32_t *n_post) {
  WW_GUARD_BEGIN
  if (!st) return WW_EINVAL;
  // a tick that failed after the host's mirrors (fill, ring positions, state parity) had advanced leaves them out of step with
  // the device: the bank refuses further ticks instead of producing posteriors of a state that never existed
  if (st->broken) return ww_fail(st->ctx, WW_ESTATE, "this stream bank failed in an earlier tick: destroy it and create a new one");
  bool mutated = false;
  const int rc = stream_step_impl(st, frames, is_speech, post, n_post, &mutated);
  if (rc != WW_OK && mutated) st->broken = true;
  return rc;
  WW_GUARD_END(st ? st->ctx : nullptr)
}

// ---- host stages of the streaming pipeline for S streams in lock step -------------------------------------------------------
// The reference runs three stage objects per stream and frame (spokestack/pipeline.py:25-28 with demo.py:29-36's stage list):
// VoiceActivityDetector -> WakewordTrigger -> ActivationTimeout, each a few comparisons on the shared SpeechContext.  For S
// streams that is 3 S Python calls per 20 ms tick (round 5: ~220 us of interpreter at 128 streams around a 35 us device tick).
// Here each stage is ONE pass over S streams on plain arrays the host language owns (wwhip/context.py: ContextBank); a stage
// reports the ids whose state changed, so the caller raises activate / deactivate events for those streams only.

// spokestack/vad/webrtc.py:59-77 - run-length hysteresis on the classifier's raw decision; context.is_speech is the state.
int ww_vad_bank_step(int32_t S, const uint8_t *raw, int32_t rise_frames, int32_t fall_frames, uint8_t *run_value,
                     int64_t *run_length, uint8_t *is_speech, int32_t *n_changed) {
  WW_GUARD_BEGIN
  if (S < 0 || (S > 0 && (!raw || !run_value || !run_length || !is_speech))) return WW_EINVAL;
  int changed = 0;
  for (int s = 0; s < S; ++s) {
    const uint8_t r = raw[s] != 0;
    if (r == run_value[s]) {
      ++run_length[s];
    } else {
      run_value[s] = r;
      run_length[s] = 1;
    }
    if (r != (is_speech[s] != 0)) {
      if (r && run_length[s] >= rise_frames) { is_speech[s] = 1; ++changed; }
      if (!r && run_length[s] >= fall_frames) { is_speech[s] = 0; ++changed; }
    }
  }
  if (n_changed) *n_changed = changed;
  return WW_OK;
  WW_GUARD_END(nullptr)
}

// spokestack/wakeword/tflite.py:134-146 (VAD edge, reset on the fall) and :232-239 (running maximum, threshold, activation)
// over the posteriors a tick delivered.  The comparison with the threshold is the reference's: a float32 posterior against a
// Python float, i.e. in double.
static void trigger_update(int S, const uint8_t *is_speech, uint8_t *is_active, const float *post, const int32_t *n_post,
                           double threshold, uint8_t *was_speech, float *posterior_max, int32_t *fired_ids, int32_t *n_fired,
                           int32_t *fall_ids, int32_t *n_fall) {
  int nf = 0, nl = 0;
  for (int s = 0; s < S; ++s) {
    const uint8_t sp = is_speech[s] != 0;
    const bool fall = was_speech[s] && !sp;
    was_speech[s] = sp;
    bool fire = false;
    for (int k = 0; k < n_post[s]; ++k) {
      const float p = post[s * 2 + k];
      if (p > posterior_max[s]) posterior_max[s] = p;
      if ((double)p > threshold) fire = true;
    }
    if (fire && !is_active[s]) {
      is_active[s] = 1;
      fired_ids[nf++] = s;
    }
    if (fall) {
      posterior_max[s] = 0.f;
      fall_ids[nl++] = s;
    }
  }
  *n_fired = nf;
  *n_fall = nl;
}

int ww_trigger_bank_step(int32_t S, const uint8_t *is_speech, uint8_t *is_active, const float *post, const int32_t *n_post,
                         double threshold, uint8_t *was_speech, float *posterior_max, int32_t *fired_ids, int32_t *n_fired,
                         int32_t *fall_ids, int32_t *n_fall) {
  WW_GUARD_BEGIN
  if (S < 0 || !n_fired || !n_fall || (S > 0 && (!is_speech || !is_active || !post || !n_post || !was_speech || !posterior_max ||
                                                   !fired_ids || !fall_ids)))
    return WW_EINVAL;
  for (int s = 0; s < S; ++s)
    if (n_post[s] < 0 || n_post[s] > 2) return WW_EINVAL;
  trigger_update(S, is_speech, is_active, post, n_post, threshold, was_speech, posterior_max, fired_ids, n_fired, fall_ids, n_fall);
  return WW_OK;
  WW_GUARD_END(nullptr)
}

// spokestack/activation_timeout.py:25-38.  min_frames / max_frames are the reference's quotients (min_active / frame_width), kept
// as doubles: the comparisons are int > float there.
int ww_timeout_bank_step(int32_t S, const uint8_t *is_speech, uint8_t *is_active, uint8_t *was_speech, int32_t *active_frames,
                         double min_frames, double max_frames, int32_t *deact_ids, int32_t *n_deact) {
  WW_GUARD_BEGIN
  if (S < 0 || !n_deact || (S > 0 && (!is_speech || !is_active || !was_speech || !active_frames || !deact_ids))) return WW_EINVAL;
  int nd = 0;
  for (int s = 0; s < S; ++s) {
    const uint8_t sp = is_speech[s] != 0;
    const bool fell = was_speech[s] && !sp;
    was_speech[s] = sp;
    if (!is_active[s]) continue;
    const int len = ++active_frames[s];
    if ((double)len > min_frames && (fell || (double)len > max_frames)) {
      active_frames[s] = 0;
      is_active[s] = 0;
      deact_ids[nd++] = s;
    }
  }
  *n_deact = nd;
  return WW_OK;
  WW_GUARD_END(nullptr)
}

// WakewordTrigger.__call__ for S streams as ONE call: the tick (ww_stream_step with bit 0 = is_speech, bit 1 = is_active as they
// stand BEFORE the tick), then ww_trigger_bank_step over its posteriors, then WakewordTrigger.reset for the streams whose VAD bit
// fell (tflite.py:143-146).  is_active is updated in place.
int ww_stream_step_trigger(ww_streams *st, const int16_t *frames, const uint8_t *is_speech, uint8_t *is_active, double threshold,
                           uint8_t *was_speech, float *posterior_max, float *post, int32_t *n_post, int32_t *fired_ids,
                           int32_t *n_fired, int32_t *fall_ids, int32_t *n_fall) {
  WW_GUARD_BEGIN
  if (!st) return WW_EINVAL;
  if (!is_speech || !is_active || !was_speech || !posterior_max || !fired_ids || !n_fired || !fall_ids || !n_fall)
    return ww_fail(st->ctx, WW_EINVAL, "NULL argument");
  *n_fired = *n_fall = 0;
  st->stage_flags.resize((size_t)st->S);
  for (int s = 0; s < st->S; ++s) st->stage_flags[s] = (uint8_t)((is_speech[s] != 0) | ((is_active[s] != 0) << 1));
  int rc = ww_stream_step(st, frames, st->stage_flags.data(), post, n_post);
  if (rc) return rc;
  trigger_update(st->S, is_speech, is_active, post, n_post, threshold, was_speech, posterior_max, fired_ids, n_fired, fall_ids, n_fall);
  if (*n_fall) rc = ww_stream_reset(st, fall_ids, *n_fall);
  return rc;
  WW_GUARD_END(st ? st->ctx : nullptr)
}

// The three stages of a tick in one call (include/wwhip.h): exactly the three entry points above, in stage order.
int ww_pipeline_bank_step(ww_streams *st, const int16_t *frames, ww_pipeline_state *ps) {
  WW_GUARD_BEGIN
  if (!st) return WW_EINVAL;
  if (!ps) return ww_fail(st->ctx, WW_EINVAL, "NULL state block");
  ps->n_vad_changed = ps->n_fired = ps->n_fall = ps->n_deact = 0;
  int rc = ww_vad_bank_step(st->S, ps->raw, ps->rise_frames, ps->fall_frames, ps->run_value, ps->run_length, ps->is_speech, &ps->n_vad_changed);
  if (rc) return ww_fail(st->ctx, rc, "ww_pipeline_bank_step: the VAD stage's arrays");
  rc = ww_stream_step_trigger(st, frames, ps->is_speech, ps->is_active, ps->threshold, ps->wake_was_speech, ps->posterior_max, ps->post,
                              ps->n_post, ps->fired_ids, &ps->n_fired, ps->fall_ids, &ps->n_fall);
  if (rc) return rc;
  rc = ww_timeout_bank_step(st->S, ps->is_speech, ps->is_active, ps->timeout_was_speech, ps->active_frames, ps->min_frames, ps->max_frames,
                            ps->deact_ids, &ps->n_deact);
  if (rc) return ww_fail(st->ctx, rc, "ww_pipeline_bank_step: the timeout stage's arrays");
  return WW_OK;
  WW_GUARD_END(st ? st->ctx : nullptr)
}

int ww_stream_timeline(ww_streams *st, double *mean_ns, int64_t *ticks, int32_t reset) {
  WW_GUARD_BEGIN
  if (!st) return WW_EINVAL;
  if (ticks) *ticks = st->tl_ticks;
  if (mean_ns)
    for (int i = 0; i < WW_STREAM_TL_PHASES; ++i) mean_ns[i] = st->tl_ticks ? (double)st->tl_ns[i] / (double)st->tl_ticks : 0.0;
  if (reset) {
    memset(st->tl_ns, 0, sizeof(st->tl_ns));
    st->tl_ticks = 0;
  }
  return WW_OK;
  WW_GUARD_END(st ? st->ctx : nullptr)
}

}  // extern "C"
