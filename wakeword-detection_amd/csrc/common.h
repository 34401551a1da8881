// Internal declarations shared by the libwwhip.so translation units (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <exception>
#include <map>
#include <new>
#include <string>
#include <vector>

#include "../../include/wwhip.h"

#define WW_WAVE 64

struct ww_prof_entry {
  int calls = 0;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
  double total_ms = 0.0;
};

// Growable device / pinned-host arenas: ensure() is called by the entry points before they start enqueueing.
struct ww_arena {
  void *ptr = nullptr;
  size_t cap = 0;
};

struct ww_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  char err[512] = {0};
  ww_arena dev;     // device workspace
  ww_arena pinned;  // pinned host staging
  bool profiling = false;
  std::map<std::string, ww_prof_entry> prof;
  hipEvent_t t0 = nullptr, t1 = nullptr;
  // sample/frame offset tables of the equal-length clip batches seen so far (tiny, built once)
  struct clip_offs_t {
    int n_clips = 0, samples = 0, hop = 0;
    int64_t *d_so = nullptr, *d_fo = nullptr;
  };
  std::vector<clip_offs_t> clip_offs;
  // host-built launch descriptors on their way to the device (ww_k_crnn_segments_forward): two page-locked buffers used
  // alternately, each free again once the event behind its copies has passed - the call never waits for its kernels
  ww_arena desc_pin[2];
  hipEvent_t desc_ev[2] = {nullptr, nullptr};
  bool desc_busy[2] = {false, false};
  unsigned desc_k = 0;
};

// Device-resident mel filterbank in banded form: band m covers bins [start[m], start[m]+len[m])
// with weights w[woff[m] .. woff[m]+len[m]).
#define WW_MELV_CHUNKS 16  // 9 + 4 + 3 float4 chunks: groups of 36, 16 and 12 taps
struct ww_filter_dev {
  int n_mel = 0, n_bins = 0;
  float floor_v = 0, log_off = 0, scale = 0;
  int *start = nullptr, *len = nullptr, *woff = nullptr;
  float *w = nullptr, *bias = nullptr;
  float *wdense = nullptr;   // [n_mel][n_bins] dense weights (filter.tflite layout)
  float *wpad = nullptr;     // [WW_MEL_TAPS][64] tap-major zero-padded weights (kernel form)
  int total_taps = 0, max_len = 0;
  double *hann = nullptr;    // [512] np.hanning(512) in fp64
  double *tw256 = nullptr;   // [256][2] e^{-2 pi i k / 256}
  double *tw512 = nullptr;   // [256][2] e^{-2 pi i k / 512}
  double *tw16 = nullptr;    // [16 k1][16 j][2] e^{-2 pi i j k1 / 256}
  // Mel filter in lane form for the batched front end (frontend.hip): the bands, sorted by width, are dealt
  // to three groups of 16 "slots"; slot s of group g accumulates one band over WW_MELV_CAP[g] padded taps.
  float *melV = nullptr;     // [WW_MELV_CHUNKS][16 slots] float4: 0.5 * weight of taps 4c..4c+3 (chunks of group 0, 1, 2)
  int *melVmeta = nullptr;   // [3][16]: first bin | band << 16 (band 0xffff: empty slot)
  int melv_aligned = 0;      // every first bin is a multiple of 4
};

struct ww_crnn_dev {
  int n_mel, T, C, KF, KT, SF, ST, PF, PT, OF, OT, H, NOUT, HEAD;
  float *conv_w = nullptr;   // [KF*KT][C]  (k-major for the implicit GEMM)
  float *conv_b = nullptr;   // [C]
  float *wx1 = nullptr;      // [2*3H][OF*C]  rows: fwd z,r,h then bwd z,r,h
  float *wx1s = nullptr;     // the same matrix in MFMA B-operand order [OF*C/4][2*3H][4] (crnn_fused_kernel)
  float *bx1 = nullptr;      // [2*3H]
  float *wh1 = nullptr;      // [2][3H][H]
  float *bh1 = nullptr;      // [2][3H]
  float *wx2 = nullptr;      // [2*3H][2H]
  float *wx2s = nullptr;     // the same in MFMA B-operand order [2H/4][2*3H][4]
  unsigned short *cwb = nullptr;   // split-bf16 mode: conv weights hi/lo planes in A-operand order (crnn_fused_bf16_kernel)
  unsigned short *wx1b = nullptr;  // split-bf16 mode: W_x1 hi/lo planes in B-operand order
  float *bx2 = nullptr;
  float *wh2 = nullptr;
  float *bh2 = nullptr;
  float *w1 = nullptr, *b1 = nullptr, *w2 = nullptr, *b2 = nullptr;
  // Any other conv geometry (utils/CRNN_files/*_old.tflite: 20x5 kernel, stride 8x2, VALID, 74 steps of 96 features)
  // takes the generic kernels of crnn.hip: direct conv, the same MFMA GEMM on K padded to FEATP, step-wise GRUs.
  bool generic = false;
  int FEATP = 0;               // OF*C rounded up to the GEMM's K tile (64)
  float *conv_wt = nullptr;    // [KF*KT][C]
  float *conv_wL = nullptr, *conv_wR = nullptr;  // conv_w with the taps that meet a window's zero padding cleared (first 6 / last 7 frames): crnn_rows_kernel
  float *wx1p = nullptr;       // [2*3H][FEATP], zero padded
};

struct ww_wave_dev {
  int T, n_mel, C, S, NB, NOUT;
  std::vector<int> dil, order, has_res;
  float *w_in = nullptr, *b_in = nullptr;          // [n_mel][C], [C]
  float *bn_s = nullptr, *bn_t = nullptr;          // [NB][C]
  float *w_gate = nullptr, *b_gate = nullptr;      // [NB][3*C][2C] (cols: sig 0..C-1, tanh C..2C-1), [NB][2C]
  float *w_rs = nullptr, *b_rs = nullptr;          // [NB][C][C+S] (cols: res 0..C-1, skip C..), [NB][C+S]
  float *d_w1 = nullptr, *d_b1 = nullptr, *d_w2 = nullptr, *d_b2 = nullptr;
  uint16_t *wpk = nullptr;                         // split-bf16 parameter pages [NB]{[14][64][8] bf16 A operands, [7][16] f32 vectors}
  bool order_is_natural = true;
};

struct ww_model {
  ww_ctx *ctx = nullptr;
  int kind = 0;
  ww_model_info info = {};
  ww_filter_dev filt;
  ww_crnn_dev crnn;
  ww_wave_dev wave;
  int precision = 0;  // WW_PRECISION_*
  int opt_split_at = 1024;  // WW_OPT_CRNN_SPLIT_AT
  int opt_slide_min = 64;   // WW_OPT_CRNN_SLIDE_MIN
  int opt_tail_mfma = 1;    // WW_OPT_CRNN_TAIL_MFMA
  int opt_wave_rowmajor = 0;  // WW_OPT_WAVENET_ROWMAJOR
  std::vector<void *> allocs;
};

int ww_fail(ww_ctx *ctx, int code, const char *fmt, ...);
int ww_ensure(ww_ctx *ctx, ww_arena &a, size_t bytes, bool pinned);

#define WW_NUM_CUS 256  // MI355X (gfx950): 8 XCDs x 32 CUs

// No exception crosses the C ABI (every entry point returns a status, never throws): the body of every exported
// function sits between WW_GUARD_BEGIN and WW_GUARD_END(ctx) - std::bad_alloc from a container or `new` becomes WW_ENOMEM,
// anything else WW_EINTERNAL, with the text where ww_last_error finds it (ctx may be nullptr: the thread's text).  The few
// exported functions that only read a field or return a constant carry WW_NOTHROW instead (tests/test_cabi_and_dist.py
// checks that every definition has one or the other).
#define WW_GUARD_BEGIN try {
#define WW_GUARD_END(ctx_)                                                                                  \
  }                                                                                                         \
  catch (const std::bad_alloc &) { return ww_fail((ctx_), WW_ENOMEM, "%s: out of host memory", __func__); } \
  catch (const std::exception &e_) { return ww_fail((ctx_), WW_EINTERNAL, "%s: %s", __func__, e_.what()); } \
  catch (...) { return ww_fail((ctx_), WW_EINTERNAL, "%s: unknown exception", __func__); }
// An object a create function has allocated and not yet handed out: destroyed if the function leaves early (an error
// return, or an exception on its way to WW_GUARD_END).
template <typename T, int (*Destroy)(T *)>
struct ww_scoped {
  T *p;
  explicit ww_scoped(T *q) : p(q) {}
  ww_scoped(const ww_scoped &) = delete;
  ~ww_scoped() {
    if (p) Destroy(p);
  }
  T *release() {
    T *r = p;
    p = nullptr;
    return r;
  }
};
#define WW_NOTHROW /* marker: the body allocates nothing and calls nothing that can throw */

#define WW_HIP(ctx, expr)                                                                       \
  do {                                                                                          \
    hipError_t e_ = (expr);                                                                     \
    if (e_ != hipSuccess)                                                                       \
      return ww_fail((ctx), WW_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)

// The device-pointer entry points enqueue on the context's stream without touching the caller's
// current device (the caller may be a framework that tracks it): switch only if needed, and restore.
struct ww_device_scope {
  int prev = -1;
  bool changed = false;
  hipError_t err = hipSuccess;  // why the switch to `dev` failed; entry points check ok() before they allocate or launch
  explicit ww_device_scope(int dev) {
    err = hipGetDevice(&prev);
    if (err == hipSuccess && prev != dev) {
      err = hipSetDevice(dev);
      changed = err == hipSuccess;
    }
  }
  bool ok() const { return err == hipSuccess; }
  ~ww_device_scope() {
    if (changed) (void)hipSetDevice(prev);
  }
};
// Declares the scope and leaves the entry point with WW_EHIP if the context's device could not be made current.
#define WW_ON_DEVICE(ctx_, name_)                                                                                   \
  ww_device_scope name_((ctx_)->device);                                                                            \
  if (!name_.ok())                                                                                                  \
    return ww_fail((ctx_), WW_EHIP, "cannot switch to device %d: %s", (ctx_)->device, hipGetErrorString(name_.err))

// Bracket a kernel launch with profiling events when enabled.
struct ww_launch_scope {
  ww_ctx *ctx;
  const char *name;
  hipEvent_t a = nullptr, b = nullptr;
  ww_launch_scope(ww_ctx *c, const char *n) : ctx(c), name(n) {
    if (ctx->profiling) {
      hipEventCreate(&a);
      hipEventCreate(&b);
      hipEventRecord(a, ctx->stream);
    }
  }
  ~ww_launch_scope() {
    if (ctx->profiling) {
      hipEventRecord(b, ctx->stream);
      auto &e = ctx->prof[name];
      e.calls++;
      e.pending.emplace_back(a, b);
    }
  }
};

// bump allocator over the ctx workspace
struct ww_bump {
  char *base;
  size_t off = 0, cap;
  ww_bump(void *p, size_t c) : base((char *)p), cap(c) {}
  template <typename T>
  T *take(size_t n) {
    size_t bytes = (n * sizeof(T) + 255) & ~size_t(255);
    T *r = (T *)(base + off);
    off += bytes;
    return r;
  }
  static size_t need(size_t n, size_t elem) { return (n * elem + 255) & ~size_t(255); }
};

// A tick's posteriors as {value, tick number} pairs in page-locked host memory, each written with ONE 8-byte store by the head
// of the model kernels (crnn.hip: cf_phases_d_to_g; wavenet.hip): the host polls them instead of waiting for the runtime's
// completion signal (streams.hip: ww_stream_step; 5.6 us of wake-up on this box, tools/launch_probe.hip).  slots == nullptr: off.
struct ww_tick_tag {
  unsigned long long *slots;  // [windows of the launch]: value bits | (uint64) seq << 32
  unsigned seq;
  int pidx;                   // the head's output element that is the posterior
};

// The streaming front end's side of crnn_stream_kernel<FE != 0> - ONE launch per tick (crnn.hip).  Workgroup 2 s + k is window k of
// stream s's tick; it reads the stream's control words and samples over the bus itself.
#define WW_ST_RING 832  // 511 + 320 rounded up
struct ww_tick_fe {
  const int16_t *frames;  // page-locked host memory [S][WW_CHUNK]
  const int32_t *ctl;     // page-locked host memory [S][4]: fill, n_frames, flags (1 speech, 2 active, 4 state parity), pos | rowq << 16
  float *ring;            // [2][S][WW_ST_RING]: a stream's sample ring, ping-pong by its state parity (read [par], written [par ^ 1])
  float *prev;            // [2][S] pre-emphasis carry, likewise
  float *hist;            // [S][HR][F] mirrored mel rings (the model's `mel`)
  int S, HR;
  float divisor;
  int clip;
  float preemph;
  int hop;
};

// ---- kernel-side entry points implemented in the .hip files ------------------------------
int ww_k_logmel(ww_ctx *ctx, const ww_model *m, const int16_t *d_pcm, const float *d_f32, const int64_t *d_sample_offs,
                const int64_t *d_frame_offs, int n_utt, int64_t total_frames, int64_t max_frames_per_utt,
                const ww_frontend_params *fp, float *d_mel, int64_t uniform_samples = 0,  // > 0: equal clips back to back from sample 0
                int64_t total_samples_hint = 0);  // > 0: sample_offs[n_utt], where the host knows it (device tables otherwise)
int ww_k_stft_mag(ww_ctx *ctx, const ww_model *m, const float *d_frames, int64_t n, int precise, float *d_mag);
int ww_k_mel_only(ww_ctx *ctx, const ww_model *m, const float *d_mag, int64_t n, float *d_mel);
int ww_k_crnn_detect(ww_ctx *ctx, const ww_model *m, const float *d_enc, int nw, float *d_out);
int ww_k_wave_detect(ww_ctx *ctx, const ww_model *m, const float *d_enc, int nw, float *d_out);
int ww_k_viterbi2(ww_ctx *ctx, const float *d_in, int64_t n, int T, float stay_bonus, int in_is_cost, unsigned char *d_path,
                  unsigned char *d_wake);

// `sliding`: the caller may pass regular sliding windows (row0 + w*hop), for which the scratch of crnn_rows_kernel is reserved too
size_t ww_crnn_workspace(const ww_model *m, int n_windows, bool sliding = true);
int ww_k_crnn_init_device(ww_ctx *ctx);  // per-device kernel attributes (dynamic LDS above 64 KB)
// ws_bytes: capacity of ws; a launch form that needs more (the model's options may have changed since ws was sized) fails with
// WW_EINVAL instead of writing past it
int ww_k_crnn_forward(ww_ctx *ctx, const ww_model *m, const float *d_mel, int64_t mel_rows, const int64_t *d_win_row,
                      const int32_t *d_win_valid, int64_t row0, int hop, int valid_const, int n_windows, void *ws, size_t ws_bytes,
                      float *d_out, float *d_enc, const ww_tick_tag *tag = nullptr);
bool ww_crnn_segments_capable(const ww_model *m, int hop);
int ww_k_crnn_segments_forward(ww_ctx *ctx, const ww_model *m, const float *d_mel, int64_t mel_rows, const int64_t *seg_row0,
                               const int32_t *seg_nw, int n_seg, int hop, float *d_out);
bool ww_crnn_stream_capable(const ww_model *m);
int ww_k_crnn_stream_forward(ww_ctx *ctx, const ww_model *m, const float *d_hist, int64_t hist_rows, const int64_t *d_win_row,
                             const int32_t *d_win_valid, const int32_t *d_win_aux, float *d_gxc, int n_windows, float *d_out,
                             const ww_tick_tag *tag = nullptr);
// one launch per tick: front end + incremental CRNN of all S streams (2 S workgroups); posteriors as tags only
int ww_k_crnn_tick(ww_ctx *ctx, const ww_model *m, const ww_tick_fe &fe, int precise, float *d_gxc, const ww_tick_tag &tag);
// does a ww_k_crnn_forward launch of n explicit windows write the tags (the one-kernel forms do)?
bool ww_crnn_forward_tags(const ww_model *m, int n_windows);
size_t ww_wave_workspace(const ww_model *m, int n_windows);
int ww_k_wave_forward(ww_ctx *ctx, const ww_model *m, const float *d_mel, int64_t mel_rows, const int64_t *d_win_row,
                      const int32_t *d_win_valid, int64_t row0, int hop, int valid_const, int n_windows, void *ws, size_t ws_bytes,
                      float *d_out, float *d_enc, const ww_tick_tag *tag = nullptr);
int ww_k_posterior_pick(ww_ctx *ctx, const float *d_rows, int64_t n, int n_out, int pidx, const int64_t *d_seg_offs, int64_t n_seg,
                        float *d_out);
bool ww_wave_tick_capable(const ww_model *m, int n_streams);
int ww_k_wave_tick(ww_ctx *ctx, const ww_model *m, const ww_tick_fe &fe, int precise, const ww_tick_tag &tag);
int ww_k_far_frr(ww_ctx *ctx, const float *d_pos, int64_t n_pos, const float *d_neg, int64_t n_neg, int win,
                 const double *d_thr, int n_thr, double *d_smoothed, unsigned long long *d_pos_cnt,
                 unsigned long long *d_fa_cnt);

// streaming CRNN: slots of the per-stream ring of projected rows (crnn.hip, crnn_stream_kernel): a row written for the window
// that ends at stream row r is last read by the window that ends at r + 128, and overwritten at r + WW_STREAM_GXC
#define WW_STREAM_GXC 144

#ifdef __HIPCC__
// max(x, 0) as ONE instruction: fmaxf compiles to a canonicalising v_max (x, x) in front of the v_max (0, x) when its
// argument comes out of an MFMA.  On the bit pattern a signed-integer max does the same job (negative floats, -0
// included, are negative integers).  Not inline asm: the compiler does not see an MFMA -> VALU read hazard through it
// and omits the wait states.
__device__ __forceinline__ float relu1(float x) {
  const int b = __float_as_int(x);
  return __int_as_float(b > 0 ? b : 0);
}
#endif
