// C ABI of libwwhip.so: context, model upload, host/device entry points (see include/wwhip.h).
#include "common.h"


#include <algorithm>
#include <cmath>
#include <mutex>

// message of a failure that has no context to carry it (ww_ctx_create, NULL handles): one buffer per host thread, so that
// two threads creating contexts concurrently each read their own text through ww_last_error(NULL)
static thread_local char g_err[512] = {0};

int ww_fail(ww_ctx *ctx, int code, const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  char *dst = ctx ? ctx->err : g_err;
  vsnprintf(dst, 512, fmt, ap);
  va_end(ap);
  return code;
}

int ww_ensure(ww_ctx *ctx, ww_arena &a, size_t bytes, bool pinned) {
  if (bytes <= a.cap) return WW_OK;
  if (a.ptr) {
    hipStreamSynchronize(ctx->stream);
    if (pinned) hipHostFree(a.ptr); else hipFree(a.ptr);
    a.ptr = nullptr;
    a.cap = 0;
  }
  size_t want = bytes + bytes / 4 + (1 << 20);
  hipError_t e = pinned ? hipHostMalloc(&a.ptr, want, hipHostMallocDefault) : hipMalloc(&a.ptr, want);
  if (e != hipSuccess) {
    a.ptr = nullptr;
    return ww_fail(ctx, WW_ENOMEM, "cannot allocate %zu bytes of %s memory: %s", want, pinned ? "pinned host" : "device",
                   hipGetErrorString(e));
  }
  a.cap = want;
  return WW_OK;
}


// Small host-pointer calls are latency-bound, and a copy-engine operation (hipMemcpyAsync) costs more than a few KB
// are worth: their inputs are staged in the context's pinned arena, which the kernels read over the bus themselves,
// and their outputs are stored into it by the kernels.  ww_small_io carves that arena and translates addresses.
#define WW_SMALL_IO_BYTES (256u << 10)
struct ww_small_io {
  ww_ctx *ctx;
  char *host = nullptr, *dev = nullptr;
  size_t off = 0;
  explicit ww_small_io(ww_ctx *c) : ctx(c) {}
  int init(size_t bytes) {
    int rc = ww_ensure(ctx, ctx->pinned, bytes + 1024, true);
    if (rc) return rc;
    host = (char *)ctx->pinned.ptr;
    if (hipHostGetDevicePointer((void **)&dev, host, 0) != hipSuccess)
      return ww_fail(ctx, WW_EHIP, "pinned arena is not visible to the device");
    return WW_OK;
  }
  template <typename T>
  T *take(size_t n) {
    T *r = (T *)(host + off);
    off += ww_bump::need(n, sizeof(T));
    return r;
  }
  template <typename T>
  T *dv(T *h) const { return (T *)(dev + ((char *)h - host)); }
};

extern "C" {

static_assert(WW_ABI == 4, "ww_version's text carries the ABI number");
const char *ww_version(void) WW_NOTHROW { return "wwhip 0.5 (gfx950; ABI 4: ww_stream_create takes flags, ww_host_stage_i16, ww_uploader_*, ww_stream_timeline)"; }

int ww_runtime_info(int32_t *built_hip_version, int32_t *runtime_version, int32_t *driver_version) {
  WW_GUARD_BEGIN
  if (built_hip_version) *built_hip_version = HIP_VERSION;  // headers the library was compiled against
  int rt = 0, drv = 0;
  const hipError_t e1 = hipRuntimeGetVersion(&rt);
  const hipError_t e2 = hipDriverGetVersion(&drv);  // may fail where no GPU is visible: reported as 0
  if (runtime_version) *runtime_version = e1 == hipSuccess ? rt : 0;
  if (driver_version) *driver_version = e2 == hipSuccess ? drv : 0;
  return e1 == hipSuccess ? WW_OK : WW_EHIP;
  WW_GUARD_END(nullptr)
}

const char *ww_last_error(const ww_ctx *ctx) WW_NOTHROW { return ctx ? ctx->err : g_err; }

int ww_ctx_create(int device, void *external_stream, ww_ctx **out) {
  WW_GUARD_BEGIN
  if (!out) return ww_fail(nullptr, WW_EINVAL, "out is NULL");
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return ww_fail(nullptr, WW_ENODEVICE, "no HIP device visible");
  if (device < 0 || device >= n) return ww_fail(nullptr, WW_EINVAL, "device %d out of range (0..%d)", device, n - 1);
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess) return ww_fail(nullptr, WW_EHIP, "hipGetDeviceProperties failed");
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return ww_fail(nullptr, WW_ENODEVICE, "device %d is %s; libwwhip.so carries gfx950 code only", device, prop.gcnArchName);
  ww_device_scope dev_scope(device);  // the stream and events below are created on `device`; the caller's current device is restored
  if (!dev_scope.ok()) return ww_fail(nullptr, WW_EHIP, "cannot switch to device %d: %s", device, hipGetErrorString(dev_scope.err));
  ww_ctx *c = new ww_ctx();
  c->device = device;
  if (external_stream) {
    c->stream = (hipStream_t)external_stream;
  } else {
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
      delete c;
      return ww_fail(nullptr, WW_EHIP, "hipStreamCreate failed");
    }
    c->own_stream = true;
  }
  ww_scoped<ww_ctx, ww_ctx_destroy> own(c);
  hipEventCreate(&c->t0);
  hipEventCreate(&c->t1);
  // kernel attributes are per device: set them for this context's device (a second GPU of the process gets its own)
  if (int rc = ww_k_crnn_init_device(c)) {
    snprintf(g_err, sizeof g_err, "%s", c->err);
    return rc;
  }
  *out = own.release();
  return WW_OK;
  WW_GUARD_END(nullptr)
}

int ww_ctx_destroy(ww_ctx *ctx) {
  WW_GUARD_BEGIN
  if (!ctx) return WW_OK;
  ww_device_scope dev_scope(ctx->device);
  hipStreamSynchronize(ctx->stream);
  for (auto &co : ctx->clip_offs) {
    hipFree(co.d_so);
    hipFree(co.d_fo);
  }
  for (auto &kv : ctx->prof)
    for (auto &p : kv.second.pending) {
      hipEventDestroy(p.first);
      hipEventDestroy(p.second);
    }
  if (ctx->dev.ptr) hipFree(ctx->dev.ptr);
  if (ctx->pinned.ptr) hipHostFree(ctx->pinned.ptr);
  for (int k = 0; k < 2; ++k) {
    if (ctx->desc_pin[k].ptr) hipHostFree(ctx->desc_pin[k].ptr);
    if (ctx->desc_ev[k]) hipEventDestroy(ctx->desc_ev[k]);
  }
  hipEventDestroy(ctx->t0);
  hipEventDestroy(ctx->t1);
  if (ctx->own_stream) hipStreamDestroy(ctx->stream);
  delete ctx;
  return WW_OK;
  WW_GUARD_END(nullptr)
}

int ww_ctx_synchronize(ww_ctx *ctx) {
  WW_GUARD_BEGIN
  if (!ctx) return WW_EINVAL;
  WW_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return WW_OK;
  WW_GUARD_END(ctx)
}

void *ww_ctx_stream(ww_ctx *ctx) WW_NOTHROW { return ctx ? (void *)ctx->stream : nullptr; }

int ww_profile_enable(ww_ctx *ctx, int on) {
  WW_GUARD_BEGIN
  if (!ctx) return WW_EINVAL;
  ctx->profiling = on != 0;
  return WW_OK;
  WW_GUARD_END(ctx)
}

int ww_profile_read(ww_ctx *ctx, char *json, size_t cap) {
  WW_GUARD_BEGIN
  if (!ctx || !json || cap < 8) return WW_EINVAL;
  WW_HIP(ctx, hipStreamSynchronize(ctx->stream));
  std::string s = "{";
  bool first = true;
  for (auto &kv : ctx->prof) {
    for (auto &p : kv.second.pending) {
      float ms = 0.f;
      hipEventElapsedTime(&ms, p.first, p.second);
      kv.second.total_ms += ms;
      hipEventDestroy(p.first);
      hipEventDestroy(p.second);
    }
    kv.second.pending.clear();
    char buf[256];
    snprintf(buf, sizeof buf, "%s\"%s\": {\"calls\": %d, \"total_ms\": %.6f}", first ? "" : ", ", kv.first.c_str(),
             kv.second.calls, kv.second.total_ms);
    s += buf;
    first = false;
  }
  s += "}";
  ctx->prof.clear();
  if (s.size() + 1 > cap) return ww_fail(ctx, WW_EINVAL, "profile buffer too small (%zu needed)", s.size() + 1);
  memcpy(json, s.c_str(), s.size() + 1);
  return WW_OK;
  WW_GUARD_END(ctx)
}

int ww_timer_start(ww_ctx *ctx) {
  WW_GUARD_BEGIN
  if (!ctx) return WW_EINVAL;
  WW_HIP(ctx, hipEventRecord(ctx->t0, ctx->stream));
  return WW_OK;
  WW_GUARD_END(ctx)
}

int ww_timer_stop(ww_ctx *ctx, float *ms) {
  WW_GUARD_BEGIN
  if (!ctx || !ms) return WW_EINVAL;
  WW_HIP(ctx, hipEventRecord(ctx->t1, ctx->stream));
  WW_HIP(ctx, hipEventSynchronize(ctx->t1));
  WW_HIP(ctx, hipEventElapsedTime(ms, ctx->t0, ctx->t1));
  return WW_OK;
  WW_GUARD_END(ctx)
}

}  // extern "C"

// ------------------------------------------------------------------------------------------
// blob parsing + upload
// ------------------------------------------------------------------------------------------
struct blob_view {
  const uint8_t *base;
  size_t len;
  uint32_t n, kind;
  const void *find(const char *name, uint32_t *count) const {
    for (uint32_t i = 0; i < n; ++i) {
      const uint8_t *e = base + 16 + 32 * (size_t)i;
      if (strncmp((const char *)e, name, 24) == 0) {
        uint32_t off, cnt;
        memcpy(&off, e + 24, 4);
        memcpy(&cnt, e + 28, 4);
        if ((size_t)off + (size_t)cnt * 4 > len) return nullptr;
        if (count) *count = cnt;
        return base + off;
      }
    }
    return nullptr;
  }
};

template <typename T>
static T *upload(ww_model *m, const std::vector<T> &v) {
  void *d = nullptr;
  size_t bytes = (v.size() ? v.size() : 1) * sizeof(T);
  if (hipMalloc(&d, bytes) != hipSuccess) return nullptr;
  m->allocs.push_back(d);
  if (v.size() && hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) return nullptr;
  return (T *)d;
}

#define NEED_F(var, name, cnt_expect)                                                                       \
  uint32_t var##_n = 0;                                                                                     \
  const float *var = (const float *)bv.find(name, &var##_n);                                                \
  if (!var || (size_t)var##_n != (size_t)(cnt_expect))                                                      \
    return ww_fail(ctx, WW_EBLOB, "blob section %s missing or has %u elements (expected %zu)", name, var##_n, \
                   (size_t)(cnt_expect));

static int load_filter(ww_ctx *ctx, ww_model *m, const blob_view &bv) {
  uint32_t cnt = 0;
  const int32_t *meta = (const int32_t *)bv.find("filter.meta", &cnt);
  if (!meta || cnt != 2) return ww_fail(ctx, WW_EBLOB, "blob lacks filter.meta");
  const int n_mel = meta[0], n_bins = meta[1];
  if (n_mel < 1 || n_mel > 40 || n_bins != WW_FFT_BINS)
    return ww_fail(ctx, WW_EBLOB, "unsupported filter geometry %d x %d (need <= 40 x 257)", n_mel, n_bins);
  NEED_F(cst, "filter.consts", 3);
  NEED_F(w, "filter.w", (size_t)n_mel * n_bins);
  NEED_F(b, "filter.b", n_mel);
  ww_filter_dev &f = m->filt;
  f.n_mel = n_mel; f.n_bins = n_bins; f.floor_v = cst[0]; f.log_off = cst[1]; f.scale = cst[2];
  std::vector<int> start(n_mel), len(n_mel), woff(n_mel);
  std::vector<float> taps;
  for (int i = 0; i < n_mel; ++i) {
    int lo = -1, hi = -1;
    for (int k = 0; k < n_bins; ++k)
      if (w[(size_t)i * n_bins + k] != 0.0f) {
        if (lo < 0) lo = k;
        hi = k;
      }
    start[i] = lo < 0 ? 0 : lo;
    len[i] = lo < 0 ? 0 : hi - lo + 1;
    woff[i] = (int)taps.size();
    for (int k = 0; k < len[i]; ++k) taps.push_back(w[(size_t)i * n_bins + start[i] + k]);
    if (len[i] > f.max_len) f.max_len = len[i];
  }
  f.total_taps = (int)taps.size();
  const int MEL_TAPS = 36;  // WW_MEL_TAPS in fft_device.h
  if (f.max_len > MEL_TAPS)
    return ww_fail(ctx, WW_EBLOB, "mel band of %d taps exceeds the kernel limit of %d", f.max_len, MEL_TAPS);
  std::vector<float> wpad((size_t)MEL_TAPS * 64, 0.f);
  for (int i = 0; i < n_mel; ++i)
    for (int k = 0; k < len[i]; ++k) wpad[(size_t)k * 64 + i] = taps[woff[i] + k];
  std::vector<float> bias(b, b + n_mel);
  std::vector<double> hann(WW_FFT_WINDOW), tw256(512), tw512(512);
  // np.hanning(M) as NumPy evaluates it: 0.5 + 0.5 cos(pi n / (M-1)), n = 1-M, 3-M, ..., M-1 (exactly symmetric)
  for (int n = 0; n < WW_FFT_WINDOW; ++n)
    hann[n] = 0.5 + 0.5 * cos(M_PI * (double)(2 * n - (WW_FFT_WINDOW - 1)) / (double)(WW_FFT_WINDOW - 1));
  for (int k = 0; k < 256; ++k) {
    tw256[2 * k] = cos(-2.0 * M_PI * k / 256.0);
    tw256[2 * k + 1] = sin(-2.0 * M_PI * k / 256.0);
    tw512[2 * k] = cos(-2.0 * M_PI * k / 512.0);
    tw512[2 * k + 1] = sin(-2.0 * M_PI * k / 512.0);
  }
  std::vector<double> tw16(512);
  for (int k1 = 0; k1 < 16; ++k1)
    for (int jj = 0; jj < 16; ++jj) {
      tw16[2 * (k1 * 16 + jj)] = cos(-2.0 * M_PI * (double)(jj * k1) / 256.0);
      tw16[2 * (k1 * 16 + jj) + 1] = sin(-2.0 * M_PI * (double)(jj * k1) / 256.0);
    }
  f.tw16 = upload(m, tw16);
  // mel filter in lane form (frontend.hip, logmel_kernel): bands sorted by width, widest first, dealt to
  // groups of 16 slots with 36 / 16 / 12 padded taps.  Weights carry the 0.5 of the real-FFT untangling
  // (exact), and a band's first bin is pulled back so that its padded taps stay inside the zero-padded
  // row of 272 magnitudes.
  {
    static const int cap[3] = {36, 16, 12}, chunk0[3] = {0, 9, 13};
    if (n_mel > 48) return ww_fail(ctx, WW_EBLOB, "mel filterbank has %d bands; the lane form holds 48", n_mel);
    std::vector<int> order(n_mel);
    for (int i = 0; i < n_mel; ++i) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return len[x] > len[y]; });
    // First bins are rounded down to a multiple of 4 when every band still fits its group's taps: the kernel
    // then reads the magnitudes 16 bytes at a time.  A ds_read_b128 is served in four groups of 16 lanes
    // ({0-3,12-15,20-27}, {4-11,16-19,28-31} and the same + 32); with lane = 4 slot + frame those hold the
    // slots {0,3,5,6}, {1,2,4,7}, {8,11,13,14}, {9,10,12,15}, and the four frames' rows lie 4 sixteen-byte bank
    // slots apart.  So the four bands of such a quad should start on different bank slots mod 4: where the
    // taps leave room a band's first bin is pulled back further to get there.
    static const int quad[4][4] = {{0, 3, 5, 6}, {1, 2, 4, 7}, {8, 11, 13, 14}, {9, 10, 12, 15}};
    bool aligned = true;
    for (int r = 0; r < n_mel; ++r) {
      const int g = r / 16, band = order[r];
      if (len[band] > cap[g])
        return ww_fail(ctx, WW_EBLOB, "mel band %d spans %d bins; the lane form takes %d for the %d widest, %d for the next 16, %d for the rest",
                       band, len[band], cap[0], 16, cap[1], cap[2]);
      int s0 = start[band] < 272 - cap[g] ? start[band] : 272 - cap[g];
      if (start[band] - (s0 & ~3) + len[band] > cap[g]) aligned = false;
    }
    std::vector<float> melV((size_t)WW_MELV_CHUNKS * 16 * 4, 0.f);
    std::vector<int> meta(3 * 16, 0xffff << 16);
    for (int g = 0; g < 3; ++g) {
      const int nb = n_mel - 16 * g < 0 ? 0 : (n_mel - 16 * g > 16 ? 16 : n_mel - 16 * g);
      int s0v[16], slotv[16], cls_n[4] = {0, 0, 0, 0};
      // least flexible bands choose their residue class first
      std::vector<int> idx(nb);
      for (int i = 0; i < nb; ++i) idx[i] = i;
      auto room = [&](int i) {  // how many steps of 4 bins band i can be pulled back beyond the plain rounding
        const int band = order[16 * g + i];
        int s0 = start[band] < 272 - cap[g] ? start[band] : 272 - cap[g];
        if (!aligned) return 0;
        s0 &= ~3;
        int n = 0;
        while (s0 - 4 * (n + 1) >= 0 && start[band] - (s0 - 4 * (n + 1)) + len[band] <= cap[g]) ++n;
        return n;
      };
      std::stable_sort(idx.begin(), idx.end(), [&](int x, int y) { return room(x) < room(y); });
      for (int i : idx) {
        const int band = order[16 * g + i];
        int s0 = start[band] < 272 - cap[g] ? start[band] : 272 - cap[g];
        if (aligned) s0 &= ~3;
        int best = 0, best_n = 1 << 30;
        for (int n = 0; n <= room(i) && n < 4; ++n) {
          const int cls = ((s0 - 4 * n) / 4) & 3;
          if (cls_n[cls] < best_n) { best_n = cls_n[cls]; best = n; }
        }
        s0 -= 4 * best;
        s0v[i] = s0;
        const int cls = (s0 / 4) & 3;
        // class member number q goes to quad q (a fifth member of a class takes any free slot below)
        slotv[i] = cls_n[cls] < 4 ? quad[cls_n[cls]][cls] : -1;
        ++cls_n[cls];
      }
      bool used[16] = {false};
      for (int i = 0; i < nb; ++i)
        if (slotv[i] >= 0) {
          if (used[slotv[i]]) slotv[i] = -1; else used[slotv[i]] = true;
        }
      for (int i = 0; i < nb; ++i)
        if (slotv[i] < 0)
          for (int sl = 0; sl < 16; ++sl)
            if (!used[sl]) { slotv[i] = sl; used[sl] = true; break; }
      for (int i = 0; i < nb; ++i) {
        const int band = order[16 * g + i], slot = slotv[i], s0 = s0v[i];
        for (int k = 0; k < len[band]; ++k) {
          const int t = start[band] - s0 + k;
          melV[((size_t)(chunk0[g] + t / 4) * 16 + slot) * 4 + t % 4] = 0.5f * w[(size_t)band * n_bins + start[band] + k];
        }
        meta[g * 16 + slot] = s0 | (band << 16);
      }
      // empty slots read (zero-weighted) magnitudes too: park each on the bank slot its quad still lacks
      for (int q = 0; q < 4; ++q) {
        bool have[4] = {false, false, false, false};
        for (int c = 0; c < 4; ++c)
          if (used[quad[q][c]]) have[((meta[g * 16 + quad[q][c]] & 0xffff) / 4) & 3] = true;
        for (int c = 0; c < 4; ++c)
          if (!used[quad[q][c]])
            for (int cls = 0; cls < 4; ++cls)
              if (!have[cls]) { have[cls] = true; meta[g * 16 + quad[q][c]] = (4 * cls) | (0xffff << 16); break; }
      }
    }
    f.melv_aligned = aligned ? 1 : 0;
    f.melV = upload(m, melV);
    f.melVmeta = upload(m, meta);
  }
  f.start = upload(m, start); f.len = upload(m, len); f.woff = upload(m, woff);
  f.w = upload(m, taps); f.bias = upload(m, bias); f.wpad = upload(m, wpad);
  f.wdense = upload(m, std::vector<float>(w, w + (size_t)n_mel * n_bins));
  f.hann = upload(m, hann); f.tw256 = upload(m, tw256); f.tw512 = upload(m, tw512);
  if (!f.wdense || !f.tw16 || !f.melV || !f.melVmeta || !f.start || !f.len || !f.woff || !f.w || !f.bias || !f.wpad || !f.hann || !f.tw256 || !f.tw512)
    return ww_fail(ctx, WW_ENOMEM, "filter upload failed");
  return WW_OK;
}

static int load_crnn(ww_ctx *ctx, ww_model *m, const blob_view &bv) {
  uint32_t cnt = 0;
  const int32_t *meta = (const int32_t *)bv.find("crnn.meta", &cnt);
  if (!meta || cnt != 14) return ww_fail(ctx, WW_EBLOB, "blob lacks crnn.meta");
  ww_crnn_dev &c = m->crnn;
  c.n_mel = meta[0]; c.T = meta[1]; c.C = meta[2]; c.KF = meta[3]; c.KT = meta[4]; c.SF = meta[5]; c.ST = meta[6];
  c.PF = meta[7]; c.PT = meta[8]; c.OF = meta[9]; c.OT = meta[10]; c.H = meta[11]; c.NOUT = meta[12]; c.HEAD = meta[13];
  const int K = c.KF * c.KT, KP = 112;
  if (c.H != 32 || c.C < 1 || c.C > 64 || c.NOUT < 1 || c.NOUT > 8 || c.OT < 1 || c.OF < 1 || K < 1 || c.n_mel != m->filt.n_mel ||
      c.T * c.n_mel > 16384)
    return ww_fail(ctx, WW_EBLOB, "unsupported CRNN geometry (C=%d H=%d K=%dx%d stride %dx%d)", c.C, c.H, c.KF, c.KT, c.SF, c.ST);
  // the geometry of wwdetect/CRNN/train.py:27-49 (every current export) runs on the kernels built for it
  c.generic = !(c.C == 32 && c.n_mel == 40 && c.T == 151 && c.KF == 5 && c.KT == 20 && c.SF == 2 && c.ST == 8 && c.PF == 1 &&
                c.PT == 6 && c.OF == 20 && c.OT == 19);
  c.FEATP = (c.OF * c.C + 63) / 64 * 64;
  NEED_F(cw, "crnn.conv_w", (size_t)c.C * K);
  NEED_F(cb, "crnn.conv_b", c.C);
  if (!c.generic) {
    std::vector<float> w4((size_t)KP / 4 * 32 * 4, 0.f);
    for (int ch = 0; ch < c.C; ++ch)
      for (int k = 0; k < K; ++k) w4[((size_t)(k / 4) * 32 + ch) * 4 + (k % 4)] = cw[(size_t)ch * K + k];
    c.conv_w = upload(m, w4);
    // the positions at a window's edges see its zero padding: the same conv with the taps over the padding cleared
    // (left: frames kt < PT; right: kt >= KT - (KT - PT - 1) = PT + 7), applied to the stream's real rows (crnn_rows_kernel)
    std::vector<float> wl(w4), wr(w4);
    for (int k = 0; k < K; ++k) {
      const int kt = k % c.KT;
      for (int ch = 0; ch < c.C; ++ch) {
        const size_t o = ((size_t)(k / 4) * 32 + ch) * 4 + (k % 4);
        if (kt < c.PT) wl[o] = 0.f;
        if (kt >= c.T - (c.OT - 1) * c.ST + c.PT) wr[o] = 0.f;   // frames past the window's last row: kt >= 151 - 144 + 6 = 13
      }
    }
    c.conv_wL = upload(m, wl);
    c.conv_wR = upload(m, wr);
    if (!c.conv_wL || !c.conv_wR) return ww_fail(ctx, WW_ENOMEM, "CRNN upload failed");
  } else {
    std::vector<float> wt((size_t)K * c.C);
    for (int ch = 0; ch < c.C; ++ch)
      for (int k = 0; k < K; ++k) wt[(size_t)k * c.C + ch] = cw[(size_t)ch * K + k];
    c.conv_wt = upload(m, wt);
    c.conv_w = c.conv_wt;
  }
  c.conv_b = upload(m, std::vector<float>(cb, cb + c.C));
  const int G = 3 * c.H;
  auto cat2 = [&](const char *a, const char *b, size_t each, std::vector<float> &out) -> int {
    uint32_t na = 0, nb = 0;
    const float *pa = (const float *)bv.find(a, &na), *pb = (const float *)bv.find(b, &nb);
    if (!pa || !pb || na != each || nb != each) return ww_fail(ctx, WW_EBLOB, "blob sections %s/%s missing or mis-sized", a, b);
    out.assign(pa, pa + each);
    out.insert(out.end(), pb, pb + each);
    return WW_OK;
  };
  std::vector<float> v;
  int rc;
  const size_t in1 = (size_t)c.OF * c.C, in2 = 2 * (size_t)c.H;
  if ((rc = cat2("crnn.g1f.wx", "crnn.g1b.wx", G * in1, v))) return rc; c.wx1 = upload(m, v);
  if (!c.generic) {
    std::vector<float> ws(v.size());
    for (size_t n = 0; n < (size_t)2 * G; ++n)
      for (size_t k = 0; k < in1; ++k) ws[((k / 4) * 2 * G + n) * 4 + (k % 4)] = v[n * in1 + k];
    c.wx1s = upload(m, ws);
    // split-bf16 mode (WW_PRECISION_BF16X3): x = hi + lo, both bf16 round-to-nearest-even
    auto rne = [](float f) -> uint16_t {
      uint32_t u;
      memcpy(&u, &f, 4);
      u += 0x7FFFu + ((u >> 16) & 1u);
      return (uint16_t)(u >> 16);
    };
    auto tof = [](uint16_t h) -> float {
      uint32_t u = (uint32_t)h << 16;
      float f;
      memcpy(&f, &u, 4);
      return f;
    };
    auto split = [&](float x, uint16_t &h, uint16_t &l) {
      h = rne(x);
      l = rne(x - tof(h));
    };
    {  // W_x1 [192][640] -> [plane][k-step 20][n-tile 12][lane = g*16 + j][8]: element e is W[nt*16 + j][ks*32 + 8 g + e]
      std::vector<unsigned short> wb((size_t)2 * 20 * 12 * 64 * 8);
      for (int ks = 0; ks < 20; ++ks)
        for (int nt = 0; nt < 12; ++nt)
          for (int ln = 0; ln < 64; ++ln)
            for (int e = 0; e < 8; ++e) {
              const int gq = ln >> 4, jj = ln & 15;
              uint16_t h, l;
              split(v[(size_t)(nt * 16 + jj) * in1 + ks * 32 + 8 * gq + e], h, l);
              const size_t o = (((size_t)ks * 12 + nt) * 64 + ln) * 8 + e;
              wb[o] = h;
              wb[(size_t)20 * 12 * 64 * 8 + o] = l;
            }
      c.wx1b = upload(m, wb);
    }
    {  // conv weights [32][5][20] -> [plane][k-step 4][m-tile 2][lane = g*16 + i][8]: group G = ks*4 + g = kf*3 + h holds
       // kt'' = 8 h + e with kt = kt'' - 2 (zero outside 0..19); G = 15 is all zero
      std::vector<unsigned short> wb((size_t)2 * 4 * 2 * 64 * 8, 0);
      for (int ks = 0; ks < 4; ++ks)
        for (int mt = 0; mt < 2; ++mt)
          for (int ln = 0; ln < 64; ++ln)
            for (int e = 0; e < 8; ++e) {
              const int gq = ln >> 4, ii = ln & 15, Gq = ks * 4 + gq;
              const int kf = Gq / 3, kt = (Gq % 3) * 8 + e - 2;
              float x = 0.f;
              if (Gq < 15 && kt >= 0 && kt < c.KT) x = cw[(size_t)(mt * 16 + ii) * K + kf * c.KT + kt];
              uint16_t h, l;
              split(x, h, l);
              const size_t o = (((size_t)ks * 2 + mt) * 64 + ln) * 8 + e;
              wb[o] = h;
              wb[(size_t)4 * 2 * 64 * 8 + o] = l;
            }
      c.cwb = upload(m, wb);
    }
    if (!c.wx1s || !c.wx1b || !c.cwb) return ww_fail(ctx, WW_ENOMEM, "CRNN upload failed");
  }
  if (c.generic) {
    std::vector<float> wp((size_t)2 * G * c.FEATP, 0.f);
    for (int r = 0; r < 2 * G; ++r) memcpy(&wp[(size_t)r * c.FEATP], &v[(size_t)r * in1], in1 * sizeof(float));
    c.wx1p = upload(m, wp);
    if (!c.wx1p) return ww_fail(ctx, WW_ENOMEM, "CRNN upload failed");
  }
  if ((rc = cat2("crnn.g1f.bx", "crnn.g1b.bx", G, v))) return rc; c.bx1 = upload(m, v);
  if ((rc = cat2("crnn.g1f.wh", "crnn.g1b.wh", (size_t)G * c.H, v))) return rc; c.wh1 = upload(m, v);
  if ((rc = cat2("crnn.g1f.bh", "crnn.g1b.bh", G, v))) return rc; c.bh1 = upload(m, v);
  if ((rc = cat2("crnn.g2f.wx", "crnn.g2b.wx", G * in2, v))) return rc; c.wx2 = upload(m, v);
  {
    std::vector<float> ws(v.size());
    for (size_t n = 0; n < (size_t)2 * G; ++n)
      for (size_t k = 0; k < in2; ++k) ws[((k / 4) * 2 * G + n) * 4 + (k % 4)] = v[n * in2 + k];
    c.wx2s = upload(m, ws);
    if (!c.wx2s) return ww_fail(ctx, WW_ENOMEM, "CRNN upload failed");
  }
  if ((rc = cat2("crnn.g2f.bx", "crnn.g2b.bx", G, v))) return rc; c.bx2 = upload(m, v);
  if ((rc = cat2("crnn.g2f.wh", "crnn.g2b.wh", (size_t)G * c.H, v))) return rc; c.wh2 = upload(m, v);
  if ((rc = cat2("crnn.g2f.bh", "crnn.g2b.bh", G, v))) return rc; c.bh2 = upload(m, v);
  NEED_F(w1, "crnn.head_w1", in2 * in2);
  NEED_F(b1, "crnn.head_b1", in2);
  NEED_F(w2, "crnn.head_w2", (size_t)c.NOUT * in2);
  NEED_F(b2, "crnn.head_b2", c.NOUT);
  c.w1 = upload(m, std::vector<float>(w1, w1 + in2 * in2));
  c.b1 = upload(m, std::vector<float>(b1, b1 + in2));
  c.w2 = upload(m, std::vector<float>(w2, w2 + c.NOUT * in2));
  c.b2 = upload(m, std::vector<float>(b2, b2 + c.NOUT));
  if (!c.conv_w || !c.conv_b || !c.wx1 || !c.bx1 || !c.wh1 || !c.bh1 || !c.wx2 || !c.bx2 || !c.wh2 || !c.bh2 || !c.w1 ||
      !c.b1 || !c.w2 || !c.b2)
    return ww_fail(ctx, WW_ENOMEM, "CRNN upload failed");
  m->info.window = c.T; m->info.n_out = c.NOUT; m->info.enc_rows = 1; m->info.enc_width = 2 * c.H;
  return WW_OK;
}

static int load_wave(ww_ctx *ctx, ww_model *m, const blob_view &bv) {
  uint32_t cnt = 0;
  const int32_t *meta = (const int32_t *)bv.find("wave.meta", &cnt);
  if (!meta || cnt != 6) return ww_fail(ctx, WW_EBLOB, "blob lacks wave.meta");
  ww_wave_dev &v = m->wave;
  v.T = meta[0]; v.n_mel = meta[1]; v.C = meta[2]; v.S = meta[3]; v.NB = meta[4]; v.NOUT = meta[5];
  if (v.C != 16 || v.S != 32 || v.T > 192 || v.T < 1 || v.n_mel > 48 || v.NOUT < 1 || v.NOUT > 16 || v.NB < 1 ||
      v.n_mel != m->filt.n_mel)
    return ww_fail(ctx, WW_EBLOB, "unsupported Wavenet geometry (T=%d C=%d S=%d)", v.T, v.C, v.S);
  const int NB = v.NB, C = v.C, S = v.S;
  const int32_t *dil = (const int32_t *)bv.find("wave.dilations", &cnt);
  if (!dil || (int)cnt != NB) return ww_fail(ctx, WW_EBLOB, "blob lacks wave.dilations");
  const int32_t *order = (const int32_t *)bv.find("wave.skip_order", &cnt);
  if (!order || (int)cnt != NB) return ww_fail(ctx, WW_EBLOB, "blob lacks wave.skip_order");
  const int32_t *has_res = (const int32_t *)bv.find("wave.has_res", &cnt);
  if (!has_res || (int)cnt != NB) return ww_fail(ctx, WW_EBLOB, "blob lacks wave.has_res");
  for (int b = 0; b < NB; ++b) {
    if (dil[b] < 1 || 2 * dil[b] > 16) return ww_fail(ctx, WW_EBLOB, "dilation %d unsupported (max 8)", dil[b]);
    if (order[b] != b) return ww_fail(ctx, WW_EBLOB, "skip connections are not summed in block order");
  }
  v.dil.assign(dil, dil + NB); v.order.assign(order, order + NB); v.has_res.assign(has_res, has_res + NB);
  NEED_F(w_in, "wave.w_in", (size_t)v.n_mel * C);
  NEED_F(b_in, "wave.b_in", C);
  NEED_F(bn_s, "wave.bn_scale", (size_t)NB * C);
  NEED_F(bn_t, "wave.bn_shift", (size_t)NB * C);
  NEED_F(w_sig, "wave.w_sig", (size_t)NB * 3 * C * C);
  NEED_F(b_sig, "wave.b_sig", (size_t)NB * C);
  NEED_F(w_tanh, "wave.w_tanh", (size_t)NB * 3 * C * C);
  NEED_F(b_tanh, "wave.b_tanh", (size_t)NB * C);
  NEED_F(w_res, "wave.w_res", (size_t)NB * C * C);
  NEED_F(b_res, "wave.b_res", (size_t)NB * C);
  NEED_F(w_skip, "wave.w_skip", (size_t)NB * C * S);
  NEED_F(b_skip, "wave.b_skip", (size_t)NB * S);
  NEED_F(dw1, "wave.det_w1", (size_t)S * S);
  NEED_F(db1, "wave.det_b1", S);
  NEED_F(dw2, "wave.det_w2", (size_t)S * v.NOUT);
  NEED_F(db2, "wave.det_b2", v.NOUT);
  // MFMA B-operand order: [k-block][kk][col][q], k = kb*16 + kk*4 + q
  std::vector<float> in4(3 * 4 * 16 * 4, 0.f);
  for (int k = 0; k < v.n_mel; ++k)
    for (int col = 0; col < C; ++col) in4[(((k / 16) * 4 + (k % 16) / 4) * 16 + col) * 4 + (k % 4)] = w_in[(size_t)k * C + col];
  std::vector<float> g4((size_t)NB * 3 * 4 * 32 * 4), bg((size_t)NB * 32), rs4((size_t)NB * 4 * 48 * 4), brs((size_t)NB * 48);
  for (int b = 0; b < NB; ++b) {
    for (int tap = 0; tap < 3; ++tap)
      for (int ch = 0; ch < C; ++ch)
        for (int col = 0; col < 32; ++col) {
          float val = col < 16 ? w_sig[(((size_t)b * 3 + tap) * C + ch) * C + col]
                               : w_tanh[(((size_t)b * 3 + tap) * C + ch) * C + col - 16];
          g4[((((size_t)b * 3 + tap) * 4 + ch / 4) * 32 + col) * 4 + (ch % 4)] = val;
        }
    for (int col = 0; col < 16; ++col) {
      bg[(size_t)b * 32 + col] = b_sig[(size_t)b * C + col];
      bg[(size_t)b * 32 + 16 + col] = b_tanh[(size_t)b * C + col];
    }
    for (int ch = 0; ch < C; ++ch)
      for (int col = 0; col < 48; ++col) {
        float val = col < 16 ? w_res[((size_t)b * C + ch) * C + col] : w_skip[((size_t)b * C + ch) * S + col - 16];
        rs4[(((size_t)b * 4 + ch / 4) * 48 + col) * 4 + (ch % 4)] = val;
      }
    for (int col = 0; col < 48; ++col) brs[(size_t)b * 48 + col] = col < 16 ? b_res[(size_t)b * C + col] : b_skip[(size_t)b * S + col - 16];
  }
  std::vector<float> d1((size_t)2 * 4 * 32 * 4), d2((size_t)2 * 4 * 16 * 4, 0.f), d2b(16, 0.f);
  for (int k = 0; k < S; ++k) {
    for (int col = 0; col < S; ++col) d1[(((size_t)(k / 16) * 4 + (k % 16) / 4) * 32 + col) * 4 + (k % 4)] = dw1[(size_t)k * S + col];
    for (int col = 0; col < v.NOUT; ++col) d2[(((size_t)(k / 16) * 4 + (k % 16) / 4) * 16 + col) * 4 + (k % 4)] = dw2[(size_t)k * v.NOUT + col];
  }
  for (int col = 0; col < v.NOUT; ++col) d2b[col] = db2[col];
  // split-bf16 A operands (wavenet.hip, SPLIT_BF16; transposed formulation: rows = output channels).
  // slot = (kstep*2 + {sig,tanh})*2 + {0,1} for the gate conv (k-step 0 = tap 2, k-step 1 = tap 0 | tap 1: the two
  // delayed taps), 8 + mtile*2 + {0,1} for res | skip.  Lane (i = lane & 15, kg = lane >> 4) holds 8 k-slots = two
  // groups of the 4 channels 4 kg .. 4 kg + 3 for output row 16 mtile + i.  Where only one tap (or the gate product) is at
  // hand the second group carries the hi x lo product instead of zeros: 5 MFMAs per gate and 2 per res | skip m-tile give
  // all three split products plus the bias (16 MFMAs per block and tile, not 21).
  {
    auto bf16_rne = [](float f) -> uint16_t {
      uint32_t u;
      memcpy(&u, &f, 4);
      u += 0x7FFFu + ((u >> 16) & 1u);
      return (uint16_t)(u >> 16);
    };
    auto bf16_f = [](uint16_t h) -> float {
      uint32_t u = (uint32_t)h << 16;
      float f;
      memcpy(&f, &u, 4);
      return f;
    };
    std::vector<uint16_t> pk((size_t)NB * 14 * 64 * 8, 0);
    auto split = [&](float wv, uint16_t &hi, uint16_t &lo) {
      hi = bf16_rne(wv);
      lo = bf16_rne(wv - bf16_f(hi));
    };
    auto put = [&](int b, int slot, int lane, int q, uint16_t val) { pk[((((size_t)b * 14 + slot) * 64) + lane) * 8 + q] = val; };
    for (int b = 0; b < NB; ++b) {
      for (int lane = 0; lane < 64; ++lane) {
        const int i = lane & 15, kg = lane >> 4;
        for (int q = 0; q < 8; ++q) {
          const int ch = 4 * kg + (q & 3);
          for (int mt = 0; mt < 2; ++mt) {
            // the gates are evaluated with v_exp_f32 (= exp2): sigmoid(s) = 1 / (1 + exp2(-log2e s)),
            // tanh(t) = 1 - 2 / (1 + exp2(2 log2e t)) - the factors ride in the weights and biases
            const float sc = mt == 0 ? -1.4426950408889634f : 2.8853900817779268f;
            const float *wsrc = mt == 0 ? w_sig : w_tanh;
            auto wt = [&](int tap) { return wsrc[(((size_t)b * 3 + tap) * C + ch) * C + i] * sc; };
            uint16_t hi, lo;
            // k-step 0, slot "hh": (hi of tap 2 | hi of tap 2) against B = (u_hi | u_lo);
            //           slot "lb": (lo of tap 2 | bias hi, bias lo in k-slots 4, 5 of lane group 0) against B = (u_hi | 1, 1, 0, 0)
            split(wt(2), hi, lo);
            put(b, (0 * 2 + mt) * 2 + 0, lane, q, hi);
            if (q < 4) {
              put(b, (0 * 2 + mt) * 2 + 1, lane, q, lo);
            } else if (kg == 0 && q < 6) {
              uint16_t bh, bl;
              split((mt == 0 ? b_sig : b_tanh)[(size_t)b * C + i] * sc, bh, bl);
              put(b, (0 * 2 + mt) * 2 + 1, lane, q, q == 4 ? bh : bl);
            }
            // k-step 1: (tap 0 | tap 1), hi and lo slots, against B = the two delayed rows (hi plane, then lo plane)
            split(wt(q < 4 ? 0 : 1), hi, lo);
            put(b, (1 * 2 + mt) * 2 + 0, lane, q, hi);
            put(b, (1 * 2 + mt) * 2 + 1, lane, q, lo);
          }
          for (int mt = 0; mt < 3; ++mt) {
            // res | skip: slot "hh" = (hi | hi) against B = (g_hi | g_lo), slot "lb" = (lo | bias hi, bias lo) against (g_hi | 1, 1, 0, 0)
            const float wv = mt == 0 ? (has_res[b] ? w_res[((size_t)b * C + ch) * C + i] : 0.f) : w_skip[((size_t)b * C + ch) * S + (mt - 1) * 16 + i];
            uint16_t hi, lo;
            split(wv, hi, lo);
            put(b, 8 + mt * 2 + 0, lane, q, hi);
            if (q < 4) {
              put(b, 8 + mt * 2 + 1, lane, q, lo);
            } else if (kg == 0 && q < 6) {
              uint16_t bh, bl;
              split(mt == 0 ? (has_res[b] ? b_res[(size_t)b * C + i] : 0.f) : b_skip[(size_t)b * S + (mt - 1) * 16 + i], bh, bl);
              put(b, 8 + mt * 2 + 1, lane, q, q == 4 ? bh : bl);
            }
          }
        }
      }
    }
    v.wpk = upload(m, pk);  // one page per block (wavenet.hip: WV_PAGE_U4 16-byte units)
    if (!v.wpk) return ww_fail(ctx, WW_ENOMEM, "Wavenet upload failed");
  }
  v.w_in = upload(m, in4); v.b_in = upload(m, std::vector<float>(b_in, b_in + C));
  v.bn_s = upload(m, std::vector<float>(bn_s, bn_s + (size_t)NB * C));
  v.bn_t = upload(m, std::vector<float>(bn_t, bn_t + (size_t)NB * C));
  v.w_gate = upload(m, g4); v.b_gate = upload(m, bg); v.w_rs = upload(m, rs4); v.b_rs = upload(m, brs);
  v.d_w1 = upload(m, d1); v.d_b1 = upload(m, std::vector<float>(db1, db1 + S));
  v.d_w2 = upload(m, d2); v.d_b2 = upload(m, d2b);
  if (!v.w_in || !v.b_in || !v.bn_s || !v.bn_t || !v.w_gate || !v.b_gate || !v.w_rs ||
      !v.b_rs || !v.d_w1 || !v.d_b1 || !v.d_w2 || !v.d_b2)
    return ww_fail(ctx, WW_ENOMEM, "Wavenet upload failed");
  m->info.window = v.T; m->info.n_out = v.NOUT; m->info.enc_rows = v.T; m->info.enc_width = S;
  return WW_OK;
}

extern "C" {

int ww_model_load(ww_ctx *ctx, const void *blob, size_t len, ww_model **out) {
  WW_GUARD_BEGIN
  if (!ctx || !blob || !out) return ww_fail(ctx, WW_EINVAL, "NULL argument");
  *out = nullptr;
  if (len < 16) return ww_fail(ctx, WW_EBLOB, "blob too short");
  uint32_t h[4];
  memcpy(h, blob, sizeof h);  // the caller's buffer need not be 4-byte aligned
  if (h[0] != 0x42485757u || h[1] != 1u) return ww_fail(ctx, WW_EBLOB, "bad blob magic/version");
  blob_view bv = {(const uint8_t *)blob, len, h[3], h[2]};
  if (16 + 32 * (size_t)bv.n > len) return ww_fail(ctx, WW_EBLOB, "section table exceeds blob");
  if (bv.kind != WW_KIND_CRNN && bv.kind != WW_KIND_WAVENET) return ww_fail(ctx, WW_EBLOB, "unknown model kind %u", bv.kind);
  WW_ON_DEVICE(ctx, dev_scope);  // the caller's current device is left as it was
  ww_model *m = new ww_model();
  ww_scoped<ww_model, ww_model_free> own(m);
  m->ctx = ctx;
  m->kind = (int)bv.kind;
  int rc = load_filter(ctx, m, bv);
  if (rc == WW_OK) rc = bv.kind == WW_KIND_CRNN ? load_crnn(ctx, m, bv) : load_wave(ctx, m, bv);
  if (rc != WW_OK) return rc;  // (`own` frees what was uploaded so far)
  m->info.kind = m->kind;
  m->info.n_mel = m->filt.n_mel;
  m->info.n_bins = m->filt.n_bins;
  *out = own.release();
  return WW_OK;
  WW_GUARD_END(ctx)
}

int ww_model_free(ww_model *m) {
  WW_GUARD_BEGIN
  if (!m) return WW_OK;
  if (m->ctx) {
    ww_device_scope dev_scope(m->ctx->device);
    hipStreamSynchronize(m->ctx->stream);
  }
  for (void *p : m->allocs) hipFree(p);
  delete m;
  return WW_OK;
  WW_GUARD_END(nullptr)
}

int ww_model_get_info(const ww_model *m, ww_model_info *out) {
  WW_GUARD_BEGIN
  if (!m || !out) return WW_EINVAL;
  *out = m->info;
  return WW_OK;
  WW_GUARD_END(m ? m->ctx : nullptr)
}

int ww_model_set_precision(ww_model *m, int precision) {
  WW_GUARD_BEGIN
  if (!m) return WW_EINVAL;
  if (precision != WW_PRECISION_FP32 && precision != WW_PRECISION_BF16X3)
    return ww_fail(m->ctx, WW_EINVAL, "unknown precision %d", precision);
  m->precision = precision;
  return WW_OK;
  WW_GUARD_END(m ? m->ctx : nullptr)
}

int ww_model_set_option(ww_model *m, int key, int64_t value) {
  WW_GUARD_BEGIN
  if (!m) return WW_EINVAL;
  if (value < 0 || value > 0x7fffffff) return ww_fail(m->ctx, WW_EINVAL, "option value %lld out of range", (long long)value);
  switch (key) {
    case WW_OPT_CRNN_SPLIT_AT: m->opt_split_at = (int)value; return WW_OK;
    case WW_OPT_CRNN_SLIDE_MIN: m->opt_slide_min = (int)value; return WW_OK;
    case WW_OPT_CRNN_TAIL_MFMA:
      // (3: development builds with -DWW_TAIL16H=1 - the hoisted-projection probe of round 6; the shipped library treats it as 2)
      if (value > 3) return ww_fail(m->ctx, WW_EINVAL, "WW_OPT_CRNN_TAIL_MFMA takes 0 (never), 1 (from 9,216 windows per launch) or 2 (always)");
      m->opt_tail_mfma = (int)value;
      return WW_OK;
    case WW_OPT_WAVENET_ROWMAJOR: m->opt_wave_rowmajor = value != 0; return WW_OK;
    default: return ww_fail(m->ctx, WW_EINVAL, "unknown model option %d", key);
  }
  WW_GUARD_END(m ? m->ctx : nullptr)
}

int64_t ww_num_frames(int64_t n, int32_t hop) WW_NOTHROW {
  if (hop <= 0 || n < WW_FFT_WINDOW) return 0;
  return (n - WW_FFT_WINDOW) / hop + 1;
}

}  // extern "C"

// ------------------------------------------------------------------------------------------
// front end entry points
// ------------------------------------------------------------------------------------------
static int check_fp(ww_ctx *ctx, const ww_frontend_params *fp, bool need_div) {
  if (!fp) return ww_fail(ctx, WW_EINVAL, "frontend params are NULL");
  if (fp->hop <= 0 || fp->hop > WW_FFT_WINDOW) return ww_fail(ctx, WW_EINVAL, "hop %d out of range (1..512)", fp->hop);
  if (need_div && !(fp->pcm_divisor > 0.f)) return ww_fail(ctx, WW_EINVAL, "pcm_divisor must be positive");
  return WW_OK;
}

static int logmel_host(ww_ctx *ctx, const ww_model *m, const void *samples, size_t elt, const int64_t *sample_offs, int n_utt,
                       const ww_frontend_params *fp, float *mel, int64_t *frame_offs) {
  if (!ctx || !m || !sample_offs || !frame_offs) return ww_fail(ctx, WW_EINVAL, "NULL argument");
  if (n_utt < 0) return ww_fail(ctx, WW_EINVAL, "negative utterance count");
  int rc = check_fp(ctx, fp, elt == 2);
  if (rc) return rc;
  WW_ON_DEVICE(ctx, dev_scope);  // the caller's current device is left as it was
  frame_offs[0] = 0;
  int64_t max_f = 0;
  for (int u = 0; u < n_utt; ++u) {
    const int64_t n = sample_offs[u + 1] - sample_offs[u];
    if (n < 0) return ww_fail(ctx, WW_EINVAL, "sample_offs not ascending at %d", u);
    const int64_t nf = ww_num_frames(n, fp->hop);
    frame_offs[u + 1] = frame_offs[u] + nf;
    if (nf > max_f) max_f = nf;
  }
  const int64_t total_f = frame_offs[n_utt];
  if (n_utt == 0 || total_f == 0) return WW_OK;
  if (!samples || !mel) return ww_fail(ctx, WW_EINVAL, "NULL sample or mel buffer");
  const int64_t base = sample_offs[0], total_s = sample_offs[n_utt] - base;
  const size_t b_s = ww_bump::need((size_t)total_s + 16, elt), b_o = ww_bump::need((size_t)n_utt + 1, 8);
  const size_t b_m = ww_bump::need((size_t)total_f * m->filt.n_mel, 4);
  if (b_s + 2 * b_o + b_m <= WW_SMALL_IO_BYTES) {
    ww_small_io io(ctx);
    if ((rc = io.init(b_s + 2 * b_o + b_m))) return rc;
    char *h_s = io.take<char>(b_s);
    int64_t *h_so = io.take<int64_t>(n_utt + 1), *h_fo = io.take<int64_t>(n_utt + 1);
    float *h_mel = io.take<float>((size_t)total_f * m->filt.n_mel);
    memcpy(h_s, (const char *)samples + (size_t)base * elt, (size_t)total_s * elt);
    memset(h_s + (size_t)total_s * elt, 0, 16 * elt);
    for (int u = 0; u <= n_utt; ++u) h_so[u] = sample_offs[u] - base;
    memcpy(h_fo, frame_offs, sizeof(int64_t) * (n_utt + 1));
    rc = ww_k_logmel(ctx, m, elt == 2 ? (const int16_t *)io.dv(h_s) : nullptr, elt == 4 ? (const float *)io.dv(h_s) : nullptr,
                     io.dv(h_so), io.dv(h_fo), n_utt, total_f, max_f, fp, io.dv(h_mel), 0, total_s);
    if (rc) return rc;
    WW_HIP(ctx, hipStreamSynchronize(ctx->stream));
    memcpy(mel, h_mel, (size_t)total_f * m->filt.n_mel * 4);
    return WW_OK;
  }
  if ((rc = ww_ensure(ctx, ctx->dev, b_s + 2 * b_o + b_m, false))) return rc;
  ww_bump bump(ctx->dev.ptr, ctx->dev.cap);
  char *d_s = bump.take<char>(b_s);
  int64_t *d_so = bump.take<int64_t>(n_utt + 1), *d_fo = bump.take<int64_t>(n_utt + 1);
  float *d_mel = bump.take<float>((size_t)total_f * m->filt.n_mel);
  std::vector<int64_t> so(n_utt + 1);
  for (int u = 0; u <= n_utt; ++u) so[u] = sample_offs[u] - base;
  WW_HIP(ctx, hipMemcpyAsync(d_s, (const char *)samples + (size_t)base * elt, (size_t)total_s * elt, hipMemcpyHostToDevice, ctx->stream));
  WW_HIP(ctx, hipMemcpyAsync(d_so, so.data(), sizeof(int64_t) * (n_utt + 1), hipMemcpyHostToDevice, ctx->stream));
  WW_HIP(ctx, hipMemcpyAsync(d_fo, frame_offs, sizeof(int64_t) * (n_utt + 1), hipMemcpyHostToDevice, ctx->stream));
  rc = ww_k_logmel(ctx, m, elt == 2 ? (const int16_t *)d_s : nullptr, elt == 4 ? (const float *)d_s : nullptr, d_so, d_fo,
                   n_utt, total_f, max_f, fp, d_mel, 0, total_s);
  if (rc) return rc;
  WW_HIP(ctx, hipMemcpyAsync(mel, d_mel, (size_t)total_f * m->filt.n_mel * 4, hipMemcpyDeviceToHost, ctx->stream));
  WW_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return WW_OK;
}

extern "C" {

int ww_logmel(ww_ctx *ctx, const ww_model *m, const int16_t *pcm, const int64_t *sample_offs, int32_t n_utt,
              const ww_frontend_params *fp, float *mel, int64_t *frame_offs) {
  WW_GUARD_BEGIN
  return logmel_host(ctx, m, pcm, 2, sample_offs, n_utt, fp, mel, frame_offs);
  WW_GUARD_END(ctx)
}

int ww_logmel_f32(ww_ctx *ctx, const ww_model *m, const float *samples, const int64_t *sample_offs, int32_t n_utt,
                  const ww_frontend_params *fp, float *mel, int64_t *frame_offs) {
  WW_GUARD_BEGIN
  return logmel_host(ctx, m, samples, 4, sample_offs, n_utt, fp, mel, frame_offs);
  WW_GUARD_END(ctx)
}

int ww_logmel_dev(ww_ctx *ctx, const ww_model *m, const int16_t *d_pcm, const int64_t *d_sample_offs,
                  const int64_t *d_frame_offs, int32_t n_utt, int64_t total_frames, int64_t max_frames_per_utt,
                  const ww_frontend_params *fp, float *d_mel) {
  WW_GUARD_BEGIN
  if (!ctx || !m || !d_pcm || !d_sample_offs || !d_frame_offs || !d_mel) return ww_fail(ctx, WW_EINVAL, "NULL argument");
  if (((uintptr_t)d_pcm & 15) != 0) return ww_fail(ctx, WW_EINVAL, "d_pcm must be 16-byte aligned");
  int rc = check_fp(ctx, fp, true);
  if (rc) return rc;
  WW_ON_DEVICE(ctx, dev);
  return ww_k_logmel(ctx, m, d_pcm, nullptr, d_sample_offs, d_frame_offs, n_utt, total_frames, max_frames_per_utt, fp, d_mel);
  WW_GUARD_END(ctx)
}

int ww_stft_mag(ww_ctx *ctx, const ww_model *m, const float *frames, int64_t n, int32_t precise, float *mag) {
  WW_GUARD_BEGIN
  if (!ctx || !m) return ww_fail(ctx, WW_EINVAL, "NULL argument");
  if (n < 0) return ww_fail(ctx, WW_EINVAL, "negative frame count");
  if (n == 0) return WW_OK;
  if (!frames || !mag) return ww_fail(ctx, WW_EINVAL, "NULL buffer");
  WW_ON_DEVICE(ctx, dev_scope);  // the caller's current device is left as it was
  const size_t b_f = ww_bump::need((size_t)n * WW_FFT_WINDOW, 4), b_m = ww_bump::need((size_t)n * WW_FFT_BINS, 4);
  int rc;
  if (b_f + b_m <= WW_SMALL_IO_BYTES) {
    ww_small_io io(ctx);
    if ((rc = io.init(b_f + b_m))) return rc;
    float *h_f = io.take<float>((size_t)n * WW_FFT_WINDOW), *h_m = io.take<float>((size_t)n * WW_FFT_BINS);
    memcpy(h_f, frames, (size_t)n * WW_FFT_WINDOW * 4);
    if ((rc = ww_k_stft_mag(ctx, m, io.dv(h_f), n, precise, io.dv(h_m)))) return rc;
    WW_HIP(ctx, hipStreamSynchronize(ctx->stream));
    memcpy(mag, h_m, (size_t)n * WW_FFT_BINS * 4);
    return WW_OK;
  }
  rc = ww_ensure(ctx, ctx->dev, b_f + b_m, false);
  if (rc) return rc;
  ww_bump bump(ctx->dev.ptr, ctx->dev.cap);
  float *d_f = bump.take<float>((size_t)n * WW_FFT_WINDOW), *d_m = bump.take<float>((size_t)n * WW_FFT_BINS);
  WW_HIP(ctx, hipMemcpyAsync(d_f, frames, (size_t)n * WW_FFT_WINDOW * 4, hipMemcpyHostToDevice, ctx->stream));
  if ((rc = ww_k_stft_mag(ctx, m, d_f, n, precise, d_m))) return rc;
  WW_HIP(ctx, hipMemcpyAsync(mag, d_m, (size_t)n * WW_FFT_BINS * 4, hipMemcpyDeviceToHost, ctx->stream));
  WW_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return WW_OK;
  WW_GUARD_END(ctx)
}

}  // extern "C"

extern "C" {

int ww_filter_apply(ww_ctx *ctx, const ww_model *m, const float *mag, int64_t n, float *mel) {
  WW_GUARD_BEGIN
  if (!ctx || !m) return ww_fail(ctx, WW_EINVAL, "NULL argument");
  if (n < 0) return ww_fail(ctx, WW_EINVAL, "negative row count");
  if (n == 0) return WW_OK;
  if (!mag || !mel) return ww_fail(ctx, WW_EINVAL, "NULL buffer");
  WW_ON_DEVICE(ctx, dev_scope);  // the caller's current device is left as it was
  const int NBN = m->filt.n_bins, F = m->filt.n_mel;
  const size_t b_a = ww_bump::need((size_t)n * NBN, 4), b_b = ww_bump::need((size_t)n * F, 4);
  int rc;
  if (b_a + b_b <= WW_SMALL_IO_BYTES) {
    ww_small_io io(ctx);
    if ((rc = io.init(b_a + b_b))) return rc;
    float *h_a = io.take<float>((size_t)n * NBN), *h_b = io.take<float>((size_t)n * F);
    memcpy(h_a, mag, (size_t)n * NBN * 4);
    if ((rc = ww_k_mel_only(ctx, m, io.dv(h_a), n, io.dv(h_b)))) return rc;
    WW_HIP(ctx, hipStreamSynchronize(ctx->stream));
    memcpy(mel, h_b, (size_t)n * F * 4);
    return WW_OK;
  }
  rc = ww_ensure(ctx, ctx->dev, b_a + b_b, false);
  if (rc) return rc;
  ww_bump bump(ctx->dev.ptr, ctx->dev.cap);
  float *d_a = bump.take<float>((size_t)n * NBN), *d_b = bump.take<float>((size_t)n * F);
  WW_HIP(ctx, hipMemcpyAsync(d_a, mag, (size_t)n * NBN * 4, hipMemcpyHostToDevice, ctx->stream));
  if ((rc = ww_k_mel_only(ctx, m, d_a, n, d_b))) return rc;
  WW_HIP(ctx, hipMemcpyAsync(mel, d_b, (size_t)n * F * 4, hipMemcpyDeviceToHost, ctx->stream));
  WW_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return WW_OK;
  WW_GUARD_END(ctx)
}

int ww_detect(ww_ctx *ctx, const ww_model *m, const float *enc, int32_t n, float *out) {
  WW_GUARD_BEGIN
  if (!ctx || !m) return ww_fail(ctx, WW_EINVAL, "NULL argument");
  if (n < 0) return ww_fail(ctx, WW_EINVAL, "negative row count");
  if (n == 0) return WW_OK;
  if (!enc || !out) return ww_fail(ctx, WW_EINVAL, "NULL buffer");
  WW_ON_DEVICE(ctx, dev_scope);  // the caller's current device is left as it was
  const size_t per = (size_t)m->info.enc_rows * m->info.enc_width;
  const size_t b_a = ww_bump::need((size_t)n * per, 4), b_b = ww_bump::need((size_t)n * m->info.n_out, 4);
  int rc;
  if (b_a + b_b <= WW_SMALL_IO_BYTES) {
    ww_small_io io(ctx);
    if ((rc = io.init(b_a + b_b))) return rc;
    float *h_a = io.take<float>((size_t)n * per), *h_b = io.take<float>((size_t)n * m->info.n_out);
    memcpy(h_a, enc, (size_t)n * per * 4);
    rc = m->kind == WW_KIND_CRNN ? ww_k_crnn_detect(ctx, m, io.dv(h_a), n, io.dv(h_b)) : ww_k_wave_detect(ctx, m, io.dv(h_a), n, io.dv(h_b));
    if (rc) return rc;
    WW_HIP(ctx, hipStreamSynchronize(ctx->stream));
    memcpy(out, h_b, (size_t)n * m->info.n_out * 4);
    return WW_OK;
  }
  rc = ww_ensure(ctx, ctx->dev, b_a + b_b, false);
  if (rc) return rc;
  ww_bump bump(ctx->dev.ptr, ctx->dev.cap);
  float *d_a = bump.take<float>((size_t)n * per), *d_b = bump.take<float>((size_t)n * m->info.n_out);
  WW_HIP(ctx, hipMemcpyAsync(d_a, enc, (size_t)n * per * 4, hipMemcpyHostToDevice, ctx->stream));
  rc = m->kind == WW_KIND_CRNN ? ww_k_crnn_detect(ctx, m, d_a, n, d_b) : ww_k_wave_detect(ctx, m, d_a, n, d_b);
  if (rc) return rc;
  WW_HIP(ctx, hipMemcpyAsync(out, d_b, (size_t)n * m->info.n_out * 4, hipMemcpyDeviceToHost, ctx->stream));
  WW_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return WW_OK;
  WW_GUARD_END(ctx)
}

}  // extern "C"

// ------------------------------------------------------------------------------------------
// encode + detect entry points
// ------------------------------------------------------------------------------------------
static size_t model_ws(const ww_model *m, int nw) {
  return m->kind == WW_KIND_CRNN ? ww_crnn_workspace(m, nw) : ww_wave_workspace(m, nw);
}

// ws: model scratch inside the context's device arena (ww_ensure'd by the caller for model_ws(m, nw) or more): its capacity is
// what is left of the arena behind it
static int model_forward(ww_ctx *ctx, const ww_model *m, const float *d_mel, int64_t mel_rows, const int64_t *d_row,
                         const int32_t *d_valid, int64_t row0, int hop, int valid_const, int nw, void *ws, float *d_out,
                         float *d_enc) {
  const char *lo = (const char *)ctx->dev.ptr, *p = (const char *)ws;
  const size_t cap = (p >= lo && p <= lo + ctx->dev.cap) ? (size_t)(lo + ctx->dev.cap - p) : 0;
  return m->kind == WW_KIND_CRNN
             ? ww_k_crnn_forward(ctx, m, d_mel, mel_rows, d_row, d_valid, row0, hop, valid_const, nw, ws, cap, d_out, d_enc)
             : ww_k_wave_forward(ctx, m, d_mel, mel_rows, d_row, d_valid, row0, hop, valid_const, nw, ws, cap, d_out, d_enc);
}

// windows are processed in chunks so that the workspace stays bounded
#ifndef WW_MAX_CHUNK
#define WW_MAX_CHUNK 16384
#endif

static int forward_host(ww_ctx *ctx, const ww_model *m, const float *mel, int64_t rows, int hop, int64_t nw, float *out,
                        float *enc) {
  const int T = m->info.window, F = m->info.n_mel, NO = m->info.n_out;
  const size_t enc_per = (size_t)m->info.enc_rows * m->info.enc_width;
  WW_ON_DEVICE(ctx, dev_scope);  // the caller's current device is left as it was
  // Small calls (the reference's per-frame use: one window in, one posterior out, utils/time_tf_models.py) are
  // latency-bound, and a copy-engine operation costs more than the few KB are worth: the window is staged in pinned
  // host memory and the kernels read it over the bus themselves; the posteriors (and the encoder output) are stored
  // into pinned host memory by the kernels.  No host-to-device or device-to-host copy on the path.
  {
    const size_t p_mel = ww_bump::need((size_t)rows * F, 4), p_out = ww_bump::need((size_t)nw * NO, 4);
    const size_t p_enc = enc ? ww_bump::need((size_t)nw * enc_per, 4) : 0;
    if (p_mel + p_out + p_enc <= WW_SMALL_IO_BYTES && nw <= 64) {
      ww_small_io io(ctx);
      int rc = io.init(p_mel + p_out + p_enc);
      if (rc) return rc;
      if ((rc = ww_ensure(ctx, ctx->dev, model_ws(m, (int)nw) + 1024, false))) return rc;
      float *h_mel = io.take<float>((size_t)rows * F), *h_out = io.take<float>((size_t)nw * NO);
      float *h_enc = enc ? io.take<float>((size_t)nw * enc_per) : nullptr;
      memcpy(h_mel, mel, (size_t)rows * F * 4);
      ww_bump db(ctx->dev.ptr, ctx->dev.cap);
      void *ws = db.take<char>(model_ws(m, (int)nw));
      rc = model_forward(ctx, m, io.dv(h_mel), rows, nullptr, nullptr, 0, hop, T, (int)nw, ws, io.dv(h_out), enc ? io.dv(h_enc) : nullptr);
      if (rc) return rc;
      WW_HIP(ctx, hipStreamSynchronize(ctx->stream));
      memcpy(out, h_out, (size_t)nw * NO * 4);
      if (enc) memcpy(enc, h_enc, (size_t)nw * enc_per * 4);
      return WW_OK;
    }
  }
  const int chunk = nw < WW_MAX_CHUNK ? (int)nw : WW_MAX_CHUNK;
  const size_t b_mel = ww_bump::need((size_t)rows * F, 4), b_out = ww_bump::need((size_t)nw * NO, 4);
  const size_t b_enc = enc ? ww_bump::need((size_t)chunk * enc_per, 4) : 0;
  const size_t b_ws = model_ws(m, chunk);
  int rc = ww_ensure(ctx, ctx->dev, b_mel + b_out + b_enc + b_ws + 1024, false);
  if (rc) return rc;
  ww_bump bump(ctx->dev.ptr, ctx->dev.cap);
  float *d_mel = bump.take<float>((size_t)rows * F);
  float *d_out = bump.take<float>((size_t)nw * NO);
  float *d_enc = enc ? bump.take<float>((size_t)chunk * enc_per) : nullptr;
  void *ws = bump.take<char>(b_ws);
  WW_HIP(ctx, hipMemcpyAsync(d_mel, mel, (size_t)rows * F * 4, hipMemcpyHostToDevice, ctx->stream));
  for (int64_t w0 = 0; w0 < nw; w0 += chunk) {
    const int n = (int)((nw - w0) < chunk ? (nw - w0) : chunk);
    rc = model_forward(ctx, m, d_mel, rows, nullptr, nullptr, w0 * hop, hop, T, n, ws, d_out + (size_t)w0 * NO, d_enc);
    if (rc) return rc;
    if (enc) {
      WW_HIP(ctx, hipMemcpyAsync(enc + (size_t)w0 * enc_per, d_enc, (size_t)n * enc_per * 4, hipMemcpyDeviceToHost, ctx->stream));
      WW_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
  }
  WW_HIP(ctx, hipMemcpyAsync(out, d_out, (size_t)nw * NO * 4, hipMemcpyDeviceToHost, ctx->stream));
  WW_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return WW_OK;
}

__global__ void iota_offs_kernel(int64_t *sample_offs, int64_t *frame_offs, int n, int64_t samples, int64_t frames) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i <= n) {
    sample_offs[i] = (int64_t)i * samples;
    frame_offs[i] = (int64_t)i * frames;
  }
}

extern "C" {

int ww_forward_enc(ww_ctx *ctx, const ww_model *m, const float *windows, int32_t nw, float *out, float *enc) {
  WW_GUARD_BEGIN
  if (!ctx || !m) return ww_fail(ctx, WW_EINVAL, "NULL argument");
  if (nw < 0) return ww_fail(ctx, WW_EINVAL, "negative window count");
  if (nw == 0) return WW_OK;
  if (!windows || !out) return ww_fail(ctx, WW_EINVAL, "NULL buffer");
  // B stacked windows are one mel sequence of B*T rows read with hop = T
  return forward_host(ctx, m, windows, (int64_t)nw * m->info.window, m->info.window, nw, out, enc);
  WW_GUARD_END(ctx)
}

int ww_forward(ww_ctx *ctx, const ww_model *m, const float *windows, int32_t nw, float *out) {
  WW_GUARD_BEGIN
  return ww_forward_enc(ctx, m, windows, nw, out, nullptr);
  WW_GUARD_END(ctx)
}

int ww_slide_forward(ww_ctx *ctx, const ww_model *m, const float *mel, int64_t rows, int32_t hop, float *out,
                     int64_t *n_windows) {
  WW_GUARD_BEGIN
  if (!ctx || !m || !n_windows) return ww_fail(ctx, WW_EINVAL, "NULL argument");
  if (hop <= 0) return ww_fail(ctx, WW_EINVAL, "hop must be positive");
  if (rows < 0) return ww_fail(ctx, WW_EINVAL, "negative row count");
  const int T = m->info.window;
  const int64_t nw = rows >= T ? (rows - T) / hop + 1 : 0;
  *n_windows = nw;
  if (nw == 0) return WW_OK;
  if (!mel || !out) return ww_fail(ctx, WW_EINVAL, "NULL buffer");
  return forward_host(ctx, m, mel, rows, hop, nw, out, nullptr);
  WW_GUARD_END(ctx)
}

int ww_forward_windows_dev(ww_ctx *ctx, const ww_model *m, const float *d_mel, int64_t mel_rows, const int64_t *d_win_row,
                           const int32_t *d_win_valid, int32_t nw, float *d_out) {
  WW_GUARD_BEGIN
  if (!ctx || !m || !d_mel || !d_out) return ww_fail(ctx, WW_EINVAL, "NULL argument");
  if (nw < 0) return ww_fail(ctx, WW_EINVAL, "negative window count");
  if (nw == 0) return WW_OK;
  if (!d_win_row || !d_win_valid) return ww_fail(ctx, WW_EINVAL, "window descriptors are NULL");
  WW_ON_DEVICE(ctx, dev);
  const int NO = m->info.n_out;
  const int chunk = nw < WW_MAX_CHUNK ? nw : WW_MAX_CHUNK;
  int rc = ww_ensure(ctx, ctx->dev, model_ws(m, chunk) + 1024, false);
  if (rc) return rc;
  for (int w0 = 0; w0 < nw; w0 += chunk) {
    const int n = (nw - w0) < chunk ? (nw - w0) : chunk;
    rc = model_forward(ctx, m, d_mel, mel_rows, d_win_row + w0, d_win_valid + w0, 0, 0, 0, n, ctx->dev.ptr,
                       d_out + (size_t)w0 * NO, nullptr);
    if (rc) return rc;
  }
  return WW_OK;
  WW_GUARD_END(ctx)
}

int ww_forward_segments_dev(ww_ctx *ctx, const ww_model *m, const float *d_mel, int64_t mel_rows, const int64_t *seg_row0,
                            const int32_t *seg_nw, int32_t n_seg, int32_t hop, float *d_out) {
  WW_GUARD_BEGIN
  if (!ctx || !m || !d_mel || !d_out) return ww_fail(ctx, WW_EINVAL, "NULL argument");
  if (n_seg < 0) return ww_fail(ctx, WW_EINVAL, "negative sequence count");
  if (hop <= 0) return ww_fail(ctx, WW_EINVAL, "hop must be positive");
  if (n_seg == 0) return WW_OK;
  if (!seg_row0 || !seg_nw) return ww_fail(ctx, WW_EINVAL, "sequence descriptors are NULL");
  WW_ON_DEVICE(ctx, dev);
  if (ww_crnn_segments_capable(m, hop)) return ww_k_crnn_segments_forward(ctx, m, d_mel, mel_rows, seg_row0, seg_nw, n_seg, hop, d_out);
  // every other model / mode: the same windows as an explicit list through the per-window kernels
  const int T = m->info.window, NO = m->info.n_out;
  std::vector<int64_t> rows;
  for (int s = 0; s < n_seg; ++s) {
    if (seg_nw[s] < 0) return ww_fail(ctx, WW_EINVAL, "negative window count in sequence %d", s);
    if (seg_nw[s] && (seg_row0[s] < 0 || seg_row0[s] + (int64_t)(seg_nw[s] - 1) * hop + T > mel_rows))
      return ww_fail(ctx, WW_EINVAL, "sequence %d: windows leave the mel buffer", s);
    for (int k = 0; k < seg_nw[s]; ++k) rows.push_back(seg_row0[s] + (int64_t)k * hop);
  }
  const int64_t nw = (int64_t)rows.size();
  if (nw == 0) return WW_OK;
  if (nw > 0x7fffffff) return ww_fail(ctx, WW_EINVAL, "too many windows in one call");
  std::vector<int32_t> valid((size_t)nw, T);
  const int chunk = nw < WW_MAX_CHUNK ? (int)nw : WW_MAX_CHUNK;
  const size_t b_rows = ww_bump::need((size_t)nw, 8), b_valid = ww_bump::need((size_t)nw, 4), b_ws = model_ws(m, chunk);
  int rc = ww_ensure(ctx, ctx->dev, b_rows + b_valid + b_ws + 1024, false);
  if (rc) return rc;
  ww_bump bump(ctx->dev.ptr, ctx->dev.cap);
  int64_t *d_rows = bump.take<int64_t>((size_t)nw);
  int32_t *d_valid = bump.take<int32_t>((size_t)nw);
  void *ws = bump.take<char>(b_ws);
  WW_HIP(ctx, hipMemcpyAsync(d_rows, rows.data(), (size_t)nw * 8, hipMemcpyHostToDevice, ctx->stream));
  WW_HIP(ctx, hipMemcpyAsync(d_valid, valid.data(), (size_t)nw * 4, hipMemcpyHostToDevice, ctx->stream));
  WW_HIP(ctx, hipStreamSynchronize(ctx->stream));  // the host vectors go out of scope
  for (int64_t w0 = 0; w0 < nw; w0 += chunk) {
    const int n = (int)((nw - w0) < chunk ? (nw - w0) : chunk);
    rc = model_forward(ctx, m, d_mel, mel_rows, d_rows + w0, d_valid + w0, 0, 0, 0, n, ws, d_out + (size_t)w0 * NO, nullptr);
    if (rc) return rc;
  }
  return WW_OK;
  WW_GUARD_END(ctx)
}

int ww_clips_forward_dev(ww_ctx *ctx, const ww_model *m, const int16_t *d_pcm, int32_t n_clips, int32_t samples,
                         const ww_frontend_params *fp, float *d_out) {
  WW_GUARD_BEGIN
  if (!ctx || !m || !d_pcm || !d_out) return ww_fail(ctx, WW_EINVAL, "NULL argument");
  if (n_clips < 0 || samples < 0) return ww_fail(ctx, WW_EINVAL, "negative size");
  if (n_clips == 0) return WW_OK;
  if (n_clips > 65535) return ww_fail(ctx, WW_EINVAL, "at most 65535 clips per call");
  if (((uintptr_t)d_pcm & 15) != 0) return ww_fail(ctx, WW_EINVAL, "d_pcm must be 16-byte aligned");
  int rc = check_fp(ctx, fp, true);
  if (rc) return rc;
  WW_ON_DEVICE(ctx, dev);
  const int64_t nf = ww_num_frames(samples, fp->hop);
  const int F = m->info.n_mel, T = m->info.window;
  // offset tables for this batch geometry: built once, outside any capture
  int64_t *d_so = nullptr, *d_fo = nullptr;
  for (auto &co : ctx->clip_offs)
    if (co.n_clips == n_clips && co.samples == samples && co.hop == fp->hop) {
      d_so = co.d_so;
      d_fo = co.d_fo;
    }
  if (!d_so) {
    // at most eight geometries stay cached (two small tables each); the oldest goes once nothing enqueued can still read it
    if (ctx->clip_offs.size() >= 8) {
      WW_HIP(ctx, hipStreamSynchronize(ctx->stream));
      hipFree(ctx->clip_offs.front().d_so);
      hipFree(ctx->clip_offs.front().d_fo);
      ctx->clip_offs.erase(ctx->clip_offs.begin());
    }
    ww_ctx::clip_offs_t co;
    co.n_clips = n_clips; co.samples = samples; co.hop = fp->hop;
    WW_HIP(ctx, hipMalloc((void **)&co.d_so, sizeof(int64_t) * (n_clips + 1)));
    WW_HIP(ctx, hipMalloc((void **)&co.d_fo, sizeof(int64_t) * (n_clips + 1)));
    hipLaunchKernelGGL(iota_offs_kernel, dim3((n_clips + 256) / 256), dim3(256), 0, ctx->stream, co.d_so, co.d_fo, n_clips,
                       (int64_t)samples, nf);
    WW_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->clip_offs.push_back(co);
    d_so = co.d_so;
    d_fo = co.d_fo;
  }
  // workspace: mel | model scratch
  const size_t b_mel = ww_bump::need((size_t)n_clips * (nf > 0 ? nf : 1) * F, 4);
  const size_t b_ws = model_ws(m, n_clips);
  if ((rc = ww_ensure(ctx, ctx->dev, b_mel + b_ws + 1024, false))) return rc;
  auto enqueue = [&]() -> int {
    ww_bump bump(ctx->dev.ptr, ctx->dev.cap);
    float *d_mel = bump.take<float>((size_t)n_clips * (nf > 0 ? nf : 1) * F);
    void *ws = bump.take<char>(b_ws);
    int r = ww_k_logmel(ctx, m, d_pcm, nullptr, d_so, d_fo, n_clips, (int64_t)n_clips * nf, nf, fp, d_mel, samples, (int64_t)n_clips * samples);
    if (r) return r;
    // one window per clip: rows [c*nf, c*nf + min(nf, T)), zero padded to T
    return model_forward(ctx, m, d_mel, (int64_t)n_clips * nf, nullptr, nullptr, 0, (int)nf, (int)(nf < T ? nf : T),
                         n_clips, ws, d_out, nullptr);
  };
  // Plain stream launches: measured on MI355X (round 1) the chain replays slower as a hipGraph (93 vs 88 us: the queue stays
  // full, so launch latency is hidden, while a graph replay has a 10-16 us floor); the capture path was removed in round 3.
  return enqueue();
  WW_GUARD_END(ctx)
}

// ------------------------------------------------------------------------------------------
// posterior smoothing + sweep
// ------------------------------------------------------------------------------------------
// pos / neg: host pointers (uploaded here) or, with on_device, device pointers the kernels read where they are
static int far_frr_impl(ww_ctx *ctx, bool on_device, const float *pos, int64_t n_pos, const float *neg, int64_t n_neg, int32_t win,
                        const double *thr, int32_t n_thr, double num_wakewords, double hours, double *frr, double *fa_per_h,
                        int64_t *fa_count, double *smoothed_host, double *smoothed_dev) {
  if (!ctx || !thr || !frr || !fa_per_h) return ww_fail(ctx, WW_EINVAL, "NULL argument");
  if (n_pos < 0 || n_neg < 0 || n_thr <= 0) return ww_fail(ctx, WW_EINVAL, "bad sizes");
  if ((n_pos && !pos) || (n_neg && !neg)) return ww_fail(ctx, WW_EINVAL, "NULL posterior buffer");
  if (win > 0 && n_neg > 0 && n_neg < win)
    return ww_fail(ctx, WW_EINVAL, "negative stream shorter than the smoothing window (np.convolve 'same' would change its length)");
  WW_ON_DEVICE(ctx, dev_scope);  // the caller's current device is left as it was
  const size_t b_p = on_device ? 0 : ww_bump::need((size_t)n_pos + 1, 4), b_n = on_device ? 0 : ww_bump::need((size_t)n_neg + 1, 4);
  const size_t b_s = smoothed_dev ? 0 : ww_bump::need((size_t)n_neg + 1, 8), b_t = ww_bump::need((size_t)n_thr, 8);
  int rc = ww_ensure(ctx, ctx->dev, b_p + b_n + b_s + 3 * b_t + 1024, false);
  if (rc) return rc;
  ww_bump bump(ctx->dev.ptr, ctx->dev.cap);
  const float *d_pos = pos, *d_neg = neg;
  if (!on_device) {
    float *up = bump.take<float>(n_pos + 1), *un = bump.take<float>(n_neg + 1);
    if (n_pos) WW_HIP(ctx, hipMemcpyAsync(up, pos, (size_t)n_pos * 4, hipMemcpyHostToDevice, ctx->stream));
    if (n_neg) WW_HIP(ctx, hipMemcpyAsync(un, neg, (size_t)n_neg * 4, hipMemcpyHostToDevice, ctx->stream));
    d_pos = up;
    d_neg = un;
  }
  double *d_sm = smoothed_dev ? smoothed_dev : bump.take<double>(n_neg + 1), *d_thr = bump.take<double>(n_thr);
  unsigned long long *d_pc = bump.take<unsigned long long>(n_thr), *d_fc = bump.take<unsigned long long>(n_thr);
  WW_HIP(ctx, hipMemcpyAsync(d_thr, thr, (size_t)n_thr * 8, hipMemcpyHostToDevice, ctx->stream));
  if ((rc = ww_k_far_frr(ctx, d_pos, n_pos, d_neg, n_neg, win, d_thr, n_thr, d_sm, d_pc, d_fc))) return rc;
  std::vector<unsigned long long> pc(n_thr), fc(n_thr);
  WW_HIP(ctx, hipMemcpyAsync(pc.data(), d_pc, (size_t)n_thr * 8, hipMemcpyDeviceToHost, ctx->stream));
  WW_HIP(ctx, hipMemcpyAsync(fc.data(), d_fc, (size_t)n_thr * 8, hipMemcpyDeviceToHost, ctx->stream));
  if (smoothed_host && n_neg) WW_HIP(ctx, hipMemcpyAsync(smoothed_host, d_sm, (size_t)n_neg * 8, hipMemcpyDeviceToHost, ctx->stream));
  WW_HIP(ctx, hipStreamSynchronize(ctx->stream));
  for (int k = 0; k < n_thr; ++k) {
    frr[k] = (num_wakewords - (double)pc[k]) / num_wakewords;
    fa_per_h[k] = (double)fc[k] / hours;
    if (fa_count) fa_count[k] = (int64_t)fc[k];
  }
  return WW_OK;
}

int ww_far_frr(ww_ctx *ctx, const float *pos, int64_t n_pos, const float *neg, int64_t n_neg, int32_t win,
               const double *thr, int32_t n_thr, double num_wakewords, double hours, double *frr, double *fa_per_h,
               int64_t *fa_count, double *smoothed) {
  WW_GUARD_BEGIN
  return far_frr_impl(ctx, false, pos, n_pos, neg, n_neg, win, thr, n_thr, num_wakewords, hours, frr, fa_per_h, fa_count, smoothed, nullptr);
  WW_GUARD_END(ctx)
}

int ww_far_frr_dev(ww_ctx *ctx, const float *d_pos, int64_t n_pos, const float *d_neg, int64_t n_neg, int32_t win,
                   const double *thr, int32_t n_thr, double num_wakewords, double hours, double *frr, double *fa_per_h,
                   int64_t *fa_count, double *d_smoothed) {
  WW_GUARD_BEGIN
  return far_frr_impl(ctx, true, d_pos, n_pos, d_neg, n_neg, win, thr, n_thr, num_wakewords, hours, frr, fa_per_h, fa_count, nullptr, d_smoothed);
  WW_GUARD_END(ctx)
}

int ww_posterior_pick_dev(ww_ctx *ctx, const float *d_rows, int64_t n, int32_t n_out, int32_t pidx, const int64_t *d_seg_offs,
                          int64_t n_seg, float *d_out) {
  WW_GUARD_BEGIN
  if (!ctx) return WW_EINVAL;
  if (n < 0 || n_seg < 0 || n_out <= 0 || pidx < 0 || pidx >= n_out) return ww_fail(ctx, WW_EINVAL, "posterior pick: bad sizes");
  if ((n > 0 && !d_rows) || ((d_seg_offs ? n_seg : n) > 0 && !d_out)) return ww_fail(ctx, WW_EINVAL, "NULL argument");
  WW_ON_DEVICE(ctx, dev_scope);  // the caller's current device is left as it was
  return ww_k_posterior_pick(ctx, d_rows, n, n_out, pidx, d_seg_offs, n_seg, d_out);
  WW_GUARD_END(ctx)
}

int ww_superframe_smooth(ww_ctx *ctx, const float *in, int64_t n, int32_t T, float stay_bonus, int32_t in_is_cost,
                         uint8_t *path, uint8_t *wake) {
  WW_GUARD_BEGIN
  if (!ctx || (n > 0 && (!in || !wake))) return ww_fail(ctx, WW_EINVAL, "NULL argument");
  if (n < 0) return ww_fail(ctx, WW_EINVAL, "bad sizes");
  if (n == 0) return WW_OK;
  WW_ON_DEVICE(ctx, dev_scope);  // the caller's current device is left as it was
  const size_t b_in = ww_bump::need((size_t)n * T * 2, 4), b_p = ww_bump::need((size_t)n * T, 1), b_w = ww_bump::need((size_t)n, 1);
  int rc = ww_ensure(ctx, ctx->dev, b_in + b_p + b_w + 1024, false);
  if (rc) return rc;
  ww_bump bump(ctx->dev.ptr, ctx->dev.cap);
  float *d_in = bump.take<float>((size_t)n * T * 2);
  unsigned char *d_p = bump.take<unsigned char>((size_t)n * T), *d_w = bump.take<unsigned char>((size_t)n);
  WW_HIP(ctx, hipMemcpyAsync(d_in, in, (size_t)n * T * 2 * 4, hipMemcpyHostToDevice, ctx->stream));
  if ((rc = ww_k_viterbi2(ctx, d_in, n, T, stay_bonus, in_is_cost, d_p, d_w))) return rc;
  if (path) WW_HIP(ctx, hipMemcpyAsync(path, d_p, (size_t)n * T, hipMemcpyDeviceToHost, ctx->stream));
  WW_HIP(ctx, hipMemcpyAsync(wake, d_w, (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
  WW_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return WW_OK;
  WW_GUARD_END(ctx)
}

}  // extern "C"
