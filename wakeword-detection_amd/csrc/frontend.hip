// PCM -> log-mel front end for gfx950.
//
// Replaces the reference's per-sample RingBuffer loop, np.fft.rfft and the filter.tflite
// invoke (utils/tf_lite/filter.py:38-75; spokestack/wakeword/tflite.py:148-191).
//
// Kernel shape: one 256-thread workgroup (4 wavefronts) stages the samples of FPB = 16 consecutive frames of
// one utterance: the 512 + 15*160 samples those frames touch are loaded ONCE with aligned 16-byte loads
// (straight-line: every load of the block is in flight before the first wait), normalised / clipped /
// pre-emphasised in registers and parked in LDS as fp32.  After that barrier (and in fp64 one more behind the
// Hann products) the four wavefronts never meet again; each carries 4 frames to the output, 16 lanes per frame and 16 points per lane: Hann product
// (fp64), a radix-16 DFT in registers, twiddles, one 16x16 transpose through LDS (real parts, then
// imaginary parts, same buffer), a second radix-16 DFT - the 256-point complex FFT of the even/odd-packed
// frame - and the real-FFT untangling, for which lane j fetches its partner Z[256-k] (lane (16-j)%16) through
// the dead transpose buffer; two magnitudes per evaluation land in LDS.  The mel filter runs on the vector ALU
// with the lanes re-dealt as (band slot, frame): 64 fused multiply-adds per lane on magnitudes read 16 bytes
// at a time (bands dealt so that these reads are bank-conflict-free: load_filter in api.hip), then the
// log/affine tail, and the wave's 4x40 tile leaves through LDS as one contiguous store.
// What bounds it (rocprofv3 --pmc, 256 clips): per CU the vector ALUs are busy ~60 % of the kernel and the
// LDS ~38 %; the rest is dependency latency inside a wave's transform, which is why occupancy pays: 4 workgroups
// per CU need <= 128 VGPRs and <= 40 KB of LDS each.  In fp64 that is met by giving wave 3 the sample tile as its
// transpose buffer (the tile is dead once every wave has formed its Hann products: one extra barrier) and by
// keeping no twiddle table in LDS: the fp64 twiddles (W256^(j k1), W512^k: two 4 KB tables every wave of the chip
// shares) are read through the vector L1 - 23 sixteen-byte loads per lane for the 88 fp64 instructions that rebuilt
// them from two per-lane constants (round 2: VALU instructions per wave 935 -> 855, kernel -1.5 %: the kernel is
// bound by the latency chains of its LDS round trips and by its 2.5 launch rounds, not by issue slots; -DWW_TW_GLOBAL=0
// restores the ALU form).  (Halving the transposes' footprint by running them two frames at a time costs more LDS
// instructions than the occupancy returns: DESIGN.md 7.1.)
// (stft_mag_kernel and the streaming kernel keep the earlier one-wave-per-frame radix-4 Stockham
// FFT of fft_device.h: they are not on the batched path.)
//
// REAL = double reproduces the reference numerics (Hann product and FFT in float64,
// spokestack/wakeword/tflite.py:175-176, result cast to float32); REAL = float is the fast mode.
#include "common.h"

#include <vector>

#include "fft_device.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#ifndef FPB
#define FPB 16  // frames per block
#endif
#define WAVES 4
static_assert(FPB == WAVES * 4, "each wave transforms exactly one group of 4 frames (mags overlay the transposes)");

struct logmel_args {
  const int16_t *pcm;
  const float *f32;
  const int64_t *sample_offs;
  const int64_t *frame_offs;
  int n_utt;
  int tiles_per_utt;  // ceil(longest utterance's frames / FPB)
  int hop;
  float divisor;
  float rdiv;       // RN(1 / divisor)
  int fast_div;     // divisor is 32767 or 32768: 3-op exact quotient (see pcm_quot)
  int clip;
  float preemph;
  // filter
  const int *start;
  const float *wpad;  // [WW_MEL_TAPS][64] tap-major, zero padded
  const float *bias;
  int n_mel;
  float floor_v, log_off, scale;
  const double *hann, *tw256, *tw512, *tw16;
  const float *melV;           // mel filter in lane form: [WW_MELV_CHUNKS][16 slots] float4 (api.hip, load_filter)
  const int *melVmeta;         // [3 groups][16 slots]: first bin | band << 16
  int melv_aligned;            // first bins are multiples of 4: 16-byte magnitude reads
  float *mel;
  // stft-only mode
  const float *frames;
  float *mag_out;
  int64_t n_frames_direct;
  // logmel_rows_kernel: mel rows are numbered through the whole launch
  int64_t total_frames;
  int uniform_nf;      // > 0: every clip has this many frames and uniform_ns samples, clip u starts at sample u * uniform_ns
  int64_t uniform_ns;
  unsigned uni_magic;  // uniform_nf >= 4: row / uniform_nf = __umulhi(row, uni_magic) >> uni_shift for every row < 2^31
  int uni_shift;
  long long *stamps;  // development (-DWW_FE_STAMPS=1): [workgroups][4 waves][12] s_memtime at the phase boundaries
};

#ifndef WW_FE_STAMPS
#define WW_FE_STAMPS 0
#endif
#if WW_FE_STAMPS
#define FE_STAMP(i_)                                                                                         \
  {                                                                                                          \
    __builtin_amdgcn_sched_barrier(0);                                                                       \
    if (a.stamps && lane == 0) a.stamps[((size_t)blockIdx.x * 4 + wave) * 12 + (i_)] = __builtin_amdgcn_s_memtime(); \
    __builtin_amdgcn_sched_barrier(0);                                                                       \
  }
#else
#define FE_STAMP(i_)
#endif

// int16 / divisor, correctly rounded (reference: frame.astype(np.float32) / 32767, tflite.py:150).
// For the two divisors in use (32767, 32768) the quotient of ANY int16 is obtained exactly by
// q0 = a*r, e = fma(-b, q0, a), q = fma(e, r, q0) with r = RN(1/b) (Markstein); verified
// exhaustively over all 65536 inputs with exact rational arithmetic (tests/test_host_logic.py).
__device__ __forceinline__ float pcm_quot(float x, const logmel_args &a) {
  if (a.fast_div) {
    const float q0 = __fmul_rn(x, a.rdiv);
    const float e = __fmaf_rn(-a.divisor, q0, x);
    return __fmaf_rn(e, a.rdiv, q0);
  }
  return __fdiv_rn(x, a.divisor);
}

template <bool F32IN>
__device__ __forceinline__ float norm_sample(const logmel_args &a, int64_t g) {
  if (F32IN) return a.f32[g];
  float v = pcm_quot((float)a.pcm[g], a);
  if (a.clip) v = fminf(fmaxf(v, -1.0f), 1.0f);
  return v;
}

// ---- radix-16 DFT in registers ------------------------------------------------------------
template <typename R>
__device__ __forceinline__ void radix4(cplx<R> &a0, cplx<R> &a1, cplx<R> &a2, cplx<R> &a3) {
  const cplx<R> t0 = {a0.re + a2.re, a0.im + a2.im}, t1 = {a0.re - a2.re, a0.im - a2.im};
  const cplx<R> t2 = {a1.re + a3.re, a1.im + a3.im}, t3 = {a1.im - a3.im, -(a1.re - a3.re)};  // (a1-a3)*(-i)
  a0 = {t0.re + t2.re, t0.im + t2.im};
  a1 = {t1.re + t3.re, t1.im + t3.im};
  a2 = {t0.re - t2.re, t0.im - t2.im};
  a3 = {t1.re - t3.re, t1.im - t3.im};
}

template <typename R>
__device__ __forceinline__ cplx<R> mulc(cplx<R> a, R wr, R wi) {
  return {a.re * wr - a.im * wi, a.re * wi + a.im * wr};
}

// Forward 16-point DFT, in place.  Input a[n]; output A[k] is left at position 4*(k%4) + k/4.
template <typename R>
__device__ __forceinline__ void dft16(cplx<R> (&a)[16]) {
  constexpr R C = (R)0.92387953251128675613, S = (R)0.38268343236508977173, H = (R)0.70710678118654752440;
#pragma unroll
  for (int q = 0; q < 4; ++q) radix4(a[q], a[q + 4], a[q + 8], a[q + 12]);
  // a[q + 4p] *= W16^(q p),  W16^m = (cos(2 pi m/16), -sin(2 pi m/16))
  a[5] = mulc(a[5], C, -S);   a[9] = mulc(a[9], H, -H);                        a[13] = mulc(a[13], S, -C);
  a[6] = mulc(a[6], H, -H);   a[10] = cplx<R>{a[10].im, -a[10].re};            a[14] = mulc(a[14], -H, -H);
  a[7] = mulc(a[7], S, -C);   a[11] = mulc(a[11], -H, -H);                     a[15] = mulc(a[15], -C, S);
#pragma unroll
  for (int p = 0; p < 4; ++p) radix4(a[4 * p], a[4 * p + 1], a[4 * p + 2], a[4 * p + 3]);
}

__host__ __device__ constexpr int k_of(int pos) { return (pos >> 2) + 4 * (pos & 3); }
__host__ __device__ constexpr int pos_of(int k) { return 4 * (k & 3) + (k >> 2); }

template <typename R> __device__ __forceinline__ R shfl_r(R v, int src);
template <> __device__ __forceinline__ float shfl_r<float>(float v, int src) { return __shfl(v, src); }
template <> __device__ __forceinline__ double shfl_r<double>(double v, int src) { return __shfl(v, src); }

// W32^m = exp(-2 pi i m / 32), m = 0..15 (compile-time constants of the untangling twiddles)
__device__ constexpr double W32_RE[16] = {1.00000000000000000000e+00, 9.80785280403230430579e-01, 9.23879532511286738483e-01, 8.31469612302545235671e-01, 7.07106781186547572737e-01, 5.55570233019602288671e-01, 3.82683432365089837290e-01, 1.95090322016128331351e-01, 6.12323399573676603587e-17, -1.95090322016128192573e-01, -3.82683432365089726268e-01, -5.55570233019601955604e-01, -7.07106781186547461715e-01, -8.31469612302545346694e-01, -9.23879532511286738483e-01, -9.80785280403230430579e-01};
__device__ constexpr double W32_IM[16] = {-0.00000000000000000000e+00, -1.95090322016128248084e-01, -3.82683432365089781779e-01, -5.55570233019602177649e-01, -7.07106781186547461715e-01, -8.31469612302545235671e-01, -9.23879532511286738483e-01, -9.80785280403230430579e-01, -1.00000000000000000000e+00, -9.80785280403230430579e-01, -9.23879532511286738483e-01, -8.31469612302545457716e-01, -7.07106781186547572737e-01, -5.55570233019602177649e-01, -3.82683432365089892802e-01, -1.95090322016128608906e-01};

// One wave's LDS instructions are executed in issue order, so a write followed by another
// lane's read (or a read followed by an overwrite) needs no s_waitcnt - only a scheduling fence.
__device__ __forceinline__ void lds_fence() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_wave_barrier();
}

// Sixteen consecutive 8-byte LDS reads as sixteen ds_read_b64 (2 LDS cycles each).  Left to itself the
// compiler pairs them into ds_read2_b64, which the LDS services at 8 cycles per instruction - twice the
// time for the same bytes (MI355X_MICROARCH.md, LDS table).  The reads are issued back to back; the
// caller must lds_wait_all() the destinations before using them.
__device__ __forceinline__ void lds_read16_b64(const double *p, double (&d)[16]) {
  const unsigned a = (unsigned)(uintptr_t)p;  // low 32 bits of a flat LDS pointer = the LDS byte address
#define WW_RD(i) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(d[i]) : "v"(a), "n"((i) * 8))
  WW_RD(0); WW_RD(1); WW_RD(2); WW_RD(3); WW_RD(4); WW_RD(5); WW_RD(6); WW_RD(7);
  WW_RD(8); WW_RD(9); WW_RD(10); WW_RD(11); WW_RD(12); WW_RD(13); WW_RD(14); WW_RD(15);
#undef WW_RD
}
// The same with a stride of 128 bytes: lane j's sample pairs (x[32 n1 + 2j], x[32 n1 + 2j + 1]), n1 = 0..15.
__device__ __forceinline__ void lds_read16_b64_s128(const float *p, double (&d)[16]) {
  const unsigned a = (unsigned)(uintptr_t)p;
#define WW_RD(i) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(d[i]) : "v"(a), "n"((i) * 128))
  WW_RD(0); WW_RD(1); WW_RD(2); WW_RD(3); WW_RD(4); WW_RD(5); WW_RD(6); WW_RD(7);
  WW_RD(8); WW_RD(9); WW_RD(10); WW_RD(11); WW_RD(12); WW_RD(13); WW_RD(14); WW_RD(15);
#undef WW_RD
}
__device__ __forceinline__ void lds_wait_all(double (&d)[16]) {
  asm volatile("s_waitcnt lgkmcnt(0)"
               : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]), "+v"(d[6]), "+v"(d[7]), "+v"(d[8]),
                 "+v"(d[9]), "+v"(d[10]), "+v"(d[11]), "+v"(d[12]), "+v"(d[13]), "+v"(d[14]), "+v"(d[15])
               :
               : "memory");
}

// An LDS pointer the compiler knows nothing about: constant element offsets from it then travel in the
// instruction's offset field instead of costing one vector add per access.
typedef __attribute__((address_space(3))) const float lds_cfloat;
typedef __attribute__((address_space(3))) const f32x4 lds_cfloat4;
__device__ __forceinline__ lds_cfloat *lds_opaque(const float *p) {
  unsigned a = (unsigned)(uintptr_t)p;  // low 32 bits of a flat LDS pointer = the LDS byte address
  asm volatile("" : "+v"(a));
  return (lds_cfloat *)(uintptr_t)a;
}

// (h[2n], h[2n+1]) for n = 16 n1 + j from the half table of 128 pairs (h[m] = h[511 - m])
template <typename H2>
__device__ __forceinline__ H2 hann_pair(const H2 *tb, int n1, int j) {
  if (n1 < 8) return tb[16 * n1 + j];
  const H2 m = tb[16 * (15 - n1) + 15 - j];
  H2 r;
  r.x = m.y;
  r.y = m.x;
  return r;
}
template <typename R> struct hann_t { typedef double2 type; };
template <> struct hann_t<float> { typedef float2 type; };  // fast mode: the window product in fp32 as well

#define MAG_LD 272  // floats per frame of magnitudes: 257 + zero pad to 17*16; 16 mod 32, so the two frames a
                    // 32-lane write group touches use disjoint banks
#define TR_LD 17    // padded row of the 16x16 transpose

template <typename R>
__host__ __device__ constexpr int wbuf_bytes() {
  // per-wave scratch: 16x16 transposes of 4 frames, later the 4 frames' magnitudes.
  // Size in floats must be 16 mod 64 so that the 16 frames of a block start on distinct banks.
  return sizeof(R) == 8 ? (4 * 16 * TR_LD * 8 + 64) : (4 * MAG_LD * 4);
}

template <typename R, bool F32IN, bool SIMPLE>
__global__ __launch_bounds__(256, sizeof(R) == 8 ? 4 : 5) void logmel_kernel(logmel_args a) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 15, sub = lane >> 4;
  // XCD-aware order: workgroup ids go round-robin to the 8 XCDs (id % 8), each with its own L2.  Adjacent
  // tiles of an utterance share 352 of their 2 912 samples, so all tiles of utterance u are given ids of
  // residue u % 8: the shared lines are then fetched once per utterance instead of once per tile.
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int u = (slot / a.tiles_per_utt) * 8 + xcd, tile_idx = slot % a.tiles_per_utt;
  if (u >= a.n_utt) return;
  const int64_t s_begin = a.sample_offs[u], s_end = a.sample_offs[u + 1];
  const int64_t n_samples = s_end - s_begin;
  const int64_t nf = n_samples >= WIN ? (n_samples - WIN) / a.hop + 1 : 0;
  const int64_t f0 = (int64_t)tile_idx * FPB;
  if (f0 >= nf) return;
  const int nfb = (int)((nf - f0) < FPB ? (nf - f0) : FPB);
  FE_STAMP(0)

  // ---- LDS carve-up
  size_t off = 0;
  // Hann table: np.hanning is symmetric (h[n] = h[511 - n]), so the first 256 values serve as 128 pairs
  typedef typename hann_t<R>::type H2;
  H2 *tb_hann = (H2 *)(smem + off); off += 128 * sizeof(double2);
  // fp32: twiddle tables in LDS.  fp64: LDS is the occupancy limiter (4 workgroups per CU need <= 40 KB each),
  // so the twiddles are rebuilt from two per-lane constants instead (W256^j and W512^j) - see below - and only
  // three of the four per-wave buffers are allocated: wave 3 uses the sample tile, which is dead once every wave
  // has formed its Hann products (one extra barrier).
  constexpr bool TW_LDS = sizeof(R) == 4;
#ifndef WW_TW_GLOBAL
#define WW_TW_GLOBAL 1
#endif
  constexpr bool TW_GLOBAL = WW_TW_GLOBAL != 0;  // fp64: twiddles from the L1-resident tables instead of the vector ALU
  cplx<R> *tb_tw = (cplx<R> *)(smem + off); off += TW_LDS ? 256 * sizeof(cplx<R>) : 0;   // [k1][j] = W256^(j k1)
  cplx<R> *tb_un = (cplx<R> *)(smem + off); off += TW_LDS ? 256 * sizeof(cplx<R>) : 0;   // W512^k
  const cplx<R> base_tw = {(R)a.tw16[2 * (16 + j)], (R)a.tw16[2 * (16 + j) + 1]};          // W256^j
  const cplx<R> base_un = {(R)a.tw512[2 * j], (R)a.tw512[2 * j + 1]};                      // W512^j
  constexpr int NWB = WAVES - 1;
  unsigned char *wbuf = smem + off; off += NWB * wbuf_bytes<R>();
  float *tile = (float *)(smem + off);  // fp32 samples, [WIN + (FPB-1)*hop + 16]

  // ---- stage the sample tile: aligned 16-byte global loads; tile[i + shift] = x[g_first + i]
  constexpr int VEC = F32IN ? 4 : 8;                      // elements per 16-byte load
  const int64_t g_first = s_begin + f0 * a.hop;           // first sample of frame f0
  const int n_need = WIN + (nfb - 1) * a.hop;             // samples used by this block
  const int shift = (int)(g_first % VEC);
  auto fill_tables = [&](const double2 hv, const double2 twv) {
    if (tid < 128) {
      H2 hq;
      hq.x = hv.x;
      hq.y = hv.y;
      tb_hann[tid] = hq;
    }
    if (TW_LDS) {
      tb_tw[tid] = {(R)twv.x, (R)twv.y};
      tb_un[tid] = {(R)a.tw512[2 * tid], (R)a.tw512[2 * tid + 1]};
    }
  };
  if (SIMPLE) {
    // No pre-emphasis, divisor 32767/32768, at most two 16-byte vectors per thread (host checks):
    // straight-line code - both sample loads and the table load are in flight together, so the
    // block pays one memory latency before its first barrier instead of four in a row.
    const int64_t ga = g_first - shift;                   // multiple of VEC, >= 0
    const int n_vec = (shift + n_need + VEC - 1) / VEC;
    const int64_t total = a.sample_offs[a.n_utt];
    const int64_t last = (total - VEC) & ~(int64_t)(VEC - 1);  // last full aligned vector (total >= WIN here)
    const int64_t tile_last = ga + (int64_t)(n_vec - 1) * VEC;
    int64_t gq[2];
    uint4 raw[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      gq[h] = ga + (int64_t)(tid + 256 * h) * VEC;
      // clamped: always a full vector inside the buffer, and never past this block's last vector (threads
      // beyond it would otherwise pull the NEXT tile's lines through this XCD's L2: +40 % fabric traffic)
      int64_t gl = gq[h] < tile_last ? gq[h] : tile_last;
      gl = gl < last ? gl : last;
      raw[h] = F32IN ? *(const uint4 *)(a.f32 + gl) : *(const uint4 *)(a.pcm + gl);
    }
    const double2 hv = *(const double2 *)(a.hann + 2 * (tid & 127));
    const double2 twv = *(const double2 *)(a.tw16 + 2 * tid);
    const float lim = a.clip ? 1.0f : __builtin_inff();
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int q = tid + 256 * h;
      if (q < n_vec) {
        float o[VEC];
        if (gq[h] <= last) {
          const unsigned int w32[4] = {raw[h].x, raw[h].y, raw[h].z, raw[h].w};
          if (F32IN) {
#pragma unroll
            for (int e = 0; e < VEC; ++e) o[e] = __uint_as_float(w32[e]);
          } else {
            // the exact quotient (see pcm_quot) on sample pairs: v_pk_mul_f32 / v_pk_fma_f32, two samples per issue
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            const f32x2 r2 = {a.rdiv, a.rdiv}, nb2 = {-a.divisor, -a.divisor};
#pragma unroll
            for (int e = 0; e < VEC; e += 2) {
              const f32x2 x = {(float)(int)(short)(w32[e >> 1] & 0xffffu), (float)((int)w32[e >> 1] >> 16)};
              const f32x2 q0 = x * r2;
              const f32x2 er = __builtin_elementwise_fma(nb2, q0, x);
              const f32x2 q = __builtin_elementwise_fma(er, r2, q0);
              o[e] = __builtin_amdgcn_fmed3f(q.x, -lim, lim);
              o[e + 1] = __builtin_amdgcn_fmed3f(q.y, -lim, lim);
            }
          }
        } else {  // ragged end of the whole buffer: element-wise, zero beyond it
#pragma unroll
          for (int e = 0; e < VEC; ++e) o[e] = (gq[h] + e < total) ? norm_sample<F32IN>(a, gq[h] + e) : 0.0f;
        }
        float4 *dst = (float4 *)(tile + (size_t)q * VEC);
        dst[0] = make_float4(o[0], o[1], o[2], o[3]);
        if (VEC == 8) dst[1] = make_float4(o[4], o[5], o[6], o[7]);
      }
    }
    fill_tables(hv, twv);
  } else {
    fill_tables(*(const double2 *)(a.hann + 2 * (tid & 127)), *(const double2 *)(a.tw16 + 2 * tid));
    const int64_t ga = g_first - shift;                   // multiple of VEC, >= 0
    const int n_vec = (shift + n_need + VEC - 1) / VEC;
    const int64_t total = a.sample_offs[a.n_utt];
    const float alpha = a.preemph;
    for (int q = tid; q < n_vec; q += 256) {
      const int64_t g = ga + (int64_t)q * VEC;
      float v[VEC + 1];
      // v[0] = sample g-1 (pre-emphasis carry; 0 at the start of the utterance)
      v[0] = (alpha != 0.0f && g - 1 >= s_begin) ? norm_sample<F32IN>(a, g - 1) : 0.0f;
      if (g + VEC <= total) {
        if (F32IN) {
          const float4 raw = *(const float4 *)(a.f32 + g);
          v[1] = raw.x; v[2] = raw.y; v[3] = raw.z; v[4] = raw.w;
        } else {
          const uint4 raw = *(const uint4 *)(a.pcm + g);
          const unsigned int w32[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const int16_t s16 = (int16_t)((w32[e >> 1] >> ((e & 1) * 16)) & 0xffffu);
            float f = pcm_quot((float)s16, a);
            if (a.clip) f = fminf(fmaxf(f, -1.0f), 1.0f);
            v[1 + e] = f;
          }
        }
      } else {
#pragma unroll
        for (int e = 0; e < VEC; ++e) v[1 + e] = (g + e < total) ? norm_sample<F32IN>(a, g + e) : 0.0f;
      }
      float o[VEC];
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        // reference: frame -= pre_emphasis * previous  (separate fp32 multiply and subtract).
        // Slots in front of the utterance start are never read by a frame of this utterance,
        // except that sample s_begin itself must see a zero carry (v[0] above / guard here).
        const bool at_start = (g + e == s_begin);
        const float prev = at_start ? 0.0f : v[e];
        o[e] = (alpha != 0.0f) ? __fsub_rn(v[1 + e], __fmul_rn(alpha, prev)) : v[1 + e];
      }
      float4 *dst = (float4 *)(tile + (size_t)q * VEC);
      dst[0] = make_float4(o[0], o[1], o[2], o[3]);
      if (VEC == 8) dst[1] = make_float4(o[4], o[5], o[6], o[7]);
    }
  }
  FE_STAMP(1)
  __syncthreads();
  FE_STAMP(2)

  // ---- FFT: every 16-lane row of a wave owns one frame (4 frames per wave at a time)
  unsigned char *wb = wave == WAVES - 1 ? (unsigned char *)tile : wbuf + (size_t)wave * wbuf_bytes<R>();
  R *tr = (R *)wb;          // [4][16][TR_LD]
  float *mg = (float *)wb;  // overlay: [4][MAG_LD]
  const int fb = wave * 4;
  const bool active = fb < nfb;
  cplx<R> v[16];
  if (active) {
    int f = fb + sub;
    f = f < nfb ? f : nfb - 1;  // surplus rows recompute the last frame (results unused)
    const float *src = tile + shift + f * a.hop;
    // pass 1: lane j holds z[16 n1 + j], n1 = 0..15; Hann product in fp64 (tflite.py:175; fp32 in the fast mode)
    if (((shift | a.hop) & 1) == 0) {  // block-uniform
      // 8-byte aligned pairs: ds_read_b64 (a quarter of the LDS time of the two-dword form, and with hop = 160
      // the four frames of a wave sit 32 banks apart: conflict-free)
      double xs[16];
      lds_read16_b64_s128(src + 2 * j, xs);
      H2 h[16];
#pragma unroll
      for (int n1 = 0; n1 < 16; ++n1) h[n1] = hann_pair(tb_hann, n1, j);
      lds_wait_all(xs);
#pragma unroll
      for (int n1 = 0; n1 < 16; ++n1) {
        v[n1].re = (R)((R)__int_as_float(__double2loint(xs[n1])) * h[n1].x);
        v[n1].im = (R)((R)__int_as_float(__double2hiint(xs[n1])) * h[n1].y);
      }
    } else {
#pragma unroll
      for (int n1 = 0; n1 < 16; ++n1) {
        const int n = 16 * n1 + j;
        const H2 h = hann_pair(tb_hann, n1, j);
        v[n1].re = (R)((R)src[2 * n] * h.x);
        v[n1].im = (R)((R)src[2 * n + 1] * h.y);
      }
    }
  }
  FE_STAMP(3)
  __syncthreads();  // the sample tile is dead now - wave 3's transposes and magnitudes move in
  FE_STAMP(4)
  // From here on the four waves never meet again: each one carries its own 4 frames to the output.
  if (!active) return;
  {
    dft16<R>(v);
    FE_STAMP(5)
    if (TW_LDS) {
#pragma unroll
      for (int pos = 1; pos < 16; ++pos) v[pos] = cmul(v[pos], tb_tw[k_of(pos) * 16 + j]);
    } else if (TW_GLOBAL) {
      // fp64: the 4 KB table [k1][j] stays in the vector L1 (every wave of the chip reads the same 4 KB): fifteen 16-byte
      // loads per lane in place of the 84 fp64 instructions of the product tree below
#pragma unroll
      for (int pos = 1; pos < 16; ++pos) {
        const double2 t = *(const double2 *)(a.tw16 + 2 * (k_of(pos) * 16 + j));
        v[pos] = cmul(v[pos], cplx<R>{(R)t.x, (R)t.y});
      }
    } else {
      // W256^(j k1) for k1 = 1..15 as powers of W256^j (product tree, depth <= 4: a few fp64 ulp)
      cplx<R> p[16];
      p[1] = base_tw;
      p[2] = cmul(p[1], p[1]);   p[3] = cmul(p[2], p[1]);   p[4] = cmul(p[2], p[2]);   p[5] = cmul(p[4], p[1]);
      p[6] = cmul(p[3], p[3]);   p[7] = cmul(p[4], p[3]);   p[8] = cmul(p[4], p[4]);   p[9] = cmul(p[8], p[1]);
      p[10] = cmul(p[5], p[5]);  p[11] = cmul(p[8], p[3]);  p[12] = cmul(p[6], p[6]);  p[13] = cmul(p[8], p[5]);
      p[14] = cmul(p[7], p[7]);  p[15] = cmul(p[8], p[7]);
#pragma unroll
      for (int pos = 1; pos < 16; ++pos) v[pos] = cmul(v[pos], p[k_of(pos)]);
    }
    FE_STAMP(6)
    // 16x16 transpose through LDS, real parts then imaginary parts (same buffer)
    cplx<R> w[16];
    R *trs = tr + sub * 16 * TR_LD;
#pragma unroll
    for (int pos = 0; pos < 16; ++pos) trs[k_of(pos) * TR_LD + j] = v[pos].re;
    lds_fence();
    if (sizeof(R) == 8) {
      double wre[16], wim[16];
      lds_read16_b64((const double *)trs + j * TR_LD, wre);
      lds_fence();
#pragma unroll
      for (int pos = 0; pos < 16; ++pos) trs[k_of(pos) * TR_LD + j] = v[pos].im;
      lds_fence();
      lds_read16_b64((const double *)trs + j * TR_LD, wim);
      lds_wait_all(wre);
      lds_wait_all(wim);
#pragma unroll
      for (int n2 = 0; n2 < 16; ++n2) {
        w[n2].re = (R)wre[n2];
        w[n2].im = (R)wim[n2];
      }
    } else {
#pragma unroll
      for (int n2 = 0; n2 < 16; ++n2) w[n2].re = trs[j * TR_LD + n2];
      lds_fence();
#pragma unroll
      for (int pos = 0; pos < 16; ++pos) trs[k_of(pos) * TR_LD + j] = v[pos].im;
      lds_fence();
#pragma unroll
      for (int n2 = 0; n2 < 16; ++n2) w[n2].im = trs[j * TR_LD + n2];
      lds_fence();
    }
    FE_STAMP(7)
    // pass 2: lane j = k1 holds Y[n2][k1]; output w[pos] = Z[k1 + 16 k_of(pos)]
    dft16<R>(w);
    FE_STAMP(8)
    // untangle: with a = Z[k], b = conj Z[256-k]:  2E = a+b, 2O = (a-b)/i, 2T = W512^k 2O and
    //   2X[k] = 2E + 2T,   2X[256-k] = conj(2E - 2T)   ->  two magnitudes per evaluation, k < 128 only;
    //   the factor 2 leaves as an exact 0.5 after the fp32 square root.
    // Lane j register k2 holds k = j + 16 k2; its partner Z[256-k] lives in lane (16-j)%16 at
    // k2' = 15-k2 (j > 0) or in the same lane at k2' = (16-k2)%16 (j = 0).  The partners travel
    // through the (dead) transpose buffer: rows 8..15 <- registers k2' = 8..15, row 7 <- k2' = 0
    // (only lane 0 reads that one), so lane j reads row 15-k2 (+1 for j = 0; row 7 for k2 = 0).
    const int pj = (16 - j) & 15;
    const R *prow0 = trs + (j == 0 ? 7 : 15) * TR_LD + pj;
    const R *prow = trs + (j == 0 ? 16 : 15) * TR_LD + pj;
    cplx<R> pz[8];
    trs[7 * TR_LD + j] = w[pos_of(0)].re;
#pragma unroll
    for (int r = 8; r < 16; ++r) trs[r * TR_LD + j] = w[pos_of(r)].re;
    lds_fence();
    pz[0].re = prow0[0];
#pragma unroll
    for (int k2 = 1; k2 < 8; ++k2) pz[k2].re = prow[-k2 * TR_LD];
    lds_fence();
    trs[7 * TR_LD + j] = w[pos_of(0)].im;
#pragma unroll
    for (int r = 8; r < 16; ++r) trs[r * TR_LD + j] = w[pos_of(r)].im;
    lds_fence();
    pz[0].im = prow0[0];
#pragma unroll
    for (int k2 = 1; k2 < 8; ++k2) pz[k2].im = prow[-k2 * TR_LD];
    lds_fence();
    FE_STAMP(9)
    float *mrow = mg + sub * MAG_LD;
#pragma unroll
    for (int k2 = 0; k2 < 8; ++k2) {
      const cplx<R> own = w[pos_of(k2)];
      // W512^(j + 16 k2) = W512^j * W32^k2
      cplx<R> un;
      if (TW_LDS) {
        un = tb_un[j + 16 * k2];
      } else if (TW_GLOBAL) {
        const double2 t = *(const double2 *)(a.tw512 + 2 * (j + 16 * k2));
        un = cplx<R>{(R)t.x, (R)t.y};
      } else {
        un = cmul(base_un, cplx<R>{(R)W32_RE[k2], (R)W32_IM[k2]});
      }
      const R er = own.re + pz[k2].re, ei = own.im - pz[k2].im;
      const R orr = own.im + pz[k2].im, oi = pz[k2].re - own.re;
      const R tr_ = orr * un.re - oi * un.im, ti_ = orr * un.im + oi * un.re;
      const R pr = er + tr_, pi = ei + ti_, qr = er - tr_, qi = ei - ti_;
      const int k = j + 16 * k2;
      mrow[k] = __builtin_amdgcn_sqrtf((float)(pr * pr + pi * pi));  // 2|X[k]|: the mel weights carry the 0.5
      mrow[256 - k] = __builtin_amdgcn_sqrtf((float)(qr * qr + qi * qi));  // k = 0 -> bin 256
    }
    {
      // k = 128 pairs with itself: X[128] = conj(Z[128]) (lane 0, k2 = 8); bins 257..271 are zero padding
      const cplx<R> z = w[pos_of(8)];
      if (j == 0) mrow[128] = 2.0f * __builtin_amdgcn_sqrtf((float)(z.re * z.re + z.im * z.im));
      else mrow[256 + j] = 0.0f;
    }
  }

  // ---- mel filterbank on the vector ALU, per wave: lane 4 s + q owns frame q and slot s of each of the three
  //   band groups (load_filter, api.hip): 36 + 16 + 12 padded taps, one fused multiply-add per tap with the
  //   magnitudes read 16 bytes at a time from this wave's LDS rows.  Frame in the low lane bits: the four
  //   16-lane groups a ds_read_b128 is served in then hold four slots x four frames each, the rows of the four
  //   frames sit 4 sixteen-byte bank slots apart, and load_filter deals the bands so that the four slots of such
  //   a group start on different slots mod 4 - conflict-free.  (The fp32 MFMA form of this contraction kept
  //   the SIMD's vector ALU idle for 32 cycles per instruction - fp32 MFMA and VALU share a datapath on gfx950
  //   - and needed three workgroup barriers for the partial sums; this form needs none.)
  FE_STAMP(10)
  lds_fence();
  {
    const int j = lane >> 2, sub = lane & 3;  // mel phase only: (slot, frame) of this lane
    constexpr int CAPQ[3] = {9, 4, 3}, C0[3] = {0, 9, 13};
    const float4 *wv = (const float4 *)a.melV + j;
    const float *mrow = mg + sub * MAG_LD;
    __builtin_amdgcn_sched_barrier(0);  // the 64 weight registers must not be live across the FFT
    int meta[3];
#pragma unroll
    for (int g = 0; g < 3; ++g) meta[g] = a.melVmeta[g * 16 + j];
    // fp64 (128-register budget): all 16 weight chunks in flight at once; fp32 (96 registers at 5 workgroups
    // per CU): each group's chunks are fetched when its turn comes
    constexpr bool WQ_LATE = sizeof(R) == 4;
    float4 wq[WW_MELV_CHUNKS];
    if (!WQ_LATE) {
#pragma unroll
      for (int c = 0; c < WW_MELV_CHUNKS; ++c) wq[c] = wv[c * 16];
    }
    int band[3];
    float bias[3];
#pragma unroll
    for (int g = 0; g < 3; ++g) {
      band[g] = (int)((unsigned)meta[g] >> 16);  // 0xffff: empty slot
      bias[g] = a.bias[band[g] < a.n_mel ? band[g] : 0];
    }
    float res[3];
    if (a.melv_aligned) {
#pragma unroll
      for (int g = 0; g < 3; ++g) {
        __builtin_amdgcn_sched_barrier(0);  // one group's magnitudes in flight at a time
        if (WQ_LATE) {
#pragma unroll
          for (int c = 0; c < CAPQ[g]; ++c) wq[C0[g] + c] = wv[(C0[g] + c) * 16];
        }
        lds_cfloat4 *mb = (lds_cfloat4 *)lds_opaque(mrow + (meta[g] & 0xffff));
        float acc = 0.f, acc1 = 0.f;  // two chains: a dependent fp32 FMA does not issue back to back
#pragma unroll
        for (int c = 0; c < CAPQ[g]; ++c) {
          const float4 w4 = wq[C0[g] + c];
          const f32x4 m4 = mb[c];
          acc = fmaf(m4[0], w4.x, acc);
          acc1 = fmaf(m4[1], w4.y, acc1);
          acc = fmaf(m4[2], w4.z, acc);
          acc1 = fmaf(m4[3], w4.w, acc1);
        }
        acc += acc1;
        res[g] = (logf(fmaxf(acc + bias[g], a.floor_v)) + a.log_off) * a.scale;
      }
    } else {
#pragma unroll
      for (int g = 0; g < 3; ++g) {
        __builtin_amdgcn_sched_barrier(0);
        if (WQ_LATE) {
#pragma unroll
          for (int c = 0; c < CAPQ[g]; ++c) wq[C0[g] + c] = wv[(C0[g] + c) * 16];
        }
        lds_cfloat *mb = lds_opaque(mrow + (meta[g] & 0xffff));
        float acc = 0.f;
#pragma unroll
        for (int c = 0; c < CAPQ[g]; ++c) {
          const float4 w4 = wq[C0[g] + c];
          acc = fmaf(mb[4 * c + 0], w4.x, acc);
          acc = fmaf(mb[4 * c + 1], w4.y, acc);
          acc = fmaf(mb[4 * c + 2], w4.z, acc);
          acc = fmaf(mb[4 * c + 3], w4.w, acc);
        }
        res[g] = (logf(fmaxf(acc + bias[g], a.floor_v)) + a.log_off) * a.scale;
      }
    }
    // park the wave's 4 x n_mel tile in LDS (its magnitudes are dead) for one contiguous store; empty slots
    // write to a spare word each so that the code stays straight-line
    lds_fence();
    float *mt = mg;
#pragma unroll
    for (int g = 0; g < 3; ++g) mt[band[g] < a.n_mel ? sub * a.n_mel + band[g] : 4 * a.n_mel + lane] = res[g];
    lds_fence();
    const int nv = (nfb - fb) < 4 ? (nfb - fb) : 4;
    float *dstf = a.mel + (a.frame_offs[u] + f0 + fb) * (int64_t)a.n_mel;
    if ((((uintptr_t)dstf) & 15) == 0 && (a.n_mel & 3) == 0) {
      for (int i = lane; i < nv * a.n_mel / 4; i += 64) ((float4 *)dstf)[i] = ((const float4 *)mt)[i];
    } else {
      for (int i = lane; i < nv * a.n_mel; i += 64) dstf[i] = mt[i];
    }
  }
  FE_STAMP(11)
}

// ---------------------------------------------------------------------------------------------------------------------
// logmel_rows_kernel (round 4): the fp64 front end with NOTHING shared between the waves of a workgroup.
//
// Why.  In the pipelined step the front end of batch i+1 runs in the shadow of batch i's crnn_fused_kernel, whose one
// wave per SIMD keeps the shared fp32-MFMA / vector datapath about half busy; how much of the other half the front end
// picks up is set by how many of its waves fit beside a CRNN workgroup.  logmel_kernel<f64> costs 10 KB of LDS and 128
// registers per wave (40 KB per 4-wave workgroup: the 16-frame sample tile + Hann table + three transpose buffers), so two
// workgroups = 8 waves fit.  This kernel costs 8.5 KB and 104 registers per wave and no workgroup barrier:
//   * a wave owns the four consecutive GLOBAL frames 4 W .. 4 W + 3 of the launch (mel rows are numbered through all
//     clips), so tiles run across clip boundaries: 37,632 frames = 9,408 full waves, none of the 256 three-frame tiles of
//     the per-clip tiling, and ragged batches leave no partly filled workgroups behind;
//   * it stages its own samples (<= 512 + 3 hop, two 16-byte loads per lane) into ITS transpose buffer, which is dead
//     until the first DFT pass is over: LDS instructions of one wave execute in order, so neither the hand-over of the
//     buffer from tile to transposes to magnitudes to the output tile nor the staging needs a barrier;
//   * Hann pairs and twiddles come from the L1-resident tables (4 + 4 + 4 KB, shared by every wave of the chip);
//   * 104 registers (amdgpu_num_vgpr counts in units of two on gfx90a+): the Hann products, the inter-pass twiddles and the
//     untangling twiddles are software-pipelined by hand in chunks of 4 / 3 / 2 instead of all at once, and the mel
//     weights are fetched group by group: three of these waves sit on a SIMD beside a 184-register CRNN wave, four alone.
// A wave whose four frames do not lie in one clip within 3 hops of each other (a clip boundary) stages them frame by frame
// (generic path, <= 1 wave in 37 for 1.5 s clips).  The arithmetic is logmel_kernel<f64>'s, instruction for instruction:
// results are bit-identical.
// ---------------------------------------------------------------------------------------------------------------------
#ifndef LW_WPB
#define LW_WPB 2  // waves per workgroup (they share nothing).  Same box, rocprofv3, 256 / 4,096 clips: 4 waves 27.96 / 353.5 us,
#endif            // 2 waves 27.2-27.5 / 350.5-352.0, 1 wave 27.6 / 350.9; logmel_kernel<f64> (rounds 1-3) 27.3-27.8 / 356.4-357.6

#ifndef LW_HC
#define LW_HC 4    // Hann pairs fetched per chunk (16 / LW_HC chunks, double-buffered)
#endif
#ifndef LW_TC
#define LW_TC 3    // inter-pass twiddles fetched per chunk (15 / LW_TC chunks, double-buffered)
#endif
#ifndef LW_VGPR
#define LW_VGPR 52  // amdgpu_num_vgpr counts in units of two on gfx90a+: 104 registers
#endif
#define LW_WBUF (4 * 16 * TR_LD * 8)  // 8,704 B per wave: 16x16 fp64 transposes of 4 frames; before that the sample tile
#define LW_ROWF 528                   // generic path: floats per staged frame (512 + up to 7 of shift, 16-byte multiple)
static_assert(4 * LW_ROWF * 4 <= LW_WBUF && (WIN + 3 * 512 + 16) * 4 <= LW_WBUF && 4 * MAG_LD * 4 <= LW_WBUF, "per-wave buffer too small");

// dst[q * VEC + e] = normalised (and pre-emphasised) sample ga + q * VEC + e, q < n_vec; ga is a multiple of VEC
template <bool F32IN>
__device__ __forceinline__ void lw_stage_generic(const logmel_args &a, float *dst, int64_t ga, int n_vec, int64_t s_begin,
                                                 int64_t total, int lane) {
  constexpr int VEC = F32IN ? 4 : 8;
  const float alpha = a.preemph;
  for (int q = lane; q < n_vec; q += 64) {
    const int64_t g = ga + (int64_t)q * VEC;
    float v[VEC + 1];
    // v[0] = sample g-1 (pre-emphasis carry; 0 at the start of the utterance)
    v[0] = (alpha != 0.0f && g - 1 >= s_begin) ? norm_sample<F32IN>(a, g - 1) : 0.0f;
    if (g + VEC <= total) {
      if (F32IN) {
        const float4 raw = *(const float4 *)(a.f32 + g);
        v[1] = raw.x; v[2] = raw.y; v[3] = raw.z; v[4] = raw.w;
      } else {
        const uint4 raw = *(const uint4 *)(a.pcm + g);
        const unsigned int w32[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int16_t s16 = (int16_t)((w32[e >> 1] >> ((e & 1) * 16)) & 0xffffu);
          float f = pcm_quot((float)s16, a);
          if (a.clip) f = fminf(fmaxf(f, -1.0f), 1.0f);
          v[1 + e] = f;
        }
      }
    } else {
#pragma unroll
      for (int e = 0; e < VEC; ++e) v[1 + e] = (g + e < total) ? norm_sample<F32IN>(a, g + e) : 0.0f;
    }
    float o[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      // reference: frame -= pre_emphasis * previous  (separate fp32 multiply and subtract); sample s_begin sees a zero carry
      const float prev = (g + e == s_begin) ? 0.0f : v[e];
      o[e] = (alpha != 0.0f) ? __fsub_rn(v[1 + e], __fmul_rn(alpha, prev)) : v[1 + e];
    }
    float4 *d4 = (float4 *)(dst + (size_t)q * VEC);
    d4[0] = make_float4(o[0], o[1], o[2], o[3]);
    if (VEC == 8) d4[1] = make_float4(o[4], o[5], o[6], o[7]);
  }
}

// The lane number, recomputed where it is needed (two instructions) instead of kept in a register across the transform:
// volatile, so the compiler cannot merge it with an earlier copy and carry that one through the register-tight phases.
__device__ __forceinline__ int lw_lane() {
  int l;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
  return l;
}

__device__ __forceinline__ int64_t lw_readlane64(int64_t v, int src_lane) {
  const int lo = __builtin_amdgcn_readlane((int)(v & 0xffffffffll), src_lane);
  const int hi = __builtin_amdgcn_readlane((int)(v >> 32), src_lane);
  return ((int64_t)hi << 32) | (unsigned int)lo;
}

__device__ __forceinline__ int lw_readlane64(int v, int src_lane) { return __builtin_amdgcn_readlane(v, src_lane); }
template <bool SMALL> struct lw_idx { typedef int64_t type; };
template <> struct lw_idx<true> { typedef int type; };  // every sample index of the launch fits 31 bits (the host checked)

// SMALL: sample indices in 32 bits and (equal clips) the row -> clip division as one multiply: the index arithmetic of a wave
// is vector instructions like everything else, ~100 of the ~960 it issues in 64-bit form.
template <bool F32IN, bool SIMPLE, bool SMALL = false>
__global__ __launch_bounds__(64 * LW_WPB) __attribute__((amdgpu_num_vgpr(LW_VGPR))) void logmel_rows_kernel(logmel_args a) {
  extern __shared__ __align__(16) unsigned char smem[];
  typedef double R;
  constexpr int VEC = F32IN ? 4 : 8;  // elements per 16-byte load
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform values in SGPRs
  const int j = lane & 15, sub = lane >> 4;
  // XCD-aware order: workgroup ids go round-robin to the 8 XCDs; ids of one residue get one contiguous eighth of the
  // frames, so the 352 samples consecutive waves share are fetched through one L2
  const int nper = gridDim.x >> 3;
  const int64_t W = ((int64_t)(blockIdx.x & 7) * nper + (blockIdx.x >> 3)) * LW_WPB + wave;
  const int64_t g0 = W * 4;
  if (g0 >= a.total_frames) return;  // no barrier anywhere below: a wave may simply leave
  unsigned char *wb = smem + (size_t)wave * LW_WBUF;
  float *tile = (float *)wb;
  FE_STAMP(0)

  // ---- which clip does this 16-lane row's frame belong to, and where do its samples start
  typedef typename lw_idx<SMALL>::type idx_t;
  idx_t s_begin, b;
  bool rv;
  {
    const int64_t g = g0 + sub;
    rv = g < a.total_frames;
    const int64_t gc = rv ? g : g0;
    if (SMALL && a.uni_magic) {
      // equal clips of >= 4 frames back to back: the wave's first row is a scalar, its clip comes out of one multiply, and a
      // row of the wave is in that clip or the next one
      const unsigned g0u = (unsigned)g0, nfu = (unsigned)a.uniform_nf;
      const unsigned u0 = __umulhi(g0u, a.uni_magic) >> a.uni_shift;
      unsigned f = g0u - u0 * nfu + (unsigned)sub;
      const bool next = f >= nfu;
      f -= next ? nfu : 0u;
      s_begin = (idx_t)((u0 + (next ? 1u : 0u)) * (unsigned)a.uniform_ns);
      b = s_begin + (idx_t)(f * (unsigned)a.hop);  // (rows past the launch's last one get the first row's values below)
    } else if (a.uniform_nf > 0) {  // equal-length clips back to back: arithmetic
      const unsigned gu = (unsigned)gc, nfu = (unsigned)a.uniform_nf;
      const unsigned u = gu / nfu, f = gu - u * nfu;
      s_begin = (idx_t)((int64_t)u * a.uniform_ns);
      b = s_begin + (idx_t)(f * (unsigned)a.hop);
    } else {
      int lo = 0, hi = a.n_utt;  // the last clip whose first mel row is <= g
      while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (a.frame_offs[mid] <= gc) lo = mid; else hi = mid;
      }
      const int64_t sb = a.sample_offs[lo];
      const int64_t f = gc - a.frame_offs[lo];
      const int64_t bb = sb + f * a.hop;
      rv = rv && f >= 0 && bb + WIN <= a.sample_offs[lo + 1];  // a row the offset tables do not cover is never stored
      s_begin = (idx_t)sb;
      b = (idx_t)bb;
    }
  }
  const unsigned long long vm = __ballot(rv);
  if (vm == 0) return;
  const int r_first = __builtin_ctzll(vm) >> 4, r_last = (63 - __builtin_clzll(vm)) >> 4;
  const idx_t b0 = lw_readlane64(b, 16 * r_first), s0 = lw_readlane64(s_begin, 16 * r_first);
  if (!rv) {  // surplus rows recompute the first valid frame (results unused)
    b = b0;
    s_begin = s0;
  }
  const idx_t total = (idx_t)a.sample_offs[a.n_utt];
  // one contiguous tile serves the wave when every row starts within 3 hops of the first one, in the same clip
  const bool contig = __all(s_begin == s0 && b >= b0 && b - b0 <= 3 * (idx_t)a.hop);
  const float *src;
  if (contig) {
    const idx_t bmax = lw_readlane64(b, 16 * r_last);
    const int shift = (int)(b0 % VEC);
    const idx_t ga = b0 - shift;  // multiple of VEC, >= 0
    const int n_vec = (shift + (int)(bmax - b0) + WIN + VEC - 1) / VEC;
    if (SIMPLE) {
      // No pre-emphasis, divisor 32767/32768, hop <= 168 (host checks): NV vectors per lane, straight-line: all loads in
      // flight together
      constexpr int NV = F32IN ? 4 : 2;
      const idx_t last = (total - VEC) & ~(idx_t)(VEC - 1);  // last full aligned vector (total >= WIN here)
      const idx_t tile_last = ga + (idx_t)(n_vec - 1) * VEC;
      idx_t gq[NV];
      uint4 raw[NV];
#pragma unroll
      for (int h = 0; h < NV; ++h) {
        gq[h] = ga + (idx_t)(lane + 64 * h) * VEC;
        idx_t gl = gq[h] < tile_last ? gq[h] : tile_last;  // never past this wave's last vector
        gl = gl < last ? gl : last;
        raw[h] = F32IN ? *(const uint4 *)(a.f32 + gl) : *(const uint4 *)(a.pcm + gl);
      }
      const float lim = a.clip ? 1.0f : __builtin_inff();
#pragma unroll
      for (int h = 0; h < NV; ++h) {
        const int q = lane + 64 * h;
        if (q < n_vec) {
          float o[VEC];
          if (gq[h] <= last) {
            const unsigned int w32[4] = {raw[h].x, raw[h].y, raw[h].z, raw[h].w};
            if (F32IN) {
#pragma unroll
              for (int e = 0; e < VEC; ++e) o[e] = __uint_as_float(w32[e]);
            } else {
              // the exact quotient (see pcm_quot) on sample pairs: v_pk_mul_f32 / v_pk_fma_f32, two samples per issue
              typedef float f32x2 __attribute__((ext_vector_type(2)));
              const f32x2 r2 = {a.rdiv, a.rdiv}, nb2 = {-a.divisor, -a.divisor};
#pragma unroll
              for (int e = 0; e < VEC; e += 2) {
                const f32x2 x = {(float)(int)(short)(w32[e >> 1] & 0xffffu), (float)((int)w32[e >> 1] >> 16)};
                const f32x2 q0 = x * r2;
                const f32x2 er = __builtin_elementwise_fma(nb2, q0, x);
                const f32x2 qq = __builtin_elementwise_fma(er, r2, q0);
                o[e] = __builtin_amdgcn_fmed3f(qq.x, -lim, lim);
                o[e + 1] = __builtin_amdgcn_fmed3f(qq.y, -lim, lim);
              }
            }
          } else {  // ragged end of the whole buffer: element-wise, zero beyond it
#pragma unroll
            for (int e = 0; e < VEC; ++e) o[e] = (gq[h] + e < total) ? norm_sample<F32IN>(a, (int64_t)gq[h] + e) : 0.0f;
          }
          float4 *d4 = (float4 *)(tile + (size_t)q * VEC);
          d4[0] = make_float4(o[0], o[1], o[2], o[3]);
          if (VEC == 8) d4[1] = make_float4(o[4], o[5], o[6], o[7]);
        }
      }
    } else {
      lw_stage_generic<F32IN>(a, tile, (int64_t)ga, n_vec, (int64_t)s0, (int64_t)total, lane);
    }
    src = tile + shift + (int)(b - b0);
  } else {
    // frames of two clips (or a very short one) in this wave: frame by frame
    for (int r = 0; r < 4; ++r) {
      if (!((vm >> (16 * r)) & 1)) continue;
      const int64_t br = lw_readlane64(b, 16 * r), sr = lw_readlane64(s_begin, 16 * r);
      const int sh = (int)(br % VEC);
      lw_stage_generic<F32IN>(a, tile + r * LW_ROWF, br - sh, (sh + WIN + VEC - 1) / VEC, sr, (int64_t)total, lane);
    }
    src = tile + (rv ? sub : r_first) * LW_ROWF + (int)(b % VEC);
  }
  FE_STAMP(1)
  lds_fence();
  FE_STAMP(2)

  // ---- FFT: every 16-lane row of the wave owns one frame
  R *tr = (R *)wb;          // [4][16][TR_LD]
  float *mg = (float *)wb;  // overlay: [4][MAG_LD]
  cplx<R> v[16];
  const double2 *twp = (const double2 *)a.tw16 + j;  // [k1][16 j] = W256^(j k1)
  double2 tq[2][LW_TC];
  {
    // pass 1: lane j holds z[16 n1 + j], n1 = 0..15; Hann product in fp64 (tflite.py:175).  Hann pairs (h[2n], h[2n+1]),
    // n = 16 n1 + j, from the half table of 128 pairs (h[m] = h[511 - m]) through the vector L1, four n1 at a time and one
    // chunk ahead: 16 + 16 registers instead of the 64 of the whole set
    const double2 *hb = (const double2 *)a.hann;
    auto hload = [&](int n1) -> double2 {
      if (n1 < 8) return hb[16 * n1 + j];
      const double2 m = hb[16 * (15 - n1) + 15 - j];
      return make_double2(m.y, m.x);
    };
    constexpr int HC = LW_HC, NHC = 16 / HC;  // Hann pairs per chunk; one chunk in flight ahead of the one being used
    double2 h[2][HC];
#pragma unroll
    for (int i = 0; i < HC; ++i) h[0][i] = hload(i);
    const bool pairs = __all((((int)(src - tile)) & 1) == 0);
    if (pairs) {
      // 8-byte aligned pairs: ds_read_b64 (with hop = 160 the four frames of a wave sit 32 banks apart: conflict-free)
      double xs[16];
      lds_read16_b64_s128(src + 2 * j, xs);
#pragma unroll
      for (int c = 0; c < NHC; ++c) {
        if (c + 1 < NHC) {
#pragma unroll
          for (int i = 0; i < HC; ++i) h[(c + 1) & 1][i] = hload(HC * (c + 1) + i);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (c == 0) lds_wait_all(xs);
#pragma unroll
        for (int i = 0; i < HC; ++i) {
          const int n1 = HC * c + i;
          v[n1].re = (R)__int_as_float(__double2loint(xs[n1])) * h[c & 1][i].x;
          v[n1].im = (R)__int_as_float(__double2hiint(xs[n1])) * h[c & 1][i].y;
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
#pragma unroll
      for (int c = 0; c < NHC; ++c) {
        if (c + 1 < NHC) {
#pragma unroll
          for (int i = 0; i < HC; ++i) h[(c + 1) & 1][i] = hload(HC * (c + 1) + i);
        }
#pragma unroll
        for (int i = 0; i < HC; ++i) {
          const int n = 16 * (HC * c + i) + j;
          v[HC * c + i].re = (R)src[2 * n] * h[c & 1][i].x;
          v[HC * c + i].im = (R)src[2 * n + 1] * h[c & 1][i].y;
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  FE_STAMP(3)
  FE_STAMP(4)
  {
    // the first inter-pass twiddles are requested before the butterflies that precede their use
    constexpr int TC = LW_TC, NTC = 15 / TC;
#pragma unroll
    for (int i = 0; i < TC; ++i) tq[0][i] = twp[k_of(1 + i) * 16];
    dft16<R>(v);
    FE_STAMP(5)
    // v[pos] *= W256^(j k_of(pos)): 15 sixteen-byte loads per lane from the L1-resident table, TC at a time, one chunk ahead
#pragma unroll
    for (int c = 0; c < NTC; ++c) {
      if (c + 1 < NTC) {
#pragma unroll
        for (int i = 0; i < TC; ++i) tq[(c + 1) & 1][i] = twp[k_of(1 + TC * (c + 1) + i) * 16];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < TC; ++i) {
        const int pos = 1 + TC * c + i;
        v[pos] = cmul(v[pos], cplx<R>{tq[c & 1][i].x, tq[c & 1][i].y});
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    FE_STAMP(6)
    // 16x16 transpose through LDS, real parts then imaginary parts (same buffer)
    cplx<R> w[16];
    const int l2 = lw_lane(), j = l2 & 15, sub = l2 >> 4;
    R *trs = tr + sub * 16 * TR_LD;
#pragma unroll
    for (int pos = 0; pos < 16; ++pos) trs[k_of(pos) * TR_LD + j] = v[pos].re;
    lds_fence();
    {
      double wre[16], wim[16];
      lds_read16_b64((const double *)trs + j * TR_LD, wre);
      lds_fence();
#pragma unroll
      for (int pos = 0; pos < 16; ++pos) trs[k_of(pos) * TR_LD + j] = v[pos].im;
      lds_fence();
      lds_read16_b64((const double *)trs + j * TR_LD, wim);
      lds_wait_all(wre);
      lds_wait_all(wim);
#pragma unroll
      for (int n2 = 0; n2 < 16; ++n2) {
        w[n2].re = wre[n2];
        w[n2].im = wim[n2];
      }
    }
    FE_STAMP(7)
    // pass 2: lane j = k1 holds Y[n2][k1]; output w[pos] = Z[k1 + 16 k_of(pos)]
    dft16<R>(w);
    FE_STAMP(8)
    // untangle (see logmel_kernel): partners Z[256 - k] travel through the dead transpose buffer
    const double2 *unp = (const double2 *)a.tw512 + j;  // W512^(j + 16 k2) at [16 k2]
    double2 uq[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i) uq[0][i] = unp[16 * i];
    const int pj = (16 - j) & 15;
    const R *prow0 = trs + (j == 0 ? 7 : 15) * TR_LD + pj;
    const R *prow = trs + (j == 0 ? 16 : 15) * TR_LD + pj;
    cplx<R> pz[8];
    trs[7 * TR_LD + j] = w[pos_of(0)].re;
#pragma unroll
    for (int r = 8; r < 16; ++r) trs[r * TR_LD + j] = w[pos_of(r)].re;
    lds_fence();
    pz[0].re = prow0[0];
#pragma unroll
    for (int k2 = 1; k2 < 8; ++k2) pz[k2].re = prow[-k2 * TR_LD];
    lds_fence();
    trs[7 * TR_LD + j] = w[pos_of(0)].im;
#pragma unroll
    for (int r = 8; r < 16; ++r) trs[r * TR_LD + j] = w[pos_of(r)].im;
    lds_fence();
    pz[0].im = prow0[0];
#pragma unroll
    for (int k2 = 1; k2 < 8; ++k2) pz[k2].im = prow[-k2 * TR_LD];
    lds_fence();
    FE_STAMP(9)
    float *mrow = mg + sub * MAG_LD;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      if (c + 1 < 4) {
#pragma unroll
        for (int i = 0; i < 2; ++i) uq[(c + 1) & 1][i] = unp[16 * (2 * (c + 1) + i)];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int k2 = 2 * c + i;
        const cplx<R> own = w[pos_of(k2)];
        const cplx<R> un = {uq[c & 1][i].x, uq[c & 1][i].y};
        const R er = own.re + pz[k2].re, ei = own.im - pz[k2].im;
        const R orr = own.im + pz[k2].im, oi = pz[k2].re - own.re;
        const R tr_ = orr * un.re - oi * un.im, ti_ = orr * un.im + oi * un.re;
        const R pr = er + tr_, pi = ei + ti_, qr = er - tr_, qi = ei - ti_;
        const int k = j + 16 * k2;
        mrow[k] = __builtin_amdgcn_sqrtf((float)(pr * pr + pi * pi));  // 2|X[k]|: the mel weights carry the 0.5
        mrow[256 - k] = __builtin_amdgcn_sqrtf((float)(qr * qr + qi * qi));  // k = 0 -> bin 256
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    {
      // k = 128 pairs with itself: X[128] = conj(Z[128]) (lane 0, k2 = 8); bins 257..271 are zero padding
      const cplx<R> z = w[pos_of(8)];
      if (j == 0) mrow[128] = 2.0f * __builtin_amdgcn_sqrtf((float)(z.re * z.re + z.im * z.im));
      else mrow[256 + j] = 0.0f;
    }
  }

  // ---- mel filterbank on the vector ALU (see logmel_kernel): lane 4 s + q owns frame q and slot s of each band group;
  //      every group's weight chunks are fetched when its turn comes
  FE_STAMP(10)
  lds_fence();
  {
    const int lane = lw_lane();
    const int j = lane >> 2, sub = lane & 3;  // mel phase only: (slot, frame) of this lane
    constexpr int CAPQ[3] = {9, 4, 3}, C0[3] = {0, 9, 13};
    const float4 *wv = (const float4 *)a.melV + j;
    const float *mrow = mg + sub * MAG_LD;
    __builtin_amdgcn_sched_barrier(0);
    int meta[3];
#pragma unroll
    for (int g = 0; g < 3; ++g) meta[g] = a.melVmeta[g * 16 + j];
    float4 wq[WW_MELV_CHUNKS];
    int band[3];
    float bias[3];
#pragma unroll
    for (int g = 0; g < 3; ++g) {
      band[g] = (int)((unsigned)meta[g] >> 16);  // 0xffff: empty slot
      bias[g] = a.bias[band[g] < a.n_mel ? band[g] : 0];
    }
    float res[3];
    if (a.melv_aligned) {
#pragma unroll
      for (int g = 0; g < 3; ++g) {
        __builtin_amdgcn_sched_barrier(0);  // one group's magnitudes and weights in flight at a time
#pragma unroll
        for (int c = 0; c < CAPQ[g]; ++c) wq[C0[g] + c] = wv[(C0[g] + c) * 16];
        lds_cfloat4 *mb = (lds_cfloat4 *)lds_opaque(mrow + (meta[g] & 0xffff));
        float acc = 0.f, acc1 = 0.f;  // two chains: a dependent fp32 FMA does not issue back to back
#pragma unroll
        for (int c = 0; c < CAPQ[g]; ++c) {
          const float4 w4 = wq[C0[g] + c];
          const f32x4 m4 = mb[c];
          acc = fmaf(m4[0], w4.x, acc);
          acc1 = fmaf(m4[1], w4.y, acc1);
          acc = fmaf(m4[2], w4.z, acc);
          acc1 = fmaf(m4[3], w4.w, acc1);
        }
        acc += acc1;
        res[g] = (logf(fmaxf(acc + bias[g], a.floor_v)) + a.log_off) * a.scale;
      }
    } else {
#pragma unroll
      for (int g = 0; g < 3; ++g) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int c = 0; c < CAPQ[g]; ++c) wq[C0[g] + c] = wv[(C0[g] + c) * 16];
        lds_cfloat *mb = lds_opaque(mrow + (meta[g] & 0xffff));
        float acc = 0.f;
#pragma unroll
        for (int c = 0; c < CAPQ[g]; ++c) {
          const float4 w4 = wq[C0[g] + c];
          acc = fmaf(mb[4 * c + 0], w4.x, acc);
          acc = fmaf(mb[4 * c + 1], w4.y, acc);
          acc = fmaf(mb[4 * c + 2], w4.z, acc);
          acc = fmaf(mb[4 * c + 3], w4.w, acc);
        }
        res[g] = (logf(fmaxf(acc + bias[g], a.floor_v)) + a.log_off) * a.scale;
      }
    }
    // park the wave's 4 x n_mel tile in LDS (its magnitudes are dead); empty slots write to a spare word each so that
    // the code stays straight-line.  Mel rows are global frame numbers: the four rows leave as one contiguous store
    // when all four are valid (16-byte aligned whenever n_mel is a multiple of 4: g0 is a multiple of 4)
    lds_fence();
    float *mt = mg;
#pragma unroll
    for (int g = 0; g < 3; ++g) mt[band[g] < a.n_mel ? sub * a.n_mel + band[g] : 4 * a.n_mel + lane] = res[g];
    lds_fence();
    float *dstf = a.mel + g0 * (int64_t)a.n_mel;
    const bool prefix = (vm & (vm + 1)) == 0;  // valid rows are 0 .. r_last
    if (prefix && (((uintptr_t)dstf) & 15) == 0 && (a.n_mel & 3) == 0) {
      const int nv = r_last + 1;
      for (int i = lane; i < nv * a.n_mel / 4; i += 64) ((float4 *)dstf)[i] = ((const float4 *)mt)[i];
    } else {
      for (int i = lane; i < 4 * a.n_mel; i += 64)
        if ((vm >> (16 * (i / a.n_mel))) & 1) dstf[i] = mt[i];
    }
  }
  FE_STAMP(11)
}

#ifndef WW_FE8
#define WW_FE8 0  // development: 1 = build and use logmel_rows8_kernel (below)
#endif
#if WW_FE8
// ---------------------------------------------------------------------------------------------------------------------
// logmel_rows8_kernel (round 5, a PROBE: compiled only with -DWW_FE8=1, tools/build_variant.sh fe8 "-DWW_FE8=1" frontend.hip) -
// the fp64 front end as 32 lanes x 8 points per frame, TWO frames per wave - half the
// registers of logmel_rows_kernel (its 16 complex fp64 points per lane are 64 registers before anything else), so that twice
// as many waves share a SIMD's latencies (profiles/r05/frontend_occupancy_probe.txt: t(w) = 225 + 503 / w us per 4,096 clips).
// 256 = 8 x 8 x 4: three DFT passes in registers, two transposes through the wave's own LDS buffer:
//   pass 1  lane n2 (0..31) holds z[32 n1 + n2], n1 = 0..7 (Hann product in fp64) -> 8-point DFT over n1 -> Y[k1][n2],
//           times W256^(n2 k1);   T1[k1][n2], row pitch 36 doubles: lane 4 k1 + m2 reads T1[k1][4 m1 + m2] conflict-free
//   pass 2  8-point DFT over m1 -> U[j1], times W32^(m2 j1);   T2[j1][4 k1 + m2], row pitch 33 doubles: lane 4 k1 + jp
//           reads T2[jp + 4 e][4 k1 + m2] conflict-free
//   pass 3  two 4-point DFTs over m2 (e = 0, 1) -> Z[k1 + 8 (jp + 4 e) + 64 j2]: lane column c = k1 + 8 jp holds the bins
//           c + 32 q, q = e + 2 j2 = 0..7
// Untangling: the partner of bin c + 32 q is 256 - k = (32 - c) + 32 (7 - q): column (32 - c) % 32 (column 0: itself, row
// 8 - q); a lane evaluates its q = 0..3 and their mirrors, so every pair is formed once.  Magnitudes, the mel filter (slots
// dealt as in logmel_rows_kernel: lane = slot + 16 half + 32 frame, half 0 = the 36-tap group, half 1 = the 16- and 12-tap
// groups) and the output tile go through the same buffer.  4.6 KB of LDS per wave.  Same formulas as logmel_rows_kernel; the
// transform's factorisation differs, so results agree to fp64 rounding (the same fp32 magnitude in all but ~1 of 10^7 bins).
// OUTCOME (profiles/r05/frontend_8point_probe.txt): correct (every front-end test passes on it) and NOT faster - 49.7 us per 256
// clips untuned against 27.6, and not tunable past the 16-point kernel: what a wave pays once (row -> clip lookup, staging its
// tile, the mel tail's three logarithms) is shared by two frames instead of four, 631 vector instructions per wave = 315 per
// frame against 229 (PMC), ~280 at best; its 78 registers give six waves per SIMD, not eight.  Kept for the record.
// ---------------------------------------------------------------------------------------------------------------------
#ifndef L8_WPB
#define L8_WPB 4    // waves per workgroup (they share nothing)
#endif
#ifndef L8_VGPR
#define L8_VGPR 32  // amdgpu_num_vgpr counts in units of two on gfx90a+: 64 registers = eight waves per SIMD
#endif
#define L8_T1 36
#define L8_T2 33
#define L8_FBUF (8 * L8_T1 * 8)   // 2,304 B per frame: T1 / T2 / partner rows, then its 272 magnitudes
#define L8_WBUF (2 * L8_FBUF)     // 4,608 B per wave; before the transform its sample tile
#define L8_ROWF 528
static_assert(2 * L8_ROWF * 4 <= L8_WBUF && (WIN + 512 + 16) * 4 <= L8_WBUF && MAG_LD * 4 <= L8_FBUF && 8 * L8_T2 * 8 <= L8_FBUF, "per-wave buffer too small");

// Forward 8-point DFT, in place, natural order (A[k] at position k).
template <typename R>
__device__ __forceinline__ void dft8(cplx<R> (&a)[8]) {
  constexpr R H = (R)0.70710678118654752440;
  radix4(a[0], a[2], a[4], a[6]);  // E[0..3] at 0, 2, 4, 6
  radix4(a[1], a[3], a[5], a[7]);  // O[0..3] at 1, 3, 5, 7
  const cplx<R> o1 = mulc(a[3], H, -H), o2 = cplx<R>{a[5].im, -a[5].re}, o3 = mulc(a[7], -H, -H);  // O[k] W8^k
  const cplx<R> e0 = a[0], e1 = a[2], e2 = a[4], e3 = a[6], o0 = a[1];
  a[0] = {e0.re + o0.re, e0.im + o0.im}; a[4] = {e0.re - o0.re, e0.im - o0.im};
  a[1] = {e1.re + o1.re, e1.im + o1.im}; a[5] = {e1.re - o1.re, e1.im - o1.im};
  a[2] = {e2.re + o2.re, e2.im + o2.im}; a[6] = {e2.re - o2.re, e2.im - o2.im};
  a[3] = {e3.re + o3.re, e3.im + o3.im}; a[7] = {e3.re - o3.re, e3.im - o3.im};
}

template <bool F32IN, bool SIMPLE, bool SMALL = false>
__global__ __launch_bounds__(64 * L8_WPB) __attribute__((amdgpu_num_vgpr(L8_VGPR))) void logmel_rows8_kernel(logmel_args a) {
  extern __shared__ __align__(16) unsigned char smem[];
  typedef double R;
  constexpr int VEC = F32IN ? 4 : 8;  // elements per 16-byte load
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int l32 = lane & 31, sub = lane >> 5;  // lane inside its frame, frame of the wave
  const int nper = gridDim.x >> 3;             // XCD-aware order, as logmel_rows_kernel
  const int64_t W = ((int64_t)(blockIdx.x & 7) * nper + (blockIdx.x >> 3)) * L8_WPB + wave;
  const int64_t g0 = W * 2;
  if (g0 >= a.total_frames) return;  // no barrier anywhere below: a wave may simply leave
  unsigned char *wb = smem + (size_t)wave * L8_WBUF;
  float *tile = (float *)wb;

  // ---- which clip does this 32-lane row's frame belong to, and where do its samples start (as logmel_rows_kernel)
  typedef typename lw_idx<SMALL>::type idx_t;
  idx_t s_begin, b;
  bool rv;
  {
    const int64_t g = g0 + sub;
    rv = g < a.total_frames;
    const int64_t gc = rv ? g : g0;
    if (SMALL && a.uni_magic) {
      const unsigned g0u = (unsigned)g0, nfu = (unsigned)a.uniform_nf;
      const unsigned u0 = __umulhi(g0u, a.uni_magic) >> a.uni_shift;
      unsigned f = g0u - u0 * nfu + (unsigned)sub;
      const bool next = f >= nfu;
      f -= next ? nfu : 0u;
      s_begin = (idx_t)((u0 + (next ? 1u : 0u)) * (unsigned)a.uniform_ns);
      b = s_begin + (idx_t)(f * (unsigned)a.hop);
    } else if (a.uniform_nf > 0) {
      const unsigned gu = (unsigned)gc, nfu = (unsigned)a.uniform_nf;
      const unsigned u = gu / nfu, f = gu - u * nfu;
      s_begin = (idx_t)((int64_t)u * a.uniform_ns);
      b = s_begin + (idx_t)(f * (unsigned)a.hop);
    } else {
      int lo = 0, hi = a.n_utt;  // the last clip whose first mel row is <= g
      while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (a.frame_offs[mid] <= gc) lo = mid; else hi = mid;
      }
      const int64_t sb = a.sample_offs[lo];
      const int64_t f = gc - a.frame_offs[lo];
      const int64_t bb = sb + f * a.hop;
      rv = rv && f >= 0 && bb + WIN <= a.sample_offs[lo + 1];  // a row the offset tables do not cover is never stored
      s_begin = (idx_t)sb;
      b = (idx_t)bb;
    }
  }
  const unsigned long long vm = __ballot(rv);
  if (vm == 0) return;
  const int r_first = __builtin_ctzll(vm) >> 5, r_last = (63 - __builtin_clzll(vm)) >> 5;
  const idx_t b0 = lw_readlane64(b, 32 * r_first), s0 = lw_readlane64(s_begin, 32 * r_first);
  if (!rv) {  // a surplus row recomputes the first valid frame (results unused)
    b = b0;
    s_begin = s0;
  }
  const idx_t total = (idx_t)a.sample_offs[a.n_utt];
  const bool contig = __all(s_begin == s0 && b >= b0 && b - b0 <= (idx_t)a.hop);
  const float *src;
  if (contig) {
    const idx_t bmax = lw_readlane64(b, 32 * r_last);
    const int shift = (int)(b0 % VEC);
    const idx_t ga = b0 - shift;  // multiple of VEC, >= 0
    const int n_vec = (shift + (int)(bmax - b0) + WIN + VEC - 1) / VEC;
    if (SIMPLE) {
      // no pre-emphasis, divisor 32767 / 32768, hop <= 168 (host checks): NV vectors per lane, all loads in flight together
      constexpr int NV = F32IN ? 3 : 2;
      const idx_t last = (total - VEC) & ~(idx_t)(VEC - 1);
      const idx_t tile_last = ga + (idx_t)(n_vec - 1) * VEC;
      idx_t gq[NV];
      uint4 raw[NV];
#pragma unroll
      for (int h = 0; h < NV; ++h) {
        gq[h] = ga + (idx_t)(lane + 64 * h) * VEC;
        idx_t gl = gq[h] < tile_last ? gq[h] : tile_last;
        gl = gl < last ? gl : last;
        raw[h] = F32IN ? *(const uint4 *)(a.f32 + gl) : *(const uint4 *)(a.pcm + gl);
      }
      const float lim = a.clip ? 1.0f : __builtin_inff();
#pragma unroll
      for (int h = 0; h < NV; ++h) {
        const int q = lane + 64 * h;
        if (q < n_vec) {
          float o[VEC];
          if (gq[h] <= last) {
            const unsigned int w32[4] = {raw[h].x, raw[h].y, raw[h].z, raw[h].w};
            if (F32IN) {
#pragma unroll
              for (int e = 0; e < VEC; ++e) o[e] = __uint_as_float(w32[e]);
            } else {
              typedef float f32x2 __attribute__((ext_vector_type(2)));
              const f32x2 r2 = {a.rdiv, a.rdiv}, nb2 = {-a.divisor, -a.divisor};
#pragma unroll
              for (int e = 0; e < VEC; e += 2) {
                const f32x2 x = {(float)(int)(short)(w32[e >> 1] & 0xffffu), (float)((int)w32[e >> 1] >> 16)};
                const f32x2 q0 = x * r2;
                const f32x2 er = __builtin_elementwise_fma(nb2, q0, x);
                const f32x2 qq = __builtin_elementwise_fma(er, r2, q0);
                o[e] = __builtin_amdgcn_fmed3f(qq.x, -lim, lim);
                o[e + 1] = __builtin_amdgcn_fmed3f(qq.y, -lim, lim);
              }
            }
          } else {
#pragma unroll
            for (int e = 0; e < VEC; ++e) o[e] = (gq[h] + e < total) ? norm_sample<F32IN>(a, (int64_t)gq[h] + e) : 0.0f;
          }
          float4 *d4 = (float4 *)(tile + (size_t)q * VEC);
          d4[0] = make_float4(o[0], o[1], o[2], o[3]);
          if (VEC == 8) d4[1] = make_float4(o[4], o[5], o[6], o[7]);
        }
      }
    } else {
      lw_stage_generic<F32IN>(a, tile, (int64_t)ga, n_vec, (int64_t)s0, (int64_t)total, lane);
    }
    src = tile + shift + (int)(b - b0);
  } else {
    // frames of two clips in this wave: frame by frame
    for (int r = 0; r < 2; ++r) {
      if (!((vm >> (32 * r)) & 1)) continue;
      const int64_t br = lw_readlane64(b, 32 * r), sr = lw_readlane64(s_begin, 32 * r);
      const int sh = (int)(br % VEC);
      lw_stage_generic<F32IN>(a, tile + r * L8_ROWF, br - sh, (sh + WIN + VEC - 1) / VEC, sr, (int64_t)total, lane);
    }
    src = tile + (rv ? sub : r_first) * L8_ROWF + (int)(b % VEC);
  }
  lds_fence();

  // ---- pass 1: z[32 n1 + n2] = (x[2n] h[2n], x[2n + 1] h[2n + 1]) in fp64 (tflite.py:175), 8-point DFT over n1
  R *fb = (R *)(wb + (size_t)sub * L8_FBUF);  // this frame's buffer
  cplx<R> v[8];
  {
    const double2 *hb = (const double2 *)a.hann;  // half table of 128 pairs (h[m] = h[511 - m])
    float2 xs[8];
#pragma unroll
    for (int n1 = 0; n1 < 8; ++n1) {
      const float *p = src + 64 * n1 + 2 * l32;
      xs[n1] = make_float2(p[0], p[1]);
    }
    lds_fence();  // every sample is in registers: the tile is dead from here on
    // Hann pairs four at a time (sixteen registers instead of thirty-two at once)
#pragma unroll
    for (int c4 = 0; c4 < 2; ++c4) {
      double2 h[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int n1 = 4 * c4 + i;
        if (n1 < 4) {
          h[i] = hb[32 * n1 + l32];
        } else {
          const double2 m = hb[32 * (7 - n1) + 31 - l32];
          h[i] = make_double2(m.y, m.x);
        }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        v[4 * c4 + i].re = (R)xs[4 * c4 + i].x * h[i].x;
        v[4 * c4 + i].im = (R)xs[4 * c4 + i].y * h[i].y;
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const double2 *tw = (const double2 *)a.tw256;  // [256] e^{-2 pi i k / 256}
  dft8<R>(v);
  // inter-pass twiddles W256^(n2 k1), k1 = 1..7: four, then three (the table lives in the vector L1)
  {
    double2 t1[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) t1[i] = tw[(l32 * (1 + i)) & 255];
#pragma unroll
    for (int i = 0; i < 4; ++i) v[1 + i] = cmul(v[1 + i], cplx<R>{t1[i].x, t1[i].y});
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 3; ++i) t1[i] = tw[(l32 * (5 + i)) & 255];
#pragma unroll
    for (int i = 0; i < 3; ++i) v[5 + i] = cmul(v[5 + i], cplx<R>{t1[i].x, t1[i].y});
  }
  // ---- transpose 1 (real parts, then imaginary parts, same buffer): T1[k1][n2] -> lane 4 k1 + m2 takes T1[k1][4 m1 + m2]
  {
    const int k1r = l32 >> 2, m2 = l32 & 3;
    const R *rd = fb + k1r * L8_T1 + m2;
    R re[8];
#pragma unroll
    for (int k1 = 0; k1 < 8; ++k1) fb[k1 * L8_T1 + l32] = v[k1].re;
    lds_fence();
#pragma unroll
    for (int m1 = 0; m1 < 8; ++m1) re[m1] = rd[4 * m1];
    lds_fence();
#pragma unroll
    for (int k1 = 0; k1 < 8; ++k1) fb[k1 * L8_T1 + l32] = v[k1].im;
    lds_fence();
#pragma unroll
    for (int m1 = 0; m1 < 8; ++m1) {
      v[m1].im = rd[4 * m1];
      v[m1].re = re[m1];
    }
    lds_fence();
    // ---- pass 2: 8-point DFT over m1, times W32^(m2 j1)
    dft8<R>(v);
    double2 t2[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) t2[i] = tw[8 * m2 * (1 + i)];
#pragma unroll
    for (int i = 0; i < 4; ++i) v[1 + i] = cmul(v[1 + i], cplx<R>{t2[i].x, t2[i].y});
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 3; ++i) t2[i] = tw[8 * m2 * (5 + i)];
#pragma unroll
    for (int i = 0; i < 3; ++i) v[5 + i] = cmul(v[5 + i], cplx<R>{t2[i].x, t2[i].y});
  }
  // ---- transpose 2: T2[j1][4 k1 + m2] -> lane 4 k1 + jp takes T2[jp + 4 e][4 k1 + m2], m2 = 0..3, e = 0, 1
  const int jp = l32 & 3, c = (l32 >> 2) + 8 * jp;  // this lane's bin column: it ends up with Z[c + 32 q], q = e + 2 j2
  {
    const R *rd = fb + jp * L8_T2 + (l32 & ~3);
    R re[8];
#pragma unroll
    for (int j1 = 0; j1 < 8; ++j1) fb[j1 * L8_T2 + l32] = v[j1].re;
    lds_fence();
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
      for (int m2 = 0; m2 < 4; ++m2) re[4 * e + m2] = rd[4 * e * L8_T2 + m2];
    lds_fence();
#pragma unroll
    for (int j1 = 0; j1 < 8; ++j1) fb[j1 * L8_T2 + l32] = v[j1].im;
    lds_fence();
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
      for (int m2 = 0; m2 < 4; ++m2) {
        v[4 * e + m2].im = rd[4 * e * L8_T2 + m2];
        v[4 * e + m2].re = re[4 * e + m2];
      }
    lds_fence();
    // ---- pass 3: 4-point DFTs over m2: v[4 e + j2] = Z[c + 32 (e + 2 j2)]
    radix4(v[0], v[1], v[2], v[3]);
    radix4(v[4], v[5], v[6], v[7]);
  }
  // ---- untangle: partner rows through the buffer, P[q][column] (pitch 36), q = e + 2 j2 <-> v[4 e + j2]
  float *mrow = (float *)fb;  // overlay (after the partner reads): this frame's magnitudes [MAG_LD]
  {
    const int pc = (32 - c) & 31;
    cplx<R> pz[4];
    // partner of own row q (0..3): row 7 - q of column pc; column 0 pairs with itself: row (8 - q) % 8
    const R *prd = fb + pc;
    const int prow[4] = {c == 0 ? 0 : 7, c == 0 ? 7 : 6, c == 0 ? 6 : 5, c == 0 ? 5 : 4};
#pragma unroll
    for (int q = 0; q < 8; ++q) fb[q * L8_T1 + c] = v[4 * (q & 1) + (q >> 1)].re;
    lds_fence();
#pragma unroll
    for (int q = 0; q < 4; ++q) pz[q].re = prd[prow[q] * L8_T1];
    lds_fence();
#pragma unroll
    for (int q = 0; q < 8; ++q) fb[q * L8_T1 + c] = v[4 * (q & 1) + (q >> 1)].im;
    lds_fence();
#pragma unroll
    for (int q = 0; q < 4; ++q) pz[q].im = prd[prow[q] * L8_T1];
    lds_fence();
    const double2 *unp = (const double2 *)a.tw512 + c;  // W512^(c + 32 q) at [32 q]
    double2 un[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) un[q] = unp[32 * q];
    const cplx<R> z4 = v[2];  // q = 4 (e = 0, j2 = 2): bin 128 in column 0
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const cplx<R> own = v[4 * (q & 1) + (q >> 1)];
      const R er = own.re + pz[q].re, ei = own.im - pz[q].im;
      const R orr = own.im + pz[q].im, oi = pz[q].re - own.re;
      const R tr_ = orr * un[q].x - oi * un[q].y, ti_ = orr * un[q].y + oi * un[q].x;
      const R pr = er + tr_, pi = ei + ti_, qr = er - tr_, qi = ei - ti_;
      const int k = c + 32 * q;
      mrow[k] = __builtin_amdgcn_sqrtf((float)(pr * pr + pi * pi));        // 2|X[k]|: the mel weights carry the 0.5
      mrow[256 - k] = __builtin_amdgcn_sqrtf((float)(qr * qr + qi * qi));  // k = 0 -> bin 256
    }
    // bin 128 pairs with itself: X[128] = conj(Z[128]) (column 0, row 4); bins 257..271 are zero padding
    if (c == 0) mrow[128] = 2.0f * __builtin_amdgcn_sqrtf((float)(z4.re * z4.re + z4.im * z4.im));
    else if (c < 16) mrow[256 + c] = 0.0f;
  }
  lds_fence();

  // ---- mel filterbank on the vector ALU: lane = slot + 16 half + 32 frame; half 0 = the 36-tap group (9 float4 chunks),
  //      half 1 = the 16- and 12-tap groups (4 + 3 chunks); tables as logmel_rows_kernel's
  {
    const int slot = lane & 15, half = (lane >> 4) & 1;
    const float4 *wv = (const float4 *)a.melV + slot;
    const float *mr = (const float *)(wb + (size_t)sub * L8_FBUF);
    float res[2];
    int band[2];
    auto group = [&](int g, int c0, int nq) -> float {  // one band slot: nq chunks of 4 taps from chunk c0 of the table
      const int meta = a.melVmeta[g * 16 + slot];
      const int bnd = (int)((unsigned)meta >> 16);
      const float bias = a.bias[bnd < a.n_mel ? bnd : 0];
      const float *mb = mr + (meta & 0xffff);
      float acc = 0.f, acc1 = 0.f;
      if (a.melv_aligned) {
        for (int q = 0; q < nq; ++q) {
          const float4 w4 = wv[(c0 + q) * 16];
          const float4 m4 = *(const float4 *)(mb + 4 * q);
          acc = fmaf(m4.x, w4.x, acc);
          acc1 = fmaf(m4.y, w4.y, acc1);
          acc = fmaf(m4.z, w4.z, acc);
          acc1 = fmaf(m4.w, w4.w, acc1);
        }
        acc += acc1;
      } else {
        for (int q = 0; q < nq; ++q) {
          const float4 w4 = wv[(c0 + q) * 16];
          acc = fmaf(mb[4 * q + 0], w4.x, acc);
          acc = fmaf(mb[4 * q + 1], w4.y, acc);
          acc = fmaf(mb[4 * q + 2], w4.z, acc);
          acc = fmaf(mb[4 * q + 3], w4.w, acc);
        }
      }
      band[g == 2 ? 1 : 0] = bnd;
      return (logf(fmaxf(acc + bias, a.floor_v)) + a.log_off) * a.scale;
    };
    if (half == 0) {
      res[0] = group(0, 0, 9);
      res[1] = 0.f;
      band[1] = 0xffff;
    } else {
      res[0] = group(1, 9, 4);
      res[1] = group(2, 13, 3);
    }
    // park the wave's 2 x n_mel tile in LDS (frame 0's buffer; every magnitude has been read), then one contiguous store
    lds_fence();
    float *mt = (float *)wb;
#pragma unroll
    for (int i = 0; i < 2; ++i) mt[band[i] < a.n_mel ? sub * a.n_mel + band[i] : 2 * a.n_mel + lane] = res[i];
    lds_fence();
    float *dstf = a.mel + g0 * (int64_t)a.n_mel;
    const bool both = (vm >> 32) & (vm & 1);
    if (both && (((uintptr_t)dstf) & 15) == 0 && (a.n_mel & 3) == 0) {
      for (int i = lane; i < 2 * a.n_mel / 4; i += 64) ((float4 *)dstf)[i] = ((const float4 *)mt)[i];
    } else {
      for (int i = lane; i < 2 * a.n_mel; i += 64)
        if ((vm >> (32 * (i / a.n_mel))) & 1) dstf[i] = mt[i];
    }
  }
}

#endif  // WW_FE8

// STFT magnitude of explicit frames [n][512] -> [n][257]; one wave per frame.
template <typename R>
__global__ __launch_bounds__(256) void stft_mag_kernel(logmel_args a) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  cplx<R> *fbuf = (cplx<R> *)smem;
  float *mag = (float *)(smem + WAVES * FFT_LD * sizeof(cplx<R>));
  fft_consts<R> fc;
  fft_load_consts<R>(fc, lane, a.hann, a.tw256, a.tw512);
  const int64_t f = (int64_t)blockIdx.x * WAVES + wave;
  if (f >= a.n_frames_direct) return;
  const float *src = a.frames + f * WIN;
  auto x2 = [&](int n) -> float2 { return *(const float2 *)(src + 2 * n); };
  float *mg = mag + wave * 260;
  frame_fft_mag<R>(x2, fc, fbuf + wave * FFT_LD, mg, lane);
  float *dst = a.mag_out + f * NB;
  for (int k = lane; k < NB; k += 64) dst[k] = mg[k];
}

template <typename R>
static size_t logmel_smem(int hop) {
  size_t off = 0;
  off += 128 * sizeof(double2);
  off += sizeof(R) == 4 ? 2 * 256 * sizeof(cplx<R>) : 0;
  off += (WAVES - 1) * wbuf_bytes<R>();
  size_t tile_b = (size_t)(WIN + (FPB - 1) * hop + 16) * 4;
  if (tile_b < (size_t)wbuf_bytes<R>()) tile_b = wbuf_bytes<R>();  // wave 3's buffer moves into the tile
  off += tile_b;
  return (off + 15) & ~size_t(15);
}

static void fill_filter_args(logmel_args &a, const ww_model *m) {
  const ww_filter_dev &f = m->filt;
  a.start = f.start; a.wpad = f.wpad; a.bias = f.bias;
  a.n_mel = f.n_mel;
  a.floor_v = f.floor_v; a.log_off = f.log_off; a.scale = f.scale;
  a.hann = f.hann; a.tw256 = f.tw256; a.tw512 = f.tw512; a.tw16 = f.tw16;
  a.melV = f.melV; a.melVmeta = f.melVmeta; a.melv_aligned = f.melv_aligned;
}

int ww_k_logmel(ww_ctx *ctx, const ww_model *m, const int16_t *d_pcm, const float *d_f32, const int64_t *d_sample_offs,
                const int64_t *d_frame_offs, int n_utt, int64_t total_frames, int64_t max_frames_per_utt,
                const ww_frontend_params *fp, float *d_mel, int64_t uniform_samples, int64_t total_samples_hint) {
  if (n_utt <= 0 || total_frames <= 0 || max_frames_per_utt <= 0) return WW_OK;
  if (fp->hop <= 0 || fp->hop > 512) return ww_fail(ctx, WW_EINVAL, "hop %d out of range (1..512)", fp->hop);
  logmel_args a = {};
  a.pcm = d_pcm; a.f32 = d_f32; a.sample_offs = d_sample_offs; a.frame_offs = d_frame_offs;
  a.n_utt = n_utt; a.hop = fp->hop; a.divisor = fp->pcm_divisor; a.clip = fp->clip; a.preemph = fp->pre_emphasis;
  a.rdiv = 1.0f / fp->pcm_divisor;
  a.fast_div = (fp->pcm_divisor == 32767.0f || fp->pcm_divisor == 32768.0f) ? 1 : 0;
  a.mel = d_mel;
  fill_filter_args(a, m);
  const bool f32in = d_f32 != nullptr;
  // straight-line staging (two vectors per thread) when nothing exotic is asked for
  const bool simple = fp->pre_emphasis == 0.0f && (f32in || a.fast_div) && WIN + (FPB - 1) * fp->hop + 16 <= 512 * (f32in ? 4 : 8);
#ifndef WW_FE_OLD
#define WW_FE_OLD 0  // development: 1 = logmel_kernel<f64> (rounds 1-3) for A/B runs
#endif
  if (fp->precise && !WW_FE_OLD) {
    // fp64: waves own four consecutive global mel rows each (logmel_rows_kernel)
    a.total_frames = total_frames;
    if (uniform_samples > 0 && total_frames < 0x7fffffff && total_frames == (int64_t)n_utt * max_frames_per_utt) {
      a.uniform_nf = (int)max_frames_per_utt;
      a.uniform_ns = uniform_samples;
    }
    const bool simple_w = fp->pre_emphasis == 0.0f && (f32in || a.fast_div) && fp->hop <= 168;
    // 32-bit sample indices when the caller could tell that every index of the launch fits 31 bits; with equal clips of >= 4
    // frames the row -> clip division becomes one multiply: M = ceil(2^(31 + l) / nf), l = ceil(log2 nf), is exact for
    // every row < 2^31 (M nf - 2^(31 + l) < nf <= 2^l)
    const bool small_w = simple_w && total_samples_hint > 0 && total_samples_hint < 0x7fff0000ll;
    if (small_w && a.uniform_nf >= 4) {
      int l = 0;
      while ((1ll << l) < a.uniform_nf) ++l;
      a.uni_magic = (unsigned)(((1ull << (31 + l)) + (unsigned)a.uniform_nf - 1) / (unsigned)a.uniform_nf);
      a.uni_shift = l - 1;
    }
#if WW_FE8
    {
      const int64_t n_waves8 = (total_frames + 1) / 2;
      const int64_t n_wg8 = 8 * ((((n_waves8 + L8_WPB - 1) / L8_WPB) + 7) / 8);
      if (n_wg8 > 0x7fffffff) return ww_fail(ctx, WW_EINVAL, "front-end launch too large (%lld workgroups): split the batch", (long long)n_wg8);
      const dim3 grid8((unsigned)n_wg8), block8(64 * L8_WPB);
      const size_t sm8 = (size_t)L8_WPB * L8_WBUF;
      ww_launch_scope scope(ctx, "logmel_rows8_kernel");
      if (small_w) {
        if (f32in) hipLaunchKernelGGL((logmel_rows8_kernel<true, true, true>), grid8, block8, sm8, ctx->stream, a);
        else hipLaunchKernelGGL((logmel_rows8_kernel<false, true, true>), grid8, block8, sm8, ctx->stream, a);
      } else if (simple_w) {
        if (f32in) hipLaunchKernelGGL((logmel_rows8_kernel<true, true>), grid8, block8, sm8, ctx->stream, a);
        else hipLaunchKernelGGL((logmel_rows8_kernel<false, true>), grid8, block8, sm8, ctx->stream, a);
      } else {
        if (f32in) hipLaunchKernelGGL((logmel_rows8_kernel<true, false>), grid8, block8, sm8, ctx->stream, a);
        else hipLaunchKernelGGL((logmel_rows8_kernel<false, false>), grid8, block8, sm8, ctx->stream, a);
      }
      WW_HIP(ctx, hipGetLastError());
      return WW_OK;
    }
#endif
    const int64_t n_waves = (total_frames + 3) / 4;
    const int64_t n_wg = 8 * ((((n_waves + LW_WPB - 1) / LW_WPB) + 7) / 8);
    if (n_wg > 0x7fffffff) return ww_fail(ctx, WW_EINVAL, "front-end launch too large (%lld workgroups): split the batch", (long long)n_wg);
    const dim3 grid_w((unsigned)n_wg), block_w(64 * LW_WPB);
#ifndef LW_LDS_PAD
#define LW_LDS_PAD 0  // development: extra LDS per workgroup - fewer resident waves with the same instruction stream (occupancy probe)
#endif
    const size_t sm = (size_t)LW_WPB * LW_WBUF + LW_LDS_PAD;
    ww_launch_scope scope(ctx, "logmel_rows_kernel");
    if (small_w) {
      if (f32in) hipLaunchKernelGGL((logmel_rows_kernel<true, true, true>), grid_w, block_w, sm, ctx->stream, a);
      else hipLaunchKernelGGL((logmel_rows_kernel<false, true, true>), grid_w, block_w, sm, ctx->stream, a);
    } else if (simple_w) {
      if (f32in) hipLaunchKernelGGL((logmel_rows_kernel<true, true>), grid_w, block_w, sm, ctx->stream, a);
      else hipLaunchKernelGGL((logmel_rows_kernel<false, true>), grid_w, block_w, sm, ctx->stream, a);
    } else {
      if (f32in) hipLaunchKernelGGL((logmel_rows_kernel<true, false>), grid_w, block_w, sm, ctx->stream, a);
      else hipLaunchKernelGGL((logmel_rows_kernel<false, false>), grid_w, block_w, sm, ctx->stream, a);
    }
    WW_HIP(ctx, hipGetLastError());
    return WW_OK;
  }
  const int64_t tiles = (max_frames_per_utt + FPB - 1) / FPB;
  const int64_t n_ids = 8 * (((int64_t)n_utt + 7) / 8) * tiles;
  if (n_ids > 0x7fffffff) return ww_fail(ctx, WW_EINVAL, "front-end launch too large (%lld workgroups): split the batch", (long long)n_ids);
  a.tiles_per_utt = (int)tiles;
  dim3 grid((unsigned)n_ids);
#if WW_FE_STAMPS
  // development build: WWHIP_FE_STAMPS=1 prints the mean phase timeline of the launch (cycles since the wave's first stamp)
  const bool want_stamps = getenv("WWHIP_FE_STAMPS") != nullptr;
  if (want_stamps) {
    WW_HIP(ctx, hipMalloc((void **)&a.stamps, (size_t)n_ids * 48 * sizeof(long long)));
    WW_HIP(ctx, hipMemsetAsync(a.stamps, 0, (size_t)n_ids * 48 * sizeof(long long), ctx->stream));
  }
  struct stamp_dump {
    ww_ctx *ctx; long long *d; int64_t n;
    ~stamp_dump() {
      if (!d) return;
      std::vector<long long> h((size_t)n * 48);
      hipStreamSynchronize(ctx->stream);
      hipMemcpy(h.data(), d, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
      hipFree(d);
      double sum[12] = {0};
      long long cnt = 0, tmin = 0, tmax = 0;
      for (int64_t w = 0; w < n * 4; ++w) {
        const long long *t = &h[(size_t)w * 12];
        if (!t[0] || !t[11]) continue;  // inactive wave / workgroup
        for (int i = 0; i < 12; ++i) sum[i] += (double)(t[i] - t[0]);
        if (!cnt || t[0] < tmin) tmin = t[0];
        if (!cnt || t[11] > tmax) tmax = t[11];
        ++cnt;
      }
      fprintf(stderr, "fe stamps: %lld waves, mean cycles since entry:", cnt);
      for (int i = 0; i < 12; ++i) fprintf(stderr, " %.0f", sum[i] / (cnt ? cnt : 1));
      fprintf(stderr, "  | first entry -> last exit (clocks of different XCDs differ): %lld\n", tmax - tmin);
    }
  } dump{ctx, a.stamps, n_ids};
#endif
  ww_launch_scope scope(ctx, fp->precise ? "logmel_kernel<f64>" : "logmel_kernel<f32>");
  if (fp->precise) {
    size_t sm = logmel_smem<double>(fp->hop);
    if (simple) {
      if (f32in) hipLaunchKernelGGL((logmel_kernel<double, true, true>), grid, dim3(256), sm, ctx->stream, a);
      else hipLaunchKernelGGL((logmel_kernel<double, false, true>), grid, dim3(256), sm, ctx->stream, a);
    } else {
      if (f32in) hipLaunchKernelGGL((logmel_kernel<double, true, false>), grid, dim3(256), sm, ctx->stream, a);
      else hipLaunchKernelGGL((logmel_kernel<double, false, false>), grid, dim3(256), sm, ctx->stream, a);
    }
  } else {
    size_t sm = logmel_smem<float>(fp->hop);
    if (simple) {
      if (f32in) hipLaunchKernelGGL((logmel_kernel<float, true, true>), grid, dim3(256), sm, ctx->stream, a);
      else hipLaunchKernelGGL((logmel_kernel<float, false, true>), grid, dim3(256), sm, ctx->stream, a);
    } else {
      if (f32in) hipLaunchKernelGGL((logmel_kernel<float, true, false>), grid, dim3(256), sm, ctx->stream, a);
      else hipLaunchKernelGGL((logmel_kernel<float, false, false>), grid, dim3(256), sm, ctx->stream, a);
    }
  }
  WW_HIP(ctx, hipGetLastError());
  return WW_OK;
}

// filter.tflite alone (reference filter_model(frame), wakeword/tflite.py:183-184): mag [n][257] -> mel [n][40]
__global__ __launch_bounds__(64) void mel_only_kernel(const float *mag, int64_t n, const float *w, const float *bias, int n_mel,
                                                      int n_bins, float floor_v, float log_off, float scale, float *mel) {
  __shared__ float m[260];
  const int lane = threadIdx.x;
  const int64_t f = blockIdx.x;
  for (int k = lane; k < n_bins; k += 64) m[k] = mag[f * n_bins + k];
  __syncthreads();
  if (lane < n_mel) {
    float acc = 0.f;
    for (int k = 0; k < n_bins; ++k) acc = fmaf(w[(size_t)lane * n_bins + k], m[k], acc);
    acc = fmaxf(acc + bias[lane], floor_v);
    mel[f * n_mel + lane] = (logf(acc) + log_off) * scale;
  }
}

int ww_k_mel_only(ww_ctx *ctx, const ww_model *m, const float *d_mag, int64_t n, float *d_mel) {
  if (n <= 0) return WW_OK;
  const ww_filter_dev &f = m->filt;
  ww_launch_scope scope(ctx, "mel_only_kernel");
  hipLaunchKernelGGL(mel_only_kernel, dim3((unsigned)n), dim3(64), 0, ctx->stream, d_mag, n, f.wdense, f.bias, f.n_mel, f.n_bins,
                     f.floor_v, f.log_off, f.scale, d_mel);
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
    size_t sm = WAVES * FFT_LD * sizeof(cplx<double>) + WAVES * 260 * sizeof(float);
    hipLaunchKernelGGL((stft_mag_kernel<double>), grid, dim3(256), sm, ctx->stream, a);
  } else {
    size_t sm = WAVES * FFT_LD * sizeof(cplx<float>) + WAVES * 260 * sizeof(float);
    hipLaunchKernelGGL((stft_mag_kernel<float>), grid, dim3(256), sm, ctx->stream, a);
  }
  WW_HIP(ctx, hipGetLastError());
  return WW_OK;
}
