// Wavenet encode + detect for gfx950 (fp32 MFMA, one persistent workgroup per window).
//
// Replaces encode.tflite + detect.tflite of the reference Wavenet (tf_lite_models/Wavenet;
// architecture wwdetect/wavenet/wavenet_model.py:11-128; call sites
// spokestack/wakeword/tflite.py:205-231, utils/evaluate_models.py:83-86).
//
// A 768-thread workgroup (12 wavefronts, 3 per SIMD) walks all 24 gated residual blocks of one
// 182x40 window without leaving the CU.  Time is the MFMA M dimension: 182 -> 12 tiles of 16
// rows, one tile per wave.  Two arithmetic modes (ww_model_set_precision):
//
// fp32 (default): v_mfma_f32_16x16x4_f32.  The residual stream x[182][16] and the skip accumulator
// [182][32] never leave registers: they sit in the accumulator layout (lane -> column, 4 rows per
// register quad), which is also the layout the next block's epilogue needs.  Only the BatchNorm
// output u (which the dilated taps of OTHER rows read) and the gate product g (D-layout ->
// A-layout transpose) go through LDS:
//     u = x*s + t                      -> LDS (double buffered, 16 zero rows in front = causal pad)
//     [sig|tanh] = u[t-(2-k)d] * Wg    3 taps x 16 ch = K 48, N 32     24 MFMA / 16 rows
//     g = tanh(.)*sigmoid(.)           -> LDS (wave-private tile)
//     [res|skip] = g * Wrs             K 16, N 48                      12 MFMA / 16 rows
//     x += relu(res); skip += relu(skip_b)
// One __syncthreads per block.  On gfx950 the fp32 MFMA shares the SIMD's fp32 datapath with the
// vector ALU, so MFMA and VALU time add up (ablations: dropping the 24 gate MFMAs saves exactly their
// 21.7 us of 77; dropping the gate transcendentals or the barrier saves < 1 us; hand-interleaving
// VALU into one wave's MFMA gaps made it slower): the kernel runs at ~86 % of that sum.
//
// split-bf16 ("bf16x3"): see the block before the kernel - transposed block loop on
// v_mfma_f32_16x16x32_bf16, operands straight from registers, parameters through LDS pages.
//
// The detect head (ReLU, 1x1 32->32 ReLU, 1x1 32->2, max over time, softmax) runs in the same launch.
#include "common.h"
#include "fft_device.h"
#undef NB   // (fft_device.h: bins of the transform; here NB is the model's block count)
#undef WIN

#include <type_traits>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define WV_T 192      // padded time (12 tiles of 16)
#define WV_C 16
#define WV_S 32
#define WV_PAD 16     // causal zero rows in front of u
#define WV_INLD 48    // staged input row stride (40 mel + zero pad to 3 k-blocks)
// wavefronts per workgroup = template parameter NW of the kernel: the 12 row tiles of 16 frames are dealt 12 / NW per wave.
// Both modes run 12 waves x 1 tile (3 waves per SIMD).  The split-bf16 loop is written over the tiles of a wave, so
// -DWV_BF16_NW=6 or 4 builds the 2- and 3-tile forms (independent MFMAs / gate evaluations back to back in one wave, a third of
// the operand reads): measured 45.2 and 44.6 us against 38.9 us for 12 x 1 - a wave's LDS and MFMA -> VALU latencies are
// covered better by two more waves on the SIMD than by two more tiles in the wave.
#ifndef WV_BF16_NW
#define WV_BF16_NW 12
#endif
#ifndef WV_F32_WIDE_FROM
#define WV_F32_WIDE_FROM 256   // the fp32 transposed loop likewise (round 5)
#endif
#ifndef WV_BF16_WIDE_FROM
#define WV_BF16_WIDE_FROM 256  // launches of more windows than this (= CUs of the chip) take the 4-wave x 3-tile form, two workgroups per CU
#endif
#ifndef WV_BF16_OCC
#define WV_BF16_OCC 3   // waves per SIMD the split-bf16 kernel is compiled for (3 = one workgroup per CU)
#endif

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
  const uint4 *wpk;     // split-bf16 mode: parameter pages [NB][WV_PAGE_U4] (A operands of v_mfma_f32_16x16x32_bf16, then the vectors)
  long long *stamps;    // development (-DWV_STAMPS=1): [windows][12 waves][WV_STAMP_NB blocks][12] s_memtime inside the split-bf16 loop
  ww_tick_tag tag;      // streaming ticks: the posterior as a {value, tick number} pair instead of the row of `out`
  // TICK != 0 - ONE launch per tick (round 5): the streaming front end's side (common.h) and the model's filterbank
  ww_tick_fe fe;
  const int *mel_start;
  const float *mel_wpad, *mel_bias;
  float floor_v, log_off, scale;
  const double *hann, *tw256, *tw512;
};

// Mel-side LDS of the one-launch tick form, behind the staged input [WV_T][WV_INLD] (floats from `lds`); dead before the block loop
#define WT_X (WV_T * WV_INLD)          // [WW_ST_RING] ring | the tick's new samples
#define WT_XS (WT_X + WW_ST_RING)      // [WW_CHUNK] int16: the raw samples
#define WT_WL (WT_XS + WW_CHUNK / 2)   // [768 x 4] the mel weights [WW_MEL_TAPS][64], padded to one store round of 12 waves
#define WT_MAG (WT_WL + 768 * 4)       // [2][260] magnitudes of the (at most) two new frames
#define WT_BUF (WT_MAG + 2 * 260 + 8)  // [2][FFT_LD] complex (16-byte aligned)
#define WT_END (WT_BUF + 2 * FFT_LD * 4)
static_assert(WT_BUF % 4 == 0 && WW_MEL_TAPS * 64 <= 768 * 4, "tick front end: LDS layout");

#ifndef WV_STAMPS
#define WV_STAMPS 0
#endif
// probes of round 6 (profiles/r06/wavenet_bf16x3_probes.txt): development builds only (tools/build_variant.sh)
#ifndef WV_PROBE_NOBAR
#define WV_PROBE_NOBAR 0
#endif
#ifndef WV_PROBE_ACC2
#define WV_PROBE_ACC2 0
#endif
#ifndef WV_PROBE_RCP1
#define WV_PROBE_RCP1 0
#endif
#define WV_STAMP_NB 24
#if WV_STAMPS
// Stamp i_ of every block (round 6: all blocks, so that the table can separate the dilations); stamp 11 = the top of the NEXT
// block, written into this block's row.  Reading s_memtime waits for the wave's outstanding LDS operations as well (one
// counter): a stamped build runs the same instructions with every counted wait turned into a full one - the table's total per
// block stands beside the unstamped kernel's so that the price of looking is on record.
#define WV_STAMP(i_)                                                                                                   \
  {                                                                                                                    \
    __builtin_amdgcn_sched_barrier(0);                                                                                 \
    if (a.stamps && lane == 0 && blk - ((i_) == 11 ? 1 : 0) >= 0 && blk < WV_STAMP_NB + ((i_) == 11 ? 1 : 0))          \
      a.stamps[(((size_t)blockIdx.x * 12 + wave) * WV_STAMP_NB + blk - ((i_) == 11 ? 1 : 0)) * 12 + (i_)] =           \
          __builtin_amdgcn_s_memtime();                                                                                \
    __builtin_amdgcn_sched_barrier(0);                                                                                 \
  }
#else
#define WV_STAMP(i_)
#endif

__device__ __forceinline__ float sigmoid_w(float x) { return 1.0f / (1.0f + expf(-x)); }

__device__ __forceinline__ void wsync() {
  // LDS traffic of one wave is processed in order; wait for it only (not for outstanding
  // global loads, which an acq_rel fence would also drain) and stop compiler reordering
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
}

// one wave's LDS operations execute in issue order: a scheduling fence is all a wave-private
// write -> read (or read -> overwrite) round trip needs
__device__ __forceinline__ void wsync_fence() {
  asm volatile("" ::: "memory");
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

// (block-uniform base pointers + one 32-bit per-lane offset each: the loads take SGPR base + VGPR offset + immediate, no
//  64-bit address arithmetic on the vector ALU - in the fp32 kernel every vector instruction costs matrix time)
__device__ __forceinline__ void wave_blk_load(const wave_args &a, int blk, int j, int kk, wave_blk &p) {
  const float *wg = a.w_gate4 + (size_t)blk * 3 * 4 * 32 * 4, *wrs = a.w_rs4 + (size_t)blk * 4 * 48 * 4;
  const unsigned og = (unsigned)(kk * 32 + j) * 4, ors = (unsigned)(kk * 48 + j) * 4;
#pragma unroll
  for (int kb = 0; kb < 3; ++kb)
#pragma unroll
    for (int n = 0; n < 2; ++n) p.wg[kb][n] = *(const float4 *)(wg + og + kb * 4 * 32 * 4 + n * 16 * 4);
#pragma unroll
  for (int n = 0; n < 3; ++n) p.wrs[n] = *(const float4 *)(wrs + ors + n * 16 * 4);
  const float *bn_s = a.bn_s + blk * WV_C, *bn_t = a.bn_t + blk * WV_C, *bg = a.b_gate + blk * 32, *brs = a.b_rs + blk * 48;
  const unsigned uj = (unsigned)j;
  p.bn_s = bn_s[uj];
  p.bn_t = bn_t[uj];
  p.bsig = bg[uj];
  p.btanh = bg[uj + 16];
  p.bres = brs[uj];
  p.bsk0 = brs[uj + 16];
  p.bsk1 = brs[uj + 32];
}


// ---- split-bf16 contractions ("bf16x3") --------------------------------------------------------
// x = hi + lo with hi = bf16(x), lo = bf16(x - hi) carries 16 mantissa bits; a*b is evaluated as
// ah*bh + al*bh + ah*bl by three bf16 MFMAs with fp32 accumulate.  Unlike v_mfma_f32_16x16x4_f32,
// which shares the SIMD's fp32 datapath with the vector ALU (their times add up), the bf16 MFMA
// runs beside it.  Error model and measurements: tools/bf16x3_error.py, DESIGN.md.
//
// The block loop is evaluated TRANSPOSED (channels x time) with v_mfma_f32_16x16x16_bf16: its
// accumulator layout (lane: column n = lane & 15, rows 4*(lane >> 4) + r) has, per lane, exactly
// the four k-values (4*kg .. 4*kg + 3, kg = lane >> 4) its B operand wants for the same column.
// With weights as the A operand (rows = output channels) and activations as B (columns = time),
// the BatchNorm output of a tile IS the undelayed tap's B operand and the gate product IS the
// res/skip conv's B operand - both straight out of registers.  Only the two delayed taps read
// LDS (other time columns, possibly another wave's tile): two 8-byte writes and four 8-byte reads
// per wave and block, one barrier.
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// four fp32 values -> (hi, lo) as the 4 x bf16 operand registers
__device__ __forceinline__ void split4(const float (&v)[4], s16x4 &hi, s16x4 &lo) {
  const bf16x2 h01 = __builtin_convertvector((f32x2){v[0], v[1]}, bf16x2);
  const bf16x2 h23 = __builtin_convertvector((f32x2){v[2], v[3]}, bf16x2);
  const unsigned u01 = __builtin_bit_cast(unsigned, h01), u23 = __builtin_bit_cast(unsigned, h23);
  const bf16x2 l01 = __builtin_convertvector((f32x2){v[0] - __uint_as_float(u01 << 16), v[1] - __uint_as_float(u01 & 0xffff0000u)}, bf16x2);
  const bf16x2 l23 = __builtin_convertvector((f32x2){v[2] - __uint_as_float(u23 << 16), v[3] - __uint_as_float(u23 & 0xffff0000u)}, bf16x2);
  const uint2 hv = {u01, u23}, lv = {__builtin_bit_cast(unsigned, l01), __builtin_bit_cast(unsigned, l23)};
  hi = __builtin_bit_cast(s16x4, hv);
  lo = __builtin_bit_cast(s16x4, lv);
}

// v_mfma_f32_16x16x32_bf16 (16 cycles; the K = 16 form costs twice that per MFMA on gfx950, measured):
// its 8 k-slots per lane are filled with TWO such 4-channel groups - k-slots 0..3 of lane group kg
// = channels 4 kg .. 4 kg + 3 of one tap, k-slots 4..7 = the same channels of a second tap (or zeros) -
// which is just a concatenation of two register pairs; the host packs the weights to match.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
struct s16x4_pair { s16x4 lo, hi; };
__device__ __forceinline__ bf16x8 cat8(s16x4 first, s16x4 second) {
  const s16x4_pair p = {first, second};
  return __builtin_bit_cast(bf16x8, p);
}
#define MFMA_BF(acc, av, bv) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv, acc, 0, 0, 0)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define WV_SLOTS 14  // A-operand slots per block: gate (2 k-steps x {sig,tanh} x 2) = 8, res | skip (3 m-tiles x 2) = 6
#define WV_PAGE_U4 (WV_SLOTS * 64)  // one block's parameter page in 16-byte units (the conv biases sit in padded k-slots)

// FP32T: the fp32 block loop in the TRANSPOSED form of the split-bf16 loop (channels x time; round 3) - see its comment below.
// TICK = 1 / 2 (fp32 / fp64 transform) - ONE launch per streaming tick (round 5; crnn.hip's crnn_stream_kernel<FE> has the full
// story): workgroup 2 s + k is window k of stream s's tick; it reads the stream's control words and samples over the bus, waves
// 0 and 1 transform the new frames (one each) straight into the staged input while the others stage the rows that were
// there before; the workgroup of the tick's newest window alone writes the stream's state (mel rows, sample ring and carry -
// the latter two ping-pong by the stream's parity, so its sibling still reads last tick's).  Twelve waves x one tile only.
template <bool HEAD_ONLY, bool SPLIT_BF16, int WV_NW, bool FP32T = false, int TICK = 0>
__global__ __launch_bounds__(WV_NW * 64, WV_NW == 12 ? (SPLIT_BF16 ? WV_BF16_OCC : 3) : 2) void wavenet_kernel(wave_args a) {
  constexpr int WV_MPW = 12 / WV_NW, WV_THREADS = WV_NW * 64;
  static_assert(!TICK || (WV_NW == 12 && !HEAD_ONLY), "the one-launch tick runs twelve waves x one tile");
  constexpr bool TRANSPOSED = SPLIT_BF16 || FP32T;  // state layout: lane = time column, four consecutive channels per register quad
  static_assert(!(SPLIT_BF16 && FP32T), "one arithmetic mode");
  static_assert(WV_MPW * WV_NW == 12, "12 row tiles");
  // LDS: region A = staged input [192][48] (prologue only), later u[2][208][16] + g[192][16]
  // split-bf16: u planes (same bytes as the fp32 u buffers) + two parameter pages (next to / under the head's tile)
  constexpr int LDS_FP32 = WV_T * WV_INLD > (2 * (WV_T + WV_PAD) * WV_C + WV_T * WV_S) ? WV_T * WV_INLD
                                                                                       : (2 * (WV_T + WV_PAD) * WV_C + WV_T * WV_S);
  constexpr int LDS_BF16 = 2 * (WV_T + WV_PAD) * WV_C + 3 * WV_PAGE_U4 * 4 + 32 * 8 * 4;  // u planes + 3 pages + BatchNorm table (NB <= 32)
  constexpr int LDS_F32T = 2 * (WV_T + WV_PAD) * WV_C + WV_T * WV_S + 32 * 7 * 16;        // u buffers + head tile + per-block vectors (NB <= 32)
  constexpr int LDS_A = SPLIT_BF16 && LDS_BF16 > LDS_FP32 ? LDS_BF16 : LDS_FP32;
  constexpr int LDS_B = FP32T && LDS_F32T > LDS_A ? LDS_F32T : LDS_A;
  __shared__ __align__(16) float lds[TICK && WT_END > LDS_B ? WT_END : LDS_B];
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
  int64_t row = 0;
  int valid = 0;
  // ---- TICK: over the bus, together: the tick's 320 samples (40 x 16 bytes) and the stream's control words
  typedef typename std::conditional<TICK == 2, double, float>::type RT;
  uint4 t_raw = make_uint4(0u, 0u, 0u, 0u);
  int4 t_cw = make_int4(0, 0, 0, 0);
  if (TICK) {
    if (tid < 40) t_raw = ((const uint4 *)(a.fe.frames + (size_t)(w >> 1) * WW_CHUNK))[tid];
    t_cw = ((const int4 *)a.fe.ctl)[w >> 1];
  } else if (a.wa.row && a.wa.valid) {  // (both tables: the two loads go out together)
    const int64_t r_ = a.wa.row[w];
    const int v_ = a.wa.valid[w];
    row = r_;
    valid = v_;
  } else {  // (plain ifs: as `p ? p[w] : constant` the compiler parked the constant in scratch memory to select between two addresses)
    row = a.wa.row0 + (int64_t)w * a.wa.hop;
    valid = a.wa.valid_const;
    if (a.wa.row) row = a.wa.row[w];
    if (a.wa.valid) valid = a.wa.valid[w];
  }
  if (!TICK) {
    if (valid > T) valid = T;
    if (row + valid > a.wa.mel_rows) valid = (int)(a.wa.mel_rows - row);
    if (valid < 0) valid = 0;
  }

  // the input conv's operands: requested now, used behind the staging
  float4 bw[3];
#pragma unroll
  for (int kb = 0; kb < 3; ++kb) bw[kb] = *(const float4 *)(a.w_in4 + ((size_t)(kb * 4 + kk) * 16 + j) * 4);
  const float bias = a.b_in[j];
  // ... and what the block loop's LDS tables are filled from (split-bf16: parameter pages 0 and 1 and the BatchNorm table; fp32
  // transposed: the per-block vector table): requested here, parked in LDS once the staged input is dead - as loops of
  // "load, store" behind the input conv they were three to four round trips to L2 in a row on every window's critical path
  constexpr int NPL = (WV_PAGE_U4 + WV_THREADS - 1) / WV_THREADS;                 // page pieces per thread (the last one partial)
  static_assert(NPL >= 2 && NPL <= 4, "page pieces per thread");
  constexpr int NVT = (32 * 7 * 16 + WV_THREADS - 1) / WV_THREADS;                // fp32 vector-table entries per thread (NB <= 32)
  // (clang ext-vector elements: arrays of HIP's struct vector types stayed in scratch memory)
  u32x4 pg0[NPL], pg1[NPL];
  f32x4 bnv = {0.f, 0.f, 0.f, 0.f};
  float vte[NVT];
  if (SPLIT_BF16) {
    const int second = a.NB > 1 ? 1 : 0;
#pragma unroll
    for (int q = 0; q < NPL; ++q) {
      const int i = tid + q * WV_THREADS < WV_PAGE_U4 ? tid + q * WV_THREADS : WV_PAGE_U4 - 1;
      pg0[q] = *(const u32x4 *)(a.wpk + i);
      pg1[q] = *(const u32x4 *)(a.wpk + (size_t)second * WV_PAGE_U4 + i);
    }
    const int bi = tid < a.NB * 8 ? tid : 0;                                    // [NB][2][4] float4 = scale, shift (NB <= 32 <= threads / 8)
    bnv = *(const f32x4 *)(((bi >> 2) & 1 ? a.bn_t : a.bn_s) + (bi >> 3) * WV_C + 4 * (bi & 3));
  }
  if (FP32T) {
#pragma unroll
    for (int q = 0; q < NVT; ++q) {
      int i = tid + q * WV_THREADS;
      i = i < a.NB * 7 * 16 ? i : 0;
      const int b = i / 112, v = (i / 16) % 7, c = i & 15;
      const float *p = v == 0 ? a.bn_s + b * WV_C + c : v == 1 ? a.bn_t + b * WV_C + c : v < 4 ? a.b_gate + b * 32 + (v - 2) * 16 + c
                                                                                             : a.b_rs + b * 48 + (v - 4) * 16 + c;
      vte[q] = *p;
    }
  }

  // ---- stage the window: in_lds[t][0..47], zero outside [0,valid) x [0,n_mel)
  float *in_lds = lds;
  if constexpr (TICK != 0) {
    const ww_tick_fe &fe = a.fe;
    const int s = w >> 1, k = w & 1;
    // device-side inputs of the front end that do not depend on the control words: requested while those cross the bus
    const f32x4 wlq = ((const f32x4 *)a.mel_wpad)[tid < WW_MEL_TAPS * 64 / 4 ? tid : 0];
    const int mel_st = lane < a.n_mel ? a.mel_start[lane] : 0;
    const float mel_bias = lane < a.n_mel ? a.mel_bias[lane] : 0.0f;
    // the sample ring: threads 0..127 ask for the copy of parity 0, threads 128..255 for parity 1 (128 x 4 = 512 > fill)
    const f32x4 ringq = ((const f32x4 *)(fe.ring + ((size_t)((tid >> 7) & 1) * fe.S + s) * WW_ST_RING))[tid & 127];
    const float carry0 = fe.prev[s], carry1 = fe.prev[fe.S + s];
    fft_consts<RT> fc;
    if (wave < 2) fft_load_consts<RT>(fc, lane, a.hann, a.tw256, a.tw512);
    // ---- what this workgroup is (uniform over it; as crnn_stream_kernel<FE>)
    const int fill = t_cw.x, nf = t_cw.y, flags = t_cw.z, pos = t_cw.w & 0xffff;
    const int par = (flags >> 2) & 1;
    const int np = (flags & 1) ? nf : 0;  // frames are analysed only while the VAD says speech (tflite.py:166)
    if ((flags & 2) || k >= (np > 1 ? np : 1)) return;  // an active stream is not sampled at all (tflite.py:139-140) | no second window
    const bool window = k < np, writer = k + 1 >= np;
    const int nfk = window ? k + 1 : 0;  // window k ends at new frame k: it needs frames 0..k
    const int slots = T + 1;
    // the rows that were there before: the block [(pos + k + 2) % (T + 1), + T - nfk) of the stream's mirrored ring
    constexpr int SQ = (WV_T * WV_INLD / 4 + WV_THREADS - 1) / WV_THREADS;
    f32x4 st[SQ];
    int n4 = 0;
    if (window) {
      int b = pos + k + 2;
      b = b >= slots ? b - slots : b;
      const float *src = a.mel + ((size_t)s * fe.HR + b) * a.n_mel;  // (160-byte rows of a hipMalloc'ed history: 16-byte aligned)
      n4 = ((T - nfk) * a.n_mel) >> 2;
#pragma unroll
      for (int q = 0; q < SQ; ++q) {
        const int i = tid + q * WV_THREADS;
        st[q] = *(const f32x4 *)(src + 4 * (i < n4 ? i : n4 - 1));
      }
    }
    float *fx = lds + WT_X, *fwl = lds + WT_WL, *fmag = lds + WT_MAG;
    short *fxs = (short *)(lds + WT_XS);
    cplx<RT> *fbuf = (cplx<RT> *)(lds + WT_BUF);
    ((f32x4 *)fwl)[tid] = wlq;
    if (tid < 256 && (tid >> 7) == par) ((f32x4 *)fx)[tid & 127] = ringq;
    if (tid < 40) ((uint4 *)fxs)[tid] = t_raw;
    if (window)
      for (int i = tid; i < WV_T * WV_INLD / 4; i += WV_THREADS) ((float4 *)in_lds)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();
    // ---- [ring | new samples]: normalise, clip, pre-emphasise (the arithmetic of stream_frontend_kernel, streams.hip)
    for (int i = tid; i < WW_CHUNK; i += WV_THREADS) {
      float v = __fdiv_rn((float)fxs[i], fe.divisor);
      if (fe.clip) v = fminf(fmaxf(v, -1.0f), 1.0f);
      float p;
      if (i == 0) {
        p = par ? carry1 : carry0;
      } else {
        p = __fdiv_rn((float)fxs[i - 1], fe.divisor);
        if (fe.clip) p = fminf(fmaxf(p, -1.0f), 1.0f);
      }
      fx[fill + i] = (fe.preemph != 0.0f) ? __fsub_rn(v, __fmul_rn(fe.preemph, p)) : v;
    }
    if (window) {  // (the zero fill is complete: the old rows go in beside the normalisation)
#pragma unroll
      for (int q = 0; q < SQ; ++q) {
        const int i = tid + q * WV_THREADS;
        if (i < n4) {
          const int e = i * 4, t = e / a.n_mel, c = e - t * a.n_mel;
          *(f32x4 *)(in_lds + t * WV_INLD + c) = st[q];
        }
      }
    }
    __syncthreads();
    if (writer && tid == 0) {
      float v = __fdiv_rn((float)fxs[WW_CHUNK - 1], fe.divisor);
      if (fe.clip) v = fminf(fmaxf(v, -1.0f), 1.0f);
      fe.prev[(size_t)(par ^ 1) * fe.S + s] = v;  // tflite.py:156-158: the carry is the un-emphasised last sample
    }
    // ---- new frames: wave f transforms frame f; its mel row goes straight into the staged input (row T - nfk + f) and, from
    // the writer, into the stream's mirrored ring
    if (wave < nfk) {
      const float *srcx = fx + wave * fe.hop;
      auto x2 = [&](int n) -> float2 { return make_float2(srcx[2 * n], srcx[2 * n + 1]); };
      float *mg = fmag + wave * 260;
      frame_fft_mag<RT>(x2, fc, fbuf + wave * FFT_LD, mg, lane);
      const float mv = mel_band(mg, fwl, mel_st, mel_bias, a.floor_v, a.log_off, a.scale, lane);
      if (lane < a.n_mel) {
        in_lds[(T - nfk + wave) * WV_INLD + lane] = mv;
        if (writer) {
          int p = pos + wave;  // mirrored ring: the row goes to p % slots and p % slots + slots
          p = p >= slots ? p - slots : p;
          float *h = fe.hist + ((size_t)s * fe.HR + p) * a.n_mel + lane;
          h[0] = mv;
          h[(size_t)slots * a.n_mel] = mv;
        }
      }
    }
    if (writer) {  // keep the ring tail (for the next tick: the other copy)
      const int keep = fill + WW_CHUNK - nf * fe.hop;
      float *ring = fe.ring + ((size_t)(par ^ 1) * fe.S + s) * WW_ST_RING;
      for (int i = tid; i < keep; i += WV_THREADS) ring[i] = fx[nf * fe.hop + i];
    }
    if (!window) return;  // the tick has no window for this stream: its ring has advanced, that is all
  } else {
    const float *src = a.mel + row * a.n_mel;
    const int n = valid * a.n_mel;
    if ((a.n_mel & 3) == 0 && ((((uintptr_t)src) & 15) == 0)) {
      // the window is one contiguous [valid][n_mel] block and a row is a whole number of float4s.  All of a thread's
      // 16-byte loads are issued first (unconditional, from clamped addresses: nothing for the next load to wait for), the
      // zero fill runs while they are in flight, then the 16-byte LDS stores
      constexpr int SQ = (WV_T * WV_INLD / 4 + WV_THREADS - 1) / WV_THREADS;
      const int n4 = n >> 2;
      f32x4 st[SQ];
#pragma unroll
      for (int q = 0; q < SQ; ++q) st[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (n4 > 0) {
#pragma unroll
        for (int q = 0; q < SQ; ++q) {
          const int i = tid + q * WV_THREADS;
          st[q] = *(const f32x4 *)(src + 4 * (i < n4 ? i : n4 - 1));
        }
      }
      for (int i = tid; i < WV_T * WV_INLD / 4; i += WV_THREADS) ((float4 *)in_lds)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      __syncthreads();
#pragma unroll
      for (int q = 0; q < SQ; ++q) {
        const int i = tid + q * WV_THREADS;
        if (i < n4) {
          const int e = i * 4, t = e / a.n_mel, c = e - t * a.n_mel;
          *(f32x4 *)(in_lds + t * WV_INLD + c) = st[q];
        }
      }
    } else {
      for (int i = tid; i < WV_T * WV_INLD / 4; i += WV_THREADS) ((float4 *)in_lds)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      __syncthreads();
      for (int i = tid; i < n; i += WV_THREADS) {
        int t = i / a.n_mel, c = i - t * a.n_mel;
        in_lds[t * WV_INLD + c] = src[i];
      }
    }
  }
  __syncthreads();

  // ---- input 1x1 conv + ReLU -> x in accumulator layout.  m-tile mi of this wave covers rows
  // (wave*3 + mi)*16 .. +15; lane holds rows kk*4 + r, column j.
  {
#pragma unroll
    for (int mi = 0; mi < WV_MPW; ++mi) {
      const int t0 = (wave * WV_MPW + mi) * 16;
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kb = 0; kb < 3; ++kb) {
        const float4 av = *(const float4 *)(in_lds + (t0 + j) * WV_INLD + kb * 16 + kk * 4);
        if (TRANSPOSED) {  // operands swapped: the transposed tile (channels x time)
          MFMA4(acc, bw[kb], av);
        } else {
          MFMA4(acc, av, bw[kb]);
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) x[mi][r] = fmaxf(acc[r] + (TRANSPOSED ? a.b_in[kk * 4 + r] : bias), 0.f);
      skip[mi][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
      skip[mi][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
  }
  __syncthreads();  // in_lds is dead from here on

  if (FP32T) {
    // ---- fp32 block loop, transposed (lane = time column t0 + j, registers = channels 4 kk + r), on v_mfma_f32_16x16x4_f32 with
    //      the weights as the A operand.  The packed weights of the row-major loop serve as they are: k-step q of lane group
    //      kk is input channel 4 kk + q in both forms.  What the transposition buys (as in the split-bf16 loop): the BatchNorm
    //      output of a tile IS the undelayed tap's B operand and the gate product IS the res | skip conv's B operand - both
    //      straight from registers.  LDS per block and wave: ONE 16-byte write (u, for the delayed taps of other rows) and
    //      TWO 16-byte reads (rows t - 2d, t - d), against 8 four-byte writes, 4 sixteen-byte reads and two wave-private
    //      round trips (u -> A operand, gate product -> A operand) in the row-major loop.  u is stored channel-group-major,
    //      [kk][row][4 floats] with the four planes a multiple of 256 bytes apart: consecutive lanes of a group touch
    //      consecutive 16 bytes and the lane groups of ds_read_b128 / ds_write_b128 (MI355X_MICROARCH.md, LDS) cover all 64 banks
    //      once (row-major [row][16] put the 8 lanes of a write group on 2 bank quads: 5.3 M conflict cycles per 256 windows).  The conv biases are the
    //      accumulators' initial values (per-lane float4 by channel group); all blocks' small vectors sit in one LDS table.
    float *vtab = lds + 2 * (WV_T + WV_PAD) * WV_C + WV_T * WV_S;  // [NB][7][16]: bn_s, bn_t, b_sig, b_tanh, b_res, b_skip0, b_skip1
#pragma unroll
    for (int q = 0; q < NVT; ++q) {
      const int i = tid + q * WV_THREADS;
      if (i < a.NB * 7 * 16) vtab[i] = vte[q];
    }
    constexpr int UPL = (WV_T + WV_PAD) * 4;  // floats per channel-group plane
    static_assert(UPL % 64 == 0, "u planes must start on the same bank");
    for (int i = tid; i < 2 * 4 * WV_PAD * 4; i += WV_THREADS) {  // causal zero rows of both u buffers, every plane
      const int b = i / (4 * WV_PAD * 4), k = (i / (WV_PAD * 4)) & 3, o = i % (WV_PAD * 4);
      ubuf[b * 4 * UPL + k * UPL + o] = 0.f;
    }
    struct wblk_t { float4 wg[3][2], wrs[3]; };
    auto wload = [&](int blk, wblk_t &p) {
      const float *wg = a.w_gate4 + (size_t)blk * 3 * 4 * 32 * 4, *wrs = a.w_rs4 + (size_t)blk * 4 * 48 * 4;
      const unsigned og = (unsigned)(kk * 32 + j) * 4, ors = (unsigned)(kk * 48 + j) * 4;
#pragma unroll
      for (int kb = 0; kb < 3; ++kb)
#pragma unroll
        for (int n = 0; n < 2; ++n) p.wg[kb][n] = *(const float4 *)(wg + og + kb * 4 * 32 * 4 + n * 16 * 4);
#pragma unroll
      for (int n = 0; n < 3; ++n) p.wrs[n] = *(const float4 *)(wrs + ors + n * 16 * 4);
    };
    wblk_t pw[2];
    wload(0, pw[0]);
    const int tl = wave * WV_MPW * 16 + j;  // this lane's time column in the wave's first tile (tile mi: + 16 mi)
    __syncthreads();               // table + zero rows
    // a block's seven small vectors (this lane's channel group): read from the table one block AHEAD, behind the barrier, so
    // that no LDS round trip sits in front of the u write, the accumulators' initial values or the res | skip products
    // (the BatchNorm pair and the gate biases, which a block needs at once; the res | skip biases are requested at the top of their
    // own block and used ~1,500 cycles later: prefetching all seven costs 56 registers and spills)
    struct vblk_t { float4 bns, bnt, bsig, btanh; };
    auto vload = [&](int blk, vblk_t &v) {
      const float4 *vt = (const float4 *)(vtab + blk * 112) + kk;  // vector q: vt[4 q]
      v.bns = vt[0]; v.bnt = vt[4]; v.bsig = vt[8]; v.btanh = vt[12];
    };
    vblk_t pv[2];
    vload(0, pv[0]);
    auto f4 = [](const float4 &v) { return (f32x4){v.x, v.y, v.z, v.w}; };
    // Written over a wave's WV_MPW tiles (round 5: launches of more than 256 windows run FOUR waves x three tiles, two
    // workgroups per CU - the form that gave the split-bf16 loop 14 % at scale; up to 256 windows - one per CU - twelve waves x
    // one tile): the tiles' MFMAs and gate evaluations are independent instructions back to back, the per-tile arithmetic is
    // the same source in both forms, so a posterior does not depend on the launch size.
    auto run_block_t = [&](int blk, const wblk_t &P, wblk_t &Pnext, const vblk_t &V, vblk_t &Vnext) {
      float *u = ubuf + (blk & 1) * 4 * UPL + kk * UPL + WV_PAD * 4;          // row 0 of this lane's channel-group plane
      const int d = (int)((a.dil4[blk >> 4] >> (4 * (blk & 15))) & 15);
      const float4 *vt = (const float4 *)(vtab + blk * 112) + kk;
      const float4 bres = vt[16], bsk0 = vt[20], bsk1 = vt[24];
      f32x4 uv[WV_MPW], as[WV_MPW], at[WV_MPW];
#pragma unroll
      for (int mi = 0; mi < WV_MPW; ++mi) {
        uv[mi] = (f32x4){x[mi][0] * V.bns.x + V.bnt.x, x[mi][1] * V.bns.y + V.bnt.y, x[mi][2] * V.bns.z + V.bnt.z,
                         x[mi][3] * V.bns.w + V.bnt.w};
        *(f32x4 *)(u + (tl + 16 * mi) * 4) = uv[mi];
      }
      const int nb = blk + 1 < a.NB ? blk + 1 : blk;
      wload(nb, Pnext);  // unconditional (clamped) prefetch, as the row-major loop
#pragma unroll
      for (int mi = 0; mi < WV_MPW; ++mi) {
        as[mi] = f4(V.bsig);
        at[mi] = f4(V.btanh);
        MFMA4(as[mi], P.wg[2][0], uv[mi]);             // tap 2 = this row: runs while the other waves arrive
        MFMA4(at[mi], P.wg[2][1], uv[mi]);
      }
      __syncthreads();  // u complete (all rows, all waves)
      f32x4 t0v[WV_MPW], t1v[WV_MPW];
#pragma unroll
      for (int mi = 0; mi < WV_MPW; ++mi) {            // rows < 0 hit the zero pad (d <= 8)
        t0v[mi] = *(const f32x4 *)(u + (tl + 16 * mi - 2 * d) * 4);
        t1v[mi] = *(const f32x4 *)(u + (tl + 16 * mi - d) * 4);
      }
      vload(nb, Vnext);
      __builtin_amdgcn_sched_barrier(0);  // the tap reads (and the table reads behind them) are in flight before the first wait
      const int has_res = (a.has_res_mask >> blk) & 1;
#pragma unroll
      for (int mi = 0; mi < WV_MPW; ++mi) {
        MFMA4(as[mi], P.wg[0][0], t0v[mi]);
        MFMA4(at[mi], P.wg[0][1], t0v[mi]);
        MFMA4(as[mi], P.wg[1][0], t1v[mi]);
        MFMA4(at[mi], P.wg[1][1], t1v[mi]);
      }
#pragma unroll
      for (int mi = 0; mi < WV_MPW; ++mi) {
        const f32x4 gv = {fast_tanh_w(at[mi][0]) * fast_sigmoid_w(as[mi][0]), fast_tanh_w(at[mi][1]) * fast_sigmoid_w(as[mi][1]),
                          fast_tanh_w(at[mi][2]) * fast_sigmoid_w(as[mi][2]), fast_tanh_w(at[mi][3]) * fast_sigmoid_w(as[mi][3])};
        f32x4 ar = f4(bres), s0 = f4(bsk0), s1 = f4(bsk1);
        // (hand-interleaving the k-steps of the accumulators - dependent MFMAs issue after 40 cycles, independent ones after 32 - was
        //  1 % SLOWER: with three waves per SIMD the other waves fill those 8 cycles, and the compiler's own order keeps fewer values live)
        if (has_res) { MFMA4(ar, P.wrs[0], gv); }
        MFMA4(s0, P.wrs[1], gv);
        MFMA4(s1, P.wrs[2], gv);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (has_res) x[mi][r] = relu1(ar[r]) + x[mi][r];
          skip[mi][0][r] = skip[mi][0][r] + relu1(s0[r]);
          skip[mi][1][r] = skip[mi][1][r] + relu1(s1[r]);
        }
      }
    };
    for (int blk = 0; blk < a.NB; blk += 2) {
      run_block_t(blk, pw[0], pw[1], pv[0], pv[1]);
      if (blk + 1 < a.NB) run_block_t(blk + 1, pw[1], pw[0], pv[1], pv[0]);
    }
    __syncthreads();
  } else if (!SPLIT_BF16) {
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


  } else {
    // ---- split-bf16 block loop (transposed: lane = time column n = lane & 15, rows = channels 4 kk + r).
    //      LDS: u for the delayed taps only, [plane hi | lo][2 buffers][4 channel groups kk][WV_T + WV_PAD rows][4 x bf16]:
    //      16 lanes of one kk touch 128 contiguous bytes and the kk chunks sit 128 bytes apart mod 256, so the 8-byte
    //      accesses are conflict-free at the full ds_read_b64 rate (2 LDS cycles per wave-instruction; right after the
    //      barrier all 12 waves fetch their delayed taps at once and that burst is on every wave's critical path).  One
    //      address add per block: the four reads (rows t - 2d and t - d, hi and lo) are immediate offsets from it.
    // LDS instruction ORDER is part of the design (a wave's LDS operations complete in issue order): BatchNorm vectors, the
    // u write, the gate operands - barrier - the delayed taps, and only then the res | skip operands, which are not needed
    // for another ~600 cycles (requested after the tap MFMAs instead: no gain, measured).  The wait in front of the barrier is counted (only the u write has to be complete), so
    // nobody waits at the barrier for 14 KB of operands per wave to stream out of LDS.
    // 12 / NW row tiles per wave (template parameter): every statement of the block body runs over the wave's tiles, so a
    // wave with several tiles issues independent MFMAs / gate evaluations back to back.
    // The loop is bound by vector-instruction ISSUE (each of the 3 waves of a SIMD issues its ~80 vector instructions per
    // block, and every bf16 MFMA holds the SIMD's vector issue for 8 of its 16 cycles), so the block body carries no
    // instruction it can avoid: ReLU is one v_max (fmaxf costs a canonicalising v_max in front), the exp2 scale factors
    // of the gates sit in the packed weights, the dilation comes out of two SGPR pairs read before the loop.
    constexpr int U_KK_B = (WV_T + WV_PAD) * 8, U_BUF_B = 4 * U_KK_B, U_PLANE_B = 2 * U_BUF_B;   // bytes
    static_assert(2 * U_PLANE_B == 2 * (WV_T + WV_PAD) * WV_C * 4, "u planes must fill exactly the fp32 u buffers' bytes");
    static_assert(U_KK_B % 256 == 128, "the two channel groups of a 32-lane read group must sit 128 bytes apart (mod 256): conflict-free ds_read_b64");
    unsigned char *ldsb = (unsigned char *)lds;
    const unsigned lds0 = (unsigned)(uintptr_t)lds;  // low 32 bits of a flat LDS pointer = the LDS byte address
    for (int i = tid; i < 2 * 2 * 4 * WV_PAD; i += WV_THREADS) {                  // causal zero rows: [plane][buffer][kk][row < PAD]
      const int pl = i / (2 * 4 * WV_PAD), b = (i / (4 * WV_PAD)) & 1, k = (i / WV_PAD) & 3, r = i % WV_PAD;
      *(uint2 *)(ldsb + pl * U_PLANE_B + b * U_BUF_B + k * U_KK_B + r * 8) = make_uint2(0u, 0u);
    }
    // this wave's tiles are consecutive: tile mi starts 16 rows = 128 bytes behind tile mi - 1 in every plane
    const int t0 = wave * WV_MPW * 16;
    const int ub = kk * U_KK_B + (WV_PAD + t0 + j) * 8;                           // this lane's (row t of tile 0, channel group kk), hi plane
    // Block parameters (14 A-operand slots = one 14 KB "page") are identical for all waves: the workgroup
    // fetches page b+2 at the top of block b and parks it in LDS at the end of the block (three buffers).  The barrier of
    // block b+1 publishes it, so in block b+2 every wave may read its operands BEFORE that block's barrier.
    // (Per-wave register prefetch cost 1 us per block: the loads can only be issued once the registers are free, i.e. late;
    // double-buffered pages read after the barrier left ~1000 cycles of operand reads on the critical path.)
    uint4 *pages = (uint4 *)(lds + 2 * (WV_T + WV_PAD) * WV_C);                  // [3][WV_PAGE_U4]
    const uint4 *gpage = a.wpk;
    auto pclamp = [&](int q) { return tid + q * WV_THREADS < WV_PAGE_U4 ? tid + q * WV_THREADS : WV_PAGE_U4 - 1; };
    const int pidx0 = pclamp(0), pidx1 = pclamp(1), pidx2 = pclamp(2), pidx3 = pclamp(3);
#pragma unroll
    for (int q = 0; q < NPL; ++q) {  // pages 0 and 1: on their way since the top of the kernel
      const int i = tid + q * WV_THREADS;
      if (i < WV_PAGE_U4) {
        *(u32x4 *)(pages + i) = pg0[q];
        *(u32x4 *)(pages + WV_PAGE_U4 + i) = pg1[q];
      }
    }
    // The BatchNorm vectors are needed BEFORE a block's barrier (they produce u), i.e. before that block's
    // page is published: all blocks' copies live in their own small table, filled once.
    float4 *bnall = (float4 *)(pages + 3 * WV_PAGE_U4);                           // [NB][2][4] float4 = scale, shift
    if (tid < a.NB * 8) *(f32x4 *)(bnall + tid) = bnv;
    __syncthreads();
    const unsigned long long dil_lo = a.dil4[0], dil_hi = a.dil4[1];             // kernel-argument SGPRs: no load inside the loop
    const short one = (short)(kk == 0 ? 0x3F80 : 0);
    const s16x4 one2 = {one, one, 0, 0};                                          // k-slots 4, 5 of lane group 0 = 1.0: the bias slots (hi, lo)
    int pbuf = 0;  // blk % 3
    for (int blk = 0; blk < a.NB; ++blk) {
      WV_STAMP(0) WV_STAMP(11)
      const int bo = (blk & 1) * U_BUF_B;
      const int d = (int)(((blk < 16 ? dil_lo : dil_hi) >> (4 * (blk & 15))) & 15);
      const int nblk = blk + 2 < a.NB ? blk + 2 : a.NB - 1;                      // unconditional prefetch target
      // (named registers, not an array: the array form stayed in scratch memory and every block waited for its own prefetch)
      const uint4 *gnext = gpage + (size_t)nblk * WV_PAGE_U4;
      uint4 np0 = gnext[pidx0], np1 = gnext[pidx1], np2 = np0, np3 = np0;
      if (NPL > 2) np2 = gnext[pidx2];
      if (NPL > 3) np3 = gnext[pidx3];
      __builtin_amdgcn_sched_barrier(0);  // keep the loads HERE (the scheduler would sink them to their use)
      const uint4 *pg = pages + pbuf * WV_PAGE_U4;
      const bf16x8 *wsl = (const bf16x8 *)pg + lane;                             // slot q: wsl[q * 64]
      WV_STAMP(1)
      const float4 bn_s = bnall[blk * 8 + kk], bn_t = bnall[blk * 8 + 4 + kk];
      // BatchNorm affine (wavenet_model.py:57): a tile's u = the undelayed tap's B operand
      s16x4 u2h[WV_MPW], u2l[WV_MPW];
      const unsigned wa = lds0 + (unsigned)(bo + ub);
#pragma unroll
      for (int mi = 0; mi < WV_MPW; ++mi) {
        const float uv[4] = {x[mi][0] * bn_s.x + bn_t.x, x[mi][1] * bn_s.y + bn_t.y, x[mi][2] * bn_s.z + bn_t.z, x[mi][3] * bn_s.w + bn_t.w};
        split4(uv, u2h[mi], u2l[mi]);
      }
      WV_STAMP(2)
      // hi and lo planes straight from the operand register pairs (immediate offsets: tile, plane)
#define WV_WR(mi_)                                                                                                   \
  if ((mi_) < WV_MPW)                                                                                                \
    asm volatile("ds_write_b64 %0, %1 offset:%3\n\tds_write_b64 %0, %2 offset:%4"                                   \
                 : : "v"(wa), "v"(u2h[(mi_) < WV_MPW ? (mi_) : 0]), "v"(u2l[(mi_) < WV_MPW ? (mi_) : 0]), "n"((mi_) * 128), "n"(U_PLANE_B + (mi_) * 128) : "memory");
      WV_WR(0) WV_WR(1) WV_WR(2)
#undef WV_WR
      // this block's gate operands (the page was published one barrier ago)
      const bf16x8 w0 = wsl[0 * 64], w1 = wsl[1 * 64], w2 = wsl[2 * 64], w3 = wsl[3 * 64];
      const bf16x8 w4 = wsl[4 * 64], w5 = wsl[5 * 64], w6 = wsl[6 * 64], w7 = wsl[7 * 64];
      WV_STAMP(3)
      // k-step 0 = tap 2, operands in registers - these MFMAs run while the other waves arrive.  The 8 k-slots of a lane
      // group hold TWO 4-channel groups: (w_hi | w_hi) x (u_hi | u_lo) is hi*hi + hi*lo in one MFMA, (w_lo | bias) x (u_hi | 1, 1)
      // the lo*hi product plus the bias (hi and lo halves in k-slots 4, 5 of lane group 0): api.hip, load_wavenet
      f32x4 as[WV_MPW], at[WV_MPW];
#pragma unroll
      for (int mi = 0; mi < WV_MPW; ++mi) {
        as[mi] = (f32x4){0.f, 0.f, 0.f, 0.f};
        at[mi] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const bf16x8 xua = cat8(u2h[mi], u2l[mi]), xub = cat8(u2h[mi], one2);
        MFMA_BF(as[mi], w0, xua); MFMA_BF(at[mi], w2, xua);
        MFMA_BF(as[mi], w1, xub); MFMA_BF(at[mi], w3, xub);
      }
      WV_STAMP(4)
      // u complete (all rows, all waves).  At most the 8 operand reads above are younger than the u writes, so "at most 8
      // LDS operations outstanding" means the writes have landed; the operands keep streaming across the barrier.
#if WV_PROBE_NOBAR  // ablation (wrong results, right instruction stream): what the twelve-wave barrier itself costs
      asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
#else
      asm volatile("s_waitcnt lgkmcnt(8)\n\ts_barrier" ::: "memory");
#endif
      WV_STAMP(5)
      // k-step 1 = (tap 0 | tap 1) = rows t - 2d and t - d (rows < 0 hit the zero pad, d <= 8)
      s16x4 u0h[WV_MPW], u1h[WV_MPW], u0l[WV_MPW], u1l[WV_MPW];
      {
        const unsigned ra = wa - 16u * (unsigned)d;
#define WV_RD4(mi_, d_)                                                                                                     \
  if ((mi_) < WV_MPW)                                                                                                       \
    asm volatile("ds_read_b64 %0, %4 offset:%5\n\tds_read_b64 %1, %4 offset:%6\n\tds_read_b64 %2, %4 offset:%7\n\tds_read_b64 %3, %4 offset:%8" \
                 : "=&v"(u0h[(mi_) < WV_MPW ? (mi_) : 0]), "=&v"(u1h[(mi_) < WV_MPW ? (mi_) : 0]),                           \
                   "=&v"(u0l[(mi_) < WV_MPW ? (mi_) : 0]), "=&v"(u1l[(mi_) < WV_MPW ? (mi_) : 0])                            \
                 : "v"(ra), "n"((mi_) * 128), "n"((mi_) * 128 + 8 * (d_)), "n"(U_PLANE_B + (mi_) * 128),                     \
                   "n"(U_PLANE_B + (mi_) * 128 + 8 * (d_))                                                                  \
                 : "memory");
#define WV_RD(d_) WV_RD4(0, d_) WV_RD4(1, d_) WV_RD4(2, d_)
        if (d == 1) { WV_RD(1) }
        else if (d == 2) { WV_RD(2) }
        else if (d == 4) { WV_RD(4) }
        else if (d == 8) { WV_RD(8) }
        else {  // any other dilation: plain loads
#pragma unroll
          for (int mi = 0; mi < WV_MPW; ++mi) {
            const unsigned char *base = ldsb + bo + ub + mi * 128;
            u0h[mi] = *(const s16x4 *)(base - 16 * d); u1h[mi] = *(const s16x4 *)(base - 8 * d);
            u0l[mi] = *(const s16x4 *)(base + U_PLANE_B - 16 * d); u1l[mi] = *(const s16x4 *)(base + U_PLANE_B - 8 * d);
          }
        }
#undef WV_RD
#undef WV_RD4
      }
      // res | skip operands: behind the taps in the LDS queue, in front of their use by a whole gate evaluation
      const bf16x8 r0 = wsl[8 * 64], r1 = wsl[9 * 64], r2 = wsl[10 * 64], r3 = wsl[11 * 64], r4 = wsl[12 * 64], r5 = wsl[13 * 64];
      // the taps (6 younger reads may be in flight)
#pragma unroll
      for (int mi = 0; mi < WV_MPW; ++mi)
        asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(u0h[mi]), "+v"(u1h[mi]), "+v"(u0l[mi]), "+v"(u1l[mi]) : : "memory");
      WV_STAMP(6)
#if WV_PROBE_ACC2  // probe (round 6): the delayed taps into accumulators of their own - two MFMA chains of depth 2 and 3 per gate
                   // instead of one of depth 5, joined by one v_add per row (other bits, still <= 2e-5)
#pragma unroll
      for (int mi = 0; mi < WV_MPW; ++mi) {
        const bf16x8 xdh = cat8(u0h[mi], u1h[mi]), xdl = cat8(u0l[mi], u1l[mi]);
        f32x4 as2 = {0.f, 0.f, 0.f, 0.f}, at2 = {0.f, 0.f, 0.f, 0.f};
        MFMA_BF(as2, w4, xdh); MFMA_BF(at2, w6, xdh);
        MFMA_BF(as2, w5, xdh); MFMA_BF(at2, w7, xdh);
        MFMA_BF(as2, w4, xdl); MFMA_BF(at2, w6, xdl);
        as[mi] += as2; at[mi] += at2;
      }
#else
#pragma unroll
      for (int mi = 0; mi < WV_MPW; ++mi) {
        const bf16x8 xdh = cat8(u0h[mi], u1h[mi]), xdl = cat8(u0l[mi], u1l[mi]);
        MFMA_BF(as[mi], w4, xdh); MFMA_BF(at[mi], w6, xdh);
        MFMA_BF(as[mi], w5, xdh); MFMA_BF(at[mi], w7, xdh);
        MFMA_BF(as[mi], w4, xdl); MFMA_BF(at[mi], w6, xdl);
      }
#endif
      WV_STAMP(7)
      // gate: tanh(t) * sigmoid(s); biases AND the exp2 scale factors (-log2 e, 2 log2 e) are inside the accumulators
      f32x4 ar[WV_MPW], s0[WV_MPW], s1[WV_MPW];
#pragma unroll
      for (int mi = 0; mi < WV_MPW; ++mi) {
        float gv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#if WV_PROBE_RCP1  // probe (round 6): tanh(t) sigmoid(s) = (et - 1) / ((1 + et)(1 + es)) - ONE reciprocal; et clamped so that inf / inf cannot occur
          const float et = __builtin_amdgcn_exp2f(__builtin_fminf(at[mi][r], 60.0f));
          const float es = __builtin_amdgcn_exp2f(as[mi][r]);
          gv[r] = (et - 1.0f) * __builtin_amdgcn_rcpf((1.0f + et) * (1.0f + es));
#else
          const float et = __builtin_amdgcn_exp2f(at[mi][r]);    // exp(2 t): inf -> tanh 1, 0 -> -1
          const float es = __builtin_amdgcn_exp2f(as[mi][r]);    // exp(-s)
          gv[r] = (1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + et)) * __builtin_amdgcn_rcpf(1.0f + es);
#endif
        }
        WV_STAMP(8)
        s16x4 g_h, g_l;
        split4(gv, g_h, g_l);  // the gate product is the res / skip conv's B operand as it stands
        ar[mi] = (f32x4){0.f, 0.f, 0.f, 0.f}; s0[mi] = ar[mi]; s1[mi] = ar[mi];
        const bf16x8 ga8 = cat8(g_h, g_l), gb8 = cat8(g_h, one2);
        MFMA_BF(ar[mi], r0, ga8); MFMA_BF(s0[mi], r2, ga8); MFMA_BF(s1[mi], r4, ga8);
        MFMA_BF(ar[mi], r1, gb8); MFMA_BF(s0[mi], r3, gb8); MFMA_BF(s1[mi], r5, gb8);
      }
      WV_STAMP(9)
      // residual / skip update; biases ride in the MFMA, and a block without a residual conv has zero
      // res weights and bias (relu(0) = 0), so no special case
#pragma unroll
      for (int mi = 0; mi < WV_MPW; ++mi)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          x[mi][r] = relu1(ar[mi][r]) + x[mi][r];
          skip[mi][0][r] = skip[mi][0][r] + relu1(s0[mi][r]);
          skip[mi][1][r] = skip[mi][1][r] + relu1(s1[mi][r]);
        }
      WV_STAMP(10)
      // park page blk+2 (loaded a whole block ago) in the buffer that held page blk-1: every wave is past its
      // reads of that one (they precede the barrier of block blk, which everyone here has passed)
      const int nbuf = pbuf == 0 ? 2 : pbuf - 1;  // (blk + 2) % 3
      uint4 *pn = pages + nbuf * WV_PAGE_U4;
      pn[pidx0] = np0;
      if (tid + WV_THREADS < WV_PAGE_U4) pn[pidx1] = np1;
      if (NPL > 2 && tid + 2 * WV_THREADS < WV_PAGE_U4) pn[pidx2] = np2;
      if (NPL > 3 && tid + 3 * WV_THREADS < WV_PAGE_U4) pn[pidx3] = np3;
      pbuf = pbuf == 2 ? 0 : pbuf + 1;
    }
    __syncthreads();
  }
  }
  // ---- encoder output (optional) + detect head
  if (a.enc) {
    float *e = a.enc + (size_t)w * T * WV_S;
#pragma unroll
    for (int mi = 0; mi < WV_MPW; ++mi) {
      if (TRANSPOSED) {  // transposed state: lane = time column, four consecutive channels per register quad
        const int t = (wave * WV_MPW + mi) * 16 + j;
        if (t < T) {
          *(float4 *)(e + (size_t)t * WV_S + kk * 4) = make_float4(skip[mi][0][0], skip[mi][0][1], skip[mi][0][2], skip[mi][0][3]);
          *(float4 *)(e + (size_t)t * WV_S + 16 + kk * 4) = make_float4(skip[mi][1][0], skip[mi][1][1], skip[mi][1][2], skip[mi][1][3]);
        }
        continue;
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int t = (wave * WV_MPW + mi) * 16 + kk * 4 + r;
        if (t < T) {
          e[(size_t)t * WV_S + j] = skip[mi][0][r];
          e[(size_t)t * WV_S + 16 + j] = skip[mi][1][r];
        }
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
    if (TRANSPOSED) {
      *(float4 *)(ht + j * WV_S + kk * 4) = make_float4(fmaxf(skip[mi][0][0], 0.f), fmaxf(skip[mi][0][1], 0.f),
                                                        fmaxf(skip[mi][0][2], 0.f), fmaxf(skip[mi][0][3], 0.f));
      *(float4 *)(ht + j * WV_S + 16 + kk * 4) = make_float4(fmaxf(skip[mi][1][0], 0.f), fmaxf(skip[mi][1][1], 0.f),
                                                             fmaxf(skip[mi][1][2], 0.f), fmaxf(skip[mi][1][3], 0.f));
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        ht[(kk * 4 + r) * WV_S + j] = fmaxf(skip[mi][0][r], 0.f);
        ht[(kk * 4 + r) * WV_S + 16 + j] = fmaxf(skip[mi][1][r], 0.f);
      }
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
    if (a.tag.slots) {  // a streaming tick: the posterior as ONE 8-byte {value, tick number} store the host polls (common.h)
      if (tid == a.tag.pidx) {
        const unsigned long long word = (unsigned long long)__float_as_uint(e / sum) | ((unsigned long long)a.tag.seq << 32);
        __hip_atomic_store(a.tag.slots + w, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    } else if (tid < a.NOUT) {
      a.out[(size_t)w * a.NOUT + tid] = e / sum;
    }
  }
}

size_t ww_wave_workspace(const ww_model *, int) { return 256; }

int ww_k_wave_forward(ww_ctx *ctx, const ww_model *m, const float *d_mel, int64_t mel_rows, const int64_t *d_win_row,
                      const int32_t *d_win_valid, int64_t row0, int hop, int valid_const, int nw, void *, size_t, float *d_out,
                      float *d_enc, const ww_tick_tag *tag) {
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
  if (tag) a.tag = *tag;
  ww_launch_scope scope(ctx, m->precision == WW_PRECISION_BF16X3 ? "wavenet_kernel<bf16x3>" : "wavenet_kernel");
  a.wpk = (const uint4 *)v.wpk;
#if WV_STAMPS
  // development build: WWHIP_WV_STAMPS=1 prints, per block, the mean cycles since the block's top at each stamp (over all windows
  // and waves), the mean wait at the barrier (stamp 5 - stamp 4) and the spread of the twelve waves' arrivals at it
  // (max - min of stamp 4 within a window): tools/wv_stamps.py turns the lines into profiles/r06/wavenet_bf16x3_phase_stamps.txt
  struct stamp_dump {
    ww_ctx *ctx; long long *d; int n;
    ~stamp_dump() {
      if (!d) return;
      const size_t per_wave = (size_t)WV_STAMP_NB * 12;
      std::vector<long long> h((size_t)n * 12 * per_wave);
      hipStreamSynchronize(ctx->stream);
      hipMemcpy(h.data(), d, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
      hipFree(d);
      for (int b = 0; b < WV_STAMP_NB; ++b) {
        double sum[12] = {0}, wait = 0, spread = 0;
        long long cnt = 0, wins = 0;
        for (int w = 0; w < n; ++w) {
          long long lo = 0, hi = 0;
          bool any = false;
          for (int v = 0; v < 12; ++v) {
            const long long *t = &h[((size_t)w * 12 + v) * per_wave + (size_t)b * 12];
            if (!t[0] || !t[10]) continue;
            for (int i = 0; i < 12; ++i) sum[i] += t[i] ? (double)(t[i] - t[0]) : 0.0;
            wait += (double)(t[5] - t[4]);
            lo = any ? (t[4] < lo ? t[4] : lo) : t[4];
            hi = any ? (t[4] > hi ? t[4] : hi) : t[4];
            any = true;
            ++cnt;
          }
          if (any) { spread += (double)(hi - lo); ++wins; }
        }
        fprintf(stderr, "wavenet stamps: block %d waves %lld cycles:", b, cnt);
        for (int i = 0; i < 12; ++i) fprintf(stderr, " %.0f", sum[i] / (cnt ? cnt : 1));
        fprintf(stderr, " barrier_wait %.0f arrival_spread %.0f\n", wait / (cnt ? cnt : 1), spread / (wins ? wins : 1));
      }
    }
  } dump{ctx, nullptr, nw};
  if (getenv("WWHIP_WV_STAMPS") && m->precision == WW_PRECISION_BF16X3 && nw <= WV_BF16_WIDE_FROM) {
    const size_t bytes = (size_t)nw * 12 * WV_STAMP_NB * 12 * sizeof(long long);
    WW_HIP(ctx, hipMalloc((void **)&a.stamps, bytes));
    WW_HIP(ctx, hipMemsetAsync(a.stamps, 0, bytes, ctx->stream));
    dump.d = a.stamps;
  }
#endif
  // split-bf16: twelve waves x one 16-row tile while every window has a CU of its own (the 24 blocks of a window are a serial
  // chain: more waves per window cover its latencies best, 39.1 vs 45.7 us per 256 windows); from the 257th window of a launch
  // on FOUR waves x three tiles (173 registers, 74 KB: TWO workgroups per CU, each wave issuing three tiles' independent MFMAs
  // and gate evaluations back to back): 1,656 vs 1,957 us per 16,384 windows, 69.8 vs 74.5 at 512.  Same arithmetic per
  // tile in both forms: a posterior does not depend on the launch size (tests/test_gpu_parity.py).  (Six waves x two tiles -
  // also two workgroups per CU - lose at every size: 2,393 us.)
  if (m->precision == WW_PRECISION_BF16X3 && WV_BF16_NW == 12 && nw > WV_BF16_WIDE_FROM)
    hipLaunchKernelGGL((wavenet_kernel<false, true, 4>), dim3(nw), dim3(4 * 64), 0, ctx->stream, a);
  else if (m->precision == WW_PRECISION_BF16X3)
    hipLaunchKernelGGL((wavenet_kernel<false, true, WV_BF16_NW>), dim3(nw), dim3(WV_BF16_NW * 64), 0, ctx->stream, a);
  else if (m->opt_wave_rowmajor)
    hipLaunchKernelGGL((wavenet_kernel<false, false, 12>), dim3(nw), dim3(12 * 64), 0, ctx->stream, a);
  else if (nw > WV_F32_WIDE_FROM)  // (round 5) fp32: the same two forms as the split-bf16 loop, the same bits in both
    hipLaunchKernelGGL((wavenet_kernel<false, false, 4, true>), dim3(nw), dim3(4 * 64), 0, ctx->stream, a);
  else
    hipLaunchKernelGGL((wavenet_kernel<false, false, 12, true>), dim3(nw), dim3(12 * 64), 0, ctx->stream, a);
  WW_HIP(ctx, hipGetLastError());
  return WW_OK;
}

// Which banks take the one-launch tick: those whose ticks stay within the twelve-wave form's range (2 S windows <= 256: one
// workgroup per CU).  Larger banks keep the front-end kernel + the model kernel in its four-wave x three-tile form, which is worth
// more than the launch it costs (tools/stream_forms.py, split-bf16 p50: 128 streams 49.3 vs 54.5 us, 192: 84.9 vs 82.6, 512: 158.7 vs
// 149.2, 1,024: 306.2 vs 273.1; fp32 alike)
bool ww_wave_tick_capable(const ww_model *m, int S) {
  const int wide_from = m->precision == WW_PRECISION_BF16X3 ? WV_BF16_WIDE_FROM : WV_F32_WIDE_FROM;
  return m->kind == WW_KIND_WAVENET && m->filt.n_mel == 40 && m->wave.n_mel == 40 && m->wave.T + 10 <= WV_T && 2 * S <= wide_from;
}

// ONE launch per tick (wavenet_kernel<..., TICK>): 2 S workgroups of twelve waves, the posteriors as tags only
int ww_k_wave_tick(ww_ctx *ctx, const ww_model *m, const ww_tick_fe &fe, int precise, const ww_tick_tag &tag) {
  const ww_wave_dev &v = m->wave;
  const ww_filter_dev &f = m->filt;
  if (f.n_mel != 40 || v.n_mel != 40 || fe.hop != 160 || v.T + 10 > WV_T)
    return ww_fail(ctx, WW_EINVAL, "one-launch streaming tick: 40 mel bands, hop 160 and a window of at most %d rows only", WV_T - 10);
  if (!tag.slots || fe.S <= 0) return ww_fail(ctx, WW_EINVAL, "one-launch streaming tick: no tag slots / no streams");
  wave_args a = {};
  a.mel = fe.hist;
  a.wa = {nullptr, nullptr, 0, 0, 0, (int64_t)fe.S * fe.HR};
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
  a.wpk = (const uint4 *)v.wpk;
  a.tag = tag;
  a.fe = fe;
  a.mel_start = f.start; a.mel_wpad = f.wpad; a.mel_bias = f.bias;
  a.floor_v = f.floor_v; a.log_off = f.log_off; a.scale = f.scale;
  a.hann = f.hann; a.tw256 = f.tw256; a.tw512 = f.tw512;
  const dim3 grid(2 * fe.S), block(12 * 64);
  ww_launch_scope scope(ctx, m->precision == WW_PRECISION_BF16X3 ? "wavenet_kernel<bf16x3,tick>" : "wavenet_kernel<tick>");
  if (m->precision == WW_PRECISION_BF16X3) {
    if (precise) hipLaunchKernelGGL((wavenet_kernel<false, true, 12, false, 2>), grid, block, 0, ctx->stream, a);
    else hipLaunchKernelGGL((wavenet_kernel<false, true, 12, false, 1>), grid, block, 0, ctx->stream, a);
  } else if (m->opt_wave_rowmajor) {
    if (precise) hipLaunchKernelGGL((wavenet_kernel<false, false, 12, false, 2>), grid, block, 0, ctx->stream, a);
    else hipLaunchKernelGGL((wavenet_kernel<false, false, 12, false, 1>), grid, block, 0, ctx->stream, a);
  } else {
    if (precise) hipLaunchKernelGGL((wavenet_kernel<false, false, 12, true, 2>), grid, block, 0, ctx->stream, a);
    else hipLaunchKernelGGL((wavenet_kernel<false, false, 12, true, 1>), grid, block, 0, ctx->stream, a);
  }
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
  hipLaunchKernelGGL((wavenet_kernel<true, false, 12>), dim3(nw), dim3(12 * 64), 0, ctx->stream, a);
  WW_HIP(ctx, hipGetLastError());
  return WW_OK;
}
