/*
 * wwhip.h - C ABI of libwwhip.so, the MI355X (gfx950) wake-word inference hot path.
 *
 * The reference (MerlinPCarson/WakeWord-Detection) has no FFI of its own: its hot path is
 * NumPy + the TensorFlow-Lite interpreter behind Python duck-typed classes.  Each entry
 * point below names the reference interface it replaces (paths relative to the reference
 * repository root).  The Python host classes in wakeword-detection_amd/wwhip/ bind these
 * with ctypes (INTEGRATION.md shows the stub).
 *
 * Conventions
 *   - every function returns 0 on success or a negative WW_E* code and never throws;
 *     ww_last_error(ctx) holds a human-readable message for the last failure on that ctx;
 *   - one ww_ctx per host thread; a ctx owns one HIP stream (or borrows the caller's) and a
 *     growable device workspace; calls on one ctx are serialised by the caller;
 *   - "host" entry points take ordinary host pointers, copy through pinned staging and
 *     return after the result is in the caller's buffer (PCIe inclusive);
 *   - "_dev" entry points take device pointers (e.g. torch tensor .data_ptr()), enqueue on
 *     the ctx stream and return without synchronising;
 *   - the caller owns every buffer it passes; nothing is retained after return except by
 *     ww_model_load (copies the blob) and the ww_streams object (owns its rings).
 */
#ifndef WWHIP_H
#define WWHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WW_OK 0
#define WW_EINVAL (-1)    /* bad argument (maps to ValueError) */
#define WW_EBLOB (-2)     /* malformed weight blob (ValueError) */
#define WW_EHIP (-3)      /* HIP runtime failure (RuntimeError) */
#define WW_ENOMEM (-4)    /* allocation failure (MemoryError) */
#define WW_ESTATE (-5)    /* object used in the wrong state (RuntimeError) */
#define WW_ENODEVICE (-6) /* no usable gfx950 device (RuntimeError) */
#define WW_EINTERNAL (-7) /* an exception stopped at the C boundary (RuntimeError); ww_last_error has its text */

#define WW_KIND_CRNN 1
#define WW_KIND_WAVENET 2

#define WW_FFT_WINDOW 512 /* (257 - 1) * 2, reference spokestack/wakeword/tflite.py:67 */
#define WW_FFT_BINS 257
#define WW_CHUNK 320 /* 20 ms @ 16 kHz, reference spokestack/io/pyaudio.py:24 */

typedef struct ww_ctx ww_ctx;
typedef struct ww_model ww_model;
typedef struct ww_streams ww_streams;

typedef struct ww_model_info {
  int32_t kind;      /* WW_KIND_* */
  int32_t window;    /* mel frames per inference: 151 (CRNN) / 182 (Wavenet) */
  int32_t n_mel;     /* 40 */
  int32_t n_bins;    /* 257 */
  int32_t n_out;     /* width of the detect output row: 1 (sigmoid) or 2 (softmax) */
  int32_t enc_rows;  /* encoder output rows: 1 (CRNN) / 182 (Wavenet) */
  int32_t enc_width; /* encoder output width: 64 (CRNN) / 32 (Wavenet) */
  int32_t reserved;
} ww_model_info;

/* Front-end parameters.  Reference: spokestack/wakeword/tflite.py:33-43,150-158 and
 * utils/tf_lite/filter.py:9-19,42-44. */
typedef struct ww_frontend_params {
  float pcm_divisor;  /* 32767 (streaming plugin, tflite.py:150) or 32768 (librosa floats) */
  int32_t clip;       /* 1: clip to [-1, 1] after the division (tflite.py:151) */
  float pre_emphasis; /* alpha of x[n] -= alpha * x[n-1]; reference default 0.0 */
  int32_t hop;        /* samples between frames: 160 (fft_hop_length 10 ms @ 16 kHz) */
  int32_t precise;    /* 1: Hann product + FFT in fp64 like the reference (tflite.py:175);
                         0: fp32 butterflies (faster, ~1e-6 relative on the magnitudes) */
} ww_frontend_params;

/* ---- context ------------------------------------------------------------------------- */
/* external_stream: NULL -> the ctx creates its own non-blocking stream; otherwise a
 * hipStream_t owned by the caller (e.g. torch.cuda.current_stream().cuda_stream). */
int ww_ctx_create(int device, void *external_stream, ww_ctx **out);
int ww_ctx_destroy(ww_ctx *ctx);
int ww_ctx_synchronize(ww_ctx *ctx);
void *ww_ctx_stream(ww_ctx *ctx);
/* ctx may be NULL: the error of the calling THREAD's last failed ww_ctx_create (thread-local text). */
const char *ww_last_error(const ww_ctx *ctx);
/* "wwhip <major>.<minor> (...; ABI <n>: ...)".  The ABI number changes whenever an existing entry point changes its
 * signature (ABI 4, round 4: ww_stream_create gained `flags` in round 3 - a host built against an older header must be
 * rebuilt); wwhip/_lib.py parses the number at load time and refuses a library of another ABI. */
#define WW_ABI 4
const char *ww_version(void);
/* Which HIP runtime the library actually runs on.  libwwhip.so links libamdhip64 by soname; a host program that has
 * already loaded another copy (PyTorch-ROCm wheels bundle their own) decides which one that is.  built_hip_version =
 * HIP_VERSION of the headers at compile time, runtime_version / driver_version = hipRuntimeGetVersion /
 * hipDriverGetVersion of the loaded runtime (0 if unavailable), all as major * 10,000,000 + minor * 100,000 + patch.
 * The Python binding warns when the major versions differ (wwhip/_lib.py). */
int ww_runtime_info(int32_t *built_hip_version, int32_t *runtime_version, int32_t *driver_version);

/* Host-side staging for the sharded evaluators (wwhip/evaluate.py; replaces the per-sample ring writes and pydub joins of
 * utils/evaluate_models.py:45-61,150-160 on the way to ONE upload): writes dst[0 .. total) once - run j copies count[j]
 * samples from src[j] to dst + dst_off[j] (runs disjoint, dst_off ascending), everything between the runs is zeroed (the
 * 0.5 s paddings and 100 ms gaps of the reference flow) - with `threads` host threads, each owning a contiguous slice of
 * dst.  No GPU call; dst is normally page-locked.  WW_EINVAL on overlapping or descending runs.  Only dst[lo .. hi) is
 * written (0 <= lo <= hi <= total): a caller stages a large buffer in a few slices and starts each slice's upload while the
 * next one is being written. */
int ww_host_stage_i16(int16_t *dst, int64_t total, int64_t n_runs, const int64_t *dst_off, const int16_t *const *src,
                      const int64_t *count, int64_t lo, int64_t hi, int32_t threads);

/* The same staging on a thread of the library's own, followed by the upload: an uploader owns `slots` page-locked buffers
 * (grown on demand), `copy_threads` copy threads and a copy stream on the context's device.  One submit = one chunk of a
 * rank's share of a test split: dst[0 .. total) assembled from the runs as ww_host_stage_i16 does, then copied to d_pcm, and
 * n_meta int64 words (the chunk's sample / frame offset tables) copied to d_meta.  The call copies the run arrays and `meta`
 * and returns at once with a ticket (1, 2, ...; chunks are processed in ticket order); the CLIP samples src[j] point to must
 * stay where they are until ww_uploader_wait(ticket) has returned.  ww_uploader_poll: 1 once the ticket's copies are enqueued,
 * 0 before.  ww_uploader_wait blocks the calling host thread until they are, then makes the context's stream wait for them
 * on the device (no host wait for the transfer itself) - the caller launches its kernels over d_pcm right behind it - and
 * reports what went wrong with the chunk if anything did (WW_EINVAL: overlapping runs; WW_ENOMEM; WW_EHIP).  Each ticket is
 * waited for at most once.  The reference has no counterpart: its evaluator feeds one wav at a time, 20 ms per step
 * (utils/evaluate_models.py:52-88); this is the transport that keeps two hours of audio per GPU moving while the interpreter
 * plans the next chunk.  One host thread drives an uploader; ww_uploader_destroy finishes what was submitted. */
typedef struct ww_uploader ww_uploader;
int ww_uploader_create(ww_ctx *ctx, int32_t slots, int32_t copy_threads, ww_uploader **out);
int ww_uploader_destroy(ww_uploader *up);
int ww_uploader_submit(ww_uploader *up, int64_t total, int64_t n_runs, const int64_t *dst_off, const int16_t *const *src,
                       const int64_t *count, int16_t *d_pcm, int64_t n_meta, const int64_t *meta, int64_t *d_meta, int64_t *ticket);
int ww_uploader_poll(ww_uploader *up, int64_t ticket);
int ww_uploader_wait(ww_uploader *up, int64_t ticket, ww_ctx *ctx);

/* Per-kernel timing with HIP events on the ctx stream (bench.py roofline leg).  While
 * enabled every kernel launch is bracketed by two events; ww_profile_read synchronises and
 * writes a JSON object {"kernel": {"calls": n, "total_ms": t}, ...}. */
int ww_profile_enable(ww_ctx *ctx, int on);
int ww_profile_read(ww_ctx *ctx, char *json, size_t cap);
/* Event pair on the ctx stream around an arbitrary region. */
int ww_timer_start(ww_ctx *ctx);
int ww_timer_stop(ww_ctx *ctx, float *elapsed_ms); /* synchronises on the stop event */

/* ---- model --------------------------------------------------------------------------- */
/* Replaces TFLiteModel.__init__ for the filter/encode/detect triple
 * (spokestack/models/tensorflow.py:24-31).  `blob` is the packed weight image produced by
 * wwhip.weights.pack_blob from the reference's .tflite files:
 *   u32 magic 'WWHB' | u32 version 1 | u32 kind | u32 n_sections
 *   n_sections x { char name[24]; u32 offset_bytes; u32 count }   then 16-byte aligned payload
 * Weights are uploaded once and stay resident in HBM. */
int ww_model_load(ww_ctx *ctx, const void *blob, size_t len, ww_model **out);
int ww_model_free(ww_model *model);
int ww_model_get_info(const ww_model *model, ww_model_info *out);

/* Arithmetic of the model contractions.  The reference runs fp32 TFLite kernels; SURVEY 8(d) cfg 3 asks
 * for the Wavenet convs on bf16 MFMA with fp32 accumulate next to an fp32 parity mode.
 *   WW_PRECISION_FP32   (default) v_mfma_f32_16x16x4_f32 everywhere: bit for bit a k-ordered fp32 fmaf chain.
 *   WW_PRECISION_BF16X3 Wavenet's 24 gated blocks as split-bf16: x = hi + lo (two bf16), a*b =
 *                       ah*bh + ah*bl + al*bh on v_mfma_f32_16x16x32_bf16, fp32 accumulate; 16 mantissa
 *                       bits per operand - posteriors within 4e-6 of fp32 (tolerance 1e-4).  Plain
 *                       single-pass bf16 misses the tolerance (6e-4 .. 1e-3) and is not offered.
 *                       CRNN models: the conv and the layer-1 input projection (88 % of the FLOPs) take the same
 *                       split form (crnn_fused_bf16_kernel); the recurrences and the head stay fp32; posteriors
 *                       within 1e-5 of fp32.  (CRNNs of another conv geometry keep fp32.)
 * (Value 2 was round 1's experimental BF16X6 projection mode; it was not faster than fp32 and is retired: WW_EINVAL.) */
#define WW_PRECISION_FP32 0
#define WW_PRECISION_BF16X3 1
int ww_model_set_precision(ww_model *model, int precision);

/* Dispatch options of one model object (they select between kernels that compute the same windows; results agree to
 * <= 2e-6, tests/test_gpu_parity.py).  These replace round 2's process-wide WWHIP_CRNN_* environment variables.
 *   WW_OPT_CRNN_SPLIT_AT   explicit-window launches above this many windows run crnn_fused_kernel<front> + gru_tail_kernel
 *                          instead of one crnn_fused_kernel (default 1024; 0 = always one fused kernel)
 *   WW_OPT_CRNN_SLIDE_MIN  regular sliding windows (ww_slide_forward, ww_forward_segments_dev) take the once-per-sequence
 *                          form crnn_rows_kernel + gru_tail_kernel from this many windows on (default 64; 0 = never)
 *   WW_OPT_CRNN_TAIL_MFMA  the recurrences of those two forms: sixteen windows per workgroup with the recurrent products and
 *                          the layer-2 projection on v_mfma_f32_16x16x4_f32 (gru_tail16_kernel), or one window per workgroup
 *                          on the vector ALU (gru_tail_kernel).  1 (default) = the matrix form from 9,216 windows per launch on
 *                          (below that its 38-step serial chain on few workgroups loses); 2 = always; 0 = never
 *   WW_OPT_WAVENET_ROWMAJOR 0 (default) = the fp32 Wavenet block loop in transposed form (channels x time: BatchNorm output and
 *                          gate product feed the next MFMA straight from registers); 1 = rounds 1-2's row-major loop (both
 *                          through LDS).  Same products, another summation order of the three taps and the bias. */
#define WW_OPT_CRNN_SPLIT_AT 1
#define WW_OPT_CRNN_SLIDE_MIN 2
#define WW_OPT_CRNN_TAIL_MFMA 3
#define WW_OPT_WAVENET_ROWMAJOR 4
int ww_model_set_option(ww_model *model, int key, int64_t value);

/* ---- front end: PCM -> log-mel ---------------------------------------------------------
 * Replaces the per-sample RingBuffer loop + np.fft.rfft + filter.tflite invoke of
 * Filter.filter_frame / WakewordTrigger._sample/_analyze/_filter
 * (utils/tf_lite/filter.py:38-75, spokestack/wakeword/tflite.py:148-191).
 *
 * Utterance u occupies samples [sample_offs[u], sample_offs[u+1]) of `pcm` and yields
 * nf(u) = max(0, (len - 512) / hop + 1) frames; frame j covers samples [hop*j, hop*j+512).
 * Row frame_offs[u] + j of `mel` receives its 40 log-mel values.  frame_offs (n_utt+1
 * entries) is written by the host variants and read by the device variants. */
int64_t ww_num_frames(int64_t n_samples, int32_t hop);

int ww_logmel(ww_ctx *ctx, const ww_model *model, const int16_t *pcm, const int64_t *sample_offs, int32_t n_utt,
              const ww_frontend_params *fp, float *mel, int64_t *frame_offs);
/* Same, for float samples already scaled to [-1, 1] (the offline path hands librosa floats
 * to Filter.filter_frame: utils/evaluate_models.py:46,64).  pcm_divisor / clip are ignored. */
int ww_logmel_f32(ww_ctx *ctx, const ww_model *model, const float *samples, const int64_t *sample_offs,
                  int32_t n_utt, const ww_frontend_params *fp, float *mel, int64_t *frame_offs);
/* STFT magnitudes only: frames [n][512] fp32 -> mag [n][257] fp32
 * (WakewordTrigger._analyze, spokestack/wakeword/tflite.py:174-176). */
int ww_stft_mag(ww_ctx *ctx, const ww_model *model, const float *frames, int64_t n, int32_t precise, float *mag);

/* filter.tflite on its own: mag [n][257] fp32 -> log-mel [n][40]  (TFLiteModel("filter.tflite")(frame),
 * spokestack/wakeword/tflite.py:183-184; utils/tf_lite/filter.py:72-73). */
int ww_filter_apply(ww_ctx *ctx, const ww_model *model, const float *mag, int64_t n, float *mel);

int ww_logmel_dev(ww_ctx *ctx, const ww_model *model, const int16_t *d_pcm, const int64_t *d_sample_offs,
                  const int64_t *d_frame_offs, int32_t n_utt, int64_t total_frames, int64_t max_frames_per_utt,
                  const ww_frontend_params *fp, float *d_mel);

/* ---- encode + detect -------------------------------------------------------------------
 * Replaces encode_model(x) followed by detect_model(x) (two TFLiteModel.__call__,
 * spokestack/wakeword/tflite.py:193-231; utils/evaluate_models.py:76-86;
 * utils/evaluate_tf_lite_opts.py:49-69).  `windows` is [B][window][n_mel] fp32, rows in
 * time order exactly as RingBuffer.read_all returns them (the CRNN transpose to
 * [1,40,151,1] of tflite.py:199-203 is folded into the kernel's indexing).
 * `out` receives [B][n_out]: the detect graph's output row. */
int ww_forward(ww_ctx *ctx, const ww_model *model, const float *windows, int32_t n_windows, float *out);
/* Optional: also return the encoder output ([B][enc_rows][enc_width]); enc may be NULL. */
int ww_forward_enc(ww_ctx *ctx, const ww_model *model, const float *windows, int32_t n_windows, float *out,
                   float *enc);

/* detect.tflite on its own: encoder outputs [n][enc_rows][enc_width] -> detect rows [n][n_out]
 * (TFLiteModel("detect.tflite")(x), spokestack/wakeword/tflite.py:228-231). */
int ww_detect(ww_ctx *ctx, const ww_model *model, const float *enc, int32_t n, float *out);

/* Sliding evaluation of one mel sequence (utils/evaluate_models.py:66-88): window i covers
 * rows [i*hop, i*hop + window); n_windows = (rows - window) / hop + 1 (0 if rows < window).
 * The windows are never materialised: kernels index the sequence directly. */
int ww_slide_forward(ww_ctx *ctx, const ww_model *model, const float *mel, int64_t rows, int32_t hop, float *out,
                     int64_t *n_windows);

/* Device form.  Window w reads mel rows [d_win_row[w], d_win_row[w] + d_win_valid[w]) and is
 * zero padded at the end up to `window` rows (evaluate_tf_lite_opts.py:43-45). */
int ww_forward_windows_dev(ww_ctx *ctx, const ww_model *model, const float *d_mel, int64_t mel_rows,
                           const int64_t *d_win_row, const int32_t *d_win_valid, int32_t n_windows, float *d_out);

/* Several mel sequences in one device buffer (the padded clips of a test set), each slid over with the same hop:
 * sequence s has seg_nw[s] complete windows, window k of it covers rows [seg_row0[s] + k*hop, + window).
 * seg_row0 / seg_nw are HOST arrays; d_out receives the detect rows sequence by sequence.  This is the window loop of
 * utils/evaluate_models.py:66-88 over many files at once; for the CRNN the conv and the layer-1 projection of a time
 * position are computed once per sequence instead of once per window that contains it.  The host arrays are consumed before
 * the call returns; the kernels are enqueued like every _dev entry point's (the CRNN's launch descriptors travel through two
 * page-locked buffers of the context, so a caller can stage its next batch while this one computes). */
int ww_forward_segments_dev(ww_ctx *ctx, const ww_model *model, const float *d_mel, int64_t mel_rows, const int64_t *seg_row0,
                            const int32_t *seg_nw, int32_t n_seg, int32_t hop, float *d_out);

/* Whole hot path for a batch of equal-length clips resident in HBM (BASELINE configs 2/3):
 * PCM [n_clips][samples_per_clip] -> log-mel -> one zero-padded window per clip ->
 * encode + detect -> d_out [n_clips][n_out].  d_mel_scratch may be NULL (ctx workspace). */
int ww_clips_forward_dev(ww_ctx *ctx, const ww_model *model, const int16_t *d_pcm, int32_t n_clips,
                         int32_t samples_per_clip, const ww_frontend_params *fp, float *d_out);

/* ---- streaming (spokestack/pipeline.py + WakewordTrigger, BASELINE config 5) -------------
 * S independent 16 kHz streams advanced in lock step, 20 ms per tick.  Device-resident
 * state per stream: sample ring (512), mel window ring (window x 40), pre-emphasis carry,
 * running posterior max (tflite.py:96-108).  Per tick and stream: 0..2 new mel frames; for
 * each new frame while is_speech[s] != 0 one encode+detect (tflite.py:163-168,187-215).
 * post[s][k] (k < n_post[s] <= 2) are the posteriors produced by this tick, in order.
 * is_speech[s] is a bit set: bit 0 = the VAD says speech (context.is_speech); bit 1 = the stream is already
 * active (context.is_active): the reference does not sample an active stream at all (tflite.py:139-140), so its
 * rings stand still and it yields no posterior until the flag is cleared.  frames = [n_streams][320] int16
 * (20 ms at 16 kHz). */
/* flags: WW_STREAM_FULL_RECOMPUTE = every streaming CRNN window recomputed from its mel rows by the batch kernels instead of
 * the incremental crnn_stream_kernel (3 new time positions per window); results agree to 2e-6.  0 = default. */
#define WW_STREAM_FULL_RECOMPUTE 1u
/* Round 5: a tick of the incremental CRNN bank is ONE kernel launch - the front end of a stream's new frames runs inside the
 * workgroups of that stream's new windows - and ww_stream_step waits for a tick's posteriors by polling {value, tick number}
 * pairs the kernels store into page-locked memory instead of the runtime's completion signal (every bank whose context owns
 * its stream; on a borrowed stream the call still returns only when the stream has drained).  Same bits either way.
 * WW_STREAM_TWO_LAUNCH keeps the front-end kernel + model kernel form, WW_STREAM_SYNC_WAIT the hipStreamSynchronize wait:
 * the forms the tests compare against. */
#define WW_STREAM_TWO_LAUNCH 2u
#define WW_STREAM_SYNC_WAIT 4u
int ww_stream_create(ww_ctx *ctx, const ww_model *model, int32_t n_streams, const ww_frontend_params *fp, uint32_t flags,
                     ww_streams **out);
int ww_stream_destroy(ww_streams *st);
int ww_stream_step(ww_streams *st, const int16_t *frames, const uint8_t *is_speech, float *post, int32_t *n_post);
/* WakewordTrigger.reset (tflite.py:241-246) for the listed streams (ids NULL -> all). */
int ww_stream_reset(ww_streams *st, const int32_t *ids, int32_t n);
/* ---- the pipeline's host stages for S streams in lock step (BASELINE config 5 at the plugin surface) --------------------------
 * The reference drives three stage objects per stream and 20 ms frame (spokestack/pipeline.py:25-28, stage list of demo.py:29-36),
 * each a few comparisons on the shared SpeechContext.  For S streams each stage is ONE pass over plain arrays the caller owns
 * (uint8 is_speech[S] / is_active[S] are the S SpeechContexts' flags, wwhip/context.py: ContextBank); a stage returns the ids whose
 * flag it changed, so that activate / deactivate events (spokestack/context.py:71-85) are raised for those streams only.  No GPU
 * call in the first three; ctx-less, errors are WW_EINVAL.
 *
 * ww_vad_bank_step - VoiceActivityDetector.__call__ after the classifier (spokestack/vad/webrtc.py:59-77): raw[s] = the frame
 * classifier's decision; run_value / run_length = the detector's run state; is_speech is read and updated in place;
 * rise_frames / fall_frames = vad_rise_delay // frame_width, vad_fall_delay // frame_width.  n_changed (may be NULL) = streams
 * whose is_speech moved. */
int ww_vad_bank_step(int32_t n_streams, const uint8_t *raw, int32_t rise_frames, int32_t fall_frames, uint8_t *run_value,
                     int64_t *run_length, uint8_t *is_speech, int32_t *n_changed);
/* ww_trigger_bank_step - WakewordTrigger.__call__ around the models (spokestack/wakeword/tflite.py:134-146,232-239) over the
 * posteriors post[S][2] / n_post[S] a tick delivered: running maximum, `posterior > threshold` (float32 against a Python
 * float: compared in double), is_active[s] set for the streams that fire and were not active (their ids -> fired_ids), VAD
 * falling edges against was_speech (updated) -> fall_ids, whose posterior_max is cleared; the caller resets those streams
 * (ww_stream_reset).  fired_ids / fall_ids hold up to S entries. */
int ww_trigger_bank_step(int32_t n_streams, const uint8_t *is_speech, uint8_t *is_active, const float *post, const int32_t *n_post,
                         double threshold, uint8_t *was_speech, float *posterior_max, int32_t *fired_ids, int32_t *n_fired,
                         int32_t *fall_ids, int32_t *n_fall);
/* ww_timeout_bank_step - ActivationTimeout.__call__ (spokestack/activation_timeout.py:25-38): min_frames / max_frames are
 * min_active / frame_width and max_active / frame_width as the reference keeps them (floats); is_active[s] cleared and
 * active_frames[s] zeroed for the streams that time out (-> deact_ids). */
int ww_timeout_bank_step(int32_t n_streams, const uint8_t *is_speech, uint8_t *is_active, uint8_t *was_speech, int32_t *active_frames,
                         double min_frames, double max_frames, int32_t *deact_ids, int32_t *n_deact);
/* ww_stream_step_trigger - the wake-word stage of S streams as ONE call: ww_stream_step with bit 0 = is_speech[s], bit 1 =
 * is_active[s] as they stand before the tick, ww_trigger_bank_step over its posteriors, ww_stream_reset of the streams whose
 * VAD bit fell (tflite.py:143-146). */
int ww_stream_step_trigger(ww_streams *st, const int16_t *frames, const uint8_t *is_speech, uint8_t *is_active, double threshold,
                           uint8_t *was_speech, float *posterior_max, float *post, int32_t *n_post, int32_t *fired_ids,
                           int32_t *n_fired, int32_t *fall_ids, int32_t *n_fall);
/* ww_pipeline_bank_step - ONE tick of the whole stage list of demo.py:29-36 for S streams (spokestack/pipeline.py:25-28: for stage in
 * stages: stage(context, frame)) as one call: ww_vad_bank_step, ww_stream_step_trigger, ww_timeout_bank_step in this order on the
 * arrays the state block points to (the caller owns them all; the block itself may be reused from tick to tick).  Equivalent to
 * the three calls, stage by stage; what it saves is two trips through the host language's call layer per tick (wwhip/pipeline.py:
 * SpeechPipelineBank takes it when its stages are exactly VadBank, WakewordBank, ActivationTimeoutBank).  On return n_vad_changed,
 * n_fired / fired_ids, n_fall / fall_ids, n_deact / deact_ids say which streams changed. */
typedef struct ww_pipeline_state {
  uint8_t *is_speech, *is_active;          /* the S contexts' flags (updated in place) */
  const uint8_t *raw;                      /* VAD: the classifier's decisions of this tick */
  uint8_t *run_value;                      /*      run state */
  int64_t *run_length;
  uint8_t *wake_was_speech;                /* wake word: VAD edge detector, */
  float *posterior_max, *post;             /*            running maxima [S], this tick's posteriors [S][2] */
  int32_t *n_post;                         /*            and their counts [S] */
  uint8_t *timeout_was_speech;             /* timeout: VAD edge detector, */
  int32_t *active_frames;                  /*          frames since activation [S] */
  int32_t *fired_ids, *fall_ids, *deact_ids; /* out: ids (up to S each) */
  double threshold, min_frames, max_frames;
  int32_t rise_frames, fall_frames;
  int32_t n_vad_changed, n_fired, n_fall, n_deact; /* out */
} ww_pipeline_state;
int ww_pipeline_bank_step(ww_streams *st, const int16_t *frames, ww_pipeline_state *ps);
/* Where a tick's time goes on the HOST side of ww_stream_step (the loop of spokestack/pipeline.py:25-28 is host-paced, so
 * BASELINE config 5's per-tick latency is this call): mean nanoseconds per phase over the ticks since the last reset -
 * [0] plan (control words and window descriptors of the tick), [1] the caller's frames into the page-locked block,
 * [2] first kernel launch, [3] second kernel launch (0 where a tick is one launch), [4] waiting for the posteriors,
 * [5] copy-out into post / n_post.  A monotonic-clock read per phase: always on. */
#define WW_STREAM_TL_PHASES 6
int ww_stream_timeline(ww_streams *st, double *mean_ns, int64_t *ticks, int32_t reset);

/* ---- posterior smoothing + threshold sweep -----------------------------------------------
 * Replaces plot_FRR_FAR's numeric core (utils/evaluate_models.py:185-218):
 *   neg' = np.convolve(neg, ones(win)/win, 'same')   (fp64)
 *   frr[k] = (num_wakewords - #(pos > thr[k])) / num_wakewords
 *   fa_count[k] = #rising edges of (neg' > thr[k]);  fa_per_h[k] = fa_count[k] / hours
 * smoothed (may be NULL) receives neg' ([N] fp64).  win <= 0 skips the smoothing. */
int ww_far_frr(ww_ctx *ctx, const float *pos, int64_t n_pos, const float *neg, int64_t n_neg, int32_t win,
               const double *thr, int32_t n_thr, double num_wakewords, double hours, double *frr,
               double *fa_per_h, int64_t *fa_count, double *smoothed);

/* The same sweep over posteriors that are already on the context's device (the sharded evaluators: the values a rank's
 * kernels produced, or what the posterior gather delivered) - d_pos / d_neg are device pointers, nothing but the thresholds goes
 * up and nothing but the n_thr counters comes back; d_smoothed (device, [n_neg] fp64) may be NULL.  Returns after the
 * counters are on the host. */
int ww_far_frr_dev(ww_ctx *ctx, const float *d_pos, int64_t n_pos, const float *d_neg, int64_t n_neg, int32_t win,
                   const double *thr, int32_t n_thr, double num_wakewords, double hours, double *frr, double *fa_per_h,
                   int64_t *fa_count, double *d_smoothed);

/* Posterior pick + per-clip reduction on the device (utils/evaluate_models.py:80,86: element [1] of a detect row - pidx per
 * SURVEY quirk C1 -; :98-99: the max over a wake-word clip's windows).  d_rows: [n][n_out] detect rows as the model entry points
 * leave them.  d_seg_offs == NULL: d_out[i] = d_rows[i][pidx] for all n rows (:105-106, the negative stream: every window
 * counts); otherwise d_seg_offs is a device table of n_seg + 1 ascending row offsets and d_out[s] = max over rows
 * [d_seg_offs[s], d_seg_offs[s + 1]).  Enqueues on the context's stream, no synchronisation. */
int ww_posterior_pick_dev(ww_ctx *ctx, const float *d_rows, int64_t n, int32_t n_out, int32_t pidx, const int64_t *d_seg_offs,
                          int64_t n_seg, float *d_out);

/* ---- superframe shortest-path smoothing -------------------------------------------------
 * Replaces wwdetect/wfst.py:17-71 `smooth()` (pynini 2-state x T lattice, tropical shortest path) as
 * wired in utils/CRNN_files/tflite.py:252-263 (superframe of 10 posteriors; trigger if the best path
 * visits 'wakeword').  in: [n][T][2] = (p_other, p_wakeword) per step, or -ln of them when in_is_cost != 0
 * (pass NumPy's -np.log for bit-exact reference costs).  stay_bonus = 1 in the reference.  path
 * (may be NULL) receives the best state sequence [n][T] (0 = other, 1 = wakeword), wake[n] = 1 if it
 * visits state 1.  T <= 64. */
int ww_superframe_smooth(ww_ctx *ctx, const float *in, int64_t n, int32_t T, float stay_bonus, int32_t in_is_cost,
                         uint8_t *path, uint8_t *wake);

#ifdef __cplusplus
}
#endif
#endif /* WWHIP_H */
