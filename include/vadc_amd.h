/*
 * vadc_amd.h -- C-ABI of the MI355X-native Silero VAD backend (libvadc_amd.so).
 *
 * This is the drop-in boundary for the per-chunk forward pass of IntendedConsequence/vadc.  In the
 * reference the boundary is the compile-time backend trio selected at vadc.c:15-19:
 *
 *     backend_init            silero.h:48  / onnx_helpers.h:62
 *     backend_create_tensors  silero.h:76  / onnx_helpers.h:81
 *     backend_run             silero.h:53  / onnx_helpers.h:75
 *
 * whose only arithmetic is silero_run_one_batch_with_context (silero_v3.c:72-215).  The entry points
 * below are what an FFI binding for that path binds (plain pointers and sizes, no reference-internal
 * types such as MemoryArena / String8).  include/vadc_backend_hip.h adapts them to the exact
 * backend_* trio so that the header can be #included from a vadc-shaped host as a third backend.
 *
 * Conventions
 *   - every function returns 0 on success or a negative VADC_AMD_E* code; vadc_amd_last_error() gives
 *     the message of the calling thread's last failure.
 *   - there is NO CPU fallback: without a usable gfx950 device vadc_amd_create fails with
 *     VADC_AMD_ENODEVICE.
 *   - audio layout: samples[stream][chunk][1536], probabilities[stream][chunk][2] (element 1 is the speech
 *     probability, vadc.c:704-713 "output_dims == 3"), both stream-major.  Chunks of one stream are
 *     consecutive in time; LSTM state (h,c [2][64] per stream) lives on the device and is carried
 *     from call to call (the C-backend convention, silero.h:36-37, silero_v3.c:178-179).
 *   - the reference's `batch` (consecutive chunks of ONE stream, lstm.c:275-277) is n_streams = 1,
 *     n_chunks = batch.
 */
#ifndef VADC_AMD_H
#define VADC_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Revision of this interface: bumped whenever a struct below grows or an entry point changes meaning.  A binding checks vadc_amd_abi_version() against the value
 * it was compiled with BEFORE it hands the library a struct (vadc_amd_get_caps writes sizeof(vadc_amd_caps) of ITS revision): include/vadc_backend_hip.h and
 * vadc_amd/_lib.py do; a caller that must work across revisions uses vadc_amd_get_caps_sized. */
#define VADC_AMD_ABI_VERSION 6
int  vadc_amd_abi_version(void);

#define VADC_AMD_CHUNK_SAMPLES 1536    /* the reference's chunk; every buffer below is [stream][chunk][window], window = 1536 unless option "window" says otherwise */
#define VADC_AMD_HIDDEN        64
#define VADC_AMD_LSTM_LAYERS   2

enum {
   VADC_AMD_OK        =  0,
   VADC_AMD_EINVAL    = -1,   /* bad argument (NULL, out-of-range stream/chunk count, ...)        */
   VADC_AMD_EWEIGHTS  = -2,   /* weights blob is not a valid 99-tensor v3.1 / 36- or 37-tensor v4 / 13-tensor v5 container */
   VADC_AMD_ENODEVICE = -3,   /* no usable gfx950 device / HIP runtime failure at creation         */
   VADC_AMD_EHIP      = -4,   /* HIP runtime error during a call                                    */
   VADC_AMD_ENOMEM    = -5
};

/* model kind, decided by the weights container handed to vadc_amd_create and reported by vadc_amd_get_caps */
enum {
   VADC_AMD_MODEL_V31 = 0,   /* Silero v3.1 / 16 kHz: the 99-tensor .testtensor the reference's C backend loads (tensor.h:114-191) */
   VADC_AMD_MODEL_V4  = 1    /* Silero v4: 36-tensor container (16 kHz branch) or 37-tensor container (8 kHz branch `model_8k.*`: own weights, third strided
                                conv with stride 1, silero_vad.py:178-181; caps.sample_rate = 8000, windows 768 / 512 / 256) written from the reference's
                                silero_vad_v4.onnx by vadc_amd/onnx_weights.py.  The reference runs v4 only through onnxruntime (silero.h:59,
                                onnx_helpers.c:83-115); arithmetic per silero_vad.py:191-236.  One probability per chunk, written
                                to BOTH slots of probs[stream][chunk][2] so that hosts index it like v3.1. */,
   VADC_AMD_MODEL_V5  = 2    /* Silero v5 SHAPES (13-tensor container in the order of the reference's C test, test.c:2045-2068): per chunk the previous 64
                                samples + a 512-sample window, STFT hop 128, four k = 3 convs, LSTM(128), one probability (vadc.c:105-162,
                                silero_vad.py:290-434).  The reference ships no v5 weights; parity is pinned on seeded weights.  The
                                context is per-stream device state like h and c: samples[stream][chunk][512], state h, c [128] each. */
};

/* precision selector for vadc_amd_create.  In EVERY mode the LSTM state is fp32 and probabilities of modes 0 and 1 stay within 1e-4 of the C backend. */
enum {
   VADC_AMD_PRECISION_FP32 = 0,   /* the parity mode (BASELINE config 2).  STFT with the reference's exact fp32 reduction tree (stft.c:115-184): bit-identical
                                     magnitudes.  After the normalization every dense contraction runs at fp32 ACCURACY on whichever pipe is fastest:
                                     split-fp16 operands (a = hi + lo, 22 significant bits, exact products, fp32 accumulation: 3 fp16 MFMAs per fp32 one) for
                                     the LSTM gate GEMMs and every GEMM of the encoder (layer 1 included: k_layer1_regs); a weight outside fp16's range
                                     selects the fp32-MFMA form of the kernel it belongs to by itself, options "encoder"=3 / "lstm"=3 / "layer1"=1 force it. */
   VADC_AMD_PRECISION_SPLIT16 = 1,/* BASELINE config 3 ("reduced-precision compute + fp32 LSTM state"), as SURVEY.md section 0 prescribes it: the STFT keeps the
                                     reference's exact tree -- any other summation order moves near-silent bins, log1p(2^20 x) amplifies that, and single chunks
                                     leave the 1e-4 bar (measured: up to 7e-4 with the STFT as a GEMM) -- and everything behind the normalization is
                                     split-precision 16-bit MFMA with fp32 accumulation.  Since k_frontend_sym made the exact tree as cheap as a GEMM this is
                                     what mode 0 runs by default too; mode 1 additionally refuses EVERY fp32-MFMA fallback: create fails with EWEIGHTS if a
                                     weight of the LSTM, of layers 2-4 or of the first layer does not fit fp16's range, and "encoder"=3 / "lstm"=3 /
                                     "layer1"=1 are rejected. */
   VADC_AMD_PRECISION_FAST_STFT = 2 /* throughput mode, NOT within the 1e-4 bar: the STFT as a folded real-input GEMM on the fp16 matrix pipe (split-fp16
                                     operands, any summation order): front end 2x faster than the exact tree; probabilities deviate from the C backend by up
                                     to ~7e-4 on single chunks (p99.9 3e-4; distribution in DESIGN.md).  Silero v4 (whose parity target is a framework
                                     convolution in any fp32 order) uses this front end in every mode.  If the loaded basis lacks the real-DFT symmetries
                                     the folding needs, the engine keeps the tree and caps.precision reports SPLIT16. */
};

typedef struct vadc_amd_engine vadc_amd_engine;

/* What backend_init reports back through Silero_Config (vadc.h:10-43; values silero.h:39-43). */
typedef struct vadc_amd_caps {
   int32_t batch_size_restriction;        /* -1: any                                   */
   int32_t is_silero_v5;                  /* 0                                         */
   int32_t input_size_min;                /* 1536; Silero v4: 512 (onnx_helpers.c:164-170) */
   int32_t input_size_max;                /* 1536                                      */
   int32_t output_dims;                   /* 3  => output [B,2,1]                      */
   int32_t output_stride;                 /* 2                                         */
   int32_t silero_probability_out_index;  /* 1                                         */
   int32_t lstm_hidden_size;              /* 64                                        */
   int32_t max_streams;
   int32_t max_chunks_per_call;
   int32_t device;
   int32_t precision;
   int32_t model_kind;                    /* VADC_AMD_MODEL_*                          */
   int32_t lstm_steps_per_chunk;          /* 7 (v3.1) / 3 ... 1 (v4: three strided stages keep 1 + (T - 1) / 2 of T = window / 64 frames each) */
   int32_t window_samples;                /* samples per chunk in effect: 1536; Silero v4 every multiple of 64 in 512 .. 1536 (option "window"); v4 8 kHz: 256 .. 768 */
   int32_t sample_rate;                   /* 16000; 8000 for the container of the v4 graph's 8 kHz branch               */
   int32_t context_size;                  /* 0; 64 for Silero v5 (vadc.c:697-701): kept per stream on the device, callers pass windows only */
   int32_t cu_partition_ok;               /* 1: the device has the CU-mask layout the LSTM partition rules were measured on (256 CUs, mask bit i -> XCD i % 8: checked at
                                             create) and the partition may be used; 0: any other layout (CPX / DPX mode, another part): no CU partition, plain streams */
   int32_t input_size_step;               /* the windows served are input_size_min + k * input_size_step <= input_size_max; 0: one window.  Silero v4: 64 -- the
                                             reference's onnxruntime path admits every count in 512 .. 1536 (onnx_helpers.c:164-170), this engine every multiple of
                                             64 samples (= one STFT frame) in that range; a host rounds a count in between DOWN (host/vadc_hip.c) */
} vadc_amd_caps;

/* ---- lifetime ------------------------------------------------------------------------------ */

/* Replaces backend_init (silero.h:21-46): parses the weights container (tensor.h:201-253 format,
 * positional order tensor.h:114-191), uploads/repacks the weights, allocates the device workspace for
 * max_streams x max_chunks_per_call chunks and zeroes every stream's LSTM state.  A 36-tensor container selects the
 * Silero v4 path (VADC_AMD_MODEL_V4).
 * device < 0 selects the current HIP device. */
int  vadc_amd_create(const void *weights_blob, size_t weights_len, int device,
                     int max_streams, int max_chunks_per_call, int precision,
                     vadc_amd_engine **out_engine);
void vadc_amd_destroy(vadc_amd_engine *e);
const char *vadc_amd_last_error(void);
int  vadc_amd_get_caps(const vadc_amd_engine *e, vadc_amd_caps *caps);
/* The same for a caller compiled against another revision of this header: vadc_amd_caps only ever grows at its end, and this entry point writes
 * min(caps_size, sizeof(vadc_amd_caps)) bytes -- pass sizeof of the struct you were compiled with. */
int  vadc_amd_get_caps_sized(const vadc_amd_engine *e, void *caps, size_t caps_size);

/* ---- the hot path: replaces backend_run (silero.h:53-74) ------------------------------------ */

/* Host buffers, synchronous (copies in, runs, copies out).  samples: f32 in [-1,1) exactly as
 * process_chunks hands them over (vadc.c:74-75); probs: [n_streams][n_chunks][2].
 * Limits of a call (else VADC_AMD_EINVAL): n_streams <= max_streams; n_streams * n_chunks <= max_streams * max_chunks_per_call; and, because the encoder ->
 * LSTM hand-off is stored in tiles of 16 streams, ceil(n_streams / 16) * n_chunks <= ceil(max_streams / 16) * max_chunks_per_call (so ONE stream may bring
 * up to ceil(max_streams / 16) * max_chunks_per_call chunks, not max_streams * max_chunks_per_call).
 * The buffers may be ordinary pageable memory: up to 32 MB per copy they travel through page-locked pieces the engine owns (no page-locking system call per call),
 * a larger one is locked in place by the runtime for the transfer.  Both are free for reuse when the call returns. */
int  vadc_amd_run_f32(vadc_amd_engine *e, const float *samples, int n_streams, int n_chunks, float *probs);
/* Same from s16le PCM; the /32768.0f of vadc.c:883,898 happens on the device (exact in fp32). */
int  vadc_amd_run_s16(vadc_amd_engine *e, const int16_t *pcm, int n_streams, int n_chunks, float *probs);

/* Device-resident buffers, asynchronous on `hip_stream` (a hipStream_t; NULL = HIP's default stream, as in
 * hipLaunchKernelGGL).  d_* are device pointers valid on the engine's device.  The call enqueues work only
 * (no allocation, no host synchronisation).  Larger calls fork onto the engine's own streams; for steady-state
 * replay use the engine's option "graph" (it captures the kernel sequences itself) -- the call queries its
 * stream and must not be issued while that stream is being captured.  Pass a stream of your own for overlap
 * between consecutive calls: HIP's NULL stream synchronises with the engine's CU-masked streams, which
 * serialises them (results are the same). */
int  vadc_amd_run_device_f32(vadc_amd_engine *e, const float *d_samples, int n_streams, int n_chunks,
                             float *d_probs, void *hip_stream);
int  vadc_amd_run_device_s16(vadc_amd_engine *e, const int16_t *d_pcm, int n_streams, int n_chunks,
                             float *d_probs, void *hip_stream);
/* Host-synchronous: returns when every call issued so far is complete. */
int  vadc_amd_synchronize(vadc_amd_engine *e);
/* Makes `hip_stream` wait (device-side, no host sync) for every call issued so far.  With option "defer_join" = 1 a vadc_amd_run_device_* call no longer
 * makes its own stream wait for its completion -- consecutive calls issued from ONE stream then overlap inside the engine (front end + encoder of call k+2
 * beside LSTM layer 0 of call k+1 beside layer 1 of call k) -- and this is how a consumer of the probabilities orders itself behind them. */
int  vadc_amd_join(vadc_amd_engine *e, void *hip_stream);

/* The speech probability alone: d_speech[stream][chunk] = d_probs[stream][chunk][1] (the element vadc reads, vadc.c:704-713), enqueued on `hip_stream`
 * (typically the stream that joined the call).  What the multi-GPU hosts gather: 4 B per chunk instead of the pair's 8. */
int  vadc_amd_speech_probabilities(vadc_amd_engine *e, const float *d_probs, int n_streams, int n_chunks, float *d_speech, void *hip_stream);

/* Host buffers, ASYNCHRONOUS: what a real backend_run caller holds -- host samples in, host probabilities out (vadc.c:873-909: the stream reader fills
 * host memory; silero.h:53-74 hands it to the backend) -- without the copy -> run -> copy serialisation of vadc_amd_run_*.  A call returns as soon as its
 * work is enqueued: the H2D copy on a copy stream of its own, the kernels behind it, the 8 bytes per chunk of probabilities back on a third stream
 * behind the call's completion; up to three calls are in flight (the fourth waits, on the host, for the first).  Both host buffers must stay valid and
 * unmodified until vadc_amd_wait_async.  Results are bit-identical to vadc_amd_run_device_* on the same inputs.  host_probs: [n_streams][n_chunks][2].
 * Page-locking: both buffers are page-locked on first sight (hipHostRegister; ranges that are page-locked already are taken as they are) and REMEMBERED --
 * up to 16 ranges, the least recently used one unregistered when a 17th arrives; a range that overlaps remembered ones without lying inside one replaces
 * them by their union.  Buffer lifetime rule: before freeing (or unmapping) a buffer that was ever passed here, call vadc_amd_unpin on it -- as for
 * any hipHostRegister'ed memory -- or run with option "pin_host" = 0 (no page-locking: pageable copies, about a third of the link rate). */
int  vadc_amd_run_s16_async(vadc_amd_engine *e, const int16_t *host_pcm, int n_streams, int n_chunks, float *host_probs);
int  vadc_amd_run_f32_async(vadc_amd_engine *e, const float *host_samples, int n_streams, int n_chunks, float *host_probs);
/* Host-synchronous: every asynchronous call issued so far has delivered its probabilities. */
int  vadc_amd_wait_async(vadc_amd_engine *e);
/* The caller is about to free the host buffer that contains `host_ptr`: waits for the asynchronous calls in flight and drops (unregisters) every
 * remembered range that contains it.  A pointer the engine never saw is not an error. */
int  vadc_amd_unpin(vadc_amd_engine *e, const void *host_ptr);

/* ---- per-stream state (the reference has one implicit stream; silero.h:36-37) ---------------- */
/* Zero the state of the listed streams (stream_ids == NULL: all max_streams). */
int  vadc_amd_reset_streams(vadc_amd_engine *e, const int32_t *stream_ids, int n);
int  vadc_amd_get_state(vadc_amd_engine *e, int stream, float *h /*[2][64]*/, float *c /*[2][64]*/);
int  vadc_amd_set_state(vadc_amd_engine *e, int stream, const float *h, const float *c);
/* Silero v5 only (caps.context_size = 64): the stream's 64-sample context, i.e. the tail of its previous window (vadc.c:697-701, 105-162) -- the third
 * piece of per-stream state next to h and c; a stream saved with get_state + get_context and restored elsewhere continues bit-identically. */
int  vadc_amd_get_context(vadc_amd_engine *e, int stream, float *ctx /*[64]*/);
int  vadc_amd_set_context(vadc_amd_engine *e, int stream, const float *ctx /*[64]*/);

/* ---- stage taps: the counterpart of the reference's bottom-up known-answer tests (test.c) ----- */
enum {
   VADC_AMD_STAGE_MAGNITUDE  = 0,   /* [n,129,25]  reflect pad + STFT + magnitude   stft.c:15-224      */
   VADC_AMD_STAGE_NORMALIZED = 1,   /* [n,129,25]  adaptive normalization           misc.c:1-124       */
   VADC_AMD_STAGE_LAYER1     = 2,   /* [n,16,13]   transformer_layer 1              transformer.c:237  */
   VADC_AMD_STAGE_LAYER2     = 3,   /* [n,32,7]                                                       */
   VADC_AMD_STAGE_LAYER3     = 4,   /* [n,32,7]                                                       */
   VADC_AMD_STAGE_LAYER4     = 5,   /* [n,64,7]    encoder output                   silero_v3.c:4-64   */
   VADC_AMD_STAGE_COUNT      = 6
};
/* Run the front end + encoder on n chunks (host f32 samples [n][1536], no LSTM, state untouched) and
 * copy out the requested stage. */
int  vadc_amd_debug_stage_from_samples(vadc_amd_engine *e, const float *samples, int n, int stage, float *out);
/* Feed `in` as the OUTPUT of stage `from_stage` (host, [n,...] in that stage's shape; MAGNITUDE or
 * NORMALIZED or LAYER1..3) and copy out stage `to_stage` (> from_stage). */
int  vadc_amd_debug_stage_from_stage(vadc_amd_engine *e, const float *in, int n, int from_stage, int to_stage, float *out);
/* LSTM + decoder only: x [n_streams][n_chunks][64][7] (encoder output layout), state from the engine. */
int  vadc_amd_debug_lstm_decoder(vadc_amd_engine *e, const float *x, int n_streams, int n_chunks, float *probs);
/* The decoder alone (silero_v3.c:231-303: ReLU -> conv 64 -> 2 -> mean over the steps -> sigmoid; reference fixture test.c:170): x [n][64][steps] stands
 * in for the second LSTM layer's output of n one-chunk streams inside the recurrence kernel, whose decoder then runs as in the product.  probs [n][2];
 * the streams' state is neither read nor written. */
int  vadc_amd_debug_decoder(vadc_amd_engine *e, const float *x, int n, float *probs);
/* Parts of the first encoder layer in isolation, for the reference's op-level fixtures.  what = 1, 2, 3, 5: `y` [n][16][25] enters the layer's own kernel
 * behind its conv block -- 1: dual_head_attention incl. the out projection (transformer.c:13-153; test.c:1105), 2: transformer_block (:160-234; test.c:1143),
 * 3: layer_norm with the block's norm1 parameters (misc.c:143-210; test.c:931), 5: the layer's tail on `y` as the strided conv's input -- conv k = 1 with
 * BatchNorm folded in, ReLU, every step (transformer.c:279-290, misc.c:98-141; test.c:966).  what = 4: `y` [n][129][25] runs through the product's input
 * pipeline and conv block (depthwise k = 5 + ReLU, pointwise + projection, ReLU: conv.c:17-113, 532-589, 761-814; test.c:545, 581, 820) with a zero
 * normalization offset and leaves behind the block's ReLU.  out [n][16][25].  Silero v3.1 only; 4 and 5 need the register-resident layer-1 kernel. */
int  vadc_amd_debug_layer1_block(vadc_amd_engine *e, int what, const float *y, int n, float *out);
/* Switches (all int-valued; an unknown key or value is VADC_AMD_EINVAL).  Each key exists for a FUNCTIONAL reason, named first -- the experiment toggles of rounds 2-5
 * ("fe_opt", "fe_gemm", "encoder" = 2 / 4 / 5, "encoder_batch", "v4_mag", "full_mask_streams", "lstm_trail" = 2) are gone with their kernels.
 *  How calls are issued (the caller's contract):
 *   "graph"       [steady-state replay]  1: kernel sequences (a whole small call; the front end + encoder of one chunk group of a forked call) are captured into hipGraphs
 *                 on first use and replayed afterwards -- the fork / join and the call-to-call ordering stay outside the graphs, so replays of consecutive steps
 *                 overlap like eager steps; 0 (default): eager launches
 *   "defer_join"  [a caller that pipelines calls]  1: a forked vadc_amd_run_device_* call does not make its own stream wait for its completion (see vadc_amd_join);
 *                 0 (default): strict stream semantics
 *   "groups"      [a single synchronous caller]  number of chunk groups a call is pipelined in: the LSTM of group g overlaps the front end + encoder of group g+1.
 *                 0 = auto (default): up to 4 for calls the caller waits for, 1 with "defer_join" (consecutive calls overlap instead)
 *   "window"      [Silero v4's other input sizes]  samples per chunk.  1536 (default; the only size of the reference's C backend, silero.h:41-42).  Silero v4: every
 *                 multiple of 64 in 512 ... 1536 (8 kHz branch: 256 ... 768) -- the v4 graph takes every count in that range (onnx_helpers.c:164-170,
 *                 --sequence_count vadc.c:743-752): samples / 64 STFT frames, every strided stage keeps 1 + (T - 1) / 2 steps.  The multiples of 256 run kernels
 *                 built for their geometry (1536 and the windows of 17 .. 23 frames in it: the register-resident first stage; stages 2 - 4 run in one launch at every window); a window in between runs the next larger built geometry with its own samples and
 *                 reflect pad staged and the surplus steps masked (no stage taps there).  Changes the stride of every samples / probability buffer; waits for
 *                 the calls issued before
 *   "h2d_streams" [host-buffer callers on a slow link]  1 (default) .. 4: pieces (= copy streams) of the H2D copy of an asynchronous host-buffer call
 *   "pin_host"    [callers whose buffers must not be page-locked]  1 (default): the asynchronous entry points page-lock the caller's buffers and remember them
 *                 (see vadc_amd_run_s16_async); 0: they do not
 *   "roctx"       [a profiler timeline of a host application]  1: every call and every kernel launch of it is bracketed by a named profiler range (roctxRangePush /
 *                 Pop: "vadc_amd_run_device [S streams x C chunks]", "k_frontend", "k_layer1", "k_enc234", "k_lstm", "k_lstm_l1") -- the counterpart of the reference's
 *                 Tracy zones (silero_v3.c:72-215); rocprofv3 --marker-trace shows them beside the kernel rows.  The marker library is looked up when the option is
 *                 switched on (EINVAL if there is none); 0 (default): no ranges, no lookup
 *  Which kernel form serves (every value other than the default is a FALLBACK the engine also takes by itself, or the literal-fp32 arithmetic):
 *   "lstm"        [weights outside fp16's range; the recurrence's two schedules]  0 = auto (default): split-fp16 operands on the fp16 matrix pipe at fp32 accuracy --
 *                 7 = layer-major (k_lstm_layer: layer 0 and layer 1 as two launches on two CU sets, pipelined over calls / chunk groups) for forked calls up to half
 *                 a chip of stream tiles and for calls of one or two chunks per stream, else 6 = k_lstm_wavefront_h3 (one workgroup per 16-stream tile, both
 *                 layers); 6 and 7 produce the same bits; 3 = k_lstm_wavefront_fused (fp32 MFMA), also what runs when an LSTM weight does not fit fp16's range
 *   "frontend"    [a basis without the DFT symmetries; unaligned input]  Silero v3.1: 0 = auto (default): k_frontend_sym (the reference's exact reduction tree for bins
 *                 0..32, the other 96 bins from the basis' DFT symmetries, bit for bit) when the loaded basis has those symmetries and the input is 16-byte
 *                 aligned, else k_frontend_fl; 1 = k_frontend_fl (the exact tree for all 129 bins).  Silero v4: 0 = the GEMM front end (default: k_frontend_gemm2
 *                 for s16 input, k_frontend_gemm for f32 input), 1 = the tree kernel with the v4 geometry (also what serves a basis the GEMM cannot fold)
 *   "encoder"     [weights outside fp16's range; literal fp32]  0 (default) = first layer as one launch, layers 2-4 of Silero v3.1 fused into ONE persistent launch
 *                 (k_enc_fused: activations in registers, split-fp16 MFMA; Silero v4, default window and rate: stages 2-4 likewise, k_enc_fused_v4; Silero v5:
 *                 k_v5_encoder_h3 + k_v5_wih); 3 (Silero v3.1, v5) = one launch per layer with fp32 MFMA -- what the engine falls back to by itself when a weight
 *                 does not fit fp16 (v5: or the basis lacks the fold symmetries)
 *   "layer1"      [weights outside fp16's range; a failed self-check]  the first encoder layer / stage: 0 (default) = k_layer1_regs (Silero v3.1) / k_layer1_regs_v4
 *                 (Silero v4, default window): input by LDS-DMA, split-fp16 MFMAs, activations in registers; 1 = the K = 1 fp32-MFMA form of k_layer_mfma, which also
 *                 serves by itself when a weight does not fit fp16 or the register-resident kernel failed its create-time self-check ("layer1_selfcheck")
 *  The device's partitioning (each has a condition under which the engine turns it off by itself):
 *   "cu_partition" [another CU layout: CPX / DPX mode, another part]  1 (default): while the LSTM needs few CUs it gets CUs of its own (CU-masked streams), shared with the
 *                 front end + encoder stream when the chain has slack, disjoint otherwise -- and without a partition the front end + encoder stream is still a
 *                 CU-masked stream (every CU: a hardware queue of its own, as a plain stream it could share one with the recurrence's: 10,240 x 1 2.95 -> 1.77 M);
 *                 2: always shared; 0: never mask -- plain non-blocking streams everywhere.  A CU-masked stream has hipStreamDefault flags: it synchronises with
 *                 HIP's legacy NULL stream, so issue calls from a stream of your own.  "lstm_cus": size of that partition (multiple of 8; 0 = sized by the engine)
 *   "cu_mask_check" 1 (default): the LSTM's CU partition is used only on a device whose CU-mask layout passed the check at create (caps.cu_partition_ok);
 *                 0: trust the rules anyway; 2: behave as if the check had failed (tests)
 *   "fe_xcd"      [bisecting a placement problem]  1 (default): the exact-tree front end's workgroups take their blocks of positions in XCD-major order -- the two
 *                 workgroups that share a chunk write its 128-byte lines of Y behind the same L2 (same bits, 2.7 % less time); 0: in launch order
 *   "lstm_trail"  [a process whose kernels do not overlap: a profiler, a time-sliced GPU]  1 (default): with the layer-major LSTM on its CU partition, layer 1 of a call
 *                 is launched BESIDE layer 0 of the same call and follows its published progress a few steps behind (same XCD, same L2: no cache maintenance) -- a
 *                 call's recurrence takes one chain instead of two (the last call of a run ends 0.45 ms earlier at 256 x 96, a single call's latency halves);
 *                 0: layer 1 starts when layer 0 has finished.  Same bits.
 *                 Used only in a process whose kernels were SEEN to overlap at create ("kernels_overlap"): a tool that lets one kernel onto the device at a
 *                 time (rocprofv3 --pmc) would start layer 0 when layer 1 has ended -- there the engine launches the two one after the other by itself
 *                 ("lstm_trail_used" says what the last call did).  "overlap_check" 2: behave as if that probe had failed (tests).
 *                 FAIL-SAFE: a hand-over that fails at run time (layer 1's bounded wait of about 2 s runs out -- a time-sliced GPU, a tool that started serialising
 *                 kernels --, or a tile's pair sits on two XCDs) neither traps nor loses state: that layer-1 workgroup writes no state, and a REDO launch behind
 *                 the pair (always there, a no-op otherwise) does its tiles again from the pre-call state, in stream order -- the call's probabilities and state are
 *                 what the pair in turn produces, bit for bit, later calls consume only repaired state, and NO error is returned.  get_option "trail_recoveries"
 *                 counts the tiles done again; from the next host synchronisation on the engine launches the two layers in turn ("lstm_trail" reads 0).
 *                 Not repairable: a tile whose LAYER 0 never ran.  Then every later call -- a deferred vadc_amd_run_device_* and vadc_amd_join included, without
 *                 any synchronisation: the word is host memory -- fails with VADC_AMD_EHIP until vadc_amd_reset_streams(e, NULL, 0).
 *                 Tests: "trail_fault" 1 = the next pair's layer 0 comes late (behind its layer 1), 2 = never; "trail_wait" = polls before layer 1 gives up */
int  vadc_amd_set_option(vadc_amd_engine *e, const char *key, int value);
/* Reads a switch back, plus read-only facts: "lstm_cus" = CUs reserved for the LSTM by the last call (0 = no partition), "lstm_shared" = 1 when those CUs
 * are also in the front end + encoder stream's mask, "lstm_kernel" = the LSTM variant it ran
 * (3 / 6 / 7), "frontend_kernel" = its front end (0 k_frontend_sym, 1 k_frontend_fl, 2 a GEMM front end -- k_frontend_gemm2 for s16, k_frontend_gemm for f32 input --, 3 k_frontend v4 tree; Silero v5: 2 = k_v5_encoder_h3, 1 = k_v5_encoder), "layer1_kernel" = the first
 * layer's form that runs (0 register-resident, 1 per-layer: option "layer1" is the request), "layer1_selfcheck" (1: the register-resident first layer agreed
 * with the per-layer form on the probe chunks at create; 0: it did not and the per-layer form serves; -1: not applicable), "zero_im0" (the basis' im row of
 * bin 0 is all zeros: k_frontend_sym skips its tree), "cu_layout_ok" (the device has the CU-mask layout the partition rules assume), "pinned_ranges" (host
 * ranges the asynchronous entry points currently keep page-locked). */
int  vadc_amd_get_option(vadc_amd_engine *e, const char *key, int *value);

/* ---- measurement: per-kernel HIP-event timing on the launch stream --------------------------- */
enum { VADC_AMD_KERNEL_FRONTEND = 0, VADC_AMD_KERNEL_LAYER1, VADC_AMD_KERNEL_LAYER2, VADC_AMD_KERNEL_LAYER3,
       VADC_AMD_KERNEL_LAYER4, VADC_AMD_KERNEL_LSTM /* both layers, or layer 0 of the layer-major form */, VADC_AMD_KERNEL_LSTM_L1 /* layer 1 + decoder of the layer-major form */,
       VADC_AMD_KERNEL_ENC234 /* encoder layers 2-4 in one launch (k_enc_fused); LAYER2..4 then stay at zero */,
       VADC_AMD_KERNEL_COUNT };
/* When enabled every kernel launch of run_* is bracketed by hipEventRecord on its stream. */
int  vadc_amd_set_profiling(vadc_amd_engine *e, int enabled);
/* Synchronizes, then returns launch count and summed duration (ms) since the last reset. */
int  vadc_amd_get_kernel_time(vadc_amd_engine *e, int kernel, int *launches, double *total_ms);
int  vadc_amd_reset_kernel_times(vadc_amd_engine *e);
const char *vadc_amd_kernel_name(int kernel);

#ifdef __cplusplus
}
#endif
#endif /* VADC_AMD_H */
