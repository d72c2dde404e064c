// engine_internal.h -- what the translation units of the engine share: the kernel launchers' prototypes, the engine's state (struct vadc_amd_engine), the weights container's
// host form and the packers' entry points.  engine.hip: lifetime, options, scheduling, the hot path, host-buffer entry points, stage taps.  engine_weights.hip: the
// .testtensor container -> the device images of the three models (fragment packing, split-fp16 operands, symmetry checks).
#pragma once
#include "../../include/vadc_amd.h"
#include "gemm2_pack.h"
#include "common.h"
#include "enc_fused_layout.h"

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dlfcn.h>
#include <string>
#include <vector>

namespace vadc {
void launch_frontend_fl_f32(const float *, const float *, float *, float *, size_t, int, ItemMap, int, hipStream_t);
void launch_frontend_fl_s16(const int16_t *, const float *, float *, float *, size_t, int, ItemMap, int, hipStream_t);
void launch_frontend_sym_f32(const float *, const float *, float *, float *, size_t, int, ItemMap, int, hipStream_t, int, int);
void launch_frontend_sym_s16(const int16_t *, const float *, float *, float *, size_t, int, ItemMap, int, hipStream_t, int, int);
void launch_frontend_v4_f32(const float *, const float *, float *, float *, float *, size_t, int, ItemMap, hipStream_t);
void launch_frontend_v4_s16(const int16_t *, const float *, float *, float *, float *, size_t, int, ItemMap, hipStream_t);
void launch_frontend_gemm_f32(const float *, const float *, const float *, float *, float *, float *, size_t, int, ItemMap, int, hipStream_t, int, int);
void launch_frontend_gemm2_s16(const int16_t *, const float *, const float *, float *, float *, size_t, int, ItemMap, int, hipStream_t, int, int);
void launch_normalize_tap(const float *, const float *, size_t, float *, int, hipStream_t, int);
void launch_lognorm_from_magnitude(const float *, float *, float *, size_t, int, hipStream_t);
void launch_lstm(int, const float *, const LstmWeights &, float *, float *, float *, int, int, int, int, hipStream_t, int, int);
void launch_lstm_layer(int, const float *, float *, const LstmWeights &, float *, float *, float *, int, int, int, int, hipStream_t, int, int, int *, int, int *, int, int *, int *, int);
void launch_lstm_decoder_tap(const float *, const LstmWeights &, float *, int, hipStream_t, int, int);
struct LayerWeightsM {
   const float *dw_w, *dw_b, *pw_f, *pj_f, *cb_b, *qkv_f, *qkv_b, *out_f, *out_b, *n1_w, *n1_b, *l1_f, *l1_b, *l2_f, *l2_b,
      *n2_w, *n2_b, *cv_f, *cv_b, *pwj_k1;
   const _Float16 *qkv_h, *out_h, *l1_h, *l2_h, *cv_h;
   const _Float16 *pw_h, *pj_h;
};
void launch_layer_mfma(int, const float *, const float *, const LayerWeightsM &, float *, int, ItemMap, int, size_t, hipStream_t);
void launch_enc_fused(const EncFusedArgs &, int, hipStream_t);
void launch_enc_fused_v4(const EncV4Args &, int, hipStream_t);
void launch_layer1_tap(int, const float *, const LayerWeightsM &, float *, int, ItemMap, hipStream_t);
void launch_layer1_regs(const L1RegsArgs &, int, hipStream_t);
void launch_layer1_regs_tap(int, const L1RegsArgs &, hipStream_t);
void launch_layer1_regs_v4(const L1RegsArgs &, int, hipStream_t);
struct V5Weights {
   const float *stft_f; const float *conv_f[4]; const float *conv_b[4]; const float *wih_f; const float *lstm_b; const float *whh; const _Float16 *whh_h;
   const float *dec_w; const float *dec_b;
   const _Float16 *h_stft; const _Float16 *h_conv[4]; const _Float16 *h_wih; const float *wny;      // k_v5_encoder_h3's split-fp16 operands (kernels_v5.hip); null: k_v5_encoder serves
};
void launch_v5_encoder_f32(const float *, float *, const V5Weights &, float *, float *, int, int, bool, int, hipStream_t);
void launch_v5_encoder_s16(const int16_t *, float *, const V5Weights &, float *, float *, int, int, bool, int, hipStream_t);
void launch_v5_lstm(const V5Weights &, const float *, float *, float *, float *, int, int, bool, hipStream_t);
void launch_layer_v4(int, const float *, const float *, const LayerWeightsM &, float *, int, ItemMap, int, size_t, hipStream_t, int, int, int);

}  // namespace vadc

using namespace vadc;

namespace vadc {
int fail(int code, const char *fmt, ...);                          // sets vadc_amd_last_error() and returns `code` (engine.hip)
hipError_t upload(void *dst, const void *src, size_t bytes);      // synchronous host -> device copy through a page-locked buffer of the call's own (engine.hip)
}

#define HIP_TRY(expr, code)                                                                       \
   do {                                                                                           \
      hipError_t e_ = (expr);                                                                     \
      if (e_ != hipSuccess) return fail(code, "%s failed: %s", #expr, hipGetErrorString(e_));     \
   } while (0)

// weights container (tensor.h:97-102, 201-253) -> host tensors, positional wiring tensor.h:114-191
struct HostTensor { std::vector<int> dims; const float *data; int size; };

struct LayerShape { int cin, d, t, stride, proj; };
static const LayerShape kLayers[4] = {{129, 16, 25, 2, 1}, {16, 32, 13, 2, 1}, {32, 32, 7, 1, 0}, {32, 64, 7, 1, 1}};
static const int kStageElemsV31[VADC_AMD_STAGE_COUNT] = {129 * 25, 129 * 25, 16 * 13, 32 * 7, 32 * 7, 64 * 7};
// Silero v4 (silero_vad.py:157-236): 24 frames, encoder [16,12] [32,6] [32,3] [64,3]
static const LayerShape kLayersV4[4] = {{258, 16, 24, 2, 1}, {16, 32, 12, 2, 1}, {32, 32, 6, 2, 0}, {32, 64, 3, 1, 1}};
// elements per chunk of every stage tap for a Silero v4 window of 64 * frames samples: T -> T/2 -> T/4 -> T/8 -> T/8 in the 16 kHz branch
// (frames 24 / 16 / 8), T -> T/2 -> T/4 -> T/4 -> T/4 in the 8 kHz branch (third strided conv with stride 1; frames 12 / 8 / 4)
static void stage_elems_v4(int frames, int stride3, int (&out)[VADC_AMD_STAGE_COUNT])
{
   const int t1 = (frames + 1) / 2, t2 = (t1 + 1) / 2, t3 = stride3 == 2 ? (t2 + 1) / 2 : t2;      // 1 + (T - 1) / stride per strided conv
   out[0] = out[1] = 129 * frames; out[2] = 16 * t1; out[3] = 32 * t2; out[4] = 32 * t3; out[5] = 64 * t3;
}

struct Packer {
   std::vector<float> buf;
   // every sub-array starts on a 64-byte boundary so scalar dwordx16 loads never straddle
   size_t add(const float *src, size_t n)
   {
      size_t off = (buf.size() + 15) & ~size_t(15);
      buf.resize(off + n);
      if (src) memcpy(buf.data() + off, src, n * sizeof(float));
      return off;
   }
};

struct vadc_amd_engine {
   int device = 0;
   int model = VADC_AMD_MODEL_V31;              // decided by the weights container: 99 tensors = v3.1, 36 = v4
   int frames = kFrames;                        // STFT frames per chunk of the geometry the kernels run: 25 (v3.1) / 24, 20, 16, 12, 8 (v4; 8 kHz branch 12, 8, 4)
   int frames_valid = kFrames;                  // ... of which the window in effect fills this many (= frames at the built windows; Silero v4 at a window that is no multiple of
                                                // 256 samples runs the next larger built geometry: the front end's `nrt`, k_layer_mfma's `tv`)
   bool padded_window() const { return frames_valid != frames; }
   int lstm_steps = 7;                          // LSTM steps per chunk: 7 (v3.1) / 3, 3, 2, 2, 1 (v4 with 1536-, 1280-, 1024-, 768-, 512-sample windows)
   int window = kChunk;                         // samples per chunk: 1536; Silero v4 also 1280 / 1024 / 768 / 512 (option "window", onnx_helpers.c:164-170); its 8 kHz branch 768 / 512 / 256
   V5Weights v5;                                // Silero v5 shapes (13-tensor container): kernels_v5.hip
   bool v5_enc_h3_ok = false;                   // ... its encoder's split-fp16 operands exist (weights x 256 inside fp16's range, basis with the fold symmetries): k_v5_encoder_h3 runs
   float *d_x35 = nullptr;                      // v5: conv 3's output as split-fp16 fragment pieces, 8 KB per 16 chunks (k_v5_encoder_h3 -> k_v5_wih, same stream)
   float *d_gx5[2] = {nullptr, nullptr}, *d_ctx5 = nullptr;   // v5: LSTM input projection [max_items][512] (one per hand-off parity); per-stream 64-sample context [max_streams][64]
   int sample_rate = 16000;                     // 8000: the 37-tensor container of the v4 graph's 8 kHz branch (third strided conv with stride 1)
   int stride3() const { return sample_rate == 8000 ? 1 : 2; }
   int v4_geo() const                           // k_frontend_gemm geometry of the (built) window the kernels run: 64 frames samples
   {
      const int tw = 64 * frames;
      if (sample_rate == 8000) return tw == 768 ? 4 : (tw == 512 ? 3 : 5);
      return tw == 1024 ? 2 : (tw == 512 ? 3 : (tw == 768 ? 4 : (tw == 1280 ? 6 : 1)));
   }
   int stage_elems[VADC_AMD_STAGE_COUNT] = {0};
   const float *d_afrag = nullptr, *d_nyq = nullptr;   // GEMM front end (v4 default, v3.1 in FAST_STFT precision): folded basis as MFMA A fragments, bin-128 weights
   const float *d_afrag2 = nullptr, *d_nyq2 = nullptr; // ... the same for its second form (k_frontend_gemm2: 32x32x16 MFMAs, s16 input)
   // (without a CU partition the front end + encoder stream is still a CU-masked stream -- with EVERY CU: a hardware queue of its own, see ensure_pipeline_streams -- unless
   // the caller asked for plain streams with "cu_partition" = 0.  Rounds 4-5 had an option for it, "full_mask_streams".)

   bool gemm_ok = false;                        // the loaded basis has the real-DFT symmetries the folded GEMM needs
   bool use_gemm_frontend() const { return gemm_ok && ((model == VADC_AMD_MODEL_V4 && frontend_variant == 0) || (model != VADC_AMD_MODEL_V4 && precision == VADC_AMD_PRECISION_FAST_STFT)); }
   float *d_MAG = nullptr;                      // v4 only: magnitudes [n][129][24] (the v4 encoder takes magnitude AND log-norm)
   int max_streams = 0, max_chunks = 0, precision = 0;
   size_t max_items = 0;
   hipStream_t stream = nullptr;
   float *d_weights = nullptr;
   const float *d_basis = nullptr;

   bool sym_ok = false;                         // the loaded basis has the bin-mirror / quarter-mirror DFT symmetries bit for bit: k_frontend_sym may run
   bool cu_layout_ok = false;                   // 256 CUs and CU-mask bit i -> XCD i % 8 (cu_mask_layout_ok): what the LSTM partition rules assume
   int cu_mask_check = 1;                       // option "cu_mask_check": 1 = the partition needs cu_layout_ok (default), 0 = trust the rules anyway, 2 = behave as if the check had failed (tests)
   double clock_scale = 1.0;                    // 2.4 GHz / the device's peak shader clock: scales the partition rules' measured times (lstm_slot_us, enc_us_per_chunk)
   bool kernels_overlap = false;                // two kernels on two masked streams were seen to run at the same time (cu_mask_layout_flags bit 1): what "lstm_trail" needs
   int overlap_check = 1;                       // option "overlap_check": 1 = "lstm_trail" needs kernels_overlap (default), 2 = behave as if the probe had failed (tests)
   bool trail_possible() const { return overlap_check == 1 && kernels_overlap; }
   bool cu_partition_usable() const { return cu_mask_check == 0 || (cu_mask_check == 1 && cu_layout_ok); }
   bool zero_im0 = false;                       // the basis' im row of bin 0 (-w[n] sin 0) is all +-0: k_frontend_sym skips that tree (its sums are +-0 whatever the input)
   int fe_xcd = 1;                              // option "fe_xcd": the exact-tree front end's workgroups take their blocks of positions in XCD-major order (kernels_frontend.hip, xcd_major_block): the two workgroups that share a chunk write its lines of Y behind one L2

   int frontend_variant = 0;                    // v3.1: 0 = auto (k_frontend_sym when the basis has the DFT symmetries, else k_frontend_fl), 1 = k_frontend_fl; v4: 0 = GEMM, 1 = tree
   LayerWeightsM lwm[4];
   int encoder_variant = 0;                     // option "encoder": 0 = default (layers 2-4 fused in one launch when the weights allow), 3 = fp32 MFMA, one launch per layer (also what serves a weight outside fp16's range)
   // k_enc_fused (kernels_encoder_fused.hip): the two LDS images (layers 2 + 3; layer 4) and the phase A -> phase B scratch
   void *d_encA = nullptr, *d_encB = nullptr;
   float *d_enc_scratch = nullptr;
   std::vector<unsigned char> h_encA, h_encB;   // built by build_weights, uploaded by vadc_amd_create
   // k_layer1_regs (kernels_layer1_regs.hip): its LDS image
   void *d_l1img = nullptr;
   std::vector<unsigned char> h_l1img;
   void *d_encv4 = nullptr;                     // Silero v4: LDS image of k_enc_fused_v4 (stages 2-4 in one launch; enc_fused_layout.h)
   std::vector<unsigned char> h_encv4;
   int layer1_selfcheck = -1;                   // -1: not run (no register-resident first layer in this engine), 1: it agrees with the per-layer form on the probe chunks, 0: it does not (the per-layer form serves)
   int layer1_variant = 0;                      // option "layer1": 0 = k_layer1_regs (registers + LDS-DMA) when the weights allow, 1 = the K = 1 fp32-MFMA form of k_layer_mfma
   bool use_l1_regs() const { return model == VADC_AMD_MODEL_V31 && d_l1img && layer1_variant == 0 && layer1_selfcheck != 0; }
   // Silero v4: k_layer1_regs_v4 serves the default window (24 frames); the magnitude half of the first stage's input is recovered from Y in every form
   // (frames = 24: the 1536-sample geometry -- the default window and, since round 6, the three windows of 21 .. 23 valid frames that run in it: the kernel's MASK form)
   bool use_l1_regs_v4() const { return model == VADC_AMD_MODEL_V4 && d_l1img && layer1_variant == 0 && encoder_variant == 0 && frames == 24 && layer1_selfcheck != 0; }
   // asynchronous host-buffer entry points (vadc_amd_run_*_async): three staging slots in flight -- H2D of call k+1 beside the kernels of call k beside
   // the D2H of call k-1, each on a stream of its own
   struct AsyncSlot { void *d_in = nullptr; float *d_probs = nullptr; hipEvent_t in_done = nullptr, out_done = nullptr; bool busy = false; };
   static constexpr int kAsyncSlots = 3;
   AsyncSlot aslot[kAsyncSlots];
   hipStream_t s_h2d = nullptr, s_d2h = nullptr, s_h2dx[3] = {nullptr, nullptr, nullptr};
   hipEvent_t ev_h2dx[3] = {nullptr, nullptr, nullptr};
   int h2d_parts = 1;                            // option "h2d_streams": pieces (= copy streams) of an asynchronous call's H2D copy
   unsigned anext = 0;
   struct HostRange { const char *p; size_t n; bool ours; unsigned long long last; };
   // host ranges the async entry points have seen: at most kMaxPinned, least recently used one evicted (and unregistered); ours: registered here.  A caller
   // that frees a buffer it has passed tells the engine first (vadc_amd_unpin); option "pin_host" = 0 turns the page-locking off altogether
   static constexpr size_t kMaxPinned = 16;
   std::vector<HostRange> pinned;
   unsigned long long pin_clock = 0;
   int pin_host = 1;

   // Silero v4: stages 2-4 in one launch (hot path only: LSTM tiles out, no stage taps) at every window of both branches; a weight outside fp16's range keeps the per-stage launches
   // (every window of both branches from 8 frames up: the step counts are kernel arguments.  The 8 kHz branch at 256 samples -- 4 frames, 2 -> 1 -> 1 steps -- keeps the three
   // small per-stage launches: the fused kernel's cost per chunk pair does not shrink with the steps, 6.31 against 6.54 M there, A/B on one box)
   bool use_enc_fused_v4() const { return model == VADC_AMD_MODEL_V4 && d_encv4 && encoder_variant == 0 && frames >= 8; }
   bool use_enc_fused() const { return model == VADC_AMD_MODEL_V31 && enc_h3_ok && d_encA && encoder_variant == 0; }
   LstmWeights lstm;
   // workspace
   float *d_in_f32 = nullptr;
   int16_t *d_in_s16 = nullptr;
   float *d_Y = nullptr, *d_FM = nullptr, *d_tap = nullptr;
   float *d_act[4] = {nullptr, nullptr, nullptr, nullptr};
   float *d_probs = nullptr;
   // The encoder -> LSTM hand-off buffer is double buffered over forked calls: d_act[3] aliases pair [xpar], so that
   // the encoder of call k+1 never waits for the LSTM of call k (it only waits for call k-1's, long finished).
   float *d_xpair[2] = {nullptr, nullptr};
   size_t x_tile_chunks = 0;                    // capacity of a hand-off buffer in (16-stream tile, chunk) blocks
   // layer-major LSTM (k_lstm_layer, variant 7): layer 0 -> layer 1 hand-off of the h0 sequence (same tile layout and size as an encoder hand-off
   // buffer), double buffered over forked calls like it: layer 1 of call k reads pair [xpar] while layer 0 of call k+1 writes the other one
   float *d_h0pair[2] = {nullptr, nullptr};
   // layer 1 beside layer 0 of the SAME call (k_lstm_layer's TRAIL form, option "lstm_trail"): per hand-off buffer and chunk group, one word per stream tile that
   // layer 0 publishes its progress in; the epoch (one per launch pair, 1 .. 2047) makes a value left by an earlier launch read as zero
   int *d_lstm_progress[2] = {nullptr, nullptr};
   size_t progress_tiles = 0;
   int lstm_epoch = 0;
   int lstm_trail = 1;
   int lstm_trail_used = 0;                     // whether the last call's layer-major launches were a TRAIL pair (option "lstm_trail_used", read only)
   int trail_fault = 0;                         // option "trail_fault" (tests): 1 = the next TRAIL pair's layer 0 is held back until its layer 1 has given up (a LATE layer 0: recovered by
                                                // the REDO launch); 2 = the next pair is launched WITHOUT its layer 0 (a layer 0 that NEVER ran: not recoverable, the fatal word)
   int trail_wait_limit = 4000000;              // option "trail_wait": polls (of ~0.5 us) after which a layer-1 workgroup gives up on its layer 0: ~2 s
   volatile int *h_trail_err = nullptr;         // FATAL word (host memory mapped into the device; one plain store of 1 by the REDO launch): a tile's layer 0 never finished, so
                                                // the call could not be recovered -- stream state is inconsistent; every later call fails until vadc_amd_reset_streams(all)
   int *d_trail_err = nullptr;                  // ... its device address
   int *d_trail_recov = nullptr;                // device counter: tiles the REDO launches have done again (read at host synchronisation points and by "trail_recoveries")
   int trail_recoveries = 0;                    // ... as of the last look
   int roctx = 0;                               // option "roctx": 1 = a named profiler range around every call and every kernel launch of it (see Roctx)
   bool trail_lost = false;                     // the fatal word was seen: calls are refused until every stream has been reset
   hipEvent_t ev_redo = nullptr;                // trail_fault 1: layer 0 of the faulted pair waits for its layer 1 to have given up
   int *d_lstm_tickets = nullptr;               // [2 layers][8 XCDs]: the counters a TRAIL workgroup draws its tile from (L2-local atomics); ticket_base = what earlier launches drew per XCD
   unsigned ticket_base = 0;
   // (round 1 also double buffered Y / FM for a front end on a third stream, option "fe_overlap"; measured slower and removed)
   int xpar = 0;
   float *d_h = nullptr, *d_c = nullptr;
   int lstm_variant = 0;
   // chunk-group pipeline: front end + encoder of group g+1 (stream A) overlap the LSTM of group g (stream B)
   static constexpr int kMaxGroups = 16;
   int groups = 0;                              // 0 = auto
   hipStream_t sA = nullptr, sB = nullptr, sC = nullptr;   // front end + encoder; LSTM (layer 0 in the layer-major form); layer 1 of the layer-major form
   bool streams_split = false;                  // sB / sC were created for the layer-major form (each half of the LSTM's CU partition)
   int n_cus = 0;
   bool enc_h3_ok = false;                      // every encoder GEMM weight fits fp16's range: layers 2-4 run their GEMMs in the split-fp16 form
   bool lstm_shared = false;                    // the LSTM partition's CUs are also in the other streams' mask
   int lstm_cus_forced = 0;                     // option "lstm_cus": CUs for the LSTM partition (0 = sized by lstm_partition_cus)
   int lstm_cus = -1;                           // CUs currently reserved for stream B (-1: streams not created)
   int last_lstm_kernel = -1;                   // what resolve_lstm chose for the last call
   int last_frontend_kernel = -1;               // what the last call's front end was: 0 = k_frontend_sym, 1 = k_frontend_fl, 2 = k_frontend_gemm, 3 = k_frontend (v4 tree)
   bool lstm_h3_ok = true;                      // every LSTM weight fits fp16's range (|w| < 3e4): the split-fp16 kernel may be used
   bool ev_b_valid[2] = {false, false};         // ev_b[p] has been recorded by a previous forked call that used pair p
   bool ev_c_valid[2] = {false, false};         // ev_c[p]: layer 1 of the last layer-major call that used pair p has read its h0 sequence

   int cu_partition = 1;                        // option "cu_partition": 0 = never mask CUs
   // hipGraph replay of the steady-state step (option "graph"): one instantiated graph per distinct call signature
   int use_graph = 0;
   int defer_join = 0;                          // option "defer_join": forked calls do not make the caller's stream wait for their completion; vadc_amd_join does
   struct GraphEntry { const void *in; float *out; int S, C, elem, G, gi, xp, lk, fe; hipGraph_t g; hipGraphExec_t x; };
   std::vector<GraphEntry> graphs;
   hipEvent_t ev_in = nullptr, ev_b[2] = {nullptr, nullptr}, ev_c[2] = {nullptr, nullptr}, ev_fe[kMaxGroups] = {nullptr},
              ev_l0[kMaxGroups] = {nullptr}, ev_wrap[2] = {nullptr, nullptr};
   // Call-to-call ordering that does not depend on which stream the caller used: last_a = the last work that touched the
   // front-end / encoder buffers, last_b / last_c = the last LSTM (per-stream state, hand-off buffers).  Every call makes the
   // stream(s) that touch these wait for them first and re-points them.
   // They are ALIASES of whichever event object was recorded at that point (one record per stream and call, not one per purpose), together with
   // the stream it was recorded on: a stream never waits for its own earlier work (it is in-order) -- every cross-queue wait and every record
   // is a barrier packet the command processor handles between kernels, and seven of them per call left 40 us between one call's last encoder
   // kernel and the next call's front end (4 % of the 256 x 96 step).
   hipEvent_t ev_last = nullptr;                                               // recorded by the calls that run on the caller's stream
   hipEvent_t last_a = nullptr, last_b = nullptr, last_c = nullptr;            // last_b: last work on layer-0 state, last_c: on layer-1 state (and probabilities)
   hipStream_t last_a_on = nullptr, last_b_on = nullptr, last_c_on = nullptr;
   bool last_on_valid = false;                                                 // false: the *_on handles may be stale (streams re-created): wait regardless
   bool ev_last_valid = false;
   // profiling
   bool profiling = false;
   struct EvPair { hipEvent_t a, b; };
   std::vector<EvPair> pending[VADC_AMD_KERNEL_COUNT];
   std::vector<EvPair> pool;
   int launches[VADC_AMD_KERNEL_COUNT] = {0};
   double total_ms[VADC_AMD_KERNEL_COUNT] = {0};
   // page-locked staging of the engine's own for the copies between a caller's PAGEABLE memory and the device (host_to_device / device_to_host below)
   static constexpr size_t kBouncePiece = (size_t)4 << 20, kBounceMax = (size_t)32 << 20;
   char *bounce[2] = {nullptr, nullptr};
   hipEvent_t bounce_ev[2] = {nullptr, nullptr};
   bool bounce_busy[2] = {false, false};
};

namespace vadc {
bool parse_testtensor(const unsigned char *p, size_t len, std::vector<HostTensor> &out);      // engine_weights.hip
int build_weights(vadc_amd_engine *e, const std::vector<HostTensor> &ts);                      // Silero v3.1 (99 tensors)
int build_weights_v4(vadc_amd_engine *e, const std::vector<HostTensor> &ts);                   // Silero v4 (36 / 37 tensors)
int build_weights_v5(vadc_amd_engine *e, const std::vector<HostTensor> &ts);                   // Silero v5 shapes (13 tensors)
}
