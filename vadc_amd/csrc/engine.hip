// engine.hip -- the C-ABI of include/vadc_amd.h: device workspace, per-stream LSTM state, scheduling and kernel launches,
// host-buffer entry points, stage taps and HIP-event timing.  (The weights container and its repacking: engine_weights.hip;
// the engine's state: engine_internal.h.)
//
// Host-side counterpart of the reference's backend glue (silero.h:21-81) and of the orchestration in
// silero_run_one_batch_with_context (silero_v3.c:72-215).  No arithmetic of the path happens on the host
// except the load-time weight repacking (transposes, basis permutation, BatchNorm folding).
#include "engine_internal.h"

static thread_local std::string g_err;

int vadc::fail(int code, const char *fmt, ...)
{
   char buf[512];
   va_list ap;
   va_start(ap, fmt);
   vsnprintf(buf, sizeof(buf), fmt, ap);
   va_end(ap);
   g_err = buf;
   return code;
}

// ---------------------------------------------------------------------------------------------------
// Copies between the caller's pageable memory and the device.  Handed a pageable pointer, the HIP runtime page-locks the range for the transfer (a userptr mapping of
// heap pages into the GPU's address space, made and torn down per call) -- the engine's synchronous entry points, its debug taps and vadc_amd_create would do that
// thousands of times in a test run, on whatever heap block the caller's allocator handed out.  Up to kBounceMax bytes go through two page-locked pieces the engine owns
// instead (memcpy of piece k + 1 beside the DMA of piece k; no system call per copy); a larger transfer is locked in place by the runtime as before (there the
// extra pass over the data would cost more than the locking: the 75 MB input of a 256 x 96 call).
// ---------------------------------------------------------------------------------------------------
static hipError_t bounce_ready(vadc_amd_engine *e)
{
   for (int i = 0; i < 2; ++i) {
      if (!e->bounce[i]) { void *p = nullptr; hipError_t he = hipHostMalloc(&p, vadc_amd_engine::kBouncePiece, hipHostMallocDefault); if (he != hipSuccess) return he; e->bounce[i] = static_cast<char *>(p); }
      if (!e->bounce_ev[i]) { hipError_t he = hipEventCreateWithFlags(&e->bounce_ev[i], hipEventDisableTiming); if (he != hipSuccess) return he; }
   }
   return hipSuccess;
}
// enqueued on `st`; `src` may be reused when the call returns
static hipError_t host_to_device(vadc_amd_engine *e, void *dst, const void *src, size_t bytes, hipStream_t st)
{
   if (bytes > vadc_amd_engine::kBounceMax) return hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, st);
   hipError_t he = bounce_ready(e);
   for (size_t off = 0, k = 0; off < bytes && he == hipSuccess; off += vadc_amd_engine::kBouncePiece, ++k) {
      const int i = (int)(k & 1);
      const size_t n = std::min(vadc_amd_engine::kBouncePiece, bytes - off);
      if (e->bounce_busy[i]) { he = hipEventSynchronize(e->bounce_ev[i]); e->bounce_busy[i] = false; if (he != hipSuccess) break; }
      memcpy(e->bounce[i], static_cast<const char *>(src) + off, n);
      he = hipMemcpyAsync(static_cast<char *>(dst) + off, e->bounce[i], n, hipMemcpyHostToDevice, st);
      if (he == hipSuccess) he = hipEventRecord(e->bounce_ev[i], st);
      e->bounce_busy[i] = he == hipSuccess;
   }
   return he;
}
// returns when the bytes are in `dst` (everything enqueued on `st` before has finished then)
static hipError_t device_to_host(vadc_amd_engine *e, void *dst, const void *src, size_t bytes, hipStream_t st)
{
   if (bytes > vadc_amd_engine::kBounceMax) { hipError_t he = hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, st); return he == hipSuccess ? hipStreamSynchronize(st) : he; }
   hipError_t he = bounce_ready(e);
   for (int i = 0; i < 2 && he == hipSuccess; ++i) if (e->bounce_busy[i]) { he = hipEventSynchronize(e->bounce_ev[i]); e->bounce_busy[i] = false; }
   const size_t P = vadc_amd_engine::kBouncePiece, pieces = (bytes + P - 1) / P;
   for (size_t k = 0; k <= pieces && he == hipSuccess; ++k) {
      if (k < pieces) {                                         // piece k on its way ...
         he = hipMemcpyAsync(e->bounce[k & 1], static_cast<const char *>(src) + k * P, std::min(P, bytes - k * P), hipMemcpyDeviceToHost, st);
         if (he == hipSuccess) he = hipEventRecord(e->bounce_ev[k & 1], st);
      }
      if (k > 0 && he == hipSuccess) {                          // ... while piece k - 1 goes to the caller
         he = hipEventSynchronize(e->bounce_ev[(k - 1) & 1]);
         if (he == hipSuccess) memcpy(static_cast<char *>(dst) + (k - 1) * P, e->bounce[(k - 1) & 1], std::min(P, bytes - (k - 1) * P));
      }
   }
   return he;
}
// vadc_amd_create's uploads (no stream yet): synchronous, through a page-locked buffer of the call's own
hipError_t vadc::upload(void *dst, const void *src, size_t bytes)
{
   if (!bytes) return hipSuccess;
   const size_t P = std::min(bytes, (size_t)8 << 20);
   void *h = nullptr;
   hipError_t he = hipHostMalloc(&h, P, hipHostMallocDefault);
   for (size_t off = 0; off < bytes && he == hipSuccess; off += P) {
      const size_t n = std::min(P, bytes - off);
      memcpy(h, static_cast<const char *>(src) + off, n);
      he = hipMemcpy(static_cast<char *>(dst) + off, h, n, hipMemcpyHostToDevice);
   }
   if (h) (void)hipHostFree(h);
   return he;
}

// ---------------------------------------------------------------------------------------------------
// lifetime
// ---------------------------------------------------------------------------------------------------
extern "C" const char *vadc_amd_last_error(void) { return g_err.c_str(); }
extern "C" int vadc_amd_abi_version(void) { return VADC_AMD_ABI_VERSION; }

extern "C" void vadc_amd_destroy(vadc_amd_engine *e)
{
   if (!e) return;
   // VADC_AMD_TRACE_TEARDOWN=1: a mark on stderr per step (a teardown that does not come back -- seen once in some hundred short-lived processes on ROCm 7.2 -- says where it stands)
   const bool trace = getenv("VADC_AMD_TRACE_TEARDOWN") != nullptr;
   auto mark = [&](const char *what) { if (trace) { fprintf(stderr, "vadc_amd_destroy: %s\n", what); fflush(stderr); } };
   mark("begin");
   (void)hipSetDevice(e->device);
   // The first hipFree below waits for the whole device, the engine's own streams included.  (An explicit hipStreamSynchronize on the idle CU-masked
   // streams here hung one short-lived process in ten -- bisected, ROCm 7.2 -- so there is none.)
   if (e->stream) (void)hipStreamSynchronize(e->stream);
   mark("engine stream idle");
   for (int k = 0; k < VADC_AMD_KERNEL_COUNT; ++k)
      for (auto &p : e->pending[k]) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
   for (auto &p : e->pool) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
   void *ptrs[] = {e->d_weights, e->d_in_f32, e->d_in_s16, e->d_Y, e->d_MAG, e->d_FM, e->d_tap, e->d_act[0], e->d_act[1],
                   e->d_act[2], e->d_xpair[0], e->d_xpair[1], e->d_probs, e->d_h, e->d_c, e->d_h0pair[0], e->d_h0pair[1], e->d_lstm_progress[0], e->d_lstm_progress[1], e->d_lstm_tickets, e->d_gx5[0], e->d_gx5[1], e->d_x35, e->d_ctx5, e->d_encA, e->d_encB, e->d_enc_scratch, e->d_l1img, e->d_encv4};
   mark("events gone");
   { bool first = true; for (void *p : ptrs) if (p) { (void)hipFree(p); if (first) mark("first hipFree (device-wide wait) back"); first = false; } }
   mark("device buffers freed");
   if (e->h_trail_err) (void)hipHostFree(const_cast<int *>(e->h_trail_err));
   for (int i = 0; i < 2; ++i) { if (e->bounce[i]) (void)hipHostFree(e->bounce[i]); if (e->bounce_ev[i]) (void)hipEventDestroy(e->bounce_ev[i]); }
   if (e->d_trail_recov) (void)hipFree(e->d_trail_recov);
   if (e->ev_redo) (void)hipEventDestroy(e->ev_redo);
   for (auto &sl : e->aslot) {
      if (sl.d_in) (void)hipFree(sl.d_in);
      if (sl.d_probs) (void)hipFree(sl.d_probs);
      if (sl.in_done) (void)hipEventDestroy(sl.in_done);
      if (sl.out_done) (void)hipEventDestroy(sl.out_done);
   }
   for (auto &r : e->pinned) if (r.ours) (void)hipHostUnregister(const_cast<char *>(r.p));
   mark("host memory freed / unregistered");
   for (int i = 0; i < 3; ++i) { if (e->s_h2dx[i]) (void)hipStreamDestroy(e->s_h2dx[i]); if (e->ev_h2dx[i]) (void)hipEventDestroy(e->ev_h2dx[i]); }
   if (e->s_h2d) (void)hipStreamDestroy(e->s_h2d);
   if (e->s_d2h) (void)hipStreamDestroy(e->s_d2h);
   if (e->stream) (void)hipStreamDestroy(e->stream);
   for (auto &ge : e->graphs) { (void)hipGraphExecDestroy(ge.x); (void)hipGraphDestroy(ge.g); }
   if (e->sA) (void)hipStreamDestroy(e->sA);
   if (e->sB) (void)hipStreamDestroy(e->sB);
   if (e->sC) (void)hipStreamDestroy(e->sC);
   for (hipEvent_t ev : {e->ev_in, e->ev_b[0], e->ev_b[1], e->ev_c[0], e->ev_c[1], e->ev_last}) if (ev) (void)hipEventDestroy(ev);
   for (hipEvent_t ev : e->ev_l0) if (ev) (void)hipEventDestroy(ev);
   for (hipEvent_t ev : e->ev_wrap) if (ev) (void)hipEventDestroy(ev);
   for (hipEvent_t ev : e->ev_fe) if (ev) (void)hipEventDestroy(ev);
   delete e;
   mark("end");
}

static int cu_mask_layout_flags(int device, int n_cus);
static int layer1_selfcheck(vadc_amd_engine *e);
extern "C" int vadc_amd_create(const void *blob, size_t len, int device, int max_streams, int max_chunks,
                               int precision, vadc_amd_engine **out)
{
   if (!out) return fail(VADC_AMD_EINVAL, "create: out_engine is NULL");
   *out = nullptr;
   if (!blob || len < 8) return fail(VADC_AMD_EWEIGHTS, "create: empty weights blob");
   if (max_streams <= 0 || max_chunks <= 0 || (long)max_streams * max_chunks > (1L << 24))
      return fail(VADC_AMD_EINVAL, "create: max_streams=%d max_chunks_per_call=%d out of range", max_streams, max_chunks);
   if (precision != VADC_AMD_PRECISION_FP32 && precision != VADC_AMD_PRECISION_SPLIT16 && precision != VADC_AMD_PRECISION_FAST_STFT)
      return fail(VADC_AMD_EINVAL, "create: unsupported precision %d", precision);

   std::vector<HostTensor> ts;
   if (!parse_testtensor(static_cast<const unsigned char *>(blob), len, ts))
      return fail(VADC_AMD_EWEIGHTS, "create: weights blob is not a valid .testtensor container");
   if (ts.size() != 99 && ts.size() != 36 && ts.size() != 37 && ts.size() != 13)
      return fail(VADC_AMD_EWEIGHTS, "create: expected 99 tensors (Silero v3.1), 36 (Silero v4, 16 kHz), 37 (Silero v4, 8 kHz branch) or 13 (Silero v5 shapes), found %zu", ts.size());
   if (ts.size() == 37) {
      float sr = 0.0f;
      if (ts[36].size == 1) memcpy(&sr, ts[36].data, 4);
      if (sr != 8000.0f) return fail(VADC_AMD_EWEIGHTS, "create: the 37th tensor of a Silero v4 container must be the sample-rate marker [8000]");
   }

   int ndev = 0;
   if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
      return fail(VADC_AMD_ENODEVICE, "create: no HIP device available (this backend has no CPU fallback)");
   if (device < 0) { if (hipGetDevice(&device) != hipSuccess) device = 0; }
   if (device >= ndev) return fail(VADC_AMD_ENODEVICE, "create: device %d out of range (%d devices)", device, ndev);
   HIP_TRY(hipSetDevice(device), VADC_AMD_ENODEVICE);
   hipDeviceProp_t prop;
   HIP_TRY(hipGetDeviceProperties(&prop, device), VADC_AMD_ENODEVICE);
   if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
      return fail(VADC_AMD_ENODEVICE, "create: device %d is %s; this library carries gfx950 code only", device, prop.gcnArchName);

   vadc_amd_engine *e = new vadc_amd_engine();
   e->device = device; e->max_streams = max_streams; e->max_chunks = max_chunks; e->precision = precision;
   e->max_items = (size_t)max_streams * max_chunks;
   e->model = ts.size() == 99 ? VADC_AMD_MODEL_V31 : (ts.size() == 13 ? VADC_AMD_MODEL_V5 : VADC_AMD_MODEL_V4);
   e->sample_rate = ts.size() == 37 ? 8000 : 16000;
   e->window = e->model == VADC_AMD_MODEL_V5 ? 512 : (e->sample_rate == 8000 ? 768 : kChunk);   // v5: 512 + 64 of context (vadc.c:105-162); v4 8 kHz: the same 96 ms
   e->frames = e->model == VADC_AMD_MODEL_V4 ? e->window / 64 : kFrames;
   e->frames_valid = e->frames;
   e->lstm_steps = e->model == VADC_AMD_MODEL_V4 ? 3 : (e->model == VADC_AMD_MODEL_V5 ? 1 : 7);
   if (e->model == VADC_AMD_MODEL_V4) stage_elems_v4(e->frames, e->stride3(), e->stage_elems);
   else memcpy(e->stage_elems, kStageElemsV31, sizeof(kStageElemsV31));
   int rc = e->model == VADC_AMD_MODEL_V4 ? build_weights_v4(e, ts) : (e->model == VADC_AMD_MODEL_V5 ? build_weights_v5(e, ts) : build_weights(e, ts));
   if (rc != VADC_AMD_OK) { vadc_amd_destroy(e); return rc; }
   if (e->model == VADC_AMD_MODEL_V4 && e->sample_rate == 8000 && !e->gemm_ok) {
      // the v4 tree front end exists for the 1536-sample / 24-frame geometry of the 16 kHz branch only; it would read n x 1536 samples from an n x 768 buffer
      vadc_amd_destroy(e);
      return fail(VADC_AMD_EWEIGHTS, "create: the 8 kHz branch of Silero v4 runs on the GEMM front end only, and this container's STFT basis lacks the real-DFT symmetries that front end needs");
   }
   if (precision == VADC_AMD_PRECISION_SPLIT16 && (!e->lstm_h3_ok || (e->model != VADC_AMD_MODEL_V4 && !e->enc_h3_ok))) {
      vadc_amd_destroy(e);
      return fail(VADC_AMD_EWEIGHTS, "create: SPLIT16 precision runs every GEMM with split-fp16 operands, but a weight of this container does not fit fp16's range; use VADC_AMD_PRECISION_FP32");
   }
   const size_t N = e->max_items;
   hipError_t he = hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking);
   e->n_cus = prop.multiProcessorCount;
   if (prop.clockRate > 500000 && prop.clockRate < 5000000) e->clock_scale = 2400000.0 / prop.clockRate;      // kHz; MI355X reports 2,400,000
   e->cu_layout_ok = (cu_mask_layout_flags(device, e->n_cus) & 1) != 0;
   e->kernels_overlap = (cu_mask_layout_flags(device, e->n_cus) & 2) != 0;
   for (hipEvent_t *ev : {&e->ev_in, &e->ev_b[0], &e->ev_b[1], &e->ev_c[0], &e->ev_c[1], &e->ev_last}) if (he == hipSuccess) he = hipEventCreateWithFlags(ev, hipEventDisableTiming);
   for (int g = 0; g < vadc_amd_engine::kMaxGroups && he == hipSuccess; ++g) he = hipEventCreateWithFlags(&e->ev_l0[g], hipEventDisableTiming);
   for (int g = 0; g < 2 && he == hipSuccess; ++g) he = hipEventCreateWithFlags(&e->ev_wrap[g], hipEventDisableTiming);
   for (int g = 0; g < vadc_amd_engine::kMaxGroups && he == hipSuccess; ++g) he = hipEventCreateWithFlags(&e->ev_fe[g], hipEventDisableTiming);
   if (he == hipSuccess) he = hipMalloc(&e->d_in_f32, N * kChunk * sizeof(float));
   if (he == hipSuccess) he = hipMalloc(&e->d_in_s16, N * kChunk * sizeof(int16_t));
   if (he == hipSuccess) he = hipMalloc(&e->d_Y, N * kBins * kFrames * sizeof(float) + kL1YSlackBytes);
   if (he == hipSuccess) he = hipMalloc(&e->d_FM, kBinSplit * N * kFrames * sizeof(float));
   if (he == hipSuccess && e->model == VADC_AMD_MODEL_V4) he = hipMalloc(&e->d_MAG, N * kBins * kFrames * sizeof(float));
   if (he == hipSuccess) he = hipMalloc(&e->d_tap, N * kBins * kFrames * sizeof(float));
   for (int l = 0; l < 3 && he == hipSuccess; ++l) he = hipMalloc(&e->d_act[l], N * kStageElemsV31[2 + l] * sizeof(float));   // >= the v4 shapes
   // encoder output: LSTM-native layout, streams padded to whole tiles of 16
   const size_t padded_streams = (size_t)((max_streams + kLstmTile - 1) / kLstmTile) * kLstmTile;
   e->x_tile_chunks = padded_streams / kLstmTile * max_chunks;
   for (int p = 0; p < 2; ++p) {
      if (he == hipSuccess) he = hipMalloc(&e->d_xpair[p], padded_streams * max_chunks * 448 * sizeof(float));
      if (he == hipSuccess) he = hipMemset(e->d_xpair[p], 0, padded_streams * max_chunks * 448 * sizeof(float));
   }
   e->d_act[3] = e->d_xpair[0];
   if (he == hipSuccess && !e->h_encA.empty()) {
      he = hipMalloc(&e->d_encA, e->h_encA.size());
      if (he == hipSuccess) he = hipMalloc(&e->d_encB, e->h_encB.size());
      if (he == hipSuccess) he = upload(e->d_encA, e->h_encA.data(), e->h_encA.size());
      if (he == hipSuccess) he = upload(e->d_encB, e->h_encB.data(), e->h_encB.size());
      if (he == hipSuccess) he = hipMalloc(&e->d_enc_scratch, (N + 4) * kEncScratchPerChunk * sizeof(float));   // batches of up to 4 chunks: the last one may be partial
      e->h_encA.clear(); e->h_encA.shrink_to_fit(); e->h_encB.clear(); e->h_encB.shrink_to_fit();
   }
   if (he == hipSuccess && !e->h_encv4.empty()) {
      he = hipMalloc(&e->d_encv4, e->h_encv4.size());
      if (he == hipSuccess) he = upload(e->d_encv4, e->h_encv4.data(), e->h_encv4.size());
      e->h_encv4.clear(); e->h_encv4.shrink_to_fit();
   }
   if (he == hipSuccess && !e->h_l1img.empty()) {
      he = hipMalloc(&e->d_l1img, e->h_l1img.size());
      if (he == hipSuccess) he = upload(e->d_l1img, e->h_l1img.data(), e->h_l1img.size());
      e->h_l1img.clear(); e->h_l1img.shrink_to_fit();
   }
   for (int p = 0; p < 2 && he == hipSuccess; ++p) he = hipMalloc(&e->d_h0pair[p], padded_streams * max_chunks * 448 * sizeof(float));
   e->progress_tiles = (padded_streams / kLstmTile + 7) / 8 * 8;
   if (he == hipSuccess) he = hipMalloc(&e->d_lstm_tickets, 16 * sizeof(int));
   if (he == hipSuccess) he = hipMemset(e->d_lstm_tickets, 0, 16 * sizeof(int));
   if (he == hipSuccess) he = hipMalloc(&e->d_trail_recov, sizeof(int));
   if (he == hipSuccess) he = hipMemset(e->d_trail_recov, 0, sizeof(int));
   if (he == hipSuccess) he = hipEventCreateWithFlags(&e->ev_redo, hipEventDisableTiming);
   // the TRAIL pair's FATAL word: host memory mapped into the device, so that the host reads it at the head of every call without a copy or a synchronisation
   { void *hp = nullptr, *dp = nullptr;
     if (he == hipSuccess) he = hipHostMalloc(&hp, 2 * sizeof(int), hipHostMallocMapped);      // [0] fatal, [1] "a tile was done again" (then the device counter is worth a copy)
     if (he == hipSuccess) { static_cast<int *>(hp)[0] = static_cast<int *>(hp)[1] = 0; he = hipHostGetDevicePointer(&dp, hp, 0); }
     e->h_trail_err = static_cast<volatile int *>(hp); e->d_trail_err = static_cast<int *>(dp); }
   for (int p = 0; p < 2 && he == hipSuccess; ++p) {
      he = hipMalloc(&e->d_lstm_progress[p], vadc_amd_engine::kMaxGroups * 3 * e->progress_tiles * sizeof(int));      // per group: [tiles] counts, [tiles] XCC ids, [tiles] layer 1 done
      if (he == hipSuccess) he = hipMemset(e->d_lstm_progress[p], 0, vadc_amd_engine::kMaxGroups * 3 * e->progress_tiles * sizeof(int));
   }
   if (he == hipSuccess && e->model == VADC_AMD_MODEL_V5) he = hipMalloc(&e->d_gx5[0], N * 512 * sizeof(float));
   if (he == hipSuccess && e->model == VADC_AMD_MODEL_V5) he = hipMalloc(&e->d_gx5[1], N * 512 * sizeof(float));
   if (he == hipSuccess && e->model == VADC_AMD_MODEL_V5) he = hipMalloc(&e->d_x35, ((N + 15) / 16) * 8192);      // conv 3 -> k_v5_wih hand-off: 8 KB per 16 chunks
   if (he == hipSuccess && e->model == VADC_AMD_MODEL_V5) he = hipMalloc(&e->d_ctx5, (size_t)max_streams * 64 * sizeof(float));
   if (he == hipSuccess && e->model == VADC_AMD_MODEL_V5) he = hipMemset(e->d_ctx5, 0, (size_t)max_streams * 64 * sizeof(float));
   if (he == hipSuccess) he = hipMalloc(&e->d_probs, N * 2 * sizeof(float));
   if (he == hipSuccess) he = hipMalloc(&e->d_h, (size_t)max_streams * 128 * sizeof(float));
   if (he == hipSuccess) he = hipMalloc(&e->d_c, (size_t)max_streams * 128 * sizeof(float));
   if (he == hipSuccess) he = hipMemset(e->d_h, 0, (size_t)max_streams * 128 * sizeof(float));
   if (he == hipSuccess) he = hipMemset(e->d_c, 0, (size_t)max_streams * 128 * sizeof(float));
   if (he == hipSuccess) he = hipDeviceSynchronize();
   if (he != hipSuccess) {
      rc = fail(VADC_AMD_ENOMEM, "create: device allocation failed: %s", hipGetErrorString(he));
      vadc_amd_destroy(e);
      return rc;
   }
   if (precision == VADC_AMD_PRECISION_SPLIT16 && e->model != VADC_AMD_MODEL_V5 && !e->d_l1img) {
      vadc_amd_destroy(e);
      return fail(VADC_AMD_EWEIGHTS, "create: SPLIT16 precision runs every GEMM with split-fp16 operands, but a first-layer weight of this container does not fit fp16's range; use VADC_AMD_PRECISION_FP32");
   }
   rc = layer1_selfcheck(e);
   if (rc == VADC_AMD_OK && e->layer1_selfcheck == 0 && precision == VADC_AMD_PRECISION_SPLIT16)
      rc = fail(VADC_AMD_EHIP, "create: the register-resident first layer failed its self-check and SPLIT16 refuses the fp32-MFMA fallback");
   if (rc) { vadc_amd_destroy(e); return rc; }
   *out = e;
   return VADC_AMD_OK;
}

// caller's struct may be an older, shorter vadc_amd_caps: never write past what it has room for
extern "C" int vadc_amd_get_caps_sized(const vadc_amd_engine *e, void *caps, size_t caps_size)
{
   if (!e || !caps || caps_size < sizeof(int32_t)) return fail(VADC_AMD_EINVAL, "get_caps_sized: NULL argument or no room for a field");
   vadc_amd_caps full = {};
   const int rc = vadc_amd_get_caps(e, &full);
   if (rc) return rc;
   memcpy(caps, &full, caps_size < sizeof(full) ? caps_size : sizeof(full));
   return VADC_AMD_OK;
}

extern "C" int vadc_amd_get_caps(const vadc_amd_engine *e, vadc_amd_caps *caps)
{
   if (!e || !caps) return fail(VADC_AMD_EINVAL, "get_caps: NULL argument");
   caps->batch_size_restriction = -1;          // silero.h:39
   caps->is_silero_v5 = e->model == VADC_AMD_MODEL_V5;   // silero.h:40 / onnx_helpers.c:154-156
   // silero.h:41-42 (the C backend: 1536 only); onnx_helpers.c:164-170 for the v4 graph: 512 ... 1536, of which this engine runs the multiples of 256
   const int wmax = e->model == VADC_AMD_MODEL_V5 ? 512 : (e->sample_rate == 8000 ? 768 : kChunk);     // v5: onnx_helpers.c:158-160
   caps->input_size_min = (e->model == VADC_AMD_MODEL_V4 && e->gemm_ok) ? wmax / 3 : wmax;
   caps->input_size_max = wmax;
   caps->input_size_step = caps->input_size_min != wmax ? 64 : 0;       // every multiple of 64 samples (one STFT frame): 16 kHz 512 .. 1536, 8 kHz 256 .. 768
   caps->context_size = e->model == VADC_AMD_MODEL_V5 ? 64 : 0;
   caps->window_samples = e->window;
   caps->sample_rate = e->sample_rate;
   caps->cu_partition_ok = e->cu_partition_usable() ? 1 : 0;
   caps->output_dims = 3;                      // silero.h:43
   caps->output_stride = 2;                    // vadc.c:704-708
   caps->silero_probability_out_index = 1;
   caps->lstm_hidden_size = e->model == VADC_AMD_MODEL_V5 ? 128 : kHidden;
   caps->max_streams = e->max_streams;
   caps->max_chunks_per_call = e->max_chunks;
   caps->device = e->device;
   caps->precision = (e->precision == VADC_AMD_PRECISION_FAST_STFT && e->model != VADC_AMD_MODEL_V4 && !e->gemm_ok) ? VADC_AMD_PRECISION_SPLIT16 : e->precision;
   caps->model_kind = e->model;
   caps->lstm_steps_per_chunk = e->lstm_steps;
   return VADC_AMD_OK;
}

// ---------------------------------------------------------------------------------------------------
// profiling helpers
// ---------------------------------------------------------------------------------------------------
static int drain_events(vadc_amd_engine *e)
{
   for (int k = 0; k < VADC_AMD_KERNEL_COUNT; ++k) {
      for (auto &p : e->pending[k]) {
         HIP_TRY(hipEventSynchronize(p.b), VADC_AMD_EHIP);
         float ms = 0.0f;
         HIP_TRY(hipEventElapsedTime(&ms, p.a, p.b), VADC_AMD_EHIP);
         e->total_ms[k] += ms;
         e->launches[k] += 1;
         e->pool.push_back(p);
      }
      e->pending[k].clear();
   }
   return VADC_AMD_OK;
}

// Profiler ranges (option "roctx"): the counterpart of the reference's Tracy zones (TracyCZoneN in silero_v3.c:72-215, conv.c, lstm.c, transformer.c: one zone per
// stage of the forward pass) for a ROCm timeline -- rocprofv3 --marker-trace shows a call and the launch of each of its kernels as named ranges on the host
// thread, beside the kernel rows.  The marker library is looked up at run time (librocprofiler-sdk-roctx, else libroctx64): libvadc_amd.so keeps linking the HIP
// runtime only, and a host without the library gets VADC_AMD_EINVAL when it switches the option on, not a load failure.
struct Roctx {
   void *lib = nullptr;
   int (*push)(const char *) = nullptr;
   int (*pop)() = nullptr;
   bool load()
   {
      if (push && pop) return true;
      for (const char *name : {"librocprofiler-sdk-roctx.so", "librocprofiler-sdk-roctx.so.1", "libroctx64.so", "libroctx64.so.4"}) {
         lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
         if (!lib) continue;
         push = reinterpret_cast<int (*)(const char *)>(dlsym(lib, "roctxRangePushA"));
         pop = reinterpret_cast<int (*)()>(dlsym(lib, "roctxRangePop"));
         if (push && pop) return true;
         dlclose(lib); lib = nullptr; push = nullptr; pop = nullptr;
      }
      return false;
   }
};
static Roctx g_roctx;
struct RoctxRange {
   bool on;
   RoctxRange(const vadc_amd_engine *e, const char *name) : on(e->roctx != 0 && g_roctx.push != nullptr) { if (on) (void)g_roctx.push(name); }
   ~RoctxRange() { if (on) (void)g_roctx.pop(); }
};

struct KernelTimer {
   vadc_amd_engine *e; int k; hipStream_t st; vadc_amd_engine::EvPair p; bool on;
   RoctxRange range;
   KernelTimer(vadc_amd_engine *e_, int k_, hipStream_t st_) : e(e_), k(k_), st(st_), on(e_->profiling), range(e_, vadc_amd_kernel_name(k_))
   {
      if (!on) return;
      if (e->pool.empty()) { (void)hipEventCreate(&p.a); (void)hipEventCreate(&p.b); }
      else { p = e->pool.back(); e->pool.pop_back(); }
      (void)hipEventRecord(p.a, st);
   }
   ~KernelTimer()
   {
      if (!on) return;
      (void)hipEventRecord(p.b, st);
      e->pending[k].push_back(p);
   }
};

extern "C" int vadc_amd_set_profiling(vadc_amd_engine *e, int enabled)
{
   if (!e) return fail(VADC_AMD_EINVAL, "set_profiling: NULL engine");
   e->profiling = enabled != 0;
   return VADC_AMD_OK;
}

extern "C" int vadc_amd_get_kernel_time(vadc_amd_engine *e, int kernel, int *launches, double *total_ms)
{
   if (!e || kernel < 0 || kernel >= VADC_AMD_KERNEL_COUNT) return fail(VADC_AMD_EINVAL, "get_kernel_time: bad argument");
   HIP_TRY(hipSetDevice(e->device), VADC_AMD_EHIP);
   int rc = drain_events(e);
   if (rc) return rc;
   if (launches) *launches = e->launches[kernel];
   if (total_ms) *total_ms = e->total_ms[kernel];
   return VADC_AMD_OK;
}

extern "C" int vadc_amd_reset_kernel_times(vadc_amd_engine *e)
{
   if (!e) return fail(VADC_AMD_EINVAL, "reset_kernel_times: NULL engine");
   HIP_TRY(hipSetDevice(e->device), VADC_AMD_EHIP);
   int rc = drain_events(e);
   if (rc) return rc;
   for (int k = 0; k < VADC_AMD_KERNEL_COUNT; ++k) { e->launches[k] = 0; e->total_ms[k] = 0.0; }
   return VADC_AMD_OK;
}

extern "C" const char *vadc_amd_kernel_name(int kernel)
{
   static const char *names[VADC_AMD_KERNEL_COUNT] = {"k_frontend", "k_layer1", "k_layer2", "k_layer3", "k_layer4", "k_lstm", "k_lstm_l1", "k_enc234"};
   return (kernel >= 0 && kernel < VADC_AMD_KERNEL_COUNT) ? names[kernel] : "?";
}

static int wait_all_prior_fwd(vadc_amd_engine *e);
static void look_at_trail_recoveries(vadc_amd_engine *e);
extern "C" int vadc_amd_set_option(vadc_amd_engine *e, const char *key, int value)
{
   if (!e || !key) return fail(VADC_AMD_EINVAL, "set_option: NULL argument");
   if (e->model == VADC_AMD_MODEL_V4 && strcmp(key, "encoder") == 0 && value == 3)
      return fail(VADC_AMD_EINVAL, "set_option: %s=%d exists for Silero v3.1 only (the v4 stages carry no split-fp16 GEMMs)", key, value);
   if (e->precision == VADC_AMD_PRECISION_SPLIT16 && value == 3 && (strcmp(key, "encoder") == 0 || strcmp(key, "lstm") == 0))
      return fail(VADC_AMD_EINVAL, "set_option: %s=3 selects fp32 MFMA; the SPLIT16 precision mode runs split-fp16 GEMMs only", key);
   if (e->precision == VADC_AMD_PRECISION_SPLIT16 && value == 1 && strcmp(key, "layer1") == 0)
      return fail(VADC_AMD_EINVAL, "set_option: layer1=1 selects the fp32-MFMA form of the first layer; the SPLIT16 precision mode runs split-fp16 GEMMs only");
   // Every accepted switch (but "graph" itself) changes the launch sequence a captured graph replays: the captured graphs are dropped (after their last
   // replay has finished) -- only once the key and value have been validated, so that a rejected call leaves them alone.
   {
      static const char *const keys[] = {"lstm", "frontend", "encoder", "groups", "window", "defer_join", "lstm_cus", "cu_partition", "h2d_streams", "layer1", "cu_mask_check", "lstm_trail", "fe_xcd"};
      bool known = false;
      for (const char *k : keys) known = known || strcmp(key, k) == 0;
      if (known && !e->graphs.empty()) {
         vadc_amd_engine probe = *e;                        // validate on a copy: same checks, no side effects on the device
         probe.graphs.clear();
         const int rc_probe = vadc_amd_set_option(&probe, key, value);
         probe.pinned.clear();
         if (rc_probe != VADC_AMD_OK) return rc_probe;
         if (e->ev_last_valid && e->last_a) HIP_TRY(hipEventSynchronize(e->last_a), VADC_AMD_EHIP);   // every replay is followed by the record last_a points at
         for (auto &ge : e->graphs) { (void)hipGraphExecDestroy(ge.x); (void)hipGraphDestroy(ge.g); }
         e->graphs.clear();
      }
   }
   if (e->model == VADC_AMD_MODEL_V4 && strcmp(key, "frontend") == 0 && value == 1 && (e->window != kChunk || e->sample_rate != 16000))
      return fail(VADC_AMD_EINVAL, "set_option: the v4 tree front end exists for 1536-sample windows of the 16 kHz branch only");
   if (e->model == VADC_AMD_MODEL_V4 && strcmp(key, "frontend") == 0 && value >= 0 && value <= 1) {
      // v4: 0 = GEMM front end on the matrix cores (default; needs the symmetric basis), 1 = the tree kernel with the v4 geometry
      e->frontend_variant = value;
      return VADC_AMD_OK;
   }
   if (strcmp(key, "lstm") == 0 && (value == 0 || value == 3 || value == 6 || value == 7)) { e->lstm_variant = value; e->lstm_cus = -1; return VADC_AMD_OK; }
   if (strcmp(key, "frontend") == 0 && (value == 0 || value == 1)) { e->frontend_variant = value; return VADC_AMD_OK; }
   if (strcmp(key, "fe_xcd") == 0 && (value == 0 || value == 1)) { e->fe_xcd = value; return VADC_AMD_OK; }
   if (strcmp(key, "pin_host") == 0 && (value == 0 || value == 1)) { e->pin_host = value; return VADC_AMD_OK; }
   if (strcmp(key, "lstm_trail") == 0 && (value == 0 || value == 1)) { e->lstm_trail = value; return VADC_AMD_OK; }
   if (strcmp(key, "trail_fault") == 0 && value >= 0 && value <= 2) { e->trail_fault = value; return VADC_AMD_OK; }
   if (strcmp(key, "trail_wait") == 0 && value >= 1000) { e->trail_wait_limit = value; return VADC_AMD_OK; }
   if (strcmp(key, "roctx") == 0 && (value == 0 || value == 1)) {
      if (value && !g_roctx.load()) return fail(VADC_AMD_EINVAL, "set_option: roctx=1 needs librocprofiler-sdk-roctx.so or libroctx64.so on the library path");
      e->roctx = value;
      return VADC_AMD_OK;
   }
   if (strcmp(key, "overlap_check") == 0 && (value == 1 || value == 2)) { e->overlap_check = value; return VADC_AMD_OK; }
   if (strcmp(key, "lstm_epoch") == 0 && value >= 0 && value <= 2047) { e->lstm_epoch = value; return VADC_AMD_OK; }      // (tests: the epoch's wrap)
   if (strcmp(key, "cu_mask_check") == 0 && value >= 0 && value <= 2) { e->cu_mask_check = value; e->lstm_cus = -1; return VADC_AMD_OK; }
   if (strcmp(key, "encoder") == 0 && (value == 0 || value == 3)) { e->encoder_variant = value; return VADC_AMD_OK; }
   if (strcmp(key, "groups") == 0 && value >= 0 && value <= vadc_amd_engine::kMaxGroups) { e->groups = value; return VADC_AMD_OK; }
   if (strcmp(key, "window") == 0) {
      // samples per chunk.  The reference's C backend takes 1536 only (silero.h:41-42); its onnxruntime path lets the v4 graph take every count in 512 ... 1536
      // (onnx_helpers.c:164-170, --sequence_count vadc.c:743-752).  Served here: every multiple of 64 samples (one STFT frame) in that range -- a count in between would add
      // samples that fill no frame and reach the model through the right reflect pad only: not built.  The multiples of 256
      // (8 / 12 / 16 / 20 / 24 frames) have kernels built for their geometry; a window in between runs the next larger of them -- the front end stages the chunk's
      // own samples and puts the right reflect pad behind them (`nrt`), so that its first frames ARE the window's frames; the surplus steps are masked stage by stage (`tv`)
      if (value == e->window) return VADC_AMD_OK;
      const int wmax = e->sample_rate == 8000 ? 768 : kChunk;     // 8 kHz branch: 256 ... 768 samples = the same 32 ... 96 ms
      if (e->model != VADC_AMD_MODEL_V4 || !e->gemm_ok || e->frontend_variant != 0 || value < wmax / 3 || value > wmax || value % 64 != 0)
         return fail(VADC_AMD_EINVAL, "set_option: window=%d: Silero v3.1 takes 1536-sample chunks only; Silero v4 (GEMM front end) every multiple of 64 in 512 .. 1536 "
                                      "(its 8 kHz branch 256 .. 768)", value);
      { int rc_ = wait_all_prior_fwd(e); if (rc_) return rc_; }
      const int fv = value / 64;
      int ft = (fv + 3) / 4 * 4;
      // 17 .. 20 frames (1088 .. 1280 samples) run the 24-frame geometry as well when its register-resident first stage and the fused stages are there: the front end computes
      // 24 frames instead of 20 (0.31 -> 0.39 ms per 65,536 chunks) and the first stage takes 0.31 instead of 0.48 ms (its per-stage form at 20 frames) -- 1280: 5.23 -> 5.73 M,
      // 1216: 4.98 -> 5.43 M, A/B on one box; from 16 frames down the longer front end costs more than the first stage gains (1024: 5.24 -> 4.76 M).  Same probabilities
      // (tools/v4_windows_parity.py: identical statistics either way)
      if (e->sample_rate == 16000 && fv >= 17 && e->d_l1img && e->d_encv4 && e->layer1_selfcheck != 0) ft = 24;
      e->window = value; e->frames = ft; e->frames_valid = fv;
      { const int t1 = (fv + 1) / 2, t2 = (t1 + 1) / 2; e->lstm_steps = e->stride3() == 2 ? (t2 + 1) / 2 : t2; }      // (the same for ft: every fv in (ft - 4, ft])
      stage_elems_v4(e->frames, e->stride3(), e->stage_elems);
      return VADC_AMD_OK;
   }
   if (strcmp(key, "graph") == 0 && (value == 0 || value == 1)) { e->use_graph = value; return VADC_AMD_OK; }
   if (strcmp(key, "h2d_streams") == 0 && value >= 1 && value <= 4) { e->h2d_parts = value; return VADC_AMD_OK; }
   if (strcmp(key, "layer1") == 0 && (value == 0 || value == 1)) { e->layer1_variant = value; return VADC_AMD_OK; }
   if (strcmp(key, "defer_join") == 0 && (value == 0 || value == 1)) { e->defer_join = value; return VADC_AMD_OK; }
   if (strcmp(key, "lstm_cus") == 0 && value >= 0 && value <= 128 && value % 8 == 0) { e->lstm_cus_forced = value; e->lstm_cus = -1; return VADC_AMD_OK; }
   if (strcmp(key, "cu_partition") == 0 && value >= 0 && value <= 2) { e->cu_partition = value; e->lstm_cus = -1; return VADC_AMD_OK; }
   return fail(VADC_AMD_EINVAL, "set_option: unknown option %s=%d", key, value);
}

extern "C" int vadc_amd_get_option(vadc_amd_engine *e, const char *key, int *value)
{
   if (!e || !key || !value) return fail(VADC_AMD_EINVAL, "get_option: NULL argument");
   if (strcmp(key, "lstm") == 0) *value = e->lstm_variant;
   else if (strcmp(key, "frontend") == 0) *value = e->frontend_variant;
   else if (strcmp(key, "fe_xcd") == 0) *value = e->fe_xcd;
   else if (strcmp(key, "layer1_selfcheck") == 0) *value = e->layer1_selfcheck;
   else if (strcmp(key, "layer1_kernel") == 0) *value = (e->use_l1_regs() || e->use_l1_regs_v4()) ? 0 : 1;      // the form that runs (option "layer1" is the request)
   else if (strcmp(key, "pin_host") == 0) *value = e->pin_host;
   else if (strcmp(key, "lstm_trail") == 0) *value = e->lstm_trail;
   else if (strcmp(key, "kernels_overlap") == 0) *value = e->kernels_overlap ? 1 : 0;
   else if (strcmp(key, "lstm_trail_used") == 0) *value = e->lstm_trail_used;
   else if (strcmp(key, "roctx") == 0) *value = e->roctx;
   else if (strcmp(key, "trail_recoveries") == 0) {           // tiles done again by the REDO launches so far (waits for the calls issued before)
      HIP_TRY(hipSetDevice(e->device), VADC_AMD_EHIP);
      { int rc_ = wait_all_prior_fwd(e); if (rc_) return rc_; }
      look_at_trail_recoveries(e);
      *value = e->trail_recoveries;
   }
   else if (strcmp(key, "lstm_epoch") == 0) *value = e->lstm_epoch;
   else if (strcmp(key, "cu_mask_check") == 0) *value = e->cu_mask_check;
   else if (strcmp(key, "cu_layout_ok") == 0) *value = e->cu_layout_ok ? 1 : 0;
   else if (strcmp(key, "pinned_ranges") == 0) *value = (int)e->pinned.size();
   else if (strcmp(key, "zero_im0") == 0) *value = e->zero_im0 ? 1 : 0;
   else if (strcmp(key, "encoder") == 0) *value = e->encoder_variant;
   else if (strcmp(key, "layer1") == 0) *value = e->layer1_variant;
   else if (strcmp(key, "groups") == 0) *value = e->groups;
   else if (strcmp(key, "graph") == 0) *value = e->use_graph;
   else if (strcmp(key, "window") == 0) *value = e->window;
   else if (strcmp(key, "defer_join") == 0) *value = e->defer_join;
   else if (strcmp(key, "cu_partition") == 0) *value = e->cu_partition;
   else if (strcmp(key, "lstm_cus") == 0) *value = e->lstm_cus < 0 ? 0 : e->lstm_cus;
   else if (strcmp(key, "lstm_kernel") == 0) *value = e->last_lstm_kernel;
   else if (strcmp(key, "lstm_shared") == 0) *value = e->lstm_shared ? 1 : 0;      // (read only) the recurrence's CUs are also in the front end + encoder stream's mask
   else if (strcmp(key, "frontend_kernel") == 0) *value = e->last_frontend_kernel;
   else return fail(VADC_AMD_EINVAL, "get_option: unknown option %s", key);
   return VADC_AMD_OK;
}

// ---------------------------------------------------------------------------------------------------
// the hot path
// ---------------------------------------------------------------------------------------------------
// ---------------------------------------------------------------------------------------------------
// The register-resident first layer (k_layer1_regs, k_layer1_regs_v4) waits for its LDS-DMA pieces with hand-counted s_waitcnt vmcnt(N) over operations the
// compiler cannot see.  The build checks the listing (tools/check_counted_waits.py); every engine also checks the RESULT once, on the device it runs on: three
// probe chunks through that kernel and through the per-layer form (k_layer_mfma, compiler-managed waits).  More than 5e-5 apart: the per-layer form serves
// from then on, with a warning on stderr (get_option "layer1_selfcheck" = 0).
// ---------------------------------------------------------------------------------------------------
static void run_encoder_layers(vadc_amd_engine *e, int first, int last, int n, ItemMap map, int lstm_layout, hipStream_t st, const float *in_stage);
static int layer1_selfcheck(vadc_amd_engine *e)
{
   if (!(e->use_l1_regs() || e->use_l1_regs_v4())) return VADC_AMD_OK;
   const int n = (int)std::min<size_t>(3, e->max_items), T = e->frames;
   const size_t fms = e->max_items * (size_t)kFrames;
   std::vector<float> y((size_t)n * kBins * T), fm((size_t)kBinSplit * n * T, 0.0f);      // (the partial sums of the probe chunks alone: the device array is [kBinSplit][fms])
   for (int c = 0; c < n; ++c)
      for (int b = 0; b < kBins; ++b)
         for (int t = 0; t < T; ++t) {
            const float v = 5.0f + 4.0f * sinf(0.37f * b + 0.61f * t + 1.3f * c) + (b % 7 == 0 ? 3.0f : 0.0f);      // log-magnitude-like values in 0 .. 13
            y[((size_t)c * kBins + b) * T + t] = v;
            fm[(size_t)(b / kBinsPerSplit) * n * T + (size_t)c * T + t] += v;
         }
   hipStream_t st = e->stream;
   HIP_TRY(host_to_device(e, e->d_Y, y.data(), y.size() * sizeof(float), st), VADC_AMD_EHIP);
   for (int sp = 0; sp < kBinSplit; ++sp) HIP_TRY(host_to_device(e, e->d_FM + (size_t)sp * fms, fm.data() + (size_t)sp * n * T, (size_t)n * T * sizeof(float), st), VADC_AMD_EHIP);
   const size_t elems = (size_t)n * e->stage_elems[VADC_AMD_STAGE_LAYER1];
   std::vector<float> a(elems), b(elems);
   for (int form = 0; form < 2; ++form) {
      e->layer1_variant = form;
      run_encoder_layers(e, 0, 0, n, ItemMap{n, 0, n}, 0, st, nullptr);
      HIP_TRY(hipGetLastError(), VADC_AMD_EHIP);
      HIP_TRY(device_to_host(e, (form == 0 ? a : b).data(), e->d_act[0], elems * sizeof(float), st), VADC_AMD_EHIP);
   }
   e->layer1_variant = 0;
   float worst = 0.0f, scale = 1.0f;
   bool finite = true;
   for (size_t i = 0; i < elems; ++i) { worst = std::max(worst, fabsf(a[i] - b[i])); scale = std::max(scale, fabsf(b[i])); finite = finite && std::isfinite(a[i]); }
   if (getenv("VADC_AMD_FORCE_L1_SELFCHECK_FAIL")) worst = 1.0f;      // tests: the fallback path
   if (getenv("VADC_AMD_ABL_SKIP_L1_SELFCHECK")) worst = 0.0f;        // timing-only ablation builds (tools/abl_build.sh: results are wrong by construction) keep the kernel they ablate
   e->layer1_selfcheck = (finite && worst <= 5e-5f * scale) ? 1 : 0;
   if (!e->layer1_selfcheck)
      fprintf(stderr, "vadc_amd: the register-resident first encoder layer disagrees with the per-layer form on the probe chunks (max |d| %.3e): the per-layer form serves this engine\n", worst);
   return VADC_AMD_OK;
}

// ---------------------------------------------------------------------------------------------------
// The CU partition's premise, checked on the device (once per process and device): the rules below (lstm_partition_cus, encoder_cus) were measured on an
// MI355X in SPX mode -- 256 CUs, and a hipExtStreamCreateWithCUMask mask deals its bits to the 8 XCDs in turn: the first 8 k bits are k CUs of EVERY XCD
// (tools/cumask_probe.hip).  On a partitioned
// (CPX / DPX) or differently built part they would be silently wrong -- slow, not incorrect -- so on any other layout the engine runs WITHOUT a
// partition (plain prioritised streams) and says so: caps.cu_partition_ok = 0.
// ---------------------------------------------------------------------------------------------------
// a workgroup that fills a CU's LDS (only one fits per CU) and stays for a while: a grid of these lands one per CU of the stream's mask
__global__ __launch_bounds__(64) void k_probe_hold(unsigned *out, int spin)
{
   __shared__ char big[140 * 1024];
   unsigned xcc;
   asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
   big[threadIdx.x] = (char)xcc;
   const unsigned long long t0 = __builtin_readcyclecounter();
   while (__builtin_readcyclecounter() - t0 < (unsigned long long)spin) { }
   if (threadIdx.x == 0) out[blockIdx.x] = (xcc & 0xf) + (big[7] & 0);
}
// role 0: wait (bounded) for w[0] and note in w[1] whether it came; role 1: set w[0].  System scope: the two may sit on different XCDs.
__global__ __launch_bounds__(64) void k_probe_overlap(unsigned *w, int role, long long ticks)
{
   if (threadIdx.x != 0) return;
   if (role == 1) { __hip_atomic_store(w, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); return; }
   const long long t0 = wall_clock64();
   unsigned seen = 0;
   while (!seen && wall_clock64() - t0 < ticks) { seen = __hip_atomic_load(w, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM); __builtin_amdgcn_s_sleep(32); }
   w[1] = seen;
}
// (a one-bit mask is no probe: every bit then reports XCC 0 -- the id is relative to the XCDs the queue may use; tools/cumask_probe.hip)
// bit 0: the layout holds.  (Which XCD workgroup i of a launch lands on is NOT a property of the layout: it is (start + i) % 8 with a start that differs between
// queues and over time -- tools/xcd_map_probe.hip -- which is why k_lstm_layer's TRAIL form lets every XCD hand out its own tiles.)
static int cu_mask_layout_flags(int device, int n_cus)
{
   static int cache[64];                                    // 0 = unknown, else flags + 4
   if (device >= 0 && device < 64 && cache[device]) return cache[device] - 4;
   bool ok = n_cus == 256;
   unsigned *d = nullptr;
   if (ok && hipMalloc(&d, 256 * sizeof(unsigned)) != hipSuccess) { (void)hipGetLastError(); ok = false; }
   const int words = (n_cus + 31) / 32;
   // grid workgroups on the mask bits [lo, hi); their XCDs into got[]
   auto run = [&](int lo, int hi, int grid, std::vector<unsigned> &got) -> bool {
      std::vector<uint32_t> m(words, 0u);
      for (int cu = lo; cu < hi; ++cu) m[cu / 32] |= 1u << (cu % 32);
      hipStream_t st = nullptr;
      got.assign(256, 99u);
      if (hipExtStreamCreateWithCUMask(&st, (uint32_t)words, m.data()) != hipSuccess) { (void)hipGetLastError(); return false; }
      hipLaunchKernelGGL(k_probe_hold, dim3(grid), dim3(64), 0, st, d, 60000);
      bool good = hipMemcpyAsync(got.data(), d, grid * sizeof(unsigned), hipMemcpyDeviceToHost, st) == hipSuccess && hipStreamSynchronize(st) == hipSuccess;
      if (!good) (void)hipGetLastError();
      (void)hipStreamDestroy(st);
      return good;
   };
   std::vector<unsigned> got;
   // the two shapes the engine uses: the first 16 / 32 mask bits for the recurrence (2 / 4 CUs of every XCD), all the others for the front end + encoder
   for (int taken : {8, 16, 32}) {                          // (8: one half of the smallest partition, see the TRAIL form of k_lstm_layer)
      for (int side = 0; side < (taken == 8 ? 1 : 2) && ok; ++side) {
         const int grid = side == 0 ? taken : n_cus - taken;
         if (!run(side == 0 ? 0 : taken, side == 0 ? taken : n_cus, grid, got)) { ok = false; break; }
         int per[8] = {0, 0, 0, 0, 0, 0, 0, 0};
         for (int i = 0; i < grid; ++i) if (got[i] < 8) ++per[got[i]];
         for (int x = 0; x < 8; ++x) if (per[x] != grid / 8) ok = false;
         if (getenv("VADC_AMD_DEBUG_CUMASK")) fprintf(stderr, "cu_mask_layout: %d bits %s: workgroups per XCD %d %d %d %d %d %d %d %d (expected %d each)\n", taken, side == 0 ? "alone" : "excluded",
                                                      per[0], per[1], per[2], per[3], per[4], per[5], per[6], per[7], grid / 8);
      }
   }
   // bit 1: kernels on two streams run AT THE SAME TIME in this process.  Under a tool that serialises kernel dispatches (rocprofv3 --pmc: one kernel on the
   // device at a time, so that its counters are its own) they do not, and a layer-1 launch that follows the progress of a layer-0 launch (k_lstm_layer's TRAIL
   // form) would wait for a kernel the tool starts only when layer 1 has ended.  Asked of the device, not of the environment: a kernel on one masked stream
   // waits -- for 20 ms at most -- for a word a kernel launched AFTER it on another stream sets.
   bool overlap = false;
   if (ok && d) {
      std::vector<uint32_t> m0(words, 0u), m1(words, 0u);
      m0[0] = 0xffu; m1[0] = 0xff00u;
      hipStream_t s0 = nullptr, s1 = nullptr;
      if (hipExtStreamCreateWithCUMask(&s0, (uint32_t)words, m0.data()) == hipSuccess && hipExtStreamCreateWithCUMask(&s1, (uint32_t)words, m1.data()) == hipSuccess &&
          hipMemsetAsync(d, 0, 2 * sizeof(unsigned), s0) == hipSuccess && hipStreamSynchronize(s0) == hipSuccess) {
         hipLaunchKernelGGL(k_probe_overlap, dim3(1), dim3(64), 0, s0, d, 0, 2000000ll);      // wall_clock64: 100 MHz
         hipLaunchKernelGGL(k_probe_overlap, dim3(1), dim3(64), 0, s1, d, 1, 0ll);
         unsigned seen = 0;
         if (hipStreamSynchronize(s1) == hipSuccess && hipStreamSynchronize(s0) == hipSuccess && hipMemcpy(&seen, d + 1, sizeof(unsigned), hipMemcpyDeviceToHost) == hipSuccess) overlap = seen == 1;
      }
      (void)hipGetLastError();
      if (s0) (void)hipStreamDestroy(s0);
      if (s1) (void)hipStreamDestroy(s1);
      if (getenv("VADC_AMD_DEBUG_CUMASK")) fprintf(stderr, "cu_mask_layout: kernels on two streams %s\n", overlap ? "overlap" : "do NOT overlap (serialised by a tool?)");
   }
   if (d) (void)hipFree(d);
   const int flags = (ok ? 1 : 0) | (overlap ? 2 : 0);
   if (device >= 0 && device < 64) cache[device] = flags + 4;
   return flags;
}

static int check_trail_error(vadc_amd_engine *e, const char *where);
static int check_shape(vadc_amd_engine *e, int n_streams, int n_chunks, const char *who)
{
   if (!e) return fail(VADC_AMD_EINVAL, "%s: NULL engine", who);
   { int rc_ = check_trail_error(e, who); if (rc_) return rc_; }      // (a memory read: a call whose streams are in an undefined state is refused at its head)
   if (n_streams <= 0 || n_chunks <= 0) return fail(VADC_AMD_EINVAL, "%s: n_streams=%d n_chunks=%d must be positive", who, n_streams, n_chunks);
   if (n_streams > e->max_streams) return fail(VADC_AMD_EINVAL, "%s: n_streams=%d exceeds max_streams=%d", who, n_streams, e->max_streams);
   if ((size_t)n_streams * n_chunks > e->max_items)
      return fail(VADC_AMD_EINVAL, "%s: %d x %d chunks exceed the workspace (%zu)", who, n_streams, n_chunks, e->max_items);
   // the encoder -> LSTM hand-off is stored in tiles of 16 streams: ceil(n_streams / 16) x n_chunks blocks must fit what
   // ceil(max_streams / 16) x max_chunks_per_call allocated (few streams x very many chunks would otherwise overrun it)
   if ((size_t)((n_streams + kLstmTile - 1) / kLstmTile) * n_chunks > e->x_tile_chunks)
      return fail(VADC_AMD_EINVAL, "%s: %d streams x %d chunks need %zu (16-stream tile, chunk) blocks, the workspace holds %zu (max_streams=%d, max_chunks_per_call=%d)",
                  who, n_streams, n_chunks, (size_t)((n_streams + kLstmTile - 1) / kLstmTile) * n_chunks, e->x_tile_chunks, e->max_streams, e->max_chunks);
   return VADC_AMD_OK;
}

// Silero v4: the first stage (K = 1 form) takes the magnitude half of its input from Y = log(1 + 2^20 m) instead of a second array

// CUs a persistent grid on the front end + encoder stream may count on: the LSTM chain's workgroups keep theirs for a whole call, and an
// 8-wave workgroup of k_enc_fused (216 registers, 132 KB of LDS) never fits beside one -- a workgroup sent there would wait for the chain
static int encoder_cus(const vadc_amd_engine *e, hipStream_t st)
{
   const int taken = (st == e->sA && e->lstm_cus > 0) ? e->lstm_cus : 0;
   if (e->n_cus == 256 && taken > 0) {
      // A CU-mask bit i selects a CU of XCD i % 8, and inside an XCD consecutive bits rotate over its 4 shader engines (tools/cumask_probe.hip).  A grid
      // of one workgroup per CU is dealt to the XCDs and their shader engines in turn, so it must not be larger than 32 x the CUs the most depleted
      // shader engine keeps: with 48 CUs taken (6 per XCD: shader engines keep 6, 6, 7, 7) a grid of 208 sent a seventh workgroup to engines with six
      // CUs, and both persistent kernels took two rounds (320 streams: k_layer1 0.27 instead of 0.15 ms)
      const int per_xcd = (taken + 7) / 8, per_se = (per_xcd + 3) / 4;
      return 32 * (8 - per_se) > 8 ? 32 * (8 - per_se) : 8;
   }
   return e->n_cus - taken > 8 ? e->n_cus - taken : 8;
}

// lstm_layout: the last layer writes the LSTM-native tile layout (hot path) instead of [n][64][7] (stage taps)
// `in_stage`: input of layer `first` when it is not the engine's own buffer (stage taps fed from the host)
static void run_encoder_layers(vadc_amd_engine *e, int first, int last, int n, ItemMap map, int lstm_layout, hipStream_t st) { run_encoder_layers(e, first, last, n, map, lstm_layout, st, nullptr); }
static void run_encoder_layers(vadc_amd_engine *e, int first, int last, int n, ItemMap map, int lstm_layout, hipStream_t st, const float *in_stage)
{
   for (int l = first; l <= last; ++l) {
      const float *in = (l == first && in_stage) ? in_stage : ((l == 0) ? e->d_Y : e->d_act[l - 1]);
      if (l >= 1 && e->use_enc_fused() && !(last == 3 && lstm_layout == 1)) {      // (fp32 LSTM tiles: the per-layer kernel writes those)
         // layers 2..4 (here: l .. last) in one launch; intermediate layer outputs are written only when they are what the caller asked for
         KernelTimer t(e, VADC_AMD_KERNEL_ENC234, st);
         EncFusedArgs a;
         a.in = in; a.imgA = e->d_encA; a.imgB = e->d_encB; a.scratch = e->d_enc_scratch;
         a.out = e->d_act[3];
         a.tap2 = last == 1 ? e->d_act[1] : nullptr;
         a.tap3 = last == 2 ? e->d_act[2] : nullptr;
         a.tap4 = (last == 3 && lstm_layout == 0) ? e->d_act[3] : nullptr;
         a.n_chunks = n; a.first = l + 1; a.last = last + 1; a.map = map;
         launch_enc_fused(a, encoder_cus(e, st), st);
         return;
      }
      if (l == 1 && last == 3 && lstm_layout == 2 && !in_stage && e->use_enc_fused_v4()) {      // Silero v4: stages 2-4 in one launch
         KernelTimer t(e, VADC_AMD_KERNEL_ENC234, st);
         EncV4Args a;
         a.in = in; a.img = e->d_encv4; a.out = e->d_act[3]; a.n_chunks = n; a.map = map;
         a.t1_pitch = (e->frames + 1) / 2; a.t1 = (e->frames_valid + 1) / 2; a.s3 = e->stride3(); a.ts = e->lstm_steps;
         launch_enc_fused_v4(a, encoder_cus(e, st), st);
         return;
      }
      KernelTimer t(e, VADC_AMD_KERNEL_LAYER1 + l, st);
      if (l == 0 && e->use_l1_regs_v4()) {
         L1RegsArgs a;
         a.y = in; a.fm = e->d_FM; a.fm_stride = e->max_items * kFrames; a.img = e->d_l1img; a.out = e->d_act[0]; a.n_chunks = n; a.map = map;
         a.tv = e->padded_window() ? e->frames_valid : 0;
         launch_layer1_regs_v4(a, encoder_cus(e, st), st);
         continue;
      }
      if (l == 0 && e->use_l1_regs()) {
         L1RegsArgs a;
         a.y = in; a.fm = e->d_FM; a.fm_stride = e->max_items * kFrames; a.img = e->d_l1img; a.out = e->d_act[0]; a.n_chunks = n; a.map = map;
         launch_layer1_regs(a, encoder_cus(e, st), st);
         continue;
      }
      if (e->model == VADC_AMD_MODEL_V4) launch_layer_v4(l, in, e->d_FM, e->lwm[l], e->d_act[l], n, map, (l == 3) ? lstm_layout : 0, e->max_items * kFrames, st, e->frames, e->stride3(), e->frames_valid);
      else                         launch_layer_mfma(l, in, e->d_FM, e->lwm[l], e->d_act[l], n, map, (l == 3) ? lstm_layout : 0, e->max_items * kFrames, st);
   }
}

// which front-end kernel serves this input: 0 = k_frontend_sym, 1 = k_frontend_fl, 2 = k_frontend_gemm, 3 = k_frontend (v4 tree)
static int pick_frontend(const vadc_amd_engine *e, const void *d_in)
{
   if (e->use_gemm_frontend()) return 2;
   if (e->model == VADC_AMD_MODEL_V4) return 3;
   return (e->sym_ok && e->frontend_variant == 0 && (reinterpret_cast<uintptr_t>(d_in) & 15) == 0) ? 0 : 1;
}

// The front end and the four encoder layers of one chunk group, enqueued on `st` (pure kernel launches: capturable).
template <typename T>
static void run_front_and_encoder(vadc_amd_engine *e, const T *d_in, int n, ItemMap map, int lstm_kernel, hipStream_t st)
{
   {
      KernelTimer t(e, VADC_AMD_KERNEL_FRONTEND, st);
      const size_t fms = e->max_items * kFrames;
      const int fk = pick_frontend(e, d_in);
      if (fk == 2) {
         const int geo = e->model == VADC_AMD_MODEL_V4 ? e->v4_geo() : 0;
         const int nrt = (e->model == VADC_AMD_MODEL_V4 && e->padded_window()) ? e->window : 0;      // a window between the built ones: the chunk's own sample count (its stride)
         // s16 input: the second form (32x32x16 MFMAs, one persistent workgroup per CU the stream may count on, pipelined across column tiles); f32 input: the first form.
         // No magnitude array on the hot path: the v4 first stage recovers the magnitudes from Y (0.8 GB per 65,536 chunks not written and not read)
         if (sizeof(T) == 2) launch_frontend_gemm2_s16(reinterpret_cast<const int16_t *>(d_in), e->d_afrag2, e->d_nyq2, e->d_Y, e->d_FM, fms, n, map, encoder_cus(e, st), st, geo, nrt);
         else                launch_frontend_gemm_f32(reinterpret_cast<const float *>(d_in), e->d_afrag, e->d_nyq, e->d_Y, nullptr, e->d_FM, fms, n, map, e->n_cus, st, geo, nrt);
      } else if (fk == 3) {
         if (sizeof(T) == 2) launch_frontend_v4_s16(reinterpret_cast<const int16_t *>(d_in), e->d_basis, e->d_Y, e->d_MAG, e->d_FM, fms, n, map, st);
         else                launch_frontend_v4_f32(reinterpret_cast<const float *>(d_in), e->d_basis, e->d_Y, e->d_MAG, e->d_FM, fms, n, map, st);
      } else if (fk == 0) {
         // bit-exact tree for bins 0..32, the other 96 bins from the basis' symmetries (kernels_frontend.hip)
         if (sizeof(T) == 2) launch_frontend_sym_s16(reinterpret_cast<const int16_t *>(d_in), e->d_basis, e->d_Y, e->d_FM, fms, n, map, 0, st, e->fe_xcd, e->zero_im0);
         else                launch_frontend_sym_f32(reinterpret_cast<const float *>(d_in), e->d_basis, e->d_Y, e->d_FM, fms, n, map, 0, st, e->fe_xcd, e->zero_im0);
      } else {
         // any basis, any alignment: the full tree for all 129 bins
         if (sizeof(T) == 2) launch_frontend_fl_s16(reinterpret_cast<const int16_t *>(d_in), e->d_basis, e->d_Y, e->d_FM, fms, n, map, 0, st);
         else                launch_frontend_fl_f32(reinterpret_cast<const float *>(d_in), e->d_basis, e->d_Y, e->d_FM, fms, n, map, 0, st);
      }
   }
   run_encoder_layers(e, 0, 3, n, map, lstm_kernel >= 6 ? 2 : 1, st);      // 2: split-fp16 tiles for k_lstm_wavefront_h3 / k_lstm_layer, 1: fp32 tiles for the fp32 kernel
}

// Cost model shared by the two scheduling decisions below (measured on MI355X): microseconds per recurrence slot of
// one stream tile, and whole-chip front-end + encoder time per chunk.
// The two measured constants of the partition rules, taken on an MI355X (256 CUs, 2.4 GHz peak): a slot of the recurrence is a latency (it scales with the clock),
// the front end + encoder's time per chunk a throughput (clock x CUs) -- on a part that reports another clock both are scaled, so that the rules compare like with like
// (the partition itself is only used on the CU layout the create-time check knows: cu_mask_layout_flags)
static double lstm_slot_us(const vadc_amd_engine *e, int lk) { return (lk == 7 ? 0.75 : (lk == 6 ? 1.6 : 3.9)) * e->clock_scale; }
static double enc_us_per_chunk(const vadc_amd_engine *e)
{
   const double cu_scale = e->n_cus > 0 ? 256.0 / e->n_cus : 1.0;
   if (e->model == VADC_AMD_MODEL_V4) return 0.022 * e->clock_scale * cu_scale;
   return (e->use_gemm_frontend() ? 0.022 : (e->sym_ok && e->frontend_variant == 0 ? 0.033 : 0.075)) * e->clock_scale * cu_scale;      // round 3: layers 2-4 fused (0.010 -> 0.0046 us per chunk)
}

// LSTM kernel for this call: option "lstm" 0 = auto.  3 = k_lstm_wavefront_fused (fp32 MFMA) when an LSTM weight does not fit fp16's range, or when
// asked for; otherwise split-fp16 operands on the fp16 matrix pipe at fp32 accuracy: 6 = k_lstm_wavefront_h3 (one workgroup per 16-stream tile),
// 7 = the layer-major form (k_lstm_layer: layer 0 and layer 1 as two launches on two CU sets, pipelined over calls / chunk groups) while the
// recurrence would otherwise be the longer of the concurrent streams -- few stream tiles: its time per call is steps x slot whatever the stream
// count, the front end + encoder's grows with it -- and 2 x tiles workgroups still get a CU each.
// 6 and 7 are BIT-identical (same MFMAs in the same k order per gate row, pinned contraction in the cell update, the same decoder summation tree:
// tests/test_gpu_parity.py::test_lstm_variants_agree), so the choice may depend on the call's shape without a stream's bits depending on how its
// chunks are cut into calls.  A small call that stays on the caller's stream takes 6: there the two layer launches would run one after the other.
static int resolve_lstm(const vadc_amd_engine *e, int n_streams, bool forked = true, int n_chunks = 1 << 20)
{
   if (e->lstm_variant == 3 || !e->lstm_h3_ok) return 3;
   if (e->lstm_variant >= 6) return e->lstm_variant;
   if (!forked) return 6;
   const int tiles = (n_streams + kLstmTile - 1) / kLstmTile;
   // Tree front end (Silero v3.1), round-3 sweep (profiles/EXPERIMENTS.md item 7): the layer-major pair on a small partition of its own is the best or within 1 % of it
   // up to half a chip of tiles (2048 streams), whether or not the chain is the critical path; beyond, one workgroup per tile beside the next call's front end
   // (round 4: calls of one or two chunks per stream over more tiles than that -- the north star's serving shape, 10,240 streams x 1 chunk -- take the pair as well: a
   // workgroup of k_lstm_wavefront_h3 loads and splits both layers' weights (128 KB) for 7 slots of work and holds half a CU's registers meanwhile; 4096 x 1 2.18 -> 2.22 M,
   // 10,240 x 1 2.82 -> 2.92 M, 16,384 x 1 2.90 -> 3.15 M audio-s/s, tools/lstm_variant_sweep.py; at 4096 x 16 the single kernel stays ahead, 3.55 against 3.52 M)
   if (!e->use_gemm_frontend()) return (tiles <= e->n_cus / 2 || n_chunks <= 2) ? 7 : 6;
   // GEMM front end.  Silero v4, round 5 (tools/v4_partition_sweep.py): with k_frontend_gemm2, the ring in the first stage and stages 2 - 4 in one launch the front end + encoder
   // stream is 25 % faster than when round 3 found one workgroup per tile (6) 1 % ahead here, and the recurrence became the step between 256 and 2048 streams: the layer-major
   // pair on CUs of its own is ahead from 16 tiles up -- 256 x 96 5.86 -> 6.57 M, 320 x 96 5.71 -> 6.46, 640 x 32 5.36 -> 6.54, 1664 x 32 6.78 -> 7.12; within 1 % at 512, 896, 1024, 2048
   if (e->model == VADC_AMD_MODEL_V4 && tiles >= 16 && tiles <= e->n_cus / 2) return 7;
   if (tiles >= 20 && tiles <= e->n_cus / 2) return 6;      // (Silero v3.1 in FAST_STFT precision: round 3's rule)
   const bool chain_critical = e->lstm_steps * lstm_slot_us(e, 6) > 0.5 * n_streams * enc_us_per_chunk(e);
   return (chain_critical && 2 * tiles <= e->n_cus / 2) ? 7 : 6;
}

// How many CUs the LSTM chain gets (0 = no partition).  Its workgroups (one per 16-stream tile, two with k_lstm_pipe) are latency-bound -- a slot
// costs the same whether a CU hosts one workgroup or takes several in turn -- while the front end + encoder scale with the CUs they are given:
// the chain must never wait for a CU behind a front-end grid that fills the machine, so while it needs few CUs it gets CUs of its own.
// *shared = true: the chain's CUs stay in the other stream's mask too (see ensure_pipeline_streams)
// lk = the LSTM kernel run_device resolved for THIS call (its chunk count and forked-ness included: one decision, not two)
static int lstm_partition_cus(const vadc_amd_engine *e, int n_streams, bool *shared, int lk)
{
   *shared = e->cu_partition == 2;
   if (!e->cu_partition || !e->cu_partition_usable()) return 0;
   if (e->lstm_cus_forced > 0) return e->lstm_cus_forced;      // option "lstm_cus" (experiments)
   const int lstm_wgs = (lk == 7 ? 2 : 1) * ((n_streams + 15) / 16);
   // measured (v3.1, audio-s/s, partition vs none): 512 streams 730 K vs 589 K, 1024: 790 K vs 722 K, 2048: 786 K vs 810 K,
   // 4096: 806 K vs 895 K -- with more than n_cus/2 tiles the chain is throughput work and gets the whole chip
   // Round 3, tree front end (v3.1), from a sweep over 64 .. 4096 streams (profiles/EXPERIMENTS.md item 7): with more than 16 stream tiles the layer-major pair gets
   // 32 CUs of its OWN (16 per layer; the mask's 4 CUs per XCD, one per shader engine) and its workgroups share them -- two co-resident, the rest in turn: a
   // slot then costs 1.3 - 3 us instead of 0.75, which the chain can afford as soon as the encoder's time per call exceeds it (320 streams: 3.10 -> 3.21 M,
   // 832: 2.48 -> 3.33 M, 1280: 2.94 -> 3.40 M, 2048: 2.96 -> 3.43 M).  Larger or shared partitions lost everywhere: the persistent encoder kernels scale
   // with the CUs they keep, and a chain that shares its CUs with the front end slows 2 - 4 x.  Beyond half a chip of tiles: no partition.
   // Silero v4 (GEMM front end): the same 32 CUs between 20 tiles and half a chip -- 768 x 32 4.34 -> 4.57 M, 1024 x 32 4.55 -> 4.74 M, 2048 x 32 4.30 -> 5.01 M;
   // at 4096 streams it keeps no partition and back-to-back calls (5.98 M; 3.15 M with the partition).
   if (e->cu_partition == 1) {
      const int tiles = (n_streams + 15) / 16;
      if (!e->use_gemm_frontend() && tiles > e->n_cus / 2) return 0;
      // Silero v4 (round 5, the same sweep): the pair on 32 CUs of its own from 16 tiles up; 17 .. 20 tiles (272 .. 320 streams) on 64, a CU per tile and layer -- on 2 x 16
      // CUs the second workgroup of a CU waits for the first (288 x 96: 5.79 M against 6.39 M, 320 x 96: 6.17 against 6.46)
      if (e->model == VADC_AMD_MODEL_V4 && e->use_gemm_frontend() && lk == 7 && tiles >= 16 && tiles <= e->n_cus / 2) { *shared = false; return (tiles >= 17 && tiles <= 20) ? 64 : 32; }
      if (tiles >= 20 && tiles <= e->n_cus / 2) { *shared = false; return 32; }      // (17 .. 19 tiles: two co-resident chains would be the step -- 288 streams 2.96 M against 3.08 M with 2 x 24 CUs)
   }
   if (lstm_wgs > e->n_cus / 2) return 0;
   const double slot_us = lstm_slot_us(e, lk), per_chunk_us = enc_us_per_chunk(e);
   // SHARED partition: when every workgroup can have a CU of its own and the chain then has slack (<= 0.7 of the other stream's time), the
   // chain is pinned to those CUs but the front end + encoder stream keeps the WHOLE chip in its mask: its workgroups fill what the
   // resident LSTM workgroup leaves of those CUs, no shader engine is a CU short, and the chain is not slowed measurably
   // (round 1, 256 streams: 1.038 M -> 1.089 M, chain 1.13 -> 1.14 ms; 512: 1.092 -> 1.140 M; 1024 on 64 CUs: 1.074 -> 1.151 M).  Not with
   // several workgroups per CU (1024 streams on 24 shared CUs: chain 1.13 -> 2.76 ms), not when the chain is the critical path (128
   // streams: 955 K -> 943 K): then the masks are DISJOINT.
   // whole groups of 8 CUs PER STREAM: the layer-major form splits the partition in two halves (layer 0 / layer 1), and a half of 12 or 20 CUs
   // put two workgroups on one CU while another idled (192 / 320 streams: layer 1's chain 1.1 - 1.2 ms instead of 0.6)
   const int tiles_ = (n_streams + 15) / 16;
   const int w1 = lk == 7 ? 2 * ((tiles_ + 7) / 8 * 8) : (lstm_wgs + 7) / 8 * 8;
   // (round 3: on shared CUs the chain runs 2.0 - 2.3 x slower than alone -- 256 streams: layer chains 0.51 / 0.60 -> 0.84 / 0.90 ms and the step 0.81 ->
   // 0.91 ms -- so sharing needs that much slack, not the 1.6 x of the round-2 kernels)
   if (e->cu_partition == 1 && 2.3 * e->lstm_steps * slot_us <= n_streams * per_chunk_us) *shared = true;
   // Silero v5: the recurrence's workgroup (8 waves x 205 VGPRs) needs a CU to itself; on a shared CU the encoder grid's workgroups keep taking each
   // other's place and it only starts when that grid has drained (measured: encoder 1.58 + recurrence 0.58 ms back to back)
   if (e->model == VADC_AMD_MODEL_V5 && e->cu_partition != 2) *shared = false;
   return w1;
}

// Streams A (front end + encoder) and B (LSTM) of the chunk-group pipeline.  The LSTM of a small batch is
// latency-bound on a handful of workgroups (16 streams each) while the front end wants every CU, and a
// front-end grid that already fills the machine would keep the LSTM workgroups waiting for a free CU.  So
// when the LSTM needs few CUs the two streams get DISJOINT CU masks (hipExtStreamCreateWithCUMask): the
// LSTM chain owns `want` CUs outright and runs truly concurrently with the next group's front end.
static int ensure_pipeline_streams(vadc_amd_engine *e, int n_streams, int lk)
{
   bool shared = false;
   int want = lstm_partition_cus(e, n_streams, &shared, lk);
   const bool split = lk == 7;                                  // layer-major LSTM: stream B = layer 0 on one half of the partition, stream C = layer 1 on the other
   if (e->lstm_cus == want && e->lstm_shared == shared && e->streams_split == split && e->sA && e->sB && e->sC) return VADC_AMD_OK;
   // drain the old streams through the events their last work is marked with (every call ends each stream it used with a record that one of
   // last_a / last_b / last_c aliases), not with hipStreamSynchronize on the CU-masked streams themselves (see vadc_amd_destroy)
   if (e->ev_last_valid)
      for (hipEvent_t ev : {e->last_a, e->last_b, e->last_c}) if (ev) HIP_TRY(hipEventSynchronize(ev), VADC_AMD_EHIP);
   for (hipStream_t *ps : {&e->sA, &e->sB, &e->sC})
      if (*ps) { (void)hipStreamDestroy(*ps); *ps = nullptr; }
   bool masked = false;
   if (want > 0) {
      const int words = (e->n_cus + 31) / 32;
      const int wb = split ? want / 2 : want;
      std::vector<uint32_t> mb(words, 0u), mc(words, 0u), ma(words, 0u);
      for (int cu = 0; cu < e->n_cus; ++cu) (cu < wb ? mb : (cu < want ? mc : ma))[cu / 32] |= 1u << (cu % 32);
      if (!split) mc = mb;
      if (shared) for (int cu = 0; cu < want; ++cu) ma[cu / 32] |= 1u << (cu % 32);   // the LSTM keeps its CUs, the other stream may use them too
      hipError_t ea = hipExtStreamCreateWithCUMask(&e->sA, (uint32_t)words, ma.data());
      hipError_t eb = (ea == hipSuccess) ? hipExtStreamCreateWithCUMask(&e->sB, (uint32_t)words, mb.data()) : ea;
      hipError_t ec = (eb == hipSuccess) ? hipExtStreamCreateWithCUMask(&e->sC, (uint32_t)words, mc.data()) : eb;
      masked = (ea == hipSuccess && eb == hipSuccess && ec == hipSuccess);
      if (!masked) {
         (void)hipGetLastError();
         for (hipStream_t *ps : {&e->sA, &e->sB, &e->sC}) if (*ps) { (void)hipStreamDestroy(*ps); *ps = nullptr; }
      }
   }
   if (!masked && e->cu_partition != 0 && e->cu_partition_usable()) {      // ("cu_partition" = 0 is the caller's word for plain streams everywhere)
      // no partition: the front end + encoder stream still gets a CU mask -- a full one.  A masked stream has a hardware queue of its own; plain streams are dealt onto a
      // few shared ones, and behind some predecessors in the same process stream A and the recurrence's stream ended up on one: the recurrence of call k then started when
      // call k + 1's front end had ENDED and ran beside its persistent kernels (10,240 x 1: 2.95 -> 1.77 M, 16,384 x 1: 3.25 -> 2.6 M, every time; tools/queue_probe.py).
      // The recurrence's streams stay plain streams of the highest priority: masked as well they lose it (4096 x 16: 3.62 -> 3.51 M)
      want = 0;
      const int words = (e->n_cus + 31) / 32;
      std::vector<uint32_t> all(words, 0u);
      for (int cu = 0; cu < e->n_cus; ++cu) all[cu / 32] |= 1u << (cu % 32);
      hipError_t ea = hipExtStreamCreateWithCUMask(&e->sA, (uint32_t)words, all.data());
      // only stream A: the recurrence's streams keep their priority (all three masked: 4096 x 16 -3 %).  NOTE: a CU-masked stream is created with
      // hipStreamDefault flags -- it synchronises with the legacy NULL stream, which callers therefore should not issue from (include/vadc_amd.h)
      int lo = 0, hi = 0;
      (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
      const hipError_t eb = (ea == hipSuccess) ? hipStreamCreateWithPriority(&e->sB, hipStreamNonBlocking, hi) : ea;
      const hipError_t ec = (eb == hipSuccess) ? hipStreamCreateWithPriority(&e->sC, hipStreamNonBlocking, hi) : eb;
      masked = (ea == hipSuccess && eb == hipSuccess && ec == hipSuccess);
      if (!masked) {
         (void)hipGetLastError();
         for (hipStream_t *ps : {&e->sA, &e->sB, &e->sC}) if (*ps) { (void)hipStreamDestroy(*ps); *ps = nullptr; }
      }
   }
   if (!masked) {
      want = 0;
      HIP_TRY(hipStreamCreateWithFlags(&e->sA, hipStreamNonBlocking), VADC_AMD_EHIP);
      int lo = 0, hi = 0;
      (void)hipDeviceGetStreamPriorityRange(&lo, &hi);          // hi = numerically lowest = highest priority
      HIP_TRY(hipStreamCreateWithPriority(&e->sB, hipStreamNonBlocking, hi), VADC_AMD_EHIP);
      HIP_TRY(hipStreamCreateWithPriority(&e->sC, hipStreamNonBlocking, hi), VADC_AMD_EHIP);
   }
   e->lstm_cus = want;
   e->lstm_shared = shared && want > 0;
   e->streams_split = split;
   e->ev_b_valid[0] = e->ev_b_valid[1] = false; // the old streams were drained above
   e->ev_c_valid[0] = e->ev_c_valid[1] = false;
   e->last_on_valid = false;                    // a new stream may get a destroyed one's handle
   return VADC_AMD_OK;
}

static int pick_groups(const vadc_amd_engine *e, int n_chunks)
{
   int g = e->groups;
   // auto: a call whose caller waits for it (strict join, the synchronous host entry points) overlaps its own LSTM with its later groups' front end +
   // encoder; with deferred joins consecutive CALLS overlap instead, and whole-call launches fill the chip better (256 x 96: 2.12 M with 4 groups, 2.47 M with 1)
   if (g <= 0) g = e->defer_join ? 1 : (n_chunks >= 16 ? 4 : (n_chunks >= 4 ? 2 : 1));
   if (g > n_chunks) g = n_chunks;
   return g < 1 ? 1 : g;
}

// hipGraph replay (option "graph"): a KERNEL SEQUENCE -- the whole small call, or the front end + encoder of one chunk group of a forked
// call -- is captured the first time it is issued with a given signature and replayed as one hipGraphLaunch afterwards.  Only kernels are
// inside a graph; the fork / join and the call-to-call ordering (events below) stay outside, exactly as for eager launches, so replays of
// consecutive steps overlap the same way eager steps do (step k+1's front end + encoder under step k's LSTM chain) instead of serialising.
// Profiling (per-kernel events) needs eager launches, so it bypasses the graphs.
// the LSTM + decoder of chunks [c0, c0 + cg) on ONE stream: one kernel for both layers, or (lk == 7 asked for on a single stream) the two layer launches in turn
static void launch_lstm_on(vadc_amd_engine *e, int lk, float *d_probs, int n_streams, int n_chunks, int c0, int cg, hipStream_t st)
{
   if (lk == 7) {
      { KernelTimer t(e, VADC_AMD_KERNEL_LSTM, st); launch_lstm_layer(0, e->d_act[3], e->d_h0pair[e->xpar], e->lstm, e->d_h, e->d_c, d_probs, n_streams, n_chunks, c0, cg, st, e->model, e->lstm_steps, nullptr, 0, nullptr, 0, nullptr, nullptr, 0); }
      {
         KernelTimer t(e, VADC_AMD_KERNEL_LSTM_L1, st);
         launch_lstm_layer(1, e->d_act[3], e->d_h0pair[e->xpar], e->lstm, e->d_h, e->d_c, d_probs, n_streams, n_chunks, c0, cg, st, e->model, e->lstm_steps, nullptr, 0, nullptr, 0, nullptr, nullptr, 0);
      }
      return;
   }
   KernelTimer t(e, VADC_AMD_KERNEL_LSTM, st);
   launch_lstm(lk, e->d_act[3], e->lstm, e->d_h, e->d_c, d_probs, n_streams, n_chunks, c0, cg, st, e->model, e->lstm_steps);
}

// call-to-call ordering (see the engine's last_a / last_b / last_c): `st` continues after the last work of that kind, wherever it ran
static void wait_last(vadc_amd_engine *e, hipStream_t st, hipEvent_t ev, hipStream_t on)
{
   if (!e->ev_last_valid || !ev) return;
   // the same in-order stream -- for the engine's own streams only: a caller may have destroyed its stream and been handed the same handle again
   if (e->last_on_valid && on == st && (st == e->sA || st == e->sB || st == e->sC)) return;
   (void)hipStreamWaitEvent(st, ev, 0);
}
static void wait_last_all(vadc_amd_engine *e, hipStream_t st)
{
   wait_last(e, st, e->last_a, e->last_a_on);
   if (e->last_b != e->last_a) wait_last(e, st, e->last_b, e->last_b_on);
   if (e->last_c != e->last_a && e->last_c != e->last_b) wait_last(e, st, e->last_c, e->last_c_on);
}
// a call that ran entirely on `st`: one record marks all three
static void record_last_on(vadc_amd_engine *e, hipStream_t st)
{
   (void)hipEventRecord(e->ev_last, st);
   e->last_a = e->last_b = e->last_c = e->ev_last;
   e->last_a_on = e->last_b_on = e->last_c_on = st;
   e->ev_last_valid = true; e->last_on_valid = true;
}

struct SeqKey { const void *in; float *out; int S, C, elem, G, gi, xp, lk, fe; };
template <typename F>
static int launch_sequence(vadc_amd_engine *e, const SeqKey &k, hipStream_t st, F &&enqueue)
{
   if (!e->use_graph || e->profiling) { enqueue(); return VADC_AMD_OK; }
   for (auto &ge : e->graphs)
      if (ge.in == k.in && ge.out == k.out && ge.S == k.S && ge.C == k.C && ge.elem == k.elem && ge.G == k.G && ge.gi == k.gi && ge.xp == k.xp &&
          ge.lk == k.lk && ge.fe == k.fe) {
         HIP_TRY(hipGraphLaunch(ge.x, st), VADC_AMD_EHIP);
         return VADC_AMD_OK;
      }
   vadc_amd_engine::GraphEntry ge{k.in, k.out, k.S, k.C, k.elem, k.G, k.gi, k.xp, k.lk, k.fe, nullptr, nullptr};
   HIP_TRY(hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed), VADC_AMD_EHIP);
   enqueue();
   hipError_t he = hipStreamEndCapture(st, &ge.g);
   if (he != hipSuccess) return fail(VADC_AMD_EHIP, "hipStreamEndCapture failed: %s", hipGetErrorString(he));
   HIP_TRY(hipGraphInstantiate(&ge.x, ge.g, nullptr, nullptr, 0), VADC_AMD_EHIP);
   if (e->graphs.size() >= 32) {                           // evict the oldest signature; its last replay may still be in flight
      if (e->ev_last_valid && e->last_a) (void)hipEventSynchronize(e->last_a);   // every replay is followed by the record last_a points at (not a cached stream handle: the caller may have destroyed that stream)
      (void)hipGraphExecDestroy(e->graphs[0].x); (void)hipGraphDestroy(e->graphs[0].g); e->graphs.erase(e->graphs.begin());
   }
   e->graphs.push_back(ge);
   HIP_TRY(hipGraphLaunch(ge.x, st), VADC_AMD_EHIP);
   return VADC_AMD_OK;
}

// The whole hot path for n_streams x n_chunks chunks; asynchronous on `st`.  No allocation, no host sync.
template <typename T>
static int run_device(vadc_amd_engine *e, const T *d_in, int n_streams, int n_chunks, float *d_probs, hipStream_t st)
{
   char rname[64];
   if (e->roctx) snprintf(rname, sizeof(rname), "vadc_amd_run_device [%d streams x %d chunks]", n_streams, n_chunks);
   RoctxRange call_range(e, rname);
   if (e->use_gemm_frontend() && (reinterpret_cast<uintptr_t>(d_in) & 15))
      return fail(VADC_AMD_EINVAL, "run: the GEMM front end stages the input with 16-byte loads; the device buffer must be 16-byte aligned");
   if (e->model == VADC_AMD_MODEL_V5) {
      // Silero v5 shapes: encoder + LSTM input projection (+ the streams' new context), then LSTM(128) + decoder.  Small calls run on the caller's
      // stream; larger ones fork like the other models: stream A = encoder, stream B = the recurrence, so that the next call's encoder runs beside it
      const bool fp32 = e->lstm_variant == 3;
      const bool enc_fp32 = e->encoder_variant == 3 || !e->v5_enc_h3_ok;      // option "encoder" = 3, or no split-fp16 operands: k_v5_encoder (fp32 MFMA)
      e->last_frontend_kernel = enc_fp32 ? 1 : 2;
      auto enc = [&](float *gx, hipStream_t s_) {
         KernelTimer t(e, VADC_AMD_KERNEL_FRONTEND, s_);
         if (sizeof(T) == 2) launch_v5_encoder_s16(reinterpret_cast<const int16_t *>(d_in), e->d_ctx5, e->v5, gx, e->d_x35, n_streams, n_chunks, enc_fp32, encoder_cus(e, s_), s_);
         else                launch_v5_encoder_f32(reinterpret_cast<const float *>(d_in), e->d_ctx5, e->v5, gx, e->d_x35, n_streams, n_chunks, enc_fp32, encoder_cus(e, s_), s_);
      };
      e->last_lstm_kernel = (fp32 || !e->v5.whh_h) ? 3 : 6;
      if ((long)n_streams * n_chunks < 2048) {
         wait_last_all(e, st);
         enc(e->d_gx5[0], st);
         { KernelTimer t(e, VADC_AMD_KERNEL_LSTM, st); launch_v5_lstm(e->v5, e->d_gx5[0], e->d_h, e->d_c, d_probs, n_streams, n_chunks, fp32, st); }
         record_last_on(e, st);
      } else {
         int rc5 = ensure_pipeline_streams(e, n_streams, 6);
         if (rc5) return rc5;
         if (hipStreamQuery(st) != hipSuccess) {                 // see the fork of the other models below
            (void)hipGetLastError();
            (void)hipEventRecord(e->ev_in, st);
            (void)hipStreamWaitEvent(e->sA, e->ev_in, 0);
            (void)hipStreamWaitEvent(e->sB, e->ev_in, 0);
         }
         wait_last(e, e->sA, e->last_a, e->last_a_on);
         wait_last(e, e->sB, e->last_b, e->last_b_on);
         if (e->last_c != e->last_b) wait_last(e, e->sB, e->last_c, e->last_c_on);
         e->xpar ^= 1;
         const int xp = e->xpar;
         if (e->ev_b_valid[xp]) (void)hipStreamWaitEvent(e->sA, e->ev_b[xp], 0);     // gx of this parity: last read by the recurrence two forked calls ago
         enc(e->d_gx5[xp], e->sA);
         (void)hipEventRecord(e->ev_fe[0], e->sA);
         e->last_a = e->ev_fe[0]; e->last_a_on = e->sA; e->ev_last_valid = true; e->last_on_valid = true;
         (void)hipStreamWaitEvent(e->sB, e->ev_fe[0], 0);
         { KernelTimer t(e, VADC_AMD_KERNEL_LSTM, e->sB); launch_v5_lstm(e->v5, e->d_gx5[xp], e->d_h, e->d_c, d_probs, n_streams, n_chunks, fp32, e->sB); }
         (void)hipEventRecord(e->ev_b[xp], e->sB);
         e->ev_b_valid[xp] = true;
         e->last_b = e->last_c = e->ev_b[xp]; e->last_b_on = e->last_c_on = e->sB;
         if (!e->defer_join) { (void)hipStreamWaitEvent(st, e->last_a, 0); (void)hipStreamWaitEvent(st, e->ev_b[xp], 0); }
      }
      hipError_t he5 = hipGetLastError();
      if (he5 != hipSuccess) return fail(VADC_AMD_EHIP, "kernel launch failed: %s", hipGetErrorString(he5));
      return VADC_AMD_OK;
   }
   const int G = pick_groups(e, n_chunks);
   // small calls stay on the caller's stream -- unless the caller pipelines calls (defer_join) and the recurrence is long: then the layer-major pair on the
   // internal streams runs layer 1 of one call beside layer 0 of the next (16 x 96: 0.13 -> 0.27 M audio-s/s)
   const bool forked = !(G == 1 && (long)n_streams * n_chunks < 2048 && !(e->defer_join && n_chunks >= 32));
   const int lk = resolve_lstm(e, n_streams, forked, n_chunks);
   e->last_lstm_kernel = lk;
   e->last_frontend_kernel = pick_frontend(e, d_in);
   int rc = VADC_AMD_OK;
   // Small calls run straight on the caller's stream.  Larger ones always fork onto the two internal streams, even
   // with one group: the internal streams are in-order across calls, so a caller that alternates between two
   // streams gets the NEXT call's front end + encoder overlapped with THIS call's LSTM (cross-call pipelining)
   // while every call keeps strict stream semantics (its results are complete when its own stream reaches the join).
   if (!forked) {
      const ItemMap map{n_chunks, 0, n_chunks};
      wait_last_all(e, st);
      rc = launch_sequence(e, SeqKey{d_in, d_probs, n_streams, n_chunks, (int)sizeof(T), 1, -1, e->xpar, lk, e->last_frontend_kernel}, st, [&] {
         run_front_and_encoder<T>(e, d_in, n_streams * n_chunks, map, lk, st);
         launch_lstm_on(e, lk, d_probs, n_streams, n_chunks, 0, n_chunks, st);
      });
      if (rc) return rc;
      record_last_on(e, st);
   } else {
      rc = ensure_pipeline_streams(e, n_streams, lk);
      if (rc) return rc;
      // fork: stream A = front end + encoder of every chunk group in order, stream B = the LSTM chain (layer-major form: B = layer 0, C = layer 1)
      const bool split = lk == 7;
      // everything the caller enqueued on `st` before this call comes first -- unless `st` has drained already (a caller that issues call after
      // call with defer_join never puts anything on it): then there is nothing to order against, and three cross-queue waits are saved
      if (hipStreamQuery(st) != hipSuccess) {
         (void)hipGetLastError();                              // hipErrorNotReady is not an error
         (void)hipEventRecord(e->ev_in, st);
         (void)hipStreamWaitEvent(e->sA, e->ev_in, 0);
         (void)hipStreamWaitEvent(e->sB, e->ev_in, 0);
         if (split) (void)hipStreamWaitEvent(e->sC, e->ev_in, 0);
      }
      {
         // last_b / last_c: the last work on the layer-0 / layer-1 halves of the per-stream state, whichever stream it ran on.  One kernel for
         // both layers waits for both; in the layer-major form each layer waits for its own half only (that is what lets layer 0 of this
         // call run beside layer 1 of the previous one)
         wait_last(e, e->sA, e->last_a, e->last_a_on);
         // With more stream tiles than half the CUs the LSTM is throughput work that fills the chip by itself: the next call's front end beside it
         // only loses (the persistent GEMM front end cannot place its two workgroups per CU: 0.57 -> 0.80 ms per 65,536 chunks, v4 4.02 M -> 3.65 M
         // audio-s/s at 4096 x 16), so the calls run back to back
         // (Round 3: only with the GEMM front end -- v4 5.96 M -> 4.92 M at 4096 x 16 when overlapped.  The tree front end of v3.1 gains from the overlap,
         // although the chain then stretches over the whole front end: 1664 x 32 3.01 -> 3.19 M, 2560 x 32 3.18 -> 3.37 M, 4096 x 16 3.43 -> 3.49 M.)
         if (e->use_gemm_frontend() && (n_streams + kLstmTile - 1) / kLstmTile > e->n_cus / 2) { wait_last(e, e->sA, e->last_b, e->last_b_on); wait_last(e, e->sA, e->last_c, e->last_c_on); }
         wait_last(e, e->sB, e->last_b, e->last_b_on);
         wait_last(e, split ? e->sC : e->sB, e->last_c, e->last_c_on);
      }
      // this call's hand-off buffer; its last reader was the LSTM of the forked call before the previous one (long finished: the wait is free)
      e->xpar ^= 1;
      const int xp = e->xpar;
      e->d_act[3] = e->d_xpair[xp];
      if (e->ev_b_valid[xp]) (void)hipStreamWaitEvent(e->sA, e->ev_b[xp], 0);
      if (split && e->ev_c_valid[xp]) (void)hipStreamWaitEvent(e->sB, e->ev_c[xp], 0);     // this call's h0 pair: last read by layer 1 two calls ago
      // group sizes: a SHORT first group (the LSTM chain starts early), the rest split evenly
      int sizes[vadc_amd_engine::kMaxGroups];
      {
         int first = n_chunks / (4 * G);
         if (first < 1) first = 1;
         if (G == 1) first = n_chunks;
         sizes[0] = first;
         int rest = n_chunks - first;
         for (int g = 1; g < G; ++g) { sizes[g] = (rest + (G - g) - 1) / (G - g); rest -= sizes[g]; }
      }
      int c0 = 0;
      for (int gi = 0; gi < G; ++gi) {
         const int cg = sizes[gi];
         if (cg <= 0) continue;
         const ItemMap map{n_chunks, c0, cg};
         rc = launch_sequence(e, SeqKey{d_in, nullptr, n_streams, n_chunks, (int)sizeof(T), G, gi, xp, lk, e->last_frontend_kernel}, e->sA, [&] {
            run_front_and_encoder<T>(e, d_in, n_streams * cg, map, lk, e->sA);
         });
         if (rc) return rc;
         const bool last_group = c0 + cg >= n_chunks;
         (void)hipEventRecord(e->ev_fe[gi], e->sA);           // one record per group: the LSTM's go-ahead, and for the last group also "encoder buffers free" (last_a) and the join
         e->last_a = e->ev_fe[gi]; e->last_a_on = e->sA; e->ev_last_valid = true; e->last_on_valid = true;
         (void)hipStreamWaitEvent(e->sB, e->ev_fe[gi], 0);
         if (!split) launch_lstm_on(e, lk, d_probs, n_streams, n_chunks, c0, cg, e->sB);
         else {
            // (Cutting a group's recurrence into several launches per layer, layer 1 of a part beside layer 0 of the next, was measured: the
            // call's last probability is ready earlier, but 256 x 96 loses 1.5 % in steady state -- 2.49 M -> 2.45 M -- to the extra launches,
            // records and cross-queue waits, and a 20-step run gains nothing measurable.)
            // TRAIL: layer 1 is launched beside layer 0 and follows its published progress a few slots behind (kernels_lstm.hip) -- only on the CU partition: there the
            // two launches have queues and CUs of their own, so a layer-1 workgroup that polls can never stand in the way of the layer-0 workgroup it waits for
            // (with per-kernel profiling on, layer 1's event pair includes its wait for layer 0's progress: the two launches overlap by design)
            // and only while either half of the partition has CUs on all 8 XCDs (a forced "lstm_cus" of 8 would not)
            // and not on a SHARED partition: a layer-1 workgroup that waits for its layer 0 holds its CU, which the front end + encoder could otherwise use
            // between two chains (Silero v4 at 256 streams: 4.72 -> 4.36 M with it)
            // and only in a process whose kernels overlap at all: a tool that serialises dispatches (rocprofv3 --pmc) would start layer 0 when layer 1 has ended
            const bool trail = e->lstm_trail == 1 && e->lstm_cus >= 16 && (e->lstm_cus / 2) % 8 == 0 && !e->lstm_shared && e->cu_partition_usable() && e->trail_possible();
            int *progress = nullptr;
            e->lstm_trail_used = trail ? 1 : 0;
            if (trail) {
               if (++e->lstm_epoch > 2047) {                    // the 11-bit epoch wraps: once in 2,047 launches the words are cleared behind everything that may read them
                  // ordered by the streams themselves, not by what kind of streams they happen to be: the clear runs on the layer-0 stream behind everything
                  // issued so far on BOTH layer streams (earlier calls and this call's earlier chunk groups, whose layer-1 workgroups poll these words), and
                  // the layer-1 stream goes on behind the clear
                  (void)hipEventRecord(e->ev_wrap[0], e->sC);
                  (void)hipStreamWaitEvent(e->sB, e->ev_wrap[0], 0);
                  for (int p = 0; p < 2; ++p) (void)hipMemsetAsync(e->d_lstm_progress[p], 0, vadc_amd_engine::kMaxGroups * 3 * e->progress_tiles * sizeof(int), e->sB);
                  (void)hipEventRecord(e->ev_wrap[1], e->sB);
                  (void)hipStreamWaitEvent(e->sC, e->ev_wrap[1], 0);
                  e->lstm_epoch = 1;
               }
               progress = e->d_lstm_progress[xp] + (size_t)gi * 3 * e->progress_tiles;
            }
            // (tests) trail_fault 1: this pair's layer 0 comes LATE -- it is held back until its layer 1 has given up; 2: it never comes (its tickets still advance,
            // as they would beside a layer 0 that never started)
            const int fault = trail ? e->trail_fault : 0;
            if (trail) e->trail_fault = 0;
            bool l0_failed = false;
            if (fault == 0) {
               KernelTimer t(e, VADC_AMD_KERNEL_LSTM, e->sB);
               launch_lstm_layer(0, e->d_act[3], e->d_h0pair[xp], e->lstm, e->d_h, e->d_c, d_probs, n_streams, n_chunks, c0, cg, e->sB, e->model, e->lstm_steps, progress, e->lstm_epoch, e->d_lstm_tickets, (int)e->ticket_base, e->d_trail_err, e->d_trail_recov, e->trail_wait_limit);
               l0_failed = trail && hipGetLastError() != hipSuccess;      // a layer 0 that was never launched must not be polled for: the pair in turn then
            }
            hipEvent_t l0_done = last_group ? e->ev_b[xp] : e->ev_l0[gi];   // the call's last record on this stream is also this hand-off pair's "layer 0 done"
            if (fault != 1) (void)hipEventRecord(l0_done, e->sB);
            if (!trail || l0_failed) (void)hipStreamWaitEvent(e->sC, l0_done, 0);
            {
               KernelTimer t(e, VADC_AMD_KERNEL_LSTM_L1, e->sC);
               launch_lstm_layer(1, e->d_act[3], e->d_h0pair[xp], e->lstm, e->d_h, e->d_c, d_probs, n_streams, n_chunks, c0, cg, e->sC, e->model, e->lstm_steps, progress, e->lstm_epoch, e->d_lstm_tickets, (int)e->ticket_base, e->d_trail_err, e->d_trail_recov, e->trail_wait_limit);
            }
            if (fault == 1) {                                   // the late layer 0: behind its layer 1 (which therefore gives up), then "layer 0 done" as always
               (void)hipEventRecord(e->ev_redo, e->sC);
               (void)hipStreamWaitEvent(e->sB, e->ev_redo, 0);
               launch_lstm_layer(0, e->d_act[3], e->d_h0pair[xp], e->lstm, e->d_h, e->d_c, d_probs, n_streams, n_chunks, c0, cg, e->sB, e->model, e->lstm_steps, progress, e->lstm_epoch, e->d_lstm_tickets, (int)e->ticket_base, e->d_trail_err, e->d_trail_recov, e->trail_wait_limit);
               (void)hipEventRecord(l0_done, e->sB);
            }
            if (trail && !l0_failed) {
               // FAIL-SAFE: behind the pair -- and behind layer 0's END (a kernel boundary) -- the REDO form of layer 1: every workgroup leaves at once unless its tile's
               // layer 1 gave up (its wait ran out; its pair on another XCD; its workgroup never ran), in which case the tile is done again from the untouched
               // pre-call state over the complete h0 sequence.  Later calls are ordered behind it like behind any layer-1 launch, so none consumes a faulted state.
               (void)hipStreamWaitEvent(e->sC, l0_done, 0);
               launch_lstm_layer(2, e->d_act[3], e->d_h0pair[xp], e->lstm, e->d_h, e->d_c, d_probs, n_streams, n_chunks, c0, cg, e->sC, e->model, e->lstm_steps, progress, e->lstm_epoch, e->d_lstm_tickets, (int)e->ticket_base, e->d_trail_err, e->d_trail_recov, e->trail_wait_limit);
            }
            if (trail) e->ticket_base += (unsigned)(((n_streams + kLstmTile - 1) / kLstmTile + 7) / 8);      // every XCD's counter of either layer has advanced by grid / 8
         }
         c0 += cg;
      }
      // one record per stream marks the end of the call there
      if (!split) (void)hipEventRecord(e->ev_b[xp], e->sB);
      e->ev_b_valid[xp] = true;
      e->last_b = e->ev_b[xp]; e->last_b_on = e->sB;
      if (split) {
         (void)hipEventRecord(e->ev_c[xp], e->sC);
         e->ev_c_valid[xp] = true;
         e->last_c = e->ev_c[xp]; e->last_c_on = e->sC;
      } else { e->last_c = e->ev_b[xp]; e->last_c_on = e->sB; }
      if (!e->defer_join) {                                    // strict stream semantics: the caller's stream continues when the call is complete
         (void)hipStreamWaitEvent(st, e->last_a, 0);
         (void)hipStreamWaitEvent(st, e->ev_b[xp], 0);
         if (split) (void)hipStreamWaitEvent(st, e->ev_c[xp], 0);
      }
   }
   hipError_t he = hipGetLastError();
   if (he != hipSuccess) return fail(VADC_AMD_EHIP, "kernel launch failed: %s", hipGetErrorString(he));
   return VADC_AMD_OK;
}

extern "C" int vadc_amd_run_device_f32(vadc_amd_engine *e, const float *d_samples, int n_streams, int n_chunks, float *d_probs, void *hip_stream)
{
   int rc = check_shape(e, n_streams, n_chunks, "run_device_f32");
   if (rc) return rc;
   if (!d_samples || !d_probs) return fail(VADC_AMD_EINVAL, "run_device_f32: NULL buffer");
   HIP_TRY(hipSetDevice(e->device), VADC_AMD_EHIP);
   return run_device<float>(e, d_samples, n_streams, n_chunks, d_probs, (hipStream_t)hip_stream);
}

extern "C" int vadc_amd_run_device_s16(vadc_amd_engine *e, const int16_t *d_pcm, int n_streams, int n_chunks, float *d_probs, void *hip_stream)
{
   int rc = check_shape(e, n_streams, n_chunks, "run_device_s16");
   if (rc) return rc;
   if (!d_pcm || !d_probs) return fail(VADC_AMD_EINVAL, "run_device_s16: NULL buffer");
   HIP_TRY(hipSetDevice(e->device), VADC_AMD_EHIP);
   return run_device<int16_t>(e, d_pcm, n_streams, n_chunks, d_probs, (hipStream_t)hip_stream);
}

// The TRAIL pair is fail-safe (kernels_lstm.hip, FAIL-SAFE): a layer 1 that gave up on its layer 0 is done again on the device by the REDO launch, in stream order, and
// the caller sees correct probabilities and no error.  What the host does about it, at its synchronisation points: it looks at the counter of redone tiles and, when it
// has moved, stops launching pairs side by side (an environment in which layer 0 comes seconds late -- a time-sliced GPU, a tool that serialises kernels -- would cost
// every call its bounded wait) and puts the ticket counters back in step.
static void look_at_trail_recoveries(vadc_amd_engine *e)
{
   if (!e->d_trail_recov || !e->h_trail_err || !e->h_trail_err[1]) return;      // (the flag is host memory: no copy unless a REDO workgroup has worked)
   e->h_trail_err[1] = 0;
   int n = 0;
   if (hipMemcpy(&n, e->d_trail_recov, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) { (void)hipGetLastError(); return; }
   if (n != e->trail_recoveries) {
      e->trail_recoveries = n;
      e->lstm_trail = 0;
      if (e->d_lstm_tickets) (void)hipMemset(e->d_lstm_tickets, 0, 16 * sizeof(int));      // (every launch that drew tickets has finished: this runs behind a synchronisation)
      e->ticket_base = 0;
   }
}
// The one failure the REDO launch cannot repair -- a tile whose LAYER 0 never ran (a workgroup that never got its tile: ticket imbalance) -- sets the fatal word: the
// layer-0 state of the streams has or has not advanced tile by tile, layer 1's has not.  The word is host memory: it is looked at WITHOUT any synchronisation at the head
// of every call (run_device, join, synchronize, the host-buffer entry points), so a caller that only ever issues deferred calls gets the error on its next call, and
// it stays set -- every call fails -- until vadc_amd_reset_streams(all) has put the streams into a defined state again.
static int check_trail_error(vadc_amd_engine *e, const char *where)
{
   if (e->h_trail_err && *e->h_trail_err) {
      if (!e->trail_lost) {
         e->trail_lost = true;
         e->lstm_trail = 0;
      }
   }
   if (!e->trail_lost) return VADC_AMD_OK;
   return fail(VADC_AMD_EHIP, "%s: a stream tile's first recurrence layer never ran beside its second (layer-major LSTM pair) and the call could not be recovered: the "
                              "state of the streams is inconsistent and the probabilities since then are invalid -- vadc_amd_reset_streams(e, NULL, 0) clears this (lstm_trail is now off)", where);
}

extern "C" int vadc_amd_synchronize(vadc_amd_engine *e)
{
   if (!e) return fail(VADC_AMD_EINVAL, "synchronize: NULL engine");
   HIP_TRY(hipSetDevice(e->device), VADC_AMD_EHIP);
   if (e->ev_last_valid) {
      HIP_TRY(hipEventSynchronize(e->last_a), VADC_AMD_EHIP);
      HIP_TRY(hipEventSynchronize(e->last_b), VADC_AMD_EHIP);
      HIP_TRY(hipEventSynchronize(e->last_c), VADC_AMD_EHIP);
   }
   HIP_TRY(hipStreamSynchronize(e->stream), VADC_AMD_EHIP);
   look_at_trail_recoveries(e);
   return check_trail_error(e, "synchronize");
}

extern "C" int vadc_amd_join(vadc_amd_engine *e, void *hip_stream)
{
   if (!e) return fail(VADC_AMD_EINVAL, "join: NULL engine");
   { int rc_ = check_trail_error(e, "join"); if (rc_) return rc_; }
   HIP_TRY(hipSetDevice(e->device), VADC_AMD_EHIP);
   if (e->ev_last_valid) {
      wait_last_all(e, (hipStream_t)hip_stream);
      hipError_t he = hipGetLastError();
      if (he != hipSuccess) return fail(VADC_AMD_EHIP, "join: %s", hipGetErrorString(he));
   }
   return VADC_AMD_OK;
}

// element 1 of every [2] pair (the speech probability, vadc.c:704-713) made contiguous: 4 B per chunk for the multi-GPU gather
__global__ __launch_bounds__(256) void k_pack_speech(const float2 *__restrict__ probs, float *__restrict__ speech, size_t n)
{
   for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) speech[i] = probs[i].y;
}

extern "C" int vadc_amd_speech_probabilities(vadc_amd_engine *e, const float *d_probs, int n_streams, int n_chunks, float *d_speech, void *hip_stream)
{
   if (!e) return fail(VADC_AMD_EINVAL, "speech_probabilities: NULL engine");
   if (!d_probs || !d_speech || n_streams < 0 || n_chunks < 0) return fail(VADC_AMD_EINVAL, "speech_probabilities: bad argument");
   if (((uintptr_t)d_probs & 7) != 0) return fail(VADC_AMD_EINVAL, "speech_probabilities: d_probs must be 8-byte aligned");
   HIP_TRY(hipSetDevice(e->device), VADC_AMD_EHIP);
   const size_t n = (size_t)n_streams * n_chunks;
   if (!n) return VADC_AMD_OK;
   const unsigned blocks = (unsigned)std::min<size_t>((n + 255) / 256, 1024);
   hipLaunchKernelGGL(k_pack_speech, dim3(blocks), dim3(256), 0, (hipStream_t)hip_stream, (const float2 *)d_probs, d_speech, n);
   hipError_t he = hipGetLastError();
   if (he != hipSuccess) return fail(VADC_AMD_EHIP, "speech_probabilities: %s", hipGetErrorString(he));
   return VADC_AMD_OK;
}

extern "C" int vadc_amd_run_f32(vadc_amd_engine *e, const float *samples, int n_streams, int n_chunks, float *probs)
{
   int rc = check_shape(e, n_streams, n_chunks, "run_f32");
   if (rc) return rc;
   if (!samples || !probs) return fail(VADC_AMD_EINVAL, "run_f32: NULL buffer");
   HIP_TRY(hipSetDevice(e->device), VADC_AMD_EHIP);
   const size_t n = (size_t)n_streams * n_chunks;
   HIP_TRY(host_to_device(e, e->d_in_f32, samples, n * e->window * sizeof(float), e->stream), VADC_AMD_EHIP);
   rc = run_device<float>(e, e->d_in_f32, n_streams, n_chunks, e->d_probs, e->stream);
   if (rc) return rc;
   if (e->defer_join) wait_last_all(e, e->stream);        // the synchronous entry points always join (option "defer_join" is about run_device)
   HIP_TRY(device_to_host(e, probs, e->d_probs, n * 2 * sizeof(float), e->stream), VADC_AMD_EHIP);
   HIP_TRY(hipStreamSynchronize(e->stream), VADC_AMD_EHIP);
   look_at_trail_recoveries(e);
   return check_trail_error(e, "run_f32");
}

extern "C" int vadc_amd_run_s16(vadc_amd_engine *e, const int16_t *pcm, int n_streams, int n_chunks, float *probs)
{
   int rc = check_shape(e, n_streams, n_chunks, "run_s16");
   if (rc) return rc;
   if (!pcm || !probs) return fail(VADC_AMD_EINVAL, "run_s16: NULL buffer");
   HIP_TRY(hipSetDevice(e->device), VADC_AMD_EHIP);
   const size_t n = (size_t)n_streams * n_chunks;
   HIP_TRY(host_to_device(e, e->d_in_s16, pcm, n * e->window * sizeof(int16_t), e->stream), VADC_AMD_EHIP);
   rc = run_device<int16_t>(e, e->d_in_s16, n_streams, n_chunks, e->d_probs, e->stream);
   if (rc) return rc;
   if (e->defer_join) wait_last_all(e, e->stream);        // the synchronous entry points always join (option "defer_join" is about run_device)
   HIP_TRY(device_to_host(e, probs, e->d_probs, n * 2 * sizeof(float), e->stream), VADC_AMD_EHIP);
   HIP_TRY(hipStreamSynchronize(e->stream), VADC_AMD_EHIP);
   look_at_trail_recoveries(e);
   return check_trail_error(e, "run_s16");
}

// ---- asynchronous host-buffer entry points: the shape a real backend_run caller has (host buffers in and out, vadc.c:873-909) without the
// synchronous copy -> run -> copy of vadc_amd_run_*.  A call returns once its work is enqueued; vadc_amd_wait_async returns when the probabilities of
// every call issued so far are in their host buffers.
// every copy the async entry points have issued has finished (a registration may only go away under no copy)
static void drain_copy_streams(vadc_amd_engine *e)
{
   for (hipStream_t cs : {e->s_h2d, e->s_d2h, e->s_h2dx[0], e->s_h2dx[1], e->s_h2dx[2]}) if (cs) (void)hipStreamSynchronize(cs);
}
static void forget_range(vadc_amd_engine *e, size_t i)
{
   if (e->pinned[i].ours && hipHostUnregister(const_cast<char *>(e->pinned[i].p)) != hipSuccess) (void)hipGetLastError();
   e->pinned.erase(e->pinned.begin() + (long)i);
}
static void pin_host_range(vadc_amd_engine *e, const void *ptr, size_t bytes)
{
   if (!e->pin_host || bytes == 0) return;
   const char *p = static_cast<const char *>(ptr);
   const char *lo = p, *hi = p + bytes;
   for (auto &r : e->pinned) if (p >= r.p && hi <= r.p + r.n) { r.last = ++e->pin_clock; return; }
   // A range that OVERLAPS remembered ones without lying inside one (a caller's buffer grew, or an allocator handed out an address inside an older,
   // freed range): the old registrations go -- after every copy in flight has finished -- and the union is registered in their place
   bool drained = false;
   for (size_t i = 0; i < e->pinned.size();) {
      const auto &r = e->pinned[i];
      if (lo < r.p + r.n && r.p < hi) {
         if (!drained) { drain_copy_streams(e); drained = true; }
         if (r.p < lo) lo = r.p;
         if (r.p + r.n > hi) hi = r.p + r.n;
         forget_range(e, i);
      } else ++i;
   }
   // page-locked memory lets the copy engine read / write the caller's buffer directly (pageable memory goes through a staging copy: 3x slower)
   hipError_t he = hipHostRegister(const_cast<char *>(lo), (size_t)(hi - lo), hipHostRegisterDefault);
   if (he != hipSuccess && (lo != p || hi != p + bytes)) {       // the union was not registrable (part of it is gone): this call's range alone
      (void)hipGetLastError();
      lo = p; hi = p + bytes;
      he = hipHostRegister(const_cast<char *>(lo), bytes, hipHostRegisterDefault);
   }
   if (he != hipSuccess) (void)hipGetLastError();          // already page-locked by the caller (hipHostMalloc, a pinned torch tensor), or not registrable: copy as is
   e->pinned.push_back({lo, (size_t)(hi - lo), he == hipSuccess, ++e->pin_clock});
   if (e->pinned.size() > vadc_amd_engine::kMaxPinned) {   // the least recently used range: at most 3 calls x 2 ranges are in flight, all of them more recent
      size_t victim = 0;
      for (size_t i = 1; i + 1 < e->pinned.size(); ++i) if (e->pinned[i].last < e->pinned[victim].last) victim = i;
      if (!drained) drain_copy_streams(e);
      forget_range(e, victim);
   }
}

extern "C" int vadc_amd_unpin(vadc_amd_engine *e, const void *host_ptr)
{
   if (!e) return fail(VADC_AMD_EINVAL, "unpin: NULL engine");
   if (!host_ptr) return VADC_AMD_OK;
   HIP_TRY(hipSetDevice(e->device), VADC_AMD_EHIP);
   const char *p = static_cast<const char *>(host_ptr);
   bool drained = false;
   for (size_t i = 0; i < e->pinned.size();) {
      if (p >= e->pinned[i].p && p < e->pinned[i].p + e->pinned[i].n) {
         if (!drained) {                                     // the asynchronous calls that may still copy from / into it
            for (auto &sl : e->aslot) if (sl.busy) { HIP_TRY(hipEventSynchronize(sl.out_done), VADC_AMD_EHIP); sl.busy = false; }
            drain_copy_streams(e);
            drained = true;
         }
         forget_range(e, i);
      } else ++i;
   }
   return VADC_AMD_OK;
}

template <typename T>
static int run_async(vadc_amd_engine *e, const T *host_in, int n_streams, int n_chunks, float *host_probs, const char *who)
{
   int rc = check_shape(e, n_streams, n_chunks, who);
   if (rc) return rc;
   if (!host_in || !host_probs) return fail(VADC_AMD_EINVAL, "%s: NULL buffer", who);
   HIP_TRY(hipSetDevice(e->device), VADC_AMD_EHIP);
   if (!e->s_h2d) {
      // The copy streams get a priority of their own (the lowest: a copy never needs to overtake a kernel): HIP multiplexes the streams of one priority
      // onto a handful of hardware queues, and a copy queued behind the front end's kernels waits for them (measured, 256 x 96: 24-29 GB/s from
      // normal-priority copy streams, 39-43 GB/s from these; tools/host_fed_probe.py)
      int lo = 0, hi = 0;
      (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
      auto mk = [&](hipStream_t *st) -> hipError_t { return hipStreamCreateWithPriority(st, hipStreamNonBlocking, lo); };
      HIP_TRY(mk(&e->s_h2d), VADC_AMD_EHIP);
      HIP_TRY(mk(&e->s_d2h), VADC_AMD_EHIP);
      for (int i = 0; i < 3; ++i) {
         HIP_TRY(mk(&e->s_h2dx[i]), VADC_AMD_EHIP);
         HIP_TRY(hipEventCreateWithFlags(&e->ev_h2dx[i], hipEventDisableTiming), VADC_AMD_EHIP);
      }
   }
   vadc_amd_engine::AsyncSlot &sl = e->aslot[e->anext % vadc_amd_engine::kAsyncSlots];
   if (!sl.d_in) {
      // all four or none: a slot left half made by a failed allocation would be taken for complete by the next call
      void *d_in = nullptr; float *d_pr = nullptr; hipEvent_t ev_in = nullptr, ev_out = nullptr;
      hipError_t he = hipMalloc(&d_in, e->max_items * kChunk * sizeof(float));
      if (he == hipSuccess) he = hipMalloc(&d_pr, e->max_items * 2 * sizeof(float));
      if (he == hipSuccess) he = hipEventCreateWithFlags(&ev_in, hipEventDisableTiming);
      if (he == hipSuccess) he = hipEventCreateWithFlags(&ev_out, hipEventDisableTiming);
      if (he != hipSuccess) {
         (void)hipGetLastError();
         if (d_in) (void)hipFree(d_in);
         if (d_pr) (void)hipFree(d_pr);
         if (ev_in) (void)hipEventDestroy(ev_in);
         if (ev_out) (void)hipEventDestroy(ev_out);
         return fail(VADC_AMD_ENOMEM, "%s: staging slot: %s", who, hipGetErrorString(he));
      }
      sl.d_in = d_in; sl.d_probs = d_pr; sl.in_done = ev_in; sl.out_done = ev_out;
   }
   ++e->anext;
   if (sl.busy) HIP_TRY(hipEventSynchronize(sl.out_done), VADC_AMD_EHIP);      // the call that used this slot three calls ago: back-pressure
   const size_t n = (size_t)n_streams * n_chunks;
   pin_host_range(e, host_in, n * e->window * sizeof(T));
   pin_host_range(e, host_probs, n * 2 * sizeof(float));
   {
      // option "h2d_streams" > 1: the input goes in that many pieces on as many copy streams (several copy engines on the link: up to 50 GB/s when the
      // pieces land on different hardware queues, 30 when they do not -- one piece is the steadier default)
      const size_t bytes = n * e->window * sizeof(T);
      const int parts = (bytes >= (size_t)8 << 20) ? e->h2d_parts : 1;
      const size_t piece = ((bytes + parts - 1) / parts + 4095) & ~(size_t)4095;
      for (int i = 0; i < parts; ++i) {
         const size_t off = (size_t)i * piece;
         if (off >= bytes) break;
         hipStream_t cs = i == 0 ? e->s_h2d : e->s_h2dx[i - 1];
         HIP_TRY(hipMemcpyAsync(static_cast<char *>(sl.d_in) + off, reinterpret_cast<const char *>(host_in) + off, std::min(piece, bytes - off), hipMemcpyHostToDevice, cs), VADC_AMD_EHIP);
         if (i > 0) { HIP_TRY(hipEventRecord(e->ev_h2dx[i - 1], cs), VADC_AMD_EHIP); HIP_TRY(hipStreamWaitEvent(e->s_h2d, e->ev_h2dx[i - 1], 0), VADC_AMD_EHIP); }
      }
   }
   HIP_TRY(hipEventRecord(sl.in_done, e->s_h2d), VADC_AMD_EHIP);
   HIP_TRY(hipStreamWaitEvent(e->stream, sl.in_done, 0), VADC_AMD_EHIP);
   const int dj = e->defer_join;
   e->defer_join = 1;                                       // consecutive calls overlap inside the engine; the D2H stream joins this one
   rc = run_device<T>(e, static_cast<const T *>(sl.d_in), n_streams, n_chunks, sl.d_probs, e->stream);
   e->defer_join = dj;
   if (rc) return rc;
   wait_last_all(e, e->s_d2h);
   HIP_TRY(hipMemcpyAsync(host_probs, sl.d_probs, n * 2 * sizeof(float), hipMemcpyDeviceToHost, e->s_d2h), VADC_AMD_EHIP);
   HIP_TRY(hipEventRecord(sl.out_done, e->s_d2h), VADC_AMD_EHIP);
   sl.busy = true;
   return VADC_AMD_OK;
}

extern "C" int vadc_amd_run_s16_async(vadc_amd_engine *e, const int16_t *host_pcm, int n_streams, int n_chunks, float *host_probs)
{
   if (!e) return fail(VADC_AMD_EINVAL, "run_s16_async: NULL engine");
   return run_async<int16_t>(e, host_pcm, n_streams, n_chunks, host_probs, "run_s16_async");
}

extern "C" int vadc_amd_run_f32_async(vadc_amd_engine *e, const float *host_samples, int n_streams, int n_chunks, float *host_probs)
{
   if (!e) return fail(VADC_AMD_EINVAL, "run_f32_async: NULL engine");
   return run_async<float>(e, host_samples, n_streams, n_chunks, host_probs, "run_f32_async");
}

extern "C" int vadc_amd_wait_async(vadc_amd_engine *e)
{
   if (!e) return fail(VADC_AMD_EINVAL, "wait_async: NULL engine");
   HIP_TRY(hipSetDevice(e->device), VADC_AMD_EHIP);
   for (auto &sl : e->aslot)
      if (sl.busy) { HIP_TRY(hipEventSynchronize(sl.out_done), VADC_AMD_EHIP); sl.busy = false; }
   look_at_trail_recoveries(e);
   return check_trail_error(e, "wait_async");
}

// ---------------------------------------------------------------------------------------------------
// per-stream state
// ---------------------------------------------------------------------------------------------------
// state accessors are synchronous; they first wait for the last LSTM enqueued through ANY stream
static int wait_last_lstm(vadc_amd_engine *e)
{
   if (e->ev_last_valid) { HIP_TRY(hipEventSynchronize(e->last_b), VADC_AMD_EHIP); HIP_TRY(hipEventSynchronize(e->last_c), VADC_AMD_EHIP); }
   HIP_TRY(hipStreamSynchronize(e->stream), VADC_AMD_EHIP);
   return VADC_AMD_OK;
}
static int wait_all_prior(vadc_amd_engine *e);
static int wait_all_prior_fwd(vadc_amd_engine *e) { HIP_TRY(hipSetDevice(e->device), VADC_AMD_EHIP); return wait_all_prior(e); }
// the stage taps overwrite the intermediates (Y, FM, layer outputs, hand-off tiles): wait for EVERYTHING enqueued before, on any stream
static int wait_all_prior(vadc_amd_engine *e)
{
   if (e->ev_last_valid) HIP_TRY(hipEventSynchronize(e->last_a), VADC_AMD_EHIP);
   return wait_last_lstm(e);
}

extern "C" int vadc_amd_reset_streams(vadc_amd_engine *e, const int32_t *ids, int n)
{
   if (!e) return fail(VADC_AMD_EINVAL, "reset_streams: NULL engine");
   HIP_TRY(hipSetDevice(e->device), VADC_AMD_EHIP);
   { int rc_ = wait_last_lstm(e); if (rc_) return rc_; }
   if (!ids) {
      if (e->trail_lost || (e->h_trail_err && e->h_trail_err[0])) {      // every stream gets a defined state again: the fatal condition of a TRAIL pair ends here
         if (e->h_trail_err) e->h_trail_err[0] = 0;
         e->trail_lost = false;
         if (e->d_lstm_tickets) HIP_TRY(hipMemset(e->d_lstm_tickets, 0, 16 * sizeof(int)), VADC_AMD_EHIP);
         e->ticket_base = 0;
      }
      if (e->d_ctx5) HIP_TRY(hipMemsetAsync(e->d_ctx5, 0, (size_t)e->max_streams * 64 * sizeof(float), e->stream), VADC_AMD_EHIP);
      HIP_TRY(hipMemsetAsync(e->d_h, 0, (size_t)e->max_streams * 128 * sizeof(float), e->stream), VADC_AMD_EHIP);
      HIP_TRY(hipMemsetAsync(e->d_c, 0, (size_t)e->max_streams * 128 * sizeof(float), e->stream), VADC_AMD_EHIP);
   } else {
      for (int i = 0; i < n; ++i) {
         if (ids[i] < 0 || ids[i] >= e->max_streams) return fail(VADC_AMD_EINVAL, "reset_streams: stream %d out of range", ids[i]);
         if (e->d_ctx5) HIP_TRY(hipMemsetAsync(e->d_ctx5 + (size_t)ids[i] * 64, 0, 64 * sizeof(float), e->stream), VADC_AMD_EHIP);
         HIP_TRY(hipMemsetAsync(e->d_h + (size_t)ids[i] * 128, 0, 128 * sizeof(float), e->stream), VADC_AMD_EHIP);
         HIP_TRY(hipMemsetAsync(e->d_c + (size_t)ids[i] * 128, 0, 128 * sizeof(float), e->stream), VADC_AMD_EHIP);
      }
   }
   HIP_TRY(hipStreamSynchronize(e->stream), VADC_AMD_EHIP);
   return VADC_AMD_OK;
}

extern "C" int vadc_amd_get_state(vadc_amd_engine *e, int stream, float *h, float *c)
{
   if (!e || !h || !c || stream < 0 || stream >= e->max_streams) return fail(VADC_AMD_EINVAL, "get_state: bad argument");
   HIP_TRY(hipSetDevice(e->device), VADC_AMD_EHIP);
   { int rc_ = wait_last_lstm(e); if (rc_) return rc_; }
   HIP_TRY(hipMemcpy(h, e->d_h + (size_t)stream * 128, 128 * sizeof(float), hipMemcpyDeviceToHost), VADC_AMD_EHIP);
   HIP_TRY(hipMemcpy(c, e->d_c + (size_t)stream * 128, 128 * sizeof(float), hipMemcpyDeviceToHost), VADC_AMD_EHIP);
   return VADC_AMD_OK;
}

extern "C" int vadc_amd_set_state(vadc_amd_engine *e, int stream, const float *h, const float *c)
{
   if (!e || !h || !c || stream < 0 || stream >= e->max_streams) return fail(VADC_AMD_EINVAL, "set_state: bad argument");
   HIP_TRY(hipSetDevice(e->device), VADC_AMD_EHIP);
   { int rc_ = wait_last_lstm(e); if (rc_) return rc_; }
   HIP_TRY(hipMemcpy(e->d_h + (size_t)stream * 128, h, 128 * sizeof(float), hipMemcpyHostToDevice), VADC_AMD_EHIP);
   HIP_TRY(hipMemcpy(e->d_c + (size_t)stream * 128, c, 128 * sizeof(float), hipMemcpyHostToDevice), VADC_AMD_EHIP);
   return VADC_AMD_OK;
}

// Silero v5: the third piece of per-stream state, the last 64 samples of the previous window (vadc.c:697-701 keeps it on the host; here it lives on the
// device next to h and c) -- saving or migrating a stream needs it too
extern "C" int vadc_amd_get_context(vadc_amd_engine *e, int stream, float *ctx)
{
   if (!e || !ctx || stream < 0 || stream >= e->max_streams) return fail(VADC_AMD_EINVAL, "get_context: bad argument");
   if (e->model != VADC_AMD_MODEL_V5) return fail(VADC_AMD_EINVAL, "get_context: only Silero v5 keeps a sample context (caps.context_size = 0)");
   HIP_TRY(hipSetDevice(e->device), VADC_AMD_EHIP);
   { int rc_ = wait_all_prior(e); if (rc_) return rc_; }
   HIP_TRY(hipMemcpy(ctx, e->d_ctx5 + (size_t)stream * 64, 64 * sizeof(float), hipMemcpyDeviceToHost), VADC_AMD_EHIP);
   return VADC_AMD_OK;
}

extern "C" int vadc_amd_set_context(vadc_amd_engine *e, int stream, const float *ctx)
{
   if (!e || !ctx || stream < 0 || stream >= e->max_streams) return fail(VADC_AMD_EINVAL, "set_context: bad argument");
   if (e->model != VADC_AMD_MODEL_V5) return fail(VADC_AMD_EINVAL, "set_context: only Silero v5 keeps a sample context (caps.context_size = 0)");
   HIP_TRY(hipSetDevice(e->device), VADC_AMD_EHIP);
   { int rc_ = wait_all_prior(e); if (rc_) return rc_; }
   HIP_TRY(hipMemcpy(e->d_ctx5 + (size_t)stream * 64, ctx, 64 * sizeof(float), hipMemcpyHostToDevice), VADC_AMD_EHIP);
   return VADC_AMD_OK;
}

// ---------------------------------------------------------------------------------------------------
// stage taps
// ---------------------------------------------------------------------------------------------------
// A built window (a multiple of 256 samples) that the hot path runs in the 24-frame geometry (17 .. 20 frames: see option "window") has its stage taps in its OWN geometry:
// for the length of a tap call the engine is put back into it (the per-stage kernels of that geometry serve the taps as before)
struct OwnGeometryForTaps {
   vadc_amd_engine *e; int frames;
   explicit OwnGeometryForTaps(vadc_amd_engine *e_) : e(e_), frames(e_ ? e_->frames : 0)
   {
      if (e && e->model == VADC_AMD_MODEL_V4 && e->padded_window() && e->frames_valid % 4 == 0) { e->frames = e->frames_valid; stage_elems_v4(e->frames, e->stride3(), e->stage_elems); }
   }
   ~OwnGeometryForTaps() { if (e && e->frames != frames) { e->frames = frames; stage_elems_v4(e->frames, e->stride3(), e->stage_elems); } }
};

static float *stage_buffer(vadc_amd_engine *e, int stage)
{
   switch (stage) {
   case VADC_AMD_STAGE_MAGNITUDE:  return e->model == VADC_AMD_MODEL_V4 ? e->d_MAG : e->d_Y;
   case VADC_AMD_STAGE_NORMALIZED: return e->d_tap;
   default:                        return e->d_act[stage - VADC_AMD_STAGE_LAYER1];
   }
}

extern "C" int vadc_amd_debug_stage_from_samples(vadc_amd_engine *e, const float *samples, int n, int stage, float *out)
{
   if (!e || !samples || !out || stage < 0 || stage >= VADC_AMD_STAGE_COUNT) return fail(VADC_AMD_EINVAL, "debug_stage_from_samples: bad argument");
   if (e->model == VADC_AMD_MODEL_V5) return fail(VADC_AMD_EINVAL, "debug_stage_from_samples: no stage taps for the Silero v5 path");
   OwnGeometryForTaps own(e);
   if (e->padded_window()) return fail(VADC_AMD_EINVAL, "debug_stage_from_samples: stage taps exist at the built windows (multiples of 256 samples); window=%d runs in the next larger geometry", e->window);
   if (n <= 0 || (size_t)n > e->max_items) return fail(VADC_AMD_EINVAL, "debug_stage_from_samples: n=%d out of range", n);
   HIP_TRY(hipSetDevice(e->device), VADC_AMD_EHIP);
   { int rc_ = wait_all_prior(e); if (rc_) return rc_; }
   hipStream_t st = e->stream;
   HIP_TRY(host_to_device(e, e->d_in_f32, samples, (size_t)n * e->window * sizeof(float), st), VADC_AMD_EHIP);
   const ItemMap map{n, 0, n};
   if (e->use_gemm_frontend() && !(e->model != VADC_AMD_MODEL_V4 && stage == VADC_AMD_STAGE_MAGNITUDE))   // v3.1 keeps no magnitude buffer: that tap comes from the tree kernel
      launch_frontend_gemm_f32(e->d_in_f32, e->d_afrag, e->d_nyq, e->d_Y, e->d_MAG, e->d_FM, e->max_items * kFrames, n, map, e->n_cus, st,
                               e->model == VADC_AMD_MODEL_V4 ? e->v4_geo() : 0, 0);
   else if (e->model == VADC_AMD_MODEL_V4) launch_frontend_v4_f32(e->d_in_f32, e->d_basis, e->d_Y, e->d_MAG, e->d_FM, e->max_items * kFrames, n, map, st);
   else if (e->sym_ok && e->frontend_variant == 0) launch_frontend_sym_f32(e->d_in_f32, e->d_basis, e->d_Y, e->d_FM, e->max_items * kFrames, n, map, stage == VADC_AMD_STAGE_MAGNITUDE ? 1 : 0, st, e->fe_xcd, e->zero_im0);
   else launch_frontend_fl_f32(e->d_in_f32, e->d_basis, e->d_Y, e->d_FM, e->max_items * kFrames, n, map, stage == VADC_AMD_STAGE_MAGNITUDE ? 1 : 0, st);
   if (stage == VADC_AMD_STAGE_NORMALIZED) launch_normalize_tap(e->d_Y, e->d_FM, e->max_items * kFrames, e->d_tap, n, st, e->frames);
   if (stage >= VADC_AMD_STAGE_LAYER1) run_encoder_layers(e, 0, stage - VADC_AMD_STAGE_LAYER1, n, map, 0, st);
   HIP_TRY(hipGetLastError(), VADC_AMD_EHIP);
   HIP_TRY(device_to_host(e, out, stage_buffer(e, stage), (size_t)n * e->stage_elems[stage] * sizeof(float), st), VADC_AMD_EHIP);
   HIP_TRY(hipStreamSynchronize(st), VADC_AMD_EHIP);
   return VADC_AMD_OK;
}

extern "C" int vadc_amd_debug_stage_from_stage(vadc_amd_engine *e, const float *in, int n, int from_stage, int to_stage, float *out)
{
   if (!e || !in || !out || from_stage < 0 || to_stage >= VADC_AMD_STAGE_COUNT || to_stage <= from_stage)
      return fail(VADC_AMD_EINVAL, "debug_stage_from_stage: bad argument");
   if (e->model == VADC_AMD_MODEL_V5) return fail(VADC_AMD_EINVAL, "debug_stage_from_stage: no stage taps for the Silero v5 path");
   OwnGeometryForTaps own(e);
   if (e->padded_window()) return fail(VADC_AMD_EINVAL, "debug_stage_from_stage: stage taps exist at the built windows (multiples of 256 samples); window=%d runs in the next larger geometry", e->window);
   if (n <= 0 || (size_t)n > e->max_items) return fail(VADC_AMD_EINVAL, "debug_stage_from_stage: n=%d out of range", n);
   if (e->model == VADC_AMD_MODEL_V4 && from_stage < VADC_AMD_STAGE_LAYER1)
      return fail(VADC_AMD_EINVAL, "debug_stage_from_stage: the v4 first block takes magnitude AND normalized; feed LAYER1..3 or use from_samples");
   HIP_TRY(hipSetDevice(e->device), VADC_AMD_EHIP);
   { int rc_ = wait_all_prior(e); if (rc_) return rc_; }
   hipStream_t st = e->stream;
   const size_t in_bytes = (size_t)n * e->stage_elems[from_stage] * sizeof(float);
   int first_layer = 0;
   if (from_stage == VADC_AMD_STAGE_MAGNITUDE) {
      HIP_TRY(host_to_device(e, e->d_tap, in, in_bytes, st), VADC_AMD_EHIP);
      launch_lognorm_from_magnitude(e->d_tap, e->d_Y, e->d_FM, e->max_items * kFrames, n, st);
   } else if (from_stage == VADC_AMD_STAGE_NORMALIZED) {
      // already normalized: feed as Y with zero frame means (offset 0)
      HIP_TRY(host_to_device(e, e->d_Y, in, in_bytes, st), VADC_AMD_EHIP);
      HIP_TRY(hipMemsetAsync(e->d_FM, 0, kBinSplit * e->max_items * kFrames * sizeof(float), st), VADC_AMD_EHIP);
   } else {
      first_layer = from_stage - VADC_AMD_STAGE_LAYER1 + 1;
      HIP_TRY(host_to_device(e, e->d_act[first_layer - 1], in, in_bytes, st), VADC_AMD_EHIP);
   }
   if (to_stage == VADC_AMD_STAGE_NORMALIZED) launch_normalize_tap(e->d_Y, e->d_FM, e->max_items * kFrames, e->d_tap, n, st, e->frames);
   else run_encoder_layers(e, first_layer, to_stage - VADC_AMD_STAGE_LAYER1, n, ItemMap{n, 0, n}, 0, st);
   HIP_TRY(hipGetLastError(), VADC_AMD_EHIP);
   HIP_TRY(device_to_host(e, out, stage_buffer(e, to_stage), (size_t)n * e->stage_elems[to_stage] * sizeof(float), st), VADC_AMD_EHIP);
   HIP_TRY(hipStreamSynchronize(st), VADC_AMD_EHIP);
   return VADC_AMD_OK;
}

extern "C" int vadc_amd_debug_layer1_block(vadc_amd_engine *e, int what, const float *y, int n, float *out)
{
   if (!e || !y || !out || what < 1 || what > 5) return fail(VADC_AMD_EINVAL, "debug_layer1_block: bad argument");
   if (e->model != VADC_AMD_MODEL_V31) return fail(VADC_AMD_EINVAL, "debug_layer1_block: Silero v3.1 only (the other models carry no transformer block)");
   if (n <= 0 || (size_t)n > e->max_items) return fail(VADC_AMD_EINVAL, "debug_layer1_block: n=%d out of range", n);
   if (what >= 4 && !e->use_l1_regs()) return fail(VADC_AMD_EINVAL, "debug_layer1_block: what=%d (conv block / tail) is a tap of k_layer1_regs, which this engine does not run (option layer1, or a weight outside fp16's range)", what);
   HIP_TRY(hipSetDevice(e->device), VADC_AMD_EHIP);
   { int rc_ = wait_all_prior(e); if (rc_) return rc_; }
   hipStream_t st = e->stream;
   const size_t bytes = (size_t)n * 16 * 25 * sizeof(float);
   if (what == 4) {
      // the conv block (conv.c:761-814) exactly as the product runs it: the chunk's [129][25] through the LDS-DMA pipeline, partial sums of zero (offset 0)
      HIP_TRY(host_to_device(e, e->d_Y, y, (size_t)n * kBins * kFrames * sizeof(float), st), VADC_AMD_EHIP);
      HIP_TRY(hipMemsetAsync(e->d_FM, 0, kBinSplit * e->max_items * kFrames * sizeof(float), st), VADC_AMD_EHIP);
      L1RegsArgs a;
      a.y = e->d_Y; a.fm = e->d_FM; a.fm_stride = e->max_items * kFrames; a.img = e->d_l1img; a.out = e->d_tap; a.n_chunks = n; a.map = ItemMap{n, 0, n};
      launch_layer1_regs_tap(what, a, st);
      HIP_TRY(hipGetLastError(), VADC_AMD_EHIP);
      HIP_TRY(device_to_host(e, out, e->d_tap, bytes, st), VADC_AMD_EHIP);
      HIP_TRY(hipStreamSynchronize(st), VADC_AMD_EHIP);
      return VADC_AMD_OK;
   }
   HIP_TRY(host_to_device(e, e->d_tap, y, bytes, st), VADC_AMD_EHIP);
   if (e->use_l1_regs()) {
      L1RegsArgs a;
      a.y = e->d_tap; a.fm = nullptr; a.fm_stride = 0; a.img = e->d_l1img; a.out = e->d_Y; a.n_chunks = n; a.map = ItemMap{n, 0, n};
      launch_layer1_regs_tap(what, a, st);
   } else
   launch_layer1_tap(what, e->d_tap, e->lwm[0], e->d_Y, n, ItemMap{n, 0, n}, st);
   HIP_TRY(hipGetLastError(), VADC_AMD_EHIP);
   HIP_TRY(device_to_host(e, out, e->d_Y, bytes, st), VADC_AMD_EHIP);
   HIP_TRY(hipStreamSynchronize(st), VADC_AMD_EHIP);
   return VADC_AMD_OK;
}

extern "C" int vadc_amd_debug_decoder(vadc_amd_engine *e, const float *x, int n, float *probs)
{
   if (!e || !x || !probs) return fail(VADC_AMD_EINVAL, "debug_decoder: NULL argument");
   if (e->model == VADC_AMD_MODEL_V5) return fail(VADC_AMD_EINVAL, "debug_decoder: no stage taps for the Silero v5 path");
   if (n <= 0 || (size_t)n > e->max_items || n > e->max_streams) return fail(VADC_AMD_EINVAL, "debug_decoder: n=%d out of range (one item per stream slot)", n);
   if (!e->lstm_h3_ok) return fail(VADC_AMD_EINVAL, "debug_decoder: a tap of k_lstm_layer, which this engine does not run (an LSTM weight outside fp16's range)");
   HIP_TRY(hipSetDevice(e->device), VADC_AMD_EHIP);
   { int rc_ = wait_all_prior(e); if (rc_) return rc_; }
   hipStream_t st = e->stream;
   HIP_TRY(host_to_device(e, e->d_tap, x, (size_t)n * 64 * e->lstm_steps * sizeof(float), st), VADC_AMD_EHIP);
   launch_lstm_decoder_tap(e->d_tap, e->lstm, e->d_probs, n, st, e->model, e->lstm_steps);
   HIP_TRY(hipGetLastError(), VADC_AMD_EHIP);
   HIP_TRY(device_to_host(e, probs, e->d_probs, (size_t)n * 2 * sizeof(float), st), VADC_AMD_EHIP);
   HIP_TRY(hipStreamSynchronize(st), VADC_AMD_EHIP);
   return VADC_AMD_OK;
}

extern "C" int vadc_amd_debug_lstm_decoder(vadc_amd_engine *e, const float *x, int n_streams, int n_chunks, float *probs)
{
   int rc = check_shape(e, n_streams, n_chunks, "debug_lstm_decoder");
   if (rc) return rc;
   if (!x || !probs) return fail(VADC_AMD_EINVAL, "debug_lstm_decoder: NULL buffer");
   if (e->model == VADC_AMD_MODEL_V5) return fail(VADC_AMD_EINVAL, "debug_lstm_decoder: no stage taps for the Silero v5 path");
   HIP_TRY(hipSetDevice(e->device), VADC_AMD_EHIP);
   { int rc_ = wait_all_prior(e); if (rc_) return rc_; }
   hipStream_t st = e->stream;
   const size_t n = (size_t)n_streams * n_chunks;
   const int lk = resolve_lstm(e, n_streams);
   {
      // pure data movement: reference layout [S][C][64][steps] -> LSTM-native tiles (common.h), fp32 or split fp16
      const size_t padded = (size_t)((n_streams + kLstmTile - 1) / kLstmTile) * kLstmTile;
      const int TS = e->lstm_steps;
      if (lk >= 6) {
         std::vector<_Float16> tiles(padded * n_chunks * 64 * TS * 2, (_Float16)0.0f);
         for (int s = 0; s < n_streams; ++s)
            for (int c = 0; c < n_chunks; ++c)
               for (int u = 0; u < 64; ++u)
                  for (int t = 0; t < TS; ++t) {
                     const float v = x[(((size_t)s * n_chunks + c) * 64 + u) * TS + t];
                     const _Float16 hi = (_Float16)v;
                     const size_t i = lstm_xh_index(s, c, n_chunks, t, u, TS);
                     tiles[i] = hi;
                     tiles[i + kLstmTile * 64] = (_Float16)(v - (float)hi);
                  }
         HIP_TRY(upload(e->d_act[3], tiles.data(), tiles.size() * sizeof(_Float16)), VADC_AMD_EHIP);
      } else {
         std::vector<float> tiles(padded * n_chunks * 64 * TS, 0.0f);
         for (int s = 0; s < n_streams; ++s)
            for (int c = 0; c < n_chunks; ++c)
               for (int u = 0; u < 64; ++u)
                  for (int t = 0; t < TS; ++t)
                     tiles[lstm_x_index(s, c, n_chunks, t, u, TS)] = x[(((size_t)s * n_chunks + c) * 64 + u) * TS + t];
         HIP_TRY(upload(e->d_act[3], tiles.data(), tiles.size() * sizeof(float)), VADC_AMD_EHIP);
      }
   }
   launch_lstm_on(e, lk, e->d_probs, n_streams, n_chunks, 0, n_chunks, st);
   HIP_TRY(hipGetLastError(), VADC_AMD_EHIP);
   HIP_TRY(device_to_host(e, probs, e->d_probs, n * 2 * sizeof(float), st), VADC_AMD_EHIP);
   HIP_TRY(hipStreamSynchronize(st), VADC_AMD_EHIP);
   return VADC_AMD_OK;
}
