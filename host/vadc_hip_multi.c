/*
 * vadc_hip_multi.c -- the multi-GPU host in C: one engine per visible MI355X, one host thread per engine, RCCL only for the final gather.
 *
 * The north star's partitioning (BASELINE.json): streams are independent, so the batch is cut into contiguous blocks --
 * global stream s lives on GPU s / streams_per_gpu -- weights are replicated, no data-path collective exists, and the one exchange is the gather of
 * the per-chunk speech probabilities on device 0: ncclGather of [streams_per_gpu x chunks] fp32 per step (4 B per chunk: element 1 of the engine's
 * pair, packed by vadc_amd_speech_probabilities), issued on a side stream behind vadc_amd_join (a device-side wait), so that step k's gather runs
 * beside step k + 1's kernels.  Counterpart of the reference's single-engine
 * main (vadc.c:1127-1276) for N engines; the per-engine calls are the ones vadc.c:56-103 makes (backend_run on a window of chunks).
 *
 *   vadc_hip_multi --model weights.testtensor [--gpus N | --devices a,b,...] [--streams-per-gpu S] [--chunks C] [--steps K] [--warmup W]
 *                  [--pcm in.s16 --dump out.f32]
 *   --devices  the HIP devices to use, in rank order (rank r = the r-th entry; the gather lands on the first); default: devices 0 .. N - 1
 *   --pcm   s16le [N * S][K * C * 1536]: step k feeds every stream its k-th window of C chunks, from reset state (warm-up steps are not run)
 *   --dump  the gathered speech probabilities of every step as float32 [K][N * S][C]  (what the tests compare with the CPU oracle)
 *   without --pcm: synthetic tones + noise, resident in HBM before the timed region
 *   stdout: ONE JSON line with the fields of bench.py's line (value = audio-seconds per second over ALL GPUs under a label that says so, value_per_gpu,
 *           total_streams, and "rccl": what the communicator itself reports -- ncclCommCount, ncclGetVersion -- and the bytes the gather ships).
 * Host code is C; the only GPU code is inside libvadc_amd.so (HIP runtime and RCCL are called through their C APIs).
 */
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include "vadc_amd.h"

#define CHUNK VADC_AMD_CHUNK_SAMPLES
#define NBUF 3                         /* step buffers used in turn */

/* A rank that fails raises the shared flag: the others stop issuing gathers (a gather whose peer never comes would never complete), abort their communicator
 * instead of waiting for what they have already enqueued, and the process exits non-zero instead of hanging. */
#define FAIL(code) do { w->rc = (code); __atomic_store_n(w->failed, 1, __ATOMIC_RELEASE); goto out; } while (0)
#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "rank %d: %s failed: %s\n", w->rank, #x, hipGetErrorString(e_)); FAIL(2); } } while (0)
#define NCCL_OK(x) do { ncclResult_t r_ = (x); if (r_ != ncclSuccess) { fprintf(stderr, "rank %d: %s failed: %s\n", w->rank, #x, ncclGetErrorString(r_)); FAIL(3); } } while (0)
#define ENG_OK(x) do { if ((x) != VADC_AMD_OK) { fprintf(stderr, "rank %d: %s failed: %s\n", w->rank, #x, vadc_amd_last_error()); FAIL(4); } } while (0)

typedef struct {
   int rank, n_ranks, device, S, C, K, W, rc;
   const void *blob; size_t blob_len;
   const int16_t *pcm;                 /* host, [n_ranks * S][K * C * CHUNK] or NULL */
   float *dump;                        /* host (rank 0), [K][n_ranks * S][C] or NULL */
   ncclComm_t comm;
   int *failed;                        /* shared: some rank has failed */
   int aborted;                        /* this rank's communicator was aborted (not to be destroyed) */
   pthread_barrier_t *bar;
   double t_begin, t_end;              /* rank 0: the timed region */
} Worker;

static double now_s(void)
{
   struct timespec t;
   clock_gettime(CLOCK_MONOTONIC, &t);
   return t.tv_sec + 1e-9 * t.tv_nsec;
}

/* synthetic stream: a tone whose level follows a slow envelope (speech-like bursts) + a little noise; deterministic per global stream */
static void synth_stream(int16_t *dst, size_t n, unsigned seed)
{
   unsigned s = seed * 2654435761u + 12345u;
   const int period = 40 + (int)(seed % 97);
   for (size_t i = 0; i < n; ++i) {
      s = s * 1664525u + 1013904223u;
      const int noise = (int)(s >> 20) - 2048;                                   /* +-2048 */
      const int burst = ((i / 12000 + seed) % 3) != 0;                           /* 0.75 s on, 0.75 s off, phase by stream */
      const int tri = (int)(i % (size_t)period) * 2 - period;                    /* triangle wave */
      int v = burst ? tri * (12000 / period) + noise / 8 : noise / 64;
      dst[i] = (int16_t)(v > 32767 ? 32767 : (v < -32768 ? -32768 : v));
   }
}

/* wait for a stream without blocking in the runtime: a peer's failure must be able to end the wait.  1 = drained, 0 = given up (communicator aborted) */
static int drain_or_abort(Worker *w, hipStream_t s)
{
   for (;;) {
      const hipError_t q = hipStreamQuery(s);
      if (q == hipSuccess) return 1;
      if (q != hipErrorNotReady) return 0;
      if (__atomic_load_n(w->failed, __ATOMIC_ACQUIRE)) {
         if (!w->aborted) { (void)ncclCommAbort(w->comm); w->aborted = 1; }
         return 0;
      }
      usleep(100);
   }
}

static void *worker_main(void *arg)
{
   Worker *w = (Worker *)arg;
   vadc_amd_engine *eng = NULL;
   hipStream_t st = NULL, sg = NULL;
   hipEvent_t ev_g[NBUF] = {0};
   int16_t *d_in[NBUF] = {0};
   int16_t **d_in_all = NULL;          /* --pcm: one resident buffer per step */
   float *d_probs[NBUF] = {0}, *d_speech[NBUF] = {0}, *d_gather[NBUF] = {0};
   const size_t step_samples = (size_t)w->S * w->C * CHUNK, step_probs = (size_t)w->S * w->C;      /* gathered: the speech probability alone, 4 B per chunk */
   int16_t *h_tmp = NULL;
   int waits = 0;                      /* barrier waits done: every rank makes exactly two, whatever happens to it */

   HIP_OK(hipSetDevice(w->device));
   ENG_OK(vadc_amd_create(w->blob, w->blob_len, w->device, w->S, w->C, VADC_AMD_PRECISION_FP32, &eng));
   ENG_OK(vadc_amd_set_option(eng, "defer_join", 1));
   ENG_OK(vadc_amd_set_option(eng, "graph", 1));
   HIP_OK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
   HIP_OK(hipStreamCreateWithFlags(&sg, hipStreamNonBlocking));
   for (int b = 0; b < NBUF; ++b) {
      HIP_OK(hipEventCreateWithFlags(&ev_g[b], hipEventDisableTiming));
      HIP_OK(hipMalloc((void **)&d_probs[b], step_probs * 2 * sizeof(float)));      /* the engine's output: [S][C][2] */
      HIP_OK(hipMalloc((void **)&d_speech[b], step_probs * sizeof(float)));
      if (w->rank == 0) HIP_OK(hipMalloc((void **)&d_gather[b], step_probs * w->n_ranks * sizeof(float)));
   }
   if (w->pcm) {
      /* verification mode: every step's window of every local stream resident before the first call */
      d_in_all = (int16_t **)calloc((size_t)w->K, sizeof(int16_t *));
      h_tmp = (int16_t *)malloc(step_samples * sizeof(int16_t));
      if (!d_in_all || !h_tmp) FAIL(5);
      const size_t per_stream = (size_t)w->K * w->C * CHUNK;
      for (int k = 0; k < w->K; ++k) {
         HIP_OK(hipMalloc((void **)&d_in_all[k], step_samples * sizeof(int16_t)));
         for (int s = 0; s < w->S; ++s)
            memcpy(h_tmp + (size_t)s * w->C * CHUNK, w->pcm + ((size_t)w->rank * w->S + s) * per_stream + (size_t)k * w->C * CHUNK, (size_t)w->C * CHUNK * sizeof(int16_t));
         HIP_OK(hipMemcpy(d_in_all[k], h_tmp, step_samples * sizeof(int16_t), hipMemcpyHostToDevice));
      }
   } else {
      h_tmp = (int16_t *)malloc(step_samples * sizeof(int16_t));
      if (!h_tmp) FAIL(5);
      for (int b = 0; b < NBUF; ++b) {
         HIP_OK(hipMalloc((void **)&d_in[b], step_samples * sizeof(int16_t)));
         for (int s = 0; s < w->S; ++s) synth_stream(h_tmp + (size_t)s * w->C * CHUNK, (size_t)w->C * CHUNK, (unsigned)((w->rank * w->S + s) * NBUF + b));
         HIP_OK(hipMemcpy(d_in[b], h_tmp, step_samples * sizeof(int16_t), hipMemcpyHostToDevice));
      }
   }
   HIP_OK(hipDeviceSynchronize());

   for (int phase = 0; phase < 2; ++phase) {                     /* 0 = warm-up (untimed), 1 = the K timed steps */
      const int n = phase == 0 ? (w->pcm ? 0 : w->W) : w->K;
      if (phase == 1) {
         HIP_OK(hipDeviceSynchronize());
         pthread_barrier_wait(w->bar); ++waits;
         if (w->rank == 0) w->t_begin = now_s();
      }
      for (int i = 0; i < n; ++i) {
         if (__atomic_load_n(w->failed, __ATOMIC_ACQUIRE)) { if (!w->rc) w->rc = 6; goto out; }      /* a peer has failed: issue nothing more */
         const int b = i % NBUF;
         const int16_t *in = w->pcm ? d_in_all[i] : d_in[b];
         /* this step's probability buffer was last read by the gather of step i - NBUF: the call (all its internal streams) waits for that */
         if (i >= NBUF) HIP_OK(hipStreamWaitEvent(st, ev_g[b], 0));
         ENG_OK(vadc_amd_run_device_s16(eng, in, w->S, w->C, d_probs[b], st));
         ENG_OK(vadc_amd_join(eng, sg));                         /* device-side: the side stream continues when this call's probabilities are complete */
         ENG_OK(vadc_amd_speech_probabilities(eng, d_probs[b], w->S, w->C, d_speech[b], sg));
         NCCL_OK(ncclGather(d_speech[b], d_gather[b], step_probs, ncclFloat, 0, w->comm, sg));
         HIP_OK(hipEventRecord(ev_g[b], sg));
         if (phase == 1 && w->dump && w->rank == 0) {            /* verification mode only: every step's gathered block to the host */
            if (!drain_or_abort(w, sg)) FAIL(6);
            HIP_OK(hipMemcpy(w->dump + (size_t)i * step_probs * w->n_ranks, d_gather[b], step_probs * w->n_ranks * sizeof(float), hipMemcpyDeviceToHost));
         }
      }
      if (!drain_or_abort(w, sg)) FAIL(6);
      ENG_OK(vadc_amd_synchronize(eng));
      if (phase == 1) {
         pthread_barrier_wait(w->bar); ++waits;                  /* the slowest rank ends the region */
         if (w->rank == 0) w->t_end = now_s();
      }
   }
out:
   while (waits < 2) { pthread_barrier_wait(w->bar); ++waits; }   /* a rank that failed must not leave the others in a barrier */
   if (w->rc && !w->aborted && w->comm) { (void)ncclCommAbort(w->comm); w->aborted = 1; }      /* whatever this rank still has enqueued on the communicator ends here */
   if (st) (void)hipStreamSynchronize(st);
   if (sg && !w->rc) (void)hipStreamSynchronize(sg);
   for (int b = 0; b < NBUF; ++b) {
      if (d_in[b]) (void)hipFree(d_in[b]);
      if (d_probs[b]) (void)hipFree(d_probs[b]);
      if (d_speech[b]) (void)hipFree(d_speech[b]);
      if (d_gather[b]) (void)hipFree(d_gather[b]);
      if (ev_g[b]) (void)hipEventDestroy(ev_g[b]);
   }
   if (d_in_all) { for (int k = 0; k < w->K; ++k) if (d_in_all[k]) (void)hipFree(d_in_all[k]); free(d_in_all); }
   free(h_tmp);
   if (eng) vadc_amd_destroy(eng);
   if (st) (void)hipStreamDestroy(st);
   if (sg) (void)hipStreamDestroy(sg);
   return NULL;
}

static void *read_file(const char *path, size_t *len)
{
   FILE *f = fopen(path, "rb");
   if (!f) return NULL;
   fseek(f, 0, SEEK_END);
   long n = ftell(f);
   fseek(f, 0, SEEK_SET);
   void *p = n > 0 ? malloc((size_t)n) : NULL;
   if (p && fread(p, 1, (size_t)n, f) != (size_t)n) { free(p); p = NULL; }
   fclose(f);
   *len = p ? (size_t)n : 0;
   return p;
}

int main(int argc, char **argv)
{
   const char *model = NULL, *pcm_path = NULL, *dump_path = NULL, *dev_list = NULL;
   int gpus = 0, S = 256, C = 96, K = 20, W = 5;
   for (int i = 1; i < argc; ++i) {
      const char *a = argv[i], *v = i + 1 < argc ? argv[i + 1] : NULL;
      if (!strcmp(a, "--model") && v) { model = v; ++i; }
      else if (!strcmp(a, "--gpus") && v) { gpus = atoi(v); ++i; }
      else if (!strcmp(a, "--devices") && v) { dev_list = v; ++i; }
      else if (!strcmp(a, "--streams-per-gpu") && v) { S = atoi(v); ++i; }
      else if (!strcmp(a, "--chunks") && v) { C = atoi(v); ++i; }
      else if (!strcmp(a, "--steps") && v) { K = atoi(v); ++i; }
      else if (!strcmp(a, "--warmup") && v) { W = atoi(v); ++i; }
      else if (!strcmp(a, "--pcm") && v) { pcm_path = v; ++i; }
      else if (!strcmp(a, "--dump") && v) { dump_path = v; ++i; }
      else { fprintf(stderr, "usage: %s --model weights.testtensor [--gpus N | --devices a,b,...] [--streams-per-gpu S] [--chunks C] [--steps K] [--warmup W] [--pcm in.s16 --dump out.f32]\n", argv[0]); return 1; }
   }
   if (!model || S <= 0 || C <= 0 || K <= 0 || W < 0) { fprintf(stderr, "vadc_hip_multi: --model is required; S, C, K > 0\n"); return 1; }
   int ndev = 0;
   if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { fprintf(stderr, "vadc_hip_multi: no HIP device (this host has no CPU path)\n"); return 2; }
   int devs_arg[64], n_arg = 0;
   if (dev_list) {                                               /* an explicit list: every entry a visible device, none twice */
      for (const char *p = dev_list; *p && n_arg < 64;) {
         char *end;
         const long d = strtol(p, &end, 10);
         if (end == p || d < 0 || d >= ndev) { fprintf(stderr, "vadc_hip_multi: --devices %s: '%s' is not one of the %d visible device(s) 0 .. %d\n", dev_list, p, ndev, ndev - 1); return 1; }
         for (int j = 0; j < n_arg; ++j) if (devs_arg[j] == (int)d) { fprintf(stderr, "vadc_hip_multi: --devices %s names device %ld twice\n", dev_list, d); return 1; }
         devs_arg[n_arg++] = (int)d;
         p = *end == ',' ? end + 1 : end;
         if (*end && *end != ',') { fprintf(stderr, "vadc_hip_multi: --devices wants a comma-separated list of device numbers\n"); return 1; }
      }
      if (gpus > 0 && gpus != n_arg) { fprintf(stderr, "vadc_hip_multi: --gpus %d but --devices lists %d\n", gpus, n_arg); return 1; }
      gpus = n_arg;
   }
   if (gpus <= 0) gpus = ndev;
   if (gpus > ndev) { fprintf(stderr, "vadc_hip_multi: %d GPUs asked for, but only %d device(s) are visible (HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES?)\n", gpus, ndev); return 1; }
   size_t blob_len = 0, pcm_len = 0;
   void *blob = read_file(model, &blob_len);
   if (!blob) { fprintf(stderr, "vadc_hip_multi: cannot read %s\n", model); return 1; }
   int16_t *pcm = NULL;
   if (pcm_path) {
      pcm = (int16_t *)read_file(pcm_path, &pcm_len);
      const size_t want = (size_t)gpus * S * K * C * CHUNK * sizeof(int16_t);
      if (!pcm || pcm_len != want) { fprintf(stderr, "vadc_hip_multi: %s must hold [%d][%d] s16 samples (%zu bytes)\n", pcm_path, gpus * S, K * C * CHUNK, want); return 1; }
   }
   float *dump = NULL;
   const size_t dump_floats = (size_t)K * gpus * S * C;
   if (dump_path) { dump = (float *)malloc(dump_floats * sizeof(float)); if (!dump) { free(blob); free(pcm); return 5; } }

   int rc = 0, failed = 0, comms_up = 0;
   ncclComm_t *comms = (ncclComm_t *)calloc((size_t)gpus, sizeof(ncclComm_t));
   int *devs = (int *)calloc((size_t)gpus, sizeof(int));
   Worker *ws = (Worker *)calloc((size_t)gpus, sizeof(Worker));
   pthread_t *th = (pthread_t *)calloc((size_t)gpus, sizeof(pthread_t));
   pthread_barrier_t bar;
   int bar_up = 0, started = 0;
   if (!comms || !devs || !ws || !th) { rc = 5; goto done; }
   for (int r = 0; r < gpus; ++r) devs[r] = dev_list ? devs_arg[r] : r;
   {
      ncclResult_t nr = ncclCommInitAll(comms, gpus, devs);       /* one process, one communicator per device: rank r = devs[r] */
      if (nr != ncclSuccess) {
         fprintf(stderr, "vadc_hip_multi: ncclCommInitAll over %d device(s) failed: %s (%d device(s) are visible)\n", gpus, ncclGetErrorString(nr), ndev);
         rc = 3; goto done;
      }
      comms_up = 1;
   }
   pthread_barrier_init(&bar, NULL, (unsigned)gpus);
   bar_up = 1;
   for (int r = 0; r < gpus; ++r) {
      ws[r] = (Worker){.rank = r, .n_ranks = gpus, .device = devs[r], .S = S, .C = C, .K = K, .W = W, .rc = 0, .blob = blob, .blob_len = blob_len,
                       .pcm = pcm, .dump = dump, .comm = comms[r], .failed = &failed, .aborted = 0, .bar = &bar};
      if (pthread_create(&th[r], NULL, worker_main, &ws[r]) != 0) {
         fprintf(stderr, "vadc_hip_multi: pthread_create failed\n");
         /* the ranks already running wait in a barrier sized for all of them: nothing sane is left but to leave */
         rc = 5; __atomic_store_n(&failed, 1, __ATOMIC_RELEASE); _exit(5);
      }
      ++started;
   }
   for (int r = 0; r < started; ++r) { pthread_join(th[r], NULL); if (ws[r].rc && (!rc || ws[r].rc != 6)) rc = ws[r].rc; }
   if (rc) { fprintf(stderr, "vadc_hip_multi: failed (rc %d)\n", rc); goto done; }
   if (dump_path) {
      FILE *f = fopen(dump_path, "wb");
      if (!f || fwrite(dump, sizeof(float), dump_floats, f) != dump_floats) { fprintf(stderr, "vadc_hip_multi: cannot write %s\n", dump_path); rc = 1; }
      if (f) fclose(f);
      if (rc) goto done;
   }
   {
   const double wall = ws[0].t_end - ws[0].t_begin;
   const double audio_s = (double)gpus * S * C * K * (CHUNK / 16000.0);
   const double value = wall > 0 ? audio_s / wall : 0.0;
   int comm_count = 0, nccl_ver = 0;
   (void)ncclCommCount(comms[0], &comm_count);                   /* what the communicator reports, not what the command line asked for */
   (void)ncclGetVersion(&nccl_ver);
   printf("{\"metric\": \"audio-seconds/sec (= real-time streams), WHOLE JOB over %d GPU(s) (value_per_gpu = value / %d), Silero v3.1 16k\", \"value\": %.1f, \"value_per_gpu\": %.1f, "
          "\"total_streams\": %d, \"unit\": \"audio-seconds/sec\", \"n_gpus\": %d, \"steps\": %d, \"warmup\": %d, "
          "\"ms_per_step\": %.4f, \"higher_is_better\": true, \"scaling\": \"weak\", \"data\": \"%s\", \"host\": \"C (host/vadc_hip_multi.c), one thread per GPU, ncclGather to device 0 per step\", "
          "\"rccl\": {\"backend\": \"rccl (ncclCommInitAll, one process)\", \"world_size\": %d, \"nccl_version\": %d, \"collective\": \"ncclGather -> device 0, one per step, on a side stream behind vadc_amd_join\", "
          "\"bytes_per_chunk\": 4, \"bytes_per_rank_and_step\": %zu}, "
          "\"config\": {\"workload\": \"Silero v3.1 16k, %d streams/GPU x %d chunks per step, contiguous stream blocks\", \"streams_per_gpu\": %d, \"chunks_per_step\": %d}}\n",
          gpus, gpus, value, value / gpus, gpus * S, gpus, K, pcm ? 0 : W, wall * 1e3 / K, pcm ? "file" : "synthetic", comm_count, nccl_ver, (size_t)S * C * 4, S, C, S, C);
   }
done:
   if (comms_up) for (int r = 0; r < gpus; ++r) if (!ws || !ws[r].aborted) (void)ncclCommDestroy(comms[r]);      /* (an aborted communicator is already gone) */
   if (bar_up) pthread_barrier_destroy(&bar);
   free(blob); free(pcm); free(dump); free(comms); free(devs); free(ws); free(th);
   fflush(NULL);
   {  /* not through the HIP runtime's exit handlers (see the end of vadc_hip.c) -- unless a profiler is to write its files from them */
      extern char **environ;
      const char *pre = getenv("LD_PRELOAD");
      int profiled = pre && strstr(pre, "rocprof");
      for (char **e = environ; e && *e; ++e) if (strncmp(*e, "ROCPROF", 7) == 0 || strncmp(*e, "ROCP_", 5) == 0) profiled = 1;
      if (!profiled) _exit(rc);
   }
   return rc;
}
