/*
 * vadc_backend_hip.h -- the MI355X backend in the exact shape of vadc's compile-time backend trio.
 *
 * vadc selects its backend by #including one implementation of three `static inline` functions
 * (vadc.c:15-19):  onnx_helpers.c  (ONNX_INFERENCE_ENABLED=1)  or  silero.h  (the C/AVX2 backend).
 * This header is the third choice: include it INSTEAD of silero.h, after vadc.h, and link libvadc_amd.so.
 *
 *     #if   VADC_BACKEND_HIP
 *     #include "vadc_backend_hip.h"        // this file
 *     #elif ONNX_INFERENCE_ENABLED
 *     #include "onnx_helpers.c"
 *     #else
 *     #include "silero.h"
 *     #endif
 *
 * Functions replaced (same names, arguments and failure behaviour):
 *   backend_init            silero.h:48 (-> silero_init :21-46)      NULL => run_inference returns -1 (vadc.c:692-695)
 *   backend_create_tensors  silero.h:76                              no-op, as for the C backend
 *   backend_run             silero.h:53-74                           reads buffers.input_samples[batch*1536] f32 (a Silero v5 container:
 *                                                                    rows of 64 + 512, vadc.c:105-162 -- the windows are taken, the context
 *                                                                    is device state), writes buffers.output[batch*2] (prob at index 1)
 * Types used from vadc.h: MemoryArena, String8, Silero_Config (:10-43), Tensor_Buffers (:45-58), VADC_Context (:65-70).
 *
 * Weights: the reference embeds silero_v31_16k.testtensor as a C array (silero.h:19,28; cembed.c).  Here either
 *   - define VADC_HIP_EMBEDDED_WEIGHTS and #include the same generated array before this header, or
 *   - pass the path of the .testtensor file as --model (model_path_arg); default "silero_v31_16k.testtensor".
 * LSTM state lives on the device (C-backend convention), one stream (stream 0).
 */
#ifndef VADC_BACKEND_HIP_H
#define VADC_BACKEND_HIP_H

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "vadc_amd.h"

#ifndef VADC_HIP_MAX_BATCH
#define VADC_HIP_MAX_BATCH 4096          /* vadc's --batch default is 96 (vadc.c:1116) */
#endif

static void *vadc_hip_read_file(const char *path, size_t *len)
{
   FILE *f = fopen(path, "rb");
   if (!f) return 0;
   fseek(f, 0, SEEK_END);
   long n = ftell(f);
   fseek(f, 0, SEEK_SET);
   void *buf = malloc((size_t)n);
   if (buf && fread(buf, 1, (size_t)n, f) != (size_t)n) { free(buf); buf = 0; }
   fclose(f);
   *len = (size_t)n;
   return buf;
}

static inline void *backend_init(MemoryArena *arena, String8 model_path_arg, Silero_Config *config)
{
   (void)arena;
   vadc_amd_engine *engine = 0;
   int rc;
   if (vadc_amd_abi_version() != VADC_AMD_ABI_VERSION) {         /* this header and the library it is linked with are two revisions of the interface */
      fprintf(stderr, "vadc_backend_hip: libvadc_amd.so speaks revision %d of vadc_amd.h, this host was compiled with revision %d\n", vadc_amd_abi_version(), VADC_AMD_ABI_VERSION);
      return 0;
   }
#ifdef VADC_HIP_EMBEDDED_WEIGHTS
   (void)model_path_arg;
   rc = vadc_amd_create(silero_v31_16k_weights, sizeof(silero_v31_16k_weights), -1, 1, VADC_HIP_MAX_BATCH,
                        VADC_AMD_PRECISION_FP32, &engine);
#else
   char path[4096] = "silero_v31_16k.testtensor";
   if (model_path_arg.size > 0 && (size_t)model_path_arg.size < sizeof(path)) {
      memcpy(path, model_path_arg.begin, (size_t)model_path_arg.size);
      path[model_path_arg.size] = 0;
   }
   size_t len = 0;
   void *blob = vadc_hip_read_file(path, &len);
   if (!blob) {
      fprintf(stderr, "vadc_backend_hip: cannot read weights '%s'\n", path);
      return 0;
   }
   rc = vadc_amd_create(blob, len, -1, 1, VADC_HIP_MAX_BATCH, VADC_AMD_PRECISION_FP32, &engine);
   free(blob);
#endif
   if (rc != VADC_AMD_OK) {
      fprintf(stderr, "vadc_backend_hip: %s\n", vadc_amd_last_error());
      return 0;                                   /* => run_inference returns -1, vadc.c:692-695 */
   }
   vadc_amd_caps caps;
   memset(&caps, 0, sizeof caps);
   vadc_amd_get_caps_sized(engine, &caps, sizeof caps);           /* never more than THIS revision's struct, whatever the library's */
   config->batch_size_restriction = caps.batch_size_restriction;     /* silero.h:39 */
   config->is_silero_v5 = caps.is_silero_v5;                         /* silero.h:40 */
   config->input_size_min = caps.input_size_min;                     /* silero.h:41 */
   config->input_size_max = caps.input_size_max;                     /* silero.h:42 */
   config->output_dims = caps.output_dims;                           /* silero.h:43 */
   config->lstm_hidden_size = caps.lstm_hidden_size;
   return engine;
}

/* The caller picks the window after backend_init -- `--sequence_count` clamped to [input_size_min, input_size_max] (vadc.c:743-752) -- and sizes
 * buffers.input_samples as input_count x batch floats (vadc.c:773-781).  The engine must run THAT window: the Silero v4 graph takes 512 ... 1536
 * samples (onnx_helpers.c:164-170), of which this engine serves every multiple of 64 (8 kHz branch: 256 ... 768).  Any other size aborts with a
 * message, like an onnxruntime error would (onnx_helpers.h:5-14) -- never a silent read past the caller's buffer. */
static void vadc_hip_sync_window(vadc_amd_engine *engine, int input_count)
{
   vadc_amd_caps caps;
   if (vadc_amd_get_caps(engine, &caps) != VADC_AMD_OK || caps.window_samples == input_count) return;
   if (vadc_amd_set_option(engine, "window", input_count) != VADC_AMD_OK) {
      fprintf(stderr, "vadc_backend_hip: --sequence_count %d is not a window this backend runs: %s\n", input_count, vadc_amd_last_error());
      abort();
   }
}

static inline void backend_run(MemoryArena *arena, void *context_, Silero_Config config)
{
   (void)arena;
   VADC_Context *context = (VADC_Context *)context_;
   const float *in = context->buffers.input_samples;
   if (config.is_silero_v5 && config.context_size > 0) {
      /* process_chunks_v5 (vadc.c:105-162) hands over rows of context_size + input_count samples -- the previous window's tail in front of every
       * window.  The engine keeps each stream's context on the device (the same 64 samples, zeros before the first window), so only the windows
       * travel: compact the rows (single caller thread, vadc.c is not re-entrant: one static scratch buffer). */
      static float *win = 0;
      static size_t win_cap = 0;
      const size_t need = (size_t)config.batch_size * (size_t)config.input_count;
      if (need > win_cap) { free(win); win = (float *)malloc(need * sizeof(float)); win_cap = win ? need : 0; }
      if (!win) { fprintf(stderr, "vadc_backend_hip: out of memory\n"); abort(); }
      for (int b = 0; b < config.batch_size; ++b)
         memcpy(win + (size_t)b * config.input_count, in + (size_t)b * (config.context_size + config.input_count) + config.context_size,
                (size_t)config.input_count * sizeof(float));
      in = win;
   }
   vadc_hip_sync_window((vadc_amd_engine *)context->backend, config.input_count);      /* a no-op after backend_create_tensors; one caps query */
   /* `batch_size` consecutive windows of the one stream: silero.h:64-68, lstm.c:275-277 */
   int rc = vadc_amd_run_f32((vadc_amd_engine *)context->backend, in, 1, config.batch_size, context->buffers.output);
   if (rc != VADC_AMD_OK) {                        /* the ORT backend aborts on error (onnx_helpers.h:5-14) */
      fprintf(stderr, "vadc_backend_hip: %s\n", vadc_amd_last_error());
      abort();
   }
}

static inline void backend_create_tensors(Silero_Config config, void *backend, Tensor_Buffers buffers)
{
   (void)buffers;                                  /* silero.h:76-81: nothing to bind, buffers are copied per run */
   vadc_hip_sync_window((vadc_amd_engine *)backend, config.input_count);
}

#endif /* VADC_BACKEND_HIP_H */
