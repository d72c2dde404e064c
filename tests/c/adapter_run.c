/* adapter_run.c -- drives the literal backend trio of include/vadc_backend_hip.h the way vadc.c does, on the GPU box.
 *
 *   adapter_run <weights.testtensor> <batch> <sequence_count> <samples.f32> <probs.f32>
 *
 * A C program shaped like the part of run_inference that surrounds the backend (vadc.c:686-795: config defaults, backend_init, probability layout,
 * batch and sequence-count clamping, buffer allocation, backend_create_tensors) and like process_chunks / process_chunks_v5 (vadc.c:56-162: slices of
 * batch x window samples, zero-padded tail, v5 rows of context + window with the context carried by the CALLER, one probability per window read at
 * output[i * output_stride + silero_probability_out_index]).  The types are the layout mirrors of tests/c/vadc_layout_mirror.h (checked against the
 * reference's vadc.h in the build container).  tests/test_gpu_adapter.py compares what it writes with the reference goldens.
 * Build: gcc -std=gnu11 -O1 -Iinclude -Itests/c tests/c/adapter_run.c -Lvadc_amd -lvadc_amd -Wl,-rpath,$PWD/vadc_amd -o tests/c/adapter_run */
#include <unistd.h>
#include "vadc_layout_mirror.h"
#include "vadc_backend_hip.h"

static float *read_f32(const char *path, size_t *count)
{
   size_t len = 0;
   void *p = vadc_hip_read_file(path, &len);
   *count = len / sizeof(float);
   return (float *)p;
}

int main(int argc, char **argv)
{
   if (argc != 6) { fprintf(stderr, "usage: adapter_run weights batch sequence_count samples.f32 probs.f32\n"); return 2; }
   const int preferred_batch_size = atoi(argv[2]);
   const int desired_sequence_count = atoi(argv[3]);
   size_t n_samples = 0;
   float *samples = read_f32(argv[4], &n_samples);
   if (!samples) { fprintf(stderr, "adapter_run: cannot read %s\n", argv[4]); return 2; }

   Silero_Config config = {0};                                  /* vadc.c:686-688 */
   config.batch_size_restriction = 1;
   config.batch_size = 1;
   String8 model_path_arg = {(const int8_t *)argv[1], (int64_t)strlen(argv[1])};
   void *backend = backend_init((MemoryArena *)0, model_path_arg, &config);
   if (!backend) return 1;                                      /* vadc.c:692-695 */
   if (config.is_silero_v5) config.context_size = 64;           /* vadc.c:697-701 (SILERO_V5_CONTEXT_SIZE) */
   if (config.output_dims == 3) { config.silero_probability_out_index = 1; config.output_stride = 2; }      /* vadc.c:703-712 */
   else                         { config.silero_probability_out_index = 0; config.output_stride = 1; }
   config.batch_size = (config.batch_size_restriction == -1) ? preferred_batch_size : config.batch_size_restriction;
   config.prob_tensor_element_count = (size_t)config.batch_size * (config.output_dims == 3 ? 2 : 1);        /* vadc.c:717-741 */
   {
      int sequence_count = desired_sequence_count;             /* vadc.c:743-755 */
      if (sequence_count < config.input_size_min) sequence_count = config.input_size_min;
      if (sequence_count > config.input_size_max) sequence_count = config.input_size_max;
      config.input_count = sequence_count;
   }
   Tensor_Buffers buffers = {0};                                /* vadc.c:772-793 */
   buffers.window_size_samples = config.input_count;
   const size_t row = (size_t)buffers.window_size_samples + (size_t)(config.is_silero_v5 ? config.context_size : 0);
   buffers.input_samples = (float *)calloc(row * (size_t)config.batch_size, sizeof(float));
   buffers.output = (float *)calloc(config.prob_tensor_element_count, sizeof(float));
   buffers.lstm_count = 128;
   buffers.lstm_h = (float *)calloc(128, sizeof(float)); buffers.lstm_c = (float *)calloc(128, sizeof(float));
   buffers.lstm_h_out = (float *)calloc(128, sizeof(float)); buffers.lstm_c_out = (float *)calloc(128, sizeof(float));
   backend_create_tensors(config, backend, buffers);
   VADC_Context context = {backend, buffers};

   const size_t window = (size_t)config.input_count;
   const size_t n_windows = n_samples / window;                 /* whole windows only (vadc.c:964) */
   float *probs = (float *)calloc(n_windows + (size_t)config.batch_size, sizeof(float));
   size_t out = 0;
   const size_t stride = window * (size_t)config.batch_size;
   for (size_t offset = 0; offset < n_windows * window; offset += stride) {
      const size_t left = n_windows * window - offset, take = left > stride ? stride : left;
      if (!config.is_silero_v5) {                               /* process_chunks, vadc.c:66-75 */
         memset(buffers.input_samples, 0, stride * sizeof(float));
         memmove(buffers.input_samples, samples + offset, take * sizeof(float));
      } else {                                                  /* process_chunks_v5, vadc.c:117-138: the caller carries the context */
         const size_t cs = (size_t)config.context_size, total = cs + window;
         float carry[64];
         memcpy(carry, buffers.input_samples + total * (size_t)config.batch_size - cs, cs * sizeof(float));
         memset(buffers.input_samples, 0, total * (size_t)config.batch_size * sizeof(float));
         memcpy(buffers.input_samples, carry, cs * sizeof(float));
         for (size_t b = 0; b * window < take; ++b) {
            if (b > 0) memcpy(buffers.input_samples + b * total, samples + offset + b * window - cs, cs * sizeof(float));
            memcpy(buffers.input_samples + b * total + cs, samples + offset + b * window, window * sizeof(float));
         }
      }
      backend_run((MemoryArena *)0, &context, config);
      for (int i = 0; i < config.batch_size; ++i)               /* vadc.c:94-99 */
         probs[out++] = buffers.output[i * config.output_stride + config.silero_probability_out_index];
   }
   FILE *f = fopen(argv[5], "wb");
   if (!f || fwrite(probs, sizeof(float), n_windows, f) != n_windows) { fprintf(stderr, "adapter_run: cannot write %s\n", argv[5]); return 2; }
   fclose(f);
   fprintf(stderr, "adapter_run: %zu windows of %d samples, batch %d, v5 %d\n", n_windows, config.input_count, config.batch_size, config.is_silero_v5);
   fflush(NULL);
   _exit(0);                                                     /* a short-lived GPU process beside the test runner's own context: not through the HIP runtime's exit handlers (host/vadc_hip.c) */
}
