/* vadc_layout_mirror.h -- LAYOUT MIRRORS of the five reference types the backend trio touches, so that tests/c/adapter_run.c can drive
 * include/vadc_backend_hip.h on a GPU box where the reference tree does not exist.  Declarations only (field order, names and widths of
 * vadc.h:10-43 Silero_Config, :45-58 Tensor_Buffers, :65-70 VADC_Context; string8.h:10-15 String8; MemoryArena is only ever passed by
 * pointer).  tests/c/layout_check.c static_asserts sizeof / offsetof equality against the reference's own vadc.h in the build container
 * (tests/test_abi.py::test_layout_mirrors_match_the_reference_headers).
 * Define VADC_MIRROR_PREFIX_M before including to get the types as M_<name> (what layout_check.c does, next to the real ones). */
#ifndef VADC_LAYOUT_MIRROR_H
#define VADC_LAYOUT_MIRROR_H
#include <stddef.h>
#include <stdint.h>

#ifdef VADC_MIRROR_PREFIX_M
#define MIRROR(name) M_##name
#else
#define MIRROR(name) name
#endif

typedef struct MIRROR(MemoryArena) MIRROR(MemoryArena);          /* memory.h: opaque here, passed through untouched */

typedef struct MIRROR(String8) MIRROR(String8);                  /* string8.h:10-15 */
struct MIRROR(String8) {
   const int8_t *begin;
   int64_t size;
};

typedef struct MIRROR(Silero_Config) MIRROR(Silero_Config);      /* vadc.h:10-43 */
struct MIRROR(Silero_Config) {
   int32_t sr_input_index;
   int32_t batch_size_restriction;
   int32_t batch_size;
   int32_t context_size;
   int32_t input_count;
   size_t prob_shape_count;
   int64_t prob_shape[4];
   size_t prob_tensor_element_count;
   int32_t output_dims;
   int32_t silero_probability_out_index;
   int32_t output_stride;
   int32_t input_size_min;
   int32_t input_size_max;
   int32_t lstm_hidden_size;
   int32_t is_silero_v5;
};

typedef struct MIRROR(Tensor_Buffers) MIRROR(Tensor_Buffers);    /* vadc.h:45-58 */
struct MIRROR(Tensor_Buffers) {
   int window_size_samples;
   float *input_samples;
   float *output;
   int lstm_count;
   float *lstm_h;
   float *lstm_c;
   float *lstm_h_out;
   float *lstm_c_out;
};

typedef struct MIRROR(VADC_Context) MIRROR(VADC_Context);        /* vadc.h:65-70 */
struct MIRROR(VADC_Context) {
   void *backend;
   MIRROR(Tensor_Buffers) buffers;
};

#endif /* VADC_LAYOUT_MIRROR_H */
