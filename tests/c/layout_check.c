/* Build-container-only check (the reference tree does not travel): the layout mirrors of tests/c/vadc_layout_mirror.h are the reference's types,
 * field for field.  gcc -fsyntax-only -I/root/reference -Itests/c tests/c/layout_check.c */
#include <stddef.h>
#include "vadc.h"
#define VADC_MIRROR_PREFIX_M
#include "vadc_layout_mirror.h"

#define SAME_FIELD(T, f) _Static_assert(offsetof(T, f) == offsetof(M_##T, f) && sizeof(((T *)0)->f) == sizeof(((M_##T *)0)->f), #T "." #f)
_Static_assert(sizeof(String8) == sizeof(M_String8), "String8");
SAME_FIELD(String8, begin); SAME_FIELD(String8, size);
_Static_assert(sizeof(Silero_Config) == sizeof(M_Silero_Config), "Silero_Config");
SAME_FIELD(Silero_Config, sr_input_index); SAME_FIELD(Silero_Config, batch_size_restriction); SAME_FIELD(Silero_Config, batch_size);
SAME_FIELD(Silero_Config, context_size); SAME_FIELD(Silero_Config, input_count); SAME_FIELD(Silero_Config, prob_shape_count);
SAME_FIELD(Silero_Config, prob_shape); SAME_FIELD(Silero_Config, prob_tensor_element_count); SAME_FIELD(Silero_Config, output_dims);
SAME_FIELD(Silero_Config, silero_probability_out_index); SAME_FIELD(Silero_Config, output_stride); SAME_FIELD(Silero_Config, input_size_min);
SAME_FIELD(Silero_Config, input_size_max); SAME_FIELD(Silero_Config, lstm_hidden_size); SAME_FIELD(Silero_Config, is_silero_v5);
_Static_assert(sizeof(Tensor_Buffers) == sizeof(M_Tensor_Buffers), "Tensor_Buffers");
SAME_FIELD(Tensor_Buffers, window_size_samples); SAME_FIELD(Tensor_Buffers, input_samples); SAME_FIELD(Tensor_Buffers, output);
SAME_FIELD(Tensor_Buffers, lstm_count); SAME_FIELD(Tensor_Buffers, lstm_h); SAME_FIELD(Tensor_Buffers, lstm_c);
SAME_FIELD(Tensor_Buffers, lstm_h_out); SAME_FIELD(Tensor_Buffers, lstm_c_out);
_Static_assert(sizeof(VADC_Context) == sizeof(M_VADC_Context), "VADC_Context");
SAME_FIELD(VADC_Context, backend); SAME_FIELD(VADC_Context, buffers);
