/* Compile-only check that include/vadc_backend_hip.h slots into a vadc-shaped translation unit: it is
 * compiled against the REAL vadc.h of the reference when that is present (build container only). */
#include "vadc.h"
#include "vadc_backend_hip.h"

int adapter_check(MemoryArena *arena, VADC_Context *ctx)
{
   Silero_Config config = {0};
   String8 model = {0};
   void *b = backend_init(arena, model, &config);
   if (!b) return -1;
   ctx->backend = b;
   config.batch_size = 1;
   backend_create_tensors(config, b, ctx->buffers);
   backend_run(arena, ctx, config);
   return 0;
}
