// hipcc --offload-arch=gfx950 -O2 -o tools/pk_opsel_probe tools/pk_opsel_probe.hip
// What op_sel / op_sel_hi / neg_lo / neg_hi do on v_pk_mul_f32 (VGPR pair x SGPR pair) and v_pk_add_f32 (VGPR pairs) on gfx950: the forms k_frontend_ri uses.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2v __attribute__((ext_vector_type(2)));
__global__ void k(const float *in, float *out)
{
   const f2v x = {in[0], in[1]};
   f2v kk;
   { float a = in[2], b = in[3]; kk.x = __builtin_amdgcn_readfirstlane(a); kk.y = __builtin_amdgcn_readfirstlane(b); }
   const f2v P = {in[4], in[5]}, Q = {in[6], in[7]};
   f2v r0, r1, r2, r3;
   asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]" : "=v"(r0) : "v"(x), "s"(kk));
   asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1]" : "=v"(r1) : "v"(x), "s"(kk));
   asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r2) : "v"(P), "v"(Q));
   asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r3) : "v"(P), "v"(Q));
   if (threadIdx.x == 0) { out[0] = r0.x; out[1] = r0.y; out[2] = r1.x; out[3] = r1.y; out[4] = r2.x; out[5] = r2.y; out[6] = r3.x; out[7] = r3.y; }
}
int main()
{
   float h[8] = {2, 3, 5, 7, 100, 200, 1, 10}, o[8], *d, *e;
   hipMalloc(&d, 32); hipMalloc(&e, 32); hipMemcpy(d, h, 32, hipMemcpyHostToDevice);
   hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, e);
   hipMemcpy(o, e, 32, hipMemcpyDeviceToHost);
   printf("x = (2, 3), k = (5, 7):  mul [0,0][0,1] -> (%g, %g) want (10, 14);  mul [1,0][1,1] -> (%g, %g) want (15, 21)\n", o[0], o[1], o[2], o[3]);
   printf("P = (100, 200), Q = (1, 10):  add swap neg_lo -> (%g, %g) want (90, 201);  add swap neg_hi -> (%g, %g) want (110, 199)\n", o[4], o[5], o[6], o[7]);
   return 0;
}
