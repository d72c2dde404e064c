// hipcc --offload-arch=gfx950 -O2 -o tools/anyorder_probe tools/anyorder_probe.hip
// Does hipExtLaunchKernelGGL(..., hipExtAnyOrderLaunch) let a kernel start beside the kernel in front of it on the SAME stream (AQL packet without the barrier
// bit) on gfx950 / ROCm 7.2?  hip_ext.h says the flag "is not supported on AMD GFX9xx boards".  Kernel A (1 workgroup) spins 1 ms and stamps start / end; kernel B
// stamps its start.  Overlap <=> B.start < A.end.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void k_spin(long long *t, long long ticks) { if (threadIdx.x) return; t[0] = wall_clock64(); while (wall_clock64() - t[0] < ticks) { __builtin_amdgcn_s_sleep(8); } t[1] = wall_clock64(); }
__global__ void k_stamp(long long *t) { if (threadIdx.x == 0 && blockIdx.x == 0) t[2] = wall_clock64(); }
int main()
{
   long long *d, h[3];
   CK(hipMalloc(&d, 3 * sizeof(long long)));
   hipStream_t st;
   CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
   hipEvent_t ev;
   CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
   for (int mode = 0; mode < 4; ++mode) {
      CK(hipMemset(d, 0, 3 * sizeof(long long)));
      // mode 0: plain launches; 1: B any-order; 2: A with a stop event, B any-order; 3: A, an event record, B any-order
      if (mode == 2) hipExtLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, st, nullptr, ev, 0, d, 100000ll);
      else hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, st, d, 100000ll);
      if (mode == 3) CK(hipEventRecord(ev, st));
      if (mode == 0) hipLaunchKernelGGL(k_stamp, dim3(1), dim3(64), 0, st, d);
      else hipExtLaunchKernelGGL(k_stamp, dim3(1), dim3(64), 0, st, nullptr, nullptr, hipExtAnyOrderLaunch, d);
      CK(hipStreamSynchronize(st));
      CK(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost));
      printf("mode %d: A ran %.1f us; B started %.1f us after A's start, %+.1f us relative to A's end -> %s\n", mode, (h[1] - h[0]) / 100.0, (h[2] - h[0]) / 100.0, (h[2] - h[1]) / 100.0,
             h[2] < h[1] ? "OVERLAP" : "in order");
   }
   return 0;
}
