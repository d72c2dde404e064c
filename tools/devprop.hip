// hipcc -o /tmp/devprop tools/devprop.hip && /tmp/devprop : what hipGetDeviceProperties reports (the engine scales its partition rules by clockRate and multiProcessorCount)
#include <hip/hip_runtime.h>
#include <cstdio>
int main() { hipDeviceProp_t p; if (hipGetDeviceProperties(&p, 0) != hipSuccess) return 1; printf("%s: %d CUs, clockRate %d kHz, memoryClockRate %d kHz, l2 %d B, gcnArch %s\n", p.name, p.multiProcessorCount, p.clockRate, p.memoryClockRate, p.l2CacheSize, p.gcnArchName); return 0; }
