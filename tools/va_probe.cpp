// where the address ranges of a HIP process lie on the box: brk heap, malloc / mmap, hipMalloc (small / large), hipHostMalloc (small / large), a hipHostRegister'ed heap buffer's device pointer
// build: hipcc -O2 --offload-arch=gfx950 tools/va_probe.cpp -o tools/va_probe
#include <hip/hip_runtime.h>
#include <unistd.h>
#include <cstdio>
#include <cstdlib>
int main()
{
   void *d0 = nullptr, *d1 = nullptr, *d2 = nullptr, *h0 = nullptr, *h1 = nullptr, *h2 = nullptr, *dp = nullptr;
   hipMalloc(&d0, 4); hipMalloc(&d1, 1 << 20); hipMalloc(&d2, (size_t)1 << 30);
   hipHostMalloc(&h0, 8, hipHostMallocMapped); hipHostMalloc(&h1, 1 << 20, hipHostMallocDefault); hipHostMalloc(&h2, (size_t)256 << 20, hipHostMallocDefault);
   void *m0 = malloc(64), *m1 = malloc(8 << 20);
   hipHostRegister(m0, 64, hipHostRegisterDefault);
   hipHostGetDevicePointer(&dp, m0, 0);
   hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
   hipEvent_t ev; hipEventCreate(&ev);
   printf("main %p  sbrk(0) %p  malloc(64) %p  malloc(8M) %p  stack %p\n", (void *)&main, sbrk(0), m0, m1, (void *)&d0);
   printf("hipMalloc 4 B %p  1 MB %p  1 GB %p\n", d0, d1, d2);
   printf("hipHostMalloc 8 B mapped %p  1 MB %p  256 MB %p\n", h0, h1, h2);
   printf("registered malloc(64): device pointer %p   stream handle %p  event handle %p\n", dp, (void *)st, (void *)ev);
   FILE *f = fopen("/proc/self/maps", "r");
   char line[512]; int n = 0;
   while (f && fgets(line, sizeof line, f) && n < 400) { ++n; if (line[0] == '5' || line[0] == '6' || n < 6) fputs(line, stdout); }
   printf("(%d map lines)\n", n);
   return 0;
}
