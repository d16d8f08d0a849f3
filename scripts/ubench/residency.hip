// Which workgroups share a CU?  512 workgroups of 256 threads with 48 KiB of LDS and long enough to be co-resident: each records
// HW_REG_XCC_ID / HW_REG_HW_ID of its wave 0 and its start time.   hipcc --offload-arch=gfx950 -O2 residency.hip -o residency && ./residency
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void __launch_bounds__(256, 2) k(unsigned* out, int spin) {
  __shared__ float pad[12288];
  unsigned hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  unsigned long long t0 = __builtin_readcyclecounter();
  float acc = threadIdx.x;
  for (int i = 0; i < spin; ++i) { acc = acc * 1.0001f + 0.5f; pad[(threadIdx.x + i) % 12288] = acc; }
  if (threadIdx.x % 64 == 0) {
    unsigned* o = out + (blockIdx.x * 4 + threadIdx.x / 64) * 4;
    o[0] = hw; o[1] = xcc; o[2] = (unsigned)(t0 >> 6); o[3] = (unsigned)pad[threadIdx.x];
  }
}
int main() {
  const int G = 1024;
  unsigned* d; hipMalloc(&d, G * 16 * 4);
  k<<<G, 256>>>(d, 20000);
  hipDeviceSynchronize();
  std::vector<unsigned> h(G * 16);
  hipMemcpy(h.data(), d, G * 64, hipMemcpyDeviceToHost);
  for (int b = 0; b < G; ++b) {
    printf("%d", b);
    for (int w = 0; w < 4; ++w) { unsigned hw = h[(b * 4 + w) * 4], x = h[(b * 4 + w) * 4 + 1]; printf("  x%u hw%08x t%u", x & 0xf, hw, h[(b*4+w)*4+2]); }
    printf("\n");
  }
}
