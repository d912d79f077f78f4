#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned *out) {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    if (threadIdx.x == 0) out[blockIdx.x] = v;
}
int main() {
    unsigned *d; hipMalloc(&d, 4 * 64);
    k<<<64, 64>>>(d);
    unsigned h[64]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    for (int i = 0; i < 64; i++) printf("%d:%x ", i, h[i]);
    printf("\n");
}
