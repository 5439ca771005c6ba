// Probe of ds_read_b64_tr_b16 (gfx950): every halfword of a 4 KiB LDS image holds its own index; lane l reads at byte
// address 8*l (case 0) or at row-structured addresses (case 1: lane l -> row (l%16), 128-byte rows, column chunk l/16) and
// prints the four halfwords it received.   hipcc --offload-arch=gfx950 tr_read_probe.hip -o /tmp/tr_probe && /tmp/tr_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void probe(unsigned short* out, int mode) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[2048];
    for (int i = threadIdx.x; i < 2048; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    const int l = threadIdx.x;
    unsigned addr = mode == 0 ? 8 * l : ((l % 16) * 128 + (l / 16) * 8);
    addr += (unsigned)(size_t)lds;      // LDS base (flat->local low bits)
    unsigned long long v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = (unsigned short)(v >> (16 * j));
}
int main() {
    unsigned short* d; hipMalloc(&d, 64 * 4 * 2);
    unsigned short h[256];
    for (int mode = 0; mode < 2; ++mode) {
        probe<<<1, 64>>>(d, mode);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("mode %d (%s)\n", mode, mode == 0 ? "lane l reads bytes 8l..8l+7 => halfwords 4l..4l+3 untransposed" : "lane l: row l%16 (64 halfwords per row), halfwords 4*(l/16)..+3");
        for (int l = 0; l < 64; ++l) printf("  lane %2d: %4d %4d %4d %4d%s", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3], (l % 4 == 3) ? "\n" : "");
    }
    return 0;
}
