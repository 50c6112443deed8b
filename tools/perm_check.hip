// v_perm_b32 / v_alignbyte_b32 operand order check (what enc4x4, rc4 and load_window rely on): prints perm(hi, lo, sel) for sel = 0x07030201
// build and run on the GPU box: hipcc --offload-arch=gfx950 -O2 tools/perm_check.hip -o /tmp/perm_check && /tmp/perm_check
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(uint32_t *o) {
    const uint32_t lo = 0x44434241u /* "ABCD" */, hi = 0x48474645u /* "EFGH" */;
    o[0] = __builtin_amdgcn_perm(hi, lo, 0x07030201u);
    o[1] = __builtin_amdgcn_alignbyte(0xAABBCCDDu, 0x11223344u, 1);
}
int main() {
    uint32_t *d, h[2];
    hipMalloc(&d, 8);
    k<<<1, 1>>>(d);
    hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
    printf("perm=%08x (want 48444342 if sel 0-3 pick the third-from-last operand's... lo bytes)  alignbyte=%08x\n", h[0], h[1]);
    return 0;
}
