// Probe of ds_read_b64_tr_b16 semantics on gfx950 (prints which LDS elements each lane receives).
//   hipcc --offload-arch=gfx950 -O2 tools/probes/probe_tr.hip -o /tmp/probe_tr && /tmp/probe_tr
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(short* out, const int* addr_elems) {
    __shared__ __attribute__((aligned(16))) short lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (short)i;
    __syncthreads();
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + addr_elems[threadIdx.x]));
    for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = v[j];
}
int main() {
    short* d_out; int* d_addr; short h_out[256]; int h_addr[64];
    hipMalloc(&d_out, sizeof(h_out)); hipMalloc(&d_addr, sizeof(h_addr));
    for (int variant = 0; variant < 3; ++variant) {
        for (int l = 0; l < 64; ++l) {
            if (variant == 0) h_addr[l] = l * 4;                                        // contiguous 8 bytes per lane
            else if (variant == 1) h_addr[l] = (l >> 4) * 1000 + (l & 15) * 4;            // each 16-lane group: own 64-element region
            else { const int g = l >> 4, i = l & 15; h_addr[l] = (4 * g + (i >> 2)) * 48 + (i & 3) * 4; }   // [key][48] rows: 4 rows x 16 cols per group
        }
        hipMemcpy(d_addr, h_addr, sizeof(h_addr), hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d_out, d_addr);
        hipMemcpy(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost);
        printf("variant %d\n", variant);
        for (int l = 0; l < 64; ++l) printf("lane %2d addr %5d -> %5d %5d %5d %5d\n", l, h_addr[l], h_out[l * 4], h_out[l * 4 + 1], h_out[l * 4 + 2], h_out[l * 4 + 3]);
    }
    return 0;
}
