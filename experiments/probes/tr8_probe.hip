// ds_read_b64_tr_b8 semantics probe (gfx950): LDS holds byte value = its own offset mod 251; every lane reads 8 bytes at a
// per-lane address; prints which LDS offsets each lane's 8 result bytes came from, for two address patterns.
// build + run on the GPU box: hipcc --offload-arch=gfx950 -O2 experiments/probes/tr8_probe.hip -o /tmp/tr8 && /tmp/tr8
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(2))) int i32x2;
#define LDS_PTR(T) __attribute__((address_space(3))) T*
__global__ void probe(const int* addr, unsigned short* out) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[4096];  // 16-bit value = element index (byte pairs)
    __shared__ __attribute__((aligned(16))) unsigned char ldb[8192];
    for (int i = threadIdx.x; i < 8192; i += 64) ldb[i] = (unsigned char)(i % 251);
    __syncthreads();
    const i32x2 v = __builtin_amdgcn_ds_read_tr8_b64_v2i32((LDS_PTR(i32x2))(ldb + addr[threadIdx.x]));
    const unsigned char* b = (const unsigned char*)&v;
    for (int e = 0; e < 8; ++e) out[threadIdx.x * 8 + e] = b[e];
}
int main() {
    int h_addr[64];
    unsigned short h_out[512];
    int* d_addr; unsigned short* d_out;
    hipMalloc(&d_addr, sizeof(h_addr)); hipMalloc(&d_out, sizeof(h_out));
    for (int pat = 0; pat < 2; ++pat) {
        // pattern 0: a [8 rows][16 bytes] block per 16-lane group, row stride 16 B: lane l -> row (l&15)>>1, half (l&1); groups 128 B apart
        // pattern 1: row stride 256 B (the GEMM tile): lane l -> row ((l&15)>>1) * 256 + (l&1) * 8, groups 2048 B apart
        for (int l = 0; l < 64; ++l) {
            const int i = l & 15, g = l >> 4;
            h_addr[l] = pat == 0 ? g * 128 + (i >> 1) * 16 + (i & 1) * 8 : g * 2048 + (i >> 1) * 256 + (i & 1) * 8;
        }
        hipMemcpy(d_addr, h_addr, sizeof(h_addr), hipMemcpyHostToDevice);
        probe<<<1, 64>>>(d_addr, d_out);
        hipMemcpy(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost);
        printf("pattern %d (values are LDS byte offsets mod 251)\n", pat);
        for (int l = 0; l < 64; ++l) {
            printf("lane %2d addr %5d:", l, h_addr[l]);
            for (int e = 0; e < 8; ++e) printf(" %3d", h_out[l * 8 + e]);
            printf("\n");
        }
    }
    return 0;
}
