// Probe: semantics of ds_read_b64_tr_b16 (gfx950).  Each lane supplies the address of 4 contiguous 16-bit elements.
// Hypothesis: within a 16-lane group, lane i = 4r+q supplies row r, columns 4q..4q+3 of a 4x16 block M; lane i receives
// column i: (M[0][i], M[1][i], M[2][i], M[3][i]).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) short v4s;
__global__ void k(short* out, int pitch) {
    extern __shared__ __attribute__((aligned(16))) short lds[];
    for (int i = threadIdx.x; i < 64 * pitch; i += 64) lds[i] = (short)i;  // value = linear element index
    __syncthreads();
    int l = threadIdx.x, g = l >> 4, i = l & 15;
    const short* p = lds + (g * 4 + (i >> 2)) * pitch + (i & 3) * 4;  // group g reads rows 4g..4g+3, cols 0..15
    v4s r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s __attribute__((address_space(3)))*)p);
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = r[j];
}
int main() {
    short* d; short h[256];
    hipMalloc(&d, 512);
    int bad_total = 0;
    for (int pitch : {16, 144, 136}) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 64 * pitch * 2, 0, d, pitch);
        hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int l = 0; l < 64; ++l) for (int j = 0; j < 4; ++j) {
            int g = l >> 4, i = l & 15;
            int expect = (g * 4 + j) * pitch + i;  // M[j][i] of group g's block
            if (h[l * 4 + j] != (short)expect) { if (bad < 6) printf("pitch %d lane %d j %d got %d expect %d\n", pitch, l, j, h[l*4+j], expect); bad++; }
        }
        printf("pitch %d: %s (%d mismatches)\n", pitch, bad ? "MISMATCH" : "hypothesis confirmed", bad);
        bad_total += bad;
    }
    return bad_total != 0;
}
