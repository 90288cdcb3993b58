// Micro-benchmark: issue rate of v_dot4_u32_u8 / v_perm_b32 / v_bfi_b32 on gfx950 (diagnostic tool).
// hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>
__global__ void k(unsigned* out, int iters) {
    unsigned a[8], b = threadIdx.x * 2654435761u + 1, c = blockIdx.x + 12345;
#pragma unroll
    for (int i = 0; i < 8; i++) a[i] = threadIdx.x + i;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 8; r++) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if (MODE == 0) a[i] = __builtin_amdgcn_udot4(b, c, a[i], false);
                if (MODE == 1) a[i] = __builtin_amdgcn_perm(a[i], b, c);
                if (MODE == 2) a[i] = (a[i] & b) | (c & ~b);
                if (MODE == 3) a[i] = a[i] * 3u + b;   // v_mad_u32_u24 / v_mul_lo
                if (MODE == 4) a[i] = __builtin_amdgcn_udot4(a[i], a[i], c, false);
                if (MODE == 5) { typedef unsigned short u2 __attribute__((ext_vector_type(2)));      // v_pk_max_u16
                                 a[i] = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(u2, a[i]), __builtin_bit_cast(u2, b))); }
                if (MODE == 6) a[i] = max(a[i], (b >> 8) & 0xffu);     // v_max_u32_sdwa (byte select of one operand)
                if (MODE == 7) a[i] = max(a[i], b);                     // v_max_u32
                if (MODE == 8) a[i] = max(max(a[i], b), c);             // v_max3_u32
            }
        }
        b += 7; c ^= b;
    }
    unsigned s = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
void run(const char* name, unsigned* d, int waves_per_simd) {
    const int iters = 4000;
    const int blocks = 256 * waves_per_simd;   // 256-thread blocks: 4 waves = 1 per SIMD per block
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<blocks, 256>>>(d, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE><<<blocks, 256>>>(d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double insts = (double)blocks * 4 * iters * 64;   // wave-instructions
    const double per_simd_per_s = insts / (256.0 * 4) / (ms * 1e-3);
    printf("%-28s waves/SIMD=%d  %.3f ms  %.2f Gwave-inst/s/SIMD (=> %.2f cycles/inst at 2.4 GHz)\n", name, waves_per_simd,
           ms, per_simd_per_s / 1e9, 2.4e9 / per_simd_per_s);
}

int main() {
    unsigned* d;
    hipMalloc(&d, 256 * 8 * 256 * 4);
    for (int w : {1, 2, 4, 8}) {
        run<0>("v_dot4_u32_u8 (acc chain)", d, w);
        run<4>("v_dot4_u32_u8 (w,w,c)", d, w);
        run<1>("v_perm_b32", d, w);
        run<2>("v_bfi_b32", d, w);
        run<3>("v_mad (u32 mul-add)", d, w);
        run<5>("v_pk_max_u16", d, w);
        run<6>("v_max_u32_sdwa", d, w);
        run<7>("v_max_u32", d, w);
        run<8>("v_max3_u32", d, w);
    }
    return 0;
}
