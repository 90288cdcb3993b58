// Micro-benchmark: issue rate of individual gfx950 VALU instructions, each as 8 independent dependent-chains per lane
// written in inline assembly so that the compiler cannot fold them (diagnostic tool).
// hipcc --offload-arch=gfx950 -O3 valu_rate2.hip -o valu_rate2.bin && ./valu_rate2.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

#define DEF_KERNEL(NAME, ASM)                                                                  \
    __global__ void NAME(unsigned* out, int iters, unsigned long long* clk) {                  \
        unsigned a[8], b = threadIdx.x * 2654435761u + 1, c = blockIdx.x + 12345;              \
        for (int i = 0; i < 8; i++) a[i] = threadIdx.x + i;                                    \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();                            \
        for (int it = 0; it < iters; it++) {                                                   \
            _Pragma("unroll") for (int r = 0; r < 8; r++) {                                    \
                _Pragma("unroll") for (int i = 0; i < 8; i++)                                  \
                    asm volatile(ASM : "+v"(a[i]) : "v"(b), "v"(c));                           \
            }                                                                                  \
        }                                                                                      \
        unsigned s = 0;                                                                        \
        for (int i = 0; i < 8; i++) s += a[i];                                                 \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();                            \
        if ((threadIdx.x & 63) == 0) clk[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;       \
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                        \
    }

DEF_KERNEL(k_add, "v_add_u32 %0, %0, %1")
DEF_KERNEL(k_max, "v_max_u32 %0, %0, %1")
DEF_KERNEL(k_max3, "v_max3_u32 %0, %0, %1, %2")
DEF_KERNEL(k_max_sdwa, "v_max_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1")
DEF_KERNEL(k_pk_max, "v_pk_max_u16 %0, %0, %1")
DEF_KERNEL(k_pk_add, "v_pk_add_u16 %0, %0, %1")
DEF_KERNEL(k_pk_sub_clamp, "v_pk_sub_u16 %0, %0, %1 clamp")
DEF_KERNEL(k_perm, "v_perm_b32 %0, %0, %1, %2")
DEF_KERNEL(k_alignbyte, "v_alignbyte_b32 %0, %0, %1, 1")
DEF_KERNEL(k_alignbit, "v_alignbit_b32 %0, %0, %1, 8")
DEF_KERNEL(k_bfe, "v_bfe_u32 %0, %0, 8, 8")
DEF_KERNEL(k_and, "v_and_b32 %0, %0, %1")
DEF_KERNEL(k_and_or, "v_and_or_b32 %0, %0, %1, %2")
DEF_KERNEL(k_lshl_or, "v_lshl_or_b32 %0, %0, 8, %1")
DEF_KERNEL(k_bfi, "v_bfi_b32 %0, %1, %0, %2")
DEF_KERNEL(k_dot4, "v_dot4_u32_u8 %0, %1, %2, %0")
DEF_KERNEL(k_dot2_u16, "v_dot2_u32_u16 %0, %1, %2, %0")
DEF_KERNEL(k_sad_u8, "v_sad_u8 %0, %1, %2, %0")
DEF_KERNEL(k_mad_u24, "v_mad_u32_u24 %0, %0, %1, %2")
DEF_KERNEL(k_mul_lo, "v_mul_lo_u32 %0, %0, %1")
DEF_KERNEL(k_cndmask, "v_cndmask_b32 %0, %0, %1, vcc")
DEF_KERNEL(k_lshlrev, "v_lshlrev_b32 %0, 3, %0")
DEF_KERNEL(k_mbcnt, "v_mbcnt_lo_u32_b32 %0, %1, %0")

template <typename K>
void run(const char* name, K kern, unsigned* d, unsigned long long* clk, int waves_per_simd) {
    const int iters = 4000;
    const int blocks = 256 * waves_per_simd;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    kern<<<blocks, 256>>>(d, iters, clk);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    kern<<<blocks, 256>>>(d, iters, clk);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double insts = (double)blocks * 4 * iters * 64;
    const double per_simd_per_s = insts / (256.0 * 4) / (ms * 1e-3);
    // in-kernel shader clocks (s_memtime) per wave: with w co-resident waves per SIMD a wave's stream of
    // iters*64 instructions takes w times the SIMD's cycles per instruction
    std::vector<unsigned long long> h(blocks * 4);
    (void)hipMemcpy(h.data(), clk, h.size() * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    const double med = (double)h[h.size() / 2];
    const double cyc = med / ((double)iters * 64) / waves_per_simd;
    printf("%-22s waves/SIMD=%d  %.3f ms  %.2f cycles/inst per SIMD by s_memtime (clock %.2f GHz; %.2f if the clock were 2.4 GHz)\n",
           name, waves_per_simd, ms, cyc, med / (ms * 1e-3) / 1e9, 2.4e9 / per_simd_per_s);
}

int main() {
    unsigned* d;
    (void)hipMalloc(&d, 256 * 8 * 256 * 4);
    unsigned long long* clk;
    (void)hipMalloc(&clk, 256 * 8 * 4 * 8);
    for (int w : {1, 2, 4, 8}) {
#define RUN(K) run(#K, K, d, clk, w)
        RUN(k_add); RUN(k_max); RUN(k_max3); RUN(k_max_sdwa); RUN(k_pk_max); RUN(k_pk_add); RUN(k_pk_sub_clamp); RUN(k_perm);
        RUN(k_alignbyte); RUN(k_alignbit); RUN(k_bfe); RUN(k_and); RUN(k_and_or); RUN(k_lshl_or); RUN(k_bfi); RUN(k_dot4);
        RUN(k_dot2_u16); RUN(k_sad_u8); RUN(k_mad_u24); RUN(k_mul_lo); RUN(k_cndmask); RUN(k_lshlrev); RUN(k_mbcnt);
    }
    return 0;
}
