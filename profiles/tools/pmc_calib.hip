// Calibration of the rocprofv3 FETCH_SIZE / WRITE_SIZE counters on known byte counts, in the access
// widths the search kernels use (one dword per lane) and in the wide form (16 B per lane).
//   hipcc -O3 --offload-arch=gfx950 pmc_calib.hip -o pmc_calib.bin
//   rocprofv3 --pmc FETCH_SIZE -d out -- ./pmc_calib.bin
// Each kernel reads BYTES (512 MiB, beyond the 256 MiB Infinity Cache) once and writes 4 B per lane
// of its last wave only, so FETCH_SIZE x factor should equal BYTES.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

static const size_t BYTES = 512ull << 20;

__global__ void calib_read_b32(const unsigned* __restrict__ src, size_t n, unsigned* __restrict__ out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t step = (size_t)gridDim.x * blockDim.x;
    unsigned acc = 0;
    for (; i < n; i += step) acc ^= src[i];
    if (acc == 0x12345u) out[threadIdx.x] = acc;     // practically never: keeps the loads alive
}

__global__ void calib_read_b128(const uint4* __restrict__ src, size_t n, unsigned* __restrict__ out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t step = (size_t)gridDim.x * blockDim.x;
    unsigned acc = 0;
    for (; i < n; i += step) { uint4 v = src[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345u) out[threadIdx.x] = acc;
}

// 40-byte row segments at a 640-byte pitch: the shape of a search tile read (10 dwords per row).
__global__ void calib_read_rows(const unsigned* __restrict__ src, size_t nrows, unsigned* __restrict__ out) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t step = (size_t)gridDim.x * blockDim.x;
    unsigned acc = 0;
    for (size_t k = t; k < nrows * 10; k += step) {
        const size_t row = k / 10, c = k % 10;
        acc ^= src[row * 160 + c];
    }
    if (acc == 0x12345u) out[threadIdx.x] = acc;
}

__global__ void calib_write_b32(unsigned* __restrict__ dst, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t step = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += step) dst[i] = (unsigned)i;
}

int main() {
    unsigned *buf, *out;
    CHECK(hipMalloc(&buf, BYTES));
    CHECK(hipMalloc(&out, 4096));
    CHECK(hipMemset(buf, 1, BYTES));
    CHECK(hipDeviceSynchronize());
    const int grid = 256 * 16, wg = 256;
    hipLaunchKernelGGL(calib_read_b32, dim3(grid), dim3(wg), 0, 0, buf, BYTES / 4, out);
    hipLaunchKernelGGL(calib_read_b128, dim3(grid), dim3(wg), 0, 0, (const uint4*)buf, BYTES / 16, out);
    hipLaunchKernelGGL(calib_read_rows, dim3(grid), dim3(wg), 0, 0, buf, BYTES / 640, out);
    hipLaunchKernelGGL(calib_write_b32, dim3(grid), dim3(wg), 0, 0, buf, BYTES / 4);
    CHECK(hipDeviceSynchronize());
    printf("bytes per read kernel: %zu; rows kernel touches %zu rows x 40 B = %zu B (%zu B as 128-B lines)\n",
           BYTES, BYTES / 640, BYTES / 640 * 40, BYTES / 640 * 128);
    CHECK(hipFree(buf));
    CHECK(hipFree(out));
    return 0;
}
