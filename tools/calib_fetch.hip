// calib_fetch.hip -- what rocprofv3's FETCH_SIZE / WRITE_SIZE report for this project's access shapes (gfx950).
// MI355X_MICROARCH.md: FETCH_SIZE shows exactly 1/2 of the bytes of a wide (16 B per lane) coalesced read and "other access
// widths are uncalibrated: calibrate on a known byte count in your own access pattern".  Every kernel below moves a KNOWN
// number of bytes (printed); run under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` and divide.
//   k_rd16   16 B per lane, unit stride                     (pair-layout tiles: c1, c0, eval-key, accumulators)
//   k_rd8    8 B per lane, 512 B contiguous per wave        (row-kernel inputs: thread (g, j) reads words j + 16 k)
//   k_rd8col 8 B per lane, 4 row segments of 128 B, 2 KiB apart per wave instruction   (column kernels)
//   k_wr16 / k_wr8col   the matching stores
// build: hipcc --offload-arch=gfx950 -O3 -o tools/calib_fetch tools/calib_fetch.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CHECK(e)                                                                 \
    do {                                                                         \
        hipError_t r = (e);                                                      \
        if (r != hipSuccess) {                                                   \
            std::fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(r));          \
            std::exit(1);                                                        \
        }                                                                        \
    } while (0)

typedef unsigned long long u64;

__global__ void k_rd16(const ulong2 *in, u64 *sink, size_t n16) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    u64 acc = 0;
    for (; i < n16; i += (size_t)gridDim.x * blockDim.x) {
        const ulong2 v = in[i];
        acc += v.x ^ v.y;
    }
    if (acc == 0x123456789ull) sink[0] = acc;
}
__global__ void k_rd8(const u64 *in, u64 *sink, size_t n8) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    u64 acc = 0;
    for (; i < n8; i += (size_t)gridDim.x * blockDim.x) acc += in[i];
    if (acc == 0x123456789ull) sink[0] = acc;
}
// limb of 256 x 256 words; workgroup = tile of 16 columns; thread (j, c) reads rows j + 16 k of column c
__global__ void k_rd8col(const u64 *in, u64 *sink, size_t limbs) {
    const int c = threadIdx.x % 16, j = threadIdx.x / 16;
    u64 acc = 0;
    for (size_t t = blockIdx.x; t < limbs * 16; t += gridDim.x) {
        const u64 *p = in + (t / 16) * 65536 + (t % 16) * 16 + c;
#pragma unroll
        for (int k = 0; k < 16; ++k) acc += p[(size_t)(j + 16 * k) * 256];
    }
    if (acc == 0x123456789ull) sink[0] = acc;
}
__global__ void k_wr16(ulong2 *out, size_t n16) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i < n16; i += (size_t)gridDim.x * blockDim.x) out[i] = ulong2{i, i + 1};
}
__global__ void k_wr8col(u64 *out, size_t limbs) {
    const int c = threadIdx.x % 16, j = threadIdx.x / 16;
    for (size_t t = blockIdx.x; t < limbs * 16; t += gridDim.x) {
        u64 *p = out + (t / 16) * 65536 + (t % 16) * 16 + c;
#pragma unroll
        for (int k = 0; k < 16; ++k) p[(size_t)(16 * j + k) * 256] = t + k;
    }
}

int main() {
    const size_t limbs = 4096, bytes = limbs * 65536 * 8;  // 2 GiB: far beyond the 256 MiB Infinity Cache
    u64 *buf = nullptr, *sink = nullptr;
    CHECK(hipMalloc(&buf, bytes));
    CHECK(hipMalloc(&sink, 64));
    CHECK(hipMemset(buf, 1, bytes));
    CHECK(hipDeviceSynchronize());
    k_rd16<<<4096, 256>>>(reinterpret_cast<const ulong2 *>(buf), sink, bytes / 16);
    k_rd8<<<4096, 256>>>(buf, sink, bytes / 8);
    k_rd8col<<<4096, 256>>>(buf, sink, limbs);
    k_wr16<<<4096, 256>>>(reinterpret_cast<ulong2 *>(buf), bytes / 16);
    k_wr8col<<<4096, 256>>>(buf, limbs);
    CHECK(hipDeviceSynchronize());
    std::printf("bytes_per_kernel=%zu\n", bytes);
    return 0;
}
