// What the memory system of one MI355X gives a streaming kernel, by read : write mix and by number of concurrent streams.
//   hipcc --offload-arch=gfx950 -O2 -o tools/ubench_hbm tools/ubench_hbm.hip ; tools/ubench_hbm
// Every workgroup streams 16-byte words; R input arrays are read and summed, W output arrays written.  The transform
// passes of the path are 1 : 1, k_qsum3_fp reads 10 bytes for every byte it writes.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                      \
    do {                                                                           \
        hipError_t e_ = (x);                                                       \
        if (e_ != hipSuccess) {                                                    \
            std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));            \
            std::exit(1);                                                          \
        }                                                                          \
    } while (0)

typedef unsigned long long v2u64 __attribute__((ext_vector_type(2)));

template <int R, int W>
__global__ __launch_bounds__(256) void k_stream(const v2u64 *in, v2u64 *out, size_t words_per_array, size_t tile) {
    // grid-stride over tiles of `tile` words; array r of the inputs starts at in + r * words_per_array
    for (size_t t0 = (size_t)blockIdx.x * tile; t0 < words_per_array; t0 += (size_t)gridDim.x * tile) {
        for (size_t i = t0 + threadIdx.x; i < t0 + tile; i += 256) {
            v2u64 acc = {(unsigned long long)i, 0ull};
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const v2u64 v = __builtin_nontemporal_load(in + (size_t)r * words_per_array + i);
                acc.x += v.x;
                acc.y ^= v.y;
            }
#pragma unroll
            for (int w = 0; w < W; ++w) __builtin_nontemporal_store(acc, out + (size_t)w * words_per_array + i);
            if (W == 0 && acc.x == 0x123456789abcdefull && acc.y == 42) out[0] = acc;  // keeps the loads alive
        }
    }
}

template <int R, int W>
static void run(const v2u64 *in, v2u64 *out, size_t words_per_array, const char *what) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    double best = 0;
    for (int grid : {2048, 8192, 32768}) {
        for (int rep = 0; rep < 4; ++rep) {
            CK(hipEventRecord(a));
            k_stream<R, W><<<grid, 256>>>(in, out, words_per_array, 1024);
            CK(hipEventRecord(b));
            CK(hipEventSynchronize(b));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, a, b));
            const double tbs = (double)(R + W) * words_per_array * 16 / (ms * 1e-3) * 1e-12;
            if (tbs > best) best = tbs;
        }
    }
    std::printf("%-34s %6.2f TB/s\n", what, best);
}

int main() {
    const size_t words_per_array = (size_t)256 << 16;  // 256 MiB per array
    v2u64 *in = nullptr, *out = nullptr;
    CK(hipMalloc(&in, words_per_array * 16 * 10));
    CK(hipMalloc(&out, words_per_array * 16 * 4));
    CK(hipMemset(in, 1, words_per_array * 16 * 10));
    std::printf("# arrays of 256 MiB, 16-byte accesses, non-temporal; best of 3 grid sizes x 4 runs\n");
    run<1, 0>(in, out, words_per_array, "read only, 1 stream");
    run<4, 0>(in, out, words_per_array, "read only, 4 streams");
    run<10, 0>(in, out, words_per_array, "read only, 10 streams");
    run<0, 1>(in, out, words_per_array, "write only, 1 stream");
    run<0, 4>(in, out, words_per_array, "write only, 4 streams");
    run<1, 1>(in, out, words_per_array, "copy 1 : 1");
    run<2, 2>(in, out, words_per_array, "2 reads : 2 writes");
    run<4, 1>(in, out, words_per_array, "4 reads : 1 write");
    run<10, 1>(in, out, words_per_array, "10 reads : 1 write (k_qsum3_fp's mix)");
    return 0;
}
