// ubench_intmul.hip -- instruction-throughput probe for the integer path on gfx950.
// Build: hipcc --offload-arch=gfx950 -O3 -o ubench_intmul ubench_intmul.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef uint64_t u64;
#define ITER 4096

template <int OP>
__global__ void k(u64 *out, u64 seed) {
    u64 a0 = seed + threadIdx.x, a1 = a0 * 3 + 1, a2 = a0 * 5 + 7, a3 = a0 * 7 + 11;
    u64 b = seed * 0x9E3779B97F4A7C15ull + blockIdx.x;
    double d0 = (double)a0, d1 = (double)a1, d2 = (double)a2, d3 = (double)a3, db = 1.0000001;
    uint32_t x0 = (uint32_t)a0, x1 = (uint32_t)a1, x2 = (uint32_t)a2, x3 = (uint32_t)a3, xb = (uint32_t)b | 1;
    for (int i = 0; i < ITER; ++i) {
        if (OP == 0) {  // v_mad_u64_u32: 32x32 + 64
            a0 = (u64)(uint32_t)a0 * xb + a0; a1 = (u64)(uint32_t)a1 * xb + a1;
            a2 = (u64)(uint32_t)a2 * xb + a2; a3 = (u64)(uint32_t)a3 * xb + a3;
        } else if (OP == 1) {  // 64-bit mul lo
            a0 = a0 * b + 1; a1 = a1 * b + 1; a2 = a2 * b + 1; a3 = a3 * b + 1;
        } else if (OP == 2) {  // 64-bit mul hi
            a0 = __umul64hi(a0, b) + b; a1 = __umul64hi(a1, b) + b; a2 = __umul64hi(a2, b) + b; a3 = __umul64hi(a3, b) + b;
        } else if (OP == 3) {  // v_fma_f64
            d0 = fma(d0, db, 1.0); d1 = fma(d1, db, 1.0); d2 = fma(d2, db, 1.0); d3 = fma(d3, db, 1.0);
        } else if (OP == 4) {  // v_mul_lo_u32
            x0 = x0 * xb + 1; x1 = x1 * xb + 1; x2 = x2 * xb + 1; x3 = x3 * xb + 1;
        } else if (OP == 5) {  // v_mul_hi_u32
            x0 = __umulhi(x0, xb) + xb; x1 = __umulhi(x1, xb) + xb; x2 = __umulhi(x2, xb) + xb; x3 = __umulhi(x3, xb) + xb;
        } else if (OP == 6) {  // Shoup lazy modmul (the butterfly core)
            const u64 q = 1152921504606748673ull, w = b % q, wp = (u64)(((unsigned __int128)w << 64) / q);
            a0 = a0 * w - __umul64hi(a0, wp) * q; a1 = a1 * w - __umul64hi(a1, wp) * q;
            a2 = a2 * w - __umul64hi(a2, wp) * q; a3 = a3 * w - __umul64hi(a3, wp) * q;
        } else if (OP == 7) {  // v_add_co / 64-bit add (reference)
            a0 += b; a1 += b; a2 += b; a3 += b; a0 ^= a1; a2 ^= a3;
        } else if (OP == 9) {  // pseudo-Mersenne lazy modmul: q = 2^60 - c, companion wx = w * 2^31 mod q
            const u64 q = 1152921504606584833ull, w = b % q, wx = (u64)(((unsigned __int128)w << 31) % q);
            const uint32_t c = (uint32_t)((1ull << 60) - q);
            auto pm = [&](u64 a) {
                const uint32_t al = (uint32_t)a & 0x7fffffffu, ah = (uint32_t)(a >> 31);
                u64 y = (u64)al * (uint32_t)w + (u64)ah * (uint32_t)wx;
                u64 z = (u64)al * (uint32_t)(w >> 32) + (y >> 32);
                z = (u64)ah * (uint32_t)(wx >> 32) + z;
                const uint32_t hi = (uint32_t)(z >> 28);
                const u64 lo = ((z & 0x0fffffffull) << 32) | (uint32_t)y;
                return (u64)hi * c + lo;
            };
            a0 = pm(a0); a1 = pm(a1); a2 = pm(a2); a3 = pm(a3);
        } else if (OP == 8) {  // v_mul_u32_u24-style: 24-bit multiply
            x0 = __umul24(x0, xb) + 1; x1 = __umul24(x1, xb) + 1; x2 = __umul24(x2, xb) + 1; x3 = __umul24(x3, xb) + 1;
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + (u64)(d0 + d1 + d2 + d3) + x0 + x1 + x2 + x3;
}

template <int OP>
void run(const char *name, int ops_per_iter) {
    const int blocks = 256 * 8, threads = 256;  // 8 blocks/CU x 4 waves = 32 waves/CU = 8 waves/SIMD
    u64 *out; hipMalloc(&out, (size_t)blocks * threads * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<OP><<<blocks, threads>>>(out, 12345);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) k<OP><<<blocks, threads>>>(out, 12345 + r);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double waves = (double)blocks * threads / 64.0;
    double wave_ops = waves * ITER * ops_per_iter * 5;
    double per_simd_per_s = wave_ops / (ms * 1e-3) / (256.0 * 4);
    printf("%-28s %8.3f ms  %7.2f G wave-ops/s/SIMD  => %.2f cycles/wave-op @2.4GHz\n", name, ms / 5, per_simd_per_s / 1e9, 2.4e9 / per_simd_per_s);
    hipFree(out);
}

int main() {
    run<0>("v_mad_u64_u32", 4);
    run<1>("mul_lo_u64 (+add)", 4);
    run<2>("mul_hi_u64 (+add)", 4);
    run<3>("v_fma_f64", 4);
    run<4>("v_mul_lo_u32 (+add)", 4);
    run<5>("v_mul_hi_u32 (+add)", 4);
    run<6>("shoup_lazy modmul", 4);
    run<7>("add64+xor", 6);
    run<8>("v_mul_u32_u24 (+add)", 4);
    run<9>("pseudo-Mersenne modmul", 4);
    return 0;
}
