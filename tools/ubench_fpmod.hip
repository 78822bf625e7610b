// ubench_fpmod.hip -- exactness + throughput probe: FMA-based modular multiplication for q < 2^51 vs integer Shoup.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
#include "../ppqsflhe_amd/csrc/modarith.hpp"  // pm_lazy / pm_fold: the library's own pseudo-Mersenne arithmetic
typedef uint64_t u64;
typedef unsigned __int128 u128;

// v = y*w - b*q exactly, |v| <= 1.5 q for |y| <= 2^52  (y, w integers held in doubles; wq = w/q rounded)
__device__ __forceinline__ double fp_mulmod(double y, double w, double wq, double q) {
    double h = y * w;
    double l = fma(y, w, -h);
    double b = rint(y * wq);
    double c = fma(-b, q, h);
    return c + l;
}
__device__ __forceinline__ double fp_reduce(double x, double q, double qinv) {  // |result| <= q/2 (+1)
    return fma(-rint(x * qinv), q, x);
}

// exactness: random |y| <= 4q, w < q; compare (fp result mod q) with integer y*w mod q
__global__ void k_check(const u64 *seeds, u64 q, unsigned long long *bad, double *maxabs, int iters) {
    u64 s = seeds[blockIdx.x * blockDim.x + threadIdx.x];
    const double qd = (double)q;
    unsigned long long nbad = 0;
    double mx = 0;
    for (int i = 0; i < iters; ++i) {
        s = s * 6364136223846793005ull + 1442695040888963407ull;
        u64 a = s; s = s * 6364136223846793005ull + 1442695040888963407ull;
        u64 b = s;
        u64 w = (u64)(((u128)(a >> 1) * q) >> 63);          // < q
        u64 ymag = (u64)(((u128)(b >> 1) * (4 * q)) >> 63);  // < 4q
        if (i % 7 == 0) ymag = 4 * q - 1 - (i % 5);         // edges
        if (i % 11 == 0) w = q - 1 - (i % 3);
        const bool neg = (b & 1);
        double y = neg ? -(double)ymag : (double)ymag;
        double wd = (double)w, wq = wd / qd;
        double v = fp_mulmod(y, wd, wq, qd);
        mx = fmax(mx, fabs(v));
        // integer check
        u64 ref = (u64)((u128)(ymag % q) * w % q);
        if (neg && ref) ref = q - ref;
        double vr = fp_reduce(v, qd, 1.0 / qd);
        long long vi = (long long)vr;
        if (vi < 0) vi += (long long)q;
        if ((u64)vi != ref || v != rint(v)) ++nbad;
    }
    atomicAdd(bad, nbad);
    if (mx > 0) { unsigned long long *p = (unsigned long long *)maxabs; atomicMax(p, (unsigned long long)__double_as_longlong(mx)); }
}

template <int OP>
__global__ void k_rate(double *out, double seed) {
    const double q = 1125899908022273.0, qinv = 1.0 / q;
    double x0 = seed + threadIdx.x, x1 = x0 * 3 + 1, x2 = x0 * 5 + 7, x3 = x0 * 7 + 11;
    double y0 = x0 + 17, y1 = x1 + 19, y2 = x2 + 23, y3 = x3 + 29;
    const double w = 998877665544331.0, wq = w / q;
    u64 a0 = (u64)x0, a1 = (u64)x1, a2 = (u64)x2, a3 = (u64)x3, b0 = (u64)y0, b1 = (u64)y1, b2 = (u64)y2, b3 = (u64)y3;
    const u64 qi = 1125899908022273ull, q2 = 2 * qi, q4 = 4 * qi, wi = 998877665544331ull;
    const u64 wp = (u64)(((u128)wi << 64) / qi);
    for (int i = 0; i < 2048; ++i) {
        if (OP == 0) {  // fp butterfly incl. reducing both outputs every 2nd iteration
#define FPB(x, y) { double v = fp_mulmod(y, w, wq, q); double u = x; x = u + v; y = u - v; if (i & 1) { x = fp_reduce(x, q, qinv); y = fp_reduce(y, q, qinv); } }
            FPB(x0, y0) FPB(x1, y1) FPB(x2, y2) FPB(x3, y3)
        } else if (OP == 1) {  // integer butterfly (c4 correction)
#define INB(x, y) { u64 t = x + (0 - q4); u64 u = (long long)t < 0 ? x : t; u64 h = __umul64hi(y, wp); u64 v = y * wi - h * qi; x = u + v; y = u - v + q2; }
            INB(a0, b0) INB(a1, b1) INB(a2, b2) INB(a3, b3)
        } else if (OP == 3) {  // pseudo-Mersenne butterfly on a 60-bit prime, x folded every 2nd iteration (radix_forward_pm)
            mk::LimbConst lc{};
            lc.q = 1152921504606584833ull; lc.k = 60; lc.pm_c = (uint32_t)((1ull << 60) - lc.q);
            const mk::PmK P = mk::pm_consts(lc);
            const u64 wt = wi << 3, wxt = (u64)(((u128)wi << 32) % lc.q) << 3;
#define PMB(x, y) { u64 u = (i & 1) ? x : mk::pm_fold(x, P); u64 v = mk::pm_lazy(y, wt, wxt, P); x = u + v; y = u - v + P.q3; }
            PMB(a0, b0) PMB(a1, b1) PMB(a2, b2) PMB(a3, b3)
        } else if (OP == 2) {  // rint rate
            x0 = rint(x0 * 1.0000001) + 0.25; x1 = rint(x1 * 1.0000001) + 0.25; x2 = rint(x2 * 1.0000001) + 0.25; x3 = rint(x3 * 1.0000001) + 0.25;
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + y0 + y1 + y2 + y3 + (double)(a0 + a1 + a2 + a3 + b0 + b1 + b2 + b3);
}

template <int OP>
void rate(const char *name, int per_iter) {
    const int blocks = 2048, threads = 256;
    double *out; (void)hipMalloc(&out, (size_t)blocks * threads * 8);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k_rate<OP><<<blocks, threads>>>(out, 3.0); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) k_rate<OP><<<blocks, threads>>>(out, 3.0 + r);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    double waves = (double)blocks * threads / 64, ops = waves * 2048 * per_iter * 5;
    double rate = ops / (ms * 1e-3) / 1024.0;
    printf("%-28s %8.3f ms  => %.1f cycles per wave-%s @2.4GHz\n", name, ms / 5, 2.4e9 / rate, "op");
    (void)hipFree(out);
}

int main() {
    const u64 qs[] = {1125899908022273ull, 1125899904679937ull, 1099511922689ull, 557057ull, (1ull << 51) - 129};
    for (u64 q : qs) {
        const int blocks = 256, threads = 256, iters = 20000;
        u64 *seeds; unsigned long long *bad; double *mx;
        (void)hipMalloc(&seeds, blocks * threads * 8); (void)hipMalloc(&bad, 8); (void)hipMalloc(&mx, 8);
        u64 *h = new u64[blocks * threads];
        for (int i = 0; i < blocks * threads; ++i) h[i] = 0x9E3779B97F4A7C15ull * (i + 1) + q;
        (void)hipMemcpy(seeds, h, blocks * threads * 8, hipMemcpyHostToDevice);
        (void)hipMemset(bad, 0, 8); (void)hipMemset(mx, 0, 8);
        k_check<<<blocks, threads>>>(seeds, q, bad, mx, iters);
        unsigned long long nb; double m;
        (void)hipMemcpy(&nb, bad, 8, hipMemcpyDeviceToHost); (void)hipMemcpy(&m, mx, 8, hipMemcpyDeviceToHost);
        printf("q=%llu (%d bits): %llu mismatches in %.1e trials, max|v|/q = %.3f\n", (unsigned long long)q,
               64 - __builtin_clzll(q), nb, (double)blocks * threads * iters, m / (double)q);
        delete[] h;
    }
    rate<0>("fp64 butterfly (+reduce/2)", 4);
    rate<1>("int Shoup butterfly", 4);
    rate<2>("rint+mul+add", 4);
    rate<3>("int pseudo-Mersenne butterfly", 4);
    return 0;
}
