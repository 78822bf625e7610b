// ubench_ops.hip -- issue cost of the VALU opcodes the library's kernels are made of, on gfx950, in SHADER CYCLES.
//
// The kernels of the PRE path are straight-line integer / fp64 arithmetic at 3-4 waves per SIMD; what a kernel costs is
// the sum of its instructions' issue costs.  This probe measures them one opcode at a time with the opcode pinned by
// inline asm: every wave runs REP x 8 independent instances of one instruction between two s_memtime stamps (shader
// clock, so the numbers do not depend on what clock the chip holds), with W waves resident per SIMD (W = 1, 2, 4, 8:
// one 256-thread workgroup = one wave per SIMD; LDS sizing decides how many workgroups share a CU).  Reported:
// cycles per wave-instruction of SIMD time = elapsed / (W * instructions per wave), and for the two multiply-class
// opcodes the latency of a DEPENDENT chain (one wave per SIMD).
// Build: hipcc --offload-arch=gfx950 -O3 -o ubench_ops ubench_ops.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <vector>

typedef uint64_t u64;
constexpr int REP = 256;

#define OP8(stmt) stmt(0) stmt(1) stmt(2) stmt(3) stmt(4) stmt(5) stmt(6) stmt(7)

enum Op {
    MAD_U64_U32, LSHL_ADD_U64, LSHRREV_B64, LSHLREV_B64, MOV_B32, AND_B32, ADD_U32, ADD_CO_PAIR, ALIGNBIT, BITOP3,
    FMA_F64, MUL_F64, ADD_F64, RNDNE_F64, CVT_F64_U32, CNDMASK, MUL_LO_U32, MUL_HI_U32, ADD3_U32, LSHRREV_B32, XOR_B32,
    FMA_F32, CVT_U32_F64, MAD_DEP, FMA_F64_DEP, N_OPS
};
static const char *NAMES[N_OPS] = {
    "v_mad_u64_u32", "v_lshl_add_u64", "v_lshrrev_b64", "v_lshlrev_b64", "v_mov_b32", "v_and_b32", "v_add_u32",
    "v_add_co_u32 + v_addc_co_u32 (pair)", "v_alignbit_b32", "v_bitop3_b32 (v_xor3 class)", "v_fma_f64", "v_mul_f64",
    "v_add_f64", "v_rndne_f64", "v_cvt_f64_u32", "v_cndmask_b32", "v_mul_lo_u32", "v_mul_hi_u32", "v_add3_u32",
    "v_lshrrev_b32", "v_xor_b32", "v_fma_f32", "v_cvt_u32_f64", "v_mad_u64_u32 DEPENDENT chain", "v_fma_f64 DEPENDENT chain"};

template <int OP, int LDS_BYTES>
__global__ __launch_bounds__(256) void k_op(u64 *out, unsigned *cycles, u64 seed) {
    __shared__ char pad[LDS_BYTES];
    pad[threadIdx.x] = (char)threadIdx.x;
    u64 a[8];
    double d[8];
    uint32_t x[8];
    float f[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        a[i] = seed * (2 * i + 3) + threadIdx.x;
        d[i] = (double)(a[i] & 0xFFFFF) + 0.5;
        x[i] = (uint32_t)a[i] | 1u;
        f[i] = (float)x[i];
    }
    const uint32_t m32 = (uint32_t)seed | 3u;
    const double dc = 1.0000001;
    __syncthreads();
    const u64 t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int r = 0; r < REP; ++r) {
        if (OP == MAD_U64_U32) {
#define S(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a[i]) : "v"(x[i]), "v"(m32) : "vcc");
            OP8(S)
#undef S
        } else if (OP == LSHL_ADD_U64) {
#define S(i) asm volatile("v_lshl_add_u64 %0, %0, 1, %1" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
            OP8(S)
#undef S
        } else if (OP == LSHRREV_B64) {
#define S(i) asm volatile("v_lshrrev_b64 %0, 1, %0" : "+v"(a[i]));
            OP8(S)
#undef S
        } else if (OP == LSHLREV_B64) {
#define S(i) asm volatile("v_lshlrev_b64 %0, 1, %0" : "+v"(a[i]));
            OP8(S)
#undef S
        } else if (OP == MOV_B32) {
#define S(i) asm volatile("v_mov_b32 %0, %1" : "=v"(x[i]) : "v"(x[(i + 1) & 7]));
            OP8(S)
#undef S
        } else if (OP == AND_B32) {
#define S(i) asm volatile("v_and_b32 %0, %0, %1" : "+v"(x[i]) : "v"(m32));
            OP8(S)
#undef S
        } else if (OP == ADD_U32) {
#define S(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x[i]) : "v"(m32));
            OP8(S)
#undef S
        } else if (OP == ADD_CO_PAIR) {
#define S(i) asm volatile("v_add_co_u32 %0, vcc, %0, %2\n\tv_addc_co_u32 %1, vcc, %1, %2, vcc" : "+v"(x[i]), "+v"(x[(i + 4) & 7]) : "v"(m32) : "vcc");
            S(0) S(1) S(2) S(3)
#undef S
        } else if (OP == ALIGNBIT) {
#define S(i) asm volatile("v_alignbit_b32 %0, %0, %1, 7" : "+v"(x[i]) : "v"(m32));
            OP8(S)
#undef S
        } else if (OP == BITOP3) {
#define S(i) asm volatile("v_bitop3_b32 %0, %0, %1, %1 bitop3:0x96" : "+v"(x[i]) : "v"(m32));
            OP8(S)
#undef S
        } else if (OP == FMA_F64) {
#define S(i) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(d[i]) : "v"(dc));
            OP8(S)
#undef S
        } else if (OP == MUL_F64) {
#define S(i) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"(dc));
            OP8(S)
#undef S
        } else if (OP == ADD_F64) {
#define S(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(dc));
            OP8(S)
#undef S
        } else if (OP == RNDNE_F64) {
#define S(i) asm volatile("v_rndne_f64 %0, %0" : "+v"(d[i]));
            OP8(S)
#undef S
        } else if (OP == CVT_F64_U32) {
#define S(i) asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(d[i]) : "v"(x[i]));
            OP8(S)
#undef S
        } else if (OP == CVT_U32_F64) {
#define S(i) asm volatile("v_cvt_u32_f64 %0, %1" : "=v"(x[i]) : "v"(d[i]));
            OP8(S)
#undef S
        } else if (OP == CNDMASK) {
#define S(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[i]) : "v"(m32) : "vcc");
            OP8(S)
#undef S
        } else if (OP == MUL_LO_U32) {
#define S(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x[i]) : "v"(m32));
            OP8(S)
#undef S
        } else if (OP == MUL_HI_U32) {
#define S(i) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(x[i]) : "v"(m32));
            OP8(S)
#undef S
        } else if (OP == ADD3_U32) {
#define S(i) asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(x[i]) : "v"(m32));
            OP8(S)
#undef S
        } else if (OP == LSHRREV_B32) {
#define S(i) asm volatile("v_lshrrev_b32 %0, 1, %0" : "+v"(x[i]));
            OP8(S)
#undef S
        } else if (OP == XOR_B32) {
#define S(i) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(x[i]) : "v"(m32));
            OP8(S)
#undef S
        } else if (OP == FMA_F32) {
#define S(i) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(f[i]) : "v"(f[(i + 1) & 7]));
            OP8(S)
#undef S
        } else if (OP == MAD_DEP) {  // 8 instructions, each needs the previous one's result
#define S(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a[0]) : "v"((uint32_t)a[0]), "v"(m32) : "vcc");
            OP8(S)
#undef S
        } else if (OP == FMA_F64_DEP) {
#define S(i) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(d[0]) : "v"(dc));
            OP8(S)
#undef S
        }
    }
    const u64 t1 = __builtin_amdgcn_s_memtime();
    u64 acc = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) acc += a[i] + (u64)d[i] + x[i] + (u64)f[i];
    out[blockIdx.x * 256 + threadIdx.x] = acc + pad[(threadIdx.x + 1) & 255];
    if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * 4 + threadIdx.x / 64] = (unsigned)(t1 - t0);
}

template <int OP, int LDS_BYTES>
static double run(int waves_per_simd, u64 *d_out, unsigned *d_cyc) {
    const int blocks = 256 * waves_per_simd;  // one generation: exactly W workgroups per CU
    for (int rep = 0; rep < 2; ++rep) k_op<OP, LDS_BYTES><<<blocks, 256>>>(d_out, d_cyc, 12345 + rep);
    (void)hipDeviceSynchronize();
    std::vector<unsigned> c(blocks * 4);
    (void)hipMemcpy(c.data(), d_cyc, c.size() * 4, hipMemcpyDeviceToHost);
    std::sort(c.begin(), c.end());
    const double med = c[c.size() / 2];
    const int per_wave = REP * ((OP == ADD_CO_PAIR) ? 4 : 8);
    return med / ((double)waves_per_simd * per_wave);
}

template <int OP>
static void bench(u64 *d_out, unsigned *d_cyc) {
    // LDS per workgroup picks the residency: 160 KiB / W
    const double w1 = run<OP, 65536>(1, d_out, d_cyc);   // 64 KiB: at most 2 fit, 1 launched per CU
    const double w2 = run<OP, 65536>(2, d_out, d_cyc);
    const double w4 = run<OP, 36864>(4, d_out, d_cyc);   // 36 KiB: 4 per CU
    const double w8 = run<OP, 16384>(8, d_out, d_cyc);   // 16 KiB: 8 per CU
    printf("%-40s %8.2f %8.2f %8.2f %8.2f\n", NAMES[OP], w1, w2, w4, w8);
}

int main() {
    u64 *d_out;
    unsigned *d_cyc;
    (void)hipMalloc(&d_out, (size_t)2048 * 256 * 8);
    (void)hipMalloc(&d_cyc, (size_t)2048 * 4 * 4);
    printf("# shader cycles of SIMD time per wave64 instruction (s_memtime; median over waves), by waves resident per SIMD\n");
    printf("# (the pair row counts the two instructions as one; DEPENDENT rows: every instruction waits for the previous one)\n");
    printf("%-40s %8s %8s %8s %8s\n", "opcode", "W=1", "W=2", "W=4", "W=8");
    bench<MAD_U64_U32>(d_out, d_cyc);
    bench<LSHL_ADD_U64>(d_out, d_cyc);
    bench<LSHRREV_B64>(d_out, d_cyc);
    bench<LSHLREV_B64>(d_out, d_cyc);
    bench<MOV_B32>(d_out, d_cyc);
    bench<AND_B32>(d_out, d_cyc);
    bench<ADD_U32>(d_out, d_cyc);
    bench<ADD_CO_PAIR>(d_out, d_cyc);
    bench<ALIGNBIT>(d_out, d_cyc);
    bench<BITOP3>(d_out, d_cyc);
    bench<ADD3_U32>(d_out, d_cyc);
    bench<LSHRREV_B32>(d_out, d_cyc);
    bench<XOR_B32>(d_out, d_cyc);
    bench<CNDMASK>(d_out, d_cyc);
    bench<MUL_LO_U32>(d_out, d_cyc);
    bench<MUL_HI_U32>(d_out, d_cyc);
    bench<FMA_F32>(d_out, d_cyc);
    bench<FMA_F64>(d_out, d_cyc);
    bench<MUL_F64>(d_out, d_cyc);
    bench<ADD_F64>(d_out, d_cyc);
    bench<RNDNE_F64>(d_out, d_cyc);
    bench<CVT_F64_U32>(d_out, d_cyc);
    bench<CVT_U32_F64>(d_out, d_cyc);
    bench<MAD_DEP>(d_out, d_cyc);
    bench<FMA_F64_DEP>(d_out, d_cyc);
    return 0;
}
