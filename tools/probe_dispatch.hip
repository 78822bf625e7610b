// probe_dispatch.hip -- which workgroups of a 1-D grid share a CU, and when they start (gfx950).
// The library's kernels are 256-thread workgroups, 3-4 resident per CU; all workgroups of a kernel take the same time, so
// the ones that start together stay in step (load phase together, compute phase together).  This probe launches such a
// grid (34 KiB of LDS per workgroup -> 4 per CU) of busy-waiting workgroups and records, per workgroup: XCC id, hardware
// id (SE / CU / SIMD of wave 0), start and end time (s_memrealtime, 100 MHz).  Output: for each generation slot how the
// block index maps to (xcc, se, cu), and the start-time spread inside a CU.
// Build: hipcc --offload-arch=gfx950 -O3 -o probe_dispatch probe_dispatch.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <map>
#include <vector>

struct Rec {
    uint32_t xcc, hwid;
    uint64_t t0, t1;
};

__global__ __launch_bounds__(256) void k_probe(Rec *out, uint32_t spin_ticks) {
    __shared__ uint64_t pad[34816 / 8];
    pad[threadIdx.x] = threadIdx.x;
    const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
    uint32_t xcc, hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    while (__builtin_amdgcn_s_memrealtime() - t0 < spin_ticks) __builtin_amdgcn_s_sleep(8);
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = Rec{xcc & 0xf, hwid, t0, __builtin_amdgcn_s_memrealtime() + pad[1]};
}

int main() {
    const int blocks = 4096;
    Rec *d;
    hipMalloc(&d, blocks * sizeof(Rec));
    for (int rep = 0; rep < 2; ++rep) k_probe<<<blocks, 256>>>(d, 2000);  // 20 us per workgroup
    hipDeviceSynchronize();
    std::vector<Rec> r(blocks);
    hipMemcpy(r.data(), d, blocks * sizeof(Rec), hipMemcpyDeviceToHost);
    uint64_t tmin = ~0ull;
    for (auto &x : r) tmin = std::min(tmin, x.t0);
    // HW_ID (gfx9): wave_id [3:0], simd_id [5:4], pipe_id [7:6], cu_id [11:8], sh_id [12], se_id [15:13]
    auto cu_of = [](const Rec &x) { return (x.xcc << 16) | (((x.hwid >> 13) & 7) << 8) | (((x.hwid >> 12) & 1) << 4) | ((x.hwid >> 8) & 15); };
    printf("# block  xcc se sh cu simd  start_us end_us\n");
    for (int b = 0; b < 80; ++b)
        printf("%5d   %2u %2u %2u %2u %2u   %8.2f %8.2f\n", b, r[b].xcc, (r[b].hwid >> 13) & 7, (r[b].hwid >> 12) & 1,
               (r[b].hwid >> 8) & 15, (r[b].hwid >> 4) & 3, (r[b].t0 - tmin) / 100.0, (r[b].t1 - tmin) / 100.0);
    std::map<uint32_t, std::vector<int>> by_cu;
    for (int b = 0; b < blocks; ++b) by_cu[cu_of(r[b])].push_back(b);
    printf("# %zu distinct CUs; workgroups per CU: %zu..%zu\n", by_cu.size(),
           std::min_element(by_cu.begin(), by_cu.end(), [](auto &a, auto &c) { return a.second.size() < c.second.size(); })->second.size(),
           std::max_element(by_cu.begin(), by_cu.end(), [](auto &a, auto &c) { return a.second.size() < c.second.size(); })->second.size());
    int shown = 0;
    for (auto &kv : by_cu) {
        if (shown++ >= 6) break;
        std::vector<int> v = kv.second;
        std::sort(v.begin(), v.end(), [&](int a, int c) { return r[a].t0 < r[c].t0; });
        printf("CU %05x:", kv.first);
        for (int b : v) printf(" %d@%.1f", b, (r[b].t0 - tmin) / 100.0);
        printf("\n");
    }
    // first-generation residents of each CU: the blocks that started within 5 us of the launch
    std::map<int, int> hist;  // how many first-generation blocks per CU
    std::map<uint32_t, int> delta_hist;
    for (auto &kv : by_cu) {
        std::vector<int> first;
        for (int b : kv.second)
            if (r[b].t0 - tmin < 500) first.push_back(b);
        hist[(int)first.size()]++;
        std::sort(first.begin(), first.end());
        for (size_t i = 1; i < first.size(); ++i) delta_hist[(uint32_t)(first[i] - first[i - 1])]++;
    }
    for (auto &kv : hist) printf("# CUs with %d first-generation workgroups: %d\n", kv.first, kv.second);
    printf("# block-index distance between first-generation workgroups sharing a CU (distance: count):");
    for (auto &kv : delta_hist) printf(" %u:%d", kv.first, kv.second);
    printf("\n");
    double spread = 0;
    int ncu = 0;
    for (auto &kv : by_cu) {
        uint64_t lo = ~0ull, hi = 0;
        for (int b : kv.second)
            if (r[b].t0 - tmin < 500) {
                lo = std::min(lo, r[b].t0);
                hi = std::max(hi, r[b].t0);
            }
        if (hi >= lo) {
            spread += (hi - lo) / 100.0;
            ++ncu;
        }
    }
    printf("# mean start-time spread of the first generation inside a CU: %.2f us\n", ncu ? spread / ncu : 0.0);
    hipFree(d);
    return 0;
}
