// Host-link probe: what one process gets out of the PCIe link on this box, in the shapes serverRound uses.
//   hipcc --offload-arch=gfx950 -O2 -o tools/pcie_probe tools/pcie_probe.hip ; tools/pcie_probe
// Uploads of 12 MiB blocks (one ciphertext at N=2^16, L=12) from pinned memory, 1.5 GiB per arm, by number of copy
// streams and hipHostMalloc flags; downloads likewise.  (HSA_ENABLE_SDMA=0 in the environment switches the runtime from
// the SDMA engines to copy kernels.)
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <thread>
#include <atomic>
#include <string>
#include <fcntl.h>
#include <unistd.h>
#include <sys/mman.h>

#define CK(x)                                                                      \
    do {                                                                           \
        hipError_t e_ = (x);                                                       \
        if (e_ != hipSuccess) {                                                    \
            std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));            \
            std::exit(1);                                                          \
        }                                                                          \
    } while (0)

static double now_ms() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// `pcie_probe reads <dir>`: how fast T threads bring 12 MiB blocks of files in <dir> (tmpfs) into memory with pread(), into
// ordinary and into pinned buffers, and with an upload of every block behind it
static int reads_probe(const char *dir) {
    const size_t blk = 12u << 20, per_file = 18, n_files = 8, n_blk = per_file * n_files;
    std::vector<int> fds;
    {
        std::vector<char> src(blk, 7);
        for (size_t f = 0; f < n_files; ++f) {
            const std::string path = std::string(dir) + "/probe" + std::to_string(f) + ".bin";
            const int fd = ::open(path.c_str(), O_RDWR | O_CREAT | O_TRUNC, 0644);
            if (fd < 0) return std::perror("open"), 1;
            for (size_t b = 0; b < per_file; ++b)
                if (::pwrite(fd, src.data(), blk, (off_t)(b * blk)) != (ssize_t)blk) return std::perror("pwrite"), 1;
            fds.push_back(fd);
            ::unlink(path.c_str());
        }
    }
    const size_t ring = 16;
    char *pin = nullptr, *d = nullptr;
    CK(hipHostMalloc(&pin, blk * ring * 2, hipHostMallocDefault));
    CK(hipMalloc(&d, blk * n_blk));
    std::vector<char> plain(blk * ring * 2, 1);
    std::memset(pin, 1, blk * ring * 2);
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    std::printf("# %zu files x %zu blocks of %zu MiB in %s; every thread owns one buffer slot per pass\n", n_files, per_file, blk >> 20, dir);
    std::printf("%-8s %16s %16s %22s\n", "threads", "pread->malloc", "pread->pinned", "pread->pinned + upload");
    for (unsigned T : {1u, 2u, 4u, 8u, 12u, 16u}) {
        double rates[3];
        for (int mode = 0; mode < 3; ++mode) {
            char *base = mode == 0 ? plain.data() : pin;
            std::atomic<size_t> next{0};
            std::vector<hipEvent_t> ev(T * 2);
            for (auto &e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            CK(hipDeviceSynchronize());
            const double t0 = now_ms();
            std::vector<std::thread> pool;
            for (unsigned t = 0; t < T; ++t)
                pool.emplace_back([&, t] {
                    int flip = 0;
                    bool used[2] = {false, false};
                    for (size_t b; (b = next.fetch_add(1)) < n_blk;) {
                        const size_t slot = (size_t)t * 2 + flip;
                        char *dst = base + (slot % (ring * 2)) * blk;
                        if (mode == 2 && used[flip]) (void)hipEventSynchronize(ev[slot]);
                        size_t got = 0;
                        while (got < blk) {
                            const ssize_t r = ::pread(fds[b % n_files], dst + got, blk - got, (off_t)((b / n_files) * blk + got));
                            if (r <= 0) std::abort();
                            got += (size_t)r;
                        }
                        if (mode == 2) {
                            (void)hipMemcpyAsync(d + b * blk, dst, blk, hipMemcpyHostToDevice, st);
                            (void)hipEventRecord(ev[slot], st);
                            used[flip] = true;
                        }
                        flip ^= 1;
                    }
                });
            for (auto &th : pool) th.join();
            CK(hipStreamSynchronize(st));
            rates[mode] = (double)(blk * n_blk) / (now_ms() - t0) * 1e-6;
            for (auto &e : ev) CK(hipEventDestroy(e));
        }
        std::printf("%-8u %11.1f GB/s %11.1f GB/s %17.1f GB/s\n", T, rates[0], rates[1], rates[2]);
    }
    return 0;
}

// `pcie_probe reg <dir>`: can the DMA engine read a tmpfs file's pages directly?  mmap(MAP_SHARED) + hipHostRegister of
// whole files (serial and one thread per file), uploads of 12 MiB blocks straight from the mappings, hipHostUnregister.
static int reg_probe(const char *dir) {
    const size_t blk = 12u << 20, per_file = 18, n_files = 8, file_bytes = blk * per_file;
    std::vector<int> fds;
    {
        std::vector<char> src(blk, 7);
        for (size_t f = 0; f < n_files; ++f) {
            const std::string path = std::string(dir) + "/probe" + std::to_string(f) + ".bin";
            const int fd = ::open(path.c_str(), O_RDWR | O_CREAT | O_TRUNC, 0644);
            if (fd < 0) return std::perror("open"), 1;
            for (size_t b = 0; b < per_file; ++b)
                if (::pwrite(fd, src.data(), blk, (off_t)(b * blk)) != (ssize_t)blk) return std::perror("pwrite"), 1;
            fds.push_back(fd);
            ::unlink(path.c_str());
        }
    }
    char *d = nullptr;
    CK(hipMalloc(&d, file_bytes * n_files));
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    for (int mode = 0; mode < 3; ++mode) {  // 0: serial registration, read-only flag; 1: one thread per file; 2: MAP_POPULATE first
        std::vector<char *> maps(n_files, nullptr);
        const double t0 = now_ms();
        for (size_t f = 0; f < n_files; ++f) {
            void *m = ::mmap(nullptr, file_bytes, PROT_READ | PROT_WRITE, MAP_SHARED | (mode == 2 ? MAP_POPULATE : 0), fds[f], 0);
            if (m == MAP_FAILED) return std::perror("mmap"), 1;
            maps[f] = static_cast<char *>(m);
        }
        const double t_map = now_ms() - t0;
        std::atomic<int> failed{0};
        auto reg = [&](size_t f) {
            if (hipHostRegister(maps[f], file_bytes, hipHostRegisterDefault) != hipSuccess) failed++;
        };
        const double t1 = now_ms();
        if (mode == 1) {
            std::vector<std::thread> pool;
            for (size_t f = 0; f < n_files; ++f) pool.emplace_back(reg, f);
            for (auto &t : pool) t.join();
        } else {
            for (size_t f = 0; f < n_files; ++f) reg(f);
        }
        const double t_reg = now_ms() - t1;
        if (failed) {
            std::printf("mode %d: hipHostRegister of a tmpfs mapping refused (%s)\n", mode, hipGetErrorString(hipGetLastError()));
            for (size_t f = 0; f < n_files; ++f) ::munmap(maps[f], file_bytes);
            continue;
        }
        CK(hipDeviceSynchronize());
        const double t2 = now_ms();
        for (size_t b = 0; b < per_file; ++b)
            for (size_t f = 0; f < n_files; ++f)
                CK(hipMemcpyAsync(d + (f * per_file + b) * blk, maps[f] + b * blk, blk, hipMemcpyHostToDevice, st));
        CK(hipStreamSynchronize(st));
        const double t_up = now_ms() - t2;
        const double t3 = now_ms();
        for (size_t f = 0; f < n_files; ++f) CK(hipHostUnregister(maps[f]));
        const double t_unreg = now_ms() - t3;
        for (size_t f = 0; f < n_files; ++f) ::munmap(maps[f], file_bytes);
        std::printf("mode %d (%s): mmap %.1f ms, register %zu x %zu MiB %.1f ms, upload %.1f ms = %.1f GB/s, unregister %.1f ms\n", mode,
                    mode == 0 ? "serial" : mode == 1 ? "one thread per file" : "serial, MAP_POPULATE", t_map, n_files, file_bytes >> 20, t_reg,
                    t_up, (double)(file_bytes * n_files) / t_up * 1e-6, t_unreg);
    }
    return 0;
}

int main(int argc, char **argv) {
    if (argc > 2 && std::string(argv[1]) == "reads") return reads_probe(argv[2]);
    if (argc > 2 && std::string(argv[1]) == "reg") return reg_probe(argv[2]);
    const size_t blk = 12u << 20, n_blk = 128, ring = 16;
    char *d = nullptr;
    CK(hipMalloc(&d, blk * n_blk));
    struct Flag {
        const char *name;
        unsigned v;
    } flags[] = {{"default", hipHostMallocDefault}, {"non-coherent", hipHostMallocNonCoherent}, {"numa-user", hipHostMallocNumaUser},
                 {"coherent", hipHostMallocCoherent}};
    std::printf("# %zu blocks of %zu MiB per arm, pinned ring of %zu blocks\n", n_blk, blk >> 20, ring);
    std::printf("%-14s %-8s %8s %8s\n", "pinned flags", "streams", "up GB/s", "down GB/s");
    for (const Flag &f : flags) {
        char *h = nullptr;
        if (hipHostMalloc(&h, blk * ring, f.v) != hipSuccess) {
            std::printf("%-14s hipHostMalloc refused\n", f.name);
            (void)hipGetLastError();
            continue;
        }
        std::memset(h, 1, blk * ring);
        for (int ns = 1; ns <= 4; ++ns) {
            hipStream_t st[4];
            for (int i = 0; i < ns; ++i) CK(hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking));
            double rate[2];
            for (int dir = 0; dir < 2; ++dir) {
                double best = 0;
                for (int rep = 0; rep < 3; ++rep) {
                    CK(hipDeviceSynchronize());
                    const double t0 = now_ms();
                    for (size_t b = 0; b < n_blk; ++b) {
                        if (dir == 0) CK(hipMemcpyAsync(d + b * blk, h + (b % ring) * blk, blk, hipMemcpyHostToDevice, st[b % ns]));
                        else CK(hipMemcpyAsync(h + (b % ring) * blk, d + b * blk, blk, hipMemcpyDeviceToHost, st[b % ns]));
                    }
                    for (int i = 0; i < ns; ++i) CK(hipStreamSynchronize(st[i]));
                    const double dt = now_ms() - t0;
                    best = std::max(best, (double)(blk * n_blk) / dt * 1e-6);
                }
                rate[dir] = best;
            }
            std::printf("%-14s %-8d %8.1f %8.1f\n", f.name, ns, rate[0], rate[1]);
            for (int i = 0; i < ns; ++i) CK(hipStreamDestroy(st[i]));
        }
        CK(hipHostFree(h));
    }
    {   // pageable memory, synchronous copies (mkckks_upload / mkckks_download)
        std::vector<char> pg(blk * ring, 1);
        CK(hipDeviceSynchronize());
        double t0 = now_ms();
        for (size_t b = 0; b < n_blk; ++b) CK(hipMemcpy(d + b * blk, pg.data() + (b % ring) * blk, blk, hipMemcpyHostToDevice));
        const double up = (double)(blk * n_blk) / (now_ms() - t0) * 1e-6;
        t0 = now_ms();
        for (size_t b = 0; b < n_blk; ++b) CK(hipMemcpy(pg.data() + (b % ring) * blk, d + b * blk, blk, hipMemcpyDeviceToHost));
        const double down = (double)(blk * n_blk) / (now_ms() - t0) * 1e-6;
        std::printf("%-14s %-8s %8.1f %8.1f\n", "pageable", "sync", up, down);
    }
    // both directions at once (one stream each)
    {
        char *h = nullptr;
        CK(hipHostMalloc(&h, blk * ring * 2, hipHostMallocDefault));
        std::memset(h, 1, blk * ring * 2);
        hipStream_t a, b2;
        CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
        CK(hipStreamCreateWithFlags(&b2, hipStreamNonBlocking));
        char *d2 = nullptr;
        CK(hipMalloc(&d2, blk * n_blk));
        CK(hipDeviceSynchronize());
        const double t0 = now_ms();
        for (size_t b = 0; b < n_blk; ++b) {
            CK(hipMemcpyAsync(d + b * blk, h + (b % ring) * blk, blk, hipMemcpyHostToDevice, a));
            CK(hipMemcpyAsync(h + (ring + b % ring) * blk, d2 + b * blk, blk, hipMemcpyDeviceToHost, b2));
        }
        CK(hipStreamSynchronize(a));
        CK(hipStreamSynchronize(b2));
        const double dt = now_ms() - t0;
        std::printf("%-14s %-8s %8.1f %8.1f   (both directions at once, each)\n", "default", "1+1", (double)(blk * n_blk) / dt * 1e-6,
                    (double)(blk * n_blk) / dt * 1e-6);
    }
    return 0;
}
