// comm.cpp -- see comm.hpp
#include "comm.hpp"

#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstring>
#include <mutex>
#include <stdexcept>
#include <string>

namespace mk {
namespace {

struct Rccl {
    void *handle = nullptr;
    std::string path;
    decltype(&ncclGetUniqueId) get_unique_id = nullptr;
    decltype(&ncclCommInitRank) comm_init_rank = nullptr;
    decltype(&ncclCommDestroy) comm_destroy = nullptr;
    decltype(&ncclReduceScatter) reduce_scatter = nullptr;
    decltype(&ncclGetErrorString) error_string = nullptr;
};

template <typename F>
void bind(void *h, const char *name, F &fn) {
    fn = reinterpret_cast<F>(dlsym(h, name));
    if (!fn) throw std::runtime_error(std::string("librccl: missing symbol ") + name);
}

Rccl &rccl() {
    static Rccl r;
    static std::once_flag once;
    static std::string failure;
    std::call_once(once, [] {
        // SONAME lookup: an RCCL already in the process (e.g. torch's) is reused, otherwise RUNPATH / ld.so.conf apply
        void *h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!h) {
            const char *e = dlerror();
            failure = std::string("cannot load librccl.so.1: ") + (e ? e : "unknown error");
            return;
        }
        try {
            r.handle = h;
            bind(h, "ncclGetUniqueId", r.get_unique_id);
            bind(h, "ncclCommInitRank", r.comm_init_rank);
            bind(h, "ncclCommDestroy", r.comm_destroy);
            bind(h, "ncclReduceScatter", r.reduce_scatter);
            bind(h, "ncclGetErrorString", r.error_string);
            Dl_info info{};
            if (dladdr(reinterpret_cast<void *>(r.reduce_scatter), &info) && info.dli_fname) r.path = info.dli_fname;
        } catch (const std::exception &e) {
            failure = e.what();
            r.handle = nullptr;
        }
    });
    if (!r.handle) throw std::runtime_error(failure);
    return r;
}

void check(ncclResult_t res, const char *what) {
    if (res != ncclSuccess) throw std::runtime_error(std::string(what) + ": " + rccl().error_string(res));
}

}  // namespace

const char *comm_library_path() { return rccl().path.c_str(); }

void comm_unique_id(void *h_id_out) {
    static_assert(sizeof(ncclUniqueId) == COMM_ID_BYTES, "unique id size");
    ncclUniqueId id;
    check(rccl().get_unique_id(&id), "ncclGetUniqueId");
    std::memcpy(h_id_out, &id, sizeof(id));
}

void *comm_create(int device, const void *h_id, int n_ranks, int rank) {
    if (n_ranks < 1 || rank < 0 || rank >= n_ranks) throw std::invalid_argument("communicator: bad rank / size");
    if (hipSetDevice(device) != hipSuccess) throw std::runtime_error("communicator: hipSetDevice failed");
    ncclUniqueId id;
    std::memcpy(&id, h_id, sizeof(id));
    ncclComm_t comm = nullptr;
    check(rccl().comm_init_rank(&comm, n_ranks, id, rank), "ncclCommInitRank");
    return comm;
}

void comm_destroy(void *comm) {
    if (comm) check(rccl().comm_destroy(static_cast<ncclComm_t>(comm)), "ncclCommDestroy");
}

void comm_reduce_scatter_u64(void *comm, const uint64_t *d_send, uint64_t *d_recv, size_t recv_words, void *stream) {
    check(rccl().reduce_scatter(d_send, d_recv, recv_words, ncclUint64, ncclSum, static_cast<ncclComm_t>(comm),
                                static_cast<hipStream_t>(stream)),
          "ncclReduceScatter");
}

}  // namespace mk
