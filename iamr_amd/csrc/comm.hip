// iamr_amd/csrc/comm.hip -- communicator back-ends for the multi-GPU path.
//
//  * RcclComm     : one process per GPU; halo / FillPatch traffic = grouped ncclSend/ncclRecv of packed device
//                   buffers on the compute stream (xGMI peer-to-peer inside a node), scalar reductions
//                   (norms, dt, Krylov dots: the ParallelDescriptor::Reduce* call sites of SURVEY 2.3) =
//                   ncclAllReduce of a few doubles.  librccl is dlopen'ed lazily so that single-GPU use
//                   never depends on it.
//  * CallbackComm : transport supplied by the host program through two C callbacks operating on host
//                   buffers.  Used by the tests to run the *same* rank-aware code with torch.distributed
//                   (gloo) as transport, including two ranks sharing one GPU.
//
// STATUS (round 1): the rank-aware code is exercised by tests/test_gpu_dist.py through CallbackComm (2 and 4 ranks on one GPU,
// results identical to 1 rank).  The RCCL back end runs on the 1-GPU box with a 1-rank communicator: ncclGetUniqueId /
// ncclCommInitRank / ncclAllReduce under two full time steps, and grouped ncclSend / ncclRecv as a loop-back to the own rank
// (iamrx_comm_probe_exchange).  Peer-to-peer between two GPUs could not be executed in this round.
#include "mf.h"
#include "../../include/iamrx.h"
#include <dlfcn.h>
#include <cstring>
#include <vector>

namespace iamrx {

// ---- minimal RCCL surface (matches <rccl/rccl.h>; resolved with dlsym) ---------------------------------
typedef struct ncclComm* ncclComm_t;
struct NcclUniqueId { char internal[128]; };
enum { kNcclSum = 0, kNcclMax = 2, kNcclMin = 3 };   // ncclRedOp_t
enum { kNcclDouble = 8 };                             // ncclDataType_t: ncclFloat64

struct RcclApi {
    void* h = nullptr;
    int (*GetUniqueId)(NcclUniqueId*) = nullptr;
    int (*CommInitRank)(ncclComm_t*, int, NcclUniqueId, int) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*Send)(const void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*Recv)(void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    void load()
    {
        if (h) return;
        h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) throw Error(std::string("iamrx: cannot load librccl: ") + dlerror());
        auto sym = [&](const char* n) { void* p = dlsym(h, n); if (!p) throw Error(std::string("iamrx: librccl lacks ") + n); return p; };
        GetUniqueId = (decltype(GetUniqueId))sym("ncclGetUniqueId");
        CommInitRank = (decltype(CommInitRank))sym("ncclCommInitRank");
        AllReduce = (decltype(AllReduce))sym("ncclAllReduce");
        Send = (decltype(Send))sym("ncclSend");
        Recv = (decltype(Recv))sym("ncclRecv");
        GroupStart = (decltype(GroupStart))sym("ncclGroupStart");
        GroupEnd = (decltype(GroupEnd))sym("ncclGroupEnd");
        GetErrorString = (decltype(GetErrorString))sym("ncclGetErrorString");
    }
};
static RcclApi g_rccl;

#define IAMRX_NCCL_CHECK(expr)                                                                                  \
    do {                                                                                                        \
        int _r = (expr);                                                                                        \
        if (_r != 0) throw Error(std::string("RCCL error: ") + (g_rccl.GetErrorString ? g_rccl.GetErrorString(_r) : "?") + " at " + __FILE__ + ":" + std::to_string(__LINE__)); \
    } while (0)

struct RcclComm : Comm {
    ncclComm_t comm = nullptr;
    double* d_red = nullptr;
    double* h_red = nullptr;
    RcclComm(const NcclUniqueId& id, int r, int n)
    {
        rank = r; nranks = n;
        g_rccl.load();
        IAMRX_NCCL_CHECK(g_rccl.CommInitRank(&comm, n, id, r));
        IAMRX_HIP_CHECK(hipMalloc(&d_red, 64 * sizeof(double)));
        IAMRX_HIP_CHECK(hipHostMalloc(&h_red, 64 * sizeof(double)));
    }
    int red_cap = 64;
    void allreduce_device(double* dev_vals, int n, ReduceOp op, hipStream_t s) override
    {
        const int rop = op == ReduceOp::Sum ? kNcclSum : (op == ReduceOp::Max ? kNcclMax : kNcclMin);
        IAMRX_NCCL_CHECK(g_rccl.AllReduce(dev_vals, dev_vals, (size_t)n, kNcclDouble, rop, comm, s));
    }
    void allreduce(double* vals, int n, ReduceOp op) override
    {
        hipStream_t s = Context::get().stream;
        if (n > red_cap) {               // rare large host-side reductions (the tag maps of a regrid)
            IAMRX_HIP_CHECK(hipStreamSynchronize(s));
            (void)hipFree(d_red); (void)hipHostFree(h_red);
            red_cap = n;
            IAMRX_HIP_CHECK(hipMalloc(&d_red, (size_t)red_cap * sizeof(double)));
            IAMRX_HIP_CHECK(hipHostMalloc(&h_red, (size_t)red_cap * sizeof(double)));
        }
        std::memcpy(h_red, vals, n * sizeof(double));
        IAMRX_HIP_CHECK(hipMemcpyAsync(d_red, h_red, n * sizeof(double), hipMemcpyHostToDevice, s));
        const int rop = op == ReduceOp::Sum ? kNcclSum : (op == ReduceOp::Max ? kNcclMax : kNcclMin);
        IAMRX_NCCL_CHECK(g_rccl.AllReduce(d_red, d_red, (size_t)n, kNcclDouble, rop, comm, s));
        IAMRX_HIP_CHECK(hipMemcpyAsync(h_red, d_red, n * sizeof(double), hipMemcpyDeviceToHost, s));
        IAMRX_HIP_CHECK(hipStreamSynchronize(s));
        std::memcpy(vals, h_red, n * sizeof(double));
    }
    void exchange(const std::vector<Message>& sends, const std::vector<Message>& recvs, hipStream_t s) override
    {
        // all point-to-point calls of one halo exchange in ONE group: each neighbour pair maps to its own xGMI link
        IAMRX_NCCL_CHECK(g_rccl.GroupStart());
        for (auto& m : recvs) IAMRX_NCCL_CHECK(g_rccl.Recv(m.dev_ptr, m.count, kNcclDouble, m.peer, comm, s));
        for (auto& m : sends) IAMRX_NCCL_CHECK(g_rccl.Send(m.dev_ptr, m.count, kNcclDouble, m.peer, comm, s));
        IAMRX_NCCL_CHECK(g_rccl.GroupEnd());
    }
};

struct CallbackComm : Comm {
    iamrx_allreduce_cb ar;
    iamrx_exchange_cb ex;
    CallbackComm(int r, int n, iamrx_allreduce_cb a, iamrx_exchange_cb e) : ar(a), ex(e) { rank = r; nranks = n; }
    void allreduce(double* vals, int n, ReduceOp op) override { ar(vals, n, op == ReduceOp::Sum ? 0 : (op == ReduceOp::Max ? 1 : 2)); }
    // host-staged transport of the tests: drain, copy down, reduce through the callback, copy up (the RCCL transport reduces in place)
    void allreduce_device(double* dev_vals, int n, ReduceOp op, hipStream_t s) override
    {
        std::vector<double> h((size_t)n);
        IAMRX_HIP_CHECK(hipMemcpyAsync(h.data(), dev_vals, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, s));
        IAMRX_HIP_CHECK(hipStreamSynchronize(s));
        allreduce(h.data(), n, op);
        IAMRX_HIP_CHECK(hipMemcpyAsync(dev_vals, h.data(), (size_t)n * sizeof(double), hipMemcpyHostToDevice, s));
        IAMRX_HIP_CHECK(hipStreamSynchronize(s));
    }
    void exchange(const std::vector<Message>& sends, const std::vector<Message>& recvs, hipStream_t s) override
    {
        IAMRX_HIP_CHECK(hipStreamSynchronize(s));
        std::vector<std::vector<double>> sb(sends.size()), rb(recvs.size());
        std::vector<int> sp, rp;
        std::vector<double*> sptr, rptr;
        std::vector<long> sc, rc;
        for (size_t i = 0; i < sends.size(); ++i) {
            sb[i].resize(sends[i].count);
            IAMRX_HIP_CHECK(hipMemcpy(sb[i].data(), sends[i].dev_ptr, sends[i].count * sizeof(double), hipMemcpyDeviceToHost));
            sp.push_back(sends[i].peer); sptr.push_back(sb[i].data()); sc.push_back((long)sends[i].count);
        }
        for (size_t i = 0; i < recvs.size(); ++i) {
            rb[i].resize(recvs[i].count);
            rp.push_back(recvs[i].peer); rptr.push_back(rb[i].data()); rc.push_back((long)recvs[i].count);
        }
        ex((int)sends.size(), sp.data(), sptr.data(), sc.data(), (int)recvs.size(), rp.data(), rptr.data(), rc.data());
        for (size_t i = 0; i < recvs.size(); ++i)
            IAMRX_HIP_CHECK(hipMemcpy(recvs[i].dev_ptr, rb[i].data(), recvs[i].count * sizeof(double), hipMemcpyHostToDevice));
    }
};

}  // namespace iamrx

using namespace iamrx;
static thread_local std::string g_cerr;
extern "C" {

int iamrx_comm_get_unique_id(char id[128])
{
    try { g_rccl.load(); NcclUniqueId u; IAMRX_NCCL_CHECK(g_rccl.GetUniqueId(&u)); std::memcpy(id, u.internal, 128); return 0; }
    catch (const std::exception& e) { g_cerr = e.what(); return 1; }
}
int iamrx_comm_init_rccl(const char id[128], int rank, int nranks)
{
    try {
        IAMRX_ASSERT(Context::get().stream != nullptr);
        NcclUniqueId u; std::memcpy(u.internal, id, 128);
        Context::get().comm = std::make_unique<RcclComm>(u, rank, nranks);
        return 0;
    } catch (const std::exception& e) { g_cerr = e.what(); return 1; }
}
int iamrx_comm_init_callback(int rank, int nranks, iamrx_allreduce_cb ar, iamrx_exchange_cb ex)
{
    try { Context::get().comm = std::make_unique<CallbackComm>(rank, nranks, ar, ex); return 0; }
    catch (const std::exception& e) { g_cerr = e.what(); return 1; }
}
int iamrx_comm_rank(int* rank, int* nranks)
{
    auto& c = Context::get().comm;
    if (rank) *rank = c ? c->rank : 0;
    if (nranks) *nranks = c ? c->nranks : 1;
    return 0;
}
// transport probe: every rank sends `count` doubles to `peer` and receives as many from it through the installed communicator's
// exchange() (the halo-exchange primitive); peer == own rank is a loop-back through the same Send / Recv calls.  Returns 0 if the
// received data are the pattern the peer sent.
int iamrx_comm_probe_exchange(int peer, long count)
{
    try {
        auto& ctx = Context::get();
        IAMRX_ASSERT(ctx.comm && count > 0);
        const int me = ctx.comm->rank;
        std::vector<double> h((size_t)count), back((size_t)count, -1.0);
        for (long i = 0; i < count; ++i) h[(size_t)i] = 1000.0 * me + 0.5 * (double)i;
        double* ds = (double*)ctx.alloc((size_t)count * sizeof(double));
        double* dr = (double*)ctx.alloc((size_t)count * sizeof(double));
        IAMRX_HIP_CHECK(hipMemcpy(ds, h.data(), (size_t)count * sizeof(double), hipMemcpyHostToDevice));
        IAMRX_HIP_CHECK(hipMemset(dr, 0, (size_t)count * sizeof(double)));
        std::vector<Message> sends{{peer, ds, (size_t)count}}, recvs{{peer, dr, (size_t)count}};
        ctx.comm->exchange(sends, recvs, ctx.stream);
        ctx.sync();
        IAMRX_HIP_CHECK(hipMemcpy(back.data(), dr, (size_t)count * sizeof(double), hipMemcpyDeviceToHost));
        ctx.free(ds); ctx.free(dr);
        for (long i = 0; i < count; ++i)
            if (back[(size_t)i] != 1000.0 * peer + 0.5 * (double)i) throw Error("iamrx_comm_probe_exchange: wrong data received");
        return 0;
    } catch (const std::exception& e) { g_cerr = e.what(); return 1; }
}
const char* iamrx_comm_last_error(void) { return g_cerr.c_str(); }

}  // extern "C"
