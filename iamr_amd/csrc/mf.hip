// iamr_amd/csrc/mf.hip -- Context (stream, caching allocator), Layout, MultiFab, ghost-exchange plans.
// Plays the role of AMReX MultiFab / FillBoundary at IAMR's call sites (SURVEY 2.3 "Same-level ghost
// exchange": reference Source/MacProj.cpp:1127, Source/Projection.cpp:338-339, ...).
#include "mf.h"
#include <climits>
#include <array>
#include <execinfo.h>
#include <dlfcn.h>
#include <chrono>
#include <functional>
#include <tuple>
#include "launch.h"
#include "kernels.h"
#include <algorithm>
#include <cstring>
#include <atomic>

namespace iamrx {

// ------------------------------------------------------------------ Context
Context& Context::get()
{
    static Context c;
    return c;
}

void Context::init(int dev)
{
    tuning_load_environment();
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev == 0)
        throw Error("iamrx: no HIP device available -- the product path has no CPU fallback");
    device = dev;
    if (live_blocks.empty() && chunk_end.empty() && free_blocks.empty()) arena = tune("ARENA", 1) != 0;      // (fixed for the life of the allocator's blocks)
    IAMRX_HIP_CHECK(hipSetDevice(dev));
    if (!stream) IAMRX_HIP_CHECK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    if (!side) {
        IAMRX_HIP_CHECK(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
        IAMRX_HIP_CHECK(hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming));
        IAMRX_HIP_CHECK(hipEventCreateWithFlags(&ev_join, hipEventDisableTiming));
    }
    if (!comm) comm = std::make_unique<Comm>();
    ensure_scratch(1 << 16);
}

void Context::fork_side()
{
    IAMRX_HIP_CHECK(hipEventRecord(ev_fork, stream));
    IAMRX_HIP_CHECK(hipStreamWaitEvent(side, ev_fork, 0));
}

void Context::join_side()
{
    IAMRX_HIP_CHECK(hipEventRecord(ev_join, side));
    IAMRX_HIP_CHECK(hipStreamWaitEvent(stream, ev_join, 0));
    for (void* p : parked) free(p);          // whoever takes these blocks next is ordered behind the side stream's work
    parked.clear();
}

void Context::ensure_scratch(size_t n)
{
    if (n <= scratch_n) return;
    if (d_scratch) { sync(); IAMRX_HIP_CHECK(hipFree(d_scratch)); IAMRX_HIP_CHECK(hipHostFree(h_scratch)); }
    IAMRX_HIP_CHECK(hipMalloc(&d_scratch, n * sizeof(double)));
    IAMRX_HIP_CHECK(hipHostMalloc(&h_scratch, n * sizeof(double)));
    scratch_n = n;
}

// IAMRX_POISON_ALLOC=1 (debugging aid): device blocks are filled with 0xFF bytes (NaN as doubles) when they are handed out, so that a
// kernel reading memory nobody has written shows up deterministically instead of depending on what the recycled block held before
// (2: zeros instead -- if a result changes between 0, 1 and 2 some kernel reads memory nobody wrote)
static int poison_allocs()
{
    return (int)tune("POISON_ALLOC", 0);
}

// The device allocator (The_Arena's role).  hipMalloc maps pages at 25-35 GB/s and synchronises, so nothing inside a time step may reach it.
// Rounds 1-5 cached whole hipMalloc blocks by size: exact for a run whose arrays keep their sizes, but every regrid of a GROWING level
// asks for arrays a few per cent larger than anything in the cache (config C5: 3 457 hipMalloc calls, 226 GB mapped, 6.7 s of a 37 s run)
// while the blocks of the levels it replaced pile up unused.  Round 6 (IAMRX_ARENA = 1, default): blocks are carved (best fit, split) out
// of chunks of at least 1/8 of what is mapped so far (>= 256 MB), and a block that comes back merges with its free neighbours inside its
// chunk -- the space of the old level's arrays becomes one extent the new level's arrays are cut from.  Ordering is the cache's: a block
// handed out again is used behind its previous owner on `stream` (side-stream users park theirs until the join).
namespace {
void arena_insert_free(Context& c, char* p, size_t sz)
{
    // merge with the free block that ends at p and the one that starts at p + sz, unless a chunk boundary lies between
    auto chunk_of = [&](char* q) { auto it = c.chunk_end.upper_bound(q); --it; return it; };
    const auto ck = chunk_of(p);
    auto nx = c.free_addr.lower_bound(p);
    if (nx != c.free_addr.begin()) {
        auto pv = std::prev(nx);
        if (pv->first + pv->second == p && pv->first >= ck->first) {
            auto rng = c.free_blocks.equal_range(pv->second);
            for (auto it = rng.first; it != rng.second; ++it) if (it->second == (void*)pv->first) { c.free_blocks.erase(it); break; }
            p = pv->first; sz += pv->second;
            c.free_addr.erase(pv);
        }
    }
    nx = c.free_addr.lower_bound(p + sz);
    if (nx != c.free_addr.end() && nx->first == p + sz && p + sz < ck->second) {
        auto rng = c.free_blocks.equal_range(nx->second);
        for (auto it = rng.first; it != rng.second; ++it) if (it->second == (void*)nx->first) { c.free_blocks.erase(it); break; }
        sz += nx->second;
        c.free_addr.erase(nx);
    }
    c.free_addr[p] = sz;
    c.free_blocks.emplace(sz, (void*)p);
}
}  // namespace

void* Context::alloc(size_t bytes)
{
    if (bytes == 0) bytes = 256;
    bytes = (bytes + 255) & ~size_t(255);
    if (arena) {
        auto it = free_blocks.lower_bound(bytes);
        if (it == free_blocks.end()) {
            // a new chunk: the request, or 1/8 of everything mapped so far if that is more (at least 256 MB: small runs map one chunk)
            size_t csz = std::max(bytes, std::max(size_t(256) << 20, bytes_chunks / 8));
            csz = (csz + ((size_t(2) << 20) - 1)) & ~((size_t(2) << 20) - 1);
            void* q = nullptr;
            ++n_device_malloc;
            static const bool trace = tune("ALLOC_TRACE", 0) != 0;
            const auto t0 = std::chrono::steady_clock::now();
            hipError_t e = hipMalloc(&q, csz);
            if (e != hipSuccess) {
                (void)hipGetLastError();
                release_cache();
                csz = (bytes + ((size_t(2) << 20) - 1)) & ~((size_t(2) << 20) - 1);
                IAMRX_HIP_CHECK(hipMalloc(&q, csz));
            }
            if (trace) fprintf(stderr, "iamrx alloc: chunk of %.1f MB for a request of %.1f MB: %.2f ms (mapped %.1f GB, live %.1f GB)\n", csz / 1048576.0, bytes / 1048576.0,
                               std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(), (bytes_chunks + csz) / 1073741824.0, bytes_live / 1073741824.0);
            chunk_end[(char*)q] = (char*)q + csz;
            bytes_chunks += csz;
            bytes_cached += csz;
            arena_insert_free(*this, (char*)q, csz);
            it = free_blocks.lower_bound(bytes);
        }
        char* p = (char*)it->second;
        const size_t have = it->first;
        free_blocks.erase(it);
        free_addr.erase(p);
        if (have > bytes) arena_insert_free(*this, p + bytes, have - bytes);       // (no neighbour to merge with: p was one free extent)
        bytes_cached -= bytes;
        live_blocks[p] = bytes;
        bytes_live += bytes;
        if (poison_allocs()) IAMRX_HIP_CHECK(hipMemsetAsync(p, poison_allocs() == 2 ? 0x00 : 0xFF, bytes, stream));
        return p;
    }
    auto it = free_blocks.lower_bound(bytes);
    if (it != free_blocks.end() && it->first <= bytes + bytes / 8 + 4096) {
        void* p = it->second;
        size_t sz = it->first;
        free_blocks.erase(it);
        bytes_cached -= sz;
        live_blocks[p] = sz;
        bytes_live += sz;
        if (poison_allocs()) IAMRX_HIP_CHECK(hipMemsetAsync(p, poison_allocs() == 2 ? 0x00 : 0xFF, sz, stream));      // debug: every block starts as NaNs
        return p;
    }
    void* p = nullptr;
    ++n_device_malloc;
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess) {
        release_cache();
        IAMRX_HIP_CHECK(hipMalloc(&p, bytes));
    }
    live_blocks[p] = bytes;
    bytes_live += bytes;
    if (poison_allocs()) IAMRX_HIP_CHECK(hipMemsetAsync(p, poison_allocs() == 2 ? 0x00 : 0xFF, bytes, stream));
    return p;
}

void Context::free(void* p)
{
    if (!p) return;
    auto it = live_blocks.find(p);
    if (it == live_blocks.end()) return;
    size_t sz = it->second;
    live_blocks.erase(it);
    bytes_live -= sz;
    bytes_cached += sz;
    if (arena) arena_insert_free(*this, (char*)p, sz);
    else free_blocks.emplace(sz, p);
}

// returns what nobody uses to the driver: cached blocks (IAMRX_ARENA = 0) / chunks that are free from end to end
void Context::release_cache()
{
    sync();
    if (arena) {
        for (auto ck = chunk_end.begin(); ck != chunk_end.end();) {
            auto f = free_addr.find(ck->first);
            const size_t csz = (size_t)(ck->second - ck->first);
            if (f == free_addr.end() || f->second != csz) { ++ck; continue; }
            auto rng = free_blocks.equal_range(csz);
            for (auto it = rng.first; it != rng.second; ++it) if (it->second == (void*)ck->first) { free_blocks.erase(it); break; }
            free_addr.erase(f);
            (void)hipFree(ck->first);
            bytes_cached -= csz; bytes_chunks -= csz;
            ck = chunk_end.erase(ck);
        }
        return;
    }
    for (auto& kv : free_blocks) (void)hipFree(kv.second);
    free_blocks.clear();
    bytes_cached = 0;
}

// ---- tuning registry
namespace {
std::map<std::string, double>& tuning_map() { static auto* m = new std::map<std::string, double>(); return *m; }
}  // namespace
namespace {
struct TuneSlot { const char* key = nullptr; double val = 0.0; bool set = false; uint64_t ver = 0; };
constexpr size_t kTuneSlots = 2048;
TuneSlot g_tune_slots[kTuneSlots];
uint64_t g_tune_version = 1;
}  // namespace
extern "C" char** environ;
void tuning_load_environment()
{
    auto& m = tuning_map();
    for (char** e = environ; e && *e; ++e) {
        if (strncmp(*e, "IAMRX_", 6) != 0) continue;
        const char* eq = strchr(*e, '=');
        if (!eq) continue;
        const std::string key(*e + 6, eq - (*e + 6));
        char* end = nullptr;
        const double v = strtod(eq + 1, &end);
        if (end != eq + 1 && !m.count(key)) m[key] = v;           // values set through iamrx_tuning_set before init win
    }
    ++g_tune_version;
}
// The registry is read at the point of use, several times per kernel launch (tile shapes, kernel forms): a std::map<std::string> lookup
// there costs a string construction and O(log n) string compares -- of the order of a microsecond per launch on launch-bound sequences
// (7 700 launches per coarse step of the 128^3 + 128^3 hierarchy).  The keys are string literals, so the ADDRESS identifies the call
// site: a small open-addressed table keyed by it holds the value, tagged with the registry's version; iamrx_tuning_set bumps the version
// (every entry is then re-read from the map on its next use).  tune_by_name: for keys that are not literals (the C-ABI getter).
double tune_by_name(const char* key, double dflt)
{
    auto& m = tuning_map();
    auto it = m.find(key);
    return it == m.end() ? dflt : it->second;
}
double tune(const char* key, double dflt)
{
    size_t i = ((uintptr_t)key >> 2) & (kTuneSlots - 1);
    for (int probe = 0; probe < 16; ++probe, i = (i + 1) & (kTuneSlots - 1)) {
        TuneSlot& t = g_tune_slots[i];
        if (t.key == key && t.ver == g_tune_version) return t.set ? t.val : dflt;
        if (t.key == nullptr || t.ver != g_tune_version) {        // free (or stale: its owner re-reads like everybody else)
            auto& m = tuning_map();
            auto it = m.find(key);
            t.key = key; t.ver = g_tune_version; t.set = it != m.end(); t.val = t.set ? it->second : 0.0;
            return t.set ? t.val : dflt;
        }
    }
    return tune_by_name(key, dflt);
}
void tuning_set(const char* key, double value) { tuning_map()[key] = value; ++g_tune_version; }

// ---- scoped profiler
bool ProfScope::enabled = false;
namespace {
struct ScopeStat { double ms = 0.0; long calls = 0; };
std::map<std::string, ScopeStat>& scope_stats() { static auto* m = new std::map<std::string, ScopeStat>(); return *m; }
std::vector<std::string>& scope_stack() { static auto* v = new std::vector<std::string>(); return *v; }
double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
}  // namespace
ProfScope::ProfScope(const char* name) : on(enabled)
{
    if (!on) return;
    auto& st = scope_stack();
    key = st.empty() ? std::string(name) : st.back() + "/" + name;
    st.push_back(key);
    Context::get().sync();
    t0 = now_ms();
}
ProfScope::~ProfScope()
{
    if (!on) return;
    Context::get().sync();
    auto& e = scope_stats()[key];
    e.ms += now_ms() - t0; e.calls += 1;
    auto& st = scope_stack();
    if (!st.empty()) st.pop_back();
}
void scope_profile_enable(bool on, bool reset) { ProfScope::enabled = on; if (reset) { scope_stats().clear(); scope_stack().clear(); } }
std::string scope_profile_report()
{
    std::string out;
    char buf[256];
    for (auto& kv : scope_stats()) { snprintf(buf, sizeof buf, "%-70s %10.3f ms %8ld calls\n", kv.first.c_str(), kv.second.ms, kv.second.calls); out += buf; }
    return out;
}

// IAMRX_SYNC_TRACE = 1: every host synchronisation prints the return addresses of its callers (resolve with addr2line -e libiamrx.so):
// the tool behind the host_syncs_per_step figure of bench.py
// IAMRX_BLAS_TRACE = 1: every level-wide copy / fill / axpy prints its size and the return addresses of its callers (tools/sync_trace.py blas)
void trace_blas_site(const char* what, long points)
{
    if (tune("BLAS_TRACE", 0) == 0) return;
    void* bt[6];
    const int n = backtrace(bt, 6);
    Dl_info info;
    fprintf(stderr, "iamrx blas: %s %ld", what, points);
    for (int i = 2; i < n; ++i)
        if (dladdr(bt[i], &info) && info.dli_fbase) fprintf(stderr, " %lx", (unsigned long)((char*)bt[i] - (char*)info.dli_fbase));
    fprintf(stderr, "\n");
}
void Context::sync()
{
    ++n_stream_sync;
    if (tune("SYNC_TRACE", 0) != 0) {
        void* bt[6];
        const int n = backtrace(bt, 6);
        Dl_info info;
        fprintf(stderr, "iamrx sync:");
        for (int i = 1; i < n; ++i) {
            if (dladdr(bt[i], &info) && info.dli_fbase) fprintf(stderr, " %lx", (unsigned long)((char*)bt[i] - (char*)info.dli_fbase));
        }
        fprintf(stderr, "\n");
    }
    if (stream) IAMRX_HIP_CHECK(hipStreamSynchronize(stream));
}

// small host->device uploads (descriptor tables, parameter blocks) without synchronising the stream: the source
// is copied into a pinned ring buffer first, so the caller's memory may die immediately.  A slot is only reused
// after RING bytes of later uploads; a stream sync is inserted on wrap-around to make that safe.
void Context::upload_async(void* dst, const void* src, size_t bytes)
{
    constexpr size_t RING = 8u << 20;
    if (bytes > RING / 4) {
        IAMRX_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, stream));
        sync();
        return;
    }
    if (!h_ring) IAMRX_HIP_CHECK(hipHostMalloc((void**)&h_ring, RING));
    const size_t need = (bytes + 63) & ~size_t(63);
    if (ring_off + need > RING) { sync(); ring_off = 0; }
    std::memcpy(h_ring + ring_off, src, bytes);
    IAMRX_HIP_CHECK(hipMemcpyAsync(dst, h_ring + ring_off, bytes, hipMemcpyHostToDevice, stream));
    ring_off += need;
}

// ------------------------------------------------------------------ Layout
static std::atomic<uint64_t> g_layout_id{1};

Layout::Layout(const std::vector<BoxD>& b, const std::vector<int>& own, int myrank) : boxes(b), owner(own)
{
    IAMRX_ASSERT(b.size() == own.size());
    id = g_layout_id++;
    local_of.assign(b.size(), -1);
    for (size_t g = 0; g < b.size(); ++g)
        if (own[g] == myrank) { local_of[g] = (int)local.size(); local.push_back((int)g); }
    std::vector<BoxD> lb;
    for (int g : local) {
        lb.push_back(boxes[g]);
        for (int d = 0; d < 3; ++d) {
            max_len[d] = std::max(max_len[d], boxes[g].len(d));
            if (boxes[g].lo[d] < 0 || (boxes[g].lo[d] & 1)) all_lo_even[d] = false;
        }
    }
    if (!lb.empty()) {
        auto& ctx = Context::get();
        d_boxes = (BoxD*)ctx.alloc(lb.size() * sizeof(BoxD));
        IAMRX_HIP_CHECK(hipMemcpyAsync(d_boxes, lb.data(), lb.size() * sizeof(BoxD), hipMemcpyHostToDevice, ctx.stream));
        ctx.sync();
    }
}

void evict_layout_tables(uint64_t lid);
// the registry and the caches it serves are never destroyed (leaked on purpose): a layout may die during static destruction
static std::vector<std::function<void(uint64_t)>>& layout_evictors() { static auto* v = new std::vector<std::function<void(uint64_t)>>(); return *v; }
void register_layout_evictor(std::function<void(uint64_t)> f) { layout_evictors().push_back(std::move(f)); }
void evict_layout_caches(uint64_t lid)
{
    evict_layout_tables(lid);
    auto& ev = layout_evictors();
    for (size_t q = 0; q < ev.size(); ++q) ev[q](lid);       // by index: an evictor may destroy a derived layout, which re-enters here
}
Layout::~Layout() { evict_layout_caches(id); if (d_boxes) Context::get().free(d_boxes); }

long Layout::local_cells() const { long n = 0; for (int g : local) n += boxes[g].npts(); return n; }
long Layout::total_cells() const { long n = 0; for (auto& b : boxes) n += b.npts(); return n; }

bool Layout::coarsenable(int ratio, int min_width) const
{
    for (auto& b : boxes)
        for (int d = 0; d < 3; ++d) {
            if (b.len(d) % ratio != 0 || b.len(d) / ratio < min_width) return false;
            int lo = b.lo[d];
            if (((lo % ratio) + ratio) % ratio != 0) return false;
        }
    return true;
}

std::shared_ptr<Layout> Layout::coarsened(int ratio) const
{
    if (m_coarse && m_coarse_ratio == ratio) return m_coarse;
    std::vector<BoxD> cb;
    for (auto& b : boxes) cb.push_back(coarsen(b, ratio));
    m_coarse = std::make_shared<Layout>(cb, owner, Context::get().comm->rank);
    m_coarse->replicated = replicated;
    m_coarse_ratio = ratio;
    return m_coarse;
}

bool Layout::slab_coarsenable(int min_width) const
{
    for (auto& b : boxes) {
        if (b.lo[1] != 0 || b.hi[1] != 1) return false;
        for (int d = 0; d < 3; d += 2) {
            if (b.len(d) % 2 != 0 || b.len(d) / 2 < min_width) return false;
            if (((b.lo[d] % 2) + 2) % 2 != 0) return false;
        }
    }
    return !boxes.empty();
}

std::shared_ptr<Layout> Layout::slab_coarsened() const
{
    if (m_slab) return m_slab;
    std::vector<BoxD> cb;
    for (auto& b : boxes) { BoxD c = coarsen(b, 2); c.lo[1] = 0; c.hi[1] = 1; cb.push_back(c); }
    m_slab = std::make_shared<Layout>(cb, owner, Context::get().comm->rank);
    m_slab->replicated = replicated;
    return m_slab;
}

// boxes that share full faces and have the same owner, merged (sweeps along x, y, z until nothing changes); the result in a deterministic
// order: by owner, then z, y, x of the lower corner.  Returns whether anything merged.
static bool merge_boxes(std::vector<BoxD>& boxes, std::vector<int>& owner)
{
    struct OB { BoxD b; int own; };
    std::vector<OB> v;
    for (size_t i = 0; i < boxes.size(); ++i) v.push_back({boxes[i], owner[i]});
    bool merged_any = false, changed = true;
    while (changed) {
        changed = false;
        for (int d = 0; d < 3; ++d) {
            const int e = (d + 1) % 3, f = (d + 2) % 3;
            std::sort(v.begin(), v.end(), [&](const OB& a, const OB& b) {
                if (a.own != b.own) return a.own < b.own;
                if (a.b.lo[e] != b.b.lo[e]) return a.b.lo[e] < b.b.lo[e];
                if (a.b.hi[e] != b.b.hi[e]) return a.b.hi[e] < b.b.hi[e];
                if (a.b.lo[f] != b.b.lo[f]) return a.b.lo[f] < b.b.lo[f];
                if (a.b.hi[f] != b.b.hi[f]) return a.b.hi[f] < b.b.hi[f];
                return a.b.lo[d] < b.b.lo[d];
            });
            std::vector<OB> w;
            for (const OB& x : v) {
                if (!w.empty()) {
                    OB& y = w.back();
                    if (y.own == x.own && y.b.lo[e] == x.b.lo[e] && y.b.hi[e] == x.b.hi[e] && y.b.lo[f] == x.b.lo[f] && y.b.hi[f] == x.b.hi[f] &&
                        y.b.hi[d] + 1 == x.b.lo[d]) { y.b.hi[d] = x.b.hi[d]; changed = true; merged_any = true; continue; }
                }
                w.push_back(x);
            }
            v.swap(w);
        }
    }
    if (!merged_any) return false;
    std::sort(v.begin(), v.end(), [](const OB& a, const OB& b) {
        if (a.own != b.own) return a.own < b.own;
        if (a.b.lo[2] != b.b.lo[2]) return a.b.lo[2] < b.b.lo[2];
        if (a.b.lo[1] != b.b.lo[1]) return a.b.lo[1] < b.b.lo[1];
        return a.b.lo[0] < b.b.lo[0];
    });
    boxes.clear(); owner.clear();
    for (const OB& x : v) { boxes.push_back(x.b); owner.push_back(x.own); }
    return true;
}

// Every rank holds the whole level.  IAMRX_MG_AGG_MERGE (1): as few boxes as tile it -- all of them belong to this rank, so boxes that share
// full faces merge (a level that covers its domain becomes ONE box spanning it: index wrap on periodic domains, wall formulas, the
// single-workgroup bottom solvers; no ghost fills between the boxes a coarse level would otherwise consist of).  The transfers between the
// distributed level and this one map every distributed box to the merged box that contains it (gather_plan, scatter_from_replicated).
std::shared_ptr<Layout> Layout::make_replicated() const
{
    if (m_repl) return m_repl;
    const int me = Context::get().comm->rank;
    std::vector<BoxD> bx = boxes;
    std::vector<int> own(boxes.size(), me);
    if (tune("MG_AGG_MERGE", 1) != 0) merge_boxes(bx, own);
    m_repl = std::make_shared<Layout>(bx, own, me);
    m_repl->replicated = true;
    m_repl->replicated_of = id;
    return m_repl;
}

// ------------------------------------------------------------------ MultiFab
MultiFab::MultiFab(LayoutP l, IndexType t, int nc, int ng) { define(std::move(l), t, nc, ng); }
MultiFab::~MultiFab() { release(); }

MultiFab::MultiFab(MultiFab&& o) noexcept { *this = std::move(o); }
MultiFab& MultiFab::operator=(MultiFab&& o) noexcept
{
    if (this != &o) {
        release();
        layout = std::move(o.layout); type = o.type; ncomp = o.ncomp; ngrow = o.ngrow;
        base = o.base; total_doubles = o.total_doubles; h_tab = std::move(o.h_tab); d_tab = o.d_tab; is_alias = o.is_alias;
        uniform_marked = o.uniform_marked; uniform_value = o.uniform_value;       // the mark describes the data: it travels with them
        o.base = nullptr; o.d_tab = nullptr; o.total_doubles = 0; o.is_alias = false; o.uniform_marked = false;
    }
    return *this;
}

// Device descriptor tables are a pure function of (layout, index type, ncomp, ngrow, base address).  The caching allocator hands the
// same blocks back step after step, so the tables are kept in a cache owned by the context instead of being re-uploaded for every
// temporary MultiFab (round 1: ~10 k small host-to-device copies per profiled run).  Entries die with their layout.
namespace {
struct TabKey {
    uint64_t lid; int t0, t1, t2, nc, ng; const double* base;
    bool operator<(const TabKey& o) const
    { return std::tie(lid, t0, t1, t2, nc, ng, base) < std::tie(o.lid, o.t0, o.t1, o.t2, o.nc, o.ng, o.base); }
};
std::map<TabKey, FabD*>& tab_cache() { static auto* c = new std::map<TabKey, FabD*>(); return *c; }   // leaked on purpose: layouts can die during static destruction
}  // namespace

void evict_layout_tables(uint64_t lid)
{
    auto& c = tab_cache();
    for (auto it = c.lower_bound(TabKey{lid, 0, 0, 0, 0, 0, nullptr}); it != c.end() && it->first.lid == lid;) {
        Context::get().free(it->second);
        it = c.erase(it);
    }
}

void MultiFab::release()
{
    auto& ctx = Context::get();
    if (is_alias) { if (d_tab) ctx.free(d_tab); }         // the data belong to the caller, the table to this object
    else if (base) ctx.free(base);
    base = nullptr; d_tab = nullptr; total_doubles = 0; h_tab.clear(); is_alias = false;      // d_tab of an owning MultiFab belongs to the table cache
    uniform_marked = false;             // (define / alias / view_of / clear / move-assignment all come through here: new data, no promise)
}

void MultiFab::alias(LayoutP l, IndexType t, int nc, int ng, double* const* fab_ptrs)
{
    release();
    layout = std::move(l); type = t; ncomp = nc; ngrow = ng;
    auto& ctx = Context::get();
    const int nl = layout->nlocal();
    h_tab.resize(nl);
    for (int li = 0; li < nl; ++li) {
        BoxD fb = fabbox(li);
        FabD& f = h_tab[li];
        for (int d = 0; d < 3; ++d) { f.lo[d] = fb.lo[d]; f.n[d] = fb.len(d); }
        f.cs = (long)f.n[0] * f.n[1] * f.n[2];
        f.p = fab_ptrs[li];
        if (!f.p) throw Error("iamrx MultiFab::alias: null fab pointer");
        total_doubles += (size_t)f.cs * nc;
    }
    if (nl == 0) return;
    is_alias = true;
    base = h_tab[0].p;
    FabD* d = (FabD*)ctx.alloc(nl * sizeof(FabD));
    ctx.upload_async(d, h_tab.data(), nl * sizeof(FabD));
    d_tab = d;
}

void MultiFab::view_of(const MultiFab& src, int comp, int nc)
{
    IAMRX_ASSERT(src.defined() && comp >= 0 && nc >= 1 && comp + nc <= src.ncomp);
    std::vector<double*> ptrs(src.h_tab.size());
    for (size_t li = 0; li < ptrs.size(); ++li) ptrs[li] = src.h_tab[li].p + src.h_tab[li].cs * comp;
    alias(src.layout, src.type, nc, src.ngrow, ptrs.data());
}

void MultiFab::clear() { release(); layout.reset(); }

void MultiFab::define(LayoutP l, IndexType t, int nc, int ng)
{
    release();
    layout = std::move(l); type = t; ncomp = nc; ngrow = ng;
    auto& ctx = Context::get();
    const int nl = layout->nlocal();
    h_tab.resize(nl);
    size_t off = 0;
    std::vector<size_t> offs(nl);
    for (int li = 0; li < nl; ++li) {
        BoxD fb = fabbox(li);
        FabD& f = h_tab[li];
        for (int d = 0; d < 3; ++d) { f.lo[d] = fb.lo[d]; f.n[d] = fb.len(d); }
        f.cs = (long)f.n[0] * f.n[1] * f.n[2];
        offs[li] = off;
        off += (size_t)f.cs * nc;
        off = (off + 31) & ~size_t(31);   // 256-byte alignment of every fab
    }
    total_doubles = off;
    if (nl == 0) return;
    base = (double*)ctx.alloc(total_doubles * sizeof(double));
    for (int li = 0; li < nl; ++li) h_tab[li].p = base + offs[li];
    const TabKey key{layout->id, type.t[0], type.t[1], type.t[2], nc, ng, base};
    auto& cache = tab_cache();
    auto it = cache.find(key);
    if (it == cache.end()) {
        FabD* d = (FabD*)ctx.alloc(nl * sizeof(FabD));
        ctx.upload_async(d, h_tab.data(), nl * sizeof(FabD));   // staged through the pinned ring: no stream sync
        it = cache.emplace(key, d).first;
    }
    d_tab = it->second;
}

void MultiFab::setVal(double v)
{
    trace_blas_site("setVal", layout ? layout->local_cells() * ncomp : 0);
    if (uniform_marked && v != uniform_value) uniform_marked = false;
    if (!base) return;
    if (is_alias) { setVal(v, 0, ncomp, ngrow); return; }      // the fabs of an alias are not one allocation
    launch_fill(base, total_doubles, v, Context::get().stream);
}

void MultiFab::setVal(double v, int comp, int nc, int ng)
{
    if (uniform_marked && v != uniform_value) uniform_marked = false;
    if (!base) return;
    const FabD* tab = d_tab;
    for_each(*layout, type, ng, Context::get().stream, [=] __device__(int i, int j, int k, int f) {
        const FabD a = tab[f];
        for (int n = 0; n < nc; ++n) a(i, j, k, comp + n) = v;
    });
}

void MultiFab::Copy(MultiFab& dst, const MultiFab& src, int scomp, int dcomp, int nc, int ng)
{
    trace_blas_site("Copy", dst.layout ? dst.layout->local_cells() * nc : 0);
    IAMRX_ASSERT(dst.layout->id == src.layout->id || dst.layout->boxes.size() == src.layout->boxes.size());
    IAMRX_ASSERT(ng <= dst.ngrow && ng <= src.ngrow);
    // a copy INTO a marked array ends the promise unless the source carries the same one
    if (dst.uniform_marked && !(src.uniform_marked && src.uniform_value == dst.uniform_value)) dst.uniform_marked = false;
    if (!dst.base) return;
    const FabD* dt = dst.d_tab;
    const FabD* st = src.d_tab;
    for_each_2ph(*dst.layout, dst.type, ng, nc, Context::get().stream,
        [=] __device__(int i, int j, int k, int f, int n) { return st[f](i, j, k, scomp + n); },
        [=] __device__(int i, int j, int k, int f, int n, double v) { dt[f](i, j, k, dcomp + n) = v; });
}

void MultiFab::copy_to_host(int li, double* dst) const
{
    auto& ctx = Context::get();
    ctx.sync();
    IAMRX_HIP_CHECK(hipMemcpy(dst, h_tab[li].p, (size_t)h_tab[li].cs * ncomp * sizeof(double), hipMemcpyDeviceToHost));
}

void MultiFab::copy_from_host(int li, const double* src)
{
    uniform_marked = false;
    auto& ctx = Context::get();
    ctx.sync();
    IAMRX_HIP_CHECK(hipMemcpy(h_tab[li].p, src, (size_t)h_tab[li].cs * ncomp * sizeof(double), hipMemcpyHostToDevice));
}

double MultiFab::norm0(int comp, int nc, int ng, bool local) const
{
    return reduce_norm0(*this, comp, nc, ng, !local);
}

void MultiFab::norm0_comps(int comp, int nc, int ng, double* out, bool local) const
{
    reduce_norm0_comps(*this, comp, nc, ng, out, !local);
}

double MultiFab::sum_unique(const Geometry& g, int comp, bool local) const
{
    return reduce_sum_unique(*this, comp, g, !local);
}

// ------------------------------------------------------------------ FillBoundary plan
CopyPlan::~CopyPlan()
{
    auto& ctx = Context::get();
    if (d_local) ctx.free(d_local);
    if (w_local.d) ctx.free(w_local.d);
    for (auto& p : peers) {
        if (p.d_pack) ctx.free(p.d_pack);
        if (p.d_unpack) ctx.free(p.d_unpack);
        if (p.w_pack.d) ctx.free(p.w_pack.d);
        if (p.w_unpack.d) ctx.free(p.w_unpack.d);
    }
}

struct PlanKey {
    uint64_t layout_id; IndexType t; int ng; int per[3]; int dlo[3], dhi[3]; int ngv[3]; int kpar; int wall_ext;
    bool operator<(const PlanKey& o) const { return std::memcmp(this, &o, sizeof(PlanKey)) < 0; }
};

static CopyDesc* upload(const std::vector<CopyDesc>& v)
{
    if (v.empty()) return nullptr;
    auto& ctx = Context::get();
    CopyDesc* d = (CopyDesc*)ctx.alloc(v.size() * sizeof(CopyDesc));
    IAMRX_HIP_CHECK(hipMemcpyAsync(d, v.data(), v.size() * sizeof(CopyDesc), hipMemcpyHostToDevice, ctx.stream));
    ctx.sync();
    return d;
}

// ------------------------------------------------------------------ flat tile lists (launch.h: level_tiling with allow_list)
// a device-resident list of int4 work items that belongs to a layout (flat tile lists, the ghost-shell list of k_cf_fill ...): built once
// per (layout, subkey) by `build`, freed with the layout
const int4* layout_int4_list(const Layout& l, const std::array<long, 5>& subkey, const std::function<void(std::vector<int4>&)>& build, int* total)
{
    struct Entry { int4* d; int n; };
    using Key = std::array<long, 6>;
    static std::map<Key, Entry>& cache = [] () -> std::map<Key, Entry>& {
        auto* c = new std::map<Key, Entry>();
        register_layout_evictor([c](uint64_t lid) {
            bool any = false;
            for (auto& kv : *c) any = any || (uint64_t)kv.first[0] == lid;
            if (!any) return;
            Context::get().sync();                     // (a launch that reads the list may still be in flight)
            for (auto it = c->begin(); it != c->end();) {
                if ((uint64_t)it->first[0] != lid) { ++it; continue; }
                if (it->second.d) Context::get().free(it->second.d);
                it = c->erase(it);
            }
        });
        return *c;
    }();
    const Key key = {(long)l.id, subkey[0], subkey[1], subkey[2], subkey[3], subkey[4]};
    auto it = cache.find(key);
    if (it == cache.end()) {
        std::vector<int4> h;
        build(h);
        Entry e{nullptr, (int)h.size()};
        if (!h.empty()) {
            auto& ctx = Context::get();
            // (stream-ordered like every use of the caching allocator's blocks: the block may have belonged to an array whose last
            // kernel is still in flight on the launch stream -- a copy on the null stream would overtake it)
            e.d = (int4*)ctx.alloc(h.size() * sizeof(int4));
            IAMRX_HIP_CHECK(hipMemcpyAsync(e.d, h.data(), h.size() * sizeof(int4), hipMemcpyHostToDevice, ctx.stream));
            ctx.sync();
        }
        it = cache.emplace(key, e).first;
    }
    *total = it->second.n;
    return it->second.d;
}

const int4* level_tile_list(const Layout& l, const IndexType& type, int ng, int tz, int* total, int xdiv)
{
    return layout_int4_list(l, {1, type.t[0] + 2 * type.t[1] + 4 * type.t[2], ng, tz, xdiv}, [&](std::vector<int4>& h) {
        for (int f = 0; f < l.nlocal(); ++f) {
            BoxD b = grow(convert(l.lbox(f), type.t), ng);
            if (xdiv > 1) b.hi[0] = b.lo[0] + (b.len(0) + xdiv - 1) / xdiv - 1;       // tile space of a kernel that owns xdiv cells per thread
            int bx = 64, bxs = 6;
            while (bx > 4 && bx / 2 >= b.len(0)) { bx /= 2; --bxs; }
            const int by = 256 / bx;
            for (int k0 = 0; k0 < b.len(2); k0 += tz)
                for (int j0 = 0; j0 < b.len(1); j0 += by)
                    for (int i0 = 0; i0 < b.len(0); i0 += bx) h.push_back(make_int4(f, i0, j0, k0 | (bxs << 26)));
        }
    }, total);
}

BoxBins::BoxBins(std::vector<BoxD> boxes) : b(std::move(boxes))
{
    const int nb = (int)b.size();
    for (int d = 0; d < 3; ++d) { bbox.lo[d] = nb ? INT_MAX : 0; bbox.hi[d] = nb ? INT_MIN : -1; }
    long len_sum[3] = {0, 0, 0};
    for (const BoxD& q : b) for (int d = 0; d < 3; ++d) { bbox.lo[d] = std::min(bbox.lo[d], q.lo[d]); bbox.hi[d] = std::max(bbox.hi[d], q.hi[d]); len_sum[d] += q.len(d); }
    for (int d = 0; d < 3; ++d) bsz[d] = nb > 0 ? std::max<int>(4, (int)(len_sum[d] / nb)) : 4;
    for (;;) {
        for (int d = 0; d < 3; ++d) nbin[d] = nb > 0 ? (bbox.hi[d] - bbox.lo[d]) / bsz[d] + 1 : 1;
        if ((long)nbin[0] * nbin[1] * nbin[2] <= 2000000L) break;
        for (int d = 0; d < 3; ++d) bsz[d] *= 2;
    }
    bins.resize((size_t)nbin[0] * nbin[1] * nbin[2]);
    for (int i = 0; i < nb; ++i) {
        int b0[3], b1[3];
        for (int d = 0; d < 3; ++d) { b0[d] = (b[i].lo[d] - bbox.lo[d]) / bsz[d]; b1[d] = (b[i].hi[d] - bbox.lo[d]) / bsz[d]; }
        for (int bz = b0[2]; bz <= b1[2]; ++bz) for (int by = b0[1]; by <= b1[1]; ++by) for (int bx = b0[0]; bx <= b1[0]; ++bx)
            bins[((size_t)bz * nbin[1] + by) * nbin[0] + bx].push_back(i);
    }
    stamp.assign(nb, -1);
}

void BoxBins::query(const BoxD& q, std::vector<int>& out) const
{
    out.clear();
    const BoxD qi = intersect(q, bbox);
    if (b.empty() || !qi.ok()) return;
    ++stamp_id;
    int b0[3], b1[3];
    for (int d = 0; d < 3; ++d) { b0[d] = (qi.lo[d] - bbox.lo[d]) / bsz[d]; b1[d] = (qi.hi[d] - bbox.lo[d]) / bsz[d]; }
    for (int bz = b0[2]; bz <= b1[2]; ++bz) for (int by = b0[1]; by <= b1[1]; ++by) for (int bx = b0[0]; bx <= b1[0]; ++bx)
        for (int i : bins[((size_t)bz * nbin[1] + by) * nbin[0] + bx]) {
            if (stamp[i] == stamp_id) continue;
            stamp[i] = stamp_id;
            if (intersect(q, b[i]).ok()) out.push_back(i);
        }
    std::sort(out.begin(), out.end());
}

// host-only construction of a ghost-exchange plan (no device access: unit-testable on CPU)
void build_fill_plan_host(const std::vector<BoxD>& boxes, const std::vector<int>& owner, const std::vector<int>& local_of, int me,
                          IndexType t, int ng, const Geometry& g, CopyPlan& plan, std::map<int, CopyPlan::Peer>& peers, const int* ngv, int kpar,
                          int wall_ext)
{
    // wall_ext > 0 (cell-centred data): a SOURCE box that touches a non-periodic side of the domain also supplies its first wall_ext
    // ghost cells beyond that side (the values its owner put there: a boundary fill), so that the edge ghost cells of its neighbours
    // beyond the wall -- (wall + 1, lo - 1) seen from the box next door -- hold exactly what the source box itself reads there.
    auto src_valid = [&](int gs) {
        BoxD sv = convert(boxes[gs], t.t);
        if (wall_ext > 0)
            for (int d = 0; d < 3; ++d) {
                if (g.periodic[d]) continue;
                if (boxes[gs].lo[d] == g.domain.lo[d]) sv.lo[d] -= wall_ext;
                if (boxes[gs].hi[d] == g.domain.hi[d]) sv.hi[d] += wall_ext;
            }
        return sv;
    };
    // ngv: ghost depth to fill per direction (<= ng; nullptr: ng everywhere).  A consumer whose stencil reaches less far in one
    // direction (the plane-fused nodal smoother: 4 nodes in-plane, 1 plane in z) exchanges correspondingly thinner slabs.
    const int gv[3] = {ngv ? ngv[0] : ng, ngv ? ngv[1] : ng, ngv ? ngv[2] : ng};
    // periodic shift candidates
    int smin[3], smax[3];
    for (int d = 0; d < 3; ++d) {
        // ghost regions wider than the domain (coarse MG levels with 4 ghost layers) need several periods
        const int nper = g.periodic[d] ? std::max(1, (ng + g.domain.len(d) - 1) / g.domain.len(d)) : 0;
        smin[d] = -nper; smax[d] = nper;
    }
    const int nb = (int)boxes.size();
    // box difference a \ b appended to out
    auto subtract = [](const BoxD& a, const BoxD& b, std::vector<BoxD>& out) {
        BoxD in = intersect(a, b);
        if (!in.ok()) { out.push_back(a); return; }
        BoxD rem = a;
        for (int d = 2; d >= 0; --d) {
            if (rem.lo[d] < in.lo[d]) { BoxD p = rem; p.hi[d] = in.lo[d] - 1; out.push_back(p); rem.lo[d] = in.lo[d]; }
            if (rem.hi[d] > in.hi[d]) { BoxD p = rem; p.lo[d] = in.hi[d] + 1; out.push_back(p); rem.hi[d] = in.hi[d]; }
        }
    };
    const bool nodal = t.t[0] || t.t[1] || t.t[2];
    // Candidate sources of a box through a uniform bin index over the source boxes (round 6): the loops below used to visit every
    // (box, box, shift) triple -- 27 n^2 intersections per plan, 150 ms per plan on a refined level of 431 boxes, a fifth of the run time
    // of BASELINE config C5.  The candidates of a destination box = the (source, shift) pairs whose shifted source box meets its grown
    // box, in the order of the old loops (source ascending, then z, y, x shift): the plan is the same plan.
    std::vector<BoxD> sval(nb);
    for (int gs = 0; gs < nb; ++gs) sval[gs] = src_valid(gs);
    const BoxBins index(sval);
    struct Cand { int gs, sx, sy, sz; };
    std::vector<Cand> cand;
    std::vector<int> hits;
    auto candidates = [&](const BoxD& dgrown) {
        cand.clear();
        for (int sz = smin[2]; sz <= smax[2]; ++sz)
        for (int sy = smin[1]; sy <= smax[1]; ++sy)
        for (int sx = smin[0]; sx <= smax[0]; ++sx) {
            const int sh[3] = {sx * g.domain.len(0), sy * g.domain.len(1), sz * g.domain.len(2)};
            BoxD q = dgrown;                                  // the grown box in the frame of the unshifted sources
            for (int d = 0; d < 3; ++d) { q.lo[d] -= sh[d]; q.hi[d] -= sh[d]; }
            index.query(q, hits);
            for (int gs : hits) cand.push_back(Cand{gs, sx, sy, sz});
        }
        std::sort(cand.begin(), cand.end(), [](const Cand& a, const Cand& b) {
            if (a.gs != b.gs) return a.gs < b.gs;
            if (a.sz != b.sz) return a.sz < b.sz;
            if (a.sy != b.sy) return a.sy < b.sy;
            return a.sx < b.sx;
        });
    };
    for (int gd = 0; gd < nb; ++gd) {
        const bool dst_mine = owner[gd] == me;
        const BoxD dvalid = convert(boxes[gd], t.t);
        BoxD dgrown = dvalid;
        for (int d = 0; d < 3; ++d) { dgrown.lo[d] -= gv[d]; dgrown.hi[d] += gv[d]; }
        candidates(dgrown);
        if (!dst_mine) {             // a box of another rank: does any box of mine reach its ghost region at all?
            bool involved = false;
            for (const Cand& c : cand) if (owner[c.gs] == me) { involved = true; break; }
            if (!involved) continue;
        }
        // Nodal / face data: boxes share the points on their common faces, so a ghost point can have several sources.  Whatever a
        // box of the destination's own rank supplies is not requested from another rank as well (both sides of a message evaluate
        // this rule on the same global box list, so the plans stay symmetric).
        std::vector<BoxD> own_cov;
        if (nodal) {
            for (const Cand& c : cand) {
                const int gs = c.gs, sx = c.sx, sy = c.sy, sz = c.sz;
                if (owner[gs] != owner[gd]) continue;
                if (gs == gd && sx == 0 && sy == 0 && sz == 0) continue;
                BoxD svalid = sval[gs];
                const int sh[3] = {sx * g.domain.len(0), sy * g.domain.len(1), sz * g.domain.len(2)};
                for (int d = 0; d < 3; ++d) svalid = shift(svalid, d, sh[d]);
                BoxD is = intersect(dgrown, svalid);
                if (is.ok()) own_cov.push_back(is);
            }
        }
        // Nodal / face data again: a ghost point of this box can also have several sources ON its own rank -- the two boxes that share the
        // point, or (a periodic direction narrower than the ghost width: the two-cell slab levels) the images of one box under one and two
        // periods.  Both copies land in one launch; if the sources differ in the last bit (the duplicates of a periodic node that two
        // threads computed) the result depends on which write comes last.  Every ghost point therefore takes ONE source: the first
        // in the order of the loops below (regions already planned for this box are cut out of the later ones).  The rule covers remote
        // sources as well (two regions of one peer, or of two peers, holding the same ghost node: one unpack launch per peer writes
        // them): it is evaluated over ALL sources of the box, in the global order, by every rank that owns the box or one of its sources,
        // so the sender's pack list and the receiver's unpack list stay the same list.
        std::vector<BoxD> planned;
        for (const Cand& c : cand) {
            {
                const int gs = c.gs, sx = c.sx, sy = c.sy, sz = c.sz;
                const bool src_mine = owner[gs] == me;
                if (!nodal && !dst_mine && !src_mine) continue;
                const bool remote_src = owner[gs] != owner[gd];
                if (gs == gd && sx == 0 && sy == 0 && sz == 0) continue;
                // source valid box translated INTO the destination's index frame
                int sh[3] = {sx * g.domain.len(0), sy * g.domain.len(1), sz * g.domain.len(2)};
                BoxD svalid = sval[gs];
                for (int d = 0; d < 3; ++d) svalid = shift(svalid, d, sh[d]);
                BoxD is = intersect(dgrown, svalid);
                if (!is.ok()) continue;
                // drop the part inside the destination's own valid region: split `is` minus dvalid
                // into up to 6 boxes (box difference)
                std::vector<BoxD> parts;
                BoxD rem = is;
                BoxD in = intersect(is, dvalid);
                if (!in.ok()) parts.push_back(is);
                else {
                    for (int d = 2; d >= 0; --d) {
                        if (rem.lo[d] < in.lo[d]) { BoxD p = rem; p.hi[d] = in.lo[d] - 1; parts.push_back(p); rem.lo[d] = in.lo[d]; }
                        if (rem.hi[d] > in.hi[d]) { BoxD p = rem; p.lo[d] = in.hi[d] + 1; parts.push_back(p); rem.hi[d] = in.hi[d]; }
                    }
                }
                if (remote_src && !own_cov.empty()) {
                    for (const BoxD& c : own_cov) {
                        std::vector<BoxD> next;
                        for (const BoxD& q : parts) subtract(q, c, next);
                        parts.swap(next);
                        if (parts.empty()) break;
                    }
                }
                if (nodal) {
                    for (const BoxD& c : planned) {
                        std::vector<BoxD> next;
                        for (const BoxD& q : parts) subtract(q, c, next);
                        parts.swap(next);
                        if (parts.empty()) break;
                    }
                    for (const BoxD& q : parts) planned.push_back(q);
                    if (!dst_mine && !src_mine) continue;      // (somebody else's pair: book-keeping only)
                }
                for (auto& p0 : parts) {
                    BoxD p = p0;
                    CopyDesc cd;
                    if (kpar >= 0) {
                        // keep the z-planes of parity kpar only
                        int k0 = p.lo[2];
                        if ((((k0 % 2) + 2) % 2) != kpar) ++k0;
                        if (k0 > p.hi[2]) continue;
                        p.lo[2] = k0;
                        p.hi[2] = k0 + 2 * ((p.hi[2] - k0) / 2);
                        cd.kstep = 2;
                    }
                    cd.region = p;
                    const long np = cd.npts();
                    for (int d = 0; d < 3; ++d) cd.shift[d] = -sh[d];
                    cd.buf_off = 0;
                    if (dst_mine && src_mine) {
                        cd.src_fab = local_of[gs]; cd.dst_fab = local_of[gd];
                        plan.local.push_back(cd);
                        plan.max_local_pts = std::max(plan.max_local_pts, np);
                    } else if (src_mine) {          // I send to owner of gd
                        auto& pr = peers[owner[gd]];
                        pr.rank = owner[gd];
                        cd.src_fab = local_of[gs]; cd.dst_fab = -1; cd.buf_off = pr.send_pts;
                        pr.send_pts += np;
                        pr.max_pack_pts = std::max(pr.max_pack_pts, np);
                        pr.pack.push_back(cd);
                    } else {                        // I receive from owner of gs
                        auto& pr = peers[owner[gs]];
                        pr.rank = owner[gs];
                        cd.src_fab = -1; cd.dst_fab = local_of[gd]; cd.buf_off = pr.recv_pts;
                        pr.recv_pts += np;
                        pr.max_unpack_pts = std::max(pr.max_unpack_pts, np);
                        pr.unpack.push_back(cd);
                    }
                }
            }
        }
    }
}


// plan cache keyed by PlanKey::layout_id: created on first use, never destroyed, entries leave with their layout
static std::map<PlanKey, std::unique_ptr<CopyPlan>>& make_plan_cache()
{
    auto* c = new std::map<PlanKey, std::unique_ptr<CopyPlan>>();
    register_layout_evictor([c](uint64_t lid) {
        std::vector<std::unique_ptr<CopyPlan>> dead;
        for (auto it = c->begin(); it != c->end();) { if (it->first.layout_id == lid) { dead.push_back(std::move(it->second)); it = c->erase(it); } else ++it; }
    });
    return *c;
}

const CopyPlan& fill_boundary_plan(const Layout& l, IndexType t, int ng, const Geometry& g, const int* ngv, int kpar, int wall_ext)
{
    static std::map<PlanKey, std::unique_ptr<CopyPlan>>& cache = make_plan_cache();
    PlanKey key;
    std::memset(&key, 0, sizeof(key));
    key.layout_id = l.id; key.t = t; key.ng = ng;
    for (int d = 0; d < 3; ++d) key.ngv[d] = ngv ? ngv[d] : ng;
    key.kpar = kpar; key.wall_ext = wall_ext;
    for (int d = 0; d < 3; ++d) { key.per[d] = g.periodic[d]; key.dlo[d] = g.domain.lo[d]; key.dhi[d] = g.domain.hi[d]; }
    auto it = cache.find(key);
    if (it != cache.end()) return *it->second;

    ProfScope ps_prof_("fill_plan_build");
    auto plan = std::make_unique<CopyPlan>();
    std::map<int, CopyPlan::Peer> peers;
    build_fill_plan_host(l.boxes, l.owner, l.local_of, Context::get().comm->rank, t, ng, g, *plan, peers, ngv, kpar, wall_ext);
    plan->d_local = upload(plan->local);
    for (auto& kv : peers) {
        kv.second.d_pack = upload(kv.second.pack);
        kv.second.d_unpack = upload(kv.second.unpack);
        plan->peers.push_back(std::move(kv.second));
        kv.second.d_pack = nullptr; kv.second.d_unpack = nullptr;
    }
    auto& ref = *plan;
    cache.emplace(key, std::move(plan));
    return ref;
}

// flat work list of a descriptor list: used where the descriptors are many and of different sizes (the plain launch gives every descriptor
// the workgroups of the largest); built and uploaded on first use
static bool copy_work(const std::vector<CopyDesc>& descs, long maxpts, CopyWork& w)
{
    if (!w.built) {
        w.built = true;
        if (descs.size() >= 64 && tune("COPY_WORK_LISTS", 1) != 0) {
            long plain_blocks = std::min<long>((maxpts + 255) / 256, 256) * (long)descs.size(), n = 0;
            for (const CopyDesc& cd : descs) n += (cd.npts() + COPY_CHUNK - 1) / COPY_CHUNK;
            if (4 * n <= 3 * plain_blocks && n < (1L << 30)) {
                std::vector<int2> h;
                h.reserve((size_t)n);
                for (size_t q = 0; q < descs.size(); ++q) {
                    const long nch = (descs[q].npts() + COPY_CHUNK - 1) / COPY_CHUNK;
                    for (long c = 0; c < nch; ++c) h.push_back(make_int2((int)q, (int)c));
                }
                auto& ctx = Context::get();
                // (ordered on the launch stream like every use of the caching allocator's blocks, then waited for: the first use of a plan may
                // come from the side stream)
                w.d = (int2*)ctx.alloc(h.size() * sizeof(int2));
                IAMRX_HIP_CHECK(hipMemcpyAsync(w.d, h.data(), h.size() * sizeof(int2), hipMemcpyHostToDevice, ctx.stream));
                ctx.sync();
                w.n = (int)h.size();
            }
        }
    }
    return w.d != nullptr;
}

void execute_plan(const CopyPlan& plan, MultiFab& dst, const MultiFab& src, int scomp, int dcomp, int nc, bool add, hipStream_t on)
{
    auto& ctx = Context::get();
    hipStream_t s = on ? on : ctx.stream;
    // remote: pack -> exchange -> unpack
    std::vector<Message> sends, recvs;
    std::vector<double*> bufs;
    for (auto& p : plan.peers) {
        if (p.send_pts > 0) {
            double* sb = (double*)ctx.alloc((size_t)p.send_pts * nc * sizeof(double));
            bufs.push_back(sb);
            if (copy_work(p.pack, p.max_pack_pts, p.w_pack)) launch_pack_w(p.d_pack, p.w_pack.d, p.w_pack.n, src.d_tab, sb, p.send_pts, scomp, nc, s);
            else launch_pack(p.d_pack, (int)p.pack.size(), p.max_pack_pts, src.d_tab, sb, p.send_pts, scomp, nc, s);
            sends.push_back({p.rank, sb, (size_t)p.send_pts * nc});
        }
        if (p.recv_pts > 0) {
            double* rb = (double*)ctx.alloc((size_t)p.recv_pts * nc * sizeof(double));
            bufs.push_back(rb);
            recvs.push_back({p.rank, rb, (size_t)p.recv_pts * nc});
        }
    }
    if (!plan.local.empty()) {
        if (copy_work(plan.local, plan.max_local_pts, plan.w_local)) launch_copy_plan_w(plan.d_local, plan.w_local.d, plan.w_local.n, src.d_tab, dst.d_tab, scomp, dcomp, nc, s, add);
        else launch_copy_plan(plan.d_local, (int)plan.local.size(), plan.max_local_pts, src.d_tab, dst.d_tab, scomp, dcomp, nc, s, add);
    }
    if (!sends.empty() || !recvs.empty()) {
        const int cls = s == ctx.stream ? 0 : 1;
        ++ctx.n_exchange[cls];
        for (auto& m : sends) ctx.exchange_doubles[cls] += m.count;
        ctx.comm->exchange(sends, recvs, s);
        size_t ri = 0;
        for (auto& p : plan.peers) {
            if (p.recv_pts > 0) {
                if (copy_work(p.unpack, p.max_unpack_pts, p.w_unpack)) launch_unpack_w(p.d_unpack, p.w_unpack.d, p.w_unpack.n, dst.d_tab, recvs[ri].dev_ptr, p.recv_pts, dcomp, nc, s, add);
                else launch_unpack(p.d_unpack, (int)p.unpack.size(), p.max_unpack_pts, dst.d_tab, recvs[ri].dev_ptr, p.recv_pts, dcomp, nc, s, add);
                ++ri;
            }
        }
    }
    // buffers are stream-ordered: safe to return them to the cache (next user is on the same stream; side stream: parked until the join)
    for (double* b : bufs) ctx.free_on(s, b);
}

// ------------------------------------------------------------------ agglomeration transfers
// dist (boxes spread over the ranks) -> repl (same boxes, all of them on every rank): an all-gather of the valid regions,
// expressed as a CopyPlan so that it runs through execute_plan (pack kernel, one message per peer, unpack kernel).
// index of the box of `repl` (a replicated layout: all boxes local, possibly merged) that contains box b
static int containing_box(const Layout& repl, const BoxD& b)
{
    for (int m = 0; m < (int)repl.boxes.size(); ++m) {
        const BoxD& r = repl.boxes[m];
        if (b.lo[0] >= r.lo[0] && b.hi[0] <= r.hi[0] && b.lo[1] >= r.lo[1] && b.hi[1] <= r.hi[1] && b.lo[2] >= r.lo[2] && b.hi[2] <= r.hi[2]) return m;
    }
    throw Error("iamrx: a replicated layout does not contain a box of the level it was made from");
}

static const CopyPlan& gather_plan(const Layout& dist, IndexType t)
{
    static std::map<PlanKey, std::unique_ptr<CopyPlan>>& cache = make_plan_cache();
    PlanKey key;
    std::memset(&key, 0, sizeof(key));
    key.layout_id = dist.id; key.t = t; key.ng = -1;
    auto it = cache.find(key);
    if (it != cache.end()) return *it->second;
    auto plan = std::make_unique<CopyPlan>();
    const int me = Context::get().comm->rank, nr = Context::get().comm->nranks;
    const Layout& repl = *dist.make_replicated();
    std::map<int, CopyPlan::Peer> peers;
    for (int g = 0; g < (int)dist.boxes.size(); ++g) {
        CopyDesc cd;
        cd.region = convert(dist.boxes[g], t.t);
        cd.shift[0] = cd.shift[1] = cd.shift[2] = 0;
        cd.buf_off = 0;
        const long np = cd.region.npts();
        const int rg = containing_box(repl, dist.boxes[g]);         // (nodal / face data: boxes inside one merged box write the points they share twice -- the same values)
        if (dist.owner[g] == me) {
            cd.src_fab = dist.local_of[g]; cd.dst_fab = rg;
            plan->local.push_back(cd);
            plan->max_local_pts = std::max(plan->max_local_pts, np);
            for (int r = 0; r < nr; ++r) {
                if (r == me) continue;
                auto& pr = peers[r];
                pr.rank = r;
                CopyDesc pd = cd;
                pd.dst_fab = -1; pd.buf_off = pr.send_pts;
                pr.send_pts += np; pr.max_pack_pts = std::max(pr.max_pack_pts, np);
                pr.pack.push_back(pd);
            }
        } else {
            auto& pr = peers[dist.owner[g]];
            pr.rank = dist.owner[g];
            cd.src_fab = -1; cd.dst_fab = rg; cd.buf_off = pr.recv_pts;
            pr.recv_pts += np; pr.max_unpack_pts = std::max(pr.max_unpack_pts, np);
            pr.unpack.push_back(cd);
        }
    }
    plan->d_local = upload(plan->local);
    for (auto& kv : peers) {
        kv.second.d_pack = upload(kv.second.pack);
        kv.second.d_unpack = upload(kv.second.unpack);
        plan->peers.push_back(std::move(kv.second));
        kv.second.d_pack = nullptr; kv.second.d_unpack = nullptr;
    }
    auto& ref = *plan;
    cache.emplace(key, std::move(plan));
    return ref;
}

void gather_to_replicated(MultiFab& repl, const MultiFab& dist)
{
    IAMRX_ASSERT(repl.layout->replicated && repl.layout->replicated_of == dist.layout->id && repl.ncomp == dist.ncomp);
    execute_plan(gather_plan(*dist.layout, dist.type), repl, dist, 0, 0, dist.ncomp);
}

// repl -> dist: every rank picks its own boxes (valid region grown by ng, ng <= both ghost widths); purely local
void scatter_from_replicated(MultiFab& dist, const MultiFab& repl, int ng)
{
    IAMRX_ASSERT(repl.layout->replicated && repl.layout->replicated_of == dist.layout->id && repl.ncomp == dist.ncomp);
    IAMRX_ASSERT(ng <= dist.ngrow && ng <= repl.ngrow);
    static std::map<PlanKey, std::unique_ptr<CopyPlan>>& cache = make_plan_cache();
    PlanKey key;
    std::memset(&key, 0, sizeof(key));
    key.layout_id = dist.layout->id; key.t = dist.type; key.ng = ng;
    auto it = cache.find(key);
    if (it == cache.end()) {
        auto plan = std::make_unique<CopyPlan>();
        const Layout& l = *dist.layout;
        for (int li = 0; li < l.nlocal(); ++li) {
            CopyDesc cd;
            cd.region = grow(convert(l.lbox(li), dist.type.t), ng);
            cd.shift[0] = cd.shift[1] = cd.shift[2] = 0;
            cd.buf_off = 0;
            cd.src_fab = containing_box(*repl.layout, l.lbox(li)); cd.dst_fab = li;
            plan->local.push_back(cd);
            plan->max_local_pts = std::max(plan->max_local_pts, cd.region.npts());
        }
        plan->d_local = upload(plan->local);
        it = cache.emplace(key, std::move(plan)).first;
    }
    execute_plan(*it->second, dist, repl, 0, 0, dist.ncomp);
}

void MultiFab::FillBoundary(const Geometry& g) { FillBoundary(g, 0, ncomp); }

void MultiFab::FillBoundary(const Geometry& g, int comp, int nc, const int* ngv, int kpar, hipStream_t on)
{
    if (ngrow == 0) return;
    const CopyPlan& plan = fill_boundary_plan(*layout, type, ngrow, g, ngv, kpar);
    execute_plan(plan, *this, *this, comp, comp, nc, false, on);
}

void MultiFab::FillBoundaryWallExt(const Geometry& g, int ext)
{
    if (ngrow == 0) return;
    if (!type.cell() || ext > ngrow) throw Error("iamrx: FillBoundaryWallExt needs cell-centred data with at least `ext` ghost layers");
    const CopyPlan& plan = fill_boundary_plan(*layout, type, ngrow, g, nullptr, -1, ext);
    execute_plan(plan, *this, *this, 0, 0, ncomp, false, nullptr);
}

// ------------------------------------------------------------------ coalescing
// sweeps along x, y, z until nothing merges: two boxes of one owner with equal extents in the two other directions and touching faces
static size_t g_coalesce_merges = 0;
size_t coalesce_merge_count() { return g_coalesce_merges; }

LayoutP coalesce_layout(const LayoutP& l)
{
    if (!l || l->replicated || tune("COALESCE", 1) == 0 || l->boxes.size() < 2) return l;
    std::vector<BoxD> nb = l->boxes;
    std::vector<int> no = l->owner;
    if (!merge_boxes(nb, no)) return l;
    ++g_coalesce_merges;
    return std::make_shared<Layout>(nb, no, Context::get().comm->rank);
}

}  // namespace iamrx
