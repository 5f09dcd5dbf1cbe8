// rccl_stub.cpp - TEST INFRASTRUCTURE ONLY.  An in-process stand-in for librccl with the ten entry points
// fastp_amd/csrc/fq_comm.cpp resolves (FASTP_GPU_RCCL_LIB points at it), so that the C ABI's collectives -
// argument marshalling, header-word preservation, the send/recv schedule of the bitmap exchange - run with
// n = 2 contexts in ONE process on the emulator build (device memory = host memory).  Collectives are recorded
// between ncclGroupStart and ncclGroupEnd and carried out at the end of the group, matching the operations of
// the communicators that share a unique id; outside a group an operation is carried out at once, which needs all
// ranks of the communicator to be in the same call (how fq_comm.cpp uses n > 1: one group per exchange).
// The ranks may also be THREADS of the process, one context each (how bench.py's ranks call the C ABI, one rank per
// process there): a group that ends while ranks of its communicators are still missing waits for the other threads'
// groups (FASTP_STUB_TIMEOUT_MS, default 30000 ms - a rank that never shows up is an error, as before).
#include <stdint.h>
#include <string.h>

#include <stdlib.h>

#include <chrono>
#include <condition_variable>
#include <map>
#include <mutex>
#include <set>
#include <vector>

#include "../hostsim/rccl/rccl.h"

namespace {
struct Comm { int id, nranks, rank; };
struct Op { int kind; const void* src; void* dst; size_t count; int dtype; int peer; Comm* c; };   // 0 allreduce, 1 send, 2 recv
std::vector<Op> g_ops;                  // the operations of every thread's finished (and not yet carried out) group
thread_local std::vector<Op> t_ops;     // the calling thread's open group
thread_local int g_depth = 0;
int g_next_id = 1;
std::mutex g_mu;
std::condition_variable g_cv;
unsigned long g_generation = 0;         // bumped whenever a flush has carried the pending operations out
ncclResult_t g_last = ncclSuccess;
size_t width(int dt) { return dt == ncclInt64 || dt == ncclUint64 ? 8 : (dt == ncclInt32 || dt == ncclUint32 ? 4 : 1); }

ncclResult_t flush_ops();
bool all_ranks_present() {   // every communicator that has an operation pending has one from each of its ranks
    std::map<int, std::set<int>> seen;
    std::map<int, int> want;
    for (const Op& o : g_ops) { seen[o.c->id].insert(o.c->rank); want[o.c->id] = o.c->nranks; }
    for (auto& kv : seen) if ((int)kv.second.size() != want[kv.first]) return false;
    return true;
}
ncclResult_t flush() {   // a failed group leaves nothing queued behind
    std::unique_lock<std::mutex> lk(g_mu);
    g_ops.insert(g_ops.end(), t_ops.begin(), t_ops.end());
    t_ops.clear();
    if (!all_ranks_present()) {   // the other ranks are other threads: wait for their groups
        const unsigned long gen = g_generation;
        const char* v = getenv("FASTP_STUB_TIMEOUT_MS");
        const auto limit = std::chrono::milliseconds(v ? atoi(v) : 30000);
        if (g_cv.wait_for(lk, limit, [&] { return g_generation != gen; })) return g_last;
        g_ops.clear();            // nobody came: the group fails
        return ncclInvalidArgument;
    }
    g_last = flush_ops();
    g_ops.clear();
    g_generation++;
    g_cv.notify_all();
    return g_last;
}
ncclResult_t flush_ops() {
    // all-reduce: group by communicator id
    std::map<int, std::vector<Op*>> ar;
    for (Op& o : g_ops) if (o.kind == 0) ar[o.c->id].push_back(&o);
    for (auto& kv : ar) {
        std::vector<Op*>& v = kv.second;
        if ((int)v.size() != v[0]->c->nranks) return ncclInvalidArgument;   // a rank is missing from the group
        const size_t n = v[0]->count;
        if (v[0]->dtype != ncclInt64) return ncclInvalidArgument;
        std::vector<int64_t> sum(n, 0);
        for (Op* o : v) {
            if (o->count != n) return ncclInvalidArgument;
            for (size_t i = 0; i < n; i++) sum[i] += ((const int64_t*)o->src)[i];
        }
        for (Op* o : v) memcpy(o->dst, sum.data(), n * 8);
    }
    // send / recv: a send of rank r to peer p pairs with the recv of rank p from peer r, in posting order
    for (size_t i = 0; i < g_ops.size(); i++) {
        Op& s = g_ops[i];
        if (s.kind != 1) continue;
        bool done = false;
        for (size_t j = 0; j < g_ops.size() && !done; j++) {
            Op& r = g_ops[j];
            if (r.kind != 2 || r.c->id != s.c->id || r.c->rank != s.peer || r.peer != s.c->rank || r.src) continue;
            if (r.count != s.count || r.dtype != s.dtype) return ncclInvalidArgument;
            r.src = s.src;   // matched
            done = true;
        }
        if (!done) return ncclInvalidArgument;
    }
    // copy after matching (a rank may send from and receive into disjoint buffers only; stage through a temporary)
    std::vector<std::vector<char>> tmp;
    for (Op& r : g_ops) if (r.kind == 2) {
        if (!r.src) return ncclInvalidArgument;
        tmp.emplace_back((const char*)r.src, (const char*)r.src + r.count * width(r.dtype));
    }
    size_t k = 0;
    for (Op& r : g_ops) if (r.kind == 2) { memcpy(r.dst, tmp[k].data(), tmp[k].size()); k++; }
    return ncclSuccess;
}
}  // namespace

extern "C" {
ncclResult_t ncclGetUniqueId(ncclUniqueId* id) { memset(id, 0, sizeof(*id)); const int v = g_next_id++; memcpy(id->internal, &v, sizeof(v)); return ncclSuccess; }
ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank) {
    Comm* c = new Comm();
    memcpy(&c->id, id.internal, sizeof(int));
    c->nranks = nranks;
    c->rank = rank;
    *comm = (ncclComm_t)c;
    return ncclSuccess;
}
ncclResult_t ncclCommInitAll(ncclComm_t* comms, int n, const int*) {
    const int id = g_next_id++;
    for (int i = 0; i < n; i++) { Comm* c = new Comm{id, n, i}; comms[i] = (ncclComm_t)c; }
    return ncclSuccess;
}
ncclResult_t ncclCommDestroy(ncclComm_t comm) { delete (Comm*)comm; return ncclSuccess; }
ncclResult_t ncclGroupStart() { g_depth++; return ncclSuccess; }
ncclResult_t ncclGroupEnd() { if (g_depth <= 0) return ncclInvalidArgument; return --g_depth == 0 ? flush() : ncclSuccess; }
ncclResult_t ncclAllReduce(const void* s, void* d, size_t n, ncclDataType_t dt, ncclRedOp_t op, ncclComm_t c, void*) {
    if (op != ncclSum) return ncclInvalidArgument;
    t_ops.push_back(Op{0, s, d, n, (int)dt, -1, (Comm*)c});
    return g_depth ? ncclSuccess : flush();
}
ncclResult_t ncclSend(const void* s, size_t n, ncclDataType_t dt, int peer, ncclComm_t c, void*) {
    t_ops.push_back(Op{1, s, nullptr, n, (int)dt, peer, (Comm*)c});
    return g_depth ? ncclSuccess : ncclInvalidArgument;
}
ncclResult_t ncclRecv(void* d, size_t n, ncclDataType_t dt, int peer, ncclComm_t c, void*) {
    t_ops.push_back(Op{2, nullptr, d, n, (int)dt, peer, (Comm*)c});
    return g_depth ? ncclSuccess : ncclInvalidArgument;
}
const char* ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "success" : "stub: invalid argument / unmatched operation"; }
}
