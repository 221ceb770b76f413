// fake_rccl.cpp -- a stand-in for librccl.so that lets the multi-device group's RCCL code path (csrc/group.cpp: ncclCommInitAll,
// the grouped ncclBroadcast of the weight blob, the grouped ncclSend / ncclRecv gather of label maps) EXECUTE on a box with one
// GPU: every "rank" may live on the same device, and a transfer is a device-to-device hipMemcpyAsync ordered behind the sender's
// stream.  Test infrastructure only (tests/test_gpu_group.py builds it with hipcc and points MIUNET_RCCL_LIB at it); it shares
// no code with RCCL and moves bytes exactly as the group's calls describe them -- so a wrong peer, offset or count shows up as
// wrong label maps.  Single-threaded group calls only (the group issues its collectives from one thread).
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdio>
#include <vector>

namespace {
struct Comm { int rank, size; };
struct Op { int kind; const void *send; void *recv; size_t bytes; int rank, peer; hipStream_t stream; };   // 0 bcast, 1 send, 2 recv
std::vector<Op> g_ops;
int g_depth = 0;
size_t dtype_bytes(int dt) { return (dt == 0 || dt == 1) ? 1 : (dt == 2 || dt == 3 || dt == 7) ? 4 : (dt == 6 || dt == 9) ? 2 : 8; }

int copy_after(void *dst, const void *src, size_t bytes, hipStream_t src_stream, hipStream_t dst_stream)
{
    hipEvent_t ev;
    if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) return 1;
    int rc = 0;
    if (hipEventRecord(ev, src_stream) != hipSuccess || hipStreamWaitEvent(dst_stream, ev, 0) != hipSuccess ||
        (dst != src && hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, dst_stream) != hipSuccess))
        rc = 1;
    (void)hipEventDestroy(ev);
    return rc;
}

int flush()
{
    int rc = 0;
    // broadcast: every non-root rank copies from the root's send buffer
    for (const Op &root : g_ops)
        if (root.kind == 0 && root.rank == root.peer)
            for (const Op &o : g_ops)
                if (o.kind == 0 && o.peer == root.peer && o.rank != root.rank) {
                    if (o.bytes != root.bytes) rc = 2;
                    else rc |= copy_after(o.recv, root.send, o.bytes, root.stream, o.stream);
                }
    // point to point: a send of rank s to peer d pairs with the recv of rank d from peer s, in issue order
    std::vector<char> used(g_ops.size(), 0);
    for (size_t i = 0; i < g_ops.size(); ++i) {
        if (g_ops[i].kind != 1) continue;
        bool matched = false;
        for (size_t j = 0; j < g_ops.size() && !matched; ++j)
            if (g_ops[j].kind == 2 && !used[j] && g_ops[j].rank == g_ops[i].peer && g_ops[j].peer == g_ops[i].rank) {
                used[j] = 1; matched = true;
                if (g_ops[j].bytes != g_ops[i].bytes) rc = 2;
                else rc |= copy_after(g_ops[j].recv, g_ops[i].send, g_ops[i].bytes, g_ops[i].stream, g_ops[j].stream);
            }
        if (!matched) rc = 3;                                   // a send nobody receives
    }
    for (size_t j = 0; j < g_ops.size(); ++j)
        if (g_ops[j].kind == 2 && !used[j]) rc = 3;             // a recv nobody feeds: the real library would hang here
    g_ops.clear();
    return rc;
}
}  // namespace

extern "C" {
int ncclCommInitAll(void **comms, int n, const int *) { for (int r = 0; r < n; ++r) comms[r] = new Comm{ r, n }; return 0; }
int ncclCommDestroy(void *c) { delete static_cast<Comm *>(c); return 0; }
int ncclGroupStart() { ++g_depth; return 0; }
int ncclGroupEnd() { return --g_depth == 0 ? flush() : 0; }
int ncclBroadcast(const void *send, void *recv, size_t count, int dt, int root, void *comm, hipStream_t s)
{
    g_ops.push_back({ 0, send, recv, count * dtype_bytes(dt), static_cast<Comm *>(comm)->rank, root, s });
    return g_depth ? 0 : flush();
}
int ncclSend(const void *send, size_t count, int dt, int peer, void *comm, hipStream_t s)
{
    g_ops.push_back({ 1, send, nullptr, count * dtype_bytes(dt), static_cast<Comm *>(comm)->rank, peer, s });
    return g_depth ? 0 : flush();
}
int ncclRecv(void *recv, size_t count, int dt, int peer, void *comm, hipStream_t s)
{
    g_ops.push_back({ 2, nullptr, recv, count * dtype_bytes(dt), static_cast<Comm *>(comm)->rank, peer, s });
    return g_depth ? 0 : flush();
}
const char *ncclGetErrorString(int rc) { return rc == 2 ? "fake rccl: byte counts of a matched pair differ" : rc == 3 ? "fake rccl: unmatched send / recv" : rc ? "fake rccl: HIP call failed" : "no error"; }
}
