// Launch-plan runner (include/mi355det_plan.h): replays a pre-marshalled table of C-ABI launches on the main / side HIP
// streams with the fork edges that order the side stream behind its producers.  Host code only.
//
// Every stream-taking entry point of include/*.h gets a typed thunk (plan_table.inc is generated from the headers by
// tools/gen_plan_table.py): the thunk reads each parameter from its 8-byte slot with the parameter's own type, so an
// int, a float, a double and a pointer all travel the same way and no calling-convention trick is involved.
#include "common.h"

#include <type_traits>
#include <utility>

#include "mi355det_bn.h"
#include "mi355det_conv.h"
#include "mi355det_effdet.h"
#include "mi355det_input.h"
#include "mi355det_optim.h"
#include "mi355det_plan.h"
#include "mi355det_syncbn.h"

using namespace mi355;

namespace {

template <typename T>
inline T slot(const uint64_t* a, size_t i) {
    static_assert(sizeof(T) <= 8 && std::is_trivially_copyable<T>::value, "a parameter must fit an 8-byte slot");
    T v;
    memcpy(&v, a + i, sizeof(T));
    return v;
}

template <typename T, bool LAST>
inline T param(const uint64_t* a, size_t i, void* stream) {
    if constexpr (LAST) {
        static_assert(std::is_same<T, void*>::value, "the last parameter of a plan entry point is `void* stream`");
        return stream;
    } else {
        return slot<T>(a, i);
    }
}

template <typename... A, size_t... I>
inline int invoke(int (*fn)(A...), const uint64_t* a, void* stream, std::index_sequence<I...>) {
    return fn(param<A, I + 1 == sizeof...(A)>(a, I, stream)...);
}

template <typename... A>
constexpr int arity_of(int (*)(A...)) { return (int)sizeof...(A) - 1; }

typedef int (*Thunk)(const uint64_t*, void*);
struct Entry { const char* name; Thunk call; int arity; };

template <typename... A>
struct Seq { typedef std::make_index_sequence<sizeof...(A)> type; };
template <typename... A>
typename Seq<A...>::type seq_of(int (*)(A...)) { return {}; }

#define MI355_PLAN_FN(NAME) \
    static int thunk_##NAME(const uint64_t* a, void* s) { return invoke(&NAME, a, s, decltype(seq_of(&NAME)){}); }
#include "plan_table.inc"
#undef MI355_PLAN_FN

const Entry g_table[] = {
#define MI355_PLAN_FN(NAME) {#NAME, &thunk_##NAME, arity_of(&NAME)},
#include "plan_table.inc"
#undef MI355_PLAN_FN
};
constexpr int g_count = (int)(sizeof(g_table) / sizeof(g_table[0]));

// Timing-free events for the fork / join edges: one pool per (thread, device) — an event belongs to the device that
// was current when it was made, and a thread may drive plans on several devices (nn.DataParallel).  hipStreamWaitEvent
// takes the record that is current when it is CALLED, so a slot can be recorded again as soon as its wait is issued.
int edge(hipStream_t from, hipStream_t to, const char* what) {
    constexpr int POOL = 64, MAXDEV = 16;
    static thread_local hipEvent_t ev[MAXDEV][POOL];
    static thread_local int made[MAXDEV] = {0};
    static thread_local int next[MAXDEV] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAXDEV) return fail(MI355_ELAUNCH, "%s: device %d", what, dev);
    const int s = next[dev];
    if (s >= made[dev]) {
        if (hipEventCreateWithFlags(&ev[dev][s], hipEventDisableTiming) != hipSuccess)
            return fail(MI355_ELAUNCH, "%s: event pool", what);
        made[dev] = s + 1;
    }
    next[dev] = (s + 1) % POOL;
    if (hipEventRecord(ev[dev][s], from) != hipSuccess) return fail(MI355_ELAUNCH, "%s: record", what);
    if (hipStreamWaitEvent(to, ev[dev][s], 0) != hipSuccess) return fail(MI355_ELAUNCH, "%s: wait", what);
    return MI355_OK;
}

}  // namespace

extern "C" int mi355_plan_lookup(const char* name) {
    if (!name) return -1;
    int lo = 0, hi = g_count - 1;               // plan_table.inc is sorted
    while (lo <= hi) {
        const int mid = (lo + hi) / 2, c = strcmp(name, g_table[mid].name);
        if (c == 0) return mid;
        if (c < 0) hi = mid - 1; else lo = mid + 1;
    }
    return -1;
}

extern "C" int mi355_plan_arity(int fn) { return (fn >= 0 && fn < g_count) ? g_table[fn].arity : -1; }

extern "C" int mi355_plan_run(const Mi355PlanStep* steps, int begin, int end, void* main_stream, void* side_stream,
                              void* side2_stream, Mi355PlanState* st) {
    MI355_REQUIRE(steps && st && begin >= 0 && end >= begin, "plan_run: bad range [%d, %d)", begin, end);
    // main has "moved" since a side stream's last fork: a run starts with both marked (the caller may have enqueued
    // anything on main between two runs)
    bool moved[2] = {true, true};
    st->forks = 0;
    for (int i = begin; i < end; ++i) {
        const Mi355PlanStep& p = steps[i];
        if (p.host) { st->next = i; return 1; }
        if (p.where == MI355_PLAN_JOIN || (p.where == MI355_PLAN_JOIN2 && !side2_stream)) {
            if (side_stream) {                             // a join edge inside the list (side2 -> side -> main)
                const int rc = mi355_plan_join(main_stream, side_stream, side2_stream, st, 0);
                if (rc) { st->failed = i; return rc; }
                moved[0] = moved[1] = true;
            }
            continue;
        }
        if (p.where == MI355_PLAN_JOIN2) {                 // main waits for the branch stream only
            if (st->side2_dirty) {
                const int rc = edge((hipStream_t)side2_stream, (hipStream_t)main_stream, "plan_run: join2");
                if (rc) { st->failed = i; return rc; }
                st->side2_dirty = 0;
                moved[0] = moved[1] = true;
            }
            continue;
        }
        if (p.fn < 0 || p.fn >= g_count || p.nargs != g_table[p.fn].arity) {
            st->failed = i;
            return fail(MI355_EINVAL, "plan_run: step %d names entry point %d with %d arguments", i, p.fn, p.nargs);
        }
        void* target = main_stream;
        if (side_stream && p.where != MI355_PLAN_MAIN) {
            int which = 0;
            if (p.where == MI355_PLAN_SIDE_RR && side2_stream) which = (st->rr++) & 1;
            if (p.where == MI355_PLAN_SIDE2 && side2_stream) which = 1;
            target = which ? side2_stream : side_stream;
            if (moved[which]) {
                const int rc = edge((hipStream_t)main_stream, (hipStream_t)target, "plan_run: fork");
                if (rc) { st->failed = i; return rc; }
                moved[which] = false;
                ++st->forks;
            }
            (which ? st->side2_dirty : st->side_dirty) = 1;
        } else {
            moved[0] = moved[1] = true;
        }
        const int rc = g_table[p.fn].call(p.arg, target);
        if (rc) { st->failed = i; return rc; }
    }
    return MI355_OK;
}

extern "C" int mi355_plan_join(void* main_stream, void* side_stream, void* side2_stream, Mi355PlanState* st, int fold_only) {
    MI355_REQUIRE(st, "plan_join: state");
    if (st->side2_dirty && side2_stream && side_stream) {
        const int rc = edge((hipStream_t)side2_stream, (hipStream_t)side_stream, "plan_join: side2 -> side");
        if (rc) return rc;
        // (a fold leaves side2 marked: main has not waited for it, and a MI355_PLAN_JOIN2 step of a later run - the
        //  backward is replayed range by range - must still issue its side2 -> main edge)
        if (!fold_only) st->side2_dirty = 0;
        st->side_dirty = 1;
    }
    if (fold_only) return MI355_OK;
    if (st->side_dirty && side_stream) {
        const int rc = edge((hipStream_t)side_stream, (hipStream_t)main_stream, "plan_join: side -> main");
        if (rc) return rc;
        st->side_dirty = 0;
    }
    return MI355_OK;
}

// ---------------------------------------------------------------------------------------------- RCCL step
// ncclAllReduce(sendbuff, recvbuff, count, datatype, op, comm, stream): resolved once from the RCCL the process already
// holds (torch's bundled librccl.so: the library the communicator belongs to), never linked against.
#include <dlfcn.h>
extern "C" int mi355_rccl_allreduce_sum_f32(void* comm, float* buf, int64_t n, void* stream) {
    typedef int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t);
    static AllReduce fn = [] {
        void* sym = dlsym(RTLD_DEFAULT, "ncclAllReduce");
        if (!sym) {
            void* h = dlopen("librccl.so", RTLD_NOW | RTLD_NOLOAD);
            if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);
            if (h) sym = dlsym(h, "ncclAllReduce");
        }
        return (AllReduce)sym;
    }();
    MI355_REQUIRE(comm && buf && n > 0, "rccl_allreduce: bad args");
    MI355_REQUIRE(fn != nullptr, "rccl_allreduce: no RCCL library is loaded in this process (ncclAllReduce not found)");
    constexpr int kFloat32 = 7, kSum = 0;                         // ncclFloat32, ncclSum (rccl.h)
    const int rc = fn(buf, buf, (size_t)n, kFloat32, kSum, comm, (hipStream_t)stream);
    if (rc != 0) return fail(MI355_ELAUNCH, "rccl_allreduce: ncclAllReduce returned %d", rc);
    return MI355_OK;
}

// ---------------------------------------------------------------------------------------------- zero ranges
struct ZeroRange { void* p; int64_t bytes; };

__global__ void __launch_bounds__(256) zero_ranges_kernel(const ZeroRange* __restrict__ tab, int n) {
    // grid = (blocks per range, n): a block strides over its range in 16-B pieces (ranges are 4-byte granular:
    // head and tail dwords by the first threads)
    const ZeroRange r = tab[blockIdx.y];
    char* p = (char*)r.p;
    const int64_t head = ((16 - ((uintptr_t)p & 15)) & 15);
    const int64_t h = head < r.bytes ? head : r.bytes;
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t nth = (int64_t)gridDim.x * blockDim.x;
    if (tid * 4 < h) *(uint32_t*)(p + tid * 4) = 0u;
    const int64_t body = (r.bytes - h) / 16;
    uint4* q = (uint4*)(p + h);
    for (int64_t i = tid; i < body; i += nth) q[i] = make_uint4(0u, 0u, 0u, 0u);
    const int64_t tail0 = h + body * 16;
    if (tail0 + tid * 4 < r.bytes) *(uint32_t*)(p + tail0 + tid * 4) = 0u;
}

extern "C" int mi355_zero_ranges(const void* table, int n, int blocks, void* stream) {
    MI355_REQUIRE(n >= 0 && n <= 65535 && (table || n == 0) && blocks >= 1 && blocks <= 4096, "zero_ranges: bad args");
    if (n == 0) return MI355_OK;
    hipLaunchKernelGGL(zero_ranges_kernel, dim3(blocks, n), dim3(256), 0, (hipStream_t)stream, (const ZeroRange*)table, n);
    return check_launch("zero_ranges");
}
