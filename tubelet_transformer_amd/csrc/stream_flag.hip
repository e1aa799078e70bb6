// Software ordering edge between two HIP streams of one device: a counter in device memory instead of hipEventRecord / hipStreamWaitEvent.
// (SURVEY.md section 8 rows a19 / e / N4: the gradient exchange of the data-parallel step; reference: DistributedDataParallel's bucketed
// all-reduce overlapped with backward, utils/model_utils.py:47-52.)
//
// Why not an event: round 6 measured what a pending hipStreamWaitEvent costs on this runtime (scripts/queue_contention_probe.py,
// profiles/r06_queue_contention_probe.txt): as soon as a SECOND stream waits for an event this stream has yet to reach, every kernel of
// this stream is dispatched ~1.3 us later -- a captured graph of 900 small kernels replays in 2.65 ms instead of 1.43 ms, whatever the
// event's flags, while a kernel that merely RUNS on the other stream costs it 0.07 ms.  The training step is such a graph (907
// launches): the two edges around the gradient exchange cost it 0.55 - 0.8 ms with nothing on the wire (rounds 3 - 5 blamed RCCL's
// kernel for that; a one-rank in-place ncclAllReduce is a no-op on the device).
//
// So the producer stream SIGNALS by a one-thread kernel at the point the windows are final -- the last node of each captured graph part;
// kernels of a stream complete in order and release their writes at their end, so everything enqueued before it is visible device-wide
// when the counter moves -- and the consumer stream WAITS in a one-wave kernel that polls the counter (agent-scope acquire loads, s_sleep
// between polls) and is followed, in stream order, by the collective.  No runtime-level dependency exists between the two queues.  The
// wait is bounded (~2 s of wall clock): it raises an error word instead of hanging the GPU when its signal never comes.
#include "common.h"

namespace {

__global__ void flag_signal_kernel(unsigned* flag) {
    __hip_atomic_fetch_add(flag, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ void flag_wait_kernel(const unsigned* flag, unsigned expect, unsigned* err) {
    if (threadIdx.x != 0) return;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();          // 100 MHz
    // RELAXED agent-scope polls (an sc1 load served by the memory side, no cache maintenance): an ACQUIRE load here is a buffer_inv sc1 per
    // poll -- this XCD's L2 lines invalidated every microsecond under the kernels of the other stream (first version: the graph parts that
    // ran beside a polling wave were 0.25 - 0.35 ms slower each).  One acquire when the counter has moved.
    while ((int)(__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - expect) < 0) {
        __builtin_amdgcn_s_sleep(127);
        if (__builtin_amdgcn_s_memrealtime() - t0 > 200000000ull) {          // 2 s: the signal is not coming
            __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}

}  // namespace

extern "C" {

// *flag += 1 once everything enqueued on `stream` before this call has completed (capturable: the last node of a graph part)
int tuber_flag_signal(unsigned* flag, hipStream_t stream) {
    if (!flag) return TUBER_EINVAL;
    hipLaunchKernelGGL(flag_signal_kernel, dim3(1), dim3(1), 0, stream, flag);
    TUBER_RETURN_LAUNCH();
}

// what is enqueued on `stream` after this call runs once *flag >= expect (wrap-around safe).  After ~2 s without the signal the kernel
// sets *err = 1 and lets the stream go on: the caller checks the word where it synchronises.
int tuber_flag_wait(const unsigned* flag, unsigned expect, unsigned* err, hipStream_t stream) {
    if (!flag || !err) return TUBER_EINVAL;
    hipLaunchKernelGGL(flag_wait_kernel, dim3(1), dim3(64), 0, stream, flag, expect, err);
    TUBER_RETURN_LAUNCH();
}

}  // extern "C"
