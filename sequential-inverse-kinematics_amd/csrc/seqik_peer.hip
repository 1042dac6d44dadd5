// seqik_peer.hip -- the final joint-angle gather as peer WRITES over xGMI (include/seqik.h, "Peer gather").
//
// The north star's only exchange is the gather of every GPU's joint-angle block on rank 0.  Done with RCCL
// send / receive it costs the root compute units at the worst moment: its receive kernels copy 7 x 336 MB per step
// while its own solver launches keep every SIMD slot busy (rehearsed on one GPU with a receive-like kernel,
// scripts/contention/gather_contention.py: step 14.4 -> 16.5-18.5 ms on the root, and the job runs at the pace of
// its slowest rank).  xGMI is point-to-point and every GPU has copy engines: here rank 0 exports its receive
// buffers once (hipIpcGetMemHandle), every other rank maps its slot (hipIpcOpenMemHandle) and pushes its block with
// one hipMemcpyAsync per step -- an SDMA transfer, no compute unit involved on either side.  Ordering / completion
// is the caller's (seqikpy_amd/peer_gather.py: one 8-byte all-reduce per step behind the copy).
//
// No reference counterpart: the reference's "gather" is multiprocessing.Pool returning pickled dicts
// (examples/example_leg_inv_kinematics_parallel.py:186-187).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>

#include "../../include/seqik.h"

extern "C" void seqik_set_error(int code, const char *msg);

namespace {

int p_fail(const char *what, hipError_t e)
{
    char buf[384];
    snprintf(buf, sizeof(buf), "%s: %s", what, hipGetErrorString(e));
    seqik_set_error(SEQIK_ERR_HIP, buf);
    return SEQIK_ERR_HIP;
}

#define PTRY(expr)                                      \
    do {                                                \
        hipError_t e_ = (expr);                         \
        if (e_ != hipSuccess) return p_fail(#expr, e_); \
    } while (0)

static_assert(sizeof(hipIpcMemHandle_t) == SEQIK_PEER_HANDLE_BYTES, "handle size of the ABI");

}  // namespace

extern "C" {

int seqik_peer_alloc(void **d_ptr, size_t bytes)
{
    if (!d_ptr || bytes == 0) { seqik_set_error(SEQIK_ERR_BAD_ARG, "seqik_peer_alloc: null pointer / zero size"); return SEQIK_ERR_BAD_ARG; }
    PTRY(hipMalloc(d_ptr, bytes));  // its own allocation: an IPC handle names a whole allocation
    return SEQIK_OK;
}

int seqik_peer_free(void *d_ptr)
{
    if (d_ptr) PTRY(hipFree(d_ptr));
    return SEQIK_OK;
}

int seqik_peer_export(const void *d_ptr, unsigned char *handle)
{
    if (!d_ptr || !handle) { seqik_set_error(SEQIK_ERR_BAD_ARG, "seqik_peer_export: null pointer"); return SEQIK_ERR_BAD_ARG; }
    hipIpcMemHandle_t h;
    PTRY(hipIpcGetMemHandle(&h, const_cast<void *>(d_ptr)));
    memcpy(handle, &h, sizeof(h));
    return SEQIK_OK;
}

int seqik_peer_open(const unsigned char *handle, void **d_ptr)
{
    if (!handle || !d_ptr) { seqik_set_error(SEQIK_ERR_BAD_ARG, "seqik_peer_open: null pointer"); return SEQIK_ERR_BAD_ARG; }
    hipIpcMemHandle_t h;
    memcpy(&h, handle, sizeof(h));
    PTRY(hipIpcOpenMemHandle(d_ptr, h, hipIpcMemLazyEnablePeerAccess));
    return SEQIK_OK;
}

int seqik_peer_close(void *d_ptr)
{
    if (d_ptr) PTRY(hipIpcCloseMemHandle(d_ptr));
    return SEQIK_OK;
}

int seqik_peer_copy(void *d_dst, const void *d_src, size_t bytes, void *hip_stream)
{
    if (!d_dst || !d_src) { seqik_set_error(SEQIK_ERR_BAD_ARG, "seqik_peer_copy: null pointer"); return SEQIK_ERR_BAD_ARG; }
    if (bytes == 0) return SEQIK_OK;
    PTRY(hipMemcpyAsync(d_dst, d_src, bytes, hipMemcpyDeviceToDevice, static_cast<hipStream_t>(hip_stream)));
    return SEQIK_OK;
}

}  // extern "C"
