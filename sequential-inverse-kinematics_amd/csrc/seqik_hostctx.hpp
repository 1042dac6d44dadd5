// seqik_hostctx.hpp -- pooled context of a host-buffer call, shared by the translation units of the library.
//
// A host-buffer entry point (seqik_solve_seq, seqik_solve_generic, seqik_head_angles) needs a stream and device
// buffers for the duration of the call.  Contexts are pooled per device (seqik_hip.hip, HostCtx): a call borrows a free
// one, carves its buffers from the context's grow-only arena and hands it back, so repeated calls neither create
// streams nor call hipMalloc / hipFree (which drains the device).  seqik_release_workspaces() frees the idle ones.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

namespace seqik {

struct HostLease {
    void *ctx = nullptr;          // opaque (HostCtx *)
    hipStream_t stream = nullptr;
    char *arena = nullptr;        // valid after host_lease_reserve
};

// Borrow a context of the CURRENT device (SEQIK_OK or SEQIK_ERR_HIP with the message set).
int host_lease_acquire(HostLease *lease);
// Make the arena at least `bytes` long (may synchronise the context's stream and reallocate).
int host_lease_reserve(HostLease *lease, size_t bytes);
void host_lease_release(HostLease *lease);

struct HostLeaseGuard {  // releases on scope exit
    HostLease lease;
    ~HostLeaseGuard() { if (lease.ctx) host_lease_release(&lease); }
};

inline size_t arena_padded(size_t bytes) { return (bytes + 255) & ~(size_t)255; }

}  // namespace seqik
