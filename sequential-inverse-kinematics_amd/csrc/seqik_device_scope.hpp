// seqik_device_scope.hpp -- the entry points that select a GPU themselves (host-buffer calls, stream / statistics
// handles) switch the calling thread to that device for the duration of the call only and put the caller's device
// back on return: in a one-process-per-GPU job a library call must not move the thread to another GPU behind the
// caller's back.  SeqikOptions.device < 0 means "the calling thread's current device".
#pragma once
#include <hip/hip_runtime.h>

namespace seqik {

struct DeviceScope {
    int prev = -1;
    bool switched = false;
    // device < 0: stay on the current device
    hipError_t enter(int device)
    {
        hipError_t e = hipGetDevice(&prev);
        if (e != hipSuccess) return e;
        if (device >= 0 && device != prev) {
            e = hipSetDevice(device);
            if (e != hipSuccess) return e;
            switched = true;
        }
        return hipSuccess;
    }
    ~DeviceScope()
    {
        if (switched) (void)hipSetDevice(prev);
    }
};

// ordinal a handle remembers: the requested one, or the current device when it is negative / absent
inline hipError_t resolve_device(int requested, int *out)
{
    if (requested >= 0) { *out = requested; return hipSuccess; }
    return hipGetDevice(out);
}

}  // namespace seqik
