// hip_raii.hpp — events and streams that are destroyed on every way out of a C-ABI entry point (the *_HIP error macros
// return early).
#pragma once
#include <hip/hip_runtime.h>

namespace hipraii {

struct Event {
    hipEvent_t e = nullptr;
    Event() = default;
    Event(const Event&) = delete;
    Event& operator=(const Event&) = delete;
    ~Event() { if (e) (void)hipEventDestroy(e); }
    hipError_t create() { return hipEventCreate(&e); }
    operator hipEvent_t() const { return e; }
};

struct Stream {
    hipStream_t s = nullptr;
    Stream() = default;
    Stream(const Stream&) = delete;
    Stream& operator=(const Stream&) = delete;
    ~Stream() { if (s) (void)hipStreamDestroy(s); }
    hipError_t create(unsigned flags = hipStreamDefault) { return hipStreamCreateWithFlags(&s, flags); }
    operator hipStream_t() const { return s; }
};

}  // namespace hipraii
