// mx_common.hpp -- error type, HIP checking, owning device buffer (internal).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <stdexcept>
#include <string>

#include "../../include/mixlab_gpu.h"

namespace mx {

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

void hip_check(hipError_t e, const char* what);

// simple owning device buffer
struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    void alloc(size_t n);
    void free_();
    ~DevBuf() { free_(); }
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    DevBuf(DevBuf&& o) noexcept : p(o.p), bytes(o.bytes) { o.p = nullptr; o.bytes = 0; }
    DevBuf& operator=(DevBuf&& o) noexcept { if (this != &o) { free_(); p = o.p; bytes = o.bytes; o.p = nullptr; o.bytes = 0; } return *this; }
};

}  // namespace mx
