// mx_dev.hpp -- device helpers shared by the kernel files (internal).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "mx_kernels.hpp"

namespace mx {

// ---------------------------------------------------------------------------------------------
// guarded quad access: full quads are one dwordx4, the (single) partial tail quad goes scalar
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float4 ld4(const float* __restrict__ p, size_t q, size_t n) {
    const size_t b = q * 4;
    if (b + 4 <= n) return reinterpret_cast<const float4*>(p)[q];
    float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
    if (b < n) r.x = p[b];
    if (b + 1 < n) r.y = p[b + 1];
    if (b + 2 < n) r.z = p[b + 2];
    return r;
}
__device__ __forceinline__ void st4(float* __restrict__ p, size_t q, size_t n, float4 v) {
    const size_t b = q * 4;
    if (b + 4 <= n) { reinterpret_cast<float4*>(p)[q] = v; return; }
    if (b < n) p[b] = v.x;
    if (b + 1 < n) p[b + 1] = v.y;
    if (b + 2 < n) p[b + 2] = v.z;
}
__device__ __forceinline__ float2 ld2(const float* __restrict__ p, size_t h, size_t n) {
    const size_t b = h * 2;
    if (b + 2 <= n) return reinterpret_cast<const float2*>(p)[h];
    float2 r = make_float2(0.f, 0.f);
    if (b < n) r.x = p[b];
    return r;
}

inline unsigned grid_x(size_t items, unsigned block, unsigned cap) {
    size_t b = (items + block - 1) / block;
    if (b < 1) b = 1;
    if (b > cap) b = cap;
    return (unsigned)b;
}

inline int env_int(const char* name, int dflt) {
    const char* e = getenv(name);
    return e ? atoi(e) : dflt;
}

}  // namespace mx
