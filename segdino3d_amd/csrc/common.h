// Internal helpers shared by the HIP translation units (gfx950 / CDNA4 only, wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define SD3D_OK 0
#define SD3D_ERR_ARG -1
#define SD3D_ERR_WS -2
#define SD3D_ERR_LAUNCH -3
#define SD3D_ERR_RANGE -4

#define SD3D_CHECK_LAUNCH()                                   \
    do {                                                      \
        hipError_t e_ = hipGetLastError();                    \
        if (e_ != hipSuccess) return sd3d_set_error(SD3D_ERR_LAUNCH, hipGetErrorString(e_)); \
    } while (0)

int sd3d_set_error(int code, const char* msg);

static inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// ------------------------------------------------------------------ voxel keys
// key = morton48(x, y, z) with x, y, z < 2^16 (bit 3i = x_i, 3i+1 = y_i, 3i+2 = z_i).  Bits 48..55
// hold the batch index.  Sorting keys therefore sorts voxels along a Z-order curve, and the key of
// the parent voxel at the next coarser level is (key >> 3) (morton part).
#define SD3D_MORTON_BITS 48
#define SD3D_MORTON_MASK ((1ull << 48) - 1ull)
#define SD3D_EMPTY_KEY 0xFFFFFFFFFFFFFFFFull

__host__ __device__ static inline uint64_t spread3_16(uint32_t v) {
    uint64_t x = v & 0xFFFFull;                     // the standard 21-bit ladder, fed 16 bits
    x = (x | (x << 32)) & 0x001f00000000ffffull;
    x = (x | (x << 16)) & 0x001f0000ff0000ffull;
    x = (x | (x << 8)) & 0x100f00f00f00f00full;
    x = (x | (x << 4)) & 0x10c30c30c30c30c3ull;
    x = (x | (x << 2)) & 0x1249249249249249ull;
    return x;
}
__host__ __device__ static inline uint32_t compact3_16(uint64_t x) {
    x &= 0x1249249249249249ull;
    x = (x ^ (x >> 2)) & 0x10c30c30c30c30c3ull;
    x = (x ^ (x >> 4)) & 0x100f00f00f00f00full;
    x = (x ^ (x >> 8)) & 0x001f0000ff0000ffull;
    x = (x ^ (x >> 16)) & 0x001f00000000ffffull;
    x = (x ^ (x >> 32)) & 0x00000000001fffffull;
    return (uint32_t)x;
}
__host__ __device__ static inline uint64_t morton_encode(uint32_t x, uint32_t y, uint32_t z) {
    return spread3_16(x) | (spread3_16(y) << 1) | (spread3_16(z) << 2);
}
__host__ __device__ static inline void morton_decode(uint64_t m, uint32_t& x, uint32_t& y, uint32_t& z) {
    x = compact3_16(m);
    y = compact3_16(m >> 1);
    z = compact3_16(m >> 2);
}

__device__ static inline uint32_t hash_u64(uint64_t k) {
    k ^= k >> 33;
    k *= 0xff51afd7ed558ccdull;
    k ^= k >> 33;
    k *= 0xc4ceb9fe1a85ec53ull;
    k ^= k >> 33;
    return (uint32_t)k;
}

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
