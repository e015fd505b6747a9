// philox.h -- Philox4x32-10 counter-based RNG (Salmon et al., SC'11) for the device-RNG mode.
// The reference draws everything from NumPy's global MT19937; that stream
// is serial by construction, so device-RNG mode uses a counter keyed by (seed, GLOBAL particle
// index, stream, block): results do not depend on the number of GPUs or on the launch geometry.
#pragma once
#ifndef __HIPCC_RTC__       // hiprtc (user_model.hip hands it this file as an in-memory header) brings its own runtime declarations
#include <hip/hip_runtime.h>
#include <stdint.h>
#endif

namespace smc {

#ifdef __HIPCC_RTC__        // hiprtc keeps its fixed-width integer types in a namespace of its own
using uint32_t = unsigned int;
using uint64_t = unsigned long long;
#endif

struct u32x4 {
    uint32_t x, y, z, w;
};

__host__ __device__ inline u32x4 philox4x32_10(u32x4 c, uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c.x;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c.z;
        u32x4 n;
        n.x = (uint32_t)(p1 >> 32) ^ c.y ^ k0;
        n.y = (uint32_t)p1;
        n.z = (uint32_t)(p0 >> 32) ^ c.w ^ k1;
        n.w = (uint32_t)p0;
        c = n;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return c;
}

// counter layout: (gidx lo, gidx hi, stream lo, stream hi[23:0] << 8 | block)
__host__ __device__ inline u32x4 philox_block(uint64_t seed, uint64_t gidx, uint64_t stream, uint32_t block) {
    u32x4 c;
    c.x = (uint32_t)gidx;
    c.y = (uint32_t)(gidx >> 32);
    c.z = (uint32_t)stream;
    c.w = (((uint32_t)(stream >> 32)) << 8) | (block & 0xFFu);
    return philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
}

// 53-bit uniform in [0,1) from two 32-bit words (the construction NumPy's random_sample uses)
__host__ __device__ inline double u01_from(uint32_t a, uint32_t b) {
    return (double)(((uint64_t)(a >> 5) << 26) | (uint64_t)(b >> 6)) * (1.0 / 9007199254740992.0);
}

#define SMC_PHILOX_BLOCK_UNIFORM 255u  // block index of the acceptance uniform

}  // namespace smc
