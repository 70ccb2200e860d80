// Shared device/host helpers for the gfx950 kernels (wave64, MFMA, LDS-DMA).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/hawkeye_hip.h"
#include <atomic>

// Once per (kernel instantiation, device): hipFuncSetAttribute acts on the CURRENT device's copy of the function, and one
// process may drive two devices (the two-device placement of set_device_ids).  `done` is a per-instantiation bit mask of
// the devices already set up; racing first calls set the same value.  Returns hipSuccess or the failing call's error.
static inline hipError_t hk_lds_attr_once(std::atomic<unsigned long long>& done, const void* func, int lds_bytes) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    if ((done.load(std::memory_order_acquire) >> dev) & 1ull) return hipSuccess;
    e = hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    if (e == hipSuccess) done.fetch_or(1ull << dev, std::memory_order_release);
    return e;
}

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

#define HK_LOG2E 1.4426950408889634f

// fp32 -> bf16 with round-to-nearest-even; hipcc lowers the cast to v_cvt_pk_bf16_f32 on gfx950.
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
    bf16x2 v;
    v[0] = (__bf16)lo;
    v[1] = (__bf16)hi;
    return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ float bf16_lo(uint32_t w) { return __builtin_bit_cast(float, w << 16); }
__device__ __forceinline__ float bf16_hi(uint32_t w) { return __builtin_bit_cast(float, w & 0xffff0000u); }

// LDS-DMA: 16 bytes per lane, destination = wave-uniform LDS base + lane * 16.
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// LDS-DMA with a scalar base: address = SGPR pair + one 32-bit per-lane byte offset (no 64-bit vector address
// arithmetic per piece).  Inline asm: the builtin always takes a 64-bit vector address.  lds = the wave's destination.
__device__ __forceinline__ void glds16_s(const void* sbase, uint32_t voff, const void* lds) {
    const uint32_t la = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)lds;
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                 :
                 : "v"(voff), "s"(sbase), "s"(la)
                 : "memory");
}
// The same with a 64-bit per-lane address.  A kernel that uses these helpers issues EVERY LDS-DMA through them, so the
// compiler itself never holds a value in M0 there (M0 is reserved and cannot be named as a clobber), and retires them with
// its own counted s_waitcnt vmcnt (the compiler does not know these statements write LDS).
__device__ __forceinline__ void glds16_v(const void* vaddr, const void* lds) {
    const uint32_t la = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)lds;
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off"
                 :
                 : "v"(vaddr), "s"(la)
                 : "memory");
}

// 4 bytes per lane (256 B per wave-instruction), scalar base + 32-bit lane offset.
__device__ __forceinline__ void glds4_s(const void* sbase, uint32_t voff, const void* lds) {
    const uint32_t la = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)lds;
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1"
                 :
                 : "v"(voff), "s"(sbase), "s"(la)
                 : "memory");
}

void hk_set_error(const char* fmt, ...);

#define HK_REQUIRE(cond, ...)            \
    do {                                 \
        if (!(cond)) {                   \
            hk_set_error(__VA_ARGS__);   \
            return HK_EINVAL;            \
        }                                \
    } while (0)

#define HK_CHECK_LAUNCH(name)                                               \
    do {                                                                    \
        hipError_t e_ = hipGetLastError();                                  \
        if (e_ != hipSuccess) {                                             \
            hk_set_error("%s: launch failed: %s", name, hipGetErrorString(e_)); \
            return HK_ELAUNCH;                                              \
        }                                                                   \
    } while (0)
