// common.hpp — shared device/host helpers for the gfx950 (MI355X, CDNA4) kernels of the Carla-ppo hot path.
// wave = 64 lanes everywhere; no CUDA-compat paths.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mi {

typedef unsigned short bf16_t;                                   // raw bfloat16 bits in HBM / LDS
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));       // MFMA operand type
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));

constexpr int WAVE = 64;

// ---- bf16 <-> f32 (round-to-nearest-even, same as torch) ----
__host__ __device__ __forceinline__ float bf16_to_f32(bf16_t v) {
    union { uint32_t u; float f; } c; c.u = (uint32_t)v << 16; return c.f;
}
__host__ __device__ __forceinline__ bf16_t f32_to_bf16(float f) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_bit_cast(unsigned short, (__bf16)f);          // gfx950: v_cvt_pk_bf16_f32 (round-to-nearest-even)
#endif
    union { uint32_t u; float f; } c; c.f = f;
    uint32_t u = c.u;
    if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)0x7fc0;  // NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}

// ---- split storage ("bf16x3" precision mode): one 4-byte element = two bf16 halves of one value, x = hi + lo ----
//   hi = bf16(x) (round to nearest even), lo = bf16(x - hi) (x - hi is exact in fp32); word = hi << 16 | lo, i.e. in memory [lo, hi].
// hi + lo carries 16-17 significand bits (|x - hi - lo| <= 2^-18 |x|).  The point of the layout: a 16-byte MFMA fragment of four such elements
// is the bf16x8 vector [l0 h0 l1 h1 l2 h2 l3 h3]; against the same fragment of the other operand the bf16 MFMA forms sum(la lb + ha hb), and
// against that fragment with the halves of every dword swapped sum(la hb + ha lb): TWO v_mfma_f32_32x32x16_bf16 (64 cycles) give the full
// (ha + la)(hb + lb) products of 8 k-values, fp32-accumulated, where exact fp32 needs four v_mfma_f32_32x32x2_f32 (256 cycles).  Every tensor keeps
// the element size, alignment and addressing of the fp32 engine, so the LDS-DMA tile kernels move split tiles exactly as they move fp32 tiles.
struct split_t { uint32_t u; };
__host__ __device__ __forceinline__ split_t split_from_f32(float x) {
    const bf16_t h = f32_to_bf16(x);
    const bf16_t l = f32_to_bf16(x - bf16_to_f32(h));
    split_t s; s.u = ((uint32_t)h << 16) | (uint32_t)l; return s;
}
__host__ __device__ __forceinline__ float split_to_f32(split_t v) {
    union { uint32_t u; float f; } h, l; h.u = v.u & 0xffff0000u; l.u = v.u << 16; return h.f + l.f;
}
template <typename T> struct is_split { static constexpr bool value = false; };
template <> struct is_split<split_t> { static constexpr bool value = true; };
// the zero element of a storage type (split_t is a struct: no (T)0)
template <typename T> __host__ __device__ __forceinline__ T zero_of() { return (T)0; }
template <> __host__ __device__ __forceinline__ split_t zero_of<split_t>() { split_t s; s.u = 0u; return s; }
// the value 1.0 in a storage type
template <typename T> __host__ __device__ __forceinline__ T one_of();
template <> __host__ __device__ __forceinline__ float one_of<float>() { return 1.0f; }
template <> __host__ __device__ __forceinline__ bf16_t one_of<bf16_t>() { return (bf16_t)0x3F80; }
template <> __host__ __device__ __forceinline__ split_t one_of<split_t>() { split_t s; s.u = 0x3F800000u; return s; }
template <> __host__ __device__ __forceinline__ unsigned char one_of<unsigned char>() { return 1; }

// ---- raw camera bytes -> [0, 1] (the reference's host preprocessing `frame.astype(np.float32) / 255.0`, vae/train_vae.py:15-18) ----
// exact: q = k * fl(1/255), one Newton correction with two FMAs gives the correctly rounded float32(k) / float32(255) for every k in 0..255
// (checked exhaustively, tests/test_oracle_golden.py); for bf16 storage the plain product already rounds to the same bf16 value.
constexpr float U8_RCP255 = 0.003921568859368563f;
__host__ __device__ __forceinline__ float u8_to_unit_exact(float k) {
    const float q = k * U8_RCP255;
    const float r = __builtin_fmaf(-q, 255.0f, k);
    return __builtin_fmaf(r, U8_RCP255, q);
}
// source element -> float for the narrow-layer loaders: float / bf16 bits / uint8 camera byte (bf16-exact form, see above)
template <typename TS> __device__ __forceinline__ float src_to_f32(TS v);
template <> __device__ __forceinline__ float src_to_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ float src_to_f32<bf16_t>(bf16_t v) { return bf16_to_f32(v); }
template <> __device__ __forceinline__ float src_to_f32<unsigned char>(unsigned char v) { return (float)v * U8_RCP255; }
template <> __device__ __forceinline__ float src_to_f32<split_t>(split_t v) { return split_to_f32(v); }

// N consecutive elements moved as one naturally aligned vector access
template <typename TT, int N> struct alignas(sizeof(TT) * N) PackN { TT v[N]; };

template <typename T> struct Elem;
template <> struct Elem<float> {
    static __host__ __device__ __forceinline__ float to_f32(float v) { return v; }
    static __host__ __device__ __forceinline__ float from_f32(float v) { return v; }
};
template <> struct Elem<bf16_t> {
    static __host__ __device__ __forceinline__ float to_f32(bf16_t v) { return bf16_to_f32(v); }
    static __host__ __device__ __forceinline__ bf16_t from_f32(float v) { return f32_to_bf16(v); }
};

template <> struct Elem<split_t> {
    static __host__ __device__ __forceinline__ float to_f32(split_t v) { return split_to_f32(v); }
    static __host__ __device__ __forceinline__ split_t from_f32(float v) { return split_from_f32(v); }
};

// 4 consecutive fp32 values -> 4 storage elements (bf16: two v_cvt_pk_bf16_f32)
typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
template <typename T> __device__ __forceinline__ PackN<T, 4> pack4(const float (&v)[4]);
template <> __device__ __forceinline__ PackN<float, 4> pack4<float>(const float (&v)[4]) {
    PackN<float, 4> o; o.v[0] = v[0]; o.v[1] = v[1]; o.v[2] = v[2]; o.v[3] = v[3]; return o;
}
template <> __device__ __forceinline__ PackN<bf16_t, 4> pack4<bf16_t>(const float (&v)[4]) {
    const f32x4 f = {v[0], v[1], v[2], v[3]};
    return __builtin_bit_cast(PackN<bf16_t, 4>, __builtin_convertvector(f, bf16x4_t));
}

template <> __device__ __forceinline__ PackN<split_t, 4> pack4<split_t>(const float (&v)[4]) {
    // hi halves with two packed converts, residuals, lo halves with two more; (hi & 0xffff0000) is the hi value's fp32 pattern
    const f32x4 f = {v[0], v[1], v[2], v[3]};
    const PackN<bf16_t, 4> h = __builtin_bit_cast(PackN<bf16_t, 4>, __builtin_convertvector(f, bf16x4_t));
    f32x4 r;
#pragma unroll
    for (int e = 0; e < 4; ++e) r[e] = v[e] - bf16_to_f32(h.v[e]);
    const PackN<bf16_t, 4> l = __builtin_bit_cast(PackN<bf16_t, 4>, __builtin_convertvector(r, bf16x4_t));
    PackN<split_t, 4> o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o.v[e].u = ((uint32_t)h.v[e] << 16) | (uint32_t)l.v[e];
    return o;
}

// ---- division by a runtime-invariant divisor: q = (n * mul) >> 40 style, exact for 0 <= n < 2^31 ----
// host computes mul = floor(2^(31+s) / d) + 1 with s = ceil(log2 d); n*mul < 2^63.
struct FastDiv {
    uint32_t mul; uint32_t shift; uint32_t d; uint32_t pad;
    __host__ __device__ __forceinline__ uint32_t div(uint32_t n) const {
        return (uint32_t)(((uint64_t)n * mul) >> shift);
    }
    __host__ __device__ __forceinline__ void divmod(uint32_t n, uint32_t& q, uint32_t& r) const {
        q = div(n); r = n - q * d;
    }
};
static inline FastDiv make_fastdiv(uint32_t d) {
    FastDiv f; f.d = d; f.pad = 0;
    if (d <= 1) { f.mul = 1; f.shift = 0; f.d = 1; return f; }
    uint32_t s = 0; while ((1ull << s) < d) ++s;
    f.shift = 31 + s;
    f.mul = (uint32_t)(((1ull << f.shift) / d) + 1);
    return f;
}

// ---- wave / block reductions ----
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

}  // namespace mi
