// f32_split.h -- float32 products on the bfloat16 matrix pipes at float32 accuracy (device code shared by conv_kernels.hip and freq_kernels.hip).
//
// Every operand x is split exactly into three bfloat16 parts, x = h + m + l with h = bf16(x), m = bf16(x - h), l = bf16(x - h - m) (round to nearest; the
// differences are exact in float32; what is left after l is below 2^-24 |x|), and a product a b is accumulated as the six part products of order up to
// 2^-16: l_a h_b, h_a l_b, m_a m_b, m_a h_b, h_a m_b, h_a h_b (each exact: 8 x 8 significand bits; the three dropped ones are below 2^-24 |a b|, the size
// of the float32 rounding of the sum itself).  Accumulation is the MFMA's float32.  Six v_mfma_f32_32x32x16_bf16 of 32 cycles replace eight
// v_mfma_f32_32x32x2_f32 of 64 cycles for the same 16 values of K: 2.67 x the matrix rate; the price is 5.5 vector instructions per operand value.
// Measured against float64 (tests/test_detector.py, tests/test_freqdomain.py) the error is at or below the float32 instruction's.
// Not the same as the instruction at the edges of the format: an infinite x gives NaN parts (x - h = inf - inf), and below |x| = 2^-118 the parts m and l
// fall under bfloat16's smallest normal number (it shares float32's exponent range) and may be flushed -- relative error up to 2^-8 on values that small.
#pragma once
#include <hip/hip_runtime.h>

namespace bf {
namespace split {

typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8v __attribute__((ext_vector_type(8)));
typedef float float2v __attribute__((ext_vector_type(2)));
typedef float float4v __attribute__((ext_vector_type(4)));
typedef float float16v __attribute__((ext_vector_type(16)));
typedef unsigned uint4v __attribute__((ext_vector_type(4)));
struct Split3 { uint4v h, m, l; };

// (The differences are single v_sub_f32: hipcc pairs them into v_pk_add_f32 when it can, and that instruction neither runs in the shadow of an MFMA nor
//  at the rate of two plain ones -- scripts/dev/mfma_valu_probe.hip: 24 MFMAs interleaved with 144 v_add_f32 take 1.38 x the MFMAs' own time at one wave
//  per SIMD and 1.2 x at two, with 144 v_pk_add_f32 2.2 x.)
__device__ __forceinline__ float sub_f32(float a, float b)
{
    float r;
    asm("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

__device__ __forceinline__ Split3 split3(const float4v& p, const float4v& q)       // eight float32 values -> three vectors of eight bfloat16
{
    Split3 r;
#ifdef BF_DIAG_NO_SPLIT                                      // (timing experiment only: the operand bits as they are, results wrong)
    r.h = __builtin_bit_cast(uint4v, p); r.m = __builtin_bit_cast(uint4v, q); r.l = r.h ^ r.m;
    return r;
#endif
    const float2v v[4] = {{p[0], p[1]}, {p[2], p[3]}, {q[0], q[1]}, {q[2], q[3]}};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const unsigned h = __builtin_bit_cast(unsigned, __builtin_convertvector(v[i], bf16x2v));
        const float2v r1 = {sub_f32(v[i][0], __uint_as_float(h << 16)), sub_f32(v[i][1], __uint_as_float(h & 0xffff0000u))};
        const unsigned m = __builtin_bit_cast(unsigned, __builtin_convertvector(r1, bf16x2v));
        const float2v r2 = {sub_f32(r1[0], __uint_as_float(m << 16)), sub_f32(r1[1], __uint_as_float(m & 0xffff0000u))};
        r.h[i] = h; r.m[i] = m;
        r.l[i] = __builtin_bit_cast(unsigned, __builtin_convertvector(r2, bf16x2v));
    }
    return r;
}

__device__ __forceinline__ void negate(Split3& x)           // -x: the sign bits of all 24 parts
{
    x.h ^= 0x80008000u; x.m ^= 0x80008000u; x.l ^= 0x80008000u;
}

// acc += (a, 16 values of K) x (b, the same 16 values): the six part products, smallest first
__device__ __forceinline__ void mfma6(float16v& acc, const Split3& a, const Split3& b)
{
#define BF_SPLIT_TERM(P, Q) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8v, a.P), __builtin_bit_cast(bf16x8v, b.Q), acc, 0, 0, 0)
    BF_SPLIT_TERM(l, h); BF_SPLIT_TERM(h, l); BF_SPLIT_TERM(m, m); BF_SPLIT_TERM(m, h); BF_SPLIT_TERM(h, m); BF_SPLIT_TERM(h, h);
#undef BF_SPLIT_TERM
}

}  // namespace split
}  // namespace bf
