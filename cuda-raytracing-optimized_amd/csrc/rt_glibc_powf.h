/* rt_glibc_powf.h — powf(x, 5.0f) as the CPU side of this project computes it, restated so that the device computes the same bits.
 *
 * schlick (/root/reference/material.h:12) calls pow(1 - cosine, 5.0f) on floats, i.e. powf; on the CPU (the oracle, the reference-header twin) that is
 * glibc's libm.  glibc 2.35's powf is one published algorithm (ARM optimized-routines, sysdeps/ieee754/flt-32/e_powf.c): log2(x) from a 16-entry table
 * (1/c, log2 c) and a degree-5 fp64 polynomial, y log2(x) in fp64, exp2 from a 32-entry table of 2^(k/32) and a degree-3 fp64 polynomial, ONE rounding to
 * fp32 at the end.  On x86-64 CPUs with FMA the ifunc selects __powf_fma; which operations that build fuses was read from libm.so.6's own code (objdump of
 * the function the powf@@GLIBC_2.27 resolver returns) - every a*b+c of the source is one fused multiply-add, spelt out below with explicit fma():
 *
 *   log2:  r = fma(z, invc, -1);  y0 = logc + k;  r2 = r r;  y = fma(r, A0, A1);  p = fma(r, A2, A3);  q = fma(r, A4, y0);  r4 = r2 r2;
 *          q = fma(r2, p, q);  log2x = fma(y, r4, q)
 *   exp2:  kd = ylogx + SHIFT;  ki = bits(kd);  kd -= SHIFT;  r = ylogx - kd;  z = fma(r, C0, C1);  r2 = r r;  y = fma(r, C2, 1);  y = fma(z, r2, y);
 *          result = (float) (y * 2^(ki/32))          (TOINT_INTRINSICS = 0 on x86-64: POWF_SCALE = 1, the SHIFT form)
 *
 * The tables are __powf_log2_data / __exp2f_data of that libm, read from its .rodata at the addresses the code loads them from.  Pinned by
 * tests/test_oracle_golden.py::test_glibc_powf5_twin_is_libm: this text compiled for the host equals libm's powf(x, 5.0f) in every bit on ALL floats of
 * [0, 2] - everything 1 - min(cos, 1) can be - and on a stride over every other bit pattern (negative, huge, subnormal, inf, NaN), and by the device
 * probe rtProbeMath against the test machine's libm.  The special cases are written for y = 5 only (an odd integer: the sign of x is the sign of
 * the result).
 *
 * Included twice: by rt_device.h (RT_POWF_FN = __device__ __forceinline__) and by oracle/rt_oracle.c (static inline), one text for both.
 */
#ifndef RT_GLIBC_POWF_H
#define RT_GLIBC_POWF_H

#include <stdint.h>

#ifndef RT_POWF_TABLE
#define RT_POWF_TABLE static const
#endif

/* __powf_log2_data.tab: (invc, logc) of the 16 subintervals of [0x1.66p-1, 0x1.66p0) */
RT_POWF_TABLE double rt_powf_log2_tab[32] = {
    0x1.661ec79f8f3bep+0, -0x1.efec65b963019p-2, 0x1.571ed4aaf883dp+0, -0x1.b0b6832d4fca4p-2, 0x1.49539f0f010bp+0, -0x1.7418b0a1fb77bp-2,
    0x1.3c995b0b80385p+0, -0x1.39de91a6dcf7bp-2, 0x1.30d190c8864a5p+0, -0x1.01d9bf3f2b631p-2, 0x1.25e227b0b8eap+0, -0x1.97c1d1b3b7afp-3,
    0x1.1bb4a4a1a343fp+0, -0x1.2f9e393af3c9fp-3, 0x1.12358f08ae5bap+0, -0x1.960cbbf788d5cp-4, 0x1.0953f419900a7p+0, -0x1.a6f9db6475fcep-5,
    0x1p+0, 0x0p+0, 0x1.e608cfd9a47acp-1, 0x1.338ca9f24f53dp-4, 0x1.ca4b31f026aap-1, 0x1.476a9543891bap-3,
    0x1.b2036576afce6p-1, 0x1.e840b4ac4e4d2p-3, 0x1.9c2d163a1aa2dp-1, 0x1.40645f0c6651cp-2, 0x1.886e6037841edp-1, 0x1.88e9c2c1b9ff8p-2,
    0x1.767dcf5534862p-1, 0x1.ce0a44eb17bccp-2,
};
/* __exp2f_data.tab: bits of 2^(i/32) with i << 47 subtracted */
RT_POWF_TABLE uint64_t rt_powf_exp2_tab[32] = {
    0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull, 0x3fef72b83c7d517bull, 0x3fef54873168b9aaull,
    0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull, 0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull, 0x3feedea64c123422ull, 0x3feece086061892dull,
    0x3feebfdad5362a27ull, 0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull, 0x3feea47eb03a5585ull, 0x3feea09e667f3bcdull, 0x3fee9f75e8ec5f74ull,
    0x3feea11473eb0187ull, 0x3feea589994cce13ull, 0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull, 0x3feed503b23e255dull,
    0x3feee89f995ad3adull, 0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull, 0x3fef3720dcef9069ull, 0x3fef5818dcfba487ull, 0x3fef7c97337b9b5full,
    0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull,
};

/* log2_tab / exp2_tab: the two tables above, or a copy of them (the render kernels keep one in the LDS) */
RT_POWF_FN float rt_glibc_powf5_tab(float x, const double* log2_tab, const uint64_t* exp2_tab) {
#ifdef __clang__
#pragma clang fp contract(off)      /* (the FAST device build contracts by default: 5 log2x + SHIFT must stay two roundings) */
#endif
    const double A0 = 0x1.27616c9496e0bp-2, A1 = -0x1.71969a075c67ap-2, A2 = 0x1.ec70a6ca7baddp-2, A3 = -0x1.7154748bef6c8p-1, A4 = 0x1.71547652ab82bp0;
    const double C0 = 0x1.c6af84b912394p-5, C1 = 0x1.ebfce50fac4f3p-3, C2 = 0x1.62e42ff0c52d6p-1, SHIFT = 0x1.8p+47;       /* shift_scaled = 0x1.8p52 / 32 */
    union { float f; uint32_t u; } fb;
    union { double d; uint64_t u; } db;
    uint32_t ix, sign = 0u;
    fb.f = x;
    ix = fb.u;
    if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u) {            /* x < 0x1p-126, or inf, or NaN (e_powf.c: the unlikely block, with y = 5) */
        if (2u * ix - 1u >= 2u * 0x7f800000u - 1u) {                /* zeroinfnan(ix): x*x, negated when x carries a sign (5 is odd) */
            const float x2 = x * x;
            return (ix & 0x80000000u) ? -x2 : x2;
        }
        if (ix & 0x80000000u) { sign = 1u; ix &= 0x7fffffffu; }     /* x < 0 and y an odd integer: SIGN_BIAS */
        if (ix < 0x00800000u) {                                     /* subnormal: normalise so that the exponent becomes negative */
            fb.u = ix;
            fb.f = fb.f * 0x1p23f;
            ix = (fb.u & 0x7fffffffu) - (23u << 23);
        }
    }
    {
        /* log2_inline: x = 2^k z, z in [OFF, 2 OFF), c near the centre of z's subinterval */
        const uint32_t tmp = ix - 0x3f330000u;
        const int i = (int)((tmp >> 19) & 15u);
        const uint32_t top = tmp & 0xff800000u;
        const int k = (int32_t)top >> 23;
        double z, r, r2, r4, y, p, q, ylogx, kd;
        uint64_t ki, t;
        fb.u = ix - top;
        z = (double)fb.f;
        r = __builtin_fma(z, log2_tab[2 * i], -1.0);
        q = log2_tab[2 * i + 1] + (double)k;                /* y0 */
        r2 = r * r;
        y = __builtin_fma(r, A0, A1);
        p = __builtin_fma(r, A2, A3);
        q = __builtin_fma(r, A4, q);
        r4 = r2 * r2;
        q = __builtin_fma(r2, p, q);
        ylogx = 5.0 * __builtin_fma(y, r4, q);                      /* y * log2(x): one rounded product */
        db.d = ylogx;
        if (((db.u >> 47) & 0xffffu) >= 0x80bfu) {                  /* |y log2 x| >= 126 */
            if (ylogx > 0x1.fffffffd1d571p+6) return sign ? -__builtin_inff() : __builtin_inff();      /* __math_oflowf */
            if (ylogx <= -150.0) return sign ? -0.0f : 0.0f;                                           /* __math_uflowf */
            if (ylogx < -149.0) return sign ? -0x1p-149f : 0x1p-149f;                                  /* __math_may_uflowf: RN(0x1.4p-75f^2) */
        }
        /* exp2_inline */
        kd = ylogx + SHIFT;
        db.d = kd;
        ki = db.u;
        kd -= SHIFT;
        r = ylogx - kd;
        t = exp2_tab[ki & 31u] + ((ki + ((uint64_t)sign << 16)) << 47);
        z = __builtin_fma(r, C0, C1);
        r2 = r * r;
        y = __builtin_fma(r, C2, 1.0);
        y = __builtin_fma(z, r2, y);
        db.u = t;
        return (float)(y * db.d);
    }
}

RT_POWF_FN float rt_glibc_powf5(float x) { return rt_glibc_powf5_tab(x, rt_powf_log2_tab, rt_powf_exp2_tab); }

#endif
