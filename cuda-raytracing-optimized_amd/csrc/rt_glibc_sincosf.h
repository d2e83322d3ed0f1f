/* rt_glibc_sincosf.h — the fp32 sine / cosine the CPU side of this project computes with, restated so that the device computes the same bits.
 *
 * generateShadowRay (/root/reference/kernels.cu:378-379) calls cosf(phi) and sinf(phi); on the CPU (the oracle, the reference-header twin) that
 * is glibc's libm.  glibc 2.35's sinf / cosf / sincosf are one published algorithm (ARM optimized-routines, sysdeps/ieee754/flt-32/s_sincosf.h):
 * the argument is widened to fp64, reduced by a multiple of pi/2 when |y| >= pi/4, and a degree-7 / degree-8 fp64 polynomial is rounded to fp32
 * once.  On x86-64 CPUs with FMA the ifunc selects the *_fma builds; which operations those fuse was read from libm.so.6's own code
 * (objdump of __sinf_fma / __cosf_fma / __sincosf_fma, identical expression DAGs in all three) and is spelt out below with explicit fma():
 *
 *   sin poly:  x2 = x x;  x3 = x x2;  x5 = x2 x3;  s1 = fma(x2, S3, S2);  s = fma(x3, S1, x);  result = fma(s1, x5, s)
 *   cos poly:  x4 = x2 x2;  x6 = x2 x4;  c1 = fma(x2, C1, C0);  c2 = fma(x2, C4, C3);  c = fma(x4, C2, c1);  result = fma(c2, x6, c)
 *   reduction: r = x (2/pi 2^24);  n = ((int32) r + 2^23) >> 24;  xr = fma(-n, pi/2, x);  sign and sin/cos swap by quadrant n
 *
 * The constants are __sincosf_table[0] of that libm.  Pinned by tests/test_oracle_golden.py::test_glibc_sincosf_twin_is_libm: this text compiled
 * for the host equals libm's sinf, cosf and sincosf on ALL 2^24 arguments phi = (float)(2 pi k / 2^24) that generateShadowRay can produce, and on
 * 2^22 arguments spread over (-120, 120).  |y| >= 120 (never reached by phi < 2 pi) is left to the caller's fallback.
 *
 * Included twice: by rt_device.h (RT_SINCOS_FN = __device__ __forceinline__) and by oracle/rt_oracle.c (static inline), one text for both.
 * Must be compiled without floating-point contraction of the un-fused products (the PARITY build and the oracle both use -ffp-contract=off).
 */
#ifndef RT_GLIBC_SINCOSF_H
#define RT_GLIBC_SINCOSF_H

#include <stdint.h>

/* returns 0 when |y| >= 120 or y is not finite (caller falls back), 1 otherwise */
RT_SINCOS_FN int rt_glibc_sincosf(float y, float* sinp, float* cosp) {
    const double C0 = 0x1p0, C1 = -0x1.ffffffd0c621cp-2, C2 = 0x1.55553e1068f19p-5, C3 = -0x1.6c087e89a359dp-10, C4 = 0x1.99343027bf8c3p-16;
    const double S1 = -0x1.555545995a603p-3, S2 = 0x1.1107605230bc4p-7, S3 = -0x1.994eb3774cf24p-13;
    const double HPI_INV_2P24 = 0x1.45F306DC9C883p+23, HPI = 0x1.921FB54442D18p0;
    union { float f; uint32_t u; } bits;
    bits.f = y;
    const uint32_t top = (bits.u >> 20) & 0x7ffu;                /* abstop12 */
    double x = (double)y;
    int n = 0;
    if (top >= 0x42fu) return 0;                                 /* |y| >= 120, inf, nan */
    if (top < 0x3f4u) {                                          /* |y| < pi/4 */
        if (top < 0x398u) {                                      /* |y| < 2^-12 */
            *sinp = y;
            *cosp = 1.0f;
            return 1;
        }
    } else {
        const double r = x * HPI_INV_2P24;
        n = ((int32_t)r + 0x800000) >> 24;
        x = __builtin_fma(-(double)n, HPI, x);
    }
    {
        const double xs = x * (((n + 1) & 2) ? -1.0 : 1.0);     /* sign[n & 3] = {1, -1, -1, 1} (n = 0, the un-reduced path: x * 1 = x exactly) */
        const double x2 = x * x;
        const double x3 = x2 * xs, x4 = x2 * x2;
        const double x5 = x2 * x3, x6 = x2 * x4;
        const double s1 = __builtin_fma(x2, S3, S2);
        const double s = __builtin_fma(x3, S1, xs);
        const double c1 = __builtin_fma(x2, C1, C0);
        const double c2 = __builtin_fma(x2, C4, C3);
        const double c = __builtin_fma(x4, C2, c1);
        const float sv = (float)__builtin_fma(s1, x5, s);
        const float cp = (float)__builtin_fma(c2, x6, c);
        const float cv = (n & 2) ? -cp : cp;                     /* __sincosf_table[1]: every cosine coefficient negated = the result negated, exactly */
        *sinp = (n & 1) ? cv : sv;
        *cosp = (n & 1) ? sv : cv;
    }
    return 1;
}

#endif
