/* rt_div64.h — IEEE-754 fp32 division by a shared denominator and fp32 square root, computed through fp64 so that they cost a fraction of the compiler's
 * expansion (v_div_scale / v_rcp / Newton chain / v_div_fmas / v_div_fixup per quotient) and return the SAME BITS.
 *
 * vec3 / float (/root/reference/vec3.h:79) is three correctly rounded divisions by one denominator; unit_vector (vec3.h:194) puts a correctly rounded square
 * root (vec3.h:35) in front of them: three call sites per ray of the render hot path.
 *   Division.  Let x = X 2^a, y = Y 2^b with 24-bit integers X, Y.  A rounding boundary of the float grid near x / y is (2M + 1) 2^c with a 24-bit M;
 *   x / y - (2M + 1) 2^c = (X 2^(a - b - c) - (2M + 1) Y) / Y 2^c: the numerator is an integer and not zero (it would need 2^24 | Y), so the quotient is at least
 *   2^-49 (relative) away from every boundary, and ANY approximation with a relative error below that rounds to the IEEE quotient.  Here: 1 / y from a seed of
 *   >= 20 good bits by two Newton steps in fp64 (error <= 2^-53 + 2^-80), times x in fp64 (one more rounding): 2^-52.
 *   Square root.  For a float s and a boundary m = (2M + 1) 2^c, s - m^2 is a non-zero multiple of 2^(2c), so |sqrt(s) - m| / m >= 1 / (8 M (M + 1)) > 2^-51.
 *   Here: the coupled (Goldschmidt) iteration g -> sqrt(s), h -> 1 / (2 sqrt(s)) from a seed of >= 20 good bits, two steps, then Markstein's correction
 *   g + (s - g g) h: within 2^-53 (1 + 2^-40) of sqrt(s).
 * Quotients that could round into the denormal range (|q| < 2^-100, q != 0) and operands outside [2^-60, 2^60] (s outside [2^-100, 2^100]) are left to the caller's
 * plain operators.  The seeds are parameters (the device passes v_rcp_f32 / v_rsq_f32; the host twin in oracle/rt_oracle.c passes perturbed ones): the result
 * does not depend on them.  Pinned by tests/test_oracle_golden.py::test_div64_twin_is_ieee and by the device probe rtProbeMath.
 */
#ifndef RT_DIV64_H
#define RT_DIV64_H

/* 1 / y in fp64 from a seed r0 with relative error <= 2^-20 */
RT_DIV64_FN double rt_recip64(double yd, double r0) {
    double e = __builtin_fma(-yd, r0, 1.0);
    double r = __builtin_fma(r0, e, r0);
    e = __builtin_fma(-yd, r, 1.0);
    return __builtin_fma(r, e, r);
}
RT_DIV64_FN int rt_q_safe(double q) { const double a = __builtin_fabs(q); return a >= 0x1p-100 || a == 0.0; }     /* (not near the denormal range) */
/* out = (ax / t, ay / t, az / t) with r = 1 / t from rt_recip64; returns 0 when the caller must use the plain operators */
RT_DIV64_FN int rt_div3_64(float ax, float ay, float az, double r, float* out) {
    const double qx = (double)ax * r, qy = (double)ay * r, qz = (double)az * r;
    if (!(rt_q_safe(qx) && rt_q_safe(qy) && rt_q_safe(qz))) return 0;
    out[0] = (float)qx; out[1] = (float)qy; out[2] = (float)qz;
    return 1;
}
/* sqrt(s) correctly rounded to fp32, for s in [2^-100, 2^100], from a seed y0 ~ 1 / sqrt(s) with relative error <= 2^-20; *half_inv = 1 / (2 sqrt(s)) in fp64 */
RT_DIV64_FN float rt_sqrt64(float s, double y0, double* half_inv) {
    const double sd = (double)s;
    double g = sd * y0, h = 0.5 * y0;
    double r = __builtin_fma(-g, h, 0.5);
    g = __builtin_fma(g, r, g); h = __builtin_fma(h, r, h);
    r = __builtin_fma(-g, h, 0.5);
    g = __builtin_fma(g, r, g); h = __builtin_fma(h, r, h);
    g = __builtin_fma(__builtin_fma(-g, g, sd), h, g);
    *half_inv = h;
    return (float)g;
}

#endif
