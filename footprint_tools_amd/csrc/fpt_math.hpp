// fpt_math.hpp -- float64 special functions for the footprint scan, as inlined
// device functions for gfx950.
//
// These follow the branch structure and coefficients of the Cephes routines the
// reference reaches through hcephes v0.4.1 (citations per function, paths under
// /root/reference/hcephes/src), so that results agree with the reference to a
// few ulp; they are not required to be bit-identical (libm differs on device and
// the compiler may contract a*b+c).  Everything is branch-light and free of
// global state (the reference's `sgngam` / `merror` globals are dropped).
//
// The header is also compilable by a host C++ compiler (FPT_HD expands to
// nothing) so that tests can check the math on a machine without a GPU.
#pragma once

#include <math.h>
#include <stdint.h>

#include "fpt_ndtr_gtab.hpp"

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define FPT_HD __host__ __device__ __forceinline__
#define FPT_D __device__ __forceinline__
#else
#define FPT_HD inline
#define FPT_D inline
#endif

namespace fptm {

// hcephes/include/hcephes.h:75-84, cprob/incbet.c:3-6
constexpr double kMachEp = 1.11022302462515654042E-16;  // 2^-53
constexpr double kMaxLog = 7.09782712893383996732E2;
constexpr double kMinLog = -7.451332191019412076235E2;
constexpr double kPi = 3.14159265358979323846;
constexpr double kSqrtH = 7.07106781186547524401E-1;
constexpr double kMaxGam = 171.624376956302725;
constexpr double kBig = 4.503599627370496e15;
constexpr double kBigInv = 2.22044604925031308085e-16;
constexpr double kSqrt2Pi = 2.50662827463100050242E0;
constexpr double kInf = __builtin_huge_val();

// polyn/polevl.c:3-17 -- Horner, N+1 coefficients, highest power first.
template <int N>
FPT_HD double horner(double x, const double (&c)[N + 1]) {
    double acc = c[0];
#pragma unroll
    for (int i = 1; i <= N; ++i) acc = acc * x + c[i];
    return acc;
}

// polyn/polevl.c:19-33 -- leading coefficient 1 implied, N coefficients given.
template <int N>
FPT_HD double horner1(double x, const double (&c)[N]) {
    double acc = x + c[0];
#pragma unroll
    for (int i = 1; i < N; ++i) acc = acc * x + c[i];
    return acc;
}

// ---------------------------------------------------------------------------
// gamma / lgam  (cprob/gamma.c)
// ---------------------------------------------------------------------------

// gamma.c:35-49
FPT_HD double stirling_gamma(double x) {
    const double kStir[5] = {7.87311395793093628397E-4, -2.29549961613378126380E-4,
                             -2.68132617805781232825E-3, 3.47222221605458667310E-3,
                             8.33333333333482257126E-2};
    double w = 1.0 / x;
    w = 1.0 + w * horner<4>(w, kStir);
    double y = exp(x);
    if (x > 143.01608) {
        double v = pow(x, 0.5 * x - 0.25);
        y = v * (v / y);
    } else {
        y = pow(x, x - 0.5) / y;
    }
    return kSqrt2Pi * y * w;
}

// gamma.c:51-127
FPT_HD double gamma_fn(double x) {
    const double kP[7] = {1.60119522476751861407E-4, 1.19135147006586384913E-3,
                          1.04213797561761569935E-2, 4.76367800457137231464E-2,
                          2.07448227648435975150E-1, 4.94214826801497100753E-1,
                          9.99999999999999996796E-1};
    const double kQ[8] = {-2.31581873324120129819E-5, 5.39605580493303397842E-4,
                          -4.45641913851797240494E-3, 1.18139785222060435552E-2,
                          3.58236398605498653373E-2,  -2.34591795718243348568E-1,
                          7.14304917030273074085E-2,  1.00000000000000000320E0};
    if (isnan(x)) return x;
    if (x == kInf) return x;
    if (x == -kInf) return NAN;
    double q = fabs(x);
    if (q > 33.0) {
        if (x >= 0.0) return stirling_gamma(x);
        double p = floor(q);
        if (p == q) return NAN;
        double sgn = (((int)p) & 1) == 0 ? -1.0 : 1.0;
        double z = q - p;
        if (z > 0.5) {
            p += 1.0;
            z = q - p;
        }
        z = q * sin(kPi * z);
        if (z == 0.0) return sgn * kInf;
        z = fabs(z);
        z = kPi / (z * stirling_gamma(q));
        return sgn * z;
    }
    double z = 1.0;
    while (x >= 3.0) {
        x -= 1.0;
        z *= x;
    }
    bool tiny = false;
    while (x < 0.0) {
        if (x > -1.E-9) {
            tiny = true;
            break;
        }
        z /= x;
        x += 1.0;
    }
    while (!tiny && x < 2.0) {
        if (x < 1.e-9) {
            tiny = true;
            break;
        }
        z /= x;
        x += 1.0;
    }
    if (tiny) {
        if (x == 0.0) return NAN;
        return z / ((1.0 + 0.5772156649015329 * x) * x);
    }
    if (x == 2.0) return z;
    x -= 2.0;
    return z * horner<6>(x, kP) / horner<7>(x, kQ);
}

// gamma.c:152-235, for x >= -34 plus the reflection branch (one level deep).
FPT_HD double lgam_pos(double x) {
#pragma clang fp contract(off)  // (x - 0.5) log x - x and log z + p are two roundings each in gamma.c
    const double kA[5] = {8.11614167470508450300E-4, -5.95061904284301438324E-4,
                          7.93650340457716943945E-4, -2.77777777730099687205E-3,
                          8.33333333333331927722E-2};
    const double kB[6] = {-1.37825152569120859100E3, -3.88016315134637840924E4,
                          -3.31612992738871184744E5, -1.16237097492762307383E6,
                          -1.72173700820839662146E6, -8.53555664245765465627E5};
    const double kC[6] = {-3.51815701436523470549E2, -1.70642106651881159223E4,
                          -2.20528590553854454839E5, -1.13933444367982507207E6,
                          -2.53252307177582951285E6, -2.01889141433532773231E6};
    // One logarithm for both ranges (of the reduction's product below 13, of x itself above): the values
    // are gamma.c's, operation for operation, but a wavefront whose lanes fall on both sides of 13 -- the
    // posterior's lgam(k + r) and lgam(r) mostly do -- runs the ~35 instructions of log once, not twice.
    const bool small = x < 13.0;
    double z = 1.0, p = 0.0, u = x;
    if (small) {
        while (u >= 3.0) {
            p -= 1.0;
            u = x + p;
            z *= u;
        }
        while (u < 2.0) {
            if (u == 0.0) return kInf;
            z /= u;
            p += 1.0;
            u = x + p;
        }
        z = fabs(z);
    } else if (x > 2.556348e305) {
        return kInf;
    }
    const double lg = log(small ? z : x);
    if (small) {
        if (u == 2.0) return lg;
        p -= 2.0;
        x = x + p;
        p = x * horner<5>(x, kB) / horner1<6>(x, kC);
        return lg + p;
    }
    double q = (x - 0.5) * lg - x + 0.91893853320467274178;
    if (x > 1.0e8) return q;
    p = 1.0 / (x * x);
    if (x >= 1000.0)
        q += ((7.9365079365079365079365e-4 * p - 2.7777777777777777777778e-3) * p +
              0.0833333333333333333333) /
             x;
    else
        q += horner<4>(p, kA) / x;
    return q;
}

FPT_HD double lgam(double x) {
    if (isnan(x)) return x;
    if (isinf(x)) return kInf;
    if (x < -34.0) {  // gamma.c:163-188
        double q = -x;
        double w = lgam_pos(q);
        double p = floor(q);
        if (p == q) return kInf;
        double z = q - p;
        if (z > 0.5) {
            p += 1.0;
            z = p - q;
        }
        z = q * sin(kPi * z);
        if (z == 0.0) return kInf;
        return 1.14472988584940017414 - log(z) - w;
    }
    return lgam_pos(x);
}

// ---------------------------------------------------------------------------
// regularised incomplete beta (cprob/incbet.c)
// ---------------------------------------------------------------------------

// incbet.c:266-285: the series part of pseries (before its gamma/pow scaling)
FPT_HD double ibeta_pseries_sum(double a, double b, double x) {
    double ai = 1.0 / a;
    double u = (1.0 - b) * x;
    double v = u / (a + 1.0);
    double t1 = v;
    double t = u;
    double n = 2.0;
    double s = 0.0;
    double z = kMachEp * ai;
    while (fabs(v) > z) {
        u = (n - b) * x / n;
        t *= u;
        v = t / (a + n);
        s += v;
        n += 1.0;
    }
    s += t1;
    s += ai;
    return s;
}

// incbet.c:100-177 (second == false) and :183-261 (second == true): one
// recurrence, two coefficient schedules.  Selecting the schedule per lane keeps
// a wavefront in a single loop instead of two divergent ones.
FPT_HD double ibeta_contfrac(bool second, double a, double b, double x) {
    double k1 = a, k3 = a, k4 = a + 1.0, k5 = 1.0, k7 = a + 1.0, k8 = a + 2.0;
    double k2 = second ? b - 1.0 : a + b;
    double k6 = second ? a + b : b - 1.0;
    double d2 = second ? -1.0 : 1.0;
    double v = second ? x / (1.0 - x) : x;
    double pkm2 = 0.0, qkm2 = 1.0, pkm1 = 1.0, qkm1 = 1.0;
    double ans = 1.0, r = 1.0;
    const double thresh = 3.0 * kMachEp;
    for (int n = 0; n < 300; ++n) {
        double xk = -(v * k1 * k2) / (k3 * k4);
        double pk = pkm1 + pkm2 * xk;
        double qk = qkm1 + qkm2 * xk;
        pkm2 = pkm1;
        pkm1 = pk;
        qkm2 = qkm1;
        qkm1 = qk;

        xk = (v * k5 * k6) / (k7 * k8);
        pk = pkm1 + pkm2 * xk;
        qk = qkm1 + qkm2 * xk;
        pkm2 = pkm1;
        pkm1 = pk;
        qkm2 = qkm1;
        qkm1 = qk;

        if (qk != 0) r = pk / qk;
        double t;
        if (r != 0) {
            t = fabs((ans - r) / r);
            ans = r;
        } else {
            t = 1.0;
        }
        if (t < thresh) break;

        k1 += 1.0;
        k2 += d2;
        k3 += 2.0;
        k4 += 2.0;
        k5 += 1.0;
        k6 -= d2;
        k7 += 2.0;
        k8 += 2.0;

        if ((fabs(qk) + fabs(pk)) > kBig) {
            pkm2 *= kBigInv;
            pkm1 *= kBigInv;
            qkm2 *= kBigInv;
            qkm1 *= kBigInv;
        }
        if ((fabs(qk) < kBigInv) || (fabs(pk) < kBigInv)) {
            pkm2 *= kBig;
            pkm1 *= kBig;
            qkm2 *= kBig;
            qkm1 *= kBig;
        }
    }
    return ans;
}

#if defined(__clang__)
#define FPT_NOUNROLL _Pragma("nounroll")
#else
#define FPT_NOUNROLL
#endif

// incbet.c:12-94 with pseries' epilogue (:287-298) merged into the common one: the power
// series and the continued fractions end in the same x^a (1-x)^b Gamma(a+b)/(Gamma(a)Gamma(b))
// scaling, so the gamma / lgam / pow evaluations are issued once for all lanes of a
// wavefront whichever expansion each lane took.  Operation order per branch is the reference's.
FPT_HD double incbet(double aa, double bb, double xx) {
    if (aa <= 0.0 || bb <= 0.0) return 0.0;
    if (xx <= 0.0 || xx >= 1.0) return (xx == 1.0) ? 1.0 : 0.0;

    const bool direct = (bb * xx) <= 1.0 && xx <= 0.95;
    const bool flipped = !direct && xx > (aa / (aa + bb));
    const double a = flipped ? bb : aa;
    const double b = flipped ? aa : bb;
    const double xc = flipped ? xx : 1.0 - xx;
    const double x = flipped ? 1.0 - xx : xx;
    const bool series = direct || (flipped && (b * x) <= 1.0 && x <= 0.95);

    double w;  // series sum, or continued-fraction value
    if (series) {
        w = ibeta_pseries_sum(a, b, x);
    } else {
        double y = x * (a + b - 2.0) - (a - 1.0);
        bool second = !(y < 0.0);
        w = ibeta_contfrac(second, a, b, x);
        if (second) w = w / xc;
    }

    const double apb = a + b;
    const double u = a * log(x);
    const double tl = series ? 0.0 : b * log(xc);
    double t;
    if (apb < kMaxGam && fabs(u) < kMaxLog && fabs(tl) < kMaxLog) {
        double g[3];
        FPT_NOUNROLL
        for (int i = 0; i < 3; ++i) g[i] = gamma_fn(i == 0 ? apb : (i == 1 ? a : b));
        double gr = g[0] / (g[1] * g[2]);
        double pw[2];
#if defined(__HIP_DEVICE_COMPILE__)
        // x^a and xc^b from the logarithms already taken for the range checks: exp(a log x)
        // is within |a log x| ulp (< 1e-13 relative here) of pow(x, a) and a third of its cost
        pw[0] = exp(u);
        pw[1] = exp(tl);
#else
        FPT_NOUNROLL
        for (int i = 0; i < 2; ++i) pw[i] = pow(i == 0 ? x : xc, i == 0 ? a : b);
#endif
        if (series) {
            t = w * gr * pw[0];
        } else {
            t = pw[1];
            t *= pw[0];
            t /= a;
            t *= w;
            t *= gr;
        }
    } else {
        double lg[3];
        FPT_NOUNROLL
        for (int i = 0; i < 3; ++i) lg[i] = lgam(i == 0 ? apb : (i == 1 ? a : b));
        double y;
        if (series) {
            y = lg[0] - lg[1] - lg[2] + u + log(w);
        } else {
            y = u;
            y += tl + lg[0] - lg[1] - lg[2];
            y += log(w / a);
        }
        t = (y < kMinLog) ? 0.0 : exp(y);
    }
    if (flipped) t = (t <= kMachEp) ? 1.0 - kMachEp : 1.0 - t;
    return t;
}

// ---------------------------------------------------------------------------
// normal cdf and quantile (cprob/ndtr.c, cprob/ndtri.c, cprob/expx2.c)
// ---------------------------------------------------------------------------

// expx2.c:6-34
FPT_HD double expx2(double x, int sign) {
    x = fabs(x);
    if (sign < 0) x = -x;
    double m = .0078125 * floor(128.0 * x + 0.5);
    double f = x - m;
    double u = m * m;
    double u1 = 2 * m * f + f * f;
    if (sign < 0) {
        u = -u;
        u1 = -u1;
    }
    if ((u + u1) > kMaxLog) return kInf;
    return exp(u) * exp(u1);
}

// ndtr.c:65-77 (numerator/denominator of exp(x^2) erfc(x), x >= 1)
FPT_HD void erfc_rational(double x, double &p, double &q) {
    const double kP[9] = {2.46196981473530512524E-10, 5.64189564831068821977E-1,
                          7.46321056442269912687E0,   4.86371970985681366614E1,
                          1.96520832956077098242E2,   5.26445194995477358631E2,
                          9.34528527171957607540E2,   1.02755188689515710272E3,
                          5.57535335369399327526E2};
    const double kQ[8] = {1.32281951154744992508E1, 8.67072140885989742329E1,
                          3.54937778887819891062E2, 9.75708501743205489753E2,
                          1.82390916687909736289E3, 2.24633760818710981792E3,
                          1.65666309194161350182E3, 5.57535340817727675546E2};
    const double kR[6] = {5.64189583547755073984E-1, 1.27536670759978104416E0,
                          5.01905042251180477414E0,  6.16021097993053585195E0,
                          7.40974269950448939160E0,  2.97886665372100240670E0};
    const double kS[6] = {2.26052863220117276590E0, 9.39603524938001434673E0,
                          1.20489539808096656605E1, 1.70814450747565897222E1,
                          9.60896809063285878198E0, 3.36907645100081516050E0};
    if (x < 8.0) {
        p = horner<8>(x, kP);
        q = horner1<8>(x, kQ);
    } else {
        p = horner<5>(x, kR);
        q = horner1<6>(x, kS);
    }
}

FPT_HD double erfce(double x) {
    double p, q;
    erfc_rational(x, p, q);
    return p / q;
}

FPT_HD double erfc_fn(double a);

// ndtr.c:79-87 (|x| <= 1 series part)
FPT_HD double erf_small(double x) {
    const double kT[5] = {9.60497373987051638749E0, 9.00260197203842689217E1,
                          2.23200534594684319226E3, 7.00332514112805075473E3,
                          5.55923013010394962768E4};
    const double kU[5] = {3.35617141647503099647E1, 5.21357949780152679795E2,
                          4.59432382970980127987E3, 2.26290000613890934246E4,
                          4.92673942608635921086E4};
    double z = x * x;
    return x * horner<4>(z, kT) / horner1<5>(z, kU);
}

// ndtr.c:89-132 for |a| >= 1
FPT_HD double erfc_large(double a) {
    double x = fabs(a);
    double z = -a * a;
    if (z < -kMaxLog) return (a < 0) ? 2.0 : 0.0;
    z = expx2(a, -1);
    double p, q;
    erfc_rational(x, p, q);
    double y = (z * p) / q;
    if (a < 0) y = 2.0 - y;
    if (y == 0.0) return (a < 0) ? 2.0 : 0.0;
    return y;
}

FPT_HD double erf_fn(double x) {
    if (fabs(x) > 1.0) return 1.0 - erfc_large(x);
    return erf_small(x);
}

FPT_HD double erfc_fn(double a) {
    if (fabs(a) < 1.0) return 1.0 - erf_small(a);
    return erfc_large(a);
}

// ndtr.c:34-59, split at its branch so that a caller with several arguments per lane can run
// the cheap central branch for all of them and the tail branch only where it is needed
FPT_HD bool ndtr_is_central(double a) { return fabs(a * kSqrtH) < 1.0; }
FPT_HD double ndtr_central(double a) { return 0.5 + 0.5 * erf_small(a * kSqrtH); }
// the tail branch before its last step: y = Phi(-|a|); ndtr is y for a < 0 and 1 - y for a > 0.
// *ec = erfce(|a| / sqrt 2), the scaled complementary error function y is made of: the hazard
// phi(a) / y = sqrt(2 / pi) / ec follows from it without another exp.
FPT_HD double ndtr_tail_y(double a, double *ec = nullptr) {
    double x = a * kSqrtH;
    double z = fabs(x);
    const double ece = erfce(z);
    if (ec) *ec = ece;
    double y = 0.5 * ece;
#if defined(__HIP_DEVICE_COMPILE__)
    // exp(-a^2/2) for the tail.  The reference takes sqrt(expx2(a, -1)) (two exps and a square
    // root); away from the underflow of exp(-a^2) the same value to ~1 ulp is one exp of the
    // rounded product times a first-order correction with the exact rounding residual (fma).
    if (fabs(a) < 26.0) {
        const double s = -0.5 * a * a;
        const double rem = fma(-0.5 * a, a, -s);
        y = y * (exp(s) * (1.0 + rem));
    } else
#endif
    {
        z = expx2(a, -1);
        y = y * sqrt(z);
    }
    return y;
}
FPT_HD double ndtr_tail(double a) {
    const double y = ndtr_tail_y(a);
    return a * kSqrtH > 0 ? 1.0 - y : y;
}
FPT_HD double ndtr(double a) { return ndtr_is_central(a) ? ndtr_central(a) : ndtr_tail(a); }
// ndtr(a) for a > 0 is base + t rounded once, base = 0.5 (central branch) or 1.0 (tail): t, the
// addend before that last rounding -- it resolves a far finer than the sum does, which is what the
// threshold search of the empirical-FDR kernel uses (ndtr_threshold_open)
FPT_HD double ndtr_addend_pos(double a, double &base, double *ec = nullptr) {
    if (ndtr_is_central(a)) {
        base = 0.5;
        return 0.5 * erf_small(a * kSqrtH);
    }
    base = 1.0;
    return -ndtr_tail_y(a, ec);
}

// One-formula normal cdf for the Stouffer windows of the fused scan (windowing.h:61-66 ends in
// hcephes_ndtr).  ndtr.c evaluates erf for |a| < sqrt(2) and exp(-a^2/2) erfce for the rest, and a
// wavefront of window sums always holds both kinds, so it pays for both branches (~140 vector
// instructions).  Here, for t = |a| < 26,
//     Phi(-t) = exp(-t^2/2) * g(t),   g(t) = Phi(-t) exp(t^2/2)   (half the scaled complementary
//                                                                   error function, smooth, ~1/t)
// with g a degree-14 polynomial in 1/(t+5) (shifted to the centre of its range) and exp as
// 2^n 2^f, 2^f a degree-8 polynomial on |f| <= 1/2: ~39 instructions, no division, no branch.  Coefficients: tools/fit_ndtr_fast.py (Chebyshev series of the functions in
// 60-digit arithmetic, truncated); measured against 60-digit values: relative error <= 4.5e-12
// over |a| < 26, against the contract of 1e-6 (degrees 17 / 10 give 2.5e-13 for six more
// instructions per evaluation; the scan evaluates this five times per base).  |a| >= 26 (where the
// reference's exp(-a^2) leaves the normal range and its value degrades, ndtr.c:49 / expx2.c),
// infinities and NaN take ndtr() above, unchanged.
constexpr double kNdtrFastLimit = 26.0;
// g as a polynomial in v = 1/(t+5) - kNdtrR0 (highest power first) and 2^f on |f| <= 1/2
#define FPT_NDTR_G_N 14
#define FPT_NDTR_E_N 8
#define FPT_NDTR_G_LIST                                                                                    \
    -6.58660840143728033e+07, -5.92359197785998415e+06, 7.06444012539418600e+06, 1.31713778471659194e+06,  \
        -4.95547112766888516e+05, -2.26485841202956974e+05, -4.44627700314815502e+03,                      \
        2.50633366749226661e+04, 1.17812450160460648e+04, 3.34532890951772060e+03,                         \
        7.06038084514002662e+02, 1.18661330011759532e+02, 1.63755034072666597e+01,                         \
        1.88100183566173196e+00, 1.03451588219912849e-01
#define FPT_NDTR_E_LIST                                                                                    \
    1.32596462502704512e-06, 1.53100811431381520e-05, 1.54034337462241957e-04, 1.33334505603844656e-03,   \
        9.61812919394216745e-03, 5.55041094121545286e-02, 2.40226506956402991e-01,                         \
        6.93147180545930497e-01, 1.00000000000001354e+00
constexpr double kNdtrR0 = 0.11612903225806452;          // centre of 1/(t+5) over t in [0, 26]
constexpr double kNdtrNegHalfLog2e = -0.7213475204444817;  // -0.5 * log2(e)
FPT_HD double ndtr_fast(double a) {
    const double kG[FPT_NDTR_G_N + 1] = {FPT_NDTR_G_LIST};
    const double kE[FPT_NDTR_E_N + 1] = {FPT_NDTR_E_LIST};
    const double t = fabs(a);
    const double d = t + 5.0;
#if defined(__HIP_DEVICE_COMPILE__)
    double r = __builtin_amdgcn_rcp(d);  // v_rcp_f64 is good to 2^-24 (measured 4.6e-8): one
    r = fma(fma(-d, r, 1.0), r, r);      // Newton step leaves 2.2e-15, which g passes on damped
#else
    const double r = 1.0 / d;
#endif
    const double g = horner<FPT_NDTR_G_N>(r - kNdtrR0, kG);
    const double t2 = t * t;
    // exp(-t^2/2) = 2^q, q = t^2 (-log2(e) / 2) = n + f: f = q - n is exact, and the rounding of q (1e-13
    // absolute at t = 26) is far below the polynomial's error -- no reduction by ln 2 in two parts
    const double q = t2 * kNdtrNegHalfLog2e;
    const double n = rint(q);
    const double e = horner<FPT_NDTR_E_N>(q - n, kE);
    const double y = ldexp(e * g, (int)n);
    return a > 0.0 ? 1.0 - y : y;
}
// The same formula with g read from a TABLE (fpt_ndtr_gtab.hpp, tools/fit_ndtr_gtab.py): 256 intervals of
// equal width in x = 1/(t + 5), a cubic in the position w inside the interval each -- a multiply-add
// for the slot, a truncation, a fraction and three multiply-adds in place of the 14-step Horner chain
// (fp64 instructions run at half the rate of the others on gfx950, tools/micro/issue.hip).  `tab`:
// FPT_NDTR_GTAB_N x (c3, c2, c1, c0).  Relative error of g 1.2e-11, of the whole 1.4e-11 (measured).
// Arguments beyond the range index a clamped slot: a > 26 still gives exactly 1 (2^n underflows),
// a < -26 is handed on by the callers as before.
FPT_HD double ndtr_fast_tab(double a, const double *tab) {
    const double kE[FPT_NDTR_E_N + 1] = {FPT_NDTR_E_LIST};
    const double t = fabs(a);
    const double d = t + 5.0;
#if defined(__HIP_DEVICE_COMPILE__)
    double r = __builtin_amdgcn_rcp(d);
    r = fma(fma(-d, r, 1.0), r, r);
#else
    const double r = 1.0 / d;
#endif
    const double kf = fma(r, FPT_NDTR_GTAB_SCALE, -(FPT_NDTR_GTAB_XLO * FPT_NDTR_GTAB_SCALE));
    int k = (int)kf;
    k = k < 0 ? 0 : (k > FPT_NDTR_GTAB_N - 1 ? FPT_NDTR_GTAB_N - 1 : k);
    const double w = kf - floor(kf);
    const double *c = tab + 4 * k;
    const double g = fma(fma(fma(c[0], w, c[1]), w, c[2]), w, c[3]);
    const double q = (t * t) * kNdtrNegHalfLog2e;
    const double n = rint(q);
    const double e = horner<FPT_NDTR_E_N>(q - n, kE);
    const double y = ldexp(e * g, (int)n);
    return a > 0.0 ? 1.0 - y : y;
}
// what phase E of the fused scan calls: the fast form inside its range, the restated ndtr.c beyond
FPT_HD double ndtr_window(double a) { return fabs(a) < kNdtrFastLimit ? ndtr_fast(a) : ndtr(a); }

// ndtri.c:48-88
FPT_HD double ndtri(double y0) {
    const double kP0[5] = {-5.99633501014107895267E1, 9.80010754185999661536E1,
                           -5.66762857469070293439E1, 1.39312609387279679503E1,
                           -1.23916583867381258016E0};
    const double kQ0[8] = {1.95448858338141759834E0,  4.67627912898881538453E0,
                           8.63602421390890590575E1,  -2.25462687854119370527E2,
                           2.00260212380060660359E2,  -8.20372256168333339912E1,
                           1.59056225126211695515E1,  -1.18331621121330003142E0};
    const double kP1[9] = {4.05544892305962419923E0,   3.15251094599893866154E1,
                           5.71628192246421288162E1,   4.40805073893200834700E1,
                           1.46849561928858024014E1,   2.18663306850790267539E0,
                           -1.40256079171354495875E-1, -3.50424626827848203418E-2,
                           -8.57456785154685413611E-4};
    const double kQ1[8] = {1.57799883256466749731E1,   4.53907635128879210584E1,
                           4.13172038254672030440E1,   1.50425385692907503408E1,
                           2.50464946208309415979E0,   -1.42182922854787788574E-1,
                           -3.80806407691578277194E-2, -9.33259480895457427372E-4};
    const double kP2[9] = {3.23774891776946035970E0,  6.91522889068984211695E0,
                           3.93881025292474443415E0,  1.33303460815807542389E0,
                           2.01485389549179081538E-1, 1.23716634817820021358E-2,
                           3.01581553508235416007E-4, 2.65806974686737550832E-6,
                           6.23974539184983293730E-9};
    const double kQ2[8] = {6.02427039364742014255E0,  3.67983563856160859403E0,
                           1.37702099489081330271E0,  2.16236993594496635890E-1,
                           1.34204006088543189037E-2, 3.28014464682127739104E-4,
                           2.89247864745380683936E-6, 6.79019408009981274425E-9};
    constexpr double kExpM2 = 0.13533528323661269189;
    if (y0 <= 0.0) return -kInf;
    if (y0 >= 1.0) return kInf;
    bool negate = true;
    double y = y0;
    if (y > (1.0 - kExpM2)) {
        y = 1.0 - y;
        negate = false;
    }
    if (y > kExpM2) {
        y = y - 0.5;
        double y2 = y * y;
        double x = y + y * (y2 * horner<4>(y2, kP0) / horner1<8>(y2, kQ0));
        return x * kSqrt2Pi;
    }
    double x = sqrt(-2.0 * log(y));
    double x0 = x - log(x) / x;
    double z = 1.0 / x;
    double x1 = (x < 8.0) ? z * horner<8>(z, kP1) / horner1<8>(z, kQ1)
                          : z * horner<8>(z, kP2) / horner1<8>(z, kQ2);
    x = x0 - x1;
    return negate ? -x : x;
}

// cprob/unity.c:29-37
FPT_HD double log1p_fn(double x) {
    const double kLP[7] = {4.5270000862445199635215E-5, 4.9854102823193375972212E-1,
                           6.5787325942061044846969E0,  2.9911919328553073277375E1,
                           6.0949667980987787057556E1,  5.7112963590585538103336E1,
                           2.0039553499201281259648E1};
    const double kLQ[6] = {1.5062909083469192043167E1, 8.3047565967967209469434E1,
                           2.2176239823732856465394E2, 3.0909872225312059774938E2,
                           2.1642788614495947685003E2, 6.0118660497603843919306E1};
    double z = 1.0 + x;
    if ((z < 0.70710678118654752440) || (z > 1.41421356237309504880)) return log(z);
    z = x * x;
    z = -0.5 * z + x * (z * horner<6>(x, kLP) / horner1<6>(x, kLQ));
    return x + z;
}

// ---------------------------------------------------------------------------
// A short natural logarithm for the posterior kernel's likelihoods (not a reference function: the
// reference calls libm's log, 92 vector instructions in the device library; this is ~35).  The
// published fdlibm algorithm (e_log.c): x = 2^k m, m in [sqrt(1/2), sqrt(2)), f = m - 1,
// s = f / (2 + f), log(1 + f) = f - f^2/2 + s (f^2/2 + R(s^2)) with R the degree-14 even polynomial of
// that file -- without its special cases: x MUST be positive, finite and normal.  The quotient s comes
// from a reciprocal refined twice; the result is within 2 ulp (tests: test_log_fast).
// log1p_unit_fast(x), 0 <= x <= 1: log(u) + (x - (u - 1)) / u with u = 1 + x rounded (the correction term
// is the rounding error of u, ~1e-16: an unrefined reciprocal is enough for it).
// ---------------------------------------------------------------------------
FPT_HD double log_pos_fast(double x) {
    const double kLg1 = 6.666666666666735130e-01, kLg2 = 3.999999999940941908e-01, kLg3 = 2.857142874366239149e-01,
                 kLg4 = 2.222219843214978396e-01, kLg5 = 1.818357216161805012e-01, kLg6 = 1.531383769920937332e-01,
                 kLg7 = 1.479819860511658591e-01;
    const double kLn2Hi = 6.93147180369123816490e-01, kLn2Lo = 1.90821492927058770002e-10;
#if defined(__HIP_DEVICE_COMPILE__)
    int e = __builtin_amdgcn_frexp_exp(x);
    double m = __builtin_amdgcn_frexp_mant(x);  // [0.5, 1)
#else
    int e;
    double m = frexp(x, &e);
#endif
    const bool low = m < 0.70710678118654752440;
    m = low ? m + m : m;
    e -= low ? 1 : 0;
    const double f = m - 1.0, d = 2.0 + f;
#if defined(__HIP_DEVICE_COMPILE__)
    double inv = __builtin_amdgcn_rcp(d);
    inv = fma(fma(-d, inv, 1.0), inv, inv);
    inv = fma(fma(-d, inv, 1.0), inv, inv);
#else
    const double inv = 1.0 / d;
#endif
    const double s = f * inv, z = s * s, w = z * z;
    const double t1 = w * fma(w, fma(w, kLg6, kLg4), kLg2);
    const double t2 = z * fma(w, fma(w, fma(w, kLg7, kLg5), kLg3), kLg1);
    const double R = t2 + t1, hfsq = 0.5 * f * f, dk = (double)e;
    return dk * kLn2Hi - ((hfsq - fma(s, hfsq + R, dk * kLn2Lo)) - f);
}
FPT_HD double log1p_unit_fast(double x) {
    const double u = 1.0 + x, c = x - (u - 1.0);
#if defined(__HIP_DEVICE_COMPILE__)
    return fma(c, __builtin_amdgcn_rcp(u), log_pos_fast(u));
#else
    return log_pos_fast(u) + c / u;
#endif
}

// ---------------------------------------------------------------------------
// incomplete gamma for Fisher's method (cprob/igam.c, cprob/chdtr.c)
// ---------------------------------------------------------------------------

FPT_HD double igam_series(double a, double x) {  // igam.c:86-100 after the guards
    double ax = a * log(x) - x - lgam(a);
    if (ax < -kMaxLog) return 0.0;
    ax = exp(ax);
    double r = a, c = 1.0, ans = 1.0;
    do {
        r += 1.0;
        c *= x / r;
        ans += c;
    } while (c / ans > kMachEp);
    return ans * ax / a;
}

FPT_HD double igamc_contfrac(double a, double x) {  // igam.c:16-63 after the guards
    double ax = a * log(x) - x - lgam(a);
    if (ax < -kMaxLog) return 0.0;
    ax = exp(ax);
    double y = 1.0 - a;
    double z = x + y + 1.0;
    double c = 0.0;
    double pkm2 = 1.0, qkm2 = x, pkm1 = x + 1.0, qkm1 = z * x;
    double ans = pkm1 / qkm1, t;
    do {
        c += 1.0;
        y += 1.0;
        z += 2.0;
        double yc = y * c;
        double pk = pkm1 * z - pkm2 * yc;
        double qk = qkm1 * z - qkm2 * yc;
        if (qk != 0) {
            double r = pk / qk;
            t = fabs((ans - r) / r);
            ans = r;
        } else {
            t = 1.0;
        }
        pkm2 = pkm1;
        pkm1 = pk;
        qkm2 = qkm1;
        qkm1 = qk;
        if (fabs(pk) > kBig) {
            pkm2 *= kBigInv;
            pkm1 *= kBigInv;
            qkm2 *= kBigInv;
            qkm1 *= kBigInv;
        }
    } while (t > kMachEp);
    return ans * ax;
}

// igam.c:6-14: igamc; igam.c:76-84: igam.  The mutual recursion of the
// reference is one level deep, so it is flattened here.
FPT_HD double igamc(double a, double x) {
    if ((x <= 0) || (a <= 0)) return 1.0;
    if ((x < 1.0) || (x < a)) return 1.0 - igam_series(a, x);
    return igamc_contfrac(a, x);
}

FPT_HD double igam(double a, double x) {
    if ((x <= 0) || (a <= 0)) return 0.0;
    if ((x > 1.0) && (x > a)) return 1.0 - igamc_contfrac(a, x);
    return igam_series(a, x);
}

// chdtr.c:3-10
FPT_HD double chdtrc(double df, double x) {
    if ((x < 0.0) || (df < 1.0)) return 0.0;
    return igamc(df / 2.0, x / 2.0);
}

// ---------------------------------------------------------------------------
// negative binomial scalars (footprint_tools/stats/distributions/nbinom.pyx:82-138)
// and the dispersion fits (footprint_tools/modeling/dispersion.pyx:26-57,127-163)
// ---------------------------------------------------------------------------

// `<int>obs[i]` (dispersion.pyx:314) as x86-64 evaluates it: NaN / out of range -> INT_MIN
FPT_HD int32_t c_int(double v) {
    if (!(v > -2147483649.0 && v < 2147483648.0)) return INT32_MIN;
    return (int32_t)v;
}

FPT_HD int32_t wrap_inc(int32_t k) { return (int32_t)((uint32_t)k + 1u); }

FPT_HD double nb_cdf(int32_t k, double p, double r) {  // nbinom.pyx:138
    return incbet(r, (double)wrap_inc(k), p);
}

FPT_HD double nb_logpmf(int32_t k, double p, double r) {  // nbinom.pyx:99-100
    double lg[3];
    FPT_NOUNROLL
    for (int i = 0; i < 3; ++i)
        lg[i] = lgam(i == 0 ? (double)k + r : (i == 1 ? (double)wrap_inc(k) : r));
    double coeff = lg[0] - lg[1] - lg[2];
    return coeff + r * log(p) + (double)k * log1p_fn(-p);
}

// Piecewise-linear fits.  The reference evaluates sum_s mask_s * (y_s + k_s x)
// in Python float arithmetic (dispersion.pyx:26-57), so a non-finite term of an
// inactive segment still poisons the sum (0 * inf = NaN); reproduced here.
template <int NSEG>
FPT_HD double piecewise(const double *par, double x) {
#if defined(__clang__)
#pragma clang fp contract(off)  // y + k*x is two roundings in the reference (an fma would miss
#endif                          // the exact zeros that raise ZeroDivisionError there)
    double acc = 0.0;
#pragma unroll
    for (int s = 0; s < NSEG; ++s) {
        bool in;
        if (s == 0) in = x < par[0];
        else if (s == NSEG - 1) in = x >= par[s - 1];
        else in = (x >= par[s - 1]) && (x < par[s]);
        double v = par[NSEG + s] + par[2 * NSEG + s] * x;
        double term = (in ? 1.0 : 0.0) * v;
        acc = (s == 0) ? term : acc + term;
    }
    return acc;
}

FPT_HD double fit_mu(const double *mu9, double x) {  // dispersion.pyx:127-144
    double res = piecewise<3>(mu9, x);
    return res > 0.0 ? res : 0.1;
}

// dispersion.pyx:146-163; *zero_div is set when the reference would raise ZeroDivisionError
FPT_HD double fit_r(const double *r15, double x, bool *zero_div) {
    double v = piecewise<5>(r15, x);
    if (v == 0.0) {
        *zero_div = true;
        return NAN;
    }
    double res = 1.0 / v;
    return res > 0.0 ? res : 1e-6;
}

// ---------------------------------------------------------------------------
// The 32-bit word of a caller-supplied uniform of the null sampler: floor(u 2^32) inside the table,
// so that u = (w + 1/2) 2^-32 gives back w (NaN and u < 0: 0).
// ---------------------------------------------------------------------------
FPT_HD uint32_t uniform_word(double u) {
    const double x = u * 4294967296.0;
    return !(x > 0.0) ? 0u : (x >= 4294967295.0 ? 0xffffffffu : (uint32_t)x);
}

}  // namespace fptm
