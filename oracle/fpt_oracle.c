/*
 * fpt_oracle.c -- TEST INFRASTRUCTURE ONLY (see fpt_oracle.h).
 *
 * CPU restatement of the reference algorithm, written from the behaviour of the
 * files cited per function (all under /root/reference).  Floating-point
 * operation ORDER follows the reference so that results are bit-identical to
 * the reference's own C build on the same libm (checked by tests/test_oracle_*).
 * Build: gcc -O2 -ffp-contract=off (no FMA contraction, like the reference's
 * distutils build).
 */
#include "fpt_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* hcephes/include/hcephes.h:75-84 */
#define K_MACHEP 1.11022302462515654042E-16
#define K_MAXLOG 7.09782712893383996732E2
#define K_MINLOG -7.451332191019412076235E2
#define K_PI 3.14159265358979323846
#define K_SQRTH 7.07106781186547524401E-1
/* hcephes/src/cprob/incbet.c:3, gamma.c:18 */
#define K_MAXGAM 171.624376956302725
/* hcephes/src/cprob/incbet.c:5-6, igam.c:3-4 */
#define K_BIG 4.503599627370496e15
#define K_BIGINV 2.22044604925031308085e-16

/* ------------------------------------------------------------------ polynomials */

/* polevl.c:3-17: Horner over n+1 coefficients, highest degree first. */
double orc_polevl(double x, const double *c, int n) {
    double acc = c[0];
    for (int i = 1; i <= n; i++) acc = acc * x + c[i];
    return acc;
}

/* polevl.c:19-33: same with an implied leading coefficient of 1 (n coefficients). */
double orc_p1evl(double x, const double *c, int n) {
    double acc = x + c[0];
    for (int i = 1; i < n; i++) acc = acc * x + c[i];
    return acc;
}

/* ------------------------------------------------------------------ gamma / lgam */

/* gamma.c:10-17 */
static const double GAM_P[7] = {1.60119522476751861407E-4, 1.19135147006586384913E-3,
                                1.04213797561761569935E-2, 4.76367800457137231464E-2,
                                2.07448227648435975150E-1, 4.94214826801497100753E-1,
                                9.99999999999999996796E-1};
static const double GAM_Q[8] = {-2.31581873324120129819E-5, 5.39605580493303397842E-4,
                                -4.45641913851797240494E-3, 1.18139785222060435552E-2,
                                3.58236398605498653373E-2,  -2.34591795718243348568E-1,
                                7.14304917030273074085E-2,  1.00000000000000000320E0};
/* gamma.c:22-27 */
static const double GAM_STIR[5] = {7.87311395793093628397E-4, -2.29549961613378126380E-4,
                                   -2.68132617805781232825E-3, 3.47222221605458667310E-3,
                                   8.33333333333482257126E-2};
#define K_MAXSTIR 143.01608
#define K_SQTPI 2.50662827463100050242E0
#define K_LOGPI 1.14472988584940017414
#define K_LS2PI 0.91893853320467274178
#define K_MAXLGM 2.556348e305

/* gamma.c:35-49: Stirling, valid for 33 <= x <= 172 */
static double stirling_gamma(double x) {
    double w = 1.0 / x;
    w = 1.0 + w * orc_polevl(w, GAM_STIR, 4);
    double y = exp(x);
    if (x > K_MAXSTIR) { /* two-step pow to dodge overflow */
        double v = pow(x, 0.5 * x - 0.25);
        y = v * (v / y);
    } else {
        y = pow(x, x - 0.5) / y;
    }
    return K_SQTPI * y * w;
}

/* gamma.c:51-127.  The reference also sets a global sign (sgngam); callers on
 * this path never read it, so only the value is restated. */
double orc_gamma(double x) {
    int sgn = 1;
    if (isnan(x)) return x;
    if (x == HUGE_VAL) return x;
    if (x == -HUGE_VAL) return NAN;
    double q = fabs(x);

    if (q > 33.0) {
        double z;
        if (x < 0.0) {
            double p = floor(q);
            if (p == q) return NAN; /* :66-69 pole */
            int i = (int)p;
            if ((i & 1) == 0) sgn = -1;
            z = q - p;
            if (z > 0.5) {
                p += 1.0;
                z = q - p;
            }
            z = q * sin(K_PI * z);
            if (z == 0.0) return sgn * HUGE_VAL;
            z = fabs(z);
            z = K_PI / (z * stirling_gamma(q));
        } else {
            z = stirling_gamma(x);
        }
        return sgn * z;
    }

    double z = 1.0;
    while (x >= 3.0) {
        x -= 1.0;
        z *= x;
    }
    while (x < 0.0) {
        if (x > -1.E-9) goto tiny;
        z /= x;
        x += 1.0;
    }
    while (x < 2.0) {
        if (x < 1.e-9) goto tiny;
        z /= x;
        x += 1.0;
    }
    if (x == 2.0) return z;
    x -= 2.0;
    {
        double p = orc_polevl(x, GAM_P, 6);
        double qq = orc_polevl(x, GAM_Q, 7);
        return z * p / qq;
    }
tiny: /* :119-126 */
    if (x == 0.0) return NAN;
    return z / ((1.0 + 0.5772156649015329 * x) * x);
}

/* gamma.c:132-143 */
static const double LGAM_A[5] = {8.11614167470508450300E-4, -5.95061904284301438324E-4,
                                 7.93650340457716943945E-4, -2.77777777730099687205E-3,
                                 8.33333333333331927722E-2};
static const double LGAM_B[6] = {-1.37825152569120859100E3, -3.88016315134637840924E4,
                                 -3.31612992738871184744E5, -1.16237097492762307383E6,
                                 -1.72173700820839662146E6, -8.53555664245765465627E5};
static const double LGAM_C[6] = {-3.51815701436523470549E2, -1.70642106651881159223E4,
                                 -2.20528590553854454839E5, -1.13933444367982507207E6,
                                 -2.53252307177582951285E6, -2.01889141433532773231E6};

/* gamma.c:152-235 (lgam_sgn; sign output dropped, unused on the path) */
double orc_lgam(double x) {
    if (isnan(x)) return x;
    if (!isfinite(x)) return HUGE_VAL;

    if (x < -34.0) { /* reflection :163-188 */
        double q = -x;
        double w = orc_lgam(q);
        double p = floor(q);
        if (p == q) return HUGE_VAL;
        double z = q - p;
        if (z > 0.5) {
            p += 1.0;
            z = p - q;
        }
        z = q * sin(K_PI * z);
        if (z == 0.0) return HUGE_VAL;
        return K_LOGPI - log(z) - w;
    }

    if (x < 13.0) { /* :190-217 */
        double z = 1.0, p = 0.0, u = x;
        while (u >= 3.0) {
            p -= 1.0;
            u = x + p;
            z *= u;
        }
        while (u < 2.0) {
            if (u == 0.0) return HUGE_VAL;
            z /= u;
            p += 1.0;
            u = x + p;
        }
        if (z < 0.0) z = -z;
        if (u == 2.0) return log(z);
        p -= 2.0;
        x = x + p;
        p = x * orc_polevl(x, LGAM_B, 5) / orc_p1evl(x, LGAM_C, 6);
        return log(z) + p;
    }

    if (x > K_MAXLGM) return HUGE_VAL; /* :219-221 (sign is +1 here) */

    double q = (x - 0.5) * log(x) - x + K_LS2PI;
    if (x > 1.0e8) return q;
    double p = 1.0 / (x * x);
    if (x >= 1000.0)
        q += ((7.9365079365079365079365e-4 * p - 2.7777777777777777777778e-3) * p +
              0.0833333333333333333333) /
             x;
    else
        q += orc_polevl(p, LGAM_A, 4) / x;
    return q;
}

/* ------------------------------------------------------------------ incomplete beta */

/* incbet.c:266-299 power series */
static double ibeta_pseries(double a, double b, double x) {
    double ai = 1.0 / a;
    double u = (1.0 - b) * x;
    double v = u / (a + 1.0);
    double t1 = v;
    double t = u;
    double n = 2.0;
    double s = 0.0;
    double z = K_MACHEP * ai;
    while (fabs(v) > z) {
        u = (n - b) * x / n;
        t *= u;
        v = t / (a + n);
        s += v;
        n += 1.0;
    }
    s += t1;
    s += ai;

    u = a * log(x);
    if ((a + b) < K_MAXGAM && fabs(u) < K_MAXLOG) {
        t = orc_gamma(a + b) / (orc_gamma(a) * orc_gamma(b));
        s = s * t * pow(x, a);
    } else {
        t = orc_lgam(a + b) - orc_lgam(a) - orc_lgam(b) + u + log(s);
        s = (t < K_MINLOG) ? 0.0 : exp(t);
    }
    return s;
}

/* incbet.c:100-177 (which=1, variable x) and :183-261 (which=2, variable x/(1-x)).
 * The two continued fractions share one recurrence; they differ in the start
 * values of k2/k6 and in the direction those two step. */
static double ibeta_contfrac(int which, double a, double b, double x) {
    double k1 = a, k3 = a, k4 = a + 1.0, k5 = 1.0, k7 = a + 1.0, k8 = a + 2.0;
    double k2, k6, step2, step6, v;
    if (which == 1) {
        k2 = a + b;
        k6 = b - 1.0;
        step2 = 1.0;
        step6 = -1.0;
        v = x;
    } else {
        k2 = b - 1.0;
        k6 = a + b;
        step2 = -1.0;
        step6 = 1.0;
        v = x / (1.0 - x);
    }
    double pkm2 = 0.0, qkm2 = 1.0, pkm1 = 1.0, qkm1 = 1.0;
    double ans = 1.0, r = 1.0, t;
    const double thresh = 3.0 * K_MACHEP;
    int n = 0;
    do {
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
        if (r != 0) {
            t = fabs((ans - r) / r);
            ans = r;
        } else {
            t = 1.0;
        }
        if (t < thresh) break;

        k1 += 1.0;
        k2 += step2;
        k3 += 2.0;
        k4 += 2.0;
        k5 += 1.0;
        k6 += step6;
        k7 += 2.0;
        k8 += 2.0;

        if ((fabs(qk) + fabs(pk)) > K_BIG) {
            pkm2 *= K_BIGINV;
            pkm1 *= K_BIGINV;
            qkm2 *= K_BIGINV;
            qkm1 *= K_BIGINV;
        }
        if ((fabs(qk) < K_BIGINV) || (fabs(pk) < K_BIGINV)) {
            pkm2 *= K_BIG;
            pkm1 *= K_BIG;
            qkm2 *= K_BIG;
            qkm1 *= K_BIG;
        }
    } while (++n < 300);
    return ans;
}

/* incbet.c:12-94 */
double orc_incbet(double aa, double bb, double xx) {
    if (aa <= 0.0 || bb <= 0.0) return 0.0; /* domain */
    if (xx <= 0.0 || xx >= 1.0) {
        if (xx == 0.0) return 0.0;
        if (xx == 1.0) return 1.0;
        return 0.0; /* domain (also NaN falls through the reference's tests to here? no: see below) */
    }
    /* NaN xx: both comparisons above are false in the reference too, so it
     * proceeds into the series exactly as the code below does. */
    double a, b, x, xc, w, y, t;
    int flipped = 0;

    if ((bb * xx) <= 1.0 && xx <= 0.95) return ibeta_pseries(aa, bb, xx);

    w = 1.0 - xx;
    if (xx > (aa / (aa + bb))) { /* past the mean: evaluate the other tail */
        flipped = 1;
        a = bb;
        b = aa;
        xc = xx;
        x = w;
    } else {
        a = aa;
        b = bb;
        xc = w;
        x = xx;
    }

    if (flipped && (b * x) <= 1.0 && x <= 0.95) {
        t = ibeta_pseries(a, b, x);
        goto finish;
    }

    y = x * (a + b - 2.0) - (a - 1.0);
    if (y < 0.0)
        w = ibeta_contfrac(1, a, b, x);
    else
        w = ibeta_contfrac(2, a, b, x) / xc;

    y = a * log(x);
    t = b * log(xc);
    if ((a + b) < K_MAXGAM && fabs(y) < K_MAXLOG && fabs(t) < K_MAXLOG) {
        t = pow(xc, b);
        t *= pow(x, a);
        t /= a;
        t *= w;
        t *= orc_gamma(a + b) / (orc_gamma(a) * orc_gamma(b));
        goto finish;
    }
    y += t + orc_lgam(a + b) - orc_lgam(a) - orc_lgam(b);
    y += log(w / a);
    t = (y < K_MINLOG) ? 0.0 : exp(y);

finish:
    if (flipped) {
        if (t <= K_MACHEP)
            t = 1.0 - K_MACHEP;
        else
            t = 1.0 - t;
    }
    return t;
}

/* ------------------------------------------------------------------ normal cdf / quantile */

/* ndtr.c:8-31 */
static const double ERF_P[9] = {2.46196981473530512524E-10, 5.64189564831068821977E-1,
                                7.46321056442269912687E0,   4.86371970985681366614E1,
                                1.96520832956077098242E2,   5.26445194995477358631E2,
                                9.34528527171957607540E2,   1.02755188689515710272E3,
                                5.57535335369399327526E2};
static const double ERF_Q[8] = {1.32281951154744992508E1, 8.67072140885989742329E1,
                                3.54937778887819891062E2, 9.75708501743205489753E2,
                                1.82390916687909736289E3, 2.24633760818710981792E3,
                                1.65666309194161350182E3, 5.57535340817727675546E2};
static const double ERF_R[6] = {5.64189583547755073984E-1, 1.27536670759978104416E0,
                                5.01905042251180477414E0,  6.16021097993053585195E0,
                                7.40974269950448939160E0,  2.97886665372100240670E0};
static const double ERF_S[6] = {2.26052863220117276590E0, 9.39603524938001434673E0,
                                1.20489539808096656605E1, 1.70814450747565897222E1,
                                9.60896809063285878198E0, 3.36907645100081516050E0};
static const double ERF_T[5] = {9.60497373987051638749E0, 9.00260197203842689217E1,
                                2.23200534594684319226E3, 7.00332514112805075473E3,
                                5.55923013010394962768E4};
static const double ERF_U[5] = {3.35617141647503099647E1, 5.21357949780152679795E2,
                                4.59432382970980127987E3, 2.26290000613890934246E4,
                                4.92673942608635921086E4};

/* expx2.c:6-34: exp(+-x*x) with x split at 1/128 */
double orc_expx2(double x, int sign) {
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
    if ((u + u1) > K_MAXLOG) return HUGE_VAL;
    return exp(u) * exp(u1);
}

/* ndtr.c:65-77: exp(x^2) erfc(x), x > 1 */
double orc_erfce(double x) {
    double p, q;
    if (x < 8.0) {
        p = orc_polevl(x, ERF_P, 8);
        q = orc_p1evl(x, ERF_Q, 8);
    } else {
        p = orc_polevl(x, ERF_R, 5);
        q = orc_p1evl(x, ERF_S, 6);
    }
    return p / q;
}

/* ndtr.c:79-87 */
double orc_erf(double x) {
    if (fabs(x) > 1.0) return 1.0 - orc_erfc(x);
    double z = x * x;
    return x * orc_polevl(z, ERF_T, 4) / orc_p1evl(z, ERF_U, 5);
}

/* ndtr.c:89-132 */
double orc_erfc(double a) {
    double x = (a < 0.0) ? -a : a;
    if (x < 1.0) return 1.0 - orc_erf(a);
    double z = -a * a;
    if (z < -K_MAXLOG) return (a < 0) ? 2.0 : 0.0;
    z = orc_expx2(a, -1);
    double p, q;
    if (x < 8.0) {
        p = orc_polevl(x, ERF_P, 8);
        q = orc_p1evl(x, ERF_Q, 8);
    } else {
        p = orc_polevl(x, ERF_R, 5);
        q = orc_p1evl(x, ERF_S, 6);
    }
    double y = (z * p) / q;
    if (a < 0) y = 2.0 - y;
    if (y == 0.0) return (a < 0) ? 2.0 : 0.0;
    return y;
}

/* ndtr.c:34-59 (USE_EXPXSQ branch) */
double orc_ndtr(double a) {
    double x = a * K_SQRTH;
    double z = fabs(x);
    double y;
    if (z < 1.0) {
        y = 0.5 + 0.5 * orc_erf(x);
    } else {
        y = 0.5 * orc_erfce(z);
        z = orc_expx2(a, -1);
        y = y * sqrt(z);
        if (x > 0) y = 1.0 - y;
    }
    return y;
}

/* ndtri.c:6-46 */
static const double NDI_P0[5] = {-5.99633501014107895267E1, 9.80010754185999661536E1,
                                 -5.66762857469070293439E1, 1.39312609387279679503E1,
                                 -1.23916583867381258016E0};
static const double NDI_Q0[8] = {1.95448858338141759834E0,  4.67627912898881538453E0,
                                 8.63602421390890590575E1,  -2.25462687854119370527E2,
                                 2.00260212380060660359E2,  -8.20372256168333339912E1,
                                 1.59056225126211695515E1,  -1.18331621121330003142E0};
static const double NDI_P1[9] = {4.05544892305962419923E0,   3.15251094599893866154E1,
                                 5.71628192246421288162E1,   4.40805073893200834700E1,
                                 1.46849561928858024014E1,   2.18663306850790267539E0,
                                 -1.40256079171354495875E-1, -3.50424626827848203418E-2,
                                 -8.57456785154685413611E-4};
static const double NDI_Q1[8] = {1.57799883256466749731E1,   4.53907635128879210584E1,
                                 4.13172038254672030440E1,   1.50425385692907503408E1,
                                 2.50464946208309415979E0,   -1.42182922854787788574E-1,
                                 -3.80806407691578277194E-2, -9.33259480895457427372E-4};
static const double NDI_P2[9] = {3.23774891776946035970E0,  6.91522889068984211695E0,
                                 3.93881025292474443415E0,  1.33303460815807542389E0,
                                 2.01485389549179081538E-1, 1.23716634817820021358E-2,
                                 3.01581553508235416007E-4, 2.65806974686737550832E-6,
                                 6.23974539184983293730E-9};
static const double NDI_Q2[8] = {6.02427039364742014255E0,  3.67983563856160859403E0,
                                 1.37702099489081330271E0,  2.16236993594496635890E-1,
                                 1.34204006088543189037E-2, 3.28014464682127739104E-4,
                                 2.89247864745380683936E-6, 6.79019408009981274425E-9};
#define K_EXPM2 0.13533528323661269189

/* ndtri.c:48-88 */
double orc_ndtri(double y0) {
    if (y0 <= 0.0) return -HUGE_VAL;
    if (y0 >= 1.0) return HUGE_VAL;
    int negate = 1;
    double y = y0;
    if (y > (1.0 - K_EXPM2)) {
        y = 1.0 - y;
        negate = 0;
    }
    if (y > K_EXPM2) { /* central region */
        y = y - 0.5;
        double y2 = y * y;
        double x = y + y * (y2 * orc_polevl(y2, NDI_P0, 4) / orc_p1evl(y2, NDI_Q0, 8));
        return x * K_SQTPI;
    }
    double x = sqrt(-2.0 * log(y));
    double x0 = x - log(x) / x;
    double z = 1.0 / x;
    double x1;
    if (x < 8.0)
        x1 = z * orc_polevl(z, NDI_P1, 8) / orc_p1evl(z, NDI_Q1, 8);
    else
        x1 = z * orc_polevl(z, NDI_P2, 8) / orc_p1evl(z, NDI_Q2, 8);
    x = x0 - x1;
    if (negate) x = -x;
    return x;
}

/* unity.c:14-25 */
static const double L1P_P[7] = {4.5270000862445199635215E-5, 4.9854102823193375972212E-1,
                                6.5787325942061044846969E0,  2.9911919328553073277375E1,
                                6.0949667980987787057556E1,  5.7112963590585538103336E1,
                                2.0039553499201281259648E1};
static const double L1P_Q[6] = {1.5062909083469192043167E1, 8.3047565967967209469434E1,
                                2.2176239823732856465394E2, 3.0909872225312059774938E2,
                                2.1642788614495947685003E2, 6.0118660497603843919306E1};

/* unity.c:29-37 */
double orc_log1p(double x) {
    double z = 1.0 + x;
    if ((z < 0.70710678118654752440) || (z > 1.41421356237309504880)) return log(z);
    z = x * x;
    z = -0.5 * z + x * (z * orc_polevl(x, L1P_P, 6) / orc_p1evl(x, L1P_Q, 6));
    return x + z;
}

/* ------------------------------------------------------------------ incomplete gamma (Fisher) */

/* igam.c:76-100 */
double orc_igam(double a, double x) {
    if ((x <= 0) || (a <= 0)) return 0.0;
    if ((x > 1.0) && (x > a)) return 1.0 - orc_igamc(a, x);
    double ax = a * log(x) - x - orc_lgam(a);
    if (ax < -K_MAXLOG) return 0.0;
    ax = exp(ax);
    double r = a, c = 1.0, ans = 1.0;
    do {
        r += 1.0;
        c *= x / r;
        ans += c;
    } while (c / ans > K_MACHEP);
    return ans * ax / a;
}

/* igam.c:6-63 */
double orc_igamc(double a, double x) {
    if ((x <= 0) || (a <= 0)) return 1.0;
    if ((x < 1.0) || (x < a)) return 1.0 - orc_igam(a, x);
    double ax = a * log(x) - x - orc_lgam(a);
    if (ax < -K_MAXLOG) return 0.0;
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
        if (fabs(pk) > K_BIG) {
            pkm2 *= K_BIGINV;
            pkm1 *= K_BIGINV;
            qkm2 *= K_BIGINV;
            qkm1 *= K_BIGINV;
        }
    } while (t > K_MACHEP);
    return ans * ax;
}

/* chdtr.c:3-10 */
double orc_chdtrc(double df, double x) {
    if ((x < 0.0) || (df < 1.0)) return 0.0;
    return orc_igamc(df / 2.0, x / 2.0);
}

void orc_map1(int op, const double *x, int64_t n, double *out) {
    for (int64_t i = 0; i < n; i++) {
        double v = x[i], r;
        switch (op) {
        case 0: r = orc_gamma(v); break;
        case 1: r = orc_lgam(v); break;
        case 2: r = orc_ndtr(v); break;
        case 3: r = orc_ndtri(v); break;
        case 4: r = orc_log1p(v); break;
        case 5: r = orc_erf(v); break;
        default: r = orc_erfc(v); break;
        }
        out[i] = r;
    }
}

void orc_incbet_v(const double *a, const double *b, const double *x, int64_t n, double *out) {
    for (int64_t i = 0; i < n; i++) out[i] = orc_incbet(a[i], b[i], x[i]);
}

void orc_chdtrc_v(const double *df, const double *x, int64_t n, double *out) {
    for (int64_t i = 0; i < n; i++) out[i] = orc_chdtrc(df[i], x[i]);
}

/* ------------------------------------------------------------------ 6-mer bias lookup */

/* 2-bit code of a base after `.upper()` (predict.pyx:140); 4 = anything else.
 * Table order: idx = sum code(s_m) * 4^(5-m), A=0 C=1 G=2 T=3. */
static int base_code(uint8_t ch) {
    switch (ch) {
    case 'A': case 'a': return 0;
    case 'C': case 'c': return 1;
    case 'G': case 'g': return 2;
    case 'T': case 't': return 3;
    default: return 4;
    }
}

/* bias.py:101-111: probs(seq)[j] = model[seq[j:j+6]], missing -> 1e-6 (bias.py:16-17);
 * predict.pyx:150-153: '-' strand = probs(reverse_complement(seq))[::-1], i.e.
 * rev[j] = model[revcomp(seq[j+1 .. j+6])]  (predict.pyx:47-61: unknown base -> 'N'). */
void orc_kmer_probs(const uint8_t *seq, int64_t seq_len, const double *table, double dflt,
                    double *fwd, double *rev, int32_t *idx_fwd, int32_t *idx_rev) {
    int64_t l = seq_len - 6;
    for (int64_t j = 0; j < l; j++) {
        int32_t fi = 0, ri = 0;
        for (int m = 0; m < 6; m++) {
            int c = base_code(seq[j + m]);
            fi = (c > 3 || fi < 0) ? -1 : fi * 4 + c;
            /* reverse complement of seq[j+1..j+6]: first letter = comp(seq[j+6]) */
            int d = base_code(seq[j + 6 - m]);
            ri = (d > 3 || ri < 0) ? -1 : ri * 4 + (3 - d);
        }
        if (fwd) fwd[j] = fi < 0 ? dflt : table[fi];
        if (rev) rev[j] = ri < 0 ? dflt : table[ri];
        if (idx_fwd) idx_fwd[j] = fi;
        if (idx_rev) idx_rev[j] = ri;
    }
}

/* ------------------------------------------------------------------ expected cleavage */

#define EXCH(p, q) do { double t_ = (p); (p) = (q); (q) = t_; } while (0)

/* smoothing.h:11-53: Numerical-Recipes selection.  The element order it
 * leaves behind decides the summation order of trimmed_sum, so the
 * partitioning steps are restated one for one. */
static double nr_select(double *v, unsigned int n, unsigned int k) {
    unsigned long lo = 0, hi = n - 1;
    for (;;) {
        if (hi <= lo + 1) {
            if (hi == lo + 1 && v[hi] < v[lo]) EXCH(v[lo], v[hi]);
            return v[k];
        }
        unsigned long mid = (lo + hi) >> 1;
        EXCH(v[mid], v[lo + 1]);
        if (v[lo] > v[hi]) EXCH(v[lo], v[hi]);
        if (v[lo + 1] > v[hi]) EXCH(v[lo + 1], v[hi]);
        if (v[lo] > v[lo + 1]) EXCH(v[lo], v[lo + 1]);
        unsigned long i = lo + 1, j = hi;
        double pivot = v[lo + 1];
        for (;;) {
            do i++; while (v[i] < pivot);
            do j--; while (v[j] > pivot);
            if (j < i) break;
            EXCH(v[i], v[j]);
        }
        v[lo + 1] = v[j];
        v[j] = pivot;
        if (j >= k) hi = j - 1;
        if (j <= k) lo = i;
    }
}

/* smoothing.h:59-70: test order matters when t1 == t2 */
static double trim_weight(double x, double t1, double t2, double w1, double w2) {
    if (x < t2 && x > t1) return x;
    if (x < t1) return 0;
    if (x > t2) return 0;
    if (x == t1) return w1 * x;
    return w2 * x;
}

/* smoothing.h:72-104 */
static double trimmed_mean_window(double *x, int n, int k) {
    double os1 = nr_select(x, n, k);
    double os2 = nr_select(x, n, n - k - 1);
    double b = 0, d = 0, dm = 0, bm = 0;
    for (int i = 0; i < n; i++) {
        double r = x[i];
        if (r < os1) bm += 1;
        else if (r == os1) b += 1;
        if (r < os2) dm += 1;
        else if (r == os2) d += 1;
    }
    double a = b + bm - k;
    double c = n - k - dm;
    double w1 = a / b;
    double w2 = c / d;
    double t = 0;
    for (int i = 0; i < n; i++) t += trim_weight(x[i], os1, os2, w1, w2);
    return t / (n - 2 * k);
}

/* predict.h:23-74 (+ smoothing.h:107-133).  exp_out/win_out have l entries. */
void orc_fast_predict(const double *obs, const double *probs, int l, int hw, int shw,
                      double clip, double *exp_out, double *win_out) {
    double *win_counts = (double *)calloc(l > 0 ? l : 1, sizeof(double));
    double *win_probs = (double *)calloc(l > 0 ? l : 1, sizeof(double));
    for (int i = 0; i < l; i++) exp_out[i] = 0.0;

    /* predict.h:41-48: 2*hw wide, left-biased window [i-hw, i+hw) */
    for (int i = hw; i < l - hw; i++)
        for (int j = -hw; j < hw; j++) {
            win_counts[i] += obs[i + j];
            win_probs[i] += probs[i + j];
        }

    if (shw > 0) { /* predict.h:50-57, smoothing.h:107-133 */
        int w = shw * 2 + 1;
        int k = (int)((double)w * clip);
        double *tmp = (double *)malloc(w * sizeof(double));
        double *sm = (double *)calloc(l > 0 ? l : 1, sizeof(double));
        for (int i = shw; i < l - shw; i++) {
            memcpy(tmp, &win_counts[i - shw], w * sizeof(double));
            sm[i] = trimmed_mean_window(tmp, w, k);
        }
        free(tmp);
        free(win_counts);
        win_counts = sm;
    }

    /* predict.h:60-63 */
    for (int i = hw; i < l - hw; i++)
        exp_out[i] = round((probs[i] / win_probs[i]) * win_counts[i]);

    for (int i = 0; i < l; i++) win_out[i] = win_counts[i];
    free(win_counts);
    free(win_probs);
}

/* ------------------------------------------------------------------ dispersion model */

/* dispersion.pyx:26-34: sum of mask*(y+k*x) terms, Python semantics
 * (False*v == 0.0*v, so non-finite v poisons the sum exactly as in Python). */
static double piecewise_masked(const double *par, int nseg, double x) {
    const double *brk = par, *icpt = par + nseg, *slope = par + 2 * nseg;
    double acc = 0.0;
    for (int s = 0; s < nseg; s++) {
        int in;
        if (s == 0) in = (x < brk[0]);
        else if (s == nseg - 1) in = (x >= brk[s - 1]);
        else in = (x >= brk[s - 1]) && (x < brk[s]);
        double v = icpt[s] + slope[s] * x;
        double term = (in ? 1.0 : 0.0) * v;
        acc = (s == 0) ? term : acc + term;
    }
    return acc;
}

/* dispersion.pyx:127-144 */
double orc_fit_mu(const double *mu_par9, double x) {
    double res = piecewise_masked(mu_par9, 3, x);
    return res > 0.0 ? res : 0.1;
}

/* dispersion.pyx:146-163; Cython raises ZeroDivisionError on 1.0/0.0 (no cdivision) */
int orc_fit_r(const double *r_par15, double x, double *r_out) {
    double v = piecewise_masked(r_par15, 5, x);
    if (v == 0.0) {
        *r_out = NAN;
        return 1;
    }
    double res = 1.0 / v;
    *r_out = res > 0.0 ? res : 1e-6;
    return 0;
}

/* `<int>obs[i]` (dispersion.pyx:314): C double->int conversion; on x86-64
 * cvttsd2si yields INT_MIN for NaN / out-of-range. */
int32_t orc_c_int(double v) {
    if (!(v > -2147483649.0 && v < 2147483648.0)) return INT32_MIN;
    return (int32_t)v;
}

/* nbinom.pyx:99-100 */
double orc_nb_logpmf(int32_t k, double p, double r) {
    int32_t k1 = (int32_t)((uint32_t)k + 1u); /* -fwrapv int add */
    double coeff = orc_lgam(k + r) - orc_lgam((double)k1) - orc_lgam(r);
    return coeff + r * log(p) + k * orc_log1p(-p);
}

/* nbinom.pyx:119 */
double orc_nb_pmf(int32_t k, double p, double r) { return exp(orc_nb_logpmf(k, p, r)); }

/* nbinom.pyx:138 */
double orc_nb_cdf(int32_t k, double p, double r) {
    int32_t k1 = (int32_t)((uint32_t)k + 1u);
    return orc_incbet(r, (double)k1, p);
}

int orc_nb_values(int what, const double *mu_par9, const double *r_par15, const double *exp_,
                  const double *obs, int64_t n, double *out) {
    for (int64_t i = 0; i < n; i++) {
        double r, mu;
        if (orc_fit_r(r_par15, exp_[i], &r)) return 1;
        mu = orc_fit_mu(mu_par9, exp_[i]);
        int32_t k = orc_c_int(obs[i]);
        double p = r / (r + mu);
        if (what == 0) out[i] = orc_nb_cdf(k, p, r);
        else if (what == 1) out[i] = orc_nb_logpmf(k, p, r);
        else out[i] = orc_nb_pmf(k, p, r);
    }
    return 0;
}

/* ------------------------------------------------------------------ sliding windows */

/* windowing.h:11-67, 86-102 applied to one window of k = 2*hw+1 values */
static double reduce_window(int op, const double *x, const double *w, int k) {
    double s = 0.0;
    switch (op) {
    case ORC_WIN_SUM:
        for (int i = 0; i < k; i++) s += x[i];
        return s;
    case ORC_WIN_PRODUCT:
        s = 1.0;
        for (int i = 0; i < k; i++) s *= x[i];
        return s;
    case ORC_WIN_FISHER:
        for (int i = 0; i < k; i++) s += log(x[i]);
        s *= -2.0;
        return orc_chdtrc((double)2.0 * k, s);
    case ORC_WIN_STOUFFER: {
        for (int i = 0; i < k; ++i) s += orc_ndtri(1.0 - x[i]);
        double z = s / sqrt((double)k);
        return orc_ndtr(-z);
    }
    default: { /* weighted Stouffer */
        double sw = 0.0;
        for (int i = 0; i < k; ++i) {
            s += w[i] * orc_ndtri(1.0 - x[i]);
            sw += w[i] * w[i];
        }
        double z = s / sqrt(sw);
        return orc_ndtr(-z);
    }
    }
}

/* windowing.h:69-84,104-122 + windowing.pyx:47-56,146-156: positions outside
 * [hw, n-hw) are 1.0 for every reducer. */
void orc_window(int op, const double *x, const double *w, int n, int hw, double *out) {
    int k = 2 * hw + 1;
    for (int i = 0; i < n; i++) out[i] = 1.0;
    for (int i = hw; i < n - hw; ++i)
        out[i] = reduce_window(op, &x[i - hw], w ? &w[i - hw] : NULL, k);
}

/* ------------------------------------------------------------------ FDR helpers */

/* utils.pyx:52-79 (two-pointer; `lo` persists across i) */
void orc_bisect(const double *a, int na, const double *b, int nb, double *out) {
    int lo = 0, hi = na;
    for (int i = 0; i < nb; i++) {
        while (lo < hi) {
            if (b[i] < a[lo]) break;
            lo = lo + 1;
        }
        out[i] = lo;
    }
}

/* numpy sort order: ascending, NaN last */
static int cmp_nan_last(const void *pa, const void *pb) {
    double a = *(const double *)pa, b = *(const double *)pb;
    int an = isnan(a), bn = isnan(b);
    if (an || bn) return an - bn;
    return (a > b) - (a < b);
}

typedef struct { double v; int i; } keyed_t;
static int cmp_keyed(const void *pa, const void *pb) {
    const keyed_t *a = (const keyed_t *)pa, *b = (const keyed_t *)pb;
    int c = cmp_nan_last(&a->v, &b->v);
    return c ? c : (a->i > b->i) - (a->i < b->i);
}

/* fdr/__init__.py:12-33.  Ties/NaNs receive equal counts regardless of the
 * argsort tie order, so a stable order is used. */
void orc_emperical_fdr(const double *pvals_null, int64_t n_null, const double *pvals, int n,
                       double *out) {
    double *nul = (double *)malloc((n_null > 0 ? n_null : 1) * sizeof(double));
    memcpy(nul, pvals_null, n_null * sizeof(double));
    qsort(nul, n_null, sizeof(double), cmp_nan_last);
    keyed_t *ord = (keyed_t *)malloc((n > 0 ? n : 1) * sizeof(keyed_t));
    double *sorted = (double *)malloc((n > 0 ? n : 1) * sizeof(double));
    double *cnt = (double *)malloc((n > 0 ? n : 1) * sizeof(double));
    for (int i = 0; i < n; i++) {
        ord[i].v = pvals[i];
        ord[i].i = i;
    }
    qsort(ord, n, sizeof(keyed_t), cmp_keyed);
    for (int i = 0; i < n; i++) sorted[i] = ord[i].v;
    orc_bisect(nul, (int)n_null, sorted, n, cnt);
    for (int i = 0; i < n; i++) {
        double f = cnt[i] / (double)n_null;
        if (f > 1) f = 1;
        out[ord[i].i] = f;
    }
    free(nul);
    free(ord);
    free(sorted);
    free(cnt);
}

/* utils.pyx:15-50 */
int orc_segment(const double *x, int n, double threshold, int w, int decreasing, int32_t *seg,
                int cap) {
    double dir = decreasing ? -1 : 1;
    int nseg = 0, cur = -1;
    for (int i = 0; i < n; i++) {
        if (cur < 0) {
            if (dir * x[i] >= dir * threshold) cur = i - w + 1;
        } else if (dir * x[i] < dir * threshold) {
            if (nseg > 0 && cur <= seg[2 * (nseg - 1) + 1]) {
                seg[2 * (nseg - 1) + 1] = i - 1 + w;
            } else if (nseg < cap) {
                seg[2 * nseg] = cur;
                seg[2 * nseg + 1] = i - 1 + w;
                nseg++;
            }
            cur = -1;
        }
    }
    return nseg;
}

/* cli/learn_dm.py:276-287: hist[int(exp), int(obs)] += 1 over a batch, IndexError ignored.  Python's int()
 * truncates toward zero; a NEGATIVE index of a numpy array counts from the end (no IndexError while it
 * is >= -dim).  int() of a NaN or an infinity raises ValueError / OverflowError, which the reference does
 * not catch (the job ends): such pairs are skipped here and counted in the return value. */
int64_t orc_hist2d(const double *ex, const double *ob, int64_t n, int rows, int cols, int64_t *hist) {
    int64_t not_finite = 0;
    for (int64_t i = 0; i < n; i++) {
        if (!isfinite(ex[i]) || !isfinite(ob[i])) {
            not_finite++;
            continue;
        }
        double te = trunc(ex[i]), to = trunc(ob[i]);
        if (te < -(double)rows || te >= (double)rows || to < -(double)cols || to >= (double)cols) continue;
        long r = (long)te, c = (long)to;
        if (r < 0) r += rows;
        if (c < 0) c += cols;
        hist[(size_t)r * cols + c] += 1;
    }
    return not_finite;
}

/* ------------------------------------------------------------------ composite path */

/* cli/detect.py:120-130 through modeling/predict.pyx:116-163 */
int orc_detect_interval(const double *cp, const double *cm, const uint8_t *seq, int L, int hw,
                        int shw, double clip, const double *table, double dflt,
                        const double *mu_par9, const double *r_par15, const int32_t *scales,
                        int n_scales, double *exp_out, double *obs_out, double *p_out,
                        double *winp_out) {
    int pad = hw + shw;           /* predict.pyx:114 */
    int l = L + 2 * pad + 1;      /* predict.pyx:132-133 */
    double *buf = (double *)malloc(6 * (size_t)l * sizeof(double));
    double *pf = buf, *pr = buf + l, *ef = buf + 2 * l, *er = buf + 3 * l, *wf = buf + 4 * l,
           *wr = buf + 5 * l;
    orc_kmer_probs(seq, (int64_t)l + 6, table, dflt, pf, pr, NULL, NULL);
    orc_fast_predict(cp, pf, l, hw, shw, clip, ef, wf);
    orc_fast_predict(cm, pr, l, hw, shw, clip, er, wr);
    /* predict.pyx:157-161 slice [pad, l-pad); detect.py:121-122 '+'[1:] + '-'[:-1] */
    for (int t = 0; t < L; t++) {
        obs_out[t] = cp[pad + 1 + t] + cm[pad + t];
        exp_out[t] = ef[pad + 1 + t] + er[pad + t];
    }
    free(buf);
    int rc = orc_nb_values(0, mu_par9, r_par15, exp_out, obs_out, L, p_out);
    if (rc) return rc;
    for (int s = 0; s < n_scales; s++)
        orc_window(ORC_WIN_STOUFFER, p_out, NULL, L, scales[s], winp_out + (size_t)s * L);
    return 0;
}

int orc_detect_batch(const double *cp, const double *cm, const uint8_t *seq, int64_t n_iv, int L,
                     int hw, int shw, double clip, const double *table, double dflt,
                     const double *mu_par9, const double *r_par15, const int32_t *scales,
                     int n_scales, double *exp_out, double *obs_out, double *p_out,
                     double *winp_out, int n_threads) {
    int64_t l = (int64_t)L + 2 * (hw + shw) + 1;
    int64_t total = n_iv * L;
    int bad = 0;
    (void)n_threads;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 16) num_threads(n_threads > 0 ? n_threads : 1) reduction(| : bad)
#endif
    for (int64_t i = 0; i < n_iv; i++) {
        double *wtmp = (double *)malloc((size_t)n_scales * L * sizeof(double));
        int rc = orc_detect_interval(cp + i * l, cm + i * l, seq + i * (l + 6), L, hw, shw, clip,
                                     table, dflt, mu_par9, r_par15, scales, n_scales,
                                     exp_out + i * L, obs_out + i * L, p_out + i * L, wtmp);
        for (int s = 0; s < n_scales; s++)
            memcpy(winp_out + (size_t)s * total + i * L, wtmp + (size_t)s * L, L * sizeof(double));
        free(wtmp);
        bad |= rc;
    }
    return bad;
}

/* stats/posterior.py:119: windowing.sum(dm.log_pmf_values(exp*delta, obs), w) */
int orc_log_likelihood_row(const double *mu_par9, const double *r_par15, const double *obs,
                           const double *exp_, const double *delta, int n, int w, double *out) {
    double *e = (double *)malloc((n > 0 ? n : 1) * sizeof(double));
    double *lp = (double *)malloc((n > 0 ? n : 1) * sizeof(double));
    for (int i = 0; i < n; i++) e[i] = exp_[i] * delta[i];
    int rc = orc_nb_values(1, mu_par9, r_par15, e, obs, n, lp);
    if (!rc) orc_window(ORC_WIN_SUM, lp, NULL, n, w, out);
    free(e);
    free(lp);
    return rc;
}

/* ------------------------------------------------------------------ synthetic inputs */

uint64_t orc_splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    uint64_t z = x;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

/* SURVEY.md 8(d) cfg 2/3 generator: stream 0/1 = '+'/'-' cut counts U{0..19},
 * stream 2 = bases.  Element at global position p: h = mix(mix(seed + stream) + p). */
void orc_synth_fill(uint64_t seed, int64_t pos0, int64_t n, int stream, double *counts,
                    uint8_t *bases) {
    uint64_t key = orc_splitmix64(seed + (uint64_t)stream);
    for (int64_t i = 0; i < n; i++) {
        uint64_t h = orc_splitmix64(key + (uint64_t)(pos0 + i));
        if (counts) counts[i] = (double)((h >> 33) % 20u);
        if (bases) bases[i] = (uint8_t)("ACGT"[(h >> 13) & 3u]);
    }
}

/* Hotspot bursts of the heavy-tailed workload (include/fpt.h: fpt_synth_hotspots_dev), added in
 * place to counts of stream 0 ('+') or 1 ('-'). */
void orc_synth_hotspots(uint64_t seed, int64_t pos0, int64_t n, int stream, int padded_len, int per_mille,
                        double *counts) {
    const uint64_t key_iv = orc_splitmix64(seed + 7), key_noise = orc_splitmix64(seed + 8 + (uint64_t)stream);
    for (int64_t i = 0; i < n; i++) {
        const int64_t p = pos0 + i, iv = p / padded_len;
        const int u = (int)(p - iv * padded_len);
        const uint64_t hv = orc_splitmix64(key_iv + (uint64_t)iv);
        if ((int)(hv % 1000u) >= per_mille) continue;
        const int width = 80 + (int)((hv >> 40) % 120u), half = width / 2;
        const int span = padded_len - 2 * half > 1 ? padded_len - 2 * half : 1;
        const int centre = half + (int)((hv >> 20) % (uint64_t)span);
        const int dist = u > centre ? u - centre : centre - u;
        if (dist >= half) continue;
        const int peak = 100 + (int)((hv >> 10) % 400u);
        const uint64_t hn = orc_splitmix64(key_noise + (uint64_t)p);
        counts[i] += (double)((peak * (half - dist)) / half + (int)((hn >> 7) & 15u));
    }
}

/* ------------------------------------------------------------------ empirical FDR with the
 * library's reproducible null sampler (cli/detect.py:132-135 with the NB draws made from Philox
 * words -- alias tables, the inverse cdf off them -- instead of numpy's MT19937; see include/fpt.h fpt_fdr_dev) */

static void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1) {
    for (int r = 0; r < 10; r++) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
}

void orc_philox_raw(uint32_t *c4, uint32_t k0, uint32_t k1) { philox4x32_10(c4, k0, k1); }

/* sample 4j + w uses word w of the Philox block with counter word 2 = j: u = (word + 1/2) 2^-32 */
double orc_philox_uniform(uint64_t seed, uint64_t base, uint32_t sample) {
    uint32_t c[4] = {(uint32_t)base, (uint32_t)(base >> 32), sample >> 2, 0x66707464u};
    philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
    return ((double)c[sample & 3u] + 0.5) * (1.0 / 4294967296.0);
}

/* The library's null sampler (include/fpt.h fpt_fdr_dev, "the null draws"): at an integer expected
 * value below table_exp a draw is an ALIAS-table lookup on the 32-bit word, anywhere else the inverse
 * cdf on u.  Restated from the header's definition, operation for operation where doubles are involved:
 *   - the row's outcomes are k = 0 .. n-2 (probability cdf(k) - cdf(k-1)) and "n-1 or more" (1 - cdf(n-2)),
 *     n = 2^lg the smallest power of two with 1 - cdf(n-2) <= 2^-32, at most 2^lg_max <= table_k (lg_max <= 11)
 *   - Vose's construction with two queues filled in index order (small: n p < 1)
 *   - entry = threshold << lg | alias, threshold = round(q 2^(32-lg)) capped at 2^(32-lg) - 1
 *   - draw: slot = w >> (32 - lg), t = w mod 2^(32-lg), outcome = t < threshold ? slot : alias
 *   - the outcome "n-1 or more": inverse cdf from k = n-1 on with u' = cdf(n-2) + (1 - cdf(n-2)) frac,
 *     frac = (position of t in its part of the slot + 1/2) / (size of that part)
 * A row with a NaN gets the identity table. */
typedef struct {
    int lg, n;
    double *cdf;      /* cdf(0 .. n-1) (the last one is not an outcome of its own) */
    uint32_t *entry;  /* n */
} alias_row;

static int alias_lg_cap(int table_k) {
    int lg = 1;
    while (lg < 11 && (2 << lg) <= table_k) ++lg;
    return lg;
}

static uint32_t uniform_word(double u) { /* floor(u 2^32) inside the table; NaN and u < 0: 0 */
    double x = u * 4294967296.0;
    return !(x > 0.0) ? 0u : (x >= 4294967295.0 ? 0xffffffffu : (uint32_t)x);
}

static alias_row *alias_row_build(double pr, double r, int table_k) {
    alias_row *a = (alias_row *)malloc(sizeof *a);
    int lg_max = alias_lg_cap(table_k);
    a->lg = lg_max;
    for (int c = 1; c <= lg_max; c++) {
        int idx = (1 << c) - 2;
        if (idx < table_k && 1.0 - orc_nb_cdf(idx, pr, r) <= 1.0 / 4294967296.0) { a->lg = c; break; }
    }
    int lg = a->lg, n = a->n = 1 << lg;
    a->cdf = (double *)malloc(n * sizeof(double));
    a->entry = (uint32_t *)malloc(n * sizeof(uint32_t));
    double *q = (double *)malloc(n * sizeof(double));
    int *sq = (int *)malloc(n * sizeof(int)), *lq = (int *)malloc(n * sizeof(int)), *al = (int *)malloc(n * sizeof(int));
    int ns = 0, nl = 0, bad = 0;
    for (int k = 0; k < n; k++) a->cdf[k] = k < table_k ? orc_nb_cdf(k, pr, r) : 1.0;
    for (int k = 0; k < n; k++) {
        double pm = (k == n - 1 ? 1.0 : a->cdf[k]) - (k > 0 ? a->cdf[k - 1] : 0.0);
        if (pm != pm) bad = 1;
        pm = pm > 0.0 ? pm : 0.0;
        q[k] = pm * (double)n;
        al[k] = k;
        if (q[k] < 1.0) sq[ns++] = k; else lq[nl++] = k;
    }
    uint32_t top = 0xffffffffu >> lg;
    if (!bad) {
        int si = 0, li = 0, se = ns;
        while (si < se && li < nl) {
            int s_ = sq[si++], l = lq[li];
            al[s_] = l;
            double ql = (q[l] + q[s_]) - 1.0;
            q[l] = ql;
            if (ql < 1.0) { sq[se++] = l; ++li; }
        }
        while (li < nl) q[lq[li++]] = 1.0;
        while (si < se) { int s_ = sq[si++]; q[s_] = 1.0; al[s_] = s_; }
    }
    for (int k = 0; k < n; k++) {
        uint32_t th = top, ak = (uint32_t)k;
        if (!bad) {
            double t = floor(ldexp(q[k], 32 - lg) + 0.5);
            th = t >= (double)top ? top : (uint32_t)t;
            ak = (uint32_t)al[k];
        }
        a->entry[k] = (th << lg) | ak;
    }
    free(q); free(sq); free(lq); free(al);
    return a;
}

static void alias_row_free(alias_row *a) {
    if (a) { free(a->cdf); free(a->entry); free(a); }
}

/* smallest k > lo with cdf(k) >= u -> cdf(k): gallop, then bisect (lo: largest k known to have cdf(k) < u) */
static double inverse_cdf_from(double pr, double r, double u, int lo) {
    int step = 1, hi = lo + 1;
    double chi = orc_nb_cdf(hi, pr, r);
    while (chi < u && hi < (1 << 28)) {
        lo = hi;
        step <<= 1;
        hi = lo + step;
        chi = orc_nb_cdf(hi, pr, r);
    }
    while (hi - lo > 1) {
        int mid = lo + ((hi - lo) >> 1);
        double cm = orc_nb_cdf(mid, pr, r);
        if (cm >= u) { hi = mid; chi = cm; } else lo = mid;
    }
    return chi;
}

/* one draw at expected value ex with the uniform u (word = uniform_word(u)) -> cdf(k) of the outcome k;
 * rows: table_exp lazily built alias rows; k_out (optional): the outcome, -1 when it came from the direct search */
static double null_draw_pvalue(const double *mu_par9, const double *r_par15, double ex, double u,
                               int table_exp, int table_k, alias_row **rows, int *k_out) {
    double r, mu;
    orc_fit_r(r_par15, ex, &r);
    mu = orc_fit_mu(mu_par9, ex);
    double pr = r / (r + mu);
    int ei = (int)ex;
    if (k_out) *k_out = -1;
    if (!(ex >= 0.0 && ex < (double)table_exp && (double)ei == ex)) return inverse_cdf_from(pr, r, u, -1);
    if (!rows[ei]) rows[ei] = alias_row_build(pr, r, table_k);
    const alias_row *a = rows[ei];
    const int lg = a->lg;
    const uint32_t w = uniform_word(u), last = (uint32_t)a->n - 1u;
    const uint32_t slot = w >> (32 - lg), t = (uint32_t)(w << lg) >> lg, e = a->entry[slot], th = e >> lg;
    const uint32_t k = t < th ? slot : (e & last);
    if (k != last) {
        if (k_out) *k_out = (int)k;
        return a->cdf[k];
    }
    const double base = a->cdf[last - 1u];
    const uint32_t span = (0xffffffffu >> lg) + 1u;
    const int own = t < th;
    const double frac = ((double)(own ? t : t - th) + 0.5) / (double)(own ? th : span - th);
    return inverse_cdf_from(pr, r, fma(1.0 - base, frac, base), (int)last - 1);
}

/* test access to one row's table: entry_out[2^lg], cdf_out[2^lg]; returns lg */
int orc_null_alias_row(const double *mu_par9, const double *r_par15, double ex, int table_k, uint32_t *entry_out,
                       double *cdf_out) {
    double r, mu;
    orc_fit_r(r_par15, ex, &r);
    mu = orc_fit_mu(mu_par9, ex);
    alias_row *a = alias_row_build(r / (r + mu), r, table_k);
    int lg = a->lg;
    if (entry_out) memcpy(entry_out, a->entry, a->n * sizeof(uint32_t));
    if (cdf_out) memcpy(cdf_out, a->cdf, a->n * sizeof(double));
    alias_row_free(a);
    return lg;
}

/* n draws at one expected value from given uniforms: outcome (or -1) and cdf of each (tests of the sampler) */
void orc_null_draws(const double *mu_par9, const double *r_par15, double ex, const double *u, int64_t n,
                    int table_exp, int table_k, int32_t *k_out, double *p_out) {
    alias_row **rows = (alias_row **)calloc(table_exp > 0 ? table_exp : 1, sizeof(alias_row *));
    for (int64_t i = 0; i < n; i++) {
        int k;
        p_out[i] = null_draw_pvalue(mu_par9, r_par15, ex, u[i], table_exp, table_k, rows, &k);
        if (k_out) k_out[i] = k;
    }
    for (int i = 0; i < table_exp; i++) alias_row_free(rows[i]);
    free(rows);
}

/* one interval: exp_[L], winp[L] -> efdr[L].  uniforms (L*times, base-major) may be NULL
 * (then Philox(seed, base0 + t, s)); null_out (L*times) optionally receives the null p-values */
void orc_fdr_null(const double *mu_par9, const double *r_par15, const double *exp_, const double *winp,
                  int L, int hw, int times, uint64_t seed, int64_t base0, const double *uniforms,
                  int table_exp, int table_k, double *efdr_out, double *null_out) {
    alias_row **rows = (alias_row **)calloc(table_exp > 0 ? table_exp : 1, sizeof(alias_row *));
    double *pn = (double *)malloc((size_t)L * times * sizeof(double));   /* [L][times] like sample() */
    double *wn = (double *)malloc((size_t)L * times * sizeof(double));
    double *col = (double *)malloc((size_t)L * sizeof(double));
    double *wcol = (double *)malloc((size_t)L * sizeof(double));
    for (int t = 0; t < L; t++)
        for (int s = 0; s < times; s++) {
            double u = uniforms ? uniforms[(size_t)t * times + s]
                                : orc_philox_uniform(seed, (uint64_t)(base0 + t), (uint32_t)s);
            pn[(size_t)t * times + s] = null_draw_pvalue(mu_par9, r_par15, exp_[t], u, table_exp, table_k, rows, NULL);
        }
    /* detect.py:133: np.apply_along_axis(win_pval_fn, 0, pvals_null) */
    for (int s = 0; s < times; s++) {
        for (int t = 0; t < L; t++) col[t] = pn[(size_t)t * times + s];
        orc_window(ORC_WIN_STOUFFER, col, NULL, L, hw, wcol);
        for (int t = 0; t < L; t++) wn[(size_t)t * times + s] = wcol[t];
    }
    if (null_out) memcpy(null_out, wn, (size_t)L * times * sizeof(double));
    orc_emperical_fdr(wn, (int64_t)L * times, winp, L, efdr_out);
    for (int i = 0; i < table_exp; i++) alias_row_free(rows[i]);
    free(rows); free(pn); free(wn); free(col); free(wcol);
}
