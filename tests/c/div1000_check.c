/* CPU check of the identity behind sitrk_geom.h::div1000 (same IEEE-754 binary64 operations as on the GPU):
 *     RN(q + r z) == x / 1000.   with z = RN(1/1000), q = RN(x z), r = fma(-q, 1000, x)
 * for 2^-900 <= |x| < 2^900.  Walks (a) pseudo-random mantissas and exponents, (b) the mantissas for which x/1000
 * lies closest to a midpoint of two doubles (X 2^g = 125 (2M+1) +- 1: X = 21 or 104 mod 125 for g = 8; 42 or 83 for
 * g = 7), (c) binade edges.  Prints the number of mismatches; exit status 0 iff none.
 * Build: gcc -O2 -ffp-contract=off -o div1000_check div1000_check.c -lm */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static double fast(double x)
{
    const double z = 0x1.0624dd2f1a9fcp-10;
    const double q = x * z;
    const double r = fma(-q, 1000., x);
    return fma(r, z, q);
}

static uint64_t bits(double d) { uint64_t u; memcpy(&u, &d, 8); return u; }
static double from(uint64_t u) { double d; memcpy(&d, &u, 8); return d; }

static long bad = 0, seen = 0;
static void one(double x)
{
    volatile double ref = x / 1000.;
    if (bits(fast(x)) != bits(ref)) { if (bad < 5) fprintf(stderr, "mismatch at %a\n", x); bad++; }
    if (bits(fast(-x)) != bits(-ref)) bad++;
    seen += 2;
}

int main(int argc, char **argv)
{
    long n = argc > 1 ? atol(argv[1]) : 4000000;
    uint64_t s = 0x9E3779B97F4A7C15ull;
    for (long k = 0; k < n; k++) {                       /* (a) */
        s = s * 6364136223846793005ull + 1442695040888963407ull;
        uint64_t m = (s >> 12);                           /* 52 mantissa bits */
        s = s * 6364136223846793005ull + 1442695040888963407ull;
        uint64_t e = 123 + (s >> 33) % 1800;              /* biased exponent in [123, 1923) */
        one(from((e << 52) | m));
    }
    const int res[4] = {21, 104, 42, 83};
    for (int r = 0; r < 4; r++)                           /* (b) */
        for (long k = 0; k < n / 4; k++) {
            s = s * 6364136223846793005ull + 1442695040888963407ull;
            uint64_t X = (1ull << 52) + (s >> 12);
            X = X - (X % 125) + res[r];
            if (X >> 53) X -= 125ull << 40;
            if (!(X >> 52)) continue;
            one(ldexp((double)X, -52 + (int)(k % 1700) - 850));
        }
    for (int e = 123; e < 1923; e++) {                    /* (c) */
        one(from((uint64_t)e << 52));
        one(from(((uint64_t)e << 52) | 0xFFFFFFFFFFFFFull));
        one(from(((uint64_t)e << 52) | 1ull));
        one(1000. * from((uint64_t)(e < 1900 ? e : 1900) << 52));
    }
    printf("%ld values, %ld mismatches\n", seen, bad);
    return bad != 0;
}
