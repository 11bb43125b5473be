#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <string.h>
static uint64_t s = 88172645463325252ULL;
static inline uint64_t rnd(void){ s ^= s<<13; s ^= s>>7; s ^= s<<17; return s; }
static inline double u01(void){ return (rnd()>>11) * (1.0/9007199254740992.0); }
int main(void){
    long bad=0, n=0;
    for (int outer=0; outer<2000; ++outer){
        // divisor: density-like 0.05..1.3, random mantissa
        double b = 0.05 + 1.25*u01();
        if (outer % 7 == 0) { uint64_t m = rnd(); double t; uint64_t bits = (0x3FFULL<<52) | (m & 0xFFFFFFFFFFFFFULL); memcpy(&t,&bits,8); b = t*0.3; }
        double y = 1.0 / b;
        for (int i=0;i<200000;++i){
            double a;
            int mode = i & 3;
            if (mode==0) a = (u01()-0.5)*200.0;          // momentum-like
            else if (mode==1) a = 250.0 + 100.0*u01()*b;  // rho*theta
            else if (mode==2) a = u01()*0.02*b;           // rho q
            else { uint64_t m = rnd(); double t; uint64_t bits = ((uint64_t)(0x3C0 + (m>>58))<<52) | (m & 0xFFFFFFFFFFFFFULL); memcpy(&t,&bits,8); a = t; }
            double q0 = a*y;
            double r = fma(-q0, b, a);
            double q = fma(r, y, q0);
            double ref = a/b;
            if (q != ref) { if (bad<5) printf("mismatch a=%.17g b=%.17g q=%.17g ref=%.17g\n", a,b,q,ref); ++bad; }
            ++n;
        }
    }
    printf("%ld mismatches of %ld\n", bad, n);
    return 0;
}
