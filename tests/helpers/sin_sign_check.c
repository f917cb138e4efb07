/* Host-side check of the Square oscillator's exact sign-of-sine (mixlab_amd/csrc/mx_k_stream.hip: sin_is_negative):
 * the same statements, compiled for the CPU (hardware fma), against the sign bit of the platform libm's sin -- which is what the
 * reference computes (src/module/oscillator.rs:15-26,80).  Sweeps the t / SR * freq * 2 pi arguments of both rates and many
 * frequencies (including days of uptime), plus the doubles next to k * pi, where a last-bit error would flip the sign. */
#include <stdio.h>
#include <math.h>
#include <stdint.h>
static int sin_is_negative(double x){
    if (x == 0.0) return signbit(x)!=0;
    if (!(fabs(x) < 1099511627776.0)) return signbit(sin(x))!=0;
    const double PI_HI = 0x1.921fb54442d18p+1, PI_MID = 0x1.1a62633145c07p-53, PI_LO = -0x1.f1976b7ed8fbcp-109;
    const double m = rint(x * 0x1.45f306dc9c883p-2);
    const double r1 = fma(-m, PI_HI, x);
    const double p2 = m * PI_MID, e2 = fma(m, PI_MID, -p2);
    const double sd = r1 - p2, bb = sd - r1;
    const double t = (r1 - (sd - bb)) + (-p2 - bb);
    const double r = sd + ((t - e2) - m * PI_LO);
    const int m_odd = ((long long)m & 1LL) != 0;
    return (r < 0.0) != m_odd;
}
int main(){ long bad=0,tot=0; double srs[2]={44100,48000}; double freqs[]={100,440,880.5,1000,12000,0.5,19999.9,-440};
  for(int s=0;s<2;s++) for(int f=0;f<8;f++) for(uint64_t t=0;t<3000000;t++){ uint64_t tt = t + (t%3==0? 4000000000ull:0) + (t%7==0?(1ull<<36):0);
    double t0=(double)tt/srs[s]; double n=t0*freqs[f]; double x=n*2.0*M_PI; int a=signbit(sin(x))!=0; int b=sin_is_negative(x); tot++; if(a!=b){ if(bad<5) printf("x=%a sin=%a\n",x,sin(x)); bad++; } }
  /* exact multiples: x = RN(k*pi) */
  for(long k=1;k<2000000;k++){ double x=(double)k*M_PI; int a=signbit(sin(x))!=0; int b=sin_is_negative(x); tot++; if(a!=b) bad++; x=nextafter(x,0); a=signbit(sin(x))!=0; b=sin_is_negative(x); tot++; if(a!=b) bad++; }
  printf("%ld / %ld disagree with glibc sin sign\n",bad,tot); return 0; }
