/* Exhaustive check of the restated glibc powf against the libm of this machine, on the domain the hot path uses:
 * every binary32 x in [0, 1] for y in {0.8f, 2.0f, 3.0f}.  Also spot-checks pw_exp against libm exp. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
float pedn_oracle_powf(float x, float y);
double pedn_oracle_exp(double x);
int main(void) {
  const float ys[3] = {0.8f, 2.0f, 3.0f};
  for (int k = 0; k < 3; ++k) {
    long bad = 0;
    float y = ys[k];
#pragma omp parallel for reduction(+ : bad)
    for (uint32_t u = 0; u <= 0x3f800000u; ++u) {
      float x; memcpy(&x, &u, 4);
      float a = pedn_oracle_powf(x, y), b = powf(x, y);
      if (memcmp(&a, &b, 4) != 0) ++bad;
    }
    printf("powf(x, %g): %ld mismatches over %u inputs\n", (double)y, bad, 0x3f800001u);
  }
  long bad = 0, n = 0; double maxulp = 0;
  for (double x = -300.0; x < 300.0; x += 1.234567e-5) {
    double a = pedn_oracle_exp(x), b = exp(x);
    ++n;
    if (a != b) { ++bad; double u = fabs(a - b) / (nextafter(b, INFINITY) - b); if (u > maxulp) maxulp = u; }
  }
  printf("exp: %ld of %ld differ from libm (max %.1f ulp)\n", bad, n, maxulp);
  return 0;
}
