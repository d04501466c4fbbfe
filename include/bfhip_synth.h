/* bfhip_synth.h -- counter-based synthetic value stream shared by host and
 * device code.  Benchmark operands at N >= 262144 are "structure-exact,
 * value-random" (SURVEY.md section 8(d)): only block shapes come from the
 * reference's structure logic, values are this stream.  Every operation
 * below is exact in IEEE double (integer -> double below 2^53, scaling by a
 * power of two, a subtraction of two multiples of 2^-52 with |result| < 1),
 * so host and device produce bit-identical operands.
 */
#ifndef BFHIP_SYNTH_H
#define BFHIP_SYNTH_H

#include <stdint.h>

#if defined(__HIPCC__)
#define BFHIP_HD __host__ __device__ static inline
#else
#define BFHIP_HD static inline
#endif

BFHIP_HD uint64_t bfhip_mix64(uint64_t z) {
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
  return z ^ (z >> 31);
}

/* uniform in [-1, 1) */
BFHIP_HD double bfhip_synth_value(uint64_t seed, uint64_t idx, int imag) {
  uint64_t z = bfhip_mix64(seed + 0x9e3779b97f4a7c15ULL * (2 * idx + (uint64_t)(imag != 0) + 1));
  return (double)(int64_t)(z >> 11) * 0x1.0p-52 - 1.0;
}

#endif
