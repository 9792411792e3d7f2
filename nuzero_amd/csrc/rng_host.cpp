// Host random streams: numpy's legacy RandomState (MT19937) restated in C++.
//
// The reference draws root noise and move-selection uniforms from the legacy
// global np.random stream (Search/Explorer.py:77-78,89,199,208).  Self-play
// results are only reproducible against it if the engine consumes the same
// stream in the same order, and the rejection samplers take a data-dependent
// number of draws, so the generator itself has to be reproduced:
//   RandomState(seed)          -> init_genrand (Knuth multiplier 1812433253)
//   random_sample()            -> (a >> 5, b >> 6) 53-bit double
//   gamma(shape, scale, n)     -> scale * legacy_standard_gamma(shape)
// The double-precision libm calls (log, pow, sqrt) are glibc's, which is what
// numpy links against on Linux.  Checked against numpy in
// tests/test_rng_host.py (golden vectors + live RandomState).
#include <cmath>
#include <cstdint>

#include "../../include/nuzero_amd.h"

struct nz_rng {
  uint32_t key[624];
  int pos;
  int has_gauss;
  double gauss;
};

namespace {

void seed_state(nz_rng* r, uint32_t seed) {
  for (int i = 0; i < 624; ++i) {
    r->key[i] = seed;
    seed = 1812433253u * (seed ^ (seed >> 30)) + (uint32_t)i + 1u;
  }
  r->pos = 624;
  r->has_gauss = 0;
  r->gauss = 0.0;
}

void regenerate(nz_rng* r) {
  const uint32_t upper = 0x80000000u, lower = 0x7fffffffu, matrix = 0x9908b0dfu;
  uint32_t* k = r->key;
  int i;
  uint32_t y;
  for (i = 0; i < 624 - 397; ++i) {
    y = (k[i] & upper) | (k[i + 1] & lower);
    k[i] = k[i + 397] ^ (y >> 1) ^ ((y & 1u) ? matrix : 0u);
  }
  for (; i < 623; ++i) {
    y = (k[i] & upper) | (k[i + 1] & lower);
    k[i] = k[i + (397 - 624)] ^ (y >> 1) ^ ((y & 1u) ? matrix : 0u);
  }
  y = (k[623] & upper) | (k[0] & lower);
  k[623] = k[396] ^ (y >> 1) ^ ((y & 1u) ? matrix : 0u);
  r->pos = 0;
}

inline uint32_t next_u32(nz_rng* r) {
  if (r->pos == 624) regenerate(r);
  uint32_t y = r->key[r->pos++];
  y ^= y >> 11;
  y ^= (y << 7) & 0x9d2c5680u;
  y ^= (y << 15) & 0xefc60000u;
  y ^= y >> 18;
  return y;
}

inline double next_double(nz_rng* r) {
  const uint32_t a = next_u32(r) >> 5, b = next_u32(r) >> 6;
  return (a * 67108864.0 + b) / 9007199254740992.0;
}

inline double std_exponential(nz_rng* r) { return -std::log(1.0 - next_double(r)); }

double gauss(nz_rng* r) {
  if (r->has_gauss) {
    const double t = r->gauss;
    r->has_gauss = 0;
    r->gauss = 0.0;
    return t;
  }
  double x1, x2, r2;
  do {
    x1 = 2.0 * next_double(r) - 1.0;
    x2 = 2.0 * next_double(r) - 1.0;
    r2 = x1 * x1 + x2 * x2;
  } while (r2 >= 1.0 || r2 == 0.0);
  const double f = std::sqrt(-2.0 * std::log(r2) / r2);
  r->gauss = f * x1;
  r->has_gauss = 1;
  return f * x2;
}

double std_gamma(nz_rng* r, double shape) {
  if (shape == 1.0) return std_exponential(r);
  if (shape == 0.0) return 0.0;
  if (shape < 1.0) {
    for (;;) {
      const double u = next_double(r);
      const double v = std_exponential(r);
      if (u <= 1.0 - shape) {
        const double x = std::pow(u, 1.0 / shape);
        if (x <= v) return x;
      } else {
        const double y = -std::log((1.0 - u) / shape);
        const double x = std::pow(1.0 - shape + shape * y, 1.0 / shape);
        if (x <= v + y) return x;
      }
    }
  }
  const double b = shape - 1.0 / 3.0;
  const double c = 1.0 / std::sqrt(9.0 * b);
  for (;;) {
    double x, v;
    do {
      x = gauss(r);
      v = 1.0 + c * x;
    } while (v <= 0.0);
    v = v * v * v;
    const double u = next_double(r);
    if (u < 1.0 - 0.0331 * (x * x) * (x * x)) return b * v;
    if (std::log(u) < 0.5 * x * x + b * (1.0 - v + std::log(v))) return b * v;
  }
}

}  // namespace

extern "C" {

// a stream that continues where a numpy RandomState stands (RandomState.get_state(): key, pos, has_gauss, cached_gaussian)
nz_rng* nz_rng_create_state(const uint32_t* key624, int32_t pos, int32_t has_gauss, double cached_gaussian) {
  if (!key624 || pos < 0 || pos > 624) return nullptr;
  nz_rng* r = new nz_rng;
  for (int i = 0; i < 624; ++i) r->key[i] = key624[i];
  r->pos = pos;
  r->has_gauss = has_gauss ? 1 : 0;
  r->gauss = cached_gaussian;
  return r;
}

nz_rng* nz_rng_clone(const nz_rng* r) { return r ? new nz_rng(*r) : nullptr; }

nz_rng* nz_rng_create(uint32_t seed) {
  nz_rng* r = new nz_rng;
  seed_state(r, seed);
  return r;
}

void nz_rng_destroy(nz_rng* r) { delete r; }

void nz_rng_seed(nz_rng* r, uint32_t seed) { seed_state(r, seed); }

uint32_t nz_rng_u32(nz_rng* r) { return next_u32(r); }

double nz_rng_double(nz_rng* r) { return next_double(r); }

void nz_rng_gamma(nz_rng* r, double shape, double scale, int32_t n, double* out) {
  for (int32_t i = 0; i < n; ++i) out[i] = scale * std_gamma(r, shape);
}

}  // extern "C"
