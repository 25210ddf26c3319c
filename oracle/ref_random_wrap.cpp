// ref_random_wrap.cpp — TEST INFRASTRUCTURE. Exposes the reference's OWN lib/random.cuh (compiled
// from where it lies under /root/reference, never copied) through a C ABI so that the oracle's
// RNG restatement can be pinned against it. Built only when /root/reference is present; the
// outputs it produced are committed as tests/golden/rng_kat.json (see tests/golden/make_rng_kat.py).
// lib/random.cuh is the only reference file on the hot path that compiles without OptiX/CUDA.
#include <stdint.h>
#define __host__
#define __device__
#define __inline__ inline
#include RTW_REF_RANDOM_CUH

extern "C" {
uint32_t ref_tea64(uint32_t a, uint32_t b) { return tea<64>(a, b); }
uint32_t ref_tea16(uint32_t a, uint32_t b) { return tea<16>(a, b); }
uint32_t ref_tea4(uint32_t a, uint32_t b) { return tea<4>(a, b); }
uint32_t ref_xorshift32(uint32_t* s) { return xorshift32(*s); }
float ref_randf(uint32_t* s) { return randf(*s); }
}
