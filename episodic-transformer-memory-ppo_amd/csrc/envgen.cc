// Host-side generator of the synthetic benchmark environment ("obs ~ U[0,1) float32 [3,84,84] from
// numpy default_rng(seed + worker_id)"; the reference's environments live in worker processes, /root/reference worker.py:20-48).
//
// What it is: numpy's PCG64 bit generator + Generator.random(dtype=float32), restated in C so that EVERY observation of the timed
// region can be a fresh draw of its worker's stream (environment `pool: 0`) without the host becoming the bottleneck: numpy fills one
// 3x84x84 observation in 30 - 50 us of one core (a serial 128-bit LCG behind a function pointer per element).  The stream is the
// same sequence of floats, bit for bit (tests/test_host_logic.py compares against numpy across lengths, seeds and chained calls):
//     state <- state * M + inc (mod 2^128);  out64 = rotr64(hi ^ lo, hi >> 58)         (PCG XSL-RR 128/64, numpy pcg64.h)
//     float32 pair = ((uint32)out64 >> 8) * 2^-24, ((out64 >> 32) >> 8) * 2^-24          (low half first: numpy's next_uint32 buffering)
// It is fast because the LCG is advanced in LANES of independent chains: s_{n+k} = A_k s_n + C_k with A_k = M^k and
// C_k = inc (M^{k-1} + ... + 1): eight scalar lanes give the multiplier pipeline eight independent 128-bit products in flight instead
// of one dependent chain (2.3 x numpy), sixteen lanes in two AVX-512 vectors where the host has them (chosen at run time).
//
// No device code, no HIP: the worker processes of environments/shm_env.py (numpy only) load this library too.
#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <mutex>
#include <new>
#include <thread>
#include <vector>
#include <immintrin.h>

#include "../../include/etm_envgen.h"

namespace {
typedef unsigned __int128 u128;
const u128 PCG_MULT = ((u128)2549297995355413924ULL << 64) | (u128)4865540595714422341ULL;   // numpy pcg64.h: PCG_DEFAULT_MULTIPLIER_128

inline uint64_t rotr64(uint64_t v, unsigned r) { return (v >> r) | (v << ((-r) & 63)); }
inline uint64_t output_xsl_rr(u128 s) { return rotr64((uint64_t)(s >> 64) ^ (uint64_t)s, (unsigned)(s >> 122)); }
inline void put_pair(float *out, uint64_t v) {
  out[0] = (float)((uint32_t)v >> 8) * (1.0f / 16777216.0f);
  out[1] = (float)((uint32_t)(v >> 32) >> 8) * (1.0f / 16777216.0f);
}

constexpr int LANES = 8;

// (A, C) of k steps at once: s -> A s + C
inline void jump_of(int64_t k, u128 inc, u128 &A, u128 &C) {
  u128 accA = 1, accC = 0, curA = PCG_MULT, curC = inc;
  for (; k > 0; k >>= 1) {
    if (k & 1) { accC = accC * curA + curC; accA *= curA; }
    curC = curC * curA + curC;
    curA *= curA;
  }
  A = accA; C = accC;
}

// ---- AVX-512 form: 16 lanes in two vectors of 64-bit halves.  Per lane and round:  (h, l) <- (h, l) * (Ah, Al) + (Ch, Cl) mod 2^128
//   l * Al as four 32 x 32 -> 64 products (vpmuludq), the two cross terms l * Ah and h * Al by vpmullq; the output permutation is a
//   variable rotate (vprorvq), and the sixteen 32-bit halves of a vector ARE the output order (low half of a lane first), so one
//   shift + one conversion + one multiply turn a vector into sixteen floats.
__attribute__((target("avx512f,avx512dq"))) inline __m512i lcg_step512(__m512i &h, __m512i l, __m512i Ah, __m512i Al, __m512i Al_hi32,
                                                                       __m512i Ch, __m512i Cl) {
  const __m512i M32 = _mm512_set1_epi64(0xffffffffLL);
  const __m512i l1 = _mm512_srli_epi64(l, 32);
  const __m512i p00 = _mm512_mul_epu32(l, Al), p01 = _mm512_mul_epu32(l, Al_hi32), p10 = _mm512_mul_epu32(l1, Al), p11 = _mm512_mul_epu32(l1, Al_hi32);
  const __m512i mid = _mm512_add_epi64(_mm512_add_epi64(_mm512_srli_epi64(p00, 32), _mm512_and_si512(p01, M32)), _mm512_and_si512(p10, M32));
  const __m512i lo = _mm512_or_si512(_mm512_and_si512(p00, M32), _mm512_slli_epi64(mid, 32));
  __m512i hi = _mm512_add_epi64(_mm512_add_epi64(p11, _mm512_srli_epi64(mid, 32)), _mm512_add_epi64(_mm512_srli_epi64(p01, 32), _mm512_srli_epi64(p10, 32)));
  hi = _mm512_add_epi64(hi, _mm512_add_epi64(_mm512_mullo_epi64(l, Ah), _mm512_mullo_epi64(h, Al)));
  const __m512i nl = _mm512_add_epi64(lo, Cl);
  const __mmask8 carry = _mm512_cmplt_epu64_mask(nl, lo);
  hi = _mm512_add_epi64(hi, Ch);
  h = _mm512_mask_add_epi64(hi, carry, hi, _mm512_set1_epi64(1));
  return nl;
}
__attribute__((target("avx512f,avx512dq"))) inline void emit512(float *out, __m512i h, __m512i l) {
  const __m512i v = _mm512_rorv_epi64(_mm512_xor_si512(h, l), _mm512_srli_epi64(h, 58));
  _mm512_storeu_ps(out, _mm512_mul_ps(_mm512_cvtepi32_ps(_mm512_srli_epi32(v, 8)), _mm512_set1_ps(1.0f / 16777216.0f)));
}
__attribute__((target("avx512f,avx512dq"))) int64_t rounds_avx512(u128 s, u128 inc, float *out, int64_t m) {
  constexpr int NV = 4, VL = 8 * NV;      // four independent vector chains cover the latency of vpmullq
  const int64_t rounds = m / VL;
  u128 A, C;
  jump_of(VL, inc, A, C);
  alignas(64) uint64_t hs[VL], ls[VL];
  u128 t = s;
  for (int j = 0; j < VL; ++j) { t = t * PCG_MULT + inc; hs[j] = (uint64_t)(t >> 64); ls[j] = (uint64_t)t; }
  __m512i h[NV], l[NV];
  for (int v = 0; v < NV; ++v) { h[v] = _mm512_load_si512(hs + 8 * v); l[v] = _mm512_load_si512(ls + 8 * v); }
  const __m512i Ah = _mm512_set1_epi64((long long)(uint64_t)(A >> 64)), Al = _mm512_set1_epi64((long long)(uint64_t)A);
  const __m512i Al_hi32 = _mm512_srli_epi64(Al, 32);
  const __m512i Ch = _mm512_set1_epi64((long long)(uint64_t)(C >> 64)), Cl = _mm512_set1_epi64((long long)(uint64_t)C);
  for (int64_t r = 0; r < rounds; ++r) {
#pragma GCC unroll 4
    for (int v = 0; v < NV; ++v) {
      emit512(out + 16 * v, h[v], l[v]);
      l[v] = lcg_step512(h[v], l[v], Ah, Al, Al_hi32, Ch, Cl);
    }
    out += 2 * VL;
  }
  return rounds * VL;
}

// ---- portable form: LANES independent 128-bit chains in scalar registers
int64_t rounds_scalar(u128 s, u128 inc, float *out, int64_t m) {
  u128 A, C;
  jump_of(LANES, inc, A, C);
  u128 lane[LANES];
  u128 t = s;
  for (int j = 0; j < LANES; ++j) { t = t * PCG_MULT + inc; lane[j] = t; }      // s_1 .. s_LANES
  const int64_t rounds = m / LANES;
  for (int64_t r = 0; r < rounds; ++r) {
#pragma GCC unroll 8
    for (int j = 0; j < LANES; ++j) {
      put_pair(out + 2 * j, output_xsl_rr(lane[j]));
      lane[j] = lane[j] * A + C;
    }
    out += 2 * LANES;
  }
  return rounds * LANES;
}

const bool g_has_avx512 = __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512dq");
int g_force_scalar = 0;      // etm_envgen_set_vector(0): tests compare the two forms

// n floats (n even, no half-consumed 64-bit output pending) of the stream (state, inc); the state is left where numpy's would be.
void fill_f32(uint64_t *st4, float *out, int64_t n) {
  u128 s = ((u128)st4[0] << 64) | st4[1];
  const u128 inc = ((u128)st4[2] << 64) | st4[3];
  int64_t m = n / 2;                      // 64-bit outputs
  if (m >= 64) {
    const int64_t done = (g_has_avx512 && !g_force_scalar) ? rounds_avx512(s, inc, out, m) : rounds_scalar(s, inc, out, m);
    u128 A, C;
    jump_of(done, inc, A, C);            // the serial state catches up: `done` steps at once
    s = s * A + C;
    out += 2 * done;
    m -= done;
  }
  for (int64_t i = 0; i < m; ++i) {
    s = s * PCG_MULT + inc;
    put_pair(out, output_xsl_rr(s));
    out += 2;
  }
  st4[0] = (uint64_t)(s >> 64);
  st4[1] = (uint64_t)s;
}

// ---- a small pool of helper threads (the pattern of csrc/host_copy.hip): rows of a step are drawn side by side
struct Pool {
  int n = 1;
  std::vector<std::thread> helpers;
  std::atomic<uint64_t> gen{0};
  std::atomic<int> pending{0};
  std::atomic<int> next{0};
  std::atomic<bool> stop{false};
  std::atomic<int> sleepers{0};
  std::atomic<int> spin_pauses{40000};
  std::mutex m;
  std::condition_variable cv;
  uint64_t *states = nullptr;
  float *out = nullptr;
  int64_t row_floats = 0;
  int rows = 0;
};

void work(Pool *p) {
  for (;;) {
    const int r = p->next.fetch_add(1, std::memory_order_relaxed);
    if (r >= p->rows) return;
    fill_f32(p->states + 4 * (int64_t)r, p->out + (int64_t)r * p->row_floats, p->row_floats);
    _mm_sfence();        // the rows may be device memory behind a write-combining mapping: drained before completion is reported
  }
}

void helper_main(Pool *p) {
  uint64_t last = 0;
  for (;;) {
    int spins = 0;
    while (p->gen.load(std::memory_order_acquire) == last && !p->stop.load(std::memory_order_relaxed)) {
      if (++spins < p->spin_pauses.load(std::memory_order_relaxed)) { _mm_pause(); continue; }
      std::unique_lock<std::mutex> lk(p->m);
      p->sleepers.fetch_add(1);
      p->cv.wait(lk, [&] { return p->gen.load(std::memory_order_acquire) != last || p->stop.load(); });
      p->sleepers.fetch_sub(1);
      spins = 0;
    }
    if (p->stop.load()) return;
    last = p->gen.load(std::memory_order_acquire);
    work(p);
    p->pending.fetch_sub(1, std::memory_order_release);
  }
}
}  // namespace

extern "C" int etm_envgen_abi_version(void) { return ETM_ENVGEN_ABI_VERSION; }
extern "C" int etm_envgen_set_vector(int on) { g_force_scalar = on ? 0 : 1; return g_has_avx512 ? 1 : 0; }

extern "C" int etm_pcg64_fill_f32(uint64_t *state4, float *out, int64_t n) {
  if (!state4 || n < 0 || (n > 0 && !out) || (n & 1)) return ETM_ENVGEN_EINVAL;
  fill_f32(state4, out, n);
  return 0;
}

extern "C" void *etm_envgen_pool_create(int threads) {
  if (threads < 1 || threads > 64) return nullptr;
  Pool *p = new (std::nothrow) Pool();
  if (!p) return nullptr;
  p->n = threads;
  try {
    for (int i = 1; i < threads; ++i) p->helpers.emplace_back(helper_main, p);
  } catch (...) {
    p->n = (int)p->helpers.size() + 1;
  }
  return p;
}

extern "C" int etm_envgen_pool_set_spin(void *pool, int pauses) {
  Pool *p = static_cast<Pool *>(pool);
  if (!p || pauses < 0) return ETM_ENVGEN_EINVAL;
  p->spin_pauses.store(pauses, std::memory_order_relaxed);
  return 0;
}

extern "C" void etm_envgen_pool_destroy(void *pool) {
  Pool *p = static_cast<Pool *>(pool);
  if (!p) return;
  {
    std::lock_guard<std::mutex> lk(p->m);
    p->stop.store(true);
  }
  p->cv.notify_all();
  for (auto &t : p->helpers) t.join();
  delete p;
}

extern "C" int etm_pcg64_fill_rows_f32(void *pool, uint64_t *states, float *out, int64_t row_floats, int rows) {
  Pool *p = static_cast<Pool *>(pool);
  if (!states || rows < 0 || row_floats < 0 || (row_floats & 1) || (rows > 0 && row_floats > 0 && !out)) return ETM_ENVGEN_EINVAL;
  if (rows == 0 || row_floats == 0) return 0;
  if (!p || p->n == 1 || rows == 1) {
    for (int r = 0; r < rows; ++r) fill_f32(states + 4 * (int64_t)r, out + (int64_t)r * row_floats, row_floats);
    return 0;
  }
  p->states = states; p->out = out; p->row_floats = row_floats; p->rows = rows;
  p->next.store(0, std::memory_order_relaxed);
  p->pending.store(p->n - 1, std::memory_order_relaxed);
  p->gen.fetch_add(1, std::memory_order_release);
  if (p->sleepers.load() > 0) {
    { std::lock_guard<std::mutex> lk(p->m); }
    p->cv.notify_all();
  }
  work(p);
  while (p->pending.load(std::memory_order_acquire) != 0) _mm_pause();
  return 0;
}
