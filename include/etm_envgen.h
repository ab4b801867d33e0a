/* C ABI of libetm_envgen.so: the host-side generator of the synthetic benchmark environment (the reference
 * steps its environments in worker processes, /root/reference worker.py:20-48, and its observations come from whatever the
 * environment computes -- here: numpy's default_rng(seed + worker_id).random([3, 84, 84], dtype=float32), restated in C so that
 * every observation of the timed region is a fresh draw).  No device code and no HIP dependency: the numpy-only worker processes
 * of environments/shm_env.py load it as well as the trainer process.  Built by csrc/Makefile from csrc/envgen.cc with g++.
 *
 * Bit-exactness contract: etm_pcg64_fill_f32 writes exactly the floats that
 *     numpy.random.Generator(numpy.random.PCG64()).random(n, dtype=numpy.float32)
 * writes for a bit generator in the same state, and leaves the state where numpy leaves it
 * (tests/test_host_logic.py::test_native_pcg64_stream_is_numpys). */
#ifndef ETM_ENVGEN_H
#define ETM_ENVGEN_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define ETM_ENVGEN_ABI_VERSION 1
#define ETM_ENVGEN_EINVAL (-2)

int etm_envgen_abi_version(void);
/* The AVX-512 form of the fill is used when the host has it (run-time check); etm_envgen_set_vector(0) forces the portable scalar
 * form (tests compare the two), (1) restores the default.  Returns 1 if the host has the vector form, else 0. */
int etm_envgen_set_vector(int on);

/* state4 = {state_hi, state_lo, inc_hi, inc_lo} of a PCG64 (numpy: bit_generator.state["state"]["state" / "inc"], split into 64-bit
 * halves; the generator must have no buffered 32-bit half, has_uint32 == 0).  Writes n float32 (n even: a 64-bit output makes two
 * floats, low half first) to out and advances state4 in place.  Returns 0, or ETM_ENVGEN_EINVAL. */
int etm_pcg64_fill_f32(uint64_t *state4, float *out, int64_t n);

/* `rows` independent streams (states [rows][4]) -> out [rows][row_floats], drawn side by side by the pool's threads (the caller's
 * thread takes part; one job at a time per pool).  pool NULL: on the calling thread.  etm_envgen_pool_create: 1 <= threads <= 64,
 * NULL on failure; helpers spin `pauses` _mm_pause iterations after a job before they sleep (default 40,000 ~ 1 ms; 0 = sleep at once). */
void *etm_envgen_pool_create(int threads);
void etm_envgen_pool_destroy(void *pool);
int etm_envgen_pool_set_spin(void *pool, int pauses);
int etm_pcg64_fill_rows_f32(void *pool, uint64_t *states, float *out, int64_t row_floats, int rows);

#ifdef __cplusplus
}
#endif
#endif
