// Minibatch gather of the small per-sample fields in ONE launch (the reference indexes every field of the flattened buffer with
// the minibatch indices, /root/reference buffer.py:84-91 `samples_flat[key][mini_batch_indices]`: eight index kernels of ~5 us
// each per minibatch step here).  Fields are byte rows: dst[f][i, :] = src[f][idx[i], :], rows of row_bytes[f] (multiples of
// 4 bytes, contiguous).  Pure data movement: bit-exact.
#include "etm_common.h"

namespace {
constexpr int GR_MAXF = 16;
struct GatherP {
  const unsigned *src[GR_MAXF];
  unsigned *dst[GR_MAXF];
  int words[GR_MAXF];          // row length in 4-byte words
  const long long *idx;
  long long n, src_rows;
};
__global__ __launch_bounds__(256) void gather_rows_kernel(const GatherP p) {
  const int f = blockIdx.y;
  const int words = p.words[f];
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e >= p.n * words) return;
  const long long i = e / words;
  const int w = (int)(e - i * words);
  long long r = p.idx[i];
  r = r < 0 ? 0 : (r >= p.src_rows ? p.src_rows - 1 : r);        // out-of-range indices are the caller's error; never fault
  p.dst[f][e] = p.src[f][r * words + w];
}
}  // namespace

// dst[f][i, :] = src[f][idx[i], :] for f < n_fields (<= 16), i < n; src[f] has src_rows rows of row_bytes[f] bytes (% 4 == 0).
extern "C" int etm_gather_rows(const void *const *src, void *const *dst, const int64_t *row_bytes, int n_fields, const int64_t *idx, int64_t n,
                               int64_t src_rows, void *stream) {
  (void)hipGetLastError();
  if (!src || !dst || !row_bytes || !idx || n_fields <= 0 || n <= 0 || src_rows <= 0) return ETM_EINVAL;
  if (n_fields > GR_MAXF) return ETM_EUNSUPPORTED;
  GatherP p{};
  long long most = 0;
  for (int f = 0; f < n_fields; ++f) {
    if (!src[f] || !dst[f] || row_bytes[f] <= 0) return ETM_EINVAL;
    if (row_bytes[f] % 4 != 0 || ((uintptr_t)src[f] % 4) != 0 || ((uintptr_t)dst[f] % 4) != 0 || row_bytes[f] / 4 > (1 << 20)) return ETM_EUNSUPPORTED;
    p.src[f] = static_cast<const unsigned *>(src[f]);
    p.dst[f] = static_cast<unsigned *>(dst[f]);
    p.words[f] = (int)(row_bytes[f] / 4);
    if (n * p.words[f] > most) most = n * p.words[f];
  }
  p.idx = (const long long *)idx; p.n = n; p.src_rows = src_rows;
  EtmProfScope prof(ETM_K_GATHER_ROWS, (hipStream_t)stream);
  hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)((most + 255) / 256), (unsigned)n_fields), dim3(256), 0, (hipStream_t)stream, p);
  return etm_launch_status();
}
