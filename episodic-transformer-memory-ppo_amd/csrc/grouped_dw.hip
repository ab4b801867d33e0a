// Weight gradients of the optimisation step's dense layers as ONE grouped launch (SURVEY.md section 8 f2 / VERDICT round 2 item 5;
// /root/reference transformer.py:26-29, :103-114 and model.py:97-107 under trainer.py:310 `loss.backward()`).
//
// Every linear map y = x W^T of the blocks (queries, the per-head key / value folds, fc_out, fc) and of the model (linear_embedding,
// lin_policy, lin_value) has the weight gradient  dW [out, in] = dy^T x : a contraction over the N samples of the minibatch
// (K = N = 2048) with a tiny 384 x 384 output.  One such GEMM cannot fill 256 CUs (9 - 12 output tiles; the library splits K over
// workgroups and runs at 0.26 of the fp32 MFMA peak, 15 us each, 19 of them per minibatch step), but the gradients of ALL layers
// are independent of each other once backward has produced the dy's: they are collected during backward and computed HERE, one
// 96 x 128 output tile per workgroup, 216 workgroups at config 3 = one round of the chip, no partial sums in memory.
//
//   C[p] (Ma x Nb, row stride ldc)  =  A[p]^T B[p],   A[p] [N, Ma] (row stride lda) = dy,   B[p] [N, Nb] (row stride ldb) = x
//   (per-head folds: A = the head's hd columns of q / of d ctx, B = the head's d u / z plane, C = the head's rows of dWk / dWv)
//
// Both operands have the contraction index as their ROW index, so a k-step of v_mfma_f32_32x32x2_f32 (lanes 0-31: row n, lanes
// 32-63: row n + 1) reads memory-contiguous runs: lane l loads A[n + l/32][m0 + 3 (l%32) .. + 3) with one 12-byte and
// B[n + l/32][n0 + 4 (l%32) .. + 4) with one 16-byte buffer load -- 2 vector-memory instructions feed 12 MFMAs (the register
// j of a lane belongs to the tile of the rows / columns {3 i + j} / {4 i + j}: a permutation of the output rows and columns that
// the epilogue undoes for free, four consecutive columns per lane = one 16-byte store).  No LDS in the loop; loads are issued
// PD k-steps ahead; rows beyond a wave's range read as zeros through the buffer range check (no tail code).
// The four waves of a workgroup split the N rows (k-range of N / 4 each: one wave per SIMD, 192 accumulator registers) and are
// summed in a fixed order through LDS: ((w0 + w2) + (w1 + w3)) -- deterministic, no atomics.
#include "etm_common.h"

namespace {
constexpr int GD_MT = 3, GD_NT = 4;                 // 32-row tiles along Ma (96) and along Nb (128) per workgroup
constexpr int GD_TM = 32 * GD_MT, GD_TN = 32 * GD_NT;
constexpr int GD_PD = 6;                            // k-steps (2 rows each) in flight per wave
constexpr int GD_MAXP = 84;                         // problems per launch (kernel-argument table: 84 x 48 + 12 bytes of the 4 KB a launch may carry)

struct GdProblem {
  const float *A, *B;
  float *C;
  int lda, ldb, ldc;
  int tiles_n;                                      // Nb / 128
  int tile_start;                                   // index of this problem's first tile in the launch
};
struct GdParams {
  GdProblem p[GD_MAXP];
  int n_problems, n_tiles, N;
};

static_assert(sizeof(GdParams) <= 4096, "kernel arguments of one launch");

// Workgroup b runs on XCD b % 8 (the dispatcher deals consecutive workgroups round the eight dies, each with an L2 of its own), and the
// 9 - 12 tiles of one problem read the SAME two operands: dealt round the dies as they come, every tile pulls its operand columns
// through its die's L2 by itself (PMC rounds 5 / 6: 397 MB fetched per launch against 226 MB of operands even without any sharing).  So
// the tiles are renumbered: die x takes the contiguous run [x T / 8, (x + 1) T / 8) of the launch's tiles -- the tiles of a problem sit on
// one die (two at a run's ends) and walk the sample rows together through its L2.  A bijection for every T (checked on the host for
// T <= 2000); the few blocks of a die that has one workgroup more than its run has tiles take the tiles left over on other dies.
__device__ __forceinline__ int gd_tile_of_block(int b, int n_tiles) {
  const int x = b & 7, slot = b >> 3;
  const int lo = (int)(((long long)x * n_tiles) >> 3), hi = (int)(((long long)(x + 1) * n_tiles) >> 3);      // this die's run
  const int full = n_tiles >> 3;
  const int wgs = full + (x < (n_tiles & 7) ? 1 : 0);        // workgroups the dispatcher gives die x
  const int run = hi - lo;
  if (slot < (run < wgs ? run : wgs)) return lo + slot;
  int k = 0;                                                 // index of this block among the leftover blocks
  for (int y = 0; y < x; ++y) {
    const int wy = full + (y < (n_tiles & 7) ? 1 : 0), ry = (int)(((long long)(y + 1) * n_tiles) >> 3) - (int)(((long long)y * n_tiles) >> 3);
    if (wy > ry) k += wy - ry;
  }
  k += slot - run;
  for (int y = 0; y < 8; ++y) {
    const int ly = (int)(((long long)y * n_tiles) >> 3), ry = (int)(((long long)(y + 1) * n_tiles) >> 3) - ly;
    const int wy = full + (y < (n_tiles & 7) ? 1 : 0);
    if (ry > wy) {
      if (k < ry - wy) return ly + wy + k;
      k -= ry - wy;
    }
  }
  return b;                                                  // (not reached: leftover blocks and leftover tiles are equally many)
}

typedef float f32x3 __attribute__((ext_vector_type(3)));

typedef int i32x4 __attribute__((ext_vector_type(4)));
// The operand loads and their waits are inline assembly: the compiler's own wait-count insertion puts s_waitcnt vmcnt(0) at the
// head of the pipelined loop (it does not carry the per-slot ages around the back edge), which serialises every round behind the
// memory latency.  Written by hand the wait before k-step s of a round is vmcnt(2 (PD - 1)): only the two loads of ITS slot.
// (gfx9 returns loads in order; every "wait" names the registers it guards, so the MFMAs that read them stay behind it.)
__device__ __forceinline__ void gd_load(f32x3 &a, f32x4 &b, i32x4 ra, i32x4 rb, int va, int vb) {
  asm volatile("buffer_load_dwordx3 %0, %2, %3, 0 offen\n\tbuffer_load_dwordx4 %1, %4, %5, 0 offen"
               : "=&v"(a), "=&v"(b) : "v"(va), "s"(ra), "v"(vb), "s"(rb) : "memory");
}
template <int OUTSTANDING>
__device__ __forceinline__ void gd_wait(f32x3 &a, f32x4 &b) {
  asm volatile("s_waitcnt vmcnt(%2)" : "+v"(a), "+v"(b) : "n"(OUTSTANDING));
}
__device__ __forceinline__ void gd_mfma(f32x16 (&acc)[GD_MT][GD_NT], const f32x3 &a, const f32x4 &b) {
#pragma unroll
  for (int i = 0; i < GD_MT; ++i)
#pragma unroll
    for (int j = 0; j < GD_NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
}

__global__ __launch_bounds__(256) void grouped_dw_kernel(const GdParams P) {
  extern __shared__ __attribute__((aligned(16))) float gd_lds[];      // 2 x 48 KB: the accumulators of two waves
  const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, half = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);       // wave-uniform: the descriptors below depend on it
  // tile -> problem (uniform; <= GD_MAXP compares on the scalar unit)
  const int tile = gd_tile_of_block(blockIdx.x, P.n_tiles);
  int pi = 0;
  for (int q = 1; q < P.n_problems; ++q)
    if (tile >= P.p[q].tile_start) pi = q;
  // (the table entry is wave-uniform; say so, or the buffer loads below are wrapped in waterfall loops)
  auto uni = [](int v) { return __builtin_amdgcn_readfirstlane(v); };
  auto uni_ptr = [&](const void *q) {
    const unsigned long long u = (unsigned long long)q;
    return (unsigned long long)(unsigned)uni((int)(u & 0xffffffffu)) | ((unsigned long long)(unsigned)uni((int)(u >> 32)) << 32);
  };
  GdProblem pr = P.p[pi];
  pr.A = (const float *)uni_ptr(pr.A); pr.B = (const float *)uni_ptr(pr.B); pr.C = (float *)uni_ptr(pr.C);
  pr.lda = uni(pr.lda); pr.ldb = uni(pr.ldb); pr.ldc = uni(pr.ldc); pr.tiles_n = uni(pr.tiles_n); pr.tile_start = uni(pr.tile_start);
  const int t = tile - pr.tile_start, tm = t / pr.tiles_n, tn = t - tm * pr.tiles_n;
  const int m0 = tm * GD_TM, n0 = tn * GD_TN;

  // this wave's rows: [r0, r1), an even number of them except in the last wave
  const int N = P.N;
  int rows_w = ((N + 3) / 4 + 1) & ~1;
  const int r0 = min(wave * rows_w, N), r1 = min(r0 + rows_w, N);

  f32x16 acc[GD_MT][GD_NT];
#pragma unroll
  for (int i = 0; i < GD_MT; ++i)
#pragma unroll
    for (int j = 0; j < GD_NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // operands through buffer descriptors that cover exactly this wave's rows: anything beyond them -- the second row of an odd last
  // k-step, the padding k-steps of the last pipeline round -- reads as zeros (buffer range check) and adds nothing
  const unsigned bytes_a = (unsigned)(r1 - r0) * (unsigned)pr.lda * 4u, bytes_b = (unsigned)(r1 - r0) * (unsigned)pr.ldb * 4u;
  const unsigned long long base_a = (unsigned long long)(pr.A + (long long)r0 * pr.lda), base_b = (unsigned long long)(pr.B + (long long)r0 * pr.ldb);
  i32x4 ra = {(int)(base_a & 0xffffffffu), (int)(base_a >> 32), (int)bytes_a, 0x00020000};
  i32x4 rb = {(int)(base_b & 0xffffffffu), (int)(base_b >> 32), (int)bytes_b, 0x00020000};
  etm_rsrc_fence(ra);                                  // (their words come from v_readfirstlane: see etm_common.h)
  etm_rsrc_fence(rb);
  int va = (half * pr.lda + m0 + GD_MT * li) * 4, vb = (half * pr.ldb + n0 + GD_NT * li) * 4;
  const int sa_step = 2 * pr.lda * 4, sb_step = 2 * pr.ldb * 4;
  const int ksteps = (r1 - r0 + 1) >> 1;                           // two rows each
  const int rounds = (ksteps + GD_PD - 1) / GD_PD;

  f32x3 a[GD_PD];
  f32x4 b[GD_PD];
#pragma unroll
  for (int s = 0; s < GD_PD; ++s) { gd_load(a[s], b[s], ra, rb, va, vb); va += sa_step; vb += sb_step; }
  for (int it = 0; it < rounds; ++it) {                            // consume slot s, refill it PD k-steps ahead
#pragma unroll
    for (int s = 0; s < GD_PD; ++s) {
      gd_wait<2 * (GD_PD - 1)>(a[s], b[s]);                        // the 2 (PD - 1) loads issued after this slot's may still fly
      gd_mfma(acc, a[s], b[s]);
      __builtin_amdgcn_sched_barrier(0);                           // the refill may not move above the MFMAs that read the slot
      gd_load(a[s], b[s], ra, rb, va, vb);
      va += sa_step; vb += sb_step;
    }
  }
#pragma unroll
  for (int s = 0; s < GD_PD; ++s) gd_wait<0>(a[s], b[s]);          // the refills of the last round (zeros) land before the registers go

  // ---- cross-wave sum in a fixed order: w0 += w2, w1 += w3 (both through LDS at once), then w0 += w1
  constexpr int GROUPS = GD_MT * GD_NT * 4;                         // 16-byte groups per lane (16 registers = 4 groups per tile)
  auto put = [&](int slot) {
    f32x4 *dst = reinterpret_cast<f32x4 *>(gd_lds) + (long long)slot * GROUPS * 64 + lane;
#pragma unroll
    for (int i = 0; i < GD_MT; ++i)
#pragma unroll
      for (int j = 0; j < GD_NT; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          dst[((i * GD_NT + j) * 4 + g) * 64] = f32x4{acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
  };
  auto add = [&](int slot) {
    const f32x4 *src = reinterpret_cast<const f32x4 *>(gd_lds) + (long long)slot * GROUPS * 64 + lane;
#pragma unroll
    for (int i = 0; i < GD_MT; ++i)
#pragma unroll
      for (int j = 0; j < GD_NT; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 v = src[((i * GD_NT + j) * 4 + g) * 64];
          acc[i][j][4 * g] += v[0]; acc[i][j][4 * g + 1] += v[1]; acc[i][j][4 * g + 2] += v[2]; acc[i][j][4 * g + 3] += v[3];
        }
  };
  if (wave >= 2) put(wave - 2);
  __syncthreads();
  if (wave < 2) add(wave);
  __syncthreads();
  if (wave == 1) put(0);
  __syncthreads();
  if (wave != 0) return;
  add(0);

  // ---- store: register r of tile (i, j) is C[m0 + 3 row(r) + i][n0 + 4 li + j]: the four j of a lane are consecutive columns
  float *crow = pr.C + (long long)m0 * pr.ldc + n0 + GD_NT * li;
#pragma unroll
  for (int i = 0; i < GD_MT; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = GD_MT * mfma32_row(r, lane) + i;
      *reinterpret_cast<f32x4 *>(crow + (long long)row * pr.ldc) = f32x4{acc[i][0][r], acc[i][1][r], acc[i][2][r], acc[i][3][r]};
    }
}
}  // namespace

// 1 when a problem's shape fits the kernel (whole 96 x 128 tiles, 16-byte aligned runs, 32-bit byte offsets).
extern "C" int etm_grouped_dw_supported(int N, int Ma, int Nb, int lda, int ldb, int ldc) {
  if (N < 2 || Ma <= 0 || Nb <= 0 || Ma % GD_TM != 0 || Nb % GD_TN != 0) return 0;
  if (lda < Ma || ldb < Nb || ldc < Nb || lda % 4 != 0 || ldb % 4 != 0 || ldc % 4 != 0) return 0;
  if ((long long)(N + 2) * lda * 4 >= 0x7fffffffLL || (long long)(N + 2) * ldb * 4 >= 0x7fffffffLL) return 0;
  return 1;
}
extern "C" int etm_grouped_dw_max_problems(void) { return GD_MAXP; }

// C[p] = A[p]^T B[p] for n_problems independent problems over the same N rows, one launch.  A / B / C: host arrays of device
// pointers; dims: host array of 5 ints per problem (Ma, Nb, lda, ldb, ldc).  Every problem must satisfy etm_grouped_dw_supported;
// A and B 4-byte, C and all row starts 16-byte aligned (B and C are accessed with 16-byte instructions).  C is overwritten.
extern "C" int etm_grouped_dw(const float *const *A, const float *const *B, float *const *C, const int32_t *dims, int n_problems, int N,
                              void *stream) {
  (void)hipGetLastError();
  if (!A || !B || !C || !dims || n_problems <= 0 || N <= 0) return ETM_EINVAL;
  if (n_problems > GD_MAXP) return ETM_EUNSUPPORTED;
  GdParams P{};
  int tiles = 0;
  for (int i = 0; i < n_problems; ++i) {
    const int Ma = dims[5 * i], Nb = dims[5 * i + 1], lda = dims[5 * i + 2], ldb = dims[5 * i + 3], ldc = dims[5 * i + 4];
    if (!A[i] || !B[i] || !C[i]) return ETM_EINVAL;
    if (!etm_grouped_dw_supported(N, Ma, Nb, lda, ldb, ldc)) return ETM_EUNSUPPORTED;
    if (((uintptr_t)A[i] % 4) || ((uintptr_t)B[i] % 16) || ((uintptr_t)C[i] % 16)) return ETM_EINVAL;
    P.p[i] = GdProblem{A[i], B[i], C[i], lda, ldb, ldc, Nb / GD_TN, tiles};
    tiles += (Ma / GD_TM) * (Nb / GD_TN);
  }
  P.n_problems = n_problems; P.n_tiles = tiles; P.N = N;
  hipStream_t st = (hipStream_t)stream;
  constexpr size_t lds = 2 * (size_t)GD_MT * GD_NT * 16 * 64 * sizeof(float);      // 96 KB
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void *)grouped_dw_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  EtmProfScope prof(ETM_K_GROUPED_DW, st);
  hipLaunchKernelGGL(grouped_dw_kernel, dim3((unsigned)tiles), dim3(256), lds, st, P);
  return etm_launch_status();
}
