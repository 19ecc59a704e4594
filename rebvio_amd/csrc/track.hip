// Edge tracking on gfx950: one thread per keyline, wavefront (64-lane) butterfly reductions, fixed-order
// block/grid reductions (deterministic), order-independent restatements of the reference's sequential
// "last writer wins" rules.
//
// Reference semantics (baumlin/rebvio): edge_map.cpp:39-259; core.cpp:39-261,417-456. Compiled with
// -ffp-contract=off; float/double promotion follows the reference expression by expression (see the
// oracle's header for the rules).
//
// Bound: HBM/L2 gather bandwidth (keyline SoA streams + distance-field / mask / matched-keyline gathers);
// no dense contraction exists on this path (largest product is 6x6), so MFMA does not apply.
#include "common.hpp"
#include "glue_dev.hpp"

#include <cstdlib>
#include <cstring>

namespace rh {

constexpr float kRhoMax = 20.0f;   // types/keyline.hpp:17
constexpr float kRhoMin = 1e-3f;   // types/keyline.hpp:18
constexpr float kRhoInit = 1.0f;   // types/keyline.hpp:19

struct Mat3 {
  float a[9];
};
struct Vec3 {
  float a[3];
};

__device__ __forceinline__ float wave_sum_f(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
// Wave total through the DPP data path (no LDS crossbar traffic, unlike __shfl_xor = ds_bpermute): Hillis-Steele steps
// inside each row of 16 lanes (row_shr 1,2,4,8, zero fill), then row_bcast15 into rows 1 and 3, row_bcast31 into rows
// 2 and 3. The total is valid in LANE 63 only; the tree is fixed, so results are reproducible.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add_f(float v) {
  const int moved = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xF, true);
  return v + __int_as_float(moved);
}
__device__ __forceinline__ float wave_total63_f(float v) {
  v = dpp_add_f<0x111, 0xF>(v);  // row_shr:1
  v = dpp_add_f<0x112, 0xF>(v);  // row_shr:2
  v = dpp_add_f<0x114, 0xF>(v);  // row_shr:4
  v = dpp_add_f<0x118, 0xF>(v);  // row_shr:8
  v = dpp_add_f<0x142, 0xA>(v);  // row_bcast:15 -> rows 1, 3
  v = dpp_add_f<0x143, 0xC>(v);  // row_bcast:31 -> rows 2, 3
  return v;
}
// The same tree for N independent values at once with fused v_add_f32_dpp (one VALU op per step and value instead of
// mov_dpp + add). Steps are emitted value-interleaved, so consecutive DPP reads of one register are N >= 8 instructions
// apart (gfx9 needs 2 wait states between a VALU write and a DPP read of the same VGPR; the leading s_nop covers the
// producer of the inputs). Disabled rows of the row_bcast steps keep their value, which is what the scan needs.
template <int N>
__device__ __forceinline__ void wave_total63_fN(float (&v)[N]) {
  static_assert(N >= 8, "dependent DPP ops must be at least 2 wait states apart");
  asm volatile("s_nop 1");
#define RH_DPP_STEP(ctrl)                                                             \
  _Pragma("unroll") for (int k = 0; k < N; ++k)                                       \
      asm volatile("v_add_f32_dpp %0, %0, %0 " ctrl : "+v"(v[k]));
  RH_DPP_STEP("row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0")
  RH_DPP_STEP("row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:0")
  RH_DPP_STEP("row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:0")
  RH_DPP_STEP("row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:0")
  RH_DPP_STEP("row_bcast:15 row_mask:0xa bank_mask:0xf")
  RH_DPP_STEP("row_bcast:31 row_mask:0xc bank_mask:0xf")
#undef RH_DPP_STEP
}
__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// Order-preserving float -> uint map (for atomicMax on rho).
__device__ __forceinline__ unsigned order_key(float f) {
  const unsigned b = __float_as_uint(f);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

// rotateKeylines for one keyline (edge_map.cpp:59-70). makeVector(float, float, 1.0) mixes float and double
// arguments: TooN's double overload is selected, so the matrix-vector product is accumulated in double and rounded
// to float once per component.
__device__ __forceinline__ void rotate_one(const Mat3& R, float fm, float2& pi, float2& rs, float2& g) {
  const double v0 = (double)(pi.x / fm), v1 = (double)(pi.y / fm);
  float q[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    double s = 0.0;
    s += (double)R.a[i * 3 + 0] * v0;
    s += (double)R.a[i * 3 + 1] * v1;
    s += (double)R.a[i * 3 + 2] * 1.0;
    q[i] = (float)s;
  }
  if (fabsf(q[2]) > 0.0f) {
    pi.x = q[0] / q[2] * fm;
    pi.y = q[1] / q[2] * fm;
    rs.x /= q[2];
    rs.y /= q[2];
  }
  float gq[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    double s = 0.0;
    s += (double)R.a[i * 3 + 0] * (double)g.x;
    s += (double)R.a[i * 3 + 1] * (double)g.y;
    s += (double)R.a[i * 3 + 2] * 0.0;
    gq[i] = (float)s;
  }
  g = make_float2(gq[0], gq[1]);
}

// ---- EdgeMap::rotateKeylines (edge_map.cpp:58-71) [+ estimateQuantile histogram, :39-46] -------------------
// makeVector(float, float, 1.0) mixes float and double arguments: TooN's double overload is selected, so the
// matrix-vector product is accumulated in double and rounded to float once per component.
__global__ __launch_bounds__(256) void k_rotate(KParams p, MapDev m, Mat3 R, int* __restrict__ hist, int hist_bins,
                                                int zero_dm) {
  __shared__ int sh[128];
  const int idx = blockIdx.x * 256 + threadIdx.x;
  float2 pi = m.pos_img[idx];  // bound-free early loads
  float2 rs = m.rs[idx];
  float2 g = m.grad[idx];
  const int n = m.st->n;
  if (hist) {
    if (threadIdx.x < 128) sh[threadIdx.x] = 0;
    __syncthreads();
  }
  if (zero_dm && blockIdx.x == 0 && threadIdx.x == 0) {
    m.st->dm_matches = 0;
    m.st->dm_kf = 0;
    m.st->reg_count = 0;
    m.st->dm_queued = 0;
  }
  if (idx < n) {
    rotate_one(R, p.fm, pi, rs, g);
    m.pos_img[idx] = pi;
    m.rs[idx] = rs;
    m.grad[idx] = g;
    if (hist) {
      m.residual[idx] = 0.f;  // minimizeVel starts from residuals[] = {0} (core.cpp:158)
      int i = cvtt_f32(hist_bins * (rs.y - kRhoMin) / (kRhoMax - kRhoMin));
      i = (i > hist_bins - 1) ? (hist_bins - 1) : i;
      i = (i < 0) ? 0 : i;
      atomicAdd(&sh[i], 1);
    }
  }
  if (hist) {
    __syncthreads();
    if ((int)threadIdx.x < hist_bins && sh[threadIdx.x]) atomicAdd(&hist[threadIdx.x], sh[threadIdx.x]);
  }
}

// Histogram only (stand-alone EdgeMap::estimateQuantile).
__global__ __launch_bounds__(256) void k_sigma_hist(MapDev m, int* __restrict__ hist, int hist_bins) {
  __shared__ int sh[128];
  const int n = m.st->n;
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (threadIdx.x < 128) sh[threadIdx.x] = 0;
  __syncthreads();
  if (idx < n) {
    int i = cvtt_f32(hist_bins * (m.rs[idx].y - kRhoMin) / (kRhoMax - kRhoMin));
    i = (i > hist_bins - 1) ? (hist_bins - 1) : i;
    i = (i < 0) ? 0 : i;
    atomicAdd(&sh[i], 1);
  }
  __syncthreads();
  if ((int)threadIdx.x < hist_bins && sh[threadIdx.x]) atomicAdd(&hist[threadIdx.x], sh[threadIdx.x]);
}

// Tail of estimateQuantile (edge_map.cpp:47-55): first bin whose running count (checked before adding the
// bin) exceeds percentile * size.
__device__ __forceinline__ float quantile_from_hist(const int* __restrict__ hist, int bins, float pct, int n) {
  float sigma_rho = 1e3f;
  int a = 0;
  for (int i = 0; i < bins; ++i) {
    if ((float)a > pct * (float)n) {
      sigma_rho = float(i) * (kRhoMax - kRhoMin) / float(bins) + kRhoMin;
      break;
    }
    a += hist[i];
  }
  return sigma_rho;
}

// The same on a whole wave (bins <= 128, two per lane): an exclusive prefix over the lanes' pairs replaces the up to `bins`
// dependent LDS reads of the loop above (1-2 us on a lone lane at the head of the persistent LM kernels). Integer counts:
// the same bin whatever the order. Every lane of ONE wave calls it; every lane returns the result.
__device__ __forceinline__ float quantile_from_hist_wave(const int* __restrict__ hist, int bins, float pct, int n, int lane) {
  const int h0 = (2 * lane < bins) ? hist[2 * lane] : 0;
  const int h1 = (2 * lane + 1 < bins) ? hist[2 * lane + 1] : 0;
  int incl = h0 + h1;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int up = __shfl_up(incl, d);
    if (lane >= d) incl += up;
  }
  const int before0 = incl - (h0 + h1), before1 = before0 + h0;  // running count when the loop reaches bin 2 lane / 2 lane + 1
  const float thr = pct * (float)n;
  const bool c0 = 2 * lane < bins && (float)before0 > thr;
  const bool c1 = 2 * lane + 1 < bins && (float)before1 > thr;
  const unsigned long long m = __ballot(c0 || c1);
  if (m == 0ull) return 1e3f;
  const int fl = __ffsll((long long)m) - 1;
  const int first = 2 * fl + (__shfl((int)c0, fl) ? 0 : 1);
  return float(first) * (kRhoMax - kRhoMin) / float(bins) + kRhoMin;
}

__global__ void k_quantile(MapDev m, const int* __restrict__ hist, int bins, float pct, float* __restrict__ out) {
  if (threadIdx.x == 0 && blockIdx.x == 0) *out = quantile_from_hist(hist, bins, pct, m.st->n);
}

// ---- 3x3 helpers with the reference's operation order -------------------------------------------------------
// TooN::determinant for N = 3: Gaussian elimination with partial pivoting.
__device__ float det3_elim(const float* A_) {
  float A[3][3];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) A[i][j] = A_[i * 3 + j];
  float det = 1;
  for (int i = 0; i < 3; ++i) {
    int argmax = i;
    float mx = fabsf(A[i][i]);
    for (int ii = i + 1; ii < 3; ++ii)
      if (fabsf(A[ii][i]) > mx) {
        mx = fabsf(A[ii][i]);
        argmax = ii;
      }
    const float pivot = A[argmax][i];
    if (argmax != i) {
      det *= -1;
      for (int ii = i; ii < 3; ++ii) {
        const float t = A[i][ii];
        A[i][ii] = A[argmax][ii];
        A[argmax][ii] = t;
      }
    }
    det *= A[i][i];
    if (det == 0) return 0;
    for (int u = i + 1; u < 3; ++u) {
      const float factor = A[u][i] / pivot;
      for (int uu = i; uu < 3; ++uu) A[u][uu] = A[u][uu] - factor * A[i][uu];
    }
  }
  return det;
}

// types::invert (types/definitions.hpp:40-53)
__device__ void invert3(const float* m, float* o) {
  o[0] = m[4] * m[8] - m[5] * m[7];
  o[1] = m[2] * m[7] - m[1] * m[8];
  o[2] = m[1] * m[5] - m[2] * m[4];
  o[3] = m[5] * m[6] - m[3] * m[8];
  o[4] = m[0] * m[8] - m[2] * m[6];
  o[5] = m[2] * m[3] - m[0] * m[5];
  o[6] = m[3] * m[7] - m[4] * m[6];
  o[7] = m[1] * m[6] - m[0] * m[7];
  o[8] = m[0] * m[4] - m[1] * m[3];
  const float d = det3_elim(m);
  for (int i = 0; i < 9; ++i) o[i] = o[i] / d;
}

__device__ __forceinline__ void sym6_to_9(const float* J, float* M) {
  M[0] = J[0]; M[1] = J[3]; M[2] = J[4];
  M[3] = J[3]; M[4] = J[1]; M[5] = J[5];
  M[6] = J[4]; M[7] = J[5]; M[8] = J[2];
}

// One Levenberg-Marquardt bookkeeping step of Core::minimizeVel (core.cpp:161-185), run redundantly and
// identically by thread 0 of every workgroup; `red` = fixed-order sum of the previous call's block records.
// call: index of the tryVel evaluation about to run (0..iterations); final: no further evaluation follows.
// The state lives in LDS and `red` may alias it as far as the compiler can tell, so working on it in place turns every
// field access into its own LDS round trip on a lone thread; the step therefore runs on register copies (same
// operations in the same order) and writes the state back once.
__device__ void lm_step_regs(LmState& s, const float (&red)[10], int call, bool final_only);
__device__ void lm_step(LmState& s_lds, const float* red_lds, int call, bool final_only) {
  LmState t = s_lds;
  float r[10];
#pragma unroll
  for (int i = 0; i < 10; ++i) r[i] = red_lds[i];
  lm_step_regs(t, r, call, final_only);
  s_lds = t;
}
// core.cpp:180-183: the matrix JtJ + u * Identity as TooN forms it (every element gets its term), and the trial point from
// its inverse
__device__ __forceinline__ void lm_step_matrix(const LmState& s, float (&M)[9]) {
  sym6_to_9(s.JtJ, M);
  M[0] = M[0] + 1.0f * s.u;
  M[4] = M[4] + 1.0f * s.u;
  M[8] = M[8] + 1.0f * s.u;
  M[1] = M[1] + 0.0f * s.u; M[2] = M[2] + 0.0f * s.u; M[3] = M[3] + 0.0f * s.u;
  M[5] = M[5] + 0.0f * s.u; M[6] = M[6] + 0.0f * s.u; M[7] = M[7] + 0.0f * s.u;
}
__device__ __forceinline__ void lm_step_apply(LmState& s, const float (&inv)[9]) {
  const float neg[3] = {-s.JtF[0], -s.JtF[1], -s.JtF[2]};
  for (int i = 0; i < 3; ++i) {
    float acc = 0;
    for (int k = 0; k < 3; ++k) acc += inv[i * 3 + k] * neg[k];
    s.h[i] = acc;
  }
  for (int i = 0; i < 3; ++i) s.Vnew[i] = s.vel[i] + s.h[i];
}
// the accept / reject bookkeeping of one step (core.cpp:161-179)
__device__ __forceinline__ void lm_step_book(LmState& s, const float (&red)[10], int call) {
  if (call == 1) {
    s.F = red[0];
    for (int i = 0; i < 6; ++i) s.JtJ[i] = red[1 + i];
    for (int i = 0; i < 3; ++i) s.JtF[i] = red[7 + i];
    s.v = 2.0f;
    const float tau = 1e-3f;
    float mx = s.JtJ[0];
    for (int i = 1; i < 6; ++i)
      if (s.JtJ[i] > mx) mx = s.JtJ[i];
    s.u = tau * mx;
    s.accept_mask = 0;
  } else if (call >= 2) {
    const float Fnew = red[0];
    double den = 0.0;
    for (int i = 0; i < 3; ++i) den += (0.5 * (double)s.h[i]) * (double)(s.u * s.h[i] - s.JtF[i]);
    const float gain = (float)((double)(s.F - Fnew) / den);
    if (gain > 0.0f) {
      s.F = Fnew;
      for (int i = 0; i < 3; ++i) s.vel[i] = s.Vnew[i];
      for (int i = 0; i < 6; ++i) s.JtJ[i] = red[1 + i];
      for (int i = 0; i < 3; ++i) s.JtF[i] = red[7 + i];
      const double g = (double)gain;
      const double c = 1.0 - ((2.0 * g - 1.0) * (2.0 * g - 1.0) * (2.0 * g - 1.0));
      s.u = (float)((double)s.u * (0.33 > c ? 0.33 : c));
      s.v = 2.0f;
      s.accept_mask |= 1 << (call - 2);
    } else {
      s.u *= s.v;
      s.v = (float)((double)s.v * 2.0);
    }
  }
}
__device__ __forceinline__ void lm_step_regs(LmState& s, const float (&red)[10], int call, bool final_only) {
  lm_step_book(s, red, call);
  if (call >= 1 && !final_only) {
    float M[9], inv[9];
    lm_step_matrix(s, M);
    invert3(M, inv);
    lm_step_apply(s, inv);
  }
}

// The same step on a whole wave (every lane of wave 0 of a workgroup calls it, every workgroup alike): the bookkeeping on
// registers in every lane, the 3x3 inverse with one cofactor per lane (nine divisions side by side instead of in a row: the
// lone thread's step cost 0.9 us three times per launch), the new trial point from the inverse in LDS. Same operations per
// element, same bits. inv_lds: 9 floats of LDS.
__device__ __forceinline__ void lm_step_wave(LmState& s_lds, const float* red_lds, int call, bool final_only, float* inv_lds, int lane) {
  LmState t = s_lds;
  float r[10];
#pragma unroll
  for (int i = 0; i < 10; ++i) r[i] = red_lds[i];
  lm_step_book(t, r, call);
  if (call >= 1 && !final_only) {
    float M[9];
    lm_step_matrix(t, M);
    const float d = glue_det3(M);
    float o;
    switch (lane) {
      case 0: o = M[4] * M[8] - M[5] * M[7]; break;
      case 1: o = M[2] * M[7] - M[1] * M[8]; break;
      case 2: o = M[1] * M[5] - M[2] * M[4]; break;
      case 3: o = M[5] * M[6] - M[3] * M[8]; break;
      case 4: o = M[0] * M[8] - M[2] * M[6]; break;
      case 5: o = M[2] * M[3] - M[0] * M[5]; break;
      case 6: o = M[3] * M[7] - M[4] * M[6]; break;
      case 7: o = M[1] * M[6] - M[0] * M[7]; break;
      default: o = M[0] * M[4] - M[1] * M[3]; break;
    }
    glue_wave_sync();  // (the previous step's inverse has been read by every lane)
    if (lane < 9) inv_lds[lane] = o / d;
    glue_wave_sync();
    float inv[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) inv[i] = inv_lds[i];
    lm_step_apply(t, inv);
  }
  if (lane == 0) s_lds = t;
}

// lm_step_wave for the step in front of a run of speculative evaluations (k_lm_chain_spec): besides the real state (q = 0) the
// wave forms, side by side in its four 16-lane rows, the states q = 1 .. 3 the NEXT steps start from if every one of them
// rejects - q rejections' (u, v) updates on the state the bookkeeping left (core.cpp:180-183: a rejection changes nothing
// else), then the trial point from that state's own inverse. One inverse's latency instead of two in a row; same operations
// per state as lm_step_book on a NaN score + lm_step_matrix + the inverse + lm_step_apply, same bits. stc[q] for q < nstates.
__device__ __forceinline__ void lm_step_wave_states(LmState& s_lds, const float* red_lds, int call, float (*inv_lds)[9], int lane,
                                                    LmState* stc, int nstates) {
  LmState t = s_lds;
  float r[10];
#pragma unroll
  for (int i = 0; i < 10; ++i) r[i] = red_lds[i];
  lm_step_book(t, r, call);
  const int q = lane >> 4, e = lane & 15;
  for (int j = 0; j < q; ++j) {
    t.u *= t.v;
    t.v = (float)((double)t.v * 2.0);
  }
  float M[9];
  lm_step_matrix(t, M);
  const float d = glue_det3(M);
  float o;
  switch (e) {
    case 0: o = M[4] * M[8] - M[5] * M[7]; break;
    case 1: o = M[2] * M[7] - M[1] * M[8]; break;
    case 2: o = M[1] * M[5] - M[2] * M[4]; break;
    case 3: o = M[5] * M[6] - M[3] * M[8]; break;
    case 4: o = M[0] * M[8] - M[2] * M[6]; break;
    case 5: o = M[2] * M[3] - M[0] * M[5]; break;
    case 6: o = M[3] * M[7] - M[4] * M[6]; break;
    case 7: o = M[1] * M[6] - M[0] * M[7]; break;
    default: o = M[0] * M[4] - M[1] * M[3]; break;
  }
  glue_wave_sync();  // (the previous step's inverse has been read by every lane)
  if (e < 9) inv_lds[q][e] = o / d;
  glue_wave_sync();
  float inv[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) inv[i] = inv_lds[q][i];
  lm_step_apply(t, inv);
  if (e == 0 && q < nstates) stc[q] = t;
  if (lane == 0) s_lds = t;
}

// Shared prologue: fixed-order reduction of the previous tryVel call's block records + carry-in of the
// "last written fi" (oracle header, H3) for this workgroup. The records are staged in LDS by one coalesced
// pass of the whole workgroup (one memory round trip), then summed in block order from LDS.

// Stages the records of ALL launched workgroups (grid size is known without reading the keyline count, so these
// loads are in flight together with the first own-keyline loads); records beyond the live count are never summed.
__device__ void stage_prev_records(const float* __restrict__ part_prev, int grid_blocks, float* rec /*LDS*/) {
  const int total = min(grid_blocks, kMaxRecBlocks) * kPartStride;
  for (int i = threadIdx.x; i < total; i += blockDim.x) rec[i] = part_prev[i];
}

// Fixed-order sum of the staged records with 160 threads: sum k (0..9) lives in the 16-lane row k, lane j adds the
// records b = j, j+16, ... in ascending order, a DPP scan inside the row (row_shr 1,2,4,8) leaves the total in lane 15.
// `first_group` = index of this workgroup's first record group, `groups` = record groups it owns (carry_in[g] for each).
__device__ __forceinline__ void reduce_staged_records(const float* rec, int nblocks, float* red /*shared[16]*/, float* carry_in,
                                                      int first_group, int groups) {
  const int t = threadIdx.x;
  if (t < 160) {
    const int k = t >> 4, j = t & 15;
    float acc = 0.f;
    for (int b = j; b < nblocks; b += 16) acc += rec[b * kPartStride + k];
    acc = dpp_add_f<0x111, 0xF>(acc);
    acc = dpp_add_f<0x112, 0xF>(acc);
    acc = dpp_add_f<0x114, 0xF>(acc);
    acc = dpp_add_f<0x118, 0xF>(acc);
    if (j == 15) red[k] = acc;
  } else if (t >= 192 && t < 192 + groups) {
    const int g = t - 192;
    float cv = 0.f;
    for (int b = first_group + g - 1; b >= 0; --b)
      if (rec[b * kPartStride + 10] != 0.f) {
        cv = rec[b * kPartStride + 11];
        break;
      }
    carry_in[g] = fabsf(cv);
  }
}

// The ten sums of one staged record set by the 160 threads t0 .. t0 + 159 (same order as reduce_staged_records).
__device__ __forceinline__ void reduce_staged_sums(const float* rec, int nblocks, float* red /*shared[16]*/, int t0) {
  const int t = (int)threadIdx.x - t0;
  if (t >= 0 && t < 160) {
    const int k = t >> 4, j = t & 15;
    float acc = 0.f;
    for (int b = j; b < nblocks; b += 16) acc += rec[b * kPartStride + k];
    acc = dpp_add_f<0x111, 0xF>(acc);
    acc = dpp_add_f<0x112, 0xF>(acc);
    acc = dpp_add_f<0x114, 0xF>(acc);
    acc = dpp_add_f<0x118, 0xF>(acc);
    if (j == 15) red[k] = acc;
  }
}

// Per-keyline body of Core::tryVel / calculatefJ / testfk (core.cpp:39-148), shared by the per-call kernel and the
// persistent LM kernel.
struct TvIn {
  float gn;
  float2 rs, pi, g2;
  unsigned nmatches;
  // velocity-independent sub-expressions of calculatefJ, evaluated once per keyline instead of once per evaluation
  // (same operands, same operations -> same bits): 1/rho and 1/sigma in double, search_range / sigma
  double inv_rho_d, inv_sig_d;
  float range_over_sig;
};
__device__ __forceinline__ TvIn make_tvin(const KParams& p, float gn, float2 rs, float2 pi, float2 g2, unsigned nmatches) {
  TvIn k;
  k.gn = gn;
  k.rs = rs;
  k.pi = pi;
  k.g2 = g2;
  k.nmatches = nmatches;
  k.inv_rho_d = 1.0 / (double)rs.x;
  k.inv_sig_d = 1.0 / (double)rs.y;
  k.range_over_sig = p.search_range / rs.y;
  return k;
}
struct TvOut {
  float f, jx, jy, jz, fi, res_out;
  int mfwd;
  bool contrib, matched, need_carry, wrote_res;
};
// try_vel_eval in three steps, so that a caller with several trial velocities at hand (k_lm_chain_spec) can have the
// dependent gathers of all of them in flight together: (1) projection of the keyline under the trial velocity - no memory,
// nothing that depends on the residual; (2) the distance-field cell, then the geometry of the keyline it names; (3) the
// arithmetic, which is where the residual of the previous evaluation (the IRLS weight) comes in.
struct TvProj {
  float rho_p, p_x, p_y, p_xc, p_yc;
  int cell;  // y * cols + x, or -1 when there is no cell to look at (keyline skipped, or penalised before the lookup)
  bool skip, penalty1;
};
struct TvGeo {
  float2 g1, pn;
  float gnn;
  int id;  // -1: empty cell
};
__device__ __forceinline__ TvProj tv_project(const KParams& p, const TvIn& k, float vx, float vy, float vz, float srm, float thr,
                                             unsigned min_matches) {
  TvProj o{};
  o.cell = -1;
  o.skip = (thr > 0.0f && k.gn < thr) || (k.rs.y > srm) || (k.nmatches < min_matches);
  if (o.skip) return o;
  const float2 pi = k.pi;
  const float z_p = (float)(k.inv_rho_d + (double)vz);
  if (z_p <= 0.0f) {
    o.penalty1 = true;
  } else {
    o.rho_p = (float)(1.0 / (double)z_p);
    o.p_x = o.rho_p * (vx * p.fm - vz * pi.x) + pi.x;
    o.p_y = o.rho_p * (vy * p.fm - vz * pi.y) + pi.y;
    o.p_xc = o.p_x + p.cx;
    o.p_yc = o.p_y + p.cy;
    const int x = cvtt_f64((double)o.p_xc + 0.5);
    const int y = cvtt_f64((double)o.p_yc + 0.5);
    if (x < 1 || y < 1 || (unsigned)x >= (unsigned)p.cols - 1u || (unsigned)y >= (unsigned)p.rows - 1u)
      o.penalty1 = true;
    else
      o.cell = y * p.cols + x;
  }
  return o;
}
__device__ __forceinline__ unsigned tv_cell(const MapDev& nm, const TvProj& pj) { return pj.cell >= 0 ? nm.df[(size_t)pj.cell] : kDfEmpty; }
__device__ __forceinline__ TvGeo tv_geometry(const KParams& p, const MapDev& nm, unsigned key) {
  TvGeo g{};
  g.id = -1;
  if (key != kDfEmpty) {
    g.id = (int)((kDfSeqMask - (key & kDfSeqMask)) / (unsigned)p.df_nr);
    g.g1 = nm.grad[g.id];
    g.pn = nm.pos[g.id];  // issued with g1: one gather round trip
    g.gnn = nm.gnorm[g.id];
  }
  return g;
}
// (3a) what does not depend on the residual: the similarity test of the cell's keyline (testfk), fi and the gradient of f
struct TvMatch {
  float fi, df_dx, df_dy, f0;  // f0: f before the IRLS weight
  int mfwd;
  bool matched, need_carry;
};
__device__ __forceinline__ TvMatch tv_match(const KParams& p, const TvIn& k, const TvProj& pj, const TvGeo& ge) {
  TvMatch m{};
  m.mfwd = -1;
  if (pj.skip || pj.penalty1) return m;
  const float gn = k.gn;
  const float2 rs = k.rs, g2 = k.g2;
  if (ge.id >= 0) {
    const float2 g1 = ge.g1;
    const float norm_squared = gn * gn;
    const float dot_product = g1.x * g2.x + g1.y * g2.y;
    if (!(fabsf(dot_product - norm_squared) > p.match_treshold * norm_squared)) {
      const float dx = pj.p_xc - ge.pn.x;
      const float dy = pj.p_yc - ge.pn.y;
      const float gnx = g1.x / ge.gnn;
      const float gny = g1.y / ge.gnn;
      m.fi = (dx * gnx + dy * gny);
      m.df_dx = gnx / rs.y;
      m.df_dy = gny / rs.y;
      m.mfwd = ge.id;
      m.f0 = m.fi / rs.y;
      m.matched = true;
    }
  }
  if (!m.matched) {
    m.f0 = k.range_over_sig;
    m.need_carry = true;
  }
  return m;
}
// (3b) the IRLS weight from the previous evaluation's residual, and the weighted terms
__device__ __forceinline__ TvOut tv_weight(const KParams& p, const TvIn& k, const TvProj& pj, const TvMatch& m, float res_in, float cin) {
  float f = 0.f, jx = 0.f, jy = 0.f, jz = 0.f, res_out = 0.f;
  bool contrib = false, wrote_res = false;
  if (!pj.skip) {
    float res = res_in;
    if (res == kResidualCarry) res = cin;
    res_out = res;
    float weight = 1.0f;
    if (res > p.reweight_distance) weight = p.reweight_distance / res;
    contrib = true;
    if (pj.penalty1) {
      f = (float)((k.inv_sig_d * (double)p.search_range) * (double)weight);
    } else {
      f = m.f0;
      f *= weight;
      jx = pj.rho_p * p.fm * m.df_dx * weight;
      jy = pj.rho_p * p.fm * m.df_dy * weight;
      jz = -pj.rho_p * (pj.p_x * m.df_dx + pj.p_y * m.df_dy) * weight;
      if (m.matched) res_out = fabsf(m.fi);
    }
    wrote_res = !m.need_carry;
  }
  return TvOut{f, jx, jy, jz, m.fi, res_out, m.mfwd, contrib, m.matched, m.need_carry, wrote_res};
}
__device__ __forceinline__ TvOut tv_finish(const KParams& p, const TvIn& k, const TvProj& pj, const TvGeo& ge, float res_in, float cin) {
  return tv_weight(p, k, pj, tv_match(p, k, pj, ge), res_in, cin);
}
__device__ __forceinline__ TvOut try_vel_eval(const KParams& p, const MapDev& nm, const TvIn& k, float res_in, float cin, float vx,
                                              float vy, float vz, float srm, float thr, unsigned min_matches) {
  const TvProj pj = tv_project(p, k, vx, vy, vz, srm, thr, min_matches);
  const TvGeo ge = tv_geometry(p, nm, tv_cell(nm, pj));
  return tv_finish(p, k, pj, ge, res_in, cin);
}

// ---- Core::tryVel + calculatefJ + testfk (core.cpp:39-148) ---------------------------------------------------
// mode_lm = 0: evaluate at st_in->Vnew with st_in->sigma_rho_min (stand-alone tryVel).
// mode_lm = 1: call `call` of minimizeVel; prologue derives the LM state from the previous call's records.
// last: the evaluation whose side effects persist (core.cpp:166-185 never re-evaluates an accepted point):
//       matched keylines publish forwardMatch keys (edge_map.cpp:78-96) so that no extra pass is needed.
__global__ __launch_bounds__(256) void k_try_vel(KParams p, MapDev om, MapDev nm, int mode_lm, int call, int last,
                                                 const LmState* __restrict__ st_in, LmState* __restrict__ st_out,
                                                 const float* __restrict__ part_prev, float* __restrict__ part_out,
                                                 const int* __restrict__ hist, unsigned frame_count) {
  __shared__ LmState s;
  __shared__ float red[16];
  __shared__ float carry_in;
  __shared__ float wsum[4][10];
  __shared__ float wlast[4];
  __shared__ int whas[4];
  __shared__ float rec[kMaxRecBlocks * kPartStride];
  __shared__ int shist[128];

  // Own-keyline loads are issued before anything else (arrays are padded to the grid, so no bound is needed):
  // their latency overlaps the prologue instead of following it.
  const int idx = blockIdx.x * 256 + threadIdx.x;
  const float gn = om.gnorm[idx];
  const float2 rs = om.rs[idx];
  const float2 pi = om.pos_img[idx];
  const float2 g2 = om.grad[idx];
  const unsigned nmatches = om.matches[idx];
  const float res_in = om.residual[idx];
  if (mode_lm && call >= 1) stage_prev_records(part_prev, gridDim.x, rec);
  const int n = om.st->n;
  const float thr = om.st->threshold;
  const int nblocks = (n + 255) / 256;
  if (mode_lm && call == 0 && threadIdx.x < 128) shist[threadIdx.x] = (threadIdx.x < (unsigned)p.quantile_num_bins) ? hist[threadIdx.x] : 0;
  if (threadIdx.x == 0) {
    carry_in = 0.f;
    s = *st_in;
  }
  __syncthreads();
  if (mode_lm && call >= 1) reduce_staged_records(rec, nblocks, red, &carry_in, (int)blockIdx.x, 1);
  __syncthreads();
  if (threadIdx.x == 0) {
    if (mode_lm) {
      if (call == 0) {
        s.sigma_rho_min = quantile_from_hist(shist, p.quantile_num_bins, p.quantile_cutoff, n);
        for (int i = 0; i < 3; ++i) s.Vnew[i] = s.vel[i];
      } else {
        lm_step(s, red, call, false);
      }
    }
    if (blockIdx.x == 0) *st_out = s;
  }
  __syncthreads();
  if ((int)blockIdx.x >= nblocks && blockIdx.x != 0) return;

  const float vx = s.Vnew[0], vy = s.Vnew[1], vz = s.Vnew[2];
  const float srm = s.sigma_rho_min;
  const float cin = carry_in;
  const unsigned min_matches = min(p.min_match_threshold, frame_count);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;

  TvOut e{};
  e.mfwd = -1;
  if (idx < n) {
    const TvIn in = make_tvin(p, gn, rs, pi, g2, nmatches);
    e = try_vel_eval(p, nm, in, res_in, cin, vx, vy, vz, srm, thr, min_matches);
    if (e.wrote_res) om.residual[idx] = e.res_out;
    if (e.matched && last) {
      const unsigned long long key = ((unsigned long long)order_key(rs.x) << 32) | (unsigned)idx;
      atomicMax(&nm.fwd_key[e.mfwd], key);
    }
    om.match_fwd[idx] = e.mfwd;
  }
  const float f = e.f, jx = e.jx, jy = e.jy, jz = e.jz, fi = e.fi;
  const bool contrib = e.contrib, matched = e.matched, need_carry = e.need_carry;

  // carry-forward of the last written fi (index order) for calculatefJ's early-return paths
  const unsigned long long mm = __ballot(matched);
  const unsigned long long below = mm & ((1ull << lane) - 1ull);
  const int src = below ? (63 - __clzll((long long)below)) : 0;
  const float fi_prev = __shfl(fi, src);
  const int wl = mm ? (63 - __clzll((long long)mm)) : 0;
  const float fi_wlast = __shfl(fi, wl);
  if (lane == 0) {
    whas[wid] = mm ? 1 : 0;
    wlast[wid] = fi_wlast;
  }

  const float sc = contrib ? f * f : 0.f;
  float v[10] = {sc, jx * jx, jy * jy, jz * jz, jx * jy, jx * jz, jy * jz, jx * f, jy * f, jz * f};
  wave_total63_fN(v);
  if (lane == 63) {
#pragma unroll
    for (int k = 0; k < 10; ++k) wsum[wid][k] = v[k];
  }
  __syncthreads();
  if (need_carry) {
    float r;
    if (below) {
      r = fabsf(fi_prev);
    } else {
      r = kResidualCarry;
      for (int w = wid - 1; w >= 0; --w)
        if (whas[w]) {
          r = fabsf(wlast[w]);
          break;
        }
    }
    om.residual[idx] = r;
  }
  if (threadIdx.x < 10) {
    float acc = 0.f;
    for (int w = 0; w < 4; ++w) acc += wsum[w][threadIdx.x];
    part_out[blockIdx.x * kPartStride + threadIdx.x] = acc;
  } else if (threadIdx.x == 10) {
    float hv = 0.f, lv = 0.f;
    for (int w = 3; w >= 0; --w)
      if (whas[w]) {
        hv = 1.f;
        lv = wlast[w];
        break;
      }
    part_out[blockIdx.x * kPartStride + 10] = hv;
    part_out[blockIdx.x * kPartStride + 11] = lv;
  }
}

// Final accept/reject of minimizeVel for the stand-alone entry (the fused path does this in k_ext_rot_vel).
__global__ __launch_bounds__(256) void k_lm_final(MapDev om, int calls, const LmState* __restrict__ st_in,
                                                 LmState* __restrict__ st_out, const float* __restrict__ part_prev) {
  __shared__ float red[16];
  __shared__ float carry_in;
  __shared__ float rec[kMaxRecBlocks * kPartStride];
  stage_prev_records(part_prev, kMaxRecBlocks, rec);
  const int nb = (om.st->n + 255) / 256;
  __syncthreads();
  reduce_staged_records(rec, nb, red, &carry_in, 0, 1);
  __syncthreads();
  if (threadIdx.x == 0) {
    LmState s = *st_in;
    lm_step(s, red, calls, true);
    *st_out = s;
  }
}

// Stand-alone forwardMatch keys from match_id_forward (for the stepwise API; the fused path publishes the
// keys from the last tryVel evaluation).
__global__ __launch_bounds__(256) void k_forward_keys(MapDev om, MapDev nm) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= om.st->n) return;
  const int j = om.match_fwd[idx];
  if (j < 0) return;
  const unsigned long long key = ((unsigned long long)order_key(om.rs[idx].x) << 32) | (unsigned)idx;
  atomicMax(&nm.fwd_key[j], key);
}

// Per-keyline body of forwardMatch's gather (edge_map.cpp:78-96) and of extRotVel's row (core.cpp:198-245), shared by
// the per-call kernel and the persistent LM kernel.
struct XrvIn {
  unsigned long long key;
  int mid;
  float2 rs, mpi, g;
  float gn;
  float2 q;
};
// The forwardMatch gather of one keyline of the new map as loads only (what the old keyline named by `key` hands over):
// a caller that knows the keys early (k_lm_chain_spec) has these in flight while it still decides where the velocity ends up.
struct XrvFwd {
  float2 rs, mpi, mgrad;
  float mgnorm;
  unsigned matches;
  int kf, o;  // o < 0: nobody matched into this keyline
};
__device__ __forceinline__ XrvFwd xrv_gather(const MapDev& om, unsigned long long key) {
  XrvFwd f{};
  f.o = -1;
  if (key != 0ull) {
    const int o = (int)(unsigned)(key & 0xFFFFFFFFull);
    f.o = o;
    f.rs = om.rs[o];
    f.mpi = om.pos_img[o];
    f.matches = om.matches[o];
    f.mgrad = om.grad[o];
    f.mgnorm = om.gnorm[o];
    f.kf = om.match_kf[o];
  }
  return f;
}
__device__ __forceinline__ void xrv_apply(const KParams& p, MapDev& nm, int idx, int do_forward, const XrvFwd& f, XrvIn k, float vx,
                                          float vy, float vz, float row[6], float* Y_out, int* cnt_out) {
  float2 rs = k.rs, mpi = k.mpi;
  int mid = k.mid;
  if (do_forward) {
    if (f.o >= 0) {
      const int o = f.o;
      rs = f.rs;
      mpi = f.mpi;
      mid = o;
      nm.rs[idx] = rs;
      nm.matches[idx] = f.matches + 1u;
      nm.match_id[idx] = o;
      nm.mpos_img[idx] = mpi;
      nm.mgrad[idx] = f.mgrad;
      nm.mgnorm[idx] = f.mgnorm;
      nm.match_kf[idx] = f.kf;
    }
  }
  if (mid >= 0) {
    const float u_x = k.g.x / k.gn;
    const float u_y = k.g.y / k.gn;
    const float rho_t = (float)(1.0 / (1.0 / (double)rs.x + (double)vz));
    const float qt_x = mpi.x + rho_t * (vx * p.fm - vz * mpi.x);
    const float qt_y = mpi.y + rho_t * (vy * p.fm - vz * mpi.y);
    const float q_x = k.q.x, q_y = k.q.y;
    row[0] = u_x * rho_t * p.fm;
    row[1] = u_y * rho_t * p.fm;
    row[2] = u_x * (-rho_t * q_x) + u_y * (-rho_t * q_y);
    row[3] = -u_x * q_x * q_y / p.fm - u_y * (p.fm + q_y * q_y / p.fm);
    row[4] = u_y * q_x * q_y / p.fm + u_x * (p.fm + q_x * q_x / p.fm);
    row[5] = -u_x * q_y + u_y * q_x;
    float Y = u_x * (q_x - qt_x) + u_y * (q_y - qt_y);
    const float dqvel = u_x * (vx * p.fm - vz * mpi.x) + u_y * (vy * p.fm - vz * mpi.y);
    const float s_y = sqrtf(rs.y * rs.y * dqvel * dqvel + p.pixel_uncertainty * p.pixel_uncertainty);
    float weight = 1.0f;
    if (fabsf(Y) > p.reweight_distance) weight = fabsf(Y) / p.reweight_distance;
    const float dv = s_y * weight;
#pragma unroll
    for (int i = 0; i < 6; ++i) row[i] /= dv;
    Y /= dv;
    *Y_out = Y;
    *cnt_out = 1;
  }
}
__device__ __forceinline__ void xrv_eval(const KParams& p, const MapDev& om, MapDev& nm, int idx, int do_forward, XrvIn k, float vx,
                                         float vy, float vz, float row[6], float* Y_out, int* cnt_out) {
  const XrvFwd f = xrv_gather(om, do_forward ? k.key : 0ull);
  xrv_apply(p, nm, idx, do_forward, f, k, vx, vy, vz, row, Y_out, cnt_out);
}

// Sequence stamp of one extRotVel block record (kXrvStride floats: 27 sums, the match count, two spare words, a checksum, the stamp),
// called by the 32 lanes that own the record's words right after they stored them (`bits` = what lane k stored, 0 for k >= 28): the
// wave waits for its stores (vmcnt(0)), lane 30 stores the XOR of the record's words ^ seq, lane 31 the stamp. A host that polls
// the record accepts it when stamp AND checksum fit: the record is 128 bytes = two 64-byte writes on the way to host memory, and
// writes to host memory may arrive out of order.
__device__ __forceinline__ void xrv_record_stamp(float* rec, int k, unsigned seq, unsigned bits) {
  unsigned x = bits;
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) x ^= (unsigned)__shfl_xor((int)x, o);  // (stays inside the 32-lane half that owns the record)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0) (see stamp_drain)
  if (k == kXrvStride - 2) __hip_atomic_store(reinterpret_cast<unsigned*>(rec + k), x ^ seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  if (k == kXrvStride - 1) __hip_atomic_store(reinterpret_cast<unsigned*>(rec + k), seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ---- EdgeMap::forwardMatch gather (edge_map.cpp:78-96) + Core::extRotVel sums (core.cpp:198-245) -------------
// Sequential rule "overwrite unless the target already holds a larger rho" == the writer with the largest
// rho wins, ties -> largest index: exactly the atomicMax key. One thread per keyline of the NEW map.
// Per-thread outer product of the 6-vector row (Phi is never materialised): 21 + 6 sums + match count.
__global__ __launch_bounds__(256) void k_ext_rot_vel(KParams p, MapDev om, MapDev nm, int do_forward, int do_lm_final,
                                                     int calls, const LmState* __restrict__ st_in,
                                                     LmState* __restrict__ st_out, const float* __restrict__ part_prev,
                                                     float* __restrict__ xrv_part, Vec3 vel_manual,
                                                     PairSlot* __restrict__ slot, int* __restrict__ hist_to_zero, unsigned seq) {
  unsigned slot_sum = 0u;  // XOR of the words this thread stored into the result slot (PairSlot::sum)
  __shared__ LmState s;
  __shared__ float red[16];
  __shared__ float carry_in;
  __shared__ float wsum[4][28];
  __shared__ float rec[kMaxRecBlocks * kPartStride];
  // early, bound-free loads of this thread's own keyline (arrays are padded to the grid)
  const int idx = blockIdx.x * 256 + threadIdx.x;
  const unsigned long long key = nm.fwd_key[idx];
  int mid = nm.match_id[idx];
  float2 rs = nm.rs[idx];
  float2 mpi = nm.mpos_img[idx];
  const float2 g = nm.grad[idx];
  const float gn = nm.gnorm[idx];
  const float2 q = nm.pos_img[idx];
  const int n = nm.st->n;
  if (do_lm_final) {
    stage_prev_records(part_prev, gridDim.x, rec);
    const int nb_old = (om.st->n + 255) / 256;
    if (threadIdx.x == 0) s = *st_in;
    __syncthreads();
    reduce_staged_records(rec, nb_old, red, &carry_in, (int)blockIdx.x, 1);
    __syncthreads();
    if (threadIdx.x == 0) {
      lm_step(s, red, calls, true);
      if (blockIdx.x == 0) {
        *st_out = s;
        if (slot) slot_sum = slot_fill(slot, s, nm.st, om.st);  // zero-copy: the host reads these once the stamps say so
      }
    }
    __syncthreads();
  }
  // every tryVel call of this pair has consumed the sigma_rho histogram: clear it for the next pair
  if (hist_to_zero && blockIdx.x == 0 && threadIdx.x < 128) hist_to_zero[threadIdx.x] = 0;
  const float vx = do_lm_final ? s.vel[0] : vel_manual.a[0];
  const float vy = do_lm_final ? s.vel[1] : vel_manual.a[1];
  const float vz = do_lm_final ? s.vel[2] : vel_manual.a[2];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;

  float row[6] = {0, 0, 0, 0, 0, 0};
  float Y = 0.f;
  int cnt = 0;
  if (idx < n) {
    XrvIn k{key, mid, rs, mpi, g, gn, q};
    xrv_eval(p, om, nm, idx, do_forward, k, vx, vy, vz, row, &Y, &cnt);
  }
  float v[28];
  {
    int k = 0;
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
      for (int j = i; j < 6; ++j) v[k++] = row[i] * row[j];
#pragma unroll
    for (int i = 0; i < 6; ++i) v[21 + i] = row[i] * Y;
    v[27] = (float)cnt;
  }
  wave_total63_fN(v);
  if (lane == 63) {
#pragma unroll
    for (int k = 0; k < 28; ++k) wsum[wid][k] = v[k];
  }
  __syncthreads();
  unsigned stored = 0u;
  if (threadIdx.x < 28) {
    float acc = 0.f;
    for (int w = 0; w < 4; ++w) acc += wsum[w][threadIdx.x];
    xrv_part[blockIdx.x * kXrvStride + threadIdx.x] = acc;
    stored = __float_as_uint(acc);
  }
  if (threadIdx.x < 32) xrv_record_stamp(xrv_part + (size_t)blockIdx.x * kXrvStride, (int)threadIdx.x, seq, stored);
  if (slot && blockIdx.x == 0 && threadIdx.x == 0) {  // (block 0's own stores: lm, map states, its record)
    slot->sum = slot_sum ^ seq;
    stamp_drain(&slot->seq, seq);
  }
}

// ---- persistent minimizeVel + forwardMatch + extRotVel (core.cpp:150-245, edge_map.cpp:78-96) --------------------
// One launch instead of iterations+2: every workgroup keeps its keylines in registers across the LM evaluations and the
// workgroups exchange their block records after each evaluation. Records stay per 256 keylines ("record groups") and
// are summed in the same fixed order as in the per-call kernels, so both paths produce identical bits.
// The exchange is the grid barrier: every record word is published as one 64-bit agent-scope store {tag, float bits},
// tag = tag_base + evaluation index + 1 (unique over the life of the context, never 0), into the parity slot of the
// evaluation; every workgroup polls the words of all live record groups until they carry the tag - one memory round
// trip per evaluation, no contended counter, cost independent of the number of workgroups. A consumer that has seen
// evaluation c of every group knows every workgroup has finished reading evaluation c-1, so two parity slots suffice.
// A poll that lasts longer than kXchSpinLimit iterations (~0.2 s) raises *err and goes on: the grid always drains.
constexpr unsigned kXchSpinLimit = 1u << 18;

// The hand-off of the forwardMatch side effects (fwd_key atomics) to the other workgroups of the launch:
//   producer: every wave that issued them runs vm_drain() - a workgroup-scope release fence (compiler ordering: nothing it
//             issued before may sink below; on gfx942 / gfx950 outside tgsplit mode it emits no cache operation) followed by
//             "s_waitcnt vmcnt(0)": the atomics are performed at the device's coherence point (agent scope) and ACKNOWLEDGED
//             before the workgroup barrier behind which the record words that announce them go out (xch_publish);
//   consumer: polls the words (xch_wait), then xch_acquire() - a workgroup-scope acquire fence (compiler ordering: nothing
//             behind it may be hoisted above the polls) - then reads the keys with agent-scope ATOMIC loads, which are served
//             by the same coherence point and are issued in program order behind the poll that saw the tag.
// Everything that crosses workgroups inside these kernels is an agent-scope atomic word; no plain store is ever read by
// another workgroup of the same launch (match_fwd etc. are consumed by later kernels). What is deliberately NOT used is an
// agent-scope fence: the release form writes the XCD's L2 back (15 us per pair), the acquire form
// (buffer_inv sc1) invalidates the XCD's non-coherently cached L2 lines for every kernel running beside this one - measured
// -4 % frames/s on one stream and -5 % with 8 lanes (13.04 k -> 12.5 k, 39.1 k -> 37.2 k, same box, alternating runs), against
// no measurable cost for the workgroup-scope pair. The s_waitcnt immediate is the gfx9 encoding, where vmcnt counts loads,
// stores AND atomics of the wave; targets with a separate store counter (gfx10+) need another one, hence the guard.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx942__) && !defined(__gfx950__)
#error "track.hip: vm_drain() encodes s_waitcnt for gfx942 / gfx950 only"
#endif
__device__ __forceinline__ void vm_drain() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0); expcnt / lgkmcnt untouched
}
__device__ __forceinline__ void fwd_key_max(unsigned long long* w, unsigned long long key) {
  (void)__hip_atomic_fetch_max(w, key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void xch_acquire() { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); }

__device__ __forceinline__ void xch_publish(unsigned long long* w, unsigned tag, float v) {
  __hip_atomic_store(w, ((unsigned long long)tag << 32) | (unsigned long long)__float_as_uint(v), __ATOMIC_RELAXED,
                     __HIP_MEMORY_SCOPE_AGENT);
}
// g_poll_sleep_long (set per launch through a kernel argument copied to a register): batched launches run several lanes'
// workgroups at once and every polling thread is an L2 request per poll - they back off longer between polls.
__device__ __forceinline__ float xch_wait(const unsigned long long* w, unsigned tag, int* err, int slow = 0) {
  unsigned spins = 0;
  unsigned long long v = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  while ((unsigned)(v >> 32) != tag) {
    if (slow)
      __builtin_amdgcn_s_sleep(12);
    else
      __builtin_amdgcn_s_sleep(1);
    if (++spins > kXchSpinLimit) {
      // err[0] = flag, err[1..7] = diagnostics of the first waiter that gave up (host-visible pinned memory)
      if (__hip_atomic_exchange(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == 0) {
        err[1] = (int)blockIdx.x;
        err[2] = (int)threadIdx.x;
        err[3] = (int)tag;
        err[4] = (int)(unsigned)(v >> 32);
        err[5] = (int)(unsigned)v;
        err[6] = (int)gridDim.x;
        err[7] = (int)(reinterpret_cast<uintptr_t>(w) & 0x7FFFFFFFu);
      }
      break;
    }
    v = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  return __uint_as_float((unsigned)v);
}

// Up to N words per thread polled TOGETHER: every round re-loads the words still missing in one batch (one memory round trip
// per round whatever their number), until all carry their tags. Waiting for them one after the other costs a round trip per
// word once the first has arrived; loading all once and then waiting one by one (tried) adds a round trip in the usual case
// that nothing is out yet. ok[j] false = no word for this thread in slot j. Same bound and diagnostics as xch_wait.
template <int N>
__device__ __forceinline__ void xch_wait_many(const unsigned long long* const (&w)[N], const unsigned (&tag)[N], const bool (&ok)[N],
                                              float (&out)[N], int* err, int slow) {
  bool have[N];
  bool all = true;
#pragma unroll
  for (int j = 0; j < N; ++j) {
    have[j] = !ok[j];
    out[j] = 0.f;
    all = all && have[j];
  }
  unsigned spins = 0;
  while (!all) {
    unsigned long long v[N];
#pragma unroll
    for (int j = 0; j < N; ++j)
      if (!have[j]) v[j] = __hip_atomic_load(w[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    all = true;
#pragma unroll
    for (int j = 0; j < N; ++j)
      if (!have[j]) {
        if ((unsigned)(v[j] >> 32) == tag[j]) {
          have[j] = true;
          out[j] = __uint_as_float((unsigned)v[j]);
        } else {
          all = false;
        }
      }
    if (all) break;
    if (slow)
      __builtin_amdgcn_s_sleep(12);
    else
      __builtin_amdgcn_s_sleep(1);
    if (++spins > kXchSpinLimit) {
      if (__hip_atomic_exchange(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == 0) {
        err[1] = (int)blockIdx.x;
        err[2] = (int)threadIdx.x;
        err[3] = (int)tag[0];
        err[4] = 0;
        err[5] = 0;
        err[6] = (int)gridDim.x;
        err[7] = -3;
      }
      break;
    }
  }
}

// Wait for the records every live group published for evaluation `tag`, stage them in LDS and reduce them.
template <int kChainThreads>
__device__ __forceinline__ void chain_collect_records(const unsigned long long* __restrict__ slot_words, unsigned tag, int nblocks,
                                                      float* rec, float* red, float* carry_in, int* err, int slow = 0) {
  constexpr int kChainGroups = kChainThreads / 256;
  const int total = min(nblocks, kMaxRecBlocks) * kPartStride;
  for (int i = threadIdx.x; i < total; i += kChainThreads)
    if ((i & (kPartStride - 1)) < 12) rec[i] = xch_wait(slot_words + i, tag, err, slow);
  __syncthreads();
  reduce_staged_records(rec, nblocks, red, carry_in, (int)blockIdx.x * kChainGroups, kChainGroups);
  __syncthreads();
}

__device__ __forceinline__ GlueArgs lane_glue_args(const LaneStatic& L, const LaneDyn& d, int calls, const GlueParams& gp) {
  GlueArgs ga;
  ga.lm = gptr(L.lm) + calls + 1;
  ga.xrv = gptr(L.xrv_part);
  ga.st_in = gptr(L.gstate) + (d.gpar & 1);
  ga.st_out = gptr(L.gstate) + ((d.gpar & 1) ^ 1);
  ga.rec = gptr(L.rec[d.slot]);
  ga.gd_copy = gptr(L.glue_dev) + d.slot;
  ga.stage = gptr(L.glue_stage) + d.slot;
  ga.seq = d.seq;
  // (only the scalars: a batch forms the gyroBiasCorrection matrices on the device - has_pre = 0 as a constant lets the compiler
  // drop the by-value matrices; copying them put the whole struct into scratch memory, 280 bytes per lane of every workgroup)
  ga.gp.frame_dt = gp.frame_dt;
  ga.gp.gyro_std_dev = gp.gyro_std_dev;
  ga.gp.gyro_bias_std_dev = gp.gyro_bias_std_dev;
  ga.gp.has_pre = 0;
  return ga;
}

// Tail of the persistent LM kernels when the pair's glue runs on the device (ga.lm != null): every workgroup publishes the
// extRotVel sums of its record groups as tagged words (the exchange of the LM records, one more round), workgroup 0 waits
// for the groups that hold live keylines of the new map, sums them in group order in double (hm::sum_xrv) and runs the
// glue (glue_dev.hpp). The other workgroups leave right after publishing. stage: LDS, stage_groups * 32 floats.
template <int kChainThreads>
__device__ __forceinline__ void lm_tail_glue(const GlueArgs& ga, unsigned long long* __restrict__ xch_xrv, unsigned tag, int n_new,
                                             float* stage, int stage_groups, GlueLds& gw, const LmState& s, int* err, int slow,
                                             unsigned long long* stamps) {
  if (blockIdx.x != 0) return;
  const int tid = threadIdx.x;
  const int nb = (n_new + 255) / 256;
  double acc = 0.0;
  for (int b0 = 0; b0 < nb; b0 += stage_groups) {
    const int cnt = min(stage_groups, nb - b0);
    // a thread's words are polled TOGETHER (one memory round trip per round whatever their number, see xch_wait_many)
    for (int i0 = 0; i0 < cnt * kXrvStride; i0 += kChainThreads * 4) {
      const unsigned long long* w4[4];
      unsigned t4[4];
      bool ok4[4];
      float o4[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int i = i0 + j * kChainThreads + tid;
        ok4[j] = i < cnt * kXrvStride && (i & (kXrvStride - 1)) < 27;
        w4[j] = xch_xrv + (size_t)b0 * kXrvStride + min(i, cnt * kXrvStride - 1);
        t4[j] = tag;
      }
      xch_wait_many<4>(w4, t4, ok4, o4, err, slow);
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (ok4[j]) stage[i0 + j * kChainThreads + tid] = o4[j];
    }
    __syncthreads();
    if (tid < 27)
      for (int b = 0; b < cnt; ++b) acc += (double)stage[b * kXrvStride + tid];
    __syncthreads();
  }
  if (tid < 21) {  // upper triangle in row-major order -> both halves of the symmetric matrix
    int i = 0, base = 0;
    while (tid >= base + (6 - i)) {
      base += 6 - i;
      ++i;
    }
    const int j = i + (tid - base);
    gw.W[i * 6 + j] = (float)acc;
    gw.W[j * 6 + i] = (float)acc;
  } else if (tid < 27) {
    gw.JtF[tid - 21] = (float)acc;
  }
  __syncthreads();
  if (stamps && tid == 0) stamps[16] = __builtin_amdgcn_s_memrealtime();
  glue_workgroup(gw, s, ga, stamps);
}

template <int kChainThreads>
__device__ __forceinline__ void lm_chain_body(KParams p, MapDev om, MapDev nm, int calls, int do_ext,
                                                            const LmState* __restrict__ st_in, LmState* __restrict__ st_out,
                                                            unsigned long long* __restrict__ xch, unsigned tag_base,
                                                            int* __restrict__ bar_err, const int* __restrict__ hist,
                                                            unsigned frame_count, float* __restrict__ xrv_part,
                                                            PairSlot* __restrict__ slot, int* __restrict__ hist_to_zero,
                                                            unsigned long long* __restrict__ stamps, int slow_poll, const GlueArgs ga) {
  unsigned slot_sum = 0u;  // XOR of the words thread 0 of workgroup 0 stored into the result slot (PairSlot::sum)
  constexpr int kChainGroups = kChainThreads / 256;
  __shared__ GlueLds gw;
  __shared__ float lm_inv[9];
  // optional phase stamps of workgroup 0 (REBVIO_HIP_LM_STAMPS diagnostic): 100 MHz constant clock
#define RH_STAMP(i) \
  do { if (stamps && blockIdx.x == 0 && threadIdx.x == 0) stamps[(i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
  __shared__ LmState s;
  __shared__ float red[16];
  __shared__ float carry_in[kChainGroups];
  __shared__ float wsum[kChainThreads / 64][28];
  __shared__ float wlast[kChainThreads / 64];
  __shared__ int whas[kChainThreads / 64];
  __shared__ float rec[kMaxRecBlocks * kPartStride];
  __shared__ int shist[128];

  const int tid = threadIdx.x;
  const int idx = blockIdx.x * kChainThreads + tid;
  const int lane = tid & 63, wid = tid >> 6, grp = tid >> 8, wig = wid & 3;
  const int nrec_launched = gridDim.x * kChainGroups;
  // own keyline of the old map (tryVel) and of the new map (extRotVel): loaded once, bound-free (arrays are padded)
  const TvIn in = make_tvin(p, om.gnorm[idx], om.rs[idx], om.pos_img[idx], om.grad[idx], om.matches[idx]);
  float res = om.residual[idx];
  XrvIn xk{};
  if (do_ext) {
    xk.mid = nm.match_id[idx];
    xk.rs = nm.rs[idx];
    xk.mpi = nm.mpos_img[idx];
    xk.g = nm.grad[idx];
    xk.gn = nm.gnorm[idx];
    xk.q = nm.pos_img[idx];
  }
  const int n = om.st->n;
  const float thr = om.st->threshold;
  const int n_new = nm.st->n;
  const int nblocks = (n + 255) / 256;
  if (tid < 128) shist[tid] = (tid < p.quantile_num_bins) ? hist[tid] : 0;
  if (tid == 0) s = *st_in;
  if (tid < kChainGroups) carry_in[tid] = 0.f;
  __syncthreads();
  if (tid < 64) {
    const float q = quantile_from_hist_wave(shist, p.quantile_num_bins, p.quantile_cutoff, n, tid);
    if (tid == 0) {
      s.sigma_rho_min = q;
      for (int i = 0; i < 3; ++i) s.Vnew[i] = s.vel[i];
    }
  }
  __syncthreads();
  const unsigned min_matches = min(p.min_match_threshold, frame_count);
  RH_STAMP(0);
  // A workgroup none of whose record groups holds a live keyline of the OLD map takes no part in the LM exchange: the
  // others do not wait for its records, so it could fall arbitrarily far behind them and find the parity slots already
  // rewritten. It only needs the final velocity for the extRotVel rows of its NEW-map keylines, which workgroup 0
  // broadcasts under the launch's last tag. (Workgroup 0 always runs the loop, also for an empty old map.)
  const bool lm_live = blockIdx.x == 0 || (int)blockIdx.x * kChainGroups < nblocks;
  unsigned long long* xch_final = xch + (size_t)2 * nrec_launched * kPartStride;
  const unsigned tag_final = tag_base + (unsigned)calls + 1u;

  for (int call = 0; lm_live && call < calls; ++call) {
    RH_STAMP(1 + call * 6 + 0);
    if (call >= 1) {
      chain_collect_records<kChainThreads>(xch + (size_t)((call - 1) & 1) * nrec_launched * kPartStride, tag_base + (unsigned)call, nblocks,
                                           rec, red, carry_in, bar_err, slow_poll);
      RH_STAMP(1 + call * 6 + 1);
      if (tid < 64) lm_step_wave(s, red, call, false, lm_inv, tid);
      __syncthreads();
    } else {
      RH_STAMP(1 + call * 6 + 1);
    }
    RH_STAMP(1 + call * 6 + 2);
    const float vx = s.Vnew[0], vy = s.Vnew[1], vz = s.Vnew[2];
    const float srm = s.sigma_rho_min;
    const float cin = carry_in[grp];
    const bool last = call == calls - 1;
    TvOut e{};
    e.mfwd = -1;
    if (idx < n) {
      e = try_vel_eval(p, nm, in, res, cin, vx, vy, vz, srm, thr, min_matches);
      if (e.wrote_res) res = e.res_out;
      if (last) {
        if (e.matched) {
          const unsigned long long key = ((unsigned long long)order_key(in.rs.x) << 32) | (unsigned)idx;
          fwd_key_max(&nm.fwd_key[e.mfwd], key);
        }
        om.match_fwd[idx] = e.mfwd;
      }
    }
    // last evaluation: this wave's forwardMatch atomics are acknowledged before the workgroup publishes its records (see
    // k_lm_chain_spec; a release fence on the publishing threads would order only their own operations)
    if (last) vm_drain();
    RH_STAMP(1 + call * 6 + 3);
    // carry-forward of the last written fi in index order (see k_try_vel)
    const unsigned long long mm = __ballot(e.matched);
    const unsigned long long below = mm & ((1ull << lane) - 1ull);
    const int src = below ? (63 - __clzll((long long)below)) : 0;
    const float fi_prev = __shfl(e.fi, src);
    const int wl = mm ? (63 - __clzll((long long)mm)) : 0;
    const float fi_wlast = __shfl(e.fi, wl);
    if (lane == 0) {
      whas[wid] = mm ? 1 : 0;
      wlast[wid] = fi_wlast;
    }
    const float sc = e.contrib ? e.f * e.f : 0.f;
    float v[10] = {sc, e.jx * e.jx, e.jy * e.jy, e.jz * e.jz, e.jx * e.jy, e.jx * e.jz, e.jy * e.jz, e.jx * e.f, e.jy * e.f, e.jz * e.f};
    wave_total63_fN(v);
    if (lane == 63) {
#pragma unroll
      for (int k = 0; k < 10; ++k) wsum[wid][k] = v[k];
    }
    __syncthreads();
    RH_STAMP(1 + call * 6 + 4);
    if (e.need_carry) {
      float r;
      if (below) {
        r = fabsf(fi_prev);
      } else {
        r = kResidualCarry;
        for (int w = wig - 1; w >= 0; --w)
          if (whas[grp * 4 + w]) {
            r = fabsf(wlast[grp * 4 + w]);
            break;
          }
      }
      res = r;
    }
    if (tid < kChainGroups * 16) {
      // publish this evaluation's records (self-contained words, relaxed); only the last evaluation has other data to
      // order before them: the forwardMatch keys this workgroup issued - and waited for - before the __syncthreads above
      const int g = tid >> 4, k = tid & 15;
      unsigned long long* out = xch + ((size_t)(call & 1) * nrec_launched + (size_t)blockIdx.x * kChainGroups + g) * kPartStride;
      const unsigned tag = tag_base + (unsigned)call + 1u;
      if (k < 10) {
        float acc = 0.f;
        for (int w = 0; w < 4; ++w) acc += wsum[g * 4 + w][k];
        xch_publish(out + k, tag, acc);
      } else if (k == 10 || k == 11) {
        float hv = 0.f, lv = 0.f;
        for (int w = 3; w >= 0; --w)
          if (whas[g * 4 + w]) {
            hv = 1.f;
            lv = wlast[g * 4 + w];
            break;
          }
        xch_publish(out + k, tag, k == 10 ? hv : lv);
      }
    }
    __syncthreads();  // wsum / whas / wlast are rewritten by the next evaluation
    RH_STAMP(1 + call * 6 + 5);
  }

  // final accept / reject of minimizeVel (core.cpp:166-185 for the last evaluation)
  if (lm_live && calls > 0) {
    chain_collect_records<kChainThreads>(xch + (size_t)((calls - 1) & 1) * nrec_launched * kPartStride, tag_base + (unsigned)calls, nblocks,
                                         rec, red, carry_in, bar_err, slow_poll);
    if (tid == 0) lm_step(s, red, calls, true);
    __syncthreads();
  }
  if (blockIdx.x == 0 && tid == 0) {
    *st_out = s;
    if (slot) slot_sum = slot_fill(slot, s, nm.st, om.st);  // zero-copy: the host reads these once the stamps say so
  }
  RH_STAMP(1 + calls * 6);
  if (!do_ext) {
    if (slot && blockIdx.x == 0 && tid == 0) {
      slot->sum = slot_sum ^ ga.seq;
      stamp_drain(&slot->seq, ga.seq);
    }
    return;
  }
  if (hist_to_zero && blockIdx.x == 0 && tid < 128) hist_to_zero[tid] = 0;  // every evaluation has consumed the histogram
  if (blockIdx.x == 0 && tid < 3) xch_publish(xch_final + tid, tag_final, s.vel[tid]);
  if (!lm_live) {
    if (tid < 3) s.vel[tid] = xch_wait(xch_final + tid, tag_final, bar_err);
    __syncthreads();
  }
  const float vx = s.vel[0], vy = s.vel[1], vz = s.vel[2];
  float row[6] = {0, 0, 0, 0, 0, 0};
  float Y = 0.f;
  int cnt = 0;
  xch_acquire();  // behind the polls of the last evaluation's records: the forwardMatch keys of every workgroup (vm_drain)
  if (idx < n_new) {
    xk.key = __hip_atomic_load(&nm.fwd_key[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    xrv_eval(p, om, nm, idx, 1, xk, vx, vy, vz, row, &Y, &cnt);
  }
  float v[28];
  {
    int k = 0;
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
      for (int j = i; j < 6; ++j) v[k++] = row[i] * row[j];
#pragma unroll
    for (int i = 0; i < 6; ++i) v[21 + i] = row[i] * Y;
    v[27] = (float)cnt;
  }
  wave_total63_fN(v);
  if (lane == 63) {
#pragma unroll
    for (int k = 0; k < 28; ++k) wsum[wid][k] = v[k];
  }
  __syncthreads();
  // (both kernels keep the same exchange layout: the extRotVel words follow the speculative kernel's words)
  unsigned long long* xch_xrv = xch + lm_xch_xrv_offset((size_t)nrec_launched, (size_t)gridDim.x);
  if (tid < kChainGroups * 32) {
    const int g = tid >> 5, k = tid & 31;
    unsigned stored = 0u;
    if (k < 28) {
      float acc = 0.f;
      for (int w = 0; w < 4; ++w) acc += wsum[g * 4 + w][k];
      if (ga.lm)
        xch_publish(xch_xrv + ((size_t)blockIdx.x * kChainGroups + g) * kXrvStride + k, tag_final, acc);
      else
        xrv_part[((size_t)blockIdx.x * kChainGroups + g) * kXrvStride + k] = acc;
      stored = __float_as_uint(acc);
    }
    // records that go straight to the host (per-pair API): each carries the pair's sequence stamp in its last word, stored once
    // this wave's 28 sums are acknowledged - a host that polls the records instead of waiting for an event reads complete ones
    if (!ga.lm) xrv_record_stamp(xrv_part + ((size_t)blockIdx.x * kChainGroups + g) * kXrvStride, k, ga.seq, stored);
  }
  RH_STAMP(2 + calls * 6);
  if (ga.lm) {
    __syncthreads();  // (rec is free: every LM collect is over)
    lm_tail_glue<kChainThreads>(ga, xch_xrv, tag_final, n_new, rec, kMaxRecBlocks * kPartStride / kXrvStride, gw, s, bar_err, slow_poll, stamps);
  }
  if (slot && blockIdx.x == 0 && tid == 0) {  // the slot's stamp: by the directedMatch launch with the glue's record, or here
    slot->sum = slot_sum ^ ga.seq;
    if (ga.lm && ga.stage)
      ga.stage->host_slot = slot;
    else
      stamp_drain(&slot->seq, ga.seq);
  }
#undef RH_STAMP
}

// ---- persistent minimizeVel with SPECULATIVE evaluation of the reject chain ------------------------------------------------
// Observation (tools/mask_hist.py, profiles/r02_divergence_report.txt): on every tested stream minimizeVel accepts its first
// step and rejects the other four (accept mask 00001). After a rejection the next trial point depends on nothing the rejected
// evaluation produced: u *= v, v *= 2, h = -(JtJ + uI)^-1 JtF with the OLD JtJ / JtF (core.cpp:180-183). So once the result of
// evaluation 1 is in, the points of evaluations 2, 3, .., calls-1 under the hypothesis "all rejected" are known, and they are
// evaluated back to back in ONE pass - three exchange rounds per launch instead of six. Then the hypothesis is checked with
// the real scores (the gain test of core.cpp:172, same expression); if an evaluation turns out accepted, the launch continues
// from there with ordinary one-evaluation passes (nothing speculative is kept: state and residuals are rolled back to that
// evaluation, the forwardMatch keys the speculative last evaluation published are cleared behind a grid barrier). Either way
// the results are those of the sequential algorithm, bit for bit (test_persistent_lm_kernel_equals_per_call_kernels).
// What consecutive evaluations of one pass need from OTHER workgroups is only the carry-forward of the last written fi
// (oracle header, H3) for keylines in front of their group's first match: each workgroup hands the fi of its last matched
// keyline to its right neighbour through one tagged word per evaluation, published right after the evaluation and polled
// after the workgroup's own reduction (the hop hides behind it).
// Exchange words: [kMaxLmCalls][record groups][kPartStride] record sets (one per evaluation index: no slot is reused within a
// launch), kPartStride words for the final velocity, [kMaxLmCalls][workgroups][2] carry words, [workgroups] barrier words.
// Tags: tag_base + phase * (calls + 1) + evaluation + 1, phase 1 = evaluations repeated after a failed hypothesis; the caller
// advances tag_base by 2 * (calls + 1) per launch.
constexpr int kSpecMax = 6;  // speculative evaluations per launch (iterations <= 7)

template <int kChainThreads>
__device__ __forceinline__ void lm_chain_spec_body(KParams p, MapDev om, MapDev nm, int calls, const LmState* __restrict__ st_in,
                                                   LmState* __restrict__ st_out, unsigned long long* __restrict__ xch, unsigned tag_base,
                                                   int* __restrict__ bar_err, const int* __restrict__ hist, unsigned frame_count,
                                                   float* __restrict__ xrv_part, PairSlot* __restrict__ slot, int* __restrict__ hist_to_zero,
                                                   unsigned long long* __restrict__ stamps, int slow_poll, const GlueArgs ga, int kf) {
  unsigned slot_sum = 0u;  // XOR of the words thread 0 of workgroup 0 stored into the result slot (PairSlot::sum)
  // kf = index of the first speculative evaluation: evaluations 0 .. kf - 1 run one per exchange round, kf .. calls - 1 in one
  // pass under the hypothesis "all rejected". kf = 2: the steady state of a young stream (accept mask 00001); kf = 3: what the
  // same stream settles into once its depths have converged (00011 on every pair from frame ~6000 on) - four
  // exchange rounds instead of the six of k_lm_chain or of a roll-back.
  constexpr int kChainGroups = kChainThreads / 256;
  constexpr int kWaves = kChainThreads / 64;
  __shared__ GlueLds gw;
  __shared__ float lm_inv[4][9];
  // REBVIO_HIP_LM_STAMPS: 1 start, 2/3 evaluations 0/1 published, 4 hypothesis states ready, speculative evaluations: 5 projected,
  // 6 gathers issued, 7 matches known, 8 neighbour round done, 9 weighted sums done, 10 published; 11 all record sets staged and
  // reduced, 12 hypothesis checked, 13 LM done, 14 extRotVel rows out (stamps[0] = 1 marks the layout)
#define RH_STAMP(i) \
  do { if (stamps && blockIdx.x == 0 && threadIdx.x == 0) stamps[(i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
  extern __shared__ float recm[];  // [nspec][cap * kPartStride] staged record sets (set 0 doubles as the ordinary collect's staging)
  __shared__ LmState s;
  __shared__ LmState stc[kSpecMax + 1];  // state before evaluation 2 + k's result is processed (k = 0 .. nspec), under the hypothesis
  __shared__ float red[16];
  __shared__ float redm[kSpecMax][16];
  __shared__ float carry_in[kChainGroups];
  __shared__ float wsum[kWaves][28];
  __shared__ float wlast[kWaves];
  __shared__ int whas[kWaves];
  __shared__ int shist[128];
  __shared__ int first_accept;
  __shared__ float wsumk[kSpecMax][kWaves][10];  // speculative evaluations: per-wave sums, match flags, last fi, carry-in per group
  __shared__ int whask[kSpecMax][kWaves];
  __shared__ float wlastk[kSpecMax][kWaves];
  __shared__ float carryk[kSpecMax + 1][kChainGroups];

  const int tid = threadIdx.x;
  const unsigned long long t_entry = (stamps && blockIdx.x == 0 && tid == 0) ? __builtin_amdgcn_s_memrealtime() : 0ull;  // diagnostic
  const int idx = blockIdx.x * kChainThreads + tid;
  const int lane = tid & 63, wid = tid >> 6, grp = tid >> 8, wig = wid & 3;
  const int nwg = gridDim.x;
  const int nrec_launched = nwg * kChainGroups;
  const TvIn in = make_tvin(p, om.gnorm[idx], om.rs[idx], om.pos_img[idx], om.grad[idx], om.matches[idx]);
  float res = om.residual[idx];
  XrvIn xk{};
  XrvFwd xf{};
  xf.o = -1;
  bool xf_valid = false;  // the forwardMatch gathers of this thread's keyline are in `xf` (issued behind the speculative collect)
  xk.mid = nm.match_id[idx];
  xk.rs = nm.rs[idx];
  xk.mpi = nm.mpos_img[idx];
  xk.g = nm.grad[idx];
  xk.gn = nm.gnorm[idx];
  xk.q = nm.pos_img[idx];
  const int n = om.st->n;
  const float thr = om.st->threshold;
  const int n_new = nm.st->n;
  const int nblocks = (n + 255) / 256;
  const int cap = min(nblocks, kMaxRecBlocks) * kPartStride;  // words staged per record set
  if (tid < 128) shist[tid] = (tid < p.quantile_num_bins) ? hist[tid] : 0;
  if (tid == 0) s = *st_in;
  if (tid < kChainGroups) carry_in[tid] = 0.f;
  __syncthreads();
  const unsigned min_matches = min(p.min_match_threshold, frame_count);
  const bool lm_live = blockIdx.x == 0 || (int)blockIdx.x * kChainGroups < nblocks;
  // Evaluation 0 is taken at the incoming velocity, so its projection and its two dependent gathers (field cell, then the
  // cell's keyline) start before the sigma quantile is known; the quantile's only part in it - the uncertainty gate of
  // tv_project - is applied to the finished projection below (a gated keyline's gathers are never looked at).
  TvProj pj0{};
  pj0.cell = -1;
  pj0.skip = true;
  TvGeo ge0{};
  ge0.id = -1;
  if (lm_live && idx < n) {
    pj0 = tv_project(p, in, s.vel[0], s.vel[1], s.vel[2], __builtin_inff(), thr, min_matches);
    ge0 = tv_geometry(p, nm, tv_cell(nm, pj0));
  }
  if (tid < 64) {
    const float q = quantile_from_hist_wave(shist, p.quantile_num_bins, p.quantile_cutoff, n, tid);
    if (tid == 0) {
      s.sigma_rho_min = q;
      for (int i = 0; i < 3; ++i) s.Vnew[i] = s.vel[i];
    }
  }
  __syncthreads();
  const float srm = s.sigma_rho_min;
  if (in.rs.y > srm) {
    pj0.skip = true;
    pj0.cell = -1;
  }
  const int live_wgs = max(1, (nblocks + kChainGroups - 1) / kChainGroups);
  unsigned long long* xch_final = xch + (size_t)kMaxLmCalls * nrec_launched * kPartStride;
  unsigned long long* xch_carry = xch_final + kPartStride;                  // [kMaxLmCalls][nwg][2]
  unsigned long long* xch_sync = xch_carry + (size_t)kMaxLmCalls * nwg * 2;  // [nwg]
  const unsigned tag_final = tag_base + 2u * ((unsigned)calls + 1u);
  auto tag_of = [&](int call, int phase) { return tag_base + (unsigned)phase * ((unsigned)calls + 1u) + (unsigned)call + 1u; };
  int fa = -1;  // index of the first speculative evaluation that turned out accepted (-1: none / not known yet)
  auto set_words = [&](int call) { return xch + (size_t)call * nrec_launched * kPartStride; };

  // One evaluation at (vx, vy, vz): tryVel body, carry-forward, wave / workgroup sums, records published under `tag`.
  auto do_eval = [&](int call, unsigned tag, const TvProj& pj, const TvGeo& ge, bool last) {
    const float cin = carry_in[grp];
    TvOut e{};
    e.mfwd = -1;
    if (idx < n) {
      e = tv_finish(p, in, pj, ge, res, cin);
      if (e.wrote_res) res = e.res_out;
      if (last) {
        if (e.matched) {
          const unsigned long long key = ((unsigned long long)order_key(in.rs.x) << 32) | (unsigned)idx;
          fwd_key_max(&nm.fwd_key[e.mfwd], key);
        }
        om.match_fwd[idx] = e.mfwd;
      }
    }
    if (last) vm_drain();  // the keys are out before the records (see the speculative phase)
    const unsigned long long mm = __ballot(e.matched);
    const unsigned long long below = mm & ((1ull << lane) - 1ull);
    const int src = below ? (63 - __clzll((long long)below)) : 0;
    const float fi_prev = __shfl(e.fi, src);
    const int wl = mm ? (63 - __clzll((long long)mm)) : 0;
    const float fi_wlast = __shfl(e.fi, wl);
    if (lane == 0) {
      whas[wid] = mm ? 1 : 0;
      wlast[wid] = fi_wlast;
    }
    const float sc = e.contrib ? e.f * e.f : 0.f;
    float v[10] = {sc, e.jx * e.jx, e.jy * e.jy, e.jz * e.jz, e.jx * e.jy, e.jx * e.jz, e.jy * e.jz, e.jx * e.f, e.jy * e.f, e.jz * e.f};
    wave_total63_fN(v);
    if (lane == 63) {
#pragma unroll
      for (int k = 0; k < 10; ++k) wsum[wid][k] = v[k];
    }
    __syncthreads();
    if (e.need_carry) {
      float r;
      if (below) {
        r = fabsf(fi_prev);
      } else {
        r = kResidualCarry;
        for (int w = wig - 1; w >= 0; --w)
          if (whas[grp * 4 + w]) {
            r = fabsf(wlast[grp * 4 + w]);
            break;
          }
      }
      res = r;
    }
    if (tid < kChainGroups * 16) {
      const int g = tid >> 4, k = tid & 15;
      unsigned long long* out = set_words(call) + ((size_t)blockIdx.x * kChainGroups + g) * kPartStride;
      if (k < 10) {
        float acc = 0.f;
        for (int w = 0; w < 4; ++w) acc += wsum[g * 4 + w][k];
        xch_publish(out + k, tag, acc);
      } else if (k == 10 || k == 11) {
        float hv = 0.f, lv = 0.f;
        for (int w = 3; w >= 0; --w)
          if (whas[g * 4 + w]) {
            hv = 1.f;
            lv = wlast[g * 4 + w];
            break;
          }
        xch_publish(out + k, tag, k == 10 ? hv : lv);
      }
    }
    __syncthreads();  // wsum / whas / wlast are rewritten by the next evaluation
  };

  // ordinary collect of one record set (+ carry-in per group) into recm[0 .. cap) / red
  auto collect = [&](int call_of_set, unsigned tag) {
    const unsigned long long* sw = set_words(call_of_set);
    for (int i0 = 0; i0 < cap; i0 += kChainThreads * 2) {  // a thread's two words of a 16k-keyline map polled together
      const int ia = i0 + tid, ib = i0 + kChainThreads + tid;
      const unsigned long long* const w2[2] = {sw + min(ia, cap - 1), sw + min(ib, cap - 1)};
      const unsigned t2[2] = {tag, tag};
      const bool ok2[2] = {ia < cap && (ia & (kPartStride - 1)) < 12, ib < cap && (ib & (kPartStride - 1)) < 12};
      float o2[2];
      xch_wait_many<2>(w2, t2, ok2, o2, bar_err, slow_poll);
      if (ok2[0]) recm[ia] = o2[0];
      if (ok2[1]) recm[ib] = o2[1];
    }
    __syncthreads();
    reduce_staged_records(recm, nblocks, red, carry_in, (int)blockIdx.x * kChainGroups, kChainGroups);
    __syncthreads();
  };
  auto normal_pass = [&](int call, int phase) {
    if (call >= 1) {
      collect(call - 1, tag_of(call - 1, (phase == 1 && call - 1 >= fa + kf + 1) ? 1 : 0));
      if (tid < 64) lm_step_wave(s, red, call, false, lm_inv[0], tid);
      __syncthreads();
    }
    TvProj pj{};
    TvGeo ge{};
    if (idx < n) {
      pj = tv_project(p, in, s.Vnew[0], s.Vnew[1], s.Vnew[2], srm, thr, min_matches);
      ge = tv_geometry(p, nm, tv_cell(nm, pj));
    }
    do_eval(call, tag_of(call, phase), pj, ge, call == calls - 1);
  };

  const int nspec = calls - kf;  // evaluations kf .. calls - 1 (the launcher guarantees 2 <= nspec <= kSpecMax)
  // diagnostic: from the end of the previous launch (its last stamp) to the start of this one = the pair's second half with
  // every stream operation between the kernels; summed on the device (the launches of a stream are serialised)
  if (stamps && blockIdx.x == 0 && tid == 0) {
    const unsigned long long now = __builtin_amdgcn_s_memrealtime(), prev_end = stamps[15];
    if (stamps[0] == 1ull && prev_end != 0ull && now > prev_end && now - prev_end < 100000ull) {
      stamps[40] += now - prev_end;
      stamps[41] += 1ull;
      const unsigned long long g0 = stamps[14], g1 = stamps[16], g2 = stamps[17], g3 = stamps[18];
      if (g1 > g0 && g2 > g1 && g3 >= g2 && prev_end > g3) {  // glue of the previous launch: sums in, solve done, all waves done, end
        stamps[51] += g1 - g0;
        stamps[52] += g2 - g1;
        stamps[53] += g3 - g2;
        stamps[54] += prev_end - g3;
        stamps[55] += 1ull;
      }
      const unsigned long long t1 = stamps[48], t2 = stamps[49], t3 = stamps[50];
      if (t1 > prev_end && t2 > t1 && t3 > t2 && now > t3) {  // start of the head, the tail, regularize / EKF of that pair
        stamps[42] += t1 - prev_end;
        stamps[43] += t2 - t1;
        stamps[44] += t3 - t2;
        stamps[45] += now - t3;
        stamps[46] += 1ull;
        stamps[56] += now - t_entry;  // kernel entry -> here: the keyline loads and the sigma quantile ahead of the first evaluation
      }
    }
  }
  RH_STAMP(1);
  if (lm_live) {
    do_eval(0, tag_of(0, 0), pj0, ge0, false);  // (calls >= kf + 2: evaluation 0 is never the last one)
    RH_STAMP(2);
    for (int call = 1; call < kf; ++call) normal_pass(call, 0);
    RH_STAMP(3);
    collect(kf - 1, tag_of(kf - 1, 0));
    // the real step and the first three states under the hypothesis in one go (lm_step_wave_states)
    if (tid < 64) lm_step_wave_states(s, red, kf, lm_inv, tid, stc, min(nspec, 4));
    __syncthreads();
    // further states under the hypothesis (iterations > 5): state g + 1 applies g rejections' (u, v) updates and then one
    // lm_step whose score is NaN (the gain test fails for any denominator): exactly the reject branch of core.cpp:180-183
    // plus the next trial point. Sixteen lanes per state, nine of them with one cofactor of its 3x3 inverse each.
    for (int g0 = 3; g0 < nspec - 1; g0 += 4) {
      if (tid < 64) {
        const int g = g0 + (tid >> 4), e = tid & 15;
        const bool act = g < nspec - 1;
        LmState t = s;
        for (int j = 0; j < g; ++j) {
          t.u *= t.v;
          t.v = (float)((double)t.v * 2.0);
        }
        float fake[10];
#pragma unroll
        for (int i = 0; i < 10; ++i) fake[i] = __uint_as_float(0x7FC00000u);
        lm_step_book(t, fake, kf + 1 + g);
        float M[9];
        lm_step_matrix(t, M);
        const float d = glue_det3(M);
        float o;
        switch (e) {
          case 0: o = M[4] * M[8] - M[5] * M[7]; break;
          case 1: o = M[2] * M[7] - M[1] * M[8]; break;
          case 2: o = M[1] * M[5] - M[2] * M[4]; break;
          case 3: o = M[5] * M[6] - M[3] * M[8]; break;
          case 4: o = M[0] * M[8] - M[2] * M[6]; break;
          case 5: o = M[2] * M[3] - M[0] * M[5]; break;
          case 6: o = M[3] * M[7] - M[4] * M[6]; break;
          case 7: o = M[1] * M[6] - M[0] * M[7]; break;
          default: o = M[0] * M[4] - M[1] * M[3]; break;
        }
        glue_wave_sync();
        if (act && e < 9) lm_inv[tid >> 4][e] = o / d;
        glue_wave_sync();
        if (act && e == 0) {
          float inv[9];
#pragma unroll
          for (int i = 0; i < 9; ++i) inv[i] = lm_inv[tid >> 4][i];
          lm_step_apply(t, inv);
          stc[g + 1] = t;
        }
        glue_wave_sync();
      }
    }
    __syncthreads();
    RH_STAMP(4);
    // the gathers of all speculative evaluations in flight together: projections (no memory), then every field cell, then
    // every matched keyline's geometry; what remains per evaluation is arithmetic and the workgroup / neighbour hand-offs
    TvProj pj[kSpecMax];
    unsigned cellkey[kSpecMax];
    TvGeo ge[kSpecMax];
#pragma unroll
    for (int k = 0; k < kSpecMax; ++k) {
      pj[k] = TvProj{};
      pj[k].cell = -1;
      pj[k].skip = true;
      if (k < nspec && idx < n) pj[k] = tv_project(p, in, stc[k].Vnew[0], stc[k].Vnew[1], stc[k].Vnew[2], srm, thr, min_matches);
    }
    RH_STAMP(5);
#pragma unroll
    for (int k = 0; k < kSpecMax; ++k) cellkey[k] = tv_cell(nm, pj[k]);
#pragma unroll
    for (int k = 0; k < kSpecMax; ++k) ge[k] = tv_geometry(p, nm, cellkey[k]);
    // What an evaluation hands to the next one - who matched, the last written fi - does not depend on the residuals, so it is
    // settled for ALL speculative evaluations first: one workgroup barrier and one neighbour round instead of one per
    // evaluation. After that the residual / weight chain of a keyline is arithmetic on its own registers.
    RH_STAMP(6);
    TvMatch mt[kSpecMax];
    float rcarry[kSpecMax];  // residual an unmatched keyline ends evaluation k with (|fi| of the last match before it, or the marker)
#pragma unroll
    for (int k = 0; k < kSpecMax; ++k) {
      mt[k] = tv_match(p, in, pj[k], ge[k]);
      rcarry[k] = kResidualCarry;
      if (k < nspec) {
        const unsigned long long mm = __ballot(mt[k].matched);
        const unsigned long long below = mm & ((1ull << lane) - 1ull);
        const float fi_prev = __shfl(mt[k].fi, below ? (63 - __clzll((long long)below)) : 0);
        const float fi_wlast = __shfl(mt[k].fi, mm ? (63 - __clzll((long long)mm)) : 0);
        if (below) rcarry[k] = fabsf(fi_prev);
        if (lane == 0) {
          whask[k][wid] = mm ? 1 : 0;
          wlastk[k][wid] = fi_wlast;
        }
        if (kf + k == calls - 1 && idx < n) {  // the side effects of the last evaluation (forwardMatch, edge_map.cpp:78-96)
          if (mt[k].matched) {
            const unsigned long long key = ((unsigned long long)order_key(in.rs.x) << 32) | (unsigned)idx;
            fwd_key_max(&nm.fwd_key[mt[k].mfwd], key);
          }
          om.match_fwd[idx] = mt[k].mfwd;
        }
      }
    }
    // (the forwardMatch atomics issued above stay in flight through the neighbour round and the weighted sums; they are
    // waited for in front of the barrier the record sets are published behind)
    RH_STAMP(7);
    __syncthreads();
    // thread kChainThreads - 1 - k: hand-off behind evaluation kf + k (carry-in of evaluation kf + 1 + k), as in do_eval
    if (tid >= kChainThreads - (nspec - 1)) {
      const int k = kChainThreads - 1 - tid;
      const int call = kf + k;
      const unsigned tag = tag_of(call, 0);
      int own_has = 0;
      float own_last = 0.f;
      for (int w = kWaves - 1; w >= 0; --w)
        if (whask[k][w]) {
          own_has = 1;
          own_last = wlastk[k][w];
          break;
        }
      unsigned long long* cw = xch_carry + ((size_t)call * nwg + blockIdx.x) * 2;
      if (own_has && (int)blockIdx.x + 1 < live_wgs) {
        xch_publish(cw, tag, 1.0f);
        xch_publish(cw + 1, tag, own_last);
      }
      float in_has = 0.f, in_fi = 0.f;
      if (blockIdx.x > 0) {
        const unsigned long long* lw = xch_carry + ((size_t)call * nwg + (blockIdx.x - 1)) * 2;
        in_has = xch_wait(lw, tag, bar_err, slow_poll);
        in_fi = xch_wait(lw + 1, tag, bar_err, slow_poll);
      }
      if (!own_has && (int)blockIdx.x + 1 < live_wgs) {
        xch_publish(cw, tag, in_has);
        xch_publish(cw + 1, tag, in_fi);
      }
      float c = (in_has != 0.f) ? fabsf(in_fi) : 0.f;
      for (int g = 0; g < kChainGroups; ++g) {
        carryk[k + 1][g] = c;
        for (int w = 3; w >= 0; --w)
          if (whask[k][g * 4 + w]) {
            c = fabsf(wlastk[k][g * 4 + w]);
            break;
          }
      }
    }
    if (tid < kChainGroups) carryk[0][tid] = carry_in[tid];  // evaluation kf: from the records of evaluation kf - 1 (collect above)
    __syncthreads();
    RH_STAMP(8);
    float res_hist[kSpecMax];
#pragma unroll
    for (int k = 0; k < kSpecMax; ++k) {
      if (k < nspec) {
        TvOut e{};
        if (idx < n) {
          e = tv_weight(p, in, pj[k], mt[k], res, carryk[k][grp]);
          if (e.wrote_res) res = e.res_out;
        }
        if (mt[k].need_carry) {  // (false for idx >= n: those never pass tv_project)
          float r = rcarry[k];
          if (r == kResidualCarry) {
            for (int w = wig - 1; w >= 0; --w)
              if (whask[k][grp * 4 + w]) {
                r = fabsf(wlastk[k][grp * 4 + w]);
                break;
              }
          }
          res = r;
        }
        res_hist[k] = res;
        const float sc = e.contrib ? e.f * e.f : 0.f;
        float v[10] = {sc, e.jx * e.jx, e.jy * e.jy, e.jz * e.jz, e.jx * e.jy, e.jx * e.jz, e.jy * e.jz, e.jx * e.f, e.jy * e.f, e.jz * e.f};
        wave_total63_fN(v);
        if (lane == 63) {
#pragma unroll
          for (int q = 0; q < 10; ++q) wsumk[k][wid][q] = v[q];
        }
      }
    }
    // Every wave's forwardMatch atomics (performed at the memory side, device scope) are ACKNOWLEDGED before the workgroup goes
    // on to publish its records: vmcnt(0) here, on the waves that issued them, in front of the barrier. (A system-scope release
    // fence on the publishing threads orders only THEIR OWN operations and writes the XCD's whole L2 back for it.)
    vm_drain();
    RH_STAMP(9);
    __syncthreads();
    if (tid < kChainGroups * 16) {
      const int g = tid >> 4, q = tid & 15;
      for (int k = 0; k < nspec; ++k) {
        const int call = kf + k;
        const unsigned tag = tag_of(call, 0);
        unsigned long long* out = set_words(call) + ((size_t)blockIdx.x * kChainGroups + g) * kPartStride;
        if (q < 10) {
          float acc = 0.f;
          for (int w = 0; w < 4; ++w) acc += wsumk[k][g * 4 + w][q];
          xch_publish(out + q, tag, acc);
        } else if (q == 10 || q == 11) {
          float hv = 0.f, lv = 0.f;
          for (int w = 3; w >= 0; --w)
            if (whask[k][g * 4 + w]) {
              hv = 1.f;
              lv = wlastk[k][g * 4 + w];
              break;
            }
          xch_publish(out + q, tag, q == 10 ? hv : lv);
        }
      }
    }
    __syncthreads();  // (the roll-back path reuses the staging arrays)
    RH_STAMP(10);
    // all nspec record sets at once: every thread first issues its loads together, then waits for the stragglers
    {
      const int total = nspec * cap;
      for (int i0 = 0; i0 < total; i0 += kChainThreads * 8) {
        const unsigned long long* w8[8];
        unsigned t8[8];
        bool ok8[8];
        float o8[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int i = min(i0 + j * kChainThreads + tid, total - 1);
          const int set = i / cap, w = i - set * cap;
          ok8[j] = i0 + j * kChainThreads + tid < total && (w & (kPartStride - 1)) < 12;
          w8[j] = set_words(kf + set) + w;
          t8[j] = tag_of(kf + set, 0);
        }
        xch_wait_many<8>(w8, t8, ok8, o8, bar_err, slow_poll);
#pragma unroll
        for (int j = 0; j < 8; ++j)
          if (ok8[j]) recm[i0 + j * kChainThreads + tid] = o8[j];
      }
      __syncthreads();
      // Every workgroup's record sets are in, hence every forwardMatch key of the last evaluation (vm_drain above): this
      // keyline's key is fetched HERE, under the hypothesis, and returns while the sums are reduced and checked; the gathers
      // through it follow the check (below) and return while the state is finished.
      xch_acquire();
      if (idx < n_new) xk.key = __hip_atomic_load(&nm.fwd_key[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      for (int set0 = 0; set0 < nspec; set0 += kChainThreads / 160) {
        const int sl = tid / 160, set = set0 + sl;
        if (sl < kChainThreads / 160 && set < nspec) reduce_staged_sums(recm + (size_t)set * cap, nblocks, redm[set], sl * 160);
      }
      __syncthreads();
    }
    RH_STAMP(11);
    // the hypothesis against the real scores: the gain test of core.cpp:172 on the state the evaluation started from
    {
      bool accepted = false;
      if (tid < nspec) {
        const LmState& t = stc[tid];
        const float Fnew = redm[tid][0];
        double den = 0.0;
        for (int i = 0; i < 3; ++i) den += (0.5 * (double)t.h[i]) * (double)(t.u * t.h[i] - t.JtF[i]);
        const float gain = (float)((double)(t.F - Fnew) / den);
        accepted = gain > 0.0f;
      }
      if (wid == 0) {
        const unsigned long long am = __ballot(accepted);
        if (lane == 0) first_accept = am ? (__ffsll((long long)am) - 1) : -1;
      }
      __syncthreads();
    }
    fa = first_accept;
    RH_STAMP(12);
    if (fa < 0 || fa == nspec - 1) {  // the keys of the speculative last evaluation stand
      if (idx < n_new) xf = xrv_gather(om, xk.key);
      xf_valid = true;
    }
    if (fa < 0) {
      if (tid == 0) {  // every evaluation rejected: the last rejection is the final step of core.cpp:166-185 (no new trial point)
        LmState t = stc[nspec - 1];
        t.u *= t.v;
        t.v = (float)((double)t.v * 2.0);
        s = t;
      }
      __syncthreads();
    } else {
      // evaluation kf + fa was accepted: go on from there the ordinary way. Roll back, clear the keys the speculative last
      // evaluation published, and let nobody publish a repeated evaluation before everybody has read the speculative sets.
      __atomic_thread_fence(__ATOMIC_ACQUIRE);  // the keys other workgroups issued before their last record set
      if (tid == 0) s = stc[fa];
#pragma unroll
      for (int k = 0; k < kSpecMax; ++k)
        if (k == fa) res = res_hist[k];
      // (fa == nspec - 1: the accepted evaluation is the last one - its trial point was the real one, its keys stand)
      if (fa < nspec - 1)
        for (int i = blockIdx.x * kChainThreads + tid; i < p.kmax; i += live_wgs * kChainThreads) nm.fwd_key[i] = 0ull;
      __syncthreads();
      if (tid == 0) {
        __atomic_thread_fence(__ATOMIC_RELEASE);
        xch_publish(xch_sync + blockIdx.x, tag_final, 1.0f);
      }
      for (int i = tid; i < live_wgs; i += kChainThreads) (void)xch_wait(xch_sync + i, tag_final, bar_err, slow_poll);
      __syncthreads();
      __atomic_thread_fence(__ATOMIC_ACQUIRE);
      for (int call = kf + 1 + fa; call < calls; ++call) normal_pass(call, 1);
      collect(calls - 1, tag_of(calls - 1, (calls - 1 >= fa + kf + 1) ? 1 : 0));
      if (tid == 0) lm_step(s, red, calls, true);
      __syncthreads();
    }
  }
  // The final state and the pair's result slot (zero-copy: the host reads them once the stamps say so) are written by the LAST
  // live workgroup, behind its extRotVel records: every live workgroup holds the same state, and workgroup 0, which goes on
  // to the device glue and is waited for by it, no longer spends a microsecond of a lone thread's stores in front of its rows.
  const int fill_wg = live_wgs - 1;
  RH_STAMP(13);
  if (hist_to_zero && blockIdx.x == 0 && tid < 128) hist_to_zero[tid] = 0;  // every evaluation has consumed the histogram
  if (blockIdx.x == 0 && tid < 3) xch_publish(xch_final + tid, tag_final, s.vel[tid]);
  if (!lm_live) {
    if (tid < 3) s.vel[tid] = xch_wait(xch_final + tid, tag_final, bar_err, slow_poll);
    __syncthreads();
  }
  const float vx = s.vel[0], vy = s.vel[1], vz = s.vel[2];
  float row[6] = {0, 0, 0, 0, 0, 0};
  float Y = 0.f;
  int cnt = 0;
  if (!xf_valid) {  // (workgroups without LM work, or after a roll-back)
    xch_acquire();  // behind the polls of the last evaluation's records: the forwardMatch keys of every workgroup (vm_drain)
    if (idx < n_new) {
      xk.key = __hip_atomic_load(&nm.fwd_key[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      xf = xrv_gather(om, xk.key);
    }
  }
  if (idx < n_new) xrv_apply(p, nm, idx, 1, xf, xk, vx, vy, vz, row, &Y, &cnt);
  float v[28];
  {
    int k = 0;
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
      for (int j = i; j < 6; ++j) v[k++] = row[i] * row[j];
#pragma unroll
    for (int i = 0; i < 6; ++i) v[21 + i] = row[i] * Y;
    v[27] = (float)cnt;
  }
  wave_total63_fN(v);
  if (lane == 63) {
#pragma unroll
    for (int k = 0; k < 28; ++k) wsum[wid][k] = v[k];
  }
  __syncthreads();
  unsigned long long* xch_xrv = xch + lm_xch_xrv_offset((size_t)nrec_launched, (size_t)nwg);
  if (tid < kChainGroups * 32) {
    const int g = tid >> 5, k = tid & 31;
    unsigned stored = 0u;
    if (k < 28) {
      float acc = 0.f;
      for (int w = 0; w < 4; ++w) acc += wsum[g * 4 + w][k];
      if (ga.lm)
        xch_publish(xch_xrv + ((size_t)blockIdx.x * kChainGroups + g) * kXrvStride + k, tag_final, acc);
      else
        xrv_part[((size_t)blockIdx.x * kChainGroups + g) * kXrvStride + k] = acc;
      stored = __float_as_uint(acc);
    }
    // records that go straight to the host (per-pair API): each carries the pair's sequence stamp in its last word, stored once
    // this wave's 28 sums are acknowledged - a host that polls the records instead of waiting for an event reads complete ones
    if (!ga.lm) xrv_record_stamp(xrv_part + ((size_t)blockIdx.x * kChainGroups + g) * kXrvStride, k, ga.seq, stored);
  }
  RH_STAMP(14);
  if ((int)blockIdx.x == fill_wg && tid == 0) {
    *st_out = s;
    if (slot) slot_sum = slot_fill(slot, s, nm.st, om.st);
  }
  if (stamps && blockIdx.x == 0 && tid == 0) {
    stamps[15] = stamps[14];  // (no device glue in this launch: a zero-length last segment)
    stamps[0] = 1ull;
  }
  if (ga.lm) {
    __syncthreads();  // (recm is free: every LM collect is over)
    // recm holds (calls - 2) record sets of 16 words per launched group: at least 32 floats per group
    lm_tail_glue<kChainThreads>(ga, xch_xrv, tag_final, n_new, recm, (calls - 2) * (cap / kPartStride) * kPartStride / kXrvStride > 0
                                    ? (calls - 2) * (cap / kPartStride) * kPartStride / kXrvStride : 1, gw, s, bar_err, slow_poll, stamps);
    RH_STAMP(15);
  }
  if (slot && (int)blockIdx.x == fill_wg && tid == 0) {  // the slot's stamp: by the directedMatch launch with the glue's record, or here
    slot->sum = slot_sum ^ ga.seq;
    if (ga.lm && ga.stage)
      ga.stage->host_slot = slot;
    else
      stamp_drain(&slot->seq, ga.seq);
  }
#undef RH_STAMP
}

template <int kChainThreads>
__global__ __launch_bounds__(kChainThreads) void k_lm_chain_spec(KParams p, MapDev om, MapDev nm, int calls, const LmState* __restrict__ st_in,
                                                                 LmState* __restrict__ st_out, unsigned long long* __restrict__ xch,
                                                                 unsigned tag_base, int* __restrict__ bar_err, const int* __restrict__ hist,
                                                                 float* __restrict__ xrv_part, PairSlot* __restrict__ slot,
                                                                 int* __restrict__ hist_to_zero, unsigned long long* __restrict__ stamps,
                                                                 GlueArgs ga, int kf) {
  lm_chain_spec_body<kChainThreads>(p, om, nm, calls, st_in, st_out, xch, tag_base, bar_err, hist, 0u, xrv_part, slot, hist_to_zero, stamps,
                                    0, ga, kf);
}
template <int kChainThreads>
__global__ __launch_bounds__(kChainThreads) void k_lm_chain_spec_b(KParams p, const LaneStatic* __restrict__ ls,
                                                                   const MapDev* __restrict__ maptab, LaneDynB dyn, int calls, int slow_poll,
                                                                   GlueParams gp, int lane0, int kf) {
  const int lane = lane0 + (int)blockIdx.z;  // a launch carries the lanes that fit the device together (launch_lm_chain_b)
  const LaneStatic& L = ls[lane];
  const LaneDyn d = dyn.v[lane];
  PairSlot* slot = gptr(L.slot[d.slot]);
  lm_chain_spec_body<kChainThreads>(p, global_map(lane_map(maptab, lane, d.om, d.om_swap)), global_map(lane_map(maptab, lane, d.nm, d.nm_swap)), calls,
                                    gptr(L.lm_zero), gptr(L.lm) + calls + 1, gptr(L.lm_xch), d.tag_base, gptr(L.lm_bar_err), gptr(L.hist), 0u, gptr(L.xrv_part), slot, gptr(L.hist),
                                    nullptr, slow_poll, lane_glue_args(L, d, calls, gp), kf);
}

template <int kChainThreads>
__global__ __launch_bounds__(kChainThreads) void k_lm_chain(KParams p, MapDev om, MapDev nm, int calls, int do_ext,
                                                            const LmState* __restrict__ st_in, LmState* __restrict__ st_out,
                                                            unsigned long long* __restrict__ xch, unsigned tag_base,
                                                            int* __restrict__ bar_err, const int* __restrict__ hist,
                                                            unsigned frame_count, float* __restrict__ xrv_part,
                                                            PairSlot* __restrict__ slot, int* __restrict__ hist_to_zero,
                                                            unsigned long long* __restrict__ stamps, GlueArgs ga) {
  lm_chain_body<kChainThreads>(p, om, nm, calls, do_ext, st_in, st_out, xch, tag_base, bar_err, hist, frame_count, xrv_part, slot,
                               hist_to_zero, stamps, 0, ga);
}
// batched form (lane = blockIdx.z): every lane's workgroups exchange records among themselves through the lane's own words
template <int kChainThreads>
__global__ __launch_bounds__(kChainThreads) void k_lm_chain_b(KParams p, const LaneStatic* __restrict__ ls,
                                                              const MapDev* __restrict__ maptab, LaneDynB dyn, int calls, int slow_poll,
                                                              GlueParams gp, int lane0) {
  const int lane = lane0 + (int)blockIdx.z;  // a launch carries the lanes that fit the device together (launch_lm_chain_b)
  const LaneStatic& L = ls[lane];
  const LaneDyn d = dyn.v[lane];
  PairSlot* slot = gptr(L.slot[d.slot]);
  lm_chain_body<kChainThreads>(p, global_map(lane_map(maptab, lane, d.om, d.om_swap)), global_map(lane_map(maptab, lane, d.nm, d.nm_swap)), calls, 1,
                               gptr(L.lm_zero), gptr(L.lm) + calls + 1, gptr(L.lm_xch), d.tag_base, gptr(L.lm_bar_err), gptr(L.hist), 0u, gptr(L.xrv_part), slot, gptr(L.hist), nullptr,
                               slow_poll, lane_glue_args(L, d, calls, gp));
}

// The device glue behind the per-call kernels (REBVIO_HIP_LM=percall): one workgroup; the extRotVel records are in memory
// (kernel boundary), summed in group order in double like hm::sum_xrv.
__global__ __launch_bounds__(256) void k_pair_glue(MapDev nm, GlueArgs ga) {
  __shared__ GlueLds gw;
  __shared__ LmState s;
  const int tid = threadIdx.x;
  if (tid == 0) s = *ga.lm;
  const int nb = (nm.st->n + 255) / 256;
  if (tid < 27) {
    double acc = 0.0;
    for (int b = 0; b < nb; ++b) acc += (double)ga.xrv[(size_t)b * kXrvStride + tid];
    if (tid < 21) {
      int i = 0, base = 0;
      while (tid >= base + (6 - i)) {
        base += 6 - i;
        ++i;
      }
      const int j = i + (tid - base);
      gw.W[i * 6 + j] = (float)acc;
      gw.W[j * 6 + i] = (float)acc;
    } else {
      gw.JtF[tid - 21] = (float)acc;
    }
  }
  __syncthreads();
  glue_workgroup(gw, s, ga);
}
void launch_pair_glue(hipStream_t s, const MapDev& newm, const GlueArgs& ga) { RH_LAUNCH(k_pair_glue, dim3(1), dim3(256), 0, s, newm, ga); }

// ---- EdgeMap::directedMatch / searchMatch (edge_map.cpp:101-218) ----------------------------------------------
// Probe geometry of one query keyline (edge_map.cpp:104-147): everything up to the probe loop.
struct SearchSetup {
  float t_x, t_y, norm_t, pi0x, pi0y, sigma2_t, dq_min, dq_max, dq_rho;
  int t_steps;
};

__device__ __forceinline__ SearchSetup search_setup(const KParams& p, float2 pi, float2 rsq, float2 gq, float gnq,
                                                    const Vec3& vel, const Mat3& Rvel, const Mat3& Rback, float max_radius) {
  SearchSetup S;
  float p_m3[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    float s = 0.f;
    s += Rback.a[i * 3 + 0] * pi.x;
    s += Rback.a[i * 3 + 1] * pi.y;
    s += Rback.a[i * 3 + 2] * p.fm;
    p_m3[i] = s;
  }
  const float pmx = p_m3[0] * p.fm / p_m3[2];
  const float pmy = p_m3[1] * p.fm / p_m3[2];
  const float k_rho = rsq.x * p.fm / p_m3[2];
  S.pi0x = pmx + p.cx;
  S.pi0y = pmy + p.cy;
  float t_x = -(vel.a[0] * p.fm - vel.a[2] * pmx);
  float t_y = -(vel.a[1] * p.fm - vel.a[2] * pmy);
  float norm_t = sqrtf(t_x * t_x + t_y * t_y);
  const float DrDv[3] = {p.fm, p.fm, -(pmx + pmy)};
  float rowv[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    float s = 0.f;
    s += DrDv[0] * Rvel.a[0 * 3 + j];
    s += DrDv[1] * Rvel.a[1 * 3 + j];
    s += DrDv[2] * Rvel.a[2 * 3 + j];
    rowv[j] = s;
  }
  float sigma2_t = 0.f;
  sigma2_t += rowv[0] * DrDv[0];
  sigma2_t += rowv[1] * DrDv[1];
  sigma2_t += rowv[2] * DrDv[2];
  S.sigma2_t = sigma2_t;
  // std::max / std::min on NaN operands keep the first argument when the comparison is false; fmaxf/fminf return the
  // non-NaN one. The operands here are finite whenever rho/sigma are, which the depth filter guarantees (core.cpp:451-455).
  if ((double)norm_t > 1e-6) {
    t_x /= norm_t;
    t_y /= norm_t;
    S.dq_rho = norm_t * k_rho;
    S.dq_min = fmaxf(0.0f, norm_t * (k_rho - rsq.y)) - p.pixel_uncertainty_match;
    S.dq_max = fminf(max_radius, norm_t * (k_rho + rsq.y)) + p.pixel_uncertainty_match;
    if (S.dq_rho > S.dq_max) {
      S.dq_rho = (float)(0.5 * (double)(S.dq_max + S.dq_min));
      S.t_steps = cvtt_f64((double)S.dq_rho + 0.5);
    } else {
      S.t_steps = cvtt_f32(fmaxf(S.dq_max - S.dq_rho, S.dq_rho - S.dq_min));
    }
  } else {
    t_x = gq.x;
    t_y = gq.y;
    norm_t = gnq;
    t_x /= norm_t;
    t_y /= norm_t;
    norm_t = 1.0f;
    S.dq_min = -max_radius - p.pixel_uncertainty_match;
    S.dq_max = max_radius + p.pixel_uncertainty_match;
    S.dq_rho = 0.0f;
    S.t_steps = cvtt_f32(S.dq_max);
  }
  S.t_x = t_x;
  S.t_y = t_y;
  S.norm_t = norm_t;
  return S;
}

// Acceptance test of one candidate (edge_map.cpp:170-177)
__device__ __forceinline__ bool search_accept(const KParams& p, const SearchSetup& S, float t, float2 cg, float cgn, float2 crs,
                                              float2 gq, float gnq) {
  const float cang = (cg.x * gq.x + cg.y * gq.y) / (cgn * gnq);
  if (cang < p.cang_min_edge || fabs((double)(cgn / gnq) - 1.0) > (double)p.match_threshold_norm) return false;
  const float v_rho_dr = (p.pixel_uncertainty_match * p.pixel_uncertainty_match + crs.y * crs.y * S.norm_t * S.norm_t +
                          S.sigma2_t * crs.x * crs.x);
  if ((t - S.norm_t * crs.x) * (t - S.norm_t * crs.x) > v_rho_dr) return false;
  return true;
}

// Old-map keyline as directedMatch sees it: stored fields, or (rot != 0) rotated on the fly by R0 - the second
// rotateKeylines of rebvio.cpp:232 folded into the match (the old map is dead afterwards, rebvio.cpp:126-127).
struct OldKl {
  float2 pi, rs, g;
  float gn;
  unsigned matches;  // fetched with the rest (same gather round trip) so that the commit of the accepted candidate does
  int kf;            // not start another one
};
__device__ __forceinline__ OldKl load_old(const MapDev& om, int i, int rot, const Mat3& R0, float fm) {
  OldKl k;
  k.pi = om.pos_img[i];
  k.rs = om.rs[i];
  k.g = om.grad[i];
  k.gn = om.gnorm[i];
  k.matches = om.matches[i];
  k.kf = om.match_kf[i];
  if (rot) rotate_one(R0, fm, k.pi, k.rs, k.g);
  return k;
}

__device__ __forceinline__ void search_commit(MapDev& nm, const MapDev& om, int idx, int found, const OldKl& k, int* kf) {
  (void)om;
  nm.rs[idx] = k.rs;
  nm.match_id[idx] = found;
  nm.matches[idx] = k.matches + 1u;
  nm.mpos_img[idx] = k.pi;
  nm.mgrad[idx] = k.g;
  nm.mgnorm[idx] = k.gn;
  nm.match_kf[idx] = k.kf;
  *kf = (k.kf >= 0) ? 1 : 0;
}

constexpr int kHeadSteps = 4;  // probe steps (2 probes each) of the head of a search: what most keylines are settled in

// vel / Rvel are already rotated by Rback on the host (edge_map.cpp:193-194).
// gd != null: vel / Rvel / Rback / R0 come from *gd (uniform scalar loads) instead of the kernel arguments.
struct DmArgs {
  Vec3 vel;
  Mat3 Rvel, Rback, R0;
  int rot;
  bool skip;
};
__device__ __forceinline__ DmArgs dm_args(const GlueDev* __restrict__ gd, const Vec3& vel, const Mat3& Rvel, const Mat3& Rback, int rot,
                                          const Mat3& R0) {
  DmArgs a;
  if (gd) {
#pragma unroll
    for (int i = 0; i < 3; ++i) a.vel.a[i] = gd->vel_r[i];
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      a.Rvel.a[i] = gd->Rvel_r[i];
      a.Rback.a[i] = gd->Rgva[i];
      a.R0.a[i] = gd->R0a[i];
    }
    a.rot = 1;
    a.skip = gd->nan_v != 0;
  } else {
    a.vel = vel;
    a.Rvel = Rvel;
    a.Rback = Rback;
    a.R0 = R0;
    a.rot = rot;
    a.skip = false;
  }
  return a;
}

// ---- directedMatch (round 4): ONE launch, dense lanes in every expensive phase -----------------------------------------------
// The two-launch forms of rounds 1-3 (a head kernel with one thread or eight lanes per keyline for the first four probe steps,
// a queue of the searches still open, a tail kernel with one wave per queued search; removed, DESIGN.md 6e has their numbers)
// spent most of their instructions on mostly idle lanes: a probe finds a keyline of the old map on
// ~5 % of the pixels it looks at (1.5 of the eight head probes of a keyline, 2 of a long search's ~35), yet the candidate
// fetch + second rotateKeylines + acceptance test (six IEEE divisions, a double-accumulated 3x3 product: ~350 instructions)
// runs once per probe SLOT of the wave, and the long searches take a second launch whose waves hold one keyline each behind
// a queue. Here a wave owns 64 / kLPK keylines of the new map through the whole of searchMatch (edge_map.cpp:101-184), kLPK
// lanes per keyline:
//   phase 1  the kLPK lanes of a keyline share the first kHeadSteps steps (2 * kHeadSteps / kLPK independent mask loads per
//            lane); the wave's candidates are COMPACTED into an LDS list {old keyline, owner, slot} and tested 64 at a time,
//            one candidate per lane; an accepted candidate bids for its owner with an LDS atomicMin on (slot, list position):
//            the smallest slot index is the reference's first hit; the winner's rotated fields are staged per owner in LDS and
//            committed by the owner's first lane.
//   phase 2  the keylines still open after that (long searches: t_steps > kHeadSteps, nothing accepted; 2 600 of 15 000 in the
//            steady state, mean t_steps 20) are searched up to kDmcOpen at a time: two lanes of each walk the reference's two
//            +-1.0f chains (edge_map.cpp:149-150: repeated ++tp / --tn, not dq_rho +- k) into LDS, then ALL 64 lanes share the
//            batch's probe slots, the candidates are compacted and tested as in phase 1.
// Testing every candidate of a keyline instead of stopping at the first accepted one changes nothing (the minimum slot wins)
// and costs little: the tests are what is dense now. Everything is wave-local (no workgroup barrier before the counters).
// kLPK = 4 is the low-latency choice of one stream (938 waves of 16 keylines: a wave's long searches fit one batch - the
// 64-keyline form measured 29 us for phase 2 because waves with 40+ open keylines search them 16 at a time, one after the other);
// kLPK = 1 executes the fewest instructions (batches, large maps).
constexpr int kDmcWin = 40;        // probe steps per long-search window
constexpr int kDmcWinPitch = 41;   // LDS pitch of one chain window (odd: the owners' rows start in different banks)
constexpr int kDmcOpen = 16;       // open keylines a wave searches together (fewer where it owns fewer)
constexpr int kDmcNone = 0x7fffffff;
template <int kKPW, int kBatch>  // keylines per wave (owners), open keylines searched together
struct DmcWave {
  static constexpr int kList = (kBatch * 2 * kDmcWin > kKPW * 2 * kHeadSteps) ? kBatch * 2 * kDmcWin : kKPW * 2 * kHeadSteps;
  unsigned list[kList];              // candidate: old keyline (16 bits) | owner (6) | slot in the reference's probe order, 2 * step + side (10)
  float t1[kKPW * 2 * kHeadSteps];   // phase 1: t of the candidate's probe (phase 2 reads it from seq)
  float4 q_a[kKPW];                  // per owner, what the acceptance test needs of the query keyline: norm_t, sigma2_t, gradient
  float q_gn[kKPW];                  // ... and its norm
  int best[kKPW];                    // per owner: min over accepted candidates of slot << 16 | list position
  float w_f[7][kKPW];                // staged winner per owner: pos_img, (rho, sigma_rho), gradient, norm (rotated) ...
  int w_i[3][kKPW];                  // ... matches, match_id_keyframe, old keyline index
  float4 s_a[kBatch];                // probe geometry of the open keylines: t_x, t_y, pi0x, pi0y
  float4 s_b[kBatch];                // dq_min, dq_max, t_steps (int bits), -
  float seq[kBatch][2][kDmcWinPitch];  // [open keyline][0: tn, 1: tp][step - window start]
};
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// Exclusive prefix sum over the wave through the DPP data path (row_shr 1, 2, 4, 8 inside each row of 16 lanes, then row_bcast 15
// into rows 1 and 3 and row_bcast 31 into rows 2 and 3: twelve VALU instructions, no LDS crossbar); *total is a scalar.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_add_i(int v) {
  return v + __builtin_amdgcn_update_dpp(0, v, CTRL, ROW_MASK, 0xF, true);
}
__device__ __forceinline__ int wave_excl_scan_i(int v, int lane, int* total) {
  (void)lane;
  int incl = v;
  incl = dpp_add_i<0x111, 0xF>(incl);  // row_shr:1
  incl = dpp_add_i<0x112, 0xF>(incl);  // row_shr:2
  incl = dpp_add_i<0x114, 0xF>(incl);  // row_shr:4
  incl = dpp_add_i<0x118, 0xF>(incl);  // row_shr:8
  incl = dpp_add_i<0x142, 0xA>(incl);  // row_bcast:15 -> rows 1, 3
  incl = dpp_add_i<0x143, 0xC>(incl);  // row_bcast:31 -> rows 2, 3
  *total = __builtin_amdgcn_readlane(incl, 63);  // (a scalar: loops over the total are uniform)
  return incl - v;
}
// Acceptance test of one candidate (edge_map.cpp:170-177) on the two query scalars it needs of the probe geometry.
__device__ __forceinline__ bool search_accept_q(const KParams& p, float norm_t, float sigma2_t, float t, float2 cg, float cgn, float2 crs,
                                                float2 gq, float gnq) {
  SearchSetup S;
  S.norm_t = norm_t;
  S.sigma2_t = sigma2_t;
  return search_accept(p, S, t, cg, cgn, crs, gq, gnq);
}
// Test the T listed candidates of a wave, 64 at a time; accepted ones bid for their owner, the current best of an owner is staged.
template <bool kPhase2, class DmcWaveT>
__device__ __forceinline__ void dmc_test_list(const KParams& p, const MapDev& om, DmcWaveT& W, int T, int lane, int rot, const Mat3& R0,
                                              int step0) {
  for (int e0 = 0; e0 < T; e0 += 64) {
    const int e = e0 + lane;
    bool acc = false;
    OldKl ck{};
    int owner = 0, bid = 0, c = -1;
    if (e < T) {
      const unsigned ent = W.list[e];
      c = (int)(ent >> 16);
      owner = (int)((ent >> 10) & 63u);
      const int slot = (int)(ent & 1023u);
      const float t = kPhase2 ? W.seq[owner][slot & 1][(slot >> 1) - step0] : W.t1[e];
      ck = load_old(om, c, rot, R0, p.fm);
      const float4 qa = W.q_a[owner];
      acc = search_accept_q(p, qa.x, qa.y, t, ck.g, ck.gn, ck.rs, make_float2(qa.z, qa.w), W.q_gn[owner]);
      bid = (slot << 16) | e;
      if (acc) atomicMin(&W.best[owner], bid);
    }
    wave_lds_sync();
    if (acc && W.best[owner] == bid) {  // best so far of this owner (a later 64 may still replace it: same wave, program order)
      W.w_f[0][owner] = ck.pi.x; W.w_f[1][owner] = ck.pi.y; W.w_f[2][owner] = ck.rs.x; W.w_f[3][owner] = ck.rs.y;
      W.w_f[4][owner] = ck.g.x; W.w_f[5][owner] = ck.g.y; W.w_f[6][owner] = ck.gn;
      W.w_i[0][owner] = (int)ck.matches; W.w_i[1][owner] = ck.kf; W.w_i[2][owner] = c;
    }
    wave_lds_sync();
  }
}
template <class DmcWaveT>
__device__ __forceinline__ int dmc_commit(MapDev& nm, const MapDev& om, DmcWaveT& W, int owner, int idx, int* kf) {
  OldKl k;
  k.pi = make_float2(W.w_f[0][owner], W.w_f[1][owner]);
  k.rs = make_float2(W.w_f[2][owner], W.w_f[3][owner]);
  k.g = make_float2(W.w_f[4][owner], W.w_f[5][owner]);
  k.gn = W.w_f[6][owner];
  k.matches = (unsigned)W.w_i[0][owner];
  k.kf = W.w_i[1][owner];
  const int found = W.w_i[2][owner];
  search_commit(nm, om, idx, found, k, kf);
  return found;
}

template <int kThreads, int kLPK>
__device__ __forceinline__ void directed_match_c_body(KParams p, MapDev nm, MapDev om, Vec3 vel_, Mat3 Rvel_, Mat3 Rback_, float max_radius,
                                                      int rot_, Mat3 R0_, const GlueDev* __restrict__ gd, const GlueStage* __restrict__ stage) {
  static_assert(kLPK == 1 || kLPK == 2 || kLPK == 4 || kLPK == 8, "lanes per keyline");
  constexpr int kWaves = kThreads / 64;
  constexpr int kKPW = 64 / kLPK;                 // keylines per wave
  constexpr int kHP = 2 * kHeadSteps / kLPK;      // head probes per lane
  constexpr int kBatch = kKPW < kDmcOpen ? kKPW : kDmcOpen;
  static_assert(kMaxRecBlocks * 256 <= 65536, "old keyline index in 16 bits of a list entry");
  __shared__ DmcWave<kKPW, kBatch> lds[kWaves];
  const uint2 vb = xcd_band_block();
  if (p.dbg && vb.x == 0 && threadIdx.x == 0) p.dbg[48] = __builtin_amdgcn_s_memrealtime();
  const bool stats = p.dm_stats != nullptr;
  unsigned long long tk[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t_prev = stats ? __builtin_amdgcn_s_memrealtime() : 0ull;
#define RH_DMC_TICK(i)                                               \
  if (stats) {                                                       \
    const unsigned long long now_ = __builtin_amdgcn_s_memrealtime(); \
    tk[i] += now_ - t_prev;                                          \
    t_prev = now_;                                                   \
  }
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int kl = lane / kLPK, sub = lane % kLPK;  // keyline of the wave, lane of the keyline
  DmcWave<kKPW, kBatch>& W = lds[wid];
  const int idx = (vb.x * kWaves + wid) * kKPW + kl;
  // bound-free early loads (arrays padded to the grid), issued before the parameter block is read
  const float2 pi = nm.pos_img[idx];
  const float2 rsq = nm.rs[idx];
  const float2 gq = nm.grad[idx];
  const float gnq = nm.gnorm[idx];
  const int n = nm.st->n;
  if (stage && vb.x == gridDim.x - 1 && wid == kWaves - 1) {
    // The pair's host record, left in device memory by the glue at the tail of the LM kernel (GlueStage): this wave - the last
    // of the grid, which owns no keyline unless the map is full - copies it into the pinned record while the launch works, waits
    // for its stores to be acknowledged and stamps the record and the pair's result slot (what the host compares, -12).
    const unsigned* src = reinterpret_cast<const unsigned*>(&stage->rec);
    GlueRec* hr = stage->host_rec;
    unsigned* dst = reinterpret_cast<unsigned*>(hr);
    constexpr int kWords = (int)(offsetof(GlueRec, seq_gs) / sizeof(unsigned));
    for (int i = lane; i < kWords; i += 64) dst[i] = src[i];
    const unsigned seq = stage->seq;
    PairSlot* hs = stage->host_slot;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): every lane's stores acknowledged (see stamp_drain)
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) {
      __hip_atomic_store(&hr->seq_gs, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store(&hr->seq_out, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      if (hs) __hip_atomic_store(&hs->seq, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
  const DmArgs A = dm_args(gd, vel_, Rvel_, Rback_, rot_, R0_);
  if (A.skip) return;
  const bool live = idx < n;
  SearchSetup S{};
  if (live) S = search_setup(p, pi, rsq, gq, gnq, A.vel, A.Rvel, A.Rback, max_radius);
  // ---- phase 1: the first kHeadSteps steps (edge_map.cpp:149-181). Slot 2 * step + side belongs to lane slot / kHP of the keyline;
  // every lane walks both chains through all kHeadSteps steps (8 adds) and keeps the values of its own slots.
  float tn = S.dq_rho, tp = S.dq_rho + 1.0f;
  float tq[kHP];
  int cand[kHP];
  {
    int prow[kHP], pcol[kHP];
#pragma unroll
    for (int h = 0; h < kHP; ++h) {
      prow[h] = -1;
      pcol[h] = 0;
      tq[h] = 0.f;
    }
#pragma unroll
    for (int j = 0; j < kHeadSteps; ++j) {
      const bool active = live && j < S.t_steps;
#pragma unroll
      for (int i_idx = 0; i_idx < 2; ++i_idx) {
        const int slot = 2 * j + i_idx;
        const int h = slot % kHP;  // compile-time after unrolling
        if (slot / kHP == sub) {
          const float t = i_idx ? tp : tn;
          const bool ok = active && (i_idx ? !(t > S.dq_max) : !(t < S.dq_min));
          if (ok) {
            const int row = cvtt_f32(roundf(S.t_y * t + S.pi0y));
            const int col = cvtt_f32(roundf(S.t_x * t + S.pi0x));
            if ((unsigned)row < (unsigned)p.rows && (unsigned)col < (unsigned)p.cols) {
              prow[h] = row;
              pcol[h] = col;
            }
          }
          tq[h] = t;
        }
      }
      tp += 1.0f;
      tn -= 1.0f;
    }
#pragma unroll
    for (int h = 0; h < kHP; ++h) cand[h] = (prow[h] >= 0) ? om.mask[(size_t)prow[h] * p.cols + pcol[h]] : -1;
  }
  if (sub == 0) {
    W.best[kl] = kDmcNone;
    W.q_a[kl] = make_float4(S.norm_t, S.sigma2_t, gq.x, gq.y);
    W.q_gn[kl] = gnq;
  }
  int cnt = 0;
#pragma unroll
  for (int h = 0; h < kHP; ++h) cnt += (cand[h] >= 0) ? 1 : 0;
  RH_DMC_TICK(0)  // set-up, head probes issued
  int T = 0;
  int pos = wave_excl_scan_i(cnt, lane, &T);
  RH_DMC_TICK(1)  // mask loads back, scan
#pragma unroll
  for (int h = 0; h < kHP; ++h) {
    if (cand[h] >= 0) {
      W.list[pos] = ((unsigned)cand[h] << 16) | ((unsigned)kl << 10) | (unsigned)(sub * kHP + h);
      W.t1[pos] = tq[h];
      ++pos;
    }
  }
  wave_lds_sync();
  dmc_test_list<false>(p, om, W, T, lane, A.rot, A.R0, 0);
  int found = -1, kf = 0;
  if (sub == 0 && W.best[kl] != kDmcNone) found = dmc_commit(nm, om, W, kl, idx, &kf);
  RH_DMC_TICK(2)  // head candidates tested, commits issued
  // ---- phase 2: long searches still open (flagged on the first lane of the keyline)
  const bool open = live && sub == 0 && found < 0 && S.t_steps > kHeadSteps;
  const unsigned long long open_mask = __ballot(open);
  const int n_open = __popcll(open_mask);
  if (p.dbg && vb.x == 0 && threadIdx.x == 0) p.dbg[49] = __builtin_amdgcn_s_memrealtime();
  const int T_head = T;
  const int found_head = found;
  unsigned long long T2_sum = 0ull;
  // rank of this lane's keyline among the open ones (the same for all lanes of the keyline)
  const int rank = __popcll(open_mask & ((1ull << (kl * kLPK)) - 1ull));
  const bool kl_open = ((open_mask >> (kl * kLPK)) & 1ull) != 0ull;
  for (int b0 = 0; b0 < n_open; b0 += kBatch) {
    wave_lds_sync();  // (the previous batch's tables are still being read until here)
    const bool mine = kl_open && rank >= b0 && rank < b0 + kBatch;  // all lanes of the batch's keylines
    const int jj = rank - b0;
    const int nb = min(kBatch, n_open - b0);
    if (mine && sub == 0) {
      W.s_a[jj] = make_float4(S.t_x, S.t_y, S.pi0x, S.pi0y);
      W.s_b[jj] = make_float4(S.dq_min, S.dq_max, __int_as_float(S.t_steps), 0.f);
      W.q_a[jj] = make_float4(S.norm_t, S.sigma2_t, gq.x, gq.y);
      W.q_gn[jj] = gnq;
      W.best[jj] = kDmcNone;
    }
    // no search is longer than the radius allows (t_steps <= search_range + pixel_uncertainty_match + 1, bounded at create):
    // one window with the default radius of 40
    const int tmax = cvtt_f32(max_radius + p.pixel_uncertainty_match) + 2;
    for (int step0 = kHeadSteps; step0 < tmax; step0 += kDmcWin) {
      // the chains continue through this window (tn, tp stand at step0): with two or more lanes per keyline lane 0 walks tn and
      // lane 1 tp (x - 1.0f and x + (-1.0f) are the same IEEE operation), a lone lane walks both
      const int wlen = min(kDmcWin, tmax - step0);
      // (always a whole window of kDmcWin steps, unrolled: one add and one LDS store with an immediate offset per step; what lies
      // beyond a keyline's t_steps is never read)
      if (kLPK >= 2) {
        if (mine && sub < 2) {
          float x = sub == 1 ? tp : tn;
          const float d = sub == 1 ? 1.0f : -1.0f;
          float* dst = &W.seq[jj][sub][0];
#pragma unroll
          for (int s = 0; s < kDmcWin; ++s) {
            dst[s] = x;
            x += d;
          }
          if (sub == 0) tn = x;
          else tp = x;
        }
      } else {
        if (mine) {  // (the owners of later batches keep their chains at step kHeadSteps)
#pragma unroll
          for (int s = 0; s < kDmcWin; ++s) {
            W.seq[jj][0][s] = tn;
            W.seq[jj][1][s] = tp;
            tp += 1.0f;
            tn -= 1.0f;
          }
        }
      }
      wave_lds_sync();
      RH_DMC_TICK(3)  // chains walked
      // probe slots of the batch, dealt to all 64 lanes: q -> (open keyline j, side, step); the first kDmcWin of a keyline's
      // 2 * kDmcWin slots walk tn, the others tp, so that neighbouring lanes probe neighbouring pixels of one line
      constexpr int kProbeIters = (kBatch * 2 * kDmcWin + 63) / 64;
      int pc[kProbeIters];
      unsigned pm[kProbeIters];
      constexpr int per = 2 * kDmcWin;
      const int nslots = nb * per;
      int pj = 0, ps = lane;  // slot of this lane in round 0: (open keyline, position); +64 slots per round
      unsigned pok = 0u;      // which of this lane's probes count
      // Rounds in chunks of kChunk behind ONE uniform test: inside a chunk nothing branches and no loaded value is looked at (a
      // select on it would make every round wait for its own load), so the LDS reads and mask loads of a chunk's rounds overlap.
      constexpr int kChunk = 5;
      static_assert(kProbeIters % kChunk == 0, "probe rounds per chunk");
#pragma unroll
      for (int c = 0; c < kProbeIters / kChunk; ++c) {
        if (c * kChunk * 64 < nslots) {  // (uniform)
#pragma unroll
          for (int u = 0; u < kChunk; ++u) {
            const int i = c * kChunk + u;
            const int j = min(pj, nb - 1);
            const int side = ps >= kDmcWin ? 1 : 0, so = ps - side * kDmcWin;
            const int step = step0 + so;
            const float4 sb = W.s_b[j];
            const float4 sa = W.s_a[j];
            const float t = W.seq[j][side][so];
            const int bj = W.best[j];
            bool ok = pj < nb && so < wlen && step < __float_as_int(sb.z) && bj == kDmcNone;
            ok = ok && (side ? !(t > sb.y) : !(t < sb.x));
            const int row = cvtt_f32(roundf(sa.y * t + sa.w));
            const int col = cvtt_f32(roundf(sa.x * t + sa.z));
            ok = ok && (unsigned)row < (unsigned)p.rows && (unsigned)col < (unsigned)p.cols;
            pc[i] = om.mask[ok ? row * p.cols + col : 0];
            pok |= ok ? (1u << i) : 0u;
            pm[i] = ((unsigned)j << 10) | (unsigned)(2 * step + side);
            ps += 64;  // (64 < per: at most one wrap)
            if (ps >= per) {
              ps -= per;
              ++pj;
            }
          }
        } else {
#pragma unroll
          for (int u = 0; u < kChunk; ++u) {
            pc[c * kChunk + u] = -1;
            pm[c * kChunk + u] = 0u;
          }
        }
      }
      RH_DMC_TICK(4)  // long-search probes issued
      int c2 = 0;
#pragma unroll
      for (int i = 0; i < kProbeIters; ++i) {
        if (!((pok >> i) & 1u)) pc[i] = -1;
        c2 += (pc[i] >= 0) ? 1 : 0;
      }
      int T2 = 0;
      int pos2 = wave_excl_scan_i(c2, lane, &T2);
      RH_DMC_TICK(5)  // mask loads back, scan
#pragma unroll
      for (int i = 0; i < kProbeIters; ++i) {
        if (pc[i] >= 0) {
          W.list[pos2] = ((unsigned)pc[i] << 16) | pm[i];
          ++pos2;
        }
      }
      wave_lds_sync();
      T2_sum += (unsigned long long)T2;
      dmc_test_list<true>(p, om, W, T2, lane, A.rot, A.R0, step0);
    }
    if (mine && sub == 0 && W.best[jj] != kDmcNone) found = dmc_commit(nm, om, W, jj, idx, &kf);
    RH_DMC_TICK(6)  // long-search candidates tested, commits issued
  }
  if (stats) {  // REBVIO_HIP_DM_STATS: one record per wave (plain stores; the host aggregates them in rebvio_hip_flush)
    const int c_live = __popcll(__ballot(live && sub == 0)), c_long = __popcll(__ballot(live && sub == 0 && S.t_steps > kHeadSteps));
    const int c_f1 = __popcll(__ballot(found_head >= 0));
    if (lane == 0) {
      unsigned long long* r = p.dm_stats + 16 * (size_t)(1 + vb.x * kWaves + wid);
#pragma unroll
      for (int i = 0; i < 7; ++i) r[i] = tk[i];
      r[8] = (unsigned long long)c_live; r[9] = (unsigned long long)c_long; r[10] = (unsigned long long)c_f1;
      r[11] = (unsigned long long)n_open; r[12] = (unsigned long long)T_head; r[13] = T2_sum;
      if (vb.x == 0 && wid == 0) p.dm_stats[0] = (unsigned long long)(gridDim.x * kWaves);
    }
  }
#undef RH_DMC_TICK
  // counters: wave -> workgroup (LDS) -> one global atomic per workgroup and counter (see k_directed_match)
  const int c_found = __popcll(__ballot(found >= 0));
  const int c_kf = __popcll(__ballot(kf != 0));
  if (kWaves == 1) {
    if (lane == 0) {
      if (c_found) atomicAdd(&nm.st->dm_matches, c_found);
      if (c_kf) atomicAdd(&nm.st->dm_kf, c_kf);
      if (n_open) atomicAdd(&nm.st->dm_queued, n_open);
    }
  } else {
    __shared__ int w_cnt[kWaves][3];
    if (lane == 0) {
      w_cnt[wid][0] = c_found;
      w_cnt[wid][1] = c_kf;
      w_cnt[wid][2] = n_open;
    }
    __syncthreads();
    if (threadIdx.x < 3) {
      int t = 0;
#pragma unroll
      for (int w = 0; w < kWaves; ++w) t += w_cnt[w][threadIdx.x];
      int* dst = threadIdx.x == 0 ? &nm.st->dm_matches : (threadIdx.x == 1 ? &nm.st->dm_kf : &nm.st->dm_queued);
      if (t) atomicAdd(dst, t);
    }
  }
}

template <int kThreads, int kLPK>
__global__ __launch_bounds__(kThreads) void k_directed_match_c(KParams p, MapDev nm, MapDev om, Vec3 vel_, Mat3 Rvel_, Mat3 Rback_,
                                                               float max_radius, int rot_, Mat3 R0_, const GlueDev* __restrict__ gd,
                                                               const GlueStage* __restrict__ stage) {
  directed_match_c_body<kThreads, kLPK>(p, nm, om, vel_, Rvel_, Rback_, max_radius, rot_, R0_, gd, stage);
}
template <int kThreads, int kLPK>
__global__ __launch_bounds__(kThreads) void k_directed_match_c_b(KParams p, const LaneStatic* __restrict__ ls, const MapDev* __restrict__ maptab,
                                                                 LaneDynB dyn, float max_radius) {
  const LaneStatic& L = ls[blockIdx.z];
  const LaneDyn d = dyn.v[blockIdx.z];
  const Vec3 z3{};
  const Mat3 z9{};
  directed_match_c_body<kThreads, kLPK>(p, global_map(lane_map(maptab, blockIdx.z, d.nm, d.nm_swap)),
                                        global_map(lane_map(maptab, blockIdx.z, d.om, d.om_swap)), z3, z9, z9, max_radius, 1, z9,
                                        gptr(L.glue_dev) + d.slot, gptr(L.glue_stage) + d.slot);
}

// ---- EdgeMap::searchMatch as a public single-keyline call (edge_map.hpp:93-94, edge_map.cpp:101-184) ----------------
// One lane walks the reference's alternating probe sequence with the same set-up / acceptance code as the directedMatch
// kernels (vel / Rvel as given: directedMatch rotates them by Rback before it calls searchMatch, edge_map.cpp:193-194).
__global__ void k_search_match_one(KParams p, MapDev om, float2 pi, float2 rsq, float2 gq, float gnq, Vec3 vel, Mat3 Rvel,
                                   Mat3 Rback, float max_radius, int* __restrict__ out) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  const SearchSetup S = search_setup(p, pi, rsq, gq, gnq, vel, Rvel, Rback, max_radius);
  Mat3 I{};
  int found = -1;
  float tn = S.dq_rho, tp = S.dq_rho + 1.0f;
  for (int t_i = 0; t_i < S.t_steps && found < 0; ++t_i, tp += 1.0f, tn -= 1.0f) {
    for (int i_idx = 0; i_idx < 2 && found < 0; ++i_idx) {
      const float t = i_idx ? tp : tn;
      if (i_idx ? (t > S.dq_max) : (t < S.dq_min)) continue;
      const int row = cvtt_f32(roundf(S.t_y * t + S.pi0y));
      const int col = cvtt_f32(roundf(S.t_x * t + S.pi0x));
      if ((unsigned)row >= (unsigned)p.rows || (unsigned)col >= (unsigned)p.cols) continue;
      const int cand = om.mask[(size_t)row * p.cols + col];
      if (cand < 0) continue;
      const OldKl ck = load_old(om, cand, 0, I, p.fm);
      if (search_accept(p, S, t, ck.g, ck.gn, ck.rs, gq, gnq)) found = cand;
    }
  }
  *out = found;
}

// ---- EdgeMap::regularize1Iter (edge_map.cpp:220-259): Jacobi step, results staged in rs_tmp ----------------------
__global__ __launch_bounds__(256) void k_regularize(KParams p, MapDev m, int gate_min_matches) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  const float2 rs = m.rs[idx];  // bound-free early loads
  const int in = m.id_next[idx], ip = m.id_prev[idx];
  const int n = m.st->n;
  if (gate_min_matches > 0 && m.st->dm_matches < gate_min_matches) return;
  int set = 0;
  if (idx < n) {
    float2 out = rs;
    if (in >= 0 && ip >= 0) {
      const float2 rn = m.rs[in], rp = m.rs[ip];
      if (!((rn.x - rp.x) * (rn.x - rp.x) > (rn.y * rn.y + rp.y * rp.y))) {
        const float2 gn = m.grad[in], gp = m.grad[ip];
        float alpha = (gn.x * gp.x + gn.y * gp.y) / (m.gnorm[in] * m.gnorm[ip]);
        if (!(alpha < p.regularization_threshold)) {
          alpha = (float)((double)(alpha - p.regularization_threshold) / (1.0 - (double)p.regularization_threshold));
          alpha = (float)((double)alpha / ((double)(fabsf(rn.x - rp.x) / (rn.y + rp.y)) + 1.0));
          const float wr = (float)(1.0 / (double)(rs.y * rs.y));
          const float wrn = alpha / (rn.y * rn.y);
          const float wrp = alpha / (rp.y * rp.y);
          out.x = (rs.x * wr + rn.x * wrn + rp.x * wrp) / (wr + wrn + wrp);
          out.y = (rs.y * wr + rn.y * wrn + rp.y * wrp) / (wr + wrn + wrp);
          set = 1;
        }
      }
    }
    m.rs_tmp[idx] = out;
  }
  const int c = wave_sum_i(set);
  if ((threadIdx.x & 63) == 0 && c) atomicAdd(&m.st->reg_count, c);
}

// ---- Core::updateInverseDepth[ARLU] (core.cpp:417-456): scalar EKF per matched keyline ----------------------------
__global__ __launch_bounds__(256) void k_depth_ekf(KParams p, MapDev m, Vec3 vel, int use_tmp, int gate_min_matches) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  float2 rs = use_tmp ? m.rs_tmp[idx] : m.rs[idx];  // bound-free early loads
  const int mid = m.match_id[idx];
  const float2 q = m.pos_img[idx];
  const float2 q0 = m.mpos_img[idx];
  const float2 mg = m.mgrad[idx];
  const float mgn = m.mgnorm[idx];
  const int n = m.st->n;
  if (gate_min_matches > 0 && m.st->dm_matches < gate_min_matches) return;
  if (idx >= n) return;
  if (mid >= 0) {
    const float vx = vel.a[0], vy = vel.a[1], vz = vel.a[2];
    float v_rho = rs.y * rs.y;
    const float ux = mg.x / mgn;
    const float uy = mg.y / mgn;
    const float Y = ux * (q.x - q0.x) + uy * (q.y - q0.y);
    const float H = ux * (vx * p.fm - vz * q0.x) + uy * (vy * p.fm - vz * q0.y);
    const float rho_p = (float)(1.0 / (1.0 / (double)rs.x + (double)vz));
    float F = (float)(1.0 / (1.0 + (double)(rs.x * vz)));
    F *= F;
    const float p_p = F * v_rho * F + p.reshape_q_abs * p.reshape_q_abs;
    const float e = Y - H * rho_p;
    const float S = H * p_p * H + p.pixel_uncertainty * p.pixel_uncertainty;
    const float K = (float)((double)(p_p * H) * (1.0 / (double)S));
    float rho = rho_p + K * e;
    v_rho = (float)((1.0 - (double)(K * H)) * (double)p_p);
    float sig = sqrtf(v_rho);
    if (rho < kRhoMin) {
      sig += kRhoMin - rho;
      rho = kRhoMin;
    } else if (rho > kRhoMax) {
      rho = kRhoMax;
    } else if (isnan(rho) || isnan(sig) || isinf(rho) || isinf(sig)) {
      rho = kRhoInit;
      sig = kRhoMax;
    }
    rs = make_float2(rho, sig);
  }
  m.rs[idx] = rs;
}

// ---- regularize1Iter + updateInverseDepth fused (edge_map.cpp:220-259, core.cpp:417-456) ------------------------
// Jacobi semantics need the neighbours' OLD depths: read rs, write rs_tmp; the host then swaps the two pointers of
// this map. Optionally (streaming driver) the keyline is also put through the NEXT pair's first rotateKeylines
// (rebvio.cpp:165) and binned for its estimateQuantile, which removes that pair's k_rotate launch.
// gd != null: vel / next rotation come from *gd; gd->nan_v (rebvio.cpp:236) leaves rho untouched like the match gate.
__device__ __forceinline__ void regularize_ekf_body(KParams p, MapDev m, Vec3 vel_, int gate_min_matches,
                                                        int next_rot_, Mat3 Rnext_,
                                                        int* __restrict__ hist, int hist_bins, const GlueDev* __restrict__ gd) {
  const uint2 vb = xcd_band_block();  // which keylines / tiles this workgroup takes: contiguous bands per XCD (common.hpp)
  __shared__ int sh[128];
  if (p.dbg && vb.x == 0 && threadIdx.x == 0) p.dbg[50] = __builtin_amdgcn_s_memrealtime();
  const int idx = vb.x * 256 + threadIdx.x;
  const float2 rs = m.rs[idx];  // bound-free early loads, issued before the parameter block is read
  const int in = m.id_next[idx], ip = m.id_prev[idx];
  const int mid = m.match_id[idx];
  float2 q = m.pos_img[idx];
  const float2 q0 = m.mpos_img[idx];
  const float2 mg = m.mgrad[idx];
  const float mgn = m.mgnorm[idx];
  float2 g = m.grad[idx];
  const int n = m.st->n;
  const int dm_matches = m.st->dm_matches;
  Vec3 vel = vel_;
  Mat3 Rnext = Rnext_;
  int next_rot = next_rot_;
  bool skip_pair = false;
  if (gd) {
#pragma unroll
    for (int i = 0; i < 3; ++i) vel.a[i] = gd->V[i];
#pragma unroll
    for (int i = 0; i < 9; ++i) Rnext.a[i] = gd->RT_next[i];
    next_rot = gd->has_next;
    skip_pair = gd->nan_v != 0;
  }
  // both neighbours' depth, gradient and norm in ONE gather round trip (index 0 stands in where there is no neighbour pair;
  // the values are only used under the same conditions as before)
  const bool has_nb = idx < n && in >= 0 && ip >= 0;  // (rows past n hold no valid ids)
  const int in_s = has_nb ? in : 0, ip_s = has_nb ? ip : 0;
  const float2 rn = m.rs[in_s], rp = m.rs[ip_s];
  const float2 gn = m.grad[in_s], gp = m.grad[ip_s];
  const float gnn = m.gnorm[in_s], gnp = m.gnorm[ip_s];
  const bool gated = skip_pair || (gate_min_matches > 0 && dm_matches < gate_min_matches);
  if (next_rot) {
    if (threadIdx.x < 128) sh[threadIdx.x] = 0;
    __syncthreads();
  }
  int set = 0;
  if (idx < n) {
    float2 out = rs;
    if (!gated) {
      if (has_nb) {
        if (!((rn.x - rp.x) * (rn.x - rp.x) > (rn.y * rn.y + rp.y * rp.y))) {
          float alpha = (gn.x * gp.x + gn.y * gp.y) / (gnn * gnp);
          if (!(alpha < p.regularization_threshold)) {
            alpha = (float)((double)(alpha - p.regularization_threshold) / (1.0 - (double)p.regularization_threshold));
            alpha = (float)((double)alpha / ((double)(fabsf(rn.x - rp.x) / (rn.y + rp.y)) + 1.0));
            const float wr = (float)(1.0 / (double)(rs.y * rs.y));
            const float wrn = alpha / (rn.y * rn.y);
            const float wrp = alpha / (rp.y * rp.y);
            out.x = (rs.x * wr + rn.x * wrn + rp.x * wrp) / (wr + wrn + wrp);
            out.y = (rs.y * wr + rn.y * wrn + rp.y * wrp) / (wr + wrn + wrp);
            set = 1;
          }
        }
      }
      if (mid >= 0) {
        const float vx = vel.a[0], vy = vel.a[1], vz = vel.a[2];
        float v_rho = out.y * out.y;
        const float ux = mg.x / mgn;
        const float uy = mg.y / mgn;
        const float Y = ux * (q.x - q0.x) + uy * (q.y - q0.y);
        const float H = ux * (vx * p.fm - vz * q0.x) + uy * (vy * p.fm - vz * q0.y);
        const float rho_p = (float)(1.0 / (1.0 / (double)out.x + (double)vz));
        float F = (float)(1.0 / (1.0 + (double)(out.x * vz)));
        F *= F;
        const float p_p = F * v_rho * F + p.reshape_q_abs * p.reshape_q_abs;
        const float e = Y - H * rho_p;
        const float S = H * p_p * H + p.pixel_uncertainty * p.pixel_uncertainty;
        const float K = (float)((double)(p_p * H) * (1.0 / (double)S));
        float rho = rho_p + K * e;
        v_rho = (float)((1.0 - (double)(K * H)) * (double)p_p);
        float sig = sqrtf(v_rho);
        if (rho < kRhoMin) {
          sig += kRhoMin - rho;
          rho = kRhoMin;
        } else if (rho > kRhoMax) {
          rho = kRhoMax;
        } else if (isnan(rho) || isnan(sig) || isinf(rho) || isinf(sig)) {
          rho = kRhoInit;
          sig = kRhoMax;
        }
        out = make_float2(rho, sig);
      }
    }
    if (next_rot) {
      rotate_one(Rnext, p.fm, q, out, g);
      m.pos_img[idx] = q;
      m.grad_tmp[idx] = g;  // neighbours still read the un-rotated m.grad in this kernel: the caller swaps the pointers
      m.residual[idx] = 0.f;
      int i = cvtt_f32(hist_bins * (out.y - kRhoMin) / (kRhoMax - kRhoMin));
      i = (i > hist_bins - 1) ? (hist_bins - 1) : i;
      i = (i < 0) ? 0 : i;
      atomicAdd(&sh[i], 1);
    }
    m.rs_tmp[idx] = out;
  }
  // regularised-keyline count: one global atomic per workgroup (see k_directed_match)
  __shared__ int b_reg;
  if (threadIdx.x == 0) b_reg = 0;
  __syncthreads();
  const int c = wave_sum_i(set);
  if ((threadIdx.x & 63) == 0 && c) atomicAdd(&b_reg, c);
  __syncthreads();
  if (threadIdx.x == 0 && b_reg) atomicAdd(&m.st->reg_count, b_reg);
  if (next_rot) {
    if ((int)threadIdx.x < hist_bins && sh[threadIdx.x]) atomicAdd(&hist[threadIdx.x], sh[threadIdx.x]);
  }
}

__global__ __launch_bounds__(256) void k_regularize_ekf(KParams p, MapDev m, Vec3 vel_, int gate_min_matches,
                                                        int next_rot_, Mat3 Rnext_,
                                                        int* __restrict__ hist, int hist_bins, const GlueDev* __restrict__ gd) {
  regularize_ekf_body(p, m, vel_, gate_min_matches, next_rot_, Rnext_, hist, hist_bins, gd);
}
__global__ __launch_bounds__(256) void k_regularize_ekf_b(KParams p, const LaneStatic* __restrict__ ls, const MapDev* __restrict__ maptab,
                                                          LaneDynB dyn, int gate_min_matches) {
  const LaneStatic& L = ls[blockIdx.z];
  const LaneDyn d = dyn.v[blockIdx.z];
  const Vec3 z3{};
  const Mat3 z9{};
  regularize_ekf_body(p, global_map(lane_map(maptab, blockIdx.z, d.nm, d.nm_swap)), z3, gate_min_matches, 0, z9, gptr(L.hist), p.quantile_num_bins,
                      gptr(L.glue_dev) + d.slot);
}

// ---- launchers ------------------------------------------------------------------------------------------------------
static Mat3 mat3(const float* r) {
  Mat3 m;
  for (int i = 0; i < 9; ++i) m.a[i] = r[i];
  return m;
}
static Vec3 vec3(const float* r) {
  Vec3 v;
  for (int i = 0; i < 3; ++i) v.a[i] = r ? r[i] : 0.f;
  return v;
}

void launch_rotate(hipStream_t s, const KParams& p, const MapDev& m, const float R[9], int* hist, int zero_dm) {
  RH_LAUNCH(k_rotate, dim3(div_up(p.kmax, 256)), dim3(256), 0, s, p, m, mat3(R), hist, p.quantile_num_bins, zero_dm);
}

void launch_quantile(hipStream_t s, const KParams& p, const MapDev& m, int* hist, float pct, int bins, float* out_dev) {
  (void)hipMemsetAsync(hist, 0, sizeof(int) * 128, s);
  RH_LAUNCH(k_sigma_hist, dim3(div_up(p.kmax, 256)), dim3(256), 0, s, m, hist, bins);
  RH_LAUNCH(k_quantile, dim3(1), dim3(64), 0, s, m, (const int*)hist, bins, pct, out_dev);
}

void launch_try_vel(hipStream_t s, const KParams& p, const MapDev& oldm, const MapDev& newm, int mode_lm, int call,
                    int last, LmState* st_in, LmState* st_out, const float* part_prev, float* part_out, const int* hist,
                    int frame_count) {
  RH_LAUNCH(k_try_vel, dim3(div_up(p.kmax, 256)), dim3(256), 0, s, p, oldm, newm, mode_lm, call, last,
                     (const LmState*)st_in, st_out, part_prev, part_out, hist, (unsigned)frame_count);
}

// Speculative form (k_lm_chain_spec; do_ext == 2 / spec != 0 allows it, the context's REBVIO_HIP_LM=seq does not). Needs at
// least two speculative evaluations, at most kSpecMax, and the staged record sets in the default 64 KB of LDS with the static part.
// The LDS bound also keeps it to maps of up to ~40k keylines: with the attribute raised it ran at 64k keylines (1280x960,
// 125 workgroups) and was slower than the sequential kernel there (5.3k vs 6.5k frames/s) - four evaluations' arithmetic
// per pass on every workgroup and a 125-hop neighbour chain outweigh the three exchange rounds saved.
static size_t lm_spec_shm(int kmax, int calls) {
  const int groups = std::min(div_up(kmax, 256), kMaxRecBlocks);
  return (size_t)(calls - 2) * groups * kPartStride * sizeof(float);
}
static bool lm_spec_usable(int kmax, int calls, int kf = 2) {
  return calls - kf >= 2 && calls - 2 <= kSpecMax && calls <= kMaxLmCalls && lm_spec_shm(kmax, calls) <= 40 * 1024;
}

void launch_lm_chain(hipStream_t s, const KParams& p, const MapDev& oldm, const MapDev& newm, int calls, int do_ext, LmState* st_in,
                     LmState* st_out, unsigned long long* xch, unsigned tag_base, int* bar_err, const int* hist, float* xrv_part,
                     PairSlot* slot, int* hist_to_zero, unsigned long long* stamps, int threads, const GlueArgs& ga) {
  const dim3 grid((p.kmax + threads - 1) / threads);
  // do_ext: 1 = k_lm_chain; 2 / 3 = the speculative kernel with its first speculative evaluation at index 2 / 3 (lm_chain_spec_body)
  if (do_ext >= 2 && threads <= 512 && lm_spec_usable(p.kmax, calls, do_ext)) {
    const int kf = do_ext;
    if (threads == 256)
      RH_LAUNCH(k_lm_chain_spec<256>, grid, dim3(256), lm_spec_shm(p.kmax, calls), s, p, oldm, newm, calls, (const LmState*)st_in, st_out,
                xch, tag_base, bar_err, hist, xrv_part, slot, hist_to_zero, stamps, ga, kf);
    else
      RH_LAUNCH(k_lm_chain_spec<512>, grid, dim3(512), lm_spec_shm(p.kmax, calls), s, p, oldm, newm, calls, (const LmState*)st_in, st_out,
                xch, tag_base, bar_err, hist, xrv_part, slot, hist_to_zero, stamps, ga, kf);
    return;
  }
  if (do_ext >= 2) do_ext = 2;  // (k_lm_chain reads do_ext as "with forwardMatch / extRotVel")
  switch (threads) {
    case 256:
      RH_LAUNCH(k_lm_chain<256>, grid, dim3(256), 0, s, p, oldm, newm, calls, do_ext, (const LmState*)st_in, st_out, xch, tag_base,
                bar_err, hist, 0u, xrv_part, slot, hist_to_zero, stamps, ga);
      break;
    case 512:
      RH_LAUNCH(k_lm_chain<512>, grid, dim3(512), 0, s, p, oldm, newm, calls, do_ext, (const LmState*)st_in, st_out, xch, tag_base,
                bar_err, hist, 0u, xrv_part, slot, hist_to_zero, stamps, ga);
      break;
    default:
      RH_LAUNCH(k_lm_chain<1024>, grid, dim3(1024), 0, s, p, oldm, newm, calls, do_ext, (const LmState*)st_in, st_out, xch, tag_base,
                bar_err, hist, 0u, xrv_part, slot, hist_to_zero, stamps, ga);
  }
}

void launch_lm_final(hipStream_t s, const MapDev& oldm, int calls, LmState* st_in, LmState* st_out, const float* part_prev) {
  RH_LAUNCH(k_lm_final, dim3(1), dim3(256), 0, s, oldm, calls, (const LmState*)st_in, st_out, part_prev);
}

void launch_forward_keys(hipStream_t s, const KParams& p, const MapDev& oldm, const MapDev& newm) {
  RH_LAUNCH(k_forward_keys, dim3(div_up(p.kmax, 256)), dim3(256), 0, s, oldm, newm);
}

void launch_ext_rot_vel(hipStream_t s, const KParams& p, const MapDev& oldm, const MapDev& newm, int do_forward,
                        int do_lm_final, int calls, LmState* st_in, LmState* st_out, const float* part_prev,
                        float* xrv_part, const float* vel_manual, PairSlot* slot, int* hist_to_zero, unsigned seq) {
  RH_LAUNCH(k_ext_rot_vel, dim3(div_up(p.kmax, 256)), dim3(256), 0, s, p, oldm, newm, do_forward, do_lm_final,
                     calls, (const LmState*)st_in, st_out, part_prev, xrv_part, vec3(vel_manual), slot, hist_to_zero, seq);
}


// Form of the directedMatch launch, k_directed_match_c<threads, lanes per keyline> (REBVIO_HIP_DM_HEAD / REBVIO_HIP_BATCH_DM_HEAD, read
// at create; 0 = chosen here): 1 "compact8" <512, 8>, 2 "compact4" <256, 4>, 3 "compact1" <64, 1>.
// One stream of up to 32 768 keylines takes <512, 8>: eight keylines per wave, 1 875 short waves on an idle chip (MI355X, 640x480 /
// 15 k keylines: 15.6 k frames/s against 15.2 k for <256, 4>, 12.1 k for <64, 1>, 13.7 k for the two-launch form of round 3).
// Larger maps and batches of four lanes or more, where the chip is full, take the form with the fewest instructions, <64, 1>
// (1280x960 / 57 k keylines: 7.07 k frames/s against 6.26 k for <256, 4> and 6.69 k for round 3's form).
static int dm_form(int kmax, int head_form, int lanes) {
  if (head_form >= 1 && head_form <= 3) return head_form;
  return (kmax <= 32768 && lanes < 4) ? 1 : 3;
}
#define RH_DMC_DISPATCH(KERNEL, form, kmax, zdim, stream, ...)                                                                                 \
  do {                                                                                                                                         \
    switch (form) {                                                                                                                            \
      case 1: RH_LAUNCH_NAMED(#KERNEL "<512,8>", (KERNEL<512, 8>), dim3(div_up(kmax, 64), 1, zdim), dim3(512), 0, stream, __VA_ARGS__); break; \
      case 2: RH_LAUNCH_NAMED(#KERNEL "<256,4>", (KERNEL<256, 4>), dim3(div_up(kmax, 64), 1, zdim), dim3(256), 0, stream, __VA_ARGS__); break; \
      default: RH_LAUNCH_NAMED(#KERNEL "<64,1>", (KERNEL<64, 1>), dim3(div_up(kmax, 64), 1, zdim), dim3(64), 0, stream, __VA_ARGS__); break;   \
    }                                                                                                                                          \
  } while (0)

void launch_directed_match(hipStream_t s, const KParams& p, const MapDev& newm, const MapDev& oldm, const float vel[3],
                           const float Rvel[9], const float Rback[9], float max_radius, const float* R0_on_the_fly, int head_form) {
  const int form = dm_form(p.kmax, head_form, 1);
  const int rot = R0_on_the_fly ? 1 : 0;
  const float I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  const Mat3 R0 = mat3(R0_on_the_fly ? R0_on_the_fly : I);
  RH_DMC_DISPATCH(k_directed_match_c, form, p.kmax, 1, s, p, newm, oldm, vec3(vel), mat3(Rvel), mat3(Rback), max_radius, rot, R0,
                  (const GlueDev*)nullptr, (const GlueStage*)nullptr);
}

// the same launch with the second half's inputs read from *gd at run time: the record the pair's LM kernel left (device glue,
// glue_dev.hpp)
void launch_directed_match_dev(hipStream_t s, const KParams& p, const MapDev& newm, const MapDev& oldm, const GlueDev* gd,
                               const GlueStage* stage, float max_radius, int head_form) {
  const int form = dm_form(p.kmax, head_form, 1);
  const float I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, z[3] = {0, 0, 0};
  RH_DMC_DISPATCH(k_directed_match_c, form, p.kmax, 1, s, p, newm, oldm, vec3(z), mat3(I), mat3(I), max_radius, 1, mat3(I), gd, stage);
}

void launch_search_match_one(hipStream_t s, const KParams& p, const MapDev& searched, const rebvio_hip_keyline& q, const float vel[3],
                             const float Rvel[9], const float Rback[9], float max_radius, int* out_dev) {
  RH_LAUNCH(k_search_match_one, dim3(1), dim3(64), 0, s, p, searched, make_float2(q.pos_img[0], q.pos_img[1]),
            make_float2(q.rho, q.sigma_rho), make_float2(q.gradient[0], q.gradient[1]), q.gradient_norm, vec3(vel), mat3(Rvel),
            mat3(Rback), max_radius, out_dev);
}

void launch_regularize(hipStream_t s, const KParams& p, const MapDev& m, int gate) {
  RH_LAUNCH(k_regularize, dim3(div_up(p.kmax, 256)), dim3(256), 0, s, p, m, gate);
}

void launch_regularize_ekf(hipStream_t s, const KParams& p, const MapDev& m, const float vel[3], int gate, const float* Rnext, int* hist) {
  const float I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  RH_LAUNCH(k_regularize_ekf, dim3(div_up(p.kmax, 256)), dim3(256), 0, s, p, m, vec3(vel), gate, Rnext ? 1 : 0,
            mat3(Rnext ? Rnext : I), hist, p.quantile_num_bins, (const GlueDev*)nullptr);
}

void launch_regularize_ekf_dev(hipStream_t s, const KParams& p, const MapDev& m, const GlueDev* g_dev, int gate, int* hist) {
  const float I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, z[3] = {0, 0, 0};
  RH_LAUNCH(k_regularize_ekf, dim3(div_up(p.kmax, 256)), dim3(256), 0, s, p, m, vec3(z), gate, 0, mat3(I), hist,
                 p.quantile_num_bins, g_dev);
}

void launch_lm_chain_b(hipStream_t s, const KParams& p, int lanes, int lanes_per_launch, const LaneStatic* ls, const MapDev* maptab,
                       const LaneDynB& dyn, int calls, int spec, const GlueParams& gp) {
  // 512-thread workgroups (the single-stream default). The workgroups of a lane wait for each other's records, and HIP promises
  // nothing about the order in which a grid's workgroups become resident (MI355X_MICROARCH.md, contract [G]): a launch carries
  // only as many lanes as fit the device TOGETHER (lm_chain_b_max_lanes: 8 lanes of 16 k keylines at one workgroup per CU), a
  // wider batch takes several launches back to back.
  const int slow_poll = lanes >= 3 ? 1 : 0;
  const int per = std::max(1, std::min(lanes, lanes_per_launch));
  const bool use_spec = spec >= 2 && lm_spec_usable(p.kmax, calls, spec);  // spec: 0 / 1 = k_lm_chain_b, 2 / 3 = first speculative evaluation
  for (int l0 = 0; l0 < lanes; l0 += per) {
    const dim3 grid((p.kmax + 511) / 512, 1, (unsigned)std::min(per, lanes - l0));
    if (use_spec)
      RH_LAUNCH(k_lm_chain_spec_b<512>, grid, dim3(512), lm_spec_shm(p.kmax, calls), s, p, ls, maptab, dyn, calls, slow_poll, gp, l0, spec);
    else
      RH_LAUNCH(k_lm_chain_b<512>, grid, dim3(512), 0, s, p, ls, maptab, dyn, calls, slow_poll, gp, l0);
  }
}

// How many lanes' workgroups of the batched persistent LM kernel the device holds at once: all workgroups of a launch poll
// their lane's records, so all of them have to be resident together whatever order they are dispatched in. One block per CU is
// taken off the occupancy query's answer where it admits several (MI355X_MICROARCH.md: the query can read one block per CU
// high); kernels of the other streams only delay residency, they do not depend on this one. 0: not even one lane fits.
int lm_chain_b_capacity_wgs(int device, int kmax, int calls) {
  int cus = 0;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || cus <= 0) return 0;
  int nb = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_lm_chain_b<512>, 512, 0) != hipSuccess) nb = 1;
  if (lm_spec_usable(kmax, calls)) {
    int nbs = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nbs, k_lm_chain_spec_b<512>, 512, lm_spec_shm(kmax, calls)) != hipSuccess) nbs = 1;
    nb = std::min(nb, nbs);
  }
  nb = nb > 1 ? nb - 1 : 1;
  return nb * cus;
}
int lm_chain_b_max_lanes(int device, int kmax, int calls) {
  const int cap = lm_chain_b_capacity_wgs(device, kmax, calls);
  if (cap <= 0) return 1;
  const int per_lane = (kmax + 511) / 512;
  return std::min(kMaxLanes, cap / per_lane);
}

// second half of a batched step (every lane's inputs come from the record its LM kernel's glue left)
void launch_b_chain_b(hipStream_t s, const KParams& p, int lanes, const LaneStatic* ls, const MapDev* maptab, const LaneDynB& dyn,
                      float max_radius, int gate, int head_form) {
  const unsigned z = (unsigned)lanes;
  const int form = dm_form(p.kmax, head_form, lanes);
  RH_DMC_DISPATCH(k_directed_match_c_b, form, p.kmax, z, s, p, ls, maptab, dyn, max_radius);
  RH_LAUNCH(k_regularize_ekf_b, dim3(div_up(p.kmax, 256), 1, z), dim3(256), 0, s, p, ls, maptab, dyn, gate);
}

void launch_depth_ekf(hipStream_t s, const KParams& p, const MapDev& m, const float vel[3], int use_tmp, int gate) {
  RH_LAUNCH(k_depth_ekf, dim3(div_up(p.kmax, 256)), dim3(256), 0, s, p, m, vec3(vel), use_tmp, gate);
}

}  // namespace rh
