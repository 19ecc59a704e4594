// Device form of the pair glue (glue.hpp: pair_glue_core), run by workgroup 0 of the persistent LM kernel behind its last
// phase: the same statements per output element as the host form, spread over three waves and a few lanes each so that the
// ~2500 dependent scalar operations of the host version become a chain of ~600:
//   wave 0  the 6x6 solve of extRotVel in double (core.cpp:244-248), one row of L per lane; afterwards the SO3 correction, the
//           covariance block, the second half's inputs and the host's record
//   wave 1  gyroBiasCorrection's 3x3 inverses (core.cpp:264-284; nine lanes, one cofactor each, independent ones side by
//           side) and inverse(Wxb), one column per lane (core.cpp:276); afterwards the filter state and the next prior
//   wave 2  W_Xgv after the correction and its inverse (rebvio.cpp:201-202), one column per lane
//   wave 3  invert(JtJ) of minimizeVel (core.cpp:186) for the host's record
// Matrices live in LDS (lane-dependent element indices are LDS addresses, never register-array indices: those would become
// scratch memory); each lane's own chain runs on registers with static indices. Bit-identical to pair_glue_core on the host:
// tests/test_parity_gpu.py::test_device_glue_equals_host_glue (streams) and ::test_glue_probe_random_inputs (random and
// ill-conditioned systems through rebvio_hip_test_glue).
#pragma once

#include "glue.hpp"

namespace rh {

struct GlueLds {
  float W[36];       // W_Xv: extRotVel JtJ
  float JtF[6];
  float Xv[6];
  // 3x3 matrices of gyroBiasCorrection, one set per wave that needs them (waves do not wait for each other inside a phase):
  // [0] Wg  [1] invert3(W_Bg) (+ RGBias)  [2] W_Bg'  [3] Wg + W_Bg'  [4] iWgWb  [5] scratch  [6] upd  [7] (Wg iWgWb) W_Bg'  [8] RGyro  [9] W_Bg
  float m3[2][10][9];
  float A[2][36];    // 0: Wxb (core.cpp:271-272)   1: W_Xgv after the correction (core.cpp:283)
  float inv[2][36];
  float X1[2][6], X[2][6];
  float P_Vg[9];
  float JtJ9[9];
  double Ld[36];     // L of the 6x6 solve (rows written by their lanes, read by column in the back substitution)
  double ws[72];     // Jacobi fall-back of the 6x6 solve
  int ldlt_ok;
};

// wave-level hand-over through LDS: the LDS pipeline of a wave is in order, so only the compiler has to be kept from moving
// accesses across this point
__device__ __forceinline__ void glue_wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// value of lane `src` (a compile-time constant after unrolling) in every lane
__device__ __forceinline__ double glue_bcast(double v, int src) {
  const long long b = __double_as_longlong(v);
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(b & 0xFFFFFFFFll), src);
  const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)((unsigned long long)b >> 32), src);
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// TooN::determinant for N = 3 (Gaussian elimination with partial pivoting): hostmath.hpp det3, from registers
__device__ __forceinline__ float glue_det3(const float (&m)[9]) {
  float A[3][3] = {{m[0], m[1], m[2]}, {m[3], m[4], m[5]}, {m[6], m[7], m[8]}};
  float det = 1;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    // pivot row: the first of the rows i.. with the largest |A[r][i]| (strict >, as the reference's loop); rows swapped by value
    float mx = fabsf(A[i][i]);
    int arg = i;
#pragma unroll
    for (int ii = i + 1; ii < 3; ++ii)
      if (fabsf(A[ii][i]) > mx) {
        mx = fabsf(A[ii][i]);
        arg = ii;
      }
#pragma unroll
    for (int ii = i + 1; ii < 3; ++ii)
      if (arg == ii) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const float t = A[i][c];
          A[i][c] = A[ii][c];
          A[ii][c] = t;
        }
      }
    const float pivot = A[i][i];
    if (arg != i) det *= -1;
    det *= A[i][i];
    if (det == 0) return 0;
#pragma unroll
    for (int u = i + 1; u < 3; ++u) {
      const float factor = A[u][i] / pivot;
#pragma unroll
      for (int uu = i; uu < 3; ++uu) A[u][uu] = A[u][uu] - factor * A[i][uu];
    }
  }
  return det;
}

// types::invert (types/definitions.hpp:40-53), up to four matrices at once: lane 16 g + e (e < 9) produces element e of
// dst[g] = inverse(src[g]) for g < ngroups. Every lane of the wave calls it.
__device__ __forceinline__ void glue_invert3_lanes(const float* const (&src)[4], float* const (&dst)[4], int ngroups, int lane) {
  const int g = lane >> 4, e = lane & 15;
  const bool act = g < ngroups && e < 9;
  const float* sp = src[act ? g : 0];
  float m[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) m[i] = sp[i];
  const float d = glue_det3(m);
  float o;
  switch (e) {
    case 0: o = m[4] * m[8] - m[5] * m[7]; break;
    case 1: o = m[2] * m[7] - m[1] * m[8]; break;
    case 2: o = m[1] * m[5] - m[2] * m[4]; break;
    case 3: o = m[5] * m[6] - m[3] * m[8]; break;
    case 4: o = m[0] * m[8] - m[2] * m[6]; break;
    case 5: o = m[2] * m[3] - m[0] * m[5]; break;
    case 6: o = m[3] * m[7] - m[4] * m[6]; break;
    case 7: o = m[1] * m[6] - m[0] * m[7]; break;
    default: o = m[0] * m[4] - m[1] * m[3]; break;
  }
  glue_wave_sync();
  if (act) dst[g][e] = o / d;
  glue_wave_sync();
}
__device__ __forceinline__ void glue_invert3_one(const float* src, float* dst, int lane) {
  const float* const s4[4] = {src, src, src, src};
  float* const d4[4] = {dst, dst, dst, dst};
  glue_invert3_lanes(s4, d4, 1, lane);
}

// element e = 3 i + j of x * y for 3x3 matrices in LDS: dot product accumulated from 0 in index order (hostmath.hpp mul)
__device__ __forceinline__ float glue_mul3_elem(const float* x, const float* y, int e) {
  const int i = e / 3, j = e - 3 * i;
  float s = 0;
#pragma unroll
  for (int k = 0; k < 3; ++k) s += x[i * 3 + k] * y[k * 3 + j];
  return s;
}

// gyroBiasCorrection's 3x3 algebra (hostmath.hpp gyro_bias_correction) up to `depth`: 1 = Wg only, 2 = everything. Nine lanes
// of the calling wave (all 64 call it); M = this wave's matrix set.
__device__ __forceinline__ void glue_gyro3(float (&M)[10][9], const GlueState* st_in, float s_g, float s_b, int depth, int lane) {
  const int e = lane < 9 ? lane : 8;
  const int di = e / 3, dj = e - 3 * di;
  if (lane < 9) {
    M[8][lane] = (di == dj) ? s_g : 0.0f;  // RGyro = diag3(s_g)
    M[9][lane] = st_in->W_Bg[lane];
  }
  glue_wave_sync();
  {
    const float* const s4[4] = {M[8], M[9], M[8], M[8]};
    float* const d4[4] = {M[0], M[1], M[0], M[0]};
    glue_invert3_lanes(s4, d4, depth >= 2 ? 2 : 1, lane);  // Wg = invert3(Rg) | invert3(Wb), side by side
  }
  if (depth < 2) return;
  if (lane < 9) M[1][lane] = M[1][lane] + ((di == dj) ? s_b : 0.0f);  // add(invert3(Wb), Rb)
  glue_wave_sync();
  glue_invert3_one(M[1], M[2], lane);  // Wb = invert3(...)
  if (lane < 9) M[3][lane] = M[0][lane] + M[2][lane];  // add(Wg, Wb)
  glue_wave_sync();
  glue_invert3_one(M[3], M[4], lane);  // iWgWb
  // mul(iWgWb, Wg) on lanes 0..8, mul(Wg, iWgWb) on lanes 16..24
  {
    const int g = lane >> 4, e2 = (lane & 15) < 9 ? (lane & 15) : 8;
    const float v = g == 0 ? glue_mul3_elem(M[4], M[0], e2) : glue_mul3_elem(M[0], M[4], e2);
    glue_wave_sync();
    if ((lane & 15) < 9 && g == 0) M[5][e2] = ((di == dj) ? 1.0f : 0.0f) - v;  // sub(identity3(), mul(iWgWb, Wg))
    if ((lane & 15) < 9 && g == 1) M[7][e2] = v;                                // mul(Wg, iWgWb)
    glue_wave_sync();
    const float w = g == 0 ? glue_mul3_elem(M[0], M[5], e2) : glue_mul3_elem(M[7], M[2], e2);
    glue_wave_sync();
    if ((lane & 15) < 9 && g == 0) M[6][e2] = w;  // upd = mul(Wg, ...)
    if ((lane & 15) < 9 && g == 1) M[7][e2] = w;  // mul(mul(Wg, iWgWb), Wb): multiplies dgbias = 0 below
    glue_wave_sync();
  }
}

// element `lane` (< 9) of a 3x3 matrix handed over by value: selects, no lane-dependent index into the kernel arguments
__device__ __forceinline__ float glue_pre_elem(const float (&m)[9], int lane) {
  float v = m[0];
#pragma unroll
  for (int e = 1; e < 9; ++e) v = (lane == e) ? m[e] : v;
  return v;
}
// the matrix set glue_gyro3 would leave, from GlueParams::pre (hm::gyro_pre on the host: the same statements, the same bits)
__device__ __forceinline__ void glue_gyro3_from_pre(float (&M)[10][9], const GlueParams& gp, int depth, int lane) {
  if (lane < 9) {
    M[0][lane] = glue_pre_elem(gp.pre[0], lane);
    if (depth >= 2) {
      M[2][lane] = glue_pre_elem(gp.pre[1], lane);
      M[3][lane] = glue_pre_elem(gp.pre[2], lane);
      M[4][lane] = glue_pre_elem(gp.pre[3], lane);
      M[6][lane] = glue_pre_elem(gp.pre[4], lane);
      M[7][lane] = glue_pre_elem(gp.pre[5], lane);
    }
  }
  glue_wave_sync();
}

// one column of inverse(A) per lane (lanes 0..5), TooN Cholesky<6>::get_inverse: every lane factorises for itself
__device__ __forceinline__ void glue_chol6_inverse_lanes(const float* A /*LDS*/, float* inv /*LDS*/, int lane) {
  if (lane < 6) {
    float L[6][6], res[6];
    hm::cholesky6_factor(A, L);
    hm::cholesky6_inverse_col(L, lane, res);
#pragma unroll
    for (int i = 0; i < 6; ++i) inv[i * 6 + lane] = res[i];
  }
  glue_wave_sync();
}

// hostmath.hpp sym6_ldlt_solve with row i of L on lane i: per column one division on every lane at once instead of up to
// five in a row, the substitutions as chains of broadcasts. Every lane of the wave calls it; returns false (all lanes) when
// the pivots ask for the pseudo-inverse instead.
__device__ __forceinline__ bool glue_sym6_ldlt_lanes(const float* A_ /*LDS*/, const float* b_ /*LDS*/, float* x_ /*LDS*/, double* Ld /*LDS*/,
                                                    int lane) {
  constexpr int N = 6;
  const int i = lane < N ? lane : N - 1;
  double Li[N] = {0, 0, 0, 0, 0, 0}, d[N], dmax = 0;
#pragma unroll
  for (int k = 0; k < N; ++k) dmax = (fabs((double)A_[k * N + k]) > dmax) ? fabs((double)A_[k * N + k]) : dmax;
  bool ok = true;
#pragma unroll
  for (int j = 0; j < N; ++j) {
    if (!ok) continue;
    double Lj[N];
#pragma unroll
    for (int k = 0; k < j; ++k) Lj[k] = glue_bcast(Li[k], j);  // row j of L
    double v = 0.5 * ((double)A_[j * N + j] + (double)A_[j * N + j]);
#pragma unroll
    for (int k = 0; k < j; ++k) v -= Lj[k] * Lj[k] * d[k];
    if (!(v * 1e7 > dmax)) ok = false;
    d[j] = v;
    if (ok) {
      double s = 0.5 * ((double)A_[i * N + j] + (double)A_[j * N + i]);
#pragma unroll
      for (int k = 0; k < j; ++k) s -= Li[k] * Lj[k] * d[k];
      Li[j] = s / v;  // (kept by the lanes below the diagonal; the others never use theirs)
    }
  }
  if (!ok) return false;
  // forward substitution: lane k's running sum is final at step k
  double s = (double)b_[i];
#pragma unroll
  for (int k = 0; k < N; ++k) {
    const double yk = glue_bcast(s, k);
    if (i > k) s -= Li[k] * yk;
  }
  double dv = d[0];
#pragma unroll
  for (int k = 1; k < N; ++k) dv = (i == k) ? d[k] : dv;
  const double y = s / dv;
  // back substitution needs L by column: through LDS
  if (lane < N) {
#pragma unroll
    for (int k = 0; k < N; ++k) Ld[lane * N + k] = Li[k];
  }
  glue_wave_sync();
  double x = y, xk[N];
#pragma unroll
  for (int ii = N - 1; ii >= 0; --ii) {
    // lane ii finishes: x = y - sum over k > ii (ascending) of L[k][ii] * x[k]
    if (i == ii) {
#pragma unroll
      for (int k = ii + 1; k < N; ++k) x -= Ld[k * N + ii] * xk[k];
    }
    xk[ii] = glue_bcast(x, ii);
  }
  if (lane < N) x_[lane] = (float)x;
  glue_wave_sync();
  return true;
}

// Called by EVERY thread of workgroup 0 (>= 256 threads) once the sums of the extRotVel records are in w.W / w.JtF
// (symmetrised float matrix and vector, as hm::sum_xrv leaves them) and `lm` holds the final minimizeVel state.
// One workgroup barrier inside. Results: *gd_copy (second half of the pair), *st_out (filter state after the pair), *rec.
__device__ __forceinline__ void glue_workgroup(GlueLds& w, const LmState& lm /*LDS*/, const GlueArgs& ga,
                                               unsigned long long* stamps = nullptr /*diagnostic: [17] solves done, [18] all waves there*/) {
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const float s_b = ga.gp.gyro_bias_std_dev * ga.gp.gyro_bias_std_dev * ga.gp.frame_dt * ga.gp.frame_dt;
  const float s_g = ga.gp.gyro_std_dev * ga.gp.gyro_std_dev * ga.gp.frame_dt * ga.gp.frame_dt;
  if (wid == 0) {
    // ---- the 6x6 solve of extRotVel (core.cpp:244-248) ----
    const bool ok = glue_sym6_ldlt_lanes(w.W, w.JtF, w.Xv, w.Ld, lane);
    if (!ok && lane == 0) hm::sym6_pinv_solve_ws(w.W, w.JtF, w.Xv, w.ws);
    if (stamps && tid == 0) stamps[17] = __builtin_amdgcn_s_memrealtime();
  } else if (wid == 1) {
    // ---- gyroBiasCorrection up to inverse(Wxb) (core.cpp:264-276) ----
    if (ga.gp.has_pre)
      glue_gyro3_from_pre(w.m3[0], ga.gp, 2, lane);
    else
      glue_gyro3(w.m3[0], ga.st_in, s_g, s_b, 2, lane);
    if (lane < 36) {  // Wxb = Wx with upd added to its lower right block (core.cpp:271-272)
      const int i = lane / 6, j = lane - 6 * i;
      const float v = w.W[lane];
      w.A[0][lane] = (i >= 3 && j >= 3) ? v + w.m3[0][6][(i - 3) * 3 + (j - 3)] : v;
    }
    glue_wave_sync();
    glue_chol6_inverse_lanes(w.A[0], w.inv[0], lane);
  } else if (wid == 2) {
    // ---- W_Xgv after the correction (core.cpp:283) and its inverse (rebvio.cpp:201-202) ----
    if (ga.gp.has_pre)
      glue_gyro3_from_pre(w.m3[1], ga.gp, 1, lane);
    else
      glue_gyro3(w.m3[1], ga.st_in, s_g, s_b, 1, lane);
    if (lane < 36) {
      const int i = lane / 6, j = lane - 6 * i;
      const float v = w.W[lane];
      w.A[1][lane] = (i >= 3 && j >= 3) ? v + w.m3[1][0][(i - 3) * 3 + (j - 3)] : v;
    }
    glue_wave_sync();
    glue_chol6_inverse_lanes(w.A[1], w.inv[1], lane);
  } else if (wid == 3) {
    // ---- invert(JtJ) of minimizeVel (core.cpp:186) for the host's record ----
    if (lane == 0) {
      w.JtJ9[0] = lm.JtJ[0]; w.JtJ9[4] = lm.JtJ[1]; w.JtJ9[8] = lm.JtJ[2];
      w.JtJ9[1] = w.JtJ9[3] = lm.JtJ[3];
      w.JtJ9[2] = w.JtJ9[6] = lm.JtJ[4];
      w.JtJ9[5] = w.JtJ9[7] = lm.JtJ[5];
    }
    glue_wave_sync();
    glue_invert3_one(w.JtJ9, w.P_Vg, lane);
  }
  __syncthreads();
  if (stamps && tid == 0) stamps[18] = __builtin_amdgcn_s_memrealtime();
  if (wid >= 2) return;
  // Both remaining waves form X (core.cpp:273-276) for themselves: X1 = Wx * X (+ (Wg iWgWb Wb) dgbias with dgbias = 0 on
  // entry, as the reference computes it), X = inverse(Wxb) * X1.
  float (&M)[10][9] = w.m3[0];
  if (lane < 6) {
    float s = 0;
#pragma unroll
    for (int k = 0; k < 6; ++k) s += w.W[lane * 6 + k] * w.Xv[k];
    if (lane >= 3) {
      float t = 0;
#pragma unroll
      for (int k = 0; k < 3; ++k) t += M[7][(lane - 3) * 3 + k] * 0.0f;
      s += t;
    }
    w.X1[wid][lane] = s;
  }
  glue_wave_sync();
  if (lane < 6) {
    float s = 0;
#pragma unroll
    for (int k = 0; k < 6; ++k) s += w.inv[0][lane * 6 + k] * w.X1[wid][k];
    w.X[wid][lane] = s;
  }
  glue_wave_sync();
  if (lane != 0) return;
  float Xgv[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) Xgv[i] = w.X[wid][i];
  if (wid == 1) {
    // ---- filter state after the pair: dgbias = iWgWb * (Wg * X[3:6] + Wb * dgbias(=0)), Wb = Wg + Wb, the next pair's prior ----
    GlueState st = *ga.st_in;
    float a3[3], b3[3], sum3[3], dg[3];
    const float xw[3] = {Xgv[3], Xgv[4], Xgv[5]}, zero3[3] = {0.f, 0.f, 0.f};
    hm::mulv(hm::load3(M[0]), xw, a3);
    hm::mulv(hm::load3(M[2]), zero3, b3);
#pragma unroll
    for (int i = 0; i < 3; ++i) sum3[i] = a3[i] + b3[i];
    hm::mulv(hm::load3(M[4]), sum3, dg);
#pragma unroll
    for (int i = 0; i < 3; ++i) st.Bg[i] += dg[i];
#pragma unroll
    for (int i = 0; i < 9; ++i) st.W_Bg[i] = M[3][i];
    const hm::M3 Rn = hm::prior_rotation(st.Bg, hm::identity3());
    hm::store3(Rn, st.R);
    st.pad = 0.f;
    hm::store3(hm::transpose(Rn), ga.gd_copy->RT_next);
    ga.gd_copy->has_next = 1;
    *ga.st_out = st;
    if (ga.stage) {
      ga.stage->rec.gs = st;
    } else {
      ga.rec->gs = st;
      stamp_drain(&ga.rec->seq_gs, ga.seq);
    }
    return;
  }
  // ---- wave 0: SO3 correction, covariance, the second half's inputs, the host's record (rebvio.cpp:195-203, 228) ----
  float Vg[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) Vg[i] = lm.vel[i];
  const float dVgv[3] = {Xgv[0], Xgv[1], Xgv[2]};
  const float dWgv[3] = {Xgv[3], Xgv[4], Xgv[5]};
  const hm::M3 R0 = hm::so3_exp(dWgv);
  const hm::M3 R = hm::transpose(hm::mul(R0, hm::transpose(hm::load3(ga.st_in->R))));
  float V[3];
  hm::mulv(R0, Vg, V);
#pragma unroll
  for (int i = 0; i < 3; ++i) V[i] += dVgv[i];
  float P_V[9];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) P_V[i * 3 + j] = w.inv[1][i * 6 + j];
  GlueDev* gl = ga.gd_copy;
  float vel_r[3], Rvel_r[9], Rgva[9], R0a[9];
  hm::mulv(R, V, vel_r);
  hm::store3(hm::mul(hm::mul(R, hm::load3(P_V)), hm::transpose(R)), Rvel_r);
  hm::store3(R, Rgva);
  hm::store3(R0, R0a);
  const int nan_v = (V[0] != V[0] || V[1] != V[1] || V[2] != V[2]) ? 1 : 0;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    gl->vel_r[i] = vel_r[i];
    gl->V[i] = V[i];
  }
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    gl->Rvel_r[i] = Rvel_r[i];
    gl->Rgva[i] = Rgva[i];
    gl->R0a[i] = R0a[i];
  }
  gl->nan_v = nan_v;
  // the host's record: staged in device memory when a directedMatch launch follows (it forwards the record, GlueStage), else
  // straight into the pinned record
  rebvio_hip_pair_out& out = ga.stage ? ga.stage->rec.out : ga.rec->out;
  out.F = lm.F;
  out.lm_accept_mask = lm.accept_mask;
  out.sigma_rho_min = lm.sigma_rho_min;
  int ext_ok = 1;
#pragma unroll
  for (int i = 0; i < 6; ++i)
    if (w.Xv[i] != w.Xv[i]) ext_ok = 0;
  out.ext_ok = ext_ok;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    out.Vg[i] = Vg[i];
    out.V[i] = V[i];
  }
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    out.P_Vg[i] = w.P_Vg[i];
    out.R[i] = Rgva[i];
    out.P_V[i] = P_V[i];
  }
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    out.Xv[i] = w.Xv[i];
    out.Xgv[i] = Xgv[i];
  }
#pragma unroll
  for (int i = 0; i < 36; ++i) out.W_Xv[i] = w.W[i];
  out.klm_num = out.kf_matches = out.reg_num = 0;
  out.status = nan_v;
  if (ga.stage) {
    ga.stage->host_rec = ga.rec;
    ga.stage->seq = ga.seq;
  } else {
    stamp_drain(&ga.rec->seq_out, ga.seq);
  }
}

}  // namespace rh
