// Device form of the pair glue (glue.hpp: pair_glue_core), run by workgroup 0 of the persistent LM kernel behind its last
// phase: the same statements per output element as the host form, spread over three waves and a few lanes each so that the
// ~2500 dependent scalar operations of the host version become a chain of ~600:
//   wave 0  the 6x6 solve of extRotVel in double (core.cpp:244-248), then everything that needs its solution
//   wave 1  gyroBiasCorrection's 3x3 inverses (core.cpp:264-284; nine lanes, one cofactor each), then the two LDL^T
//           inverses (core.cpp:276, rebvio.cpp:201-202), one column per lane
//   wave 2  invert(JtJ) of minimizeVel (core.cpp:186) for the host's record
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
  float m3[8][9];    // 0 Wg  1 inv(W_Bg)+RGBias  2 W_Bg'  3 Wg+W_Bg'  4 iWgWb  5 iWgWb*Wg  6 upd  7 (Wg*iWgWb)*W_Bg'
  float A[2][36];    // 0: Wxb (core.cpp:271-272)   1: W_Xgv after the correction (core.cpp:283)
  float inv[2][36];
  float X1[6], X[6];
  float P_Vg[9];
  float tmp3[9];
  double ws[72];     // Jacobi fall-back of the 6x6 solve
};

// wave-level hand-over through LDS: the LDS pipeline of a wave is in order, so only the compiler has to be kept from moving
// accesses across this point
__device__ __forceinline__ void glue_wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
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

// types::invert (types/definitions.hpp:40-53): lanes 0..8 of the calling wave each produce one element of dst = inverse(src)
__device__ __forceinline__ void glue_invert3_lanes(const float* src /*LDS*/, float* dst /*LDS*/, int lane) {
  float m[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) m[i] = src[i];
  const float d = glue_det3(m);
  float o;
  switch (lane) {
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
  glue_wave_sync();  // (src may alias nothing written here, but keep the phases apart for the compiler)
  if (lane < 9) dst[lane] = o / d;
  glue_wave_sync();
}

// element (i, j) = lane of x * y for 3x3 matrices in LDS: dot product accumulated from 0 in index order (hostmath.hpp mul)
__device__ __forceinline__ float glue_mul3_elem(const float* x, const float* y, int lane) {
  const int i = lane / 3, j = lane - 3 * i;
  float s = 0;
#pragma unroll
  for (int k = 0; k < 3; ++k) s += x[i * 3 + k] * y[k * 3 + j];
  return s;
}

// Called by EVERY thread of workgroup 0 (>= 192 threads) once the sums of the extRotVel records are in w.W / w.JtF
// (symmetrised float matrix and vector, as hm::sum_xrv leaves them) and `lm` holds the final minimizeVel state.
// Workgroup barriers inside. Results: *gd_out (second half of the pair), *st_out (filter state after the pair), *rec.
__device__ __forceinline__ void glue_workgroup(GlueLds& w, const LmState& lm /*LDS or registers*/, const GlueArgs& ga) {
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const float s_b = ga.gp.gyro_bias_std_dev * ga.gp.gyro_bias_std_dev * ga.gp.frame_dt * ga.gp.frame_dt;
  const float s_g = ga.gp.gyro_std_dev * ga.gp.gyro_std_dev * ga.gp.frame_dt * ga.gp.frame_dt;
  if (wid == 0) {
    if (lane == 0) hm::sym6_solve_ws(w.W, w.JtF, w.Xv, w.ws);
  } else if (wid == 1) {
    // ---- gyroBiasCorrection's 3x3 algebra (hostmath.hpp gyro_bias_correction), nine lanes ----
    const int l9 = lane < 9 ? lane : 8;
    const int di = l9 / 3, dj = l9 - 3 * di;
    if (lane < 9) {
      w.m3[5][lane] = (di == dj) ? s_g : 0.0f;  // RGyro = diag3(s_g) (scratch slot 5 until iWgWb*Wg is formed)
      w.m3[7][lane] = ga.st_in->W_Bg[lane];     // W_Bg (scratch slot 7)
    }
    glue_wave_sync();
    glue_invert3_lanes(w.m3[5], w.m3[0], l9);   // Wg = invert3(Rg)
    glue_invert3_lanes(w.m3[7], w.tmp3, l9);    // invert3(Wb)
    if (lane < 9) w.m3[1][lane] = w.tmp3[lane] + ((di == dj) ? s_b : 0.0f);  // add(invert3(Wb), Rb)
    glue_wave_sync();
    glue_invert3_lanes(w.m3[1], w.m3[2], l9);   // Wb = invert3(...)
    if (lane < 9) w.m3[3][lane] = w.m3[0][lane] + w.m3[2][lane];  // add(Wg, Wb)
    glue_wave_sync();
    glue_invert3_lanes(w.m3[3], w.m3[4], l9);   // iWgWb
    if (lane < 9) w.m3[5][lane] = glue_mul3_elem(w.m3[4], w.m3[0], l9);  // mul(iWgWb, Wg)
    glue_wave_sync();
    if (lane < 9) w.tmp3[lane] = ((di == dj) ? 1.0f : 0.0f) - w.m3[5][lane];  // sub(identity3(), ...)
    glue_wave_sync();
    if (lane < 9) w.m3[6][lane] = glue_mul3_elem(w.m3[0], w.tmp3, l9);  // upd = mul(Wg, ...)
    glue_wave_sync();
    if (lane < 9) w.tmp3[lane] = glue_mul3_elem(w.m3[0], w.m3[4], l9);  // mul(Wg, iWgWb)
    glue_wave_sync();
    if (lane < 9) w.m3[7][lane] = glue_mul3_elem(w.tmp3, w.m3[2], l9);  // mul(mul(Wg, iWgWb), Wb): multiplies dgbias = 0 below
    // Wxb = Wx with upd added to its lower right block; Wx' = Wx with Wg added there (core.cpp:271-272, 283)
    if (lane < 36) {
      const int i = lane / 6, j = lane - 6 * i;
      const bool blk = i >= 3 && j >= 3;
      const float v = w.W[lane];
      w.A[0][lane] = blk ? v + w.m3[6][(i - 3) * 3 + (j - 3)] : v;
      w.A[1][lane] = blk ? v + w.m3[0][(i - 3) * 3 + (j - 3)] : v;
    }
    glue_wave_sync();
    // the two LDL^T inverses, one column per lane: lanes 0..5 -> inverse(Wxb), lanes 8..13 -> inverse(W_Xgv')
    if ((lane & 7) < 6 && lane < 16) {
      const int which = lane >> 3, c = lane & 7;
      float L[6][6], res[6];
      hm::cholesky6_factor(w.A[which], L);
      hm::cholesky6_inverse_col(L, c, res);
#pragma unroll
      for (int i = 0; i < 6; ++i) w.inv[which][i * 6 + c] = res[i];
    }
  } else if (wid == 2) {
    if (lane == 0) hm::lm_rvel(lm, w.P_Vg);
  }
  __syncthreads();
  if (wid == 0) {
    // X1 = Wx * X (+ (Wg iWgWb Wb) dgbias with dgbias = 0 on entry, as the reference computes it), X = inverse(Wxb) * X1
    if (lane < 6) {
      float s = 0;
#pragma unroll
      for (int k = 0; k < 6; ++k) s += w.W[lane * 6 + k] * w.Xv[k];
      if (lane >= 3) {
        float t = 0;
#pragma unroll
        for (int k = 0; k < 3; ++k) t += w.m3[7][(lane - 3) * 3 + k] * 0.0f;
        s += t;
      }
      w.X1[lane] = s;
    }
    glue_wave_sync();
    if (lane < 6) {
      float s = 0;
#pragma unroll
      for (int k = 0; k < 6; ++k) s += w.inv[0][lane * 6 + k] * w.X1[k];
      w.X[lane] = s;
    }
    glue_wave_sync();
    if (lane == 0) {
      GlueState st = *ga.st_in;
      GlueDev gl;
      GlueRec rec;
      rebvio_hip_pair_out& out = rec.out;
      float Xgv[6];
#pragma unroll
      for (int i = 0; i < 6; ++i) Xgv[i] = w.X[i];
      // dgbias = iWgWb * (Wg * X[3:6] + Wb * dgbias(=0))
      float a3[3], b3[3], sum3[3], dg[3];
      const float xw[3] = {Xgv[3], Xgv[4], Xgv[5]}, zero3[3] = {0.f, 0.f, 0.f};
      hm::mulv(hm::load3(w.m3[0]), xw, a3);
      hm::mulv(hm::load3(w.m3[2]), zero3, b3);
#pragma unroll
      for (int i = 0; i < 3; ++i) sum3[i] = a3[i] + b3[i];
      hm::mulv(hm::load3(w.m3[4]), sum3, dg);
#pragma unroll
      for (int i = 0; i < 3; ++i) st.Bg[i] += dg[i];
#pragma unroll
      for (int i = 0; i < 9; ++i) st.W_Bg[i] = w.m3[3][i];  // Wb = add(Wg, Wb)
      float Vg[3];
#pragma unroll
      for (int i = 0; i < 3; ++i) Vg[i] = lm.vel[i];
      const float dVgv[3] = {Xgv[0], Xgv[1], Xgv[2]};
      const float dWgv[3] = {Xgv[3], Xgv[4], Xgv[5]};
      const hm::M3 R0 = hm::so3_exp(dWgv);
      const hm::M3 R = hm::transpose(hm::mul(R0, hm::transpose(hm::load3(st.R))));
      float V[3];
      hm::mulv(R0, Vg, V);
#pragma unroll
      for (int i = 0; i < 3; ++i) V[i] += dVgv[i];
      float P_V[9];
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) P_V[i * 3 + j] = w.inv[1][i * 6 + j];
      hm::mulv(R, V, gl.vel_r);
      hm::store3(hm::mul(hm::mul(R, hm::load3(P_V)), hm::transpose(R)), gl.Rvel_r);
      hm::store3(R, gl.Rgva);
      hm::store3(R0, gl.R0a);
#pragma unroll
      for (int i = 0; i < 3; ++i) gl.V[i] = V[i];
      gl.nan_v = (V[0] != V[0] || V[1] != V[1] || V[2] != V[2]) ? 1 : 0;
      const hm::M3 Rn = hm::prior_rotation(st.Bg, hm::identity3());
      hm::store3(Rn, st.R);
      hm::store3(hm::transpose(Rn), gl.RT_next);
      gl.has_next = 1;
      st.pad = 0.f;
      out.F = lm.F;
      out.lm_accept_mask = lm.accept_mask;
      out.sigma_rho_min = lm.sigma_rho_min;
      out.ext_ok = 1;
#pragma unroll
      for (int i = 0; i < 6; ++i)
        if (w.Xv[i] != w.Xv[i]) out.ext_ok = 0;
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        out.Vg[i] = Vg[i];
        out.V[i] = V[i];
      }
#pragma unroll
      for (int i = 0; i < 9; ++i) {
        out.P_Vg[i] = w.P_Vg[i];
        out.R[i] = gl.Rgva[i];
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
      out.status = gl.nan_v ? 1 : 0;
      rec.gs = st;
      *ga.gd_copy = gl;
      *ga.st_out = st;
      *ga.rec = rec;
    }
  }
}

}  // namespace rh
