// Edge detection on gfx950: integral-image box filters (bit-exact evaluation order of the reference),
// DoG + squared gradient, keyline extraction with ordered (raster-rank) compaction, edge chaining,
// auto-threshold statistics and the integer distance field.
//
// Reference semantics (baumlin/rebvio): scale_space.cpp:48-128,203-233; edge_detector.cpp:30-186;
// core.hpp:37-71. All kernels are compiled with -ffp-contract=off: every +,-,* is a separately
// rounded fp32 operation exactly as in the CPU path, divisions and sqrt are IEEE correctly rounded.
//
// Bound: HBM/L2 bandwidth for the per-pixel sweeps; the two scan kernels are additionally bound by
// their sequential fp32 add chains (cols resp. rows dependent adds) which bit-exactness requires.
#include <algorithm>
#include <cstdio>
#include <cstdlib>

#include <cstring>

#include "common.hpp"

namespace rh {

__constant__ float c_recip[128];  // c_recip[n] = float(1.0 / double(n))  (scale_space.cpp:166-170)
__constant__ float c_pinv[75];    // Pinv = invert(Phi^T Phi) Phi^T, 3 x 25 (edge_detector.cpp:55-68)

void upload_tables(const float* recip128, const float* pinv75) {
  (void)hipMemcpyToSymbol(HIP_SYMBOL(c_recip), recip128, sizeof(float) * 128);
  (void)hipMemcpyToSymbol(HIP_SYMBOL(c_pinv), pinv75, sizeof(float) * 75);
}

// FastGaussian::average (scale_space.cpp:69-128): the nine border/interior cases with their distinct
// operand orders. II is the integral image, d the box width.
// `ld` = row pitch of II in floats (cols rounded up to a multiple of 4: the scan kernels move 16-byte vectors; the
// columns >= C of a row hold prefix values past the image and are never read here: every tap is clamped to C - 1).
__device__ __forceinline__ float box_avg(const float* __restrict__ II, int r, int c, int d, int R, int C, int ld) {
  const int d2 = d >> 1;
  const bool top = r < d2 + 1, bot = r >= R - d2;
  const bool left = c < d2 + 1, right = c >= C - d2;
  const int r1 = (bot && !top) ? R - 1 : r + d2;
  const int c1 = (right && !left) ? C - 1 : c + d2;
  const int ny = top ? r + d2 + 1 : (bot ? R - r + d2 : d);
  const int nx = left ? c + d2 + 1 : (right ? C - c + d2 : d);
  const float div = c_recip[nx * ny];
  const float A = II[(size_t)r1 * ld + c1];
  float acc;
  if (top) {
    acc = left ? A : A - II[(size_t)r1 * ld + (c - d2 - 1)];
  } else {
    const int r2 = r - d2 - 1;
    const float Cc = II[(size_t)r2 * ld + c1];
    if (left) {
      acc = A - Cc;
    } else {
      const int c2 = c - d2 - 1;
      const float B = II[(size_t)r1 * ld + c2];
      const float D = II[(size_t)r2 * ld + c2];
      acc = bot ? (((A - Cc) - B) + D) : (((A - B) - Cc) + D);
    }
  }
  return acc * div;
}

// Straight-line form of box_avg for border pixels: four unconditional (clamped) taps, absent terms replaced by
// +0 (x - 0 and x + 0 are exact and no -0 can arise from the differences), the operand order of the bottom rows
// selected by swapping B and C. Same bits as box_avg, no control flow, so the taps of several pixels overlap.
__device__ __forceinline__ float box_avg_flat(const float* __restrict__ II, int r, int c, int d, int R, int C, int ld) {
  const int d2 = d >> 1;
  const bool top = r < d2 + 1, bot = r >= R - d2;
  const bool left = c < d2 + 1, right = c >= C - d2;
  const int r1 = (bot && !top) ? R - 1 : r + d2;
  const int c1 = min((right && !left) ? C - 1 : c + d2, C - 1);  // (the clamp only acts in the pitch padding, c >= C)
  const int r2 = max(r - d2 - 1, 0), c2 = max(c - d2 - 1, 0);
  const int ny = top ? r + d2 + 1 : (bot ? R - r + d2 : d);
  const int nx = left ? c + d2 + 1 : (right ? C - c + d2 : d);
  const float div = c_recip[max(nx, 0) * ny];
  const float A = II[(size_t)r1 * ld + c1];
  const float Bv = II[(size_t)r1 * ld + c2];
  const float Cv = II[(size_t)r2 * ld + c1];
  const float Dv = II[(size_t)r2 * ld + c2];
  const float B = left ? 0.0f : Bv;
  const float Cc = top ? 0.0f : Cv;
  const float D = (left || top) ? 0.0f : Dv;
  const float X = (bot && !top) ? Cc : B;
  const float Y = (bot && !top) ? B : Cc;
  return (((A - X) - Y) + D) * div;
}

// ---- row prefix (scale_space.cpp:50-57) ---------------------------------------------------------------
// One workgroup stages a strip of kStrip rows in LDS with coalesced 16-byte accesses (for passes 2 and 3
// the box average of the previous integral image is evaluated on the fly), then one lane per row walks
// its row left to right: a sequential fp32 chain, which is what keeps the result bit-identical to the
// CPU. LDS row pitch = cols + pad with (pitch/4) odd: the row-lanes' ds_read/write_b128 hit distinct
// 4-bank groups. Measured (in-kernel stamps): the chain costs ~20 cycles per element on its lone wave
// (LDS b128 issue, not the 4.8-cycle dependent add), the staging phase is bound by per-CU load bandwidth,
// hence short strips on many CUs.
constexpr int kStrip = 4;   // rows per workgroup: 120 workgroups per 480-row image spread the tap traffic over the CUs

// Interior fast path of box_avg for four consecutive columns: rows r1 = r+d2 and r2 = r-d2-1 are read as two
// 4-float windows each (dword-aligned 16-byte loads) instead of 16 scalar taps. Same operand order as box_avg.
__device__ __forceinline__ float4 box_avg4_interior(const float* __restrict__ II, int r, int c, int d, int ld) {
  const int d2 = d >> 1;
  const float* p1 = II + (size_t)(r + d2) * ld + c;
  const float* p2 = II + (size_t)(r - d2 - 1) * ld + c;
  const float4 A = *reinterpret_cast<const float4*>(p1 + d2);
  const float4 B = *reinterpret_cast<const float4*>(p1 - d2 - 1);
  const float4 Cc = *reinterpret_cast<const float4*>(p2 + d2);
  const float4 D = *reinterpret_cast<const float4*>(p2 - d2 - 1);
  const float a = c_recip[d * d];
  float4 o;
  o.x = (((A.x - B.x) - Cc.x) + D.x) * a;
  o.y = (((A.y - B.y) - Cc.y) + D.y) * a;
  o.z = (((A.z - B.z) - Cc.z) + D.z) * a;
  o.w = (((A.w - B.w) - Cc.w) + D.w) * a;
  return o;
}

// Input images (MODE 0 / 1) are dense (row stride C). Widths that are not a multiple of 4 take element loads guarded at
// the row end (the vector forms would straddle rows and run past the last one); the columns >= C of the pitch padding
// read 0. Integral images (MODE 2) have the padded pitch `ld`.
template <int MODE>
__device__ __forceinline__ float4 rowscan_fetch(const void* __restrict__ src, int r, int c4, int d, int d2, int R, int C, int ld) {
  const int c = c4 * 4;
  if (MODE == 0) {
    if ((C & 3) == 0) {
      const uchar4 u = reinterpret_cast<const uchar4*>(src)[((size_t)r * C + c) >> 2];
      return make_float4((float)u.x * 3.0f, (float)u.y * 3.0f, (float)u.z * 3.0f, (float)u.w * 3.0f);
    }
    const uint8_t* row = reinterpret_cast<const uint8_t*>(src) + (size_t)r * C;
    float4 v;
    v.x = (c < C) ? (float)row[c] * 3.0f : 0.0f;
    v.y = (c + 1 < C) ? (float)row[c + 1] * 3.0f : 0.0f;
    v.z = (c + 2 < C) ? (float)row[c + 2] * 3.0f : 0.0f;
    v.w = (c + 3 < C) ? (float)row[c + 3] * 3.0f : 0.0f;
    return v;
  } else if (MODE == 1) {
    if ((C & 3) == 0) return *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(src) + (size_t)r * C + c);
    const float* row = reinterpret_cast<const float*>(src) + (size_t)r * C;
    float4 v;
    v.x = (c < C) ? row[c] : 0.0f;
    v.y = (c + 1 < C) ? row[c + 1] : 0.0f;
    v.z = (c + 2 < C) ? row[c + 2] : 0.0f;
    v.w = (c + 3 < C) ? row[c + 3] : 0.0f;
    return v;
  } else {
    const float* II = reinterpret_cast<const float*>(src);
    if (r > d2 && r < R - d2 && c > d2 && c + 3 < C - d2) return box_avg4_interior(II, r, c, d, ld);
    float4 v;
    v.x = box_avg_flat(II, r, c, d, R, C, ld);
    v.y = box_avg_flat(II, r, c + 1, d, R, C, ld);
    v.z = box_avg_flat(II, r, c + 2, d, R, C, ld);
    v.w = box_avg_flat(II, r, c + 3, d, R, C, ld);
    return v;
  }
}

// ---- the sequential fp32 chain of a scan, on one wave with the data in registers ------------------------------------
// Bit-exactness pins the ORDER of the adds (s += x[i], i ascending), not who performs them. A lone lane walking an LDS
// array pays an LDS round trip per group of elements (reads and writes share the in-order lgkm counter: ~20 cycles per
// element measured). Here the n4 float4 chunks of the array are dealt to the lanes of one wave in blocks of KQ consecutive
// chunks (one LDS read), and the chain hops from lane to lane: step j runs the 4*KQ dependent adds of lane j's block with
// only that lane enabled, the running sum crosses to the next lane through v_readlane. No memory operation sits on the
// chain: ~5 cycles per element. Same adds in the same order as the one-lane loop, so the same bits.
// The first element is taken as it is (the reference starts the sum from it, scale_space.cpp:52,60).
template <int KQ>
__device__ __forceinline__ void wave_chain(float* __restrict__ arr /* LDS, 16-byte aligned */, int n4) {
  const int lane = (int)threadIdx.x & 63;
  const int first = lane * KQ;
  float4 a[KQ];
#pragma unroll
  for (int q = 0; q < KQ; ++q)
    a[q] = (first + q < n4) ? *reinterpret_cast<const float4*>(arr + (size_t)(first + q) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
  const int nl = (n4 + KQ - 1) / KQ;  // lanes that hold data
  float total = 0.f;
  for (int j = 0; j < nl; ++j) {
    const float carry = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(total), j > 0 ? j - 1 : 0));
    if (lane == j) {
      float p = (j == 0) ? a[0].x : carry + a[0].x;
#pragma unroll
      for (int q = 0; q < KQ; ++q) {
        if (q > 0) p = p + a[q].x;
        a[q].x = p;
        p = p + a[q].y;
        a[q].y = p;
        p = p + a[q].z;
        a[q].z = p;
        p = p + a[q].w;
        a[q].w = p;
      }
      total = p;
    }
  }
#pragma unroll
  for (int q = 0; q < KQ; ++q)
    if (first + q < n4) *reinterpret_cast<float4*>(arr + (size_t)(first + q) * 4) = a[q];
}
// The same array layout, for the FIRST row pass only: its input is the u8 frame times 3.0f, so every partial sum of a row is an
// integer below 2^24 (cols <= 4096: at most 3.1e6) and exactly representable - any order of the adds gives the bits of the
// sequential one. So the lanes prefix their own blocks, an inclusive scan over the lanes' totals (six shuffles) gives every
// lane its offset, and one more add per element finishes: ~40 dependent adds instead of one per element.
template <int KQ>
__device__ __forceinline__ void wave_prefix_exact(float* __restrict__ arr /* LDS, 16-byte aligned */, int n4) {
  const int lane = (int)threadIdx.x & 63;
  const int first = lane * KQ;
  float4 a[KQ];
  float p = 0.f;
#pragma unroll
  for (int q = 0; q < KQ; ++q) {
    a[q] = (first + q < n4) ? *reinterpret_cast<const float4*>(arr + (size_t)(first + q) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    p = p + a[q].x;
    a[q].x = p;
    p = p + a[q].y;
    a[q].y = p;
    p = p + a[q].z;
    a[q].z = p;
    p = p + a[q].w;
    a[q].w = p;
  }
  float incl = p;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const float t = __shfl_up(incl, d);
    if (lane >= d) incl = incl + t;
  }
  const float off = incl - p;  // sum of the lower lanes' blocks (exact)
#pragma unroll
  for (int q = 0; q < KQ; ++q)
    if (first + q < n4) {
      a[q].x = off + a[q].x;
      a[q].y = off + a[q].y;
      a[q].z = off + a[q].z;
      a[q].w = off + a[q].w;
      *reinterpret_cast<float4*>(arr + (size_t)(first + q) * 4) = a[q];
    }
}
__device__ __forceinline__ bool wave_prefix_exact_any(float* __restrict__ arr, int n4) {
  switch ((n4 + 63) >> 6) {
    case 1: wave_prefix_exact<1>(arr, n4); return true;
    case 2: wave_prefix_exact<2>(arr, n4); return true;
    case 3: wave_prefix_exact<3>(arr, n4); return true;
    case 4: wave_prefix_exact<4>(arr, n4); return true;
    case 5: wave_prefix_exact<5>(arr, n4); return true;
    case 6: wave_prefix_exact<6>(arr, n4); return true;
    case 7: wave_prefix_exact<7>(arr, n4); return true;
    case 8: wave_prefix_exact<8>(arr, n4); return true;
    default: return false;
  }
}
// n4 <= 512 chunks (2048 elements); longer arrays keep the one-lane loop of the caller
__device__ __forceinline__ bool wave_chain_any(float* __restrict__ arr, int n4) {
  switch ((n4 + 63) >> 6) {
    case 1: wave_chain<1>(arr, n4); return true;
    case 2: wave_chain<2>(arr, n4); return true;
    case 3: wave_chain<3>(arr, n4); return true;
    case 4: wave_chain<4>(arr, n4); return true;
    case 5: wave_chain<5>(arr, n4); return true;
    case 6: wave_chain<6>(arr, n4); return true;
    case 7: wave_chain<7>(arr, n4); return true;
    case 8: wave_chain<8>(arr, n4); return true;
    default: return false;
  }
}

template <int MODE>  // 0: u8 image * 3.0f, 1: fp32 image, 2: box average (width d) of an integral image
__device__ __forceinline__ void rowscan_body(const void* __restrict__ src0, const void* __restrict__ src1,
                                                 float* __restrict__ dst0, float* __restrict__ dst1, int R, int Cimg, int d0,
                                                 int d1, int ldw) {
  const uint2 vb = xcd_band_block();  // tile coordinates: contiguous bands of tiles per XCD (see xcd_band_block)
  // C = row pitch of the integral images = the image width rounded up to a multiple of 4; the chain also runs over the
  // padding columns (they follow the image's columns, so the image's prefix values do not depend on them)
  const int C = (Cimg + 3) & ~3;
  extern __shared__ float4 smem4[];
  float* tile = reinterpret_cast<float*>(smem4);
  const int f = vb.y;
  const void* __restrict__ src = f ? src1 : src0;
  float* __restrict__ dst = f ? dst1 : dst0;
  const int d = f ? d1 : d0;
  const int d2 = d >> 1;
  const int r0 = vb.x * kStrip;
  const int nrows = min(kStrip, R - r0);
  const int C4 = C >> 2;
  // thread -> (strip row, 16-byte chunk): one wave per row, 64 chunks (1 KiB) per pass; no divisions.
  const int lr = (int)threadIdx.x >> 6;  // kStrip == 4 rows, 256 threads
  const int cb = (int)threadIdx.x & 63;
  const int r = min(r0 + lr, R - 1);
  const bool rvalid = lr < nrows;
  constexpr int kBatch = 3;  // 640 columns = 160 chunks = 2.5 passes
  constexpr int kCPP = 64;   // chunks per pass
  for (int base = 0; base < C4; base += kCPP * kBatch) {
    // named registers, not an array (arrays of float4 end up in scratch with this compiler)
    const float4 v0 = rowscan_fetch<MODE>(src, r, min(base + 0 * kCPP + cb, C4 - 1), d, d2, R, Cimg, C);
    const float4 v1 = rowscan_fetch<MODE>(src, r, min(base + 1 * kCPP + cb, C4 - 1), d, d2, R, Cimg, C);
    const float4 v2 = rowscan_fetch<MODE>(src, r, min(base + 2 * kCPP + cb, C4 - 1), d, d2, R, Cimg, C);
    __builtin_amdgcn_sched_barrier(0);  // all loads of the batch are issued before the first LDS write
#define RH_RS_ST(k, v)                                                                  \
    {                                                                                   \
      const int c4 = base + (k) * kCPP + cb;                                            \
      if (c4 < C4 && rvalid) *reinterpret_cast<float4*>(&tile[lr * ldw + c4 * 4]) = v;  \
    }
    RH_RS_ST(0, v0) RH_RS_ST(1, v1) RH_RS_ST(2, v2)
#undef RH_RS_ST
  }
  __syncthreads();
  const bool wide = C4 <= 512;  // wave lr chains strip row lr in registers (wave_chain); rows of more than 2048 columns: one lane per row below
  if (wide && rvalid) {
    if (MODE == 0)
      (void)wave_prefix_exact_any(tile + lr * ldw, C4);  // integer-valued sums: order free (see wave_prefix_exact)
    else
      (void)wave_chain_any(tile + lr * ldw, C4);
  }
  if (!wide && (int)threadIdx.x < nrows) {
    // sequential chain over the row; LDS reads run four float4 ahead of the adds (two register groups)
    float* rowp = tile + threadIdx.x * ldw;
    float4 v = *reinterpret_cast<float4*>(rowp);
    v.y = v.x + v.y;
    v.z = v.y + v.z;
    v.w = v.z + v.w;
    *reinterpret_cast<float4*>(rowp) = v;
    float s = v.w;
    constexpr int G = 4;
    float4 a[G], bq[G];
    int c = 4;
    const int lastc = C - 4;
#pragma unroll
    for (int k = 0; k < G; ++k) a[k] = *reinterpret_cast<float4*>(rowp + min(c + 4 * k, lastc));
    for (; c + 8 * G <= C; c += 8 * G) {
#pragma unroll
      for (int k = 0; k < G; ++k) bq[k] = *reinterpret_cast<float4*>(rowp + c + 4 * G + 4 * k);
#pragma unroll
      for (int k = 0; k < G; ++k) {
        float4 w = a[k];
        w.x = s + w.x;
        w.y = w.x + w.y;
        w.z = w.y + w.z;
        w.w = w.z + w.w;
        s = w.w;
        *reinterpret_cast<float4*>(rowp + c + 4 * k) = w;
      }
#pragma unroll
      for (int k = 0; k < G; ++k) a[k] = *reinterpret_cast<float4*>(rowp + min(c + 8 * G + 4 * k, lastc));
#pragma unroll
      for (int k = 0; k < G; ++k) {
        float4 w = bq[k];
        w.x = s + w.x;
        w.y = w.x + w.y;
        w.z = w.y + w.z;
        w.w = w.z + w.w;
        s = w.w;
        *reinterpret_cast<float4*>(rowp + c + 4 * G + 4 * k) = w;
      }
    }
    for (; c < C; c += 4) {
      float4 w = *reinterpret_cast<float4*>(rowp + c);
      w.x = s + w.x;
      w.y = w.x + w.y;
      w.z = w.y + w.z;
      w.w = w.z + w.w;
      s = w.w;
      *reinterpret_cast<float4*>(rowp + c) = w;
    }
  }
  __syncthreads();
  for (int base = 0; base < C4; base += kCPP * kBatch) {
#pragma unroll
    for (int k = 0; k < kBatch; ++k) {
      const int c4 = base + k * kCPP + cb;
      if (c4 < C4 && rvalid)
        reinterpret_cast<float4*>(dst)[(size_t)(r0 + lr) * C4 + c4] = *reinterpret_cast<float4*>(&tile[lr * ldw + c4 * 4]);
    }
  }
}

template <int MODE>
__global__ __launch_bounds__(256) void k_rowscan(const void* __restrict__ src0, const void* __restrict__ src1, float* __restrict__ dst0,
                                                 float* __restrict__ dst1, int R, int Cimg, int d0, int d1, int ldw) {
  rowscan_body<MODE>(src0, src1, dst0, dst1, R, Cimg, d0, d1, ldw);
}
// batched form (lane = blockIdx.z). stage 0: the lane's u8 frame -> a[0]; 1: a[0] -> b[0], b[1]; 2: b[0], b[1] -> a[0], a[1]
template <int MODE>
__global__ __launch_bounds__(256) void k_rowscan_b(const LaneStatic* __restrict__ ls, LaneDynB dyn, int lane0, int stage, int R, int Cimg,
                                                   int d0, int d1, int ldw) {
  const int lane = lane0 + blockIdx.z;
  const LaneStatic& L = ls[lane];
  // stage 0: first pass on the lane's u8 frame; 3: first pass on the fp32 frame its front end left (lens model set)
  const bool first = stage == 0 || stage == 3;
  const void* s0 = stage == 0 ? dyn.v[lane].img
                              : (stage == 3 ? (const void*)gptr(L.undist_img[dyn.v[lane].parity]) : (stage == 1 ? (const void*)gptr(L.sa[0]) : (const void*)gptr(L.sb[0])));
  const void* s1 = first ? s0 : (stage == 1 ? (const void*)gptr(L.sa[0]) : (const void*)gptr(L.sb[1]));
  float* o0 = stage == 1 ? gptr(L.sb[0]) : gptr(L.sa[0]);
  float* o1 = first ? gptr(L.sa[0]) : (stage == 1 ? gptr(L.sb[1]) : gptr(L.sa[1]));
  if (stage == 4) {  // stage 2 for the fused candidate kernel: that one reads the third pass's integral images on the keyline stream
    // while this stream already scans the next step - they go to the step parity's DoG / gradient buffers, which the fused
    // path does not use otherwise (same size, same hand-over events)
    const int par = dyn.v[lane].parity;
    s0 = gptr(L.sb[0]);
    s1 = gptr(L.sb[1]);
    o0 = gptr(L.dog2[par]);
    o1 = gptr(L.mag2[par]);
  }
  rowscan_body<MODE>(s0, s1, o0, o1, R, Cimg, d0, d1, ldw);
}

// ---- column accumulation (scale_space.cpp:59-65) --------------------------------------------------------------
// The row pass transposed: a workgroup stages a strip of kColStrip columns x all rows in LDS, TRANSPOSED
// (tile[col][row]), so that the lane that owns a column walks down it with ds_read/write_b128 (four rows per LDS
// instruction; the chain is LDS-issue bound on its lone wave). Column pitch = rows + pad with (pitch/4) odd.
// Narrow strips spread the staging traffic over many CUs. No global-memory latency sits on the chain.
constexpr int kColStrip = 16;

// zero != null: the first workgroup also clears zero[0 .. nzero) - the per-row candidate counters of the frame, for the fused
// candidate kernel behind the last column pass (k_keyline_flag_ii; k_dog_mag does it on the unfused path)
__device__ __forceinline__ void colscan_body(float* __restrict__ buf0, float* __restrict__ buf1, int R, int C /* pitch */, int ldh,
                                             int* __restrict__ zero = nullptr, int nzero = 0) {
  const uint2 vb = xcd_band_block();  // tile coordinates: contiguous bands of tiles per XCD (see xcd_band_block)
  if (zero && vb.x == 0 && vb.y == 0)
    for (int i = threadIdx.x; i < nzero; i += 256) zero[i] = 0;
  extern __shared__ float4 smem4[];
  float* tile = reinterpret_cast<float*>(smem4);  // [kColStrip][ldh]
  float* __restrict__ buf = vb.y ? buf1 : buf0;
  const int c0 = vb.x * kColStrip;
  const int ncols = min(kColStrip, C - c0);  // multiple of 4 (cols % 4 == 0)
  const int q = ncols >> 2;                  // float4 per row of the strip (<= 4)
  // thread -> (row offset, 16-byte column chunk): 4 chunks per strip row, 64 rows per pass of the 256 threads
  const int c4 = min((int)threadIdx.x & 3, q - 1);
  const bool cvalid = ((int)threadIdx.x & 3) < q;
  const int rb = (int)threadIdx.x >> 2;
  const float* gsrc = buf + c0 + c4 * 4;
  constexpr int kBatch = 4;
  constexpr int kRowsPerPass = 64;
  for (int base = 0; base < R; base += kRowsPerPass * kBatch) {
    // named registers, not an array: hipcc sends even a 4-entry float4 staging array to scratch here
#define RH_CS_LD(k) *reinterpret_cast<const float4*>(gsrc + (size_t)min(base + (k) * kRowsPerPass + rb, R - 1) * C)
    const float4 v0 = RH_CS_LD(0), v1 = RH_CS_LD(1), v2 = RH_CS_LD(2), v3 = RH_CS_LD(3);
#undef RH_CS_LD
    __builtin_amdgcn_sched_barrier(0);  // keep every load of the batch in flight before the first LDS write
#define RH_CS_ST(k, v)                                   \
    {                                                    \
      const int r = base + (k) * kRowsPerPass + rb;      \
      if (r < R && cvalid) {                             \
        float* t = tile + (c4 * 4) * ldh + r;            \
        t[0] = v.x;                                      \
        t[ldh] = v.y;                                    \
        t[2 * ldh] = v.z;                                \
        t[3 * ldh] = v.w;                                \
      }                                                  \
    }
    RH_CS_ST(0, v0) RH_CS_ST(1, v1) RH_CS_ST(2, v2) RH_CS_ST(3, v3)
#undef RH_CS_ST
  }
  __syncthreads();
  if ((int)threadIdx.x < ncols) {
    // sequential chain down the column; LDS reads run four float4 (16 rows) ahead of the adds
    float* colp = tile + threadIdx.x * ldh;
    float4 v = *reinterpret_cast<float4*>(colp);
    v.y = v.x + v.y;
    v.z = v.y + v.z;
    v.w = v.z + v.w;
    *reinterpret_cast<float4*>(colp) = v;
    float s = v.w;
    constexpr int G = 4;
    float4 a[G], bq[G];
    int r = 4;
    const int R4 = R & ~3;
    const int lastr = R4 - 4;
#pragma unroll
    for (int k = 0; k < G; ++k) a[k] = *reinterpret_cast<float4*>(colp + min(r + 4 * k, lastr));
    for (; r + 8 * G <= R4; r += 8 * G) {
#pragma unroll
      for (int k = 0; k < G; ++k) bq[k] = *reinterpret_cast<float4*>(colp + r + 4 * G + 4 * k);
#pragma unroll
      for (int k = 0; k < G; ++k) {
        float4 w = a[k];
        w.x = s + w.x;
        w.y = w.x + w.y;
        w.z = w.y + w.z;
        w.w = w.z + w.w;
        s = w.w;
        *reinterpret_cast<float4*>(colp + r + 4 * k) = w;
      }
#pragma unroll
      for (int k = 0; k < G; ++k) a[k] = *reinterpret_cast<float4*>(colp + min(r + 8 * G + 4 * k, lastr));
#pragma unroll
      for (int k = 0; k < G; ++k) {
        float4 w = bq[k];
        w.x = s + w.x;
        w.y = w.x + w.y;
        w.z = w.y + w.z;
        w.w = w.z + w.w;
        s = w.w;
        *reinterpret_cast<float4*>(colp + r + 4 * G + 4 * k) = w;
      }
    }
    for (; r < R4; r += 4) {
      float4 w = *reinterpret_cast<float4*>(colp + r);
      w.x = s + w.x;
      w.y = w.x + w.y;
      w.z = w.y + w.z;
      w.w = w.z + w.w;
      s = w.w;
      *reinterpret_cast<float4*>(colp + r) = w;
    }
    for (; r < R; ++r) {  // rows % 4 tail
      s = s + colp[r];
      colp[r] = s;
    }
  }
  __syncthreads();
  float* gdst = buf + c0 + c4 * 4;
  for (int base = 0; base < R; base += kRowsPerPass * kBatch) {
#pragma unroll
    for (int k = 0; k < kBatch; ++k) {
      const int r = base + k * kRowsPerPass + rb;
      if (r < R && cvalid) {
        const float* t = tile + (c4 * 4) * ldh + r;
        *reinterpret_cast<float4*>(gdst + (size_t)r * C) = make_float4(t[0], t[ldh], t[2 * ldh], t[3 * ldh]);
      }
    }
  }
}

__global__ __launch_bounds__(256) void k_colscan(float* __restrict__ buf0, float* __restrict__ buf1, int R, int C, int ldh, int* __restrict__ zero,
                                                 int nzero) {
  colscan_body(buf0, buf1, R, C, ldh, zero, nzero);
}
// batched form. which 0: a[0] alone; 1: b[0], b[1]; 2: a[0], a[1]. zero_parity >= 0: clears the lane's row counters of that parity
__global__ __launch_bounds__(256) void k_colscan_b(const LaneStatic* __restrict__ ls, int lane0, int which, int R, int C, int ldh, int zero_parity) {
  const LaneStatic& L = ls[lane0 + blockIdx.z];
  if (which == 3) {  // the third pass's images of the fused path (k_rowscan_b stage 4), row counters of the same parity cleared
    colscan_body(gptr(L.dog2[zero_parity]), gptr(L.mag2[zero_parity]), R, C, ldh, gptr(L.rowcount2[zero_parity]), R);
    return;
  }
  colscan_body(which == 1 ? gptr(L.sb[0]) : gptr(L.sa[0]), which == 0 ? gptr(L.sa[0]) : (which == 1 ? gptr(L.sb[1]) : gptr(L.sa[1])), R, C, ldh,
               zero_parity >= 0 ? gptr(L.rowcount2[zero_parity]) : nullptr, R);
}

// ---- last box pass of both filters fused with DoG and squared gradient (scale_space.cpp:210-233) ------
// Tile of the last box pass + DoG + gradient: 64 columns x kDogRows rows per workgroup (256 threads, four rows each). The
// two integral images are staged in LDS once per tile - the rows and columns the tile's box sums touch: (kDogRows + 2 + d)
// x (64 + 2 + d) for scale 0 (its 3 x 3 gradient stencil needs a one-pixel ring of averages), (kDogRows + d) x (64 + d)
// for scale 1 - and every box average takes its four corners from there: ~3.6 global loads per pixel instead of ~10 with the
// corners fetched per pixel (PMC, 8-lane batch: 2 x FETCH_SIZE 88 MB per launch against 39 MB of integral images).
// box_avg is handed the LDS tile as if it were the whole image (base pointer shifted by the tile's origin, LDS pitch): the
// same border cases, the same operand order, the same bits.
constexpr int kDogMaxD = 11;                                   // box widths 3..11 (rebvio_hip_create checks)
constexpr int kDogPitch = 64 + 2 + kDogMaxD + 1;               // 78 (even pitch, rows staged with consecutive lanes: conflict-free)
// kDogRows: see kTileRowsSingle
template <int kDogRows>
__device__ __forceinline__ void dog_mag_body(const float* __restrict__ II0, const float* __restrict__ II1, int d0,
                                                 int d1, float* __restrict__ dog, float* __restrict__ mag,
                                                 float* __restrict__ scale0, float* __restrict__ scale1, int R, int C,
                                                 int* __restrict__ rowcount) {
  const uint2 vb = xcd_band_block();  // tile coordinates: contiguous bands of tiles per XCD (see xcd_band_block)
  constexpr int kDogTileRows = kDogRows + 2 + kDogMaxD;
  const int ld = (C + 3) & ~3;  // pitch of the integral images
  __shared__ float t0[kDogTileRows * kDogPitch];
  __shared__ float t1[kDogTileRows * kDogPitch];
  __shared__ float s0[kDogRows + 2][66];
  const int c0 = vb.x * 64, r0 = vb.y * kDogRows;
  const int h0 = d0 >> 1, h1 = d1 >> 1;
  // scale 0: averages at rows r0-1 .. r0+kDogRows, columns c0-1 .. c0+64 -> integral rows r0-2-h0 .. r0+kDogRows+h0
  // (staging: wave w takes tile rows w, w + 4, ..., its lanes the row's columns lane, lane + 64 - consecutive lanes read and
  // write consecutive words, and no index needs a division by the run-time tile width)
  const int ro0 = r0 - 2 - h0, co0 = c0 - 2 - h0, nr0 = kDogRows + 3 + 2 * h0, nc0 = 64 + 3 + 2 * h0;
  const int tx = threadIdx.x, ty = threadIdx.y;
  // scale 1: averages at the tile's own pixels -> integral rows r0-1-h1 .. r0+kDogRows-1+h1
  const int ro1 = r0 - 1 - h1, co1 = c0 - 1 - h1, nr1 = kDogRows + 1 + 2 * h1, nc1 = 64 + 1 + 2 * h1;
  // every load of both tiles is issued before the first LDS write (loops of constant length over registers: with the tile
  // heights as run-time trip counts the compiler waits for each row's load before it issues the next - seven dependent round
  // trips per tile)
  constexpr int kIt = (kDogRows + 3 + kDogMaxD - 1 + 3) / 4;  // rows per wave, both tiles (nr0 <= kDogRows + 3 + kDogMaxD - 1)
  float va[kIt], vb2[kIt], wa[kIt], wb[kIt];
  const int ca0 = min(max(co0 + tx, 0), C - 1), cb0 = min(max(co0 + tx + 64, 0), C - 1);
  const int ca1 = min(max(co1 + tx, 0), C - 1), cb1 = min(max(co1 + tx + 64, 0), C - 1);
#pragma unroll
  for (int k = 0; k < kIt; ++k) {
    const int lr = ty + 4 * k;
    const float* __restrict__ row0 = II0 + (size_t)min(max(ro0 + lr, 0), R - 1) * ld;  // (clamped positions are never used)
    const float* __restrict__ row1 = II1 + (size_t)min(max(ro1 + lr, 0), R - 1) * ld;
    va[k] = row0[ca0];
    vb2[k] = row0[cb0];
    wa[k] = row1[ca1];
    wb[k] = row1[cb1];
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int k = 0; k < kIt; ++k) {
    const int lr = ty + 4 * k;
    if (lr < nr0) {
      t0[lr * kDogPitch + tx] = va[k];
      if (tx + 64 < nc0) t0[lr * kDogPitch + tx + 64] = vb2[k];
    }
    if (lr < nr1) {
      t1[lr * kDogPitch + tx] = wa[k];
      if (tx + 64 < nc1) t1[lr * kDogPitch + tx + 64] = wb[k];
    }
  }
  __syncthreads();
  // (pointer arithmetic only: box_avg indexes [r * pitch + c] with image coordinates)
  const float* v0 = t0 - ((ptrdiff_t)ro0 * kDogPitch + co0);
  const float* v1 = t1 - ((ptrdiff_t)ro1 * kDogPitch + co1);
  // A tile none of whose box sums touches the image border (3 of 4 at 640x480) takes box_avg's interior case directly: the
  // same four corners in the same operand order times the same reciprocal (scale_space.cpp:121-126), without the nine-way
  // border logic and the per-pixel table lookup. Uniform per workgroup.
  const bool in0 = r0 - 1 >= h0 + 1 && r0 + kDogRows < R - h0 && c0 - 1 >= h0 + 1 && c0 + 64 < C - h0;
  const bool in1 = r0 >= h1 + 1 && r0 + kDogRows - 1 < R - h1 && c0 >= h1 + 1 && c0 + 63 < C - h1;
  const float rc0 = c_recip[d0 * d0], rc1 = c_recip[d1 * d1];
  for (int lr = ty; lr < kDogRows + 2; lr += 4) {
    const int r = r0 + lr - 1;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const int lc = tx + 64 * half;
      if (lc >= 66) break;
      const int c = c0 + lc - 1;
      float a;
      if (in0) {
        const float* q1 = v0 + (r + h0) * kDogPitch + c;
        const float* q2 = v0 + (r - h0 - 1) * kDogPitch + c;
        a = (((q1[h0] - q1[-h0 - 1]) - q2[h0]) + q2[-h0 - 1]) * rc0;
      } else {
        a = (r >= 0 && r < R && c >= 0 && c < C) ? box_avg(v0, r, c, d0, R, C, kDogPitch) : 0.0f;
      }
      s0[lr][lc] = a;
    }
  }
  __syncthreads();
  const int c = c0 + threadIdx.x;
  if (c >= C) return;
#pragma unroll
  for (int k = 0; k < kDogRows / 4; ++k) {
    const int lr = threadIdx.y + 4 * k, r = r0 + lr;
    if (r >= R) break;
    if (c == 0) rowcount[r] = 0;  // reset for the candidate kernel of this frame
    const float a0 = s0[lr + 1][threadIdx.x + 1];
    float a1;
    if (in1) {
      const float* q1 = v1 + (r + h1) * kDogPitch + c;
      const float* q2 = v1 + (r - h1 - 1) * kDogPitch + c;
      a1 = (((q1[h1] - q1[-h1 - 1]) - q2[h1]) + q2[-h1 - 1]) * rc1;
    } else {
      a1 = box_avg(v1, r, c, d1, R, C, kDogPitch);
    }
    const size_t i = (size_t)r * C + c;
    dog[i] = a1 - a0;
    float m = 0.0f;
    if (r >= 1 && r < R - 1 && c >= 1 && c < C - 1) {
      const float dx = s0[lr + 1][threadIdx.x + 2] - s0[lr + 1][threadIdx.x];
      const float dy = s0[lr + 2][threadIdx.x + 1] - s0[lr][threadIdx.x + 1];
      m = dx * dx + dy * dy;
    }
    mag[i] = m;
    if (scale0) scale0[i] = a0;
    if (scale1) scale1[i] = a1;
  }
}

template <int TR>
__global__ __launch_bounds__(256) void k_dog_mag(const float* __restrict__ II0, const float* __restrict__ II1, int d0, int d1,
                                                 float* __restrict__ dog, float* __restrict__ mag, float* __restrict__ scale0,
                                                 float* __restrict__ scale1, int R, int C, int* __restrict__ rowcount) {
  dog_mag_body<TR>(II0, II1, d0, d1, dog, mag, scale0, scale1, R, C, rowcount);
}
__global__ __launch_bounds__(256) void k_dog_mag_b(const LaneStatic* __restrict__ ls, LaneDynB dyn, int lane0, int d0, int d1, int R, int C) {
  const LaneStatic& L = ls[lane0 + blockIdx.z];
  const int b = dyn.v[lane0 + blockIdx.z].parity;
  dog_mag_body<16>(gptr(L.sa[0]), gptr(L.sa[1]), d0, d1, gptr(L.dog2[b]), gptr(L.mag2[b]), nullptr, nullptr, R, C, gptr(L.rowcount2[b]));
}

// Threshold servo of EdgeDetector::detect (edge_detector.cpp:33-36), evaluated identically by every thread.
__device__ __forceinline__ float servo_threshold(const KParams& p, const DetState& d) {
  float t = d.threshold;
  if (p.gain > 0) {
    t -= p.gain * float(p.kref - d.count);
    t = (t > p.max_threshold) ? p.max_threshold : ((t < p.min_threshold) ? p.min_threshold : t);
  }
  return t;
}

// ---- candidate test + plane fit (edge_detector.cpp:73-107) ---------------------------------------------
// Tile = kFlagRows rows x 64 columns per workgroup of four waves; a wavefront works on one 64-pixel row segment at a time
// (rows wave, wave + 4, ...), so a __ballot is exactly the raster-ordered candidate set of that segment. The DoG tile with
// its two-pixel ring is staged once: (kFlagRows + 4) x 68 values for kFlagRows x 64 pixels (1.33 x; the 4-row tile of the
// first rounds staged 2.1 x and had a quarter of the loads in flight per thread).
template <int kFlagRows>
__device__ __forceinline__ void keyline_flag_body(const float* __restrict__ dog, const float* __restrict__ mag,
                                                      KParams p, const DetState* __restrict__ det_in,
                                                      float4* __restrict__ stash, unsigned long long* __restrict__ bits,
                                                      int* __restrict__ rowcount) {
  const uint2 vb = xcd_band_block();  // tile coordinates: contiguous bands of tiles per XCD (see xcd_band_block)
  __shared__ float sd[kFlagRows + 4][68];
  const int R = p.rows, C = p.cols;
  const int c0 = vb.x * 64, r0 = vb.y * kFlagRows;
  const int tid = threadIdx.y * 64 + threadIdx.x;
  for (int i = tid; i < (kFlagRows + 4) * 68; i += 256) {
    const int lr = i / 68, lc = i - lr * 68;
    const int r = r0 + lr - 2, c = c0 + lc - 2;
    sd[lr][lc] = (r >= 0 && r < R && c >= 0 && c < C) ? dog[(size_t)r * C + c] : 0.0f;
  }
  const int c = c0 + threadIdx.x;
  float mgv[kFlagRows / 4];  // the gradient magnitudes of this thread's pixels: in flight while the tile lands
#pragma unroll
  for (int k = 0; k < kFlagRows / 4; ++k) {
    const int r = r0 + threadIdx.y + 4 * k;
    mgv[k] = (r >= 2 && r < R - 2 && c >= 2 && c < C - 2) ? mag[(size_t)r * C + c] : 0.0f;
  }
  __syncthreads();
  const float thr = servo_threshold(p, *det_in);
  const float pn_threshold = float((2.0 * 2 + 1.0) * (2.0 * 2 + 1.0)) * p.pos_neg_threshold;
  const float gradient_threshold_squared = (thr * kMaxImageValue * p.dog_threshold) * (thr * kMaxImageValue * p.dog_threshold);
  const float mag_threshold = (thr * kMaxImageValue) * (thr * kMaxImageValue);
#pragma unroll
  for (int k = 0; k < kFlagRows / 4; ++k) {
    const int lr = threadIdx.y + 4 * k, r = r0 + lr;
    if (r >= R) break;  // (whole wave)
    bool cand = false;
    float4 fit = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r >= 2 && r < R - 2 && c >= 2 && c < C - 2) {
      const float mg = mgv[k];
      if (!(mg < mag_threshold)) {
        int pn = 0;
        float t0 = 0.f, t1 = 0.f, t2 = 0.f;
#pragma unroll
        for (int q = 0; q < 25; ++q) {
          const float y = sd[lr + q / 5][threadIdx.x + q % 5];
          pn = (y > 0.0f) ? pn + 1 : pn - 1;
          t0 += c_pinv[q] * y;
          t1 += c_pinv[25 + q] * y;
          t2 += c_pinv[50 + q] * y;
        }
        if (!(fabsf((float)pn) > pn_threshold)) {
          const float g2 = t0 * t0 + t1 * t1;
          const float tmp = t2 / g2;
          const float xs = -t0 * tmp;
          const float ys = -t1 * tmp;
          if (!(fabsf(xs) > 0.5f || fabsf(ys) > 0.5f)) {
            if (!(g2 < gradient_threshold_squared)) {
              cand = true;
              fit = make_float4(t0, t1, xs, ys);
            }
          }
        }
      }
    }
    const unsigned long long b = __ballot(cand);
    if (c < C && cand) stash[(size_t)r * C + c] = fit;
    if (threadIdx.x == 0) {
      bits[(size_t)r * p.nseg + vb.x] = b;
      const int n = __popcll(b);
      if (n) atomicAdd(&rowcount[r], n);
    }
  }
}

template <int TR>
__global__ __launch_bounds__(256) void k_keyline_flag(const float* __restrict__ dog, const float* __restrict__ mag, KParams p,
                                                      const DetState* __restrict__ det_in, float4* __restrict__ stash,
                                                      unsigned long long* __restrict__ bits, int* __restrict__ rowcount) {
  keyline_flag_body<TR>(dog, mag, p, det_in, stash, bits, rowcount);
}
__global__ __launch_bounds__(256) void k_keyline_flag_b(KParams p, const LaneStatic* __restrict__ ls, LaneDynB dyn) {
  const LaneStatic& L = ls[blockIdx.z];
  const LaneDyn d = dyn.v[blockIdx.z];
  keyline_flag_body<16>(gptr(L.dog2[d.parity]), gptr(L.mag2[d.parity]), p, gptr(L.det) + d.det_in, gptr(L.stash), gptr(L.bits), gptr(L.rowcount2[d.parity]));
}

// ---- candidate test + plane fit straight from the integral images: the last box pass, DoG and squared gradient of the tile
// (k_dog_mag) are formed in LDS instead of travelling through two per-pixel arrays and a launch of their own ----------------
// Tile = kFlagRows x 64 pixels; DoG is needed with a two-pixel ring (5 x 5 plane fit), the scale-0 average with the same ring
// (its 3 x 3 gradient stencil sits inside it): both box averages are evaluated on (kFlagRows + 4) x 68 pixels from the two
// integral-image tiles staged as in dog_mag_body - the same corners, operand order and reciprocals, hence the same DoG and
// gradient bits as the unfused kernels, and from there on keyline_flag_body's statements.
template <int kFlagRows>
__device__ __forceinline__ void keyline_flag_ii_body(const float* __restrict__ II0, const float* __restrict__ II1, int d0, int d1, KParams p,
                                                     const DetState* __restrict__ det_in, float4* __restrict__ stash,
                                                     unsigned long long* __restrict__ bits, int* __restrict__ rowcount) {
  const uint2 vb = xcd_band_block();
  constexpr int kReg = kFlagRows + 4;              // rows of the DoG region
  constexpr int kTRows = kReg + kDogMaxD;          // integral-image tile: kReg + 2 h + 1 rows, h <= 5
  constexpr int kPitch = 68 + kDogMaxD + 1;        // 80 >= 68 + 2 h + 1 columns
  __shared__ float t0[kTRows * kPitch];
  __shared__ float t1[kTRows * kPitch];
  __shared__ float sa[kReg][68];  // scale-0 averages
  __shared__ float sd[kReg][68];  // DoG
  const int R = p.rows, C = p.cols;
  const int ld = (C + 3) & ~3;  // pitch of the integral images
  const int c0 = vb.x * 64, r0 = vb.y * kFlagRows;
  const int h0 = d0 >> 1, h1 = d1 >> 1;
  const int tx = threadIdx.x, ty = threadIdx.y;
  // region rows r0-2 .. r0+kFlagRows+1, columns c0-2 .. c0+65 -> integral rows / columns from (r0-2) - h - 1, (c0-2) - h - 1
  const int ro0 = r0 - 3 - h0, co0 = c0 - 3 - h0, nr0 = kReg + 2 * h0 + 1, nc0 = 68 + 2 * h0 + 1;
  const int ro1 = r0 - 3 - h1, co1 = c0 - 3 - h1, nr1 = kReg + 2 * h1 + 1, nc1 = 68 + 2 * h1 + 1;
  constexpr int kIt = (kTRows + 3) / 4;  // tile rows per wave (every load of both tiles in flight before the first LDS write)
  float va[kIt], vb2[kIt], wa[kIt], wb[kIt];
  const int ca0 = min(max(co0 + tx, 0), C - 1), cb0 = min(max(co0 + tx + 64, 0), C - 1);
  const int ca1 = min(max(co1 + tx, 0), C - 1), cb1 = min(max(co1 + tx + 64, 0), C - 1);
#pragma unroll
  for (int k = 0; k < kIt; ++k) {
    const int lr = ty + 4 * k;
    const float* __restrict__ row0 = II0 + (size_t)min(max(ro0 + lr, 0), R - 1) * ld;  // (clamped positions are never used)
    const float* __restrict__ row1 = II1 + (size_t)min(max(ro1 + lr, 0), R - 1) * ld;
    va[k] = row0[ca0];
    vb2[k] = row0[cb0];
    wa[k] = row1[ca1];
    wb[k] = row1[cb1];
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int k = 0; k < kIt; ++k) {
    const int lr = ty + 4 * k;
    if (lr < nr0) {
      t0[lr * kPitch + tx] = va[k];
      if (tx + 64 < nc0) t0[lr * kPitch + tx + 64] = vb2[k];
    }
    if (lr < nr1) {
      t1[lr * kPitch + tx] = wa[k];
      if (tx + 64 < nc1) t1[lr * kPitch + tx + 64] = wb[k];
    }
  }
  __syncthreads();
  // (pointer arithmetic only: box_avg indexes [r * pitch + c] with image coordinates)
  const float* v0 = t0 - ((ptrdiff_t)ro0 * kPitch + co0);
  const float* v1 = t1 - ((ptrdiff_t)ro1 * kPitch + co1);
  // a region none of whose box sums touches the image border takes box_avg's interior case directly (see dog_mag_body)
  const bool in0 = r0 - 2 >= h0 + 1 && r0 + kFlagRows + 1 < R - h0 && c0 - 2 >= h0 + 1 && c0 + 65 < C - h0;
  const bool in1 = r0 - 2 >= h1 + 1 && r0 + kFlagRows + 1 < R - h1 && c0 - 2 >= h1 + 1 && c0 + 65 < C - h1;
  const float rc0 = c_recip[d0 * d0], rc1 = c_recip[d1 * d1];
  for (int lr = ty; lr < kReg; lr += 4) {
    const int r = r0 + lr - 2;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const int lc = tx + 64 * half;
      if (lc >= 68) break;
      const int c = c0 + lc - 2;
      float a0 = 0.0f, dg = 0.0f;
      if (in0 && in1) {
        const float* q1 = v0 + (r + h0) * kPitch + c;
        const float* q2 = v0 + (r - h0 - 1) * kPitch + c;
        a0 = (((q1[h0] - q1[-h0 - 1]) - q2[h0]) + q2[-h0 - 1]) * rc0;
        const float* s1 = v1 + (r + h1) * kPitch + c;
        const float* s2 = v1 + (r - h1 - 1) * kPitch + c;
        const float a1 = (((s1[h1] - s1[-h1 - 1]) - s2[h1]) + s2[-h1 - 1]) * rc1;
        dg = a1 - a0;
      } else if (r >= 0 && r < R && c >= 0 && c < C) {
        a0 = box_avg(v0, r, c, d0, R, C, kPitch);
        dg = box_avg(v1, r, c, d1, R, C, kPitch) - a0;
      }
      sa[lr][lc] = a0;
      sd[lr][lc] = dg;
    }
  }
  __syncthreads();
  const int c = c0 + tx;
  const float thr = servo_threshold(p, *det_in);
  const float pn_threshold = float((2.0 * 2 + 1.0) * (2.0 * 2 + 1.0)) * p.pos_neg_threshold;
  const float gradient_threshold_squared = (thr * kMaxImageValue * p.dog_threshold) * (thr * kMaxImageValue * p.dog_threshold);
  const float mag_threshold = (thr * kMaxImageValue) * (thr * kMaxImageValue);
#pragma unroll
  for (int k = 0; k < kFlagRows / 4; ++k) {
    const int lr = ty + 4 * k, r = r0 + lr;
    if (r >= R) break;  // (whole wave)
    bool cand = false;
    float4 fit = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r >= 2 && r < R - 2 && c >= 2 && c < C - 2) {
      // calculateGradientMagnitude (scale_space.cpp:221-232) of this pixel
      const float dx = sa[lr + 2][tx + 3] - sa[lr + 2][tx + 1];
      const float dy = sa[lr + 3][tx + 2] - sa[lr + 1][tx + 2];
      const float mg = dx * dx + dy * dy;
      if (!(mg < mag_threshold)) {
        int pn = 0;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int q = 0; q < 25; ++q) {
          const float y = sd[lr + q / 5][tx + q % 5];
          pn = (y > 0.0f) ? pn + 1 : pn - 1;
          s0 += c_pinv[q] * y;
          s1 += c_pinv[25 + q] * y;
          s2 += c_pinv[50 + q] * y;
        }
        if (!(fabsf((float)pn) > pn_threshold)) {
          const float g2 = s0 * s0 + s1 * s1;
          const float tmp = s2 / g2;
          const float xs = -s0 * tmp;
          const float ys = -s1 * tmp;
          if (!(fabsf(xs) > 0.5f || fabsf(ys) > 0.5f)) {
            if (!(g2 < gradient_threshold_squared)) {
              cand = true;
              fit = make_float4(s0, s1, xs, ys);
            }
          }
        }
      }
    }
    const unsigned long long b = __ballot(cand);
    if (c < C && cand) stash[(size_t)r * C + c] = fit;
    if (tx == 0) {
      bits[(size_t)r * p.nseg + vb.x] = b;
      const int n = __popcll(b);
      if (n) atomicAdd(&rowcount[r], n);
    }
  }
}

template <int TR>
__global__ __launch_bounds__(256) void k_keyline_flag_ii(const float* __restrict__ II0, const float* __restrict__ II1, int d0, int d1, KParams p,
                                                         const DetState* __restrict__ det_in, float4* __restrict__ stash,
                                                         unsigned long long* __restrict__ bits, int* __restrict__ rowcount) {
  keyline_flag_ii_body<TR>(II0, II1, d0, d1, p, det_in, stash, bits, rowcount);
}
__global__ __launch_bounds__(256) void k_keyline_flag_ii_b(KParams p, const LaneStatic* __restrict__ ls, LaneDynB dyn, int d0, int d1) {
  const LaneStatic& L = ls[blockIdx.z];
  const LaneDyn d = dyn.v[blockIdx.z];
  keyline_flag_ii_body<16>(gptr(L.dog2[d.parity]), gptr(L.mag2[d.parity]), d0, d1, p, gptr(L.det) + d.det_in, gptr(L.stash), gptr(L.bits),
                           gptr(L.rowcount2[d.parity]));
}

__device__ __forceinline__ int wave_sum(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// tuneThreshold (edge_detector.cpp:167-186). With size() <= keylines_max the cumulative loop can never
// reach keylines_max (bin 0 always holds the maximum), so it always ends at i = num_bins and the result
// depends on min/max only; the histogram itself is dead code there.
__device__ __forceinline__ float auto_threshold_from(const MapState& st, float previous) {
  if (st.n <= 0) return previous;
  const float max_dog = __uint_as_float(st.gmax_bits);
  const float min_dog = __uint_as_float(st.gmin_bits);
  return max_dog - float(kNumBins * (max_dog - min_dog)) / float(kNumBins);
}

// ---- ordered emission (edge_detector.cpp:109-119) -------------------------------------------------------
// rank = (#candidates in earlier rows) + (#candidates in earlier segments of this row) + (#lower lanes):
// the raster rank of the reference's sequential loop, with truncation at keylines_max. Also rewrites the
// whole dense mask and clears this map's distance-field cells.
// kEmitRows rows x 64 columns per workgroup: the tile's row bases are formed once, by the whole workgroup
template <int kEmitRows>
__device__ __forceinline__ void keyline_emit_body(KParams p, MapDev m, const float4* __restrict__ stash,
                                                      const unsigned long long* __restrict__ bits,
                                                      const int* __restrict__ rowcount, const DetState* __restrict__ det_in,
                                                      DetState* __restrict__ det_out, const MapState* prev_st,
                                                      int clear_df, int ntiles) {
  const uint2 vb = xcd_band_block();  // tile coordinates: contiguous bands of tiles per XCD (see xcd_band_block)
  const int R = p.rows, C = p.cols;
  const int r0 = vb.y * kEmitRows;
  const int lane = threadIdx.x, wid = threadIdx.y, tid = wid * 64 + lane;
  __shared__ int s_rc[kEmitRows];    // candidates per row of the tile
  __shared__ int s_seg[kEmitRows];   // ... of each row in the segments left of this one
  __shared__ int s_base[kEmitRows];  // raster rank of the row's first candidate in THIS segment
  __shared__ int s_pw[4];            // candidates in the rows above the tile, one partial sum per wave
  // rows above the tile: the whole workgroup, four loads per thread in flight together (one memory round trip per 1024 rows;
  // a loop with the row count as its trip count waits for every load before it issues the next: up to eight dependent round
  // trips at 480 rows on the path of every workgroup); the tile's own rows and the segments to the left: sixteen lanes per row
  {
    int part = 0;
    for (int base = 0; base < r0; base += 1024) {
      const int i0 = base + tid, i1 = i0 + 256, i2 = i0 + 512, i3 = i0 + 768;
      const int v0 = i0 < r0 ? rowcount[i0] : 0, v1 = i1 < r0 ? rowcount[i1] : 0;
      const int v2 = i2 < r0 ? rowcount[i2] : 0, v3 = i3 < r0 ? rowcount[i3] : 0;
      part += (v0 + v1) + (v2 + v3);
    }
    const int tot = wave_sum(part);
    if (lane == 0) s_pw[wid] = tot;
  }
  {
    const int lr = tid >> 4, sg = tid & 15, r = r0 + lr;
    int acc = 0;
    if (lr < kEmitRows && r < R)
      for (int q = sg; q < (int)vb.x; q += 16) acc += __popcll(bits[(size_t)r * p.nseg + q]);
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) acc += __shfl_xor(acc, o);  // (the sixteen lanes of a row are neighbours in one wave)
    if (sg == 0 && lr < kEmitRows) {
      s_seg[lr] = acc;
      s_rc[lr] = (r < R) ? rowcount[r] : 0;
    }
  }
  {  // the distance-field tile counters of this map start at zero (k_join_edges bins into them)
    const int wg = vb.y * gridDim.x + vb.x, t = wg * 256 + tid;
    if (t < ntiles) m.tile_cnt[t] = 0;
  }
  __syncthreads();
  if (tid < kEmitRows) {
    int base = (s_pw[0] + s_pw[1]) + (s_pw[2] + s_pw[3]);
    for (int j = 0; j < tid; ++j) base += s_rc[j];
    s_base[tid] = base + s_seg[tid];
    // raster rank of the row's first keyline (segment 0: the rows above only), clamped like the ranks below
    if (vb.x == 0 && r0 + tid < R) m.row_start[r0 + tid] = min(base, p.kmax);
  }
  if (vb.y == 0 && vb.x == 0 && wid == 0) {  // one wave publishes the frame's scalars
    int tp = 0;
    for (int base = 0; base < R; base += 256) {  // (four loads in flight per lane, as above)
      const int i0 = base + lane, i1 = i0 + 64, i2 = i0 + 128, i3 = i0 + 192;
      const int v0 = i0 < R ? rowcount[i0] : 0, v1 = i1 < R ? rowcount[i1] : 0;
      const int v2 = i2 < R ? rowcount[i2] : 0, v3 = i3 < R ? rowcount[i3] : 0;
      tp += (v0 + v1) + (v2 + v3);
    }
    const int total = wave_sum(tp);
    if (lane == 0) {
      // auto_threshold_ as left by the previous detect's tuneThreshold (its min/max are final: same stream). Read before
      // this map's scalars are reset: the pool may hand the previous frame's map back for this one (prev_st == m.st).
      const float prev_auto = prev_st ? auto_threshold_from(*prev_st, det_in->auto_threshold) : det_in->auto_threshold;
      const int n = min(total, p.kmax);
      m.row_start[R] = n;
      m.st->n = n;
      m.st->total = total;
      m.st->gmin_bits = 0x7F800000u;
      m.st->gmax_bits = 0u;
      m.st->dm_matches = 0;
      m.st->dm_kf = 0;
      m.st->reg_count = 0;
      m.st->dm_queued = 0;
      det_out->threshold = servo_threshold(p, *det_in);
      det_out->count = n;
      det_out->auto_threshold = prev_auto;
    }
  }
  __syncthreads();
  const int c = vb.x * 64 + lane;
  if (c >= C) return;
#pragma unroll
  for (int k = 0; k < kEmitRows / 4; ++k) {
    const int lr = wid + 4 * k, r = r0 + lr;
    if (r >= R) break;  // (whole wave)
    const unsigned long long b = bits[(size_t)r * p.nseg + vb.x];
    const bool cand = (b >> lane) & 1ull;
    const int rank = s_base[lr] + __popcll(b & ((1ull << lane) - 1ull));
    const size_t pix = (size_t)r * C + c;
    int mk = -1;
    if (cand && rank < p.kmax) {
      mk = rank;
      const float4 fit = stash[pix];
      const float px = float(c) + fit.z, py = float(r) + fit.w;
      m.pos[rank] = make_float2(px, py);
      const float2 pi = make_float2(px - p.cx, py - p.cy);
      m.pos_img[rank] = pi;
      m.mpos_img[rank] = pi;
      m.grad[rank] = make_float2(fit.x, fit.y);
      m.mgrad[rank] = make_float2(0.f, 0.f);
      m.gnorm[rank] = sqrtf(fit.x * fit.x + fit.y * fit.y);
      m.mgnorm[rank] = 0.f;
      m.rs[rank] = make_float2(1.0f, 20.0f);
      m.id_prev[rank] = -1;
      m.id_next[rank] = -1;
      m.match_id[rank] = -1;
      m.match_fwd[rank] = -1;
      m.match_kf[rank] = -1;
      m.matches[rank] = 0u;
      m.fwd_key[rank] = 0ull;
      m.residual[rank] = 0.f;
    }
    m.mask[pix] = mk;
    if (clear_df) m.df[pix] = kDfEmpty;  // only the scatter build needs a cleared field, the tiled build writes every cell
  }
}

// ---- joinEdges (edge_detector.cpp:125-165) + min/max of gradient_norm for tuneThreshold (:168-174) -----
template <int TR>
__global__ __launch_bounds__(256) void k_keyline_emit(KParams p, MapDev m, const float4* __restrict__ stash,
                                                      const unsigned long long* __restrict__ bits, const int* __restrict__ rowcount,
                                                      const DetState* __restrict__ det_in, DetState* __restrict__ det_out,
                                                      const MapState* prev_st, int clear_df, int ntiles) {
  keyline_emit_body<TR>(p, m, stash, bits, rowcount, det_in, det_out, prev_st, clear_df, ntiles);
}
__global__ __launch_bounds__(256) void k_keyline_emit_b(KParams p, const LaneStatic* __restrict__ ls, const MapDev* __restrict__ maptab,
                                                        LaneDynB dyn, int clear_df, int ntiles) {
  const LaneStatic& L = ls[blockIdx.z];
  const LaneDyn d = dyn.v[blockIdx.z];
  const MapDev m = global_map(lane_map(maptab, blockIdx.z, d.nm, d.nm_swap));
  const MapState* prev = d.prev >= 0 ? gptr(maptab[blockIdx.z * kLaneMaps + d.prev].st) : nullptr;
  keyline_emit_body<16>(p, m, gptr(L.stash), gptr(L.bits), gptr(L.rowcount2[d.parity]), gptr(L.det) + d.det_in, gptr(L.det) + d.det_out, prev, clear_df, ntiles);
}

// r-range of a keyline's probe segment (cells round(pos + u r), r in [-half, half)) that can fall into the pixel box
// [x0, x1) x [y0, y1): conservative (one cell of slack on each side, the exact test per cell is made by whoever walks the
// range), so the two reciprocals are the 1-ulp hardware ones. Returns false when no cell can fall into the box.
__device__ __forceinline__ bool df_clip_range(float2 pos, float2 u, int half, int x0, int x1, int y0, int y1, int* r_lo, int* r_hi) {
  const float fx0 = (float)x0 - 1.0f, fx1 = (float)x1, fy0 = (float)y0 - 1.0f, fy1 = (float)y1;
  float lo = -(float)half, hi = (float)(half - 1);
  if (fabsf(u.x) > 1e-6f) {
    const float iv = __builtin_amdgcn_rcpf(u.x);
    const float a = (fx0 - pos.x) * iv, b = (fx1 - pos.x) * iv;
    lo = fmaxf(lo, fminf(a, b));
    hi = fminf(hi, fmaxf(a, b));
  } else if (pos.x < fx0 || pos.x > fx1) {
    hi = lo - 1.0f;
  }
  if (fabsf(u.y) > 1e-6f) {
    const float iv = __builtin_amdgcn_rcpf(u.y);
    const float a = (fy0 - pos.y) * iv, b = (fy1 - pos.y) * iv;
    lo = fmaxf(lo, fminf(a, b));
    hi = fminf(hi, fmaxf(a, b));
  } else if (pos.y < fy0 || pos.y > fy1) {
    hi = lo - 1.0f;
  }
  *r_lo = max((int)floorf(lo) - 1, -half);
  *r_hi = min((int)ceilf(hi) + 1, half - 1);
  return hi >= lo && *r_hi >= *r_lo;
}

// joinEdges also prepares the distance field of this map (core.hpp:37-59): it leaves every keyline's unit gradient and
// bins the keyline into the T x T tiles its +-search_range segment crosses (a segment of 80 cells crosses ~3.5 tiles of
// 32 x 32). A workgroup's 256 keylines are neighbours in raster order, so they share tiles: counts are aggregated in an
// LDS table, one returning global atomic per touched tile and workgroup reserves the slots, and the crossing test is
// simply evaluated twice (count, then place) instead of keeping per-thread tile lists.
__device__ __forceinline__ void join_edges_body(KParams p, MapDev m, int T, int ntx, int nty) {
  const uint2 vb = xcd_band_block();  // which keylines / tiles this workgroup takes: contiguous bands per XCD (common.hpp)
  extern __shared__ int t_cnt[];  // [ntx * nty] count, then cursor, of this workgroup's keylines per tile
  const int n = m.st->n;
  const int idx = vb.x * 256 + threadIdx.x;
  const int C = p.cols;
  const int ntiles = ntx * nty;
  const bool live_wg = (int)vb.x * 256 < n;
  if (live_wg)
    for (int t = threadIdx.x; t < ntiles; t += 256) t_cnt[t] = 0;
  unsigned gb = 0x7F800000u, gB = 0u;
  float2 b_pos = make_float2(0.f, 0.f), b_u = make_float2(0.f, 0.f);
  float b_gn = 0.f;
  int tx0 = 0, tx1 = -1, ty0 = 0, ty1 = -1;  // tile bounding box of the segment (empty for threads without a keyline)
  if (idx < n) {
    const float2 pos = m.pos[idx];
    const float2 g = m.grad[idx];
    const int x = cvtt_f64((double)pos.x + 0.5);
    const int y = cvtt_f64((double)pos.y + 0.5);
    const float tx = -g.y, ty = g.x;
    int dx1, dy2;  // probes: (y, x+dx1), (y+dy2, x), (y+dy2, x+dx1)
    if (ty > 0.0f) {
      dy2 = 1;
      dx1 = (tx > 0.0f) ? 1 : -1;
    } else {
      dy2 = -1;
      dx1 = (tx < 0.0f) ? -1 : 1;
    }
    int nxt = -1;
    if (x >= 1 && x < C - 1 && y >= 1 && y < p.rows - 1) {  // always true for detected keylines
      const int a = m.mask[(size_t)y * C + x + dx1];  // three independent loads, first hit in probe order wins
      const int b = m.mask[(size_t)(y + dy2) * C + x];
      const int c3 = m.mask[(size_t)(y + dy2) * C + x + dx1];
      nxt = (a >= 0) ? a : ((b >= 0) ? b : c3);
    }
    if (nxt >= 0) {
      atomicMax(&m.id_prev[nxt], idx);  // sequential last-writer-wins == largest index
      m.id_next[idx] = nxt;
    }
    const float gnv = m.gnorm[idx];
    gb = gB = __float_as_uint(gnv);
    // unit gradient of DistanceField::build (core.hpp:50-51: gradient / gradient_norm, IEEE division), once per keyline
    b_u = make_float2(g.x / gnv, g.y / gnv);
    m.unit[idx] = b_u;
    b_pos = pos;
    b_gn = gnv;
    // tiles the segment's bounding box touches (end points +- one cell for the rounding)
    const int half = p.df_nr >> 1;
    const float ex = fabsf(b_u.x) * (float)half + 1.5f, ey = fabsf(b_u.y) * (float)half + 1.5f;
    tx0 = max(cvtt_f32(floorf(pos.x - ex)), 0) / T;
    tx1 = min(cvtt_f32(floorf(pos.x + ex)), C - 1) / T;
    ty0 = max(cvtt_f32(floorf(pos.y - ey)), 0) / T;
    ty1 = min(cvtt_f32(floorf(pos.y + ey)), p.rows - 1) / T;
  }
  if (live_wg) {
    const int half = p.df_nr >> 1;
    __syncthreads();
    for (int ty = ty0; ty <= ty1; ++ty)
      for (int tx = tx0; tx <= tx1; ++tx) {
        int rl, rh;
        if (df_clip_range(b_pos, b_u, half, tx * T, min(tx * T + T, C), ty * T, min(ty * T + T, p.rows), &rl, &rh))
          atomicAdd(&t_cnt[ty * ntx + tx], 1);
      }
    __syncthreads();
    for (int t = threadIdx.x; t < ntiles; t += 256) {
      const int c = t_cnt[t];
      if (c) t_cnt[t] = atomicAdd(&m.tile_cnt[t], c);  // base of this workgroup's entries in the tile's list
    }
    __syncthreads();
    for (int ty = ty0; ty <= ty1; ++ty)
      for (int tx = tx0; tx <= tx1; ++tx) {
        int rl, rh;
        if (df_clip_range(b_pos, b_u, half, tx * T, min(tx * T + T, C), ty * T, min(ty * T + T, p.rows), &rl, &rh)) {
          const int slot = atomicAdd(&t_cnt[ty * ntx + tx], 1);
          if (slot < kDfTileCap) {  // the entry carries everything the tile kernel needs: no gather through the index
            float4* e = m.tile_list + ((size_t)(ty * ntx + tx) * kDfTileCap + slot) * 2;
            e[0] = make_float4(b_pos.x, b_pos.y, b_u.x, b_u.y);
            e[1] = make_float4(__int_as_float(idx), __int_as_float(((rl + 256) << 16) | (rh + 256)), b_gn, 0.f);
          }
        }
      }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    gb = min(gb, (unsigned)__shfl_xor((int)gb, o));
    gB = max(gB, (unsigned)__shfl_xor((int)gB, o));
  }
  // min / max of the gradient norm: wave -> workgroup (LDS) -> one global atomic pair per workgroup
  __shared__ unsigned b_min, b_max;
  __syncthreads();
  if (threadIdx.x == 0) {
    b_min = 0xFFFFFFFFu;
    b_max = 0u;
  }
  __syncthreads();
  if ((threadIdx.x & 63) == 0 && gB != 0u) {
    atomicMin(&b_min, gb);
    atomicMax(&b_max, gB);
  }
  __syncthreads();
  if (threadIdx.x == 0 && b_max != 0u) {
    atomicMin(&m.st->gmin_bits, b_min);
    atomicMax(&m.st->gmax_bits, b_max);
  }
}

__global__ __launch_bounds__(256) void k_join_edges(KParams p, MapDev m, int T, int ntx, int nty) { join_edges_body(p, m, T, ntx, nty); }
__global__ __launch_bounds__(256) void k_join_edges_b(KParams p, const MapDev* __restrict__ maptab, LaneDynB dyn, int T, int ntx, int nty) {
  const LaneDyn d = dyn.v[blockIdx.z];
  join_edges_body(p, global_map(lane_map(maptab, blockIdx.z, d.nm, d.nm_swap)), T, ntx, nty);
}

// ---- DistanceField::build (core.hpp:37-59) ---------------------------------------------------------------
// One thread per (keyline, r). Sequential semantics "smallest |r| wins, ties -> last (idx, r) visited" become
// an order-independent atomicMin on key = |r| << 23 | (2^23-1 - (idx*nr + r+range)).
__global__ __launch_bounds__(256) void k_df_build(KParams p, MapDev m, const DetState* __restrict__ det_prev) {
  const int n = m.st->n;
  const float thr = auto_threshold_from(*m.st, det_prev->auto_threshold);
  if (blockIdx.x == 0 && threadIdx.x == 0) m.st->threshold = thr;  // EdgeMap::threshold(auto_threshold_) (:185)
  // Persistent, deliberately small grid (grid-stride over the (keyline, r) pairs): this kernel is bound by the
  // memory-side atomic rate, not by CUs, and it runs on a low-priority stream beside the latency-critical tracking
  // kernels - it must not occupy every CU.
  const int nr = p.df_nr;
  const int total = n * nr;
  for (int gid = blockIdx.x * 256 + threadIdx.x; gid < total; gid += gridDim.x * 256) {
    const int idx = gid / nr;
    const int ri = gid - idx * nr;
    const int r = ri - (nr >> 1);
    const float gn = m.gnorm[idx];
    if (thr > 0.0f && gn < thr) continue;
    const float2 g = m.grad[idx];
    const float2 pos = m.pos[idx];
    const float fr = (g.y / gn) * float(r) + pos.y;
    const float fc = (g.x / gn) * float(r) + pos.x;
    const int row = cvtt_f32(roundf(fr));
    const int col = cvtt_f32(roundf(fc));
    if ((unsigned)row >= (unsigned)p.rows || (unsigned)col >= (unsigned)p.cols) continue;
    const unsigned seq = (unsigned)(idx * nr + ri);
    const unsigned key = ((unsigned)abs(r) << kDfSeqBits) | (kDfSeqMask - seq);
    atomicMin(&m.df[(size_t)row * p.cols + col], key);
  }
}

// ---- DistanceField::build, keyline driven (core.hpp:37-59) ----------------------------------------------------------
// The keylines of a detected map are in raster order (their index IS the raster rank, edge_detector.cpp:109-113) and a
// keyline detected in pixel row r only reaches field rows r-41 .. r+41 (unit gradient, 40 steps, sub-pixel offset <= 0.5).
// A workgroup owns a strip of S field rows x W columns in LDS; the keylines that can reach it are the CONTIGUOUS index range
// row_start[y0 - 41] .. row_start[y1 + 41] (k_keyline_emit leaves the per-row first rank with the map). No mask scan, no
// binning pass: every candidate is read once per strip that it can reach (20 bytes of geometry), its unit gradient is
// formed once (the two IEEE divisions of core.hpp:50-51), the r-range that can fall into the strip is clipped, and a wave
// walks it with LDS atomicMin on the same key as k_df_build (min |r|, ties -> last visited). The finished strip is
// written with coalesced stores, empty cells included.
#ifdef RH_DF_PROBE  // tools/df_strips_probe.hip: phase stamps of one workgroup (100 MHz constant clock)
unsigned long long* g_df_stamps_host = nullptr;
#define RH_DFS_STAMP(i) do { if (stamps && blockIdx.x == 0 && blockIdx.y == gridDim.y / 2 && tid == 0) stamps[(i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define RH_DFS_STAMP_ARG , unsigned long long* stamps
#define RH_DFS_STAMP_PASS , g_df_stamps_host
#define RH_DFS_STAMP_FWD , stamps
#else
#define RH_DFS_STAMP(i) do { } while (0)
#define RH_DFS_STAMP_ARG
#define RH_DFS_STAMP_PASS
#define RH_DFS_STAMP_FWD
#endif
constexpr int kDfsThreads = 512;
constexpr int kDfsList = 1024;  // candidates staged per chunk (24 KB)
constexpr int kDfSpecSlots = 256;  // list entries a workgroup reads before it knows its tile's count
constexpr int kDfsSub = 16;     // lanes that walk one keyline's r-range (four keylines per wave)

// Field cells [y0, y1) x [x0, x1) from the row range: every keyline detected in the rows that can reach the box is a
// candidate. Used by the tile kernel for a tile whose list overflowed.
// df_smem: S * W cells, then the staging list (kDfsList entries of 24 bytes).
__device__ __forceinline__ void df_rowrange_body(const KParams& p, const MapDev& m, float thr, int n, int S, int W, int x0, int x1, int y0,
                                                 int y1, unsigned* df_smem, int* l_n_ptr RH_DFS_STAMP_ARG) {
  unsigned* cell = df_smem;                                             // [S][W]
  float4* l_geo = reinterpret_cast<float4*>(df_smem + (size_t)S * W);   // pos.x, pos.y, g.x/gn, g.y/gn   (S*W is a multiple of 4)
  int* l_idx = reinterpret_cast<int*>(l_geo + kDfsList);
  int* l_rr = l_idx + kDfsList;                                         // (r_lo + 256) << 16 | (r_hi + 256)
  int& l_n = *l_n_ptr;
  const int tid = threadIdx.x, lane = tid & 63;
  const int nr = p.df_nr, half = nr >> 1, reach = half + 1;
  const int ia = min(m.row_start[max(y0 - reach, 0)], n), ib = min(m.row_start[min(y1 + reach, p.rows)], n);
  for (int i = tid; i < S * W; i += kDfsThreads) cell[i] = kDfEmpty;
  int stamp_i = 1;
  (void)stamp_i;
  RH_DFS_STAMP(stamp_i++);
  for (int base = ia; base < ib; base += kDfsList) {
    __syncthreads();  // the previous chunk's walk is finished by every wave (first chunk: the cells are initialised)
    if (tid == 0) l_n = 0;
    __syncthreads();
    const int cend = min(base + kDfsList, ib);
    for (int i0 = base; i0 < cend; i0 += kDfsThreads) {
      const int ci = i0 + tid;
      bool keep = false;
      float2 pos = make_float2(0.f, 0.f), u = make_float2(0.f, 0.f);
      int r_lo = 0, r_hi = -1, id = 0;
      if (ci < cend) {
        id = ci;
        const float gn = m.gnorm[id];
        u = m.unit[id];
        pos = m.pos[id];
        if (!(thr > 0.0f && gn < thr)) keep = df_clip_range(pos, u, half, x0, x1, y0, y1, &r_lo, &r_hi);
      }
      const unsigned long long km = __ballot(keep);
      if (km) {
        int slot0 = 0;
        if (lane == 0) slot0 = atomicAdd(&l_n, __popcll(km));
        slot0 = __shfl(slot0, 0);
        if (keep) {
          const int slot = slot0 + __popcll(km & ((1ull << lane) - 1ull));
          l_idx[slot] = id;
          l_geo[slot] = make_float4(pos.x, pos.y, u.x, u.y);
          l_rr[slot] = ((r_lo + 256) << 16) | (r_hi + 256);
        }
      }
    }
    __syncthreads();
    RH_DFS_STAMP(stamp_i++);
    const int ln = l_n;
    const int l16 = tid & (kDfsSub - 1);
    for (int k = tid / kDfsSub; k < ln; k += kDfsThreads / kDfsSub) {
      const float4 q = l_geo[k];
      const int rr = l_rr[k];
      const int r_hi = (rr & 0xFFFF) - 256;
      const unsigned seq0 = (unsigned)(l_idx[k] * nr);
      for (int r = (rr >> 16) - 256 + l16; r <= r_hi; r += kDfsSub) {
        const float fr = q.w * float(r) + q.y;
        const float fc = q.z * float(r) + q.x;
        const int row = cvtt_f32(roundf(fr));
        const int col = cvtt_f32(roundf(fc));
        if (row < y0 || row >= y1 || col < x0 || col >= x1) continue;
        const unsigned key = ((unsigned)abs(r) << kDfSeqBits) | (kDfSeqMask - (seq0 + (unsigned)(r + half)));
        atomicMin(&cell[(row - y0) * W + (col - x0)], key);
      }
    }
    RH_DFS_STAMP(stamp_i++);
  }
  __syncthreads();
  RH_DFS_STAMP(stamp_i++);
  const int tw = x1 - x0, th = y1 - y0;
  if (((p.cols | x0 | tw | W) & 3) == 0) {  // 16-byte rows: vector stores
    const int tw4 = tw >> 2;
    for (int i = tid; i < th * tw4; i += kDfsThreads) {
      const int ty = i / tw4, t4 = i - ty * tw4;
      *reinterpret_cast<uint4*>(&m.df[(size_t)(y0 + ty) * p.cols + x0 + t4 * 4]) = *reinterpret_cast<const uint4*>(&cell[ty * W + t4 * 4]);
    }
  } else {
    for (int i = tid; i < th * tw; i += kDfsThreads) {
      const int ty = i / tw, tx = i - ty * tw;
      m.df[(size_t)(y0 + ty) * p.cols + x0 + tx] = cell[ty * W + tx];
    }
  }
  RH_DFS_STAMP(stamp_i++);
}

// ---- DistanceField::build, one workgroup per T x T tile fed by the tile's list (the default) --------------------------------
// k_join_edges left, per tile, one 32-byte entry for every keyline whose segment crosses it: position, unit gradient, index,
// the r-range that can fall into the tile, gradient norm. The workgroup reads its entries with one coalesced pass into LDS,
// 16-lane groups walk the ranges with LDS atomicMin on the key of k_df_build (min |r|, ties -> last visited), and the
// finished tile is stored, empty cells included (no clearing pass). A tile whose list overflowed (more than kDfTileCap
// crossing keylines) is rebuilt from the row range instead: same cells, just slower.
template <int T>
__device__ __forceinline__ void df_lists_body(KParams p, MapDev m, const DetState* __restrict__ det_prev RH_DFS_STAMP_ARG) {
  extern __shared__ __align__(16) unsigned df_smem[];
  __shared__ int l_n;
  const int tid = threadIdx.x;
  RH_DFS_STAMP(0);
  const int tile = blockIdx.y * gridDim.x + blockIdx.x;
  // every scalar this workgroup needs, and its own entry, in one round of independent loads
  const int cnt = m.tile_cnt[tile];
  const float4* ent = m.tile_list + (size_t)tile * kDfTileCap * 2;
  // the first kDfSpecSlots entries are read before cnt is known (speculative: garbage beyond cnt, never used) - the usual tile
  // holds ~200; slots beyond that only when the count says so (a dependent round trip for the few crowded tiles instead of
  // 16 KB fetched per tile whatever its list holds: FETCH_SIZE of k_df_lists_b 37.8 -> 20.3 MB per 8-lane launch)
  float4 e0 = make_float4(0.f, 0.f, 0.f, 0.f), e1 = e0;
  if (tid < kDfSpecSlots) {
    e0 = ent[2 * tid];
    e1 = ent[2 * tid + 1];
  }
  const MapState st = *m.st;
  const float prev_auto = det_prev->auto_threshold;
  const int n = st.n;
  const float thr = auto_threshold_from(st, prev_auto);
  if (tile == 0 && tid == 0) m.st->threshold = thr;  // EdgeMap::threshold(auto_threshold_) (:185)
  const int x0 = blockIdx.x * T, y0 = blockIdx.y * T;
  const int x1 = min(x0 + T, p.cols), y1 = min(y0 + T, p.rows);
  if (tid >= kDfSpecSlots && tid < cnt && cnt <= kDfTileCap) {
    e0 = ent[2 * tid];
    e1 = ent[2 * tid + 1];
  }
  if (cnt > kDfTileCap) {
    df_rowrange_body(p, m, thr, n, T, T, x0, x1, y0, y1, df_smem, &l_n RH_DFS_STAMP_FWD);
    return;
  }
  // cell pitch T + 1: a group's lanes step along the segment, for a vertical one down a column - with a pitch of T words
  // (32 or 64) those 16 cells would share two LDS banks
  constexpr int TP = T + 1;
  unsigned* cell = df_smem;                                                        // [T][TP]
  float4* l_ent = reinterpret_cast<float4*>(df_smem + ((T * TP + 3) & ~3));        // [kDfTileCap][2]
  for (int i = tid; i < T * TP; i += kDfsThreads) cell[i] = kDfEmpty;
  static_assert(kDfTileCap <= kDfsThreads, "one entry per thread");
  if (tid < cnt) {
    const bool skip = thr > 0.0f && e1.z < thr;  // threshold on the gradient norm (core.hpp:47)
    l_ent[2 * tid] = e0;
    l_ent[2 * tid + 1] = make_float4(e1.x, skip ? __int_as_float((256 << 16) | 255) : e1.y, e1.z, 0.f);  // skip: empty r-range
  }
  __syncthreads();
  RH_DFS_STAMP(1);
  const int nr = p.df_nr, half = nr >> 1;
  const int l16 = tid & (kDfsSub - 1);
  const unsigned tw = (unsigned)(x1 - x0), th = (unsigned)(y1 - y0);
  for (int k = tid / kDfsSub; k < cnt; k += kDfsThreads / kDfsSub) {
    const float4 q = l_ent[2 * k], a = l_ent[2 * k + 1];
    const int rr = __float_as_int(a.y);
    const int r_hi = (rr & 0xFFFF) - 256;
    const unsigned kbase = kDfSeqMask - (unsigned)(__float_as_int(a.x) * nr + half);  // key = |r| << 23 | (kbase - r)
    for (int r = (rr >> 16) - 256 + l16; r <= r_hi; r += kDfsSub) {
      const float fr = q.w * float(r) + q.y;
      const float fc = q.z * float(r) + q.x;
      // (the positions are finite and far inside the int range: a plain conversion equals cvtt_f32 here)
      const unsigned ty = (unsigned)((int)roundf(fr) - y0), tx = (unsigned)((int)roundf(fc) - x0);
      if (ty >= th || tx >= tw) continue;
      atomicMin(&cell[ty * TP + tx], ((unsigned)abs(r) << kDfSeqBits) | (kbase - (unsigned)r));
    }
  }
  __syncthreads();
  RH_DFS_STAMP(2);
  for (int i = tid; i < (int)(th * T); i += kDfsThreads) {  // a row of the tile = consecutive lanes = one 128/256-byte segment
    const int ty = i / T, tx = i - ty * T;
    if (tx < (int)tw) m.df[(size_t)(y0 + ty) * p.cols + x0 + tx] = cell[ty * TP + tx];
  }
  RH_DFS_STAMP(3);
}

template <int T>
__global__ __launch_bounds__(kDfsThreads) void k_df_lists(KParams p, MapDev m, const DetState* __restrict__ det_prev RH_DFS_STAMP_ARG) {
  df_lists_body<T>(p, m, det_prev RH_DFS_STAMP_FWD);
}
#ifndef RH_DF_PROBE
template <int T>
__global__ __launch_bounds__(kDfsThreads) void k_df_lists_b(KParams p, const LaneStatic* __restrict__ ls, const MapDev* __restrict__ maptab,
                                                            LaneDynB dyn) {
  const LaneDyn d = dyn.v[blockIdx.z];
  df_lists_body<T>(p, global_map(lane_map(maptab, blockIdx.z, d.nm, d.nm_swap)), gptr(ls[blockIdx.z].det) + d.det_out);
}
#endif

__global__ __launch_bounds__(256) void k_df_decode(KParams p, MapDev m, int* __restrict__ id_out, int* __restrict__ dist_out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= p.rows * p.cols) return;
  const unsigned key = m.df[i];
  int id = -1, dist = 0x7FFFFFFF;
  if (key != kDfEmpty) {
    id = (int)((kDfSeqMask - (key & kDfSeqMask)) / (unsigned)p.df_nr);
    dist = (int)(key >> kDfSeqBits);
  }
  if (id_out) id_out[i] = id;
  if (dist_out) dist_out[i] = dist;
}

// ---- AoS <-> SoA for the lazy host mirror of EdgeMap::keylines() (edge_map.hpp:50) ------------------------
__global__ __launch_bounds__(256) void k_map_pack(MapDev m, rebvio_hip_keyline* __restrict__ out) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= m.st->n) return;
  rebvio_hip_keyline k;
  const float2 a = m.pos[idx], b = m.pos_img[idx], c = m.mpos_img[idx], g = m.grad[idx], mg = m.mgrad[idx], rs = m.rs[idx];
  k.pos[0] = a.x; k.pos[1] = a.y;
  k.pos_img[0] = b.x; k.pos_img[1] = b.y;
  k.match_pos_img[0] = c.x; k.match_pos_img[1] = c.y;
  k.gradient[0] = g.x; k.gradient[1] = g.y;
  k.match_gradient[0] = mg.x; k.match_gradient[1] = mg.y;
  k.gradient_norm = m.gnorm[idx];
  k.match_gradient_norm = m.mgnorm[idx];
  k.rho = rs.x; k.sigma_rho = rs.y;
  k.id = -1;
  k.id_prev = m.id_prev[idx];
  k.id_next = m.id_next[idx];
  k.match_id = m.match_id[idx];
  k.match_id_forward = m.match_fwd[idx];
  k.match_id_keyframe = m.match_kf[idx];
  k.matches = m.matches[idx];
  out[idx] = k;
}

__global__ __launch_bounds__(256) void k_map_unpack(MapDev m, const rebvio_hip_keyline* __restrict__ in, int n) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= n) return;
  const rebvio_hip_keyline k = in[idx];
  m.pos[idx] = make_float2(k.pos[0], k.pos[1]);
  m.pos_img[idx] = make_float2(k.pos_img[0], k.pos_img[1]);
  m.mpos_img[idx] = make_float2(k.match_pos_img[0], k.match_pos_img[1]);
  m.grad[idx] = make_float2(k.gradient[0], k.gradient[1]);
  m.mgrad[idx] = make_float2(k.match_gradient[0], k.match_gradient[1]);
  m.gnorm[idx] = k.gradient_norm;
  m.mgnorm[idx] = k.match_gradient_norm;
  m.rs[idx] = make_float2(k.rho, k.sigma_rho);
  m.id_prev[idx] = k.id_prev;
  m.id_next[idx] = k.id_next;
  m.match_id[idx] = k.match_id;
  m.match_fwd[idx] = k.match_id_forward;
  m.match_kf[idx] = k.match_id_keyframe;
  m.matches[idx] = k.matches;
}

// ---- edge image of the callback consumers (what ros_rebvio.cpp:32-50 draws on the host): the grey frame
// replicated to RGB, every keyline's pixel (round(pos.y), round(pos.x)) set to (255, 0, 0) ----------------------------
__global__ __launch_bounds__(256) void k_render_gray(const uint8_t* __restrict__ gray, uint8_t* __restrict__ rgb, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const uint8_t g = gray ? gray[i] : (uint8_t)0;
  rgb[3 * i] = g;
  rgb[3 * i + 1] = g;
  rgb[3 * i + 2] = g;
}
__global__ __launch_bounds__(256) void k_render_keylines(KParams p, MapDev m, uint8_t* __restrict__ rgb) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= m.st->n) return;
  const float2 q = m.pos[idx];
  const int row = cvtt_f32(roundf(q.y)), col = cvtt_f32(roundf(q.x));
  if ((unsigned)row >= (unsigned)p.rows || (unsigned)col >= (unsigned)p.cols) return;
  uint8_t* o = rgb + 3 * ((size_t)row * p.cols + col);
  o[0] = 255;
  o[1] = 0;
  o[2] = 0;
}

// ---- launchers -----------------------------------------------------------------------------------------------
// Rows per 64-column tile of k_dog_mag / k_keyline_flag / k_keyline_emit. Batches: 16 (fewest staged values and prefix sums per
// pixel: the 8-lane step went from 232 to 202 us with these three kernels re-tiled). A single stream: 8 - its 600 workgroups
// per kernel finish sooner on an otherwise idle chip than 300 of 16 rows (rocprofv3, 640x480: k_dog_mag 8.1 / 11.3 us,
// k_keyline_flag 7.7 / 10.7, k_keyline_emit 7.5 / 9.0 for 8 / 16 rows).
constexpr int kTileRowsSingle = 8;
static int lds_pitch(int cols) {
  int pad = 4;
  if (((cols + pad) / 4) % 2 == 0) pad = 8;
  return cols + pad;
}
// ---- front end: convertTo(CV_32F, 3.0) + cv::undistort as one gather kernel (rebvio.cpp:43-47) -------
// map = fixed-point source coordinates (1/32 px, hostmath.hpp undistort_fixed_map). Weights (1-a)(1-b) .. with a, b
// multiples of 1/32 and 8-bit*3 sources make every product and the 4-term sum exact in fp32, so the result does not
// depend on evaluation order; taps outside the image read the constant border 0 (BORDER_CONSTANT).
__device__ __forceinline__ void front_end_body(const uint8_t* __restrict__ src, const int2* __restrict__ map, float* __restrict__ dst,
                                               int rows, int cols) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= rows * cols) return;
  const int2 m = map[i];
  const int sx = m.x >> 5, sy = m.y >> 5;
  const float ax = (float)(m.x & 31) * 0.03125f, ay = (float)(m.y & 31) * 0.03125f;
  const bool x0 = (unsigned)sx < (unsigned)cols, x1 = (unsigned)(sx + 1) < (unsigned)cols;
  const bool y0 = (unsigned)sy < (unsigned)rows, y1 = (unsigned)(sy + 1) < (unsigned)rows;
  const uint8_t* r0 = src + (size_t)(y0 ? sy : 0) * cols;
  const uint8_t* r1 = src + (size_t)(y1 ? sy + 1 : 0) * cols;
  const float s00 = (x0 && y0) ? (float)r0[sx] * 3.0f : 0.0f;
  const float s01 = (x1 && y0) ? (float)r0[sx + 1] * 3.0f : 0.0f;
  const float s10 = (x0 && y1) ? (float)r1[sx] * 3.0f : 0.0f;
  const float s11 = (x1 && y1) ? (float)r1[sx + 1] * 3.0f : 0.0f;
  const float w00 = (1.0f - ay) * (1.0f - ax), w01 = (1.0f - ay) * ax, w10 = ay * (1.0f - ax), w11 = ay * ax;
  dst[i] = s00 * w00 + s01 * w01 + s10 * w10 + s11 * w11;
}
__global__ __launch_bounds__(256) void k_front_end_u8(const uint8_t* __restrict__ src, const int2* __restrict__ map,
                                                        float* __restrict__ dst, int rows, int cols) {
  front_end_body(src, map, dst, rows, cols);
}
// batched form (lane = blockIdx.z): every lane gathers through ITS lens model into its own fp32 frame of this step's parity
__global__ __launch_bounds__(256) void k_front_end_u8_b(const LaneStatic* __restrict__ ls, LaneDynB dyn, int lane0, int rows, int cols) {
  const int lane = lane0 + blockIdx.z;
  const LaneStatic& L = ls[lane];
  front_end_body(static_cast<const uint8_t*>(dyn.v[lane].img), gptr(L.undist_map), gptr(L.undist_img[dyn.v[lane].parity]), rows, cols);
}

// Host frame -> device staging frame as a kernel of the scan stream (16 bytes per lane straight from the pinned ring slot over
// PCIe): a hipMemcpyAsync there is a packet for the DMA engine with its own cross-engine synchronisation in front of and
// behind it, which cost the scan stream more than the transfer itself (measured: 8.2k frames/s with the DMA copy).
__global__ __launch_bounds__(256) void k_copy16(const uint4* __restrict__ src, uint4* __restrict__ dst, int n16) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n16) dst[i] = src[i];
}
void launch_copy_from_pinned(hipStream_t s, const void* src_pinned, void* dst_dev, size_t bytes) {
  const int n16 = (int)((bytes + 15) / 16);  // (both buffers are allocated in whole 16-byte units)
  RH_LAUNCH(k_copy16, dim3(div_up(n16, 256)), dim3(256), 0, s, (const uint4*)src_pinned, (uint4*)dst_dev, n16);
}

void launch_front_end_u8(hipStream_t s, const KParams& p, const uint8_t* src, const int2* map, float* dst) {
  RH_LAUNCH(k_front_end_u8, dim3(div_up(p.rows * p.cols, 256)), dim3(256), 0, s, src, map, dst, p.rows, p.cols);
}

// part: 1 = everything up to the row pass of the third box filter (five kernels), 2 = its column pass + k_dog_mag, 3 = both.
// The streaming driver runs part 2 on the keyline stream: the scan stream is the busiest of a frame's three.
void launch_scale_space(hipStream_t s, const KParams& p, const void* img, int img_is_u8, const ScaleBufs& sb,
                        const int widths[2][3], int* rowcount_to_zero, int part, bool fuse_dog) {
  const int R = p.rows, C = p.cols;
  const int Cp = (C + 3) & ~3;  // pitch of the scan buffers sb.a / sb.b
  const int ldw_abs = lds_pitch(Cp);
  const size_t shm = (size_t)kStrip * ldw_abs * sizeof(float);
  const int ldw = ldw_abs;
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_rowscan<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_rowscan<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_rowscan<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_colscan), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done = true;
  }
  const dim3 g1(div_up(R, kStrip), 1), g2(div_up(R, kStrip), 2);
  const dim3 c1(div_up(Cp, kColStrip), 1), c2(div_up(Cp, kColStrip), 2);
  const int ldh = lds_pitch(R + (4 - R % 4) % 4);
  const size_t cshm = (size_t)ldh * kColStrip * sizeof(float);
#define RH_COLSCAN(one, b0, b1) RH_LAUNCH(k_colscan, (one) ? c1 : c2, dim3(256), cshm, s, b0, b1, R, Cp, ldh, (int*)nullptr, 0)
  if (part & 1) {
    // pass 1: both filters share the integral image of the input (scale_space.cpp:175)
    if (img_is_u8)
      RH_LAUNCH(k_rowscan<0>, g1, dim3(256), shm, s, img, img, sb.a[0], sb.a[0], R, C, 0, 0, ldw);
    else
      RH_LAUNCH(k_rowscan<1>, g1, dim3(256), shm, s, img, img, sb.a[0], sb.a[0], R, C, 0, 0, ldw);
    RH_COLSCAN(true, sb.a[0], sb.a[0]);
    // pass 2: average(width[0]) fused into the row scan, per filter
    RH_LAUNCH(k_rowscan<2>, g2, dim3(256), shm, s, (const void*)sb.a[0], (const void*)sb.a[0], sb.b[0], sb.b[1], R,
                       C, widths[0][0], widths[1][0], ldw);
    RH_COLSCAN(false, sb.b[0], sb.b[1]);
    // pass 3: filter f averages its own integral image
    RH_LAUNCH(k_rowscan<2>, g2, dim3(256), shm, s, (const void*)sb.b[0], (const void*)sb.b[1], sb.a[0], sb.a[1], R, C, widths[0][1],
              widths[1][1], ldw);
  }
  if ((part & 2) && fuse_dog) {
    // the candidate kernel forms the last box pass, DoG and gradient itself (launch_keylines, k_keyline_flag_ii): only the
    // column pass remains here, and it clears the frame's row counters
    RH_LAUNCH(k_colscan, c2, dim3(256), cshm, s, sb.a[0], sb.a[1], R, Cp, ldh, rowcount_to_zero, R);
  } else if (part & 2) {
    RH_COLSCAN(false, sb.a[0], sb.a[1]);
    const dim3 gt(div_up(C, 64), div_up(R, kTileRowsSingle));
    RH_LAUNCH(k_dog_mag<kTileRowsSingle>, gt, dim3(64, 4), 0, s, (const float*)sb.a[0], (const float*)sb.a[1], widths[0][2], widths[1][2], sb.dog,
              sb.mag, sb.scale0, sb.scale1, R, C, rowcount_to_zero);
  }
#undef RH_COLSCAN
}

// FastGaussian(camera, sigma, n).smooth for any number of box passes (scale_space.cpp:173-182): createIntegralImage, then n - 1
// times average + createIntegralImage, then the last average - the detector's kernels with ONE filter slot (grids of height 1)
// and k_dog_mag as the last average (its scale0 output; DoG and gradient go to the scratch buffers and are ignored).
void launch_smooth_n(hipStream_t s, const KParams& p, const float* img, const ScaleBufs& sb, const int* widths, int n, int* rowcount_to_zero) {
  const int R = p.rows, C = p.cols;
  const int Cp = (C + 3) & ~3;
  const int ldw = lds_pitch(Cp);
  const size_t shm = (size_t)kStrip * ldw * sizeof(float);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_rowscan<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_rowscan<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_colscan), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  const dim3 g1(div_up(R, kStrip), 1), c1(div_up(Cp, kColStrip), 1);
  const int ldh = lds_pitch(R + (4 - R % 4) % 4);
  const size_t cshm = (size_t)ldh * kColStrip * sizeof(float);
  float* cur = sb.a[0];
  float* other = sb.b[0];
  RH_LAUNCH(k_rowscan<1>, g1, dim3(256), shm, s, (const void*)img, (const void*)img, cur, cur, R, C, 0, 0, ldw);
  RH_LAUNCH(k_colscan, c1, dim3(256), cshm, s, cur, cur, R, Cp, ldh, (int*)nullptr, 0);
  for (int i = 0; i + 1 < n; ++i) {
    RH_LAUNCH(k_rowscan<2>, g1, dim3(256), shm, s, (const void*)cur, (const void*)cur, other, other, R, C, widths[i], widths[i], ldw);
    RH_LAUNCH(k_colscan, c1, dim3(256), cshm, s, other, other, R, Cp, ldh, (int*)nullptr, 0);
    std::swap(cur, other);
  }
  const dim3 gt(div_up(C, 64), div_up(R, kTileRowsSingle));
  RH_LAUNCH(k_dog_mag<kTileRowsSingle>, gt, dim3(64, 4), 0, s, (const float*)cur, (const float*)cur, widths[n - 1], widths[n - 1], sb.dog, sb.mag,
            sb.scale0, sb.scale1, R, C, rowcount_to_zero);
}

// ---- batched launchers (lane = blockIdx.z): the same grids with a third dimension ------------------------------------------
void launch_scale_space_b(hipStream_t s, const KParams& p, int lane0, int lanes, const LaneStatic* ls, const LaneDynB& dyn,
                          const int widths[2][3], bool lens, bool fuse_dog) {
  const int R = p.rows, C = p.cols;
  const int Cp = (C + 3) & ~3;
  const int ldw_abs = lds_pitch(Cp);
  const size_t shm = (size_t)kStrip * ldw_abs * sizeof(float);
  const int ldw = ldw_abs;
  const int ldh = lds_pitch(R + (4 - R % 4) % 4);
  const size_t cshm = (size_t)ldh * kColStrip * sizeof(float);
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_rowscan_b<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_rowscan_b<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_rowscan_b<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_colscan_b), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done = true;
  }
  const unsigned z = (unsigned)lanes;
  const dim3 g1(div_up(R, kStrip), 1, z), g2(div_up(R, kStrip), 2, z);
  const dim3 c1(div_up(Cp, kColStrip), 1, z), c2(div_up(Cp, kColStrip), 2, z);
#define RH_COLSCAN_B(which) RH_LAUNCH(k_colscan_b, (which) == 0 ? c1 : c2, dim3(256), cshm, s, ls, lane0, which, R, Cp, ldh, -1)
  if (lens) {  // x3 + undistort of every lane's frame (rebvio.cpp:43-47), then the first pass on the fp32 result
    RH_LAUNCH(k_front_end_u8_b, dim3(div_up(R * C, 256), 1, z), dim3(256), 0, s, ls, dyn, lane0, R, C);
    RH_LAUNCH(k_rowscan_b<1>, g1, dim3(256), shm, s, ls, dyn, lane0, 3, R, C, 0, 0, ldw);
  } else {
    RH_LAUNCH(k_rowscan_b<0>, g1, dim3(256), shm, s, ls, dyn, lane0, 0, R, C, 0, 0, ldw);
  }
  RH_COLSCAN_B(0);
  RH_LAUNCH(k_rowscan_b<2>, g2, dim3(256), shm, s, ls, dyn, lane0, 1, R, C, widths[0][0], widths[1][0], ldw);
  RH_COLSCAN_B(1);
  if (fuse_dog) {  // k_keyline_flag_ii_b forms the last box pass, DoG and gradient itself (launch_keylines_b)
    RH_LAUNCH(k_rowscan_b<2>, g2, dim3(256), shm, s, ls, dyn, lane0, 4, R, C, widths[0][1], widths[1][1], ldw);
    RH_LAUNCH(k_colscan_b, c2, dim3(256), cshm, s, ls, lane0, 3, R, Cp, ldh, (int)dyn.v[lane0].parity);
    return;
  }
  RH_LAUNCH(k_rowscan_b<2>, g2, dim3(256), shm, s, ls, dyn, lane0, 2, R, C, widths[0][1], widths[1][1], ldw);
  RH_COLSCAN_B(2);
  RH_LAUNCH(k_dog_mag_b, dim3(div_up(C, 64), div_up(R, 16), z), dim3(64, 4), 0, s, ls, dyn, lane0, widths[0][2], widths[1][2], R, C);
#undef RH_COLSCAN_B
}

void launch_keylines_b(hipStream_t s, const KParams& p, int lanes, const LaneStatic* ls, const MapDev* maptab, const LaneDynB& dyn,
                       const int* fuse_widths) {
  const unsigned z = (unsigned)lanes;
  const DfGrid dg = df_grid(p.rows, p.cols);
  if (fuse_widths)
    RH_LAUNCH(k_keyline_flag_ii_b, dim3(div_up(p.cols, 64), div_up(p.rows, 16), z), dim3(64, 4), 0, s, p, ls, dyn, fuse_widths[0], fuse_widths[1]);
  else
    RH_LAUNCH(k_keyline_flag_b, dim3(div_up(p.cols, 64), div_up(p.rows, 16), z), dim3(64, 4), 0, s, p, ls, dyn);
  RH_LAUNCH(k_keyline_emit_b, dim3(div_up(p.cols, 64), div_up(p.rows, 16), z), dim3(64, 4), 0, s, p, ls, maptab, dyn, 0, dg.ntx * dg.nty);
  RH_LAUNCH(k_join_edges_b, dim3(div_up(p.kmax, 256), 1, z), dim3(256), (size_t)dg.ntx * dg.nty * sizeof(int), s, p, maptab, dyn, dg.T, dg.ntx,
            dg.nty);
}

void launch_df_build_b(hipStream_t s, const KParams& p, int lanes, const LaneStatic* ls, const MapDev* maptab, const LaneDynB& dyn) {
  const DfGrid dg = df_grid(p.rows, p.cols);
  const size_t list_bytes = (size_t)kDfsList * (sizeof(float4) + 2 * sizeof(int));
  const size_t shm = (((size_t)dg.T * (dg.T + 1) + 3) & ~(size_t)3) * sizeof(unsigned) + list_bytes;
  static size_t have[2] = {0, 0};
  const dim3 grid(dg.ntx, dg.nty, (unsigned)lanes);
  if (dg.T == 32) {
    if (shm > have[0]) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_df_lists_b<32>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm) != hipSuccess)
        (void)hipGetLastError();
      have[0] = shm;
    }
    RH_LAUNCH(k_df_lists_b<32>, grid, dim3(kDfsThreads), shm, s, p, ls, maptab, dyn);
  } else {
    if (shm > have[1]) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_df_lists_b<64>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm) != hipSuccess)
        (void)hipGetLastError();
      have[1] = shm;
    }
    RH_LAUNCH(k_df_lists_b<64>, grid, dim3(kDfsThreads), shm, s, p, ls, maptab, dyn);
  }
}

void launch_keylines(hipStream_t s, const KParams& p, const ScaleBufs& sb, const DetectBufs& db, const MapDev& m,
                     const DetState* det_in, DetState* det_out, const MapState* prev_st, const int* fuse_widths) {
  const DfGrid dg = df_grid(p.rows, p.cols);
  const dim3 gt(div_up(p.cols, 64), div_up(p.rows, kTileRowsSingle));
  if (fuse_widths)  // sb.a[] hold the third filter pass's integral images (launch_scale_space with fuse_dog)
    RH_LAUNCH(k_keyline_flag_ii<kTileRowsSingle>, gt, dim3(64, 4), 0, s, (const float*)sb.a[0], (const float*)sb.a[1], fuse_widths[0], fuse_widths[1], p,
              det_in, db.stash, db.bits, db.rowcount);
  else
    RH_LAUNCH(k_keyline_flag<kTileRowsSingle>, gt, dim3(64, 4), 0, s, (const float*)sb.dog, (const float*)sb.mag, p, det_in, db.stash, db.bits,
              db.rowcount);
  RH_LAUNCH(k_keyline_emit<kTileRowsSingle>, gt, dim3(64, 4), 0, s, p, m, (const float4*)db.stash, (const unsigned long long*)db.bits,
            (const int*)db.rowcount, det_in, det_out, prev_st, 0, dg.ntx * dg.nty);
  RH_LAUNCH(k_join_edges, dim3(div_up(p.kmax, 256)), dim3(256), (size_t)dg.ntx * dg.nty * sizeof(int), s, p, m, dg.T, dg.ntx, dg.nty);
}

void launch_df_build(hipStream_t s, const KParams& p, const MapDev& m, const DetState* det_prev, bool mask_is_current) {
  if (mask_is_current) {
    // detection path: one workgroup per tile of df_grid, fed by the per-tile keyline lists k_join_edges left with the map
    const DfGrid dg = df_grid(p.rows, p.cols);
    const size_t list_bytes = (size_t)kDfsList * (sizeof(float4) + 2 * sizeof(int));  // (>= kDfTileCap entries of 32 bytes)
    static size_t attr_shm[2] = {0, 0};
    auto want = [](const void* f, size_t shm, size_t* have) {  // (the kernels also have static LDS: ask for what is needed)
      if (shm > *have) {
        if (hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm) != hipSuccess) (void)hipGetLastError();
        *have = shm;
      }
    };
    const size_t shm = (((size_t)dg.T * (dg.T + 1) + 3) & ~(size_t)3) * sizeof(unsigned) + list_bytes;
    if (dg.T == 32) {
      want(reinterpret_cast<const void*>(&k_df_lists<32>), shm, &attr_shm[0]);
      RH_LAUNCH(k_df_lists<32>, dim3(dg.ntx, dg.nty), dim3(kDfsThreads), shm, s, p, m, det_prev RH_DFS_STAMP_PASS);
    } else {  // (cols <= 4096 and rows <= 2548 keep 64-pixel tiles below kDfMaxTiles)
      want(reinterpret_cast<const void*>(&k_df_lists<64>), shm, &attr_shm[1]);
      RH_LAUNCH(k_df_lists<64>, dim3(dg.ntx, dg.nty), dim3(kDfsThreads), shm, s, p, m, det_prev RH_DFS_STAMP_PASS);
    }
    return;
  }
  // rebuild after rebvio_hip_map_upload (arbitrary keylines, no raster order, no tile lists): global-atomic scatter into a
  // cleared field. Measured (MI355X, 640x480): 16 workgroups -> 316 us, 32 -> 172, 64..128 -> 132, 512 -> 156: bound by the
  // memory-side atomic rate, a larger grid only takes CUs and memory queues from the latency-critical streams
  const long long threads = (long long)p.kmax * p.df_nr;
  const unsigned blocks = (unsigned)std::min<long long>((threads + 255) / 256, 128);
  RH_LAUNCH(k_df_build, dim3(blocks), dim3(256), 0, s, p, m, det_prev);
}

void launch_df_decode(hipStream_t s, const KParams& p, const MapDev& m, int* id_out, int* dist_out) {
  RH_LAUNCH(k_df_decode, dim3(div_up(p.rows * p.cols, 256)), dim3(256), 0, s, p, m, id_out, dist_out);
}

void launch_render_edge_image(hipStream_t s, const KParams& p, const MapDev& m, const uint8_t* gray_or_null, uint8_t* rgb) {
  RH_LAUNCH(k_render_gray, dim3(div_up(p.rows * p.cols, 256)), dim3(256), 0, s, gray_or_null, rgb, p.rows * p.cols);
  RH_LAUNCH(k_render_keylines, dim3(div_up(p.kmax, 256)), dim3(256), 0, s, p, m, rgb);
}

void launch_map_pack(hipStream_t s, const KParams& p, const MapDev& m, rebvio_hip_keyline* aos_dev) {
  RH_LAUNCH(k_map_pack, dim3(div_up(p.kmax, 256)), dim3(256), 0, s, m, aos_dev);
}

void launch_map_unpack(hipStream_t s, const KParams& p, const MapDev& m, const rebvio_hip_keyline* aos_dev, int n) {
  RH_LAUNCH(k_map_unpack, dim3(div_up(p.kmax, 256)), dim3(256), 0, s, m, aos_dev, n);
}

}  // namespace rh
