// Internal structures shared by the HIP kernels and the C-ABI host code (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <cstdlib>
#include <string>
#include <stdint.h>

#include "rebvio_hip.h"

namespace rh {

constexpr int kWave = 64;           // CDNA wavefront
constexpr int kBlock = 256;         // 4 waves
constexpr int kNumBins = 100;       // EdgeDetectorConfig::num_bins (edge_detector.hpp:29)
constexpr int kMaxImageValue = 765; // edge_detector.cpp:21
constexpr unsigned kDfEmpty = 0xFFFFFFFFu;
constexpr int kDfSeqBits = 23;
constexpr unsigned kDfSeqMask = (1u << kDfSeqBits) - 1u;
constexpr int kPartStride = 16;     // floats per block record of k_try_vel
constexpr int kXrvStride = 32;      // floats per block record of k_ext_rot_vel
constexpr int kMaxLmCalls = 8;
// 64-bit exchange words of the persistent LM kernels for `groups` record groups of 256 keylines (track.hip: record sets per
// evaluation, the final velocity, neighbour carry words and barrier words of the speculative form), then the words of the
// extRotVel sums the device glue collects (kXrvStride per group)
__host__ __device__ constexpr size_t lm_xch_xrv_offset(size_t groups, size_t workgroups) {
  return ((size_t)kMaxLmCalls * groups + 1) * 16 + (2 * (size_t)kMaxLmCalls + 1) * workgroups;
}
constexpr size_t lm_xch_words(size_t groups) { return lm_xch_xrv_offset(groups, groups) + groups * (size_t)kXrvStride; }
constexpr int kDfTile = 32;         // distance-field tile edge (pixels); 64 for sensors with more than kDfMaxTiles 32-pixel tiles
constexpr int kDfMaxTiles = 4096;   // per-workgroup LDS counter table of the binning pass
constexpr int kDfTileCap = 512;     // list capacity per tile (32-byte entries); a fuller tile falls back to scanning its candidate rows
constexpr int kMaxRecBlocks = 256;   // record groups (256 keylines each) the LM reduction stages in LDS: keylines_max <= 65536
constexpr int kDetRing = 4;         // DetState ring depth (distance-field stream may lag the detect stream)
constexpr float kResidualCarry = -1.0f;  // marker: "|fi| carried in from an earlier block" (see try_vel)

// Device-resident per-map scalars.
struct MapState {
  int n;               // keylines kept (min(total, keylines_max))
  int total;           // candidates found before truncation
  unsigned gmin_bits;  // min / max gradient_norm as IEEE bits (values are >= 0)
  unsigned gmax_bits;
  float threshold;     // EdgeMap::threshold_ (edge_map.hpp:132)
  int dm_matches;      // directedMatch counters
  int dm_kf;
  int reg_count;
  int dm_queued;       // directedMatch: long searches, i.e. keylines still open after the first four probe steps (diagnostic)
  int pad[7];
};

// EdgeDetector servo state (edge_detector.hpp:84-91), ping-ponged between frames.
struct DetState {
  float threshold;       // config_->threshold
  int count;             // keylines_count_ of the previous detect
  float auto_threshold;  // auto_threshold_
  int pad;
};

// SoA view of one edge map in HBM (capacity keylines_max). float2 pairs keep gathers at 8 bytes.
struct MapDev {
  float2* pos;       // KeyLine::pos
  float2* pos_img;   // KeyLine::pos_img
  float2* mpos_img;  // KeyLine::match_pos_img
  float2* grad;      // KeyLine::gradient
  float2* mgrad;     // KeyLine::match_gradient
  float* gnorm;      // KeyLine::gradient_norm
  float* mgnorm;     // KeyLine::match_gradient_norm
  float2* rs;        // (rho, sigma_rho)
  float2* rs_tmp;    // regularize1Iter staging / ping-pong partner of rs
  float2* grad_tmp;  // ping-pong partner of grad (fused regularize+EKF+rotate)
  int* id_prev;
  int* id_next;
  int* match_id;
  int* match_fwd;    // match_id_forward
  int* match_kf;     // match_id_keyframe
  unsigned* matches;
  unsigned long long* fwd_key;  // forwardMatch winner key per keyline of this (new) map
  float* residual;              // minimizeVel's residuals[] for this (old) map
  int* mask;                    // dense image index -> keyline index
  unsigned* df;                 // distance field keys built from this map
  float2* unit;                 // gradient / gradient_norm as DistanceField::build forms it (core.hpp:50-51), left by joinEdges
  int* tile_cnt;                // [tiles] keylines whose +-search_range segment crosses each distance-field tile (joinEdges bins them)
  float4* tile_list;            // [tiles][kDfTileCap][2] their entries {pos.x, pos.y, u.x, u.y} {idx, r-range, gradient_norm, -}, any order
  int* row_start;               // [rows + 1] index of the first keyline detected in each pixel row (raster rank; [rows] = n)
  MapState* st;
};

struct KParams {
  int rows, cols;
  float fm, cx, cy;
  int kmax, kref;
  float pos_neg_threshold, dog_threshold, gain, max_threshold, min_threshold;
  float search_range, reweight_distance, match_treshold;
  unsigned min_match_threshold;
  float pixel_uncertainty, quantile_cutoff;
  int quantile_num_bins;
  float reshape_q_abs;
  float pixel_uncertainty_match, match_threshold_norm, cang_min_edge, regularization_threshold;
  int nseg;  // ceil(cols / 64)
  int df_nr; // 2 * search_range
  unsigned long long* dbg;  // REBVIO_HIP_LM_STAMPS: pinned stamp buffer (null otherwise); [48..50] = start of the pair's second-half kernels
  unsigned long long* dm_stats;  // REBVIO_HIP_DM_STATS: device buffer, one 16-word record per wave of k_directed_match_c (workload + phase times)
};

// Levenberg-Marquardt state of minimizeVel kept on the device (core.cpp:150-189).
struct LmState {
  float vel[3];
  float F;
  float JtJ[6];  // (0,0) (1,1) (2,2) (0,1) (0,2) (1,2)
  float JtF[3];
  float u, v;
  float Vnew[3];
  float h[3];
  float sigma_rho_min;
  int accept_mask;
  int pad[4];
};

// x86 cvttss2si / cvttsd2si semantics (NaN / out of range -> INT_MIN), see oracle header.
__host__ __device__ inline int cvtt_f32(float v) {
  if (!(v >= -2147483648.0f && v < 2147483648.0f)) return (int)0x80000000;
  return (int)v;
}
__host__ __device__ inline int cvtt_f64(double v) {
  if (!(v > -2147483649.0 && v < 2147483648.0)) return (int)0x80000000;
  return (int)v;
}

// ---- launchers (defined in detect.hip / track.hip) ------------------------------------------------
struct ScaleBufs {
  float* a[2];  // per filter: scan buffer A
  float* b[2];  // per filter: scan buffer B
  float* dog;
  float* mag;
  float* scale0;  // optional diagnostics (may be null)
  float* scale1;
};

struct DetectBufs {
  float4* stash;              // per pixel plane-fit result of candidates
  unsigned long long* bits;   // [rows][nseg] candidate ballots
  int* rowcount;              // [rows]
};

void upload_tables(const float* recip128, const float* pinv75);

void launch_front_end_u8(hipStream_t s, const KParams& p, const uint8_t* src, const int2* map, float* dst);
void launch_copy_from_pinned(hipStream_t s, const void* src_pinned, void* dst_dev, size_t bytes);
void launch_scale_space(hipStream_t s, const KParams& p, const void* img, int img_is_u8, const ScaleBufs& sb,
                        const int widths[2][3], int* rowcount_to_zero, int part = 3, bool fuse_dog = false);
void launch_smooth_n(hipStream_t s, const KParams& p, const float* img, const ScaleBufs& sb, const int* widths, int n, int* rowcount_to_zero);
void launch_keylines(hipStream_t s, const KParams& p, const ScaleBufs& sb, const DetectBufs& db, const MapDev& m,
                     const DetState* det_in, DetState* det_out, const MapState* prev_st, const int* fuse_widths = nullptr);
// Tile grid of the keyline-driven distance-field build (shared by the binning pass in k_join_edges and the tile kernel).
struct DfGrid {
  int T, ntx, nty;
};
inline DfGrid df_grid(int rows, int cols) {
  DfGrid g;
  g.T = kDfTile;
  while (((cols + g.T - 1) / g.T) * ((rows + g.T - 1) / g.T) > kDfMaxTiles) g.T *= 2;
  g.ntx = (cols + g.T - 1) / g.T;
  g.nty = (rows + g.T - 1) / g.T;
  return g;
}
// mask_is_current: the map is as detection left it (raster order, tile lists from k_join_edges): the tile kernel builds the
// field; after rebvio_hip_map_upload the scatter kernel rebuilds it from the keyline arrays (it needs a cleared field).
void launch_df_build(hipStream_t s, const KParams& p, const MapDev& m, const DetState* det_prev, bool mask_is_current);
void launch_df_decode(hipStream_t s, const KParams& p, const MapDev& m, int* id_out, int* dist_out);

void launch_rotate(hipStream_t s, const KParams& p, const MapDev& m, const float R[9], int* hist_or_null,
                   int zero_dm_counters);
void launch_quantile(hipStream_t s, const KParams& p, const MapDev& m, int* hist, float pct, int bins, float* out_dev);
void launch_try_vel(hipStream_t s, const KParams& p, const MapDev& oldm, const MapDev& newm, int mode_lm, int call,
                    int last, LmState* st_in, LmState* st_out, const float* part_prev, float* part_out, const int* hist,
                    int frame_count);
// Host-visible result slot of one frame pair, written directly by k_ext_rot_vel (zero-copy, pinned memory).
struct PairSlot {
  LmState lm;          // final minimizeVel state
  MapState new_st;     // snapshot of the new map's scalars
  MapState old_st;     // snapshot of the old map's scalars (directedMatch / regularize counters of the previous pair)
  unsigned seq;        // sequence stamp of the pair that wrote this slot: the LAST word its kernel stores, so a host that reads the
                       // slot before the pair has run sees the previous user's stamp and reports -12
  unsigned sum;        // XOR of every word of lm / new_st / old_st, ^ seq: a host that POLLS the slot (rebvio_hip_track_pair_begin) instead
  unsigned pad_[2];    // of waiting for an event accepts it only when the words it read add up (writes to host memory may arrive out of order)
  float xrv[1];        // [nblocks][kXrvStride] block records follow
};
// Result of the host glue of one pair (rebvio.cpp:186-233) as the second half of the pair step reads it from memory
// (streaming driver): the host writes it into pinned memory and releases the stream (hipStreamWaitValue32 on a pinned
// flag); the first kernel behind the wait reads it in place and leaves a device copy for the kernels after it.
struct GlueDev {
  float vel_r[3];    // Rback * V            (edge_map.cpp:193)
  float Rvel_r[9];   // Rback * P_V * Rback' (edge_map.cpp:194)
  float Rgva[9];     // Rback
  float R0a[9];      // second rotateKeylines of the old map, applied on the fly by directedMatch
  float V[3];        // translation for the depth EKF
  float RT_next[9];  // first rotateKeylines of the NEXT pair (new map becomes its old map)
  int nan_v;         // rebvio.cpp:236: no matching / regularisation / depth update for this pair
  int has_next;      // RT_next valid
};
// Gyro-bias filter state of the glue (types/imu.hpp:180-183) plus the prior rotation the pair's first rotateKeylines applied.
struct GlueState {
  float Bg[3];
  float W_Bg[9];
  float R[9];
  float pad;
};

// Everything the glue of one pair produces for the host (pinned record, read a few pairs later).
struct GlueRec {
  rebvio_hip_pair_out out;  // counters / status are filled in by the host from the map state records
  GlueState gs;             // state AFTER this pair (the host mirrors it: rebvio_hip_get_gyro_state)
  unsigned seq_gs, seq_out; // sequence stamps behind `gs` and behind `out` (two waves of the glue write them; see PairSlot::seq)
};

// Where the device glue leaves a pair's host record when a directedMatch launch follows it (streaming and batch drivers): device
// memory, one per result slot. Stores into pinned host memory at the tail of the LM kernel sat on the pair's critical path (a PCIe
// write latency before the kernel could end, and a release in front of the sequence stamp on top: 3 us per pair measured); the
// last wave of the directedMatch launch - which owns no keyline - copies the record to the host and stamps it while the rest of
// that launch works (directed_match_c_body).
struct GlueStage {
  GlueRec rec;          // as the glue wrote it (stamps unset)
  GlueRec* host_rec;    // pinned destination
  PairSlot* host_slot;  // the pair's result slot, stamped with the record (its contents were stored by the LM kernel itself)
  unsigned seq;
  unsigned pad;
};

struct GlueParams {
  float frame_dt;
  float gyro_std_dev, gyro_bias_std_dev;
  // The 3x3 matrices of gyroBiasCorrection (core.cpp:264-270, 282) depend on the filter's information matrix W_Bg and on
  // frame_dt only - never on the pair's data - and W_Bg itself follows the recurrence W_Bg <- Wg + invert(invert(W_Bg) + Rb).
  // The streaming driver keeps a host shadow of W_Bg and hands the pair's matrices over by value (hm::gyro_pre, the same
  // statements as gyro_bias_correction): four dependent 3x3 inversions and two products leave the device glue's critical
  // path (wave 1 of glue_workgroup, ~1.5 us per pair). has_pre == 0: the device forms them itself (batches, per-call API).
  // [0] Wg  [1] Wb' = invert(invert(W_Bg) + Rb)  [2] Wg + Wb'  [3] iWgWb  [4] upd  [5] (Wg iWgWb) Wb'
  int has_pre;
  float pre[6][9];
};

// Device glue of a pair, run by workgroup 0 of the persistent LM kernel behind its last phase (glue_dev.hpp). lm == null: no
// device glue - the extRotVel records go to memory for the host (per-pair API: the host runs the glue).
struct GlueArgs {
  const LmState* lm;       // final minimizeVel state of this pair (device memory)
  const float* xrv;        // extRotVel block records of this pair (device memory, kXrvStride floats per record group)
  const GlueState* st_in;  // filter state before this pair ...
  GlueState* st_out;       // ... and after it (the other parity slot: late workgroups still read st_in)
  GlueRec* rec;            // pinned host record of this pair
  GlueDev* gd_copy;        // device copy of the second half's inputs for the kernels queued behind the head
  GlueStage* stage;        // != null: the host record goes here and the directedMatch launch forwards it (GlueStage); null: straight to *rec
  unsigned seq;            // the pair's sequence stamp (PairSlot::seq, GlueRec::seq_gs / seq_out); also used when lm == null
  GlueParams gp;
};
void launch_ext_rot_vel(hipStream_t s, const KParams& p, const MapDev& oldm, const MapDev& newm, int do_forward,
                        int do_lm_final, int calls, LmState* st_in, LmState* st_out, const float* part_prev,
                        float* xrv_part, const float* vel_manual, PairSlot* slot, int* hist_to_zero, unsigned seq = 0u);
void launch_forward_keys(hipStream_t s, const KParams& p, const MapDev& oldm, const MapDev& newm);
// Persistent minimizeVel (+ forwardMatch + extRotVel when do_ext): one launch; the workgroups exchange their records
// through `xch` ([2][record groups][kPartStride] 64-bit words {tag, value}). The caller passes tag_base = tags consumed so
// far and advances it by 2 * (calls + 1) per launch. threads = workgroup size 256 / 512 / 1024 (REBVIO_HIP_LM_THREADS, read when
// the context is created; default 512).
void launch_lm_chain(hipStream_t s, const KParams& p, const MapDev& oldm, const MapDev& newm, int calls, int do_ext, LmState* st_in,
                     LmState* st_out, unsigned long long* xch, unsigned tag_base, int* bar_err, const int* hist, float* xrv_part,
                     PairSlot* slot, int* hist_to_zero, unsigned long long* stamps, int threads, const GlueArgs& ga);
void launch_lm_final(hipStream_t s, const MapDev& oldm, int calls, LmState* st_in, LmState* st_out, const float* part_prev);
// lanes whose persistent LM workgroups (512 threads, ceil(kmax / 512) per lane) the device holds resident at once
int lm_chain_b_max_lanes(int device, int kmax, int calls);
// ... and the workgroups of 512 threads that is (occupancy of the batched LM kernels x CUs, one block per CU taken off)
int lm_chain_b_capacity_wgs(int device, int kmax, int calls);
// the device glue as a launch of its own, behind k_ext_rot_vel (REBVIO_HIP_LM=percall: no persistent kernel to run it in)
void launch_pair_glue(hipStream_t s, const MapDev& newm, const GlueArgs& ga);
void launch_directed_match(hipStream_t s, const KParams& p, const MapDev& newm, const MapDev& oldm, const float vel[3],
                           const float Rvel[9], const float Rback[9], float max_radius, const float* R0_on_the_fly, int head_form);
// the same launch with the second half's inputs read from *gd (device memory) at run time
void launch_directed_match_dev(hipStream_t s, const KParams& p, const MapDev& newm, const MapDev& oldm, const GlueDev* gd,
                               const GlueStage* stage, float max_radius, int head_form);
void launch_regularize_ekf_dev(hipStream_t s, const KParams& p, const MapDev& m, const GlueDev* g_dev, int gate, int* hist);
// fused regularize1Iter + depth EKF: reads m.rs, writes m.rs_tmp (caller swaps the pointers); Rnext != null also
// applies the next pair's first rotation and bins sigma_rho into hist
void launch_regularize_ekf(hipStream_t s, const KParams& p, const MapDev& m, const float vel[3], int gate, const float* Rnext, int* hist);
void launch_search_match_one(hipStream_t s, const KParams& p, const MapDev& searched, const rebvio_hip_keyline& query,
                             const float vel[3], const float Rvel[9], const float Rback[9], float max_radius, int* out_dev);
void launch_regularize(hipStream_t s, const KParams& p, const MapDev& m, int min_matches_gate);
void launch_depth_ekf(hipStream_t s, const KParams& p, const MapDev& m, const float vel[3], int use_tmp,
                      int min_matches_gate);
void launch_render_edge_image(hipStream_t s, const KParams& p, const MapDev& m, const uint8_t* gray_or_null, uint8_t* rgb);
void launch_map_pack(hipStream_t s, const KParams& p, const MapDev& m, rebvio_hip_keyline* aos_dev);
void launch_map_unpack(hipStream_t s, const KParams& p, const MapDev& m, const rebvio_hip_keyline* aos_dev, int n);

inline int div_up(int a, int b) { return (a + b - 1) / b; }

// ---- several camera streams ("lanes") of one GPU advanced in lock-step by batched launches (rebvio_hip_batch_*) --------
// A batched kernel is the single-stream kernel body run for lane = blockIdx.z; what differs between lanes comes from two
// tables: LaneStatic (device memory, fixed for the life of the lane: its scratch buffers) and the lane's map table
// (device memory, one MapDev per pooled edge map), plus a small by-value LaneDyn per launch (which maps, which ring slots).
constexpr int kMaxLanes = 16;
constexpr int kPairSlots = 16;  // result slots of the streaming / batch drivers: pairs in flight between the device and the host
constexpr int kLaneMaps = 24;  // map-table entries per lane (the batch driver bounds its pool by this)
// Scale-space outputs (DoG, squared gradient, per-row counts, undistorted frame) exist kDetPar times per context: the scans of
// step k + 1 overlap the keyline extraction of step k. Two is enough: with four, an 8-lane batch ran at the same rate (38.8 k
// frames/s either way; the scan stream's wait for the keyline stage of step k - kDetPar is not what limits a batched step,
// DESIGN.md 6c).
constexpr int kDetPar = 2;
struct LaneStatic {
  float* sa[2];   // scan buffer A per filter
  float* sb[2];   // scan buffer B per filter
  float* dog2[kDetPar];
  float* mag2[kDetPar];
  int* rowcount2[kDetPar];
  float4* stash;
  unsigned long long* bits;
  DetState* det;  // ring [kDetRing + 1]
  LmState* lm;
  LmState* lm_zero;
  unsigned long long* lm_xch;
  int* lm_bar_err;
  int* hist;
  PairSlot* slot[kPairSlots];  // pinned: LM state + map state records of a pair, written by the LM kernel
  GlueRec* rec[kPairSlots];    // pinned: what the device glue of a pair reports
  GlueDev* glue_dev;           // [kPairSlots] second-half inputs left by the device glue
  GlueStage* glue_stage;       // [kPairSlots] host records on their way out (GlueStage)
  GlueState* gstate;           // [2] gyro-bias filter state + prior rotation, by pair parity
  float* xrv_part;             // extRotVel block records of the pair in flight
  const int2* undist_map;      // the lane's lens model (fixed-point source coordinates), null without one
  float* undist_img[kDetPar];  // x3 + undistorted fp32 frame by step parity (step % kDetPar)
};
struct LaneDyn {
  const void* img;              // u8 frame of this step (device memory)
  short nm, om, prev;           // map-table indices: detected / new map, old map, previously detected map (-1: none)
  unsigned char nm_swap, om_swap;  // bit 0: rs <-> rs_tmp, bit 1: grad <-> grad_tmp relative to the table entry
  unsigned char parity, det_in, det_out, slot;
  unsigned char gpar;           // parity slot of the glue state this pair reads (it writes the other one)
  unsigned tag_base;
  unsigned seq;                 // the pair's sequence stamp (PairSlot::seq, GlueRec::seq_gs / seq_out)
};
struct LaneDynB {
  LaneDyn v[kMaxLanes];
};
__host__ __device__ inline MapDev lane_map(const MapDev* __restrict__ tab, int lane, int idx, unsigned swap) {
  MapDev m = tab[lane * kLaneMaps + idx];
  if (swap & 1u) {
    float2* t = m.rs;
    m.rs = m.rs_tmp;
    m.rs_tmp = t;
  }
  if (swap & 2u) {
    float2* t = m.grad;
    m.grad = m.grad_tmp;
    m.grad_tmp = t;
  }
  return m;
}
#if defined(__HIPCC__)
// Sequence stamp of a host-visible record: stored behind everything else the writer stored (system-scope release).
// The writer waits until everything it has stored itself is acknowledged (s_waitcnt vmcnt(0): stores to host-visible memory are
// written through) and stores the stamp relaxed: a system-scope RELEASE here writes the XCD's L2 back first - +3 us per pair when
// it sat at the tail of the LM kernel (in-kernel stamps, round 4).
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx942__) && !defined(__gfx950__)
#error "common.hpp: stamp_drain() encodes s_waitcnt for gfx942 / gfx950 only"
#endif
// Fills a result slot and returns the XOR of the words stored (PairSlot::sum ^ seq, stored with the stamp at the kernel's end).
__device__ __forceinline__ unsigned slot_fill(PairSlot* slot, const LmState& lm, const MapState* new_st, const MapState* old_st) {
  const LmState l = lm;
  const MapState a = *new_st, b = *old_st;
  slot->lm = l;
  slot->new_st = a;
  slot->old_st = b;
  unsigned x = 0u;
  const unsigned* p = reinterpret_cast<const unsigned*>(&l);
#pragma unroll
  for (int i = 0; i < (int)(sizeof(LmState) / 4); ++i) x ^= p[i];
  p = reinterpret_cast<const unsigned*>(&a);
#pragma unroll
  for (int i = 0; i < (int)(sizeof(MapState) / 4); ++i) x ^= p[i];
  p = reinterpret_cast<const unsigned*>(&b);
#pragma unroll
  for (int i = 0; i < (int)(sizeof(MapState) / 4); ++i) x ^= p[i];
  return x;
}
__device__ __forceinline__ void stamp_drain(unsigned* w, unsigned seq) {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // (compiler ordering; no cache operation outside tgsplit mode)
  __builtin_amdgcn_s_waitcnt(0x0F70);                     // vmcnt(0); expcnt / lgkmcnt untouched
  __hip_atomic_store(w, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
#endif
#if defined(__HIPCC__)
// ---- pointers that reach a kernel through a table in memory -----------------------------------------------------------------
// A pointer passed as a kernel argument (alone or inside a by-value struct such as MapDev) is known to the compiler to be
// global and gets global_load / global_store. One that a batched kernel LOADS from a table - LaneStatic, the lanes' map table -
// is a generic pointer: every access through it becomes a FLAT instruction, which counts on BOTH the vector-memory and the LDS
// counter (an LDS wait then also waits for every load in flight: the staging loops of the per-pixel kernels lose the overlap
// they were written for) and cannot use the scalar-base addressing form. gptr() states what the table holds: device or pinned
// memory, never LDS or scratch. (ISA of the batched kernels before: 0 global / 7..129 flat loads each; the single-stream
// kernels: all global.)
template <class T>
__device__ __forceinline__ T* gptr(T* p) {
  // The value is the same for every thread of the workgroup (read from a table entry chosen by the lane index), so it lives in
  // scalar registers. A plain cast to the global address space and back is folded away before address spaces are inferred, and
  // assumptions on __builtin_amdgcn_is_shared / _is_private did not take either (hipcc of ROCm 7.2); passing the typed pointer
  // through an empty asm keeps the type, and the "s" constraint keeps the scalar base for the saddr addressing form.
  typedef __attribute__((address_space(1))) T* G;
  G q = (G)p;
  asm("" : "+s"(q));
  return (T*)q;
}
__device__ __forceinline__ MapDev global_map(MapDev m) {
  m.pos = gptr(m.pos); m.pos_img = gptr(m.pos_img); m.mpos_img = gptr(m.mpos_img); m.grad = gptr(m.grad); m.mgrad = gptr(m.mgrad);
  m.gnorm = gptr(m.gnorm); m.mgnorm = gptr(m.mgnorm); m.rs = gptr(m.rs); m.rs_tmp = gptr(m.rs_tmp); m.grad_tmp = gptr(m.grad_tmp);
  m.id_prev = gptr(m.id_prev); m.id_next = gptr(m.id_next); m.match_id = gptr(m.match_id); m.match_fwd = gptr(m.match_fwd);
  m.match_kf = gptr(m.match_kf); m.matches = gptr(m.matches); m.fwd_key = gptr(m.fwd_key); m.residual = gptr(m.residual);
  m.mask = gptr(m.mask); m.df = gptr(m.df); m.unit = gptr(m.unit); m.tile_cnt = gptr(m.tile_cnt); m.tile_list = gptr(m.tile_list);
  m.row_start = gptr(m.row_start); m.st = gptr(m.st);
  return m;
}
#endif
#if defined(__HIPCC__)
// ---- XCD-aware tile order ------------------------------------------------------------------------------------------------
// The hardware deals the workgroups of a launch to the 8 XCDs round-robin in the order of their linear index (observed,
// MI355X_MICROARCH.md "Workgroup dispatch": blocks b and b + 8 share an XCD; speed only, nothing depends on it for
// correctness), and every XCD has its own 4 MB L2. With tile = blockIdx, neighbouring tiles of an image always sit on
// DIFFERENT XCDs: the halo rows / columns two tiles share, and the two 64-byte halves of a 128-byte line that two 16-column
// strips of k_colscan share, are fetched into two L2s (rocprofv3 FETCH_SIZE: k_colscan_b and k_rowscan_b<2> fetched 2x their
// algorithmic bytes, k_dog_mag_b 2.8x). Here the workgroups of one XCD take a CONTIGUOUS band of tiles instead: with n tiles
// in row-major order, workgroup t (its class c = t % 8 names its XCD, up to a per-launch constant) takes tile
//   c * (n / 8) + min(c, n % 8) + t / 8
// - a bijection on [0, n) - so that what neighbouring tiles share is found in the L2 that already holds it. The per-keyline
// kernels (one workgroup per 256 keylines in raster order) use it with gridDim.y = 1: neighbouring keyline blocks probe the
// same mask rows and old-map keylines. NOT used by the persistent LM kernels (their workgroup index is the record group's).
__device__ __forceinline__ uint2 xcd_band_block() {
  const unsigned gx = gridDim.x, n = gx * gridDim.y;
  const unsigned t = blockIdx.x + gx * blockIdx.y;
  const unsigned c = t & 7u;
  const unsigned tile = c * (n >> 3) + min(c, n & 7u) + (t >> 3);
  return make_uint2(tile % gx, tile / gx);
}

#endif
void launch_scale_space_b(hipStream_t s, const KParams& p, int lane0, int lanes, const LaneStatic* ls, const LaneDynB& dyn,
                          const int widths[2][3], bool lens, bool fuse_dog = false);
void launch_keylines_b(hipStream_t s, const KParams& p, int lanes, const LaneStatic* ls, const MapDev* maptab, const LaneDynB& dyn,
                       const int* fuse_widths = nullptr);
void launch_df_build_b(hipStream_t s, const KParams& p, int lanes, const LaneStatic* ls, const MapDev* maptab, const LaneDynB& dyn);
void launch_lm_chain_b(hipStream_t s, const KParams& p, int lanes, int lanes_per_launch, const LaneStatic* ls, const MapDev* maptab,
                       const LaneDynB& dyn, int calls, int spec, const GlueParams& gp);
void launch_b_chain_b(hipStream_t s, const KParams& p, int lanes, const LaneStatic* ls, const MapDev* maptab, const LaneDynB& dyn,
                      float max_radius, int gate, int head_form);

// Optional per-kernel timing with HIP events recorded on the launching stream (api.hip).
void prof_begin(hipStream_t s, const char* name);
void prof_end(hipStream_t s);
void prof_group_begin(hipStream_t s, const char* name, int count);
void prof_group_end(hipStream_t s);
struct ProfScope {
  hipStream_t s;
  ProfScope(hipStream_t s_, const char* name) : s(s_) { prof_begin(s, name); }
  ~ProfScope() { prof_end(s); }
};
#define RH_LAUNCH(kernel, grid, block, shm, stream, ...)                 \
  do {                                                                   \
    ::rh::ProfScope _ps(stream, #kernel);                                \
    hipLaunchKernelGGL(kernel, grid, block, shm, stream, __VA_ARGS__);   \
  } while (0)

// the same for a kernel whose template arguments hold a comma: the name is given separately, the kernel in parentheses
#define RH_LAUNCH_NAMED(name, kernel, grid, block, shm, stream, ...)      \
  do {                                                                   \
    ::rh::ProfScope _ps(stream, name);                                   \
    hipLaunchKernelGGL(kernel, grid, block, shm, stream, __VA_ARGS__);   \
  } while (0)

}  // namespace rh
