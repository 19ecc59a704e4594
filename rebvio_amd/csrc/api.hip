// C-ABI of the gfx950 backend (include/rebvio_hip.h): context / edge-map pool management, stream ordering
// between the detect stream and the track stream, and the O(1) host glue of one frame pair. All heavy work
// is in detect.hip / track.hip; nothing here falls back to a CPU implementation of the hot path.
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <atomic>
#include <condition_variable>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <shared_mutex>
#include <thread>
#include <string>
#include <vector>

#include "common.hpp"
#include "glue.hpp"
#include "hostmath.hpp"

using namespace rh;

namespace {
thread_local std::string g_err;
// streams a context under construction adopts instead of creating its own (lanes of a batch share one set)
struct SharedStreams {
  hipStream_t s_det, s_key, s_trk;
};
thread_local const SharedStreams* t_adopt_streams = nullptr;

// ---- device-wide registry of the persistent LM kernels' users (this process) -------------------------------------------------
// The workgroups of a persistent LM launch wait for each other's records, so all of them have to be resident together; the
// occupancy check of a batch (lm_chain_b_max_lanes) assumes it has the device to itself. Every context with the persistent
// kernel and every batch registers here when it is created; a batch sizes each LM launch to ITS SHARE of the device - capacity /
// live users on that GPU - so that two batches (or a batch beside single streams) still fit together: slower, not timed out.
// Kernels of other processes are invisible to this registry: what remains for them is the bounded poll (-9).
struct Residency {
  std::mutex mu;
  std::map<int, std::map<const void*, int>> users;  // device -> owner -> workgroups of its largest LM launch
};
Residency& residency() {
  static Residency r;
  return r;
}
void residency_add(int device, const void* owner, int wgs) {
  std::lock_guard<std::mutex> lk(residency().mu);
  residency().users[device][owner] = wgs;
}
void residency_remove(int device, const void* owner) {
  std::lock_guard<std::mutex> lk(residency().mu);
  auto it = residency().users.find(device);
  if (it != residency().users.end()) it->second.erase(owner);
}
int residency_users(int device) {
  std::lock_guard<std::mutex> lk(residency().mu);
  auto it = residency().users.find(device);
  return it == residency().users.end() ? 0 : (int)it->second.size();
}
std::string residency_describe(int device, const void* self) {
  std::lock_guard<std::mutex> lk(residency().mu);
  auto it = residency().users.find(device);
  int others = 0, wgs = 0;
  if (it != residency().users.end())
    for (auto& u : it->second)
      if (u.first != self) {
        ++others;
        wgs += u.second;
      }
  char msg[200];
  std::snprintf(msg, sizeof(msg), "%d other context(s) / batch(es) of this process run persistent LM kernels on GPU %d (up to %d workgroups "
                "together); kernels of other processes are not visible to this check", others, device, wgs);
  return msg;
}

// REBVIO_HIP_DM_HEAD / REBVIO_HIP_BATCH_DM_HEAD -> form of the directedMatch launch (track.hip: dm_head_wide, dm_compact)
int dm_form_by_name(const char* e) {
  static const char* const names[] = {"", "compact8", "compact4", "compact1"};
  for (int i = 1; i < 4; ++i)
    if (std::strcmp(e, names[i]) == 0) return i;
  return 0;
}

int fail(const char* what, hipError_t e) {
  g_err = std::string(what) + ": " + hipGetErrorString(e);
  return -(int)e - 1000;
}
int fail_msg(const char* what, int code) {
  g_err = what;
  return code;
}
#define HIPCHK(expr)                          \
  do {                                        \
    hipError_t _e = (expr);                   \
    if (_e != hipSuccess) return fail(#expr, _e); \
  } while (0)

// ---- profiler ------------------------------------------------------------------------------------
struct Profiler {
  bool on = false;
  std::string only;
  int stride = 1;        // record every stride-th launch of a kernel (keeps the timed region cheap)
  std::map<std::string, unsigned> seen;
  struct Sample {
    int name;
    hipEvent_t e0, e1;
    int count;  // launches bracketed by this event pair
  };
  std::vector<std::string> names;
  std::vector<Sample> samples;
  std::vector<hipEvent_t> free_events;
  std::map<int, std::pair<double, int>> acc;  // name -> (sum us, calls)
  size_t cap = 1 << 16;
  std::mutex mu;

  int name_id(const char* n) {
    for (size_t i = 0; i < names.size(); ++i)
      if (names[i] == n) return (int)i;
    names.emplace_back(n);
    return (int)names.size() - 1;
  }
  hipEvent_t get_event() {
    if (!free_events.empty()) {
      hipEvent_t e = free_events.back();
      free_events.pop_back();
      return e;
    }
    hipEvent_t e;
    (void)hipEventCreate(&e);
    return e;
  }
  void drain() {
    for (auto& s : samples) {
      float ms = 0.f;
      if (hipEventSynchronize(s.e1) == hipSuccess && hipEventElapsedTime(&ms, s.e0, s.e1) == hipSuccess) {
        auto& a = acc[s.name];
        a.first += (double)ms * 1e3;
        a.second += s.count;
      }
      free_events.push_back(s.e0);
      free_events.push_back(s.e1);
    }
    samples.clear();
  }
};
Profiler g_prof;
// The open bracket belongs to the launching THREAD (the caller thread and the detect worker launch concurrently).
struct ProfOpen {
  int group_depth = 0;  // inside prof_group_begin/end the individual launches are not bracketed again
  int count = 1;
  int name = -1;
  hipEvent_t e0{}, e1{};
};
thread_local ProfOpen t_open;
}  // namespace

namespace rh {
// selection by exact name, or by prefix when the selector ends with '*'
static bool prof_name_selected(const std::string& only, const char* name) {
  if (!only.empty() && only.back() == '*') return std::strncmp(only.c_str(), name, only.size() - 1) == 0;
  return only == name;
}
void prof_begin(hipStream_t s, const char* name) {
  if (!g_prof.on) return;
  if (t_open.group_depth > 0) return;
  t_open.count = 1;
  t_open.name = -1;
  {
    std::lock_guard<std::mutex> g(g_prof.mu);
    if (!g_prof.only.empty() && !prof_name_selected(g_prof.only, name)) return;
    if (g_prof.stride > 1 && (g_prof.seen[name]++ % (unsigned)g_prof.stride) != 0) return;
    if (g_prof.samples.size() >= g_prof.cap) return;
    t_open.name = g_prof.name_id(name);
    t_open.e0 = g_prof.get_event();
    t_open.e1 = g_prof.get_event();
  }
  (void)hipEventRecord(t_open.e0, s);
}
void prof_end(hipStream_t s) {
  if (!g_prof.on) return;
  if (t_open.group_depth > 0) return;
  if (t_open.name < 0) return;
  (void)hipEventRecord(t_open.e1, s);
  std::lock_guard<std::mutex> g(g_prof.mu);
  g_prof.samples.push_back({t_open.name, t_open.e0, t_open.e1, t_open.count});
  t_open.name = -1;
}
// One event pair around `count` back-to-back launches of the same kernel on one stream: the fixed cost of an event
// pair (~4.5 us measured around an empty kernel) is amortised, so the per-launch average agrees with rocprofv3's.
void prof_group_begin(hipStream_t s, const char* name, int count) {
  if (!g_prof.on) return;
  prof_begin(s, name);
  t_open.count = count;
  t_open.group_depth = 1;
}
void prof_group_end(hipStream_t s) {
  if (!g_prof.on) return;
  t_open.group_depth = 0;
  prof_end(s);
}
}  // namespace rh

// ---- objects ----------------------------------------------------------------------------------------
// Shared by a context and every map handle it has handed out: a map released (or queried) after rebvio_hip_destroy must not
// touch the freed context. The block outlives the context for as long as a handle is still out; `dead` is set by destroy,
// under `mu`, which also serialises destroy against a release from another thread (an EdgeMap::SharedPtr kept by an
// edge-image consumer, ros_rebvio.cpp:32-51, is released whenever that consumer lets go of it).
struct LifeBlock {
  std::shared_mutex mu;  // shared: every entry point that takes a map alone, for its whole duration; exclusive: rebvio_hip_destroy
  std::atomic<bool> dead{false};
};

struct rebvio_hip_map {
  rebvio_hip_ctx* ctx = nullptr;
  std::shared_ptr<LifeBlock> life;
  MapDev d{};
  bool in_use = false;
  uint64_t ts = 0;
  hipEvent_t detected{};  // keylines + mask + chaining finished (detect stream)
  hipEvent_t ready{};     // ... and distance field built (distance-field stream)
  hipEvent_t done{};   // last track-stream consumer finished, recorded at release
  hipEvent_t done_ref{};  // ... or (streaming driver) the result-slot event of the pair that used the map last: no packet of its own
  bool has_done = false;
  uint64_t release_seq = 0;  // order of release (pool reuse is oldest-first)
  bool df_built = false;
  bool raster_order = false;  // keylines + row_start as detection left them (false after map_upload)
  std::atomic<int> enqueued{1};  // 0 while the detect worker still has to record `ready` (streaming driver)
  bool pre_rotated = false;  // the next pair's first rotateKeylines (+ histogram) was already applied by the fused B-chain
  int n_host = -1;
  float thr_host = -1.0f;
  bool trk_waited = false;  // the track stream already holds a wait on `ready` (a second one is another barrier packet)
  // set once any track-stream operation referencing this map has been enqueued: until then a download only has to wait for
  // the map's own detection (`ready`), not for the track stream (which may hold another pair's parked second half)
  std::atomic<bool> trk_touched{false};
  hm::M3 pre_R{};           // ... with this rotation (per-pair API: checked against the prior the next _begin is given)
  int tab_idx = -1;         // entry of this map in its lane's device map table (batch driver)
  MapDev canon{};           // ... as uploaded there (the live `d` differs from it by the ping-pong swaps only)
};

struct rebvio_hip_ctx {
  std::shared_ptr<LifeBlock> life = std::make_shared<LifeBlock>();
  rebvio_hip_params P{};
  KParams K{};
  int device = 0;
  hipStream_t s_det{}, s_key{}, s_trk{};  // scans | keylines + distance field (+ the synchronous API's copies) | tracking
  double t_begin_enq = 0, t_begin_wait = 0;  // REBVIO_HIP_DEBUG: host time of track_pair_begin (enqueue / wait for the first half)
  uint64_t t_begin_n = 0;
  std::mutex dl_mu;             // aos_dev / scratch_i users (map_download, map_upload, render, field decode) and trk_touched
  float* sa2[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};  // [frame parity][filter]: sb.a per parity ([0] aliases sb.a)
  // scale-space outputs (DoG, squared gradient, per-row counts) are double buffered: the scans of frame f+1 (s_det)
  // overlap the keyline extraction of frame f (s_key)
  float* dog2[kDetPar]{};  // (the single-stream driver alternates between [0] and [1], a batch uses all: common.hpp kDetPar)
  float* mag2[kDetPar]{};
  int* rowcount2[kDetPar]{};
  hipEvent_t ev_scan[2]{};
  // `ready` event of the map detected two frames ago (same parity): what the scans of this frame wait for before they rewrite
  // that frame's buffers. It is recorded a few microseconds behind the point the buffers are free (after the distance field
  // instead of after joinEdges), two frame times before anyone waits for it - and it is an event the frame records anyway: one
  // hipEventRecord less per frame on the launching thread (3.5 us of its 62, round 4).
  hipEvent_t prev_ready[2]{};
  uint64_t launch_index = 0;
  uint64_t release_counter = 0;
  ScaleBufs sb{};
  DetectBufs db{};
  DetState* det = nullptr;  // [kDetRing] servo-state ring + [kDetRing]: scratch sink
  uint64_t frame_index = 0;
  rebvio_hip_map* last_detected = nullptr;
  int widths[2][3]{};
  std::vector<rebvio_hip_map*> pool;
  float* img_dev = nullptr;
  uint8_t* img8_dev = nullptr;
  static constexpr int kPin = 16;  // pinned staging ring of the host-frame detect entries (allocated on first use)
  void* pin[kPin]{};
  hipEvent_t pin_ev[kPin]{};
  bool pin_used[kPin]{};
  std::atomic<int> pin_staged[kPin]{};  // 1: the slot holds a frame whose host-to-device copy the detect worker has not queued yet
  uint64_t pin_next = 0;
  int2* undist_map = nullptr;      // fixed-point source coordinates (null: no lens distortion, front end = x3 only)
  float* undist_img[kDetPar]{};    // undistorted fp32 frame, buffered like dog2 / mag2
  rebvio_hip_keyline* aos_dev = nullptr;
  int* scratch_i = nullptr;  // 2 * rows*cols ints (df decode)
  float* diag0 = nullptr;
  float* diag1 = nullptr;
  // tracking
  LmState* lm = nullptr;    // [16]
  float* part = nullptr;    // [kMaxLmCalls+1][maxblocks][kPartStride]
  float* xrv_part = nullptr;
  int* hist = nullptr;      // [128]
  float* fscratch = nullptr;
  int maxblocks = 0;
  // persistent LM kernel: record exchange words, tags consumed so far, sticky time-out flag
  unsigned long long* lm_xch = nullptr;
  unsigned lm_tag_base = 0;
  bool lm_spec = true;          // the speculative kernel may be used ...
  bool lm_spec_forced = false;  // ... always (REBVIO_HIP_LM=spec / spec3), instead of by the stream's recent miss rates
  int lm_spec_forced_kf = 2;    // first speculative evaluation of the forced form
  int lm_mix = 0;               // REBVIO_HIP_LM=mix<seed> (seed >= 1): see lm_kernel_choice
  float lm_miss_ema = 0.f;      // recent rate of "a step after the first was accepted" (the hypothesis of kf = 2 failed)
  float lm_miss3_ema = 0.f;     // ... "a step after the second was accepted" (the hypothesis of kf = 3 failed)
  int* lm_bar_err = nullptr;  // pinned, zero-copy
  unsigned long long* lm_stamps = nullptr;  // pinned; REBVIO_HIP_LM_STAMPS diagnostic (phase stamps of workgroup 0)
  unsigned long long* dm_stats = nullptr;   // device; REBVIO_HIP_DM_STATS diagnostic (workload shape + per-wave phase times of k_directed_match_c)
  size_t dm_stats_words = 0;
  double lm_stamp_acc[64]{};
  uint64_t lm_stamp_n = 0;
  bool lm_stamp_spec = false;
  bool lm_persistent = true;
  bool residency_registered = false;  // this context launches persistent LM kernels of its own (Residency)
  rebvio_hip_map* df_map = nullptr;
  // pinned host staging
  LmState* h_lm = nullptr;   // [2]
  float* h_part = nullptr;
  float* h_xrv = nullptr;
  MapState* h_st = nullptr;  // [2]
  float* h_f = nullptr;
  // glue state (types/imu.hpp:171-187)
  float Bg[3]{};
  hm::M3 W_Bg{}, RGBias{}, RGyro{};
  LmState* lm_zero = nullptr;  // constant start state of minimizeVel (Vg = 0, rebvio.cpp:167)
  int lm_threads = 512;        // workgroup size of the persistent LM kernels (REBVIO_HIP_LM_THREADS: 256 | 512 | 1024)
  int dm_head_form = 0;        // directedMatch launch form: 0 by map size, 1 compact8, 2 compact4, 3 compact1 (REBVIO_HIP_DM_HEAD; track.hip dm_form)
  // Result slots. slot[0] also serves the per-pair API. The streaming driver (rebvio_hip_push_frame_u8_device) keeps the
  // WHOLE pair step on the device - the glue between the halves runs in front of the directedMatch head (glue.hpp) - and the
  // host reads a pair's records kSlots - 1 pairs late at most.
  static constexpr int kSlots = kPairSlots;
  PairSlot* slot[kSlots]{};    // pinned: LM state + map state records, written by the pair's first kernel
  hipEvent_t slot_ev[kSlots]{};  // recorded behind the pair's last kernel
  GlueRec* rec[kSlots]{};      // pinned: what the device glue of the pair reports
  GlueDev* glue_dev = nullptr;    // [kSlots] second-half inputs left by the device glue for the kernels behind it
  GlueStage* glue_stage = nullptr;  // [kSlots] host records on their way out: written by the glue, forwarded by the directedMatch launch
  GlueState* gstate = nullptr;    // [2] device: gyro-bias filter state + prior rotation, by pair parity
  GlueState* h_gstate = nullptr;  // [2] pinned staging for the upload in front of a stream's first pair
  hm::M3 gs_R{};                  // host mirror of the device state's prior rotation (with Bg / W_Bg above)
  int min_pool = 0;               // edge maps to allocate before one is reused (see acquire_map)
  int lead = 5;                   // detected frames queued when a pair is started (REBVIO_HIP_LEAD 3..12, see push_frame)
  int group = 4;                  // pairs queued together (REBVIO_HIP_GROUP 1..6, see stream_enqueue_group)
  // Sequence stamps of the host-visible records (PairSlot::seq, GlueRec::seq_gs / seq_out): every pair that is queued takes
  // the next non-zero value, its kernels store it behind everything else they write into the records, and whoever reads a
  // record compares. A record read before its pair ran (an event or a synchronisation that reported too early - seen once with
  // kernel-bound stop events) still holds its previous user's stamp: status -12 instead of silently stale poses.
  uint32_t stamp_seq = 0;         // last value handed out
  uint32_t last_stamp = 0;        // stamp of the pair enqueue_pair_lm queued last
  bool forge_stamp = false;       // rebvio_hip_test_forge_record_stamp: the next pair's kernels are handed a wrong stamp
  struct InFlight {               // a pair whose kernels are queued and whose record has not been read yet
    rebvio_hip_map* nm = nullptr;
    uint32_t seq = 0;             // its sequence stamp
    int slot = -1;
    int ev_slot = -1;             // the slot whose event stands for this pair's completion (its group's last pair)
    float frame_dt = 0.f;
  };
  struct Done {                   // a complete record waiting to be handed to the caller
    rebvio_hip_pair_out out;
    int keylines;
  };
  // Pool bookkeeping (in_use, has_done / done / done_ref, release_seq, release_counter, df_map, the pool vector) is written by
  // whoever releases a map - the tracking thread, or any thread that lets go of an EdgeMap (an edge-image consumer,
  // ros_rebvio.cpp:32-51) - and read by the acquisition thread: one mutex around both sides, so that a map seen as free is seen
  // with the event its next user has to wait for.
  std::mutex pool_mu;
  std::vector<rebvio_hip_map*> frames;  // detected maps not yet consumed as "old"
  std::deque<InFlight> inflight;
  std::deque<Done> done;
  uint64_t pair_seq = 0;
  // detect-enqueue worker (the reference's data-acquisition thread, rebvio.cpp:28): launches the detect kernels so
  // that the caller thread's launches (track chains) and the detect launches proceed in parallel on the host
  struct DetJob {
    rebvio_hip_map* m;
    const void* img;
    int is_u8;
    const DetState* det_in;
    DetState* det_out;
    const MapState* prev_st;
    int pin_slot = -1;      // host-frame entries: pinned ring slot to copy to `img` (device staging frame) ahead of the scans
    size_t pin_bytes = 0;
  };
  std::thread det_thread;
  std::mutex det_mu;
  std::condition_variable det_cv;
  std::deque<DetJob> det_jobs;
  std::atomic<int> det_pending{0};
  bool det_stop = false;
  std::string det_error;
  // host-side phase timing of the streaming driver (printed by flush when REBVIO_HIP_DEBUG is set)
  double t_detect_enq = 0, t_wait = 0, t_enq = 0, t_queued = 0;
  std::atomic<uint64_t> t_worker_ns{0}, t_worker_n{0};  // the detect worker's launches (one job = one frame)
  bool dbg = false;
  // host shadow of the information matrix W_Bg the DEVICE glue works on (streaming driver, glue_params_pre)
  hm::M3 wbg_shadow{};
  bool wbg_shadow_valid = false;
  bool fuse_dog = true;  // REBVIO_HIP_FUSE_DOG=0: k_dog_mag + k_keyline_flag instead of k_keyline_flag_ii
  bool gyro_pre_on = true;  // REBVIO_HIP_GYRO_PRE=0: the device forms the gyroBiasCorrection matrices itself
  // REBVIO_HIP_DETECT_WORKER=1: a worker thread launches the streaming driver's detect kernels. Off by default since round 3: the
  // runtime calls of two threads largely serialise AND slow each other down (detect launches 36 us per frame alone, 65 us beside
  // the caller's pair launches), so a burst of pushes - the driver's 20-frame window - ran at 10.4 k frames/s with the worker and
  // runs at 13.0 k with the caller launching everything itself (47 + 21 us of host time per 74 us frame).
  bool det_worker = false;
  uint64_t t_frames = 0;
  bool owns_streams = true;  // false for the lanes of a batch (rebvio_hip_batch_*)
  rebvio_hip_map* bf_map[2] = {nullptr, nullptr};  // new map of the pair whose counters result slot r will report
  bool bf_have[2] = {false, false};           // h_bf[r] already holds them (taken from the next pair's slot, or copied)
  bool bf_copy_queued[2] = {false, false};    // a device-to-host copy + bf_done[r] are queued for them
  // two result slots: the next pair's first half (and its parked second half) may be queued before the caller fetches the
  // previous pair's counters (rebvio_hip_track_pair_finish_async / _result)
  hipEvent_t bf_done[2]{};
  MapState* h_bf = nullptr;  // [2] pinned
  int bf_cur = 0;            // slot of the pair between _begin and _finish_async
  int bf_res = -1;           // slot whose result is to be fetched (-1: none)
  bool bf_nan[2] = {false, false};
};

namespace {
void release_map(rebvio_hip_map* m, hipEvent_t done_ref);

// A map handle whose context has been destroyed: every entry point that takes a map alone answers with this.
inline bool map_dead(const rebvio_hip_map* m) { return !m || !m->life || m->life->dead.load(std::memory_order_acquire); }
// Holds the life block (shared) for the rest of the calling function: a rebvio_hip_destroy on another thread waits until the
// entry has returned instead of freeing the context under it (the check alone would leave a window between test and use).
#define MAP_ALIVE_OR(m, ret)                                                   \
  const std::shared_ptr<LifeBlock> life_hold_ = (m) ? (m)->life : nullptr;     \
  std::shared_lock<std::shared_mutex> life_lock_;                              \
  if (life_hold_) life_lock_ = std::shared_lock<std::shared_mutex>(life_hold_->mu); \
  do {                                                                         \
    if (map_dead(m)) {                                                         \
      g_err = "the map's context has been destroyed (rebvio_hip_destroy)";     \
      return ret;                                                              \
    }                                                                          \
  } while (0)

// events of a map may only be waited on once the detect worker has recorded them
inline void wait_enqueued(rebvio_hip_map* m) {
  while (!m->enqueued.load(std::memory_order_acquire)) std::this_thread::yield();
}

size_t part_call_stride(const rebvio_hip_ctx* c) { return (size_t)c->maxblocks * kPartStride; }

// Wait for the track stream by polling: hipStreamQuery holds no runtime lock while it waits, so launches of other threads
// of the process (rebvio::Rebvio's acquisition thread) are not held up behind a blocked hipStreamSynchronize.
int trk_sync(rebvio_hip_ctx* c) {
  for (;;) {
    const hipError_t e = hipStreamQuery(c->s_trk);
    if (e == hipSuccess) return 0;
    if (e != hipErrorNotReady) HIPCHK(e);
    std::this_thread::yield();
  }
}

int alloc_map(rebvio_hip_ctx* c, rebvio_hip_map* m) {
  // keyline arrays are padded to the launch grids (256- and 1024-thread workgroups) so kernels may load before checking n
  const size_t M = (size_t)div_up(c->P.keylines_max, 1024) * 1024, Pn = (size_t)c->P.rows * c->P.cols;
  MapDev& d = m->d;
  HIPCHK(hipMalloc(&d.pos, M * sizeof(float2)));
  HIPCHK(hipMalloc(&d.pos_img, M * sizeof(float2)));
  HIPCHK(hipMalloc(&d.mpos_img, M * sizeof(float2)));
  HIPCHK(hipMalloc(&d.grad, M * sizeof(float2)));
  HIPCHK(hipMalloc(&d.mgrad, M * sizeof(float2)));
  HIPCHK(hipMalloc(&d.gnorm, M * sizeof(float)));
  HIPCHK(hipMalloc(&d.mgnorm, M * sizeof(float)));
  HIPCHK(hipMalloc(&d.rs, M * sizeof(float2)));
  HIPCHK(hipMalloc(&d.rs_tmp, M * sizeof(float2)));
  HIPCHK(hipMalloc(&d.grad_tmp, M * sizeof(float2)));
  HIPCHK(hipMalloc(&d.id_prev, M * sizeof(int)));
  HIPCHK(hipMalloc(&d.id_next, M * sizeof(int)));
  HIPCHK(hipMalloc(&d.match_id, M * sizeof(int)));
  HIPCHK(hipMalloc(&d.match_fwd, M * sizeof(int)));
  HIPCHK(hipMalloc(&d.match_kf, M * sizeof(int)));
  HIPCHK(hipMalloc(&d.matches, M * sizeof(unsigned)));
  HIPCHK(hipMalloc(&d.fwd_key, M * sizeof(unsigned long long)));
  HIPCHK(hipMalloc(&d.residual, M * sizeof(float)));
  HIPCHK(hipMalloc(&d.mask, Pn * sizeof(int)));
  HIPCHK(hipMalloc(&d.df, Pn * sizeof(unsigned)));
  HIPCHK(hipMalloc(&d.unit, M * sizeof(float2)));
  {
    const DfGrid g = df_grid(c->P.rows, c->P.cols);
    const size_t nt = (size_t)g.ntx * g.nty;
    HIPCHK(hipMalloc(&d.tile_cnt, nt * sizeof(int)));
    HIPCHK(hipMemset(d.tile_cnt, 0, nt * sizeof(int)));
    HIPCHK(hipMalloc(&d.tile_list, nt * kDfTileCap * 2 * sizeof(float4)));
  }
  HIPCHK(hipMalloc(&d.row_start, ((size_t)c->P.rows + 1) * sizeof(int)));
  HIPCHK(hipMemset(d.row_start, 0, ((size_t)c->P.rows + 1) * sizeof(int)));
  HIPCHK(hipMalloc(&d.st, sizeof(MapState)));
  HIPCHK(hipMemset(d.st, 0, sizeof(MapState)));
  HIPCHK(hipMemset(d.mask, 0xFF, Pn * sizeof(int)));
  HIPCHK(hipMemset(d.df, 0xFF, Pn * sizeof(unsigned)));
  // hipMemset on device memory returns BEFORE the fill has run (tools/memset_probe.hip on this runtime: the call takes 2.5 us, the
  // device finishes 10 us later for 1.2 MB), on the null stream, which the context's non-blocking streams do not wait for. A map
  // the pool grows by while the pipeline is running is handed to the detect kernels microseconds later: without this wait a late
  // fill wiped what they had written - the state record (n = 0), the dense mask, the distance field (every tryVel evaluation a
  // penalty: zero velocity, NaN covariance) - on a few per cent of fresh streams' first frames. (Round 3 saw records of this kind
  // on its 192x144 stream with kernel-bound stop events; the race explains them, that configuration was not re-run.)
  HIPCHK(hipStreamSynchronize(nullptr));
  HIPCHK(hipEventCreateWithFlags(&m->ready, hipEventDisableTiming));
  HIPCHK(hipEventCreateWithFlags(&m->detected, hipEventDisableTiming));
  HIPCHK(hipEventCreateWithFlags(&m->done, hipEventDisableTiming));
  return 0;
}

// device arrays and events of a map; the host struct stays (a handle the caller still holds is deleted by its release)
void free_map_device(rebvio_hip_map* m) {
  MapDev& d = m->d;
  void* ptrs[] = {d.pos, d.pos_img, d.mpos_img, d.grad, d.grad_tmp, d.mgrad, d.gnorm, d.mgnorm, d.rs, d.rs_tmp, d.id_prev, d.id_next,
                  d.match_id, d.match_fwd, d.match_kf, d.matches, d.fwd_key, d.residual, d.mask, d.df, d.unit, d.tile_cnt, d.tile_list, d.row_start, d.st};
  for (void* p : ptrs)
    if (p) (void)hipFree(p);
  d = MapDev{};
  m->canon = MapDev{};
  if (m->ready) (void)hipEventDestroy(m->ready);
  if (m->detected) (void)hipEventDestroy(m->detected);
  if (m->done) (void)hipEventDestroy(m->done);
  m->ready = m->detected = m->done = hipEvent_t{};
  m->ctx = nullptr;
}

int alloc_map(rebvio_hip_ctx* c, rebvio_hip_map* m);
rebvio_hip_map* acquire_map(rebvio_hip_ctx* c) {
  std::lock_guard<std::mutex> pool_lk(c->pool_mu);
  // Oldest release first: a map is released in stream order, i.e. before its last consumer (the pair's B-chain) has
  // run; reusing the most recently released one makes the new frame's keyline kernels wait for that consumer.
  rebvio_hip_map* best = nullptr;
  for (auto* m : c->pool)
    if (!m->in_use && (!best || m->release_seq < best->release_seq)) best = m;
  // The streaming drivers release a map when its group of pairs is QUEUED, up to a ring of pairs before the tracker has run it
  // on the device; the frame that reuses it waits (in stream order) for that group. With too few maps the detect stage can
  // only work a few pairs ahead of the tracker and the tracker ends up waiting for its maps: min_pool keeps the reuse distance
  // at two groups and more.
  if (best && (int)c->pool.size() < c->min_pool) best = nullptr;
  if (best) {
    rebvio_hip_map* m = best;
    m->in_use = true;
    m->df_built = false;
    m->raster_order = false;
    m->pre_rotated = false;
    m->n_host = -1;
    m->thr_host = -1.0f;
    m->trk_waited = false;
    m->trk_touched.store(false, std::memory_order_relaxed);  // (its previous life drained before its detection: m->done)
    return m;
  }
  // every pooled map is alive (a caller queues detections faster than it tracks, like the reference's unbounded
  // edge_map_buffer_): grow the pool, ~4 MB per map at 640x480, bounded
  if (c->pool.size() >= 256) return nullptr;
  rebvio_hip_map* m = new rebvio_hip_map;
  m->ctx = c;
  m->life = c->life;
  if (alloc_map(c, m) != 0) {
    free_map_device(m);
    delete m;
    return nullptr;
  }
  c->pool.push_back(m);
  m->in_use = true;
  return m;
}

int fetch_map_state(rebvio_hip_map* m, MapState* out, hipStream_t after) {
  // copy stream: wait for the producer, copy, block the host only on this small transfer
  rebvio_hip_ctx* c = m->ctx;
  wait_enqueued(m);
  HIPCHK(hipStreamWaitEvent(c->s_key, m->ready, 0));
  if (after) {
    hipEvent_t e;
    HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    HIPCHK(hipEventRecord(e, after));
    HIPCHK(hipStreamWaitEvent(c->s_key, e, 0));
    (void)hipEventDestroy(e);
  }
  HIPCHK(hipMemcpyAsync(c->h_st, m->d.st, sizeof(MapState), hipMemcpyDeviceToHost, c->s_key));
  HIPCHK(hipStreamSynchronize(c->s_key));
  *out = *c->h_st;
  m->n_host = out->n;
  m->thr_host = out->threshold;
  return 0;
}

// Every use of a map on the track stream starts with this wait on its detection (+ distance field).
hipError_t trk_wait_ready(hipStream_t s, rebvio_hip_map* m) {
  if (!m->trk_touched.load(std::memory_order_relaxed)) {
    // first use by the tracker: not while a download of an "untouched" map is packing it on the copy stream
    // (rebvio_hip_map_download decides and packs under the same mutex)
    std::lock_guard<std::mutex> dl(m->ctx->dl_mu);
    m->trk_touched.store(true, std::memory_order_release);
  }
  return hipStreamWaitEvent(s, m->ready, 0);
}
// The same, once per map and context stream: every wait is a barrier packet of its own between two kernels of the track
// stream (~2 us each on the pair-to-pair critical path); `ready` is recorded once per detection, a second wait adds nothing.
hipError_t trk_wait_ready_once(rebvio_hip_ctx* c, rebvio_hip_map* m) {
  if (m->trk_waited) return hipSuccess;
  m->trk_waited = true;
  return trk_wait_ready(c->s_trk, m);
}

int ensure_size(rebvio_hip_map* m) {
  if (m->n_host >= 0) return 0;
  MapState st;
  return fetch_map_state(m, &st, nullptr);
}

int detect_launch(rebvio_hip_ctx* c, const rebvio_hip_ctx::DetJob& j) {
  rebvio_hip_map* m = j.m;
  const int b = (int)(c->launch_index & 1);
  c->launch_index++;
  ScaleBufs sb = c->sb;
  sb.scale0 = sb.scale1 = nullptr;
  sb.dog = c->dog2[b];
  sb.mag = c->mag2[b];
  sb.a[0] = c->sa2[b][0];  // the third filter's integral images change streams (below): one pair per frame parity
  sb.a[1] = c->sa2[b][1];
  DetectBufs db = c->db;
  db.rowcount = c->rowcount2[b];
  if (j.pin_slot >= 0) {  // host frame: pinned slot -> device staging frame, read by this stream's own kernels only
    launch_copy_from_pinned(c->s_det, c->pin[j.pin_slot], const_cast<void*>(j.img), j.pin_bytes);
    HIPCHK(hipEventRecord(c->pin_ev[j.pin_slot], c->s_det));
    c->pin_staged[j.pin_slot].store(0, std::memory_order_release);
  }
  // scans of this frame (s_det); its DoG / gradient buffers were last read by the candidate kernel two frames ago
  if (c->prev_ready[b]) HIPCHK(hipStreamWaitEvent(c->s_det, c->prev_ready[b], 0));
  const void* img = j.img;
  int is_u8 = j.is_u8;
  if (is_u8 && c->undist_map) {  // x3 + undistort (rebvio.cpp:43-47); its output was last read by the scans two frames ago
    launch_front_end_u8(c->s_det, c->K, (const uint8_t*)img, c->undist_map, c->undist_img[b]);
    img = c->undist_img[b];
    is_u8 = 0;
  }
  // The scan stream is the busiest of the three: it hands the frame over after the last ROW pass, the last column pass and
  // the DoG / gradient kernel run at the head of the keyline stream.
  // Their inputs sb.a[] are then read while the scan stream already works on the next frame, hence the pair per parity;
  // the frame after next waits for this frame's `ready` event (prev_ready[b]) above.
  launch_scale_space(c->s_det, c->K, img, is_u8, sb, c->widths, db.rowcount, 1);
  HIPCHK(hipEventRecord(c->ev_scan[b], c->s_det));
  // keyline extraction + chaining (s_key), overlapping the next frame's scans
  HIPCHK(hipStreamWaitEvent(c->s_key, c->ev_scan[b], 0));
  launch_scale_space(c->s_key, c->K, img, is_u8, sb, c->widths, db.rowcount, 2, c->fuse_dog);
  if (m->has_done) HIPCHK(hipStreamWaitEvent(c->s_key, m->done_ref ? m->done_ref : m->done, 0));
  const int fw[2] = {c->widths[0][2], c->widths[1][2]};
  launch_keylines(c->s_key, c->K, sb, db, m->d, j.det_in, j.det_out, j.prev_st, c->fuse_dog ? fw : nullptr);
  HIPCHK(hipGetLastError());
  // distance field of this map, behind its keylines on the same stream (stream order is the dependency)
  launch_df_build(c->s_key, c->K, m->d, j.det_out, true);
  HIPCHK(hipGetLastError());
  HIPCHK(hipEventRecord(m->ready, c->s_key));
  c->prev_ready[b] = m->ready;  // (pooled maps and their events live as long as the context)
  m->enqueued.store(1, std::memory_order_release);
  return 0;
}

// caller-thread half: takes a pooled map and fixes the servo-state ring slots of this frame
int detect_prepare(rebvio_hip_ctx* c, const void* img_dev, int is_u8, uint64_t ts, rebvio_hip_ctx::DetJob* job) {
  rebvio_hip_map* m = acquire_map(c);
  if (!m) return fail_msg("edge-map pool exhausted (release maps or raise map_pool)", -2);
  m->ts = ts;
  m->df_built = true;
  m->raster_order = true;
  job->m = m;
  job->img = img_dev;
  job->is_u8 = is_u8;
  job->det_in = c->det + (c->frame_index % kDetRing);
  job->det_out = c->det + ((c->frame_index + 1) % kDetRing);
  job->prev_st = c->last_detected ? c->last_detected->d.st : nullptr;
  c->frame_index++;
  c->last_detected = m;
  return 0;
}

// A detect_launch that failed half way: the frame's map has been taken, the servo ring and last_detected have advanced and some of
// its kernels may be queued. The map goes back to the pool, a staged host frame is dropped, and the context is marked failed
// (every later push / detect reports -8 with this error) - the detector's state is one frame out of step and cannot be trusted.
void detect_failed(rebvio_hip_ctx* c, const rebvio_hip_ctx::DetJob& job) {
  {
    std::lock_guard<std::mutex> lk(c->det_mu);
    if (c->det_error.empty()) c->det_error = g_err.empty() ? std::string("detect launch failed") : g_err;
  }
  if (job.pin_slot >= 0) c->pin_staged[job.pin_slot].store(0, std::memory_order_release);
  job.m->enqueued.store(1, std::memory_order_release);
  release_map(job.m, nullptr);
}

int detect_common(rebvio_hip_ctx* c, const void* img_dev, int is_u8, uint64_t ts, rebvio_hip_map** out) {
  {
    std::lock_guard<std::mutex> lk(c->det_mu);
    if (!c->det_error.empty()) return fail_msg(c->det_error.c_str(), -8);
  }
  rebvio_hip_ctx::DetJob job;
  int rc = detect_prepare(c, img_dev, is_u8, ts, &job);
  if (rc) return rc;
  // a queued asynchronous detect must be launched first (stream order = frame order)
  while (c->det_pending.load(std::memory_order_acquire) > 0) std::this_thread::yield();
  rc = detect_launch(c, job);
  if (rc) {
    detect_failed(c, job);
    return rc;
  }
  *out = job.m;
  return 0;
}

void det_worker_main(rebvio_hip_ctx* c) {
  (void)hipSetDevice(c->device);
  for (;;) {
    rebvio_hip_ctx::DetJob j;
    {
      std::unique_lock<std::mutex> lk(c->det_mu);
      c->det_cv.wait(lk, [&] { return c->det_stop || !c->det_jobs.empty(); });
      if (c->det_jobs.empty()) return;
      j = c->det_jobs.front();
      c->det_jobs.pop_front();
    }
    const auto tw0 = std::chrono::steady_clock::now();
    if (detect_launch(c, j) != 0) {
      std::lock_guard<std::mutex> lk(c->det_mu);
      c->det_error = g_err;
      j.m->enqueued.store(1, std::memory_order_release);
    }
    if (c->dbg) {
      c->t_worker_ns.fetch_add((uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - tw0).count(),
                               std::memory_order_relaxed);
      c->t_worker_n.fetch_add(1, std::memory_order_relaxed);
    }
    c->det_pending.fetch_sub(1, std::memory_order_release);
  }
}

int stage_host_frame(rebvio_hip_ctx* c, const void* img, size_t pitch_bytes, size_t row_bytes, int* slot_out, size_t* bytes_out);
// host_u8 != null: a MONO8 frame in host memory, staged through the pinned ring; the worker queues its copy to the device
// staging frame ahead of the scans (same stream: stream order is reuse order)
int detect_async(rebvio_hip_ctx* c, const void* img_dev, int is_u8, uint64_t ts, rebvio_hip_map** out, const uint8_t* host_u8 = nullptr,
                 size_t host_pitch = 0) {
  rebvio_hip_ctx::DetJob job;
  if (host_u8) {
    const int rcs = stage_host_frame(c, host_u8, host_pitch ? host_pitch : (size_t)c->P.cols, (size_t)c->P.cols, &job.pin_slot, &job.pin_bytes);
    if (rcs) return rcs;
    img_dev = c->img8_dev;
  }
  int rc = detect_prepare(c, img_dev, is_u8, ts, &job);
  if (rc && job.pin_slot >= 0) c->pin_staged[job.pin_slot].store(0, std::memory_order_release);
  if (rc) return rc;
  if (!c->det_worker) {  // REBVIO_HIP_DETECT_WORKER=0: the caller launches the detect kernels itself (one thread issues every runtime call)
    const auto tw0 = std::chrono::steady_clock::now();
    rc = detect_launch(c, job);
    if (c->dbg) {
      c->t_worker_ns.fetch_add((uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - tw0).count(),
                               std::memory_order_relaxed);
      c->t_worker_n.fetch_add(1, std::memory_order_relaxed);
    }
    if (rc) {
      detect_failed(c, job);
      return rc;
    }
    job.m->enqueued.store(1, std::memory_order_release);
    *out = job.m;
    return 0;
  }
  if (!c->det_thread.joinable()) c->det_thread = std::thread(det_worker_main, c);
  job.m->enqueued.store(0, std::memory_order_relaxed);
  c->det_pending.fetch_add(1, std::memory_order_release);
  {
    std::lock_guard<std::mutex> lk(c->det_mu);
    c->det_jobs.push_back(job);
  }
  c->det_cv.notify_one();
  *out = job.m;
  return 0;
}

void rotate_inputs(const rebvio_hip_ctx* c, const float vel[3], const float Rvel[9], const float Rback[9], float vel_r[3],
                   float Rvel_r[9]) {
  // EdgeMap::directedMatch prologue (edge_map.cpp:193-194)
  (void)c;
  const hm::M3 Rb = hm::load3(Rback);
  hm::mulv(Rb, vel, vel_r);
  hm::store3(hm::mul(hm::mul(Rb, hm::load3(Rvel)), hm::transpose(Rb)), Rvel_r);
}

void sum_xrv(const float* h_xrv, int nblocks, float Wx[36], float JtF[6], int* nm) {
  double acc[28];
  for (int k = 0; k < 28; ++k) acc[k] = 0.0;
  for (int b = 0; b < nblocks; ++b)
    for (int k = 0; k < 28; ++k) acc[k] += (double)h_xrv[(size_t)b * kXrvStride + k];
  int k = 0;
  for (int i = 0; i < 6; ++i)
    for (int j = i; j < 6; ++j) {
      Wx[i * 6 + j] = (float)acc[k];
      Wx[j * 6 + i] = (float)acc[k];
      ++k;
    }
  for (int i = 0; i < 6; ++i) JtF[i] = (float)acc[21 + i];
  if (nm) *nm = (int)std::lround(acc[27]);
}

// The speculative LM kernel bets that minimizeVel rejects every step after the first accepted one (track.hip). Where the
// bet fails often - large inter-frame motion: 35 % of the pairs when a 20 Hz stream is replayed in jumps of 1..6 frames - it
// costs more than it saves (measured there: 3.7 % slower than the sequential kernel; 4-5 % faster on consecutive frames;
// break-even near one miss in six). Both kernels give the same bits, so the choice follows the recent miss rate of the
// stream (exponential average over ~16 pairs, known with a lag of two pairs in the streaming driver).
// 1 = k_lm_chain, 2 / 3 = the speculative kernel with its first speculative evaluation at that index: the earliest hypothesis
// the stream's recent accept masks have been supporting (a failed hypothesis costs a roll-back: break-even near one miss in six)
int lm_kernel_choice(const rebvio_hip_ctx* c) {
  if (c->lm_mix) {  // REBVIO_HIP_LM=mix<seed>: a pseudo-random kernel per pair (tests: every sequence of choices gives the same records)
    unsigned x = (unsigned)c->pair_seq * 2654435761u + (unsigned)c->lm_mix * 40503u;
    x ^= x >> 15;
    x *= 2246822519u;
    x ^= x >> 13;
    return 1 + (int)(x % 3u);
  }
  if (!c->lm_spec) return 1;
  if (c->lm_spec_forced) return c->lm_spec_forced_kf;
  if (c->lm_miss_ema < 0.17f) return 2;
  if (c->lm_miss3_ema < 0.17f) return 3;
  return 1;
}
void note_accept_mask(rebvio_hip_ctx* c, int mask) {
  const float miss = (mask >> 1) != 0 ? 1.0f : 0.0f;  // an accept after the second evaluation = a failed hypothesis
  c->lm_miss_ema += (miss - c->lm_miss_ema) * (1.0f / 16.0f);
  const float miss3 = (mask >> 2) != 0 ? 1.0f : 0.0f;
  c->lm_miss3_ema += (miss3 - c->lm_miss3_ema) * (1.0f / 16.0f);
}

void lm_to_out(const LmState& s, float vel[3], float Rvel[9], float* F, int* mask, float* srm) {
  for (int i = 0; i < 3; ++i) vel[i] = s.vel[i];
  hm::M3 J;
  J.a[0][0] = s.JtJ[0]; J.a[1][1] = s.JtJ[1]; J.a[2][2] = s.JtJ[2];
  J.a[0][1] = J.a[1][0] = s.JtJ[3];
  J.a[0][2] = J.a[2][0] = s.JtJ[4];
  J.a[1][2] = J.a[2][1] = s.JtJ[5];
  hm::store3(hm::invert3(J), Rvel);  // _Rvel = invert(JtJ) (core.cpp:186)
  if (F) *F = s.F;
  if (mask) *mask = s.accept_mask;
  if (srm) *srm = s.sigma_rho_min;
}

// minimizeVel on the track stream: histogram must already be in c->hist and residuals zeroed.
void enqueue_lm_chain(rebvio_hip_ctx* c, rebvio_hip_map* om, rebvio_hip_map* nm, const float vel0[3]) {
  LmState* first = c->lm_zero;
  if (vel0[0] != 0.f || vel0[1] != 0.f || vel0[2] != 0.f) {
    LmState init;
    std::memset(&init, 0, sizeof(init));
    for (int i = 0; i < 3; ++i) init.vel[i] = vel0[i];
    c->h_lm[1] = init;
    (void)hipMemcpyAsync(c->lm, &c->h_lm[1], sizeof(LmState), hipMemcpyHostToDevice, c->s_trk);
    first = c->lm;
  }
  const int calls = (int)c->P.iterations + 1;
  const size_t cs = part_call_stride(c);
  prof_group_begin(c->s_trk, "k_try_vel", calls);
  for (int i = 0; i < calls; ++i)
    launch_try_vel(c->s_trk, c->K, om->d, nm->d, 1, i, i == calls - 1, i ? c->lm + i : first, c->lm + i + 1,
                   i ? c->part + (size_t)(i - 1) * cs : c->part, c->part + (size_t)i * cs, c->hist, 0);
  prof_group_end(c->s_trk);
}

// minimizeVel + forwardMatch + extRotVel of one pair on the track stream (rebvio.cpp:167-177); results land in `slot`.
// Default: the persistent kernel (one launch); REBVIO_HIP_LM=percall selects the per-evaluation kernels, which compute
// identical bits (same per-keyline code, same record order).
// xrv_dst: where the extRotVel block records go - the pinned slot's tail (per-pair API: the host sums them) or device
// memory (streaming driver: the device glue sums them).
// ga.lm != null: the pair's glue runs on the device behind the extRotVel sums (streaming driver).
uint32_t next_stamp(uint32_t* counter) {
  if (++*counter == 0u) ++*counter;  // (zero is what a never-written record holds)
  return *counter;
}
constexpr int kStaleRecord = -12;
const char* const kStaleRecordMsg = "a pair's record was read before the device had written it (sequence stamp mismatch)";

// Waits for a pair's host records (per-pair API: the result slot and the extRotVel block records behind it) by polling their
// sequence stamps. The stream is queried now and then so that a device-side failure ends the wait with its error; after two
// seconds without the stamps the stream is synchronised and the stamps decide (-12).
int poll_pair_records(rebvio_hip_ctx* c, const PairSlot* slot, uint32_t seq) {
  const auto t0 = std::chrono::steady_clock::now();
  unsigned spins = 0;
  bool synced = false;
  // one step of waiting: 0 keep polling, < 0 error. The stream is queried now and then so that a device-side failure ends the wait
  // with its error; after two seconds the stream is synchronised and the records get one last look.
  auto wait_step = [&]() -> int {
    if (synced) return fail_msg(kStaleRecordMsg, kStaleRecord);
    __builtin_ia32_pause();
    if ((++spins & 0x3FFFu) != 0u) return 0;
    const hipError_t q = hipStreamQuery(c->s_trk);
    if (q != hipSuccess && q != hipErrorNotReady) return fail("track stream", q);
    if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(2)) {
      const hipError_t e = hipStreamSynchronize(c->s_trk);
      if (e != hipSuccess) return fail("track stream", e);
      synced = true;
    }
    return 0;
  };
  // A record counts once its stamp AND its checksum fit: the words are re-read on every try (writes to host memory may arrive out
  // of order; a torn read fails the sum and is simply repeated).
  auto slot_ok = [&]() -> bool {
    const volatile unsigned* w = reinterpret_cast<const volatile unsigned*>(slot);
    if (*reinterpret_cast<const volatile unsigned*>(&slot->seq) != seq) return false;
    unsigned x = 0u;
    for (size_t i = 0; i < offsetof(PairSlot, seq) / sizeof(unsigned); ++i) x ^= w[i];
    return (x ^ seq) == *reinterpret_cast<const volatile unsigned*>(&slot->sum);
  };
  static_assert(offsetof(PairSlot, seq) == sizeof(LmState) + 2 * sizeof(MapState), "PairSlot::sum covers lm, new_st, old_st");
  while (!slot_ok()) {
    const int e = wait_step();
    if (e) return e;
  }
  std::atomic_thread_fence(std::memory_order_acquire);
  const int nb = std::min(div_up(std::max(slot->new_st.n, 0), 256), c->maxblocks);
  for (int b = 0; b < nb; ++b) {
    const volatile unsigned* w = reinterpret_cast<const volatile unsigned*>(slot->xrv + (size_t)b * kXrvStride);
    auto rec_ok = [&]() -> bool {
      if (w[kXrvStride - 1] != seq) return false;
      unsigned x = 0u;
      for (int k = 0; k < 28; ++k) x ^= w[k];
      return (x ^ seq) == w[kXrvStride - 2];
    };
    while (!rec_ok()) {
      const int e = wait_step();
      if (e) return e;
    }
  }
  std::atomic_thread_fence(std::memory_order_acquire);
  return 0;
}

int enqueue_pair_lm(rebvio_hip_ctx* c, rebvio_hip_map* om, rebvio_hip_map* nm, const float vel0[3], PairSlot* slot, float* xrv_dst,
                    const GlueArgs& ga_in) {
  const int calls = (int)c->P.iterations + 1;
  GlueArgs ga = ga_in;
  c->last_stamp = next_stamp(&c->stamp_seq);
  ga.seq = c->last_stamp;
  if (c->forge_stamp) {  // test hook: what a record looks like when it is read before its pair wrote it
    ga.seq ^= 0x40000000u;
    c->forge_stamp = false;
  }
  if (!c->lm_persistent) {
    enqueue_lm_chain(c, om, nm, vel0);
    launch_ext_rot_vel(c->s_trk, c->K, om->d, nm->d, 1, 1, calls, c->lm + calls, c->lm + calls + 1,
                       c->part + (size_t)(calls - 1) * part_call_stride(c), xrv_dst, vel0, slot, c->hist, ga.seq);
    if (ga.lm) launch_pair_glue(c->s_trk, nm->d, ga);
    return 0;
  }
  if (*c->lm_bar_err) {
    const int* e = c->lm_bar_err;
    char msg[256];
    std::snprintf(msg, sizeof(msg),
                  "persistent LM kernel: record exchange timed out (workgroup %d/%d thread %d waited for tag %u, last saw tag %u; "
                  "tags issued so far %u); ",
                  e[1], e[6], e[2], (unsigned)e[3], (unsigned)e[4], c->lm_tag_base);
    return fail_msg((std::string(msg) + residency_describe(c->device, c)).c_str(), -9);
  }
  if (!c->residency_registered) {  // (the lane contexts of a batch never come here: the batch registers for them)
    residency_add(c->device, c, (c->K.kmax + c->lm_threads - 1) / c->lm_threads);
    c->residency_registered = true;
  }
  LmState* first = c->lm_zero;
  if (vel0[0] != 0.f || vel0[1] != 0.f || vel0[2] != 0.f) {
    LmState init;
    std::memset(&init, 0, sizeof(init));
    for (int i = 0; i < 3; ++i) init.vel[i] = vel0[i];
    c->h_lm[1] = init;
    (void)hipMemcpyAsync(c->lm, &c->h_lm[1], sizeof(LmState), hipMemcpyHostToDevice, c->s_trk);
    first = c->lm;
  }
  if (c->lm_stamps && c->lm_stamps[0]) {  // stamps of the previous launch (the caller has synchronised on its slot since)
    // the buffer may be half rewritten by a launch that is already running (streaming driver): take a snapshot and use it
    // only if it is monotonic and spans less than a millisecond
    const bool spec = c->lm_stamps[0] == 1ull;  // k_lm_chain_spec: stamps 1..14, [0] is a marker
    const int i0 = spec ? 1 : 0;
    const int ns = spec ? 16 : 3 + calls * 6;
    unsigned long long snap[64];
    for (int i = 0; i < ns; ++i) snap[i] = c->lm_stamps[i];
    bool sane = snap[ns - 1] > snap[i0] && snap[ns - 1] - snap[i0] < 100000ull;
    for (int i = i0 + 1; i < ns && sane; ++i) sane = snap[i] >= snap[i - 1];
    if (sane) {
      for (int i = i0 + 1; i < ns; ++i) c->lm_stamp_acc[i] += (double)(snap[i] - snap[i - 1]) * 0.01;
      c->lm_stamp_n++;
      c->lm_stamp_spec = spec;
    }
  }
  launch_lm_chain(c->s_trk, c->K, om->d, nm->d, calls, lm_kernel_choice(c), first, c->lm + calls + 1, c->lm_xch, c->lm_tag_base, c->lm_bar_err, c->hist,
                  xrv_dst, slot, c->hist, c->lm_stamps, c->lm_threads, ga);
  c->lm_tag_base += 2u * ((unsigned)calls + 1u);  // (the speculative kernel numbers repeated evaluations in a second range)
  if (c->lm_tag_base > 0xFFFFFF00u) {  // tags must stay unique and non-zero: restart the sequence on clean exchange words
    (void)hipMemsetAsync(c->lm_xch, 0, lm_xch_words(c->maxblocks) * sizeof(unsigned long long), c->s_trk);
    c->lm_tag_base = 0;
  }
  return 0;
}

// Host frame -> pinned ring slot (plain memcpy) -> asynchronous copy on the scan stream. A copy straight from pageable
// memory is staged inside the runtime, which stays busy for the whole transfer and stalls the launches of the tracking
// thread of rebvio::Rebvio (measured: second half of the pair step 215 us -> 47 us); it also must not outlive the caller's
// buffer. The device staging frames are read by the scan stream's own kernels only, so stream order is reuse order.
// Host half (caller's thread): frame -> pinned ring slot. The copy to the device and the reuse event are queued by whoever
// launches the frame's detection (detect_launch), in stream order with its kernels.
int stage_host_frame(rebvio_hip_ctx* c, const void* img, size_t pitch_bytes, size_t row_bytes, int* slot_out, size_t* bytes_out) {
  if (!c->pin[0]) {
    for (int i = 0; i < rebvio_hip_ctx::kPin; ++i) {
      HIPCHK(hipHostMalloc(&c->pin[i], (size_t)c->P.rows * c->P.cols * sizeof(float) + 16, hipHostMallocDefault));  // (+16: copied in whole 16-byte units)
      HIPCHK(hipEventCreateWithFlags(&c->pin_ev[i], hipEventDisableTiming));
    }
  }
  const int ps = (int)(c->pin_next++ % rebvio_hip_ctx::kPin);
  while (c->pin_staged[ps].load(std::memory_order_acquire)) std::this_thread::yield();  // the worker is kPin frames behind
  if (c->pin_used[ps]) HIPCHK(hipEventSynchronize(c->pin_ev[ps]));  // its previous copy has left the slot
  uint8_t* dst = static_cast<uint8_t*>(c->pin[ps]);
  const uint8_t* src = static_cast<const uint8_t*>(img);
  if (pitch_bytes == row_bytes)
    std::memcpy(dst, src, (size_t)c->P.rows * row_bytes);
  else
    for (int r = 0; r < c->P.rows; ++r) std::memcpy(dst + (size_t)r * row_bytes, src + (size_t)r * pitch_bytes, row_bytes);
  c->pin_used[ps] = true;
  c->pin_staged[ps].store(1, std::memory_order_release);
  *slot_out = ps;
  *bytes_out = (size_t)c->P.rows * row_bytes;
  return 0;
}

// Detection of a host frame: staged through the pinned ring and launched by the caller. (Handing the launch to the context's
// detect worker was measured with rebvio::Rebvio in round 2: the acquisition thread's time per frame dropped, the fusion
// thread's own launches slowed down by as much - runtime calls of different threads largely serialise - no gain, removed.)
int detect_host_frame(rebvio_hip_ctx* c, const void* img, size_t pitch_bytes, size_t row_bytes, void* dst_dev, int is_u8, uint64_t ts,
                      rebvio_hip_map** out) {
  {
    std::lock_guard<std::mutex> lk(c->det_mu);
    if (!c->det_error.empty()) return fail_msg(c->det_error.c_str(), -8);
  }
  rebvio_hip_ctx::DetJob job;
  int rc = stage_host_frame(c, img, pitch_bytes, row_bytes, &job.pin_slot, &job.pin_bytes);
  if (rc) return rc;
  rc = detect_prepare(c, dst_dev, is_u8, ts, &job);
  if (rc) {
    c->pin_staged[job.pin_slot].store(0, std::memory_order_release);
    return rc;
  }
  while (c->det_pending.load(std::memory_order_acquire) > 0) std::this_thread::yield();  // (a streaming push's detection goes first)
  rc = detect_launch(c, job);
  if (rc) return rc;
  *out = job.m;
  return 0;
}

}  // namespace

extern "C" {

int rebvio_hip_abi_version(void) { return REBVIO_HIP_ABI_VERSION; }
const char* rebvio_hip_last_error(void) { return g_err.c_str(); }

void rebvio_hip_default_params(rebvio_hip_params* p, int rows, int cols) {
  std::memset(p, 0, sizeof(*p));
  p->rows = rows; p->cols = cols;
  const float fx = 458.654, fy = 457.296;  // camera.hpp:26-30
  p->fm = 0.5 * (fx + fy);
  p->cx = 367.215; p->cy = 248.375;
  p->keylines_ref = 12000; p->keylines_max = 16000;
  p->pos_neg_threshold = 0.4; p->dog_threshold = 0.095259868922420; p->threshold = 0.01; p->gain = 5e-7;
  p->max_threshold = 0.5; p->min_threshold = 0.005;
  p->search_range = 40.0; p->reweight_distance = 2.0; p->match_treshold = 0.5;
  p->min_match_threshold = 0; p->iterations = 5; p->global_min_matches_threshold = 500;
  p->pixel_uncertainty = 1; p->quantile_cutoff = 0.9; p->quantile_num_bins = 100; p->reshape_q_abs = 1e-4;
  p->pixel_uncertainty_match = 2.0; p->match_threshold_norm = 1.0; p->match_threshold_angle = 45.0;
  p->regularization_threshold = 0.5;
  p->gyro_std_dev = 1.6968e-04; p->gyro_bias_std_dev = 1.9393e-05;
  p->device_id = 0; p->map_pool = 0;
}

void rebvio_hip_reset_state(rebvio_hip_ctx* c) {
  // While the streaming driver has pairs in flight the filter state lives on the device: they are completed first (their records
  // stay available through rebvio_hip_next_record), so that no harvested record writes the old state back over the reset one.
  if (!c->inflight.empty() || !c->frames.empty()) (void)rebvio_hip_flush(c);
  c->wbg_shadow_valid = false;  // (the next stream's first pair uploads this state and starts the shadow from it)
  c->Bg[0] = c->Bg[1] = c->Bg[2] = 0.f;
  c->RGBias = hm::identity3();
  c->RGyro = hm::identity3();
  c->W_Bg = hm::invert3(hm::diag3(100.0f));  // W_Bg{invert(100.0*RGBias)} (types/imu.hpp:181)
}

int rebvio_hip_get_gyro_state(rebvio_hip_ctx* c, float Bg[3], float W_Bg[9]) {
  for (int i = 0; i < 3; ++i) Bg[i] = c->Bg[i];
  hm::store3(c->W_Bg, W_Bg);
  return 0;
}

int rebvio_hip_set_gyro_state(rebvio_hip_ctx* c, const float Bg[3], const float W_Bg[9]) {
  // While the streaming driver has pairs or frames in flight the filter state lives on the device (and the next pair's first
  // rotation has already been applied with it): it can only be replaced between streams.
  if (!c->inflight.empty() || !c->frames.empty())
    return fail_msg("set_gyro_state: the streaming driver has frames in flight (rebvio_hip_flush first)", -7);
  for (int i = 0; i < 3; ++i) c->Bg[i] = Bg[i];
  c->W_Bg = hm::load3(W_Bg);
  c->wbg_shadow_valid = false;  // (the next stream's first pair uploads this state and starts the shadow from it)
  return 0;
}

int rebvio_hip_create(const rebvio_hip_params* p, rebvio_hip_ctx** out) {
  *out = nullptr;
  if (p->rows < 32 || p->cols < 32) return fail_msg("rows/cols must be >= 32", -3);
  if (p->cols > 4096) return fail_msg("cols > 4096 unsupported", -3);
  if ((size_t)(p->rows + 12) * 16 * sizeof(float) > 160 * 1024 || (size_t)(p->cols + 3 + 8) * 4 * sizeof(float) > 160 * 1024)
    return fail_msg("image too large for the LDS-staged scan strips (rows <= 2548)", -3);
  if (p->quantile_num_bins > 128 || p->quantile_num_bins < 1) return fail_msg("quantile_num_bins must be in 1..128", -3);
  if ((int)p->iterations + 2 > 15) return fail_msg("iterations too large", -3);
  const int nr = 2 * (int)p->search_range;
  if (nr < 2 || (long long)p->keylines_max * nr >= (1ll << kDfSeqBits) || (int)p->search_range > 255)
    return fail_msg("keylines_max * 2*search_range must stay below 2^23", -3);
  if (p->keylines_max < 1 || div_up(p->keylines_max, 256) > kMaxRecBlocks)
    return fail_msg("keylines_max must be in 1..65536 (the LM reduction stages at most 256 record groups of 256 keylines)", -3);
  if (!(p->pixel_uncertainty_match >= 0.0f) || p->search_range + 2.0f * p->pixel_uncertainty_match + 2.0f > 260.0f)
    return fail_msg("search_range + 2 * pixel_uncertainty_match must stay below 258 (probe sequence buffer)", -3);
  int ndev = 0;
  HIPCHK(hipGetDeviceCount(&ndev));
  if (ndev <= 0) return fail_msg("no HIP device present: the gfx950 backend has no CPU fallback", -4);
  if (p->device_id < 0 || p->device_id >= ndev) return fail_msg("device_id out of range", -3);
  HIPCHK(hipSetDevice(p->device_id));

  rebvio_hip_ctx* c = new rebvio_hip_ctx;
  struct Guard {  // every early return below (HIPCHK, fail_msg) frees what was allocated so far
    rebvio_hip_ctx* c;
    ~Guard() {
      if (c) rebvio_hip_destroy(c);
    }
  } guard{c};
  c->P = *p;
  c->device = p->device_id;
  KParams& K = c->K;
  K.rows = p->rows; K.cols = p->cols; K.fm = p->fm; K.cx = p->cx; K.cy = p->cy;
  K.kmax = p->keylines_max; K.kref = p->keylines_ref;
  K.pos_neg_threshold = p->pos_neg_threshold; K.dog_threshold = p->dog_threshold; K.gain = p->gain;
  K.max_threshold = p->max_threshold; K.min_threshold = p->min_threshold;
  K.search_range = p->search_range; K.reweight_distance = p->reweight_distance; K.match_treshold = p->match_treshold;
  K.min_match_threshold = p->min_match_threshold;
  K.pixel_uncertainty = p->pixel_uncertainty; K.quantile_cutoff = p->quantile_cutoff;
  K.quantile_num_bins = p->quantile_num_bins; K.reshape_q_abs = p->reshape_q_abs;
  K.pixel_uncertainty_match = p->pixel_uncertainty_match; K.match_threshold_norm = p->match_threshold_norm;
  K.cang_min_edge = std::cos(p->match_threshold_angle * M_PI / 180.0);  // edge_map.cpp:104
  K.regularization_threshold = p->regularization_threshold;
  K.nseg = div_up(p->cols, 64);
  K.df_nr = nr;

  // FastGaussian widths (scale_space.cpp:14-41,186)
  float st0 = 0, st1 = 0;
  hm::kovesi_widths(3.56359, 3, c->widths[0], &st0);
  hm::kovesi_widths(st0 * 1.2599, 3, c->widths[1], &st1);
  for (int f = 0; f < 2; ++f)
    for (int k = 0; k < 3; ++k)
      if (c->widths[f][k] < 3 || c->widths[f][k] > 11) return fail_msg("unexpected box width", -5);
  float recip[128], pinv[75];
  recip[0] = 0.f;
  for (int n = 1; n < 128; ++n) recip[n] = (float)(1.0 / (double)(float)n);
  hm::plane_fit_pinv(pinv);
  upload_tables(recip, pinv);

  // the tracking chain is the serial dependency of the pipeline: highest priority; the atomics-bound distance field is
  // needed one frame later: lowest
  int prio_least = 0, prio_greatest = 0;
  (void)hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
  int prio_mid = (prio_least + prio_greatest) / 2;
  if (const char* e = std::getenv("REBVIO_HIP_PRIO"))  // "flat": every stream at the default priority (diagnostic, several contexts per GPU)
    if (std::strcmp(e, "flat") == 0) prio_least = prio_greatest = prio_mid = 0;
  if (t_adopt_streams) {
    c->s_det = t_adopt_streams->s_det;
    c->s_key = t_adopt_streams->s_key;
    c->s_trk = t_adopt_streams->s_trk;
    c->owns_streams = false;
  } else {
    HIPCHK(hipStreamCreateWithPriority(&c->s_det, hipStreamNonBlocking, prio_mid));
    // one stream per priority class: the runtime pools hardware queues per class (GPU_MAX_HW_QUEUES each) and lets streams of
    // a class share a queue once the pool is full, so two of our streams in one class can end up serialised behind each
    // other depending on what else the process created (measured with extra torch streams: 9.3k -> 7.0k frames/s)
    HIPCHK(hipStreamCreateWithPriority(&c->s_key, hipStreamNonBlocking, prio_least));
    HIPCHK(hipStreamCreateWithPriority(&c->s_trk, hipStreamNonBlocking, prio_greatest));
    // Three streams per context (scans | keylines + distance field | tracking), deliberately not more: on this runtime every
    // additional stream of the process slowed the whole pipeline (measured, same code: 3 streams 9.3k frames/s, 4 streams
    // 9.2k, 5 streams 9.1k; creating a sixth, even unused, 2.6k). The rare copies of the synchronous API ride on the keyline stream.
  }
  const size_t Pn = (size_t)p->rows * p->cols;
  // the integral-image buffers have a row pitch of cols rounded up to 4 floats (the scans move 16-byte vectors); every
  // other per-pixel array is dense
  const size_t Pp = (size_t)p->rows * ((p->cols + 3) & ~3);
  for (int f = 0; f < 2; ++f) {
    HIPCHK(hipMalloc(&c->sb.a[f], Pp * sizeof(float)));
    HIPCHK(hipMalloc(&c->sb.b[f], Pp * sizeof(float)));
  }
  // (DoG / gradient buffers have the size of an integral image, padded pitch: a batch's fused path keeps the third pass's
  // integral images of a step parity in them)
  HIPCHK(hipMalloc(&c->sb.dog, Pp * sizeof(float)));
  HIPCHK(hipMalloc(&c->sb.mag, Pp * sizeof(float)));
  HIPCHK(hipMalloc(&c->db.stash, Pn * sizeof(float4)));
  HIPCHK(hipMalloc(&c->db.bits, (size_t)p->rows * K.nseg * sizeof(unsigned long long)));
  HIPCHK(hipMalloc(&c->db.rowcount, (size_t)p->rows * sizeof(int)));
  HIPCHK(hipMemset(c->db.rowcount, 0, (size_t)p->rows * sizeof(int)));
  c->dog2[0] = c->sb.dog;
  c->mag2[0] = c->sb.mag;
  c->rowcount2[0] = c->db.rowcount;
  for (int i = 0; i < 2; ++i) {
    HIPCHK(hipEventCreateWithFlags(&c->ev_scan[i], hipEventDisableTiming));
  }
  for (int f = 0; f < 2; ++f) {
    c->sa2[0][f] = c->sb.a[f];
    HIPCHK(hipMalloc(&c->sa2[1][f], Pp * sizeof(float)));
  }
  for (int i = 1; i < kDetPar; ++i) {
    HIPCHK(hipMalloc(&c->dog2[i], Pp * sizeof(float)));
    HIPCHK(hipMalloc(&c->mag2[i], Pp * sizeof(float)));
    HIPCHK(hipMalloc(&c->rowcount2[i], (size_t)p->rows * sizeof(int)));
    HIPCHK(hipMemset(c->rowcount2[i], 0, (size_t)p->rows * sizeof(int)));
  }
  HIPCHK(hipMalloc(&c->det, (kDetRing + 1) * sizeof(DetState)));
  DetState d0[kDetRing + 1];
  for (int i = 0; i < kDetRing + 1; ++i) {
    d0[i].threshold = p->threshold;       // config_->threshold
    d0[i].count = 0;                      // keylines_count_(0) (edge_detector.cpp:18)
    d0[i].auto_threshold = p->threshold;  // auto_threshold_(config_->threshold) (edge_detector.cpp:20)
    d0[i].pad = 0;
  }
  HIPCHK(hipMemcpy(c->det, d0, sizeof(d0), hipMemcpyHostToDevice));
  HIPCHK(hipMalloc(&c->img_dev, Pn * sizeof(float) + 16));  // (+16: host frames arrive in whole 16-byte units)
  HIPCHK(hipMalloc(&c->img8_dev, Pn + 16));
  HIPCHK(hipMalloc(&c->aos_dev, (size_t)p->keylines_max * sizeof(rebvio_hip_keyline)));
  HIPCHK(hipMalloc(&c->scratch_i, 2 * Pn * sizeof(int)));

  c->maxblocks = div_up(p->keylines_max, 1024) * 4;  // record groups of 256 keylines, padded to whole 1024-thread workgroups
  // [2 parity slots][record groups][kPartStride] + the final-velocity broadcast words
  HIPCHK(hipMalloc(&c->lm_xch, lm_xch_words(c->maxblocks) * sizeof(unsigned long long)));
  HIPCHK(hipMemset(c->lm_xch, 0, lm_xch_words(c->maxblocks) * sizeof(unsigned long long)));  // tag 0 = never published
  HIPCHK(hipHostMalloc(&c->lm_bar_err, 8 * sizeof(int), hipHostMallocDefault));
  std::memset(c->lm_bar_err, 0, 8 * sizeof(int));
  if (std::getenv("REBVIO_HIP_LM_STAMPS")) {
    HIPCHK(hipHostMalloc(&c->lm_stamps, 64 * sizeof(unsigned long long), hipHostMallocDefault));
    std::memset(c->lm_stamps, 0, 64 * sizeof(unsigned long long));
    K.dbg = c->lm_stamps;
    if (std::getenv("REBVIO_HIP_DM_STATS")) {
      c->dm_stats_words = 16 * (size_t)(p->keylines_max / 8 + 130);
      HIPCHK(hipMalloc(&c->dm_stats, c->dm_stats_words * sizeof(unsigned long long)));
      HIPCHK(hipMemset(c->dm_stats, 0, c->dm_stats_words * sizeof(unsigned long long)));
      K.dm_stats = c->dm_stats;
    }
  }
  // REBVIO_HIP_LM = percall (one kernel per evaluation) | seq (persistent kernel, one evaluation per exchange round) |
  // anything else / unset: persistent kernel with the speculative reject chain (track.hip, k_lm_chain_spec)
  if (const char* e = std::getenv("REBVIO_HIP_LM")) {
    c->lm_persistent = std::strcmp(e, "percall") != 0;
    c->lm_spec = std::strcmp(e, "seq") != 0;
    c->lm_spec_forced = std::strcmp(e, "spec") == 0 || std::strcmp(e, "spec3") == 0;
    c->lm_spec_forced_kf = std::strcmp(e, "spec3") == 0 ? 3 : 2;
    if (std::strncmp(e, "mix", 3) == 0) c->lm_mix = std::max(1, std::atoi(e + 3));
  }
  HIPCHK(hipMalloc(&c->lm, 16 * sizeof(LmState)));
  HIPCHK(hipMemset(c->lm, 0, 16 * sizeof(LmState)));
  HIPCHK(hipMalloc(&c->part, (size_t)(kMaxLmCalls + 1) * part_call_stride(c) * sizeof(float)));
  HIPCHK(hipMemset(c->part, 0, (size_t)(kMaxLmCalls + 1) * part_call_stride(c) * sizeof(float)));
  HIPCHK(hipMalloc(&c->xrv_part, (size_t)c->maxblocks * kXrvStride * sizeof(float)));
  HIPCHK(hipMalloc(&c->hist, 128 * sizeof(int)));
  HIPCHK(hipMemset(c->hist, 0, 128 * sizeof(int)));
  // long-search queue: keyline indices, then three float4 of probe geometry per entry (dm_queue_put in track.hip)
  HIPCHK(hipMalloc(&c->lm_zero, sizeof(LmState)));
  HIPCHK(hipMemset(c->lm_zero, 0, sizeof(LmState)));
  for (int i = 0; i < rebvio_hip_ctx::kSlots; ++i) {
    const size_t sz = sizeof(PairSlot) + (size_t)c->maxblocks * kXrvStride * sizeof(float);
    HIPCHK(hipHostMalloc(&c->slot[i], sz, hipHostMallocDefault));
    std::memset(c->slot[i], 0, sz);
    HIPCHK(hipHostMalloc(&c->rec[i], sizeof(GlueRec), hipHostMallocDefault));
    std::memset(c->rec[i], 0, sizeof(GlueRec));
    HIPCHK(hipEventCreateWithFlags(&c->slot_ev[i], hipEventDisableTiming));
  }
  HIPCHK(hipMalloc(&c->fscratch, 64 * sizeof(float)));
  HIPCHK(hipMalloc(&c->glue_dev, rebvio_hip_ctx::kSlots * sizeof(GlueDev)));
  HIPCHK(hipMemset(c->glue_dev, 0, rebvio_hip_ctx::kSlots * sizeof(GlueDev)));
  HIPCHK(hipMalloc(&c->glue_stage, rebvio_hip_ctx::kSlots * sizeof(GlueStage)));
  HIPCHK(hipMemset(c->glue_stage, 0, rebvio_hip_ctx::kSlots * sizeof(GlueStage)));
  HIPCHK(hipMalloc(&c->gstate, 2 * sizeof(GlueState)));
  HIPCHK(hipMemset(c->gstate, 0, 2 * sizeof(GlueState)));
  HIPCHK(hipHostMalloc(&c->h_gstate, 2 * sizeof(GlueState), hipHostMallocDefault));
  if (const char* l = std::getenv("REBVIO_HIP_LEAD")) c->lead = std::min(12, std::max(3, std::atoi(l)));
  if (const char* g = std::getenv("REBVIO_HIP_GROUP")) c->group = std::min(6, std::max(1, std::atoi(g)));
  if (const char* e = std::getenv("REBVIO_HIP_LM_THREADS")) {
    const int v = std::atoi(e);
    if (v == 256 || v == 512 || v == 1024) c->lm_threads = v;
  }
  if (const char* e = std::getenv("REBVIO_HIP_DM_HEAD"))  // directedMatch head form (track.hip, dm_head_wide): thread | wide; default by map size
    c->dm_head_form = dm_form_by_name(e);
  HIPCHK(hipHostMalloc(&c->h_lm, 2 * sizeof(LmState), hipHostMallocDefault));
  HIPCHK(hipHostMalloc(&c->h_part, part_call_stride(c) * sizeof(float), hipHostMallocDefault));
  HIPCHK(hipHostMalloc(&c->h_xrv, (size_t)c->maxblocks * kXrvStride * sizeof(float), hipHostMallocDefault));
  HIPCHK(hipHostMalloc(&c->h_st, 2 * sizeof(MapState), hipHostMallocDefault));
  HIPCHK(hipHostMalloc(&c->h_f, 64 * sizeof(float), hipHostMallocDefault));

  int pool = p->map_pool > 0 ? p->map_pool : 6;
  if (pool < 4) pool = 4;
  for (int i = 0; i < pool; ++i) {
    rebvio_hip_map* m = new rebvio_hip_map;
    m->ctx = c;
    m->life = c->life;
    c->pool.push_back(m);
    int rc = alloc_map(c, m);
    if (rc) return rc;
  }
  rebvio_hip_reset_state(c);
  c->dbg = std::getenv("REBVIO_HIP_DEBUG") != nullptr;
  if (const char* e = std::getenv("REBVIO_HIP_DETECT_WORKER")) c->det_worker = e[0] != '0';
  if (const char* e = std::getenv("REBVIO_HIP_GYRO_PRE")) c->gyro_pre_on = e[0] != '0';
  if (const char* e = std::getenv("REBVIO_HIP_FUSE_DOG")) c->fuse_dog = e[0] != '0';
  HIPCHK(hipDeviceSynchronize());
  guard.c = nullptr;
  *out = c;
  return 0;
}

void rebvio_hip_destroy(rebvio_hip_ctx* c) {
  if (!c) return;
  // Map handles the caller still holds outlive the context as inert husks: every map entry point checks `dead`, and the
  // husk is deleted by its rebvio_hip_map_release. The lock serialises this against a release running on another thread.
  const std::shared_ptr<LifeBlock> life = c->life;
  std::unique_lock<std::shared_mutex> life_lk(life->mu);
  residency_remove(c->device, c);
  (void)hipSetDevice(c->device);
  if (std::getenv("REBVIO_HIP_DEBUG") && c->t_begin_n)
    std::fprintf(stderr, "[rebvio_hip] track_pair_begin over %llu pairs (us): enqueue %.1f  wait for the first half %.1f\n",
                 (unsigned long long)c->t_begin_n, c->t_begin_enq / c->t_begin_n, c->t_begin_wait / c->t_begin_n);
  if (c->det_thread.joinable()) {
    {
      std::lock_guard<std::mutex> lk(c->det_mu);
      c->det_stop = true;
    }
    c->det_cv.notify_all();
    c->det_thread.join();
  }
  (void)hipDeviceSynchronize();
  {
    std::lock_guard<std::mutex> g(g_prof.mu);
    g_prof.drain();
  }
  // maps the streaming driver still holds are the library's own, not the caller's
  for (auto* m : c->frames) m->in_use = false;
  life->dead.store(true, std::memory_order_release);
  for (auto* m : c->pool) {
    free_map_device(m);
    if (!m->in_use) delete m;  // (else: a handle is still out; its release deletes the husk)
  }
  c->pool.clear();
  void* dptr[] = {c->sb.a[0], c->sb.a[1], c->sb.b[0], c->sb.b[1], c->sb.dog, c->sb.mag, c->db.stash, c->db.bits,
                  c->db.rowcount, c->det, c->img_dev, c->img8_dev, c->aos_dev, c->scratch_i, c->diag0, c->diag1, c->lm,
                  c->part, c->xrv_part, c->hist, c->fscratch};
  for (void* p : dptr)
    if (p) (void)hipFree(p);
  for (int f = 0; f < 2; ++f)
    if (c->sa2[1][f]) (void)hipFree(c->sa2[1][f]);
  void* hptr[] = {c->h_lm, c->h_part, c->h_xrv, c->h_st, c->h_f};
  for (void* p : hptr)
    if (p) (void)hipHostFree(p);
  if (c->owns_streams) {
    if (c->s_det) (void)hipStreamDestroy(c->s_det);
    if (c->s_trk) (void)hipStreamDestroy(c->s_trk);
    if (c->s_key) (void)hipStreamDestroy(c->s_key);
  }
  for (int i = 0; i < 2; ++i) {
    if (c->ev_scan[i]) (void)hipEventDestroy(c->ev_scan[i]);
  }
  for (int i = 1; i < kDetPar; ++i) {
    if (c->dog2[i]) (void)hipFree(c->dog2[i]);
    if (c->mag2[i]) (void)hipFree(c->mag2[i]);
    if (c->rowcount2[i]) (void)hipFree(c->rowcount2[i]);
  }
  for (int i = 0; i < rebvio_hip_ctx::kSlots; ++i) {
    if (c->slot[i]) (void)hipHostFree(c->slot[i]);
    if (c->rec[i]) (void)hipHostFree(c->rec[i]);
    if (c->slot_ev[i]) (void)hipEventDestroy(c->slot_ev[i]);
  }
  if (c->gstate) (void)hipFree(c->gstate);
  if (c->h_gstate) (void)hipHostFree(c->h_gstate);
  if (c->undist_map) (void)hipFree(c->undist_map);
  for (int i = 0; i < kDetPar; ++i)
    if (c->undist_img[i]) (void)hipFree(c->undist_img[i]);
  if (c->lm_xch) (void)hipFree(c->lm_xch);
  for (int i = 0; i < rebvio_hip_ctx::kPin; ++i) {
    if (c->pin[i]) (void)hipHostFree(c->pin[i]);
    if (c->pin_ev[i]) (void)hipEventDestroy(c->pin_ev[i]);
  }
  if (c->glue_dev) (void)hipFree(c->glue_dev);
  if (c->glue_stage) (void)hipFree(c->glue_stage);
  if (c->lm_bar_err) (void)hipHostFree(c->lm_bar_err);
  if (c->lm_stamps) (void)hipHostFree(c->lm_stamps);
  if (c->dm_stats) (void)hipFree(c->dm_stats);
  for (auto& e : c->bf_done)
    if (e) (void)hipEventDestroy(e);
  if (c->h_bf) (void)hipHostFree(c->h_bf);
  if (c->lm_zero) (void)hipFree(c->lm_zero);
  delete c;
}

int rebvio_hip_scale_space(rebvio_hip_ctx* c, const float* img, float* scale0, float* scale1, float* dog, float* mag) {
  HIPCHK(hipSetDevice(c->device));
  const size_t nb = (size_t)c->P.rows * c->P.cols * sizeof(float);
  if (!c->diag0) {
    HIPCHK(hipMalloc(&c->diag0, nb));
    HIPCHK(hipMalloc(&c->diag1, nb));
  }
  HIPCHK(hipMemcpyAsync(c->img_dev, img, nb, hipMemcpyHostToDevice, c->s_det));
  ScaleBufs sb = c->sb;
  sb.scale0 = c->diag0;
  sb.scale1 = c->diag1;
  launch_scale_space(c->s_det, c->K, c->img_dev, 0, sb, c->widths, c->db.rowcount);
  HIPCHK(hipGetLastError());
  if (scale0) HIPCHK(hipMemcpyAsync(scale0, c->diag0, nb, hipMemcpyDeviceToHost, c->s_det));
  if (scale1) HIPCHK(hipMemcpyAsync(scale1, c->diag1, nb, hipMemcpyDeviceToHost, c->s_det));
  if (dog) HIPCHK(hipMemcpyAsync(dog, c->sb.dog, nb, hipMemcpyDeviceToHost, c->s_det));
  if (mag) HIPCHK(hipMemcpyAsync(mag, c->sb.mag, nb, hipMemcpyDeviceToHost, c->s_det));
  HIPCHK(hipStreamSynchronize(c->s_det));
  return 0;
}

int rebvio_hip_detect(rebvio_hip_ctx* c, const float* img, size_t pitch_bytes, uint64_t ts_us, rebvio_hip_map** out) {
  HIPCHK(hipSetDevice(c->device));
  const size_t rowb = (size_t)c->P.cols * sizeof(float);
  if (pitch_bytes == 0) pitch_bytes = rowb;
  return detect_host_frame(c, img, pitch_bytes, rowb, c->img_dev, 0, ts_us, out);
}

int rebvio_hip_detect_u8_device(rebvio_hip_ctx* c, const uint8_t* frame_dev, uint64_t ts_us, rebvio_hip_map** out) {
  HIPCHK(hipSetDevice(c->device));
  return detect_common(c, frame_dev, 1, ts_us, out);
}

int rebvio_hip_detect_u8(rebvio_hip_ctx* c, const uint8_t* img, size_t pitch_bytes, uint64_t ts_us, rebvio_hip_map** out) {
  HIPCHK(hipSetDevice(c->device));
  const size_t rowb = (size_t)c->P.cols;
  if (pitch_bytes == 0) pitch_bytes = rowb;
  return detect_host_frame(c, img, pitch_bytes, rowb, c->img8_dev, 1, ts_us, out);
}

int rebvio_hip_set_undistort(rebvio_hip_ctx* c, const float K4[4], const float D5[5]) {
  HIPCHK(hipSetDevice(c->device));
  HIPCHK(hipDeviceSynchronize());  // no frame in flight may still read the old map
  bool any = false;
  for (int i = 0; i < 5; ++i) any = any || (D5[i] != 0.0f);
  if (!any) {  // identity: the fixed-point map reproduces x3 exactly, skip the gather
    if (c->undist_map) (void)hipFree(c->undist_map);
    c->undist_map = nullptr;
    return 0;
  }
  if (!(K4[0] > 0.0f) || !(K4[1] > 0.0f)) return fail_msg("set_undistort: focal lengths must be positive", -3);
  const size_t Pn = (size_t)c->P.rows * c->P.cols;
  std::vector<int> map(2 * Pn);
  hm::undistort_fixed_map(c->P.rows, c->P.cols, K4[0], K4[1], K4[2], K4[3], D5[0], D5[1], D5[2], D5[3], D5[4], map.data());
  if (!c->undist_map) HIPCHK(hipMalloc(&c->undist_map, Pn * sizeof(int2)));
  for (int i = 0; i < kDetPar; ++i)
    if (!c->undist_img[i]) HIPCHK(hipMalloc(&c->undist_img[i], Pn * sizeof(float)));
  HIPCHK(hipMemcpy(c->undist_map, map.data(), Pn * sizeof(int2), hipMemcpyHostToDevice));
  return 0;
}

int rebvio_hip_front_end_u8(rebvio_hip_ctx* c, const uint8_t* img, float* out) {
  HIPCHK(hipSetDevice(c->device));
  if (!c->undist_map) return fail_msg("front_end_u8: no distortion model set (rebvio_hip_set_undistort)", -3);
  const size_t Pn = (size_t)c->P.rows * c->P.cols;
  HIPCHK(hipMemcpyAsync(c->img8_dev, img, Pn, hipMemcpyHostToDevice, c->s_det));
  launch_front_end_u8(c->s_det, c->K, c->img8_dev, c->undist_map, c->undist_img[0]);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(out, c->undist_img[0], Pn * sizeof(float), hipMemcpyDeviceToHost, c->s_det));
  HIPCHK(hipStreamSynchronize(c->s_det));
  return 0;
}

int rebvio_hip_detector_state(rebvio_hip_ctx* c, float* threshold, float* auto_threshold, int* count) {
  HIPCHK(hipSetDevice(c->device));
  while (c->det_pending.load(std::memory_order_acquire) > 0) std::this_thread::yield();
  HIPCHK(hipStreamSynchronize(c->s_det));
  HIPCHK(hipStreamSynchronize(c->s_key));
  DetState d;
  HIPCHK(hipMemcpy(&d, c->det + (c->frame_index % kDetRing), sizeof(d), hipMemcpyDeviceToHost));
  float at = d.auto_threshold;
  if (c->last_detected) {  // tuneThreshold of the last frame (edge_detector.cpp:184), same fp32 formula as the kernels
    MapState st;
    HIPCHK(hipMemcpy(&st, c->last_detected->d.st, sizeof(st), hipMemcpyDeviceToHost));
    if (st.n > 0) {
      float mx, mn;
      std::memcpy(&mx, &st.gmax_bits, 4);
      std::memcpy(&mn, &st.gmin_bits, 4);
      at = mx - float(kNumBins * (mx - mn)) / float(kNumBins);
    }
  }
  if (threshold) *threshold = d.threshold;
  if (auto_threshold) *auto_threshold = at;
  if (count) *count = d.count;
  return 0;
}

int rebvio_hip_map_size(rebvio_hip_map* m) {
  MAP_ALIVE_OR(m, -10);
  if (ensure_size(m)) return -1;
  return m->n_host;
}
float rebvio_hip_map_threshold(rebvio_hip_map* m) {
  MAP_ALIVE_OR(m, std::numeric_limits<float>::quiet_NaN());
  if (ensure_size(m)) return std::numeric_limits<float>::quiet_NaN();
  return m->thr_host;
}
uint64_t rebvio_hip_map_ts(rebvio_hip_map* m) { return m->ts; }

int rebvio_hip_map_download(rebvio_hip_map* m, rebvio_hip_keyline* keylines, int* mask) {
  MAP_ALIVE_OR(m, -10);
  rebvio_hip_ctx* c = m->ctx;
  HIPCHK(hipSetDevice(c->device));
  int rc = ensure_size(m);
  if (rc) return rc;
  // one packing buffer per context (callbacks and the fusion thread may both mirror maps); the same mutex orders the
  // "has the tracker touched this map" decision below against the tracker's first use of it (trk_wait_ready)
  std::lock_guard<std::mutex> dl(c->dl_mu);
  // The mirror reflects everything enqueued so far that concerns THIS map: its detection and distance field (`ready`,
  // recorded behind them; ensure_size has waited for the detect worker to enqueue them) and, once the tracker has used the
  // map, the track stream. A map fresh from detect() (edge-image callbacks, ros_rebvio.cpp:32-50) does not wait for the
  // tracker - which may be a pair ahead, or hold a second half parked until the fusion thread releases it.
  if (m->trk_touched.load(std::memory_order_acquire) || !c->owns_streams) {
    HIPCHK(hipStreamSynchronize(c->s_det));
    HIPCHK(hipStreamSynchronize(c->s_key));
    HIPCHK(hipStreamSynchronize(c->s_key));
    { const int rc_ts = trk_sync(c); if (rc_ts) return rc_ts; }
  } else {
    wait_enqueued(m);
    HIPCHK(hipEventSynchronize(m->ready));
  }
  if (keylines && m->n_host > 0) {
    launch_map_pack(c->s_key, c->K, m->d, c->aos_dev);
    HIPCHK(hipMemcpyAsync(keylines, c->aos_dev, (size_t)m->n_host * sizeof(rebvio_hip_keyline), hipMemcpyDeviceToHost, c->s_key));
  }
  if (mask)
    HIPCHK(hipMemcpyAsync(mask, m->d.mask, (size_t)c->P.rows * c->P.cols * sizeof(int), hipMemcpyDeviceToHost, c->s_key));
  HIPCHK(hipStreamSynchronize(c->s_key));
  return 0;
}

int rebvio_hip_render_edge_image(rebvio_hip_map* m, const uint8_t* gray, uint8_t* rgb_out) {
  MAP_ALIVE_OR(m, -10);
  rebvio_hip_ctx* c = m->ctx;
  HIPCHK(hipSetDevice(c->device));
  std::lock_guard<std::mutex> dl(c->dl_mu);
  HIPCHK(hipStreamSynchronize(c->s_det));  // the staging frame and the scratch are shared with the detect path
  HIPCHK(hipStreamSynchronize(c->s_key));
  HIPCHK(hipStreamSynchronize(c->s_key));
  { const int rc_ts = trk_sync(c); if (rc_ts) return rc_ts; }
  const size_t Pn = (size_t)c->P.rows * c->P.cols;
  uint8_t* rgb = reinterpret_cast<uint8_t*>(c->scratch_i);  // 8 bytes/pixel available, 3 used
  if (gray) HIPCHK(hipMemcpyAsync(c->img8_dev, gray, Pn, hipMemcpyHostToDevice, c->s_key));
  launch_render_edge_image(c->s_key, c->K, m->d, gray ? c->img8_dev : nullptr, rgb);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(rgb_out, rgb, 3 * Pn, hipMemcpyDeviceToHost, c->s_key));
  HIPCHK(hipStreamSynchronize(c->s_key));
  return 0;
}

int rebvio_hip_map_upload(rebvio_hip_map* m, const rebvio_hip_keyline* keylines, int n) {
  MAP_ALIVE_OR(m, -10);
  rebvio_hip_ctx* c = m->ctx;
  HIPCHK(hipSetDevice(c->device));
  int rc = ensure_size(m);
  if (rc) return rc;
  if (n != m->n_host) return fail_msg("map_upload: count differs from map size", -6);
  std::lock_guard<std::mutex> dl(c->dl_mu);
  HIPCHK(hipStreamSynchronize(c->s_key));
  HIPCHK(hipStreamSynchronize(c->s_det));
  HIPCHK(hipStreamSynchronize(c->s_key));
  { const int rc_ts = trk_sync(c); if (rc_ts) return rc_ts; }
  if (n > 0) {
    HIPCHK(hipMemcpyAsync(c->aos_dev, keylines, (size_t)n * sizeof(rebvio_hip_keyline), hipMemcpyHostToDevice, c->s_key));
    launch_map_unpack(c->s_key, c->K, m->d, c->aos_dev, n);
  }
  HIPCHK(hipStreamSynchronize(c->s_key));
  m->df_built = false;
  m->raster_order = false;  // arbitrary keylines: the rebuild goes through the scatter kernel
  return 0;
}

void rebvio_hip_map_release(rebvio_hip_map* m) {
  if (!m) return;
  const std::shared_ptr<LifeBlock> life = m->life;
  std::shared_lock<std::shared_mutex> lk(life->mu);  // (against destroy; the pool's own state is under the context's pool_mu)
  if (life->dead.load(std::memory_order_acquire)) {  // the context is gone (its memory with it): drop the husk
    if (m->in_use) delete m;
    return;
  }
  release_map(m, nullptr);
}

}  // extern "C"

namespace {
// Stream-ordered return of a map to its pool (the library's own releases call this directly; the C-ABI entry adds the
// lifetime check).
// done_ref: an event already recorded behind the map's last consumer (null: one is recorded here).
void release_map(rebvio_hip_map* m, hipEvent_t done_ref = nullptr) {
  if (!m || !m->in_use) return;
  rebvio_hip_ctx* c = m->ctx;
  (void)hipSetDevice(c->device);
  wait_enqueued(m);
  // a pair's counters still waiting in this map's state record (track_pair_finish_async without its _result yet): copy them
  // out in stream order before the map can be handed to another frame
  if (c->h_bf && c->bf_res >= 0 && c->bf_map[c->bf_res] == m && !c->bf_have[c->bf_res] && !c->bf_copy_queued[c->bf_res]) {
    const int r = c->bf_res;
    if (hipMemcpyAsync(&c->h_bf[r], m->d.st, sizeof(MapState), hipMemcpyDeviceToHost, c->s_trk) == hipSuccess &&
        hipEventRecord(c->bf_done[r], c->s_trk) == hipSuccess)
      c->bf_copy_queued[r] = true;
  }
  std::lock_guard<std::mutex> pool_lk(c->pool_mu);
  m->done_ref = done_ref;
  if (!done_ref) (void)hipEventRecord(m->done, c->s_trk);
  m->has_done = true;
  if (c->df_map == m) c->df_map = nullptr;
  m->release_seq = ++c->release_counter;
  m->in_use = false;  // (c->last_detected may keep pointing at it: only its MapState is read, stream-ordered)
}
}  // namespace

extern "C" {

int rebvio_hip_build_distance_field(rebvio_hip_ctx* c, rebvio_hip_map* m) {
  HIPCHK(hipSetDevice(c->device));
  wait_enqueued(m);
  HIPCHK(trk_wait_ready_once(c, m));
  if (!m->df_built) {
    if (!m->raster_order) HIPCHK(hipMemsetAsync(m->d.df, 0xFF, (size_t)c->P.rows * c->P.cols * sizeof(unsigned), c->s_trk));
    launch_df_build(c->s_trk, c->K, m->d, c->det + (c->frame_index % kDetRing), m->raster_order);  // keylines may have been uploaded
    HIPCHK(hipGetLastError());
    m->df_built = true;
  }
  c->df_map = m;
  return 0;
}

int rebvio_hip_distance_field(rebvio_hip_ctx* c, int* id_out, int* dist_out) {
  HIPCHK(hipSetDevice(c->device));
  if (!c->df_map) return fail_msg("no distance field built", -7);
  std::lock_guard<std::mutex> dl(c->dl_mu);
  const size_t Pn = (size_t)c->P.rows * c->P.cols;
  launch_df_decode(c->s_trk, c->K, c->df_map->d, c->scratch_i, c->scratch_i + Pn);
  HIPCHK(hipGetLastError());
  if (id_out) HIPCHK(hipMemcpyAsync(id_out, c->scratch_i, Pn * sizeof(int), hipMemcpyDeviceToHost, c->s_trk));
  if (dist_out) HIPCHK(hipMemcpyAsync(dist_out, c->scratch_i + Pn, Pn * sizeof(int), hipMemcpyDeviceToHost, c->s_trk));
  { const int rc_ts = trk_sync(c); if (rc_ts) return rc_ts; }
  return 0;
}

int rebvio_hip_map_distance_field(rebvio_hip_map* m, int* id_out, int* dist_out) {
  MAP_ALIVE_OR(m, -10);
  rebvio_hip_ctx* c = m->ctx;
  HIPCHK(hipSetDevice(c->device));
  if (!m->df_built) return fail_msg("map_distance_field: no distance field has been built from this map", -7);
  wait_enqueued(m);
  HIPCHK(trk_wait_ready(c->s_trk, m));
  std::lock_guard<std::mutex> dl(c->dl_mu);
  const size_t Pn = (size_t)c->P.rows * c->P.cols;
  launch_df_decode(c->s_trk, c->K, m->d, c->scratch_i, c->scratch_i + Pn);
  HIPCHK(hipGetLastError());
  if (id_out) HIPCHK(hipMemcpyAsync(id_out, c->scratch_i, Pn * sizeof(int), hipMemcpyDeviceToHost, c->s_trk));
  if (dist_out) HIPCHK(hipMemcpyAsync(dist_out, c->scratch_i + Pn, Pn * sizeof(int), hipMemcpyDeviceToHost, c->s_trk));
  { const int rc_ts = trk_sync(c); if (rc_ts) return rc_ts; }
  return 0;
}

int rebvio_hip_search_match(rebvio_hip_ctx* c, rebvio_hip_map* searched, const rebvio_hip_keyline* query, const float vel[3],
                            const float Rvel[9], const float Rback[9], float max_radius, int* idx_out) {
  HIPCHK(hipSetDevice(c->device));
  if (!(query->gradient_norm > 0.0f)) return fail_msg("search_match: the query keyline needs a positive gradient_norm", -3);
  wait_enqueued(searched);
  HIPCHK(trk_wait_ready(c->s_trk, searched));
  int* out_dev = reinterpret_cast<int*>(c->fscratch) + 32;
  launch_search_match_one(c->s_trk, c->K, searched->d, *query, vel, Rvel, Rback, max_radius, out_dev);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(c->h_f + 32, out_dev, sizeof(int), hipMemcpyDeviceToHost, c->s_trk));
  { const int rc_ts = trk_sync(c); if (rc_ts) return rc_ts; }
  std::memcpy(idx_out, c->h_f + 32, sizeof(int));
  return 0;
}

int rebvio_hip_smooth_n(rebvio_hip_ctx* c, const float* img, const int* widths, int n, float* out) {
  HIPCHK(hipSetDevice(c->device));
  if (n < 1 || n > 16) return fail_msg("smooth: 1..16 box passes", -3);
  for (int k = 0; k < n; ++k)
    if (widths[k] < 3 || widths[k] > 11 || (widths[k] & 1) == 0) return fail_msg("smooth: box widths must be odd and in 3..11", -3);
  const size_t nb = (size_t)c->P.rows * c->P.cols * sizeof(float);
  if (!c->diag0) {
    HIPCHK(hipMalloc(&c->diag0, nb));
    HIPCHK(hipMalloc(&c->diag1, nb));
  }
  while (c->det_pending.load(std::memory_order_acquire) > 0) std::this_thread::yield();
  HIPCHK(hipStreamSynchronize(c->s_key));  // the scratch DoG / gradient buffers below belong to frames in flight
  HIPCHK(hipMemcpyAsync(c->img_dev, img, nb, hipMemcpyHostToDevice, c->s_det));
  ScaleBufs sb = c->sb;
  sb.scale0 = c->diag0;
  sb.scale1 = c->diag1;
  launch_smooth_n(c->s_det, c->K, c->img_dev, sb, widths, n, c->db.rowcount);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(out, c->diag0, nb, hipMemcpyDeviceToHost, c->s_det));
  HIPCHK(hipStreamSynchronize(c->s_det));
  return 0;
}

int rebvio_hip_smooth(rebvio_hip_ctx* c, const float* img, const int widths3[3], float* out) { return rebvio_hip_smooth_n(c, img, widths3, 3, out); }

int rebvio_hip_rotate(rebvio_hip_ctx* c, rebvio_hip_map* m, const float R[9]) {
  HIPCHK(hipSetDevice(c->device));
  HIPCHK(trk_wait_ready(c->s_trk, m));
  launch_rotate(c->s_trk, c->K, m->d, R, nullptr, 0);
  HIPCHK(hipGetLastError());
  return 0;
}

int rebvio_hip_quantile(rebvio_hip_ctx* c, rebvio_hip_map* m, float percentile, int num_bins, float* out) {
  HIPCHK(hipSetDevice(c->device));
  if (num_bins < 1 || num_bins > 128) return fail_msg("num_bins must be in 1..128", -3);
  HIPCHK(trk_wait_ready(c->s_trk, m));
  launch_quantile(c->s_trk, c->K, m->d, c->hist, percentile, num_bins, c->fscratch);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(c->h_f, c->fscratch, sizeof(float), hipMemcpyDeviceToHost, c->s_trk));
  HIPCHK(hipMemsetAsync(c->hist, 0, 128 * sizeof(int), c->s_trk));  // invariant: the histogram is zero between uses
  { const int rc_ts = trk_sync(c); if (rc_ts) return rc_ts; }
  *out = c->h_f[0];
  return 0;
}

int rebvio_hip_try_vel(rebvio_hip_ctx* c, rebvio_hip_map* m, const float vel[3], float sigma_rho_min, float* residuals,
                       float out10[10]) {
  HIPCHK(hipSetDevice(c->device));
  if (!c->df_map) return fail_msg("try_vel needs a distance field", -7);
  int rc = ensure_size(m);
  if (rc) return rc;
  const int n = m->n_host;
  HIPCHK(trk_wait_ready(c->s_trk, m));
  if (n > 0) HIPCHK(hipMemcpyAsync(m->d.residual, residuals, (size_t)n * sizeof(float), hipMemcpyHostToDevice, c->s_trk));
  LmState st;
  std::memset(&st, 0, sizeof(st));
  for (int i = 0; i < 3; ++i) st.Vnew[i] = st.vel[i] = vel[i];
  st.sigma_rho_min = sigma_rho_min;
  c->h_lm[1] = st;
  HIPCHK(hipMemcpyAsync(c->lm, &c->h_lm[1], sizeof(LmState), hipMemcpyHostToDevice, c->s_trk));
  launch_try_vel(c->s_trk, c->K, m->d, c->df_map->d, 0, 0, 0, c->lm, c->lm + 1, c->part, c->part, c->hist, 0);
  HIPCHK(hipGetLastError());
  const int nb = std::max(1, div_up(n, 256));
  HIPCHK(hipMemcpyAsync(c->h_part, c->part, (size_t)nb * kPartStride * sizeof(float), hipMemcpyDeviceToHost, c->s_trk));
  if (n > 0) HIPCHK(hipMemcpyAsync(residuals, m->d.residual, (size_t)n * sizeof(float), hipMemcpyDeviceToHost, c->s_trk));
  { const int rc_ts = trk_sync(c); if (rc_ts) return rc_ts; }
  for (int k = 0; k < 10; ++k) {
    // the order of the device-side reducer (track.hip reduce_staged_records): record b goes to lane b % 16, a lane adds its
    // records in ascending order from 0, the sixteen lanes are scanned with shifts 1, 2, 4, 8 (zero fill) - the sums a
    // stand-alone tryVel reports are bit for bit the ones minimizeVel's kernels work with
    float part[16];
    const int nblk = div_up(n, 256);
    for (int j = 0; j < 16; ++j) {
      float acc = 0.f;
      for (int b = j; b < nblk; b += 16) acc += c->h_part[(size_t)b * kPartStride + k];
      part[j] = acc;
    }
    for (int d = 1; d < 16; d <<= 1)
      for (int i = 15; i >= 0; --i) part[i] = part[i] + (i >= d ? part[i - d] : 0.0f);
    out10[k] = part[15];
  }
  // resolve cross-workgroup carry markers (a later device call would do this in its prologue)
  float carry = 0.f;
  for (int b = 0; b < div_up(n, 256); ++b) {
    const int lo = b * 256, hi = std::min(n, lo + 256);
    for (int i = lo; i < hi; ++i)
      if (residuals[i] == kResidualCarry) residuals[i] = std::fabs(carry);
    if (c->h_part[(size_t)b * kPartStride + 10] != 0.f) carry = c->h_part[(size_t)b * kPartStride + 11];
  }
  return 0;
}

int rebvio_hip_minimize_vel(rebvio_hip_ctx* c, rebvio_hip_map* m, float vel[3], float Rvel[9], float* F, int* accept_mask,
                            float* sigma_rho_min) {
  HIPCHK(hipSetDevice(c->device));
  if (!c->df_map) return fail_msg("minimize_vel needs a distance field", -7);
  HIPCHK(trk_wait_ready(c->s_trk, m));
  HIPCHK(hipMemsetAsync(m->d.residual, 0, (size_t)c->P.keylines_max * sizeof(float), c->s_trk));
  HIPCHK(hipMemsetAsync(c->hist, 0, 128 * sizeof(int), c->s_trk));
  // histogram of sigma_rho for estimateQuantile (core.cpp:153)
  launch_quantile(c->s_trk, c->K, m->d, c->hist, c->P.quantile_cutoff, c->P.quantile_num_bins, c->fscratch);
  enqueue_lm_chain(c, m, c->df_map, vel);
  const int calls = (int)c->P.iterations + 1;
  launch_lm_final(c->s_trk, m->d, calls, c->lm + calls, c->lm + calls + 1,
                  c->part + (size_t)(calls - 1) * part_call_stride(c));
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(&c->h_lm[0], c->lm + calls + 1, sizeof(LmState), hipMemcpyDeviceToHost, c->s_trk));
  HIPCHK(hipMemsetAsync(c->hist, 0, 128 * sizeof(int), c->s_trk));  // invariant: the histogram is zero between uses
  { const int rc_ts = trk_sync(c); if (rc_ts) return rc_ts; }
  lm_to_out(c->h_lm[0], vel, Rvel, F, accept_mask, sigma_rho_min);
  return 0;
}

int rebvio_hip_forward_match(rebvio_hip_ctx* c, rebvio_hip_map* om, rebvio_hip_map* nm) {
  HIPCHK(hipSetDevice(c->device));
  HIPCHK(trk_wait_ready(c->s_trk, om));
  HIPCHK(trk_wait_ready(c->s_trk, nm));
  launch_forward_keys(c->s_trk, c->K, om->d, nm->d);
  const float v0[3] = {0, 0, 0};
  launch_ext_rot_vel(c->s_trk, c->K, om->d, nm->d, 1, 0, 0, c->lm, c->lm, c->part, c->xrv_part, v0, nullptr, nullptr);
  HIPCHK(hipGetLastError());
  return 0;
}

int rebvio_hip_ext_rot_vel(rebvio_hip_ctx* c, const float vel[3], float Wx[36], float JtF[6], float X[6], int* ok) {
  HIPCHK(hipSetDevice(c->device));
  if (!c->df_map) return fail_msg("ext_rot_vel needs a distance field", -7);
  rebvio_hip_map* nm = c->df_map;
  int rc = ensure_size(nm);
  if (rc) return rc;
  launch_ext_rot_vel(c->s_trk, c->K, nm->d, nm->d, 0, 0, 0, c->lm, c->lm, c->part, c->xrv_part, vel, nullptr, nullptr);
  HIPCHK(hipGetLastError());
  const int nb = std::max(1, div_up(nm->n_host, 256));
  HIPCHK(hipMemcpyAsync(c->h_xrv, c->xrv_part, (size_t)nb * kXrvStride * sizeof(float), hipMemcpyDeviceToHost, c->s_trk));
  { const int rc_ts = trk_sync(c); if (rc_ts) return rc_ts; }
  float jtf[6];
  sum_xrv(c->h_xrv, div_up(nm->n_host, 256), Wx, jtf, nullptr);
  if (JtF) std::memcpy(JtF, jtf, sizeof(jtf));
  hm::sym6_pinv_solve(Wx, jtf, X);
  int good = 1;
  for (int i = 0; i < 6; ++i)
    if (std::isnan(X[i])) good = 0;
  if (ok) *ok = good;
  return 0;
}

int rebvio_hip_directed_match(rebvio_hip_ctx* c, rebvio_hip_map* nm, rebvio_hip_map* om, const float vel[3],
                              const float Rvel[9], const float Rback[9], float max_radius, int* matches, int* kf_matches) {
  HIPCHK(hipSetDevice(c->device));
  // (a probe's slot index 2 * step + side takes 10 bits of a candidate-list entry of k_directed_match_c; rebvio_hip_create bounds
  // search_range the same way)
  if (!(max_radius >= 0.f) || max_radius > 255.f) return fail_msg("directed_match: max_radius outside 0..255", -3);
  HIPCHK(trk_wait_ready(c->s_trk, om));
  HIPCHK(trk_wait_ready(c->s_trk, nm));
  float vel_r[3], Rvel_r[9];
  rotate_inputs(c, vel, Rvel, Rback, vel_r, Rvel_r);
  HIPCHK(hipMemsetAsync(&nm->d.st->dm_matches, 0, 4 * sizeof(int), c->s_trk));  // dm_matches, dm_kf, reg_count, dm_queued
  launch_directed_match(c->s_trk, c->K, nm->d, om->d, vel_r, Rvel_r, Rback, max_radius, nullptr, c->dm_head_form);
  HIPCHK(hipGetLastError());
  MapState st;
  int rc = fetch_map_state(nm, &st, c->s_trk);
  if (rc) return rc;
  if (matches) *matches = st.dm_matches;
  if (kf_matches) *kf_matches = st.dm_kf;
  return 0;
}

int rebvio_hip_regularize(rebvio_hip_ctx* c, rebvio_hip_map* m, int* count) {
  HIPCHK(hipSetDevice(c->device));
  HIPCHK(trk_wait_ready(c->s_trk, m));
  HIPCHK(hipMemsetAsync(&m->d.st->reg_count, 0, sizeof(int), c->s_trk));
  launch_regularize(c->s_trk, c->K, m->d, 0);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(m->d.rs, m->d.rs_tmp, (size_t)c->P.keylines_max * sizeof(float2), hipMemcpyDeviceToDevice, c->s_trk));
  MapState st;
  int rc = fetch_map_state(m, &st, c->s_trk);
  if (rc) return rc;
  if (count) *count = st.reg_count;
  return 0;
}

int rebvio_hip_update_inverse_depth(rebvio_hip_ctx* c, const float vel[3]) {
  HIPCHK(hipSetDevice(c->device));
  if (!c->df_map) return fail_msg("update_inverse_depth needs a distance field", -7);
  launch_depth_ekf(c->s_trk, c->K, c->df_map->d, vel, 0, 0);
  HIPCHK(hipGetLastError());
  return 0;
}

namespace {
// Host glue between extRotVel and directedMatch (rebvio.cpp:177-233, accelerometer/SAB branch excluded) for the per-pair
// API: the shared core of glue.hpp (the streaming and batch drivers run the same statements on the device). R is the prior
// rotation used for the first rotateKeylines.
struct GlueOut {
  float R0a[9], Rgva[9], V[3], P_V[9];
  bool nan_v;
};
GlueParams glue_params(const rebvio_hip_ctx* c, float frame_dt) {
  GlueParams gp;
  gp.frame_dt = frame_dt;
  gp.gyro_std_dev = c->P.gyro_std_dev;
  gp.gyro_bias_std_dev = c->P.gyro_bias_std_dev;
  gp.has_pre = 0;
  std::memset(gp.pre, 0, sizeof(gp.pre));
  return gp;
}
// Streaming driver: the pair's gyroBiasCorrection matrices from the host's shadow of the device's W_Bg (GlueParams::pre); the
// shadow advances with every pair queued. Not valid (after rebvio_hip_set_gyro_state, before the first pair of a stream has
// uploaded the host's state): the device forms them itself.
void glue_params_pre(rebvio_hip_ctx* c, GlueParams* gp) {
  if (!c->gyro_pre_on || !c->wbg_shadow_valid) return;
  const float s_b = gp->gyro_bias_std_dev * gp->gyro_bias_std_dev * gp->frame_dt * gp->frame_dt;  // (rebvio.cpp:186-191, as the glue forms them)
  const float s_g = gp->gyro_std_dev * gp->gyro_std_dev * gp->frame_dt * gp->frame_dt;
  hm::gyro_pre(c->wbg_shadow, s_g, s_b, gp->pre);
  gp->has_pre = 1;
  c->wbg_shadow = hm::load3(gp->pre[2]);
}
GlueOut pair_glue(rebvio_hip_ctx* c, const LmState& lm, const float* xrv, int n_new, float frame_dt, hm::M3 R,
                  rebvio_hip_pair_out* out) {
  GlueState st;
  for (int i = 0; i < 3; ++i) st.Bg[i] = c->Bg[i];
  hm::store3(c->W_Bg, st.W_Bg);
  hm::store3(R, st.R);
  st.pad = 0.f;
  GlueDev gl;
  hm::pair_glue_core(lm, xrv, n_new, glue_params(c, frame_dt), st, gl, *out);
  for (int i = 0; i < 3; ++i) c->Bg[i] = st.Bg[i];
  c->W_Bg = hm::load3(st.W_Bg);
  note_accept_mask(c, out->lm_accept_mask);
  GlueOut g;
  std::memcpy(g.R0a, gl.R0a, sizeof(g.R0a));
  std::memcpy(g.Rgva, gl.Rgva, sizeof(g.Rgva));
  std::memcpy(g.V, gl.V, sizeof(g.V));
  std::memcpy(g.P_V, out->P_V, sizeof(g.P_V));
  g.nan_v = gl.nan_v != 0;
  return g;
}

// rotate by R0, directedMatch, regularize1Iter, updateInverseDepth (rebvio.cpp:232-259) on the track stream:
// three launches - the rotation of the old map is applied on the fly inside directedMatch (the old map is dead
// afterwards), regularize + depth EKF are one kernel. RT_next (streaming driver only) additionally applies the next
// pair's first rotateKeylines to the new map and bins its sigma_rho histogram.
void enqueue_b_chain(rebvio_hip_ctx* c, rebvio_hip_map* om, rebvio_hip_map* nm, const GlueOut& g, const float* RT_next) {
  hipStream_t s = c->s_trk;
  if (g.nan_v) {  // rebvio.cpp:236: no matching / depth update
    if (RT_next) {
      launch_rotate(s, c->K, nm->d, RT_next, c->hist, 0);
      nm->pre_rotated = true;
    }
    return;
  }
  float vel_r[3], Rvel_r[9];
  rotate_inputs(c, g.V, g.P_V, g.Rgva, vel_r, Rvel_r);
  launch_directed_match(s, c->K, nm->d, om->d, vel_r, Rvel_r, g.Rgva, c->P.search_range, g.R0a, c->dm_head_form);
  const int gate = (int)c->P.global_min_matches_threshold;
  launch_regularize_ekf(s, c->K, nm->d, g.V, gate > 0 ? gate : 0, RT_next, c->hist);  // rebvio.cpp:256-259
  std::swap(nm->d.rs, nm->d.rs_tmp);
  if (RT_next) {
    std::swap(nm->d.grad, nm->d.grad_tmp);
    nm->pre_rotated = true;
  }
}

hm::M3 prior_rotation(rebvio_hip_ctx* c, const float* R_prior) {
  // R = imu.R(); R.T() = SO3(Bg) * R.T()  (rebvio.cpp:163-164)
  return hm::prior_rotation(c->Bg, R_prior ? hm::load3(R_prior) : hm::identity3());
}
}  // namespace

int rebvio_hip_track_pair(rebvio_hip_ctx* c, rebvio_hip_map* om, rebvio_hip_map* nm, const float* R_prior, float frame_dt,
                          rebvio_hip_pair_out* out) {
  HIPCHK(hipSetDevice(c->device));
  std::memset(out, 0, sizeof(*out));
  hipStream_t s = c->s_trk;
  HIPCHK(trk_wait_ready(s, om));
  HIPCHK(trk_wait_ready(s, nm));
  int rc = rebvio_hip_build_distance_field(c, nm);  // rebvio.cpp:142 (no-op when detect already built it)
  if (rc) return rc;
  const hm::M3 R = prior_rotation(c, R_prior);
  float RT[9];
  hm::store3(hm::transpose(R), RT);
  launch_rotate(s, c->K, om->d, RT, c->hist, 0);  // rebvio.cpp:165 (+ histogram for estimateQuantile; hist is zero here)
  const float v0[3] = {0, 0, 0};                  // imu_state_.Vg = Zeros (rebvio.cpp:167)
  // minimizeVel, forwardMatch + extRotVel (rebvio.cpp:169-177); results land in slot 0
  PairSlot* slot = c->slot[0];
  rc = enqueue_pair_lm(c, om, nm, v0, slot, slot->xrv, GlueArgs{});
  if (rc) return rc;
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(s));
  if (slot->seq != c->last_stamp) return fail_msg(kStaleRecordMsg, kStaleRecord);
  nm->n_host = slot->new_st.n;
  nm->thr_host = slot->new_st.threshold;
  const GlueOut g = pair_glue(c, slot->lm, slot->xrv, nm->n_host, frame_dt, R, out);
  enqueue_b_chain(c, om, nm, g, nullptr);
  HIPCHK(hipGetLastError());
  if (g.nan_v) {
    out->status = 1;
    return 0;
  }
  HIPCHK(hipMemcpyAsync(&c->h_st[1], nm->d.st, sizeof(MapState), hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  out->klm_num = c->h_st[1].dm_matches;
  out->kf_matches = c->h_st[1].dm_kf;
  out->reg_num = c->h_st[1].reg_count;
  if ((unsigned)out->klm_num < c->P.global_min_matches_threshold) out->status = 2;  // rebvio.cpp:247-252
  return 0;
}

int rebvio_hip_track_pair_begin(rebvio_hip_ctx* c, rebvio_hip_map* om, rebvio_hip_map* nm, const float* R_prior, float frame_dt,
                                rebvio_hip_pair_mid* mid) {
  HIPCHK(hipSetDevice(c->device));
  const auto tb0 = std::chrono::steady_clock::now();
  std::memset(mid, 0, sizeof(*mid));
  hipStream_t s = c->s_trk;
  if (!c->h_bf) {
    HIPCHK(hipHostMalloc(&c->h_bf, 2 * sizeof(MapState), hipHostMallocDefault));
    for (auto& e : c->bf_done) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  }
  c->bf_cur = (c->bf_res == 0) ? 1 : 0;  // the slot that does not hold an unfetched result
  // The previous pair's match counters (unfetched: result slot bf_res). If this pair continues from its new map they arrive
  // for free with this pair's slot (the first-half kernel copies its old map's state record into pinned memory); otherwise a
  // copy is queued here, AHEAD of this pair (and of its parked second half).
  const int pr = c->bf_res;
  const bool prev_from_slot = pr >= 0 && !c->bf_have[pr] && !c->bf_copy_queued[pr] && c->bf_map[pr] == om;
  if (pr >= 0 && !c->bf_have[pr] && !c->bf_copy_queued[pr] && !prev_from_slot) {
    HIPCHK(hipMemcpyAsync(&c->h_bf[pr], c->bf_map[pr]->d.st, sizeof(MapState), hipMemcpyDeviceToHost, s));
    HIPCHK(hipEventRecord(c->bf_done[pr], s));
    c->bf_copy_queued[pr] = true;
  }
  wait_enqueued(om);
  wait_enqueued(nm);
  HIPCHK(trk_wait_ready_once(c, om));
  HIPCHK(trk_wait_ready_once(c, nm));
  int rc = rebvio_hip_build_distance_field(c, nm);
  if (rc) return rc;
  const hm::M3 R = prior_rotation(c, R_prior);
  if (om->pre_rotated) {
    // the previous pair's _finish_async applied this pair's first rotateKeylines (and binned the sigma histogram) inside its
    // last kernel; it was told the same prior
    if (std::memcmp(&om->pre_R, &R, sizeof(R)) != 0)
      return fail_msg("track_pair_begin: R_prior (or the gyro state) differs from the R_prior_next the previous finish_async applied", -7);
  } else {
    float RT[9];
    hm::store3(hm::transpose(R), RT);
    launch_rotate(s, c->K, om->d, RT, c->hist, 0);
  }
  const float v0[3] = {0, 0, 0};
  PairSlot* slot = c->slot[0];
  rc = enqueue_pair_lm(c, om, nm, v0, slot, slot->xrv, GlueArgs{});
  if (rc) return rc;
  HIPCHK(hipGetLastError());
  const auto tb1 = std::chrono::steady_clock::now();
  // No event behind the first half: a completion event is a packet of its own on the track stream (~5 us until its signal fires
  // on this runtime, plus the wake-up) between the LM kernel's end and this thread - the path the inertial fusion and the second
  // half's launch wait on. The records carry sequence stamps (slot: the LM kernel's last store; every extRotVel block record:
  // stored behind the record's own words), so the thread polls them in pinned memory instead.
  {
    const int prc = poll_pair_records(c, slot, c->last_stamp);
    if (prc) return prc;
  }
  const auto tb2 = std::chrono::steady_clock::now();
  c->t_begin_enq += std::chrono::duration<double, std::micro>(tb1 - tb0).count();
  c->t_begin_wait += std::chrono::duration<double, std::micro>(tb2 - tb1).count();
  c->t_begin_n++;
  if (slot->seq != c->last_stamp) return fail_msg(kStaleRecordMsg, kStaleRecord);
  if (prev_from_slot) {
    c->h_bf[pr] = slot->old_st;
    c->bf_have[pr] = true;
  }
  nm->n_host = slot->new_st.n;
  nm->thr_host = slot->new_st.threshold;
  lm_to_out(slot->lm, mid->Vg, mid->P_Vg, &mid->F, &mid->lm_accept_mask, &mid->sigma_rho_min);
  note_accept_mask(c, mid->lm_accept_mask);
  float JtF6[6];
  sum_xrv(slot->xrv, div_up(nm->n_host, 256), mid->W_Xv, JtF6, nullptr);
  hm::sym6_solve(mid->W_Xv, JtF6, mid->Xv);
  mid->ext_ok = 1;
  for (int i = 0; i < 6; ++i)
    if (std::isnan(mid->Xv[i])) mid->ext_ok = 0;
  std::memcpy(mid->Xgv, mid->Xv, sizeof(mid->Xv));
  std::memcpy(mid->W_Xgv, mid->W_Xv, sizeof(mid->W_Xv));
  const float s_b = c->P.gyro_bias_std_dev * c->P.gyro_bias_std_dev * frame_dt * frame_dt;
  const float s_g = c->P.gyro_std_dev * c->P.gyro_std_dev * frame_dt * frame_dt;
  c->RGBias = hm::diag3(s_b);
  c->RGyro = hm::diag3(s_g);
  float dg[3];
  hm::gyro_bias_correction(mid->Xgv, mid->W_Xgv, c->W_Bg, c->RGyro, c->RGBias, dg);
  for (int i = 0; i < 3; ++i) c->Bg[i] += dg[i];
  hm::store3(R, mid->R);
  return 0;
}

int rebvio_hip_track_pair_finish_async(rebvio_hip_ctx* c, rebvio_hip_map* om, rebvio_hip_map* nm, const float V[3], const float P_V[9],
                                       const float Rgva[9], const float R_second[9], const float* R_prior_next) {
  HIPCHK(hipSetDevice(c->device));
  GlueOut g;
  std::memcpy(g.V, V, sizeof(g.V));
  std::memcpy(g.P_V, P_V, sizeof(g.P_V));
  std::memcpy(g.Rgva, Rgva, sizeof(g.Rgva));
  std::memcpy(g.R0a, R_second, sizeof(g.R0a));
  g.nan_v = std::isnan(V[0]) || std::isnan(V[1]) || std::isnan(V[2]);
  if (!c->h_bf) return fail_msg("track_pair_finish without track_pair_begin", -7);
  if (c->bf_res >= 0) return fail_msg("track_pair_finish_async: fetch the previous pair's result first (rebvio_hip_track_pair_result)", -7);
  c->bf_nan[c->bf_cur] = g.nan_v;
  c->bf_res = c->bf_cur;
  c->bf_map[c->bf_cur] = nm;  // its state record will hold the counters: fetched by the next _begin or by _result
  c->bf_have[c->bf_cur] = false;
  c->bf_copy_queued[c->bf_cur] = false;
  float RT_next[9];
  if (R_prior_next) {  // the next pair's first rotateKeylines rides in this pair's last kernel (rebvio.cpp:163-165 of that pair)
    nm->pre_R = prior_rotation(c, R_prior_next);
    hm::store3(hm::transpose(nm->pre_R), RT_next);
  }
  enqueue_b_chain(c, om, nm, g, R_prior_next ? RT_next : nullptr);
  HIPCHK(hipGetLastError());
  return 0;
}

int rebvio_hip_track_pair_hint_next(rebvio_hip_ctx* c, rebvio_hip_map* next_new_map) {
  if (!c || !next_new_map || next_new_map->ctx != c) return fail_msg("track_pair_hint_next: map of another context", -3);
  HIPCHK(hipSetDevice(c->device));
  wait_enqueued(next_new_map);
  HIPCHK(trk_wait_ready_once(c, next_new_map));
  return 0;
}

int rebvio_hip_track_pair_result(rebvio_hip_ctx* c, int* klm_num, int* kf_matches, int* reg_num, int* status) {
  HIPCHK(hipSetDevice(c->device));
  if (klm_num) *klm_num = 0;
  if (kf_matches) *kf_matches = 0;
  if (reg_num) *reg_num = 0;
  if (c->bf_res < 0) return fail_msg("track_pair_result: no finished pair to report", -7);
  const int r = c->bf_res;
  c->bf_res = -1;
  if (!c->bf_have[r]) {
    if (!c->bf_copy_queued[r]) {  // no later pair has been begun: fetch the record now (nothing is parked at this point)
      HIPCHK(hipMemcpyAsync(&c->h_bf[r], c->bf_map[r]->d.st, sizeof(MapState), hipMemcpyDeviceToHost, c->s_trk));
      HIPCHK(hipEventRecord(c->bf_done[r], c->s_trk));
    }
    HIPCHK(hipEventSynchronize(c->bf_done[r]));  // (a copy queued by _begin sits ahead of that pair's parked second half)
    c->bf_have[r] = true;
  }
  if (c->bf_nan[r]) {
    if (status) *status = 1;
    return 0;
  }
  const MapState& st = c->h_bf[r];
  if (klm_num) *klm_num = st.dm_matches;
  if (kf_matches) *kf_matches = st.dm_kf;
  if (reg_num) *reg_num = st.reg_count;
  if (status) *status = ((unsigned)st.dm_matches < c->P.global_min_matches_threshold) ? 2 : 0;
  return 0;
}

int rebvio_hip_track_pair_finish(rebvio_hip_ctx* c, rebvio_hip_map* om, rebvio_hip_map* nm, const float V[3], const float P_V[9],
                                 const float Rgva[9], const float R_second[9], int* klm_num, int* kf_matches, int* reg_num,
                                 int* status) {
  const int rc = rebvio_hip_track_pair_finish_async(c, om, nm, V, P_V, Rgva, R_second, nullptr);
  if (rc) return rc;
  return rebvio_hip_track_pair_result(c, klm_num, kf_matches, reg_num, status);
}

namespace {
// ---- streaming driver -------------------------------------------------------------------------------------------------
// Per pair, on the track stream, four launches and nothing else (one event per GROUP of pairs, see stream_enqueue_group):
//   persistent minimizeVel / forwardMatch / extRotVel, with the pair's GLUE at its tail: every workgroup publishes its
//             extRotVel sums as tagged words, workgroup 0 collects them and evaluates glue.hpp (6x6 solve, gyroBiasCorrection,
//             SO3, Cholesky) on four waves (glue_dev.hpp); filter state in device memory (double-buffered by pair parity),
//             the second half's inputs in glue_dev[slot], the host's record in a pinned GlueRec
//   directedMatch head (reads glue_dev[slot]) -> directedMatch tail -> regularize / depth EKF / next pair's first rotation
// The host never stands between a pair's halves (round 2: kernel end -> event -> host glue -> flag -> wait kernel, 8-10 us
// of an ~80 us frame, and the reason the rate moved with the box's host): it queues pair k as soon as frame k + lead - 2 has
// been handed to the detect worker and reads the pairs' records up to kSlots - 1 pairs later. A pair's match counters are
// read from the NEXT pair's slot (its first kernel copies its old map's state record), so pair k is reported once pair
// k + 1's event has fired; rebvio_hip_flush() fetches the last pair's counters itself.

int stream_finish_record(rebvio_hip_ctx* c, const rebvio_hip_ctx::InFlight& a, const MapState& st) {
  const GlueRec* r = c->rec[a.slot];
  if (r->seq_out != a.seq || r->seq_gs != a.seq) return fail_msg(kStaleRecordMsg, kStaleRecord);
  rebvio_hip_ctx::Done d;
  d.out = r->out;
  if (d.out.status != 1) {
    d.out.klm_num = st.dm_matches;
    d.out.kf_matches = st.dm_kf;
    d.out.reg_num = st.reg_count;
    if ((unsigned)d.out.klm_num < c->P.global_min_matches_threshold) d.out.status = 2;  // rebvio.cpp:247-252
  }
  d.keylines = st.n;
  // host mirror of the filter state (rebvio_hip_get_gyro_state; authoritative again after a flush)
  for (int i = 0; i < 3; ++i) c->Bg[i] = r->gs.Bg[i];
  c->W_Bg = hm::load3(r->gs.W_Bg);
  c->gs_R = hm::load3(r->gs.R);
  note_accept_mask(c, d.out.lm_accept_mask);
  c->t_queued += (double)st.dm_queued;
  c->done.push_back(d);
  return 0;
}

// Reports the pairs whose successor has completed; `need` > 0: blocks until at least that many have been reported.
int stream_harvest(rebvio_hip_ctx* c, int need) {
  while (c->inflight.size() >= 2) {
    const rebvio_hip_ctx::InFlight& a = c->inflight[0];
    const rebvio_hip_ctx::InFlight& b = c->inflight[1];
    if (need > 0) {
      const auto t0 = std::chrono::steady_clock::now();
      HIPCHK(hipEventSynchronize(c->slot_ev[b.ev_slot]));
      c->t_wait += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    } else {
      const hipError_t q = hipEventQuery(c->slot_ev[b.ev_slot]);
      if (q == hipErrorNotReady) break;
      HIPCHK(q);
    }
    if (*c->lm_bar_err) return fail_msg(("persistent LM kernel: record exchange timed out; " + residency_describe(c->device, c)).c_str(), -9);
    if (c->slot[b.slot]->seq != b.seq) return fail_msg(kStaleRecordMsg, kStaleRecord);  // (pair a's counters ride in pair b's slot)
    const int frc = stream_finish_record(c, a, c->slot[b.slot]->old_st);
    if (frc) return frc;
    c->inflight.pop_front();
    --need;
  }
  return 0;
}

// `npairs` consecutive pairs (frames[0..npairs]) as ONE group on the track stream. Every stream operation between two
// kernels is a packet of its own for the command processor, and with the kernels queued back to back those packets are what
// is left between them (measured with in-kernel stamps: ~6 us per event record / event wait on the pair-to-pair path). A group
// carries two of them whatever its size: one wait for the detection of its NEWEST map (the keyline stream is in order, so the
// maps before it are ready too) and one event behind its last kernel, which stands for every pair's completion and for the
// release of every old map.
int stream_enqueue_group(rebvio_hip_ctx* c, int npairs) {
  // the slot of pair k is reused by pair k + kSlots: its record (and its successor's) must have been read
  while ((int)c->inflight.size() + npairs > rebvio_hip_ctx::kSlots - 1) {
    const int rc = stream_harvest(c, 1);
    if (rc) return rc;
  }
  hipStream_t s = c->s_trk;
  if (!c->frames[0]->trk_waited) {  // first group of a stream
    wait_enqueued(c->frames[0]);
    HIPCHK(trk_wait_ready(s, c->frames[0]));
    c->frames[0]->trk_waited = true;
  }
  {
    rebvio_hip_map* newest = c->frames[(size_t)npairs];
    wait_enqueued(newest);  // (the detect worker launches in order: the maps before it are enqueued too)
    HIPCHK(trk_wait_ready(s, newest));
    for (int g = 1; g <= npairs; ++g) {
      c->frames[(size_t)g]->trk_waited = true;
      if (g < npairs) {  // (as trk_wait_ready does for the map it is given)
        std::lock_guard<std::mutex> dl(c->dl_mu);
        c->frames[(size_t)g]->trk_touched.store(true, std::memory_order_release);
      }
    }
  }
  int last_slot = -1;
  for (int g = 0; g < npairs; ++g) {
    rebvio_hip_map *om = c->frames[(size_t)g], *nm = c->frames[(size_t)g + 1];
    c->df_map = nm;
    const int slot = (int)(c->pair_seq % rebvio_hip_ctx::kSlots);
    const int gpar = (int)(c->pair_seq & 1);
    if (!om->pre_rotated) {
      // first pair of a stream (or after a flush): no second half has applied the prior rotation yet, and the filter state
      // the device works on is the host's
      const hm::M3 R = prior_rotation(c, nullptr);
      GlueState& gs = c->h_gstate[gpar];
      for (int i = 0; i < 3; ++i) gs.Bg[i] = c->Bg[i];
      hm::store3(c->W_Bg, gs.W_Bg);
      hm::store3(R, gs.R);
      gs.pad = 0.f;
      HIPCHK(hipMemcpyAsync(c->gstate + gpar, &gs, sizeof(GlueState), hipMemcpyHostToDevice, s));
      c->wbg_shadow = c->W_Bg;  // what the device's filter state starts from
      c->wbg_shadow_valid = true;
      float RT[9];
      hm::store3(hm::transpose(R), RT);
      launch_rotate(s, c->K, om->d, RT, c->hist, 0);
    }
    const float v0[3] = {0, 0, 0};
    const float frame_dt = (float)((double)(float)(nm->ts - om->ts) / 1000000.0);  // rebvio.cpp:183
    const int calls = (int)c->P.iterations + 1;
    GlueArgs ga;
    ga.lm = c->lm + calls + 1;
    ga.xrv = c->xrv_part;
    ga.st_in = c->gstate + gpar;
    ga.st_out = c->gstate + (gpar ^ 1);
    ga.rec = c->rec[slot];
    ga.gd_copy = c->glue_dev + slot;
    ga.stage = c->glue_stage + slot;
    ga.gp = glue_params(c, frame_dt);
    glue_params_pre(c, &ga.gp);
    int rc = enqueue_pair_lm(c, om, nm, v0, c->slot[slot], c->xrv_part, ga);  // rebvio.cpp:167-177 + the glue of rebvio.cpp:177-233
    if (rc) return rc;
    launch_directed_match_dev(s, c->K, nm->d, om->d, c->glue_dev + slot, c->glue_stage + slot, c->P.search_range, c->dm_head_form);
    const int gate = (int)c->P.global_min_matches_threshold;
    launch_regularize_ekf_dev(s, c->K, nm->d, c->glue_dev + slot, gate > 0 ? gate : 0, c->hist);  // rebvio.cpp:256-259
    std::swap(nm->d.rs, nm->d.rs_tmp);
    std::swap(nm->d.grad, nm->d.grad_tmp);
    nm->pre_rotated = true;
    HIPCHK(hipGetLastError());
    rebvio_hip_ctx::InFlight f;
    f.nm = nm;
    f.seq = c->last_stamp;
    f.slot = slot;
    f.ev_slot = -1;
    f.frame_dt = frame_dt;
    c->inflight.push_back(f);
    last_slot = slot;
    c->pair_seq++;
  }
  HIPCHK(hipEventRecord(c->slot_ev[last_slot], s));
  for (int g = 0; g < npairs; ++g) {
    c->inflight[c->inflight.size() - 1 - (size_t)g].ev_slot = last_slot;
    release_map(c->frames[(size_t)g], c->slot_ev[last_slot]);  // stream-ordered: reusable once the group has drained
  }
  c->frames.erase(c->frames.begin(), c->frames.begin() + npairs);
  return 0;
}

// everything queued completes; every pair's record ends in c->done
int stream_drain(rebvio_hip_ctx* c) {
  int rc = stream_harvest(c, (int)c->inflight.size());
  if (rc) return rc;
  if (c->inflight.size() == 1) {  // the last pair has no successor to carry its counters
    const rebvio_hip_ctx::InFlight a = c->inflight[0];
    HIPCHK(hipEventSynchronize(c->slot_ev[a.ev_slot]));
    if (*c->lm_bar_err) return fail_msg(("persistent LM kernel: record exchange timed out; " + residency_describe(c->device, c)).c_str(), -9);
    HIPCHK(hipMemcpyAsync(&c->h_st[1], a.nm->d.st, sizeof(MapState), hipMemcpyDeviceToHost, c->s_trk));
    HIPCHK(hipStreamSynchronize(c->s_trk));
    if (c->slot[a.slot]->seq != a.seq) return fail_msg(kStaleRecordMsg, kStaleRecord);
    const int frc = stream_finish_record(c, a, c->h_st[1]);
    if (frc) return frc;
    c->inflight.pop_front();
  }
  return 0;
}
}  // namespace

namespace {
int push_frame(rebvio_hip_ctx* c, const uint8_t* frame_dev, const uint8_t* frame_host, size_t host_pitch, uint64_t ts_us, rebvio_hip_pair_out* out,
               int* keylines) {
  // Software pipeline over the three HIP streams of the context:
  //   scan / keyline streams : frame f (this call, through the detect worker)
  //   track stream           : see the comment above stream_wait_maps
  // The returned record is the oldest COMPLETE pair not yet handed out, several frames behind f in steady state; status -1
  // while there is none.
  // Lead: pair (k-1, k) is started once frame k+lead-2 has been queued for detection. A frame's detection takes ~130 us
  // from enqueue to its distance field (scan chain, then keyline chain) while the tracker needs a new map every ~70-80 us: with
  // the minimum lead of 3 the kernel trace showed the LM kernel starting 11-12 us after the previous pair's last kernel,
  // waiting for that map; lead 5 keeps two more detections in flight. Costs latency of the returned records, not throughput;
  // flush() drops the frames no pair was started for.
  rebvio_hip_map* m = nullptr;
  const auto td0 = std::chrono::steady_clock::now();
  HIPCHK(hipSetDevice(c->device));
  if (out) {
    std::memset(out, 0, sizeof(*out));
    out->status = -1;
  }
  if (keylines) *keylines = -1;
  c->min_pool = c->lead + 3 * c->group + 3;
  int rc = detect_async(c, frame_dev, 1, ts_us, &m, frame_host, host_pitch);
  if (rc) return rc;
  {
    std::lock_guard<std::mutex> lk(c->det_mu);  // written by the detect worker
    if (!c->det_error.empty()) return fail_msg(c->det_error.c_str(), -8);
  }
  const auto td1 = std::chrono::steady_clock::now();
  c->t_detect_enq += std::chrono::duration<double, std::micro>(td1 - td0).count();
  c->t_frames++;
  c->frames.push_back(m);
  {
    // Full groups while the device has pairs queued (fewest stream operations per pair); as soon as it is about to run dry - a
    // stream's start, or right after the caller synchronised - whatever can start is started at once, down to single pairs.
    // Either way the newest map of a group was handed to the detect worker at least lead - 2 calls ago.
    const int Q = (int)c->frames.size();
    const bool shallow = (int)c->inflight.size() < c->group;
    const int npairs = shallow ? std::min(c->group, Q - c->lead + 1) : (Q >= c->lead + c->group - 1 ? c->group : 0);
    if (npairs >= 1) {
      rc = stream_enqueue_group(c, npairs);
      if (rc) return rc;
    }
  }
  c->t_enq += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - td1).count();
  rc = stream_harvest(c, 0);
  if (rc) return rc;
  if (!c->done.empty()) {
    if (out) *out = c->done.front().out;
    if (keylines) *keylines = c->done.front().keylines;
    c->done.pop_front();
  }
  return 0;
}

}  // namespace

int rebvio_hip_push_frame_u8_device(rebvio_hip_ctx* c, const uint8_t* frame_dev, uint64_t ts_us, rebvio_hip_pair_out* out,
                                    int* keylines) {
  return push_frame(c, frame_dev, nullptr, 0, ts_us, out, keylines);
}

int rebvio_hip_push_frame_u8(rebvio_hip_ctx* c, const uint8_t* frame_host, size_t pitch_bytes, uint64_t ts_us, rebvio_hip_pair_out* out,
                             int* keylines) {
  if (!frame_host) return fail_msg("push_frame_u8: null frame", -3);
  return push_frame(c, nullptr, frame_host, pitch_bytes, ts_us, out, keylines);
}

uint64_t rebvio_hip_pairs_started(rebvio_hip_ctx* c) { return c->pair_seq; }

// Test hook: the kernels of the NEXT pair this context queues (per-pair API or streaming driver) are handed a wrong sequence
// stamp, so its records look exactly like records read before the pair wrote them; the call that reads them must fail with -12.
int rebvio_hip_test_forge_record_stamp(rebvio_hip_ctx* c) {
  if (!c) return -3;
  c->forge_stamp = true;
  return 0;
}


int rebvio_hip_next_record(rebvio_hip_ctx* c, rebvio_hip_pair_out* out, int* keylines) {
  if (c->done.empty()) return 0;
  if (out) *out = c->done.front().out;
  if (keylines) *keylines = c->done.front().keylines;
  c->done.pop_front();
  return 1;
}

int rebvio_hip_flush(rebvio_hip_ctx* c) {
  HIPCHK(hipSetDevice(c->device));
  if (c->lm_stamps && c->lm_stamp_n && c->lm_stamp_spec) {
    const double* a = c->lm_stamp_acc;
    const double n = (double)c->lm_stamp_n;
    std::fprintf(stderr,
                 "[rebvio_hip] k_lm_chain_spec workgroup 0, mean us over %llu launches: eval0 %.2f  eval1 %.2f  collect+states %.2f  "
                 "speculative evals [project %.2f  issue gathers %.2f  match %.2f  neighbour round %.2f  weighted sums %.2f  publish %.2f]  "
                 "collect all %.2f  check %.2f  finish %.2f  forwardMatch+extRotVel %.2f  collect sums + device glue %.2f\n",
                 (unsigned long long)c->lm_stamp_n, a[2] / n, a[3] / n, a[4] / n, a[5] / n, a[6] / n, a[7] / n, a[8] / n, a[9] / n, a[10] / n,
                 a[11] / n, a[12] / n, a[13] / n, a[14] / n, a[15] / n);
    if (c->lm_stamps[41])
      std::fprintf(stderr, "[rebvio_hip] end of an LM launch -> start of the next (second half of the pair + stream operations), mean over %llu: %.2f us\n",
                   (unsigned long long)c->lm_stamps[41], (double)c->lm_stamps[40] * 0.01 / (double)c->lm_stamps[41]);
    if (c->lm_stamps[55]) {
      const double m = 0.01 / (double)c->lm_stamps[55];
      std::fprintf(stderr, "[rebvio_hip]   device glue: wait for + sum the extRotVel records %.2f  6x6 solve (wave 0) %.2f  until every wave is there %.2f  "
                   "X, SO3, covariance, records %.2f\n", (double)c->lm_stamps[51] * m, (double)c->lm_stamps[52] * m, (double)c->lm_stamps[53] * m,
                   (double)c->lm_stamps[54] * m);
    }
    if (c->lm_stamps[46]) {
      const double m = 0.01 / (double)c->lm_stamps[46];
      std::fprintf(stderr, "[rebvio_hip]   of which: LM end -> head start %.2f  head start -> tail start %.2f  tail start -> regularize/EKF start %.2f  "
                   "regularize/EKF start -> next LM start %.2f (of which the LM kernel's own prologue: keyline loads + sigma quantile %.2f)\n",
                   (double)c->lm_stamps[42] * m, (double)c->lm_stamps[43] * m, (double)c->lm_stamps[44] * m, (double)c->lm_stamps[45] * m,
                   (double)c->lm_stamps[56] * m);
    }
    if (c->dm_stats) {  // per-wave records of the LAST k_directed_match_c launch
      std::vector<unsigned long long> h(c->dm_stats_words);
      (void)hipMemcpy(h.data(), c->dm_stats, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
      const size_t nw = std::min((size_t)h[0], c->dm_stats_words / 16 - 1);
      double sum[14] = {0}, mx[14] = {0}, slow = 0;
      size_t with_open = 0;
      for (size_t w = 0; w < nw; ++w) {
        const unsigned long long* r = &h[16 * (w + 1)];
        double tot = 0;
        for (int i = 0; i < 14; ++i) {
          sum[i] += (double)r[i];
          mx[i] = std::max(mx[i], (double)r[i]);
          if (i < 7) tot += (double)r[i];
        }
        slow = std::max(slow, tot);
        with_open += r[11] ? 1 : 0;
      }
      if (nw) {
        static const char* const seg[7] = {"set-up + head probes issued", "mask loads back + scan", "head candidates tested", "chains walked",
                                           "long probes issued", "mask loads back + scan", "long candidates tested"};
        std::fprintf(stderr, "[rebvio_hip]   directedMatch (compact), last launch, %zu waves: keylines %.0f  t_steps > 4: %.0f  matched in the head %.0f  long "
                     "searches %.0f (max per wave %.0f, waves with any %zu)  head candidates %.0f (max %.0f)  long-search candidates %.0f (max %.0f)\n",
                     nw, sum[8], sum[9], sum[10], sum[11], mx[11], with_open, sum[12], mx[12], sum[13], mx[13]);
        std::fprintf(stderr, "[rebvio_hip]   per wave, us mean / max:");
        for (int i = 0; i < 7; ++i) std::fprintf(stderr, "  %s %.2f / %.2f", seg[i], sum[i] * 0.01 / (double)nw, mx[i] * 0.01);
        std::fprintf(stderr, "  | slowest wave %.2f\n", slow * 0.01);
      }
    }
  } else if (c->lm_stamps && c->lm_stamp_n) {
    const int calls = (int)c->P.iterations + 1;
    std::fprintf(stderr, "[rebvio_hip] k_lm_chain workgroup 0, mean us per segment over %llu launches\n", (unsigned long long)c->lm_stamp_n);
    std::fprintf(stderr, "  prologue %.2f\n", c->lm_stamp_acc[1] / c->lm_stamp_n);
    for (int k = 0; k < calls; ++k) {
      const double* a = c->lm_stamp_acc + 1 + k * 6;
      const double n = (double)c->lm_stamp_n;
      std::fprintf(stderr, "  eval %d: collect %.2f  lm_step %.2f  evaluate %.2f  wave-reduce %.2f  carry+publish %.2f  (loop edge %.2f)\n", k,
                   a[1] / n, a[2] / n, a[3] / n, a[4] / n, a[5] / n, k + 1 < calls ? a[6] / n : 0.0);
    }
    std::fprintf(stderr, "  final collect+lm_step %.2f  forwardMatch+extRotVel %.2f\n", c->lm_stamp_acc[1 + calls * 6] / c->lm_stamp_n,
                 c->lm_stamp_acc[2 + calls * 6] / c->lm_stamp_n);
  }
  if (std::getenv("REBVIO_HIP_DEBUG") && c->t_frames) {
    const double n = (double)c->t_frames;
    std::fprintf(stderr,
                 "[rebvio_hip] host time per frame (us): detect hand-over %.1f  pair enqueue %.1f  waiting for result slots %.1f | long "
                 "directedMatch searches per pair %.0f\n",
                 c->t_detect_enq / n, c->t_enq / n, c->t_wait / n, c->t_queued / n);
    const uint64_t wn = c->t_worker_n.load();
    if (wn)
      std::fprintf(stderr, "[rebvio_hip] detect worker: %.1f us of launches per frame over %llu frames\n", (double)c->t_worker_ns.load() * 1e-3 / (double)wn,
                   (unsigned long long)wn);
    // (the counters start again: a caller that flushes between phases reads each phase on its own)
    c->t_detect_enq = c->t_enq = c->t_wait = c->t_queued = 0;
    c->t_frames = 0;
    c->t_worker_ns.store(0);
    c->t_worker_n.store(0);
  }
  while (c->det_pending.load(std::memory_order_acquire) > 0) std::this_thread::yield();
  int rc = 0;
  while (rc == 0 && c->frames.size() >= 2)  // the pairs no group was started for yet
    rc = stream_enqueue_group(c, std::min(c->group, (int)c->frames.size() - 1));
  if (rc == 0) rc = stream_drain(c);
  for (auto* m : c->frames)
    if (m->in_use) release_map(m, nullptr);
  c->frames.clear();
  // The last second half binned the sigma histogram for a pair that will not come (its next-rotation rides in the last
  // kernel): a stream that continues after the flush must not find those counts under its first pair's (they put the
  // quantile cut below every fresh keyline's sigma and the pair came back with status 1).
  HIPCHK(hipMemsetAsync(c->hist, 0, 128 * sizeof(int), c->s_trk));
  HIPCHK(hipStreamSynchronize(c->s_det));
  HIPCHK(hipStreamSynchronize(c->s_key));
  { const int rc_ts = trk_sync(c); if (rc_ts) return rc_ts; }
  return rc;
}

int rebvio_hip_test_glue(rebvio_hip_ctx* c, const float vel[3], const float JtJ6[6], float F, float sigma_rho_min, int accept_mask,
                         const float* xrv, int n_new, float frame_dt, const float Bg[3], const float W_Bg[9], const float R_prior[9],
                         rebvio_hip_pair_out* out_dev, float* state_dev, float* second_dev, rebvio_hip_pair_out* out_host, float* state_host,
                         float* second_host) {
  HIPCHK(hipSetDevice(c->device));
  if (n_new < 0 || n_new > c->P.keylines_max) return fail_msg("test_glue: n_new out of range", -3);
  static_assert(sizeof(GlueState) == 22 * sizeof(float) && sizeof(GlueDev) == 44 * sizeof(float), "record layouts of the test hook");
  LmState lm;
  std::memset(&lm, 0, sizeof(lm));
  for (int i = 0; i < 3; ++i) lm.vel[i] = vel[i];
  for (int i = 0; i < 6; ++i) lm.JtJ[i] = JtJ6[i];
  lm.F = F;
  lm.sigma_rho_min = sigma_rho_min;
  lm.accept_mask = accept_mask;
  GlueState st;
  for (int i = 0; i < 3; ++i) st.Bg[i] = Bg[i];
  std::memcpy(st.W_Bg, W_Bg, sizeof(st.W_Bg));
  std::memcpy(st.R, R_prior, sizeof(st.R));
  st.pad = 0.f;
  const int nb = div_up(n_new, 256);
  // host form (what rebvio_hip_track_pair runs)
  {
    GlueState sh = st;
    GlueDev gl;
    std::memset(&gl, 0, sizeof(gl));
    std::memset(out_host, 0, sizeof(*out_host));
    hm::pair_glue_core(lm, xrv, n_new, glue_params(c, frame_dt), sh, gl, *out_host);
    std::memcpy(state_host, &sh, sizeof(sh));
    std::memcpy(second_host, &gl, sizeof(gl));
  }
  // device form (what the streaming and batch drivers run), through the stand-alone glue kernel
  { const int rc_ts = trk_sync(c); if (rc_ts) return rc_ts; }
  char* buf = nullptr;
  const size_t xb = (size_t)std::max(nb, 1) * kXrvStride * sizeof(float);
  HIPCHK(hipMalloc(&buf, sizeof(LmState) + sizeof(MapState) + 2 * sizeof(GlueState) + sizeof(GlueDev) + sizeof(GlueRec) + xb));
  struct Free {
    char* p;
    ~Free() { (void)hipFree(p); }
  } guard{buf};
  LmState* d_lm = reinterpret_cast<LmState*>(buf);
  MapState* d_ms = reinterpret_cast<MapState*>(d_lm + 1);
  GlueState* d_st = reinterpret_cast<GlueState*>(d_ms + 1);
  GlueDev* d_gl = reinterpret_cast<GlueDev*>(d_st + 2);
  GlueRec* d_rec = reinterpret_cast<GlueRec*>(d_gl + 1);
  float* d_xrv = reinterpret_cast<float*>(d_rec + 1);
  MapState ms;
  std::memset(&ms, 0, sizeof(ms));
  ms.n = n_new;
  HIPCHK(hipMemcpy(d_lm, &lm, sizeof(lm), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(d_ms, &ms, sizeof(ms), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(d_st, &st, sizeof(st), hipMemcpyHostToDevice));
  if (nb > 0) HIPCHK(hipMemcpy(d_xrv, xrv, (size_t)nb * kXrvStride * sizeof(float), hipMemcpyHostToDevice));
  MapDev fake{};
  fake.st = d_ms;
  GlueArgs ga;
  ga.lm = d_lm;
  ga.xrv = d_xrv;
  ga.st_in = d_st;
  ga.st_out = d_st + 1;
  ga.rec = d_rec;
  ga.gd_copy = d_gl;
  ga.stage = nullptr;
  ga.seq = 1u;
  ga.gp = glue_params(c, frame_dt);
  launch_pair_glue(c->s_trk, fake, ga);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(c->s_trk));
  GlueRec rec;
  GlueState sd;
  GlueDev gd;
  HIPCHK(hipMemcpy(&rec, d_rec, sizeof(rec), hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(&sd, d_st + 1, sizeof(sd), hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(&gd, d_gl, sizeof(gd), hipMemcpyDeviceToHost));
  *out_dev = rec.out;
  std::memcpy(state_dev, &sd, sizeof(sd));
  std::memcpy(second_dev, &gd, sizeof(gd));
  if (std::memcmp(&rec.gs, &sd, sizeof(sd)) != 0) return fail_msg("test_glue: the record's state copy differs from the state written", -5);
  return 0;
}

int rebvio_hip_profile_enable(rebvio_hip_ctx* c, int on) {
  (void)c;
  std::lock_guard<std::mutex> g(g_prof.mu);
  g_prof.on = on != 0;
  g_prof.stride = on > 1 ? on : 1;  // on = N > 1: sample every N-th launch of each kernel
  if (g_prof.on && g_prof.free_events.size() < 256)
    for (int i = 0; i < 512; ++i) {
      hipEvent_t e;
      if (hipEventCreate(&e) == hipSuccess) g_prof.free_events.push_back(e);
    }
  return 0;
}

int rebvio_hip_profile_select(rebvio_hip_ctx* c, const char* only_kernel) {
  (void)c;
  std::lock_guard<std::mutex> g(g_prof.mu);
  g_prof.only = only_kernel ? only_kernel : "";
  return 0;
}

int rebvio_hip_profile_reset(rebvio_hip_ctx* c) {
  (void)c;
  std::lock_guard<std::mutex> g(g_prof.mu);
  g_prof.drain();
  g_prof.acc.clear();
  return 0;
}

int rebvio_hip_profile_read(rebvio_hip_ctx* c, char* names, size_t names_cap, double* avg_us, int* calls, int cap) {
  (void)hipSetDevice(c->device);
  (void)hipDeviceSynchronize();
  std::lock_guard<std::mutex> g(g_prof.mu);
  g_prof.drain();
  std::string all;
  int k = 0;
  for (auto& kv : g_prof.acc) {
    if (k >= cap) break;
    const std::string& nm = g_prof.names[kv.first];
    if (all.size() + nm.size() + 2 > names_cap) break;
    all += nm;
    all += '\n';
    avg_us[k] = kv.second.second ? kv.second.first / kv.second.second : 0.0;
    calls[k] = kv.second.second;
    ++k;
  }
  if (names_cap) {
    std::strncpy(names, all.c_str(), names_cap - 1);
    names[names_cap - 1] = 0;
  }
  return k;
}

int rebvio_hip_device_alloc(rebvio_hip_ctx* c, size_t bytes, void** out) {
  HIPCHK(hipSetDevice(c->device));
  HIPCHK(hipMalloc(out, bytes));
  return 0;
}
int rebvio_hip_device_free(rebvio_hip_ctx* c, void* p) {
  HIPCHK(hipSetDevice(c->device));
  HIPCHK(hipFree(p));
  return 0;
}
int rebvio_hip_device_upload(rebvio_hip_ctx* c, void* dst, const void* src, size_t bytes) {
  HIPCHK(hipSetDevice(c->device));
  HIPCHK(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
  return 0;
}

}  // extern "C"

// ---- batch of camera streams on one GPU (rebvio_hip_batch_*) ----------------------------------------------------------------------------------
// B independent rebvio pipelines (B instances of rebvio::Rebvio, each with its own detector, tracker and maps: rebvio.hpp:91-112)
// advanced in LOCK-STEP by batched launches: every kernel of the per-frame path runs once per step with lane = blockIdx.z.
// Why not B contexts side by side: measured on MI355X (tools/multi_ctx.py), several contexts per process do not overlap -
// with stream priorities two pipelines starve each other's low-priority stages (2 x 3.0k frames/s against 11.8k for one),
// without them they merely share the single-stream rate, and with more hardware queues every pipeline gets slower - while a
// single 640x480 stream keeps under a tenth of the chip busy because each of its kernels is a short latency chain. A batched
// launch costs about the same latency as a single one, so B lanes raise the frame rate almost B-fold until the chip fills.
// Each lane is a full rebvio_hip_ctx (its buffers, maps, glue state) that adopts the batch's three streams; per lane results
// are bit-identical to a stand-alone context fed the same frames (same kernel bodies, same per-lane reduction order).
struct rebvio_hip_batch {
  int B = 0;
  int device = 0;
  rebvio_hip_params P{};
  KParams K{};
  SharedStreams st{};
  std::vector<rebvio_hip_ctx*> lane;
  LaneStatic* ls_dev = nullptr;
  std::vector<LaneStatic> ls_host;  // what ls_dev holds (re-uploaded when a lane's lens model changes)
  bool lens = false;                // every lane has a lens model: the batched front end runs ahead of the scans
  MapDev* maptab_dev = nullptr;
  hipEvent_t ev_scan[kDetPar]{}, ev_flag[kDetPar]{};
  bool ev_flag_used[kDetPar]{};
  static constexpr int kReadyRing = 16;
  hipEvent_t ev_ready[kReadyRing]{};  // keylines + distance fields of a step finished (keyline stream)
  hipEvent_t slot_ev[rebvio_hip_ctx::kSlots]{};
  uint64_t step = 0;
  struct Frame {
    std::vector<rebvio_hip_map*> m;  // one detected map per lane
    uint64_t step;
  };
  std::deque<Frame> frames;
  struct InFlight {  // a step's pairs (one per lane) whose kernels are queued and whose records have not been read yet
    Frame nf;
    uint32_t seq = 0;  // the step's sequence stamp (all lanes)
    int slot = -1;
    int ev_slot = -1;  // the slot whose event stands for this step's completion (its group's last step)
  };
  struct Done {
    std::vector<rebvio_hip_pair_out> out;
    std::vector<int> keylines;
  };
  std::deque<InFlight> inflight;
  std::deque<Done> done;
  uint64_t pair_seq = 0;
  uint32_t stamp_seq = 0;   // sequence stamps of the lanes' records, one value per step (rebvio_hip_ctx::stamp_seq)
  bool forge_stamp = false;
  int lead = 4;
  int group = 2;          // steps queued together (REBVIO_HIP_BATCH_GROUP 1..4, see stream_enqueue_group)
  int lm_lanes_per_launch = 1;  // lanes whose LM workgroups the device holds together (lm_chain_b_max_lanes)
  int lm_capacity_wgs = 0;      // ... and the workgroups that is (shared with the other registered users, see Residency)
  // REBVIO_HIP_DEBUG: host time per step of the detect worker's launches, of the caller's track enqueue and of its waits for
  // result slots (printed by rebvio_hip_batch_flush)
  bool dbg = false;
  std::atomic<uint64_t> t_det_ns{0};
  bool det_worker = true;  // REBVIO_HIP_DETECT_WORKER
  bool fuse_dog = false;   // REBVIO_HIP_BATCH_FUSE_DOG=1
  double t_trk_enq = 0, t_slot_wait = 0;
  int dm_head_form = 0;   // REBVIO_HIP_BATCH_DM_HEAD: 0 by lane count and map size, 1 compact8, 2 compact4, 3 compact1
  bool poisoned = false;  // a step failed half way (some lanes prepared, others not): every later call is refused
  // detect-enqueue worker: launches the detect stage of a step while the caller thread launches the track stage (the
  // reference's data-acquisition thread, rebvio.cpp:28; same split as the single-stream driver)
  struct DetStep {
    LaneDynB dyn;
    int par;
    uint64_t step;
    hipEvent_t reuse_done;  // last consumer of the maps this step reuses (null: fresh maps)
    std::vector<rebvio_hip_map*> maps;
    bool lens;
  };
  std::thread det_thread;
  std::mutex det_mu;
  std::condition_variable det_cv;
  std::deque<DetStep> det_jobs;
  std::atomic<uint64_t> det_done_steps{0};  // steps whose detect stage has been enqueued (their events are recorded)
  bool det_stop = false;
  std::string det_error;
};

namespace {
int batch_upload_map_entry(rebvio_hip_batch* b, int lane, rebvio_hip_map* m) {
  rebvio_hip_ctx* c = b->lane[lane];
  int idx = -1;
  for (size_t i = 0; i < c->pool.size(); ++i)
    if (c->pool[i] == m) idx = (int)i;
  if (idx < 0 || idx >= kLaneMaps) return fail_msg("batch: edge-map pool of a lane outgrew the lane's map table (release maps)", -2);
  m->tab_idx = idx;
  m->canon = m->d;
  HIPCHK(hipMemcpy(b->maptab_dev + (size_t)lane * kLaneMaps + idx, &m->canon, sizeof(MapDev), hipMemcpyHostToDevice));
  HIPCHK(hipStreamSynchronize(nullptr));  // (null-stream work is not ordered against the batch's non-blocking streams: see alloc_map)
  return 0;
}
inline unsigned map_swap_bits(const rebvio_hip_map* m) { return (m->d.rs != m->canon.rs ? 1u : 0u) | (m->d.grad != m->canon.grad ? 2u : 0u); }

// a map of a batch goes back to its lane's pool; only the last lane's release is marked in the stream (the releases of a
// step sit at one point of the track stream, and the detect stage waits for that one event)
void batch_release_map(rebvio_hip_map* m, bool record_done, hipEvent_t done_ref = nullptr) {
  if (!m || !m->in_use) return;
  rebvio_hip_ctx* c = m->ctx;
  wait_enqueued(m);
  std::lock_guard<std::mutex> pool_lk(c->pool_mu);
  m->done_ref = done_ref;
  if (done_ref) {
    m->has_done = true;  // (the step's slot event, already recorded behind the maps' last consumer)
  } else if (record_done) {
    (void)hipEventRecord(m->done, c->s_trk);
    m->has_done = true;
  }
  if (c->df_map == m) c->df_map = nullptr;
  m->release_seq = ++c->release_counter;
  m->in_use = false;
}

int batch_detect_launch(rebvio_hip_batch* b, const rebvio_hip_batch::DetStep& j) {
  const int B = b->B, par = j.par;
  if (b->ev_flag_used[par]) HIPCHK(hipStreamWaitEvent(b->st.s_det, b->ev_flag[par], 0));
  // the fused candidate kernel moves the last box pass from the scan stream to the keyline stream: measured SLOWER for batches
  // (8 lanes 42.4 k -> 39.7 k frames/s, 4 lanes 32.9 k -> 32.2 k), like the single stream's other moves of scan work to the
  // keyline stream - opt-in here (REBVIO_HIP_BATCH_FUSE_DOG=1), the default for one stream
  const bool fuse = b->fuse_dog;
  launch_scale_space_b(b->st.s_det, b->K, 0, B, b->ls_dev, j.dyn, b->lane[0]->widths, j.lens, fuse);
  HIPCHK(hipEventRecord(b->ev_scan[par], b->st.s_det));
  HIPCHK(hipStreamWaitEvent(b->st.s_key, b->ev_scan[par], 0));
  if (j.reuse_done) HIPCHK(hipStreamWaitEvent(b->st.s_key, j.reuse_done, 0));
  const int fw[2] = {b->lane[0]->widths[0][2], b->lane[0]->widths[1][2]};
  launch_keylines_b(b->st.s_key, b->K, B, b->ls_dev, b->maptab_dev, j.dyn, fuse ? fw : nullptr);
  HIPCHK(hipGetLastError());
  HIPCHK(hipEventRecord(b->ev_flag[par], b->st.s_key));
  b->ev_flag_used[par] = true;
  launch_df_build_b(b->st.s_key, b->K, B, b->ls_dev, b->maptab_dev, j.dyn);
  HIPCHK(hipGetLastError());
  HIPCHK(hipEventRecord(b->ev_ready[j.step % rebvio_hip_batch::kReadyRing], b->st.s_key));
  for (int l = 0; l < B; ++l) {
    // The single-map entries (size, download, ...) wait on the map's own event. Letting the step's one event stand for every
    // lane's map (eight marker packets fewer on the keyline stream) was measured SLOWER, 42.5 k -> 41.9 k frames/s at 8 lanes,
    // three alternating runs: the packets space the keyline stream's kernels apart, to the benefit of the other two streams.
    HIPCHK(hipEventRecord(j.maps[l]->ready, b->st.s_key));
    j.maps[l]->enqueued.store(1, std::memory_order_release);
  }
  return 0;
}

void batch_det_worker(rebvio_hip_batch* b) {
  (void)hipSetDevice(b->device);
  for (;;) {
    rebvio_hip_batch::DetStep j;
    {
      std::unique_lock<std::mutex> lk(b->det_mu);
      b->det_cv.wait(lk, [&] { return b->det_stop || !b->det_jobs.empty(); });
      if (b->det_jobs.empty()) return;
      j = b->det_jobs.front();
      b->det_jobs.pop_front();
    }
    const auto t0 = std::chrono::steady_clock::now();
    const int drc = batch_detect_launch(b, j);
    if (b->dbg) b->t_det_ns.fetch_add((uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count());
    if (drc != 0) {
      std::lock_guard<std::mutex> lk(b->det_mu);
      b->det_error = g_err;
      for (auto* m : j.maps) m->enqueued.store(1, std::memory_order_release);
    }
    b->det_done_steps.store(j.step + 1, std::memory_order_release);
  }
}

int batch_finish_records(rebvio_hip_batch* b, const rebvio_hip_batch::InFlight& a, const MapState* st_of_lane /*[B]*/) {
  for (int l = 0; l < b->B; ++l) {
    const GlueRec* r = b->lane[l]->rec[a.slot];
    if (r->seq_out != a.seq || r->seq_gs != a.seq) return fail_msg(kStaleRecordMsg, kStaleRecord);
  }
  rebvio_hip_batch::Done d;
  d.out.resize((size_t)b->B);
  d.keylines.resize((size_t)b->B);
  for (int l = 0; l < b->B; ++l) {
    rebvio_hip_ctx* c = b->lane[l];
    const GlueRec* r = c->rec[a.slot];
    const MapState& st = st_of_lane[l];
    rebvio_hip_pair_out o = r->out;
    if (o.status != 1) {
      o.klm_num = st.dm_matches;
      o.kf_matches = st.dm_kf;
      o.reg_num = st.reg_count;
      if ((unsigned)o.klm_num < c->P.global_min_matches_threshold) o.status = 2;
    }
    d.out[(size_t)l] = o;
    d.keylines[(size_t)l] = st.n;
    for (int i = 0; i < 3; ++i) c->Bg[i] = r->gs.Bg[i];
    c->W_Bg = hm::load3(r->gs.W_Bg);
    c->gs_R = hm::load3(r->gs.R);
    note_accept_mask(c, o.lm_accept_mask);
  }
  b->done.push_back(std::move(d));
  return 0;
}

// as stream_harvest, for all lanes of a step at once (one event per step)
int batch_harvest(rebvio_hip_batch* b, int need) {
  std::vector<MapState> st((size_t)b->B);
  while (b->inflight.size() >= 2) {
    const rebvio_hip_batch::InFlight& a = b->inflight[0];
    const rebvio_hip_batch::InFlight& n = b->inflight[1];
    if (need > 0) {
      const auto t0 = std::chrono::steady_clock::now();
      HIPCHK(hipEventSynchronize(b->slot_ev[n.ev_slot]));
      b->t_slot_wait += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    } else {
      const hipError_t q = hipEventQuery(b->slot_ev[n.ev_slot]);
      if (q == hipErrorNotReady) break;
      HIPCHK(q);
    }
    for (int l = 0; l < b->B; ++l) {
      if (*b->lane[l]->lm_bar_err) return fail_msg(("batch: persistent LM kernel: record exchange timed out; " + residency_describe(b->device, b)).c_str(), -9);
      if (b->lane[l]->slot[n.slot]->seq != n.seq) return fail_msg(kStaleRecordMsg, kStaleRecord);  // (step a's counters ride in step n's slots)
      st[(size_t)l] = b->lane[l]->slot[n.slot]->old_st;
    }
    const int frc = batch_finish_records(b, a, st.data());
    if (frc) return frc;
    b->inflight.pop_front();
    --need;
  }
  return 0;
}

// `nsteps` consecutive steps' pairs, all lanes, as one group on the track stream (see stream_enqueue_group): per step [rotate +]
// LM kernel with every lane's glue at its end, directedMatch head, tail, regularize / EKF / next rotation; per group one wait
// for the newest step's detection and one event
int batch_enqueue_group(rebvio_hip_batch* b, int nsteps) {
  while ((int)b->inflight.size() + nsteps > rebvio_hip_ctx::kSlots - 1) {
    const int rc = batch_harvest(b, 1);
    if (rc) return rc;
  }
  hipStream_t s = b->st.s_trk;
  {
    const uint64_t newest = b->frames[(size_t)nsteps].step;
    while (b->det_done_steps.load(std::memory_order_acquire) <= newest) std::this_thread::yield();  // its event has been recorded
    HIPCHK(hipStreamWaitEvent(s, b->ev_ready[newest % rebvio_hip_batch::kReadyRing], 0));  // (earlier steps: same stream order)
  }
  const int calls = (int)b->P.iterations + 1;
  int last_slot = -1;
  std::vector<rebvio_hip_batch::Frame> olds;
  for (int g = 0; g < nsteps; ++g) {
    const rebvio_hip_batch::Frame of = b->frames[0], nf = b->frames[1];
    const int slot = (int)(b->pair_seq % rebvio_hip_ctx::kSlots);
    const int gpar = (int)(b->pair_seq & 1);
    const uint32_t step_seq = next_stamp(&b->stamp_seq);
    LaneDynB dyn{};
    for (int l = 0; l < b->B; ++l) {
      rebvio_hip_ctx* c = b->lane[l];
      rebvio_hip_map *om = of.m[l], *nm = nf.m[l];
      c->df_map = nm;
      dyn.v[l].seq = (b->forge_stamp && l == b->B - 1) ? (step_seq ^ 0x40000000u) : step_seq;
      if (!om->pre_rotated) {  // first pair of the lane: no second half has applied the prior rotation yet; the host's state goes up
        const hm::M3 R = prior_rotation(c, nullptr);
        GlueState& gs = c->h_gstate[gpar];
        for (int i = 0; i < 3; ++i) gs.Bg[i] = c->Bg[i];
        hm::store3(c->W_Bg, gs.W_Bg);
        hm::store3(R, gs.R);
        gs.pad = 0.f;
        HIPCHK(hipMemcpyAsync(c->gstate + gpar, &gs, sizeof(GlueState), hipMemcpyHostToDevice, s));
        float RT[9];
        hm::store3(hm::transpose(R), RT);
        launch_rotate(s, c->K, om->d, RT, c->hist, 0);
      }
      LaneDyn& d = dyn.v[l];
      d.nm = (short)nm->tab_idx;
      d.om = (short)om->tab_idx;
      d.nm_swap = (unsigned char)map_swap_bits(nm);
      d.om_swap = (unsigned char)map_swap_bits(om);
      d.slot = (unsigned char)slot;
      d.gpar = (unsigned char)gpar;
      d.tag_base = c->lm_tag_base;
      c->lm_tag_base += 2u * ((unsigned)calls + 1u);
      if (c->lm_tag_base > 0xFFFFFF00u) {
        (void)hipMemsetAsync(c->lm_xch, 0, lm_xch_words(c->maxblocks) * sizeof(unsigned long long), s);
        c->lm_tag_base = 0;
      }
    }
    // one launch for all lanes: the most cautious of the lanes' choices (1 sequential; otherwise the LATEST first speculative
    // evaluation any lane asks for)
    int spec_now = 2;
    for (auto* c : b->lane) {
      const int ch = lm_kernel_choice(c);
      if (ch == 1 || spec_now == 1)
        spec_now = 1;
      else
        spec_now = std::max(spec_now, ch);
    }
    const float frame_dt = (float)((double)(float)(nf.m[0]->ts - of.m[0]->ts) / 1000000.0);  // rebvio.cpp:183
    // this batch's share of the device: the capacity split evenly among the live users of the persistent kernels on this GPU
    const int users = std::max(1, residency_users(b->device));
    const int per_lane_wgs = (b->K.kmax + 511) / 512;
    const int lanes_now = users <= 1 ? b->lm_lanes_per_launch : std::max(1, std::min(b->lm_lanes_per_launch, b->lm_capacity_wgs / users / per_lane_wgs));
    launch_lm_chain_b(s, b->K, b->B, lanes_now, b->ls_dev, b->maptab_dev, dyn, calls, spec_now, glue_params(b->lane[0], frame_dt));
    const int gate = (int)b->P.global_min_matches_threshold;
    launch_b_chain_b(s, b->K, b->B, b->ls_dev, b->maptab_dev, dyn, b->P.search_range, gate > 0 ? gate : 0, b->dm_head_form);
    HIPCHK(hipGetLastError());
    for (int l = 0; l < b->B; ++l) {  // (a lane whose pair is skipped for a NaN velocity still gets its next rotation applied)
      rebvio_hip_map* nm = nf.m[l];
      std::swap(nm->d.rs, nm->d.rs_tmp);
      std::swap(nm->d.grad, nm->d.grad_tmp);
      nm->pre_rotated = true;
    }
    b->forge_stamp = false;
    rebvio_hip_batch::InFlight f;
    f.nf = nf;
    f.seq = step_seq;
    f.slot = slot;
    f.ev_slot = -1;
    b->inflight.push_back(f);
    olds.push_back(of);
    b->frames.pop_front();
    b->pair_seq++;
    last_slot = slot;
  }
  HIPCHK(hipEventRecord(b->slot_ev[last_slot], s));
  for (int g = 0; g < nsteps; ++g) b->inflight[b->inflight.size() - 1 - (size_t)g].ev_slot = last_slot;
  for (auto& of : olds)
    for (auto* m : of.m) batch_release_map(m, false, b->slot_ev[last_slot]);  // the group's one event covers every old map
  return 0;
}

int batch_drain(rebvio_hip_batch* b) {
  int rc = batch_harvest(b, (int)b->inflight.size());
  if (rc) return rc;
  if (b->inflight.size() == 1) {  // the last step has no successor to carry its counters
    const rebvio_hip_batch::InFlight a = b->inflight[0];
    HIPCHK(hipEventSynchronize(b->slot_ev[a.ev_slot]));
    std::vector<MapState> st((size_t)b->B);
    for (int l = 0; l < b->B; ++l) {
      if (*b->lane[l]->lm_bar_err) return fail_msg(("batch: persistent LM kernel: record exchange timed out; " + residency_describe(b->device, b)).c_str(), -9);
      if (b->lane[l]->slot[a.slot]->seq != a.seq) return fail_msg(kStaleRecordMsg, kStaleRecord);
      HIPCHK(hipMemcpy(&st[(size_t)l], a.nf.m[l]->d.st, sizeof(MapState), hipMemcpyDeviceToHost));
    }
    const int frc = batch_finish_records(b, a, st.data());
    if (frc) return frc;
    b->inflight.pop_front();
  }
  return 0;
}

int batch_pop(rebvio_hip_batch* b, rebvio_hip_pair_out* out, int* keylines) {
  if (b->done.empty()) return 0;
  const rebvio_hip_batch::Done& d = b->done.front();
  for (int l = 0; l < b->B; ++l) {
    if (out) out[l] = d.out[(size_t)l];
    if (keylines) keylines[l] = d.keylines[(size_t)l];
  }
  b->done.pop_front();
  return 1;
}
}  // namespace

extern "C" {

void rebvio_hip_batch_destroy(rebvio_hip_batch* b) {
  if (!b) return;
  residency_remove(b->device, b);
  (void)hipSetDevice(b->device);
  if (b->det_thread.joinable()) {
    {
      std::lock_guard<std::mutex> lk(b->det_mu);
      b->det_stop = true;
    }
    b->det_cv.notify_all();
    b->det_thread.join();
  }
  (void)hipDeviceSynchronize();
  // the maps of queued steps and of the pairs in flight are the batch's own, not handles a caller holds
  for (auto& f : b->frames)
    for (auto* m : f.m) m->in_use = false;
  for (auto& f : b->inflight)
    for (auto* m : f.nf.m) m->in_use = false;
  for (auto* c : b->lane) rebvio_hip_destroy(c);
  if (b->ls_dev) (void)hipFree(b->ls_dev);
  if (b->maptab_dev) (void)hipFree(b->maptab_dev);
  for (int i = 0; i < kDetPar; ++i) {
    if (b->ev_scan[i]) (void)hipEventDestroy(b->ev_scan[i]);
    if (b->ev_flag[i]) (void)hipEventDestroy(b->ev_flag[i]);
  }
  for (auto& e : b->ev_ready)
    if (e) (void)hipEventDestroy(e);
  for (auto& e : b->slot_ev)
    if (e) (void)hipEventDestroy(e);
  if (b->st.s_det) (void)hipStreamDestroy(b->st.s_det);
  if (b->st.s_key) (void)hipStreamDestroy(b->st.s_key);
  if (b->st.s_trk) (void)hipStreamDestroy(b->st.s_trk);
  delete b;
}

int rebvio_hip_batch_create(const rebvio_hip_params* p, int lanes, rebvio_hip_batch** out) {
  *out = nullptr;
  if (lanes < 1 || lanes > kMaxLanes) return fail_msg("batch: lanes must be in 1..16", -3);
  int ndev = 0;
  HIPCHK(hipGetDeviceCount(&ndev));
  if (ndev <= 0) return fail_msg("no HIP device present: the gfx950 backend has no CPU fallback", -4);
  if (p->device_id < 0 || p->device_id >= ndev) return fail_msg("device_id out of range", -3);
  HIPCHK(hipSetDevice(p->device_id));
  int lm_lanes_per_launch = 1;
  {
    // The persistent LM kernel exchanges records among the workgroups of a lane, and every workgroup of a launch has to be
    // resident for that (no assumption about dispatch order): the batch driver launches it for as many lanes at a time as the
    // device holds together (lm_chain_b_max_lanes; 8 lanes of 16 k keylines on an MI355X). Not even one lane: refused here
    // instead of finding out from an exchange time-out (-9).
    lm_lanes_per_launch = lm_chain_b_max_lanes(p->device_id, p->keylines_max, (int)p->iterations + 1);
    if (lm_lanes_per_launch < 1) {
      char msg[200];
      std::snprintf(msg, sizeof(msg), "batch: the %d LM workgroups of one lane (%d keylines) do not fit this device together",
                    (p->keylines_max + 511) / 512, p->keylines_max);
      return fail_msg(msg, -3);
    }
  }
  rebvio_hip_batch* b = new rebvio_hip_batch;
  b->lm_lanes_per_launch = lm_lanes_per_launch;
  b->lm_capacity_wgs = lm_chain_b_capacity_wgs(p->device_id, p->keylines_max, (int)p->iterations + 1);
  b->dbg = std::getenv("REBVIO_HIP_DEBUG") != nullptr;
  if (const char* e = std::getenv("REBVIO_HIP_DETECT_WORKER")) b->det_worker = e[0] != '0';
  if (const char* e = std::getenv("REBVIO_HIP_BATCH_FUSE_DOG")) b->fuse_dog = e[0] == '1';
  struct Guard {
    rebvio_hip_batch* b;
    ~Guard() {
      if (b) rebvio_hip_batch_destroy(b);
    }
  } guard{b};
  b->B = lanes;
  b->device = p->device_id;
  residency_add(b->device, b, std::min(lanes, lm_lanes_per_launch) * ((p->keylines_max + 511) / 512));
  b->P = *p;
  // the three stages of a step overlap across steps like the stages of one stream do; one priority class for all
  // (see the comment on rebvio_hip_batch: priorities are what made pipelines starve each other)
  HIPCHK(hipStreamCreateWithFlags(&b->st.s_det, hipStreamNonBlocking));
  HIPCHK(hipStreamCreateWithFlags(&b->st.s_key, hipStreamNonBlocking));
  HIPCHK(hipStreamCreateWithFlags(&b->st.s_trk, hipStreamNonBlocking));
  for (int l = 0; l < lanes; ++l) {
    rebvio_hip_ctx* c = nullptr;
    rebvio_hip_params pl = *p;
    if (pl.map_pool <= 0) pl.map_pool = 8;
    t_adopt_streams = &b->st;
    const int rc = rebvio_hip_create(&pl, &c);
    t_adopt_streams = nullptr;
    if (rc) return rc;
    b->lane.push_back(c);
  }
  b->K = b->lane[0]->K;
  std::vector<LaneStatic> ls((size_t)lanes);
  HIPCHK(hipMalloc(&b->maptab_dev, (size_t)lanes * kLaneMaps * sizeof(MapDev)));
  HIPCHK(hipMemset(b->maptab_dev, 0, (size_t)lanes * kLaneMaps * sizeof(MapDev)));
  for (int l = 0; l < lanes; ++l) {
    rebvio_hip_ctx* c = b->lane[l];
    LaneStatic& L = ls[l];
    for (int f = 0; f < 2; ++f) {
      L.sa[f] = c->sb.a[f];
      L.sb[f] = c->sb.b[f];
    }
    for (int i = 0; i < kDetPar; ++i) {
      L.dog2[i] = c->dog2[i];
      L.mag2[i] = c->mag2[i];
      L.rowcount2[i] = c->rowcount2[i];
      L.undist_img[i] = c->undist_img[i];
    }
    L.stash = c->db.stash;
    L.bits = c->db.bits;
    L.det = c->det;
    L.lm = c->lm;
    L.lm_zero = c->lm_zero;
    L.lm_xch = c->lm_xch;
    L.lm_bar_err = c->lm_bar_err;
    L.hist = c->hist;
    for (int i = 0; i < rebvio_hip_ctx::kSlots; ++i) {
      L.slot[i] = c->slot[i];
      L.rec[i] = c->rec[i];
      L.glue_stage = c->glue_stage;
    }
    L.glue_dev = c->glue_dev;
    L.gstate = c->gstate;
    L.xrv_part = c->xrv_part;
    L.undist_map = c->undist_map;
    for (auto* m : c->pool) {
      const int rc = batch_upload_map_entry(b, l, m);
      if (rc) return rc;
    }
  }
  HIPCHK(hipMalloc(&b->ls_dev, ls.size() * sizeof(LaneStatic)));
  HIPCHK(hipMemcpy(b->ls_dev, ls.data(), ls.size() * sizeof(LaneStatic), hipMemcpyHostToDevice));
  b->ls_host = ls;
  for (int i = 0; i < kDetPar; ++i) {
    HIPCHK(hipEventCreateWithFlags(&b->ev_scan[i], hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&b->ev_flag[i], hipEventDisableTiming));
  }
  for (auto& e : b->ev_ready) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  for (auto& e : b->slot_ev) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  if (const char* e = std::getenv("REBVIO_HIP_BATCH_LEAD")) b->lead = std::min(8, std::max(3, std::atoi(e)));
  if (const char* e = std::getenv("REBVIO_HIP_BATCH_GROUP")) b->group = std::min(4, std::max(1, std::atoi(e)));
  if (const char* e = std::getenv("REBVIO_HIP_BATCH_DM_HEAD"))
    b->dm_head_form = dm_form_by_name(e);
  HIPCHK(hipDeviceSynchronize());
  guard.b = nullptr;
  *out = b;
  return 0;
}

int rebvio_hip_batch_lanes(rebvio_hip_batch* b) { return b->B; }

// as rebvio_hip_test_forge_record_stamp, for the last lane of the next step
int rebvio_hip_batch_test_forge_record_stamp(rebvio_hip_batch* b) {
  if (!b) return -3;
  b->forge_stamp = true;
  return 0;
}
rebvio_hip_ctx* rebvio_hip_batch_lane(rebvio_hip_batch* b, int lane) { return (lane >= 0 && lane < b->B) ? b->lane[lane] : nullptr; }

int rebvio_hip_batch_push_u8_device(rebvio_hip_batch* b, const uint8_t* const* frames_dev, uint64_t ts_us, rebvio_hip_pair_out* out,
                                    int* keylines) {
  HIPCHK(hipSetDevice(b->device));
  if (b->poisoned) return fail_msg("batch: an earlier step failed half way; the lanes are out of lock-step (destroy the batch)", -11);
  const int B = b->B;
  for (int l = 0; l < B; ++l) {
    if (out) {
      std::memset(&out[l], 0, sizeof(out[l]));
      out[l].status = -1;
    }
    if (keylines) keylines[l] = -1;
  }
  {
    std::lock_guard<std::mutex> lk(b->det_mu);
    if (!b->det_error.empty()) return fail_msg(b->det_error.c_str(), -8);
  }
  {
    // lens models (rebvio_hip_set_undistort on a lane's context, camera.hpp:39-40,54-58): all lanes or none; a change is
    // picked up here (set_undistort has synchronised the device)
    int with = 0;
    bool changed = false;
    for (int l = 0; l < B; ++l) {
      rebvio_hip_ctx* c = b->lane[l];
      with += c->undist_map ? 1 : 0;
      LaneStatic& L = b->ls_host[(size_t)l];
      if (L.undist_map != c->undist_map || std::memcmp(L.undist_img, c->undist_img, sizeof(L.undist_img)) != 0) {
        L.undist_map = c->undist_map;
        for (int i = 0; i < kDetPar; ++i) L.undist_img[i] = c->undist_img[i];
        changed = true;
      }
    }
    if (with != 0 && with != B) return fail_msg("batch: set the lens model on every lane or on none", -3);
    if (changed) {
      while (b->det_done_steps.load(std::memory_order_acquire) < b->step) std::this_thread::yield();
      HIPCHK(hipStreamSynchronize(b->st.s_det));
      HIPCHK(hipMemcpy(b->ls_dev, b->ls_host.data(), b->ls_host.size() * sizeof(LaneStatic), hipMemcpyHostToDevice));
      HIPCHK(hipStreamSynchronize(nullptr));
    }
    b->lens = with == B;
  }
  // ---- detect stage of this step (detect_launch for every lane at once) ----
  rebvio_hip_batch::Frame fr;
  fr.step = b->step;
  fr.m.resize((size_t)B);
  LaneDynB dyn{};
  const int par = (int)(b->step % kDetPar);
  rebvio_hip_map* last_reused = nullptr;
  for (int l = 0; l < B; ++l) {
    rebvio_hip_ctx* c = b->lane[l];
    rebvio_hip_ctx::DetJob job;
    c->min_pool = std::min(kLaneMaps - 2, b->lead + 3 * b->group + 3);
    int rc = detect_prepare(c, frames_dev[l], 1, ts_us, &job);
    if (rc == 0) {
      rebvio_hip_map* m = job.m;
      if (m->tab_idx < 0 || std::memcmp(&m->canon.pos, &m->d.pos, sizeof(void*)) != 0)  // a map the pool has just grown by
        rc = batch_upload_map_entry(b, l, m);
    }
    if (rc) {
      // lanes 0 .. l-1 have taken maps and advanced their servo rings for a step that will not run: the batch cannot
      // continue in lock-step
      if (l > 0) b->poisoned = true;
      return rc;
    }
    rebvio_hip_map* m = job.m;
    // a pooled map comes back with whatever ping-pong state its last pair left: the detector writes the canonical arrays
    m->d = m->canon;
    fr.m[l] = m;
    if (l == B - 1) last_reused = m;
    LaneDyn& d = dyn.v[l];
    d.img = frames_dev[l];
    d.nm = (short)m->tab_idx;
    d.om = -1;
    d.prev = -1;
    if (job.prev_st)
      for (auto* pm : c->pool)
        if (pm->d.st == job.prev_st) d.prev = (short)pm->tab_idx;
    d.parity = (unsigned char)par;
    d.det_in = (unsigned char)(job.det_in - c->det);
    d.det_out = (unsigned char)(job.det_out - c->det);
  }
  rebvio_hip_batch::DetStep job;
  job.dyn = dyn;
  job.par = par;
  job.step = b->step;
  // maps are released in lane order at one point of the track stream: the last lane's event covers all of them
  job.reuse_done = (last_reused && last_reused->has_done) ? (last_reused->done_ref ? last_reused->done_ref : last_reused->done) : nullptr;
  job.maps = fr.m;
  job.lens = b->lens;
  for (auto* m : fr.m) m->enqueued.store(0, std::memory_order_relaxed);
  if (!b->det_worker) {  // the caller launches the step's detect kernels itself (see detect_async)
    const auto t0 = std::chrono::steady_clock::now();
    const int drc = batch_detect_launch(b, job);
    if (b->dbg) b->t_det_ns.fetch_add((uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count());
    for (auto* m : fr.m) m->enqueued.store(1, std::memory_order_release);
    b->det_done_steps.store(job.step + 1, std::memory_order_release);
    if (drc) {
      b->poisoned = true;
      return drc;
    }
  } else {
    if (!b->det_thread.joinable()) b->det_thread = std::thread(batch_det_worker, b);
    {
      std::lock_guard<std::mutex> lk(b->det_mu);
      b->det_jobs.push_back(job);
    }
    b->det_cv.notify_one();
  }
  b->frames.push_back(fr);
  b->step++;

  // ---- track stage: the step's pairs, whole, for every lane; then whatever records have become complete ----
  {
    const int Q = (int)b->frames.size();  // (group size by queue depth: see push_frame)
    const bool shallow = (int)b->inflight.size() < b->group;
    const int nsteps = shallow ? std::min(b->group, Q - b->lead + 1) : (Q >= b->lead + b->group - 1 ? b->group : 0);
    if (nsteps >= 1) {
      const auto t0 = std::chrono::steady_clock::now();
      const double w0 = b->t_slot_wait;
      const int rc = batch_enqueue_group(b, nsteps);
      b->t_trk_enq += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() - (b->t_slot_wait - w0);
      if (rc) {
        b->poisoned = true;
        return rc;
      }
    }
  }
  int rc = batch_harvest(b, 0);
  if (rc) return rc;
  (void)batch_pop(b, out, keylines);
  return 0;
}

int rebvio_hip_batch_next_records(rebvio_hip_batch* b, rebvio_hip_pair_out* out, int* keylines) { return batch_pop(b, out, keylines); }

int rebvio_hip_batch_flush(rebvio_hip_batch* b) {
  HIPCHK(hipSetDevice(b->device));
  while (b->det_done_steps.load(std::memory_order_acquire) < b->step) std::this_thread::yield();
  if (b->dbg && b->step)
    std::fprintf(stderr, "[rebvio_hip] batch of %d lanes, host time per step (us): detect worker's launches %.1f  track enqueue %.1f  waiting "
                 "for result slots %.1f\n", b->B, (double)b->t_det_ns.load() / 1e3 / (double)b->step, b->t_trk_enq / (double)b->step,
                 b->t_slot_wait / (double)b->step);
  if (b->dbg)
    std::fprintf(stderr, "[rebvio_hip] batch: pool of lane 0 holds %d maps (min_pool %d), %d steps in flight, %d detected steps queued\n",
                 (int)b->lane[0]->pool.size(), b->lane[0]->min_pool, (int)b->inflight.size(), (int)b->frames.size());
  int rc = 0;
  while (rc == 0 && b->frames.size() >= 2)  // the steps no group was started for yet
    rc = batch_enqueue_group(b, std::min(b->group, (int)b->frames.size() - 1));
  if (rc == 0) rc = batch_drain(b);
  for (auto& f : b->frames)
    for (size_t l = 0; l < f.m.size(); ++l) batch_release_map(f.m[l], l + 1 == f.m.size());
  b->frames.clear();
  for (auto* c : b->lane)  // as rebvio_hip_flush: no histogram counts of a pair that will not come
    HIPCHK(hipMemsetAsync(c->hist, 0, 128 * sizeof(int), b->st.s_trk));
  HIPCHK(hipStreamSynchronize(b->st.s_det));
  HIPCHK(hipStreamSynchronize(b->st.s_key));
  HIPCHK(hipStreamSynchronize(b->st.s_trk));
  return rc;
}

}  // extern "C"
