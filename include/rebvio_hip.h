/*
 * rebvio_hip.h — C-ABI of the MI355X (gfx950) backend for rebvio's per-frame edge-detection +
 * edge-tracking hot path. Plain C types, caller-allocated outputs, int status (0 = ok, <0 = error,
 * see rebvio_hip_last_error; the codes are listed in INTEGRATION.md "Status codes"). No exceptions cross this boundary. One context per camera stream and
 * GPU; a context is not thread-safe, but its detect-side entries (detect*) and track-side entries
 * run on two private HIP streams and overlap on the device, mirroring the reference's two workers
 * (rebvio/src/rebvio.cpp:28-29).
 *
 * Each entry names the reference interface it replaces (paths relative to the reference repo).
 * The C++ classes in include/rebvio/ (EdgeDetector, EdgeMap, Core, Rebvio) are thin hosts over this
 * ABI; INTEGRATION.md shows the binding.
 */
#ifndef REBVIO_HIP_H_
#define REBVIO_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 3: status -12 (a pair's record read before the device wrote it: sequence stamps), rebvio_hip_test_forge_record_stamp,
 *    REBVIO_HIP_DM_HEAD values compact8 / compact4 / compact1 (2: map handles outlive their context, -10) */
#define REBVIO_HIP_ABI_VERSION 3

/* Host mirror of one keyline: field-for-field rebvio::types::KeyLine
 * (rebvio/include/rebvio/types/keyline.hpp:24-40), 84 bytes. Device storage is SoA. */
typedef struct rebvio_hip_keyline {
  float pos[2];
  float pos_img[2];
  float match_pos_img[2];
  float gradient[2];
  float match_gradient[2];
  float gradient_norm;
  float match_gradient_norm;
  float rho;
  float sigma_rho;
  int id;
  int id_prev;
  int id_next;
  int match_id;
  int match_id_forward;
  int match_id_keyframe;
  unsigned int matches;
} rebvio_hip_keyline;

/* Camera constants (camera.hpp:61-72), EdgeDetectorConfig (edge_detector.hpp:19-32), CoreConfig
 * (core.hpp:82-95), EdgeMapConfig (edge_map.hpp:19-26), gyro noise of ImuStateConfig
 * (types/imu.hpp:158-159). */
typedef struct rebvio_hip_params {
  int rows, cols;
  float fm, cx, cy;
  int keylines_ref, keylines_max;
  float pos_neg_threshold, dog_threshold, threshold, gain, max_threshold, min_threshold;
  float search_range, reweight_distance, match_treshold;
  unsigned int min_match_threshold, iterations, global_min_matches_threshold;
  float pixel_uncertainty, quantile_cutoff;
  int quantile_num_bins;
  float reshape_q_abs;
  float pixel_uncertainty_match, match_threshold_norm, match_threshold_angle, regularization_threshold;
  float gyro_std_dev, gyro_bias_std_dev;
  int device_id;  /* HIP device ordinal for this camera stream */
  int map_pool;   /* edge maps kept alive at once (>= 3: old, new, in-flight detect); 0 = default */
} rebvio_hip_params;

typedef struct rebvio_hip_ctx rebvio_hip_ctx;
typedef struct rebvio_hip_map rebvio_hip_map;

/* Result of one frame-pair step (rebvio.cpp:142-259 without accelerometer/SAB fusion). */
typedef struct rebvio_hip_pair_out {
  float Vg[3];
  float P_Vg[9];
  float F;
  float Xv[6];
  float W_Xv[36];
  float Xgv[6];
  float V[3];
  float R[9];
  float P_V[9];
  float sigma_rho_min;
  int ext_ok;
  int klm_num;
  int kf_matches;
  int reg_num;
  int lm_accept_mask;
  int status; /* 0 ok, 1 minimization NaN, 2 insufficient matches */
} rebvio_hip_pair_out;

int rebvio_hip_abi_version(void);
const char* rebvio_hip_last_error(void);

/* Defaults of the reference's config structs and Camera() for a rows x cols sensor. */
void rebvio_hip_default_params(rebvio_hip_params* p, int rows, int cols);

/* Replaces the constructors EdgeDetector(camera, config) (edge_detector.cpp:17-26), ScaleSpace /
 * FastGaussian (scale_space.cpp:14-41,184-190), Core(camera, config) + DistanceField (core.cpp:17-24,
 * core.hpp:22-28): allocates every device buffer once.
 * Sensor sizes: rows, cols >= 32, any width (the integral images get a row pitch of cols rounded up to 4 internally),
 * cols <= 4096, rows <= 2548 (LDS-staged scan strips); -3 otherwise.
 * Also -3: search_range > 255 or search_range + 2 * pixel_uncertainty_match >= 258 (per-wave probe sequence buffer),
 * quantile_num_bins outside 1..128, keylines_max * 2 * search_range >= 2^23 (distance-field key encoding),
 * keylines_max outside 1..65536 (the LM reduction stages at most 256 record groups of 256 keylines in LDS). */
int rebvio_hip_create(const rebvio_hip_params* p, rebvio_hip_ctx** out);
void rebvio_hip_destroy(rebvio_hip_ctx* ctx);

/* ScaleSpace::build (scale_space.cpp:203-208) on a host fp32 image (0..765); any output may be NULL.
 * Test/diagnostic entry: detect() runs the same kernels without the downloads. */
int rebvio_hip_scale_space(rebvio_hip_ctx* ctx, const float* img_host, float* scale0, float* scale1, float* dog,
                           float* mag);

/* EdgeDetector::detect (edge_detector.cpp:30-43): threshold servo, buildEdgeMap, joinEdges,
 * tuneThreshold. The image is the undistorted fp32 frame the reference passes in types::Image
 * (rebvio.cpp:43-47). Returns a pooled map (release with rebvio_hip_map_release). Asynchronous: the
 * keyline count is fetched by rebvio_hip_map_size. */
int rebvio_hip_detect(rebvio_hip_ctx* ctx, const float* img_host, size_t pitch_bytes, uint64_t ts_us,
                      rebvio_hip_map** out);
/* Same, for a u8 frame already resident in device memory: fuses convertTo(CV_32F, 3.0) (rebvio.cpp:43)
 * into the first scan kernel. `frame_dev` is a device pointer to rows*cols bytes, dense. */
int rebvio_hip_detect_u8_device(rebvio_hip_ctx* ctx, const uint8_t* frame_dev, uint64_t ts_us, rebvio_hip_map** out);
/* Same, for a u8 frame in host memory (the MONO8 image ros_rebvio hands to imageCallback, ros_rebvio.cpp:96-104):
 * uploads 1 byte/pixel instead of the fp32 frame. */
int rebvio_hip_detect_u8(rebvio_hip_ctx* ctx, const uint8_t* img_host, size_t pitch_bytes, uint64_t ts_us,
                         rebvio_hip_map** out);
/* Lens model of the front end (camera.hpp:39-40,54-58: cv::undistort(in, out, K(fm,0,cx;0,fm,cy), D(k1,k2,p1,p2,k3))).
 * K4 = fx, fy, cx, cy; D5 = k1, k2, p1, p2, k3. Once set, every *_u8 detect entry runs convertTo(CV_32F,3.0) +
 * undistort on the device before the scale space (rebvio.cpp:43-47); all-zero D5 switches it off. Synchronises. */
int rebvio_hip_set_undistort(rebvio_hip_ctx* ctx, const float K4[4], const float D5[5]);
/* The front end alone: u8 host frame -> undistorted fp32 host frame (rows*cols floats). Needs a lens model. */
int rebvio_hip_front_end_u8(rebvio_hip_ctx* ctx, const uint8_t* img_host, float* out_host);
/* config_->threshold after the servo and auto_threshold_ (edge_detector.hpp:84,91). Synchronises. */
int rebvio_hip_detector_state(rebvio_hip_ctx* ctx, float* threshold, float* auto_threshold, int* keylines_count);

/* EdgeMap accessors (edge_map.hpp:40-75). size/threshold synchronise with the detect stream. */
int rebvio_hip_map_size(rebvio_hip_map* m);
float rebvio_hip_map_threshold(rebvio_hip_map* m);
uint64_t rebvio_hip_map_ts(rebvio_hip_map* m);
/* Lazy host mirror of keylines() / mask() (edge_map.hpp:50,75): AoS keylines (may be NULL) and the
 * dense image-index -> keyline-index table (may be NULL; -1 = none). A map fresh from detect() only waits for its own detection;
 * once a tracking call has been given the map the download waits for the track stream as well. Downloading a map from one
 * thread WHILE another thread's tracking call works on it mirrors some state in between (as reading a reference EdgeMap
 * during its tracking step would). */
int rebvio_hip_map_download(rebvio_hip_map* m, rebvio_hip_keyline* keylines, int* mask);
/* Edge image for registerEdgeImageCallback consumers, rendered on the device the way ros_rebvio.cpp:32-50 draws it on
 * the host: the grey frame (rows*cols bytes, NULL = black) replicated to RGB, each keyline's pixel
 * (round(pos[1]), round(pos[0])) set to (255,0,0). rgb_out = rows*cols*3 bytes. Synchronises. */
int rebvio_hip_render_edge_image(rebvio_hip_map* map, const uint8_t* gray_host, uint8_t* rgb_out_host);

/* Test hook: overwrite the device keylines (count must equal the map size). */
int rebvio_hip_map_upload(rebvio_hip_map* m, const rebvio_hip_keyline* keylines, int n);
/* Map handles may outlive their context: after rebvio_hip_destroy the entries that take a map alone (size, threshold, download,
 * render, upload, map_distance_field) fail with -10 (NaN for the threshold) and rebvio_hip_map_release frees what is left of the
 * handle - nothing of the destroyed context is touched. (A handle must still be released exactly once.) */
void rebvio_hip_map_release(rebvio_hip_map* m);

/* Core::buildDistanceField / DistanceField::build (core.cpp:33-37, core.hpp:37-59). */
int rebvio_hip_build_distance_field(rebvio_hip_ctx* ctx, rebvio_hip_map* m);
/* DistanceField::operator[] for all cells (core.hpp:61): ids (-1 = none) and distances (valid where id >= 0). */
int rebvio_hip_distance_field(rebvio_hip_ctx* ctx, int* id_out, int* dist_out);

/* rebvio::DistanceField::operator[] (core.hpp:61) for the field built FROM map `m` (by detect, or by
 * rebvio_hip_build_distance_field): the field lives with its map, so it stays readable while the map is alive. */
int rebvio_hip_map_distance_field(rebvio_hip_map* m, int* id_out, int* dist_out);

/* EdgeMap::searchMatch (edge_map.hpp:93-94, edge_map.cpp:101-184): searched->searchMatch(query, vel, Rvel, Rback, max_radius),
 * one keyline of another map against `searched`. vel / Rvel are used as given (directedMatch passes them rotated by Rback,
 * edge_map.cpp:193-194). *idx_out = index of the match in `searched`, or -1. Synchronises. */
int rebvio_hip_search_match(rebvio_hip_ctx* ctx, rebvio_hip_map* searched, const rebvio_hip_keyline* query, const float vel[3],
                            const float Rvel[9], const float Rback[9], float max_radius, int* idx_out);

/* FastGaussian::smooth (scale_space.hpp:31, scale_space.cpp:173-182) on a host fp32 image: three integral-image box
 * passes of the given (odd, 3..11) widths, as FastGaussian's constructor derives them from sigma (scale_space.cpp:14-41).
 * Test/diagnostic entry like rebvio_hip_scale_space. */
int rebvio_hip_smooth(rebvio_hip_ctx* ctx, const float* img_host, const int widths3[3], float* out_host);
/* The same for FastGaussian's general n (scale_space.cpp:14-41 takes any number of box passes; the reference itself only uses
 * n = 3, scale_space.cpp:186): n in 1..16 passes of the given odd widths in 3..11. */
int rebvio_hip_smooth_n(rebvio_hip_ctx* ctx, const float* img_host, const int* widths, int n, float* out_host);

/* EdgeMap::rotateKeylines (edge_map.cpp:58-71); R row-major 3x3. */
int rebvio_hip_rotate(rebvio_hip_ctx* ctx, rebvio_hip_map* m, const float R[9]);
/* EdgeMap::estimateQuantile (edge_map.cpp:39-56). */
int rebvio_hip_quantile(rebvio_hip_ctx* ctx, rebvio_hip_map* m, float percentile, int num_bins, float* out);
/* Core::tryVel (core.cpp:78-148). `residuals` (host, map-size floats) is read and updated like the
 * reference's `_residuals`; out10 = score, JtJ(0,0),(1,1),(2,2),(0,1),(0,2),(1,2), JtF[0..2]. */
int rebvio_hip_try_vel(rebvio_hip_ctx* ctx, rebvio_hip_map* m, const float vel[3], float sigma_rho_min,
                       float* residuals, float out10[10]);
/* Core::minimizeVel (core.cpp:150-189) with the Levenberg-Marquardt loop kept on the device.
 * vel is in/out; Rvel = invert(JtJ); returns score in *F. */
int rebvio_hip_minimize_vel(rebvio_hip_ctx* ctx, rebvio_hip_map* m, float vel[3], float Rvel[9], float* F,
                            int* accept_mask, float* sigma_rho_min);
/* EdgeMap::forwardMatch (edge_map.cpp:73-99): old_map->forwardMatch(new_map). */
int rebvio_hip_forward_match(rebvio_hip_ctx* ctx, rebvio_hip_map* old_map, rebvio_hip_map* new_map);
/* Core::extRotVel (core.cpp:191-261) on the distance field's map. JtJ 6x6 row-major, JtF[6], X[6]. */
int rebvio_hip_ext_rot_vel(rebvio_hip_ctx* ctx, const float vel[3], float Wx[36], float JtF[6], float X[6], int* ok);
/* EdgeMap::directedMatch (edge_map.cpp:186-218): new_map->directedMatch(old_map, ...). */
int rebvio_hip_directed_match(rebvio_hip_ctx* ctx, rebvio_hip_map* new_map, rebvio_hip_map* old_map, const float vel[3],
                              const float Rvel[9], const float Rback[9], float max_radius, int* matches,
                              int* kf_matches);
/* EdgeMap::regularize1Iter (edge_map.cpp:220-259). */
int rebvio_hip_regularize(rebvio_hip_ctx* ctx, rebvio_hip_map* m, int* count);
/* Core::updateInverseDepth (core.cpp:417-456) on the distance field's map. */
int rebvio_hip_update_inverse_depth(rebvio_hip_ctx* ctx, const float vel[3]);

/* Gyro-bias state of the frame-pair glue (types/imu.hpp:180-183) back to its initial value. */
void rebvio_hip_reset_state(rebvio_hip_ctx* ctx);
/* imu_state_.Bg / imu_state_.W_Bg (types/imu.hpp:180-182): read / set by the orchestrator's gyro-bias initialisation
 * (rebvio.cpp:146-160). */
int rebvio_hip_get_gyro_state(rebvio_hip_ctx* ctx, float Bg[3], float W_Bg[9]);
int rebvio_hip_set_gyro_state(rebvio_hip_ctx* ctx, const float Bg[3], const float W_Bg[9]);
/* One frame pair, rebvio.cpp:142-259 (accelerometer/SAB branch excluded): distance field of new_map
 * (if not yet built), rotate, minimizeVel, forwardMatch, extRotVel, gyroBiasCorrection, rotate,
 * directedMatch, regularize1Iter, updateInverseDepth. R_prior = IMU inter-frame rotation or NULL. */
int rebvio_hip_track_pair(rebvio_hip_ctx* ctx, rebvio_hip_map* old_map, rebvio_hip_map* new_map, const float* R_prior,
                          float frame_dt, rebvio_hip_pair_out* out);

/* The same step in two halves, for hosts that fuse inertial data between them (rebvio.cpp:206-233: accelerometer /
 * scale-attitude-bias filter decides the final V, Rgva and the second rotation):
 *   begin  = rebvio.cpp:142-191: distance field, rotate by the gyro prior, minimizeVel, forwardMatch, extRotVel,
 *            gyroBiasCorrection. Returns Vg, P_Vg, Xv, W_Xv, Xgv, W_Xgv (after correction) and R (prior corrected by Bg).
 *   finish = rebvio.cpp:223/232 + 236-259: rotate the old map by R_second, directedMatch with (V, P_V, Rgva),
 *            regularize1Iter, updateInverseDepth; status 0 / 1 (NaN in V) / 2 (< global_min_matches_threshold). */
typedef struct rebvio_hip_pair_mid {
  float Vg[3];
  float P_Vg[9];
  float F;
  float sigma_rho_min;
  int lm_accept_mask;
  int ext_ok;
  float Xv[6];
  float W_Xv[36];
  float Xgv[6];
  float W_Xgv[36];
  float R[9];
} rebvio_hip_pair_mid;
int rebvio_hip_track_pair_begin(rebvio_hip_ctx* ctx, rebvio_hip_map* old_map, rebvio_hip_map* new_map, const float* R_prior,
                                float frame_dt, rebvio_hip_pair_mid* mid);
int rebvio_hip_track_pair_finish(rebvio_hip_ctx* ctx, rebvio_hip_map* old_map, rebvio_hip_map* new_map, const float V[3],
                                 const float P_V[9], const float Rgva[9], const float R_second[9], int* klm_num,
                                 int* kf_matches, int* reg_num, int* status);
/* _finish in two steps, so that a host can queue the NEXT pair's first half behind this pair's second half before it waits for
 * anything: _finish_async hands over the fusion's results and returns at once; _result waits for the second half and returns
 * its counters / status (what rebvio.cpp:245-259 needs for the "insufficient matches" stop and what the odometry record carries).
 * Allowed order: begin(k), finish_async(k), begin(k+1), result(k), finish_async(k+1), ... - one result may be outstanding.
 * Between _begin and _finish(_async) of a pair only _result of the previous pair may be called.
 * A pair's match counters ride to the host in the NEXT pair's first-half record when that pair continues from this pair's new
 * map (no copy, no extra wait); otherwise _begin / _result copy them. (Releasing that map before _result is allowed: the
 * release copies them out first.) */
/* R_prior_next (may be NULL): the IMU inter-frame rotation the NEXT pair's _begin will be given as R_prior, when the caller
 * already has it. That pair's first rotateKeylines (rebvio.cpp:163-165) then runs inside this pair's last kernel and its
 * _begin launches one kernel less. R_prior_next is a promise: the old map is rotated in place with it, so the next _begin must
 * be given the same R_prior and must find the gyro state this pair's _begin left (no rebvio_hip_set_gyro_state / _reset_state
 * in between); otherwise it returns -7 and that pair cannot be tracked from this map any more. */
int rebvio_hip_track_pair_finish_async(rebvio_hip_ctx* ctx, rebvio_hip_map* old_map, rebvio_hip_map* new_map, const float V[3],
                                       const float P_V[9], const float Rgva[9], const float R_second[9], const float* R_prior_next);
int rebvio_hip_track_pair_result(rebvio_hip_ctx* ctx, int* klm_num, int* kf_matches, int* reg_num, int* status);
/* Optional, before _finish(_async) of a pair: the map the NEXT pair will track as its new map (already handed to detect).
 * The track stream's wait for that map's detection is queued now, ahead of this pair's second half, instead of between the two
 * pairs (one barrier packet less on the pair-to-pair path). The reference has no counterpart: its maps are host objects. */
int rebvio_hip_track_pair_hint_next(rebvio_hip_ctx* ctx, rebvio_hip_map* next_new_map);

/* Streaming driver used by the bench: a software pipeline, detect(frame) on the scan / keyline streams overlapped with
 * the tracking of EARLIER pairs on the track stream. A pair runs on the device from end to end: the glue between its halves
 * (6x6 solve, gyroBiasCorrection, SO3, covariance: rebvio.cpp:177-233 without the accelerometer branch) is evaluated in front
 * of the directedMatch kernel from the first half's records, with the gyro-bias state kept in device memory, so the host only
 * queues launches and reads records. `out` receives the oldest COMPLETE pair not handed out yet, in pair order, several calls
 * behind `frame` (the detect stage leads the tracker by REBVIO_HIP_LEAD frames, default 5; pairs are queued on the track stream
 * in groups of REBVIO_HIP_GROUP, default 4, with one stream wait and one event per group; a pair's match counters arrive with
 * the next pair; up to fifteen pairs are in flight between the device and the host); status -1 while there is none.
 * rebvio_hip_flush() tracks the pairs not started yet and finishes everything in flight: a stream of n frames yields n - 1
 * records; those not handed out by a push are fetched with rebvio_hip_next_record: 1 = *out / *keylines filled, 0 = none left.
 * rebvio_hip_get_gyro_state follows the stream with that lag and is exact after a flush; rebvio_hip_set_gyro_state is refused
 * (-7) while frames are in flight. */
int rebvio_hip_push_frame_u8_device(rebvio_hip_ctx* ctx, const uint8_t* frame_dev, uint64_t ts_us,
                                    rebvio_hip_pair_out* out, int* keylines);
/* The same for a MONO8 frame in HOST memory (what imageCallback is handed, rebvio.cpp:38-48): copied into a pinned ring here
 * (the caller's buffer is free when the call returns) and from there to the device ahead of the frame's scans, in stream order. */
int rebvio_hip_push_frame_u8(rebvio_hip_ctx* ctx, const uint8_t* frame_host, size_t pitch_bytes, uint64_t ts_us,
                             rebvio_hip_pair_out* out, int* keylines);
int rebvio_hip_next_record(rebvio_hip_ctx* ctx, rebvio_hip_pair_out* out, int* keylines);
/* Frame pairs the streaming driver has queued on the device so far (a measurement aid: pairs are queued in groups, so a short
 * window of pushes may start a few pairs more or fewer than it pushes frames). */
uint64_t rebvio_hip_pairs_started(rebvio_hip_ctx* ctx);
int rebvio_hip_flush(rebvio_hip_ctx* ctx);

/* Several camera streams on ONE GPU, advanced in lock-step ("lanes" of a batch). The reference runs one rebvio::Rebvio per
 * camera stream, each with its own detector, tracker and queues (rebvio.hpp:91-112); a batch is `lanes` such pipelines whose
 * per-frame kernels are launched once per step for all lanes (lane = blockIdx.z). One 640x480 stream keeps well under a tenth
 * of an MI355X busy - its kernels are short latency chains - so a batched step costs about what a single stream's step costs
 * and the frame rate scales with the lane count until the chip fills. Every lane is a full context (own maps, servo, gyro-bias
 * state) sharing the batch's three streams; its records are bit-identical to those of a stand-alone context fed the same
 * frames. lanes in 1..16, any keylines_max a context accepts. The persistent tracking kernel's workgroups exchange records
 * within a lane and must be resident together for that: the driver launches it for as many lanes at a time as the device holds
 * (8 lanes of 16k keylines on an MI355X; 16 lanes take two such launches per step), and rebvio_hip_batch_create refuses (-3)
 * a keylines_max whose single lane does not fit. A lens model (rebvio_hip_set_undistort on every lane's context, or on none) puts the batched front end
 * (x3 + undistort, rebvio.cpp:43-47) ahead of the scans, each lane through its own model.
 * push: frames_dev[l] = this step's u8 frame of lane l in device memory (all lanes share ts_us); out[l] / keylines[l] receive
 * lane l's oldest COMPLETE pair not handed out yet, like rebvio_hip_push_frame_u8_device (status -1 while there is none);
 * after rebvio_hip_batch_flush the remaining steps' records are fetched with rebvio_hip_batch_next_records (1 = filled, 0 = none).
 * A push that fails after some lanes have been prepared leaves the batch out of lock-step: it is marked and every later call
 * returns -11. */
typedef struct rebvio_hip_batch rebvio_hip_batch;
int rebvio_hip_batch_create(const rebvio_hip_params* p, int lanes, rebvio_hip_batch** out);
void rebvio_hip_batch_destroy(rebvio_hip_batch* b);
int rebvio_hip_batch_lanes(rebvio_hip_batch* b);
/* The context of one lane (device_alloc / device_upload for its frames, detector_state, get/set_gyro_state, ...). Owned by the batch. */
rebvio_hip_ctx* rebvio_hip_batch_lane(rebvio_hip_batch* b, int lane);
int rebvio_hip_batch_push_u8_device(rebvio_hip_batch* b, const uint8_t* const* frames_dev, uint64_t ts_us, rebvio_hip_pair_out* out,
                                    int* keylines);
int rebvio_hip_batch_next_records(rebvio_hip_batch* b, rebvio_hip_pair_out* out, int* keylines);
int rebvio_hip_batch_flush(rebvio_hip_batch* b);

/* Test hook: the glue of one pair (rebvio.cpp:177-233 without the accelerometer branch: sum of the extRotVel block records, 6x6
 * solve, gyroBiasCorrection, SO3 correction, covariance, next prior) evaluated on the device, as the streaming and batch drivers
 * run it, AND on the host, as rebvio_hip_track_pair runs it, from the same inputs: final minimizeVel state (vel, the six
 * unique JtJ entries (0,0) (1,1) (2,2) (0,1) (0,2) (1,2), F, sigma_rho_min, accept mask), ceil(n_new / 256) extRotVel records of
 * 32 floats, frame_dt, filter state and prior rotation. Outputs per side: the pair record, the filter state after the pair
 * (22 floats: Bg, W_Bg, next prior rotation, pad) and the second half's inputs (44 words). The two sides must agree bit for bit. */
int rebvio_hip_test_glue(rebvio_hip_ctx* ctx, const float vel[3], const float JtJ6[6], float F, float sigma_rho_min, int accept_mask,
                         const float* xrv, int n_new, float frame_dt, const float Bg[3], const float W_Bg[9], const float R_prior[9],
                         rebvio_hip_pair_out* out_dev, float* state_dev, float* second_dev, rebvio_hip_pair_out* out_host,
                         float* state_host, float* second_host);

/* Test hook for the sequence stamps of the host-visible pair records. Every pair the library queues carries a non-zero sequence
 * number; the pair's kernels store it as the LAST word of each record they write into host-visible memory (result slot, glue
 * record), and every entry that reads such a record - rebvio_hip_track_pair, _track_pair_begin, _push_frame_u8(_device),
 * _next_record's producer, _flush and the batch forms - compares it first: a record read before its pair wrote it (a completion
 * event or a synchronisation that reported too early) fails with status -12 instead of handing out stale poses. This hook hands
 * the kernels of the NEXT pair (batch: the last lane of the next step) a wrong number, which is exactly what such a record looks
 * like to the reader. */
int rebvio_hip_test_forge_record_stamp(rebvio_hip_ctx* ctx);
int rebvio_hip_batch_test_forge_record_stamp(rebvio_hip_batch* b);

/* Per-kernel device timing of the last N launches of each kernel, measured with HIP events on the
 * stream the kernel runs on. names: '\n'-separated. Used by bench.py's roofline leg. */
int rebvio_hip_profile_enable(rebvio_hip_ctx* ctx, int on); /* 0 off, 1 every launch, N>1 every N-th launch */
int rebvio_hip_profile_select(rebvio_hip_ctx* ctx, const char* only_kernel); /* NULL/"" = every kernel */
int rebvio_hip_profile_reset(rebvio_hip_ctx* ctx);
int rebvio_hip_profile_read(rebvio_hip_ctx* ctx, char* names, size_t names_cap, double* avg_us, int* calls, int cap);

/* Device memory helpers so that hosts without a HIP runtime binding (ctypes, cgo) can stage frames. */
int rebvio_hip_device_alloc(rebvio_hip_ctx* ctx, size_t bytes, void** out);
int rebvio_hip_device_free(rebvio_hip_ctx* ctx, void* p);
int rebvio_hip_device_upload(rebvio_hip_ctx* ctx, void* dst_dev, const void* src_host, size_t bytes);

#ifdef __cplusplus
}
#endif
#endif
